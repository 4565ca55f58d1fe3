"""The driver keeps only the tail of bench.py's stdout: the ONE line has to be short and strictly parseable (round 5's
28 KB line was cut mid-object and the round went unmeasured).  CPU tests of bench.compact_line on a real full result --
profiles/r05_final_bench_default.txt, the very line that was lost."""
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _fail_constant(name):
    raise AssertionError("non-standard JSON constant %r in the bench line" % name)


def _full_result():
    text = open(os.path.join(ROOT, "profiles", "r05_final_bench_default.txt")).read().strip().splitlines()[-1]
    return json.loads(text)


def test_default_line_is_short_strict_json_with_contract_keys():
    import bench

    full = _full_result()
    assert len(json.dumps(full)) > 20000  # (the mock really is the oversized result)
    line = bench.compact_line(full, bench.EXTRA_FILE)
    text = json.dumps(line, allow_nan=False)
    assert len(text) < bench.LINE_LIMIT
    back = json.loads(text, parse_constant=_fail_constant)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "extra"):
        assert k in back, k
    assert back["vs_baseline"] is None and back["dtype"] == "f32"
    assert set(back["config"]) >= {"workload", "solver", "launch"} and "model" not in back["config"]
    for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "algorithmic_bytes_per_launch", "mean_us"):
        assert k in back["roofline"], k
    assert abs(back["roofline"]["frac"] - back["roofline"]["achieved"] / back["roofline"]["peak"]) < 1e-12
    for k in ("value", "unit", "cores", "kind", "sample", "ms_per_step"):
        assert k in back["cpu_baseline"], k
    # nothing nested beyond the contract's objects: no other legs, no fidelity blobs, no histograms
    for k in ("other_configs", "run_loop", "real_plate", "newton_iters"):
        assert k not in back
    assert "fidelity" not in back["cpu_baseline"] and "other_kernels" not in back["roofline"]


def test_line_stays_short_with_multi_gpu_fields_and_missing_objects():
    import bench

    full = _full_result()
    full.update(n_gpus=8, rows_per_step=288, n_ranks_seen=8, eager_ms_per_step=0.2,
                strong_scaling_config3={"workload": "x" * 500, "scaling": "strong", "value": 1.0, "unit": "steps/s",
                                        "ms_per_step": 1.0, "steps": 50, "n_iwae_per_gpu": 125, "final_loss": 1.0},
                strong_scaling_config5={"error": "RuntimeError: " + "y" * 2000})
    full["roofline"] = None
    full["cpu_baseline"] = None
    full["note"] = "z" * 5000
    line = bench.compact_line(full, None)
    text = json.dumps(line, allow_nan=False)
    assert len(text) < bench.LINE_LIMIT
    back = json.loads(text, parse_constant=_fail_constant)
    assert back["roofline"] is None and back["cpu_baseline"] is None
    assert back["rows_per_step"] == 288 and back["n_ranks_seen"] == 8
    assert len(back["strong_scaling_config5"]["error"]) <= 120


def test_emit_line_writes_the_side_file_and_one_line(tmp_path, monkeypatch):
    import io

    import bench

    buf = io.StringIO()
    monkeypatch.setattr(bench, "_OUT", buf)
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    bench.emit_line(_full_result())
    out = buf.getvalue()
    assert out.count("\n") == 1 and len(out) < bench.LINE_LIMIT
    side = json.load(open(tmp_path / bench.EXTRA_FILE))
    assert "other_configs" in side and json.loads(out)["extra"] == bench.EXTRA_FILE


def test_stale_counter_files_are_not_attached(tmp_path):
    """A committed PMC reduction counts as this tree's only if it carries this tree's kernel-source fingerprint."""
    import bench

    sys.path.insert(0, os.path.join(ROOT, "profiles"))
    from srcsha import csrc_sha16

    for sha, fresh in ((csrc_sha16(), True), ("0" * 16, False), (None, False)):
        d = {"kernels": {"k": {"hbm_bytes_corrected": 1}}}
        if sha is not None:
            d["csrc_sha16"] = sha
        p = tmp_path / "x.json"
        p.write_text(json.dumps(d))
        ks, ok = bench.load_pmc(str(p))
        assert ok is fresh and "k" in ks
