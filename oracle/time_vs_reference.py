#!/usr/bin/env python3
"""Is the oracle a fair CPU baseline?  Time one training step of the *imported reference* (microsoft/vi-hds at
/root/reference, `Training._run_batch` semantics, training.py:324-340) and one step of the oracle
(oracle/vihds_oracle.py, exactly the closure bench.py's `cpu_baseline` times) on the same cores, same shape
(dr_constant_icml, B=36 rows, n_iwae=200, T=86), same solver.

Build container only ("timing vs the imported reference on this container's 8 cores for the same step,
reported as a ratio so a reader can see the restatement is not a strawman").  The reference runs with the stand-ins of
tests/golden/make_fixtures.py, so only `modeuler` / `modeulerwhile` are available for the ratio (torchdiffeq==0.1 is
absent); the oracle's rk4 step is timed beside it for scale.  Writes oracle/cpu_fidelity.json, which bench.py echoes
under `cpu_baseline.fidelity`.  Test infrastructure: nothing in the product path imports this.

usage: python oracle/time_vs_reference.py [--steps 5] [--threads 1 8]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = "/root/reference"


def reference_step_fn(solver, n_iwae, rows):
    """The reference's own objects, built the way tests/golden/make_fixtures.py builds them."""
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import make_fixtures as MF

    MF.install_standins()
    sys.path.insert(0, REF)
    cwd = os.getcwd()
    os.chdir(REF)
    os.environ.setdefault("INFERENCE_DATA_DIR", os.path.join(REF, "data"))
    os.environ.setdefault("INFERENCE_RESULTS_DIR", "/tmp/vihds_ref_results")
    import torch
    from munch import munchify
    from vihds.config import Config
    from vihds.datasets import build_datasets
    from vihds.parameters import Parameters
    from vihds.run_xval import create_parser
    from vihds.training import Training
    from vihds.vae import build_model

    MF.patch_merge_observations()
    args = create_parser(True).parse_args(
        ["--train_samples=%d" % n_iwae, "--test_samples=%d" % n_iwae, "--seed=0", "specs/dr_constant_icml.yaml"])
    settings = Config(args)
    settings.params.solver = solver
    data = build_datasets(args, settings)
    parameters = Parameters(settings.params)
    model = build_model(args, settings, data, parameters)
    training = Training(args, settings, data, parameters, model)
    model.train()
    full = training.train_data
    sel = slice(0, rows)
    batch = munchify({"devices": np.asarray(full.devices)[sel], "dev_1hot": full.dev_1hot[sel],
                      "inputs": full.inputs[sel], "observations": full.observations[sel], "times": full.times})
    os.chdir(cwd)
    shape = {"rows": int(len(batch.inputs)), "n_times": int(len(batch.times))}

    def one_step():  # training.py:329-337
        t0 = time.perf_counter()
        batch_results, theta, q, p = training.model(batch, n_iwae)
        elbo = training.cost(batch, batch_results, theta, q, p).elbo
        assert not torch.isnan(elbo)
        elbo.backward()
        training.optimizer.step()
        training.optimizer.zero_grad()
        return time.perf_counter() - t0

    return one_step, shape


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--threads", type=int, nargs="+", default=[1, 8])
    ap.add_argument("--solver", default="modeuler")
    a = ap.parse_args()
    for p in (ROOT, os.path.join(ROOT, "vi-hds_amd")):
        if p not in sys.path:
            sys.path.append(p)  # after the reference: `vihds` must resolve to /root/reference/vihds for the reference leg
    # the reference and the repo's package share the name `vihds`: time the reference in a child process
    if os.environ.get("VIHDS_TIME_LEG") == "reference":
        import torch

        step, shape = reference_step_fn(a.solver, 200, 36)
        out = {"shape": shape}
        for n in a.threads:
            torch.set_num_threads(n)
            step()
            out[str(n)] = float(np.median([step() for _ in range(a.steps)]))
        print("RESULT " + json.dumps(out))
        return
    import subprocess

    env = dict(os.environ, VIHDS_TIME_LEG="reference")
    r = subprocess.run([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env, capture_output=True, text=True)
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("RESULT ")]
    if not line:
        raise SystemExit("reference leg failed:\n" + r.stdout[-2000:] + r.stderr[-4000:])
    ref = json.loads(line[-1][7:])
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "vi-hds_amd"))
    import torch

    import bench

    rows = {}
    steps = {s: bench.make_oracle_step(s) for s in (a.solver, "rk4")}
    for n in a.threads:
        torch.set_num_threads(n)
        row = {"reference_%s_s_per_step" % a.solver: ref[str(n)]}
        for s, fn in steps.items():
            fn()
            row["oracle_%s_s_per_step" % s] = float(np.median([fn() for _ in range(a.steps)]))
        row["reference_over_oracle_%s" % a.solver] = row["reference_%s_s_per_step" % a.solver] / row["oracle_%s_s_per_step" % a.solver]
        rows["%d thread%s" % (n, "" if n == 1 else "s")] = {k: round(v, 4) for k, v in row.items()}
    out = {"what": "one full training step (encoder, sample/clip, integrate, observe, log-probs, IWAE loss, backward, Adam) "
                   "of dr_constant_icml at B=36 rows, n_iwae=200, T=%d: the imported reference (/root/reference, "
                   "Training._run_batch semantics, stand-ins of tests/golden/make_fixtures.py) vs the oracle closure "
                   "bench.py's cpu_baseline times; medians of %d steps after one warm-up, build container (%d CPUs)"
                   % (ref["shape"]["n_times"], a.steps, os.cpu_count()),
           "solver_of_the_ratio": a.solver,
           "note": "reference_over_oracle > 1 means the oracle is the FASTER of the two (a conservative baseline); the "
                   "reference cannot run rk4 here (torchdiffeq==0.1 absent), the oracle's rk4 step is listed for scale",
           "rows": rows, "script": "oracle/time_vs_reference.py", "torch": torch.__version__}
    json.dump(out, open(os.path.join(HERE, "cpu_fidelity.json"), "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
