"""CPU-only tests of the host side: the C ABI library loads and exports every symbol include/vihds_hip.h
declares, the YAML -> parameter tables match what the reference built, the batched encoder initialises to the
reference's weights under the same seed, and the product path refuses to run without a GPU."""
import os
import re

import numpy as np
import pytest
import torch

from fixture_util import ALL_FIXTURES as _PINNED, PATCHED_FIXTURES, Fixture, rel_err

ALL_FIXTURES = _PINNED + PATCHED_FIXTURES

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from vihds import hip

    header = open(os.path.join(ROOT, "include", "vihds_hip.h")).read()
    declared = sorted(set(re.findall(r"\b(vihds_[a-z_0-9]+)\s*\(", header)))
    assert declared, "no declarations parsed"
    lib = hip.lib()
    for name in declared:
        assert hasattr(lib, name), "libvihds_hip.so does not export %s" % name
    assert sorted(hip.exported_symbols()) == declared
    assert lib.vihds_abi_version() == 14


def test_problem_descriptor_agrees_in_header_binding_and_integration_doc():
    """struct vihds_ode_problem three times over: include/vihds_hip.h, the ctypes mirror the package binds with, and the
    stub INTEGRATION.md section 1 tells a maintainer to paste (round 5's stub lacked the last two fields: a struct 8 bytes
    short, the library reading kernel_variant past its end)."""
    import ctypes

    from vihds import hip

    header = open(os.path.join(ROOT, "include", "vihds_hip.h")).read()
    body = re.search(r"typedef struct vihds_ode_problem \{(.*?)\} vihds_ode_problem;", header, re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    h_fields = []
    for ctype, names in re.findall(r"\b(int|float)\s+([^;]+);", body):
        for n in names.split(","):
            m = re.match(r"\s*(\w+)\s*(\[\s*(\w+)\s*\])?\s*$", n)
            h_fields.append((m.group(1), ctype, m.group(3)))
    max_slots = int(re.search(r"#define\s+VIHDS_MAX_SLOTS\s+(\d+)", header).group(1))
    h_size = sum(4 * (max_slots if dim else 1) for _, _, dim in h_fields)
    py_fields = hip.OdeProblem._fields_
    assert [f[0] for f in py_fields] == [f[0] for f in h_fields]
    for (name, ct), (_, ctype, dim) in zip(py_fields, h_fields):
        base = ct._type_ if dim else ct
        assert base is (ctypes.c_int if ctype == "int" else ctypes.c_float), name
        assert (ct._length_ if dim else None) == (max_slots if dim else None), name
    assert ctypes.sizeof(hip.OdeProblem) == h_size
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    stub = re.search(r"class OdeProblem\(ctypes\.Structure\):.*?_fields_ = \[(.*?)\]\n", doc, re.S).group(1)
    d_fields = re.findall(r'\("(\w+)",\s*ctypes\.(c_int|c_float)(\s*\*\s*(\d+))?\)', stub)
    assert [f[0] for f in d_fields] == [f[0] for f in h_fields]
    for (name, ct, _, dim), (_, ctype, hdim) in zip(d_fields, h_fields):
        assert ct == "c_" + ctype, name
        assert (int(dim) if dim else None) == (max_slots if hdim else None), name


def test_model_slot_tables():
    from vihds import hip

    dr = hip.model_slots("dr_constant")
    assert len(dr) == 37 and dr[:4] == ["r", "K", "tlag", "rc"] and dr[-4:] == ["prec_x", "prec_rfp", "prec_yfp", "prec_cfp"]
    assert "eS6" in hip.model_slots("dr_constant_v2") and "KR6" not in hip.model_slots("dr_constant_v2")
    assert hip.lib().vihds_model_n_states(hip.MODELS["relay_constant"]) == 12
    assert hip.lib().vihds_model_n_states(hip.MODELS["degrader_constant"]) == 11
    assert hip.lib().vihds_model_n_states(99) < 0


@pytest.mark.parametrize("name", ALL_FIXTURES)
def test_parameter_tables_match_reference(name):
    """names / order / kinds / prior (mu, prec) / clip bounds as the reference derived them from the same YAML."""
    import e2e_util as E
    from vihds.encoders import Encoder

    fx = Fixture(name)
    args, settings, data, parameters = E.build_from_fixture(fx)
    assert [d.name for d in parameters.ordered()] == fx.names
    assert [d.kind for d in parameters.ordered()] == fx.kinds
    enc = Encoder(parameters, data, False, device="cpu")
    _, pm, pp = enc.p.image("cpu", 1)
    live = torch.tensor([k != 2 for k in fx.kinds])
    assert torch.equal(pm[:, 0][live], fx.t("p_mu")[live])
    assert torch.allclose(pp[:, 0][live], fx.t("p_prec")[live], rtol=1e-7)
    # clip at 4 sigma reproduces the reference's clipped samples exactly where clipping was active
    lo, hi = enc.p.clip_image(4.0, "cpu")
    un, cl = fx.t("theta_unclipped"), fx.t("theta")
    assert torch.allclose(torch.minimum(torch.maximum(un, lo[:, None, None]), hi[:, None, None]), cl, rtol=1e-6)


@pytest.mark.parametrize("name", ["dr_constant_icml_tiny_modeuler", "dr_blackbox_icml_tiny_modeuler",
                                  "dr_constant_one_modeuler", "dr_constant_precisions_tiny_modeuler"])
def test_encoder_initialises_to_reference_weights_and_q(name):
    """Same torch seed => the batched heads hold exactly the weights of the reference's per-parameter Linear(n,1)
    heads, and q(mu, prec) for the fixture batch equals the reference's."""
    import e2e_util as E
    from vihds.vae import build_model

    fx = Fixture(name)
    args, settings, data, parameters = E.build_from_fixture(fx)
    model = build_model(args, settings, data, parameters)
    enc = model.encoder
    ref = {k[len("encoder_param/"):]: fx.t(k) for k in fx.z.files if k.startswith("encoder_param/")}
    assert torch.equal(enc.conditional.conv.weight, ref["conditional.conv.weight"])
    assert torch.equal(enc.conditional.lin.weight, ref["conditional.lin.weight"])
    nl, ng = len(enc.local), len(enc.gcond)  # heads are stored [all mu ; all log_prec]
    for i, d in enumerate(enc.local):
        assert torch.equal(enc.local_heads.weight[i], ref["q_local_defs.%s.layers.mu.weight" % d.name][0])
        assert torch.equal(enc.local_heads.weight[nl + i], ref["q_local_defs.%s.layers.log_prec.weight" % d.name][0])
        assert torch.equal(enc.local_heads.bias[i], ref["q_local_defs.%s.layers.mu.bias" % d.name][0])
        assert torch.equal(enc.local_heads.bias[nl + i], ref["q_local_defs.%s.layers.log_prec.bias" % d.name][0])
    for i, d in enumerate(enc.gcond):
        assert torch.equal(enc.gcond_heads.weight[i], ref["q_global_cond_defs.%s.layers.mu.weight" % d.name][0])
        assert torch.equal(enc.gcond_heads.weight[ng + i], ref["q_global_cond_defs.%s.layers.log_prec.weight" % d.name][0])
    for i, d in enumerate(enc.glob):
        assert float(enc.global_free[0, i]) == float(ref["q_global_defs.%s.free_params.mu" % d.name])
        assert float(enc.global_free[1, i]) == float(ref["q_global_defs.%s.free_params.log_prec" % d.name])
    # decoder-side neural weights follow in the same RNG stream
    dref = {k[len("decoder_param/"):]: fx.t(k) for k in fx.z.files if k.startswith("decoder_param/")}
    for k, v in dict(model.decoder.named_parameters()).items():
        assert torch.equal(v.detach(), dref[k]), k
    q = enc(E.batch_from_fixture(fx, "cpu"))
    _, q_mu, q_prec = q.image("cpu", fx.B)
    live = torch.tensor([k != 2 for k in fx.kinds])
    assert rel_err(q_mu[live], fx.t("q_mu")[live], dim=0) < 1e-5
    assert rel_err(q_prec[live], fx.t("q_prec")[live], dim=0) < 1e-5
    assert q.get_tensor_names()[:2] == ["%s.mu" % fx.names[0], "%s.prec" % fx.names[0]]


def test_product_path_refuses_cpu_tensors():
    """No CPU fallback: the ops raise instead of silently computing elsewhere."""
    from vihds import ops

    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.IwaeRows.apply(torch.zeros(4, 2, 3), None, None)
    fx = Fixture("dr_constant_icml_tiny_modeuler")
    import hip_util as H

    th, row_of = H.pack_theta(fx, "cpu")
    spec = H.spec_for(fx, row_of, th.shape[0])
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.OdeSolveObserve.apply(spec, th, fx.t("inputs"), fx.t("times"), fx.t("observations"), None, None)


def test_lookup_has_reference_keys():
    import models

    keys = {"debug_constant", "auto_constant", "auto_constant_precisions", "degrader_constant_precisions",
            "dr_constant", "dr_constant_v2", "dr_constant_precisions", "dr_constant_precisions_v2", "dr_blackbox",
            "inducer_constant", "inducer_constant_precisions", "prpr_constant", "prpr_constant_precisions",
            "relay_constant", "relay_constant_precisions"}  # reference models/__init__.py:19-35
    assert keys <= set(models.LOOKUP)
    from vihds import hip

    for k in keys:  # every reference key maps to a kernel model of the C ABI
        assert getattr(models.LOOKUP[k], "model_key", None) in hip.MODELS, k


def test_device_conditioner_reproduces_reference_values():
    """aR/aS of the full fixture = 1 + relu(w . (dev_1hot * relevance)) with the reference's tiling quirk."""
    import e2e_util as E
    from vihds.vae import build_model

    fx = Fixture("dr_constant_icml_full_modeuler")
    args, settings, data, parameters = E.build_from_fixture(fx)
    model = build_model(args, settings, data, parameters)
    np.random.seed(fx.cfg["seed"] + 1)
    torch.manual_seed(fx.cfg["seed"] + 1)
    ode = model.decoder.ode_model
    ones = torch.ones(fx.B, fx.S)
    aR = ode.device_conditioner(ones, "aR", fx.t("dev_1hot"))
    aS = ode.device_conditioner(ones, "aS", fx.t("dev_1hot"))
    assert rel_err(aR, fx.t("extra_theta")[0]) < 1e-6
    assert rel_err(aS, fx.t("extra_theta")[1]) < 1e-6


def test_philox_known_answers():
    """Random123 known-answer vectors for philox4x32-10 pin the numpy restatement the kernel RNG is checked against."""
    from philox_ref import philox4x32_10

    kat = [((0, 0, 0, 0), (0, 0), (0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8)),
           ((0xFFFFFFFF,) * 4, (0xFFFFFFFF,) * 2, (0x408F276D, 0x41C83B0E, 0xA20BC7C6, 0x6D5451FD)),
           ((0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344), (0xA4093822, 0x299F31D0),
            (0xD16CFE09, 0x94FDCCEB, 0x5001E420, 0x24126EA1))]
    for ctr, key, want in kat:
        got = philox4x32_10(*ctr, *key)
        assert tuple(int(g) for g in got) == want


def test_run_batch_with_nan_check_disabled():
    """nan_check_every: 0 means "never check" (INTEGRATION.md); Training._run_batch must not divide by it.  (Until round 3
    hip_graph with the reference's host-side random streams was refused; round 4 stages them -- vihds/hostdraws.py, covered on
    the GPU by test_training_run_tracks_reference_trace, which runs graph replays with numpy / CPU draws by default.)"""
    from vihds import synthetic
    from vihds.utils import TrainingLogData

    args, settings, data, parameters, model, training = synthetic.build(
        "dr_constant_icml", 4, 3, solver="modeuler", device="cpu", seed=0, nan_check_every=0)
    assert training.nan_check_every == 0
    calls = []
    training.step = lambda batch: calls.append(1) or torch.tensor(float("nan"))
    log = TrainingLogData()
    assert training._run_batch(0.0, training.train_data, log) is True  # NaN not looked at, and no ZeroDivisionError
    training.nan_check_every = 1
    assert training._run_batch(0.0, training.train_data, log) is False
    assert len(calls) == 2
    assert training.use_graph is False  # (automatic graphs are for the GPU)


def test_fast_switch_sets_the_eight_keys_coherently(monkeypatch):
    """`params: {fast: true}` / VIHDS_FAST=1: one switch instead of eight keys; explicit keys win;
    without it every addition of this implementation keeps the reference's behaviour."""
    from vihds import config as C

    monkeypatch.delenv("VIHDS_FAST", raising=False)
    base = C.apply_defaults_params({"learning_rate": 0.01})
    assert (base.u_rng, base.conditioner_rng, base.hip_graph, base.nan_check_every) == ("numpy", "cpu", None, 1)
    assert base.fused_ode_training and base.fused_iwae_backward and base.fused_step_tail  # (value-preserving: on by default)
    fast = C.apply_defaults_params({"fast": True, "nan_check_every": 7})
    for k, v in C.FAST_PARAMS.items():
        assert fast[k] == (7 if k == "nan_check_every" else v), k
    monkeypatch.setenv("VIHDS_FAST", "1")
    env = C.apply_defaults_params({"learning_rate": 0.01})
    assert all(env[k] == v for k, v in C.FAST_PARAMS.items())
    monkeypatch.setenv("VIHDS_FAST", "0")
    assert C.apply_defaults_params({"fast": True}).hip_graph is None


def test_native_numpy_normal_stream_is_bit_identical():
    """vihds.nprand.randn_f32 (csrc/host/vihds_nprand.cpp) against np.random.randn(...).astype(float32) of the reference
    (vihds/vae.py:22-24): the same numbers and the same global RandomState afterwards -- odd and even sizes (the cached
    second deviate), a pending cached deviate, sizes around the generator's 624-word blocks, numpy calls in between."""
    from vihds import nprand

    if not nprand.available():
        pytest.skip("libvihds_host.so not built")
    for seed in (0, 7):
        sizes = [(1,), (7,), (36, 200, 35), (3,), (1,), (18, 200, 35), (2,), (623,), (624,), (625,), (4097, 2), (5,)]
        np.random.seed(seed)
        ref = [np.random.randn(*s_).astype(np.float32) for s_ in sizes]
        mid_ref = np.random.rand(3)
        ref2 = np.random.randn(11).astype(np.float32)
        st_ref = np.random.get_state()
        np.random.seed(seed)
        got = [nprand.randn_f32(s_) for s_ in sizes]
        mid = np.random.rand(3)
        got2 = nprand.randn_f32((11,))
        st = np.random.get_state()
        assert all(np.array_equal(a, b) and a.shape == b.shape for a, b in zip(ref, got))
        assert np.array_equal(mid_ref, mid) and np.array_equal(ref2, got2)
        assert np.array_equal(st_ref[1], st[1]) and st_ref[2:] == st[2:]
    # consecutive even-sized draws take the in-place path (numpy's state read and advanced where it lives); a foreign draw
    # in between, a re-seed and an odd size fall back to get_state / set_state -- the numbers must not care
    np.random.seed(3)
    ref = [np.random.randn(252).astype(np.float32), np.random.randn(1000).astype(np.float32), np.random.randn(3),
           np.random.randn(64).astype(np.float32), np.random.randn(7).astype(np.float32), np.random.randn(8).astype(np.float32),
           np.random.randn(8).astype(np.float32)]
    st_ref = np.random.get_state()
    np.random.seed(3)
    got = [nprand.randn_f32((252,)), nprand.randn_f32((1000,)), np.random.randn(3), nprand.randn_f32((64,)),
           nprand.randn_f32((7,)), nprand.randn_f32((8,)), nprand.randn_f32((8,))]
    st = np.random.get_state()
    assert all(np.array_equal(a, b) for a, b in zip(ref, got))
    assert np.array_equal(st_ref[1], st[1]) and st_ref[2:] == st[2:]
    np.random.seed(4)
    a = np.random.randn(10).astype(np.float32)
    np.random.seed(4)
    assert np.array_equal(a, nprand.randn_f32((10,)))
    # the same draw on the library's helper thread (start / finish: what a captured step's next draw uses), a started draw
    # that nobody finishes, and other thread counts than the default
    sh = (36, 200, 35)
    np.random.seed(5)
    ref = [np.random.randn(*sh).astype(np.float32) for _ in range(4)]
    tail_ref = np.random.randn(2).astype(np.float32)
    np.random.seed(5)
    draw = nprand.Draw(sh)
    bufs = [np.empty(sh, np.float32) for _ in range(4)]
    draw(bufs[0])
    assert draw.start(bufs[1]) and draw.start(bufs[2])  # (two may be queued: the helper goes from one into the next)
    assert not draw.start(np.empty(sh, np.float32))      # (... a third may not)
    draw.finish()
    draw.finish()
    assert draw.start(bufs[3])  # ... collected by the next call, whoever makes it
    tail = nprand.randn_f32((2,))
    assert all(np.array_equal(a, b) for a, b in zip(ref, bufs)) and np.array_equal(tail_ref, tail)
    assert not nprand.Draw((3,)).start(np.empty(3, np.float32))  # (an odd count leaves a cached deviate: not startable)
    keep = nprand._THREADS
    try:
        for th in (1, 3, 16):
            nprand._THREADS = th
            np.random.seed(8)
            a = [np.random.randn(*s_).astype(np.float32) for s_ in ((36, 200, 35), (8193,), (2,))]
            np.random.seed(8)
            assert all(np.array_equal(x, nprand.randn_f32(s_)) for x, s_ in zip(a, ((36, 200, 35), (8193,), (2,))))
    finally:
        nprand._THREADS = keep
    # a long stream: the float32 values come from a vectorised log with libm's log wherever the float32 rounding could
    # depend on the last bits (about 1.5e-5 of the values: some hundreds of them in this draw)
    np.random.seed(12)
    a = np.random.randn(6_000_000).astype(np.float32)
    st_ref = np.random.get_state()
    np.random.seed(12)
    b = nprand.randn_f32((6_000_000,))
    st = np.random.get_state()
    assert np.array_equal(a, b)
    assert np.array_equal(st_ref[1], st[1]) and st_ref[2:] == st[2:]


def test_prefetched_draw_collected_by_another_consumer_is_not_waited_for_twice():
    """a graph's prefetch bookkeeping (vihds/hostdraws.py `staged`) and nprand's count of draws in flight could
    desynchronise -- an evaluate() refresh, another graph's refresh, an eager draw or a training-state snapshot drains the
    helper's queue through nprand._collect_stray, and the owner's later finish() then waited for a draw that was not there
    (count -1, vihds_np_randn_f32_wait -> -3, RuntimeError).  The draws now carry an ownership token (nprand.generation):
    a draw somebody else had to wait for is left alone by its owner, its numbers are in place, and the stream is unchanged."""
    from vihds import nprand

    if not nprand.available():
        pytest.skip("libvihds_host.so not built")
    np.random.seed(21)
    ref = [np.random.randn(4096).astype(np.float32) for _ in range(4)]
    np.random.seed(21)
    a, b, c = (np.empty(4096, np.float32) for _ in range(3))
    first = nprand.randn_f32((4096,))  # (leaves the state "ours": a start() is allowed)
    da, db = nprand.Draw((4096,)), nprand.Draw((4096,))
    assert da.start(a) and db.start(b)
    tok = da.generation()
    other = nprand.randn_f32((4096,))  # another consumer of the stream: collects both started draws first
    assert nprand._IN_FLIGHT == 0 and nprand.generation() == tok + 1
    da.finish()  # the owners come later: nothing to wait for, nothing raised, the count stays at rest
    db.finish()
    assert nprand._IN_FLIGHT == 0
    assert np.array_equal(first, ref[0]) and np.array_equal(a, ref[1]) and np.array_equal(b, ref[2]) and np.array_equal(other, ref[3])
    # and the ordinary path still waits for its own draw
    dc = nprand.Draw((4096,))
    np.random.seed(5)
    want = np.random.randn(8192).astype(np.float32)
    np.random.seed(5)
    nprand.randn_f32((4096,))
    assert dc.start(c) and dc.generation() == nprand.generation()
    dc.finish()
    assert np.array_equal(c, want[4096:]) and nprand._IN_FLIGHT == 0


def test_host_library_exports_what_its_header_declares():
    """include/vihds_host.h: every declared entry point is exported by libvihds_host.so, and the ABI number matches."""
    import ctypes
    import re

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib_path = os.path.join(root, "vi-hds_amd", "lib", "libvihds_host.so")
    if not os.path.exists(lib_path):
        pytest.skip("libvihds_host.so not built")
    header = open(os.path.join(root, "include", "vihds_host.h")).read()
    names = sorted(set(re.findall(r"\b(vihds_[a-z0-9_]+)\s*\(", header)))
    assert names == ["vihds_host_abi_version", "vihds_host_cpu_ok", "vihds_np_randn_f32", "vihds_np_randn_f32_start",
                     "vihds_np_randn_f32_wait"]
    lib = ctypes.CDLL(lib_path)
    for n in names:
        assert hasattr(lib, n), n
    assert lib.vihds_host_abi_version() == int(re.search(r"#define VIHDS_HOST_ABI_VERSION (\d+)", header).group(1))


def test_sized_blackbox_libraries_load_on_their_own():
    """The dr_blackbox size-set libraries (lib/libvihds_bb_<L>_<HS>_<HP>_<NLAT>.so, csrc/sized/) are opened by libvihds_hip.so
    with RTLD_LOCAL and do not link against it: every symbol they use must be their own.  (A header once grew an `extern`
    that only the main library defined; the size sets then failed to load after their next rebuild.)"""
    import glob
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    libs = sorted(glob.glob(os.path.join(root, "vi-hds_amd", "lib", "libvihds_bb_*.so")))
    if not libs:
        pytest.skip("no size-set library built")
    for path in libs:
        code = ("import ctypes, os; h = ctypes.CDLL(%r, mode=os.RTLD_NOW | os.RTLD_LOCAL); "
                "assert h.vihds_bb_variant_v2" % path)
        out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
        assert out.returncode == 0, (path, out.stderr[-400:])


def test_index_batches_without_the_loader_are_the_loaders():
    """Training.run() takes an epoch's row-index batches from _shuffled_index_batches instead of iterating its
    DataLoader(shuffle=True): the same draws from torch's default generator in the same order (reference training.py:108-113
    builds exactly that loader), so the same shuffles AND the same generator state afterwards -- checked against the loader
    itself, full and ragged last batches, a batch larger than the set."""
    from torch.utils.data import DataLoader

    from vihds.training import _RowIndices, _index_batches_match_loader, _shuffled_index_batches

    for n, b in ((234, 36), (20, 8), (7, 7), (5, 10), (312, 36)):
        loader = DataLoader(dataset=_RowIndices(n), batch_size=b, shuffle=True,
                            collate_fn=lambda rows: torch.tensor(rows, dtype=torch.int64))
        torch.manual_seed(17 + n)
        want = [list(loader) for _ in range(3)]
        end = torch.get_rng_state()
        torch.manual_seed(17 + n)
        got = [_shuffled_index_batches(n, b) for _ in range(3)]
        assert torch.equal(end, torch.get_rng_state())
        for x, y in zip(want, got):
            assert len(x) == len(y) and all(torch.equal(u, v) and u.dtype == v.dtype for u, v in zip(x, y))
        torch.manual_seed(5)
        keep = torch.get_rng_state()
        assert _index_batches_match_loader(loader, n, b)
        assert torch.equal(keep, torch.get_rng_state())  # (the check consumes nothing)
