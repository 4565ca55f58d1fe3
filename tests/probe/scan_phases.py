"""Ad-hoc probe: cumulative time of the scan kernel's phases (kernel_variant = 3 | phase << 8 returns after a phase)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "vi-hds_amd"), os.path.join(ROOT, "tests")]
import torch
from vihds import hip, ops
from test_hip_parity import _synthetic_theta

DEV = "cuda"
L = hip.lib()
B, S, T = (int(v) for v in (sys.argv[1:4] if len(sys.argv) > 3 else (36, 200, 86)))
names = {1: "parameters, hill, sigmoids (waves 1-3)", 2: "x chains (wave 0) + barrier", 3: "gamma", 4: "level-1 maps + scan", 5: "level-1 steps, promoters, level-2 maps + scan",
         6: "log-likelihood", 7: "adjoint level 2 + promoters", 8: "adjoint level 1", 9: "adjoint x", 0: "epilogue (full kernel)"}
for solver in ("rk4", "midpoint", "euler"):
    slots = hip.model_slots("dr_constant")
    th = _synthetic_theta(slots, B, S, 13)
    theta = torch.stack([th[n] for n in slots]).to(DEV)
    g = torch.Generator().manual_seed(6)
    cond = torch.log1p(torch.rand(B, 2, generator=g) * 1000.0).to(DEV)
    times = (torch.arange(T, dtype=torch.float32) * 0.1933).to(DEV)
    obs = torch.rand(B, 4, T, generator=g).to(DEV)
    row_of = {n: i for i, n in enumerate(slots)}
    logp = torch.empty(4, B, S, device=DEV); g3 = torch.empty_like(theta)
    st = torch.cuda.current_stream().cuda_stream
    args = (theta.data_ptr(), cond.data_ptr(), None, times.data_ptr(), obs.data_ptr())
    prev = 0.0
    print("== %s B=%d S=%d T=%d" % (solver, B, S, T))
    for ph in list(range(1, 10)) + [0]:
        prob = ops.OdeProblemSpec("dr_constant", solver, row_of, len(slots), C=2, kernel_variant=3 | (ph << 8)).bind(B, S, T)
        fn = lambda: L.vihds_ode_logp_grad(ctypes.byref(prob), *args, logp.data_ptr(), g3.data_ptr(), st)
        for _ in range(5): assert fn() == 0, L.vihds_last_error()
        torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(100): fn()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 100 * 1e3
        print("  after %-48s %6.1f us  (+%.1f)" % (names[ph], us, us - prev))
        prev = us
