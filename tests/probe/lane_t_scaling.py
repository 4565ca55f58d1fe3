"""Ad-hoc probe (not a test): lane kernel time vs number of time points (separates launch + prologue from the loop)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "vi-hds_amd"), os.path.join(ROOT, "tests")]
import torch
from vihds import hip, ops
from test_hip_parity import _full_problem

B, S = 36, 200
for solver in ("rk4", "modeuler", "euler"):
    for T in (2, 12, 44, 86, 170):
        slots, theta, cond, times, obs = _full_problem(B, S, T)
        spec = ops.OdeProblemSpec("dr_constant", solver, {nm: i for i, nm in enumerate(slots)}, len(slots), C=2)
        th = theta.clone().requires_grad_(True)
        out = {}
        def fwd(): out["o"] = ops.OdeSolveObserve.apply(spec, th, cond, times, obs, None, None)
        fwd(); gl = torch.ones_like(out["o"][2])
        def bwd():
            th.grad = None
            out["o"][2].backward(gl, retain_graph=True)
        res = []
        for fn in (fwd, bwd):
            for _ in range(5): fn()
            torch.cuda.synchronize()
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(50): fn()
            e1.record(); torch.cuda.synchronize()
            res.append(e0.elapsed_time(e1) / 50 * 1e3)
        print("%-9s T=%3d fwd %.1f us bwd %.1f us" % (solver, T, res[0], res[1]))
