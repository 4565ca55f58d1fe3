"""A/B of two builds of the library on the relay lane kernels at BASELINE config 5's shape (forward + adjoint launch times):
   VIHDS_HIP_LIB=<path> python tests/probe/relay_lib_ab.py"""
import os, sys, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/vi-hds_amd"); sys.path.insert(0, "/root/repo/tests")
from test_hip_parity import _relay_problem
from vihds import ops
B, S, TT = 36, 200, 99
for model in ("relay_constant", "relay_constant_precisions"):
    slots, theta, cond, times, obs, wts = _relay_problem(model, B, S, TT, 3, dt=0.17)
    row_of = {n: i for i, n in enumerate(slots)}
    spec = ops.OdeProblemSpec(model, "midpoint", row_of, len(slots), C=cond.shape[1], kernel_variant=0)
    th = theta.clone().requires_grad_(True)
    w = wts.clone().requires_grad_(True) if wts is not None else None
    g = torch.full((4, B, S), -1.0 / (B * S), device="cuda")
    def run():
        traj, xpred, logp = ops.OdeSolveObserve.apply(spec, th, cond, times, obs, None, w)
        logp.backward(g)
    for _ in range(5):
        run()
    torch.cuda.synchronize()
    rec = ops.KernelTimer()
    ops.TIMER = rec
    for _ in range(40):
        run()
    ops.TIMER = None
    print(os.environ.get("VIHDS_HIP_LIB", "default"), model, {k: (round(x["mean_us"], 1), round(x["min_us"], 1)) for k, x in rec.summary().items()})
