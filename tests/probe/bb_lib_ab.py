"""A/B of builds of the library on the dr_blackbox cooperating-wavefront kernels at BASELINE config 4's shape (forward and
adjoint launch times; results are not looked at, so experimental builds that compute garbage can be timed):
   VIHDS_HIP_LIB=<path> python tests/probe/bb_lib_ab.py [S]"""
import os, sys, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/vi-hds_amd"); sys.path.insert(0, "/root/repo/tests")
from test_hip_parity import _blackbox_problem
from vihds import ops
B, S, T = 36, int(sys.argv[1]) if len(sys.argv) > 1 else 200, 86
spec, theta, wts, cond, dev, times, obs = _blackbox_problem(B, S, T)
th = theta.clone().requires_grad_(True)
w = wts.clone().requires_grad_(True)
g = torch.full((4, B, S), -1.0 / (B * S), device="cuda")
def run():
    traj, xpred, logp = ops.OdeSolveObserve.apply(spec, th, cond, times, obs, dev, w)
    logp.backward(g)
for _ in range(5):
    run()
torch.cuda.synchronize()
rec = ops.KernelTimer()
ops.TIMER = rec
for _ in range(40):
    run()
ops.TIMER = None
print(os.path.basename(os.environ.get("VIHDS_HIP_LIB", "default")), "S=%d" % S,
      {k: (round(x["mean_us"], 1), round(x["min_us"], 1)) for k, x in rec.summary().items()})
