"""Ad-hoc probe (not a test): thread-per-trajectory forward at the eval shape with outputs switched off -- how much of
the time is arithmetic and how much is the trajectory / x_predict stream."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "vi-hds_amd"), os.path.join(ROOT, "tests")]
import torch
from vihds import hip, ops
from test_hip_parity import _full_problem
B, S, T = 234, 1000, 86
L = hip.lib(); st = torch.cuda.current_stream().cuda_stream
slots, theta, cond, times, obs = _full_problem(B, S, T)
traj = torch.empty(T, 8, B, S, device="cuda"); xpred = torch.empty(T, 4, B, S, device="cuda"); logp = torch.empty(4, B, S, device="cuda")
g = torch.ones(B, S, device="cuda"); g_theta = torch.empty_like(theta)
def timeit(fn, n=10):
    for _ in range(2): assert fn() == 0
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for solver in ("rk4", "modeuler"):
    spec = ops.OdeProblemSpec("dr_constant", solver, {nm: i for i, nm in enumerate(slots)}, len(slots), C=2)
    prob = spec.bind(B, S, T); prob.logp_grad_broadcast = 1
    args = (theta.data_ptr(), cond.data_ptr(), None, times.data_ptr(), obs.data_ptr())
    for name, (tr, xp) in (("all outputs", (traj.data_ptr(), xpred.data_ptr())), ("traj only", (traj.data_ptr(), None)), ("no traj/xpred", (None, None))):
        print(solver, name, "%.1f us" % timeit(lambda: L.vihds_ode_fwd(ctypes.byref(prob), *args, None, tr, xp, logp.data_ptr(), st)), flush=True)
    print(solver, "bwd", "%.1f us" % timeit(lambda: L.vihds_ode_bwd(ctypes.byref(prob), *args, None, traj.data_ptr(), None, None, g.data_ptr(), g_theta.data_ptr(), None, None, st)), flush=True)
