"""Ad-hoc probe (not a test): thread-per-trajectory vs lane-split dr_constant kernels (and the fused training kernel)
over the number of trajectories, through the C ABI alone -- where to put VIHDS_LANE_SPLIT_MAX_N."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "vi-hds_amd"), os.path.join(ROOT, "tests")]
import torch
from vihds import hip, ops
from test_hip_parity import _full_problem

T = 86
L = hip.lib()
st = torch.cuda.current_stream().cuda_stream
def timeit(fn, n=20):
    for _ in range(3): assert fn() == 0, L.vihds_last_error()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for (B, S) in [(36, 200), (36, 400), (36, 1000), (36, 2000), (234, 1000)]:
    slots, theta, cond, times, obs = _full_problem(B, S, T)
    traj = torch.empty(T, 8, B, S, device="cuda"); xpred = torch.empty(T, 4, B, S, device="cuda")
    logp = torch.empty(4, B, S, device="cuda"); g = torch.ones(B, S, device="cuda"); g_theta = torch.empty_like(theta)
    row = []
    for variant in (1, 2):
        spec = ops.OdeProblemSpec("dr_constant", "rk4", {nm: i for i, nm in enumerate(slots)}, len(slots), C=2, kernel_variant=variant)
        prob = spec.bind(B, S, T); prob.logp_grad_broadcast = 1
        args = (theta.data_ptr(), cond.data_ptr(), None, times.data_ptr(), obs.data_ptr())
        f = timeit(lambda: L.vihds_ode_fwd(ctypes.byref(prob), *args, None, traj.data_ptr(), xpred.data_ptr(), logp.data_ptr(), st))
        b = timeit(lambda: L.vihds_ode_bwd(ctypes.byref(prob), *args, None, traj.data_ptr(), None, None, g.data_ptr(), g_theta.data_ptr(), None, None, st))
        row.append("variant %d: fwd %7.1f bwd %7.1f" % (variant, f, b))
        if variant == 2:
            if L.vihds_ode_logp_grad(ctypes.byref(prob), *args, logp.data_ptr(), g_theta.data_ptr(), st) == 0:
                fu = timeit(lambda: L.vihds_ode_logp_grad(ctypes.byref(prob), *args, logp.data_ptr(), g_theta.data_ptr(), st))
                row.append("fused %7.1f" % fu)
            else:
                row.append("fused: declined")
    print("n=%6d (B=%d,S=%d) rk4: %s" % (B * S, B, S, " | ".join(row)), flush=True)
