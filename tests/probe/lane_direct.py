"""Ad-hoc probe (not a test): lane-kernel time through the C ABI alone (no autograd / allocator time on the host),
forward with some outputs switched off and the adjoint."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "vi-hds_amd"), os.path.join(ROOT, "tests")]
import torch
from vihds import hip, ops
from test_hip_parity import _full_problem

B, S = 36, 200
shapes = [(86,), (170,)] if len(sys.argv) < 2 else [(int(sys.argv[1]),)]
for (T,) in shapes:
    slots, theta, cond, times, obs = _full_problem(B, S, T)
    for solver in ("rk4", "midpoint", "modeuler"):
        spec = ops.OdeProblemSpec("dr_constant", solver, {nm: i for i, nm in enumerate(slots)}, len(slots), C=2)
        prob = spec.bind(B, S, T)
        prob.logp_grad_broadcast = 0
        traj = torch.empty(T, 8, B, S, device="cuda"); xpred = torch.empty(T, 4, B, S, device="cuda")
        logp = torch.empty(4, B, S, device="cuda"); g_logp = torch.ones(4, B, S, device="cuda")
        g_theta = torch.empty_like(theta)
        L = hip.lib()
        st = torch.cuda.current_stream().cuda_stream
        def fwd(tr, xp, lp):
            return L.vihds_ode_fwd(ctypes.byref(prob), theta.data_ptr(), cond.data_ptr(), None, times.data_ptr(),
                                   obs.data_ptr(), None, tr, xp, lp, st)
        def bwd():
            return L.vihds_ode_bwd(ctypes.byref(prob), theta.data_ptr(), cond.data_ptr(), None, times.data_ptr(),
                                   obs.data_ptr(), None, traj.data_ptr(), None, None, g_logp.data_ptr(),
                                   g_theta.data_ptr(), None, None, st)
        cases = [("fwd all outputs", lambda: fwd(traj.data_ptr(), xpred.data_ptr(), logp.data_ptr())),
                 ("fwd no traj/xpred stores", lambda: fwd(None, None, logp.data_ptr())),
                 ("fwd no logp", lambda: fwd(traj.data_ptr(), xpred.data_ptr(), None)),
                 ("bwd (g_logp only)", bwd)]
        for name, fn in cases:
            for _ in range(5): assert fn() == 0, L.vihds_last_error()
            torch.cuda.synchronize()
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(50): fn()
            e1.record(); torch.cuda.synchronize()
            print("T=%3d %-9s %-26s %.1f us" % (T, solver, name, e0.elapsed_time(e1) / 50 * 1e3))
        glogp2 = torch.empty(4, B, S, device="cuda"); g_unit = torch.empty_like(theta)
        def fused():
            return L.vihds_ode_logp_grad(ctypes.byref(prob), theta.data_ptr(), cond.data_ptr(), None, times.data_ptr(),
                                         obs.data_ptr(), glogp2.data_ptr(), g_unit.data_ptr(), st)
        for _ in range(5): assert fused() == 0, L.vihds_last_error()
        torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50): fused()
        e1.record(); torch.cuda.synchronize()
        print("T=%3d %-9s %-26s %.1f us" % (T, solver, "fused logp + unit adjoint", e0.elapsed_time(e1) / 50 * 1e3))
