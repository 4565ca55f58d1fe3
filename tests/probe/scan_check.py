"""Ad-hoc probe: the time-parallel training kernel (kernel_variant 3) against the forward + adjoint pair through the
C ABI -- per-signal log-likelihoods and per-parameter unit-weight gradients -- and its launch time."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "vi-hds_amd"), os.path.join(ROOT, "tests")]
import torch
from vihds import hip, ops
from test_hip_parity import _synthetic_theta

DEV = "cuda"
L = hip.lib()
shapes = [(36, 200, 86), (7, 5, 31), (3, 9, 100), (5, 4, 2), (4, 8, 129)]
timing = len(sys.argv) > 1
for model in ("dr_constant", "dr_constant_v2"):
    for solver in ("rk4", "midpoint", "modeuler", "modeulerwhile", "euler"):
        for (B, S, T) in shapes:
            slots = hip.model_slots(model)
            th = _synthetic_theta(slots, B, S, 13)
            theta = torch.stack([th[n] for n in slots]).to(DEV)
            g = torch.Generator().manual_seed(6)
            cond = torch.log1p(torch.rand(B, 2, generator=g) * 1000.0).to(DEV)
            times = (torch.arange(T, dtype=torch.float32) * 0.1933 + 0.003 * torch.rand(T, generator=g)).to(DEV)
            obs = torch.rand(B, 4, T, generator=g).to(DEV)
            row_of = {n: i for i, n in enumerate(slots)}
            prob = ops.OdeProblemSpec(model, solver, row_of, len(slots), C=2, kernel_variant=2).bind(B, S, T)
            prob.logp_grad_broadcast = 1
            prob3 = ops.OdeProblemSpec(model, solver, row_of, len(slots), C=2, kernel_variant=3).bind(B, S, T)
            st = torch.cuda.current_stream().cuda_stream
            traj = torch.empty(T, 8, B, S, device=DEV); xpred = torch.empty(T, 4, B, S, device=DEV)
            logp = torch.empty(4, B, S, device=DEV); ones = torch.ones(B, S, device=DEV)
            g_ref = torch.empty_like(theta)
            args = (theta.data_ptr(), cond.data_ptr(), None, times.data_ptr(), obs.data_ptr())
            assert L.vihds_ode_fwd(ctypes.byref(prob), *args, None, traj.data_ptr(), xpred.data_ptr(), logp.data_ptr(), st) == 0
            assert L.vihds_ode_bwd(ctypes.byref(prob), *args, None, traj.data_ptr(), None, None, ones.data_ptr(),
                                   g_ref.data_ptr(), None, None, st) == 0
            logp3 = torch.full_like(logp, float("nan")); g3 = torch.full_like(theta, float("nan"))
            rc = L.vihds_ode_logp_grad(ctypes.byref(prob3), *args, logp3.data_ptr(), g3.data_ptr(), st)
            if rc != 0:
                print("%-15s %-13s B=%d S=%d T=%d: declined (%s)" % (model, solver, B, S, T, L.vihds_last_error().decode()))
                continue
            torch.cuda.synchronize()
            e_lp = max(float((logp3[j] - logp[j]).abs().max() / logp[j].abs().max()) for j in range(4))
            worst, wn = 0.0, ""
            for r, n in enumerate(slots):
                scale = float(g_ref[r].abs().max())
                if scale > 0:
                    e = float((g3[r] - g_ref[r]).abs().max() / scale)
                    if not e <= worst:
                        worst, wn = e, n
            flag = "" if (e_lp < 1e-5 and worst < 2e-4) else "   <<<<<< MISMATCH"
            print("%-15s %-13s B=%2d S=%3d T=%3d  logp %.1e  grad %.1e (%s)%s" % (model, solver, B, S, T, e_lp, worst, wn, flag))
            if timing and (B, S, T) == (36, 200, 86):
                for name, p in (("lane kernel", prob), ("scan kernel", prob3)):
                    fn = lambda: L.vihds_ode_logp_grad(ctypes.byref(p), *args, logp3.data_ptr(), g3.data_ptr(), st)
                    for _ in range(5): fn()
                    torch.cuda.synchronize()
                    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(50): fn()
                    e1.record(); torch.cuda.synchronize()
                    print("      %-12s %.1f us" % (name, e0.elapsed_time(e1) / 50 * 1e3))
