"""Which trajectories of the extreme-parameter test differ in finiteness between the step-by-step and the time-parallel kernel."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "vi-hds_amd"), os.path.join(ROOT, "tests")]
import torch
from vihds import hip, ops
from test_hip_parity import _synthetic_theta
DEV = "cuda:0"
L = hip.lib()
solver = sys.argv[1] if len(sys.argv) > 1 else "rk4"
model = "dr_constant"
slots = hip.model_slots(model)
row_of = {n: i for i, n in enumerate(slots)}
st = torch.cuda.current_stream().cuda_stream
B, S, T = 6, 64, 86
th = _synthetic_theta(slots, B, S, 31)
g = torch.Generator().manual_seed(17)
pick = lambda vals: torch.tensor(vals)[torch.randint(0, len(vals), (B, S), generator=g)]
th["r"] = pick([-1.0, 0.0, 0.01, 0.5, 2.0, 4.0, 9.0])
th["tlag"] = pick([-10.0, 0.0, 1.0, 8.0, 24.9, 60.0])
th["K"] = pick([0.0011, 0.05, 1.0, 4.0, 7.0])
th["init_x"] = pick([1e-6, 0.002, 0.05, 0.5])
theta = torch.stack([th[n] for n in slots]).to(DEV)
cond = torch.log1p(torch.rand(B, 2, generator=g) * 1000.0).to(DEV)
times = (torch.arange(T, dtype=torch.float32) * 0.3).to(DEV)
obs = torch.rand(B, 4, T, generator=g).to(DEV)
prob = ops.OdeProblemSpec(model, solver, row_of, len(slots), C=2, kernel_variant=2).bind(B, S, T)
prob.logp_grad_broadcast = 1
prob3 = ops.OdeProblemSpec(model, solver, row_of, len(slots), C=2, kernel_variant=3).bind(B, S, T)
traj = torch.empty(T, 8, B, S, device=DEV); xpred = torch.empty(T, 4, B, S, device=DEV)
logp = torch.empty(4, B, S, device=DEV); ones = torch.ones(B, S, device=DEV)
g_ref = torch.empty_like(theta)
args = (theta.data_ptr(), cond.data_ptr(), None, times.data_ptr(), obs.data_ptr())
assert L.vihds_ode_fwd(ctypes.byref(prob), *args, None, traj.data_ptr(), xpred.data_ptr(), logp.data_ptr(), st) == 0
assert L.vihds_ode_bwd(ctypes.byref(prob), *args, None, traj.data_ptr(), None, None, ones.data_ptr(), g_ref.data_ptr(), None, None, st) == 0
logp3 = torch.full_like(logp, float("nan")); g3 = torch.full_like(theta, float("nan"))
assert L.vihds_ode_logp_grad(ctypes.byref(prob3), *args, logp3.data_ptr(), g3.data_ptr(), st) == 0
torch.cuda.synchronize()
ok = torch.isfinite(logp).all(0) & torch.isfinite(g_ref).all(0) & (logp.abs().amax(0) < 1e30)
ok3 = torch.isfinite(logp3).all(0) & torch.isfinite(g3).all(0) & (logp3.abs().amax(0) < 1e30)
print("finite: ref %.3f  scan %.3f" % (ok.float().mean(), ok3.float().mean()))
diff = (ok != ok3).nonzero()
for b, s_ in diff.tolist()[:20]:
    print("b %d s %d: r %.3g tlag %.3g K %.3g x0 %.3g | ref finite %s logp %s | scan finite %s logp %s | OD traj ref first/last %s"
          % (b, s_, th["r"][b, s_], th["tlag"][b, s_], th["K"][b, s_], th["init_x"][b, s_], bool(ok[b, s_]),
             logp[:, b, s_].tolist(), bool(ok3[b, s_]), logp3[:, b, s_].tolist(), traj[[0, 1, 2, 5, 20, -1], 0, b, s_].tolist()))
both = ok & ok3
scale = logp.abs().amax(0, keepdim=True).clamp_min(1.0)
print("max logp err on both-finite:", float(((logp3 - logp).abs() / scale)[:, both].max()))
gscale = g_ref.abs().amax(0, keepdim=True).clamp_min(1e-3)
print("max grad err on both-finite:", float(((g3 - g_ref).abs() / gscale)[:, both].max()))
