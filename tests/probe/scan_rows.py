"""Ad-hoc probe: per-row difference of the time-parallel kernel's gradients (kernel_variant 3) against forward + adjoint."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "vi-hds_amd"), os.path.join(ROOT, "tests")]
import torch
from vihds import hip, ops
from test_hip_parity import _synthetic_theta

DEV = "cuda:0"
model, solver = "dr_constant", (sys.argv[1] if len(sys.argv) > 1 else "rk4")
L = hip.lib()
slots = hip.model_slots(model)
row_of = {n: i for i, n in enumerate(slots)}
st = torch.cuda.current_stream().cuda_stream
B, S, T = 36, 200, 86
th = _synthetic_theta(slots, B, S, 13)
theta = torch.stack([th[n] for n in slots]).to(DEV)
g = torch.Generator().manual_seed(6)
cond = torch.log1p(torch.rand(B, 2, generator=g) * 1000.0).to(DEV)
times = (torch.arange(T, dtype=torch.float32) * 0.1933 + 0.003 * torch.rand(T, generator=g)).to(DEV)
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
rc = L.vihds_ode_logp_grad(ctypes.byref(prob3), *args, logp3.data_ptr(), g3.data_ptr(), st)
torch.cuda.synchronize()
for r, n in enumerate(slots):
    a, b = g3[r].flatten(), g_ref[r].flatten()
    e = float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))
    print("%2d %-12s rel %.3e   got %s   want %s" % (r, n, e, a[:3].tolist(), b[:3].tolist()))
