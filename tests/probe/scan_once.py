"""Ad-hoc probe for counter collection: launches the scan kernel (or variant 2) a few times at B=36, S=200, T=86, rk4."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "vi-hds_amd"), os.path.join(ROOT, "tests")]
import torch
from vihds import hip, ops
from test_hip_parity import _synthetic_theta
DEV = "cuda"; L = hip.lib()
variant = int(sys.argv[1]) if len(sys.argv) > 1 else 3
solver = sys.argv[2] if len(sys.argv) > 2 else "rk4"
B, S, T = 36, 200, 86
slots = hip.model_slots("dr_constant")
th = _synthetic_theta(slots, B, S, 13)
theta = torch.stack([th[n] for n in slots]).to(DEV)
g = torch.Generator().manual_seed(6)
cond = torch.log1p(torch.rand(B, 2, generator=g) * 1000.0).to(DEV)
times = (torch.arange(T, dtype=torch.float32) * 0.1933).to(DEV)
obs = torch.rand(B, 4, T, generator=g).to(DEV)
row_of = {n: i for i, n in enumerate(slots)}
prob = ops.OdeProblemSpec("dr_constant", solver, row_of, len(slots), C=2, kernel_variant=variant).bind(B, S, T)
logp = torch.empty(4, B, S, device=DEV); g3 = torch.empty_like(theta)
st = torch.cuda.current_stream().cuda_stream
for _ in range(5):
    assert L.vihds_ode_logp_grad(ctypes.byref(prob), theta.data_ptr(), cond.data_ptr(), None, times.data_ptr(), obs.data_ptr(),
                                 logp.data_ptr(), g3.data_ptr(), st) == 0, L.vihds_last_error()
torch.cuda.synchronize()
