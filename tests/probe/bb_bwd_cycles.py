"""Ad-hoc probe: shader-cycle stamps of one step of the four wavefronts of the split dr_blackbox ADJOINT (config 4).
Needs a library whose ode_dr_blackbox.o was built with -DVIHDS_BB_STAMPS -DVIHDS_BB_CYCLES_ONLY -DVIHDS_BB_CYC_COARSE."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "vi-hds_amd"), os.path.join(ROOT, "tests")]
import numpy as np
import torch
from test_hip_parity import _blackbox_problem
from vihds import hip, ops

L = hip.lib()
spec, theta, wts, cond, dev, times, obs = _blackbox_problem(36, 200, 86)
th = theta.clone().requires_grad_(True)
w = wts.clone().requires_grad_(True)
g = torch.full((4, 36, 200), -1.0 / 7200, device="cuda")
def run():
    traj, xpred, logp = ops.OdeSolveObserve.apply(spec, th, cond, times, obs, dev, w)
    logp.backward(g)
for _ in range(3):
    run()
torch.cuda.synchronize()
buf = torch.zeros(8192 + 8 * 64, dtype=torch.int64, device="cuda:0")
L.vihds_debug_bb_stamps.argtypes = [ctypes.c_void_p]
assert L.vihds_debug_bb_stamps(buf.data_ptr()) == 0
run()
torch.cuda.synchronize()
L.vihds_debug_bb_stamps(None)
st = buf.cpu().numpy()[8192:].reshape(8, 64).astype(np.float64)
print("A: step start | at barrier 1 | at barrier 2 | step_vjp done | next step start;  B: step start | at barrier 1 | at barrier 2 | done;  H: at barrier 1 | at barrier 2")
for blk in (0, 1, 5):
    t0 = st[blk][st[blk] > 0].min()
    for name, lo, hi in (("A", 16, 32), ("B", 32, 48), ("H1", 48, 56), ("H2", 56, 64)):
        row = st[blk][lo:hi]
        row = row[row > 0]
        print("block %d %-2s: %s" % (blk, name, " ".join("%6.0f" % (v - t0) for v in row)))
