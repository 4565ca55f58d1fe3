"""Ad-hoc probe: shader-cycle stamps of two consecutive steps of the state wavefront of the split dr_blackbox FORWARD (config 4).
Needs a library whose ode_dr_blackbox_fwd.o was built with -DVIHDS_BB_STAMPS (VIHDS_HIP_LIB=...)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "vi-hds_amd"), os.path.join(ROOT, "tests")]
import numpy as np
import torch
from test_hip_parity import _blackbox_problem
from vihds import hip, ops

L = hip.lib()
spec, theta, wts, cond, dev, times, obs = _blackbox_problem(36, 200, 86)
for _ in range(3):
    ops.OdeSolveObserve.apply(spec, theta, cond, times, obs, dev, wts)
torch.cuda.synchronize()
buf = torch.zeros(8192 + 8 * 64, dtype=torch.int64, device="cuda:0")
L.vihds_debug_bb_fwd_stamps.argtypes = [ctypes.c_void_p]
assert L.vihds_debug_bb_fwd_stamps(buf.data_ptr()) == 0
ops.OdeSolveObserve.apply(spec, theta, cond, times, obs, dev, wts)
torch.cuda.synchronize()
L.vihds_debug_bb_fwd_stamps(None)
st = buf.cpu().numpy()[8192:].reshape(8, 64).astype(np.float64)
names = ["step", "eval1 in", "published", "net done", "rates done", "eval2 in", "published", "net done", "rates done", "step end"]
for blk in (0, 1, 5):
    row = st[blk]
    row = row[row > 0]
    print("block %d: %d stamps; deltas (shader cycles):" % (blk, len(row)))
    d = np.diff(row)
    print("   " + " ".join("%6.0f" % v for v in d))
