"""Ad-hoc probe: wall-clock stamps (100 MHz) of ONE time step of the split dr_blackbox adjoint (config 4), per wavefront
role (A states, B precisions, H1 / H2 Gram helpers).  Needs the stamps library (make -C vi-hds_amd/csrc stamps;
VIHDS_HIP_LIB=.../libvihds_hip_stamps.so)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "vi-hds_amd"), os.path.join(ROOT, "tests")]
import numpy as np
import torch
from vihds import hip, synthetic

solver = sys.argv[1] if len(sys.argv) > 1 else "midpoint"
L = hip.lib()
args, settings, data, parameters, model, training = synthetic.build(
    "dr_blackbox_icml", 36, 200, solver=solver, device="cuda:0", seed=1, u_rng="kernel", conditioner_rng="kernel",
    hip_graph=False, nan_check_every=0, learning_rate=0.001)
model.train()
batch = training.train_data
for _ in range(3):
    training.step(batch)
torch.cuda.synchronize()
buf = torch.zeros(64 * 4 * 32, dtype=torch.int64, device="cuda:0")
L.vihds_debug_bb_stamps.argtypes = [ctypes.c_void_p]
assert L.vihds_debug_bb_stamps(buf.data_ptr()) == 0
training.step(batch)
torch.cuda.synchronize()
L.vihds_debug_bb_stamps(None)
st = buf.cpu().numpy().reshape(64, 4, 32).astype(np.float64)
for blk in (0, 1, 17, 40):
    t0 = st[blk][st[blk] > 0].min()
    print("block %d" % blk)
    for role, name in enumerate(("A states", "B precisions", "H1", "H2")):
        row = st[blk, role]
        row = row[row > 0]
        us = (row - t0) / 100.0
        print("  %-13s %s" % (name, " ".join("%6.2f" % v for v in us)))
        print("  %-13s %s" % ("  (deltas)", " ".join("%6.2f" % v for v in np.diff(us))))
