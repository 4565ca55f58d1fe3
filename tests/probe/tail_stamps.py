"""Ad-hoc probe: per-wavefront wall-clock stamps (100 MHz) at the phase boundaries of vihds_step_tail's two launches at the
bench shape.  Needs the profiling build:  make -C vi-hds_amd/csrc stamps ; VIHDS_HIP_LIB=vi-hds_amd/lib/libvihds_hip_stamps.so"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "vi-hds_amd"), os.path.join(ROOT, "tests")]
import numpy as np
import torch
from vihds import hip, ops, synthetic

L = hip.lib()
args, settings, data, parameters, model, training = synthetic.build(
    "dr_constant_icml", 36, 200, solver="rk4", device="cuda:0", seed=1, u_rng="kernel", conditioner_rng="kernel",
    hip_graph=False, nan_check_every=0, learning_rate=0.001, fused_ode_training=True, fused_decoder_step=True,
    fused_iwae_backward=True, fused_step_tail=True)
model.train()
batch = training.train_data
training.step(batch)
rec = ops.LaunchRecorder()
ops.TIMER = rec
training.step(batch)
ops.TIMER = None
fn = rec.calls["step_tail"]
buf = torch.zeros(2 * 1024 * 16 * 8, dtype=torch.int64, device="cuda:0")
for _ in range(3):
    fn()
torch.cuda.synchronize()
L.vihds_debug_tail_stamps.argtypes = [ctypes.c_void_p]
assert L.vihds_debug_tail_stamps(buf.data_ptr()) == 0
fn()
torch.cuda.synchronize()
L.vihds_debug_tail_stamps(None)
st = buf.cpu().numpy().reshape(2, 1024, 16, 8).astype(np.float64)
t0 = st[st > 0].min()
us = (st - t0) / 100.0
us[st == 0] = np.nan
rows = us[0, :36]
print("rows kernel: 36 blocks x 16 waves; phases: 0 entry, 1 loads issued, 2 first barrier passed (loads landed), 3 weights done, "
      "4 theta done (own wave), 5 barrier after theta, 6 end")
for ph in range(7):
    print("  phase %d: min %6.2f  median %6.2f  max %6.2f us" % (ph, np.nanmin(rows[:, :, ph]), np.nanmedian(rows[:, :, ph]),
                                                                np.nanmax(rows[:, :, ph])))
print("  per-wave theta time (4 - 3), block 5:", np.round(rows[5, :, 4] - rows[5, :, 3], 2).tolist())
print("  block 0 vs others, phase 2:", np.round(rows[0, 0, 2], 2), np.round(np.nanmedian(rows[1:, 0, 2]), 2))
upd = us[1]
nb = int(np.isfinite(upd[:, 0, 0]).sum())
print("update kernel: %d blocks stamped; phases: 0 entry, 1 gate passed, 2 sum formed, 3 adam written" % nb)
for ph in range(4):
    v = upd[:, :, ph]
    print("  phase %d: min %6.2f  median %6.2f  max %6.2f us" % (ph, np.nanmin(v), np.nanmedian(v), np.nanmax(v)))
lin = upd[:141]
conv = upd[141:181]
for name, v in (("lin_w blocks", lin), ("conv_w blocks", conv)):
    print("  %s: entry median %.2f, gate %.2f, sum %.2f, end %.2f (max end %.2f)" % (
        name, np.nanmedian(v[:, :, 0]), np.nanmedian(v[:, :, 1]), np.nanmedian(v[:, :, 2]), np.nanmedian(v[:, :, 3]),
        np.nanmax(v[:, :, 3])))
