"""Ad-hoc probe: per-wavefront wall-clock stamps (100 MHz) at the phase boundaries of vihds_encoder_fwd at the bench shape.
Needs the profiling build:  make -C vi-hds_amd/csrc stamps ; VIHDS_HIP_LIB=vi-hds_amd/lib/libvihds_hip_stamps.so"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "vi-hds_amd"), os.path.join(ROOT, "tests")]
import numpy as np
import torch
from vihds import hip, synthetic

L = hip.lib()
args, settings, data, parameters, model, training = synthetic.build(
    "dr_constant_icml", 36, 200, solver="rk4", device="cuda:0", seed=1, u_rng="kernel", conditioner_rng="kernel",
    hip_graph=False, nan_check_every=0, learning_rate=0.001, fused_ode_training=True, fused_decoder_step=True,
    fused_iwae_backward=True, fused_step_tail=True)
model.train()
batch = training.train_data
for _ in range(3):
    training.step(batch)
buf = torch.zeros(256 * 16 * 8, dtype=torch.int64, device="cuda:0")
torch.cuda.synchronize()
L.vihds_debug_enc_stamps.argtypes = [ctypes.c_void_p]
assert L.vihds_debug_enc_stamps(buf.data_ptr()) == 0
with torch.no_grad():
    model.encoder(batch)
torch.cuda.synchronize()
L.vihds_debug_enc_stamps(None)
st = buf.cpu().numpy().reshape(256, 16, 8).astype(np.float64)[:36]
t0 = st[st > 0].min()
us = (st - t0) / 100.0
us[st == 0] = np.nan
names = ["entry", "small loads issued", "small loads landed", "barrier 1 (inputs in LDS)", "conv done + barrier", "pool done + barrier",
         "linear + tanh done + barrier", "end"]
for ph in range(8):
    print("  phase %d %-32s min %6.2f  median %6.2f  max %6.2f us" % (ph, names[ph], np.nanmin(us[:, :, ph]),
                                                                     np.nanmedian(us[:, :, ph]), np.nanmax(us[:, :, ph])))
