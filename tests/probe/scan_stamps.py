"""Ad-hoc probe: per-wavefront wall-clock stamps (100 MHz) at the phase boundaries of the bench's decoder launch.
Needs a library built with -DVIHDS_SCAN_STAMPS (VIHDS_HIP_LIB=.../libvihds_hip_stamps.so):
  hipcc ... -DVIHDS_SCAN_STAMPS -c ode_dr_constant_v1.hip ; link with the other objects.
Prints, per phase, the median / max duration over blocks (wave 0 and the slowest wave), and the block start / end
distribution (how many rounds of blocks the launch takes)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "vi-hds_amd"), os.path.join(ROOT, "tests")]
import numpy as np
import torch
from vihds import hip, ops, synthetic

solver = sys.argv[1] if len(sys.argv) > 1 else "rk4"
theta = (sys.argv[2] if len(sys.argv) > 2 else "theta") == "theta"
S = 200
L = hip.lib()
args, settings, data, parameters, model, training = synthetic.build(
    "dr_constant_icml", 36, S, solver=solver, device="cuda:0", seed=1, u_rng="kernel", conditioner_rng="kernel",
    hip_graph=False, nan_check_every=0, learning_rate=0.001, fused_ode_training=True, fused_decoder_step=theta)
model.train()
batch = training.train_data
training.step(batch)
rec = ops.LaunchRecorder()
ops.TIMER = rec
training.step(batch)
ops.TIMER = None
print(sorted(rec.calls))
fn = rec.calls["decoder_step" if theta else "ode_logp_grad"]
TPB = 8
nblk, nw = (36 * S + TPB - 1) // TPB, TPB // 2
buf = torch.zeros(nblk * nw * 16, dtype=torch.int64, device="cuda:0")
for _ in range(3):
    fn()
torch.cuda.synchronize()
L.vihds_debug_scan_stamps.argtypes = [ctypes.c_void_p]
assert L.vihds_debug_scan_stamps(buf.data_ptr()) == 0
fn()
torch.cuda.synchronize()
L.vihds_debug_scan_stamps(None)
st = buf.cpu().numpy().reshape(nblk, nw, 16).astype(np.float64)
t0 = st[:, :, 0].min()
us = (st - t0) / 100.0  # 100 MHz
us[st == 0] = np.nan
order = [0, 14, 15, 1, 11, 12, 2, 3, 4, 5, 6, 7, 8, 9, 13, 10]
names = {0: "start", 14: "requests, row indirection, draws", 15: "sampling arithmetic, conditioner rows, stores", 11: "x chain (Newton over the lanes), gamma", 12: "Hill terms", 1: "coefficients, barrier A",
         2: "(nothing: the old barrier slot)", 3: "parameters", 4: "level-1 maps + scan", 5: "level-1 steps, level-2 maps + scan",
         6: "log-likelihood", 7: "adjoint level 2", 8: "adjoint level 1", 9: "adjoint x", 13: "epilogue barrier", 10: "epilogue (end)"}
print("launch: first start 0, last end %.1f us; %d blocks x %d waves" % (np.nanmax(us[:, :, 10]), nblk, nw))
start = us[:, 0, 0]
end = np.nanmax(us[:, :, 10], axis=1)
print("block start times: %s" % np.round(np.percentile(start, [0, 25, 50, 56, 60, 75, 90, 100]), 1))
print("block end times:   %s" % np.round(np.percentile(end, [0, 25, 50, 75, 90, 100]), 1))
print("block duration:    median %.1f  min %.1f  max %.1f" % (np.median(end - start), (end - start).min(), (end - start).max()))
first = start < 3.0
print("blocks starting in the first 3 us: %d; their duration median %.1f; later blocks' duration median %.1f"
      % (first.sum(), np.median((end - start)[first]), np.median((end - start)[~first]) if (~first).any() else float("nan")))
prev = order[0]
for ph in order[1:]:
    d = us[:, :, ph] - us[:, :, prev]
    print("  %-44s wave0 median %5.2f  all-waves median %5.2f  max %5.2f   [first-round blocks %5.2f | later %5.2f]"
          % (names[ph], np.nanmedian(d[:, 0]), np.nanmedian(d), np.nanmax(d), np.nanmedian(d[first]),
             np.nanmedian(d[~first]) if (~first).any() else float("nan")))
    if ph == 11:
        for w in range(nw):
            print("      wave %d: median %5.2f  max %5.2f" % (w, np.nanmedian(d[:, w]), np.nanmax(d[:, w])))
    prev = ph

