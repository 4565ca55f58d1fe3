"""Per-step losses of this package's run at seed 2 (reference keys, modeuler, n_iwae 200, eager), one float per step."""
import contextlib, io, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "vi-hds_amd"), os.path.join(ROOT, "tests")]
import numpy as np, torch
from vihds import synthetic
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 2
extra = {}
for a in sys.argv[2:]:
    k, v = a.split("=")
    extra[k] = {"True": True, "False": False}.get(v, v)
path = os.path.join(ROOT, "tests", "golden", "trace_dr_constant_icml_s200_modeuler.npz")
args, settings, data, parameters, model, training = synthetic.build_recorded_plate(path, 200, solver="modeuler", device="cuda:0", seed=seed, hip_graph=False, **extra)
args.epochs, args.test_epoch, args.test_samples = 15, 15, 200
losses = []
orig = training.step
def rec(batch, zero_grad=True):
    out = orig(batch, zero_grad)
    losses.append(float(out))
    return out
training.step = rec
with contextlib.redirect_stdout(io.StringIO()):
    out = training.run()
np.set_printoptions(linewidth=220, precision=5)
print("OURS", np.array(losses))
np.save(os.path.join(ROOT, "gpurun_out", "ours_seed%d_losses.npy" % seed), np.array(losses))
