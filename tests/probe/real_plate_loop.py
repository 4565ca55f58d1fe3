"""Training.run() on the reference's processed ICML plate at the spec's own learning rate / schedule (the bench's `real_plate`
leg): the evaluation lines of the whole run, to see where (if anywhere) the objective leaves the finite range.
usage: python tests/probe/real_plate_loop.py [epochs] [lr or 'spec'] [test_epoch]"""
import contextlib, io, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "vi-hds_amd")]
import torch
from vihds import synthetic

epochs = int(sys.argv[1]) if len(sys.argv) > 1 else 300
lr = sys.argv[2] if len(sys.argv) > 2 else "spec"
test_epoch = int(sys.argv[3]) if len(sys.argv) > 3 else 20
keys = dict(u_rng="kernel", conditioner_rng="kernel", hip_graph=True, nan_check_every=7, epoch_graph=True, lazy_cache_dump=True,
            fused_ode_training=True, fused_iwae_backward=True, fused_step_tail=True)
if lr != "spec":
    keys["learning_rate"] = float(lr)
args, settings, data, parameters, model, training = synthetic.build_recorded_plate(
    os.path.join(ROOT, "tests", "golden", "trace_dr_constant_icml_modeuler.npz"), 200, solver="rk4", device="cuda:0", seed=0, **keys)
args.epochs, args.test_epoch, args.test_samples = epochs, test_epoch, 1000
buf = io.StringIO()
with contextlib.redirect_stdout(buf):
    out = training.run()
for ln in buf.getvalue().splitlines():
    if "iwae-elbo" in ln or "Cannot" in ln:
        print(ln[:150])
print("lr", settings.params.learning_rate, "boundaries", settings.params.learning_boundaries, "final", None if out is None else float(out.elbo))
