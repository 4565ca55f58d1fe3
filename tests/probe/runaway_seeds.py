"""Is the real-plate run-away at lr 0.01 a matter of the draws (seed luck) or of a key?  15 epochs of Training.run() on the recorded
plate per (solver, rng, seed); prints the final validation ELBO (the reference, modeuler, numpy stream, seed 0: 579.2)."""
import contextlib, io, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "vi-hds_amd"), os.path.join(ROOT, "tests")]
import numpy as np, torch
from vihds import synthetic

path = os.path.join(ROOT, "tests", "golden", "trace_dr_constant_icml_s200_modeuler.npz")
epochs = int(sys.argv[1]) if len(sys.argv) > 1 else 15
for solver in ("modeuler", "rk4", "midpoint"):
    for rng in ("numpy", "kernel", "device"):
        vals = []
        for seed in range(4):
            kw = {} if rng == "numpy" else dict(u_rng=rng, conditioner_rng=rng)
            args, settings, data, parameters, model, training = synthetic.build_recorded_plate(
                path, 200, solver=solver, device="cuda:0", seed=0, **kw)
            # (the split and the initial weights are seed 0's -- the reference's; only the random streams of the run differ)
            np.random.seed(1000 + seed); torch.manual_seed(1000 + seed)
            model._rng_state = None
            args.epochs, args.test_epoch, args.test_samples = epochs, epochs, 200
            with contextlib.redirect_stdout(io.StringIO()):
                out = training.run()
            vals.append(None if out is None else float(out.elbo))
            del training, model
        print("%-9s %-7s" % (solver, rng), " ".join("%12.5g" % v if v is not None else "        None" for v in vals), flush=True)
