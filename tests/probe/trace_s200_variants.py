"""Where does the real-plate run leave the reference's trajectory?  Training.run() on the plate of
tests/golden/trace_dr_constant_icml_s200_modeuler.npz (reference: 105 steps, n_iwae 200, lr 0.01, modeuler -> loss -522, finite)
under several key sets; prints the loss every 10 steps beside the reference's."""
import contextlib, io, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "vi-hds_amd"), os.path.join(ROOT, "tests")]
import numpy as np, torch
from vihds import synthetic

path = os.path.join(ROOT, "tests", "golden", "trace_dr_constant_icml_s200_modeuler.npz")
ref = np.load(path)["step_losses"]
variants = {
    "reference keys (numpy u, cpu conditioner, modeuler)": dict(solver="modeuler"),
    "modeuler + kernel rng": dict(solver="modeuler", u_rng="kernel", conditioner_rng="kernel"),
    "rk4 + reference rng": dict(solver="rk4"),
    "rk4 + kernel rng (bench keys)": dict(solver="rk4", u_rng="kernel", conditioner_rng="kernel"),
    "rk4 + kernel rng, no step tail": dict(solver="rk4", u_rng="kernel", conditioner_rng="kernel", fused_step_tail=False),
    "rk4 + kernel rng, no fused decoder": dict(solver="rk4", u_rng="kernel", conditioner_rng="kernel", fused_ode_training=False),
    "rk4 + kernel rng, eager": dict(solver="rk4", u_rng="kernel", conditioner_rng="kernel", hip_graph=False),
}
only = sys.argv[1] if len(sys.argv) > 1 else None
for name, kw in variants.items():
    if only and only not in name:
        continue
    solver = kw.pop("solver")
    args, settings, data, parameters, model, training = synthetic.build_recorded_plate(path, 200, solver=solver, device="cuda:0", seed=0, **kw)
    args.epochs, args.test_epoch, args.test_samples = 15, 15, 200
    losses = []
    for attr in ("step_rows",):
        orig = getattr(training, attr)
        def rec(rows, next_rows=None, ahead=1, _o=orig):
            out = _o(rows, next_rows, ahead); losses.append(out); return out
        setattr(training, attr, rec)
    orig_e = training.epoch_rows
    def rec_e(batches, _o=orig_e):
        out = _o(batches); losses.extend(out); return out
    training.epoch_rows = rec_e
    orig_s = training.step
    if not training.use_graph:
        def rec_s(batch, zero_grad=True, _o=orig_s):
            out = _o(batch, zero_grad); losses.append(out); return out
        training.step = rec_s
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        out = training.run()
    ls = np.array([float(v) for v in losses])
    print("%-48s n=%d" % (name, len(ls)), " ".join("%9.4g" % v for v in ls[::10]), "| last %.4g  valid %s" % (ls[-1], None if out is None else float(out.elbo)))
print("%-48s n=%d" % ("REFERENCE (recorded)", len(ref)), " ".join("%9.4g" % v for v in ref[::10]), "| last %.4g" % ref[-1])
