"""Ad-hoc probe: where the wall time of Training.run() goes (training epochs vs evaluations vs scheduler)."""
import contextlib, io, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "vi-hds_amd")]
import torch
from vihds import synthetic

args, settings, data, parameters, model, training = synthetic.build(
    "dr_constant_icml", 234, 200, solver="rk4", device="cuda:0", seed=1, n_batch=36, u_rng="kernel", conditioner_rng="kernel",
    hip_graph=True, nan_check_every=7, learning_rate=0.001, fused_ode_training=True, fused_iwae_backward=True, fused_step_tail=True, lazy_cache_dump=True)
args.epochs, args.test_epoch, args.test_samples = 2, 1, 1000
with contextlib.redirect_stdout(io.StringIO()):
    training.run()
orig_eval = training._evaluate_elbo_and_plot
acc = {"eval": 0.0, "n_eval": 0, "sched": 0.0, "loader": 0.0, "epoch_rows": 0.0}
def timed_eval(*a, **k):
    torch.cuda.synchronize(); t = time.perf_counter()
    out = orig_eval(*a, **k)
    torch.cuda.synchronize(); acc["eval"] += time.perf_counter() - t; acc["n_eval"] += 1
    return out
training._evaluate_elbo_and_plot = timed_eval
orig_sched = training.scheduler.step
def timed_sched(*a, **k):
    t = time.perf_counter(); out = orig_sched(*a, **k); acc["sched"] += time.perf_counter() - t; return out
training.scheduler.step = timed_sched
orig_rows = training.epoch_rows
def timed_rows(b):
    t = time.perf_counter(); out = orig_rows(b); acc["epoch_rows"] += time.perf_counter() - t; return out
training.epoch_rows = timed_rows
args.epochs, args.test_epoch = 100, 20
torch.cuda.synchronize(); t0 = time.perf_counter()
with contextlib.redirect_stdout(io.StringIO()):
    training.run()
torch.cuda.synchronize(); el = time.perf_counter() - t0
print("run(): %.1f ms for 700 steps = %.0f steps/s" % (el * 1e3, 700 / el))
print("evaluations: %d, %.1f ms in total (%.2f ms each)" % (acc["n_eval"], acc["eval"] * 1e3, acc["eval"] * 1e3 / max(1, acc["n_eval"])))
print("scheduler.step: %.2f ms in total; epoch_rows host time: %.2f ms in total" % (acc["sched"] * 1e3, acc["epoch_rows"] * 1e3))
print("training part: %.1f ms = %.0f steps/s" % ((el - acc["eval"]) * 1e3, 700 / (el - acc["eval"])))
