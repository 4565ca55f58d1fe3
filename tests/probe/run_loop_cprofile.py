"""Ad-hoc probe: where the HOST time of Training.run() goes (the loop of bench.py's run_loop leg: 234 rows in batches of 36,
n_iwae 200, an evaluation every 20 epochs, one hipGraph launch per epoch): cProfile of 200 epochs after the warm-up run."""
import contextlib, cProfile, io, os, pstats, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "vi-hds_amd")]
import torch
from vihds import synthetic

keys = dict(u_rng="kernel", conditioner_rng="kernel", hip_graph=True, nan_check_every=7, epoch_graph=True, lazy_cache_dump=True,
            fused_ode_training=True, fused_iwae_backward=True, fused_step_tail=True, epoch_lookahead=True)
args, settings, data, parameters, model, training = synthetic.build("dr_constant_icml", 234, 200, solver="rk4", device="cuda:0",
                                                                    seed=0, n_batch=36, learning_rate=0.001, **keys)
args.epochs, args.test_epoch, args.test_samples = 2, 1, 1000
with contextlib.redirect_stdout(io.StringIO()):
    training.run()
args.epochs, args.test_epoch = 200, 20
torch.cuda.synchronize()
pr = cProfile.Profile()
t0 = time.perf_counter()
with contextlib.redirect_stdout(io.StringIO()):
    pr.enable()
    training.run()
    pr.disable()
torch.cuda.synchronize()
el = time.perf_counter() - t0
print("200 epochs = 1400 steps in %.1f ms: %.0f steps/s; %.3f ms per epoch (GPU work of an epoch: 7 x 0.067 = 0.47 ms)" % (1e3 * el, 1400 / el, 1e3 * el / 200))
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(45)
print(s.getvalue()[:9000])
