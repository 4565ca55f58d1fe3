"""Ad-hoc probe: cProfile of Training.run() (100 epochs of 7 steps, epoch graphs) -- which host calls the epoch loop spends
its time in.  usage: python tests/probe/run_loop_cprofile.py"""
import contextlib, cProfile, io, os, pstats, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "vi-hds_amd")]
from vihds import synthetic

args, settings, data, parameters, model, training = synthetic.build(
    "dr_constant_icml", 234, 200, solver="rk4", device="cuda:0", seed=1, n_batch=36, u_rng="kernel", conditioner_rng="kernel",
    hip_graph=True, nan_check_every=7, learning_rate=0.001, fused_ode_training=True, fused_iwae_backward=True,
    fused_step_tail=True, lazy_cache_dump=True)
args.epochs, args.test_epoch, args.test_samples = 2, 1, 1000
with contextlib.redirect_stdout(io.StringIO()):
    training.run()
args.epochs, args.test_epoch = 100, 1000
pr = cProfile.Profile()
with contextlib.redirect_stdout(io.StringIO()):
    pr.enable()
    training.run()
    pr.disable()
st = pstats.Stats(pr, stream=sys.stdout)
st.sort_stats("cumulative").print_stats(28)
