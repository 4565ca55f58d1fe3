"""Ad-hoc probe (not a test): where the time of a --host-rng step (the reference's RNG streams: numpy u, CPU-drawn
DeviceConditioner weights; eager) goes."""
import cProfile, pstats, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "vi-hds_amd")]
import torch
from vihds import synthetic
args, settings, data, parameters, model, training = synthetic.build(
    "dr_constant_icml", 36, 200, solver="rk4", device="cuda:0", seed=1, u_rng="numpy", conditioner_rng="cpu",
    hip_graph=False, nan_check_every=0, learning_rate=0.001, fused_ode_training=True)
model.train()
batch = training.train_data
for _ in range(5): training.step(batch)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20): training.step(batch)
torch.cuda.synchronize(); print("ms/step", (time.perf_counter() - t0) / 20 * 1e3)
pr = cProfile.Profile(); pr.enable()
for _ in range(10): training.step(batch)
torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
