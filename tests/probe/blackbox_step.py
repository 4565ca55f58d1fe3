"""Ad-hoc probe (not a test): full training step rate of the black-box workload (BASELINE config 4:
dr_blackbox_icml, B=36, S=200, T=86, midpoint), hipGraph replay, with the per-kernel breakdown under rocprofv3."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "vi-hds_amd")]
import torch
from vihds import synthetic
solver = sys.argv[1] if len(sys.argv) > 1 else "midpoint"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 300
args, settings, data, parameters, model, training = synthetic.build(
    "dr_blackbox_icml", 36, 200, solver=solver, device="cuda:0", seed=1, u_rng="kernel", conditioner_rng="kernel",
    hip_graph=True, nan_check_every=0, learning_rate=0.001)
model.train()
batch = training.train_data
for _ in range(20): loss = training.graph_step(batch)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(steps): loss = training.graph_step(batch)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / steps
print("dr_blackbox_icml %s B=36 S=200: %.3f ms/step = %.0f steps/s, loss %.3f" % (solver, dt * 1e3, 1 / dt, float(loss)), flush=True)
