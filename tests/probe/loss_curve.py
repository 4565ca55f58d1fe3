"""Ad-hoc probe (not a test): loss trajectory of the bench workload over many steps, graph vs eager."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "vi-hds_amd")]
import torch
from vihds import synthetic

mode = sys.argv[1] if len(sys.argv) > 1 else "graph"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
seed = int(sys.argv[3]) if len(sys.argv) > 3 else 1
use_graph = mode == "graph"
args, settings, data, parameters, model, training = synthetic.build(
    "dr_constant_icml", 36, 200, solver="rk4", device="cuda:0", seed=seed, shard=None, u_rng="kernel",
    conditioner_rng="kernel", hip_graph=use_graph, nan_check_every=0)
model.train()
batch = training.train_data
step = training.graph_step if use_graph else training.step
first_bad = None
for it in range(steps):
    loss = step(batch)
    if it % 100 == 0 or it == steps - 1:
        print(mode, it, float(loss), flush=True)
    elif first_bad is None and it % 10 == 0 and not torch.isfinite(loss):
        first_bad = it
        print(mode, "first non-finite near", it, flush=True)
        bad = [n for n, p in model.named_parameters() if not torch.isfinite(p).all()]
        print("non-finite params:", bad[:10])
        break
