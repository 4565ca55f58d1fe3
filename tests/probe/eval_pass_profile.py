"""Ad-hoc probe: where an evaluation pass (bench.py --workload config3_eval) spends its time, per host-side stage."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "vi-hds_amd")]
import torch
from vihds import synthetic
B, S = 234, 1000
args, settings, data, parameters, model, training = synthetic.build(
    "dr_constant_icml", B, S, solver="rk4", device="cuda:0", seed=1, u_rng="kernel", conditioner_rng="kernel",
    hip_graph=False, nan_check_every=0)
model.eval()
batch = training.train_data
def sync(): torch.cuda.synchronize(); return time.perf_counter()
for rep in range(3):
    with torch.no_grad():
        t0 = sync(); results, theta, q, p = model(batch, S); t1 = sync()
        out = training.cost(batch, results, theta, q, p, full_output=True); t2 = sync()
    print("forward %.2f ms, cost(full_output) %.2f ms" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3))
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    with torch.no_grad():
        results, theta, q, p = model(batch, S)
        out = training.cost(batch, results, theta, q, p, full_output=True)
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=18, max_name_column_width=60))
