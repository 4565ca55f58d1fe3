"""probe: where one evaluation pass (config 3's evaluation shape) spends its time -- graph replay until the device is idle,
then the host side of Results.  usage: python tests/probe/eval_pass_parts.py"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "vi-hds_amd"))
from vihds import synthetic  # noqa: E402
from vihds.utils import Results  # noqa: E402

args, settings, data, parameters, model, training = synthetic.build(
    "dr_constant_icml", 234, 1000, solver="rk4", device="cuda:0", seed=0, u_rng="kernel", conditioner_rng="kernel",
    hip_graph=True, nan_check_every=0)
model.eval()
batch = training.train_data
for _ in range(5):
    training.evaluate(batch, 1000)
g, staged = training._eval_graphs[(id(batch), 1000)]
torch.cuda.synchronize()
n = 200
t0 = time.perf_counter()
for _ in range(n):
    g.replay()
    torch.cuda.synchronize()
t1 = time.perf_counter()
for _ in range(n):
    staged["host_ring"][0].copy_(staged["flat"], non_blocking=True)
    out = Results()
    out.init_from_staged(model.decoder.state_names, staged, staged["host_ring"][0])
t2 = time.perf_counter()
for _ in range(n):
    training.evaluate(batch, 1000)
t3 = time.perf_counter()
print("graph replay + sync      %7.1f us" % ((t1 - t0) / n * 1e6))
print("copy to host + Results  %7.1f us" % ((t2 - t1) / n * 1e6))
print("Training.evaluate        %7.1f us" % ((t3 - t2) / n * 1e6))
