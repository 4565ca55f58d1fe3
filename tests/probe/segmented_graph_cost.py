"""Ad-hoc probe (not a test): replay cost of a training step captured as 1 hipGraph vs 3 segments (no collectives,
single process) -- isolates the per-segment launch cost from communicator / GPU-sharing effects."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "vi-hds_amd")]
import torch
from vihds import parallel, synthetic

args, settings, data, parameters, model, training = synthetic.build(
    "dr_constant_icml", 36, 200, solver="rk4", device="cuda:0", seed=1, shard=None, u_rng="kernel",
    conditioner_rng="kernel", hip_graph=True, nan_check_every=0)
model.train()
batch = training.train_data
one = torch.ones((), device="cuda:0")

def step_with_breaks(nbreaks):
    batch_results, theta, q, p = model(batch, args.train_samples)
    elbo = training.cost(batch, batch_results, theta, q, p).elbo
    if nbreaks >= 1: parallel.graph_break(lambda: None)
    elbo.backward(one.expand_as(elbo))
    if nbreaks >= 2: parallel.graph_break(lambda: None)
    training.optimizer.step()
    return elbo.detach()

for nb in (0, 1, 2):
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3): step_with_breaks(0); training.optimizer.zero_grad(set_to_none=True)
    torch.cuda.current_stream().wait_stream(s)
    training.optimizer.zero_grad(set_to_none=True)
    g = parallel.SegmentedGraph()
    g.capture(lambda: step_with_breaks(nb))
    for _ in range(20): g.replay()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(300): g.replay()
    torch.cuda.synchronize()
    print("segments=%d: %.3f ms/step, loss %.3f" % (nb + 1, (time.perf_counter() - t0) / 300 * 1e3, float(g.result)), flush=True)
    training.optimizer.zero_grad(set_to_none=True)
