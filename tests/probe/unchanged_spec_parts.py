"""Where the time of an unchanged-spec step goes on the host (graph replay with staged host draws)."""
import os, sys, time, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "vi-hds_amd")]
from vihds import synthetic, hostdraws
from vihds.utils import TrainingLogData
args, settings, data, parameters, model, training = synthetic.build("dr_constant_icml", 36, 200, solver="rk4", device="cuda:0", seed=1, learning_rate=0.001)
model.train()
batch = training.train_data
log = TrainingLogData()
for _ in range(10): training._run_batch(time.time(), batch, log)
for _ in range(100): training._run_batch(time.time(), batch, log, next_batch=batch)  # (the helper thread and its pool warm)
torch.cuda.synchronize()
g = list(training._graphs.values())[0][0]
slots = g.host_draws.slots
print("slots:", [tuple(s[0].shape) for s in slots])
def t(fn, n=200):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
print("whole _run_batch        %.3f ms" % t(lambda: training._run_batch(time.time(), batch, log, next_batch=batch), 1000))
print("graph_step (no nan look) %.3f ms" % t(lambda: training.graph_step(batch)))
print("refresh all slots        %.3f ms" % t(lambda: g.host_draws.refresh()))
for k, s in enumerate(slots):
    buf = torch.empty(s[0].numel(), dtype=torch.float32).pin_memory().numpy()
    print("  fill slot %d %s       %.3f ms" % (k, tuple(s[0].shape), t(lambda: s[1](buf))))
print("replay only              %.3f ms" % t(lambda: g.replay()))

from vihds import nprand
import numpy as np
out = np.empty(36 * 200 * 35, np.float32)
for th in (1, 2, 4, 8):
    nprand._THREADS = th
    nprand.randn_f32((36, 200, 35), out)
    ts = []
    for _ in range(60):
        t0 = time.perf_counter(); nprand.randn_f32((36, 200, 35), out); ts.append(time.perf_counter() - t0)
    print("nprand %d threads: median %.3f ms  min %.3f ms" % (th, np.median(ts) * 1e3, min(ts) * 1e3))
