"""Ad-hoc probe: a dr_blackbox training step at BASELINE config 4's shape (B=36, S=200, T=86, midpoint) with the
reference's DEFAULT n_hidden_decoder = 50 (vihds/config.py:71), i.e. through the per-size side library
libvihds_bb_2_50_20_12.so (round 3: its matrix-core kernels on cooperating wavefronts; thread-per-trajectory kernels +
vihds_gram_blocks before), next to the ICML sizes
(matrix-core kernels of libvihds_hip.so).  Prints the step time and, with synchronising timers around them, the ODE
launches and the weight-gradient contraction."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "vi-hds_amd"), os.path.join(ROOT, "tests")]
import torch
from vihds import ops, synthetic


class SyncTimer(object):
    def __init__(self):
        self.t = {}

    def launch(self, name, fn):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = fn()
        torch.cuda.synchronize()
        self.t.setdefault(name, []).append(time.perf_counter() - t0)
        return out


for hs in (25, 50):
    args, settings, data, parameters, model, training = synthetic.build(
        "dr_blackbox_icml", 36, 200, solver="midpoint", device="cuda:0", seed=1, u_rng="kernel", conditioner_rng="kernel",
        hip_graph=False, nan_check_every=0, learning_rate=0.001, n_hidden_decoder=hs)
    model.train()
    batch = training.train_data
    for _ in range(3):
        loss = training.step(batch)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 20
    for _ in range(n):
        loss = training.step(batch)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print("n_hidden_decoder %d: %.3f ms/step = %.0f steps/s (eager launches), loss %.4f" % (hs, dt * 1e3, 1.0 / dt, float(loss)))
    tm = SyncTimer()
    ops.TIMER = tm
    orig = ops.blackbox_weight_grads

    def timed(*a, **k):
        return tm.launch("weight-gradient contraction", lambda: orig(*a, **k))

    ops.blackbox_weight_grads = timed
    for _ in range(5):
        training.step(batch)
    ops.TIMER = None
    ops.blackbox_weight_grads = orig
    for k, v in tm.t.items():
        print("    %-32s %.3f ms" % (k, 1e3 * sum(v[1:]) / max(len(v) - 1, 1)))

# the same 50-unit step replayed from a hipGraph (two contraction passes and all)
args, settings, data, parameters, model, training = synthetic.build(
    "dr_blackbox_icml", 36, 200, solver="midpoint", device="cuda:0", seed=1, u_rng="kernel", conditioner_rng="kernel",
    hip_graph=True, nan_check_every=0, learning_rate=0.001, n_hidden_decoder=50)
model.train()
batch = training.train_data
for _ in range(3):
    loss = training.graph_step(batch)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    loss = training.graph_step(batch)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 20
print("n_hidden_decoder 50, hipGraph replay: %.3f ms/step = %.0f steps/s, loss %.4f" % (dt * 1e3, 1.0 / dt, float(loss)))
