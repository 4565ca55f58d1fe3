"""cProfile of the unchanged-spec step's host side (Training._run_batch with staged host draws, two draws queued)."""
import cProfile, os, pstats, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "vi-hds_amd")]
import torch
from vihds import synthetic
from vihds.utils import TrainingLogData
args, settings, data, parameters, model, training = synthetic.build("dr_constant_icml", 36, 200, solver="rk4", device="cuda:0", seed=1, learning_rate=0.001)
model.train()
batch = training.train_data
log = TrainingLogData()
for _ in range(200):
    training._run_batch(time.time(), batch, log, next_batch=batch, ahead=2)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(2000):
    training._run_batch(time.time(), batch, log, next_batch=batch, ahead=2)
torch.cuda.synchronize()
print("plain: %.4f ms/step" % ((time.perf_counter() - t0) / 2000 * 1e3))
pr = cProfile.Profile()
pr.enable()
for _ in range(2000):
    training._run_batch(time.time(), batch, log, next_batch=batch, ahead=2)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(22)
