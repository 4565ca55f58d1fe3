"""Ad-hoc probe: does the reference's learning rate (0.01, specs/dr_constant_icml.yaml) run away on the REAL plate data
as it does on the synthetic plate (DESIGN.md measurement log, bench.py --lr)?  Trains on the processed dataset the
reference trained on (the rows recorded in tests/golden/trace_dr_constant_icml_modeuler.npz: 234 wells, 86 time points)
with the bench's fast settings (rk4, in-kernel RNG, hipGraph) for N steps of 36-row batches at n_iwae = 200 and prints
the loss every 100 steps.  usage: python tests/probe/real_data_long_run.py [lr] [steps]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "vi-hds_amd"), os.path.join(ROOT, "tests")]
import numpy as np
import torch
import e2e_util as E
from test_e2e_gpu import _TraceDataset
from vihds.config import Config
from vihds.datasets import split_dataset
from vihds.parameters import Parameters
from vihds.training import Training, batch_to_device
from vihds.vae import build_model

lr = float(sys.argv[1]) if len(sys.argv) > 1 else 0.01
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
z = np.load(os.path.join(ROOT, "tests", "golden", "trace_dr_constant_icml_modeuler.npz"))
cfg = json.loads(str(z["config_json"]))
spec = json.loads(str(z["spec_json"]))
spec["params"].update(solver="rk4", u_rng="kernel", conditioner_rng="kernel", hip_graph=True, nan_check_every=0,
                      fused_ode_training=True, fused_iwae_backward=True, learning_rate=lr, learning_boundaries=[10 ** 9])
args = E.make_args(200, seed=0, gpu=0)
np.random.seed(0); torch.manual_seed(0)
settings = Config(args=None, spec=spec)
settings.device = torch.device("cuda:0")
data = split_dataset(_TraceDataset(z), args, settings.data)
parameters = Parameters(settings.params)
model = build_model(args, settings, data, parameters)
training = Training(args, settings, data, parameters, model)
model.train()
ds = data.train.dataset
ids = np.asarray(data.train.indices)
rng = np.random.default_rng(0)
batches = [batch_to_device(ds.times, settings.device, ds[np.sort(rng.choice(ids, 36, replace=False))]) for _ in range(8)]
print("real plate rows: %d train wells, lr %g, %d steps of 36 rows x 200 samples, rk4" % (len(ids), lr, steps))
hist = []
for k in range(steps):
    loss = training.graph_step(batches[k % len(batches)])
    if k % 100 == 99 or k == 0:
        v = float(loss)
        hist.append(v)
        print("step %5d  -ELBO %.4g" % (k + 1, v), flush=True)
        if not np.isfinite(v) or v < -1e8:
            print("objective ran away at step %d" % (k + 1))
            break
else:
    print("finite throughout: first %.4g, min %.4g, last %.4g" % (hist[0], min(hist), hist[-1]))
