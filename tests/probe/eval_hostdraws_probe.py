import sys, os, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "vi-hds_amd"), os.path.join(ROOT, "tests")]
from fixture_util import Fixture
import e2e_util as E
from vihds.training import Training
from vihds.vae import build_model
fx = Fixture("dr_constant_icml_tiny_modeuler")
res = {}
import itertools
combo = sys.argv[1:] if len(sys.argv) > 1 else ["numpy", "cpu"]
for graph in (False, True):
    args, settings, data, parameters = E.build_from_fixture(fx, gpu=0, hip_graph=graph, u_rng=combo[0], conditioner_rng=combo[1])
    model = build_model(args, settings, data, parameters)
    training = Training(args, settings, data, parameters, model)
    batch = E.batch_from_fixture(fx, settings.device)
    model.eval()
    np.random.seed(11); torch.manual_seed(11)
    outs = []
    for k in range(4):
        o = training.evaluate(batch, 8)
        outs.append((float(o.elbo), np.asarray(o.theta).copy(), o.iw_predict_mu.copy(), float(model.decoder.ode_model.aR.sum()) if hasattr(model.decoder.ode_model, "aR") and model.decoder.ode_model.aR is not None else 0.0))
    res[graph] = outs
    print("graph", graph, [x[0] for x in outs], "numpy next", np.random.rand(), "torch next", float(torch.rand(1)))
for a, b in zip(res[False], res[True]):
    print(a[0] == b[0], np.abs(a[1] - b[1]).max(), np.abs(a[2] - b[2]).max(), a[3], b[3])
