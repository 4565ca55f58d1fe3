import sys, os, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "vi-hds_amd"), os.path.join(ROOT, "tests")]
from fixture_util import Fixture
import e2e_util as E
from vihds.training import Training
from vihds.vae import build_model
from vihds import ops
fx = Fixture("dr_constant_icml_tiny_modeuler")
orig = ops.device_condition
def spy(z, *a, **k):
    print("  device_condition z:", None if z is None else (tuple(z.shape), z.data_ptr(), z.flatten()[:3].tolist() if not torch.cuda.is_current_stream_capturing() else "capturing"))
    return orig(z, *a, **k)
ops.device_condition = spy
args, settings, data, parameters = E.build_from_fixture(fx, gpu=0, hip_graph=True)
model = build_model(args, settings, data, parameters)
training = Training(args, settings, data, parameters, model)
batch = E.batch_from_fixture(fx, settings.device)
model.eval()
np.random.seed(11); torch.manual_seed(11)
for k in range(3):
    o = training.evaluate(batch, 8)
    g, staged = list(training._eval_graphs.values())[0]
    print("pass", k, float(o.elbo), "slots", [(tuple(s[0].shape), s[0].data_ptr(), s[0].flatten()[:3].tolist()) for s in g.host_draws.slots])
zbuf = g.host_draws.slots[1][0]
print("z after replay:", zbuf.flatten().tolist())
print("pinned:", [p.tolist() for p in (g.host_draws.slots[1][2] or [])])
# refresh only (no replay): does the copy land?
g.host_draws.refresh(); torch.cuda.synchronize()
print("z after refresh only:", zbuf.flatten()[:5].tolist())
g.replay(); torch.cuda.synchronize()
print("z after replay:", zbuf.flatten()[:5].tolist())
print("q_values:", [np.asarray(v).ravel()[:3] for v in o.q_values][:4])
lo = zbuf.data_ptr(); hi = lo + zbuf.numel() * 4
def chk(name, t):
    a = t.data_ptr(); b = a + t.numel() * t.element_size()
    if a < hi and lo < b: print("OVERLAP with", name, tuple(t.shape), a - lo, b - lo)
chk("flat", staged["flat"])
for k, t in enumerate(staged["theta_rows"]): chk("theta_row%d" % k, t)
print("flat range", staged["flat"].data_ptr() - lo, staged["flat"].numel())
print("z storage:", zbuf.untyped_storage().data_ptr() - lo, zbuf.untyped_storage().nbytes(), "u storage", g.host_draws.slots[0][0].untyped_storage().nbytes())
