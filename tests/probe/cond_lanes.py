"""Which device columns reach the decoder launch's conditioner sum (fast path), one column at a time."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "vi-hds_amd"), os.path.join(ROOT, "tests")]
import torch
from fixture_util import Fixture
from vihds import ops
import hip_util as H
DEV = "cuda:0"
D = int(sys.argv[1]) if len(sys.argv) > 1 else 16
fx = Fixture("dr_constant_icml_tiny_modeuler")
P, B, S = len(fx.names), fx.B, fx.S
E = len(fx.extra_names)
th, row_of = H.pack_theta(fx, DEV)
spec3 = ops.OdeProblemSpec(fx.model, fx.solver, row_of, th.shape[0], C=fx.z["inputs"].shape[1], D=D, kernel_variant=3)
kind, q_mu, q_prec, p_mu, p_prec, lo, hi = H.theta_inputs(fx, DEV)
q_all = torch.cat([q_mu, q_prec.log()], 0).contiguous()
rows = torch.arange(2 * P, dtype=torch.int32, device=DEV)
out = []
for col in range(D):
    dev = torch.zeros(B, D); dev[:, col] = 1.0
    rel = torch.ones(E, D); dflt = torch.zeros(E, dtype=torch.int32)
    z = torch.arange(E * D, dtype=torch.float32).reshape(E, D) + 1.0
    cond_job = (E, P, 0.0, 1.0, z.to(DEV), None, rel.to(DEV), dflt.to(DEV))
    with torch.no_grad():
        theta, *_ = ops.DecoderStepFused.apply(q_all, kind, p_mu, p_prec, lo, hi, fx.t("u", DEV), P + E, rows, spec3, fx.t("inputs", DEV),
                                               fx.t("times", DEV), fx.t("observations", DEV), dev.to(DEV), cond_job)
    out.append([float(theta[P + e, 0, 0]) for e in range(E)])
print("column -> (row 0, row 1); expected (col + 1, D + col + 1)")
print(out)
