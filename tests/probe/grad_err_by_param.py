#!/usr/bin/env python3
"""Per-parameter relative error of the HIP theta gradient vs the reference fixtures (debugging aid for the
per-parameter parity norm): python tests/probe/grad_err_by_param.py [fixture ...]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "vi-hds_amd"), os.path.join(ROOT, "tests")]
import torch  # noqa: E402

import hip_util as H  # noqa: E402
from fixture_util import Fixture  # noqa: E402
from oracle import vihds_oracle as O  # noqa: E402
from vihds import ops  # noqa: E402

DEV = "cuda:0"
names = sys.argv[1:] or ["dr_constant_v2_tiny_modeuler", "dr_constant_icml_tiny_modeuler"]
for name in names:
    fx = Fixture(name)
    for variant in (1, 2):
        th, row_of = H.pack_theta(fx, DEV)
        th.requires_grad_(True)
        spec = H.spec_for(fx, row_of, th.shape[0], None, variant)
        traj, xpred, logp = ops.OdeSolveObserve.apply(spec, th, fx.t("inputs", DEV), fx.t("times", DEV),
                                                      fx.t("observations", DEV), None, None)
        loss, log_w, lse = ops.iwae_loss(logp, fx.t("log_p", DEV), fx.t("log_q", DEV))
        loss.backward()
        thc = fx.theta_dict(requires_grad=True)
        qm, qp = fx.q_params()
        pm, pp = fx.p_params()
        vals = [thc[n] for n in fx.names]
        lw_extra = O.chained_log_prob(fx.kinds, pm, pp, vals) - O.chained_log_prob(fx.kinds, qm, qp, vals)
        w = torch.softmax(log_w.detach().cpu(), dim=1) * (-1.0 / fx.B)
        (lw_extra * w).sum().backward()
        extra = torch.stack([thc[n].grad if thc[n].grad is not None else torch.zeros(fx.B, fx.S) for n in fx.names])
        got = th.grad[: len(fx.names)].cpu() + extra
        ref = fx.t("theta_grad")
        # the oracle's own autograd on the CPU for comparison
        tho = fx.theta_dict(requires_grad=True)
        xs, xp, prec = O.decode(fx.model, tho, fx.t("inputs"), fx.t("times"), fx.solver)
        lpo = O.log_prob_observations(xp, fx.t("observations"), prec)
        lo, _ = O.iwae_loss(lpo, fx.t("log_p"), fx.t("log_q"))
        lo.backward()
        print("== %s variant %d" % (name, variant))
        for i, n in enumerate(fx.names):
            if fx.kinds[i] == O.CONSTANT:
                continue
            d = (got[i] - ref[i]).abs().max()
            m = ref[i].abs().max()
            og = tho[n].grad if tho[n].grad is not None else torch.zeros(fx.B, fx.S)
            d_ode = (th.grad[i].cpu() - og).abs().max()
            print("  %-10s ref max %.3e  err %.2e  rel %.2e | ODE part: kernel vs oracle autograd abs %.2e (max %.2e) | extra max %.2e"
                  % (n, m, d, d / (m + 1e-30), d_ode, og.abs().max(), extra[i].abs().max()))
