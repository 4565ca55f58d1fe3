#!/usr/bin/env python3
"""Per-parameter relative error of the HIP theta gradient (ODE adjoint, IWAE weights applied) against the oracle's
autograd in float32 AND float64, to tell kernel error from the float32 oracle's own rounding (debugging aid for the
per-parameter parity norm):  python tests/probe/grad_err_by_param.py [fixture] [solver ...]"""
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
name = sys.argv[1] if len(sys.argv) > 1 else "dr_constant_v2_tiny_modeuler"
solvers = sys.argv[2:] or ["modeuler", "rk4"]
fx = Fixture(name)


def oracle_grads(dtype, solver):
    th = {k: v.to(dtype).clone().requires_grad_(v.dtype.is_floating_point) for k, v in fx.theta_dict().items()}
    xs, xp, prec = O.decode(fx.model, th, fx.t("inputs").to(dtype), fx.t("times").to(dtype), solver)
    lpo = O.log_prob_observations(xp, fx.t("observations").to(dtype), prec)
    loss, _ = O.iwae_loss(lpo, fx.t("log_p").to(dtype), fx.t("log_q").to(dtype))
    loss.backward()
    return {n: (th[n].grad if th[n].grad is not None else torch.zeros(fx.B, fx.S, dtype=dtype)) for n in fx.names}


for solver in solvers:
    g32, g64 = oracle_grads(torch.float32, solver), oracle_grads(torch.float64, solver)
    for variant in (1, 2):
        th, row_of = H.pack_theta(fx, DEV)
        th.requires_grad_(True)
        spec = H.spec_for(fx, row_of, th.shape[0], solver, variant)
        traj, xpred, logp = ops.OdeSolveObserve.apply(spec, th, fx.t("inputs", DEV), fx.t("times", DEV),
                                                      fx.t("observations", DEV), None, None)
        loss, log_w, lse = ops.iwae_loss(logp, fx.t("log_p", DEV), fx.t("log_q", DEV))
        loss.backward()
        print("== %s %s variant %d   (columns: max |ref64|, HIP vs 64, oracle32 vs 64)" % (name, solver, variant))
        for i, n in enumerate(fx.names):
            if fx.kinds[i] == O.CONSTANT:
                continue
            m = float(g64[n].abs().max()) + 1e-300
            e_hip = float((th.grad[i].cpu().double() - g64[n]).abs().max()) / m
            e_o32 = float((g32[n].double() - g64[n]).abs().max()) / m
            flag = "  <<<" if e_hip > 5e-4 or e_o32 > 5e-4 else ""
            print("  %-10s %.3e   hip %.2e   oracle32 %.2e%s" % (n, m, e_hip, e_o32, flag))
