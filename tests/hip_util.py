"""Helpers for the GPU parity tests: pack fixture / synthetic inputs into the HIP path's buffers."""
import numpy as np
import torch

from vihds import hip, ops

PREC_NAMES = ["prec_x", "prec_rfp", "prec_yfp", "prec_cfp"]


def pack_theta(fx, device):
    """theta [R,B,S] = the fixture's P sampled rows followed by the extra rows (aR, aS) condition_theta adds."""
    rows = [fx.t("theta", device)]
    names = list(fx.names)
    if fx.extra_names:
        rows.append(fx.t("extra_theta", device))
        names += fx.extra_names
    theta = torch.cat(rows, 0).contiguous()
    return theta, {n: i for i, n in enumerate(names)}


def spec_for(fx, row_of, n_rows, solver=None, kernel_variant=0):
    # hidden units of the precision network, when the fixture was recorded with --precision_hidden_layers
    key = "decoder_param/ode_model.precisions.prec_hidden.weight"
    n_hidden_prec = int(fx.z[key].shape[0]) if (key in fx.z.files and fx.model != "dr_blackbox") else 0
    return ops.OdeProblemSpec(fx.model, solver or fx.solver, row_of, n_rows, C=fx.z["inputs"].shape[1],
                              D=fx.z["dev_1hot"].shape[1], kernel_variant=kernel_variant, n_hidden_prec=n_hidden_prec)


def view_bsnt(buf):
    """[T,N,B,S] kernel buffer -> the reference's [B,S,N,T] view (vihds/ode.py:82)."""
    return buf.permute(2, 3, 1, 0)


def view_bs4(logp):
    """[4,B,S] -> [B,S,4]"""
    return logp.permute(1, 2, 0)


def theta_inputs(fx, device):
    """q/p tensors for ThetaSampleLogProb from a fixture."""
    kind = fx.t("kind", device, torch.int32)
    q_mu = fx.t("q_mu", device)
    q_prec = fx.t("q_prec", device)
    p_mu = fx.t("p_mu", device)
    p_prec = fx.t("p_prec", device)
    lo, hi = clip_bounds(fx.kinds, p_mu.cpu(), p_prec.cpu(), 4.0)
    return kind, q_mu, q_prec, p_mu, p_prec, lo.to(device), hi.to(device)


def clip_bounds(kinds, p_mu, p_prec, stddevs):
    """p.clip bounds as the reference forms them (distributions.py:332-336, :377-381)."""
    sigma = 1.0 / p_prec.sqrt()
    lo = p_mu - stddevs * sigma
    hi = p_mu + stddevs * sigma
    ln = torch.tensor([k == 1 for k in kinds])
    lo = torch.where(ln, lo.exp(), lo)
    hi = torch.where(ln, hi.exp(), hi)
    return lo.float(), hi.float()
