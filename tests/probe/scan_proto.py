#!/usr/bin/env python3
"""Design check for the time-parallel ("scan") form of the dr_constant decoder step (csrc/vihds_dr_scan.hpp).

The double-receiver system (reference models/dr_constant.py:77-112) is linear in every state except OD:
    dx   = gamma(x, t) x,                     gamma = r sigmoid(4 (t - tlag)) (1 - x / K)
    dy_j = F_j(t) - (gamma + delta_j) y_j      j = rfp, f530, f480, luxR, lasR   (F_j = rc a_j, constant)
    dy_j = c_j P_j(luxR, lasR) - (gamma + delta_j) y_j      j = yfp, cfp
so, once the scalar x chain has been walked, one explicit Runge-Kutta step of every other species is an AFFINE map
y_{k+1} = A_k y_k + B_k whose coefficients depend on the x stage values of step k only, and the discrete adjoint is
a linear recurrence in every component (x included: lambda_k = J_k lambda_{k+1} + offset_k).  All per-step work is then
independent across k and the recurrences are prefix scans.  This script states that algorithm with plain loops in
float64 and checks loss / log-likelihoods / theta gradients against the oracle's autograd on a reference fixture,
for all five fixed-grid solvers.  (Run here, CPU only:  python tests/probe/scan_proto.py)"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
from fixture_util import Fixture  # noqa: E402
from oracle import vihds_oracle as O  # noqa: E402


def tableau(solver):
    """(c [NS] stage-time fractions, a [NS][NS] strictly lower, b [NS], fixed_h) -- steps in units of h."""
    if solver == "euler":
        return [0.0], [[0.0]], [1.0], False
    if solver == "midpoint":
        return [0.0, 0.5], [[0, 0], [0.5, 0]], [0.0, 1.0], False
    if solver in ("modeuler", "modeulerwhile"):
        return [0.0, 1.0], [[0, 0], [1.0, 0]], [0.5, 0.5], solver == "modeuler"
    if solver == "rk4":
        return ([0.0, 1 / 3, 2 / 3, 1.0], [[0, 0, 0, 0], [1 / 3, 0, 0, 0], [-1 / 3, 1, 0, 0], [1, -1, 1, 0]],
                [1 / 8, 3 / 8, 3 / 8, 1 / 8], False)
    raise KeyError(solver)


def affine_step(h, a_s, F_s, A_tab, b_tab):
    """One explicit RK step of dy = F_s - a_s y as an affine map: returns (A, B, alpha_s, beta_s) with
    Y_s = alpha_s y + beta_s the stage values and y' = A y + B."""
    NS = len(b_tab)
    kap, rho, al, be = [], [], [], []
    for s in range(NS):
        al_s = 1.0 + h * sum(A_tab[s][r] * kap[r] for r in range(s))
        be_s = h * sum(A_tab[s][r] * rho[r] for r in range(s))
        al.append(al_s)
        be.append(be_s)
        kap.append(-a_s[s] * al_s)
        rho.append(F_s[s] - a_s[s] * be_s)
    A = 1.0 + h * sum(b_tab[s] * kap[s] for s in range(NS))
    B = h * sum(b_tab[s] * rho[s] for s in range(NS))
    return A, B, al, be


def reverse_step(h, a_s, lam1, J_s, A_tab, b_tab):
    """Transposed step of dy = F_s - a_s y: given lam1 = adjoint of y' and stage injections J_s (adjoints arriving at
    the stage values Y_s from elsewhere), returns (lam = adjoint of y, kbar_s).  Linear in (lam1, J)."""
    NS = len(b_tab)
    kbar = [None] * NS
    Ybar = [None] * NS
    lam = lam1
    for s in reversed(range(NS)):
        kbar[s] = h * b_tab[s] * lam1 + h * sum(A_tab[r][s] * Ybar[r] for r in range(s + 1, NS))
        Ybar[s] = J_s[s] - a_s[s] * kbar[s]
        lam = lam + Ybar[s]
    return lam, kbar


def scan_decoder(th, cond, times, obs, solver, version=1):
    """theta dict name -> [n] float64 (n = B*S flattened), cond [n,2] treatments ALREADY clamped (c6, c12), obs [n,4,T].
    Returns logp [n,4] and d sum(logp) / d (effective parameters) as a dict."""
    c_t, A_tab, b_tab, fixed_h = tableau(solver)
    NS = len(b_tab)
    T = len(times)
    K = T - 1
    n = th["r"].shape[0]
    clamp = lambda v, lo, hi: np.minimum(np.maximum(v, lo), hi)  # noqa: E731
    r, Kc = clamp(th["r"], 0, 4), clamp(th["K"], 0, 4)
    invK = 1.0 / Kc
    tlag, rc = th["tlag"], th["rc"]
    delta = {"rfp": clamp(th["drfp"], 1e-12, 2), "yfp": clamp(th["dyfp"], 1e-12, 2), "cfp": clamp(th["dcfp"], 1e-12, 2),
             "f530": 0.0 * r, "f480": 0.0 * r, "luxR": clamp(th["dR"], 1e-12, 5), "lasR": clamp(th["dS"], 1e-12, 5)}
    a_of = {"rfp": 1.0 + 0 * r, "yfp": th["aYFP"], "cfp": th["aCFP"], "f530": th["a530"], "f480": th["a480"],
            "luxR": th["aR"], "lasR": th["aS"]}
    y0 = {"x": th["init_x"], "rfp": th["init_rfp"], "yfp": th["init_yfp"], "cfp": th["init_cfp"], "f530": 0 * r,
          "f480": 0 * r, "luxR": th["init_luxR"], "lasR": th["init_lasR"]}
    fR, fS = th["fR"], th["fS"]  # Hill fractions (prepare stage, unchanged by this design)
    prom = {"yfp": (th["e81"], th["KGR_81"], th["KGS_81"]), "cfp": (th["e76"], th["KGR_76"], th["KGS_76"])}
    prec = [th["prec_x"], th["prec_rfp"], th["prec_yfp"], th["prec_cfp"]]
    h0 = times[1] - times[0]

    # ---- 1. growth-rate table gr[k][s] (state independent) and the serial x chain -------------------------------
    hs = [h0 if fixed_h else times[k + 1] - times[k] for k in range(K)]
    sig = [[1.0 / (1.0 + np.exp(-4.0 * (times[k] + c_t[s] * (times[k + 1] - times[k]) - tlag))) for s in range(NS)]
           for k in range(K)]
    gr = [[r * sig[k][s] for s in range(NS)] for k in range(K)]
    xs = [[None] * NS for _ in range(K)]  # stage values of x
    x = [y0["x"]]
    for k in range(K):
        ks = []
        for s in range(NS):
            xs[k][s] = x[k] + hs[k] * sum(A_tab[s][q] * ks[q] for q in range(s))
            ks.append(gr[k][s] * (1.0 - xs[k][s] * invK) * xs[k][s])
        x.append(x[k] + hs[k] * sum(b_tab[s] * ks[s] for s in range(NS)))
    gam = [[gr[k][s] * (1.0 - xs[k][s] * invK) for s in range(NS)] for k in range(K)]

    # ---- 2. level-1 species: affine maps per step (parallel over k), prefix scan -------------------------------
    Y = {"x": x}
    Amap, stage = {}, {}

    def run_species(j, F_ks):
        Aj, al_j, be_j = [], [], []
        y = [y0[j]]
        for k in range(K):  # (each k independent; the composition below is the scan)
            a_s = [gam[k][s] + delta[j] for s in range(NS)]
            A, Bc, al, be = affine_step(hs[k], a_s, F_ks[k], A_tab, b_tab)
            Aj.append(A)
            al_j.append(al)
            be_j.append(be)
            y.append(A * y[k] + Bc)
        Y[j], Amap[j] = y, Aj
        stage[j] = [[al_j[k][s] * y[k] + be_j[k][s] for s in range(NS)] for k in range(K)]
        return al_j

    alpha = {}
    for j in ("rfp", "f530", "f480", "luxR", "lasR"):
        alpha[j] = run_species(j, [[rc * a_of[j]] * NS for _ in range(K)])
    # ---- 3. promoters at the stage values of luxR / lasR, level-2 species ----------------------------------------
    tfrac, rd, b1, b2 = {}, {}, {}, {}
    for j in ("yfp", "cfp"):
        e, KGR, KGS = prom[j]
        tfrac[j], rd[j] = [[None] * NS for _ in range(K)], [[None] * NS for _ in range(K)]
        F = []
        for k in range(K):
            Fk = []
            for s in range(NS):
                bR = stage["luxR"][k][s] ** 2 * fR
                bS = stage["lasR"][k][s] ** 2 * fS
                kb = KGR * bR + KGS * bS
                rd[j][k][s] = 1.0 / (1.0 + kb)
                tfrac[j][k][s] = kb * rd[j][k][s]
                Fk.append(rc * a_of[j] * (e + (1.0 - e) * tfrac[j][k][s]))
            F.append(Fk)
        alpha[j] = run_species(j, F)

    # ---- 4. log-likelihood ----------------------------------------------------------------------------------------
    LOG2PI = np.log(2 * np.pi)
    logp = np.zeros((n, 4))
    q_inj = [[None] * 4 for _ in range(T)]  # d logp_j / d xpred_j at time k
    precb = [0.0 * r for _ in range(4)]
    inner = lambda k: [1.0 + 0 * r, Y["rfp"][k], Y["yfp"][k] + Y["f530"][k], Y["cfp"][k] + Y["f480"][k]]  # noqa: E731
    for k in range(T):
        inn = inner(k)
        for j in range(4):
            e_ = x[k] * inn[j] - obs[:, j, k]
            logp[:, j] += -0.5 * (LOG2PI - np.log(prec[j]) + prec[j] * e_ * e_)
            q_inj[k][j] = -prec[j] * e_
            precb[j] = precb[j] + 0.5 / prec[j] - 0.5 * e_ * e_

    def grid_inj(j, k):
        if j == "x":
            inn = inner(k)
            return sum(q_inj[k][m] * inn[m] for m in range(4))
        return {"rfp": q_inj[k][1], "yfp": q_inj[k][2], "f530": q_inj[k][2], "cfp": q_inj[k][3], "f480": q_inj[k][3],
                "luxR": 0.0 * r, "lasR": 0.0 * r}[j] * x[k]

    # ---- 5. adjoint: reverse scans, then per-step parameter VJPs ----------------------------------------------------
    g = {}  # parameter adjoints (effective parameters)
    zero = 0.0 * r
    gam_bar = [[zero] * NS for _ in range(K)]   # sum over species of a_bar (others) -- injected into x below
    Lam = {}

    def reverse_scan(j, off):
        """Lam_k = g_k + A_k Lam_{k+1} + off_k, Lam_K = g_K."""
        lam = [None] * T
        lam[K] = grid_inj(j, K)
        for k in reversed(range(K)):
            lam[k] = grid_inj(j, k) + Amap[j][k] * lam[k + 1] + off[k]
        Lam[j] = lam

    def species_vjp(j, J):
        """per-step VJP of species j given Lam[j] (all k independent).  Returns kbar[k][s]."""
        kb_all = []
        dsum, csum = zero, zero
        for k in range(K):
            a_s = [gam[k][s] + delta[j] for s in range(NS)]
            _, kbar = reverse_step(hs[k], a_s, Lam[j][k + 1], J[k], A_tab, b_tab)
            kb_all.append(kbar)
            for s in range(NS):
                abar = -stage[j][k][s] * kbar[s]
                gam_bar[k][s] = gam_bar[k][s] + abar
                dsum = dsum + abar
                csum = csum + kbar[s]
        g["delta_" + j], g["F_" + j] = dsum, csum
        return kb_all

    noJ = [[zero] * NS for _ in range(K)]
    # level 2
    J_R, J_S = [[zero] * NS for _ in range(K)], [[zero] * NS for _ in range(K)]
    for j in ("yfp", "cfp"):
        reverse_scan(j, [zero] * K)
        kb = species_vjp(j, noJ)
        e, KGR, KGS = prom[j]
        c = rc * a_of[j]
        svt, c1b, c2b = zero, zero, zero
        for k in range(K):
            for s in range(NS):
                t_, rd_ = tfrac[j][k][s], rd[j][k][s]
                svt = svt + kb[k][s] * t_
                kbb = kb[k][s] * c * (1.0 - e) * rd_ * (1.0 - t_)   # adjoint of kb = KGR bR + KGS bS
                c1b = c1b + kbb * stage["luxR"][k][s] ** 2          # -> KGR fR
                c2b = c2b + kbb * stage["lasR"][k][s] ** 2          # -> KGS fS
                J_R[k][s] = J_R[k][s] + kbb * KGR * fR * 2.0 * stage["luxR"][k][s]
                J_S[k][s] = J_S[k][s] + kbb * KGS * fS * 2.0 * stage["lasR"][k][s]
        sv = g.pop("F_" + j)
        g["c_" + j] = sv * e + svt * (1.0 - e)   # c enters as c e + c (1 - e) t
        g["e_" + j] = c * (sv - svt)
        g["KGR_" + j], g["KGS_" + j] = c1b * fR, c2b * fS
        g["fR"] = g.get("fR", zero) + c1b * KGR
        g["fS"] = g.get("fS", zero) + c2b * KGS
    # level 1
    for j, J in (("luxR", J_R), ("lasR", J_S), ("rfp", noJ), ("f530", noJ), ("f480", noJ)):
        off = [sum(alpha[j][k][s] * J[k][s] for s in range(NS)) for k in range(K)]
        reverse_scan(j, off)
        species_vjp(j, J)
        g["c_" + j] = g.pop("F_" + j)
    # x: tangent multiplier and offsets from the injections -gam_bar gr / K
    ax = [[-gr[k][s] * (1.0 - 2.0 * xs[k][s] * invK) for s in range(NS)] for k in range(K)]
    Jx = [[-gam_bar[k][s] * gr[k][s] * invK for s in range(NS)] for k in range(K)]
    Ax, offx = [], []
    for k in range(K):
        A, _, al, _ = affine_step(hs[k], ax[k], [zero] * NS, A_tab, b_tab)
        Ax.append(A)
        offx.append(sum(al[s] * Jx[k][s] for s in range(NS)))
    Amap["x"] = Ax
    reverse_scan("x", offx)
    rb, tlb, Kb = zero, zero, zero
    for k in range(K):
        _, kbar = reverse_step(hs[k], ax[k], Lam["x"][k + 1], Jx[k], A_tab, b_tab)
        for s in range(NS):
            gtot = gam_bar[k][s] + kbar[s] * xs[k][s]      # adjoint of gamma_s from every species
            grb = gtot * (1.0 - xs[k][s] * invK)
            rb = rb + grb * sig[k][s]
            tlb = tlb - 4.0 * grb * r * sig[k][s] * (1.0 - sig[k][s])
            Kb = Kb + gtot * gr[k][s] * xs[k][s] * invK * invK
    g["r"], g["tlag"], g["K"] = rb, tlb, Kb
    g["init"] = {j: Lam[j][0] for j in Lam}
    g["prec"] = precb
    return logp, g, Y


def main():
    torch.set_default_dtype(torch.float64)
    fx = Fixture("dr_constant_icml_tiny_modeuler")
    B, S = fx.B, fx.S
    names = fx.names + fx.extra_names
    worst = 0.0
    for solver in ("modeuler", "modeulerwhile", "euler", "midpoint", "rk4"):
        th = {n: v.double().clone().requires_grad_(True) for n, v in fx.theta_dict().items()}
        times, obs, cond = fx.t("times").double(), fx.t("observations").double(), fx.t("inputs").double()
        xs, xp, prec = O.decode(fx.model, th, cond, times, solver)
        lpo = O.log_prob_observations(xp, obs, prec)
        lpo.sum().backward()
        # effective-parameter view for the prototype
        c6, c12 = O._treatments(cond, S)
        thd = {k: v.detach() for k, v in th.items()}
        fR, fS = O._hill_fracs(thd, c6, c12)
        flat = {k: v.reshape(-1).numpy() for k, v in thd.items()}
        flat["fR"], flat["fS"] = fR.reshape(-1).numpy(), fS.reshape(-1).numpy()
        obs_n = obs[:, None].expand(B, S, 4, obs.shape[2]).reshape(B * S, 4, -1).numpy()
        logp, g, Y = scan_decoder(flat, None, times.numpy(), obs_n, solver)
        err = lambda a, b: float(np.abs(a - b).max() / (np.abs(b).max() + 1e-300))  # noqa: E731
        e_lp = max(err(logp[:, j], lpo[:, :, j].detach().reshape(-1).numpy()) for j in range(4))
        e_traj = max(err(np.stack(Y[n_], 1), xs[:, :, i].detach().reshape(B * S, -1).numpy())
                     for i, n_ in enumerate(["x", "rfp", "yfp", "cfp", "f530", "f480", "luxR", "lasR"]))
        G = lambda k: th[k].grad.reshape(-1).numpy()  # noqa: E731
        rcv = flat["rc"]
        checks = {
            "r": (g["r"] * ((flat["r"] >= 0) & (flat["r"] <= 4)), G("r")), "K": (g["K"] * ((flat["K"] >= 0) & (flat["K"] <= 4)), G("K")),
            "tlag": (g["tlag"], G("tlag")),
            "drfp": (g["delta_rfp"], G("drfp")), "dyfp": (g["delta_yfp"], G("dyfp")), "dcfp": (g["delta_cfp"], G("dcfp")),
            "dR": (g["delta_luxR"], G("dR")), "dS": (g["delta_lasR"], G("dS")),
            "aYFP": (g["c_yfp"] * rcv, G("aYFP")), "aCFP": (g["c_cfp"] * rcv, G("aCFP")),
            "a530": (g["c_f530"] * rcv, G("a530")), "a480": (g["c_f480"] * rcv, G("a480")),
            "e81": (g["e_yfp"], G("e81")), "e76": (g["e_cfp"], G("e76")),
            "KGR_81": (g["KGR_yfp"], G("KGR_81")), "KGS_81": (g["KGS_yfp"], G("KGS_81")),
            "KGR_76": (g["KGR_cfp"], G("KGR_76")), "KGS_76": (g["KGS_cfp"], G("KGS_76")),
            "rc": (g["c_rfp"] + g["c_yfp"] * flat["aYFP"] + g["c_cfp"] * flat["aCFP"] + g["c_f530"] * flat["a530"] +
                   g["c_f480"] * flat["a480"] + g["c_luxR"] * flat["aR"] + g["c_lasR"] * flat["aS"], G("rc")),
            "prec_x": (g["prec"][0], G("prec_x")), "prec_cfp": (g["prec"][3], G("prec_cfp")),
        }
        e_g = {k: err(a, b) for k, (a, b) in checks.items()}
        print("%-14s traj %.1e  logp %.1e  grads max %.1e (%s)" % (solver, e_traj, e_lp, max(e_g.values()),
                                                                    max(e_g, key=e_g.get)))
        worst = max(worst, e_traj, e_lp, max(e_g.values()))
    assert worst < 1e-9, worst
    print("scan formulation == oracle autograd (float64)")


if __name__ == "__main__":
    main()
