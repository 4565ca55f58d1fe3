"""GPU parity of BASELINE config 5's real model (relay_constant_precisions) and its siblings -- the `*_precisions` forms of
relay / degrader / inducer / prpr, the only specs of those models the reference ships -- through the C ABI, for BOTH
kernel families (kernel_variant 0 = one lane per state, csrc/vihds_relay_lanes.hpp; 1 = one thread per trajectory,
csrc/vihds_ode_kernels.hpp; inducer_constant has no lane model: its variant 0 is variant 1), against

  (1) fixtures recorded from the MODIFIED reference (`make_fixtures.py --patched`: OdeFunc.__init__'s arity and the
      non-existent init_with_params repaired in memory, equations untouched: fixture_util.PATCHED_FIXTURES) -- the
      reference's own Relay_Constant_RHS.forward / Degrader_Constant_RHS.forward integrated by its own modeuler /
      modeulerwhile (models/relay_constant.py:91-134,199-264; degrader_constant.py:103-143; vihds/precisions.py:55-61,76-87);
  (2) the oracle (itself pinned on those fixtures, tests/test_oracle_golden.py) for the spec's solver `midpoint`, `rk4`
      and `modeuler`, on a ragged shape and -- sub-sampled -- at config 5's full size B=36, S=200, T=99.

Tolerances: 1e-4 relative per species / signal for trajectories, precisions, predictions, log-likelihoods and the loss;
5e-4 per parameter (max-norm) for every theta gradient and the network's weight gradients (north_star's bound; for a
parameter whose float32 gradient is itself unresolved the yardstick is the float64 oracle, as in
test_all_solvers_match_oracle_forward_and_gradient)."""
import pytest
import torch

from fixture_util import PATCHED_FIXTURES, Fixture, rel_err
from oracle import vihds_oracle as O

pytestmark = pytest.mark.gpu

TOL = 1e-4
GTOL = 5e-4
DEV = "cuda:0"
KEYS = ("prod_w", "prod_b", "degr_w", "degr_b")
N_CORE = {"relay": 12, "degrader": 11, "inducer": 5, "prpr": 6}


def _flat(prec_w):
    return torch.cat([prec_w[k].detach().reshape(-1) for k in KEYS])


def _assert_theta_grads(names, got, ref32, ref64, B, S, kinds=None):
    """Per parameter: GTOL, or eight times the float32 oracle's own distance from its float64 run where float32 does not
    resolve the gradient."""
    for r, n in enumerate(names):
        if kinds is not None and kinds[r] == O.CONSTANT:
            continue
        g64 = ref64[r]
        scale = float(g64.abs().max())
        if scale == 0.0:
            assert float(got[r].abs().max()) == 0.0, n
            continue
        e32 = float((ref32[r].double() - g64).abs().max()) / scale
        e_hip = float((got[r].double() - g64).abs().max()) / scale
        assert e_hip < max(GTOL, 8.0 * e32), (n, e_hip, e32)


def _oracle_run(model, th, cond, times, obs, solver, prec_w, log_p, log_q, dtype=torch.float32):
    """The oracle's loss and its gradients w.r.t. every theta tensor and the network weights, in `dtype`."""
    c = lambda v: v.detach().to(dtype).clone()  # noqa: E731
    thc = {k: c(v).requires_grad_(True) for k, v in th.items()}
    w = {k: c(v).requires_grad_(True) for k, v in prec_w.items()}
    xs, xp, prec = O.decode(model, thc, c(cond), c(times), solver, prec_w=w)
    lpo = O.log_prob_observations(xp, c(obs), prec)
    loss, _ = O.iwae_loss(lpo, c(log_p), c(log_q))
    loss.backward()
    return dict(xs=xs.detach(), xp=xp.detach(), prec=prec.detach(), lpo=lpo.detach(), loss=loss.detach(),
                th_grad={k: (v.grad if v.grad is not None else torch.zeros_like(v)) for k, v in thc.items()},
                w_grad={k: v.grad for k, v in w.items()})


def _hip_run(model, names, theta, cond, times, obs, solver, wts, log_p, log_q, variant):
    from vihds import ops

    th = theta.to(DEV).clone().requires_grad_(True)
    w = wts.to(DEV).clone().requires_grad_(True)
    spec = ops.OdeProblemSpec(model, solver, {n: i for i, n in enumerate(names)}, len(names), C=cond.shape[1],
                              kernel_variant=variant)
    traj, xpred, logp = ops.OdeSolveObserve.apply(spec, th, cond.to(DEV), times.to(DEV), obs.to(DEV), None, w)
    loss, log_w, _ = ops.iwae_loss(logp, log_p.to(DEV), log_q.to(DEV))
    loss.backward()
    return dict(traj=traj.detach().permute(2, 3, 1, 0).cpu(), xp=xpred.detach().permute(2, 3, 1, 0).cpu(),
                lpo=logp.detach().permute(1, 2, 0).cpu(), loss=loss.detach().cpu(), log_w=log_w.detach().cpu(),
                th_grad=th.grad.cpu(), w_grad=w.grad.cpu())


@pytest.mark.parametrize("variant", [0, 1])
@pytest.mark.parametrize("name", PATCHED_FIXTURES)
def test_precisions_models_match_the_modified_reference(name, variant):
    """HIP vs the MODIFIED reference's own output: trajectories of all species, the four precision states, x_predict,
    log-likelihood per signal, loss, d loss / d theta for every sampled parameter and all network-weight gradients
    (relay: 2 x (4 x 13 + 4) = 112)."""
    fx = Fixture(name)
    n_core = N_CORE[fx.model.split("_")[0]]
    prec_w, _, _ = fx.decoder_weights()
    theta = fx.t("theta")
    out = _hip_run(fx.model, fx.names, theta, fx.t("inputs"), fx.t("times"), fx.t("observations"), fx.solver,
                   _flat(prec_w), fx.t("log_p"), fx.t("log_q"), variant)
    assert out["traj"].shape[2] == n_core + 4
    st = int(fx.z["sample_stride"])  # (the full-size fixture -- config 5's own 36 x 200 -- keeps every 25th sample's trajectory)
    assert rel_err(out["traj"][:, ::st, :n_core], fx.t("x_states")) < TOL
    assert rel_err(out["traj"][:, ::st, n_core:], fx.t("precisions")) < TOL
    assert rel_err(out["xp"][:, ::st], fx.t("x_predict")) < TOL
    assert rel_err(out["lpo"], fx.t("log_p_by_species"), dim=2) < TOL
    assert rel_err(out["loss"], fx.t("loss")) < TOL
    # the reference's d loss / d theta_i also holds the log p - log q terms: added analytically (oracle functions), so
    # that the ODE adjoint alone is under test
    thc = fx.theta_dict(requires_grad=True)
    qm, qp = fx.q_params()
    pm, pp = fx.p_params()
    vals = [thc[n] for n in fx.names]
    lw_extra = O.chained_log_prob(fx.kinds, pm, pp, vals) - O.chained_log_prob(fx.kinds, qm, qp, vals)
    (lw_extra * (torch.softmax(out["log_w"], dim=1) * (-1.0 / fx.B))).sum().backward()
    extra = torch.stack([thc[n].grad if thc[n].grad is not None else torch.zeros(fx.B, fx.S) for n in fx.names])
    live = torch.tensor([k != O.CONSTANT for k in fx.kinds])
    assert rel_err((out["th_grad"] + extra)[live], fx.t("theta_grad")[live], dim=0) < GTOL
    ref = fx.decoder_weight_grads()
    off = 0
    for k, key in zip(KEYS, ("prec_production.weight", "prec_production.bias", "prec_degradation.weight",
                             "prec_degradation.bias")):
        g = ref["ode_model.precisions." + key]
        assert rel_err(out["w_grad"][off: off + g.numel()].reshape(g.shape), g) < GTOL, key
        off += g.numel()
    assert off == out["w_grad"].numel()


_ORACLE_CACHE = {}


@pytest.mark.parametrize("variant", [0, 1])
@pytest.mark.parametrize("solver", ["midpoint", "modeuler", "rk4"])
@pytest.mark.parametrize("name", ["relay_constant_precisions_tiny_modeulerwhile",  # B=3, S=5: ragged (15 trajectories)
                                  "relay_constant_precisions_tiny_modeuler",
                                  "degrader_constant_precisions_tiny_modeuler"])
def test_config5_models_match_oracle_every_solver(name, solver, variant):
    """relay_constant_precisions (config 5; `midpoint` is its spec's solver) and degrader_constant_precisions on the
    fixtures' inputs, HIP vs the oracle's decode: trajectories, the four precision states, x_predict, log-likelihood,
    loss; every theta gradient and every network-weight gradient, per parameter."""
    fx = Fixture(name)
    n_core = N_CORE[fx.model.split("_")[0]]
    prec_w, _, _ = fx.decoder_weights()
    th = fx.theta_dict()
    args = (fx.model, th, fx.t("inputs"), fx.t("times"), fx.t("observations"), solver, prec_w, fx.t("log_p"), fx.t("log_q"))
    # (the oracle's two runs -- float32 and the float64 yardstick -- are most of this test's time and do not depend on the
    # kernel family: computed once per (fixture, solver), shared by the variants)
    if (name, solver) not in _ORACLE_CACHE:
        _ORACLE_CACHE.clear()
        _ORACLE_CACHE[(name, solver)] = (_oracle_run(*args), _oracle_run(*args, dtype=torch.float64))
    o32, o64 = _ORACLE_CACHE[(name, solver)]
    out = _hip_run(fx.model, fx.names, fx.t("theta"), fx.t("inputs"), fx.t("times"), fx.t("observations"), solver,
                   _flat(prec_w), fx.t("log_p"), fx.t("log_q"), variant)
    assert rel_err(out["traj"][:, :, :n_core], o32["xs"]) < TOL
    assert rel_err(out["traj"][:, :, n_core:], o32["prec"]) < TOL
    assert rel_err(out["xp"], o32["xp"]) < TOL
    assert rel_err(out["lpo"], o32["lpo"], dim=2) < TOL
    assert rel_err(out["loss"], o32["loss"]) < TOL
    _assert_theta_grads(fx.names, out["th_grad"], [o32["th_grad"][n] for n in fx.names],
                        [o64["th_grad"][n] for n in fx.names], fx.B, fx.S, fx.kinds)
    off = 0
    for k in KEYS:
        g32, g64 = o32["w_grad"][k], o64["w_grad"][k]
        got = out["w_grad"][off: off + g32.numel()].reshape(g32.shape)
        scale = float(g64.abs().max())
        e32 = float((g32.double() - g64).abs().max()) / scale
        e_hip = float((got.double() - g64).abs().max()) / scale
        assert e_hip < max(GTOL, 8.0 * e32), (k, e_hip, e32)
        off += g32.numel()


@pytest.mark.parametrize("model", ["relay_constant_precisions", "degrader_constant_precisions"])
def test_config5_full_size_subsample_against_oracle(model):
    """BASELINE config 5's shape (B=36, S=200, T=99, midpoint) on the device; the oracle on a sub-sample of its
    trajectories (3 rows x 8 samples, spread over blocks and wavefronts).  The upstream gradient of the log-likelihood is
    non-zero on the sub-sample only, so the kernel's theta gradients there AND its network-weight gradients (sums over
    all 7 200 trajectories, 7 176 of them with a zero upstream gradient) are comparable with the oracle's."""
    from vihds import hip, ops
    from test_hip_parity import _relay_problem

    B, S, T = 36, 200, 99
    slots, theta, cond, times, obs, wts = _relay_problem(model, B, S, T, 3, dt=0.17)
    n_core = N_CORE[model.split("_")[0]]
    rows = torch.tensor([0, 17, 35])
    cols = torch.tensor([0, 1, 63, 64, 77, 130, 198, 199])
    g = torch.Generator().manual_seed(12)
    up_sub = torch.randn(4, len(rows), len(cols), generator=g) * 1e-2
    up = torch.zeros(4, B, S)
    up[:, rows[:, None], cols[None, :]] = up_sub

    n_in = 1 + n_core
    wc = wts.cpu()
    sizes = [4 * n_in, 4, 4 * n_in, 4]
    parts = torch.split(wc, sizes)
    res = {}
    for dtype in (torch.float32, torch.float64):
        prec_w = {"prod_w": parts[0].reshape(4, n_in), "prod_b": parts[1], "degr_w": parts[2].reshape(4, n_in), "degr_b": parts[3]}
        prec_w = {k: v.to(dtype).clone().requires_grad_(True) for k, v in prec_w.items()}
        thc = {n: theta[i].cpu()[rows[:, None], cols[None, :]].to(dtype).clone().requires_grad_(True) for i, n in enumerate(slots)}
        xs, xp, prec = O.decode(model, thc, cond.cpu()[rows].to(dtype), times.cpu().to(dtype), "midpoint", prec_w=prec_w)
        lpo = O.log_prob_observations(xp, obs.cpu()[rows].to(dtype), prec)  # [b,s,4]
        (lpo * up_sub.permute(1, 2, 0).to(dtype)).sum().backward()
        res[dtype] = dict(xs=xs.detach(), prec=prec.detach(), xp=xp.detach(), lpo=lpo.detach(),
                          th=[thc[n].grad if thc[n].grad is not None else torch.zeros(len(rows), len(cols), dtype=dtype) for n in slots],
                          w=torch.cat([prec_w[k].grad.reshape(-1) for k in KEYS]))
    o32, o64 = res[torch.float32], res[torch.float64]
    for variant in (0, 1):
        th = theta.clone().requires_grad_(True)
        w = wts.clone().requires_grad_(True)
        spec = ops.OdeProblemSpec(model, "midpoint", {n: i for i, n in enumerate(slots)}, len(slots), C=cond.shape[1],
                                  kernel_variant=variant)
        traj, xpred, logp = ops.OdeSolveObserve.apply(spec, th, cond, times, obs, None, w)
        (logp * up.to(DEV)).sum().backward()
        assert torch.isfinite(traj).all() and torch.isfinite(th.grad).all() and torch.isfinite(w.grad).all()

        sub = lambda x: x[:, :, rows[:, None], cols[None, :]]  # noqa: E731  ([T,N,B,S] -> [T,N,b,s])
        tr = sub(traj.detach().cpu()).permute(2, 3, 1, 0)
        assert rel_err(tr[:, :, :n_core], o32["xs"]) < TOL
        assert rel_err(tr[:, :, n_core:], o32["prec"]) < TOL
        assert rel_err(sub(xpred.detach().cpu()).permute(2, 3, 1, 0), o32["xp"]) < TOL
        assert rel_err(logp.detach().cpu()[:, rows[:, None], cols[None, :]].permute(1, 2, 0), o32["lpo"], dim=2) < TOL
        got = th.grad.cpu()
        outside = torch.ones(B, S, dtype=torch.bool)
        outside[rows[:, None], cols[None, :]] = False
        assert float(got[:, outside].abs().max()) == 0.0  # a trajectory with no upstream gradient gets none
        _assert_theta_grads(slots, [got[i][rows[:, None], cols[None, :]] for i in range(len(slots))], o32["th"], o64["th"],
                            len(rows), len(cols))
        gw = w.grad.cpu()
        off = 0
        for k, n in zip(KEYS, sizes):
            g64 = o64["w"][off: off + n]
            scale = float(g64.abs().max())
            e32 = float((o32["w"][off: off + n].double() - g64).abs().max()) / scale
            e_hip = float((gw[off: off + n].double() - g64).abs().max()) / scale
            assert e_hip < max(GTOL, 8.0 * e32), (k, e_hip, e32)
            off += n
