"""GPU parity of ops.GeneralTail (vihds_step_tail, ABI 13): the training step of ANY model with everything behind the
ODE forward -- IWAE loss, ODE adjoint, decoder-network weight gradients, theta / encoder adjoints, Adam on encoder AND
decoder-side tensors -- outside autograd, against (a) the reference's own gradients in the golden fixtures and (b) the
autograd path of this package (fused_step_tail: false) from the same state and draws.  Reference call sequence:
vihds/training.py:324-340 (`_run_batch`), which is model-agnostic."""
import numpy as np
import pytest
import torch

from fixture_util import PATCHED_FIXTURES, Fixture, rel_err

pytestmark = pytest.mark.gpu

# every model family the general tail serves: white-box without weights (auto / prpr), white-box + neural precisions through
# the thread-per-trajectory kernels and the Gram contraction (dr_constant_precisions, hidden layer included), the lane-split
# kernels with per-block weight partials (relay / degrader / prpr / auto _precisions), dr_blackbox (MFMA kernels, on-chip
# Gram tiles, the offset layer) at both compiled sizes
FIXTURE_CASES = ["auto_constant_tiny_modeuler", "prpr_constant_tiny_modeuler", "dr_constant_precisions_tiny_modeuler",
                 "dr_constant_precisions_hidden20_tiny_modeuler", "auto_constant_precisions_tiny_modeuler",
                 "dr_blackbox_icml_tiny_modeuler", "dr_blackbox_sized_tiny_modeuler"] + PATCHED_FIXTURES


def _build(fx, tail, graph=False, **over):
    import e2e_util as E
    from vihds.training import Training
    from vihds.vae import build_model

    args, settings, data, parameters = E.build_from_fixture(fx, gpu=0, fused_step_tail=tail, hip_graph=graph, **over)
    model = build_model(args, settings, data, parameters)
    training = Training(args, settings, data, parameters, model)
    model.train()
    return args, settings, model, training, E.batch_from_fixture(fx, settings.device)


def _one_step(fx, tail, n_steps=1, graph=False, **over):
    args, settings, model, training, batch = _build(fx, tail, graph, **over)
    np.random.seed(fx.cfg["seed"] + 1)
    torch.manual_seed(fx.cfg["seed"] + 1)
    losses = []
    from vihds import ops

    launched = []
    for k in range(n_steps):
        if graph:
            loss = training.graph_step(batch)
        else:
            rec = ops.LaunchRecorder()
            ops.TIMER = rec
            try:
                loss = training.step(batch, zero_grad=False)
            finally:
                ops.TIMER = None
            launched = list(rec.calls)
        losses.append(float(loss))
        if k < n_steps - 1 and not graph:
            training.optimizer.zero_grad(set_to_none=True)
    torch.cuda.synchronize()
    grads = {k: v.grad.detach().clone() for k, v in model.named_parameters() if v.grad is not None}
    params = {k: v.detach().clone() for k, v in model.named_parameters()}
    return losses, grads, params, training, launched, model


@pytest.mark.parametrize("name", FIXTURE_CASES)
def test_general_tail_gradients_match_reference(name):
    """One Training.step through the general tail on the fixture's batch with the reference's RNG streams: -ELBO and the
    gradient of EVERY encoder and decoder parameter (the tail leaves them in .grad) against the reference's autograd."""
    from test_e2e_gpu import _ref_encoder_grads

    fx = Fixture(name)
    losses, grads, _params, training, launched, model = _one_step(fx, True)
    assert training._gtail_ok is True, "the general tail did not take this model"
    assert "step_tail" in launched and "ode_bwd" in launched, launched
    assert rel_err(torch.tensor(losses[0]), fx.t("loss")) < 1e-4
    ref = _ref_encoder_grads(fx, model.encoder)
    for k, g in ref.items():
        assert rel_err(grads["encoder." + k].cpu(), g) < 1e-3, k
    dref = {k[len("decoder_grad/"):]: fx.t(k) for k in fx.z.files if k.startswith("decoder_grad/")}
    n_dec = 0
    for k, v in model.decoder.named_parameters():
        assert rel_err(grads["decoder." + k].cpu(), dref[k]) < 1e-3, k
        n_dec += 1
    assert n_dec == len(dref)


@pytest.mark.parametrize("graph", [False, True])
@pytest.mark.parametrize("name", ["relay_constant_precisions_tiny_modeuler", "dr_blackbox_icml_tiny_modeuler",
                                  "dr_constant_precisions_tiny_modeuler", "prpr_constant_tiny_modeuler"])
def test_general_tail_matches_autograd_path(name, graph):
    """The same steps with fused_step_tail on and off, same seeds: loss of every step, last gradients, every parameter (Adam
    included: encoder, decoder networks, offset layer) after 1 and after 4 steps, and the step counter -- eagerly and replayed
    from the step's hipGraph."""
    fx = Fixture(name)
    # (parameters after ONE Adam step: the update is lr * g / (|g| + eps)-like, so gradient elements of size ~eps = 1e-8 -- far
    # below the max-norm the gradient check uses -- move their parameter by up to a few 1e-5 of the tensor's scale)
    for n_steps, tol_p in ((1, 5e-5), (4, 2e-4)):
        ref = _one_step(fx, False, n_steps, graph)
        got = _one_step(fx, True, n_steps, graph)
        assert got[3]._gtail_ok is True and ref[3]._gtail_ok is None
        for a, b in zip(ref[0], got[0]):
            assert abs(a - b) <= 2e-5 * max(1.0, abs(a)), (ref[0], got[0])
        assert set(ref[1]) == set(got[1]), set(ref[1]) ^ set(got[1])
        for k, g in ref[1].items():
            assert rel_err(got[1][k], g) < (2e-5 if n_steps == 1 else 5e-4), (k, n_steps)
        for k, v in ref[2].items():
            assert rel_err(got[2][k], v) < tol_p, (k, n_steps)
        assert ref[3].optimizer.step_count() == got[3].optimizer.step_count() == n_steps


def test_general_tail_at_config5_shape_trains_and_counts_launches():
    """BASELINE config 5's shape (relay_constant_precisions, B=36, S=200, T=99, midpoint) through the general tail: the
    step is encoder + theta + ODE forward + IWAE + ODE adjoint + the tail's two launches, the objective stays finite over
    a few steps and the decoder's 112 network weights move."""
    from vihds import ops, synthetic

    args, settings, data, parameters, model, training = synthetic.build(
        "relay_constant_precisions", 36, 200, solver="midpoint", device="cuda:0", seed=1, u_rng="kernel",
        conditioner_rng="kernel", hip_graph=False, nan_check_every=0, learning_rate=0.001)
    model.train()
    batch = training.train_data
    w0 = model.decoder.ode_model.precisions.flat_weights().detach().clone()
    rec = ops.LaunchRecorder()
    ops.TIMER = rec
    try:
        losses = [float(training.step(batch)) for _ in range(3)]
    finally:
        ops.TIMER = None
    assert training._gtail_ok is True
    assert all(np.isfinite(losses)), losses
    assert set(rec.calls) >= {"ode_fwd", "ode_bwd", "step_tail"}
    w1 = model.decoder.ode_model.precisions.flat_weights().detach()
    assert (w1 != w0).any() and torch.isfinite(w1).all()
    assert training.optimizer.step_count() == 3


@pytest.mark.parametrize("name", ["relay_constant_precisions_tiny_modeuler", "dr_blackbox_icml_tiny_modeuler",
                                  "dr_constant_precisions_tiny_modeuler", "auto_constant_tiny_modeuler",
                                  "dr_blackbox_sized_tiny_modeuler"])
def test_importance_weights_formed_inside_the_adjoint_launch(name):
    """vihds_ode_bwd_elbo (params.inkernel_iwae, default): every wavefront of the adjoint forms the row-wise logsumexp of its
    trajectories' data rows itself -- lane-split, thread-per-trajectory and both dr_blackbox kernel families -- against the
    same step with the IWAE launch (vihds_iwae_loss_fwd) in front of vihds_ode_bwd: loss, every gradient."""
    fx = Fixture(name)
    ref = _one_step(fx, True, inkernel_iwae=False)
    got = _one_step(fx, True, inkernel_iwae=True)
    assert ref[3]._gtail.inkernel_iwae is False and got[3]._gtail.inkernel_iwae is True
    assert abs(ref[0][0] - got[0][0]) <= 1e-6 * max(1.0, abs(ref[0][0]))
    assert set(ref[1]) == set(got[1])
    for k, g in ref[1].items():
        assert rel_err(got[1][k], g) < 1e-5, k


@pytest.mark.parametrize("rng", ["numpy", "kernel"])
@pytest.mark.parametrize("name", ["relay_constant_precisions_tiny_modeuler", "dr_blackbox_icml_tiny_modeuler",
                                  "prpr_constant_tiny_modeuler", "degrader_constant_precisions_tiny_modeuler"])
def test_sampling_stage_inside_the_forward_launch(name, rng):
    """vihds_theta_ode_fwd (params.fused_theta_ode, default): theta = clip(sample(q, u)), log q, log p -- and dr_blackbox's
    condition_theta -- as a prologue of the ODE forward launch, against the separate vihds_theta_fwd [+ vihds_offset_rows_fwd] +
    vihds_ode_fwd launches: three steps each (the fused forward serves from the second step on), same draws -- the host's
    numpy stream, or the in-kernel generator whose step the tail now advances: losses, gradients, parameters."""
    fx = Fixture(name)
    over = {} if rng == "numpy" else {"u_rng": "kernel", "conditioner_rng": "kernel"}
    ref = _one_step(fx, True, 3, fused_theta_ode=False, **over)
    got = _one_step(fx, True, 3, fused_theta_ode=True, **over)
    assert "ode_fwd" in got[4] and got[3]._gtail_ok is True
    node_names = []
    for a, b in zip(ref[0], got[0]):
        assert abs(a - b) <= 1e-5 * max(1.0, abs(a)), (ref[0], got[0])
    if rng == "kernel":  # three different draws: the generator's step moved every step
        assert len({round(v, 3) for v in got[0]}) == 3, got[0]
    for k, g in ref[1].items():
        assert rel_err(got[1][k], g) < 2e-4, k
    for k, v in ref[2].items():
        assert rel_err(got[2][k], v) < 1e-4, k


def test_fused_forward_node_differentiates_through_autograd_too():
    """ops.ThetaOdeFused's own backward (a caller that runs loss.backward() on the fused forward's outputs): replays the unfused
    ops on the saved draws -- gradients of q's tables and the network weights against the unfused autograd path."""
    from vihds import ops

    fx = Fixture("relay_constant_precisions_tiny_modeuler")
    args, settings, model, training, batch = _build(fx, False)
    outs = []
    for fused in (False, True):
        np.random.seed(3)
        torch.manual_seed(3)
        model.zero_grad(set_to_none=True)
        model._fuse_theta_ode = fused
        try:
            results, theta, q, p = model(batch, args.train_samples)
        finally:
            model._fuse_theta_ode = False
        node = results.solution.logp_buffer.grad_fn
        assert (type(node).__name__ == "ThetaOdeFusedBackward") == fused
        loss = training.cost(batch, results, theta, q, p).elbo
        loss.backward()
        outs.append((float(loss), {k: v.grad.detach().clone() for k, v in model.named_parameters() if v.grad is not None}))
    assert abs(outs[0][0] - outs[1][0]) <= 1e-6 * abs(outs[0][0])
    assert set(outs[0][1]) == set(outs[1][1])
    for k, g in outs[0][1].items():
        assert rel_err(outs[1][1][k], g) < 1e-5, k
