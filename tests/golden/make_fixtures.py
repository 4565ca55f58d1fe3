#!/usr/bin/env python3
"""Generate golden fixtures by running the *reference* (microsoft/vi-hds at /root/reference).

This script only runs in the build container, where /root/reference exists.  Nothing in it
travels to the GPU box except its OUTPUT (tests/golden/*.npz): inputs and outputs recorded at
the hot-path boundary.  No reference source is copied anywhere.

Provenance of every fixture (also stored inside each .npz under the key ``provenance``):

* the reference is imported as-is from /root/reference, with throw-away stand-ins for four
  modules that are not installed here and are not on the arithmetic path:
    - ``munch``       : attribute-access dict + recursive ``munchify``
    - ``torchdiffeq`` : ``odeint`` / ``odeint_adjoint`` that raise (so only the reference's own
                        ``modeuler`` / ``modeulerwhile`` integrators can produce fixtures)
    - ``torch.utils.tensorboard`` : no-op ``SummaryWriter``
    - ``seaborn``     : empty module (imported by vihds/plotting.py only)
* ``vihds.datasets.merge_observations`` is replaced by a same-logic version that keeps ragged
  per-file arrays in Python lists, because datasets.py:137 (np.asarray of a ragged list) raises
  on numpy >= 1.24.  Single-file specs do not go through it.
* ``settings.params.solver`` is forced to ``modeuler`` or ``modeulerwhile``: the spec default
  ``midpoint`` lives in torchdiffeq==0.1 which is absent (no network).  midpoint/rk4 parity is
  therefore *unpinned* (see oracle/vihds_oracle.py header and DESIGN.md).

* ``--patched`` (PATCHED_CASES): the relay / degrader / inducer / prpr ``*_precisions`` specs.  The reference raises at
  construction for them: ``OdeFunc.__init__`` takes four arguments and the RHS classes call it with five
  (vihds/ode.py:21 vs models/relay_constant.py:17, degrader_constant.py:17, ...), and the ``*_Precisions`` model classes
  call a method that does not exist (``init_with_params``, relay_constant.py:201, ...).  ``install_reference_patches``
  repairs exactly those two defects IN MEMORY (the extra argument is dropped; ``init_with_params`` = ``OdeModel.__init__``)
  -- no equation, constant or default is touched -- and every fixture of that leg says "MODIFIED REFERENCE" in its
  provenance.  They also hold the first evaluation of the RHS class's own ``forward`` (``rhs_t``, ``rhs_state``,
  ``rhs_out``), taken by a forward hook.

Usage:  python tests/golden/make_fixtures.py            (writes tests/golden/*.npz)
        python tests/golden/make_fixtures.py --patched  (writes only the MODIFIED-REFERENCE fixtures)
"""
import argparse
import json
import os
import sys
import tempfile
import types

import numpy as np

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))


def install_standins():
    import torch  # noqa: F401  (real torch first)

    class Munch(dict):
        def __getattr__(self, k):
            try:
                return self[k]
            except KeyError:
                raise AttributeError(k)

        def __setattr__(self, k, v):
            self[k] = v

    def munchify(x):
        if isinstance(x, dict):
            return Munch((k, munchify(v)) for k, v in x.items())
        if isinstance(x, (list, tuple)):
            return type(x)(munchify(v) for v in x)
        return x

    m = types.ModuleType("munch")
    m.Munch = Munch
    m.munchify = munchify
    sys.modules["munch"] = m

    td = types.ModuleType("torchdiffeq")

    def _absent(*a, **k):
        raise RuntimeError("torchdiffeq==0.1 is not installed in this container")

    td.odeint = _absent
    td.odeint_adjoint = _absent
    sys.modules["torchdiffeq"] = td

    tb = types.ModuleType("torch.utils.tensorboard")

    class SummaryWriter:  # no-op
        def __init__(self, *a, **k):
            pass

        def __getattr__(self, name):
            return lambda *a, **k: None

    tb.SummaryWriter = SummaryWriter
    sys.modules["torch.utils.tensorboard"] = tb
    sys.modules["seaborn"] = types.ModuleType("seaborn")


def patch_merge_observations():
    import vihds.datasets as D

    def merge_observations(times_list, observations_list):
        # same logic as datasets.py:136-145, ragged inputs kept as lists
        n_list = np.array([len(t) for t in times_list])
        loc = int(np.argmin(n_list))
        chosen = times_list[loc]
        out = []
        for t, obs in zip(times_list, observations_list):
            locs = [D.find_nearest(t, ti) for ti in chosen]
            out.append(obs[:, :, locs])
        return chosen, np.concatenate(out)

    D.merge_observations = merge_observations


RHS_RECORD = {}


def install_reference_patches():
    """The two construction defects of SURVEY 2.1, repaired in memory (see the module docstring); plus a forward hook on
    every OdeFunc that records the FIRST evaluation of the model's own RHS ``forward`` after RHS_RECORD was cleared."""
    import vihds.ode as ode

    orig_init = ode.OdeFunc.__init__

    def init(self, config, theta, conditions, dev_1hot, *dropped):  # ode.py:21 takes four; the models pass five
        orig_init(self, config, theta, conditions, dev_1hot)

        def hook(_module, args, out):
            if "rhs_out" not in RHS_RECORD:
                RHS_RECORD["rhs_t"] = to_np(args[0])
                RHS_RECORD["rhs_state"] = to_np(args[1])
                RHS_RECORD["rhs_out"] = to_np(out)

        self.register_forward_hook(hook)

    ode.OdeFunc.__init__ = init
    ode.OdeModel.init_with_params = ode.OdeModel.__init__  # relay_constant.py:201 and its siblings


def to_np(x):
    import torch

    if isinstance(x, torch.Tensor):
        return x.detach().cpu().numpy().copy()
    return np.asarray(x)


def run_case(spec, solver, n_iwae, rows, seed, sample_stride, with_training_step=False, precision_hidden_layers=None,
             params_override=None):
    """Run one reference forward + cost + backward and record everything at the boundary."""
    import torch
    from vihds.config import Config
    from vihds.datasets import build_datasets
    from vihds.parameters import Parameters
    from vihds.run_xval import create_parser
    from vihds.training import Training, log_prob_observations
    from vihds.vae import build_model
    from munch import munchify

    parser = create_parser(True)
    args = parser.parse_args(
        ["--train_samples=%d" % n_iwae, "--test_samples=%d" % n_iwae, "--seed=%d" % seed, "specs/%s.yaml" % spec]
        + (["--precision_hidden_layers=%d" % precision_hidden_layers] if precision_hidden_layers is not None else [])
    )
    settings = Config(args)
    settings.params.solver = solver
    for k, v in (params_override or {}).items():  # e.g. dr_blackbox at other network sizes (models/dr_blackbox.py:61-84)
        settings.params[k] = v
    data = build_datasets(args, settings)
    parameters = Parameters(settings.params)
    model = build_model(args, settings, data, parameters)
    training = Training(args, settings, data, parameters, model)
    model.train()

    # deterministic batch: first `rows` rows of the full training set (already device tensors)
    full = training.train_data
    sel = slice(0, rows)
    batch = munchify(
        {
            "devices": np.asarray(full.devices)[sel],
            "dev_1hot": full.dev_1hot[sel],
            "inputs": full.inputs[sel],
            "observations": full.observations[sel],
            "times": full.times,
        }
    )
    n_batch = len(batch.inputs)

    # ---- BaseVAE.forward, step by step (vae.py:26-36) so intermediates can be recorded
    np.random.seed(seed + 1)
    torch.manual_seed(seed + 1)
    u = model.sample_u(n_batch, n_iwae)
    q = model.encoder(batch)
    theta = q.sample(u, model.device)
    clipped = model.encoder.p.clip(theta, stddevs=4)
    for v in clipped.samples.values():
        if v.requires_grad:
            v.retain_grad()
    for dist in q.distributions.values():
        for pname in ("mu", "log_prec", "prec"):
            t = getattr(dist, pname, None)
            if isinstance(t, torch.Tensor) and t.requires_grad and not t.is_leaf:
                t.retain_grad()
    RHS_RECORD.clear()
    result, cond_theta = model.decoder(clipped, batch, None, None)
    x_states, x_predict, precisions = result
    p = model.encoder.p

    log_p_by_species = log_prob_observations(model, x_predict, batch.observations, precisions, False)
    log_q = q.log_prob(cond_theta)
    log_p = p.log_prob(cond_theta)
    loss = training.cost(batch, result, cond_theta, q, p).elbo
    loss.backward()

    fx = {}
    fx["times"] = to_np(batch.times)
    fx["inputs"] = to_np(batch.inputs)
    fx["dev_1hot"] = to_np(batch.dev_1hot)
    fx["observations"] = to_np(batch.observations)
    fx["u"] = to_np(u)
    names = list(clipped.samples.keys())
    fx["theta_names"] = np.array(names)
    fx["theta_unclipped"] = np.stack([to_np(theta.samples[k]) for k in names])  # [P,B,S]
    fx["theta"] = np.stack([to_np(clipped.samples[k]) for k in names])  # [P,B,S]
    fx["theta_grad"] = np.stack(
        [
            to_np(clipped.samples[k].grad) if clipped.samples[k].grad is not None else np.zeros((n_batch, n_iwae), np.float32)
            for k in names
        ]
    )
    extra = [k for k in ("aR", "aS") if hasattr(cond_theta, k) and k not in names]
    fx["extra_names"] = np.array(extra)
    if extra:
        fx["extra_theta"] = np.stack([to_np(getattr(cond_theta, k)) for k in extra])
    # q / p distribution parameters; kind: 0 Normal, 1 LogNormal, 2 Constant
    kinds, q_mu, q_prec, p_mu, p_prec, q_mu_grad, q_logprec_grad = [], [], [], [], [], [], []
    for k in names:
        dq = q.distributions[k]
        dp = p.distributions[k]
        cls = type(dq).__name__
        kind = {"TfNormal": 0, "TfLogNormal": 1, "TfConstant": 2}[cls]
        kinds.append(kind)
        if kind == 2:
            val = float(to_np(dq.value).reshape(-1)[0])
            q_mu.append(np.full((n_batch,), val, np.float32))
            q_prec.append(np.ones((n_batch,), np.float32))
            p_mu.append(val)
            p_prec.append(1.0)
            q_mu_grad.append(np.zeros((n_batch,), np.float32))
            q_logprec_grad.append(np.zeros((n_batch,), np.float32))
        else:
            mu = to_np(dq.mu).reshape(-1)
            pr = to_np(dq.prec).reshape(-1)
            q_mu.append(np.broadcast_to(mu, (n_batch,)).astype(np.float32))
            q_prec.append(np.broadcast_to(pr, (n_batch,)).astype(np.float32))
            p_mu.append(float(to_np(dp.mu).reshape(-1)[0]))
            p_prec.append(float(to_np(dp.prec).reshape(-1)[0]))
            gm = dq.mu.grad
            gl = dq.log_prec.grad
            q_mu_grad.append(np.broadcast_to(to_np(gm).reshape(-1), (n_batch,)) if gm is not None else np.zeros(n_batch))
            q_logprec_grad.append(
                np.broadcast_to(to_np(gl).reshape(-1), (n_batch,)) if gl is not None else np.zeros(n_batch)
            )
    fx["kind"] = np.array(kinds, np.int32)
    fx["q_mu"] = np.stack(q_mu).astype(np.float32)  # [P,B] (globals broadcast over B)
    fx["q_prec"] = np.stack(q_prec).astype(np.float32)
    fx["q_is_global"] = np.array(
        [0 if (k in q.distributions and getattr(q.distributions[k], "mu", None) is not None
               and to_np(q.distributions[k].mu).size == n_batch and n_batch > 1) else 1 for k in names],
        np.int32,
    )
    fx["p_mu"] = np.array(p_mu, np.float32)
    fx["p_prec"] = np.array(p_prec, np.float32)
    # NOTE: for global (size-1) q tensors the recorded grad is the TOTAL grad (already summed over B)
    fx["q_mu_grad"] = np.stack(q_mu_grad).astype(np.float32)
    fx["q_logprec_grad"] = np.stack(q_logprec_grad).astype(np.float32)

    st = slice(None, None, sample_stride)
    fx["sample_stride"] = np.array(sample_stride)
    fx["x_states"] = to_np(x_states)[:, st]  # [B,S',N,T]
    fx["x_predict"] = to_np(x_predict)[:, st]
    fx["precisions"] = to_np(precisions)[:, st]
    fx["x_states_sum_over_samples"] = to_np(x_states).astype(np.float64).sum(1)
    fx["log_p_by_species"] = to_np(log_p_by_species)
    fx["log_q"] = to_np(log_q)
    fx["log_p"] = to_np(log_p)
    fx["loss"] = to_np(loss)
    # Results.init (utils.py:79-99) through Training.cost(full_output=True), training.py:150-172: the
    # importance-weighted summaries of the evaluation path, on the same forward pass
    with torch.no_grad():
        res = training.cost(batch, result, cond_theta, q, p, full_output=True)
    fx["iw_predict_mu"] = np.asarray(res.iw_predict_mu, np.float32)    # [B,4,T]
    fx["iw_predict_std"] = np.asarray(res.iw_predict_std, np.float32)  # [B,4,T]
    fx["iw_states"] = np.asarray(res.iw_states, np.float32)            # [B,N,T]
    fx["iw_variance"] = np.asarray(res.iw_variance, np.float32)        # [B,4,T]
    fx["results_elbo"] = np.asarray(res.elbo, np.float32)
    # decoder-side nn weights (blackbox / neural precisions) and their grads
    for n_, par in model.decoder.named_parameters():
        fx["decoder_param/" + n_] = to_np(par)
        fx["decoder_grad/" + n_] = to_np(par.grad) if par.grad is not None else np.zeros_like(to_np(par))
    for n_, par in model.encoder.named_parameters():
        fx["encoder_param/" + n_] = to_np(par)
        fx["encoder_grad/" + n_] = to_np(par.grad) if par.grad is not None else np.zeros_like(to_np(par))
    # config scalars the path needs
    cfg = {
        "spec": spec,
        "model": settings.model,
        "solver": solver,
        "n_iwae": n_iwae,
        "rows": rows,
        "seed": seed,
        "device_depth": int(settings.data.device_depth),
        "relevance": {k: [float(x) for x in v] for k, v in settings.data.relevance_vectors.items()},
        "default_devices": dict(settings.data.default_devices),
        "params": {
            k: settings.params[k]
            for k in (
                "n_x", "n_y", "n_z", "n_latent_species", "n_hidden_decoder", "n_hidden_decoder_precisions",
                "init_prec", "init_latent_species",
            )
            if k in settings.params
        },
    }
    fx["config_json"] = np.array(json.dumps(cfg))
    # the experiment definition this case ran with (parsed YAML: data + model + params blocks), so the GPU
    # end-to-end test can build the same model without /root/reference
    import yaml

    with open("specs/%s.yaml" % spec) as fh:
        spec_dict = yaml.safe_load(fh)
    spec_dict["params"].update(params_override or {})
    fx["spec_json"] = np.array(json.dumps(spec_dict))
    fx["devices"] = np.asarray(batch.devices)
    fx.update(RHS_RECORD)  # only with install_reference_patches(): first evaluation of the RHS class's forward
    return fx


def run_training_trace(spec, solver, n_iwae, epochs, seed, params_override=None):
    """Run the reference's own Training.run() (run_xval.run_on_split) for a few epochs and record the loss of
    every training step plus the evaluation ELBOs, together with the processed dataset it ran on, so the GPU
    build can be driven through the identical sequence (same seeds => same shuffles, u draws and conditioner
    weights) on a box that has neither the reference nor its CSV files."""
    import torch
    from vihds.config import Config
    from vihds.datasets import build_datasets
    from vihds.parameters import Parameters
    from vihds.run_xval import create_parser
    from vihds.training import Training
    from vihds.vae import build_model

    parser = create_parser(True)
    args = parser.parse_args(["--train_samples=%d" % n_iwae, "--test_samples=%d" % n_iwae, "--seed=%d" % seed,
                              "--epochs=%d" % epochs, "--test_epoch=%d" % epochs, "--plot_epoch=0",
                              "specs/%s.yaml" % spec])
    args.heldout = None
    settings = Config(args)
    settings.params.solver = solver
    for k, v in (params_override or {}).items():  # e.g. dr_blackbox at other network sizes (models/dr_blackbox.py:61-84)
        settings.params[k] = v
    data = build_datasets(args, settings)
    parameters = Parameters(settings.params)
    model = build_model(args, settings, data, parameters)
    training = Training(args, settings, data, parameters, model)
    losses = []
    orig_cost = training.cost

    def recording_cost(*a, **k):
        out = orig_cost(*a, **k)
        if not k.get("full_output", False):
            losses.append(float(out.elbo))
        return out

    training.cost = recording_cost
    evals = []
    orig_eval = training._evaluate_elbo_and_plot

    def recording_eval(*a, **k):
        out = orig_eval(*a, **k)
        evals.append(float(out.elbo))
        return out

    training._evaluate_elbo_and_plot = recording_eval
    os.makedirs(".vihds_cache_fixture", exist_ok=True)
    cwd = os.getcwd()
    os.chdir(tempfile.mkdtemp())  # Results.dump writes .vihds_cache relative to cwd
    try:
        training.run()
    finally:
        os.chdir(cwd)
    ds = data.train.dataset
    import yaml

    with open("specs/%s.yaml" % spec) as fh:
        spec_dict = yaml.safe_load(fh)
    return {
        "step_losses": np.array(losses, np.float64), "valid_elbo": np.array(evals, np.float64),
        "times": to_np(ds.times), "devices": np.asarray(ds.devices), "dev_1hot": to_np(ds.dev_1hot),
        "inputs": to_np(ds.inputs), "observations": to_np(ds.observations),
        "train_ids": np.asarray(data.train.indices), "valid_ids": np.asarray(data.test.indices),
        "spec_json": np.array(json.dumps(spec_dict)),
        "config_json": np.array(json.dumps({"spec": spec, "model": settings.model, "solver": solver, "n_iwae": n_iwae,
                                            "epochs": epochs, "seed": seed, "folds": 4, "split": 1})),
    }


PROVENANCE = (
    "generated by tests/golden/make_fixtures.py from /root/reference (microsoft/vi-hds) imported with "
    "stand-ins for munch, torchdiffeq(raises), torch.utils.tensorboard(no-op), seaborn(empty); "
    "datasets.merge_observations replaced by a same-logic ragged-list version (numpy>=1.24); "
    "params.solver forced to the value recorded in config_json; rows = first `rows` rows of the "
    "training split (folds=4, split=1, seed as recorded); np.random.seed(seed+1) and "
    "torch.manual_seed(seed+1) set immediately before sample_u. "
)

CASES = [
    # name,                      spec,                    solver,          S,   rows, stride
    ("dr_constant_one_modeuler", "dr_constant_one", "modeuler", 1, 36, 1),
    ("dr_constant_one_s5_modeulerwhile", "dr_constant_one", "modeulerwhile", 5, 8, 1),
    ("dr_constant_icml_tiny_modeuler", "dr_constant_icml", "modeuler", 8, 4, 1),
    ("dr_constant_icml_tiny_modeulerwhile", "dr_constant_icml", "modeulerwhile", 8, 4, 1),
    ("dr_constant_icml_full_modeuler", "dr_constant_icml", "modeuler", 200, 36, 25),
    ("dr_constant_v2_tiny_modeuler", "dr_constant_v2", "modeuler", 8, 4, 1),
    ("auto_constant_tiny_modeuler", "auto_constant", "modeuler", 8, 4, 1),
    ("dr_constant_precisions_tiny_modeuler", "dr_constant_precisions", "modeuler", 8, 4, 1),
    ("auto_constant_precisions_tiny_modeuler", "auto_constant_precisions", "modeuler", 8, 4, 1),
    ("dr_blackbox_icml_tiny_modeuler", "dr_blackbox_icml", "modeuler", 8, 4, 1),
    # BASELINE config 4's own shape (36 rows x 200 samples) from the reference itself (round 6): the matrix-core kernels at the
    # size the bench times them, against the reference's modified Euler (midpoint, the spec's solver, is torchdiffeq's)
    ("dr_blackbox_icml_full_modeuler", "dr_blackbox_icml", "modeuler", 200, 36, 25),
    # BASELINE config 3's training shape (36 rows x 1 000 samples) from the reference itself (round 6), LIGHT: the four
    # [.., 36, 1000, ..] inputs (u, theta, theta_unclipped, theta_grad: 20 MB) are dropped -- u is np.random.seed(seed + 1);
    # np.random.randn(36, 1000, 35).astype(float32), numpy's legacy stream (recorded as `u_seed`, checked on the 36 x 200
    # fixture), theta follows from u and the q / p tables -- the outputs stay: loss, log-likelihoods, log q, log p, q gradients,
    # every 125th sample's trajectory
    ("dr_constant_icml_s1000_light_modeuler", "dr_constant_icml", "modeuler", 1000, 36, 125, None, None, True),
    ("prpr_constant_tiny_modeuler", "prpr_constant", "modeuler", 8, 4, 1),
    # NeuralPrecisions with a hidden layer (reference precisions.py:63-74) through the CLI flag --precision_hidden_layers
    ("dr_constant_precisions_hidden20_tiny_modeuler", "dr_constant_precisions", "modeuler", 8, 4, 1, 20),
    # dr_blackbox at network sizes other than the ICML spec's (models/dr_blackbox.py:61-84 reads them from the YAML's
    # params block; overridden here after Config() parsed specs/dr_blackbox_icml.yaml)
    ("dr_blackbox_sized_tiny_modeuler", "dr_blackbox_icml", "modeuler", 8, 4, 1, None,
     {"n_z": 4, "n_x": 3, "n_y": 1, "n_latent_species": 3, "n_hidden_decoder": 12, "n_hidden_decoder_precisions": 6}),
]


PATCHED_PROVENANCE = (
    "MODIFIED REFERENCE: vihds.ode.OdeFunc.__init__ wrapped to drop the fifth positional argument the model RHS classes "
    "pass (ode.py:21 vs relay_constant.py:17 / degrader_constant.py:17 / inducer_constant.py / prpr_constant.py), and "
    "OdeModel.init_with_params bound to OdeModel.__init__ (called by the *_Precisions classes, e.g. relay_constant.py:201); "
    "nothing else changed. rhs_t/rhs_state/rhs_out: first evaluation of the RHS class's forward (forward hook). "
)

# the only relay / degrader / inducer specs the reference ships are the *_precisions ones
PATCHED_CASES = [
    ("relay_constant_precisions_tiny_modeuler", "relay_constant_precisions", "modeuler", 8, 4, 1),
    ("relay_constant_precisions_tiny_modeulerwhile", "relay_constant_precisions", "modeulerwhile", 5, 3, 1),
    # BASELINE config 5's own shape (36 rows x 200 samples, T = 99) from the MODIFIED reference (round 6)
    ("relay_constant_precisions_full_modeuler", "relay_constant_precisions", "modeuler", 200, 36, 25),
    ("degrader_constant_precisions_tiny_modeuler", "degrader_constant_precisions", "modeuler", 8, 4, 1),
    ("inducer_constant_precisions_tiny_modeuler", "inducer_constant_precisions", "modeuler", 8, 4, 1),
    ("prpr_constant_precisions_tiny_modeuler", "prpr_constant_precisions", "modeuler", 8, 4, 1),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default=None)
    ap.add_argument("--patched", action="store_true", help="the MODIFIED-REFERENCE leg (PATCHED_CASES) instead of CASES")
    ap.add_argument("--out", default=HERE, help="directory the .npz files are written to (default: next to this script; "
                    "a scratch directory lets a regeneration be compared with the committed files)")
    a = ap.parse_args()
    out_dir = os.path.abspath(a.out)
    os.makedirs(out_dir, exist_ok=True)
    install_standins()
    sys.path.insert(0, REF)
    os.chdir(REF)  # reference resolves specs/ and data/ relative to cwd
    os.environ["INFERENCE_RESULTS_DIR"] = tempfile.mkdtemp()
    patch_merge_observations()
    import torch

    if a.patched:
        install_reference_patches()
        for name, spec, solver, S, rows, stride in PATCHED_CASES:
            if a.only and a.only not in name:
                continue
            fx = run_case(spec, solver, S, rows, 0, stride)
            fx["provenance"] = np.array(PATCHED_PROVENANCE + PROVENANCE + "torch %s numpy %s python %s"
                                        % (torch.__version__, np.__version__, sys.version.split()[0]))
            out = os.path.join(out_dir, name + ".npz")
            np.savez_compressed(out, **fx)
            print("wrote %s  loss=%s  (%.1f kB)" % (out, fx["loss"], os.path.getsize(out) / 1e3))
        return
    if not a.only or "trace" in a.only:
        for name, spec, solver, S, epochs in [("trace_dr_constant_icml_modeuler", "dr_constant_icml", "modeuler", 20, 4),
                                              ("trace_auto_constant_modeuler", "auto_constant", "modeuler", 20, 6),
                                              # the headline's sample count at the spec's own learning rate, long enough to show
                                              # what the objective does past the first epochs (round 5: 15 epochs = 105 steps)
                                              ("trace_dr_constant_icml_s200_modeuler", "dr_constant_icml", "modeuler", 200, 15)]:
            if a.only and a.only not in name:
                continue
            fx = run_training_trace(spec, solver, S, epochs, 0)
            fx["provenance"] = np.array(PROVENANCE + " [training trace: run_on_split semantics, losses per step] torch %s numpy %s"
                                        % (torch.__version__, np.__version__))
            out = os.path.join(out_dir, name + ".npz")
            np.savez_compressed(out, **fx)
            print("wrote %s  steps=%d first=%.4f last=%.4f valid=%s" % (out, len(fx["step_losses"]), fx["step_losses"][0],
                                                                        fx["step_losses"][-1], fx["valid_elbo"]))
    for name, spec, solver, S, rows, stride, *more in CASES:
        if a.only and a.only not in name:
            continue
        try:
            fx = run_case(spec, solver, S, rows, 0, stride, precision_hidden_layers=more[0] if more else None,
                          params_override=more[1] if len(more) > 1 else None)
        except Exception as e:  # a reference defect is recorded, not hidden
            print("FAILED %s: %s: %s" % (name, type(e).__name__, e))
            continue
        if len(more) > 2 and more[2]:  # LIGHT: the sample-sized inputs are reproducible from the seed, see CASES
            u = fx["u"]
            np.random.seed(int(json.loads(str(fx["config_json"]))["seed"]) + 1)
            assert np.array_equal(np.random.randn(*u.shape).astype(np.float32), u), "u is not the seeded legacy stream"
            fx["u_seed"] = np.array(int(json.loads(str(fx["config_json"]))["seed"]) + 1)
            fx["u_shape"] = np.array(u.shape)
            for k in ("u", "theta", "theta_unclipped", "theta_grad"):
                fx.pop(k, None)
        fx["provenance"] = np.array(
            PROVENANCE + "torch %s numpy %s python %s" % (torch.__version__, np.__version__, sys.version.split()[0])
        )
        out = os.path.join(out_dir, name + ".npz")
        np.savez_compressed(out, **fx)
        print("wrote %s  loss=%s  (%.1f kB)" % (out, fx["loss"], os.path.getsize(out) / 1e3))


if __name__ == "__main__":
    main()
