"""Synthetic workloads of the shapes BASELINE.json names: no CSVs and no network are needed on
the GPU box.  Each workload is an experiment definition in the same dict form a spec YAML parses to, plus a
seeded generator of plate-reader-like observations, so the ordinary host path (Config -> Parameters -> Encoder
-> Decoder -> Training) runs unchanged on it."""
import argparse

import numpy as np
import torch
from torch.utils.data import Dataset, Subset

from vihds.datasets import TimeSeriesDatasetPair, get_cassettes


def _ln(mu, sigma=None, prec=None):
    d = {"distribution": "LogNormal", "mu": mu}
    if sigma is not None:
        d["sigma"] = sigma
    if prec is not None:
        d["prec"] = prec
    return d


def dr_constant_icml_spec(solver="rk4"):
    """The double-receiver ICML experiment: model, device groups and priors as in the reference's
    specs/dr_constant_icml.yaml:6-80 (6 devices, device_depth 7, P = 35 parameters)."""
    shared = {"data_prec": _ln(8.0, 2.0), "auto_prec": _ln(-5.0, 2.0), "dfp_prec": _ln(-2.0, 1.5)}
    ref = lambda k: {"distribution": k}  # noqa: E731
    glob = {"prec_x": ref("data_prec"), "prec_rfp": ref("data_prec"), "prec_yfp": ref("data_prec"),
            "prec_cfp": ref("data_prec"), "e76": _ln(-3.0, 1.0), "e81": _ln(-3.0, 1.0), "KGR_76": _ln(2.0, 3.0),
            "KGR_81": _ln(-2.0, 3.0), "KGS_76": _ln(-2.0, 3.0), "KGS_81": _ln(2.0, 3.0), "KR6": _ln(-6.0, 3.0),
            "KR12": _ln(-12.0, 3.0), "KS6": _ln(-12.0, 3.0), "KS12": _ln(-6.0, 3.0), "nR": _ln(0.0, 0.25),
            "nS": _ln(0.0, 0.25), "aYFP": _ln(0.0, 2.0), "aCFP": _ln(0.0, 2.0), "dR": _ln(-2.0, 1.0),
            "dS": _ln(-2.0, 1.0), "drfp": ref("dfp_prec"), "dyfp": ref("dfp_prec"), "dcfp": ref("dfp_prec"),
            "a530": ref("auto_prec"), "a480": ref("auto_prec")}
    local = {"conditioning": {"devices": True, "treatments": False}, "r": _ln(0.0, 0.25), "K": _ln(1.0, prec=2.0),
             "tlag": _ln(0.0, prec=2.0), "rc": _ln(0.0, 2.0)}
    return {
        "data": {
            "devices": ["Pcat_Y81C76", "RS100S32_Y81C76", "RS100S34_Y81C76", "R33S32_Y81C76", "R33S34_Y81C76",
                        "R33S175_Y81C76"],
            "groups": {"aR": [0, 1, 1, 2, 2, 2], "aS": [0, 1, 2, 1, 2, 3]},
            "default_devices": {"aR": 0, "aS": 0},
            "files": [], "signals": ["OD", "mRFP1", "EYFP", "ECFP"], "conditions": ["C6", "C12"],
            "separate_conditions": True,
        },
        "model": "dr_constant",
        "params": {
            "learning_boundaries": [250, 1000], "learning_rate": 0.01, "learning_gamma": 0.2, "solver": solver,
            "constant": {"init_x": 0.002, "init_rfp": 0.0, "init_yfp": 0.0, "init_cfp": 0.0, "init_luxR": 0.0,
                         "init_lasR": 0.0},
            "shared": shared,
            "global_conditioned": {"conditioning": {"devices": True, "treatments": False}},
            "global": glob, "local": local,
        },
    }


def dr_blackbox_icml_spec(solver="midpoint"):
    """The black-box ICML experiment (BASELINE config 4): same plate and devices, MLP right-hand side 27->25->6 and
    MLP precisions 28->20->4, standard-normal z / y / x as in the reference's specs/dr_blackbox_icml.yaml."""
    sn = {"distribution": "standard_normal"}
    base = dr_constant_icml_spec(solver)
    return {
        "data": {k: v for k, v in base["data"].items() if k != "default_devices"},
        "model": "dr_blackbox",
        "params": {
            "n_z": 5, "n_y": 2, "n_x": 5, "n_latent_species": 2, "n_hidden_decoder": 25,
            "n_hidden_decoder_precisions": 20, "lambda_l2": 0.1, "lambda_l2_hidden": 0.1,
            "learning_boundaries": [250], "learning_rate": 0.005, "learning_gamma": 0.2, "solver": solver,
            "constant": {"init_x": 0.002, "init_rfp": 0.0, "init_yfp": 0.0, "init_cfp": 0.0},
            "shared": {"init_data_log_precision": {"distribution": "Normal", "mu": 5000.0, "sigma": 200.0},
                       "standard_normal": {"distribution": "Normal", "mu": 0.0, "sigma": 1.0}},
            "global": {"x%d" % k: dict(sn) for k in range(1, 6)},
            "global_conditioned": dict({"conditioning": {"devices": True, "treatments": False}},
                                       **{"y%d" % k: dict(sn) for k in (1, 2)}),
            "local": dict({"conditioning": {"devices": True, "treatments": False}},
                          **{"z%d" % k: dict(sn) for k in range(1, 6)}),
        },
    }


def relay_constant_precisions_spec(solver="midpoint"):
    """The relay experiment (BASELINE config 5): two relay devices, 12 species + 4 neural precision states, P = 45
    parameters, priors as in the reference's specs/relay_constant_precisions.yaml:6-97 (the reference's model classes for
    this spec raise at construction -- relay_constant.py:17,201 -- so this workload runs on this package only)."""
    ref = lambda k: {"distribution": k}  # noqa: E731
    glob = {"e76": _ln(-3.0, 1.0), "e81": _ln(-3.0, 1.0), "KGR_76": _ln(2.0, 3.0), "KGR_81": _ln(-2.0, 3.0),
            "KGS_76": _ln(-2.0, 3.0), "KGS_81": _ln(2.0, 3.0), "KR6": _ln(-6.0, 3.0), "KR12": _ln(-12.0, 3.0),
            "KS6": _ln(-12.0, 3.0), "KS12": _ln(-6.0, 3.0), "KC6": _ln(-6.0, 3.0), "KC12": _ln(-6.0, 3.0),
            "Klux": _ln(-2.0, 3.0), "Klas": _ln(-2.0, 3.0), "nR": _ln(0.0, 0.25), "nS": _ln(0.0, 0.25),
            "dR": _ln(-2.0, 1.0), "dS": _ln(-2.0, 1.0), "dluxI": _ln(-2.0, 1.0), "dlasI": _ln(-2.0, 1.0),
            "aYFP": _ln(0.0, 2.0), "aCFP": _ln(0.0, 2.0), "aR": _ln(0.0, 0.25), "aS": _ln(0.0, 0.25),
            "drfp": ref("dfp_prec"), "dyfp": ref("dfp_prec"), "dcfp": ref("dfp_prec"), "a530": ref("auto_prec"),
            "a480": ref("auto_prec"), "init_prec_x": ref("init_prec"), "init_prec_rfp": ref("init_prec"),
            "init_prec_yfp": ref("init_prec"), "init_prec_cfp": ref("init_prec")}
    local = {"conditioning": {"devices": True, "treatments": False}, "r": _ln(0.0, 0.25), "K": _ln(0.0, prec=2.0),
             "tlag": _ln(0.0, prec=2.0), "rc": _ln(0.0, 2.0)}
    return {
        "data": {"devices": ["R33S175DR_P76LasI", "R33S175DR_P81LuxI"], "files": [],
                 "signals": ["OD", "mRFP1", "EYFP", "ECFP"], "conditions": ["C6", "C12"], "separate_conditions": True},
        "model": "relay_constant_precisions",
        "params": {
            "learning_boundaries": [250, 500], "learning_rate": 0.01, "learning_gamma": 0.2, "solver": solver,
            "n_hidden_decoder_precisions": 0,
            "constant": {"init_x": 0.002, "init_rfp": 0.0, "init_yfp": 0.0, "init_cfp": 0.0, "init_luxR": 0.0,
                         "init_lasR": 0.0, "init_lasI": 0.0, "init_luxI": 0.0},
            "shared": {"auto_prec": _ln(-5.0, 2.0), "dfp_prec": _ln(-2.0, 1.5), "init_prec": _ln(6.0, 2.0)},
            "global_conditioned": {"conditioning": {"devices": True, "treatments": False}},
            "global": glob, "local": local,
        },
    }


MODEL_SIMULATED = ("dr_constant_icml", "relay_constant_precisions")
WORKLOADS = {"dr_constant_icml": (dr_constant_icml_spec, 86), "dr_blackbox_icml": (dr_blackbox_icml_spec, 86),
             "relay_constant_precisions": (relay_constant_precisions_spec, 99)}


class SyntheticPlateDataset(Dataset):
    """Seeded plate-reader-like rows: logistic OD growth and product signals scaled to [0,1] with the background
    subtracted, treatments log1p'ed, device one-hot blocks -- the tensors vihds.datasets produces from CSVs."""

    def __init__(self, data_settings, n_rows, n_times, seed=0, dt=0.1933):
        rng = np.random.default_rng(seed)
        self.times = torch.tensor((np.arange(n_times) * dt).astype(np.float32))
        self.n_times, self.n_species = n_times, 4
        n_dev = len(data_settings.devices)
        self.devices = (np.arange(n_rows) % n_dev).astype(int)
        self.dev_1hot = torch.tensor(get_cassettes(self.devices, data_settings))
        levels = np.array([0.0, 5.0, 25.0, 250.0, 1000.0, 5000.0, 25000.0], dtype=np.float32)
        n_cond = len(data_settings.conditions)
        self.inputs = torch.tensor(np.log1p(levels[rng.integers(0, len(levels), size=(n_rows, n_cond))]))
        t = self.times.numpy()[None, :]
        r = rng.uniform(0.8, 1.4, (n_rows, 1))
        lag = rng.uniform(1.0, 3.0, (n_rows, 1))
        od = 1.0 / (1.0 + np.exp(-r * (t - lag - 3.0)))
        obs = np.stack([od] + [od * rng.uniform(0.2, 1.0, (n_rows, 1)) * (1 - np.exp(-t / rng.uniform(2, 6, (n_rows, 1))))
                               for _ in range(3)], axis=1)
        obs = obs + rng.normal(0, 0.005, obs.shape)
        obs = obs / obs.max(axis=(0, 2), keepdims=True)
        obs = obs - obs.min(axis=2, keepdims=True)
        self.observations = torch.tensor(obs.astype(np.float32))

    def __len__(self):
        return len(self.devices)

    def __getitem__(self, idx):
        if torch.is_tensor(idx):
            idx = idx.tolist()
        return {"devices": self.devices[idx], "dev_1hot": self.dev_1hot[idx], "inputs": self.inputs[idx],
                "observations": self.observations[idx]}


def simulate_observations(settings, parameters, dataset, device, seed=0, noise=0.01):
    """Replace the dataset's observations by signals simulated from the model itself (one trajectory per row,
    theta drawn around the prior medians), plus noise -- so the synthetic workload is a well-posed inference
    problem like the plate-reader data, not just tensors of the right shape.  Runs the HIP forward kernel once."""
    import models
    from vihds import hip
    from vihds.distributions import DotOperatorSamples

    ode = models.LOOKUP[settings.model](settings).to(device)  # (neural precisions: their weights feed the kernel from HBM)
    g = torch.Generator().manual_seed(seed)
    n = len(dataset)
    th = DotOperatorSamples()
    for d in parameters.ordered():
        if d.kind == 2:
            v = torch.full((n, 1), d.value)
        else:
            mu, sigma, _ = d.prior_mu_sigma_prec()
            spread = 0.3 if d.level == "local" else 0.0
            z = float(mu) + spread * float(sigma) * torch.randn(n, 1, generator=g)
            v = z.exp() if d.kind == 1 else z
        th.add(d.name, v.to(device))
    dev1 = dataset.dev_1hot.to(device)
    for name in ode.kernel_slots():
        if name not in th.samples:  # aR / aS of the double-receiver family
            setattr(th, name, torch.full((n, 1), 1.5, device=device))
    sol = ode.solve(settings, dataset.times, th, dataset.inputs.to(device), dev1)
    xp = sol.x_predict[:, 0].detach().cpu()  # [n,4,T]
    scale = xp.amax(dim=(0, 2), keepdim=True).clamp_min(1e-6)
    obs = xp / scale + noise * torch.randn(xp.shape, generator=g)
    obs = obs - obs.amin(dim=2, keepdim=True)
    dataset.observations = obs.float().contiguous()  # (x_predict is a permuted view: keep the plate layout [n,4,T])


def make_args(n_iwae, seed=0, gpu=None):
    """An argparse namespace with the reference CLI's fields (vihds/run_xval.py:17-57)."""
    return argparse.Namespace(yaml=None, experiment="synthetic", seed=seed, epochs=1, test_epoch=1, plot_epoch=0,
                              train_samples=n_iwae, test_samples=n_iwae, dreg=True, precision_hidden_layers=None,
                              verbose=False, gpu=gpu, heldout=None, split=1, figures=False, folds=4)


def build(workload, n_rows, n_iwae, solver="rk4", device="cpu", seed=0, shard=None, observations=None,
          replica=None, replica_same_data=False, n_batch=None, **param_overrides):
    """(args, settings, data_pair, parameters, model, training) for a named synthetic workload.
    shard: parallel.SampleShard (the IWAE-sample axis split over ranks).  replica: parallel.RowReplica (every rank
    its own rows and draws, gradients averaged): the model is initialised from `seed` on every rank, the plate and the
    random streams from a rank-specific seed (replica_same_data: from `seed` too -- every replica then computes the
    same step, which is what the plumbing test wants)."""
    from vihds.config import Config
    from vihds.parameters import Parameters
    from vihds.training import Training
    from vihds.vae import build_model

    spec_fn, n_times = WORKLOADS[workload]
    spec = spec_fn(solver)
    spec["params"].update(param_overrides)
    spec["params"]["n_batch"] = n_batch or n_rows  # (n_batch < n_rows: several batches per epoch, the last one ragged)
    args = make_args(n_iwae, seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    settings = Config(args=None, spec=spec)
    settings.device = torch.device(device)
    settings.seed = seed
    data_seed = seed if (replica is None or replica_same_data) else seed + 1009 * (replica.rank + 1)
    ds = SyntheticPlateDataset(settings.data, n_rows, n_times, data_seed)
    idx = np.arange(n_rows)
    data = TimeSeriesDatasetPair(Subset(ds, idx), Subset(ds, idx), settings.data)
    parameters = Parameters(settings.params)
    # white-box workloads observe what their own model produces (a well-posed inference problem: the objective does not run
    # away at the spec's learning rate); dr_blackbox keeps the plate-like curves -- its network starts
    # from random weights, there is no "own model" to simulate from, and its objective stays finite on them
    if settings.device.type == "cuda" and workload in MODEL_SIMULATED:
        simulate_observations(settings, parameters, ds, settings.device, data_seed)
    elif observations is not None:
        ds.observations = observations
    torch.manual_seed(seed)
    model = build_model(args, settings, data, parameters, shard=shard)
    model.replica = replica
    training = Training(args, settings, data, parameters, model)
    torch.manual_seed(data_seed + 1)  # the in-kernel generators are seeded from torch's stream at their first use
    return args, settings, data, parameters, model, training


class RecordedPlate(Dataset):
    """A processed plate as the reference's data pipeline produced it (vihds/datasets.py:173-224 of the reference: CSVs ->
    merged, scaled, background-subtracted observations, log1p'ed treatments, device one-hot blocks), read from the arrays a
    golden trace file recorded (tests/golden/trace_*.npz: `times, devices, dev_1hot, inputs, observations`)."""

    def __init__(self, z):
        self.times = torch.tensor(np.asarray(z["times"]))
        self.n_times, self.n_species = len(self.times), 4
        self.devices = np.asarray(z["devices"])
        self.dev_1hot = torch.tensor(np.asarray(z["dev_1hot"]))
        self.inputs = torch.tensor(np.asarray(z["inputs"]))
        self.observations = torch.tensor(np.asarray(z["observations"]))

    def __len__(self):
        return len(self.devices)

    def __getitem__(self, idx):
        if torch.is_tensor(idx):
            idx = idx.tolist()
        return {"devices": self.devices[idx], "dev_1hot": self.dev_1hot[idx], "inputs": self.inputs[idx],
                "observations": self.observations[idx]}


def build_recorded_plate(npz_path, n_iwae, solver="rk4", device="cpu", seed=0, **param_overrides):
    """(args, settings, data_pair, parameters, model, training) on the REAL plate of the reference's specs/dr_constant_icml.yaml
    -- 312 wells, 234 / 78 train / validation rows of its own seeded split (datasets.py:199-222) -- with the experiment
    definition recorded next to it (spec_json: the YAML as parsed), e.g. learning_rate 0.01 and MultiStepLR [250, 1000]."""
    import json

    from vihds.config import Config
    from vihds.datasets import split_dataset
    from vihds.parameters import Parameters
    from vihds.training import Training
    from vihds.vae import build_model

    z = np.load(npz_path)
    spec = json.loads(str(z["spec_json"]))
    spec["params"]["solver"] = solver
    spec["params"].update(param_overrides)
    args = make_args(n_iwae, seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    settings = Config(args=None, spec=spec)
    settings.device = torch.device(device)
    settings.seed = seed
    data = split_dataset(RecordedPlate(z), args, settings.data)
    parameters = Parameters(settings.params)
    torch.manual_seed(seed)
    model = build_model(args, settings, data, parameters)
    training = Training(args, settings, data, parameters, model)
    torch.manual_seed(seed + 1)
    return args, settings, data, parameters, model, training
