"""Build the host-side model (Config -> Parameters -> Encoder/Decoder -> Training) from a golden fixture's
recorded experiment definition, with a stand-in dataset pair holding the fixture's batch."""
import argparse
import json

import numpy as np
import torch

from vihds.config import Config
from vihds.parameters import Parameters
from vihds.utils import attrify


class _FakeDataset(torch.utils.data.Dataset):
    def __init__(self, fx):
        self.times = fx.t("times")
        self.n_times = len(self.times)
        self.n_species = 4
        self.devices = np.asarray(fx.z["devices"])
        self.dev_1hot = fx.t("dev_1hot")
        self.inputs = fx.t("inputs")
        self.observations = fx.t("observations")

    def __len__(self):
        return len(self.devices)

    def __getitem__(self, idx):
        return {"devices": self.devices[idx], "dev_1hot": self.dev_1hot[idx], "inputs": self.inputs[idx],
                "observations": self.observations[idx]}


class _Pair(object):
    def __init__(self, ds, settings):
        idx = np.arange(len(ds))
        self.train = torch.utils.data.Subset(ds, idx)
        self.test = torch.utils.data.Subset(ds, idx)
        self.n_train = self.n_test = len(ds)
        self.depth = settings.data.device_depth
        self.n_conditions = len(settings.data.conditions)


def make_args(n_iwae, seed=0, gpu=None):
    return argparse.Namespace(yaml=None, experiment="test", seed=seed, epochs=1, test_epoch=1, plot_epoch=0,
                              train_samples=n_iwae, test_samples=n_iwae, dreg=True, precision_hidden_layers=None,
                              verbose=False, gpu=gpu, heldout=None, split=1, figures=False, folds=4)


def build_from_fixture(fx, gpu=None, **param_overrides):
    """Returns (args, settings, data_pair, parameters).  Seeds torch like the reference's Config(args) does."""
    spec = json.loads(str(fx.z["spec_json"]))
    spec["params"]["solver"] = fx.solver
    hid = "decoder_param/ode_model.precisions.prec_hidden.weight"
    if hid in fx.z.files and fx.model != "dr_blackbox":  # recorded with --precision_hidden_layers (run_xval.py:38)
        spec["params"]["n_hidden_decoder_precisions"] = int(fx.z[hid].shape[0])
    spec["params"].update(param_overrides)
    args = make_args(fx.S, seed=fx.cfg["seed"], gpu=gpu)
    np.random.seed(args.seed)
    torch.manual_seed(args.seed)
    settings = Config(args=None, spec=spec)
    if gpu is not None and torch.cuda.is_available():
        settings.device = torch.device("cuda:%d" % gpu)
    settings.seed = args.seed
    data = _Pair(_FakeDataset(fx), settings)
    return args, settings, data, Parameters(settings.params)


def batch_from_fixture(fx, device):
    return attrify({"devices": np.asarray(fx.z["devices"]), "dev_1hot": fx.t("dev_1hot", device),
                    "inputs": fx.t("inputs", device), "observations": fx.t("observations", device),
                    "times": fx.t("times", device)})
