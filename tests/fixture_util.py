"""Shared helpers for tests: load a golden fixture (tests/golden/*.npz, produced by the reference via
tests/golden/make_fixtures.py) into the argument shapes the oracle and the HIP path take."""
import json
import os
from collections import OrderedDict

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

ALL_FIXTURES = [
    "dr_constant_one_modeuler",
    "dr_constant_one_s5_modeulerwhile",
    "dr_constant_icml_tiny_modeuler",
    "dr_constant_icml_tiny_modeulerwhile",
    "dr_constant_icml_full_modeuler",
    "dr_constant_v2_tiny_modeuler",
    "auto_constant_tiny_modeuler",
    "dr_constant_precisions_tiny_modeuler",
    "dr_constant_precisions_hidden20_tiny_modeuler",
    "auto_constant_precisions_tiny_modeuler",
    "dr_blackbox_icml_tiny_modeuler",
    "dr_blackbox_icml_full_modeuler",  # BASELINE config 4's own shape (36 x 200) from the reference (round 6)
    "dr_blackbox_sized_tiny_modeuler",
    "prpr_constant_tiny_modeuler",
]

# Produced by `make_fixtures.py --patched`: the reference with its two CONSTRUCTION defects repaired in memory
# (OdeFunc.__init__'s arity, the non-existent init_with_params) -- "MODIFIED REFERENCE" in their provenance.
# The equations are the reference's own forward(); the only relay / degrader / inducer specs it ships are these.
# BASELINE config 3's training shape (36 rows x 1 000 samples) from the reference, LIGHT: u by seed, no theta arrays (see
# make_fixtures.py); used by dedicated tests, not by the parametrised fixture lists
LIGHT_FIXTURE_S1000 = "dr_constant_icml_s1000_light_modeuler"

PATCHED_FIXTURES = [
    "relay_constant_precisions_tiny_modeuler",
    "relay_constant_precisions_tiny_modeulerwhile",
    "relay_constant_precisions_full_modeuler",  # BASELINE config 5's own shape (36 x 200, T = 99), MODIFIED reference (round 6)
    "degrader_constant_precisions_tiny_modeuler",
    "inducer_constant_precisions_tiny_modeuler",
    "prpr_constant_precisions_tiny_modeuler",
]


class Fixture:
    def __init__(self, name):
        self.name = name
        self.z = np.load(os.path.join(GOLDEN, name + ".npz"))
        self.cfg = json.loads(str(self.z["config_json"]))
        self.model = self.cfg["model"]
        self.solver = self.cfg["solver"]
        self.names = [str(n) for n in self.z["theta_names"]]
        self.extra_names = [str(n) for n in self.z["extra_names"]]
        self.kinds = [int(k) for k in self.z["kind"]]

    def t(self, key, device="cpu", dtype=torch.float32):
        if key == "u" and "u" not in self.z.files:
            # LIGHT fixtures (make_fixtures.py, CASES): u is numpy's legacy normal stream from the recorded seed -- the
            # reference's own draw, vae.py:22-24 -- checked against the recorded u when the fixture was made
            state = np.random.get_state()
            np.random.seed(int(self.z["u_seed"]))
            u = np.random.randn(*[int(v) for v in self.z["u_shape"]]).astype(np.float32)
            np.random.set_state(state)
            return torch.tensor(u, dtype=dtype, device=device)
        return torch.tensor(np.asarray(self.z[key]), dtype=dtype, device=device)

    @property
    def B(self):
        return int(self.z["u_shape"][0]) if "theta" not in self.z.files else self.z["theta"].shape[1]

    @property
    def S(self):
        return int(self.z["u_shape"][1]) if "theta" not in self.z.files else self.z["theta"].shape[2]

    def theta_dict(self, requires_grad=False, device="cpu"):
        """Clipped theta as the decoder saw it (+ aR/aS from condition_theta), as leaf tensors."""
        th = OrderedDict()
        arr = self.t("theta", device)
        for i, n in enumerate(self.names):
            th[n] = arr[i].clone().requires_grad_(requires_grad)
        if self.extra_names:
            ex = self.t("extra_theta", device)
            for i, n in enumerate(self.extra_names):
                th[n] = ex[i].clone()
        return th

    def q_params(self, device="cpu"):
        """Per-parameter (mu, prec) of q as [B,1] tensors (globals were broadcast over B by the fixture)."""
        qm = self.t("q_mu", device)
        qp = self.t("q_prec", device)
        return [qm[i][:, None] for i in range(len(self.names))], [qp[i][:, None] for i in range(len(self.names))]

    def p_params(self, device="cpu"):
        pm = self.t("p_mu", device)
        pp = self.t("p_prec", device)
        return [pm[i] for i in range(len(self.names))], [pp[i] for i in range(len(self.names))]

    def decoder_weights(self, device="cpu"):
        """Neural-precision / neural-state weights in the oracle's naming."""
        d = {k[len("decoder_param/"):]: self.t(k, device) for k in self.z.files if k.startswith("decoder_param/")}
        prec_w = None
        if "ode_model.precisions.prec_production.weight" in d:
            prec_w = {
                "prod_w": d["ode_model.precisions.prec_production.weight"],
                "prod_b": d["ode_model.precisions.prec_production.bias"],
                "degr_w": d["ode_model.precisions.prec_degradation.weight"],
                "degr_b": d["ode_model.precisions.prec_degradation.bias"],
            }
            if "ode_model.precisions.prec_hidden.weight" in d:
                prec_w["hid_w"] = d["ode_model.precisions.prec_hidden.weight"]
                prec_w["hid_b"] = d["ode_model.precisions.prec_hidden.bias"]
        states_w = None
        if "ode_model.neural_states.states_hidden.weight" in d:
            states_w = {
                "hid_w": d["ode_model.neural_states.states_hidden.weight"],
                "hid_b": d["ode_model.neural_states.states_hidden.bias"],
                "prod_w": d["ode_model.neural_states.states_production.weight"],
                "prod_b": d["ode_model.neural_states.states_production.bias"],
                "degr_w": d["ode_model.neural_states.states_degradation.weight"],
                "degr_b": d["ode_model.neural_states.states_degradation.bias"],
            }
        offset = None
        if "ode_model.offset_layer.weight" in d:
            offset = (d["ode_model.offset_layer.weight"], d["ode_model.offset_layer.bias"])
        return prec_w, states_w, offset

    def decoder_weight_grads(self):
        return {k[len("decoder_grad/"):]: self.t(k) for k in self.z.files if k.startswith("decoder_grad/")}


def rel_err(a, b, dim="auto"):
    """Relative max-norm error, normalised PER SLICE along `dim`: max over slices k of
    max|a_k - b_k| / max|b_k|.  north_star's "1e-4 relative on trajectories" is meant per species (species differ by
    orders of magnitude, so one global max-norm would let a small species be 100 % off), per observed signal for the
    log-likelihoods and per parameter for gradients.  dim="auto": 4-D tensors are the reference's [B,S,N,T] views ->
    per species / signal (dim 2); everything else is one slice unless `dim` is given ([B,S,4] log-likelihoods: dim=2,
    [P,B,S] / [P,B] parameter rows: dim=0)."""
    a = torch.as_tensor(a).detach().to(torch.float64).cpu()
    b = torch.as_tensor(b).detach().to(torch.float64).cpu()
    if a.shape != b.shape:
        raise AssertionError("shape mismatch %s vs %s" % (tuple(a.shape), tuple(b.shape)))
    if dim == "auto":
        dim = 2 if a.dim() == 4 else None
    if dim is None or a.dim() == 0:
        return float((a - b).abs().max() / (b.abs().max() + 1e-30))
    a = a.movedim(dim, 0).reshape(a.shape[dim], -1)
    b = b.movedim(dim, 0).reshape(b.shape[dim], -1)
    if a.shape[1] == 0:
        return 0.0
    err = (a - b).abs().max(1).values
    ref = b.abs().max(1).values
    return float((err / (ref + 1e-30)).max())
