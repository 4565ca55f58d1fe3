"""YAML `params:` block -> ordered parameter descriptions (counterpart of the reference's vihds/parameters.py).

Host-side configuration parsing; the numbers it yields (prior mu / sigma / prec, initial free values, the
local / global_conditioned / global / constant ordering) feed the theta kernel's packed [P] tables."""
import math

import numpy as np

NORMAL, LOGNORMAL, CONSTANT = 0, 1, 2
KIND_OF = {"Normal": NORMAL, "LogNormal": LOGNORMAL, "Constant": CONSTANT}
LEVELS = ("local", "global_cond", "global", "constant")  # theta order (reference encoders.py:78-84,402)


class DistributionDescription(object):
    """One named parameter: distribution kind, prior (mu, sigma | prec), q initial free values, conditioning."""

    def __init__(self, name, level, spec, conditioning=None):
        self.name, self.level, self.conditioning = name, level, conditioning
        dist = spec["distribution"]
        if dist not in KIND_OF:
            # TruncNormal / Kumaraswamy raise NotImplementedError inside the reference (distributions.py:384-529)
            raise NotImplementedError("distribution '%s' of parameter '%s' is not implemented" % (dist, name))
        self.kind = KIND_OF[dist]
        if self.kind == CONSTANT:
            self.value = float(spec.get("value", 0.0))
            self.free_params, self.params = ["value"], ["value"]
            self.init_free_params = [self.value]
            self.defaults = {"value": self.value}
            return
        self.mu = spec.get("mu", 0.0)
        self.sigma = spec.get("sigma", None)
        self.prec = spec.get("prec", None)
        if isinstance(self.mu, str) or isinstance(self.prec, str) or isinstance(self.sigma, str):
            raise NotImplementedError("slot dependencies between distributions ('%s') are not used by any spec" % name)
        self.defaults = {"mu": self.mu, "sigma": self.sigma, "prec": self.prec}
        self.free_params, self.params = ["mu", "log_prec"], ["mu", "prec"]
        # reference parameters.py:29-58: the q initial precision is 1 unless `prec:` is given explicitly -- a
        # `sigma:`-only spec never reaches the sigma branch because the "prec" key always exists (value None).
        init_prec = float(self.prec) if self.prec is not None else 1.0
        self.init_free_params = [float(self.mu), math.log(init_prec)]

    def prior_mu_sigma_prec(self):
        """The prior's tensors as TfNormal.__init__ derives them in fp32 (reference distributions.py:281-296)."""
        mu = np.float32(self.mu)
        if self.sigma is None:
            prec = np.float32(self.prec)
            sigma = np.float32(1.0) / np.sqrt(prec)
        else:
            sigma = np.float32(self.sigma)
            prec = np.float32(1.0) / (sigma * sigma)
        return mu, sigma, prec


class DotOperatorParams(object):
    def __init__(self):
        self.list_of_params = []

    def add(self, desc):
        if hasattr(self, desc.name):
            print("already have param named: ", desc.name)
            return
        setattr(self, desc.name, desc)
        self.list_of_params.append(desc.name)

    def get_parameter_counts(self):
        return len(self.list_of_params)

    def descriptions(self):
        return [getattr(self, n) for n in self.list_of_params]


class Parameters(object):
    """Parses shared / global / global_conditioned / local / constant (reference parameters.py:246-453)."""

    def __init__(self, params_dict):
        self.params_dict = params_dict
        shared = params_dict.get("shared", {}) or {}

        def resolve(v):
            d = v["distribution"]
            return shared[d] if d in shared else v

        for level, key in (("global", "global"), ("global_cond", "global_conditioned"), ("local", "local")):
            if key not in params_dict or params_dict[key] is None:
                print("load_%s_params:: None found in params_dict" % key)
                continue
            block = params_dict[key]
            cond = block.get("conditioning", None)
            if level == "global" and cond is not None:
                raise Exception("global_params can no longer have conditioning")
            if level == "global_cond" and cond is None:
                raise Exception("global_cond MUST have conditioning")
            p = DotOperatorParams()
            for k, v in block.items():
                if k != "conditioning":
                    p.add(DistributionDescription(k, level, resolve(v), cond))
            setattr(self, "_" + level, p)
        if "constant" in params_dict and params_dict["constant"] is not None:
            p = DotOperatorParams()
            for k, v in params_dict["constant"].items():
                p.add(DistributionDescription(k, "constant", {"distribution": "Constant", "value": v}))
            self._constant = p

    def level(self, level):
        return getattr(self, "_" + level, None)

    def get_parameter_counts(self):
        return tuple(self.level(lv).get_parameter_counts() if self.level(lv) else 0 for lv in LEVELS)

    def ordered(self):
        """All descriptions in theta order: local, global_cond, global, constant."""
        out = []
        for lv in LEVELS:
            if self.level(lv):
                out += self.level(lv).descriptions()
        return out

    def pretty_print(self):
        for lv in LEVELS:
            if self.level(lv):
                print("-----------------\n%s parameters\n-----------------" % lv.upper())
                for d in self.level(lv).descriptions():
                    print(d.name, d.defaults)
