"""Sample containers and q / p distribution chains with the reference's names and call signatures
(vihds/distributions.py), backed by packed device buffers so that sampling, clipping and both log-densities of
ALL parameters are one HIP kernel (vihds_theta_fwd / vihds_theta_bwd) instead of ~35 x (sample, clip, 2 log-probs)
python-loop iterations.

Layout: theta is one [R,B,S] buffer (row = parameter, S fastest) -- the structure-of-arrays layout the ODE
kernels read; every named attribute (`theta.r`, `theta.K`, ...) is a view of one row.
"""
import math
from collections import OrderedDict

import numpy as np
import torch

from vihds import ops
from vihds.utils import variable_summaries

LOG2PI = math.log(2 * math.pi)
NORMAL, LOGNORMAL, CONSTANT = 0, 1, 2


class DotOperatorSamples(object):
    """Named samples with attribute access (reference distributions.py:29-55) over a packed [R,B,S] buffer."""

    def __init__(self):
        object.__setattr__(self, "samples", OrderedDict())
        object.__setattr__(self, "keys", [])
        object.__setattr__(self, "values", [])
        object.__setattr__(self, "_packed", None)
        object.__setattr__(self, "_row_of", {})
        object.__setattr__(self, "_rebound", {})
        object.__setattr__(self, "_log_prob_cache", {})
        object.__setattr__(self, "_u", None)
        object.__setattr__(self, "_sample_window", None)  # (S_total, s_offset) when S is sharded over ranks
        object.__setattr__(self, "_row_offset", None)  # (offset [B,n], (src, dst, n)): see ops.OdeSolveObserve

    @classmethod
    def from_packed(cls, names, packed):
        self = cls()
        object.__setattr__(self, "_packed", packed)
        for i, n in enumerate(names):
            self._row_of[n] = i
            self.add(n, packed[i])
        return self

    def add(self, distribution_name, distribution_sample):
        assert distribution_name not in self.samples, "DotOperatorSamples already has %s" % distribution_name
        self.samples[distribution_name] = distribution_sample
        self.keys.append(distribution_name)
        self.values.append(distribution_sample)
        object.__setattr__(self, distribution_name, distribution_sample)

    def __setattr__(self, name, value):
        # condition_theta re-binds attributes (theta.aR = ..., theta.y1 = y1 + offset) without touching
        # `.samples` (reference dr_constant.py:129-130, dr_blackbox.py:92-95); remember them for pack()
        if isinstance(value, torch.Tensor):
            self._rebound[name] = value
        object.__setattr__(self, name, value)

    def __str__(self):
        return "".join("%s = %s\n" % kv for kv in self.samples.items())

    def get_n_batch(self):
        return self.values[0].size()[0]

    def get_n_samples(self):
        return self.values[0].size()[1]

    def get_tensors(self):
        return self.values

    def bind_reserved_row(self, name, row):
        """Expose a reserved row of the packed buffer (filled in place by a kernel, e.g. the device conditioner)
        as an attribute -- what `theta.aR = ...` does in the reference -- without a copy."""
        object.__setattr__(self, name, self._packed[row])
        self._row_of[name] = row
        self._rebound.pop(name, None)

    def n_reserved_rows(self):
        return 0 if self._packed is None else self._packed.shape[0] - len(self.samples)

    def pack(self, slot_names):
        """([R,B,S] buffer, name -> row) holding at least `slot_names`, as the simulator sees them (i.e. with
        re-bound attributes).  Zero-copy when nothing was re-bound."""
        if self._packed is not None:
            extra = [n for n in slot_names if n in self._rebound or n not in self._row_of]
            if not extra:
                return self._packed, self._row_of
            row_of = dict(self._row_of)
            base = self._packed.shape[0]
            for j, n in enumerate(extra):
                row_of[n] = base + j
            shape = self._packed.shape[1:]
            rows = torch.stack([getattr(self, n).expand(shape) for n in extra])
            return torch.cat([self._packed, rows], 0), row_of
        shape = self.values[0].shape
        return torch.stack([getattr(self, n).expand(shape) for n in slot_names]), {n: i for i, n in enumerate(slot_names)}


class TfCrnDistribution(object):
    def __init__(self, variable):
        self.variable = variable
        self.waiting_slots = {}

    def slots_are_pending(self):
        return False

    def clip(self, sample, stddevs=3):
        return sample


class TfConstant(TfCrnDistribution):
    kind = CONSTANT

    def __init__(self, c=None, value=None, wait_for_assigned=False, variable=False):
        super(TfConstant, self).__init__(variable)
        self.value = value
        self.nbr_params = 1
        self.param_names = ["value"]

    def assign_free_and_constrained(self, value):
        self.value = value

    def sample(self, u, stop_grad=False):
        return torch.zeros_like(u) + self.value.to(u.device)

    def log_prob(self, x, stop_grad=False):
        return torch.zeros_like(x)

    def get_tensors(self):
        return [self.value]

    def get_tensor_names(self, name):
        return ["%s.value" % name]

    def attach_summaries(self, writer, epoch, name, plot_histograms):
        pass

    def __str__(self):
        return "%s value = %s" % (self.__class__, self.value)


class TfNormal(TfCrnDistribution):
    """mu / prec / sigma container; arithmetic as reference distributions.py:266-366."""

    kind = NORMAL

    def __init__(self, mu=None, c=None, sigma=None, prec=None, variable=True, wait_for_assigned=False):
        super(TfNormal, self).__init__(variable)
        self.mu = mu
        self.log_prec = None
        if not wait_for_assigned:
            if sigma is None:
                if prec is not None:
                    sigma = 1.0 / prec.sqrt()
            else:
                prec = 1.0 / (sigma * sigma)
        self._sigma, self.prec = sigma, prec
        self.nbr_params = 2
        self.param_names = ["mu", "prec"]

    @property
    def sigma(self):
        # derived lazily: the hot path never needs the per-distribution sigma tensors (the theta kernel forms
        # 1/sqrt(prec) itself), so q's ~30 distributions cost no launches per step
        if self._sigma is None and self.prec is not None:
            self._sigma = 1.0 / self.prec.sqrt()
        return self._sigma

    def assign_free_and_constrained(self, mu, log_prec, prec):
        self.mu, self.log_prec, self.prec = mu, log_prec, prec
        self._sigma = None

    def _transform(self, z):
        return z

    def _untransform(self, x):
        return x

    def sample(self, u, stop_grad=False):
        mu, sigma = (self.mu.detach(), self.sigma.detach()) if stop_grad else (self.mu, self.sigma)
        return self._transform(mu + sigma * u)

    def clip_bounds(self, stddevs):
        lower = self._transform(self.mu - stddevs * self.sigma).data.reshape(-1)[0]
        upper = self._transform(self.mu + stddevs * self.sigma).data.reshape(-1)[0]
        return lower, upper

    def clip(self, x, stddevs=3):
        lower, upper = self.clip_bounds(stddevs)
        return x.clamp(lower, upper)

    def log_prob(self, x, stop_grad=False):
        prec, mu = (self.prec.detach(), self.mu.detach()) if stop_grad else (self.prec, self.mu)
        v = self._untransform(x)
        lp = -LOG2PI + 0.5 * (prec + 1e-12).log() - 0.5 * prec * (mu - v).pow(2)
        return lp - v if self.kind == LOGNORMAL else lp

    def get_tensors(self):
        return [self.mu, self.prec]

    def get_tensor_names(self, name):
        return ["%s.mu" % name, "%s.prec" % name]

    def attach_summaries(self, writer, epoch, name, plot_histograms):
        if writer is None:
            return
        if self.variable:
            variable_summaries(writer, epoch, self.mu, name + ".mu", plot_histograms)
            variable_summaries(writer, epoch, self.prec, name + ".prec", plot_histograms)
        else:
            writer.add_scalar("%s/mu" % name, self.mu.mean(), epoch)
            writer.add_scalar("%s/prec" % name, self.prec.mean(), epoch)

    def __str__(self):
        return "%s mu = %s  prec = %s" % (self.__class__, self.mu, self.prec)


class TfLogNormal(TfNormal):
    kind = LOGNORMAL

    def _transform(self, z):
        return z.exp()

    def _untransform(self, x):
        return (x + 1e-12).log()


CLASS_OF_KIND = {NORMAL: TfNormal, LOGNORMAL: TfLogNormal, CONSTANT: TfConstant}


class ChainedDistribution(object):
    """Ordered set of named distributions (reference distributions.py:58-187) with a packed device image:
    kind int32 [P], mu [P,B] (or [P,1] for a prior), prec [P,B]."""

    def __init__(self, name="unknown"):
        self.name = name
        self._distributions = OrderedDict()
        self.slot_dependencies = OrderedDict()
        self._image = None      # (kind_dev, mu_PB, prec_PB)
        self._packed_q = None   # (kind_dev, q_all [2P,B] = [mu ; log_prec], names) supplied by the Encoder
        self._builder = None    # fills _distributions on first use (the hot path never needs the member objects)
        self._clip_cache = {}

    @property
    def distributions(self):
        if self._builder is not None:
            builder, self._builder = self._builder, None
            builder(self)
        return self._distributions

    # ---- construction --------------------------------------------------------------------------
    def add_distribution(self, key, value, slots=None):
        assert key not in self._distributions, "ChainedDistribution (%s) already has %s" % (self.name, key)
        self._distributions[key] = value
        setattr(self, key, value)
        self.slot_dependencies[key] = slots or {}
        self._image = None

    def attach_image(self, kind_dev, mu, prec):
        self._image = (kind_dev, mu, prec)

    def attach_packed(self, kind_dev, q_all, names, builder, q_rows=None):
        """q_all [2P,B]: means and log-precisions; q_rows [2P] (int32, device): the row of mu_p, then of log_prec_p
        (None: [mu rows ; log_prec rows])."""
        P = len(names)
        if q_rows is None:
            q_rows = torch.arange(2 * P, dtype=torch.int32, device=q_all.device)
        self._packed_q = (kind_dev, q_all, list(names))
        self._q_rows = q_rows
        self._builder = builder

    def names(self):
        return self._packed_q[2] if self._packed_q is not None else list(self.distributions.keys())

    def order_distributions(self):
        return OrderedDict((n, i) for i, n in enumerate(self.distributions))

    def get_theta_names(self):
        return list(self.distributions.keys())

    def kinds(self):
        return [d.kind for d in self.distributions.values()]

    def image(self, device, n_batch=1):
        """Packed (kind, mu [P,Bq], prec [P,Bq]); built from the member tensors when the Encoder did not attach one."""
        if self._image is None and self._packed_q is not None:
            kind, q_all, names = self._packed_q
            P = len(names)
            rows = self._q_rows.long()
            self._image = (kind, q_all[rows[:P]], q_all[rows[P:]].exp())
        if self._image is None:
            mus, precs = [], []
            for d in self.distributions.values():
                if d.kind == CONSTANT:
                    mus.append(torch.as_tensor(d.value, dtype=torch.float32).reshape(-1)[:1].to(device).expand(n_batch))
                    precs.append(torch.ones(n_batch, device=device))
                else:
                    mus.append(d.mu.reshape(-1).to(device).expand(n_batch))
                    precs.append(d.prec.reshape(-1).to(device).expand(n_batch))
            kind = torch.tensor(self.kinds(), dtype=torch.int32, device=device)
            self._image = (kind, torch.stack(mus).contiguous(), torch.stack(precs).contiguous())
        return self._image

    def clip_image(self, stddevs, device):
        """Per-parameter clip bounds [P] as the reference forms them (distributions.py:332-336, :377-381)."""
        key = (float(stddevs), str(device))
        if key not in self._clip_cache:
            lo, hi = [], []
            for d in self.distributions.values():
                if d.kind == CONSTANT:
                    lo.append(-float("inf"))
                    hi.append(float("inf"))
                else:
                    a, b = d.clip_bounds(stddevs)
                    lo.append(float(a))
                    hi.append(float(b))
            self._clip_cache[key] = (torch.tensor(lo, dtype=torch.float32, device=device),
                                     torch.tensor(hi, dtype=torch.float32, device=device))
        return self._clip_cache[key]

    # ---- the fused hot path ----------------------------------------------------------------------
    def _kernel_inputs(self, list_of_u, p, stddevs):
        """(names, P, p_mu, p_prec, clip_lo, clip_hi) for the theta kernels."""
        names = self.names()
        P = len(names)
        assert list_of_u.shape[-1] == P, (
            "ChainedDistribution (%s #= %d):: must give a list of u's, one for each distribution."
            % (self.name, list_of_u.shape[-1]))
        dev = list_of_u.device
        if p is None:
            p_mu = p_prec = torch.ones(P, device=dev)
            inf = torch.full((P,), float("inf"), device=dev)
            lo, hi = -inf, inf
        else:
            assert p.names() == names, "q and p must chain the same names"
            _, pm, pp = p.image(dev, 1)
            p_mu, p_prec = pm[:, 0], pp[:, 0]
            lo, hi = p.clip_image(stddevs, dev)
        return names, P, p_mu, p_prec, lo, hi

    def _wrap_samples(self, names, theta, log_q, log_p, p, u_used):
        samples = DotOperatorSamples.from_packed(names, theta)
        # the standard-normal draws behind these samples (an output when the kernel drew them)
        object.__setattr__(samples, "_u", u_used)
        samples._log_prob_cache[id(self)] = log_q
        if p is not None:
            samples._log_prob_cache[id(p)] = log_p
        return samples

    def sample_clip_log_prob(self, list_of_u, p, stddevs, n_extra_rows=0):
        """q.sample(u) -> p.clip(., stddevs) -> (theta, log q(theta), log p(theta)) in ONE kernel
        (reference vae.py:31-34 + training.py:136-137).  log q / log p are cached on the returned theta so
        that the later q.log_prob(theta) / p.log_prob(theta) calls (Training.cost) are free."""
        names, P, p_mu, p_prec, lo, hi = self._kernel_inputs(list_of_u, p, stddevs)
        dev = list_of_u.device
        n_batch = list_of_u.shape[0]
        if isinstance(list_of_u, ops.KernelNormal) and self._packed_q is None:
            raise RuntimeError("u_rng: kernel needs the encoder's packed q tables")
        if self._packed_q is not None:
            kind, q_all, _ = self._packed_q
            theta, log_q, log_p, u_used = ops.ThetaSampleLogProbPacked.apply(q_all, kind, p_mu, p_prec, lo, hi,
                                                                             list_of_u, P + n_extra_rows, self._q_rows)
        else:
            kind, q_mu, q_prec = self.image(dev, n_batch)
            theta, log_q, log_p = ops.ThetaSampleLogProb.apply(q_mu, q_prec, kind, p_mu, p_prec, lo, hi, list_of_u,
                                                               P + n_extra_rows)
            u_used = list_of_u
        return self._wrap_samples(names, theta, log_q, log_p, p, u_used)

    def decoder_step_fused(self, list_of_u, p, stddevs, n_extra_rows, spec_of, cond, times, obs, dev1hot, cond_job):
        """sample_clip_log_prob AND the decoder (conditioner rows, log-likelihood, unit-weight adjoint) in one launch
        (ops.DecoderStepFused).  `spec_of(names)` -> OdeProblemSpec for the packed row order.  Returns
        (theta samples, logp [4,B,S]); raises ops.FusedTrainingUnsupported when the library declines."""
        if self._packed_q is None:
            raise ops.FusedTrainingUnsupported("needs the encoder's packed q tables")
        names, P, p_mu, p_prec, lo, hi = self._kernel_inputs(list_of_u, p, stddevs)
        kind, q_all, _ = self._packed_q
        theta, log_q, log_p, u_used, logp = ops.DecoderStepFused.apply(
            q_all, kind, p_mu, p_prec, lo, hi, list_of_u, P + n_extra_rows, self._q_rows, spec_of(names), cond, times,
            obs, dev1hot, cond_job)
        return self._wrap_samples(names, theta, log_q, log_p, p, u_used), logp

    def theta_ode_fused(self, list_of_u, p, stddevs, n_extra_rows, spec_of, cond, times, obs, dev1hot, weights, offset):
        """sample_clip_log_prob with the sampling stage INSIDE the ODE forward launch (ops.ThetaOdeFused): returns (theta
        samples, traj [T,N,B,S], logp [4,B,S]); offset = None or (weight, bias, (src row, dst row, n)) of dr_blackbox's
        condition_theta.  Raises ops.FusedTrainingUnsupported when the library declines."""
        if self._packed_q is None:
            raise ops.FusedTrainingUnsupported("needs the encoder's packed q tables")
        names, P, p_mu, p_prec, lo, hi = self._kernel_inputs(list_of_u, p, stddevs)
        kind, q_all, _ = self._packed_q
        off_w, off_b, off_rows = offset if offset is not None else (None, None, None)
        theta, log_q, log_p, u_used, traj, logp = ops.ThetaOdeFused.apply(
            q_all, kind, p_mu, p_prec, lo, hi, list_of_u, P + n_extra_rows, self._q_rows, spec_of(names), cond, times, obs,
            dev1hot, weights, off_w, off_b, off_rows)
        return self._wrap_samples(names, theta, log_q, log_p, p, u_used), traj, logp

    # ---- reference-compatible entry points -------------------------------------------------------------
    def sample(self, list_of_u, device, stop_grad=False):
        """reference distributions.py:119-142 (no clipping); same kernel with infinite bounds."""
        u = list_of_u if isinstance(list_of_u, torch.Tensor) else torch.as_tensor(np.asarray(list_of_u))
        u = u.to(device)
        if stop_grad:
            kind, mu, prec = self.image(u.device, u.shape[0])
            saved = (self._image, self._packed_q)
            self._image, self._packed_q = (kind, mu.detach(), prec.detach()), None
            try:
                return self.sample_clip_log_prob(u, None, 0.0)
            finally:
                self._image, self._packed_q = saved
        return self.sample_clip_log_prob(u, None, 0.0)

    def clip(self, theta, stddevs=3, skip=None):
        """reference distributions.py:76-85: one clamp over the packed buffer."""
        names = list(theta.samples.keys())
        packed, row_of = theta.pack(names)
        dev = packed.device
        lo, hi = self.clip_image(stddevs, dev)
        idx = {n: i for i, n in enumerate(self.distributions)}
        lo_r = torch.full((packed.shape[0],), -float("inf"), device=dev)
        hi_r = torch.full((packed.shape[0],), float("inf"), device=dev)
        for n in names:
            if n in idx and not (skip is not None and n in skip):
                lo_r[row_of[n]] = lo[idx[n]]
                hi_r[row_of[n]] = hi[idx[n]]
        clipped = torch.clamp(packed, lo_r[:, None, None], hi_r[:, None, None])
        order = [row_of[n] for n in names]
        if order != list(range(len(names))):
            clipped = clipped[torch.tensor(order, device=dev)]
        elif clipped.shape[0] != len(names):
            clipped = clipped[: len(names)]
        return DotOperatorSamples.from_packed(names, clipped)

    def _log_prob_rows(self, theta, stop_grad):
        names = [n for n in theta.samples if n in self.distributions]
        if not names:
            return None
        x = torch.stack([theta.samples[n] for n in names])  # [P',B,S]
        dev = x.device
        kind, mu, prec = self.image(dev, x.shape[1])
        idx = torch.tensor([list(self.distributions).index(n) for n in names], device=dev)
        kind, mu, prec = kind[idx], mu[idx][:, :, None], prec[idx][:, :, None]
        if stop_grad:
            mu, prec = mu.detach(), prec.detach()
        ln = (kind == LOGNORMAL)[:, None, None]
        const = (kind == CONSTANT)[:, None, None]
        v = torch.where(ln, (x + 1e-12).log(), x)
        lp = -LOG2PI + 0.5 * (prec + 1e-12).log() - 0.5 * prec * (mu - v).pow(2)
        lp = torch.where(ln, lp - v, lp)
        return torch.where(const, torch.zeros_like(lp), lp)

    def log_prob(self, theta, stop_grad=False):
        """reference distributions.py:64-74.  Free when theta came out of sample_clip_log_prob of this chain."""
        if not stop_grad and id(self) in theta._log_prob_cache:
            return theta._log_prob_cache[id(self)]
        rows = self._log_prob_rows(theta, stop_grad)
        return 0.0 if rows is None else rows.sum(0)

    def log_prob_mat(self, theta, stop_grad=False):
        return self._log_prob_rows(theta, stop_grad).permute(1, 2, 0)

    def get_tensors(self):
        tensors = []
        for d in self.distributions.values():
            tensors.extend(d.get_tensors())
        return tensors

    def get_tensor_names(self):
        names = []
        for name, d in self.distributions.items():
            names.extend(d.get_tensor_names(name))
        return names

    def attach_summaries(self, writer, epoch, plot_histograms):
        for name, d in self.distributions.items():
            d.attach_summaries(writer, epoch, name, plot_histograms)

    def __str__(self):
        return "".join("%s = %s slots=[%s]\n" % (k, d, self.slot_dependencies[k]) for k, d in self.distributions.items())

    pretty_print = __str__
