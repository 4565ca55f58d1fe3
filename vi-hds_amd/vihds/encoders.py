"""Encoder: data -> q(theta | x, d), plus the prior p (counterpart of the reference's vihds/encoders.py).

Stays PyTorch-ROCm (conv / linear trunk: SURVEY.md 2 row 15, out of scope as kernels).  Two things differ from
a literal transcription, both value-preserving:
  * the per-parameter `nn.Linear(n, 1)` heads (reference encoders.py:137-141, :183-185) are evaluated as ONE
    matmul per level; they are *initialised* one head at a time in the reference's construction order, so the same
    torch seed yields the same weights;
  * the heads write straight into the packed [P,B] (mu, log_prec) tables the theta kernel reads.
"""
from collections import OrderedDict

import torch
from torch import nn

from vihds.distributions import CLASS_OF_KIND, CONSTANT, ChainedDistribution, TfConstant


class ConditionalEncoder(nn.Module):
    """Conv1d -> AvgPool1d(stride 1) -> flatten -> Linear -> tanh (reference encoders.py:16-55)."""

    def __init__(self, n_channels, n_obs, params):
        super(ConditionalEncoder, self).__init__()
        self.n_outputs = params.n_hidden
        n_conv = n_obs - (params.filter_size - 1)
        n_pool = n_conv - (params.pool_size - 1)
        self.conv = nn.Conv1d(n_channels, params.n_filters, params.filter_size)
        nn.init.orthogonal_(self.conv.weight)
        self.pool = nn.AvgPool1d(params.pool_size, stride=1)
        self.lin = nn.Linear(n_pool * params.n_filters, self.n_outputs)
        nn.init.orthogonal_(self.lin.weight)
        if params.transfer_func != "tanh":
            raise Exception("Unknown activation layer %s" % params["transfer_func"])
        self.act = nn.Tanh()

    def forward(self, x):
        x = self.pool(self.conv(x))
        return self.act(self.lin(x.view(x.size(0), -1)))


class LocalAndGlobal:
    """Tuple of local, global-conditional, global and constant items (reference encoders.py:58-92)."""

    def __init__(self, loc, glob_cond, glob, const):
        self.loc, self.glob_cond, self.glob, self.const = loc, glob_cond, glob, const

    @classmethod
    def from_list(cls, seq):
        return cls(seq[0], seq[1], seq[2], seq[3])

    def to_list(self):
        return [self.loc, self.glob_cond, self.glob, self.const]

    def sum(self):
        return self.loc + self.glob_cond + self.glob + self.const


class _Heads(nn.Module):
    """All (mu, log_prec) heads of one conditioning level as a single Linear."""

    def __init__(self, descs, n_inputs, use_bias):
        super(_Heads, self).__init__()
        ws, bs = [], []
        for _d in descs:
            for _free in ("mu", "log_prec"):  # reference construction order, one default-initialised head each
                layer = nn.Linear(n_inputs, 1, use_bias)
                ws.append(layer.weight.data)
                if use_bias:
                    bs.append(layer.bias.data)
        # stored as [all mu heads ; all log_prec heads] so both outputs are contiguous row blocks
        order = list(range(0, len(ws), 2)) + list(range(1, len(ws), 2))
        self.weight = nn.Parameter(torch.cat([ws[i] for i in order], 0))
        self.bias = nn.Parameter(torch.cat([bs[i] for i in order], 0)) if use_bias else None
        self.n = len(descs)

    def forward(self, x):
        return torch.nn.functional.linear(x, self.weight, self.bias).t()  # [2*n, B]: mu rows then log_prec rows


class _PackQTables(torch.autograd.Function):
    """Means and log-precisions of all P parameters as one [2P, B] table, level by level:
    [local mu; local log_prec; global-cond mu; global-cond log_prec; global mu; global log_prec; const values; zeros]
    -- ONE concatenation of the pieces as the heads emit them; the backward hands each level a contiguous row range
    (heads) / one reduction over B (globals) instead of autograd's per-slice zero-fill + copy + add chains.
    `rows(...)` gives the [2P] row map (mu rows, then log_prec rows) the theta kernel consumes."""

    @staticmethod
    def rows(nl, ng, ngl, nc):
        mu, lp, base = [], [], 0
        for n in (nl, ng, ngl, nc):
            mu += list(range(base, base + n))
            lp += list(range(base + n, base + 2 * n))
            base += 2 * n
        return mu + lp

    @staticmethod
    def forward(ctx, local_t, gcond_t, global_free, const_values, B, const_zeros=None):
        dev = (local_t if local_t is not None else global_free if global_free is not None else const_values).device
        nl = 0 if local_t is None else local_t.shape[0] // 2
        ng = 0 if gcond_t is None else gcond_t.shape[0] // 2
        ngl = 0 if global_free is None else global_free.shape[1]
        nc = const_values.shape[0]
        ctx.sizes = (nl, ng, ngl, nc, B)
        parts = []
        if nl:
            parts.append(local_t)
        if ng:
            parts.append(gcond_t)
        if ngl:
            parts.append(global_free.reshape(2 * ngl)[:, None].expand(-1, B))
        if nc:
            zeros = const_zeros if const_zeros is not None else torch.zeros(nc, device=dev)
            parts.append(const_values[:, None].expand(-1, B))
            parts.append(zeros[:, None].expand(-1, B))
        return torch.cat(parts, 0)

    @staticmethod
    def backward(ctx, g):
        nl, ng, ngl, nc, B = ctx.sizes
        g_local = g[: 2 * nl] if nl else None
        g_gcond = g[2 * nl: 2 * (nl + ng)] if ng else None
        g_glob = g[2 * (nl + ng): 2 * (nl + ng + ngl)].sum(1).view(2, ngl) if ngl else None
        return g_local, g_gcond, g_glob, None, None, None


class Encoder(nn.Module):
    """reference encoders.py:348-418.  forward(data) -> q (a ChainedDistribution); `.p` is the prior."""

    def __init__(self, parameters, data, verbose=False, device=None):
        super(Encoder, self).__init__()
        print("Initialising encoder")
        self.verbose = verbose
        self.parameter_specs = parameters  # (the reference shadows nn.Module.parameters here, encoders.py:359)
        self.n_species = data.train.dataset.n_species
        self.n_times = data.train.dataset.n_times
        self.device = torch.device(device) if device is not None else torch.device("cpu")
        pd = parameters.params_dict
        self.conditional = ConditionalEncoder(self.n_species, self.n_times - 1, pd)
        lvl = lambda name: parameters.level(name).descriptions() if parameters.level(name) else []  # noqa: E731
        self.local, self.gcond, self.glob, self.const = lvl("local"), lvl("global_cond"), lvl("global"), lvl("constant")
        self.descs = self.local + self.gcond + self.glob + self.const
        self.names = [d.name for d in self.descs]

        def n_in(descs, with_data):
            if not descs:
                return 0, False, False
            cond = descs[0].conditioning or {}
            tr, dv = bool(cond.get("treatments", False)), bool(cond.get("devices", False))
            return (self.conditional.n_outputs if with_data else 0) + (data.n_conditions if tr else 0) + (data.depth if dv else 0), tr, dv

        n_l, self.l_tr, self.l_dv = n_in(self.local, True)
        n_g, self.g_tr, self.g_dv = n_in(self.gcond, False)
        self.local_heads = _Heads(self.local, n_l, True) if self.local else None
        self.gcond_heads = _Heads(self.gcond, n_g, False) if self.gcond else None
        if self.glob:  # [2, n_global]: row 0 = mu, row 1 = log_prec
            self.global_free = nn.Parameter(torch.tensor([d.init_free_params for d in self.glob],
                                                         dtype=torch.float32).t().contiguous())
        else:
            self.global_free = None
        self.register_buffer("const_values", torch.tensor([d.value for d in self.const], dtype=torch.float32))
        self.register_buffer("const_zeros", torch.zeros(len(self.const), dtype=torch.float32))
        self.register_buffer("kind", torch.tensor([d.kind for d in self.descs], dtype=torch.int32))
        self.register_buffer("q_rows", torch.tensor(_PackQTables.rows(len(self.local), len(self.gcond), len(self.glob),
                                                                      len(self.const)), dtype=torch.int32))
        # "encoder_kernel" (default on): the fused HIP encoder on the GPU; off = the nn.Module path below (the only
        # one on the CPU, where this class is used for construction / host-side checks)
        self.use_kernel = bool(pd.get("encoder_kernel", True)) if hasattr(pd, "get") else True
        self._shapes = {}
        self.to(self.device)
        self.set_up_p()

    def set_up_p(self):
        """Prior chain in theta order (reference encoders.py:406-414, :285-345)."""
        p = ChainedDistribution(name="p")
        dev = self.device
        for d in self.descs:
            if d.kind == CONSTANT:
                p.add_distribution(d.name, TfConstant(value=torch.tensor([d.value], device=dev)))
            else:
                kw = {k: (None if v is None else torch.tensor([v], dtype=torch.float32, device=dev))
                      for k, v in d.defaults.items()}
                p.add_distribution(d.name, CLASS_OF_KIND[d.kind](**kw))
        self.p = p

    def _kernel_shape(self, B, data):
        """struct vihds_encoder_shape for this encoder and batch size (cached)."""
        from vihds import hip

        if B not in self._shapes:
            c = self.conditional
            s = hip.EncoderShape()
            s.B, s.C_in, s.L = B, c.conv.in_channels, self.n_times - 1
            s.F, s.K, s.pool, s.H = c.conv.out_channels, c.conv.kernel_size[0], c.pool.kernel_size[0], c.n_outputs
            s.n_tr, s.D = data.inputs.shape[1], data.dev_1hot.shape[1]
            s.nl, s.l_tr, s.l_dv = len(self.local), int(self.l_tr), int(self.l_dv)
            s.ng, s.g_tr, s.g_dv = len(self.gcond), int(self.g_tr), int(self.g_dv)
            s.ngl, s.nc = len(self.glob), len(self.const)
            self._shapes[B] = s
        return self._shapes[B]

    def evaluate_q(self, data):
        B = data.observations.shape[0]
        obs = data.observations
        delta_obs = data.get("delta_obs", None) if hasattr(data, "get") else None  # staged with the batch (graph mode)
        if delta_obs is None:
            delta_obs = obs[:, :, 1: self.n_times] - obs[:, :, : self.n_times - 1]
        if self.use_kernel and delta_obs.is_cuda:
            # one forward / two backward launches instead of the 8 + 15 framework launches below
            from vihds import ops

            lh, gh = self.local_heads, self.gcond_heads
            q_all = ops.EncoderQTables.apply(
                self._kernel_shape(B, data), delta_obs, data.inputs, data.dev_1hot, self.conditional.conv.weight,
                self.conditional.conv.bias, self.conditional.lin.weight, self.conditional.lin.bias,
                None if lh is None else lh.weight, None if lh is None else lh.bias,
                None if gh is None else gh.weight, self.global_free, self.const_values)
            q = ChainedDistribution(name="q")
            q.attach_packed(self.kind, q_all, self.names, lambda chain: self._build_members(chain, q_all), self.q_rows)
            return q
        encoded = self.conditional(delta_obs)
        local_t = gcond_t = None
        if self.local:
            x = [encoded] + ([data.inputs] if self.l_tr else []) + ([data.dev_1hot] if self.l_dv else [])
            local_t = self.local_heads(torch.cat(x, 1))
        if self.gcond:
            x = ([data.inputs] if self.g_tr else []) + ([data.dev_1hot] if self.g_dv else [])
            gcond_t = self.gcond_heads(torch.cat(x, 1) if len(x) > 1 else x[0])
        P = len(self.descs)
        q_all = _PackQTables.apply(local_t, gcond_t, self.global_free, self.const_values, B, self.const_zeros)  # [2P, B]
        q = ChainedDistribution(name="q")
        q.attach_packed(self.kind, q_all, self.names, lambda chain: self._build_members(chain, q_all), self.q_rows)
        return q

    def _build_members(self, q, q_all):
        """The per-parameter TfNormal / TfLogNormal / TfConstant views (reference encoders.py:143-169, 187-253); built
        on first use only (evaluation, summaries, the reference-style call sequence)."""
        P = len(self.descs)
        rows = self.q_rows.long()
        q_mu, q_lp = q_all[rows[:P]], q_all[rows[P:]]
        q_prec = q_lp.exp()
        n_lg = len(self.local) + len(self.gcond)
        for i, d in enumerate(self.descs):
            if d.kind == CONSTANT:
                dist = TfConstant(value=self.const_values[i - n_lg - len(self.glob)].reshape(1))
            else:
                dist = CLASS_OF_KIND[d.kind](wait_for_assigned=True, variable=i < n_lg)
                if i < n_lg:  # one value per data row: [B,1] like the reference's Linear(n,1) output
                    dist.assign_free_and_constrained(q_mu[i][:, None], q_lp[i][:, None], q_prec[i][:, None])
                else:  # a single global value: shape [1]
                    dist.assign_free_and_constrained(q_mu[i][:1], q_lp[i][:1], q_prec[i][:1])
            q.add_distribution(d.name, dist)
        q._image = (self.kind, q_mu, q_prec)

    def forward(self, data):
        return self.evaluate_q(data)
