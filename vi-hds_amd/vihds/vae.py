"""BaseVAE glue (counterpart of the reference's vihds/vae.py): sample u -> encoder -> q.sample -> p.clip ->
decoder.  The sample/clip/log-prob chain is one kernel; the decoder is one kernel."""
import numpy as np
import torch
import torch.nn as nn

from vihds.decoders import Decoder
from vihds.encoders import Encoder
from vihds.utils import default_get_value


class BaseVAE(nn.Module):
    def __init__(self, encoder, decoder, device, u_rng="numpy", shard=None):
        super(BaseVAE, self).__init__()
        self.encoder = encoder
        self.decoder = decoder
        self.device = device
        self.n_theta = None
        self.u_rng = u_rng
        self.shard = shard  # vihds.parallel.SampleShard or None
        self._rng_state = None
        self._u_staging = {}
        self._fused_declined = {}

    def sample_u(self, n_batch, n_samples, device=None):
        """Standard-normal draws u [B,S,P].  "numpy" = the reference's host RNG stream (vae.py:22-24);
        "device" = torch's Philox generator on the GPU (graph-capturable, no host->device copy);
        "kernel" = drawn inside the theta kernel (counter-based Philox, vihds_theta_opts.rng): no launch of its own,
        and under S-sharding each rank draws only its slice of the same global stream."""
        if self.u_rng == "device":
            return torch.randn((n_batch, n_samples, self.n_theta), device=self.device)
        if self.u_rng == "kernel":
            from vihds import ops

            if self._rng_state is None:  # seeded once from torch's generator so torch.manual_seed controls it
                seed = int(torch.randint(0, 2 ** 62, (1,)).item())
                self._rng_state = ops.KernelNormal.new_state(seed, self.device)
            return ops.KernelNormal((n_batch, n_samples, self.n_theta), self._rng_state)
        # The reference's stream (np.random.randn of the global RandomState), bit for bit, from native code (vihds/nprand.py:
        # 2.3 ms of numpy per draw at B=36, S=200 otherwise), straight into one of a few pinned staging buffers -- used in
        # turn, each reused only after the copy that read it has run -- and on to the device without waiting.
        from vihds import hostdraws, nprand

        shape = (int(n_batch), int(n_samples), int(self.n_theta))
        if hostdraws.capturing():  # a captured step: the draw happens before every replay (vihds/hostdraws.py)
            return hostdraws.ACTIVE.add(shape, self.device, nprand.Draw(shape), prefetchable=True)
        hostdraws.note(shape)
        if not (torch.cuda.is_available() and torch.device(self.device).type == "cuda"):
            return torch.from_numpy(nprand.randn_f32(shape)).to(self.device)
        n = shape[0] * shape[1] * shape[2]
        ring = self._u_staging.get(n)
        if ring is None:
            ring = self._u_staging[n] = {"bufs": [torch.empty(n, dtype=torch.float32).pin_memory() for _ in range(4)],
                                         "events": [None] * 4, "pos": 0}
        k = ring["pos"]
        ring["pos"] = (k + 1) % 4
        if ring["events"][k] is not None:
            ring["events"][k].synchronize()
        host = ring["bufs"][k]
        nprand.randn_f32(shape, out=host.numpy())
        u = host.view(shape).to(self.device, non_blocking=True)
        if ring["events"][k] is None:
            ring["events"][k] = torch.cuda.Event()
        ring["events"][k].record()
        return u

    def forward(self, data, samples, writer=None, epoch=None):
        u = self.sample_u(len(data.inputs), samples)
        if self.shard is not None:
            u = self.shard.take(u)  # every rank drew the same full u; keep this rank's slice of S
        q = self.encoder(data)
        p = self.encoder.p
        ode_model = self.decoder.ode_model
        n_extra = len(ode_model.extra_theta_names) if self.decoder.condition_on_device else 0
        fused = self._decoder_step_fused(data, samples, u, q, p, n_extra)
        if fused is not None:
            return fused + (q, p)
        fused = self._theta_ode_fused(data, samples, u, q, p, n_extra)
        if fused is not None:
            return fused + (q, p)
        clipped_theta = q.sample_clip_log_prob(u, p, stddevs=4, n_extra_rows=n_extra)
        if self.shard is not None:
            lo, _ = self.shard.bounds(samples)
            object.__setattr__(clipped_theta, "_sample_window", (samples, lo))
        result, conditioned_theta = self.decoder(clipped_theta, data, writer, epoch)
        return result, conditioned_theta, q, p


def _bind_fused(BaseVAE):
    def _decoder_step_fused(self, data, samples, u, q, p, n_extra):
        """Training fast path (params.fused_ode_training): sampling, device conditioning, log-likelihood and the
        unit-weight adjoint in ONE launch (ops.DecoderStepFused).  None when it does not apply."""
        from vihds import hip, ops
        from vihds.decoders import LazyDecoderResult
        from vihds.ode import LazySolution

        dec = self.decoder
        ode = dec.ode_model
        cfg = dec.config
        obs = data.get("observations", None) if hasattr(data, "get") else None
        if (not torch.is_grad_enabled() or obs is None or not default_get_value(cfg.params, "fused_ode_training", True)
                or not default_get_value(cfg.params, "fused_decoder_step", True) or ode.model_key not in ode.fused_training_keys or getattr(q, "_packed_q", None) is None
                or not obs.is_cuda or (n_extra and not dec.condition_on_device)
                or cfg.params.solver in hip.ADAPTIVE_SOLVERS):
            return None
        key = (tuple(obs.shape), samples, cfg.params.solver)
        if self._fused_declined.get(key):
            return None
        extra = list(ode.extra_theta_names) if n_extra else []
        window = None
        if self.shard is not None:
            lo, _ = self.shard.bounds(samples)
            window = (samples, lo)

        def spec_of(names):
            row_of = {n: k for k, n in enumerate(list(names) + extra)}
            return ode._spec(cfg, row_of, len(names) + len(extra))

        cond_job = None
        if extra:
            rel, dflt, z, mean, std, rng_state = ode.conditioner_job(extra, data.dev_1hot)
            cond_job = (len(extra), None, mean, std, z, rng_state, rel, dflt)
        try:
            n_q = len(q.names())
            if cond_job is not None:
                cond_job = (cond_job[0], n_q) + cond_job[2:]
            if window is not None and not isinstance(u, ops.KernelNormal):
                return None  # (the conditioner tiling needs the window; only the in-kernel draw carries it)
            theta, logp = q.decoder_step_fused(u, p, 4, n_extra, spec_of, data.inputs, data.times.to(obs.device), obs,
                                               data.dev_1hot, cond_job)
        except ops.FusedTrainingUnsupported:
            self._fused_declined[key] = True
            return None
        for k, n in enumerate(extra):
            theta.bind_reserved_row(n, n_q + k)
        if window is not None:
            object.__setattr__(theta, "_sample_window", window)
        if hasattr(ode, "aR") and extra:
            ode.aR, ode.aS = getattr(theta, "aR", None), getattr(theta, "aS", None)
        sol = LazySolution(logp, lambda: ode.solve(cfg, data.times, theta, data.inputs, data.dev_1hot, obs))
        # params.fused_iwae_backward: Training.cost may leave the IWAE loss to this step's theta-adjoint launch
        sol.defer_iwae = bool(default_get_value(cfg.params, "fused_iwae_backward", True)) and self.shard is None
        ode._last = sol

        def build():
            full = sol.full()
            xs, prec = ode.expand_precisions(theta, data.times, full.sol)
            return xs, ode.observe(full.sol, theta), prec

        result = LazyDecoderResult(build)
        result.solution = sol
        result.log_p_by_species = sol.log_p_by_species
        return result, theta

    BaseVAE._decoder_step_fused = _decoder_step_fused

    def _theta_ode_fused(self, data, samples, u, q, p, n_extra):
        """Training steps whose backward is ops.GeneralTail (Training.step sets `_fuse_theta_ode`): the sampling stage, for
        dr_blackbox also condition_theta, runs inside the ODE forward launch (ops.ThetaOdeFused, vihds_theta_ode_fwd) -- the
        models whose forward kernels carry it: relay / degrader / prpr / auto_constant (+ _precisions), dr_blackbox at the
        built-in sizes.  None when it does not apply (then the separate launches)."""
        from vihds import hip, ops
        from vihds.decoders import LazyDecoderResult
        from vihds.ode import DecodedSolution

        dec = self.decoder
        ode, cfg = dec.ode_model, dec.config
        obs = data.get("observations", None) if hasattr(data, "get") else None
        if (not getattr(self, "_fuse_theta_ode", False) or not torch.is_grad_enabled() or obs is None or not obs.is_cuda
                or getattr(q, "_packed_q", None) is None or self.shard is not None or cfg.params.solver in hip.ADAPTIVE_SOLVERS):
            return None
        blackbox = ode.model_key == "dr_blackbox"
        if not blackbox and (n_extra or getattr(ode, "extra_theta_names", ())):
            return None  # (a device conditioner: not a stage of these kernels)
        key = ("theta_ode", tuple(obs.shape), samples, cfg.params.solver)
        if self._fused_declined.get(key):
            return None
        extra = list(ode.extra_theta_names) if (blackbox and n_extra) else []
        n_q = len(q.names())
        offset = None
        if blackbox and getattr(ode, "n_y", 0) > 0:
            if not extra:
                return None
            names = q.names()
            rows = [names.index("y%d" % (i + 1)) if ("y%d" % (i + 1)) in names else -1 for i in range(ode.n_y)]
            if rows != list(range(rows[0], rows[0] + ode.n_y)) or rows[0] < 0:
                return None
            offset = (ode.offset_layer.weight, ode.offset_layer.bias, (rows[0], n_q, ode.n_y))

        def spec_of(names):
            row_of = {n: k for k, n in enumerate(list(names) + extra)}
            if blackbox:  # the integrator reads the device-conditioned rows (condition_theta re-binds y1.. to them)
                for i in range(ode.n_y):
                    row_of["y%d" % (i + 1)] = n_q + i
            return ode._spec(cfg, row_of, len(names) + len(extra))

        try:
            theta, traj, logp = q.theta_ode_fused(u, p, 4, len(extra), spec_of, data.inputs, data.times.to(obs.device), obs,
                                                  data.dev_1hot, ode.neural_weights(), offset)
        except ops.FusedTrainingUnsupported:
            self._fused_declined[key] = True
            return None
        if blackbox:
            for i in range(ode.n_y):  # (what condition_theta does: the attribute now names the conditioned row)
                theta.bind_reserved_row("y%d" % (i + 1), n_q + i)
        sol = DecodedSolution(traj, None, logp, observe=lambda s_: ode._observe_map(s_))
        sol.has_logp = True
        ode._last = sol

        def build():
            xs, prec = ode.expand_precisions(theta, data.times, sol.sol)
            return xs, ode.observe(sol.sol, theta), prec

        result = LazyDecoderResult(build)
        result.solution = sol
        result.log_p_by_species = sol.log_p_by_species
        return result, theta

    BaseVAE._theta_ode_fused = _theta_ode_fused


_bind_fused(BaseVAE)


def build_model(args, settings, dataset, parameters, shard=None):
    """reference vae.py:39-51.  Construction order (encoder, then decoder) matches the reference so that the
    same torch seed gives the same initial weights."""
    encoder = Encoder(parameters, dataset, getattr(args, "verbose", False), device=settings.device)
    if settings.data.device_depth > 1:
        decoder_condition_on_device = True
    else:
        print("- Only a single device being considered, so disabling device-conditioning in the decoder")
        decoder_condition_on_device = False
    decoder = Decoder(settings, decoder_condition_on_device).to(settings.device)
    return BaseVAE(encoder, decoder, settings.device, u_rng=default_get_value(settings.params, "u_rng", "numpy"),
                   shard=shard)
