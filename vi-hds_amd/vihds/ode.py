"""OdeModel: the plugin base class of models.LOOKUP (counterpart of the reference's vihds/ode.py).

`simulate` keeps the reference signature and return value ([B,S,N,T], a strided view) but runs the whole time
loop -- and, fused in the same kernel, observe + the Gaussian log-likelihood -- as ONE HIP kernel
(vihds_ode_fwd) with a hand-written discrete adjoint (vihds_ode_bwd).  There is no torchdiffeq and no CPU path.
"""
import numpy as np
import math

import torch
import torch.nn as nn

from vihds import ops
from vihds.utils import default_get_value


class DecodedSolution(object):
    """What one fused kernel launch produced for a batch: views in the reference's layouts."""

    def __init__(self, traj, xpred, logp, observe=None):
        self.traj_buffer, self._xpred, self.logp_buffer = traj, xpred, logp  # [T,N,B,S], [T,4,B,S], [4,B,S]
        self.sol = traj.permute(2, 3, 1, 0)          # [B,S,N,T]  (reference ode.py:82)
        self.log_p_by_species = logp.permute(1, 2, 0)  # [B,S,4] (reference training.py:24-33)
        self._observe = observe  # xpred None (params.lazy_x_predict): the map that forms it from `sol` if somebody asks

    @property
    def has_x_predict(self):
        return self._xpred is not None

    @property
    def xpred_buffer(self):
        if self._xpred is None:
            self._xpred = self._observe(self.sol).permute(3, 2, 0, 1).contiguous()
        return self._xpred

    @property
    def x_predict(self):
        return self.xpred_buffer.permute(2, 3, 1, 0)   # [B,S,4,T]  (reference ode.py:84-93)


class LazySolution(object):
    """What the fused training kernel produced: the per-species log-likelihood only.  Trajectory and x_predict are
    computed (by the ordinary forward kernel, as a fresh autograd node on the same theta) if and when somebody asks."""

    def __init__(self, logp, materialise):
        self.logp_buffer = logp
        self.log_p_by_species = logp.permute(1, 2, 0)
        self.has_logp = True
        self._materialise, self._full = materialise, None

    def full(self):
        if self._full is None:
            self._full = self._materialise()
        return self._full

    traj_buffer = property(lambda self: self.full().traj_buffer)
    xpred_buffer = property(lambda self: self.full().xpred_buffer)
    sol = property(lambda self: self.full().sol)
    x_predict = property(lambda self: self.full().x_predict)


class DeviceConditioner(nn.Module):
    """Linear(D,1) -> ReLU with weights ~ N(2, 1.5) (reference ode.py:99-116)."""

    def __init__(self, n_inputs, use_bias=False, activation="relu"):
        super(DeviceConditioner, self).__init__()
        self.cond = nn.Linear(n_inputs, 1, use_bias)
        nn.init.xavier_uniform_(self.cond.weight)
        nn.init.normal_(self.cond.weight, mean=2.0, std=1.5)
        self.act = nn.ReLU()

    def forward(self, x):
        return self.act(self.cond(x))


class OdeModel(nn.Module):
    """Plugin interface (reference ode.py:28-96): subclasses set `species`, `n_species`, `precisions`,
    `model_key` (the models.LOOKUP key = the kernel to run) and may override condition_theta / observe."""

    model_key = None
    observe_kind = "default"

    def __init__(self, config):
        super(OdeModel, self).__init__()
        self.device_depth = config.data.device_depth
        self.n_treatments = len(config.data.conditions)
        self.use_laplace = default_get_value(config.params, "use_laplace", False, verbose=True)
        if self.use_laplace:
            raise NotImplementedError("use_laplace: the reference's Laplace log-prob (training.py:36-38) cannot "
                                      "run (torch.log of a float); only the Gaussian likelihood is implemented")
        self.precisions = None
        self.species = None
        self.n_species = None
        self.relevance = config.data.relevance_vectors
        self.default_devices = config.data.default_devices
        self.device = config.device
        self.conditioner_rng = default_get_value(config.params, "conditioner_rng", "cpu")
        self._spec_cache = {}
        self._tile_index = {}
        self._relevance_dev = {}
        self._rng_state = None
        self._fused_unsupported = {}
        self._last = None

    # ---- device conditioning (reference ode.py:43-58) ----------------------------------------------
    def device_conditioner(self, param, param_name, dev_1hot, use_bias=False, activation="relu"):
        """NB reference behaviour kept: a NEW DeviceConditioner with fresh N(2,1.5) weights on every call
         -- the weights are not trained."""
        n_batch, n_iwae = param.shape[0], param.shape[1]
        n_inputs = dev_1hot.shape[1]
        if self.conditioner_rng in ("device", "kernel") and dev_1hot.is_cuda:
            weight = 2.0 + 1.5 * torch.randn((1, n_inputs), device=dev_1hot.device)
        else:
            weight = DeviceConditioner(n_inputs, use_bias=use_bias, activation=activation).cond.weight.detach()
            weight = weight.to(dev_1hot.device)
        rkey = (param_name, str(dev_1hot.device))
        if rkey not in self._relevance_dev:  # uploaded once: no host->device copy inside a captured step
            self._relevance_dev[rkey] = torch.as_tensor(self.relevance[param_name], device=dev_1hot.device)
        rel = self._relevance_dev[rkey]
        cond = torch.relu(torch.nn.functional.linear(dev_1hot * rel, weight)).reshape(-1)  # [B]
        # Reference quirk kept on purpose (ode.py:46,52-57): `param` is flattened row-major (k = b*S+s) but the
        # conditioner output is tiled with .repeat([S,1]) (k -> cond[k mod B]), so entry (b,s) is scaled by the
        # value of row (b*S+s) mod B, not row b.
        window = getattr(param, "_sample_window", None) or (n_iwae, 0)  # (S_total, s_offset) under S-sharding
        key = (n_batch, n_iwae, window, str(dev_1hot.device))
        if key not in self._tile_index:
            b = torch.arange(n_batch, device=dev_1hot.device)[:, None]
            sidx = torch.arange(n_iwae, device=dev_1hot.device)[None, :]
            self._tile_index[key] = (b * window[0] + window[1] + sidx) % n_batch
        param_cond = cond[self._tile_index[key]]
        if param_name in self.default_devices:
            return param * (1.0 + param_cond)
        return param * param_cond

    # names of attributes condition_theta adds to theta (rows reserved behind the sampled parameters)
    extra_theta_names = ()

    def conditioner_job(self, names, dev_1hot):
        """(relevance [E,D], is_default [E], z or None, w_mean, w_std, rng_state or None) for the conditioner kernel:
        a fresh N(2, 1.5) weight draw per call as in the reference (ode.py:48), on the stream `conditioner_rng` names."""
        dev = dev_1hot.device
        key = (tuple(names), str(dev))
        if key not in self._relevance_dev:
            rel = torch.tensor(np.stack([self.relevance[n] for n in names]), dtype=torch.float32, device=dev)
            dflt = torch.tensor([1 if n in self.default_devices else 0 for n in names], dtype=torch.int32, device=dev)
            self._relevance_dev[key] = (rel, dflt)
        rel, dflt = self._relevance_dev[key]
        D = dev_1hot.shape[1]
        rng_state = None
        if self.conditioner_rng == "kernel":  # drawn inside the kernel (no launch for the draw, graph-capturable)
            if self._rng_state is None:
                self._rng_state = ops.KernelNormal.new_state(int(torch.randint(0, 2 ** 62, (1,)).item()), dev)
            z, mean, std, rng_state = None, 2.0, 1.5, self._rng_state
        elif self.conditioner_rng == "device":
            z, mean, std = torch.randn((len(names), D), device=dev), 2.0, 1.5
        else:  # the reference's stream: a fresh DeviceConditioner per name, drawn on the host
            from vihds import hostdraws

            def draw(host=None, k=len(names)):
                # what constructing k DeviceConditioner(D) modules draws from torch's CPU generator, without the modules:
                # nn.Linear's own initialisation and xavier_uniform_ (a uniform_ of the weight each: the stream advances by
                # the same amount whatever the bounds), then the N(2, 1.5) weights that are kept
                w = torch.empty((k, D)) if host is None else torch.from_numpy(host).view(k, D)
                for row in w:
                    row = row.view(1, D)
                    row.uniform_()
                    row.uniform_()
                    row.normal_(mean=2.0, std=1.5)
                return w

            if hostdraws.capturing():  # a captured step: drawn before every replay (vihds/hostdraws.py)
                z = hostdraws.ACTIVE.add((len(names), D), dev, draw)
            else:
                hostdraws.note((len(names), D))
                z = draw().to(dev)
            mean, std = 0.0, 1.0
        return rel, dflt, z, mean, std, rng_state

    def condition_ones(self, theta, names, dev_1hot):
        """theta.<name> = device_conditioner(ones, name, dev_1hot) for several names in ONE kernel launch, written
        straight into rows reserved in theta's packed buffer (falls back to the op-by-op path when theta has no
        reserved rows, e.g. when it was built by hand)."""
        n_batch, n_iwae = theta.get_n_batch(), theta.get_n_samples()
        base = len(theta.samples)
        if theta.n_reserved_rows() < len(names) or not dev_1hot.is_cuda:
            ones = torch.ones((n_batch, n_iwae), device=dev_1hot.device)
            for n in names:
                setattr(theta, n, self.device_conditioner(ones, n, dev_1hot))
            return theta
        rel, dflt, z, mean, std, rng_state = self.conditioner_job(names, dev_1hot)
        with torch.no_grad():
            ops.device_condition(z, dev_1hot, rel, dflt, theta._packed[base: base + len(names)], mean, std,
                                 rng_state, getattr(theta, "_sample_window", None))
        for j, n in enumerate(names):
            theta.bind_reserved_row(n, base + j)
        return theta

    def condition_theta(self, theta, dev_1hot, writer, epoch):
        raise NotImplementedError("TODO: write your condition_theta")

    # ---- kernel problem description -----------------------------------------------------------------
    def neural_weights(self):
        """Flat weight buffer for models with neural blocks (None for white-box models)."""
        return None

    def flat_weight_tensors(self):
        """The nn.Parameters behind neural_weights(), in the flat buffer's order ([] for white-box models): what
        ops.GeneralTail updates in the step's last launch."""
        prec = self.precisions
        return list(prec.weight_tensors()) if getattr(prec, "dynamic", False) else []

    def problem_kwargs(self, config):
        return {}

    def kernel_slots(self):
        """Parameter names in the order the kernel reads them (the library's table; dr_blackbox builds its own from
        n_z / n_x / n_y)."""
        import vihds.hip as hip

        return hip.model_slots(self.model_key)

    def _spec(self, config, row_of, n_rows):
        key = (config.params.solver, n_rows, tuple(sorted(row_of.items())))
        if key not in self._spec_cache:
            # params.adjoint_solver (reference ode.py:80: torchdiffeq.odeint_adjoint, the continuous adjoint) selects
            # nothing here: every solver differentiates through the discrete adjoint of its own accepted steps, which is
            # the gradient of what was actually computed
            kw = dict(self.problem_kwargs(config))
            # params.kernel_variant: 0 = the library's choice, 1 thread-per-trajectory, 2 lane-split, 3 time-parallel
            kw.setdefault("kernel_variant", int(default_get_value(config.params, "kernel_variant", 0)))
            self._spec_cache[key] = ops.OdeProblemSpec(self.model_key, config.params.solver, row_of, n_rows,
                                                       C=self.n_treatments, D=self.device_depth, **kw)
        return self._spec_cache[key]

    def solve(self, config, times, theta, conditions, dev_1hot, observations=None):
        """One fused launch: trajectory, observed signals and (if observations are given) the per-species
        log-likelihood.  Returns a DecodedSolution."""
        import vihds.hip as hip

        slots = self.kernel_slots()
        packed, row_of = theta.pack(slots)
        spec = self._spec(config, row_of, packed.shape[0])
        dev = packed.device
        times = times.to(dev)
        if config.params.solver in hip.ADAPTIVE_SOLVERS:
            return self._solve_adaptive(config, spec, packed, row_of, times, theta, conditions, dev_1hot, observations)
        obs = observations
        if obs is None:  # likelihood not requested: feed zeros (logp output is then meaningless and unused)
            obs = torch.zeros((packed.shape[1], 4, times.shape[0]), device=dev)
        row_offset, row_offset_map = getattr(theta, "_row_offset", None) or (None, None)
        # evaluation passes (no graph to differentiate) leave x_predict to whoever asks for it: Training.cost's summaries
        # form it inside their kernel, plugin code reading DecoderResult gets it from the map below
        # (... and so do training steps whose backward is ops.GeneralTail: nobody reads x_predict there)
        lazy = ((not torch.is_grad_enabled() or getattr(self, "_train_without_x_predict", False))
                and bool(default_get_value(config.params, "lazy_x_predict", True)))
        traj, xpred, logp = ops.OdeSolveObserve.apply(spec, packed, conditions.to(dev), times, obs.to(dev),
                                                      dev_1hot.to(dev) if dev_1hot is not None else None,
                                                      self.neural_weights(), row_offset, row_offset_map, not lazy)
        self._last = DecodedSolution(traj, xpred, logp, observe=lambda sol: self._observe_map(sol))
        self._last.has_logp = observations is not None
        return self._last

    def _solve_adaptive(self, config, spec, packed, row_of, times, theta, conditions, dev_1hot, observations):
        """torchdiffeq's adaptive pairs (reference ode.py:79-81; dopri5 / bosh3 / adaptive_heun): the controller picks ONE
        accepted grid for the batch (ops.adaptive_grid), the ordinary kernels integrate on it with the pair's
        higher-order tableau, the rows of the output times are gathered, and the log-likelihood (observations exist at
        the output times only) is evaluated with torch ops on the gathered x_predict -- autograd then hands the adjoint
        kernel upstream gradients for the trajectory / x_predict on the accepted grid."""
        dev = packed.device
        cond = conditions.to(dev)
        d1 = dev_1hot.to(dev) if dev_1hot is not None else None
        weights = self.neural_weights()
        rtol = float(default_get_value(config.params, "solver_rtol", 1e-7))  # torchdiffeq.odeint defaults
        atol = float(default_get_value(config.params, "solver_atol", 1e-9))
        # The accepted grid lives in a caller-sized buffer: params.solver_max_grid (default 4096 points); when the controller
        # runs out of it the buffer is quadrupled up to 2^17 points before giving up with a message that names the cause --
        # all arithmetic is fp32, so tolerances near its epsilon (torchdiffeq's defaults 1e-7 / 1e-9 with a low-order pair
        # such as adaptive_heun) ask for step sizes the state cannot resolve
        max_grid = int(default_get_value(config.params, "solver_max_grid", 4096))
        # torchdiffeq's own algorithm, resident on the device (round 4): steps run past the output times, outputs come from
        # the accepted step's quartic interpolant, the adjoint walks the logged accepted steps -- models without shared
        # neural weights; params.adaptive_device: false keeps the clipped-grid controller below for them too
        B, S = packed.shape[1], packed.shape[2]
        # (round 5: also the *_precisions models without a hidden layer -- the library says which problems it takes)
        if (bool(default_get_value(config.params, "adaptive_device", True))
                and ops.adaptive_device_supported(spec, B, S, int(times.shape[0]), max_grid) is not None):
            return self._solve_adaptive_device(config, spec, packed, row_of, times, cond, d1, observations, rtol, atol, max_grid,
                                               weights)
        while True:
            try:
                grid, index = ops.adaptive_grid(spec, packed, cond, times, d1, weights, rtol, atol, max_grid=max_grid)
                break
            except RuntimeError as e:
                # only the controller running out of its buffer is worth a larger one (the C ABI's message for that exact
                # condition); step-size underflow, a non-finite error estimate or bad arguments are re-raised as they are
                if "does not fit max_grid" not in str(e):
                    raise
                if max_grid >= (1 << 17):
                    raise RuntimeError(
                        "solver %r with rtol=%g, atol=%g needs more than %d accepted steps on this batch: in fp32 a relative "
                        "tolerance below ~1e-6 is at rounding level for a low-order pair -- raise params.solver_rtol / "
                        "solver_atol (or params.solver_max_grid)" % (config.params.solver, rtol, atol, max_grid)) from e
                max_grid *= 4
        self.last_adaptive_grid = grid
        dummy = torch.zeros((packed.shape[1], 4, grid.shape[0]), device=dev)
        row_offset, row_offset_map = getattr(theta, "_row_offset", None) or (None, None)
        traj_g, xpred_g, _ = ops.OdeSolveObserve.apply(spec, packed, cond, grid, dummy, d1, weights, row_offset,
                                                       row_offset_map)
        traj, xpred = traj_g.index_select(0, index), xpred_g.index_select(0, index)  # [T,N,B,S], [T,4,B,S]
        if observations is not None:
            if spec.n_species < spec.n_states:  # neural precisions: the last four states (reference precisions.py:89-94)
                prec = traj[:, spec.n_species:, :, :]
            else:  # constant precisions: four theta rows broadcast over time (reference precisions.py:31-35)
                rows = [row_of[n] for n in spec.slots[-4:]]
                prec = packed[rows][None]
            err = xpred - observations.to(dev).permute(2, 1, 0)[:, :, :, None]
            logp = (-0.5 * (math.log(2 * math.pi) - torch.log(prec) + prec * err * err)).sum(0)  # training.py:24-44
        else:
            logp = torch.zeros((4,) + tuple(packed.shape[1:]), device=dev)
        self._last = DecodedSolution(traj, xpred, logp)
        self._last.has_logp = observations is not None
        return self._last

    def _solve_adaptive_device(self, config, spec, packed, row_of, times, cond, d1, observations, rtol, atol, max_steps,
                               weights=None):
        """ops.AdaptiveOdeSolve + the observation map and the Gaussian log-likelihood with torch ops on the solution at the
        output times (autograd hands the adjoint kernel the upstream gradient of the trajectory)."""
        check = bool(default_get_value(config.params, "adaptive_check", True))  # False: no synchronisation (capturable)
        stats = [0, 0, 0]
        while True:
            try:
                traj = ops.AdaptiveOdeSolve.apply(spec, packed, cond, times, d1, rtol, atol, max_steps, check, stats, weights)
                break
            except ops.GridOverflow as e:
                if max_steps >= (1 << 17):
                    raise RuntimeError(
                        "solver %r with rtol=%g, atol=%g needs more than %d accepted steps on this batch: in fp32 a relative "
                        "tolerance below ~1e-6 is at rounding level for a low-order pair -- raise params.solver_rtol / "
                        "solver_atol (or params.solver_max_grid)" % (config.params.solver, rtol, atol, max_steps)) from e
                max_steps *= 4
        self.last_adaptive_stats = {"accepted": stats[1], "rejected": stats[2]}
        self.last_adaptive_grid = None
        sol = traj.permute(2, 3, 1, 0)                               # [B,S,N,T]
        xpred = self._observe_map(sol).permute(3, 2, 0, 1)           # [T,4,B,S]
        dev = packed.device
        if observations is not None:
            if spec.n_species < spec.n_states:  # neural precisions: the last four states (reference precisions.py:89-94)
                prec = traj[:, spec.n_species:, :, :]
            else:
                rows = [row_of[n] for n in spec.slots[-4:]]          # constant precisions (reference precisions.py:31-35)
                prec = packed[rows][None]
            err = xpred - observations.to(dev).permute(2, 1, 0)[:, :, :, None]
            logp = (-0.5 * (math.log(2 * math.pi) - torch.log(prec) + prec * err * err)).sum(0)  # training.py:24-44
        else:
            logp = torch.zeros((4,) + tuple(packed.shape[1:]), device=dev)
        self._last = DecodedSolution(traj, xpred.contiguous(), logp)
        self._last.has_logp = observations is not None
        return self._last

    fused_training_keys = ("dr_constant", "dr_constant_v2")

    def solve_for_training(self, config, times, theta, conditions, dev_1hot, observations):
        """Training fast path (params.fused_ode_training): ONE launch gives the log-likelihood and the unit-weight
        adjoint (ops.OdeLogLikFused); returns a LazySolution, or None when the path does not apply (then use solve)."""
        import vihds.hip as hip

        if observations is not None and not torch.is_grad_enabled():
            return self._solve_for_evaluation(config, times, theta, conditions, dev_1hot, observations)
        if (observations is None or self.model_key not in self.fused_training_keys or not torch.is_grad_enabled()
                or not default_get_value(config.params, "fused_ode_training", True)
                or config.params.solver in hip.ADAPTIVE_SOLVERS):
            return None
        slots = self.kernel_slots()
        packed, row_of = theta.pack(slots)
        if not packed.is_cuda or not packed.requires_grad:
            return None
        spec = self._spec(config, row_of, packed.shape[0])
        key = (packed.shape[1], packed.shape[2], times.shape[0], config.params.solver)
        if self._fused_unsupported.get(key):
            return None
        dev = packed.device
        args = (spec, packed, conditions.to(dev), times.to(dev), observations.to(dev),
                dev_1hot.to(dev) if dev_1hot is not None else None)
        try:
            logp = ops.OdeLogLikFused.apply(*args)
        except ops.FusedTrainingUnsupported:
            self._fused_unsupported[key] = True
            return None
        self._last = LazySolution(logp, lambda: self.solve(config, times, theta, conditions, dev_1hot, observations))
        return self._last

    def _solve_for_evaluation(self, config, times, theta, conditions, dev_1hot, observations):
        """Evaluation without the trajectory's round trip (params.online_summaries, default on): the forward launch writes
        the log-likelihoods ONLY, and Results' importance-weighted summaries come from a second forward launch that adds
        them up on the way (ops.ode_fwd_summaries) -- the trajectory (644 MB at 234 rows x 1 000 samples) is neither
        allocated, written nor read back: 1.33 GB -> 0.12 GB of HBM traffic for the two launches, at the same time per pass
        (0.68 ms either way: the integration is VALU-bound, the second one costs what streaming the trajectory back did;
        DESIGN.md section 1; profiles/LOG.md, round 5).  `online_summaries: false` keeps the stored form.
        Returns a LazySolution carrying `online_summaries(log_w, lse)`; trajectory and x_predict are computed (the ordinary
        forward launch) only if somebody asks.  None where the path does not apply (dr_blackbox, the adaptive solvers,
        launches below the evaluation size, which run other kernel families): then use solve."""
        import vihds.hip as hip

        # (opt-in per call: Training.evaluate sets `_evaluating` around its passes.  Any other forward under no_grad -- plots, xval
        # scripts, plugin code that reads x_states / x_predict -- keeps the stored trajectory: for it the lazy solution's full
        # solve would be a THIRD integration)
        if (not getattr(self, "_evaluating", False)
                or not default_get_value(config.params, "online_summaries", True) or not default_get_value(config.params, "lazy_x_predict", True)
                or config.params.solver in hip.ADAPTIVE_SOLVERS or getattr(theta, "_row_offset", None)
                or getattr(self, "_no_online_summaries", False)):  # (Training: samples sharded over ranks)
            return None
        packed, row_of = theta.pack(self.kernel_slots())
        if not packed.is_cuda:
            return None
        spec = self._spec(config, row_of, packed.shape[0])
        if not ops.ode_fwd_summaries_supported(spec, packed.shape[1], packed.shape[2], times.shape[0]):
            return None
        dev = packed.device
        args = (packed, conditions.to(dev), times.to(dev), dev_1hot.to(dev) if dev_1hot is not None else None,
                self.neural_weights())
        logp = ops.ode_logp_only(spec, args[0], args[1], args[2], observations.to(dev), args[3], args[4])
        self._last = LazySolution(logp, lambda: self.solve(config, times, theta, conditions, dev_1hot, observations))
        self._last.online_summaries = lambda log_w, lse: ops.ode_fwd_summaries(spec, args[0], args[1], args[2], args[3],
                                                                               args[4], log_w, lse)
        return self._last

    # ---- reference entry points ---------------------------------------------------------------------
    def simulate(self, config, times, theta, conditions, dev_1hot, condition_on_device=True, observations=None):
        """reference ode.py:66-82 -> [B,S,N,T]."""
        return self.solve(config, times, theta, conditions, dev_1hot, observations).sol

    def expand_precisions(self, theta, times, x_states):
        return self.precisions.expand(theta, len(times), x_states)

    def observe(self, x_sample, _theta):
        """reference ode.py:84-93.  When x_sample is (a view of) the solution just simulated, the fused
        kernel's x_predict is returned; otherwise the observation map is evaluated with torch ops."""
        last = self._last
        if last is not None and x_sample.data_ptr() == last.sol.data_ptr() and x_sample.shape[3] == last.sol.shape[3]:
            return last.x_predict
        return self._observe_map(x_sample)

    def _observe_map(self, x_sample):
        x0 = x_sample[:, :, 0, :]
        if self.observe_kind == "default":
            xp = [x0, x0 * x_sample[:, :, 1, :], x0 * (x_sample[:, :, 2, :] + x_sample[:, :, 4, :]),
                  x0 * (x_sample[:, :, 3, :] + x_sample[:, :, 5, :])]
        elif self.observe_kind == "inducer":  # models/inducer_constant.py:106-114
            xp = [x0, x0 * x_sample[:, :, 1, :], x0 * (x_sample[:, :, 2, :] + x_sample[:, :, 3, :]),
                  x0 * x_sample[:, :, 4, :]]
        else:
            xp = [x0, x0 * x_sample[:, :, 1, :], x0 * x_sample[:, :, 2, :], x0 * x_sample[:, :, 3, :]]
        return torch.stack(xp, dim=-1).permute(0, 1, 3, 2)

    def summaries(self, writer, epoch):
        pass
