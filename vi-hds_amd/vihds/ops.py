"""torch.autograd.Function wrappers around the HIP kernels (C ABI in include/vihds_hip.h).

PyTorch is plumbing here: it owns the device buffers and the stream the kernels are enqueued on, and its
autograd graph calls the hand-written backward kernels.  Nothing in this module computes on the CPU and there
is no fallback: tensors must live on a HIP device.
"""
import ctypes
import math

import torch

from vihds import hip


class KernelTimer(object):
    """HIP-event timing of individual kernel launches on the stream they are enqueued on (torch's current
    stream = the stream handed to the C ABI).  Enabled by bench.py for its roofline leg only."""

    def __init__(self):
        self.spans = {}

    def launch(self, name, fn):
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        rc = fn()
        e1.record()
        self.spans.setdefault(name, []).append((e0, e1))
        return rc

    def summary(self):
        torch.cuda.synchronize()
        out = {}
        for name, evs in self.spans.items():
            ms = [a.elapsed_time(b) for a, b in evs]
            out[name] = {"launches": len(ms), "mean_us": 1e3 * sum(ms) / len(ms), "min_us": 1e3 * min(ms)}
        return out


class LaunchRecorder(object):
    """Keeps the launch closure of every named ODE kernel launch (pointers bound, buffers kept alive by the
    closure) so that bench.py can re-issue exactly the step's own launch back to back for its roofline leg."""

    def __init__(self):
        self.calls = {}

    def launch(self, name, fn):
        self.calls[name] = fn
        return fn()


TIMER = None  # set to a KernelTimer / LaunchRecorder to time / record every ODE kernel launch


def _launch(name, fn):
    return TIMER.launch(name, fn) if TIMER is not None else fn()


def _require_cuda(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError("vihds HIP ops need tensors on the GPU (got a %s tensor); there is no CPU fallback"
                               % t.device)


def _c(t):
    return None if t is None else t.contiguous()


class OdeProblemSpec:
    """Host-side description of one decoder problem (struct vihds_ode_problem) for a fixed model/solver."""

    def __init__(self, model, solver, row_of, n_rows, C, D=0, n_hidden_prec=0, n_hidden_states=0,
                 n_latent_states=0, n_const=0, init_latent=0.001, init_prec=1e-5, kernel_variant=0, slots=None):
        if model not in hip.MODELS:
            raise KeyError("unknown model '%s'" % model)
        if solver not in hip.SOLVERS:
            raise NotImplementedError(
                "solver '%s' is not implemented by the HIP path (available: %s)" % (solver, ", ".join(sorted(hip.SOLVERS))))
        self.model, self.solver = model, solver
        sized = False
        if model == "dr_blackbox":
            # network sizes other than specs/dr_blackbox_icml.yaml: per-size kernels in a side library (built on first
            # use); the latent slot names depend on n_z / n_x / n_y, so the plugin hands them in (`slots`)
            sizes = (n_latent_states, n_hidden_states, n_hidden_prec, n_const - C - D)
            sized = sizes != hip.BLACKBOX_BUILTIN
            if sized:
                hip.ensure_blackbox_variant(*sizes)
                if slots is None or len(slots) != sizes[3] + 4:
                    raise KeyError("dr_blackbox at sizes %s: pass the %d slot names (latents z.., x.., y.. then "
                                   "init_x, init_rfp, init_yfp, init_cfp)" % (sizes, sizes[3] + 4))
        self.slots = list(slots) if (sized and slots is not None) else hip.model_slots(model)
        missing = [s for s in self.slots if s not in row_of]
        if missing:
            raise KeyError("model '%s' needs parameters %s which the spec does not define" % (model, missing))
        self.n_rows = n_rows
        self.proto = hip.OdeProblem()
        self.proto.model = hip.MODELS[model]
        self.proto.solver = hip.SOLVERS[solver]
        self.proto.C, self.proto.D, self.proto.n_rows = C, D, n_rows
        for q, s in enumerate(self.slots):
            self.proto.slot_row[q] = row_of[s]
        self.proto.n_hidden_prec = n_hidden_prec
        self.proto.n_hidden_states = n_hidden_states
        self.proto.n_latent_states = n_latent_states
        self.proto.n_const = n_const
        self.proto.init_latent = init_latent
        self.proto.init_prec = init_prec
        self.proto.kernel_variant = kernel_variant
        if sized:
            self.n_states = hip.lib().vihds_problem_n_states(ctypes.byref(self.proto))
            if self.n_states < 0:
                hip.check(self.n_states, "vihds_problem_n_states")
            self.n_species = self.n_states - 4
        else:
            self.n_states = hip.lib().vihds_model_n_states(hip.MODELS[model])
            self.n_species = hip.lib().vihds_model_n_species(hip.MODELS[model])
        self.covers_all_rows = len({row_of[s] for s in self.slots}) == n_rows
        self.unwritten_rows = sorted(set(range(n_rows)) - {row_of[s] for s in self.slots})  # rows the adjoint leaves alone
        self.cache = {}  # device-side constants derived from this spec

    def bind(self, B, S, T):
        p = hip.OdeProblem()
        ctypes.pointer(p)[0] = self.proto
        p.B, p.S, p.T = B, S, T
        return p


def adaptive_grid(spec, theta, cond, times, dev1hot, weights, rtol=1e-7, atol=1e-9, max_grid=4096):
    """Step-size controller of an adaptive solver (vihds_ode_adaptive_grid; synchronous, no gradient): the accepted
    time grid for the whole batch -- one step size for all trajectories, as torchdiffeq 0.1 does it -- and the positions
    of the output times in it.  Returns (grid [G] float32 tensor on the device, index [T] int64 tensor on the device)."""
    _require_cuda(theta, cond)
    theta, cond = _c(theta.detach()), _c(cond)
    R, B, S = theta.shape
    th = times.detach().to("cpu", torch.float32).contiguous()
    T = th.shape[0]
    prob = spec.bind(B, S, T)
    L = hip.lib()
    ws = torch.empty(int(L.vihds_ode_adaptive_workspace_floats(ctypes.byref(prob))), device=theta.device,
                     dtype=torch.float32)
    grid = torch.empty(max_grid, dtype=torch.float32)
    index = torch.empty(T, dtype=torch.int32)
    rc = L.vihds_ode_adaptive_grid(ctypes.byref(prob), hip.ptr(theta), hip.ptr(cond), hip.ptr(_c(dev1hot)),
                                   hip.ptr(_c(weights.detach()) if weights is not None else None), th.data_ptr(),
                                   float(rtol), float(atol), hip.ptr(ws), grid.data_ptr(), max_grid, index.data_ptr(),
                                   hip.current_stream())
    if rc < 0:
        hip.check(rc, "vihds_ode_adaptive_grid")
    return grid[:rc].to(theta.device), index.to(theta.device, torch.int64)


class GridOverflow(RuntimeError):
    """The adaptive controller accepted more steps than its buffer holds (raise solver_max_grid or the tolerances)."""


ADAPTIVE_DEVICE_ERRORS = {1: "the grid barrier timed out", 2: "more accepted steps than params.solver_max_grid",
                          3: "step-size underflow", 4: "non-finite error estimate"}


def adaptive_device_supported(spec, B, S, T, max_steps):
    """Floats of the tape vihds_ode_adaptive_fwd needs, or None when the device-resident solver does not serve this
    problem (shared neural weights, dopri8, more than 65 536 trajectories)."""
    n = hip.lib().vihds_ode_adaptive_tape_floats(ctypes.byref(spec.bind(B, S, T)), int(max_steps))
    return int(n) if n > 0 else None


class AdaptiveOdeSolve(torch.autograd.Function):
    """torchdiffeq 0.1's adaptive-step algorithm on the device (vihds_ode_adaptive_fwd_w / _bwd_w): one persistent launch
    forward, one launch for the discrete adjoint over the logged accepted steps.  forward(theta [R,B,S], cond, times [T],
    weights or None) -> traj [T,N,B,S] (the solution at the output times; steps run past them, outputs come from the accepted
    step's quartic interpolant).  `stats` (host ints [error, accepted, rejected]) is filled when check=True, which
    synchronises; with check=False nothing waits and the pair of launches can be captured in a hipGraph.  weights: the
    precision network of a *_precisions model (no hidden layer); its gradient comes back from the adjoint launch."""

    @staticmethod
    def forward(ctx, spec, theta, cond, times, dev1hot, rtol, atol, max_steps, check, stats, weights=None):
        _require_cuda(theta, cond, times)
        theta, cond, times = _c(theta), _c(cond), _c(times.to(torch.float32))
        R, B, S = theta.shape
        T = times.shape[0]
        prob = spec.bind(B, S, T)
        n_ws = hip.lib().vihds_ode_adaptive_tape_floats(ctypes.byref(prob), int(max_steps))
        if n_ws <= 0:
            hip.check(int(n_ws), "vihds_ode_adaptive_tape_floats")
        ws = torch.empty(int(n_ws), device=theta.device, dtype=torch.float32)
        traj = torch.empty((T, spec.n_states, B, S), device=theta.device, dtype=torch.float32)
        rc = _launch("ode_adaptive_fwd", lambda: hip.lib().vihds_ode_adaptive_fwd_w(
            ctypes.byref(prob), hip.ptr(theta), hip.ptr(cond), hip.ptr(dev1hot), hip.ptr(weights), hip.ptr(times), float(rtol),
            float(atol), int(max_steps), hip.ptr(ws), hip.ptr(traj), hip.current_stream()))
        hip.check(rc, "vihds_ode_adaptive_fwd")
        if check:
            err, acc, rej = (int(v) for v in ws[:4].view(torch.int32)[1:4].tolist())
            if stats is not None:
                stats[:] = [err, acc, rej]
            if err == 2:
                raise GridOverflow("adaptive solver '%s': more than %d accepted steps" % (spec.solver, max_steps))
            if err != 0:
                raise RuntimeError("adaptive solver '%s' failed on the device: %s" % (spec.solver, ADAPTIVE_DEVICE_ERRORS.get(err, err)))
        ctx.spec, ctx.prob, ctx.max_steps = spec, prob, int(max_steps)
        ctx.save_for_backward(theta, cond, times, dev1hot, ws, weights)
        ctx.set_materialize_grads(False)
        return traj

    @staticmethod
    def backward(ctx, g_traj):
        theta, cond, times, dev1hot, ws, weights = ctx.saved_tensors
        g_theta = torch.empty_like(theta) if ctx.spec.covers_all_rows else torch.zeros_like(theta)
        if g_traj is None:
            return (None, torch.zeros_like(theta)) + (None,) * 9
        g_traj = _c(g_traj)
        g_w = torch.zeros_like(weights) if weights is not None else None
        rc = _launch("ode_adaptive_bwd", lambda: hip.lib().vihds_ode_adaptive_bwd_w(
            ctypes.byref(ctx.prob), hip.ptr(theta), hip.ptr(cond), hip.ptr(dev1hot), hip.ptr(weights), hip.ptr(times),
            ctx.max_steps, hip.ptr(ws), hip.ptr(g_traj), hip.ptr(g_theta), hip.ptr(g_w), hip.current_stream()))
        hip.check(rc, "vihds_ode_adaptive_bwd")
        return (None, g_theta) + (None,) * 8 + (g_w,)


class OdeSolveObserve(torch.autograd.Function):
    """simulate + observe + Gaussian log-likelihood in one kernel; adjoint in one kernel.

    forward(theta[R,B,S], cond[B,C], times[T], obs[B,4,T]) ->
        traj  [T,N,B,S]  (the host hands out .permute(2,3,1,0) = the reference's [B,S,N,T] view, ode.py:82)
        xpred [T,4,B,S]
        logp  [4,B,S]
    Any subset of the three outputs may be used downstream; unused ones cost nothing in backward.

    row_offset [B,n] with row_offset_map = (src, dst, n) (optional): rows dst..dst+n-1 of theta already hold
    theta[src+i] + row_offset[:, i] (written in place by the caller, e.g. dr_blackbox's y + offset_layer(dev_1hot)) and
    the kernel reads those; backward then routes their gradient to rows src.. and, summed over S, to row_offset.
    row_offset_map = (src, dst, n, "linear"): the rows were written by OffsetRows (offset = Linear(D, n) of dev1hot) and
    row_offset is its token [n*D + n]; backward hands it the layer's weight and bias gradients from ONE launch
    (vihds_offset_rows_bwd) that also routes the row gradients.
    """

    @staticmethod
    def forward(ctx, spec, theta, cond, times, obs, dev1hot, weights, row_offset=None, row_offset_map=None,
                want_xpred=True):
        _require_cuda(theta, cond, times, obs)
        theta, cond, times, obs = _c(theta), _c(cond), _c(times), _c(obs)
        R, B, S = theta.shape
        T = times.shape[0]
        if R != spec.n_rows:
            raise RuntimeError("theta has %d rows, problem expects %d" % (R, spec.n_rows))
        prob = spec.bind(B, S, T)
        N = spec.n_states
        traj = torch.empty((T, N, B, S), device=theta.device, dtype=torch.float32)
        # (want_xpred False: the observed signals are not written -- they are a pointwise map of the trajectory and the
        # evaluation summaries form them in registers, vihds_iw_summaries_states; the second output is then None)
        xpred = torch.empty((T, 4, B, S), device=theta.device, dtype=torch.float32) if want_xpred else None
        logp = torch.empty((4, B, S), device=theta.device, dtype=torch.float32)
        rc = _launch("ode_fwd", lambda: hip.lib().vihds_ode_fwd(
            ctypes.byref(prob), hip.ptr(theta), hip.ptr(cond), hip.ptr(dev1hot), hip.ptr(times), hip.ptr(obs),
            hip.ptr(weights), hip.ptr(traj), hip.ptr(xpred), hip.ptr(logp), hip.current_stream()))
        hip.check(rc, "vihds_ode_fwd")
        ctx.spec, ctx.prob = spec, prob
        ctx.row_offset_map = row_offset_map if row_offset is not None else None
        ctx.logp_out = logp.detach()  # (GeneralTail reads the step's forward state off this node)
        ctx.save_for_backward(theta, cond, times, obs, traj, dev1hot, weights)
        ctx.set_materialize_grads(False)
        return traj, xpred, logp

    @staticmethod
    def backward(ctx, g_traj, g_xpred, g_logp):
        theta, cond, times, obs, traj, dev1hot, weights = ctx.saved_tensors
        # the kernel writes every slot row; rows that are not slots (if any) must read as zero -- unless they are exactly
        # the rows the "linear" row-offset launch below assigns (dr_blackbox: the sampled y rows)
        rom = ctx.row_offset_map
        assign_src = (rom is not None and len(rom) == 4
                      and ctx.spec.unwritten_rows == list(range(rom[0], rom[0] + rom[2])))
        g_theta = torch.empty_like(theta) if (ctx.spec.covers_all_rows or assign_src) else torch.zeros_like(theta)
        n_aux = hip.lib().vihds_ode_bwd_aux_floats(ctypes.byref(ctx.prob))
        blackbox = ctx.spec.model == "dr_blackbox"
        if not blackbox and not (weights is not None and ctx.needs_input_grad[6]):
            n_aux = 0  # neural precisions: the dump is only worth it when the weight gradients are wanted
        # (dr_blackbox: the weight gradients come from the dump below, the kernel does not touch g_weights)
        g_w = torch.zeros_like(weights) if weights is not None and not blackbox else None
        prob = ctx.prob
        if g_logp is not None and g_logp.dim() == 3 and g_logp.stride(0) == 0 and g_logp[0].is_contiguous():
            prob.logp_grad_broadcast = 1  # IwaeLoss hands back one [B,S] gradient for all four species: no copy
            g_logp = g_logp[0]
        else:
            prob.logp_grad_broadcast = 0
            g_logp = _c(g_logp)
        g_traj, g_xpred = _c(g_traj), _c(g_xpred)
        aux = torch.empty(n_aux, device=theta.device, dtype=torch.float32) if n_aux > 0 else None
        rc = _launch("ode_bwd", lambda: hip.lib().vihds_ode_bwd(
            ctypes.byref(ctx.prob), hip.ptr(theta), hip.ptr(cond), hip.ptr(dev1hot), hip.ptr(times), hip.ptr(obs),
            hip.ptr(weights), hip.ptr(traj), hip.ptr(g_traj), hip.ptr(g_xpred), hip.ptr(g_logp), hip.ptr(g_theta),
            hip.ptr(g_w), hip.ptr(aux), hip.current_stream()))
        hip.check(rc, "vihds_ode_bwd")
        if aux is not None and ctx.needs_input_grad[6] and not blackbox and hip.lib().vihds_ode_bwd_reduces_weights(
                ctypes.byref(ctx.prob)):
            pass  # (lane-split relay adjoint: the call above has already added the weight gradients up into g_w)
        elif aux is not None and ctx.needs_input_grad[6]:
            if blackbox:
                g_w = blackbox_weight_grads(ctx.spec, ctx.prob, aux, theta, cond, dev1hot)
            else:
                neural_precision_weight_grads(ctx.spec, ctx.prob, aux, g_w)
        grads = (None, g_theta, None, None, None, None, g_w)
        if len(ctx.needs_input_grad) > 7:
            g_off = None
            if ctx.row_offset_map is not None and len(ctx.row_offset_map) == 4:
                src, dst, n, _ = ctx.row_offset_map
                R, B, S = theta.shape
                D = dev1hot.shape[1]
                if ctx.needs_input_grad[7]:
                    g_off = torch.empty(n * D + n, device=theta.device, dtype=torch.float32)
                rc = hip.lib().vihds_offset_rows_bwd(B, S, D, n, R, src, dst, 0 if assign_src else 1, hip.ptr(dev1hot),
                                                     hip.ptr(g_theta), hip.ptr(g_off), hip.current_stream())
                hip.check(rc, "vihds_offset_rows_bwd")
            elif ctx.row_offset_map is not None:
                src, dst, n = ctx.row_offset_map
                if ctx.needs_input_grad[7]:
                    g_off = g_theta[dst:dst + n].sum(2).t()
                g_theta[src:src + n] += g_theta[dst:dst + n]
            grads = grads + (g_off, None, None)[:len(ctx.needs_input_grad) - 7]
        return grads


class OffsetRows(torch.autograd.Function):
    """dr_blackbox's condition_theta (reference models/dr_blackbox.py:86-96) in one launch: rows dst.. of the packed theta
    buffer <- rows src.. + Linear(D, n)(dev1hot), written in place (the buffer's autograd history is untouched: the rows
    are reserved scratch rows of it).  Returns a token [n*D + n] to hand to OdeSolveObserve as row_offset with
    row_offset_map = (src, dst, n, "linear"); the gradient that comes back for it is [g_weight | g_bias]."""

    @staticmethod
    def forward(ctx, weight, bias, dev1hot, packed, src, dst):
        _require_cuda(weight, bias, dev1hot, packed)
        n, D = weight.shape
        R, B, S = packed.shape
        rc = hip.lib().vihds_offset_rows_fwd(B, S, D, n, R, src, dst, hip.ptr(_c(weight)), hip.ptr(_c(bias)),
                                             hip.ptr(_c(dev1hot)), hip.ptr(packed), hip.current_stream())
        hip.check(rc, "vihds_offset_rows_fwd")
        ctx.shape = (n, D)
        ctx.set_materialize_grads(False)
        return torch.empty(n * D + n, device=packed.device, dtype=torch.float32)

    @staticmethod
    def backward(ctx, g):
        if g is None:
            return None, None, None, None, None, None
        n, D = ctx.shape
        return g[: n * D].view(n, D), g[n * D:], None, None, None, None


class _FlatParameterView(torch.autograd.Function):
    @staticmethod
    def forward(ctx, holder, *params):
        ctx.shapes = [p.shape for p in params]
        return holder.flat.detach()

    @staticmethod
    def backward(ctx, g):
        out, o = [], 0
        for sh in ctx.shapes:
            n = sh.numel()
            out.append(g[o:o + n].view(sh))
            o += n
        return (None,) + tuple(out)


class FlatParameters(object):
    """Keeps a fixed list of nn.Parameters as views of ONE flat device buffer, in the order the kernels expect the
    weights back to back: the kernel reads them where the optimizer updates them (no concatenation per step) and the
    flat weight gradient is handed to the parameters as views.  Re-aliases itself when a parameter's storage was
    replaced (module.to(), a new .data)."""

    def __init__(self):
        self.flat = None

    def _aliased(self, params):
        if self.flat is None or self.flat.device != params[0].device:
            return False
        at = self.flat.data_ptr()
        for p in params:
            if p.data_ptr() != at or not p.is_contiguous():
                return False
            at += 4 * p.numel()
        return at == self.flat.data_ptr() + 4 * self.flat.numel()

    def __call__(self, params):
        params = list(params)
        if not params[0].is_cuda:  # host-side inspection only; the kernels never see CPU tensors
            return torch.cat([p.reshape(-1) for p in params])
        if not self._aliased(params):
            with torch.no_grad():
                self.flat = torch.cat([p.detach().reshape(-1).float() for p in params])
                o = 0
                for p in params:
                    p.data = self.flat[o:o + p.numel()].view(p.shape)
                    o += p.numel()
        return _FlatParameterView.apply(self, *params)


class FusedTrainingUnsupported(RuntimeError):
    """vihds_ode_logp_grad declined this (model, shape): the caller uses OdeSolveObserve instead."""


class OdeLogLikFused(torch.autograd.Function):
    """Training fast path (vihds_ode_logp_grad): per-species log-likelihood [4,B,S] and, in the same launch, the theta
    gradient for a unit upstream gradient; neither trajectory nor x_predict is produced.  backward: the ELBO hands
    back ONE [B,S] weight for all four signals (a stride-0 expand, see IwaeLoss), and the adjoint is linear in it:
    g_theta = w * g_unit.  Any other upstream gradient falls back to the two-kernel path."""

    @staticmethod
    def forward(ctx, spec, theta, cond, times, obs, dev1hot):
        _require_cuda(theta, cond, times, obs)
        theta, cond, times, obs = _c(theta), _c(cond), _c(times), _c(obs)
        R, B, S = theta.shape
        T = times.shape[0]
        if R != spec.n_rows:
            raise RuntimeError("theta has %d rows, problem expects %d" % (R, spec.n_rows))
        prob = spec.bind(B, S, T)
        logp = torch.empty((4, B, S), device=theta.device, dtype=torch.float32)
        g_unit = torch.empty_like(theta) if spec.covers_all_rows else torch.zeros_like(theta)
        rc = _launch("ode_logp_grad", lambda: hip.lib().vihds_ode_logp_grad(
            ctypes.byref(prob), hip.ptr(theta), hip.ptr(cond), hip.ptr(dev1hot), hip.ptr(times), hip.ptr(obs),
            hip.ptr(logp), hip.ptr(g_unit), hip.current_stream()))
        if rc == hip.E_UNSUPPORTED:
            raise FusedTrainingUnsupported(hip.lib().vihds_last_error().decode())
        hip.check(rc, "vihds_ode_logp_grad")
        ctx.spec = spec
        ctx.save_for_backward(theta, cond, times, obs, dev1hot, g_unit)
        return logp

    @staticmethod
    def backward(ctx, g_logp):
        theta, cond, times, obs, dev1hot, g_unit = ctx.saved_tensors
        if g_logp.dim() == 3 and g_logp.stride(0) == 0:
            return None, g_unit * g_logp[0], None, None, None, None
        with torch.enable_grad():  # per-species weights: the general adjoint needs the trajectory after all
            th = theta.detach().requires_grad_(True)
            _traj, _xpred, logp = OdeSolveObserve.apply(ctx.spec, th, cond, times, obs, dev1hot, None)
            (g_theta,) = torch.autograd.grad(logp, th, g_logp)
        return None, g_theta, None, None, None, None


class DecoderStepFused(torch.autograd.Function):
    """The decoder side of a training step in ONE launch (vihds_theta_ode_logp_grad): theta = clip(sample(q, u)) with
    log q / log p, the device-conditioner rows, the per-species log-likelihood and the unit-weight theta adjoint.
    Inputs as ThetaSampleLogProbPacked + OdeLogLikFused; `cond_job` = None or (names-count E, first_row, w_mean,
    w_std, z or None, rng_state or None, relevance, is_default).  Returns (theta, log_q, log_p, u, logp).
    backward: one theta_bwd launch consuming g_unit scaled by the IWAE weight inside the kernel."""

    @staticmethod
    def forward(ctx, q_all, kind, p_mu, p_prec, clip_lo, clip_hi, u, n_rows, q_rows, spec, cond, times, obs, dev1hot,
                cond_job):
        _require_cuda(q_all, kind, p_mu, p_prec, clip_lo, clip_hi, q_rows, cond, times, obs)
        q_all, cond, times, obs = _c(q_all), _c(cond), _c(times), _c(obs)
        dev1hot = _c(dev1hot)
        P, B = q_all.shape[0] // 2, q_all.shape[1]
        opts = hip.ThetaOpts()
        opts.q_rows, opts.q_prec_is_log = hip.ptr(q_rows), 1
        if isinstance(u, KernelNormal):
            rng, u = u, torch.empty(u.shape, device=q_all.device, dtype=torch.float32)
            opts.rng, opts.S_total, opts.s_offset = rng.state.data_ptr(), rng.S_total, rng.s_offset
        else:
            _require_cuda(u)
            u = _c(u)
        S, T = u.shape[1], times.shape[0]
        if n_rows != spec.n_rows:
            raise RuntimeError("theta has %d rows, problem expects %d" % (n_rows, spec.n_rows))
        prob = spec.bind(B, S, T)
        dev = q_all.device
        theta = torch.empty((n_rows, B, S), device=dev, dtype=torch.float32)
        log_q = torch.empty((B, S), device=dev, dtype=torch.float32)
        log_p = torch.empty((B, S), device=dev, dtype=torch.float32)
        logp = torch.empty((4, B, S), device=dev, dtype=torch.float32)
        g_unit = torch.empty_like(theta) if spec.covers_all_rows else torch.zeros_like(theta)
        co = None
        if cond_job is not None:
            E, first_row, w_mean, w_std, z, rng_state, rel, dflt = cond_job
            co = hip.Conditioner()
            co.E, co.first_row, co.w_mean, co.w_std = E, first_row, w_mean, w_std
            co.z, co.rng, co.relevance, co.is_default = hip.ptr(z), hip.ptr(rng_state), hip.ptr(rel), hip.ptr(dflt)
        rc = _launch("decoder_step", lambda: hip.lib().vihds_theta_ode_logp_grad(
            ctypes.byref(prob), P, hip.ptr(kind), hip.ptr(q_all), hip.ptr(q_all), hip.ptr(p_mu), hip.ptr(p_prec),
            hip.ptr(clip_lo), hip.ptr(clip_hi), hip.ptr(u), ctypes.byref(opts),
            ctypes.byref(co) if co is not None else None, hip.ptr(cond), hip.ptr(dev1hot), hip.ptr(times), hip.ptr(obs),
            hip.ptr(theta), hip.ptr(log_q), hip.ptr(log_p), hip.ptr(logp), hip.ptr(g_unit), hip.current_stream()))
        if rc == hip.E_UNSUPPORTED:
            raise FusedTrainingUnsupported(hip.lib().vihds_last_error().decode())
        hip.check(rc, "vihds_theta_ode_logp_grad")
        ctx.spec = spec
        ctx.save_for_backward(q_all, kind, p_mu, p_prec, clip_lo, clip_hi, u, q_rows, g_unit, theta, cond, times, obs,
                              dev1hot)
        ctx.set_materialize_grads(False)
        ctx.mark_non_differentiable(u)
        return theta, log_q, log_p, u, logp

    @staticmethod
    def backward(ctx, g_theta, g_log_q, g_log_p, _g_u, g_logp):
        (q_all, kind, p_mu, p_prec, clip_lo, clip_hi, u, q_rows, g_unit, theta, cond, times, obs,
         dev1hot) = ctx.saved_tensors
        P, B, S = q_all.shape[0] // 2, q_all.shape[1], u.shape[1]
        opts = hip.ThetaOpts()
        opts.q_rows, opts.q_prec_is_log = hip.ptr(q_rows), 1
        keep = job = ijob = None
        if g_logp is not None and g_logp.dim() == 3 and g_logp.stride(0) == 0:
            job = _PENDING_IWAE.pop(g_logp[0].data_ptr(), None)  # a deferred IwaeLoss node upstream (fused_iwae_backward)
        if (job is not None and g_theta is None and (g_log_p is None or g_log_p.data_ptr() == job["ug"].data_ptr())
                and (g_log_q is None or (job["ugn"] is not None and g_log_q.data_ptr() == job["ugn"].data_ptr()))):
            # the loss and its importance weights are formed inside the theta-adjoint launch: no IWAE launch this step
            ijob = hip.IwaeJob()
            ijob.logp, ijob.log_p, ijob.log_q = hip.ptr(job["logp"]), hip.ptr(job["log_p"]), hip.ptr(job["log_q"])
            ijob.n_iwae_total = job["n_total"]
            ijob.log_w, ijob.lse, ijob.loss = hip.ptr(job["log_w"]), hip.ptr(job["rows"][2]), hip.ptr(job["loss"])
            ijob.ticket = hip.ptr(job["ticket"])
            opts.iwae = ctypes.pointer(ijob)
            g_th, g_log_q, g_log_p = g_unit, None, None
        elif job is not None:
            _run_iwae_job(job)  # some other combination of upstream gradients: the ordinary launch, then as usual
        if ijob is not None:
            pass
        elif g_logp is None:
            g_th = _c(g_theta)
        elif g_logp.dim() == 3 and g_logp.stride(0) == 0 and g_theta is None:
            keep = _c(g_logp[0])
            g_th, opts.g_theta_scale = g_unit, keep.data_ptr()  # the kernel applies the IWAE weight
        else:
            if g_logp.dim() == 3 and g_logp.stride(0) == 0:
                g_ode = g_unit * g_logp[0]
            else:  # per-species weights: general adjoint through the trajectory
                with torch.enable_grad():
                    th = theta.detach().requires_grad_(True)
                    _t, _x, lp = OdeSolveObserve.apply(ctx.spec, th, cond, times, obs, dev1hot, None)
                    (g_ode,) = torch.autograd.grad(lp, th, g_logp)
            g_th = g_ode if g_theta is None else g_ode + g_theta
        g_log_q, g_log_p = _c(g_log_q), _c(g_log_p)
        g_all = torch.empty_like(q_all)
        rc = hip.lib().vihds_theta_bwd(P, B, S, hip.ptr(kind), hip.ptr(q_all), hip.ptr(q_all), hip.ptr(p_mu),
                                       hip.ptr(p_prec), hip.ptr(clip_lo), hip.ptr(clip_hi), hip.ptr(u),
                                       hip.ptr(g_th), hip.ptr(g_log_q), hip.ptr(g_log_p), hip.ptr(g_all),
                                       hip.ptr(g_all), ctypes.byref(opts), hip.current_stream())
        hip.check(rc, "vihds_theta_bwd")
        return (g_all,) + (None,) * 14


class ThetaOdeFused(torch.autograd.Function):
    """The sampling stage inside the ODE FORWARD launch (vihds_theta_ode_fwd): theta = clip(sample(q, u)) with log q / log p,
    dr_blackbox's condition_theta (offset layer) and the trajectory + log-likelihood, one launch where ThetaSampleLogProbPacked
    [+ OffsetRows] + OdeSolveObserve are two or three.  Training fast path of the models WITHOUT a fused decoder step; the
    step's backward is ops.GeneralTail (which reads this node's saved tensors).  Returns (theta, log_q, log_p, u, traj, logp).
    backward (a caller that runs autograd after all): the unfused ops are replayed on the saved draws and differentiated."""

    @staticmethod
    def forward(ctx, q_all, kind, p_mu, p_prec, clip_lo, clip_hi, u, n_rows, q_rows, spec, cond, times, obs, dev1hot, weights,
                off_w, off_b, off_rows):
        _require_cuda(q_all, kind, p_mu, p_prec, clip_lo, clip_hi, q_rows, cond, times, obs)
        q_all, cond, times, obs, dev1hot = _c(q_all), _c(cond), _c(times), _c(obs), _c(dev1hot)
        P, B = q_all.shape[0] // 2, q_all.shape[1]
        opts = hip.ThetaOpts()
        opts.q_rows, opts.q_prec_is_log = hip.ptr(q_rows), 1
        rng_state = None
        if isinstance(u, KernelNormal):
            rng, u = u, torch.empty(u.shape, device=q_all.device, dtype=torch.float32)
            opts.rng, opts.S_total, opts.s_offset = rng.state.data_ptr(), rng.S_total, rng.s_offset
            rng_state = rng.state
        else:
            _require_cuda(u)
            u = _c(u)
        S, T = u.shape[1], times.shape[0]
        if n_rows != spec.n_rows:
            raise RuntimeError("theta has %d rows, problem expects %d" % (n_rows, spec.n_rows))
        prob = spec.bind(B, S, T)
        dev = q_all.device
        theta = torch.empty((n_rows, B, S), device=dev, dtype=torch.float32)
        log_q = torch.empty((B, S), device=dev, dtype=torch.float32)
        log_p = torch.empty((B, S), device=dev, dtype=torch.float32)
        traj = torch.empty((T, spec.n_states, B, S), device=dev, dtype=torch.float32)
        logp = torch.empty((4, B, S), device=dev, dtype=torch.float32)
        off = None
        if off_rows is not None:
            off = hip.OffsetLayer()
            off.src_row, off.dst_row, off.n = off_rows
            off.W, off.bias = hip.ptr(_c(off_w)), hip.ptr(_c(off_b))
        rc = _launch("ode_fwd", lambda: hip.lib().vihds_theta_ode_fwd(
            ctypes.byref(prob), P, hip.ptr(kind), hip.ptr(q_all), hip.ptr(q_all), hip.ptr(p_mu), hip.ptr(p_prec),
            hip.ptr(clip_lo), hip.ptr(clip_hi), hip.ptr(u), ctypes.byref(opts), ctypes.byref(off) if off is not None else None,
            hip.ptr(cond), hip.ptr(dev1hot), hip.ptr(times), hip.ptr(obs), hip.ptr(weights), hip.ptr(theta), hip.ptr(log_q),
            hip.ptr(log_p), hip.ptr(traj), None, hip.ptr(logp), hip.current_stream()))
        if rc == hip.E_UNSUPPORTED:
            raise FusedTrainingUnsupported(hip.lib().vihds_last_error().decode())
        hip.check(rc, "vihds_theta_ode_fwd")
        ctx.spec, ctx.prob = spec, prob
        ctx.row_offset_map = None if off_rows is None else (off_rows[0], off_rows[1], off_rows[2], "linear")
        ctx.logp_out = logp.detach()
        ctx.rng_state = rng_state  # (its step is advanced by the launch that follows: GeneralTail, or vihds_rng_advance)
        ctx.n_rows = n_rows
        ctx.save_for_backward(q_all, kind, p_mu, p_prec, clip_lo, clip_hi, u, q_rows, theta, cond, times, obs, traj, dev1hot,
                              weights, off_w, off_b)
        ctx.set_materialize_grads(False)
        ctx.mark_non_differentiable(u)
        return theta, log_q, log_p, u, traj, logp

    @staticmethod
    def backward(ctx, g_theta, g_log_q, g_log_p, _g_u, g_traj, g_logp):
        (q_all, kind, p_mu, p_prec, clip_lo, clip_hi, u, q_rows, _theta, cond, times, obs, _traj, dev1hot, weights, off_w,
         off_b) = ctx.saved_tensors
        with torch.enable_grad():
            q = q_all.detach().requires_grad_(True)
            w = weights.detach().requires_grad_(True) if weights is not None else None
            W = off_w.detach().requires_grad_(True) if off_w is not None else None
            bvec = off_b.detach().requires_grad_(True) if off_b is not None else None
            th, lq, lp, _u = ThetaSampleLogProbPacked.apply(q, kind, p_mu, p_prec, clip_lo, clip_hi, u, ctx.n_rows, q_rows)
            token, rom = None, None
            if ctx.row_offset_map is not None:
                src, dst, n, _ = ctx.row_offset_map
                token = OffsetRows.apply(W, bvec, dev1hot, th.detach(), src, dst)
                rom = ctx.row_offset_map
            tr, _xp, lg = OdeSolveObserve.apply(ctx.spec, th, cond, times, obs, dev1hot, w, token, rom, False)
            outs, gouts = [], []
            for o, g in ((th, g_theta), (lq, g_log_q), (lp, g_log_p), (tr, g_traj), (lg, g_logp)):
                if g is not None:
                    outs.append(o)
                    gouts.append(g)
            ins = [t for t in (q, w, W, bvec) if t is not None]
            grads = dict(zip([id(t) for t in ins], torch.autograd.grad(outs, ins, gouts, allow_unused=True))) if outs else {}
        g = lambda t: None if t is None else grads.get(id(t))  # noqa: E731
        return (g(q), None, None, None, None, None, None, None, None, None, None, None, None, None, g(w), g(W), g(bvec), None)


def neural_precision_weight_grads(spec, prob, aux, g_w):
    """White-box model + neural precisions: the two weight matrices of NeuralPrecisions (reference precisions.py:55-61,
    76-87; buffer order Wp [4][NIN], bp [4], Wd [4][NIN], bd [4]) from the adjoint kernel's dump [8+NIN][E][n] --
    production / degradation pre-activation adjoints (fields 0..3 / 4..7) times the layer inputs (fields 8..) -- written
    into g_w by vihds_gram_blocks; the biases were already added to g_w by the kernel."""
    NIN = spec.n_states - 4 + 1
    H = max(int(spec.proto.n_hidden_prec), 0)
    F, n = 8 + NIN + 2 * H, prob.B * prob.S
    C = aux.numel() // F
    key = "prec_rects"
    if key not in spec.cache:
        if H < 1:
            plan = [(0, 4, 8, NIN, 0, NIN), (4, 4, 8, NIN, 4 * NIN + 4, NIN)]
        else:  # hidden layer (reference precisions.py:63-74): Wh = hidden adjoints x inputs, Wp / Wd = output adjoints x hidden
            o_wp = H * NIN + H
            plan = [(8 + NIN, H, 8, NIN, 0, NIN), (0, 4, 8 + NIN + H, H, o_wp, H), (4, 4, 8 + NIN + H, H, o_wp + 4 * H + 4, H)]
        rects = (hip.GramRect * len(plan))()
        for k, (a0, na, b0, nb, d0, sa) in enumerate(plan):
            (rects[k].a0, rects[k].na, rects[k].b0, rects[k].nb, rects[k].dest0, rects[k].dest_stride_a,
             rects[k].dest_stride_b) = (a0, na, b0, nb, d0, sa, 1)
        spec.cache[key] = rects
    rects = spec.cache[key]
    n_scr = hip.lib().vihds_gram_scratch_floats(C, len(rects), rects)
    if n_scr <= 0:
        raise RuntimeError("vihds_gram_scratch_floats: %s" % hip.lib().vihds_last_error().decode())
    scratch = torch.empty(n_scr, device=aux.device, dtype=torch.float32)
    rc = hip.lib().vihds_gram_blocks(F, C, len(rects), rects, hip.ptr(aux), hip.ptr(scratch), hip.ptr(g_w),
                                     hip.current_stream())
    hip.check(rc, "vihds_gram_blocks")
    if H >= 1:  # hidden biases: row sums of the hidden pre-activation adjoints
        g_w[H * NIN: H * NIN + H] = aux.view(F, C)[8 + NIN: 8 + NIN + H].sum(1)
    return g_w


def _blackbox_grad_plan(spec, prob, device):
    """Index tables for the dr_blackbox weight gradients (built once per spec and device): which two dump rows every
    Gram-type gradient entry multiplies and where it lands in the flat weight buffer (NeuralStates.flat / the order
    DR_Blackbox.neural_weights() concatenates: Wh, bh, Wp, bp, Wd, bd of the state network, then of the precision
    network), and where the remaining (time-invariant-input and bias) entries go."""
    key = ("grad_plan", str(device))
    if key in spec.cache:
        return spec.cache[key]
    HS, HP, L = prob.n_hidden_states, prob.n_hidden_prec, prob.n_latent_states
    NX, nc = 4 + L, prob.n_const
    ZA, ZD, RHS, RGS = 0, NX, 2 * NX, 2 * NX + HS
    RY = RGS + HS
    RT = RY + NX
    ZAP, ZDP = RT + 1, RT + 5
    RHP = RT + 9
    RGP = RHP + HP
    ws, wp = NX + nc, 1 + NX + nc  # row widths of the two hidden layers
    o = [0]
    for size in (HS * ws, HS, NX * HS, NX, NX * HS, NX, HP * wp, HP, 4 * HP, 4, 4 * HP, 4):
        o.append(o[-1] + size)
    # (a0, na, b0, nb, dest0, dest stride over a, dest stride over b)
    rects = [(RGS, HS, RY, NX, o[0], ws, 1),          # Wh[:, :NX] = gs x y
             (ZA, NX, RHS, HS, o[2], HS, 1),          # Wp of the states = za x hs
             (ZD, NX, RHS, HS, o[4], HS, 1),          # Wd of the states = zd x hs
             (RGP, HP, RT, 1, o[6], wp, 1),           # Vh[:, 0] = gp x t
             (RGP, HP, RY, NX, o[6] + 1, wp, 1),      # Vh[:, 1:1+NX] = gp x y
             (ZAP, 4, RHP, HP, o[8], HP, 1),          # Wp of the precisions = zap x hp
             (ZDP, 4, RHP, HP, o[10], HP, 1)]         # Wd of the precisions = zdp x hp
    dest = []
    for (a0, na, b0, nb, d0, sa, sb) in rects:
        dest += [d0 + i * sa + j * sb for i in range(na) for j in range(nb)]
    # the rest, in the order [g_const (HS+HP rows x nc), b_hid (HS+HP), bias_sums (2NX + 8)]
    rest = []
    for h in range(HS):
        rest += [o[0] + h * ws + NX + k for k in range(nc)]
    for h in range(HP):
        rest += [o[6] + h * wp + 1 + NX + k for k in range(nc)]
    rest += list(range(o[1], o[2])) + list(range(o[7], o[8]))
    rest += list(range(o[3], o[4])) + list(range(o[5], o[6])) + list(range(o[9], o[10])) + list(range(o[11], o[12]))
    ti = lambda v, dt=torch.int32: torch.tensor(v, dtype=dt, device=device)  # noqa: E731
    rect_arr = (hip.GramRect * len(rects))()
    for k, r in enumerate(rects):
        (rect_arr[k].a0, rect_arr[k].na, rect_arr[k].b0, rect_arr[k].nb, rect_arr[k].dest0, rect_arr[k].dest_stride_a,
         rect_arr[k].dest_stride_b) = r
    # groups of rectangles that fit one vihds_gram_blocks plan each (vihds_gram.hip: GM_MAX_PROD = 16 products and 16
    # distinct 16-row blocks on the matrix cores; 128 4x4 register tiles in the LDS-tiled kernel), greedy in order
    def cost(group):
        blocks, prods, tiles = set(), 0, 0
        for (a0, na, b0, nb, _, _, _) in group:
            ta = [(a0 + 16 * i, min(16, na - 16 * i)) for i in range((na + 15) // 16)]
            tb = [(b0 + 16 * i, min(16, nb - 16 * i)) for i in range((nb + 15) // 16)]
            blocks |= set(ta) | set(tb)
            prods += len(ta) * len(tb)
            tiles += ((na + 3) // 4) * ((nb + 3) // 4)
        return prods <= 16 and len(blocks) <= 16 and tiles <= 128
    groups, cur = [], []
    for r in rects:
        if cur and not cost(cur + [r]):
            groups.append(cur)
            cur = []
        cur.append(r)
    groups.append(cur)
    group_arrs, k0 = [], 0
    for grp in groups:
        arr = (hip.GramRect * len(grp))()
        for k in range(len(grp)):
            ctypes.pointer(arr[k])[0] = rect_arr[k0 + k]
        k0 += len(grp)
        group_arrs.append(arr)
    plan = {"rects": rect_arr, "n_rects": len(rects), "groups": group_arrs, "rest": ti(rest), "total": o[12]}
    assert len(set(dest) | set(rest)) == o[12] == len(dest) + len(rest)
    spec.cache[key] = plan
    return plan


def blackbox_weight_grads(spec, prob, aux, theta, cond, dev1hot):
    """dr_blackbox weight gradients from the adjoint kernel's dump.  The contraction over (RHS evaluation x trajectory)
    -- K ~ 10^6 columns, seven dense (row block x row block) rectangles -- is ONE pass over the dump
    (vihds_gram_blocks) writing straight into the flat weight-gradient buffer; the time-invariant input columns and
    the biases come from the dump's tail (Delta = sum_evals(gs), bias sums) in a second launch
    (vihds_blackbox_tail_grads)."""
    B, S = prob.B, prob.S
    n = B * S
    if hip.lib().vihds_blackbox_gram_on_chip(ctypes.byref(prob)):
        # matrix-core adjoint: the Gram tiles were accumulated on chip; add up the wavefronts' partial sums, then the tail
        plan = _blackbox_grad_plan(spec, prob, theta.device)
        g_w = torch.empty(plan["total"], device=theta.device, dtype=torch.float32)
        rc = hip.lib().vihds_blackbox_gram_reduce(ctypes.byref(prob), hip.ptr(aux), hip.ptr(g_w), hip.current_stream())
        hip.check(rc, "vihds_blackbox_gram_reduce")
        off = int(hip.lib().vihds_blackbox_tail_offset_floats(ctypes.byref(prob)))
        rc = hip.lib().vihds_blackbox_tail_grads(ctypes.byref(prob), hip.ptr(theta), hip.ptr(cond), hip.ptr(dev1hot),
                                                 aux.data_ptr() + 4 * off, hip.ptr(plan["rest"]), hip.ptr(g_w),
                                                 hip.current_stream())
        hip.check(rc, "vihds_blackbox_tail_grads")
        return g_w
    F = hip.lib().vihds_problem_dump_fields(ctypes.byref(prob))
    HS, HP, L = prob.n_hidden_states, prob.n_hidden_prec, prob.n_latent_states
    NX = 4 + L
    NP = HS + HP
    n_tail = NP + 2 * NX + 8
    E = (aux.numel() - n_tail * n) // (F * n)
    plan = _blackbox_grad_plan(spec, prob, theta.device)
    g_w = torch.empty(plan["total"], device=theta.device, dtype=torch.float32)
    # one pass over the dump when the seven rectangles fit the contraction kernels' plan (<= 16 products of 16-row
    # blocks, <= 16 distinct blocks: the ICML sizes), otherwise one pass per group of rectangles that does (a wider
    # network, e.g. the default n_hidden_decoder = 50: two groups)
    for group in plan["groups"]:
        n_scr = hip.lib().vihds_gram_scratch_floats(E * n, len(group), group)
        rc = hip.E_UNSUPPORTED
        if n_scr > 0:
            scratch = torch.empty(n_scr, device=theta.device, dtype=torch.float32)
            rc = hip.lib().vihds_gram_blocks(F, E * n, len(group), group, hip.ptr(aux), hip.ptr(scratch), hip.ptr(g_w),
                                             hip.current_stream())
        if rc == hip.E_UNSUPPORTED:
            # outside the kernels' regime (a single rectangle past the plan limits, or a column count that is not a
            # multiple of 64 with more than 126 dump fields): the rectangles as library GEMMs over the dump, on the device
            X = aux[: F * E * n].view(F, E * n)
            for r in group:
                blk = X[r.a0: r.a0 + r.na] @ X[r.b0: r.b0 + r.nb].t()
                idx = (r.dest0 + r.dest_stride_a * torch.arange(r.na, device=X.device)[:, None]
                       + r.dest_stride_b * torch.arange(r.nb, device=X.device)[None, :])
                g_w[idx.reshape(-1)] = blk.reshape(-1)
        else:
            hip.check(rc, "vihds_gram_blocks")
    # the time-invariant input columns and the biases: one launch over the dump's tail
    rc = hip.lib().vihds_blackbox_tail_grads(ctypes.byref(prob), hip.ptr(theta), hip.ptr(cond), hip.ptr(dev1hot),
                                             aux.data_ptr() + 4 * F * E * n, hip.ptr(plan["rest"]), hip.ptr(g_w),
                                             hip.current_stream())
    hip.check(rc, "vihds_blackbox_tail_grads")
    return g_w


class ThetaSampleLogProb(torch.autograd.Function):
    """q.sample(u) -> p.clip -> (theta, log q(theta), log p(theta)) for all P parameters in one kernel.

    q_mu, q_prec: [P,B];  p_mu, p_prec, clip_lo, clip_hi: [P];  kind: int32 [P];  u: [B,S,P]
    returns theta [n_rows,B,S] (rows P.. are zero, for the caller to fill), log_q [B,S], log_p [B,S]
    """

    @staticmethod
    def forward(ctx, q_mu, q_prec, kind, p_mu, p_prec, clip_lo, clip_hi, u, n_rows):
        _require_cuda(q_mu, q_prec, kind, p_mu, p_prec, clip_lo, clip_hi, u)
        q_mu, q_prec, u = _c(q_mu), _c(q_prec), _c(u)
        P, B = q_mu.shape
        S = u.shape[1]
        if u.shape[0] != B or u.shape[2] != P:
            raise RuntimeError("u must be [B,S,P]")
        # rows P.. are reserved for the caller (filled in place, e.g. by the device-conditioner kernel)
        theta = torch.empty((max(n_rows, P), B, S), device=u.device, dtype=torch.float32)
        log_q = torch.empty((B, S), device=u.device, dtype=torch.float32)
        log_p = torch.empty((B, S), device=u.device, dtype=torch.float32)
        rc = hip.lib().vihds_theta_fwd(P, B, S, hip.ptr(kind), hip.ptr(q_mu), hip.ptr(q_prec), hip.ptr(p_mu),
                                       hip.ptr(p_prec), hip.ptr(clip_lo), hip.ptr(clip_hi), hip.ptr(u),
                                       hip.ptr(theta), hip.ptr(log_q), hip.ptr(log_p), None, hip.current_stream())
        hip.check(rc, "vihds_theta_fwd")
        ctx.save_for_backward(q_mu, q_prec, kind, p_mu, p_prec, clip_lo, clip_hi, u)
        ctx.set_materialize_grads(False)
        return theta, log_q, log_p

    @staticmethod
    def backward(ctx, g_theta, g_log_q, g_log_p):
        q_mu, q_prec, kind, p_mu, p_prec, clip_lo, clip_hi, u = ctx.saved_tensors
        P, B = q_mu.shape
        S = u.shape[1]
        g_theta, g_log_q, g_log_p = _c(g_theta), _c(g_log_q), _c(g_log_p)
        g_mu = torch.empty_like(q_mu)
        g_prec = torch.empty_like(q_prec)
        rc = hip.lib().vihds_theta_bwd(P, B, S, hip.ptr(kind), hip.ptr(q_mu), hip.ptr(q_prec), hip.ptr(p_mu),
                                       hip.ptr(p_prec), hip.ptr(clip_lo), hip.ptr(clip_hi), hip.ptr(u),
                                       hip.ptr(g_theta), hip.ptr(g_log_q), hip.ptr(g_log_p), hip.ptr(g_mu),
                                       hip.ptr(g_prec), None, hip.current_stream())
        hip.check(rc, "vihds_theta_bwd")
        return g_mu, g_prec, None, None, None, None, None, None, None


class EncoderQTables(torch.autograd.Function):
    """Encoder.evaluate_q (reference encoders.py:383-404) in one forward and two backward launches
    (csrc/vihds_encoder.hip): returns the level-blocked [2P,B] table of means and log-precisions."""

    @staticmethod
    def forward(ctx, shape, delta_obs, inputs, dev_1hot, conv_w, conv_b, lin_w, lin_b, local_w, local_b, gcond_w,
                global_free, const_values):
        _require_cuda(delta_obs, conv_w, conv_b, lin_w, lin_b)
        delta_obs, inputs, dev_1hot = _c(delta_obs), _c(inputs), _c(dev_1hot)
        s = shape
        B, dev = s.B, delta_obs.device
        n_pool = s.F * (s.L - s.K + 1 - s.pool + 1)
        q_all = torch.empty((2 * (s.nl + s.ng + s.ngl + s.nc), B), device=dev, dtype=torch.float32)
        pooled = torch.empty((B, n_pool), device=dev, dtype=torch.float32)
        hidden = torch.empty((B, s.H), device=dev, dtype=torch.float32)
        rc = hip.lib().vihds_encoder_fwd(ctypes.byref(s), hip.ptr(delta_obs), hip.ptr(inputs), hip.ptr(dev_1hot),
                                         hip.ptr(conv_w), hip.ptr(conv_b), hip.ptr(lin_w), hip.ptr(lin_b),
                                         hip.ptr(local_w), hip.ptr(local_b), hip.ptr(gcond_w), hip.ptr(global_free),
                                         hip.ptr(const_values), hip.ptr(q_all), hip.ptr(pooled), hip.ptr(hidden),
                                         hip.current_stream())
        hip.check(rc, "vihds_encoder_fwd")
        ctx.shape = s
        ctx.save_for_backward(delta_obs, inputs, dev_1hot, conv_w, lin_w, local_w, local_b, gcond_w, global_free,
                              pooled, hidden)
        ctx.hidden = hidden
        return q_all

    @staticmethod
    def backward(ctx, g_all):
        s = ctx.shape
        delta_obs, inputs, dev_1hot, conv_w, lin_w, local_w, local_b, gcond_w, global_free, pooled, hidden = \
            ctx.saved_tensors
        g_all = _c(g_all)
        dev = g_all.device
        # all parameter gradients are carved out of ONE buffer, in the order Encoder.parameters() lists them (the
        # module's own global_free first, then conv.weight, conv.bias, lin.weight, lin.bias, local heads, gcond heads):
        # a sharded step can then all-reduce the buffer in place instead of flattening and scattering
        # (vihds/parallel.py)
        shapes = [None if global_free is None else global_free.shape, conv_w.shape, (s.F,), lin_w.shape, (s.H,)] + \
                 [None if t is None else t.shape for t in (local_w, local_b, gcond_w)]
        sizes = [0 if sh is None else int(torch.Size(sh).numel()) for sh in shapes]
        arena = torch.empty(sum(sizes), device=dev, dtype=torch.float32)
        views, off = [], 0
        for sh, k in zip(shapes, sizes):
            views.append(None if sh is None else arena[off: off + k].view(sh))
            off += k
        g_glob, g_conv_w, g_conv_b, g_lin_w, g_lin_b, g_local_w, g_local_b, g_gcond_w = views
        g_pre = torch.empty((s.B, s.H), device=dev)
        g_conv = torch.empty((s.B, s.F, s.L - s.K + 1), device=dev)
        rc = hip.lib().vihds_encoder_bwd(ctypes.byref(s), hip.ptr(g_all), hip.ptr(delta_obs), hip.ptr(inputs),
                                         hip.ptr(dev_1hot), hip.ptr(lin_w), hip.ptr(local_w), hip.ptr(pooled),
                                         hip.ptr(hidden), hip.ptr(g_pre), hip.ptr(g_conv), hip.ptr(g_conv_w),
                                         hip.ptr(g_conv_b), hip.ptr(g_lin_w), hip.ptr(g_lin_b), hip.ptr(g_local_w),
                                         hip.ptr(g_local_b), hip.ptr(g_gcond_w), hip.ptr(g_glob),
                                         hip.current_stream())
        hip.check(rc, "vihds_encoder_bwd")
        return (None, None, None, None, g_conv_w, g_conv_b, g_lin_w, g_lin_b, g_local_w, g_local_b, g_gcond_w, g_glob,
                None)


class KernelNormal(object):
    """Stand-in for the u [B,S,P] tensor when the theta kernel draws the standard normals itself
    (u_rng: kernel).  `state` is the 4-word device RNG state {seed lo, seed hi, step, ticket} of
    vihds_theta_opts.rng; (S_total, s_offset) select this rank's slice of the global draw."""

    def __init__(self, shape, state, S_total=None, s_offset=0):
        self.shape = tuple(shape)
        self.state = state
        self.S_total = shape[1] if S_total is None else S_total
        self.s_offset = s_offset
        self.device = state.device

    @staticmethod
    def new_state(seed, device):
        seed = int(seed) & 0xFFFFFFFFFFFFFFFF
        lo, hi = seed & 0xFFFFFFFF, seed >> 32
        as_i32 = lambda v: v - (1 << 32) if v >= (1 << 31) else v  # noqa: E731
        return torch.tensor([as_i32(lo), as_i32(hi), 0, 0], dtype=torch.int32, device=device)

    def take(self, lo, hi):
        B, S, P = self.shape
        return KernelNormal((B, hi - lo, P), self.state, S_total=S, s_offset=lo)


class ThetaSampleLogProbPacked(torch.autograd.Function):
    """Same kernel as ThetaSampleLogProb, fed by the encoder's single [2P,B] table of means and LOG-precisions in
    the encoder's own row order (`q_rows` [2P]: the row of mu_p, then the row of log_prec_p): the kernel
    exponentiates, and the backward returns ONE [2P,B] gradient (d/d mu, d/d log_prec) in the same row order, so
    autograd needs no slice / exp / mul nodes between the encoder heads and the kernel.  `u` is either the [B,S,P]
    tensor of standard normals or a KernelNormal (the kernel then draws them and the tensor is an output)."""

    @staticmethod
    def forward(ctx, q_all, kind, p_mu, p_prec, clip_lo, clip_hi, u, n_rows, q_rows):
        _require_cuda(q_all, kind, p_mu, p_prec, clip_lo, clip_hi, q_rows)
        q_all = _c(q_all)
        P = q_all.shape[0] // 2
        B = q_all.shape[1]
        opts = hip.ThetaOpts()
        opts.q_rows = hip.ptr(q_rows)
        opts.q_prec_is_log = 1
        if isinstance(u, KernelNormal):
            rng, u = u, torch.empty(u.shape, device=q_all.device, dtype=torch.float32)
            opts.rng, opts.S_total, opts.s_offset = rng.state.data_ptr(), rng.S_total, rng.s_offset
        else:
            _require_cuda(u)
            u = _c(u)
        S = u.shape[1]
        if u.shape[0] != B or u.shape[2] != P:
            raise RuntimeError("u must be [B,S,P]")
        theta = torch.empty((max(n_rows, P), B, S), device=u.device, dtype=torch.float32)
        log_q = torch.empty((B, S), device=u.device, dtype=torch.float32)
        log_p = torch.empty((B, S), device=u.device, dtype=torch.float32)
        rc = hip.lib().vihds_theta_fwd(P, B, S, hip.ptr(kind), hip.ptr(q_all), hip.ptr(q_all), hip.ptr(p_mu),
                                       hip.ptr(p_prec), hip.ptr(clip_lo), hip.ptr(clip_hi), hip.ptr(u),
                                       hip.ptr(theta), hip.ptr(log_q), hip.ptr(log_p), ctypes.byref(opts),
                                       hip.current_stream())
        hip.check(rc, "vihds_theta_fwd")
        ctx.save_for_backward(q_all, kind, p_mu, p_prec, clip_lo, clip_hi, u, q_rows)
        ctx.set_materialize_grads(False)
        ctx.mark_non_differentiable(u)
        return theta, log_q, log_p, u

    @staticmethod
    def backward(ctx, g_theta, g_log_q, g_log_p, _g_u):
        q_all, kind, p_mu, p_prec, clip_lo, clip_hi, u, q_rows = ctx.saved_tensors
        P, B, S = q_all.shape[0] // 2, q_all.shape[1], u.shape[1]
        g_theta, g_log_q, g_log_p = _c(g_theta), _c(g_log_q), _c(g_log_p)
        g_all = torch.empty_like(q_all)
        opts = hip.ThetaOpts()
        opts.q_rows = hip.ptr(q_rows)
        opts.q_prec_is_log = 1
        rc = hip.lib().vihds_theta_bwd(P, B, S, hip.ptr(kind), hip.ptr(q_all), hip.ptr(q_all), hip.ptr(p_mu),
                                       hip.ptr(p_prec), hip.ptr(clip_lo), hip.ptr(clip_hi), hip.ptr(u),
                                       hip.ptr(g_theta), hip.ptr(g_log_q), hip.ptr(g_log_p), hip.ptr(g_all),
                                       hip.ptr(g_all), ctypes.byref(opts), hip.current_stream())
        hip.check(rc, "vihds_theta_bwd")
        return g_all, None, None, None, None, None, None, None, None


class IwaeRows(torch.autograd.Function):
    """log_w = sum_j logp[j] + log_p - log_q and its per-row (max, sum-exp).  Returns (log_w, row_max,
    row_sumexp); gradients flow back through `lse` via :func:`iwae_lse`."""

    @staticmethod
    def forward(ctx, logp, log_p, log_q):
        _require_cuda(logp, log_p, log_q)
        logp, log_p, log_q = _c(logp), _c(log_p), _c(log_q)
        _, B, S = logp.shape
        log_w = torch.empty((B, S), device=logp.device, dtype=torch.float32)
        row_max = torch.empty((B,), device=logp.device, dtype=torch.float32)
        row_se = torch.empty((B,), device=logp.device, dtype=torch.float32)
        rc = hip.lib().vihds_iwae_fwd(B, S, hip.ptr(logp), hip.ptr(log_p), hip.ptr(log_q), hip.ptr(log_w),
                                      hip.ptr(row_max), hip.ptr(row_se), hip.current_stream())
        hip.check(rc, "vihds_iwae_fwd")
        ctx.has = (log_p is not None, log_q is not None)
        ctx.mark_non_differentiable(row_max, row_se)
        return log_w, row_max, row_se

    @staticmethod
    def backward(ctx, g_logw, _gm, _gs):
        if g_logw is None:
            return None, None, None
        g4 = g_logw.unsqueeze(0).expand(4, -1, -1)
        return g4, (g_logw if ctx.has[0] else None), (-g_logw if ctx.has[1] else None)


class _LseFromLogw(torch.autograd.Function):
    """lse[b] given log_w and the (possibly cross-rank combined) lse; backward = softmax weights kernel."""

    @staticmethod
    def forward(ctx, log_w, lse):
        ctx.save_for_backward(log_w, lse)
        return lse.clone()

    @staticmethod
    def backward(ctx, g_lse):
        log_w, lse = ctx.saved_tensors
        B, S = log_w.shape
        g_lse = _c(g_lse)
        g_logw = torch.empty_like(log_w)
        rc = hip.lib().vihds_iwae_bwd(B, S, hip.ptr(log_w), hip.ptr(lse), hip.ptr(g_lse), hip.ptr(g_logw),
                                      hip.current_stream())
        hip.check(rc, "vihds_iwae_bwd")
        return g_logw, None


def iwae_lse(logp, log_p, log_q, group=None):
    """Row-wise logsumexp of the importance weights (vihds/training.py:141-144).

    With ``group`` (a torch.distributed process group over which the S axis is sharded) the per-row
    (max, sum-exp) pairs are combined with two tiny all-reduces; the backward needs no communication because
    the local softmax weights only need the global lse."""
    log_w, row_max, row_se = IwaeRows.apply(logp, log_p, log_q)
    if group is not None:
        from vihds.parallel import combine_row_lse

        lse = combine_row_lse(row_max, row_se, group)
    else:
        lse = row_max + torch.log(row_se)
    return _LseFromLogw.apply(log_w, lse), log_w


_UNIT = {}


def unit_gradient(device):
    """The scalar 1.0 to seed `loss.backward(unit_gradient(dev))` with: no ones_like fill launch per step, and the
    IWAE loss recognises it (by address) and hands out the gradient its forward kernel already wrote."""
    key = str(device)
    if key not in _UNIT:
        _UNIT[key] = torch.ones((), device=device, dtype=torch.float32)
    return _UNIT[key]


_TICKETS = {}


def _iwae_ticket(device):
    """The zero-initialised block counter vihds_iwae_loss_fwd finishes with (the kernel leaves it at zero again).  One
    per (device, stream), because two launches that may overlap must not share it; launches being captured into a
    hipGraph use one counter per device that was allocated beforehand (an allocation inside the capture would put a
    memset node into every replay)."""
    capturing = torch.cuda.is_current_stream_capturing()
    key = (str(device), "capture" if capturing else torch.cuda.current_stream(device).cuda_stream)
    if key not in _TICKETS:
        _TICKETS[key] = torch.zeros(1, device=device, dtype=torch.int32)
        if not capturing:
            _TICKETS.setdefault((str(device), "capture"), torch.zeros(1, device=device, dtype=torch.int32))
    return _TICKETS[key]


class IwaeLoss(torch.autograd.Function):
    """Single-process -ELBO: one launch for small batches (rows kernel + finish otherwise); the gradient w.r.t.
    logp is returned as a stride-0 view over the four species (consumed without a copy by the ODE adjoint).  For
    small batches the forward kernel also writes the gradient for a unit upstream gradient, so a backward seeded with
    unit_gradient() launches nothing here."""

    @staticmethod
    def forward(ctx, logp, log_p, log_q, n_total, defer=False):
        _require_cuda(logp, log_p, log_q)
        logp, log_p, log_q = _c(logp), _c(log_p), _c(log_q)
        _, B, S = logp.shape
        dev = logp.device
        log_w = torch.empty((B, S), device=dev, dtype=torch.float32)
        rows = torch.empty((3, B), device=dev, dtype=torch.float32)  # row_max, row_sumexp, lse
        loss = torch.empty((), device=dev, dtype=torch.float32)
        ug = ugn = None
        ticket = _iwae_ticket(dev)
        if any(ctx.needs_input_grad) and hip.lib().vihds_iwae_loss_unit_grad(B, S, 1):
            ug = torch.empty((B, S), device=dev, dtype=torch.float32)
            ugn = torch.empty((B, S), device=dev, dtype=torch.float32) if log_q is not None else None
        ctx.deferred = None
        if defer and ug is not None and S * 4 <= 60 * 1024:
            # params.fused_iwae_backward: nothing is launched here.  The decoder step's backward (DecoderStepFused), which
            # is the only consumer of this node's gradients, evaluates the loss inside its theta-adjoint launch
            # (vihds_iwae_job) and fills `loss`, `log_w`, `lse`; the job waits in _PENDING_IWAE under the address of the
            # (still unwritten) unit-gradient buffer this node's backward hands down.  A backward that is not seeded with
            # the unit gradient, or a consumer that cannot take the job, runs the ordinary kernel instead (_run_iwae_job).
            job = {"logp": logp, "log_p": log_p, "log_q": log_q, "n_total": int(n_total), "log_w": log_w, "rows": rows,
                   "loss": loss, "ug": ug, "ugn": ugn, "ticket": ticket}
            ctx.deferred = job
            _PENDING_IWAE[ug.data_ptr()] = job
            lse = rows[2]
            ctx.save_for_backward(log_w, lse, ug, ugn)
            ctx.has = (log_p is not None, log_q is not None)
            ctx.mark_non_differentiable(log_w, lse)
            ctx.set_materialize_grads(False)
            return loss, log_w, lse
        if ug is not None:
            _PENDING_IWAE.pop(ug.data_ptr(), None)  # (a deferred job abandoned without a backward may have owned this address)
        rc = hip.lib().vihds_iwae_loss_fwd(B, S, int(n_total), hip.ptr(logp), hip.ptr(log_p), hip.ptr(log_q),
                                           hip.ptr(log_w), hip.ptr(rows[0]), hip.ptr(rows[1]), hip.ptr(rows[2]),
                                           hip.ptr(loss), hip.ptr(ug), hip.ptr(ugn), hip.ptr(ticket),
                                           hip.current_stream())
        hip.check(rc, "vihds_iwae_loss_fwd")
        lse = rows[2]
        ctx.save_for_backward(log_w, lse, ug, ugn)
        ctx.has = (log_p is not None, log_q is not None)
        ctx.mark_non_differentiable(log_w, lse)
        ctx.set_materialize_grads(False)
        return loss, log_w, lse

    @staticmethod
    def backward(ctx, g_loss, _g1, _g2):
        if g_loss is None:
            return None, None, None, None, None
        log_w, lse, ug, ugn = ctx.saved_tensors
        B, S = log_w.shape
        unit = _UNIT.get(str(log_w.device))
        if ug is not None and unit is not None and g_loss.data_ptr() == unit.data_ptr():
            return ug.unsqueeze(0).expand(4, -1, -1), ug if ctx.has[0] else None, ugn, None, None
        if getattr(ctx, "deferred", None) is not None and _PENDING_IWAE.pop(ug.data_ptr(), None) is not None:
            _run_iwae_job(ctx.deferred)  # general upstream gradient: the forward's kernel after all
        g_logw = torch.empty_like(log_w)
        g_neg = torch.empty_like(log_w) if ctx.has[1] else None
        rc = hip.lib().vihds_iwae_loss_bwd(B, S, hip.ptr(log_w), hip.ptr(lse), hip.ptr(_c(g_loss)), hip.ptr(g_logw),
                                           hip.ptr(g_neg), hip.current_stream())
        hip.check(rc, "vihds_iwae_loss_bwd")
        return g_logw.unsqueeze(0).expand(4, -1, -1), g_logw if ctx.has[0] else None, g_neg, None, None


_PENDING_IWAE = {}  # address of a deferred IwaeLoss node's unit-gradient buffer -> its job (see IwaeLoss.forward)


def _run_iwae_job(job):
    """The ordinary IWAE launch for a deferred job (fills loss, log_w, lse and the unit-gradient buffers)."""
    logp, rows = job["logp"], job["rows"]
    _, B, S = logp.shape
    rc = hip.lib().vihds_iwae_loss_fwd(B, S, job["n_total"], hip.ptr(logp), hip.ptr(job["log_p"]), hip.ptr(job["log_q"]),
                                       hip.ptr(job["log_w"]), hip.ptr(rows[0]), hip.ptr(rows[1]), hip.ptr(rows[2]),
                                       hip.ptr(job["loss"]), hip.ptr(job["ug"]), hip.ptr(job["ugn"]),
                                       hip.ptr(job["ticket"]), hip.current_stream())
    hip.check(rc, "vihds_iwae_loss_fwd")


class StepTail(object):
    """Host side of vihds_step_tail (params.fused_step_tail): everything of a training step behind the decoder launch --
    IWAE loss, the backward through theta / log q / log p to q's tables and through the encoder, Adam -- in two launches
    instead of five, for a single process whose trainable parameters are exactly the encoder's.  It takes what the
    step's forward left in its autograd nodes (DecoderStepFused: draws, unit-weight theta gradient, q tables;
    EncoderQTables: pooled / hidden activations) and the deferred IWAE job, launches, and hands the gradients to the
    parameters' .grad as views of one arena; autograd's backward and optimizer.step() are not run for that step."""

    def __init__(self, encoder, optimizer):
        lh, gh = encoder.local_heads, encoder.gcond_heads
        c = encoder.conditional
        self.tensors = [encoder.global_free, c.conv.weight, c.conv.bias, c.lin.weight, c.lin.bias,
                        None if lh is None else lh.weight, None if lh is None else lh.bias,
                        None if gh is None else gh.weight]
        self.optimizer = optimizer
        self._arena = {}

    def applicable(self):
        """Single parameter group holding exactly the encoder's tensors, HipAdam on the GPU."""
        from vihds.optim import HipAdam

        opt = self.optimizer
        if not isinstance(opt, HipAdam) or len(opt.param_groups) != 1:
            return False
        group = [p for p in opt.param_groups[0]["params"] if p.requires_grad]
        mine = [t for t in self.tensors if t is not None]
        return len(group) == len(mine) and {id(p) for p in group} == {id(t) for t in mine} and all(t.is_cuda for t in mine)

    def launch(self, dec_node, enc_node, job, apply_adam=True):
        """apply_adam False (row replicas): the second launch only forms the gradient sums into the arena -- the caller
        all-reduces them and runs the Adam launch on the result."""
        (q_all, kind, p_mu, p_prec, clip_lo, clip_hi, u, q_rows, g_unit, _theta, _cond, _times, _obs,
         _dev1hot) = dec_node.saved_tensors
        delta_obs, inputs, dev_1hot, _cw, lin_w, local_w, _lb, _gw, _gf, pooled, hidden = enc_node.saved_tensors
        s = enc_node.shape
        P, B, S = q_all.shape[0] // 2, q_all.shape[1], u.shape[1]
        dev = q_all.device
        opt = self.optimizer
        group = opt.param_groups[0]
        st = opt._group_state(0, group)
        offsets, off = {}, 0
        for prm in st["params"]:
            offsets[id(prm)] = off
            off += prm.numel()
        key = (B, str(dev))
        if key not in self._arena or torch.cuda.is_current_stream_capturing():
            # (a capture gets buffers of its own from the graph's pool: they must stay put for every replay)
            sizes = [0 if t is None else t.numel() for t in self.tensors]
            arena = torch.empty(sum(sizes), device=dev, dtype=torch.float32)
            Lc = s.L - s.K + 1
            bufs = (arena, sizes, torch.empty((2 * P, B), device=dev), torch.empty((B, s.H), device=dev),
                    torch.empty((B, s.F, Lc), device=dev))
            if torch.cuda.is_current_stream_capturing():
                self._arena[key + ("capture", len(self._arena))] = bufs
            else:
                self._arena[key] = bufs
        else:
            bufs = self._arena[key]
        arena, sizes, g_all, g_pre, g_conv = bufs
        a = hip.StepTailArgs()
        a.P, a.S = P, S
        a.kind, a.q_all, a.q_rows = hip.ptr(kind), hip.ptr(q_all), hip.ptr(q_rows)
        a.p_mu, a.p_prec, a.clip_lo, a.clip_hi = hip.ptr(p_mu), hip.ptr(p_prec), hip.ptr(clip_lo), hip.ptr(clip_hi)
        a.u, a.g_theta_unit = hip.ptr(u), hip.ptr(g_unit)
        a.iwae.logp, a.iwae.log_p, a.iwae.log_q = hip.ptr(job["logp"]), hip.ptr(job["log_p"]), hip.ptr(job["log_q"])
        a.iwae.n_iwae_total = job["n_total"]
        a.iwae.log_w, a.iwae.lse, a.iwae.loss = hip.ptr(job["log_w"]), hip.ptr(job["rows"][2]), hip.ptr(job["loss"])
        a.g_all = hip.ptr(g_all)
        a.delta_obs, a.inputs, a.dev1hot = hip.ptr(delta_obs), hip.ptr(inputs), hip.ptr(dev_1hot)
        a.lin_w, a.local_w, a.pooled, a.hidden = hip.ptr(lin_w), hip.ptr(local_w), hip.ptr(pooled), hip.ptr(hidden)
        a.g_pre, a.g_conv = hip.ptr(g_pre), hip.ptr(g_conv)
        views, o = [], 0
        for k, t in enumerate(self.tensors):
            if t is None:
                views.append(None)
                continue
            if not t.is_contiguous():
                raise RuntimeError("vihds_step_tail needs contiguous parameters")
            views.append(arena[o:o + sizes[k]].view(t.shape))
            a.param[k], a.grad[k], a.mv_offset[k] = t.data_ptr(), views[-1].data_ptr(), offsets[id(t)]
            o += sizes[k]
        lr = group["lr"]
        a.m, a.v, a.state = st["m"].data_ptr(), st["v"].data_ptr(), (st["state"].data_ptr() if apply_adam else None)
        a.lr_dev = hip.ptr(lr) if isinstance(lr, torch.Tensor) else None
        a.lr = 0.0 if isinstance(lr, torch.Tensor) else float(lr)
        a.beta1, a.beta2 = group["betas"]
        a.eps = group["eps"]
        rc = _launch("step_tail", lambda: hip.lib().vihds_step_tail(ctypes.byref(s), ctypes.byref(a), hip.current_stream()))
        hip.check(rc, "vihds_step_tail")
        for t, v in zip(self.tensors, views):
            if t is not None:
                t.grad = v
        return job["loss"]


class GeneralTail(StepTail):
    """vihds_step_tail for ANY model (ABI 13; reference training.py:324-340 is model-agnostic): the step's forward ran as
    the ordinary autograd-tracked launches (encoder -> theta kernel -> [conditioning] -> vihds_ode_fwd); this object runs
    everything behind them WITHOUT autograd: the IWAE launch (loss, importance weights), vihds_ode_bwd fed those weights
    as its log-likelihood gradient, the decoder networks' weight-gradient contraction where the model has one, and the two
    tail launches -- whose second now also applies Adam to the decoder-side tensors (NeuralPrecisions / NeuralStates
    weights, dr_blackbox's offset layer) and whose first forms the offset layer's row sums.  What used to be ten to thirteen
    launches (theta adjoint, offset adjoint, weight reduce, encoder adjoint x 2, Adam, several aten fills / adds) is two."""

    def __init__(self, encoder, optimizer, ode_model):
        super(GeneralTail, self).__init__(encoder, optimizer)
        self.ode = ode_model
        self.flat_tensors = ode_model.flat_weight_tensors()  # [] for white-box models; the kernels' weight buffer order
        off = getattr(ode_model, "offset_layer", None)
        self.offset = off if (off is not None and getattr(ode_model, "n_y", 0) > 0) else None
        self._bufs = {}
        self._declined = {}  # shape key -> why launch() declines it (looked at before any kernel is queued)
        self._maps = {}
        self._sides = {}
        self.inkernel_iwae = True  # (False: vihds_iwae_loss_fwd + vihds_ode_bwd -- the same numbers, one launch more)

    def applicable(self):
        """HipAdam, one parameter group = the encoder's tensors + every decoder-side parameter, each of which this path
        updates (the flat weight buffer's tensors, the offset layer)."""
        from vihds.optim import HipAdam

        opt = self.optimizer
        if not isinstance(opt, HipAdam) or len(opt.param_groups) != 1:
            return False
        group = [p for p in opt.param_groups[0]["params"] if p.requires_grad]
        mine = [t for t in self.tensors if t is not None] + list(self.flat_tensors)
        if self.offset is not None:
            mine += [self.offset.weight, self.offset.bias]
        decoder = [p for p in self.ode.parameters() if p.requires_grad]
        covered = {id(t) for t in mine}
        if any(id(p) not in covered for p in decoder):
            return False
        return (len(group) == len(mine) and {id(p) for p in group} == covered and all(t.is_cuda for t in mine)
                and len(self._chunks_static()) <= hip.TAIL_MAX_EXTRA)

    def _chunks_static(self):
        """Runs of the flat weight buffer's tensors that are also back to back in Adam's flat m / v (= consecutive in
        model.parameters() order): [(first tensor index, n tensors, flat offset, size)]."""
        opt = self.optimizer
        order = {id(p): k for k, p in enumerate(p for p in opt.param_groups[0]["params"] if p.requires_grad)}
        chunks, o = [], 0
        for k, t in enumerate(self.flat_tensors):
            if chunks and order.get(id(t), -9) == order.get(id(self.flat_tensors[k - 1]), -9) + 1:
                first, cnt, fo, size = chunks[-1]
                chunks[-1] = (first, cnt + 1, fo, size + t.numel())
            else:
                chunks.append((k, 1, o, t.numel()))
            o += t.numel()
        return chunks

    def _lane_map(self, NIN, dev):
        """flat element (Wp [4][NIN], bp [4], Wd [4][NIN], bd [4]) -> its place in a lane-split adjoint's partial row
        (vihds_relay_lanes.hpp: [4][2 NIN + 2] = rows (Wp row, Wd row, bp, bd) of output o)."""
        key = (NIN, str(dev))
        if key not in self._maps:
            nwrow = 2 * NIN + 2
            m = []
            for o in range(4):
                m += [o * nwrow + j for j in range(NIN)]
            m += [o * nwrow + 2 * NIN for o in range(4)]
            for o in range(4):
                m += [o * nwrow + NIN + j for j in range(NIN)]
            m += [o * nwrow + 2 * NIN + 1 for o in range(4)]
            self._maps[key] = torch.tensor(m, dtype=torch.int32, device=dev)
        return self._maps[key]

    @staticmethod
    def forward_state(theta_node, ode_node):
        """The step's forward state as GeneralTail.launch takes it, from ThetaSampleLogProbPacked's and OdeSolveObserve's
        backward nodes -- or from the one node of ThetaOdeFused (pass it twice)."""
        if type(ode_node).__name__ == "ThetaOdeFusedBackward":
            (q_all, kind, p_mu, p_prec, clip_lo, clip_hi, u, q_rows, theta, cond, times, obs, traj, dev1hot, weights, _ow,
             _ob) = ode_node.saved_tensors
            rng = ode_node.rng_state
        else:
            q_all, kind, p_mu, p_prec, clip_lo, clip_hi, u, q_rows = theta_node.saved_tensors
            theta, cond, times, obs, traj, dev1hot, weights = ode_node.saved_tensors
            rng = None
        return {"q_all": q_all, "kind": kind, "p_mu": p_mu, "p_prec": p_prec, "clip_lo": clip_lo, "clip_hi": clip_hi, "u": u,
                "q_rows": q_rows, "theta": theta, "cond": cond, "times": times, "obs": obs, "traj": traj, "dev1hot": dev1hot,
                "weights": weights, "spec": ode_node.spec, "prob": ode_node.prob, "rom": ode_node.row_offset_map,
                "logp": ode_node.logp_out, "rng_advance": rng}

    def launch(self, fwd, enc_node, log_q, log_p, n_total, apply_adam=True):
        """fwd: forward_state(...) of the step; enc_node: EncoderQTables' backward node.  Returns the loss tensor (-ELBO), or
        None when the step is outside this path's regime (time-fastest trajectory layout, an offset that is not the
        one-launch linear form)."""
        q_all, kind, p_mu, p_prec, clip_lo, clip_hi, u, q_rows = (fwd[k] for k in ("q_all", "kind", "p_mu", "p_prec", "clip_lo",
                                                                                   "clip_hi", "u", "q_rows"))
        theta, cond, times, obs, traj, dev1hot, weights = (fwd[k] for k in ("theta", "cond", "times", "obs", "traj", "dev1hot",
                                                                            "weights"))
        delta_obs, inputs, dev_1hot_e, _cw, lin_w, local_w, _lb, _gw, _gf, pooled, hidden = enc_node.saved_tensors
        spec, prob, rom = fwd["spec"], fwd["prob"], fwd["rom"]
        s = enc_node.shape
        P, B, S = q_all.shape[0] // 2, q_all.shape[1], u.shape[1]
        dev = q_all.device
        R = theta.shape[0]
        blackbox = spec.model == "dr_blackbox"
        shift = (0, 0, 0)
        off_n = off_row0 = 0
        if rom is not None:
            # (the sampled rows must not be rows the adjoint writes: their gradient is then exactly the conditioned rows'.
            # Other rows the adjoint leaves alone stay zero in the persistent buffer below)
            if (len(rom) != 4 or self.offset is None
                    or any(r not in spec.unwritten_rows for r in range(rom[0], rom[0] + rom[2]))):
                return None
            src, dst, n_off, _ = rom
            shift, off_n, off_row0 = (src, n_off, dst - src), n_off, dst
        elif self.offset is not None:
            return None
        L = hip.lib()
        logp = fwd["logp"]
        key = (B, S, R, str(dev))
        # ---- everything vihds_step_tail (or the bookkeeping below) would refuse is looked at HERE, before the adjoint is
        # queued: a refusal is a decline (None: the caller runs cost() + autograd + optimizer.step()), remembered per shape --
        # never an exception after vihds_ode_bwd_elbo has already run
        if key in self._declined:
            return None
        why = None
        if off_n > 0 and s.D <= 0:
            why = "an offset layer without device columns in the encoder's shape"
        elif off_n * (s.D + 1) > 256:  # (vihds_api.hip: the offset layer's update is one block of the update launch)
            why = "offset layer of %d x %d weights: more than one block of the update launch" % (off_n, s.D)
        elif weights is not None:
            fo = 0
            for t in self.flat_tensors:
                if t.data_ptr() != weights.data_ptr() + 4 * fo or not t.is_contiguous():
                    why = "the decoder's parameters are not views of its flat weight buffer"
                    break
                fo += t.numel()
            if (why is None and not blackbox and len(self._chunks_static()) != 1
                    and bool(L.vihds_ode_bwd_reduces_weights(ctypes.byref(prob)))):
                why = "the precision network's tensors are not one run of Adam's flat state"
        if why is None and any(t is not None and not t.is_contiguous() for t in self.tensors):
            why = "a non-contiguous encoder parameter"
        if why is not None:
            self._declined[key] = why
            return None
        # ---- work buffers: allocated once per shape OUTSIDE any capture (the warm-up steps come first) and reused
        capturing = torch.cuda.is_current_stream_capturing()
        if key not in self._bufs:
            n_aux = int(L.vihds_ode_bwd_aux_floats(ctypes.byref(prob)))
            lanes_reduce = (weights is not None and not blackbox and bool(L.vihds_ode_bwd_reduces_weights(ctypes.byref(prob))))
            if not blackbox and weights is None:
                n_aux = 0
            n_flat = sum(t.numel() for t in self.flat_tensors)
            n_all = sum(p.numel() for p in self.optimizer.param_groups[0]["params"] if p.requires_grad)
            Lc = s.L - s.K + 1
            bufs = {
                # every gradient, laid out like Adam's flat m / v (= model.parameters() order): a replica's all-reduce
                # takes the arena in place (parallel.allreduce_gradients)
                "arena": torch.empty(n_all, device=dev),
                "g_all": torch.empty((2 * P, B), device=dev), "g_pre": torch.empty((B, s.H), device=dev),
                "g_conv": torch.empty((B, s.F, Lc), device=dev),
                # rows the adjoint never writes (parameters the integrator does not read) stay zero for good
                "g_theta": torch.zeros((R, B, S), device=dev),
                "aux": torch.empty(max(n_aux, 1), device=dev), "n_aux": n_aux, "lanes_reduce": lanes_reduce,
                "log_w": torch.empty((B, S), device=dev), "rows": torch.empty((3, B), device=dev),
                "loss": torch.empty((), device=dev), "ug": torch.empty((B, S), device=dev),
                "g_w": torch.zeros(max(n_flat, 1), device=dev),
                "rowsum": torch.empty((max(off_n, 1), B), device=dev),
            }
            if capturing:
                raise RuntimeError("GeneralTail: first use of a shape inside a graph capture (the warm-up steps allocate)")
            self._bufs[key] = bufs
        bf = self._bufs[key]
        logp_c, log_p_c, log_q_c = _c(logp.detach()), _c(log_p.detach()), _c(log_q.detach())
        rows = bf["rows"]
        # ---- the ODE adjoint, its log-likelihood gradient = the importance weights broadcast over the four signals
        prob.logp_grad_broadcast = 1
        g_theta = bf["g_theta"]
        aux = bf["aux"] if bf["n_aux"] > 0 else None
        g_w = bf["g_w"] if (weights is not None and not blackbox and not bf["lanes_reduce"]) else None
        if g_w is not None:
            g_w.zero_()  # (thread-per-trajectory kernels ADD the bias gradients; the contraction below adds the matrices)
        if self.inkernel_iwae:
            # the weights are formed inside the adjoint launch (vihds_ode_bwd_elbo): no IWAE launch in this step; the tail's
            # rows kernel recomputes them for the theta adjoint and leaves log_w, lse and -ELBO
            rc = _launch("ode_bwd", lambda: L.vihds_ode_bwd_elbo(
                ctypes.byref(prob), hip.ptr(theta), hip.ptr(cond), hip.ptr(dev1hot), hip.ptr(times), hip.ptr(obs),
                hip.ptr(weights), hip.ptr(traj), hip.ptr(logp_c), hip.ptr(log_p_c), hip.ptr(log_q_c), hip.ptr(g_theta),
                hip.ptr(g_w), hip.ptr(aux), hip.current_stream()))
            hip.check(rc, "vihds_ode_bwd_elbo")
        else:
            # IWAE launch: loss, log_w, lse and the unit-upstream gradient d loss / d log_w (training.py:135-149)
            ticket = _iwae_ticket(dev)
            rc = L.vihds_iwae_loss_fwd(B, S, int(n_total), hip.ptr(logp_c), hip.ptr(log_p_c), hip.ptr(log_q_c),
                                       hip.ptr(bf["log_w"]), hip.ptr(rows[0]), hip.ptr(rows[1]), hip.ptr(rows[2]),
                                       hip.ptr(bf["loss"]), hip.ptr(bf["ug"]), None, hip.ptr(ticket), hip.current_stream())
            hip.check(rc, "vihds_iwae_loss_fwd")
            rc = _launch("ode_bwd", lambda: L.vihds_ode_bwd(
                ctypes.byref(prob), hip.ptr(theta), hip.ptr(cond), hip.ptr(dev1hot), hip.ptr(times), hip.ptr(obs),
                hip.ptr(weights), hip.ptr(traj), None, None, hip.ptr(bf["ug"]), hip.ptr(g_theta), hip.ptr(g_w), hip.ptr(aux),
                hip.current_stream()))
            hip.check(rc, "vihds_ode_bwd")
        extras = []  # (flat offset, size, grad_src tensor, src offset, map, nparts, stride)
        chunks = self._chunks_static()
        side = None
        if weights is not None:
            if bf["lanes_reduce"]:
                NIN = spec.n_species + 1
                nwg = 4 * (2 * NIN + 2)
                nblk = bf["n_aux"] // nwg
                assert len(chunks) == 1
                extras = [(0, chunks[0][3], aux, 0, self._lane_map(NIN, dev), nblk, nwg)]
            else:
                # the weight-gradient contraction / reductions (dr_blackbox: Gram-tile sums + the tail launch, ~19 us) feed
                # only the tail's UPDATE launch: they run on a second stream beside the rows launch and join before the update
                # (in a captured step: a parallel branch of the hipGraph)
                main = torch.cuda.current_stream()
                side = self._side_stream(dev)
                # (a THIRD stream for dr_blackbox's tail launch beside its Gram reduce was measured slower: 0.380 against 0.360
                # ms/step at config 4, three runs each on one box)
                side.wait_stream(main)
                with torch.cuda.stream(side):
                    if blackbox:
                        gw = blackbox_weight_grads(spec, prob, aux, theta, cond, dev1hot)
                    else:
                        gw = neural_precision_weight_grads(spec, prob, aux, g_w)
                extras = [(fo, size, gw, fo, None, 1, 0) for (_f, _c2, fo, size) in chunks]
        # ---- the tail
        opt = self.optimizer
        group = opt.param_groups[0]
        st = opt._group_state(0, group)
        offsets, off = {}, 0
        for prm in st["params"]:
            offsets[id(prm)] = off
            off += prm.numel()
        arena = bf["arena"]

        def gview(t):
            return arena[offsets[id(t)]:offsets[id(t)] + t.numel()].view(t.shape)

        a = hip.StepTailArgs()
        a.P, a.S = P, S
        a.kind, a.q_all, a.q_rows = hip.ptr(kind), hip.ptr(q_all), hip.ptr(q_rows)
        a.p_mu, a.p_prec, a.clip_lo, a.clip_hi = hip.ptr(p_mu), hip.ptr(p_prec), hip.ptr(clip_lo), hip.ptr(clip_hi)
        a.u, a.g_theta_unit = hip.ptr(u), hip.ptr(g_theta)
        a.iwae.logp, a.iwae.log_p, a.iwae.log_q = hip.ptr(logp_c), hip.ptr(log_p_c), hip.ptr(log_q_c)
        a.iwae.n_iwae_total = int(n_total)
        a.iwae.log_w, a.iwae.lse, a.iwae.loss = hip.ptr(bf["log_w"]), hip.ptr(rows[2]), hip.ptr(bf["loss"])
        a.g_all = hip.ptr(bf["g_all"])
        a.delta_obs, a.inputs, a.dev1hot = hip.ptr(delta_obs), hip.ptr(inputs), hip.ptr(dev_1hot_e if dev_1hot_e is not None else dev1hot)
        a.lin_w, a.local_w, a.pooled, a.hidden = hip.ptr(lin_w), hip.ptr(local_w), hip.ptr(pooled), hip.ptr(hidden)
        a.g_pre, a.g_conv = hip.ptr(bf["g_pre"]), hip.ptr(bf["g_conv"])
        grads = []
        for k, t in enumerate(self.tensors):
            if t is None:
                continue
            if not t.is_contiguous():
                raise RuntimeError("vihds_step_tail needs contiguous parameters")
            view = gview(t)
            grads.append((t, view))
            a.param[k], a.grad[k], a.mv_offset[k] = t.data_ptr(), view.data_ptr(), offsets[id(t)]
        a.g_theta_weighted = 1
        a.rng_advance = hip.ptr(fwd["rng_advance"])  # (ThetaOdeFused read the generator's step and left the increment to us)
        a.g_shift_lo, a.g_shift_n, a.g_shift = shift
        flat0 = weights.data_ptr() if weights is not None else 0
        a.n_extra = len(extras)
        keep = []
        for x, (fo, size, src, so, mp, nparts, stride) in enumerate(extras):
            first = chunks[x][0]
            e = a.extra[x]
            e.param = flat0 + 4 * fo
            e.grad = arena.data_ptr() + 4 * offsets[id(self.flat_tensors[first])]
            e.grad_src = src.data_ptr() + 4 * so
            e.map = hip.ptr(mp)
            e.size, e.nparts, e.part_stride = size, nparts, stride
            e.mv_offset = offsets[id(self.flat_tensors[first])]
            keep.append(src)
        fo = 0
        for t in self.flat_tensors:
            if t.data_ptr() != flat0 + 4 * fo:
                raise RuntimeError("GeneralTail: the decoder's parameters are not views of its flat weight buffer")
            grads.append((t, gview(t)))
            fo += t.numel()
        if off_n > 0:
            W, bvec = self.offset.weight, self.offset.bias
            a.off_n, a.off_row0 = off_n, off_row0
            a.off_w, a.off_b = W.data_ptr(), bvec.data_ptr()
            gW, gb = gview(W), gview(bvec)
            a.off_gw, a.off_gb = gW.data_ptr(), gb.data_ptr()
            a.off_mv_w, a.off_mv_b = offsets[id(W)], offsets[id(bvec)]
            a.off_rowsum = hip.ptr(bf["rowsum"])
            grads += [(W, gW), (bvec, gb)]
        lr = group["lr"]
        a.m, a.v, a.state = st["m"].data_ptr(), st["v"].data_ptr(), (st["state"].data_ptr() if apply_adam else None)
        a.lr_dev = hip.ptr(lr) if isinstance(lr, torch.Tensor) else None
        a.lr = 0.0 if isinstance(lr, torch.Tensor) else float(lr)
        a.beta1, a.beta2 = group["betas"]
        a.eps = group["eps"]
        if side is None:
            rc = _launch("step_tail", lambda: L.vihds_step_tail(ctypes.byref(s), ctypes.byref(a), hip.current_stream()))
            hip.check(rc, "vihds_step_tail")
        else:
            a.phase = 1
            rc = _launch("step_tail", lambda: L.vihds_step_tail(ctypes.byref(s), ctypes.byref(a), hip.current_stream()))
            hip.check(rc, "vihds_step_tail (rows)")
            torch.cuda.current_stream().wait_stream(side)
            # (the gradient buffers were allocated on the side stream and are read on this one: they are kept until the next
            # step's side-stream work -- which waits for this stream first -- could be handed their memory again)
            self._keep_alive = keep
            a.phase = 2
            rc = L.vihds_step_tail(ctypes.byref(s), ctypes.byref(a), hip.current_stream())
            hip.check(rc, "vihds_step_tail (update)")
        for t, v in grads:
            t.grad = v
        return bf["loss"]

    def _side_stream(self, dev):
        key = str(dev)
        if key not in self._sides:
            self._sides[key] = torch.cuda.Stream(device=dev)
        return self._sides[key]


class IwaeLossSharded(torch.autograd.Function):
    """-ELBO with the S axis sharded over the ranks of `group`: rows kernel, ONE all-gather of the [2,B]
    (max, sum-exp) pairs (a graph break when the step is being captured), then one combine launch giving the global
    lse, the loss and this rank's unit-gradient backward.  Backward: no communication (the local softmax weights only
    need the global lse)."""

    @staticmethod
    def forward(ctx, logp, log_p, log_q, n_total, group):
        import torch.distributed as dist

        from vihds.parallel import graph_break

        _require_cuda(logp, log_p, log_q)
        logp, log_p, log_q = _c(logp), _c(log_p), _c(log_q)
        _, B, S = logp.shape
        dev = logp.device
        world = dist.get_world_size(group)
        log_w = torch.empty((B, S), device=dev, dtype=torch.float32)
        pair = torch.empty((2, B), device=dev, dtype=torch.float32)  # row_max ; row_sumexp
        rc = hip.lib().vihds_iwae_fwd(B, S, hip.ptr(logp), hip.ptr(log_p), hip.ptr(log_q), hip.ptr(log_w),
                                      hip.ptr(pair[0]), hip.ptr(pair[1]), hip.current_stream())
        hip.check(rc, "vihds_iwae_fwd")
        gathered = torch.empty((world * 2, B), device=dev, dtype=torch.float32)
        graph_break(lambda: dist.all_gather_into_tensor(gathered, pair, group=group))
        lse = torch.empty((B,), device=dev, dtype=torch.float32)
        loss = torch.empty((), device=dev, dtype=torch.float32)
        ug = ugn = None
        if any(ctx.needs_input_grad):
            ug = torch.empty((B, S), device=dev, dtype=torch.float32)
            ugn = torch.empty((B, S), device=dev, dtype=torch.float32) if log_q is not None else None
        rc = hip.lib().vihds_iwae_combine(world, B, S, int(n_total), hip.ptr(gathered), hip.ptr(log_w), hip.ptr(lse),
                                          hip.ptr(loss), hip.ptr(ug), hip.ptr(ugn), hip.current_stream())
        hip.check(rc, "vihds_iwae_combine")
        ctx.save_for_backward(log_w, lse, ug, ugn)
        ctx.has = (log_p is not None, log_q is not None)
        ctx.mark_non_differentiable(log_w, lse)
        ctx.set_materialize_grads(False)
        return loss, log_w, lse

    @staticmethod
    def backward(ctx, g_loss, _g1, _g2):  # vihds_iwae_loss_bwd with the global lse, or the unit-gradient buffers
        return IwaeLoss.backward(ctx, g_loss, _g1, _g2)


def device_condition(z, dev_1hot, relevance, is_default, out, w_mean, w_std, rng_state=None, sample_window=None):
    """OdeModel.device_conditioner applied to ones for E parameters in one launch, written into `out` [E,B,S].
    z [E,D] standard normals, or None with `rng_state` (KernelNormal.new_state): the kernel draws them."""
    _require_cuda(dev_1hot, relevance, is_default, out)
    if z is None and rng_state is None:
        raise ValueError("device_condition needs z or rng_state")
    E, B, S = out.shape
    z = None if z is None else _c(z)
    S_total, s_off = sample_window if sample_window is not None else (S, 0)  # this rank's slice of the global S axis
    rc = hip.lib().vihds_device_condition(E, B, S, S_total, s_off, dev_1hot.shape[1], float(w_mean), float(w_std), hip.ptr(z),
                                          hip.ptr(rng_state), hip.ptr(_c(dev_1hot)), hip.ptr(relevance),
                                          hip.ptr(is_default), hip.ptr(out), hip.current_stream())
    hip.check(rc, "vihds_device_condition")
    return out


def iwae_loss(logp, log_p, log_q, n_iwae_total=None, group=None, defer=False):
    """-ELBO exactly as Training.cost forms it (vihds/training.py:144-149).  defer: see IwaeLoss.forward."""
    if group is None:
        S = n_iwae_total if n_iwae_total is not None else logp.shape[2]
        return IwaeLoss.apply(logp, log_p, log_q, S, defer)
    S = n_iwae_total if n_iwae_total is not None else logp.shape[2]
    return IwaeLossSharded.apply(logp, log_p, log_q, S, group)


OBSERVE_KINDS = {"default": 0, "direct": 1, "inducer": 2}  # VIHDS_OBS_* of include/vihds_hip.h


def ode_logp_only(spec, theta, cond, times, obs, dev1hot, weights):
    """The forward launch with the per-species log-likelihood as its ONLY output ([4,B,S]; no trajectory, no x_predict, no
    autograd node): the first pass of an evaluation that takes its summaries from ode_fwd_summaries."""
    _require_cuda(theta, cond, times, obs)
    theta, cond, times, obs = _c(theta), _c(cond), _c(times), _c(obs)
    R, B, S = theta.shape
    if R != spec.n_rows:
        raise RuntimeError("theta has %d rows, problem expects %d" % (R, spec.n_rows))
    prob = spec.bind(B, S, times.shape[0])
    logp = torch.empty((4, B, S), device=theta.device, dtype=torch.float32)
    rc = _launch("ode_fwd", lambda: hip.lib().vihds_ode_fwd(
        ctypes.byref(prob), hip.ptr(theta), hip.ptr(cond), hip.ptr(dev1hot), hip.ptr(times), hip.ptr(obs),
        hip.ptr(weights), None, None, hip.ptr(logp), hip.current_stream()))
    hip.check(rc, "vihds_ode_fwd")
    return logp


def ode_fwd_summaries_supported(spec, B, S, T):
    """True when the evaluation pass of this problem can take its summaries from a second forward launch
    (vihds_ode_fwd_summaries) instead of writing the trajectory and streaming it back."""
    return bool(hip.lib().vihds_ode_fwd_summaries_supported(ctypes.byref(spec.bind(B, S, T))))


def ode_fwd_summaries(spec, theta, cond, times, dev1hot, weights, log_w, lse):
    """Results.init's importance-weighted summaries (vihds/utils.py:79-99) from a SECOND forward integration with the
    normalised weights known, added up on the way: the trajectory never goes through HBM (include/vihds_hip.h,
    vihds_ode_fwd_summaries).  theta [R,B,S] packed as for OdeSolveObserve; log_w [B,S], lse [B].
    Returns (iw_predict_mu [B,4,T], iw_predict_std [B,4,T], iw_states [B,n_species,T], iw_variance [B,4,T])."""
    _require_cuda(theta, times, log_w, lse)
    theta, cond, times, log_w, lse = _c(theta), _c(cond), _c(times), _c(log_w), _c(lse)
    R, B, S = theta.shape
    T = times.shape[0]
    prob = spec.bind(B, S, T)
    n_ws = hip.lib().vihds_ode_fwd_summaries_workspace_floats(ctypes.byref(prob))
    if n_ws < 0:
        hip.check(int(n_ws), "vihds_ode_fwd_summaries_workspace_floats")
    n_species = spec.n_species
    dev = theta.device
    ws = torch.empty(int(n_ws), device=dev, dtype=torch.float32)
    mu = torch.empty((B, 4, T), device=dev)
    sd = torch.empty((B, 4, T), device=dev)
    st = torch.empty((B, n_species, T), device=dev)
    var = torch.empty((B, 4, T), device=dev)
    rc = _launch("ode_fwd_summaries", lambda: hip.lib().vihds_ode_fwd_summaries(
        ctypes.byref(prob), hip.ptr(theta), hip.ptr(cond), hip.ptr(dev1hot), hip.ptr(times), hip.ptr(weights),
        hip.ptr(log_w), hip.ptr(lse), hip.ptr(ws), hip.ptr(mu), hip.ptr(sd), hip.ptr(st), hip.ptr(var),
        hip.current_stream()))
    hip.check(rc, "vihds_ode_fwd_summaries")
    return mu, sd, st, var


def iw_summaries(log_w, lse, traj, xpred, n_species, theta=None, prec_rows=None, observe_kind="default"):
    """Results.init's importance-weighted summaries on device (vihds/utils.py:79-99).  xpred None: the observed signals
    are formed from the trajectory by the model's observation map (observe_kind) inside the kernel."""
    _require_cuda(log_w, lse, traj)
    # (the kernels read [T][N][B][S] / [T][4][B][S] storage through raw pointers)
    traj, xpred = _c(traj), _c(xpred)
    T, N, B, S = traj.shape
    dev = traj.device
    mu = torch.empty((B, 4, T), device=dev)
    sd = torch.empty((B, 4, T), device=dev)
    st = torch.empty((B, n_species, T), device=dev)
    var = torch.empty((B, 4, T), device=dev)
    rows = (ctypes.c_int * 4)(*prec_rows) if prec_rows is not None else None
    if xpred is None:
        rc = hip.lib().vihds_iw_summaries_states(B, S, T, N, n_species, OBSERVE_KINDS[observe_kind], hip.ptr(_c(log_w)),
                                                 hip.ptr(_c(lse)), hip.ptr(traj), hip.ptr(theta), rows, hip.ptr(mu),
                                                 hip.ptr(sd), hip.ptr(st), hip.ptr(var), hip.current_stream())
        hip.check(rc, "vihds_iw_summaries_states")
        return mu, sd, st, var
    _require_cuda(xpred)
    rc = hip.lib().vihds_iw_summaries(B, S, T, N, n_species, hip.ptr(_c(log_w)), hip.ptr(_c(lse)), hip.ptr(traj),
                                      hip.ptr(xpred), hip.ptr(theta), rows, hip.ptr(mu), hip.ptr(sd), hip.ptr(st),
                                      hip.ptr(var), hip.current_stream())
    hip.check(rc, "vihds_iw_summaries")
    return mu, sd, st, var
