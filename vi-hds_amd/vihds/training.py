"""Training loop and the IWAE cost (counterpart of the reference's vihds/training.py).

`Training.cost` keeps the reference signature.  Its three reductions -- Gaussian observation log-likelihood over
time (fused into the ODE kernel), log p(theta) - log q(theta) (fused into the theta kernel) and the
importance-weight logsumexp (vihds_iwae_fwd) -- all run in HIP kernels.
"""
import functools
import math
import os
import time
import weakref

import numpy as np
import torch
from torch.utils.data import DataLoader

from vihds import hostdraws, ops, parallel
from vihds.encoders import LocalAndGlobal
from vihds.utils import Results, TrainingLogData, attrify, default_get_value, variable_summaries


def log_prob_gaussian(x_obs, x_predict, precisions):
    """reference training.py:41-44 (generic torch fallback for callers that bring their own tensors; still GPU)."""
    return -0.5 * (math.log(2.0 * math.pi) - precisions.log() + precisions * (x_predict - x_obs).pow(2))


def log_prob_observations(model, x_predict, x_obs, precisions, use_laplace=False):
    """reference training.py:24-33 -> [B,S,4]."""
    if use_laplace:
        raise NotImplementedError("Laplace likelihood is dead code in the reference (training.py:36-38)")
    return torch.sum(log_prob_gaussian(torch.unsqueeze(x_obs, 1), x_predict, precisions), 3)


def _delta_obs(obs):
    return (obs[:, :, 1:] - obs[:, :, :-1]).contiguous()


def batch_to_device(times, device, d):
    """reference training.py:47-52"""
    d["times"] = times.to(device)
    d["dev_1hot"] = d["dev_1hot"].to(device)
    d["inputs"] = d["inputs"].to(device)
    d["observations"] = d["observations"].to(device).contiguous()
    return attrify(d)


def collate_merged(times, device, batch):
    """reference training.py:55-68"""
    dd = {
        "devices": torch.stack([torch.tensor(b["devices"]) for b in batch]),
        "dev_1hot": torch.stack([torch.as_tensor(b["dev_1hot"], dtype=torch.float32) for b in batch]),
        "inputs": torch.stack([torch.as_tensor(b["inputs"], dtype=torch.float32) for b in batch]),
        "observations": torch.stack([torch.as_tensor(b["observations"], dtype=torch.float32) for b in batch]),
    }
    return batch_to_device(times, device, dd)


def _shuffled_index_batches(n, batch_size):
    """One epoch of DataLoader(range(n), batch_size, shuffle=True) WITHOUT the loader: the same two draws from torch's
    default generator in the same order (the loader iterator's base seed, torch/utils/data/dataloader.py
    _BaseDataLoaderIter.__init__; RandomSampler's seed for its own generator, sampler.py RandomSampler.__iter__), the same
    permutation, cut into consecutive batches.  (The loader's per-batch machinery -- a profiler scope, the fetcher, the
    collate call -- cost 0.24 ms per epoch of seven batches, half of the epoch's GPU time: measured with cProfile in round 5, profiles/LOG.md.)
    Training checks it against the loader itself once (_index_batches_match_loader) and keeps the loader if a torch
    version draws differently."""
    torch.empty((), dtype=torch.int64).random_()
    seed = int(torch.empty((), dtype=torch.int64).random_().item())
    g = torch.Generator()
    g.manual_seed(seed)
    perm = torch.randperm(n, generator=g)
    return [perm[i:i + batch_size] for i in range(0, n, batch_size)]


def _index_batches_match_loader(loader, n, batch_size):
    """True when two epochs of _shuffled_index_batches equal two passes over `loader` from the same generator state, batches
    AND the state the default generator is left in (the state is restored afterwards: nothing is consumed)."""
    keep = torch.get_rng_state()
    try:
        a = [list(loader), list(loader)]
        end_a = torch.get_rng_state()
        torch.set_rng_state(keep)
        b = [_shuffled_index_batches(n, batch_size), _shuffled_index_batches(n, batch_size)]
        end_b = torch.get_rng_state()
        same = torch.equal(end_a, end_b) and all(
            len(x) == len(y) and all(isinstance(u, torch.Tensor) and u.dtype == v.dtype and torch.equal(u, v) for u, v in zip(x, y))
            for x, y in zip(a, b))
    except Exception:  # noqa: BLE001 (anything unexpected: keep the loader)
        same = False
    finally:
        torch.set_rng_state(keep)
    return bool(same)


class _EpochLook(object):
    """params.epoch_lookahead: "does this epoch hold a NaN loss" as ONE flag per epoch that travels to pinned host memory
    behind the epoch's own graph launch (two slots, used in turn), with an event: the loop can queue the NEXT epoch's launch
    and only then wait for this flag -- reading the losses themselves would wait for whatever has been queued behind them,
    and the graph's loss buffers are the next replay's as well."""

    def __init__(self):
        self.host = torch.zeros(2, dtype=torch.bool).pin_memory()
        self.events = [torch.cuda.Event(), torch.cuda.Event()]
        self.k = 0

    def post(self, losses):
        i = self.k
        self.k ^= 1
        flag = (~torch.isfinite(torch.stack([l.detach().reshape(()) for l in losses]))).any()
        self.host[i: i + 1].copy_(flag.reshape(1), non_blocking=True)
        self.events[i].record()
        return i

    def take(self, i):
        self.events[i].synchronize()
        return bool(self.host[i])


class _RowIndices(torch.utils.data.Dataset):
    """range(n) as a dataset: what DataLoader shuffles when the rows themselves already live on the device."""

    def __init__(self, n):
        self.n = int(n)

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        return int(i)


class _ElboLook(object):
    """The per-step look at the ELBO (reference training.py:331), one step late and without draining the stream: `push`
    queues a copy of the step's ELBO into one of two pinned slots and an event behind it, `take` waits for that event only."""

    def __init__(self):
        self.host = torch.empty(2, dtype=torch.float32).pin_memory()
        self.slots = [self.host[0:1], self.host[1:2]]
        self.events = [torch.cuda.Event(), torch.cuda.Event()]
        self.pos, self.pending = 0, None
        self._src, self._view = None, None

    def push(self, elbo):
        """Queue the copy of `elbo` (a device scalar); returns the slot pushed before, not looked at yet (or None)."""
        i, self.pos = self.pos, self.pos ^ 1
        if elbo is not self._src:  # (a replayed graph hands back the same static tensor every step: one view, kept)
            self._src, self._view = elbo, elbo.detach().reshape(1)
        self.slots[i].copy_(self._view, non_blocking=True)
        self.events[i].record()
        prev, self.pending = self.pending, i
        return prev

    def take(self, i):
        self.events[i].synchronize()
        return float(self.host[i])


class Training:
    """Orchestrates IWAE training of the VAE (reference training.py:71-383)."""

    def __init__(self, args, settings, data, parameters, model):
        self.args = args
        self.settings = settings
        self.dataset_pair = data
        self.model = model
        self.shard = getattr(model, "shard", None)
        self.replica = getattr(model, "replica", None)  # parallel.RowReplica: gradients averaged over ranks
        p = settings.params
        on_gpu = settings.device.type == "cuda"
        # hip_graph: true / false, or (the default, None) automatic: on the GPU whenever the step holds no host round trip --
        # i.e. not with an adaptive solver, whose controller reports back to the host.  Replaying the step changes no number;
        # host-side random streams are fed through vihds/hostdraws.py
        want_graph = default_get_value(p, "hip_graph", None)
        if want_graph is None:
            from vihds import hip as _hip

            want_graph = p.solver not in _hip.ADAPTIVE_SOLVERS and os.environ.get("VIHDS_AUTO_GRAPH", "1") != "0"
        self._graph_auto = default_get_value(p, "hip_graph", None) is None  # (automatic: a capture that fails falls back to eager launches)
        self.use_graph = bool(want_graph) and on_gpu
        self._pending_elbo = None
        self._elbo_look = None
        self.nan_check_every = int(default_get_value(p, "nan_check_every", 1))  # 0 = never check
        # run(): with hip_graph, a single process and a NaN check at most once per epoch, an epoch is ONE graph launch
        # (run(): queue an epoch's graph launch before the look at the previous epoch's losses -- see run())
        self.epoch_lookahead = bool(default_get_value(p, "epoch_lookahead", False)) and self.use_graph
        self._epoch_look = _EpochLook() if self.epoch_lookahead else None
        self.epoch_graph = (self.use_graph and bool(default_get_value(p, "epoch_graph", True)) and self.shard is None
                            and self.replica is None)
        # (a captured step that keeps the reference's host-side streams -- u_rng: numpy, conditioner_rng: cpu, the defaults --
        # reads the draws from static device buffers that are refreshed before every replay: vihds/hostdraws.py)
        # capturable Adam keeps step counts on the device => the whole step can live in one hipGraph
        self.lr = torch.tensor(float(p.learning_rate), device=settings.device) if self.use_graph else p.learning_rate
        # one launch for the whole update, step counter on the device (vihds/optim.py)
        if on_gpu:
            from vihds.optim import HipAdam

            self.optimizer = HipAdam(model.parameters(recurse=True), lr=self.lr,
                                     grad_scale=1.0 / self.replica.world if self.replica is not None else 1.0)
        else:  # host-side construction only (CPU unit tests of the control flow); the decoder itself has no CPU path
            self.optimizer = torch.optim.Adam(model.parameters(recurse=True), lr=self.lr)
        self.scheduler = torch.optim.lr_scheduler.MultiStepLR(self.optimizer, p.learning_boundaries,
                                                              gamma=p.learning_gamma)
        n_vals = LocalAndGlobal.from_list(parameters.get_parameter_counts())
        self.model.n_theta = n_vals.sum()
        self.n_batch = min(p.n_batch, data.n_train)
        self.train_data = batch_to_device(data.train.dataset.times, settings.device,
                                          data.train.dataset[data.train.indices])
        for k in ("dev_1hot", "inputs", "observations"):  # (vihds_gather_batch reads them as fp32 rows)
            self.train_data[k] = self.train_data[k].float().contiguous()
        self.valid_data = batch_to_device(data.test.dataset.times, settings.device,
                                          data.test.dataset[data.test.indices])
        # The training rows are resident on the device as a whole (self.train_data); the loader only decides WHICH rows make
        # up a batch.  It is the reference's DataLoader(shuffle=True) (training.py:108-113) over a dataset of the same
        # length, so it consumes torch's generator exactly as the reference's loader does (same shuffles under the same
        # seed: tests/golden/trace_*.npz), but yields row indices; the rows are gathered on the device
        # (vihds_gather_batch) instead of being stacked sample by sample on the host every step.
        self.train_loader = DataLoader(dataset=_RowIndices(data.n_train if hasattr(data, "n_train") else len(data.train)),
                                       batch_size=self.n_batch, shuffle=True,
                                       collate_fn=lambda rows: torch.tensor(rows, dtype=torch.int64))
        n_train = len(self.train_loader.dataset)
        self._fast_index_batches = _index_batches_match_loader(self.train_loader, n_train, self.n_batch)
        self.host_loader = DataLoader(
            dataset=data.train, batch_size=self.n_batch, shuffle=True,
            collate_fn=functools.partial(collate_merged, data.train.dataset.times, settings.device),
        )  # (the reference's own loader: kept for callers that want host-built batches; Training.run does not use it)
        if settings.trainer is not None:
            held_out_name = args.heldout or "%d_of_%d" % (args.split, args.folds)
            self.train_path = os.path.join(settings.trainer.tb_log_dir, "train_%s" % held_out_name)
            self.valid_path = os.path.join(settings.trainer.tb_log_dir, "valid_%s" % held_out_name)
            os.makedirs(self.train_path, exist_ok=True)
            os.makedirs(self.valid_path, exist_ok=True)
        else:
            self.train_path = self.valid_path = None
        self.empty_cache = True
        # the best evaluation's results go to the on-disk cache (utils.py:127-141 in the reference) at every improvement, as
        # the reference does, or -- lazy_cache_dump -- once, when run() leaves its loop (also on an exception): same final cache
        self.lazy_cache_dump = bool(default_get_value(p, "lazy_cache_dump", False))
        self._best_output = None
        # the step's tail (loss, backward, Adam) as two launches: vihds_step_tail (params.fused_step_tail, default on)
        self.fused_tail = bool(default_get_value(p, "fused_step_tail", True)) and on_gpu
        self._tail, self._tail_ok, self._tail_shapes = None, False, {}
        self._gtail, self._gtail_ok = None, None  # ops.GeneralTail (any model); None: not decided yet
        self._gtail_declined = set()  # (batch shape, n_iwae) pairs GeneralTail.launch has declined
        if on_gpu:
            self.optimizer.gate = None
        self._graphs = {}
        self.collectives_captured = False  # multi-rank captured steps: were the collectives recorded inside the graph?
        self._eval_graphs = {}
        # evaluation passes replayed from a hipGraph (with hip_graph; Training.evaluate)
        self.eval_graph = bool(default_get_value(p, "eval_graph", True))
        self._staged = {}
        self._grad_buffer = None
        self._one = None
        self._steps = 0

    # ------------------------------------------------------------------------------------------------
    def cost(self, batch_data, batch_results, theta, q, p, full_output=False, writer=None, epoch=None):
        """reference training.py:127-174.  Returns {"elbo": -ELBO} (sic) or a Results object."""
        fused = getattr(batch_results, "log_p_by_species", None)
        sol = getattr(batch_results, "solution", None)
        if fused is None or (full_output and sol is None):  # (the kernel paths never materialise these)
            x_states, x_predict, precisions = batch_results
        if fused is not None:
            logp = batch_results.solution.logp_buffer  # [4,B,S] straight from the ODE kernel
            log_p_by_species = fused
        else:
            log_p_by_species = log_prob_observations(self.model, x_predict, batch_data.observations, precisions,
                                                     self.settings.params.use_laplace)
            logp = log_p_by_species.permute(2, 0, 1).contiguous()
        log_q_theta = q.log_prob(theta)
        log_p_theta = p.log_prob(theta)
        n_local = logp.shape[2]
        n_iwae = n_local * (self.shard.world if self.shard is not None else 1)
        group = (self.shard.group or torch.distributed.group.WORLD) if self.shard is not None else None
        # (fused decoder step + params.fused_iwae_backward: the loss is evaluated inside the step's theta-adjoint launch,
        # i.e. `iwae_cost` holds its value once backward() has run -- Training.step returns it after that)
        # Only inside Training.step, which is known to run backward() with the unit seed right after: a caller that
        # evaluates cost() on its own gets the loss value immediately, as in the reference.
        defer = (not full_output and group is None and torch.is_grad_enabled() and getattr(self, "_in_step", False)
                 and getattr(getattr(batch_results, "solution", None), "defer_iwae", False))
        iwae_cost, log_unnormalized_iws, lse = ops.iwae_loss(logp, log_p_theta, log_q_theta, n_iwae_total=n_iwae,
                                                             group=group, defer=defer)
        if not full_output:
            return attrify({"elbo": iwae_cost})
        elbo = -iwae_cost
        if writer is not None:
            normalized_iws = (log_unnormalized_iws - lse[:, None]).exp()
            self._update_summaries(writer, epoch, q, log_unnormalized_iws, normalized_iws, logp.sum(0),
                                   log_p_by_species, elbo, log_p_theta, log_q_theta)
        output = Results()
        ode_model = self.model.decoder.ode_model
        online = getattr(sol, "online_summaries", None) if self.shard is None else None
        if online is not None:  # (evaluation without the trajectory's round trip: OdeModel._solve_for_evaluation)
            summ = online(log_unnormalized_iws.detach(), lse.detach())
        elif sol is not None:
            traj = sol.traj_buffer.detach()
            # (a solution without a stored x_predict -- params.lazy_x_predict -- has the kernel form it from the states)
            xpred = sol.xpred_buffer.detach() if getattr(sol, "has_x_predict", True) else None
            if ode_model.precisions.dynamic:  # the last four states are the precisions (reference precisions.py:89-94)
                summ = ops.iw_summaries(log_unnormalized_iws.detach(), lse.detach(), traj, xpred, traj.shape[1] - 4,
                                        observe_kind=ode_model.observe_kind)
            else:
                packed, row_of = theta.pack(ode_model.precisions.precision_vars)
                rows = [row_of[v] for v in ode_model.precisions.precision_vars]
                summ = ops.iw_summaries(log_unnormalized_iws.detach(), lse.detach(), traj, xpred, traj.shape[1],
                                        theta=packed.detach(), prec_rows=rows, observe_kind=ode_model.observe_kind)
        else:
            w = (log_unnormalized_iws - lse[:, None]).exp()[:, :, None, None]
            mu = (w * x_predict).sum(1)
            summ = (mu, ((w * (x_predict ** 2 + 1.0 / precisions)).sum(1) - mu ** 2).sqrt(), (w * x_states).sum(1),
                    (w / precisions).sum(1))
        if self.shard is not None:
            summ = parallel.combine_iw_summaries(summ, group)
        if full_output == "device":  # (Training.evaluate: the host side of Results is built outside the captured graph)
            return elbo, summ
        output.init_from_device(self.model.decoder.state_names, q, theta, elbo, summ)
        return output

    def _update_summaries(self, writer, epoch, q, log_unnormalized_iws, normalized_iws, log_p_observations,
                          log_p_by_species, elbo, log_p_theta, log_q_theta):
        """TensorBoard scalars (reference training.py:176-210)."""
        plot_histograms = self.settings.params.plot_histograms
        q.attach_summaries(writer, epoch, plot_histograms)
        row = min(1, log_unnormalized_iws.shape[0] - 1)
        variable_summaries(writer, epoch, log_unnormalized_iws[row, :], "IWS_unn_log", plot_histograms)
        variable_summaries(writer, epoch, normalized_iws[row, :], "IWS_normed", plot_histograms)
        writer.add_scalar("ELBO/elbo", elbo, epoch)
        writer.add_scalar("ELBO/log_p", log_p_observations.logsumexp(axis=1).mean(), epoch)
        for i, plot in enumerate(self.settings.data.signals):
            writer.add_scalar("ELBO/log_p_" + plot, log_p_by_species[:, :, i].logsumexp(axis=1).mean(), epoch)
        writer.add_scalar("ELBO/log_prior", log_p_theta.logsumexp(axis=1).mean(), epoch)
        writer.add_scalar("ELBO/loq_q", log_q_theta.logsumexp(axis=1).mean(), epoch)

    # ------------------------------------------------------------------------------------------------
    def evaluate(self, data, n_samples, writer=None, epoch=None):
        """One evaluation pass (reference training.py:283-307): forward without grad, cost(full_output=True) -> Results.
        With params.hip_graph and params.eval_graph (single process, no TensorBoard writer) the device side of the pass --
        encoder, sampling, conditioning, ODE forward, log-probs, IWAE weights, summaries, and the packing of everything
        Results copies to the host into one buffer -- is captured once per (data set, sample count) and replayed: the pass
        is some 25 launches whose host-side cost was a third of its time.  The draws come from the device generators, which
        advance inside the kernels, so every replay is a fresh evaluation, as in the eager pass."""
        ode_model = self.model.decoder.ode_model
        ode_model._no_online_summaries = self.shard is not None
        ode_model._evaluating = True  # (the summaries-on-the-way form of the pass is this method's: OdeModel._solve_for_evaluation)
        try:
            return self._evaluate(data, n_samples, writer, epoch)
        finally:
            ode_model._evaluating = False

    def _evaluate(self, data, n_samples, writer, epoch):
        dev_ok = (self.eval_graph and self.use_graph and writer is None and self.shard is None and self.replica is None)
        if not dev_ok:
            with torch.no_grad():
                results, theta, q, p = self.model(data, n_samples, writer=writer, epoch=epoch)
                return self.cost(data, results, theta, q, p, full_output=True, writer=writer, epoch=epoch)
        # keyed by the data object's identity; the entry HOLDS the object, so its id cannot be reused by another batch of
        # another shape while the graph that captured its tensors is cached; a caller that evaluates many
        # throw-away batches keeps at most eight captured passes alive
        key = (id(data), int(n_samples))
        if key not in self._eval_graphs:
            while len(self._eval_graphs) >= 8:
                self._eval_graphs.pop(next(iter(self._eval_graphs)))
            try:
                g, staged = self._capture_evaluation(data, int(n_samples))
            except Exception as exc:  # noqa: BLE001
                self._graphs_off(exc)
                self.eval_graph = False
                with torch.no_grad():
                    results, theta, q, p = self.model(data, n_samples, writer=writer, epoch=epoch)
                    return self.cost(data, results, theta, q, p, full_output=True, writer=writer, epoch=epoch)
            staged["data"] = data
            self._eval_graphs[key] = (g, staged)
        g, staged = self._eval_graphs[key]
        assert staged["data"] is data
        # the theta samples of the previous pass live in the graph's memory pool: if its Results is still around and nobody
        # has read them yet, they are copied (on the device) before this replay overwrites them -- a Results that was
        # dropped, the usual case, costs nothing
        last = staged["last_results"]() if staged.get("last_results") is not None else None
        if last is not None:
            last.detach_theta()
        # what Results holds on the host arrives in one of a few pinned buffers, used in turn, and the numpy members are views
        # of it: the Results that had this buffer before, if it is still alive, takes copies first
        ring = staged["host_ring"]
        k = staged["ring_pos"]
        staged["ring_pos"] = (k + 1) % len(ring)
        holder = staged["ring_refs"][k]() if staged["ring_refs"][k] is not None else None
        if holder is not None:
            holder.detach_host()
        hostdraws.replay(g)
        ring[k].copy_(staged["flat"], non_blocking=True)
        out = Results()
        out.init_from_staged(self.model.decoder.state_names, staged, ring[k])
        staged["last_results"] = weakref.ref(out)
        staged["ring_refs"][k] = weakref.ref(out)
        return out

    def _evaluation_device_side(self, data, n_samples):
        ode_model = self.model.decoder.ode_model
        was, ode_model._evaluating = getattr(ode_model, "_evaluating", False), True
        try:
            with torch.no_grad():
                results, theta, q, p = self.model(data, n_samples)
                elbo, summ = self.cost(data, results, theta, q, p, full_output="device")
        finally:
            ode_model._evaluating = was
        q_tensors = [t.detach() for t in q.get_tensors()]
        parts = q_tensors + [elbo.detach().reshape(1)] + [x.detach() for x in summ]
        flat = torch.cat([t.reshape(-1).float() for t in parts])  # ONE device->host transfer per pass
        rows = [t.detach() for t in theta.get_tensors()]
        return {"q_names": q.get_tensor_names(), "q_shapes": [tuple(t.shape) for t in q_tensors],
                "summary_shapes": [tuple(x.shape) for x in summ], "flat": flat, "theta_rows": rows,
                "last_results": None}

    def _capture_evaluation(self, data, n_samples):
        """Warm up on a side stream (generator states rolled back afterwards: the warm-up passes must not consume draws),
        then capture one evaluation pass.  Returns (graph, staged outputs living in the graph's memory pool)."""
        side = torch.cuda.Stream()
        snap = self._snapshot_training_state()
        side.wait_stream(torch.cuda.current_stream())
        draws = hostdraws.HostDraws()
        hostdraws.ACTIVE = draws  # (measuring what one pass draws on the host)
        try:
            with torch.cuda.stream(side):
                for _ in range(2):
                    draws.noted = 0
                    self._evaluation_device_side(data, n_samples)
        finally:
            hostdraws.ACTIVE = None
        torch.cuda.current_stream().wait_stream(side)
        self._restore_training_state(snap)
        if draws.noted:
            draws.reserve(draws.noted, data.observations.device)
        g = torch.cuda.CUDAGraph()
        hostdraws.ACTIVE = draws
        try:
            with torch.cuda.graph(g):
                staged = self._evaluation_device_side(data, n_samples)
        finally:
            hostdraws.ACTIVE = None
        g.host_draws = draws
        n = staged["flat"].numel()
        staged["host_ring"] = [torch.empty(n, dtype=torch.float32, pin_memory=True) for _ in range(4)]
        staged["ring_refs"], staged["ring_pos"] = [None] * 4, 0
        return g, staged

    def _evaluate_elbo_and_plot(self, epoch, log_data, train_writer, valid_writer):
        """reference training.py:267-322 (stdout format is parsed by the reference's tests/test_run_xval.py:55-60;
        figure plotting is out of scope)."""
        print("epoch %4d" % epoch, end="", flush=True)
        log_data.n_test += 1
        test_start = time.time()
        train_output = self.evaluate(self.train_data, self.args.train_samples, writer=train_writer, epoch=epoch)
        print(" | train (iwae-elbo = %0.4f, time = %0.2f, total = %0.2f)"
              % (train_output.elbo, log_data.total_train_time / epoch, log_data.total_train_time), end="", flush=True)
        valid_output = self.evaluate(self.valid_data, self.args.test_samples, writer=valid_writer, epoch=epoch)
        for w in (train_writer, valid_writer):
            if w is not None:
                w.flush()
        log_data.total_test_time += time.time() - test_start
        print(" | val (iwae-elbo = %0.4f, time = %0.2f, total = %0.2f)"
              % (valid_output.elbo, log_data.total_test_time / log_data.n_test, log_data.total_test_time))
        # plain floats: a Results' numpy members may be views of a staging buffer a later pass rewrites (utils.Results)
        if float(valid_output.elbo) > log_data.max_val_elbo:
            log_data.max_val_elbo = float(valid_output.elbo)
            if self.lazy_cache_dump:
                self._best_output = valid_output  # written once, when run() leaves its loop (nine files per improvement otherwise)
            else:
                valid_output.dump()
            self.empty_cache = False
        log_data.training_elbo_list.append(float(train_output.elbo))
        log_data.validation_elbo_list.append(float(valid_output.elbo))
        return valid_output

    # ------------------------------------------------------------------------------------------------
    def step(self, batch, zero_grad=True):
        """One ELBO training step on a device batch: forward, cost, backward, (gradient all-reduce), Adam.
        Returns the loss tensor (-ELBO) without synchronising."""
        ode = self.model.decoder.ode_model
        # (a step whose backward is ops.GeneralTail reads neither trajectory views nor x_predict: the forward skips the latter)
        # (a batch shape the tail has declined once -- LDS budget, offset layer, row layout -- keeps the ordinary forward: the fused
        # one would integrate twice and spend a vihds_rng_advance launch per step for nothing)
        obs = getattr(batch, "observations", None)
        shape_key = (tuple(obs.shape) if obs is not None else None, int(self.args.train_samples))
        gtail_on = bool(self._gtail_ok) and shape_key not in self._gtail_declined
        ode._train_without_x_predict = gtail_on
        # (... and the sampling stage can run inside the forward launch: params.fused_theta_ode, vihds_theta_ode_fwd)
        self.model._fuse_theta_ode = gtail_on and bool(default_get_value(self.settings.params, "fused_theta_ode", True))
        try:
            batch_results, theta, q, p = self.model(batch, self.args.train_samples)
        finally:
            ode._train_without_x_predict = False
            self.model._fuse_theta_ode = False
        loss = self._general_tail(batch_results, theta, q, p)
        if loss is None and gtail_on:
            self._gtail_declined.add(shape_key)
        if loss is not None:
            # params.fused_step_tail, any model: IWAE loss, ODE adjoint, weight gradients, theta / encoder adjoints and Adam
            # ran as GeneralTail's launches; autograd's backward and optimizer.step() do not run for this step
            if self.replica is not None:
                self._grad_buffer = parallel.allreduce_gradients(self.model.parameters(), self.replica.group, self._grad_buffer)
                self.optimizer.gate = None
                self.optimizer.step()
            if zero_grad:
                self.optimizer.zero_grad(set_to_none=True)
            return loss.detach()
        node = getattr(getattr(getattr(batch_results, "solution", None), "logp_buffer", None), "grad_fn", None)
        if type(node).__name__ == "ThetaOdeFusedBackward" and node.rng_state is not None:
            # (the fused forward left the generator's step to the tail, which did not take this step after all)
            from vihds import hip

            hip.check(hip.lib().vihds_rng_advance(hip.ptr(node.rng_state), hip.current_stream()), "vihds_rng_advance")
        self._in_step = True
        ops._PENDING_IWAE.clear()  # (a deferred loss whose backward never ran must not be mistaken for this step's)
        try:
            elbo = self.cost(batch, batch_results, theta, q, p).elbo
        finally:
            self._in_step = False
        if self._step_tail(batch_results, q) is not None:
            # params.fused_step_tail: loss, backward and Adam ran as vihds_step_tail's two launches
            if self.replica is not None:
                # row replicas: the tail formed this rank's gradients (no Adam); ONE in-place all-reduce over the arena they
                # sit in, then the Adam launch on the sums -- three launches and a collective where the autograd path has five
                self._grad_buffer = parallel.allreduce_gradients(self.model.parameters(), self.replica.group, self._grad_buffer)
                self.optimizer.gate = None
                self.optimizer.step()
            if zero_grad:
                self.optimizer.zero_grad(set_to_none=True)
            return elbo.detach()
        if elbo.is_cuda:
            elbo.backward(ops.unit_gradient(elbo.device))  # no ones_like fill, and no launch for the loss's backward
        else:
            elbo.backward()
        if ops._PENDING_IWAE:
            # a deferred loss (fused_iwae_backward) is only evaluated if the decoder step's backward took the job; if it
            # did not (the log-likelihood had a second consumer, a hook, ...) the gradients just computed are garbage
            for job in list(ops._PENDING_IWAE.values()):
                ops._run_iwae_job(job)
            ops._PENDING_IWAE.clear()
            raise RuntimeError("fused_iwae_backward: the decoder step's backward did not evaluate the deferred IWAE loss "
                               "(its gradient did not arrive as the unit-gradient broadcast); set "
                               "params.fused_iwae_backward: false for this model / training loop")
        sync = self.shard if self.shard is not None else self.replica
        if sync is not None:
            self._grad_buffer = parallel.allreduce_gradients(self.model.parameters(), sync.group, self._grad_buffer)
        if elbo.is_cuda:
            # a non-finite loss makes the update a no-op on the device, step count included (the reference stops before
            # optimizer.step on a NaN ELBO, training.py:331-334; here the host reads the loss after the launches)
            # Row replicas each have their OWN loss: gating on it would let one rank skip an update its peers apply.  There
            # the all-reduced gradients -- identical on every rank, non-finite wherever any rank's were -- decide, element
            # by element, inside the Adam kernel: the replicas stay in step
            self.optimizer.gate = elbo.detach() if self.replica is None else None
        self.optimizer.step()
        if zero_grad:
            self.optimizer.zero_grad(set_to_none=True)
        return elbo.detach()

    def _step_tail(self, batch_results, q):
        """params.fused_step_tail (single process, fused decoder step with a deferred IWAE loss, all trainable parameters
        in the encoder): hand the rest of the step to ops.StepTail.  Returns the loss tensor, or None when it does not
        apply (the caller then runs autograd's backward and optimizer.step())."""
        if not self.fused_tail or self.shard is not None or len(ops._PENDING_IWAE) != 1:
            return None
        sol = getattr(batch_results, "solution", None)
        dec_node = getattr(getattr(sol, "logp_buffer", None), "grad_fn", None)
        packed = getattr(q, "_packed_q", None)
        enc_node = getattr(packed[1], "grad_fn", None) if packed is not None else None
        if (type(dec_node).__name__ != "DecoderStepFusedBackward" or type(enc_node).__name__ != "EncoderQTablesBackward"):
            return None
        if self._tail is None:
            self._tail = ops.StepTail(self.model.encoder, self.optimizer)
            self._tail_ok = self._tail.applicable()
        if not self._tail_ok:
            return None
        q_all, u = dec_node.saved_tensors[0], dec_node.saved_tensors[6]
        key = (q_all.shape, u.shape[1])
        if key not in self._tail_shapes:  # (the library declines shapes past its LDS budget: the five-launch path then)
            from vihds import hip

            self._tail_shapes[key] = bool(hip.lib().vihds_step_tail_supported(enc_node.shape, q_all.shape[0] // 2, u.shape[1]))
        if not self._tail_shapes[key]:
            return None
        (job,) = ops._PENDING_IWAE.values()
        ops._PENDING_IWAE.clear()
        return self._tail.launch(dec_node, enc_node, job, apply_adam=self.replica is None)

    def _general_tail(self, batch_results, theta, q, p):
        """params.fused_step_tail for the models WITHOUT a fused decoder step (anything but dr_constant: relay / degrader /
        prpr / auto and their _precisions forms, dr_blackbox, dr_constant_precisions; reference training.py:324-340 is
        model-agnostic): the forward ran as encoder -> theta kernel -> [conditioning] -> vihds_ode_fwd; hand everything
        behind it to ops.GeneralTail.  Returns the loss tensor, or None when it does not apply (the caller then runs cost(),
        autograd's backward and optimizer.step())."""
        if not self.fused_tail or self.shard is not None or self._gtail_ok is False:
            return None
        sol = getattr(batch_results, "solution", None)
        logp = getattr(sol, "logp_buffer", None)
        ode_node = getattr(logp, "grad_fn", None)
        packed = getattr(theta, "_packed", None)
        theta_node = getattr(packed, "grad_fn", None)
        pq = getattr(q, "_packed_q", None)
        enc_node = getattr(pq[1], "grad_fn", None) if pq is not None else None
        fused_fwd = type(ode_node).__name__ == "ThetaOdeFusedBackward" and type(theta_node).__name__ == "ThetaOdeFusedBackward"
        if (not (fused_fwd or (type(ode_node).__name__ == "OdeSolveObserveBackward"
                               and type(theta_node).__name__ == "ThetaSampleLogProbPackedBackward"))
                or type(enc_node).__name__ != "EncoderQTablesBackward" or not getattr(sol, "has_logp", False)):
            return None
        if self._gtail is None:
            self._gtail = ops.GeneralTail(self.model.encoder, self.optimizer, self.model.decoder.ode_model)
            self._gtail.inkernel_iwae = bool(default_get_value(self.settings.params, "inkernel_iwae", True))
            self._gtail_ok = self._gtail.applicable()
            if not self._gtail_ok:
                return None
        fwd = ops.GeneralTail.forward_state(theta_node, ode_node)
        # the integrator must have read the sampling kernel's own buffer (nothing re-bound into a copy on the way)
        if fwd["theta"].data_ptr() != packed.data_ptr():
            return None
        q_all, u = fwd["q_all"], fwd["u"]
        key = (q_all.shape, u.shape[1])
        if key not in self._tail_shapes:
            from vihds import hip

            self._tail_shapes[key] = bool(hip.lib().vihds_step_tail_supported(enc_node.shape, q_all.shape[0] // 2, u.shape[1]))
        if not self._tail_shapes[key]:
            return None
        return self._gtail.launch(fwd, enc_node, q.log_prob(theta), p.log_prob(theta), logp.shape[2],
                                  apply_adam=self.replica is None)

    def _snapshot_training_state(self):
        """Parameters + optimizer state before a capture's warm-up steps (a new batch shape, e.g. an epoch's last partial
        batch, would otherwise take three extra optimizer steps that an eager run does not take)."""
        params = [p.detach().clone() for p in self.model.parameters()]
        flat = getattr(self.optimizer, "_flat", None)
        opt = None
        if flat:
            opt = {gi: {k: st[k].clone() for k in ("m", "v", "state")} for gi, st in flat.items()}
        rng = {id(t): t.clone() for t in self._rng_states()}
        # the host-side streams too (numpy's global RandomState: u; torch's CPU generator: the conditioner's weights and the
        # loader's shuffles): the warm-up steps of a capture must not consume draws the training run is entitled to
        from vihds import nprand

        nprand._collect_stray()  # (a helper-thread draw started ahead mutates numpy's state in place: at rest first)
        rng["__numpy__"] = np.random.get_state()
        rng["__torch_cpu__"] = torch.get_rng_state()
        return params, opt, rng

    def _rng_states(self):
        """Device-side generator states the kernels advance (u draws, conditioner weights)."""
        out = []
        st = getattr(self.model, "_rng_state", None)
        if st is not None:
            out.append(st)
        ode = self.model.decoder.ode_model
        st = getattr(ode, "_rng_state", None)
        if st is not None:
            out.append(st)
        return out

    def _restore_training_state(self, snap):
        params, opt, rng = snap
        with torch.no_grad():
            for p, q in zip(self.model.parameters(), params):
                p.copy_(q)
            flat = getattr(self.optimizer, "_flat", None)
            if flat is not None:
                for gi, st in flat.items():
                    if opt is not None and gi in opt:
                        for k in ("m", "v", "state"):
                            st[k].copy_(opt[gi][k])
                    else:  # the optimizer state was created by the warm-up: back to its initial value
                        for k in ("m", "v", "state"):
                            st[k].zero_()
            if "__numpy__" in rng:
                from vihds import nprand

                nprand._collect_stray()
                np.random.set_state(rng["__numpy__"])
                torch.set_rng_state(rng["__torch_cpu__"])
            for t in self._rng_states():
                if id(t) in rng:
                    t.copy_(rng[id(t)])
                else:  # created (seeded) by the warm-up: keep the seed, rewind the step / ticket words
                    t[2:].zero_()

    def _capture(self, static, repeat, prologue=None):
        """Warm up on a side stream (rolled back afterwards), then capture `repeat` steps on the buffers `static` --
        behind `prologue()` when given (e.g. the gather that fills them) -- into a hipGraph.  Returns (graph, static, loss)."""
        g, losses = self._capture_segments([(static, prologue if k == 0 else None) for k in range(repeat)])
        return g, static, (losses[0] if repeat == 1 else losses)

    def _graphs_off(self, exc):
        """hip_graph was chosen automatically and a capture failed (a step that synchronises, an allocation the capture does
        not permit, ...): say so once and run eagerly from here on -- the numbers are the same.  An explicit hip_graph: true
        re-raises."""
        # (only failures of the capture itself qualify -- a step that synchronises, an allocation or a call the capture does
        # not permit; anything else, e.g. a shape error or the host-draw reservation mismatch, is a bug and is raised as it is)
        msg = str(exc).lower()
        capture_related = isinstance(exc, RuntimeError) and any(k in msg for k in ("captur", "hiperror", "cuda error", "hip error",
                                                                                     "graph", "stream is capturing"))
        if not self._graph_auto or not capture_related:
            raise exc
        print("- hipGraph capture failed (%s: %s): continuing with eager launches" % (type(exc).__name__, str(exc)[:200]))
        torch.cuda.synchronize()
        self.use_graph = self.epoch_graph = False
        self.optimizer.zero_grad(set_to_none=True)

    def _capture_segments(self, segments):
        """One hipGraph holding len(segments) consecutive training steps: segment k = `prologue_k()` (or nothing), then the
        step on the buffers `static_k`.  Returns (graph, [loss_k])."""
        s = torch.cuda.Stream()
        # the snapshot's clone kernels are enqueued on the current stream BEFORE the side stream is made to wait for
        # it, so the warm-up steps (which run Adam and advance the generator states) are ordered after the copies
        snap = self._snapshot_training_state()
        s.wait_stream(torch.cuda.current_stream())
        draws = hostdraws.HostDraws()
        per_static = {}
        hostdraws.ACTIVE = draws  # (measuring: how many host-drawn numbers a step on each buffer set consumes)
        try:
            with torch.cuda.stream(s):
                seen = set()
                for static, prologue in segments:  # allocator warm-up, lazy initialisations, Adam state: once per batch shape
                    if id(static) in seen and prologue is None:
                        continue
                    seen.add(id(static))
                    for _ in range(3):
                        if prologue is not None:
                            prologue()
                        draws.noted = 0
                        self.step(static)
                        per_static[id(static)] = draws.noted
        finally:
            hostdraws.ACTIVE = None
        torch.cuda.current_stream().wait_stream(s)
        need = sum(per_static.get(id(static), 0) for static, _ in segments)
        if need:
            draws.reserve(need, self.train_data.observations.device)
        self._restore_training_state(snap)  # the warm-up steps must not count as training steps
        self.optimizer.zero_grad(set_to_none=True)
        # The parameters' AccumulateGrad nodes were created during the warm-up steps, on the side stream; the capture
        # runs on the graph's own stream, where every producer and consumer of a gradient is recorded in program
        # order, so the "stream does not match" warning autograd prints once per capture does not apply.
        _warn = getattr(torch.autograd.graph, "set_warn_on_accumulate_grad_stream_mismatch", None)
        if _warn is not None:
            _warn(False)
        hostdraws.ACTIVE = draws
        try:
            sync = self.shard if self.shard is not None else self.replica
            self.collectives_captured = sync is not None and parallel.collectives_capturable(sync.group)
            if sync is not None and not self.collectives_captured:  # cut the captured step at its collectives
                if len(segments) != 1:
                    raise ValueError("several steps per graph need the collectives inside the graph "
                                     "(parallel.collectives_capturable())")
                static, prologue = segments[0]
                g = parallel.SegmentedGraph()

                def fn():
                    if prologue is not None:
                        prologue()
                    return self.step(static, zero_grad=False)

                loss = [g.capture(fn)]
            else:
                # single process -- or several ranks whose communicator records into the capture: the collectives sit in
                # the graph between the kernels they separate, one graph launch per `len(segments)` steps
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, **({"capture_error_mode": "thread_local"} if sync is not None else {})):
                    loss = []
                    for k, (static, prologue) in enumerate(segments):
                        if prologue is not None:
                            prologue()
                        # (every step but the last drops its gradients: left standing, autograd would ADD the next step's)
                        loss.append(self.step(static, zero_grad=k < len(segments) - 1))
        finally:
            hostdraws.ACTIVE = None
            if _warn is not None:
                _warn(True)  # only the capture itself is exempt, not the rest of the process
        g.host_draws = draws
        return g, loss

    def graph_step(self, batch, repeat=1, next_same=False, ahead=1):
        """The same step replayed from a hipGraph: the ~10^2 small launches of encoder + kernels + Adam become one
        graph launch.  Needs device-side RNG (u_rng=device, conditioner_rng=device) and a fixed batch shape.
        Capture follows PyTorch's whole-network recipe: warm up on a side stream, drop the .grad tensors, then
        capture forward + backward + optimizer.step() so the gradients live in the graph's private pool and are
        rewritten (not accumulated) by every replay.
        repeat > 1: that many CONSECUTIVE steps on this batch in one graph (the same launches `repeat` times over, each
        with its own draws and its own Adam step: the generators and the step counter live on the device) -- between two
        graph launches the GPU idles for 6-8 us, which at ~90 us per step is worth amortising when the batch is resident
        anyway.  Returns the loss of the last step; all of them are in self.last_losses."""
        repeat = int(repeat)
        if repeat > 1 and (self.shard is not None or self.replica is not None):
            sync = self.shard if self.shard is not None else self.replica
            if not parallel.collectives_capturable(sync.group):
                raise ValueError("graph_step(repeat > 1) with several ranks needs the collectives inside the graph "
                                 "(parallel.collectives_capturable()); the segmented step is one step per replay")
        key = tuple(batch.observations.shape) + ((repeat,) if repeat > 1 else ())
        if key not in self._graphs:
            static = attrify({k: (v.clone(memory_format=torch.contiguous_format) if isinstance(v, torch.Tensor) else v)
                              for k, v in batch.items()})
            # input-only preprocessing of the encoder (reference encoders.py:385) is done when a batch is staged, not
            # inside every replay
            static["delta_obs"] = _delta_obs(static.observations)
            try:
                self._graphs[key] = self._capture(static, repeat)
            except Exception as exc:  # noqa: BLE001
                self._graphs_off(exc)
                out = None
                for _ in range(repeat):
                    out = self.step(batch)
                return out
        g, static, loss = self._graphs[key]
        if self._staged.get(key) is not batch:  # a batch that is already resident in the graph's inputs is not re-copied
            for k in ("dev_1hot", "inputs", "observations", "times"):
                static[k].copy_(batch[k], non_blocking=True)
            static["delta_obs"].copy_(_delta_obs(static.observations))
            self._staged[key] = batch
        hostdraws.replay(g, g if next_same else None, ahead)
        if repeat > 1:
            self.last_losses = loss
            return loss[-1]
        return loss

    # ------------------------------------------------------------------------------------------------
    def gather_rows(self, rows, out=None):
        """Batch = rows `rows` (int64 tensor on the device) of the resident training set, gathered by ONE launch
        (vihds_gather_batch; delta_obs, the encoder's input-only preprocessing, formed on the way).  `out`: buffers of a
        previous call with the same number of rows, written in place (the captured step's inputs)."""
        from vihds import hip

        src = self.train_data
        B = int(rows.shape[0])
        n_src, C4, T = src.observations.shape
        n_tr, D = src.inputs.shape[1], src.dev_1hot.shape[1]
        dev = src.observations.device
        if out is None:
            out = attrify({"observations": torch.empty((B, C4, T), device=dev), "inputs": torch.empty((B, n_tr), device=dev),
                           "dev_1hot": torch.empty((B, D), device=dev), "delta_obs": torch.empty((B, C4, T - 1), device=dev),
                           "times": src.times, "devices": None})
        rc = hip.lib().vihds_gather_batch(B, n_src, C4, T, n_tr, D, hip.ptr(rows), hip.ptr(src.observations),
                                          hip.ptr(src.inputs), hip.ptr(src.dev_1hot), hip.ptr(out.observations),
                                          hip.ptr(out.inputs), hip.ptr(out.dev_1hot), hip.ptr(out.delta_obs),
                                          hip.current_stream())
        hip.check(rc, "vihds_gather_batch")
        return out

    class _IndexStaging:
        """Pinned staging slots for a captured step's row indices.  The host->device copy of step k is asynchronous and
        sits behind step k-1 on the stream, so the slot it reads must not be rewritten until it has run: a slot is reused
        only after the event recorded behind its copy has completed (one pinned buffer, rewritten per step,
        let step k+1's indices overtake copy k whenever nothing else synchronised in between)."""

        def __init__(self, n, slots=4):
            self.bufs = [torch.empty(n, dtype=torch.int64).pin_memory() for _ in range(slots)]
            self.events = [None] * slots
            self.pos = 0

        def upload(self, dst, fill):
            k = self.pos
            self.pos = (k + 1) % len(self.bufs)
            if self.events[k] is not None:
                self.events[k].synchronize()
            fill(self.bufs[k])
            dst.copy_(self.bufs[k], non_blocking=True)
            if self.events[k] is None:
                self.events[k] = torch.cuda.Event()
            self.events[k].record()

    def step_rows(self, rows_host, next_rows=None, ahead=1):
        """One training step on the rows `rows_host` (host int64 tensor) of the resident training set: eager, or -- with
        params.hip_graph -- one hipGraph per batch size holding the gather and the step; per step the host then only
        refreshes the graph's index buffer (one small copy) and launches it.  next_rows: the rows of the step that is KNOWN to
        follow (Training.run: the epoch's next batch) -- its host-side numpy draw then starts on a helper thread as soon as
        this step is queued (vihds/hostdraws.py); ahead: how many of the steps that follow are known to be of next_rows' size
        (the helper may then draw for two of them in a row)."""
        n = int(rows_host.shape[0])
        dev = self.train_data.observations.device
        if not self.use_graph:
            return self.step(self.gather_rows(rows_host.to(dev, non_blocking=True)))
        key = ("rows", n)
        if key not in self._graphs:
            idx = rows_host.to(dev).clone()
            static = self.gather_rows(idx)
            try:
                self._graphs[key] = (self._capture(static, 1, prologue=lambda: self.gather_rows(idx, out=static))
                                     + (idx, self._IndexStaging(n)))
            except Exception as exc:  # noqa: BLE001
                self._graphs_off(exc)
                return self.step(self.gather_rows(rows_host.to(dev, non_blocking=True)))
        g, static, loss, idx, staging = self._graphs[key]
        staging.upload(idx, lambda buf: buf.copy_(rows_host))
        nxt = self._graphs.get(("rows", int(next_rows.shape[0]))) if next_rows is not None else None
        hostdraws.replay(g, nxt[0] if nxt is not None else None, ahead)
        return loss

    def epoch_rows(self, batches):
        """All the steps of one epoch -- `batches`: the loader's row-index tensors, in order -- as ONE hipGraph launch: the
        epoch's indices travel in one pinned copy, every step is gather + step on the buffers of its batch size.  One graph
        per sequence of batch sizes (an epoch of the reference's loader is always the same sequence: full batches, then
        the ragged one).  Returns the steps' losses (device scalars)."""
        sizes = tuple(int(b.shape[0]) for b in batches)
        dev = self.train_data.observations.device
        key = ("epoch",) + sizes
        if key not in self._graphs:
            idx = torch.cat(list(batches)).to(dev).clone()
            statics, segments, o = {}, [], 0
            for n in sizes:
                view = idx[o: o + n]
                if n not in statics:
                    statics[n] = self.gather_rows(view)
                segments.append((statics[n], (lambda v=view, st=statics[n]: self.gather_rows(v, out=st))))
                o += n
            try:
                self._graphs[key] = self._capture_segments(segments) + (idx, self._IndexStaging(int(idx.shape[0])))
            except Exception as exc:  # noqa: BLE001 -- the same policy as graph_step / step_rows: automatic mode falls back to eager
                self._graphs_off(exc)
                return [self.step(self.gather_rows(b.to(dev, non_blocking=True))) for b in batches]
        g, losses, idx, staging = self._graphs[key]
        staging.upload(idx, lambda buf: torch.cat(list(batches), out=buf))
        hostdraws.replay(g)
        return losses

    def _run_batch(self, epoch_start, batch, log_data, next_batch=None, ahead=1):
        """reference training.py:324-340.  `batch`: a batch of the reference's form, or the row indices of one (host int64
        tensor: what self.train_loader yields).  next_batch: the batch that is known to follow (see step_rows)."""
        log_data.batch_feed_time += time.time() - epoch_start
        train_start = time.time()
        if isinstance(batch, torch.Tensor):
            elbo = self.step_rows(batch, next_batch if isinstance(next_batch, torch.Tensor) else None, ahead)
        else:
            elbo = (self.graph_step(batch, next_same=next_batch is batch, ahead=ahead) if self.use_graph else self.step(batch))
        self._steps += 1
        if self.use_graph and self.nan_check_every == 1 and self.replica is None and self.shard is None:
            # The reference looks at every step's ELBO (training.py:331), a device synchronisation per step.  With the step
            # replayed from a graph the look is one step late: step k is queued -- its host-side draws included -- before the
            # ELBO of step k-1 is read, so the host's work for the next step overlaps the GPU's for this one.  Every update
            # is gated on its own loss on the device, so the parameters are those of the last finite step either way.
            # The look itself waits for nothing queued since: step k's ELBO is copied to a pinned slot behind step k, with an
            # event behind the copy; one step later the host waits for THAT event and reads the slot.
            if self._elbo_look is None:
                self._elbo_look = _ElboLook()
            self._pending_elbo = elbo
            prev = self._elbo_look.push(elbo)
            if prev is not None and not math.isfinite(self._elbo_look.take(prev)):
                self._pending_elbo = None
                self._elbo_look.pending = None
                print("Cannot proceed with ELBO = nan. Exiting.")
                return False
        elif self.nan_check_every > 0 and self._steps % self.nan_check_every == 0 and self._loss_is_nan(elbo):
            # (the reference aborts before backward / optimizer.step, training.py:331-334; here the step that produced the
            # NaN has already been launched -- the Adam launch is gated on the loss on the device (vihds_adam_step's
            # `gate`, vihds_step_tail's row check), so parameters, moments and the step count are those of the last
            # finite step)
            print("Cannot proceed with ELBO = nan. Exiting.")
            return False
        log_data.batch_train_time += time.time() - train_start
        return True

    def _loss_is_nan(self, elbo):
        """The reference's per-step check (training.py:331).  Row replicas must all leave the loop together -- a rank that
        stopped alone would leave its peers blocked in the next gradient all-reduce -- so they agree on the flag first."""
        if self.replica is None:
            # (one device->host copy; torch.isfinite would be a launch of its own before it.  NOT FINITE rather than NaN: where
            # the reference's eager graph ends in NaN, the kernels' max / sum reductions can end in +-inf -- a NaN observation
            # gives -ELBO = +inf from vihds_step_tail -- and the step was a no-op on the device either way)
            return not math.isfinite(float(elbo))
        bad = ~torch.isfinite(elbo)
        if self.replica is not None:
            import torch.distributed as dist

            flag = bad.to(torch.float32).reshape(1)
            dist.all_reduce(flag, op=dist.ReduceOp.MAX, group=self.replica.group)
            bad = flag[0] > 0
        return bool(bad)

    def epoch_index_batches(self):
        """The row-index batches of one epoch, as iterating self.train_loader yields them (same draws from torch's generator,
        same batches): without the loader's per-batch machinery when the check at construction found the short form equal."""
        if self._fast_index_batches:
            return _shuffled_index_batches(len(self.train_loader.dataset), self.n_batch)
        return list(self.train_loader)

    def run(self):
        """reference training.py:342-383"""
        train_writer = valid_writer = None
        if self.settings.trainer is not None:
            try:
                from torch.utils.tensorboard import SummaryWriter

                train_writer, valid_writer = SummaryWriter(self.train_path), SummaryWriter(self.valid_path)
            except ImportError:
                print("- tensorboard is not installed: summaries disabled")
        log_data = TrainingLogData()
        print("---------------------------")
        if self.args.heldout:
            split_name = "heldout device = %s" % self.args.heldout
        else:
            split_name = "split %d of %d" % (self.args.split, self.args.folds)
        print("Training: %s" % split_name)
        iterating = True
        epoch = 1
        valid_output = None
        pending = None  # epoch-graph path: the losses of the epoch launched last, not looked at yet

        def settle():
            """NaN check of the epoch launched last (training.py:331-334 looks at every step's ELBO; here: at the epoch's
            ELBOs, before anything else is launched -- every update is gated on its own loss on the device either way)."""
            nonlocal pending
            ok = True
            if self._pending_elbo is not None:  # (the per-step check that runs one step late: _run_batch)
                self._pending_elbo = None
                last, self._elbo_look.pending = self._elbo_look.pending, None
                if last is not None and not math.isfinite(self._elbo_look.take(last)):
                    print("Cannot proceed with ELBO = nan. Exiting.")
                    ok = False
            if pending is not None and self.nan_check_every > 0:
                if isinstance(pending, int):  # (epoch_lookahead: the epoch's flag, _EpochLook)
                    nan = self._epoch_look.take(pending)
                else:
                    nan = bool((~torch.isfinite(torch.stack([l.detach().reshape(()) for l in pending]))).any())
                if nan:
                    print("Cannot proceed with ELBO = nan. Exiting.")
                    ok = False
            pending = None
            return ok

        try:
            while iterating is True and (epoch < self.args.epochs + 1):
                if not self.model.training:  # (walking the module tree every epoch: 50 us)
                    self.model.train()
                epoch_start = time.time()
                batches = None
                if self.epoch_graph:
                    # the sampler's draws for the epoch (same generator stream as iterating the loader step by step); this host
                    # work runs while the GPU is still in the previous epoch: its losses are looked at only afterwards
                    batches = self.epoch_index_batches()
                    if not all(isinstance(b, torch.Tensor) for b in batches):
                        batches = None
                whole = batches is not None and (self.nan_check_every == 0 or self.nan_check_every >= len(batches))
                if whole and self.epoch_lookahead and isinstance(pending, int):
                    # params.epoch_lookahead: this epoch's graph is queued BEHIND the one whose losses have not been looked at
                    # yet, and only then is that look taken -- the index upload and the graph launch (0.1 ms, a fifth of an
                    # epoch of seven steps) no longer find the GPU idle.  Every update is gated on its own loss on the device
                    # either way; what changes is how late the loop notices a NaN: one more epoch is queued by then.
                    log_data.batch_feed_time += time.time() - epoch_start
                    train_start = time.time()
                    queued = self._epoch_look.post(self.epoch_rows(batches))
                    self._steps += len(batches)
                    iterating = settle()
                    pending = queued
                    log_data.batch_train_time += time.time() - train_start
                    if not iterating:
                        break
                    whole = False  # (done)
                    batches = ()
                else:
                    iterating = settle()
                    if not iterating:
                        break
                if whole:
                    # the whole epoch in one graph launch
                    log_data.batch_feed_time += time.time() - epoch_start
                    train_start = time.time()
                    pending = self.epoch_rows(batches)
                    if self.epoch_lookahead and self.nan_check_every > 0 and self.use_graph:
                        pending = self._epoch_look.post(pending)
                    self._steps += len(batches)
                    log_data.batch_train_time += time.time() - train_start
                elif batches == ():
                    pass
                else:
                    todo = batches if batches is not None else self.epoch_index_batches()
                    def n_rows(b):
                        return int(b.shape[0]) if isinstance(b, torch.Tensor) else len(b.observations)

                    for j, batch in enumerate(todo):
                        if iterating:  # (the batches that follow inside the epoch are known: their numpy draws run ahead)
                            nxt = todo[j + 1] if j + 1 < len(todo) else None
                            two = nxt is not None and j + 2 < len(todo) and n_rows(todo[j + 2]) == n_rows(nxt)
                            iterating = self._run_batch(epoch_start, batch, log_data, next_batch=nxt, ahead=2 if two else 1)
                log_data.total_train_time += time.time() - epoch_start
                if iterating and (np.mod(epoch, self.args.test_epoch) == 0):
                    iterating = settle()
                    if iterating:
                        self.model.eval()
                        valid_output = self._evaluate_elbo_and_plot(epoch, log_data, train_writer, valid_writer)
                self.scheduler.step()
                epoch += 1
            settle()
        finally:
            if self._best_output is not None:
                self._best_output.dump()
                self._best_output = None
        for w in (train_writer, valid_writer):
            if w is not None:
                w.close()
        if self.empty_cache or valid_output is None:
            print("Exiting with no results in cache")
            return None
        valid_output.load()
        valid_output.elbo_list = log_data.validation_elbo_list
        return valid_output
