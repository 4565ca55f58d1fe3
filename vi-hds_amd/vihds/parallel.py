"""Multi-GPU: shard the IWAE-sample axis S across ranks.

Every (row, sample) trajectory, its log-likelihood and its log p - log q term are independent; the only coupling
is the row-wise logsumexp over S, the mean over B and the parameter gradients.  So each rank integrates S/world
samples of every row and the step needs exactly
  * two [B]-float all-reduces in the forward (row max, then rescaled row sum-exp) -- vihds.ops.iwae_lse;
  * one flat-buffer all-reduce(SUM) of the parameter gradients in the backward.
Payloads are tens of bytes to ~150 KB: latency-bound, one RCCL call each.  Backend "nccl" is RCCL on ROCm;
"gloo" is used by the CPU tests of this plumbing."""
import os

import torch
import torch.distributed as dist


class SampleShard(object):
    """This rank's contiguous slice of the S axis."""

    def __init__(self, rank, world, group=None):
        self.rank, self.world, self.group = rank, world, group

    def bounds(self, S):
        if S % self.world != 0:
            raise ValueError("n_iwae=%d is not divisible by the %d ranks it is sharded over" % (S, self.world))
        n = S // self.world
        return self.rank * n, (self.rank + 1) * n

    def take(self, u):
        """u [B,S,P] (identical on every rank: same host seed) -> this rank's [B,S/world,P] slice."""
        lo, hi = self.bounds(u.shape[1])
        if hasattr(u, "s_offset"):  # ops.KernelNormal: the kernel draws this rank's slice of the same global stream
            return u.take(lo, hi)
        return u[:, lo:hi].contiguous()


# ---- captured steps with collectives ---------------------------------------------------------------------------
# A training step sharded over ranks has two exchange points (row statistics before the backward, gradients before
# Adam).  Whether a communicator can be captured inside a hipGraph depends on the backend and its version, and a
# failed capture is not recoverable.  So the captured step is cut AT the collectives instead: a SegmentedGraph is a
# list of hipGraphs sharing one memory pool with one eager collective between consecutive graphs.  Host cost per
# step: one launch per segment (3) + the collectives themselves (2), instead of ~30 kernel launches.
_ACTIVE_CAPTURE = None


class SegmentedGraph(object):
    def __init__(self):
        self.graphs, self.between, self.pool, self.result = [], [], None, None

    def _begin(self):
        g = torch.cuda.CUDAGraph()
        if self.pool is None:
            self.pool = torch.cuda.graph_pool_handle()
        # thread_local: the communicator's watchdog thread may touch the device while this thread captures
        g.capture_begin(pool=self.pool, capture_error_mode="thread_local")
        self.graphs.append(g)

    def _end(self):
        self.graphs[-1].capture_end()

    def capture(self, fn):
        """Run fn() once under capture on a side stream; fn reaches its collectives through graph_break()."""
        global _ACTIVE_CAPTURE
        torch.cuda.synchronize()
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            _ACTIVE_CAPTURE = self
            try:
                self._begin()
                self.result = fn()
                self._end()
            finally:
                _ACTIVE_CAPTURE = None
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        return self.result

    def replay(self):
        for k, g in enumerate(self.graphs):
            g.replay()
            if k < len(self.between):
                self.between[k]()
        return self.result


# ---- collectives INSIDE the captured step ------------------------------------------------------------------------
# RCCL (torch's "nccl" backend on ROCm) can record its kernels into a stream capture; whether THIS software stack does -- at
# THIS world size -- is found out once, in throw-away child processes (a capture that fails cannot be undone in the process it
# failed in, and one that hangs would take the job with it): every rank starts a child on its own GPU, the children form a
# communicator of the job's size among themselves, capture one all_reduce and one all_gather_into_tensor into a hipGraph,
# replay it twice and check the numbers; a child that fails, or is not done within the time limit, means "do not capture"
# for every rank.  VIHDS_CAPTURE_COLLECTIVES=0 / 1 overrides the probe; gloo (the CPU-side test backend) is never captured.
_CAPTURABLE = None
_PROBE = r"""
import os, sys, torch, torch.distributed as dist
dev, port, rank, world = int(sys.argv[1]), sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
torch.cuda.set_device(dev)
dist.init_process_group("nccl", init_method="tcp://127.0.0.1:" + port, rank=rank, world_size=world)
x = torch.ones(4096, device="cuda"); y = torch.empty(8 * world, device="cuda"); z = torch.arange(8, device="cuda", dtype=torch.float32)
dist.all_reduce(x); dist.all_gather_into_tensor(y, z)   # communicator set-up outside the capture
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph(); s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    g.capture_begin(capture_error_mode="thread_local")
    x.mul_(2.0); dist.all_reduce(x); dist.all_gather_into_tensor(y, z); x.add_(1.0)
    g.capture_end()
torch.cuda.current_stream().wait_stream(s)
x.fill_(1.0); y.zero_(); g.replay(); g.replay(); torch.cuda.synchronize()
want = 1.0
for _ in range(2):
    want = want * 2.0 * world + 1.0
ok = bool((x == want).all()) and bool((y.view(world, 8) == z).all())
dist.destroy_process_group()
print("VIHDS_CAPTURE_PROBE", "ok" if ok else "wrong")
"""


def collectives_capturable(group=None):
    """True when the step's collectives may be recorded inside ONE hipGraph with the kernels around them (then a
    multi-rank step is one graph launch, several steps per launch included); False: the step is cut at its collectives
    (SegmentedGraph).  All ranks take part in the probe and agree on the answer, so all of them capture the same way."""
    global _CAPTURABLE
    if _CAPTURABLE is not None:
        return _CAPTURABLE
    env = os.environ.get("VIHDS_CAPTURE_COLLECTIVES", "").strip()
    ans = None
    if env in ("0", "1"):
        ans = env == "1"
    elif not dist.is_initialized() or dist.get_backend(group) != "nccl" or not torch.cuda.is_available():
        ans = False
    if ans is None:
        import socket

        rank, world = dist.get_rank(group), dist.get_world_size(group)
        src = dist.get_global_rank(group, 0) if group is not None else 0
        port = torch.zeros(1, device="cuda", dtype=torch.int64)
        if rank == 0:
            with socket.socket() as sock:
                sock.bind(("127.0.0.1", 0))
                port[0] = sock.getsockname()[1]
        dist.broadcast(port, src=src, group=group)
        flag = torch.tensor([1.0 if _probe_capture(rank, world, int(port.item())) else 0.0], device="cuda")
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
        ans = bool(flag.item() > 0)
    _CAPTURABLE = ans
    return ans


def _probe_capture(rank=0, world=1, port=None, timeout=90):
    import socket
    import subprocess
    import sys

    if port is None:
        with socket.socket() as sock:
            sock.bind(("127.0.0.1", 0))
            port = sock.getsockname()[1]
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "GROUP_RANK", "LOCAL_WORLD_SIZE",
              "TORCHELASTIC_RUN_ID"):
        env.pop(k, None)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    try:
        out = subprocess.run([sys.executable, "-c", _PROBE, str(torch.cuda.current_device()), str(port), str(rank), str(world)],
                             env=env, capture_output=True, text=True, timeout=timeout)
    except Exception:  # noqa: BLE001 (a probe that hangs or cannot start means: do not capture)
        return False
    return out.returncode == 0 and "VIHDS_CAPTURE_PROBE ok" in out.stdout


def graph_break(op):
    """Run `op` -- a collective over tensors whose addresses do not change between steps -- now; if a SegmentedGraph
    is being captured, close the current segment before it and open the next one after it.  (Inside a plain hipGraph
    capture -- collectives_capturable() -- the call is simply recorded with the rest.)"""
    seg = _ACTIVE_CAPTURE
    if seg is None:
        op()
        return
    seg._end()
    op()  # (values are garbage during capture -- nothing has executed -- but every rank makes the same call)
    seg.between.append(op)
    seg._begin()


class RowReplica(object):
    """Data parallelism over data rows: every rank runs the whole step on its own batch of rows (its own draws, the
    reference's per-batch semantics unchanged) and the parameter gradients are averaged -- ONE all-reduce per step
    (the SUM is scaled by 1/world inside the Adam kernel).  No exchange in the forward pass."""

    def __init__(self, rank, world, group=None):
        self.rank, self.world, self.group = rank, world, group


def combine_row_lse(row_max, row_sumexp, group):
    """Global row-wise logsumexp from per-rank (max, sum exp(. - max)) pairs.  ONE all-gather of the [2,B] pairs
    (288 B per rank at B=36), then every rank combines the N pairs locally: lse = M + log sum_r se_r exp(m_r - M)."""
    world = dist.get_world_size(group)
    pair = torch.stack([row_max, row_sumexp])  # [2,B]
    gathered = torch.empty((world * 2, pair.shape[1]), device=pair.device, dtype=pair.dtype)
    graph_break(lambda: dist.all_gather_into_tensor(gathered, pair, group=group))
    g3 = gathered.view(world, 2, -1)
    m, se = g3[:, 0], g3[:, 1]  # [N,B]
    gmax = m.max(0).values
    return gmax + torch.log((se * torch.exp(m - gmax)).sum(0))


def init_from_env(backend=None):
    """One process per GPU, launched by torch.distributed.run (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world == 1 and os.environ.get("VIHDS_FORCE_DIST") != "1":
        return None  # (VIHDS_FORCE_DIST=1: a one-rank job through the distributed path -- communicator, collectives and all)
    if not dist.is_initialized():
        if backend is None:  # VIHDS_DIST_BACKEND=gloo lets the plumbing be exercised with several ranks on one GPU
            backend = os.environ.get("VIHDS_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
        dist.init_process_group(backend=backend)
    return SampleShard(dist.get_rank(), dist.get_world_size())


STATS = {"grads_in_place": 0, "grads_packed": 0}  # which path allreduce_gradients took (tests / probes)


def allreduce_gradients(parameters, group=None, buffer=None):
    """Sum the gradients of all parameters over ranks with ONE all-reduce.  If the gradients already sit back to back
    in one buffer (the fused encoder's backward carves them out of one arena) that buffer is reduced in place -- no
    flatten, no scatter; otherwise they are packed into `buffer`, reduced, and the .grad fields re-pointed at views
    of it (no copy back)."""
    params = [p for p in parameters if p.grad is not None]
    if not params:
        return buffer
    grads = [p.grad for p in params]
    n = sum(g.numel() for g in grads)
    g0 = grads[0]
    contiguous_run = all(g.is_contiguous() and g.dtype == g0.dtype for g in grads)
    if contiguous_run:
        ptr = g0.data_ptr()
        for g in grads:
            if g.data_ptr() != ptr:
                contiguous_run = False
                break
            ptr += g.numel() * g.element_size()
    if contiguous_run and g0.untyped_storage().nbytes() - (g0.data_ptr() - g0.untyped_storage().data_ptr()) >= n * g0.element_size():
        flat = torch.empty(0, device=g0.device, dtype=g0.dtype).set_(g0.untyped_storage(), g0.storage_offset(), (n,))
        graph_break(lambda: dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group))
        STATS["grads_in_place"] += 1
        return buffer
    if buffer is None or buffer.numel() != n or buffer.device != g0.device:
        buffer = torch.empty(n, device=g0.device, dtype=g0.dtype)
    STATS["grads_packed"] += 1
    torch.cat([g.reshape(-1) for g in grads], out=buffer)
    graph_break(lambda: dist.all_reduce(buffer, op=dist.ReduceOp.SUM, group=group))
    off = 0
    for p in params:
        k = p.grad.numel()
        p.grad = buffer[off: off + k].view_as(p.grad)
        off += k
    return buffer


def combine_iw_summaries(summ, group):
    """Results.init's importance-weighted summaries (reference utils.py:79-99) when the S axis is sharded: every rank
    holds sums over ITS samples weighted with the globally normalised weights, so the mean, the states and the
    variance add up across ranks; the standard deviation goes through the second moment (sd^2 + mu^2 is the local
    weighted sum of x^2 + 1/prec).  One all-reduce over the four tensors back to back."""
    mu, sd, st, var = summ
    m2 = sd * sd + mu * mu
    parts = [mu, m2, st, var]
    flat = torch.cat([t.reshape(-1) for t in parts])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    out, o = [], 0
    for t in parts:
        out.append(flat[o:o + t.numel()].view(t.shape))
        o += t.numel()
    mu, m2, st, var = out
    return mu, (m2 - mu * mu).clamp_min(0).sqrt(), st, var
