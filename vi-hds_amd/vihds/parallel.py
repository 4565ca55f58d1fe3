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


def combine_row_lse(row_max, row_sumexp, group):
    """Global row-wise logsumexp from per-rank (max, sum exp(. - max)) pairs: all-reduce(MAX) of the maxima, rescale
    the local sums to the global maximum, all-reduce(SUM).  2 x B floats on the wire (144 B at B=36)."""
    gmax = row_max.clone()
    dist.all_reduce(gmax, op=dist.ReduceOp.MAX, group=group)
    se = row_sumexp * torch.exp(row_max - gmax)
    dist.all_reduce(se, op=dist.ReduceOp.SUM, group=group)
    return gmax + torch.log(se)


def init_from_env(backend=None):
    """One process per GPU, launched by torch.distributed.run (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world == 1:
        return None
    if not dist.is_initialized():
        if backend is None:  # VIHDS_DIST_BACKEND=gloo lets the plumbing be exercised with several ranks on one GPU
            backend = os.environ.get("VIHDS_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
        dist.init_process_group(backend=backend)
    return SampleShard(dist.get_rank(), dist.get_world_size())


def allreduce_gradients(parameters, group=None, buffer=None):
    """Sum the gradients of all parameters over ranks with ONE all-reduce of a flat buffer."""
    params = [p for p in parameters if p.grad is not None]
    if not params:
        return buffer
    n = sum(p.grad.numel() for p in params)
    if buffer is None or buffer.numel() != n or buffer.device != params[0].grad.device:
        buffer = torch.empty(n, device=params[0].grad.device, dtype=params[0].grad.dtype)
    torch.cat([p.grad.reshape(-1) for p in params], out=buffer)
    dist.all_reduce(buffer, op=dist.ReduceOp.SUM, group=group)
    off = 0
    for p in params:
        k = p.grad.numel()
        p.grad.copy_(buffer[off: off + k].view_as(p.grad))
        off += k
    return buffer
