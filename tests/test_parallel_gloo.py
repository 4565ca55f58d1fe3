"""world_size-2 gloo tests (CPU) of the multi-GPU plumbing: S-axis sharding, the two-all-reduce row logsumexp
combine, and the single flat-buffer gradient all-reduce (vihds/parallel.py).  The kernels themselves need a GPU;
what is checked here is that the cross-rank algebra reproduces the single-process result exactly."""
import math
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, tmp):
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path[:0] = [root, os.path.join(root, "vi-hds_amd")]
    from vihds import parallel

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    shard = parallel.init_from_env(backend="gloo")
    assert shard.rank == rank and shard.world == world
    B, S, P = 5, 12, 3
    g = torch.Generator().manual_seed(0)  # same seed on every rank => identical "full" tensors
    u = torch.randn(B, S, P, generator=g)
    log_w = torch.randn(B, S, generator=g) * 30.0
    mine = shard.take(u)
    lo, hi = shard.bounds(S)
    assert torch.equal(mine, u[:, lo:hi]) and mine.is_contiguous()
    # row logsumexp from local (max, sum-exp) pairs == global logsumexp
    lw = log_w[:, lo:hi]
    row_max = lw.max(1).values
    row_se = torch.exp(lw - row_max[:, None]).sum(1)
    lse = parallel.combine_row_lse(row_max, row_se, None)
    ref = torch.logsumexp(log_w, 1)
    assert torch.allclose(lse, ref, rtol=1e-6, atol=1e-6)
    # local softmax weights with the global lse sum to 1 across ranks => gradient needs no extra exchange
    wsum = torch.exp(lw - lse[:, None]).sum(1)
    dist.all_reduce(wsum)
    assert torch.allclose(wsum, torch.ones(B), atol=1e-5)
    # loss = -mean_b(lse - log S_total): d loss / d param via local pieces + one flat all-reduce == single process
    w = torch.nn.Parameter(torch.tensor([0.3, -0.2, 0.7]))
    b = torch.nn.Parameter(torch.tensor(0.1))
    unused = torch.nn.Parameter(torch.zeros(2))

    def log_w_of(uu):
        return (uu * w).sum(-1) * 3.0 + b

    full = log_w_of(u)
    loss_full = -(torch.logsumexp(full, 1) - math.log(S)).mean()
    gw, gb = torch.autograd.grad(loss_full, [w, b])
    local = log_w_of(mine)
    m = local.detach().max(1).values
    lse_g = parallel.combine_row_lse(m, torch.exp(local.detach() - m[:, None]).sum(1), None)
    # surrogate whose gradient is the local part of the global gradient: sum_s softmax_s(global) * log_w_s
    sw = torch.exp(local.detach() - lse_g[:, None])
    (-(sw * local).sum(1).mean()).backward()
    buf = parallel.allreduce_gradients([w, b, unused])
    assert buf.numel() == 4
    assert torch.allclose(w.grad, gw, rtol=1e-5, atol=1e-6) and torch.allclose(b.grad, gb, rtol=1e-5, atol=1e-6)
    try:
        shard.bounds(13)
        raise AssertionError("uneven shard accepted")
    except ValueError:
        pass
    open(os.path.join(tmp, "ok%d" % rank), "w").write("ok")
    dist.destroy_process_group()


def test_two_rank_gloo(tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert (tmp_path / "ok0").exists() and (tmp_path / "ok1").exists()


def test_bench_gpus_n_launches_its_own_ranks():
    """`python bench.py --gpus 2` as typed (no torch.distributed.run around it, WORLD_SIZE unset) must start the two ranks
    itself.  Without a GPU the ranks then stop at bench.py's own "needs an MI355X" check -- which proves they were
    started: the old behaviour was an argument error before anything ran."""
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["VIHDS_DIST_BACKEND"] = "gloo"
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                        "--roofline-steps", "0", "--no-cpu-baseline"], env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, text=True, timeout=600)
    assert "but WORLD_SIZE=1" not in r.stdout, r.stdout[-2000:]
    if not torch.cuda.is_available():
        assert r.returncode != 0 and "needs an MI355X" in r.stdout, r.stdout[-2000:]
