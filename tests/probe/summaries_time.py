"""probe: duration of the evaluation summaries launch at config 3's evaluation shape (B=234, S=1000, N=8, T=86), for the
one-block-per-time-point kernel and the pipelined kernel at several time points per block.
usage: python tests/probe/summaries_time.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "vi-hds_amd"))
from vihds import hip, ops  # noqa: E402

dev = torch.device("cuda:0")
B, S, T, N = 234, 1000, 86, 8
traj = torch.rand(T, N, B, S, device=dev) + 0.5
xpred = torch.rand(T, 4, B, S, device=dev)
log_w = torch.randn(B, S, device=dev)
lse = torch.logsumexp(log_w, 1)
theta = torch.rand(35, B, S, device=dev) + 0.5
L = hip.lib()
nbytes = 4 * (T * N * B * S + 5 * B * S)
for stored in (False, True):
    for tpb in (-1, 1, 2, 3, 4, 6, 8, 0):
        L.vihds_iw_summaries_plan(tpb)
        f = lambda: ops.iw_summaries(log_w, lse, traj, xpred if stored else None, N, theta=theta, prec_rows=[31, 32, 33, 34])
        for _ in range(3):
            f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            f()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 20
        nb = nbytes + (4 * T * 4 * B * S if stored else 0)
        print("stored x_predict %-5s  time points per block %2d: %7.1f us  %.2f TB/s" % (stored, tpb, us, nb / us / 1e6))
L.vihds_iw_summaries_plan(0)
