"""Ad-hoc kernel timing probe (not a test): HIP-event timing of the ODE fwd/bwd kernels at BASELINE shapes."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "vi-hds_amd"), os.path.join(ROOT, "tests")]
import torch
from vihds import ops, hip
from test_hip_parity import _full_problem

def timeit(fn, n=20, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3  # us

from test_hip_parity import _blackbox_problem
B, S, T = 36, 200, 86
spec, theta, wts, cond, dev, times, obs = _blackbox_problem(B, S, T)
th = theta.clone().requires_grad_(True); w = wts.clone().requires_grad_(True); out = {}
def bfwd(): out["o"] = ops.OdeSolveObserve.apply(spec, th, cond, times, obs, dev, w)
bfwd(); gl = torch.ones_like(out["o"][2])
def bbwd():
    th.grad = None; w.grad = None
    out["o"][2].backward(gl, retain_graph=True)
print("dr_blackbox midpoint B=36 S=200: fwd %.1f us, bwd (adjoint + dump + weight-grad GEMMs) %.1f us; 4.15 GFLOP fwd declared -> %.2f TFLOP/s" % (timeit(bfwd), timeit(bbwd), 4.15e9 / timeit(bfwd) / 1e6))
from vihds import hip as _hip
from test_hip_parity import _synthetic_theta
Br, Sr, Tr = 36, 200, 99
slots = _hip.model_slots("relay_constant_precisions")
thr = _synthetic_theta(slots, Br, Sr, 3)
for nme in slots:
    if nme.startswith("init_prec"):
        thr[nme] = torch.exp(3.0 + 0.3 * torch.randn(Br, Sr))
thetar = torch.stack([thr[nme] for nme in slots]).cuda().requires_grad_(True)
condr = torch.log1p(torch.rand(Br, 2) * 1000.0).cuda(); timesr = (torch.arange(Tr, dtype=torch.float32) * 0.17).cuda(); obsr = torch.rand(Br, 4, Tr).cuda()
specr = ops.OdeProblemSpec("relay_constant_precisions", "midpoint", {nme: k for k, nme in enumerate(slots)}, len(slots), C=2)
wr = (torch.randn(2 * (4 * 13 + 4)) * 0.2).cuda().requires_grad_(True); outr = {}
def rfwd(): outr["o"] = ops.OdeSolveObserve.apply(specr, thetar, condr, timesr, obsr, None, wr)
rfwd(); glr = torch.ones_like(outr["o"][2])
def rbwd():
    thetar.grad = None; wr.grad = None
    outr["o"][2].backward(glr, retain_graph=True)
fb = 4 * (45 * Br * Sr + Br * Sr * 16 * Tr + Br * Sr * 4 * Tr); bb_ = 4 * (Br * Sr * 16 * Tr + 45 * Br * Sr)
tf_, tb_ = timeit(rfwd), timeit(rbwd)
print("relay_constant_precisions midpoint B=36 S=200 T=99 N=16: fwd %.1f us (%.0f GB/s), bwd %.1f us (%.0f GB/s)" % (tf_, fb / tf_ / 1e3, tb_, bb_ / tb_ / 1e3))
for model in ["dr_constant"]:
  for (B, S) in [(36, 200), (36, 1000), (234, 1000)]:
    for solver in ["modeuler", "midpoint", "rk4"]:
        T = 86
        slots, theta, cond, times, obs = _full_problem(B, S, T)
        row_of = {n: i for i, n in enumerate(slots)}
        spec = ops.OdeProblemSpec(model, solver, row_of, len(slots), C=2)
        th = theta.clone().requires_grad_(True)
        out = {}
        def fwd():
            out["o"] = ops.OdeSolveObserve.apply(spec, th, cond, times, obs, None, None)
        fwd()
        g = torch.ones_like(out["o"][2])
        def bwd():
            th.grad = None
            out["o"][2].backward(g, retain_graph=True)
        tf = timeit(fwd); tb = timeit(bwd)
        N = 8; P = 35
        fbytes = 4 * (P * B * S + B * S * N * T + B * S * 4 * T); bbytes = 4 * (B * S * N * T + P * B * S)
        print("%s B=%d S=%d %-9s fwd %8.1f us (%.0f GB/s)  bwd %8.1f us (%.0f GB/s)" % (model, B, S, solver, tf, fbytes / tf / 1e3, tb, bbytes / tb / 1e3))
