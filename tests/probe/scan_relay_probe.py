import sys, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/vi-hds_amd"); sys.path.insert(0, "/root/repo/tests")
from fixture_util import Fixture, rel_err
import test_config5_parity as T
from test_hip_parity import _relay_problem
from vihds import ops
name = sys.argv[1] if len(sys.argv) > 1 else "relay_constant_precisions_tiny_modeuler"
fx = Fixture(name)
prec_w, _, _ = fx.decoder_weights()
res = {}
for v in (1, 5):
    res[v] = T._hip_run(fx.model, fx.names, fx.t("theta"), fx.t("inputs"), fx.t("times"), fx.t("observations"), fx.solver, T._flat(prec_w), fx.t("log_p"), fx.t("log_q"), v)
a, b = res[1], res[5]
nc = T.N_CORE[fx.model.split("_")[0]]
for j in range(a["traj"].shape[2]):
    print("state", j, rel_err(b["traj"][:, :, j], a["traj"][:, :, j], dim=None))
print("xp", rel_err(b["xp"], a["xp"]), "lpo", rel_err(b["lpo"], a["lpo"], dim=2), "loss", float(a["loss"]), float(b["loss"]))
for r, n in enumerate(fx.names):
    sc = float(a["th_grad"][r].abs().max())
    if sc > 0: print("%-12s %.2e  (scale %.2e)" % (n, float((a["th_grad"][r] - b["th_grad"][r]).abs().max()) / sc, sc))
print("w", rel_err(b["w_grad"], a["w_grad"]))

# ---- timing at BASELINE config 5's shape (B=36, S=200, T=99, midpoint): forward + adjoint launches, variants 0 and 5 ----
import time
B, S, TT = 36, 200, 99
slots, theta, cond, times, obs, wts = _relay_problem("relay_constant_precisions", B, S, TT, 3, dt=0.17)
row_of = {n: i for i, n in enumerate(slots)}
for v in (0, 5):
    spec = ops.OdeProblemSpec("relay_constant_precisions", "midpoint", row_of, len(slots), C=2, kernel_variant=v)
    th = theta.clone().requires_grad_(True)
    w = wts.clone().requires_grad_(True)
    g = torch.full((4, B, S), -1.0 / (B * S), device="cuda")
    def run():
        traj, xpred, logp = ops.OdeSolveObserve.apply(spec, th, cond, times, obs, None, w)
        logp.backward(g)
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    rec = ops.KernelTimer() if hasattr(ops, "KernelTimer") else None
    ops.TIMER = rec
    for _ in range(10):
        run()
    ops.TIMER = None
    print("variant", v, {k: round(x["mean_us"], 1) for k, x in rec.summary().items()})
