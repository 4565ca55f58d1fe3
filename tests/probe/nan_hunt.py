"""Ad-hoc probe (not a test): find the first step at which anything in the training step goes non-finite and report
where (encoder table, theta, log-probs, trajectories, gradients)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "vi-hds_amd")]
import torch
from vihds import synthetic

use_kernel = (sys.argv[1] == "kernel") if len(sys.argv) > 1 else True
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
n_steps = int(sys.argv[3]) if len(sys.argv) > 3 else 400
quiet = len(sys.argv) > 4
lr = float(sys.argv[5]) if len(sys.argv) > 5 else 0.01
args, settings, data, parameters, model, training = synthetic.build(
    "dr_constant_icml", 36, 200, solver="rk4", device="cuda:0", seed=seed, shard=None, u_rng="kernel",
    conditioner_rng="kernel", hip_graph=False, nan_check_every=0, learning_rate=lr)
model.encoder.use_kernel = use_kernel
model.train()
batch = training.train_data
one = torch.ones((), device="cuda:0")
for it in range(n_steps):
    batch_results, theta, q, p = model(batch, args.train_samples)
    q_all = q._packed_q[1]
    elbo = training.cost(batch, batch_results, theta, q, p).elbo
    elbo.backward(one.expand_as(elbo))
    bad = {}
    bad["q_all"] = not torch.isfinite(q_all).all()
    bad["theta"] = not torch.isfinite(theta._packed).all()
    bad["elbo"] = not torch.isfinite(elbo)
    grads = {n: p_.grad for n, p_ in model.named_parameters() if p_.grad is not None}
    bad_g = [n for n, g in grads.items() if not torch.isfinite(g).all()]
    if any(bad.values()) or bad_g or (it % 50 == 0 and not quiet):
        lp = q_all[model.encoder.q_rows.long()[len(model.encoder.names):]]
        print(it, "elbo %.3f" % float(elbo), bad, "bad grads:", bad_g, "max|q_all| %.2f max log_prec %.2f min %.2f" % (float(q_all.abs().max()), float(lp.max()), float(lp.min())),
              "max|grad| %.3g" % max(float(g.abs().max()) for g in grads.values()), flush=True)
    if any(bad.values()) or bad_g:
        print("BLOWUP mode=%s seed=%d lr=%g step=%d" % (sys.argv[1], seed, lr, it), flush=True)
        break
    training.optimizer.step()
    training.optimizer.zero_grad(set_to_none=True)
else:
    print("SURVIVED mode=%s seed=%d lr=%g steps=%d elbo %.2f" % (sys.argv[1], seed, lr, n_steps, float(elbo)), flush=True)
# which parameters ran away?
enc = model.encoder
P = len(enc.names)
rows = enc.q_rows.long()
mu, lp = q_all[rows[:P]].detach(), q_all[rows[P:]].detach()
_, pm, pp = enc.p.image("cuda:0", 1)
for i, n in enumerate(enc.names):
    sig = float(1.0 / pp[i, 0].sqrt())
    z = (mu[i] - pm[i, 0]) / sig
    if float(lp[i].abs().max()) > 5.5 or float(z.abs().max()) > 3.5:
        print("  %-10s kind %d  q_mu-p_mu in prior sigmas: min %.2f max %.2f   log_prec min %.2f max %.2f" %
              (n, int(enc.kind[i]), float(z.min()), float(z.max()), float(lp[i].min()), float(lp[i].max())))
