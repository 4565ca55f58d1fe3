"""Ad-hoc probe: cumulative time of the phases of the bench's own decoder launch (dr_scan_train_theta_kernel):
the step's launch closure is recorded as bench.py does, then re-issued with kernel_variant = 3 | phase << 8 (the kernel
returns after that phase).  The generators' ticket words are zeroed after every phase run (an early return never
advances them)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "vi-hds_amd"), os.path.join(ROOT, "tests")]
import torch
from vihds import ops, synthetic

solver = sys.argv[1] if len(sys.argv) > 1 else "rk4"
S = int(sys.argv[2]) if len(sys.argv) > 2 else 200
args, settings, data, parameters, model, training = synthetic.build(
    "dr_constant_icml", 36, S, solver=solver, device="cuda:0", seed=1, u_rng="kernel", conditioner_rng="kernel",
    hip_graph=False, nan_check_every=0, learning_rate=0.001, fused_ode_training=True)
model.train()
batch = training.train_data
training.step(batch)
rec = ops.LaunchRecorder()
ops.TIMER = rec
training.step(batch)
ops.TIMER = None
fn = rec.calls["decoder_step"]
cells = {n: c.cell_contents for n, c in zip(fn.__code__.co_freevars, fn.__closure__)}
prob = cells["prob"]
states = [model._rng_state, model.decoder.ode_model._rng_state]
names = {1: "sampling stage, sigmoid table", 2: "x chains (wave 0) / hill (waves 1-3), gamma", 3: "parameters", 4: "level-1 maps + scan",
         5: "level-1 steps, promoters, level-2 maps + scan", 6: "log-likelihood", 7: "adjoint level 2 + promoters",
         8: "adjoint level 1", 9: "adjoint x", 0: "epilogue (full kernel)"}
base = prob.kernel_variant
prev = 0.0
print("== decoder launch, %s, B=36 S=%d" % (solver, S))
for ph in list(range(1, 10)) + [0]:
    prob.kernel_variant = 3 | (ph << 8)
    def run(nrep):
        for _ in range(nrep):
            fn()
            if ph:
                for s in states:
                    s[3:4].zero_()
    run(3)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    # (the zero-fills between launches are ~2 us kernels on the same stream: time the launches alone through a
    # second measurement of the fills)
    e0.record()
    run(50)
    e1.record(); torch.cuda.synchronize()
    tot = e0.elapsed_time(e1) / 50 * 1e3
    fill = 0.0
    if ph:
        e0.record()
        for _ in range(50):
            for s in states:
                s[3:4].zero_()
        e1.record(); torch.cuda.synchronize()
        fill = e0.elapsed_time(e1) / 50 * 1e3
    us = tot - fill
    print("  after %-50s %6.1f us  (+%.1f)" % (names[ph], us, us - prev))
    prev = us
prob.kernel_variant = base
