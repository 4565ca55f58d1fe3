"""What a short timed window costs beyond its kernels: the driver's `--steps 20` window is one graph launch (20 captured steps,
~1.35 ms of kernels) between two synchronisations.  Measures that window with (a) torch.cuda.synchronize(), (b) a spin on
event.query() followed by the synchronize, under the default runtime and with HSA_ENABLE_INTERRUPT=0 (signals polled instead of
interrupt-driven); the evaluation pass's per-pass synchronisation is the same cost.  usage: python tests/probe/sync_latency.py"""
import os
import statistics
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "vi-hds_amd"))


def child():
    import torch
    from vihds import synthetic

    args, settings, data, parameters, model, training = synthetic.build(
        "dr_constant_icml", 36, 200, solver="rk4", device="cuda:0", seed=1, u_rng="kernel", conditioner_rng="kernel",
        hip_graph=True, nan_check_every=0, learning_rate=0.001)
    model.train()
    batch = training.train_data
    training.graph_step(batch)
    for k in (20, 32):
        for _ in range(3):
            training.graph_step(batch, repeat=k)
    torch.cuda.synchronize()
    ev = torch.cuda.Event()
    out = {}
    for name in ("synchronize", "spin_then_synchronize"):
        for k in (20, 640):
            ts = []
            for _ in range(40 if k == 20 else 5):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(k // 20 if k == 20 else k // 32):
                    training.graph_step(batch, repeat=20 if k == 20 else 32)
                if name != "synchronize":
                    ev.record()
                    while not ev.query():
                        pass
                torch.cuda.synchronize()
                ts.append(time.perf_counter() - t0)
            out["%s_%d" % (name, k)] = 1e3 * statistics.median(ts) / k
    print(os.environ.get("HSA_ENABLE_INTERRUPT", "default"), {k: round(v, 5) for k, v in out.items()}, flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1:
        child()
    else:
        for env in ({}, {"HSA_ENABLE_INTERRUPT": "0"}):
            e = dict(os.environ)
            e.update(env)
            subprocess.call([sys.executable, os.path.abspath(__file__), "child"], env=e)
