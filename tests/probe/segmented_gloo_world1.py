"""Ad-hoc probe (not a test): the sharded code path (segmented hipGraph + gloo collectives) with a world of ONE rank,
i.e. without two processes time-slicing the GPU -- separates communicator/segment cost from GPU sharing."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "vi-hds_amd")]
import torch, torch.distributed as dist
from vihds import parallel, synthetic
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29577", RANK="0", WORLD_SIZE="1")
dist.init_process_group(sys.argv[1] if len(sys.argv) > 1 else "gloo")
shard = parallel.SampleShard(0, 1)
for mode in ("graph", "eager"):
    args, settings, data, parameters, model, training = synthetic.build(
        "dr_constant_icml", 36, 200, solver="rk4", device="cuda:0", seed=1, shard=shard, u_rng="kernel",
        conditioner_rng="kernel", hip_graph=(mode == "graph"), nan_check_every=0, fused_ode_training=True)
    model.train()
    batch = training.train_data
    step = training.graph_step if mode == "graph" else training.step
    for _ in range(20): step(batch)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(200): loss = step(batch)
    torch.cuda.synchronize()
    print("%s world=1 %s: %.3f ms/step loss %.2f %s" % (dist.get_backend(), mode, (time.perf_counter() - t0) / 200 * 1e3, float(loss), parallel.STATS), flush=True)
dist.destroy_process_group()
