"""Ad-hoc probe (not a test): the row-replica data-parallel step (one gradient all-reduce between two hipGraph
segments) with a world of ONE rank on RCCL -- launch / communicator-call cost without wire latency."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "vi-hds_amd")]
import torch, torch.distributed as dist
from vihds import parallel, synthetic
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29578", RANK="0", WORLD_SIZE="1")
dist.init_process_group(sys.argv[1] if len(sys.argv) > 1 else "nccl")
replica = parallel.RowReplica(0, 1)
args, settings, data, parameters, model, training = synthetic.build(
    "dr_constant_icml", 36, 200, solver="rk4", device="cuda:0", seed=1, replica=replica, u_rng="kernel",
    conditioner_rng="kernel", hip_graph=True, nan_check_every=0, fused_ode_training=True, learning_rate=0.001)
model.train()
batch = training.train_data
for _ in range(20): training.graph_step(batch)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(500): loss = training.graph_step(batch)
torch.cuda.synchronize()
print("%s world=1 row replica, graph: %.3f ms/step loss %.2f %s" % (dist.get_backend(), (time.perf_counter() - t0) / 500 * 1e3, float(loss), parallel.STATS), flush=True)
dist.destroy_process_group()
