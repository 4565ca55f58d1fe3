"""Worker for the multi-rank GPU test (launched by torch.distributed.run, or directly for the single-process
reference): a few training steps of the synthetic dr_constant_icml workload with the IWAE-sample axis sharded over the
ranks, printing the loss trajectory from rank 0."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "vi-hds_amd")]
import torch  # noqa: E402

from vihds import parallel, synthetic  # noqa: E402


def main():
    mode, steps, s_total = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    shard = parallel.init_from_env()
    rank = shard.rank if shard is not None else 0
    replica = None
    if mode.startswith("dp-"):  # data parallel over rows; "dp-same-*": every replica gets the same rows and draws
        replica, shard = (parallel.RowReplica(shard.rank, shard.world) if shard is not None else None), None
    same = mode.startswith("dp-same")
    mode = mode.split("-")[-1]
    dev = "cuda:%s" % os.environ.get("VIHDS_FORCE_DEVICE", os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(dev)
    args, settings, data, parameters, model, training = synthetic.build(
        "dr_constant_icml", 12, s_total, solver="midpoint", device=dev, seed=5, shard=shard, replica=replica,
        replica_same_data=same, u_rng="kernel",
        conditioner_rng="kernel", hip_graph=(mode in ("graph", "graph4")), nan_check_every=0, fused_ode_training=True)
    model.train()
    batch = training.train_data
    if mode == "graph4":  # four consecutive steps per graph launch (multi-rank: needs the collectives inside the graph)
        training.use_graph = True
        losses = []
        for _ in range(steps // 4):
            training.graph_step(batch, repeat=4)
            losses += [float(x) for x in training.last_losses]
    else:
        step = training.graph_step if mode == "graph" else training.step
        losses = [float(step(batch)) for _ in range(steps)]
    if rank == 0:
        print("CAPTURED %d" % int(bool(getattr(training, "collectives_captured", False))), flush=True)
        print("LOSSES " + json.dumps(losses), flush=True)
    if shard is not None or replica is not None:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
