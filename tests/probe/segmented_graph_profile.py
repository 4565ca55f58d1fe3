"""Ad-hoc probe (two ranks on ONE GPU over gloo): where the time of the SEGMENTED captured step goes (the multi-rank step cut
at its collectives, vihds/parallel.py SegmentedGraph) against the eager step.
   python tests/probe/segmented_graph_profile.py           (starts its two ranks itself)"""
import cProfile, io, os, pstats, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "vi-hds_amd")]
if "RANK" not in os.environ:
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29577", WORLD_SIZE="2")
    ps = [subprocess.Popen([sys.executable, os.path.abspath(__file__)], env=dict(env, RANK=str(r), LOCAL_RANK="0")) for r in range(2)]
    sys.exit(max(p.wait() for p in ps))
import torch
import torch.distributed as dist
from vihds import parallel, synthetic
rank = int(os.environ["RANK"])
torch.cuda.set_device(0)
dist.init_process_group("gloo", rank=rank, world_size=2)
kw = dict(solver="rk4", device="cuda:0", seed=0, u_rng="kernel", conditioner_rng="kernel", nan_check_every=0, learning_rate=0.001)
for graph in (False, True):
    args, settings, data, parameters, model, training = synthetic.build("dr_constant_icml", 36, 200, replica=parallel.RowReplica(rank, 2, None), hip_graph=graph, **kw)
    model.train()
    batch = training.train_data
    step = training.graph_step if graph else training.step
    for _ in range(5):
        step(batch)
    torch.cuda.synchronize(); dist.barrier()
    pr = cProfile.Profile()
    t0 = time.perf_counter()
    pr.enable()
    for _ in range(20):
        step(batch)
    torch.cuda.synchronize()
    pr.disable()
    el = time.perf_counter() - t0
    if rank == 0:
        print("graph" if graph else "eager", "%.2f ms per step" % (1e3 * el / 20))
        s = io.StringIO()
        pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(18)
        print(s.getvalue()[:3500])
dist.destroy_process_group()
