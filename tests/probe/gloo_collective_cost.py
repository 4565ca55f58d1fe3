"""Ad-hoc probe (not a test): cost of the step's two collectives with the gloo backend on CUDA tensors (the only way
to run two ranks on a one-GPU box) -- to tell communicator time from everything else in a 2-rank bench run."""
import os, sys, time
import torch, torch.distributed as dist
dist.init_process_group("gloo")
torch.cuda.set_device(0)
pair = torch.randn(2, 36, device="cuda"); gathered = torch.empty(4, 36, device="cuda"); flat = torch.randn(37000, device="cuda")
for name, fn in (("all_gather [2,36]", lambda: dist.all_gather_into_tensor(gathered, pair)),
                 ("all_reduce 37k floats", lambda: dist.all_reduce(flat))):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(50): fn()
    torch.cuda.synchronize()
    if dist.get_rank() == 0: print("%s: %.2f ms" % (name, (time.perf_counter() - t0) / 50 * 1e3), flush=True)
dist.destroy_process_group()
