import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "vi-hds_amd")
for p in (ROOT, PKG, os.path.dirname(os.path.abspath(__file__))):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # The oracle's tensors are a few thousand elements: on a 128-thread host PyTorch's default intra-op pool makes every op
    # ~10x SLOWER than on 8 threads (bench.py's cpu_baseline probe: 3.8 s against 0.35 s per step), and the oracle is most of
    # the GPU suite's wall time.
    try:
        import torch

        torch.set_num_threads(min(8, torch.get_num_threads()))
    except ImportError:
        pass
