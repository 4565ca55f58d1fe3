"""Adam for the training step (reference training.py:82: torch.optim.Adam with default betas / eps), as one HIP
launch over all parameter tensors (csrc/vihds_elbo.hip adam_kernel through vihds_adam_step).

Why not torch's fused Adam: its multi-tensor kernel hands each tensor to blocks in 64k-element chunks, so the 36 000
element encoder matrix is updated by a single workgroup (measured 23 us of a 360 us step at the headline shape) and
the per-tensor step counters cost a second launch.  Here the step counter lives on the device and is advanced by the
kernel itself, so the update is graph-capturable as is; `lr` may be a device scalar for the same reason.
"""
import ctypes

import torch

from vihds import hip


class HipAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, grad_scale=1.0):
        """grad_scale multiplies every gradient as the kernel reads it (1/world after a SUM all-reduce)."""
        super(HipAdam, self).__init__(params, dict(lr=lr, betas=betas, eps=eps, grad_scale=grad_scale))
        self._flat = {}
        self.gate = None  # one device float (the step's loss): a non-finite value makes step() a no-op on the device

    def _group_state(self, gi, group):
        st = self._flat.get(gi)
        if st is None:
            ps = [p for p in group["params"] if p.requires_grad]
            for p in ps:
                if p.dtype != torch.float32 or not p.is_contiguous() or p.device.type != "cuda":
                    raise RuntimeError("HipAdam needs contiguous fp32 parameters on the GPU (there is no CPU path)")
            total = sum(p.numel() for p in ps)
            dev = ps[0].device
            st = {"params": ps, "m": torch.zeros(total, device=dev), "v": torch.zeros(total, device=dev),
                  "state": torch.zeros(4, device=dev)}  # {step count, block ticket, 2 words of vihds_step_tail}
            self._flat[gi] = st
        return st

    @torch.no_grad()
    def step(self, closure=None):
        loss = closure() if closure is not None else None
        L = hip.lib()
        for gi, group in enumerate(self.param_groups):
            if not group["params"]:
                continue
            st = self._group_state(gi, group)
            lr = group["lr"]
            lr_dev = lr if isinstance(lr, torch.Tensor) else None
            beta1, beta2 = group["betas"]
            ps, off = st["params"], 0
            # tensors go in table order; the flat m / v offset of a launch's first tensor is applied to the pointers
            for lo in range(0, len(ps), hip.ADAM_MAX_TENSORS):
                chunk = ps[lo:lo + hip.ADAM_MAX_TENSORS]
                tab = hip.AdamTensors()
                tab.n = len(chunk)
                keep = []
                for k, p in enumerate(chunk):
                    g = p.grad
                    if g is not None and not g.is_contiguous():
                        g = g.contiguous()
                        keep.append(g)
                    tab.size[k] = p.numel()
                    tab.param[k] = p.data_ptr()
                    tab.grad[k] = None if g is None else g.data_ptr()
                n_chunk = sum(p.numel() for p in chunk)
                last = lo + hip.ADAM_MAX_TENSORS >= len(ps)
                # only the last launch of a group may advance the step counter: earlier ones get a scratch copy
                state = st["state"] if last else st["state"].clone()
                rc = L.vihds_adam_step(ctypes.byref(tab), st["m"][off:].data_ptr(), st["v"][off:].data_ptr(),
                                       state.data_ptr(), hip.ptr(lr_dev), 0.0 if lr_dev is not None else float(lr),
                                       beta1, beta2, group["eps"], float(group.get("grad_scale", 1.0)),
                                       hip.ptr(self.gate), hip.current_stream())
                hip.check(rc, "vihds_adam_step")
                off += n_chunk
        return loss

    def step_count(self, group=0):
        st = self._flat.get(group)
        return 0 if st is None else int(st["state"][0].item())
