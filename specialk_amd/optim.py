"""Fused multi-tensor Adam on the HIP kernel, as a torch.optim.Optimizer.

Arithmetic and state layout follow torch.optim.Adam (no weight decay, no amsgrad) as the
reference configures it in setup_optimiser (lib/TransformerModel.py:155-168); one kernel
launch updates every parameter, its moments and — when bfloat16 compute copies exist — the
bf16 working copy, instead of several launches per tensor.
"""
import torch

from . import _lib as L
from . import ops


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps))
        # Pointer tables travel host -> device asynchronously, in stream order.  The host may run several
        # steps ahead of the GPU, so a staging buffer is only rewritten after the copy that last read it
        # has executed (an event per buffer), and an unchanged table is not uploaded again.
        self._staging = []      # [pinned (6, n) int64, event recorded after its last upload]
        self._next_staging = 0
        self._last_rows = None  # rows of the table the device currently holds in self._last_dev
        self._last_dev = None

    N_STAGING = 4

    def _tables(self, plist):
        n = len(plist)
        rows = []
        for p in plist:
            st = self.state[p]
            c = ops.registered_copy(p)
            rows.append((p.data_ptr(), p.grad.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(),
                         c.data_ptr() if c is not None else 0, p.numel()))
        device = plist[0].device
        if rows != self._last_rows or self._last_dev is None or self._last_dev.device != device:
            host = torch.tensor(rows, dtype=torch.int64).t().contiguous()  # (6, n)
            if self._staging and self._staging[0][0].shape != host.shape:
                for _, ev in self._staging:
                    ev.synchronize()
                self._staging, self._next_staging = [], 0
            if len(self._staging) < self.N_STAGING:
                self._staging.append([torch.empty_like(host).pin_memory(), torch.cuda.Event()])
                slot = self._staging[-1]
            else:
                slot = self._staging[self._next_staging]
                self._next_staging = (self._next_staging + 1) % self.N_STAGING
                slot[1].synchronize()  # the upload that last read this buffer has run
            slot[0].copy_(host)
            self._last_dev = slot[0].to(device, non_blocking=True)
            slot[1].record(torch.cuda.current_stream(device))
            self._last_rows = rows
        dev = self._last_dev
        return {"p": dev[0], "g": dev[1], "m": dev[2], "v": dev[3], "b": dev[4], "n": dev[5]}, n

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for group in self.param_groups:
            plist = []
            for p in group["params"]:
                if p.grad is None:
                    continue
                if p.dtype != torch.float32 or p.grad.dtype != torch.float32 or not p.is_contiguous():
                    raise RuntimeError("FusedAdam needs contiguous fp32 parameters and gradients")
                if not p.grad.is_contiguous():
                    p.grad = p.grad.contiguous()
                st = self.state[p]
                if not st:
                    st["step"] = 0
                    st["exp_avg"] = torch.zeros_like(p)
                    st["exp_avg_sq"] = torch.zeros_like(p)
                plist.append(p)
            if not plist:
                continue
            steps = {self.state[p]["step"] for p in plist}
            if len(steps) != 1:
                raise RuntimeError("FusedAdam: parameters of one group must share a step count")
            step = steps.pop() + 1
            tables, n = self._tables(plist)
            b1, b2 = group["betas"]
            L.adam_multi(tables, n, max(p.numel() for p in plist), group["lr"], b1, b2, group["eps"], step)
            ops.refresh_derived()  # stacked / transposed working copies follow the new weights
            for p in plist:
                self.state[p]["step"] = step
        return loss
