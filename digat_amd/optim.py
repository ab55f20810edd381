"""The optimiser of the training step on HIP kernels: gradient clipping by the global norm + Adam in three launches.

``ClipAdam`` is ``torch.optim.Adam`` (trainer.py:30: coupled weight decay per parameter group, bias-corrected moments) with
``clip_grad_norm_`` (trainer.py:103-104) folded into the update: ``step(max_norm)`` computes the squared norm of every gradient
(two launches), then one pass over (p, g, m, v) applies ``min(1, max_norm / (norm + 1e-6))`` to the gradients on the fly and
updates the parameters (``digat_clip_adam_step``).  torch's pair is 15 launches and ~340 us of a 5 ms step on the benchmark's
parameters (5.3 M in ~70 tensors + the 26 M-element news table).  The gradients themselves are left unscaled.
"""
from __future__ import annotations

import numpy as np
import torch

from . import _lib

_REC = np.dtype([("p", "<u8"), ("g", "<u8"), ("m", "<u8"), ("v", "<u8"), ("n", "<i8"), ("wd", "<f4"), ("pad", "<i4")])      # digat_opt_tensor


class ClipAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self._chunk = int(_lib.lib().digat_opt_chunk())

    @torch.no_grad()
    def step(self, max_norm: float = 0.0):
        """One update of every parameter that has a gradient; ``max_norm`` > 0 clips by the global norm first (all groups together,
        as ``clip_grad_norm_(model.parameters(), max_norm)``)."""
        groups = {}
        for group in self.param_groups:
            key = (float(group["lr"]), tuple(group["betas"]), float(group["eps"]))
            for p in group["params"]:
                if p.grad is None:
                    continue
                if not (p.is_cuda and p.dtype == torch.float32 and p.grad.dtype == torch.float32 and p.is_contiguous()):
                    raise RuntimeError("ClipAdam: contiguous fp32 parameters on the GPU only")
                st = self.state[p]
                if not st:
                    st["step"] = 0
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                st["step"] += 1
                groups.setdefault((key, st["step"]), []).append((p, p.grad if p.grad.is_contiguous() else p.grad.contiguous(), st,
                                                                 float(group["weight_decay"])))
        if not groups:
            return
        if len(groups) != 1:
            # different hyper-parameters or step counts: the clipping coefficient must still come from ALL gradients; not needed by the
            # trainer (one lr, parameters updated together): keep the arithmetic honest rather than approximate
            raise RuntimeError("ClipAdam: every parameter group must share lr / betas / eps and every parameter its step count")
        ((lr, (b1, b2), eps), step), items = next(iter(groups.items()))
        dev = items[0][0].device
        rec = np.zeros(len(items), dtype=_REC)
        chunk_tensor, chunk_off = [], []
        for k, (p, g, st, wd) in enumerate(items):
            n = p.numel()
            rec[k] = (p.data_ptr(), g.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(), n, wd, 0)
            offs = np.arange(0, n, self._chunk, dtype=np.int64)
            chunk_off.append(offs)
            chunk_tensor.append(np.full(len(offs), k, dtype=np.int32))
        chunk_off = np.concatenate(chunk_off)
        chunk_tensor = np.concatenate(chunk_tensor)
        nch = len(chunk_off)
        a, b = rec.nbytes, rec.nbytes + chunk_off.nbytes
        packed = np.concatenate([rec.view(np.uint8).reshape(-1), chunk_off.view(np.uint8), chunk_tensor.view(np.uint8)])
        # the tables change only when a tensor moved (the gradients are new tensors every step, but the caching allocator hands the
        # same blocks back): the device copy of the last step is reused while the bytes are equal — otherwise one pinned, asynchronous
        # copy (a fresh pinned buffer: the previous copy may still be in flight)
        last = getattr(self, "_tables", None)
        if last is None or last[0].shape != packed.shape or not np.array_equal(last[0], packed) or last[2].device != dev:
            host = torch.from_numpy(packed.copy()).pin_memory()
            self._tables = (packed, host, host.to(dev, non_blocking=True))
        tab = self._tables[2]
        scratch = _lib.workspace((nch + 1) * 4, dev, "clip_adam")
        _lib.check(_lib.lib().digat_clip_adam_step(tab.data_ptr(), tab.data_ptr() + b, tab.data_ptr() + a, nch, scratch.data_ptr(),
                                                   float(max_norm), lr, b1, b2, eps, step, _lib.stream_ptr()), "digat_clip_adam_step")
