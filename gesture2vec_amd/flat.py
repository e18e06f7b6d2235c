"""FlatParams: re-home a list of nn.Parameters into ONE flat fp32 buffer (+ flat grad / Adam m / Adam v) so that
clip_grad_norm_ + Adam is the single fused kernel pair g2v_clip_adam_step (train_eval/train_seq2seq.py:743-744)."""
from __future__ import annotations

from typing import List

import torch

from . import _lib, ops
from ._lib import check


class FlatParams:
    def __init__(self, params: List[torch.nn.Parameter]):
        self.params = [p for p in params if p.requires_grad]
        assert self.params, "no trainable parameters"
        dev = self.params[0].device
        if dev.type != "cuda":
            raise RuntimeError("FlatParams needs GPU parameters (there is no CPU path)")
        self.lib = _lib.load()
        self.offsets, off = [], 0
        for p in self.params:
            self.offsets.append(off)
            off += (p.numel() + 3) // 4 * 4
        self.n = off
        self.flat = torch.zeros(off, device=dev)
        self.gflat = torch.zeros(off, device=dev)
        self.m = torch.zeros(off, device=dev)
        self.v = torch.zeros(off, device=dev)
        self.step_counter = torch.zeros(1, dtype=torch.int32, device=dev)
        self.partial = torch.zeros(self.lib.g2v_adam_blocks(off), device=dev)
        self.gnorm = torch.zeros(1, device=dev)
        for p, o in zip(self.params, self.offsets):
            v = self.flat[o:o + p.numel()].view(p.shape)
            v.copy_(p.data)
            p.data = v

    def gather_grads(self):
        """Copy (device-to-device, stream ordered) each .grad into the flat grad buffer.  A parameter whose .grad is None
        this step is remembered in `self.skipped`: torch.optim.Adam (the reference's optimizer) leaves such a parameter
        and its moments completely untouched, and clip_grad_norm_ ignores it; step() reproduces that."""
        self.gflat.zero_()
        self.skipped = []
        for p, o in zip(self.params, self.offsets):
            if p.grad is not None:
                self.gflat[o:o + p.numel()].view(p.shape).copy_(p.grad)
            else:
                self.skipped.append((o, p.numel()))

    def step(self, lr, betas=(0.5, 0.999), eps=1e-8, max_norm=5.0, grad_scale=1.0):
        st = torch.cuda.current_stream().cuda_stream
        # grad-less tensors: a zero gradient adds nothing to the clip norm, but the fused kernel would still decay their
        # moments and move them by stale momentum -> keep (param, m, v) of those ranges and put them back afterwards.
        # (Their bias-correction step count is the global one here, a per-parameter one in torch: only visible for a tensor
        # that skips steps AND later receives gradients again.)
        saved = [(o, n, self.flat[o:o + n].clone(), self.m[o:o + n].clone(), self.v[o:o + n].clone())
                 for o, n in getattr(self, "skipped", [])]
        check(self.lib.g2v_clip_adam_step(self.flat.data_ptr(), self.gflat.data_ptr(), self.m.data_ptr(), self.v.data_ptr(),
                                          self.n, self.partial.data_ptr(), self.step_counter.data_ptr(),
                                          self.gnorm.data_ptr(), max_norm, grad_scale, lr, betas[0], betas[1], eps, st))
        for o, n, w, m, v in saved:
            self.flat[o:o + n].copy_(w)
            self.m[o:o + n].copy_(m)
            self.v[o:o + n].copy_(v)


class FlatClipAdam:
    """Optimizer-like wrapper (zero_grad / step) for the thin models (DAE): clip_grad_norm_(max_norm) + Adam, fused."""

    def __init__(self, params, lr, betas=(0.5, 0.999), eps=1e-8, max_norm=5.0):
        self.fp = FlatParams(list(params))
        self.lr, self.betas, self.eps, self.max_norm = float(lr), tuple(betas), float(eps), float(max_norm)

    def zero_grad(self, set_to_none: bool = True):
        for p in self.fp.params:
            p.grad = None

    def step(self):
        self.fp.gather_grads()
        self.fp.step(self.lr, self.betas, self.eps, self.max_norm)
