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

    def _copy_segments(self, src_ptrs, dst_ptrs, ns):
        import ctypes as C
        k = len(ns)
        if k == 0:
            return
        src = (C.c_void_p * k)(*src_ptrs)
        dst = (C.c_void_p * k)(*dst_ptrs)
        n = (C.c_int64 * k)(*ns)
        check(self.lib.g2v_copy_segments(src, dst, n, k, torch.cuda.current_stream().cuda_stream), "copy_segments")

    def gather_grads(self):
        """Copy (device-to-device, stream ordered, ONE launch per 48 tensors) each .grad into the flat grad buffer.  A parameter
        whose .grad is None this step gets zeros and is remembered in `self.skipped` (adjacent ones merged into one range):
        torch.optim.Adam (the reference's optimizer) leaves such a parameter and its moments completely untouched, and
        clip_grad_norm_ ignores it; step() reproduces that."""
        from . import ops
        ops.join_side()          # (gradients produced on side branches of the iteration: ops.side_branches)
        src, dst, ns = [], [], []
        self.skipped = []
        base = self.gflat.data_ptr()
        if getattr(self, "_had_grad", None) is None:
            self._had_grad = [False] * len(self.params)
        for i, (p, o) in enumerate(zip(self.params, self.offsets)):
            g = p.grad
            if g is not None:
                if g.dtype != torch.float32 or not g.is_contiguous():
                    g = g.contiguous().float()
                    p.grad = g
                src.append(g.data_ptr())
                self._had_grad[i] = True
            else:
                src.append(None)
                if not self._had_grad[i]:
                    # never received a gradient: its moments are exactly zero, and the fused step with g = m = v = 0 writes
                    # m = v = 0 and p - lr * 0 / (0 + eps) = p back -> nothing to keep (Part d without attention: the
                    # encoder's second layer, every iteration)
                    dst.append(base + 4 * o)
                    ns.append(p.numel())
                    continue
                span = (p.numel() + 3) // 4 * 4
                if self.skipped and self.skipped[-1][0] + self.skipped[-1][1] == o:
                    self.skipped[-1] = (self.skipped[-1][0], self.skipped[-1][1] + span)
                else:
                    self.skipped.append((o, span))
            dst.append(base + 4 * o)
            ns.append(p.numel())
        self._copy_segments(src, dst, ns)

    def step(self, lr, betas=(0.5, 0.999), eps=1e-8, max_norm=5.0, grad_scale=1.0):
        st = torch.cuda.current_stream().cuda_stream
        # grad-less tensors: a zero gradient adds nothing to the clip norm, but the fused kernel would still decay their
        # moments and move them by stale momentum -> keep (param, m, v) of those ranges (one launch) and put them back
        # afterwards (one launch).  (Their bias-correction step count is the global one here, a per-parameter one in torch:
        # only visible for a tensor that skips steps AND later receives gradients again.)
        skipped = getattr(self, "skipped", [])
        if skipped:
            total = sum(n for _, n in skipped)
            if getattr(self, "_keep", None) is None or self._keep.numel() < 3 * total:
                self._keep = torch.empty(3 * total, device=self.flat.device)
            live, kept, ns, off = [], [], [], 0
            for o, n in skipped:
                for buf in (self.flat, self.m, self.v):
                    live.append(buf.data_ptr() + 4 * o)
                    kept.append(self._keep.data_ptr() + 4 * off)
                    ns.append(n)
                    off += n
            self._copy_segments(live, kept, ns)
        check(self.lib.g2v_clip_adam_step(self.flat.data_ptr(), self.gflat.data_ptr(), self.m.data_ptr(), self.v.data_ptr(),
                                          self.n, self.partial.data_ptr(), self.step_counter.data_ptr(),
                                          self.gnorm.data_ptr(), max_norm, grad_scale, lr, betas[0], betas[1], eps, st))
        if skipped:
            self._copy_segments(kept, live, ns)


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
