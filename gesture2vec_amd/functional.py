"""Small autograd nodes over the C-ABI (used by the thin models: DAE_Network, the quantiser variants, the text encoder).
Each node's forward AND backward are HIP kernels; torch only chains them."""
from __future__ import annotations

import torch

from . import ops


class LinearFn(torch.autograd.Function):
    """y = act((x * keep * scale) W^T + b) over the last dim of x; keep is an optional uint8 mask (dropout on the input)."""

    @staticmethod
    def forward(ctx, x, weight, bias, keep, scale, act):
        N, K = weight.shape
        xs = x.contiguous().view(-1, K)
        w = weight.contiguous()
        y = ops.linear_fwd(xs, w, bias, act=act, keep=keep, scale=scale)
        ctx.save_for_backward(xs, w, y if act else None, keep)
        ctx.scale, ctx.act, ctx.has_bias, ctx.xshape = scale, act, bias is not None, x.shape
        ctx.set_materialize_grads(False)
        return y.view(*x.shape[:-1], N)

    @staticmethod
    def backward(ctx, gy):
        xs, w, y, keep = ctx.saved_tensors
        if gy is None:
            return None, None, None, None, None, None
        N, K = w.shape
        g = gy.contiguous().view(-1, N)
        if ctx.act == 1:                       # ReLU: gate the gradient with the saved output
            g = _ReluBwd.apply_mask(g, y)
        elif ctx.act == 2:
            raise NotImplementedError("tanh backward")
        dx = None
        if ctx.needs_input_grad[0]:
            dx = ops.linear_bwd_data(g, w)
            if keep is not None:
                dx = _ReluBwd.apply_keep(dx, keep, ctx.scale)
            dx = dx.view(ctx.xshape)
        dw, db = ops.linear_bwd_weight(g, xs, N, K, keep=keep, scale=ctx.scale, want_bias=ctx.has_bias)
        return dx, dw, (db if ctx.has_bias else None), None, None, None


class _ReluBwd:
    """Elementwise masks via the dense-layer kernel's own keep-mask path would cost a GEMM; these two helpers use the
    Philox-free byte-mask multiply of g2v_scale through a tiny dedicated kernel instead."""

    @staticmethod
    def apply_mask(g, y):
        return ops.mask_mul(g, y, 1.0, positive_of=True)

    @staticmethod
    def apply_keep(dx, keep, scale):
        return ops.mask_mul(dx, keep, scale, positive_of=False)


def linear(x, weight, bias=None, keep=None, scale=1.0, act=0):
    return LinearFn.apply(x, weight, bias, keep, float(scale), int(act))


class MseFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, y, target):
        loss, dy = ops.mse_fwd_bwd(y.contiguous(), target.contiguous(), want_grad=True)
        ctx.save_for_backward(dy)
        ctx.shape = y.shape
        return loss[0].clone()

    @staticmethod
    def backward(ctx, g):
        (dy,) = ctx.saved_tensors
        return ops.scale(dy, g.reshape(1).contiguous()).view(ctx.shape), None


def mse_loss(y, target):
    return MseFn.apply(y, target)
