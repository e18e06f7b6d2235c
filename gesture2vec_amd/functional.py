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
        # many rows: apply the dropout mask ONCE (one elementwise pass) and give the mask-free streaming / wave-autonomous
        # kernels a plain operand for the forward product and the weight gradient (a mask in the contraction pins both to the
        # LDS-tiled generic kernels: 353 vs ~100 us for the (81920 x 400)^T (81920 x 600) gradient of Part d's encoder)
        ctx.premasked = keep is not None and xs.shape[0] >= 4096
        if ctx.premasked:
            xs = ops.mask_mul(xs, keep, scale, positive_of=False)
            y = ops.linear_fwd(xs, w, bias, act=act)
        else:
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
        if ctx.premasked:
            dw, db = ops.linear_bwd_weight(g, xs, N, K, want_bias=ctx.has_bias)
        else:
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


class LinearPairFn(torch.autograd.Function):
    """(x W_a^T + b_a, x W_b^T + b_b): the two directions' input projections of a bidirectional GRU layer on the same input --
    one launch forward where that pays (ops.linear_fwd_pair: bitwise two linear_fwd calls), the two weight gradients as one
    batched launch, dx = g_a W_a + g_b W_b accumulated in place.  keep / scale (round 6): nn.GRU's inter-layer dropout on the
    shared input -- x * keep * scale is formed ONCE (the two LinearFn nodes it replaces each masked it in their forward, their
    weight-gradient and their dx launch, and autograd added the two dx), so the mask-free kernels of both passes apply."""

    @staticmethod
    def forward(ctx, x, w_a, b_a, w_b, b_b, keep=None, scale=1.0):
        N, K = w_a.shape
        xs = x.contiguous().view(-1, K)
        if keep is not None:
            keep = keep.contiguous().view(-1, K)
            xs = ops.mask_mul(xs, keep, scale, positive_of=False)
        w_a, w_b = w_a.contiguous(), w_b.contiguous()
        ya, yb = ops.linear_fwd_pair(xs, w_a, b_a, w_b, b_b)
        ctx.save_for_backward(xs, w_a, w_b, keep)
        ctx.xshape, ctx.scale = x.shape, float(scale)
        ctx.set_materialize_grads(False)
        return ya.view(*x.shape[:-1], N), yb.view(*x.shape[:-1], N)

    @staticmethod
    def backward(ctx, ga, gb):
        xs, w_a, w_b, keep = ctx.saved_tensors
        N, K = w_a.shape
        gs = [g.contiguous().view(-1, N) if g is not None else None for g in (ga, gb)]
        out = [None] * 4
        live = [k for k in (0, 1) if gs[k] is not None]
        if live:
            # nothing downstream waits for the weight gradients: a side branch of the iteration (ops.side_branches), forked
            # BEFORE the input gradient's launches so that the two run side by side
            with ops.side_branch(1, keep=[xs] + [gs[k] for k in live], rows=xs.shape[0], min_rows=22000):
                items = []
                for k in live:
                    out[2 * k] = torch.empty((N, K), dtype=torch.float32, device=xs.device)
                    out[2 * k + 1] = torch.empty((N,), dtype=torch.float32, device=xs.device)
                    items.append((gs[k], xs, out[2 * k], out[2 * k + 1]))
                ops.linear_bwd_weight_batch(items, N, K, M=xs.shape[0])
        dx = None
        if ctx.needs_input_grad[0]:
            for k in live:
                if dx is None:
                    dx = ops.linear_bwd_data(gs[k], (w_a, w_b)[k])
                else:
                    ops.linear_bwd_data(gs[k], (w_a, w_b)[k], out=dx, accumulate=True)
        if dx is not None and keep is not None:
            dx = ops.mask_mul(dx, keep, ctx.scale, positive_of=False)
        return (dx.view(ctx.xshape) if dx is not None else None), out[0], out[1], out[2], out[3], None, None


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


class EmbeddingFn(torch.autograd.Function):
    """rows = table[ids] * keep * scale (nn.Embedding + optional fused dropout); gradient = scatter-add into the table."""

    @staticmethod
    def forward(ctx, table, ids, keep, scale):
        ids = ids.contiguous().view(-1)
        out = ops.embedding_fwd(table.contiguous(), ids, keep, scale)
        ctx.save_for_backward(ids, keep)
        ctx.V, ctx.scale = table.shape[0], scale
        ctx.set_materialize_grads(False)
        return out

    @staticmethod
    def backward(ctx, g):
        ids, keep = ctx.saved_tensors
        if g is None:
            return None, None, None, None
        return ops.embedding_bwd(g.contiguous(), ids, ctx.V, keep, ctx.scale), None, None, None


class EmbedProjectPairFn(torch.autograd.Function):
    """gi_p = table[ids] W_p^T + b_p for the two directions p of a bidirectional GRU layer fed straight by nn.Embedding (ref
    text2embedding_model.py EncoderRNN: self.embedding(input_seqs) -> self.gru, no dropout in between), for MANY rows per
    vocabulary entry: the projection commutes with the gather, so
        forward : G_p = table W_p^T + b_p (V x 3H, both directions in one launch), gi_p = G_p[ids];
        backward: S_p = the scatter-add of dgi_p by id (V x 3H: the embedding backward on 3H-wide rows), then
                  d_table = S_f W_f + S_b W_b,  dW_p = S_p^T table,  db_p = column sums of S_p
    -- three V-row products per direction instead of three (rows)-row ones.  The forward is bitwise the gather-then-project
    form (the same dot product per element); the backward equals autograd's to summation order."""

    @staticmethod
    def forward(ctx, table, ids, w_f, b_f, w_b, b_b):
        table, ids = table.contiguous(), ids.contiguous().view(-1)
        w_f, w_b = w_f.contiguous(), w_b.contiguous()
        g_f, g_b = ops.linear_fwd_pair(table, w_f, b_f, w_b, b_b)
        gi_f, gi_b = ops.embedding_fwd(g_f, ids), ops.embedding_fwd(g_b, ids)
        ctx.save_for_backward(table, ids, w_f, w_b)
        ctx.set_materialize_grads(False)
        return gi_f, gi_b

    @staticmethod
    def backward(ctx, d_f, d_b):
        table, ids, w_f, w_b = ctx.saved_tensors
        V, (N, K) = table.shape[0], w_f.shape
        d_table, out, items = None, [], []
        for d, w in ((d_f, w_f), (d_b, w_b)):
            if d is None:
                out += [None, None]
                continue
            # (the second direction scatter-adds by the same words: its counting sort is the first one's, still in the workspace)
            s = ops.embedding_bwd(d.contiguous().view(-1, N), ids, V, reuse_sort=d_table is not None)
            if d_table is None:
                d_table = ops.linear_bwd_data(s, w)
            else:
                ops.linear_bwd_data(s, w, out=d_table, accumulate=True)
            dw = torch.empty((N, K), dtype=torch.float32, device=table.device)
            db = torch.empty((N,), dtype=torch.float32, device=table.device)
            items.append((s, table, dw, db))
            out += [dw, db]
        if items:
            ops.linear_bwd_weight_batch(items, N, K, M=V)      # both directions' dW = S^T table in one launch
        return d_table, None, out[0], out[1], out[2], out[3]


class BatchNormReluFn(torch.autograd.Function):
    """nn.BatchNorm1d(+ReLU) on (B,H): batch statistics (and running-stat update) in training, running stats in eval."""

    @staticmethod
    def forward(ctx, x, weight, bias, running_mean, running_var, training, relu):
        x = x.contiguous()
        y, sm, si = ops.batchnorm_fwd(x, weight, bias, running_mean, running_var, training, relu)
        if training:
            ctx.save_for_backward(x, y, weight, sm, si)
        ctx.training, ctx.relu = training, relu
        ctx.set_materialize_grads(False)
        return y

    @staticmethod
    def backward(ctx, gy):
        if gy is None:
            return (None,) * 7
        if not ctx.training:
            raise NotImplementedError("BatchNorm backward in eval mode is not used on the training path")
        x, y, weight, sm, si = ctx.saved_tensors
        dx, dw, db = ops.batchnorm_bwd(gy.contiguous(), x, y, weight, sm, si, ctx.relu)
        return dx, dw, db, None, None, None, None


class GRUDirFn(torch.autograd.Function):
    """One direction of one GRU layer over a (T,B,.) sequence given its input projections gi = x W_ih^T + b_ih.
    Returns (hs (T,B,H), h_n (B,H)).  `lengths` (int32, B) gives pack_padded_sequence semantics (h0 must be None)."""

    @staticmethod
    def forward(ctx, gi, w_hh, b_hh, h0, lengths, reverse):
        T, B, G = gi.shape
        H = G // 3
        gi = gi.contiguous()
        w = w_hh.contiguous()
        hs, h_n, gates = ops.gru_seq_fwd(gi, w, b_hh, T, B, H, h0=h0.contiguous() if h0 is not None else None,
                                         lengths=lengths, reverse=reverse, save_gates=True)
        ctx.save_for_backward(hs, gates, w, h0 if h0 is not None else None, lengths)
        ctx.dims, ctx.reverse, ctx.has_h0 = (T, B, H), reverse, h0 is not None
        ctx.set_materialize_grads(False)
        return hs, h_n

    @staticmethod
    def backward(ctx, g_hs, g_hn):
        hs, gates, w, h0, lengths = ctx.saved_tensors
        T, B, H = ctx.dims
        d_hs = g_hs.contiguous() if g_hs is not None else None
        d_hn = g_hn.contiguous() if g_hn is not None else None
        if d_hs is None and d_hn is None:
            return (None,) * 6
        dgi, dgh, dh0 = ops.gru_seq_bwd(d_hs, H, d_hn, hs, H, h0, gates, w, T, B, H, lengths=lengths, reverse=ctx.reverse,
                                        want_dh0=ctx.has_h0)
        # h_prev sequence for the W_hh gradient (data movement only): shifted hs with the initial state at the open end
        first = h0.unsqueeze(0) if h0 is not None else torch.zeros((1, B, H), dtype=hs.dtype, device=hs.device)
        hprev = torch.cat([hs[1:], first], 0) if ctx.reverse else torch.cat([first, hs[:-1]], 0)
        dw, db = ops.linear_bwd_weight(dgh, hprev.contiguous(), 3 * H, H, M=T * B)
        return dgi, dw, db, (dh0 if ctx.has_h0 else None), None, None


class GRUBiDirFn(torch.autograd.Function):
    """Both directions of one bidirectional GRU layer in ONE launch each way (at small batch a direction fills only B/16
    workgroups, so one launch per direction leaves the chip idle twice as long).  Inputs: the two input projections
    gi_f / gi_b (T,B,3H) and the recurrent weights; returns (hs_f, hn_f, hs_b, hn_b)."""

    @staticmethod
    def forward(ctx, gi_f, gi_b, w_f, b_f, w_b, b_b, lengths, packed=None, gather=None):
        """packed = (T, B, row_off): gi_f / gi_b are (sum(lengths), 3H) PACKED arrays (ops.gru_dirs_fwd); their gradients
        come back packed too.
        gather (round 6; needs packed) = int64 ids, one per packed position: gi_f / gi_b are (V, 3H) TABLES (the projected
        embedding table) and position r reads row gather[r] inside the recurrent kernel (include/g2v.h: gi_gather) -- the two
        (positions x 3H) gathers are never materialised; the gradients come back as the scatter-add by id, (V, 3H) like the inputs."""
        if packed is None:
            T, B, G = gi_f.shape
            row_off = None
        else:
            T, B, row_off = packed
            G = gi_f.shape[1]
        if gather is not None and packed is None:
            raise ValueError("GRUBiDirFn: gather needs the packed layout")
        H = G // 3
        dev = gi_f.device
        gi_f, gi_b, w_f, w_b = gi_f.contiguous(), gi_b.contiguous(), w_f.contiguous(), w_b.contiguous()
        out = []
        dirs = []
        # one extra zero step in front of (behind, for the reverse direction) the states: the BPTT's weight-gradient product
        # reads "the state before step t" as a VIEW of the same buffer (it was two 65 MB cat copies per call at B = 4096).
        # Both directions in one allocation, reverse first: its zero step (the last) and the forward's (the first) are adjacent
        # -> one fill.
        both = torch.empty((2, T + 1, B, H), dtype=torch.float32, device=dev)
        both.view(2 * (T + 1), B, H)[T:T + 2].zero_()
        for gi, w, b, rev in ((gi_f, w_f, b_f, False), (gi_b, w_b, b_b, True)):
            full = both[0 if rev else 1]
            hs = full[:T] if rev else full[1:]
            h_n = torch.empty((B, H), dtype=torch.float32, device=dev)
            gates = torch.empty((T, B, 4 * H), dtype=torch.float32, device=dev)
            dirs.append(dict(gi=gi, w_hh=w, b_hh=b, h0=None, hs=hs, h_n=h_n, gates=gates, reverse=rev, full=full, gi_gather=gather))
            out += [hs, h_n]
        ops.gru_dirs_fwd(dirs, T, B, H, lengths=lengths, hs_ld=H, row_off=row_off)
        ctx.save_for_backward(dirs[0]["full"], dirs[0]["gates"], w_f, dirs[1]["full"], dirs[1]["gates"], w_b, lengths, gather)
        ctx.dims = (T, B, H)
        ctx.table_rows = gi_f.shape[0]
        ctx.packed = (row_off, gather.numel() if gather is not None else gi_f.shape[0]) if packed is not None else None
        ctx.set_materialize_grads(False)
        return tuple(out)

    @staticmethod
    def backward(ctx, g_hs_f, g_hn_f, g_hs_b, g_hn_b):
        full_f, gates_f, w_f, full_b, gates_b, w_b, lengths, gather = ctx.saved_tensors
        T, B, H = ctx.dims
        hs_f, hprev_f = full_f[1:], full_f[:T]                           # (forward() laid the zero step beside the states)
        hs_b, hprev_b = full_b[:T], full_b[1:]
        if all(g is None for g in (g_hs_f, g_hn_f, g_hs_b, g_hn_b)):
            return (None,) * 9
        row_off, n_packed = ctx.packed if ctx.packed is not None else (None, 0)
        dev = hs_f.device
        dirs, outs = [], []
        for g_hs, g_hn, hs, gates, w, rev in ((g_hs_f, g_hn_f, hs_f, gates_f, w_f, False), (g_hs_b, g_hn_b, hs_b, gates_b, w_b, True)):
            d_hs = g_hs.contiguous() if g_hs is not None else None
            d_hn = g_hn.contiguous() if g_hn is not None else None
            if d_hs is None and d_hn is None:
                d_hn = torch.zeros((B, H), dtype=torch.float32, device=dev)
            dgi = torch.empty((T, B, 3 * H) if row_off is None else (n_packed, 3 * H), dtype=torch.float32, device=dev)
            dgh = torch.empty((T, B, 3 * H), dtype=torch.float32, device=dev)
            dirs.append(dict(d_hs=d_hs, d_hn=d_hn, hs=hs, h0=None, gates=gates, w_hh=w, dgi=dgi, dgh=dgh, dh0=None, reverse=rev))
            outs.append((dgi, dgh))
        # (the W_hh-resident BPTT only while nothing else is in flight: beside a side branch's weight gradients -- the layer above,
        #  with attention -- it lost to the streaming kernel, 8.54 against 8.45 ms at B = 4096; alone it wins, 3.00 against 3.07)
        ops.gru_dirs_bwd(dirs, T, B, H, lengths=lengths, d_hs_ld=H, hs_ld=H, row_off=row_off, allow_resident=not ops.side_pending())
        items = []
        # the W_hh gradients beside the input side's chain (large batches: ops.side_branch has the measured rule)
        with ops.side_branch(1, keep=[outs[0][1], outs[1][1], full_f, full_b], rows=T * B, min_rows=32768):
            for (dgi, dgh), hprev in zip(outs, (hprev_f, hprev_b)):
                items.append((dgh, hprev, torch.empty((3 * H, H), dtype=torch.float32, device=dev),
                              torch.empty((3 * H,), dtype=torch.float32, device=dev)))
            ops.linear_bwd_weight_batch(items, 3 * H, H, M=T * B)
        d_gi = [outs[0][0], outs[1][0]]
        if gather is not None:      # the inputs were tables: their gradients are the scatter-adds of dgi by id (one sort for both)
            d_gi = [ops.embedding_bwd(d_gi[0], gather, ctx.table_rows), ops.embedding_bwd(d_gi[1], gather, ctx.table_rows, reuse_sort=True)]
        return d_gi[0], d_gi[1], items[0][2], items[0][3], items[1][2], items[1][3], None, None, None


class CrossEntropyFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, targets, skip_rows=0):
        """Mean cross entropy over rows [skip_rows:] of logits (M,K) / targets (M); the gradient comes back for all M rows
        (zero on the skipped ones).  skip_rows lets the caller hand over a whole step-major (S*B, K) buffer whose first step is
        not part of the loss WITHOUT slicing it (a slice's backward is a zero fill + a copy of the full array)."""
        lg = logits.contiguous()
        tg = targets.contiguous().view(-1)
        if skip_rows:
            dl = torch.empty_like(lg)
            dl[:skip_rows].zero_()
            loss, _ = ops.cross_entropy_fwd_bwd(lg[skip_rows:], tg[skip_rows:], want_grad=True, dl_out=dl[skip_rows:])
        else:
            loss, dl = ops.cross_entropy_fwd_bwd(lg, tg, want_grad=True)
        ctx.save_for_backward(dl)
        ctx.shape = logits.shape
        return loss.view(())          # (a fresh one-element array per call: no copy needed)

    @staticmethod
    def backward(ctx, g):
        (dl,) = ctx.saved_tensors
        return ops.scale(dl, g.reshape(1).contiguous()).view(ctx.shape), None, None


def cross_entropy(logits, targets, skip_rows=0):
    return CrossEntropyFn.apply(logits, targets, skip_rows)


class AttnFn(torch.autograd.Function):
    """Bahdanau attention of one decode step (reference Attn :160-198 + context :353-359) given the two halves of the
    energy pre-activation: hp (B,H) = h W_h^T + b, ep (T,B,H) = enc W_e^T.  Returns (context (B,H), weights (B,T))."""

    @staticmethod
    def forward(ctx, hp, ep, enc, v):
        hp, ep, enc, v = hp.contiguous(), ep.contiguous(), enc.contiguous(), v.contiguous()
        weights, context = ops.attn_fwd(hp, ep, enc, v)
        ctx.save_for_backward(hp, ep, enc, v, weights)
        ctx.set_materialize_grads(False)
        ctx.mark_non_differentiable(weights)
        return context, weights

    @staticmethod
    def backward(ctx, g_ctx, _g_w):
        if g_ctx is None:
            return None, None, None, None
        hp, ep, enc, v, weights = ctx.saved_tensors
        d_hp, d_ep, d_enc, d_v = ops.attn_bwd(g_ctx.contiguous(), hp, ep, enc, v, weights)
        return d_hp, d_ep, d_enc, d_v


class SumHalvesFn(torch.autograd.Function):
    """out = a + b (the two GRU directions, :133-135); the gradient goes unchanged to both."""

    @staticmethod
    def forward(ctx, a, b):
        a, b = a.contiguous(), b.contiguous()
        H = a.shape[-1]
        out = torch.empty_like(a)
        ops.add_halves(a, H, b, H, out, H, a.numel() // H, H)
        return out

    @staticmethod
    def backward(ctx, g):
        return g, g
