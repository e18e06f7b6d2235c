"""The S-1 training steps of the text -> gesture-code decoder as ONE autograd node.

`text2embedding_model.forward` (reference model/text2embedding_model.py:701-744) calls `BahdanauAttnDecoderRNN.forward`
(:338-395) once per code position: Embedding + Dropout(0.5) [-> Bahdanau context] -> Linear + BatchNorm1d + ReLU -> GRU(L)
-> Linear(H -> K) -> greedy argmax feedback.  Chaining that through one small autograd node per operator costs, per
training iteration, one weight-gradient product + one slab reduction + one torch add PER STEP AND WEIGHT, and a cat / copy
around every slice -- about two thirds of the ~450 launches of an iteration at B = 128.  Here the forward writes every
per-step activation straight into (S-1, B, .) arrays and the backward is written out by hand:

  * the recurrence (GRU cells, BatchNorm, attention) is walked step by step, data gradients only;
  * everything that does not feed the recurrence is batched over all S-1 steps: dLogits W_out before the loop, d(input)
    = dU W_pre and the embedding gradient after it (without attention), and EVERY weight gradient -- one product over the
    (S-1) B rows per weight, the four GRU matrices of L = 2 in a single launch.

Same operators (include/g2v.h), same arithmetic per element as the step-at-a-time path (which stays as the module-level
`forward` of BahdanauAttnDecoderRNN for inference and as the cross-check in tests/test_gpu_text2embedding.py); only the
summation order of the weight gradients over the steps differs (one sum over (S-1) B rows instead of S-1 partial sums)."""
from __future__ import annotations

from typing import List, Optional

import torch

from . import ops


class RolloutSpec:
    """Non-tensor arguments of CodeDecoderRollout (one object so that autograd sees a single opaque input)."""

    def __init__(self, cod, steps, n_pre, L, att, dropout_p, mask_emb, mask_l0, bn_running_mean, bn_running_var, defer_bn=None):
        self.cod, self.steps, self.n_pre, self.L, self.att, self.dropout_p = cod, steps, n_pre, L, att, dropout_p
        self.mask_emb, self.mask_l0 = mask_emb, mask_l0
        self.bn_running_mean, self.bn_running_var = bn_running_mean, bn_running_var
        # defer_bn: a list.  The rollout then leaves BatchNorm's running statistics ALONE (the kernels get NULL) and appends
        # (save_mean, save_invstd, step_stride, steps, B, H) of its S-1 applications: the trainer commits them behind the backward,
        # where a persistent-kernel fault of any part of the iteration is known (commit_bn: latch-gated on the device)
        self.defer_bn = defer_bn
        if defer_bn is not None:
            self.bn_running_mean = self.bn_running_var = None


def decoder_params(dec) -> List[torch.Tensor]:
    """Parameter order of CodeDecoderRollout for a text2embedding BahdanauAttnDecoderRNN."""
    lin, bn, g = dec.pre_linear[0], dec.pre_linear[1], dec.gru
    ps = [dec.embedding.weight, lin.weight, lin.bias, bn.weight, bn.bias]
    for l in range(dec.n_layers):
        ps += [getattr(g, f"weight_ih_l{l}"), getattr(g, f"weight_hh_l{l}"), getattr(g, f"bias_ih_l{l}"), getattr(g, f"bias_hh_l{l}")]
    ps += [dec.out.weight, dec.out.bias]
    if dec.att_use:
        ps += [dec.attn.attn.weight, dec.attn.attn.bias, dec.attn.v]
    return ps


# Batch size from which the decode steps run as the FUSED per-step kernels (g2v_attn_code_rollout_fwd / _bwd, csrc/t2e_rollout.hip:
# one 512-thread workgroup per 16 batch rows, everything between two BatchNorm seams in one launch).  Below it a batch has too
# few row tiles to fill the chip and the column-split per-operator kernels (one launch each, every CU streaming a slice of the
# weights) are faster -- the same crossover as the pose decoder's (dec_rollout.hip, "split" kernels).
FUSED_MIN_ROWS = 1024
# Below FUSED_MIN_ROWS, without attention, while the (hidden-unit tile x row group) grid has a CU per workgroup (B <= 304 at
# H = 200): the FORWARD rollout is one persistent cluster launch behind the same C entry (csrc/t2e_rollout.hip:
# code_cluster_fwd_kernel; round 5: ~9 launches per decode step at the reference's B = 128 before); the backward stays the
# per-operator chain below, which reads the arrays that launch saved.  False = the per-operator forward (tests, A/B).
CLUSTER_FORWARD = True
CLUSTER_BACKWARD = True      # behind a cluster forward: the GRU cells' BPTT as one persistent cluster launch too (g2v_code_cluster_bptt)
CLUSTER_CALLS = 0
CLUSTER_BPTT_CALLS = 0
FUSED_CALLS = 0          # forwards served by the fused kernels (tests assert that the path under test actually ran)
LAST_SAVED = None        # the last forward's greedy codes and post-BatchNorm activations, whichever kernels served it (tests read the
                         # discrete decisions the kernels took and hand them to the oracle)


def _cluster_ok(hidden0, enc_out, spec: RolloutSpec, params) -> bool:
    if not CLUSTER_FORWARD or spec.L != 2 or spec.att or hidden0.shape[1] >= FUSED_MIN_ROWS:
        return False
    B, H = hidden0.shape[1], hidden0.shape[2]
    K = params[5 + 4 * spec.L].shape[0]
    return ops.code_rollout_ok(spec.steps, B, H, K, 0, False) and ops.code_rollout_cluster_ok(spec.steps, B, H, K, False)


def _fused_ok(hidden0, enc_out, spec: RolloutSpec, params) -> bool:
    if spec.L != 2 or hidden0.shape[1] < FUSED_MIN_ROWS:
        return False
    B, H = hidden0.shape[1], hidden0.shape[2]
    K = params[5 + 4 * spec.L].shape[0]
    Tw = enc_out.shape[0] if spec.att else 0
    return ops.code_rollout_ok(spec.steps, B, H, K, Tw, spec.att)


def _fused_weights(spec, params, H):
    emb_w, pre_w, pre_b, bn_w, bn_b = params[:5]
    (w_ih0, w_hh0, b_ih0, b_hh0), (w_ih1, w_hh1, b_ih1, b_hh1) = params[5:9], params[9:13]
    out_w, out_b = params[13:15]
    wd = dict(emb=emb_w.contiguous(), w_pre=pre_w.contiguous(), b_pre=pre_b, bn_w=bn_w, bn_b=bn_b,
              bn_running_mean=spec.bn_running_mean, bn_running_var=spec.bn_running_var,
              w_ih0=w_ih0.contiguous(), w_hh0=w_hh0.contiguous(), b_ih0=b_ih0, b_hh0=b_hh0,
              w_ih1=w_ih1.contiguous(), w_hh1=w_hh1.contiguous(), b_ih1=b_ih1, b_hh1=b_hh1,
              w_out=out_w.contiguous(), b_out=out_b)
    if spec.att:
        attn_w, attn_b, attn_v = params[15:18]
        wd.update(w_attn=attn_w.contiguous(), b_attn=attn_b, v_attn=attn_v.contiguous())
    return wd


def _fused_forward(ctx, hidden0, enc_out, spec: RolloutSpec, params, cluster: bool = False):
    """The S1 decode steps as S1 + 1 launches of ONE kernel (include/g2v.h: g2v_attn_code_rollout_fwd); cluster=True: the same entry
    at small batch (one persistent launch), with the context laid out for the PER-OPERATOR backward."""
    global FUSED_CALLS, CLUSTER_CALLS, LAST_SAVED
    if cluster:
        CLUSTER_CALLS += 1
    else:
        FUSED_CALLS += 1
    S1, att = spec.steps, spec.att
    B, H = hidden0.shape[1], hidden0.shape[2]
    K = params[13].shape[0]
    Hin = 2 * H if att else H
    dev = hidden0.device
    f32 = lambda *s: torch.empty(s, dtype=torch.float32, device=dev)
    wd = _fused_weights(spec, params, H)
    drop = spec.dropout_p > 0 and spec.mask_l0 is not None
    mask_emb = spec.mask_emb.contiguous()
    mask_l0 = spec.mask_l0.contiguous() if drop else None
    nblk = (B + 15) // 16
    Tw = enc_out.shape[0] if att else 0
    sv = dict(ids=torch.empty((S1, B), dtype=torch.int64, device=dev), ec=f32(S1, B, Hin), u=f32(S1, B, H), a=f32(S1, B, H),
              bn_stats=f32(S1, 2, H), h0=f32(S1 + 1, B, H), h1=f32(S1 + 1, B, H), x1=f32(S1, B, H) if drop else None,
              gates0=f32(S1, B, 4 * H), gates1=f32(S1, B, 4 * H), bn_partial=f32(2, nblk, 2, H))
    full = _outputs_buffer(spec, S1, B, K, dev)
    sv["logits"] = full[1:]
    enc = ep = None
    if att:
        enc = enc_out.contiguous()
        W_e = wd["w_attn"][:, H:].contiguous()
        ep = ops.linear_fwd(enc.view(Tw * B, H), W_e).view(Tw, B, H)          # step-independent half of the energies
        sv.update(hp=f32(S1, B, H), attw=f32(S1, B, Tw))
    ops.code_rollout_fwd(spec.cod.contiguous(), hidden0.contiguous(), enc, ep, wd, sv, mask_emb, mask_l0,
                         spec.dropout_p if drop else 0.0, spec.n_pre, True, S1, B, H, K, Tw)
    if spec.defer_bn is not None:
        spec.defer_bn.append((sv["bn_stats"], sv["bn_stats"].view(-1)[H:], 2 * H, S1, B, H))
    ctx.save_for_backward(hidden0, enc_out, *params)
    if cluster:
        # what CodeDecoderRollout.backward's per-operator chain reads, as views of the arrays the launch saved
        ctx.spec, ctx.dims, ctx.fused, ctx.cell = spec, (S1, B, H, K, Hin, spec.L), False, ops.gru_cell_ok(H, H, B)
        gru = [(wd["w_ih0"], wd["w_hh0"], wd["b_ih0"], wd["b_hh0"]), (wd["w_ih1"], wd["w_hh1"], wd["b_ih1"], wd["b_hh1"])]
        ctx.bufs = dict(ids=sv["ids"], EC=sv["ec"], U=sv["u"], A=sv["a"], SM=sv["bn_stats"][:, 0], SI=sv["bn_stats"][:, 1],
                        Hs=[sv["h0"], sv["h1"]], GATES=[sv["gates0"], sv["gates1"]], mask_emb=mask_emb, mask_l0=mask_l0,
                        scale_l0=1.0 / (1.0 - spec.dropout_p) if drop else 1.0, emb_w=wd["emb"], pre_w=wd["w_pre"],
                        out_w=wd["w_out"], gru=gru)
        # (and for the cells' BPTT as a cluster launch: the C structs of the same arrays, without this node's output)
        ctx.cluster = dict(wd=wd, sv={k: t for k, t in sv.items() if k != "logits"}, p=spec.dropout_p if drop else 0.0)
        LAST_SAVED = {"ids": sv["ids"], "a": sv["a"]}
        AW = f32(0)
        ctx.mark_non_differentiable(AW)
        ctx.set_materialize_grads(False)
        return full, AW
    ctx.spec, ctx.dims, ctx.fused = spec, (S1, B, H, K, Hin, spec.L), True
    # (the logits are this node's OUTPUT: kept on ctx they would close a reference cycle output -> grad_fn -> ctx -> output, the
    #  step's buffers would live until the garbage collector runs -- inside a hipGraph capture that ended in a segfault of
    #  capture_end; the backward does not read them)
    ctx.bufs = dict(wd=wd, sv={k: t for k, t in sv.items() if k != "logits"}, enc=enc, ep=ep, mask_emb=mask_emb, mask_l0=mask_l0,
                    drop=drop, Tw=Tw)
    LAST_SAVED = {"ids": sv["ids"], "a": sv["a"]}
    AW = sv["attw"] if att else f32(0)
    ctx.mark_non_differentiable(AW)
    ctx.set_materialize_grads(False)
    return full, AW


def _outputs_buffer(spec: RolloutSpec, S1, B, K, dev):
    """The model's `outputs` in step-major order, (S1 + 1, B, K): slot 0 = one_hot(codes[0]) (reference :676-677), slots 1.. =
    the decode steps' logits, written in place by the kernels (it used to be a torch.cat of the two: a 110 MB copy at B = 4096)."""
    full = torch.empty((S1 + 1, B, K), dtype=torch.float32, device=dev)
    ops.one_hot_rows(spec.cod[0], K, full[0])
    return full


def _fused_backward(ctx, dLOG):
    S1, B, H, K, Hin, L = ctx.dims
    b, spec = ctx.bufs, ctx.spec
    att = spec.att
    dev = dLOG.device
    f32 = lambda *s: torch.empty(s, dtype=torch.float32, device=dev)
    G = 3 * H
    gr = dict(d_hidden0=f32(2, B, H), d_emb=f32(K, H), d_w_pre=f32(H, Hin), d_b_pre=f32(H), d_bn_w=f32(H), d_bn_b=f32(H),
              d_w_ih0=f32(G, H), d_w_hh0=f32(G, H), d_b_ih0=f32(G), d_b_hh0=f32(G),
              d_w_ih1=f32(G, H), d_w_hh1=f32(G, H), d_b_ih1=f32(G), d_b_hh1=f32(G), d_w_out=f32(K, H), d_b_out=f32(K))
    if att:
        gr.update(d_w_attn=f32(H, 2 * H), d_b_attn=f32(H), d_v_attn=f32(H), d_enc=f32(b["Tw"], B, H))
    ops.code_rollout_bwd(dLOG, b["enc"], b["ep"], b["wd"], b["sv"], gr, b["mask_emb"], b["mask_l0"],
                         spec.dropout_p if b["drop"] else 0.0, S1, B, H, K, b["Tw"])
    grads = [gr["d_emb"], gr["d_w_pre"], gr["d_b_pre"], gr["d_bn_w"], gr["d_bn_b"],
             gr["d_w_ih0"], gr["d_w_hh0"], gr["d_b_ih0"], gr["d_b_hh0"], gr["d_w_ih1"], gr["d_w_hh1"], gr["d_b_ih1"], gr["d_b_hh1"],
             gr["d_w_out"], gr["d_b_out"]]
    if att:
        grads += [gr["d_w_attn"], gr["d_b_attn"], gr["d_v_attn"]]
    return (gr["d_hidden0"], gr["d_enc"] if att else None, None, *grads)


class CodeDecoderRollout(torch.autograd.Function):
    """(hidden0 (L,B,H), encoder_outputs (Tw,B,H) or None, spec, *decoder_params) -> outputs (S,B,K) [slot 0 = one_hot(codes[0]),
    slots 1.. = the S-1 decode steps' logits], attention weights (S-1,B,Tw) (empty without attention; not differentiable)."""

    @staticmethod
    def forward(ctx, hidden0, enc_out, spec: RolloutSpec, *params):
        global LAST_SAVED
        L, att, S1 = spec.L, spec.att, spec.steps
        if _fused_ok(hidden0, enc_out, spec, params):
            return _fused_forward(ctx, hidden0, enc_out, spec, params)
        if _cluster_ok(hidden0, enc_out, spec, params):
            return _fused_forward(ctx, hidden0, enc_out, spec, params, cluster=True)
        ctx.fused = False
        emb_w, pre_w, pre_b, bn_w, bn_b = params[:5]
        gru = [params[5 + 4 * l: 9 + 4 * l] for l in range(L)]            # (w_ih, w_hh, b_ih, b_hh) per layer
        out_w, out_b = params[5 + 4 * L: 7 + 4 * L]
        B, H, K = hidden0.shape[1], hidden0.shape[2], out_w.shape[0]
        Hin = 2 * H if att else H
        dev = hidden0.device
        f32 = lambda *s: torch.empty(s, dtype=torch.float32, device=dev)
        emb_w, pre_w, out_w = emb_w.contiguous(), pre_w.contiguous(), out_w.contiguous()
        gru = [(w_ih.contiguous(), w_hh.contiguous(), b_ih, b_hh) for (w_ih, w_hh, b_ih, b_hh) in gru]
        drop = spec.dropout_p > 0 and spec.mask_l0 is not None
        scale_l0 = 1.0 / (1.0 - spec.dropout_p) if drop else 1.0
        mask_emb = spec.mask_emb.contiguous()
        mask_l0 = spec.mask_l0.contiguous() if drop else None

        ids = torch.empty((S1, B), dtype=torch.int64, device=dev)
        npre = max(1, min(spec.n_pre, S1))
        ids[:npre].copy_(spec.cod[:npre])                              # teacher-forced positions (:740)
        EC, U, A = f32(S1, B, Hin), f32(S1, B, H), f32(S1, B, H)
        SM, SI = f32(S1, H), f32(S1, H)
        Hs = [f32(S1 + 1, B, H) for _ in range(L)]
        GATES = [f32(S1, B, 4 * H) for _ in range(L)]
        GI, HN = f32(B, 3 * H), f32(B, H)
        cell = ops.gru_cell_ok(H, H, B)
        LOGF = _outputs_buffer(spec, S1, B, K, dev)
        LOG = LOGF[1:]
        for l in range(L):
            Hs[l][0].copy_(hidden0[l])
        if att:
            attn_w, attn_b, attn_v = params[7 + 4 * L: 10 + 4 * L]
            Tw = enc_out.shape[0]
            enc = enc_out.contiguous()
            W_h, W_e = attn_w[:, :H].contiguous(), attn_w[:, H:].contiguous()
            attn_v = attn_v.contiguous()
            EP = ops.linear_fwd(enc.view(Tw * B, H), W_e).view(Tw, B, H)          # step-independent half of the energies
            HP, AW = f32(S1, B, H), f32(S1, B, Tw)
        else:
            AW = f32(0)
        # With attention a step's row-local head -- the greedy argmax of the previous logits, the code embedding, the attention
        # -- is one launch, and the top state is multiplied once for the logits and the NEXT step's query (round 6: at the
        # reference's batch size a step is a chain of dependent ~5 us launches, 9 -> 7 per step)
        fused_head = att and B <= 1024
        if fused_head:
            ops.linear_fwd(Hs[L - 1][0], W_h, attn_b, out=HP[0])
        for t in range(S1):
            if fused_head:
                greedy = t >= npre
                ops.attn_step_fwd(LOG[t - 1] if greedy else None, ids[t], emb_w, mask_emb[t], 2.0, EC[t], HP[t], EP, enc, attn_v, AW[t])
            else:
                ops.embedding_fwd(emb_w, ids[t], mask_emb[t], 2.0, out=EC[t], ldo=Hin)            # Embedding + Dropout(0.5)
                if att:
                    ops.linear_fwd(Hs[L - 1][t], W_h, attn_b, out=HP[t])
                    ops.attn_fwd(HP[t], EP, enc, attn_v, ctx_out=EC[t][:, H:], ldctx=Hin, weights=AW[t])
            ops.linear_fwd(EC[t], pre_w, pre_b, out=U[t])
            ops.batchnorm_fwd(U[t], bn_w, bn_b, spec.bn_running_mean, spec.bn_running_var, True, True, out=A[t],
                              save=(SM[t], SI[t]))
            layer_in, keep, scale = A[t], None, 1.0
            for l, (w_ih, w_hh, b_ih, b_hh) in enumerate(gru):
                if cell:        # small batch: input projection + cell in one launch
                    ops.gru_cell_fwd(layer_in, Hs[l][t], w_ih, w_hh, b_ih, b_hh, keep=keep, scale=scale, h_new=Hs[l][t + 1],
                                     gates=GATES[l][t])
                else:
                    ops.linear_fwd(layer_in, w_ih, b_ih, keep=keep, scale=scale, out=GI)
                    ops.gru_dirs_fwd([dict(gi=GI, w_hh=w_hh, b_hh=b_hh, h0=Hs[l][t], hs=Hs[l][t + 1], h_n=HN, gates=GATES[l][t],
                                           reverse=False)], 1, B, H)
                layer_in = Hs[l][t + 1]
                if drop:
                    keep, scale = mask_l0[t], scale_l0                   # nn.GRU inter-layer dropout, fused into the next Linear
            if fused_head and t + 1 < S1:
                ops.linear_fwd_dual(Hs[L - 1][t + 1], out_w, out_b, LOG[t], W_h, attn_b, HP[t + 1])
            else:
                ops.linear_fwd(Hs[L - 1][t + 1], out_w, out_b, out=LOG[t])
            if not fused_head and t + 1 < S1 and t + 1 >= npre:
                ops.argmax_rows(LOG[t], out=ids[t + 1])                  # greedy feedback (:740)
        ctx.save_for_backward(hidden0, enc_out, *params)
        ctx.spec, ctx.dims, ctx.cell = spec, (S1, B, H, K, Hin, L), cell
        ctx.bufs = dict(ids=ids, EC=EC, U=U, A=A, SM=SM, SI=SI, Hs=Hs, GATES=GATES, mask_emb=mask_emb, mask_l0=mask_l0,
                        scale_l0=scale_l0, emb_w=emb_w, pre_w=pre_w, out_w=out_w, gru=gru)
        if att:
            ctx.bufs.update(enc=enc, W_h=W_h, W_e=W_e, attn_v=attn_v, EP=EP, HP=HP, AW=AW)
        LAST_SAVED = {"ids": ids, "a": A}
        if spec.defer_bn is not None:
            spec.defer_bn.append((SM, SI, H, S1, B, H))
        ctx.mark_non_differentiable(AW)
        ctx.set_materialize_grads(False)
        return LOGF, AW

    @staticmethod
    def backward(ctx, dFULL, _dAW):
        S1, B, H, K, Hin, L = ctx.dims
        n_in = 3 + 5 + 4 * L + 2 + (3 if ctx.spec.att else 0)
        if dFULL is None:
            return (None,) * n_in
        dLOG = dFULL.contiguous()[1:]                                    # slot 0 is a constant
        if ctx.fused:
            return _fused_backward(ctx, dLOG)
        b, att = ctx.bufs, ctx.spec.att
        bn_w = ctx.saved_tensors[2 + 3]
        dev = dLOG.device
        f32 = lambda *s: torch.empty(s, dtype=torch.float32, device=dev)
        Hs, GATES, gru, drop = b["Hs"], b["GATES"], b["gru"], b["mask_l0"] is not None
        M = S1 * B
        dLOG = dLOG.contiguous()
        DH_top = ops.linear_bwd_data(dLOG.view(M, K), b["out_w"]).view(S1, B, H)        # every step's dLogits W_out at once
        cl = getattr(ctx, "cluster", None)
        cluster_bptt = cl is not None and CLUSTER_BACKWARD and not att
        if not cluster_bptt:
            carry = [torch.zeros((B, H), dtype=torch.float32, device=dev) for _ in range(L)]   # dLoss / d(previous state), per layer
            carry_next = [f32(B, H) for _ in range(L)]
        DGI = [f32(S1, B, 3 * H) for _ in range(L)]
        DGH = [f32(S1, B, 3 * H) for _ in range(L)]
        if not cluster_bptt:
            DU, DBW, DBB = f32(S1, B, H), f32(S1, H), f32(S1, H)
        DXs = [f32(B, H) for _ in range(L)]                             # gradient w.r.t. each layer's input
        if att:
            DEC, DHP = f32(S1, B, Hin), f32(S1, B, H)
            D_EP, D_ENC, D_V = torch.empty_like(b["EP"]), torch.empty_like(b["enc"]), f32(H)
            DV_SLABS = f32(S1, B, H) if B <= 1024 else None      # d_v's per-row partials of every step: ONE reduction behind the loop
        if cluster_bptt:
            # small batch (round 5): the cells' BPTT as ONE persistent cluster launch (include/g2v.h: g2v_code_cluster_bptt); BatchNorm's
            # backward per step behind it (nothing there feeds the recurrence: the greedy feedback carries no gradient)
            global CLUSTER_BPTT_CALLS
            CLUSTER_BPTT_CALLS += 1
            DGI[0], DGH[0], DGI[1], DGH[1], DA, d_h0 = ops.code_cluster_bptt(DH_top, cl["wd"], cl["sv"], b["mask_l0"], cl["p"], S1, B, H)
            d_hidden0 = d_h0          # (BatchNorm's backward feeds parameters and the code embedding only: the side branch below)
        for t in (reversed(range(S1)) if not cluster_bptt else ()):
            d_in = DH_top[t]                                             # gradient arriving at Hs[l][t+1] from above
            for l in reversed(range(L)):
                w_ih, w_hh = gru[l][0], gru[l][1]
                if ctx.cell:    # gate gradients, then d_hprev += dgh W_hh and dx = (dgi W_ih) * mask in one launch
                    ops.gru_cell_bwd(d_in, carry[l], GATES[l][t], Hs[l][t], w_ih, w_hh,
                                     keep=b["mask_l0"][t] if (drop and l > 0) else None, scale=b["scale_l0"] if l > 0 else 1.0,
                                     dgi=DGI[l][t], dgh=DGH[l][t], d_hprev=carry_next[l], dx=DXs[l])
                else:
                    ops.gru_dirs_bwd([dict(d_hs=d_in, d_hn=carry[l], hs=Hs[l][t + 1], h0=Hs[l][t], gates=GATES[l][t], w_hh=w_hh,
                                           dgi=DGI[l][t], dgh=DGH[l][t], dh0=carry_next[l], reverse=False)], 1, B, H)
                    ops.linear_bwd_data(DGI[l][t], w_ih, out=DXs[l])
                    if l > 0 and drop:
                        ops.mask_mul(DXs[l], b["mask_l0"][t], b["scale_l0"], out=DXs[l])
                if l > 0:
                    d_in = DXs[l]
            ops.batchnorm_bwd(DXs[0], b["U"][t], b["A"][t], bn_w, b["SM"][t], b["SI"][t], True, out=(DU[t], DBW[t], DBB[t]))
            if att:
                ops.linear_bwd_data(DU[t], b["pre_w"], out=DEC[t])
                ops.attn_bwd(DEC[t][:, H:], b["HP"][t], b["EP"], b["enc"], b["attn_v"], b["AW"][t], ldd=Hin,
                             out=(DHP[t], D_EP, D_ENC, D_V), accumulate=(t != S1 - 1),
                             dv_slab=DV_SLABS[t] if DV_SLABS is not None else None)
                ops.linear_bwd_data(DHP[t], b["W_h"], out=carry_next[L - 1], accumulate=True)     # the state the attention scored
            carry, carry_next = carry_next, carry
        # ---- everything that did not feed the recurrence: one launch over all (S-1) B rows each ---------------------------
        d_enc = None
        if att:      # (the encoder's backward waits for this one: it stays on the chain)
            if DV_SLABS is not None:
                ops.slab_sum(DV_SLABS.view(S1 * B, H), D_V)
            Tw = b["enc"].shape[0]
            ops.linear_bwd_data(D_EP.view(Tw * B, H), b["W_e"], out=D_ENC.view(Tw * B, H), accumulate=True)
            d_enc = D_ENC
        if not cluster_bptt:
            d_hidden0 = torch.stack(carry)
        # (Nothing downstream waits for the parameters' gradients, but as a side branch of the iteration beside the encoder's
        #  backward they never paid: ops.side_branch.)
        if cluster_bptt:
            if B < 1024:     # every step's BatchNorm backward + the sums over the steps: one launch
                DU, d_bn_w, d_bn_b = ops.batchnorm_bwd_steps(DA, b["U"], b["A"], bn_w, b["SM"], b["SI"], True)
            else:
                DU, DBW, DBB = f32(S1, B, H), f32(S1, H), f32(S1, H)
                for t in range(S1):
                    ops.batchnorm_bwd(DA[t], b["U"][t], b["A"][t], bn_w, b["SM"][t], b["SI"][t], True, out=(DU[t], DBW[t], DBB[t]))
                d_bn_w, d_bn_b = DBW.sum(0), DBB.sum(0)
        else:
            d_bn_w, d_bn_b = DBW.sum(0), DBB.sum(0)
        if att:
            d_e = DEC[:, :, :H].contiguous().view(M, H)
        else:
            d_e = ops.linear_bwd_data(DU.view(M, H), b["pre_w"])
        V = b["emb_w"].shape[0]
        d_emb = ops.embedding_bwd(d_e, b["ids"].view(-1), V, b["mask_emb"].view(M, H), 2.0)
        d_out_w, d_out_b = ops.linear_bwd_weight(dLOG.view(M, K), Hs[L - 1][1:].view(M, H), K, H)
        d_pre_w, d_pre_b = ops.linear_bwd_weight(DU.view(M, H), b["EC"].view(M, Hin), H, Hin)
        items, g_gru = [], []
        for l in range(L):
            if l == 0:
                x = b["A"].view(M, H)
            else:
                x = Hs[l - 1][1:].view(M, H)
                if drop:
                    x = ops.mask_mul(x, b["mask_l0"].view(M, H), b["scale_l0"])
            dw_ih, dw_hh, db_ih, db_hh = f32(3 * H, H), f32(3 * H, H), f32(3 * H), f32(3 * H)
            items += [(DGI[l].view(M, 3 * H), x, dw_ih, db_ih), (DGH[l].view(M, 3 * H), Hs[l][:-1].view(M, H), dw_hh, db_hh)]
            g_gru += [dw_ih, dw_hh, db_ih, db_hh]
        for k in range(0, len(items), 4):
            ops.linear_bwd_weight_batch(items[k:k + 4], 3 * H, H, M=M)
        grads = [d_emb, d_pre_w, d_pre_b, d_bn_w, d_bn_b] + g_gru + [d_out_w, d_out_b]
        if att:
            dW_h, d_attn_b = ops.linear_bwd_weight(DHP.view(M, H), Hs[L - 1][:-1].view(M, H), H, H)
            dW_e, _ = ops.linear_bwd_weight(D_EP.view(Tw * B, H), b["enc"].view(Tw * B, H), H, H, want_bias=False)
            grads += [torch.cat([dW_h, dW_e], 1), d_attn_b, D_V]
        return (d_hidden0, d_enc, None, *grads)
