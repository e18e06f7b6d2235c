"""CPU oracle for the Gesture2Vec hot path  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
module.  The product path (gesture2vec_amd/) never does, and fails loudly when the HIP
extension is missing.

This is a from-scratch functional restatement (torch CPU fp32 tensors + explicit
formulas, no reference classes, no nn.Module) of what the reference computes on the
path BASELINE.json names.  Citations are `file:line` into /root/reference/scripts/.
Parity status: PINNED -- tests/test_oracle_golden.py checks every function here
against golden vectors captured by importing the reference in the build container
(tests/golden/make_fixtures.py), because the reference itself ships no tests
(SURVEY.md §4).

Conventions
  sd        dict[str, Tensor] with the reference's state_dict keys
            (e.g. "encoder.gru.weight_ih_l0_reverse", "decoder.decoder.pre_linear.1.running_mean",
            "vq_layer._ema_w")
  masks     explicit dropout KEEP masks (uint8/bool/float 0-1); a dropped element is 0, a kept
            one is scaled by 1/(1-p) exactly as ATen's dropout does
  shapes    B batch, T frames, D pose dim, H hidden, L layers (2), E = H*L, K codes
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch

Tensor = torch.Tensor


# ----------------------------------------------------------------------------------------------
# elementary pieces
# ----------------------------------------------------------------------------------------------
def linear(x: Tensor, w: Tensor, b: Optional[Tensor]) -> Tensor:
    """nn.Linear: y = x W^T + b."""
    y = x @ w.t()
    return y if b is None else y + b


def dropout_apply(x: Tensor, keep: Optional[Tensor], p: float) -> Tensor:
    """ATen dropout with an explicit keep mask: x * keep / (1 - p)."""
    if keep is None or p == 0.0:
        return x
    return x * keep.to(x.dtype) / (1.0 - p)


def gru_cell(gi: Tensor, h: Tensor, w_hh: Tensor, b_hh: Tensor) -> Tensor:
    """One PyTorch GRU cell given the input projection gi = x W_ih^T + b_ih  (B,3H).

    r = sigmoid(gi_r + gh_r); z = sigmoid(gi_z + gh_z); n = tanh(gi_n + r * gh_n);
    h' = (1 - z) * n + z * h        (torch.nn.GRU docs; gate order r,z,n)
    Used by Autoencoder_VQVAE_model.py:94 (encoder nn.GRU) and :584 (decoder nn.GRU).
    """
    H = h.shape[-1]
    gh = h @ w_hh.t() + b_hh
    r = torch.sigmoid(gi[..., :H] + gh[..., :H])
    z = torch.sigmoid(gi[..., H:2 * H] + gh[..., H:2 * H])
    n = torch.tanh(gi[..., 2 * H:] + r * gh[..., 2 * H:])
    return (1.0 - z) * n + z * h


def gru_direction(x: Tensor, w_ih: Tensor, w_hh: Tensor, b_ih: Tensor, b_hh: Tensor,
                  reverse: bool, lengths: Optional[Tensor] = None) -> Tuple[Tensor, Tensor]:
    """A full-sequence single-direction GRU layer, h0 = 0.  x (T,B,I) -> out (T,B,H), h_n (B,H).

    With `lengths` (B,) it reproduces pack_padded_sequence semantics
    (text2embedding_model.py:126-133): rows stop updating after their own last valid step,
    padded output positions are zero, reverse direction starts at each row's own last step.
    """
    T, B, _ = x.shape
    H = w_hh.shape[1]
    gi_all = linear(x, w_ih, b_ih)
    h = x.new_zeros(B, H)
    outs: List[Optional[Tensor]] = [None] * T
    steps = range(T - 1, -1, -1) if reverse else range(T)
    for t in steps:
        hn = gru_cell(gi_all[t], h, w_hh, b_hh)
        if lengths is not None:
            valid = (lengths > t).to(x.dtype).unsqueeze(1)
            h = valid * hn + (1.0 - valid) * h
            outs[t] = valid * hn
        else:
            h = hn
            outs[t] = hn
    return torch.stack(outs), h


def bigru(x: Tensor, sd: Dict[str, Tensor], prefix: str, n_layers: int, p: float,
          inter_masks: Optional[List[Tensor]] = None, lengths: Optional[Tensor] = None
          ) -> Tuple[Tensor, Tensor]:
    """nn.GRU(I, H, n_layers, dropout=p, bidirectional=True), h0 = 0.

    Returns (out (T,B,2H) of the last layer, h_n (2*n_layers,B,H) ordered l0f,l0b,l1f,l1b).
    inter_masks[l] is the keep mask (T,B,2H) ATen draws on layer l's output (l < n_layers-1).
    """
    hs = []
    inp = x
    for l in range(n_layers):
        of, hf = gru_direction(inp, sd[f"{prefix}weight_ih_l{l}"], sd[f"{prefix}weight_hh_l{l}"],
                               sd[f"{prefix}bias_ih_l{l}"], sd[f"{prefix}bias_hh_l{l}"], False, lengths)
        ob, hb = gru_direction(inp, sd[f"{prefix}weight_ih_l{l}_reverse"], sd[f"{prefix}weight_hh_l{l}_reverse"],
                               sd[f"{prefix}bias_ih_l{l}_reverse"], sd[f"{prefix}bias_hh_l{l}_reverse"], True, lengths)
        out = torch.cat([of, ob], dim=2)
        hs += [hf, hb]
        if l < n_layers - 1 and p > 0.0 and inter_masks is not None:
            out = dropout_apply(out, inter_masks[l], p)
        inp = out
    return inp, torch.stack(hs)


def batchnorm1d(x: Tensor, weight: Tensor, bias: Tensor, running_mean: Tensor, running_var: Tensor,
                training: bool, momentum: float = 0.1, eps: float = 1e-5
                ) -> Tuple[Tensor, Tensor, Tensor]:
    """nn.BatchNorm1d on (B,H).  Returns (y, new_running_mean, new_running_var).

    train: normalise with the batch mean and BIASED batch variance; running stats move by
    `momentum` towards the batch mean / UNBIASED variance.  eval: use running stats.
    (Autoencoder_VQVAE_model.py:478, invoked once per decode step :572.)
    """
    if training:
        B = x.shape[0]
        mean = x.mean(0)
        var_b = ((x - mean) ** 2).mean(0)
        y = (x - mean) / torch.sqrt(var_b + eps) * weight + bias
        var_u = var_b * (B / max(B - 1, 1))
        nrm = (1 - momentum) * running_mean + momentum * mean.detach()
        nrv = (1 - momentum) * running_var + momentum * var_u.detach()
        return y, nrm, nrv
    y = (x - running_mean) / torch.sqrt(running_var + eps) * weight + bias
    return y, running_mean, running_var


# ----------------------------------------------------------------------------------------------
# quantizers
# ----------------------------------------------------------------------------------------------
def vq_distances(flat: Tensor, codebook: Tensor) -> Tensor:
    """||x||^2 + ||W||^2 - 2 x W^T, in exactly this algebraic form (Autoencoder_VQVAE_model.py:1234-1238)."""
    return (flat ** 2).sum(1, keepdim=True) + (codebook ** 2).sum(1) - 2 * flat @ codebook.t()


def vq_ema_forward(inputs: Tensor, sd: Dict[str, Tensor], prefix: str, commitment_cost: float,
                   training: bool, decay: float = 0.85, eps: float = 1e-5, use_pre_linear: bool = True,
                   forced_idx: Optional[Tensor] = None) -> Dict[str, Tensor]:
    """VQ_Payam_EMA.forward (Autoencoder_VQVAE_model.py:1217-1296).

    * rows are the contiguous reinterpretation `inputs.view(-1, E)` (:1229)
    * distances/argmin use pre_linear(z) (:1230,1234-1244); loss and straight-through use RAW z (:1285,1292)
    * `quantized` uses the PRE-update codebook (:1256); the EMA update happens afterwards (:1262-1282)
    Returns dict(loss, quantized, perplexity, idx, flat, new _ema_cluster_size/_ema_w/_embedding.weight).
    The (N,K) one-hot `encodings` the reference returns is onehot(idx).
    """
    W = sd[prefix + "_embedding.weight"]
    K, E = W.shape
    z_flat = inputs.reshape(-1, E)
    flat = linear(z_flat, sd[prefix + "pre_linear.weight"], sd[prefix + "pre_linear.bias"]) if use_pre_linear else z_flat
    flat_d = flat.detach()
    d = vq_distances(flat_d, W.detach())
    idx = torch.argmin(d, dim=1) if forced_idx is None else forced_idx      # (forced_idx: see vqvae_forward, cfg["forced"])
    q = W.detach()[idx].reshape(inputs.shape)                       # == onehot @ W bit-exactly (:1256)
    out = {"idx": idx, "flat": flat_d, "dist": d}
    N = z_flat.shape[0]
    cnt = torch.bincount(idx, minlength=K).to(W.dtype)
    if training:
        cs = sd[prefix + "_ema_cluster_size"] * decay + (1 - decay) * cnt              # :1263-1265
        n = cs.sum()
        cs = (cs + eps) / (n + K * eps) * n                                              # :1268-1273
        onehot = torch.zeros(N, K, dtype=W.dtype)
        onehot[torch.arange(N), idx] = 1.0
        dw = onehot.t() @ flat_d                                                         # :1275
        ema_w = sd[prefix + "_ema_w"].detach() * decay + (1 - decay) * dw                # :1276-1278
        out["_ema_cluster_size"] = cs
        out["_ema_w"] = ema_w
        out["_embedding.weight"] = ema_w / cs.unsqueeze(1)                               # :1280-1282
    e_latent = ((q - inputs) ** 2).mean()                                                # :1285
    out["loss"] = commitment_cost * e_latent                                             # :1289
    out["quantized"] = inputs + (q - inputs).detach()                                    # :1292
    avg = cnt / N
    out["perplexity"] = torch.exp(-(avg * torch.log(avg + 1e-10)).sum())                 # :1293-1294
    return out


def vq_plain_forward(inputs: Tensor, codebook: Tensor, commitment_cost: float) -> Dict[str, Tensor]:
    """VQ_Payam.forward (Autoencoder_VQVAE_model.py:1114-1173): no pre_linear, codebook learns by gradient,
    loss = q_latent + beta * e_latent."""
    K, E = codebook.shape
    flat = inputs.reshape(-1, E)
    d = vq_distances(flat.detach(), codebook.detach())
    idx = torch.argmin(d, 1)
    q = codebook[idx].reshape(inputs.shape)
    e_latent = ((q.detach() - inputs) ** 2).mean()
    q_latent = ((q - inputs.detach()) ** 2).mean()
    cnt = torch.bincount(idx, minlength=K).to(codebook.dtype)
    avg = cnt / flat.shape[0]
    return {"idx": idx, "loss": q_latent + commitment_cost * e_latent,
            "quantized": inputs + (q - inputs).detach(),
            "perplexity": torch.exp(-(avg * torch.log(avg + 1e-10)).sum())}


# ----------------------------------------------------------------------------------------------
# Part-b chunk VQ-VAE
# ----------------------------------------------------------------------------------------------
def vq_gssoft_forward(inputs: Tensor, sd: Dict[str, Tensor], prefix: str, commitment_cost: float) -> Dict[str, Tensor]:
    """VQ_Payam_GSSoft.forward (Autoencoder_VQVAE_model.py:1377-1433), the soft quantiser the reference's
    Autoencoder_VQVAE ships with (:816-820).  flat = mean_layer(x) (pre_linear is unused, :1389); distances to the
    codebook; smooth = 1/exp(logvar_layer(flat))^2; prob = exp(-(d/400) * 0.5 * smooth)/sqrt(smooth), row-normalised
    (:1349-1372); q = probs @ W; loss = mse(q, x.detach()) + beta * mse(q.detach(), x); straight-through output."""
    E = sd[prefix + "_embedding.weight"].shape[1]
    W = sd[prefix + "_embedding.weight"]
    flat = linear(inputs.reshape(-1, E), sd[prefix + "mean_layer.weight"], sd[prefix + "mean_layer.bias"])   # :1391
    z_logvar = linear(flat, sd[prefix + "logvar_layer.weight"], sd[prefix + "logvar_layer.bias"])            # :1392
    d = vq_distances(flat, W)                                                                                # :1396-1400
    smooth = 1.0 / torch.exp(z_logvar) ** 2                                                                  # :1411
    prob = torch.exp(-((d / 400) * (0.5 * smooth))) / torch.sqrt(smooth)                                     # :1351,1361
    probs = prob / prob.sum(1, keepdim=True)                                                                 # :1368
    q = (probs @ W).reshape(inputs.shape)                                                                    # :1417-1421
    e_latent = ((q.detach() - inputs) ** 2).mean()                                                           # :1424
    q_latent = ((q - inputs.detach()) ** 2).mean()                                                           # :1425
    loss = q_latent + commitment_cost * e_latent                                                             # :1427
    quantized = inputs + (q - inputs).detach()                                                               # :1431
    avg = probs.mean(0)
    perplexity = torch.exp(-(avg * torch.log(avg + 1e-10)).sum())                                            # :1432-1433
    return {"loss": loss, "quantized": quantized, "perplexity": perplexity, "probs": probs, "flat": flat, "dist": d}


def encoder_forward(x_tbd: Tensor, sd: Dict[str, Tensor], n_layers: int, p: float,
                    inter_masks: Optional[List[Tensor]] = None) -> Tuple[Tensor, Tensor]:
    """EncoderRNN.forward (Autoencoder_VQVAE_model.py:73-100): Linear(D->H) -> bi-GRU -> sum directions."""
    H = sd["encoder.in_layer.weight"].shape[0]
    xin = linear(x_tbd, sd["encoder.in_layer.weight"], sd["encoder.in_layer.bias"])      # :93
    out, hidden = bigru(xin, sd, "encoder.gru.", n_layers, p, inter_masks)               # :94
    return out[:, :, :H] + out[:, :, H:], hidden                                         # :95-97


def decoder_step(y_prev: Tensor, hidden: Tensor, sd: Dict[str, Tensor], n_layers: int, training: bool,
                 keep95: Tensor, p: float, inter_mask: Optional[Tensor], bn_state: Dict[str, Tensor],
                 conditioned: bool = True, relu_mask: Optional[Tensor] = None) -> Tuple[Tensor, Tensor]:
    """Generator.forward -> BahdanauAttnDecoderRNN.forward, no attention (Autoencoder_VQVAE_model.py:499-592).

    Dropout(0.95) is constructed inline (:570) so it is ACTIVE IN EVAL TOO; BN batch stats in train (:572).
    bn_state holds running_mean / running_var / num_batches_tracked and is updated in place.
    """
    pre = "decoder.decoder."
    inp = y_prev if conditioned else torch.zeros_like(y_prev)                            # :568-569
    u = dropout_apply(inp, keep95, 0.95)                                                 # :570
    u = linear(u, sd[pre + "pre_linear.0.weight"], sd[pre + "pre_linear.0.bias"])
    a, nrm, nrv = batchnorm1d(u, sd[pre + "pre_linear.1.weight"], sd[pre + "pre_linear.1.bias"],
                              bn_state["running_mean"], bn_state["running_var"], training)
    if training:
        bn_state["running_mean"], bn_state["running_var"] = nrm, nrv
        bn_state["num_batches_tracked"] = bn_state["num_batches_tracked"] + 1
    a = torch.relu(a) if relu_mask is None else a * relu_mask      # (relu_mask: see vqvae_forward, cfg["forced"])
    new_h = []
    layer_in = a
    for l in range(n_layers):                                                            # :584 nn.GRU(H,H,L), seq len 1
        gi = linear(layer_in, sd[pre + f"gru.weight_ih_l{l}"], sd[pre + f"gru.bias_ih_l{l}"])
        h = gru_cell(gi, hidden[l], sd[pre + f"gru.weight_hh_l{l}"], sd[pre + f"gru.bias_hh_l{l}"])
        new_h.append(h)
        layer_in = h
        if l < n_layers - 1 and training and p > 0.0 and inter_mask is not None:
            layer_in = dropout_apply(h, inter_mask, p)
    y = linear(new_h[-1], sd[pre + "out_layer.weight"], sd[pre + "out_layer.bias"])      # :590
    return y, torch.stack(new_h)


def vqvae_forward(sd: Dict[str, Tensor], in_poses: Tensor, out_poses: Tensor, cfg: dict, training: bool,
                  masks: dict) -> Dict[str, Tensor]:
    """Autoencoder_VQVAE.forward with vq=True, vae=False, CNN=False, att=False
    (Autoencoder_VQVAE_model.py:901-1072).

    cfg: n_layers, dropout_prob, commitment_cost, n_pre_poses, conditioned (bool)
    masks: 'in' (T,B,D) [if p>0 and training], 'enc_l0' (T,B,2H) [same], 'dec' (T-1,B,D) ALWAYS,
           'dec_l0' (T-1,B,H) [if p>0 and training]
    Returns outputs (B,T,D), first_hidden (L,B,H), loss_vq, perplexity, idx, encoder_hidden and the
    updated buffers (EMA state, BN running stats).
    """
    L, p = cfg["n_layers"], cfg["dropout_prob"]
    x = in_poses.transpose(0, 1)                                                         # :956
    if training and p > 0.0:
        x = dropout_apply(x, masks["in"], p)                                             # :957
    tgt = out_poses.transpose(0, 1)                                                      # :958
    T, B, D = tgt.shape
    enc_out, enc_hidden = encoder_forward(x, sd, L, p if training else 0.0,
                                          [masks["enc_l0"]] if (training and p > 0.0) else None)  # :966
    dec_hidden = enc_hidden[:L].contiguous()                                             # :971-973 (layer-0 fwd/bwd!)
    # cfg["forced"] (tests of the LARGE-batch parity only; absent everywhere else): the DISCRETE decisions of the run under test --
    # "idx" (N,) code indices, "relu" (T-1,B,H) 0/1 mask of the decoder's ReLU, "sign_l1" (B,T,D) / "sign_cont" (B,T-1,D) signs
    # of custom_loss's |.| terms.  A float64 oracle and an fp32 kernel legitimately disagree on a decision whose argument is
    # inside fp32 rounding of its threshold, and one flipped ReLU / sign moves single gradient elements by far more than
    # rounding; with the decisions pinned the two sides compute the same smooth function and differ by rounding only.
    forced = cfg.get("forced") or {}
    vq = vq_ema_forward(dec_hidden, sd, "vq_layer.", cfg["commitment_cost"], training,   # :977
                        forced_idx=forced.get("idx"))
    hidden = vq["quantized"]                                                             # :978
    first_hidden = hidden
    bn = {"running_mean": sd["decoder.decoder.pre_linear.1.running_mean"],
          "running_var": sd["decoder.decoder.pre_linear.1.running_var"],
          "num_batches_tracked": sd["decoder.decoder.pre_linear.1.num_batches_tracked"]}
    outs = [tgt[0]]                                                                      # :1039-1040
    dec_in = tgt[0]
    for t in range(1, T):                                                                # :1041-1054
        il = masks["dec_l0"][t - 1] if (training and p > 0.0) else None
        y, hidden = decoder_step(dec_in, hidden, sd, L, training, masks["dec"][t - 1], p, il, bn,
                                 cfg.get("conditioned", True),
                                 relu_mask=forced["relu"][t - 1] if "relu" in forced else None)
        outs.append(y)
        dec_in = tgt[t] if t < cfg["n_pre_poses"] else y                                 # :1049-1052
    outputs = torch.stack(outs).transpose(0, 1)                                          # :1066
    return {"outputs": outputs, "first_hidden": first_hidden, "loss_vq": vq["loss"],
            "perplexity": vq["perplexity"], "idx": vq["idx"], "encoder_hidden": enc_hidden,
            "flat": vq["flat"], "dist": vq["dist"], "vq": vq, "bn": bn}


def custom_loss(output: Tensor, target: Tensor, w_l1: float, w_cont: float, w_var: float,
                forced: Optional[dict] = None) -> Tensor:
    """train_eval/train_seq2seq.py:40-88.  output/target (B,T,D).  forced: see vqvae_forward (|x| as x * sign with the signs
    of the run under test: the same value wherever the sign is the true one, and the same gradient everywhere)."""
    n = output.numel()
    if forced and "sign_l1" in forced:
        l1 = ((output - target) * forced["sign_l1"]).mean() * w_l1
        cont = ((output[:, 1:, :] - output[:, :-1, :]) * forced["sign_cont"]).sum() / n * w_cont
    else:
        l1 = (output - target).abs().mean() * w_l1                                       # :58-59
        cont = (output[:, 1:, :] - output[:, :-1, :]).abs().sum() / n * w_cont           # :62-67
    norm = torch.sqrt((output ** 2).sum(dim=1))                                          # :70 torch.norm(output, 2, 1): over TIME
    var = -norm.sum() / n * w_var                                                        # :71-72
    return l1 + cont + var


# ----------------------------------------------------------------------------------------------
# optimiser pieces
# ----------------------------------------------------------------------------------------------
def clip_grad_norm(grads: Dict[str, Tensor], max_norm: float) -> Tuple[Dict[str, Tensor], Tensor]:
    """torch.nn.utils.clip_grad_norm_ (train_seq2seq.py:743): coef = min(1, max_norm / (||g||_2 + 1e-6))."""
    total = torch.linalg.vector_norm(torch.stack([torch.linalg.vector_norm(g) for g in grads.values()]))
    coef = torch.clamp(max_norm / (total + 1e-6), max=1.0)
    return {k: g * coef for k, g in grads.items()}, total


def adam_step(params: Dict[str, Tensor], grads: Dict[str, Tensor], state: dict, lr: float,
              betas=(0.5, 0.999), eps: float = 1e-8) -> None:
    """torch.optim.Adam defaults (train_autoencoder_VQVAE.py:193-195): in-place on params/state."""
    b1, b2 = betas
    for k, g in grads.items():
        st = state.setdefault(k, {"step": 0, "m": torch.zeros_like(g), "v": torch.zeros_like(g)})
        st["step"] += 1
        t = st["step"]
        st["m"] = st["m"] * b1 + (1 - b1) * g
        st["v"] = st["v"] * b2 + (1 - b2) * g * g
        bc1 = 1 - b1 ** t
        bc2 = 1 - b2 ** t
        denom = st["v"].sqrt() / math.sqrt(bc2) + eps
        params[k] = params[k] - (lr / bc1) * st["m"] / denom


VQVAE_TRAINABLE_PREFIXES = ("encoder.", "decoder.")


def vqvae_trainable_keys(sd: Dict[str, Tensor]) -> List[str]:
    """Parameters that receive a (possibly all-zero) gradient in the reference's step [SURVEY §0 probe]:
    everything under encoder.* and decoder.* except BN buffers.  out_layer_encoder/decoder,
    vq_layer.pre_linear, vq_layer._ema_w and vq_layer._embedding.weight get grad=None and are therefore
    neither clipped nor stepped."""
    return [k for k in sd if k.startswith(VQVAE_TRAINABLE_PREFIXES)
            and "running_" not in k and "num_batches_tracked" not in k]


def vqvae_train_step(sd: Dict[str, Tensor], adam_state: dict, x: Tensor, masks: dict, cfg: dict,
                     epoch: int = 1) -> Dict[str, Tensor]:
    """train_iter_Autoencoder_VQ_seq2seq (train_eval/train_seq2seq.py:664-758) on (x, x):
    zero_grad; forward; loss = custom_loss + loss_vq/400 (epoch>0); backward; clip 5; Adam; returns loss/perplexity.
    sd and adam_state are updated IN PLACE (new tensors stored under the same keys)."""
    keys = vqvae_trainable_keys(sd)
    leaves = {k: sd[k].detach().clone().requires_grad_(True) for k in keys}
    work = dict(sd)
    work.update(leaves)
    fw = vqvae_forward(work, x, x, cfg, True, masks)
    closs = custom_loss(fw["outputs"], x, cfg["w_l1"], cfg["w_cont"], cfg["w_var"], cfg.get("forced"))      # :707
    loss = closs + fw["loss_vq"] / 400 if epoch > 0 else closs                           # :731-738
    gl = torch.autograd.grad(loss, [leaves[k] for k in keys], allow_unused=True)
    grads = {k: (g if g is not None else torch.zeros_like(leaves[k])) for k, g in zip(keys, gl)}
    raw_grads = {k: g.clone() for k, g in grads.items()}
    grads, gnorm = clip_grad_norm(grads, 5.0)                                            # :743
    params = {k: sd[k] for k in keys}
    adam_step(params, grads, adam_state, cfg["lr"])                                      # :744
    sd.update(params)
    for k in ("_ema_cluster_size", "_ema_w", "_embedding.weight"):
        sd["vq_layer." + k] = fw["vq"][k]
    pre = "decoder.decoder.pre_linear.1."
    sd[pre + "running_mean"], sd[pre + "running_var"] = fw["bn"]["running_mean"], fw["bn"]["running_var"]
    sd[pre + "num_batches_tracked"] = fw["bn"]["num_batches_tracked"]
    return {"loss": loss.detach(), "custom_loss": closs.detach(), "loss_vq": fw["loss_vq"].detach(),
            "perplexity": fw["perplexity"].detach(), "idx": fw["idx"], "outputs": fw["outputs"].detach(),
            "grads": raw_grads, "grad_norm": gnorm, "encoder_hidden": fw["encoder_hidden"].detach(),
            "quantized": fw["first_hidden"].detach(), "flat": fw["flat"], "dist": fw["dist"]}


# ----------------------------------------------------------------------------------------------
# Part-a frame DAE
# ----------------------------------------------------------------------------------------------
def dae_forward(x: Tensor, sd: Dict[str, Tensor], keep: Optional[Tensor], training: bool) -> Tuple[Tensor, Tensor]:
    """DAE_Network.forward (model/DAE_model.py:105-114): squeeze -> Dropout(0.2) -> Linear+ReLU -> Linear -> unsqueeze(2)."""
    inp = torch.squeeze(x)
    if training:
        inp = dropout_apply(inp, keep, 0.2)
    lat = torch.relu(linear(inp, sd["encoder.0.weight"], sd["encoder.0.bias"]))
    out = linear(lat, sd["decoder.0.weight"], sd["decoder.0.bias"])
    return out.unsqueeze(2), lat


def dae_train_step(sd: Dict[str, Tensor], adam_state: dict, x: Tensor, target: Tensor, keep: Tensor,
                   lr: float) -> Dict[str, Tensor]:
    """train_iter_DAE, vq=False, vae=False (train_eval/train_seq2seq.py:161-241): MSE, clip 5, Adam."""
    keys = list(sd.keys())
    leaves = {k: sd[k].detach().clone().requires_grad_(True) for k in keys}
    out, _ = dae_forward(x, leaves, keep, True)
    loss = ((out - target) ** 2).mean()
    gl = torch.autograd.grad(loss, [leaves[k] for k in keys])
    grads = dict(zip(keys, gl))
    raw = {k: g.clone() for k, g in grads.items()}
    grads, _ = clip_grad_norm(grads, 5.0)
    params = {k: sd[k] for k in keys}
    adam_step(params, grads, adam_state, lr)
    sd.update(params)
    return {"loss": loss.detach(), "grads": raw}


# ----------------------------------------------------------------------------------------------
# Part-d text -> gesture-code seq2seq
# ----------------------------------------------------------------------------------------------
def attn_weights(h_top: Tensor, enc_out: Tensor, w_attn: Tensor, b_attn: Tensor, v: Tensor) -> Tensor:
    """Attn.forward / Attn.score (model/text2embedding_model.py:160-198): energy = tanh(attn([h ; enc_out[t]])),
    score = v . energy, softmax over ALL Tw positions (padded positions are not masked; their encoder rows are 0).
    h_top (B,H), enc_out (Tw,B,H) -> weights (B,Tw)."""
    Tw = enc_out.shape[0]
    hrep = h_top.unsqueeze(0).expand(Tw, -1, -1)                                          # :177
    energy = torch.tanh(linear(torch.cat([hrep, enc_out], 2), w_attn, b_attn))            # (Tw,B,H)  :192-194
    score = (energy * v).sum(2)                                                           # (Tw,B)    :195-198
    return torch.softmax(score.t(), dim=1)                                                # :180


def t2e_forward(sd: Dict[str, Tensor], ids: Tensor, lengths: Tensor, codes: Tensor, cfg: dict, training: bool,
                masks: dict, vid_indices: Optional[Tensor] = None) -> Dict[str, Tensor]:
    """text2embedding_model.forward, discrete codes, EncoderRNN path
    (model/text2embedding_model.py:606-746; encoder :126-135; attention :160-198; decoder step :338-395).

    ids (B,Tw) int64 word ids (0 = PAD), lengths (B,) sorted descending, codes (B,S) int64 code ids.
    cfg: n_layers, dropout_prob, n_pre_poses, att (bool, autoencoder_att).  masks: 'emb' (S-1,B,H) keep mask of
    Dropout(0.5) on the code embedding, 'dec_l0' (S-1,B,H) decoder GRU inter-layer dropout, 'enc_l0' (Tw,B,2H) encoder
    GRU inter-layer dropout in the padded layout (ATen draws it on the packed data; padded rows are zero either way)
    (all training only).
    Without attention the decoder reads only encoder_hidden[:L] = layer-0 final states (:667-669), so the encoder's
    inter-layer dropout cannot influence any output and is not modelled.  With attention the decoder also reads
    encoder_outputs = sum of the LAST layer's two directions (:133-135).
    Returns outputs (B,S,K) with outputs[:,0] = one_hot(codes[:,0]) (:676-677).
    vid_indices (B,) int64 (the reference's inference branch, :685-692; eval mode here): one extra decode step fed with
    vid_indices runs first, its logits replace outputs[:,0] and its argmax is the next input."""
    L, p = cfg["n_layers"], cfg["dropout_prob"]
    att = bool(cfg.get("att", False))
    x = sd["encoder.embedding.weight"][ids.t()]                                           # (Tw,B,300)  :126
    H = sd["encoder.gru.weight_hh_l0"].shape[1]
    if att:
        inter = [masks["enc_l0"]] if (training and p > 0.0 and L > 1) else None
        enc_cat, enc_hidden = bigru(x, sd, "encoder.gru.", L, p if inter is not None else 0.0, inter, lengths)
        enc_out = enc_cat[:, :, :H] + enc_cat[:, :, H:]                                   # :133-135
    else:
        _, enc_hidden = bigru(x, sd, "encoder.gru.", L, 0.0, None, lengths)               # :127-131
        enc_out = None
    hidden = enc_hidden[:L]                                                               # :667-669
    pre = "decoder.decoder."
    K = sd[pre + "out.weight"].shape[0]
    bn = {"running_mean": sd[pre + "pre_linear.1.running_mean"], "running_var": sd[pre + "pre_linear.1.running_var"],
          "num_batches_tracked": sd[pre + "pre_linear.1.num_batches_tracked"]}
    cod = codes.t()                                                                       # (S,B)
    S = cod.shape[0]
    outs = [torch.nn.functional.one_hot(cod[0], K).to(x.dtype)]
    dec_in = cod[0]
    attn_list = []
    if vid_indices is not None and training:
        raise NotImplementedError("vid_indices is the inference branch: eval mode only in the oracle")
    steps = ([0] if vid_indices is not None else []) + list(range(1, S))
    for t in steps:                                                                       # :685-692, :701-744
        if t == 0:
            dec_in = vid_indices
        e = sd[pre + "embedding.weight"][dec_in]                                          # :340-343
        if training:
            e = dropout_apply(e, masks["emb"][t - 1], 0.5)                                # nn.Dropout(0.5) :253
        if att:
            w = attn_weights(hidden[-1], enc_out, sd[pre + "attn.attn.weight"], sd[pre + "attn.attn.bias"],
                             sd[pre + "attn.v"])                                          # :353-355
            context = torch.einsum("bt,tbh->bh", w, enc_out)                              # :356-359
            e = torch.cat([e, context], 1)                                                # :362-364
            attn_list.append(w)
        u = linear(e, sd[pre + "pre_linear.0.weight"], sd[pre + "pre_linear.0.bias"])
        a, nrm, nrv = batchnorm1d(u, sd[pre + "pre_linear.1.weight"], sd[pre + "pre_linear.1.bias"],
                                  bn["running_mean"], bn["running_var"], training)
        if training:
            bn["running_mean"], bn["running_var"] = nrm, nrv
            bn["num_batches_tracked"] = bn["num_batches_tracked"] + 1
        # cfg["forced"] (large-batch parity tests only, see vqvae_forward): "relu" (S-1,B,H) 0/1 pattern of the decoder's ReLU,
        # "ids" (S-1,B) the code fed at every decode step (the greedy argmax of the run under test)
        forced = cfg.get("forced") or {}
        a = torch.relu(a) if "relu" not in forced else a * forced["relu"][t - 1]
        new_h, layer_in = [], a
        for l in range(L):
            gi = linear(layer_in, sd[pre + f"gru.weight_ih_l{l}"], sd[pre + f"gru.bias_ih_l{l}"])
            h = gru_cell(gi, hidden[l], sd[pre + f"gru.weight_hh_l{l}"], sd[pre + f"gru.bias_hh_l{l}"])
            new_h.append(h)
            layer_in = h
            if l < L - 1 and training and p > 0.0:
                layer_in = dropout_apply(h, masks["dec_l0"][t - 1], p)
        hidden = torch.stack(new_h)
        logits = linear(new_h[-1], sd[pre + "out.weight"], sd[pre + "out.bias"])         # :390
        if t == 0:
            outs[0] = logits                                                              # :690
            dec_in = logits.argmax(1)                                                     # :691
            continue
        outs.append(logits)
        dec_in = cod[t] if t < cfg["n_pre_poses"] else logits.argmax(1)                   # :734-744
        if "ids" in forced and t < S - 1:
            dec_in = forced["ids"][t]                                                     # (ids[t] feeds decode step t + 1)
    return {"outputs": torch.stack(outs).transpose(0, 1), "bn": bn, "encoder_hidden": enc_hidden,
            "encoder_outputs": enc_out, "attn": attn_list}


def t2e_new_forward(sd: Dict[str, Tensor], ids: Tensor, codes: Tensor, teacher_forcing: bool) -> Tensor:
    """text2embedding_model_New.forward (model/text2embedding_model.py:933-1002) with EncoderRNN_New (:754-802) and
    DecoderRNN_New (:805-844), n_layer = 1.  The encoder's bidirectional nn.GRU is fed ONE time step per call with the
    carried hidden state (:953-955), so BOTH of its "directions" run forward in time (two independent GRUs with the
    l0 / l0_reverse weights); no length masking.  decoder_hidden = h_dir0 + h_dir1 (:971-974).  Teacher forcing
    (:979-986) runs di = 0..S-1 (overwriting outputs[0]) with inputs codes[0], codes[0], codes[1], ...; the free-running
    branch (:987-996) runs di = 1..S-1 feeding back argmax.  Returns outputs (S,B,K+2); outputs[0] = one_hot(codes[0], 514)
    unless overwritten."""
    ids_t, cod = ids.t(), codes.t()                                                       # (Tw,B), (S,B)
    x = sd["encoder.embedding.weight"][ids_t]                                             # (Tw,B,300)
    hs = []
    for suf in ("", "_reverse"):
        _, h = gru_direction(x, sd["encoder.gru.weight_ih_l0" + suf], sd["encoder.gru.weight_hh_l0" + suf],
                             sd["encoder.gru.bias_ih_l0" + suf], sd["encoder.gru.bias_hh_l0" + suf], False, None)
        hs.append(h)
    hidden = hs[0] + hs[1]                                                                # (B,H)
    S, B = cod.shape
    Kp = sd["decoder.fc_out.weight"].shape[0]
    outs = [torch.nn.functional.one_hot(cod[0], Kp).to(x.dtype)] + [x.new_zeros(B, Kp) for _ in range(S - 1)]

    def dec_step(inp, h):
        e = sd["decoder.embedding.weight"][inp]
        gi = linear(e, sd["decoder.gru.weight_ih_l0"], sd["decoder.gru.bias_ih_l0"])
        h = gru_cell(gi, h, sd["decoder.gru.weight_hh_l0"], sd["decoder.gru.bias_hh_l0"])
        return linear(h, sd["decoder.fc_out.weight"], sd["decoder.fc_out.bias"]), h

    dec_in = cod[0]
    if teacher_forcing:
        for di in range(S):
            out, hidden = dec_step(dec_in, hidden)
            dec_in = cod[di]
            outs[di] = out
    else:
        for di in range(1, S):
            out, hidden = dec_step(dec_in, hidden)
            dec_in = out.argmax(1)
            outs[di] = out
    return torch.stack(outs)


def t2e_loss(outputs: Tensor, codes: Tensor) -> Tensor:
    """CrossEntropyLoss over decode steps 1..S-1 (train_eval/train_seq2seq.py:520-530)."""
    K = outputs.shape[2]
    return torch.nn.functional.cross_entropy(outputs[:, 1:, :].reshape(-1, K), codes[:, 1:].reshape(-1))


def t2e_trainable_keys(sd: Dict[str, Tensor]) -> List[str]:
    return [k for k in sd if "running_" not in k and "num_batches_tracked" not in k]


def t2e_train_step(sd: Dict[str, Tensor], adam_state: dict, ids: Tensor, lengths: Tensor, codes: Tensor, masks: dict,
                   cfg: dict) -> Dict[str, Tensor]:
    """train_iter_text2embedding (train_eval/train_seq2seq.py:462-538): CE, clip 5, Adam(betas (0.5,0.999))."""
    keys = t2e_trainable_keys(sd)
    leaves = {k: sd[k].detach().clone().requires_grad_(True) for k in keys}
    work = dict(sd)
    work.update(leaves)
    fw = t2e_forward(work, ids, lengths, codes, cfg, True, masks)
    loss = t2e_loss(fw["outputs"], codes)
    gl = torch.autograd.grad(loss, [leaves[k] for k in keys], allow_unused=True)
    grads = {k: (g if g is not None else torch.zeros_like(leaves[k])) for k, g in zip(keys, gl)}
    raw = {k: g.clone() for k, g in grads.items()}
    grads, gnorm = clip_grad_norm(grads, 5.0)
    params = {k: sd[k] for k in keys}
    adam_step(params, grads, adam_state, cfg["lr"])
    sd.update(params)
    pre = "decoder.decoder.pre_linear.1."
    sd[pre + "running_mean"], sd[pre + "running_var"] = fw["bn"]["running_mean"], fw["bn"]["running_var"]
    sd[pre + "num_batches_tracked"] = fw["bn"]["num_batches_tracked"]
    return {"loss": loss.detach(), "outputs": fw["outputs"].detach(), "grads": raw, "grad_norm": gnorm}


# ----------------------------------------------------------------------------------------------
# helpers shared by tests / bench
# ----------------------------------------------------------------------------------------------
def unpack_mask(bits: np.ndarray, shape) -> Tensor:
    n = int(np.prod(shape))
    return torch.from_numpy(np.unpackbits(bits)[:n].reshape(shape).copy())


def init_vqvae_state(D: int, H: int, L: int, K: int, seed: int = 0) -> Dict[str, Tensor]:
    """Random-init weights with the reference's state_dict keys/shapes/init distributions
    (nn.Linear/nn.GRU default U(-1/sqrt(fan), 1/sqrt(fan)); codebook U(-1,1) :1204; _ema_w N(0,1) :1211)."""
    g = torch.Generator().manual_seed(seed)
    sd: Dict[str, Tensor] = {}

    def U(shape, bound):
        return (torch.rand(shape, generator=g) * 2 - 1) * bound

    def lin(name, out_f, in_f):
        b = 1.0 / math.sqrt(in_f)
        sd[name + ".weight"], sd[name + ".bias"] = U((out_f, in_f), b), U((out_f,), b)

    def gru(prefix, in_f, bidir):
        b = 1.0 / math.sqrt(H)
        for l in range(L):
            i_f = in_f if l == 0 else (2 * H if bidir else H)
            for suf in ([""] + (["_reverse"] if bidir else [])):
                sd[f"{prefix}weight_ih_l{l}{suf}"] = U((3 * H, i_f), b)
                sd[f"{prefix}weight_hh_l{l}{suf}"] = U((3 * H, H), b)
                sd[f"{prefix}bias_ih_l{l}{suf}"] = U((3 * H,), b)
                sd[f"{prefix}bias_hh_l{l}{suf}"] = U((3 * H,), b)

    lin("encoder.in_layer", H, D)
    gru("encoder.gru.", H, True)
    lin("out_layer_encoder.0", H, H)
    lin("out_layer_decoder.0", D, H)
    lin("decoder.decoder.pre_linear.0", H, D)
    sd["decoder.decoder.pre_linear.1.weight"] = torch.ones(H)
    sd["decoder.decoder.pre_linear.1.bias"] = torch.zeros(H)
    sd["decoder.decoder.pre_linear.1.running_mean"] = torch.zeros(H)
    sd["decoder.decoder.pre_linear.1.running_var"] = torch.ones(H)
    sd["decoder.decoder.pre_linear.1.num_batches_tracked"] = torch.zeros((), dtype=torch.int64)
    gru("decoder.decoder.gru.", H, False)
    lin("decoder.decoder.out_layer", D, H)
    E = H * L
    sd["vq_layer._ema_cluster_size"] = torch.zeros(K)
    sd["vq_layer._ema_w"] = torch.randn(K, E, generator=g)
    lin("vq_layer.pre_linear", E, E)
    sd["vq_layer._embedding.weight"] = U((K, E), 1.0)
    return sd
