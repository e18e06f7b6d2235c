"""TEST / BASELINE INFRASTRUCTURE ONLY (never imported by the product path; see oracle/g2v_oracle.py's header).

The second CPU leg of bench.py's `cpu_baseline`: the same chunk VQ-VAE train step as oracle/g2v_oracle.py's `vqvae_train_step`,
but with the recurrent layers as `torch.nn.GRU` MODULES the way the reference builds them -- i.e. on ATen's fused CPU RNN
kernels -- instead of the oracle's explicit per-step Python formulas.  The functional oracle is the parity checker (pinned to the
reference's golden vectors); it is about 2x slower than the reference's own modules on the same host,
which flatters a GPU / CPU ratio.  This module is pinned to the functional oracle in tests/test_oracle_golden.py (same loss and
same gradients at dropout_prob = 0 with the decoder's explicit Dropout(0.95) masks) and is what `cpu_baseline.fused_rnn` times.

Follows, module for module:
  EncoderRNN                 model/Autoencoder_VQVAE_model.py:30-100   Linear(D,H) -> nn.GRU(H,H,L,bidirectional) -> sum of directions
  VQ_Payam_EMA               :1182-1301   (the arithmetic is oracle/g2v_oracle.py's vq_ema_forward: shared, not restated)
  BahdanauAttnDecoderRNN     :401-592     Dropout(0.95) -> Linear(D,H)+BatchNorm1d+ReLU -> nn.GRU(H,H,L) one step -> Linear(H,D)
  Autoencoder_VQVAE.forward  :901-1072    the T-1 step loop
  train_iter_Autoencoder_VQ_seq2seq   train_eval/train_seq2seq.py:664-758
"""
from __future__ import annotations

from typing import Dict, Optional

import torch
from torch import nn

from . import g2v_oracle as O


class VQVAEModules(nn.Module):
    """Parameters under the reference's state_dict names, so that `load_state_dict(sd, strict=False)` takes an oracle state."""

    def __init__(self, D: int, H: int, L: int, p: float):
        super().__init__()
        self.encoder = nn.Module()
        self.encoder.in_layer = nn.Linear(D, H)
        self.encoder.gru = nn.GRU(H, H, L, dropout=p, bidirectional=True)
        self.decoder = nn.Module()
        self.decoder.decoder = nn.Module()
        dd = self.decoder.decoder
        dd.pre_linear = nn.Sequential(nn.Linear(D, H), nn.BatchNorm1d(H), nn.ReLU())
        dd.gru = nn.GRU(H, H, L, dropout=p)
        dd.out_layer = nn.Linear(H, D)
        self.H, self.L = H, L


def build(sd: Dict[str, torch.Tensor], D: int, H: int, L: int, p: float) -> VQVAEModules:
    m = VQVAEModules(D, H, L, p)
    own = m.state_dict()
    m.load_state_dict({k: v for k, v in sd.items() if k in own}, strict=False)
    m.train(True)
    return m


def train_step(m: VQVAEModules, opt: torch.optim.Optimizer, vq_sd: Dict[str, torch.Tensor], x: torch.Tensor, keep95: Optional[torch.Tensor],
               cfg: dict, epoch: int = 1) -> Dict[str, torch.Tensor]:
    """One train_iter on (x, x).  keep95 (T-1,B,D) uint8: the decoder's Dropout(0.95) masks (None: drawn with torch's RNG).
    The encoder's input dropout and the GRUs' inter-layer dropout (dropout_prob > 0) use torch's own RNG: this leg is a TIMING
    baseline; it is compared with the functional oracle at dropout_prob = 0.  vq_sd holds the quantiser state
    (`vq_layer.*` keys), updated in place like the oracle's."""
    L, H, p = m.L, m.H, cfg["dropout_prob"]
    opt.zero_grad(set_to_none=True)
    xt = x.transpose(0, 1)                                                    # (T,B,D)  :956
    T, B, D = xt.shape
    xin = torch.nn.functional.dropout(xt, p, True) if p > 0 else xt           # :957
    _, hidden = m.encoder.gru(m.encoder.in_layer(xin))                        # :93-94 (fused bidirectional GRU)
    dec_hidden = hidden[:L].contiguous()                                      # :971-973
    vq = O.vq_ema_forward(dec_hidden, vq_sd, "vq_layer.", cfg["commitment_cost"], True)
    h = vq["quantized"]
    dd = m.decoder.decoder
    outs, dec_in = [xt[0]], xt[0]
    for t in range(1, T):                                                     # :1041-1054
        inp = dec_in if cfg.get("conditioned", True) else torch.zeros_like(dec_in)
        u = O.dropout_apply(inp, keep95[t - 1], 0.95) if keep95 is not None else torch.nn.functional.dropout(inp, 0.95, True)
        a = dd.pre_linear(u)                                                  # Linear + BatchNorm1d (batch statistics) + ReLU
        out, h = dd.gru(a.unsqueeze(0), h)                                    # one fused step of the L-layer GRU
        y = dd.out_layer(out[0])
        outs.append(y)
        dec_in = xt[t] if t < cfg["n_pre_poses"] else y
    outputs = torch.stack(outs).transpose(0, 1)
    closs = O.custom_loss(outputs, x, cfg["w_l1"], cfg["w_cont"], cfg["w_var"])
    loss = closs + vq["loss"] / 400 if epoch > 0 else closs
    loss.backward()
    torch.nn.utils.clip_grad_norm_(m.parameters(), 5.0)                       # :743
    opt.step()                                                                # :744
    for k in ("_ema_cluster_size", "_ema_w", "_embedding.weight"):
        vq_sd["vq_layer." + k] = vq[k]
    return {"loss": loss.detach(), "custom_loss": closs.detach(), "idx": vq["idx"], "outputs": outputs.detach()}


def make(sd: Dict[str, torch.Tensor], D: int, H: int, L: int, cfg: dict):
    """(modules, Adam(lr, betas (0.5, 0.999)) over them, quantiser state) from an oracle state dict"""
    m = build(sd, D, H, L, cfg["dropout_prob"])
    opt = torch.optim.Adam(m.parameters(), lr=cfg["lr"], betas=(0.5, 0.999))  # train_autoencoder_VQVAE.py:193-195
    vq_sd = {k: v.clone() for k, v in sd.items() if k.startswith("vq_layer.")}
    return m, opt, vq_sd
