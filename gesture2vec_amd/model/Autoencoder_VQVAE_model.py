"""Part b: gesture-chunk VQ-VAE -- the reference's operator surface on the MI355X-native kernels.

Mirrors `scripts/model/Autoencoder_VQVAE_model.py` of pjyazdian/Gesture2Vec: same class names, constructor
signatures, `forward` signatures / return tuples and `state_dict` keys (so reference `.bin` checkpoints load
and vice-versa), but no arithmetic happens in torch.nn: `nn.Linear` / `nn.BatchNorm1d` / `nn.Embedding` are
used purely as parameter containers (names, shapes, init), and every forward/backward number comes from the HIP
kernels behind include/g2v.h via `gesture2vec_amd.engine.VQVAEEngine`.

Differences from the reference, all deliberate:
  * `Autoencoder_VQVAE` wires `VQ_Payam_EMA` (the reference constructs it :801-807 and then overwrites it with the
    soft GSSoft quantiser :816-820; the task's north star is the EMA quantiser with hard argmin codes).
  * `_ema_w` / `_embedding.weight` are updated IN PLACE by the EMA kernel instead of being re-created as new
    nn.Parameter objects every forward (:1276-1282); values and state_dict keys are identical, and like in the
    reference they never receive a gradient.
  * Dropout masks come from a counter-based Philox kernel (or are supplied explicitly, `set_dropout_masks`),
    because CPU and GPU RNG streams cannot agree anyway.
  * GPU only: `.forward` raises if the module is not on an MI355X (no CPU path).
Supported configuration: autoencoder_vq "True", autoencoder_vae "False", n_layers 2; autoencoder_att "False" (fused engine +
rollout kernels) and "True" (module-level path: step-level decoder with Bahdanau attention).
"""
from __future__ import annotations

import math
from typing import Optional, Tuple

import torch
import torch.nn as nn

from .. import functional as Fn
from .. import ops
from ..engine import VQVAEEngine

debug = False


class _GRUParams(nn.Module):
    """Parameter container with nn.GRU's names, shapes and init (U(-1/sqrt(H), 1/sqrt(H)))."""

    def __init__(self, input_size: int, hidden_size: int, num_layers: int, dropout: float = 0.0,
                 bidirectional: bool = False):
        super().__init__()
        self.input_size, self.hidden_size, self.num_layers = input_size, hidden_size, num_layers
        self.dropout, self.bidirectional = dropout, bidirectional
        H = hidden_size
        k = 1.0 / math.sqrt(H)
        for l in range(num_layers):
            in_f = input_size if l == 0 else H * (2 if bidirectional else 1)
            for suf in ([""] + (["_reverse"] if bidirectional else [])):
                for name, shape in ((f"weight_ih_l{l}{suf}", (3 * H, in_f)), (f"weight_hh_l{l}{suf}", (3 * H, H)),
                                    (f"bias_ih_l{l}{suf}", (3 * H,)), (f"bias_hh_l{l}{suf}", (3 * H,))):
                    self.register_parameter(name, nn.Parameter(torch.empty(shape).uniform_(-k, k)))

    def flatten_parameters(self):  # API compatibility; the flat buffer is managed by the engine
        return None


class EncoderRNN(nn.Module):
    """Linear(D->H) + bidirectional GRU, directions summed (reference :30-100)."""

    def __init__(self, input_size: int, hidden_size: int, n_layers: int = 1, dropout: float = 0.5,
                 pre_trained_embedding=None):
        super().__init__()
        self.input_size, self.hidden_size, self.n_layers, self.dropout = input_size, hidden_size, n_layers, dropout
        self.in_layer = nn.Linear(input_size, hidden_size)
        self.gru = _GRUParams(hidden_size, hidden_size, n_layers, dropout=dropout, bidirectional=True)
        self.do_flatten_parameters = False

    def forward(self, input_seqs: torch.Tensor, hidden: Optional[torch.Tensor] = None, keep_in: Optional[torch.Tensor] = None,
                in_scale: float = 1.0, keep_inter: Optional[torch.Tensor] = None):
        """(T,B,D) -> outputs (T,B,H) [sum of directions of the LAST layer], hidden (2L,B,H), all layers evaluated.
        With autograd enabled the layers are chained through autograd nodes (the attention model trains through this path;
        `keep_in` (T,B,D) / `keep_inter` (T,B,2H) are the uint8 keep masks of the input dropout and of nn.GRU's inter-layer
        dropout); otherwise the plain inference sequence below runs.  Inside the attention-free Autoencoder_VQVAE the engine
        runs the fused training path instead."""
        if hidden is not None:
            raise NotImplementedError("non-zero initial hidden state")
        if torch.is_grad_enabled() or keep_in is not None or keep_inter is not None:
            return self._forward_autograd(input_seqs, keep_in, in_scale, keep_inter)
        T, B, D = input_seqs.shape
        H, L = self.hidden_size, self.n_layers
        x = input_seqs.contiguous()
        xin = ops.linear_fwd(x, self.in_layer.weight.data, self.in_layer.bias.data, M=T * B)
        layer_in, in_f = xin, H
        hiddens = []
        out = None
        for l in range(L):
            out = torch.empty((T, B, 2 * H), dtype=torch.float32, device=x.device)
            for d, suf in enumerate(("", "_reverse")):
                g = self.gru
                gi = ops.linear_fwd(layer_in, getattr(g, f"weight_ih_l{l}{suf}").data, getattr(g, f"bias_ih_l{l}{suf}").data,
                                    M=T * B)
                _, h_n, _ = ops.gru_seq_fwd(gi, getattr(g, f"weight_hh_l{l}{suf}").data.contiguous(),
                                            getattr(g, f"bias_hh_l{l}{suf}").data, T, B, H, reverse=bool(d),
                                            hs=out[:, :, d * H:], hs_ld=2 * H, save_gates=False)
                hiddens.append(h_n)
            layer_in, in_f = out.view(T * B, 2 * H), 2 * H   # (inter-layer dropout is a training-only effect)
        summed = torch.empty((T, B, H), dtype=torch.float32, device=x.device)
        ops.add_halves(out, 2 * H, out[:, :, H:], 2 * H, summed, H, T * B, H)
        return summed, torch.stack(hiddens)

    def _forward_autograd(self, input_seqs, keep_in, in_scale, keep_inter):
        T, B, D = input_seqs.shape
        H, L = self.hidden_size, self.n_layers
        x = input_seqs.contiguous().view(T * B, D)
        k_in = keep_in.contiguous().view(T * B, D) if keep_in is not None else None
        layer_in = Fn.linear(x, self.in_layer.weight, self.in_layer.bias, keep=k_in, scale=in_scale)
        hiddens, keep, scale = [], None, 1.0
        out_f = out_b = None
        g = self.gru
        for l in range(L):
            gis = [Fn.linear(layer_in, getattr(g, f"weight_ih_l{l}{suf}"), getattr(g, f"bias_ih_l{l}{suf}"),
                             keep=keep, scale=scale).view(T, B, 3 * H) for suf in ("", "_reverse")]
            out_f, hn_f, out_b, hn_b = Fn.GRUBiDirFn.apply(
                gis[0], gis[1], getattr(g, f"weight_hh_l{l}"), getattr(g, f"bias_hh_l{l}"),
                getattr(g, f"weight_hh_l{l}_reverse"), getattr(g, f"bias_hh_l{l}_reverse"), None)
            hiddens += [hn_f, hn_b]
            if l + 1 < L:
                layer_in = torch.cat([out_f, out_b], dim=2).view(T * B, 2 * H)       # layout only
                if self.training and self.dropout > 0 and keep_inter is not None:
                    keep, scale = keep_inter.contiguous().view(T * B, 2 * H), 1.0 / (1.0 - self.dropout)
        return Fn.SumHalvesFn.apply(out_f, out_b), torch.stack(hiddens)


class Attn(nn.Module):
    """Bahdanau attention scoring (reference :337-398; the same class text2embedding_model.py:138-198 defines): parameters
    `attn` = Linear(2H -> H) and `v` (H); forward(hidden (B,H), encoder_outputs (T,B,H)) -> softmax weights (B,1,T).
    energy_t = v . tanh(W [h ; enc_t] + b) is evaluated as tanh(hp + ep_t) with hp = h W[:, :H]^T + b (per step) and
    ep = enc W[:, H:]^T (once per sequence) by g2v_attn_fwd / g2v_attn_bwd."""

    def __init__(self, hidden_size: int):
        super().__init__()
        self.hidden_size = hidden_size
        self.attn = nn.Linear(hidden_size * 2, hidden_size)
        self.v = nn.Parameter(torch.rand(hidden_size))
        stdv = 1.0 / math.sqrt(self.v.size(0))
        self.v.data.normal_(mean=0, std=stdv)

    def project_encoder(self, encoder_outputs: torch.Tensor) -> torch.Tensor:
        """ep = enc W_attn[:, H:]^T (T,B,H): the step-independent half of the energy pre-activation."""
        H = self.hidden_size
        T, B, _ = encoder_outputs.shape
        return Fn.linear(encoder_outputs.reshape(T * B, H), self.attn.weight[:, H:]).view(T, B, H)

    def context(self, hidden: torch.Tensor, encoder_outputs: torch.Tensor, enc_proj: Optional[torch.Tensor] = None):
        """(context (B,H), weights (B,T)) for decoder state `hidden` (B,H)."""
        H = self.hidden_size
        if enc_proj is None:
            enc_proj = self.project_encoder(encoder_outputs)
        hp = Fn.linear(hidden, self.attn.weight[:, :H], self.attn.bias)
        return Fn.AttnFn.apply(hp, enc_proj, encoder_outputs, self.v)

    def forward(self, hidden: torch.Tensor, encoder_outputs: torch.Tensor) -> torch.Tensor:
        _, w = self.context(hidden, encoder_outputs)
        return w.unsqueeze(1)


class BahdanauAttnDecoderRNN(nn.Module):
    """One decode step (reference :401-592).  Inside `Autoencoder_VQVAE.forward` the T-1 steps run in the rollout kernels;
    `forward` below is the reference's step-at-a-time API that its other callers use (inference_Autoencoder.py:207-214,
    Clustering.py:217-224: `rnn.decoder(None, x, h, enc, None)`), composed from the same HIP operators through small
    autograd nodes, with or without attention (`autoencoder_att`)."""

    def __init__(self, args, input_size: int, hidden_size: int, output_size: int, n_layers: int = 1,
                 dropout_p: float = 0.1, discrete_representation: bool = False, speaker_model=None):
        super().__init__()
        self.hidden_size, self.output_size, self.n_layers, self.dropout_p = hidden_size, output_size, n_layers, dropout_p
        self.discrete_representation = discrete_representation
        self.speaker_model = speaker_model
        if discrete_representation:
            raise NotImplementedError("discrete_representation decoder belongs to Part d (text2embedding_model)")
        self.autoencoder_conditioned = args.autoencoder_conditioned == "True"
        self.att_use = args.autoencoder_att == "True"
        if self.att_use:
            self.attn = Attn(hidden_size)
        linear_input_size = input_size + hidden_size if self.att_use else input_size          # :472-475
        self.pre_linear = nn.Sequential(nn.Linear(linear_input_size, hidden_size), nn.BatchNorm1d(hidden_size), nn.ReLU(inplace=True))
        self.gru = _GRUParams(hidden_size, hidden_size, n_layers, dropout=dropout_p)
        if args.autoencoder_fixed_weight == "True":
            self.autoencoder_fixed_weight = True
            for param in self.gru.parameters():
                param.requires_grad = False
        self.out_layer = nn.Linear(hidden_size, output_size)
        self.do_flatten_parameters = False
        self._mask_queue = []          # explicit (keep95, keep_l0) pairs for the next calls (parity tests)
        self._rng_counter = None
        self.rng_seed = 0

    def freeze_attn(self) -> None:
        for param in self.attn.parameters():
            param.requires_grad = False

    # ---- dropout masks: the inline nn.Dropout(0.95) (:570) is ALWAYS active, nn.GRU's inter-layer dropout in training only
    def set_step_masks(self, keep95_list, keep_l0_list=None):
        """Explicit uint8 keep masks for the next len(keep95_list) calls: keep95 (B, linear_input_size) for the inline
        Dropout(0.95), keep_l0 (B,H) for the dropout between GRU layer 0 and 1."""
        n = len(keep95_list)
        l0 = list(keep_l0_list) if keep_l0_list is not None else [None] * n
        self._mask_queue = [(k.contiguous(), (m.contiguous() if m is not None else None)) for k, m in zip(keep95_list, l0)]

    def _draw(self, shape, keep_prob, dev):
        if self._rng_counter is None or self._rng_counter.device != dev:
            self._rng_counter = torch.zeros(1, dtype=torch.int64, device=dev)
        return ops.keep_mask(torch.empty(shape, dtype=torch.uint8, device=dev), keep_prob, self.rng_seed, self._rng_counter)

    def forward(self, motion_input: torch.Tensor, last_hidden: torch.Tensor, encoder_outputs: Optional[torch.Tensor] = None,
                vid_indices=None, enc_proj: Optional[torch.Tensor] = None):
        """motion_input (B,D), last_hidden (L,B,H), encoder_outputs (T,B,H) [read only with attention] ->
        (output (B,D), hidden (L,B,H), attn_weights (B,1,T) | None)   (:499-592)."""
        if not motion_input.is_cuda:
            raise RuntimeError("BahdanauAttnDecoderRNN runs on the MI355X kernels only (no CPU fallback)")
        B = motion_input.size(0)
        H, L = self.hidden_size, self.n_layers
        training = self.training
        x = motion_input.reshape(B, -1)
        attn_weights = None
        if self.att_use:
            context, w = self.attn.context(last_hidden[-1], encoder_outputs, enc_proj)       # :545-551
            x = torch.cat((x, context), 1)                                                   # :554-556 (layout only)
            attn_weights = w.unsqueeze(1)
        if not self.autoencoder_conditioned:
            x = torch.zeros_like(x)                                                          # :568-569
        if self._mask_queue:
            keep95, keep_l0 = self._mask_queue.pop(0)
        else:
            keep95 = self._draw(tuple(x.shape), 0.05, x.device)
            keep_l0 = self._draw((B, H), 1.0 - self.dropout_p, x.device) if (training and self.dropout_p > 0 and L > 1) else None
        lin, bn = self.pre_linear[0], self.pre_linear[1]
        u = Fn.linear(x.contiguous(), lin.weight, lin.bias, keep=keep95, scale=20.0)          # Dropout(0.95) fused: 1/(1-0.95)
        a = Fn.BatchNormReluFn.apply(u, bn.weight, bn.bias, bn.running_mean, bn.running_var, training, True)
        if training:
            bn.num_batches_tracked += 1
        new_h, layer_in, keep, scale = [], a, None, 1.0
        g = self.gru
        for l in range(L):
            gi = Fn.linear(layer_in, getattr(g, f"weight_ih_l{l}"), getattr(g, f"bias_ih_l{l}"), keep=keep, scale=scale)
            _, h_n = Fn.GRUDirFn.apply(gi.view(1, B, 3 * H), getattr(g, f"weight_hh_l{l}"), getattr(g, f"bias_hh_l{l}"),
                                       last_hidden[l], None, False)
            new_h.append(h_n)
            layer_in = h_n
            if training and self.dropout_p > 0 and keep_l0 is not None:
                keep, scale = keep_l0, 1.0 / (1.0 - self.dropout_p)      # nn.GRU inter-layer dropout, fused into the next Linear
        output = Fn.linear(new_h[-1], self.out_layer.weight, self.out_layer.bias)
        return output, torch.stack(new_h), attn_weights


class Generator(nn.Module):
    """Decoder wrapper (reference :595-683)."""

    def __init__(self, args, motion_dim: int, discrete_representation: bool = False, speaker_model=None):
        super().__init__()
        self.output_size = motion_dim
        self.n_layers = args.n_layers
        self.discrete_representation = discrete_representation
        self.decoder = BahdanauAttnDecoderRNN(args=args, input_size=args.rep_learning_dim, hidden_size=args.hidden_size,
                                              output_size=args.rep_learning_dim, n_layers=self.n_layers,
                                              dropout_p=args.dropout_prob,
                                              discrete_representation=discrete_representation, speaker_model=speaker_model)
        self.is_training = True

    def freeze_attn(self) -> None:
        self.decoder.freeze_attn()

    def forward(self, z, motion_input: torch.Tensor, last_hidden: torch.Tensor, encoder_output: Optional[torch.Tensor],
                vid_indices=None, **kw):
        """One decode step (:646-683): `z` (noise vector) is appended to motion_input when given."""
        if z is None:
            input_with_noise_vec = motion_input
        else:
            assert not self.discrete_representation
            input_with_noise_vec = torch.cat([motion_input, z], dim=1)
        return self.decoder(input_with_noise_vec, last_hidden, encoder_output, vid_indices, **kw)


class VQ_Payam_EMA(nn.Module):
    """EMA vector quantiser (reference :1182-1301).  forward(inputs) -> (loss, quantized, perplexity, encodings)."""

    def __init__(self, num_embeddings: int, embedding_dim: int, commitment_cost: float, decay: float,
                 epsilon: float = 1e-5):
        super().__init__()
        self._embedding_dim, self._num_embeddings = embedding_dim, num_embeddings
        self.pre_linear = nn.Linear(embedding_dim, embedding_dim)
        self._embedding = nn.Embedding(num_embeddings, embedding_dim)
        self._embedding.weight.data.uniform_(-1, 1)                                  # :1204
        self._commitment_cost = commitment_cost
        self.register_buffer("_ema_cluster_size", torch.zeros(num_embeddings))
        self._ema_w = nn.Parameter(torch.Tensor(num_embeddings, embedding_dim))
        self._ema_w.data.normal_()                                                   # :1211
        self._decay, self._epsilon = decay, epsilon
        for p in (self._ema_w, self._embedding.weight, self.pre_linear.weight, self.pre_linear.bias):
            p.requires_grad_(False)   # they never get a gradient in the reference either

    def forward(self, inputs: torch.Tensor):
        """Standalone call (datasets / clustering / inference call sites, lmdb_data_loader.py:1274-1281): rows are
        `inputs.view(-1, E)`.  Gradient wrt `inputs` flows through `_VQFn` (straight-through + commitment)."""
        return _VQFn.apply(inputs, self)

    def assign(self, inputs: torch.Tensor) -> torch.Tensor:
        """Code indices only (bulk code-assignment path): argmin_k ||pre_linear(x) - W_k||^2, int64 (N,)."""
        E, K = self._embedding_dim, self._num_embeddings
        z = inputs.contiguous().view(-1, E)
        flat = ops.linear_fwd(z, self.pre_linear.weight.data, self.pre_linear.bias.data)
        W = self._embedding.weight.data
        wsq = ops.vq_code_sqnorm(W)
        if z.shape[0] >= 131072 and E == 128 and K % 128 == 0:
            # a corpus' worth of rows: bf16 split screening on the bf16 matrix pipe + exact fp32 re-check of the undecided
            # rows (g2v_vq_assign_bulk: the fp32 kernel's indices, 2.0x faster at 2^18 rows and 2.6x at 2^20; break-even 2^16)
            return ops.vq_assign_bulk(flat, W, wsq)
        # (every other shape, e.g. the checkpoints' E = 400: ops.vq_assign takes the packed eight-wave kernel from 2048 rows, with
        #  2 / 4 row tiles per workgroup for a corpus' worth of rows -- round 5; it used to degrade to the generic kernel)
        idx, _, _, _ = ops.vq_assign(flat, None, W, wsq, want_quantized=False)
        return idx


class _VQFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, inputs, mod, use_pre=True):
        E, K = mod._embedding_dim, mod._num_embeddings
        z = inputs.contiguous().view(-1, E)
        N = z.shape[0]
        W = mod._embedding.weight.data
        # VQ_Payam_EMA measures distances on pre_linear(z) but takes loss / straight-through on the raw z (:1230,1285)
        flat = ops.linear_fwd(z, mod.pre_linear.weight.data, mod.pre_linear.bias.data) if use_pre else z
        wsq = ops.vq_code_sqnorm(W)
        idx, quant, _, sse = ops.vq_assign(flat, z, W, wsq)
        stats = ops.vq_stats(idx, flat, K)
        scalars = ops.vq_ema_update(stats, sse, mod._ema_cluster_size, mod._ema_w.data, W, wsq, N, N, E, K,
                                    mod._commitment_cost, mod._decay, mod._epsilon, mod.training)
        encodings = torch.zeros((N, K), dtype=torch.float32, device=z.device)
        encodings.scatter_(1, idx.unsqueeze(1), 1.0)     # layout glue: the (N,K) one-hot callers argmax over (:1246-1250)
        ctx.set_materialize_grads(False)
        ctx.save_for_backward(z, quant)
        ctx.beta, ctx.shape = mod._commitment_cost, inputs.shape
        ctx.mark_non_differentiable(encodings)
        return scalars[0].clone(), quant.view(inputs.shape), scalars[1].clone(), encodings

    @staticmethod
    def backward(ctx, g_loss, g_quant, g_perp, g_enc):
        z, quant = ctx.saved_tensors
        gl = g_loss.reshape(1).contiguous() if g_loss is not None else None
        gq = g_quant.contiguous().view(z.shape) if g_quant is not None else None
        gz = ops.vq_bwd(gq, gl, z, quant, None, ctx.beta)
        return gz.view(ctx.shape), None, None


class Autoencoder_VQVAE(nn.Module):
    """Chunk autoencoder: EncoderRNN -> VQ_Payam_EMA on encoder_hidden[:L] -> T-1 autoregressive decode steps
    (reference :686-1085).  forward(in_poses, out_poses, vq_layer_active) ->
    (outputs (B,T,D), decoder_first_hidden (L,B,H), loss_vq, perplexity_vq)."""

    def __init__(self, args, pose_dim: int, n_frames: int):
        super().__init__()
        self.CNN = False
        self.encoder = EncoderRNN(args.rep_learning_dim, args.hidden_size, args.n_layers, dropout=args.dropout_prob,
                                  pre_trained_embedding=None)
        # unused by forward but part of the reference's state_dict (:750-755)
        self.out_layer_encoder = nn.Sequential(nn.Linear(args.hidden_size, args.hidden_size), nn.Tanh())
        self.out_layer_decoder = nn.Sequential(nn.Linear(args.hidden_size, pose_dim))
        self.decoder = Generator(args, pose_dim, speaker_model=None)
        if args.autoencoder_vae == "True":
            raise NotImplementedError("autoencoder_vae == 'True' is outside the accelerated hot path")
        self.VAE = False
        if args.autoencoder_vq != "True":
            raise NotImplementedError("autoencoder_vq == 'False' is outside the accelerated hot path")
        self.vq = True
        self.vq_components = int(args.autoencoder_vq_components)
        self.commitment_cost = float(args.autoencoder_vq_commitment_cost)
        # The reference builds VQ_Payam_EMA (:801-807) and then overwrites it with VQ_Payam_GSSoft (:816-820).  The EMA
        # quantiser is the north star and the fused engine path; `args.autoencoder_vq_quantizer = "gssoft"` (not a
        # reference key; load_checkpoint_and_model sets it when a checkpoint carries the soft quantiser's tensors) builds the
        # model exactly as the reference ships it: same state_dict, encoder / decoder stages of the engine around the module.
        self.quantizer = str(getattr(args, "autoencoder_vq_quantizer", "ema")).lower()
        if self.quantizer == "ema":
            self.vq_layer = VQ_Payam_EMA(self.vq_components, args.hidden_size * args.n_layers, self.commitment_cost, 0.85)
        elif self.quantizer == "gssoft":
            self.vq_layer = VQ_Payam_GSSoft(self.vq_components, args.hidden_size * args.n_layers, self.commitment_cost)
        else:
            raise ValueError(f"autoencoder_vq_quantizer must be 'ema' or 'gssoft', got {self.quantizer!r}")
        self.n_frames = n_frames
        self.n_pre_poses = args.n_pre_poses
        self.pose_dim = args.rep_learning_dim
        self.autoencoder_conditioned = args.autoencoder_conditioned == "True"
        self.dropout_prob = float(args.dropout_prob)
        self.hidden_size, self.n_layers = args.hidden_size, args.n_layers
        for p in list(self.out_layer_encoder.parameters()) + list(self.out_layer_decoder.parameters()):
            p.requires_grad_(False)   # grad is None in the reference: never reached by forward
        self._engine: Optional[VQVAEEngine] = None
        self._engine_bound = False
        self._engine_probe = None
        self._explicit_masks = False
        self.rng_seed = 0
        # autoencoder_att == "True": the decoder attends over the encoder outputs (:545-556).  That model runs through the
        # module-level path (_forward_attention): full encoder -> quantiser -> T-1 step-level decoder calls, every operator a
        # HIP kernel chained by autograd nodes; the fused engine / rollout kernels are the attention-free model's.
        self.att_use = self.decoder.decoder.att_use
        self._att_masks = None
        self._rng_counter = None

    # ------------------------------------------------------------------ engine binding
    def engine(self) -> VQVAEEngine:
        if self.att_use:
            raise NotImplementedError("the fused engine implements the attention-free decoder; autoencoder_att == 'True' "
                                      "models run through Autoencoder_VQVAE.forward's module-level path")
        dev = self.encoder.in_layer.weight.device
        if dev.type != "cuda":
            raise RuntimeError("Autoencoder_VQVAE runs on the MI355X kernels only: move the module to the GPU first "
                               "(there is deliberately no CPU fallback)")
        eng = self._engine
        # fast path (150 us -> a few us per call; train_iter calls this every iteration): the full re-homing walk below runs on
        # the first call and after anything that can re-create tensors (Module._apply: .to() / .cuda() / .float(); a state
        # load); between those, three representative tensors are probed
        if eng is not None and self._engine_bound and eng.device == dev:
            probe = self._engine_probe
            if (probe[0].data_ptr() == probe[1] and probe[2].data_ptr() == probe[3] and probe[4].data_ptr() == probe[5]
                    and probe[0].grad is not None):
                return eng
        sd_params = dict(self.named_parameters())
        if eng is None or eng.device != dev:
            eng = VQVAEEngine(self.pose_dim, self.hidden_size, self.n_layers, self.vq_components, self.n_frames,
                              beta=self.commitment_cost, dropout_prob=self.dropout_prob, n_pre_poses=self.n_pre_poses,
                              conditioned=self.autoencoder_conditioned, device=dev, seed=self.rng_seed,
                              quantizer=self.quantizer)
            self._engine = eng
            if getattr(self.decoder.decoder, "autoencoder_fixed_weight", False):
                eng.frozen = ["decoder.decoder.gru." + n for n, _ in self.decoder.decoder.gru.named_parameters()]
        # (re)home every trainable tensor into the flat buffer; cheap pointer check per call
        for name, _ in eng.layout:
            p = sd_params[name]
            v = eng.view(name)
            if p.data_ptr() != v.data_ptr():
                v.copy_(p.data)
                p.data = v
            g = eng.view(name, True)
            if p.grad is None or p.grad.data_ptr() != g.data_ptr():
                p.grad = g
        vq, bn = self.vq_layer, self.decoder.decoder.pre_linear[1]
        if self.quantizer == "ema":
            eng.vq_pre_w, eng.vq_pre_b = vq.pre_linear.weight.data, vq.pre_linear.bias.data
            eng.codebook, eng.ema_w, eng.ema_cs = vq._embedding.weight.data, vq._ema_w.data, vq._ema_cluster_size
        if eng.bn_rm.data_ptr() != bn.running_mean.data_ptr() or eng.bn_rv.data_ptr() != bn.running_var.data_ptr():
            eng.bn_rm, eng.bn_rv = bn.running_mean, bn.running_var
            eng._wstruct = None
        first, last = sd_params[eng.layout[0][0]], sd_params[eng.layout[-1][0]]
        cb = vq._embedding.weight if self.quantizer == "ema" else bn.running_mean
        self._engine_probe = (first, first.data_ptr(), last, last.data_ptr(), cb, cb.data_ptr())
        self._engine_bound = True
        return eng

    def _apply(self, fn, *a, **kw):            # .to() / .cuda() / .float() ...: tensors may be re-created
        self._engine_bound = False
        return super()._apply(fn, *a, **kw)

    def load_state_dict(self, *a, **kw):
        self._engine_bound = False
        return super().load_state_dict(*a, **kw)

    def set_dropout_masks(self, keep95, keep_in=None, keep_l0=None):
        """Explicit keep masks for the NEXT forward calls (parity tests): keep95 (T-1,B,D) for the inline
        Dropout(0.95) (:570), keep_in (T,B,D) for self.do (:957), keep_l0 (T-1,B,H) for the decoder GRU."""
        eng = self.engine()
        eng.set_masks(keep95.shape[1], keep95, keep_in, keep_l0)
        self._explicit_masks = True

    def use_random_masks(self):
        self._explicit_masks = False

    # ------------------------------------------------------------------ attention model (module-level path)
    def set_attention_masks(self, keep_in, keep_enc_inter, keep95_list, keep_l0_list=None):
        """Explicit keep masks for the NEXT forward of an autoencoder_att == "True" model (parity tests): keep_in (T,B,D) for
        self.do (:957), keep_enc_inter (T,B,2H) for the encoder GRU's inter-layer dropout, keep95_list: T-1 masks (B,D+H)
        for the decoder's inline Dropout(0.95), keep_l0_list: T-1 masks (B,H) for the decoder GRU's inter-layer dropout."""
        self._att_masks = (keep_in, keep_enc_inter, list(keep95_list), list(keep_l0_list) if keep_l0_list is not None else None)

    def _draw(self, shape, keep_prob, dev):
        if self._rng_counter is None or self._rng_counter.device != dev:
            self._rng_counter = torch.zeros(1, dtype=torch.int64, device=dev)
        return ops.keep_mask(torch.empty(shape, dtype=torch.uint8, device=dev), keep_prob, self.rng_seed, self._rng_counter)

    def _forward_attention(self, in_poses: torch.Tensor, out_poses: torch.Tensor):
        """Autoencoder_VQVAE.forward (:901-1085) with attention: every layer of the encoder is evaluated (the decoder reads
        encoder_outputs), the quantiser module runs on encoder_hidden[:L], then T-1 calls of the step-level decoder."""
        if not in_poses.is_cuda:
            raise RuntimeError("Autoencoder_VQVAE runs on the MI355X kernels only (no CPU fallback)")
        B, T, L, H = in_poses.shape[0], self.n_frames, self.n_layers, self.hidden_size
        dev, p = in_poses.device, self.dropout_prob
        dec = self.decoder.decoder
        x_tbd = in_poses.transpose(0, 1).contiguous()
        tgt = out_poses.transpose(0, 1).contiguous()
        keep_in = keep_inter = None
        if self._att_masks is not None:
            keep_in, keep_inter, k95, kl0 = self._att_masks
            dec.set_step_masks(k95, kl0)
            self._att_masks = None
        elif self.training and p > 0:
            keep_in = self._draw((T, B, self.pose_dim), 1.0 - p, dev)
            keep_inter = self._draw((T, B, 2 * H), 1.0 - p, dev) if L > 1 else None
        if not self.training:
            keep_in = keep_inter = None
        enc_out, enc_hidden = self.encoder(x_tbd, None, keep_in=keep_in, in_scale=1.0 / (1.0 - p) if keep_in is not None else 1.0,
                                           keep_inter=keep_inter)
        decoder_hidden = enc_hidden[:L]                                                   # :971-973
        loss_vq, quantized, perp, _ = self.vq_layer(decoder_hidden.contiguous())
        hidden = quantized
        enc_proj = dec.attn.project_encoder(enc_out)                                      # shared by the T-1 steps
        outputs = [tgt[0]]
        dec_in = tgt[0]
        for t in range(1, T):
            y, hidden, _ = self.decoder(None, dec_in, hidden, enc_out, None, enc_proj=enc_proj)
            outputs.append(y)
            dec_in = tgt[t] if t < self.n_pre_poses else y                                # :1049-1052
        return torch.stack(outputs).transpose(0, 1), quantized[:L], loss_vq, perp

    def forward(self, in_poses: torch.Tensor, out_poses: torch.Tensor, vq_layer_active: bool = False):
        if self.att_use:
            return self._forward_attention(in_poses, out_poses)
        eng = self.engine()
        in_poses = in_poses.contiguous()
        out_poses = out_poses.contiguous()
        B = in_poses.shape[0]
        if not self._explicit_masks:
            eng.draw_masks(B, self.training)
        if self.quantizer != "ema":
            y, first_hidden, loss_vq, perp = self._forward_staged(eng, in_poses, out_poses)
        elif self.training and torch.is_grad_enabled():
            y, first_hidden, loss_vq, perp = _VQVAEFn.apply(self.encoder.in_layer.weight, self, in_poses, out_poses)
        else:
            b = eng.forward(in_poses, out_poses, self.training)
            y, first_hidden = b["y"].clone(), b["quant"].clone()
            loss_vq, perp = eng.vq_scalars[0].clone(), eng.vq_scalars[1].clone()
        if self.training:
            self.decoder.decoder.pre_linear[1].num_batches_tracked += self.n_frames - 1   # one BN call per decode step
        return y.transpose(0, 1), first_hidden[: self.n_layers], loss_vq, perp

    def _forward_staged(self, eng, in_poses, out_poses):
        """encoder stage -> self.vq_layer (an nn.Module of HIP-backed autograd functions) -> decoder stage.  The
        quantiser's gradients accumulate into its slice of the engine's flat grad buffer (two uses of the codebook), so that
        slice is cleared here; everything else is overwritten by the stage backwards."""
        B = in_poses.shape[0]
        if self.training and torch.is_grad_enabled():
            eng.gflat[eng.q_off:eng.n_flat].zero_()
            hidden = _EncFn.apply(self.encoder.in_layer.weight, self, in_poses)
            loss_vq, quantized, perp, _ = self.vq_layer(hidden)
            y = _DecFn.apply(quantized, self, out_poses)
            return y, quantized, loss_vq, perp
        with torch.no_grad():
            b = eng.forward_encoder(in_poses, self.training)
            loss_vq, quantized, perp, _ = self.vq_layer(b["enc_hidden"])
            b["quant"].copy_(quantized.reshape(b["quant"].shape))
            eng.forward_decoder(out_poses, B, self.training)
            return b["y"].clone(), quantized.clone(), loss_vq, perp


def _rebind_grads(net, eng):
    for name, _ in eng.layout:      # in case zero_grad(set_to_none=True) dropped the views
        p = net.get_parameter(name)
        g = eng.view(name, True)
        if p.grad is None or p.grad.data_ptr() != g.data_ptr():
            p.grad = g


class _EncFn(torch.autograd.Function):
    """in_poses -> encoder_hidden[:L] (2,B,H) through VQVAEEngine.forward_encoder / backward_encoder (module-level path of
    the non-EMA quantisers)."""

    @staticmethod
    def forward(ctx, anchor, net, in_poses):
        eng = net._engine
        b = eng.forward_encoder(in_poses, True)
        ctx.net = net
        ctx.set_materialize_grads(False)
        ctx.save_for_backward(in_poses)
        return b["enc_hidden"].clone()

    @staticmethod
    def backward(ctx, g_hidden):
        net = ctx.net
        eng = net._engine
        (in_poses,) = ctx.saved_tensors
        B = in_poses.shape[0]
        b = eng.buffers(B)
        if g_hidden is None:
            b["gz"].zero_()
        else:
            b["gz"].copy_(g_hidden.reshape(b["gz"].shape))
        eng.backward_encoder(in_poses, B)
        _rebind_grads(net, eng)
        return None, None, None


class _DecFn(torch.autograd.Function):
    """(initial hidden (2,B,H), out_poses) -> y (T,B,D) through VQVAEEngine.forward_decoder / backward_decoder."""

    @staticmethod
    def forward(ctx, hidden, net, out_poses):
        eng = net._engine
        B = out_poses.shape[0]
        b = eng.buffers(B)
        b["quant"].copy_(hidden.reshape(b["quant"].shape))
        eng.forward_decoder(out_poses, B, True)
        ctx.net, ctx.B, ctx.hshape = net, B, hidden.shape
        ctx.set_materialize_grads(False)
        return b["y"].clone()

    @staticmethod
    def backward(ctx, gy):
        net, B = ctx.net, ctx.B
        eng = net._engine
        b = eng.buffers(B)
        if gy is None:
            b["dy"].zero_()
        else:
            b["dy"].copy_(gy)
        eng.backward_decoder(B)
        return b["dh_init"].clone().reshape(ctx.hshape), None, None


class _VQVAEFn(torch.autograd.Function):
    """Training-mode forward/backward of the whole module as ONE autograd node: the backward writes every parameter
    gradient straight into the engine's flat grad buffer (the parameters' .grad are views of it)."""

    @staticmethod
    def forward(ctx, anchor, net: Autoencoder_VQVAE, in_poses, out_poses):
        eng = net._engine
        b = eng.forward(in_poses, out_poses, True)
        ctx.net, ctx.B = net, in_poses.shape[0]
        ctx.set_materialize_grads(False)
        ctx.save_for_backward(in_poses)
        perp = eng.vq_scalars[1].clone()
        ctx.mark_non_differentiable(perp)
        return b["y"].clone(), b["quant"].clone(), eng.vq_scalars[0].clone(), perp

    @staticmethod
    def backward(ctx, gy, g_first_hidden, g_loss_vq, g_perp):
        net, B = ctx.net, ctx.B
        eng = net._engine
        (in_poses,) = ctx.saved_tensors
        b = eng.buffers(B)
        if gy is None:
            b["dy"].zero_()
        else:
            b["dy"].copy_(gy)
        gl = g_loss_vq.reshape(1).contiguous() if g_loss_vq is not None else torch.zeros(1, device=eng.device)
        if g_first_hidden is not None:
            raise NotImplementedError("gradient through decoder_first_hidden is not used by the reference's losses")
        eng.backward(in_poses, B, gl)
        _rebind_grads(net, eng)
        return None, None, None, None


class VectorQuantizerEMA(nn.Module):
    """The EMA quantiser variant of reference :1713-1812 (unused by the reference's model): input (2,B,H) is
    hstack-ed per sample to (B,2H), `pre_lin` is applied IN the graph (it does receive gradients), loss and
    straight-through use the projected input, output is the row-major reinterpretation `reshape(q, (2, B, -1))` (:1810)."""

    def __init__(self, num_embeddings: int, embedding_dim: int, commitment_cost: float, decay: float, epsilon: float = 1e-5):
        super().__init__()
        self._embedding_dim, self._num_embeddings = embedding_dim, num_embeddings
        self.pre_lin = nn.Linear(embedding_dim, embedding_dim)
        self._embedding = nn.Embedding(num_embeddings, embedding_dim)
        self._embedding.weight.data.normal_()
        self._commitment_cost = commitment_cost
        self.register_buffer("_ema_cluster_size", torch.zeros(num_embeddings))
        self._ema_w = nn.Parameter(torch.Tensor(num_embeddings, embedding_dim))
        self._ema_w.data.normal_()
        self._decay, self._epsilon = decay, epsilon
        for p in (self._ema_w, self._embedding.weight):
            p.requires_grad_(False)

    def forward(self, inputs: torch.Tensor):
        from .. import functional as Fn
        x = torch.hstack((inputs[0], inputs[1]))                       # layout only (:1751)
        zp = Fn.linear(x, self.pre_lin.weight, self.pre_lin.bias)      # in-graph projection (:1754)
        loss, quant, perp, enc = _VQFn.apply(zp, self, False)
        return loss, torch.reshape(quant, (2, quant.shape[0], -1)).contiguous(), perp, enc


class VQ_Payam(nn.Module):
    """Non-EMA quantiser of reference :1088-1179: no pre_linear in the path, loss = q_latent + beta * e_latent,
    the codebook learns by gradient."""

    def __init__(self, num_embeddings: int, embedding_dim: int, commitment_cost: float):
        super().__init__()
        self._embedding_dim, self._num_embeddings = embedding_dim, num_embeddings
        self.pre_linear = nn.Linear(embedding_dim, embedding_dim)      # present in the state_dict, unused (:1099,1123)
        self._embedding = nn.Embedding(num_embeddings, embedding_dim)
        self._embedding.weight.data.normal_()                           # :1104
        self._commitment_cost = commitment_cost

    def embedding_grad(self, what: bool) -> None:
        for p in self._embedding.parameters():
            p.requires_grad = what

    def forward(self, inputs: torch.Tensor):
        return _VQPlainFn.apply(inputs, self._embedding.weight, self)

    def assign(self, inputs: torch.Tensor) -> torch.Tensor:
        """Code indices only, int64 (N,): what callers take as `argmax(encodings, 1)` of forward() (lmdb_data_loader.py:1274-1281)
        -- argmin_k ||x - W_k||^2 on the raw rows (this quantiser has no projection in its path, :1123)."""
        z = inputs.contiguous().view(-1, self._embedding_dim)
        W = self._embedding.weight.data.contiguous()
        idx, _, _, _ = ops.vq_assign(z, None, W, ops.vq_code_sqnorm(W), want_quantized=False)
        return idx


class _VQPlainFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, inputs, weight, mod):
        E, K = mod._embedding_dim, mod._num_embeddings
        z = inputs.contiguous().view(-1, E)
        N = z.shape[0]
        W = weight.data.contiguous()
        wsq = ops.vq_code_sqnorm(W)
        idx, quant, _, sse = ops.vq_assign(z, z, W, wsq)
        stats = ops.vq_stats(idx, z, K)
        beta = mod._commitment_cost
        scalars = ops.vq_ema_update(stats, sse, None, None, None, None, N, N, E, K, 1.0 + beta, 0.0, 0.0, False)
        encodings = torch.zeros((N, K), dtype=torch.float32, device=z.device)
        encodings.scatter_(1, idx.unsqueeze(1), 1.0)
        ctx.set_materialize_grads(False)
        ctx.save_for_backward(z, quant, stats, W)
        ctx.beta, ctx.shape, ctx.N = beta, inputs.shape, N
        ctx.mark_non_differentiable(encodings)
        return scalars[0].clone(), quant.view(inputs.shape), scalars[1].clone(), encodings

    @staticmethod
    def backward(ctx, g_loss, g_quant, g_perp, g_enc):
        z, quant, stats, W = ctx.saved_tensors
        gl = g_loss.reshape(1).contiguous() if g_loss is not None else None
        gq = g_quant.contiguous().view(z.shape) if g_quant is not None else None
        gz = ops.vq_bwd(gq, gl, z, quant, None, ctx.beta)              # e_latent term + straight-through (:1158,1166)
        gw = ops.vq_codebook_grad(stats, W, gl, ctx.N) if gl is not None else None      # q_latent term (:1159)
        return gz.view(ctx.shape), gw, None


class _SoftAssignFn(torch.autograd.Function):
    """(flat (N,E), logvar (N,K), codebook (K,E)) -> (probs (N,K), perplexity): distances + the soft assignment
    probabilities of VQ_Payam_GSSoft.soft_prob (reference :1349-1372,1396-1411)."""

    @staticmethod
    def forward(ctx, flat, logvar, weight):
        flat, logvar, W = flat.contiguous(), logvar.contiguous(), weight.contiguous()
        dots = ops.linear_fwd(flat, W)
        probs, dist, perp = ops.vq_soft_fwd(flat, dots, logvar, ops.vq_code_sqnorm(W))
        ctx.save_for_backward(flat, logvar, W, probs, dist)
        ctx.set_materialize_grads(False)
        ctx.mark_non_differentiable(perp)
        return probs, perp

    @staticmethod
    def backward(ctx, dprobs, _dperp):
        if dprobs is None:
            return None, None, None
        flat, logvar, W, probs, dist = ctx.saved_tensors
        K, E = W.shape
        dd, dlv, rowsum = ops.vq_soft_bwd(probs, dprobs.contiguous(), dist, logvar)
        dflat = ops.rowscale_combine(flat, rowsum, ops.linear_bwd_data(dd, W))       # 2 f sum_k dd - 2 dd W
        tw, colsum = ops.linear_bwd_weight(dd, flat, K, E, want_bias=True)            # dd^T f, column sums of dd
        dW = ops.rowscale_combine(W, colsum, tw)                                      # 2 W sum_n dd - 2 dd^T f
        return dflat, dlv, dW


class _ProbsCodebookFn(torch.autograd.Function):
    """q = probs @ W (reference :1417-1419)."""

    @staticmethod
    def forward(ctx, probs, weight):
        probs, W = probs.contiguous(), weight.contiguous()
        ctx.save_for_backward(probs, W)
        ctx.set_materialize_grads(False)
        return ops.linear_bwd_data(probs, W)

    @staticmethod
    def backward(ctx, dq):
        if dq is None:
            return None, None
        probs, W = ctx.saved_tensors
        K, E = W.shape
        dq = dq.contiguous()
        dprobs = ops.linear_fwd(dq, W)
        dW, _ = ops.linear_bwd_weight(probs, dq, K, E, want_bias=False)
        return dprobs, dW


class _STEFn(torch.autograd.Function):
    """inputs + (q - inputs).detach(): value from the kernel, gradient to `inputs` only (:1431)."""

    @staticmethod
    def forward(ctx, z, q):
        return ops.ste(z.contiguous(), q.contiguous())

    @staticmethod
    def backward(ctx, g):
        return g, None


class VQ_Payam_GSSoft(nn.Module):
    """The soft quantiser the reference's Autoencoder_VQVAE ships with (:816-820; class :1304-1438): mean_layer ->
    distances -> probabilities with a learnt per-code smoothness (logvar_layer) -> q = probs @ W; both latent losses;
    `encodings` are the soft probabilities.  Same parameters / state_dict keys (pre_linear exists but is unused,
    :1389).  Standalone module (the fused Autoencoder_VQVAE engine uses the EMA quantiser of the north star)."""

    def __init__(self, num_embeddings: int, embedding_dim: int, commitment_cost: float):
        super().__init__()
        self._embedding_dim, self._num_embeddings = embedding_dim, num_embeddings
        self.pre_linear = nn.Linear(embedding_dim, embedding_dim)
        self._embedding = nn.Embedding(num_embeddings, embedding_dim)
        self._embedding.weight.data.normal_()
        self._commitment_cost = commitment_cost
        self.mean_layer = nn.Linear(embedding_dim, embedding_dim)
        self.logvar_layer = nn.Linear(embedding_dim, num_embeddings)

    def embedding_grad(self, what: bool) -> None:
        for param in self._embedding.parameters():
            param.requires_grad = what

    def forward(self, inputs: torch.Tensor):
        if not inputs.is_cuda:
            raise RuntimeError("VQ_Payam_GSSoft runs on the MI355X kernels only (no CPU fallback)")
        E = self._embedding_dim
        x = inputs.contiguous().view(-1, E)
        flat = Fn.linear(x, self.mean_layer.weight, self.mean_layer.bias)
        logvar = Fn.linear(flat, self.logvar_layer.weight, self.logvar_layer.bias)
        probs, perplexity = _SoftAssignFn.apply(flat, logvar, self._embedding.weight)
        q = _ProbsCodebookFn.apply(probs, self._embedding.weight)
        e_latent = Fn.mse_loss(x, q.detach())                                  # gradient to the input only  (:1424)
        q_latent = Fn.mse_loss(q, x.detach())                                  # gradient to probs / codebook (:1425)
        loss = q_latent + self._commitment_cost * e_latent                     # :1427 (scalar glue)
        quantized = _STEFn.apply(x, q).view(inputs.shape)
        return loss, quantized, perplexity[0], probs

    def assign(self, inputs: torch.Tensor) -> torch.Tensor:
        """Code indices only, int64 (N,): `argmax(encodings, 1)` of forward() -- the soft probabilities' mode -- which is what
        the sentence-level dataset takes from whichever quantiser the checkpoint carries (lmdb_data_loader.py:1274-1281).
        mean_layer -> logvar_layer, distances -> probabilities, no autograd graph, no state change."""
        with torch.no_grad():
            x = inputs.contiguous().view(-1, self._embedding_dim)
            flat = ops.linear_fwd(x, self.mean_layer.weight.data, self.mean_layer.bias.data)
            logvar = ops.linear_fwd(flat, self.logvar_layer.weight.data, self.logvar_layer.bias.data)
            W = self._embedding.weight.data.contiguous()
            probs, _, _ = ops.vq_soft_fwd(flat, ops.linear_fwd(flat, W), logvar, ops.vq_code_sqnorm(W))
            return torch.argmax(probs, dim=1)


class VectorQuantizer(nn.Module):
    """Reference `VectorQuantizer` (:1584-1710): its forward returns on its first statement (`return loss, inputs,
    perplexity_vq, encodings` with zero scalars, :1617) -- an identity pass-through whose parameters (`pre_lin`,
    `_embedding` ~ U(-1/K, 1/K)) never receive gradients.  Reproduced as is (no kernels involved)."""

    def __init__(self, num_embeddings: int, embedding_dim: int, commitment_cost: float):
        super().__init__()
        self._embedding_dim, self._num_embeddings = embedding_dim, num_embeddings
        self.pre_lin = nn.Linear(embedding_dim, embedding_dim)
        self._embedding = nn.Embedding(num_embeddings, embedding_dim)
        self._embedding.weight.data.uniform_(-1 / num_embeddings, 1 / num_embeddings)
        self._commitment_cost = commitment_cost

    def forward(self, inputs: torch.Tensor):
        return torch.tensor(0), inputs, torch.tensor(0), torch.tensor(0)
