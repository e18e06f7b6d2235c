"""Part d: text -> gesture-code seq2seq -- mirror of `scripts/model/text2embedding_model.py` (reference :46-746) on the
MI355X kernels: `EncoderRNN` (word Embedding -> length-packed bi-GRU -> sum of directions), `BahdanauAttnDecoderRNN`
(code Embedding + Dropout(0.5) -> Linear + BatchNorm1d + ReLU -> GRU(L) -> Linear(H->K) logits), `Generator`,
`text2embedding_model` (S-1 decode steps with greedy argmax feedback).  Same class names, ctor/forward signatures and
state_dict keys.  Every operator is a HIP kernel behind include/g2v.h, chained by small autograd nodes
(gesture2vec_amd/functional.py).

Module-level switches of the reference (:40-43) are all False here: `use_TCN = True` as checked in makes the
reference's forward crash, `audio_context`, `noisy`, `GPT3_embedding_active` select out-of-scope
encoders.  Scope: text2_embedding_discrete == "True"; autoencoder_att == "False" (config/seq2seq.yml:27) and "True"
(config/seq2seqtxt.yml:37, Bahdanau attention `Attn` :138-198 through g2v_attn_fwd / g2v_attn_bwd)."""
from __future__ import annotations

import math
from typing import List, Optional

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import functional as Fn
from .. import ops
from ..rollout_t2e import CodeDecoderRollout, RolloutSpec, decoder_params
from .Autoencoder_VQVAE_model import Attn, _GRUParams

debug = False
noisy = False
audio_context = False
use_TCN = False
GPT3_embedding_active = False


class EncoderRNN(nn.Module):
    def __init__(self, input_size: int, embed_size: int, hidden_size: int, n_layers: int = 1, dropout: float = 0.5,
                 pre_trained_embedding: np.ndarray = None):
        super().__init__()
        self.input_size, self.hidden_size, self.embed_size = input_size, hidden_size, embed_size
        self.n_layers, self.dropout = n_layers, dropout
        if pre_trained_embedding is not None:
            assert pre_trained_embedding.shape[0] == input_size
            assert pre_trained_embedding.shape[1] == embed_size
            self.embedding = nn.Embedding.from_pretrained(torch.FloatTensor(pre_trained_embedding), freeze=False)
        else:
            self.embedding = nn.Embedding(input_size, embed_size)
        self.gru = _GRUParams(embed_size, hidden_size, n_layers, dropout=dropout, bidirectional=True)
        self.do_flatten_parameters = False
        self.packed_inputs = True        # layer 0 on the packed positions when the lengths arrive on the host (forward)
        self._len_cache = None

    def forward(self, input_seqs: torch.Tensor, input_lengths: torch.Tensor, hidden=None, n_layers_needed: Optional[int] = None,
                keep_inter: Optional[torch.Tensor] = None, want_outputs: bool = True):
        """(Tw,B) ids + lengths (sorted descending) -> outputs (Tw,B,H) [sum of the LAST evaluated layer's directions],
        hidden (2*layers_evaluated, B, H) ordered l0f,l0b,l1f,l1b.  `n_layers_needed` lets the caller skip layers whose
        result it never reads (the attention-free decoder only needs layer 0).  `keep_inter` (Tw,B,2H) uint8 is the
        keep mask of nn.GRU's inter-layer dropout (training, dropout > 0; fused into the next layer's input product;
        ATen draws it on the packed rows, padded rows are zero either way)."""
        if hidden is not None:
            raise NotImplementedError("non-zero initial hidden state")
        Tw, B = input_seqs.shape
        H = self.hidden_size
        L = self.n_layers if n_layers_needed is None else min(self.n_layers, n_layers_needed)
        dev = input_seqs.device
        # (host lengths: the device copy and the packing tables of the LAST set of lengths are cached -- a hipGraph capture of
        #  the step, whose warm-up ran with the same lengths, then finds them without a host-to-device copy)
        cache = None
        if not input_lengths.is_cuda:
            key = (tuple(int(v) for v in input_lengths.tolist()), Tw, B, str(dev))
            cache = self._len_cache if (self._len_cache is not None and self._len_cache["key"] == key) else None
            if cache is None:
                cache = self._len_cache = {"key": key, "lengths": input_lengths.to(device=dev, dtype=torch.int32).contiguous()}
            lengths = cache["lengths"]
        else:
            lengths = input_lengths.to(device=dev, dtype=torch.int32).contiguous()
        # Round 5: what pack_padded_sequence does for the reference (:127-131).  With the lengths known on the HOST (the reference
        # hands them over as a CPU tensor) and sorted descending, layer 0's word embeddings, input projections and their
        # gradients run over the sum(lengths) positions inside the sentences only, step-major (the first n_t rows of step t):
        # 40 % fewer rows than the padded (Tw, B) grid at lengths U{4..20}.  The recurrent kernels read / write those packed
        # arrays through per-step row offsets (include/g2v.h: g2v_gru_dir.gi_row_off); states keep the (Tw,B,H) layout.
        packed = None
        if self.packed_inputs and cache is not None and ops.gru_packed_ok(Tw, B, H):
            if "packed" not in cache:
                cache["packed"] = None
                ln = list(cache["key"][0])
                if ln and all(ln[k] >= ln[k + 1] for k in range(len(ln) - 1)) and 1 <= ln[-1] and ln[0] <= Tw:
                    n_t = [sum(1 for v in ln if v > t) for t in range(Tw)]
                    row_off = [0] * Tw
                    for t in range(1, Tw):
                        row_off[t] = row_off[t - 1] + n_t[t - 1]
                    if sum(n_t) < Tw * B:
                        rows = torch.cat([torch.arange(t * B, t * B + n_t[t]) for t in range(Tw) if n_t[t] > 0]).to(dev)
                        cache["packed"] = (Tw, B, row_off, rows)
            packed = cache["packed"]
        ids_flat = input_seqs.contiguous().view(-1)
        if packed is not None:
            ids_flat = ids_flat.index_select(0, packed[3])
        # many rows per vocabulary entry (B = 4096: 49 k packed rows, 3863 words): layer 0's input projections commute with the
        # gather -- project the TABLE, gather the projected rows (Fn.EmbedProjectPairFn); else embed, then project
        via_table = ids_flat.numel() >= 2 * self.embedding.weight.shape[0]
        x = None if via_table else Fn.EmbeddingFn.apply(self.embedding.weight, ids_flat, None, 1.0)   # (Tw*B, E), or (sum(lengths), E) packed
        hiddens, layer_in = [], x
        out_f = out_b = None
        keep, scale = None, 1.0
        for l in range(L):
            g = self.gru
            pk = packed[:3] if (packed is not None and l == 0) else None
            gather = None
            if l == 0 and via_table and pk is not None and ops.gru_gather_ok(Tw, B, H, 2):
                # (round 6) ... and where the recurrent kernel can gather (the W_hh-resident forward), the projected TABLES go in
                # as they are: the two (positions x 3H) gathers, 118 MB each at B = 4096, are never written
                gis = list(Fn.LinearPairFn.apply(self.embedding.weight, g.weight_ih_l0, g.bias_ih_l0, g.weight_ih_l0_reverse,
                                                 g.bias_ih_l0_reverse))
                gather = ids_flat
            elif l == 0 and via_table:
                gis = list(Fn.EmbedProjectPairFn.apply(self.embedding.weight, ids_flat, g.weight_ih_l0, g.bias_ih_l0,
                                                       g.weight_ih_l0_reverse, g.bias_ih_l0_reverse))
            else:                   # both directions' projections as one Function (the inter-layer dropout applied once)
                gis = list(Fn.LinearPairFn.apply(layer_in, getattr(g, f"weight_ih_l{l}"), getattr(g, f"bias_ih_l{l}"),
                                                 getattr(g, f"weight_ih_l{l}_reverse"), getattr(g, f"bias_ih_l{l}_reverse"),
                                                 keep, scale))
            if pk is None:
                gis = [gi.view(Tw, B, 3 * H) for gi in gis]
            # both directions of the layer in one launch (each way)
            out_f, hn_f, out_b, hn_b = Fn.GRUBiDirFn.apply(
                gis[0], gis[1], getattr(g, f"weight_hh_l{l}"), getattr(g, f"bias_hh_l{l}"),
                getattr(g, f"weight_hh_l{l}_reverse"), getattr(g, f"bias_hh_l{l}_reverse"), lengths, pk, gather)
            hiddens += [hn_f, hn_b]
            if l + 1 < L:
                cat = torch.cat([out_f, out_b], dim=2).view(Tw * B, 2 * H)       # layout only
                layer_in = cat
                if self.training and self.dropout > 0 and keep_inter is not None:
                    keep, scale = keep_inter.contiguous().view(Tw * B, 2 * H), 1.0 / (1.0 - self.dropout)
        # :133-135 (want_outputs=False: a caller that reads the final states only -- the sum is a 200 MB pass at B = 4096)
        outputs = Fn.SumHalvesFn.apply(out_f, out_b) if want_outputs else None
        return outputs, torch.stack(hiddens)


class BahdanauAttnDecoderRNN(nn.Module):
    def __init__(self, args, input_size: int, hidden_size: int, output_size: int, n_layers: int = 1,
                 dropout_p: float = 0.1, discrete_representation: bool = False, speaker_model=None):
        super().__init__()
        self.hidden_size, self.output_size, self.n_layers, self.dropout_p = hidden_size, output_size, n_layers, dropout_p
        self.discrete_representation, self.speaker_model = discrete_representation, speaker_model
        if not discrete_representation:
            raise NotImplementedError("text2_embedding_discrete == 'False' is outside the accelerated hot path")
        if speaker_model:
            raise NotImplementedError("speaker embedding is outside the accelerated hot path")
        self.embedding = nn.Embedding(output_size, hidden_size)
        self.dropout = nn.Dropout(0.5)
        self.att_use = args.autoencoder_att == "True"
        if self.att_use:
            self.attn = Attn(hidden_size)
        linear_input_size = 2 * hidden_size if self.att_use else hidden_size          # :277-281
        self.pre_linear = nn.Sequential(nn.Linear(linear_input_size, hidden_size), nn.BatchNorm1d(hidden_size), nn.ReLU(inplace=True))
        self.gru = _GRUParams(hidden_size, hidden_size, n_layers, dropout=dropout_p)
        self.out = nn.Linear(hidden_size, output_size)
        self.softmax = nn.Softmax(dim=1)
        self.do_flatten_parameters = False

    def freeze_attn(self) -> None:
        for param in self.attn.parameters():
            param.requires_grad = False

    def forward(self, motion_input, last_hidden, encoder_outputs=None, vid_indices=None, keep_emb=None, keep_l0=None,
                enc_proj=None):
        """One decode step: code ids (B,) + hidden (L,B,H) [+ encoder outputs (T,B,H) with attention] -> logits (B,K),
        new hidden (L,B,H), attention weights (B,1,T) or None.  `enc_proj` = Attn.project_encoder(encoder_outputs)
        may be passed in so that the S-1 steps share it."""
        B = motion_input.shape[0]
        H, L = self.hidden_size, self.n_layers
        training = self.training
        e = Fn.EmbeddingFn.apply(self.embedding.weight, motion_input, keep_emb if training else None, 2.0)   # Dropout(0.5)
        attn_weights = None
        if self.att_use:
            context, w = self.attn.context(last_hidden[-1], encoder_outputs, enc_proj)     # :353-359
            e = torch.cat((e, context), 1)                                                 # :362-364 (layout only)
            attn_weights = w.unsqueeze(1)
        lin, bn = self.pre_linear[0], self.pre_linear[1]
        u = Fn.linear(e, lin.weight, lin.bias)
        a = Fn.BatchNormReluFn.apply(u, bn.weight, bn.bias, bn.running_mean, bn.running_var, training, True)
        if training:
            bn.num_batches_tracked += 1
        new_h, layer_in, keep, scale = [], a, None, 1.0
        for l in range(L):
            g = self.gru
            gi = Fn.linear(layer_in, getattr(g, f"weight_ih_l{l}"), getattr(g, f"bias_ih_l{l}"), keep=keep, scale=scale)
            _, h_n = Fn.GRUDirFn.apply(gi.view(1, B, 3 * H), getattr(g, f"weight_hh_l{l}"), getattr(g, f"bias_hh_l{l}"),
                                       last_hidden[l], None, False)
            new_h.append(h_n)
            layer_in = h_n
            if training and self.dropout_p > 0 and keep_l0 is not None:
                keep, scale = keep_l0, 1.0 / (1.0 - self.dropout_p)      # nn.GRU inter-layer dropout, fused into the next Linear
        logits = Fn.linear(new_h[-1], self.out.weight, self.out.bias)
        return logits, torch.stack(new_h), attn_weights


class Generator(nn.Module):
    def __init__(self, args, motion_dim: int, discrete_representation: bool = False, speaker_model=None):
        super().__init__()
        self.output_size = motion_dim
        self.n_layers = args.n_layers
        self.discrete_representation = discrete_representation
        self.decoder = BahdanauAttnDecoderRNN(args, input_size=motion_dim, hidden_size=args.hidden_size,
                                              output_size=self.output_size, n_layers=self.n_layers,
                                              dropout_p=args.dropout_prob, discrete_representation=discrete_representation,
                                              speaker_model=speaker_model)

    def forward(self, z, motion_input, last_hidden, encoder_output, vid_indices=None, **kw):
        assert z is None
        return self.decoder(motion_input, last_hidden, encoder_output, vid_indices, **kw)


class text2embedding_model(nn.Module):
    def __init__(self, args, pose_dim: int, n_frames: int, n_words: int, word_embed_size: int, word_embeddings,
                 speaker_model=None):
        super().__init__()
        self.text2_embedding_discrete = args.text2_embedding_discrete == "True"
        if not self.text2_embedding_discrete:
            raise NotImplementedError("text2_embedding_discrete == 'False' is outside the accelerated hot path")
        self.n_layers = args.n_layers
        pose_dim = int(args.autoencoder_vq_components)
        self.encoder = EncoderRNN(n_words, word_embed_size, args.hidden_size, args.n_layers, dropout=args.dropout_prob,
                                  pre_trained_embedding=word_embeddings)
        self.decoder = Generator(args, pose_dim, discrete_representation=True, speaker_model=speaker_model)
        self.n_frames, self.n_pre_poses, self.pose_dim = n_frames, args.n_pre_poses, pose_dim
        self.sentence_frame_length = args.sentence_frame_length
        self.dropout_prob = float(args.dropout_prob)
        self._masks = None
        self._rng_counter = None
        self.rng_seed = 0
        # training: the S-1 decode steps as one autograd node (gesture2vec_amd/rollout_t2e.py); False = one node per
        # operator and step (the cross-check, and what inference uses either way)
        self.fused_rollout = True
        # Trainers set this to a list for the duration of their forward (train_iter_text2embedding, GraphedText2EmbeddingStep): the
        # decoder rollout then leaves BatchNorm's running statistics alone and `commit_bn_running_stats()` -- called behind the
        # backward -- applies them from the saved batch statistics unless a persistent kernel of the iteration latched a fault.
        # None (a plain forward in train mode): committed step by step inside the rollout, as nn.BatchNorm1d does.
        self.deferred_bn = None

    def commit_bn_running_stats(self) -> None:
        """Apply the decoder BatchNorm's running-statistics updates that the forward passes since `deferred_bn = []` held back
        (reference: nn.BatchNorm1d updates them inside every decode step, text2embedding_model.py:286-290 under :701-744).  One
        launch per held-back rollout, stream-ordered, a no-op on the device while the persistent kernels' fault latch is set."""
        pend, self.deferred_bn = self.deferred_bn, None
        if not pend:
            return
        bn = self.decoder.decoder.pre_linear[1]
        for sm, si, stride, steps, B, H in pend:
            ops.bn_running_update_invstd(sm, si, stride, bn.running_mean, bn.running_var, steps, H, B)

    def set_dropout_masks(self, mask_emb, mask_dec_l0=None, mask_enc_l0=None):
        """Explicit keep masks for the next training forward (parity tests): (S-1,B,H) for the code-embedding dropout and
        the decoder GRU inter-layer dropout, (Tw,B,2H) for the encoder GRU inter-layer dropout (attention only)."""
        self._masks = (mask_emb, mask_dec_l0, mask_enc_l0)

    def _draw(self, shape, keep_prob, dev):
        if self._rng_counter is None or self._rng_counter.device != dev:
            self._rng_counter = torch.zeros(1, dtype=torch.int64, device=dev)
        return ops.keep_mask(torch.empty(shape, dtype=torch.uint8, device=dev), keep_prob, self.rng_seed, self._rng_counter)

    def _draw_many(self, want, dev):
        """The masks of one forward, each at its own Philox offset (counter, counter + 1, ...: what consecutive _draw calls use),
        and ONE counter bump behind them instead of one per mask."""
        if self._rng_counter is None or self._rng_counter.device != dev:
            self._rng_counter = torch.zeros(1, dtype=torch.int64, device=dev)
        masks = [ops.keep_mask_at(torch.empty(shape, dtype=torch.uint8, device=dev), kp, self.rng_seed, self._rng_counter, k)
                 for k, (shape, kp) in enumerate(want)]
        ops.counter_add(self._rng_counter, len(want))
        return masks

    def forward(self, in_text, in_lengths, in_audio, poses, GPT3_embeddings, vid_indices):
        if not in_text.is_cuda:
            raise RuntimeError("text2embedding_model runs on the MI355X kernels only (no CPU fallback)")
        if vid_indices is not None and self.training:
            raise NotImplementedError("vid_indices is the reference's inference branch (:685-692): call it in eval mode")
        dev = in_text.device
        ids = in_text.transpose(0, 1).contiguous()                 # (Tw,B)
        cod = poses.transpose(0, 1).contiguous().to(torch.int64)   # (S,B)
        S_model = self.sentence_frame_length // self.n_frames
        B, K, H, L = cod.shape[1], self.pose_dim, self.encoder.hidden_size, self.n_layers
        training = self.training
        att = self.decoder.decoder.att_use
        Tw = ids.shape[0]
        mask_emb = mask_l0 = mask_enc = None
        if training:
            if self._masks is not None:
                mask_emb, mask_l0, mask_enc = self._masks
            else:
                want = [((S_model - 1, B, H), 0.5)]
                if self.dropout_prob > 0:
                    want.append(((S_model - 1, B, H), 1.0 - self.dropout_prob))
                    if att and L > 1:
                        want.append(((Tw, B, 2 * H), 1.0 - self.dropout_prob))
                mask_emb, mask_l0, mask_enc = (self._draw_many(want, dev) + [None, None])[:3]
        if att:
            # attention reads encoder_outputs = sum of the LAST layer's directions (:133-135): every layer is evaluated
            enc_out, enc_hidden = self.encoder(ids, in_lengths, None, keep_inter=mask_enc)
            enc_proj = self.decoder.decoder.attn.project_encoder(enc_out)
        else:
            # the attention-free decoder reads only encoder_hidden[:L] = the layer-0 final states (:667-669)
            _, enc_hidden = self.encoder(ids, in_lengths, None, n_layers_needed=1, want_outputs=False)
            enc_out = enc_proj = None
        # (a slice that keeps everything still costs a zero fill + a copy in its backward)
        hidden = enc_hidden[:L] if enc_hidden.shape[0] != L else enc_hidden
        if training and vid_indices is None and self.fused_rollout and S_model > 1:
            dec = self.decoder.decoder
            bn = dec.pre_linear[1]
            spec = RolloutSpec(cod, S_model - 1, self.n_pre_poses, L, att, self.dropout_prob, mask_emb, mask_l0,
                               bn.running_mean, bn.running_var, defer_bn=self.deferred_bn)
            full, attw = CodeDecoderRollout.apply(hidden, enc_out, spec, *decoder_params(dec))   # (S,B,K), slot 0 = one-hot :676-677
            bn.num_batches_tracked += S_model - 1
            attentions_list = [attw[t].unsqueeze(1) for t in range(S_model - 1)] if att else []
            outputs = full.transpose(0, 1)                                            # (B,S,K), a view
            # the step-major array behind the view: train_iter_text2embedding takes its loss on it directly (the reference's
            # outputs[:, 1:, :].reshape(-1, K) on the view is a 50 MB strided copy each way at B = 4096)
            outputs._g2v_step_major = full
            outputs._g2v_targets = cod          # (S,B): the loss's targets in the same step-major order (train_seq2seq._code_loss_backward)
            return outputs, attentions_list
        outs: List[torch.Tensor] = [F.one_hot(cod[0], K).to(torch.float32)]          # :676-677
        dec_in = cod[0]
        attentions_list = []
        if vid_indices is not None:
            # inference (:685-692): one extra step fed with vid_indices; its logits replace outputs[0], its argmax is fed on
            logits, hidden, _ = self.decoder(None, vid_indices.to(torch.int64).contiguous(), hidden, enc_out, vid_indices,
                                             keep_emb=None, keep_l0=None, enc_proj=enc_proj)
            outs[0] = logits
            dec_in = ops.argmax_rows(logits.detach().contiguous())
        for t in range(1, S_model):                                                    # :701-744
            ke = mask_emb[t - 1].contiguous() if training else None
            kl = mask_l0[t - 1].contiguous() if (training and mask_l0 is not None) else None
            logits, hidden, attn_w = self.decoder(None, dec_in, hidden, enc_out, None, keep_emb=ke, keep_l0=kl,
                                                  enc_proj=enc_proj)
            if att:
                attentions_list.append(attn_w)
            outs.append(logits)
            dec_in = cod[t] if t < self.n_pre_poses else ops.argmax_rows(logits.detach().contiguous())
        return torch.stack(outs).transpose(0, 1), attentions_list


# ---- the "*_New" tutorial-style classes (reference :754-1002; imported by train_text2embedding.py:57, never
# instantiated there).  Same parameters / state_dict keys / forward semantics on the HIP operators. ------------------------
class _GRULayer1(nn.Module):
    """Parameters of a 1-layer nn.GRU (optionally bidirectional) under nn.GRU's own names."""

    def __init__(self, input_size: int, hidden_size: int, bidirectional: bool):
        super().__init__()
        self.input_size, self.hidden_size, self.bidirectional = input_size, hidden_size, bidirectional
        k = 1.0 / math.sqrt(hidden_size)
        for suf in ("", "_reverse") if bidirectional else ("",):
            for name, shape in (("weight_ih_l0", (3 * hidden_size, input_size)), ("weight_hh_l0", (3 * hidden_size, hidden_size)),
                                ("bias_ih_l0", (3 * hidden_size,)), ("bias_hh_l0", (3 * hidden_size,))):
                self.register_parameter(name + suf, nn.Parameter(torch.empty(*shape).uniform_(-k, k)))


class EncoderRNN_New(nn.Module):
    """Embedding(input_size, 300) -> 1-layer bidirectional nn.GRU that the reference feeds ONE time step per call with the
    carried hidden state (:953-955): both "directions" therefore run forward in time (two independent GRUs with the
    l0 / l0_reverse weights).  `forward(input (B,), hidden (2,B,H))` is that single step; `run(ids (Tw,B))` is the whole
    loop as two full-sequence kernels."""

    def __init__(self, input_size: int, hidden_size: int, n_layer: int = 2, pre_trained_embedding=None):
        super().__init__()
        if n_layer != 1:
            raise NotImplementedError("text2embedding_model_New builds EncoderRNN_New with n_layer = 1 (:918)")
        self.hidden_size, self.n_layer, self.embed_size = hidden_size, n_layer, 300
        if pre_trained_embedding is not None:
            assert pre_trained_embedding.shape[0] == input_size and pre_trained_embedding.shape[1] == self.embed_size
            self.embedding = nn.Embedding.from_pretrained(torch.FloatTensor(pre_trained_embedding), freeze=False)
        else:
            self.embedding = nn.Embedding(input_size, self.embed_size)
        self.gru = _GRULayer1(self.embed_size, hidden_size, bidirectional=True)

    def run(self, ids_tb: torch.Tensor, hidden: Optional[torch.Tensor] = None):
        """(Tw,B) ids -> (outputs (Tw,B,2H), hidden (2,B,H)): every time step of the reference's loop at once."""
        Tw, B = ids_tb.shape
        H = self.hidden_size
        x = Fn.EmbeddingFn.apply(self.embedding.weight, ids_tb.contiguous().view(-1), None, 1.0)
        outs, hs = [], []
        for k, suf in enumerate(("", "_reverse")):
            gi = Fn.linear(x, getattr(self.gru, "weight_ih_l0" + suf), getattr(self.gru, "bias_ih_l0" + suf))
            h0 = hidden[k].contiguous() if hidden is not None else None
            o, h = Fn.GRUDirFn.apply(gi.view(Tw, B, 3 * H), getattr(self.gru, "weight_hh_l0" + suf),
                                     getattr(self.gru, "bias_hh_l0" + suf), h0, None, False)      # forward in time, both
            outs.append(o)
            hs.append(h)
        return torch.cat(outs, 2), torch.stack(hs)

    def forward(self, input: torch.Tensor, hidden: torch.Tensor):
        out, hidden = self.run(input.view(1, -1), hidden)
        return out, hidden

    def initHidden(self) -> torch.Tensor:
        return torch.zeros(2 * self.n_layer, 128, self.hidden_size, device=self.embedding.weight.device)   # batch 128 (:795-802)


class DecoderRNN_New(nn.Module):
    """Embedding(output_size, H) -> nn.GRU(H, H, 1 layer) -> Linear(H, output_size); one step per call (:805-844)."""

    def __init__(self, hidden_size: int, output_size: int, n_layer: int = 2):
        super().__init__()
        if n_layer != 1:
            raise NotImplementedError("text2embedding_model_New builds DecoderRNN_New with n_layer = 1 (:918)")
        self.hidden_size, self.ouput_size, self.n_layer = hidden_size, output_size, n_layer
        self.embedding = nn.Embedding(output_size, hidden_size)
        self.gru = _GRULayer1(hidden_size, hidden_size, bidirectional=False)
        self.fc_out = nn.Linear(hidden_size, output_size)

    def forward(self, input: torch.Tensor, hidden: torch.Tensor):
        """input (B,) code ids, hidden (1,B,H) -> (output (1,B,output_size), hidden (1,B,H))"""
        B, H = input.shape[0], self.hidden_size
        e = Fn.EmbeddingFn.apply(self.embedding.weight, input, None, 1.0)
        gi = Fn.linear(e, self.gru.weight_ih_l0, self.gru.bias_ih_l0)
        _, h = Fn.GRUDirFn.apply(gi.view(1, B, 3 * H), self.gru.weight_hh_l0, self.gru.bias_hh_l0, hidden[0].contiguous(), None, False)
        out = Fn.linear(h, self.fc_out.weight, self.fc_out.bias)
        return out.unsqueeze(0), h.unsqueeze(0)

    def initHidden(self) -> torch.Tensor:
        return torch.zeros(1, 1, self.hidden_size, device=self.embedding.weight.device)


MAX_LENGTH = 4


class AttnDecoderRNN_New(nn.Module):
    """Parameters of the reference's batch-1 tutorial attention decoder (:847-903) so that its state_dict round-trips;
    the class is dead code in the reference (nothing constructs it) and its forward is not on the accelerated path."""

    def __init__(self, hidden_size: int, output_size: int, dropout_p: float = 0.1, max_length: int = MAX_LENGTH):
        super().__init__()
        self.hidden_size, self.output_size, self.dropout_p, self.max_length = hidden_size, output_size, dropout_p, max_length
        self.embedding = nn.Embedding(output_size, hidden_size)
        self.attn = nn.Linear(hidden_size * 2, max_length)
        self.attn_combine = nn.Linear(hidden_size * 2, hidden_size)
        self.dropout = nn.Dropout(dropout_p)
        self.gru = _GRULayer1(hidden_size, hidden_size, bidirectional=False)
        self.out = nn.Linear(hidden_size, output_size)

    def forward(self, input, hidden, encoder_outputs):
        raise NotImplementedError("AttnDecoderRNN_New.forward (batch-1 tutorial code, never called by the reference) is not built")


class text2embedding_model_New(nn.Module):
    """Reference :906-1002: EncoderRNN_New(3863, H, 1) + DecoderRNN_New(H, K + 2, 1); SOS = 512 / EOS = 513 and the one-hot
    width 514 are hard-coded there (:929-930,:968); decoder_hidden = h_dir0 + h_dir1; teacher forcing decided by
    `random.random() < 0.5` once per call (:976-977)."""

    def __init__(self, args, pose_dim: int, n_frames: int, n_words: int, word_embed_size: int, word_embeddings, speaker_model=None):
        super().__init__()
        self.n_layer = 1
        self.encoder = EncoderRNN_New(3863, args.hidden_size, self.n_layer, pre_trained_embedding=word_embeddings)
        pose_dim = int(args.autoencoder_vq_components) + 2
        self.decoder = DecoderRNN_New(args.hidden_size, pose_dim, self.n_layer)
        self.max_length, self.SOS_token, self.eos_token = 90, 512, 513

    def forward(self, in_text: torch.Tensor, in_lengths, poses: torch.Tensor, vid_indices):
        import random
        if not in_text.is_cuda:
            raise RuntimeError("text2embedding_model_New runs on the MI355X kernels only (no CPU fallback)")
        ids = in_text.transpose(0, 1).contiguous()
        cod = poses.transpose(0, 1).contiguous().to(torch.int64)
        S, B = cod.shape
        Kp = self.decoder.ouput_size
        _, enc_hidden = self.encoder.run(ids, None)                       # the reference's per-time-step loop :953-955
        hidden = (enc_hidden[: self.decoder.n_layer] + enc_hidden[self.decoder.n_layer:])     # :971-974 (layout glue)
        outs = [F.one_hot(cod[0], 514).to(torch.float32)] + [None] * (S - 1)
        dec_in = cod[0]
        if random.random() < 0.5:                                          # teacher forcing :979-986
            for di in range(S):
                out, hidden = self.decoder(dec_in, hidden)
                dec_in = cod[di]
                outs[di] = out[0]
        else:                                                              # free running :987-996
            for di in range(1, S):
                out, hidden = self.decoder(dec_in, hidden)
                dec_in = ops.argmax_rows(out[0].detach().contiguous())
                outs[di] = out[0]
        zero = torch.zeros((B, Kp), dtype=torch.float32, device=in_text.device)
        return torch.stack([o if o is not None else zero for o in outs])
