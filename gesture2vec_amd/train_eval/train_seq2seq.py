"""Per-iteration training functions -- mirror of `scripts/train_eval/train_seq2seq.py` of the reference on the
MI355X kernels: `custom_loss` (:40-88), `train_iter_Autoencoder_VQ_seq2seq` (:664-758), plus the fused
clip_grad_norm_(5)+Adam optimiser that replaces `torch.nn.utils.clip_grad_norm_` + `optim.Adam.step` (:743-744)."""
from __future__ import annotations

import logging
from typing import Tuple

import torch

from .. import ops

debug = False
loss_i = 0


class _CustomLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, output, target, w1, w2, w3):
        # the rollout's buffer is time-major; `output` normally IS its transposed view, so this is copy-free
        y_tbd = output.transpose(0, 1).contiguous()
        terms, dy = ops.custom_loss_fwd_bwd(y_tbd, target.contiguous(), w1, w2, w3, 1.0, want_grad=True)
        ctx.set_materialize_grads(False)
        ctx.save_for_backward(dy)
        ctx.terms = terms
        return terms[0].clone()

    @staticmethod
    def backward(ctx, g):
        (dy,) = ctx.saved_tensors
        out = ops.scale(dy, g.reshape(1).contiguous())
        return out.transpose(0, 1), None, None, None, None


def custom_loss(output: torch.Tensor, target: torch.Tensor, args) -> torch.Tensor:
    """w_l1 * mean|o-t| + w_cont * sum_t|o_t-o_{t-1}|/numel - w_var * sum ||o[:, :, d]||_2 (over time)/numel."""
    global loss_i
    loss = _CustomLossFn.apply(output, target, float(args.loss_l1_weight), float(args.loss_cont_weight),
                               float(args.loss_var_weight))
    if loss_i == 100:
        logging.debug("  (loss) %.5f" % loss.item())
        loss_i = 0
    loss_i += 1
    return loss


class FusedClipAdam:
    """Adam(lr, betas=(0.5, 0.999)) + clip_grad_norm_(max_norm) as one fused kernel pair over the model's flat
    parameter buffer (train_autoencoder_VQVAE.py:193-195, train_seq2seq.py:743-744).  API subset of
    torch.optim.Optimizer: zero_grad(), step(), state_dict(), load_state_dict()."""

    def __init__(self, net, lr: float, betas=(0.5, 0.999), eps: float = 1e-8, max_norm: float = 5.0):
        self.net, self.lr, self.betas, self.eps, self.max_norm = net, float(lr), tuple(betas), float(eps), float(max_norm)
        self.grad_scale = 1.0

    def zero_grad(self, set_to_none: bool = False):
        # every backward OVERWRITES the flat grad buffer, so there is nothing to clear
        return None

    def step(self):
        eng = self.net.engine()
        eng.optimizer_step(self.lr, self.betas, self.eps, self.max_norm, self.grad_scale)

    def state_dict(self):
        eng = self.net.engine()
        return {"m": eng.m.clone(), "v": eng.v.clone(), "step": eng.step_counter.clone(), "lr": self.lr,
                "betas": self.betas, "eps": self.eps}

    def load_state_dict(self, sd):
        eng = self.net.engine()
        eng.m.copy_(sd["m"]); eng.v.copy_(sd["v"]); eng.step_counter.copy_(sd["step"])
        self.lr, self.betas, self.eps = float(sd["lr"]), tuple(sd["betas"]), float(sd["eps"])


def train_iter_Autoencoder_VQ_seq2seq(args, epoch: int, input_poses: torch.Tensor, target_poses: torch.Tensor,
                                      net: torch.nn.Module, optim) -> Tuple[dict, torch.Tensor]:
    """One training iteration of the chunk VQ-VAE; same signature / return value as the reference."""
    if getattr(net, "att_use", False):
        # attention decoder: module-level autograd path; its parameter set (attn.*, the wider pre_linear) is not the fused
        # engine's flat layout, so clip + Adam run over a FlatClipAdam of net.parameters()
        from ..flat import FlatClipAdam
        if not isinstance(optim, FlatClipAdam):
            raise TypeError("autoencoder_att == 'True': use gesture2vec_amd.flat.FlatClipAdam(net.parameters(), lr, betas=(0.5, 0.999))")
        optim.zero_grad()
        outputs, _, loss_vq, perplexity_vq = net(input_poses, target_poses, epoch > 0)
        loss = custom_loss(outputs, target_poses, args)
        if epoch > 0:
            loss = loss + 1 * loss_vq / 400
        loss.backward()
        optim.step()
        return {"loss": loss.item()}, perplexity_vq.detach()
    optim = _as_fused(net, optim)
    if getattr(net, "quantizer", "ema") in ("ema", "gssoft") and optim.net is net and net.training and _FUSED_GSSOFT_OK(net):
        # the whole iteration as ONE kernel sequence of the engine (no autograd graph, one host sync for loss.item())
        return _fused_iteration(args, epoch, input_poses, target_poses, net, optim, None, 1)
    optim.zero_grad()
    vq_start_epoch = 0
    outputs, _, loss_vq, perplexity_vq = net(input_poses, target_poses, epoch > vq_start_epoch)
    loss = custom_loss(outputs, target_poses, args)
    if epoch > vq_start_epoch:
        loss = loss + 1 * loss_vq / 400
    loss.backward()
    optim.step()            # clip_grad_norm_(net.parameters(), 5) is fused into the step
    return {"loss": loss.item()}, perplexity_vq.detach()


def train_iter_Autoencoder_VQ_seq2seq_dp(args, epoch: int, input_poses: torch.Tensor, target_poses: torch.Tensor,
                                         net: torch.nn.Module, optim, reduce_fn, world: int) -> Tuple[dict, torch.Tensor]:
    """The data-parallel form of train_iter_Autoencoder_VQ_seq2seq (one process per GPU): this rank's shard goes through
    forward / loss / backward, ONE all-reduce (`reduce_fn`, RCCL) sums [gradients | codebook EMA statistics] over the
    ranks, then every rank applies the identical EMA update (global statistics) and clip + Adam on the mean gradient
    (gesture2vec_amd/dp.py).  Returns this rank's loss and the perplexity of the global code histogram."""
    return _fused_iteration(args, epoch, input_poses, target_poses, net, _as_fused(net, optim), reduce_fn, world)


def _as_fused(net, optim):
    """The reference's harness builds `torch.optim.Adam(net.parameters(), lr, betas=(0.5, 0.999))`
    (train_autoencoder_VQVAE.py:193-195) and hands it to train_iter.  Such an optimiser is ADOPTED: its hyper-parameters are read
    on every call (so an lr schedule keeps working) into the FusedClipAdam that does the work (clip_grad_norm_(5) + Adam as one
    fused launch over the flat buffer); the torch object's own step() is never called and its state stays empty."""
    if isinstance(optim, FusedClipAdam):
        return optim
    if isinstance(optim, torch.optim.Adam) and type(optim) is torch.optim.Adam:
        if len(optim.param_groups) != 1:
            raise TypeError("an adopted torch.optim.Adam must have exactly one parameter group")
        g = optim.param_groups[0]
        if g.get("weight_decay", 0) != 0 or g.get("amsgrad", False) or g.get("maximize", False):
            raise TypeError("the fused clip + Adam step implements plain Adam (no weight_decay / amsgrad / maximize)")
        fused = getattr(net, "_adopted_optim", None)
        if fused is None or fused.net is not net:
            fused = net._adopted_optim = FusedClipAdam(net, g["lr"], betas=tuple(g["betas"]), eps=g["eps"])
        fused.lr, fused.betas, fused.eps = float(g["lr"]), tuple(float(b) for b in g["betas"]), float(g["eps"])
        return fused
    raise TypeError("use gesture2vec_amd.train_eval.train_seq2seq.FusedClipAdam (clip + Adam are one fused HIP launch), or a "
                    "plain torch.optim.Adam, whose hyper-parameters are adopted")


def _fused_iteration(args, epoch, input_poses, target_poses, net, optim, reduce_fn, world):
    """masks -> forward -> custom_loss + loss_vq / 400 -> backward -> [all-reduce] -> EMA update -> clip + Adam through
    VQVAEEngine.train_step; loss = custom_loss + loss_vq / 400 for epoch > 0 (:707,738)."""
    eng = net.engine()
    x, tgt = input_poses.contiguous(), target_poses.contiguous()
    kw = dict(lr=optim.lr, w_l1=float(args.loss_l1_weight), w_cont=float(args.loss_cont_weight),
              w_var=float(args.loss_var_weight), epoch=epoch, draw_masks=not getattr(net, "_explicit_masks", False),
              betas=optim.betas, eps=optim.eps, max_norm=optim.max_norm)
    # one BatchNorm call per decode step: num_batches_tracked += n_frames - 1, by the step itself (a one-thread launch on the
    # step's side branch, inside the replayed graph) instead of a torch launch between the step and the read-back
    eng.tracked_counters = [(net.decoder.decoder.pre_linear[1].num_batches_tracked, net.n_frames - 1)]
    if reduce_fn is None and world == 1 and x.shape[0] >= _GRAPH_MIN_ROWS and _GRAPH_REPLAY and eng.vq_bx_check_every == 0:
        _replayed_step(eng, x, tgt, kw)
    else:
        eng.train_step(x, tgt, reduce_fn=reduce_fn, world=world, **kw)
    # the iteration's one host sync: [custom_loss, loss_vq, perplexity, fault latch] gathered on the device by train_step_apply
    both = eng.readback.tolist()
    if both[3] != 0.0 or eng.vq_bx_check_every > 0:
        try:
            eng.check_faults()    # raises: a faulted step was not applied (the commit kernels gate on the latch)
        except RuntimeError as e:
            # A residency fault of the persistent rollouts (another tenant on the device, a CU mask): the step changed nothing and
            # the engine has switched itself to the per-step kernels -- repeat the SAME iteration once, in-process and eagerly,
            # instead of ending a training run (round-4 advisor finding).  Not under data parallelism: the other ranks applied
            # nothing either (their all-reduced gradients were poisoned by this rank's), but only this rank knows -- raise there.
            if "persistent rollout kernel" not in str(e) or reduce_fn is not None or world != 1:
                raise
            logging.warning("train_iter: %s -- repeating the iteration on the per-step kernels", str(e).split(" -- ")[0])
            for t, n in eng.tracked_counters:         # (the faulted step bumped them; the repeated one does it again)
                t.sub_(int(n))
            eng.train_step(x, tgt, reduce_fn=None, world=1, **kw)
            both = eng.readback.tolist()
            if both[3] != 0.0:
                eng.check_faults()
    if eng.fault_policy.tick():    # enough fault-free iterations since the last fault: the persistent / cluster kernels are back
        eng.rearm()
    loss = both[0] + (both[1] / 400 if epoch > 0 else 0.0)
    return {"loss": loss}, eng.readback[2].clone()


import os as _os
_GRAPH_REPLAY = _os.environ.get("G2V_TRAIN_ITER_GRAPH", "1") != "0"
# rows per batch from which the iteration is replayed from a hipGraph.  Small batches are launch-bound (the reference's own
# config/VQ-VAE.yml shape at B = 128: ~220 launches per iteration), so they gain the most; the parallel branches inside the
# step stay a large-batch matter (VQVAEEngine.overlap_min_rows).
_GRAPH_MIN_ROWS = int(_os.environ.get("G2V_TRAIN_ITER_GRAPH_MIN_ROWS", "0"))


def _FUSED_GSSOFT_OK(net) -> bool:
    """the EMA quantiser always takes the fused iteration; the soft quantiser (what the reference ships) does unless
    G2V_GSSOFT_FUSED=0 selects the module-level autograd path (A/B, parity tests)"""
    return getattr(net, "quantizer", "ema") == "ema" or _os.environ.get("G2V_GSSOFT_FUSED", "1") != "0"


def _replayed_step(eng, x, tgt, kw):
    """The fused step of a large-batch iteration replayed from a hipGraph (round 3: 1.71 ms of eager launches -> 1.62 ms at
    B = 4096; profiles/r03_trainer_gap.json).  The first iteration of a configuration runs eagerly (it sizes every workspace and
    IS a training step), the second captures, later ones replay.  Inputs that do not live at the captured addresses (a data
    loader hands over a new tensor per batch) are copied into the engine's static input buffers first (75 MB at B = 4096:
    ~40 us).  Everything the step reads besides the inputs -- weights, Adam moments, step / RNG counters -- lives at fixed device
    addresses and is read by the kernels at replay time; the scalars baked into the graph are part of the cache key."""
    key = (tuple(x.shape), kw["lr"], tuple(kw["betas"]), kw["eps"], kw["max_norm"], kw["w_l1"], kw["w_cont"], kw["w_var"],
           kw["epoch"] > 0, kw["draw_masks"], eng.flat.data_ptr(), eng.codebook.data_ptr(), eng.vq_pre_w.data_ptr(),
           eng.bn_rm.data_ptr(), eng.overlap, tuple((t.data_ptr(), n) for t, n in eng.tracked_counters))
    # one slot per configuration (a data loader's short last batch alternates with the full ones: neither may evict the other),
    # at most _GRAPH_SLOTS of them; eng._iter_graph = the slot of the latest call (tests / check_faults look at it)
    slots = getattr(eng, "_iter_graphs", None)
    if slots is None or getattr(eng, "_iter_graph", 0) is None:          # (check_faults() drops every graph by setting it to None)
        slots = eng._iter_graphs = {}
    st = slots.get(key)
    if st is None:
        eng.train_step(x, tgt, **kw)                           # first sight of this configuration: a plain eager step
        if len(slots) >= _GRAPH_SLOTS:
            slots.pop(next(iter(slots)))                       # oldest configuration out
        eng._iter_graph = slots[key] = {"key": key, "seen_key": key, "graph": None}
        return
    eng._iter_graph = st
    if st["graph"] is None:
        gx = torch.empty_like(x)
        gt = gx if tgt.data_ptr() == x.data_ptr() else torch.empty_like(tgt)
        gx.copy_(x)
        if gt is not gx:
            gt.copy_(tgt)
        graph = torch.cuda.CUDAGraph()
        torch.cuda.synchronize()
        try:
            with torch.cuda.graph(graph):
                eng.train_step(gx, gt, **kw)
        except Exception as e:                                  # capture is an optimisation, never a requirement
            logging.warning("train_iter: hipGraph capture unavailable (%s: %s); eager launches from now on", type(e).__name__, e)
            st.update({"graph": False, "gx": None, "gt": None, "same": True})
            # the aborted capture may have left branches recorded as open / deferred: the eager retry starts from a clean slate
            eng._open.clear()
            eng._deferred.clear()
            eng.train_step(x, tgt, **kw)
            return
        st.update({"graph": graph, "gx": gx, "gt": gt, "same": gt is gx})
        graph.replay()
        return
    if st["graph"] is False:
        eng.train_step(x, tgt, **kw)
        return
    xp, tp = x.data_ptr(), tgt.data_ptr()
    if xp != st["gx"].data_ptr():
        # Inputs at an address this configuration has been handed before (a loader's rotating device buffers; the caching
        # allocator giving every batch the previous batch's block; a benchmark's resident tensor): from the second sighting on
        # the step is captured ONCE MORE, reading that address directly -- no 75 MB copy into the static buffers (~30 us + its
        # launch gap per iteration at B = 4096).  The address is checked on every call, so the graph always reads the tensor it
        # was handed; at most _ADDR_GRAPHS of them per configuration, oldest out.
        by_addr = st.setdefault("by_addr", {})
        g = by_addr.get((xp, tp))
        if g is None:
            seen = st.setdefault("addr_seen", {})
            n = seen.get((xp, tp), 0) + 1
            if len(seen) > 64:
                seen.clear()
            seen[(xp, tp)] = n
            if n >= 2 and _ADDR_GRAPHS > 0:
                g = torch.cuda.CUDAGraph()
                torch.cuda.synchronize()
                try:
                    with torch.cuda.graph(g):
                        eng.train_step(x, tgt, **kw)
                except Exception as e:
                    logging.warning("train_iter: per-address hipGraph capture failed (%s: %s); inputs are copied", type(e).__name__, e)
                    eng._open.clear()
                    eng._deferred.clear()
                    g = None
                    seen[(xp, tp)] = -(1 << 30)          # never again for this address
                if g is not None:
                    if len(by_addr) >= _ADDR_GRAPHS:
                        by_addr.pop(next(iter(by_addr)))
                    by_addr[(xp, tp)] = g
        if g is not None:
            g.replay()
            return
        st["gx"].copy_(x)
    if not st["same"] and tp != st["gt"].data_ptr():
        st["gt"].copy_(tgt)
    elif st["same"] and tp != xp:
        # captured with target == input; a separate target now: leave the replayed path for good
        st.update({"graph": False, "gx": None, "gt": None, "same": True})
        eng.train_step(x, tgt, **kw)
        return
    st["graph"].replay()


_ADDR_GRAPHS = 4
_GRAPH_SLOTS = 4


def train_iter_DAE(args, epoch: int, noisy_poses: torch.Tensor, target_poses: torch.Tensor, net: torch.nn.Module, optim):
    """One training iteration of the Part-a frame DAE (reference :161-241, autoencoder_vq/vae == "False"):
    MSE(outputs, target) -> backward -> clip_grad_norm_(5) + Adam (fused in `optim`, a gesture2vec_amd.flat.FlatClipAdam)."""
    from ..flat import FlatClipAdam
    from ..functional import mse_loss
    if not isinstance(optim, FlatClipAdam):
        raise TypeError("use gesture2vec_amd.flat.FlatClipAdam (clip + Adam are one fused HIP launch)")
    if getattr(args, "autoencoder_vq", "False") == "True" or getattr(args, "autoencoder_vae", "False") == "True":
        raise NotImplementedError("VQ_Frame / VAE_Network variants of Part a are outside the accelerated hot path")
    optim.zero_grad()
    outputs = net(noisy_poses)
    loss = mse_loss(outputs, target_poses)
    loss.backward()
    optim.step()
    return {"loss": loss.item()}


def _code_loss(outputs, codes):
    """CrossEntropyLoss(outputs[:, 1:, :].reshape(-1, K), codes[:, 1:].reshape(-1)) (reference :506-513).  The model's outputs
    (B,S,K) are a transposed view of a step-major (S,B,K) array; when the model says so the loss runs on that array in place
    (same set of (sample, step) rows, same mean) instead of on a strided copy of it."""
    from ..functional import cross_entropy
    K = outputs.shape[2]
    full = getattr(outputs, "_g2v_step_major", None)
    if full is not None and full.shape[0] == outputs.shape[1] and full.shape[1] == outputs.shape[0]:
        S, B = full.shape[0], full.shape[1]
        return cross_entropy(full.view(S * B, K), codes.t().reshape(-1).long(), skip_rows=B)
    return cross_entropy(outputs[:, 1:, :].reshape(-1, K), codes[:, 1:].reshape(-1).long())


def _code_loss_backward(outputs, codes):
    """loss = _code_loss(outputs, codes); loss.backward() -- returns the (detached) loss.  The loss is the root of the iteration's
    graph, so its backward starts from d loss / d logits with an upstream gradient of exactly 1: when the outputs are the
    step-major array of the fused rollout node, the cross-entropy kernel's own gradient (it writes softmax - onehot over M beside
    the loss) is handed to that node directly -- no ones_like fill, no scaling pass, no zero fill of slot 0's rows (the rollout's
    backward never reads them: outputs[:, 0] is a constant, reference :676-677), no second transposed copy of the targets."""
    from .. import ops
    full = getattr(outputs, "_g2v_step_major", None)
    tgt = getattr(outputs, "_g2v_targets", None)
    if (full is None or tgt is None or not full.requires_grad or full.shape[0] != outputs.shape[1]
            or full.shape[1] != outputs.shape[0] or tgt.shape != full.shape[:2] or tgt.dtype != torch.int64):
        loss = _code_loss(outputs, codes)
        loss.backward()
        return loss.detach()
    S, B, K = full.shape
    d_full = torch.empty_like(full)
    with torch.no_grad():
        loss, _ = ops.cross_entropy_fwd_bwd(full.view(S * B, K)[B:], tgt.view(-1)[B:], want_grad=True, dl_out=d_full.view(S * B, K)[B:])
    full.backward(gradient=d_full)
    return loss.view(())


def train_iter_text2embedding(args, epoch: int, in_text, in_lengths, in_audio, target_poses, cluster_targets,
                              GPT3_Embedding, net: torch.nn.Module, optim):
    """One training iteration of Part d (reference :462-538), discrete codes: CrossEntropyLoss over decode steps 1..S-1,
    clip_grad_norm_(5) + Adam (fused in `optim`, a gesture2vec_amd.flat.FlatClipAdam)."""
    from ..flat import FlatClipAdam
    from ..functional import cross_entropy
    if not isinstance(optim, FlatClipAdam):
        raise TypeError("use gesture2vec_amd.flat.FlatClipAdam (clip + Adam are one fused HIP launch)")
    if args.text2_embedding_discrete != "True":
        raise NotImplementedError("text2_embedding_discrete == 'False' is outside the accelerated hot path")
    from .. import _lib, ops
    lib = _lib.load()
    defer = hasattr(net, "commit_bn_running_stats")
    for attempt in (0, 1):
        optim.zero_grad()
        if defer:
            net.deferred_bn = []           # BatchNorm's running statistics: held back until the whole iteration is known to be valid
        outputs, _ = net(in_text, in_lengths, in_audio, cluster_targets, GPT3_Embedding, None)
        with ops.side_branches():          # the backward's parameter gradients beside its chain; joined in front of the optimiser
            loss = _code_loss_backward(outputs, cluster_targets)
        if defer:
            net.commit_bn_running_stats()  # behind the backward, latch-gated on the device like clip + Adam below
        optim.step()
        ret = {"loss": loss.item()}
        # The encoder's small-batch GRU kernels keep their workgroups resident for the whole sequence (include/g2v.h:
        # g2v_gru_seq_set_cluster) and latch the persistent kernels' fault word when a bounded wait runs out (a workgroup of the
        # launch was not resident: CU mask, another tenant).  A faulted iteration changed nothing -- clip + Adam and the commit of
        # BatchNorm's running statistics sit BEHIND the backward and read the latch on the device (round 6: the statistics used
        # to be committed inside the forward rollout, before a fault of a later step or of the backward could be known) -- so it
        # is repeated once on the per-step kernels.
        f = int(lib.g2v_dec_rollout_persist_fault(0))      # (one 4-byte read at the point where loss.item() synchronised anyway)
        from ..fault_policy import POLICY
        if f == 0:
            POLICY.tick()                                  # (re-arms the fast path after enough clean iterations: nothing is cached here)
            return ret
        POLICY.on_fault()                                  # clear the latch; per-step kernels (the code decoder's cluster forward too)
        if attempt == 1:
            raise RuntimeError(f"persistent kernel fault latch {f} set again on the per-step kernels")
        import warnings
        warnings.warn(f"persistent GRU kernels: fault latch {f} (a workgroup of the launch was not resident); the iteration was not "
                      "applied and is repeated on the per-step kernels, which stay selected until re-armed", RuntimeWarning)
        bn = getattr(getattr(getattr(net, "decoder", None), "decoder", None), "pre_linear", None)
        if bn is not None and hasattr(bn[1], "num_batches_tracked"):
            bn[1].num_batches_tracked -= outputs.shape[1] - 1      # (the repeated iteration counts its decode steps again)
    return ret


class GraphedText2EmbeddingStep:
    """train_iter_text2embedding as ONE hipGraph: zero_grad -> forward -> CE -> backward -> BatchNorm commit -> clip+Adam are
    captured once and replayed (the operator chain of Part d is ~100 small launches per step: host-bound when launched one by one
    at the reference's B=128).  Inputs are static device tensors: overwrite `in_text` / `codes` (and, with the default
    static_lengths=False, `lengths`) in place between replays.  Dropout masks come from the Philox kernels (device-side counters),
    so every replay draws fresh masks.  `loss` is a device scalar updated by each replay (read it when needed: no per-step host sync).

    static_lengths=True (opt-in; bench.py does and says so on its line): the sentence lengths become part of the captured graph --
    the encoder then runs its layer-0 products on the packed positions only (EncoderRNN.forward), whose row counts are baked into
    the launches.  Such a graph is valid for THESE lengths only: `set_lengths()` re-captures, and `replay()` refuses a `lengths`
    argument that differs from the baked tuple.

    Faults (round 6, advisor finding).  The captured step contains persistent cluster kernels at small batch; their fault latch is
    sticky and every kernel that commits the step reads it, so after ONE residency timeout every later replay would be a silent
    no-op (parameters frozen, a loss still reported).  `replay()` therefore reads the latch every `check_every` replays (one
    4-byte read = one host sync; default 32), `read_loss()` always does: on a fault the latch is cleared, the per-step kernels are
    selected (fault_policy.POLICY, which also re-arms the fast path later), the graph is captured again and the step is repeated.
    The replays between the fault and its detection changed nothing and are reported in `lost_replays`."""

    def __init__(self, args, net, optim, in_text, in_lengths, codes, warmup: int = 3, static_lengths: bool = False,
                 check_every: int = 32):
        from ..flat import FlatClipAdam
        if not isinstance(optim, FlatClipAdam):
            raise TypeError("use gesture2vec_amd.flat.FlatClipAdam")
        from .. import _lib
        self._lib = _lib.load()
        self.in_text = in_text
        self.codes = codes
        self.static_lengths = bool(static_lengths)
        self.net, self.optim = net, optim
        self.warmup = max(int(warmup), 1)
        self.check_every = int(check_every)
        self.replays = self.lost_replays = self.recaptures = 0
        self._since_check = 0
        self._set_lengths(in_lengths)
        self._capture(self.warmup)

    def _set_lengths(self, in_lengths):
        dev = self.in_text.device
        if self.static_lengths:
            self.lengths = in_lengths.cpu()
            self._baked = tuple(int(v) for v in self.lengths.tolist())
        else:
            self.lengths = in_lengths.to(device=dev, dtype=torch.int32).contiguous()
            self._baked = None

    def _step(self):
        from .. import ops
        net, optim = self.net, self.optim
        defer = hasattr(net, "commit_bn_running_stats")
        optim.zero_grad()
        if defer:
            net.deferred_bn = []
        outputs, _ = net(self.in_text, self.lengths, None, self.codes, None, None)
        with ops.side_branches():
            loss = _code_loss_backward(outputs, self.codes)
        if defer:
            net.commit_bn_running_stats()
        optim.step()
        return loss

    def _capture(self, warmup: int):
        from ..fault_policy import POLICY
        from .. import ops
        import gc
        # (a graph this object captured earlier is released before the next capture, and the capture forks onto side streams no
        #  earlier graph has used: ops.reset_side_streams has the measurement)
        self.graph = None
        gc.collect()
        ops.reset_side_streams()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup):
                self._step()
        torch.cuda.current_stream().wait_stream(side)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.loss = self._step()
        self._generation = POLICY.generation

    def set_lengths(self, in_lengths):
        """new sentence lengths: in place for a device-lengths graph, a re-capture for a static-lengths one"""
        if self.static_lengths:
            if tuple(int(v) for v in in_lengths.tolist()) != self._baked:
                self._set_lengths(in_lengths)
                self._capture(1)
                self.recaptures += 1
        else:
            self.lengths.copy_(in_lengths.to(device=self.lengths.device, dtype=torch.int32))

    def check_faults(self) -> bool:
        """True: a persistent kernel of some replay since the last check had latched a fault -- none of those replays was applied;
        the graph has been re-captured on the per-step kernels, and the capture's eager warm-up step -- a real training step on the
        current inputs -- stands in for them.  (A re-capture behind a re-armed fast path costs one such extra step too.)"""
        from ..fault_policy import POLICY
        n, self._since_check = self._since_check, 0
        f = int(self._lib.g2v_dec_rollout_persist_fault(0))       # (synchronises with the device)
        if f == 0:
            for _ in range(max(n, 1)):
                if POLICY.tick():
                    break
            if POLICY.generation != self._generation:               # the fast path was re-armed: capture it again
                self._capture(1)
                self.recaptures += 1
            return False
        import warnings
        POLICY.on_fault()
        self.lost_replays += n
        warnings.warn(f"GraphedText2EmbeddingStep: persistent kernel fault latch {f}; the last {n} replay(s) were not applied -- "
                      "re-capturing on the per-step kernels and repeating one step", RuntimeWarning)
        bn = getattr(getattr(getattr(self.net, "decoder", None), "decoder", None), "pre_linear", None)
        if bn is not None and hasattr(bn[1], "num_batches_tracked"):
            bn[1].num_batches_tracked -= n * (self.codes.shape[1] - 1)     # (the unapplied replays counted their decode steps)
        self._capture(1)            # (its eager warm-up step IS the repeated step)
        self.recaptures += 1
        if int(self._lib.g2v_dec_rollout_persist_fault(0)) != 0:
            raise RuntimeError("persistent kernel fault latch set again on the per-step kernels")
        return True

    def replay(self, lengths=None):
        if lengths is not None and self.static_lengths and tuple(int(v) for v in lengths.tolist()) != self._baked:
            raise ValueError("this graph was captured with static_lengths=True for other sentence lengths: call set_lengths()")
        self.graph.replay()
        self.replays += 1
        self._since_check += 1
        if self.check_every > 0 and self._since_check >= self.check_every:
            self.check_faults()
        return self.loss

    def read_loss(self) -> float:
        """the last replay's loss as a float, behind a fault check (the value of an unapplied replay is never returned)"""
        self.check_faults()
        return float(self.loss.detach())
