"""VQVAEEngine -- host-side orchestration of the chunk VQ-VAE hot path over the C-ABI.

Mirrors, call for call, what `Autoencoder_VQVAE.forward` (model/Autoencoder_VQVAE_model.py:901-1072),
`custom_loss` (train_eval/train_seq2seq.py:40-88) and `train_iter_Autoencoder_VQ_seq2seq` (:664-758) do in the
reference, but every tensor op is one of the HIP kernels behind include/g2v.h.  The engine owns

  * ONE flat fp32 parameter buffer (and matching flat grad / Adam m / Adam v buffers) holding every trainable
    tensor of the model; the nn.Parameters the user sees are views into it, so clip+Adam is a single fused
    launch pair and the data-parallel gradient exchange is a single RCCL all-reduce;
  * all activations / saved-for-backward tensors, preallocated per batch size (static addresses: the whole
    train step can be captured into a hipGraph);
  * explicit dropout keep-masks (drawn by the Philox kernel, or supplied by the caller for parity tests).

Scope: autoencoder_vq == "True", autoencoder_vae == "False", autoencoder_att == "False", n_layers == 2 (every
shipped VQ-VAE config).  With attention off the decoder only consumes encoder_hidden[:2] = the layer-0
forward/backward final states (:971-973), so encoder GRU layer 1 is never evaluated: its outputs reach nothing and
its gradients are exactly zero in the reference [SURVEY.md §0]; its weights stay in the state_dict untouched.
"""
from __future__ import annotations

import contextlib
import ctypes as C
import os
from typing import Dict, Optional

import torch

from . import _lib, ops
from ._lib import DecGrads, DecSaved, DecWeights, check

_p = ops._p


def _enc_gru_names(L):
    names = []
    for l in range(L):
        for suf in ("", "_reverse"):
            names += [f"weight_ih_l{l}{suf}", f"weight_hh_l{l}{suf}", f"bias_ih_l{l}{suf}", f"bias_hh_l{l}{suf}"]
    return names


def _dec_gru_names(L):
    names = []
    for l in range(L):
        names += [f"weight_ih_l{l}", f"weight_hh_l{l}", f"bias_ih_l{l}", f"bias_hh_l{l}"]
    return names


QUANTIZER_GSSOFT_PARAMS = ("vq_layer._embedding.weight", "vq_layer.mean_layer.weight", "vq_layer.mean_layer.bias",
                           "vq_layer.logvar_layer.weight", "vq_layer.logvar_layer.bias")


def quantizer_layout(quantizer: str, E: int, K: int):
    """trainable tensors of the quantiser itself.  EMA (:1182-1301): none (codebook moves by EMA, pre_linear gets no gradient).
    GSSoft (:1304-1438, what the reference's Autoencoder_VQVAE ships with): codebook, mean_layer, logvar_layer."""
    if quantizer == "ema":
        return []
    if quantizer == "gssoft":
        return list(zip(QUANTIZER_GSSOFT_PARAMS, ((K, E), (E, E), (E,), (K, E), (K,))))
    raise ValueError(f"unknown quantizer {quantizer!r}")


def trainable_layout(D: int, H: int, L: int):
    """(name, shape) of every tensor that receives a gradient in the reference's step, in flat-buffer order."""
    out = [("encoder.in_layer.weight", (H, D)), ("encoder.in_layer.bias", (H,))]
    for n in _enc_gru_names(L):
        l = int(n.split("_l")[1][0])
        if n.startswith("weight_ih"):
            shp = (3 * H, H if l == 0 else 2 * H)
        elif n.startswith("weight_hh"):
            shp = (3 * H, H)
        else:
            shp = (3 * H,)
        out.append(("encoder.gru." + n, shp))
    out += [("decoder.decoder.pre_linear.0.weight", (H, D)), ("decoder.decoder.pre_linear.0.bias", (H,)),
            ("decoder.decoder.pre_linear.1.weight", (H,)), ("decoder.decoder.pre_linear.1.bias", (H,))]
    for n in _dec_gru_names(L):
        shp = (3 * H, H) if n.startswith("weight") else (3 * H,)
        out.append(("decoder.decoder.gru." + n, shp))
    out += [("decoder.decoder.out_layer.weight", (D, H)), ("decoder.decoder.out_layer.bias", (D,))]
    return out


class _DeferredReductions:
    """The slab reductions of one branch's weight-gradient products, held back and run as ONE launch (include/g2v.h:
    g2v_linear_bwd_weight_deferred / _reduce).  Every product gets a region of `ws` of its own (its slabs live there until
    `flush`).  Round 6: a chain of immediate calls made every product wait for the reduction of the one in front of it, and
    beside the encoder's BPTT kernel -- whose two workgroups per CU leave a late-dispatched kernel no registers -- those
    reductions took 17-95 us instead of 5 (profiles/r05_az_step_timeline.txt: 133 us of them on the decoder branch's chain)."""

    def __init__(self, eng, ws: torch.Tensor):
        self.eng, self.ws, self.off, self.pend = eng, ws, 0, []

    def call(self, items, nprob, lddy, ldx, row_map, dy_b, M, K, N, flags):
        lib = self.eng.lib
        need = int(nprob * lib.g2v_linear_bwd_weight_workspace(M, K, N))
        off = self.off
        self.off = (off + need + 255) & ~255
        assert self.off <= self.ws.numel(), "deferred weight-gradient workspace too small"
        pd = _lib.WgradPending()
        check(lib.g2v_linear_bwd_weight_deferred(items, nprob, lddy, ldx, row_map[0], row_map[1], row_map[2], dy_b, M, K, N, flags,
                                                 self.ws.data_ptr() + off, need, C.byref(pd), self.eng._stream()))
        self.pend.append(pd)

    def flush(self):
        live = [p for p in self.pend if p.nprob > 0]
        self.pend, self.off = [], 0
        for k in range(0, len(live), 8):
            chunk = live[k:k + 8]
            arr = (_lib.WgradPending * len(chunk))(*chunk)
            check(self.eng.lib.g2v_linear_bwd_weight_reduce(arr, len(chunk), self.eng._stream()))


class VQVAEEngine:
    def __init__(self, D: int, H: int, L: int, K: int, T: int, *, beta: float, dropout_prob: float,
                 n_pre_poses: int = 1, conditioned: bool = True, decay: float = 0.85, eps: float = 1e-5,
                 device="cuda:0", seed: int = 0, quantizer: str = "ema"):
        if L != 2:
            raise NotImplementedError("the gfx950 rollout kernels implement n_layers == 2 (every shipped config)")
        self.lib = _lib.load()
        # The library's implementation switches (persistent / cluster kernels, small-row-count threshold) of THIS engine: a
        # caller-owned context (include/g2v.h: g2v_ctx) that every public method binds to the calling thread around its library
        # calls.  Two engines in one process do not share switches; a residency fault here flips only this engine's (round 6;
        # they used to be process-global variables).  `fault_policy` switches the fast path off on a fault and re-arms it later.
        self.ctx = _lib.Context()
        # (the encoder's BPTT runs BESIDE the decoder's weight-gradient products here: the W_hh-resident kernel would take every CU
        #  whole -- native dims at B = 4096 5.90 -> 6.44 ms with it, profiles/r06_n_engine_ab.log -- so this engine keeps the
        #  streaming BPTT; the resident FORWARD, alone on the chain, stays: 6.02 -> 5.91)
        self.ctx.set(_lib.OPT_GRU_RESIDENT_BWD, 0)
        from .fault_policy import PersistentPathPolicy
        self.fault_policy = PersistentPathPolicy(ctx=self.ctx)
        # Opt-in: run the weight-gradient products on the bf16 matrix pipe as 3-term splits (G2V_WGRAD_BF16X3: ~3e-5 max-norm
        # relative error on dW instead of 3e-7; -0.17 ms / step at the BASELINE shape).  Default: exact fp32 MFMA.
        self.wgrad_bf16x3 = False
        # Round 6: the slab reductions of a branch's weight-gradient products as ONE launch at the branch's end (_DeferredReductions)
        # instead of one behind every product (the round-5 verdict's "deferred slab reductions").  Built, bitwise the same results,
        # and measured: 1.554-1.559 ms against 1.537-1.546 with the immediate calls (same box, 3 x 300 steps,
        # profiles/r06_c_defer_ab.log) -- the decoder branch's products then run entirely beside the encoder BPTT, which starves
        # them, instead of partly behind it.  OFF by default; G2V_DEFER_REDUCE=1 / eng.defer_reduce = True select it.
        self.defer_reduce = os.environ.get("G2V_DEFER_REDUCE", "0") != "0"
        self.D, self.H, self.L, self.K, self.T = D, H, L, K, T
        self.E = H * L
        self.beta, self.p, self.n_pre, self.conditioned = float(beta), float(dropout_prob), int(n_pre_poses), bool(conditioned)
        self.decay, self.eps = float(decay), float(eps)
        self.device = torch.device(device)
        self.seed = int(seed)
        # "ema": the fused path (north star).  "gssoft": the encoder / decoder stages of this engine around the
        # VQ_Payam_GSSoft module (model/Autoencoder_VQVAE_model.py); its trainable tensors live at the END of the same flat
        # buffer so that clip_grad_norm_ + Adam stay one fused launch over everything.
        self.quantizer = quantizer
        self.layout = trainable_layout(D, H, L) + quantizer_layout(quantizer, H * L, K)
        self.offsets: Dict[str, tuple] = {}
        off = 0
        for name, shp in self.layout:
            n = 1
            for s in shp:
                n *= s
            self.offsets[name] = (off, n, shp)
            off += (n + 3) // 4 * 4          # keep every tensor 16-byte aligned inside the flat buffer
        self.n_flat = off
        q_names = [n for n, _ in quantizer_layout(quantizer, H * L, K)]
        self.q_off = self.offsets[q_names[0]][0] if q_names else off        # start of the quantiser's slice
        dev = self.device
        self.flat = torch.zeros(off, dtype=torch.float32, device=dev)
        # ONE communication buffer [flat grads | cnt (K) | dw (K*E)]: a single RCCL all-reduce per step under DP
        # [grads | cnt (K) | dw (K E) | fault flag (4 floats: one used)]: the flag travels in the same all-reduce (round 5)
        self.comm = torch.zeros(off + K + K * (H * L) + 4, dtype=torch.float32, device=dev)
        self.gflat = self.comm[:off]
        self.m = torch.zeros(off, dtype=torch.float32, device=dev)
        self.v = torch.zeros(off, dtype=torch.float32, device=dev)
        self.step_counter = torch.zeros(1, dtype=torch.int32, device=dev)
        self.rng_counter = torch.zeros(1, dtype=torch.int64, device=dev)
        self.adam_partial = torch.zeros(self.lib.g2v_adam_blocks(off), dtype=torch.float32, device=dev)
        self.gnorm = torch.zeros(1, dtype=torch.float32, device=dev)
        # [custom_loss, loss_vq, perplexity, fault latch] of the latest train_step_apply: train_iter's ONE device-to-host copy
        self.readback = torch.zeros(4, dtype=torch.float32, device=dev)
        # quantiser state (not trainable by gradient: grad=None in the reference, :1276-1282)
        E = self.E
        self.vq_pre_w = torch.zeros(E, E, device=dev)
        self.vq_pre_b = torch.zeros(E, device=dev)
        self.codebook = torch.zeros(K, E, device=dev)
        self.ema_w = torch.zeros(K, E, device=dev)
        self.ema_cs = torch.zeros(K, device=dev)
        self.code_sqnorm = torch.zeros(K, device=dev)
        # What the fused assign kernels read besides the codebook itself, all DERIVED from (codebook, pre_linear) by vq_derive():
        #   bf16-screened kernel (g2v_vq_fused_assign_bx_fwd, where g2v_vq_fused_assign_bx_ok says so: E == 128, K in {128..512}):
        #     pre_linear's weight as fp32 MFMA fragments + the screening image (bf16 fragments of U = W w_pre, s'_k, norm bounds);
        #   fp32 kernel (every other shape): fragment-major image of the codebook.
        self._vq_bx = bool(self.lib.g2v_vq_fused_assign_bx_ok(1, self.E, K))
        # custom_loss by the CHASER kernel beside the persistent forward rollout + the backward rollout's own tile load
        # (include/g2v.h: g2v_custom_loss_chase, g2v_dec_saved.loss_*) instead of its own launch between the rollouts, wherever the
        # fused train step runs its parallel branches and g2v_dec_rollout_fuses_loss says so.  Bitwise the same gradients.
        self.loss_chase = True
        self._fuse_vq_bwd = True            # quantiser backward inside the encoder's BPTT kernel (H == 64); False: its own launch (parity tests)
        self.vq_bx_flags = 0                # include/g2v.h G2V_VQ_BX_*: 1 = exact fp32 sweep on every tile (the A/B reference of the screening)
        self.vq_bx_check_every = 0          # > 0: every n-th fused train step re-assigns the batch with the exact sweep and compares (debug)
        self._vq_bx_mismatch = torch.zeros(1, dtype=torch.int64, device=dev)
        self._steps = 0
        self.vq_wpre_frag = torch.zeros(self.E * self.E, device=dev) if self._vq_bx else None
        self.vq_bx_image = (torch.zeros(int(self.lib.g2v_vq_bx_image_bytes(K, self.E)), dtype=torch.uint8, device=dev)
                            if self._vq_bx else None)
        self.vq_diag = torch.zeros(4, dtype=torch.int32, device=dev)          # [0] tiles on the exact sweep, [1] pairs re-evaluated
        self._vq_diag_on = False
        self.codebook_frag = torch.zeros(K * self.E, device=dev) if (self.E == 128 and K % 128 == 0 and not self._vq_bx) else None
        # every other shape the packed kernel serves (E = 400: the reference's own): pre_linear as a dense launch + the eight-wave
        # assignment kernel on the fragment-major codebook image (round 5; g2v_vq_assign_packed_fwd), from 2048 rows
        self.codebook_frag_generic = (torch.zeros(K * self.E, device=dev)
                                      if (not self._vq_bx and self.codebook_frag is None and
                                          self.lib.g2v_vq_assign_packed_ok(1, self.E, K)) else None)
        self.bn_rm = torch.zeros(H, device=dev)
        self.bn_rv = torch.ones(H, device=dev)
        self.vq_stats = self.comm[self.n_flat:self.n_flat + K + K * self.E]
        self.fault_flag = self.comm[self.n_flat + K + K * self.E:]
        self.vq_scalars = torch.zeros(2, device=dev)          # loss_vq, perplexity
        self.loss_terms = torch.zeros(5, device=dev)          # custom_loss total, l1, cont, var, mse
        self.g_loss_vq = torch.full((1,), 1.0 / 400.0, device=dev)
        # the soft quantiser's fused sequence (_forward_gssoft / _backward_gssoft): d total / d loss_vq as the step has it (host
        # float baked into the launch arguments + the same value on the device), and 1 + beta
        self._g_vq_host, self._g_vq_dev = 1.0 / 400.0, self.g_loss_vq
        self._one_plus_beta = torch.full((1,), 1.0 + float(beta), device=dev)
        self._bufs: Dict[int, dict] = {}
        self._wstruct = None
        # tensors with requires_grad == False in the reference (autoencoder_fixed_weight == "True" freezes the decoder GRU,
        # :483-486): their gradient slots are zeroed after the backward, so they add nothing to the clip norm and Adam
        # (m = v = 0, g = 0) leaves them exactly where they are -- what the reference's clip_grad_norm_ / Adam do by skipping them
        self.frozen: list = []
        # code_sqnorm (||W_k||^2) and the images above are recomputed from the codebook by EVERY entry point that assigns codes
        # (vq_derive: 2-3 small launches; the fused train step runs them in the branch beside the encoder, so they are captured
        # in its hipGraph and cost nothing on the chain).  Round 2 trusted them from one step to the next behind a Python flag
        # that in-place writers of the codebook (vq_layer(x) in train mode, a state load on the sub-module, a broadcast) did not
        # clear: nothing is trusted across calls any more.
        # Independent kernel chains of the fused train step run as parallel branches (side HIP streams; parallel branches of
        # the hipGraph when the step is captured) -- see _branch().  Bit 0: the dropout keep-masks beside the encoder
        # forward; bit 1: the EMA statistics + codebook update beside the decoder rollout; bit 2: the decoder's weight
        # gradients beside the encoder's backward; bit 3: the weight-fragment packs of the four recurrent launches and the clearing
        # of the rollout's exchange regions (everything that depends on the weights only and is not needed at once) inside
        # branch 0, beside the encoder (a branch of their own costs more at its fork and join than the 25 us it hides).
        # G2V_OVERLAP=0 serialises everything on the caller's stream (debugging).
        # Bit 4: the encoder's GRU weight gradients beside its input layer's gradient chain -- pays on the generic dims (native
        # B = 128 1.235 -> 1.206 ms, GENEA dims 0.745 -> 0.727, native B = 4096 7.15 -> 7.10), costs 1.5 % at H = 64 where those
        # products are fused elsewhere (profiles/r05_u2_branch4.log): on by default for H != 64 only.
        self.overlap = int(os.environ.get("G2V_OVERLAP", "15" if H == 64 else "31"))
        self.tracked_counters = []          # [(int64 device tensor, increment)]: bumped once per train step on the side branch
        self._prepared = False          # the workspaces of this step's recurrent launches hold their packs already
        # A fork / join costs an event record + wait on the host when launched eagerly and ~10-20 us inside a replayed graph: where
        # the step is ~300 small launches (generic dims on the per-step kernels) that is more than the overlap returns (native
        # VQ-VAE.yml dims at B = 512: 2.61 -> 2.79 ms), so there the branches are used from 1024 rows per batch only.  Where the
        # recurrent stages are single launches -- the persistent rollouts at H = 64, the cluster kernels of the generic dims at small
        # batch (round 5) -- they pay at every batch size: full shape B = 128 1.035 -> 0.970 ms, B = 512 1.136 -> 1.046, native
        # B = 128 1.352 -> 1.303 (profiles/r05_u_branches_small_batch.log); _branches_ok() asks the library which case a shape is.
        self.overlap_min_rows = int(os.environ.get("G2V_OVERLAP_MIN_ROWS", "1024"))
        # the cluster launches' exchange records cleared on branch 0 instead of in front of each launch (switches)
        self.xch_preclear = os.environ.get("G2V_XCH_PRECLEAR", "1") != "0"
        self._xch_pre = False
        self.compose_in = True               # in_layer + GRU input projections on composed weights (forward_encoder; A/B only)
        self._branches_on = True
        self._sides: Dict[int, torch.cuda.Stream] = {}
        self._open: list = []
        self._deferred: list = []
        self._fused_in_drop = False
        # bit k: branch k is launched behind the main chain's next kernel (_fork): in a captured hipGraph the first successor
        # recorded at a fork keeps the fork node's hardware queue.  Measured per branch at B = 4096 (round 3): branch 0 (masks, packs,
        # vq_derive beside the encoder GRU) -5..-9 us per step; branch 1 (EMA statistics beside the forward rollout) +30 us; branch 2
        # (decoder weight gradients) +10 us.  Hence 1.
        self._fork_late = 1
        # round 5: branch 0 forked in front of the input layer, masks last; its packs + exchange clears as one launch
        # (same-box A/B, 3 x 300 steps each, profiles/r05_b_side_ab.log: 1.558 -> 1.540 ms with both; attributes, not environment)
        self.side_early = True
        self.merged_prepare = True
        # Round 5 (advisor finding): inside the fused train step the kernels that COMMIT the step to the model state beside Adam --
        # the EMA codebook update and the BatchNorm running statistics -- run BEHIND the backward rollout (end of branch 2), where the
        # persistent rollouts' and the chaser's fault latch is final; they used to run beside / at the end of the FORWARD rollout, i.e.
        # before a fault of this very step could be known, so "the step was not applied" did not hold for them.
        self._defer_commit = False          # set for the duration of _train_step_local
        self._commit_pending = None         # (B, ema: bool) between forward() and _commit_state()
        self._commit_in_apply = False

    # ------------------------------------------------------------------ parallel branches
    @contextlib.contextmanager
    def _branch(self, k: int, bit: Optional[int] = None):
        """Launch the enclosed kernels on side stream k, ordered after everything launched so far on the current stream.
        The branch stays open until _join(k) makes the current stream wait for it.  A branch may only touch buffers (and a
        workspace) that nothing launched on the main stream between the fork and the join touches."""
        if not self._branches_on or not (self.overlap >> (k if bit is None else bit)) & 1:
            yield
            return
        side = self._sides.get(k)
        if side is None:
            side = self._sides[k] = torch.cuda.Stream(device=self.device)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            yield
        self._open.append(k)

    def _fork(self, k: int, fn, late: Optional[bool] = None):
        """Branch k = fn(), ordered after everything launched so far on the current stream -- but LAUNCHED by the next _release(),
        which the caller places behind the main chain's next kernel.  In a captured hipGraph the first successor recorded at a
        fork stays on the hardware queue of the fork node and the others move to another queue, whose first packet pays a
        cross-queue dependency (measured in the replayed step's timeline: ~5 us to the next kernel of the same queue, 10-19 us
        across queues): where nothing else decides, the main chain should be that first successor.  Per branch: self._fork_late."""
        if not self._branches_on or not (self.overlap >> k) & 1:
            fn()
            return
        if not ((self._fork_late >> k) & 1 if late is None else late):
            with self._branch(k):
                fn()
            return
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream())
        self._deferred.append((k, ev, fn))

    def _release(self, behind_current: bool = False):
        """launch the deferred branches; behind_current: ordered after everything launched on the current stream SO FAR (not only
        up to their _fork), for a branch that should not compete with the kernel just launched on the main chain"""
        for k, ev, fn in self._deferred:
            side = self._sides.get(k)
            if side is None:
                side = self._sides[k] = torch.cuda.Stream(device=self.device)
            if behind_current:
                ev = torch.cuda.Event()
                ev.record(torch.cuda.current_stream())
            side.wait_event(ev)
            with torch.cuda.stream(side):
                fn()
            self._open.append(k)
        self._deferred.clear()

    def _join(self, k: Optional[int] = None):
        """current stream waits for branch k (default: every open branch)"""
        assert not self._deferred, "a forked branch was never released"
        for j in [x for x in self._open if k is None or x == k]:
            torch.cuda.current_stream().wait_stream(self._sides[j])
            self._open.remove(j)

    def check_faults(self):
        """The persistent rollout kernels' fault latch (include/g2v.h: a bounded wait of the grid-wide exchange, or of the loss
        chaser, ran out because a workgroup of the launch was not resident).  Synchronous one-word read: call it at a host sync
        point (train_iter does, every iteration, where it reads the loss back).  The kernels that COMMIT a step to the model state
        -- clip + Adam, the EMA codebook update, the BatchNorm running statistics -- read the same latch on the device and leave the
        state untouched when it is set, so a faulted step (and every step replayed until the host notices) changes nothing: the
        persistent path is switched off for the process, the graphs are dropped, and the caller repeats the step."""
        if self.vq_bx_check_every > 0:
            n = self.vq_bx_mismatches()
            if n:
                raise RuntimeError(f"quantiser self-check: the bf16-screened kernel and its exact fp32 sweep disagreed on {n} row(s) "
                                   "(the screening's error radius was violated: please report the codebook / batch)")
        f = int(self.lib.g2v_dec_rollout_persist_fault(0))
        if f != 0:
            self.fault_policy.on_fault()   # clears the latch, selects the per-step kernels IN THIS ENGINE'S CONTEXT; re-arms them later
            self._iter_graph = None
            self._open.clear()
            self._deferred.clear()
            # the per-batch-size buffers cache what the persistent pair fuses (the W_hh1 gradient pointers in the rollout's
            # argument struct, g2v_dec_rollout_bwd_fuses_wgrad): rebuilt on the next call, for the per-step kernels
            self._bufs.clear()
            raise RuntimeError(f"persistent rollout kernel (latch {f}: {'the loss chaser waited for the rollout' if f == 2 else 'the exchange'}): "
                               "a workgroup of the launch was not resident (CU mask / another tenant of "
                               "the device?) -- this step's results are invalid and were NOT applied (parameters, Adam moments, "
                               "codebook, EMA and BatchNorm statistics are as before the step); the per-step kernels are selected "
                               "until the fast path is re-armed (fault_policy.py): repeat the step")

    def rearm(self):
        """the fast path was switched back on (self.fault_policy.tick() returned True): forget what was planned and captured
        for the per-step kernels"""
        self.lib.g2v_cluster_exchange_preclear_drop(None, 0)
        self._iter_graph = None
        self._open.clear()
        self._deferred.clear()
        self._bufs.clear()

    def _branches_ok(self, B: int) -> bool:
        """large-batch regime (the parallel branches are on): what train_iter replays from a hipGraph"""
        if B >= self.overlap_min_rows:
            return True
        return bool(self.lib.g2v_dec_rollout_tiles_per_workgroup(B, self.D, self.H) >= 1 or
                    self.lib.g2v_dec_rollout_cluster_ok(B, self.D, self.H))

    # ------------------------------------------------------------------ parameter views
    def view(self, name: str, grad: bool = False) -> torch.Tensor:
        off, n, shp = self.offsets[name]
        return (self.gflat if grad else self.flat)[off:off + n].view(shp)

    def _w(self, name):
        return self.view(name).data_ptr()

    def _compose_in(self, b, rows: int, drop_in: bool) -> bool:
        """generic dims: in_layer + GRU input projections on the composed weights (in_layer's output is then never formed: only where
        backward_encoder re-associates the W_ih gradients through in_layer, its chain_ih).  Measured (r05_av / r05_aw logs): native
        dims at B = 4096 6.40 -> 6.13 ms, GENEA 3.23 -> 3.16, native B = 128 0.997 -> 0.992; GENEA dims at B = 128 (no input
        dropout: two row-mapped launches, D % 4 != 0) 0.597 -> 0.604, hence the last condition."""
        return (self.compose_in and self.H != 64 and self.D < self.H and not self.wgrad_bf16x3 and b["enc_fused_wgrad"] == 0
                and (rows >= 4096 or (drop_in and self.D % 4 == 0)))

    def _g(self, name):
        return self.view(name, True).data_ptr()

    def dec_wstruct(self) -> DecWeights:
        if self._wstruct is None:
            pre = "decoder.decoder."
            s = DecWeights()
            s.w_pre, s.b_pre = self._w(pre + "pre_linear.0.weight"), self._w(pre + "pre_linear.0.bias")
            s.bn_w, s.bn_b = self._w(pre + "pre_linear.1.weight"), self._w(pre + "pre_linear.1.bias")
            s.bn_running_mean, s.bn_running_var = self.bn_rm.data_ptr(), self.bn_rv.data_ptr()
            for l in (0, 1):
                setattr(s, f"w_ih{l}", self._w(pre + f"gru.weight_ih_l{l}"))
                setattr(s, f"w_hh{l}", self._w(pre + f"gru.weight_hh_l{l}"))
                setattr(s, f"b_ih{l}", self._w(pre + f"gru.bias_ih_l{l}"))
                setattr(s, f"b_hh{l}", self._w(pre + f"gru.bias_hh_l{l}"))
            s.w_out, s.b_out = self._w(pre + "out_layer.weight"), self._w(pre + "out_layer.bias")
            self._wstruct = s
            # the same weights WITHOUT the running statistics: a training rollout given this one does not commit them (the fused
            # train step does, behind the backward rollout: _commit_state)
            d = DecWeights()
            C.memmove(C.byref(d), C.byref(s), C.sizeof(DecWeights))
            d.bn_running_mean = d.bn_running_var = None
            self._wstruct_deferred = d
        return self._wstruct_deferred if self._defer_commit else self._wstruct

    # ------------------------------------------------------------------ buffers
    def buffers(self, B: int) -> dict:
        b = self._bufs.get(B)
        if b is not None:
            return b
        T, D, H, E, K = self.T, self.D, self.H, self.E, self.K
        dev = self.device
        z = lambda *s: torch.zeros(*s, dtype=torch.float32, device=dev)
        u8 = lambda *s: torch.ones(*s, dtype=torch.uint8, device=dev)
        nblk = self.lib.g2v_dec_rollout_blocks(B)
        G = 3 * H
        b = {
            "keep_in": u8(T, B, D) if self.p > 0 else None,
            # the encoder's dropped input self.do(inputs) (:88-93) as a (T,B,D) tensor of its own: in_layer's forward product and
            # its weight gradient then run on their unmasked fast kernels (the masked forms: 79 + 196 us at B = 4096, these 82 + 53)
            "x_drop": z(T * B, D) if self.p > 0 else None,
            "keep95": u8(T - 1, B, D),
            "keep_l0": u8(T - 1, B, H) if self.p > 0 else None,
            "xin": z(T * B, H), "gi_f": z(T * B, G), "gi_b": z(T * B, G),
            # (T+1) slots: slot 0 (forward dir) / slot T (reverse dir) stays zero = h_prev of the first step
            "hs_f": z(T + 1, B, H), "hs_b": z(T + 1, B, H),
            "gates_f": z(T, B, 4 * H), "gates_b": z(T, B, 4 * H),
            "enc_hidden": z(2, B, H),            # encoder_hidden[:2] = (layer-0 fwd, layer-0 bwd) final states
            "flat": z(B, E), "idx": torch.zeros(B, dtype=torch.int64, device=dev), "quant": z(2, B, H),
            "sse": z(self.lib.g2v_vq_assign_blocks(B)),
            "y": z(T, B, D), "dec_xin": z(T - 1, B, D), "u": z(T - 1, B, H), "a": z(T - 1, B, H),
            "h0": z(T, B, H), "h1": z(T, B, H), "x1": z(T - 1, B, H) if self.p > 0 else None,
            "gates0": z(T - 1, B, 4 * H), "gates1": z(T - 1, B, 4 * H),
            "bn_partial": z(2, nblk, 2, H), "bn_stats": z(T - 1, 2, H),
            "loss_partial": z(self.lib.g2v_custom_loss_blocks(B, D) * 4),
            # custom_loss folded into the rollout pair (g2v_dec_saved.loss_*): sign codes + Dropout(0.95) flags, column coefficients
            "loss_code": torch.zeros(T, B, D, dtype=torch.uint8, device=dev), "loss_coef": z(B, D),
            # backward
            "dy": z(T, B, D), "du": z(T - 1, B, H), "dbn": z(T - 1, B, H),
            "dgi0": z(T - 1, B, G), "dgh0": z(T - 1, B, G), "dgi1": z(T - 1, B, G), "dgh1": z(T - 1, B, G),
            "dh_init": z(2, B, H), "bn_bwd_partial": z(2, nblk, 2, H),
            "gz": z(2, B, H),
            "dgi_f": z(T, B, G), "dgh_f": z(T, B, G), "dgi_b": z(T, B, G), "dgh_b": z(T, B, G),
            "dxin": z(T * B, H),
            "wc_in": z(2, 3 * H, D), "bc_in": z(2, 3 * H),     # W_ih W_in and W_ih b_in + b_ih per direction (g2v_linear_compose2)
            "p_in": z(2, 3 * H, D), "c_in": z(2, 3 * H),       # dgi^T x and the column sums of dgi per direction (g2v_linear_bwd_weight_fold2)
        }
        sv = DecSaved()
        sv.y, sv.xin, sv.u, sv.a = _p(b["y"]), _p(b["dec_xin"]), _p(b["u"]), _p(b["a"])
        sv.h0, sv.h1, sv.x1 = _p(b["h0"]), _p(b["h1"]), _p(b["x1"])
        sv.gates0, sv.gates1 = _p(b["gates0"]), _p(b["gates1"])
        sv.bn_partial, sv.bn_stats = _p(b["bn_partial"]), _p(b["bn_stats"])
        b["sv"] = sv
        svl = DecSaved()          # the same arrays + the loss fold's (weights filled in by forward_decoder)
        C.memmove(C.byref(svl), C.byref(sv), C.sizeof(DecSaved))
        svl.loss_code, svl.loss_coef = _p(b["loss_code"]), _p(b["loss_coef"])
        svl.loss_partial, svl.loss_terms = _p(b["loss_partial"]), _p(self.loss_terms)
        b["sv_loss"] = svl
        b["loss_folded"] = False
        sve = DecSaved()          # inference: nothing saved for backward
        sve.y, sve.u, sve.h0, sve.h1, sve.bn_partial = sv.y, sv.u, sv.h0, sv.h1, sv.bn_partial
        b["sv_eval"] = sve
        pre = "decoder.decoder."
        gr = DecGrads()
        gr.dy, gr.du, gr.dbn = _p(b["dy"]), _p(b["du"]), _p(b["dbn"])
        gr.dgi0, gr.dgh0, gr.dgi1, gr.dgh1 = _p(b["dgi0"]), _p(b["dgh0"]), _p(b["dgi1"]), _p(b["dgh1"])
        gr.dh_init = _p(b["dh_init"])
        gr.d_bn_w, gr.d_bn_b = self._g(pre + "pre_linear.1.weight"), self._g(pre + "pre_linear.1.bias")
        gr.bn_bwd_partial = _p(b["bn_bwd_partial"])
        # GRU weight gradients accumulated inside the persistent backward kernel: bit m <-> (ih0, hh0, ih1, hh1)
        b["fused_wgrad"] = int(self.lib.g2v_dec_rollout_bwd_fuses_wgrad(B, D, H))
        names = [("gru.weight_ih_l0", "gru.bias_ih_l0"), ("gru.weight_hh_l0", "gru.bias_hh_l0"),
                 ("gru.weight_ih_l1", "gru.bias_ih_l1"), ("gru.weight_hh_l1", "gru.bias_hh_l1")]
        for m, (wn, bn) in enumerate(names):
            if (b["fused_wgrad"] >> m) & 1:
                gr.dw_gru[m] = self._g(pre + wn)
                gr.db_gru[m] = self._g(pre + bn)
        b["gr"] = gr
        ws_bytes = max(self.lib.g2v_dec_rollout_bwd_workspace(D, H), self.lib.g2v_dec_rollout_fwd_workspace(D, H),
                       self.lib.g2v_gru_seq_bwd_workspace(2, H), self.lib.g2v_gru_seq_fwd_workspace(2, H),
                       self.lib.g2v_vq_stats_workspace(B, E, K),
                       self.lib.g2v_linear_bwd_weight_workspace(T * B, max(D, H), 3 * H),
                       self.lib.g2v_linear_bwd_weight_workspace(T * B, H, max(D, 3 * H)),
                       4 * self.lib.g2v_linear_bwd_weight_workspace(T * B, H, 3 * H),     # batches of four GRU weight gradients
                       2 * self.lib.g2v_linear_bwd_weight_workspace(T * B, D, 3 * H))
        b["ws"] = torch.zeros(ws_bytes, dtype=torch.uint8, device=dev)
        # scratch of the parallel branches (see _branch): never shared with the main chain
        b["ws_stats"] = torch.zeros(max(self.lib.g2v_vq_stats_workspace(B, E, K), 256), dtype=torch.uint8, device=dev)
        wsz = self.lib.g2v_linear_bwd_weight_workspace
        # (deferred reductions: the decoder branch's three calls keep their slabs side by side until the branch's one reduction)
        b["ws_dec_wgrad"] = torch.zeros(4 * wsz(T * B, H, 3 * H) + wsz(T * B, D, H) + wsz(T * B, H, D) + 1024,
                                        dtype=torch.uint8, device=dev)
        # encoder GRU weight gradients accumulated inside its backward kernel (H == 64): 1 = W_hh (0 = separate products and
        # 2 = W_ih as well were measured +50 / +90 us per step in round 3; the kernel keeps both forms, g2v_gru_dir_bwd)
        b["enc_fused_wgrad"] = 1 if H == 64 else 0
        if b["enc_fused_wgrad"]:
            n = int(self.lib.g2v_gru_seq_bwd_wslab_bytes(B, H))
            b["enc_wslab"] = [torch.zeros(n, dtype=torch.uint8, device=dev) for _ in range(2)]
        # (the encoder's: four GRU weight gradients + the input layer's product(s), side by side for the same reason)
        b["ws_enc_wgrad"] = torch.zeros(4 * wsz(T * B, H, 3 * H) + 2 * wsz(T * B, D, max(H, 3 * H)) + 1024, dtype=torch.uint8, device=dev)
        # dedicated workspaces of the four recurrent launches: prepare_recurrent() fills them ahead of their kernels
        for key, nbytes in (("ws_grub", self.lib.g2v_gru_seq_bwd_workspace(2, H)),
                            ("ws_decf", self.lib.g2v_dec_rollout_fwd_workspace(D, H)), ("ws_decb", self.lib.g2v_dec_rollout_bwd_workspace(D, H))):
            b[key] = torch.zeros(max(int(nbytes), 256), dtype=torch.uint8, device=dev)
        self._bufs[B] = b
        return b

    # ------------------------------------------------------------------ pieces
    def _stream(self):
        return torch.cuda.current_stream().cuda_stream

    def prepare_recurrent(self, B: int, which: str):
        """Weight-fragment packs of the encoder GRU's backward (which == "gru_bwd") or of the decoder rollout forward /
        backward plus the clearing of the rollout's two exchange regions (which == "dec"), into their dedicated workspaces
        (g2v_gru_seq_prepare, g2v_dec_rollout_prepare).  Valid until the weights change: the fused train step calls both once
        per step inside the branch that also draws the masks, and then uses the *_prepared entry points.  (The encoder GRU's
        forward runs 40 us into the step: its pack stays in front of it.)"""
        lib, st = self.lib, self._stream()
        b = self.buffers(B)
        enc = "encoder."
        if which in ("gru_bwd", "all"):
            whh = (C.c_void_p * 2)(self._w(enc + "gru.weight_hh_l0"), self._w(enc + "gru.weight_hh_l0_reverse"))
            wih = (C.c_void_p * 2)(self._w(enc + "gru.weight_ih_l0"), self._w(enc + "gru.weight_ih_l0_reverse"))
        if which == "all":      # both rollout workspaces + the BPTT workspace in ONE launch (H == 64, D == 135)
            check(lib.g2v_train_step_prepare(C.byref(self.dec_wstruct()), self.D, self.H, _p(b["ws_decf"]), b["ws_decf"].numel(),
                                             _p(b["ws_decb"]), b["ws_decb"].numel(), whh, wih, 2, 1, _p(b["ws_grub"]),
                                             b["ws_grub"].numel(), st))
        elif which == "gru_bwd":
            check(lib.g2v_gru_seq_prepare(whh, wih, 2, self.H, int(self.H == 64), None, 0, _p(b["ws_grub"]),
                                          b["ws_grub"].numel(), st))
        else:
            check(lib.g2v_dec_rollout_prepare(C.byref(self.dec_wstruct()), self.D, self.H, _p(b["ws_decf"]), b["ws_decf"].numel(),
                                              _p(b["ws_decb"]), b["ws_decb"].numel(), st))

    def draw_masks(self, B: int, training: bool, which: str = "all"):
        """The step's keep masks from the Philox stream (seed, offset counter): the encoder's input mask at counter + 0 (with input
        dropout), the rollout's Dropout(0.95) mask at + 1 (+ 0 without input dropout), the decoder GRU's inter-layer mask at + 2;
        ONE advance of the counter behind them.  which = "all", or "rest": everything but the input mask -- the fused step forms
        the dropped input with the mask drawn inside that kernel (forward_encoder, g2v_dropout_rows: the same stream at + 0)."""
        b = self.buffers(B)
        st = self._stream()
        drop = training and self.p > 0
        lib, ctr = self.lib, _p(self.rng_counter)
        if drop and which == "all":
            check(lib.g2v_keep_mask_at(_p(b["keep_in"]), b["keep_in"].numel(), 1 - self.p, self.seed + 1, ctr, 0, st))
        check(lib.g2v_keep_mask_at(_p(b["keep95"]), b["keep95"].numel(), 0.05, self.seed, ctr, 1 if drop else 0, st))
        if drop:
            check(lib.g2v_keep_mask_at(_p(b["keep_l0"]), b["keep_l0"].numel(), 1 - self.p, self.seed + 2, ctr, 2, st))
        check(lib.g2v_counter_add(ctr, 3 if drop else 1, st))

    def set_masks(self, B: int, keep95, keep_in=None, keep_l0=None):
        """Explicit keep masks (parity tests): keep95 (T-1,B,D), keep_in (T,B,D), keep_l0 (T-1,B,H)."""
        b = self.buffers(B)
        b["keep95"].copy_(keep95)
        if self.p > 0:
            if keep_in is not None:
                b["keep_in"].copy_(keep_in)
            if keep_l0 is not None:
                b["keep_l0"].copy_(keep_l0)

    def forward(self, in_poses: torch.Tensor, out_poses: torch.Tensor, training: bool, ema_update: bool = True,
                n_global: Optional[int] = None, derived_ready: bool = False, loss_w=None, join_stats: bool = True):
        """Autoencoder_VQVAE.forward.  in_poses/out_poses (B,T,D) contiguous fp32 on the GPU.
        Fills buffers: y (T,B,D), quant (2,B,H) first hidden, idx, vq_scalars (loss_vq, perplexity)."""
        if self.quantizer == "gssoft":
            return self._forward_gssoft(in_poses, out_poses, training, loss_w)
        lib, st = self.lib, self._stream()
        B = in_poses.shape[0]
        H, E, K = self.H, self.E, self.K
        b = self.forward_encoder(in_poses, training)
        # ---- VQ_Payam_EMA (:1217-1296) on decoder_hidden.view(-1, E) ---------------------------------------
        N = (2 * B * H) // E
        self._join(0)                       # branch 0: keep masks, ahead-of-time packs and (derived_ready) vq_derive()
        if not derived_ready:
            self.vq_derive()
        if self._vq_bx:
            # pre_linear (fp32 MFMA) + distances screened on the bf16 pipe + exact fp32 re-evaluation of every candidate +
            # argmin + straight-through / SSE in one launch (flat is written for the statistics)
            check(lib.g2v_vq_fused_assign_bx_fwd(_p(b["enc_hidden"]), _p(self.vq_wpre_frag), _p(self.vq_pre_b), _p(self.codebook),
                                                 _p(self.vq_bx_image), _p(self.code_sqnorm), _p(b["flat"]), _p(b["idx"]),
                                                 _p(b["quant"]), _p(b["sse"]), _p(self.vq_diag) if self._vq_diag_on else None,
                                                 N, E, K, self.vq_bx_flags, st))
            if training and self.vq_bx_check_every > 0 and self._steps % self.vq_bx_check_every == 0:
                self._vq_bx_selfcheck(b, N)
        elif self.codebook_frag is not None:
            # pre_linear + distances + argmin + straight-through / SSE in one launch (flat is written for the statistics)
            check(lib.g2v_vq_fused_assign_packed_fwd(_p(b["enc_hidden"]), _p(self.vq_pre_w), _p(self.vq_pre_b),
                                                     _p(self.codebook), _p(self.codebook_frag), _p(self.code_sqnorm),
                                                     _p(b["flat"]), _p(b["idx"]), _p(b["quant"]), _p(b["sse"]), N, E, K, st))
        else:
            check(lib.g2v_linear_fwd(_p(b["enc_hidden"]), E, 0, 0, 0, None, 1.0, _p(self.vq_pre_w), _p(self.vq_pre_b),
                                     _p(b["flat"]), E, N, E, E, 0, st))
            if self.codebook_frag_generic is not None and N >= ops.VQ_PACKED_MIN_ROWS:
                check(lib.g2v_vq_assign_packed_fwd(_p(b["flat"]), _p(b["enc_hidden"]), _p(self.codebook), _p(self.codebook_frag_generic),
                                                   _p(self.code_sqnorm), _p(b["idx"]), _p(b["quant"]), None, _p(b["sse"]), N, E, K, st))
            else:
                check(lib.g2v_vq_assign_fwd(_p(b["flat"]), _p(b["enc_hidden"]), _p(self.codebook), _p(self.code_sqnorm),
                                            _p(b["idx"]), _p(b["quant"]), None, _p(b["sse"]), N, E, K, st))
        # The statistics and the codebook update feed nothing in this forward (the rollout starts from `quant`, taken from the
        # codebook as it was): a parallel branch beside the rollout.  The custom_loss CHASER rides at the end of the same branch:
        # it must be dispatched behind the rollout (g2v.h), and these ~30 us of small kernels -- every one of them fits beside the
        # rollout's one wave per SIMD -- put it there.
        chase = self._chase_ok(B, training, loss_w)
        if chase:
            b["sv_loss"].loss_w[0], b["sv_loss"].loss_w[1], b["sv_loss"].loss_w[2] = (float(w) for w in loss_w)
            b["loss_target"] = out_poses.data_ptr()
        def stats():
            check(lib.g2v_vq_stats(_p(b["idx"]), _p(b["flat"]), _p(self.vq_stats), N, E, K, _p(b["ws_stats"]),
                                   b["ws_stats"].numel(), self._stream()))
            if ema_update:
                self.vq_finish(B, training, n_global)
        def chaser():
            check(lib.g2v_custom_loss_chase(_p(out_poses), C.byref(b["sv_loss"]), _p(b["keep95"]), self.T, B, self.D, self.H,
                                            _p(b["ws_decf"]), b["ws_decf"].numel(), self._stream()))
        self._fork(1, stats)
        if chase:
            # LATE: enqueued by _release(), i.e. behind the rollout in host order.  Two HIP streams may share one hardware queue
            # (they are dealt round-robin onto a few); a chaser enqueued AHEAD of the rollout on a shared queue would wait for a
            # kernel that cannot start behind it (seen: the bounded wait ran out, latch 2).  Behind it, sharing a queue only
            # costs the overlap.
            self._fork(1, chaser, late=True)
        b = self.forward_decoder(out_poses, B, training, chase=chase)
        self._release()
        if join_stats or chase:             # (the backward rollout reads what the chaser writes)
            self._join(1)
        return b

    def _vq_bx_selfcheck(self, b, N):
        """Debug switch vq_bx_check_every = n (round-3 verdict: the screening's exactness rests on a hand-budgeted error radius, and
        a violated bound would be a SILENT argmin mismatch): on every n-th training step the same rows are assigned once more with
        the kernel's exact fp32 sweep over all codes (G2V_VQ_BX_EXACT, its own kernel symbol) into scratch buffers, and the number
        of rows whose index differs is added to a device counter.  vq_bx_mismatches() reads it (host sync); check_faults() raises
        when it is non-zero.  Costs one extra 16 us launch on the checked steps; the checked steps are launched eagerly
        (train_iter does not replay a hipGraph while the switch is on)."""
        lib, E, K = self.lib, self.E, self.K
        sc = b.get("bx_check")
        if sc is None:
            dev = self.device
            sc = b["bx_check"] = (torch.empty(N, E, device=dev), torch.empty(N, dtype=torch.int64, device=dev),
                                  torch.empty(N, E, device=dev), torch.empty_like(b["sse"]))
        check(lib.g2v_vq_fused_assign_bx_fwd(_p(b["enc_hidden"]), _p(self.vq_wpre_frag), _p(self.vq_pre_b), _p(self.codebook),
                                             _p(self.vq_bx_image), _p(self.code_sqnorm), _p(sc[0]), _p(sc[1]), _p(sc[2]), _p(sc[3]),
                                             None, N, E, K, 1, self._stream()))
        self._vq_bx_mismatch += (sc[1] != b["idx"]).sum()

    def vq_bx_mismatches(self) -> int:
        """rows on which the screened quantiser kernel and its exact sweep disagreed so far (vq_bx_check_every; host sync)"""
        return int(self._vq_bx_mismatch.item())

    def _chase_ok(self, B: int, training: bool, loss_w) -> bool:
        """custom_loss by the chaser (self.loss_chase): a training forward that is told the loss weights, inside the fused step's
        parallel-branch regime (the chaser needs the side stream and the prepared workspace whose progress words
        g2v_dec_rollout_prepare has cleared), where the persistent pair runs."""
        return bool(training and loss_w is not None and self.loss_chase and self._prepared and self._branches_on and
                    (self.overlap >> 1) & 1 and self.lib.g2v_dec_rollout_fuses_loss(B, self.D, self.H, self.T))

    def forward_encoder(self, in_poses: torch.Tensor, training: bool):
        """EncoderRNN (:73-100): in_layer, then layer-0 of the bidirectional GRU.  Fills buffers['enc_hidden'] (2,B,H) =
        the layer-0 forward / backward final states, i.e. encoder_hidden[:L] of the reference (:971-973)."""
        lib, st = self.lib, self._stream()
        B, T, D = in_poses.shape
        assert T == self.T and D == self.D, "shape does not match the engine"
        ops._chk(in_poses, name="in_poses")
        H, G = self.H, 3 * self.H
        b = self.buffers(B)
        drop_in = training and self.p > 0
        enc = "encoder."
        # ---- EncoderRNN (:73-100): in_layer, then layer-0 of the bidirectional GRU -----------------------
        b["enc_dropped"] = drop_in
        if drop_in and self._fused_in_drop:          # mask drawn inside the kernel (the fused step; draw_masks("rest") follows)
            check(lib.g2v_dropout_rows(_p(in_poses), D, B, D, T * D, 1 - self.p, 1.0 / (1.0 - self.p), self.seed + 1,
                                       _p(self.rng_counter), 0, _p(b["x_drop"]), D, T * B, D, st))
        elif drop_in:
            check(lib.g2v_mask_rows(_p(in_poses), D, B, D, T * D, _p(b["keep_in"]), 1.0 / (1.0 - self.p), _p(b["x_drop"]), D,
                                    T * B, D, st))
        # work of branch 0 (operand images of the quantiser, ahead-of-time packs, rollout masks).  Round 5 (side_early): forked
        # HERE, in front of the input layer, and ordered so that the HBM-heavy mask kernel comes LAST -- the small MFMA / pack
        # kernels run beside the (HBM-bound) input layer, the masks beside the (latency-bound) GRU.  Up to round 4 the whole
        # branch was forked behind the input layer with the masks in the middle: beside the input layer the mask kernel doubled
        # that kernel's time (36 -> 67 us on the main chain), but behind it the branch (163 us of kernels) outlasted the 137 us
        # GRU and the quantiser waited 36 us for it (profiles/r04_h_step_timeline.txt).
        side, self._side_work = getattr(self, "_side_work", None), None
        early = side is not None and self.side_early
        if early:
            self._fork(0, side)
        # Generic dims (D < H): in_layer and the GRU's input projections as ONE layer on the composed weights -- in_layer's output is
        # never formed (its only other reader, the W_ih gradient, is re-associated through in_layer too: backward_encoder's chain_ih)
        compose = self._compose_in(b, T * B, drop_in)
        # in_layer (:93)
        if compose:
            wc, bc = b["wc_in"], b["bc_in"]
            check(lib.g2v_linear_compose2(self._w(enc + "gru.weight_ih_l0"), self._w(enc + "gru.bias_ih_l0"),
                                          self._w(enc + "gru.weight_ih_l0_reverse"), self._w(enc + "gru.bias_ih_l0_reverse"),
                                          self._w(enc + "in_layer.weight"), self._w(enc + "in_layer.bias"), wc[0].data_ptr(),
                                          bc[0].data_ptr(), wc[1].data_ptr(), bc[1].data_ptr(), G, H, D, st))
        elif drop_in:
            check(lib.g2v_linear_fwd(_p(b["x_drop"]), D, 0, 0, 0, None, 1.0, self._w(enc + "in_layer.weight"),
                                     self._w(enc + "in_layer.bias"), _p(b["xin"]), H, T * B, D, H, 0, st))
        else:
            check(lib.g2v_linear_fwd(_p(in_poses), D, B, D, T * D, None, 1.0,
                                     self._w(enc + "in_layer.weight"), self._w(enc + "in_layer.bias"),
                                     _p(b["xin"]), H, T * B, D, H, 0, st))
        if early:
            self._release()                 # launched behind the input layer in host order (the main chain keeps its queue)
        elif side is not None:
            self._fork(0, side)
        # H == 64: the input projections x W_ih^T + b_ih are fused into the recurrent kernel (no gi array at all);
        # other sizes compute gi with the dense-layer kernel first
        fuse_gi = (H == 64)
        if compose:
            if drop_in:
                check(lib.g2v_linear_fwd_pair(_p(b["x_drop"]), D, wc[0].data_ptr(), bc[0].data_ptr(), _p(b["gi_f"]), wc[1].data_ptr(),
                                              bc[1].data_ptr(), _p(b["gi_b"]), G, T * B, D, G, 0, st))
            else:
                for k, key in enumerate(("gi_f", "gi_b")):
                    check(lib.g2v_linear_fwd(_p(in_poses), D, B, D, T * D, None, 1.0, wc[k].data_ptr(), bc[k].data_ptr(), _p(b[key]), G,
                                             T * B, D, G, 0, st))
        elif not fuse_gi:
            check(lib.g2v_linear_fwd_pair(_p(b["xin"]), H, self._w(enc + "gru.weight_ih_l0"), self._w(enc + "gru.bias_ih_l0"),
                                          _p(b["gi_f"]), self._w(enc + "gru.weight_ih_l0_reverse"),
                                          self._w(enc + "gru.bias_ih_l0_reverse"), _p(b["gi_b"]), G, T * B, H, G, 0, st))
        dirs = (_lib.GruDir * 2)()
        for k, (suf, key, hs_ptr) in enumerate((("", "f", b["hs_f"][1:].data_ptr()), ("_reverse", "b", b["hs_b"].data_ptr()))):
            dirs[k].gi = None if fuse_gi else _p(b["gi_" + key])
            dirs[k].x = _p(b["xin"])
            dirs[k].w_ih = self._w(enc + "gru.weight_ih_l0" + suf)
            dirs[k].b_ih = self._w(enc + "gru.bias_ih_l0" + suf)
            dirs[k].in_dim = H
            dirs[k].w_hh = self._w(enc + "gru.weight_hh_l0" + suf)
            dirs[k].b_hh = self._w(enc + "gru.bias_hh_l0" + suf)
            dirs[k].h0 = None
            dirs[k].hs = hs_ptr
            dirs[k].h_n = b["enc_hidden"][k].data_ptr()
            dirs[k].gates = _p(b["gates_" + key]) if training else None
            dirs[k].reverse = k
        check(lib.g2v_gru_seq_fwd(dirs, 2, None, H, T, B, H, _p(b["ws"]), b["ws"].numel(), st))
        self._release()                     # branch 0 (forked above) is launched behind the main chain's kernel
        return b

    def forward_decoder(self, out_poses: torch.Tensor, B: int, training: bool, chase: bool = False):
        """decoder rollout (:1039-1054) from buffers['quant'] (2,B,H) = the initial hidden state; fills buffers['y'] (T,B,D).
        chase (forward() decides): the rollout hands every y_t over to the custom_loss chaser that forward() has forked;
        buffers['loss_folded'] then says so, loss() launches nothing and loss_terms is written by backward_decoder's launch."""
        lib, st = self.lib, self._stream()
        T, D, H = self.T, self.D, self.H
        ops._chk(out_poses, name="out_poses")
        assert tuple(out_poses.shape) == (B, T, D), "shape does not match the engine"
        b = self.buffers(B)
        drop_in = training and self.p > 0
        fn, ws = (lib.g2v_dec_rollout_fwd_prepared, b["ws_decf"]) if self._prepared else (lib.g2v_dec_rollout_fwd, b["ws"])
        if self._xch_pre and training:
            ws = b["ws_decf"]                  # its exchange records were cleared on branch 0 (side())
        fold = b["loss_folded"] = bool(training and chase)
        check(fn(_p(out_poses), _p(b["quant"]), C.byref(self.dec_wstruct()),
                 C.byref((b["sv_loss"] if fold else b["sv"]) if training else b["sv_eval"]), _p(b["keep95"]),
                 _p(b["keep_l0"]) if drop_in else None, self.p, self.n_pre,
                 int(self.conditioned), int(training), T, B, D, H, _p(ws), ws.numel(), st))
        return b

    def vq_derive(self):
        """Everything the assign kernels read that is derived from (codebook, pre_linear): ||W_k||^2, and the operand images of
        the fused kernel in use.  Launched on the current stream (inside branch 0 of the fused train step)."""
        lib, st = self.lib, self._stream()
        K, E = self.K, self.E
        check(lib.g2v_vq_code_sqnorm(_p(self.codebook), _p(self.code_sqnorm), K, E, st))
        if self._vq_bx:
            check(lib.g2v_vq_pack_codebook(_p(self.vq_pre_w), _p(self.vq_wpre_frag), E, E, st))
            check(lib.g2v_vq_bx_pack(_p(self.codebook), _p(self.code_sqnorm), _p(self.vq_pre_w), _p(self.vq_pre_b),
                                     _p(self.vq_bx_image), K, E, st))
        elif self.codebook_frag is not None:
            check(lib.g2v_vq_pack_codebook(_p(self.codebook), _p(self.codebook_frag), K, E, st))
        elif self.codebook_frag_generic is not None:
            check(lib.g2v_vq_pack_codebook(_p(self.codebook), _p(self.codebook_frag_generic), K, E, st))

    def refresh_codebook_state(self):
        """kept for callers of round 2: the derived state is recomputed by every entry point now"""
        self.vq_derive()

    def vq_finish(self, B: int, training: bool, n_global: Optional[int] = None):
        """K4 + the loss / perplexity scalars; under data parallelism call it after vq_stats has been all-reduced."""
        b = self.buffers(B)
        N = (2 * B * self.H) // self.E
        check(self.lib.g2v_vq_ema_update(_p(self.vq_stats), _p(b["sse"]), b["sse"].numel(), _p(self.ema_cs),
                                         _p(self.ema_w), _p(self.codebook), _p(self.code_sqnorm), _p(self.vq_scalars),
                                         N, n_global or N, self.E, self.K, self.beta, self.decay, self.eps,
                                         int(training), self._stream()))

    def loss(self, B: int, target: torch.Tensor, w_l1: float, w_cont: float, w_var: float, want_grad: bool = True):
        """custom_loss on the rollout output; fills loss_terms and (want_grad) the dy buffer with dLoss/dy."""
        b = self.buffers(B)
        if b["loss_folded"]:           # forward(loss_w=...) sent the chaser along: backward_decoder finishes the loss
            w = b["sv_loss"].loss_w
            if not (want_grad and target.data_ptr() == b["loss_target"] and
                    (w[0], w[1], w[2]) == tuple(C.c_float(float(v)).value for v in (w_l1, w_cont, w_var))):
                raise ValueError("loss(): the forward was run with the loss chaser (forward(loss_w=...)) "
                                 "for another target / other weights, or without the gradient")
            return
        check(self.lib.g2v_custom_loss_fwd_bwd(_p(b["y"]), _p(target), _p(b["dy"]) if want_grad else None,
                                               _p(self.loss_terms), _p(b["loss_partial"]), w_l1, w_cont, w_var, 1.0,
                                               self.T, B, self.D, self._stream()))

    def backward(self, in_poses: torch.Tensor, B: int, g_loss_vq: Optional[torch.Tensor] = None):
        """Backward of forward(training=True): expects buffers['dy'] = dLoss/d y (T,B,D).  Writes every parameter
        gradient into the flat grad buffer (overwrite, not accumulate)."""
        if self.quantizer == "gssoft":
            return self._backward_gssoft(in_poses, B)
        H, E = self.H, self.E
        b = self.backward_decoder(B, wgrad_branch=True)
        # ---- quantiser backward: straight-through + commitment (:1285-1292) ------------------------------------
        N = (2 * B * H) // E
        gl = g_loss_vq if g_loss_vq is not None else self.g_loss_vq
        if self._fuse_vq_bwd and (H == 64 or self.lib.g2v_gru_seq_cluster_ok(self.T, B, H, 2)):
            # straight-through + commitment gradient formed by the BPTT kernel where it reads its incoming gradient: a 4 us kernel
            # that took 12-15 us beside the decoder's products, plus a kernel boundary, off the critical chain
            self._release()                 # branch 2 (the decoder's weight gradients, forked in backward_decoder)
            self.backward_encoder(in_poses, B, quant_bwd=(gl, 2.0 * self.beta / (float(N) * float(E))))
        else:
            check(self.lib.g2v_vq_bwd(_p(b["dh_init"]), _p(gl), _p(b["enc_hidden"]), _p(b["quant"]), None, _p(b["gz"]), N, E,
                                      self.beta, self._stream()))
            self._release()
            self.backward_encoder(in_poses, B)
        self._join(2)

    # ------------------------------------------------------------------ the as-shipped soft quantiser, without autograd
    def _gs_buffers(self, B: int) -> dict:
        b = self.buffers(B)
        if "gs_probs" not in b:
            N, E, K, dev = (2 * B * self.H) // self.E, self.E, self.K, self.device
            z = lambda *s: torch.zeros(*s, dtype=torch.float32, device=dev)
            b.update({"gs_flat": z(N, E), "gs_logvar": z(N, K), "gs_dist": z(N, K), "gs_probs": z(N, K), "gs_q": z(N, E),
                      "gs_dq": z(N, E), "gs_dprobs": z(N, K), "gs_dd": z(N, K), "gs_dlv": z(N, K), "gs_rowsum": z(N),
                      "gs_t": z(N, E), "gs_dflat": z(N, E), "gs_tw": z(K, E), "gs_colsum": z(K), "gs_mse": z(1),
                      "gs_mse_partial": z(max(self.lib.g2v_mse_blocks(N * E), self.lib.g2v_vq_soft_fused_blocks(N))),
                      "gs_tp": z(K, E),
                      "gs_ws": torch.zeros(max(3 * self.lib.g2v_linear_bwd_weight_workspace(N, E, K),
                                               self.lib.g2v_linear_bwd_weight_workspace(N, E, E),
                                               self.lib.g2v_vq_soft_perplexity_workspace(N, K), 256), dtype=torch.uint8, device=dev)})
        return b

    def _forward_gssoft(self, in_poses, out_poses, training, loss_w):
        """Autoencoder_VQVAE.forward with VQ_Payam_GSSoft (reference :816-820, class :1304-1438) as one kernel sequence:
        mean_layer -> logvar_layer, distances -> soft assignment probabilities (+ perplexity) -> q = probs W ->
        loss_vq = q_latent + beta e_latent, straight-through value -> rollout.  The arithmetic (and the kernels) of the module
        path gesture2vec_amd/model/Autoencoder_VQVAE_model.py: VQ_Payam_GSSoft.forward, no autograd graph."""
        lib, st = self.lib, self._stream()
        B = in_poses.shape[0]
        H, E, K = self.H, self.E, self.K
        N = (2 * B * H) // E
        b = self.forward_encoder(in_poses, training)
        self._join(0)
        g = self._gs_buffers(B)
        vq = "vq_layer."
        W = self._w(vq + "_embedding.weight")
        x = _p(b["enc_hidden"])
        if lib.g2v_vq_soft_fused_ok(N, E, K):
            # round 4: the whole quantiser forward as ONE launch (csrc/vq_soft.hip) + a one-workgroup finish (loss mean,
            # perplexity) beside the rollout; the separate kernels below remain for the shapes it does not serve
            if not getattr(self, "_gs_wsq_ready", False):      # (train_step computes it in the branch beside the encoder)
                check(lib.g2v_vq_code_sqnorm(W, _p(self.code_sqnorm), K, E, st))
            self._gs_wsq_ready = False
            check(lib.g2v_vq_soft_fused_fwd(x, self._w(vq + "mean_layer.weight"), self._w(vq + "mean_layer.bias"),
                                            self._w(vq + "logvar_layer.weight"), self._w(vq + "logvar_layer.bias"), W,
                                            _p(self.code_sqnorm), _p(g["gs_flat"]), _p(g["gs_logvar"]), _p(g["gs_dist"]),
                                            _p(g["gs_probs"]), _p(g["gs_q"]), _p(g["gs_dq"]), _p(b["quant"]), _p(g["gs_mse_partial"]),
                                            _p(g["gs_ws"]), float(self._g_vq_host), N, E, K, st))

            def finish():
                check(lib.g2v_vq_soft_finish(_p(g["gs_mse_partial"]), _p(g["gs_ws"]), _p(self._one_plus_beta), _p(g["gs_mse"]),
                                             _p(self.vq_scalars), self.vq_scalars[1:].data_ptr(), N, E, K, self._stream()))
            self._fork(1, finish)
            # custom_loss by the chaser beside the rollout, exactly as forward() sends it along (see there)
            chase = self._chase_ok(B, training, loss_w)
            if chase:
                b["sv_loss"].loss_w[0], b["sv_loss"].loss_w[1], b["sv_loss"].loss_w[2] = (float(w) for w in loss_w)
                b["loss_target"] = out_poses.data_ptr()

                def chaser():
                    check(lib.g2v_custom_loss_chase(_p(out_poses), C.byref(b["sv_loss"]), _p(b["keep95"]), self.T, B, self.D, self.H,
                                                    _p(b["ws_decf"]), b["ws_decf"].numel(), self._stream()))
                self._fork(1, chaser, late=True)
            b = self.forward_decoder(out_poses, B, training, chase=chase)
            self._release()
            self._join(1)
            return b
        check(lib.g2v_linear_fwd(x, E, 0, 0, 0, None, 1.0, self._w(vq + "mean_layer.weight"), self._w(vq + "mean_layer.bias"),
                                 _p(g["gs_flat"]), E, N, E, E, 0, st))
        check(lib.g2v_linear_fwd(_p(g["gs_flat"]), E, 0, 0, 0, None, 1.0, self._w(vq + "logvar_layer.weight"),
                                 self._w(vq + "logvar_layer.bias"), _p(g["gs_logvar"]), K, N, E, K, 0, st))
        check(lib.g2v_vq_code_sqnorm(W, _p(self.code_sqnorm), K, E, st))
        check(lib.g2v_linear_fwd(_p(g["gs_flat"]), E, 0, 0, 0, None, 1.0, W, None, _p(g["gs_dist"]), K, N, E, K, 0, st))
        check(lib.g2v_vq_soft_fwd(_p(g["gs_flat"]), _p(g["gs_dist"]), _p(g["gs_logvar"]), _p(self.code_sqnorm), _p(g["gs_probs"]),
                                  None, N, E, K, st))
        # the perplexity of the mean assignment feeds nothing but the log line: a branch beside the rollout (round 4; it sat on
        # the chain between the encoder and the rollout, two launches)
        def perplexity():
            check(lib.g2v_vq_soft_perplexity(_p(g["gs_probs"]), self.vq_scalars[1:].data_ptr(), N, K, _p(g["gs_ws"]),
                                             g["gs_ws"].numel(), self._stream()))
        self._fork(1, perplexity)
        check(lib.g2v_linear_bwd_data(_p(g["gs_probs"]), K, W, _p(g["gs_q"]), E, N, E, K, 0, st))          # q = probs W (:1417-1419)
        # both latent losses are mean((x - q)^2) (:1424-1425); their gradients go to x and to q separately (backward)
        check(lib.g2v_mse_fwd_bwd(_p(g["gs_q"]), x, _p(g["gs_dq"]), _p(g["gs_mse"]), _p(g["gs_mse_partial"]), N * E,
                                  float(self._g_vq_host), st))
        check(lib.g2v_scale_f32(_p(g["gs_mse"]), _p(self._one_plus_beta), _p(self.vq_scalars), 1, st))      # loss_vq (:1427)
        check(lib.g2v_ste_f32(x, _p(g["gs_q"]), _p(b["quant"]), N * E, st))                                 # inputs + (q - inputs).detach()
        b = self.forward_decoder(out_poses, B, training)
        self._release()
        self._join(1)
        return b

    def _backward_gssoft(self, in_poses, B):
        """Backward of _forward_gssoft (expects buffers['dy']): the module path's autograd functions (_STEFn, _ProbsCodebookFn,
        _SoftAssignFn, the two nn.Linear) as one kernel sequence; every parameter gradient into the flat grad buffer."""
        lib, st = self.lib, self._stream()
        H, E, K = self.H, self.E, self.K
        N = (2 * B * H) // E
        fused = bool(lib.g2v_vq_soft_fused_ok(N, E, K))
        b = self.backward_decoder(B, wgrad_branch=True, wgrad_late=True if fused else None)
        g = self._gs_buffers(B)
        vq = "vq_layer."
        W, gW = self._w(vq + "_embedding.weight"), self._g(vq + "_embedding.weight")
        ws, wsn = _p(g["gs_ws"]), g["gs_ws"].numel()
        x = _p(b["enc_hidden"])
        if not fused:
            self._release()
        if fused:
            # ONE launch: straight-through + commitment term, dprobs, the (distance, logvar) -> probs backward, dflat, and
            # gz = ... + dflat W_mean (csrc/vq_soft.hip); dd / dlogvar / dflat are written for the weight gradients
            check(lib.g2v_vq_soft_fused_bwd(_p(b["dh_init"]), _p(self._g_vq_dev), x, _p(g["gs_q"]), _p(g["gs_dq"]), _p(g["gs_flat"]),
                                            _p(g["gs_probs"]), _p(g["gs_dist"]), _p(g["gs_logvar"]), self._w(vq + "mean_layer.weight"),
                                            self._w(vq + "logvar_layer.weight"), W, _p(g["gs_dd"]), _p(g["gs_dlv"]),
                                            _p(g["gs_dflat"]), _p(b["gz"]), self.beta, N, E, K, st))
            # the decoder's weight-gradient branch (forked in backward_decoder) starts BEHIND this kernel: it sits on the chain to
            # the encoder's BPTT, they do not (beside a product that fills every CU it took 124 us instead of ~45)
            self._release(behind_current=True)
        else:
            self._backward_gssoft_chain(b, g, N, x, W, st)
        # The quantiser's five weight-gradient launches feed only clip + Adam: beside the encoder's BPTT, behind the decoder's
        # products on their stream (round 4; they sat on the chain in front of the BPTT).  Everything they read is final here.
        def q_wgrads():
            st2 = self._stream()
            # the three (K x E) products -- dd^T flat (+ column sums of dd), probs^T dq, dlogvar^T flat (+ its bias gradient) -- have
            # one shape: ONE launch + one slab reduction (round 4; three of each before)
            arr = (_lib.WgradItem * 3)()
            arr[0].dy, arr[0].x, arr[0].dw, arr[0].db = _p(g["gs_dd"]), _p(g["gs_flat"]), _p(g["gs_tw"]), _p(g["gs_colsum"])
            arr[1].dy, arr[1].x, arr[1].dw, arr[1].db = _p(g["gs_probs"]), _p(g["gs_dq"]), _p(g["gs_tp"]), None
            arr[2].dy, arr[2].x, arr[2].dw, arr[2].db = (_p(g["gs_dlv"]), _p(g["gs_flat"]), self._g(vq + "logvar_layer.weight"),
                                                         self._g(vq + "logvar_layer.bias"))
            check(lib.g2v_linear_bwd_weight_batch(arr, 3, K, E, N, E, K, 0, ws, wsn, st2))
            check(lib.g2v_rowscale_combine(W, _p(g["gs_colsum"]), _p(g["gs_tw"]), gW, K, E, st2))           # 2 W sum_n dd - 2 dd^T f
            check(lib.g2v_add_halves(gW, E, _p(g["gs_tp"]), E, gW, E, K, E, st2))                            # += probs^T dq
            check(lib.g2v_linear_bwd_weight(_p(g["gs_dflat"]), E, x, E, 0, 0, 0, None, 1.0, self._g(vq + "mean_layer.weight"),
                                            self._g(vq + "mean_layer.bias"), N, E, E, 0, ws, wsn, st2))
        self._fork(2, q_wgrads, late=False)
        if not fused:
            check(lib.g2v_linear_bwd_data(_p(g["gs_dflat"]), E, self._w(vq + "mean_layer.weight"), _p(b["gz"]), E, N, E, E, 1, st))
        self.backward_encoder(in_poses, B)
        self._join(2)

    def _backward_gssoft_chain(self, b, g, N, x, W, st):
        """the quantiser's data gradients as separate launches (shapes csrc/vq_soft.hip does not serve); gz lacks the mean_layer
        term, which the caller adds"""
        lib, E, K, vq = self.lib, self.E, self.K, "vq_layer."
        # to the encoder state directly: the straight-through path + the commitment term (gs_dq holds g 2 (q - x) / n)
        check(lib.g2v_vq_bwd(_p(b["dh_init"]), _p(self._g_vq_dev), x, _p(g["gs_q"]), None, _p(b["gz"]), N, E, self.beta, st))
        # q = probs W  <-  dq (the q_latent term):  dprobs = dq W^T,  dW += probs^T dq
        check(lib.g2v_linear_fwd(_p(g["gs_dq"]), E, 0, 0, 0, None, 1.0, W, None, _p(g["gs_dprobs"]), K, N, E, K, 0, st))
        # probabilities <- distances, logvar (reference :1349-1372, 1396-1411)
        check(lib.g2v_vq_soft_bwd(_p(g["gs_probs"]), _p(g["gs_dprobs"]), _p(g["gs_dist"]), _p(g["gs_logvar"]), _p(g["gs_dd"]),
                                  _p(g["gs_dlv"]), _p(g["gs_rowsum"]), N, K, st))
        check(lib.g2v_linear_bwd_data(_p(g["gs_dd"]), K, W, _p(g["gs_t"]), E, N, E, K, 0, st))
        check(lib.g2v_rowscale_combine(_p(g["gs_flat"]), _p(g["gs_rowsum"]), _p(g["gs_t"]), _p(g["gs_dflat"]), N, E, st))
        check(lib.g2v_linear_bwd_data(_p(g["gs_dlv"]), K, self._w(vq + "logvar_layer.weight"), _p(g["gs_dflat"]), E, N, E, K, 1, st))

    def _wgrad_fns(self, b, M_default, ws_key="ws", deferred: Optional[_DeferredReductions] = None):
        """deferred: the products leave their slab reductions to deferred.flush() (and take their workspace from it)"""
        lib = self.lib
        ws, wsn = _p(b[ws_key]), b[ws_key].numel()
        G, H = 3 * self.H, self.H
        flags = 2 if self.wgrad_bf16x3 else 0

        def wgrad(dy, lddy, x, ldx, wname, bname, N_, K_, rows=M_default, row_map=(0, 0, 0), keep=None, scale=1.0):
            if deferred is not None and keep is None:
                arr = (_lib.WgradItem * 1)()
                arr[0].dy, arr[0].x, arr[0].dw, arr[0].db = dy, x, self._g(wname), self._g(bname) if bname else None
                deferred.call(arr, 1, lddy, ldx, row_map, None, rows, K_, N_, flags)
                return
            check(lib.g2v_linear_bwd_weight(dy, lddy, x, ldx, row_map[0], row_map[1], row_map[2], keep, scale,
                                            self._g(wname), self._g(bname) if bname else None, rows, K_, N_,
                                            flags, ws, wsn, self._stream()))

        def wgrad4(rows, items):
            """four (3H x H) GRU weight gradients of one shape in ONE launch + one slab reduction"""
            arr = (_lib.WgradItem * 4)()
            for k, (dy, x, wname, bname) in enumerate(items):
                arr[k].dy, arr[k].x, arr[k].dw, arr[k].db = dy, x, self._g(wname), self._g(bname)
            if deferred is not None:
                deferred.call(arr, len(items), G, H, (0, 0, 0), None, rows, H, G, flags)
                return
            check(lib.g2v_linear_bwd_weight_batch(arr, len(items), G, H, rows, H, G, flags, ws, wsn, self._stream()))
        return wgrad, wgrad4

    def _commit_state(self, n_global: Optional[int] = None):
        """The fused train step's deferred commits (see __init__): EMA codebook update + loss / perplexity scalars (single GPU;
        under data parallelism train_step_apply runs it behind the all-reduce) and BatchNorm's running statistics from the
        rollout's saved batch statistics.  Both kernels are gated on the persistent rollouts' fault latch on the device.  Launched
        on the current stream: backward_decoder calls it at the end of its weight-gradient branch, i.e. behind the backward rollout."""
        if self._commit_pending is None:
            return
        B, ema = self._commit_pending
        self._commit_pending = None
        b = self.buffers(B)
        if ema:
            self.vq_finish(B, True, n_global=n_global)
        check(self.lib.g2v_bn_running_update(_p(b["bn_stats"]), _p(self.bn_rm), _p(self.bn_rv), self.T - 1, self.H, B, self._stream()))

    def backward_decoder(self, B: int, wgrad_branch: bool = False, wgrad_late: Optional[bool] = None):
        """Backward of forward_decoder(training=True): expects buffers['dy'] = dLoss/d y (T,B,D); writes the decoder's
        parameter gradients (overwrite) and buffers['dh_init'] (2,B,H) = dLoss / d(initial hidden state).
        wgrad_branch: the weight-gradient products are launched as parallel branch 2 (own workspace); the caller joins it
        (backward() does, after the encoder's backward has been launched on the main chain)."""
        lib, st = self.lib, self._stream()
        T, D, H, E, G = self.T, self.D, self.H, self.E, 3 * self.H
        b = self.buffers(B)
        ws, wsn = _p(b["ws"]), b["ws"].numel()
        drop = self.p > 0
        fn, wsd = (lib.g2v_dec_rollout_bwd_prepared, b["ws_decb"]) if self._prepared else (lib.g2v_dec_rollout_bwd, b["ws"])
        if self._xch_pre:
            wsd = b["ws_decb"]
        pre = "decoder.decoder."
        M = (T - 1) * B
        check(fn(C.byref(self.dec_wstruct()), C.byref(b["sv_loss"] if b["loss_folded"] else b["sv"]), C.byref(b["gr"]), _p(b["keep95"]),
                 _p(b["keep_l0"]) if drop else None, self.p, self.n_pre, int(self.conditioned),
                 T, B, D, H, _p(wsd), wsd.numel(), st))
        x1 = b["x1"] if drop else b["h0"][1:]
        def products():
            dfr = _DeferredReductions(self, b["ws_dec_wgrad"]) if (wgrad_branch and self.defer_reduce) else None
            wgrad, wgrad4 = self._wgrad_fns(b, M, "ws_dec_wgrad" if wgrad_branch else "ws", deferred=dfr)
            items = [(_p(b["dgi0"]), _p(b["a"]), pre + "gru.weight_ih_l0", pre + "gru.bias_ih_l0"),
                     (_p(b["dgh0"]), _p(b["h0"]), pre + "gru.weight_hh_l0", pre + "gru.bias_hh_l0"),
                     (_p(b["dgi1"]), x1.data_ptr(), pre + "gru.weight_ih_l1", pre + "gru.bias_ih_l1"),
                     (_p(b["dgh1"]), _p(b["h1"]), pre + "gru.weight_hh_l1", pre + "gru.bias_hh_l1")]
            rest = [it for m, it in enumerate(items) if not (b["fused_wgrad"] >> m) & 1]      # the others came out of the rollout kernel
            # The largest product FIRST.  A product and the encoder's BPTT kernel do not share a CU's registers (the BPTT's two
            # workgroups per CU hold ~430 of the 512 registers per lane, a product's wave needs 128-256): a product dispatched
            # while the BPTT is resident makes no progress until its workgroups drain (gpurun_tools/corun_probe.py: 65 us alone,
            # 225 us beside it); one that is resident first -- the fork is 18 us ahead of the BPTT -- runs at its stand-alone
            # speed and the BPTT waits for the registers instead (235 -> 293 us).  Largest first measured 0-12 us per step better
            # than the other order (round 3) -- the work only moves between the BPTT and the tail behind it.
            if rest:
                wgrad4(M, rest)
            wgrad(_p(b["du"]), H, _p(b["dec_xin"]), D, pre + "pre_linear.0.weight", pre + "pre_linear.0.bias", H, D)
            # (the rollout's backward ADDS the feedback path's gradient into dy: this product needs the finished dy, it cannot run
            # beside the rollout -- tried in round 3, wrong by construction)
            wgrad(b["dy"][1:].data_ptr(), D, b["h1"][1:].data_ptr(), H, pre + "out_layer.weight", pre + "out_layer.bias", D, H)
            if dfr is not None:
                dfr.flush()             # the branch's ONE slab reduction
            for name in self.frozen:
                g = self.view(name, True)
                check(lib.g2v_fill_f32(_p(g), 0.0, g.numel(), self._stream()))
            if not self._commit_in_apply:
                self._commit_state()        # behind the backward rollout: the fault latch of this step is final
        if wgrad_branch:
            self._fork(2, products, late=wgrad_late)      # (wgrad_late: launched by the caller's _release())
        else:
            products()
        return b

    def backward_encoder(self, in_poses: torch.Tensor, B: int, quant_bwd=None):
        """Encoder layer-0 BPTT from buffers['gz'] (2,B,H) = dLoss / d enc_hidden; writes the encoder's parameter gradients.
        quant_bwd = (g_loss_vq tensor, coef): the quantiser's backward has NOT been run -- the recurrent kernel forms
        gz = dh_init + g_loss_vq coef (enc_hidden - quant) where it reads its incoming gradient (g2v_gru_dir_bwd.hn_*)."""
        lib, st = self.lib, self._stream()
        T, D, H, G = self.T, self.D, self.H, 3 * self.H
        b = self.buffers(B)
        ws, wsn = _p(b["ws"]), b["ws"].numel()
        drop = bool(b.get("enc_dropped", False))      # did this batch's forward_encoder drop its input (x_drop is its tensor)
        wgrad, wgrad4 = self._wgrad_fns(b, T * B)
        enc = "encoder."
        dirs = (_lib.GruDirBwd * 2)()
        for k, (suf, key, hs_ptr) in enumerate((("", "f", b["hs_f"][1:].data_ptr()), ("_reverse", "b", b["hs_b"].data_ptr()))):
            dirs[k].d_hs = None
            dirs[k].d_hn = b["gz"][k].data_ptr() if quant_bwd is None else b["dh_init"][k].data_ptr()
            if quant_bwd is not None:
                dirs[k].hn_z, dirs[k].hn_q = b["enc_hidden"][k].data_ptr(), b["quant"][k].data_ptr()
                dirs[k].hn_gloss, dirs[k].hn_coef = _p(quant_bwd[0]), float(quant_bwd[1])
            dirs[k].hs = hs_ptr
            dirs[k].h0 = None
            dirs[k].gates = _p(b["gates_" + key])
            dirs[k].w_hh = self._w(enc + "gru.weight_hh_l0" + suf)
            dirs[k].dgi = _p(b["dgi_" + key])
            dirs[k].dgh = _p(b["dgh_" + key])
            dirs[k].dh0 = None
            dirs[k].reverse = k
            # H == 64: dxin = dgi W_ih comes out of the recurrent kernel (per direction, into the unused gi buffers)
            dirs[k].w_ih = self._w(enc + "gru.weight_ih_l0" + suf)
            dirs[k].dx = _p(b["gi_" + key]) if H == 64 else None
            dirs[k].in_dim = H
            if b["enc_fused_wgrad"]:        # the recurrent kernel accumulates dW_hh (and, mode 2, dW_ih) itself
                dirs[k].dw_hh, dirs[k].db_hh = self._g(enc + "gru.weight_hh_l0" + suf), self._g(enc + "gru.bias_hh_l0" + suf)
                dirs[k].wslab = _p(b["enc_wslab"][k])
                if b["enc_fused_wgrad"] == 2:
                    dirs[k].dw_ih, dirs[k].db_ih = self._g(enc + "gru.weight_ih_l0" + suf), self._g(enc + "gru.bias_ih_l0" + suf)
                    dirs[k].x = _p(b["xin"])
        if self._prepared:
            check(lib.g2v_gru_seq_bwd_prepared(dirs, 2, None, H, H, T, B, H, _p(b["ws_grub"]), b["ws_grub"].numel(), st))
        elif self._xch_pre:
            check(lib.g2v_gru_seq_bwd(dirs, 2, None, H, H, T, B, H, _p(b["ws_grub"]), b["ws_grub"].numel(), st))
        else:
            check(lib.g2v_gru_seq_bwd(dirs, 2, None, H, H, T, B, H, ws, wsn, st))
        TB = T * B
        # generic dims (D < H): in_layer's gradient and the W_ih gradients are re-associated through in_layer
        # (g2v_linear_bwd_weight_fold2 / _chain2) from P = dgi^T x and c = column sums of dgi per direction
        fold_in = H != 64 and not self.wgrad_bf16x3
        chain_ih = fold_in and b["enc_fused_wgrad"] == 0
        c_in = None

        def p_products():
            nonlocal c_in
            arr = (_lib.WgradItem * 4)()
            # (the column sums of dgi ARE the gradients of bias_ih: written in place when the W_ih gradients are folded too)
            c_in = [self._g(enc + "gru.bias_ih_l0" + suf) if chain_ih else b["c_in"][k].data_ptr() for k, suf in enumerate(("", "_reverse"))]
            for k, key in enumerate(("f", "b")):
                arr[k].dy, arr[k].x = _p(b["dgi_" + key]), _p(b["x_drop"]) if drop else _p(in_poses)
                arr[k].dw, arr[k].db = b["p_in"][k].data_ptr(), c_in[k]
            if drop:        # the dropped input is a (T B, D) tensor of its own; without dropout the (B,T,D) input in (T,B) row order
                check(lib.g2v_linear_bwd_weight_batch(arr, 2, G, D, TB, D, G, 0, ws, wsn, st))
            else:
                check(lib.g2v_linear_bwd_weight_batch_mapped(arr, 2, G, D, B, D, T * D, TB, D, G, 0, ws, wsn, st))

        # Small row counts: the products FIRST, in front of the fork -- beside the W_hh gradients' launch they take 42 us instead of
        # 19, and the fold / chain kernels behind them are what the step's tail waits for (native shape 1.017 -> 1.006 ms).  Large
        # batch: behind the fork, beside the W_hh gradients (in front of it: native dims at B = 4096 6.38 -> 6.73 ms).
        p_first = fold_in and TB < 4096
        if p_first:
            p_products()
        # Round 6: at H = 64 (the W_ih gradients' product + the input layer's two-addend product behind the BPTT) both leave their
        # slab reductions to ONE launch behind the second product (dfr_e); the generic dims keep the immediate calls (their tail
        # is the fold / chain kernels, which read the reduced P, c).
        dfr_e = _DeferredReductions(self, b["ws_enc_wgrad"]) if (self.defer_reduce and H == 64 and not self.wgrad_bf16x3) else None
        with self._branch(4):       # beside the input layer's gradient below (joined there)
            _, wgrad4s = self._wgrad_fns(b, TB, "ws_enc_wgrad" if (self.overlap >> 4) & 1 else "ws", deferred=dfr_e)
            items = [(_p(b["dgi_f"]), _p(b["xin"]), enc + "gru.weight_ih_l0", enc + "gru.bias_ih_l0"),
                     (_p(b["dgh_f"]), b["hs_f"].data_ptr(), enc + "gru.weight_hh_l0", enc + "gru.bias_hh_l0"),
                     (_p(b["dgi_b"]), _p(b["xin"]), enc + "gru.weight_ih_l0_reverse", enc + "gru.bias_ih_l0_reverse"),
                     (_p(b["dgh_b"]), b["hs_b"][1:].data_ptr(), enc + "gru.weight_hh_l0_reverse", enc + "gru.bias_hh_l0_reverse")]
            if chain_ih:
                items = [items[1], items[3]]        # the W_ih gradients: g2v_linear_bwd_weight_chain2 below
            if b["enc_fused_wgrad"] == 1:
                items = [items[0], items[2]]        # the W_hh gradients came out of the recurrent kernel
            if b["enc_fused_wgrad"] < 2:
                if H != 64:
                    # Generic dims: the four-matrix launch is ONE statically partitioned workgroup per CU for ~1 ms.  Started while
                    # the decoder's weight-gradient branch is still finishing (its last product, the same kernel, ~100 us) the
                    # dispatcher doubles some of its workgroups up on the CUs that happen to be free and leaves the others idle
                    # once that product ends: 1.84 ms instead of 0.97, every time the BPTT ends 20 us earlier than the branch
                    # (round 4: seen when the recurrent kernels got faster).  Wait for the branch: it is ~25 us.
                    self._join(2)
                wgrad4s(TB, items)
        sum2 = H == 64 and not self.wgrad_bf16x3 and lib.g2v_linear_bwd_weight_sum2_ok(TB, D, H)
        if sum2 and dfr_e is not None:
            arr = (_lib.WgradItem * 1)()
            arr[0].dy, arr[0].x = _p(b["gi_f"]), _p(b["x_drop"]) if drop else _p(in_poses)
            arr[0].dw, arr[0].db = self._g(enc + "in_layer.weight"), self._g(enc + "in_layer.bias")
            dfr_e.call(arr, 1, H, D, (0, 0, 0) if drop else (B, D, T * D), _p(b["gi_b"]), TB, D, H, 0)
            self._join(4)
            dfr_e.flush()
            return
        if dfr_e is not None:       # (a shape without the two-addend product: reduce what the branch deferred, go on as before)
            self._join(4)
            dfr_e.flush()
        if sum2:
            # the two directions' dx are summed inside the input layer's weight-gradient product (no add pass); with input
            # dropout the layer's input is the dropped tensor the forward left in x_drop
            if drop:
                check(lib.g2v_linear_bwd_weight_sum2(_p(b["gi_f"]), _p(b["gi_b"]), H, _p(b["x_drop"]), D, 0, 0, 0,
                                                     self._g(enc + "in_layer.weight"), self._g(enc + "in_layer.bias"),
                                                     TB, D, H, 0, ws, wsn, st))
            else:
                check(lib.g2v_linear_bwd_weight_sum2(_p(b["gi_f"]), _p(b["gi_b"]), H, _p(in_poses), D, B, D, T * D,
                                                     self._g(enc + "in_layer.weight"), self._g(enc + "in_layer.bias"),
                                                     TB, D, H, 0, ws, wsn, st))
            self._join(4)
            return
        if H == 64:
            check(lib.g2v_add_halves(_p(b["gi_f"]), H, _p(b["gi_b"]), H, _p(b["dxin"]), H, TB, H, st))     # sum of the two directions
        elif fold_in:
            # in_layer's gradient WITHOUT the (T B x H) gradient of its output (nothing else reads it: the layer's input is the
            # network's input): dW_in = W_f^T (dgi_f^T x) + W_b^T (dgi_b^T x) -- two weight-gradient products with K = D in one
            # launch and one small fold, instead of two (T B x 3H)(3H x H) products and a weight-gradient product
            if not p_first:
                p_products()
            if chain_ih:      # ... and dW_ih = dgi^T in_layer(x) = (dgi^T x) W_in^T + (dgi^T 1) b_in^T, in the same launch
                check(lib.g2v_linear_bwd_weight_fold_chain2(
                    self._w(enc + "gru.weight_ih_l0"), self._w(enc + "gru.weight_ih_l0_reverse"), b["p_in"][0].data_ptr(),
                    b["p_in"][1].data_ptr(), c_in[0], c_in[1], self._w(enc + "in_layer.weight"), self._w(enc + "in_layer.bias"),
                    self._g(enc + "in_layer.weight"), self._g(enc + "in_layer.bias"), self._g(enc + "gru.weight_ih_l0"),
                    self._g(enc + "gru.weight_ih_l0_reverse"), G, H, D, st))
            else:
                check(lib.g2v_linear_bwd_weight_fold2(self._w(enc + "gru.weight_ih_l0"), self._w(enc + "gru.weight_ih_l0_reverse"),
                                                      b["p_in"][0].data_ptr(), b["p_in"][1].data_ptr(), c_in[0], c_in[1],
                                                      self._g(enc + "in_layer.weight"), self._g(enc + "in_layer.bias"), G, H, D, 0, st))
            self._join(4)
            return
        else:
            check(lib.g2v_linear_bwd_data(_p(b["dgi_f"]), G, self._w(enc + "gru.weight_ih_l0"), _p(b["dxin"]), H, TB, H, G, 0, st))
            check(lib.g2v_linear_bwd_data(_p(b["dgi_b"]), G, self._w(enc + "gru.weight_ih_l0_reverse"), _p(b["dxin"]), H, TB, H, G, 1, st))
        if drop:
            wgrad(_p(b["dxin"]), H, _p(b["x_drop"]), D, enc + "in_layer.weight", enc + "in_layer.bias", H, D, rows=TB)
        else:
            wgrad(_p(b["dxin"]), H, _p(in_poses), D, enc + "in_layer.weight", enc + "in_layer.bias", H, D, rows=TB,
                  row_map=(B, D, T * D))
        self._join(4)
        # encoder GRU layer 1 receives exactly-zero gradients (dead compute in the reference); the flat grad
        # buffer is zero there from construction and nothing ever writes it.

    def optimizer_step(self, lr: float, betas=(0.5, 0.999), eps: float = 1e-8, max_norm: float = 5.0,
                       grad_scale: float = 1.0, readback: bool = False):
        """clip_grad_norm_ + Adam over the flat buffers; readback: the same launch also gathers [custom_loss, loss_vq, perplexity,
        fault latch] into self.readback (train_iter's one device-to-host copy; written also when the latch holds the update back)"""
        if readback:
            check(self.lib.g2v_clip_adam_step_readback(_p(self.flat), _p(self.gflat), _p(self.m), _p(self.v), self.n_flat,
                                                       _p(self.adam_partial), _p(self.step_counter), _p(self.gnorm), max_norm,
                                                       grad_scale, lr, betas[0], betas[1], eps, _p(self.loss_terms),
                                                       _p(self.vq_scalars), self.vq_scalars[1:].data_ptr(), _p(self.readback),
                                                       self._stream()))
            return
        check(self.lib.g2v_clip_adam_step(_p(self.flat), _p(self.gflat), _p(self.m), _p(self.v), self.n_flat,
                                          _p(self.adam_partial), _p(self.step_counter), _p(self.gnorm), max_norm,
                                          grad_scale, lr, betas[0], betas[1], eps, self._stream()))

    # ------------------------------------------------------------------ one fused train iteration
    def train_step(self, x: torch.Tensor, target: torch.Tensor, *, lr: float, w_l1: float, w_cont: float,
                   w_var: float, epoch: int = 1, draw_masks: bool = True, reduce_fn=None, world: int = 1,
                   betas=(0.5, 0.999), eps: float = 1e-8, max_norm: float = 5.0):
        """train_iter_Autoencoder_VQ_seq2seq (train_eval/train_seq2seq.py:664-758) without host syncs.
        reduce_fn(comm) performs the data-parallel SUM all-reduce (RCCL) of [grads | EMA stats] when world > 1."""
        dp = reduce_fn is not None and world > 1
        self.train_step_local(x, target, w_l1=w_l1, w_cont=w_cont, w_var=w_var, epoch=epoch, draw_masks=draw_masks, dp=dp)
        if dp:
            reduce_fn(self.comm)
        self.train_step_apply(x.shape[0], lr=lr, world=world, dp=dp, betas=betas, eps=eps, max_norm=max_norm)

    # The two halves around the data-parallel exchange.  Each is a fixed kernel sequence with no host sync, so each can
    # be captured in its own hipGraph; the RCCL all-reduce of `comm` runs between the two replays.
    def train_step_local(self, x: torch.Tensor, target: torch.Tensor, *, w_l1: float, w_cont: float, w_var: float,
                         epoch: int = 1, draw_masks: bool = True, dp: bool = False):
        """masks -> forward -> loss -> backward; leaves comm = [grads | cnt | dw] holding this rank's contribution."""
        B = x.shape[0]
        self._branches_on = self._branches_ok(B)
        self._prepared = self._branches_on and (self.overlap & 9) == 9
        try:
            self._train_step_local(x, target, w_l1, w_cont, w_var, epoch, draw_masks, dp, B)
            self._steps += 1
        finally:
            self._prepared = False
            self._side_work = None
            if self._xch_pre:
                # a step that ended early (an exception between the pre-clear and the cluster launches) must not leave its
                # "already clear" notes behind: the next launch over those workspaces clears for itself (advisor finding)
                self.lib.g2v_cluster_exchange_preclear_drop(None, 0)
            self._xch_pre = False
            self._fused_in_drop = False
            self._defer_commit = False
            if not dp:
                self._commit_pending = None

    def _train_step_local(self, x, target, w_l1, w_cont, w_var, epoch, draw_masks, dp, B):
        self._fused_in_drop = bool(draw_masks and self.p > 0)      # the encoder's input mask is drawn inside its dropout kernel
        def side():                            # branch 0: beside the encoder GRU (forked in forward_encoder), joined before the quantiser
            if self.xch_preclear and self.H != 64:
                # generic dims at small batch: the exchange records of the three cluster launches behind the quantiser are cleared
                # HERE, off the chain, in workspaces only those launches use (g2v_cluster_exchange_preclear; a no-op for shapes
                # that do not run as clusters)
                bb = self.buffers(B)
                for kind, key in ((2, "ws_decf"), (3, "ws_decb"), (1, "ws_grub")):
                    check(self.lib.g2v_cluster_exchange_preclear(kind, self.T, B, self.D, self.H, 2, _p(bb[key]), bb[key].numel(),
                                                                 self._stream()))
                self._xch_pre = True
            if self.quantizer == "ema":
                self.vq_derive()               # needed first: the quantiser follows the encoder directly
            elif self.quantizer == "gssoft":   # |W_k|^2 of the codebook as it is now: off the chain in front of the fused quantiser kernel
                check(self.lib.g2v_vq_code_sqnorm(self._w("vq_layer._embedding.weight"), _p(self.code_sqnorm), self.K, self.E,
                                                  self._stream()))
                self._gs_wsq_ready = True
            if draw_masks and not self.side_early:
                self.draw_masks(B, True, "rest")       # only the rollout consumes keep95 / keep_l0
            for t, n in self.tracked_counters:         # (train_iter: BatchNorm's num_batches_tracked)
                check(self.lib.g2v_counter_add(_p(t), int(n), self._stream()))
            if self._prepared:
                if self.merged_prepare and self.H == 64 and self.D == 135:
                    self.prepare_recurrent(B, "all")   # one launch (g2v_train_step_prepare)
                else:
                    self.prepare_recurrent(B, "dec")
                    self.prepare_recurrent(B, "gru_bwd")
            if draw_masks and self.side_early:
                self.draw_masks(B, True, "rest")       # last: the one HBM-heavy kernel of the branch (see forward_encoder)
        self._side_work = side
        g_vq = self.g_loss_vq if epoch > 0 else torch.zeros_like(self.g_loss_vq)
        self._g_vq_host, self._g_vq_dev = (1.0 / 400.0 if epoch > 0 else 0.0), g_vq       # (:707, 738: loss + loss_vq / 400 from epoch 1)
        self._defer_commit = True
        self._commit_pending = (B, self.quantizer == "ema")
        self._commit_in_apply = dp          # data parallel: every commit behind the all-reduce (global statistics, global fault flag)
        self.forward(x, target, True, ema_update=False, derived_ready=True, loss_w=(w_l1, w_cont, w_var),
                     join_stats=False)
        self.loss(B, target, w_l1, w_cont, w_var, True)
        self.backward(x, B, g_vq)
        if dp:
            # this rank's latch into the communication buffer's flag slot: a rank whose rollout faulted has fed garbage into the
            # SUM, so all ranks must skip the step (train_step_apply turns a non-zero reduced flag into every rank's latch)
            check(self.lib.g2v_dec_rollout_fault_flag(_p(self.fault_flag), 0, self._stream()))
        else:
            self._commit_state()               # (a no-op when backward_decoder's branch has run it)
        self._join(1)                          # the statistics / codebook-update branch (forward(join_stats=False); a no-op behind the chaser's join)

    def train_step_apply(self, B: int, *, lr: float, world: int = 1, dp: bool = False, betas=(0.5, 0.999),
                         eps: float = 1e-8, max_norm: float = 5.0):
        """(after the all-reduce) EMA codebook update from the GLOBAL statistics, then clip + Adam on the averaged grads."""
        if dp:
            check(self.lib.g2v_dec_rollout_fault_flag(_p(self.fault_flag), 1, self._stream()))
            pend, self._commit_pending = self._commit_pending, None
            if pend is not None:            # (train_step_local left them: global statistics, global fault flag)
                self._commit_pending = pend
                self._commit_state(n_global=world * ((2 * B * self.H) // self.E))
            elif self.quantizer == "ema":   # (a caller that ran the local half some other way)
                self.vq_finish(B, True, n_global=world * ((2 * B * self.H) // self.E))
        self.optimizer_step(lr, betas=betas, eps=eps, max_norm=max_norm, grad_scale=1.0 / world if dp else 1.0, readback=True)


def _bind_context(fn):
    import functools

    @functools.wraps(fn)
    def bound(self, *a, **k):
        with self.ctx:
            return fn(self, *a, **k)
    return bound


# every public method that reaches the library runs with the engine's own context bound to the calling thread
for _name in ("buffers", "prepare_recurrent", "draw_masks", "forward", "forward_encoder", "forward_decoder", "vq_derive",
              "refresh_codebook_state", "vq_finish", "loss", "backward", "backward_decoder", "backward_encoder", "optimizer_step",
              "train_step", "train_step_local", "train_step_apply", "check_faults", "rearm", "vq_bx_mismatches"):
    setattr(VQVAEEngine, _name, _bind_context(getattr(VQVAEEngine, _name)))
del _name

