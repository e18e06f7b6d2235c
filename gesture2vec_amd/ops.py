"""Tensor-level wrappers over the C-ABI (include/g2v.h).  PyTorch is plumbing only: it owns device
memory and the stream; every number is produced by the HIP kernels in libg2v_hip.so.

All tensors must live on the GPU, be fp32 (uint8 masks, int64 indices, int32 lengths) and contiguous
unless a stride argument says otherwise.  Nothing here falls back to torch arithmetic."""
from __future__ import annotations

import ctypes as C
from typing import Optional

import torch

from . import _lib
from ._lib import CodeDecGrads, CodeDecSaved, CodeDecWeights, DecGrads, DecSaved, DecWeights, check


def _lib_():
    return _lib.load()


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _p(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


def _chk(t: torch.Tensor, dtype=torch.float32, name="tensor"):
    if not t.is_cuda:
        raise _lib.G2VLibraryError(f"{name} must be a GPU tensor: the g2v kernels have no CPU path")
    if t.dtype != dtype:
        raise TypeError(f"{name}: expected {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise ValueError(f"{name} must be contiguous")
    return t


# ---- side branches (round 6) ------------------------------------------------------------------------------------------------
# Part d's iteration is an autograd chain on ONE stream; a good part of its backward is weight-gradient work nothing downstream
# waits for (a layer's, while its input gradient moves on).
# Inside `with side_branches():` (train_iter_text2embedding, GraphedText2EmbeddingStep) a block under `with side_branch(k, keep)`
# is launched on side stream k, forked from the current stream; `join_side()` (FlatParams.gather_grads, and the scope's exit)
# makes the current stream wait for all of them.  Outside such a scope a side_branch block runs inline: somebody who calls
# backward() and reads .grad without an optimiser step never sees an unjoined stream.
#   * `keep`: every tensor the block READS that was allocated on the forking stream -- the caching allocator hands a block
#     back to its own stream's pool the moment the last reference dies (the end of the autograd node), while the side launch may
#     still be reading; they stay referenced until the join.  What the block ALLOCATES belongs to the side stream's pool and is
#     only ever recycled there, stream-ordered behind the fork.
#   * the reusable workspaces are per branch (a tag suffix): two weight-gradient launches on two streams never share slabs.
#   * a gradient produced on a side stream must not be ACCUMULATED into an existing .grad by autograd (an add on the main
#     stream); every parameter of Part d's model receives one contribution per iteration and zero_grad() resets to None.
import contextlib
import os
import threading

_side_state = {"active": 0, "streams": {}, "pending": []}
_side_local = threading.local()
SIDE_BRANCHES = os.environ.get("G2V_SIDE_BRANCHES", "1") != "0"


def join_side():
    """the current stream waits for every side branch forked since the last join (their kept tensors are released)"""
    pend, _side_state["pending"] = _side_state["pending"], []
    if pend:
        cur = torch.cuda.current_stream()
        for s, _keep in pend:
            cur.wait_stream(s)


def reset_side_streams():
    """Forget the cached side streams (new ones are drawn at the next fork).  GraphedText2EmbeddingStep does this in front of every
    capture: on ROCm 7.2 the replay of the FOURTH multi-stream graph captured over the same side streams in one process, with the
    third still alive, died inside hipGraphLaunch (gpurun_tools/r06_side_repro.py: seq4 against seq4fresh / seq4gc)."""
    join_side()
    _side_state["streams"].clear()


@contextlib.contextmanager
def side_branches(enabled: bool = True):
    on = bool(enabled) and SIDE_BRANCHES
    _side_state["active"] += 1 if on else 0
    try:
        yield
    finally:
        if on:
            _side_state["active"] -= 1
            join_side()


@contextlib.contextmanager
def side_branch(k: int = 0, keep=(), rows: int = 1 << 62, min_rows: int = 0):
    """rows / min_rows: the caller's size rule -- a fork costs two cross-queue edges of the replayed graph (5-10 us each) and the
    co-running kernels slow each other down; measured on Part d (gpurun_tools/r06_side_mask.py, profiles/r06_i_side_mask.log):
    the encoder's weight gradients beside its input-gradient chain pay from B = 2048 (-2 ... -4 %), cost 1-3 % at B = 1024 and
    10 % at B = 128 with attention; the decoder's parameter gradients beside the encoder's BPTT never paid and stay inline."""
    if not _side_state["active"] or rows < min_rows:
        yield False
        return
    cur = torch.cuda.current_stream()
    key = (cur.device.index, int(k))
    s = _side_state["streams"].get(key)
    if s is None:
        s = _side_state["streams"][key] = torch.cuda.Stream(device=cur.device)
    s.wait_stream(cur)
    prev = getattr(_side_local, "suffix", "")
    _side_local.suffix = f"@side{int(k)}"
    try:
        with torch.cuda.stream(s):
            yield True
    finally:
        _side_local.suffix = prev
        _side_state["pending"].append((s, keep))


_ws_cache = {}
_ws_retired = []      # superseded workspaces stay allocated: see workspace()


def workspace(nbytes: int, device, tag: str = "ws") -> torch.Tensor:
    """A reusable byte buffer per (device, tag); grows monotonically (allocation never happens inside the C-ABI).

    A buffer that is outgrown is RETIRED, never freed: its address may be baked into a captured hipGraph
    (train_eval.train_seq2seq.GraphedText2EmbeddingStep replays ops.* calls), and handing the block back to the caching
    allocator would let a later replay scribble its weight-gradient slabs over whatever tensor reuses the memory.  The cost
    is bounded: sizes only grow, so at most a geometric series of small blocks per tag stays behind."""
    key = (str(device), tag + getattr(_side_local, "suffix", ""))
    buf = _ws_cache.get(key)
    if buf is None or buf.numel() < nbytes:
        if buf is not None:
            _ws_retired.append(buf)
        grow = max(int(nbytes), 256, 0 if buf is None else 2 * buf.numel())
        buf = torch.empty(grow, dtype=torch.uint8, device=device)
        _ws_cache[key] = buf
    return buf


# ------------------------------------------------------------------------------------------ linear
def linear_fwd(x, w, bias=None, act=0, *, M=None, ldx=None, row_map=None, keep=None, scale=1.0, out=None, ldy=None):
    """y = act(x w^T + b).  row_map = (rows_inner, stride_outer, stride_inner) in elements."""
    N, K = w.shape
    if M is None:
        M = x.numel() // K
    if ldx is None:
        ldx = K
    if out is None:
        out = torch.empty((M, N), dtype=torch.float32, device=x.device)
    if ldy is None:
        ldy = N
    ri, so, si = row_map if row_map is not None else (0, 0, 0)
    check(_lib_().g2v_linear_fwd(_p(x), ldx, ri, so, si, _p(keep), float(scale), _p(_chk(w, name="w")), _p(bias),
                                 _p(out), ldy, M, K, N, act, _stream()), "linear_fwd")
    return out


def linear_compose2(w0, b0, w1, b1, w_in, b_in):
    """(w_p w_in, w_p b_in + b_p) for p = 0, 1 (g2v_linear_compose2): two linear layers in a row as one."""
    G, H = w0.shape
    D = w_in.shape[1]
    dev = w0.device
    wc = [torch.empty((G, D), dtype=torch.float32, device=dev) for _ in range(2)]
    bc = [torch.empty((G,), dtype=torch.float32, device=dev) for _ in range(2)]
    check(_lib_().g2v_linear_compose2(_p(_chk(w0, name="w0")), _p(b0), _p(_chk(w1, name="w1")), _p(b1), _p(_chk(w_in, name="w_in")),
                                      _p(b_in), _p(wc[0]), _p(bc[0]), _p(wc[1]), _p(bc[1]), G, H, D, _stream()), "linear_compose2")
    return wc[0], bc[0], wc[1], bc[1]


def linear_fwd_pair(x, w_a, bias_a, w_b, bias_b, act=0, *, M=None):
    """(act(x w_a^T + b_a), act(x w_b^T + b_b)) in one launch where that pays (g2v_linear_fwd_pair)."""
    N, K = w_a.shape
    if M is None:
        M = x.numel() // K
    ya = torch.empty((M, N), dtype=torch.float32, device=x.device)
    yb = torch.empty((M, N), dtype=torch.float32, device=x.device)
    check(_lib_().g2v_linear_fwd_pair(_p(x), K, _p(_chk(w_a, name="w_a")), _p(bias_a), _p(ya), _p(_chk(w_b, name="w_b")), _p(bias_b),
                                      _p(yb), N, M, K, N, act, _stream()), "linear_fwd_pair")
    return ya, yb


def linear_bwd_data(dy, w, *, M=None, lddy=None, out=None, lddx=None, accumulate=False):
    N, K = w.shape
    if M is None:
        M = dy.numel() // N
    if lddy is None:
        lddy = N
    if out is None:
        out = torch.empty((M, K), dtype=torch.float32, device=dy.device)
    if lddx is None:
        lddx = K
    check(_lib_().g2v_linear_bwd_data(_p(dy), lddy, _p(w), _p(out), lddx, M, K, N, int(accumulate), _stream()),
          "linear_bwd_data")
    return out


def linear_bwd_weight(dy, x, N, K, *, M=None, lddy=None, ldx=None, row_map=None, keep=None, scale=1.0,
                      dw=None, db=None, want_bias=True, accumulate=False, bf16x3=False):
    """dw = dy^T xin (+ db).  bf16x3=True allows the 3-term bf16 split products (G2V_WGRAD_BF16X3, ~1.5e-5 relative)."""
    if M is None:
        M = dy.numel() // N
    if lddy is None:
        lddy = N
    if ldx is None:
        ldx = K
    dev = dy.device
    if dw is None:
        dw = torch.empty((N, K), dtype=torch.float32, device=dev)
    if db is None and want_bias:
        db = torch.empty((N,), dtype=torch.float32, device=dev)
    lib = _lib_()
    nbytes = lib.g2v_linear_bwd_weight_workspace(M, K, N)
    ws = workspace(nbytes, dev, "bwdw")
    ri, so, si = row_map if row_map is not None else (0, 0, 0)
    check(lib.g2v_linear_bwd_weight(_p(dy), lddy, _p(x), ldx, ri, so, si, _p(keep), float(scale), _p(dw), _p(db),
                                    M, K, N, int(bool(accumulate)) | (2 if bf16x3 else 0), _p(ws), ws.numel(), _stream()),
          "linear_bwd_weight")
    return dw, db


def linear_bwd_weight_sum2(dy_a, dy_b, x, N, K, *, M, lddy=None, ldx=None, row_map=None, dw=None, db=None, want_bias=True,
                           accumulate=False):
    """dw = (dy_a + dy_b)^T x (+ db) without the add pass; raises for shapes g2v_linear_bwd_weight_sum2_ok() rejects."""
    dev = dy_a.device
    if dw is None:
        dw = torch.empty((N, K), dtype=torch.float32, device=dev)
    if db is None and want_bias:
        db = torch.empty((N,), dtype=torch.float32, device=dev)
    lib = _lib_()
    ws = workspace(lib.g2v_linear_bwd_weight_workspace(M, K, N), dev, "bwdw")
    ri, so, si = row_map if row_map is not None else (0, 0, 0)
    check(lib.g2v_linear_bwd_weight_sum2(_p(dy_a), _p(dy_b), lddy if lddy is not None else N, _p(x),
                                         ldx if ldx is not None else K, ri, so, si, _p(dw), _p(db), M, K, N,
                                         int(bool(accumulate)), _p(ws), ws.numel(), _stream()), "linear_bwd_weight_sum2")
    return dw, db


def linear_bwd_weight_batch(items, N, K, *, M, lddy=None, ldx=None, accumulate=False, bf16x3=False, row_map=None):
    """items: up to 4 tuples (dy, x, dw, db-or-None) of ONE shape -> one launch + one slab reduction (large M).
    row_map = (rows_inner, stride_outer, stride_inner): x row-mapped as in linear_fwd (g2v_linear_bwd_weight_batch_mapped)."""
    lib = _lib_()
    arr = (_lib.WgradItem * len(items))()
    for k, (dy, x, dw, db) in enumerate(items):
        arr[k].dy, arr[k].x, arr[k].dw, arr[k].db = _p(dy), _p(x), _p(dw), _p(db)
    dev = items[0][0].device
    ws = workspace(len(items) * lib.g2v_linear_bwd_weight_workspace(M, K, N), dev, "bwdw")
    if row_map is not None:
        check(lib.g2v_linear_bwd_weight_batch_mapped(arr, len(items), lddy if lddy is not None else N, ldx if ldx is not None else K,
                                                     row_map[0], row_map[1], row_map[2], M, K, N,
                                                     int(bool(accumulate)) | (2 if bf16x3 else 0), _p(ws), ws.numel(), _stream()),
              "linear_bwd_weight_batch_mapped")
        return
    check(lib.g2v_linear_bwd_weight_batch(arr, len(items), lddy if lddy is not None else N, ldx if ldx is not None else K, M, K, N,
                                          int(bool(accumulate)) | (2 if bf16x3 else 0), _p(ws), ws.numel(), _stream()),
          "linear_bwd_weight_batch")


def linear_bwd_weight_deferred(calls, *, accumulate=False):
    """Several weight-gradient calls whose slab reductions run as ONE launch behind the last product (include/g2v.h:
    g2v_linear_bwd_weight_deferred / _reduce).  calls: list of dicts {items: [(dy, x, dw, db-or-None), ...] of one shape, N, K, M,
    lddy, ldx, row_map, dy_b}.  Results are bitwise those of the immediate calls."""
    lib = _lib_()
    dev = calls[0]["items"][0][0].device
    need = [len(c["items"]) * int(lib.g2v_linear_bwd_weight_workspace(c["M"], c["K"], c["N"])) for c in calls]
    offs, tot = [], 0
    for nb in need:
        offs.append(tot)
        tot = (tot + nb + 255) & ~255
    ws = workspace(tot + 256, dev, "bwdw_deferred")
    pend = []
    for c, off, nb in zip(calls, offs, need):
        arr = (_lib.WgradItem * len(c["items"]))()
        for k, (dy, x, dw, db) in enumerate(c["items"]):
            arr[k].dy, arr[k].x, arr[k].dw, arr[k].db = _p(dy), _p(x), _p(dw), _p(db)
        rm = c.get("row_map") or (0, 0, 0)
        pd = _lib.WgradPending()
        check(lib.g2v_linear_bwd_weight_deferred(arr, len(c["items"]), c.get("lddy", c["N"]), c.get("ldx", c["K"]), rm[0], rm[1], rm[2],
                                                 _p(c.get("dy_b")), c["M"], c["K"], c["N"], int(bool(accumulate)), ws.data_ptr() + off, nb,
                                                 C.byref(pd), _stream()), "linear_bwd_weight_deferred")
        pend.append(pd)
    live = [p_ for p_ in pend if p_.nprob > 0]
    for k in range(0, len(live), 8):
        chunk = live[k:k + 8]
        check(lib.g2v_linear_bwd_weight_reduce((_lib.WgradPending * len(chunk))(*chunk), len(chunk), _stream()), "linear_bwd_weight_reduce")
    return [p_.nprob for p_ in pend]


def linear_bwd_weight_fold2(w0, w1, p0, p1, c0, c1, dw=None, db=None, accumulate=False):
    """dw = w0^T p0 + w1^T p1, db = w0^T c0 + w1^T c1 (g2v_linear_bwd_weight_fold2: the weight gradient of a layer in front of
    two parallel layers from their weight-gradient-shaped products).  w: (G,H), p: (G,D), c: (G) -> dw (H,D), db (H)."""
    G, H = w0.shape
    D = p0.shape[1]
    if dw is None:
        dw = torch.empty((H, D), dtype=torch.float32, device=w0.device)
    if db is None:
        db = torch.empty((H,), dtype=torch.float32, device=w0.device)
    check(_lib_().g2v_linear_bwd_weight_fold2(_p(_chk(w0, name="w0")), _p(_chk(w1, name="w1")), _p(p0), _p(p1), _p(c0), _p(c1),
                                              _p(dw), _p(db), G, H, D, int(bool(accumulate)), _stream()), "linear_bwd_weight_fold2")
    return dw, db


def linear_bwd_weight_chain2(p0, p1, c0, c1, w_in, b_in, accumulate=False, dw0=None, dw1=None):
    """dw_p = p_p w_in^T + c_p b_in^T (g2v_linear_bwd_weight_chain2).  p: (G,D), c: (G), w_in: (H,D), b_in: (H) -> two (G,H)."""
    G, D = p0.shape
    H = w_in.shape[0]
    if dw0 is None:
        dw0 = torch.empty((G, H), dtype=torch.float32, device=p0.device)
    if dw1 is None:
        dw1 = torch.empty((G, H), dtype=torch.float32, device=p0.device)
    check(_lib_().g2v_linear_bwd_weight_chain2(_p(p0), _p(p1), _p(c0), _p(c1), _p(_chk(w_in, name="w_in")), _p(b_in), _p(dw0), _p(dw1),
                                               G, H, D, int(bool(accumulate)), _stream()), "linear_bwd_weight_chain2")
    return dw0, dw1


def linear_bwd_weight_fold_chain2(w0, w1, p0, p1, c0, c1, w_in, b_in):
    """linear_bwd_weight_fold2 + linear_bwd_weight_chain2 in one launch -> (dw_in, db_in, dw0, dw1)."""
    G, H = w0.shape
    D = p0.shape[1]
    dev = w0.device
    dw_in, db_in = torch.empty((H, D), dtype=torch.float32, device=dev), torch.empty((H,), dtype=torch.float32, device=dev)
    dw0, dw1 = torch.empty((G, H), dtype=torch.float32, device=dev), torch.empty((G, H), dtype=torch.float32, device=dev)
    check(_lib_().g2v_linear_bwd_weight_fold_chain2(_p(_chk(w0, name="w0")), _p(_chk(w1, name="w1")), _p(p0), _p(p1), _p(c0), _p(c1),
                                                    _p(_chk(w_in, name="w_in")), _p(b_in), _p(dw_in), _p(db_in), _p(dw0), _p(dw1),
                                                    G, H, D, _stream()), "linear_bwd_weight_fold_chain2")
    return dw_in, db_in, dw0, dw1


# ------------------------------------------------------------------------------------------ quantiser
def vq_code_sqnorm(codebook, out=None):
    K, E = codebook.shape
    if out is None:
        out = torch.empty((K,), dtype=torch.float32, device=codebook.device)
    check(_lib_().g2v_vq_code_sqnorm(_p(_chk(codebook)), _p(out), K, E, _stream()), "vq_code_sqnorm")
    return out


def vq_assign(flat, z, codebook, code_sqnorm, want_quantized=True, want_dist=False):
    """-> idx (N) int64, quantized (N,E) | None, dist_min (N) | None, sse_partial | None"""
    N, E = flat.shape
    K = codebook.shape[0]
    dev = flat.device
    lib = _lib_()
    idx = torch.empty((N,), dtype=torch.int64, device=dev)
    quant = torch.empty((N, E), dtype=torch.float32, device=dev) if want_quantized else None
    dmin = torch.empty((N,), dtype=torch.float32, device=dev) if want_dist else None
    sse = torch.empty((lib.g2v_vq_assign_blocks(N),), dtype=torch.float32, device=dev) if want_quantized else None
    if vq_assign_packed_pays(N, E, K):
        # the reference's own quantiser shapes (E = 400): fragment-major codebook image + the eight-wave kernel, same outputs bit
        # for bit (g2v_vq_assign_packed_fwd); the image is rebuilt per call (6 us: nothing is trusted across calls)
        frag = workspace(K * E * 4, flat.device, "vqpack").view(torch.float32)[:K * E]
        check(lib.g2v_vq_pack_codebook(_p(_chk(codebook)), _p(frag), K, E, _stream()), "vq_pack_codebook")
        check(lib.g2v_vq_assign_packed_fwd(_p(_chk(flat)), _p(z), _p(codebook), _p(frag), _p(code_sqnorm), _p(idx), _p(quant),
                                           _p(dmin), _p(sse), N, E, K, _stream()), "vq_assign_packed_fwd")
        return idx, quant, dmin, sse
    check(lib.g2v_vq_assign_fwd(_p(_chk(flat)), _p(z), _p(_chk(codebook)), _p(code_sqnorm), _p(idx), _p(quant),
                                _p(dmin), _p(sse), N, E, K, _stream()), "vq_assign_fwd")
    return idx, quant, dmin, sse


VQ_PACKED_MIN_ROWS = 2048      # below: g2v_vq_assign_fwd splits the CODES over workgroups (few row tiles), which is faster there


def vq_assign_packed_pays(N, E, K) -> bool:
    """shapes g2v_vq_assign_fwd has no tuned kernel for (anything but E = 128 with K % 128 = 0) at enough rows to fill the chip"""
    return bool(N >= VQ_PACKED_MIN_ROWS and not (E == 128 and K % 128 == 0) and _lib_().g2v_vq_assign_packed_ok(int(N), int(E), int(K)))


def vq_assign_bulk(flat, codebook, code_sqnorm, want_undecided=False):
    """idx (N,) = argmin_k |flat - W_k|^2 for many rows: bf16 split screening + exact fp32 re-check (g2v_vq_assign_bulk)."""
    N, E = flat.shape
    K = codebook.shape[0]
    lib = _lib_()
    idx = torch.empty((N,), dtype=torch.int64, device=flat.device)
    nb = int(lib.g2v_vq_assign_bulk_workspace(N, E, K))
    ws = workspace(nb, flat.device, "vqbulk")
    und = torch.zeros((1,), dtype=torch.int32, device=flat.device) if want_undecided else None
    check(lib.g2v_vq_assign_bulk(_p(_chk(flat)), _p(_chk(codebook)), _p(_chk(code_sqnorm)), _p(idx), N, E, K, _p(ws), nb, _p(und),
                                 _stream()), "vq_assign_bulk")
    return (idx, und) if want_undecided else idx


def vq_pack_codebook(codebook, out=None):
    """fragment-major image of the codebook for vq_fused_assign(..., codebook_frag=)"""
    K, E = codebook.shape
    if out is None:
        out = torch.empty((K * E,), dtype=torch.float32, device=codebook.device)
    check(_lib_().g2v_vq_pack_codebook(_p(_chk(codebook)), _p(out), K, E, _stream()), "vq_pack_codebook")
    return out


def vq_fused_assign(z, w_pre, b_pre, codebook, code_sqnorm, codebook_frag=None):
    """pre_linear + assign in one launch (E == 128, K % 128 == 0) -> flat (N,E), idx (N) int64, quantized (N,E), sse_partial.
    codebook_frag (vq_pack_codebook): read the distance operands from the fragment-major image (same results, faster loads)."""
    N, E = z.shape
    K = codebook.shape[0]
    dev = z.device
    lib = _lib_()
    flat = torch.empty((N, E), dtype=torch.float32, device=dev)
    idx = torch.empty((N,), dtype=torch.int64, device=dev)
    quant = torch.empty((N, E), dtype=torch.float32, device=dev)
    sse = torch.empty((lib.g2v_vq_assign_blocks(N),), dtype=torch.float32, device=dev)
    if codebook_frag is not None:
        check(lib.g2v_vq_fused_assign_packed_fwd(_p(_chk(z)), _p(_chk(w_pre)), _p(_chk(b_pre)), _p(_chk(codebook)),
                                                 _p(_chk(codebook_frag)), _p(_chk(code_sqnorm)), _p(flat), _p(idx), _p(quant),
                                                 _p(sse), N, E, K, _stream()), "vq_fused_assign_packed_fwd")
        return flat, idx, quant, sse
    check(lib.g2v_vq_fused_assign_fwd(_p(_chk(z)), _p(_chk(w_pre)), _p(_chk(b_pre)), _p(_chk(codebook)), _p(_chk(code_sqnorm)),
                                      _p(flat), _p(idx), _p(quant), _p(sse), N, E, K, _stream()), "vq_fused_assign_fwd")
    return flat, idx, quant, sse


VQ_BX_EXACT = 1          # include/g2v.h G2V_VQ_BX_*


def vq_bx_pack(codebook, code_sqnorm, w_pre, b_pre, out=None):
    """screening operands of vq_fused_assign_bx (g2v_vq_bx_pack): bf16 fragments of U = W w_pre, s'_k, per-code radius coefficients"""
    K, E = codebook.shape
    nb = int(_lib_().g2v_vq_bx_image_bytes(K, E))
    if out is None:
        out = torch.empty((nb,), dtype=torch.uint8, device=codebook.device)
    check(_lib_().g2v_vq_bx_pack(_p(_chk(codebook)), _p(_chk(code_sqnorm)), _p(_chk(w_pre)), _p(_chk(b_pre)), _p(out), K, E,
                                 _stream()), "vq_bx_pack")
    return out


def vq_fused_assign_bx(z, w_pre_frag, b_pre, codebook, image, code_sqnorm, flags=0, want_diag=False):
    """pre_linear + bf16-screened / fp32-re-checked assign in one launch (g2v_vq_fused_assign_bx_fwd) -> flat, idx, quantized,
    sse_partial[, diag int32[4]: tiles on the exact sweep, pairs re-evaluated].  w_pre_frag = vq_pack_codebook(w_pre)."""
    N, E = z.shape
    K = codebook.shape[0]
    dev = z.device
    lib = _lib_()
    flat = torch.empty((N, E), dtype=torch.float32, device=dev)
    idx = torch.empty((N,), dtype=torch.int64, device=dev)
    quant = torch.empty((N, E), dtype=torch.float32, device=dev)
    sse = torch.empty((lib.g2v_vq_assign_blocks(N),), dtype=torch.float32, device=dev)
    diag = torch.zeros((4,), dtype=torch.int32, device=dev) if want_diag else None
    check(lib.g2v_vq_fused_assign_bx_fwd(_p(_chk(z)), _p(_chk(w_pre_frag)), _p(_chk(b_pre)), _p(_chk(codebook)), _p(image),
                                         _p(_chk(code_sqnorm)), _p(flat), _p(idx), _p(quant), _p(sse), _p(diag), N, E, K,
                                         int(flags), _stream()), "vq_fused_assign_bx_fwd")
    return (flat, idx, quant, sse, diag) if want_diag else (flat, idx, quant, sse)


def vq_stats(idx, flat, K, out=None):
    N, E = flat.shape
    dev = flat.device
    lib = _lib_()
    if out is None:
        out = torch.empty((K + K * E,), dtype=torch.float32, device=dev)
    ws = workspace(lib.g2v_vq_stats_workspace(N, E, K), dev, "vqstats")
    check(lib.g2v_vq_stats(_p(_chk(idx, torch.int64)), _p(_chk(flat)), _p(out), N, E, K, _p(ws), ws.numel(),
                           _stream()), "vq_stats")
    return out


def vq_ema_update(stats, sse_partial, ema_cluster_size, ema_w, codebook, code_sqnorm, N_loss, N_cnt, E, K,
                  beta, decay, eps, update, scalars=None):
    if scalars is None:
        scalars = torch.empty((2,), dtype=torch.float32, device=stats.device)
    check(_lib_().g2v_vq_ema_update(_p(stats), _p(sse_partial), 0 if sse_partial is None else sse_partial.numel(),
                                    _p(ema_cluster_size), _p(ema_w), _p(codebook), _p(code_sqnorm), _p(scalars),
                                    N_loss, N_cnt, E, K, float(beta), float(decay), float(eps), int(update),
                                    _stream()), "vq_ema_update")
    return scalars


def vq_bwd(g_quantized, g_loss, z, codebook, idx, beta, out=None):
    """idx=None: `codebook` is the dense (N,E) saved `quantized` output."""
    N, E = z.shape
    gz = torch.empty_like(z) if out is None else out
    check(_lib_().g2v_vq_bwd(_p(g_quantized), _p(g_loss), _p(_chk(z)), _p(_chk(codebook)), _p(idx), _p(gz), N, E,
                             float(beta), _stream()), "vq_bwd")
    return gz


# ------------------------------------------------------------------------------------------ GRU direction(s)
def gru_packed_ok(T, B, H) -> bool:
    return bool(_lib_().g2v_gru_seq_packed_ok(int(T), int(B), int(H)))


def _row_off_array(row_off, T):
    """ctypes int32 array of the T packed row offsets (host memory: the library reads it at call time)"""
    if row_off is None:
        return None
    assert len(row_off) == T
    return (C.c_int32 * T)(*[int(v) for v in row_off])


def gru_gather_ok(T, B, H, ndir=2) -> bool:
    """g2v_gru_seq_fwd can gather its input projections from a table (g2v_gru_dir.gi_gather) for this shape"""
    return bool(_lib_().g2v_gru_seq_gather_ok(int(T), int(B), int(H), int(ndir)))


def gru_dirs_fwd(dirs, T, B, H, *, lengths=None, hs_ld=None, row_off=None):
    """dirs: list (1 or 2) of dicts gi, w_hh, b_hh, h0, hs, h_n, gates, reverse -- ONE launch for both directions.
    With gi=None and x, w_ih, b_ih, in_dim given the input projection is fused into the kernel (H == in_dim == 64).
    row_off (T ints): gi is PACKED -- row (t,b) at row_off[t] + b, only positions inside their sequences (g2v.h)."""
    lib = _lib_()
    arr = (_lib.GruDir * len(dirs))()
    ro = _row_off_array(row_off, T)
    for k, d in enumerate(dirs):
        for name in ("gi", "w_hh", "b_hh", "h0", "hs", "h_n", "gates", "x", "w_ih", "b_ih", "gi_gather"):
            setattr(arr[k], name, _p(d.get(name)))
        arr[k].reverse = int(bool(d.get("reverse", False)))
        arr[k].in_dim = int(d.get("in_dim", 0))
        if ro is not None:
            arr[k].gi_row_off = ro
    dev = dirs[0]["hs"].device
    ws = workspace(lib.g2v_gru_seq_fwd_workspace(len(dirs), H), dev, "grufwd")
    check(lib.g2v_gru_seq_fwd(arr, len(dirs), _p(lengths), hs_ld if hs_ld is not None else H, T, B, H, _p(ws),
                              ws.numel(), _stream()), "gru_seq_fwd")


def side_pending() -> bool:
    """True while a side branch forked in this iteration has not been joined (its launches may still be running)"""
    return bool(_side_state["pending"])


def gru_dirs_bwd(dirs, T, B, H, *, lengths=None, d_hs_ld=None, hs_ld=None, row_off=None, allow_resident=True):
    """dirs: list of dicts d_hs, d_hn, hs, h0, gates, w_hh, dgi, dgh, dh0, reverse.  row_off: dgi is PACKED (see gru_dirs_fwd).
    allow_resident=False: this call keeps the streaming BPTT whatever G2V_OPT_GRU_RESIDENT_BWD says (a W_hh-resident launch takes
    every CU whole: a caller with other launches in flight beside it says so)."""
    lib = _lib_()
    arr = (_lib.GruDirBwd * len(dirs))()
    ro = _row_off_array(row_off, T)
    for k, d in enumerate(dirs):
        if ro is not None:
            arr[k].dgi_row_off = ro
        for name in ("d_hs", "d_hn", "hs", "h0", "gates", "w_hh", "dgi", "dgh", "dh0", "w_ih", "dx",
                     "x", "dw_hh", "db_hh", "dw_ih", "db_ih", "wslab"):        # the last six: optional fused weight gradients
            setattr(arr[k], name, _p(d.get(name)))
        arr[k].reverse = int(bool(d.get("reverse", False)))
        arr[k].in_dim = int(d.get("in_dim", 0))
    dev = dirs[0]["hs"].device
    ws = workspace(lib.g2v_gru_seq_bwd_workspace(len(dirs), H), dev, "grubwd")
    prev = lib.g2v_ctx_set_option(None, _lib.OPT_GRU_RESIDENT_BWD, 0) if not allow_resident else None
    try:
        check(lib.g2v_gru_seq_bwd(arr, len(dirs), _p(lengths), d_hs_ld if d_hs_ld is not None else H,
                                  hs_ld if hs_ld is not None else H, T, B, H, _p(ws), ws.numel(), _stream()), "gru_seq_bwd")
    finally:
        if prev is not None:
            lib.g2v_ctx_set_option(None, _lib.OPT_GRU_RESIDENT_BWD, prev)


def gru_seq_fwd(gi, w_hh, b_hh, T, B, H, *, h0=None, lengths=None, reverse=False, hs=None, hs_ld=None,
                save_gates=True):
    dev = gi.device
    if hs is None:
        hs = torch.empty((T, B, H), dtype=torch.float32, device=dev)
        hs_ld = H
    h_n = torch.empty((B, H), dtype=torch.float32, device=dev)
    gates = torch.empty((T, B, 4 * H), dtype=torch.float32, device=dev) if save_gates else None
    gru_dirs_fwd([dict(gi=_chk(gi), w_hh=_chk(w_hh), b_hh=b_hh, h0=h0, hs=hs, h_n=h_n, gates=gates, reverse=reverse)],
                 T, B, H, lengths=lengths, hs_ld=hs_ld)
    return hs, h_n, gates


def gru_seq_bwd(d_hs, d_hs_ld, d_hn, hs, hs_ld, h0, gates, w_hh, T, B, H, *, lengths=None, reverse=False,
                want_dh0=False):
    dev = hs.device
    dgi = torch.empty((T, B, 3 * H), dtype=torch.float32, device=dev)
    dgh = torch.empty((T, B, 3 * H), dtype=torch.float32, device=dev)
    dh0 = torch.empty((B, H), dtype=torch.float32, device=dev) if want_dh0 else None
    gru_dirs_bwd([dict(d_hs=d_hs, d_hn=d_hn, hs=hs, h0=h0, gates=gates, w_hh=w_hh, dgi=dgi, dgh=dgh, dh0=dh0,
                       reverse=reverse)], T, B, H, lengths=lengths, d_hs_ld=d_hs_ld, hs_ld=hs_ld)
    return dgi, dgh, dh0


def gru_cell_ok(in_dim, H, B):
    """shapes served by the one-launch GRU cell (small batch, generic hidden size)"""
    return in_dim % 4 == 0 and H % 4 == 0 and in_dim <= 256 and H <= 256 and B <= 1024


def gru_cell_fwd(x, h_prev, w_ih, w_hh, b_ih, b_hh, *, keep=None, scale=1.0, h_new=None, gates=None):
    """One GRU cell step incl. its input projection (g2v_gru_cell_fwd): x (B,in), h_prev (B,H) -> h_new (B,H), gates (B,4H)."""
    B, in_dim = x.shape
    H = h_prev.shape[1]
    if h_new is None:
        h_new = torch.empty((B, H), dtype=torch.float32, device=x.device)
    check(_lib_().g2v_gru_cell_fwd(_p(_chk(x)), in_dim, _p(keep), float(scale), _p(_chk(h_prev)), _p(_chk(w_ih)), _p(_chk(w_hh)),
                                   _p(b_ih), _p(b_hh), _p(h_new), _p(gates), B, H, _stream()), "gru_cell_fwd")
    return h_new


def gru_cell_bwd(d_h_a, d_h_b, gates, h_prev, w_ih, w_hh, *, keep=None, scale=1.0, dgi, dgh, d_hprev, dx=None):
    """Backward of gru_cell_fwd (g2v_gru_cell_bwd): fills dgi, dgh (B,3H), d_hprev (B,H), dx (B,in) [masked by keep * scale]."""
    B, H = h_prev.shape
    in_dim = w_ih.shape[1]
    check(_lib_().g2v_gru_cell_bwd(_p(d_h_a), _p(d_h_b), _p(_chk(gates)), _p(_chk(h_prev)), _p(_chk(w_ih)), _p(_chk(w_hh)), _p(keep),
                                   float(scale), _p(dgi), _p(dgh), _p(d_hprev), _p(dx), in_dim, B, H, _stream()), "gru_cell_bwd")


# ------------------------------------------------------------------------------------------ decoder rollout
def dec_weights_struct(wd: dict) -> DecWeights:
    s = DecWeights()
    for name, _ in DecWeights._fields_:
        setattr(s, name, _p(wd[name]))
    return s


def struct_from(cls, d: dict):
    """ctypes struct from a dict of tensors; array fields (e.g. DecGrads.dw_gru) take a list of tensors / None entries"""
    s = cls()
    for name, typ in cls._fields_:
        v = d.get(name)
        if isinstance(typ, type) and issubclass(typ, C.Array):
            for k, t in enumerate(v or ()):
                getattr(s, name)[k] = _p(t)
        else:
            setattr(s, name, _p(v))
    return s


def dec_rollout_blocks(B):
    return _lib_().g2v_dec_rollout_blocks(B)


def dec_rollout_fwd(target, h_init, wstruct, saved: dict, keep95, keep_l0, p_drop, n_pre, conditioned, training,
                    T, B, D, H):
    sv = struct_from(DecSaved, saved)
    lib = _lib_()
    ws = workspace(lib.g2v_dec_rollout_fwd_workspace(D, H), target.device, "decfwd")
    check(lib.g2v_dec_rollout_fwd(_p(_chk(target)), _p(_chk(h_init)), C.byref(wstruct), C.byref(sv),
                                  _p(_chk(keep95, torch.uint8)), _p(keep_l0), float(p_drop), int(n_pre),
                                  int(conditioned), int(training), T, B, D, H, _p(ws), ws.numel(), _stream()),
          "dec_rollout_fwd")


def dec_rollout_bwd(wstruct, saved: dict, grads: dict, keep95, keep_l0, p_drop, n_pre, conditioned, T, B, D, H):
    lib = _lib_()
    sv = struct_from(DecSaved, saved)
    gr = struct_from(DecGrads, grads)
    ws = workspace(lib.g2v_dec_rollout_bwd_workspace(D, H), grads["dy"].device, "decbwd")
    check(lib.g2v_dec_rollout_bwd(C.byref(wstruct), C.byref(sv), C.byref(gr), _p(keep95), _p(keep_l0),
                                  float(p_drop), int(n_pre), int(conditioned), T, B, D, H, _p(ws), ws.numel(),
                                  _stream()), "dec_rollout_bwd")


# ------------------------------------------------------------------------------------------ Part d: fused code-decoder rollout
def code_rollout_ok(S1, B, H, K, Tw, att) -> bool:
    return bool(_lib_().g2v_attn_code_rollout_ok(int(S1), int(B), int(H), int(K), int(Tw), int(bool(att))))


def code_rollout_cluster_ok(S1, B, H, K, att) -> bool:
    """g2v_attn_code_rollout_fwd runs this shape as one persistent cluster launch (small batch, no attention: include/g2v.h)"""
    return bool(_lib_().g2v_attn_code_rollout_cluster_ok(int(S1), int(B), int(H), int(K), int(bool(att))))


def code_cluster_bptt(dh_top, weights: dict, saved: dict, keep_l0, p_drop, S1, B, H):
    """g2v_code_cluster_bptt: the GRU cells' BPTT of the attention-free rollout as one persistent cluster launch (small batch).
    Returns (dgi0, dgh0, dgi1, dgh1 (S1,B,3H), da (S1,B,H), d_hidden0 (2,B,H))."""
    lib = _lib_()
    dev = dh_top.device
    f32 = lambda *sh: torch.empty(sh, dtype=torch.float32, device=dev)
    dgi0, dgh0, dgi1, dgh1 = f32(S1, B, 3 * H), f32(S1, B, 3 * H), f32(S1, B, 3 * H), f32(S1, B, 3 * H)
    da, d_h0 = f32(S1, B, H), f32(2, B, H)
    ws = workspace(lib.g2v_code_cluster_bptt_workspace(B, H), dev, "codebptt")
    check(lib.g2v_code_cluster_bptt(_p(_chk(dh_top)), C.byref(struct_from(CodeDecWeights, weights)), C.byref(struct_from(CodeDecSaved, saved)),
                                    _p(keep_l0), float(p_drop), _p(dgi0), _p(dgh0), _p(dgi1), _p(dgh1), _p(da), _p(d_h0), S1, B, H,
                                    _p(ws), ws.numel(), _stream()), "code_cluster_bptt")
    return dgi0, dgh0, dgi1, dgh1, da, d_h0


def code_rollout_fwd(codes, h_init, enc, enc_proj, weights: dict, saved: dict, keep_emb, keep_l0, p_drop, n_pre, training,
                     S1, B, H, K, Tw):
    """g2v_attn_code_rollout_fwd: S1 + 1 launches of the fused decoder-step kernel (weights / saved: dicts of tensors named as
    the fields of g2v_code_dec_weights / g2v_code_dec_saved; missing = NULL)."""
    lib = _lib_()
    att = weights.get("w_attn") is not None
    ws = workspace(lib.g2v_attn_code_rollout_fwd_workspace(H, K, int(att)), h_init.device, "codefwd")
    check(lib.g2v_attn_code_rollout_fwd(_p(_chk(codes, torch.int64)), _p(_chk(h_init)), _p(enc), _p(enc_proj),
                                        C.byref(struct_from(CodeDecWeights, weights)), C.byref(struct_from(CodeDecSaved, saved)),
                                        _p(keep_emb), _p(keep_l0), float(p_drop), int(n_pre), int(training), S1, B, H, K, int(Tw),
                                        _p(ws), ws.numel(), _stream()), "attn_code_rollout_fwd")


def code_rollout_bwd(d_logits, enc, enc_proj, weights: dict, saved: dict, grads: dict, keep_emb, keep_l0, p_drop, S1, B, H, K, Tw):
    lib = _lib_()
    att = weights.get("w_attn") is not None
    ws = workspace(lib.g2v_attn_code_rollout_bwd_workspace(S1, B, H, K, int(Tw), int(att)), d_logits.device, "codebwd")
    check(lib.g2v_attn_code_rollout_bwd(_p(_chk(d_logits)), _p(enc), _p(enc_proj), C.byref(struct_from(CodeDecWeights, weights)),
                                        C.byref(struct_from(CodeDecSaved, saved)), C.byref(struct_from(CodeDecGrads, grads)),
                                        _p(keep_emb), _p(keep_l0), float(p_drop), S1, B, H, K, int(Tw), _p(ws), ws.numel(),
                                        _stream()), "attn_code_rollout_bwd")


# ------------------------------------------------------------------------------------------ loss / optimiser / rng
def custom_loss_fwd_bwd(y_tbd, target_btd, w_l1, w_cont, w_var, g_scale=1.0, want_grad=True):
    T, B, D = y_tbd.shape
    dev = y_tbd.device
    lib = _lib_()
    dy = torch.empty_like(y_tbd) if want_grad else None
    terms = torch.empty((5,), dtype=torch.float32, device=dev)
    partial = torch.empty((lib.g2v_custom_loss_blocks(B, D) * 4,), dtype=torch.float32, device=dev)
    check(lib.g2v_custom_loss_fwd_bwd(_p(_chk(y_tbd)), _p(_chk(target_btd)), _p(dy), _p(terms), _p(partial),
                                      float(w_l1), float(w_cont), float(w_var), float(g_scale), T, B, D, _stream()),
          "custom_loss_fwd_bwd")
    return terms, dy


def clip_adam_step(param, grad, m, v, step_counter, partial, gnorm_out, max_norm, grad_scale, lr, b1, b2, eps):
    check(_lib_().g2v_clip_adam_step(_p(_chk(param)), _p(_chk(grad)), _p(m), _p(v), param.numel(), _p(partial),
                                     _p(step_counter), _p(gnorm_out), float(max_norm), float(grad_scale), float(lr),
                                     float(b1), float(b2), float(eps), _stream()), "clip_adam_step")


def adam_blocks(n):
    return _lib_().g2v_adam_blocks(n)


def keep_mask(out_u8, keep_prob, seed, offset_counter):
    check(_lib_().g2v_keep_mask(_p(_chk(out_u8, torch.uint8)), out_u8.numel(), float(keep_prob), int(seed),
                                _p(offset_counter), _stream()), "keep_mask")
    return out_u8


def keep_mask_at(out_u8, keep_prob, seed, offset_counter, offset_add):
    """the mask keep_mask would draw after `offset_add` earlier draws, without bumping the counter (counter_add does, once)"""
    check(_lib_().g2v_keep_mask_at(_p(_chk(out_u8, torch.uint8)), out_u8.numel(), float(keep_prob), int(seed),
                                   _p(offset_counter), int(offset_add), _stream()), "keep_mask_at")
    return out_u8


def counter_add(counter, n):
    check(_lib_().g2v_counter_add(_p(_chk(counter, torch.int64)), int(n), _stream()), "counter_add")


def fill(t, v):
    check(_lib_().g2v_fill_f32(_p(t), float(v), t.numel(), _stream()), "fill")
    return t


def add_halves(a, lda, b, ldb, out, ldo, M, H):
    check(_lib_().g2v_add_halves(_p(a), lda, _p(b), ldb, _p(out), ldo, M, H, _stream()), "add_halves")
    return out


def scale(inp, scalar_dev, out=None):
    if out is None:
        out = torch.empty_like(inp)
    check(_lib_().g2v_scale_f32(_p(_chk(inp)), _p(scalar_dev), _p(out), inp.numel(), _stream()), "scale")
    return out


def mse_fwd_bwd(y, target, want_grad=True, g_scale=1.0):
    lib = _lib_()
    n = y.numel()
    dy = torch.empty_like(y) if want_grad else None
    loss = torch.empty((1,), dtype=torch.float32, device=y.device)
    partial = torch.empty((lib.g2v_mse_blocks(n),), dtype=torch.float32, device=y.device)
    check(lib.g2v_mse_fwd_bwd(_p(_chk(y)), _p(_chk(target)), _p(dy), _p(loss), _p(partial), n, float(g_scale), _stream()),
          "mse_fwd_bwd")
    return loss, dy


def vq_codebook_grad(stats, codebook, g_loss, N):
    K, E = codebook.shape
    out = torch.empty_like(codebook)
    check(_lib_().g2v_vq_codebook_grad(_p(stats), _p(_chk(codebook)), _p(g_loss), _p(out), N, E, K, _stream()),
          "vq_codebook_grad")
    return out


def mask_mul(inp, mask, scale=1.0, positive_of=False, out=None):
    """out = inp * scale where the mask is on (uint8 keep mask, or `mask > 0` for a float tensor when positive_of);
    out may be inp (elementwise)."""
    if out is None:
        out = torch.empty_like(inp)
    keep = None if positive_of else mask
    pos = mask if positive_of else None
    check(_lib_().g2v_mask_mul(_p(_chk(inp)), _p(keep), _p(pos), float(scale), _p(out), inp.numel(), _stream()), "mask_mul")
    return out


# ------------------------------------------------------------------------------------------ Part d operators
def embedding_fwd(table, ids, keep=None, scale=1.0, out=None, ldo=None):
    V, dim = table.shape
    n = ids.numel()
    if out is None:
        out, ldo = torch.empty((n, dim), dtype=torch.float32, device=table.device), dim
    check(_lib_().g2v_embedding_fwd(_p(_chk(table)), _p(_chk(ids, torch.int64)), _p(keep), float(scale), out.data_ptr(),
                                    ldo if ldo is not None else dim, n, dim, V, _stream()), "embedding_fwd")
    return out


def embedding_bwd(d_out, ids, V, keep=None, scale=1.0, reuse_sort=False):
    """reuse_sort: the PREVIOUS embedding_bwd call on this stream had the same ids, n, V and dim (its sort is still in the workspace)"""
    n, dim = d_out.shape
    d_table = torch.empty((V, dim), dtype=torch.float32, device=d_out.device)
    nb = int(_lib_().g2v_embedding_bwd_ws_bytes(n, dim, V))
    ws = workspace(nb, d_out.device, "embedding_bwd")
    check(_lib_().g2v_embedding_bwd(_p(_chk(d_out)), _p(_chk(ids, torch.int64)), _p(keep), float(scale), _p(d_table), n, dim,
                                    V, 1 | (2 if reuse_sort else 0), _p(ws), nb, _stream()), "embedding_bwd")
    return d_table


def batchnorm_fwd(x, weight, bias, running_mean, running_var, training, relu=True, out=None, save=None):
    """out: y buffer; save: (save_mean, save_invstd) buffers (training)"""
    B, H = x.shape
    y = torch.empty_like(x) if out is None else out
    if save is not None:
        sm, si = save
    else:
        sm = torch.empty((H,), dtype=torch.float32, device=x.device) if training else None
        si = torch.empty((H,), dtype=torch.float32, device=x.device) if training else None
    check(_lib_().g2v_batchnorm_fwd(_p(_chk(x)), _p(weight), _p(bias), _p(running_mean), _p(running_var), int(training),
                                    int(relu), _p(y), _p(sm), _p(si), B, H, _stream()), "batchnorm_fwd")
    return y, sm, si


def bn_running_update_invstd(save_mean, save_invstd, step_stride, running_mean, running_var, steps, H, B):
    """Deferred commit of BatchNorm's running statistics from the saved (mean, invstd) of `steps` training calls that were given
    no running statistics; a no-op on the device while the persistent kernels' fault latch is set (include/g2v.h)."""
    check(_lib_().g2v_bn_running_update_invstd(_p(save_mean), _p(save_invstd), int(step_stride), _p(running_mean), _p(running_var),
                                               int(steps), int(H), int(B), _stream()), "bn_running_update_invstd")


def batchnorm_bwd_steps(dy, x, y, weight, save_mean, save_invstd, relu=True):
    """BatchNorm's backward of the `steps` calls of a decode loop in one launch: dy, x, y (steps,B,H), save_* (steps,H) ->
    dx (steps,B,H) and the parameters' gradients summed over the steps (include/g2v.h: g2v_batchnorm_bwd_steps; B < 1024)."""
    S, B, H = x.shape
    dx = torch.empty_like(x)
    dw = torch.empty((H,), dtype=torch.float32, device=x.device)
    db = torch.empty((H,), dtype=torch.float32, device=x.device)
    if save_mean.stride(1) != 1 or save_invstd.stride(1) != 1 or save_mean.stride(0) != save_invstd.stride(0):
        raise ValueError("save_mean / save_invstd: rows of H contiguous floats at one common row stride")
    check(_lib_().g2v_batchnorm_bwd_steps(_p(_chk(dy)), _p(_chk(x)), _p(_chk(y)), _p(weight), _p(save_mean), _p(save_invstd),
                                          int(save_mean.stride(0)), int(relu), _p(dx), _p(dw), _p(db), S, B, H, _stream()),
          "batchnorm_bwd_steps")
    return dx, dw, db


def one_hot_rows(ids, K, out):
    """out[r, :] = one_hot(ids[r]) (float32; out a (M,K) view with contiguous rows)"""
    check(_lib_().g2v_one_hot_rows(_p(_chk(ids, torch.int64)), _p(out), int(out.stride(0)), int(ids.numel()), int(K), _stream()),
          "one_hot_rows")
    return out


def batchnorm_bwd(dy, x, y, weight, save_mean, save_invstd, relu=True, out=None):
    """out: (dx, dw, db) buffers"""
    B, H = x.shape
    if out is not None:
        dx, dw, db = out
    else:
        dx = torch.empty_like(x)
        dw = torch.empty((H,), dtype=torch.float32, device=x.device)
        db = torch.empty((H,), dtype=torch.float32, device=x.device)
    check(_lib_().g2v_batchnorm_bwd(_p(_chk(dy)), _p(x), _p(y), _p(weight), _p(save_mean), _p(save_invstd), int(relu), _p(dx),
                                    _p(dw), _p(db), B, H, _stream()), "batchnorm_bwd")
    return dx, dw, db


def cross_entropy_fwd_bwd(logits, targets, want_grad=True, ld=None, dl_out=None):
    M, K = logits.shape
    dev = logits.device
    loss = torch.empty((1,), dtype=torch.float32, device=dev)
    row = torch.empty((M,), dtype=torch.float32, device=dev)
    dl = (dl_out if dl_out is not None else torch.empty((M, K), dtype=torch.float32, device=dev)) if want_grad else None
    if dl is not None and (tuple(dl.shape) != (M, K) or not dl.is_contiguous()):
        raise ValueError("cross_entropy: dl_out must be a contiguous (M, K) array")
    check(_lib_().g2v_cross_entropy_fwd_bwd(_p(logits), ld if ld is not None else K, _p(_chk(targets, torch.int64)), _p(loss),
                                            _p(row), _p(dl), K, M, K, 1.0, _stream()), "cross_entropy")
    return loss, dl


def argmax_rows(x, out=None):
    M, K = x.shape
    if out is None:
        out = torch.empty((M,), dtype=torch.int64, device=x.device)
    check(_lib_().g2v_argmax_rows(_p(_chk(x)), K, _p(out), M, K, _stream()), "argmax_rows")
    return out


def attn_fwd(hp, ep, enc, v, ctx_out=None, ldctx=None, weights=None):
    """Bahdanau attention step: hp (B,H), ep/enc (T,B,H), v (H) -> weights (B,T), context (B,H) (optionally written
    into `ctx_out` with row stride ldctx, e.g. the second half of the decoder's (B,2H) input)."""
    T, B, H = enc.shape
    if weights is None:
        weights = torch.empty((B, T), dtype=torch.float32, device=enc.device)
    if ctx_out is None:
        ctx_out, ldctx = torch.empty((B, H), dtype=torch.float32, device=enc.device), H
    check(_lib_().g2v_attn_fwd(_p(_chk(hp)), _p(_chk(ep)), _p(_chk(enc)), _p(_chk(v)), _p(weights), ctx_out.data_ptr(),
                               ldctx, T, B, H, _stream()), "attn_fwd")
    return weights, ctx_out


def attn_step_fwd(logits, ids, table, keep, emb_scale, ec, hp, ep, enc, v, weights):
    """The row-local head of a decode step in one launch (include/g2v.h: g2v_attn_step_fwd): ids[b] = argmax(logits[b]) (logits
    None: ids given), ec[b, :H] = table[ids[b]] * keep * emb_scale, ec[b, H:] = attention context, weights (B,T)."""
    T, B, H = enc.shape
    K = table.shape[0]
    check(_lib_().g2v_attn_step_fwd(_p(logits), int(logits.stride(0)) if logits is not None else 0, K, _p(_chk(ids, torch.int64)),
                                    _p(_chk(table)), _p(keep), float(emb_scale), _p(ec), int(ec.stride(0)), _p(_chk(hp)), _p(_chk(ep)),
                                    _p(_chk(enc)), _p(_chk(v)), _p(weights), T, B, H, _stream()), "attn_step_fwd")


def slab_sum(slabs, out, accumulate=False):
    """out (+)= slabs.sum(0) in slab order (slabs (n, ...) contiguous)"""
    n = slabs.shape[0]
    check(_lib_().g2v_slab_sum(_p(_chk(slabs)), n, slabs.numel() // n, _p(out), int(bool(accumulate)), _stream()), "slab_sum")
    return out


def linear_fwd_dual(x, w_a, bias_a, out_a, w_b, bias_b, out_b):
    """out_a = x w_a^T + bias_a, out_b = x w_b^T + bias_b: one launch at small row counts (include/g2v.h: g2v_linear_fwd_dual)"""
    M, K = x.shape
    check(_lib_().g2v_linear_fwd_dual(_p(x), int(x.stride(0)), _p(_chk(w_a, name="w_a")), _p(bias_a), _p(out_a), int(out_a.stride(0)),
                                      w_a.shape[0], _p(_chk(w_b, name="w_b")), _p(bias_b), _p(out_b), int(out_b.stride(0)),
                                      w_b.shape[0], M, K, _stream()), "linear_fwd_dual")


def attn_bwd(d_ctx, hp, ep, enc, v, weights, ldd=None, out=None, accumulate=False, dv_slab=None):
    """out: (d_hp, d_ep, d_enc, d_v) buffers; accumulate adds into d_ep / d_enc / d_v (d_hp is always overwritten).
    dv_slab (B,H): d_v's per-row partials go there and stay unsummed (d_v is not touched: the caller sums the steps' slabs once)."""
    T, B, H = enc.shape
    dev = enc.device
    if out is not None:
        d_hp, d_ep, d_enc, d_v = out
    else:
        d_hp = torch.empty((B, H), dtype=torch.float32, device=dev)
        d_ep = torch.empty((T, B, H), dtype=torch.float32, device=dev)
        d_enc = torch.empty((T, B, H), dtype=torch.float32, device=dev)
        d_v = torch.empty((H,), dtype=torch.float32, device=dev)
    if dv_slab is not None:
        check(_lib_().g2v_attn_bwd(d_ctx.data_ptr(), ldd if ldd is not None else H, _p(hp), _p(ep), _p(enc), _p(v), _p(weights),
                                   _p(d_hp), _p(d_ep), _p(d_enc), None, int(bool(accumulate)), T, B, H, _p(_chk(dv_slab)),
                                   dv_slab.numel() * 4, _stream()), "attn_bwd")
        return d_hp, d_ep, d_enc, d_v
    nb = _lib_().g2v_attn_bwd_workspace(B, H)
    ws = workspace(nb, dev, "attn")
    check(_lib_().g2v_attn_bwd(d_ctx.data_ptr(), ldd if ldd is not None else H, _p(hp), _p(ep), _p(enc), _p(v), _p(weights),
                               _p(d_hp), _p(d_ep), _p(d_enc), _p(d_v), int(bool(accumulate)), T, B, H, _p(ws), ws.numel(),
                               _stream()), "attn_bwd")
    return d_hp, d_ep, d_enc, d_v


# ------------------------------------------------------------------------------------------ soft quantiser (GSSoft)
def vq_soft_fwd(flat, dots, logvar, wsq, want_perplexity=True):
    """dots (N,K) = flat W^T is overwritten with the squared distances; returns (probs (N,K), dist, perplexity (1,))."""
    N, E = flat.shape
    K = dots.shape[1]
    probs = torch.empty((N, K), dtype=torch.float32, device=flat.device)
    perp = torch.empty((1,), dtype=torch.float32, device=flat.device) if want_perplexity else None
    check(_lib_().g2v_vq_soft_fwd(_p(_chk(flat)), _p(_chk(dots)), _p(_chk(logvar)), _p(_chk(wsq)), _p(probs), None, N, E, K,
                                  _stream()), "vq_soft_fwd")
    if want_perplexity:        # its own device-wide pair of launches (the one built into g2v_vq_soft_fwd is a single workgroup)
        ws = torch.empty(int(_lib_().g2v_vq_soft_perplexity_workspace(N, K)), dtype=torch.uint8, device=flat.device)
        check(_lib_().g2v_vq_soft_perplexity(_p(probs), _p(perp), N, K, _p(ws), ws.numel(), _stream()), "vq_soft_perplexity")
    return probs, dots, perp


def vq_soft_bwd(probs, dprobs, dist, logvar):
    N, K = probs.shape
    dd = torch.empty_like(probs)
    dlv = torch.empty_like(probs)
    rowsum = torch.empty((N,), dtype=torch.float32, device=probs.device)
    check(_lib_().g2v_vq_soft_bwd(_p(_chk(probs)), _p(_chk(dprobs)), _p(_chk(dist)), _p(_chk(logvar)), _p(dd), _p(dlv), _p(rowsum),
                                  N, K, _stream()), "vq_soft_bwd")
    return dd, dlv, rowsum


def vq_soft_fused_ok(N, E, K):
    return bool(_lib_().g2v_vq_soft_fused_ok(N, E, K))


def vq_soft_fused_fwd(x, w_mean, b_mean, w_logvar, b_logvar, codebook, beta, g_scale=1.0, want_perplexity=True):
    """The whole soft-quantiser forward (csrc/vq_soft.hip): returns a dict flat, logvar, dist, probs, q, dq, quant, mse (1,),
    loss_vq (1,), perplexity (1,) or None."""
    lib = _lib_()
    N, E = x.shape
    K = codebook.shape[0]
    dev = x.device
    f = lambda *s: torch.empty(s, dtype=torch.float32, device=dev)
    wsq = vq_code_sqnorm(codebook)
    nblk = lib.g2v_vq_soft_fused_blocks(N)
    o = {"flat": f(N, E), "logvar": f(N, K), "dist": f(N, K), "probs": f(N, K), "q": f(N, E), "dq": f(N, E), "quant": f(N, E),
         "mse": f(1), "loss_vq": f(1), "perplexity": f(1) if want_perplexity else None}
    part, colsum = f(nblk), f(nblk, K) if want_perplexity else None
    check(lib.g2v_vq_soft_fused_fwd(_p(_chk(x)), _p(_chk(w_mean)), _p(_chk(b_mean)), _p(_chk(w_logvar)), _p(_chk(b_logvar)),
                                    _p(_chk(codebook)), _p(wsq), _p(o["flat"]), _p(o["logvar"]), _p(o["dist"]), _p(o["probs"]),
                                    _p(o["q"]), _p(o["dq"]), _p(o["quant"]), _p(part), _p(colsum), float(g_scale), N, E, K, _stream()),
          "vq_soft_fused_fwd")
    opb = torch.full((1,), 1.0 + float(beta), dtype=torch.float32, device=dev)
    check(lib.g2v_vq_soft_finish(_p(part), _p(colsum), _p(opb), _p(o["mse"]), _p(o["loss_vq"]), _p(o["perplexity"]), N, E, K,
                                 _stream()), "vq_soft_finish")
    return o


def vq_soft_fused_bwd(dh, g_loss, x, fwd, w_mean, w_logvar, codebook, beta):
    """Backward of vq_soft_fused_fwd (`fwd` = its dict): returns (gz (N,E), dd (N,K), dlogvar (N,K), dflat (N,E))."""
    N, E = x.shape
    K = codebook.shape[0]
    gz, dflat = torch.empty_like(x), torch.empty_like(x)
    dd, dlv = torch.empty_like(fwd["probs"]), torch.empty_like(fwd["probs"])
    check(_lib_().g2v_vq_soft_fused_bwd(_p(dh), _p(g_loss), _p(_chk(x)), _p(fwd["q"]), _p(fwd["dq"]), _p(fwd["flat"]), _p(fwd["probs"]),
                                        _p(fwd["dist"]), _p(fwd["logvar"]), _p(_chk(w_mean)), _p(_chk(w_logvar)), _p(_chk(codebook)),
                                        _p(dd), _p(dlv), _p(dflat), _p(gz), float(beta), N, E, K, _stream()), "vq_soft_fused_bwd")
    return gz, dd, dlv, dflat


def rowscale_combine(a, v, t):
    """out[r,c] = 2 a[r,c] v[r] - 2 t[r,c]"""
    rows, cols = a.shape
    out = torch.empty_like(a)
    check(_lib_().g2v_rowscale_combine(_p(_chk(a)), _p(_chk(v)), _p(_chk(t)), _p(out), rows, cols, _stream()), "rowscale_combine")
    return out


def ste(z, q):
    out = torch.empty_like(z)
    check(_lib_().g2v_ste_f32(_p(_chk(z)), _p(_chk(q)), _p(out), z.numel(), _stream()), "ste")
    return out
