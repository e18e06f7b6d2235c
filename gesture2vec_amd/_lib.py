"""ctypes binding of libg2v_hip.so (include/g2v.h).  No torch types cross the boundary: only raw
device pointers, sizes and a hipStream_t.  Loading FAILS LOUDLY when the library is missing; there is
no CPU or PyTorch fallback anywhere in the product path."""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libg2v_hip.so")

c_fp = C.c_void_p  # device pointers travel as void*
c_i64 = C.c_int64
c_int = C.c_int
c_f = C.c_float
c_sz = C.c_size_t
c_u64 = C.c_uint64


class DecWeights(C.Structure):
    """g2v_dec_weights"""
    _fields_ = [(n, c_fp) for n in (
        "w_pre", "b_pre", "bn_w", "bn_b", "bn_running_mean", "bn_running_var",
        "w_ih0", "w_hh0", "b_ih0", "b_hh0", "w_ih1", "w_hh1", "b_ih1", "b_hh1", "w_out", "b_out")]


class DecSaved(C.Structure):
    """g2v_dec_saved"""
    _fields_ = [(n, c_fp) for n in (
        "y", "xin", "u", "a", "h0", "h1", "x1", "gates0", "gates1", "bn_partial", "bn_stats",
        "loss_code", "loss_coef", "loss_partial", "loss_terms")] + [("loss_w", c_f * 3)]     # optional: custom_loss by the chaser + the backward


class GruDir(C.Structure):
    """g2v_gru_dir"""
    _fields_ = ([(n, c_fp) for n in ("gi", "w_hh", "b_hh", "h0", "hs", "h_n", "gates")] + [("reverse", c_int)] +
                [(n, c_fp) for n in ("x", "w_ih", "b_ih")] + [("in_dim", c_int)] +
                [("gi_row_off", C.POINTER(C.c_int32)),                  # HOST array of T packed row offsets, or NULL
                 ("gi_gather", c_fp)])                                    # DEVICE int64 row indices into a gi TABLE, or NULL


class WgradItem(C.Structure):
    """g2v_wgrad_item"""
    _fields_ = [(n, c_fp) for n in ("dy", "x", "dw", "db")]


class WgradPending(C.Structure):
    """g2v_wgrad_pending"""
    _fields_ = ([("slab_w", c_fp * 4), ("out_w", c_fp * 4), ("slab_b", c_fp * 4), ("out_b", c_fp * 4), ("n", c_i64), ("nb", c_i64)] +
                [(n, c_int) for n in ("nsplit", "nprob", "accumulate", "reserved")])


class GruDirBwd(C.Structure):
    """g2v_gru_dir_bwd"""
    _fields_ = ([(n, c_fp) for n in ("d_hs", "d_hn", "hs", "h0", "gates", "w_hh", "dgi", "dgh", "dh0")] + [("reverse", c_int)] +
                [(n, c_fp) for n in ("w_ih", "dx")] + [("in_dim", c_int)] +
                [(n, c_fp) for n in ("x", "dw_hh", "db_hh", "dw_ih", "db_ih", "wslab")] +      # optional fused weight gradients
                [(n, c_fp) for n in ("hn_z", "hn_q", "hn_gloss")] + [("hn_coef", c_f)] +        # optional fused quantiser backward
                [("dgi_row_off", C.POINTER(C.c_int32))])                # HOST array of T packed row offsets, or NULL


class CodeDecWeights(C.Structure):
    """g2v_code_dec_weights"""
    _fields_ = [(n, c_fp) for n in (
        "emb", "w_pre", "b_pre", "bn_w", "bn_b", "bn_running_mean", "bn_running_var",
        "w_ih0", "w_hh0", "b_ih0", "b_hh0", "w_ih1", "w_hh1", "b_ih1", "b_hh1", "w_out", "b_out", "w_attn", "b_attn", "v_attn")]


class CodeDecSaved(C.Structure):
    """g2v_code_dec_saved"""
    _fields_ = [(n, c_fp) for n in (
        "ids", "ec", "u", "a", "bn_stats", "h0", "h1", "x1", "gates0", "gates1", "logits", "bn_partial", "hp", "attw")]


class CodeDecGrads(C.Structure):
    """g2v_code_dec_grads"""
    _fields_ = [(n, c_fp) for n in (
        "d_hidden0", "d_emb", "d_w_pre", "d_b_pre", "d_bn_w", "d_bn_b", "d_w_ih0", "d_w_hh0", "d_b_ih0", "d_b_hh0",
        "d_w_ih1", "d_w_hh1", "d_b_ih1", "d_b_hh1", "d_w_out", "d_b_out", "d_w_attn", "d_b_attn", "d_v_attn", "d_enc")]


class DecGrads(C.Structure):
    """g2v_dec_grads"""
    _fields_ = [(n, c_fp) for n in (
        "dy", "du", "dbn", "dgi0", "dgh0", "dgi1", "dgh1", "dh_init", "d_bn_w", "d_bn_b", "bn_bwd_partial")] + [
        ("dw_gru", c_fp * 4), ("db_gru", c_fp * 4)]          # optional fused GRU weight / bias gradients: ih0, hh0, ih1, hh1


_SIGS = {
    "g2v_version": (C.c_char_p, []),
    "g2v_last_error": (C.c_char_p, []),
    "g2v_device_ok": (c_int, []),
    "g2v_linear_fwd": (c_int, [c_fp, c_i64, c_int, c_i64, c_i64, c_fp, c_f, c_fp, c_fp, c_fp, c_i64,
                               c_int, c_int, c_int, c_int, c_fp]),
    "g2v_linear_compose2": (c_int, [c_fp] * 10 + [c_int, c_int, c_int, c_fp]),
    "g2v_linear_fwd_pair": (c_int, [c_fp, c_i64, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_i64, c_int, c_int, c_int, c_int, c_fp]),
    "g2v_linear_bwd_data": (c_int, [c_fp, c_i64, c_fp, c_fp, c_i64, c_int, c_int, c_int, c_int, c_fp]),
    "g2v_linear_bwd_weight_workspace": (c_sz, [c_int, c_int, c_int]),
    "g2v_linear_bwd_weight": (c_int, [c_fp, c_i64, c_fp, c_i64, c_int, c_i64, c_i64, c_fp, c_f, c_fp, c_fp,
                                      c_int, c_int, c_int, c_int, c_fp, c_sz, c_fp]),
    "g2v_linear_bwd_weight_sum2_ok": (c_int, [c_int, c_int, c_int]),
    "g2v_linear_bwd_weight_sum2": (c_int, [c_fp, c_fp, c_i64, c_fp, c_i64, c_int, c_i64, c_i64, c_fp, c_fp, c_int, c_int, c_int,
                                           c_int, c_fp, c_sz, c_fp]),
    "g2v_linear_bwd_weight_batch": (c_int, [C.POINTER(WgradItem), c_int, c_i64, c_i64, c_int, c_int, c_int, c_int, c_fp, c_sz, c_fp]),
    "g2v_linear_bwd_weight_batch_mapped": (c_int, [C.POINTER(WgradItem), c_int, c_i64, c_i64, c_int, c_i64, c_i64, c_int, c_int, c_int,
                                                   c_int, c_fp, c_sz, c_fp]),
    "g2v_cluster_exchange_preclear_drop": (c_int, [c_fp, c_sz]),
    "g2v_ctx_create": (C.c_void_p, []),
    "g2v_ctx_destroy": (None, [C.c_void_p]),
    "g2v_ctx_bind": (C.c_void_p, [C.c_void_p]),
    "g2v_ctx_set_option": (c_int, [C.c_void_p, c_int, c_int]),
    "g2v_ctx_get_option": (c_int, [C.c_void_p, c_int]),
    "g2v_linear_bwd_weight_deferred": (c_int, [C.POINTER(WgradItem), c_int, c_i64, c_i64, c_int, c_i64, c_i64, c_fp, c_int, c_int, c_int,
                                               c_int, c_fp, c_sz, C.POINTER(WgradPending), c_fp]),
    "g2v_linear_bwd_weight_reduce": (c_int, [C.POINTER(WgradPending), c_int, c_fp]),
    "g2v_linear_bwd_weight_chain2": (c_int, [c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_int, c_int, c_int, c_int, c_fp]),
    "g2v_linear_bwd_weight_fold_chain2": (c_int, [c_fp] * 12 + [c_int, c_int, c_int, c_fp]),
    "g2v_linear_bwd_weight_fold2": (c_int, [c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_int, c_int, c_int, c_int, c_fp]),
    "g2v_vq_assign_blocks": (c_int, [c_int]),
    "g2v_vq_code_sqnorm": (c_int, [c_fp, c_fp, c_int, c_int, c_fp]),
    "g2v_vq_assign_fwd": (c_int, [c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_int, c_int, c_int, c_fp]),
    "g2v_vq_fused_assign_fwd": (c_int, [c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_int, c_int, c_int, c_fp]),
    "g2v_vq_pack_codebook": (c_int, [c_fp, c_fp, c_int, c_int, c_fp]),
    "g2v_vq_assign_packed_ok": (c_int, [c_int, c_int, c_int]),
    "g2v_vq_assign_packed_fwd": (c_int, [c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_int, c_int, c_int, c_fp]),
    "g2v_vq_fused_assign_packed_fwd": (c_int, [c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_int, c_int, c_int, c_fp]),
    "g2v_vq_bx_image_bytes": (c_sz, [c_int, c_int]),
    "g2v_vq_bx_pack": (c_int, [c_fp, c_fp, c_fp, c_fp, c_fp, c_int, c_int, c_fp]),
    "g2v_vq_fused_assign_bx_ok": (c_int, [c_int, c_int, c_int]),
    "g2v_vq_fused_assign_bx_fwd": (c_int, [c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_int, c_int, c_int, c_int, c_fp]),
    "g2v_vq_stats_workspace": (c_sz, [c_int, c_int, c_int]),
    "g2v_vq_stats": (c_int, [c_fp, c_fp, c_fp, c_int, c_int, c_int, c_fp, c_sz, c_fp]),
    "g2v_vq_ema_update": (c_int, [c_fp, c_fp, c_int, c_fp, c_fp, c_fp, c_fp, c_fp, c_int, c_int, c_int, c_int,
                                  c_f, c_f, c_f, c_int, c_fp]),
    "g2v_vq_bwd": (c_int, [c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_int, c_int, c_f, c_fp]),
    "g2v_gru_seq_packed_ok": (c_int, [c_int, c_int, c_int]),
    "g2v_gru_seq_gather_ok": (c_int, [c_int, c_int, c_int, c_int]),
    "g2v_gru_seq_fwd_workspace": (c_sz, [c_int, c_int]),
    "g2v_gru_seq_fwd": (c_int, [C.POINTER(GruDir), c_int, c_fp, c_i64, c_int, c_int, c_int, c_fp, c_sz, c_fp]),
    "g2v_gru_seq_bwd_workspace": (c_sz, [c_int, c_int]),
    "g2v_gru_seq_bwd_wslab_bytes": (c_sz, [c_int, c_int]),
    "g2v_gru_seq_bwd": (c_int, [C.POINTER(GruDirBwd), c_int, c_fp, c_i64, c_i64, c_int, c_int, c_int, c_fp, c_sz, c_fp]),
    "g2v_gru_cell_fwd": (c_int, [c_fp, c_int, c_fp, c_f, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_int, c_int, c_fp]),
    "g2v_gru_cell_bwd": (c_int, [c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_f, c_fp, c_fp, c_fp, c_fp, c_int, c_int, c_int, c_fp]),
    "g2v_gru_seq_prepare": (c_int, [c_fp, c_fp, c_int, c_int, c_int, c_fp, c_sz, c_fp, c_sz, c_fp]),
    "g2v_gru_seq_fwd_prepared": (c_int, [C.POINTER(GruDir), c_int, c_fp, c_i64, c_int, c_int, c_int, c_fp, c_sz, c_fp]),
    "g2v_gru_seq_bwd_prepared": (c_int, [C.POINTER(GruDirBwd), c_int, c_fp, c_i64, c_i64, c_int, c_int, c_int, c_fp, c_sz, c_fp]),
    "g2v_dec_rollout_prepare": (c_int, [C.POINTER(DecWeights), c_int, c_int, c_fp, c_sz, c_fp, c_sz, c_fp]),
    "g2v_bn_running_update": (c_int, [c_fp, c_fp, c_fp, c_int, c_int, c_int, c_fp]),
    "g2v_train_step_prepare": (c_int, [C.POINTER(DecWeights), c_int, c_int, c_fp, c_sz, c_fp, c_sz, c_fp, c_fp, c_int, c_int, c_fp,
                                       c_sz, c_fp]),
    "g2v_dec_rollout_fwd_prepared": (c_int, [c_fp, c_fp, C.POINTER(DecWeights), C.POINTER(DecSaved), c_fp, c_fp, c_f,
                                             c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_fp, c_sz, c_fp]),
    "g2v_dec_rollout_bwd_prepared": (c_int, [C.POINTER(DecWeights), C.POINTER(DecSaved), C.POINTER(DecGrads), c_fp, c_fp,
                                             c_f, c_int, c_int, c_int, c_int, c_int, c_int, c_fp, c_sz, c_fp]),
    "g2v_dec_rollout_blocks": (c_int, [c_int]),
    "g2v_dec_rollout_set_persistent": (c_int, [c_int]),
    "g2v_gru_seq_set_cluster": (c_int, [c_int]),
    "g2v_gru_seq_cluster_ok": (c_int, [c_int, c_int, c_int, c_int]),
    "g2v_cluster_exchange_preclear": (c_int, [c_int, c_int, c_int, c_int, c_int, c_int, c_fp, c_sz, c_fp]),
    "g2v_dec_rollout_tiles_per_workgroup": (c_int, [c_int, c_int, c_int]),
    "g2v_dec_rollout_cluster_ok": (c_int, [c_int, c_int, c_int]),
    "g2v_dec_rollout_persist_fault": (c_int, [c_int]),
    "g2v_dec_rollout_fault_flag": (c_int, [c_fp, c_int, c_fp]),
    "g2v_dec_rollout_fuses_loss": (c_int, [c_int, c_int, c_int, c_int]),
    "g2v_custom_loss_chase": (c_int, [c_fp, C.POINTER(DecSaved), c_fp, c_int, c_int, c_int, c_int, c_fp, c_sz, c_fp]),
    "g2v_dec_rollout_fwd_workspace": (c_sz, [c_int, c_int]),
    "g2v_dec_rollout_fwd": (c_int, [c_fp, c_fp, C.POINTER(DecWeights), C.POINTER(DecSaved), c_fp, c_fp, c_f,
                                    c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_fp, c_sz, c_fp]),
    "g2v_dec_rollout_bwd_workspace": (c_sz, [c_int, c_int]),
    "g2v_dec_rollout_bwd_fuses_wgrad": (c_int, [c_int, c_int, c_int]),
    "g2v_dec_rollout_bwd": (c_int, [C.POINTER(DecWeights), C.POINTER(DecSaved), C.POINTER(DecGrads), c_fp, c_fp,
                                    c_f, c_int, c_int, c_int, c_int, c_int, c_int, c_fp, c_sz, c_fp]),
    "g2v_custom_loss_blocks": (c_int, [c_int, c_int]),
    "g2v_custom_loss_fwd_bwd": (c_int, [c_fp, c_fp, c_fp, c_fp, c_fp, c_f, c_f, c_f, c_f, c_int, c_int, c_int, c_fp]),
    "g2v_vq_codebook_grad": (c_int, [c_fp, c_fp, c_fp, c_fp, c_int, c_int, c_int, c_fp]),
    "g2v_mse_blocks": (c_int, [c_i64]),
    "g2v_mse_fwd_bwd": (c_int, [c_fp, c_fp, c_fp, c_fp, c_fp, c_i64, c_f, c_fp]),
    "g2v_adam_blocks": (c_int, [c_i64]),
    "g2v_iteration_readback": (c_int, [c_fp, c_fp, c_fp, c_fp, c_fp]),
    "g2v_clip_adam_step_readback": (c_int, [c_fp, c_fp, c_fp, c_fp, c_i64, c_fp, c_fp, c_fp, c_f, c_f, c_f, c_f, c_f, c_f,
                                            c_fp, c_fp, c_fp, c_fp, c_fp]),
    "g2v_clip_adam_step": (c_int, [c_fp, c_fp, c_fp, c_fp, c_i64, c_fp, c_fp, c_fp, c_f, c_f, c_f, c_f, c_f, c_f, c_fp]),
    "g2v_linear_set_smallm_rows": (c_int, [c_int]),
    "g2v_vq_assign_bulk_workspace": (c_sz, [c_int, c_int, c_int]),
    "g2v_vq_assign_bulk": (c_int, [c_fp, c_fp, c_fp, c_fp, c_int, c_int, c_int, c_fp, c_sz, c_fp, c_fp]),
    "g2v_copy_segments": (c_int, [c_fp, c_fp, c_fp, c_int, c_fp]),
    "g2v_embedding_fwd": (c_int, [c_fp, c_fp, c_fp, c_f, c_fp, c_i64, c_i64, c_int, c_i64, c_fp]),
    "g2v_embedding_bwd_ws_bytes": (c_sz, [c_i64, c_int, c_i64]),
    "g2v_embedding_bwd": (c_int, [c_fp, c_fp, c_fp, c_f, c_fp, c_i64, c_int, c_i64, c_int, c_fp, c_sz, c_fp]),
    "g2v_batchnorm_fwd": (c_int, [c_fp, c_fp, c_fp, c_fp, c_fp, c_int, c_int, c_fp, c_fp, c_fp, c_int, c_int, c_fp]),
    "g2v_bn_running_update_invstd": (c_int, [c_fp, c_fp, c_i64, c_fp, c_fp, c_int, c_int, c_int, c_fp]),
    "g2v_batchnorm_bwd": (c_int, [c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_int, c_fp, c_fp, c_fp, c_int, c_int, c_fp]),
    "g2v_batchnorm_bwd_steps": (c_int, [c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_i64, c_int, c_fp, c_fp, c_fp, c_int, c_int, c_int, c_fp]),
    "g2v_one_hot_rows": (c_int, [c_fp, c_fp, c_i64, c_int, c_int, c_fp]),
    "g2v_cross_entropy_fwd_bwd": (c_int, [c_fp, c_i64, c_fp, c_fp, c_fp, c_fp, c_i64, c_int, c_int, c_f, c_fp]),
    "g2v_attn_step_fwd": (c_int, [c_fp, c_i64, c_int, c_fp, c_fp, c_fp, c_f, c_fp, c_i64, c_fp, c_fp, c_fp, c_fp, c_fp, c_int, c_int, c_int, c_fp]),
    "g2v_slab_sum": (c_int, [c_fp, c_int, c_i64, c_fp, c_int, c_fp]),
    "g2v_linear_fwd_dual": (c_int, [c_fp, c_i64, c_fp, c_fp, c_fp, c_i64, c_int, c_fp, c_fp, c_fp, c_i64, c_int, c_int, c_int, c_fp]),
    "g2v_argmax_rows": (c_int, [c_fp, c_i64, c_fp, c_int, c_int, c_fp]),
    "g2v_vq_soft_fwd": (c_int, [c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_int, c_int, c_int, c_fp]),
    "g2v_vq_soft_bwd": (c_int, [c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_int, c_int, c_fp]),
    "g2v_vq_soft_fused_ok": (c_int, [c_int, c_int, c_int]),
    "g2v_vq_soft_fused_blocks": (c_int, [c_int]),
    "g2v_vq_soft_fused_fwd": (c_int, [c_fp] * 16 + [c_f, c_int, c_int, c_int, c_fp]),
    "g2v_vq_soft_finish": (c_int, [c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_int, c_int, c_int, c_fp]),
    "g2v_vq_soft_fused_bwd": (c_int, [c_fp] * 16 + [c_f, c_int, c_int, c_int, c_fp]),
    "g2v_vq_soft_perplexity_workspace": (c_sz, [c_int, c_int]),
    "g2v_vq_soft_perplexity": (c_int, [c_fp, c_fp, c_int, c_int, c_fp, c_sz, c_fp]),
    "g2v_rowscale_combine": (c_int, [c_fp, c_fp, c_fp, c_fp, c_i64, c_int, c_fp]),
    "g2v_ste_f32": (c_int, [c_fp, c_fp, c_fp, c_i64, c_fp]),
    "g2v_attn_fwd": (c_int, [c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_i64, c_int, c_int, c_int, c_fp]),
    "g2v_attn_bwd_workspace": (c_sz, [c_int, c_int]),
    "g2v_attn_bwd": (c_int, [c_fp, c_i64, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_int, c_int, c_int, c_int,
                             c_fp, c_sz, c_fp]),
    "g2v_attn_code_rollout_ok": (c_int, [c_int, c_int, c_int, c_int, c_int, c_int]),
    "g2v_attn_code_rollout_cluster_ok": (c_int, [c_int, c_int, c_int, c_int, c_int]),
    "g2v_code_cluster_bptt_workspace": (c_sz, [c_int, c_int]),
    "g2v_code_cluster_bptt": (c_int, [c_fp, C.POINTER(CodeDecWeights), C.POINTER(CodeDecSaved), c_fp, c_f, c_fp, c_fp, c_fp, c_fp,
                                      c_fp, c_fp, c_int, c_int, c_int, c_fp, c_sz, c_fp]),
    "g2v_attn_code_rollout_blocks": (c_int, [c_int]),
    "g2v_attn_code_rollout_fwd_workspace": (c_sz, [c_int, c_int, c_int]),
    "g2v_attn_code_rollout_fwd": (c_int, [c_fp, c_fp, c_fp, c_fp, C.POINTER(CodeDecWeights), C.POINTER(CodeDecSaved), c_fp, c_fp, c_f,
                                          c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_fp, c_sz, c_fp]),
    "g2v_attn_code_rollout_bwd_workspace": (c_sz, [c_int, c_int, c_int, c_int, c_int, c_int]),
    "g2v_attn_code_rollout_bwd": (c_int, [c_fp, c_fp, c_fp, C.POINTER(CodeDecWeights), C.POINTER(CodeDecSaved),
                                          C.POINTER(CodeDecGrads), c_fp, c_fp, c_f, c_int, c_int, c_int, c_int, c_int, c_fp, c_sz,
                                          c_fp]),
    "g2v_probe_mfma_f32": (c_int, [c_fp, c_int, c_int, c_fp]),
    "g2v_probe_copy": (c_int, [c_fp, c_fp, c_i64, c_fp]),
    "g2v_keep_mask": (c_int, [c_fp, c_i64, c_f, c_u64, c_fp, c_fp]),
    "g2v_keep_mask_at": (c_int, [c_fp, c_i64, c_f, c_u64, c_fp, c_i64, c_fp]),
    "g2v_counter_add": (c_int, [c_fp, c_i64, c_fp]),
    "g2v_dropout_rows": (c_int, [c_fp, c_i64, c_int, c_i64, c_i64, c_f, c_f, c_u64, c_fp, c_i64, c_fp, c_i64, c_int, c_int, c_fp]),
    "g2v_fill_f32": (c_int, [c_fp, c_f, c_i64, c_fp]),
    "g2v_scale_f32": (c_int, [c_fp, c_fp, c_fp, c_i64, c_fp]),
    "g2v_mask_mul": (c_int, [c_fp, c_fp, c_fp, c_f, c_fp, c_i64, c_fp]),
    "g2v_mask_rows": (c_int, [c_fp, c_i64, c_int, c_i64, c_i64, c_fp, c_f, c_fp, c_i64, c_int, c_int, c_fp]),
    "g2v_transpose": (c_int, [c_fp, c_fp, c_int, c_int, c_fp]),
    "g2v_add_halves": (c_int, [c_fp, c_i64, c_fp, c_i64, c_fp, c_i64, c_i64, c_int, c_fp]),
}

EXPORTS = tuple(_SIGS.keys())

_lib = None


class G2VLibraryError(RuntimeError):
    pass


def load():
    """Load libg2v_hip.so once and attach the signatures.  Raises if the library is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise G2VLibraryError(
            f"{LIB_PATH} is missing: build it with `make -C gesture2vec_amd/csrc` "
            "(or `python -c 'import __graft_entry__ as g; g.build()'`).  There is no fallback path.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in _SIGS.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc: int, what: str = ""):
    if rc != 0:
        msg = load().g2v_last_error().decode()
        raise G2VLibraryError(f"g2v call failed ({rc}) {what}: {msg}")


OPT_PERSISTENT, OPT_GRU_CLUSTER, OPT_SMALLM_ROWS, OPT_GRU_RESIDENT_ROWS, OPT_GRU_RESIDENT_BWD = 1, 2, 3, 4, 5


class Context:
    """A caller-owned set of the library's three implementation switches (include/g2v.h: g2v_ctx).  `with ctx:` binds it to the
    calling thread for the duration of the block (re-entrant: the previous binding comes back); every library call inside reads
    ITS switches.  An engine owns one, so that two engines in one process do not share switches and a residency fault in one
    does not switch off the fast path of the other (round-5 verdict: "no hidden global state")."""

    def __init__(self):
        self._lib = load()
        self._h = self._lib.g2v_ctx_create()
        if not self._h:
            raise MemoryError("g2v_ctx_create failed")
        self._stack = []

    def set(self, option: int, value: int) -> int:
        return int(self._lib.g2v_ctx_set_option(self._h, int(option), int(value)))

    def get(self, option: int) -> int:
        return int(self._lib.g2v_ctx_get_option(self._h, int(option)))

    def __enter__(self):
        self._stack.append(self._lib.g2v_ctx_bind(self._h))
        return self

    def __exit__(self, *exc):
        self._lib.g2v_ctx_bind(self._stack.pop())
        return False

    def __del__(self):
        try:
            if self._h:
                self._lib.g2v_ctx_destroy(self._h)
                self._h = None
        except Exception:
            pass
