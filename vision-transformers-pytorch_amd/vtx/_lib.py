"""ctypes binding of libvtx.so (C ABI declared in include/vtx.h).

The library is the product: there is NO fallback.  If libvtx.so is missing or a call returns a
non-zero code this module raises -- it never routes to PyTorch ops or to the CPU oracle.
"""
import ctypes
import os
from ctypes import c_char_p, c_float, c_int, c_int64, c_size_t, c_uint, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("VTX_LIBVTX") or os.path.join(_HERE, "libvtx.so")   # (override: A/B of two builds on one box)

F32, BF16 = 0, 1
ABI_VERSION = 27


class VtxError(RuntimeError):
    pass


_I, _L, _F, _P, _Z = ctypes.c_int, ctypes.c_int64, ctypes.c_float, ctypes.c_void_p, ctypes.c_size_t


class LayerFwd(ctypes.Structure):
    """include/vtx.h VtxLayerFwd (field order = the header's; load() checks sizeof against the library)."""
    _fields_ = ([("dtype", _I), ("attn_kind", _I), ("M", _L)] +
                [(n, _I) for n in ("C", "ff", "nH", "L", "B", "rows_per_scale", "H", "W", "win", "shift")] + [("eps", _F)] +
                [(n, _P) for n in ("x", "ln1_w", "ln1_b", "ln2_w", "ln2_b", "wq", "wo", "w1", "w2", "bq", "bo", "b1", "b2",
                                   "rel_pos", "pos", "region", "s1", "s2", "ln1", "qkv", "o", "x1", "ln2", "z", "h", "y",
                                   "mean1", "rstd1", "mean2", "rstd2", "lse", "perm1", "perm2")] + [("Bk1", _I), ("Bk2", _I)])


class LayerBwd(ctypes.Structure):
    """include/vtx.h VtxLayerBwd."""
    _fields_ = ([("dtype", _I), ("attn_kind", _I), ("M", _L)] +
                [(n, _I) for n in ("C", "ff", "nH", "L", "B", "rows_per_scale", "H", "W", "win", "shift")] +
                [("scale_const", _F), ("accumulate", _I), ("inv_count", _I)] +
                [(n, _P) for n in ("dy", "x", "ln1", "qkv", "o", "x1", "ln2", "z", "h", "mean1", "rstd1", "mean2", "rstd2", "lse",
                                   "ln1_w", "ln2_w", "wq", "wo", "w1", "w2", "wqt", "wot", "w1t", "w2t", "rel_pos", "pos",
                                   "region", "inv_cells", "s1", "s2", "dz", "dln2", "dx1", "dout", "dqkv", "dln1", "dx",
                                   "ln1_ws", "ln2_ws", "attn_ws", "wgrad_ws")] +
                [(n, _Z) for n in ("ln_ws_bytes", "attn_ws_bytes", "wgrad_ws_bytes")] +
                [(n, _P) for n in ("dWq", "dbq", "dWo", "dbo", "dW1", "db1", "dW2", "db2", "dg1", "dbe1", "dg2", "dbe2", "drel",
                                   "perm1", "perm2")] + [("Bk1", _I), ("Bk2", _I), ("b1", _P)])


class SrLayerFwd(ctypes.Structure):
    """include/vtx.h VtxSrLayerFwd."""
    _fields_ = ([("dtype", _I), ("twins", _I), ("M", _L)] +
                [(n, _I) for n in ("C", "ff", "nH", "L", "B", "rows_per_scale", "H", "W", "r", "skip", "Lk", "splitk")] +
                [("eps", _F)] +
                [(n, _P) for n in ("x", "ln1_w", "ln1_b", "ln2_w", "ln2_b", "srn_w", "srn_b", "wq", "wkv", "wsr", "wsr_t", "wo",
                                   "w1", "w2", "bsr", "bo", "b1", "b2", "s1", "s2", "ln1", "q", "patches", "patches_t", "red32",
                                   "red", "kvin", "kv", "o", "x1", "ln2", "z", "h", "y", "mean1", "rstd1", "mean2", "rstd2",
                                   "means", "rstds", "lse", "splitk_ws")] + [("splitk_ws_bytes", _Z)])


class SrLayerBwd(ctypes.Structure):
    """include/vtx.h VtxSrLayerBwd."""
    _fields_ = ([("dtype", _I), ("twins", _I), ("M", _L)] +
                [(n, _I) for n in ("C", "ff", "nH", "L", "B", "rows_per_scale", "H", "W", "r", "skip", "Lk", "reserved")] +
                [("scale_const", _F)] +
                [(n, _P) for n in ("dy", "x", "ln1", "q", "patches", "red", "kvin", "kv", "o", "x1", "ln2", "z", "h", "mean1",
                                   "rstd1", "mean2", "rstd2", "means", "rstds", "lse", "ln1_w", "ln2_w", "srn_w", "wq", "wkv",
                                   "wsr", "wo", "w1", "w2", "wqt", "wkvt", "wsrt", "wot", "w1t", "w2t", "s1", "s2", "dz", "dln2",
                                   "dx1", "dout", "dq", "dkv", "dkvin", "dred", "dpatches", "dln1", "dx", "ln1_ws", "ln2_ws",
                                   "lns_ws", "attn_ws", "wgrad_ws", "wgrad2_ws")] +
                [(n, _Z) for n in ("ln_ws_bytes", "lns_ws_bytes", "attn_ws_bytes", "wgrad_ws_bytes", "wgrad2_ws_bytes")] +
                [(n, _P) for n in ("dWq", "dWkv", "dWsr", "dbsr", "dWo", "dbo", "dW1", "db1", "dW2", "db2", "dg1", "dbe1", "dg2",
                                   "dbe2", "dgs", "dbs", "b1")])


ATTN_WINDOW, ATTN_GLOBAL = 1, 2


class TimerRec(ctypes.Structure):
    """include/vtx.h VtxTimerRec."""
    _fields_ = [("tag", _I), ("n", _I), ("k", _I), ("flags", _I), ("rows", _L), ("ms", _F)]


_SIGNATURES = {
    "vtx_layer_fwd": (c_int, [c_void_p, c_void_p]),
    "vtx_layer_bwd": (c_int, [c_void_p, c_void_p, c_void_p]),
    "vtx_layer_desc_bytes": (c_int, [c_int]),
    "vtx_srlayer_fwd": (c_int, [c_void_p, c_void_p]),
    "vtx_srlayer_bwd": (c_int, [c_void_p, c_void_p, c_void_p]),
    "vtx_timer_start": (c_int, []),
    "vtx_timer_stop": (c_int, [c_void_p, c_int]),
    "vtx_layernorm_fwd_mapped": (c_int, [c_void_p] * 6 + [c_int64, c_int, c_float, c_int, c_void_p, c_int, c_void_p]),
    "vtx_layernorm_bwd_mapped": (c_int, [c_void_p] * 8 + [c_size_t, c_int64, c_int64, c_int, c_int, c_void_p, c_int, c_void_p]),
    "vtx_attention_fwd_mapped": (c_int, [c_void_p] * 4 + [c_int] * 5 + [c_void_p]),
    "vtx_attention_bwd_mapped": (c_int, [c_void_p] * 6 + [c_int] * 5 + [c_void_p]),
    "vtx_wattn_fwd_mapped": (c_int, [c_void_p] * 7 + [c_int] * 9 + [c_void_p]),
    "vtx_wattn_bwd_mapped": (c_int, [c_void_p] * 9 + [c_size_t, c_void_p, c_int, c_void_p] + [c_int] * 9 + [c_void_p]),
    "vtx_wgrad_group_mapped": (c_int, [c_int, c_int] + [c_void_p] * 12 + [c_int, c_float, c_int64, c_void_p, c_size_t, c_int] +
                               [c_void_p] * 6 + [c_int, c_void_p]),
    "vtx_strerror": (c_char_p, [c_int]),
    "vtx_abi_version": (c_int, []),
    "vtx_cu_count": (c_int, []),
    "vtx_debug_lds_poison": (c_int, [c_uint, c_int, c_void_p]),
    "vtx_debug_spin": (c_int, [c_int, c_void_p]),
    "vtx_sattn_waves": (c_int, [c_int]),
    "vtx_debug_mfma_peak": (c_int, [c_int, c_int, c_void_p, ctypes.POINTER(ctypes.c_double), c_void_p]),
    "vtx_option_count": (c_int, []),
    "vtx_option_name": (c_char_p, [c_int]),
    "vtx_get_option": (c_int, [c_int]),
    "vtx_set_option": (c_int, [c_int, c_int]),
    "vtx_layernorm_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int,
                                  c_float, c_int, c_int, c_int, c_int, c_void_p]),
    "vtx_layernorm_bwd_workspace": (c_size_t, [c_int64, c_int]),
    "vtx_layernorm_bwd_blocks": (c_int, [c_int64, c_int]),
    "vtx_colreduce_multi": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "vtx_layernorm_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                  c_void_p, c_void_p, c_size_t, c_int64, c_int, c_int, c_int, c_int, c_int,
                                  c_void_p]),
    "vtx_gemm": (c_int, [c_int, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int64, c_int64,
                         c_int64, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int, c_void_p]),
    "vtx_mlp_fused_ok": (c_int, [c_int, c_int64, c_int, c_int]),
    "vtx_mlp_fwd": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p,
                            c_void_p, c_int64, c_int, c_int, c_void_p]),
    "vtx_mlp_bwd": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p,
                            c_int64, c_int, c_int, c_void_p]),
    "vtx_mlp_bwd_ln": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p,
                       c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int64, c_int, c_int, c_void_p]),
    "vtx_dgrad_ln": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int,
                     c_int64, c_int, c_int, c_void_p]),
    "vtx_mlp_fwd_ln": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_float, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                       c_void_p, c_void_p, c_int, c_void_p, c_int64, c_int, c_int, c_void_p]),
    "vtx_ln_gemm": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_float, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                    c_int64, c_int, c_int, c_void_p]),
    "vtx_wgrad_workspace": (c_size_t, [c_int64, c_int, c_int]),
    "vtx_wgrad": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_int64, c_int64,
                          c_void_p, c_int, c_float, c_void_p, c_size_t, c_void_p]),
    "vtx_wgrad_group_max": (c_int, []),
    "vtx_wgrad_group_ok": (c_int, [c_int, c_int, c_void_p, c_void_p, c_int64, c_int, c_int, c_float]),
    "vtx_wgrad_group_workspace": (c_size_t, [c_int, c_void_p, c_void_p, c_int64]),
    "vtx_wgrad_group": (c_int, [c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                c_void_p, c_void_p, c_int, c_float, c_int64, c_void_p, c_size_t, c_int, c_void_p, c_void_p,
                                c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    "vtx_wgrad_group_slices": (c_int, [c_int, c_void_p, c_void_p, c_int64]),
    "vtx_relpos_bias": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "vtx_attention_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int,
                                  c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "vtx_attention_bwd_workspace": (c_size_t, [c_int, c_int, c_int, c_int, c_int, c_int, c_int]),
    "vtx_attention_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                  c_void_p, c_void_p, c_int, c_void_p, c_size_t, c_int, c_int, c_int, c_int, c_int,
                                  c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "vtx_attention_fwd_drop": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int,
                                       c_int, c_int, c_int, c_int, c_int, c_int, c_float, ctypes.c_uint64, c_void_p, c_void_p]),
    "vtx_attention_bwd_drop": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                       c_void_p, c_void_p, c_int, c_void_p, c_size_t, c_int, c_int, c_int, c_int, c_int,
                                       c_int, c_int, c_int, c_int, c_int, c_float, ctypes.c_uint64, c_void_p, c_void_p]),
    "vtx_attn_keep_mask": (c_int, [c_void_p, c_int64, c_int, c_int, c_float, ctypes.c_uint64, c_void_p]),
    "vtx_wattn_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int,
                              c_int, c_int, c_int, c_int, c_void_p]),
    "vtx_wattn_bwd_workspace": (c_size_t, [c_int, c_int, c_int, c_int, c_int]),
    "vtx_wattn_bwd_parts": (c_int, [c_int, c_int, c_int, c_int, c_int]),
    "vtx_wattn_bwd_part_ld": (c_int, [c_int]),
    "vtx_wattn_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                              c_void_p, c_size_t, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int,
                              c_void_p]),
    "vtx_srattn_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "vtx_srattn_scores": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "vtx_srattn_bwd_workspace": (c_size_t, [c_int, c_int, c_int, c_int, c_int]),
    "vtx_srattn_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t,
                               c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "vtx_srattn_fwd_drop": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_float,
                                    ctypes.c_uint64, c_void_p, c_void_p]),
    "vtx_srattn_bwd_drop": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t,
                                    c_int, c_int, c_int, c_int, c_int, c_int, c_float, ctypes.c_uint64, c_void_p, c_void_p]),
    "vtx_window_gather": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int64, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "vtx_window_scatter": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int64, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "vtx_table_bias": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_void_p]),
    "vtx_table_bias_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_void_p]),
    "vtx_xattn_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "vtx_xattn_bwd_workspace": (c_size_t, [c_int, c_int, c_int]),
    "vtx_xattn_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                              c_size_t, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "vtx_xattn_fwd_drop": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_float,
                                   ctypes.c_uint64, c_void_p, c_void_p]),
    "vtx_xattn_bwd_drop": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                   c_size_t, c_int, c_int, c_int, c_int, c_int, c_int, c_float, ctypes.c_uint64, c_void_p, c_void_p]),
    "vtx_dwconv3_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "vtx_dwconv3_wgrad_workspace": (c_size_t, [c_int, c_int, c_int, c_int]),
    "vtx_dwconv3_wgrad": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "vtx_twins_subsample_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "vtx_bias_cast": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_void_p]),
    "vtx_twins_subsample_bwd": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "vtx_patchify_fwd": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "vtx_patchify_bwd": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "vtx_add_pos_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "vtx_add_pos_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "vtx_opt_chunk": (c_int, []),
    "vtx_grad_sqnorm": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "vtx_adamw_step": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                               c_float, c_float, c_float, c_float, c_int, c_void_p]),
    "vtx_l2norm_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_float, c_int, c_void_p]),
    "vtx_l2norm_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_void_p]),
    "vtx_mix_plan_bytes": (c_size_t, []),
    "vtx_mix_max_rects": (c_int, []),
    "vtx_mix_normalize_erase": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int,
                                        c_int, c_int, c_int, c_void_p]),
    "vtx_ema_update": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_float, c_void_p]),
    "vtx_dino_loss_workspace": (c_size_t, [c_int, c_int]),
    "vtx_dino_loss": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p, c_void_p, c_void_p, c_int,
                              c_int, c_int, c_float, c_float, c_float, c_int, c_void_p]),
    "vtx_mix_loss": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_float, c_float,
                             c_int, c_void_p]),
    "vtx_cast_desc_bytes": (c_size_t, []),
    "vtx_cast_weights": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "vtx_patch_gather": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int,
                                 c_void_p]),
    "vtx_patch_gather_nhwc": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int,
                                      c_void_p]),
    "vtx_token_mean_fwd": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "vtx_token_mean_bwd": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "vtx_vit_assemble_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "vtx_vit_assemble_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
}

_lib = None


def load():
    """Load libvtx.so once; raise VtxError (never fall back) if it is absent or stale."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise VtxError(
            f"{LIB_PATH} not found: the HIP extension is the product path and has no fallback. "
            "Build it with `python -c 'import __graft_entry__ as g; g.build()'`.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in _SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise VtxError(f"libvtx.so does not export {name} (stale build?)") from e
        fn.restype = res
        fn.argtypes = args
    if lib.vtx_abi_version() != ABI_VERSION:
        raise VtxError(f"libvtx.so ABI {lib.vtx_abi_version()} != binding ABI {ABI_VERSION}: rebuild")
    if lib.vtx_layer_desc_bytes(0) != ctypes.sizeof(LayerFwd) or lib.vtx_layer_desc_bytes(1) != ctypes.sizeof(LayerBwd):
        raise VtxError("libvtx.so layer descriptors do not match the bindings (VtxLayerFwd / VtxLayerBwd): rebuild")
    if lib.vtx_layer_desc_bytes(2) != ctypes.sizeof(SrLayerFwd) or lib.vtx_layer_desc_bytes(3) != ctypes.sizeof(SrLayerBwd):
        raise VtxError("libvtx.so layer descriptors do not match the bindings (VtxSrLayerFwd / VtxSrLayerBwd): rebuild")
    _lib = lib
    return lib


def exported_symbols():
    return sorted(_SIGNATURES)


def check(code, what):
    if code != 0:
        msg = load().vtx_strerror(code).decode()
        raise VtxError(f"{what} failed: {msg} (code {code})")
