"""ctypes binding of libmatpbr.so (include/matpbr.h).  There is NO fallback: if the HIP library is
missing or a call fails, the product path raises."""
from __future__ import annotations

import ctypes
import os
import threading

from . import build as _build

_c_f = ctypes.c_void_p  # device pointers travel as opaque addresses


class MatpbrCamera(ctypes.Structure):
    _fields_ = [("fov_x_deg", ctypes.c_float)]


class MatpbrBrdfPhase(ctypes.Structure):
    """Mirror of `MatpbrBrdfPhase` in include/matpbr.h (field order matters)."""
    _fields_ = ([(k, ctypes.c_void_p) for k in ("pa", "pr", "pm", "n", "light", "gt_srgb", "a0", "r0", "m0", "dcache", "pred", "jac", "d_a",
                                                "d_r", "d_m")] +
                [("adam_m", ctypes.c_void_p * 3), ("adam_v", ctypes.c_void_p * 3)] +
                [(k, ctypes.c_void_p) for k in ("best_a", "best_r", "best_m", "best_img", "stats", "history", "workspace")] +
                [("workspace_bytes", ctypes.c_size_t), ("H", ctypes.c_int), ("W", ctypes.c_int), ("batch", ctypes.c_int), ("spp", ctypes.c_int),
                 ("fov_x_deg", ctypes.c_float), ("scale_delta", ctypes.c_float), ("part_mask", ctypes.c_uint32), ("es_patience", ctypes.c_int),
                 ("es_min_delta", ctypes.c_float), ("hist_len", ctypes.c_int), ("s1cache", ctypes.c_void_p), ("lazy_state", ctypes.c_void_p),
                 ("lazy_tol", ctypes.c_float), ("pred_next", ctypes.c_void_p), ("flags", ctypes.c_uint32), ("lazy_fold", ctypes.c_void_p)])


class MatpbrNormalStep(ctypes.Structure):
    """Mirror of `MatpbrNormalStep` in include/matpbr.h (field order matters)."""
    _fields_ = ([(k, ctypes.c_void_p) for k in ("pa", "pr", "pm", "pn", "ca", "cr", "cm", "cn", "d_a", "d_r", "d_m", "d_n", "a0", "r0", "m0", "n0")] +
                [("adam_m", ctypes.c_void_p * 4), ("adam_v", ctypes.c_void_p * 4)] +
                [(k, ctypes.c_void_p) for k in ("best_a", "best_r", "best_m", "best_n", "best_img", "pred", "stats", "ln_part")] +
                [("H", ctypes.c_int), ("W", ctypes.c_int), ("batch", ctypes.c_int), ("part_mask", ctypes.c_uint32), ("scale_delta", ctypes.c_float)])


class MatpbrError(RuntimeError):
    pass


# name -> (restype, argtypes); must list every symbol include/matpbr.h declares
class ReduceJob(ctypes.Structure):
    """include/matpbr.h `MatpbrReduceJob`: a deferred fold of per-workgroup partial sums (matpbr_mlp_reduce_jobs)."""
    _fields_ = [("kind", ctypes.c_int), ("groups", ctypes.c_int), ("src", ctypes.c_void_p), ("src_b", ctypes.c_void_p), ("src_g", ctypes.c_void_p),
                ("dst", ctypes.c_void_p), ("dst_b", ctypes.c_void_p), ("dst_g", ctypes.c_void_p), ("n0", ctypes.c_int), ("n1", ctypes.c_int),
                ("n2", ctypes.c_int), ("n3", ctypes.c_int), ("ld_j", ctypes.c_long), ("ld_c", ctypes.c_long)]


SIGNATURES = {
    "matpbr_version": (ctypes.c_int, []),
    "matpbr_strerror": (ctypes.c_char_p, [ctypes.c_int]),
    "matpbr_shade_fwd": (ctypes.c_int, [_c_f] * 5 + [ctypes.c_int, ctypes.c_int, _c_f, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                       ctypes.c_int, ctypes.POINTER(MatpbrCamera), ctypes.c_uint32, ctypes.c_void_p]),
    "matpbr_shade_bwd": (ctypes.c_int, [_c_f] * 5 + [ctypes.c_int, ctypes.c_int] + [_c_f] * 6 + [ctypes.c_void_p, ctypes.c_size_t,
                                       ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.POINTER(MatpbrCamera),
                                       ctypes.c_uint32, ctypes.c_void_p]),
    "matpbr_shade_bwd_workspace_bytes": (ctypes.c_size_t, [ctypes.c_int] * 4),
    "matpbr_plane9_bytes": (ctypes.c_size_t, [ctypes.c_int] * 3),
    "matpbr_shade_fwd_ex": (ctypes.c_int, [_c_f] * 5 + [ctypes.c_int, ctypes.c_int, _c_f, _c_f, _c_f, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                          ctypes.c_int, ctypes.POINTER(MatpbrCamera), ctypes.c_uint32, ctypes.c_void_p]),
    "matpbr_shade_fwd_keep": (ctypes.c_int, [_c_f] * 5 + [ctypes.c_int, ctypes.c_int, _c_f, _c_f, _c_f, _c_f, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                            ctypes.c_int, ctypes.POINTER(MatpbrCamera), ctypes.c_uint32, ctypes.c_void_p]),
    "matpbr_shade_fwd_cached": (ctypes.c_int, [_c_f] * 5 + [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_uint32, ctypes.c_void_p]),
    "matpbr_diffuse_cache": (ctypes.c_int, [_c_f, _c_f, ctypes.c_int, ctypes.c_int, _c_f, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                           ctypes.POINTER(MatpbrCamera), ctypes.c_void_p]),
    "matpbr_lazy_state_bytes": (ctypes.c_size_t, [ctypes.c_int] * 3),
    "matpbr_lazy_fold_bytes": (ctypes.c_size_t, [ctypes.c_int] * 3),
    "matpbr_env_texel_phase_step": (ctypes.c_int, [_c_f] * 6 + [ctypes.c_int, ctypes.c_int, ctypes.c_float, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int,
                                                   ctypes.c_int, _c_f, ctypes.c_int] + [_c_f] * 8 + [ctypes.c_float] * 3 + [ctypes.c_int, ctypes.c_int,
                                                                                                                          ctypes.c_void_p]),
    "matpbr_brdf_loss_dpred": (ctypes.c_int, [_c_f] * 4 + [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]),
    "matpbr_brdf_normal_step": (ctypes.c_int, [ctypes.POINTER(MatpbrNormalStep), ctypes.c_int, ctypes.c_float, ctypes.c_void_p]),
    "matpbr_env_mlp_phase_step": (ctypes.c_int, [_c_f] * 7 + [ctypes.c_int, ctypes.c_int, ctypes.c_float, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int,
                                                 ctypes.c_int, _c_f, ctypes.c_int] + [_c_f] * 4 + [ctypes.c_int, ctypes.c_int, ctypes.c_void_p]),
    "matpbr_brdf_phase_stages_timed": (ctypes.c_int, [ctypes.POINTER(MatpbrBrdfPhase), ctypes.c_int, ctypes.c_float, ctypes.c_uint32, ctypes.c_void_p,
                                                      ctypes.c_void_p, ctypes.c_void_p]),
    "matpbr_lazy_sums_count": (ctypes.c_int, [ctypes.c_int] * 2),
    "matpbr_shade_fwd_lazy": (ctypes.c_int, [_c_f] * 5 + [ctypes.c_int, ctypes.c_int] + [_c_f] * 6 + [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                            ctypes.POINTER(MatpbrCamera), ctypes.c_uint32, ctypes.c_float, ctypes.c_float, ctypes.c_void_p]),
    "matpbr_lazy_state_unpack": (ctypes.c_int, [_c_f, _c_f, _c_f, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]),
    "matpbr_jac16_unpack": (ctypes.c_int, [_c_f, _c_f, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]),
    "matpbr_shade_bwd_jac": (ctypes.c_int, [_c_f] * 8 + [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]),
    "matpbr_brdf_loss_workspace_bytes": (ctypes.c_size_t, [ctypes.c_int]),
    "matpbr_brdf_loss_stats": (ctypes.c_int, [_c_f] * 9 + [ctypes.c_float, _c_f, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_int,
                                             ctypes.c_int, ctypes.c_uint32, ctypes.c_void_p]),
    "matpbr_brdf_loss_stats_es": (ctypes.c_int, [_c_f] * 9 + [ctypes.c_float, _c_f, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_int,
                                                ctypes.c_int, ctypes.c_uint32, ctypes.c_int, ctypes.c_float, _c_f, ctypes.c_int, ctypes.c_void_p]),
    "matpbr_brdf_loss_bwd_jac": (ctypes.c_int, [_c_f] * 10 + [ctypes.c_float] + [_c_f] * 7 + [ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                                                                              ctypes.c_uint32, ctypes.c_void_p]),
    "matpbr_brdf_phase_workspace_bytes": (ctypes.c_size_t, [ctypes.c_int] * 3),
    "matpbr_brdf_phase_step": (ctypes.c_int, [ctypes.POINTER(MatpbrBrdfPhase), ctypes.c_int, ctypes.c_float, ctypes.c_void_p]),
    "matpbr_brdf_phase_stages": (ctypes.c_int, [ctypes.POINTER(MatpbrBrdfPhase), ctypes.c_int, ctypes.c_float, ctypes.c_uint32, ctypes.c_void_p]),
    "matpbr_brdf_phase_resolve": (ctypes.c_int, [ctypes.POINTER(MatpbrBrdfPhase), ctypes.c_int, ctypes.c_void_p]),
    "matpbr_env_phase_workspace_bytes": (ctypes.c_size_t, [ctypes.c_int] * 3),
    "matpbr_env_phase_step": (ctypes.c_int, [_c_f] * 7 + [ctypes.c_int, ctypes.c_int, ctypes.c_float, ctypes.c_void_p, ctypes.c_size_t,
                                            ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]),
    "matpbr_env_project": (ctypes.c_int, [_c_f, ctypes.c_int, _c_f, _c_f, _c_f, ctypes.c_int, ctypes.c_void_p]),
    "matpbr_env_project_bwd": (ctypes.c_int, [_c_f, ctypes.c_int, _c_f, _c_f, _c_f, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]),
    "matpbr_select_improved": (ctypes.c_int, [_c_f, _c_f, _c_f, ctypes.c_int, ctypes.c_long, ctypes.c_void_p]),
    "matpbr_mlp_wsplit_bytes": (ctypes.c_size_t, [ctypes.c_int]),
    "matpbr_mlp_split_weights": (ctypes.c_int, [_c_f, ctypes.c_int, ctypes.c_int, ctypes.c_int, _c_f, ctypes.c_void_p]),
    "matpbr_mlp_layer_fwd_bx": (ctypes.c_int, [_c_f, ctypes.c_int, _c_f, _c_f, _c_f, _c_f, ctypes.c_int, ctypes.c_long, ctypes.c_int, ctypes.c_int,
                                              ctypes.c_int, ctypes.c_void_p]),
    "matpbr_mlp_layer_bwd_input_bx": (ctypes.c_int, [_c_f, ctypes.c_int, _c_f, _c_f, _c_f, ctypes.c_int, _c_f, ctypes.c_void_p, ctypes.c_size_t,
                                                    ctypes.c_long, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]),
    "matpbr_sin_bwd": (ctypes.c_int, [_c_f, ctypes.c_long, _c_f, ctypes.c_long, _c_f, ctypes.c_long, ctypes.c_int, ctypes.c_void_p]),
    "matpbr_mlp_layer_fwd": (ctypes.c_int, [_c_f, ctypes.c_int, _c_f, ctypes.c_int, _c_f, _c_f, _c_f, ctypes.c_int, ctypes.c_long, ctypes.c_int,
                                           ctypes.c_int, ctypes.c_void_p]),
    "matpbr_mlp_bwd_input_workspace_bytes": (ctypes.c_size_t, [ctypes.c_long]),
    "matpbr_mlp_layer_bwd_input": (ctypes.c_int, [_c_f, ctypes.c_int, _c_f, ctypes.c_int, _c_f, _c_f, ctypes.c_int, _c_f, ctypes.c_void_p,
                                                 ctypes.c_size_t, ctypes.c_long, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]),
    "matpbr_mlp_bwd_weight_workspace_bytes": (ctypes.c_size_t, [ctypes.c_long]),
    "matpbr_mlp_layer_bwd_weight": (ctypes.c_int, [_c_f, ctypes.c_int, _c_f, ctypes.c_int, _c_f, ctypes.c_int, ctypes.c_void_p, ctypes.c_size_t,
                                                  ctypes.c_long, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]),
    "matpbr_mlp_layer_fwd_sgn": (ctypes.c_int, [_c_f, ctypes.c_int, _c_f, ctypes.c_int, _c_f, _c_f, ctypes.c_int, _c_f, ctypes.c_int, ctypes.c_long,
                                               ctypes.c_int, ctypes.c_int, ctypes.c_void_p]),
    "matpbr_mlp_layer_bwd_input_sgn": (ctypes.c_int, [_c_f, ctypes.c_int, _c_f, ctypes.c_int, _c_f, _c_f, ctypes.c_int, _c_f, ctypes.c_void_p,
                                                     ctypes.c_size_t, ctypes.c_long, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]),
    "matpbr_mlp_layer_bwd_input_bx_sgn": (ctypes.c_int, [_c_f, ctypes.c_int, _c_f, _c_f, _c_f, ctypes.c_int, _c_f, ctypes.c_void_p, ctypes.c_size_t,
                                                        ctypes.c_long, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]),
    "matpbr_mlp_layer_bwd_weight_bx": (ctypes.c_int, [_c_f, ctypes.c_int, _c_f, ctypes.c_int, _c_f, ctypes.c_int, ctypes.c_void_p, ctypes.c_size_t,
                                                     ctypes.c_long, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]),
    "matpbr_mlp_out_layer_bwd_tmax": (ctypes.c_int, [_c_f, ctypes.c_int, _c_f, _c_f, ctypes.c_int, _c_f, ctypes.c_int, _c_f, ctypes.c_int, ctypes.c_void_p, _c_f,
                                                     ctypes.c_long, ctypes.c_long, _c_f, _c_f, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_long, ctypes.c_int,
                                                     ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]),
    "matpbr_mlp_reduce_jobs": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]),
    "matpbr_mlp_layer_bwd_input_blk": (ctypes.c_int, [_c_f, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, _c_f, _c_f, ctypes.c_int, ctypes.c_void_p, _c_f,
                                                      ctypes.c_void_p, ctypes.c_size_t, ctypes.c_long, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]),
    "matpbr_mlp_first_layer_bwd_blk": (ctypes.c_int, [_c_f, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, _c_f, ctypes.c_int, _c_f, ctypes.c_int, _c_f,
                                                      ctypes.c_long, ctypes.c_long, ctypes.c_int, _c_f, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p,
                                                      ctypes.c_size_t, ctypes.c_long, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]),
    "matpbr_mlp_layer_bwd_weight_blk": (ctypes.c_int, [_c_f, ctypes.c_int, ctypes.c_void_p, _c_f, ctypes.c_int, _c_f, ctypes.c_int, ctypes.c_void_p,
                                                       ctypes.c_size_t, ctypes.c_long, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]),
    "matpbr_mlp_chain_images_bytes": (ctypes.c_size_t, []),
    "matpbr_mlp_chain_prep": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p,
                                             ctypes.c_void_p, ctypes.c_long, ctypes.c_void_p]),
    "matpbr_mlp_chain_fwd": (ctypes.c_int, [_c_f, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, _c_f, ctypes.c_int, _c_f, _c_f, _c_f,
                                            _c_f, ctypes.c_int, ctypes.c_long, ctypes.c_void_p]),
    "matpbr_mlp_split_weights_fmt": (ctypes.c_int, [_c_f, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]),
    "matpbr_mlp_skinny_fwd": (ctypes.c_int, [_c_f, ctypes.c_int, _c_f, ctypes.c_int, _c_f, _c_f, ctypes.c_int, ctypes.c_long, ctypes.c_int, ctypes.c_int,
                                            ctypes.c_void_p]),
    "matpbr_mlp_arm_head_fwd": (ctypes.c_int, [_c_f, ctypes.c_int, _c_f, ctypes.c_int, _c_f, _c_f, ctypes.c_int, _c_f, _c_f, _c_f, _c_f, ctypes.c_long,
                                              ctypes.c_int, ctypes.c_void_p]),
    "matpbr_mlp_arm_head_bwd": (ctypes.c_int, [_c_f, _c_f, _c_f, _c_f, _c_f, ctypes.c_long, ctypes.c_void_p]),
    "matpbr_mlp_skinny_workspace_bytes": (ctypes.c_size_t, [ctypes.c_int]),
    "matpbr_mlp_set_lds_dma": (ctypes.c_int, [ctypes.c_int]),
    "matpbr_mlp_skinny_bwd_weight": (ctypes.c_int, [_c_f, ctypes.c_int, _c_f, ctypes.c_int, _c_f, ctypes.c_long, ctypes.c_long, _c_f, ctypes.c_void_p,
                                                   ctypes.c_size_t, ctypes.c_long, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]),
    "matpbr_adamw_step_dev": (ctypes.c_int, [_c_f] * 4 + [ctypes.c_long, _c_f, ctypes.c_float, ctypes.c_float, ctypes.c_float, ctypes.c_float,
                                                         ctypes.c_void_p]),
    "matpbr_light_to_sh25": (ctypes.c_int, [_c_f, ctypes.c_int, ctypes.c_int, _c_f, ctypes.c_int, ctypes.c_void_p]),
    "matpbr_light_to_sh25_bwd": (ctypes.c_int, [_c_f, ctypes.c_int, ctypes.c_int, _c_f, ctypes.c_int, ctypes.c_void_p]),
    "matpbr_mlp_split_weights_multi": (ctypes.c_int, [ctypes.c_void_p] * 6 + [ctypes.c_int, ctypes.c_void_p]),
    "matpbr_adamw_step_snapshot_dev": (ctypes.c_int, [_c_f] * 4 + [ctypes.c_long, _c_f, ctypes.c_float, ctypes.c_float, ctypes.c_float, ctypes.c_float,
                                                                  _c_f, _c_f, ctypes.c_void_p]),
    "matpbr_mlp_layer_fwd_tail": (ctypes.c_int, [_c_f, ctypes.c_int, _c_f, ctypes.c_int, _c_f, _c_f, _c_f, ctypes.c_int, _c_f, ctypes.c_int, ctypes.c_long,
                                                ctypes.c_int, ctypes.c_int, ctypes.c_void_p]),
    "matpbr_mlp_layer_fwd_bx_tail": (ctypes.c_int, [_c_f, ctypes.c_int, _c_f, _c_f, _c_f, _c_f, ctypes.c_int, _c_f, ctypes.c_int, ctypes.c_long,
                                                   ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]),
    "matpbr_sample_brdf_dr": (ctypes.c_int, [_c_f] * 10 + [ctypes.c_long, ctypes.c_void_p]),
    "matpbr_mlp_layer_fwd_bx_head": (ctypes.c_int, [_c_f, ctypes.c_int, _c_f, _c_f, _c_f, _c_f, ctypes.c_int, _c_f, ctypes.c_int, _c_f, _c_f, ctypes.c_int,
                                                   _c_f, _c_f, _c_f, _c_f, ctypes.c_long, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]),
    "matpbr_column_sum_workspace_bytes": (ctypes.c_size_t, [ctypes.c_int]),
    "matpbr_column_sum": (ctypes.c_int, [_c_f, _c_f, ctypes.c_long, ctypes.c_int, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]),
    "matpbr_shade_transfer": (ctypes.c_int, [_c_f] * 5 + [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.POINTER(MatpbrCamera),
                                            ctypes.c_uint32, ctypes.c_void_p]),
    "matpbr_transfer_bytes": (ctypes.c_size_t, [ctypes.c_int] * 3),
    "matpbr_relight": (ctypes.c_int, [_c_f] * 3 + [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]),
    "matpbr_adam_step": (ctypes.c_int, [_c_f] * 4 + [ctypes.c_long, ctypes.c_float, ctypes.c_float, ctypes.c_float, ctypes.c_float, ctypes.c_int,
                                       ctypes.c_void_p]),
    "matpbr_eval_brdf": (ctypes.c_int, [_c_f] * 8 + [ctypes.c_long, ctypes.c_void_p]),
    "matpbr_eval_brdf_bwd": (ctypes.c_int, [_c_f] * 11 + [ctypes.c_long, ctypes.c_void_p]),
    "matpbr_brdf_terms": (ctypes.c_int, [_c_f] * 5 + [ctypes.c_long, ctypes.c_void_p]),
    "matpbr_sample_brdf": (ctypes.c_int, [_c_f] * 10 + [ctypes.c_long, ctypes.c_void_p]),
    "matpbr_sh_eval": (ctypes.c_int, [_c_f] * 3 + [ctypes.c_long, ctypes.c_void_p]),
    "matpbr_depth_to_mesh_host": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_float, ctypes.c_float] + [ctypes.c_void_p] * 6),
    "matpbr_mlp_small_bwd_step": (ctypes.c_int, [_c_f, ctypes.c_int, _c_f, ctypes.c_int, _c_f, _c_f, ctypes.c_int, _c_f, ctypes.c_int, _c_f, ctypes.c_int,
                                                 _c_f, ctypes.c_int, ctypes.c_int, _c_f, ctypes.c_int, ctypes.c_int, _c_f, ctypes.c_long, ctypes.c_int,
                                                 ctypes.c_void_p]),
    "matpbr_mlp_out_layer_bwd": (ctypes.c_int, [_c_f, ctypes.c_int, _c_f, _c_f, ctypes.c_int, _c_f, ctypes.c_int, _c_f, ctypes.c_int, _c_f, ctypes.c_long,
                                                ctypes.c_long, _c_f, _c_f, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_long, ctypes.c_int, ctypes.c_int,
                                                ctypes.c_void_p]),
    "matpbr_mlp_first_layer_bwd_bx": (ctypes.c_int, [_c_f, ctypes.c_int, ctypes.c_void_p, _c_f, ctypes.c_int, ctypes.c_int, _c_f, ctypes.c_int, _c_f, ctypes.c_long,
                                                     ctypes.c_long, ctypes.c_int, _c_f, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t,
                                                     ctypes.c_long, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]),
    "matpbr_masked_mean_fill": (ctypes.c_int, [_c_f, ctypes.c_void_p, _c_f, ctypes.c_float, ctypes.c_float, _c_f, ctypes.c_long, ctypes.c_int,
                                               ctypes.c_void_p]),
    "matpbr_normals_from_depth": (ctypes.c_int, [_c_f, _c_f, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.POINTER(MatpbrCamera),
                                                 ctypes.c_void_p]),
}

_lock = threading.Lock()
_lib = None


def library_path() -> str:
    """The product library, or the file MATPBR_LIB names (an instrumented / experimental build beside it: tools/ never overwrite the product)."""
    return os.environ.get("MATPBR_LIB") or _build.LIB_PATH


def load(build_if_missing: bool = False) -> ctypes.CDLL:
    """Load libmatpbr.so and bind every declared symbol.  Raises MatpbrError when it cannot."""
    global _lib
    with _lock:
        if _lib is not None:
            return _lib
        path = library_path()
        if not os.path.exists(path):
            if build_if_missing:
                _build.build_library()
            else:
                raise MatpbrError(
                    f"{path} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                    "(hipcc --offload-arch=gfx950). matpbr has no CPU or PyTorch fallback.")
        try:
            lib = ctypes.CDLL(path)
        except OSError as e:  # pragma: no cover - depends on the host
            raise MatpbrError(f"cannot load {path}: {e}") from e
        for name, (res, args) in SIGNATURES.items():
            try:
                fn = getattr(lib, name)
            except AttributeError as e:
                raise MatpbrError(f"{path} does not export {name}") from e
            fn.restype = res
            fn.argtypes = args
        _lib = lib
        return lib


def check(code: int, what: str) -> None:
    if code != 0:
        msg = load().matpbr_strerror(code)
        raise MatpbrError(f"{what} failed: {msg.decode() if msg else code} ({code})")
