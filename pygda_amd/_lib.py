"""ctypes binding of libgda_hip.so (the C ABI declared in include/gda_hip.h).

The product path has no fallback: if the library is missing or a call fails this module
raises.  Tensors cross the boundary as raw device pointers (``tensor.data_ptr()``), the
stream as ``torch.cuda.current_stream().cuda_stream``.
"""
import ctypes
import os
from ctypes import c_char_p, c_float, c_int, c_int64, c_size_t, c_void_p

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libgda_hip.so")

_P = c_void_p
_SIGNATURES = {
    "gda_abi_version": (c_int, []),
    "gda_status_string": (c_char_p, [c_int]),
    "gda_graph_workspace_bytes": (c_size_t, [c_int64, c_int64]),
    "gda_build_csr_norm": (c_int, [_P, _P, _P, c_int64, c_int64, c_float, c_int, c_int, c_int,
                                   _P, _P, _P, _P, _P, _P, _P, c_size_t, _P]),
    "gda_build_csr_norm_map": (c_int, [_P, _P, _P, c_int64, c_int64, c_float, c_int, c_int, c_int,
                                       _P, _P, _P, _P, _P, _P, _P, _P, c_size_t, _P]),
    "gda_gat_fwd_f32": (c_int, [_P, _P, c_int64, c_int64, _P, _P, _P, c_float, _P, _P, _P]),
    "gda_gat_bwd_f32": (c_int, [_P, _P, _P, _P, _P, c_int64, c_int64, _P, _P, _P, c_float, _P, _P,
                                _P, _P, _P, _P, _P]),
    "gda_csr_to_coo": (c_int, [_P, _P, c_int64, c_int64, _P, _P, _P]),
    "gda_spmm_csr_f32": (c_int, [_P, _P, _P, c_int64, c_int64, _P, c_int64, _P, c_int64, _P, _P]),
    "gda_spmm_csr_tout_f32": (c_int, [_P, _P, _P, c_int64, c_int64, _P, c_int64, _P, c_int64, _P, _P]),
    "gda_spmm_csr_kstep_f32": (c_int, [_P, _P, _P, c_int64, c_int64, c_int, _P, c_int64, _P, c_int64,
                                       _P, _P, _P]),
    "gda_spmm_csr_interior_kstep_f32": (c_int, [_P, _P, _P, c_int64, c_int64, c_int64, c_int, c_int, _P, _P, _P, _P, _P, _P]),
    "gda_interior_max_rows": (c_int, []),
    "gda_interior_max_width": (c_int, []),
    "gda_interior_plan_bytes": (c_size_t, []),
    "gda_interior_kstep_lds_workspace_bytes": (c_size_t, [c_int64, c_int64]),
    "gda_interior_plan_build": (c_int, [_P, _P, _P, _P, _P, c_size_t, _P, _P]),
    "gda_interior_kstep_lds_f32": (c_int, [_P, _P, _P, c_int64, c_int64, c_int64, c_int, c_int, _P, _P, _P, _P, _P,
                                           c_size_t, _P]),
    "gda_interior_kstep_lds_act_f32": (c_int, [_P, _P, _P, c_int64, c_int64, c_int64, c_int, _P, _P, _P, _P, _P,
                                               ctypes.c_float, ctypes.c_uint64, _P, ctypes.c_uint32, ctypes.c_uint32,
                                               _P, c_size_t, _P]),
    "gda_row_split_workspace_bytes": (c_size_t, [c_int64]),
    "gda_row_split_build": (c_int, [_P, c_int64, ctypes.c_int32, _P, _P, _P, _P, _P, c_size_t, _P]),
    "gda_spmm_csr_split_f32": (c_int, [_P, _P, _P, c_int64, c_int64, c_int, _P, c_int64, _P, c_int64,
                                       _P, _P, _P, _P]),
    "gda_mmd_workspace_bytes": (c_size_t, [c_int, c_int64, c_int64]),
    "gda_mmd_fwd_f32": (c_int, [_P, c_int64, _P, c_int64, c_int64, _P, _P, c_int, c_int64, c_float,
                                c_int, c_float, _P, _P, _P, _P, c_size_t, _P]),
    "gda_mmd_bwd_f32": (c_int, [_P, c_int64, _P, c_int64, c_int64, _P, _P, c_int, c_int64, c_float,
                                c_int, _P, _P, _P, _P, _P, c_size_t, _P]),
    "gda_mmd_fwd_ex_f32": (c_int, [_P, c_int64, _P, c_int64, c_int64, _P, _P, c_int, c_int64, c_float, c_int, c_float,
                                   c_float, _P, _P, _P, _P, _P, c_size_t, _P]),
    "gda_mmd_fwd_gather_f32": (c_int, [_P, c_int64, _P, c_int64, c_int64, _P, _P, c_int, c_int64, c_float, c_int, c_float,
                                       c_float, _P, _P, _P, _P, _P, _P, _P, c_size_t, _P]),
    "gda_mmd_bwd_ex_f32": (c_int, [_P, c_int64, _P, c_int64, c_int64, _P, _P, c_int, c_int64, c_float, c_int,
                                   _P, _P, _P, c_float, _P, _P, _P, c_int64, _P, _P, _P, c_int64, _P,
                                   _P, c_size_t, _P]),
    "gda_mmd_fused_nseg": (c_int, [c_int, c_int64, c_int64, c_float, c_int]),
    "gda_mmd_fused_layout": (c_int, [c_int, c_int64, c_int64, _P, c_int]),
    "gda_mmd_fused_fwd_f32": (c_int, [_P, c_int64, _P, c_int64, c_int64, _P, _P, c_int, c_int64, c_float, c_int, c_float,
                                      c_float, _P, _P, _P, _P, _P, _P, c_int, _P, c_size_t, _P]),
    "gda_mmd_fused_bwd_f32": (c_int, [_P, c_int, c_int, c_int64, c_int64, _P, c_float, _P,
                                      _P, _P, c_int64, _P, _P, _P, c_int64, _P, _P]),
    "gda_mmd_fused_bwd_mask_f32": (c_int, [_P, c_int, c_int, c_int64, c_int64, _P, c_float, _P,
                                           _P, _P, c_int64, _P, _P, _P, c_int64, _P, _P, c_float, _P, c_float, _P]),
    "gda_copy_from_pinned": (c_int, [_P, _P, c_size_t, _P]),
    "gda_softmax_entropy_fwd_f32": (c_int, [_P, c_int64, c_int64, c_int, c_float, _P, _P, c_size_t, _P]),
    "gda_softmax_entropy_bwd_f32": (c_int, [_P, c_int64, c_int64, c_int, c_float, _P, _P, c_int64, _P]),
    "gda_mmd_chunked_plan": (c_int, [c_int, c_int64, c_int64, c_float, c_int, _P, c_int]),
    "gda_mmd_chunked_workspace_bytes": (c_size_t, [c_int, c_int64, c_int64]),
    "gda_mmd_chunked_fwd_f32": (c_int, [_P, c_int64, _P, c_int64, c_int64, _P, _P, c_int, c_int64, c_float, c_int, c_float,
                                        c_float, _P, _P, _P, c_int64, _P, _P, _P, c_int64, c_int, _P, c_size_t, _P]),
    "gda_mmd_fused_bwd_ld_f32": (c_int, [_P, c_int64, c_int, c_int, c_int64, c_int64, _P, c_float, _P,
                                         _P, _P, c_int64, _P, _P, _P, c_int64, _P, _P, c_float, _P, c_float, _P]),
    "gda_gemm_tall_fwd_ex_f32": (c_int, [c_int64, c_int64, c_int64, _P, c_int64, _P, _P, c_int64, _P, c_int64, _P, c_int,
                                         c_float, ctypes.c_uint64, _P, ctypes.c_uint32, ctypes.c_uint32, _P]),
    "gda_gemm_tall_wgrad_gather_f32": (c_int, [c_int64, c_int64, _P, c_int64, _P, c_int64, _P, _P, c_int64, _P, _P, c_size_t, _P]),
    "gda_gemm_nn_mask_f32": (c_int, [c_int64, c_int64, c_int64, _P, c_int64, _P, c_int64, _P, c_int64, _P, c_int64, c_float, _P]),
    "gda_grl_disc_workspace_bytes": (c_size_t, [c_int64, c_int64, c_int]),
    "gda_grl_mlp_ce_workspace_bytes": (c_size_t, [c_int64, c_int64]),
    "gda_lsgan_head_workspace_bytes": (c_size_t, [c_int64]),
    "gda_lsgan_head_fwd_f32": (c_int, [_P, c_int64, c_int64, c_int64, _P, _P, c_float, _P, _P, _P, c_size_t, _P]),
    "gda_lsgan_head_bwd_f32": (c_int, [_P, c_int64, c_int64, c_int64, _P, _P, c_float, _P, _P, c_int64, _P, _P,
                                       _P, c_size_t, _P]),
    "gda_grl_mlp_ce_fwd_f32": (c_int, [_P, c_int64, c_int64, _P, c_int64, c_int64, c_int64, c_int64, _P, _P, _P, _P,
                                       c_float, ctypes.c_uint64, _P, ctypes.c_uint32, _P, _P, c_size_t, _P]),
    "gda_grl_mlp_ce_bwd_f32": (c_int, [_P, c_int64, c_int64, _P, c_int64, c_int64, c_int64, c_int64, _P, _P, _P, _P,
                                       c_float, ctypes.c_uint64, _P, ctypes.c_uint32, _P, c_int, c_float, _P,
                                       _P, _P, _P, _P, _P, _P, _P, c_size_t, _P]),
    "gda_mlp_head_fwd_f32": (c_int, [c_int, _P, c_int64, c_int64, _P, c_int64, c_int64, c_int64, c_int64, _P, _P, _P, _P,
                                     c_float, ctypes.c_uint64, _P, ctypes.c_uint32, _P, _P, c_size_t, _P]),
    "gda_mlp_head_bwd_f32": (c_int, [c_int, _P, c_int64, c_int64, _P, c_int64, c_int64, c_int64, c_int64, _P, _P, _P, _P,
                                     c_float, ctypes.c_uint64, _P, ctypes.c_uint32, _P, c_int, c_float, _P,
                                     _P, _P, _P, _P, _P, _P, _P, c_size_t, _P]),
    "gda_grl_disc_ce_fwd_f32": (c_int, [_P, c_int64, c_int64, _P, c_int64, c_int64, c_int64, c_int,
                                        _P, _P, _P, _P, _P, _P, c_size_t, _P]),
    "gda_grl_disc_ce_bwd_f32": (c_int, [_P, c_int64, c_int64, _P, c_int64, c_int64, c_int64, c_int,
                                        _P, _P, _P, _P, c_float, _P, _P, _P, _P, _P, c_size_t, _P]),
    "gda_relu_dropout_fwd_f32": (c_int, [_P, _P, c_int64, c_float, ctypes.c_uint64, _P, ctypes.c_uint32, _P]),
    "gda_relu_dropout_bwd_f32": (c_int, [_P, _P, _P, c_int64, c_float, _P]),
    "gda_relu_dropout_tiled_fwd_f32": (c_int, [_P, c_int64, c_int64, _P, c_float, ctypes.c_uint64, _P, ctypes.c_uint32, _P]),
    "gda_gather_rows_f32": (c_int, [_P, c_int64, c_int64, _P, c_int64, _P, c_int64, _P]),
    "gda_segment_mean_fwd_f32": (c_int, [_P, c_int64, _P, c_int64, c_int64, _P, c_int64, _P]),
    "gda_segment_mean_bwd_f32": (c_int, [_P, c_int64, _P, _P, c_int64, c_int64, _P, c_int64, _P]),
    "gda_sampler_create": (c_int, [_P, _P, c_int64, c_int64, ctypes.POINTER(c_void_p)]),
    "gda_sampler_destroy": (None, [_P]),
    "gda_sampler_set_threads": (c_int, [_P, c_int]),
    "gda_sampler_sample": (c_int, [_P, _P, c_int64, _P, c_int, ctypes.c_uint64,
                                   ctypes.POINTER(c_int64), ctypes.POINTER(c_int64)]),
    "gda_sampler_fetch": (c_int, [_P, _P, _P, _P]),
    "gda_sampler_csr_norm": (c_int, [_P, _P, _P, _P, _P, _P, _P]),
    "gda_dsampler_graph_workspace_bytes": (c_size_t, [c_int64, c_int64]),
    "gda_dsampler_build_graph": (c_int, [_P, _P, c_int64, c_int64, _P, _P, _P, _P, c_size_t, _P]),
    "gda_dsampler_caps": (c_int, [c_int64, _P, c_int, c_int64, c_int64, c_int64, _P, _P]),
    "gda_dsampler_workspace_bytes": (c_size_t, [c_int64, _P, c_int, c_int64, c_int64, c_int64]),
    "gda_dsampler_sample": (c_int, [_P, _P, c_int64, c_int64, c_int64, _P, c_int64, _P, c_int, ctypes.c_uint64,
                                    _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, c_size_t, _P]),
    "gda_event_create": (c_int, [_P]),
    "gda_event_destroy": (c_int, [_P]),
    "gda_event_record": (c_int, [_P, _P]),
    "gda_event_synchronize": (c_int, [_P]),
    "gda_stream_wait_event": (c_int, [_P, _P]),
    "gda_dsampler_batch": (c_int, [_P, _P, c_int64, c_int64, c_int64, _P, c_int64, _P, _P, c_int, ctypes.c_uint64,
                                   _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, c_size_t, _P, _P, _P, _P, c_size_t, _P]),
    "gda_dsampler_batch_ex": (c_int, [_P, _P, c_int64, c_int64, c_int64, _P, c_int64, _P, _P, c_int, ctypes.c_uint64,
                                      _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, c_size_t, _P, _P, _P, c_int64, _P,
                                      c_size_t, _P]),
    "gda_selection_csr_host": (c_int, [_P, c_int, c_int64, c_int64, c_int64, c_int64, _P, _P]),
    "gda_ppmi_build_host": (c_int, [_P, _P, c_int64, c_int64, c_int, c_int, ctypes.c_uint64,
                                    ctypes.POINTER(c_void_p)]),
    "gda_ppmi_workspace_bytes": (c_size_t, [c_int64, c_int64, c_int, c_int]),
    "gda_ppmi_build": (c_int, [_P, _P, c_int64, c_int64, c_int, c_int, ctypes.c_uint64, _P, _P, _P, _P, _P,
                               c_size_t, _P]),
    "gda_edge_list_size": (c_int64, [_P]),
    "gda_edge_list_fetch": (c_int, [_P, _P, _P, _P]),
    "gda_edge_list_destroy": (None, [_P]),
    "gda_spmm_csr_axpby_f32": (c_int, [_P, _P, _P, c_int64, c_int64, _P, c_int64, _P, c_int64, c_float, c_float,
                                       _P, c_int64, c_float, _P, _P, _P]),
    "gda_kstep_max_rows": (c_int, []),
    "gda_kstep_plan_bytes": (c_size_t, [c_int]),
    "gda_kstep_plan_host": (c_int, [_P, _P, _P, c_int64, _P, c_size_t]),
    "gda_kstep_plan_host_ex": (c_int, [_P, _P, _P, c_int64, c_int, _P, c_size_t]),
    "gda_kstep_lds_f32": (c_int, [_P, c_int, c_int64, c_int64, c_int, _P, c_int64, c_int, _P, c_int64, c_int,
                                  _P, _P, _P, _P]),
    "gda_kstep_lds_colmajor_f32": (c_int, [_P, c_int, c_int64, c_int64, c_int, _P, c_int64, _P, c_int64, _P, _P, _P]),
    "gda_relu_dropout_pair_fwd_f32": (c_int, [_P, _P, c_int64, c_int64, c_float, ctypes.c_uint64, _P, ctypes.c_uint32,
                                              ctypes.c_uint32, _P]),
    "gda_relu_dropout_pair_workspace_bytes": (ctypes.c_size_t, [c_int64]),
    "gda_relu_dropout_pair_bwd_f32": (c_int, [_P, _P, _P, c_int64, c_int64, c_float, _P, _P, ctypes.c_size_t, _P]),
    "gda_stack2_f32": (c_int, [_P, _P, _P, c_int64, _P]),
    "gda_relu_dropout_bwd2_f32": (c_int, [_P, _P, _P, _P, c_int64, ctypes.c_float, _P]),
    "gda_colsum_workspace_bytes": (ctypes.c_size_t, [c_int64, c_int64]),
    "gda_colsum_f32": (c_int, [_P, c_int64, c_int64, c_int64, _P, _P, ctypes.c_size_t, _P]),
    "gda_softmax_nll_fwd_ex_f32": (c_int, [_P, c_int64, _P, c_int64, c_int, _P, _P, _P, ctypes.c_size_t, _P]),
    "gda_mixup_combine_workspace_bytes": (ctypes.c_size_t, [c_int64, c_int64]),
    "gda_mixup_combine_fwd_f32": (c_int, [_P, _P, _P, c_int, _P, _P, c_int64, c_int64, c_float, c_float, ctypes.c_uint64,
                                          _P, ctypes.c_uint32, ctypes.c_uint32, _P, _P, _P]),
    "gda_mixup_combine_bwd_f32": (c_int, [_P, _P, _P, _P, c_int, c_int64, c_int64, c_float, c_float, _P, _P, _P, _P, _P,
                                          ctypes.c_size_t, _P]),
    "gda_relu_dropout_fwd_cm_f32": (c_int, [_P, c_int64, _P, c_int64, c_int64, c_float, ctypes.c_uint64, _P,
                                           ctypes.c_uint32, _P]),
    "gda_relu_dropout_bwd_cm_f32": (c_int, [_P, _P, _P, c_int64, c_int64, c_int64, c_float, _P]),
    "gda_transpose_f32": (c_int, [_P, c_int64, _P, c_int64, c_int64, c_int64, _P]),
    "gda_wgan_critic_workspace_bytes": (c_size_t, [c_int64, c_int64, c_int64, c_int, c_int]),
    "gda_wgan_critic_f32": (c_int, [_P, c_int64, _P, c_int64, c_int, _P, _P, _P, c_int64, _P, _P, _P, _P, c_int,
                                    c_float, ctypes.c_uint64, _P, ctypes.c_uint32, c_float, _P, _P, _P, _P, _P,
                                    _P, c_size_t, _P]),
    "gda_wgan_critic_adam_f32": (c_int, [_P, c_int64, _P, c_int64, c_int, _P, _P, _P, c_int64, c_int,
                                         c_float, ctypes.c_uint64, _P, ctypes.c_uint32, c_float, _P, _P,
                                         c_float, c_float, c_float, c_float, c_float, _P, c_size_t, _P]),
    "gda_laplacian_workspace_bytes": (c_size_t, [c_int64]),
    "gda_laplacian_fwd_f32": (c_int, [_P, _P, c_int64, c_int, _P, c_int64, _P, _P, _P, c_size_t, _P]),
    "gda_laplacian_bwd_f32": (c_int, [_P, _P, _P, _P, c_int64, c_int, _P, c_int64, _P, _P, _P, c_int64, _P]),
    "gda_softmax_nll_workspace_bytes": (c_size_t, []),
    "gda_softmax_nll_fwd_f32": (c_int, [_P, c_int64, _P, c_int64, c_int, _P, _P, c_size_t, _P]),
    "gda_softmax_nll_bwd_f32": (c_int, [_P, c_int64, _P, c_int64, c_int, _P, _P, c_int64, _P]),
    "gda_softmax_nll_fwd_nv_f32": (c_int, [_P, c_int64, _P, c_int64, c_int, _P, _P, _P, _P, c_size_t, _P]),
    "gda_softmax_nll_bwd_nv_f32": (c_int, [_P, c_int64, _P, c_int64, c_int, _P, _P, _P, c_int64, _P]),
    "gda_gemm_workspace_bytes": (c_size_t, [c_int, c_int64, c_int64, c_int64]),
    "gda_gemm_f32": (c_int, [c_int, c_int64, c_int64, c_int64, _P, c_int64, _P, c_int64, _P, c_int64,
                             _P, c_size_t, _P]),
    "gda_gemm_ex_f32": (c_int, [c_int, c_int64, c_int64, c_int64, _P, c_int64, _P, c_int64, _P, c_int64, _P, _P,
                                _P, c_size_t, _P]),
    "gda_gemm_skinny_workspace_bytes": (c_size_t, [c_int, c_int64, c_int64, c_int64]),
    "gda_gemm_skinny_f32": (c_int, [c_int, c_int64, c_int64, c_int64, _P, c_int64, _P, c_int64, _P, c_int64, _P, _P,
                                    _P, c_size_t, _P]),
    "gda_gemm_tall_workspace_bytes": (c_size_t, [c_int, c_int64, c_int64, c_int64]),
    "gda_gemm_tall_f32": (c_int, [c_int, c_int64, c_int64, c_int64, _P, c_int64, _P, c_int64, _P, c_int64, _P, _P,
                                  _P, c_size_t, _P]),
    "gda_attention_workspace_bytes": (c_size_t, [c_int64, c_int64]),
    "gda_attention_fuse_fwd_f32": (c_int, [c_int, _P, _P, c_int64, c_int64, _P, _P, _P, c_int64, _P, _P]),
    "gda_attention_fuse_bwd_f32": (c_int, [c_int, _P, _P, c_int64, c_int64, _P, _P, _P, c_int64, _P, _P, _P, _P, c_size_t, _P]),
    "gda_adam_multi_f32": (c_int, [_P, c_int, c_float, c_float, c_float, c_float, c_float, _P]),
    "gda_adam_multi_ex_f32": (c_int, [_P, c_int, c_float, c_float, c_float, c_float, c_float, c_int, _P]),
    "gda_adam_multi_sum_f32": (c_int, [_P, _P, c_int, c_float, c_float, c_float, c_float, c_float, c_int, _P]),
    "gda_step_bump": (c_int, [_P, _P, c_int, _P]),
    "gda_rccl_load": (c_int, [ctypes.c_char_p]),
    "gda_comm_unique_id": (c_int, [_P, c_size_t]),
    "gda_comm_init_rank": (c_int, [_P, c_size_t, c_int, c_int, ctypes.POINTER(c_void_p)]),
    "gda_comm_destroy": (c_int, [_P]),
    "gda_allreduce_f32": (c_int, [_P, c_int64, _P, _P]),
    "gda_allgather_f32": (c_int, [_P, _P, c_int64, _P, _P]),
    "gda_csr_square_host": (c_int, [_P, _P, _P, c_int64, c_int, c_int64, ctypes.POINTER(c_void_p)]),
    "gda_two_hop_host": (c_int, [_P, _P, c_int64, c_int64, c_int, c_int, ctypes.POINTER(c_void_p)]),
    "gda_walk_smooth_host": (c_int, [_P, _P, c_int64, c_int64, c_int, ctypes.c_uint64, c_int,
                                     ctypes.POINTER(c_void_p)]),
}
EXPORTED_SYMBOLS = tuple(_SIGNATURES)

class AdamTensorStruct(ctypes.Structure):
    """``gda_adam_tensor`` of include/gda_hip.h."""
    _fields_ = [("param", c_void_p), ("grad", c_void_p), ("exp_avg", c_void_p), ("exp_avg_sq", c_void_p),
                ("step", c_void_p), ("numel", c_int64)]


class RowSplitStruct(ctypes.Structure):
    """``gda_row_split`` of include/gda_hip.h."""
    _fields_ = [("threshold", ctypes.c_int32), ("n_long", ctypes.c_int32), ("n_chunks", ctypes.c_int32),
                ("long_rows", c_void_p), ("long_chunk_ptr", c_void_p), ("chunk_long", c_void_p),
                ("scratch", c_void_p), ("counts_dev", c_void_p)]


_lib = None


class GdaError(RuntimeError):
    pass


def lib():
    """The loaded library; raises (never falls back) when it is absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise GdaError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(pygda_amd has no CPU / eager fallback)")
        handle = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(handle, name)
            fn.restype, fn.argtypes = res, args
        _lib = handle
    return _lib


_DEBUG_SYNC = os.environ.get("PYGDA_AMD_DEBUG_SYNC") == "1"


def check(status, what):
    if _DEBUG_SYNC:                       # serialise and name every C-ABI call (debugging aid)
        import sys
        print("[gda] launched", what, file=sys.stderr, flush=True)
        torch.cuda.synchronize()
        print("[gda] done    ", what, file=sys.stderr, flush=True)
    if status != 0:
        msg = lib().gda_status_string(int(status)).decode()
        raise GdaError(f"{what} failed with status {status}: {msg}")


def ptr(t):
    return None if t is None else c_void_p(t.data_ptr())


def stream():
    """The raw handle of torch's current stream on the current device.  (``torch.cuda.current_stream()`` builds a
    Stream object per call: ~10 us, a hundred times per eager training step.)"""
    return c_void_p(torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice()))


def require_gpu_tensor(t, name, dtype=None):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise GdaError(f"{name} must be a tensor on the MI355X device (got {getattr(t, 'device', type(t))}); "
                       "pygda_amd has no CPU path")
    if dtype is not None and t.dtype != dtype:
        raise GdaError(f"{name} must have dtype {dtype}, got {t.dtype}")
    return t


_ws_cache = {}


def workspace(nbytes, device, tag):
    """A reusable byte scratch buffer per (device, tag, stream): kernels never allocate."""
    key = (str(device), tag, torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice()))
    buf = _ws_cache.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device)
        _ws_cache[key] = buf
    return buf
