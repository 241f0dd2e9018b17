"""ctypes binding of libvqcpc_hip.so (include/vqcpc.h).  No torch types cross the ABI: only raw device pointers,
sizes and the current HIP stream handle.  There is NO CPU fallback: if the library is missing or a call fails this
module raises -- a silent eager path would void every parity claim.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'libvqcpc_hip.so')
LAB_LIB_PATH = os.path.join(_HERE, 'libvqcpc_hip_lab.so')
_is_lab = False
ABI_VERSION = 2       # 2 (round 6): vqcpc_gemm_nt_bf16 gained two arguments in round 5, the dropout mixer of csrc/common.h changed (masks of a given
                      # (seed, index) differ from ABI-1 builds), vqcpc_grad_scale_roll_logged and the plane-operand GEMM entry points were added

_lib = None

c_i64, c_int, c_f32, c_u64, c_ptr = ctypes.c_int64, ctypes.c_int, ctypes.c_float, ctypes.c_uint64, ctypes.c_void_p

# name -> (restype, argtypes); must list every symbol include/vqcpc.h declares (tests/test_abi.py checks it)
SIGNATURES = {
    'vqcpc_abi_version': (c_int, []),
    'vqcpc_last_error': (ctypes.c_char_p, []),
    'vqcpc_clear_runtime_error': (c_int, []),
    'vqcpc_dropout_mask': (c_int, [c_ptr, c_i64, c_f32, c_u64, c_ptr]),
    'vqcpc_check_tokens': (c_int, [c_ptr, c_i64, c_int, c_ptr, c_ptr, c_ptr, c_ptr]),
    'vqcpc_embed_pos_fwd': (c_int, [c_ptr, c_i64, c_int, c_int, c_ptr, c_int, c_int, c_ptr, c_ptr, c_int, c_ptr, c_ptr]),
    'vqcpc_embed_pos_bwd_workspace': (c_i64, [c_i64, c_int, c_int, c_int, c_int, c_int]),
    'vqcpc_embed_pos_bwd': (c_int, [c_ptr, c_i64, c_int, c_int, c_int, c_int, c_int, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr,
                                    c_i64, c_ptr]),
    'vqcpc_block_table_gather': (c_int, [c_ptr, c_ptr, c_ptr, c_i64, c_int, c_int, c_int, c_ptr]),
    'vqcpc_block_table_segsum_workspace': (c_i64, [c_i64, c_int, c_int, c_int]),
    'vqcpc_block_table_segsum': (c_int, [c_ptr, c_ptr, c_ptr, c_i64, c_int, c_int, c_int, c_ptr, c_i64, c_ptr]),
    'vqcpc_block_table_segsum_b16': (c_int, [c_ptr, c_ptr, c_ptr, c_i64, c_int, c_int, c_int, c_ptr, c_i64, c_ptr]),
    'vqcpc_gemm_nt': (c_int, [c_ptr, c_i64, c_ptr, c_i64, c_ptr, c_i64, c_i64, c_int, c_int, c_ptr, c_int, c_f32, c_u64,
                              c_ptr, c_i64, c_f32, c_ptr, c_i64, c_ptr, c_i64, c_ptr]),
    'vqcpc_cast_bf16': (c_int, [c_ptr, c_i64, c_ptr, c_i64, c_int, c_ptr]),
    'vqcpc_gemm_nt_bf16_supported': (c_int, [c_i64, c_int, c_int]),
    'vqcpc_gemm_nt_bf16': (c_int, [c_ptr, c_i64, c_ptr, c_i64, c_ptr, c_i64, c_ptr, c_i64, c_i64, c_int, c_int, c_ptr, c_int,
                                   c_f32, c_u64, c_ptr, c_i64, c_ptr, c_i64, c_f32, c_ptr, c_i64, c_ptr, c_i64, c_ptr]),
    'vqcpc_gemm_tn_bf16_supported': (c_int, [c_i64, c_int, c_int]),
    'vqcpc_gemm_tn_bf16_workspace': (c_i64, [c_i64, c_int, c_int]),
    'vqcpc_gemm_tn_bf16': (c_int, [c_ptr, c_i64, c_ptr, c_i64, c_ptr, c_ptr, c_i64, c_int, c_int, c_int, c_ptr, c_i64, c_ptr]),
    'vqcpc_count_distinct_codes_supported': (c_int, [c_int, c_int]),
    'vqcpc_count_distinct_codes': (c_int, [c_ptr, c_i64, c_ptr, c_i64, c_int, c_int, c_ptr, c_ptr]),
    'vqcpc_accumulate8': (c_int, [c_ptr, c_ptr, c_ptr, c_int, c_ptr]),
    'vqcpc_gemm_gatebits_supported': (c_int, [c_i64, c_int, c_int]),
    'vqcpc_gemm_gatebits_bytes': (c_i64, [c_i64, c_int]),
    'vqcpc_gemm_nt_relu_mask': (c_int, [c_ptr, c_i64, c_ptr, c_i64, c_ptr, c_i64, c_i64, c_int, c_int, c_ptr, c_f32, c_u64, c_ptr,
                                        c_ptr]),
    'vqcpc_gemm_nt_splitk_workspace': (c_i64, [c_i64, c_int, c_int]),
    'vqcpc_gemm_nt_main_rows': (c_i64, [c_i64, c_int, c_int]),
    'vqcpc_gemm_nt_splitk': (c_int, [c_ptr, c_i64, c_ptr, c_i64, c_ptr, c_i64, c_i64, c_int, c_int, c_ptr, c_ptr, c_i64, c_ptr,
                                     c_i64, c_ptr]),
    'vqcpc_gemm_nt_gatebits': (c_int, [c_ptr, c_i64, c_ptr, c_i64, c_ptr, c_i64, c_i64, c_int, c_int, c_ptr, c_f32, c_ptr]),
    'vqcpc_gemm_set_mode': (c_int, [c_int]),
    'vqcpc_gemm_get_mode': (c_int, []),
    'vqcpc_gemm_set_gradient_products': (c_int, [c_int]),
    'vqcpc_gemm_get_gradient_products': (c_int, []),
    'vqcpc_gemm_gradient_scope': (c_int, [c_int]),
    'vqcpc_gemm_tn_workspace': (c_i64, [c_i64, c_int, c_int]),
    'vqcpc_gemm_nt_grad_supported': (c_int, [c_i64, c_int, c_int]),
    'vqcpc_gemm_nt_grad': (c_int, [c_ptr, c_i64, c_ptr, c_i64, c_ptr, c_i64, c_i64, c_int, c_int, c_ptr, c_i64, c_ptr, c_i64,
                                   c_ptr, c_f32, c_ptr, c_ptr]),
    'vqcpc_gemm_nt_grad_splitk_workspace': (c_i64, [c_i64, c_int, c_int]),
    'vqcpc_gemm_nt_grad_splitk': (c_int, [c_ptr, c_i64, c_ptr, c_i64, c_ptr, c_i64, c_i64, c_int, c_int, c_int, c_ptr, c_f32, c_u64,
                                          c_i64, c_ptr, c_i64, c_ptr, c_i64, c_ptr, c_i64, c_ptr, c_ptr]),
    'vqcpc_gemm_nt_grad_tail_supported': (c_int, [c_i64, c_int, c_int]),
    'vqcpc_gemm_nt_grad_tail': (c_int, [c_ptr, c_i64, c_ptr, c_i64, c_ptr, c_i64, c_i64, c_int, c_int, c_ptr, c_f32, c_u64, c_i64, c_ptr,
                                        c_i64, c_ptr, c_i64, c_ptr, c_ptr]),
    'vqcpc_gemm_tn_grad_supported': (c_int, [c_i64, c_int, c_int]),
    'vqcpc_gemm_tn_grad_workspace': (c_i64, [c_i64, c_int, c_int]),
    'vqcpc_gemm_tn_grad': (c_int, [c_ptr, c_i64, c_ptr, c_i64, c_ptr, c_ptr, c_i64, c_int, c_int, c_int, c_ptr, c_i64, c_ptr,
                                   c_ptr]),
    'vqcpc_gemm_nt_f16x3': (c_int, [c_ptr, c_i64, c_ptr, c_i64, c_ptr, c_i64, c_i64, c_int, c_int, c_ptr, c_int, c_f32, c_u64, c_ptr,
                                    c_i64, c_ptr, c_ptr, c_ptr]),
    'vqcpc_weight_planes_many': (c_int, [c_ptr, c_ptr, c_int, c_i64, c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_ptr]),
    'vqcpc_gemm_nt_g3_pl': (c_int, [c_ptr, c_i64, c_ptr, c_i64, c_ptr, c_i64, c_i64, c_int, c_int, c_ptr, c_int, c_f32, c_u64, c_ptr,
                                    c_i64, c_ptr, c_i64, c_ptr, c_f32, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr]),
    'vqcpc_gemm_nt_g3_small': (c_int, [c_ptr, c_i64, c_ptr, c_i64, c_ptr, c_i64, c_i64, c_int, c_int, c_ptr, c_int, c_f32, c_u64, c_i64,
                                       c_ptr, c_i64, c_f32, c_ptr, c_i64, c_ptr, c_i64, c_ptr, c_ptr, c_ptr, c_ptr]),
    'vqcpc_grad_amax': (c_int, [c_ptr, c_i64, c_i64, c_int, c_ptr, c_ptr]),
    'vqcpc_grad_scale_roll': (c_int, [c_ptr, c_int, c_ptr]),
    'vqcpc_grad_scale_roll_counted': (c_int, [c_ptr, c_int, c_ptr, c_ptr]),
    'vqcpc_grad_scale_roll_logged': (c_int, [c_ptr, c_int, c_ptr, c_int, c_ptr]),
    'vqcpc_gemm_tn': (c_int, [c_ptr, c_i64, c_ptr, c_i64, c_ptr, c_ptr, c_i64, c_int, c_int, c_int, c_ptr, c_i64, c_ptr]),
    'vqcpc_gemm_tn_groupable': (c_int, [c_i64, c_int, c_int]),
    'vqcpc_gemm_tn_grouped_workspace': (c_i64, [c_int, c_ptr, c_ptr, c_ptr]),
    'vqcpc_gemm_tn_grouped': (c_int, [c_int, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_int, c_ptr, c_i64,
                                      c_ptr]),
    'vqcpc_transpose': (c_int, [c_ptr, c_ptr, c_int, c_int, c_ptr]),
    'vqcpc_transpose_many': (c_int, [c_ptr, c_ptr, c_ptr, c_int, c_i64, c_ptr]),
    'vqcpc_relattn_force_general': (c_int, [c_int]),
    'vqcpc_relattn_fwd': (c_int, [c_ptr, c_i64, c_ptr, c_ptr, c_ptr, c_i64, c_ptr, c_i64, c_int, c_int, c_int, c_f32,
                                  c_u64, c_ptr]),
    'vqcpc_relattn_bwd_workspace': (c_i64, [c_i64, c_int, c_int, c_int]),
    'vqcpc_relattn_bwd': (c_int, [c_ptr, c_i64, c_ptr, c_i64, c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_ptr, c_ptr, c_i64,
                                  c_int, c_int, c_int, c_f32, c_u64, c_ptr, c_i64, c_ptr]),
    'vqcpc_relattn_tab_fwd': (c_int, [c_ptr, c_i64, c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_ptr, c_i64, c_int, c_int, c_int, c_f32,
                                      c_u64, c_ptr]),
    'vqcpc_relattn_tab_bwd': (c_int, [c_ptr, c_i64, c_ptr, c_i64, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_ptr, c_ptr,
                                      c_i64, c_int, c_int, c_int, c_f32, c_u64, c_ptr, c_i64, c_ptr]),
    'vqcpc_relattn_sub_fwd': (c_int, [c_ptr, c_i64, c_ptr, c_i64, c_ptr, c_ptr, c_ptr, c_i64, c_ptr, c_i64, c_int, c_int,
                                      c_int, c_int, c_f32, c_u64, c_ptr]),
    'vqcpc_relattn_sub_bwd_workspace': (c_i64, [c_i64, c_int, c_int, c_int, c_int]),
    'vqcpc_relattn_sub_bwd': (c_int, [c_ptr, c_i64, c_ptr, c_i64, c_ptr, c_i64, c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_ptr,
                                      c_i64, c_ptr, c_ptr, c_i64, c_int, c_int, c_int, c_int, c_f32, c_u64, c_ptr, c_i64,
                                      c_ptr]),
    'vqcpc_relattn_b16_supported': (c_int, [c_int, c_int, c_int]),
    'vqcpc_relattn_sub_b16_supported': (c_int, [c_int, c_int, c_int, c_int]),
    'vqcpc_relattn_fwd_b16': (c_int, [c_ptr, c_i64, c_ptr, c_ptr, c_ptr, c_i64, c_ptr, c_i64, c_int, c_int, c_int, c_f32,
                                      c_u64, c_ptr]),
    'vqcpc_relattn_bwd_b16': (c_int, [c_ptr, c_i64, c_ptr, c_i64, c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_ptr, c_ptr, c_i64,
                                      c_int, c_int, c_int, c_f32, c_u64, c_ptr, c_i64, c_ptr]),
    'vqcpc_relattn_sub_fwd_b16': (c_int, [c_ptr, c_i64, c_ptr, c_i64, c_ptr, c_ptr, c_ptr, c_i64, c_ptr, c_i64, c_int, c_int,
                                          c_int, c_int, c_f32, c_u64, c_ptr]),
    'vqcpc_relattn_sub_bwd_b16': (c_int, [c_ptr, c_i64, c_ptr, c_i64, c_ptr, c_i64, c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_ptr,
                                          c_i64, c_ptr, c_ptr, c_i64, c_int, c_int, c_int, c_int, c_f32, c_u64, c_ptr, c_i64,
                                          c_ptr]),
    'vqcpc_relattn_x_fwd': (c_int, [c_ptr, c_i64, c_ptr, c_i64, c_ptr, c_i64, c_ptr, c_ptr, c_ptr, c_i64, c_ptr, c_i64, c_int,
                                    c_int, c_int, c_int, c_int, c_f32, c_u64, c_ptr]),
    'vqcpc_relattn_x_bwd_workspace': (c_i64, [c_i64, c_int, c_int, c_int, c_int]),
    'vqcpc_relattn_x_bwd': (c_int, [c_ptr, c_i64, c_ptr, c_i64, c_ptr, c_i64, c_ptr, c_i64, c_ptr, c_ptr, c_ptr, c_ptr, c_i64,
                                    c_ptr, c_i64, c_ptr, c_i64, c_ptr, c_ptr, c_i64, c_int, c_int, c_int, c_int, c_int, c_f32,
                                    c_u64, c_ptr, c_i64, c_ptr]),
    'vqcpc_embedding_bwd': (c_int, [c_ptr, c_i64, c_ptr, c_ptr, c_ptr, c_i64, c_i64, c_int, c_ptr]),
    'vqcpc_add_layernorm_fwd': (c_int, [c_ptr, c_i64, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_int, c_f32,
                                        c_f32, c_u64, c_ptr]),
    'vqcpc_add_layernorm_fwd_b16': (c_int, [c_ptr, c_i64, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_int, c_f32,
                                            c_f32, c_u64, c_ptr]),
    'vqcpc_add_layernorm_bwd_b16': (c_int, [c_ptr, c_ptr, c_i64, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr,
                                            c_i64, c_int, c_f32, c_u64, c_ptr, c_i64, c_ptr]),
    'vqcpc_layernorm_fwd_xb16': (c_int, [c_ptr, c_i64, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_int, c_f32, c_ptr]),
    'vqcpc_layernorm_bwd_xb16': (c_int, [c_ptr, c_ptr, c_i64, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_i64,
                                         c_int, c_f32, c_u64, c_ptr, c_i64, c_ptr]),
    'vqcpc_layernorm_bwd_b16io': (c_int, [c_ptr, c_ptr, c_i64, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_i64,
                                          c_int, c_f32, c_u64, c_ptr, c_i64, c_ptr]),
    'vqcpc_add_layernorm_bwd_workspace': (c_i64, [c_i64, c_int]),
    'vqcpc_add_layernorm_bwd_partials': (c_int, [c_i64, c_int, c_int]),
    'vqcpc_reduce_grouped': (c_int, [c_int, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_int, c_ptr]),
    'vqcpc_reduce_grouped_vec': (c_int, [c_int, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_int, c_ptr]),
    'vqcpc_gemm_tn_deferred_splits': (c_int, [c_i64, c_int, c_int]),
    'vqcpc_gemm_tn_bf16_deferred_splits': (c_int, [c_i64, c_int, c_int]),
    'vqcpc_add_layernorm_bwd': (c_int, [c_ptr, c_ptr, c_i64, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr,
                                        c_i64, c_int, c_f32, c_u64, c_ptr, c_i64, c_ptr]),
    'vqcpc_vq_fwd': (c_int, [c_ptr, c_ptr, c_i64, c_int, c_int, c_int, c_f32, c_int, c_int, c_ptr, c_ptr, c_ptr, c_ptr]),
    'vqcpc_vq_bwd_workspace': (c_i64, [c_i64, c_int, c_int, c_int]),
    'vqcpc_vq_bwd': (c_int, [c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_int, c_int, c_int, c_f32, c_int, c_ptr, c_ptr,
                             c_ptr, c_i64, c_ptr]),
    'vqcpc_dropout_selu_fwd': (c_int, [c_ptr, c_ptr, c_i64, c_f32, c_u64, c_ptr]),
    'vqcpc_dropout_selu_bwd': (c_int, [c_ptr, c_ptr, c_ptr, c_i64, c_f32, c_u64, c_ptr]),
    'vqcpc_nce_fwd': (c_int, [c_ptr, c_ptr, c_ptr, c_ptr, c_int, c_int, c_int, c_int, c_int, c_ptr, c_ptr, c_ptr, c_ptr,
                              c_ptr]),
    'vqcpc_nce_bwd_workspace': (c_i64, [c_int, c_int, c_int, c_int, c_int]),
    'vqcpc_nce_bwd': (c_int, [c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_int, c_int, c_int, c_int, c_int, c_ptr,
                              c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_ptr]),
    'vqcpc_sumsq_workspace': (c_i64, [c_i64]),
    'vqcpc_sumsq': (c_int, [c_ptr, c_i64, c_f32, c_ptr, c_ptr, c_i64, c_ptr]),
    'vqcpc_adam_step': (c_int, [c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_f32, c_f32, c_f32, c_f32, c_int, c_f32, c_f32, c_ptr,
                                c_ptr]),
    'vqcpc_rng_salt_set': (c_int, [c_u64, c_ptr]),
    'vqcpc_rng_salt_advance': (c_int, [c_ptr, c_u64, c_ptr]),
    'vqcpc_rng_salt_from_counter': (c_int, [c_ptr, c_u64, c_ptr]),
    'vqcpc_adam_step_dev': (c_int, [c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_ptr, c_f32, c_f32, c_f32, c_ptr, c_f32, c_f32, c_ptr,
                                    c_ptr]),
    'vqcpc_gru_cell_fwd': (c_int, [c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_int, c_f32, c_u64, c_u64, c_ptr]),
    'vqcpc_gru_cell_bwd': (c_int, [c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_int, c_f32, c_u64, c_u64,
                                   c_ptr]),
    'vqcpc_relattn16_b16_supported': (c_int, [c_int, c_int, c_int]),
    'vqcpc_relattn16_fwd_b16': (c_int, [c_ptr, c_i64, c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_ptr, c_i64, c_int, c_int, c_f32, c_u64,
                                        c_ptr]),
    'vqcpc_relattn16_bwd_b16': (c_int, [c_ptr, c_i64, c_ptr, c_i64, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_ptr, c_ptr, c_i64,
                                        c_int, c_int, c_f32, c_u64, c_ptr, c_i64, c_ptr]),
    'vqcpc_relattn16_fwd_b16io': (c_int, [c_ptr, c_i64, c_ptr, c_ptr, c_ptr, c_i64, c_ptr, c_i64, c_int, c_int, c_f32, c_u64, c_ptr]),
    'vqcpc_relattn16_bwd_b16io': (c_int, [c_ptr, c_i64, c_ptr, c_i64, c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_ptr, c_ptr, c_i64,
                                          c_int, c_int, c_f32, c_u64, c_ptr, c_i64, c_ptr]),
    'vqcpc_gru_step_supported': (c_int, [c_i64, c_int]),
    'vqcpc_gru_step_fwd': (c_int, [c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_int, c_f32, c_u64, c_u64, c_ptr]),
    'vqcpc_gru_step_bwd': (c_int, [c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_int, c_f32, c_u64,
                                   c_u64, c_ptr]),
    'vqcpc_same_sequence_negatives': (c_int, [c_ptr, c_ptr, c_ptr, c_i64, c_int, c_int, c_int, c_ptr]),
    'vqcpc_softmax_ce': (c_int, [c_ptr, c_i64, c_ptr, c_ptr, c_i64, c_ptr, c_ptr, c_i64, c_int, c_ptr]),
    'vqcpc_scale_rows': (c_int, [c_ptr, c_ptr, c_ptr, c_i64, c_int, c_ptr]),
    'vqcpc_upscale_fwd': (c_int, [c_ptr, c_ptr, c_ptr, c_i64, c_int, c_int, c_ptr]),
    'vqcpc_upscale_bwd_workspace': (c_i64, [c_i64, c_int, c_int]),
    'vqcpc_upscale_bwd': (c_int, [c_ptr, c_ptr, c_ptr, c_i64, c_int, c_int, c_ptr, c_i64, c_ptr]),
}

# Entry points of LAB builds only (`VQCPC_LAB=1 python -m vqcpc_bach_amd.build` -> libvqcpc_hip_lab.so, the `#ifdef VQCPC_LAB`
# section of include/vqcpc.h): rejected kernel designs kept for A/B measurements by the tools under tools/
LAB_SIGNATURES = {
    'vqcpc_gemm_bf16_set_variant': (c_int, [c_int]),
    'vqcpc_planes_bytes': (c_i64, [c_i64, c_int]),
    'vqcpc_split3_planes': (c_int, [c_ptr, c_i64, c_i64, c_int, c_ptr, c_ptr]),
    'vqcpc_join3_planes': (c_int, [c_ptr, c_i64, c_int, c_ptr, c_i64, c_ptr]),
    'vqcpc_gemm_nt_planes_supported': (c_int, [c_i64, c_int, c_int]),
    'vqcpc_gemm_nt_planes': (c_int, [c_ptr, c_ptr, c_ptr, c_i64, c_i64, c_int, c_int, c_ptr, c_int, c_f32, c_u64, c_ptr, c_i64,
                                     c_f32, c_ptr, c_i64, c_ptr]),
}


class VqcpcHipError(RuntimeError):
    pass


def load(path=None):
    """dlopen the library (after torch, so that its libamdhip64.so.7 is the one HIP runtime in the process).  The product
    library unless VQCPC_LAB=1 asks for the lab build (measurement tools; see build.py)."""
    global _lib, _is_lab
    if _lib is not None:
        return _lib
    want_lab = os.environ.get('VQCPC_LAB', '0') == '1'
    path = path or os.environ.get('VQCPC_HIP_LIB', LAB_LIB_PATH if want_lab else LIB_PATH)
    if not os.path.exists(path):
        raise VqcpcHipError(f'{path} not found: build it with `{"VQCPC_LAB=1 " if want_lab else ""}python -m vqcpc_bach_amd.build` '
                            f'(hipcc --offload-arch=gfx950). There is no CPU fallback.')
    lib = ctypes.CDLL(path)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = res, args
    _is_lab = hasattr(lib, 'vqcpc_gemm_nt_planes')
    if _is_lab:
        for name, (res, args) in LAB_SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = res, args
    if lib.vqcpc_abi_version() != ABI_VERSION:
        raise VqcpcHipError(f'ABI version mismatch: library {lib.vqcpc_abi_version()} != binding {ABI_VERSION}')
    _lib = lib
    return lib


def clear_runtime_error():
    """Returns and clears the HIP runtime's pending (non-sticky) error of this thread (include/vqcpc.h)."""
    return int(load().vqcpc_clear_runtime_error())


def is_lab():
    """True when the loaded library is a lab build (LAB_SIGNATURES are bound, the tools' environment switches are live)."""
    load()
    return _is_lab


def is_loaded():
    return _lib is not None


def _stream():
    # raw handle of torch's current stream on the current device; the C-level getters avoid building a
    # torch.cuda.Stream object (and its lazy-init / device-count checks: ~30 us) on every kernel launch
    return torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice())


def _p(t):
    if t is None:
        return None
    assert t.is_cuda, 'libvqcpc_hip takes device pointers only'
    return t.data_ptr()              # a plain int: ctypes converts it for a c_void_p parameter


def _check(rc, name):
    if rc != 0:
        raise VqcpcHipError(f'{name} failed ({rc}): {load().vqcpc_last_error().decode()}')


_TENSOR_TYPES = frozenset((torch.Tensor, torch.nn.Parameter))
_FN = {}


def call(name, *args):
    """Invoke an int-returning entry point; tensors become device pointers, the stream is appended.
    Hot path (hundreds of calls per training step): exact-type test instead of isinstance, bound function cached."""
    fn = _FN.get(name)
    if fn is None:
        fn = _FN[name] = getattr(_lib if _lib is not None else load(), name)
    rc = fn(*[a.data_ptr() if type(a) in _TENSOR_TYPES else a for a in args],
            torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice()))
    if rc != 0:
        _check(rc, name)


def query(name, *args):
    """Invoke an int64-returning host-only workspace query."""
    return int(getattr(load(), name)(*args))


def set_gemm_mode(mode):
    """0 = fp32 MFMA (exact), 1 = bf16x6 split MFMA (fp32-class accuracy, faster); +2: 128-tile kernels only,
    +4: 256-tile NT kernel without the ping-pong wave groups, +16: LDS-DMA 256-tile NT kernel, +32: one-wave-per-SIMD software-pipelined 256-tile NT kernel (A/B switches); 8 = plain bf16 operands (one bf16 MFMA per
    product, fp32 accumulation: reduced precision, for BASELINE configs[4] only).  get_gemm_mode() returns 0 / 1 / 2."""
    global _gemm_mode, _gemm_mode_explicit
    rc = load().vqcpc_gemm_set_mode(int(mode))
    _check(rc, 'vqcpc_gemm_set_mode')
    _gemm_mode = None
    _gemm_mode_explicit = True


def use_training_default_gemm_mode():
    """What `train_model()` selects when the caller chose nothing (neither set_gemm_mode() nor VQCPC_GEMM_MODE): the
    bf16x6 split-MFMA arithmetic, i.e. the configuration bench.py measures (fp32-class accuracy: every parity suite runs in
    this mode too).  The bare library default stays the exact fp32 MFMA."""
    global _gemm_mode_explicit
    if not _gemm_mode_explicit and 'VQCPC_GEMM_MODE' not in os.environ:
        set_gemm_mode(1)
        _gemm_mode_explicit = False          # still "nobody chose": a later explicit choice wins as usual


def gemm_mode_state():
    """(arithmetic in force: 0 fp32 MFMA / 1 bf16x6 / 2 bf16, whether a caller chose it): what restore_gemm_mode_state() puts
    back (the A/B sub-switches of set_gemm_mode are not part of it)."""
    return (get_gemm_mode(), _gemm_mode_explicit)


def restore_gemm_mode_state(state):
    global _gemm_mode_explicit
    mode, explicit = state
    if mode != get_gemm_mode():
        set_gemm_mode({0: 0, 1: 1, 2: 8}[mode])
    _gemm_mode_explicit = explicit


def set_gradient_products(products):
    """Opt-in gradient arithmetic of the bf16x6 mode (include/vqcpc.h): 6 (default) or 3 MFMAs per product for the
    256-tile GEMMs launched inside a gradient scope (ops.direct_weight_gradients, i.e. the trainers' loss.backward())."""
    _check(load().vqcpc_gemm_set_gradient_products(int(products)), 'vqcpc_gemm_set_gradient_products')


def get_gradient_products():
    return int(load().vqcpc_gemm_get_gradient_products())


def gradient_scope(open_):
    _check(load().vqcpc_gemm_gradient_scope(1 if open_ else 0), 'vqcpc_gemm_gradient_scope')


def force_general_attention(on):
    """Tests: route L = 16 / 4 through the general-L strip kernels too."""
    _check(load().vqcpc_relattn_force_general(int(bool(on))), 'vqcpc_relattn_force_general')


_gemm_mode = None
_gemm_mode_explicit = False


def get_gemm_mode():
    """0 fp32 MFMA, 1 bf16x6, 2 bf16 (cached: asked on every GEMM of the bf16 path)."""
    global _gemm_mode
    if _gemm_mode is None:
        _gemm_mode = int(load().vqcpc_gemm_get_mode())
    return _gemm_mode


def workspace(nbytes, device):
    return torch.empty(max(int(nbytes), 16), dtype=torch.uint8, device=device)
