"""ctypes binding of libasrhip.so (include/asr_hip.h).  The library must exist: there
is no CPU or PyTorch fallback -- a missing build raises at import of this module."""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, 'libasrhip.so')


class GemmDesc(C.Structure):
    _fields_ = [('M', C.c_int), ('K', C.c_int), ('N', C.c_int), ('lda', C.c_int), ('ldw', C.c_int),
                ('ldo_a', C.c_int), ('ldo_y', C.c_int), ('ntaps', C.c_int), ('B', C.c_int),
                ('H', C.c_int), ('W', C.c_int), ('wmode', C.c_int), ('relu', C.c_int),
                ('accumulate', C.c_int), ('y_unpadded', C.c_int)]


class PixMap(C.Structure):
    _fields_ = [('kind', C.c_int), ('B', C.c_int), ('H', C.c_int), ('W', C.c_int), ('C', C.c_int), ('ld', C.c_int)]


_P = C.c_void_p
_I = C.c_int
_F = C.c_float
_D = C.c_double
_Z = C.c_size_t
_L = C.c_long

SIGNATURES = {
    'asr_version': (C.c_int, []),
    'asr_last_error': (C.c_char_p, []),
    'asr_last_kernel': (C.c_char_p, []),
    'asr_fbank': (_I, [_P, _P, _I, _I, _I, _I, _I, _D, _I, _P, _P, _P, _I, _P, _P, _I, _P, _I, _P, _P]),
    'asr_lfr': (_I, [_P, _P, _I, _I, _I, _I, _I, _I, _P, _P]),
    'asr_tap_gemm': (_I, [C.POINTER(GemmDesc), _P, _P, _P, _P, _P, _P, _P, _P]),
    'asr_arrange_weights_bytes': (_Z, [_I, _I, _I]),
    'asr_arrange_weights': (_I, [_P, _I, _I, _I, _I, _I, _P, _P]),
    'asr_tap_gemm_pw': (_I, [C.POINTER(GemmDesc), _P, _P, _P, _P, _P, _P, _P, _P]),
    'asr_tap_gemm_gated_workspace': (_Z, [C.POINTER(GemmDesc)]),
    'asr_poolmax_index_bytes': (_Z, [_I, _I, _I, _I]),
    'asr_winograd_poolmax_supported': (_I, [_P, _P]),
    'asr_tap_gemm_wino_poolmax': (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    'asr_tap_gemm_gated_poolmax': (_I, [_P, _P, _P, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    'asr_poolavg_index_bytes': (_Z, [_I, _I, _I, _I]),
    'asr_tap_gemm_wino_poolavg': (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    'asr_tap_gemm_gated_poolavg': (_I, [_P, _P, _P, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    'asr_tap_gemm_gated': (_I, [C.POINTER(GemmDesc), _P, _P, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    'asr_tap_gemm_gated_dense_supported': (_I, [C.POINTER(GemmDesc), _I, _I, _I]),
    'asr_tap_gemm_gated_dense_workspace': (_Z, [C.POINTER(GemmDesc), _I, _I]),
    'asr_tap_gemm_gated_dense': (_I, [C.POINTER(GemmDesc), _P, _P, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    'asr_tap_wgrad_workspace': (_Z, [C.POINTER(GemmDesc)]),
    'asr_tap_wgrad': (_I, [C.POINTER(GemmDesc), _P, _P, _I, _P, _P, _P]),
    'asr_tap_wgrad_direct': (_I, [C.POINTER(GemmDesc), _P, _P, _I, _P, _P, _P]),
    'asr_cell1_fwd': (_I, [_P, _I, _I, _I, _I, _P, _P, _P, _P, _I, _P, _P]),
    'asr_cell1_bwd_workspace': (_Z, [_I, _I, _I, _I]),
    'asr_cell1_bwd': (_I, [_P, _I, _I, _I, _I, _P, _P, _P, _P, _I, _P, _P, _P, _P, _P, _P, _P]),
    'asr_pool_fwd': (_I, [_P, _I, _I, _I, _I, _P, _P, _I, _P, _P]),
    'asr_cell_bwd_pre_workspace': (_Z, [_I, _I, _I, _I]),
    'asr_cell_bwd_pre': (_I, [_P, _I, _P, _I, _I, _I, _I, _P, _P, _I, _P, _P, _P, _P, _P, _P]),
    'asr_se_state_floats': (_Z, [_I, _I, _I]),
    'asr_se_fwd_workspace': (_Z, [_I, _I, _I, _I]),
    'asr_se_fwd': (_I, [_P, _P, _I, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    'asr_se_fwd_sums': (_I, [_P, _P, _I, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _I, _P, _P]),
    'asr_se_bwd_workspace': (_Z, [_I, _I, _I, _I, _I]),
    'asr_se_bwd': (_I, [_P, _P, _I, _I, _I, _I, _I, _P, _P, _P, _P, _P, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    'asr_se_bwd_cell_workspace': (_Z, [_I, _I, _I, _I, _I]),
    'asr_se_bwd_cell': (_I, [_P, _P, _I, _I, _I, _I, _I, _P, _P, _P, _P, _P, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    'asr_se_bwd_cell_sums': (_I, [_P, _P, _I, _I, _I, _I, _I, _P, _P, _P, _P, _P, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _P, _P]),
    'asr_axpy': (_I, [_P, _P, _Z, _F, _I, _P]),
    'asr_softmax_log_fwd': (_I, [_P, _I, _I, _I, _F, _P, _P]),
    'asr_softmax_log_bwd': (_I, [_P, _P, _I, _I, _I, _F, _F, _P, _P]),
    'asr_relu_bwd': (_I, [_P, _P, _Z, _P, _P]),
    'asr_relu_bwd_scaled': (_I, [_P, _P, _Z, _F, _P, _P]),
    'asr_colsum_workspace': (_Z, [_I, _I]),
    'asr_colsum': (_I, [_P, _I, _I, _I, _P, _P, _P]),
    'asr_ctc_workspace': (_Z, [_I, _I, _I]),
    'asr_ctc_loss': (_I, [_P, _I, _I, _I, _P, _I, _P, _P, _I, _P, _P, _P, _P, _P]),
    'asr_ctc_greedy_workspace': (_Z, [_I, _I]),
    'asr_ctc_greedy': (_I, [_P, _I, _I, _I, _P, _I, _P, _P, _P, _P, _P]),
    'asr_edit_distance': (_I, [_P, _I, _P, _P, _I, _P, _I, _P, _P]),
    'asr_adam_tf': (_I, [_P, _P, _P, _P, _Z, _F, _F, _F, _F, _F, _P]),
    'asr_attention_fwd': (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _F, C.c_uint, _P, _P, _P]),
    'asr_attention_bwd': (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _F, C.c_uint, _P, _P, _P, _P, _P]),
    'asr_attention_fwd_p': (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _F, C.c_uint, _P, _P, _P]),
    'asr_attention_bwd_p': (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _F, C.c_uint, _P, _P, _P, _P, _P]),
    'asr_attention_stats_floats': (_Z, [_I, _I, _I, _I]),
    'asr_attention_stats': (_I, [_P, _P, _I, _I, _I, _I, _I, _I, _I, _P, _P]),
    'asr_attention_fwd_s': (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _F, C.c_uint, _P, _P, _P, _P]),
    'asr_attention_bwd_s': (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _F, C.c_uint, _P, _P, _P, _P, _P, _P]),
    'asr_copy2d': (_I, [_P, _I, _P, _I, _I, _I, _I, _P]),
    'asr_copy2d_batch': (_I, [_P, _I, _I, _I, _P]),
    'asr_transpose_batch': (_I, [_P, _I, _I, _P]),
    'asr_tap_gemm_nt': (_I, [C.POINTER(GemmDesc), _P, _P, _P, _I, _P, _P, _P, _P, _P, _P]),
    'asr_winograd_weights_bytes': (_Z, [_I, _I]),
    'asr_winograd_weights2': (_I, [_P, _I, _I, _I, _I, _P, _Z, _P]),
    'asr_winograd_supported': (_I, [_P]),
    'asr_tap_gemm_wino': (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P]),
    'asr_winograd_sum_rows': (_I, [_P]),
    'asr_tap_gemm_wino_sums': (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    'asr_tap_gemm_wino_sesum': (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P]),
    'asr_tap_gemm_wino_pool': (_I, [_P, _P, _P, _P, _P, _P, _P, _I, _P, _P]),
    'asr_tap_gemm_splitk_workspace': (C.c_size_t, [_P, _I]),
    'asr_tap_gemm_splitk': (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _I, _P, _P]),
    'asr_tap_gemm_relu_bwd': (_I, [C.POINTER(GemmDesc), _P, _P, _P, _P, _P]),
    'asr_tap_gemm_nt_splitk_workspace': (C.c_size_t, [_P, _I]),
    'asr_tap_gemm_nt_splitk': (_I, [_P, _P, _P, _I, _P, _P, _P, _P, _P, _I, _P, C.c_size_t, _P]),
    'asr_dropout': (_I, [_P, _Z, _F, C.c_uint, _P, _P]),
    'asr_add_layernorm_fwd': (_I, [_P, _P, _P, _P, _I, _I, _F, _P, _P, _P, _P]),
    'asr_layernorm_bwd_workspace': (_Z, [_I, _I]),
    'asr_layernorm_bwd': (_I, [_P, _P, _P, _P, _I, _I, _P, _I, _P, _P, _P, _P]),
    'asr_layernorm_bwd_fused': (_I, [_P, _P, _P, _P, _I, _I, _P, _P, _I, _P, _F, C.c_uint, _F, _P, _P, _P, _P, _P]),
    'asr_layernorm_bwd_blocks': (_I, [_I]),
    'asr_colsum_multi_batch': (_I, [_P, _I, _I, _P]),
    'asr_add_layernorm_fwd_dropout': (_I, [_P, _P, _P, _P, _I, _I, _F, _F, C.c_uint, _P, _P, _P, _P]),
    'asr_embed_fwd': (_I, [_P, _P, _P, _I, _I, _I, _I, _F, _P, _P]),
    'asr_embed_bwd': (_I, [_P, _P, _P, _P, _I, _I, _I, _F, _P, _P]),
    'asr_embed_bwd_ids': (_I, [_P, _P, _I, _I, _I, _I, _F, _P, _P]),
    'asr_smoothed_ce': (_I, [_P, _I, _P, _I, _I, _F, _I, _F, _P, _P, _P, _P, _P]),
    'asr_prenet_conv1_fwd': (_I, [_P, _P, _P, _I, _I, _I, _P, _P]),
    'asr_prenet_conv1_bwd_workspace': (_Z, [_I, _I, _I]),
    'asr_prenet_conv1_bwd': (_I, [_P, _P, _I, _I, _I, _P, _P, _P, _P]),
    'asr_bn_workspace': (_Z, [C.POINTER(PixMap)]),
    'asr_bn_stats': (_I, [_P, C.POINTER(PixMap), _F, _P, _P, _P, _P]),
    'asr_bn_moving': (_I, [_P, _P, _I, _F, _F, _F, _P, _P, _P, _P]),
    'asr_bn_apply': (_I, [_P, C.POINTER(PixMap), _P, _P, _P, _P, _P, C.POINTER(PixMap), _I, _P, C.POINTER(PixMap), _P]),
    'asr_bn_bwd': (_I, [_P, C.POINTER(PixMap), _P, C.POINTER(PixMap), _P, _P, _P, _I, _P, C.POINTER(PixMap), _P, _P, _P, _P]),
    'asr_relu_mask': (_I, [_P, C.POINTER(PixMap), _P, C.POINTER(PixMap), _P, C.POINTER(PixMap), _P]),
    'asr_maxpool_bwd': (_I, [_P, _P, _I, _I, _I, _I, _P, _P]),
    'asr_conv_s2_expand': (_I, [_P, _I, _I, _P, _P]),
    'asr_conv_s2_gather': (_I, [_P, _I, _I, _P, _P]),
    'asr_conv_s2_arrange_bytes': (C.c_size_t, [_I]),
    'asr_conv_s2_arrange': (_I, [_P, _I, _P, C.c_size_t, _P]),
    'asr_conv_s2_dgrad': (_I, [_P, _P, _P, _P, _P]),
    'asr_plane_to_T': (_I, [_P, _I, _I, _I, _I, _I, _P, _P]),
    'asr_T_to_plane': (_I, [_P, _P, _I, _I, _I, _I, _I, _P, _P]),
    'asr_attention_nomask_fwd': (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _P, _P, _P]),
    'asr_attention_nomask_bwd': (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P, _P, _P, _P, _P]),
    'asr_freq_attention_fwd': (_I, [_P, _P, _P, _I, _I, _P, _P, _P]),
    'asr_freq_attention_bwd': (_I, [_P, _P, _P, _P, _P, _I, _I, _P, _P, _P, _P, _P]),
    'asr_pix_add_ln_fwd': (_I, [_P, _P, C.POINTER(PixMap), _P, _P, _F, _P, _P, _P, _P]),
    'asr_pix_ln_bwd_workspace': (_Z, [C.POINTER(PixMap)]),
    'asr_pix_ln_bwd': (_I, [_P, _P, _P, C.POINTER(PixMap), _P, _P, _P, _P, _P, _P]),
}

_lib = None


def load():
    """Returns the loaded library (process-wide singleton).  torch is imported first so
    the HIP runtime both sides use is the one PyTorch-ROCm ships."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError('libasrhip.so is not built (%s): run `python -m asr_dfcnn_transformer_amd._build` '
                           'or __graft_entry__.build(); there is no fallback path' % LIB_PATH)
    import torch  # noqa: F401  (loads libamdhip64 of the PyTorch-ROCm wheel)
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)           # AttributeError if the symbol is missing
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


class AsrError(RuntimeError):
    pass


def check(rc, what):
    if rc != 0:
        msg = load().asr_last_error().decode() if rc == -2 else ''
        raise AsrError('%s failed with status %d %s' % (what, rc, msg))
