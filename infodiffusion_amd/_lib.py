"""ctypes binding of libinfodiff_hip.so (include/infodiff_hip.h).

The product path has no CPU or eager-PyTorch fallback: if the shared library is
missing or a kernel rejects its arguments, this raises.
"""
import ctypes as C
import os

import torch  # noqa: F401  -- first: the library must bind to the HIP runtime PyTorch ships, not load a second one

from . import knobs

HERE = os.path.dirname(os.path.abspath(__file__))
# IDF_LIB: another build of the same library (A/B of compile-time variants, tools/build_variant.sh); same no-fallback rule
LIB_PATH = knobs.raw('IDF_LIB') or os.path.join(HERE, 'libinfodiff_hip.so')

F32, BF16 = 0, 1
ERR_UNSUPPORTED, ERR_BADARG = 1001, 1002

_p, _i, _l, _f, _u32 = C.c_void_p, C.c_int, C.c_long, C.c_float, C.c_uint32

class ResblockStage(C.Structure):
    """IdfResblockStage (include/infodiff_hip.h)."""
    _fields_ = [('w', _p), ('bias', _p), ('gamma', _p), ('beta', _p), ('film_t', _p), ('film_a', _p), ('ld_t', _i), ('ld_a', _i),
                ('salt', _u32), ('drop', _i), ('a_out', _p), ('mean', _p), ('rstd', _p), ('sc', _p), ('sh', _p), ('h_out', _p)]


class ResblockArgs(C.Structure):
    """IdfResblockArgs (include/infodiff_hip.h)."""
    _fields_ = [('x', _p), ('x2', _p), ('C1', _i), ('Cin', _i), ('st1', _p), ('st2', _p), ('T1', _i), ('T2', _i),
                ('nstage', _i), ('s', ResblockStage * 3), ('w_sc', _p), ('b_sc', _p), ('y', _p), ('st_out', _p),
                ('seed', _p), ('p_drop', _f), ('eps', _f), ('B', _i), ('w_layout', _i)]


class ResblockBwdStage(C.Structure):
    """IdfResblockBwdStage (include/infodiff_hip.h)."""
    _fields_ = [('w_frag', _p), ('x', _p), ('gamma', _p), ('beta', _p), ('film_t', _p), ('film_a', _p), ('ld_t', _i), ('ld_a', _i),
                ('mean', _p), ('rstd', _p), ('sc', _p), ('sh', _p), ('salt', _u32), ('drop', _i), ('dfilm_t', _p), ('dfilm_a', _p),
                ('dgb', _p), ('dgamma_acc', _p), ('dbeta_acc', _p), ('dx', _p)]


class ResblockBwdArgs(C.Structure):
    """IdfResblockBwdArgs (include/infodiff_hip.h)."""
    _fields_ = [('dy', _p), ('nstage', _i), ('first', _i), ('s', ResblockBwdStage * 3), ('dres2', _p), ('seed', _p),
                ('p_drop', _f), ('B', _i)]


SIGNATURES = {
    'idf_version': ([], C.c_int),
    'idf_last_error': ([], C.c_char_p),
    'idf_conv2d_fwd': ([_p, _p, _p, _p, _p, _p, _p, _p, _u32, _f] + [_i] * 11 + [_p], C.c_int),
    'idf_conv3x3_bf16': ([_p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _p, _p], C.c_int),
    'idf_conv1x1_bf16': ([_p, _p, _i, _p, _p, _p, _p, _i, _i, _i, _i, _i, _p, _p], C.c_int),
    'idf_conv_tiles': ([_i] * 8, C.c_int),
    'idf_conv_gn_advice': ([_i] * 6, C.c_int),
    'idf_conv_gn_bf16': ([_p, _p, _i, _p, _i, _p, _i, _p, _p, _p, _p, _i, _i, _f, _i, _p, _u32, _f] + [_p] * 11 + [_i] * 6 + [_p], C.c_int),
    'idf_gn_coef_from_stats': ([_p, _i, _p, _i, _i, _p, _p, _p, _p, _i, _i, _f, _p, _p, _p, _p, _p, _i, _i, _i, _p], C.c_int),
    'idf_conv_gn_sc_bf16': ([_p, _p, _i, _p, _i, _p, _i, _p, _p, _p, _p, _i, _i, _f, _i, _p, _u32, _f] + [_p] * 11 + [_i] * 6 + [_p] +
                            [_p, _p, _p, _i], C.c_int),
    'idf_conv_dgrad_chain_sc_bf16': ([_p, _p, _p, _p, _i, _p, _p, _p, _u32, _f, _i, _p, _p] + [_i] * 5 + [_p] + [_p, _p, _p, _i],
                                     C.c_int),
    'idf_conv_dgrad_gn_ok': ([_i] * 6, C.c_int),
    'idf_resblock_small_ok': ([_i] * 7, C.c_int),
    'idf_conv_wr_tiles': ([_i] * 6, C.c_int),
    'idf_conv_rs_tiles': ([_i] * 5, C.c_int),
    'idf_conv_rs_fwd_tiles': ([_i] * 5, C.c_int),
    'idf_conv_rs_gn_bf16': ([_p, _p, _i, _p, _p, _p, _p, _i, _i, _f, _i, _p, _u32, _f] + [_p] * 10 + [_i] * 5 + [_p], C.c_int),
    'idf_conv_rs_dgrad_chain_bf16': ([_p, _p, _p, _p, _i, _p, _p, _p, _u32, _f, _i, _p, _p] + [_i] * 5 + [_p], C.c_int),
    'idf_conv_rs_dgrad_gn_tiles': ([_i] * 5, C.c_int),
    'idf_conv_rs_dgrad_gn_bf16': ([_p, _p, _p, _p, _i, _p, _p, _p, _u32, _f, _i] + [_p] * 9 + [_i, _i] + [_p] * 8 + [_i] * 5 + [_p], C.c_int),
    'idf_conv_rs_sync_words': ([], C.c_int),
    'idf_conv_rs_set_spin_limit': ([C.c_uint], C.c_uint),
    'idf_gn_rows_desc_bytes': ([], C.c_int),
    'idf_gn_param_reduce_batched': ([_p, _i, _i, _p], C.c_int),
    'idf_conv_fewc_tiles': ([_i] * 5, C.c_int),
    'idf_conv3x3_fewc_bf16': ([_p, _p, _p, _p, _i, _i, _i, _i, _i, _p, _p], C.c_int),
    'idf_conv_wr_gn_bf16': ([_p, _p, _i, _p, _i, _p, _i, _p, _p, _p, _p, _i, _i, _f, _p, _u32, _f] + [_p] * 10 + [_i] * 5 + [_p], C.c_int),
    'idf_conv_wr_dgrad_gn_bf16': ([_p] * 10 + [_i, _i] + [_p] * 10 + [_u32, _f] + [_i] * 5 + [_p], C.c_int),
    'idf_resblock_small_fwd': ([C.POINTER(ResblockArgs), _p], C.c_int),
    'idf_resblock_small_bwd': ([C.POINTER(ResblockBwdArgs), _p], C.c_int),
    'idf_conv_dgrad_gn_bf16': ([_p] * 10 + [_i, _i] + [_p] * 10 + [_u32, _f] + [_i] * 7 + [_p], C.c_int),
    'idf_conv_dgrad_chain_tiles': ([_i] * 6, C.c_int),
    'idf_conv_dgrad_chain_bf16': ([_p, _p, _p, _i] + [_p] * 7 + [_i, _i] + [_p] * 5 + [_p] + [_p, _p, _p, _i] + [_p] * 3 +
                                  [_u32, _f, _i] + [_p, _p] + [_i] * 6 + [_p], C.c_int),
    'idf_gn_bwd_apply': ([_p, _p, _i, _p, _p, _i] + [_p] * 8 + [_i, _i] + [_p] * 8 + [_i] * 3 + [_p], C.c_int),
    'idf_gn_partials_chunks': ([_i, _i], C.c_int),
    'idf_gn_partials': ([_p, _p, _i, _i, _i, _i, _p], C.c_int),
    'idf_conv2d_wgrad': ([_p, _p, _p, _p, _p, _p, _u32, _f] + [_i] * 11 + [_p], C.c_int),
    'idf_pack_conv_weight': ([_p, _l, _l, _l, _p, _p, _i, _i, _i, _i, _p], C.c_int),
    'idf_pack_conv_weights_batched': ([_p, _i, _i, _p], C.c_int),
    'idf_gn_workspace_floats': ([_i, _i, _i], C.c_int),
    'idf_gn_fused_ok': ([_i, _i, _i, _i, _i], C.c_int),
    'idf_gn_coef_fwd': ([_p, _p, _p, _p, _p, _i, _i, _f, _p, _p, _p, _p, _p, _i, _i, _i, _i, _p], C.c_int),
    'idf_gn_fused_fwd': ([_p, _p, _i] + [_p] * 5 + [_i, _i, _f] + [_p] * 5 + [_u32, _f, _i, _i, _i, _i, _i, _p], C.c_int),
    'idf_gn_fused_bwd': ([_p, _p, _p, _i, _p, _p, _p, _p] + [_p] * 4 + [_i, _i] + [_p] * 10 + [_u32, _f, _i, _i, _i, _i, _i, _p], C.c_int),
    'idf_gn_apply': ([_p, _p, _p, _p, _p, _u32, _f, _i, _i, _i, _i, _i, _p], C.c_int),
    'idf_gn_apply2': ([_p, _p, _i, _p, _p, _p, _p, _u32, _f, _i, _i, _i, _i, _i, _p], C.c_int),
    'idf_conv_wgrad_bf16': ([_p, _p, _p, _p] + [_i] * 10 + [_p], C.c_int),
    'idf_wgrad_desc_bytes': ([], C.c_int),
    'idf_wgrad_kr3_ok': ([_i, _i], C.c_int),
    'idf_wgrad_ring_ok': ([_i] * 5, C.c_int),
    'idf_wgrad_upsub_ok': ([_i, _i], C.c_int),
    'idf_wgrad_desc_fill': ([_p, _i, _p, _p, _i, _p, _p, _p] + [_i] * 11 + [_p, _p, _p, _i, _p, _p], C.c_int),
    'idf_wgrad_reduce_batched': ([_p, _i, _i, _p], C.c_int),
    'idf_conv_wgrad_bf16_batched': ([_p, _i, _i, _i, _i, _i, _p], C.c_int),
    'idf_gn_coef_bwd': ([_p] * 8 + [_i, _i] + [_p] * 12 + [_p, _u32, _f, _i, _i, _i, _i, _i, _p], C.c_int),
    'idf_bgemm': ([_p, _p, _p, _p, _p, _i, _l, _l, _l, _i, _i, _i, _i, _i, _i, _i, _i, _f, _i, _i, _i, _p], C.c_int),
    'idf_temb_film_fwd': ([_p, _p, _i, _p, _p, _p, _p, _i, _p, _i, _p, _p, _i, _p, _p, _i, _p, _p, _i] + [_p] * 8 + [_i, _p],
                          C.c_int),
    'idf_temb_film_parts': ([_i], C.c_int),
    'idf_temb_film_bwd': ([_p, _p, _p, _p, _i, _p, _i, _p, _i, _p, _i, _p, _i, _p, _i] + [_p] * 6 + [_p] * 13 + [_i, _p],
                          C.c_int),
    'idf_softmax_fwd': ([_p, _l, _i, _i, _p], C.c_int),
    'idf_softmax_bwd': ([_p, _p, _l, _i, _i, _p], C.c_int),
    'idf_attn_fused_ok': ([_i, _i, _i], C.c_int),
    'idf_attnblock_ok': ([_i, _i, _i], C.c_int),
    'idf_attn_fold_batched': ([_p, _i, _i, _p], C.c_int),
    'idf_attn_fold_bwd_batched': ([_p, _i, _i, _p], C.c_int),
    'idf_attn_res_tiles': ([_i, _i, _i, _i], C.c_int),
    'idf_attn_fwd_res': ([_p, _p, _p, _p, _p, _p, _i, _i, _i, _f, _p], C.c_int),
    'idf_upconv_tiles': ([_i, _i, _i, _i], C.c_int),
    'idf_upconv_pack_batched': ([_p, _i, _l, _p], C.c_int),
    'idf_upconv_dgrad_ok': ([_i, _i, _i, _i], C.c_int),
    'idf_downconv_dgrad_ok': ([_i, _i, _i, _i], C.c_int),
    'idf_downconv_dgrad_bf16': ([_p, _p, _p, _p, _i, _i, _i, _i, _i, _p], C.c_int),
    'idf_upconv_dgrad_bf16': ([_p, _p, _p, _i, _i, _i, _i, _i, _p], C.c_int),
    'idf_upconv_bf16': ([_p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _p], C.c_int),
    'idf_attnblock_fwd': ([_p, _p, _i, _p, _p, _f] + [_p] * 12 + [_f, _i, _i, _i, _p], C.c_int),
    'idf_attn_fwd': ([_p, _p, _p, _i, _i, _i, _f, _i, _p], C.c_int),
    'idf_attn_bwd': ([_p, _p, _p, _p, _p, _i, _i, _i, _f, _i, _p], C.c_int),
    'idf_attn_bwd_o': ([_p, _p, _p, _p, _p, _i, _i, _i, _f, _i, _p], C.c_int),
    'idf_prep_u8': ([_p, _p, _p, _i, _i, _i, _i, _p], C.c_int),
    'idf_qsample': ([_p, _p, _p, _p, _p, _p, _p, _l, _l, _i, _p], C.c_int),
    'idf_gather_rows': ([_p, _p, _p, _i, _i, _p], C.c_int),
    'idf_silu_fwd': ([_p, _p, _l, _p], C.c_int),
    'idf_silu_bwd': ([_p, _p, _p, _l, _p], C.c_int),
    'idf_loss_fwd': ([_p, _p, _p, _f, _f, _f, _p, _p, _l, _i, _p], C.c_int),
    'idf_loss_bwd': ([_p, _p, _p, _f, _f, _f, _p, _i, _p, _l, _i, _p], C.c_int),
    'idf_objective_fwd': ([_p, _p, _p, _f, _f, _f, _p, _p, _i, _i, _i, _f, _p, _p, _l, _i, _p], C.c_int),
    'idf_sampler_step': ([_p] * 7 + [_i, _l, _i, _p], C.c_int),
    'idf_mmd_fwd': ([_p, _p, _i, _i, _i, _p, _p, _p], C.c_int),
    'idf_mmd_bwd': ([_p, _p, _i, _i, _i, _p, _f, _p, _p], C.c_int),
    'idf_colsum_blocks': ([_l], C.c_int),
    'idf_colsum': ([_p, _p, _p, _l, _i, _i, _p], C.c_int),
    'idf_pool2_sum': ([_p, _p, _i, _i, _i, _i, _i, _p], C.c_int),
    'idf_ln_silu_fwd': ([_p, _p, _p, _p, _p, _p, _i, _i, _f, _p, _u32, _f, _p], C.c_int),
    'idf_ln_silu_bwd': ([_p] * 9 + [_i, _i, _p, _u32, _f, _p], C.c_int),
    'idf_clip_adamw': ([_p, _i, _p, _p, _p, _f, _f, _f, _f, _f, _i, _p], C.c_int),
    'idf_dropout_mask': ([_p, _u32, _f, _p, _l, _p], C.c_int),
}

_lib = None


class HipKernelError(RuntimeError):
    pass


def load():
    """Load the library (build it first with `python -m infodiffusion_amd.build`
    or `__graft_entry__.build()`).  Raises if it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            'libinfodiff_hip.so not found at %s -- the HIP extension is required '
            '(no fallback path); run `python -m infodiffusion_amd.build`' % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name, (args, res) in SIGNATURES.items():
        fn = getattr(lib, name)   # AttributeError if the symbol is not exported
        fn.argtypes = args
        fn.restype = res
    _lib = lib
    return lib


def check(rc, name):
    if rc == 0:
        return
    msg = load().idf_last_error().decode(errors='replace')
    if rc == ERR_UNSUPPORTED:
        raise NotImplementedError('%s: %s' % (name, msg))
    raise HipKernelError('%s failed (code %d): %s' % (name, rc, msg))


def call(name, *args):
    rc = getattr(load(), name)(*args)
    check(rc, name)
