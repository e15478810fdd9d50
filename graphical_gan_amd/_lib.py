"""ctypes binding of libggan.so (the C ABI declared in include/ggan.h).

The product path has NO CPU fallback: if the library is missing or a call fails this module raises.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'libggan.so')

ACT_NONE, ACT_LRELU, ACT_RELU, ACT_TANH, ACT_SIGMOID = 0, 1, 2, 3, 4
PACK_MAX = 64
HEAD_BCE_MAX_ROWS = 2048             # include/ggan.h: GGAN_HEAD_BCE_MAX_ROWS
BCE_HEADS = 2                        # include/ggan.h: GGAN_BCE_HEADS
PACK_ARRIVE_INTS = 33 * 1024          # include/ggan.h: GGAN_PACK_ARRIVE_INTS (arrival counters of ggan_pack_adam)
BCE_MAX = 16
ABI_VERSION = 600                    # include/ggan.h: GGAN_ABI_VERSION (struct layouts and entry points this module binds)


class ConvGeom(C.Structure):
    _fields_ = [(n, C.c_int) for n in ('N', 'Ci', 'H', 'W', 'Co', 'Ho', 'Wo', 'k', 'stride', 'pad_t', 'pad_l',
                                       'plan_wgs', 'plan_wgs_filter', 'plan_flags')]      # (the last three default to 0)


PLAN_PLAIN = 1


class ProfRec(C.Structure):
    _fields_ = [('name', C.c_char * 48), ('total_ms', C.c_double), ('launches', C.c_long),
                ('flops', C.c_double), ('bytes', C.c_double), ('grid', C.c_long)]


class GganError(RuntimeError):
    pass


_P = C.c_void_p
_F = C.c_float
_I = C.c_int
_Z = C.c_size_t
_G = C.POINTER(ConvGeom)

# name -> (restype, argtypes); must list every symbol include/ggan.h declares
SIGNATURES = {
    'ggan_version': (_I, []),
    'ggan_last_error': (C.c_char_p, []),
    'ggan_conv2d_workspace': (_Z, [_G]),
    'ggan_conv2d_fwd': (_I, [_G, _P, _P, _P, _P, _I, _F, _P, _Z, _P]),
    'ggan_conv2d_bwd_data': (_I, [_G, _P, _P, _P, _P, _I, _F, _P, _Z, _P]),
    'ggan_conv2d_bwd_filter': (_I, [_G, _P, _P, _P, _P, _P, _Z, _P]),
    'ggan_conv2d_bwd_data_act': (_I, [_G, _P, _P, _I, _F, _P, _P, _P, _Z, _P]),
    'ggan_conv2d_bwd_filter_act': (_I, [_G, _P, _P, _P, _I, _F, _P, _P, _P, _Z, _P]),
    'ggan_conv2d_bwd_filter_parts': (_I, [_G, _P, _P, _P, _I, _F, _I, _P, _Z, C.POINTER(_I), C.POINTER(_Z), _P]),
    'ggan_conv2d_fwd_cast_ring': (_I, [_G, _P, _I, _P, _P, _I, _P, _F, _F, _P, _P, _P, _P, _I, _F, _P]),
    'ggan_conv2d_fwd_masked': (_I, [_G, _P, _P, _P, _P, _I, _F, _P, _Z, _P]),
    'ggan_linear_bn_rows_fwd': (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _F, _I, _F, _P]),
    'ggan_linear_bn_rows_bwd': (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _F, _P]),
    'ggan_deconv2d_fwd': (_I, [_G, _P, _P, _P, _P, _I, _F, _P, _Z, _P]),
    'ggan_deconv2d_bwd_data': (_I, [_G, _P, _P, _P, _P, _Z, _P]),
    'ggan_deconv2d_bwd_filter': (_I, [_G, _P, _P, _P, _P, _P, _Z, _P]),
    'ggan_gemm_workspace': (_Z, [_I, _I, _I]),
    'ggan_gemm': (_I, [_I, _I, _I, _I, _I, _P, _P, _P, _P, _I, _F, _P, _Z, _P]),
    'ggan_conv3d_out_shape': (_I, [_P, _P]),
    'ggan_im2col3d': (_I, [_P, _P, _P, _P]),
    'ggan_col2im3d': (_I, [_P, _P, _P, _P]),
    'ggan_conv3d_igemm_ok': (_I, [_P, _I]),
    'ggan_conv3d_fwd': (_I, [_P, _P, _P, _P, _P, _I, _F, _P, _Z, _P]),
    'ggan_conv3d_wgrad': (_I, [_P, _P, _P, _P, _P, _Z, _P]),
    'ggan_conv3d_dgrad': (_I, [_P, _P, _P, _P, _P]),
    'ggan_mix_rbf_mmd2_fwd': (_I, [_P, _P, _I, _I, _I, _P, _P, _I, _P, _P, _P]),
    'ggan_mix_rbf_mmd2_bwd': (_I, [_P, _P, _I, _I, _I, _P, _P, _I, _P, _P, _P, _P]),
    'ggan_noise_fill': (_I, [_P, _P, _P, _P, _P, _P, _I, _P, _P]),
    'ggan_gmm_latent_fwd': (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _F, _F, _P]),
    'ggan_gmm_latent_bwd': (_I, [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _F, _P]),
    'ggan_gemm_split': (_I, [_I, _I, _I, _I, _I, _P, _P, _I, _P, _P, _P, _P, _I, _P, _I, _F, _P, _Z, _P]),
    'ggan_gemm_colsum': (_I, [_I, _I, _I, _I, _P, _P, _P, _P, _P, _Z, _P]),
    'ggan_dyn_scan_fwd': (_I, [_I, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _F, _P, _P, _P, _P]),
    'ggan_dyn_scan_bwd': (_I, [_I, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _F, _P, _P, _P, _P, _P, _P, _P]),
    'ggan_critic_head_fwd': (_I, [_I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _F, _P, _P, _P, _Z, _P]),
    'ggan_critic_head_bwd': (_I, [_I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _F, _P, _P, _P, _P, _P, _P, _P, _P, _Z, _P]),
    'ggan_critic_head_fwd_bce': (_I, [_I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _F, _P, _P, _I, _I, _P, _P, _P, _P, _P, _P, _Z, _P]),
    'ggan_critic_head_bwd_tail': (_I, [_I, _I, _I, _I, _P, _P, _P, _P, _P, _F, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _P, _P, _P, _P, _P, _P, _Z, _P]),
    'ggan_linear_bwd_data_act': (_I, [_I, _I, _I, _P, _P, _I, _F, _P, _P, _P, _Z, _P]),
    'ggan_linear_bwd_weight_act': (_I, [_I, _I, _I, _P, _P, _P, _I, _F, _P, _P, _P, _Z, _P]),
    'ggan_colsum': (_I, [_P, _P, _I, _I, _P]),
    'ggan_reparam_fwd': (_I, [_P, _P, _P, _P, _P, _Z, _P]),
    'ggan_reparam_bwd': (_I, [_P, _P, _P, _P, _P, _P, _Z, _P]),
    'ggan_agg_div_fwd': (_I, [_I, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P]),
    'ggan_agg_div_bwd': (_I, [_I, _P, _P, _P, _P, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    'ggan_colsum_tall': (_I, [_P, _P, _I, _I, _P, _Z, _P]),
    'ggan_chansum': (_I, [_P, _P, _I, _I, _I, _P, _Z, _P]),
    'ggan_bn_fwd_train': (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _F, _I, _F, _P]),
    'ggan_bn_bwd': (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _P]),
    'ggan_bn_bwd_act': (_I, [_P, _P, _P, _I, _F, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _P]),
    'ggan_bn_bwd_bwd': (_I, [_P, _P, _P, _I, _F, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _P]),
    'ggan_bn_sync_stats': (_I, [_P, _P, _I, _I, _I, _P]),
    'ggan_bn_sync_apply': (_I, [_P, _P, _I, _P, _P, _P, _P, _P, _I, _I, _I, _F, _I, _F, _P]),
    'ggan_bn_sync_bwd_stats': (_I, [_P, _P, _P, _I, _F, _P, _P, _P, _I, _I, _I, _P]),
    'ggan_bn_sync_bwd_apply': (_I, [_P, _P, _P, _I, _F, _P, _P, _P, _P, _I, _I, _P, _P, _P, _I, _I, _I, _P]),
    'ggan_act_fwd': (_I, [_P, _P, _Z, _I, _F, _P]),
    'ggan_act_bwd': (_I, [_P, _P, _P, _Z, _I, _F, _P]),
    'ggan_act_bwd_chansum': (_I, [_P, _P, _P, _P, _I, C.POINTER(_I), _I, _I, _I, _I, _F, _P]),
    'ggan_bias_add': (_I, [_P, _P, _P, _I, _I, _I, _P]),
    'ggan_cast_scale_i32': (_I, [_P, _P, _P, _Z, _F, _F, _P]),
    'ggan_cast_scale_ring_i32': (_I, [_P, _I, _P, _P, _I, _P, _P, _Z, _F, _F, _P]),
    'ggan_axpby': (_I, [_P, _P, _P, _Z, _F, _F, _F, _P]),
    'ggan_mix_mean': (_I, [_P, _P, _P, _P, _I, _I, _I, _P]),
    'ggan_row_lerp': (_I, [_P, _P, _P, _P, _I, _I, _P]),
    'ggan_bce_logits_fwd': (_I, [_P, _F, _F, _P, _I, _I, _P]),
    'ggan_bce_logits_bwd': (_I, [_P, _F, _F, _P, _P, _I, _P]),
    'ggan_bce_logits_multi_fwd': (_I, [C.POINTER(_P), C.POINTER(_F), C.POINTER(_F), C.POINTER(_I), _I, _P, _P]),
    'ggan_bce_logits_multi_bwd': (_I, [C.POINTER(_P), C.POINTER(_F), C.POINTER(_F), C.POINTER(_I), _I, _P, C.POINTER(_P), _P]),
    'ggan_bce_logits_multi_fwd_grad': (_I, [C.POINTER(_P), C.POINTER(_F), C.POINTER(_F), C.POINTER(_I), _I, _P, C.POINTER(_P), _P]),
    'ggan_bce_head_bwd': (_I, [C.POINTER(_P), C.POINTER(_F), C.POINTER(_F), C.POINTER(_I), _I, _P, C.POINTER(_P), _I, _I, _P, _P, _F, _P, _P,
                          _P, _P]),
    'ggan_bce_heads_bwd': (_I, [C.POINTER(_P), C.POINTER(_F), C.POINTER(_F), C.POINTER(_I), _I, _P, C.POINTER(_P), _I, C.POINTER(_I),
                           C.POINTER(_I), C.POINTER(_I), C.POINTER(_P), C.POINTER(_P), C.POINTER(_F), C.POINTER(_P), C.POINTER(_P),
                           C.POINTER(_P), _P]),
    'ggan_mean_fwd': (_I, [_P, _F, _P, _I, _I, _P]),
    'ggan_mean_bwd': (_I, [_P, _F, _P, _I, _P]),
    'ggan_mean_multi_fwd_grad': (_I, [C.POINTER(_P), C.POINTER(_F), C.POINTER(_I), _I, _P, C.POINTER(_P), _P]),
    'ggan_dist_fwd': (_I, [_P, _P, _P, _Z, _I, _F, _I, _P]),
    'ggan_dist_bwd': (_I, [_P, _P, _P, _P, _P, _Z, _I, _F, _P]),
    'ggan_gp_penalty_fwd': (_I, [_P, _P, _P, _I, _I, _F, _P]),
    'ggan_gp_penalty_bwd': (_I, [_P, _P, _P, _P, _I, _I, _F, _P]),
    'ggan_gp_penalty_fwd_grad': (_I, [_P, _P, _P, _P, _P, _I, _I, _F, _P]),
    'ggan_adam_step': (_I, [_P, _P, _P, _P, _Z, _P, _F, _F, _F, _F, _F, _P]),
    'ggan_adam_advance': (_I, [_P, _P]),
    'ggan_rmsprop_step': (_I, [_P, _P, _P, _Z, _F, _F, _F, _F, _F, _F, _P]),
    'ggan_adam_step_counted': (_I, [_P, _P, _P, _P, _Z, _P, _F, _F, _F, _F, _F, _P]),
    'ggan_pack': (_I, [C.POINTER(_P), C.POINTER(_Z), C.POINTER(_Z), _I, _P, _P]),
    'ggan_pack_parts': (_I, [C.POINTER(_P), C.POINTER(_Z), C.POINTER(_Z), C.POINTER(_I), C.POINTER(_Z), _I, _P, _P, _P]),
    'ggan_pack_parts2': (_I, [C.POINTER(_P), C.POINTER(_Z), C.POINTER(_Z), C.POINTER(_I), C.POINTER(_Z), C.POINTER(_P), C.POINTER(_I),
                              C.POINTER(_Z), _I, _P, _P, _P]),
    'ggan_pack_adam': (_I, [C.POINTER(_P), C.POINTER(_Z), C.POINTER(_Z), C.POINTER(_I), C.POINTER(_Z), C.POINTER(_P), C.POINTER(_I),
                            C.POINTER(_Z), _I, _P, _P, _P, _P, _P, _P, _F, _F, _F, _F, _F, _P]),
    'ggan_prof_enable': (_I, [_I]),
    'ggan_prof_reset': (_I, []),
    'ggan_prof_report': (_I, [C.POINTER(ProfRec), _I]),
}

_lib = None


def load():
    """dlopen libggan.so and bind every symbol; raises GganError when anything is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise GganError('libggan.so not found at %s -- run `python -c "import __graft_entry__ as g; g.build()"` '
                        '(there is no CPU fallback)' % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError:
            raise GganError('libggan.so does not export %s' % name)
        fn.restype = res
        fn.argtypes = args
    if lib.ggan_version() != ABI_VERSION:
        raise GganError('libggan.so speaks ABI version %d, this binding %d: a stale build -- run `python -c "import __graft_entry__ as g; g.build()"`'
                        % (lib.ggan_version(), ABI_VERSION))
    _lib = lib
    return lib


def check(rc, what):
    if rc != 0:
        msg = load().ggan_last_error()
        raise GganError('%s failed (%d): %s' % (what, rc, msg.decode() if msg else ''))


def prof_report(cap=256):
    lib = load()
    recs = (ProfRec * cap)()
    n = lib.ggan_prof_report(recs, cap)
    out = []
    for i in range(n):
        r = recs[i]
        out.append(dict(name=r.name.decode(), total_ms=r.total_ms, launches=int(r.launches),
                        flops=r.flops, bytes=r.bytes, grid=int(r.grid)))
    return out
