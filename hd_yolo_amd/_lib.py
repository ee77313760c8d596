"""ctypes binding of libhdyolo_hip.so (include/hdyolo.h).

The product path has no CPU fallback: if the library is missing or a call fails this module raises.
"""
import ctypes
import os
from ctypes import c_char_p, c_float, c_int, c_longlong, c_size_t, c_void_p

import torch  # noqa: F401  -- must come first: the library has to bind to the HIP runtime torch already loaded,
#                              otherwise a second libamdhip64 gets mapped and sees no device

_HERE = os.path.dirname(os.path.abspath(__file__))
# HDY_LIB: load another build of the same ABI (kernel A/B experiments); it must still sit under csrc/build/
LIB_PATH = os.path.join(_HERE, 'csrc', 'build', os.path.basename(os.environ.get('HDY_LIB', 'libhdyolo_hip.so')))

ABI_VERSION = 6                   # = HDY_ABI_VERSION of the include/hdyolo.h that SIGNATURES below was written for (tests/test_abi.py holds the two together)
F32, BF16 = 0, 1
OK, EINVAL, EUNSUPPORTED = 0, -1, -2      # status codes (include/hdyolo.h); positive = hipError_t
PACK_FWD, PACK_DGRAD, PACK_STEM = 0, 1, 2
ACT_NONE, ACT_SILU, ACT_RELU = 0, 1, 2

_P, _I, _L, _F, _Z = c_void_p, c_int, c_longlong, c_float, c_size_t

# name -> (restype, argtypes); must list every symbol include/hdyolo.h declares (tests check this)
SIGNATURES = {
    'hdy_last_error': (c_char_p, []),
    'hdy_last_dispatch': (c_char_p, []),
    'hdy_dispatch_log': (c_char_p, []),
    'hdy_dispatch_log_reset': (None, []),
    'hdy_set_option': (_I, [c_char_p, _I]),
    'hdy_get_option': (_I, [c_char_p]),
    'hdy_conv_wgrad_stem_fused_ok': (_I, [_I, _I, _I, _I]),
    'hdy_conv_wgrad_stem_fused': (_I, [_P, _P, _I, _P, _I] + [_P] * 6 + [_I, _I, _I, _I, _P, _I, _P, _I, _I, _P, _Z, _P]),
    'hdy_bn_slab_sums': (_I, [_P, _I, _I, _I, _L, _P, _P]),
    'hdy_bn_finalize_sums': (_I, [_P, _I, _P, _I, _I] + [_P] * 8 + [_F, _F] + [_P] * 5),
    'hdy_bn_bwd_coeffs_sums': (_I, [_P, _I, _P, _P, _P]),
    'hdy_sgd_blocks': (_I, [_L]),
    'hdy_sgd_step': (_I, [_P, _I, _I, _P, _P, _P, _P, _I, _I, _P]),
    'hdy_version': (_I, []),
    'hdy_fastdiv_magic': (_I, [ctypes.c_uint, _P, _P]),
    'hdy_conv_out_dim': (_I, [_I, _I, _I, _I]),
    'hdy_conv_mtiles': (_I, [_L]),
    'hdy_conv_stat_slabs': (_I, [_I] * 10),
    'hdy_conv_pack_elems': (_Z, [_I, _I, _I, _I, _I, _I, _I, _I]),
    'hdy_conv_pack': (_I, [_P, _I, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P, _P]),
    'hdy_conv_pack_describe': (_I, [_P, _I, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P, _P, _I]),
    'hdy_conv_pack_run': (_I, [_P, _I, _I, _P]),
    'hdy_conv_fwd': (_I, [_P, _I, _P, _P, _P, _P, _I, _P, _I, _P] + [_I] * 15 + [_P]),
    'hdy_conv_dgrad': (_I, [_P, _I, _P, _P, _I] + [_I] * 11 + [_P]),
    'hdy_conv_wgrad_workspace_bytes': (_Z, [_I] * 11),
    'hdy_conv_wgrad': (_I, [_P, _I, _P, _I] + [_I] * 9 + [_P, _I, _P, _I, _I, _P, _Z, _I, _I, _P]),
    'hdy_conv_dgrad_stat_slabs': (_I, [_I] * 10),
    'hdy_conv_dgrad_stats': (_I, [_P, _I, _P, _P, _I] + [_I] * 11 + [_P, _I, _P]),
    'hdy_conv1x1_bwd_fused_stats': (_I, [_P, _I, _P, _I, _I, _P, _I, _P, _P, _P, _P, _P, _P, _P, _I, _P, _P, _I, _I, _P, _I, _P, _I, _I, _L, _I, _I, _P, _Z, _I, _P, _I, _P]),
    'hdy_bn_bwd_finalize_slabs': (_I, [_P, _I, _I, _L, _P, _P, _P, _P, _I, _P, _P, _P]),
    'hdy_conv1x1_bwd_fused_stat_slabs': (_I, [_L, _I, _I, _I]),
    'hdy_bn_act_bwd_apply': (_I, [_P, _I, _P, _I, _I, _P, _I, _P, _P, _P, _P, _P, _P, _P, _I, _L, _I, _I, _I, _P]),
    'hdy_conv1x1_bwd_fused_ok': (_I, [_I, _I, _I]),
    'hdy_conv1x1_bwd_fused_grid': (_I, [_L, _I]),
    'hdy_conv1x1_bwd_fused_workspace_bytes': (_Z, [_L, _I, _I]),
    'hdy_exec_op': (_I, [c_char_p]),
    'hdy_exec_run': (_I, [_P, _Z, _P, _P]),
    'hdy_exec_join': (_I, [ctypes.c_ulonglong, _P]),
    'hdy_copy_f32': (_I, [_P, _P, _L, _P]),
    'hdy_conv1x1_bwd_fused': (_I, [_P, _I, _P, _I, _I, _P, _I, _P, _P, _P, _P, _P, _P, _P, _I, _P, _P, _I, _I, _P, _I, _P, _I, _I, _L, _I, _I, _P, _Z, _I, _P]),
    'hdy_groupnorm_workspace_floats': (_Z, [_I, _I]),
    'hdy_groupnorm_fwd': (_I, [_P, _I, _P, _P, _P, _I, _P, _P, _I, _I, _I, _I, _F, _I, _I, _P, _Z, _P]),
    'hdy_groupnorm_bwd': (_I, [_P, _I, _P, _I, _P, _P, _P, _P, _I, _P, _P, _I, _P, _I, _I, _I, _I, _I, _I, _P, _Z, _P]),
    'hdy_bilinear_fwd': (_I, [_P, _I, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    'hdy_bilinear_bwd': (_I, [_P, _I, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    'hdy_bilinear_bwd_axis': (_I, [_P, _I, _P, _I, _L, _I, _I, _I, _I, _I, _I, _P]),
    'hdy_softdice_workspace_floats': (_Z, [_I, _I]),
    'hdy_softdice': (_I, [_P, _I, _P, _P, _I, _I, _I, _P, _P, _P, _I, _P, _Z, _P]),
    'hdy_softmax2d': (_I, [_P, _I, _P, _I, _L, _I, _P]),
    'hdy_bn_finalize_workspace_bytes': (_Z, [_I, _I]),
    'hdy_bn_finalize': (_I, [_P, _I, _I, _I, _L, _P, _P, _P, _P, _F, _F, _P, _P, _P, _P, _P, _Z, _P]),
    'hdy_bn_eval_coeffs': (_I, [_P, _P, _P, _P, _F, _I, _P, _P, _P]),
    'hdy_bn_eval_coeffs_batch': (_I, [_P, _I, _P]),
    'hdy_bn_act_fwd': (_I, [_P, _I, _P, _P, _P, _I, _P, _I, _L, _I, _I, _I, _P]),
    'hdy_bn_bwd_blocks': (_I, [_L]),
    'hdy_bn_bwd_workspace_bytes': (_Z, [_L, _I]),
    'hdy_colsum_workspace_bytes': (_Z, [_L, _I]),
    'hdy_bn_finalize_pair': (_I, [_P, _I, _I, _I, _I, _L] + [_P] * 8 + [_F, _F, _P, _P, _P, _P, _P, _Z, _P]),
    'hdy_bn_act_fwd_pair': (_I, [_P, _I, _P, _P, _P, _I, _P, _I, _I, _L, _I, _I, _I, _P]),
    'hdy_bn_act_bwd_pair': (_I, [_P, _I, _P, _I, _I, _P, _I, _P, _P, _P, _P, _P, _I, _P, _P, _P, _P, _I, _L, _I, _I, _I, _P, _Z, _P]),
    'hdy_bn_act_bwd': (_I, [_P, _I, _P, _I, _P, _P, _P, _P, _P, _I, _P, _P, _I, _L, _I, _I, _I, _P, _Z, _P]),
    'hdy_add_inplace': (_I, [_P, _I, _P, _I, _L, _I, _I, _P]),
    'hdy_colsum': (_I, [_P, _I, _L, _I, _P, _I, _I, _P, _Z, _P]),
    'hdy_det_grad_pack': (_I, [_P, _L, _L, _L, _L, _L, _P, _I, _I, _I, _I, _I, _I, _I, _P]),
    'hdy_det_targets': (_I, [_P, _P, _P, _I, _I, _P, _P, _P]),
    'hdy_det_loss_workspace_bytes': (_Z, [_I, _P, _P, _I, _I, _I, _I]),
    'hdy_det_loss': (_I, [_P, _I, _P, _I, _I, _P, _P, _I, _I, _I, _I, _P, _P, _P, _P, _I, _P, _F, _F, _F, _F, _F, _F, _F, _P, _P, _Z, _P]),
    'hdy_scale_inplace': (_I, [_P, _L, _P, _I, _P]),
    'hdy_sppf_pool_fwd': (_I, [_P, _P, _P, _P, _I, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    'hdy_sppf_pool_bwd': (_I, [_P, _P, _P, _P, _I, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    'hdy_upsample2x_fwd': (_I, [_P, _I, _P, _I, _I, _I, _I, _I, _I, _P]),
    'hdy_upsample2x_bwd': (_I, [_P, _I, _P, _I, _I, _I, _I, _I, _I, _I, _P]),
    'hdy_stem_prep': (_I, [_P, _P, _I, _I, _I, _I, _I, _P]),
    'hdy_nchw_to_nhwc': (_I, [_P, _P, _I, _I, _I, _I, _I, _I, _P]),
    'hdy_decode': (_I, [_P, _L, _L, _L, _L, _P, _F, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    'hdy_nms_workspace_bytes': (_Z, [_I, _I]),
    'hdy_nms_workspace_bytes_for': (_Z, [_I, _I, _I]),
    'hdy_mask_select': (_I, [_P, _I, _P, _P, _I, _I, _I, _I, _P, _P, _P, _P, _I, _F, _F, _P, _P, _P, _P, _P, _Z, _P]),
    'hdy_softdice_wgrad': (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _P, _P, _P, _Z, _P]),
    'hdy_det_outputs': (_I, [_P, _P, _P, _I, _I, _I, _P, _I, _F, _I, _P, _P, _P, _P, _P]),
    'hdy_nms_batched': (_I, [_P, _I, _I, _I, _I, _F, _F, _I, _F, _I, _P, _P, _P, _P, _P, _P, _P, _P, _Z, _P]),
    'hdy_nms_boxes': (_I, [_P, _I, _I, _F, _I, _P, _P, _P, _Z, _P]),
    'hdy_roi_align_fwd': (_I, [_P, _I, _I, _I, _I, _I, _P, _I, _F, _I, _I, _I, _P, _I, _P]),
    'hdy_roi_align_bwd': (_I, [_P, _P, _I, _I, _I, _I, _P, _I, _F, _I, _I, _I, _I, _P]),
    'hdy_relu_bwd': (_I, [_P, _P, _P, _L, _I, _P]),
    'hdy_cast_store': (_I, [_P, _P, _I, _L, _I, _I, _I, _P]),
}



class PackDesc(ctypes.Structure):
    """mirror of hdy_pack_desc (include/hdyolo.h)"""
    _fields_ = [('w_a', c_void_p), ('w_b', c_void_p), ('out', c_void_p)] + [(n, c_int) for n in (
        'K_a', 'K_b', 'Kl', 'C', 'R', 'S', 'transpose', 'TH', 'TW', 'rbase', 'rstep', 'sbase', 'sstep', 'stem', 'rows_total', 'Kdp',
        'dtype', 'first_block', 'nblocks', 'pad_')]


class BnEvalDesc(ctypes.Structure):
    """mirror of hdy_bn_eval_desc (include/hdyolo.h)"""
    _fields_ = [(n, c_void_p) for n in ('gamma', 'beta', 'running_mean', 'running_var', 'scale', 'shift')] + [('K', c_int), ('eps', c_float)]


class SgdDesc(ctypes.Structure):
    """mirror of hdy_sgd_desc (include/hdyolo.h)"""
    _fields_ = [('p', c_void_p), ('g', c_void_p), ('buf', c_void_p), ('n', c_longlong), ('group', c_int), ('first', c_int), ('first_block', c_int),
                ('pad_', c_int)]


class StatReq(ctypes.Structure):
    """mirror of hdy_stat_req (include/hdyolo.h)"""
    _fields_ = [('y', c_void_p), ('ldy', c_int)] + [(n, c_void_p) for n in ('scale', 'shift', 'slabs')] + \
               [('c0', c_int), ('c1', c_int), ('act', c_int), ('nslabs', c_int)]


_lib = None


class HdyError(RuntimeError):
    pass


def load():
    """Load the shared library (once).  Raises HdyError with build instructions if it is missing."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise HdyError(f'{LIB_PATH} is missing: run `python -m hd_yolo_amd.build` (hipcc, gfx950). '
                           'hd_yolo_amd has no CPU fallback.')
        lib = ctypes.CDLL(LIB_PATH)
        lib.hdy_version.restype, lib.hdy_version.argtypes = _I, []
        got = lib.hdy_version()
        if got != ABI_VERSION:
            # SIGNATURES below is this revision's parameter lists: an older / newer build would receive shifted arguments instead of an error
            raise HdyError(f'{LIB_PATH} was built for ABI revision {got}, this binding is written for {ABI_VERSION} (include/hdyolo.h HDY_ABI_VERSION): '
                           'rebuild with `python -m hd_yolo_amd.build --force`')
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = res, args
        _lib = lib
    return _lib


def call(name, *args):
    """Call an int-returning entry point; raise HdyError(hdy_last_error()) on a non-zero status."""
    lib = load()
    rc = getattr(lib, name)(*args)
    if rc != 0:
        raise HdyError(f'{name} failed (status {rc}): {lib.hdy_last_error().decode()}')


def query(name, *args):
    return getattr(load(), name)(*args)


# ---- kernel selection: switches and the dispatch log (include/hdyolo.h) -----------------------------------------------------------
class option:
    """`with _lib.option('HDY_NO_CONV3X3', 1): ...` — a process-wide switch for the duration of the block (set it BEFORE sizing buffers or
    building plans: the switches also steer the slab / workspace queries)."""

    def __init__(self, name, value):
        self.name, self.value = name.encode(), int(value)

    def __enter__(self):
        self.prev = load().hdy_set_option(self.name, self.value)
        if self.prev < 0 and load().hdy_get_option(self.name) != self.value:
            raise HdyError(f'unknown option {self.name.decode()}')
        return self

    def __exit__(self, *exc):
        load().hdy_set_option(self.name, self.prev)
        return False


def dispatch_log(reset=False):
    """kernel families picked (by any thread) since the last reset, in launch order"""
    lib = load()
    names = [n for n in lib.hdy_dispatch_log().decode().split(';') if n]
    if reset:
        lib.hdy_dispatch_log_reset()
    return names
