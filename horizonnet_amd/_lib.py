"""ctypes binding of libhorizonnet_hip.so (the C ABI in include/horizonnet_hip.h).

The library is the product: if it is missing or fails to load this module raises
immediately -- there is deliberately no fallback path.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libhorizonnet_hip.so")

_c = ctypes
_vp, _i, _f, _sz, _i64 = _c.c_void_p, _c.c_int, _c.c_float, _c.c_size_t, _c.c_int64

# name -> (restype, argtypes); mirrors include/horizonnet_hip.h one to one
SIGNATURES = {
    "hn_last_error": (_c.c_char_p, []),
    "hn_abi_version": (_i, []),
    "hn_create": (_i, [_c.POINTER(_vp), _i]),
    "hn_destroy": (_i, [_vp]),
    "hn_bind_tensor": (_i, [_vp, _c.c_char_p, _vp, _i64]),
    "hn_packed_bytes": (_sz, []),
    "hn_pack_weights": (_i, [_vp, _vp, _sz, _vp]),
    "hn_workspace_bytes": (_sz, [_i]),
    "hn_forward": (_i, [_vp, _vp, _i, _i, _vp, _vp, _vp, _sz, _vp]),
    "hn_workspace_pipelined_bytes": (_sz, [_i]),
    "hn_forward_submit": (_i, [_vp, _vp, _i, _i, _vp, _vp, _vp, _sz, _i, _vp]),
    "hn_forward_collect": (_i, [_vp, _i, _vp]),
    "hn_pipelined_status_offset_f32": (_i, [_i, _i, _c.POINTER(_sz)]),
    "hn_lstm_layer_wide": (_i, [_vp, _vp, _vp, _vp, _i, _i, _vp, _vp]),
    "hn_check_status": (_i, [_vp, _vp, _c.POINTER(_i)]),
    "hn_set_option": (_i, [_vp, _c.c_char_p, _i]),
    "hn_set_forward_tap": (_i, [_vp, _c.c_char_p, _vp]),
    "hn_set_profiling": (_i, [_vp, _i]),
    "hn_profile_count": (_i, [_vp]),
    "hn_profile_entry": (_i, [_vp, _i, _c.c_char_p, _i, _c.POINTER(_f), _c.POINTER(_c.c_double)]),
    "hn_packed_bf16_bytes": (_sz, []),
    "hn_pack_weights_bf16": (_i, [_vp, _vp, _sz, _vp]),
    "hn_workspace_bf16_bytes": (_sz, [_i]),
    "hn_forward_bf16": (_i, [_vp, _vp, _i, _i, _vp, _vp, _vp, _sz, _vp]),
    "hn_stem_pool_bf16": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp]),
    "hn_conv2d_nhwc_bf16": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "hn_conv2d_nhwc_bf16_ws": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _sz, _vp]),
    "hn_workspace_bf16_pipelined_bytes": (_sz, [_i]),
    "hn_forward_bf16_submit": (_i, [_vp, _vp, _i, _i, _vp, _vp, _vp, _sz, _i, _vp]),
    "hn_forward_bf16_collect": (_i, [_vp, _i, _vp]),
    "hn_pipelined_status_offset": (_i, [_i, _i, _c.POINTER(_sz)]),
    "hn_lstm_layer_bf16_wide": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _vp, _i, _i, _vp]),
    "hn_lstm_bf16_exchange_bytes": (_sz, []),
    "hn_lstm_layer_bf16": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _vp, _vp, _vp]),
    "hn_lstm_layer_bf16_train": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _vp, _vp, _vp]),
    "hn_lstm_bwd_bf16_exchange_bytes": (_sz, []),
    "hn_lstm_layer_bwd_bf16": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _vp, _vp, _vp]),
    "hn_train_workspace_bytes": (_sz, [_i]),
    "hn_train_forward": (_i, [_vp, _vp, _i, _i, _vp, _vp, _vp, _sz, _f, _f, _f, _c.c_uint64, _vp]),
    "hn_set_bn_eval": (_i, [_vp, _c.c_char_p, _i]),
    "hn_train_backward": (_i, [_vp, _vp, _vp, _i, _vp, _sz, _vp, _f, _f, _c.c_uint64, _vp]),
    "hn_train_backward_segment": (_i, [_vp, _vp, _vp, _i, _vp, _sz, _vp, _f, _f, _c.c_uint64, _i, _vp]),
    "hn_grad_segments": (_i, []),
    "hn_grad_segment_range": (_i, [_i, _c.POINTER(_i64), _c.POINTER(_i64)]),
    "hn_set_train_precision": (_i, [_vp, _i]),
    "hn_grad_floats": (_sz, []),
    "hn_loss_l1_bce": (_i, [_vp, _vp, _c.c_longlong, _vp, _vp, _c.c_longlong, _vp, _vp, _vp, _vp, _vp]),
    "hn_scale2": (_i, [_vp, _c.c_longlong, _vp, _c.c_longlong, _vp, _vp]),
    "hn_probe_mfma": (_i, [_i, _i, _i, _vp, _c.POINTER(_c.c_double), _vp]),
    "hn_adam_step": (_i, [_vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _c.c_longlong, _f, _f, _f, _f, _f, _i, _f, _vp]),
    "hn_train_debug_unit": (_i, [_i, _i, _c.POINTER(_i64)]),
    "hn_train_debug_unit_yh": (_i64, [_i, _i]),
    "hn_train_debug_set": (_i, [_vp, _i, _vp, _vp]),
    "hn_train_debug_set2": (_i, [_vp, _i, _vp, _vp]),
    "hn_grad_offset": (_i64, [_c.c_char_p]),
    "hn_conv2d_dgrad_nhwc": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "hn_conv2d_wgrad_nhwc_bf16": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "hn_conv2d_dgrad_nhwc_bf16": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "hn_conv2d_dgrad_nhwc_bf16g": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "hn_conv2d_wgrad_nhwc": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "hn_pano_stretch": (_i, [_vp, _vp, _c.POINTER(_c.c_double), _c.POINTER(_c.c_double), _i, _i, _i, _i, _vp]),
    "hn_pano_stretch_tables": (_i, [_vp, _vp, _c.POINTER(_c.c_double), _c.POINTER(_c.c_double), _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "hn_labels_rasterise": (_i, [_vp, _i, _i, _i, _i, _i, _i, _c.c_double, _vp, _vp, _vp, _vp]),
    "hn_augment_batch": (_i, [_vp, _i, _c.POINTER(_c.c_int), _vp, _c.POINTER(_c.c_double), _c.POINTER(_c.c_double),
                               _c.POINTER(_c.c_int), _c.POINTER(_c.c_int), _c.POINTER(_c.c_double), _i, _i, _i, _vp]),
    "hn_find_peaks": (_i, [_vp, _i, _i, _i, _f, _i, _vp, _vp, _vp]),
    "hn_vote_scan": (_i, [_vp, _i, _c.c_double, _vp]),
    "hn_interquartile_mean_f32": (_i, [_vp, _vp, _vp, _i, _i, _vp]),
    "hn_layout_fit_batch": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp]),
    "hn_pack_conv_weight": (_i, [_vp, _vp, _i, _i, _i, _i, _vp]),
    "hn_packed_conv_weight_floats": (_sz, [_i, _i, _i, _i]),
    "hn_fold_bn": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp]),
    "hn_conv2d_nhwc": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "hn_stem": (_i, [_vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "hn_stem_pool_f32": (_i, [_vp, _i, _i, _vp, _vp, _vp, _vp, _vp]),
    "hn_upsample_flatten": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "hn_lstm_layer": (_i, [_vp, _vp, _vp, _vp, _i, _i, _vp, _vp]),
    "hn_linear_head": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _vp]),
}

_lib = None


def load():
    """Load (once) and return the ctypes library; raises if it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "libhorizonnet_hip.so not found at %s -- build it with horizonnet_amd/csrc/build.sh "
            "(python -c 'import __graft_entry__ as g; g.build()'); there is no fallback path." % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError here = ABI drift, also fatal
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


class HipEngineError(RuntimeError):
    pass


def check(rc, what):
    if rc != 0:
        msg = load().hn_last_error()
        raise HipEngineError("%s failed (rc=%d): %s" % (what, rc, msg.decode() if msg else "?"))


def ptr(t):
    """Device pointer of a torch tensor (or None)."""
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def stream_ptr(device=None):
    import torch
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def source_fingerprint():
    """SHA-256 over the kernel sources (csrc/*.hip, *.h, build.sh, the public header), in name order.  The counter records under
    profiles/ carry it (tools/merge_pmc*.py) and bench.py compares it with the tree it runs from: `traffic_stale`.  (The hash of the
    built library is NOT reproducible from a fresh build, the sources are.)"""
    import glob
    import hashlib
    root = os.path.dirname(os.path.abspath(__file__))
    files = sorted(glob.glob(os.path.join(root, "csrc", "*.hip")) + glob.glob(os.path.join(root, "csrc", "*.h")) +
                   [os.path.join(root, "csrc", "build.sh"), os.path.join(os.path.dirname(root), "include", "horizonnet_hip.h")])
    h = hashlib.sha256()
    for f in files:
        h.update(os.path.basename(f).encode() + b"\0")
        h.update(open(f, "rb").read())
    return h.hexdigest()

