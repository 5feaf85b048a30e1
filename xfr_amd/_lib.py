"""ctypes binding of libxfr_amd.so (C ABI: include/xfr_amd.h).  Fails loudly; there is no fallback path."""
import ctypes
import os

from .program import OpDesc

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('XFR_AMD_LIB') or os.path.join(_HERE, 'csrc', 'libxfr_amd.so')      # XFR_AMD_LIB: A/B builds of the same ABI (tools/ab_env.sh)

XFR_OK, XFR_INVALID_ARG, XFR_UNSUPPORTED_LAYER, XFR_OOM, XFR_HIP_ERROR, XFR_STATE_ERROR, XFR_RCCL_ERROR = range(7)
ABI_VERSION = 6


class TensorView(ctypes.Structure):
    _fields_ = [('data', ctypes.c_void_p), ('numel', ctypes.c_int64)]


class U8Preprocess(ctypes.Structure):
    _fields_ = [('kind', ctypes.c_int32), ('channels', ctypes.c_int32), ('mean', ctypes.c_double * 4), ('weight', ctypes.c_double * 4)]


class XfrError(RuntimeError):
    def __init__(self, status, msg):
        RuntimeError.__init__(self, msg)
        self.status = status


# every symbol include/xfr_amd.h declares: (name, restype, argtypes)
_P = ctypes.c_void_p
_I = ctypes.c_int32
_F = ctypes.c_float
SYMBOLS = [
    ('xfr_abi_version', _I, []),
    ('xfr_last_error', ctypes.c_char_p, []),
    ('xfr_engine_create', _I, [ctypes.POINTER(OpDesc), _I, _I, _I, _I, _I, _I, _I, ctypes.POINTER(_P)]),
    ('xfr_engine_destroy', _I, [_P]),
    ('xfr_engine_load_weights', _I, [_P, ctypes.POINTER(TensorView), _I]),
    ('xfr_engine_weight_arena', _I, [_P, ctypes.POINTER(_P), ctypes.POINTER(ctypes.c_size_t)]),
    ('xfr_engine_mark_weights_loaded', _I, [_P]),
    ('xfr_comm_unique_id', _I, [_P]),
    ('xfr_comm_init', _I, [_I, _I, _P, _I, ctypes.POINTER(_P)]),
    ('xfr_broadcast_weights', _I, [_P, _P, _I, _P]),
    ('xfr_comm_destroy', _I, [_P]),
    ('xfr_engine_set_mode', _I, [_P, _I, _F, _I]),
    ('xfr_engine_tensor_shape', _I, [_P, _I, ctypes.POINTER(_I), ctypes.POINTER(_I), ctypes.POINTER(_I)]),
    ('xfr_forward', _I, [_P, _P, _I, _I, _P, _P]),
    ('xfr_ebp', _I, [_P, _P, _I, _I, _I, _P, _P, _P, _P]),
    ('xfr_contrastive', _I, [_P, _P, _I, _I, _P, _F, _P, _P]),
    ('xfr_contrastive_raw', _I, [_P, _P, _I, _I, _P, _F, _P, _P]),
    ('xfr_triplet_contrastive', _I, [_P, _P, _P, _I, _I, _F, _F, _P, _P, _I]),
    ('xfr_engine_set_u8_preprocess', _I, [_P, ctypes.POINTER(U8Preprocess)]),
    ('xfr_forward_u8', _I, [_P, _P, _I, _I, _P, _P]),
    ('xfr_triplet_contrastive_u8', _I, [_P, _P, _P, _I, _I, _F, _F, _P, _P, _I]),
    ('xfr_triplet_contrastive_u8_host', _I, [_P, _P, _P, _I, _I, _F, _F, _P, _P]),
    ('xfr_engine_wait_inputs_copied', _I, [_P]),
    ('xfr_debug_u8_preprocess', _I, [_P, _P, _I, _P, _P]),
    ('xfr_engine_set_pipeline', _I, [_P, _I]),
    ('xfr_engine_set_inputs_ready', _I, [_P, _I]),
    ('xfr_engine_set_tail_balance', _I, [_P, _I]),
    ('xfr_engine_set_lean', _I, [_P, _I]),
    ('xfr_engine_set_split_gemm', _I, [_P, _I]),
    ('xfr_engine_split_gemm_stats', _I, [_P, ctypes.POINTER(ctypes.c_int64)]),
    ('xfr_engine_lean_stats', _I, [_P, ctypes.POINTER(ctypes.c_int64)]),
    ('xfr_engine_set_forward_split', _I, [_P, _I]),
    ('xfr_engine_hold_forward', _I, [_P, _I]),
    ('xfr_engine_set_epilogue_fusion', _I, [_P, _I]),
    ('xfr_mwp_to_saliency', _I, [_P, _P, _I, _I, _I, _P, _P]),
    ('xfr_firing_count', _I, [_P, _I, ctypes.POINTER(_I)]),
    ('xfr_firing_kinds', _I, [_P, _I, ctypes.POINTER(_I), _I]),
    ('xfr_subtree_weights', _I, [_P, _P, _I, _I, _P, _I, ctypes.POINTER(_F), ctypes.POINTER(_I), _I, _P]),
    ('xfr_ebp_capture', _I, [_P, _P, _I, _I, _P, ctypes.POINTER(_I), ctypes.POINTER(_F), _I, _P]),
    ('xfr_layerwise_ebp', _I, [_P, _P, _I, _I, _I, ctypes.POINTER(_I), ctypes.POINTER(_I), ctypes.POINTER(_F), _P, _P, _P]),
    ('xfr_ebp_store_firing', _I, [_P, _P, _I, _I, _P, _I, _P, ctypes.POINTER(_I), ctypes.POINTER(_I), ctypes.POINTER(_I), _P]),
    ('xfr_engine_set_trace', _I, [_P, _I]),
    ('xfr_engine_trace_size', _I, [_P, ctypes.POINTER(_I)]),
    ('xfr_engine_get_trace', _I, [_P, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(_I), _I]),
    ('xfr_debug_conv', _I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, ctypes.POINTER(_F)]),
    ('xfr_debug_conv_stamps', _I, [_P, _I]),
    ('xfr_debug_conv_log', _I, [_P, _I, ctypes.c_char_p]),
    ('xfr_engine_memory', _I, [_P, ctypes.POINTER(ctypes.c_size_t), ctypes.POINTER(ctypes.c_size_t)]),
    ('xfr_engine_set_profile', _I, [_P, _I]),
    ('xfr_engine_get_profile', _I, [_P, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_int64),
                                    ctypes.POINTER(ctypes.c_double)]),
    ('xfr_engine_get_profile_by_kernel', _I, [_P, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_double)]),
    ('xfr_engine_profile_csv', _I, [_P, ctypes.c_char_p]),
    ('xfr_chain_epilogue_stats', _I, [ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(_I)]),
    ('xfr_plan_describe', _I, [ctypes.POINTER(OpDesc), _I, _I, _I, _I, _I, _I, _I, _I, ctypes.c_char_p, ctypes.c_size_t,
                               ctypes.POINTER(ctypes.c_size_t)]),
]

_lib = None


def load():
    """Load the HIP library.  Raises if it has not been built: the product has no other compute path."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError('xfr_amd: %s is missing -- build it with `python -c "import __graft_entry__ as g; g.build()"` '
                          'or `make -C xfr_amd/csrc`.  There is no CPU fallback.' % LIB_PATH)
    # torch first: PyTorch-ROCm ships its own libamdhip64; loaded after ours, the process would hold two HIP runtimes (ours resolved against
    # /opt/rocm's) and the engine would see no device through the second one
    import torch  # noqa: F401
    lib = ctypes.CDLL(LIB_PATH)
    for name, restype, argtypes in SYMBOLS:
        fn = getattr(lib, name)      # AttributeError if the ABI symbol is absent
        fn.restype = restype
        fn.argtypes = argtypes
    v = lib.xfr_abi_version()
    if v != ABI_VERSION:
        raise ImportError('xfr_amd: ABI version mismatch (library %d, binding %d)' % (v, ABI_VERSION))
    _lib = lib
    return lib


def check(status):
    if status == XFR_OK:
        return
    msg = load().xfr_last_error().decode('utf-8', 'replace')
    if status == XFR_INVALID_ARG:
        raise ValueError(msg)
    if status == XFR_UNSUPPORTED_LAYER:
        raise ValueError(msg)
    if status == XFR_OOM:
        raise MemoryError(msg)
    raise XfrError(status, msg)
