"""ctypes binding of `libgadapt_hip.so` (C-ABI declared in include/gadapt_hip.h).

There is no CPU fallback: if the library is missing, or a call fails, this raises.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('GADAPT_LIB') or os.path.join(_HERE, 'libgadapt_hip.so')   # override: A/B builds only

SUPPORTED_HIDDEN = (4, 8, 16, 32, 64, 128)


class GadaptGraph(C.Structure):
    _fields_ = [('n_nodes', C.c_int32), ('n_edges', C.c_int32),
                ('rowptr_t', C.c_void_p), ('col_t', C.c_void_p),
                ('rowptr_s', C.c_void_p), ('col_s', C.c_void_p), ('perm_s', C.c_void_p), ('tpos_s', C.c_void_p),
                ('meta_t', C.c_void_p * 3), ('meta_s', C.c_void_p * 3),
                ('ell_t', C.c_void_p), ('ell_s', C.c_void_p), ('wide_deg_t', C.c_int32), ('wide_deg_s', C.c_int32),
                ('wide_big_deg_t', C.c_int32), ('wide_half_deg_t', C.c_int32)]


TILE_HEIGHTS = (64, 128, 256)


_P, _I, _L, _F = C.c_void_p, C.c_int, C.c_int64, C.c_float
_G = C.POINTER(GadaptGraph)

# name -> (restype, argtypes); must list every symbol include/gadapt_hip.h declares
PROTOTYPES = {
    'gadapt_supported_hidden_dim': (_I, [_I]),
    'gadapt_last_error': (C.c_char_p, []),
    'gadapt_abi_version': (_I, []),
    'gadapt_clear_error': (_I, []),
    'gadapt_csr_build_host': (_I, [_P, _P, _L, _L, _P, _P, _P, _P, _P, _P, _P]),
    'gadapt_tile_meta_host': (_I, [_P, _P, _L, _I, _P]),
    'gadapt_ell_build_host': (_I, [_P, _P, _L, _P, _P]),
    'gadapt_wide_window_host': (_I, [_P, _P, _L, _I, _I, _I, _P]),
    'gadapt_coeffs_forward': (_I, [_P, _P, _P, _P, _P, _I, _P]),
    'gadapt_coeffs_backward': (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _P]),
    'gadapt_encode_linear': (_I, [_P, _P, _P, _P, _L, _I, _I, _P]),
    'gadapt_encode_features': (_I, [_P, _I, _P, _P, _P, _P, _P, _L, _I, _P]),
    'gadapt_encode_features_coeffs': (_I, [_P, _I, _P, _P, _P, _P, _P, _L, _I, _P, _P, _P, _P, _P, _I, _P]),
    'gadapt_slab_reduce_coeffs_backward': (_I, [_P, _I, _P, _P, _P, _P, _P, _P, _P, _P, _I, _P, _P, _I, _I, _P]),
    'gadapt_spmm': (_I, [_G, _I, _P, _P, _P, _I, _F, _P]),
    'gadapt_sddmm': (_I, [_G, _I, _P, _P, _P, _I, _F, _P]),
    'gadapt_edge_softmax_forward': (_I, [_G, _P, _P, _P]),
    'gadapt_edge_softmax_backward': (_I, [_G, _P, _P, _P, _P]),
    'gadapt_edge_combine': (_I, [_G, _P, _P, _P, _I, _P]),
    'gadapt_edge_rowsum': (_I, [_G, _I, _P, _P, _P]),
    'gadapt_edge_vector_op': (_I, [_G, _I, _P, _P, _P, _P, _I, _P]),
    'gadapt_loss_forward': (_I, [_P, _L, _P, _L, _I, _I, _P, _P, _P, _P]),
    'gadapt_loss_scratch_floats': (_I, []),
    'gadapt_layer_forward': (_I, [_G, _P, _P, _P, _P, _P, _P, _I, _I, _P]),
    'gadapt_backward_slab_rows': (_I, [_L, _I]),
    'gadapt_backward_slab_floats': (_L, [_L, _I]),
    'gadapt_layer_backward': (_I, [_G, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _P, _P, _I, _I, _P]),
    'gadapt_slab_reduce': (_I, [_P, _I, _P, _P, _P, _I, _P]),
    'gadapt_block_forward': (_I, [_G, _P, _I, _I, _P, _L, _P, _L, _P, _P, _P, _I, _P]),
    'gadapt_block_backward': (_I, [_G, _P, _I, _P, _P, _I, _I, _P, _L, _P, _L, _P, _P, _P, _P, _P, _P, _I, _P, _I, _P]),
    'gadapt_loss_partials_max': (_I, []),
    'gadapt_forward_computes_coeffs': (_I, [_G, _I]),
    'gadapt_block_forward_loss': (_I, [_G, _P, _P, _I, _P, _P, _I, _P, _P, _P, _P, _P, _P, _P, _I, _I, _P, _P, _I, _P]),
    'gadapt_step_tail': (_I, [_P, _I, _P, _P, _P, _P, _P, _F, _F, _F, _F, _F, _P, _F, _P, _P, _P, _I, _P, _L, _I, _P]),
    'gadapt_small_forward_lds_bytes': (_L, [_I, _I, _I]),
    'gadapt_small_forward': (_I, [_G, _P, _P, _I, _I, _I, _P, _I, _P, _P, _P, _I, _P, _P, _P, _L, _L, _P, _I, _P, _I, _P, _P, _I, _P]),
    'gadapt_small_forward_loss': (_I, [_G, _P, _P, _I, _I, _I, _P, _I, _P, _P, _P, _I, _P, _P, _P, _L, _L, _P, _I, _P, _I, _P, _P, _P, _I, _P, _P, _I, _P]),
    'gadapt_small_backward_lds_bytes': (_L, [_I, _I, _I]),
    'gadapt_small_backward': (_I, [_G, _P, _P, _I, _I, _I, _P, _P, _P, _I, _P, _P, _P, _L, _L, _P, _I, _P, _I, _P]),
    'gadapt_layer_params_reduce': (_I, [_P, _I, _I, _I, _P, _P]),
    'gadapt_mesh_loss_seed': (_I, [_P, _P, _P, _P, _P, _L, _I, _I, _I, _F, _P]),
    'gadapt_pad_columns': (_I, [_P, _P, _L, _I, _I, _P]),
    'gadapt_adam_step': (_I, [_P, _P, _P, _P, _L, _F, _F, _F, _F, _F, _I, _F, _P]),
    'gadapt_adam_step_dev': (_I, [_P, _P, _P, _P, _L, _F, _F, _F, _F, _F, _P, _F, _P]),
    'gadapt_allreduce_flat': (_I, [_P, _P, _L, _I, _P]),
    'gadapt_gat_plus_block_forward': (_I, [_G, _P, _I, _P, _P, _L, _F, _I, _I, _I, _P, _P, _I, _P]),
    'gadapt_gat_plus_block_backward': (_I, [_G, _P, _P, _P, _P, _I, _P, _P, _L, _F, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _I, _P]),
    'gadapt_gat_plus_partial_rows': (_I, [_L, _I]),
    'gadapt_gather_fields': (_I, [_I, _P, _P, _P, _P, _I, _P]),
    'gadapt_profile_enable': (_I, [_I]),
    'gadapt_profile_read': (_I, [_I, C.POINTER(C.c_double), C.POINTER(C.c_int)]),
    'gadapt_profile_samples': (_I, [_I, C.POINTER(C.c_double), _I]),
    'gadapt_profile_variants': (_I, [_I, C.POINTER(C.c_int), _I]),
    'gadapt_profile_reset': (_I, []),
    'gadapt_profile_calibrate': (_I, [_I, _P]),
    'gadapt_debug_set_backward_inplace': (_I, [_I]),
    'gadapt_debug_occupancy': (_I, [_I, C.POINTER(C.c_int)]),
}

_lib = None


class NativeError(RuntimeError):
    pass


def lib():
    """The loaded library; raises if it has not been built (python -c 'import __graft_entry__ as g; g.build()')."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise NativeError(f"{LIB_PATH} not found: build it with `make` (hipcc --offload-arch=gfx950); "
                              "there is no CPU fallback for the message-passing path")
        handle = C.CDLL(LIB_PATH)
        for name, (res, args) in PROTOTYPES.items():
            fn = getattr(handle, name)            # AttributeError if the symbol is missing
            fn.restype, fn.argtypes = res, args
        _lib = handle
    return _lib


def check(rc: int, what: str):
    if rc != 0:
        msg = lib().gadapt_last_error().decode() or f"error {rc}"
        raise NativeError(f"{what}: {msg} (code {rc})")


def clear_error() -> int:
    """Drop a stale per-thread error (see gadapt_clear_error): call after recovering from a failed hipGraph capture."""
    return int(lib().gadapt_clear_error())


def ptr(t):
    """Device/host address of a contiguous tensor (None -> NULL)."""
    if t is None:
        return None
    assert t.is_contiguous(), "native entry points take contiguous tensors"
    return t.data_ptr()


def current_stream(device) -> int:
    import torch
    return torch.cuda.current_stream(device).cuda_stream
