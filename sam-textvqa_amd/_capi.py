"""ctypes binding of libsam_hip.so — the C-ABI declared in include/sam_hip.h.

The library is the product: if it is missing or a call fails this module raises; nothing here
(or anywhere in the package) falls back to a CPU or eager-PyTorch implementation."""
import ctypes as C
import os

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_PKG, "lib", "libsam_hip.so")

_vp, _i, _i64, _u64, _f, _u = C.c_void_p, C.c_int, C.c_int64, C.c_uint64, C.c_float, C.c_uint

# name -> argtypes (all return int status except where noted)
SIGNATURES = {
    "sam_attn_fwd": [_vp, _vp, _i64, _i64, _i, _i, _i, _i, _f, _f, _u64, _u64, _vp, _vp, _vp, _vp],
    "sam_attn_bwd": [_vp, _vp, _vp, _vp, _i64, _i64, _vp, _i, _i, _i, _i, _f, _f, _vp, _vp, _vp],
    "sam_attn_words_per_row": [_i],
    "sam_mask_bits_prefix_lm": [_vp, _i, _i, _i, _i, _vp, _vp],
    "sam_mask_bits_from_additive": [_vp, _i, _i, _i, _vp, _vp],
    "sam_mask_bits_spatial": [_vp, _vp, _i, _i, _i, _i, _i, _i, _i, _u, _vp, _vp],
    "sam_abi_version": [],
}
NO_STATUS = {"sam_attn_words_per_row", "sam_abi_version"}

_lib = None


class SamHipError(RuntimeError):
    pass


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise SamHipError("libsam_hip.so not built (%s): run `python __graft_entry__.py` or "
                              "sam_textvqa_amd._build.build(); there is no fallback path" % LIB_PATH)
        l = C.CDLL(LIB_PATH)
        l.sam_last_error.restype = C.c_char_p
        for name, args in SIGNATURES.items():
            fn = getattr(l, name)
            fn.argtypes = args
            fn.restype = _i
        _lib = l
    return _lib


def call(name, *args):
    """invoke a status-returning entry point; non-zero -> SamHipError with the library's message"""
    l = lib()
    rc = getattr(l, name)(*args)
    if name not in NO_STATUS and rc != 0:
        raise SamHipError("%s failed (rc=%d): %s" % (name, rc, l.sam_last_error().decode()))
    return rc


def ptr(t):
    """device pointer of a torch tensor (None -> NULL)"""
    return None if t is None else C.c_void_p(t.data_ptr())


def stream_handle():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)
