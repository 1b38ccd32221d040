"""torch.ops.sam_hip.*: the PyTorch-ROCm custom-op layer over the C ABI (csrc_torch/sam_torch_ops.cpp).

`ns()` builds (if the sources changed) and loads lib/libsam_torch_ops.so and returns `torch.ops.sam_hip`.  The fine-grained ops mirror the
C entry points one to one; `encoder_layer_fwd/_bwd` enqueue a whole encoder layer from C++ (autograd.EncoderLayerFn uses them)."""
import os

import torch

from . import _capi as capi

_ns = None


def ns():
    global _ns
    if _ns is None:
        capi.lib()                     # libsam_hip.so first (and torch's own libamdhip64 before it, see _capi.lib)
        from . import _build
        try:
            path = _build.build_torch_ops()
        except Exception as e:
            raise capi.SamHipError("libsam_torch_ops.so could not be built (%s); run `python __graft_entry__.py`" % e)
        torch.ops.load_library(path)
        _ns = torch.ops.sam_hip
    return _ns


def enabled():
    """SAM_COARSE_OPS=0 routes the encoder layers through the per-kernel ctypes calls again (A/B, per-kernel event profiling in bench.py)"""
    return os.environ.get("SAM_COARSE_OPS", "1") != "0"
