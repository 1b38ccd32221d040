"""RCCL called directly on the caller's HIP stream, over the communicator torch.distributed's "nccl" (= RCCL) process group already built.

Why not `dist.all_reduce`: every ProcessGroupNCCL collective creates a Work object whose end event the group's WATCHDOG THREAD polls with
hipEventQuery every 100 ms until it reads complete.  On this HIP runtime (7.0 / ROCm 7.2; tools/debug/event_query_probe.py, profiles/r5_event_query_probe.txt)
  * under a global-mode stream capture ANY hipEventQuery from another thread fails and invalidates the capture, and
  * in EVERY capture mode hipEventQuery of an eagerly recorded event fails with hipErrorCapturedEvent -- and invalidates the capture -- when the stream
    the event was last recorded on is capturing NOW (the group's internal stream, or the reducer's, joins the capture of the data-parallel step).
The watchdog turns either into std::terminate: the round-4 driver run lost 191 tests to it (a Work of the eager warm-up step was still in the watchdog's list
when the capture of the next step opened; 2 of 16 runs here).  Enqueuing ncclAllReduce / ncclAllGather / ncclAllToAll ourselves creates no Work, no event
and no watchdog traffic: the collective is one more stream-ordered launch of the step -- eager or captured -- on the stream the reducer chose.
The process group stays what it is good at: rendezvous, communicator construction (xGMI topology), barriers and host-visible reductions outside the step.

Only the library that built the communicator may use it: the handle is looked up among the shared objects already mapped into this process."""
import ctypes
import os

import torch

_DTYPES = {torch.int8: 0, torch.uint8: 1, torch.int32: 2, torch.int64: 4, torch.float16: 6, torch.float32: 7, torch.float64: 8, torch.bfloat16: 9}
SUM, PROD, MAX, MIN = 0, 1, 2, 3
_lib = None


def _loaded_rccl_path():
    try:
        with open("/proc/self/maps") as f:
            for line in f:
                p = line.rsplit(" ", 1)[-1].strip()
                if os.path.basename(p).startswith("librccl.so"):
                    return p
    except OSError:
        pass
    return None


def library_path():
    """the librccl.so this process would bind: the one already mapped (the one that built the group's communicator), else torch's own; None when neither exists"""
    path = _loaded_rccl_path() or os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
    return path if os.path.exists(path) else None


def lib():
    global _lib
    if _lib is None:
        path = library_path()
        if path is None:
            raise RuntimeError("sam_textvqa_amd.rccl: no librccl.so is mapped into this process and none sits next to torch (%s): the direct RCCL transport "
                               "needs the ROCm build of torch; gloo / SAM_RCCL_DIRECT=0 use torch.distributed calls instead" % os.path.join(os.path.dirname(torch.__file__), "lib"))
        l = ctypes.CDLL(path)
        vp, sz, i = ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int
        l.ncclAllReduce.argtypes, l.ncclAllReduce.restype = [vp, vp, sz, i, i, vp, vp], i
        l.ncclAllGather.argtypes, l.ncclAllGather.restype = [vp, vp, sz, i, vp, vp], i
        l.ncclAllToAll.argtypes, l.ncclAllToAll.restype = [vp, vp, sz, i, vp, vp], i
        l.ncclBroadcast.argtypes, l.ncclBroadcast.restype = [vp, vp, sz, i, i, vp, vp], i
        l.ncclCommCount.argtypes, l.ncclCommCount.restype = [vp, ctypes.POINTER(ctypes.c_int)], i
        l.ncclGetErrorString.argtypes, l.ncclGetErrorString.restype = [i], ctypes.c_char_p
        _lib = l
    return _lib


def _check(rc, what):
    if rc != 0:
        raise RuntimeError("%s failed: %s (ncclResult %d)" % (what, lib().ncclGetErrorString(rc).decode(), rc))


def communicator(group=None):
    """the ncclComm_t of `group`'s RCCL backend on the current device, or None when the group is not an RCCL group (gloo: the CPU / shared-GPU tests) or its
    communicator cannot be reached (older torch without _comm_ptr): callers then stay on the process-group API"""
    import torch.distributed as dist
    if not dist.is_initialized() or os.environ.get("SAM_RCCL_DIRECT", "1") == "0":
        return None
    try:
        if dist.get_backend(group) != "nccl":
            return None
        pg = group if group is not None else dist.distributed_c10d._get_default_group()
        backend = pg._get_backend(torch.device("cuda", torch.cuda.current_device()))
        ptr = int(backend._comm_ptr())
        if not ptr:
            return None
        n = ctypes.c_int(0)
        _check(lib().ncclCommCount(ptr, ctypes.byref(n)), "ncclCommCount")
        if n.value != dist.get_world_size(group):
            return None
        return ptr
    except Exception:
        return None


def _stream(stream):
    return (stream if stream is not None else torch.cuda.current_stream()).cuda_stream


def _ok(t):
    if not (t.is_cuda and t.is_contiguous() and t.dtype in _DTYPES):
        raise ValueError("rccl: need a contiguous device tensor of a supported dtype, got %s %s" % (t.dtype, t.device))


def all_reduce(comm, t, op=SUM, stream=None):
    """t := reduce over the ranks of t, in place, enqueued on `stream` (default: torch's current stream)"""
    _ok(t)
    if t.numel():
        _check(lib().ncclAllReduce(t.data_ptr(), t.data_ptr(), t.numel(), _DTYPES[t.dtype], op, comm, _stream(stream)), "ncclAllReduce")


def all_gather(comm, out, t, stream=None):
    """out[r * n : (r + 1) * n] := rank r's t   (n = t.numel(); out holds world * n elements)"""
    _ok(t); _ok(out)
    if out.dtype != t.dtype or out.numel() % max(t.numel(), 1):
        raise ValueError("rccl.all_gather: output must hold a whole number of input-sized pieces of the same dtype")
    if t.numel():
        _check(lib().ncclAllGather(t.data_ptr(), out.data_ptr(), t.numel(), _DTYPES[t.dtype], comm, _stream(stream)), "ncclAllGather")


def all_to_all(comm, out, t, world, stream=None):
    """out[r * n : (r + 1) * n] := rank r's t[me * n : (me + 1) * n]   (n = t.numel() / world)"""
    _ok(t); _ok(out)
    if out.dtype != t.dtype or out.numel() != t.numel() or t.numel() % world:
        raise ValueError("rccl.all_to_all: equal-sized buffers of world * n elements")
    if t.numel():
        _check(lib().ncclAllToAll(t.data_ptr(), out.data_ptr(), t.numel() // world, _DTYPES[t.dtype], comm, _stream(stream)), "ncclAllToAll")


def broadcast(comm, t, root=0, stream=None):
    _ok(t)
    if t.numel():
        _check(lib().ncclBroadcast(t.data_ptr(), t.data_ptr(), t.numel(), _DTYPES[t.dtype], root, comm, _stream(stream)), "ncclBroadcast")
