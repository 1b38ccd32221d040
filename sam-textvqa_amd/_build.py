"""Compile the HIP sources of this package into lib/libsam_hip.so for gfx950 (hipcc cross-compiles
without a GPU).  Sources are hashed so repeated calls are no-ops; the .so stays in-tree so it
travels with the repo snapshot to the GPU box."""
import hashlib
import os
import subprocess

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, "csrc")
LIB_DIR = os.path.join(PKG, "lib")
LIB = os.path.join(LIB_DIR, "libsam_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
# -amdgpu-mfma-vgpr-form: MFMA accumulators that the VALU consumes right away (attention scores) stay in VGPRs; without it the compiler
# put them in AGPRs and copied every value across with v_accvgpr_read/write (80 and 136 copies per loop trip in the two attention
# backward kernels, none left with the flag; the GEMMs never had any)
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value", "-Wno-unused-result", "-mllvm", "-amdgpu-mfma-vgpr-form=1"]


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".cpp")))


def _digest():
    h = hashlib.sha256(" ".join(FLAGS).encode())
    for f in sorted(os.listdir(CSRC)):
        if f.endswith((".hip", ".cpp", ".h")):
            h.update(f.encode())
            h.update(open(os.path.join(CSRC, f), "rb").read())
    inc = os.path.join(os.path.dirname(PKG), "include", "sam_hip.h")
    if os.path.exists(inc):
        h.update(open(inc, "rb").read())
    return h.hexdigest()


def _headers_digest():
    h = hashlib.sha256(" ".join(FLAGS).encode())
    for f in sorted(os.listdir(CSRC)):
        if f.endswith(".h"):
            h.update(f.encode())
            h.update(open(os.path.join(CSRC, f), "rb").read())
    inc = os.path.join(os.path.dirname(PKG), "include", "sam_hip.h")
    if os.path.exists(inc):
        h.update(open(inc, "rb").read())
    return h


def build(force=False, verbose=False):
    """compile what changed (per-object stamps: source + every header + flags), link, stamp the library with the digest of the whole tree.
    Any compile or link error raises: a stale library is never left in place as if it were current."""
    os.makedirs(LIB_DIR, exist_ok=True)
    stamp = os.path.join(LIB_DIR, "libsam_hip.sha256")
    dig = _digest()
    if not force and os.path.exists(LIB) and os.path.exists(stamp) and open(stamp).read().strip() == dig:
        return LIB
    hdr = _headers_digest()
    objs, procs = [], []
    for src in sources():
        obj = os.path.join(LIB_DIR, os.path.basename(src) + ".o")
        objs.append(obj)
        is_runtime = os.path.basename(src) == "runtime.cpp"
        h = hdr.copy()
        h.update(open(src, "rb").read())
        if is_runtime:
            h.update(dig.encode())                # carries the digest string of the whole tree
        odig, ostamp = h.hexdigest(), obj + ".sha256"
        if not force and os.path.exists(obj) and os.path.exists(ostamp) and open(ostamp).read().strip() == odig:
            continue
        cmd = [HIPCC] + FLAGS + (["-x", "hip"] if src.endswith(".cpp") else []) + (['-DSAM_BUILD_DIGEST="%s"' % dig] if is_runtime else []) + \
              ["-I", os.path.join(os.path.dirname(PKG), "include"), "-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd))
        if os.path.exists(ostamp):
            os.remove(ostamp)
        procs.append((src, ostamp, odig, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for src, ostamp, odig, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError("hipcc failed on %s:\n%s" % (src, out.decode(errors="replace")))
        with open(ostamp, "w") as f:
            f.write(odig)
    if os.path.exists(stamp):
        os.remove(stamp)
    cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    if r.returncode != 0:
        raise RuntimeError("link failed:\n%s" % r.stdout.decode(errors="replace"))
    with open(stamp, "w") as f:
        f.write(dig)
    return LIB


# ---- PyTorch-ROCm custom ops (csrc_torch/sam_torch_ops.cpp): TORCH_LIBRARY registration + coarse per-layer entry points over the C ABI ----
TORCH_OPS_SRC = os.path.join(PKG, "csrc_torch", "sam_torch_ops.cpp")
TORCH_OPS_LIB = os.path.join(LIB_DIR, "libsam_torch_ops.so")
CXX = os.environ.get("CXX", "g++")


def build_torch_ops(force=False, verbose=False):
    """g++ against the torch headers (no device code in this file); links libsam_hip.so from its own directory ($ORIGIN)"""
    import torch
    from torch.utils.cpp_extension import include_paths, library_paths
    build()
    h = hashlib.sha256(open(TORCH_OPS_SRC, "rb").read())
    h.update(open(os.path.join(os.path.dirname(PKG), "include", "sam_hip.h"), "rb").read())
    h.update(torch.__version__.encode())
    dig, stamp = h.hexdigest(), TORCH_OPS_LIB + ".sha256"
    if not force and os.path.exists(TORCH_OPS_LIB) and os.path.exists(stamp) and open(stamp).read().strip() == dig:
        return TORCH_OPS_LIB
    rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
    cmd = [CXX, "-O2", "-std=c++17", "-fPIC", "-shared", "-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1",
           "-D_GLIBCXX_USE_CXX11_ABI=%d" % int(torch._C._GLIBCXX_USE_CXX11_ABI), TORCH_OPS_SRC, "-o", TORCH_OPS_LIB,
           "-I", os.path.join(os.path.dirname(PKG), "include"), "-I", os.path.join(rocm, "include")]
    cmd += ["-I" + i for i in include_paths()] + ["-L" + l for l in library_paths()] + ["-L", LIB_DIR, "-lsam_hip", "-Wl,-rpath,$ORIGIN",
                                                                                         "-lc10", "-ltorch_cpu", "-ltorch", "-lc10_hip", "-ltorch_hip"]
    if verbose:
        print(" ".join(cmd))
    if os.path.exists(stamp):
        os.remove(stamp)
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    if r.returncode != 0:
        raise RuntimeError("building the torch ops failed:\n%s" % r.stdout.decode(errors="replace")[-4000:])
    with open(stamp, "w") as f:
        f.write(dig)
    return TORCH_OPS_LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))
    print(build_torch_ops(force=True, verbose=True))
