"""build helper for tests/cabi/host_smoke.cpp: a C++ host program that uses ONLY include/sam_hip.h + libsam_hip.so (no Python / torch)"""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def build_host_smoke(out_dir):
    import sam_textvqa_amd._build as b
    lib = b.build()
    exe = os.path.join(str(out_dir), "host_smoke")
    cmd = [b.HIPCC, "--offload-arch=gfx950", "-O2", "-std=c++17", "-Wno-unused-result", "-I", os.path.join(ROOT, "include"),
           os.path.join(ROOT, "tests", "cabi", "host_smoke.cpp"), "-L", os.path.dirname(lib), "-lsam_hip", "-Wl,-rpath," + os.path.dirname(lib), "-o", exe]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, r.stdout[-3000:]
    return exe
