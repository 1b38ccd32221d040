"""what shader clock does the device run at under (a) a chain of small kernels, (b) back-to-back large GEMMs?  (rocm-smi polled from a thread)"""
import os, subprocess, sys, threading, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from sam_textvqa_amd import ops, _capi as capi
def poll(stop, out):
    while not stop.is_set():
        try:
            r = subprocess.run(["rocm-smi", "--showclocks"], capture_output=True, text=True, timeout=10).stdout
            out.append([l.strip() for l in r.splitlines() if "sclk" in l or "mclk" in l or "fclk" in l])
        except Exception as e:
            out.append([repr(e)])
        time.sleep(0.3)
def rnd(*s): return torch.randn(*s, device="cuda").to(torch.bfloat16)
a_s, b_s = rnd(1280, 768), rnd(2304, 768)
a_l, b_l = rnd(11648, 3072), rnd(768, 3072)
x = rnd(1280, 768)
g, bta = torch.ones(768, device="cuda"), torch.zeros(768, device="cuda")
def small():
    for _ in range(200):
        ops.gemm(a_s, b_s)
        ops.layernorm_fwd(x, g, bta, 1e-12)
def large():
    for _ in range(60):
        ops.gemm(a_l, b_l)
for name, fn in (("idle", lambda: time.sleep(0.5)), ("small-kernel chain", small), ("large GEMMs", large)):
    stop, out = threading.Event(), []
    th = threading.Thread(target=poll, args=(stop, out)); th.start()
    t0 = time.time()
    while time.time() - t0 < 3.0:
        fn()
        torch.cuda.synchronize()
    stop.set(); th.join()
    print(name, out[len(out) // 2:][:3])
# and the time of the same small kernel right after idle vs right after heavy work
def t_small():
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): ops.gemm(a_s, b_s)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 50 * 1e3
time.sleep(1.0); print("small GEMM after idle: %.1f us" % t_small())
large(); print("small GEMM right after large GEMMs: %.1f us" % t_small())
