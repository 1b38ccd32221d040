import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sam_textvqa_amd import ops
R = 11648
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
rnd = lambda *s: torch.randn(*s, device="cuda").to(torch.bfloat16)
for (N, K) in [(3072, 768), (768, 3072), (3072, 3072)]:
    x, w = rnd(R, K), rnd(N, K)
    print("N=%d K=%d: full %.1f us | fills only %.1f us | mfma only %.1f us   (flops %.1f G)" % (N, K, t(lambda: ops.gemm(x, w, force_tile=128)),
          t(lambda: ops.gemm(x, w, force_tile=1128)), t(lambda: ops.gemm(x, w, force_tile=2128)), 2e-9 * R * N * K))
