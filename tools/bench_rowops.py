import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sam_textvqa_amd import ops
M, D = 11648, 768
def t(fn, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
x = torch.randn(M, D, device="cuda").to(torch.bfloat16); dy = torch.randn(M, D, device="cuda").to(torch.bfloat16)
g, b = torch.ones(D, device="cuda"), torch.zeros(D, device="cuda")
y, mean, rstd = ops.layernorm_fwd(x, g, b, 1e-12)
dg, db, dbias = torch.zeros(D, device="cuda"), torch.zeros(D, device="cuda"), torch.zeros(D, device="cuda")
u = t(lambda: ops.layernorm_fwd(x, g, b, 1e-12)); print("ln_fwd  %.1f us  %.0f GB/s" % (u, 2 * M * D * 2 / u / 1e3))
u = t(lambda: ops.layernorm_bwd(dy, x, mean, rstd, g, dg, db, dbias, want_dropped=True, p_drop=0.1, seed=1, offset=1)); print("ln_bwd(+drop) %.1f us  %.0f GB/s" % (u, 4 * M * D * 2 / u / 1e3))
u = t(lambda: ops.layernorm_bwd(dy, x, mean, rstd, g, dg, db, dbias, want_dropped=True, p_drop=0.0)); print("ln_bwd(no drop) %.1f us  %.0f GB/s" % (u, 3 * M * D * 2 / u / 1e3))
