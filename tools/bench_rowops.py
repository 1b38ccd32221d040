"""LayerNorm fwd/bwd micro-benchmark at the encoder shape (B=64 -> 11648 x 768), HBM-cold: rotates over enough buffer sets to exceed the 256 MB MALL"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sam_textvqa_amd import ops
M, D, SETS = 11648, 768, 8
def t(fn, n=32):
    for i in range(SETS): fn(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n): fn(i % SETS)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
xs = [torch.randn(M, D, device="cuda").to(torch.bfloat16) for _ in range(SETS)]
dys = [torch.randn(M, D, device="cuda").to(torch.bfloat16) for _ in range(SETS)]
junk = [torch.empty(M, D * 2, device="cuda", dtype=torch.bfloat16) for _ in range(SETS)]   # keeps the outputs of a call from staying cache-resident
g, b = torch.ones(D, device="cuda"), torch.zeros(D, device="cuda")
y, mean, rstd = ops.layernorm_fwd(xs[0], g, b, 1e-12)
dg, db, dbias = torch.zeros(D, device="cuda"), torch.zeros(D, device="cuda"), torch.zeros(D, device="cuda")
u = t(lambda i: ops.layernorm_fwd(xs[i], g, b, 1e-12)); print("ln_fwd          %.1f us  %.0f GB/s" % (u, 2 * M * D * 2 / u / 1e3))
u = t(lambda i: ops.layernorm_bwd(dys[i], xs[i], mean, rstd, g, dg, db, dbias, want_dropped=True, p_drop=0.1, seed=1, offset=1)); print("ln_bwd(+drop)   %.1f us  %.0f GB/s" % (u, 4 * M * D * 2 / u / 1e3))
u = t(lambda i: ops.layernorm_bwd(dys[i], xs[i], mean, rstd, g, dg, db, dbias, want_dropped=True, p_drop=0.0)); print("ln_bwd(no drop) %.1f us  %.0f GB/s" % (u, 3 * M * D * 2 / u / 1e3))
# the same traffic as a plain strided copy (sam_copy_blocks: 2 x 18 MB in, 2 x 18 MB out): what a kernel that only moves the bytes gets at this size
outs = [torch.empty(M, D * 2, device="cuda", dtype=torch.bfloat16) for _ in range(SETS)]
u = t(lambda i: ops.copy_blocks([(xs[i].view(1, M, D), outs[i][:, :D].unsqueeze(0)), (dys[i].view(1, M, D), outs[i][:, D:].unsqueeze(0))])); print("copy 2x18 MB    %.1f us  %.0f GB/s" % (u, 4 * M * D * 2 / u / 1e3))
