import sys, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sam_textvqa_amd import ops
kind = sys.argv[1] if len(sys.argv) > 1 else "fwd"
R, N, K = 11648, int(sys.argv[2]) if len(sys.argv) > 2 else 3072, int(sys.argv[3]) if len(sys.argv) > 3 else 768
rnd = lambda *s: torch.randn(*s, device="cuda").to(torch.bfloat16)
x, w, dy = rnd(R, K), rnd(N, K), rnd(R, N)
out = torch.zeros(N, K, device="cuda")
for _ in range(5):
    if kind == "fwd": ops.gemm(x, w)
    elif kind == "dgrad": ops.gemm(dy, w, b_kcontig=False)
    else: ops.gemm(dy, x, a_kcontig=False, b_kcontig=False, out=out, accumulate=True, split_k=-1)
torch.cuda.synchronize()
