"""Adam + sumsq + gradient clear at the model size (96.6 M parameters)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sam_textvqa_amd import ops
n = 96633224
p, g, m, v = (torch.randn(n, device="cuda") * 0.01 for _ in range(4))
v.abs_()
pb = torch.empty(n, dtype=torch.bfloat16, device="cuda")
gn = torch.zeros(1, device="cuda")
def t(fn, k=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(k): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / k * 1e3
print("sumsq %.1f us   adam %.1f us (%.2f TB/s)   zero %.1f us" % (t(lambda: ops.sumsq(g, gn)), (a := t(lambda: ops.adam_step(p, g, m, v, pb, [n], [1e-4], 3, gnorm_sq=gn, max_norm=0.25))), 30.0 * n / a / 1e6, t(lambda: g.zero_())))
