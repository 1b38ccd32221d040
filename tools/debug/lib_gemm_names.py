"""prints nothing itself: run under rocprofv3 --kernel-trace --stats to see which library (hipBLASLt/Tensile) kernels and
macro-tiles torch.matmul picks for the shapes of one SA-M4C step (calibration only, not part of the product path)"""
import torch
R = 11648
def rnd(*s): return torch.randn(*s, device="cuda").to(torch.bfloat16)
for (N, K) in [(768, 768), (2304, 768), (3072, 768), (768, 3072)]:
    x, w, dy = rnd(R, K), rnd(N, K), rnd(R, N)
    for _ in range(5):
        torch.matmul(x, w.t()); torch.matmul(dy, w); torch.matmul(dy.t(), x)
torch.cuda.synchronize()
