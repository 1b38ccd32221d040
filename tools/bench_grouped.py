import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sam_textvqa_amd import ops
R = 11648
rnd = lambda *s: torch.randn(*s, device="cuda").to(torch.bfloat16)
jobs = []
for (m, n) in [(768, 3072), (3072, 768), (768, 768), (2304, 768)]:
    jobs.append((rnd(R, m), rnd(R, n), torch.zeros(m, n, device="cuda"), torch.zeros(m, device="cuda")))
for _ in range(3): ops.wgrad_grouped(jobs)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): ops.wgrad_grouped(jobs)
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) / 10 * 1e3
fl = sum(2.0 * R * j[0].shape[1] * j[1].shape[1] for j in jobs)
print("grouped wgrad (MMT layer): %.1f us  %.1f TFLOP/s" % (us, fl / us / 1e6))
