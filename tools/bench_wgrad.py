"""Grouped weight-gradient GEMM (one encoder layer's dWqkv, dWo, dW1, dW2 at the bench batch): 4-wave 128x128 kernel vs the 8-wave 256x256 kernel
with in-launch pair exchange.  Prints us per launch and TFLOP/s.  Run on the GPU box: python tools/bench_wgrad.py [R]"""
import os
import sys
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sam_textvqa_amd as pkg  # noqa: E402
from sam_textvqa_amd import ops  # noqa: E402


def main():
    R = int(sys.argv[1]) if len(sys.argv) > 1 else 64 * 182
    shapes = [(768, 3072), (3072, 768), (768, 768), (2304, 768)]
    g = torch.Generator(device="cuda").manual_seed(0)
    jobs = []
    for m, n in shapes:
        dy = (torch.randn(R, m, device="cuda", generator=g) * 0.5).bfloat16()
        x = (torch.randn(R, n, device="cuda", generator=g) * 0.5).bfloat16()
        jobs.append((dy, x, torch.zeros(m, n, device="cuda"), torch.zeros(m, device="cuda")))
    flops = sum(2.0 * R * m * n for m, n in shapes)
    for name, ft in [("4-wave 128x128", 128), ("8-wave 256x256 pair", 1256), ("default", 0)]:
        try:
            for _ in range(5):
                ops.wgrad_grouped(jobs, force_tile=ft)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            n = 50
            e0.record()
            for _ in range(n):
                ops.wgrad_grouped(jobs, force_tile=ft)
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / n
            print("%-24s R=%d  %8.1f us  %7.1f TFLOP/s" % (name, R, us, flops / us * 1e-6), flush=True)
        except Exception as e:  # noqa: BLE001
            print(name, "failed:", e, flush=True)


if __name__ == "__main__":
    main()
