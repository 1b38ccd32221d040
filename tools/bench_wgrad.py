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


def mixed():
    """an MMT layer pair (R rows) and TextBert's three layers (1280 rows): two launches against one launch of 20 problems"""
    R = 64 * 182
    shapes = [(768, 3072), (3072, 768), (768, 768), (2304, 768)]
    g = torch.Generator(device="cuda").manual_seed(0)
    pad = int(os.environ.get("PAD", "0"))          # experiment: leading dimensions padded by PAD elements (L2 channel conflicts of power-of-two-ish row strides?)
    def mk(rows, n):
        out = []
        for _ in range(n):
            for m, k in shapes:
                dy = (torch.randn(rows, m + pad, device="cuda", generator=g) * 0.5).bfloat16()[:, :m]
                x = (torch.randn(rows, k + pad, device="cuda", generator=g) * 0.5).bfloat16()[:, :k]
                out.append((dy, x, torch.zeros(m, k, device="cuda"), torch.zeros(m, device="cuda")))
        return out
    pair, tb = mk(R, 2), mk(1280, 3)
    fl = sum(2.0 * j[0].shape[0] * j[0].shape[1] * j[1].shape[1] for j in pair + tb)
    def t(fn, n=30):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e3 / n
    a = t(lambda: ops.wgrad_grouped(pair, accumulate=False))
    b = t(lambda: ops.wgrad_grouped(tb, accumulate=False))
    c = t(lambda: ops.wgrad_grouped(pair + tb, accumulate=False))
    print("pair alone %.1f us, TextBert alone %.1f us (sum %.1f us, %.1f TFLOP/s); one launch of 20 problems %.1f us (%.1f TFLOP/s)" % (a, b, a + b, fl / (a + b) * 1e-6, c, fl / c * 1e-6))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "mixed":
        mixed()
        sys.exit(0)
    main()
