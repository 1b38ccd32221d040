"""Tolerances for the bf16 HIP path (BASELINE.json north_star: "within 1e-3 bf16 tolerance").

A kernel is compared with the fp32 oracle evaluated on the SAME bf16-rounded inputs.  The bound is
    |got - ref| <= 1e-3 * max|ref|  +  ulps * 2^-8 * |ref|
i.e. the north-star's 1e-3 (relative to the tensor's scale) for the kernel's own arithmetic
(accumulation order, P rounded to bf16 ahead of the PV MFMA, exp2/erf approximations) plus the
quantum of storing the result itself in bf16 (one ulp = 2^-8 relative, element-wise)."""
import numpy as np
import torch


def unpack_bits(words, n):
    """uint32 words [..., NW] -> bool [..., n]"""
    w = words.detach().cpu().numpy().astype(np.uint32)
    bits = ((w[..., :, None] >> np.arange(32, dtype=np.uint32)) & 1).astype(bool)
    return torch.from_numpy(bits.reshape(w.shape[:-1] + (-1,))[..., :n])


def assert_close_bf16(got, ref, frac=1e-3, ulps=1, name=""):
    # the comparison itself runs where `got` lives (fp64 on the device for device tensors: the element-wise passes over 10 M-element outputs on the host were
    # most of the GPU suite's wall time); the REFERENCE values are whatever the caller computed, on the host
    dev = got.device
    got = got.detach().to(torch.float64)
    ref = ref.detach().to(device=dev, dtype=torch.float64)
    assert got.shape == ref.shape, (name, got.shape, ref.shape)
    assert torch.isfinite(got).all(), "%s: non-finite values" % name
    bound = frac * ref.abs().max() + ulps * (2.0 ** -8) * ref.abs()
    err = (got - ref).abs()
    worst = (err - bound).max().item()
    assert worst <= 0, "%s: max err %.4g (scale %.4g) exceeds bound by %.4g" % (name, err.max().item(), ref.abs().max().item(), worst)
    return err.max().item() / max(ref.abs().max().item(), 1e-30)


def run_child(cmd, env, marker, name, timeout=600, cwd=None):
    """run a child process (a data-parallel rank, bench.py under torch.distributed.run, ...), keep its FULL stdout / stderr on disk and fail with the path
    and a long tail: pytest's assertion repr cuts captured text to a few hundred characters, which in round 4 hid the one line that named the cause
    (`Process group watchdog thread terminated with exception: ...`).  Files: <repo>/gpurun_out/test_children/<name>.{stdout,stderr} (gpurun merges that
    directory back), returns the CompletedProcess."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out_dir = os.path.join(root, "gpurun_out", "test_children")
    os.makedirs(out_dir, exist_ok=True)
    try:
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout, cwd=cwd or root)
        rc, so, se = r.returncode, r.stdout, r.stderr
    except subprocess.TimeoutExpired as e:
        dec = lambda b: b.decode("utf-8", "replace") if isinstance(b, bytes) else (b or "")
        r, rc, so, se = None, "timeout after %ds" % timeout, dec(e.stdout), dec(e.stderr)
    paths = []
    for ext, text in (("stdout", so), ("stderr", se)):
        path = os.path.join(out_dir, "%s.%s" % (name, ext))
        with open(path, "w") as f:
            f.write(text)
        paths.append(path)
    if rc != 0 or (marker and marker not in so):
        import sys
        print("child %s: rc %s; full output in %s" % (name, rc, paths), file=sys.stderr)
        raise AssertionError("child %s ended with rc %s%s; full output: %s\n---- stdout tail ----\n%s\n---- stderr tail ----\n%s"
                             % (name, rc, "" if not marker or marker in so else " without printing %r" % marker, paths, so[-3000:], se[-8000:]))
    return r
