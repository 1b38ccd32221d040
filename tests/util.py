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
    got = got.detach().float().cpu().double()
    ref = ref.detach().float().cpu().double()
    assert got.shape == ref.shape, (name, got.shape, ref.shape)
    assert torch.isfinite(got).all(), "%s: non-finite values" % name
    bound = frac * ref.abs().max() + ulps * (2.0 ** -8) * ref.abs()
    err = (got - ref).abs()
    worst = (err - bound).max().item()
    assert worst <= 0, "%s: max err %.4g (scale %.4g) exceeds bound by %.4g" % (name, err.max().item(), ref.abs().max().item(), worst)
    return err.max().item() / max(ref.abs().max().item(), 1e-30)
