import torch, math
torch.manual_seed(0)
N, d = 182, 64
def bf(x): return x.to(torch.bfloat16).to(torch.float64)
def f16(x): return x.to(torch.float32).to(torch.float16).to(torch.float64)
def run(seed, p_drop=0.1, qscale=1.5, mask_density=0.4):
    g = torch.Generator().manual_seed(seed)
    q, k, v = (bf(torch.randn(N, d, generator=g) * qscale) for _ in range(3))
    do = bf(torch.randn(N, d, generator=g))
    allow = torch.rand(N, N, generator=g) < mask_density
    allow[:, 0] = True
    keep = (torch.rand(N, N, generator=g) >= p_drop).double() if p_drop else torch.ones(N, N, dtype=torch.float64)
    inv_keep = 1.0 / (1.0 - p_drop)
    scale = 1 / 8.0
    # fp64 reference
    qr, kr, vr = q.clone().requires_grad_(), k.clone().requires_grad_(), v.clone().requires_grad_()
    s = (qr @ kr.T) * scale
    s = s.masked_fill(~allow, float("-inf"))
    P = torch.softmax(s, -1)
    O = (P * keep * inv_keep) @ vr
    (O * do).sum().backward()
    # kernel emulation (fp32 accumulations emulated in fp64; only the operand roundings modelled)
    S = (q @ k.T) * scale
    S = S.masked_fill(~allow, float("-inf"))
    M = S.max(-1, keepdim=True).values
    pe = torch.exp(S - M)
    ssum = pe.sum(-1, keepdim=True)
    p16 = f16(pe * keep)                       # fwd: unnormalised P~ (without inv_keep) in fp16
    ev = 14 - math.floor(math.log2(v.abs().max()))
    v16 = f16(v * 2.0 ** ev)
    Ok = (p16 @ v16) * 2.0 ** -ev * inv_keep / ssum
    O_hi = bf(Ok); O_lo = bf(Ok - O_hi)
    res = {}
    res["out"] = (O_hi, O.detach())
    lse = M + torch.log(ssum)
    # backward
    edo = 14 - math.floor(math.log2((do.abs().max() * inv_keep)))
    do16 = f16(do * inv_keep * 2.0 ** edo)      # scaled
    dop = do16 * 2.0 ** -edo                     # value seen
    for variant in ("hi_lo", "hi_only"):
        Ouse = O_hi + O_lo if variant == "hi_lo" else O_hi
        delta = (dop * Ouse).sum(-1, keepdim=True) / inv_keep
        Pn = torch.exp(S - lse)
        dPk = (dop @ v16.T) * 2.0 ** -ev * keep
        dS = Pn * (dPk - delta) * scale
        # fp16 dS with a block scale from the bound
        bound = 2 * 64 * dop.abs().max() * v.abs().max() * scale
        es = 14 - math.ceil(math.log2(bound))
        dS16 = f16(dS * 2.0 ** es)
        eq = 14 - math.floor(math.log2(q.abs().max())); ek = 14 - math.floor(math.log2(k.abs().max()))
        q16 = f16(q * 2.0 ** eq); k16 = f16(k * 2.0 ** ek)
        dQ = (dS16 @ k16) * 2.0 ** (-es - ek)
        dK = (dS16.T @ q16) * 2.0 ** (-es - eq)
        Pt16 = f16(Pn * keep)                    # P~ without inv_keep (folded into dO')
        dV = (Pt16.T @ do16) * 2.0 ** -edo
        for nm, got, ref in (("dq", dQ, qr.grad), ("dk", dK, kr.grad), ("dv", dV, vr.grad)):
            res[variant + "." + nm] = (bf(got), ref)
        # bf16 single-rounded dS for comparison
        if variant == "hi_lo":
            dSb = bf(dS); 
            res["bf16dS.dq"] = (bf(dSb @ k), qr.grad)
    out = {}
    for nm, (got, ref) in res.items():
        err = (got - ref).abs()
        bound = 1e-3 * ref.abs().max() + 2.0 ** -8 * ref.abs()
        out[nm] = ((err / ref.abs().max()).max().item(), ((err - bound).max() / ref.abs().max()).item())
    return out
import collections
agg = collections.defaultdict(list)
for seed in range(6):
    for dens in (0.4, 1.0):
        r = run(seed, mask_density=dens)
        for k_, v_ in r.items(): agg[k_].append(v_)
for k_, v_ in agg.items():
    print("%-14s max err/max %.2e   worst (err-bound)/max %+.2e" % (k_, max(a for a, b in v_), max(b for a, b in v_)))
