import numpy as np
M = np.uint64(0xFFFFFFFF)
def u(x): return np.asarray(x, dtype=np.uint64) & M
def mul32(a, b): return (u(a) * u(b)) & M
def mul24(a, b): return ((u(a) & np.uint64(0xFFFFFF)) * (u(b) & np.uint64(0xFFFFFF))) & M
def mix32(x):
    x = u(x); x ^= x >> np.uint64(16); x = mul32(x, 0x7feb352d); x ^= x >> np.uint64(15); x = mul32(x, 0x846ca68b); x ^= x >> np.uint64(16); return x
def cur(row, cg, off_lo, off_hi, k0, k1):
    base = mix32(mul32(row, 0x9E3779B1) + (u(off_lo) ^ u(k0))) ^ ((mul32(cg, 0x85EBCA77) + mul32(off_hi, 0xC2B2AE3D) + u(k1)) & M)
    return [mix32(base), mix32(base + np.uint64(0x68E31DA4)), mix32(base + np.uint64(0xB5297A4D)), mix32(base + np.uint64(0x1B56C4E9))]
def fin24(y, c1=0x6B43A9, c2=0x52A6B5, s1=15, s2=13):
    y = u(y); y ^= y >> np.uint64(s1); y = mul24(y, c1); y ^= y >> np.uint64(s2); y = mul24(y, c2); y ^= y >> np.uint64(16); return y
def new(row, cg, off_lo, off_hi, k0, k1, fin=fin24):
    rk = mix32(mul32(row, 0x9E3779B1) + (u(off_lo) ^ u(k0))) ^ ((mul32(off_hi, 0xC2B2AE3D) + u(k1)) & M)
    x = (rk + mul32(cg, 0x85EBCA77)) & M
    return [fin(x), fin(x + np.uint64(0x68E31DA4)), fin(x + np.uint64(0xB5297A4D)), fin(x + np.uint64(0x1B56C4E9))]
def fields(words):   # -> [..., 8] 16-bit fields, e order: word j low, word j high
    f = []
    for w in words:
        f.append(w & np.uint64(0xFFFF)); f.append(w >> np.uint64(16))
    return np.stack(f, -1)
def keep_tensor(fn, B, H, N, seed, offset, thr=6554):
    # emulate the attention layout: row = bh*N+q, cg = w*4+g, fields e -> key 32w+16(e>>2)+4g+(e&3)
    NW = 6
    bh = np.arange(B * H)[:, None, None, None]; q = np.arange(N)[None, :, None, None]
    w = np.arange(NW)[None, None, :, None]; g = np.arange(4)[None, None, None, :]
    row = bh * N + q + 0 * w + 0 * g; cg = w * 4 + g + 0 * row
    F = fields(fn(row, cg, offset & 0xFFFFFFFF, offset >> 32, seed & 0xFFFFFFFF, seed >> 32))     # [BH,N,NW,4,8]
    kept = F >= thr
    out = np.zeros((B * H, N, NW * 32), bool)
    for e in range(8):
        for gg in range(4):
            out[:, :, (np.arange(NW) * 32 + 16 * (e >> 2) + 4 * gg + (e & 3))] = kept[:, :, :, gg, e]
    return out.reshape(B, H, N, NW * 32)[..., :N]
def agree(a, b): return (a == b).mean()
def report(name, fn):
    B, H, N = 4, 12, 182
    p = 0.1; indep = p * p + (1 - p) ** 2
    k0 = keep_tensor(fn, B, H, N, 11, 5)
    res = {"mean": k0.mean() - 0.9}
    res["row+1"] = agree(k0[:, :, :-1], k0[:, :, 1:]) - indep
    res["key+1"] = agree(k0[..., :-1], k0[..., 1:]) - indep
    for lag in (2, 4, 16, 32, 64): res["key+%d" % lag] = agree(k0[..., :-lag], k0[..., lag:]) - indep
    res["head+1"] = agree(k0[:, :-1], k0[:, 1:]) - indep
    res["sample+1"] = agree(k0[:-1], k0[1:]) - indep
    for nm, (s, o) in {"off+1": (11, 6), "off+2^32": (11, 5 + (1 << 32)), "seed+1": (12, 5), "seed+2^32": (11 + (1 << 32), 5), "off+2": (11, 7), "off+256": (11, 5 + 256)}.items():
        res[nm] = agree(k0, keep_tensor(fn, B, H, N, s, o)) - indep
    res["colrate"] = np.abs(k0.mean((0, 1, 2)) - 0.9).max()
    # avalanche on raw words: flip cg / row lowest bits
    rng = np.random.RandomState(0)
    row = rng.randint(0, 1 << 20, 200000); cg = rng.randint(0, 48, 200000)
    a = np.stack(fn(row, cg, 5, 0, 11, 0), -1)
    worst = 0
    for dr, dc in ((1, 0), (0, 1), (2, 0), (0, 2), (16, 0), (0, 4)):
        b = np.stack(fn(row ^ dr, cg ^ dc, 5, 0, 11, 0), -1)
        x = a ^ b
        bits = ((x[..., None] >> np.arange(32, dtype=np.uint64)) & np.uint64(1)).mean(0)   # [4,32]
        worst = max(worst, np.abs(bits - 0.5).max())
    res["avalanche_worst_bias"] = worst
    # pairwise field correlation within one 128-bit draw
    F = fields(fn(row, cg, 5, 0, 11, 0)) >= 6554
    cmax = 0
    for i in range(8):
        for j in range(i + 1, 8):
            cmax = max(cmax, abs(agree(F[:, i], F[:, j]) - indep))
    res["field_pair_max"] = cmax
    print(name, " ".join("%s=%+.4f" % kv for kv in res.items()))
report("current", cur)
report("fin24", new)
def fin24b(y): return fin24(y, 0xD35A2D, 0xA6B52B, 16, 12)
report("fin24b", lambda *a: new(*a, fin=fin24b))
def fin1(y):   # one multiply only
    y = u(y); y ^= y >> np.uint64(15); y = mul24(y, 0x6B43A9); y ^= y >> np.uint64(13); y = (y + (y << np.uint64(7))) & M; y ^= y >> np.uint64(16); return y
report("fin1mul", lambda *a: new(*a, fin=fin1))
def new_s(row, cg, off_lo, off_hi, k0, k1):
    rk = mix32(mul32(row, 0x9E3779B1) + (u(off_lo) ^ u(k0))) ^ ((mul32(off_hi, 0xC2B2AE3D) + u(k1)) & M)
    x = (rk + mul32(cg, 0x85EBCA77)) & M
    t = x ^ (x >> np.uint64(15))
    out = []
    for c1 in (0x6B43A9, 0xD35A2D, 0x9E3B71, 0xB5297B):
        y = mul24(t, c1); y ^= y >> np.uint64(13); y = mul24(y, 0x52A6B5); y ^= y >> np.uint64(16); out.append(y)
    return out
report("fin24s", new_s)
def new_t(row, cg, off_lo, off_hi, k0, k1):
    rk = mix32(mul32(row, 0x9E3779B1) + (u(off_lo) ^ u(k0))) ^ ((mul32(off_hi, 0xC2B2AE3D) + u(k1)) & M)
    x = (rk + mul32(cg, 0x85EBCA77)) & M
    t = x ^ (x >> np.uint64(15)); tb = t >> np.uint64(11)
    out = []
    for c1, d1 in ((0x6B43A9, 0x3C6EF3), (0xD35A2D, 0x7F4A7D), (0x9E3B71, 0x2545F5), (0xB5297B, 0x5851F5)):
        y = (mul24(t, c1) + mul24(tb, d1)) & M; y ^= y >> np.uint64(13); y = mul24(y, 0x52A6B5); y ^= y >> np.uint64(16); out.append(y)
    return out
report("fin24t", new_t)
# collision census over one launch's (row, cg) domain
B, H, N = 64, 12, 182
row = np.arange(B * H * N, dtype=np.uint64)[:, None]; cg = np.arange(24, dtype=np.uint64)[None, :]
for nm, fn in (("fin24s", new_s), ("fin24t", new_t), ("current", cur)):
    w = fn(row, cg, 5, 0, 11, 0)
    key = (w[0] << np.uint64(32)) | w[1]
    n = key.size; nu = np.unique(key).size
    print(nm, "draws", n, "distinct 64-bit prefixes", nu, "duplicates", n - nu)


def hidden_report(name, fn, M_=4096, D=768, thr=6554):
    """the hidden-state layout: one draw per (row, 8-column group), fields e = column within the group (tests/test_attention_gpu.py hidden-state checks)"""
    p = 0.1; indep = p * p + (1 - p) ** 2
    def mask(seed, offset):
        row = np.arange(M_)[:, None]; c8 = np.arange(D // 8)[None, :]
        F = fields(fn(row + 0 * c8, c8 + 0 * row, offset & 0xFFFFFFFF, offset >> 32, (seed & 0xFFFFFFFF) ^ 0x5bd1e995, (seed >> 32) ^ 0x1b873593))   # [M, D/8, 8]
        return (F >= thr).reshape(M_, D)
    h0 = mask(3, 9)
    res = {"mean": h0.mean() - 0.9, "row+1": agree(h0[:-1], h0[1:]) - indep, "col+1": agree(h0[:, :-1], h0[:, 1:]) - indep, "col+8": agree(h0[:, :-8], h0[:, 8:]) - indep,
           "row+256": agree(h0[:-256], h0[256:]) - indep, "off+1": agree(h0, mask(3, 10)) - indep, "seed+1": agree(h0, mask(4, 9)) - indep,
           "off+2^32": agree(h0, mask(3, 9 + (1 << 32))) - indep, "colrate": np.abs(h0.mean(0) - 0.9).max(), "rowrate": np.abs(h0.mean(1) - 0.9).max()}
    # against the attention stream under the same (seed, offset): keep_tensor flips nothing, hidden flips the key -> independent
    print("hidden", name, " ".join("%s=%+.4f" % kv for kv in res.items()))


hidden_report("current", cur)
hidden_report("fin24t", new_t)
