"""Thin torch-facing wrappers over the C-ABI (raw ops; autograd lives in autograd.py).

Every function takes/returns CUDA(ROCm) tensors, allocates outputs with torch, passes raw device
pointers + the current HIP stream to libsam_hip.so and never synchronises."""
import torch

from . import _capi as capi

BF16 = torch.bfloat16


def _chk(t, dtype, name):
    if not t.is_cuda:
        raise capi.SamHipError("%s must live on the GPU (no CPU path in this package)" % name)
    if t.dtype != dtype:
        raise capi.SamHipError("%s: expected dtype %s, got %s" % (name, dtype, t.dtype))
    if not t.is_contiguous():
        raise capi.SamHipError("%s must be contiguous" % name)
    return t


def words_per_row(n):
    nw = capi.call("sam_attn_words_per_row", int(n))
    if nw <= 0:
        raise capi.SamHipError("sequence length %d exceeds the fused-attention limit (384 keys)" % n)
    return nw


# ----------------------------------------------------------------------------- masks
def mask_bits_prefix_lm(key_valid, n_dec):
    """key_valid: uint8 [B, n_enc] -> uint32 [B, 1, N, NW] (MMT prefix-LM/causal mask, sa_m4c.py:805-844)."""
    _chk(key_valid, torch.uint8, "key_valid")
    b, n_enc = key_valid.shape
    n = n_enc + n_dec
    nw = words_per_row(n)
    out = torch.empty((b, 1, n, nw), dtype=torch.int32, device=key_valid.device)
    capi.call("sam_mask_bits_prefix_lm", capi.ptr(key_valid), b, n_enc, n_dec, nw, capi.ptr(out), capi.stream_handle())
    return out


def mask_bits_from_additive(mask):
    """mask: float32 [B,1,N,N] additive (0 / -10000) -> uint32 [B,1,N,NW]."""
    _chk(mask, torch.float32, "attention_mask")
    b, one, n, n2 = mask.shape
    if one != 1 or n != n2:
        raise capi.SamHipError("attention_mask must be [B,1,N,N], got %s" % (tuple(mask.shape),))
    nw = words_per_row(n)
    out = torch.empty((b, 1, n, nw), dtype=torch.int32, device=mask.device)
    capi.call("sam_mask_bits_from_additive", capi.ptr(mask), b, n, nw, capi.ptr(out), capi.stream_handle())
    return out


def mask_bits_spatial(base_bits, adj, n_txt, n_heads, quadrants):
    """base_bits [B,1,N,NW] & relation tensor int8 [B,Noo,Noo,R] -> uint32 [B,H,N,NW] (sa_m4c.py:470-552,568)."""
    _chk(base_bits, torch.int32, "base_bits")
    _chk(adj, torch.int8, "spatial_adj_matrix")
    b, _, n, nw = base_bits.shape
    n_oo, r = adj.shape[1], adj.shape[3]
    qbits = 0
    for quad in quadrants:
        if quad not in (1, 2, 4, 7, 8, 9):
            raise ValueError("illegal attention_mask_quadrants entry %r" % (quad,))  # sa_m4c.py:548-549
        qbits |= 1 << quad
    out = torch.empty((b, n_heads, n, nw), dtype=torch.int32, device=adj.device)
    capi.call("sam_mask_bits_spatial", capi.ptr(base_bits), capi.ptr(adj), b, n, nw, n_txt, n_oo, r, n_heads, qbits,
              capi.ptr(out), capi.stream_handle())
    return out


# ----------------------------------------------------------------------------- attention
def attn_fwd(qkv, allow, batch, n_heads, scale, p_drop=0.0, seed=0, offset=0):
    """qkv bf16 [B*N, 3*H*64]; allow uint32 [B, H or 1, N, NW] -> (out bf16 [B*N, H*64], lse2 f32 [B,H,N], keep or None)."""
    _chk(qkv, BF16, "qkv"); _chk(allow, torch.int32, "allow")
    rows, three_d = qkv.shape
    n = rows // batch
    d_model = three_d // 3
    out = torch.empty((rows, d_model), dtype=BF16, device=qkv.device)
    lse2 = torch.empty((batch, n_heads, n), dtype=torch.float32, device=qkv.device)
    keep = torch.empty((batch, n_heads, n, allow.shape[-1]), dtype=torch.int32, device=qkv.device) if p_drop > 0 else None
    sh = 0 if allow.shape[1] == 1 else allow.stride(1)
    capi.call("sam_attn_fwd", capi.ptr(qkv), capi.ptr(allow), allow.stride(0), sh, batch, n, n_heads, d_model // n_heads,
              float(scale), float(p_drop), int(seed), int(offset), capi.ptr(out), capi.ptr(lse2), capi.ptr(keep), capi.stream_handle())
    return out, lse2, keep


def attn_bwd(dout, qkv, lse2, allow, keep, batch, n_heads, scale, p_drop=0.0):
    """-> dqkv bf16 [B*N, 3*H*64]."""
    _chk(dout, BF16, "dout"); _chk(qkv, BF16, "qkv")
    rows, three_d = qkv.shape
    n = rows // batch
    d_model = three_d // 3
    dqkv = torch.empty_like(qkv)
    delta = torch.empty((batch, n_heads, n), dtype=torch.float32, device=qkv.device)
    sh = 0 if allow.shape[1] == 1 else allow.stride(1)
    capi.call("sam_attn_bwd", capi.ptr(dout), capi.ptr(qkv), capi.ptr(lse2), capi.ptr(allow), allow.stride(0), sh,
              capi.ptr(keep), batch, n, n_heads, d_model // n_heads, float(scale), float(p_drop), capi.ptr(dqkv), capi.ptr(delta),
              capi.stream_handle())
    return dqkv


# ----------------------------------------------------------------------------- GEMM
def _dp(t):
    return None if t is None else t.data_ptr()


def gemm(a, b, *, a_kcontig=True, b_kcontig=True, m=None, n=None, k=None, out=None, out_dtype=BF16, epilogue=capi.EPI_NONE,
         bias=None, residual=None, aux_out=None, aux_in=None, accumulate=False, p_drop=0.0, seed=0, offset=0):
    """C[M,N] = epilogue(sum_k A(m,k) B(k,n)); see include/sam_hip.h `sam_gemm_bf16` for layouts and epilogues.
    a, b: 2-D bf16 tensors whose LAST dim is contiguous (row stride = leading dimension)."""
    for t, nm in ((a, "A"), (b, "B")):
        if not t.is_cuda or t.dtype != BF16 or t.dim() != 2 or t.stride(1) != 1:
            raise capi.SamHipError("gemm %s: need a 2-D bf16 GPU tensor with contiguous last dim" % nm)
    M = m if m is not None else (a.shape[0] if a_kcontig else a.shape[1])
    K = k if k is not None else (a.shape[1] if a_kcontig else a.shape[0])
    N = n if n is not None else (b.shape[0] if b_kcontig else b.shape[1])
    if out is None:
        out = torch.empty((M, N), dtype=out_dtype, device=a.device)
    d = capi.GemmDesc()
    d.M, d.N, d.K = M, N, K
    d.a_kcontig, d.b_kcontig = int(a_kcontig), int(b_kcontig)
    d.c_is_f32, d.accumulate, d.epilogue = int(out.dtype == torch.float32), int(accumulate), int(epilogue)
    d.A, d.lda, d.B, d.ldb, d.C, d.ldc = a.data_ptr(), a.stride(0), b.data_ptr(), b.stride(0), out.data_ptr(), out.stride(0)
    d.bias = _dp(bias)
    d.residual, d.ldr = _dp(residual), (residual.stride(0) if residual is not None else 0)
    d.aux_out, d.aux_in = _dp(aux_out), _dp(aux_in)
    aux = aux_out if aux_out is not None else aux_in
    d.ld_aux = aux.stride(0) if aux is not None else 0
    d.p_drop, d.seed, d.offset = float(p_drop), int(seed), int(offset)
    capi.call("sam_gemm_bf16", d, capi.stream_handle())
    return out
