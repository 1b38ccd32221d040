"""Thin torch-facing wrappers over the C-ABI (raw ops; autograd lives in autograd.py).

Every function takes/returns CUDA(ROCm) tensors, allocates outputs with torch, passes raw device
pointers + the current HIP stream to libsam_hip.so and never synchronises."""
import os

import torch

from . import _capi as capi
from . import torchops

BF16 = torch.bfloat16


class _TorchOpsProxy:
    """torch.ops.sam_hip with the package's error type: TORCH_CHECK failures (plain RuntimeError) surface as SamHipError, like the ctypes route's"""
    _cache = {}

    def __getattr__(self, name):
        fn = self._cache.get(name)
        if fn is None:
            op = getattr(torchops.ns(), name)

            def fn(*a, _op=op, **kw):
                try:
                    return _op(*a, **kw)
                except capi.SamHipError:
                    raise
                except RuntimeError as e:
                    raise capi.SamHipError(str(e).split("\n")[0]) from None
            self._cache[name] = fn
        return fn


_PROXY = _TorchOpsProxy()


def _tops():
    """torch.ops.sam_hip when the custom-op route is on (default) and bench.py's per-call event profiler is not recording (it brackets ctypes calls)"""
    return _PROXY if (torchops.enabled() and capi.profiler is None) else None


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
def pack_masks(question_mask, obj_mask, ocr_mask):
    """the batch's three padding masks -> (key_valid uint8 [B, T+No+Nc], question uint8 [B,T], ocr uint8 [B,Nc]) in one launch"""
    ms = []
    for m in (question_mask, obj_mask, ocr_mask):
        if not m.is_cuda:
            raise capi.SamHipError("pack_masks: masks must live on the GPU")
        ms.append(m.contiguous() if m.dtype == torch.int64 else m.ne(0).to(torch.int64).contiguous())
    q, o, c = ms
    t_ = _tops()
    if t_ is not None:
        return t_.pack_masks(q, o, c)
    b, t, no, nc = q.shape[0], q.shape[1], o.shape[1], c.shape[1]
    kv = torch.empty((b, t + no + nc), dtype=torch.uint8, device=q.device)
    q8 = torch.empty((b, t), dtype=torch.uint8, device=q.device)
    c8 = torch.empty((b, nc), dtype=torch.uint8, device=q.device)
    capi.call("sam_pack_masks_u8", capi.ptr(q), t, capi.ptr(o), no, capi.ptr(c), nc, b, capi.ptr(kv), capi.ptr(q8), capi.ptr(c8), capi.stream_handle())
    return kv, q8, c8


def mask_bits_prefix_lm(key_valid, n_dec):
    """key_valid: uint8 [B, n_enc] -> uint32 [B, 1, N, NW] (MMT prefix-LM/causal mask, sa_m4c.py:805-844)."""
    _chk(key_valid, torch.uint8, "key_valid")
    t_ = _tops()
    if t_ is not None:
        return t_.mask_bits_prefix_lm(key_valid, int(n_dec))
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
    t_ = _tops()
    if t_ is not None:
        return t_.mask_bits_from_additive(mask)
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
    t_ = _tops()
    if t_ is not None:
        return t_.pack_relations(base_bits, adj, int(n_txt), int(n_heads), qbits)
    out = torch.empty((b, n_heads, n, nw), dtype=torch.int32, device=adj.device)
    capi.call("sam_mask_bits_spatial", capi.ptr(base_bits), capi.ptr(adj), b, n, nw, n_txt, n_oo, r, n_heads, qbits,
              capi.ptr(out), capi.stream_handle())
    return out


# ----------------------------------------------------------------------------- attention
def attn_fwd(qkv, allow, batch, n_heads, scale, p_drop=0.0, seed=0, offset=0, want_residual=False):
    """qkv bf16 [B*N, 3*H*64]; allow uint32 [B, H or 1, N, NW] -> (out bf16 [B*N, H*64], lse2 f32 [B,H,N], keep or None);
    want_residual=True (training: sam_attn_fwd_train) appends out_lo, the bf16 rounding residual of `out` the one-pass backward takes delta from."""
    _chk(qkv, BF16, "qkv"); _chk(allow, torch.int32, "allow")
    rows, three_d = qkv.shape
    n = rows // batch
    d_model = three_d // 3
    out = torch.empty((rows, d_model), dtype=BF16, device=qkv.device)
    out_lo = torch.empty_like(out) if want_residual else None
    lse2 = torch.empty((batch, n_heads, n), dtype=torch.float32, device=qkv.device)
    keep = torch.empty((batch, n_heads, n, allow.shape[-1]), dtype=torch.int32, device=qkv.device) if p_drop > 0 else None
    sh = 0 if allow.shape[1] == 1 else allow.stride(1)
    nw = allow.shape[-1]
    alg_bytes = batch * ((5 if want_residual else 4) * n * d_model * 2 + n_heads * n * nw * 4 * (2 if p_drop > 0 else 1) + n_heads * n * 4)
    meta = dict(kernel="attn_fwd", bytes=alg_bytes, flops=4.0 * batch * n * n * d_model, shape=(batch, n, n_heads))
    if want_residual:
        capi.call("sam_attn_fwd_train", capi.ptr(qkv), capi.ptr(allow), allow.stride(0), sh, batch, n, n_heads, d_model // n_heads,
                  float(scale), float(p_drop), int(seed), int(offset), capi.ptr(out), capi.ptr(out_lo), capi.ptr(lse2), capi.ptr(keep), capi.stream_handle(), meta=meta)
        return out, lse2, keep, out_lo
    capi.call("sam_attn_fwd", capi.ptr(qkv), capi.ptr(allow), allow.stride(0), sh, batch, n, n_heads, d_model // n_heads,
              float(scale), float(p_drop), int(seed), int(offset), capi.ptr(out), capi.ptr(lse2), capi.ptr(keep), capi.stream_handle(), meta=meta)
    return out, lse2, keep


def attn_probs(qkv, allow, lse2, keep, batch, n_heads, scale, p_drop=0.0, head_scale=None):
    """the attention probabilities the fused forward did not materialise: fp32 [B, H, N, N] from its q | k rows, allow bits, log2-sum-exps and keep bits
    (include/sam_hip.h: sam_attn_probs; sa_m4c.py:600-609 `output_attentions`)"""
    _chk(qkv, BF16, "qkv"); _chk(allow, torch.int32, "allow"); _chk(lse2, torch.float32, "lse2")
    rows, three_d = qkv.shape
    n, d_model = rows // batch, three_d // 3
    sh = 0 if allow.shape[1] == 1 else allow.stride(1)
    if head_scale is not None:
        _chk(head_scale, torch.float32, "head_scale")
    out = torch.empty((batch, n_heads, n, n), dtype=torch.float32, device=qkv.device)
    capi.call("sam_attn_probs", capi.ptr(qkv), capi.ptr(allow), allow.stride(0), sh, capi.ptr(lse2), capi.ptr(keep if p_drop > 0 else None), capi.ptr(head_scale),
              batch, n, n_heads, d_model // n_heads, float(scale), float(p_drop if keep is not None else 0.0), capi.ptr(out), capi.stream_handle(),
              meta=dict(kernel="attn_probs", bytes=4.0 * out.numel()))
    return out


def attn_fwd_rows(qkv, allow, batch, n_heads, scale, q_begin, out, lse2):
    """inference: recompute only query rows >= q_begin of `out` (bf16 [B*N, H*64], updated in place) against all keys of qkv"""
    _chk(qkv, BF16, "qkv"); _chk(allow, torch.int32, "allow"); _chk(out, BF16, "out")
    rows, three_d = qkv.shape
    n = rows // batch
    d_model = three_d // 3
    sh = 0 if allow.shape[1] == 1 else allow.stride(1)
    capi.call("sam_attn_fwd_rows", capi.ptr(qkv), capi.ptr(allow), allow.stride(0), sh, batch, n, n_heads, d_model // n_heads, float(scale), int(q_begin),
              capi.ptr(out), capi.ptr(lse2), capi.stream_handle())
    return out


def attn_fwd_dec(qkv_enc, qkv_dec, allow, batch, n, n_dec, n_heads, scale, out_dec=None, kv_group=1):
    """decoding step: decoder rows' q|k|v in their own compact buffer qkv_dec [B*n_dec, 3*H*64], encoder rows read from the full-pass cache
    qkv_enc [B*N, 3*H*64] -> attention output of the decoder rows [B*n_dec, H*64] (sam_attn_fwd_dec).  kv_group > 1 (beam search): qkv_enc and
    allow hold B / kv_group samples, each shared by kv_group consecutive decoder blocks (sam_attn_fwd_dec_shared)."""
    _chk(qkv_enc, BF16, "qkv_enc"); _chk(qkv_dec, BF16, "qkv_dec"); _chk(allow, torch.int32, "allow")
    d_model = qkv_enc.shape[1] // 3
    if out_dec is None:
        out_dec = torch.empty((batch * n_dec, d_model), dtype=BF16, device=qkv_enc.device)
    sh = 0 if allow.shape[1] == 1 else allow.stride(1)
    if kv_group > 1:
        capi.call("sam_attn_fwd_dec_shared", capi.ptr(qkv_enc), capi.ptr(qkv_dec), capi.ptr(allow), allow.stride(0), sh, batch, int(kv_group), n, n_dec, n_heads,
                  d_model // n_heads, float(scale), capi.ptr(out_dec), capi.stream_handle())
        return out_dec
    capi.call("sam_attn_fwd_dec", capi.ptr(qkv_enc), capi.ptr(qkv_dec), capi.ptr(allow), allow.stride(0), sh, batch, n, n_dec, n_heads, d_model // n_heads,
              float(scale), capi.ptr(out_dec), capi.stream_handle())
    return out_dec


def greedy_pick(fixed, ocr, prev_inds):
    """prev_inds[r, s + 1] = argmax over [fixed | ocr] of row (r, s), s < S - 1, in place (sa_m4c.py:299-302); fixed f32 [R*S, V], ocr f32 [R*S, No]"""
    _chk(prev_inds, torch.int64, "prev_inds")
    r, s = prev_inds.shape
    capi.call("sam_greedy_pick", capi.ptr(fixed), fixed.stride(0), capi.ptr(ocr), ocr.stride(0), r, s, fixed.shape[1], ocr.shape[1], capi.ptr(prev_inds), capi.stream_handle())
    return prev_inds


def attn_dec_row(qkv_enc, qkv_dec, allow, batch, n, n_dec, t, n_heads, scale, kv_group=1):
    """attention output of decoder row t only, bf16 [B, H*64] (sam_attn_dec_row): qkv_enc [B/kv_group * N, 3*H*64] cached by the full pass, qkv_dec
    [B*n_dec, 3*H*64] with rows 0..t of every decoder sample valid"""
    _chk(qkv_enc, BF16, "qkv_enc"); _chk(qkv_dec, BF16, "qkv_dec"); _chk(allow, torch.int32, "allow")
    d_model = qkv_enc.shape[1] // 3
    out = torch.empty((batch, d_model), dtype=BF16, device=qkv_enc.device)
    sh = 0 if allow.shape[1] == 1 else allow.stride(1)
    capi.call("sam_attn_dec_row", capi.ptr(qkv_enc), capi.ptr(qkv_dec), capi.ptr(allow), allow.stride(0), sh, batch, int(kv_group), n, n_dec, int(t), n_heads,
              d_model // n_heads, float(scale), capi.ptr(out), out.stride(0), capi.stream_handle())
    return out


def beam_step(fixed, ocr, n_samples, beam, seqs, cum, done, eos, t=0, ctl=None, prev_pos=None):
    """one BeamSearch.decode step (sam_beam_step), state (seqs int64 [B*K, S], cum f32 [B*K], done u8 [B*K]) updated in place"""
    _chk(seqs, torch.int64, "seqs"); _chk(cum, torch.float32, "cum"); _chk(done, torch.uint8, "done")
    s = seqs.shape[1]
    if beam > 1 and os.environ.get("SAM_BEAM_STEP_SPLIT", "1") != "0":
        # scan over one block per (sample, beam) + merge (sam_beam_step_split): the same result in a quarter of the time at beam 5
        ws = _workspace(int(capi.call("sam_beam_step_ws_bytes", int(n_samples), int(beam))), fixed.device, "beam_step")
        capi.call("sam_beam_step_split", capi.ptr(fixed), fixed.stride(0), capi.ptr(ocr), ocr.stride(0), int(n_samples), int(beam), s, fixed.shape[1], ocr.shape[1], int(eos),
                  int(t), capi.ptr(ctl), capi.ptr(cum), capi.ptr(done), capi.ptr(seqs), capi.ptr(prev_pos), capi.ptr(ws), capi.stream_handle())
        return
    capi.call("sam_beam_step", capi.ptr(fixed), fixed.stride(0), capi.ptr(ocr), ocr.stride(0), int(n_samples), int(beam), s, fixed.shape[1], ocr.shape[1], int(eos), int(t),
              capi.ptr(ctl), capi.ptr(cum), capi.ptr(done), capi.ptr(seqs), capi.ptr(prev_pos), capi.stream_handle())


_FUSED_MAX_N = None


def attn_bwd_fused_max_n():
    global _FUSED_MAX_N
    if _FUSED_MAX_N is None:
        _FUSED_MAX_N = int(capi.call("sam_attn_bwd_fused_max_n"))
    return _FUSED_MAX_N


def attn_bwd(dout, qkv, lse2, allow, keep, batch, n_heads, scale, p_drop=0.0, out=None, out_lo=None):
    """-> dqkv bf16 [B*N, 3*H*64].  With the forward's output and residual (attn_fwd(..., want_residual=True)) and N <= attn_bwd_fused_max_n() (384: 193..384 keys run
    as 2 x 2 sub-problems of 192) this is the one-pass kernel (sam_attn_bwd_fused); otherwise the two-kernel form (sam_attn_bwd: a dQ pass that also produces delta, then dK / dV)."""
    _chk(dout, BF16, "dout"); _chk(qkv, BF16, "qkv")
    rows, three_d = qkv.shape
    n = rows // batch
    d_model = three_d // 3
    dqkv = torch.empty_like(qkv)
    sh = 0 if allow.shape[1] == 1 else allow.stride(1)
    bits_bytes = n_heads * n * (allow.shape[-1] * 4 * (2 if keep is not None else 1) + 8)
    if out is not None and out_lo is not None and n <= attn_bwd_fused_max_n() and os.environ.get("SAM_ATTN_BWD_FUSED", "1") != "0":
        _chk(out, BF16, "out"); _chk(out_lo, BF16, "out_lo")
        capi.call("sam_attn_bwd_fused", capi.ptr(dout), capi.ptr(qkv), capi.ptr(out), capi.ptr(out_lo), capi.ptr(lse2), capi.ptr(allow), allow.stride(0), sh,
                  capi.ptr(keep), batch, n, n_heads, d_model // n_heads, float(scale), float(p_drop), capi.ptr(dqkv), capi.stream_handle(),
                  meta=dict(kernel="attn_bwd(fused)", flops=10.0 * batch * n * n * d_model, shape=(batch, n, n_heads),
                            bytes=batch * (10 * n * d_model * 2 + bits_bytes)))        # reads q,k,v,dO,O,O_lo; writes dq,dk,dv
        return dqkv
    delta = torch.empty((batch, n_heads, n), dtype=torch.float32, device=qkv.device)
    capi.call("sam_attn_bwd", capi.ptr(dout), capi.ptr(qkv), capi.ptr(lse2), capi.ptr(allow), allow.stride(0), sh,
              capi.ptr(keep), batch, n, n_heads, d_model // n_heads, float(scale), float(p_drop), capi.ptr(dqkv), capi.ptr(delta),
              capi.stream_handle(),
              meta=dict(kernel="attn_bwd(dq+dkdv)", flops=10.0 * batch * n * n * d_model, shape=(batch, n, n_heads),
                        bytes=batch * (8 * n * d_model * 2 + bits_bytes)))
    return dqkv


# ----------------------------------------------------------------------------- GEMM
def _dp(t):
    return None if t is None else t.data_ptr()


def gemm(a, b, *, a_kcontig=True, b_kcontig=True, m=None, n=None, k=None, out=None, out_dtype=BF16, epilogue=capi.EPI_NONE,
         bias=None, residual=None, aux_out=None, aux_in=None, accumulate=False, p_drop=0.0, seed=0, offset=0, split_k=0, bias_grad=None, force_tile=0, ln=None):
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
    wgrad_split = bool(accumulate) and out.dtype == torch.float32 and epilogue == capi.EPI_NONE      # partials reduced INTO C by sam_gemm_splitk_reduce
    if split_k == 0 and not accumulate and bias_grad is None and force_tile == 0 and M <= 4096 and K >= 1536 and M * N <= (4 << 20):
        split_k = -1       # skinny problem with a long K (TextBert's 20 tokens/sample, classifier dgrad): let the library split K and fold
                           # the epilogue into the partial-sum reduction (it declines when the grid already fills the chip)
    d.split_k, d.bias_grad, d.force_tile = int(split_k), _dp(bias_grad), int(force_tile)
    if ln is not None:          # capi.LnFuse, filled by gemm_ln: the library normalises the rows itself when it runs this GEMM split over K
        import ctypes as C
        d.ln = C.pointer(ln)
    if split_k not in (0, 1):
        want = split_k if split_k > 0 else (32 if wgrad_split else 8)
        ws = _workspace(min(want * (M * N + M) * 4, 96 << 20), a.device, "splitk")
        d.ws, d.ws_bytes, d.defer_reduce = ws.data_ptr(), ws.numel() * 4, 1
    capi.call("sam_gemm_bf16", d, capi.stream_handle(),
              meta=dict(kernel="gemm<a_kc=%d,b_kc=%d,epi=%d,f32=%d>" % (d.a_kcontig, d.b_kcontig, d.epilogue, d.c_is_f32), flops=2.0 * M * N * K, shape=(M, N, K)))
    if d.split_k_used > 1 and wgrad_split:   # the fixed-order reduction of the split-K partials is its own launch (and its own profile row)
        capi.call("sam_gemm_splitk_reduce", d.ws, d.split_k_used, M, N, out.data_ptr(), out.stride(0), d.bias_grad, capi.stream_handle(),
                  meta=dict(kernel="splitk_reduce", bytes=4.0 * M * N * (d.split_k_used + 2)))
    return out


def gemm_ln(a, b, gamma, beta, eps, **kw):
    """LayerNorm(gemm(a, b, epilogue=EPI_BIAS_DROPOUT_RES, ...)) -> (z, y, mean, rstd): z = the pre-LayerNorm sums (bf16, what the backward reads), y / mean /
    rstd as layernorm_fwd(z).  One launch less when the GEMM runs split over K (sam_ln_fuse: the reduction pass normalises the rows it owns); the
    separate sam_layernorm_fwd otherwise -- the same bits either way."""
    m = a.shape[0] if kw.get("a_kcontig", True) else a.shape[1]
    n = b.shape[0] if kw.get("b_kcontig", True) else b.shape[1]
    if capi.profiler is not None or n % 4 or n > 2048 or gamma.dtype != torch.float32 or beta.dtype != torch.float32:
        z = gemm(a, b, **kw)
        return (z,) + tuple(layernorm_fwd(z, gamma, beta, eps))
    y = torch.empty((m, n), dtype=BF16, device=a.device)
    mean = torch.empty((m,), dtype=torch.float32, device=a.device)
    rstd = torch.empty((m,), dtype=torch.float32, device=a.device)
    ln = capi.LnFuse()
    ln.gamma, ln.beta, ln.eps, ln.y, ln.ldy, ln.mean, ln.rstd, ln.done = gamma.data_ptr(), beta.data_ptr(), float(eps), y.data_ptr(), y.stride(0), mean.data_ptr(), rstd.data_ptr(), 0
    if m >= 2048:          # MMT-size products: the exchange workspace of the LayerNorm inside the launch (zero-filled once per device and stream; sam_ln_fuse.xws)
        xws = _ln_xws(a.device, int(capi.call("sam_gemm_ln_ws_bytes", m, n)))
        ln.xws, ln.xws_bytes = xws.data_ptr(), xws.numel() * 4
    z = gemm(a, b, ln=ln, **kw)
    if ln.done:
        return z, y, mean, rstd
    return (z,) + tuple(layernorm_fwd(z, gamma, beta, eps))


# ----------------------------------------------------------------------------- scratch
_WS = {}
_LN_XWS = {}


def _ln_xws(device, nbytes):
    """per-(device, stream) exchange workspace of the in-launch LayerNorm: zero-filled when (re)allocated, left with zero counters by every launch"""
    key = (device, capi.stream_handle().value)
    buf = _LN_XWS.get(key)
    if buf is None or buf.numel() * 4 < nbytes:
        buf = torch.zeros((nbytes + 3) // 4, dtype=torch.float32, device=device)
        _LN_XWS[key] = buf
    return buf


def capi_ln_words():
    """words in front of the (mean, M2) pairs of an in-launch LayerNorm workspace: error word + counters (csrc/gemm_common.h: LN_WS_PART0)"""
    return 64 + 4 * 1024


def ln_xws_check():
    """raise if a launch's bounded wait for a row's statistics ran out (word 0 of a workspace); synchronises"""
    for buf in _LN_XWS.values():
        if int(buf[:1].view(torch.int32).item()) != 0:
            buf[:1].zero_()
            raise capi.SamHipError("LayerNorm inside the GEMM launch: a block waited in vain for the other parts of its rows (another kernel held the CUs its partners needed)")


def _workspace(nbytes, device, tag):
    """per-(device, stream, tag) scratch that only grows; the kernels that use one buffer are ordered by its stream (TextBert runs on a
    side stream next to the object / OCR encoders: each stream gets its own scratch)"""
    key = (device, capi.stream_handle().value, tag)
    buf = _WS.get(key)
    if buf is None or buf.numel() * 4 < nbytes:
        buf = torch.empty((nbytes + 3) // 4, dtype=torch.float32, device=device)
        _WS[key] = buf
    return buf


# ----------------------------------------------------------------------------- layernorm / reductions
def layernorm_fwd(x, gamma, beta, eps):
    """x [M,D] bf16 or fp32 (last dim contiguous) -> (y bf16 [M,D], mean f32 [M], rstd f32 [M])"""
    if x.dim() != 2 or x.stride(1) != 1 or x.dtype not in (BF16, torch.float32) or not x.is_cuda:
        raise capi.SamHipError("layernorm_fwd: need a 2-D bf16/fp32 GPU tensor with contiguous rows")
    m, d = x.shape
    y = torch.empty((m, d), dtype=BF16, device=x.device)
    mean = torch.empty(m, dtype=torch.float32, device=x.device)
    rstd = torch.empty(m, dtype=torch.float32, device=x.device)
    capi.call("sam_layernorm_fwd", capi.ptr(x), int(x.dtype == torch.float32), x.stride(0), capi.ptr(gamma), capi.ptr(beta), float(eps), m, d,
              capi.ptr(y), y.stride(0), capi.ptr(mean), capi.ptr(rstd), capi.stream_handle())
    return y, mean, rstd


class LnFinalizeQueue:
    """deferred LayerNorm-backward finalizes of the per-kernel (ctypes) route: see sam_layernorm_bwd_finalize_batch in include/sam_hip.h.  The Trainer
    switches deferral on for the span of a backward pass and flushes before the gradient norm; the C++ custom ops keep their own queue
    (torch.ops.sam_hip.set_ln_defer / ln_finalize_flush)."""
    defer = False
    items = []      # (ws tensor, rows, accumulate, dgamma, dbeta, dbias, D)

    @classmethod
    def flush(cls):
        if not cls.items:
            return
        by_d = {}
        for it in cls.items:
            by_d.setdefault(it[6], []).append(it)
        for d, its in by_d.items():
            arr = (capi.LnFinalizeItem * len(its))()
            for a, (ws, rows, acc, dg, db, dbias, _) in zip(arr, its):
                a.ws, a.rows, a.accumulate, a.dgamma, a.dbeta, a.dbias = ws.data_ptr(), rows, acc, dg.data_ptr(), db.data_ptr(), _dp(dbias)
            capi.call("sam_layernorm_bwd_finalize_batch", arr, len(its), d, capi.stream_handle())
        cls.items = []

    @classmethod
    def clear(cls):
        """drop queued finalizes without running them (a backward pass that raised: their workspaces may be gone)"""
        cls.items = []
        cls.defer = False


def layernorm_bwd(dy, x, mean, rstd, gamma, dgamma, dbeta, dbias=None, want_dropped=False, p_drop=0.0, seed=0, offset=0, accumulate=True, may_defer=False):
    """-> (dx bf16, dx_dropped bf16 or None); dgamma/dbeta/dbias (fp32 [D]) are accumulated in place (overwritten with accumulate=False).
    may_defer: the parameter-gradient reduction may be left to LnFinalizeQueue.flush() when deferral is on"""
    _chk(dy, BF16, "dy")
    m, d = x.shape
    dx = torch.empty((m, d), dtype=BF16, device=x.device)
    dxd = torch.empty((m, d), dtype=BF16, device=x.device) if (want_dropped and p_drop > 0) else None
    defer = may_defer and LnFinalizeQueue.defer
    nbytes = capi.call("sam_layernorm_bwd_ws_bytes", d)
    ws = torch.empty(((nbytes + 3) // 4,), dtype=torch.float32, device=x.device) if defer else _workspace(nbytes, x.device, "ln")
    capi.call("sam_layernorm_bwd", capi.ptr(dy), dy.stride(0), capi.ptr(x), int(x.dtype == torch.float32), x.stride(0), capi.ptr(mean), capi.ptr(rstd),
              capi.ptr(gamma), m, d, capi.ptr(dx), capi.ptr(dxd), dx.stride(0), float(p_drop), int(seed), int(offset), capi.ptr(dgamma), capi.ptr(dbeta),
              capi.ptr(dbias), int(bool(accumulate)) | (4 if defer else 0), capi.ptr(ws), capi.stream_handle())
    if defer:
        LnFinalizeQueue.items.append((ws, int(capi.call("sam_layernorm_bwd_partial_rows", m)), int(bool(accumulate)), dgamma, dbeta, dbias, d))
    return dx, (dxd if dxd is not None else (dx if want_dropped else None))


def add_dropout(a, b, p_drop, seed=0, offset=0):
    """dropout(a + b) -> bf16 [M, D] (b may be None: plain dropout, which is also its own backward on dy); hidden-state dropout stream"""
    for name, t in (("a", a), ("b", b)):
        if t is not None and (not t.is_cuda or t.dtype != BF16 or t.dim() != 2 or t.stride(1) != 1):
            raise capi.SamHipError("add_dropout: %s must be a bf16 [M, D] GPU tensor with unit column stride" % name)
    m, d = a.shape
    out = torch.empty((m, d), dtype=BF16, device=a.device)
    capi.call("sam_add_dropout_bf16", capi.ptr(a), a.stride(0), capi.ptr(b), 0 if b is None else b.stride(0), capi.ptr(out), out.stride(0), m, d,
              float(p_drop), int(seed), int(offset), capi.stream_handle())
    return out


def input_encoder_fwd(za, bbox, wb, bias_b, ln_a, ln_b, p_drop=0.0, seed=0, offset=0):
    """dropout(LN_a(za) + LN_b(bbox W_b^T + b_b)) -> (out bf16 [R, D], stats f32 [R, 4]); za bf16 [R, D], bbox fp32 [R, >= 4] (row stride free),
    wb bf16 [D, ldw] (sam_input_encoder_fwd)"""
    _chk(za, BF16, "za")
    if not bbox.is_cuda or bbox.dtype != torch.float32 or bbox.dim() != 2 or bbox.stride(1) != 1 or bbox.shape[1] < 4:
        raise capi.SamHipError("input_encoder_fwd: bbox must be a 2-D fp32 GPU tensor with >= 4 contiguous columns")
    r, d = za.shape
    out = torch.empty((r, d), dtype=BF16, device=za.device)
    stats = torch.empty((r, 4), dtype=torch.float32, device=za.device)
    capi.call("sam_input_encoder_fwd", capi.ptr(za), za.stride(0), capi.ptr(bbox), bbox.stride(0), capi.ptr(wb), wb.stride(0), capi.ptr(bias_b), capi.ptr(ln_a.weight),
              capi.ptr(ln_a.bias), capi.ptr(ln_b.weight), capi.ptr(ln_b.bias), float(ln_a.variance_epsilon), r, d, float(p_drop), int(seed), int(offset), capi.ptr(out),
              out.stride(0), capi.ptr(stats), capi.stream_handle())
    return out, stats


def input_encoder_bwd(dy, za, bbox, wb, bias_b, ln_a, ln_b, stats, dwb, dbias_b, p_drop=0.0, seed=0, offset=0, accumulate=True):
    """-> d za bf16 [R, D]; d gamma / d beta of both LayerNorms, d bias_b and d wb (fp32 [D, ldgw] view, columns 0..3) are accumulated in place"""
    _chk(dy, BF16, "dy")
    r, d = za.shape
    dza = torch.empty((r, d), dtype=BF16, device=za.device)
    ws = _workspace(capi.call("sam_input_encoder_bwd_ws_bytes", r, d), za.device, "enc_in")
    capi.call("sam_input_encoder_bwd", capi.ptr(dy), dy.stride(0), capi.ptr(za), za.stride(0), capi.ptr(bbox), bbox.stride(0), capi.ptr(wb), wb.stride(0), capi.ptr(bias_b),
              capi.ptr(ln_a.weight), capi.ptr(ln_b.weight), capi.ptr(stats), r, d, float(p_drop), int(seed), int(offset), capi.ptr(dza), dza.stride(0),
              capi.ptr(ln_a.weight.grad), capi.ptr(ln_a.bias.grad), capi.ptr(ln_b.weight.grad), capi.ptr(ln_b.bias.grad), capi.ptr(dbias_b), capi.ptr(dwb), dwb.stride(0),
              int(bool(accumulate)), capi.ptr(ws), capi.stream_handle())
    return dza


def colsum(x, out, accumulate=True):
    """out[n] (+)= sum_m x[m,n]  (x bf16 [M,N], out fp32 [N])"""
    m, n = x.shape
    ws = _workspace(capi.call("sam_colsum_ws_bytes", n), x.device, "colsum")
    capi.call("sam_colsum_bf16", capi.ptr(x), x.stride(0), m, n, capi.ptr(out), int(accumulate), capi.ptr(ws), capi.stream_handle())
    return out


# ----------------------------------------------------------------------------- front end (csrc/embed.hip)
def l2norm_pack(x, out, col0=0, normalize=True, zero_upto=0, eps=1e-12):
    """out[:, col0:col0+D] = bf16(F.normalize(x, dim=-1)) (normalize=False: plain cast); x fp32 [M, D]; out bf16 [M, ldo];
    columns [col0+D, zero_upto) of out are zeroed"""
    _chk(out, BF16, "out")
    if not x.is_cuda or x.dtype != torch.float32 or x.dim() != 2:
        raise capi.SamHipError("l2norm_pack: x must be a 2-D fp32 GPU tensor")
    m, d = x.shape
    if x.stride(1) != 1 or out.stride(1) != 1 or out.shape[0] != m:
        raise capi.SamHipError("l2norm_pack: x / out must be row-major with the same number of rows")
    capi.call("sam_l2norm_pack_bf16", capi.ptr(x), x.stride(0), m, d, int(bool(normalize)), float(eps), capi.ptr(out), out.stride(0), int(col0), int(zero_upto),
              capi.stream_handle())
    return out


def embed_sum_fwd(pos, tt, rows, seq, table=None, ids=None, type_ids=None):
    """fp32 [rows, D] = table[ids] (bf16, optional) + pos[r % seq] + tt[type_ids[r]] (type 0 when type_ids is None)"""
    d = pos.shape[1]
    out = torch.empty((rows, d), dtype=torch.float32, device=pos.device)
    capi.call("sam_embed_sum_fwd", capi.ptr(table), table.stride(0) if table is not None else 0, capi.ptr(ids), table.shape[0] if table is not None else 0,
              capi.ptr(pos), pos.stride(0), int(seq), capi.ptr(tt), tt.stride(0), capi.ptr(type_ids), tt.shape[0], rows, d, capi.ptr(out), out.stride(0),
              capi.stream_handle())
    return out


def embed_sum_bwd(d_e, seq, d_pos, d_tt, type_ids=None, n_types=1):
    """d_pos[s] += sum_b d_e[b*seq+s];  d_tt[t] += sum_{type==t} d_e  (d_e bf16 [R, D]; d_pos / d_tt fp32 gradient rows, accumulated)"""
    _chk(d_e, BF16, "d_e")
    r, d = d_e.shape
    ws = _workspace(capi.call("sam_embed_sum_bwd_ws_bytes", int(seq), int(n_types), d), d_e.device, "embed")
    capi.call("sam_embed_sum_bwd", capi.ptr(d_e), d_e.stride(0), r, d, int(seq), capi.ptr(type_ids), int(n_types), capi.ptr(d_pos), d_pos.stride(0),
              capi.ptr(d_tt), d_tt.stride(0), capi.ptr(ws), capi.stream_handle())


def gather2_add_fwd(ans, ocr, inds, n_ocr, emb=None, p_drop=0.0, seed=0, offset=0):
    """bf16 [B*S, D] = (ind < V ? ans[ind] : ocr[b*n_ocr + ind - V]) + dropout(emb);  ans bf16 [V,D], ocr bf16 [B*n_ocr,D], inds int64 [B,S]"""
    _chk(ans, BF16, "ans"); _chk(ocr, BF16, "ocr"); _chk(inds, torch.int64, "inds")
    b, s = inds.shape
    v, d = ans.shape
    out = torch.empty((b * s, d), dtype=BF16, device=ans.device)
    capi.call("sam_gather2_add_fwd", capi.ptr(ans), ans.stride(0), v, capi.ptr(ocr), ocr.stride(0), int(n_ocr), capi.ptr(inds), b, s, d, capi.ptr(emb),
              emb.stride(0) if emb is not None else 0, float(p_drop), int(seed), int(offset), capi.ptr(out), out.stride(0), capi.stream_handle())
    return out


def gather2_add_bwd(dy, inds, v, n_ocr, want_emb=True, p_drop=0.0, seed=0, offset=0):
    """-> (d_ans fp32 [V,D], d_ocr fp32 [B*n_ocr,D], d_emb bf16 [B*S,D] | None)"""
    _chk(dy, BF16, "dy")
    b, s = inds.shape
    d = dy.shape[1]
    d_ans = torch.empty((v, d), dtype=torch.float32, device=dy.device)
    d_ocr = torch.empty((b * n_ocr, d), dtype=torch.float32, device=dy.device)
    copy_blocks([(None, d_ans.view(1, 1, -1)), (None, d_ocr.view(1, 1, -1))])       # both atomics buffers cleared by one launch
    d_emb = torch.empty((b * s, d), dtype=BF16, device=dy.device) if want_emb else None
    capi.call("sam_gather2_add_bwd", capi.ptr(dy), dy.stride(0), int(v), int(n_ocr), capi.ptr(inds), b, s, d, capi.ptr(d_ans), d_ans.stride(0), capi.ptr(d_ocr),
              d_ocr.stride(0), float(p_drop), int(seed), int(offset), capi.ptr(d_emb), d_emb.stride(0) if want_emb else 0, capi.stream_handle())
    return d_ans, d_ocr, d_emb


# ----------------------------------------------------------------------------- loss / pointer net / optimizer
def bce_loss(fixed, ocr, targets, loss_mask, grad_scale=1.0, global_count=None):
    """fixed f32 [R,V], ocr f32 [R,No], targets f32 [R,V+No], loss_mask f32 [R] -> (loss f32 [1], d_fixed bf16 [R,V], d_ocr f32 [R,No]);
    global_count (device f32 scalar): the all-reduced number of unmasked steps of the global batch, used as the normaliser instead of this rank's"""
    r, v = fixed.shape
    no = ocr.shape[1]
    t_ = _tops()
    if t_ is not None and loss_mask.is_contiguous():
        return t_.bce_loss(fixed, ocr, targets, loss_mask, float(grad_scale), global_count)
    loss = torch.empty(1, dtype=torch.float32, device=fixed.device)
    d_fixed = torch.empty((r, v), dtype=BF16, device=fixed.device)
    d_ocr = torch.empty((r, no), dtype=torch.float32, device=fixed.device)
    capi.call("sam_bce_loss", capi.ptr(fixed), fixed.stride(0), capi.ptr(ocr), ocr.stride(0), capi.ptr(targets), targets.stride(0), capi.ptr(loss_mask),
              r, v, no, float(grad_scale), capi.ptr(global_count), capi.ptr(loss), capi.ptr(d_fixed), d_fixed.stride(0), capi.ptr(d_ocr), d_ocr.stride(0), capi.stream_handle())
    return loss, d_fixed, d_ocr


def ptr_scores_fwd(q, k, ocr_mask_u8, scale):
    """q bf16 [B,S,D], k bf16 [B,No,D], mask u8 [B,No] -> f32 [B,S,No]"""
    b, s, d = q.shape
    no = k.shape[1]
    t_ = _tops()
    if t_ is not None:
        return t_.ptr_scores(q, k, ocr_mask_u8, float(scale))
    out = torch.empty((b, s, no), dtype=torch.float32, device=q.device)
    capi.call("sam_ptr_scores_fwd", capi.ptr(q), capi.ptr(k), capi.ptr(ocr_mask_u8), b, s, no, d, float(scale), capi.ptr(out), out.stride(0), out.stride(1),
              capi.stream_handle())
    return out


def ptr_scores_bwd(dscores, q, k, scale):
    b, s, d = q.shape
    no = k.shape[1]
    t_ = _tops()
    if t_ is not None:
        return t_.ptr_scores_bwd(dscores, q, k, float(scale))
    dq, dk = torch.empty_like(q), torch.empty_like(k)
    capi.call("sam_ptr_scores_bwd", capi.ptr(dscores), dscores.stride(0), dscores.stride(1), capi.ptr(q), capi.ptr(k), b, s, no, d, float(scale),
              capi.ptr(dq), capi.ptr(dk), capi.stream_handle())
    return dq, dk


def _sparse_struct(sparse):
    """sparse = (lo, hi, row_len, touched uint8 tensor) or None -> (ctypes pointer or None, keep-alive)"""
    if sparse is None:
        return None, None
    import ctypes as C
    s = capi.SparseRows()
    s.lo, s.hi, s.row_len, s.touched = int(sparse[0]), int(sparse[1]), int(sparse[2]), sparse[3].data_ptr()
    return C.cast(C.pointer(s), C.c_void_p), s


def sumsq(g, out, sparse=None):
    """out[0] = sum g^2; sparse = (lo, hi, row_len, touched): rows of that region whose flag is 0 are skipped (sam_sparse_rows)"""
    t_ = _tops()
    if t_ is not None:
        t_.sumsq(g, out, list(sparse[:3]) if sparse else [], sparse[3] if sparse else None)
        return out
    ws = _workspace(capi.call("sam_sumsq_ws_bytes"), g.device, "sumsq")
    sp, keep = _sparse_struct(sparse)
    capi.call("sam_sumsq_f32", capi.ptr(g), g.numel(), sp, capi.ptr(out), capi.ptr(ws), capi.stream_handle())
    return out


def adam_step(p, g, m, v, p_bf16, seg_end, seg_lr, step, gnorm_sq=None, max_norm=0.0, betas=(0.9, 0.999), eps=1e-8, sparse=None):
    t_ = _tops()
    if t_ is not None:
        t_.adam_step(p, g, m, v, p_bf16, [int(e) for e in seg_end], [float(l) for l in seg_lr], int(step), float(betas[0]), float(betas[1]), float(eps), gnorm_sq,
                     float(max_norm), None, list(sparse[:3]) if sparse else [], sparse[3] if sparse else None)
        return
    import ctypes as C
    n = len(seg_end)
    ends = (C.c_int64 * n)(*[int(e) for e in seg_end])
    lrs = (C.c_float * n)(*[float(l) for l in seg_lr])
    sp, keep = _sparse_struct(sparse)
    capi.call("sam_adam_step", capi.ptr(p), capi.ptr(g), capi.ptr(m), capi.ptr(v), capi.ptr(p_bf16), p.numel(), ends, lrs, n, float(betas[0]), float(betas[1]),
              float(eps), int(step), capi.ptr(gnorm_sq), float(max_norm), sp, capi.stream_handle())


def adam_step_dev(p, g, m, v, p_bf16, seg_end, dev_sched, gnorm_sq=None, max_norm=0.0, betas=(0.9, 0.999), eps=1e-8, sparse=None):
    """adam_step with the schedule [lr per segment, 1 - beta1^t, 1 - beta2^t] read from the device tensor `dev_sched` (graph-captured steps)"""
    t_ = _tops()
    if t_ is not None:
        t_.adam_step(p, g, m, v, p_bf16, [int(e) for e in seg_end], [], 0, float(betas[0]), float(betas[1]), float(eps), gnorm_sq, float(max_norm), dev_sched,
                     list(sparse[:3]) if sparse else [], sparse[3] if sparse else None)
        return
    import ctypes as C
    n = len(seg_end)
    ends = (C.c_int64 * n)(*[int(e) for e in seg_end])
    sp, keep = _sparse_struct(sparse)
    capi.call("sam_adam_step_dev", capi.ptr(p), capi.ptr(g), capi.ptr(m), capi.ptr(v), capi.ptr(p_bf16), p.numel(), ends, n, float(betas[0]), float(betas[1]),
              float(eps), capi.ptr(dev_sched), capi.ptr(gnorm_sq), float(max_norm), sp, capi.stream_handle())


def adam_step_range(p, g, m, v, p_bf16, seg_end, dev_sched, lo, hi, gnorm_sq=None, max_norm=0.0, betas=(0.9, 0.999), eps=1e-8, sparse=None, zero_grad=True, gate=None,
                    max_blocks=0):
    """adam_step_dev over the elements [lo, hi) only (sam_adam_step_range): the update in pieces, each on the stream that needs it; `zero_grad`: the piece's
    gradient is cleared after use; `gate`: int32 device scalar, 0 = the launch does nothing"""
    import ctypes as C
    n = len(seg_end)
    ends = (C.c_int64 * n)(*[int(e) for e in seg_end])
    sp, keep = _sparse_struct(sparse)          # (the row-sparse region is walked by the piece that contains it)
    capi.call("sam_adam_step_range", capi.ptr(p), capi.ptr(g), capi.ptr(m), capi.ptr(v), capi.ptr(p_bf16), p.numel(), ends, n, float(betas[0]), float(betas[1]),
              float(eps), capi.ptr(dev_sched), capi.ptr(gnorm_sq), float(max_norm), sp, int(lo), int(hi), int(bool(zero_grad)), capi.ptr(gate), int(max_blocks), capi.stream_handle())


def copy_blocks(blocks):
    """up to 8 strided copies / casts / accumulations / zero-fills in one launch (sam_copy_blocks).  Each block: (src | None, dst, accumulate=False) with
    src / dst 3-D views [batches, rows, cols] (any batch / row stride, unit column stride, bf16 or fp32); src None fills dst with zeros."""
    import ctypes as C
    for i in range(0, len(blocks), 8):
        part = blocks[i: i + 8]
        arr = (capi.CopyDesc * len(part))()
        for e, blk in zip(arr, part):
            src, dst = blk[0], blk[1]
            acc = bool(blk[2]) if len(blk) > 2 else False
            if dst.dim() != 3 or dst.stride(2) != 1 or dst.dtype not in (BF16, torch.float32) or not dst.is_cuda:
                raise capi.SamHipError("copy_blocks: dst must be a 3-D bf16 / fp32 GPU view with unit column stride")
            if src is not None and (src.shape != dst.shape or src.stride(2) != 1 or src.dtype not in (BF16, torch.float32) or not src.is_cuda):
                raise capi.SamHipError("copy_blocks: src must match dst's shape, bf16 / fp32, unit column stride")
            e.src, e.dst = (src.data_ptr() if src is not None else None), dst.data_ptr()
            e.batches, e.rows, e.cols = dst.shape
            e.src_batch_stride, e.src_row_stride = (src.stride(0), src.stride(1)) if src is not None else (0, 0)
            e.dst_batch_stride, e.dst_row_stride = dst.stride(0), dst.stride(1)
            e.src_f32, e.dst_f32, e.accumulate = int(src is not None and src.dtype == torch.float32), int(dst.dtype == torch.float32), int(acc)
        capi.call("sam_copy_blocks", C.cast(arr, C.c_void_p), len(part), capi.stream_handle())


def rowvec(mode, a, b=None, vec=None):
    """bf16 rows: "mul" a * b | "add_vec" a + vec[cols] | "mul_vec" a * vec[cols] (vec fp32) -> bf16 [rows, cols] (include/sam_hip.h: sam_rowvec_bf16)"""
    m = {"mul": 0, "add_vec": 1, "mul_vec": 2}[mode]
    _chk(a, BF16, "a")
    if a.dim() != 2 or a.stride(1) != 1:
        raise capi.SamHipError("rowvec: a must be 2-D with contiguous rows")
    if m == 0:
        if b is None:
            raise capi.SamHipError("rowvec mul: the second operand is missing")
        _chk(b, BF16, "b")
        if b.shape != a.shape or b.stride(1) != 1:
            raise capi.SamHipError("rowvec mul: b must have a's shape and contiguous rows")
    else:
        if vec is None:
            raise capi.SamHipError("rowvec %s: the fp32 vector is missing" % mode)
        _chk(vec, torch.float32, "vec")
        if vec.numel() != a.shape[1] or not vec.is_contiguous():
            raise capi.SamHipError("rowvec: vec must be a contiguous fp32 vector of a's width")
    out = torch.empty(a.shape, dtype=BF16, device=a.device)
    capi.call("sam_rowvec_bf16", m, capi.ptr(a), a.stride(0), capi.ptr(b), b.stride(0) if b is not None else 0, capi.ptr(vec), capi.ptr(out), out.stride(0),
              a.shape[0], a.shape[1], capi.stream_handle(), meta=dict(kernel="rowvec", bytes=2.0 * a.numel() * (3 if m == 0 else 2)))
    return out


def ge_u8(x, threshold):
    """uint8 [n] = x >= threshold for an int64 tensor (token types of the previous predictions, sa_m4c.py:936)"""
    _chk(x, torch.int64, "x")
    out = torch.empty(x.numel(), dtype=torch.uint8, device=x.device)
    capi.call("sam_ge_u8", capi.ptr(x), x.numel(), int(threshold), capi.ptr(out), capi.stream_handle())
    return out


def greedy_decode_ws(batch, steps, n_layers, device):
    """zero-filled workspace of sam_greedy_decode_steps for `batch` rows (int32 word 256 = its sticky error flag)"""
    nbytes = capi.call("sam_greedy_decode_ws_bytes", int(batch), int(steps), int(n_layers))
    return torch.zeros((nbytes + 3) // 4, dtype=torch.int32, device=device)


def tile_weight(w):
    """bf16 [N, K] (any row stride) -> the fragment-tiled copy [ceil(N / 16), K / 8, 16, 8] sam_greedy_decode_steps reads (rows zero-padded to 16)"""
    n, k = w.shape
    if k % 8:
        raise capi.SamHipError("tile_weight: K must be a multiple of 8")
    npad = (n + 15) // 16 * 16
    if npad != n:
        w = torch.cat([w, w.new_zeros((npad - n, k))], dim=0)
    return w.reshape(npad // 16, 16, k // 8, 8).permute(0, 2, 1, 3).contiguous()


def greedy_decode_steps(layers, d, ws, t_begin, t_end):
    """greedy decoding steps t_begin .. t_end-1 in one persistent launch (sam_greedy_decode_steps, include/sam_hip.h).
    layers: per encoder layer a dict of tensors {wqkv, wo, w1, w2 (bf16, fragment-tiled: tile_weight), bqkv, bo, b1, b2, ln1_g, ln1_b, ln2_g, ln2_b (fp32), qkv (bf16 [B*N, 3D] cache),
    allow (int32 bits [B, Hm, N, NW])}; d: dict with the scalar fields and tensors of sam_decode_desc.  Raises SamHipError(UNSUPPORTED) for shapes the
    kernel is not built for."""
    import ctypes as C
    arr = (capi.DecodeLayer * len(layers))()
    for e, l in zip(arr, layers):
        for k in ("wqkv", "wo", "w1", "w2"):
            _chk(l[k], BF16, k)
            if l[k].dim() != 4 or l[k].shape[2:] != (16, 8):
                raise capi.SamHipError("greedy_decode_steps: %s must be fragment-tiled (ops.tile_weight)" % k)
        _chk(l["qkv"], BF16, "qkv")
        for k in ("wqkv", "wo", "w1", "w2", "bqkv", "bo", "b1", "b2", "ln1_g", "ln1_b", "ln2_g", "ln2_b", "qkv", "allow"):
            setattr(e, k, l[k].data_ptr())
        al = l["allow"]
        e.allow_stride_b, e.allow_stride_h = al.stride(0), (0 if al.shape[1] == 1 else al.stride(1))
    desc = capi.DecodeDesc()
    for k in ("n_layers", "B", "N", "n_enc", "S", "H", "D", "F", "V", "No"):
        setattr(desc, k, int(d[k]))
    desc.t_begin, desc.t_end = int(t_begin), int(t_end)
    for k in ("scale", "ln_eps", "emb_ln_eps", "ptr_scale"):
        setattr(desc, k, float(d[k]))
    desc.layers = arr
    for k in ("pos_emb", "type_emb", "emb_ln_g", "emb_ln_b", "bc", "bq", "fixed_scores", "ocr_scores"):
        _chk(d[k], torch.float32, k)
    for k in ("ans_ln", "ocr_ln", "wc", "wq", "ptr_k"):
        _chk(d[k], BF16, k)
    _chk(d["prev_inds"], torch.int64, "prev_inds"); _chk(d["ocr_mask"], torch.uint8, "ocr_mask")
    for k in ("pos_emb", "type_emb", "emb_ln_g", "emb_ln_b", "ans_ln", "ocr_ln", "wc", "bc", "wq", "bq", "ptr_k", "ocr_mask", "prev_inds", "fixed_scores", "ocr_scores"):
        setattr(desc, k, d[k].data_ptr())
    desc.seq_out = d["seq_out"].data_ptr() if d.get("seq_out") is not None else None
    for k in ("wc", "wq"):
        if d[k].dim() != 4 or d[k].shape[2:] != (16, 8):
            raise capi.SamHipError("greedy_decode_steps: %s must be fragment-tiled (ops.tile_weight)" % k)
    desc.ld_pos, desc.ld_type, desc.ld_fixed = d["pos_emb"].stride(0), d["type_emb"].stride(0), int(d["ld_fixed"])
    capi.call("sam_greedy_decode_steps", C.cast(C.pointer(desc), C.c_void_p), capi.ptr(ws), ws.numel() * ws.element_size(), capi.stream_handle())


def step_advance(rng_state, offset_stride, step_counter, base_lrs, dev_sched, betas=(0.9, 0.999), warmup_iters=1000, warmup_factor=0.2,
                 lr_decay_iters=(14000, 19000), lr_decay=0.1):
    """head node of a captured training step (sam_step_advance): rng_state[1] += offset_stride; t = ++step_counter[0];
    dev_sched = [base_lr * lambda(t - 1) per segment, 1 - beta1^t, 1 - beta2^t], all in device memory"""
    t_ = _tops()
    if t_ is not None:
        t_.step_advance(rng_state, int(offset_stride), step_counter, [float(l) for l in base_lrs], int(warmup_iters), float(warmup_factor),
                        [int(i) for i in lr_decay_iters], float(lr_decay), float(betas[0]), float(betas[1]), dev_sched)
        return
    import ctypes as C
    sc = capi.LrSchedule()
    sc.nseg = len(base_lrs)
    for i, l in enumerate(base_lrs):
        sc.base_lr[i] = float(l)
    sc.warmup_iters, sc.warmup_factor, sc.n_decay, sc.lr_decay = int(warmup_iters), float(warmup_factor), len(lr_decay_iters), float(lr_decay)
    for i, it in enumerate(lr_decay_iters):
        sc.decay_iters[i] = int(it)
    sc.beta1, sc.beta2 = float(betas[0]), float(betas[1])
    capi.call("sam_step_advance", capi.ptr(rng_state), int(offset_stride), capi.ptr(step_counter), C.cast(C.pointer(sc), C.c_void_p), capi.ptr(dev_sched),
              capi.stream_handle())


def set_cu_reserve(n):
    """CUs withheld from every persistent grid of the training step (include/sam_hip.h: sam_set_cu_reserve); returns the value in force (a multiple of 8).
    Grids already captured into a hipGraph keep what they were captured with: set it BEFORE the Trainer's first captured step."""
    capi.call("sam_set_cu_reserve", int(n))
    return cu_reserve()


def cu_reserve():
    return int(capi.call("sam_get_cu_reserve"))


def debug_cu_hog(blocks, microseconds, stream=None):
    """measurement aid: `blocks` workgroups each holding one CU (64 KB of LDS) for `microseconds` on `stream` (default: the current one)"""
    st = capi.stream_handle() if stream is None else capi.C.c_void_p(stream.cuda_stream)
    capi.call("sam_debug_cu_hog", int(blocks), float(microseconds), st)


def set_rng_state(state):
    """state: uint64-as-int64 [2] device tensor {seed, offset base} (or None): see sam_set_rng_state in include/sam_hip.h"""
    capi.call("sam_set_rng_state", capi.ptr(state))


def cast_bf16(x, y):
    capi.call("sam_cast_f32_to_bf16", capi.ptr(x), capi.ptr(y), x.numel(), capi.stream_handle())
    return y


def embedding_bwd(dy, idx, grad_table, padding_idx=-1):
    """grad_table[idx[t], :] += dy[t, :]  (dy bf16 [T,D], idx int64 [T], grad_table fp32 [rows, D]); padding_idx rows are skipped.
    grad_table._sam_touched (uint8 [rows], set by the Trainer): the rows that receive a gradient are flagged (row-sparse optimizer)"""
    t, d = dy.shape
    capi.call("sam_embedding_bwd", capi.ptr(dy), dy.stride(0), capi.ptr(idx), t, d, grad_table.shape[0], int(padding_idx), capi.ptr(grad_table), grad_table.stride(0),
              capi.ptr(getattr(grad_table, "_sam_touched", None)), capi.stream_handle())


def embedding_bwd_sorted(dy, idx_sorted, grad_table, padding_idx=-1):
    """embedding_bwd for an index list sorted ascending: fixed summation order, one writer per table row (no atomics)"""
    t, d = dy.shape
    capi.call("sam_embedding_bwd_sorted", capi.ptr(dy), dy.stride(0), capi.ptr(idx_sorted), t, d, grad_table.shape[0], int(padding_idx), capi.ptr(grad_table),
              grad_table.stride(0), capi.ptr(getattr(grad_table, "_sam_touched", None)), capi.stream_handle())


def mask_bits_from_int8_bhnn(rel, base_bits=None):
    """rel int8 [B,H,N,N] (non-zero = visible), optional base bits [B,1,N,NW] -> uint32 [B,H,N,NW]"""
    _chk(rel, torch.int8, "rel")
    b, h, n, n2 = rel.shape
    t_ = _tops()
    if t_ is not None:
        return t_.pack_relations_bhnn(rel, base_bits)
    nw = words_per_row(n)
    out = torch.empty((b, h, n, nw), dtype=torch.int32, device=rel.device)
    capi.call("sam_mask_bits_from_int8_bhnn", capi.ptr(rel), capi.ptr(base_bits), b, h, n, nw, capi.ptr(out), capi.stream_handle())
    return out


def spatial_relation_tensor(boxes, context=3, distance_threshold=0.5):
    """boxes f64 [B,N,4] on the GPU -> int8 [B,N,N,12] (HIP kernel; same semantics as spatial_graph.relation_tensor)"""
    _chk(boxes, torch.float64, "boxes")
    b, n, _ = boxes.shape
    out = torch.empty((b, n, n, 12), dtype=torch.int8, device=boxes.device)
    capi.call("sam_spatial_relation_tensor", capi.ptr(boxes), b, n, int(context), float(distance_threshold), capi.ptr(out), capi.stream_handle())
    return out


_GROUPED_WS = {}


def _grouped_ws(device, nbytes):
    """exchange workspace of sam_gemm_bf16_grouped: one zero-filled buffer per (device, stream), grown on demand.  The kernel leaves its flag
    words zero, so the buffer is filled once; launches on one stream are ordered, so they can share it."""
    key = (device.index, torch.cuda.current_stream(device).cuda_stream)
    buf = _GROUPED_WS.get(key)
    if buf is None or buf.numel() * 4 < nbytes:
        buf = torch.zeros(((nbytes + 3) // 4,), dtype=torch.float32, device=device)
        _GROUPED_WS[key] = buf
    return buf


def grouped_ws_check():
    """raise if any pair exchange of sam_gemm_bf16_grouped gave up waiting for its partner block (error word of the workspace; synchronises)"""
    for key, buf in _GROUPED_WS.items():
        if int(buf[:1].view(torch.int32).item()) != 0:
            buf[:1].zero_()
            raise capi.SamHipError("sam_gemm_bf16_grouped: a workgroup pair never became co-resident (device %s); the step's weight gradients are invalid" % (key[0],))


def reset_workspaces():
    """forget every cached scratch buffer (after a failed step: a grouped-wgrad workspace may hold stale pair flags, an LN scratch stale partials);
    the next call allocates fresh, zero-filled ones"""
    _GROUPED_WS.clear()
    _WS.clear()
    _LN_XWS.clear()
    LnFinalizeQueue.clear()


def wgrad_grouped(jobs, force_tile=0, accumulate=True):
    """jobs: list of (dy [R,M] bf16, x [R,N] bf16, dW fp32 [M,N] view, dbias fp32 [M] or None).  dW += dy^T x (and dbias += colsum(dy))
    for all jobs in ONE launch (sam_gemm_bf16_grouped); accumulate=False: both are overwritten instead (no read of the old gradient)."""
    n = len(jobs)
    arr = (capi.GemmDesc * n)()
    flops = 0.0
    for d, job in zip(arr, jobs):
        dy, x, dw, db = job[:4]
        acc_j = job[4] if len(job) > 4 else accumulate          # (a job may carry its own flag: problems that accumulate next to problems that overwrite)
        d.M, d.N, d.K = dy.shape[1], x.shape[1], dy.shape[0]
        d.a_kcontig = d.b_kcontig = 0
        d.c_is_f32, d.accumulate, d.epilogue = 1, int(bool(acc_j)), capi.EPI_NONE
        d.A, d.lda, d.B, d.ldb, d.C, d.ldc = dy.data_ptr(), dy.stride(0), x.data_ptr(), x.stride(0), dw.data_ptr(), dw.stride(0)
        d.bias_grad = _dp(db)
        flops += 2.0 * d.M * d.N * d.K
    if force_tile != 128:
        ws = _grouped_ws(jobs[0][0].device, int(capi.lib().sam_gemm_grouped_ws_bytes(arr, n)))
        arr[0].ws, arr[0].ws_bytes = ws.data_ptr(), ws.numel() * 4
    arr[0].force_tile = force_tile
    capi.call("sam_gemm_bf16_grouped", arr, n, capi.stream_handle(), meta=dict(kernel="gemm_grouped_wgrad", flops=flops, shape=(n, int(arr[0].K))))
