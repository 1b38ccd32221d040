// Shared pieces of the bf16 GEMM family (gemm.hip: 4-wave kernels; gemm8.hip: 8-wave persistent kernels): argument block, output stores,
// and the fused epilogue.  Internal to the library (the C ABI is include/sam_hip.h).
#pragma once
#include <type_traits>
#include "common.h"
#include "sam_hip.h"
#ifndef SAM_GEMM_SC1_STORES
#define SAM_GEMM_SC1_STORES 0   // measured: no gain (within noise) on any shape of the step; kept as a build-time switch
#endif

namespace samgemm {

constexpr int BK = 64;

struct GemmArgs {
  int M, N, K;
  const bf16_t* A; int64_t lda;
  const bf16_t* B; int64_t ldb;
  void* C; int64_t ldc;
  const float* bias;
  const bf16_t* residual; int64_t ldr;
  bf16_t* aux_out; const bf16_t* aux_in; int64_t ld_aux;
  int accumulate;
  unsigned thr16; float inv_keep;
  unsigned seed_lo, seed_hi, off_lo, off_hi;
  const unsigned long long* rng_state;   // device-side {seed, offset base} (hipGraph replays), or NULL
  int tiles_m, tiles_n, group_m;
  int split_k;        // >1: grid = tiles * split_k; split s stores its fp32 partial tile into ws[s] (wgrad: few tiles, very long K)
  float* ws;          // [split_k][M*N] partial outputs, then [split_k][M] partial bias gradients; reduced by splitk_reduce_kernel
  float* bias_grad;   // wgrad only: bias_grad[m] += sum_k A(m,k)  (column sums of dy), from the A tile already in LDS
  int defer_reduce;   // split-K: leave the partials in ws, the caller runs sam_gemm_splitk_reduce itself
  int* split_used;    // host pointer: receives the split factor actually launched
  int stagger;        // gemm4 kernels: start delay of the second block slot of every CU, in units of ~3.7 us (s_sleep 127)
  // LayerNorm inside the launch (gemm12.hip, round 6): ln_y != NULL = after its epilogue every compute wave normalises the rows of its own sub-tile; the row
  // statistics are exchanged between the waves / blocks that share a row through ln_ws (gemm_ln_pass below).  Host-side: *ln_done = 1 when the launch did it.
  const float* ln_gamma; const float* ln_beta; float ln_eps;
  bf16_t* ln_y; int64_t ln_ldy; float* ln_mean; float* ln_rstd;
  float* ln_ws; int* ln_done;
  int dbg;            // tuning experiments only (SAM_GEMM8_DBG; results are garbage): 1 = gemm8 kernels skip the epilogue, 2 = every tile's epilogue
                      // lands on the first tile row (outputs / residual / auxiliary rows stay in the L2: the epilogue without its HBM traffic)
};

template <typename OutT> struct Store4;
template <> struct Store4<bf16_t> {
  static __device__ __forceinline__ void st(void* C, int64_t idx, const float* v, int) {
    const uint2 val = make_uint2(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]));
#if SAM_GEMM_SC1_STORES
    // write-through store: the output tile is never re-read by this kernel, keep it from evicting operand panels in the XCD's L2
    const bf16_t* addr = reinterpret_cast<bf16_t*>(C) + idx;
    asm volatile("global_store_dwordx2 %0, %1, off sc1" ::"v"(addr), "v"(val) : "memory");
#else
    *reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(C) + idx) = val;
#endif
  }
};
template <> struct Store4<float> {
  // accumulate: 0 = store, 1 = read-modify-write (every element has exactly one writer)
  static __device__ __forceinline__ void st(void* C, int64_t idx, const float* v, int accumulate) {
    float4* p = reinterpret_cast<float4*>(reinterpret_cast<float*>(C) + idx);
    float4 o = make_float4(v[0], v[1], v[2], v[3]);
    if (accumulate) { const float4 c = *p; o.x += c.x; o.y += c.y; o.z += c.z; o.w += c.w; }
    *p = o;
  }
};

// Fused epilogue of one wave's TM x TN fragment grid.  acc[tn][tm] comes from mfma(B fragment, A fragment): lane (i = lane & 15, g = lane >> 4)
// owns row m = mw + tm*16 + i and the four consecutive columns n = nw + tn*16 + 4g .. +3.
// Every global operand of the epilogue (bias, residual, GELU pre-activation, old C) is fetched up front for ALL fragments, unconditionally
// at clamped addresses: a load under a per-lane predicate compiles to branch + s_waitcnt vmcnt(0), i.e. TM*TN dependent HBM round
// trips per wave instead of one.  Interior tiles (FULL) run with no per-lane predicate at all: under a predicate the compiler sinks each
// fragment's arithmetic into the guarded block and opens it with s_waitcnt vmcnt(0), which also waits for the PREVIOUS fragment's
// store -- TM*TN serialized store round trips.  Edge tiles take the same code with clamped loads and guarded stores.
// (rows are handled in fragment-row ranges [T0, T1): the 8-wave kernels with 32 fragments per wave run two halves to bound the prefetch registers)
template <int TM, int TN, int EPI, typename OutT, bool FULL, int T0 = 0, int T1 = TM, int N0 = 0, int N1 = TN>
__device__ __forceinline__ void gemm_epilogue_impl(const GemmArgs& p, const f32x4 (&acc)[TN][TM], int mw, int nw, void* Cout, int64_t ldc, int accumulate, int i, int g) {
  const int n_last = max(p.N - 4, 0), m_last = p.M - 1;
  unsigned seed_lo = p.seed_lo, seed_hi = p.seed_hi, off_lo = p.off_lo, off_hi = p.off_hi;
  if (EPI == SAM_EPI_BIAS_DROPOUT_RES) rng_resolve(p.rng_state, seed_lo, seed_hi, off_lo, off_hi);
  constexpr bool HAS_BIAS = EPI == SAM_EPI_BIAS || EPI == SAM_EPI_BIAS_GELU || EPI == SAM_EPI_BIAS_DROPOUT_RES || EPI == SAM_EPI_BIAS_GELU_GRAD;
  constexpr bool HAS_PRE = EPI == SAM_EPI_DGELU || EPI == SAM_EPI_BIAS_DROPOUT_RES || EPI == SAM_EPI_MUL_AUX;
  float4 b4[TN];
  uint2 pre[HAS_PRE ? TM : 1][HAS_PRE ? TN : 1];   // (only rows [T0, T1) are touched: the rest is never materialised)
  if (HAS_BIAS) {
#pragma unroll
    for (int tn = N0; tn < N1; ++tn) {
      b4[tn] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (p.bias) b4[tn] = *reinterpret_cast<const float4*>(p.bias + (FULL ? nw + tn * 16 + 4 * g : min(nw + tn * 16 + 4 * g, n_last)));
    }
  }
  if (HAS_PRE) {
    const bf16_t* src = EPI == SAM_EPI_BIAS_DROPOUT_RES ? p.residual : p.aux_in;
    const int64_t lds_ = EPI == SAM_EPI_BIAS_DROPOUT_RES ? p.ldr : p.ld_aux;
#pragma unroll
    for (int tm = T0; tm < T1; ++tm)
#pragma unroll
      for (int tn = N0; tn < N1; ++tn) pre[tm][tn] = make_uint2(0u, 0u);
    if (src)
#pragma unroll
      for (int tm = T0; tm < T1; ++tm)
#pragma unroll
        for (int tn = N0; tn < N1; ++tn)
          pre[tm][tn] = *reinterpret_cast<const uint2*>(src + (int64_t)(FULL ? mw + tm * 16 + i : min(mw + tm * 16 + i, m_last)) * lds_ +
                                                        (FULL ? nw + tn * 16 + 4 * g : min(nw + tn * 16 + 4 * g, n_last)));
  }
  constexpr bool F32_OUT = sizeof(OutT) == 4 && (T1 - T0) * TN <= 16;   // (the 256-wide test tiles keep the per-fragment read-modify-write: no registers left)
  float4 cpre[F32_OUT ? TM : 1][F32_OUT ? TN : 1];    // accumulate=1 (wgrad into the gradient buffer): the old C values, same batching
  if (F32_OUT && accumulate) {
#pragma unroll
    for (int tm = T0; tm < T1; ++tm)
#pragma unroll
      for (int tn = N0; tn < N1; ++tn)
        cpre[F32_OUT ? tm : 0][F32_OUT ? tn : 0] = *reinterpret_cast<const float4*>(
            reinterpret_cast<const float*>(Cout) + (int64_t)(FULL ? mw + tm * 16 + i : min(mw + tm * 16 + i, m_last)) * ldc + (FULL ? nw + tn * 16 + 4 * g : min(nw + tn * 16 + 4 * g, n_last)));
  }
#pragma unroll
  for (int tm = T0; tm < T1; ++tm) {
    const int m = mw + tm * 16 + i;
#pragma unroll
    for (int tn = N0; tn < N1; ++tn) {
      const int n = nw + tn * 16 + 4 * g;
      float v[4] = {acc[tn][tm][0], acc[tn][tm][1], acc[tn][tm][2], acc[tn][tm][3]};
      if (F32_OUT && accumulate) {
        const float4 c = cpre[F32_OUT ? tm : 0][F32_OUT ? tn : 0];
        v[0] += c.x; v[1] += c.y; v[2] += c.z; v[3] += c.w;
      }
      if (HAS_BIAS) { v[0] += b4[tn].x; v[1] += b4[tn].y; v[2] += b4[tn].z; v[3] += b4[tn].w; }
      if (EPI == SAM_EPI_BIAS_GELU) {
        if (FULL || (m < p.M && n < p.N))
          *reinterpret_cast<uint2*>(p.aux_out + (int64_t)m * p.ld_aux + n) = make_uint2(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]));
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = gelu_erf(v[r]);
      }
      if (EPI == SAM_EPI_BIAS_GELU_GRAD) {
        float d[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = gelu_erf_and_grad(v[r], d[r]);
        if (p.aux_out && (FULL || (m < p.M && n < p.N)))          // (aux_out NULL: inference, the derivative is not wanted)
          *reinterpret_cast<uint2*>(p.aux_out + (int64_t)m * p.ld_aux + n) = make_uint2(pack_bf16x2(d[0], d[1]), pack_bf16x2(d[2], d[3]));
      }
      if (EPI == SAM_EPI_DGELU) {
        const uint2 x = pre[HAS_PRE ? tm : 0][HAS_PRE ? tn : 0];
        v[0] *= gelu_erf_grad(bf_lo(x.x)); v[1] *= gelu_erf_grad(bf_hi(x.x));
        v[2] *= gelu_erf_grad(bf_lo(x.y)); v[3] *= gelu_erf_grad(bf_hi(x.y));
      }
      if (EPI == SAM_EPI_MUL_AUX) {
        const uint2 x = pre[HAS_PRE ? tm : 0][HAS_PRE ? tn : 0];
        v[0] *= bf_lo(x.x); v[1] *= bf_hi(x.x); v[2] *= bf_lo(x.y); v[3] *= bf_hi(x.y);
      }
      if (EPI == SAM_EPI_BIAS_DROPOUT_RES) {
        if (p.thr16) {
          const u32x4 rn = hidden_dropout_bits((unsigned)m, (unsigned)(n >> 3), off_lo, off_hi, seed_lo, seed_hi);
          const unsigned lo = (n & 4) ? rn.z : rn.x, hi = (n & 4) ? rn.w : rn.y;
          v[0] = (lo & 0xffffu) >= p.thr16 ? v[0] * p.inv_keep : 0.f;
          v[1] = (lo >> 16) >= p.thr16 ? v[1] * p.inv_keep : 0.f;
          v[2] = (hi & 0xffffu) >= p.thr16 ? v[2] * p.inv_keep : 0.f;
          v[3] = (hi >> 16) >= p.thr16 ? v[3] * p.inv_keep : 0.f;
        }
        const uint2 x = pre[HAS_PRE ? tm : 0][HAS_PRE ? tn : 0];   // zeros when there is no residual
        v[0] += bf_lo(x.x); v[1] += bf_hi(x.x); v[2] += bf_lo(x.y); v[3] += bf_hi(x.y);
      }
      if (FULL || (m < p.M && n < p.N)) Store4<OutT>::st(Cout, (int64_t)m * ldc + n, v, F32_OUT ? 0 : accumulate);
    }
  }
}
template <int TM, int TN, int EPI, typename OutT, int T0 = 0, int T1 = TM>
__device__ __forceinline__ void gemm_epilogue(const GemmArgs& p, const f32x4 (&acc)[TN][TM], int mw, int nw, bool full, void* Cout, int64_t ldc, int accumulate, int i, int g) {
  if (full) gemm_epilogue_impl<TM, TN, EPI, OutT, true, T0, T1>(p, acc, mw, nw, Cout, ldc, accumulate, i, g);
  else gemm_epilogue_impl<TM, TN, EPI, OutT, false, T0, T1>(p, acc, mw, nw, Cout, ldc, accumulate, i, g);
}

// ---- 8-wide epilogue (8-wave kernels).  A row-per-lane epilogue that stores 8 bytes per lane writes 32-byte row segments and is bound by
// store ISSUE, not bandwidth (a 192x192 bf16 tile took ~6 us to leave through 18 dwordx2 stores per lane: a third of a 12-k-tile tile).
// v_permlane16_swap exchanges the odd 16-lane rows of one register with the even rows of another: applied to the accumulators of the column
// fragment pair (tn, tn+1), lane (i, g) ends up with EIGHT consecutive columns n = nw + 16 tn + 16 (g & 1) + 8 (g >> 1) .. +7 of row
// m = mw + 16 tm + i.  Everything downstream is then 16 bytes wide: residual / pre-activation loads, one Philox call per 8 outputs (the
// (row, col/8) stream of the 4-wide epilogue, which computed it twice), and dwordx4 stores covering 64 contiguous bytes per row.
// An odd last column fragment goes through the 4-wide code.
template <int TM, int TN, int EPI, typename OutT, bool FULL, int T0, int T1>
__device__ __forceinline__ void gemm_epilogue8_impl(const GemmArgs& p, const f32x4 (&acc)[TN][TM], int mw, int nw, void* Cout, int64_t ldc, int accumulate, int i, int g) {
  constexpr int NP = TN / 2;
  constexpr bool HAS_BIAS = EPI == SAM_EPI_BIAS || EPI == SAM_EPI_BIAS_GELU || EPI == SAM_EPI_BIAS_DROPOUT_RES || EPI == SAM_EPI_BIAS_GELU_GRAD;
  constexpr bool HAS_PRE = EPI == SAM_EPI_DGELU || EPI == SAM_EPI_BIAS_DROPOUT_RES || EPI == SAM_EPI_MUL_AUX;
  const int n_last = max(p.N - 8, 0), m_last = p.M - 1;
  unsigned seed_lo = p.seed_lo, seed_hi = p.seed_hi, off_lo = p.off_lo, off_hi = p.off_hi;
  if (EPI == SAM_EPI_BIAS_DROPOUT_RES) rng_resolve(p.rng_state, seed_lo, seed_hi, off_lo, off_hi);
  const int ncol = nw + 16 * (g & 1) + 8 * (g >> 1);
  float4 b4[NP][2];
  uint4 pre[HAS_PRE ? T1 - T0 : 1][HAS_PRE ? NP : 1];
  if (HAS_BIAS) {
#pragma unroll
    for (int q = 0; q < NP; ++q) {
      b4[q][0] = b4[q][1] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (p.bias) {
        const float* bp = p.bias + (FULL ? ncol + 32 * q : min(ncol + 32 * q, n_last));
        b4[q][0] = *reinterpret_cast<const float4*>(bp);
        b4[q][1] = *reinterpret_cast<const float4*>(bp + 4);
      }
    }
  }
  if (HAS_PRE) {
    const bf16_t* src = EPI == SAM_EPI_BIAS_DROPOUT_RES ? p.residual : p.aux_in;
    const int64_t lds_ = EPI == SAM_EPI_BIAS_DROPOUT_RES ? p.ldr : p.ld_aux;
#pragma unroll
    for (int tm = T0; tm < T1; ++tm)
#pragma unroll
      for (int q = 0; q < NP; ++q) pre[tm - T0][q] = make_uint4(0u, 0u, 0u, 0u);
    if (src)
#pragma unroll
      for (int tm = T0; tm < T1; ++tm)
#pragma unroll
        for (int q = 0; q < NP; ++q)
          pre[tm - T0][q] = *reinterpret_cast<const uint4*>(src + (int64_t)(FULL ? mw + tm * 16 + i : min(mw + tm * 16 + i, m_last)) * lds_ +
                                                            (FULL ? ncol + 32 * q : min(ncol + 32 * q, n_last)));
  }
#pragma unroll
  for (int tm = T0; tm < T1; ++tm) {
    const int m = mw + tm * 16 + i;
#pragma unroll
    for (int q = 0; q < NP; ++q) {
      const int n = ncol + 32 * q;
      float v[8];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const auto sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(acc[2 * q][tm][r]), __float_as_uint(acc[2 * q + 1][tm][r]), false, false);
        v[r] = __uint_as_float(sw[0]);
        v[4 + r] = __uint_as_float(sw[1]);
      }
      const bool ok = FULL || (m < p.M && n < p.N);
      if (sizeof(OutT) == 4 && accumulate) {
        if (ok) {
          const float4 c0 = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(Cout) + (int64_t)m * ldc + n);
          const float4 c1 = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(Cout) + (int64_t)m * ldc + n + 4);
          v[0] += c0.x; v[1] += c0.y; v[2] += c0.z; v[3] += c0.w; v[4] += c1.x; v[5] += c1.y; v[6] += c1.z; v[7] += c1.w;
        }
      }
      if (HAS_BIAS) {
        v[0] += b4[q][0].x; v[1] += b4[q][0].y; v[2] += b4[q][0].z; v[3] += b4[q][0].w;
        v[4] += b4[q][1].x; v[5] += b4[q][1].y; v[6] += b4[q][1].z; v[7] += b4[q][1].w;
      }
      if (EPI == SAM_EPI_BIAS_GELU) {
        if (ok)
          *reinterpret_cast<uint4*>(p.aux_out + (int64_t)m * p.ld_aux + n) =
              make_uint4(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7]));
#pragma unroll
        for (int r = 0; r < 8; ++r) v[r] = gelu_erf(v[r]);
      }
      if (EPI == SAM_EPI_BIAS_GELU_GRAD) {
        float d[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) v[r] = gelu_erf_and_grad(v[r], d[r]);
        if (ok && p.aux_out)
          *reinterpret_cast<uint4*>(p.aux_out + (int64_t)m * p.ld_aux + n) =
              make_uint4(pack_bf16x2(d[0], d[1]), pack_bf16x2(d[2], d[3]), pack_bf16x2(d[4], d[5]), pack_bf16x2(d[6], d[7]));
      }
      if (EPI == SAM_EPI_MUL_AUX) {
        const uint4 x = pre[HAS_PRE ? tm - T0 : 0][HAS_PRE ? q : 0];
        v[0] *= bf_lo(x.x); v[1] *= bf_hi(x.x); v[2] *= bf_lo(x.y); v[3] *= bf_hi(x.y);
        v[4] *= bf_lo(x.z); v[5] *= bf_hi(x.z); v[6] *= bf_lo(x.w); v[7] *= bf_hi(x.w);
      }
      if (EPI == SAM_EPI_DGELU) {
        const uint4 x = pre[HAS_PRE ? tm - T0 : 0][HAS_PRE ? q : 0];
        v[0] *= gelu_erf_grad(bf_lo(x.x)); v[1] *= gelu_erf_grad(bf_hi(x.x)); v[2] *= gelu_erf_grad(bf_lo(x.y)); v[3] *= gelu_erf_grad(bf_hi(x.y));
        v[4] *= gelu_erf_grad(bf_lo(x.z)); v[5] *= gelu_erf_grad(bf_hi(x.z)); v[6] *= gelu_erf_grad(bf_lo(x.w)); v[7] *= gelu_erf_grad(bf_hi(x.w));
      }
      if (EPI == SAM_EPI_BIAS_DROPOUT_RES) {
        if (p.thr16) {
          const u32x4 rn = hidden_dropout_bits((unsigned)m, (unsigned)(n >> 3), off_lo, off_hi, seed_lo, seed_hi);
          const unsigned w4[4] = {rn.x, rn.y, rn.z, rn.w};
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            v[2 * r] = (w4[r] & 0xffffu) >= p.thr16 ? v[2 * r] * p.inv_keep : 0.f;
            v[2 * r + 1] = (w4[r] >> 16) >= p.thr16 ? v[2 * r + 1] * p.inv_keep : 0.f;
          }
        }
        const uint4 x = pre[HAS_PRE ? tm - T0 : 0][HAS_PRE ? q : 0];   // zeros when there is no residual
        v[0] += bf_lo(x.x); v[1] += bf_hi(x.x); v[2] += bf_lo(x.y); v[3] += bf_hi(x.y);
        v[4] += bf_lo(x.z); v[5] += bf_hi(x.z); v[6] += bf_lo(x.w); v[7] += bf_hi(x.w);
      }
      if (ok) {
        if (sizeof(OutT) == 4) {
          float* dst = reinterpret_cast<float*>(Cout) + (int64_t)m * ldc + n;
          *reinterpret_cast<float4*>(dst) = make_float4(v[0], v[1], v[2], v[3]);
          *reinterpret_cast<float4*>(dst + 4) = make_float4(v[4], v[5], v[6], v[7]);
        } else {
          *reinterpret_cast<uint4*>(reinterpret_cast<bf16_t*>(Cout) + (int64_t)m * ldc + n) =
              make_uint4(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7]));
        }
      }
    }
  }
  if constexpr (TN % 2 == 1) gemm_epilogue_impl<TM, TN, EPI, OutT, FULL, T0, T1, TN - 1, TN>(p, acc, mw, nw, Cout, ldc, accumulate, i, g);
}
template <int TM, int TN, int EPI, typename OutT, int T0 = 0, int T1 = TM>
__device__ __forceinline__ void gemm_epilogue8(const GemmArgs& p, const f32x4 (&acc)[TN][TM], int mw, int nw, bool full, void* Cout, int64_t ldc, int accumulate, int i, int g) {
  if (full) gemm_epilogue8_impl<TM, TN, EPI, OutT, true, T0, T1>(p, acc, mw, nw, Cout, ldc, accumulate, i, g);
  else gemm_epilogue8_impl<TM, TN, EPI, OutT, false, T0, T1>(p, acc, mw, nw, Cout, ldc, accumulate, i, g);
}

// ---- LayerNorm inside an MMT-size GEMM launch (linear_bias_dropout_residual_ln as ONE kernel; sa_m4c.py:653, 680, 1016-1028) -------------------------
// The launch is ONE round of tiles (tiles <= blocks, every block resident) of BM x BN with N a multiple of BN: a row of the output is spread over tiles_n blocks x 4
// waves.  After the epilogue has stored z = bf16(dropout(x W^T + b) + residual) -- what the backward reads -- every compute wave
//   1. re-reads ITS OWN (BM/2) x (BN/4) sub-tile of z (it wrote it: one s_waitcnt away, L2-resident), 4 lanes per row;
//   2. reduces each row's BN/4 values to (mean_i, M2_i) -- two passes, in registers -- and publishes the pair (agent-scope store) in ln_ws[row][part];
//   3. counts itself in on the (row block, row half) counter and waits for the P = 4 tiles_n parts of its rows (bounded spin: see gemm8w.hip);
//   4. merges the P pairs in a fixed order (equal counts: mean = avg(mean_i), M2 = sum M2_i + n sum (mean_i - mean)^2), normalises the values it still holds and
//      stores y; part 0 also stores mean / rstd; the last wave to leave a counter resets it (the next launch finds zeros again).
// No block-wide barrier (the loader waves of gemm12_kernel have returned by then), no atomics on data, bit-reproducible.  The statistics differ from the
// stand-alone kernel's (one 768-term sum per row) by fp32 rounding only; y within one bf16 ulp (tests/test_gemm_gpu.py).
// Workspace layout (floats): [0] error word | [64 ..) counters: (arrive, depart) per (tile_m, half) | [LN_WS_PART0 ..) pairs [row][part].
constexpr int LN_WS_COUNTERS = 64, LN_WS_PART0 = 64 + 4 * 1024;          // up to 1024 row tiles
constexpr int AUX_SC1_LN = 0x10;

template <int BM, int BN>
__device__ __forceinline__ void gemm_ln_pass(const GemmArgs& p, int m0, int n0, int wr, int wc, int lane) {
  constexpr int RW = BM / 2, CW = BN / 4, TR = RW / 16, NPC = CW / 16;         // rows / columns per wave, 16-row groups, 16-column pieces (4 columns per lane each)
  const int r16 = lane >> 2, q = lane & 3;
  const int tile_m = m0 / BM, tile_n = n0 / BN, P = 4 * p.tiles_n, part = 4 * tile_n + wc;
  const bf16_t* z = reinterpret_cast<const bf16_t*>(p.C);
  if (p.dbg & 32) return;                                                         // (tuning: the launch with the pass switched off inside the kernel)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                               // this wave's epilogue stores have reached the L2
  typedef unsigned lnvu2 __attribute__((ext_vector_type(2)));
  const __amdgpu_buffer_rsrc_t rz = __builtin_amdgcn_make_buffer_rsrc((void*)z, 0, 0x7fffffff, 0x00020000);
  uint2 zr[TR][NPC];
#pragma unroll
  for (int t = 0; t < TR; ++t) {
    const int row = min(m0 + wr * RW + 16 * t + r16, p.M - 1);
#pragma unroll
    for (int c = 0; c < NPC; ++c) {
      const lnvu2 u = __builtin_amdgcn_raw_buffer_load_b64(rz, (unsigned)(((int64_t)row * p.ldc + n0 + wc * CW + 16 * c + 4 * q) * 2), 0, 0);
      zr[t][c] = make_uint2(u[0], u[1]);
    }
  }
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(p.ln_ws + LN_WS_PART0), 0, 0x7fffffff, 0x00020000);
  // ---- 2. this wave's part of every row
#pragma unroll
  for (int t = 0; t < TR; ++t) {
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < NPC; ++c) s += (bf_lo(zr[t][c].x) + bf_hi(zr[t][c].x)) + (bf_lo(zr[t][c].y) + bf_hi(zr[t][c].y));
    s += __shfl_xor(s, 1);
    s += __shfl_xor(s, 2);
    const float mi = s * (1.0f / CW);
    float m2 = 0.f;
#pragma unroll
    for (int c = 0; c < NPC; ++c) {
      const float d0 = bf_lo(zr[t][c].x) - mi, d1 = bf_hi(zr[t][c].x) - mi, d2 = bf_lo(zr[t][c].y) - mi, d3 = bf_hi(zr[t][c].y) - mi;
      m2 += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
    }
    m2 += __shfl_xor(m2, 1);
    m2 += __shfl_xor(m2, 2);
    const int row = m0 + wr * RW + 16 * t + r16;
    if (q == 0 && row < p.M) {
      const lnvu2 pr = {__float_as_uint(mi), __float_as_uint(m2)};
      __builtin_amdgcn_raw_buffer_store_b64(pr, rs, (unsigned)(((int64_t)row * P + part) * 8), 0, AUX_SC1_LN);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  // ---- 3. count in, wait for the other parts of these rows
  unsigned* cnt = reinterpret_cast<unsigned*>(p.ln_ws) + LN_WS_COUNTERS + 2 * (2 * tile_m + wr);
  if (lane == 0 && !(p.dbg & 16)) {
    __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned spins = 0;
    while (!(p.dbg & 8) && __hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)P) {
      __builtin_amdgcn_s_sleep(4);
      if (++spins > (1u << 21)) { __hip_atomic_store(reinterpret_cast<unsigned*>(p.ln_ws), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }      // a part never arrived: error word, garbage out, no hang
    }
  }
  __builtin_amdgcn_wave_barrier();
  asm volatile("" ::: "memory");
  // ---- 4. merge, normalise, store
  float g4[NPC][4], b4[NPC][4];
#pragma unroll
  for (int c = 0; c < NPC; ++c) {
    const float4 gv = *reinterpret_cast<const float4*>(p.ln_gamma + n0 + wc * CW + 16 * c + 4 * q), bv = *reinterpret_cast<const float4*>(p.ln_beta + n0 + wc * CW + 16 * c + 4 * q);
    g4[c][0] = gv.x; g4[c][1] = gv.y; g4[c][2] = gv.z; g4[c][3] = gv.w;
    b4[c][0] = bv.x; b4[c][1] = bv.y; b4[c][2] = bv.z; b4[c][3] = bv.w;
  }
  bf16_t* y = p.ln_y;
  // every pair this lane needs is requested BEFORE the first is used (lane q takes parts q, q + 4, ...; P / 4 = tiles_n <= 4 of them per row, at a clamped index
  // with weight 0 past P): agent-scope loads come from the memory side at 1-2 us each, and the first version of this loop -- one dependent pair of loads per
  // trip, 24 trips per lane -- cost 36 us per launch
  lnvu2 pp[TR][4];
#pragma unroll
  for (int t = 0; t < TR; ++t) {
    const int rc = min(m0 + wr * RW + 16 * t + r16, p.M - 1);
#pragma unroll
    for (int u = 0; u < 4; ++u)
      pp[t][u] = __builtin_amdgcn_raw_buffer_load_b64(rs, (unsigned)(((int64_t)rc * P + min(q + 4 * u, P - 1)) * 8), 0, AUX_SC1_LN);
  }
#pragma unroll
  for (int t = 0; t < TR; ++t) {
    const int row = m0 + wr * RW + 16 * t + r16;
    float s1 = 0.f, s2 = 0.f, sm = 0.f;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const float wgt = (q + 4 * u < P) ? 1.f : 0.f;
      const float mi = __uint_as_float(pp[t][u][0]) * wgt, m2 = __uint_as_float(pp[t][u][1]) * wgt;
      s1 += mi; s2 += mi * mi; sm += m2;
    }
    s1 += __shfl_xor(s1, 1); s1 += __shfl_xor(s1, 2);
    s2 += __shfl_xor(s2, 1); s2 += __shfl_xor(s2, 2);
    sm += __shfl_xor(sm, 1); sm += __shfl_xor(sm, 2);
    const float mean = s1 / (float)P;
    const float m2_all = sm + (float)CW * fmaxf(s2 - (float)P * mean * mean, 0.f);
    const float rstd = 1.0f / sqrtf(m2_all / (float)p.N + p.ln_eps);
    if (row < p.M) {
#pragma unroll
      for (int c = 0; c < NPC; ++c) {
        const float v0 = bf_lo(zr[t][c].x), v1 = bf_hi(zr[t][c].x), v2 = bf_lo(zr[t][c].y), v3 = bf_hi(zr[t][c].y);
        uint2 o;
        o.x = pack_bf16x2(g4[c][0] * ((v0 - mean) * rstd) + b4[c][0], g4[c][1] * ((v1 - mean) * rstd) + b4[c][1]);
        o.y = pack_bf16x2(g4[c][2] * ((v2 - mean) * rstd) + b4[c][2], g4[c][3] * ((v3 - mean) * rstd) + b4[c][3]);
        *reinterpret_cast<uint2*>(y + (int64_t)row * p.ln_ldy + n0 + wc * CW + 16 * c + 4 * q) = o;
      }
      if (part == 0 && q == 0) { p.ln_mean[row] = mean; p.ln_rstd[row] = rstd; }
    }
  }
  // ---- leave: the last of the P waves puts the counters back to zero
  if (lane == 0 && !(p.dbg & 16)) {
    const unsigned left = __hip_atomic_fetch_add(cnt + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (left == (unsigned)P - 1) {
      __hip_atomic_store(cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(cnt + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}
// workspace bytes of the in-launch LayerNorm for an M x N output (16 parts per row cover N <= 768 at 192-wide tiles; P = 4 * ceil(N / 192) in general)
inline int64_t gemm_ln_ws_bytes(int M, int N) { return ((int64_t)LN_WS_PART0 + (int64_t)M * (4 * ((N + 191) / 192)) * 2) * 4 + 256; }

constexpr int SAM_MAX_GROUP = 12;        // problems per grouped weight-gradient launch of the 4-wave kernel (an encoder layer has 4; TextBert's three layers go out together)
constexpr int SAM_MAX_GROUP8 = 20;       // ... of the 8-wave kernel (gemm8w.hip): an MMT pair (8) + TextBert's three layers (12) in one launch

// CU count of the CURRENT device (cached per device ordinal: a process that drives several GPUs must not size grids from the first one it queried)
inline int device_cu_count() {
  static int cache[64] = {};
  int dev = 0;
  hipGetDevice(&dev);
  if (dev < 0 || dev >= 64) dev = 0;
  if (cache[dev] == 0) {
    hipDeviceProp_t prop;
    cache[dev] = (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
  }
  return cache[dev];
}

// CUs the persistent grids may fill: the device's count minus the reserve (sam_set_cu_reserve, runtime.cpp).  Every hot kernel of the training step is
// persistent with one block per CU by LDS / registers: a kernel of another stream that holds k CUs for the length of the backward pass -- RCCL's channel
// blocks under data-parallel training -- leaves k blocks of each of them without a CU, and those blocks form a SECOND round: the launch takes twice as
// long, and the grouped weight gradient's pair partners stop being co-resident.  With a reserve of r the grids are sized for (CUs - r): one round, every
// block 256 / (256 - r) longer.  Decoding (no collectives beside it) sizes its grid from device_cu_count().
extern "C" int sam_get_cu_reserve(void);
inline int grid_cu_count() {
  const int n = device_cu_count(), r = sam_get_cu_reserve();
  return (r > 0 && n - r >= 8) ? n - r : n;
}

// 8-wave persistent kernels (gemm8.hip).  tile: 0 = heuristic, 1192 / 1256 / 1448 = force the 192x192 / 256x256 / 192x256 configuration.
// Returns SAM_ERR_UNSUPPORTED (without touching the error string) when the problem or the (layout, epilogue, output type) combination
// has no instance there: the caller then uses the 4-wave kernels.
int gemm8_launch(const GemmArgs& a, int lay, int epilogue, int c_is_f32, int tile, hipStream_t st);
// 12-wave kernels with loader waves (gemm12.hip): 192 x 192 (three stages) / 192 x 256 tiles; tile: 0 = heuristic, 12192 / 12448 = force one.  Same contract.
int gemm12_launch(const GemmArgs& a, int lay, int epilogue, int c_is_f32, int tile, hipStream_t st);
// (the round-3 experiment with two 4-wave blocks per CU, 256 x 128 x 32 tiles, measured 0.78x the 8-wave k-loop: it lives in tools/probes/gemm4.hip, outside the library)
// grouped weight gradients on the 8-wave core (gemm8w.hip): 256x256 tiles, the K range of each tile split over a PAIR of blocks that exchange
// halves inside the launch.  descs[0].ws / ws_bytes: the exchange workspace (gemm8w_ws_bytes(total tiles); its first words are the pair flags,
// which must be zero before the first launch and are left zero by every launch).  SAM_ERR_UNSUPPORTED: not a problem set for this kernel.
int gemm8w_grouped(const sam_gemm_desc* descs, int count, hipStream_t st);
// grouped weight gradients on the loader-wave core (gemm12w.hip): 192 x 256 tiles, one persistent workgroup per CU on a static schedule (whole tiles, then one
// K slice of a left-over tile with an in-launch fixed-order reduction, then the shallow tiles).  Same contract; workspace gemm12w_ws_bytes (0 = none needed / not a set for it).
int gemm12w_grouped(const sam_gemm_desc* descs, int count, hipStream_t st);
int64_t gemm12w_ws_bytes(const sam_gemm_desc* descs, int count);
int64_t gemm8w_ws_bytes(int tiles);

}  // namespace samgemm
