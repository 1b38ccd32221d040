// Row-wise / reduction kernels of the SA-M4C training path (gfx950): all HBM-bound, one pass each,
// vectorised 8-byte (4 x bf16) or 16-byte accesses, 64-lane wave reductions.
//   layernorm fwd/bwd  <- BertLayerNorm, sam/sa_m4c.py:1016-1028 (TF style: eps inside sqrt, biased variance);
//                         bwd also applies the hidden-dropout mask of the preceding dense (BertSelfOutput /
//                         BertOutput) and reduces dgamma / dbeta / dbias in the same pass
//   colsum             <- bias gradients of the nn.Linear sites
//   bce loss           <- M4CDecodingBCEWithMaskLoss, sam/task_utils.py:19-30 (forward + analytic gradient)
//   ptr scores         <- OcrPtrNet.forward bilinear + additive mask, sam/sa_m4c.py:878-897
//   adam / sumsq       <- clip_grad_norm_ + Adam step of train.py:139-142 over one flat parameter buffer
#include "common.h"
#include <stdlib.h>
#include "sam_hip.h"

namespace {

constexpr int LN_PARTIAL_BLOCKS = 512;


template <typename T> struct Ld4;
template <> struct Ld4<bf16_t> {
  static __device__ __forceinline__ void ld(const void* p, int64_t idx, float* v) {
    const uint2 x = *reinterpret_cast<const uint2*>(reinterpret_cast<const bf16_t*>(p) + idx);
    v[0] = bf_lo(x.x); v[1] = bf_hi(x.x); v[2] = bf_lo(x.y); v[3] = bf_hi(x.y);
  }
};
template <> struct Ld4<float> {
  static __device__ __forceinline__ void ld(const void* p, int64_t idx, float* v) {
    const float4 x = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(p) + idx);
    v[0] = x.x; v[1] = x.y; v[2] = x.z; v[3] = x.w;
  }
};
__device__ __forceinline__ void st4_bf16(void* p, int64_t idx, const float* v) {
  *reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(p) + idx) = make_uint2(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]));
}

// ------------------------------------------------------------------------------------------ layernorm
constexpr int LN_FWD_ROWS = 1;   // rows per wave (2 measured slower: 16.0 vs 14.7 us at 11648 x 768, HBM-cold).  16-byte (8 x bf16) accesses instead of
                                 // 8-byte ones, forward and backward, measured identical (11.3 vs 11.5 us, 25.9 vs 26.0 us): at 36-72 MB per launch these
                                 // kernels are bounded by ramp-up / tail and the dependent row reductions, not by access width
template <typename InT, int NCH>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const void* x, int64_t ldx, const float* gamma, const float* beta, float eps, int M, int D,
                                                     void* y, int64_t ldy, float* mean_out, float* rstd_out) {
  const int lane = threadIdx.x & 63, row0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * LN_FWD_ROWS;
  if (row0 >= M) return;
  const int nchunk = D >> 2;
  float v[LN_FWD_ROWS][NCH][4];
#pragma unroll
  for (int i = 0; i < LN_FWD_ROWS; ++i) {   // unconditional loads at a clamped (row, chunk): a predicated load becomes a branch + vmcnt(0)
    const int row = min(row0 + i, M - 1);
#pragma unroll
    for (int j = 0; j < NCH; ++j) Ld4<InT>::ld(x, (int64_t)row * ldx + 4 * min(lane + 64 * j, nchunk - 1), v[i][j]);
  }
  float g4[NCH][4], b4[NCH][4];
#pragma unroll
  for (int j = 0; j < NCH; ++j) {
    const int c = min(lane + 64 * j, nchunk - 1);
    Ld4<float>::ld(gamma, 4 * c, g4[j]);
    Ld4<float>::ld(beta, 4 * c, b4[j]);
  }
#pragma unroll
  for (int i = 0; i < LN_FWD_ROWS; ++i) {
    const int row = row0 + i;
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
      if (lane + 64 * j >= nchunk) v[i][j][0] = v[i][j][1] = v[i][j][2] = v[i][j][3] = 0.f;
      s += (v[i][j][0] + v[i][j][1]) + (v[i][j][2] + v[i][j][3]);
    }
    const float mean = wave_sum_v(s) / D;
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < NCH; ++j)
      if (lane + 64 * j < nchunk)
#pragma unroll
        for (int e = 0; e < 4; ++e) { const float d = v[i][j][e] - mean; q += d * d; }
    const float rstd = 1.0f / sqrtf(wave_sum_v(q) / D + eps);
    if (row >= M) continue;
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
      const int c = lane + 64 * j;
      if (c < nchunk) {
        float o[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = g4[j][e] * ((v[i][j][e] - mean) * rstd) + b4[j][e];
        st4_bf16(y, (int64_t)row * ldy + 4 * c, o);
      }
    }
    if (lane == 0) { mean_out[row] = mean; rstd_out[row] = rstd; }
  }
}

// a 4-element chunk in its stored form (what a prefetched row keeps in registers until its turn: 2 registers per bf16 chunk instead of 4)
template <typename T> struct Raw4;
template <> struct Raw4<bf16_t> {
  typedef uint2 R;
  static __device__ __forceinline__ R ld(const void* p, int64_t idx) { return *reinterpret_cast<const uint2*>(reinterpret_cast<const bf16_t*>(p) + idx); }
  static __device__ __forceinline__ void expand(const R& x, float* v) { v[0] = bf_lo(x.x); v[1] = bf_hi(x.x); v[2] = bf_lo(x.y); v[3] = bf_hi(x.y); }
};
template <> struct Raw4<float> {
  typedef float4 R;
  static __device__ __forceinline__ R ld(const void* p, int64_t idx) { return *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(p) + idx); }
  static __device__ __forceinline__ void expand(const R& x, float* v) { v[0] = x.x; v[1] = x.y; v[2] = x.z; v[3] = x.w; }
};

// empty volatile asm statements that "touch" a value: the compiler keeps them in program order, so everything computed FROM the value stays behind
// the statement and everything it was computed from in front of it (the LayerNorm backward uses them to keep one row's work in one piece)
__device__ __forceinline__ void pin(unsigned& v) { asm volatile("" : "+v"(v)); }
__device__ __forceinline__ void pin(float& v) { asm volatile("" : "+v"(v)); }
__device__ __forceinline__ void pin(uint2& v) { pin(v.x); pin(v.y); }
__device__ __forceinline__ void pin(float4& v) { pin(v.x); pin(v.y); pin(v.z); pin(v.w); }

// ws layout: [LN_PARTIAL_BLOCKS][3][D]  (dgamma, dbeta, dbias partials)
// Wave w of block b takes rows b*4 + w, + 4*grid, + 8*grid, ... ONE AT A TIME, with the next NS - 1 rows of its sequence already requested (raw bf16 in
// registers: a ring of NS slots, the loop unrolled NS times so that every slot is a fixed register set).  Round 4's first version took three rows per
// trip -- loads, reductions, stores, then the next three loads: with 2048 waves for 11648 rows every wave of the chip was in the same phase at the
// same time (a 19 MB read burst, then VALU with the memory system idle, then a store burst, twice over: 35 us for 72 MB in the step).  Here the
// loads of rows r+1, r+2 are in flight while row r is reduced and stored, and the stores of row r while r+1 is reduced.
// Row order per wave and the order of every addition are those of the first version: identical dx / dxd and identical partial sums.
// FULL: D == 256 * NCH (every chunk of every lane is inside the row: no per-chunk bounds selects).
// MODE: 0 = dx only, 1 = dx and an identical dxd (dropout off), 2 = dx and the dropout-masked dxd (compile-time: run-time tests of dxd / thr16 put six
// uniform branches between the stores of every chunk).
template <typename InT, int NCH, bool FULL, int MODE>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const bf16_t* __restrict__ dy, int64_t ldd, const void* __restrict__ x, int64_t ldx, const float* __restrict__ mean,
                                                     const float* __restrict__ rstd, const float* __restrict__ gamma, int M, int D, bf16_t* __restrict__ dx,
                                                     bf16_t* __restrict__ dxd, int64_t ldo, unsigned thr16, float inv_keep, unsigned seed_lo, unsigned seed_hi,
                                                     unsigned off_lo, unsigned off_hi, const unsigned long long* rng_state, float* __restrict__ ws) {
  if (MODE == 2) rng_resolve(rng_state, seed_lo, seed_hi, off_lo, off_hi);
  constexpr int NS = NCH <= 3 ? 3 : (NCH == 4 ? 2 : 1);      // ring slots = rows a wave has requested or is working on (VGPR budget)
  typedef typename Raw4<InT>::R XR;
  __shared__ float red[4][64 * 4];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int nchunk = D >> 2;
  float ag[NCH][4], ab[NCH][4], ad[NCH][4];
#pragma unroll
  for (int j = 0; j < NCH; ++j)
#pragma unroll
    for (int e = 0; e < 4; ++e) ag[j][e] = ab[j][e] = ad[j][e] = 0.f;
  float gm[NCH][4];
#pragma unroll
  for (int j = 0; j < NCH; ++j) {
    const int c = lane + 64 * j;
    Ld4<float>::ld(gamma, 4 * min(c, nchunk - 1), gm[j]);
    if (!FULL && c >= nchunk) gm[j][0] = gm[j][1] = gm[j][2] = gm[j][3] = 0.f;
  }
  __builtin_amdgcn_sched_barrier(0);        // gamma first: the first row needs it, and loads are waited for in issue order
  const DropHalfConsts hk = dropout_half_consts((lane & 1) != 0);     // chunk c = lane + 64 j: (c & 1) == (lane & 1)
  const int rstride = gridDim.x * 4;
  XR xr[NS][NCH];
  uint2 dr[NS][NCH];
  float mu[NS], rs[NS];
  // every load is unconditional, at a clamped (row, chunk): a predicated load compiles to a branch + s_waitcnt vmcnt(0)
#define LN_BWD_REQUEST(S, ROW)                                                                     \
  do {                                                                                             \
    const int row_ = min((ROW), M - 1);                                                            \
    mu[S] = mean[row_]; rs[S] = rstd[row_];                                                        \
    _Pragma("unroll") for (int j = 0; j < NCH; ++j) {                                              \
      const int c_ = FULL ? lane + 64 * j : min(lane + 64 * j, nchunk - 1);                        \
      xr[S][j] = Raw4<InT>::ld(x, (int64_t)row_ * ldx + 4 * c_);                                   \
      dr[S][j] = Raw4<bf16_t>::ld(dy, (int64_t)row_ * ldd + 4 * c_);                               \
    }                                                                                              \
  } while (0)
  // One slot's turn: reduce and store the row it holds, request the row NS places further on.  The main loop runs the trips in which EVERY slot of EVERY
  // wave holds a live row (a uniform count, no test inside: the compiler's s_waitcnt placement merges the pending-load state of joining paths to the
  // more conservative one, and a skipped slot re-joining the loop -- or leaving it: the loop exits are unified into flags and joins -- turned every wait
  // into a wait for nearly all outstanding loads); the ragged rest, at most NS rows per wave and already requested by the last trip, follows behind it.
  // The first trip is peeled so that the loop header joins two identical end-of-trip states.
#define LN_BWD_SLOT(S, ROW, REQ)                                                                                                     \
  {                                                                                                                                \
    const int row = (ROW);                                                                                                         \
    float xh[NCH][4], g[NCH][4];                                                                                                   \
    float t1 = 0.f, t2 = 0.f;                                                                                                      \
    asm volatile("" ::: "memory");                                                                                                 \
    _Pragma("unroll") for (int j = 0; j < NCH; ++j) { pin(xr[S][j]); pin(dr[S][j]); }     /* this row's arithmetic starts here, not earlier */ \
    _Pragma("unroll") for (int j = 0; j < NCH; ++j) {                                                                              \
      const int c = lane + 64 * j;                                                                                                 \
      float xv[4], dv[4];                                                                                                          \
      Raw4<InT>::expand(xr[S][j], xv);                                                                                             \
      Raw4<bf16_t>::expand(dr[S][j], dv);                                                                                          \
      if (!FULL && c >= nchunk) { _Pragma("unroll") for (int e = 0; e < 4; ++e) xv[e] = dv[e] = 0.f; }                             \
      _Pragma("unroll") for (int e = 0; e < 4; ++e) {                                                                              \
        xh[j][e] = (FULL || c < nchunk) ? (xv[e] - mu[S]) * rs[S] : 0.f;                                                           \
        g[j][e] = dv[e] * gm[j][e];                                                                                                \
        t1 += g[j][e];                                                                                                             \
        t2 += g[j][e] * xh[j][e];                                                                                                  \
        ag[j][e] += dv[e] * xh[j][e];                                                                                              \
        ab[j][e] += dv[e];                                                                                                         \
      }                                                                                                                            \
    }                                                                                                                              \
    const float rsv = rs[S];                                                                                                       \
    pin(t1); pin(t2);                                                                                                              \
    if (REQ) {                                    /* this slot's next row: its registers are free from here on (pinned: left alone, the */ \
      asm volatile("" ::: "memory");              /* scheduler sinks these loads behind the other slots' arithmetic)                    */ \
      __builtin_amdgcn_sched_barrier(0);                                                                                           \
      LN_BWD_REQUEST(S, row + NS * rstride);                                                                                       \
      __builtin_amdgcn_sched_barrier(0);                                                                                           \
      asm volatile("" ::: "memory");                                                                                               \
    }                                                                                                                              \
    const float s1 = wave_sum_v(t1) / D, s2 = wave_sum_v(t2) / D;                                                                      \
    const unsigned rkey = MODE == 2 ? hidden_dropout_row_key((unsigned)row, off_lo, off_hi, seed_lo, seed_hi) : 0u;                \
    _Pragma("unroll") for (int j = 0; j < NCH; ++j) {                                                                              \
      const int c = lane + 64 * j;                                                                                                 \
      if (!FULL && c >= nchunk) continue;                                                                                          \
      float o[4];                                                                                                                  \
      _Pragma("unroll") for (int e = 0; e < 4; ++e) o[e] = rsv * (g[j][e] - s1 - xh[j][e] * s2);                                   \
      st4_bf16(dx, (int64_t)row * ldo + 4 * c, o);                                                                                 \
      if (MODE >= 1) {                                                                                                             \
        if (MODE == 2) { /* the (row, col/8) stream of the GEMM epilogue that drew the forward mask; this lane's four columns = half a draw */ \
          unsigned lo, hi;                                                                                                         \
          dropout_bits_half(rkey, (unsigned)(c >> 1), hk, lo, hi);                                                                 \
          o[0] = (lo & 0xffffu) >= thr16 ? o[0] * inv_keep : 0.f;                                                                  \
          o[1] = (lo >> 16) >= thr16 ? o[1] * inv_keep : 0.f;                                                                      \
          o[2] = (hi & 0xffffu) >= thr16 ? o[2] * inv_keep : 0.f;                                                                  \
          o[3] = (hi >> 16) >= thr16 ? o[3] * inv_keep : 0.f;                                                                      \
        }                                                                                                                          \
        st4_bf16(dxd, (int64_t)row * ldo + 4 * c, o);                                                                              \
      }                                                                                                                            \
      _Pragma("unroll") for (int e = 0; e < 4; ++e) ad[j][e] += o[e];                                                              \
    }                                                                                                                              \
    __builtin_amdgcn_sched_barrier(0);                                                                                             \
  }
#define LN_BWD_TRIP(R0)                                                  \
  LN_BWD_SLOT(0, (R0), true)                                             \
  if constexpr (NS > 1) LN_BWD_SLOT(1, (R0) + rstride, true)             \
  if constexpr (NS > 2) LN_BWD_SLOT(2, (R0) + 2 * rstride, true)
  const int row00 = blockIdx.x * 4 + wave;
  // requests in slot order (sched_barrier: the scheduler would interleave them, and loads return -- and are waited for -- in issue order)
  LN_BWD_REQUEST(0, row00);
  __builtin_amdgcn_sched_barrier(0);
  if constexpr (NS > 1) { LN_BWD_REQUEST(1, row00 + rstride); __builtin_amdgcn_sched_barrier(0); }
  if constexpr (NS > 2) { LN_BWD_REQUEST(2, row00 + 2 * rstride); __builtin_amdgcn_sched_barrier(0); }
  const int full = M / (rstride * NS);
  int r0 = row00;
  if (full > 0) {
    LN_BWD_TRIP(r0)
    r0 += NS * rstride;
    for (int t = 1; t < full; ++t, r0 += NS * rstride) { LN_BWD_TRIP(r0) }
  }
  if (r0 < M) LN_BWD_SLOT(0, r0, false)
  if constexpr (NS > 1) { if (r0 + rstride < M) LN_BWD_SLOT(1, r0 + rstride, false) }
  if constexpr (NS > 2) { if (r0 + 2 * rstride < M) LN_BWD_SLOT(2, r0 + 2 * rstride, false) }
#undef LN_BWD_TRIP
#undef LN_BWD_SLOT
#undef LN_BWD_REQUEST
  // block reduction over the 4 waves, then one partial row per block
#pragma unroll
  for (int which = 0; which < 3; ++which) {
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
      const int c = lane + 64 * j;
      __syncthreads();
#pragma unroll
      for (int e = 0; e < 4; ++e) red[wave][lane * 4 + e] = which == 0 ? ag[j][e] : (which == 1 ? ab[j][e] : ad[j][e]);
      __syncthreads();
      if (wave == 0 && c < nchunk) {
        float o[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = red[0][lane * 4 + e] + red[1][lane * 4 + e] + red[2][lane * 4 + e] + red[3][lane * 4 + e];
        *reinterpret_cast<float4*>(ws + ((int64_t)blockIdx.x * 3 + which) * D + 4 * c) = make_float4(o[0], o[1], o[2], o[3]);
      }
    }
  }
}

// out[c] (+)= sum_r ws[r*stride + c]   (deterministic order).  64 columns x 16 row lanes per block (every thread keeps 8 independent,
// unconditional loads in flight: the kernel is pure latency); grid.y selects one of up to three (column offset, output) pairs so the
// LN backward finishes dgamma/dbeta/dbias in one launch.
struct FinalizeOuts { float* out[3]; int64_t col0[3]; };
constexpr int FIN_RL = 16;
__global__ __launch_bounds__(64 * FIN_RL) void partial_finalize_kernel(const float* ws, int nrows, int64_t stride, int ncols, FinalizeOuts o, int accumulate) {
  __shared__ float red[FIN_RL][64];
  const int cx = threadIdx.x & 63, ry = threadIdx.x >> 6, c = min((int)blockIdx.x * 64 + cx, ncols - 1);
  const float* base = ws + o.col0[blockIdx.y] + c;
  float s = 0.f;
  for (int r = ry; r < nrows; r += FIN_RL * 8) {
    float t[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) t[u] = base[(int64_t)min(r + FIN_RL * u, nrows - 1) * stride];
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (r + FIN_RL * u >= nrows) t[u] = 0.f;
    s += ((t[0] + t[1]) + (t[2] + t[3])) + ((t[4] + t[5]) + (t[6] + t[7]));
  }
  red[ry][cx] = s;
  __syncthreads();
  if (ry == 0 && (int)blockIdx.x * 64 + cx < ncols) {
    float tot = 0.f;
#pragma unroll
    for (int u = 0; u < FIN_RL; ++u) tot += red[u][cx];
    float* out = o.out[blockIdx.y];
    out[c] = accumulate ? out[c] + tot : tot;
  }
}

// the same reduction for up to 32 LayerNorm backward calls in ONE launch (blockIdx.z = call): a training step runs 26 LayerNorm backwards, each of
// which used to be followed by its own 6 us finalize launch; with SAM_LN_DEFER_FINALIZE they leave their partial rows in place and one launch at the
// end of the backward pass finishes all of them (the parameter gradients are not needed before the optimizer)
struct FinalizeBatch { const float* ws[32]; float* out[32][3]; int nrows[32]; int accumulate[32]; int D; };
__global__ __launch_bounds__(64 * FIN_RL) void partial_finalize_batch_kernel(FinalizeBatch b) {
  __shared__ float red[FIN_RL][64];
  const int z = blockIdx.z, D = b.D;
  float* out = b.out[z][blockIdx.y];
  const int cx = threadIdx.x & 63, ry = threadIdx.x >> 6, c = min((int)blockIdx.x * 64 + cx, D - 1);
  const int nrows = b.nrows[z];
  const int64_t stride = 3 * (int64_t)D;
  const float* base = b.ws[z] + (int64_t)blockIdx.y * D + c;
  float s = 0.f;
  if (out) {
    for (int r = ry; r < nrows; r += FIN_RL * 8) {
      float t[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) t[u] = base[(int64_t)min(r + FIN_RL * u, nrows - 1) * stride];
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (r + FIN_RL * u >= nrows) t[u] = 0.f;
      s += ((t[0] + t[1]) + (t[2] + t[3])) + ((t[4] + t[5]) + (t[6] + t[7]));
    }
  }
  red[ry][cx] = s;
  __syncthreads();
  if (out && ry == 0 && (int)blockIdx.x * 64 + cx < D) {
    float tot = 0.f;
#pragma unroll
    for (int u = 0; u < FIN_RL; ++u) tot += red[u][cx];
    out[c] = b.accumulate[z] ? out[c] + tot : tot;
  }
}

// ------------------------------------------------------------------------------------------ colsum
constexpr int COLSUM_CHUNKS = 64;
__global__ __launch_bounds__(256) void colsum_partial_kernel(const bf16_t* x, int64_t ldx, int M, int N, float* ws) {
  const int c4 = blockIdx.x * 256 + threadIdx.x;  // group of 4 columns
  if (c4 * 4 >= N) return;
  const int rows_per = (M + COLSUM_CHUNKS - 1) / COLSUM_CHUNKS;
  const int r0 = blockIdx.y * rows_per, r1 = min(M, r0 + rows_per);
  float a[4] = {0.f, 0.f, 0.f, 0.f};
  int r = r0;
  for (; r + 8 <= r1; r += 8) {
    float v[8][4];
#pragma unroll
    for (int u = 0; u < 8; ++u) Ld4<bf16_t>::ld(x, (int64_t)(r + u) * ldx + 4 * c4, v[u]);
#pragma unroll
    for (int u = 0; u < 8; ++u) { a[0] += v[u][0]; a[1] += v[u][1]; a[2] += v[u][2]; a[3] += v[u][3]; }
  }
  for (; r < r1; ++r) {
    float v[4];
    Ld4<bf16_t>::ld(x, (int64_t)r * ldx + 4 * c4, v);
    a[0] += v[0]; a[1] += v[1]; a[2] += v[2]; a[3] += v[3];
  }
  *reinterpret_cast<float4*>(ws + (int64_t)blockIdx.y * N + 4 * c4) = make_float4(a[0], a[1], a[2], a[3]);
}

// ------------------------------------------------------------------------------------------ BCE loss
// loss = sum_{r,c} bce(x, t) * mask[r] / max(sum(mask), 1);  d x = (sigmoid(x) - t) * mask[r] * gscale / count
// One block row of the grid per decoding row r (no index division), two adjacent columns per thread (8-byte accesses: V and the row strides are
// even, so a pair never straddles the classifier / pointer boundary); exp and log through the hardware units.  (The first version went element by
// element with a 64-bit division and libm's log1pf per element: 39 us for 3.9 M scores, now ~12.)
__device__ __forceinline__ float softplus_neg_abs(float e) {      // log(1 + e), e = exp(-|x|) in (0, 1]
  return e < 1e-3f ? e * (1.0f - e * (0.5f - e * 0.33333334f)) : __logf(1.0f + e);
}
template <int VEC>      // 2: adjacent column pairs (everything even and 8-byte aligned); 1: any shape
__global__ __launch_bounds__(256) void bce_kernel(const float* fixed, int64_t ldf, const float* ocr, int64_t ldoc, const float* targets, int64_t ldt,
                                                  const float* mask, int R, int V, int No, float gscale, const float* global_count, float* loss,
                                                  bf16_t* d_fixed, int64_t lddf, float* d_ocr, int64_t lddo) {
  __shared__ float sred[4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float cnt = 0.f;
  for (int r = threadIdx.x; r < R; r += 256) cnt += mask[r];
  cnt = wave_sum(cnt);
  if (lane == 0) sred[wave] = cnt;
  __syncthreads();
  // data parallel: the normaliser is the number of unmasked decoding steps of the GLOBAL batch (all-reduced by the caller, raw: the
  // clamp is applied to the global value as the reference does, task_utils.py:28-29); every rank then contributes sum_local / max(C, 1)
  cnt = fmaxf(global_count ? global_count[0] : sred[0] + sred[1] + sred[2] + sred[3], 1.0f);
  __syncthreads();
  const float inv_cnt = 1.0f / cnt;
  const int W = V + No, r = blockIdx.x;
  const float m = mask[r];
  const float gs = m * inv_cnt * gscale;
  if (m == 0.f) {                    // a masked decoding step (about half of them): zero gradient, nothing for the loss
    for (int c = VEC * (blockIdx.y * 256 + threadIdx.x); c < W; c += VEC * 256 * gridDim.y) {
      if (VEC == 2) {
        if (c < V) *reinterpret_cast<unsigned*>(d_fixed + (int64_t)r * lddf + c) = 0u;
        else *reinterpret_cast<float2*>(d_ocr + (int64_t)r * lddo + (c - V)) = make_float2(0.f, 0.f);
      } else {
        if (c < V) d_fixed[(int64_t)r * lddf + c] = (bf16_t)0;
        else d_ocr[(int64_t)r * lddo + (c - V)] = 0.f;
      }
    }
    return;
  }
  float acc = 0.f;
  for (int c = VEC * (blockIdx.y * 256 + threadIdx.x); c < W; c += VEC * 256 * gridDim.y) {
    const bool in_fixed = c < V;
    float xs[2] = {0.f, 0.f}, ts[2] = {0.f, 0.f};
    if (VEC == 2) {
      const float2 x2 = in_fixed ? *reinterpret_cast<const float2*>(fixed + (int64_t)r * ldf + c) : *reinterpret_cast<const float2*>(ocr + (int64_t)r * ldoc + (c - V));
      const float2 t2 = *reinterpret_cast<const float2*>(targets + (int64_t)r * ldt + c);
      xs[0] = x2.x; xs[1] = x2.y; ts[0] = t2.x; ts[1] = t2.y;
    } else {
      xs[0] = in_fixed ? fixed[(int64_t)r * ldf + c] : ocr[(int64_t)r * ldoc + (c - V)];
      ts[0] = targets[(int64_t)r * ldt + c];
    }
    float gx[2];
#pragma unroll
    for (int e_ = 0; e_ < VEC; ++e_) {
      const float x = xs[e_], t = ts[e_];
      const float e = __expf(-fabsf(x));
      acc += fmaxf(x, 0.f) - x * t + softplus_neg_abs(e);
      const float inv = __builtin_amdgcn_rcpf(1.0f + e);
      const float sig = x >= 0.f ? inv : e * inv;
      gx[e_] = (sig - t) * gs;
    }
    if (VEC == 2) {
      if (in_fixed) *reinterpret_cast<unsigned*>(d_fixed + (int64_t)r * lddf + c) = pack_bf16x2(gx[0], gx[1]);
      else *reinterpret_cast<float2*>(d_ocr + (int64_t)r * lddo + (c - V)) = make_float2(gx[0], gx[1]);
    } else {
      if (in_fixed) d_fixed[(int64_t)r * lddf + c] = f2bf(gx[0]);
      else d_ocr[(int64_t)r * lddo + (c - V)] = gx[0];
    }
  }
  acc = wave_sum(acc * m);
  if (lane == 0) sred[wave] = acc;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(loss, (sred[0] + sred[1] + sred[2] + sred[3]) * inv_cnt);
}

// ------------------------------------------------------------------------------------------ pointer network
// scores[b,s,o] = scale * <q[b,s,:], k[b,o,:]> + (1 - mask[b,o]) * -10000     (fp32, literal -10000 kept: it is an OUTPUT)
__global__ __launch_bounds__(256) void ptr_fwd_kernel(const bf16_t* q, const bf16_t* k, const uint8_t* mask, int S, int No, int D, float scale,
                                                      float* out, int64_t ldo_b, int64_t ldo_s) {
  const int b = blockIdx.x, s = blockIdx.y, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const bf16_t* qrow = q + ((int64_t)b * S + s) * D;
  const bf16_t* kb = k + (int64_t)b * No * D;
  for (int o = wave; o < No; o += 4) {  // one wave per (s, o) pair: 64-lane dot product over D
    float acc = 0.f;
    for (int c = lane; c * 4 < D; c += 64) {
      float a[4], bb[4];
      Ld4<bf16_t>::ld(qrow, 4 * c, a);
      Ld4<bf16_t>::ld(kb, (int64_t)o * D + 4 * c, bb);
      acc += a[0] * bb[0] + a[1] * bb[1] + a[2] * bb[2] + a[3] * bb[3];
    }
    acc = wave_sum(acc);
    if (lane == 0) out[(int64_t)b * ldo_b + (int64_t)s * ldo_s + o] = acc * scale + (mask[(int64_t)b * No + o] ? 0.f : -10000.0f);
  }
}
// The same scores on the matrix cores: one block (4 waves) per sample; S <= 16 decoding rows x No <= 64 OCR columns = four 16 x 16 tiles; every wave
// takes a quarter of D (6 k-steps of 32 at D = 768) with all of its fragment loads -- straight from global memory, rows are k-contiguous -- in flight
// at once, and the four partial tiles are added in a fixed order through 16 KB of LDS.  (The dot-product kernel above: 19 us for 38 k scores.)
template <int KS>        // k-steps of 32 per wave
__global__ __launch_bounds__(256) void ptr_fwd_mfma_kernel(const bf16_t* q, const bf16_t* k, const uint8_t* mask, int S, int No, int D, float scale, float* out,
                                                           int64_t ldo_b, int64_t ldo_s) {
  __shared__ __attribute__((aligned(16))) float part[4][4][64][4];       // [wave][tile][lane][r]
  const int b = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6, i = lane & 15, g = lane >> 4;
  const bf16_t* qrow = q + ((int64_t)b * S + min(i, S - 1)) * D + wave * (KS * 32) + 8 * g;
  bf16x8 qf[KS], kf[4][KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) qf[ks] = *reinterpret_cast<const bf16x8*>(qrow + 32 * ks);
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const bf16_t* krow = k + ((int64_t)b * No + min(16 * t + i, No - 1)) * D + wave * (KS * 32) + 8 * g;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) kf[t][ks] = *reinterpret_cast<const bf16x8*>(krow + 32 * ks);
  }
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[t][ks], qf[ks], acc, 0, 0, 0);      // D[o = 4g + r][s = i]
    *reinterpret_cast<f32x4*>(&part[wave][t][lane][0]) = acc;
  }
  __syncthreads();
  const int t = wave;                                    // wave t finishes tile t
  const f32x4 a0 = *reinterpret_cast<const f32x4*>(&part[0][t][lane][0]), a1 = *reinterpret_cast<const f32x4*>(&part[1][t][lane][0]),
              a2 = *reinterpret_cast<const f32x4*>(&part[2][t][lane][0]), a3 = *reinterpret_cast<const f32x4*>(&part[3][t][lane][0]);
  if (i < S) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int o = 16 * t + 4 * g + r;
      if (o < No) out[(int64_t)b * ldo_b + (int64_t)i * ldo_s + o] = ((a0[r] + a1[r]) + (a2[r] + a3[r])) * scale + (mask[(int64_t)b * No + o] ? 0.f : -10000.0f);
    }
  }
}

// dq[b,s,:] = scale * sum_o ds[b,s,o] k[b,o,:] ; dk[b,o,:] = scale * sum_s ds[b,s,o] q[b,s,:]   (one block per output row)
__global__ __launch_bounds__(256) void ptr_bwd_kernel(const float* ds, int64_t ld_b, int64_t ld_s, const bf16_t* q, const bf16_t* k, int S, int No, int D,
                                                      float scale, bf16_t* dq, bf16_t* dk) {
  __shared__ float w[256];
  const int b = blockIdx.x, row = blockIdx.y;
  const bool is_q = row < S;
  const int nsum = is_q ? No : S;
  const int o_fixed = row - S;
  for (int t = threadIdx.x; t < nsum; t += 256)
    w[t] = (is_q ? ds[(int64_t)b * ld_b + (int64_t)row * ld_s + t] : ds[(int64_t)b * ld_b + (int64_t)t * ld_s + o_fixed]) * scale;
  __syncthreads();
  const bf16_t* src = is_q ? k + (int64_t)b * No * D : q + (int64_t)b * S * D;
  bf16_t* dst = is_q ? dq + ((int64_t)b * S + row) * D : dk + ((int64_t)b * No + o_fixed) * D;
  for (int c = threadIdx.x; c * 4 < D; c += 256) {
    float a[4] = {0.f, 0.f, 0.f, 0.f};
    for (int t = 0; t < nsum; ++t) {
      float v[4];
      Ld4<bf16_t>::ld(src, (int64_t)t * D + 4 * c, v);
      a[0] += w[t] * v[0]; a[1] += w[t] * v[1]; a[2] += w[t] * v[2]; a[3] += w[t] * v[3];
    }
    st4_bf16(dst, 4 * c, a);
  }
}

// ------------------------------------------------------------------------------------------ optimizer
constexpr int SUMSQ_BLOCKS = 1024, SPARSE_ROW_BLOCKS = 256;
// A row-sparse region of the flat buffers (the word-embedding table: 30522 x 768 of the 96.6 M parameters, of which a step touches at most B * 20
// rows): rows [lo, hi) / row_len whose `touched` flag is 0 have NEVER received a gradient -- g = m = v = 0 there, Adam's update is exactly zero and
// their squares add nothing to the norm, so they are skipped (bit-identical to the dense pass).  The dense kernels walk the rest of the buffer
// through an index remap; the last SPARSE_ROW_BLOCKS blocks of the same launch walk the flagged rows.
struct SparseRows { int64_t lo4, hi4; int row_len4, rows; const unsigned char* touched; };      // (float4 units)
__global__ __launch_bounds__(256) void sumsq_partial_kernel(const float* g, int64_t n, SparseRows sp, int dense_blocks, float* partial) {
  __shared__ float sred[4];
  float acc = 0.f;
  const int64_t n4 = n >> 2;
  if ((int)blockIdx.x < dense_blocks) {
    const int64_t gap = sp.hi4 - sp.lo4, n4d = n4 - gap;
    for (int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x; j < n4d; j += (int64_t)dense_blocks * 256) {
      const int64_t i = j < sp.lo4 ? j : j + gap;
      const float4 v = reinterpret_cast<const float4*>(g)[i];
      acc += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
    }
  } else {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int row = ((int)blockIdx.x - dense_blocks) * 4 + wave; row < sp.rows; row += SPARSE_ROW_BLOCKS * 4) {
      if (!sp.touched[row]) continue;
      for (int c = lane; c < sp.row_len4; c += 64) {
        const float4 v = reinterpret_cast<const float4*>(g)[sp.lo4 + (int64_t)row * sp.row_len4 + c];
        acc += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
      }
    }
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) { const float t = g[(n4 << 2) + threadIdx.x]; acc += t * t; }
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) sred[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = (sred[0] + sred[1]) + (sred[2] + sred[3]);
}
__global__ __launch_bounds__(256) void sumsq_final_kernel(const float* partial, int nblk, float* out) {
  __shared__ float sred[4];
  float acc = 0.f;
  for (int i = threadIdx.x; i < nblk; i += 256) acc += partial[i];
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) sred[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) out[0] = (sred[0] + sred[1]) + (sred[2] + sred[3]);
}

struct AdamSegs { int64_t end[8]; float lr[8]; int n; };
// dev_sched (may be NULL): [lr of segment 0..n-1, 1 - beta1^t, 1 - beta2^t] in DEVICE memory -- a captured (hipGraph) launch reads the
// schedule of the current step from there instead of from its frozen by-value arguments
__global__ __launch_bounds__(256) void adam_kernel(float* p, float* g, float* m, float* v, bf16_t* pb, int64_t n, AdamSegs segs, float b1, float b2,
                                                   float eps, float bc1, float rsqrt_bc2, const float* gnorm_sq, float max_norm, const float* dev_sched, SparseRows sp,
                                                   int dense_blocks, int64_t jlo, int64_t jhi, int zero_g, const int* gate) {
  // [jlo, jhi): this launch's share of the dense index space (all float4 outside the row-sparse region); gate: a device word -- 0 = do nothing
  // (sam_adam_step_range: the pending update at the head of a captured step, which a replay must skip when the host has already applied it)
  if (gate && *gate == 0) return;
  if (dev_sched) {
    for (int s = 0; s < segs.n; ++s) segs.lr[s] = dev_sched[s];
    bc1 = dev_sched[segs.n];
    rsqrt_bc2 = 1.0f / sqrtf(dev_sched[segs.n + 1]);
  }
  float clip = 1.0f;
  if (gnorm_sq && max_norm > 0.f) clip = fminf(1.0f, max_norm / (sqrtf(gnorm_sq[0]) + 1e-6f));   // torch clip_grad_norm_
  const int64_t gap = sp.hi4 - sp.lo4;
  const bool row_block = (int)blockIdx.x >= dense_blocks;
  // dense blocks: every float4 outside the row-sparse region; row blocks: the touched rows of that region, one wave per row -- whose gradient is
  // cleared on the way out (the region is not zero-filled per step: untouched rows stay zero for ever)
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  int row = row_block ? ((int)blockIdx.x - dense_blocks) * 4 + wave : 0, c = lane;
  if (row_block) { while (row < sp.rows && !sp.touched[row]) row += SPARSE_ROW_BLOCKS * 4; }
  for (int64_t j = jlo + (int64_t)blockIdx.x * 256 + threadIdx.x;; j += (int64_t)dense_blocks * 256) {
    int64_t i;
    if (!row_block) {
      if (j >= jhi) break;
      i = j < sp.lo4 ? j : j + gap;
    } else {
      if (row >= sp.rows) break;
      i = sp.lo4 + (int64_t)row * sp.row_len4 + c;
    }
    const int64_t e0 = i << 2;
    float lr = 0.f;
#pragma unroll
    for (int s = 7; s >= 0; --s)
      if (s < segs.n && e0 < segs.end[s]) lr = segs.lr[s];
    // streaming access: every byte is touched once per step, nontemporal loads/stores keep 2.9 GB from churning the L2 / Infinity Cache
    // (538 -> 497 us, 5.4 -> 5.85 TB/s, A/B on one box)
    typedef float v4f __attribute__((ext_vector_type(4)));
    const v4f g_ = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(g) + i), p_ = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(p) + i),
              m_ = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(m) + i), v_ = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(v) + i);
    const float4 g4 = make_float4(g_[0], g_[1], g_[2], g_[3]);
    float4 p4 = make_float4(p_[0], p_[1], p_[2], p_[3]), m4 = make_float4(m_[0], m_[1], m_[2], m_[3]), v4 = make_float4(v_[0], v_[1], v_[2], v_[3]);
    const float gg[4] = {g4.x * clip, g4.y * clip, g4.z * clip, g4.w * clip};
    float pp[4] = {p4.x, p4.y, p4.z, p4.w}, mm[4] = {m4.x, m4.y, m4.z, m4.w}, vv[4] = {v4.x, v4.y, v4.z, v4.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      mm[e] = b1 * mm[e] + (1.0f - b1) * gg[e];
      vv[e] = b2 * vv[e] + (1.0f - b2) * gg[e] * gg[e];
      const float denom = sqrtf(vv[e]) * rsqrt_bc2 + eps;
      pp[e] -= (lr / bc1) * (mm[e] / denom);
    }
    __builtin_nontemporal_store((v4f){pp[0], pp[1], pp[2], pp[3]}, reinterpret_cast<v4f*>(p) + i);
    __builtin_nontemporal_store((v4f){mm[0], mm[1], mm[2], mm[3]}, reinterpret_cast<v4f*>(m) + i);
    __builtin_nontemporal_store((v4f){vv[0], vv[1], vv[2], vv[3]}, reinterpret_cast<v4f*>(v) + i);
    if (pb) reinterpret_cast<uint2*>(pb)[i] = make_uint2(pack_bf16x2(pp[0], pp[1]), pack_bf16x2(pp[2], pp[3]));
    if (row_block || zero_g) reinterpret_cast<float4*>(g)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (row_block) {
      c += 64;
      if (c >= sp.row_len4) {
        c = lane;
        row += SPARSE_ROW_BLOCKS * 4;
        while (row < sp.rows && !sp.touched[row]) row += SPARSE_ROW_BLOCKS * 4;
      }
    }
  }
}
__global__ __launch_bounds__(256) void cast_bf16_kernel(const float* x, bf16_t* y, int64_t n4) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    const float4 v = reinterpret_cast<const float4*>(x)[i];
    reinterpret_cast<uint2*>(y)[i] = make_uint2(pack_bf16x2(v.x, v.y), pack_bf16x2(v.z, v.w));
  }
}

// embedding backward: grad_table[idx[t], :] += dy[t, :]   (fp32 hardware atomics: rows may repeat)
__global__ __launch_bounds__(256) void embedding_bwd_kernel(const bf16_t* dy, int64_t ldd, const int64_t* idx, int T, int D, int rows, int64_t padding_idx, float* grad, int64_t ldg,
                                                            unsigned char* touched) {
  const int t = blockIdx.x;
  const int64_t row = idx[t];
  if (row < 0 || row >= rows || row == padding_idx) return;   // nn.Embedding(padding_idx=...) never accumulates into that row
  if (touched && threadIdx.x == 0) touched[row] = 1;
  for (int c = threadIdx.x; c * 4 < D; c += 256) {
    float v[4];
    Ld4<bf16_t>::ld(dy, (int64_t)t * ldd + 4 * c, v);
    float* g = grad + row * ldg + 4 * c;
#pragma unroll
    for (int e = 0; e < 4; ++e) unsafeAtomicAdd(g + e, v[e]);
  }
}

// the default: no atomics, no sort.  The block of the FIRST occurrence of a table row is the row's only writer: it adds its own dy row and every later
// duplicate's, in list order (fixed summation order: run-to-run and -- the data-parallel scatter hands every rank the same gathered list -- rank-to-rank
// bit-identical).  Finding the duplicates is a scan of the index list: 256 indices per step and thread group, T / 256 steps (T = 1280 tokens per rank).
// Why not the atomics above: device-scope fp32 atomics do not execute in the (per-XCD, mutually incoherent) L2s but at the memory side of the fabric --
// 983 k of them for 1280 rows took 89 us inside the replayed step (200 us queued behind other work), all of it on the tail's critical chain.
__global__ __launch_bounds__(256) void embedding_bwd_dedup_kernel(const bf16_t* dy, int64_t ldd, const int64_t* idx, int T, int D, int rows, int64_t padding_idx, float* grad, int64_t ldg,
                                                                  unsigned char* touched) {
  const int t = blockIdx.x, tid = threadIdx.x;
  const int64_t row = idx[t];
  if (row < 0 || row >= rows || row == padding_idx) return;
  int dup = 0;
  for (int s = tid; s < t; s += 256) dup |= idx[s] == row;
  if (__syncthreads_or(dup)) return;                      // an earlier block owns this row
  if (touched && tid == 0) touched[row] = 1;
  const int nch = (D / 4 + 255) / 256;                    // float4 chunks per thread (D = 768: one, on 192 of the 256 threads)
  for (int cc = 0; cc < nch; ++cc) {
    const int c = tid + 256 * cc;
    const bool live = c * 4 < D;
    float a[4] = {0.f, 0.f, 0.f, 0.f};
    if (live) Ld4<bf16_t>::ld(dy, (int64_t)t * ldd + 4 * c, a);
    for (int base = t + 1; base < T; base += 256) {
      const int s = base + tid;
      const int hit = s < T && idx[s] == row;
      if (__syncthreads_or(hit)) {                        // rare: a duplicate among these 256 positions -- walk them in order (block-uniform loop)
        const int end = min(base + 256, T);
        for (int k = base; k < end; ++k)
          if (idx[k] == row && live) {
            float v[4];
            Ld4<bf16_t>::ld(dy, (int64_t)k * ldd + 4 * c, v);
#pragma unroll
            for (int e = 0; e < 4; ++e) a[e] += v[e];
          }
      }
    }
    if (live) {
      float4* g = reinterpret_cast<float4*>(grad + row * ldg + 4 * c);
      const float4 o = *g;
      *g = make_float4(o.x + a[0], o.y + a[1], o.z + a[2], o.w + a[3]);
    }
  }
}

// deterministic variant: idx sorted ascending; the block of the FIRST occurrence of a row walks all its duplicates in order and is the
// only writer of that table row (data parallel: every rank scatters the same gathered list and must end with bit-identical gradients)
__global__ __launch_bounds__(256) void embedding_bwd_sorted_kernel(const bf16_t* dy, int64_t ldd, const int64_t* idx, int T, int D, int rows, int64_t padding_idx, float* grad, int64_t ldg,
                                                                   unsigned char* touched) {
  const int t = blockIdx.x;
  const int64_t row = idx[t];
  if (row < 0 || row >= rows || row == padding_idx || (t > 0 && idx[t - 1] == row)) return;
  if (touched && threadIdx.x == 0) touched[row] = 1;
  int end = t + 1;
  while (end < T && idx[end] == row) ++end;
  for (int c = threadIdx.x; c * 4 < D; c += 256) {
    float a[4] = {0.f, 0.f, 0.f, 0.f};
    for (int s = t; s < end; ++s) {
      float v[4];
      Ld4<bf16_t>::ld(dy, (int64_t)s * ldd + 4 * c, v);
#pragma unroll
      for (int e = 0; e < 4; ++e) a[e] += v[e];
    }
    float4* g = reinterpret_cast<float4*>(grad + row * ldg + 4 * c);
    const float4 o = *g;
    *g = make_float4(o.x + a[0], o.y + a[1], o.z + a[2], o.w + a[3]);
  }
}

template <typename InT>
int ln_fwd_dispatch(int nch, dim3 grid, hipStream_t st, const void* x, int64_t ldx, const float* gamma, const float* beta, float eps, int M, int D,
                    void* y, int64_t ldy, float* mean, float* rstd) {
#define LN_FWD_CASE(NC) case NC: ln_fwd_kernel<InT, NC><<<grid, dim3(256), 0, st>>>(x, ldx, gamma, beta, eps, M, D, y, ldy, mean, rstd); break;
  switch (nch) { LN_FWD_CASE(1) LN_FWD_CASE(2) LN_FWD_CASE(3) LN_FWD_CASE(4) LN_FWD_CASE(5) LN_FWD_CASE(6) LN_FWD_CASE(7) LN_FWD_CASE(8) default: return SAM_ERR_UNSUPPORTED; }
#undef LN_FWD_CASE
  return SAM_OK;
}
template <typename InT>
int ln_bwd_dispatch(int nch, dim3 grid, hipStream_t st, const bf16_t* dy, int64_t ldd, const void* x, int64_t ldx, const float* mean, const float* rstd,
                    const float* gamma, int M, int D, bf16_t* dx, bf16_t* dxd, int64_t ldo, unsigned thr16, float inv_keep, uint64_t seed, uint64_t offset, float* ws) {
#define LN_BWD_LAUNCH(NC, FULL_, MODE_) ln_bwd_kernel<InT, NC, FULL_, MODE_><<<grid, dim3(256), 0, st>>>(dy, ldd, x, ldx, mean, rstd, gamma, M, D, dx, dxd, ldo, thr16, inv_keep, \
      (unsigned)seed, (unsigned)(seed >> 32), (unsigned)offset, (unsigned)(offset >> 32), sam_get_rng_state(), ws)
#define LN_BWD_CASE(NC) case NC: \
    if (D == 256 * NC) { if (mode == 2) LN_BWD_LAUNCH(NC, true, 2); else if (mode == 1) LN_BWD_LAUNCH(NC, true, 1); else LN_BWD_LAUNCH(NC, true, 0); } \
    else { if (mode == 2) LN_BWD_LAUNCH(NC, false, 2); else if (mode == 1) LN_BWD_LAUNCH(NC, false, 1); else LN_BWD_LAUNCH(NC, false, 0); } \
    break;
  const int mode = dxd ? (thr16 ? 2 : 1) : 0;
  switch (nch) { LN_BWD_CASE(1) LN_BWD_CASE(2) LN_BWD_CASE(3) LN_BWD_CASE(4) LN_BWD_CASE(5) LN_BWD_CASE(6) LN_BWD_CASE(7) LN_BWD_CASE(8) default: return SAM_ERR_UNSUPPORTED; }
#undef LN_BWD_LAUNCH
#undef LN_BWD_CASE
  return SAM_OK;
}

}  // namespace

extern "C" int sam_layernorm_fwd(const void* x, int x_is_f32, int64_t ldx, const float* gamma, const float* beta, float eps, int M, int D, void* y,
                                 int64_t ldy, float* mean, float* rstd, void* stream) {
  SAM_REQUIRE(x && gamma && beta && y && mean && rstd, "sam_layernorm_fwd: null pointer");
  SAM_REQUIRE(M > 0 && D > 0 && D % 4 == 0 && D <= 2048 && ldx % 4 == 0 && ldy % 4 == 0, "sam_layernorm_fwd: need D %% 4 == 0, D <= 2048 (M=%d D=%d)", M, D);
  const int nch = (D / 4 + 63) / 64;
  const dim3 grid((M + 4 * LN_FWD_ROWS - 1) / (4 * LN_FWD_ROWS));
  int rc = x_is_f32 ? ln_fwd_dispatch<float>(nch, grid, (hipStream_t)stream, x, ldx, gamma, beta, eps, M, D, y, ldy, mean, rstd)
                    : ln_fwd_dispatch<bf16_t>(nch, grid, (hipStream_t)stream, x, ldx, gamma, beta, eps, M, D, y, ldy, mean, rstd);
  if (rc) return rc;
  SAM_LAUNCH_CHECK();
  return SAM_OK;
}

extern "C" int64_t sam_layernorm_bwd_ws_bytes(int D) { return (int64_t)LN_PARTIAL_BLOCKS * 3 * D * sizeof(float); }

extern "C" int sam_layernorm_bwd(const void* dy, int64_t ldd, const void* x, int x_is_f32, int64_t ldx, const float* mean, const float* rstd,
                                 const float* gamma, int M, int D, void* dx, void* dx_dropped, int64_t ldo, float p_drop, uint64_t seed, uint64_t offset,
                                 float* dgamma, float* dbeta, float* dbias, int accumulate, float* ws, void* stream) {
  SAM_REQUIRE(dy && x && mean && rstd && gamma && dx && dgamma && dbeta && ws, "sam_layernorm_bwd: null pointer");
  SAM_REQUIRE(M > 0 && D > 0 && D % 4 == 0 && D <= 2048 && ldx % 4 == 0 && ldd % 4 == 0 && ldo % 4 == 0, "sam_layernorm_bwd: bad shape M=%d D=%d", M, D);
  SAM_REQUIRE(p_drop >= 0.f && p_drop < 1.f, "sam_layernorm_bwd: p_drop out of range");
  SAM_REQUIRE(!dbias || dx_dropped || p_drop == 0.f, "sam_layernorm_bwd: dbias with dropout needs dx_dropped");
  const unsigned thr16 = dropout_thr16(p_drop);
  const float inv_keep = thr16 ? 1.0f / (1.0f - (float)thr16 / 65536.0f) : 1.0f;
  const int nch = (D / 4 + 63) / 64;
  const int nblk = min(LN_PARTIAL_BLOCKS, (M + 3) / 4);
  hipStream_t st = (hipStream_t)stream;
  bf16_t* dxd = (bf16_t*)dx_dropped;
  int rc = x_is_f32 ? ln_bwd_dispatch<float>(nch, dim3(nblk), st, (const bf16_t*)dy, ldd, x, ldx, mean, rstd, gamma, M, D, (bf16_t*)dx, dxd, ldo, thr16, inv_keep, seed, offset, ws)
                    : ln_bwd_dispatch<bf16_t>(nch, dim3(nblk), st, (const bf16_t*)dy, ldd, x, ldx, mean, rstd, gamma, M, D, (bf16_t*)dx, dxd, ldo, thr16, inv_keep, seed, offset, ws);
  if (rc) return rc;
  SAM_LAUNCH_CHECK();
  if (accumulate & 4) return SAM_OK;      // deferred: the partial rows stay in ws for sam_layernorm_bwd_finalize_batch
  // dbias of the dense in front of this LN = column sums of the (dropout-masked) dx
  FinalizeOuts fo = {{dgamma, dbeta, dbias}, {0, D, 2 * (int64_t)D}};
  partial_finalize_kernel<<<dim3((D + 63) / 64, dbias ? 3 : 2), dim3(64 * FIN_RL), 0, st>>>(ws, nblk, 3 * (int64_t)D, D, fo, accumulate & 1);
  SAM_LAUNCH_CHECK();
  return SAM_OK;
}

extern "C" int sam_layernorm_bwd_partial_rows(int M) { return min(LN_PARTIAL_BLOCKS, (M + 3) / 4); }

extern "C" int sam_layernorm_bwd_finalize_batch(const sam_ln_finalize_item* items, int count, int D, void* stream) {
  SAM_REQUIRE(items && count >= 1 && D > 0 && D % 4 == 0, "sam_layernorm_bwd_finalize_batch: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  for (int i0 = 0; i0 < count; i0 += 32) {
    FinalizeBatch b = {};
    const int n = min(32, count - i0);
    b.D = D;
    for (int i = 0; i < n; ++i) {
      const sam_ln_finalize_item& it = items[i0 + i];
      SAM_REQUIRE(it.ws && it.dgamma && it.dbeta && it.rows >= 1 && it.rows <= LN_PARTIAL_BLOCKS, "sam_layernorm_bwd_finalize_batch: item %d is malformed", i0 + i);
      b.ws[i] = it.ws; b.out[i][0] = it.dgamma; b.out[i][1] = it.dbeta; b.out[i][2] = it.dbias; b.nrows[i] = it.rows; b.accumulate[i] = it.accumulate & 1;
    }
    partial_finalize_batch_kernel<<<dim3((D + 63) / 64, 3, n), dim3(64 * FIN_RL), 0, st>>>(b);
    SAM_LAUNCH_CHECK();
  }
  return SAM_OK;
}

extern "C" int64_t sam_colsum_ws_bytes(int N) { return (int64_t)COLSUM_CHUNKS * N * sizeof(float); }

extern "C" int sam_colsum_bf16(const void* x, int64_t ldx, int M, int N, float* out, int accumulate, float* ws, void* stream) {
  SAM_REQUIRE(x && out && ws, "sam_colsum_bf16: null pointer");
  SAM_REQUIRE(M > 0 && N > 0 && N % 4 == 0 && ldx % 4 == 0, "sam_colsum_bf16: need N %% 4 == 0 (M=%d N=%d)", M, N);
  hipStream_t st = (hipStream_t)stream;
  colsum_partial_kernel<<<dim3((N / 4 + 255) / 256, COLSUM_CHUNKS), dim3(256), 0, st>>>((const bf16_t*)x, ldx, M, N, ws);
  FinalizeOuts fo = {{out, nullptr, nullptr}, {0, 0, 0}};
  partial_finalize_kernel<<<dim3((N + 63) / 64, 1), dim3(64 * FIN_RL), 0, st>>>(ws, COLSUM_CHUNKS, N, N, fo, accumulate);
  SAM_LAUNCH_CHECK();
  return SAM_OK;
}

extern "C" int sam_bce_loss(const float* fixed_scores, int64_t ld_fixed, const float* ocr_scores, int64_t ld_ocr, const float* targets, int64_t ld_t,
                            const float* loss_mask, int R, int V, int No, float grad_scale, const float* global_count, float* loss, void* d_fixed,
                            int64_t ld_dfixed, float* d_ocr, int64_t ld_docr, void* stream) {
  SAM_REQUIRE(fixed_scores && ocr_scores && targets && loss_mask && loss && d_fixed && d_ocr, "sam_bce_loss: null pointer");
  SAM_REQUIRE(R > 0 && V > 0 && No >= 0, "sam_bce_loss: bad shape");
  hipStream_t st = (hipStream_t)stream;
  hipError_t e = hipMemsetAsync(loss, 0, sizeof(float), st);
  if (e != hipSuccess) { sam_set_error("sam_bce_loss: memset: %s", hipGetErrorString(e)); return (int)e; }
  const bool pairs = V % 2 == 0 && No % 2 == 0 && ld_fixed % 2 == 0 && ld_ocr % 2 == 0 && ld_t % 2 == 0 && ld_dfixed % 2 == 0 && ld_docr % 2 == 0 &&
                     ((uintptr_t)fixed_scores % 8 == 0) && ((uintptr_t)ocr_scores % 8 == 0) && ((uintptr_t)targets % 8 == 0) && ((uintptr_t)d_ocr % 8 == 0) &&
                     ((uintptr_t)d_fixed % 4 == 0);
  const int per = pairs ? 2 : 1;
  // one fp32 atomic per block lands on `loss`: keep the block count near 1-2 k (6144 blocks spent 60 us queueing on that one address)
  const int chunks = max(1, min(min(8, 1024 / R), ((V + No + per - 1) / per + 255) / 256));
  if (pairs)
    bce_kernel<2><<<dim3(R, chunks), dim3(256), 0, st>>>(fixed_scores, ld_fixed, ocr_scores, ld_ocr, targets, ld_t, loss_mask, R, V, No, grad_scale, global_count, loss,
                                                         (bf16_t*)d_fixed, ld_dfixed, d_ocr, ld_docr);
  else
    bce_kernel<1><<<dim3(R, chunks), dim3(256), 0, st>>>(fixed_scores, ld_fixed, ocr_scores, ld_ocr, targets, ld_t, loss_mask, R, V, No, grad_scale, global_count, loss,
                                                         (bf16_t*)d_fixed, ld_dfixed, d_ocr, ld_docr);
  SAM_LAUNCH_CHECK();
  return SAM_OK;
}

extern "C" int sam_ptr_scores_fwd(const void* q, const void* k, const uint8_t* ocr_mask, int B, int S, int No, int D, float scale, float* out,
                                  int64_t ld_out_b, int64_t ld_out_s, void* stream) {
  SAM_REQUIRE(q && k && ocr_mask && out, "sam_ptr_scores_fwd: null pointer");
  SAM_REQUIRE(B > 0 && S > 0 && No > 0 && D > 0 && D % 4 == 0, "sam_ptr_scores_fwd: bad shape");
  if (S <= 16 && No <= 64 && D == 768 && ((uintptr_t)q % 16) == 0 && ((uintptr_t)k % 16) == 0)      // the model's shape: matrix cores
    ptr_fwd_mfma_kernel<6><<<dim3(B), dim3(256), 0, (hipStream_t)stream>>>((const bf16_t*)q, (const bf16_t*)k, ocr_mask, S, No, D, scale, out, ld_out_b, ld_out_s);
  else
    ptr_fwd_kernel<<<dim3(B, S), dim3(256), 0, (hipStream_t)stream>>>((const bf16_t*)q, (const bf16_t*)k, ocr_mask, S, No, D, scale, out, ld_out_b, ld_out_s);
  SAM_LAUNCH_CHECK();
  return SAM_OK;
}
extern "C" int sam_ptr_scores_bwd(const float* dscores, int64_t ld_b, int64_t ld_s, const void* q, const void* k, int B, int S, int No, int D, float scale,
                                  void* dq, void* dk, void* stream) {
  SAM_REQUIRE(dscores && q && k && dq && dk, "sam_ptr_scores_bwd: null pointer");
  SAM_REQUIRE(B > 0 && S > 0 && No > 0 && D > 0 && D % 4 == 0 && S <= 256 && No <= 256, "sam_ptr_scores_bwd: bad shape (S, No <= 256)");
  ptr_bwd_kernel<<<dim3(B, S + No), dim3(256), 0, (hipStream_t)stream>>>(dscores, ld_b, ld_s, (const bf16_t*)q, (const bf16_t*)k, S, No, D,
                                                                                          scale, (bf16_t*)dq, (bf16_t*)dk);
  SAM_LAUNCH_CHECK();
  return SAM_OK;
}

extern "C" int sam_embedding_bwd(const void* dy, int64_t ldd, const int64_t* idx, int T, int D, int rows, int64_t padding_idx, float* grad, int64_t ldg, uint8_t* touched,
                                void* stream) {
  SAM_REQUIRE(dy && idx && grad, "sam_embedding_bwd: null pointer");
  SAM_REQUIRE(T > 0 && D > 0 && D % 4 == 0 && ldd % 4 == 0 && rows > 0, "sam_embedding_bwd: bad shape");
  static int atomic = -1;
  if (atomic < 0) { const char* e = getenv("SAM_EMBED_BWD_ATOMIC"); atomic = e ? atoi(e) : 0; }       // (A/B: the round-1..4 kernel)
  if (atomic || ldg % 4 != 0 || ((uintptr_t)grad % 16) != 0)
    embedding_bwd_kernel<<<dim3(T), dim3(256), 0, (hipStream_t)stream>>>((const bf16_t*)dy, ldd, idx, T, D, rows, padding_idx, grad, ldg, touched);
  else
    embedding_bwd_dedup_kernel<<<dim3(T), dim3(256), 0, (hipStream_t)stream>>>((const bf16_t*)dy, ldd, idx, T, D, rows, padding_idx, grad, ldg, touched);
  SAM_LAUNCH_CHECK();
  return SAM_OK;
}

extern "C" int sam_embedding_bwd_sorted(const void* dy, int64_t ldd, const int64_t* idx_sorted, int T, int D, int rows, int64_t padding_idx, float* grad, int64_t ldg,
                                       uint8_t* touched, void* stream) {
  SAM_REQUIRE(dy && idx_sorted && grad, "sam_embedding_bwd_sorted: null pointer");
  SAM_REQUIRE(T > 0 && D > 0 && D % 4 == 0 && ldd % 4 == 0 && ldg % 4 == 0 && rows > 0 && ((uintptr_t)grad % 16 == 0), "sam_embedding_bwd_sorted: bad shape");
  embedding_bwd_sorted_kernel<<<dim3(T), dim3(256), 0, (hipStream_t)stream>>>((const bf16_t*)dy, ldd, idx_sorted, T, D, rows, padding_idx, grad, ldg, touched);
  SAM_LAUNCH_CHECK();
  return SAM_OK;
}

static int fill_sparse(SparseRows& sp, const sam_sparse_rows* s, int64_t n) {
  sp = SparseRows{0, 0, 1, 0, nullptr};
  if (!s || !s->touched) return SAM_OK;
  SAM_REQUIRE(s->row_len > 0 && s->row_len % 4 == 0 && s->lo % 4 == 0 && s->lo >= 0 && s->hi <= n && s->hi > s->lo && (s->hi - s->lo) % s->row_len == 0,
              "row-sparse region: need 0 <= lo < hi <= n, lo and row_len multiples of 4, (hi - lo) a whole number of rows");
  sp.lo4 = s->lo >> 2; sp.hi4 = s->hi >> 2; sp.row_len4 = s->row_len >> 2; sp.rows = (int)((s->hi - s->lo) / s->row_len); sp.touched = s->touched;
  return SAM_OK;
}
extern "C" int64_t sam_sumsq_ws_bytes(void) { return (int64_t)(SUMSQ_BLOCKS + SPARSE_ROW_BLOCKS) * sizeof(float); }
extern "C" int sam_sumsq_f32(const float* g, int64_t n, const sam_sparse_rows* sparse, float* out, float* ws, void* stream) {
  SAM_REQUIRE(g && out && ws && n > 0 && ((uintptr_t)g % 16 == 0), "sam_sumsq_f32: bad arguments");
  SparseRows sp;
  if (int rc = fill_sparse(sp, sparse, n)) return rc;
  hipStream_t st = (hipStream_t)stream;
  const int dense = (int)min((int64_t)SUMSQ_BLOCKS, ((n >> 2) + 255) / 256 + 1), blocks = dense + (sp.touched ? SPARSE_ROW_BLOCKS : 0);
  sumsq_partial_kernel<<<dim3(blocks), dim3(256), 0, st>>>(g, n, sp, dense, ws);
  sumsq_final_kernel<<<dim3(1), dim3(256), 0, st>>>(ws, blocks, out);
  SAM_LAUNCH_CHECK();
  return SAM_OK;
}

static int adam_launch(float* p, float* g, float* m, float* v, void* p_bf16, int64_t n, const int64_t* seg_end, const float* seg_lr, int nseg, float beta1,
                       float beta2, float eps, int64_t step, const float* gnorm_sq, float max_norm, const float* dev_sched, const sam_sparse_rows* sparse, void* stream,
                       int64_t lo = 0, int64_t hi = -1, int zero_g = 0, const int* gate = nullptr, int max_blocks = 0) {
  SAM_REQUIRE(p && g && m && v && seg_end && (seg_lr || dev_sched), "sam_adam_step: null pointer");
  SAM_REQUIRE(n > 0 && n % 4 == 0 && nseg >= 1 && nseg <= 8 && (step >= 1 || dev_sched), "sam_adam_step: need n %% 4 == 0, 1..8 segments, step >= 1");
  AdamSegs segs = {};
  segs.n = nseg;
  for (int s = 0; s < nseg; ++s) {
    SAM_REQUIRE(seg_end[s] % 4 == 0 && (s == 0 || seg_end[s] >= seg_end[s - 1]), "sam_adam_step: segment ends must be ascending multiples of 4");
    segs.end[s] = seg_end[s]; segs.lr[s] = seg_lr ? seg_lr[s] : 0.f;
  }
  SAM_REQUIRE(seg_end[nseg - 1] == n, "sam_adam_step: last segment must end at n");
  const float bc1 = 1.0f - powf(beta1, (float)(step >= 1 ? step : 1));
  const float bc2 = 1.0f - powf(beta2, (float)(step >= 1 ? step : 1));
  SparseRows sp;
  if (int rc = fill_sparse(sp, sparse, n)) return rc;
  // the launch's range [lo, hi) of the buffer: the row-sparse region lies inside it (its touched rows are walked) or outside (they are not)
  if (hi < 0) hi = n;
  SAM_REQUIRE(lo >= 0 && lo < hi && hi <= n && lo % 4 == 0 && hi % 4 == 0, "sam_adam_step: range [lo, hi) must be a non-empty multiple-of-4 piece of [0, n)");
  const int64_t lo4 = lo >> 2, hi4 = hi >> 2, gap = sp.hi4 - sp.lo4;
  const bool rows_inside = sp.touched && lo4 <= sp.lo4 && sp.hi4 <= hi4;
  SAM_REQUIRE(!sp.touched || rows_inside || hi4 <= sp.lo4 || lo4 >= sp.hi4, "sam_adam_step: the range may not cut the row-sparse region");
  const int64_t jlo = lo4 <= sp.lo4 ? lo4 : lo4 - gap, jhi = hi4 <= sp.lo4 ? hi4 : hi4 - gap;      // the same range in the dense index space
  if (!rows_inside) sp.touched = nullptr, sp.rows = 0;
  // max_blocks: a piece that runs UNDERNEATH other work keeps to a few blocks per CU -- 4096 grid-stride blocks fill every wave slot of the GPU for the
  // whole launch and the small kernels beside it wait for them to retire
  const int dense = (int)max((int64_t)1, min((int64_t)(max_blocks > 0 ? max_blocks : 4096), (jhi - jlo + 255) / 256)), blocks = dense + (sp.touched ? SPARSE_ROW_BLOCKS : 0);
  adam_kernel<<<dim3(blocks), dim3(256), 0, (hipStream_t)stream>>>(p, g, m, v, (bf16_t*)p_bf16, n, segs, beta1, beta2, eps, bc1, 1.0f / sqrtf(bc2), gnorm_sq, max_norm,
                                                                   dev_sched, sp, dense, jlo, jhi, zero_g, gate);
  SAM_LAUNCH_CHECK();
  return SAM_OK;
}
extern "C" int sam_adam_step(float* p, float* g, float* m, float* v, void* p_bf16, int64_t n, const int64_t* seg_end, const float* seg_lr, int nseg,
                             float beta1, float beta2, float eps, int64_t step, const float* gnorm_sq, float max_norm, const sam_sparse_rows* sparse, void* stream) {
  return adam_launch(p, g, m, v, p_bf16, n, seg_end, seg_lr, nseg, beta1, beta2, eps, step, gnorm_sq, max_norm, nullptr, sparse, stream);
}
extern "C" int sam_adam_step_dev(float* p, float* g, float* m, float* v, void* p_bf16, int64_t n, const int64_t* seg_end, int nseg, float beta1, float beta2,
                                 float eps, const float* dev_sched, const float* gnorm_sq, float max_norm, const sam_sparse_rows* sparse, void* stream) {
  SAM_REQUIRE(dev_sched, "sam_adam_step_dev: null schedule");
  return adam_launch(p, g, m, v, p_bf16, n, seg_end, nullptr, nseg, beta1, beta2, eps, 0, gnorm_sq, max_norm, dev_sched, sparse, stream);
}

extern "C" int sam_adam_step_range(float* p, float* g, float* m, float* v, void* p_bf16, int64_t n, const int64_t* seg_end, int nseg, float beta1, float beta2,
                                   float eps, const float* dev_sched, const float* gnorm_sq, float max_norm, const sam_sparse_rows* sparse, int64_t lo, int64_t hi,
                                   int zero_grad, const int32_t* gate, int max_blocks, void* stream) {
  SAM_REQUIRE(dev_sched && max_blocks >= 0, "sam_adam_step_range: null schedule");
  return adam_launch(p, g, m, v, p_bf16, n, seg_end, nullptr, nseg, beta1, beta2, eps, 0, gnorm_sq, max_norm, dev_sched, sparse, stream, lo, hi, zero_grad, gate,
                     max_blocks);
}

// Head node of a captured (hipGraph) training step: everything that changes from replay to replay and used to be a by-value argument lives in
// device memory and is advanced HERE, on the device, in stream order -- the host never writes it while replays are in flight (a pinned
// host buffer copied per step raced with the replays the host had queued ahead of the GPU).
//   rng_state[1] += offset_stride             fresh dropout masks for this replay (rng_state may be NULL)
//   t = ++step[0]                              optimizer step number, 1-based
//   dev_sched[s] = base_lr[s] * lambda(t - 1)  LambdaLR of sam/task_utils.py:48-54, evaluated at the pre-increment step like current_lrs()
//   dev_sched[n] = 1 - beta1^t, dev_sched[n+1] = 1 - beta2^t   (double precision, rounded once)
struct StepSched { double base_lr[8]; double warmup_factor, lr_decay, beta1, beta2; long long warmup_iters, decay_iters[4]; int nseg, n_decay; };
__global__ void step_advance_kernel(unsigned long long* rng_state, unsigned long long offset_stride, long long* step, StepSched sc, float* dev_sched) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  if (rng_state) rng_state[1] += offset_stride;
  const long long it = step[0], t = it + 1;
  step[0] = t;
  double lam;
  if (it <= sc.warmup_iters) {
    const double alpha = (double)it / (double)sc.warmup_iters;
    lam = sc.warmup_factor * (1.0 - alpha) + alpha;
  } else {
    int k = 0;
    for (int q = 0; q < sc.n_decay; ++q) k += sc.decay_iters[q] <= it ? 1 : 0;      // bisect_right
    lam = pow(sc.lr_decay, (double)k);
  }
  for (int s = 0; s < sc.nseg; ++s) dev_sched[s] = (float)(sc.base_lr[s] * lam);
  dev_sched[sc.nseg] = (float)(1.0 - pow(sc.beta1, (double)t));
  dev_sched[sc.nseg + 1] = (float)(1.0 - pow(sc.beta2, (double)t));
}
extern "C" int sam_step_advance(unsigned long long* rng_state, uint64_t offset_stride, int64_t* step_counter, const sam_lr_schedule* sched, float* dev_sched,
                                void* stream) {
  SAM_REQUIRE(step_counter && sched && dev_sched, "sam_step_advance: null pointer");
  SAM_REQUIRE(sched->nseg >= 1 && sched->nseg <= 8 && sched->n_decay >= 0 && sched->n_decay <= 4 && sched->warmup_iters >= 1,
              "sam_step_advance: 1..8 segments, at most 4 decay points, warmup_iters >= 1");
  StepSched sc = {};
  for (int s = 0; s < sched->nseg; ++s) sc.base_lr[s] = sched->base_lr[s];
  for (int q = 0; q < sched->n_decay; ++q) sc.decay_iters[q] = sched->decay_iters[q];
  sc.warmup_factor = sched->warmup_factor; sc.lr_decay = sched->lr_decay; sc.beta1 = sched->beta1; sc.beta2 = sched->beta2;
  sc.warmup_iters = sched->warmup_iters; sc.nseg = sched->nseg; sc.n_decay = sched->n_decay;
  step_advance_kernel<<<dim3(1), dim3(64), 0, (hipStream_t)stream>>>(rng_state, (unsigned long long)offset_stride, (long long*)step_counter, sc, dev_sched);
  SAM_LAUNCH_CHECK();
  return SAM_OK;
}

// out = dropout(a [+ b]) on bf16 [M, D] rows: the element-wise dropout of the object / OCR input encoders (sam/sa_m4c.py:224,263: F.dropout on the
// sum of the two LayerNorm outputs) and, with b = NULL and dy in place of a, its backward (the mask is regenerated from the same counters).
// One 16-byte chunk per thread = one (row, col / 8) draw of the hidden-state dropout stream.
__global__ __launch_bounds__(256) void add_dropout_kernel(const bf16_t* __restrict__ a, int64_t lda, const bf16_t* __restrict__ b, int64_t ldb, bf16_t* __restrict__ out,
                                                          int64_t ldo, int M, int chunks, unsigned thr16, float inv_keep, unsigned seed_lo, unsigned seed_hi,
                                                          unsigned off_lo, unsigned off_hi, const unsigned long long* rng_state) {
  rng_resolve(rng_state, seed_lo, seed_hi, off_lo, off_hi);
  const int64_t total = (int64_t)M * chunks;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    const int row = (int)(t / chunks), c = (int)(t - (int64_t)row * chunks);
    const uint4 va = *reinterpret_cast<const uint4*>(a + (int64_t)row * lda + 8 * c);
    float v[8] = {bf_lo(va.x), bf_hi(va.x), bf_lo(va.y), bf_hi(va.y), bf_lo(va.z), bf_hi(va.z), bf_lo(va.w), bf_hi(va.w)};
    if (b) {
      const uint4 vb = *reinterpret_cast<const uint4*>(b + (int64_t)row * ldb + 8 * c);
      v[0] += bf_lo(vb.x); v[1] += bf_hi(vb.x); v[2] += bf_lo(vb.y); v[3] += bf_hi(vb.y);
      v[4] += bf_lo(vb.z); v[5] += bf_hi(vb.z); v[6] += bf_lo(vb.w); v[7] += bf_hi(vb.w);
    }
    if (thr16) {
      const u32x4 rn = hidden_dropout_bits((unsigned)row, (unsigned)c, off_lo, off_hi, seed_lo, seed_hi);
      const unsigned w4[4] = {rn.x, rn.y, rn.z, rn.w};
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        v[2 * r] = (w4[r] & 0xffffu) >= thr16 ? v[2 * r] * inv_keep : 0.f;
        v[2 * r + 1] = (w4[r] >> 16) >= thr16 ? v[2 * r + 1] * inv_keep : 0.f;
      }
    }
    *reinterpret_cast<uint4*>(out + (int64_t)row * ldo + 8 * c) =
        make_uint4(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7]));
  }
}

extern "C" int sam_add_dropout_bf16(const void* a, int64_t lda, const void* b, int64_t ldb, void* out, int64_t ldo, int M, int D, float p_drop, uint64_t seed,
                                    uint64_t offset, void* stream) {
  SAM_REQUIRE(a && out && M > 0 && D > 0 && D % 8 == 0 && lda % 8 == 0 && ldo % 8 == 0 && (!b || ldb % 8 == 0), "sam_add_dropout_bf16: D and the row strides must be multiples of 8");
  SAM_REQUIRE(((uintptr_t)a % 16 == 0) && ((uintptr_t)out % 16 == 0) && ((uintptr_t)b % 16 == 0), "sam_add_dropout_bf16: operands must be 16-byte aligned");
  SAM_REQUIRE(p_drop >= 0.f && p_drop < 1.f, "sam_add_dropout_bf16: p_drop out of range");
  const unsigned thr16 = dropout_thr16(p_drop);
  const float inv_keep = thr16 ? 1.0f / (1.0f - (float)thr16 / 65536.0f) : 1.0f;
  const int64_t total = (int64_t)M * (D / 8);
  const int blocks = (int)min((int64_t)2048, (total + 255) / 256);
  add_dropout_kernel<<<dim3(blocks), dim3(256), 0, (hipStream_t)stream>>>((const bf16_t*)a, lda, (const bf16_t*)b, ldb, (bf16_t*)out, ldo, M, D / 8, thr16, inv_keep,
                                                                         (unsigned)seed, (unsigned)(seed >> 32), (unsigned)offset, (unsigned)(offset >> 32),
                                                                         sam_get_rng_state());
  SAM_LAUNCH_CHECK();
  return SAM_OK;
}

extern "C" int sam_cast_f32_to_bf16(const float* x, void* y, int64_t n, void* stream) {
  SAM_REQUIRE(x && y && n > 0 && n % 4 == 0, "sam_cast_f32_to_bf16: need n %% 4 == 0");
  const int blocks = (int)min((int64_t)4096, ((n >> 2) + 255) / 256);
  cast_bf16_kernel<<<dim3(blocks), dim3(256), 0, (hipStream_t)stream>>>(x, (bf16_t*)y, n >> 2);
  SAM_LAUNCH_CHECK();
  return SAM_OK;
}
