// Grouped weight-gradient GEMM on the 8-wave core (gfx950): dW_p += dy_p^T x_p (and db_p += colsum(dy_p)) for up to 12 problems in ONE launch --
// the four weight gradients of an encoder layer (sam/sa_m4c.py:554-560,653,678-680 in the backward direction).
//
// Both operands are k-strided (the contraction index R = B*N token rows is the slow index of dy [R, M] and x [R, N]): 64-column panel images in LDS,
// fragments by ds_read_b64_tr_b16, the staggered two-group schedule of gemm8.hip.  256 x 256 tiles: an encoder layer has 9 + 36 + 36 + 27 = 108 of
// them against 256 CUs, each 182 k-tiles deep at B = 64 -- so the K range of every tile is SPLIT IN TWO (216 blocks, one per CU, all co-resident)
// and the two halves are combined inside the launch: each block of a pair owns one 128-row half of the tile, ships the other half of its
// accumulators to the partner through a workspace slot (agent-scope 16-byte stores -> s_waitcnt -> flag), waits for the partner's half
// (flag poll -> agent-scope loads), adds it in a fixed order and accumulates into the fp32 gradient.  No atomics, bit-reproducible, no
// partial-sum pass through HBM beyond the 128 KB each block hands over (27 MB per layer against 286 MB of operands).
// Block -> (tile, half): each XCD (block id % 8, own L2) gets a contiguous run of tiles with BOTH halves, tiles walked n-fastest: the blocks of an
// XCD share dy / x panels and -- halves progressing in lock step -- touch them at the same time.
#include "gemm8_dev.h"
#include <stdlib.h>

using namespace samgemm;
using namespace samgemm8;
namespace {

constexpr int BM = 256, BN = 256;
constexpr int TM = BM / 32, TN = BN / 64, SA = BM / 64, SB = BN / 64, RB = TM / 2;
constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, STAGE = A_BYTES + B_BYTES;
constexpr int SLOT_FLOATS = 128 * 256 + 128;          // half a tile of accumulators + its 128 bias-gradient partials

struct WProb {
  const bf16_t* A; int64_t lda;      // dy [K rows, M]
  const bf16_t* B; int64_t ldb;      // x  [K rows, N]
  float* C; int64_t ldc;             // dW [M, N] fp32, accumulated
  float* bias_grad;                  // [M] or NULL
  int M, N, K, tiles_m, tiles_n, accumulate;
};
struct WArgs {
  WProb p[samgemm::SAM_MAX_GROUP8];
  int tile_start[samgemm::SAM_MAX_GROUP8 + 1];
  int count, split, dbg;
  int n_long;          // > 0: the first n_long items (the tiles of the deepest problems, a multiple of 8) form a segment of their own (see below)
  float* ws;           // [tiles][2] slots of SLOT_FLOATS
  unsigned* flags;     // [tiles][2], zero between launches
};

// agent-scope (sc1) slot accesses: see the exchange at the end of the kernel
constexpr int AUX_SC1 = 0x10;
typedef unsigned int vu4 __attribute__((ext_vector_type(4)));
template <int T0>
__device__ __forceinline__ void slot_send(float* slot, const f32x4 (&acc)[TN][TM], int wave, int lane) {
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)slot, 0, 0x7fffffff, 0x00020000);
#pragma unroll
  for (int tn = 0; tn < TN; ++tn)
#pragma unroll
    for (int t = 0; t < RB; ++t) {
      const f32x4 v = acc[tn][T0 + t];
      const vu4 u = {__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3])};
      __builtin_amdgcn_raw_buffer_store_b128(u, rs, (((tn * RB + t) * 8 + wave) * 64 + lane) * 16, 0, AUX_SC1);
    }
}
template <int T0, bool MINE_FIRST>
__device__ __forceinline__ void slot_recv_add(const float* slot, f32x4 (&acc)[TN][TM], int wave, int lane) {
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)slot, 0, 0x7fffffff, 0x00020000);
#pragma unroll
  for (int tn = 0; tn < TN; ++tn)
#pragma unroll
    for (int t = 0; t < RB; ++t) {
      const vu4 u = __builtin_amdgcn_raw_buffer_load_b128(rs, (((tn * RB + t) * 8 + wave) * 64 + lane) * 16, 0, AUX_SC1);
      const f32x4 o = {__uint_as_float(u[0]), __uint_as_float(u[1]), __uint_as_float(u[2]), __uint_as_float(u[3])};
      acc[tn][T0 + t] = MINE_FIRST ? acc[tn][T0 + t] + o : o + acc[tn][T0 + t];
    }
}
__device__ __forceinline__ void slot_store1(float* slot, int idx, float v) {
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)slot, 0, 0x7fffffff, 0x00020000);
  __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), rs, idx * 4, 0, AUX_SC1);
}
__device__ __forceinline__ float slot_load1(const float* slot, int idx) {
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)slot, 0, 0x7fffffff, 0x00020000);
  return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, idx * 4, 0, AUX_SC1));
}

__global__ __launch_bounds__(512, 2) void gemm8w_kernel(WArgs w) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), i = lane & 15, g = lane >> 4;
  const int wr = wave >> 2, wc = wave & 3;
  // ---- block -> (problem, tile, k half)
  // Mixed depths (an MMT pair, 216 tiles of 182 k-tiles, with TextBert's 324 tiles of 20): workgroups are dispatched in id order, so the first n_long
  // ids take the deep tiles -- one per CU, all starting together, the lock-step panel sharing inside an XCD as before -- and the ids behind them
  // the shallow ones: 40 go to the CUs the deep round leaves idle, the rest follow as those finish (7 rounds of ~35 us inside the deep round's 300).
  // Each segment deals its items to the XCDs in contiguous runs (n_long is a multiple of 8: id % 8 is the XCD in both).
  const int nitem = w.tile_start[w.count] * w.split;
  int item;
  {
    int bid = blockIdx.x, seg0 = 0, seglen = nitem;
    if (w.n_long > 0) {
      if (bid < w.n_long) seglen = w.n_long;
      else { seg0 = w.n_long; seglen = nitem - w.n_long; bid -= w.n_long; }
    }
    const int q = seglen / 8, r = seglen % 8, xcd = bid % 8, loc = bid / 8;
    item = seg0 + (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
  }
  const int gtile = item / w.split, half = item - gtile * w.split;
  int pi = 0;
#pragma unroll
  for (int q = 1; q < samgemm::SAM_MAX_GROUP8; ++q)
    if (q < w.count && gtile >= w.tile_start[q]) pi = q;
  const WProb& P = w.p[pi];
  const int tile = gtile - w.tile_start[pi];
  // walk the tiles so that an XCD's contiguous run partitions the WIDER operand and replicates only the narrower one (dW2 = dy2^T h walked
  // n-fastest would make every XCD stream all of h)
  const bool m_fast = P.tiles_n > P.tiles_m;
  const int m0 = (m_fast ? tile % P.tiles_m : tile / P.tiles_n) * BM, n0 = (m_fast ? tile / P.tiles_m : tile % P.tiles_n) * BN;
  const int KT = P.K / BK, per = (KT + w.split - 1) / w.split;
  const int kt0 = half * per, total = min(KT, kt0 + per) - kt0;      // >= 1 by construction (host keeps split <= KT)

  unsigned offA[SA], offB[SB];
  src_offsets<false, SA>(offA, P.lda, m0, P.M, wave, lane);
  src_offsets<false, SB>(offB, P.ldb, n0, P.N, wave, lane);
  const unsigned kstepA = (unsigned)(BK * P.lda * 2), kstepB = (unsigned)(BK * P.ldb * 2);
  int ua = 0, ub = 0;
  // (dbg bit 3, tuning: every DMA re-reads the tile's first k-tile -- L2-resident: the loop without the HBM wait; bit 2: no DMA inside the loop at all)
#define SAM_DMA_A() do { dma_slices<SA>(P.A, smem + (ua & 1) * STAGE + wave * (SA * 1024), offA, (w.dbg & 8) ? 0u : (kt0 + ua) * kstepA); ++ua; } while (0)
#define SAM_DMA_B() do { dma_slices<SB>(P.B, smem + (ub & 1) * STAGE + A_BYTES + wave * (SB * 1024), offB, (w.dbg & 8) ? 0u : (kt0 + ub) * kstepB); ++ub; } while (0)

  f32x4 acc[TN][TM], accb = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int a = 0; a < TN; ++a)
#pragma unroll
    for (int b = 0; b < TM; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
  // bias gradient = column sums of dy = sum_k A(m, k): one extra MFMA per A fragment that is in registers anyway, against a SELECTOR operand --
  // ones in row tm, zeros elsewhere -- so that all TM fragments of the wave accumulate into one 16 x 16 block (row tm = the sums of fragment tm;
  // a plain ones operand would need an accumulator per fragment, which this kernel has no registers for).  Only tiles of the first tile column
  // compute it, and the four waves that share the same rows take the k-tiles in turn (one wave doing all of them would load its SIMD with
  // 25 % more MFMA work than the other three).
  const bool do_bias = P.bias_grad != nullptr && n0 == 0;
  const short one_bf16 = (short)0x3F80;
  const bf16x8 ones = {one_bf16, one_bf16, one_bf16, one_bf16, one_bf16, one_bf16, one_bf16, one_bf16};
  const bf16x8 zeros = {0, 0, 0, 0, 0, 0, 0, 0};

  SAM_DMA_A(); SAM_DMA_B();
  if (total > 1) { SAM_DMA_B(); vmwait<SB>(); }
  else vmwait<0>();
  __builtin_amdgcn_s_barrier();
  if (wr == 1) __builtin_amdgcn_s_barrier();

  const int sig = ((i >> 3) & 1) | ((g & 1) << 1);
  bf16x8 af[RB][2], bfr[TN][2];
  for (int u = 0; u < total; ++u) {
    const unsigned char* stA = smem + (u & 1) * STAGE;
    const unsigned char* stB = stA + A_BYTES;
    const bool my_bias = do_bias && ((u & 3) == wc);
    // ---- phase 0
#pragma unroll
    for (int x = 0; x < TN; ++x)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) bfr[x][ks] = frag<false>(stB, wc * (BN / 4) + x * 16, ks, i, g, sig);
#pragma unroll
    for (int x = 0; x < RB; ++x)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) af[x][ks] = frag<false>(stA, wr * (BM / 2) + x * 16, ks, i, g, sig);
    if (ua < total && !(w.dbg & 4)) SAM_DMA_A();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int x = 0; x < RB; ++x)
#pragma unroll
        for (int y = 0; y < TN; ++y) acc[y][x] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[y][ks], af[x][ks], acc[y][x], 0, 0, 0);
    if (my_bias) {
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int x = 0; x < RB; ++x) accb = __builtin_amdgcn_mfma_f32_16x16x32_bf16(i == x ? ones : zeros, af[x][ks], accb, 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    // ---- phase 1
#pragma unroll
    for (int x = 0; x < RB; ++x)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) af[x][ks] = frag<false>(stA, wr * (BM / 2) + (RB + x) * 16, ks, i, g, sig);
    if (ub < total && !(w.dbg & 4)) { SAM_DMA_B(); vmwait<SB>(); }
    else vmwait<0>();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int x = 0; x < RB; ++x)
#pragma unroll
        for (int y = 0; y < TN; ++y) acc[y][RB + x] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[y][ks], af[x][ks], acc[y][RB + x], 0, 0, 0);
    if (my_bias) {
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int x = 0; x < RB; ++x) accb = __builtin_amdgcn_mfma_f32_16x16x32_bf16(i == RB + x ? ones : zeros, af[x][ks], accb, 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  }
  if (wr == 0) __builtin_amdgcn_s_barrier();       // both wave groups in step again: plain __syncthreads() below
#undef SAM_DMA_A
#undef SAM_DMA_B

  if (w.dbg & 1) return;                           // (tuning: the k loop alone)
  // ---- bias partials of the four waves sharing a row range -> one vector per row half in LDS: bsum[wr * 128 + row]
  float* bsum = reinterpret_cast<float*>(smem);            // (the operand stages are dead)
  if (do_bias) {
    __syncthreads();
    float* part = bsum + 256 + wave * 128;
    if (g < TM / 4) {
#pragma unroll
      for (int r = 0; r < 4; ++r) part[(4 * g + r) * 16 + i] = accb[r];      // D[n][m], n = 4 g + r = fragment, m = i
    }
    __syncthreads();
    if (wc == 0) {
      for (int r = lane; r < 128; r += 64) bsum[wr * 128 + r] = part[r] + part[128 + r] + part[256 + r] + part[384 + r];     // waves wr*4 .. wr*4+3 (fixed order)
    }
    __syncthreads();
  }

  const int mw = m0 + wr * (BM / 2), nw = n0 + wc * (BN / 4);
  const bool full = m0 + BM <= P.M && n0 + BN <= P.N;
  GemmArgs ea = {};                                  // what the shared fp32-accumulate epilogue reads
  ea.M = P.M; ea.N = P.N;
  if (w.split == 1) {
    gemm_epilogue8<TM, TN, SAM_EPI_NONE, float, 0, RB>(ea, acc, mw, nw, full, P.C, P.ldc, P.accumulate, i, g);
    gemm_epilogue8<TM, TN, SAM_EPI_NONE, float, RB, TM>(ea, acc, mw, nw, full, P.C, P.ldc, P.accumulate, i, g);
    if (do_bias && wc == 0)
      for (int r = lane; r < 128; r += 64)
        if (mw + r < P.M) P.bias_grad[mw + r] = (P.accumulate ? P.bias_grad[mw + r] : 0.f) + bsum[wr * 128 + r];
    return;
  }
  // ---- exchange: of every wave's TM row fragments this block keeps the half `half` (rows [64 half, 64 half + 64) of the wave's 128) and ships
  // the other half to the partner.  All eight waves send, receive and store.  The slot traffic and the flags are agent-scope accesses (sc1:
  // written through to / read from the memory side, past the per-CU L1 and the per-XCD L2), ordered by the s_waitcnt between the data stores
  // and the flag store on one side and by the flag load preceding the data loads on the other: no cache write-back or invalidate is needed
  // (a release / acquire fence pair measured ~10 us here: it writes back / invalidates the whole L2 of the XCD, 27 blocks doing so at once).
  float* slot_out = w.ws + ((int64_t)gtile * 2 + half) * SLOT_FLOATS;            // written by this block, read by the partner
  const float* slot_in = w.ws + ((int64_t)gtile * 2 + (1 - half)) * SLOT_FLOATS;
  if (half == 0) slot_send<RB>(slot_out, acc, wave, lane);
  else slot_send<0>(slot_out, acc, wave, lane);
  if (do_bias && wc == 0) slot_store1(slot_out, 128 * 256 + wr * 64 + lane, bsum[wr * 128 + (1 - half) * 64 + lane]);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid == 0) {
    __hip_atomic_store(w.flags + gtile * 2 + half, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned* fin = w.flags + gtile * 2 + (1 - half);
    // bounded: HIP does not PROMISE that both blocks of a pair are resident at once (the argument in gemm8w_grouped rests on in-order dispatch).
    // Should the partner never show up the block gives up after ~1 s, raises the workspace's error word (word 0; the host reads it when it next synchronises: ops.grouped_ws_check)
    // and finishes with garbage instead of hanging the device.
    unsigned spins = 0;
    while (__hip_atomic_load(fin, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) {
      __builtin_amdgcn_s_sleep(1);
      if (++spins > (1u << 24)) { __hip_atomic_store(w.flags - 64, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
    }
    __hip_atomic_store(fin, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // consumed: the next launch finds every flag at zero again
  }
  __syncthreads();
  // fixed order: C + ((k half 0) + (k half 1)), whichever of the two halves this block computed
  if (half == 0) {
    slot_recv_add<0, true>(slot_in, acc, wave, lane);
    gemm_epilogue8<TM, TN, SAM_EPI_NONE, float, 0, RB>(ea, acc, mw, nw, full, P.C, P.ldc, P.accumulate, i, g);
  } else {
    slot_recv_add<RB, false>(slot_in, acc, wave, lane);
    gemm_epilogue8<TM, TN, SAM_EPI_NONE, float, RB, TM>(ea, acc, mw, nw, full, P.C, P.ldc, P.accumulate, i, g);
  }
  if (do_bias && wc == 0) {
    const int r = half * 64 + lane;
    if (mw + r < P.M) {
      const float a0 = bsum[wr * 128 + r], a1 = slot_load1(slot_in, 128 * 256 + wr * 64 + lane);
      P.bias_grad[mw + r] = (P.accumulate ? P.bias_grad[mw + r] : 0.f) + (half == 0 ? a0 + a1 : a1 + a0);
    }
  }
}

}  // namespace

namespace samgemm {

// [64 words: word 0 = error flag][tiles * 2 pair flags, padded to 64 words][tiles * 2 slots]
int64_t gemm8w_ws_bytes(int tiles) { return (int64_t)tiles * 2 * SLOT_FLOATS * 4 + (int64_t)(64 + (tiles * 2 + 63) / 64 * 64) * 4 + 256; }

// returns SAM_ERR_UNSUPPORTED when the problem set is not one for this kernel (the caller falls back to the 4-wave grouped kernel)
int gemm8w_grouped(const sam_gemm_desc* descs, int count, hipStream_t st) {
  WArgs w = {};
  w.count = count;
  int tiles = 0, min_kt = 1 << 30, max_k = 0;
  // deepest problems first (stable): their tiles become the launch's first segment
  int order[SAM_MAX_GROUP8];
  for (int q = 0; q < count; ++q) { order[q] = q; max_k = descs[q].K > max_k ? descs[q].K : max_k; }
  for (int a = 1; a < count; ++a)
    for (int b = a; b > 0 && descs[order[b]].K > descs[order[b - 1]].K; --b) { const int t = order[b]; order[b] = order[b - 1]; order[b - 1] = t; }
  int n_long = 0;
  for (int q = 0; q < count; ++q) {
    const sam_gemm_desc* d = descs + order[q];
    if (d->K % BK != 0 || d->M % 8 != 0 || d->N % 8 != 0 ) return SAM_ERR_UNSUPPORTED;
    if ((int64_t)d->K * d->lda * 2 >= (int64_t)0x7fffffff || (int64_t)d->K * d->ldb * 2 >= (int64_t)0x7fffffff) return SAM_ERR_UNSUPPORTED;
    WProb& p = w.p[q];
    p.A = (const bf16_t*)d->A; p.lda = d->lda; p.B = (const bf16_t*)d->B; p.ldb = d->ldb; p.C = (float*)d->C; p.ldc = d->ldc; p.bias_grad = d->bias_grad;
    p.M = d->M; p.N = d->N; p.K = d->K; p.accumulate = d->accumulate ? 1 : 0;
    p.tiles_m = (d->M + BM - 1) / BM; p.tiles_n = (d->N + BN - 1) / BN;
    w.tile_start[q] = tiles;
    tiles += p.tiles_m * p.tiles_n;
    if (d->K == max_k) n_long = tiles;
    min_kt = d->K / BK < min_kt ? d->K / BK : min_kt;
  }
  w.tile_start[count] = tiles;
  const int n_cu = grid_cu_count();
  // the pair exchange spins on the partner, so both blocks of a tile must become resident: one launch round (tiles * 2 <= CUs, one block per
  // CU).  Partners have adjacent item ids (block ids b, b + 8), and workgroups are dispatched in id order: should other work hold some CUs,
  // the resident set is still a prefix of the ids, every pair inside it completes and frees its CUs -- the wait cannot deadlock.
  w.split = (tiles * 2 <= n_cu && min_kt >= 8) ? 2 : 1;
  if (w.split == 1 && (tiles < n_cu / 2 || tiles > 4 * n_cu)) return SAM_ERR_UNSUPPORTED;      // too few tiles to fill the chip unsplit / a many-round grid: the 4-wave kernel
  w.n_long = (w.split == 1 && n_long < tiles && n_long <= n_cu && n_long % 8 == 0 && max_k >= 4 * min_kt * BK) ? n_long : 0;
  if (w.split == 2) {
    const sam_gemm_desc* d0 = descs;
    if (!d0->ws || d0->ws_bytes < gemm8w_ws_bytes(tiles) || ((uintptr_t)d0->ws % 16) != 0) return SAM_ERR_UNSUPPORTED;
    w.flags = reinterpret_cast<unsigned*>(d0->ws) + 64;                           // [tiles * 2] words, zero between launches (the caller zero-fills once);
    w.ws = d0->ws + 64 + ((tiles * 2 + 63) / 64) * 64;                            // word 0 of the workspace: raised by a block whose partner never arrived
  }
  { static int dbg = -1; if (dbg < 0) { const char* e = getenv("SAM_GEMM8W_DBG"); dbg = e ? atoi(e) : 0; } w.dbg = dbg; }
  constexpr size_t LDS = (size_t)2 * STAGE;
  static bool once = false;
  if (!once) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(gemm8w_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS);
    once = true;
  }
  gemm8w_kernel<<<dim3(tiles * w.split), dim3(512), LDS, st>>>(w);
  SAM_LAUNCH_CHECK();
  return SAM_OK;
}

}  // namespace samgemm
