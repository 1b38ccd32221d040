// bf16 MFMA GEMM, 8-wave persistent kernels (gfx950): the large nn.Linear sites of the SA-M4C step (forward and dgrad of the encoder layers).
//
// Block tile BM x BN x 64 (256x256 or 192x192), 512 threads = 2 (M) x 4 (N) waves of (BM/2) x (BN/4), ONE block per CU (grid = min(#CUs, #tiles)),
// each block walks its tiles b, b+G, b+2G, ... and treats their k-tiles as one stream:
//   * operands go global -> LDS by buffer_load ... lds (1 KB per wave instruction, no VGPR staging, no ds_write pass) into two stages;
//     the DMA queue is never drained inside the stream: one counted s_waitcnt vmcnt per k-tile, raw s_barrier (no __syncthreads);
//   * a k-tile is consumed in two phases (upper / lower half of the wave's rows); phase 0 issues the A tile of k-tile u+1, phase 1 the B tile
//     of k-tile u+2 -- across tile boundaries, so the next tile's first operands land while this tile's epilogue runs (with 12 k-tiles per
//     tile at K = 768 the pipeline fill + epilogue of a non-persistent block was a third of its life);
//   * the two wave groups (rows [0,BM/2) and [BM/2,BM)) run one barrier apart: on every SIMD one wave issues MFMAs while its partner reads
//     fragments and queues DMA (+15 % measured against the lock-step version of the same loop).
// LDS image of an operand tile: 128-byte rows, 8 rows per DMA slice.  k-contiguous operand: row = m (or n), 16-byte chunk c at c ^ ((row>>1)&7),
// fragments by ds_read_b128.  k-strided operand (dgrad's W, both operands of wgrad): 64-column panels, row = k inside a panel, chunk c at
// c ^ (sigma(k)<<1) with sigma(k) = bit1(k) | bit3(k)<<1, fragments by two ds_read_b64_tr_b16.  The swizzle is applied through each lane's
// SOURCE address (the DMA writes lane-linear).  Both images are bank-conflict free.
// Measured stand-alone (tools/probes/gemm8_probe.hip, MI355X): 1.36 / 1.51 PFLOP/s at 4096^3 / 8192^3 (256x256), 1.09 PFLOP/s (192x192).
#include "gemm8_dev.h"
#include <stdlib.h>

using namespace samgemm;
using namespace samgemm8;
namespace {

// tile j of this block -> (m0, n0).  Tile ids are dealt round-robin over blocks (id = block + j * grid, grid a multiple of 8 or < 8 ids apart never
// matter), block b runs on XCD b % 8: every XCD (own 4 MB L2) gets a contiguous run of ids, walked in GROUP_M x tiles_n super-columns.
template <int BM, int BN>
__device__ __forceinline__ void tile_origin(const GemmArgs& p, int id, int& m0, int& n0) {
  const int nblk = p.tiles_m * p.tiles_n;
  const int q = nblk / 8, r = nblk % 8, xcd = id % 8, loc = id / 8;
  const int bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
  const int per_group = p.group_m * p.tiles_n;
  const int group = bid / per_group, first_m = group * p.group_m;
  const int gsize = min(p.tiles_m - first_m, p.group_m);
  const int in_group = bid - group * per_group;
  m0 = (first_m + in_group % gsize) * BM;
  n0 = (in_group / gsize) * BN;
}

template <int BM, int BN, bool AKC, bool BKC, int EPI, typename OutT, int NST = 2>
__global__ __launch_bounds__(512, (BM * BN <= 128 * 128 ? 4 : 2)) void gemm8_kernel(GemmArgs p) {
  constexpr int TM = BM / 32, TN = BN / 64, SA = BM / 64, SB = BN / 64, RB = TM / 2;
  constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, STAGE = A_BYTES + B_BYTES;
  // LDS stages.  Two everywhere but the 192 x 192 tile, whose 48 KB stages fit THREE times: A is then fetched two k-tiles ahead and B three (one / two with two
  // stages).  The issue points do not move (phase 0 / phase 1 read segments); what changes is the distance between a DMA's issue and the wait that needs it:
  // with COLD operands (a training step never finds an activation in the Infinity Cache) one k-tile -- ~1.1 us -- is less than the HBM round trip under load, and
  // the long-K N = 768 products (FFN2 forward, FFN1 / QKV dgrad: A is 72 / 54 MB streamed once, 128 bytes per row and k-tile) stalled ~10 us of their 60 in
  // that wait (SAM_GEMM8_DBG = 1 / 9 / 5: loop only 61.1 us, with every DMA re-reading an L2-resident k-tile 49.7, without DMA 40.4).
  static_assert(NST == 2 || NST == 3, "two or three LDS stages");
  static_assert(BM % 64 == 0 && BN % 64 == 0 && SB <= 4, "tile shape");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), i = lane & 15, g = lane >> 4;
  const int wr = wave >> 2, wc = wave & 3;        // waves w and w+4 share a SIMD: one from each row group
  const int G = gridDim.x, nblk = p.tiles_m * p.tiles_n;
  const int my_tiles = (nblk - (int)blockIdx.x + G - 1) / G;
  const int KT = p.K / BK;
  const int total = my_tiles * KT;
  const unsigned kstepA = AKC ? BK * 2 : (unsigned)(BK * p.lda * 2), kstepB = BKC ? BK * 2 : (unsigned)(BK * p.ldb * 2);

  // DMA cursors: ua / ub = stream index of the next A / B tile to fetch; (ja, ka) / (jb, kb) = its tile and k-tile
  unsigned offA[SA], offB[SB];
  int m0, n0, ma, na_, mb_, nb;
  tile_origin<BM, BN>(p, blockIdx.x, m0, n0);
  src_offsets<AKC, SA>(offA, p.lda, m0, p.M, wave, lane);
  src_offsets<BKC, SB>(offB, p.ldb, n0, p.N, wave, lane);
  int ua = 0, ka = 0, ja = 0, ub = 0, kb = 0, jb = 0, sa_ = 0, sb_ = 0;       // sa_ / sb_: LDS stage of the next A / B fetch (= ua % NST, ub % NST)
  (void)ma; (void)na_; (void)mb_; (void)nb;
#define SAM_DMA_A()                                                                                                     \
  do {                                                                                                                  \
    dma_slices<SA>(p.A, smem + sa_ * STAGE + wave * (SA * 1024), offA, (p.dbg & 8) ? 0u : ka * kstepA);            \
    ++ua; sa_ = sa_ + 1 == NST ? 0 : sa_ + 1;                                                                           \
    if (++ka == KT) {                                                                                                   \
      ka = 0; ++ja;                                                                                                     \
      if (ja < my_tiles && !(p.dbg & 8)) { tile_origin<BM, BN>(p, blockIdx.x + ja * G, ma, na_); src_offsets<AKC, SA>(offA, p.lda, ma, p.M, wave, lane); } \
    }                                                                                                                   \
  } while (0)
#define SAM_DMA_B()                                                                                                     \
  do {                                                                                                                  \
    dma_slices<SB>(p.B, smem + sb_ * STAGE + A_BYTES + wave * (SB * 1024), offB, (p.dbg & 8) ? 0u : kb * kstepB);  \
    ++ub; sb_ = sb_ + 1 == NST ? 0 : sb_ + 1;                                                                           \
    if (++kb == KT) {                                                                                                   \
      kb = 0; ++jb;                                                                                                     \
      if (jb < my_tiles && !(p.dbg & 8)) { tile_origin<BM, BN>(p, blockIdx.x + jb * G, mb_, nb); src_offsets<BKC, SB>(offB, p.ldb, nb, p.N, wave, lane); } \
    }                                                                                                                   \
  } while (0)
  // (dbg bit 3, tuning: every DMA re-reads k-tile 0 of the block's first tile -- L2-resident after the first touch: the issue + LDS-write cost without the HBM wait)

  f32x4 acc[TN][TM];
#pragma unroll
  for (int a = 0; a < TN; ++a)
#pragma unroll
    for (int b = 0; b < TM; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

  // prologue: k-tile 0 complete; in flight behind it B(1) [two stages] or A(1), B(1), B(2) [three]
  SAM_DMA_A(); SAM_DMA_B();
  if constexpr (NST == 2) {
    if (total > 1) { SAM_DMA_B(); vmwait<SB>(); }
    else vmwait<0>();
  } else {
    if (total > 2) { SAM_DMA_A(); SAM_DMA_B(); SAM_DMA_B(); vmwait<SA + 2 * SB>(); }
    else if (total > 1) { SAM_DMA_A(); SAM_DMA_B(); vmwait<SA + SB>(); }
    else vmwait<0>();
  }
  __builtin_amdgcn_s_barrier();
  if (wr == 1) __builtin_amdgcn_s_barrier();      // lower row group: one barrier behind from here on

  const int sig = ((i >> 3) & 1) | ((g & 1) << 1);     // sigma(krow) for krow = 32 ks + 8 g + (i >> 2)
  bf16x8 af[RB][2], bfr[TN][2];
  int kt = 0, j = 0;
  int su = 0;                                        // stage of k-tile u
  for (int u = 0; u < total; ++u) {
    const unsigned char* stA = smem + su * STAGE;
    const unsigned char* stB = stA + A_BYTES;
    su = su + 1 == NST ? 0 : su + 1;
    // ================= phase 0: all B fragments + upper A rows; DMA of A(u+1) [A(u+2) with three stages] (its stage was last read in phase 1 of k-tile u-1)
#pragma unroll
    for (int x = 0; x < TN; ++x)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) bfr[x][ks] = frag<BKC>(stB, wc * (BN / 4) + x * 16, ks, i, g, sig);
#pragma unroll
    for (int x = 0; x < RB; ++x)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) af[x][ks] = frag<AKC>(stA, wr * (BM / 2) + x * 16, ks, i, g, sig);
    if (ua < total && !(p.dbg & 4)) SAM_DMA_A();           // (dbg bit 2, tuning: no operand DMA inside the loop -- stale operands, wrong results: what does the DMA issue cost the loop?)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // before the barrier: when the partner group passes it, this stage's B region may be refilled
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int x = 0; x < RB; ++x)
#pragma unroll
        for (int y = 0; y < TN; ++y) acc[y][x] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[y][ks], af[x][ks], acc[y][x], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    // ================= phase 1: lower A rows; DMA of B(u+2) into THIS stage (its B region was last read in phase 0)
#pragma unroll
    for (int x = 0; x < RB; ++x)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) af[x][ks] = frag<AKC>(stA, wr * (BM / 2) + (RB + x) * 16, ks, i, g, sig);
    if constexpr (NST == 2) {
      if (ub < total && !(p.dbg & 4)) { SAM_DMA_B(); vmwait<SB>(); }          // k-tile u+1 has landed (loads retire in order); B(u+2) stays in flight
      else vmwait<0>();
    } else {
      // three stages: B(u+3) goes out; A(u+1) and everything older must have landed, i.e. what may stay in flight is exactly what was issued after A(u+1):
      // B(u+2), A(u+2), B(u+3).  A count that is too large is a race, so the last three k-tiles of the block's stream simply drain.
      if (ub < total && !(p.dbg & 4)) { SAM_DMA_B(); vmwait<SA + 2 * SB>(); }
      else vmwait<0>();
    }
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
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    // ================= end of a tile: epilogue (the next tile's first operands are already in flight), fresh accumulators
    if (++kt == KT) {
      // The upper row group waits one barrier here, so both groups run their epilogues SIDE BY SIDE (left staggered, each epilogue would only
      // be covered by one 18-MFMA phase of the partner: two serial epilogues per tile); the lower group drops back behind afterwards.
      if (wr == 0) __builtin_amdgcn_s_barrier();
      const bool full = m0 + BM <= p.M && n0 + BN <= p.N;
      if (p.dbg & 1) {        // experiment: how long does the tile stream take without any epilogue?  (one dummy store keeps the accumulators live)
        float sacc = 0.f;
#pragma unroll
        for (int a = 0; a < TN; ++a)
#pragma unroll
          for (int b = 0; b < TM; ++b) sacc += acc[a][b][0] + acc[a][b][1] + acc[a][b][2] + acc[a][b][3];
        if (sacc == 12345.678f) reinterpret_cast<bf16_t*>(p.C)[tid] = (bf16_t)1;
      } else if constexpr (TM * TN > 18) {     // 32 fragments per wave: two halves, so that the batched operand prefetch of the epilogue fits the register file
        const int me = p.dbg == 2 ? 0 : m0;     // (dbg 2: every tile's epilogue lands on the first tile row -- the outputs stay in the L2, no HBM traffic)
        gemm_epilogue8<TM, TN, EPI, OutT, 0, RB>(p, acc, me + wr * (BM / 2), n0 + wc * (BN / 4), full, p.C, p.ldc, p.accumulate, i, g);
        gemm_epilogue8<TM, TN, EPI, OutT, RB, TM>(p, acc, me + wr * (BM / 2), n0 + wc * (BN / 4), full, p.C, p.ldc, p.accumulate, i, g);
      } else {
        const int me = p.dbg == 2 ? 0 : m0;
        gemm_epilogue8<TM, TN, EPI, OutT>(p, acc, me + wr * (BM / 2), n0 + wc * (BN / 4), full, p.C, p.ldc, p.accumulate, i, g);
      }
      kt = 0;
      if (++j < my_tiles) {
        if (wr == 1) __builtin_amdgcn_s_barrier();
        tile_origin<BM, BN>(p, blockIdx.x + j * G, m0, n0);
#pragma unroll
        for (int a = 0; a < TN; ++a)
#pragma unroll
          for (int b = 0; b < TM; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
    }
  }
#undef SAM_DMA_A
#undef SAM_DMA_B
}

// (Measured and not kept -- a three-stage ring for the 192x192 tile (3 x 48 KB) that fetches two k-tiles ahead and issues the DMA slices between
// MFMA groups inside the MFMA segment, so that the read segment holds fragment reads only: 58.0 vs 50.5 us at 11648 x 768 x 3072, 69.9 vs 60.3 us
// at N = 3072 / K = 768.  Splitting the MFMA stream around the DMA issue costs more than the shorter read segment returns.)
// ---------------------------------------------------------------------------------------------------------------------------------
// Deferred epilogue (192x192 tiles).  With 12 k-tiles per tile (K = 768) an epilogue that runs between two tiles costs a third of the block's
// life -- the MFMA pipes idle while 36 K outputs per block are converted, activated and stored in one burst, and every CU bursts at the same
// time (measured: N = 3072, K = 768: 47 us without any epilogue, 61 us with bias, 81 us with bias + GELU + the pre-activation copy).
// Here a finished tile's accumulators are PARKED in a second register set (72 of the 116 registers the 192x192 configuration leaves free) and
// the epilogue is cut into 12 slices -- row fragment tm x {column-fragment pair 8-wide, single fragment 4-wide} -- one per phase of the NEXT
// tile's main loop, executed in the read segment, i.e. underneath the partner wave group's MFMAs.  Slice operands never travel through VGPRs
// ahead of time: the bias vector sits in LDS for the whole launch, the residual / pre-activation rows of the next slice are DMA-ed into a
// per-wave LDS slot one phase ahead (same queue and the same counted vmcnt as the operand tiles: nothing drains).
template <int TM, int TN, int EPI, int TMI, int PART>
__device__ __forceinline__ void defer_slice(const GemmArgs& p, const f32x4 (&pacc)[TN][TM], int prow0, int pc8, int pc4, bool pfull, const unsigned char* slot,
                                            const float* bias_lds, int lane, int g, unsigned seed_lo, unsigned seed_hi, unsigned off_lo, unsigned off_hi) {
  constexpr int W = PART == 0 ? 8 : 4;
  constexpr bool HAS_BIAS = EPI == SAM_EPI_BIAS || EPI == SAM_EPI_BIAS_GELU || EPI == SAM_EPI_BIAS_DROPOUT_RES || EPI == SAM_EPI_BIAS_GELU_GRAD;
  constexpr bool HAS_PRE = EPI == SAM_EPI_DGELU || EPI == SAM_EPI_BIAS_DROPOUT_RES || EPI == SAM_EPI_MUL_AUX;
  const int m = prow0 + 16 * TMI, n = PART == 0 ? pc8 : pc4;
  float v[W];
  if constexpr (PART == 0) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const auto sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(pacc[0][TMI][r]), __float_as_uint(pacc[1][TMI][r]), false, false);
      v[r] = __uint_as_float(sw[0]);
      v[4 + r] = __uint_as_float(sw[1]);
    }
  } else {
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = pacc[2][TMI][r];
  }
  const bool ok = pfull || (m < p.M && n < p.N);
  unsigned pre[W / 2];
  if (HAS_PRE) {
    if constexpr (PART == 0) {
      const uint4 x = *reinterpret_cast<const uint4*>(slot + lane * 16);
      pre[0] = x.x; pre[1] = x.y; pre[2] = x.z; pre[3] = x.w;
    } else {
      const uint2 x = *reinterpret_cast<const uint2*>(slot + lane * 16 + (g & 1) * 8);
      pre[0] = x.x; pre[1] = x.y;
    }
  }
  if (HAS_BIAS) {
#pragma unroll
    for (int q = 0; q < W / 4; ++q) {
      const float4 b = *reinterpret_cast<const float4*>(bias_lds + n + 4 * q);
      v[4 * q] += b.x; v[4 * q + 1] += b.y; v[4 * q + 2] += b.z; v[4 * q + 3] += b.w;
    }
  }
  if (EPI == SAM_EPI_BIAS_GELU) {
    if (ok) {
      bf16_t* dst = p.aux_out + (int64_t)m * p.ld_aux + n;
      if constexpr (PART == 0) *reinterpret_cast<uint4*>(dst) = make_uint4(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7]));
      else *reinterpret_cast<uint2*>(dst) = make_uint2(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]));
    }
#pragma unroll
    for (int r = 0; r < W; ++r) v[r] = gelu_erf(v[r]);
  }
  if (EPI == SAM_EPI_BIAS_GELU_GRAD) {
    float d[W];
#pragma unroll
    for (int r = 0; r < W; ++r) v[r] = gelu_erf_and_grad(v[r], d[r]);
    if (ok && p.aux_out) {                  // (aux_out NULL: inference -- the 71.6 MB derivative block of an MMT-size FFN1 is not written)
      bf16_t* dst = p.aux_out + (int64_t)m * p.ld_aux + n;
      if constexpr (PART == 0) *reinterpret_cast<uint4*>(dst) = make_uint4(pack_bf16x2(d[0], d[1]), pack_bf16x2(d[2], d[3]), pack_bf16x2(d[4], d[5]), pack_bf16x2(d[6], d[7]));
      else *reinterpret_cast<uint2*>(dst) = make_uint2(pack_bf16x2(d[0], d[1]), pack_bf16x2(d[2], d[3]));
    }
  }
  if (EPI == SAM_EPI_DGELU) {
#pragma unroll
    for (int r = 0; r < W / 2; ++r) { v[2 * r] *= gelu_erf_grad(bf_lo(pre[r])); v[2 * r + 1] *= gelu_erf_grad(bf_hi(pre[r])); }
  }
  if (EPI == SAM_EPI_MUL_AUX) {
#pragma unroll
    for (int r = 0; r < W / 2; ++r) { v[2 * r] *= bf_lo(pre[r]); v[2 * r + 1] *= bf_hi(pre[r]); }
  }
  if (EPI == SAM_EPI_BIAS_DROPOUT_RES) {
    if (p.thr16) {      // the (row, col/8) Philox stream of every other epilogue of the family
      const u32x4 rn = hidden_dropout_bits((unsigned)m, (unsigned)(n >> 3), off_lo, off_hi, seed_lo, seed_hi);
      const unsigned w4[4] = {rn.x, rn.y, rn.z, rn.w};
#pragma unroll
      for (int r = 0; r < W / 2; ++r) {
        const unsigned w = PART == 0 ? w4[r] : w4[r + ((n & 4) ? 2 : 0)];
        v[2 * r] = (w & 0xffffu) >= p.thr16 ? v[2 * r] * p.inv_keep : 0.f;
        v[2 * r + 1] = (w >> 16) >= p.thr16 ? v[2 * r + 1] * p.inv_keep : 0.f;
      }
    }
    if (p.residual) {
#pragma unroll
      for (int r = 0; r < W / 2; ++r) { v[2 * r] += bf_lo(pre[r]); v[2 * r + 1] += bf_hi(pre[r]); }
    }
  }
  if (ok) {
    bf16_t* dst = reinterpret_cast<bf16_t*>(p.C) + (int64_t)m * p.ldc + n;
    if constexpr (PART == 0) *reinterpret_cast<uint4*>(dst) = make_uint4(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7]));
    else *reinterpret_cast<uint2*>(dst) = make_uint2(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]));
  }
}

template <bool AKC, bool BKC, int EPI>
__global__ __launch_bounds__(512, 2) void gemm8d_kernel(GemmArgs p) {
  constexpr int BM = 192, BN = 192;
  constexpr int TM = BM / 32, TN = BN / 64, SA = BM / 64, SB = BN / 64, RB = TM / 2, NSL = 2 * TM;
  constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, STAGE = A_BYTES + B_BYTES;
  constexpr int BIAS_BYTES = 17408, SLOT_BYTES = 8192;         // bias: N <= 4096 columns + one tile of slack; two residual slots of 1 KB per wave
  constexpr bool HAS_PRE = EPI == SAM_EPI_DGELU || EPI == SAM_EPI_BIAS_DROPOUT_RES || EPI == SAM_EPI_MUL_AUX;
  static_assert(TN == 3 && SA == SB, "deferred epilogue is laid out for the 192x192 tile");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), i = lane & 15, g = lane >> 4;
  const int wr = wave >> 2, wc = wave & 3;
  const int G = gridDim.x, nblk = p.tiles_m * p.tiles_n;
  const int my_tiles = (nblk - (int)blockIdx.x + G - 1) / G;
  const int KT = p.K / BK;
  const int total = my_tiles * KT;
  const unsigned kstepA = AKC ? BK * 2 : (unsigned)(BK * p.lda * 2), kstepB = BKC ? BK * 2 : (unsigned)(BK * p.ldb * 2);
  float* bias_lds = reinterpret_cast<float*>(smem + 2 * STAGE);
  unsigned char* res_lds = smem + 2 * STAGE + BIAS_BYTES + wave * 1024;

  unsigned offA[SA], offB[SB];
  int m0, n0, ma, na_, mb_, nb;
  tile_origin<BM, BN>(p, blockIdx.x, m0, n0);
  src_offsets<AKC, SA>(offA, p.lda, m0, p.M, wave, lane);
  src_offsets<BKC, SB>(offB, p.ldb, n0, p.N, wave, lane);
  int ua = 0, ka = 0, ja = 0, ub = 0, kb = 0, jb = 0;
  (void)ma; (void)na_; (void)mb_; (void)nb;
#define SAM_DMA_A()                                                                                                     \
  do {                                                                                                                  \
    dma_slices<SA>(p.A, smem + (ua & 1) * STAGE + wave * (SA * 1024), offA, ka * kstepA);                               \
    ++ua;                                                                                                               \
    if (++ka == KT) {                                                                                                   \
      ka = 0; ++ja;                                                                                                     \
      if (ja < my_tiles) { tile_origin<BM, BN>(p, blockIdx.x + ja * G, ma, na_); src_offsets<AKC, SA>(offA, p.lda, ma, p.M, wave, lane); } \
    }                                                                                                                   \
  } while (0)
#define SAM_DMA_B()                                                                                                     \
  do {                                                                                                                  \
    dma_slices<SB>(p.B, smem + (ub & 1) * STAGE + A_BYTES + wave * (SB * 1024), offB, kb * kstepB);                     \
    ++ub;                                                                                                               \
    if (++kb == KT) {                                                                                                   \
      kb = 0; ++jb;                                                                                                     \
      if (jb < my_tiles) { tile_origin<BM, BN>(p, blockIdx.x + jb * G, mb_, nb); src_offsets<BKC, SB>(offB, p.ldb, nb, p.N, wave, lane); } \
    }                                                                                                                   \
  } while (0)

  // the bias vector (zeros when there is none, and behind column N) into LDS, once
  for (int c4 = tid; c4 * 4 < BIAS_BYTES / 4; c4 += 512) {
    float4 b = make_float4(0.f, 0.f, 0.f, 0.f);
    if (p.bias && c4 * 4 < p.N) b = *reinterpret_cast<const float4*>(p.bias + c4 * 4);
    reinterpret_cast<float4*>(bias_lds)[c4] = b;
  }
  unsigned seed_lo = p.seed_lo, seed_hi = p.seed_hi, off_lo = p.off_lo, off_hi = p.off_hi;
  if (EPI == SAM_EPI_BIAS_DROPOUT_RES) rng_resolve(p.rng_state, seed_lo, seed_hi, off_lo, off_hi);

  f32x4 acc[TN][TM], pacc[TN][TM];
#pragma unroll
  for (int a = 0; a < TN; ++a)
#pragma unroll
    for (int b = 0; b < TM; ++b) { acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f}; pacc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f}; }
  int pslice = NSL, prow0 = 0, pc8 = 0, pc4 = 0;       // parked tile: slices [pslice, NSL) still to do; this lane's first row / pair column / single column
  bool pfull = true;

  // residual / pre-activation piece of slice s of the parked tile: 16 bytes per lane into this wave's slot (s & 1)
  const bf16_t* pre_src = EPI == SAM_EPI_BIAS_DROPOUT_RES ? p.residual : p.aux_in;
  const int64_t pre_ld = EPI == SAM_EPI_BIAS_DROPOUT_RES ? p.ldr : p.ld_aux;
#define SAM_RES_DMA(s)                                                                                                                     \
  do {                                                                                                                                     \
    if (HAS_PRE && pre_src) {                                                                                                              \
      const int row_ = min(prow0 + 16 * ((s) >> 1), p.M - 1);                                                                              \
      const int col_ = min(((s) & 1) ? (pc4 & ~7) : pc8, p.N - 8);                                                                         \
      const unsigned off_ = (unsigned)(((int64_t)row_ * pre_ld + col_) * 2);                                                               \
      dma_slices<1>(pre_src, res_lds + ((s) & 1) * SLOT_BYTES, &off_, 0u);                                                                 \
    }                                                                                                                                      \
  } while (0)
#define SAM_SLICE_CASE(S) \
  case S: defer_slice<TM, TN, EPI, ((S) >> 1), ((S) & 1)>(p, pacc, prow0, pc8, pc4, pfull, res_lds + ((S) & 1) * SLOT_BYTES, bias_lds, lane, g, seed_lo, seed_hi, off_lo, off_hi); break;
#define SAM_SLICE_CASE2(S) \
  case S: defer_slice<TM, TN, EPI, ((S) >> 1), 1>(p, pacc, prow0, pc8, pc4, pfull, res_lds + SLOT_BYTES, bias_lds, lane, g, seed_lo, seed_hi, off_lo, off_hi); break;
#define SAM_RUN_SLICE()                                                                                                                    \
  do {                                                                                                                                     \
    switch (pslice) {                                                                                                                      \
      SAM_SLICE_CASE(0) SAM_SLICE_CASE(1) SAM_SLICE_CASE(2) SAM_SLICE_CASE(3) SAM_SLICE_CASE(4) SAM_SLICE_CASE(5)                          \
      SAM_SLICE_CASE(6) SAM_SLICE_CASE(7) SAM_SLICE_CASE(8) SAM_SLICE_CASE(9) SAM_SLICE_CASE(10) SAM_SLICE_CASE(11)                        \
      default: break;                                                                                                                      \
    }                                                                                                                                      \
    ++pslice;                                                                                                                              \
    if (pslice < NSL) SAM_RES_DMA(pslice);                                                                                                 \
  } while (0)

#define SAM_RUN_PAIR()                                                                                                                     \
  do {                                                                                                                                     \
    switch (pslice) {                                                                                                                      \
      SAM_SLICE_CASE(0) SAM_SLICE_CASE(2) SAM_SLICE_CASE(4) SAM_SLICE_CASE(6) SAM_SLICE_CASE(8) SAM_SLICE_CASE(10)                         \
      default: break;                                                                                                                      \
    }                                                                                                                                      \
    switch (pslice) {                                                                                                                      \
      SAM_SLICE_CASE2(0) SAM_SLICE_CASE2(2) SAM_SLICE_CASE2(4) SAM_SLICE_CASE2(6) SAM_SLICE_CASE2(8) SAM_SLICE_CASE2(10)                   \
      default: break;                                                                                                                      \
    }                                                                                                                                      \
    pslice += 2;                                                                                                                           \
    if (pslice < NSL) { SAM_RES_DMA(pslice); SAM_RES_DMA(pslice + 1); }                                                                    \
  } while (0)

  SAM_DMA_A(); SAM_DMA_B();
  if (total > 1) { SAM_DMA_B(); vmwait<SB>(); }
  else vmwait<0>();
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        // the bias image
  __builtin_amdgcn_s_barrier();
  if (wr == 1) __builtin_amdgcn_s_barrier();

  const int sig = ((i >> 3) & 1) | ((g & 1) << 1);
  bf16x8 af[RB][2], bfr[TN][2];
  int kt = 0, j = 0;
  for (int u = 0; u < total; ++u) {
    const unsigned char* stA = smem + (u & 1) * STAGE;
    const unsigned char* stB = stA + A_BYTES;
    // ================= phase 0
#pragma unroll
    for (int x = 0; x < TN; ++x)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) bfr[x][ks] = frag<BKC>(stB, wc * (BN / 4) + x * 16, ks, i, g, sig);
#pragma unroll
    for (int x = 0; x < RB; ++x)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) af[x][ks] = frag<AKC>(stA, wr * (BM / 2) + x * 16, ks, i, g, sig);
    if (ua < total) SAM_DMA_A();
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
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    // ================= phase 1
#pragma unroll
    for (int x = 0; x < RB; ++x)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) af[x][ks] = frag<AKC>(stA, wr * (BM / 2) + (RB + x) * 16, ks, i, g, sig);
    if (ub < total) { SAM_DMA_B(); vmwait<SB>(); }          // k-tile u+1 has landed -- and so have the residual pieces queued during k-tile u-1
    else vmwait<0>();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    if (pslice < NSL) SAM_RUN_PAIR();                       // two slices of the parked tile's epilogue, underneath the partner group's MFMAs
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int x = 0; x < RB; ++x)
#pragma unroll
        for (int y = 0; y < TN; ++y) acc[y][RB + x] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[y][ks], af[x][ks], acc[y][RB + x], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    // ================= end of a tile: park the accumulators
    if (++kt == KT) {
      while (pslice < NSL) { vmwait<0>(); SAM_RUN_PAIR(); }        // (only when a tile has fewer than 6 k-tiles: the previous one is not finished yet)
#pragma unroll
      for (int a = 0; a < TN; ++a)
#pragma unroll
        for (int b = 0; b < TM; ++b) { pacc[a][b] = acc[a][b]; acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f}; }
      pfull = m0 + BM <= p.M && n0 + BN <= p.N;
      prow0 = m0 + wr * (BM / 2) + i;
      pc8 = n0 + wc * (BN / 4) + 16 * (g & 1) + 8 * (g >> 1);
      pc4 = n0 + wc * (BN / 4) + 32 + 4 * g;
      pslice = p.dbg == 1 ? NSL : 0;
      if (pslice < NSL) { SAM_RES_DMA(0); SAM_RES_DMA(1); }
      kt = 0;
      if (++j < my_tiles) tile_origin<BM, BN>(p, blockIdx.x + j * G, m0, n0);
    }
  }
  while (pslice < NSL) { vmwait<0>(); SAM_RUN_PAIR(); }
  if (wr == 0) __builtin_amdgcn_s_barrier();
#undef SAM_DMA_A
#undef SAM_DMA_B
#undef SAM_RES_DMA
#undef SAM_SLICE_CASE
#undef SAM_RUN_SLICE
#undef SAM_RUN_PAIR
#undef SAM_SLICE_CASE2
}

template <bool AKC, bool BKC, int EPI>
int launch8d(GemmArgs a, int n_cu, hipStream_t st) {
  constexpr size_t LDS = (size_t)2 * (192 + 192) * 128 + 17408 + 2 * 8192;
  static bool once = false;
  if (!once) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(gemm8d_kernel<AKC, BKC, EPI>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS);
    once = true;
  }
  a.tiles_m = (a.M + 191) / 192; a.tiles_n = (a.N + 191) / 192;
  const int tiles = a.tiles_m * a.tiles_n;
  gemm8d_kernel<AKC, BKC, EPI><<<dim3(tiles < n_cu ? tiles : n_cu), dim3(512), LDS, st>>>(a);
  SAM_LAUNCH_CHECK();
  return SAM_OK;
}

template <int BM, int BN, bool AKC, bool BKC, int EPI, typename OutT, int NST = 2>
int launch8(GemmArgs a, int n_cu, hipStream_t st) {
  constexpr size_t LDS = (size_t)NST * (BM + BN) * 128;
  static_assert(LDS <= 160 * 1024, "LDS stages");
  static bool once = false;
  if (!once) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(gemm8_kernel<BM, BN, AKC, BKC, EPI, OutT, NST>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS);
    once = true;
  }
  a.tiles_m = (a.M + BM - 1) / BM; a.tiles_n = (a.N + BN - 1) / BN;
  const int tiles = a.tiles_m * a.tiles_n;
  const int slots = BM * BN <= 128 * 128 ? 2 * n_cu : n_cu;      // the 128 x 128 configuration (64 KB of stages, <= 128 VGPRs) runs two blocks per CU
  gemm8_kernel<BM, BN, AKC, BKC, EPI, OutT, NST><<<dim3(tiles < slots ? tiles : slots), dim3(512), LDS, st>>>(a);
  SAM_LAUNCH_CHECK();
  return SAM_OK;
}

// Tile choice: a makespan model, calibrated with COLD operands (tools/bench_gemm_cold.py: every call on another buffer set, as in a training step,
// where no activation is still in the Infinity Cache when it is needed; the cache-resident numbers of tools/bench_gemm8.py had put 192x192 ahead
// on the wide outputs, which the step did not confirm).  A block runs `rounds` tiles back to back, each KT k-tiles of tk us plus an epilogue that
// nothing overlaps (one block per CU) and that moves `streams` tile-sized bf16 operands (output, auxiliary output, residual / auxiliary
// input) at es us each: time = rounds * (KT * tk + streams * es).  Measured: tk = 1.43 / 1.14 us (256^2 / 192^2), es = 6.0 / 4.7 us, and 7.2 us
// per stream for the 192^2 tile when it writes two outputs (its 96-byte row segments per wave share 128-byte lines between waves).  The model is
// within 6 % of the measurements on all seven encoder-layer shapes and picks the faster tile on each: 256^2 for N = 2304, 192 x 256 for N = 3072
// (732 tiles = 2.86 rounds of 0.75-size tiles instead of 2.16 -> 3 rounds of full-size ones; 64 output columns per wave = whole 128-byte lines:
// tk 1.27, es 4.9 / 5.9; FFN1 forward 113 -> 88 -> 81 us), 192^2 for N = 768 (138 tiles of 256^2 would leave half the chip idle).
struct TileCfg { int bm, bn; float tk, es, es2; };
constexpr TileCfg kCfg[4] = {{256, 256, 1.43f, 6.0f, 6.0f}, {192, 192, 1.14f, 4.7f, 7.2f}, {192, 256, 1.27f, 4.9f, 5.9f}, {128, 128, 0.f, 0.f, 0.f}};

template <bool AKC, bool BKC, int EPI, typename OutT>
int pick8(const GemmArgs& a, int tile, hipStream_t st) {
  const int n_cu = grid_cu_count();
  int best = -1; float best_score = 0.f;
  for (int c = 0; c < 4; ++c) {
    // 1192 / 1256 / 1448 / 1128: force 192x192 / 256x256 / 192x256 / 128x128; 3192: 192x192 with the deferred epilogue (measured slower)
    if (tile != 0 && (tile == 1448 ? c != 2 : (c == 2 || tile % 1000 != kCfg[c].bm))) continue;
    if (c == 3 && tile == 0) continue;          // 128x128: forced only (see pick_small below)
    const int tiles = ((a.M + kCfg[c].bm - 1) / kCfg[c].bm) * ((a.N + kCfg[c].bn - 1) / kCfg[c].bn);
    const int rounds = (tiles + n_cu - 1) / n_cu;
    const bool two_out = EPI == SAM_EPI_BIAS_GELU_GRAD || EPI == SAM_EPI_BIAS_GELU;
    const int streams = 1 + (two_out ? 1 : 0) + ((EPI == SAM_EPI_BIAS_DROPOUT_RES && a.residual) || EPI == SAM_EPI_MUL_AUX || EPI == SAM_EPI_DGELU ? 1 : 0);
    const float t = (float)rounds * ((float)(a.K / BK) * kCfg[c].tk + (float)streams * (two_out ? kCfg[c].es2 : kCfg[c].es));
    const float score = 1.0f / t;
    if (best < 0 || score > best_score) { best = c; best_score = score; }
  }
  if (best == 3) return launch8<128, 128, AKC, BKC, EPI, OutT>(a, n_cu, st);
  if (best == 0) return launch8<256, 256, AKC, BKC, EPI, OutT>(a, n_cu, st);
  if (best == 2) return launch8<192, 256, AKC, BKC, EPI, OutT>(a, n_cu, st);
  if (best == 1) {
    // 192x192: the deferred-epilogue kernel whenever its LDS bias image fits and it has something to do per tile
    static int defer = -1;
    if (defer < 0) { const char* v = getenv("SAM_GEMM8_DEFER"); defer = v ? atoi(v) : 0; }      // measured slower (see gemm8d_kernel): opt-in
    if constexpr (std::is_same<OutT, bf16_t>::value) {
      if ((defer || tile == 3192) && tile != 2192 && a.N <= 4096 && a.M * (int64_t)(EPI == SAM_EPI_BIAS_DROPOUT_RES ? a.ldr : a.ld_aux) * 2 < (int64_t)0x7fffffff) return launch8d<AKC, BKC, EPI>(a, n_cu, st);
    }
    static int stages = -1;
    if (stages < 0) { const char* v = getenv("SAM_GEMM8_STAGES"); stages = v ? atoi(v) : 3; }      // (2: the two-stage ring of rounds 2-4, for an A/B)
    if (stages == 2) return launch8<192, 192, AKC, BKC, EPI, OutT, 2>(a, n_cu, st);
    return launch8<192, 192, AKC, BKC, EPI, OutT, 3>(a, n_cu, st);
  }
  return SAM_ERR_UNSUPPORTED;
}

}  // namespace

int samgemm::gemm8_launch(const GemmArgs& a_in, int lay, int e, int c_is_f32, int tile, hipStream_t st) {
  GemmArgs a = a_in;
  { static int dbg = -1; if (dbg < 0) { const char* v = getenv("SAM_GEMM8_DBG"); dbg = v ? atoi(v) : 0; } a.dbg = dbg; }
  // the DMA addresses are 32-bit byte offsets from the operand base; k-tiles are whole; nothing here splits K or reduces a bias gradient
  if (a.K % BK != 0 || a.split_k > 1 || a.bias_grad != nullptr || c_is_f32) return SAM_ERR_UNSUPPORTED;
  const int64_t a_rows = (lay & 2) ? a.M : a.K, b_rows = (lay & 1) ? a.N : a.K;
  if (a_rows * a.lda * 2 >= (int64_t)0x7fffffff || b_rows * a.ldb * 2 >= (int64_t)0x7fffffff) return SAM_ERR_UNSUPPORTED;
  if (tile == 0 && ((int64_t)((a.M + 191) / 192) * ((a.N + 191) / 192) < 160 || a.K < 256)) {
    // small grids: the 4-wave kernels (2-4 blocks per CU) -- except a short-K problem whose 128 x 128 tiles fill the chip once (TextBert's FFN1 forward and
    // FFN2 dgrad, 1280 x 3072 x 768: 240 tiles, two blocks per CU; cold operands 15.4 / 14.1 us against 18.2 / 17.6 for the 64 x 64 four-wave tiles.  At MMT
    // size the configuration loses to the larger tiles on every shape -- 56 / 91 / 91 us for QKV / FFN1 / FFN2 against 49 / 82 / 69: twice the LDS traffic per flop)
    const int64_t t128 = (int64_t)((a.M + 127) / 128) * ((a.N + 127) / 128);
    if (a.K >= 256 && a.K <= 1024 && t128 >= 200 && t128 <= 2 * grid_cu_count()) tile = 1128;
    else return SAM_ERR_UNSUPPORTED;
  }
  if (lay == 3) {
    if (e == SAM_EPI_NONE) return pick8<true, true, SAM_EPI_NONE, bf16_t>(a, tile, st);
    if (e == SAM_EPI_BIAS) return pick8<true, true, SAM_EPI_BIAS, bf16_t>(a, tile, st);
    if (e == SAM_EPI_BIAS_GELU_GRAD) return pick8<true, true, SAM_EPI_BIAS_GELU_GRAD, bf16_t>(a, tile, st);      // (plain BIAS_GELU / DGELU: inference, old callers -> 4-wave kernels)
    if (e == SAM_EPI_BIAS_DROPOUT_RES) return pick8<true, true, SAM_EPI_BIAS_DROPOUT_RES, bf16_t>(a, tile, st);
  } else if (lay == 2) {
    if (e == SAM_EPI_NONE) return pick8<true, false, SAM_EPI_NONE, bf16_t>(a, tile, st);
    if (e == SAM_EPI_MUL_AUX) return pick8<true, false, SAM_EPI_MUL_AUX, bf16_t>(a, tile, st);
    if (e == SAM_EPI_BIAS_DROPOUT_RES) return pick8<true, false, SAM_EPI_BIAS_DROPOUT_RES, bf16_t>(a, tile, st);
  }
  return SAM_ERR_UNSUPPORTED;
}
