// bf16 MFMA GEMM with LOADER WAVES (gfx950): the 8-wave persistent kernel of gemm8.hip with the operand DMA moved to four waves of their own.
//
// Why (profiles/r5_gemm_experiments.txt): in gemm8_kernel every compute wave also issues its share of the operand DMA (buffer_load ... lds, 3-4 pieces of 1 KB per
// phase at 60-180 cycles of issue each) inside its fragment-read segment and waits for it with a counted s_waitcnt vmcnt -- the read segment, not the MFMA segment of
// the partner wave, sets the pace of the loop (FFN1 forward: 48 us loop, 39 us without the DMA), and because loads and stores retire in order on one counter, the
// stores of a tile's epilogue stand in front of the next tile's operand waits: the drain of 143 MB of stores is serial with the loop (82 us complete).
// Here a workgroup is 12 waves, three per SIMD: waves 0-7 compute exactly as in gemm8_kernel (2 x 4 waves of (BM/2) x (BN/4), the two row groups one barrier
// apart) but never touch the operand queue -- no DMA, no vmcnt in the loop, their epilogue stores drain underneath the next tile's k-loop -- and waves 8-11, one
// per SIMD, do nothing but issue the DMA pieces (a quarter of every operand tile each, spread over the four barrier intervals of a k-tile) and wait for them.
// The price is the register file: three waves per SIMD leave 168 registers per wave, i.e. the 192-row tiles (96 / 72 accumulators), not 256 x 256.
// The LDS images, the fragment reads, the epilogues and the tile order are gemm8's (gemm8_dev.h, gemm_common.h).
#include "gemm8_dev.h"
#include <stdlib.h>

using namespace samgemm;
using namespace samgemm8;
namespace {

template <int BM, int BN>
__device__ __forceinline__ void tile_origin12(const GemmArgs& p, int id, int& m0, int& n0) {
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

// pieces [S0, S1) of a loader's share
template <int S0, int S1>
__device__ __forceinline__ void dma_range(const bf16_t* base, unsigned char* dst, const unsigned* off, unsigned soff) {
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, 0x7fffffff, 0x00020000);
#pragma unroll
  for (int s = S0; s < S1; ++s)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(dst + s * 1024), 16, off[s], soff, 0, 0);
}

// PH = read / MFMA phases per k-tile.  2: gemm8's schedule (phase 0: all B fragments + the upper A rows, phase 1: the lower A rows).  1 (192 x 192, three stages
// only): ALL fragments of a k-tile in one read segment, all 36 MFMAs in one MFMA segment -- two barriers per k-tile instead of four.  With 18 MFMAs per phase
// (306 cycles) the fixed costs of a phase (barrier, fragment-read latency, lgkmcnt drain) were as long as the work they separate: 538 cycles per interval
// measured.  It needs both halves of the A fragments at once (48 + 24 fragment registers next to 72 accumulators: fits 168) and a third LDS stage (a stage is
// read by both row groups before the loader may refill it, i.e. one barrier interval later than with two phases).
template <int BM, int BN, bool AKC, bool BKC, int EPI, typename OutT, int NST, int PH = 2>
__global__ __launch_bounds__(768, 3) void gemm12_kernel(GemmArgs p) {
  constexpr int TM = BM / 32, TN = BN / 64, RB = TM / 2;
  constexpr int SAL = BM / 32, SBL = BN / 32;          // 1 KB pieces per loader wave and operand tile
  constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, STAGE = A_BYTES + B_BYTES;
  static_assert(NST == 2 || NST == 3, "two or three LDS stages");
  static_assert(PH == 2 || (PH == 1 && NST == 3), "one phase per k-tile needs three stages");
  static_assert(BM % 64 == 0 && BN % 64 == 0 && SAL % 2 == 0 && SBL % 2 == 0, "tile shape");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), i = lane & 15, g = lane >> 4;
  const int G = gridDim.x, nblk = p.tiles_m * p.tiles_n;
  const int my_tiles = (nblk - (int)blockIdx.x + G - 1) / G;
  const int KT = p.K / BK;
  const int total = my_tiles * KT;

  if (wave >= 8) {
    // =============================================================== loader waves: the barrier sequence of the UPPER compute row group
    const int lw = wave - 8;
    const unsigned kstepA = AKC ? BK * 2 : (unsigned)(BK * p.lda * 2), kstepB = BKC ? BK * 2 : (unsigned)(BK * p.ldb * 2);
    unsigned offA[SAL], offB[SBL];
    int m0, n0;
    tile_origin12<BM, BN>(p, blockIdx.x, m0, n0);
    src_offsets<AKC, SAL>(offA, p.lda, m0, p.M, lw, lane);
    src_offsets<BKC, SBL>(offB, p.ldb, n0, p.N, lw, lane);
    int ua = 0, ka = 0, ja = 0, ub = 0, kb = 0, jb = 0, sa_ = 0, sb_ = 0;
    // one operand tile goes out in two halves (H = 0, 1), a barrier interval apart; the cursor moves with the second half
#define SAM_LDMA_A(H)                                                                                                                 \
  do {                                                                                                                                \
    dma_range<(H) * (SAL / 2), ((H) + 1) * (SAL / 2)>(p.A, smem + sa_ * STAGE + lw * (SAL * 1024), offA, ka * kstepA);                \
    if (H) {                                                                                                                          \
      ++ua; sa_ = sa_ + 1 == NST ? 0 : sa_ + 1;                                                                                       \
      if (++ka == KT) {                                                                                                               \
        ka = 0; ++ja;                                                                                                                 \
        if (ja < my_tiles) { tile_origin12<BM, BN>(p, blockIdx.x + ja * G, m0, n0); src_offsets<AKC, SAL>(offA, p.lda, m0, p.M, lw, lane); } \
      }                                                                                                                               \
    }                                                                                                                                 \
  } while (0)
#define SAM_LDMA_B(H)                                                                                                                 \
  do {                                                                                                                                \
    dma_range<(H) * (SBL / 2), ((H) + 1) * (SBL / 2)>(p.B, smem + sb_ * STAGE + A_BYTES + lw * (SBL * 1024), offB, kb * kstepB);      \
    if (H) {                                                                                                                          \
      ++ub; sb_ = sb_ + 1 == NST ? 0 : sb_ + 1;                                                                                       \
      if (++kb == KT) {                                                                                                               \
        kb = 0; ++jb;                                                                                                                 \
        if (jb < my_tiles) { tile_origin12<BM, BN>(p, blockIdx.x + jb * G, m0, n0); src_offsets<BKC, SBL>(offB, p.ldb, n0, p.N, lw, lane); } \
      }                                                                                                                               \
    }                                                                                                                                 \
  } while (0)
    // prologue: k-tile 0 complete; behind it B(1) [two stages] or A(1), B(1), B(2) [three] or k-tile 1 [one phase per k-tile]
    SAM_LDMA_A(0); SAM_LDMA_A(1); SAM_LDMA_B(0); SAM_LDMA_B(1);
    if constexpr (PH == 1) {
      if (total > 1) { SAM_LDMA_A(0); SAM_LDMA_A(1); SAM_LDMA_B(0); SAM_LDMA_B(1); vmwait<SAL + SBL>(); }
      else vmwait<0>();
    } else if constexpr (NST == 2) {
      if (total > 1) { SAM_LDMA_B(0); SAM_LDMA_B(1); vmwait<SBL>(); }
      else vmwait<0>();
    } else {
      if (total > 2) { SAM_LDMA_A(0); SAM_LDMA_A(1); SAM_LDMA_B(0); SAM_LDMA_B(1); SAM_LDMA_B(0); SAM_LDMA_B(1); vmwait<SAL + 2 * SBL>(); }
      else if (total > 1) { SAM_LDMA_A(0); SAM_LDMA_A(1); SAM_LDMA_B(0); SAM_LDMA_B(1); vmwait<SAL + SBL>(); }
      else vmwait<0>();
    }
    __builtin_amdgcn_s_barrier();
    int kt = 0;
    if constexpr (PH == 1) {
      // two intervals per k-tile (upper group: reads of k-tile u | its MFMAs; the lower group one barrier behind): k-tile u+2 goes out, A in the first interval and
      // B in the second, into the stage k-tile u-1 was read from -- by the lower group one interval ago --, then k-tile u+1 must have landed
      for (int u = 0; u < total; ++u) {
        const bool issue = ua < total;
        if (issue) { SAM_LDMA_A(0); SAM_LDMA_A(1); }
        __builtin_amdgcn_s_barrier();
        if (issue) { SAM_LDMA_B(0); SAM_LDMA_B(1); vmwait<SAL + SBL>(); }
        else vmwait<0>();
        __builtin_amdgcn_s_barrier();
        if (++kt == KT) { kt = 0; __builtin_amdgcn_s_barrier(); }
      }
      return;
    }
    for (int u = 0; u < total; ++u) {
      // interval 1 / 2 (upper group: reads of phase 0, MFMAs of phase 0): A of k-tile u+1 [u+2 with three stages] -- its stage was last read in phase 1 of
      // k-tile u-1, by the lower group one barrier ago
      const bool issue_a = ua < total;
      if (issue_a) SAM_LDMA_A(0);
      __builtin_amdgcn_s_barrier();
      if (issue_a) SAM_LDMA_A(1);
      __builtin_amdgcn_s_barrier();
      // interval 3 / 4: B of k-tile u+2 [u+3] into the stage of k-tile u, whose B fragments both groups have read by now; then k-tile u+1 must have landed
      const bool issue_b = ub < total;
      if (issue_b) SAM_LDMA_B(0);
      __builtin_amdgcn_s_barrier();
      if (issue_b) {
        SAM_LDMA_B(1);
        if constexpr (NST == 2) vmwait<SBL>();
        else vmwait<SAL + 2 * SBL>();          // B(u+2), A(u+2), B(u+3) may stay in flight (see gemm8_kernel)
      } else {
        vmwait<0>();
      }
      __builtin_amdgcn_s_barrier();
      if (++kt == KT) { kt = 0; __builtin_amdgcn_s_barrier(); }      // the upper group's alignment barrier at a tile's end
    }
#undef SAM_LDMA_A
#undef SAM_LDMA_B
    return;
  }

  // =============================================================== compute waves
  const int wr = wave >> 2, wc = wave & 3;        // waves w and w+4 share a SIMD: one from each row group
  int m0, n0;
  tile_origin12<BM, BN>(p, blockIdx.x, m0, n0);
  f32x4 acc[TN][TM];
#pragma unroll
  for (int a = 0; a < TN; ++a)
#pragma unroll
    for (int b = 0; b < TM; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
  __builtin_amdgcn_s_barrier();
  if (wr == 1) __builtin_amdgcn_s_barrier();      // lower row group: one barrier behind from here on

  const int sig = ((i >> 3) & 1) | ((g & 1) << 1);
  constexpr int AFR = PH == 1 ? TM : RB;
  bf16x8 af[AFR][2], bfr[TN][2];
  int kt = 0, j = 0, su = 0;
  for (int u = 0; u < total; ++u) {
    const unsigned char* stA = smem + su * STAGE;
    const unsigned char* stB = stA + A_BYTES;
    su = su + 1 == NST ? 0 : su + 1;
    if constexpr (PH == 1) {
      // ---- the whole k-tile: every fragment, then every MFMA
#pragma unroll
      for (int x = 0; x < TN; ++x)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) bfr[x][ks] = frag<BKC>(stB, wc * (BN / 4) + x * 16, ks, i, g, sig);
#pragma unroll
      for (int x = 0; x < AFR; ++x)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) af[x][ks] = frag<AKC>(stA, wr * (BM / 2) + x * 16, ks, i, g, sig);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int x = 0; x < AFR; ++x)
#pragma unroll
          for (int y = 0; y < TN; ++y) acc[y][x] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[y][ks], af[x][ks], acc[y][x], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
    } else {
    // ---- phase 0: all B fragments + upper A rows
#pragma unroll
    for (int x = 0; x < TN; ++x)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) bfr[x][ks] = frag<BKC>(stB, wc * (BN / 4) + x * 16, ks, i, g, sig);
#pragma unroll
    for (int x = 0; x < RB; ++x)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) af[x][ks] = frag<AKC>(stA, wr * (BM / 2) + x * 16, ks, i, g, sig);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // before the barrier: behind it this stage's B region may be refilled
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
    // ---- phase 1: lower A rows
#pragma unroll
    for (int x = 0; x < RB; ++x)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) af[x][ks] = frag<AKC>(stA, wr * (BM / 2) + (RB + x) * 16, ks, i, g, sig);
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
    }
    // ---- end of a tile: epilogue.  Its stores are NOT waited for: the operand queue belongs to the loader waves, this wave's next counted wait is its own
    // epilogue's operand prefetch one tile later
    if (++kt == KT) {
      if (wr == 0) __builtin_amdgcn_s_barrier();
      const bool full = m0 + BM <= p.M && n0 + BN <= p.N;
      if (p.dbg & 1) {
        float sacc = 0.f;
#pragma unroll
        for (int a = 0; a < TN; ++a)
#pragma unroll
          for (int b = 0; b < TM; ++b) sacc += acc[a][b][0] + acc[a][b][1] + acc[a][b][2] + acc[a][b][3];
        if (sacc == 12345.678f) reinterpret_cast<bf16_t*>(p.C)[tid] = (bf16_t)1;
      } else {
        // (fragment rows in halves: the epilogue's batched operand prefetch has to fit 168 registers next to the accumulators)
        gemm_epilogue8<TM, TN, EPI, OutT, 0, RB>(p, acc, m0 + wr * (BM / 2), n0 + wc * (BN / 4), full, p.C, p.ldc, p.accumulate, i, g);
        gemm_epilogue8<TM, TN, EPI, OutT, RB, TM>(p, acc, m0 + wr * (BM / 2), n0 + wc * (BN / 4), full, p.C, p.ldc, p.accumulate, i, g);
      }
      kt = 0;
      if (++j < my_tiles) {
        if (wr == 1) __builtin_amdgcn_s_barrier();
        tile_origin12<BM, BN>(p, blockIdx.x + j * G, m0, n0);
#pragma unroll
        for (int a = 0; a < TN; ++a)
#pragma unroll
          for (int b = 0; b < TM; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
    }
  }
  if constexpr (EPI == SAM_EPI_BIAS_DROPOUT_RES && std::is_same<OutT, bf16_t>::value && AKC) {
    // LayerNorm inside the launch (gemm_common.h: gemm_ln_pass): single-round launches only (launch12 checks), so the block's one tile is behind it.  BEHIND the
    // k-loop on purpose: inside it the pass's 60-odd registers were live next to the loop's and the 192 x 192 kernels spilled 200-400 bytes per lane
    if (p.ln_y) {
      int lm0, ln0;
      tile_origin12<BM, BN>(p, blockIdx.x, lm0, ln0);
      gemm_ln_pass<BM, BN>(p, lm0, ln0, wr, wc, lane);
    }
  }
}

template <int BM, int BN, bool AKC, bool BKC, int EPI, typename OutT, int NST, int PH = 2>
int launch12(GemmArgs a, int n_cu, hipStream_t st) {
  constexpr size_t LDS = (size_t)NST * (BM + BN) * 128;
  static_assert(LDS <= 160 * 1024, "LDS stages");
  static bool once = false;
  if (!once) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(gemm12_kernel<BM, BN, AKC, BKC, EPI, OutT, NST, PH>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS);
    once = true;
  }
  a.tiles_m = (a.M + BM - 1) / BM; a.tiles_n = (a.N + BN - 1) / BN;
  const int tiles = a.tiles_m * a.tiles_n;
  // LayerNorm inside the launch: every tile resident at once (its waves wait for each other), whole column tiles, the workspace's counter range, 32-bit byte offsets
  if (a.ln_y && !(EPI == SAM_EPI_BIAS_DROPOUT_RES && std::is_same<OutT, bf16_t>::value && AKC && tiles <= n_cu && a.N % BN == 0 && a.tiles_n <= 4 && a.tiles_m <= 1024 &&
                  (int64_t)a.M * a.ldc * 2 < (int64_t)0x7fffffff && (int64_t)a.M * 4 * a.tiles_n * 8 < (int64_t)0x7fffffff))
    a.ln_y = nullptr;
  gemm12_kernel<BM, BN, AKC, BKC, EPI, OutT, NST, PH><<<dim3(tiles < n_cu ? tiles : n_cu), dim3(768), LDS, st>>>(a);
  SAM_LAUNCH_CHECK();
  if (a.ln_y && a.ln_done) *a.ln_done = 1;
  return SAM_OK;
}

template <bool BKC, int EPI>
int pick12(const GemmArgs& a, int tile, hipStream_t st) {
  const int n_cu = grid_cu_count();
  // 192 x 256 for the wide outputs (N a multiple of 256 with >= 2.5 rounds of tiles), 192 x 192 (three stages) otherwise; `tile`: 12192 / 12448 force one
  const int t256 = ((a.M + 191) / 192) * ((a.N + 255) / 256);
  bool wide = tile == 12448 || (tile != 12192 && a.N % 256 == 0 && t256 >= 2 * n_cu + n_cu / 2);
  if (tile == 0 && a.N % 256 == 0 && n_cu != device_cu_count()) {
    // CUs withheld (sam_set_cu_reserve): the rule above was tuned on the full chip, where 11648 x 768 is 244 tiles of 192 x 192 on 256 CUs -- ONE round; on
    // 240 or 224 CUs it is two, the second nearly empty.  Makespan of both shapes, same model as gemm8.hip's picker (rounds x (k-tiles x tk + streams x es))
    const int64_t t192 = (int64_t)((a.M + 191) / 192) * ((a.N + 191) / 192);
    const float streams = 1.f + ((EPI == SAM_EPI_BIAS_GELU_GRAD) ? 1.f : 0.f) + (((EPI == SAM_EPI_BIAS_DROPOUT_RES && a.residual) || EPI == SAM_EPI_MUL_AUX) ? 1.f : 0.f);
    const float kt = (float)(a.K / BK);
    const float c192 = (float)((t192 + n_cu - 1) / n_cu) * (kt * 1.14f + streams * 4.7f), c256 = (float)((t256 + n_cu - 1) / n_cu) * (kt * 1.27f + streams * 4.9f);
    wide = c256 < c192;
  }
  if (wide) return launch12<192, 256, true, BKC, EPI, bf16_t, 2>(a, n_cu, st);
  static int ph = -1;
  if (ph < 0) { const char* v = getenv("SAM_GEMM12_PHASES"); ph = v ? atoi(v) : 1; }          // (2: the two-phase schedule, for an A/B)
  if (ph == 2) return launch12<192, 192, true, BKC, EPI, bf16_t, 3, 2>(a, n_cu, st);
  return launch12<192, 192, true, BKC, EPI, bf16_t, 3, 1>(a, n_cu, st);
}

}  // namespace

// SAM_ERR_UNSUPPORTED (error string untouched) when the problem has no instance here: the caller goes on to the 8-wave kernels
int samgemm::gemm12_launch(const GemmArgs& a_in, int lay, int e, int c_is_f32, int tile, hipStream_t st) {
  GemmArgs a = a_in;
  { static int dbg = -1; if (dbg < 0) { const char* v = getenv("SAM_GEMM8_DBG"); dbg = v ? atoi(v) : 0; } a.dbg = dbg & (1 | 8 | 16 | 32); }
  if (lay == 0 && c_is_f32 && e == SAM_EPI_NONE && tile == 12448 && a.K % BK == 0 && a.bias_grad == nullptr && a.M % 8 == 0) {
    // (experiment: the weight-gradient layout -- both operands k-strided, fp32 accumulate -- on the loader-wave core, one problem; profiles/r5_gemm_experiments.txt)
    if ((int64_t)a.K * a.lda * 2 >= (int64_t)0x7fffffff || (int64_t)a.K * a.ldb * 2 >= (int64_t)0x7fffffff) return SAM_ERR_UNSUPPORTED;
    return launch12<192, 256, false, false, SAM_EPI_NONE, float, 2>(a, grid_cu_count(), st);
  }
  if (a.K % BK != 0 || a.split_k > 1 || a.bias_grad != nullptr || c_is_f32 || !(lay & 2)) return SAM_ERR_UNSUPPORTED;
  const int64_t a_rows = a.M, b_rows = (lay & 1) ? a.N : a.K;
  if (a_rows * a.lda * 2 >= (int64_t)0x7fffffff || b_rows * a.ldb * 2 >= (int64_t)0x7fffffff) return SAM_ERR_UNSUPPORTED;
  if (tile == 0 && ((int64_t)((a.M + 191) / 192) * ((a.N + 191) / 192) < 160 || a.K < 256)) return SAM_ERR_UNSUPPORTED;      // small grids: the 4-wave / two-block kernels
  if (tile == 0 && a.N % 256 == 0) {
    // a problem that 256 x 256 tiles cover in whole rounds stays with the 8-wave kernel: per flop its tile reads a third less from LDS than a 192-row tile, which
    // is worth more than the loader waves (QKV forward, N = 2304: 414 tiles = 2 rounds at 81 %: 48.9 us there, 55.0 us here on 732 tiles of 192 x 192)
    const int n_cu = grid_cu_count();
    const int64_t t256 = (int64_t)((a.M + 255) / 256) * (a.N / 256), rounds = (t256 + n_cu - 1) / n_cu;
    if (2 * t256 >= 3 * (int64_t)n_cu && 5 * t256 >= 4 * rounds * n_cu) return SAM_ERR_UNSUPPORTED;
  }
  if (lay == 3) {
    if (e == SAM_EPI_NONE) return pick12<true, SAM_EPI_NONE>(a, tile, st);
    if (e == SAM_EPI_BIAS) return pick12<true, SAM_EPI_BIAS>(a, tile, st);
    if (e == SAM_EPI_BIAS_GELU_GRAD) return pick12<true, SAM_EPI_BIAS_GELU_GRAD>(a, tile, st);
    if (e == SAM_EPI_BIAS_DROPOUT_RES) return pick12<true, SAM_EPI_BIAS_DROPOUT_RES>(a, tile, st);
  } else if (lay == 2) {
    if (e == SAM_EPI_NONE) return pick12<false, SAM_EPI_NONE>(a, tile, st);
    if (e == SAM_EPI_MUL_AUX) return pick12<false, SAM_EPI_MUL_AUX>(a, tile, st);
    if (e == SAM_EPI_BIAS_DROPOUT_RES) return pick12<false, SAM_EPI_BIAS_DROPOUT_RES>(a, tile, st);
  }
  return SAM_ERR_UNSUPPORTED;
}
