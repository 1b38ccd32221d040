// NOT part of libsam_hip.so since round 4: the two-4-wave-blocks-per-CU GEMM of round 3 (force_tile 2256), built, parity-tested and measured slower
// (k-loop 0.78x the 8-wave kernel: profiles/r3z_micro_gemm.txt).  Kept as a probe; it needs csrc/gemm_common.h + gemm8_dev.h to compile.
// bf16 MFMA GEMM, TWO 4-wave blocks per CU (gfx950): the forward / dgrad nn.Linear sites whose fused epilogue is a large part of the launch.
//
// The 8-wave kernels (gemm8.hip) give a CU to ONE block: while that block converts, activates and stores a finished tile (4-12 us per tile: VALU
// for erf-GELU and its derivative, the dropout hash, packing; a burst of 64-256 KB of stores) the matrix pipes idle, and because every CU
// reaches its tile boundary at the same time the chip alternates between all-MFMA and all-store phases (FFN1 forward: 48 us of k-loop,
// 84 us with the epilogue).  Here a CU hosts two INDEPENDENT blocks of four waves -- one wave of each per SIMD, nothing couples them: no
// common barrier, no common LDS -- so one block's epilogue (VALU + stores) runs underneath the other's k-loop (MFMA + LDS), and blocks
// drift apart in time, which spreads the store bursts.  The price: each block stages its own B tile (1.5 x the L2 -> LDS traffic per flop of
// the 256 x 256 tile) and the k-step is 32 deep (three 24 KB stages per block = 72 KB, two blocks = 144 KB of the 160 KB).
//   block tile 256 x 128 x 32, 4 waves as 2 (M) x 2 (N), 128 x 64 per wave (the same per-wave tile as the 8-wave kernels: 32 MFMAs per k-step
//   against 12 fragment reads), persistent over tiles b, b + G, ...
//   k-step u: wait for this wave's DMA pieces of stage u % 3 (counted vmcnt) -> s_barrier -> fragment reads -> DMA of k-step u + 2 into the
//   stage read in step u - 1 (every wave has passed its reads of it: that is what the barrier says) -> 32 MFMAs.  ONE barrier per k-step.
// LDS image of a k-contiguous operand tile at BK = 32: 64-byte rows (32 k), 16 rows per 1 KB DMA slice, 16-byte chunk c of row r stored at
// c ^ ((r >> 1) & 3): conflict-free for the ds_read_b128 row fragments (checked exhaustively over the four lane groups of the instruction).
// k-strided operand (dgrad's W): the 64-column panels of gemm8_dev.h with 32 k-rows per panel instead of 64.
#include "gemm8_dev.h"
#include <stdlib.h>
#include <stdio.h>

using namespace samgemm;
using namespace samgemm8;
namespace {

constexpr int BM4 = 256, BN4 = 128, BK4 = 32, NST = 3;
constexpr int TM4 = 8, TN4 = 4;                       // 16-row / 16-column fragments per wave
constexpr int A4_BYTES = BM4 * BK4 * 2, B4_BYTES = BN4 * BK4 * 2, STAGE4 = A4_BYTES + B4_BYTES;
constexpr int SA4 = A4_BYTES / 1024 / 4, SB4 = B4_BYTES / 1024 / 4;     // 1 KB DMA slices per wave per k-step: 4 + 2

// source byte offsets of this thread's 16-byte pieces of an operand tile starting at row0 (k-step 0)
template <bool KC, int S>
__device__ __forceinline__ void src_offsets4(unsigned* off, int64_t ld, int row0, int rows, int wave, int lane) {
#pragma unroll
  for (int s = 0; s < S; ++s) {
    const int j = wave * S + s;                       // slice index inside the tile
    if (KC) {
      const int row = 16 * j + (lane >> 2), c = (lane & 3) ^ ((row >> 1) & 3);
      const int grow = min(row0 + row, rows - 1);
      off[s] = (unsigned)((grow * ld + c * 8) * 2);
    } else {
      const int panel = j >> 2, krow = 8 * (j & 3) + (lane >> 3), pos = lane & 7, c = pos ^ (ks_sigma(krow) << 1);
      const int col = min(row0 + panel * 64 + c * 8, rows - 8);
      off[s] = (unsigned)((krow * ld + col) * 2);
    }
  }
}
// 16 rows x 32 k fragment for lane (i, g): k = 8 g + e
template <bool KC>
__device__ __forceinline__ bf16x8 frag4(const unsigned char* region, int row, int i, int g, int sig) {
  if constexpr (KC) return *reinterpret_cast<const bf16x8*>(region + (row + i) * 64 + ((g ^ ((i >> 1) & 3)) << 4));      // (row is a multiple of 16)
  else {
    const int krow = 8 * g + (i >> 2);
    const unsigned char* q = region + (row >> 6) * 4096 + krow * 128 + (((((row & 63) >> 3) + ((i & 3) >> 1)) ^ (sig << 1)) << 4) + (i & 1) * 8;
    return cat4(lds_read_tr16(q), lds_read_tr16(q + 512));
  }
}
template <int N>
__device__ __forceinline__ void vmwait4() {
  if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  else if constexpr (N == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  else static_assert(N == 0 || N == 6, "vmwait4");
}

template <int BM, int BN>
__device__ __forceinline__ void tile_origin4(const GemmArgs& p, int id, int& m0, int& n0) {
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

template <bool BKC, int EPI, typename OutT>
__global__ __launch_bounds__(256, 2) void gemm4_kernel(GemmArgs p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), i = lane & 15, g = lane >> 4;
  const int wr = wave >> 1, wc = wave & 1;
  const int G = gridDim.x, nblk = p.tiles_m * p.tiles_n;
  const int KT = p.K / BK4;
  const unsigned kstepA = BK4 * 2, kstepB = BKC ? BK4 * 2 : (unsigned)(BK4 * p.ldb * 2);
  const int sig = ((i >> 3) & 1) | ((g & 1) << 1);
  // The two blocks of a CU are dispatched together and every tile takes the same time: left alone they reach every tile boundary together and their
  // epilogues coincide instead of hiding under each other's k-loops.  The second wave of blocks (ids >= half the grid: the second slot of
  // every CU) therefore starts `stagger` x ~3.7 us late.
  if (p.stagger > 0) {
    // which of the CU's two slots this block got: the hardware wave slot of its waves on their SIMDs (HW_REG_HW_ID.WAVE_ID, bits 3:0) -- the block
    // dispatched second sits in the odd slot.  (Placement-independent for correctness: the delay is only a speed knob.)
    const unsigned slot = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 4) & 1u;
    if (slot)
      for (int s = 0; s < p.stagger; ++s) __builtin_amdgcn_s_sleep(127);
  }
  bool first = true;
  for (int id = blockIdx.x; id < nblk; id += G) {
    int m0, n0;
    tile_origin4<BM4, BN4>(p, id, m0, n0);
    unsigned offA[SA4], offB[SB4];
    src_offsets4<true, SA4>(offA, p.lda, m0, p.M, wave, lane);
    src_offsets4<BKC, SB4>(offB, p.ldb, n0, p.N, wave, lane);
    if (!first) __builtin_amdgcn_s_barrier();          // every wave is out of the previous tile's last fragment reads before its stages are refilled
    first = false;
#define SAM4_DMA(u)                                                                                             \
  do {                                                                                                          \
    unsigned char* st_ = smem + ((u) % NST) * STAGE4;                                                           \
    dma_slices<SA4>(p.A, st_ + wave * (SA4 * 1024), offA, (u) * kstepA);                                        \
    dma_slices<SB4>(p.B, st_ + A4_BYTES + wave * (SB4 * 1024), offB, (u) * kstepB);                             \
  } while (0)
    f32x4 acc[TN4][TM4];
#pragma unroll
    for (int a = 0; a < TN4; ++a)
#pragma unroll
      for (int b = 0; b < TM4; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    SAM4_DMA(0);
    if (KT > 1) SAM4_DMA(1);
    for (int u = 0; u < KT; ++u) {
      if (u + 1 < KT) vmwait4<6>();                    // this wave's pieces of k-step u have landed (k-step u + 1 may still be in flight)
      else vmwait4<0>();
      __builtin_amdgcn_s_barrier();
      const unsigned char* stA = smem + (u % NST) * STAGE4;
      const unsigned char* stB = stA + A4_BYTES;
      bf16x8 af[TM4], bfr[TN4];
#pragma unroll
      for (int x = 0; x < TN4; ++x) bfr[x] = frag4<BKC>(stB, wc * 64 + x * 16, i, g, sig);
#pragma unroll
      for (int x = 0; x < TM4; ++x) af[x] = frag4<true>(stA, wr * 128 + x * 16, i, g, sig);
      if (u + 2 < KT) SAM4_DMA(u + 2);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int x = 0; x < TM4; ++x)
#pragma unroll
        for (int y = 0; y < TN4; ++y) acc[y][x] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[y], af[x], acc[y][x], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
#undef SAM4_DMA
    const bool full = m0 + BM4 <= p.M && n0 + BN4 <= p.N;
    if (p.dbg == 1) {                                   // (tuning: the k-loop alone; one impossible store keeps the accumulators live)
      float sacc = 0.f;
#pragma unroll
      for (int a = 0; a < TN4; ++a)
#pragma unroll
        for (int b = 0; b < TM4; ++b) sacc += acc[a][b][0] + acc[a][b][1] + acc[a][b][2] + acc[a][b][3];
      if (sacc == 12345.678f) reinterpret_cast<bf16_t*>(p.C)[tid] = (bf16_t)1;
    } else {
      gemm_epilogue8<TM4, TN4, EPI, OutT, 0, TM4 / 2>(p, acc, m0 + wr * 128, n0 + wc * 64, full, p.C, p.ldc, p.accumulate, i, g);
      gemm_epilogue8<TM4, TN4, EPI, OutT, TM4 / 2, TM4>(p, acc, m0 + wr * 128, n0 + wc * 64, full, p.C, p.ldc, p.accumulate, i, g);
    }
  }
}

template <bool BKC, int EPI, typename OutT>
int launch4(GemmArgs a, hipStream_t st) {
  constexpr size_t LDS = (size_t)NST * STAGE4;
  static bool once = false;
  if (!once) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(gemm4_kernel<BKC, EPI, OutT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS);
    once = true;
  }
  if (getenv("SAM_GEMM4_OCC")) {
    int nb = -1;
    hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, reinterpret_cast<const void*>(gemm4_kernel<BKC, EPI, OutT>), 256, LDS);
    fprintf(stderr, "gemm4 occupancy: %d blocks per CU (LDS %zu)\n", nb, LDS);
  }
  a.tiles_m = (a.M + BM4 - 1) / BM4; a.tiles_n = (a.N + BN4 - 1) / BN4;
  const int tiles = a.tiles_m * a.tiles_n, slots = 2 * device_cu_count();
  gemm4_kernel<BKC, EPI, OutT><<<dim3(tiles < slots ? tiles : slots), dim3(256), LDS, st>>>(a);
  SAM_LAUNCH_CHECK();
  return SAM_OK;
}

}  // namespace

// force_tile 2256: this kernel or SAM_ERR_UNSUPPORTED
int samgemm::gemm4_launch(const GemmArgs& a_in, int lay, int e, int c_is_f32, hipStream_t st) {
  GemmArgs a = a_in;
  { static int dbg = -1; if (dbg < 0) { const char* v = getenv("SAM_GEMM8_DBG"); dbg = v ? atoi(v) : 0; } a.dbg = dbg; }
  { static int sg = -1; if (sg < 0) { const char* v = getenv("SAM_GEMM4_STAGGER"); sg = v ? atoi(v) : 0; } a.stagger = sg; }
  if (a.group_m <= 0) a.group_m = 8;
  if (a.K % BK4 != 0 || a.split_k > 1 || a.bias_grad != nullptr || c_is_f32) return SAM_ERR_UNSUPPORTED;
  const int64_t b_rows = (lay & 1) ? a.N : a.K;
  if ((int64_t)a.M * a.lda * 2 >= (int64_t)0x7fffffff || b_rows * a.ldb * 2 >= (int64_t)0x7fffffff) return SAM_ERR_UNSUPPORTED;
  if (lay == 3) {
    if (e == SAM_EPI_NONE) return launch4<true, SAM_EPI_NONE, bf16_t>(a, st);
    if (e == SAM_EPI_BIAS) return launch4<true, SAM_EPI_BIAS, bf16_t>(a, st);
    if (e == SAM_EPI_BIAS_GELU_GRAD) return launch4<true, SAM_EPI_BIAS_GELU_GRAD, bf16_t>(a, st);
    if (e == SAM_EPI_BIAS_DROPOUT_RES) return launch4<true, SAM_EPI_BIAS_DROPOUT_RES, bf16_t>(a, st);
  } else if (lay == 2) {
    if (e == SAM_EPI_NONE) return launch4<false, SAM_EPI_NONE, bf16_t>(a, st);
    if (e == SAM_EPI_MUL_AUX) return launch4<false, SAM_EPI_MUL_AUX, bf16_t>(a, st);
    if (e == SAM_EPI_BIAS_DROPOUT_RES) return launch4<false, SAM_EPI_BIAS_DROPOUT_RES, bf16_t>(a, st);
  }
  return SAM_ERR_UNSUPPORTED;
}
