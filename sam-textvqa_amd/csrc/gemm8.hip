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
#include "gemm_common.h"

using namespace samgemm;
namespace {

__device__ __forceinline__ int kc_off(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }
__device__ __forceinline__ int ks_sigma(int krow) { return ((krow >> 1) & 1) | (((krow >> 3) & 1) << 1); }

// byte offsets (from the operand base, k-tile 0) of the S 16-byte pieces this thread fetches per k-tile of an operand tile starting at row0
template <bool KC, int S>
__device__ __forceinline__ void src_offsets(unsigned* off, int64_t ld, int row0, int rows, int wave, int lane) {
#pragma unroll
  for (int s = 0; s < S; ++s) {
    const int j = wave * S + s;
    if (KC) {
      const int row = 8 * j + (lane >> 3), pos = lane & 7, c = pos ^ ((row >> 1) & 7);
      const int grow = min(row0 + row, rows - 1);          // rows past the edge: clamped, they only feed outputs that are never stored
      off[s] = (unsigned)((grow * ld + c * 8) * 2);
    } else {
      const int panel = j >> 3, krow = 8 * (j & 7) + (lane >> 3), pos = lane & 7, c = pos ^ (ks_sigma(krow) << 1);
      const int col = min(row0 + panel * 64 + c * 8, rows - 8);
      off[s] = (unsigned)((krow * ld + col) * 2);
    }
  }
}

// slices [0, S) of this wave's share of an operand tile: global -> LDS.  (The resource descriptor is built here, from a plain pointer:
// a __amdgpu_buffer_rsrc_t crossing a template boundary breaks the host-side pass.)
template <int S>
__device__ __forceinline__ void dma_slices(const bf16_t* base, unsigned char* dst, const unsigned* off, unsigned soff) {
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, 0x7fffffff, 0x00020000);
#pragma unroll
  for (int s = 0; s < S; ++s)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(dst + s * 1024), 16, off[s], soff, 0, 0);
}
template <int N>
__device__ __forceinline__ void vmwait() {
  if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  else if constexpr (N == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
  else if constexpr (N == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
  else if constexpr (N == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
  else if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  else static_assert(N <= 4, "vmwait");
}
// 16 rows x 32 k fragment for lane (i, g): k = 32 ks + 8 g + e in both storage kinds
template <bool KC>
__device__ __forceinline__ bf16x8 frag(const unsigned char* region, int row, int ks, int i, int g, int sig) {
  if constexpr (KC) return *reinterpret_cast<const bf16x8*>(region + kc_off(row + i, 4 * ks + g));
  else {
    const int krow = 32 * ks + 8 * g + (i >> 2);
    const unsigned char* q = region + (row >> 6) * 8192 + krow * 128 + (((((row & 63) >> 3) + ((i & 3) >> 1)) ^ (sig << 1)) << 4) + (i & 1) * 8;
    return cat4(lds_read_tr16(q), lds_read_tr16(q + 512));
  }
}

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

template <int BM, int BN, bool AKC, bool BKC, int EPI, typename OutT>
__global__ __launch_bounds__(512, 2) void gemm8_kernel(GemmArgs p) {
  constexpr int TM = BM / 32, TN = BN / 64, SA = BM / 64, SB = BN / 64, RB = TM / 2;
  constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, STAGE = A_BYTES + B_BYTES;
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

  f32x4 acc[TN][TM];
#pragma unroll
  for (int a = 0; a < TN; ++a)
#pragma unroll
    for (int b = 0; b < TM; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

  // prologue: k-tile 0 complete, B of k-tile 1 in flight
  SAM_DMA_A(); SAM_DMA_B();
  if (total > 1) { SAM_DMA_B(); vmwait<SB>(); }
  else vmwait<0>();
  __builtin_amdgcn_s_barrier();
  if (wr == 1) __builtin_amdgcn_s_barrier();      // lower row group: one barrier behind from here on

  const int sig = ((i >> 3) & 1) | ((g & 1) << 1);     // sigma(krow) for krow = 32 ks + 8 g + (i >> 2)
  bf16x8 af[RB][2], bfr[TN][2];
  int kt = 0, j = 0;
  for (int u = 0; u < total; ++u) {
    const unsigned char* stA = smem + (u & 1) * STAGE;
    const unsigned char* stB = stA + A_BYTES;
    // ================= phase 0: all B fragments + upper A rows; DMA of A(u+1) (its stage was last read in phase 1 of k-tile u-1)
#pragma unroll
    for (int x = 0; x < TN; ++x)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) bfr[x][ks] = frag<BKC>(stB, wc * (BN / 4) + x * 16, ks, i, g, sig);
#pragma unroll
    for (int x = 0; x < RB; ++x)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) af[x][ks] = frag<AKC>(stA, wr * (BM / 2) + x * 16, ks, i, g, sig);
    if (ua < total) SAM_DMA_A();
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
    if (ub < total) { SAM_DMA_B(); vmwait<SB>(); }          // k-tile u+1 has landed (loads retire in order); B(u+2) stays in flight
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
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    // ================= end of a tile: epilogue (the next tile's first operands are already in flight), fresh accumulators
    if (++kt == KT) {
      // The upper row group waits one barrier here, so both groups run their epilogues SIDE BY SIDE (left staggered, each epilogue would only
      // be covered by one 18-MFMA phase of the partner: two serial epilogues per tile); the lower group drops back behind afterwards.
      if (wr == 0) __builtin_amdgcn_s_barrier();
      const bool full = m0 + BM <= p.M && n0 + BN <= p.N;
      if constexpr (TM * TN > 18) {     // 32 fragments per wave: two halves, so that the batched operand prefetch of the epilogue fits the register file
        gemm_epilogue8<TM, TN, EPI, OutT, 0, RB>(p, acc, m0 + wr * (BM / 2), n0 + wc * (BN / 4), full, p.C, p.ldc, p.accumulate, i, g);
        gemm_epilogue8<TM, TN, EPI, OutT, RB, TM>(p, acc, m0 + wr * (BM / 2), n0 + wc * (BN / 4), full, p.C, p.ldc, p.accumulate, i, g);
      } else {
        gemm_epilogue8<TM, TN, EPI, OutT>(p, acc, m0 + wr * (BM / 2), n0 + wc * (BN / 4), full, p.C, p.ldc, p.accumulate, i, g);
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

template <int BM, int BN, bool AKC, bool BKC, int EPI, typename OutT>
int launch8(GemmArgs a, int n_cu, hipStream_t st) {
  constexpr size_t LDS = (size_t)2 * (BM + BN) * 128;
  static bool once = false;
  if (!once) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(gemm8_kernel<BM, BN, AKC, BKC, EPI, OutT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS);
    once = true;
  }
  a.tiles_m = (a.M + BM - 1) / BM; a.tiles_n = (a.N + BN - 1) / BN;
  const int tiles = a.tiles_m * a.tiles_n;
  gemm8_kernel<BM, BN, AKC, BKC, EPI, OutT><<<dim3(tiles < n_cu ? tiles : n_cu), dim3(512), LDS, st>>>(a);
  SAM_LAUNCH_CHECK();
  return SAM_OK;
}

// Tile choice.  Measured on the shapes of the step (M = 11648; tools/bench_gemm8.py, one box, both configurations forced in turn): what decides
// is how well the tile count fills whole rounds of the chip's CUs, fill = tiles / (rounds * CUs), times the per-flop rate of the tile
// (256x256 moves 1.3x fewer LDS bytes per flop: 1.36 vs 1.09 PFLOP/s on a full square problem).  N = 2304: fill 0.81 vs 0.95 -> 256 wins by 11 %;
// N = 3072: 0.72 vs 0.95 -> tie; N = 768 (138 vs 244 tiles): 0.54 vs 0.95 -> 192 wins by 12 %.
struct TileCfg { int bm, bn; float rate; };
constexpr TileCfg kCfg[2] = {{256, 256, 1.30f}, {192, 192, 1.0f}};

template <bool AKC, bool BKC, int EPI, typename OutT>
int pick8(const GemmArgs& a, int tile, hipStream_t st) {
  static int n_cu = 0;
  if (n_cu == 0) {
    int dev = 0; hipGetDevice(&dev);
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, dev);
    n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  }
  int best = -1; float best_score = 0.f;
  for (int c = 0; c < 2; ++c) {
    if (tile != 0 && tile != 1000 + kCfg[c].bm) continue;
    const int tiles = ((a.M + kCfg[c].bm - 1) / kCfg[c].bm) * ((a.N + kCfg[c].bn - 1) / kCfg[c].bn);
    const int rounds = (tiles + n_cu - 1) / n_cu;
    const float useful = (float)a.M * (float)a.N / ((float)tiles * kCfg[c].bm * kCfg[c].bn);      // edge tiles compute rows / columns nobody stores
    const float score = kCfg[c].rate * useful * (float)tiles / (float)(rounds * n_cu);
    if (best < 0 || score > best_score) { best = c; best_score = score; }
  }
  if (best == 0) return launch8<256, 256, AKC, BKC, EPI, OutT>(a, n_cu, st);
  if (best == 1) return launch8<192, 192, AKC, BKC, EPI, OutT>(a, n_cu, st);
  return SAM_ERR_UNSUPPORTED;
}

}  // namespace

int samgemm::gemm8_launch(const GemmArgs& a, int lay, int e, int c_is_f32, int tile, hipStream_t st) {
  // the DMA addresses are 32-bit byte offsets from the operand base; k-tiles are whole; nothing here splits K or reduces a bias gradient
  if (a.K % BK != 0 || a.split_k > 1 || a.bias_grad != nullptr || c_is_f32) return SAM_ERR_UNSUPPORTED;
  const int64_t a_rows = (lay & 2) ? a.M : a.K, b_rows = (lay & 1) ? a.N : a.K;
  if (a_rows * a.lda * 2 >= (int64_t)0x7fffffff || b_rows * a.ldb * 2 >= (int64_t)0x7fffffff) return SAM_ERR_UNSUPPORTED;
  if (tile == 0 && ((int64_t)((a.M + 191) / 192) * ((a.N + 191) / 192) < 160 || a.K < 256)) return SAM_ERR_UNSUPPORTED;   // small grids: the 4-wave kernels (2-4 blocks per CU)
  if (lay == 3) {
    if (e == SAM_EPI_NONE) return pick8<true, true, SAM_EPI_NONE, bf16_t>(a, tile, st);
    if (e == SAM_EPI_BIAS) return pick8<true, true, SAM_EPI_BIAS, bf16_t>(a, tile, st);
    if (e == SAM_EPI_BIAS_GELU) return pick8<true, true, SAM_EPI_BIAS_GELU, bf16_t>(a, tile, st);
    if (e == SAM_EPI_BIAS_DROPOUT_RES) return pick8<true, true, SAM_EPI_BIAS_DROPOUT_RES, bf16_t>(a, tile, st);
  } else if (lay == 2) {
    if (e == SAM_EPI_NONE) return pick8<true, false, SAM_EPI_NONE, bf16_t>(a, tile, st);
    if (e == SAM_EPI_DGELU) return pick8<true, false, SAM_EPI_DGELU, bf16_t>(a, tile, st);
    if (e == SAM_EPI_BIAS_DROPOUT_RES) return pick8<true, false, SAM_EPI_BIAS_DROPOUT_RES, bf16_t>(a, tile, st);
  }
  return SAM_ERR_UNSUPPORTED;
}
