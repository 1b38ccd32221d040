// bf16 MFMA GEMM with fused epilogues for the nn.Linear sites of the SA-M4C path (gfx950).
//
// 128x128x64 block tile, 256 threads = 2x2 waves of 64x64, v_mfma_f32_16x16x32_bf16, fp32 accumulate.
// Operands go global -> LDS directly (global_load_lds_dwordx4, no VGPR staging, no ds_write pass) into a
// double-buffered LDS image: the DMA of k-tile t+1 is in flight during the MFMAs of tile t, one barrier per tile.
// The LDS image is XOR-swizzled; since the DMA writes lane-linear, the swizzle is applied to each lane's SOURCE
// address.  Only a partial last k-tile (K % 64 != 0) takes the predicated register-staged path.
// Either operand may be stored k-contiguous (fragments by ds_read_b128/b64) or k-strided (row index =
// contraction index: fragments by ds_read_b64_tr_b16), which gives forward / dgrad / wgrad from one
// template without any transpose pass through HBM.
// The MFMA is issued "transposed" (rows = n, cols = m) so each lane ends up with 4 CONSECUTIVE n of one
// output row: bias/residual/aux are 8- or 16-byte vector accesses and stores are 8 B (bf16) / 16 B (fp32).
#include "gemm_common.h"
#include <stdlib.h>

using namespace samgemm;
namespace {

// ---- LDS images (one 64-deep k-tile) -----------------------------------------------------------------
// k-contiguous operand: [R rows][64 k] bf16, 128-byte rows, 16-byte chunk c stored at c ^ ((row>>1)&7)
__device__ __forceinline__ int kc_off(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }
// k-strided operand: [64 k][C cols] bf16, 2C-byte rows, chunk c (8 cols) stored at c ^ (S(k)<<1), S(k) = k&3 | ((k>>3)&1)<<2:
// a 32-lane ds_read_b64_tr_b16 group reads the 8 k-rows {8g..8g+3, 8(g+1)..8(g+1)+3}; S makes them 8 distinct 32-byte bank ranges
__device__ __forceinline__ int ks_swz(int krow) { return ((krow & 3) | (((krow >> 3) & 1) << 2)) << 1; }
// (for C = 64 only 8 chunks exist per row: the XOR is masked to stay inside the row, leaving a 2-way conflict on that small-tile config)
template <int C>
__device__ __forceinline__ int ks_off(int krow, int chunk) { return krow * (2 * C) + ((chunk ^ (ks_swz(krow) & (C / 8 - 1))) << 4); }

// predicated register-staged fill (partial last k-tile only). R = tile extent along the non-k index; NT threads.
template <bool KC, int R, int NT>
__device__ __forceinline__ void load_tile(uint4 (&r)[R * 8 / NT], const bf16_t* base, int64_t ld, int row0, int rows, int k0, int K, int tid) {
#pragma unroll
  for (int j = 0; j < R * 8 / NT; ++j) {
    const int c = tid + NT * j;
    r[j] = make_uint4(0, 0, 0, 0);
    if (KC) {
      const int row = row0 + (c >> 3), k = k0 + (c & 7) * 8;
      if (row < rows && k < K) r[j] = *reinterpret_cast<const uint4*>(base + (int64_t)row * ld + k);
    } else {
      const int k = k0 + c / (R / 8), col = row0 + (c % (R / 8)) * 8;
      if (k < K && col < rows) r[j] = *reinterpret_cast<const uint4*>(base + (int64_t)k * ld + col);
    }
  }
}
template <bool KC, int R, int NT>
__device__ __forceinline__ void store_tile(unsigned char* lds, const uint4 (&r)[R * 8 / NT], int tid) {
#pragma unroll
  for (int j = 0; j < R * 8 / NT; ++j) {
    const int c = tid + NT * j;
    const int off = KC ? kc_off(c >> 3, c & 7) : ks_off<R>(c / (R / 8), c % (R / 8));
    *reinterpret_cast<uint4*>(lds + off) = r[j];
  }
}

// direct-to-LDS tile fill.  One wave instruction writes 1 KB lane-linearly: lane l lands at byte 16*l of the slice, so
// lane l must FETCH the chunk that the swizzled image keeps there.  Rows / columns past the matrix edge are clamped to
// the last valid one: they only feed output rows/cols that are never stored.  K must cover the whole 64-wide tile.
template <bool KC, int R, int NWAVES>
__device__ __forceinline__ void glds_tile(unsigned char* lds, const bf16_t* base, int64_t ld, int row0, int rows, int k0, int wave, int lane) {
  constexpr int SLICES = R / 8;   // 1 KB slices in the 16*R-byte... (R rows x 128 B, or 64 k-rows x 2R B): both R*128 bytes
#pragma unroll
  for (int jj = 0; jj < SLICES / NWAVES; ++jj) {
    const int j = wave * (SLICES / NWAVES) + jj;
    const bf16_t* src;
    if (KC) {
      const int row = 8 * j + (lane >> 3), pos = lane & 7;
      const int c = pos ^ ((row >> 1) & 7);
      const int grow = min(row0 + row, rows - 1);
      src = base + (int64_t)grow * ld + k0 + c * 8;
    } else {
      const int byte = j * 1024 + lane * 16;
      const int krow = byte / (2 * R), pos = (byte % (2 * R)) >> 4;
      const int cc = pos ^ (ks_swz(krow) & (R / 8 - 1));
      const int col = min(row0 + cc * 8, rows - 8);
      src = base + (int64_t)(k0 + krow) * ld + col;
    }
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)(lds + j * 1024), 16, 0, 0);
  }
}

// fragment of 16 rows (row0..row0+15 of the operand's M/N index) x 32 k (step ks) for lane (i,g): k = 32ks + 8g + e for BOTH
// storage kinds, so any pairing of operands agrees on the contraction order.  k-contiguous: one ds_read_b128; k-strided: two
// ds_read_b64_tr_b16 (k-rows 8g..8g+3 and 8g+4..8g+7; S(k+4) = S(k) for those rows, so the second read is a fixed offset).
// (An earlier version used a split k map with two ds_read_b64 on the k-contiguous side: the compiler fused them into
// ds_read2_b64, whose 32-bank rule gave 40% LDS conflict cycles in dgrad - SQ_LDS_BANK_CONFLICT/SQ_LDS_IDX_ACTIVE.)
template <bool KC, int R>
__device__ __forceinline__ bf16x8 load_frag(const unsigned char* lds, int row0, int ks, int i, int g) {
  if (KC) {
    return *reinterpret_cast<const bf16x8*>(lds + kc_off(row0 + i, 4 * ks + g));
  } else {
    const int krow = 32 * ks + 8 * g + (i >> 2);
    const unsigned char* p = lds + ks_off<R>(krow, (row0 >> 3) + ((i & 3) >> 1)) + (i & 1) * 8;
    return cat4(lds_read_tr16(p), lds_read_tr16(p + 4 * 2 * R));
  }
}

// Tile configuration: BM x BN block tile, WM x WN waves, each wave (BM/WM) x (BN/WN).
//   <128,128,2,2>: 256 threads, 64 KB LDS, 2 blocks/CU  -- 64 flop per byte staged
//   <256,256,2,4>: 512 threads, 128 KB LDS, 1 block/CU  -- 128 flop per byte staged (opt-in via force_tile, see launch())
template <int N>
__device__ __forceinline__ void vmwait_n() {
  static_assert(N >= 0 && N <= 63, "vmcnt is a 6-bit counter");
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// NST = LDS stages.  2: the round-1 loop (one k-tile of lead, __syncthreads per k-tile -- the compiler drains vmcnt to 0 in front of it, so every k-tile pays
// most of a memory latency).  3 / 4 (round 6, the 64 x 64 configuration): a ring of NST stages with NST - 1 k-tiles requested ahead, raw s_barrier and a COUNTED
// s_waitcnt vmcnt(tiles still allowed in flight x DMA instructions per wave and tile): the launches this configuration serves are the latency chains of the step --
// TextBert's 1280-row products (12 k-tiles at K = 768: one block per tile slot, nothing to hide a k-tile's latency behind), the heads, the split-K partials --
// where a block's time was 12 x (latency + 0.05 us of MFMAs).  Same k order, same accumulation: bit-identical results.
template <int BM, int BN, int WM, int WN, bool AKC, bool BKC, int EPI, typename OutT, int NST = 2>
__device__ __forceinline__ void gemm_block(const GemmArgs& p, const int linear_block) {
  constexpr int NT = 64 * WM * WN, NWAVES = WM * WN;
  constexpr int TM = BM / WM / 16, TN = BN / WN / 16;      // 16x16 fragments per wave
  static_assert(BM % (16 * WM) == 0 && BN % (16 * WN) == 0 && (BM / 8) % NWAVES == 0 && (BN / 8) % NWAVES == 0 && (BM * 8) % NT == 0 && (BN * 8) % NT == 0, "tile/wave shape");
  static_assert(AKC || (BM & (BM - 1)) == 0, "k-strided A tiles need a power-of-two BM (XOR swizzle range)");
  constexpr int A_BYTES = BM * BK * 2, B_BYTES = BN * BK * 2, STAGE_BYTES = A_BYTES + B_BYTES;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];   // [stage][A|B]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, i = lane & 15, g = lane >> 4;
  const int wm = wave % WM, wn = wave / WM;

  // Tile order.  (1) XCD-aware: block b runs on XCD b%8 (its own 4 MB L2), so each XCD gets a contiguous run of tile ids.
  // (2) Inside that run tiles are visited in GROUP_M x tiles_n super-columns (m fastest within a group of 8 rows): the ~64 tiles
  // an XCD works on at once then form an ~8x8 patch that needs 8 A panels + 8 B panels (3 MB at K=768: fits the L2) instead of
  // 2-3 A panels + ALL B panels of an n-fastest walk (5+ MB: the weight matrix kept falling out to the Infinity Cache).
  const int nblk = p.tiles_m * p.tiles_n;
  int bid = linear_block % nblk;
  const int split = linear_block / nblk;
  {
    const int q = nblk / 8, r = nblk % 8, xcd = bid % 8, loc = bid / 8;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
  }
  const int GROUP_M = p.group_m;
  const int per_group = GROUP_M * p.tiles_n;
  const int group = bid / per_group, first_m = group * GROUP_M;
  const int gsize = min(p.tiles_m - first_m, GROUP_M);
  const int in_group = bid - group * per_group;
  const int m0 = (first_m + in_group % gsize) * BM, n0 = (in_group / gsize) * BN;

  f32x4 acc[TN][TM];
#pragma unroll
  for (int a = 0; a < TN; ++a)
#pragma unroll
    for (int b = 0; b < TM; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int KT_all = (p.K + BK - 1) / BK;
  const int per = (KT_all + p.split_k - 1) / p.split_k;
  const int kt_begin = split * per, KT = min(KT_all, kt_begin + per);
  if (kt_begin >= KT) return;   // whole block leaves together: no barrier has been reached
  const bool do_bias = !AKC && p.bias_grad != nullptr && n0 == 0 && wn == 0;
  f32x4 accb[TM];
#pragma unroll
  for (int a = 0; a < TM; ++a) accb[a] = f32x4{0.f, 0.f, 0.f, 0.f};
  const short one_bf16 = (short)0x3F80;
  const bf16x8 ones = {one_bf16, one_bf16, one_bf16, one_bf16, one_bf16, one_bf16, one_bf16, one_bf16};

  const int KT_full = p.K / BK;   // k-tiles entirely inside K go direct-to-LDS; a partial last tile is register staged
  auto fill = [&](int kt, int stage) {
    unsigned char* As_ = smem + stage * STAGE_BYTES;
    unsigned char* Bs_ = As_ + A_BYTES;
    if (kt < KT_full) {
      glds_tile<AKC, BM, NWAVES>(As_, p.A, p.lda, m0, p.M, kt * BK, wave, lane);
      glds_tile<BKC, BN, NWAVES>(Bs_, p.B, p.ldb, n0, p.N, kt * BK, wave, lane);
    } else {
      uint4 ra[BM * 8 / NT], rb[BN * 8 / NT];
      load_tile<AKC, BM, NT>(ra, p.A, p.lda, m0, p.M, kt * BK, p.K, tid);
      load_tile<BKC, BN, NT>(rb, p.B, p.ldb, n0, p.N, kt * BK, p.K, tid);
      store_tile<AKC, BM, NT>(As_, ra, tid);
      store_tile<BKC, BN, NT>(Bs_, rb, tid);
    }
  };
  constexpr int LPT = (BM / 8 + BN / 8) / NWAVES;          // direct-to-LDS instructions per wave and k-tile
  const bool ring = NST > 2 && KT <= KT_full;                // (a partial last k-tile is register staged: ordinary loads + ds_writes, the simple loop handles it)
  if (ring) {
#pragma unroll
    for (int s0 = 0; s0 < NST - 1; ++s0)
      if (kt_begin + s0 < KT) fill(kt_begin + s0, s0);
  } else {
    fill(kt_begin, 0);
  }
  int stage = 0;
  for (int kt = kt_begin; kt < KT; ++kt) {
    if (ring) {
      // k-tiles kt .. min(kt + NST - 2, KT - 1) are out; this wave's pieces of k-tile kt have landed once no more than (tiles behind it) x LPT loads are pending
      const int ahead = min(NST - 2, KT - 1 - kt);
      if (ahead >= 2) vmwait_n<(NST > 3 ? 2 : 0) * LPT>();      // (rings deeper than three: measured slower -- fewer resident blocks -- and not instantiated)
      else if (ahead == 1) vmwait_n<LPT>();
      else vmwait_n<0>();
      __builtin_amdgcn_s_barrier();          // everyone's pieces of k-tile kt are in LDS, and everyone has consumed the fragments of k-tile kt - 1
      if (kt + NST - 1 < KT) fill(kt + NST - 1, stage == 0 ? NST - 1 : stage - 1);      // into the stage k-tile kt - 1 was read from
    } else {
      __syncthreads();   // tile kt has landed (the compiler drains vmcnt before the barrier) and everyone left stage^1
      if (kt + 1 < KT) fill(kt + 1, stage ^ 1);
    }
    const unsigned char* As = smem + stage * STAGE_BYTES;
    const unsigned char* Bs = As + A_BYTES;
    stage = ring ? (stage + 1 == NST ? 0 : stage + 1) : stage ^ 1;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 af[TM], bf[TN];
#pragma unroll
      for (int t = 0; t < TM; ++t) af[t] = load_frag<AKC, BM>(As, wm * (BM / WM) + t * 16, ks, i, g);
#pragma unroll
      for (int t = 0; t < TN; ++t) bf[t] = load_frag<BKC, BN>(Bs, wn * (BN / WN) + t * 16, ks, i, g);
#pragma unroll
      for (int tn = 0; tn < TN; ++tn)
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
          acc[tn][tm] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[tn], af[tm], acc[tn][tm], 0, 0, 0);
      if (do_bias) {  // D[n][m] = sum_k 1 * A(m,k): every row n holds the column sums of this wave's m range
#pragma unroll
        for (int tm = 0; tm < TM; ++tm) accb[tm] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, af[tm], accb[tm], 0, 0, 0);
      }
    }
  }

  if (do_bias && g == 0) {   // each m has exactly one writer per split: no atomics
    float* bdst = p.split_k > 1 ? p.ws + (int64_t)p.split_k * p.M * p.N + (int64_t)split * p.M : p.bias_grad;
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
      const int m = m0 + wm * (BM / WM) + tm * 16 + i;
      if (m < p.M) bdst[m] = (p.split_k > 1 || !p.accumulate) ? accb[tm][0] : bdst[m] + accb[tm][0];      // accumulate = 0: C and the bias gradient are overwritten
    }
  }
  void* Cout = p.C;
  int64_t ldc = p.ldc;
  int accumulate = p.accumulate;
  if (p.split_k > 1) { Cout = p.ws + (int64_t)split * p.M * p.N; ldc = p.N; accumulate = 0; }
  gemm_epilogue<TM, TN, EPI, OutT>(p, acc, m0 + wm * (BM / WM), n0 + wn * (BN / WN), m0 + BM <= p.M && n0 + BN <= p.N, Cout, ldc, accumulate, i, g);
}

template <int BM, int BN, int WM, int WN, bool AKC, bool BKC, int EPI, typename OutT, int NST = 2>
__global__ __launch_bounds__(64 * WM * WN, 2) void gemm_kernel(GemmArgs p) {   // 2 waves per SIMD = two 4-wave blocks (or one 8-wave block) per CU
  gemm_block<BM, BN, WM, WN, AKC, BKC, EPI, OutT, NST>(p, blockIdx.x);
}

// Grouped launch: up to 12 independent problems of the same kind in ONE grid (block ranges start at multiples of 8 so the XCD-aware
// tile remap still sees its hardware XCD).  Used for the four weight-gradient GEMMs of an encoder layer: 36+108+144+144 = 432 tiles
// fill the 512 block slots in a single round, so no split-K (and no partial-sum reduction pass) is needed at all.
struct GroupArgs { GemmArgs a[SAM_MAX_GROUP]; int start[SAM_MAX_GROUP + 1]; int count; };
template <int BM, int BN, int WM, int WN, bool AKC, bool BKC, int EPI, typename OutT>
__global__ __launch_bounds__(64 * WM * WN, 2) void gemm_group_kernel(GroupArgs g) {
  int prob = 0;
#pragma unroll
  for (int q = 1; q < SAM_MAX_GROUP; ++q)
    if (q < g.count && (int)blockIdx.x >= g.start[q]) prob = q;
  const int local = blockIdx.x - g.start[prob];
  if (local >= g.a[prob].tiles_m * g.a[prob].tiles_n) return;     // padding block
  gemm_block<BM, BN, WM, WN, AKC, BKC, EPI, OutT>(g.a[prob], local);
}

// a (+)= ws[s][i] for s = 1 .. S-1 in that order, the loads of four partials requested together (the plain loop leaves ONE load in flight per thread and
// waits for it before the next: S - 1 memory round trips in a row in kernels that are nothing but that); the additions keep the order of the plain loop
__device__ __forceinline__ void splitk_sum(float4& a, const float* ws, int S, int64_t MN, int64_t i4) {
  int s = 1;
  for (; s + 4 <= S; s += 4) {
    float4 b[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) b[u] = reinterpret_cast<const float4*>(ws + (int64_t)(s + u) * MN)[i4];
#pragma unroll
    for (int u = 0; u < 4; ++u) { a.x += b[u].x; a.y += b[u].y; a.z += b[u].z; a.w += b[u].w; }
  }
  for (; s < S; ++s) {
    const float4 b = reinterpret_cast<const float4*>(ws + (int64_t)s * MN)[i4];
    a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
  }
}

// C[m,n] += sum_s ws[s][m,n]; bias_grad[m] += sum_s wsb[s][m]   (fixed summation order: reproducible)
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* ws, int S, int M, int N, float* C, int64_t ldc, float* bias_grad) {
  const int64_t mn4 = (int64_t)M * N / 4;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < mn4; i += (int64_t)gridDim.x * 256) {
    float4 a = reinterpret_cast<const float4*>(ws)[i];
    splitk_sum(a, ws, S, (int64_t)M * N, i);
    const int64_t e = i * 4, m = e / N, n = e - m * N;
    float4* dst = reinterpret_cast<float4*>(C + m * ldc + n);
    const float4 c = *dst;
    *dst = make_float4(c.x + a.x, c.y + a.y, c.z + a.z, c.w + a.w);
  }
  if (bias_grad) {
    const float* wsb = ws + (int64_t)S * M * N;
    for (int64_t m = (int64_t)blockIdx.x * 256 + threadIdx.x; m < M; m += (int64_t)gridDim.x * 256) {
      float a = 0.f;
      for (int s = 0; s < S; ++s) a += wsb[(int64_t)s * M + m];
      bias_grad[m] += a;
    }
  }
}

// Split-K for skinny problems WITH an epilogue (TextBert at 20 tokens/sample: 240 64x64 tiles x 48 k-tiles on 256 CUs is pure
// latency, 38 us for 6 GFLOP; the classifier dgrad: 144 tiles x 79 k-tiles, 58 us): the GEMM runs as an fp32 / no-epilogue split
// into the workspace, this kernel sums the S partials in fixed order and applies the epilogue of the requested kind.
template <int EPI, typename OutT>
__global__ __launch_bounds__(256) void splitk_epilogue_kernel(const float* ws, int S, GemmArgs p) {
  const int64_t mn4 = (int64_t)p.M * p.N / 4, MN = (int64_t)p.M * p.N;
  unsigned seed_lo = p.seed_lo, seed_hi = p.seed_hi, off_lo = p.off_lo, off_hi = p.off_hi;
  if (EPI == SAM_EPI_BIAS_DROPOUT_RES) rng_resolve(p.rng_state, seed_lo, seed_hi, off_lo, off_hi);
  for (int64_t i4 = (int64_t)blockIdx.x * 256 + threadIdx.x; i4 < mn4; i4 += (int64_t)gridDim.x * 256) {
    float4 a = reinterpret_cast<const float4*>(ws)[i4];
    splitk_sum(a, ws, S, MN, i4);
    const int64_t e = i4 * 4;
    const int m = (int)(e / p.N), n = (int)(e - (int64_t)m * p.N);
    float v[4] = {a.x, a.y, a.z, a.w};
    if (EPI == SAM_EPI_BIAS || EPI == SAM_EPI_BIAS_GELU || EPI == SAM_EPI_BIAS_DROPOUT_RES || EPI == SAM_EPI_BIAS_GELU_GRAD) {
      if (p.bias) {
        const float4 b4 = *reinterpret_cast<const float4*>(p.bias + n);
        v[0] += b4.x; v[1] += b4.y; v[2] += b4.z; v[3] += b4.w;
      }
    }
    if (EPI == SAM_EPI_BIAS_GELU) {
      *reinterpret_cast<uint2*>(p.aux_out + (int64_t)m * p.ld_aux + n) = make_uint2(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]));
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = gelu_erf(v[r]);
    }
    if (EPI == SAM_EPI_BIAS_GELU_GRAD) {
      float d[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = gelu_erf_and_grad(v[r], d[r]);
      if (p.aux_out) *reinterpret_cast<uint2*>(p.aux_out + (int64_t)m * p.ld_aux + n) = make_uint2(pack_bf16x2(d[0], d[1]), pack_bf16x2(d[2], d[3]));
    }
    if (EPI == SAM_EPI_DGELU) {
      const uint2 x = *reinterpret_cast<const uint2*>(p.aux_in + (int64_t)m * p.ld_aux + n);
      v[0] *= gelu_erf_grad(bf_lo(x.x)); v[1] *= gelu_erf_grad(bf_hi(x.x));
      v[2] *= gelu_erf_grad(bf_lo(x.y)); v[3] *= gelu_erf_grad(bf_hi(x.y));
    }
    if (EPI == SAM_EPI_MUL_AUX) {
      const uint2 x = *reinterpret_cast<const uint2*>(p.aux_in + (int64_t)m * p.ld_aux + n);
      v[0] *= bf_lo(x.x); v[1] *= bf_hi(x.x); v[2] *= bf_lo(x.y); v[3] *= bf_hi(x.y);
    }
    if (EPI == SAM_EPI_BIAS_DROPOUT_RES) {
      if (p.thr16) {   // same (row, col/8) Philox stream as the in-GEMM epilogue
        const u32x4 rn = hidden_dropout_bits((unsigned)m, (unsigned)(n >> 3), off_lo, off_hi, seed_lo, seed_hi);
        const unsigned lo = (n & 4) ? rn.z : rn.x, hi = (n & 4) ? rn.w : rn.y;
        v[0] = (lo & 0xffffu) >= p.thr16 ? v[0] * p.inv_keep : 0.f;
        v[1] = (lo >> 16) >= p.thr16 ? v[1] * p.inv_keep : 0.f;
        v[2] = (hi & 0xffffu) >= p.thr16 ? v[2] * p.inv_keep : 0.f;
        v[3] = (hi >> 16) >= p.thr16 ? v[3] * p.inv_keep : 0.f;
      }
      if (p.residual) {
        const uint2 x = *reinterpret_cast<const uint2*>(p.residual + (int64_t)m * p.ldr + n);
        v[0] += bf_lo(x.x); v[1] += bf_hi(x.x); v[2] += bf_lo(x.y); v[3] += bf_hi(x.y);
      }
    }
    Store4<OutT>::st(p.C, (int64_t)m * p.ldc + n, v, 0);
  }
}

// split factor for the epilogue split: 64x64 tiles, fill the ~1024 resident slots (4 blocks per CU) once, >= 6 k-tiles per split
int pick_epilogue_split(int M, int N, int K, int64_t ws_bytes) {
  const int kt = (K + BK - 1) / BK;
  const int64_t tiles = (int64_t)((M + 63) / 64) * ((N + 63) / 64);
  if (kt < 24 || tiles >= 512 || N % 4) return 1;
  int64_t s = 1024 / tiles;
  if (s > kt / 6) s = kt / 6;
  if (s > 8) s = 8;
  const int64_t per = (int64_t)M * N * (int64_t)sizeof(float);
  if (s * per > ws_bytes) s = ws_bytes / per;
  return s < 2 ? 1 : (int)s;
}

// The same pass for SAM_EPI_BIAS_DROPOUT_RES followed by a LayerNorm over the output row (sam_ln_fuse): one wave per row sums the partials, applies bias /
// dropout / residual exactly as splitk_epilogue_kernel does, stores the bf16 sums (the backward's z) and normalises the ROUNDED values with the arithmetic of
// ln_fwd_kernel (rowops.hip) -- bit-identical to the two launches it replaces.
struct LnFuseArgs { const float* gamma; const float* beta; float eps; bf16_t* y; int64_t ldy; float* mean; float* rstd; };
template <int NCH>
__global__ __launch_bounds__(256) void splitk_epilogue_ln_kernel(const float* ws, int S, GemmArgs p, LnFuseArgs ln) {
  const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= p.M) return;
  const int D = p.N, nchunk = D >> 2;
  const int64_t MN = (int64_t)p.M * p.N;
  unsigned seed_lo = p.seed_lo, seed_hi = p.seed_hi, off_lo = p.off_lo, off_hi = p.off_hi;
  rng_resolve(p.rng_state, seed_lo, seed_hi, off_lo, off_hi);
  float v[NCH][4];
  float4 acc[NCH];
  uint2 res[NCH];
#pragma unroll
  for (int j = 0; j < NCH; ++j) {          // every load up front, at a clamped chunk
    const int c = min(lane + 64 * j, nchunk - 1);
    acc[j] = *reinterpret_cast<const float4*>(ws + (int64_t)row * D + 4 * c);
    res[j] = p.residual ? *reinterpret_cast<const uint2*>(p.residual + (int64_t)row * p.ldr + 4 * c) : make_uint2(0u, 0u);
  }
  for (int s = 1; s < S; ++s)
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
      const int c = min(lane + 64 * j, nchunk - 1);
      const float4 b = *reinterpret_cast<const float4*>(ws + (int64_t)s * MN + (int64_t)row * D + 4 * c);
      acc[j].x += b.x; acc[j].y += b.y; acc[j].z += b.z; acc[j].w += b.w;
    }
  float g4[NCH][4], b4[NCH][4];
#pragma unroll
  for (int j = 0; j < NCH; ++j) {
    const int c = min(lane + 64 * j, nchunk - 1), n = 4 * c;
    v[j][0] = acc[j].x; v[j][1] = acc[j].y; v[j][2] = acc[j].z; v[j][3] = acc[j].w;
    if (p.bias) {
      const float4 bb = *reinterpret_cast<const float4*>(p.bias + n);
      v[j][0] += bb.x; v[j][1] += bb.y; v[j][2] += bb.z; v[j][3] += bb.w;
    }
    if (p.thr16) {
      const u32x4 rn = hidden_dropout_bits((unsigned)row, (unsigned)(n >> 3), off_lo, off_hi, seed_lo, seed_hi);
      const unsigned lo = (n & 4) ? rn.z : rn.x, hi = (n & 4) ? rn.w : rn.y;
      v[j][0] = (lo & 0xffffu) >= p.thr16 ? v[j][0] * p.inv_keep : 0.f;
      v[j][1] = (lo >> 16) >= p.thr16 ? v[j][1] * p.inv_keep : 0.f;
      v[j][2] = (hi & 0xffffu) >= p.thr16 ? v[j][2] * p.inv_keep : 0.f;
      v[j][3] = (hi >> 16) >= p.thr16 ? v[j][3] * p.inv_keep : 0.f;
    }
    v[j][0] += bf_lo(res[j].x); v[j][1] += bf_hi(res[j].x); v[j][2] += bf_lo(res[j].y); v[j][3] += bf_hi(res[j].y);
    const uint2 z = make_uint2(pack_bf16x2(v[j][0], v[j][1]), pack_bf16x2(v[j][2], v[j][3]));
    if (lane + 64 * j < nchunk) *reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(p.C) + (int64_t)row * p.ldc + n) = z;
    v[j][0] = bf_lo(z.x); v[j][1] = bf_hi(z.x); v[j][2] = bf_lo(z.y); v[j][3] = bf_hi(z.y);      // LayerNorm sees what it would read back
    *reinterpret_cast<float4*>(g4[j]) = *reinterpret_cast<const float4*>(ln.gamma + n);
    *reinterpret_cast<float4*>(b4[j]) = *reinterpret_cast<const float4*>(ln.beta + n);
  }
  float sm = 0.f;
#pragma unroll
  for (int j = 0; j < NCH; ++j) {
    if (lane + 64 * j >= nchunk) v[j][0] = v[j][1] = v[j][2] = v[j][3] = 0.f;
    sm += (v[j][0] + v[j][1]) + (v[j][2] + v[j][3]);
  }
  const float mean = wave_sum(sm) / D;
  float q = 0.f;
#pragma unroll
  for (int j = 0; j < NCH; ++j)
    if (lane + 64 * j < nchunk)
#pragma unroll
      for (int e = 0; e < 4; ++e) { const float d_ = v[j][e] - mean; q += d_ * d_; }
  const float rstd = 1.0f / sqrtf(wave_sum(q) / D + ln.eps);
#pragma unroll
  for (int j = 0; j < NCH; ++j) {
    const int c = lane + 64 * j;
    if (c < nchunk) {
      float o[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = g4[j][e] * ((v[j][e] - mean) * rstd) + b4[j][e];
      *reinterpret_cast<uint2*>(ln.y + (int64_t)row * ln.ldy + 4 * c) = make_uint2(pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3]));
    }
  }
  if (lane == 0) { ln.mean[row] = mean; ln.rstd[row] = rstd; }
}

template <int BM, int BN, int WM, int WN, bool AKC, bool BKC, int EPI, typename OutT, int NST = 2>
int launch_cfg(GemmArgs a, hipStream_t st) {
  constexpr size_t LDS = (size_t)NST * (BM + BN) * BK * 2;
  static bool once = false;
  if (!once) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_kernel<BM, BN, WM, WN, AKC, BKC, EPI, OutT, NST>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS);
    once = true;
  }
  gemm_kernel<BM, BN, WM, WN, AKC, BKC, EPI, OutT, NST><<<dim3(a.tiles_m * a.tiles_n * a.split_k), dim3(64 * WM * WN), LDS, st>>>(a);
  if (a.split_used) *a.split_used = a.split_k;
  if (a.split_k > 1 && !a.defer_reduce) {
    const int64_t mn4 = (int64_t)a.M * a.N / 4;
    splitk_reduce_kernel<<<dim3((unsigned)min((int64_t)2048, (mn4 + 255) / 256)), dim3(256), 0, st>>>(a.ws, a.split_k, a.M, a.N, (float*)a.C, a.ldc, a.bias_grad);
  }
  SAM_LAUNCH_CHECK();
  return SAM_OK;
}

// every split s covers k-tiles [s*per, (s+1)*per), per = ceil(kt / S): shrink S until the last split is non-empty (an empty split would
// leave its workspace slice unwritten and the reduction would add garbage)
int no_empty_split(int kt, int s) {
  if (s <= 1) return 1;
  const int per = (kt + s - 1) / s;
  return (kt + per - 1) / per;
}

// split-K factor for a given tile count: fill the `per_cu` x 256 resident block slots exactly ONCE (tiles x S just above a
// multiple of the slot count costs a whole extra round: measured 91 us at S=3 vs 123 us at S=4 for 144 tiles), >= 4 k-tiles per split
int pick_split(int requested, int tiles, int kt, int per_cu, int64_t per_split_bytes, int64_t ws_bytes) {
  int s = requested;
  if (s < 0) {
    s = (256 * per_cu) / tiles;
    if (s > kt / 4) s = kt / 4;
    if (s > 32) s = 32;
  }
  if (s > kt) s = kt;
  if (per_split_bytes > 0 && (int64_t)s * per_split_bytes > ws_bytes) s = (int)(ws_bytes / per_split_bytes);
  return no_empty_split(kt, s < 1 ? 1 : s);
}

// Block-tile choice.  Two 128x128 blocks fit a CU (LDS), i.e. 512 concurrent tiles; a grid of 546 tiles (M=11648, N=768, the
// O-projection / FFN2 / two dgrads of every layer) then runs as one full round plus a nearly empty one.  Taller tiles
// (BM = 160, 192; still 2 blocks per CU) trade a little per-tile time for a whole round: pick the BM that minimises
// rounds(BM) x BM.  Only the k-contiguous-A layouts (forward, dgrad) need it; wgrad tops its grid up with split-K instead.
template <bool AKC, bool BKC, int EPI, typename OutT>
int launch(GemmArgs a, hipStream_t st, int want_split, int64_t ws_bytes, int force_tile) {
  const int kt = (a.K + BK - 1) / BK;
  const int64_t per_split = ((int64_t)a.M * a.N + a.M) * (int64_t)sizeof(float);
  const int tn = (a.N + 127) / 128;
  if (want_split == 0 && (force_tile == 0 || force_tile >= 12000)) {
    // the 12-wave kernels with loader waves (gemm12.hip) are the first choice for the large products (SAM_GEMM12=0: the 8-wave kernels, for an A/B);
    // force_tile 12192 / 12448 forces one
    static int v12 = -1;
    if (v12 < 0) { const char* e = getenv("SAM_GEMM12"); v12 = e ? atoi(e) : 1; }
    if (v12 || force_tile >= 12000) {
      const int rc = gemm12_launch(a, (AKC ? 2 : 0) | (BKC ? 1 : 0), EPI, sizeof(OutT) == 4, force_tile, st);
      if (rc != SAM_ERR_UNSUPPORTED) { if (a.split_used) *a.split_used = 1; return rc; }
      if (force_tile >= 12000) { sam_set_error("sam_gemm_bf16: force_tile=%d: the 12-wave kernels have no instance for this problem", force_tile); return rc; }
    }
  }
  if (want_split == 0 && (force_tile == 0 || (force_tile >= 1000 && force_tile < 12000))) {
    // large problems: the 8-wave persistent kernels (gemm8.hip); they decline (SAM_ERR_UNSUPPORTED) what they have no instance for
    static int v2 = -1;
    if (v2 < 0) { const char* e = getenv("SAM_GEMM8"); v2 = e ? atoi(e) : 1; }
    if (v2 || force_tile >= 1000) {
      const int rc = gemm8_launch(a, (AKC ? 2 : 0) | (BKC ? 1 : 0), EPI, sizeof(OutT) == 4, force_tile, st);
      if (rc != SAM_ERR_UNSUPPORTED) { if (a.split_used) *a.split_used = 1; return rc; }
      if (force_tile >= 1000) { sam_set_error("sam_gemm_bf16: force_tile=%d: the 8-wave kernels have no instance for this problem", force_tile); return rc; }
    }
  }
  if (force_tile == 256) {
    a.tiles_m = (a.M + 255) / 256; a.tiles_n = (a.N + 255) / 256;
    if (want_split != 0) a.split_k = pick_split(want_split, a.tiles_m * a.tiles_n, kt, 1, per_split, ws_bytes);
    return launch_cfg<256, 256, 2, 4, AKC, BKC, EPI, OutT>(a, st);
  }
  // small problems (TextBert at 20 tokens/sample, the classifier): 128x128 tiles leave most CUs idle; 64x64 tiles (32 KB of LDS,
  // >= 4 blocks per CU) quadruple the grid.  Used when the 128x128 grid would not even fill half of the 512 block slots.
  if (force_tile == 64 || (force_tile == 0 && want_split == 0 && (int64_t)((a.M + 127) / 128) * tn < 256)) {
    a.tiles_m = (a.M + 63) / 64; a.tiles_n = (a.N + 63) / 64;
    // LDS stages of the 64 x 64 configuration (16 KB each).  2 (default): the round-1 loop.  SAM_GEMM64_STAGES=3: a ring of three with two k-tiles requested ahead
    // and counted vmcnt waits (gemm_block<..., NST>) -- built in round 6 on the theory that TextBert's 1280-row products are chains of exposed k-tile latencies,
    // measured, and NOT faster (profiles/r6_gemm_experiments.txt: QKV 13.5 -> 13.3 us, O-projection 9.2 -> 8.7, four stages 16.5 / 9.1 at two blocks per CU;
    // only the unsplit K = 3072 product gains, 26.8 -> 20.5, which the split-K form already beats): 720 tiles of 64 x 64 re-fetch A 36 and B 20 times --
    // 138 MB of L2 -> LDS traffic for a 4.5 GFLOP product -- and that stream, not its latency, is what a launch waits for.  Bit-identical results either way.
    static int nst = -1;
    if (nst < 0) { const char* e = getenv("SAM_GEMM64_STAGES"); nst = e ? atoi(e) : 2; if (nst != 3) nst = 2; }
    if (want_split != 0) a.split_k = pick_split(want_split, a.tiles_m * a.tiles_n, kt, nst == 2 ? 4 : 3, per_split, ws_bytes);
    if (nst == 3) return launch_cfg<64, 64, 2, 2, AKC, BKC, EPI, OutT, 3>(a, st);
    return launch_cfg<64, 64, 2, 2, AKC, BKC, EPI, OutT>(a, st);
  }
  int bm = 128;
  if constexpr (AKC) {
    if (force_tile == 0 && want_split == 0) {
      int64_t best = -1;
      for (int cand : {128, 160, 192}) {
        const int64_t tiles = (int64_t)((a.M + cand - 1) / cand) * tn;
        const int64_t cost = ((tiles + 511) / 512) * (cand + 128) * 16 + (cand - 128);   // rounds x per-tile ingest (A+B rows), ties -> smaller
        if (best < 0 || cost < best) { best = cost; bm = cand; }
      }
    } else if (force_tile == 160 || force_tile == 192) bm = force_tile;
  }
  a.tiles_m = (a.M + bm - 1) / bm; a.tiles_n = tn;
  if (want_split != 0) a.split_k = pick_split(want_split, a.tiles_m * a.tiles_n, kt, 2, per_split, ws_bytes);
  if constexpr (AKC) {
    if (bm == 160) return launch_cfg<160, 128, 2, 2, AKC, BKC, EPI, OutT>(a, st);
    if (bm == 192) return launch_cfg<192, 128, 2, 2, AKC, BKC, EPI, OutT>(a, st);
  }
  return launch_cfg<128, 128, 2, 2, AKC, BKC, EPI, OutT>(a, st);
}

}  // namespace

static int grouped_impl(const sam_gemm_desc* descs, int count, void* stream, int ft);

extern "C" int sam_gemm_bf16_grouped(const sam_gemm_desc* descs, int count, void* stream) {
  SAM_REQUIRE(descs && count >= 1 && count <= SAM_MAX_GROUP8, "sam_gemm_bf16_grouped: 1..%d problems", SAM_MAX_GROUP8);
  return grouped_impl(descs, count, stream, descs[0].force_tile);
}

static int grouped_impl(const sam_gemm_desc* descs, int count, void* stream, const int ft) {
  SAM_REQUIRE(descs && count >= 1 && count <= SAM_MAX_GROUP8, "sam_gemm_bf16_grouped: 1..%d problems", SAM_MAX_GROUP8);
  GroupArgs g = {};             // (the 4-wave kernel's arguments: up to SAM_MAX_GROUP problems; larger sets exist for the 8-wave kernel only)
  g.count = count;
  int total = 0;
  static int bm_env = -1;
  if (bm_env < 0) { const char* e = getenv("SAM_WGRAD_TILE_M"); bm_env = e ? atoi(e) : 0; }
  const int bm = bm_env == 256 ? 256 : 128;
  for (int q = 0; q < count; ++q) {
    const sam_gemm_desc* d = descs + q;
    SAM_REQUIRE(!d->a_kcontig && !d->b_kcontig && d->c_is_f32 && d->epilogue == SAM_EPI_NONE && (d->split_k == 0 || d->split_k == 1),
                "sam_gemm_bf16_grouped: problem %d is not a plain fp32 wgrad-layout GEMM", q);
    SAM_REQUIRE(d->M > 0 && d->N > 0 && d->K > 0 && d->A && d->B && d->C, "sam_gemm_bf16_grouped: problem %d is empty", q);
    SAM_REQUIRE(d->M % 8 == 0 && d->N % 8 == 0 && d->lda % 8 == 0 && d->ldb % 8 == 0 && d->ldc % 4 == 0, "sam_gemm_bf16_grouped: problem %d: M, N, lda, ldb must be multiples of 8", q);
    SAM_REQUIRE(((uintptr_t)d->A % 16 == 0) && ((uintptr_t)d->B % 16 == 0) && ((uintptr_t)d->C % 16 == 0), "sam_gemm_bf16_grouped: problem %d: operands must be 16-byte aligned", q);
    if (q >= SAM_MAX_GROUP) continue;
    GemmArgs& a = g.a[q];
    a.M = d->M; a.N = d->N; a.K = d->K;
    a.A = (const bf16_t*)d->A; a.lda = d->lda; a.B = (const bf16_t*)d->B; a.ldb = d->ldb; a.C = d->C; a.ldc = d->ldc;
    a.accumulate = d->accumulate ? 1 : 0;
    a.inv_keep = 1.0f;
    a.tiles_m = (d->M + bm - 1) / bm; a.tiles_n = (d->N + 127) / 128;
    // tile order per problem: each XCD (private 4 MB L2) gets a contiguous run of tile ids, so the run should partition the WIDER operand
    // and replicate only the narrower one.  n-fastest (group_m = 1): a run = a few m-rows x all n => A (dy) partitioned, B (x) read by all 8
    // XCDs; m-fastest (group_m = tiles_m): the opposite.  dW2 = dy2^T h (A 768 wide, B 3072 wide) walked n-fastest made every XCD
    // stream all 71.6 MB of h: 573 MB of L2 misses for one problem.
    { static int gm = -1; if (gm < 0) { const char* e = getenv("SAM_GEMM_GROUP_M_WGRAD"); gm = e ? atoi(e) : 0; }
      a.group_m = gm > 0 ? gm : (gm == 0 && a.tiles_n > a.tiles_m ? a.tiles_m : 1); }
    a.split_k = 1;
    a.bias_grad = d->bias_grad;
    g.start[q] = total;
    total += (a.tiles_m * a.tiles_n + 7) / 8 * 8;     // keep every problem's first block on XCD 0
  }
  if (count <= SAM_MAX_GROUP) g.start[count] = total;
  // first choice: the 8-wave kernel with in-launch pair exchange (gemm8w.hip); it declines (K % 64, no workspace, tile counts it is not built for)
  // without touching the error string.  descs[0].force_tile: 128 = this file's 4-wave kernel, 1256 = the 8-wave kernel or an error.
  {
    static int use8 = -1;
    if (use8 < 0) { const char* e = getenv("SAM_GEMM8W"); use8 = e ? atoi(e) : 1; }
    SAM_REQUIRE(ft == 0 || ft == 128 || ft == 1256 || ft == 12448, "sam_gemm_bf16_grouped: force_tile must be 0, 128, 1256 or 12448");
    // the loader-wave grouped kernel (gemm12w.hip): opt-in (SAM_GEMM12W=1; force_tile 12448 forces it).  Measured SLOWER than the 8-wave kernel below on the
    // step's sets -- the weight-gradient loop is bound by L2 -> LDS bytes (~8 TB/s chip-wide for k-strided 128-byte segments), and a 192 x 256 tile moves 17 % more
    // bytes per flop than a 256 x 256 one: pair alone 440 vs 367 us although its 256 CUs are all busy (profiles/r5_gemm_experiments.txt).
    static int use12 = -1;
    if (use12 < 0) { const char* e = getenv("SAM_GEMM12W"); use12 = e ? atoi(e) : 0; }
    if ((use12 && ft == 0) || ft == 12448) {
      const int rc = gemm12w_grouped(descs, count, (hipStream_t)stream);
      if (rc == SAM_OK) return SAM_OK;
      SAM_REQUIRE(rc == SAM_ERR_UNSUPPORTED && ft != 12448, "sam_gemm_bf16_grouped: the loader-wave grouped kernel cannot run this problem set (fewer deep tiles than CUs, K %% 64, workspace)");
    }
    if ((use8 && ft == 0) || ft == 1256) {
      const int rc = gemm8w_grouped(descs, count, (hipStream_t)stream);
      if (rc == SAM_OK) return SAM_OK;
      SAM_REQUIRE(rc == SAM_ERR_UNSUPPORTED && ft != 1256, "sam_gemm_bf16_grouped: the 8-wave grouped kernel cannot run this problem set (K %% 64, workspace, tile count)");
    }
  }
  if (count > SAM_MAX_GROUP) {
    // sets of 13..SAM_MAX_GROUP8 problems exist for the 8-wave kernel; when it declines one (a K that is not a multiple of 64: batch sizes other than
    // multiples of 32 / the last partial batch of an epoch; a tile count outside its limits: a GPU with fewer CUs) the set goes out as consecutive
    // chunks of at most SAM_MAX_GROUP problems, each of which may still take the 8-wave kernel by itself and has the 4-wave kernel below it.
    // Whether a call succeeds never depends on the device's CU count or on K % 64.
    for (int q0 = 0; q0 < count; q0 += SAM_MAX_GROUP) {
      const int rc = grouped_impl(descs + q0, count - q0 < SAM_MAX_GROUP ? count - q0 : SAM_MAX_GROUP, stream, ft);
      if (rc != SAM_OK) return rc;
    }
    return SAM_OK;
  }
  static bool once = false;
  if (!once) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_group_kernel<128, 128, 2, 2, false, false, SAM_EPI_NONE, float>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * (128 + 128) * BK * 2);
    hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_group_kernel<256, 128, 4, 2, false, false, SAM_EPI_NONE, float>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * (256 + 128) * BK * 2);
    once = true;
  }
  if (bm == 256)
    gemm_group_kernel<256, 128, 4, 2, false, false, SAM_EPI_NONE, float><<<dim3(total), dim3(512), (size_t)2 * (256 + 128) * BK * 2, (hipStream_t)stream>>>(g);
  else
    gemm_group_kernel<128, 128, 2, 2, false, false, SAM_EPI_NONE, float><<<dim3(total), dim3(256), (size_t)2 * (128 + 128) * BK * 2, (hipStream_t)stream>>>(g);
  SAM_LAUNCH_CHECK();
  return SAM_OK;
}

extern "C" int64_t sam_gemm_grouped_ws_bytes(const sam_gemm_desc* descs, int count) {
  if (!descs || count < 1 || count > SAM_MAX_GROUP8) return 0;
  int tiles = 0;
  for (int q = 0; q < count; ++q) tiles += ((descs[q].M + 255) / 256) * ((descs[q].N + 255) / 256);
  const int64_t a = gemm8w_ws_bytes(tiles), b = gemm12w_ws_bytes(descs, count);
  return a > b ? a : b;
}

extern "C" int64_t sam_gemm_ln_ws_bytes(int M, int N) { return (M > 0 && N > 0) ? gemm_ln_ws_bytes(M, N) : 0; }

extern "C" int sam_gemm_splitk_reduce(const float* ws, int split_k, int M, int N, float* C, int64_t ldc, float* bias_grad, void* stream) {
  SAM_REQUIRE(ws && C && split_k >= 1 && M > 0 && N > 0 && N % 4 == 0 && ldc % 4 == 0, "sam_gemm_splitk_reduce: bad arguments");
  const int64_t mn4 = (int64_t)M * N / 4;
  splitk_reduce_kernel<<<dim3((unsigned)min((int64_t)2048, (mn4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream>>>(ws, split_k, M, N, C, ldc, bias_grad);
  SAM_LAUNCH_CHECK();
  return SAM_OK;
}

extern "C" int sam_gemm_bf16(const sam_gemm_desc* d, void* stream) {
  SAM_REQUIRE(d, "sam_gemm_bf16: null descriptor");
  if (d->ln) d->ln->done = 0;
  SAM_REQUIRE(d->M > 0 && d->N > 0 && d->K > 0, "sam_gemm_bf16: empty problem %dx%dx%d", d->M, d->N, d->K);
  SAM_REQUIRE(d->A && d->B && d->C, "sam_gemm_bf16: null operand");
  SAM_REQUIRE(d->N % 8 == 0 && d->lda % 8 == 0 && d->ldb % 8 == 0 && d->ldc % 4 == 0, "sam_gemm_bf16: N, lda, ldb must be multiples of 8 (N=%d lda=%lld ldb=%lld)", d->N, (long long)d->lda, (long long)d->ldb);
  SAM_REQUIRE(!d->a_kcontig || d->K % 8 == 0, "sam_gemm_bf16: K must be a multiple of 8 for k-contiguous A (K=%d)", d->K);
  SAM_REQUIRE(!d->b_kcontig || d->K % 8 == 0, "sam_gemm_bf16: K must be a multiple of 8 for k-contiguous B (K=%d)", d->K);
  SAM_REQUIRE(d->a_kcontig || d->M % 8 == 0, "sam_gemm_bf16: M must be a multiple of 8 for k-strided A (M=%d)", d->M);
  SAM_REQUIRE(((uintptr_t)d->A % 16 == 0) && ((uintptr_t)d->B % 16 == 0) && ((uintptr_t)d->C % 16 == 0), "sam_gemm_bf16: operands must be 16-byte aligned");
  SAM_REQUIRE(!d->accumulate || d->c_is_f32, "sam_gemm_bf16: accumulate needs an fp32 C");
  SAM_REQUIRE(d->p_drop >= 0.f && d->p_drop < 1.f, "sam_gemm_bf16: p_drop out of range");
  GemmArgs a = {};
  a.M = d->M; a.N = d->N; a.K = d->K;
  a.A = (const bf16_t*)d->A; a.lda = d->lda; a.B = (const bf16_t*)d->B; a.ldb = d->ldb; a.C = d->C; a.ldc = d->ldc;
  a.bias = d->bias; a.residual = (const bf16_t*)d->residual; a.ldr = d->ldr;
  a.aux_out = (bf16_t*)d->aux_out; a.aux_in = (const bf16_t*)d->aux_in; a.ld_aux = d->ld_aux;
  a.accumulate = d->accumulate;
  a.thr16 = dropout_thr16(d->p_drop);
  a.inv_keep = a.thr16 ? 1.0f / (1.0f - (float)a.thr16 / 65536.0f) : 1.0f;
  a.seed_lo = (unsigned)d->seed; a.seed_hi = (unsigned)(d->seed >> 32); a.off_lo = (unsigned)d->offset; a.off_hi = (unsigned)(d->offset >> 32);
  a.rng_state = sam_get_rng_state();
  a.split_k = 1;
  { static int gm = -1; if (gm < 0) { const char* e = getenv("SAM_GEMM_GROUP_M"); gm = e ? atoi(e) : 0; } a.group_m = gm > 0 ? gm : (d->split_k != 0 && d->split_k != 1 ? 1 : 8); }   // measured L2 hit rate: fwd 74% -> 81% with 8; split-K wgrad prefers 1
  a.bias_grad = d->bias_grad;
  a.ws = d->ws;
  a.defer_reduce = d->defer_reduce;
  a.split_used = const_cast<int32_t*>(&d->split_k_used);
  SAM_REQUIRE(!d->bias_grad || (!d->a_kcontig && !d->b_kcontig), "sam_gemm_bf16: bias_grad is a wgrad-layout (0,0) feature");
  if (d->epilogue == SAM_EPI_BIAS_GELU) SAM_REQUIRE(d->aux_out && d->ld_aux % 4 == 0, "sam_gemm_bf16: BIAS_GELU needs aux_out");
  if (d->epilogue == SAM_EPI_BIAS_GELU_GRAD) SAM_REQUIRE(!d->aux_out || d->ld_aux % 4 == 0, "sam_gemm_bf16: BIAS_GELU_GRAD: ld_aux must be a multiple of 4");      // aux_out NULL = do not store GELU'"'"'

  if (d->epilogue == SAM_EPI_DGELU || d->epilogue == SAM_EPI_MUL_AUX) SAM_REQUIRE(d->aux_in && d->ld_aux % 4 == 0, "sam_gemm_bf16: DGELU / MUL_AUX need aux_in");
  if (d->epilogue == SAM_EPI_BIAS_DROPOUT_RES) SAM_REQUIRE(!d->residual || d->ldr % 4 == 0, "sam_gemm_bf16: bad residual ld");
  int want_split = (d->split_k == 1) ? 0 : d->split_k;
  hipStream_t st = (hipStream_t)stream;
  if (want_split != 0 && !(d->c_is_f32 && d->epilogue == SAM_EPI_NONE && d->accumulate)) {
    // epilogue split (see splitk_epilogue_kernel): split_k = -1 lets the library decide, and it may decide not to
    SAM_REQUIRE(!d->accumulate && !d->bias_grad, "sam_gemm_bf16: split_k with an epilogue cannot accumulate / reduce a bias gradient");
    SAM_REQUIRE(d->ws && ((uintptr_t)d->ws % 16 == 0), "sam_gemm_bf16: split_k needs a 16-byte aligned workspace");
    int S = want_split < 0 ? pick_epilogue_split(d->M, d->N, d->K, d->ws_bytes) : want_split;
    if (S > (d->K + BK - 1) / BK) S = (d->K + BK - 1) / BK;
    S = no_empty_split((d->K + BK - 1) / BK, S);
    SAM_REQUIRE((int64_t)S * d->M * d->N * (int64_t)sizeof(float) <= d->ws_bytes || S <= 1, "sam_gemm_bf16: workspace too small for split_k=%d", S);
    if (S > 1) {
      GemmArgs g = a;
      g.bias = nullptr; g.residual = nullptr; g.aux_out = nullptr; g.aux_in = nullptr; g.thr16 = 0; g.accumulate = 0;
      g.tiles_m = (d->M + 63) / 64; g.tiles_n = (d->N + 63) / 64; g.split_k = S; g.group_m = 1; g.defer_reduce = 1; g.bias_grad = nullptr;
      const int lay2 = (d->a_kcontig ? 2 : 0) | (d->b_kcontig ? 1 : 0);
      int rc = lay2 == 3 ? launch_cfg<64, 64, 2, 2, true, true, SAM_EPI_NONE, float>(g, st)
             : lay2 == 2 ? launch_cfg<64, 64, 2, 2, true, false, SAM_EPI_NONE, float>(g, st)
             : lay2 == 0 ? launch_cfg<64, 64, 2, 2, false, false, SAM_EPI_NONE, float>(g, st) : SAM_ERR_UNSUPPORTED;
      if (rc) { if (rc == SAM_ERR_UNSUPPORTED) sam_set_error("sam_gemm_bf16: no split kernel for layout (0,1)"); return rc; }
      const int e2 = d->epilogue;
      if (d->ln && e2 == SAM_EPI_BIAS_DROPOUT_RES && !d->c_is_f32 && d->N % 4 == 0 && d->N <= 2048) {
        const sam_ln_fuse* l = d->ln;
        SAM_REQUIRE(l->gamma && l->beta && l->y && l->mean && l->rstd && l->ldy >= d->N, "sam_gemm_bf16: incomplete sam_ln_fuse");
        LnFuseArgs la = {l->gamma, l->beta, l->eps, (bf16_t*)l->y, l->ldy, l->mean, l->rstd};
        const dim3 lgrid((unsigned)((d->M + 3) / 4));
        switch ((d->N / 4 + 63) / 64) {
#define SAM_LN_CASE(NC) case NC: splitk_epilogue_ln_kernel<NC><<<lgrid, dim3(256), 0, st>>>(d->ws, S, a, la); break;
          SAM_LN_CASE(1) SAM_LN_CASE(2) SAM_LN_CASE(3) SAM_LN_CASE(4) SAM_LN_CASE(5) SAM_LN_CASE(6) SAM_LN_CASE(7) SAM_LN_CASE(8)
#undef SAM_LN_CASE
        }
        SAM_LAUNCH_CHECK();
        d->ln->done = 1;
        *const_cast<int32_t*>(&d->split_k_used) = S;
        return SAM_OK;
      }
      const int64_t mn4 = (int64_t)d->M * d->N / 4;
      const dim3 grid((unsigned)min((int64_t)2048, (mn4 + 255) / 256));
#define SAM_SPLIT_EPI(E, T) splitk_epilogue_kernel<E, T><<<grid, dim3(256), 0, st>>>(d->ws, S, a)
      if (d->c_is_f32) {
        if (e2 == SAM_EPI_NONE) SAM_SPLIT_EPI(SAM_EPI_NONE, float);
        else if (e2 == SAM_EPI_BIAS) SAM_SPLIT_EPI(SAM_EPI_BIAS, float);
        else { sam_set_error("sam_gemm_bf16: fp32 output supports epilogues NONE / BIAS only"); return SAM_ERR_UNSUPPORTED; }
      } else {
        if (e2 == SAM_EPI_NONE) SAM_SPLIT_EPI(SAM_EPI_NONE, bf16_t);
        else if (e2 == SAM_EPI_BIAS) SAM_SPLIT_EPI(SAM_EPI_BIAS, bf16_t);
        else if (e2 == SAM_EPI_BIAS_GELU) SAM_SPLIT_EPI(SAM_EPI_BIAS_GELU, bf16_t);
        else if (e2 == SAM_EPI_DGELU) SAM_SPLIT_EPI(SAM_EPI_DGELU, bf16_t);
        else if (e2 == SAM_EPI_BIAS_GELU_GRAD) SAM_SPLIT_EPI(SAM_EPI_BIAS_GELU_GRAD, bf16_t);
        else if (e2 == SAM_EPI_MUL_AUX) SAM_SPLIT_EPI(SAM_EPI_MUL_AUX, bf16_t);
        else if (e2 == SAM_EPI_BIAS_DROPOUT_RES) SAM_SPLIT_EPI(SAM_EPI_BIAS_DROPOUT_RES, bf16_t);
        else { sam_set_error("sam_gemm_bf16: unknown epilogue %d", e2); return SAM_ERR_UNSUPPORTED; }
      }
#undef SAM_SPLIT_EPI
      SAM_LAUNCH_CHECK();
      *const_cast<int32_t*>(&d->split_k_used) = S;
      return SAM_OK;
    }
    want_split = 0;      // not worth splitting: the plain path below
    *const_cast<int32_t*>(&d->split_k_used) = 1;
  }
  if (want_split != 0) {
    SAM_REQUIRE(d->c_is_f32 && d->epilogue == SAM_EPI_NONE && d->accumulate, "sam_gemm_bf16: split_k needs an fp32 C with accumulate=1 and no epilogue");
    SAM_REQUIRE(d->ws && ((uintptr_t)d->ws % 16 == 0), "sam_gemm_bf16: split_k needs a 16-byte aligned workspace");
    const int64_t per_split = ((int64_t)d->M * d->N + d->M) * (int64_t)sizeof(float);
    SAM_REQUIRE(want_split < 0 || (int64_t)want_split * per_split <= d->ws_bytes || want_split > (d->K + BK - 1) / BK,
                "sam_gemm_bf16: workspace too small for split_k=%d (%lld bytes)", d->split_k, (long long)d->ws_bytes);
  }
  if (d->ln && d->ln->xws && want_split == 0 && d->epilogue == SAM_EPI_BIAS_DROPOUT_RES && !d->c_is_f32 && d->a_kcontig && d->N % 16 == 0 &&
      d->ln->xws_bytes >= gemm_ln_ws_bytes(d->M, d->N) && ((uintptr_t)d->ln->xws % 16) == 0) {
    // LayerNorm inside the launch (gemm12.hip): offered to the launch; the launcher keeps it only for a single-round loader-wave launch and reports through `done`
    // OFF by default: built, parity-tested (tests/test_gemm_gpu.py::test_layernorm_inside_the_mmt_size_launch) and measured SLOWER than the two launches it replaces
    // -- O-projection 22.3 + 10.3 us as two launches, 72.7 us as one; 41 us with the waiting switched off (profiles/r6_gemm_experiments.txt #16): the rows'
    // statistics cross XCDs, i.e. four dependent memory-side round trips (store, count in, poll, load) of ~2 us each on a CU that has nothing else to run, and
    // 1952 waves issuing them at the same moment.  SAM_GEMM_LN_FUSE=1 selects it (read per call: the tests switch it).
    const char* e = getenv("SAM_GEMM_LN_FUSE");
    const int lnx = e ? atoi(e) : 0;
    const sam_ln_fuse* l = d->ln;
    if (lnx) {
      SAM_REQUIRE(l->gamma && l->beta && l->y && l->mean && l->rstd && l->ldy >= d->N && l->ldy % 4 == 0, "sam_gemm_bf16: incomplete sam_ln_fuse");
      a.ln_gamma = l->gamma; a.ln_beta = l->beta; a.ln_eps = l->eps; a.ln_y = (bf16_t*)l->y; a.ln_ldy = l->ldy; a.ln_mean = l->mean; a.ln_rstd = l->rstd;
      a.ln_ws = l->xws; a.ln_done = const_cast<int32_t*>(&l->done);
    }
  }
  const int64_t wsb = d->ws_bytes;
  const int ft = d->force_tile;
  SAM_REQUIRE(ft == 0 || ft == 64 || ft == 128 || ft == 160 || ft == 192 || ft == 256 || ft == 1192 || ft == 1256 || ft == 1448 || ft == 3192 || ft == 1128 || ft == 12192 || ft == 12448, "sam_gemm_bf16: force_tile must be 0, 64, 128, 160, 192, 256, 1128, 1192, 1256, 1448, 3192, 12192 or 12448");
  const int lay = (d->a_kcontig ? 2 : 0) | (d->b_kcontig ? 1 : 0);
  const int e = d->epilogue;
  if (lay == 3) {  // forward: x[M,K] . W[N,K]^T
    if (d->c_is_f32) {
      if (e == SAM_EPI_NONE) return launch<true, true, SAM_EPI_NONE, float>(a, st, want_split, wsb, ft);
      if (e == SAM_EPI_BIAS) return launch<true, true, SAM_EPI_BIAS, float>(a, st, want_split, wsb, ft);
    } else {
      if (e == SAM_EPI_NONE) return launch<true, true, SAM_EPI_NONE, bf16_t>(a, st, want_split, wsb, ft);
      if (e == SAM_EPI_BIAS) return launch<true, true, SAM_EPI_BIAS, bf16_t>(a, st, want_split, wsb, ft);
      if (e == SAM_EPI_BIAS_GELU) return launch<true, true, SAM_EPI_BIAS_GELU, bf16_t>(a, st, want_split, wsb, ft);
      if (e == SAM_EPI_BIAS_GELU_GRAD) return launch<true, true, SAM_EPI_BIAS_GELU_GRAD, bf16_t>(a, st, want_split, wsb, ft);
      if (e == SAM_EPI_BIAS_DROPOUT_RES) return launch<true, true, SAM_EPI_BIAS_DROPOUT_RES, bf16_t>(a, st, want_split, wsb, ft);
    }
  } else if (lay == 2) {  // dgrad: dy[M,N'] . W[N',K']
    if (!d->c_is_f32) {
      if (e == SAM_EPI_NONE) return launch<true, false, SAM_EPI_NONE, bf16_t>(a, st, want_split, wsb, ft);
      if (e == SAM_EPI_DGELU) return launch<true, false, SAM_EPI_DGELU, bf16_t>(a, st, want_split, wsb, ft);
      if (e == SAM_EPI_MUL_AUX) return launch<true, false, SAM_EPI_MUL_AUX, bf16_t>(a, st, want_split, wsb, ft);
      if (e == SAM_EPI_BIAS_DROPOUT_RES) return launch<true, false, SAM_EPI_BIAS_DROPOUT_RES, bf16_t>(a, st, want_split, wsb, ft);
    } else if (e == SAM_EPI_NONE) return launch<true, false, SAM_EPI_NONE, float>(a, st, want_split, wsb, ft);
  } else if (lay == 0) {  // wgrad: dy[rows,M]^T . x[rows,N]
    if (d->c_is_f32 && e == SAM_EPI_NONE) return launch<false, false, SAM_EPI_NONE, float>(a, st, want_split, wsb, ft);
    if (!d->c_is_f32 && e == SAM_EPI_NONE) return launch<false, false, SAM_EPI_NONE, bf16_t>(a, st, want_split, wsb, ft);
  }
  sam_set_error("sam_gemm_bf16: no kernel for layout (a_kcontig=%d,b_kcontig=%d) epilogue=%d c_is_f32=%d", d->a_kcontig, d->b_kcontig, e, d->c_is_f32);
  return SAM_ERR_UNSUPPORTED;
}
