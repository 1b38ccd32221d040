// Grouped weight-gradient GEMM on the LOADER-WAVE core (gfx950): dW_p (+)= dy_p^T x_p and db_p (+)= colsum(dy_p) for up to 20 problems in ONE persistent launch --
// the weight gradients of an MMT layer pair (sam/sa_m4c.py:554-560,653,678-680 in the backward direction), optionally with TextBert's three layers and the heads.
//
// Why a second grouped kernel (csrc/gemm8w.hip stays, for problem sets this one declines): the 8-wave kernel runs ONE 256 x 256 tile per CU -- a layer pair is 216
// tiles, 40 of the 256 CUs idle -- and every compute wave issues and waits for its own share of the operand DMA (1.74 us per k-tile).  Here
//   * the loop is gemm12.hip's: 8 compute waves + 4 loader waves per workgroup, 192 x 256 tiles (three waves per SIMD leave 168 registers): 1.03 us per k-tile,
//     1.37 us per 256 x 256 equivalent (tools/bench_wgrad12.py);
//   * a pair is 288 tiles of 182 k-tiles.  One workgroup per CU walks a STATIC schedule: first whole tiles, one per CU (256 of them), then ONE SLICE of the K range
//     of one of the 32 tiles that are left (8 slices per tile: 22-23 k-tiles each), then the shallow tiles of the set (TextBert's 432 tiles of 20 k-tiles, the heads'),
//     dealt round robin -- every CU gets 182 + 22.75 (+ 20..40) k-tiles.  The loader waves stream across the items: the next item's first operands land while an
//     epilogue runs;
//   * the eight slices of a tile are summed INSIDE the launch in a fixed order (bit-reproducible, no atomics on data): every wave ships its 24 accumulator
//     fragments to a workspace slot (write-through stores, then a counter), waits for the same wave of the other seven workgroups, and reduces THREE of the 24
//     fragments over the eight slots -- each slice workgroup finishes an eighth of the tile.  No workgroup barrier is involved (the loader waves own the
//     barrier schedule); waits are bounded and raise the workspace's error word instead of hanging.
// Both operands are k-strided (the contraction index R = B*N token rows is the slow index of dy [R, M] and x [R, N]): gemm8's panel images and
// ds_read_b64_tr_b16 fragments.
#include "gemm8_dev.h"
#include <stdlib.h>

using namespace samgemm;
using namespace samgemm8;
namespace {

constexpr int BM = 192, BN = 256;
constexpr int TM = BM / 32, TN = BN / 64, RB = TM / 2;            // 6 x 4 fragments per compute wave (96 rows x 64 columns)
constexpr int SAL = BM / 32, SBL = BN / 32;                        // 1 KB DMA pieces per loader wave and operand tile
constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, STAGE = A_BYTES + B_BYTES, NST = 2;
constexpr int FRAGS = TM * TN;                                      // 24
constexpr int SLOT_FLOATS = FRAGS * 8 * 64 * 4 + BM;                // a workgroup's accumulators, lane-linear, + its 192 bias-gradient partials
constexpr int MAXP = samgemm::SAM_MAX_GROUP8, MAXG = 320;          // MAXG: workgroups (= CUs) of a launch

struct WProb {
  const bf16_t* A; int64_t lda;      // dy [K rows, M]
  const bf16_t* B; int64_t ldb;      // x  [K rows, N]
  float* C; int64_t ldc;             // dW [M, N] fp32
  float* bias_grad;                  // [M] or NULL
  int M, N, K, tiles_m, tiles_n, accumulate;
};
struct WArgs {
  WProb p[MAXP];
  int tile_start[MAXP + 1];          // problems sorted deepest first
  int count, dbg;
  int n_deep, n_total;               // tiles of the deepest problems (all of one K) / all tiles
  int rounds1, R, S, kt_deep;        // whole-tile rounds of the deep tiles, tiles left over, slices per left-over tile, k-tiles of a deep tile
  short slice_of[MAXG];              // block -> tl + R * sl of its K slice in round `rounds1`, -1: none (the block goes straight on to the shallow tiles).  Blocks whose
                                     // whole tiles carried a bias gradient get none when the rest can take all slices: their loader waves' extra work made them
                                     // ~15 % slower per k-tile, about the length of a slice -- they finish level with the others (experiment 13)
  float* slots;                      // [R][S] x SLOT_FLOATS
  unsigned* cnt;                     // [R][8 waves][2] arrive / done, then [R][4 loader waves] bias arrivals; zero between launches
  unsigned* err;                     // error word of the workspace
};

struct Item { WProb P; int m0, n0, kt0, nkt, kind, tl, sl; };     // kind 0: whole tile, 1: slice sl of left-over tile tl; P: the tile's problem (a COPY picked with
                                                                   // static indices: indexing the kernel-argument array with a run-time value sends it to scratch)

__device__ __forceinline__ int xcd_run(int b, int seglen, int G) {            // index inside a segment of `seglen` tiles dealt to the XCDs in contiguous runs; -1: none
  const int q = seglen / 8, r = seglen % 8, xcd = b % 8, loc = b / 8;
  if (loc >= q + (xcd < r ? 1 : 0)) return -1;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
}

// The schedule reads the argument block where it lives -- the kernel-argument segment (constant address space: scalar loads with a run-time offset) -- through
// WPtr: indexing the by-value parameter itself with a run-time problem number makes the compiler copy the 1.3 KB table to scratch, a static select over its 20
// entries spills 600 SGPRs.
typedef const WArgs __attribute__((address_space(4)))* WPtr;
__device__ __forceinline__ WProb load_prob(WPtr wp, int q) {
  WProb P;
  P.A = wp->p[q].A; P.lda = wp->p[q].lda; P.B = wp->p[q].B; P.ldb = wp->p[q].ldb; P.C = wp->p[q].C; P.ldc = wp->p[q].ldc; P.bias_grad = wp->p[q].bias_grad;
  P.M = wp->p[q].M; P.N = wp->p[q].N; P.K = wp->p[q].K; P.tiles_m = wp->p[q].tiles_m; P.tiles_n = wp->p[q].tiles_n; P.accumulate = wp->p[q].accumulate;
  return P;
}
__device__ __forceinline__ bool get_item(WPtr wp, const WArgs& w, int j, int b, int G, Item& it) {
  int gt;
  it.kind = 0; it.tl = 0; it.sl = 0;
  const int e = w.R > 0 ? (int)wp->slice_of[b] : -1;
  if (w.R > 0 && e < 0 && j >= w.rounds1) ++j;          // no slice for this block: its item sequence skips the slice round
  if (j < w.rounds1) {
    gt = j * G + (b % 8) * (G / 8) + b / 8;
  } else if (w.R > 0 && j == w.rounds1) {
    it.tl = e % w.R; it.sl = e / w.R;
    it.kind = 1;
    gt = w.rounds1 * G + it.tl;
  } else {
    const int js = j - w.rounds1 - (w.R > 0 ? 1 : 0), base = w.n_deep + js * G, rem = w.n_total - base;
    if (rem <= 0) return false;
    const int idx = xcd_run(b, rem < G ? rem : G, G);
    if (idx < 0) return false;
    gt = base + idx;
  }
  int pi = 0;
  for (int q = 1; q < w.count; ++q)
    if (gt >= wp->tile_start[q]) pi = q;
  it.P = load_prob(wp, pi);
  const WProb& P = it.P;
  const int tile = gt - wp->tile_start[pi];
  const bool m_fast = P.tiles_n > P.tiles_m;       // walk so that a run of tiles partitions the wider operand
  it.m0 = (m_fast ? tile % P.tiles_m : tile / P.tiles_n) * BM;
  it.n0 = (m_fast ? tile / P.tiles_m : tile % P.tiles_n) * BN;
  const int KT = P.K / BK;
  if (it.kind == 1) {
    const int per = (KT + w.S - 1) / w.S;
    it.kt0 = it.sl * per;
    it.nkt = min(KT, it.kt0 + per) - it.kt0;
  } else {
    it.kt0 = 0; it.nkt = KT;
  }
  return true;
}

template <int S0, int S1>
__device__ __forceinline__ void dma_range(const bf16_t* base, unsigned char* dst, const unsigned* off, unsigned soff) {
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, 0x7fffffff, 0x00020000);
#pragma unroll
  for (int s = S0; s < S1; ++s)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(dst + s * 1024), 16, off[s], soff, 0, 0);
}

constexpr int AUX_SC1 = 0x10;
typedef unsigned int vu4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc_of(const void* p) { return __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, 0x7fffffff, 0x00020000); }

// bounded wait for an agent-scope counter to reach `want` (every lane polls the same word: one request per wave and trip)
__device__ __forceinline__ void wait_count(const unsigned* c, unsigned want, unsigned* err) {
  unsigned spins = 0;
  while (__hip_atomic_load(c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
    __builtin_amdgcn_s_sleep(2);
    if (++spins > (1u << 22)) { __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
  }
}

__global__ __launch_bounds__(768, 3) void gemm12w_kernel(WArgs w) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), i = lane & 15, g = lane >> 4;
  const int G = gridDim.x, b = blockIdx.x;
  const WPtr wp = (WPtr)__builtin_amdgcn_kernarg_segment_ptr();

  if (wave >= 8) {
    // =============================================================== loader waves (the barrier sequence of the upper compute row group, gemm12.hip)
    const int lw = wave - 8;
    unsigned offA[SAL], offB[SBL];
    const bf16_t *baseA = nullptr, *baseB = nullptr;
    unsigned kstepA = 0, kstepB = 0;
    int ja = 0, ka = 0, nka = 0, k0a = 0, jb = 0, kb = 0, nkb = 0, k0b = 0, sa_ = 0, sb_ = 0;
    bool va, vb;
    Item it;
    auto load_a = [&]() {
      va = get_item(wp, w, ja, b, G, it);
      if (va) {
        const WProb& P = it.P;
        baseA = P.A; kstepA = (unsigned)(BK * P.lda * 2); nka = it.nkt; k0a = it.kt0; ka = 0;
        src_offsets<false, SAL>(offA, P.lda, it.m0, P.M, lw, lane);
      }
    };
    auto load_b = [&]() {
      vb = get_item(wp, w, jb, b, G, it);
      if (vb) {
        const WProb& P = it.P;
        baseB = P.B; kstepB = (unsigned)(BK * P.ldb * 2); nkb = it.nkt; k0b = it.kt0; kb = 0;
        src_offsets<false, SBL>(offB, P.ldb, it.n0, P.N, lw, lane);
      }
    };
    load_a(); load_b();
#define SAM_LDMA_A(H)                                                                                                              \
  do {                                                                                                                             \
    dma_range<(H) * (SAL / 2), ((H) + 1) * (SAL / 2)>(baseA, smem + sa_ * STAGE + lw * (SAL * 1024), offA, (k0a + ka) * kstepA);   \
    if (H) { sa_ ^= 1; if (++ka == nka) { ++ja; load_a(); } }                                                                     \
  } while (0)
#define SAM_LDMA_B(H)                                                                                                              \
  do {                                                                                                                             \
    dma_range<(H) * (SBL / 2), ((H) + 1) * (SBL / 2)>(baseB, smem + sb_ * STAGE + A_BYTES + lw * (SBL * 1024), offB, (k0b + kb) * kstepB); \
    if (H) { sb_ ^= 1; if (++kb == nkb) { ++jb; load_b(); } }                                                                     \
  } while (0)
    // prologue: k-tile 0 complete, B of k-tile 1 in flight (every block has at least one item: the host keeps G <= number of first-round items)
    SAM_LDMA_A(0); SAM_LDMA_A(1); SAM_LDMA_B(0); SAM_LDMA_B(1);
    if (vb) { SAM_LDMA_B(0); SAM_LDMA_B(1); vmwait<SBL>(); }
    else vmwait<0>();
    __builtin_amdgcn_s_barrier();
    // the compute side's position: item jc (cur), k-tiles left in it, LDS stage of its current k-tile
    int jc = 0, left = 0, su = 0;
    Item cur;
    get_item(wp, w, 0, b, G, cur);
    left = cur.nkt;
    // Bias gradient = column sums of dy = sum over k of the A tile, taken HERE: the loader waves are idle most of a k-tile, the A tile is in LDS anyway, and
    // the compute waves have no register to spare (an accumulator and a selector operand for a 25th MFMA put them over 168 and into scratch inside the loop).
    // Loader lw owns the three 16-column fragment rows [48 lw, 48 lw + 48) of the tile's 192 and sums them on the MATRIX pipe: D = ones x A^T, six MFMAs per k-tile
    // (lane (i, g) ends up with the sum of column 16 x + i in every register of accb[x]).  (The first version summed on the VALU -- 6 ds_read_b128 + ~100
    // unpack / add instructions per k-tile: the CUs whose tile carries a bias gradient ran 16 % slower and set the launch's time: 341 -> 428 us per layer pair.)
    // Only tiles of a problem's first tile column carry it.
    f32x4 accb[3];
#pragma unroll
    for (int x = 0; x < 3; ++x) accb[x] = f32x4{0.f, 0.f, 0.f, 0.f};
    bf16x8 ones;
#pragma unroll
    for (int e = 0; e < 8; ++e) ones[e] = (bf16_t)0x3F80;
    const int bi = lane & 15, bg = lane >> 4, bsig = ((bi >> 3) & 1) | ((bg & 1) << 1);
    bf16x8 fa[3][2];
    bool pend = false;
    bool cbias = cur.P.bias_grad != nullptr && cur.n0 == 0;
    for (;;) {
      const bool issue_a = va;
      if (issue_a) SAM_LDMA_A(0);
      __builtin_amdgcn_s_barrier();
      if (issue_a) SAM_LDMA_A(1);
      // the bias gradient's fragments of THIS k-tile are requested here and multiplied one k-tile later (at the item's end for the last one): the loader never waits
      // for an LDS read in front of a barrier.  (Measured per layer pair with bias gradients, 339 us without: VALU sums 428; MFMAs right behind the reads, all in
      // this interval 394, half of them in the interval before 464.)
      if (cbias) {
        if (pend) {
#pragma unroll
          for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int x = 0; x < 3; ++x) accb[x] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, fa[x][ks], accb[x], 0, 0, 0);
        }
        const unsigned char* stA = smem + su * STAGE;
#pragma unroll
        for (int x = 0; x < 3; ++x)
#pragma unroll
          for (int ks = 0; ks < 2; ++ks) fa[x][ks] = frag<false>(stA, (3 * lw + x) * 16, ks, bi, bg, bsig);
        pend = true;
      }
      su ^= 1;
      __builtin_amdgcn_s_barrier();
      const bool issue_b = vb;
      if (issue_b) SAM_LDMA_B(0);
      __builtin_amdgcn_s_barrier();
      if (issue_b) { SAM_LDMA_B(1); vmwait<SBL>(); }
      else vmwait<0>();
      __builtin_amdgcn_s_barrier();
      if (--left == 0) {
        __builtin_amdgcn_s_barrier();        // the upper group's alignment barrier at an item's end
        if (cbias) {
          if (pend) {          // the last k-tile's fragments
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
              for (int x = 0; x < 3; ++x) accb[x] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, fa[x][ks], accb[x], 0, 0, 0);
            pend = false;
          }
          // column c = 16 x + i of this loader's 48 is in lane i of accb[x]: lane c takes it
          float mine = 0.f;
#pragma unroll
          for (int x = 0; x < 3; ++x) {
            const float t = __shfl(accb[x][0], lane & 15);
            if ((lane >> 4) == x) mine = t;
            accb[x] = f32x4{0.f, 0.f, 0.f, 0.f};
          }
          const int m = cur.m0 + 48 * lw + lane;
          if (cur.kind == 0) {
            if (lane < 48 && m < cur.P.M) cur.P.bias_grad[m] = (cur.P.accumulate ? cur.P.bias_grad[m] : 0.f) + mine;
          } else {
            // a slice: the partial goes to the slot, the same loader wave of slice 0 adds the S partials in order
            float* const slot0 = w.slots + (int64_t)cur.tl * w.S * SLOT_FLOATS;
            if (lane < 48) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(mine), rsrc_of(slot0 + (int64_t)cur.sl * SLOT_FLOATS), (FRAGS * 8 * 64 * 4 + 48 * lw + lane) * 4, 0, AUX_SC1);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            unsigned* const barrive = w.cnt + (int64_t)w.R * 16 + cur.tl * 4 + lw;
            if (lane == 0) __hip_atomic_fetch_add(barrive, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (cur.sl == 0) {
              wait_count(barrive, (unsigned)w.S, w.err);
              float s4 = 0.f;
              for (int t = 0; t < w.S; ++t)
                s4 += __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsrc_of(slot0 + (int64_t)t * SLOT_FLOATS), (FRAGS * 8 * 64 * 4 + 48 * lw + (lane < 48 ? lane : 0)) * 4, 0, AUX_SC1));
              if (lane < 48 && m < cur.P.M) cur.P.bias_grad[m] = (cur.P.accumulate ? cur.P.bias_grad[m] : 0.f) + s4;
              asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
              if (lane == 0) __hip_atomic_store(barrive, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
          }
        }
        if (!get_item(wp, w, ++jc, b, G, cur)) break;
        left = cur.nkt;
        cbias = cur.P.bias_grad != nullptr && cur.n0 == 0;
      }
    }
#undef SAM_LDMA_A
#undef SAM_LDMA_B
    return;
  }

  // =============================================================== compute waves
  const int wr = wave >> 2, wc = wave & 3;
  const int sig = ((i >> 3) & 1) | ((g & 1) << 1);

  Item it;
  bool have = get_item(wp, w, 0, b, G, it);
  f32x4 acc[TN][TM];
#pragma unroll
  for (int a = 0; a < TN; ++a)
#pragma unroll
    for (int c = 0; c < TM; ++c) acc[a][c] = f32x4{0.f, 0.f, 0.f, 0.f};
  __builtin_amdgcn_s_barrier();
  if (wr == 1) __builtin_amdgcn_s_barrier();      // lower row group: one barrier behind from here on

  bf16x8 af[RB][2], bfr[TN][2];
  int su = 0, j = 0;
  while (have) {
    const WProb& P = it.P;
    for (int u = 0; u < it.nkt; ++u) {
      const unsigned char* stA = smem + su * STAGE;
      const unsigned char* stB = stA + A_BYTES;
      su ^= 1;
      // ---- phase 0
#pragma unroll
      for (int x = 0; x < TN; ++x)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) bfr[x][ks] = frag<false>(stB, wc * (BN / 4) + x * 16, ks, i, g, sig);
#pragma unroll
      for (int x = 0; x < RB; ++x)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) af[x][ks] = frag<false>(stA, wr * (BM / 2) + x * 16, ks, i, g, sig);
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
      // ---- phase 1
#pragma unroll
      for (int x = 0; x < RB; ++x)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) af[x][ks] = frag<false>(stA, wr * (BM / 2) + (RB + x) * 16, ks, i, g, sig);
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
    // ================= end of an item
    if (wr == 0) __builtin_amdgcn_s_barrier();           // both row groups in step for the epilogue
    const int mw = it.m0 + wr * (BM / 2), nw = it.n0 + wc * (BN / 4);
    if (w.dbg & 1) {
      float sacc = 0.f;
#pragma unroll
      for (int a = 0; a < TN; ++a)
#pragma unroll
        for (int c = 0; c < TM; ++c) sacc += acc[a][c][0] + acc[a][c][1] + acc[a][c][2] + acc[a][c][3];
      if (sacc == 12345.678f) P.C[tid] = 1.f;
    } else if (it.kind == 0) {
      const bool full = it.m0 + BM <= P.M && it.n0 + BN <= P.N;
      GemmArgs ea = {};
      ea.M = P.M; ea.N = P.N;
      gemm_epilogue8<TM, TN, SAM_EPI_NONE, float, 0, RB>(ea, acc, mw, nw, full, P.C, P.ldc, P.accumulate, i, g);
      gemm_epilogue8<TM, TN, SAM_EPI_NONE, float, RB, TM>(ea, acc, mw, nw, full, P.C, P.ldc, P.accumulate, i, g);
    } else {
      // ---- slice: ship all fragments, wait for the same wave of the other slices, finish three fragments of the tile
      const int S = w.S;
      float* const slot0 = w.slots + (int64_t)it.tl * S * SLOT_FLOATS;
      {
        const __amdgpu_buffer_rsrc_t rs = rsrc_of(slot0 + (int64_t)it.sl * SLOT_FLOATS);
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
#pragma unroll
          for (int tm = 0; tm < TM; ++tm) {
            const f32x4 v = acc[tn][tm];
            const vu4 uu = {__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3])};
            __builtin_amdgcn_raw_buffer_store_b128(uu, rs, (((tn * TM + tm) * 8 + wave) * 64 + lane) * 16, 0, AUX_SC1);
          }
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      unsigned* const arrive = w.cnt + ((int64_t)it.tl * 8 + wave) * 2;
      if (lane == 0) __hip_atomic_fetch_add(arrive, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      wait_count(arrive, (unsigned)S, w.err);
      // fragments [f0, f0 + nf) of the tile are this slice's to finish (24 fragments over S slices; S = 8: three each)
      const int per = (FRAGS + S - 1) / S, f0 = it.sl * per, f1 = min(FRAGS, f0 + per);
      for (int f = f0; f < f1; ++f) {
        f32x4 sum = {0.f, 0.f, 0.f, 0.f};
        for (int t = 0; t < S; ++t) {                      // fixed order: slice 0, 1, ..., S - 1 (the k ranges in ascending order)
          const vu4 uu = __builtin_amdgcn_raw_buffer_load_b128(rsrc_of(slot0 + (int64_t)t * SLOT_FLOATS), ((f * 8 + wave) * 64 + lane) * 16, 0, AUX_SC1);
          sum += f32x4{__uint_as_float(uu[0]), __uint_as_float(uu[1]), __uint_as_float(uu[2]), __uint_as_float(uu[3])};
        }
        const int tn = f / TM, tm = f - tn * TM;
        const int m = mw + tm * 16 + i, n = nw + tn * 16 + 4 * g;
        if (m < P.M && n < P.N) {
          float4* dst = reinterpret_cast<float4*>(P.C + (int64_t)m * P.ldc + n);
          float4 o = P.accumulate ? *dst : make_float4(0.f, 0.f, 0.f, 0.f);
          o.x += sum[0]; o.y += sum[1]; o.z += sum[2]; o.w += sum[3];
          *dst = o;
        }
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the slot reads are done before this wave reports so
      if (lane == 0) {
        const unsigned d = __hip_atomic_fetch_add(arrive + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (d == (unsigned)S - 1) {                         // the last of the S readers: the next launch finds both words at zero again
          __hip_atomic_store(arrive, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          __hip_atomic_store(arrive + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
    }
    // ---- next item
    have = get_item(wp, w, ++j, b, G, it);
    if (have) {
      if (wr == 1) __builtin_amdgcn_s_barrier();
#pragma unroll
      for (int a = 0; a < TN; ++a)
#pragma unroll
        for (int c = 0; c < TM; ++c) acc[a][c] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
  }
}

}  // namespace

namespace samgemm {

// workspace of the slice exchange for a set with `r` left-over tiles of `s` slices: [64 words: word 0 = error][counters, padded to 64 words][slots]
static int64_t ws12_bytes(int r, int s) { return (int64_t)r * s * SLOT_FLOATS * 4 + (int64_t)(64 + ((r * 20 + 63) / 64) * 64) * 4 + 256; }

// the schedule of a problem set on `n_cu` CUs; false: not a set for this kernel.  slice_of (may be NULL): the block -> slice table of WArgs; S_ws: the largest
// slice count any table for this set may use (sizes the workspace: the bias-aware table depends on which problems carry a bias gradient at launch time)
static bool plan12(const sam_gemm_desc* descs, int count, int n_cu, int* order, int& n_deep, int& n_total, int& rounds1, int& R, int& S, int& kt_deep, short* slice_of,
                   int& S_ws) {
  if (count < 1 || count > MAXP || n_cu % 8 != 0 || n_cu > MAXG) return false;
  int max_k = 0;
  for (int q = 0; q < count; ++q) { order[q] = q; max_k = descs[q].K > max_k ? descs[q].K : max_k; }
  for (int a = 1; a < count; ++a)
    for (int b = a; b > 0 && descs[order[b]].K > descs[order[b - 1]].K; --b) { const int t = order[b]; order[b] = order[b - 1]; order[b - 1] = t; }
  n_deep = n_total = 0;
  int tstart[MAXP + 1];
  for (int q = 0; q < count; ++q) {
    const sam_gemm_desc* d = descs + order[q];
    if (d->K % BK != 0 || d->K < BK || d->M % 8 != 0 || d->N % 8 != 0) return false;
    if ((int64_t)d->K * d->lda * 2 >= (int64_t)0x7fffffff || (int64_t)d->K * d->ldb * 2 >= (int64_t)0x7fffffff) return false;
    const int t = ((d->M + BM - 1) / BM) * ((d->N + BN - 1) / BN);
    tstart[q] = n_total;
    n_total += t;
    if (d->K == max_k) n_deep += t;
  }
  tstart[count] = n_total;
  kt_deep = max_k / BK;
  rounds1 = n_deep / n_cu;
  R = n_deep - rounds1 * n_cu;
  S = S_ws = 0;
  if (rounds1 < 1) return false;                              // fewer deep tiles than CUs: the pair-exchange kernel (gemm8w.hip) does better
  if (n_total > 16 * n_cu) return false;
  if (R == 0) return true;
  auto valid = [&](int s_) { const int per = (kt_deep + s_ - 1) / s_; return s_ >= 2 && s_ <= FRAGS && kt_deep >= 2 * s_ && (s_ - 1) * per < kt_deep; };   // (every slice needs a k-tile)
  if (n_cu % R != 0) return false;
  S = S_ws = n_cu / R;
  if (!valid(S)) return false;
  if (!slice_of) return true;
  // every block takes a slice: XCD x gets R / 8 of the left-over tiles with all their slices (tiles next to each other share operand panels slice by slice)
  const bool by_xcd = (R & 7) == 0;
  const int rpx = R / 8;
  for (int b = 0; b < n_cu; ++b) {
    const int xcd = b % 8, loc = b / 8;
    const int tl = by_xcd ? xcd * rpx + loc % rpx : b % R, sl = by_xcd ? loc / rpx : b / R;
    slice_of[b] = (short)(tl + R * sl);
  }
  // bias-aware table: which blocks' whole tiles carry a bias gradient (the tile walk of get_item, rounds 0 .. rounds1 - 1)
  static int aware = -1;
  if (aware < 0) { const char* e = getenv("SAM_GEMM12W_BIAS_AWARE"); aware = e ? atoi(e) : 1; }
  if (!aware || !by_xcd) return true;
  bool has_bias[MAXG];
  int free_x[8] = {0, 0, 0, 0, 0, 0, 0, 0}, n_bias = 0;
  for (int b = 0; b < n_cu; ++b) {
    has_bias[b] = false;
    for (int j = 0; j < rounds1; ++j) {
      const int gt = j * n_cu + (b % 8) * (n_cu / 8) + b / 8;
      int pi = 0;
      for (int q = 1; q < count; ++q)
        if (gt >= tstart[q]) pi = q;
      const sam_gemm_desc* d = descs + order[pi];
      const int tm = (d->M + BM - 1) / BM, tn = (d->N + BN - 1) / BN, tile = gt - tstart[pi];
      const int n0 = (tn > tm ? tile / tm : tile % tn) * BN;
      if (d->bias_grad && n0 == 0) has_bias[b] = true;
    }
    if (has_bias[b]) ++n_bias; else ++free_x[b % 8];
  }
  int s2 = FRAGS;
  for (int x = 0; x < 8; ++x) s2 = free_x[x] / rpx < s2 ? free_x[x] / rpx : s2;
  // worth it when a slice stays about as long as what the bias gradient costs a block (~15 % of a tile): at least five slices per tile
  if (n_bias == 0 || s2 < 5 || s2 >= S || !valid(s2)) return true;
  int k_x[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int b = 0; b < n_cu; ++b) {
    const int xcd = b % 8;
    if (has_bias[b] || k_x[xcd] >= rpx * s2) { slice_of[b] = -1; continue; }
    const int k = k_x[xcd]++;
    slice_of[b] = (short)(xcd * rpx + k % rpx + R * (k / rpx));
  }
  S = s2;
  return true;
}

int64_t gemm12w_ws_bytes(const sam_gemm_desc* descs, int count) {
  int order[MAXP], n_deep, n_total, rounds1, R, S, ktd, s_ws;
  if (!plan12(descs, count, grid_cu_count(), order, n_deep, n_total, rounds1, R, S, ktd, nullptr, s_ws) || R == 0) return 0;
  return ws12_bytes(R, s_ws);
}

// returns SAM_ERR_UNSUPPORTED when the set is not one for this kernel (the caller goes on to gemm8w_grouped / the 4-wave kernel)
int gemm12w_grouped(const sam_gemm_desc* descs, int count, hipStream_t st) {
  const int n_cu = grid_cu_count();
  int order[MAXP];
  WArgs w = {};
  int s_ws = 0;
  if (!plan12(descs, count, n_cu, order, w.n_deep, w.n_total, w.rounds1, w.R, w.S, w.kt_deep, w.slice_of, s_ws)) return SAM_ERR_UNSUPPORTED;
  w.count = count;
  int tiles = 0;
  for (int q = 0; q < count; ++q) {
    const sam_gemm_desc* d = descs + order[q];
    WProb& p = w.p[q];
    p.A = (const bf16_t*)d->A; p.lda = d->lda; p.B = (const bf16_t*)d->B; p.ldb = d->ldb; p.C = (float*)d->C; p.ldc = d->ldc; p.bias_grad = d->bias_grad;
    p.M = d->M; p.N = d->N; p.K = d->K; p.accumulate = d->accumulate ? 1 : 0;
    p.tiles_m = (d->M + BM - 1) / BM; p.tiles_n = (d->N + BN - 1) / BN;
    w.tile_start[q] = tiles;
    tiles += p.tiles_m * p.tiles_n;
  }
  w.tile_start[count] = tiles;
  const sam_gemm_desc* d0 = descs;
  if (!d0->ws || ((uintptr_t)d0->ws % 16) != 0) return SAM_ERR_UNSUPPORTED;
  w.err = reinterpret_cast<unsigned*>(d0->ws);
  if (w.R > 0) {
    if (d0->ws_bytes < ws12_bytes(w.R, s_ws)) return SAM_ERR_UNSUPPORTED;
    w.cnt = reinterpret_cast<unsigned*>(d0->ws) + 64;                                   // zero between launches (the caller zero-fills once)
    w.slots = d0->ws + 64 + ((w.R * 20 + 63) / 64) * 64;
  }
  { static int dbg = -1; if (dbg < 0) { const char* e = getenv("SAM_GEMM8W_DBG"); dbg = e ? atoi(e) : 0; } w.dbg = dbg; }
  constexpr size_t LDS = (size_t)NST * STAGE;
  static bool once = false;
  if (!once) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(gemm12w_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS);
    once = true;
  }
  gemm12w_kernel<<<dim3(n_cu), dim3(768), LDS, st>>>(w);
  SAM_LAUNCH_CHECK();
  return SAM_OK;
}

}  // namespace samgemm
