// PROBE (not part of the library): main loop of the round-2 GEMM core, stand-alone, so that schedule variants can be A/B-ed in one process.
//   C[M,N] (bf16) = A . B, 256x256x64 block tile, 8 waves as 2 (M) x 4 (N), 128x64 per wave, one block per CU.
//   LDS: 2 stages x 4 half-tiles (A0 A1 B0 B1, 128 rows x 64 k each = 16 KB), filled by buffer_load ... lds (no VGPR staging), XOR-swizzled
//   through the SOURCE address.  A k-tile is consumed in four phases (one 64x32 quadrant of the wave's tile each); every phase issues ONE
//   half-tile of a tile two ahead; the DMA queue is never drained inside the loop (counted vmcnt once per k-tile); the two wave groups
//   (upper / lower 128 rows) run one barrier apart so that on every SIMD one wave is in its MFMA segment while the other reads fragments.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I sam-textvqa_amd/csrc -I include tools/probes/gemm8_probe.hip -o tools/probes/gemm8_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>
#include <type_traits>
#include "common.h"

extern "C" void sam_set_error(const char*, ...) {}

typedef __attribute__((ext_vector_type(16))) float f32x16;
constexpr int BK = 64;

struct P { const bf16_t* A; const bf16_t* B; bf16_t* C; int M, N, K; int64_t lda, ldb, ldc; int tiles_m, tiles_n, group_m; };

// Every operand tile in LDS is a sequence of 128-byte rows, 8 rows per 1 KB DMA slice:
//   k-contiguous operand: row = m (or n) index, 64 k per row, 16-byte chunk c stored at c ^ ((row>>1)&7)          (ds_read_b128 fragments)
//   k-strided operand:    64-column panels, inside a panel row = k, 64 columns per row, chunk c stored at c ^ (sigma(k)<<1),
//                         sigma(k) = bit1(k) | bit3(k)<<1                                                          (ds_read_b64_tr_b16 fragments)
__device__ __forceinline__ int kc_off(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }
__device__ __forceinline__ int ks_sigma(int krow) { return ((krow >> 1) & 1) | (((krow >> 3) & 1) << 1); }

template <bool KC, int S>
__device__ __forceinline__ void src_offsets(unsigned (&off)[S], int64_t ld, int row0, int rows, int wave, int lane) {
#pragma unroll
  for (int s = 0; s < S; ++s) {
    const int j = wave * S + s;
    if (KC) {
      const int row = 8 * j + (lane >> 3), pos = lane & 7, c = pos ^ ((row >> 1) & 7);
      const int grow = min(row0 + row, rows - 1);
      off[s] = (unsigned)((grow * ld + c * 8) * 2);
    } else {
      const int panel = j >> 3, krow = 8 * (j & 7) + (lane >> 3), pos = lane & 7, c = pos ^ (ks_sigma(krow) << 1);
      const int col = min(row0 + panel * 64 + c * 8, rows - 8);
      off[s] = (unsigned)((krow * ld + col) * 2);
    }
  }
}

// slices [S0, S1) of this wave's share of an operand tile: global -> LDS, 1 KB per wave instruction, no VGPR staging
template <int S0, int S1>
__device__ __forceinline__ void dma_slices(const bf16_t* base, unsigned char* dst, const unsigned* off, unsigned soff) {
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, 0x7fffffff, 0x00020000);
#pragma unroll
  for (int s = S0; s < S1; ++s)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(dst + s * 1024), 16, off[s], soff, 0, 0);
}
template <int N>
__device__ __forceinline__ void vmwait() {
  if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  else if constexpr (N == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
  else if constexpr (N == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
  else if constexpr (N == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
  else if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  else if constexpr (N == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
  else static_assert(N < 6, "vmwait");
}
template <bool KC>
__device__ __forceinline__ bf16x8 frag(const unsigned char* region, int row, int ks, int i, int g, int sig) {
  if constexpr (KC) return *reinterpret_cast<const bf16x8*>(region + kc_off(row + i, 4 * ks + g));
  else {
    const int krow = 32 * ks + 8 * g + (i >> 2);
    const unsigned char* q = region + (row >> 6) * 8192 + krow * 128 + (((((row & 63) >> 3) + ((i & 3) >> 1)) ^ (sig << 1)) << 4) + (i & 1) * 8;
    return cat4(lds_read_tr16(q), lds_read_tr16(q + 512));
  }
}

template <int BM, int BN, int CB, bool AKC, bool BKC, int VAR>
__global__ __launch_bounds__(512, 2) void gemm8(P p) {
  constexpr bool STAGGER = (VAR & 1) != 0, PRIO = (VAR & 2) != 0;
  constexpr int TM = BM / 32, TN = BN / 64, SA = BM / 64, SB = BN / 64, RB = TM / 2;
  constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, STAGE = A_BYTES + B_BYTES;
  constexpr int C0 = CB == 2 ? (TN + 1) / 2 : TN;       // column fragments of the first column block
  constexpr int NPH = 2 * CB;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), i = lane & 15, g = lane >> 4;
  const int wr = wave >> 2, wc = wave & 3;
  const int nblk = p.tiles_m * p.tiles_n;
  int bid = blockIdx.x;
  {
    const int q = nblk / 8, r = nblk % 8, xcd = bid % 8, loc = bid / 8;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
  }
  const int per_group = p.group_m * p.tiles_n;
  const int group = bid / per_group, first_m = group * p.group_m;
  const int gsize = min(p.tiles_m - first_m, p.group_m);
  const int in_group = bid - group * per_group;
  const int m0 = (first_m + in_group % gsize) * BM, n0 = (in_group / gsize) * BN;

  unsigned offA[SA], offB[SB];
  src_offsets<AKC, SA>(offA, p.lda, m0, p.M, wave, lane);
  src_offsets<BKC, SB>(offB, p.ldb, n0, p.N, wave, lane);
  const unsigned kstepA = AKC ? BK * 2 : (unsigned)(BK * p.lda * 2), kstepB = BKC ? BK * 2 : (unsigned)(BK * p.ldb * 2);

#define ISSUE_A(t, s0, s1) dma_slices<s0, s1>(p.A, smem + ((t) & 1) * STAGE + wave * (SA * 1024), offA, (t) * kstepA)
#define ISSUE_B(t, s0, s1) dma_slices<s0, s1>(p.B, smem + ((t) & 1) * STAGE + A_BYTES + wave * (SB * 1024), offB, (t) * kstepB)
  constexpr int SA0 = CB == 2 ? (SA + 1) / 2 : SA, SB0 = CB == 2 ? (SB + 1) / 2 : SB;

  f32x4 acc[TN][TM];
#pragma unroll
  for (int a = 0; a < TN; ++a)
#pragma unroll
    for (int b = 0; b < TM; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int KT = p.K / BK;
  ISSUE_A(0, 0, SA); ISSUE_B(0, 0, SB);
  if (KT > 1) { ISSUE_B(1, 0, SB); vmwait<SB>(); }
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  if (STAGGER && wr == 1) __builtin_amdgcn_s_barrier();

  const int sig = ((i >> 3) & 1) | ((g & 1) << 1);     // sigma(krow) for krow = 32 ks + 8 g + (i >> 2)
  bf16x8 af[RB][2], bfr[TN][2];
  for (int t = 0; t < KT; ++t) {
    const unsigned char* stA = smem + (t & 1) * STAGE;
    const unsigned char* stB = stA + A_BYTES;
#pragma unroll
    for (int ph = 0; ph < NPH; ++ph) {
      // row block r and column fragments [cb, ce) of this phase
      const int r = CB == 2 ? (ph >> 1) : ph;
      const int cb = CB == 2 ? ((ph == 1 || ph == 2) ? C0 : 0) : 0, ce = CB == 2 ? ((ph == 1 || ph == 2) ? TN : C0) : TN;
      // ---- read segment
      if (ph == 0 || (CB == 2 && ph == 1)) {
#pragma unroll
        for (int x = 0; x < TN; ++x)
          if (x >= cb && x < ce)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) bfr[x][ks] = frag<BKC>(stB, wc * (BN / 4) + x * 16, ks, i, g, sig);
      }
      if (ph == 0 || ph == CB) {
#pragma unroll
        for (int x = 0; x < RB; ++x)
#pragma unroll
          for (int ks = 0; ks < 2; ++ks) af[x][ks] = frag<AKC>(stA, wr * (BM / 2) + (r * RB + x) * 16, ks, i, g, sig);
      }
      // ---- one DMA unit of a future k-tile
      if (CB == 2) {
        if (ph == 0) { if (t + 1 < KT) ISSUE_A(t + 1, 0, SA0); }
        else if (ph == 1) { if (t + 1 < KT) ISSUE_A(t + 1, SA0, SA); }
        else if (ph == 2) { if (t + 2 < KT) ISSUE_B(t + 2, 0, SB0); }
        else { if (t + 2 < KT) ISSUE_B(t + 2, SB0, SB); }
      } else {
        if (ph == 0) { if (t + 1 < KT) ISSUE_A(t + 1, 0, SA); }
        else { if (t + 2 < KT) ISSUE_B(t + 2, 0, SB); }
      }
      if (ph == NPH - 1) {   // all of tile t+1 must have landed; B of t+2 may stay in flight
        if (t + 2 < KT) vmwait<SB>();
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      // ---- MFMA segment
      if (PRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int x = 0; x < RB; ++x)
#pragma unroll
          for (int y = 0; y < TN; ++y)
            if (y >= cb && y < ce)
              acc[y][r * RB + x] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[y][ks], af[x][ks], acc[y][r * RB + x], 0, 0, 0);
      if (PRIO) __builtin_amdgcn_s_setprio(0);
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  if (STAGGER && wr == 0) __builtin_amdgcn_s_barrier();

#pragma unroll
  for (int tm = 0; tm < TM; ++tm) {
    const int m = m0 + wr * (BM / 2) + tm * 16 + i;
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
      const int n = n0 + wc * (BN / 4) + tn * 16 + 4 * g;
      if (m < p.M && n < p.N)
        *reinterpret_cast<uint2*>(p.C + (int64_t)m * p.ldc + n) = make_uint2(pack_bf16x2(acc[tn][tm][0], acc[tn][tm][1]), pack_bf16x2(acc[tn][tm][2], acc[tn][tm][3]));
    }
  }
}

// reference: one thread per output, fp32 accumulate
template <bool AKC, bool BKC>
__global__ void ref_kernel(P p, float* out) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x, m = blockIdx.y;
  if (n >= p.N) return;
  float s = 0.f;
  for (int k = 0; k < p.K; ++k) {
    const float a = bf2f(AKC ? p.A[(int64_t)m * p.lda + k] : p.A[(int64_t)k * p.lda + m]);
    const float b = bf2f(BKC ? p.B[(int64_t)n * p.ldb + k] : p.B[(int64_t)k * p.ldb + n]);
    s += a * b;
  }
  out[(int64_t)m * p.N + n] = s;
}
__global__ void cmp_kernel(const bf16_t* c, const float* ref, int64_t n, float* worst) {
  float w = 0.f;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float r = ref[i], v = bf2f(c[i]);
    w = fmaxf(w, fabsf(v - r) / (fabsf(r) + 1.0f));
  }
  atomicMax(reinterpret_cast<int*>(worst), __float_as_int(w));
}
__global__ void fill_kernel(bf16_t* p, int64_t n, unsigned seed, float scale) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    unsigned x = (unsigned)i * 2654435761u + seed; x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    p[i] = f2bf(((x & 0xffff) / 32768.0f - 1.0f) * scale);
  }
}

template <int BM, int BN, int CB, bool AKC, bool BKC, int VAR>
double run(const char* tag, P p, bool check, float* ref, float* worst_d) {
  auto k = gemm8<BM, BN, CB, AKC, BKC, VAR>;
  const size_t lds = 2 * (BM + BN) * 128;
  hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  p.tiles_m = (p.M + BM - 1) / BM; p.tiles_n = (p.N + BN - 1) / BN;
  const int grid = p.tiles_m * p.tiles_n;
  hipMemset(p.C, 0, (size_t)p.M * p.N * 2);
  for (int it = 0; it < 3; ++it) k<<<grid, 512, lds>>>(p);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 20;
  double best = 1e30;
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    for (int it = 0; it < iters; ++it) k<<<grid, 512, lds>>>(p);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    best = std::fmin(best, ms * 1e3 / iters);
  }
  float worst = -1.f;
  if (check) {
    hipMemset(worst_d, 0, 4);
    cmp_kernel<<<1024, 256>>>(p.C, ref, (int64_t)p.M * p.N, worst_d);
    hipMemcpy(&worst, worst_d, 4, hipMemcpyDeviceToHost);
  }
  std::printf("%-6s %3dx%3d cb%d var %d gm %2d  M=%5d N=%5d K=%5d  tiles=%4d  %7.1f us  %7.1f TFLOP/s  err %.4f %s (%s)\n", tag, BM, BN, CB, VAR, p.group_m, p.M, p.N, p.K, grid, best,
              2.0 * p.M * p.N * p.K / best / 1e6, worst, (!check || worst < 0.02f) ? "ok" : "FAIL", hipGetErrorString(hipGetLastError()));
  return best;
}

int main(int argc, char** argv) {
  const int MAXM = 11648, MAXD = 8192;
  bf16_t *dA, *dB, *dC; float *ref, *worst_d;
  const size_t na = (size_t)MAXD * MAXD, nc = (size_t)MAXD * MAXD;
  hipMalloc(&dA, na * 2); hipMalloc(&dB, na * 2); hipMalloc(&dC, nc * 2); hipMalloc(&ref, nc * 4); hipMalloc(&worst_d, 4);
  fill_kernel<<<2048, 256>>>(dA, na, 1u, 1.0f);
  fill_kernel<<<2048, 256>>>(dB, na, 77u, 0.1f);
  struct S { int M, N, K; };
  const S shapes[] = {{11648, 768, 768}, {11648, 2304, 768}, {11648, 3072, 768}, {11648, 768, 3072}, {11648, 768, 2304}, {4096, 4096, 4096}, {8192, 8192, 8192}};
  const bool quick = argc > 1 && !strcmp(argv[1], "quick");
  for (const S& s : shapes) {
    if (quick && s.M == 8192) continue;
    const bool check = s.M * (double)s.N * s.K < 1e11;
    P p{dA, dB, dC, s.M, s.N, s.K, s.K, s.K, s.N, 0, 0, 8};
    if (check) { ref_kernel<true, true><<<dim3((s.N + 255) / 256, s.M), 256>>>(p, ref); }
    run<256, 256, 2, true, true, 1>("fwd", p, check, ref, worst_d);
    run<256, 256, 1, true, true, 1>("fwd", p, check, ref, worst_d);
    run<192, 192, 2, true, true, 1>("fwd", p, check, ref, worst_d);
    run<192, 192, 1, true, true, 1>("fwd", p, check, ref, worst_d);
    run<192, 192, 1, true, true, 3>("fwd", p, check, ref, worst_d);
    run<256, 128, 1, true, true, 1>("fwd", p, check, ref, worst_d);
    run<128, 256, 1, true, true, 1>("fwd", p, check, ref, worst_d);
    run<128, 128, 1, true, true, 1>("fwd", p, check, ref, worst_d);
    P q{dA, dB, dC, s.M, s.N, s.K, s.K, s.N, s.N, 0, 0, 8};
    if (check) { ref_kernel<true, false><<<dim3((s.N + 255) / 256, s.M), 256>>>(q, ref); }
    run<256, 256, 2, true, false, 1>("dgrad", q, check, ref, worst_d);
    run<192, 192, 2, true, false, 1>("dgrad", q, check, ref, worst_d);
    run<192, 192, 1, true, false, 1>("dgrad", q, check, ref, worst_d);
    P w{dA, dB, dC, s.M, s.N, s.K, s.M, s.N, s.N, 0, 0, 8};
    if (check) { ref_kernel<false, false><<<dim3((s.N + 255) / 256, s.M), 256>>>(w, ref); }
    run<256, 256, 2, false, false, 1>("wgrad", w, check, ref, worst_d);
    run<192, 192, 1, false, false, 1>("wgrad", w, check, ref, worst_d);
  }
  const S wg[] = {{768, 768, 11648}, {2304, 768, 11648}, {3072, 768, 11648}, {768, 3072, 11648}};
  for (const S& s : wg) {
    P w{dA, dB, dC, s.M, s.N, s.K, s.M, s.N, s.N, 0, 0, 8};
    ref_kernel<false, false><<<dim3((s.N + 255) / 256, s.M), 256>>>(w, ref);
    run<256, 256, 2, false, false, 1>("wgrad", w, true, ref, worst_d);
    run<256, 128, 1, false, false, 1>("wgrad", w, true, ref, worst_d);
    run<128, 128, 1, false, false, 1>("wgrad", w, true, ref, worst_d);
  }
  return 0;
}
