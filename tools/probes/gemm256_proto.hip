// PROTOTYPE (not part of the library): C[M,N] = A[M,K] . B[N,K]^T in bf16 with a 256x256x64 block tile, 4 waves, 128x128 per-wave
// register tile (64 accumulators of 4 = 256 registers: the accumulators have to live in AGPRs, one wave per SIMD, 512 registers).
// Question it answers: can hipcc generate a usable kernel of the shape the library uses for these sizes?
// Answer (MI355X, round 1): it allocates 256 AGPR accumulators + 68-76 VGPRs without spills and MFMAs accumulate in AGPRs directly, but with
// one wave per SIMD the schedule is everything: BK=64 double buffer, scheduler's own order: 447 TFLOP/s at 11648x3072x768 (768 at 8192^3);
// BK=32 4-stage ring with counted vmcnt + fragment prefetch forced by sched_group_barrier: 514 / 571 (N=2304) / 678 (K=3072) / 842 (8192^3).
// The library's 2x2-wave kernels of this shape reach 890-1100 on the same problems, ours (2 blocks x 4 waves of 96x64) 670-910: the
// register tile alone does not pay without the hand-placed 8-phase schedule.  (Same ring with 8 waves of 128x64: 545 / 607 / 703 / 850.)
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I sam-textvqa_amd/csrc -I include tools/probes/gemm256_proto.hip -o tools/probes/gemm256_proto
//   tools/probes/gemm256_proto [M N K]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "common.h"

extern "C" void sam_set_error(const char*, ...) {}

constexpr int BM = 256, BN = 256, BK = 32, STAGES = 4;
struct P { const bf16_t* A; const bf16_t* B; bf16_t* C; int M, N, K; int64_t lda, ldb, ldc; int tiles_m, tiles_n; };

// [R rows][32 k] bf16, 64-byte rows, 16-byte chunk c of row r stored at c ^ ((r >> 2) & 3): the 16 rows x 4 chunks of one MFMA fragment
// (ds_read_b128, lane (i,g) = row i, chunk g) then cover all 64 banks in every 16-lane phase
__device__ __forceinline__ int kc32_off(int row, int chunk) { return row * 64 + ((chunk ^ ((row >> 2) & 3)) << 4); }

// one 256-row x 32-k bf16 slice (16 KB) global -> LDS by DMA: 4 instructions per wave (4 waves); a wave instruction fills 16 rows
__device__ __forceinline__ void glds256(unsigned char* lds, const bf16_t* base, int64_t ld, int row0, int rows, int k0, int wave, int lane) {
#pragma unroll
  for (int jj = 0; jj < 4; ++jj) {
    const int j = wave * 4 + jj;                 // 1 KB slice = 16 rows
    const int row = 16 * j + (lane >> 2), pos = lane & 3, c = pos ^ ((row >> 2) & 3);
    const int grow = min(row0 + row, rows - 1);
    const bf16_t* src = base + (int64_t)grow * ld + k0 + c * 8;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (__attribute__((address_space(3))) void*)(lds + j * 1024), 16, 0, 0);
  }
}

__global__ __launch_bounds__(256, 1) void gemm256(P p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int A_BYTES = BM * BK * 2, STAGE = 2 * A_BYTES;      // 16 KB + 16 KB
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, i = lane & 15, g = lane >> 4;
  const int wm = wave & 1, wn = wave >> 1;
  const int nblk = p.tiles_m * p.tiles_n;
  int bid = blockIdx.x;
  {  // XCD-contiguous tile ranges
    const int q = nblk / 8, r = nblk % 8, xcd = bid % 8, loc = bid / 8;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
  }
  const int m0 = (bid / p.tiles_n) * BM, n0 = (bid % p.tiles_n) * BN;

  f32x4 acc[8][8];   // [tn][tm]
#pragma unroll
  for (int a = 0; a < 8; ++a)
#pragma unroll
    for (int b = 0; b < 8; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int KT = p.K / BK;
  auto fill = [&](int kt) {      // 8 DMA instructions per thread
    unsigned char* As_ = smem + (kt % STAGES) * STAGE;
    glds256(As_, p.A, p.lda, m0, p.M, kt * BK, wave, lane);
    glds256(As_ + A_BYTES, p.B, p.ldb, n0, p.N, kt * BK, wave, lane);
  };
  // ring of 4 stages, 3 k-steps in flight: vmcnt never drains inside the loop
  fill(0);
  if (KT > 1) fill(1);
  if (KT > 2) fill(2);
  for (int kt = 0; kt < KT; ++kt) {
    // k-step kt has landed when at most the younger fills remain outstanding (8 instructions each)
    const int younger = min(KT - 1 - kt, 2);
    if (younger == 2) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    else if (younger == 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                 // everyone's part of k-step kt is in LDS, and everyone is done reading stage (kt-1)%4
    if (kt + 3 < KT) fill(kt + 3);                // into stage (kt+3)%4 == (kt-1)%4
    const unsigned char* As = smem + (kt % STAGES) * STAGE;
    const unsigned char* Bs = As + A_BYTES;
    bf16x8 bf[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) bf[t] = *reinterpret_cast<const bf16x8*>(Bs + kc32_off(wn * 128 + t * 16 + i, g));
    bf16x8 af = *reinterpret_cast<const bf16x8*>(As + kc32_off(wm * 128 + i, g));
#pragma unroll
    for (int tm = 0; tm < 8; ++tm) {
      bf16x8 af_next = af;
      if (tm + 1 < 8) af_next = *reinterpret_cast<const bf16x8*>(As + kc32_off(wm * 128 + (tm + 1) * 16 + i, g));
#pragma unroll
      for (int tn = 0; tn < 8; ++tn) acc[tn][tm] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[tn], af, acc[tn][tm], 0, 0, 0);
      af = af_next;
    }
    // issue order: 8 B fragments + A0, then {A[tm+1]; 8 MFMAs on A[tm]}: each fragment read is covered by the 8 MFMAs (128 cycles) before its use
    __builtin_amdgcn_sched_group_barrier(0x100, 9, 0);
#pragma unroll
    for (int tm = 0; tm < 8; ++tm) {
      if (tm + 1 < 8) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
    }
  }
#pragma unroll
  for (int tm = 0; tm < 8; ++tm) {
    const int m = m0 + wm * 128 + tm * 16 + i;
#pragma unroll
    for (int tn = 0; tn < 8; ++tn) {
      const int n = n0 + wn * 128 + tn * 16 + 4 * g;
      if (m < p.M && n < p.N)
        *reinterpret_cast<uint2*>(p.C + (int64_t)m * p.ldc + n) = make_uint2(pack_bf16x2(acc[tn][tm][0], acc[tn][tm][1]), pack_bf16x2(acc[tn][tm][2], acc[tn][tm][3]));
    }
  }
}

static uint16_t f2bf_h(float f) { uint32_t u; std::memcpy(&u, &f, 4); u += 0x7fffu + ((u >> 16) & 1u); return (uint16_t)(u >> 16); }
static float bf2f_h(uint16_t h) { uint32_t u = (uint32_t)h << 16; float f; std::memcpy(&f, &u, 4); return f; }

int main(int argc, char** argv) {
  const int M = argc > 3 ? atoi(argv[1]) : 11648, N = argc > 3 ? atoi(argv[2]) : 3072, K = argc > 3 ? atoi(argv[3]) : 768;
  std::vector<uint16_t> a((size_t)M * K), b((size_t)N * K);
  uint32_t s = 1;
  for (auto& v : a) { s = s * 1664525u + 1013904223u; v = f2bf_h(((s >> 8) & 0xffff) / 32768.0f - 1.0f); }
  for (auto& v : b) { s = s * 1664525u + 1013904223u; v = f2bf_h((((s >> 8) & 0xffff) / 32768.0f - 1.0f) * 0.1f); }
  bf16_t *dA, *dB, *dC;
  hipMalloc(&dA, a.size() * 2); hipMalloc(&dB, b.size() * 2); hipMalloc(&dC, (size_t)M * N * 2);
  hipMemcpy(dA, a.data(), a.size() * 2, hipMemcpyHostToDevice); hipMemcpy(dB, b.data(), b.size() * 2, hipMemcpyHostToDevice);
  P p{dA, dB, dC, M, N, K, K, K, N, (M + BM - 1) / BM, (N + BN - 1) / BN};
  const size_t lds = (size_t)STAGES * 2 * BM * BK * 2;
  hipFuncSetAttribute(reinterpret_cast<const void*>(gemm256), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  const int grid = p.tiles_m * p.tiles_n;
  for (int it = 0; it < 3; ++it) gemm256<<<grid, 256, lds>>>(p);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  const int iters = 20;
  for (int it = 0; it < iters; ++it) gemm256<<<grid, 256, lds>>>(p);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms = 0; hipEventElapsedTime(&ms, e0, e1);
  const double us = ms * 1e3 / iters;
  std::printf("gemm256 proto  M=%d N=%d K=%d  tiles=%d  %.1f us  %.1f TFLOP/s  (%s)\n", M, N, K, grid, us, 2.0 * M * N * K / us / 1e6, hipGetErrorString(hipGetLastError()));
  std::vector<uint16_t> c((size_t)M * N);
  hipMemcpy(c.data(), dC, c.size() * 2, hipMemcpyDeviceToHost);
  double worst = 0;
  for (int t = 0; t < 200; ++t) {
    s = s * 1664525u + 1013904223u; const int m = (s >> 8) % M;
    s = s * 1664525u + 1013904223u; const int n = (s >> 8) % N;
    double r = 0;
    for (int k = 0; k < K; ++k) r += (double)bf2f_h(a[(size_t)m * K + k]) * bf2f_h(b[(size_t)n * K + k]);
    worst = std::fmax(worst, std::fabs(bf2f_h(c[(size_t)m * N + n]) - r) / (std::fabs(r) + 0.05));
  }
  std::printf("max rel err over 200 samples: %.4f %s\n", worst, worst < 0.02 ? "ok" : "FAIL");
  return 0;
}
