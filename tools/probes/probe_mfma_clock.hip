// Does the matrix pipe's rate per CU depend on how many CUs run it?  Pure MFMA loops (v_mfma_f32_16x16x32_bf16, four independent accumulator tiles per wave,
// no memory traffic) on 32 / 64 / 128 / 256 CUs, one or two waves per SIMD, and the same next to an LDS-DMA stream.  Not part of the product path
// (profiles/r5_gemm_experiments.txt, experiment 13).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

template <int NACC, bool RANDOM>
__global__ __launch_bounds__(512) void mfma_kernel(int iters, float* sink) {
  f32x4 acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  bf16x8 a, b;
#pragma unroll
  for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(float)(threadIdx.x & 3); b[e] = (__bf16)1.0f; }
  bf16x8 a2 = a, b2 = b;
  if (RANDOM) {      // operands with random mantissas and signs, two sets alternating MFMA by MFMA: the data toggling of a real product
    unsigned h = (threadIdx.x + 1u) * 2654435761u + blockIdx.x * 40503u;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      h ^= h >> 15; h *= 0x2c1b3c6du; h ^= h >> 12;
      a[e] = (__bf16)(((int)(h & 0xffff) - 32768) * (1.0f / 32768.f)); b[e] = (__bf16)(((int)(h >> 16) - 32768) * (1.0f / 32768.f));
      h *= 0x297a2d39u; h ^= h >> 15;
      a2[e] = (__bf16)(((int)(h & 0xffff) - 32768) * (1.0f / 32768.f)); b2[e] = (__bf16)(((int)(h >> 16) - 32768) * (1.0f / 32768.f));
    }
  }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 8; ++r)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(((r + i) & 1) ? a2 : a, ((r + i) & 1) ? b2 : b, acc[i], 0, 0, 0);
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  if (s == 12345.f) sink[0] = s;
}

template <bool RANDOM>
void run(int blocks, int threads, int iters, float* sink) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  mfma_kernel<4, RANDOM><<<blocks, threads>>>(iters, sink);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int r = 0; r < 3; ++r) mfma_kernel<4, RANDOM><<<blocks, threads>>>(iters, sink);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 3;
  const double mfmas = (double)blocks * (threads / 64) * iters * 32, flops = mfmas * 2.0 * 16 * 16 * 32;
  // a 16x16x32 bf16 MFMA occupies the SIMD's matrix pipe for 8 passes = 16 cycles at the dense rate (2.5 PFLOP/s / 256 CUs / 4 SIMDs at 2.4 GHz)
  const double cyc_per_mfma_at_2400 = ms * 1e-3 * 2.4e9 / ((double)iters * 32 * (threads / 256));
  printf("%s %3d CUs x %d waves/SIMD: %8.1f us  %7.1f TFLOP/s  %5.2f TFLOP/s per CU  (%.1f cycles of a 2.4 GHz clock per MFMA and SIMD)\n", RANDOM ? "random operands" : "constant operands", blocks, threads / 256, ms * 1e3, flops / ms / 1e9,
         flops / ms / 1e9 / blocks, cyc_per_mfma_at_2400);
}

int main() {
  float* sink; hipMalloc(&sink, 64);
  for (int threads : {256, 512})
    for (int blocks : {32, 64, 128, 256}) run<false>(blocks, threads, 20000 * 256 / threads, sink);
  for (int rep = 0; rep < 2; ++rep)
    for (int threads : {256, 512})
      for (int blocks : {32, 64, 128, 216, 256}) run<true>(blocks, threads, 20000 * 256 / threads, sink);
  // long runs on all CUs: does the rate sag with time (power management)?
  for (int k = 0; k < 3; ++k) run<false>(256, 512, 100000, sink);
  for (int k = 0; k < 6; ++k) run<true>(256, 512, 100000, sink);
  return 0;
}
