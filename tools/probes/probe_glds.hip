// LDS-DMA fill-rate probe: how many bytes/s can a CU pull into LDS with global_load_lds_dwordx4, as a function of
// resident blocks per CU, waves per block, ring depth (stages kept in flight with counted vmcnt) and source locality.
// Not part of the product path; it sized the GEMM's staging scheme.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int NWAVES, int PER_WAVE, int DEPTH>
__global__ __launch_bounds__(64 * NWAVES) void fill_kernel(const unsigned short* src, long ld, int rows_total, int iters, float* sink) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  constexpr int TILE_ROWS = NWAVES * PER_WAVE * 8;       // rows of 128 B per stage
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const long row0 = ((long)blockIdx.x * TILE_ROWS) % (rows_total - TILE_ROWS);
  for (int it = 0; it < iters; ++it) {
    unsigned char* stage = lds + (it % DEPTH) * TILE_ROWS * 128;
#pragma unroll
    for (int jj = 0; jj < PER_WAVE; ++jj) {
      const int j = wave * PER_WAVE + jj;
      const int row = 8 * j + (lane >> 3), c = lane & 7;
      const unsigned short* p = src + (row0 + row) * ld + (long)(it % 48) * 64 + c * 8;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)p, (__attribute__((address_space(3))) void*)(stage + j * 1024), 16, 0, 0);
    }
    // consume point: stage it-(DEPTH-1) must have landed; (DEPTH-1) younger stages stay in flight
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((DEPTH - 1) * PER_WAVE) : "memory");
    __builtin_amdgcn_s_barrier();
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid == 0) sink[blockIdx.x] = lds[0];
}
template <int NWAVES, int PER_WAVE, int DEPTH>
void run(const char* tag, const unsigned short* src, long ld, long rows, float* sink, int blocks, long foot_rows) {
  const int iters = 96;
  const size_t lds = (size_t)DEPTH * NWAVES * PER_WAVE * 1024;
  auto k = fill_kernel<NWAVES, PER_WAVE, DEPTH>;
  hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<<<blocks, 64 * NWAVES, lds>>>(src, ld, (int)foot_rows, iters, sink);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int r = 0; r < 5; ++r) k<<<blocks, 64 * NWAVES, lds>>>(src, ld, (int)foot_rows, iters, sink);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
  const double bytes = (double)blocks * iters * NWAVES * PER_WAVE * 1024;
  printf("%-10s waves %d stage %3d KB depth %d (LDS %3zu KB) blocks %4d foot %6ld rows: %8.1f us %6.2f TB/s %6.1f GB/s/CU\n", tag, NWAVES, NWAVES * PER_WAVE, DEPTH,
         lds / 1024, blocks, foot_rows, ms * 1e3, bytes / ms / 1e9, bytes / ms / 1e6 / 256);
}
int main() {
  const long rows = 11648 * 4, K = 4096, ld = K;   // 380 MB source: larger than the Infinity Cache
  unsigned short* src; float* sink;
  hipMalloc(&src, rows * ld * 2); hipMalloc(&sink, 8192 * 4);
  hipMemset(src, 0, rows * ld * 2);
  for (long foot : {2048L, rows}) {
    run<4, 8, 1>("4w sync", src, ld, rows, sink, 256, foot);
    run<4, 8, 1>("4w sync", src, ld, rows, sink, 512, foot);
    run<4, 8, 2>("4w ring", src, ld, rows, sink, 512, foot);
    run<4, 8, 2>("4w ring", src, ld, rows, sink, 256, foot);
    run<8, 4, 1>("8w sync", src, ld, rows, sink, 256, foot);
    run<8, 4, 2>("8w ring", src, ld, rows, sink, 256, foot);
    run<8, 4, 3>("8w ring", src, ld, rows, sink, 256, foot);
    run<8, 4, 4>("8w ring", src, ld, rows, sink, 256, foot);
    run<8, 5, 4>("8w ring", src, ld, rows, sink, 256, foot);
    run<8, 8, 2>("8w ring", src, ld, rows, sink, 256, foot);
  }
  return 0;
}
