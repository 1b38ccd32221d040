// LDS-DMA fill-rate probe: how many bytes/s can a CU pull into LDS with global_load_lds_dwordx4, as a function of
// resident blocks per CU, waves per block, outstanding depth and source locality.  Not part of the product path.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int DEPTH>
__global__ __launch_bounds__(256) void fill_kernel(const unsigned short* src, long ld, int rows_total, int iters, int tile_rows, float* sink) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int nslice = tile_rows / 8;          // 1 KB slices per stage (rows x 128 B)
  const int per_wave = nslice / 4;
  const long row0 = ((long)blockIdx.x * tile_rows) % (rows_total - tile_rows);
  for (int it = 0; it < iters; ++it) {
    unsigned char* stage = lds + (it % DEPTH) * tile_rows * 128;
    for (int jj = 0; jj < per_wave; ++jj) {
      const int j = wave * per_wave + jj;
      const int row = 8 * j + (lane >> 3), c = lane & 7;
      const unsigned short* p = src + (row0 + row) * ld + (long)it * 64 + c * 8;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)p, (__attribute__((address_space(3))) void*)(stage + j * 1024), 16, 0, 0);
    }
    if (DEPTH == 1) { __syncthreads(); }
    else {
      // keep DEPTH-1 stages in flight
      if (it >= DEPTH - 1) {
        if (DEPTH == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(8) : "memory");   // placeholder, overwritten below per per_wave
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid == 0) sink[blockIdx.x] = lds[0];
}
int main() {
  const long rows = 11648 * 4, K = 4096, ld = K;   // 380 MB source: larger than MALL
  unsigned short* src; float* sink;
  hipMalloc(&src, rows * ld * 2); hipMalloc(&sink, 4096 * 4);
  hipMemset(src, 0, rows * ld * 2);
  hipFuncSetAttribute((const void*)fill_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int tile_rows : {256, 512}) {          // 32 KB / 64 KB per stage (like A+B of a 128^2 / 256^2 tile)
    for (int blocks : {256, 512, 1024, 2048}) {
      for (long foot_rows : {2048L, rows}) {   // L2/MALL-resident vs HBM-streaming source
        const int iters = 48;
        size_t lds = (size_t)tile_rows * 128;
        fill_kernel<1><<<blocks, 256, lds>>>(src, ld, (int)foot_rows, iters, tile_rows, sink);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        for (int r = 0; r < 5; ++r) fill_kernel<1><<<blocks, 256, lds>>>(src, ld, (int)foot_rows, iters, tile_rows, sink);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
        double bytes = (double)blocks * iters * tile_rows * 128;
        printf("stage %3d KB  blocks %5d  footprint %6ld rows : %7.1f us  %7.2f TB/s total  %6.1f GB/s per CU  (LDS/block %zu KB)\n",
               tile_rows * 128 / 1024, blocks, foot_rows, ms * 1e3, bytes / ms / 1e9, bytes / ms / 1e6 / 256, lds / 1024);
      }
    }
  }
  return 0;
}
