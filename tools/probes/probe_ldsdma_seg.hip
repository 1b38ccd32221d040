// LDS-DMA streaming rate of the weight-gradient access pattern as a function of the SHAPE of a 1 KB DMA piece: eight 128-byte row segments (the
// current 64-column panel image of gemm8w.hip / gemm12w.hip), four of 256 bytes, two of 512 bytes (a whole 256-column tile row per half wave).
// 216 blocks (108 tiles of 256 x 256 x two K halves, XCD-contiguous runs as in gemm8w.hip) stream both k-strided operands of a [R, 3072]^T [R, 2304]
// product through a two-stage LDS ring with counted vmcnt; no MFMAs, no fragment reads: the rate the memory side can deliver into LDS.
// Not part of the product path (profiles/r5_gemm_experiments.txt, experiment 13).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned short bf16_t;

template <int SEG>   // bytes per contiguous row segment of a piece: 128, 256, 512
__device__ __forceinline__ unsigned piece_off(int j, int lane, long ld) {      // byte offset of this lane's 16 bytes of piece j (0..31) of a 64-row x 256-column tile
  if (SEG == 128) { const int panel = j >> 3, krow = 8 * (j & 7) + (lane >> 3), c = lane & 7; return (unsigned)((krow * ld + panel * 64 + c * 8) * 2); }
  if (SEG == 256) { const int half = j & 1, krow = 4 * (j >> 1) + (lane >> 4), c = lane & 15; return (unsigned)((krow * ld + half * 128 + c * 8) * 2); }
  const int krow = 2 * j + (lane >> 5), c = lane & 31;
  return (unsigned)((krow * ld + c * 8) * 2);
}

// (the resource descriptor is built inside a plain device function: a __amdgpu_buffer_rsrc_t local of a template kernel breaks the host-side pass)
__device__ __forceinline__ void dma16(const void* base, unsigned char* dst, unsigned off, unsigned soff) {
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, 0x7fffffff, 0x00020000);
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)dst, 16, off, soff, 0, 0);
}

template <int SEG, int NWAVES>
__global__ __launch_bounds__(64 * NWAVES) void stream_kernel(const bf16_t* A, long lda, const bf16_t* B, long ldb, int tiles_n, int kt_per_half, float* sink) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  constexpr int PW = 32 / NWAVES;            // pieces per wave per operand per k-tile
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  // block -> (tile, half): XCD x = block % 8 owns items [x * per, (x + 1) * per), walked n-fastest, both halves of a tile adjacent
  const int per = gridDim.x / 8, item = (blockIdx.x % 8) * per + blockIdx.x / 8;
  const int tile = item >> 1, half = item & 1, tm = tile / tiles_n, tn = tile % tiles_n;
  const bf16_t* a0 = A + (long)half * kt_per_half * 64 * lda + tm * 256;
  const bf16_t* b0 = B + (long)half * kt_per_half * 64 * ldb + tn * 256;
  unsigned offa[PW], offb[PW];
#pragma unroll
  for (int s = 0; s < PW; ++s) { offa[s] = piece_off<SEG>(wave * PW + s, lane, lda); offb[s] = piece_off<SEG>(wave * PW + s, lane, ldb); }
  for (int u = 0; u < kt_per_half; ++u) {
    unsigned char* st = lds + (u & 1) * 65536;
    const unsigned sa = (unsigned)(u * 64 * lda * 2), sb = (unsigned)(u * 64 * ldb * 2);
#pragma unroll
    for (int s = 0; s < PW; ++s)
      dma16(a0, st + (wave * PW + s) * 1024, offa[s], sa);
#pragma unroll
    for (int s = 0; s < PW; ++s)
      dma16(b0, st + 32768 + (wave * PW + s) * 1024, offb[s], sb);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PW) : "memory");      // the previous k-tile has landed, this one stays in flight
    __builtin_amdgcn_s_barrier();
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) sink[blockIdx.x] = lds[0];
}

template <int SEG, int NWAVES>
void run(const bf16_t* A, long lda, const bf16_t* B, long ldb, int R, float* sink, int blocks, const char* note) {
  auto k = stream_kernel<SEG, NWAVES>;
  hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
  const int tiles_n = (int)(ldb / 256), kth = R / 64 / 2;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<<<blocks, 64 * NWAVES, 131072>>>(A, lda, B, ldb, tiles_n, kth, sink);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int r = 0; r < 5; ++r) k<<<blocks, 64 * NWAVES, 131072>>>(A, lda, B, ldb, tiles_n, kth, sink);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
  const double bytes = (double)blocks * kth * 65536;
  printf("segments of %4d B, %d waves, %3d blocks %-28s: %7.1f us  %5.2f TB/s into LDS  (%.2f us per k-tile)\n", SEG, NWAVES, blocks, note, ms * 1e3, bytes / ms / 1e9, ms * 1e3 / kth);
}

int main() {
  const int R = 11648;
  const long lda = 3072, ldb = 2304;
  bf16_t *A, *B; float* sink;
  hipMalloc(&A, (size_t)R * lda * 2 + 65536); hipMalloc(&B, (size_t)R * ldb * 2 + 65536); hipMalloc(&sink, 4096);
  hipMemset(A, 0, (size_t)R * lda * 2); hipMemset(B, 0, (size_t)R * ldb * 2);
  for (int rep = 0; rep < 2; ++rep) {
    run<128, 8>(A, lda, B, ldb, R, sink, 216, "(the kernels' panel image)");
    run<256, 8>(A, lda, B, ldb, R, sink, 216, "");
    run<512, 8>(A, lda, B, ldb, R, sink, 216, "(whole tile rows)");
    run<128, 4>(A, lda, B, ldb, R, sink, 216, "(4 loader waves)");
    run<512, 4>(A, lda, B, ldb, R, sink, 216, "(4 loader waves)");
    run<128, 8>(A, lda, B, ldb, R, sink, 48, "(48 blocks: 6 per XCD)");
    run<512, 8>(A, lda, B, ldb, R, sink, 48, "(48 blocks: 6 per XCD)");
  }
  return 0;
}
