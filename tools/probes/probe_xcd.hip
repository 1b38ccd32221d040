// PROBE (not part of the library): questions behind the XCD-partitioned decoding kernel (csrc/decode_steps.hip):
//  1. does a block see its XCD in HW_REG_XCC_ID, and is it blockIdx % 8 for a 256-block launch?
//  2. what does a barrier among the 32 blocks of one XCD cost when its atomics execute in that XCD's L2 (workgroup scope: no sc1) against
//     a barrier among all 256 blocks with agent-scope atomics?
//  3. how long do 8 XCDs take to stream the same 4.7 MB (one FFN weight) when each of them reads all of it?
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/probes/probe_xcd.hip -o tools/probes/probe_xcd && tools/probes/probe_xcd
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__device__ __forceinline__ unsigned xcc_id() { return __builtin_amdgcn_s_getreg((4 - 1) << 11 | 0 << 6 | 20) & 0xf; }   // HW_REG_XCC_ID = 20, bits [3:0]

__global__ void k_ids(unsigned* out) { if (threadIdx.x == 0) out[blockIdx.x] = xcc_id(); }

template <int SCOPE>
__global__ __launch_bounds__(256) void k_barrier(unsigned* cnt, int per_group, int iters, long long* t) {
  const int grp = SCOPE == __HIP_MEMORY_SCOPE_WORKGROUP ? blockIdx.x % 8 : 0;
  unsigned* c = cnt + grp * 64;
  unsigned epoch = 0;
  long long t0 = 0;
  for (int it = 0; it < iters + 1; ++it) {
    if (it == 1 && threadIdx.x == 0) t0 = wall_clock64();
    epoch += per_group;
    __syncthreads();
    if (threadIdx.x == 0) {
      __hip_atomic_fetch_add(c, 1u, __ATOMIC_RELAXED, SCOPE);
      while (__hip_atomic_load(c, __ATOMIC_RELAXED, SCOPE) < epoch) __builtin_amdgcn_s_sleep(1);
    }
    __syncthreads();
  }
  if (threadIdx.x == 0 && blockIdx.x == 0) t[0] = wall_clock64() - t0;
}

// every XCD group (blocks with the same blockIdx % 8) reads all of w (n16 16-byte pieces), coalesced, 8 loads in flight per thread
__global__ __launch_bounds__(512) void k_stream(const uint4* w, long long n16, unsigned* sink, long long* t) {
  const int grp = blockIdx.x % 8, rank = blockIdx.x / 8, nb = gridDim.x / 8;
  long long t0 = wall_clock64();
  uint4 acc = {0, 0, 0, 0};
  for (long long i = (long long)rank * 512 + threadIdx.x; i < n16; i += (long long)nb * 512 * 8) {
    uint4 v[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) { const long long j = i + (long long)q * nb * 512; v[q] = j < n16 ? w[j] : uint4{0, 0, 0, 0}; }
#pragma unroll
    for (int q = 0; q < 8; ++q) { acc.x ^= v[q].x; acc.y ^= v[q].y; acc.z ^= v[q].z; acc.w ^= v[q].w; }
  }
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[0] = grp;
  __syncthreads();
  if (threadIdx.x == 0) t[blockIdx.x] = wall_clock64() - t0;
}

int main() {
  unsigned* ids; hipMalloc(&ids, 256 * 4);
  k_ids<<<256, 64>>>(ids);
  std::vector<unsigned> h(256); hipMemcpy(h.data(), ids, 256 * 4, hipMemcpyDeviceToHost);
  int match = 0; for (int b = 0; b < 256; ++b) match += h[b] == (unsigned)(b % 8);
  printf("XCC_ID of blocks 0..15:"); for (int b = 0; b < 16; ++b) printf(" %u", h[b]); printf("   blockIdx %% 8 == XCC_ID for %d / 256 blocks\n", match);
  unsigned* cnt; hipMalloc(&cnt, 8 * 64 * 4); long long* t; hipMalloc(&t, 256 * 8);
  for (int rep = 0; rep < 2; ++rep) {
    long long ht;
    hipMemset(cnt, 0, 8 * 64 * 4);
    k_barrier<__HIP_MEMORY_SCOPE_AGENT><<<256, 256>>>(cnt, 256, 200, t); hipMemcpy(&ht, t, 8, hipMemcpyDeviceToHost);
    printf("barrier over 256 blocks, agent-scope atomics:      %.2f us each\n", ht / 100.0 / 200);
    hipMemset(cnt, 0, 8 * 64 * 4);
    k_barrier<__HIP_MEMORY_SCOPE_WORKGROUP><<<256, 256>>>(cnt, 32, 200, t); hipMemcpy(&ht, t, 8, hipMemcpyDeviceToHost);
    printf("barrier over the 32 blocks of an XCD, L2 atomics:  %.2f us each\n", ht / 100.0 / 200);
  }
  const long long bytes = 3072LL * 768 * 2; uint4* w; hipMalloc(&w, bytes * 4); hipMemset(w, 1, bytes * 4); unsigned* sink; hipMalloc(&sink, 4);
  for (int rep = 0; rep < 3; ++rep) {
    k_stream<<<256, 512>>>(w + rep * (bytes / 16), bytes / 16, sink, t);
    std::vector<long long> ht(256); hipMemcpy(ht.data(), t, 256 * 8, hipMemcpyDeviceToHost);
    long long mx = 0; for (auto v : ht) mx = v > mx ? v : mx;
    printf("8 XCDs each streaming the same %.1f MB (cold, rep %d): slowest block %.2f us  -> %.2f TB/s aggregate, %.0f GB/s per XCD\n", bytes / 1e6, rep, mx / 100.0,
           8 * bytes / (mx / 100.0) / 1e6, bytes / (mx / 100.0) / 1e3);
  }
  k_stream<<<256, 512>>>(w, bytes / 16, sink, t);
  { std::vector<long long> ht(256); hipMemcpy(ht.data(), t, 256 * 8, hipMemcpyDeviceToHost); long long mx = 0; for (auto v : ht) mx = v > mx ? v : mx;
    printf("same, second touch of rep 0's buffer (MALL-warm?): %.2f us\n", mx / 100.0); }
  return 0;
}
