// Device helpers shared by the 8-wave GEMM kernels (gemm8.hip: forward / dgrad; gemm8w.hip: grouped weight gradients): LDS images, DMA slices,
// fragment reads.  See the header comment of gemm8.hip for the layouts.
#pragma once
#include "gemm_common.h"

namespace samgemm8 {
using namespace samgemm;

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
  static_assert(N >= 0 && N <= 63, "vmwait: vmcnt is a 6-bit counter");
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
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

}  // namespace samgemm8
