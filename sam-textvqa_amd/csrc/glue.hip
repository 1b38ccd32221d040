// Strided block copies / casts / zero-fills of the glue between the kernels of a step (gfx950): the `cat` of the four token groups into the
// [B, 182, 768] sequence (sam/sa_m4c.py:814-818) and its backward (four contiguous gradient slices), the OCR / decoder row slices of the MMT output
// (sa_m4c.py:270-278) and the zero-padded gradient they return, fp32 <-> bf16 casts and accumulations around PrevPredEmbeddings' backward, the
// zero-fills of atomics buffers and of the gradient ranges that are accumulated -- up to eight of them per launch, where eager PyTorch spends one
// launch (or three) each.  HBM-bound, trivially.
#include "common.h"
#include "sam_hip.h"

namespace {

constexpr int NT = 256, MAXD = 8;
struct CDesc {
  const void* src; void* dst;
  long long rows_total, cols4;          // batches * rows, cols / 4
  int rows;                             // rows per batch
  long long sb, sr, db, dr;             // batch / row strides in elements
  int src_f32, dst_f32, accumulate;
  int first_block, nblocks;
};
struct CArgs { CDesc d[MAXD]; int count; };

__global__ __launch_bounds__(NT) void copy_blocks_kernel(CArgs a) {
  int q = 0;
#pragma unroll
  for (int i = 1; i < MAXD; ++i)
    if (i < a.count && (int)blockIdx.x >= a.d[i].first_block) q = i;
  const CDesc& d = a.d[q];
  const long long total = d.rows_total * d.cols4;
  for (long long e = (long long)(blockIdx.x - d.first_block) * NT + threadIdx.x; e < total; e += (long long)d.nblocks * NT) {
    const long long r = e / d.cols4, c = (e - r * d.cols4) * 4;
    const long long b = r / d.rows, i = r - b * d.rows;
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    if (d.src) {
      const long long so = b * d.sb + i * d.sr + c;
      if (d.src_f32) { const float4 f = *reinterpret_cast<const float4*>((const float*)d.src + so); v[0] = f.x; v[1] = f.y; v[2] = f.z; v[3] = f.w; }
      else { const uint2 u = *reinterpret_cast<const uint2*>((const bf16_t*)d.src + so); v[0] = bf_lo(u.x); v[1] = bf_hi(u.x); v[2] = bf_lo(u.y); v[3] = bf_hi(u.y); }
    }
    const long long dofs = b * d.db + i * d.dr + c;
    if (d.dst_f32) {
      float4* p = reinterpret_cast<float4*>((float*)d.dst + dofs);
      if (d.accumulate) { const float4 o = *p; v[0] += o.x; v[1] += o.y; v[2] += o.z; v[3] += o.w; }
      *p = make_float4(v[0], v[1], v[2], v[3]);
    } else {
      uint2* p = reinterpret_cast<uint2*>((bf16_t*)d.dst + dofs);
      if (d.accumulate) { const uint2 o = *p; v[0] += bf_lo(o.x); v[1] += bf_hi(o.x); v[2] += bf_lo(o.y); v[3] += bf_hi(o.y); }
      uint2 w; w.x = pack_bf16x2(v[0], v[1]); w.y = pack_bf16x2(v[2], v[3]);
      *p = w;
    }
  }
}

__global__ __launch_bounds__(NT) void ge_u8_kernel(const long long* x, long long n, long long thr, unsigned char* out) {
  const long long i = (long long)blockIdx.x * NT + threadIdx.x;
  if (i < n) out[i] = x[i] >= thr ? 1 : 0;
}

// out[r, c] = a[r, c] * b[r, c] (mode 0) | a[r, c] + vec[c] (mode 1) | a[r, c] * vec[c] (mode 2): bf16 rows, fp32 arithmetic, one rounding.  Four columns per thread.
template <int MODE>
__global__ __launch_bounds__(NT) void rowvec_kernel(const bf16_t* a, long long lda, const bf16_t* b, long long ldb, const float* vec, bf16_t* out, long long ldo, long long rows, int cols4) {
  const long long total = rows * cols4;
  for (long long e = (long long)blockIdx.x * NT + threadIdx.x; e < total; e += (long long)gridDim.x * NT) {
    const long long r = e / cols4;
    const int c = (int)(e - r * cols4) * 4;
    const uint2 u = *reinterpret_cast<const uint2*>(a + r * lda + c);
    float v[4] = {bf_lo(u.x), bf_hi(u.x), bf_lo(u.y), bf_hi(u.y)};
    if (MODE == 0) {
      const uint2 w = *reinterpret_cast<const uint2*>(b + r * ldb + c);
      v[0] *= bf_lo(w.x); v[1] *= bf_hi(w.x); v[2] *= bf_lo(w.y); v[3] *= bf_hi(w.y);
    } else {
      const float4 f = *reinterpret_cast<const float4*>(vec + c);
      if (MODE == 1) { v[0] += f.x; v[1] += f.y; v[2] += f.z; v[3] += f.w; }
      else { v[0] *= f.x; v[1] *= f.y; v[2] *= f.z; v[3] *= f.w; }
    }
    uint2 o; o.x = pack_bf16x2(v[0], v[1]); o.y = pack_bf16x2(v[2], v[3]);
    *reinterpret_cast<uint2*>(out + r * ldo + c) = o;
  }
}

}  // namespace

extern "C" int sam_rowvec_bf16(int mode, const void* a, int64_t lda, const void* b, int64_t ldb, const float* vec, void* out, int64_t ldo, int64_t rows, int cols, void* stream) {
  SAM_REQUIRE(a && out && rows > 0 && cols > 0 && cols % 4 == 0 && lda % 4 == 0 && ldo % 4 == 0 && (uintptr_t)a % 8 == 0 && (uintptr_t)out % 8 == 0,
              "sam_rowvec_bf16: null / empty operand, or width / strides / pointers not multiples of 4 elements");
  SAM_REQUIRE(mode >= 0 && mode <= 2, "sam_rowvec_bf16: mode %d (0 = a * b, 1 = a + vec, 2 = a * vec)", mode);
  SAM_REQUIRE(mode == 0 ? (b && ldb % 4 == 0 && (uintptr_t)b % 8 == 0) : (vec && (uintptr_t)vec % 16 == 0), "sam_rowvec_bf16: mode %d needs %s", mode, mode == 0 ? "b (8-byte aligned rows)" : "vec (fp32, 16-byte aligned)");
  const long long total = rows * (cols / 4);
  long long nb = (total + (long long)NT * 4 - 1) / ((long long)NT * 4);
  nb = nb < 1 ? 1 : (nb > 4096 ? 4096 : nb);
  const hipStream_t st = (hipStream_t)stream;
  if (mode == 0) rowvec_kernel<0><<<dim3((unsigned)nb), dim3(NT), 0, st>>>((const bf16_t*)a, lda, (const bf16_t*)b, ldb, vec, (bf16_t*)out, ldo, rows, cols / 4);
  else if (mode == 1) rowvec_kernel<1><<<dim3((unsigned)nb), dim3(NT), 0, st>>>((const bf16_t*)a, lda, nullptr, 0, vec, (bf16_t*)out, ldo, rows, cols / 4);
  else rowvec_kernel<2><<<dim3((unsigned)nb), dim3(NT), 0, st>>>((const bf16_t*)a, lda, nullptr, 0, vec, (bf16_t*)out, ldo, rows, cols / 4);
  SAM_LAUNCH_CHECK();
  return SAM_OK;
}

extern "C" int sam_copy_blocks(const sam_copy_desc* descs, int count, void* stream) {
  SAM_REQUIRE(descs && count >= 1 && count <= MAXD, "sam_copy_blocks: 1..%d blocks per launch", MAXD);
  CArgs a = {};
  a.count = count;
  int total = 0;
  for (int q = 0; q < count; ++q) {
    const sam_copy_desc& s = descs[q];
    SAM_REQUIRE(s.dst && s.batches > 0 && s.rows > 0 && s.cols > 0 && s.cols % 4 == 0, "sam_copy_blocks: block %d is empty or its width is not a multiple of 4", q);
    const int sa = s.src_f32 ? 4 : 4, da = 4;
    SAM_REQUIRE((!s.src || (s.src_batch_stride % sa == 0 && s.src_row_stride % sa == 0 && (uintptr_t)s.src % (s.src_f32 ? 16 : 8) == 0)) && s.dst_batch_stride % da == 0 &&
                    s.dst_row_stride % da == 0 && (uintptr_t)s.dst % (s.dst_f32 ? 16 : 8) == 0,
                "sam_copy_blocks: block %d: strides must be multiples of 4 elements, pointers 8 / 16-byte aligned", q);
    CDesc& d = a.d[q];
    d.src = s.src; d.dst = s.dst; d.rows = s.rows; d.rows_total = (long long)s.batches * s.rows; d.cols4 = s.cols / 4;
    d.sb = s.src_batch_stride; d.sr = s.src_row_stride; d.db = s.dst_batch_stride; d.dr = s.dst_row_stride;
    d.src_f32 = s.src_f32; d.dst_f32 = s.dst_f32; d.accumulate = s.accumulate;
    const long long chunks = d.rows_total * d.cols4;
    long long nb = (chunks + (long long)NT * 4 - 1) / ((long long)NT * 4);           // ~4 chunks of 4 elements per thread
    nb = nb < 1 ? 1 : (nb > 2048 ? 2048 : nb);
    d.first_block = total; d.nblocks = (int)nb;
    total += (int)nb;
  }
  copy_blocks_kernel<<<dim3(total), dim3(NT), 0, (hipStream_t)stream>>>(a);
  SAM_LAUNCH_CHECK();
  return SAM_OK;
}

extern "C" int sam_ge_u8(const int64_t* x, int64_t n, int64_t threshold, uint8_t* out, void* stream) {
  SAM_REQUIRE(x && out && n > 0, "sam_ge_u8: null pointer or empty");
  ge_u8_kernel<<<dim3((unsigned)((n + NT - 1) / NT)), dim3(NT), 0, (hipStream_t)stream>>>((const long long*)x, n, threshold, out);
  SAM_LAUNCH_CHECK();
  return SAM_OK;
}
