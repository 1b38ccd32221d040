// Spatial relation graph on the GPU: one thread per ordered box pair, float64 in the reference's operation
// order, emitting the multi-hot int8 [B,N,N,12] relation tensor the model consumes (batch_dict["spatial_adj_matrices"][c]).
// Replaces the O(N^2) Python loop of /root/reference/sam/spatial_utils.py:92-218 (0.49 s/sample) + the one-hot broadcast
// (:33-52) + the context composition of sam/datasets/textvqa_dataset.py:378-409.
// Relation codes: 1 a covers b, 2 a inside b, 3 IoU >= 0.5, 4..11 sector of the centre direction (if centre distance <
// threshold*sqrt(2)), 12 self; channel = code-1; context c adds the sector channels within +-(c-1)/2 (wrapping in 4..11).
#include "common.h"
#include "sam_hip.h"

namespace {

struct Box { double x0, y0, x1, y1; };
__device__ __forceinline__ Box load_box(const double* p) { return {p[0], p[1], p[2], p[3]}; }
__device__ __forceinline__ bool covers(const Box& a, const Box& b) { return a.x0 < b.x0 && a.x1 > b.x1 && a.y0 < b.y0 && a.y1 > b.y1; }

// sector codes of the pair (i, j), i < j: first = i -> j, second = j -> i   (spatial_utils.py:168-203)
__device__ __forceinline__ void sector_pair(const Box& bi, const Box& bj, int& cij, int& cji) {
  const double pi = 3.141592653589793;
  const double dy = 0.5 * (bi.y0 + bi.y1) - 0.5 * (bj.y0 + bj.y1);
  const double dx = 0.5 * (bi.x0 + bi.x1) - 0.5 * (bj.x0 + bj.x1);
  const double dist = sqrt(dy * dy + dx * dx);
  if (dist == 0.0) { cij = cji = 4; return; }    // reference: 0/0 -> nan -> both codes 4
  const double s = dy / dist, c = dx / dist;
  double li, lj;
  if (s >= 0 && c >= 0) { li = asin(s); lj = pi + li; }
  else if (s < 0 && c >= 0) { li = asin(s) + 2 * pi; lj = li - pi; }
  else if (s >= 0 && c < 0) { li = acos(c); lj = li + pi; }
  else { li = 2 * pi - acos(c); lj = li - pi; }
  const double q = pi / 4.0;
  cij = (int)ceil(li / q) + 3;
  cji = (int)ceil(lj / q) + 3;
}

__global__ __launch_bounds__(256) void relation_kernel(const double* boxes, int B, int N, int width, double limit, int8_t* out) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (int64_t)B * N * N) return;
  const int bcol = idx % N, arow = (idx / N) % N, b = idx / ((int64_t)N * N);
  const Box A = load_box(boxes + ((int64_t)b * N + arow) * 4), Bx = load_box(boxes + ((int64_t)b * N + bcol) * 4);
  const bool va = (A.x0 + A.y0 + A.x1 + A.y1) != 0.0, vb = (Bx.x0 + Bx.y0 + Bx.x1 + Bx.y1) != 0.0;
  int code = 0;
  if (va && vb) {
    if (arow == bcol) code = 12;
    else if (covers(A, Bx)) code = 1;
    else if (covers(Bx, A)) code = 2;
    else {
      const double iw = fmax(0.0, fmin(A.x1, Bx.x1) - fmax(A.x0, Bx.x0)), ih = fmax(0.0, fmin(A.y1, Bx.y1) - fmax(A.y0, Bx.y0));
      const double inter = iw * ih;
      const double areaA = (A.x1 - A.x0) * (A.y1 - A.y0), areaB = (Bx.x1 - Bx.x0) * (Bx.y1 - Bx.y0);
      // reference evaluates IoU(i, j) with i < j: boxAArea + boxBArea - interArea in that order
      const double uni = arow < bcol ? (areaA + areaB) - inter : (areaB + areaA) - inter;
      if (inter / uni >= 0.5) code = 3;
      else {
        const Box& bi = arow < bcol ? A : Bx;
        const Box& bj = arow < bcol ? Bx : A;
        const double dy = 0.5 * (bi.y0 + bi.y1) - 0.5 * (bj.y0 + bj.y1), dx = 0.5 * (bi.x0 + bi.x1) - 0.5 * (bj.x0 + bj.x1);
        if (sqrt(dy * dy + dx * dx) < limit) {
          int cij, cji;
          sector_pair(bi, bj, cij, cji);
          code = arow < bcol ? cij : cji;
        }
      }
    }
  }
  // 12 channels = 3 aligned dwords
  unsigned chan = 0;
  if (code > 0) chan |= 1u << (code - 1);
  if (code >= 4 && code <= 11)
    for (int k = 1; k <= width; ++k) {
      chan |= 1u << (4 + ((code - 4 + k) & 7) - 1);
      chan |= 1u << (4 + ((code - 4 - k) & 7) - 1);
    }
  unsigned w[3] = {0, 0, 0};
#pragma unroll
  for (int h = 0; h < 12; ++h) w[h >> 2] |= ((chan >> h) & 1u) << (8 * (h & 3));
  unsigned* dst = reinterpret_cast<unsigned*>(out + idx * 12);
  dst[0] = w[0]; dst[1] = w[1]; dst[2] = w[2];
}

}  // namespace

extern "C" int sam_spatial_relation_tensor(const double* boxes, int B, int N, int context, double distance_threshold, int8_t* out, void* stream) {
  SAM_REQUIRE(boxes && out, "sam_spatial_relation_tensor: null pointer");
  SAM_REQUIRE(B > 0 && N > 0, "sam_spatial_relation_tensor: empty problem");
  SAM_REQUIRE(context == 1 || context == 3 || context == 5 || context == 7 || context == 9, "sam_spatial_relation_tensor: context must be 1,3,5,7 or 9 (got %d)", context);
  SAM_REQUIRE(((uintptr_t)out % 4) == 0, "sam_spatial_relation_tensor: output must be 4-byte aligned");
  const int64_t total = (int64_t)B * N * N;
  relation_kernel<<<dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream>>>(boxes, B, N, (context - 1) / 2, distance_threshold * sqrt(2.0), out);
  SAM_LAUNCH_CHECK();
  return SAM_OK;
}
