// `output_attentions` (sam/sa_m4c.py:600-609, 765-769): the attention probabilities [B, H, N, N] the fused kernels never materialise, rebuilt AFTER the fact
// from what the forward saved -- q | k rows, the allow bits, the rows' log2-sum-exps and (training) the dropout keep bits:
//     P[b,h,q,k] = allow ? exp2(scale * log2(e) * <q, k> - lse2[b,h,q]) : 0,   times keep / (1 - p),   times head_scale[h] (head_mask, :591-592)
// i.e. softmax(scores + mask) * entity_probs_mask (fully masked rows have lse2 = +inf: exact zeros, :574-584), after dropout and head mask -- what the reference
// returns as `attention_probs`.  A debugging / analysis output: one wave per query row, keys over the lanes, fp32 dot products; nothing here is on the training path.
#include "common.h"
#include "sam_hip.h"

namespace {

__global__ __launch_bounds__(256) void attn_probs_kernel(const bf16_t* qkv, const uint32_t* allow, int64_t allow_sb, int64_t allow_sh, int NW, const float* lse2,
                                                         const uint32_t* keep, const float* head_scale, int N, int H, float scale_log2, float inv_keep, float* out) {
  const int bh = blockIdx.x, b = bh / H, h = bh % H;
  const int q = blockIdx.y * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (q >= N) return;
  const int64_t ld = 3 * (int64_t)H * 64;
  const bf16_t* qrow = qkv + ((int64_t)b * N + q) * ld + h * 64;
  float qf[64];
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    const uint4 u = *reinterpret_cast<const uint4*>(qrow + 8 * c);
    qf[8 * c + 0] = bf_lo(u.x); qf[8 * c + 1] = bf_hi(u.x); qf[8 * c + 2] = bf_lo(u.y); qf[8 * c + 3] = bf_hi(u.y);
    qf[8 * c + 4] = bf_lo(u.z); qf[8 * c + 5] = bf_hi(u.z); qf[8 * c + 6] = bf_lo(u.w); qf[8 * c + 7] = bf_hi(u.w);
  }
  const float l2 = lse2[(int64_t)bh * N + q];
  const uint32_t* arow = allow + b * allow_sb + h * allow_sh + (int64_t)q * NW;
  const uint32_t* krow_bits = keep ? keep + ((int64_t)bh * N + q) * NW : nullptr;
  const float hs = head_scale ? head_scale[h] : 1.0f;
  float* orow = out + ((int64_t)bh * N + q) * N;
  for (int k = lane; k < N; k += 64) {
    const bf16_t* krow = qkv + ((int64_t)b * N + k) * ld + H * 64 + h * 64;
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const uint4 u = *reinterpret_cast<const uint4*>(krow + 8 * c);
      s += qf[8 * c + 0] * bf_lo(u.x) + qf[8 * c + 1] * bf_hi(u.x) + qf[8 * c + 2] * bf_lo(u.y) + qf[8 * c + 3] * bf_hi(u.y) +
           qf[8 * c + 4] * bf_lo(u.z) + qf[8 * c + 5] * bf_hi(u.z) + qf[8 * c + 6] * bf_lo(u.w) + qf[8 * c + 7] * bf_hi(u.w);
    }
    const bool ok = (arow[k >> 5] >> (k & 31)) & 1u;
    float p = ok ? exp2f(s * scale_log2 - l2) : 0.f;          // (dead rows: l2 = +inf -> 0)
    if (krow_bits) p = ((krow_bits[k >> 5] >> (k & 31)) & 1u) ? p * inv_keep : 0.f;
    orow[k] = p * hs;
  }
}

}  // namespace

extern "C" int sam_attn_probs(const void* qkv, const uint32_t* allow, int64_t allow_stride_b, int64_t allow_stride_h, const float* lse2, const uint32_t* keep,
                              const float* head_scale, int B, int N, int H, int head_dim, float scale, float p_drop, float* out, void* stream) {
  SAM_REQUIRE(qkv && allow && lse2 && out, "sam_attn_probs: null pointer");
  SAM_REQUIRE(head_dim == 64, "sam_attn_probs: head_dim must be 64 (got %d)", head_dim);
  SAM_REQUIRE(B > 0 && H > 0 && N > 0, "sam_attn_probs: empty problem");
  const int NW = sam_attn_words_per_row(N);
  SAM_REQUIRE(NW > 0, "sam_attn_probs: N=%d exceeds the fused attention kernels (384)", N);
  SAM_REQUIRE(p_drop >= 0.f && p_drop < 1.f && (p_drop == 0.f || keep), "sam_attn_probs: dropout %.3f needs the forward's keep bits", p_drop);
  float inv_keep = 1.0f;
  if (keep) {
    const unsigned thr16 = dropout_thr16(p_drop);                        // the forward's 16-bit threshold (common.h)
    inv_keep = thr16 ? 1.0f / (1.0f - (float)thr16 / 65536.0f) : 1.0f;
  }
  attn_probs_kernel<<<dim3(B * H, (N + 3) / 4), dim3(256), 0, (hipStream_t)stream>>>((const bf16_t*)qkv, allow, allow_stride_b, allow_stride_h, NW, lse2, keep, head_scale, N, H,
                                                                                     scale * 1.4426950408889634f, inv_keep, out);
  SAM_LAUNCH_CHECK();
  return SAM_OK;
}
