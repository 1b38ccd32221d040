// The tail of the object / OCR input encoders of SAM4C (gfx950), sam/sa_m4c.py:204-224 and :226-263:
//     out = dropout( LN_a(feat W_a^T + b_a) + LN_b(bbox W_b^T + b_b) )
// The wide projection (2048 / 3002 -> 768) stays a GEMM; everything behind it -- the 4 -> 768 box projection, both LayerNorms, the sum and the
// dropout -- is ONE row-wise kernel here, and one more in the backward direction.  Upstream that is six eager ops per encoder; the round-2 path still
// spent nine launches forward (box pack, box GEMM, two LayerNorms, add + dropout, ...) and fourteen backward on rows of 768 values, i.e. on the
// ~5 us a dependent launch costs inside the captured step: the head and the tail of the step are bound by their NUMBER of kernels.
//   forward   one wave per row: z_b = bf16(bbox) . bf16(W_b)^T + b_b in registers (4 FMAs per output), the two rows' statistics (fp32, two-pass),
//             out = dropout(gamma_a xhat_a + beta_a + gamma_b xhat_b + beta_b) rounded to bf16 once; saves (mean_a, rstd_a, mean_b, rstd_b) per row.
//   backward  g = dropout mask regenerated from the counters; LayerNorm backward of both branches in one pass over (dy, z_a); z_b is recomputed from
//             the boxes; d z_a (bf16: the operand of the wide weight gradient) is the only row output; the eight per-column sums -- d gamma_a,
//             d beta (shared: both betas see g), d gamma_b, d b_b, d W_b[:, 0..3] -- go to per-block partial rows and a fixed-order finalize.
#include "common.h"
#include "sam_hip.h"

namespace {

constexpr int ENC_VECS = 8;          // per-column sums of the backward: d gamma_a, d beta, d gamma_b, d b_b, d W_b[:, 0], .., d W_b[:, 3]
constexpr int ENC_MAX_BLOCKS = 256;

__device__ __forceinline__ void ld4bf(const bf16_t* p, float* v) {
  const uint2 x = *reinterpret_cast<const uint2*>(p);
  v[0] = bf_lo(x.x); v[1] = bf_hi(x.x); v[2] = bf_lo(x.y); v[3] = bf_hi(x.y);
}
__device__ __forceinline__ void ld4f(const float* p, float* v) {
  const float4 x = *reinterpret_cast<const float4*>(p);
  v[0] = x.x; v[1] = x.y; v[2] = x.z; v[3] = x.w;
}
__device__ __forceinline__ float bf_round(float x) { return bf2f(f2bf(x)); }
// keep mask of the 4 columns of chunk c of `row`: the (row, col / 8) hidden-state dropout stream (common.h)
__device__ __forceinline__ void keep4(float* v, unsigned row, int c, unsigned thr16, float inv_keep, unsigned seed_lo, unsigned seed_hi, unsigned off_lo, unsigned off_hi) {
  const u32x4 rn = hidden_dropout_bits(row, (unsigned)(c >> 1), off_lo, off_hi, seed_lo, seed_hi);
  const unsigned lo = (c & 1) ? rn.z : rn.x, hi = (c & 1) ? rn.w : rn.y;
  v[0] = (lo & 0xffffu) >= thr16 ? v[0] * inv_keep : 0.f;
  v[1] = (lo >> 16) >= thr16 ? v[1] * inv_keep : 0.f;
  v[2] = (hi & 0xffffu) >= thr16 ? v[2] * inv_keep : 0.f;
  v[3] = (hi >> 16) >= thr16 ? v[3] * inv_keep : 0.f;
}

struct EncArgs {
  const bf16_t* za; int64_t ldza;
  const float* bbox; int64_t ldbox;
  const bf16_t* wb; int64_t ldw;
  const float* bias_b;
  const float *gamma_a, *beta_a, *gamma_b, *beta_b;
  float eps;
  int R, D;
  unsigned thr16; float inv_keep;
  unsigned seed_lo, seed_hi, off_lo, off_hi;
  const unsigned long long* rng_state;
  float* stats;            // [R][4]
};

template <int NCH>
__global__ __launch_bounds__(256) void enc_fwd_kernel(EncArgs a, bf16_t* out, int64_t ldo) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nchunk = a.D >> 2;
  unsigned seed_lo = a.seed_lo, seed_hi = a.seed_hi, off_lo = a.off_lo, off_hi = a.off_hi;
  if (a.thr16) rng_resolve(a.rng_state, seed_lo, seed_hi, off_lo, off_hi);
  float w[NCH][4][4], bb[NCH][4], ga[NCH][4], ba[NCH][4], gb[NCH][4], bt[NCH][4];
#pragma unroll
  for (int j = 0; j < NCH; ++j) {
    const int c = min(lane + 64 * j, nchunk - 1);
#pragma unroll
    for (int e = 0; e < 4; ++e) ld4bf(a.wb + (int64_t)(4 * c + e) * a.ldw, w[j][e]);
    ld4f(a.bias_b + 4 * c, bb[j]); ld4f(a.gamma_a + 4 * c, ga[j]); ld4f(a.beta_a + 4 * c, ba[j]); ld4f(a.gamma_b + 4 * c, gb[j]); ld4f(a.beta_b + 4 * c, bt[j]);
  }
  const float invD = 1.0f / a.D;
  for (int row = blockIdx.x * 4 + wave; row < a.R; row += gridDim.x * 4) {
    float x[NCH][4], z[NCH][4], bx[4];
#pragma unroll
    for (int j = 0; j < NCH; ++j) ld4bf(a.za + (int64_t)row * a.ldza + 4 * min(lane + 64 * j, nchunk - 1), x[j]);
#pragma unroll
    for (int k = 0; k < 4; ++k) bx[k] = bf_round(a.bbox[(int64_t)row * a.ldbox + k]);
    float sa = 0.f, sb = 0.f;
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
      const bool live = lane + 64 * j < nchunk;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        z[j][e] = fmaf(w[j][e][3], bx[3], fmaf(w[j][e][2], bx[2], fmaf(w[j][e][1], bx[1], fmaf(w[j][e][0], bx[0], bb[j][e]))));
        if (!live) x[j][e] = z[j][e] = 0.f;
        sa += x[j][e]; sb += z[j][e];
      }
    }
    const float ma = wave_sum(sa) * invD, mb = wave_sum(sb) * invD;
    float qa = 0.f, qb = 0.f;
#pragma unroll
    for (int j = 0; j < NCH; ++j)
      if (lane + 64 * j < nchunk)
#pragma unroll
        for (int e = 0; e < 4; ++e) { const float da = x[j][e] - ma, db = z[j][e] - mb; qa += da * da; qb += db * db; }
    const float ra = 1.0f / sqrtf(wave_sum(qa) * invD + a.eps), rb = 1.0f / sqrtf(wave_sum(qb) * invD + a.eps);
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
      const int c = lane + 64 * j;
      if (c >= nchunk) continue;
      float o[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = (ga[j][e] * ((x[j][e] - ma) * ra) + ba[j][e]) + (gb[j][e] * ((z[j][e] - mb) * rb) + bt[j][e]);
      if (a.thr16) keep4(o, (unsigned)row, c, a.thr16, a.inv_keep, seed_lo, seed_hi, off_lo, off_hi);
      *reinterpret_cast<uint2*>(out + (int64_t)row * ldo + 4 * c) = make_uint2(pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3]));
    }
    if (lane == 0) *reinterpret_cast<float4*>(a.stats + 4 * (int64_t)row) = make_float4(ma, ra, mb, rb);
  }
}

// ws: [blocks][ENC_VECS][D] partial column sums
template <int NCH>
__global__ __launch_bounds__(256) void enc_bwd_kernel(EncArgs a, const bf16_t* dy, int64_t ldd, bf16_t* dza, int64_t ldo, float* ws) {
  __shared__ float red[4][64 * 4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nchunk = a.D >> 2;
  unsigned seed_lo = a.seed_lo, seed_hi = a.seed_hi, off_lo = a.off_lo, off_hi = a.off_hi;
  if (a.thr16) rng_resolve(a.rng_state, seed_lo, seed_hi, off_lo, off_hi);
  float w[NCH][4][4], bb[NCH][4], ga[NCH][4], gb[NCH][4];
  float acc[ENC_VECS][NCH][4];
#pragma unroll
  for (int j = 0; j < NCH; ++j) {
    const int c = min(lane + 64 * j, nchunk - 1);
#pragma unroll
    for (int e = 0; e < 4; ++e) ld4bf(a.wb + (int64_t)(4 * c + e) * a.ldw, w[j][e]);
    ld4f(a.bias_b + 4 * c, bb[j]); ld4f(a.gamma_a + 4 * c, ga[j]); ld4f(a.gamma_b + 4 * c, gb[j]);
#pragma unroll
    for (int v = 0; v < ENC_VECS; ++v)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[v][j][e] = 0.f;
  }
  const float invD = 1.0f / a.D;
  for (int row = blockIdx.x * 4 + wave; row < a.R; row += gridDim.x * 4) {
    float g[NCH][4], xa[NCH][4], xb[NCH][4], bx[4];
    const float4 st = *reinterpret_cast<const float4*>(a.stats + 4 * (int64_t)row);      // mean_a, rstd_a, mean_b, rstd_b
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
      const int c = min(lane + 64 * j, nchunk - 1);
      ld4bf(dy + (int64_t)row * ldd + 4 * c, g[j]);
      ld4bf(a.za + (int64_t)row * a.ldza + 4 * c, xa[j]);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) bx[k] = bf_round(a.bbox[(int64_t)row * a.ldbox + k]);
    float s1a = 0.f, s2a = 0.f, s1b = 0.f, s2b = 0.f;
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
      const int c = lane + 64 * j;
      const bool live = c < nchunk;
      if (a.thr16) keep4(g[j], (unsigned)row, min(c, nchunk - 1), a.thr16, a.inv_keep, seed_lo, seed_hi, off_lo, off_hi);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float zb = fmaf(w[j][e][3], bx[3], fmaf(w[j][e][2], bx[2], fmaf(w[j][e][1], bx[1], fmaf(w[j][e][0], bx[0], bb[j][e]))));
        if (!live) g[j][e] = 0.f;
        xa[j][e] = live ? (xa[j][e] - st.x) * st.y : 0.f;
        xb[j][e] = live ? (zb - st.z) * st.w : 0.f;
        const float ta = g[j][e] * ga[j][e], tb = g[j][e] * gb[j][e];
        s1a += ta; s2a += ta * xa[j][e]; s1b += tb; s2b += tb * xb[j][e];
        acc[0][j][e] += g[j][e] * xa[j][e];        // d gamma_a
        acc[1][j][e] += g[j][e];                    // d beta_a == d beta_b
        acc[2][j][e] += g[j][e] * xb[j][e];        // d gamma_b
      }
    }
    s1a = wave_sum(s1a) * invD; s2a = wave_sum(s2a) * invD; s1b = wave_sum(s1b) * invD; s2b = wave_sum(s2b) * invD;
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
      const int c = lane + 64 * j;
      if (c >= nchunk) continue;
      float da[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        da[e] = st.y * (g[j][e] * ga[j][e] - s1a - xa[j][e] * s2a);
        const float db = st.w * (g[j][e] * gb[j][e] - s1b - xb[j][e] * s2b);
        acc[3][j][e] += db;                         // d b_b
        acc[4][j][e] += db * bx[0]; acc[5][j][e] += db * bx[1]; acc[6][j][e] += db * bx[2]; acc[7][j][e] += db * bx[3];     // d W_b[:, k]
      }
      *reinterpret_cast<uint2*>(dza + (int64_t)row * ldo + 4 * c) = make_uint2(pack_bf16x2(da[0], da[1]), pack_bf16x2(da[2], da[3]));
    }
  }
  // block reduction over the four waves (fixed order), one partial row per block and vector
#pragma unroll
  for (int v = 0; v < ENC_VECS; ++v)
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
      const int c = lane + 64 * j;
      __syncthreads();
#pragma unroll
      for (int e = 0; e < 4; ++e) red[wave][lane * 4 + e] = acc[v][j][e];
      __syncthreads();
      if (wave == 0 && c < nchunk) {
        float o[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (red[0][lane * 4 + e] + red[1][lane * 4 + e]) + (red[2][lane * 4 + e] + red[3][lane * 4 + e]);
        *reinterpret_cast<float4*>(ws + ((int64_t)blockIdx.x * ENC_VECS + v) * a.D + 4 * c) = make_float4(o[0], o[1], o[2], o[3]);
      }
    }
}

// out (+)= sum over the partial rows, in a fixed order; blockIdx.y = vector (0..7): d gamma_a | d beta (to BOTH betas) | d gamma_b | d b_b | d W_b[:, y - 4]
struct EncOuts { float *dgamma_a, *dbeta_a, *dgamma_b, *dbeta_b, *dbias_b, *dwb; int64_t ldgw; };
// 16 row lanes x 64 columns per block, eight independent, unconditional loads in flight per thread (the first version walked the partial rows with ONE
// dependent load per trip -- 128 L2 round trips in a row: 27 us per call, twice per step on the tail's chain)
constexpr int ENC_FIN_RL = 16;
__global__ __launch_bounds__(64 * ENC_FIN_RL) void enc_finalize_kernel(const float* ws, int nblocks, int D, EncOuts o, int accumulate) {
  __shared__ float red[ENC_FIN_RL][64];
  const int cx = threadIdx.x & 63, ry = threadIdx.x >> 6, v = blockIdx.y, c = min((int)blockIdx.x * 64 + cx, D - 1);
  const float* base = ws + (int64_t)v * D + c;
  const int64_t stride = (int64_t)ENC_VECS * D;
  float s = 0.f;
  for (int r = ry; r < nblocks; r += ENC_FIN_RL * 8) {
    float t[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) t[u] = base[(int64_t)min(r + ENC_FIN_RL * u, nblocks - 1) * stride];
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (r + ENC_FIN_RL * u >= nblocks) t[u] = 0.f;
    s += ((t[0] + t[1]) + (t[2] + t[3])) + ((t[4] + t[5]) + (t[6] + t[7]));
  }
  red[ry][cx] = s;
  __syncthreads();
  if (ry != 0 || (int)blockIdx.x * 64 + cx >= D) return;
  float tot = 0.f;
#pragma unroll
  for (int u = 0; u < ENC_FIN_RL; ++u) tot += red[u][cx];
  if (v == 1) {
    o.dbeta_a[c] = accumulate ? o.dbeta_a[c] + tot : tot;
    o.dbeta_b[c] = accumulate ? o.dbeta_b[c] + tot : tot;
    return;
  }
  float* p = v == 0 ? o.dgamma_a + c : v == 2 ? o.dgamma_b + c : v == 3 ? o.dbias_b + c : o.dwb + (int64_t)c * o.ldgw + (v - 4);
  *p = accumulate ? *p + tot : tot;
}

int fill_args(EncArgs& a, const void* za, int64_t ldza, const float* bbox, int64_t ldbox, const void* wb, int64_t ldw, const float* bias_b, const float* gamma_a,
              const float* gamma_b, float* stats, int R, int D, float p_drop, uint64_t seed, uint64_t offset) {
  SAM_REQUIRE(za && bbox && wb && bias_b && gamma_a && gamma_b && stats, "sam_input_encoder: null pointer");
  SAM_REQUIRE(R > 0 && D > 0 && D % 4 == 0 && D <= 1024 && ldza % 4 == 0 && ldw % 4 == 0 && ldbox >= 4, "sam_input_encoder: need D %% 4 == 0, D <= 1024, aligned strides (R=%d D=%d)", R, D);
  SAM_REQUIRE(p_drop >= 0.f && p_drop < 1.f, "sam_input_encoder: p_drop out of range");
  a.za = (const bf16_t*)za; a.ldza = ldza; a.bbox = bbox; a.ldbox = ldbox; a.wb = (const bf16_t*)wb; a.ldw = ldw; a.bias_b = bias_b;
  a.gamma_a = gamma_a; a.gamma_b = gamma_b; a.stats = stats; a.R = R; a.D = D;
  a.thr16 = dropout_thr16(p_drop);
  a.inv_keep = a.thr16 ? 1.0f / (1.0f - (float)a.thr16 / 65536.0f) : 1.0f;
  a.seed_lo = (unsigned)seed; a.seed_hi = (unsigned)(seed >> 32); a.off_lo = (unsigned)offset; a.off_hi = (unsigned)(offset >> 32);
  a.rng_state = sam_get_rng_state();
  return SAM_OK;
}

}  // namespace

extern "C" int sam_input_encoder_fwd(const void* za, int64_t ldza, const float* bbox, int64_t ldbox, const void* wb, int64_t ldw, const float* bias_b,
                                     const float* gamma_a, const float* beta_a, const float* gamma_b, const float* beta_b, float eps, int R, int D, float p_drop,
                                     uint64_t seed, uint64_t offset, void* out, int64_t ldo, float* stats, void* stream) {
  EncArgs a = {};
  if (int rc = fill_args(a, za, ldza, bbox, ldbox, wb, ldw, bias_b, gamma_a, gamma_b, stats, R, D, p_drop, seed, offset)) return rc;
  SAM_REQUIRE(beta_a && beta_b && out && ldo % 4 == 0, "sam_input_encoder_fwd: null pointer / output stride");
  a.beta_a = beta_a; a.beta_b = beta_b; a.eps = eps;
  const int blocks = min((R + 3) / 4, 2048), nch = (D / 4 + 63) / 64;
  hipStream_t st = (hipStream_t)stream;
  if (nch <= 1) enc_fwd_kernel<1><<<dim3(blocks), dim3(256), 0, st>>>(a, (bf16_t*)out, ldo);
  else if (nch == 2) enc_fwd_kernel<2><<<dim3(blocks), dim3(256), 0, st>>>(a, (bf16_t*)out, ldo);
  else if (nch == 3) enc_fwd_kernel<3><<<dim3(blocks), dim3(256), 0, st>>>(a, (bf16_t*)out, ldo);
  else enc_fwd_kernel<4><<<dim3(blocks), dim3(256), 0, st>>>(a, (bf16_t*)out, ldo);
  SAM_LAUNCH_CHECK();
  return SAM_OK;
}

static int enc_bwd_blocks(int R) { return min((R + 3) / 4, ENC_MAX_BLOCKS); }
extern "C" int64_t sam_input_encoder_bwd_ws_bytes(int R, int D) { return (int64_t)enc_bwd_blocks(R) * ENC_VECS * D * 4; }

extern "C" int sam_input_encoder_bwd(const void* dy, int64_t ldd, const void* za, int64_t ldza, const float* bbox, int64_t ldbox, const void* wb, int64_t ldw,
                                     const float* bias_b, const float* gamma_a, const float* gamma_b, const float* stats, int R, int D, float p_drop, uint64_t seed,
                                     uint64_t offset, void* dza, int64_t ldo, float* dgamma_a, float* dbeta_a, float* dgamma_b, float* dbeta_b, float* dbias_b, float* dwb,
                                     int64_t ldgw, int accumulate, float* ws, void* stream) {
  EncArgs a = {};
  if (int rc = fill_args(a, za, ldza, bbox, ldbox, wb, ldw, bias_b, gamma_a, gamma_b, const_cast<float*>(stats), R, D, p_drop, seed, offset)) return rc;
  SAM_REQUIRE(dy && dza && ws && dgamma_a && dbeta_a && dgamma_b && dbeta_b && dbias_b && dwb && ldd % 4 == 0 && ldo % 4 == 0 && ldgw >= 4, "sam_input_encoder_bwd: null pointer / stride");
  const int blocks = enc_bwd_blocks(R), nch = (D / 4 + 63) / 64;
  hipStream_t st = (hipStream_t)stream;
  if (nch <= 1) enc_bwd_kernel<1><<<dim3(blocks), dim3(256), 0, st>>>(a, (const bf16_t*)dy, ldd, (bf16_t*)dza, ldo, ws);
  else if (nch == 2) enc_bwd_kernel<2><<<dim3(blocks), dim3(256), 0, st>>>(a, (const bf16_t*)dy, ldd, (bf16_t*)dza, ldo, ws);
  else if (nch == 3) enc_bwd_kernel<3><<<dim3(blocks), dim3(256), 0, st>>>(a, (const bf16_t*)dy, ldd, (bf16_t*)dza, ldo, ws);
  else enc_bwd_kernel<4><<<dim3(blocks), dim3(256), 0, st>>>(a, (const bf16_t*)dy, ldd, (bf16_t*)dza, ldo, ws);
  EncOuts o = {dgamma_a, dbeta_a, dgamma_b, dbeta_b, dbias_b, dwb, ldgw};
  enc_finalize_kernel<<<dim3((D + 63) / 64, ENC_VECS), dim3(64 * ENC_FIN_RL), 0, st>>>(ws, blocks, D, o, accumulate);
  SAM_LAUNCH_CHECK();
  return SAM_OK;
}
