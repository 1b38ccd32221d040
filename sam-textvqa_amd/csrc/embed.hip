// Front-end kernels of the SA-M4C step (gfx950): everything that happens to the inputs before the first encoder layer and is
// neither a GEMM nor a LayerNorm.  All row-parallel, HBM/launch bound, 8- or 16-byte accesses, one pass.
//   l2norm_pack    <- F.normalize(feat, dim=-1) of the Faster-RCNN / FastText / PHOC features and the torch.cat that builds
//                     the OCR input row, sam/sa_m4c.py:217-253: normalised, rounded to bf16 and written at a column offset of
//                     the (K-padded) GEMM operand in one pass
//   embed_sum      <- BertEmbeddings.forward (words + positions + token types, pytorch-transformers) and the position/type
//                     half of PrevPredEmbeddings.forward, sam/sa_m4c.py:932-945
//   gather2_add    <- _batch_gather over cat([ans_emb, ocr_emb]) + the embedding dropout and sum, sam/sa_m4c.py:921-948,
//                     without materialising the [B, V + n_ocr, D] table
#include "common.h"
#include "sam_hip.h"

namespace {

__device__ __forceinline__ void ld4f(const float* p, float* v) {
  const float4 x = *reinterpret_cast<const float4*>(p);
  v[0] = x.x; v[1] = x.y; v[2] = x.z; v[3] = x.w;
}
__device__ __forceinline__ void ld4bf(const bf16_t* p, float* v) {
  const uint2 x = *reinterpret_cast<const uint2*>(p);
  v[0] = bf_lo(x.x); v[1] = bf_hi(x.x); v[2] = bf_lo(x.y); v[3] = bf_hi(x.y);
}
__device__ __forceinline__ void st4bf(bf16_t* p, const float* v) {
  *reinterpret_cast<uint2*>(p) = make_uint2(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]));
}
// keep mask of 4 consecutive columns (chunk c = col / 4) of `row`: the (row, col / 8) Philox stream the GEMM / LN epilogues use
__device__ __forceinline__ void dropout4(float* v, unsigned row, int c, unsigned thr16, float inv_keep, unsigned seed_lo, unsigned seed_hi,
                                         unsigned off_lo, unsigned off_hi) {
  const u32x4 rn = hidden_dropout_bits(row, (unsigned)(c >> 1), off_lo, off_hi, seed_lo, seed_hi);
  const unsigned lo = (c & 1) ? rn.z : rn.x, hi = (c & 1) ? rn.w : rn.y;
  v[0] = (lo & 0xffffu) >= thr16 ? v[0] * inv_keep : 0.f;
  v[1] = (lo >> 16) >= thr16 ? v[1] * inv_keep : 0.f;
  v[2] = (hi & 0xffffu) >= thr16 ? v[2] * inv_keep : 0.f;
  v[3] = (hi >> 16) >= thr16 ? v[3] * inv_keep : 0.f;
}

// one wave per row; the row (<= 8 KB) is read twice, the second time from L1/L2
__global__ __launch_bounds__(256) void l2norm_pack_kernel(const float* x, int64_t ldx, int M, int D, int normalize, float eps, bf16_t* out, int64_t ldo,
                                                          int col0, int zero_upto) {
  const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  const float* xr = x + (int64_t)row * ldx;
  bf16_t* orow = out + (int64_t)row * ldo;
  const int nchunk = D >> 2;
  float scale = 1.f;
  if (normalize) {
    float q = 0.f;
    for (int c = lane; c < nchunk; c += 64) {
      float v[4];
      ld4f(xr + 4 * c, v);
      q += (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
    }
    scale = 1.0f / fmaxf(sqrtf(wave_sum(q)), eps);       // x / max(||x||, eps)
  }
  for (int c = lane; c < nchunk; c += 64) {
    float v[4];
    ld4f(xr + 4 * c, v);
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] *= scale;
    st4bf(orow + col0 + 4 * c, v);
  }
  for (int c = col0 + D + lane; c < zero_upto; c += 64) orow[c] = 0;
}

// rows of up to 2048 values: the whole row in registers (all of its loads in flight at once, one pass over memory); unconditional loads at clamped
// chunk indices, as in the LayerNorm kernels
template <int NCH>
__global__ __launch_bounds__(256) void l2norm_pack_reg_kernel(const float* x, int64_t ldx, int M, int D, int normalize, float eps, bf16_t* out, int64_t ldo,
                                                              int col0, int zero_upto) {
  const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  const float* xr = x + (int64_t)row * ldx;
  bf16_t* orow = out + (int64_t)row * ldo;
  const int nchunk = D >> 2;
  float v[NCH][4];
#pragma unroll
  for (int j = 0; j < NCH; ++j) ld4f(xr + 4 * min(lane + 64 * j, nchunk - 1), v[j]);
  float scale = 1.f;
  if (normalize) {
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < NCH; ++j)
      if (lane + 64 * j < nchunk) q += (v[j][0] * v[j][0] + v[j][1] * v[j][1]) + (v[j][2] * v[j][2] + v[j][3] * v[j][3]);
    scale = 1.0f / fmaxf(sqrtf(wave_sum(q)), eps);       // x / max(||x||, eps)
  }
#pragma unroll
  for (int j = 0; j < NCH; ++j) {
    const int c = lane + 64 * j;
    if (c < nchunk) {
#pragma unroll
      for (int e = 0; e < 4; ++e) v[j][e] *= scale;
      st4bf(orow + col0 + 4 * c, v[j]);
    }
  }
  for (int c = col0 + D + lane; c < zero_upto; c += 64) orow[c] = 0;
}

// out[r, :] = table[ids[r], :] (bf16 rows, optional) + pos[r % S, :] + tt[type[r], :]   (fp32 out: the LayerNorm input)
__global__ __launch_bounds__(256) void embed_sum_fwd_kernel(const bf16_t* table, int64_t ld_table, const int64_t* ids, int table_rows, const float* pos,
                                                            int64_t ld_pos, int S, const float* tt, int64_t ld_tt, const uint8_t* type_ids, int n_types,
                                                            int R, int D, float* out, int64_t ldo) {
  const int lane = threadIdx.x & 63, r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= R) return;
  const int t = type_ids ? min((int)type_ids[r], n_types - 1) : 0;
  int64_t id = ids ? ids[r] : 0;
  id = id < 0 ? 0 : (id >= table_rows ? table_rows - 1 : id);
  const float* pr = pos + (int64_t)(r % S) * ld_pos;
  const float* tr = tt + (int64_t)t * ld_tt;
  for (int c = lane; 4 * c < D; c += 64) {
    float a[4], b[4], w[4] = {0.f, 0.f, 0.f, 0.f};
    ld4f(pr + 4 * c, a);
    ld4f(tr + 4 * c, b);
    if (table) ld4bf(table + id * ld_table + 4 * c, w);
    *reinterpret_cast<float4*>(out + (int64_t)r * ldo + 4 * c) = make_float4((w[0] + a[0]) + b[0], (w[1] + a[1]) + b[1], (w[2] + a[2]) + b[2], (w[3] + a[3]) + b[3]);
  }
}

// d_pos[s, :] += sum_b d[b*S + s, :]  and  ws[s][t][:] = sum_{b: type[b*S+s] == t} d[b*S + s, :]   (block (s, 64-chunk column group):
// 64 chunk lanes x 4 row lanes, fixed summation order => deterministic); embed_type_finalize_kernel then adds the S partials
// of each type row into d_tt.
constexpr int EMBED_MAX_TYPES = 4;
__global__ __launch_bounds__(256) void embed_sum_bwd_kernel(const bf16_t* d, int64_t ldd, int R, int D, int S, const uint8_t* type_ids, int n_types,
                                                            float* d_pos, int64_t ld_pos, float* ws) {
  __shared__ float red[4][EMBED_MAX_TYPES][64 * 4];
  const int s = blockIdx.x, cx = threadIdx.x & 63, ry = threadIdx.x >> 6, c4 = blockIdx.y * 64 + cx;
  const bool live = 4 * c4 < D;
  float acc[EMBED_MAX_TYPES][4];
#pragma unroll
  for (int t = 0; t < EMBED_MAX_TYPES; ++t) acc[t][0] = acc[t][1] = acc[t][2] = acc[t][3] = 0.f;
  if (live)
    for (int r0 = s + ry * S; r0 < R; r0 += 16 * S) {
      // four rows per trip, all eight loads issued before the first add (a trip per row was 16 dependent HBM / L2 round trips per thread on a 60-block
      // grid: 30 us on the tail's chain for 2 MB); the rows are still added in ascending order
      int tt[4];
      float v[4][4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int r = r0 + 4 * S * k;
        tt[k] = -1;
        if (r < R) {
          tt[k] = type_ids ? type_ids[r] : 0;
          ld4bf(d + (int64_t)r * ldd + 4 * c4, v[k]);
        }
      }
#pragma unroll
      for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int u = 0; u < EMBED_MAX_TYPES; ++u)
          if (tt[k] == u) { acc[u][0] += v[k][0]; acc[u][1] += v[k][1]; acc[u][2] += v[k][2]; acc[u][3] += v[k][3]; }
    }
#pragma unroll
  for (int t = 0; t < EMBED_MAX_TYPES; ++t)
#pragma unroll
    for (int e = 0; e < 4; ++e) red[ry][t][cx * 4 + e] = acc[t][e];
  __syncthreads();
  if (ry == 0 && live) {
    float tot[4] = {0.f, 0.f, 0.f, 0.f};
    for (int t = 0; t < n_types; ++t) {
      float o[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        o[e] = (red[0][t][cx * 4 + e] + red[1][t][cx * 4 + e]) + (red[2][t][cx * 4 + e] + red[3][t][cx * 4 + e]);
        tot[e] += o[e];
      }
      *reinterpret_cast<float4*>(ws + ((int64_t)s * n_types + t) * D + 4 * c4) = make_float4(o[0], o[1], o[2], o[3]);
    }
    float4* dst = reinterpret_cast<float4*>(d_pos + (int64_t)s * ld_pos + 4 * c4);
    float4 o = *dst;
    o.x += tot[0]; o.y += tot[1]; o.z += tot[2]; o.w += tot[3];
    *dst = o;
  }
}
__global__ __launch_bounds__(256) void embed_type_finalize_kernel(const float* ws, int S, int n_types, int D, float* d_tt, int64_t ld_tt) {
  const int t = blockIdx.x, c = blockIdx.y * 256 + threadIdx.x;
  if (c >= D) return;
  float a = 0.f;
  for (int s0 = 0; s0 < S; s0 += 8) {          // eight partials requested at a time, added in the same order
    float v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = s0 + k < S ? ws[((int64_t)(s0 + k) * n_types + t) * D + c] : 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k)
      if (s0 + k < S) a += v[k];
  }
  d_tt[(int64_t)t * ld_tt + c] += a;
}

struct Gather2Args {
  const bf16_t* ans; int64_t ld_ans; int V;
  const bf16_t* ocr; int64_t ld_ocr; int n_ocr;
  const int64_t* inds; int B, S, D;
  unsigned thr16; float inv_keep; unsigned seed_lo, seed_hi, off_lo, off_hi;
  const unsigned long long* rng_state;
};
__device__ __forceinline__ int64_t clamp_ind(int64_t i, int n) { return i < 0 ? 0 : (i >= n ? n - 1 : i); }

// out[r, :] = (ind < V ? ans[ind] : ocr[b * n_ocr + ind - V]) + dropout(emb[r, :])
__global__ __launch_bounds__(256) void gather2_add_fwd_kernel(Gather2Args a, const bf16_t* emb, int64_t ld_emb, bf16_t* out, int64_t ldo) {
  const int lane = threadIdx.x & 63, r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= a.B * a.S) return;
  if (a.thr16) rng_resolve(a.rng_state, a.seed_lo, a.seed_hi, a.off_lo, a.off_hi);
  const int64_t ind = clamp_ind(a.inds[r], a.V + a.n_ocr);
  const bf16_t* src = ind < a.V ? a.ans + ind * a.ld_ans : a.ocr + ((int64_t)(r / a.S) * a.n_ocr + (ind - a.V)) * a.ld_ocr;
  for (int c = lane; 4 * c < a.D; c += 64) {
    float s[4], e[4] = {0.f, 0.f, 0.f, 0.f};
    ld4bf(src + 4 * c, s);
    if (emb) {
      ld4bf(emb + (int64_t)r * ld_emb + 4 * c, e);
      if (a.thr16) dropout4(e, (unsigned)r, c, a.thr16, a.inv_keep, a.seed_lo, a.seed_hi, a.off_lo, a.off_hi);
    }
    const float o[4] = {s[0] + e[0], s[1] + e[1], s[2] + e[2], s[3] + e[3]};
    st4bf(out + (int64_t)r * ldo + 4 * c, o);
  }
}

// d_ans[ind, :] += dy[r, :]  or  d_ocr[b * n_ocr + ind - V, :] += dy[r, :]  (fp32 hardware atomics: rows repeat);
// d_emb[r, :] = dropout mask applied to dy[r, :]
__global__ __launch_bounds__(256) void gather2_add_bwd_kernel(Gather2Args a, const bf16_t* dy, int64_t ldd, float* d_ans, int64_t ld_dans, float* d_ocr,
                                                              int64_t ld_docr, bf16_t* d_emb, int64_t ld_demb) {
  const int lane = threadIdx.x & 63, r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= a.B * a.S) return;
  if (a.thr16) rng_resolve(a.rng_state, a.seed_lo, a.seed_hi, a.off_lo, a.off_hi);
  const int64_t ind = clamp_ind(a.inds[r], a.V + a.n_ocr);
  float* dst = ind < a.V ? d_ans + ind * ld_dans : d_ocr + ((int64_t)(r / a.S) * a.n_ocr + (ind - a.V)) * ld_docr;
  for (int c = lane; 4 * c < a.D; c += 64) {
    float v[4];
    ld4bf(dy + (int64_t)r * ldd + 4 * c, v);
#pragma unroll
    for (int e = 0; e < 4; ++e) unsafeAtomicAdd(dst + 4 * c + e, v[e]);
    if (d_emb) {
      if (a.thr16) dropout4(v, (unsigned)r, c, a.thr16, a.inv_keep, a.seed_lo, a.seed_hi, a.off_lo, a.off_hi);
      st4bf(d_emb + (int64_t)r * ld_demb + 4 * c, v);
    }
  }
}

int fill_gather_args(Gather2Args& a, const void* ans, int64_t ld_ans, int V, const void* ocr, int64_t ld_ocr, int n_ocr, const int64_t* inds, int B, int S,
                     int D, float p_drop, uint64_t seed, uint64_t offset) {
  SAM_REQUIRE(ans && ocr && inds, "sam_gather2_add: null pointer");
  SAM_REQUIRE(B > 0 && S > 0 && V > 0 && n_ocr > 0 && D > 0 && D % 4 == 0 && ld_ans % 4 == 0 && ld_ocr % 4 == 0, "sam_gather2_add: bad shape B=%d S=%d D=%d", B, S, D);
  SAM_REQUIRE(p_drop >= 0.f && p_drop < 1.f, "sam_gather2_add: p_drop out of range");
  const unsigned thr16 = dropout_thr16(p_drop);
  a = Gather2Args{(const bf16_t*)ans, ld_ans, V, (const bf16_t*)ocr, ld_ocr, n_ocr, inds, B, S, D, thr16, thr16 ? 1.0f / (1.0f - (float)thr16 / 65536.0f) : 1.0f,
                  (unsigned)seed, (unsigned)(seed >> 32), (unsigned)offset, (unsigned)(offset >> 32), sam_get_rng_state()};
  return SAM_OK;
}

// scalar twin for rows that are not 16-byte aligned / not a multiple of 4 wide (the 4 box coordinates sliced out of [.., 5] rows)
__global__ __launch_bounds__(256) void l2norm_pack_scalar_kernel(const float* x, int64_t ldx, int M, int D, int normalize, float eps, bf16_t* out, int64_t ldo,
                                                                 int col0, int zero_upto) {
  const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  const float* xr = x + (int64_t)row * ldx;
  bf16_t* orow = out + (int64_t)row * ldo;
  float scale = 1.f;
  if (normalize) {
    float q = 0.f;
    for (int c = lane; c < D; c += 64) q += xr[c] * xr[c];
    scale = 1.0f / fmaxf(sqrtf(wave_sum(q)), eps);
  }
  for (int c = lane; c < D; c += 64) orow[col0 + c] = f2bf(xr[c] * scale);
  for (int c = col0 + D + lane; c < zero_upto; c += 64) orow[c] = 0;
}

}  // namespace

extern "C" int sam_l2norm_pack_bf16(const float* x, int64_t ldx, int M, int D, int normalize, float eps, void* out, int64_t ldo, int col0, int zero_upto,
                                    void* stream) {
  SAM_REQUIRE(x && out, "sam_l2norm_pack_bf16: null pointer");
  SAM_REQUIRE(M > 0 && D > 0 && col0 >= 0 && col0 + D <= ldo && zero_upto <= ldo, "sam_l2norm_pack_bf16: need col0 + D <= ldo (M=%d D=%d col0=%d ldo=%ld)", M, D, col0,
              (long)ldo);
  const bool vec = D % 4 == 0 && ldx % 4 == 0 && ldo % 4 == 0 && col0 % 4 == 0 && ((uintptr_t)x % 16) == 0 && ((uintptr_t)out % 8) == 0;
  const int nch = (D / 4 + 63) / 64;
  const dim3 grid((M + 3) / 4), blk(256);
  hipStream_t st = (hipStream_t)stream;
  if (vec && nch <= 2) l2norm_pack_reg_kernel<2><<<grid, blk, 0, st>>>(x, ldx, M, D, normalize, eps, (bf16_t*)out, ldo, col0, zero_upto);
  else if (vec && nch <= 4) l2norm_pack_reg_kernel<4><<<grid, blk, 0, st>>>(x, ldx, M, D, normalize, eps, (bf16_t*)out, ldo, col0, zero_upto);
  else if (vec && nch <= 8) l2norm_pack_reg_kernel<8><<<grid, blk, 0, st>>>(x, ldx, M, D, normalize, eps, (bf16_t*)out, ldo, col0, zero_upto);
  else if (vec)
    l2norm_pack_kernel<<<dim3((M + 3) / 4), dim3(256), 0, (hipStream_t)stream>>>(x, ldx, M, D, normalize, eps, (bf16_t*)out, ldo, col0, zero_upto);
  else   // unaligned / odd-width rows (box coordinates): scalar accesses
    l2norm_pack_scalar_kernel<<<dim3((M + 3) / 4), dim3(256), 0, (hipStream_t)stream>>>(x, ldx, M, D, normalize, eps, (bf16_t*)out, ldo, col0, zero_upto);
  SAM_LAUNCH_CHECK();
  return SAM_OK;
}

extern "C" int sam_embed_sum_fwd(const void* table, int64_t ld_table, const int64_t* ids, int table_rows, const float* pos, int64_t ld_pos, int S,
                                 const float* tt, int64_t ld_tt, const uint8_t* type_ids, int n_types, int R, int D, float* out, int64_t ldo,
                                 void* stream) {
  SAM_REQUIRE(pos && tt && out && (!table || ids), "sam_embed_sum_fwd: null pointer");
  SAM_REQUIRE(R > 0 && S > 0 && n_types > 0 && D > 0 && D % 4 == 0 && ld_pos % 4 == 0 && ld_tt % 4 == 0 && ldo % 4 == 0 && (!table || (ld_table % 4 == 0 && table_rows > 0)),
              "sam_embed_sum_fwd: bad shape R=%d S=%d D=%d", R, S, D);
  embed_sum_fwd_kernel<<<dim3((R + 3) / 4), dim3(256), 0, (hipStream_t)stream>>>((const bf16_t*)table, ld_table, ids, table_rows, pos, ld_pos, S, tt, ld_tt,
                                                                                 type_ids, n_types, R, D, out, ldo);
  SAM_LAUNCH_CHECK();
  return SAM_OK;
}

extern "C" int64_t sam_embed_sum_bwd_ws_bytes(int S, int n_types, int D) { return (int64_t)S * n_types * D * sizeof(float); }

extern "C" int sam_embed_sum_bwd(const void* d, int64_t ldd, int R, int D, int S, const uint8_t* type_ids, int n_types, float* d_pos, int64_t ld_pos,
                                 float* d_tt, int64_t ld_tt, float* ws, void* stream) {
  SAM_REQUIRE(d && d_pos && d_tt && ws, "sam_embed_sum_bwd: null pointer");
  SAM_REQUIRE(R > 0 && S > 0 && n_types > 0 && n_types <= EMBED_MAX_TYPES && D > 0 && D % 4 == 0 && ldd % 4 == 0 && ld_pos % 4 == 0,
              "sam_embed_sum_bwd: bad shape R=%d S=%d D=%d n_types=%d (<= %d)", R, S, D, n_types, EMBED_MAX_TYPES);
  hipStream_t st = (hipStream_t)stream;
  embed_sum_bwd_kernel<<<dim3(S, (D / 4 + 63) / 64), dim3(256), 0, st>>>((const bf16_t*)d, ldd, R, D, S, type_ids, n_types, d_pos, ld_pos, ws);
  embed_type_finalize_kernel<<<dim3(n_types, (D + 255) / 256), dim3(256), 0, st>>>(ws, S, n_types, D, d_tt, ld_tt);
  SAM_LAUNCH_CHECK();
  return SAM_OK;
}

extern "C" int sam_gather2_add_fwd(const void* ans, int64_t ld_ans, int V, const void* ocr, int64_t ld_ocr, int n_ocr, const int64_t* inds, int B, int S, int D,
                                   const void* emb, int64_t ld_emb, float p_drop, uint64_t seed, uint64_t offset, void* out, int64_t ldo, void* stream) {
  Gather2Args a;
  if (int rc = fill_gather_args(a, ans, ld_ans, V, ocr, ld_ocr, n_ocr, inds, B, S, D, p_drop, seed, offset)) return rc;
  SAM_REQUIRE(out && ldo % 4 == 0 && (!emb || ld_emb % 4 == 0), "sam_gather2_add_fwd: bad output / emb");
  gather2_add_fwd_kernel<<<dim3((B * S + 3) / 4), dim3(256), 0, (hipStream_t)stream>>>(a, (const bf16_t*)emb, ld_emb, (bf16_t*)out, ldo);
  SAM_LAUNCH_CHECK();
  return SAM_OK;
}

extern "C" int sam_gather2_add_bwd(const void* dy, int64_t ldd, int V, int n_ocr, const int64_t* inds, int B, int S, int D, float* d_ans, int64_t ld_dans,
                                   float* d_ocr, int64_t ld_docr, float p_drop, uint64_t seed, uint64_t offset, void* d_emb, int64_t ld_demb, void* stream) {
  Gather2Args a;
  if (int rc = fill_gather_args(a, dy, 4, V, dy, 4, n_ocr, inds, B, S, D, p_drop, seed, offset)) return rc;
  SAM_REQUIRE(dy && d_ans && d_ocr && ldd % 4 == 0 && (!d_emb || ld_demb % 4 == 0), "sam_gather2_add_bwd: bad pointers / strides");
  gather2_add_bwd_kernel<<<dim3((B * S + 3) / 4), dim3(256), 0, (hipStream_t)stream>>>(a, (const bf16_t*)dy, ldd, d_ans, ld_dans, d_ocr, ld_docr, (bf16_t*)d_emb, ld_demb);
  SAM_LAUNCH_CHECK();
  return SAM_OK;
}
