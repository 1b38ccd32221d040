// Shared device helpers for the SA-M4C hot-path kernels (gfx950 / CDNA4 only).
// Hardware lane maps used below were pinned on an MI355X by tools/probes/probe_layouts.hip:
//   v_mfma_f32_16x16x32_bf16:  A[i=l&15][k=8*(l>>4)+e]  B[k=8*(l>>4)+e][j=l&15]  D[i=4*(l>>4)+r][j=l&15]
//   ds_read_b64_tr_b16: inside each 16-lane group, lane i receives column i of the 4x16 block whose
//   row r is supplied (as four 8-byte pieces) by lanes 4r..4r+3.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef unsigned short bf16_t;  // raw bf16 storage
typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) __bf16 hw_bf16x2;
typedef __attribute__((ext_vector_type(2))) float f32x2;

#define SAM_OK 0
#define SAM_ERR_ARG (-1)        // bad shape / alignment / null pointer
#define SAM_ERR_UNSUPPORTED (-2)  // valid request this build has no kernel for

extern "C" void sam_set_error(const char* fmt, ...);
// dropout RNG state in DEVICE memory (thread-local pointer set by sam_set_rng_state; NULL = seeds and offsets are taken by value only)
extern "C" const unsigned long long* sam_get_rng_state(void);

#define SAM_REQUIRE(cond, ...)                    \
  do {                                            \
    if (!(cond)) {                                \
      sam_set_error(__VA_ARGS__);                 \
      return SAM_ERR_ARG;                         \
    }                                             \
  } while (0)

#define SAM_LAUNCH_CHECK()                                                       \
  do {                                                                           \
    hipError_t e_ = hipGetLastError();                                           \
    if (e_ != hipSuccess) {                                                      \
      sam_set_error("%s:%d launch failed: %s", __FILE__, __LINE__, hipGetErrorString(e_)); \
      return (int)e_;                                                            \
    }                                                                            \
  } while (0)

__device__ __forceinline__ float bf2f(bf16_t h) { return __uint_as_float(((unsigned)h) << 16); }
__device__ __forceinline__ unsigned pack_bf16x2(float lo, float hi) {  // RNE, v_cvt_pk_bf16_f32
  f32x2 f = {lo, hi};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(f, hw_bf16x2));
}
__device__ __forceinline__ bf16_t f2bf(float f) { return (bf16_t)(pack_bf16x2(f, 0.f) & 0xffffu); }
__device__ __forceinline__ float bf_lo(unsigned u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float bf_hi(unsigned u) { return __uint_as_float(u & 0xffff0000u); }

// LDS transpose read: 4 bf16 (8 bytes) per lane, see header comment.
__device__ __forceinline__ s16x4 lds_read_tr16(const void* p) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p);
}
__device__ __forceinline__ bf16x8 cat4(s16x4 a, s16x4 b) {
  bf16x8 r;
  r[0] = a[0]; r[1] = a[1]; r[2] = a[2]; r[3] = a[3];
  r[4] = b[0]; r[5] = b[1]; r[6] = b[2]; r[7] = b[3];
  return r;
}

// reductions across the four 16-lane groups that share (lane & 15).  v_permlane16_swap / v_permlane32_swap (gfx950) hand every lane its partner's value
// through the VALU: __shfl_xor compiles to ds_bpermute_b32, an LDS round trip (~100 cycles) on the critical path of every softmax row.
//   permlane16_swap(v, v) -> {rows 0,0,2,2 | rows 1,1,3,3}: combining the two results reduces over the row pair (lane ^ 16);
//   permlane32_swap(v, v) -> {lower half twice | upper half twice}: reduces over lane ^ 32.
__device__ __forceinline__ float xgroup_max(float v) {
  auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  v = fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1]));
  auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return fmaxf(__uint_as_float(b[0]), __uint_as_float(b[1]));
}
__device__ __forceinline__ float xgroup_sum(float v) {
  auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  v = __uint_as_float(a[0]) + __uint_as_float(a[1]);
  auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}
__device__ __forceinline__ unsigned xgroup_or(unsigned v) {
  auto a = __builtin_amdgcn_permlane16_swap(v, v, false, false);
  v = a[0] | a[1];
  auto b = __builtin_amdgcn_permlane32_swap(v, v, false, false);
  return b[0] | b[1];
}
// Sum over the 64 lanes through the VALU (LayerNorm forward / backward), result in all of them: the xor-butterfly 32, 16, 8, 4, 2, 1 -- the SAME additions in the same order as
//   for (o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
// (every step adds a lane's own value and its partner's: commutative, so which of the two is "own" does not matter), hence bit-identical sums,
// but through the VALU: v_permlane32_swap / v_permlane16_swap for the half and row partners, DPP inside a row of 16 (row_ror:8 = lane ^ 8;
// lane ^ 4 = row_shl:4 into banks 0, 2 and row_shr:4 into banks 1, 3; quad_perm for lane ^ 2, lane ^ 1).  __shfl_xor is ds_bpermute_b32: six
// dependent LDS round trips (~100 cycles each) per reduction, on the critical path between a row's loads and its stores in every row kernel.
__device__ __forceinline__ float wave_sum_v(float v) {
  auto h = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  v = __uint_as_float(h[0]) + __uint_as_float(h[1]);
  auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  v = __uint_as_float(r[0]) + __uint_as_float(r[1]);
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x128, 0xf, 0xf, false));
  int t = __builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x104, 0xf, 0x5, false);
  t = __builtin_amdgcn_update_dpp(t, __float_as_int(v), 0x114, 0xf, 0xa, false);
  v += __int_as_float(t);
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xf, 0xf, false));
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xf, 0xf, false));
  return v;
}

// Captured (hipGraph) launches freeze their by-value arguments: with a device-side state {seed, offset_base} every replay still draws fresh
// masks -- the effective key is state[0] (when non-zero) and the effective offset is state[1] + the by-value offset of the dropout site.
__device__ __forceinline__ void rng_resolve(const unsigned long long* st, unsigned& seed_lo, unsigned& seed_hi, unsigned& off_lo, unsigned& off_hi) {
  if (st) {
    const unsigned long long s = st[0], o = st[1] + (((unsigned long long)off_hi << 32) | off_lo);
    if (s) { seed_lo = (unsigned)s; seed_hi = (unsigned)(s >> 32); }
    off_lo = (unsigned)o; off_hi = (unsigned)(o >> 32);
  }
}

struct u32x4 { unsigned x, y, z, w; };
// Counter-based dropout bits: same (seed, offset, row, column group) -> same bits in the forward and in the backward pass.
__device__ __forceinline__ unsigned mix32(unsigned x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}
// Round-4 generator: the expensive mixing (mix32: two 32-bit multiplies) happens once per ROW (dropout_row_key; loop-invariant wherever a thread or wave stays
// on one row: the attention strips, the LayerNorm backward, the GEMM epilogue's row loop); a draw of 8 x 16 bits for (row, column group) then costs full-rate
// operations only -- two 24-bit multiplies (v_mul_u32_u24 / v_mad_u32_u24) around xor-shifts per 32-bit word.  The generator of rounds 1-3 (four lowbias32 mixes per draw: 13 quarter-rate
// 32-bit multiplies) was a fifth of the attention forward's VALU time and 3 us of a 17 us LayerNorm backward.  Keep rate, independence across rows / columns /
// heads / samples / offsets / seeds, avalanche and a duplicate census were checked on the CPU against the old generator (tools/debug/hash_eval.py) and are
// tested on the device (tests/test_attention_gpu.py::test_dropout_streams_are_independent_...).
__device__ __forceinline__ unsigned dropout_row_key(unsigned row, unsigned off_lo, unsigned off_hi, unsigned seed_lo, unsigned seed_hi) {
  return mix32(row * 0x9E3779B1u + (off_lo ^ seed_lo)) ^ (off_hi * 0xC2B2AE3Du + seed_hi);
}
__device__ __forceinline__ u32x4 dropout_bits_fast(unsigned row_key, unsigned col_group) {
  const unsigned x = row_key + col_group * 0x85EBCA77u;
  const unsigned t = x ^ (x >> 15), tb = t >> 11;
  u32x4 r;
  unsigned y;
#define SAM_FIN24(c1, d1) (y = __umul24(t, c1) + __umul24(tb, d1), y ^= y >> 13, y = __umul24(y, 0x52A6B5u), y ^ (y >> 16))
  r.x = SAM_FIN24(0x6B43A9u, 0x3C6EF3u);
  r.y = SAM_FIN24(0xD35A2Du, 0x7F4A7Du);
  r.z = SAM_FIN24(0x9E3B71u, 0x2545F5u);
  r.w = SAM_FIN24(0xB5297Bu, 0x5851F5u);
#undef SAM_FIN24
  return r;
}
// The half of dropout_bits_fast a lane needs when it owns 4 of the 8 columns of a group: words (x, y) for the lower four, (z, w) for the upper four.
struct DropHalfConsts { unsigned c1a, d1a, c1b, d1b; };
__device__ __forceinline__ DropHalfConsts dropout_half_consts(bool upper) {
  DropHalfConsts k;
  k.c1a = upper ? 0x9E3B71u : 0x6B43A9u; k.d1a = upper ? 0x2545F5u : 0x3C6EF3u;
  k.c1b = upper ? 0xB5297Bu : 0xD35A2Du; k.d1b = upper ? 0x5851F5u : 0x7F4A7Du;
  return k;
}
__device__ __forceinline__ void dropout_bits_half(unsigned row_key, unsigned col_group, const DropHalfConsts& k, unsigned& lo, unsigned& hi) {
  const unsigned x = row_key + col_group * 0x85EBCA77u;
  const unsigned t = x ^ (x >> 15), tb = t >> 11;
  unsigned y;
#define SAM_FIN24(c1, d1) (y = __umul24(t, c1) + __umul24(tb, d1), y ^= y >> 13, y = __umul24(y, 0x52A6B5u), y ^ (y >> 16))
  lo = SAM_FIN24(k.c1a, k.d1a);
  hi = SAM_FIN24(k.c1b, k.d1b);
#undef SAM_FIN24
}
__device__ __forceinline__ unsigned hidden_dropout_row_key(unsigned row, unsigned off_lo, unsigned off_hi, unsigned seed_lo, unsigned seed_hi) {
  return dropout_row_key(row, off_lo, off_hi, seed_lo ^ 0x5bd1e995u, seed_hi ^ 0x1b873593u);
}
// every wave-wide sum of the library goes through the VALU butterfly (wave_sum_v above).  (Round 4 first moved the LayerNorm pair only: with embed.hip's
// l2norm kernels on it test_incremental_beam_steps_decode_like_the_full_recompute failed.  Not the reduction -- wave_sum_v is bit-identical to the ds_bpermute
// butterfly there too (tools/debug/l2norm_ab.py: same hashes under a run-time switch) -- but the different code around it contracted the sums of squares
// differently, one-ulp changes of a few features, and on that data a near-tie at the edge of the beam one step before the end then fell the other way in the
// full-recompute mode: the test now allows one such hidden flip, as it did a visible one; tools/debug/beam_inc_dbg.py compares both modes with the fp32 oracle.)
__device__ __forceinline__ float wave_sum(float v) { return wave_sum_v(v); }
// Hidden-state dropout (GEMM epilogues of BertSelfOutput / BertOutput, regenerated by the LayerNorm backward; the embedding dropout of
// PrevPredEmbeddings; the input encoders): 8 x 16 random bits per (row, 8-column group).
__device__ __forceinline__ u32x4 hidden_dropout_bits(unsigned row, unsigned col8, unsigned off_lo, unsigned off_hi, unsigned seed_lo, unsigned seed_hi) {
  // (domain separation: the key is flipped by a constant so that a hidden-state site and an attention site never draw the same 128 bits even
  // when handed the same (seed, offset) -- without it the two masks agreed 4.6 % more often than independent draws)
  return dropout_bits_fast(dropout_row_key(row, off_lo, off_hi, seed_lo ^ 0x5bd1e995u, seed_hi ^ 0x1b873593u), col8);
}
// dropout threshold on 16-bit lanes of the random words: element kept iff rnd16 >= thr16
__host__ __device__ __forceinline__ unsigned dropout_thr16(float p) {
  float t = p * 65536.0f + 0.5f;
  return t <= 0.f ? 0u : (t >= 65535.f ? 65535u : (unsigned)t);
}

// erf-GELU (sam/sa_m4c.py:985-991) and its derivative.  libm's erff costs ~40 VALU with branches and made the GELU epilogues
// 30% of their GEMMs; Abramowitz-Stegun 7.1.26 (|abs error| <= 1.5e-7, far below bf16 resolution) needs one rcp, one exp2 and
// a 5-term Horner polynomial, and the SAME exponential exp(-x^2/2) serves the Gaussian term of the derivative.
__device__ __forceinline__ float gelu_phi_and_cdf(float x, float& cdf) {   // returns exp(-x^2/2); cdf = 0.5*(1+erf(x/sqrt2))
  const float z = fabsf(x) * 0.70710678118654752f;
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
  const float e = __builtin_amdgcn_exp2f(-0.72134752044448170f * x * x);        // exp(-x^2/2) = 2^(-x^2 * log2(e)/2)
  float poly = fmaf(1.061405429f, t, -1.453152027f);
  poly = fmaf(poly, t, 1.421413741f);
  poly = fmaf(poly, t, -0.284496736f);
  poly = fmaf(poly, t, 0.254829592f);
  const float erf_abs = 1.0f - poly * t * e;                                     // erf(|x|/sqrt2)
  cdf = 0.5f + 0.5f * copysignf(erf_abs, x);
  return e;
}
__device__ __forceinline__ float gelu_erf(float x) {
  float cdf;
  gelu_phi_and_cdf(x, cdf);
  return x * cdf;
}
// activation and derivative from one exponential (training forward: the derivative is what the backward needs, not the pre-activation)
__device__ __forceinline__ float gelu_erf_and_grad(float x, float& grad) {
  float cdf;
  const float e = gelu_phi_and_cdf(x, cdf);
  grad = cdf + x * 0.39894228040143268f * e;
  return x * cdf;
}
__device__ __forceinline__ float gelu_erf_grad(float x) {
  float cdf;
  const float e = gelu_phi_and_cdf(x, cdf);
  return cdf + x * 0.39894228040143268f * e;
}
