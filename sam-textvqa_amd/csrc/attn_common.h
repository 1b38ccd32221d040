// Pieces shared by the fused attention kernels (attention.hip: forward + the two-kernel backward kept for long sequences;
// attention_bwd_fused.hip: the one-pass backward).  gfx950 only.
//
// Second-stage operands.  The probabilities P and the score gradients dS are fp32 values that must enter a 16-bit MFMA.  bf16 (8-bit
// mantissa) costs 2.5e-3 * max in the worst output element, which is why rounds 1-3 split them into hi + lo bf16 pairs (two MFMAs and
// ~3 VALU per value).  fp16 has 11 bits: ONE rounding is 8x finer and meets the 1e-3 bound (tools/debug/attn_numerics_sim.py), provided
// the other operand is fp16 too and nothing leaves fp16's range.  So every tile that feeds a second-stage MFMA is converted once per block
// from bf16 to fp16 with a per-(batch, head) power-of-two scale taken from the tile's largest magnitude: the conversion is EXACT (8-bit
// mantissas fit; values more than 2^29 below the tile maximum flush to zero), three packed integer ops per pair, and the accumulators
// are scaled back by the inverse power of two.
#pragma once
#include "common.h"

typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2;
typedef __attribute__((ext_vector_type(2))) unsigned short u16x2;
typedef __attribute__((ext_vector_type(2))) short i16x2;

namespace attn {

constexpr int HD = 64;             // head dim
constexpr int ROW_BYTES = HD * 2;  // one LDS tile row
constexpr float LOG2E = 1.4426950408889634f;
constexpr int P_SHIFT = 14;        // probabilities are carried as P * 2^14: fp16 keeps full precision down to P = 2^-28

// byte offset of 16-byte chunk `ch` (0..7) of tile row `row`; the XOR keeps both ds_read_b128 row reads (16 rows x same chunk) and
// ds_read_b64_tr_b16 column reads (8 rows x same 32-B block) conflict-free
__device__ __forceinline__ int tile_off(int row, int ch) { return row * ROW_BYTES + ((ch ^ (((row >> 1) & 3) << 1)) << 4); }

__device__ __forceinline__ bf16x8 lds_row_frag(const unsigned char* tile, int row, int ch) {
  return *reinterpret_cast<const bf16x8*>(tile + tile_off(row, ch));
}
// transposed fragment: lane (i,g) gets tile[32*s + 16*(e>>2) + 4*g + (e&3)][16*dt + i], e = 0..7
__device__ __forceinline__ bf16x8 lds_col_frag(const unsigned char* tile, int s, int dt, int i, int g) {
  const int row = 32 * s + 4 * g + (i >> 2);
  const unsigned char* p = tile + tile_off(row, 2 * dt + ((i & 3) >> 1)) + (i & 1) * 8;
  return cat4(lds_read_tr16(p), lds_read_tr16(p + 16 * ROW_BYTES));  // row+16 has the same swizzle
}
__device__ __forceinline__ f16x8 as_f16(bf16x8 v) { return __builtin_bit_cast(f16x8, v); }

__device__ __forceinline__ unsigned pack_f16x2(float lo, float hi) {   // RNE, v_cvt_pk_f16_f32
  f32x2 f = {lo, hi};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(f, f16x2));
}

// ---- bf16 -> fp16 with a power-of-two scale --------------------------------------------------------------------------------------
// |x| as 15 bits = e8 m7.  With c = max(E_max - 29, 0) (E_max = biased exponent of the tile's largest magnitude) the fp16 pattern is
// sign | (e8 - c) << 10 | m7 << 3: the largest value lands in [2^14, 2^15), anything with e8 <= c flushes to (almost) zero.
// fp16 value * 2^(c - 112) == bf16 value.
__device__ __forceinline__ unsigned absmax_acc(unsigned acc, unsigned x) {          // running packed max of |bf16| pairs
  return __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(u16x2, acc), __builtin_bit_cast(u16x2, x & 0x7fff7fffu)));
}
__device__ __forceinline__ unsigned pk_max_u16(unsigned a, unsigned b) {
  return __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(u16x2, a), __builtin_bit_cast(u16x2, b)));
}
__device__ __forceinline__ unsigned absmax_acc4(unsigned acc, uint4 v) {
  return absmax_acc(absmax_acc(absmax_acc(absmax_acc(acc, v.x), v.y), v.z), v.w);
}
__device__ __forceinline__ unsigned absmax_fold(unsigned packed) {                   // two packed maxima -> one 15-bit pattern
  const unsigned lo = packed & 0xffffu, hi = packed >> 16;
  return lo > hi ? lo : hi;
}
__device__ __forceinline__ unsigned wave_max_u32(unsigned v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const unsigned w = (unsigned)__shfl_xor((int)v, o);
    v = v > w ? v : w;
  }
  return v;
}
__device__ __forceinline__ int scale_c_of(unsigned absmax15) {                      // c >= 0
  const int e = (int)(absmax15 >> 7);
  return e > 29 ? e - 29 : 0;
}
__device__ __forceinline__ unsigned csub_of(int c) { const unsigned s = (unsigned)c << 7; return s | (s << 16); }
__device__ __forceinline__ float pow2f(int e) { return __uint_as_float((unsigned)(e + 127) << 23); }    // -126 <= e <= 127
__device__ __forceinline__ unsigned bf2h_pk(unsigned x, unsigned csub) {
  u16x2 r = __builtin_elementwise_sub_sat(__builtin_bit_cast(u16x2, x & 0x7fff7fffu), __builtin_bit_cast(u16x2, csub));
  r = r << (unsigned short)3;
  return (x & 0x80008000u) | __builtin_bit_cast(unsigned, r);
}
__device__ __forceinline__ uint4 bf2h_pk4(uint4 v, unsigned csub) {
  return make_uint4(bf2h_pk(v.x, csub), bf2h_pk(v.y, csub), bf2h_pk(v.z, csub), bf2h_pk(v.w, csub));
}

// ---- attention-probability dropout: 8 x 16 random bits per (row, 32-key slab, lane group): dropout_row_key / dropout_bits_fast of common.h
__device__ __forceinline__ unsigned attn_row_key(unsigned row, unsigned off_lo, unsigned off_hi, unsigned seed_lo, unsigned seed_hi) {
  return dropout_row_key(row, off_lo, off_hi, seed_lo, seed_hi);
}
__device__ __forceinline__ u32x4 attn_dropout_bits(unsigned row_key, unsigned col_group) { return dropout_bits_fast(row_key, col_group); }
// dropped-lane mask of a random word: each 16-bit half becomes 0xFFFF iff its value < thr16 (thr2 = (thr16 ^ 0x8000) in both halves)
__device__ __forceinline__ unsigned drop_mask16x2(unsigned rnd, unsigned thr2) {
  i16x2 d = __builtin_elementwise_sub_sat(__builtin_bit_cast(i16x2, rnd ^ 0x80008000u), __builtin_bit_cast(i16x2, thr2));
  d = d >> (short)15;
  return __builtin_bit_cast(unsigned, d);
}

struct AttnArgs {
  const bf16_t* qkv;    // [B*N, 3*H*64]  q | k | v
  const bf16_t* dout;   // [B*N, H*64]    (bwd only)
  bf16_t* out_w;        // fwd output
  bf16_t* out_lo_w;     // fwd: bf16 residual of the output (out_exact - bf16(out)), for the one-pass backward's delta; may be null
  const bf16_t* out;    // fused bwd: forward output and its residual
  const bf16_t* out_lo;
  bf16_t* dqkv;         // [B*N, 3*H*64]  (bwd output)
  const uint32_t* allow;  // [B, Hm, N, NW]
  int64_t allow_sb, allow_sh;
  uint32_t* keep_w;       // [B, H, N, NW] fwd writes (dropout only)
  const uint32_t* keep;   // bwd reads (nullptr = everything kept)
  float* lse2_w;          // [B, H, N] fwd writes: log2-domain logsumexp of scale*s (+inf for dead rows)
  const float* lse2;
  float* delta;           // [B, H, N] two-kernel bwd workspace: sum_k P*dP per query row
  int B, N, H, NW, nkt, q_begin;   // q_begin: first query row to compute (forward only; rounded down to a 16-row tile)
  // decoding (attn_fwd_kernel<NKT, true>): the q|k|v rows of the n_dec = N - n_enc decoder tokens live in their own compact buffer
  // qkv_dec [B * n_dec, 3*H*64] (written by this step's QKV projection, no copy into the cache), the encoder rows in `qkv` as ever;
  // only decoder rows are written, compactly, to out_dec [B * n_dec, H*64]
  const bf16_t* qkv_dec;
  bf16_t* out_dec;
  int n_enc;
  int kv_group;      // decoding: consecutive groups of kv_group decoder samples (the beams of one sample) share ONE sample's encoder rows and allow words (>= 1)
  float scale, scale_log2, p_drop, inv_keep;
  float ds_c1;            // fused bwd: inv_keep * scale * 2^(-36 - ds_sh), see attention_bwd_fused.hip
  int ds_sh;
  unsigned thr16, seed_lo, seed_hi, off_lo, off_hi;
  const unsigned long long* rng_state;
};

int pick_nkt(int N);
int fill_common(AttnArgs& a, int B, int N, int H, int head_dim, float scale, float p_drop);

}  // namespace attn
