// Fused relation-masked multi-head attention for SA-M4C (gfx950).
//
// Replaces the eager-op cluster of /root/reference/sam/sa_m4c.py:563-598 (scores, min-combined
// additive masks, softmax, fully-masked-row zeroing, dropout, PV, head merge) and its autograd,
// for the spatial ('s') layers, the plain ('n') layers and TextBert alike: the per-(batch, head,
// query) key allow-bitmask built by masks.hip carries ALL mask semantics, so these kernels only
// test bits.  One workgroup = one (batch, head); head_dim = 64; whole K/V (or Q/dO) of the head
// lives in LDS; softmax is single-pass in registers (N <= 384 keys).
//
// Orientation ("query per lane"): scores are produced as S^T tiles, D[key][query] with
// query = lane&15 and keys 16t + 4*(lane>>4) + r, so a softmax row lives in 4 lanes x NKT*4
// registers, P feeds the PV MFMA as its B operand with NO cross-lane movement, and V (stored
// row-major [key][d] in LDS) is consumed through ds_read_b64_tr_b16.  The contraction index of
// every second-stage MFMA is enumerated as k(g,e) = 32*s + 16*(e>>2) + 4*g + (e&3).
#include "attn_common.h"

using namespace attn;

namespace {

// stage rows [0,n_valid) of a [*, ld] bf16 matrix slice (64 columns) into an LDS tile of npad rows
__device__ __forceinline__ void stage_tile(unsigned char* lds, const bf16_t* base, int64_t ld, int n_valid, int npad, int tid) {
  for (int c = tid; c < npad * 8; c += blockDim.x) {
    const int row = c >> 3, ch = c & 7;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (row < n_valid) v = *reinterpret_cast<const uint4*>(base + (int64_t)row * ld + ch * 8);
    *reinterpret_cast<uint4*>(lds + tile_off(row, ch)) = v;
  }
}

// Second-stage operands of the two-kernel backward (long sequences): v = hi + lo bf16 pairs (two MFMAs on the same accumulator); the
// forward and the one-pass backward use fp16 instead (attn_common.h).
__device__ __forceinline__ void split_pack8(const float* v, bf16x8& hi, bf16x8& lo) {
  typedef __attribute__((ext_vector_type(4))) unsigned u4;
  u4 h, l;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    h[j] = pack_bf16x2(v[2 * j], v[2 * j + 1]);
    l[j] = pack_bf16x2(v[2 * j] - bf_lo(h[j]), v[2 * j + 1] - bf_hi(h[j]));
  }
  hi = __builtin_bit_cast(bf16x8, h);
  lo = __builtin_bit_cast(bf16x8, l);
}

// --------------------------------------------------------------------------------------------
// forward
// --------------------------------------------------------------------------------------------
// One workgroup = one (batch, head).  K (bf16, as stored) and V (converted to block-scaled fp16) live in LDS; each wave owns 16-query
// strips.  Per score the VALU does: mask as the MFMA's C operand (-inf where the allow bit is clear: 2 ops), running max (1/2), one
// fma + exp2, the row sum, half a v_cvt_pk_f16_f32 -- and with dropout a packed 16-bit threshold test on the fp16 pairs, the keep-bit
// word and the counter hash.  Rounds 1-3 spent 23 VALU per score here (hi/lo bf16 split, select-based masks, 32-bit multiplies in the
// hash); this is ~13 with dropout, ~6 without.
// one 16-query strip of one (batch, head): scores -> softmax -> dropout -> PV -> stores.  Ks (bf16) / Vs (block-scaled fp16) are the head's LDS tiles.
template <int NKT, bool DEC, bool DROP>
__device__ __forceinline__ void attn_fwd_strip(const AttnArgs& a, const unsigned char* Ks, const unsigned char* Vs, const bf16x8 (&qf)[2], const unsigned (&naw)[NKT / 2],
                                               int bh, int b, int h, int q, int qc, int N, int Dm, int i, int g, float o_unscale, unsigned thr2, unsigned nibmask,
                                               unsigned seed_lo, unsigned seed_hi, unsigned off_lo, unsigned off_hi) {
  // scores with the mask as the accumulator's initial value: S = -inf wherever the allow bit is clear
  f32x4 s[NKT];
#pragma unroll
  for (int t = 0; t < NKT; ++t) {
    f32x4 acc;
#pragma unroll
    for (int r = 0; r < 4; ++r)
      acc[r] = __int_as_float(__builtin_amdgcn_sbfe((int)naw[t >> 1], (unsigned)((t & 1) * 16 + r), 1u) & (int)0xff800000u);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(lds_row_frag(Ks, 16 * t + i, 4 * ks + g), qf[ks], acc, 0, 0, 0);
    s[t] = acc;
  }
  float mx = -INFINITY;
#pragma unroll
  for (int t = 0; t < NKT; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) mx = fmaxf(mx, s[t][r]);
  mx = xgroup_max(mx);
  const bool alive = mx > -INFINITY;  // reference: fully masked rows give exactly 0 (sa_m4c.py:574-584)
  const float bias = (float)P_SHIFT - (alive ? mx : 0.f) * a.scale_log2;
  float sum = 0.f;
  unsigned pk[2 * NKT];               // P * 2^14 as fp16 pairs: (keys 4g, 4g+1) and (4g+2, 4g+3) of every tile
#pragma unroll
  for (int t = 0; t < NKT; ++t) {
    float p[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      p[r] = __builtin_amdgcn_exp2f(fmaf(s[t][r], a.scale_log2, bias));     // 2^14 * exp(scale * s - max); -inf -> 0
      sum += p[r];
    }
    pk[2 * t] = pack_f16x2(p[0], p[1]);
    pk[2 * t + 1] = pack_f16x2(p[2], p[3]);
  }
  sum = xgroup_sum(sum);
  const unsigned rk = DROP ? attn_row_key((unsigned)(bh * N + qc), off_lo, off_hi, seed_lo, seed_hi) : 0u;
  // PV, one 32-key slab (two score tiles) at a time: dropout (sa_m4c.py:588: after the row zeroing, before PV) as a packed mask on the fp16
  // pairs, one MFMA per 16-column block of V
  const float inv = alive ? o_unscale * __builtin_amdgcn_rcpf(sum) : 0.f;
  f32x4 o[4];
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) o[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int w = 0; w < NKT / 2; ++w) {
    typedef __attribute__((ext_vector_type(4))) unsigned u4;
    u4 pw = {pk[4 * w], pk[4 * w + 1], pk[4 * w + 2], pk[4 * w + 3]};
    if (DROP) {
      const u32x4 rn = attn_dropout_bits(rk, (unsigned)(w * 4 + g));
      const unsigned dm0 = drop_mask16x2(rn.x, thr2), dm1 = drop_mask16x2(rn.y, thr2), dm2 = drop_mask16x2(rn.z, thr2), dm3 = drop_mask16x2(rn.w, thr2);
      pw[0] &= ~dm0; pw[1] &= ~dm1; pw[2] &= ~dm2; pw[3] &= ~dm3;
      // keep word of this query row: bit (e>>2)*16 + 4g + (e&3) for the lane's e-th key of the slab (e = 2j / 2j+1 = low / high half of pair j)
      const unsigned x01 = (dm0 & 0x00020001u) | (dm1 & 0x00080004u), x23 = (dm2 & 0x00020001u) | (dm3 & 0x00080004u);
      const unsigned dropped = (((x01 | (x01 >> 16)) & 0xFu) | (((x23 | (x23 >> 16)) & 0xFu) << 16)) << (4 * g);
      const unsigned bits = xgroup_or(nibmask & ~dropped);
      if (q < N && g == (w & 3)) a.keep_w[((int64_t)bh * N + q) * a.NW + w] = bits;
    }
    const f16x8 pa = __builtin_bit_cast(f16x8, pw);
    bf16x8 vt[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) vt[dt] = lds_col_frag(Vs, w, dt, i, g);
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(as_f16(vt[dt]), pa, o[dt], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);   // keep the slabs sequential: hoisting every V fragment and hash costs a wave of occupancy
  }
  // epilogue, 16 bytes per store: v_permlane16_swap pairs the 4-column fragments of two neighbouring lane groups, lane (i, g) then owns the
  // eight consecutive columns 32 jp + 16 (g & 1) + 8 (g >> 1) .. +7 of query row i
  const bool wr = DEC ? (q < N && q >= a.n_enc) : (q < N);
  const int64_t orow = DEC ? ((int64_t)b * (N - a.n_enc) + q - a.n_enc) * Dm : ((int64_t)b * N + q) * Dm;
  const int ocol = h * HD + 16 * (g & 1) + 8 * (g >> 1);
#pragma unroll
  for (int jp = 0; jp < 2; ++jp) {
    float v[8];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const auto sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(o[2 * jp][r]), __float_as_uint(o[2 * jp + 1][r]), false, false);
      v[r] = __uint_as_float(sw[0]) * inv;
      v[4 + r] = __uint_as_float(sw[1]) * inv;
    }
    const uint4 hi = make_uint4(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7]));
    if (wr) {
      *reinterpret_cast<uint4*>((DEC ? a.out_dec : a.out_w) + orow + ocol + 32 * jp) = hi;
      if (!DEC && a.out_lo_w)
        *reinterpret_cast<uint4*>(a.out_lo_w + orow + ocol + 32 * jp) =
            make_uint4(pack_bf16x2(v[0] - bf_lo(hi.x), v[1] - bf_hi(hi.x)), pack_bf16x2(v[2] - bf_lo(hi.y), v[3] - bf_hi(hi.y)),
                       pack_bf16x2(v[4] - bf_lo(hi.z), v[5] - bf_hi(hi.z)), pack_bf16x2(v[6] - bf_lo(hi.w), v[7] - bf_hi(hi.w)));
    }
  }
  if (!DEC && wr && g == 0) a.lse2_w[(int64_t)bh * N + q] = alive ? mx * a.scale_log2 + __builtin_amdgcn_logf(sum) - (float)P_SHIFT : INFINITY;
}

constexpr int fwd_waves_per_eu(int nkt, int nt) {     // blocks per CU by LDS (2 tiles of nkt * 2 KB) x waves per block / 4 SIMDs, rounded up
  return nkt >= 16 ? 2 : nkt == 12 ? (nt == 512 ? 4 : nt == 384 ? 5 : 3) : nkt == 2 ? 2 : 4;
}
template <int NKT, bool DEC, bool DROP, int NT>
__global__ __launch_bounds__(NT, fwd_waves_per_eu(NKT, NT)) void attn_fwd_kernel(AttnArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int NPAD = NKT * 16, NCH = NPAD * 8, PER = NCH / NT, NWV = NT / 64;
  static_assert(NCH % NT == 0, "tile chunks must divide evenly over the block");
  unsigned char* Ks = smem;
  unsigned char* Vs = smem + NPAD * ROW_BYTES;
  unsigned* red = reinterpret_cast<unsigned*>(Vs + NPAD * ROW_BYTES);      // [NWV]
  const int tid = threadIdx.x, bh = blockIdx.x, b = bh / a.H, h = bh % a.H;
  const int N = a.N, Dm = a.H * HD;
  const int64_t ld = 3 * (int64_t)Dm;
  const int be = DEC ? b / a.kv_group : b;            // the sample whose encoder rows / allow words this block reads (beams share them)
  const bf16_t* qbase = a.qkv + (int64_t)be * N * ld + h * HD;
  const bf16_t* dbase = DEC ? a.qkv_dec + (int64_t)b * (N - a.n_enc) * ld + h * HD : nullptr;
  const int lane = tid & 63, wave = tid >> 6, i = lane & 15, g = lane >> 4;

  // the first strip's query rows and allow words are requested before the tiles are staged, every later strip's while the previous one computes:
  // with one or two strips per wave a load issued at the top of its own strip is a full memory latency nobody else covers
  const int mt0 = (a.q_begin >> 4) + wave;
  bf16x8 qf_n[2];
  unsigned naw_n[NKT / 2];        // inverted allow words, pre-shifted so that this lane's four keys of a tile sit at bits 0..3 / 16..19
  auto load_strip = [&](int mt) {
    int q = mt * 16 + i;
    q = q < N ? q : N - 1;
    const bf16_t* qrow = (DEC && q >= a.n_enc) ? dbase + (int64_t)(q - a.n_enc) * ld : qbase + (int64_t)q * ld;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) qf_n[ks] = *reinterpret_cast<const bf16x8*>(qrow + 32 * ks + 8 * g);
    const uint32_t* ap = a.allow + be * a.allow_sb + h * a.allow_sh + (int64_t)q * a.NW;
#pragma unroll
    for (int w = 0; w < NKT / 2; ++w) naw_n[w] = ap[w];
  };
  load_strip(mt0);

  // ---- stage K as it is and V as block-scaled fp16; every load is unconditional at a clamped row (no branch in front of a load)
  uint4 vreg[PER];
  unsigned vmax = 0;
#pragma unroll
  for (int j = 0; j < PER; ++j) {
    const int c = tid + j * NT, row = c >> 3, ch = c & 7, rc = row < N ? row : N - 1;
    const bf16_t* src = (DEC && rc >= a.n_enc) ? dbase + (int64_t)(rc - a.n_enc) * ld : qbase + (int64_t)rc * ld;
    uint4 kv = *reinterpret_cast<const uint4*>(src + Dm + ch * 8);
    uint4 vv = *reinterpret_cast<const uint4*>(src + 2 * Dm + ch * 8);
    if (row >= N) { kv = make_uint4(0, 0, 0, 0); vv = make_uint4(0, 0, 0, 0); }
    *reinterpret_cast<uint4*>(Ks + tile_off(row, ch)) = kv;
    vreg[j] = vv;
    vmax = absmax_acc4(vmax, vv);
  }
  vmax = wave_max_u32(absmax_fold(vmax));
  if (lane == 0) red[wave] = vmax;
  __syncthreads();
  unsigned bmax = 0;
#pragma unroll
  for (int w = 0; w < NWV; ++w) bmax = red[w] > bmax ? red[w] : bmax;
  const int cv = scale_c_of(bmax);
  const unsigned csub = csub_of(cv);
#pragma unroll
  for (int j = 0; j < PER; ++j) {
    const int c = tid + j * NT;
    *reinterpret_cast<uint4*>(Vs + tile_off(c >> 3, c & 7)) = bf2h_pk4(vreg[j], csub);
  }
  __syncthreads();

  unsigned seed_lo = a.seed_lo, seed_hi = a.seed_hi, off_lo = a.off_lo, off_hi = a.off_hi;
  if (DROP) rng_resolve(a.rng_state, seed_lo, seed_hi, off_lo, off_hi);
  const unsigned thr2 = ((a.thr16 ^ 0x8000u) & 0xffffu) * 0x00010001u;
  const unsigned nibmask = 0x000F000Fu << (4 * g);
  const float o_unscale = ldexpf(a.inv_keep, cv - 112);          // V16 * 2^(cv-112) = V; the 2^P_SHIFT of P cancels against the row sum
  for (int mt = mt0; mt * 16 < N; mt += NWV) {
    const int q = mt * 16 + i, qc = q < N ? q : N - 1;
    bf16x8 qf[2] = {qf_n[0], qf_n[1]};
    unsigned naw[NKT / 2];
#pragma unroll
    for (int w = 0; w < NKT / 2; ++w) naw[w] = ~naw_n[w] >> (4 * g);
    load_strip(mt + NWV);          // (clamped to the last row when there is no next strip)

    attn_fwd_strip<NKT, DEC, DROP>(a, Ks, Vs, qf, naw, bh, b, h, q, qc, N, Dm, i, g, o_unscale, thr2, nibmask, seed_lo, seed_hi, off_lo, off_hi);
  }
}

// --------------------------------------------------------------------------------------------
// backward, pass 1: dQ (query per lane; K and V in LDS); also emits delta = rowsum(dO * O)
// --------------------------------------------------------------------------------------------
template <int NKT>
__global__ __launch_bounds__(NKT <= 12 ? 256 : 512, NKT <= 12 ? 3 : 2) void attn_bwd_dq_kernel(AttnArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int NPAD = NKT * 16;
  unsigned char* Ks = smem;
  unsigned char* Vs = smem + NPAD * ROW_BYTES;
  const int tid = threadIdx.x, bh = blockIdx.x, b = bh / a.H, h = bh % a.H;
  const int N = a.N, Dm = a.H * HD;
  const int64_t ld = 3 * (int64_t)Dm;
  const bf16_t* qbase = a.qkv + (int64_t)b * N * ld + h * HD;
  stage_tile(Ks, qbase + Dm, ld, N, NPAD, tid);
  stage_tile(Vs, qbase + 2 * Dm, ld, N, NPAD, tid);
  __syncthreads();

  const int lane = tid & 63, wave = tid >> 6, i = lane & 15, g = lane >> 4;
  const int nwaves = blockDim.x >> 6;
  for (int mt = wave; mt * 16 < N; mt += nwaves) {
    const int q = mt * 16 + i, qc = q < N ? q : N - 1;
    const int64_t orow = ((int64_t)b * N + qc) * Dm + h * HD;
    bf16x8 qf[2], dof[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      qf[ks] = *reinterpret_cast<const bf16x8*>(qbase + (int64_t)qc * ld + 32 * ks + 8 * g);
      dof[ks] = *reinterpret_cast<const bf16x8*>(a.dout + orow + 32 * ks + 8 * g);
    }
    const float lse = a.lse2[(int64_t)bh * N + qc];
    const uint32_t* ap = a.allow + b * a.allow_sb + h * a.allow_sh + (int64_t)qc * a.NW;
    const uint32_t* kp = a.keep ? a.keep + ((int64_t)bh * N + qc) * a.NW : nullptr;
    // One pass over the keys: P and the dropout-scaled dP of the whole row stay in registers (N <= 384 keys = 96 + 96 values per lane at most),
    // delta = sum_k P_k * dP_k in fp32 from them (== rowsum(dO*O) of the exact forward; taking it from the bf16-stored O instead costs ~3e-3
    // relative error in dQ/dK).  (The first version recomputed S and dP with the MFMAs -- and the exponentials -- in a second pass.)
    constexpr bool KEEP_DP = true;
    f32x4 pr[NKT], dpr[KEEP_DP ? NKT : 1];
    float delta = 0.f;
#pragma unroll
    for (int w = 0; w < NKT / 2; ++w) {
      const unsigned aw = ap[w], kw = kp ? kp[w] : 0xffffffffu;
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        const int t = 2 * w + half;
        f32x4 acc_s = {0.f, 0.f, 0.f, 0.f}, acc_dp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          acc_s = __builtin_amdgcn_mfma_f32_16x16x32_bf16(lds_row_frag(Ks, 16 * t + i, 4 * ks + g), qf[ks], acc_s, 0, 0, 0);
          acc_dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(lds_row_frag(Vs, 16 * t + i, 4 * ks + g), dof[ks], acc_dp, 0, 0, 0);
        }
        const unsigned na = (aw >> (half * 16 + 4 * g)) & 0xFu, nk = (kw >> (half * 16 + 4 * g)) & 0xFu;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float p = ((na >> r) & 1u) ? __builtin_amdgcn_exp2f(acc_s[r] * a.scale_log2 - lse) : 0.f;
          const float dpe = ((nk >> r) & 1u) ? acc_dp[r] * a.inv_keep : 0.f;
          pr[t][r] = p;
          if (KEEP_DP) dpr[KEEP_DP ? t : 0][r] = dpe;
          delta += p * dpe;
        }
      }
      __builtin_amdgcn_sched_barrier(0);      // one 32-key slab at a time: hoisting every K / V fragment read costs the registers of a wave of occupancy
    }
    delta = xgroup_sum(delta);
    f32x4 dq[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) dq[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int w = 0; w < NKT / 2; ++w) {
      float dsv[8];
      if (KEEP_DP) {
#pragma unroll
        for (int e = 0; e < 8; ++e) dsv[e] = pr[2 * w + (e >> 2)][e & 3] * (dpr[KEEP_DP ? 2 * w + (e >> 2) : 0][e & 3] - delta) * a.scale;
      } else {
        const unsigned kw = kp ? kp[w] : 0xffffffffu;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
          const int t = 2 * w + half;
          f32x4 acc_dp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int ks = 0; ks < 2; ++ks) acc_dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(lds_row_frag(Vs, 16 * t + i, 4 * ks + g), dof[ks], acc_dp, 0, 0, 0);
          const unsigned nk = (kw >> (half * 16 + 4 * g)) & 0xFu;
#pragma unroll
          for (int r = 0; r < 4; ++r) dsv[half * 4 + r] = pr[t][r] * ((((nk >> r) & 1u) ? acc_dp[r] * a.inv_keep : 0.f) - delta) * a.scale;
        }
      }
      bf16x8 dsa, dsl;
      split_pack8(dsv, dsa, dsl);
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        const bf16x8 kt_ = lds_col_frag(Ks, w, dt, i, g);
        dq[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kt_, dsa, dq[dt], 0, 0, 0);
        dq[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kt_, dsl, dq[dt], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if (q < N) {
      bf16_t* dst = a.dqkv + ((int64_t)b * N + q) * ld + h * HD + 4 * g;
#pragma unroll
      for (int dt = 0; dt < 4; ++dt)
        *reinterpret_cast<uint2*>(dst + 16 * dt) = make_uint2(pack_bf16x2(dq[dt][0], dq[dt][1]), pack_bf16x2(dq[dt][2], dq[dt][3]));
      if (g == 0) a.delta[(int64_t)bh * N + q] = delta;
    }
  }
}

// two-pass form (S and dP recomputed in the second pass): 24 key tiles, where a whole row of P does not fit in registers next to the rest
__global__ __launch_bounds__(512, 2) void attn_bwd_dq2_kernel(AttnArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int NPAD = a.nkt * 16;
  unsigned char* Ks = smem;
  unsigned char* Vs = smem + NPAD * ROW_BYTES;
  const int tid = threadIdx.x, bh = blockIdx.x, b = bh / a.H, h = bh % a.H;
  const int N = a.N, Dm = a.H * HD;
  const int64_t ld = 3 * (int64_t)Dm;
  const bf16_t* qbase = a.qkv + (int64_t)b * N * ld + h * HD;
  stage_tile(Ks, qbase + Dm, ld, N, NPAD, tid);
  stage_tile(Vs, qbase + 2 * Dm, ld, N, NPAD, tid);
  __syncthreads();

  const int lane = tid & 63, wave = tid >> 6, i = lane & 15, g = lane >> 4;
  const int nwaves = blockDim.x >> 6;
  for (int mt = wave; mt * 16 < N; mt += nwaves) {
    const int q = mt * 16 + i, qc = q < N ? q : N - 1;
    const int64_t orow = ((int64_t)b * N + qc) * Dm + h * HD;
    bf16x8 qf[2], dof[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      qf[ks] = *reinterpret_cast<const bf16x8*>(qbase + (int64_t)qc * ld + 32 * ks + 8 * g);
      dof[ks] = *reinterpret_cast<const bf16x8*>(a.dout + orow + 32 * ks + 8 * g);
    }
    const float lse = a.lse2[(int64_t)bh * N + qc];
    const uint32_t* ap = a.allow + b * a.allow_sb + h * a.allow_sh + (int64_t)qc * a.NW;
    const uint32_t* kp = a.keep ? a.keep + ((int64_t)bh * N + qc) * a.NW : nullptr;
    // pass 1: delta = sum_k P_k * dP_k in fp32 from the recomputed probabilities (== rowsum(dO*O) of the
    // exact forward; taking it from the bf16-stored O instead costs ~3e-3 relative error in dQ/dK)
    float delta = 0.f;
    for (int w = 0; w < a.nkt / 2; ++w) {
      const unsigned aw = ap[w], kw = kp ? kp[w] : 0xffffffffu;
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        const int t = 2 * w + half;
        f32x4 acc_s = {0.f, 0.f, 0.f, 0.f}, acc_dp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          acc_s = __builtin_amdgcn_mfma_f32_16x16x32_bf16(lds_row_frag(Ks, 16 * t + i, 4 * ks + g), qf[ks], acc_s, 0, 0, 0);
          acc_dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(lds_row_frag(Vs, 16 * t + i, 4 * ks + g), dof[ks], acc_dp, 0, 0, 0);
        }
        const unsigned nb = ((aw & kw) >> (half * 16 + 4 * g)) & 0xFu;
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if ((nb >> r) & 1u) delta += __builtin_amdgcn_exp2f(acc_s[r] * a.scale_log2 - lse) * acc_dp[r];
      }
    }
    delta = xgroup_sum(delta) * a.inv_keep;
    f32x4 dq[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) dq[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int w = 0; w < a.nkt / 2; ++w) {
      const unsigned aw = ap[w], kw = kp ? kp[w] : 0xffffffffu;
      float dsv[8];
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        const int t = 2 * w + half;
        f32x4 acc_s = {0.f, 0.f, 0.f, 0.f}, acc_dp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          acc_s = __builtin_amdgcn_mfma_f32_16x16x32_bf16(lds_row_frag(Ks, 16 * t + i, 4 * ks + g), qf[ks], acc_s, 0, 0, 0);
          acc_dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(lds_row_frag(Vs, 16 * t + i, 4 * ks + g), dof[ks], acc_dp, 0, 0, 0);
        }
        const unsigned na = (aw >> (half * 16 + 4 * g)) & 0xFu, nk = (kw >> (half * 16 + 4 * g)) & 0xFu;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float p = ((na >> r) & 1u) ? __builtin_amdgcn_exp2f(acc_s[r] * a.scale_log2 - lse) : 0.f;
          const float dpe = ((nk >> r) & 1u) ? acc_dp[r] * a.inv_keep : 0.f;
          dsv[half * 4 + r] = p * (dpe - delta) * a.scale;
        }
      }
      bf16x8 dsa, dsl;
      split_pack8(dsv, dsa, dsl);
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        const bf16x8 kt_ = lds_col_frag(Ks, w, dt, i, g);
        dq[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kt_, dsa, dq[dt], 0, 0, 0);
        dq[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kt_, dsl, dq[dt], 0, 0, 0);
      }
    }
    if (q < N) {
      bf16_t* dst = a.dqkv + ((int64_t)b * N + q) * ld + h * HD + 4 * g;
#pragma unroll
      for (int dt = 0; dt < 4; ++dt)
        *reinterpret_cast<uint2*>(dst + 16 * dt) = make_uint2(pack_bf16x2(dq[dt][0], dq[dt][1]), pack_bf16x2(dq[dt][2], dq[dt][3]));
      if (g == 0) a.delta[(int64_t)bh * N + q] = delta;
    }
  }
}

// --------------------------------------------------------------------------------------------
// backward, pass 2: dK, dV (key per lane; Q, dO, lse, delta and the transposed bit rows in LDS)
// --------------------------------------------------------------------------------------------
template <int NT>      // 256 threads (<= 12 key tiles) or 512 (16 / 24): the bound is the register budget the compiler works with
__global__ __launch_bounds__(NT) void attn_bwd_dkdv_kernel(AttnArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int NPAD = a.nkt * 16, NW = a.NW;
  unsigned char* Qs = smem;
  unsigned char* dOs = Qs + NPAD * ROW_BYTES;
  float* lse_s = reinterpret_cast<float*>(dOs + NPAD * ROW_BYTES);
  float* del_s = lse_s + NPAD;
  uint32_t* allowT = reinterpret_cast<uint32_t*>(del_s + NPAD);  // [NW][NPAD]
  uint32_t* keepT = allowT + NW * NPAD;                           // [NW][NPAD]  (reading the keep bits from global instead, to fit a
                                                                  // third block per CU, measured slower: 100 -> 110 us)
  const int tid = threadIdx.x, bh = blockIdx.x, b = bh / a.H, h = bh % a.H;
  const int N = a.N, Dm = a.H * HD;
  const int64_t ld = 3 * (int64_t)Dm;
  const bf16_t* qbase = a.qkv + (int64_t)b * N * ld + h * HD;
  stage_tile(Qs, qbase, ld, N, NPAD, tid);
  stage_tile(dOs, a.dout + (int64_t)b * N * Dm + h * HD, Dm, N, NPAD, tid);
  for (int qi = tid; qi < NPAD; qi += blockDim.x) {
    lse_s[qi] = qi < N ? a.lse2[(int64_t)bh * N + qi] : INFINITY;
    del_s[qi] = qi < N ? a.delta[(int64_t)bh * N + qi] : 0.f;
  }
  const uint32_t* ap = a.allow + b * a.allow_sb + h * a.allow_sh;
  for (int c = tid; c < NPAD * NW; c += blockDim.x) {
    const int qi = c / NW, w = c - qi * NW;
    allowT[w * NPAD + qi] = qi < N ? ap[(int64_t)qi * NW + w] : 0u;
    keepT[w * NPAD + qi] = (qi < N && a.keep) ? a.keep[((int64_t)bh * N + qi) * NW + w] : 0xffffffffu;
  }
  __syncthreads();

  const int lane = tid & 63, wave = tid >> 6, i = lane & 15, g = lane >> 4;
  const int nwaves = blockDim.x >> 6;
  for (int kt = wave; kt * 16 < N; kt += nwaves) {
    const int key = kt * 16 + i, kc = key < N ? key : N - 1;
    bf16x8 kf[2], vf[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      kf[ks] = *reinterpret_cast<const bf16x8*>(qbase + Dm + (int64_t)kc * ld + 32 * ks + 8 * g);
      vf[ks] = *reinterpret_cast<const bf16x8*>(qbase + 2 * Dm + (int64_t)kc * ld + 32 * ks + 8 * g);
    }
    const int wsel = kt >> 1, bit = key & 31;
    f32x4 dk[4], dv[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) { dk[dt] = f32x4{0.f, 0.f, 0.f, 0.f}; dv[dt] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    for (int s = 0; s < a.nkt / 2; ++s) {
      float pv[8], dsv[8];
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        const int t = 2 * s + half;
        f32x4 acc_s = {0.f, 0.f, 0.f, 0.f}, acc_dp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          acc_s = __builtin_amdgcn_mfma_f32_16x16x32_bf16(lds_row_frag(Qs, 16 * t + i, 4 * ks + g), kf[ks], acc_s, 0, 0, 0);
          acc_dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(lds_row_frag(dOs, 16 * t + i, 4 * ks + g), vf[ks], acc_dp, 0, 0, 0);
        }
        const int q4 = 16 * t + 4 * g;
        const f32x4 l4 = *reinterpret_cast<const f32x4*>(lse_s + q4);
        const f32x4 d4 = *reinterpret_cast<const f32x4*>(del_s + q4);
        const uint4 a4 = *reinterpret_cast<const uint4*>(allowT + wsel * NPAD + q4);
        const uint4 k4 = *reinterpret_cast<const uint4*>(keepT + wsel * NPAD + q4);
        const unsigned aa[4] = {a4.x, a4.y, a4.z, a4.w}, kk[4] = {k4.x, k4.y, k4.z, k4.w};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const bool allowed = (aa[r] >> bit) & 1u, kept = (kk[r] >> bit) & 1u;
          const float p = allowed ? __builtin_amdgcn_exp2f(acc_s[r] * a.scale_log2 - l4[r]) : 0.f;
          const float dpe = kept ? acc_dp[r] * a.inv_keep : 0.f;
          pv[half * 4 + r] = kept ? p * a.inv_keep : 0.f;
          dsv[half * 4 + r] = p * (dpe - d4[r]) * a.scale;
        }
      }
      bf16x8 pa, pl, dsa, dsl;
      split_pack8(pv, pa, pl);
      split_pack8(dsv, dsa, dsl);
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        const bf16x8 dot_ = lds_col_frag(dOs, s, dt, i, g), qt_ = lds_col_frag(Qs, s, dt, i, g);
        dv[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(dot_, pa, dv[dt], 0, 0, 0);
        dv[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(dot_, pl, dv[dt], 0, 0, 0);
        dk[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qt_, dsa, dk[dt], 0, 0, 0);
        dk[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qt_, dsl, dk[dt], 0, 0, 0);
      }
    }
    if (key < N) {
      bf16_t* dst = a.dqkv + ((int64_t)b * N + key) * ld + h * HD + 4 * g;
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        *reinterpret_cast<uint2*>(dst + Dm + 16 * dt) = make_uint2(pack_bf16x2(dk[dt][0], dk[dt][1]), pack_bf16x2(dk[dt][2], dk[dt][3]));
        *reinterpret_cast<uint2*>(dst + 2 * Dm + 16 * dt) = make_uint2(pack_bf16x2(dv[dt][0], dv[dt][1]), pack_bf16x2(dv[dt][2], dv[dt][3]));
      }
    }
  }
}

}  // namespace

namespace attn {

int pick_nkt(int N) {
  const int need = (N + 15) / 16;
  const int opts[] = {2, 4, 8, 12, 16, 24};
  for (int o : opts)
    if (o >= need) return o;
  return -1;
}

int fill_common(AttnArgs& a, int B, int N, int H, int head_dim, float scale, float p_drop) {
  SAM_REQUIRE(head_dim == HD, "sam_attn: head_dim must be 64 (got %d)", head_dim);
  SAM_REQUIRE(B > 0 && N > 0 && H > 0, "sam_attn: empty problem B=%d N=%d H=%d", B, N, H);
  SAM_REQUIRE(p_drop >= 0.f && p_drop < 1.f, "sam_attn: p_drop=%f out of [0,1)", p_drop);
  SAM_REQUIRE(scale > 0.f, "sam_attn: scale=%f must be positive", scale);
  const int nkt = pick_nkt(N);
  SAM_REQUIRE(nkt > 0, "sam_attn: N=%d exceeds the 384-key single-pass limit", N);
  a.B = B; a.N = N; a.H = H; a.nkt = nkt; a.NW = nkt / 2;
  a.scale = scale; a.scale_log2 = scale * LOG2E;
  a.p_drop = p_drop;
  a.thr16 = dropout_thr16(p_drop);
  a.inv_keep = a.thr16 ? 1.0f / (1.0f - (float)a.thr16 / 65536.0f) : 1.0f;
  return SAM_OK;
}

}  // namespace attn

namespace {

// threads per block: long sequences (16 / 24 key tiles: the stress shape's 350 tokens) need 64-96 KB of LDS per block, i.e. ONE block per CU, and
// run 8 waves (two per SIMD); 12 key tiles (N = 182) run 512 threads, two blocks per CU
template <int NKT, bool DEC, bool DROP, int NT>
int launch_fwd_nt(const AttnArgs& a, hipStream_t st) {
  const size_t lds = (size_t)2 * NKT * 16 * ROW_BYTES + 64;
  static bool once = false;
  if (!once) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(attn_fwd_kernel<NKT, DEC, DROP, NT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    once = true;
  }
  attn_fwd_kernel<NKT, DEC, DROP, NT><<<dim3(a.B * a.H), dim3(NT), lds, st>>>(a);
  SAM_LAUNCH_CHECK();
  return SAM_OK;
}

// 12 key tiles (N = 182): 512 threads, two blocks per CU.  Measured and dropped in round 4 (profiles/r4m_micro_attn.txt, git history of this file): 256 / 384
// threads per block (slower by 4-9 %), and a 12-wave block owning THREE heads with the next head's tiles prefetched (one launch round instead of one and a
// half at B = 64: 30.5 us against 28.6 -- twelve waves in lock step hide less latency than two independent 8-wave blocks).
template <int NKT, bool DEC, bool DROP>
int launch_fwd_d(const AttnArgs& a, hipStream_t st) {
  return launch_fwd_nt<NKT, DEC, DROP, (NKT <= 2 ? 128 : NKT <= 8 ? 256 : 512)>(a, st);
}

template <bool DEC>
int launch_fwd_any(const AttnArgs& a, hipStream_t st) {
  const bool drop = !DEC && a.thr16 != 0;
#define SAM_FWD_CASE(K) case K: return drop ? launch_fwd_d<K, DEC, !DEC>(a, st) : launch_fwd_d<K, DEC, false>(a, st);
  switch (a.nkt) {
    SAM_FWD_CASE(2) SAM_FWD_CASE(4) SAM_FWD_CASE(8) SAM_FWD_CASE(12) SAM_FWD_CASE(16) SAM_FWD_CASE(24)
  }
#undef SAM_FWD_CASE
  return SAM_ERR_UNSUPPORTED;
}

}  // namespace

extern "C" int sam_attn_fwd_dec_shared(const void* qkv_enc, const void* qkv_dec, const uint32_t* allow, int64_t allow_stride_b, int64_t allow_stride_h, int B, int group,
                                       int N, int n_dec, int H, int head_dim, float scale, void* out_dec, void* stream) {
  AttnArgs a = {};
  int rc = fill_common(a, B, N, H, head_dim, scale, 0.f);
  if (rc) return rc;
  SAM_REQUIRE(qkv_enc && qkv_dec && allow && out_dec, "sam_attn_fwd_dec: null pointer");
  SAM_REQUIRE(n_dec > 0 && n_dec < N, "sam_attn_fwd_dec: n_dec=%d outside (0,%d)", n_dec, N);
  SAM_REQUIRE(group >= 1 && B % group == 0, "sam_attn_fwd_dec_shared: B=%d is not a whole number of groups of %d", B, group);
  a.qkv = (const bf16_t*)qkv_enc; a.qkv_dec = (const bf16_t*)qkv_dec; a.out_dec = (bf16_t*)out_dec; a.n_enc = N - n_dec; a.kv_group = group;
  a.allow = allow; a.allow_sb = allow_stride_b; a.allow_sh = allow_stride_h; a.q_begin = N - n_dec;
  return launch_fwd_any<true>(a, (hipStream_t)stream);
}
extern "C" int sam_attn_fwd_dec(const void* qkv_enc, const void* qkv_dec, const uint32_t* allow, int64_t allow_stride_b, int64_t allow_stride_h, int B, int N, int n_dec,
                                int H, int head_dim, float scale, void* out_dec, void* stream) {
  return sam_attn_fwd_dec_shared(qkv_enc, qkv_dec, allow, allow_stride_b, allow_stride_h, B, 1, N, n_dec, H, head_dim, scale, out_dec, stream);
}

extern "C" int sam_attn_words_per_row(int N) {
  const int nkt = pick_nkt(N);
  return nkt > 0 ? nkt / 2 : -1;
}

static int attn_fwd_impl(const void* qkv, const uint32_t* allow, int64_t allow_stride_b, int64_t allow_stride_h, int B, int N, int H,
                         int head_dim, float scale, float p_drop, uint64_t seed, uint64_t offset, int q_begin, void* out, void* out_lo, float* lse2,
                         uint32_t* keep, void* stream) {
  AttnArgs a = {};
  int rc = fill_common(a, B, N, H, head_dim, scale, p_drop);
  if (rc) return rc;
  SAM_REQUIRE(qkv && allow && out && lse2, "sam_attn_fwd: null pointer");
  SAM_REQUIRE(a.thr16 == 0 || keep, "sam_attn_fwd: dropout needs a keep-bits buffer");
  a.qkv = (const bf16_t*)qkv; a.out_w = (bf16_t*)out; a.out_lo_w = (bf16_t*)out_lo; a.allow = allow; a.allow_sb = allow_stride_b; a.allow_sh = allow_stride_h;
  a.lse2_w = lse2; a.keep_w = keep; a.q_begin = q_begin;
  a.seed_lo = (unsigned)seed; a.seed_hi = (unsigned)(seed >> 32); a.off_lo = (unsigned)offset; a.off_hi = (unsigned)(offset >> 32);
  a.rng_state = sam_get_rng_state();
  return launch_fwd_any<false>(a, (hipStream_t)stream);
}

extern "C" int sam_attn_fwd(const void* qkv, const uint32_t* allow, int64_t allow_stride_b, int64_t allow_stride_h, int B, int N, int H,
                            int head_dim, float scale, float p_drop, uint64_t seed, uint64_t offset, void* out, float* lse2,
                            uint32_t* keep, void* stream) {
  return attn_fwd_impl(qkv, allow, allow_stride_b, allow_stride_h, B, N, H, head_dim, scale, p_drop, seed, offset, 0, out, nullptr, lse2, keep, stream);
}

extern "C" int sam_attn_fwd_train(const void* qkv, const uint32_t* allow, int64_t allow_stride_b, int64_t allow_stride_h, int B, int N, int H,
                                  int head_dim, float scale, float p_drop, uint64_t seed, uint64_t offset, void* out, void* out_lo, float* lse2,
                                  uint32_t* keep, void* stream) {
  SAM_REQUIRE(out_lo, "sam_attn_fwd_train: out_lo is null (use sam_attn_fwd when the residual is not wanted)");
  return attn_fwd_impl(qkv, allow, allow_stride_b, allow_stride_h, B, N, H, head_dim, scale, p_drop, seed, offset, 0, out, out_lo, lse2, keep, stream);
}

extern "C" int sam_attn_fwd_rows(const void* qkv, const uint32_t* allow, int64_t allow_stride_b, int64_t allow_stride_h, int B, int N, int H,
                                 int head_dim, float scale, int q_begin, void* out, float* lse2, void* stream) {
  SAM_REQUIRE(q_begin >= 0 && q_begin < N, "sam_attn_fwd_rows: q_begin=%d outside [0,%d)", q_begin, N);
  return attn_fwd_impl(qkv, allow, allow_stride_b, allow_stride_h, B, N, H, head_dim, scale, 0.f, 0, 0, q_begin, out, nullptr, lse2, nullptr, stream);
}

extern "C" int sam_attn_bwd(const void* dout, const void* qkv, const float* lse2, const uint32_t* allow,
                            int64_t allow_stride_b, int64_t allow_stride_h, const uint32_t* keep, int B, int N, int H, int head_dim,
                            float scale, float p_drop, void* dqkv, float* delta_ws, void* stream) {
  AttnArgs a = {};
  int rc = fill_common(a, B, N, H, head_dim, scale, p_drop);
  if (rc) return rc;
  SAM_REQUIRE(dout && qkv && lse2 && allow && dqkv && delta_ws, "sam_attn_bwd: null pointer");
  SAM_REQUIRE(a.thr16 == 0 || keep, "sam_attn_bwd: dropout needs the keep bits written by sam_attn_fwd");
  a.qkv = (const bf16_t*)qkv; a.dout = (const bf16_t*)dout; a.dqkv = (bf16_t*)dqkv;
  a.allow = allow; a.allow_sb = allow_stride_b; a.allow_sh = allow_stride_h; a.keep = a.thr16 ? keep : nullptr;
  a.lse2 = lse2; a.delta = delta_ws;
  hipStream_t st = (hipStream_t)stream;
  const int NPAD = a.nkt * 16;
  const size_t lds1 = (size_t)2 * NPAD * ROW_BYTES;
  const size_t lds2 = lds1 + (size_t)NPAD * 8 + (size_t)2 * a.NW * NPAD * 4;
  static bool once = false;
  if (!once) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_dq_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_dq_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_dq_kernel<8>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_dq_kernel<12>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_dq_kernel<16>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_dq2_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_dkdv_kernel<256>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_dkdv_kernel<512>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    once = true;
  }
  const dim3 blk(a.nkt <= 12 ? 256 : 512);
  switch (a.nkt) {
    case 2: attn_bwd_dq_kernel<2><<<dim3(B * H), blk, lds1, st>>>(a); break;
    case 4: attn_bwd_dq_kernel<4><<<dim3(B * H), blk, lds1, st>>>(a); break;
    case 8: attn_bwd_dq_kernel<8><<<dim3(B * H), blk, lds1, st>>>(a); break;
    case 12: attn_bwd_dq_kernel<12><<<dim3(B * H), blk, lds1, st>>>(a); break;
    case 16: attn_bwd_dq_kernel<16><<<dim3(B * H), blk, lds1, st>>>(a); break;
    case 24: attn_bwd_dq2_kernel<<<dim3(B * H), blk, lds1, st>>>(a); break;
    default: return SAM_ERR_UNSUPPORTED;
  }
  SAM_LAUNCH_CHECK();
  // dK / dV: 59 KB of LDS per block at 12 key tiles = two blocks per CU, so the 768 (batch, head) blocks of a B = 64 step take one and a half rounds;
  // with 8 waves per block (1.5 key tiles per wave instead of 3) the half-empty second round runs at two waves per SIMD instead of one:
  // dq + dkdv 87 -> 77 us (tools/bench_attn.py).  Short sequences (TextBert's 20 tokens = 2 key tiles) have nothing for 8 waves to do.
  if (a.nkt < 8) attn_bwd_dkdv_kernel<256><<<dim3(B * H), dim3(256), lds2, st>>>(a);
  else attn_bwd_dkdv_kernel<512><<<dim3(B * H), dim3(512), lds2, st>>>(a);
  SAM_LAUNCH_CHECK();
  return SAM_OK;
}
