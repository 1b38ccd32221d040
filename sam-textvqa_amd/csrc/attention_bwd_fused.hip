// One-pass backward of the fused relation-masked attention (gfx950): autograd of /root/reference/sam/sa_m4c.py:563-598.
//
// Rounds 1-3 ran two kernels (dQ with the query per lane, dK/dV with the key per lane), each re-reading q|k|v, dO and both bit planes and
// each recomputing S, P and dP -- 38 VALU instructions per score between them, 1.45-1.6x the one-pass HBM traffic.  Here one workgroup
// owns a (batch, head) and computes every score ONCE:
//
//   staging   Q, K, dO -> LDS as block-scaled fp16 (attn_common.h); V stays in registers (each wave only ever needs its own 16 keys);
//             delta[q] = rowsum(dO * O) from the forward's output AND its bf16 residual (sam_attn_fwd_train): the bf16-rounded O alone
//             costs 3e-3 of max in dQ/dK, O + residual is exact to 2^-17; lse, delta and the transposed allow / keep words -> LDS
//   phase 1   wave w owns key tile w (16 keys) and walks the query tiles: S^T and dP^T tiles by MFMA (key = lane & 15, query = 4g + r, the
//             mask as the accumulator's initial value), P = exp2(..), dS = P * (keep * dP - delta) * scale; P~ and dS feed the dV / dK MFMAs
//             directly as B operands (contraction over the query index they hold in-lane); dS is also written to LDS as fp16 [key][query]
//   phase 2   wave w owns QUERY tile w: dQ = dS . K with dS read back through ds_read_b64_tr_b16 (the transposition the two orientations
//             need costs one 8-byte LDS write and one transposed read per 4 scores instead of a second S / exp / mask pass)
//
// ~10 VALU per score + ~3 for the fp16 staging.  LDS at 12 key tiles (N <= 192): 3 x 24 KB tiles + 72 KB dS + 11 KB = 155 KB, one
// block of 12 waves per CU.  Longer sequences keep the two-kernel form (attention.hip).
#include "attn_common.h"
extern "C" int sam_get_cu_reserve(void);

using namespace attn;

namespace {

template <int NKT>
struct FusedLds {
  static constexpr int NPAD = NKT * 16, NW = NKT / 2;
  static constexpr int TILE = NPAD * ROW_BYTES;
  static constexpr int DS_ROW = NPAD * 2;                      // bytes per key row of dS [key][query] fp16
  static constexpr int OFF_Q = 0, OFF_DO = TILE, OFF_K = 2 * TILE, OFF_DS = 3 * TILE;
  static constexpr int OFF_NL = OFF_DS + NPAD * DS_ROW;        // f32 [NPAD]  14 - lse2 (-inf for dead / padded rows)
  static constexpr int OFF_ND = OFF_NL + NPAD * 4;             // f32 [NPAD]  -delta * scale * 2^(eS - 14)
  static constexpr int OFF_NA = OFF_ND + NPAD * 4;             // u32 [NW][NPAD]  ~allow, transposed
  static constexpr int OFF_KP = OFF_NA + NW * NPAD * 4;        // u32 [NW][NPAD]  keep, transposed
  static constexpr int OFF_RED = OFF_KP + NW * NPAD * 4;       // u32 [NKT][2]
  static constexpr int BYTES = OFF_RED + NKT * 8;
};

// dS [key][query] fp16 in LDS: 32-byte blocks (one query tile each) XOR-swizzled by the key row so that the 8-byte writes of phase 1
// (16 keys x one 4-query piece) and the transposed reads of phase 2 (8 keys x one 32-byte block) are both bank-conflict-free
template <int NKT>
__device__ __forceinline__ int ds_off(int key, int qtile, int piece) {
  constexpr int BM = NKT >= 4 ? 3 : NKT - 1;
  int blk = qtile ^ ((key >> 1) & BM);
  if (NKT == 8) blk ^= (key & 1) << 2;       // 256-byte rows: consecutive rows would otherwise start on the same bank
  return key * (NKT * 32) + (blk << 5) + ((piece ^ ((key >> 3) & 1)) << 3);
}

// Blocks are PERSISTENT: block x walks the (batch, head) pairs x, x + gridDim.x, ... (three each at B = 64 on 256 CUs; the host sizes the grid).
// A head's 155 KB of LDS allow one block per CU, so nothing else could cover a block's load phase (140 KB per head, ~5 us when every CU asks
// at once: a third of the kernel in the first version, 54 us).  The NEXT head's q | k | v | dO chunks are therefore requested into registers at
// the start of phase 1 (32 VGPRs), its output / residual rows, bit planes and log-sum-exps at the start of phase 2 (when the dK / dV
// accumulators are dead): the memory system works on head j+1 while the matrix cores and the VALU work on head j.
template <int NKT, bool DROP>
__global__ __launch_bounds__(64 * NKT) void attn_bwd_fused_kernel(AttnArgs a) {
  typedef FusedLds<NKT> L;
  constexpr int NPAD = L::NPAD, NW = L::NW, NT = 64 * NKT;
  constexpr int NBW = (NPAD * NW + NT - 1) / NT;            // bit-plane words per thread
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* Qs = smem + L::OFF_Q;
  unsigned char* dOs = smem + L::OFF_DO;
  unsigned char* Ks = smem + L::OFF_K;
  unsigned char* dSs = smem + L::OFF_DS;
  float* nl_s = reinterpret_cast<float*>(smem + L::OFF_NL);
  float* nd_s = reinterpret_cast<float*>(smem + L::OFF_ND);
  uint32_t* naT = reinterpret_cast<uint32_t*>(smem + L::OFF_NA);
  uint32_t* kpT = reinterpret_cast<uint32_t*>(smem + L::OFF_KP);
  unsigned* red = reinterpret_cast<unsigned*>(smem + L::OFF_RED);

  const int N = a.N, Dm = a.H * HD, BH = a.B * a.H;
  const int64_t ld = 3 * (int64_t)Dm;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // The lane id is re-derived (opaquely) at the top of every head: everything computed from it -- two dozen LDS addresses -- would otherwise be
  // hoisted out of the head loop as loop invariants and kept in registers across it (168 VGPRs + 75 spilled in the first persistent build).
  int lane, tid, i, g, key, kc;
  auto derive_lane = [&]() {
    int l;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
    lane = l; tid = wave * 64 + l; i = l & 15; g = l >> 4;
    key = wave * 16 + i; kc = key < N ? key : N - 1;
  };
  derive_lane();

  // raw chunks of the head being staged (two 16-byte chunks per thread of each of Q, K, dO, O, O_lo; this lane's V fragments; bit words; lse)
  uint4 rq[2], rk[2], rd[2], rv[2], roh[2], rol[2];
  unsigned rna[NBW], rkp[NBW];
  float rls = 0.f;
  auto load_qkvd = [&](int bh) {                              // every load unconditional at a clamped row
    const int b = bh / a.H, h = bh - b * a.H;
    const bf16_t* qbase = a.qkv + (int64_t)b * N * ld + h * HD;
    const bf16_t* dobase = a.dout + (int64_t)b * N * Dm + h * HD;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int c = tid + j * NT, row = c >> 3, ch = c & 7, rc = row < N ? row : N - 1;
      const bf16_t* src = qbase + (int64_t)rc * ld + ch * 8;
      rq[j] = *reinterpret_cast<const uint4*>(src);
      rk[j] = *reinterpret_cast<const uint4*>(src + Dm);
      rd[j] = *reinterpret_cast<const uint4*>(dobase + (int64_t)rc * Dm + ch * 8);
    }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) rv[ks] = *reinterpret_cast<const uint4*>(qbase + 2 * Dm + (int64_t)kc * ld + 32 * ks + 8 * g);
  };
  auto load_rest = [&](int bh) {
    const int b = bh / a.H, h = bh - b * a.H;
    const bf16_t* obase = a.out + (int64_t)b * N * Dm + h * HD;
    const bf16_t* olbase = a.out_lo + (int64_t)b * N * Dm + h * HD;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int c = tid + j * NT, row = c >> 3, ch = c & 7, rc = row < N ? row : N - 1;
      roh[j] = *reinterpret_cast<const uint4*>(obase + (int64_t)rc * Dm + ch * 8);
      rol[j] = *reinterpret_cast<const uint4*>(olbase + (int64_t)rc * Dm + ch * 8);
    }
    const uint32_t* ap = a.allow + b * a.allow_sb + h * a.allow_sh;
#pragma unroll
    for (int j = 0; j < NBW; ++j) {
      int c = tid + j * NT;
      c = c < N * NW ? c : N * NW - 1;                        // the bit planes are [N][NW] row-major: word c of the head
      rna[j] = ap[c];
      if (DROP) rkp[j] = a.keep[(int64_t)bh * N * NW + c];
    }
    rls = a.lse2[(int64_t)bh * N + (tid < N ? tid : N - 1)];
  };

  int bh = blockIdx.x;
  if (bh >= BH) return;
  load_qkvd(bh);
  load_rest(bh);
  for (; bh < BH; bh += gridDim.x) {
    derive_lane();
    const int b = bh / a.H, h = bh - b * a.H;
    const int bh_next = bh + (int)gridDim.x;
    // ---- staging: maxima -> power-of-two scales -> fp16 tiles, delta, bit planes
    unsigned mq = 0, mk = 0, md = 0, mv = 0;
    float dot[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int c = tid + j * NT, row = c >> 3;
      if (row >= N) { rq[j] = make_uint4(0, 0, 0, 0); rk[j] = rq[j]; rd[j] = rq[j]; }
      mq = absmax_acc4(mq, rq[j]); mk = absmax_acc4(mk, rk[j]); md = absmax_acc4(md, rd[j]);
      // delta partial: dO . (O + O_lo) over this chunk's 8 columns, on the bf16 values themselves (the fp16 image of dO is exact)
      typedef __attribute__((ext_vector_type(2))) __bf16 hb2;
      float acc = 0.f;
      const unsigned dv4[4] = {rd[j].x, rd[j].y, rd[j].z, rd[j].w}, hv[4] = {roh[j].x, roh[j].y, roh[j].z, roh[j].w}, lv[4] = {rol[j].x, rol[j].y, rol[j].z, rol[j].w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(hb2, dv4[e]), __builtin_bit_cast(hb2, hv[e]), acc, false);
        acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(hb2, dv4[e]), __builtin_bit_cast(hb2, lv[e]), acc, false);
      }
      acc += __shfl_xor(acc, 1); acc += __shfl_xor(acc, 2); acc += __shfl_xor(acc, 4);      // the 8 chunks of a row sit in 8 neighbouring lanes
      dot[j] = acc;
    }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      if (key >= N) rv[ks] = make_uint4(0, 0, 0, 0);
      mv = absmax_acc4(mv, rv[ks]);
    }
    unsigned m01 = absmax_fold(mq) | (absmax_fold(mk) << 16), m23 = absmax_fold(md) | (absmax_fold(mv) << 16);     // two 15-bit maxima per word
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      m01 = pk_max_u16(m01, (unsigned)__shfl_xor((int)m01, o));
      m23 = pk_max_u16(m23, (unsigned)__shfl_xor((int)m23, o));
    }
    if (lane == 0) *reinterpret_cast<uint2*>(red + 2 * wave) = make_uint2(m01, m23);
    // transposed bit planes and the per-query scalars that do not depend on the scales
#pragma unroll
    for (int j = 0; j < NBW; ++j) {
      const int c = tid + j * NT;
      if (c < NPAD * NW) {
        const int qi = c / NW, w = c - qi * NW;
        naT[w * NPAD + qi] = qi < N ? ~rna[j] : 0xffffffffu;
        if (DROP) kpT[w * NPAD + qi] = qi < N ? rkp[j] : 0u;
      }
    }
    if (tid < NPAD) nl_s[tid] = tid < N ? (float)P_SHIFT - rls : -INFINITY;
    __syncthreads();
    unsigned mm01 = 0, mm23 = 0;
#pragma unroll
    for (int w = 0; w < NKT; ++w) {
      const uint2 r2 = *reinterpret_cast<const uint2*>(red + 2 * w);
      mm01 = pk_max_u16(mm01, r2.x); mm23 = pk_max_u16(mm23, r2.y);
    }
    const int cq = scale_c_of(mm01 & 0xffffu), ck = scale_c_of(mm01 >> 16), cd = scale_c_of(mm23 & 0xffffu), cv = scale_c_of(mm23 >> 16);
    // exponents: X16 = X * 2^eX with eX = 112 - cX.  dS16 = dS * 2^eS with eS = eD + eV - 22 - ds_sh (|dS16| < 2^15 whatever the data: see
    // DESIGN.md), which makes the factor on the raw dP accumulator a launch constant (a.ds_c1).
    const int eq = 112 - cq, ek = 112 - ck, ed = 112 - cd, ev = 112 - cv, es = ed + ev - 22 - a.ds_sh;
    {
      const unsigned sq = csub_of(cq), sk = csub_of(ck), sd = csub_of(cd);
      const int e_nd = ed + ev - 36 - a.ds_sh;            // -delta * scale * 2^(eS - 14); ldexp: the exponent may leave [-126, 127] for extreme inputs
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int c = tid + j * NT, row = c >> 3, ch = c & 7, off = tile_off(row, ch);
        *reinterpret_cast<uint4*>(Qs + off) = bf2h_pk4(rq[j], sq);
        *reinterpret_cast<uint4*>(Ks + off) = bf2h_pk4(rk[j], sk);
        *reinterpret_cast<uint4*>(dOs + off) = bf2h_pk4(rd[j], sd);
        if (ch == 0) nd_s[row] = ldexpf(-a.scale * dot[j], e_nd);
      }
    }
    const unsigned sv = csub_of(cv);
    const f16x8 vf[2] = {__builtin_bit_cast(f16x8, bf2h_pk4(rv[0], sv)), __builtin_bit_cast(f16x8, bf2h_pk4(rv[1], sv))};
    __syncthreads();
    if (bh_next < BH) load_qkvd(bh_next);                  // in flight during phase 1
    __builtin_amdgcn_sched_barrier(0);

    // ---- phase 1: this wave's 16 keys against every query tile
    const f16x8 kf[2] = {as_f16(lds_row_frag(Ks, 16 * wave + i, g)), as_f16(lds_row_frag(Ks, 16 * wave + i, 4 + g))};
    const int wsel = wave >> 1;
    const unsigned bit = (unsigned)((wave & 1) * 16 + i);
    const float cs = ldexpf(a.scale_log2, -eq - ek);       // raw score -> log2 domain
    f32x4 dk[4], dv[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) { dk[dt] = f32x4{0.f, 0.f, 0.f, 0.f}; dv[dt] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
    for (int s = 0; s < NKT / 2; ++s) {
      typedef __attribute__((ext_vector_type(4))) unsigned u4;
      u4 ppk, dpk;
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        const int t = 2 * s + half, q4 = 16 * t + 4 * g;
        const uint4 na4 = *reinterpret_cast<const uint4*>(naT + wsel * NPAD + q4);
        const f32x4 nl4 = *reinterpret_cast<const f32x4*>(nl_s + q4);
        const f32x4 nd4 = *reinterpret_cast<const f32x4*>(nd_s + q4);
        const unsigned nav[4] = {na4.x, na4.y, na4.z, na4.w};
        f32x4 acc_s, acc_dp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int r = 0; r < 4; ++r) acc_s[r] = __int_as_float(__builtin_amdgcn_sbfe((int)nav[r], bit, 1u) & (int)0xff800000u);     // -inf where masked
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          acc_s = __builtin_amdgcn_mfma_f32_16x16x32_f16(as_f16(lds_row_frag(Qs, 16 * t + i, 4 * ks + g)), kf[ks], acc_s, 0, 0, 0);
          acc_dp = __builtin_amdgcn_mfma_f32_16x16x32_f16(as_f16(lds_row_frag(dOs, 16 * t + i, 4 * ks + g)), vf[ks], acc_dp, 0, 0, 0);
        }
        float p[4], ds[4];
        if (DROP) {
          const uint4 kp4 = *reinterpret_cast<const uint4*>(kpT + wsel * NPAD + q4);
          const unsigned kpv[4] = {kp4.x, kp4.y, kp4.z, kp4.w};
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int km = __builtin_amdgcn_sbfe((int)kpv[r], bit, 1u);
            const float pr = __builtin_amdgcn_exp2f(fmaf(acc_s[r], cs, nl4[r]));                       // 2^14 * P
            const float dpe = __int_as_float(__float_as_int(acc_dp[r]) & km);                          // keep * raw dP
            ds[r] = pr * fmaf(dpe, a.ds_c1, nd4[r]);
            p[r] = __int_as_float(__float_as_int(pr) & km);                                            // dropped probabilities
          }
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            p[r] = __builtin_amdgcn_exp2f(fmaf(acc_s[r], cs, nl4[r]));
            ds[r] = p[r] * fmaf(acc_dp[r], a.ds_c1, nd4[r]);
          }
        }
        ppk[2 * half] = pack_f16x2(p[0], p[1]); ppk[2 * half + 1] = pack_f16x2(p[2], p[3]);
        dpk[2 * half] = pack_f16x2(ds[0], ds[1]); dpk[2 * half + 1] = pack_f16x2(ds[2], ds[3]);
        *reinterpret_cast<uint2*>(dSs + ds_off<NKT>(16 * wave + i, t, g)) = make_uint2(dpk[2 * half], dpk[2 * half + 1]);
      }
      const f16x8 pa = __builtin_bit_cast(f16x8, ppk), dsa = __builtin_bit_cast(f16x8, dpk);
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        dv[dt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(as_f16(lds_col_frag(dOs, s, dt, i, g)), pa, dv[dt], 0, 0, 0);
        dk[dt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(as_f16(lds_col_frag(Qs, s, dt, i, g)), dsa, dk[dt], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    {
      // dV = sum_q P~ dO: the accumulator carries 2^14 (P) * 2^eD (dO) and lacks inv_keep; dK = sum_q dS Q carries 2^(eS + eQ).
      // 16 bytes per store: v_permlane16_swap pairs the 4-column fragments of two neighbouring lane groups (columns 32 jp + 16 (g & 1) + 8 (g >> 1) .. +7)
      const float fv = ldexpf(a.inv_keep, -P_SHIFT - ed);
      const int ekq = -es - eq;
      bf16_t* dst = a.dqkv + ((int64_t)b * N + kc) * ld + h * HD + 16 * (g & 1) + 8 * (g >> 1);
#pragma unroll
      for (int jp = 0; jp < 2; ++jp) {
        float vk[8], vv[8];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const auto sk2 = __builtin_amdgcn_permlane16_swap(__float_as_uint(dk[2 * jp][r]), __float_as_uint(dk[2 * jp + 1][r]), false, false);
          const auto sv2 = __builtin_amdgcn_permlane16_swap(__float_as_uint(dv[2 * jp][r]), __float_as_uint(dv[2 * jp + 1][r]), false, false);
          vk[r] = ldexpf(__uint_as_float(sk2[0]), ekq); vk[4 + r] = ldexpf(__uint_as_float(sk2[1]), ekq);
          vv[r] = __uint_as_float(sv2[0]) * fv; vv[4 + r] = __uint_as_float(sv2[1]) * fv;
        }
        if (key < N) {
          *reinterpret_cast<uint4*>(dst + Dm + 32 * jp) = make_uint4(pack_bf16x2(vk[0], vk[1]), pack_bf16x2(vk[2], vk[3]), pack_bf16x2(vk[4], vk[5]), pack_bf16x2(vk[6], vk[7]));
          *reinterpret_cast<uint4*>(dst + 2 * Dm + 32 * jp) = make_uint4(pack_bf16x2(vv[0], vv[1]), pack_bf16x2(vv[2], vv[3]), pack_bf16x2(vv[4], vv[5]), pack_bf16x2(vv[6], vv[7]));
        }
      }
    }
    __syncthreads();
    if (bh_next < BH) load_rest(bh_next);                  // in flight during phase 2
    __builtin_amdgcn_sched_barrier(0);

    // ---- phase 2: dQ of query tile `wave` = dS . K over all keys, dS^T fragments by transposed LDS reads
    {
      f32x4 dq[4];
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) dq[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < NKT / 2; ++s) {
        const int kr = 32 * s + 4 * g + (i >> 2);
        const f16x8 dsf = as_f16(cat4(lds_read_tr16(dSs + ds_off<NKT>(kr, wave, i & 3)), lds_read_tr16(dSs + ds_off<NKT>(kr + 16, wave, i & 3))));
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) dq[dt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(as_f16(lds_col_frag(Ks, s, dt, i, g)), dsf, dq[dt], 0, 0, 0);
      }
      const int q = wave * 16 + i, qcl = q < N ? q : N - 1, eqk = -es - ek;
      bf16_t* dst = a.dqkv + ((int64_t)b * N + qcl) * ld + h * HD + 16 * (g & 1) + 8 * (g >> 1);
#pragma unroll
      for (int jp = 0; jp < 2; ++jp) {
        float v[8];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const auto sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(dq[2 * jp][r]), __float_as_uint(dq[2 * jp + 1][r]), false, false);
          v[r] = ldexpf(__uint_as_float(sw[0]), eqk); v[4 + r] = ldexpf(__uint_as_float(sw[1]), eqk);
        }
        if (q < N) *reinterpret_cast<uint4*>(dst + 32 * jp) = make_uint4(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7]));
      }
    }
    __syncthreads();             // K, dS and the scalars are rewritten by the next head's staging
  }
}

// ---- sequences of 193 .. 384 tokens (the stress shape's 350): the same one-pass scheme on 192 x 192 SUB-PROBLEMS -------------------------
// Q | dO and K | V of a head no longer fit the CU's LDS next to the dS exchange, so a block walks (key chunk kh) x (query chunk qh), chunks of
// 192 rows: every sub-problem stages its Q, dO (and for qh = 0 its K, V) chunk, runs phase 1 (S, P, dP, dS once per score; dK / dV of the key
// chunk accumulate over qh in registers) and phase 2 (dQ of the query chunk accumulates over kh in registers: 2 x 16 more).  The fp16 block scales are
// per CHUNK; an accumulator that lives across chunks is moved to the new chunk's power-of-two scale before the first MFMA adds to it (exact).
// delta of a query row is computed once (kh = 0) and kept in LDS.  No cross-head prefetch here (one round of 1.5 heads per CU at the stress batch).
template <bool DROP>
__global__ __launch_bounds__(768) void attn_bwd_fused_long_kernel(AttnArgs a) {
  constexpr int NKT = 12;
  typedef FusedLds<NKT> L;
  constexpr int NPAD = L::NPAD, NW = L::NW, NT = 64 * NKT, CH = NPAD;
  constexpr int NBW = (NPAD * NW + NT - 1) / NT;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* Qs = smem + L::OFF_Q;
  unsigned char* dOs = smem + L::OFF_DO;
  unsigned char* Ks = smem + L::OFF_K;
  unsigned char* dSs = smem + L::OFF_DS;
  float* nl_s = reinterpret_cast<float*>(smem + L::OFF_NL);
  float* nd_s = reinterpret_cast<float*>(smem + L::OFF_ND);
  uint32_t* naT = reinterpret_cast<uint32_t*>(smem + L::OFF_NA);
  uint32_t* kpT = reinterpret_cast<uint32_t*>(smem + L::OFF_KP);
  unsigned* red = reinterpret_cast<unsigned*>(smem + L::OFF_RED);
  float* dl_s = reinterpret_cast<float*>(smem + L::BYTES);          // raw delta of both query chunks, f32 [2][NPAD]

  const int bh = blockIdx.x, b = bh / a.H, h = bh % a.H;
  const int N = a.N, Dm = a.H * HD, NWR = a.NW;                       // NWR: words per allow / keep row in memory (12)
  const int64_t ld = 3 * (int64_t)Dm;
  const bf16_t* qbase = a.qkv + (int64_t)b * N * ld + h * HD;
  const bf16_t* obase = a.out + (int64_t)b * N * Dm + h * HD;
  const bf16_t* olbase = a.out_lo + (int64_t)b * N * Dm + h * HD;
  const bf16_t* dobase = a.dout + (int64_t)b * N * Dm + h * HD;
  const uint32_t* ap = a.allow + b * a.allow_sb + h * a.allow_sh;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int n_chunks = (N + CH - 1) / CH;                               // 2
  int lane, tid, i, g;
  auto derive_lane = [&]() {      // re-derived (opaquely) per sub-problem: otherwise every LDS address becomes a loop invariant held in a register across both loops
    int l;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
    lane = l; tid = wave * 64 + l; i = l & 15; g = l >> 4;
  };
  derive_lane();

  f32x4 dq[2][4];
#pragma unroll
  for (int c = 0; c < 2; ++c)
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) dq[c][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
  int e_dq[2] = {0, 0};                                                // exponent the dQ accumulators currently carry (es + ek of the last sub-problem)

  for (int kh = 0; kh < n_chunks; ++kh) {
    const int k0 = kh * CH;
    f32x4 dk[4], dv[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) { dk[dt] = f32x4{0.f, 0.f, 0.f, 0.f}; dv[dt] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    int e_dk = 0, e_dv = 0, ck = 0, cv = 0, eq_last = 0;
    f16x8 vf[2], kf[2];
    for (int qh = 0; qh < n_chunks; ++qh) {
      derive_lane();
      const int q0 = qh * CH;
      // ---- staging of the sub-problem
      uint4 rq[2], rk[2], rd[2], rv[2];
      unsigned mq = 0, mk = 0, md = 0, mv = 0;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int c = tid + j * NT, row = c >> 3, ch = c & 7;
        const int qr = q0 + row, qrc = qr < N ? qr : N - 1, kr = k0 + row, krc = kr < N ? kr : N - 1;
        rq[j] = *reinterpret_cast<const uint4*>(qbase + (int64_t)qrc * ld + ch * 8);
        rd[j] = *reinterpret_cast<const uint4*>(dobase + (int64_t)qrc * Dm + ch * 8);
        if (qr >= N) { rq[j] = make_uint4(0, 0, 0, 0); rd[j] = rq[j]; }
        mq = absmax_acc4(mq, rq[j]); md = absmax_acc4(md, rd[j]);
        if (qh == 0) {
          rk[j] = *reinterpret_cast<const uint4*>(qbase + Dm + (int64_t)krc * ld + ch * 8);
          if (kr >= N) rk[j] = make_uint4(0, 0, 0, 0);
          mk = absmax_acc4(mk, rk[j]);
        }
        if (kh == 0) {      // delta of this query chunk: dO . (O + O_lo), once
          typedef __attribute__((ext_vector_type(2))) __bf16 hb2;
          const uint4 oh = *reinterpret_cast<const uint4*>(obase + (int64_t)qrc * Dm + ch * 8);
          const uint4 ol = *reinterpret_cast<const uint4*>(olbase + (int64_t)qrc * Dm + ch * 8);
          float acc = 0.f;
          const unsigned dv4[4] = {rd[j].x, rd[j].y, rd[j].z, rd[j].w}, hv[4] = {oh.x, oh.y, oh.z, oh.w}, lv[4] = {ol.x, ol.y, ol.z, ol.w};
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(hb2, dv4[e]), __builtin_bit_cast(hb2, hv[e]), acc, false);
            acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(hb2, dv4[e]), __builtin_bit_cast(hb2, lv[e]), acc, false);
          }
          acc += __shfl_xor(acc, 1); acc += __shfl_xor(acc, 2); acc += __shfl_xor(acc, 4);
          if (ch == 0) dl_s[qh * NPAD + row] = acc;
        }
      }
      if (qh == 0) {
        const int key = k0 + wave * 16 + i, kc = key < N ? key : N - 1;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          rv[ks] = *reinterpret_cast<const uint4*>(qbase + 2 * Dm + (int64_t)kc * ld + 32 * ks + 8 * g);
          if (key >= N) rv[ks] = make_uint4(0, 0, 0, 0);
          mv = absmax_acc4(mv, rv[ks]);
        }
      }
      unsigned m01 = absmax_fold(mq) | (absmax_fold(mk) << 16), m23 = absmax_fold(md) | (absmax_fold(mv) << 16);
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        m01 = pk_max_u16(m01, (unsigned)__shfl_xor((int)m01, o));
        m23 = pk_max_u16(m23, (unsigned)__shfl_xor((int)m23, o));
      }
      if (lane == 0) *reinterpret_cast<uint2*>(red + 2 * wave) = make_uint2(m01, m23);
      for (int c = tid; c < NPAD * NW; c += NT) {
        const int qi = c / NW, w = c - qi * NW, qr = q0 + qi;
        const bool in_row = qr < N && kh * NW + w < NWR;          // (16 key tiles: 8-word rows, the second chunk has two of them)
        naT[w * NPAD + qi] = in_row ? ~ap[(int64_t)qr * NWR + kh * NW + w] : 0xffffffffu;
        if (DROP) kpT[w * NPAD + qi] = in_row ? a.keep[((int64_t)bh * N + qr) * NWR + kh * NW + w] : 0u;
      }
      for (int qi = tid; qi < NPAD; qi += NT) nl_s[qi] = q0 + qi < N ? (float)P_SHIFT - a.lse2[(int64_t)bh * N + q0 + qi] : -INFINITY;
      __syncthreads();
      unsigned mm01 = 0, mm23 = 0;
#pragma unroll
      for (int w = 0; w < NKT; ++w) {
        const uint2 r2 = *reinterpret_cast<const uint2*>(red + 2 * w);
        mm01 = pk_max_u16(mm01, r2.x); mm23 = pk_max_u16(mm23, r2.y);
      }
      const int cq = scale_c_of(mm01 & 0xffffu), cd = scale_c_of(mm23 & 0xffffu);
      if (qh == 0) { ck = scale_c_of(mm01 >> 16); cv = scale_c_of(mm23 >> 16); }
      const int eq = 112 - cq, ek = 112 - ck, ed = 112 - cd, ev = 112 - cv, es = ed + ev - 22 - a.ds_sh;
      {
        const unsigned sq = csub_of(cq), sk = csub_of(ck), sd = csub_of(cd);
        const int e_nd = ed + ev - 36 - a.ds_sh;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int c = tid + j * NT, row = c >> 3, ch = c & 7, off = tile_off(row, ch);
          *reinterpret_cast<uint4*>(Qs + off) = bf2h_pk4(rq[j], sq);
          *reinterpret_cast<uint4*>(dOs + off) = bf2h_pk4(rd[j], sd);
          if (qh == 0) *reinterpret_cast<uint4*>(Ks + off) = bf2h_pk4(rk[j], sk);
        }
        for (int qi = tid; qi < NPAD; qi += NT) nd_s[qi] = ldexpf(-a.scale * dl_s[qh * NPAD + qi], e_nd);     // (dl_s of this chunk: written before the barrier above)
      }
      if (qh == 0) {
        const unsigned sv = csub_of(cv);
        vf[0] = __builtin_bit_cast(f16x8, bf2h_pk4(rv[0], sv));
        vf[1] = __builtin_bit_cast(f16x8, bf2h_pk4(rv[1], sv));
      }
      __syncthreads();
      if (qh == 0) { kf[0] = as_f16(lds_row_frag(Ks, 16 * wave + i, g)); kf[1] = as_f16(lds_row_frag(Ks, 16 * wave + i, 4 + g)); }
      // accumulators that live across chunks: onto this sub-problem's scales
      {
        const int n_dk = es + eq, n_dv = P_SHIFT + ed;
        if (qh > 0 && (n_dk != e_dk || n_dv != e_dv)) {
#pragma unroll
          for (int dt = 0; dt < 4; ++dt)
#pragma unroll
            for (int r = 0; r < 4; ++r) { dk[dt][r] = ldexpf(dk[dt][r], n_dk - e_dk); dv[dt][r] = ldexpf(dv[dt][r], n_dv - e_dv); }
        }
        e_dk = n_dk; e_dv = n_dv; eq_last = eq;
      }
      // ---- phase 1
      const int wsel = wave >> 1;
      const unsigned bit = (unsigned)((wave & 1) * 16 + i);
      const float cs = ldexpf(a.scale_log2, -eq - ek);
#pragma unroll
      for (int s = 0; s < NKT / 2; ++s) {
        typedef __attribute__((ext_vector_type(4))) unsigned u4;
        u4 ppk, dpk;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
          const int t = 2 * s + half, q4 = 16 * t + 4 * g;
          const uint4 na4 = *reinterpret_cast<const uint4*>(naT + wsel * NPAD + q4);
          const f32x4 nl4 = *reinterpret_cast<const f32x4*>(nl_s + q4);
          const f32x4 nd4 = *reinterpret_cast<const f32x4*>(nd_s + q4);
          const unsigned nav[4] = {na4.x, na4.y, na4.z, na4.w};
          f32x4 acc_s, acc_dp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int r = 0; r < 4; ++r) acc_s[r] = __int_as_float(__builtin_amdgcn_sbfe((int)nav[r], bit, 1u) & (int)0xff800000u);
#pragma unroll
          for (int ks = 0; ks < 2; ++ks) {
            acc_s = __builtin_amdgcn_mfma_f32_16x16x32_f16(as_f16(lds_row_frag(Qs, 16 * t + i, 4 * ks + g)), kf[ks], acc_s, 0, 0, 0);
            acc_dp = __builtin_amdgcn_mfma_f32_16x16x32_f16(as_f16(lds_row_frag(dOs, 16 * t + i, 4 * ks + g)), vf[ks], acc_dp, 0, 0, 0);
          }
          float p[4], ds[4];
          if (DROP) {
            const uint4 kp4 = *reinterpret_cast<const uint4*>(kpT + wsel * NPAD + q4);
            const unsigned kpv[4] = {kp4.x, kp4.y, kp4.z, kp4.w};
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int km = __builtin_amdgcn_sbfe((int)kpv[r], bit, 1u);
              const float pr = __builtin_amdgcn_exp2f(fmaf(acc_s[r], cs, nl4[r]));
              const float dpe = __int_as_float(__float_as_int(acc_dp[r]) & km);
              ds[r] = pr * fmaf(dpe, a.ds_c1, nd4[r]);
              p[r] = __int_as_float(__float_as_int(pr) & km);
            }
          } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              p[r] = __builtin_amdgcn_exp2f(fmaf(acc_s[r], cs, nl4[r]));
              ds[r] = p[r] * fmaf(acc_dp[r], a.ds_c1, nd4[r]);
            }
          }
          ppk[2 * half] = pack_f16x2(p[0], p[1]); ppk[2 * half + 1] = pack_f16x2(p[2], p[3]);
          dpk[2 * half] = pack_f16x2(ds[0], ds[1]); dpk[2 * half + 1] = pack_f16x2(ds[2], ds[3]);
          *reinterpret_cast<uint2*>(dSs + ds_off<NKT>(16 * wave + i, t, g)) = make_uint2(dpk[2 * half], dpk[2 * half + 1]);
        }
        const f16x8 pa = __builtin_bit_cast(f16x8, ppk), dsa = __builtin_bit_cast(f16x8, dpk);
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
          dv[dt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(as_f16(lds_col_frag(dOs, s, dt, i, g)), pa, dv[dt], 0, 0, 0);
          dk[dt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(as_f16(lds_col_frag(Qs, s, dt, i, g)), dsa, dk[dt], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      __syncthreads();
      // ---- phase 2: dQ of query tile (qh, wave) += dS . K over this key chunk
      {
        const int n_dq = es + ek;
        if (kh > 0 && n_dq != e_dq[qh]) {
#pragma unroll
          for (int dt = 0; dt < 4; ++dt)
#pragma unroll
            for (int r = 0; r < 4; ++r) dq[qh][dt][r] = ldexpf(dq[qh][dt][r], n_dq - e_dq[qh]);
        }
        e_dq[qh] = n_dq;
#pragma unroll
        for (int s = 0; s < NKT / 2; ++s) {
          const int kr = 32 * s + 4 * g + (i >> 2);
          const f16x8 dsf = as_f16(cat4(lds_read_tr16(dSs + ds_off<NKT>(kr, wave, i & 3)), lds_read_tr16(dSs + ds_off<NKT>(kr + 16, wave, i & 3))));
#pragma unroll
          for (int dt = 0; dt < 4; ++dt) dq[qh][dt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(as_f16(lds_col_frag(Ks, s, dt, i, g)), dsf, dq[qh][dt], 0, 0, 0);
        }
      }
      __syncthreads();                 // Q, dO, dS and the scalars are rewritten by the next sub-problem
    }
    // ---- dK, dV of this key chunk
    {
      derive_lane();
      const int key = k0 + wave * 16 + i, kc = key < N ? key : N - 1;
      const float fv = a.inv_keep;
      bf16_t* dst = a.dqkv + ((int64_t)b * N + kc) * ld + h * HD + 16 * (g & 1) + 8 * (g >> 1);
#pragma unroll
      for (int jp = 0; jp < 2; ++jp) {
        float vk[8], vv[8];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const auto sk2 = __builtin_amdgcn_permlane16_swap(__float_as_uint(dk[2 * jp][r]), __float_as_uint(dk[2 * jp + 1][r]), false, false);
          const auto sv2 = __builtin_amdgcn_permlane16_swap(__float_as_uint(dv[2 * jp][r]), __float_as_uint(dv[2 * jp + 1][r]), false, false);
          vk[r] = ldexpf(__uint_as_float(sk2[0]), -e_dk); vk[4 + r] = ldexpf(__uint_as_float(sk2[1]), -e_dk);
          vv[r] = ldexpf(__uint_as_float(sv2[0]) * fv, -e_dv); vv[4 + r] = ldexpf(__uint_as_float(sv2[1]) * fv, -e_dv);
        }
        if (key < N) {
          *reinterpret_cast<uint4*>(dst + Dm + 32 * jp) = make_uint4(pack_bf16x2(vk[0], vk[1]), pack_bf16x2(vk[2], vk[3]), pack_bf16x2(vk[4], vk[5]), pack_bf16x2(vk[6], vk[7]));
          *reinterpret_cast<uint4*>(dst + 2 * Dm + 32 * jp) = make_uint4(pack_bf16x2(vv[0], vv[1]), pack_bf16x2(vv[2], vv[3]), pack_bf16x2(vv[4], vv[5]), pack_bf16x2(vv[6], vv[7]));
        }
      }
    }
    (void)eq_last;
  }
  // ---- dQ of both query chunks
  derive_lane();
#pragma unroll
  for (int qh = 0; qh < 2; ++qh) {
    if (qh < n_chunks) {
      const int q = qh * CH + wave * 16 + i, qcl = q < N ? q : N - 1;
      bf16_t* dst = a.dqkv + ((int64_t)b * N + qcl) * ld + h * HD + 16 * (g & 1) + 8 * (g >> 1);
#pragma unroll
      for (int jp = 0; jp < 2; ++jp) {
        float v[8];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const auto sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(dq[qh][2 * jp][r]), __float_as_uint(dq[qh][2 * jp + 1][r]), false, false);
          v[r] = ldexpf(__uint_as_float(sw[0]), -e_dq[qh]); v[4 + r] = ldexpf(__uint_as_float(sw[1]), -e_dq[qh]);
        }
        if (q < N) *reinterpret_cast<uint4*>(dst + 32 * jp) = make_uint4(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7]));
      }
    }
  }
}

int launch_fused_long(const AttnArgs& a, hipStream_t st) {
  constexpr int BYTES = FusedLds<12>::BYTES + 2 * FusedLds<12>::NPAD * 4;
  static_assert(BYTES <= 160 * 1024, "the long-sequence one-pass backward must fit one CU's LDS");
  static bool once = false;
  if (!once) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_fused_long_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, BYTES);
    hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_fused_long_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, BYTES);
    once = true;
  }
  if (a.keep) attn_bwd_fused_long_kernel<true><<<dim3(a.B * a.H), dim3(768), BYTES, st>>>(a);
  else attn_bwd_fused_long_kernel<false><<<dim3(a.B * a.H), dim3(768), BYTES, st>>>(a);
  SAM_LAUNCH_CHECK();
  return SAM_OK;
}

template <int NKT>
int launch_fused(const AttnArgs& a, hipStream_t st) {
  typedef FusedLds<NKT> L;
  static bool once = false;
  static int n_phys = 0;
  if (!once) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_fused_kernel<NKT, true>), hipFuncAttributeMaxDynamicSharedMemorySize, L::BYTES);
    hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_fused_kernel<NKT, false>), hipFuncAttributeMaxDynamicSharedMemorySize, L::BYTES);
    int dev = 0;
    hipGetDevice(&dev);
    hipDeviceGetAttribute(&n_phys, hipDeviceAttributeMultiprocessorCount, dev);
    if (n_phys <= 0) n_phys = 256;
    once = true;
  }
  // (CUs withheld from persistent grids while collectives run beside the backward pass: sam_set_cu_reserve, gemm_common.h)
  const int reserve = sam_get_cu_reserve();
  const int n_cu = (reserve > 0 && n_phys - reserve >= 8) ? n_phys - reserve : n_phys;
  // blocks per CU the LDS footprint admits; with one (12 key tiles) the grid is one block per CU and every block walks several heads,
  // prefetching the next one's operands while it computes (the loop degenerates to a single trip when there are at least as many slots as heads)
  const int per_cu = (160 * 1024) / L::BYTES > 0 ? (160 * 1024) / L::BYTES : 1;
  const int BH = a.B * a.H;
  int grid = BH;
  if (per_cu == 1) {
    const char* e = getenv("SAM_ATTN_BWD_PERSIST");
    if (!(e && e[0] == '0')) {
      // equal trip counts: ceil(BH / n_cu) heads per block, as many blocks as that needs
      const int per_block = (BH + n_cu - 1) / n_cu;
      grid = (BH + per_block - 1) / per_block;
    }
  }
  if (a.keep) attn_bwd_fused_kernel<NKT, true><<<dim3(grid), dim3(64 * NKT), L::BYTES, st>>>(a);
  else attn_bwd_fused_kernel<NKT, false><<<dim3(grid), dim3(64 * NKT), L::BYTES, st>>>(a);
  SAM_LAUNCH_CHECK();
  return SAM_OK;
}

}  // namespace

// largest sequence the one-pass backward takes: up to 192 tokens a head is one problem (12 key tiles: Q, K, dO tiles + the dS exchange fill the CU's
// 160 KB of LDS), 193 .. 384 tokens run as 2 x 2 sub-problems of 192 inside one block (attn_bwd_fused_long_kernel)
extern "C" int sam_attn_bwd_fused_max_n(void) { return 384; }

extern "C" int sam_attn_bwd_fused(const void* dout, const void* qkv, const void* out, const void* out_lo, const float* lse2, const uint32_t* allow,
                                  int64_t allow_stride_b, int64_t allow_stride_h, const uint32_t* keep, int B, int N, int H, int head_dim,
                                  float scale, float p_drop, void* dqkv, void* stream) {
  AttnArgs a = {};
  int rc = fill_common(a, B, N, H, head_dim, scale, p_drop);
  if (rc) return rc;
  SAM_REQUIRE(dout && qkv && out && out_lo && lse2 && allow && dqkv, "sam_attn_bwd_fused: null pointer");
  SAM_REQUIRE(a.thr16 == 0 || keep, "sam_attn_bwd_fused: dropout needs the keep bits written by sam_attn_fwd_train");
  a.qkv = (const bf16_t*)qkv; a.dout = (const bf16_t*)dout; a.out = (const bf16_t*)out; a.out_lo = (const bf16_t*)out_lo; a.dqkv = (bf16_t*)dqkv;
  a.allow = allow; a.allow_sb = allow_stride_b; a.allow_sh = allow_stride_h; a.keep = a.thr16 ? keep : nullptr;
  a.lse2 = lse2;
  // |dS| < 2 * 64 * max|dO| * max|V| * inv_keep * scale; with kappa = inv_keep * scale < 2^sh the fp16 image dS * 2^eS, eS = eD + eV - 22 - sh,
  // stays below 2^15, and the factor that takes the raw dO16 . V16 accumulator to dS16 / P16 is the constant kappa * 2^(-36 - sh)
  const float kappa = a.inv_keep * scale;
  int sh = 0;
  frexpf(kappa, &sh);                       // kappa = m * 2^sh, m in [0.5, 1)
  SAM_REQUIRE(sh > -60 && sh < 60, "sam_attn_bwd_fused: scale=%g out of range", scale);
  a.ds_sh = sh;
  a.ds_c1 = ldexpf(kappa, -36 - sh);
  hipStream_t st = (hipStream_t)stream;
  if (a.nkt > 12) {
    return launch_fused_long(a, st);                     // 16 or 24 key tiles: 2 x 2 sub-problems of 192
  }
  switch (a.nkt) {
    case 2: return launch_fused<2>(a, st);
    case 4: return launch_fused<4>(a, st);
    case 8: return launch_fused<8>(a, st);
    case 12: return launch_fused<12>(a, st);
  }
  return SAM_ERR_UNSUPPORTED;
}
