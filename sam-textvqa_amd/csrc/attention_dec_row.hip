// Decoding-time attention for ONE new decoder row per beam (gfx950) -- the core of DecodeSession._step_inc (beam search, sam/sa_m4c.py:304-314 +
// sam/beam_search.py:84-160 over the attention of sam/sa_m4c.py:563-598).
//
// Step t of a beam search only needs the attention output of decoder row t of every beam: one query row against the sample's text / object / OCR
// keys (step-invariant, shared by the beams of a sample, cached by the first pass) and the beam's own decoder rows 0..t.  sam_attn_fwd_dec would
// stage all N keys / values per (beam, head) and run 16-row MFMA strips to produce one useful row: 3840 blocks x 46 KB per layer and step at
// beam 5.  Here one block per (SAMPLE, head) stages the encoder keys / values once (bf16, as they sit in the cache) and its four waves take the
// sample's beams in turn: lanes = keys for q.k (fp32 dot products over the 64 dims from LDS), a wave-wide softmax, lanes = dims for P.V.  HBM/L2
// traffic per layer and step: samples x heads x 2 x n_enc x 128 B (33 MB at B = 64, N = 182) instead of beams x that.
//
// Arithmetic: fp32 throughout on the bf16 cache values (the MFMA kernels round P to fp16 and V to block-scaled fp16; this one does not round at all
// -- inside the 1e-3 bound of either).  Masked keys take no part; a row with no allowed key returns zeros (sa_m4c.py:586-590).
#include "attn_common.h"

namespace {
using namespace attn;

constexpr int NT = 256, NWV = NT / 64;
constexpr int KROW = HD * 2 + 4;        // bytes per staged row: 33 dwords, so that 64 lanes reading 64 different rows at one dim hit 64 different banks
constexpr int MAXKEYS = 384;

struct RowArgs {
  const bf16_t* qkv_enc; const bf16_t* qkv_dec; const uint32_t* allow; bf16_t* out;
  int64_t allow_sb, allow_sh, ldo;
  int B0, group, N, n_enc, n_dec, t, H, NW;
  float scale;
};

__device__ __forceinline__ float bf(unsigned short v) { return __uint_as_float((unsigned)v << 16); }
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
// acc + a.lo * b.lo + a.hi * b.hi on packed bf16 pairs (v_dot2c_f32_bf16): one instruction where unpack + two multiply-adds were four
__device__ __forceinline__ float dot2_bf16(unsigned a, unsigned b, float acc) {
  return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, a), __builtin_bit_cast(bf16x2_t, b), acc, false);
}

__global__ __launch_bounds__(NT) void attn_dec_row_kernel(RowArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* Ks = smem;                                             // [n_enc][KROW]
  unsigned char* Vs = smem + (size_t)a.n_enc * KROW;                    // [n_enc][KROW]
  float* qs = reinterpret_cast<float*>(Vs + (size_t)a.n_enc * KROW);    // [NWV][64]
  float* ps = qs + NWV * HD;                                            // [NWV][MAXKEYS]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b0 = blockIdx.x / a.H, h = blockIdx.x % a.H;
  const int Dm = a.H * HD;
  const int64_t ld = 3 * (int64_t)Dm;
  // ---- the sample's encoder keys / values: 8 x 16-byte chunks per row and operand
  const bf16_t* ebase = a.qkv_enc + (int64_t)b0 * a.N * ld + h * HD;
  for (int c = tid; c < a.n_enc * 8; c += NT) {
    const int row = c >> 3, ch = c & 7;
    const uint4 kv = *reinterpret_cast<const uint4*>(ebase + (int64_t)row * ld + Dm + ch * 8);
    const uint4 vv = *reinterpret_cast<const uint4*>(ebase + (int64_t)row * ld + 2 * Dm + ch * 8);
    unsigned* kd = reinterpret_cast<unsigned*>(Ks + row * KROW + ch * 16);
    unsigned* vd = reinterpret_cast<unsigned*>(Vs + row * KROW + ch * 16);
    kd[0] = kv.x; kd[1] = kv.y; kd[2] = kv.z; kd[3] = kv.w;             // (rows are 4-byte aligned only: dword stores)
    vd[0] = vv.x; vd[1] = vv.y; vd[2] = vv.z; vd[3] = vv.w;
  }
  __syncthreads();
  const int q_row = a.n_enc + a.t;
  const uint32_t* ap = a.allow + b0 * a.allow_sb + h * a.allow_sh + (int64_t)q_row * a.NW;
  const int nkeys = a.n_enc + a.t + 1;                                  // decoder keys after position t are masked by the causal part anyway
  float* q_w = qs + wave * HD;
  float* p_w = ps + wave * MAXKEYS;
  for (int j = wave; j < a.group; j += NWV) {
    const int b = b0 * a.group + j;
    const bf16_t* dbase = a.qkv_dec + (int64_t)b * a.n_dec * ld + h * HD;
    // the query row as 32 packed pairs, in registers for the whole beam (lanes 0..31 fetch a dword each, everybody reads all of them back)
    unsigned* q_u = reinterpret_cast<unsigned*>(q_w);
    if (lane < HD / 2) q_u[lane] = reinterpret_cast<const unsigned*>(dbase + (int64_t)a.t * ld)[lane];
    __builtin_amdgcn_wave_barrier();
    unsigned qp[HD / 2];
#pragma unroll
    for (int c = 0; c < HD / 8; ++c) {
      const uint4 t4 = *reinterpret_cast<const uint4*>(q_u + 4 * c);
      qp[4 * c] = t4.x; qp[4 * c + 1] = t4.y; qp[4 * c + 2] = t4.z; qp[4 * c + 3] = t4.w;
    }
    // ---- scores: lane = key
    float sc[MAXKEYS / 64];
    float mx = -INFINITY;
#pragma unroll
    for (int m = 0; m < MAXKEYS / 64; ++m) {
      const int key = lane + 64 * m;
      sc[m] = -INFINITY;
      if (64 * m >= nkeys) continue;
      const bool in = key < nkeys && ((ap[key >> 5] >> (key & 31)) & 1u);
      float acc = 0.f;
      if (key < a.n_enc) {
        const unsigned* kr = reinterpret_cast<const unsigned*>(Ks + key * KROW);
        float a1 = 0.f;
#pragma unroll
        for (int d2 = 0; d2 < HD / 2; d2 += 2) { acc = dot2_bf16(qp[d2], kr[d2], acc); a1 = dot2_bf16(qp[d2 + 1], kr[d2 + 1], a1); }
        acc += a1;
      } else if (key < nkeys) {
        const unsigned* kr = reinterpret_cast<const unsigned*>(dbase + (int64_t)(key - a.n_enc) * ld + Dm);
        float a1 = 0.f;
#pragma unroll
        for (int d2 = 0; d2 < HD / 2; d2 += 2) { acc = dot2_bf16(qp[d2], kr[d2], acc); a1 = dot2_bf16(qp[d2 + 1], kr[d2 + 1], a1); }
        acc += a1;
      }
      if (in) { sc[m] = acc * a.scale; mx = fmaxf(mx, sc[m]); }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    float sum = 0.f;
#pragma unroll
    for (int m = 0; m < MAXKEYS / 64; ++m) {
      if (64 * m >= nkeys) continue;
      const float p = sc[m] == -INFINITY ? 0.f : __expf(sc[m] - mx);
      sum += p;
      p_w[lane + 64 * m] = p;
    }
    sum = wave_sum(sum);
    __builtin_amdgcn_wave_barrier();
    // ---- output: lane = dim
    float o_acc = 0.f;
    const unsigned short* vcol = reinterpret_cast<const unsigned short*>(Vs) + lane;
    for (int key = 0; key < a.n_enc; ++key) o_acc = fmaf(p_w[key], bf(vcol[key * (KROW / 2)]), o_acc);
    for (int key = a.n_enc; key < nkeys; ++key) o_acc = fmaf(p_w[key], bf(dbase[(int64_t)(key - a.n_enc) * ld + 2 * Dm + lane]), o_acc);
    const float res = sum > 0.f ? o_acc / sum : 0.f;
    const unsigned u = __float_as_uint(res);
    a.out[(int64_t)b * a.ldo + h * HD + lane] = (bf16_t)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);      // round to nearest even
    __builtin_amdgcn_wave_barrier();
  }
}

}  // namespace

extern "C" int sam_attn_dec_row(const void* qkv_enc, const void* qkv_dec, const uint32_t* allow, int64_t allow_stride_b, int64_t allow_stride_h, int B, int group, int N,
                                int n_dec, int t, int H, int head_dim, float scale, void* out, int64_t ldo, void* stream) {
  SAM_REQUIRE(qkv_enc && qkv_dec && allow && out, "sam_attn_dec_row: null pointer");
  SAM_REQUIRE(head_dim == HD, "sam_attn_dec_row: head_dim must be 64 (got %d)", head_dim);
  SAM_REQUIRE(B > 0 && H > 0 && group >= 1 && B % group == 0, "sam_attn_dec_row: B=%d must be a positive multiple of group=%d", B, group);
  SAM_REQUIRE(n_dec > 0 && n_dec < N && t >= 0 && t < n_dec, "sam_attn_dec_row: need 0 < n_dec < N and 0 <= t < n_dec (N=%d n_dec=%d t=%d)", N, n_dec, t);
  SAM_REQUIRE(scale > 0.f && ldo >= (int64_t)H * HD, "sam_attn_dec_row: scale must be positive, ldo >= H * 64");
  const int nkt = pick_nkt(N);
  SAM_REQUIRE(nkt > 0 && N <= MAXKEYS, "sam_attn_dec_row: N=%d exceeds the %d-key limit", N, MAXKEYS);
  RowArgs a;
  a.qkv_enc = (const bf16_t*)qkv_enc; a.qkv_dec = (const bf16_t*)qkv_dec; a.allow = allow; a.out = (bf16_t*)out;
  a.allow_sb = allow_stride_b; a.allow_sh = allow_stride_h; a.ldo = ldo;
  a.B0 = B / group; a.group = group; a.N = N; a.n_enc = N - n_dec; a.n_dec = n_dec; a.t = t; a.H = H; a.NW = nkt / 2;
  a.scale = scale;
  const size_t lds = (size_t)2 * a.n_enc * KROW + (size_t)NWV * HD * 4 + (size_t)NWV * MAXKEYS * 4;
  static bool once = false;
  if (!once) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(attn_dec_row_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    once = true;
  }
  attn_dec_row_kernel<<<dim3(a.B0 * H), dim3(NT), lds, (hipStream_t)stream>>>(a);
  SAM_LAUNCH_CHECK();
  return SAM_OK;
}
