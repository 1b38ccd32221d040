// Greedy decoding steps 1 .. S-1 of SA-M4C (sam/sa_m4c.py:285-302) as ONE persistent launch (gfx950).
//
// What a step is.  Under the prefix-LM mask (sa_m4c.py:834-844) decoder row t sees the encoder rows and decoder rows 0..t only, and its input
// token is fixed once step t-1 has picked it: row t of the LAST of the reference's twelve full forwards equals row t computed at step t.  So a
// step runs ONE new row per sample (B rows, 64 at the bench batch) through the embedding of the previous prediction (sa_m4c.py:928-948), the
// encoder layers (sa_m4c.py:660-684,718-760) against the cached keys / values of the first full pass, the classifier + pointer network
// (sa_m4c.py:270-278, 866-897) and the argmax (sa_m4c.py:299-302) -- about 0.2 GFLOP and 70 MB of weights + cache per step.  As separate
// launches that is ~56 kernels of 5-10 us each (measured 595 us per captured step); here every stage is a PHASE of one kernel, separated by
// grid barriers (one block per CU, all resident; an agent-scope atomic counter), and all S-1 steps run inside the launch.
//
// Phases of a step (7 per layer + 2):  per layer  Q  x -> q|k|v row, written straight into the layer's [B, N, 3D] cache at row n_enc + t
//                                            A  one wave per (sample, head): scores over the cache rows 0 .. n_enc + t under the allow bits, softmax, PV
//                                            O  ctx Wo^T, split-K 4 -> fp32 partials          F1  partials + bias + x -> LayerNorm -> x1
//                                            G  gelu(x1 W1^T + b1) -> h                        H   h W2^T, split-K 4 -> partials
//                                            F2 partials + bias + x1 -> LayerNorm -> x (and the row of the final hidden states after the last layer)
//                        then                C  classifier logits (fp32, into row t of the score block) | pointer-network query, split-K 4
//                                            P  one block per sample: query . OCR keys (+ -10000 on padded OCR slots), argmax over [logits | pointer
//                                               scores] (first index wins ties), prev_inds[b, t + 1], and the NEXT step's input row x[b]
// Coherence.  Blocks of one launch sit on eight XCDs with private, mutually non-coherent L2s.  A release / acquire fence pair per barrier (write
// back + invalidate the whole L2, every block at once) measured ~30 us per phase (1.46 ms per step); agent-scope (sc1) loads for everything a
// phase reads from an earlier one -- uncached, every block fetching its operand rows from the memory side -- 13 us per GEMM phase and 34 us for the
// attention (0.63 ms per step).  What runs now: every hand-over buffer is written ONCE per launch -- the activations x / x1 / ctx / h live in an
// arena indexed by (step, layer), each slot a whole number of cache lines, and cache row n_enc + t belongs to step t -- with write-through (sc1) stores, s_waitcnt vmcnt(0)
// before the block arrives at the barrier; an address that was never read before the barrier cannot sit stale in any L1 / L2, so the readers use
// ordinary cached loads (one miss per XCD, L2 hits for its other blocks).  The K-split partials (reused) and the logits rows (not line-aligned)
// are read with sc1 loads, by 64 waves / 64 blocks.  Kernel boundaries invalidate the caches, so the arena is reused by the next launch.  Weights, first-pass cache rows, masks: read-only.
// The skinny GEMMs (M = B <= 64 rows per group): a block takes one 16-column slice of the output (and one K split), its four waves the four
// 16-row tiles; operands are staged through LDS once per task (gemm_task below), every global load of the task in flight at once.
#include "common.h"
#include "gemm_common.h"
#include "sam_hip.h"

namespace {

constexpr int NT = 256, MAXL = 8, D = 768, F = 3072, HD = 64, KSPLIT = 4, NCH = D / 256;
constexpr unsigned SPIN_LIMIT = 1u << 20;

struct Best { float v; int i; };
__device__ __forceinline__ Best better(Best a, Best b) { return (b.v > a.v || (b.v == a.v && b.i < a.i)) ? b : a; }

constexpr int AUX_SC1 = 0x10;
typedef unsigned int vu4 __attribute__((ext_vector_type(4)));
typedef unsigned int vu2 __attribute__((ext_vector_type(2)));
typedef __amdgpu_buffer_rsrc_t rsrc_t;
__device__ __forceinline__ rsrc_t rsrc(const void* p) { return __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, 0x7fffffff, 0x00020000); }
__device__ __forceinline__ vu4 ld16(rsrc_t r, int off) { return __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, AUX_SC1); }
__device__ __forceinline__ vu2 ld8(rsrc_t r, int off) { return __builtin_amdgcn_raw_buffer_load_b64(r, off, 0, AUX_SC1); }
__device__ __forceinline__ unsigned ld4(rsrc_t r, int off) { return __builtin_amdgcn_raw_buffer_load_b32(r, off, 0, AUX_SC1); }
__device__ __forceinline__ unsigned short ld2(rsrc_t r, int off) { return __builtin_amdgcn_raw_buffer_load_b16(r, off, 0, AUX_SC1); }
__device__ __forceinline__ void st16(vu4 v, rsrc_t r, int off) { __builtin_amdgcn_raw_buffer_store_b128(v, r, off, 0, AUX_SC1); }
__device__ __forceinline__ void st8(vu2 v, rsrc_t r, int off) { __builtin_amdgcn_raw_buffer_store_b64(v, r, off, 0, AUX_SC1); }
__device__ __forceinline__ void st4(unsigned v, rsrc_t r, int off) { __builtin_amdgcn_raw_buffer_store_b32(v, r, off, 0, AUX_SC1); }
__device__ __forceinline__ void st2(unsigned short v, rsrc_t r, int off) { __builtin_amdgcn_raw_buffer_store_b16(v, r, off, 0, AUX_SC1); }
__device__ __forceinline__ vu4 f4_bits(const f32x4& v) { return vu4{__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3])}; }

struct DLayer {
  const bf16_t *wqkv, *wo, *w1, *w2;
  const float *bqkv, *bo, *b1, *b2, *g1, *be1, *g2, *be2;
  bf16_t* qkv;
  const uint32_t* allow;
  long long allow_sb, allow_sh;
};
struct DArgs {
  DLayer L[MAXL];
  int n_layers, B, Bp, N, n_enc, S, H, NW, V, No, t_begin, t_end;      // Bp = B rounded up to 16 (tiled activation slots)
  long long ldf, ldwc, ld_pos, ld_type;
  float scale_log2, eps, eps_emb, ptr_scale;
  const float *pos_emb, *type_emb, *emb_g, *emb_b;
  const bf16_t *ans_ln, *ocr_ln, *wc, *wq, *ptr_k;
  const float *bc, *bq;
  const unsigned char* ocr_mask;
  long long* prev;
  float *fixed_all, *dyn_all;
  bf16_t* seq;
  bf16_t *x, *x1, *ctx, *h;         // arenas: x [S][L+1][B, D] (layer inputs; slot L = the final hidden row), x1 / ctx [S][L][B, D], h [S][L][B, F]
  float* part;                      // [KSPLIT][B, D], reused
  unsigned* bar;
  int* err;
  long long* prof;          // SAM_DECODE_PROF=1: block 0 stamps wall_clock64() (100 MHz) after every barrier of the first profiled step
};

__device__ __forceinline__ bf16_t* xbuf(const DArgs& a, int t, int li) { return a.x + ((long long)t * (a.n_layers + 1) + li) * a.Bp * D; }
__device__ __forceinline__ bf16_t* x1buf(const DArgs& a, int t, int li) { return a.x1 + ((long long)t * a.n_layers + li) * a.Bp * D; }
__device__ __forceinline__ bf16_t* ctxbuf(const DArgs& a, int t, int li) { return a.ctx + ((long long)t * a.n_layers + li) * a.Bp * D; }
__device__ __forceinline__ bf16_t* hbuf(const DArgs& a, int t, int li) { return a.h + ((long long)t * a.n_layers + li) * a.Bp * F; }

// every block arrives, then waits for the counter to reach `epoch` (monotonic within a launch; zero when a launch begins: the previous one's last block resets it).  false = give up
// (a block failed to arrive within the spin limit, or another block already gave up): the caller returns, the error word stays set.
__device__ __forceinline__ bool grid_sync(const DArgs& a, unsigned& epoch) {
  __shared__ int ok;
  epoch += gridDim.x;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // this wave's sc1 stores have reached the memory side
  if (a.prof && blockIdx.x == 0 && threadIdx.x == 0) a.prof[1016] = wall_clock64();
  __syncthreads();
  if (a.prof && blockIdx.x == 0 && threadIdx.x == 0) a.prof[1017] = wall_clock64();
  if (threadIdx.x == 0) {
    __hip_atomic_fetch_add(a.bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned spins = 0;
    int good = 1;
    while (__hip_atomic_load(a.bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < epoch) {
      __builtin_amdgcn_s_sleep(1);
      if (++spins > SPIN_LIMIT || __hip_atomic_load(a.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) {
        __hip_atomic_store(a.err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        good = 0;
        break;
      }
    }
    ok = good;
  }
  __syncthreads();
  asm volatile("" ::: "memory");
  if (a.prof && blockIdx.x == 0 && threadIdx.x == 0) {
    const unsigned k = epoch / gridDim.x;
    if (k < 1000) a.prof[k] = wall_clock64();
  }
  return ok != 0;
}
#define DS_STAMP(slot) do { if (a.prof && blockIdx.x == 0 && threadIdx.x == 0 && t == a.t_begin + 1 && li == 1) a.prof[1000 + (slot)] = wall_clock64(); } while (0)

// Skinny-GEMM task = (m group of 64 rows, 16-column slice, K split); a wave = one 16 x 16 output tile.  The MFMA fragment layout has lane l read
// row (l & 15): from row-major operands that is 64 different 16-byte pieces per load instruction, one tag lookup each -- measured 6.5 us per
// task, warm or cold (staging through LDS with coalesced loads cost more in registers and barriers than it saved).  So both operands live in the
// FRAGMENT-TILED layout [rows / 16][K / 8][16 rows][8 elements]: the 16 rows' chunks of one k-group are 256 contiguous bytes, a load instruction
// of the wave covers 1 KB contiguous.  The weights are re-tiled once per batch by the host side (sam_decode_layer: "tiled"), the activations are
// written tiled by this kernel's own epilogues.  epi(ks, m, n, acc): the lane owns columns n .. n+3 of row m.
__device__ __forceinline__ long long tiled(int row, int k, int K) { return ((((long long)(row >> 4) * (K >> 3) + (k >> 3)) << 4) + (row & 15)) * 8 + (k & 7); }
template <int KS, typename Epi>
__device__ __forceinline__ void gemm_task(const DArgs& a, const bf16_t* X, int Kx, const bf16_t* W, int Kw, int Nout, int ksplit, int task, Epi epi) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, i = lane & 15, g = lane >> 4;
  const int ntiles = (Nout + 15) >> 4;
  const int ks = task % ksplit, rest = task / ksplit, nt = rest % ntiles, mg = rest / ntiles;
  const int mt = mg * 4 + wave, k8 = ks * KS * 4 + g;
  if (mt * 16 >= a.B) return;
  const bf16_t* wp = W + ((((long long)nt * (Kw >> 3) + k8) << 4) + i) * 8;
  const bf16_t* xp = X + ((((long long)mt * (Kx >> 3) + k8) << 4) + i) * 8;
  bf16x8 wf[KS], xf[KS];
#pragma unroll
  for (int q = 0; q < KS; ++q) {
    wf[q] = *reinterpret_cast<const bf16x8*>(wp + q * 512);          // 4 k-groups x 16 rows x 8 elements per k-step
    xf[q] = *reinterpret_cast<const bf16x8*>(xp + q * 512);
  }
  __builtin_amdgcn_sched_barrier(0);      // every load issued before the first MFMA waits (left alone the scheduler kept ~12 in flight)
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int q = 0; q < KS; ++q) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[q], xf[q], acc, 0, 0, 0);
  const int m = mt * 16 + i;
  if (m < a.B) epi(ks, m, nt * 16 + 4 * g, acc);
}
__device__ __forceinline__ int gemm_tasks(const DArgs& a, int Nout, int ksplit) { return ((Nout + 15) >> 4) * ksplit * ((a.B + 63) >> 6); }

__device__ __forceinline__ void st_bf16x4(bf16_t* p, const float* v) {          // normal store (outputs nobody reads inside the launch)
  uint2 o;
  o.x = pack_bf16x2(v[0], v[1]); o.y = pack_bf16x2(v[2], v[3]);
  *reinterpret_cast<uint2*>(p) = o;
}
__device__ __forceinline__ void ld_bf16x4(const bf16_t* p, float* v) {          // normal load (read-only data)
  const uint2 u = *reinterpret_cast<const uint2*>(p);
  v[0] = bf_lo(u.x); v[1] = bf_hi(u.x); v[2] = bf_lo(u.y); v[3] = bf_hi(u.y);
}
__device__ __forceinline__ void st_bf16x4_sc1(rsrc_t r, long long elem, const float* v) { st8(vu2{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])}, r, (int)(elem * 2)); }
__device__ __forceinline__ void ld_bf16x4_sc1(rsrc_t r, long long elem, float* v) {
  const vu2 u = ld8(r, (int)(elem * 2));
  v[0] = bf_lo(u[0]); v[1] = bf_hi(u[0]); v[2] = bf_lo(u[1]); v[3] = bf_hi(u[1]);
}

// LayerNorm of one 768-wide row held as v[j][e] (element 4 * (lane + 64 j) + e), two-pass like sam_layernorm_fwd
__device__ __forceinline__ void ln_row(float (&v)[NCH][4], const float* gamma, const float* beta, float eps, int lane) {
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < NCH; ++j) s += (v[j][0] + v[j][1]) + (v[j][2] + v[j][3]);
  const float mean = wave_sum(s) / D;
  float q = 0.f;
#pragma unroll
  for (int j = 0; j < NCH; ++j)
#pragma unroll
    for (int e = 0; e < 4; ++e) { const float d = v[j][e] - mean; q += d * d; }
  const float rstd = 1.0f / sqrtf(wave_sum(q) / D + eps);
#pragma unroll
  for (int j = 0; j < NCH; ++j) {
    const int c = 4 * (lane + 64 * j);
    const float4 g4 = *reinterpret_cast<const float4*>(gamma + c), b4 = *reinterpret_cast<const float4*>(beta + c);
    v[j][0] = g4.x * ((v[j][0] - mean) * rstd) + b4.x; v[j][1] = g4.y * ((v[j][1] - mean) * rstd) + b4.y;
    v[j][2] = g4.z * ((v[j][2] - mean) * rstd) + b4.z; v[j][3] = g4.w * ((v[j][3] - mean) * rstd) + b4.w;
  }
}

// x[b] = (tok < V ? LN(answer table)[tok] : LN(OCR rows)[b, tok - V]) + bf16(LN(position[t] + type[tok >= V]))   (sa_m4c.py:928-948, eval: no dropout); one wave
__device__ __forceinline__ void embed_row(const DArgs& a, int b, int t, long long tok) {
  const int lane = threadIdx.x & 63;
  const bool is_ocr = tok >= a.V;
  const float* pe = a.pos_emb + (long long)t * a.ld_pos;
  const float* te = a.type_emb + (is_ocr ? a.ld_type : 0);
  const long long oi = min(max(tok - a.V, 0LL), (long long)a.No - 1);
  const bf16_t* src = is_ocr ? a.ocr_ln + ((long long)b * a.No + oi) * D : a.ans_ln + min(max(tok, 0LL), (long long)a.V - 1) * D;
  float v[NCH][4];
#pragma unroll
  for (int j = 0; j < NCH; ++j) {
    const int c = 4 * (lane + 64 * j);
    const float4 p4 = *reinterpret_cast<const float4*>(pe + c), t4 = *reinterpret_cast<const float4*>(te + c);
    v[j][0] = p4.x + t4.x; v[j][1] = p4.y + t4.y; v[j][2] = p4.z + t4.z; v[j][3] = p4.w + t4.w;
  }
  ln_row(v, a.emb_g, a.emb_b, a.eps_emb, lane);
#pragma unroll
  for (int j = 0; j < NCH; ++j) {
    const int c = 4 * (lane + 64 * j);
    float s4[4], o[4];
    ld_bf16x4(src + c, s4);
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = s4[e] + bf2f(f2bf(v[j][e]));
    st_bf16x4_sc1(rsrc(xbuf(a, t, 0)), tiled(b, c, D), o);
  }
}

// out[row] = LayerNorm(sum of the K-split partials (fixed order) + bias + res[row]); one wave per row
__device__ __forceinline__ void finalize_rows(const DArgs& a, const float* bias, const bf16_t* res, const float* gamma, const float* beta, bf16_t* out, bf16_t* out2,
                                              long long ld2) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const rsrc_t pr = rsrc(a.part), orr = rsrc(out);
  for (int row = blockIdx.x * 4 + wave; row < a.B; row += gridDim.x * 4) {
    float v[NCH][4];
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
      const int c = 4 * (lane + 64 * j);
      const vu4 p0 = ld16(pr, ((0 * a.B + row) * D + c) * 4), p1 = ld16(pr, ((1 * a.B + row) * D + c) * 4), p2 = ld16(pr, ((2 * a.B + row) * D + c) * 4),
                p3 = ld16(pr, ((3 * a.B + row) * D + c) * 4);
      const float4 b4 = *reinterpret_cast<const float4*>(bias + c);
      float r4[4];
      ld_bf16x4(res + tiled(row, c, D), r4);
      const float bb[4] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
      for (int e = 0; e < 4; ++e)
        v[j][e] = ((__uint_as_float(p0[e]) + __uint_as_float(p1[e])) + (__uint_as_float(p2[e]) + __uint_as_float(p3[e]))) + bb[e] + r4[e];
    }
    ln_row(v, gamma, beta, a.eps, lane);
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
      const int c = 4 * (lane + 64 * j);
      st_bf16x4_sc1(orr, tiled(row, c, D), v[j]);
      if (out2) st_bf16x4(out2 + (long long)row * ld2 + c, v[j]);
    }
  }
}

// attention of decoder row t for (sample b, head h): one wave.  Lane (kg = lane >> 3, dc = lane & 7) takes the 16-byte chunk dc of keys kg, kg + 8, ...:
// every load instruction covers eight whole 128-byte head rows; a key's score is completed across its eight lanes, the probabilities stay in the
// lanes that load the matching value chunks, and the eight key groups are added at the end.  NI = ceil(N / 8) iterations, all loads in flight at once.
template <int NI>
__device__ __forceinline__ void attn_task(const DArgs& a, const DLayer& L, int li, int b, int h, int t) {
  const int lane = threadIdx.x & 63, kg = lane >> 3, dc = lane & 7;
  const int qc = a.n_enc + t, nk = qc + 1;
  const bf16_t* base = L.qkv + (long long)b * a.N * (3 * D) + h * HD + dc * 8;
  const uint4 qu = *reinterpret_cast<const uint4*>(base + (long long)qc * (3 * D));
  const float q8[8] = {bf_lo(qu.x), bf_hi(qu.x), bf_lo(qu.y), bf_hi(qu.y), bf_lo(qu.z), bf_hi(qu.z), bf_lo(qu.w), bf_hi(qu.w)};
  const uint32_t* ap = L.allow + b * L.allow_sb + h * L.allow_sh + (long long)qc * a.NW;
  uint4 kf[NI], vf[NI];
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const bf16_t* rp = base + (long long)min(kg + 8 * i, nk - 1) * (3 * D);
    kf[i] = *reinterpret_cast<const uint4*>(rp + D);
    vf[i] = *reinterpret_cast<const uint4*>(rp + 2 * D);
  }
  __builtin_amdgcn_sched_barrier(0);
  float s[NI];
  float mx = -INFINITY;
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int j = kg + 8 * i;
    const uint4 u = kf[i];
    float d = q8[0] * bf_lo(u.x);
    d = fmaf(q8[1], bf_hi(u.x), d); d = fmaf(q8[2], bf_lo(u.y), d); d = fmaf(q8[3], bf_hi(u.y), d);
    d = fmaf(q8[4], bf_lo(u.z), d); d = fmaf(q8[5], bf_hi(u.z), d); d = fmaf(q8[6], bf_lo(u.w), d); d = fmaf(q8[7], bf_hi(u.w), d);
    d += __shfl_xor(d, 1); d += __shfl_xor(d, 2); d += __shfl_xor(d, 4);
    const bool valid = j < nk && ((ap[min(j, a.N - 1) >> 5] >> (j & 31)) & 1u);
    s[i] = valid ? d * a.scale_log2 : -INFINITY;
    mx = fmaxf(mx, s[i]);
  }
  mx = fmaxf(mx, __shfl_xor(mx, 8)); mx = fmaxf(mx, __shfl_xor(mx, 16)); mx = fmaxf(mx, __shfl_xor(mx, 32));
  float l = 0.f, acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const float p = s[i] == -INFINITY ? 0.f : __builtin_amdgcn_exp2f(s[i] - mx);
    l += p;
    const uint4 u = vf[i];
    acc[0] = fmaf(p, bf_lo(u.x), acc[0]); acc[1] = fmaf(p, bf_hi(u.x), acc[1]); acc[2] = fmaf(p, bf_lo(u.y), acc[2]); acc[3] = fmaf(p, bf_hi(u.y), acc[3]);
    acc[4] = fmaf(p, bf_lo(u.z), acc[4]); acc[5] = fmaf(p, bf_hi(u.z), acc[5]); acc[6] = fmaf(p, bf_lo(u.w), acc[6]); acc[7] = fmaf(p, bf_hi(u.w), acc[7]);
  }
#pragma unroll
  for (int o = 8; o < 64; o <<= 1) {
    l += __shfl_xor(l, o);
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] += __shfl_xor(acc[e], o);
  }
  if (kg == 0) {
    const float inv = l > 0.f ? 1.0f / l : 0.f;
    const vu4 o4 = {pack_bf16x2(acc[0] * inv, acc[1] * inv), pack_bf16x2(acc[2] * inv, acc[3] * inv), pack_bf16x2(acc[4] * inv, acc[5] * inv), pack_bf16x2(acc[6] * inv, acc[7] * inv)};
    st16(o4, rsrc(ctxbuf(a, t, li)), (int)(tiled(b, h * HD + dc * 8, D) * 2));
  }
}

// pointer scores, argmax, the next token and the next step's input row of sample b: one block
__device__ __forceinline__ void pick_task(const DArgs& a, int b, int t, float* lds, Best* red) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float* qs = lds;            // [D]
  float* dyn = lds + D;       // [No <= 64]
  const rsrc_t pr = rsrc(a.part);
  for (int d = tid; d < D; d += NT) {
    const float p = (__uint_as_float(ld4(pr, ((0 * a.B + b) * D + d) * 4)) + __uint_as_float(ld4(pr, ((1 * a.B + b) * D + d) * 4))) +
                    (__uint_as_float(ld4(pr, ((2 * a.B + b) * D + d) * 4)) + __uint_as_float(ld4(pr, ((3 * a.B + b) * D + d) * 4)));
    qs[d] = bf2f(f2bf(p + a.bq[d]));
  }
  __syncthreads();
  for (int o = wave; o < a.No; o += NT / 64) {
    const bf16_t* kp = a.ptr_k + ((long long)b * a.No + o) * D;
    float sum = 0.f;
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
      const int c = 4 * (lane + 64 * j);
      float k4[4];
      ld_bf16x4(kp + c, k4);
      sum += (qs[c] * k4[0] + qs[c + 1] * k4[1]) + (qs[c + 2] * k4[2] + qs[c + 3] * k4[3]);
    }
    sum = wave_sum(sum);
    if (lane == 0) {
      const float sc = sum * a.ptr_scale + (a.ocr_mask[(long long)b * a.No + o] ? 0.f : -10000.0f);
      dyn[o] = sc;
      a.dyn_all[((long long)b * a.S + t) * a.No + o] = sc;
    }
  }
  __syncthreads();
  Best x = {-INFINITY, 0x7fffffff};
  // (sc1 loads: logits rows are not cache-line aligned -- 20 000 bytes at V = 5000 --, the line row t shares with row t + 1 would be stale next step)
  const rsrc_t fr = rsrc(a.fixed_all + ((long long)b * a.S + t) * a.ldf);
  for (int j = tid; j < a.V + a.No; j += NT) {
    const float v = j < a.V ? __uint_as_float(ld4(fr, j * 4)) : dyn[j - a.V];
    x = better(x, Best{v, j});
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    Best y;
    y.v = __shfl_xor(x.v, o);
    y.i = __shfl_xor(x.i, o);
    x = better(x, y);
  }
  if (lane == 0) red[wave] = x;
  __syncthreads();
  Best r = red[0];
#pragma unroll
  for (int w = 1; w < NT / 64; ++w) r = better(r, red[w]);
  const long long tok = r.i == 0x7fffffff ? 0 : r.i;
  if (t + 1 < a.S) {
    if (tid == 0) a.prev[(long long)b * a.S + t + 1] = tok;
    if (wave == 0) embed_row(a, b, t + 1, tok);
  }
  __syncthreads();
}

template <int NI>
__global__ __launch_bounds__(NT, 1) void decode_steps_kernel(DArgs a) {
  __shared__ __attribute__((aligned(16))) float lds[D + 64];
  __shared__ Best red[NT / 64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4;
  (void)g;
  unsigned epoch = 0;
  const int nblk = gridDim.x;
  for (int b = blockIdx.x * 4 + wave; b < a.B; b += nblk * 4) embed_row(a, b, a.t_begin, a.prev[(long long)b * a.S + a.t_begin]);
  if (!grid_sync(a, epoch)) return;
  for (int t = a.t_begin; t < a.t_end; ++t) {
    for (int li = 0; li < a.n_layers; ++li) {
      const DLayer& L = a.L[li];
      const bool last = li + 1 == a.n_layers;
      // Q: q|k|v row of every sample into the cache
      {
        const rsrc_t dst = rsrc(L.qkv + (long long)(a.n_enc + t) * (3 * D));
        const long long ldq = (long long)a.N * (3 * D);
        const float* bias = L.bqkv;
        DS_STAMP(0);
        for (int task = blockIdx.x, nt = gemm_tasks(a, 3 * D, 1); task < nt; task += nblk)
          gemm_task<D / 32>(a, xbuf(a, t, li), D, L.wqkv, D, 3 * D, 1, task, [&](int, int m, int n, const f32x4& acc) {
            const float4 b4 = *reinterpret_cast<const float4*>(bias + n);
            const float v[4] = {acc[0] + b4.x, acc[1] + b4.y, acc[2] + b4.z, acc[3] + b4.w};
            st_bf16x4_sc1(dst, m * ldq + n, v);
          });
        DS_STAMP(1);
        if (a.prof && blockIdx.x == 0 && threadIdx.x == 0 && t == a.t_begin + 1 && li == 1) { a.prof[1002] = a.prof[1016]; a.prof[1003] = a.prof[1017]; }
      }
      if (!grid_sync(a, epoch)) return;
      if (a.prof && blockIdx.x == 0 && threadIdx.x == 0 && t == a.t_begin + 1 && li == 1) { a.prof[1004] = a.prof[1016]; a.prof[1005] = a.prof[1017]; a.prof[1006] = wall_clock64(); }
      // A
      for (int task = blockIdx.x * 4 + wave; task < a.B * a.H; task += nblk * 4) attn_task<NI>(a, L, li, task / a.H, task % a.H, t);
      if (!grid_sync(a, epoch)) return;
      // O: split-K partials
      for (int task = blockIdx.x, nt = gemm_tasks(a, D, KSPLIT); task < nt; task += nblk)
        gemm_task<D / 32 / KSPLIT>(a, ctxbuf(a, t, li), D, L.wo, D, D, KSPLIT, task, [&](int ks, int m, int n, const f32x4& acc) {
          st16(f4_bits(acc), rsrc(a.part), ((ks * a.B + m) * D + n) * 4);
        });
      if (!grid_sync(a, epoch)) return;
      finalize_rows(a, L.bo, xbuf(a, t, li), L.g1, L.be1, x1buf(a, t, li), nullptr, 0);
      if (!grid_sync(a, epoch)) return;
      // G: FFN1 + erf-GELU
      {
        const float* bias = L.b1;
        for (int task = blockIdx.x, nt = gemm_tasks(a, F, 1); task < nt; task += nblk)
          gemm_task<D / 32>(a, x1buf(a, t, li), D, L.w1, D, F, 1, task, [&](int, int m, int n, const f32x4& acc) {
            const float4 b4 = *reinterpret_cast<const float4*>(bias + n);
            const float v[4] = {gelu_erf(acc[0] + b4.x), gelu_erf(acc[1] + b4.y), gelu_erf(acc[2] + b4.z), gelu_erf(acc[3] + b4.w)};
            st_bf16x4_sc1(rsrc(hbuf(a, t, li)), tiled(m, n, F), v);
          });
      }
      if (!grid_sync(a, epoch)) return;
      // H: FFN2 split-K partials
      for (int task = blockIdx.x, nt = gemm_tasks(a, D, KSPLIT); task < nt; task += nblk)
        gemm_task<F / 32 / KSPLIT>(a, hbuf(a, t, li), F, L.w2, F, D, KSPLIT, task, [&](int ks, int m, int n, const f32x4& acc) {
          st16(f4_bits(acc), rsrc(a.part), ((ks * a.B + m) * D + n) * 4);
        });
      if (!grid_sync(a, epoch)) return;
      finalize_rows(a, L.b2, x1buf(a, t, li), L.g2, L.be2, xbuf(a, t, li + 1), last && a.seq ? a.seq + (long long)(a.n_enc + t) * D : nullptr, (long long)a.N * D);
      if (!grid_sync(a, epoch)) return;
    }
    // C: classifier logits of row t | pointer-network query partials
    {
      const int t1 = gemm_tasks(a, a.V, 1), t2 = gemm_tasks(a, D, KSPLIT);
      const rsrc_t frow = rsrc(a.fixed_all + (long long)t * a.ldf);
      const long long ldfs = (long long)a.S * a.ldf;
      for (int task = blockIdx.x; task < t1 + t2; task += nblk) {
        if (task < t1)
          gemm_task<D / 32>(a, xbuf(a, t, a.n_layers), D, a.wc, D, a.V, 1, task, [&](int, int m, int n, const f32x4& acc) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
              if (n + r < a.V) st4(__float_as_uint(acc[r] + a.bc[n + r]), frow, (int)((m * ldfs + n + r) * 4));
          });
        else
          gemm_task<D / 32 / KSPLIT>(a, xbuf(a, t, a.n_layers), D, a.wq, D, D, KSPLIT, task - t1, [&](int ks, int m, int n, const f32x4& acc) {
            st16(f4_bits(acc), rsrc(a.part), ((ks * a.B + m) * D + n) * 4);
          });
      }
    }
    if (!grid_sync(a, epoch)) return;
    for (int b = blockIdx.x; b < a.B; b += nblk) pick_task(a, b, t, lds, red);
    if (!grid_sync(a, epoch)) return;
  }
  // the last block to leave puts the barrier back to zero for the next launch (every block is past the final barrier by then)
  if (threadIdx.x == 0) {
    const unsigned left = __hip_atomic_fetch_add(a.bar + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (left == gridDim.x - 1) {
      __hip_atomic_store(a.bar, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(a.bar + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

}  // namespace

constexpr int64_t WS_HEADER = 256 + 8192;        // barrier words, error word, 1024 profile stamps
extern "C" int64_t sam_greedy_decode_ws_bytes(int B, int S, int n_layers) {
  if (B <= 0 || S <= 0 || n_layers <= 0) return 0;
  const int64_t b = (B + 15) / 16 * 16, s = S, l = n_layers;
  return WS_HEADER + (int64_t)KSPLIT * b * D * (int64_t)sizeof(float) + s * ((l + 1) * b * D + l * (2 * b * D + b * F)) * (int64_t)sizeof(bf16_t);
}

extern "C" int sam_greedy_decode_steps(const sam_decode_desc* d, void* ws, int64_t ws_bytes, void* stream) {
  SAM_REQUIRE(d && ws && d->layers, "sam_greedy_decode_steps: null pointer");
  if (d->D != D || d->F != F || d->H * HD != D || d->n_layers < 1 || d->n_layers > MAXL || d->N > 256 || d->No > 64) {
    sam_set_error("sam_greedy_decode_steps: built for D=768, F=3072, head_dim=64, <= %d layers, N <= 256, <= 64 OCR slots (got D=%d F=%d H=%d L=%d N=%d No=%d)", MAXL, d->D, d->F,
                  d->H, d->n_layers, d->N, d->No);
    return SAM_ERR_UNSUPPORTED;
  }
  SAM_REQUIRE(d->B > 0 && d->S >= 1 && d->n_enc >= 0 && d->n_enc + d->S == d->N && d->V > 0 && d->No >= 1, "sam_greedy_decode_steps: bad shape");
  SAM_REQUIRE(d->t_begin >= 1 && d->t_begin <= d->t_end && d->t_end <= d->S, "sam_greedy_decode_steps: need 1 <= t_begin <= t_end <= S");
  SAM_REQUIRE(ws_bytes >= sam_greedy_decode_ws_bytes(d->B, d->S, d->n_layers) && ((uintptr_t)ws % 256 == 0), "sam_greedy_decode_steps: workspace too small or not 256-byte aligned");
  SAM_REQUIRE(d->pos_emb && d->type_emb && d->emb_ln_g && d->emb_ln_b && d->ans_ln && d->ocr_ln && d->wc && d->bc && d->wq && d->bq && d->ptr_k && d->ocr_mask && d->prev_inds &&
                  d->fixed_scores && d->ocr_scores,
              "sam_greedy_decode_steps: null pointer in the descriptor");
  SAM_REQUIRE(d->ld_fixed >= d->V && d->ld_pos % 4 == 0 && d->ld_type % 4 == 0, "sam_greedy_decode_steps: bad leading dimension");
  if (d->t_begin == d->t_end) return SAM_OK;
  DArgs a = {};
  for (int l = 0; l < d->n_layers; ++l) {
    const sam_decode_layer& s = d->layers[l];
    SAM_REQUIRE(s.wqkv && s.wo && s.w1 && s.w2 && s.bqkv && s.bo && s.b1 && s.b2 && s.ln1_g && s.ln1_b && s.ln2_g && s.ln2_b && s.qkv && s.allow,
                "sam_greedy_decode_steps: null pointer in layer %d", l);
    DLayer& L = a.L[l];
    L.wqkv = (const bf16_t*)s.wqkv; L.wo = (const bf16_t*)s.wo; L.w1 = (const bf16_t*)s.w1; L.w2 = (const bf16_t*)s.w2;
    L.bqkv = s.bqkv; L.bo = s.bo; L.b1 = s.b1; L.b2 = s.b2; L.g1 = s.ln1_g; L.be1 = s.ln1_b; L.g2 = s.ln2_g; L.be2 = s.ln2_b;
    L.qkv = (bf16_t*)s.qkv; L.allow = s.allow; L.allow_sb = s.allow_stride_b; L.allow_sh = s.allow_stride_h;
  }
  a.n_layers = d->n_layers; a.B = d->B; a.Bp = (d->B + 15) / 16 * 16; a.N = d->N; a.n_enc = d->n_enc; a.S = d->S; a.H = d->H; a.NW = (d->N + 31) / 32; a.V = d->V; a.No = d->No;
  a.t_begin = d->t_begin; a.t_end = d->t_end;
  a.ldf = d->ld_fixed; a.ldwc = D; a.ld_pos = d->ld_pos; a.ld_type = d->ld_type;
  a.scale_log2 = d->scale * 1.44269504088896341f; a.eps = d->ln_eps; a.eps_emb = d->emb_ln_eps; a.ptr_scale = d->ptr_scale;
  a.pos_emb = d->pos_emb; a.type_emb = d->type_emb; a.emb_g = d->emb_ln_g; a.emb_b = d->emb_ln_b;
  a.ans_ln = (const bf16_t*)d->ans_ln; a.ocr_ln = (const bf16_t*)d->ocr_ln; a.wc = (const bf16_t*)d->wc; a.wq = (const bf16_t*)d->wq; a.ptr_k = (const bf16_t*)d->ptr_k;
  a.bc = d->bc; a.bq = d->bq; a.ocr_mask = d->ocr_mask; a.prev = (long long*)d->prev_inds; a.fixed_all = d->fixed_scores; a.dyn_all = d->ocr_scores;
  a.seq = (bf16_t*)d->seq_out;
  char* w = (char*)ws;
  a.bar = (unsigned*)w; a.err = (int*)(w + 128);
  { static int prof = -1; if (prof < 0) { const char* e = getenv("SAM_DECODE_PROF"); prof = e ? atoi(e) : 0; } a.prof = prof ? (long long*)(w + 256) : nullptr; }
  w += WS_HEADER;
  const int64_t b = a.Bp, sl = (int64_t)d->S * d->n_layers;
  a.part = (float*)w; w += (int64_t)KSPLIT * b * D * sizeof(float);
  a.x = (bf16_t*)w; w += (int64_t)d->S * (d->n_layers + 1) * b * D * sizeof(bf16_t);
  a.x1 = (bf16_t*)w; w += sl * b * D * sizeof(bf16_t);
  a.ctx = (bf16_t*)w; w += sl * b * D * sizeof(bf16_t);
  a.h = (bf16_t*)w;
  hipStream_t st = (hipStream_t)stream;
  // (no memset node in front of the kernel: the launch leaves the barrier words at zero itself.  A hipMemsetAsync captured ahead of the kernel
  // left the counter non-zero on graph replays -- barriers fell through, tokens came out wrong)
  const int grid = samgemm::device_cu_count();            // one block per CU: every block is resident, which the grid barrier relies on
  if (d->N <= 192) decode_steps_kernel<24><<<dim3(grid), dim3(NT), 0, st>>>(a);
  else decode_steps_kernel<32><<<dim3(grid), dim3(NT), 0, st>>>(a);
  SAM_LAUNCH_CHECK();
  return SAM_OK;
}
