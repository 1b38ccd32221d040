// Greedy decoding steps 1 .. S-1 of SA-M4C (sam/sa_m4c.py:285-302) as ONE persistent launch (gfx950).
//
// What a step is.  Under the prefix-LM mask (sa_m4c.py:834-844) decoder row t sees the encoder rows and decoder rows 0..t only, and its input
// token is fixed once step t-1 has picked it: row t of the LAST of the reference's twelve full forwards equals row t computed at step t.  So a
// step runs ONE new row per sample (B rows, 64 at the bench batch) through the embedding of the previous prediction (sa_m4c.py:928-948), the
// encoder layers (sa_m4c.py:660-684,718-760) against the cached keys / values of the first full pass, the classifier + pointer network
// (sa_m4c.py:270-278, 866-897) and the argmax (sa_m4c.py:299-302) -- about 0.2 GFLOP, 85 MB of weights and 36 MB of cache per layer and step.
// As separate launches that is ~56 kernels of 5-10 us each (measured 595 us per captured step).  Here every stage is a PHASE of one kernel and
// all S-1 steps run inside the launch.
//
// Partition.  Samples do not interact, so the batch is split over the eight XCDs: XCD g (the blocks that find g in HW_REG_XCC_ID; 32 of the 256)
// decodes samples [g spx, (g+1) spx), spx = ceil(B / 8) <= 16 rows = one MFMA row tile, through all phases of all steps on its own.
//   * Barriers are then among the 32 blocks of ONE XCD, on a counter that lives in that XCD's L2 (atomics without the agent-scope bit): 0.7 us,
//     against 3.7 us for an agent-scope barrier over all 256 blocks (tools/probes/probe_xcd.hip); with 44 phases per step the first version of
//     this kernel -- phases over the whole batch, global barriers -- spent more than half of its 470 us per step waiting at them.
//   * Hand-over buffers (activations, the new cache row, logits) are written and read inside one XCD, whose L2 is coherent for its CUs: plain
//     stores (write-through L1) + s_waitcnt vmcnt(0) before the barrier, plain loads after it.  The only hazard left is a STALE L1 line, so the
//     activations live in an arena indexed by (step, layer) -- every address is written once per launch and never read before that -- and the
//     two reused / unaligned buffers (K-split partials, logits rows: 20 000 bytes, not a multiple of the line) are read with L1-bypassing loads.
//     (Global version: fences cost ~30 us per phase, agent-scope loads 13 us per GEMM phase, see the git history of this file.)
//   * The price: every XCD streams ALL the weights, 8 x 85 MB per step.  Measured: eight XCDs reading the same 4.7 MB take 6.6 us (715 GB/s per
//     XCD, 5.7 TB/s aggregate) -- ~130 us per step, which is now the floor, against >160 us of pure barrier time before.
// Phases of a step (7 per layer + 2):   per layer  Q  x -> q|k|v row, written straight into the layer's [B, N, 3D] cache at row n_enc + t
//                                                  A  one wave per (sample, head): scores over cache rows 0 .. n_enc + t under the allow bits, softmax, PV
//                                                  O  ctx Wo^T, split-K 4 -> fp32 partials          F1  partials + bias + x -> LayerNorm -> x1
//                                                  G  gelu(x1 W1^T + b1) -> h                        H   h W2^T, split-K 4 -> partials
//                                                  F2 partials + bias + x1 -> LayerNorm -> x (and the row of the final hidden states after the last layer)
//                              then                C  classifier logits (fp32, into row t of the score block) | pointer-network query, split-K 4
//                                                  P  one block per sample: query . OCR keys (+ -10000 on padded OCR slots), argmax over [logits | pointer
//                                                     scores] (first index wins ties), prev_inds[b, t + 1], and the NEXT step's input row x[b]
// The skinny GEMMs (M = spx <= 16 rows): a WAVE takes one 16-column slice of the output (and one K split) and issues all of its operand loads
// (24 k-steps x 2 x 16 bytes per lane) before the first MFMA.  Both operands are in the FRAGMENT-TILED layout [rows / 16][K / 8][16][8]: the MFMA
// fragment has lane l read row (l & 15), which from row-major operands is 64 different 16-byte pieces per load instruction, one tag lookup each
// (measured 6.5 us per task, warm or cold; staging through LDS cost more in registers and barriers than it saved); tiled, a load instruction of
// the wave covers 1 KB contiguous.  The weights are re-tiled once per batch by the host side, the activations are written tiled by the epilogues.
#include "common.h"
#include "gemm_common.h"
#include "sam_hip.h"

extern "C" int sam_attn_words_per_row(int N);

namespace {

constexpr int NT = 256, NW = NT / 64, MAXL = 12, MAXNO = 128, D = 768, F = 3072, HD = 64, KSPLIT = 4, NCH = D / 256, NXCD = 8;
constexpr unsigned SPIN_LIMIT = 1u << 22;

struct Best { float v; int i; };
__device__ __forceinline__ Best better(Best a, Best b) { return (b.v > a.v || (b.v == a.v && b.i < a.i)) ? b : a; }

// L1-bypassing (sc0) loads for the two buffers whose lines an earlier phase may have left in this CU's L1
constexpr int AUX_SC0 = 0x1;
typedef unsigned int vu4 __attribute__((ext_vector_type(4)));
typedef __amdgpu_buffer_rsrc_t rsrc_t;
__device__ __forceinline__ rsrc_t rsrc(const void* p) { return __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, 0x7fffffff, 0x00020000); }
__device__ __forceinline__ vu4 ld16_l2(rsrc_t r, int off) { return __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, AUX_SC0); }
__device__ __forceinline__ unsigned ld4_l2(rsrc_t r, int off) { return __builtin_amdgcn_raw_buffer_load_b32(r, off, 0, AUX_SC0); }

struct DLayer {
  const bf16_t *wqkv, *wo, *w1, *w2;
  const float *bqkv, *bo, *b1, *b2, *g1, *be1, *g2, *be2;
  bf16_t* qkv;                      // [B, N, 3D]: the first pass's q|k|v rows; rows n_enc + t are written here
  const uint32_t* allow;
  long long allow_sb, allow_sh;
};
struct DArgs {
  DLayer L[MAXL];
  int n_layers, B, spx, N, n_enc, S, H, NWORDS, V, No, t_begin, t_end;
  long long ldf, ld_pos, ld_type;
  float scale_log2, eps, eps_emb, ptr_scale;
  const float *pos_emb, *type_emb, *emb_g, *emb_b;
  const bf16_t *ans_ln, *ocr_ln, *wc, *wq, *ptr_k;
  const float *bc, *bq;
  const unsigned char* ocr_mask;
  long long* prev;
  float *fixed_all, *dyn_all;
  bf16_t* seq;
  bf16_t *x, *x1, *ctx, *h;         // per-XCD arenas of 16-row tiled slots: x [8][S][L+1][16, D], x1 / ctx [8][S][L][16, D], h [8][S][L][16, F]
  float* part;                      // [8][KSPLIT][16, D] row-major, reused
  unsigned* bar;                    // [8] x 32 words: word 0 = barrier counter of the XCD, word 1 = blocks that left, word 2 = blocks that arrived (ranks)
  int* err;
  long long* prof;                  // SAM_DECODE_PROF=1: block 0 stamps wall_clock64() (100 MHz) after every barrier
};

// this XCD's slot for (step, layer)
struct Grp { int g, r, b0, nloc; };         // XCD, rank of the block inside it, first sample, number of samples
__device__ __forceinline__ bf16_t* xbuf(const DArgs& a, const Grp& G, int t, int li) { return a.x + (((long long)G.g * a.S + t) * (a.n_layers + 1) + li) * 16 * D; }
__device__ __forceinline__ bf16_t* x1buf(const DArgs& a, const Grp& G, int t, int li) { return a.x1 + (((long long)G.g * a.S + t) * a.n_layers + li) * 16 * D; }
__device__ __forceinline__ bf16_t* ctxbuf(const DArgs& a, const Grp& G, int t, int li) { return a.ctx + (((long long)G.g * a.S + t) * a.n_layers + li) * 16 * D; }
__device__ __forceinline__ bf16_t* hbuf(const DArgs& a, const Grp& G, int t, int li) { return a.h + (((long long)G.g * a.S + t) * a.n_layers + li) * 16 * F; }
__device__ __forceinline__ float* partbuf(const DArgs& a, const Grp& G) { return a.part + (long long)G.g * KSPLIT * 16 * D; }
// element (row, k) of a one-tile [K / 8][16][8] slot
__device__ __forceinline__ int tiled(int row, int k) { return (((k >> 3) << 4) + row) * 8 + (k & 7); }

__device__ __forceinline__ unsigned l2_read(unsigned* p) {
  unsigned v;
  const unsigned zero = 0u;
  asm volatile("global_atomic_add %0, %1, %2, off sc0\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p), "v"(zero) : "memory");
  return v;
}

// every block of the XCD arrives, then waits for the XCD's counter to reach `epoch` (zero when a launch begins: the last block to leave resets
// it).  The atomics carry no agent-scope bit: they execute in this XCD's L2, which all of its blocks share.  false = a block failed to arrive
// within the spin limit (or another one gave up): the caller returns, the error word stays set.
__device__ __forceinline__ bool xcd_sync(const DArgs& a, const Grp& G, unsigned nblk, unsigned& epoch) {
  __shared__ int ok;
  unsigned* bar = a.bar + G.g * 32;
  epoch += nblk;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // this wave's stores have reached the L2
  __syncthreads();
  if (threadIdx.x == 0) {
    __hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    unsigned spins = 0;
    int good = 1;
    // polled with a returning read-modify-write, which always executes in the L2: a workgroup-scope LOAD may be served by this CU's L1 (a block
    // that started polling before the last arrival then never sees it), and the compiler turns `fetch_add(p, 0)` into exactly that load
    while (l2_read(bar) < epoch) {
      __builtin_amdgcn_s_sleep(1);
      if ((++spins & 1023u) == 0 && (spins > SPIN_LIMIT || __hip_atomic_load(a.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) {
        int expected = 0;        // keep an earlier, more specific code (2 = uneven XCD deal): only an unset word becomes "barrier timed out"
        __hip_atomic_compare_exchange_strong(a.err, &expected, 1, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        good = 0;
        break;
      }
    }
    ok = good;
  }
  __syncthreads();
  asm volatile("" ::: "memory");
  if (a.prof && blockIdx.x == 0 && threadIdx.x == 0) {
    const unsigned k = epoch / nblk;
    if (k < 1000) a.prof[k] = wall_clock64();
  }
  return ok != 0;
}

// one wave: the 16 (n) x 16 (m) tile sum_k W[n][k] X[m][k] over KS k-steps of 32 starting at k-group k8_0 (both operands fragment-tiled; X is a
// one-tile slot); lane (i = l & 15, g = l >> 4) ends up with n = nt 16 + 4g + r (r = 0..3) of local row m = i
template <int KS>
__device__ __forceinline__ void load_w(bf16x8 (&wf)[KS], const bf16_t* W, int Kw, int nt, int k8_0) {
  const int lane = threadIdx.x & 63, i = lane & 15, g = lane >> 4;
  const bf16_t* wp = W + ((((long long)nt * (Kw >> 3) + k8_0 + g) << 4) + i) * 8;
#pragma unroll
  for (int q = 0; q < KS; ++q) wf[q] = *reinterpret_cast<const bf16x8*>(wp + q * 512);          // 4 k-groups x 16 rows x 8 elements per k-step
}
// the activation half of the tile: loads X (all k-steps in flight), then the MFMAs.  One wave per SIMD (256 threads per block): a wave may hold
// all 2 x KS fragments (192 registers at KS = 24).  (Eight waves per CU with a 14-deep ring of fragments spilled ~140 registers in this kernel.)
template <int KS>
__device__ __forceinline__ f32x4 tile_w(const bf16x8 (&wf)[KS], const bf16_t* X, int k8_0) {
  const int lane = threadIdx.x & 63, i = lane & 15, g = lane >> 4;
  const bf16_t* xp = X + (((k8_0 + g) << 4) + i) * 8;
  bf16x8 xf[KS];
#pragma unroll
  for (int q = 0; q < KS; ++q) xf[q] = *reinterpret_cast<const bf16x8*>(xp + q * 512);
  __builtin_amdgcn_sched_barrier(0);      // every load issued before the first MFMA waits (left alone the scheduler kept ~12 in flight)
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int q = 0; q < KS; ++q) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[q], xf[q], acc, 0, 0, 0);
  return acc;
}
template <int KS>
__device__ __forceinline__ f32x4 wave_tile(const bf16_t* W, int Kw, int nt, const bf16_t* X, int k8_0) {
  bf16x8 wf[KS];
  load_w<KS>(wf, W, Kw, nt, k8_0);
  return tile_w<KS>(wf, X, k8_0);
}

__device__ __forceinline__ void st_bf16x4(bf16_t* p, const float* v) {
  uint2 o;
  o.x = pack_bf16x2(v[0], v[1]); o.y = pack_bf16x2(v[2], v[3]);
  *reinterpret_cast<uint2*>(p) = o;
}
__device__ __forceinline__ void ld_bf16x4(const bf16_t* p, float* v) {
  const uint2 u = *reinterpret_cast<const uint2*>(p);
  v[0] = bf_lo(u.x); v[1] = bf_hi(u.x); v[2] = bf_lo(u.y); v[3] = bf_hi(u.y);
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

// x[row] = (tok < V ? LN(answer table)[tok] : LN(OCR rows)[b, tok - V]) + bf16(LN(position[t] + type[tok >= V]))   (sa_m4c.py:928-948, eval: no dropout); one wave
__device__ __forceinline__ void embed_row(const DArgs& a, const Grp& G, int row, int t, long long tok) {
  const int lane = threadIdx.x & 63, b = G.b0 + row;
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
  bf16_t* dst = xbuf(a, G, t, 0);
#pragma unroll
  for (int j = 0; j < NCH; ++j) {
    const int c = 4 * (lane + 64 * j);
    float s4[4], o[4];
    ld_bf16x4(src + c, s4);
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = s4[e] + bf2f(f2bf(v[j][e]));
    st_bf16x4(dst + tiled(row, c), o);
  }
}

// out[row] = LayerNorm(sum of the K-split partials (fixed order) + bias + res[row]); one wave per local row
__device__ __forceinline__ void finalize_row(const DArgs& a, const Grp& G, int row, const float* bias, const bf16_t* res, const float* gamma, const float* beta, bf16_t* out,
                                             bf16_t* out2) {
  const int lane = threadIdx.x & 63;
  const rsrc_t pr = rsrc(partbuf(a, G));
  float v[NCH][4];
#pragma unroll
  for (int j = 0; j < NCH; ++j) {
    const int c = 4 * (lane + 64 * j);
    const vu4 p0 = ld16_l2(pr, ((0 * 16 + row) * D + c) * 4), p1 = ld16_l2(pr, ((1 * 16 + row) * D + c) * 4), p2 = ld16_l2(pr, ((2 * 16 + row) * D + c) * 4),
              p3 = ld16_l2(pr, ((3 * 16 + row) * D + c) * 4);
    const float4 b4 = *reinterpret_cast<const float4*>(bias + c);
    float r4[4];
    ld_bf16x4(res + tiled(row, c), r4);
    const float bb[4] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) v[j][e] = ((__uint_as_float(p0[e]) + __uint_as_float(p1[e])) + (__uint_as_float(p2[e]) + __uint_as_float(p3[e]))) + bb[e] + r4[e];
  }
  ln_row(v, gamma, beta, a.eps, lane);
#pragma unroll
  for (int j = 0; j < NCH; ++j) {
    const int c = 4 * (lane + 64 * j);
    st_bf16x4(out + tiled(row, c), v[j]);
    if (out2) st_bf16x4(out2 + c, v[j]);
  }
}

// attention of decoder row t for (local row, head h): one wave.  Lane (kg = lane >> 3, dc = lane & 7) takes the 16-byte chunk dc of keys kg, kg + 8, ...:
// every load instruction covers eight whole 128-byte head rows; a key's score is completed across its eight lanes, the probabilities stay in the
// lanes that load the matching value chunks, and the eight key groups are added at the end.  NI = ceil(N / 8) iterations.
// cross-lane adds inside a row of 16 lanes by DPP (one VALU cycle each) instead of ds_bpermute: quad_perm [1,0,3,2] / [2,3,0,1], row_half_mirror (lane i <-> 7 - i
// inside 8 lanes: the quads already hold equal sums), row_ror:8 (lane i <-> i ^ 8)
#define SAM_DPP(v, ctrl) __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), (ctrl), 0xf, 0xf, false))
__device__ __forceinline__ float sum8(float d) {            // sum over the 8 lanes (l & ~7) .. (l | 7), result in all of them
  d += SAM_DPP(d, 0xB1);
  d += SAM_DPP(d, 0x4E);
  d += SAM_DPP(d, 0x141);
  return d;
}

// keys / values of every cache row before the current one, all loads in flight at once
template <int NI>
struct AttnCur { uint4 q, k, v; uint32_t aw[NI / 4]; };      // the current row's q / k / v chunk of this lane and the row's allow words (32 keys each)
template <int NI>
__device__ __forceinline__ void attn_load(uint4 (&kf)[NI], uint4 (&vf)[NI], AttnCur<NI>& c, const DArgs& a, const DLayer& L, const Grp& G, int li, int row, int h, int t) {
  const int lane = threadIdx.x & 63, b = G.b0 + row, kg = lane >> 3, dc = lane & 7;
  const int qc = a.n_enc + t;
  const bf16_t* base = L.qkv + (long long)b * a.N * (3 * D) + h * HD + dc * 8;
  {   // requested FIRST: the score loop needs them before anything else, and loads return in order
    const bf16_t* cur = base + (long long)qc * (3 * D);                  // the row phase Q has just written
    c.q = *reinterpret_cast<const uint4*>(cur);
    c.k = *reinterpret_cast<const uint4*>(cur + D);
    c.v = *reinterpret_cast<const uint4*>(cur + 2 * D);
    const uint32_t* ap = L.allow + b * L.allow_sb + h * L.allow_sh + (long long)qc * a.NWORDS;
#pragma unroll
    for (int w = 0; w < NI / 4; ++w) c.aw[w] = ap[min(w, a.NWORDS - 1)];
  }
  // (a head-major copy of the cache -- [B, H, N, 64], one (sample, head)'s rows a contiguous 23 KB stream instead of 128-byte pieces 4.6 KB apart --
  // was built and measured: the keys / values arrive after 6.4 us either way, the XCD's share of the fabric, and the copy cost 0.22 ms per batch)
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const bf16_t* rp = base + (long long)min(kg + 8 * i, max(qc - 1, 0)) * (3 * D);
    kf[i] = *reinterpret_cast<const uint4*>(rp + D);
    vf[i] = *reinterpret_cast<const uint4*>(rp + 2 * D);
  }
}
template <int NI>
__device__ __forceinline__ void attn_finish(const uint4 (&kf)[NI], const uint4 (&vf)[NI], const AttnCur<NI>& c, const DArgs& a, const DLayer& L, const Grp& G, int li, int row, int h,
                                            int t) {
  const int lane = threadIdx.x & 63, kg = lane >> 3, dc = lane & 7, b = G.b0 + row;
  const int qc = a.n_enc + t, nk = qc + 1;
  const uint4 qu = c.q, kcur = c.k, vcur = c.v;
  const float q8[8] = {bf_lo(qu.x), bf_hi(qu.x), bf_lo(qu.y), bf_hi(qu.y), bf_lo(qu.z), bf_hi(qu.z), bf_lo(qu.w), bf_hi(qu.w)};
  float s[NI];
  float mx = -INFINITY;
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int j = kg + 8 * i;
    const uint4 u = j == qc ? kcur : kf[i];
    float d = q8[0] * bf_lo(u.x);
    d = fmaf(q8[1], bf_hi(u.x), d); d = fmaf(q8[2], bf_lo(u.y), d); d = fmaf(q8[3], bf_hi(u.y), d);
    d = fmaf(q8[4], bf_lo(u.z), d); d = fmaf(q8[5], bf_hi(u.z), d); d = fmaf(q8[6], bf_lo(u.w), d); d = fmaf(q8[7], bf_hi(u.w), d);
    d = sum8(d);
    const bool valid = j < nk && ((c.aw[i / 4] >> (j & 31)) & 1u);      // (key 8 i + kg lies in word i / 4)
    s[i] = valid ? d * a.scale_log2 : -INFINITY;
    mx = fmaxf(mx, s[i]);
  }
  mx = fmaxf(mx, __shfl_xor(mx, 8)); mx = fmaxf(mx, __shfl_xor(mx, 16)); mx = fmaxf(mx, __shfl_xor(mx, 32));
  float l = 0.f, acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const float p = s[i] == -INFINITY ? 0.f : __builtin_amdgcn_exp2f(s[i] - mx);
    l += p;
    uint4 u = (kg + 8 * i) == qc ? vcur : vf[i];
    if (p == 0.f) u = uint4{0u, 0u, 0u, 0u};                    // (rows past the current one are never meant to be read: whatever they hold, 0 x it must be 0)
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
    uint4 o4;
    o4.x = pack_bf16x2(acc[0] * inv, acc[1] * inv); o4.y = pack_bf16x2(acc[2] * inv, acc[3] * inv);
    o4.z = pack_bf16x2(acc[4] * inv, acc[5] * inv); o4.w = pack_bf16x2(acc[6] * inv, acc[7] * inv);
    *reinterpret_cast<uint4*>(ctxbuf(a, G, t, li) + tiled(row, h * HD + dc * 8)) = o4;
  }
}

// The same task for caches of more than 256 rows (the 350-token shape: 48 key/value chunks per lane would not fit the register file): the keys are
// taken 8 * NI at a time with a running maximum -- the sum and the accumulators are rescaled when a later chunk raises it (exact in fp32 up to
// the rounding of one extra multiply, so this path is not bit-identical to the one-pass version and serves only the sizes that one cannot).
template <int NI>
__device__ __forceinline__ void attn_chunked(const DArgs& a, const DLayer& L, const Grp& G, int li, int row, int h, int t) {
  const int lane = threadIdx.x & 63, b = G.b0 + row, kg = lane >> 3, dc = lane & 7;
  const int qc = a.n_enc + t, nk = qc + 1;
  const bf16_t* base = L.qkv + (long long)b * a.N * (3 * D) + h * HD + dc * 8;
  const bf16_t* cur = base + (long long)qc * (3 * D);
  const uint4 qu = *reinterpret_cast<const uint4*>(cur), kcur = *reinterpret_cast<const uint4*>(cur + D), vcur = *reinterpret_cast<const uint4*>(cur + 2 * D);
  const uint32_t* ap = L.allow + b * L.allow_sb + h * L.allow_sh + (long long)qc * a.NWORDS;
  const float q8[8] = {bf_lo(qu.x), bf_hi(qu.x), bf_lo(qu.y), bf_hi(qu.y), bf_lo(qu.z), bf_hi(qu.z), bf_lo(qu.w), bf_hi(qu.w)};
  float m = -INFINITY, l = 0.f, acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (int k0 = 0; k0 < nk; k0 += 8 * NI) {                     // (8 NI is a multiple of 32: a chunk starts on an allow word)
    uint4 kf[NI], vf[NI];
    uint32_t aw[NI / 4];
#pragma unroll
    for (int w = 0; w < NI / 4; ++w) aw[w] = ap[min((k0 >> 5) + w, a.NWORDS - 1)];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const bf16_t* rp = base + (long long)min(k0 + kg + 8 * i, max(qc - 1, 0)) * (3 * D);
      kf[i] = *reinterpret_cast<const uint4*>(rp + D);
      vf[i] = *reinterpret_cast<const uint4*>(rp + 2 * D);
    }
    float s[NI];
    float mx = -INFINITY;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int j = k0 + kg + 8 * i;
      const uint4 u = j == qc ? kcur : kf[i];
      float d = q8[0] * bf_lo(u.x);
      d = fmaf(q8[1], bf_hi(u.x), d); d = fmaf(q8[2], bf_lo(u.y), d); d = fmaf(q8[3], bf_hi(u.y), d);
      d = fmaf(q8[4], bf_lo(u.z), d); d = fmaf(q8[5], bf_hi(u.z), d); d = fmaf(q8[6], bf_lo(u.w), d); d = fmaf(q8[7], bf_hi(u.w), d);
      d = sum8(d);
      const bool valid = j < nk && ((aw[i / 4] >> (j & 31)) & 1u);
      s[i] = valid ? d * a.scale_log2 : -INFINITY;
      mx = fmaxf(mx, s[i]);
    }
    mx = fmaxf(mx, __shfl_xor(mx, 8)); mx = fmaxf(mx, __shfl_xor(mx, 16)); mx = fmaxf(mx, __shfl_xor(mx, 32));
    const float mn = fmaxf(m, mx);
    const float f = m == -INFINITY ? 0.f : __builtin_amdgcn_exp2f(m - mn);
    l *= f;
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] *= f;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const float p = s[i] == -INFINITY ? 0.f : __builtin_amdgcn_exp2f(s[i] - mn);
      l += p;
      uint4 u = (k0 + kg + 8 * i) == qc ? vcur : vf[i];
      if (p == 0.f) u = uint4{0u, 0u, 0u, 0u};
      acc[0] = fmaf(p, bf_lo(u.x), acc[0]); acc[1] = fmaf(p, bf_hi(u.x), acc[1]); acc[2] = fmaf(p, bf_lo(u.y), acc[2]); acc[3] = fmaf(p, bf_hi(u.y), acc[3]);
      acc[4] = fmaf(p, bf_lo(u.z), acc[4]); acc[5] = fmaf(p, bf_hi(u.z), acc[5]); acc[6] = fmaf(p, bf_lo(u.w), acc[6]); acc[7] = fmaf(p, bf_hi(u.w), acc[7]);
    }
    m = mn;
  }
#pragma unroll
  for (int o = 8; o < 64; o <<= 1) {
    l += __shfl_xor(l, o);
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] += __shfl_xor(acc[e], o);
  }
  if (kg == 0) {
    const float inv = l > 0.f ? 1.0f / l : 0.f;
    uint4 o4;
    o4.x = pack_bf16x2(acc[0] * inv, acc[1] * inv); o4.y = pack_bf16x2(acc[2] * inv, acc[3] * inv);
    o4.z = pack_bf16x2(acc[4] * inv, acc[5] * inv); o4.w = pack_bf16x2(acc[6] * inv, acc[7] * inv);
    *reinterpret_cast<uint4*>(ctxbuf(a, G, t, li) + tiled(row, h * HD + dc * 8)) = o4;
  }
}

// pointer scores, argmax, the next token and the next step's input row of local sample `row`: one block
__device__ __forceinline__ void pick_task(const DArgs& a, const Grp& G, int row, int t, float* lds, Best* red) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, b = G.b0 + row;
  float* qs = lds;            // [D]
  float* dyn = lds + D;       // [No <= MAXNO]
  const rsrc_t pr = rsrc(partbuf(a, G));
  for (int d = tid; d < D; d += NT) {
    const float p = (__uint_as_float(ld4_l2(pr, ((0 * 16 + row) * D + d) * 4)) + __uint_as_float(ld4_l2(pr, ((1 * 16 + row) * D + d) * 4))) +
                    (__uint_as_float(ld4_l2(pr, ((2 * 16 + row) * D + d) * 4)) + __uint_as_float(ld4_l2(pr, ((3 * 16 + row) * D + d) * 4)));
    qs[d] = bf2f(f2bf(p + a.bq[d]));
  }
  // (L1-bypassing loads: logits rows are not cache-line aligned -- 20 000 bytes at V = 5000 --, the line row t shares with row t + 1 would be stale
  // in this CU's L1 next step).  Issued before the pointer scores: the two round trips overlap.
  const rsrc_t fr = rsrc(a.fixed_all + ((long long)b * a.S + t) * a.ldf);
  constexpr int FPT = 24;                                   // logits per thread held in registers: covers V <= 6144, the rest is looped
  float fv[FPT];
#pragma unroll
  for (int q = 0; q < FPT; ++q) { const int j = tid + q * NT; fv[q] = j < a.V ? __uint_as_float(ld4_l2(fr, j * 4)) : -INFINITY; }
  __syncthreads();
  for (int o0 = 0; o0 < a.No; o0 += 64) {
    // this wave's OCR keys of this group of 64 (o = o0 + wave, + 4, ...: at most 16), every row chunk requested before the first dot product
    constexpr int KPW = 64 / NW;
    uint2 kr[KPW][NCH];
#pragma unroll
    for (int q = 0; q < KPW; ++q) {
      const bf16_t* kp = a.ptr_k + ((long long)b * a.No + min(o0 + wave + q * NW, a.No - 1)) * D;
#pragma unroll
      for (int j = 0; j < NCH; ++j) kr[q][j] = *reinterpret_cast<const uint2*>(kp + 4 * (lane + 64 * j));
    }
#pragma unroll
    for (int q = 0; q < KPW; ++q) {
      const int o = o0 + wave + q * NW;
      float sum = 0.f;
#pragma unroll
      for (int j = 0; j < NCH; ++j) {
        const int c = 4 * (lane + 64 * j);
        const uint2 u = kr[q][j];
        sum += (qs[c] * bf_lo(u.x) + qs[c + 1] * bf_hi(u.x)) + (qs[c + 2] * bf_lo(u.y) + qs[c + 3] * bf_hi(u.y));
      }
      sum = wave_sum(sum);
      if (lane == 0 && o < a.No) {
        const float sc = sum * a.ptr_scale + (a.ocr_mask[(long long)b * a.No + o] ? 0.f : -10000.0f);
        dyn[o] = sc;
        a.dyn_all[((long long)b * a.S + t) * a.No + o] = sc;
      }
    }
  }
  __syncthreads();
  Best x = {-INFINITY, 0x7fffffff};
#pragma unroll
  for (int q = 0; q < FPT; ++q) { const int j = tid + q * NT; if (j < a.V) x = better(x, Best{fv[q], j}); }
  for (int j = tid + FPT * NT; j < a.V; j += NT) x = better(x, Best{__uint_as_float(ld4_l2(fr, j * 4)), j});
  if (tid < a.No) x = better(x, Best{dyn[tid], a.V + tid});
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
  for (int w = 1; w < NW; ++w) r = better(r, red[w]);
  const long long tok = r.i == 0x7fffffff ? 0 : r.i;
  if (t + 1 < a.S) {
    if (tid == 0) a.prev[(long long)b * a.S + t + 1] = tok;
    if (wave == 0) embed_row(a, G, row, t + 1, tok);
  }
  __syncthreads();
}

template <int NI, bool CHUNKED>
__global__ __launch_bounds__(NT, 1) void decode_steps_kernel(DArgs a) {
  __shared__ __attribute__((aligned(16))) float lds[D + MAXNO];
  __shared__ Best red[NW];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g4 = lane >> 4, li16 = lane & 15;
  // which XCD this block landed on (HW_REG_XCC_ID[3:0]) and its rank among that XCD's blocks (an L2 counter, first come first served).  The
  // dispatcher deals workgroups to the XCDs round-robin, but the deal continues where the previous kernel's stopped: blockIdx % 8 is the XCD only
  // by accident (measured: true in a process that launches nothing but 256-block grids, false in the middle of the model's kernels).  Any window
  // of 256 consecutive workgroups still puts gridDim / 8 on each XCD, which is all the partition needs.
  __shared__ int s_rank;
  const unsigned nblk = gridDim.x / NXCD;
  const unsigned xcc = __builtin_amdgcn_s_getreg(((4 - 1) << 11) | (0 << 6) | 20) & 0xfu;
  if (threadIdx.x == 0) s_rank = xcc < NXCD ? (int)__hip_atomic_fetch_add(a.bar + xcc * 32 + 2, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) : -1;
  __syncthreads();
  Grp G;
  G.g = (int)xcc; G.r = s_rank;
  if (G.r < 0 || G.r >= (int)nblk) {          // not eight XCDs with gridDim / 8 blocks each: refuse to run (the launch reports it)
    if (threadIdx.x == 0) __hip_atomic_store(a.err, 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return;
  }
  G.b0 = G.g * a.spx; G.nloc = min(a.spx, a.B - G.b0);
  if (G.nloc <= 0) {                  // a whole XCD without samples: nobody waits for its blocks; the last of them to leave clears the rank counter
    if (threadIdx.x == 0) {
      unsigned* bar = a.bar + G.g * 32;
      if (__hip_atomic_fetch_add(bar + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == nblk - 1) {
        __hip_atomic_store(bar + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __hip_atomic_store(bar + 2, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
    }
    return;
  }
  const int wg = G.r * NW + wave, nwaves = nblk * NW;
  unsigned epoch = 0;
  float* part = partbuf(a, G);
  // (Tried: requesting a wave's first-task weight fragments -- and the cache rows below the current one -- BEFORE it arrives at the preceding
  // barrier, to keep the fabric streaming through the barriers.  The register arrays that have to stay live around the loop pushed the kernel
  // over 512 registers per lane (130-820 spilled, depending on how much was prefetched) and the step got slower: 360 us against 330 us.  A small
  // version -- the first 8 of 24 k-steps, requested inside the barrier after the stores have drained -- fits, and is neutral: 5.58 against 5.60 ms.)
  const int QT = 3 * D / 16, OT = (D / 16) * KSPLIT, GT = F / 16, CT = (a.V + 15) >> 4;
  for (int row = wg; row < G.nloc; row += nwaves) embed_row(a, G, row, a.t_begin, a.prev[(long long)(G.b0 + row) * a.S + a.t_begin]);
  if (!xcd_sync(a, G, nblk, epoch)) return;
  for (int t = a.t_begin; t < a.t_end; ++t) {
    for (int li = 0; li < a.n_layers; ++li) {
      const DLayer& L = a.L[li];
      const bool last = li + 1 == a.n_layers;
      // Q: q|k|v row of every sample into the cache
      for (int nt = wg; nt < QT; nt += nwaves) {
        const f32x4 acc = wave_tile<D / 32>(L.wqkv, D, nt, xbuf(a, G, t, li), 0);
        if (li16 < G.nloc) {
          const int n = nt * 16 + 4 * g4;
          const float4 b4 = *reinterpret_cast<const float4*>(L.bqkv + n);
          const float v[4] = {acc[0] + b4.x, acc[1] + b4.y, acc[2] + b4.z, acc[3] + b4.w};
          st_bf16x4(L.qkv + ((long long)(G.b0 + li16) * a.N + a.n_enc + t) * (3 * D) + n, v);
        }
      }
      if (!xcd_sync(a, G, nblk, epoch)) return;
      // A
      for (int task = wg; task < G.nloc * a.H; task += nwaves) {
        if constexpr (CHUNKED) {
          attn_chunked<NI>(a, L, G, li, task / a.H, task % a.H, t);
        } else {
          uint4 kf[NI], vf[NI];
          AttnCur<NI> cur;
          attn_load<NI>(kf, vf, cur, a, L, G, li, task / a.H, task % a.H, t);
          attn_finish<NI>(kf, vf, cur, a, L, G, li, task / a.H, task % a.H, t);
        }
      }
      if (!xcd_sync(a, G, nblk, epoch)) return;
      // O: split-K partials
      for (int task = wg; task < OT; task += nwaves) {
        const int nt = task / KSPLIT, ks = task % KSPLIT;
        const f32x4 acc = wave_tile<D / 32 / KSPLIT>(L.wo, D, nt, ctxbuf(a, G, t, li), ks * (D / 8 / KSPLIT));
        if (li16 < G.nloc) *reinterpret_cast<f32x4*>(part + ((long long)(ks * 16 + li16)) * D + nt * 16 + 4 * g4) = acc;
      }
      if (!xcd_sync(a, G, nblk, epoch)) return;
      for (int row = wg; row < G.nloc; row += nwaves) finalize_row(a, G, row, L.bo, xbuf(a, G, t, li), L.g1, L.be1, x1buf(a, G, t, li), nullptr);
      if (!xcd_sync(a, G, nblk, epoch)) return;
      // G: FFN1 + erf-GELU
      for (int nt = wg; nt < GT; nt += nwaves) {
        const f32x4 acc = wave_tile<D / 32>(L.w1, D, nt, x1buf(a, G, t, li), 0);
        if (li16 < G.nloc) {
          const int n = nt * 16 + 4 * g4;
          const float4 b4 = *reinterpret_cast<const float4*>(L.b1 + n);
          const float v[4] = {gelu_erf(acc[0] + b4.x), gelu_erf(acc[1] + b4.y), gelu_erf(acc[2] + b4.z), gelu_erf(acc[3] + b4.w)};
          st_bf16x4(hbuf(a, G, t, li) + tiled(li16, n), v);
        }
      }
      if (!xcd_sync(a, G, nblk, epoch)) return;
      // H: FFN2 split-K partials
      for (int task = wg; task < OT; task += nwaves) {
        const int nt = task / KSPLIT, ks = task % KSPLIT;
        const f32x4 acc = wave_tile<F / 32 / KSPLIT>(L.w2, F, nt, hbuf(a, G, t, li), ks * (F / 8 / KSPLIT));
        if (li16 < G.nloc) *reinterpret_cast<f32x4*>(part + ((long long)(ks * 16 + li16)) * D + nt * 16 + 4 * g4) = acc;
      }
      if (!xcd_sync(a, G, nblk, epoch)) return;
      for (int row = wg; row < G.nloc; row += nwaves)
        finalize_row(a, G, row, L.b2, x1buf(a, G, t, li), L.g2, L.be2, xbuf(a, G, t, li + 1),
                     last && a.seq ? a.seq + ((long long)(G.b0 + row) * a.N + a.n_enc + t) * D : nullptr);
      if (!xcd_sync(a, G, nblk, epoch)) return;
    }
    // C: classifier logits of row t | pointer-network query partials
    {
      const bf16_t* xl = xbuf(a, G, t, a.n_layers);
      for (int task = wg; task < CT + OT; task += nwaves) {
        if (task < CT) {
          const f32x4 acc = wave_tile<D / 32>(a.wc, D, task, xl, 0);
          if (li16 < G.nloc) {
            float* frow = a.fixed_all + ((long long)(G.b0 + li16) * a.S + t) * a.ldf;
            const int n = task * 16 + 4 * g4;
#pragma unroll
            for (int r = 0; r < 4; ++r)
              if (n + r < a.V) frow[n + r] = acc[r] + a.bc[n + r];
          }
        } else {
          const int nt = (task - CT) / KSPLIT, ks = (task - CT) % KSPLIT;
          const f32x4 acc = wave_tile<D / 32 / KSPLIT>(a.wq, D, nt, xl, ks * (D / 8 / KSPLIT));
          if (li16 < G.nloc) *reinterpret_cast<f32x4*>(part + ((long long)(ks * 16 + li16)) * D + nt * 16 + 4 * g4) = acc;
        }
      }
    }
    if (!xcd_sync(a, G, nblk, epoch)) return;
    for (int row = G.r; row < G.nloc; row += nblk) pick_task(a, G, row, t, lds, red);
    if (!xcd_sync(a, G, nblk, epoch)) return;
  }
  // the last block of the XCD to leave puts its barrier words back to zero for the next launch (every block is past the final barrier by then)
  if (threadIdx.x == 0) {
    unsigned* bar = a.bar + G.g * 32;
    const unsigned left = __hip_atomic_fetch_add(bar + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (left == nblk - 1) {
      __hip_atomic_store(bar, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      __hip_atomic_store(bar + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      __hip_atomic_store(bar + 2, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
  }
}

constexpr int64_t WS_HEADER = NXCD * 128 + 128 + 8192;        // barrier words per XCD, error word, 1024 profile stamps

}  // namespace

extern "C" int64_t sam_greedy_decode_ws_bytes(int B, int S, int n_layers) {
  if (B <= 0 || S <= 0 || n_layers <= 0) return 0;
  const int64_t s = S, l = n_layers;
  return WS_HEADER + (int64_t)NXCD * KSPLIT * 16 * D * (int64_t)sizeof(float) + (int64_t)NXCD * s * ((l + 1) * 16 * D + l * (2 * 16 * D + 16 * F)) * (int64_t)sizeof(bf16_t);
}

extern "C" int sam_greedy_decode_steps(const sam_decode_desc* d, void* ws, int64_t ws_bytes, void* stream) {
  SAM_REQUIRE(d && ws && d->layers, "sam_greedy_decode_steps: null pointer");
  const int grid = samgemm::device_cu_count();
  if (d->D != D || d->F != F || d->H * HD != D || d->n_layers < 1 || d->n_layers > MAXL || d->N > 384 || d->No > MAXNO || d->B > 16 * NXCD || grid % NXCD != 0) {
    sam_set_error("sam_greedy_decode_steps: built for D=768, F=3072, head_dim=64, <= %d layers, N <= 384, <= %d OCR slots, B <= %d, a CU count that is a multiple of 8 "
                  "(got D=%d F=%d H=%d L=%d N=%d No=%d B=%d CUs=%d)", MAXL, MAXNO, 16 * NXCD, d->D, d->F, d->H, d->n_layers, d->N, d->No, d->B, grid);
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
  a.n_layers = d->n_layers; a.B = d->B; a.spx = (d->B + NXCD - 1) / NXCD; a.N = d->N; a.n_enc = d->n_enc; a.S = d->S; a.H = d->H;
  a.NWORDS = sam_attn_words_per_row(d->N);      // the ROW STRIDE of the allow masks as the packers write them (1, 2, 4, 6, 8 or 12 words), not ceil(N / 32)
  a.V = d->V; a.No = d->No; a.t_begin = d->t_begin; a.t_end = d->t_end;
  a.ldf = d->ld_fixed; a.ld_pos = d->ld_pos; a.ld_type = d->ld_type;
  a.scale_log2 = d->scale * 1.44269504088896341f; a.eps = d->ln_eps; a.eps_emb = d->emb_ln_eps; a.ptr_scale = d->ptr_scale;
  a.pos_emb = d->pos_emb; a.type_emb = d->type_emb; a.emb_g = d->emb_ln_g; a.emb_b = d->emb_ln_b;
  a.ans_ln = (const bf16_t*)d->ans_ln; a.ocr_ln = (const bf16_t*)d->ocr_ln; a.wc = (const bf16_t*)d->wc; a.wq = (const bf16_t*)d->wq; a.ptr_k = (const bf16_t*)d->ptr_k;
  a.bc = d->bc; a.bq = d->bq; a.ocr_mask = d->ocr_mask; a.prev = (long long*)d->prev_inds; a.fixed_all = d->fixed_scores; a.dyn_all = d->ocr_scores;
  a.seq = (bf16_t*)d->seq_out;
  char* w = (char*)ws;
  a.bar = (unsigned*)w; a.err = (int*)(w + NXCD * 128);
  { static int prof = -1; if (prof < 0) { const char* e = getenv("SAM_DECODE_PROF"); prof = e ? atoi(e) : 0; } a.prof = prof ? (long long*)(w + NXCD * 128 + 128) : nullptr; }
  w += WS_HEADER;
  const int64_t sl = (int64_t)NXCD * d->S * d->n_layers;
  a.part = (float*)w; w += (int64_t)NXCD * KSPLIT * 16 * D * sizeof(float);
  a.x = (bf16_t*)w; w += (int64_t)NXCD * d->S * (d->n_layers + 1) * 16 * D * sizeof(bf16_t);
  a.x1 = (bf16_t*)w; w += sl * 16 * D * sizeof(bf16_t);
  a.ctx = (bf16_t*)w; w += sl * 16 * D * sizeof(bf16_t);
  a.h = (bf16_t*)w;
  hipStream_t st = (hipStream_t)stream;
  // (no memset node in front of the kernel: the launch leaves the barrier words at zero itself.  A hipMemsetAsync captured ahead of the kernel
  // left the counter non-zero on graph replays -- barriers fell through, tokens came out wrong)
  if (d->N <= 192) decode_steps_kernel<24, false><<<dim3(grid), dim3(NT), 0, st>>>(a);
  else if (d->N <= 256) decode_steps_kernel<32, false><<<dim3(grid), dim3(NT), 0, st>>>(a);
  else decode_steps_kernel<24, true><<<dim3(grid), dim3(NT), 0, st>>>(a);          // two chunks of 192 keys with a running maximum
  SAM_LAUNCH_CHECK();
  return SAM_OK;
}
