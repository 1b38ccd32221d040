// Token selection of the auto-regressive decoding loops (gfx950): the greedy pick of sam/sa_m4c.py:294-302 and one step of the beam search of
// sam/beam_search.py:84-160 -- on the two score blocks the model produces (classifier logits [rows, V] and pointer scores [rows, No]; the
// reference concatenates them into `scores` first), in place on the decoder's state, with nothing going through the host: both are nodes of the
// captured decoding step (decoder.py).
#include "common.h"
#include "sam_hip.h"

namespace {

constexpr int NT = 256;

// (value, index) maximum with the FIRST index winning ties (torch.argmax / the stable reading of topk)
struct Best { float v; int i; };
__device__ __forceinline__ Best better(Best a, Best b) { return (b.v > a.v || (b.v == a.v && b.i < a.i)) ? b : a; }
__device__ __forceinline__ Best wave_best(Best x) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    Best y;
    y.v = __shfl_xor(x.v, o);
    y.i = __shfl_xor(x.i, o);
    x = better(x, y);
  }
  return x;
}
__device__ __forceinline__ Best block_best(Best x, Best* red) {      // red: NT / 64 entries of LDS; result valid in every thread
  x = wave_best(x);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = x;
  __syncthreads();
  Best r = red[0];
#pragma unroll
  for (int w = 1; w < NT / 64; ++w) r = better(r, red[w]);
  return r;
}

// prev_inds[b, s + 1] = argmax_j scores[b, s, j] for s < S - 1 (sa_m4c.py:299-302: `train_prev_inds[:, 1:] = argmax_inds[:, :-1]`); one block per (b, s)
__global__ __launch_bounds__(NT) void greedy_pick_kernel(const float* __restrict__ fixed, int64_t ldf, const float* __restrict__ ocr, int64_t ldo, int V, int No, int S,
                                                         int64_t* __restrict__ prev_inds) {
  __shared__ Best red[NT / 64];
  const int row = blockIdx.x, s = row % S;
  if (s == S - 1) return;
  Best x = {-INFINITY, 0x7fffffff};
  const float* f = fixed + (int64_t)row * ldf;
  const float* o = ocr + (int64_t)row * ldo;
  for (int j = threadIdx.x; j < V + No; j += NT) {
    const float v = j < V ? f[j] : o[j - V];
    x = better(x, Best{v, j});
  }
  x = block_best(x, red);
  if (threadIdx.x == 0) prev_inds[row + 1] = x.i == 0x7fffffff ? 0 : x.i;
}

// One step of BeamSearch.decode for every sample (one block per sample, K beams = rows b*K .. b*K+K-1 of every state tensor).
//   candidates (beam j, token c): log(sigmoid(score[j, t, c])) + cum[j]; a COMPLETED beam (done[j]: its sequence holds EOS at position t) only offers EOS, at
//   log-probability 0 (beam_search.py:89-93); at t == 0 only beam 0 is live (:98-105).  The K best over the flattened [K * Vt] axis, in descending
//   order, lower flat index first among equals (:107-110), give source beam = idx / Vt (integer division, :113) and token = idx % Vt;
//   seqs[j] = seqs[src], seqs[j][t + 1] = token when t + 1 < S (:170-174); cum[j] = cum[src] + value (:126-128 -- the value already contains cum[src]:
//   the reference counts it twice, and so does this); done[j] = seqs[j][t + 1] == EOS, or every beam when the steps ran out (:140-147).
// ctl (device): ctl[0] = t (advanced here by the last block to finish), ctl[1] = finished (set once every beam of every sample is complete or the
// steps ran out: beam_search.py:149-158 -- later launches of the same captured step are then no-ops, as the reference leaves its loop), ctl[2] =
// blocks that have finished this launch, ctl[3] = number of rows found done this launch.
template <int KMAX>
__global__ __launch_bounds__(NT) void beam_step_kernel(const float* __restrict__ fixed, int64_t ldf, const float* __restrict__ ocr, int64_t ldo, int V, int No, int S,
                                                       int K, int B, int eos, int t_by_value, int* __restrict__ ctl, float* __restrict__ cum, unsigned char* __restrict__ done,
                                                       int64_t* __restrict__ seqs, int64_t* __restrict__ prev_pos) {
  __shared__ Best red[NT / 64];
  __shared__ float s_cum[KMAX], s_val[KMAX];
  __shared__ int s_idx[KMAX], s_done[KMAX];
  __shared__ int64_t s_seq[KMAX * 64];
  const int b = blockIdx.x, tid = threadIdx.x, Vt = V + No;
  const int t = ctl ? ctl[0] : t_by_value;
  if (ctl && ctl[1]) {                             // the search is over: state untouched, every beam its own source
    if (prev_pos && tid < K) prev_pos[b * K + tid] = b * K + tid;
    return;
  }
  if (tid < K) { s_cum[tid] = cum[b * K + tid]; s_done[tid] = done[b * K + tid]; }
  for (int e = tid; e < K * S; e += NT) s_seq[e] = seqs[(int64_t)b * K * S + e];
  __syncthreads();
  // per-thread sorted list of its K best candidates (descending, earlier index first among equals)
  float lv[KMAX]; int li[KMAX];
#pragma unroll
  for (int q = 0; q < KMAX; ++q) { lv[q] = -INFINITY; li[q] = 0x7fffffff; }
  float wv = -INFINITY; int wi = 0x7fffffff;
  const int live = t == 0 ? 1 : K;
  for (int j = 0; j < live; ++j) {
    const int64_t row = (int64_t)(b * K + j) * S + t;
    const float* f = fixed + row * ldf;
    const float* o = ocr + row * ldo;
    const float cj = s_cum[j];
    if (s_done[j]) {
      if (tid == 0) {       // the single candidate of a completed beam
        float v = cj; int id = j * Vt + eos;
#pragma unroll
        for (int q = 0; q < KMAX; ++q)
          if (q < K && (v > lv[q] || (v == lv[q] && id < li[q]))) { const float tv = lv[q]; const int ti = li[q]; lv[q] = v; li[q] = id; v = tv; id = ti; }
#pragma unroll
        for (int q = 0; q < KMAX; ++q)
          if (q == K - 1) { wv = lv[q]; wi = li[q]; }
      }
      continue;
    }
    // UN scores per thread are requested before the first is looked at: one load per iteration was a chain of ~20 exposed memory latencies per
    // beam (17-20 us per beam and step, 100 us per step at beam 5, with 64 blocks on the device and nothing else to hide them)
    constexpr int UN = 8;
    for (int c0 = tid; c0 < Vt; c0 += NT * UN) {
      float xs[UN];
#pragma unroll
      for (int u = 0; u < UN; ++u) {
        const int c = min(c0 + u * NT, Vt - 1);
        xs[u] = c < V ? f[c] : o[c - V];
      }
#pragma unroll
      for (int u = 0; u < UN; ++u) {
        const int c = c0 + u * NT;
        if (c >= Vt) break;
        float v = logf(1.0f / (1.0f + expf(-xs[u]))) + cj;      // torch.log(torch.sigmoid(x)) + topkscores
        int id = j * Vt + c;
        if (v > wv || (v == wv && id < wi)) {       // beats this thread's K-th best so far
#pragma unroll
          for (int q = 0; q < KMAX; ++q)
            if (q < K && (v > lv[q] || (v == lv[q] && id < li[q]))) { const float tv = lv[q]; const int ti = li[q]; lv[q] = v; li[q] = id; v = tv; id = ti; }
#pragma unroll
          for (int q = 0; q < KMAX; ++q)
            if (q == K - 1) { wv = lv[q]; wi = li[q]; }      // (static register indices only: a runtime index would send the lists to scratch)
        }
      }
    }
  }
  // K rounds: every thread offers the head of its list, the block's best is taken, its owner moves on
  for (int r = 0; r < K; ++r) {
    const Best w = block_best(Best{lv[0], li[0]}, red);
    if (w.i == li[0] && w.i != 0x7fffffff) {       // (flat indices are unique: exactly one owner)
#pragma unroll
      for (int q = 0; q + 1 < KMAX; ++q) { lv[q] = lv[q + 1]; li[q] = li[q + 1]; }
      lv[KMAX - 1] = -INFINITY; li[KMAX - 1] = 0x7fffffff;
    }
    if (tid == 0) { s_val[r] = w.v; s_idx[r] = w.i == 0x7fffffff ? 0 : w.i; }
  }
  __syncthreads();
  // new state, in place (everything of this sample was read into LDS above)
  for (int e = tid; e < K * S; e += NT) {
    const int j = e / S, pos = e - j * S, src = s_idx[j] / Vt;
    int64_t v = s_seq[src * S + pos];
    if (pos == t + 1) v = s_idx[j] % Vt;
    seqs[(int64_t)b * K * S + e] = v;
  }
  int ndone = 0;
  if (tid < K) {
    const int src = s_idx[tid] / Vt, tok = s_idx[tid] % Vt;
    cum[b * K + tid] = s_cum[src] + s_val[tid];
    const int d = (t + 1 < S) ? (tok == eos ? 1 : 0) : 1;
    done[b * K + tid] = (unsigned char)d;
    if (prev_pos) prev_pos[b * K + tid] = b * K + src;
    ndone = d;
  }
  if (ctl) {
    // (K <= 16 lanes of wave 0 hold the flags)
    if (tid < 64) {
      ndone = (int)wave_sum((float)ndone);
      if (tid == 0) {
        atomicAdd(ctl + 3, ndone);
        __threadfence();
        const int arrived = atomicAdd(ctl + 2, 1);
        if (arrived == B - 1) {                  // last block: every block has read ctl[0] / ctl[1] long ago
          const int total = atomicAdd(ctl + 3, 0);
          ctl[0] = t + 1;
          if (total == B * K || t + 1 >= S) ctl[1] = 1;
          ctl[2] = 0; ctl[3] = 0;
          __threadfence();
        }
      }
    }
  }
}

// ---- the same step as two launches: the candidate scan -- K * (V + No) log-sigmoids per sample, the step's whole cost -- spread over one block per
// (sample, beam) instead of one per sample (64 blocks of four waves on 256 CUs: 76 us per step at beam 5), then a merge of the K lists per sample.
// ws: [B][K][KMAXW] (value, flat index) pairs, written by the scan, read by the merge.
constexpr int KMAXW = 16;
template <int KMAX>
__global__ __launch_bounds__(NT) void beam_scan_kernel(const float* __restrict__ fixed, int64_t ldf, const float* __restrict__ ocr, int64_t ldo, int V, int No, int S, int K,
                                                       int eos, int t_by_value, const int* __restrict__ ctl, const float* __restrict__ cum,
                                                       const unsigned char* __restrict__ done, Best* __restrict__ ws) {
  __shared__ Best red[NT / 64];
  const int b = blockIdx.x / K, j = blockIdx.x % K, tid = threadIdx.x, Vt = V + No;
  const int t = ctl ? ctl[0] : t_by_value;
  if (ctl && ctl[1]) return;
  Best* out = ws + (int64_t)blockIdx.x * KMAXW;
  const float cj = cum[b * K + j];
  if (t == 0 && j > 0) {                           // only beam 0 is live at the first step
    if (tid < K) out[tid] = Best{-INFINITY, 0x7fffffff};
    return;
  }
  if (done[b * K + j]) {                           // the single candidate of a completed beam
    if (tid < K) out[tid] = tid == 0 ? Best{cj, j * Vt + eos} : Best{-INFINITY, 0x7fffffff};
    return;
  }
  float lv[KMAX]; int li[KMAX];
#pragma unroll
  for (int q = 0; q < KMAX; ++q) { lv[q] = -INFINITY; li[q] = 0x7fffffff; }
  float wv = -INFINITY; int wi = 0x7fffffff;
  const int64_t row = (int64_t)(b * K + j) * S + t;
  const float* f = fixed + row * ldf;
  const float* o = ocr + row * ldo;
  constexpr int UN = 8;
  for (int c0 = tid; c0 < Vt; c0 += NT * UN) {
    float xs[UN];
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const int c = min(c0 + u * NT, Vt - 1);
      xs[u] = c < V ? f[c] : o[c - V];
    }
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const int c = c0 + u * NT;
      if (c >= Vt) break;
      float v = logf(1.0f / (1.0f + expf(-xs[u]))) + cj;
      int id = j * Vt + c;
      if (v > wv || (v == wv && id < wi)) {
#pragma unroll
        for (int q = 0; q < KMAX; ++q)
          if (q < K && (v > lv[q] || (v == lv[q] && id < li[q]))) { const float tv = lv[q]; const int ti = li[q]; lv[q] = v; li[q] = id; v = tv; id = ti; }
#pragma unroll
        for (int q = 0; q < KMAX; ++q)
          if (q == K - 1) { wv = lv[q]; wi = li[q]; }
      }
    }
  }
  for (int r = 0; r < K; ++r) {
    const Best w = block_best(Best{lv[0], li[0]}, red);
    if (w.i == li[0] && w.i != 0x7fffffff) {
#pragma unroll
      for (int q = 0; q + 1 < KMAX; ++q) { lv[q] = lv[q + 1]; li[q] = li[q + 1]; }
      lv[KMAX - 1] = -INFINITY; li[KMAX - 1] = 0x7fffffff;
    }
    if (tid == 0) out[r] = w;
  }
}

__global__ __launch_bounds__(NT) void beam_merge_kernel(const Best* __restrict__ ws, int Vt, int S, int K, int B, int eos, int t_by_value, int* __restrict__ ctl,
                                                        float* __restrict__ cum, unsigned char* __restrict__ done, int64_t* __restrict__ seqs, int64_t* __restrict__ prev_pos) {
  __shared__ Best red[NT / 64];
  __shared__ float s_cum[KMAXW], s_val[KMAXW];
  __shared__ int s_idx[KMAXW];
  __shared__ int64_t s_seq[KMAXW * 64];
  const int b = blockIdx.x, tid = threadIdx.x;
  const int t = ctl ? ctl[0] : t_by_value;
  if (ctl && ctl[1]) {
    if (prev_pos && tid < K) prev_pos[b * K + tid] = b * K + tid;
    return;
  }
  if (tid < K) s_cum[tid] = cum[b * K + tid];
  for (int e = tid; e < K * S; e += NT) s_seq[e] = seqs[(int64_t)b * K * S + e];
  // thread (j, r) holds candidate r of beam j's list (K * K <= 256 = NT)
  Best mine = {-INFINITY, 0x7fffffff};
  if (tid < K * K) mine = ws[((int64_t)b * K + tid / K) * KMAXW + tid % K];
  for (int r = 0; r < K; ++r) {
    const Best w = block_best(mine, red);
    if (w.i == mine.i && w.i != 0x7fffffff) mine = Best{-INFINITY, 0x7fffffff};      // (flat indices are unique: exactly one owner)
    if (tid == 0) { s_val[r] = w.v; s_idx[r] = w.i == 0x7fffffff ? 0 : w.i; }
  }
  __syncthreads();
  for (int e = tid; e < K * S; e += NT) {
    const int j = e / S, pos = e - j * S, src = s_idx[j] / Vt;
    int64_t v = s_seq[src * S + pos];
    if (pos == t + 1) v = s_idx[j] % Vt;
    seqs[(int64_t)b * K * S + e] = v;
  }
  int ndone = 0;
  if (tid < K) {
    const int src = s_idx[tid] / Vt, tok = s_idx[tid] % Vt;
    cum[b * K + tid] = s_cum[src] + s_val[tid];
    const int d = (t + 1 < S) ? (tok == eos ? 1 : 0) : 1;
    done[b * K + tid] = (unsigned char)d;
    if (prev_pos) prev_pos[b * K + tid] = b * K + src;
    ndone = d;
  }
  if (ctl) {
    if (tid < 64) {
      ndone = (int)wave_sum((float)ndone);
      if (tid == 0) {
        atomicAdd(ctl + 3, ndone);
        __threadfence();
        const int arrived = atomicAdd(ctl + 2, 1);
        if (arrived == B - 1) {
          const int total = atomicAdd(ctl + 3, 0);
          ctl[0] = t + 1;
          if (total == B * K || t + 1 >= S) ctl[1] = 1;
          ctl[2] = 0; ctl[3] = 0;
          __threadfence();
        }
      }
    }
  }
}

}  // namespace

extern "C" int64_t sam_beam_step_ws_bytes(int B, int K) { return (int64_t)B * K * KMAXW * (int64_t)sizeof(Best); }

extern "C" int sam_beam_step_split(const float* fixed_scores, int64_t ld_fixed, const float* ocr_scores, int64_t ld_ocr, int B, int K, int S, int V, int No, int eos, int t,
                                   int32_t* ctl, float* cum, uint8_t* done, int64_t* seqs, int64_t* prev_pos, void* ws, void* stream) {
  SAM_REQUIRE(fixed_scores && ocr_scores && cum && done && seqs && ws, "sam_beam_step_split: null pointer");
  SAM_REQUIRE(B > 0 && K >= 1 && K <= 16 && S >= 1 && S <= 64 && V > 0 && No >= 0, "sam_beam_step: need 1 <= beam size <= 16, 1 <= decoding steps <= 64");
  SAM_REQUIRE(eos >= 0 && eos < V + No && (ctl || (t >= 0 && t < S)), "sam_beam_step: EOS index / step out of range");
  hipStream_t st = (hipStream_t)stream;
  Best* w = (Best*)ws;
  if (K <= 4) beam_scan_kernel<4><<<dim3(B * K), dim3(NT), 0, st>>>(fixed_scores, ld_fixed, ocr_scores, ld_ocr, V, No, S, K, eos, t, ctl, cum, done, w);
  else if (K <= 8) beam_scan_kernel<8><<<dim3(B * K), dim3(NT), 0, st>>>(fixed_scores, ld_fixed, ocr_scores, ld_ocr, V, No, S, K, eos, t, ctl, cum, done, w);
  else beam_scan_kernel<16><<<dim3(B * K), dim3(NT), 0, st>>>(fixed_scores, ld_fixed, ocr_scores, ld_ocr, V, No, S, K, eos, t, ctl, cum, done, w);
  beam_merge_kernel<<<dim3(B), dim3(NT), 0, st>>>(w, V + No, S, K, B, eos, t, ctl, cum, done, seqs, prev_pos);
  SAM_LAUNCH_CHECK();
  return SAM_OK;
}

extern "C" int sam_greedy_pick(const float* fixed_scores, int64_t ld_fixed, const float* ocr_scores, int64_t ld_ocr, int R, int S, int V, int No, int64_t* prev_inds,
                               void* stream) {
  SAM_REQUIRE(fixed_scores && ocr_scores && prev_inds, "sam_greedy_pick: null pointer");
  SAM_REQUIRE(R > 0 && S > 0 && V > 0 && No >= 0, "sam_greedy_pick: empty problem");
  greedy_pick_kernel<<<dim3(R * S), dim3(NT), 0, (hipStream_t)stream>>>(fixed_scores, ld_fixed, ocr_scores, ld_ocr, V, No, S, prev_inds);
  SAM_LAUNCH_CHECK();
  return SAM_OK;
}

extern "C" int sam_beam_step(const float* fixed_scores, int64_t ld_fixed, const float* ocr_scores, int64_t ld_ocr, int B, int K, int S, int V, int No, int eos, int t,
                             int32_t* ctl, float* cum, uint8_t* done, int64_t* seqs, int64_t* prev_pos, void* stream) {
  SAM_REQUIRE(fixed_scores && ocr_scores && cum && done && seqs, "sam_beam_step: null pointer");
  SAM_REQUIRE(B > 0 && K >= 1 && K <= 16 && S >= 1 && S <= 64 && V > 0 && No >= 0, "sam_beam_step: need 1 <= beam size <= 16, 1 <= decoding steps <= 64");
  SAM_REQUIRE(eos >= 0 && eos < V + No && (ctl || (t >= 0 && t < S)), "sam_beam_step: EOS index / step out of range");
  hipStream_t st = (hipStream_t)stream;
  if (K <= 4) beam_step_kernel<4><<<dim3(B), dim3(NT), 0, st>>>(fixed_scores, ld_fixed, ocr_scores, ld_ocr, V, No, S, K, B, eos, t, ctl, cum, done, seqs, prev_pos);
  else if (K <= 8) beam_step_kernel<8><<<dim3(B), dim3(NT), 0, st>>>(fixed_scores, ld_fixed, ocr_scores, ld_ocr, V, No, S, K, B, eos, t, ctl, cum, done, seqs, prev_pos);
  else beam_step_kernel<16><<<dim3(B), dim3(NT), 0, st>>>(fixed_scores, ld_fixed, ocr_scores, ld_ocr, V, No, S, K, B, eos, t, ctl, cum, done, seqs, prev_pos);
  SAM_LAUNCH_CHECK();
  return SAM_OK;
}
