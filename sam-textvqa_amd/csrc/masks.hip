// Allow-bitmask builders: the whole mask algebra of the reference collapsed into one bit per
// (batch, head, query, key), built ONCE per batch and reused by every layer's forward and backward.
//
//   sam_mask_bits_prefix_lm    <- MMT.forward's [B,1,N,N] additive mask, sa_m4c.py:805-844 (also TextBert's
//                                 key-padding mask, sa_m4c.py:386-387, with n_dec = 0)
//   sam_mask_bits_from_additive<- any caller-supplied additive [B,1,N,N] mask (module-level drop-in API)
//   sam_mask_bits_spatial      <- SpatialBertSelfAttention's mask build + min-combine, sa_m4c.py:470-552,568:
//                                 relation tensor int8 [B,Noo,Noo,R] (multi-hot, head-minor as the dataset
//                                 emits it) + quadrant zeroing, AND-ed with the base bits
// Bit k of word w of row (b,h,q) <=> key 32*w+k is visible.  Keys >= N always read 0.
#include "common.h"
#include <stdlib.h>

namespace {

__global__ void prefix_lm_kernel(const uint8_t* key_valid, int B, int n_enc, int n_dec, int NW, uint32_t* out) {
  const int N = n_enc + n_dec;
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (int64_t)B * N * NW) return;
  const int w = idx % NW, q = (idx / NW) % N, b = idx / ((int64_t)NW * N);
  uint32_t bits = 0;
  for (int k = 0; k < 32; ++k) {
    const int key = 32 * w + k;
    bool ok = false;
    if (key < n_enc) ok = key_valid[(int64_t)b * n_enc + key] != 0;   // every row sees valid encoder keys
    else if (key < N) ok = (q >= n_enc) && (key <= q);                // decoder keys: causal, decoder rows only
    bits |= (ok ? 1u : 0u) << k;
  }
  out[idx] = bits;
}

__global__ void from_additive_kernel(const float* mask, int B, int N, int NW, uint32_t* out) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (int64_t)B * N * NW) return;
  const int w = idx % NW;
  const int64_t row = idx / NW;  // b*N + q
  uint32_t bits = 0;
  for (int k = 0; k < 32; ++k) {
    const int key = 32 * w + k;
    if (key < N && mask[row * N + key] > -5000.0f) bits |= 1u << k;  // reference masks are exactly 0 / -10000
  }
  out[idx] = bits;
}

// relation tensor already in per-head layout int8 [B,H,N,N] (the format BASELINE.json's north_star names): bit = rel != 0 (& base)
__global__ void from_int8_bhnn_kernel(const int8_t* rel, const uint32_t* base, int B, int H, int N, int NW, uint32_t* out) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (int64_t)B * H * N * NW) return;
  const int w = idx % NW;
  const int64_t row = idx / NW;                 // (b*H + h)*N + q
  const int q = row % N;
  const int64_t b = row / ((int64_t)H * N);
  const int8_t* p = rel + row * N + 32 * w;
  uint32_t bits = 0;
  for (int k = 0; k < 32; ++k)
    if (32 * w + k < N && p[k] != 0) bits |= 1u << k;
  if (base) bits &= base[(b * N + q) * NW + w];
  out[idx] = bits;
}

// quadrant ids follow the reference's 3x3 numbering over (text, obj+ocr, dec) x (text, obj+ocr, dec)
__device__ __forceinline__ int region_of(int x, int T, int n_oo) { return x < T ? 0 : (x < T + n_oo ? 1 : 2); }

__global__ void spatial_kernel(const uint32_t* base, const int8_t* adj, int B, int N, int NW, int T, int n_oo, int R, int H,
                               unsigned quadrant_bits, uint32_t* out) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (int64_t)B * N * NW) return;
  const int w = idx % NW, q = (idx / NW) % N, b = idx / ((int64_t)NW * N);
  const uint32_t base_bits = base[idx];
  const int rq = region_of(q, T, n_oo);
  uint32_t sp[16];
#pragma unroll
  for (int h = 0; h < 16; ++h) sp[h] = 0;
  for (int k = 0; k < 32; ++k) {
    const int key = 32 * w + k;
    if (key >= N) break;
    const int rk = region_of(key, T, n_oo);
    const bool zeroed = (quadrant_bits >> (3 * rq + rk + 1)) & 1u;
    if (zeroed) continue;
    if (rq == 1 && rk == 1) {
      const int8_t* p = adj + (((int64_t)b * n_oo + (q - T)) * n_oo + (key - T)) * R;
      if ((R & 3) == 0) {   // dataset layout: R = 12 relation bytes per pair = 3 aligned dwords
        const uint32_t* pw = reinterpret_cast<const uint32_t*>(p);
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (4 * j < R) {
            const uint32_t d = pw[j];
#pragma unroll
            for (int e = 0; e < 4; ++e) sp[4 * j + e] |= (((d >> (8 * e)) & 0xffu) != 0 ? 1u : 0u) << k;
          }
      } else {
#pragma unroll
        for (int h = 0; h < 16; ++h)
          if (h < R) sp[h] |= (p[h] != 0 ? 1u : 0u) << k;
      }
    } else {
#pragma unroll
      for (int h = 0; h < 16; ++h) sp[h] |= 1u << k;
    }
  }
#pragma unroll
  for (int h = 0; h < 16; ++h)
    if (h < R) out[(((int64_t)b * H + h) * N + q) * NW + w] = base_bits & sp[h];
  // heads >= R ("implicit" heads, sa_m4c.py:488-495) carry no spatial restriction
  for (int h = R; h < H; ++h) out[(((int64_t)b * H + h) * N + q) * NW + w] = base_bits;
}

// The same result with ONE WAVE per (batch, query) row and one key per lane: the row's relation bytes (n_oo x R, contiguous) are read in coalesced
// 12-byte-per-lane pieces and every head's 64 key bits come out of a ballot.  (The kernel above gives each (row, word) to one thread, which walks
// 32 keys on its own: 64 lanes, 64 cache lines per load -- 51 us per batch at B = 64 for 17 MB of relation bytes; this one ~12.)
__global__ __launch_bounds__(256) void spatial_wave_kernel(const uint32_t* base, const int8_t* adj, int B, int N, int NW, int T, int n_oo, int R, int H,
                                                           unsigned quadrant_bits, uint32_t* out) {
  const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= B * N) return;
  const int b = row / N, q = row - b * N;
  const int rq = region_of(q, T, n_oo);
  const uint32_t* brow = base + (int64_t)row * NW;
  for (int c = 0; c * 64 < NW * 32; ++c) {                 // 64 keys per pass = two words of every head; ALL NW words of the row are written (the row
                                                           // stride is 12 words for 257..384 keys: lengths up to 320 leave words 10 and 11 to the padding, which must read 0)
    const int key = c * 64 + lane;
    unsigned bits = 0;                                      // bit h: key visible for head h
    if (key < N) {
      const int rk = region_of(key, T, n_oo);
      const bool zeroed = (quadrant_bits >> (3 * rq + rk + 1)) & 1u;
      if (!zeroed) {
        if (rq == 1 && rk == 1) {
          const int8_t* p = adj + (((int64_t)b * n_oo + (q - T)) * n_oo + (key - T)) * R;
          if ((R & 3) == 0) {
            const uint32_t* pw = reinterpret_cast<const uint32_t*>(p);
            for (int j = 0; 4 * j < R; ++j) {
              const uint32_t d = pw[j];
#pragma unroll
              for (int e = 0; e < 4; ++e) bits |= (((d >> (8 * e)) & 0xffu) != 0 ? 1u : 0u) << (4 * j + e);
            }
          } else {
            for (int h = 0; h < R; ++h) bits |= (p[h] != 0 ? 1u : 0u) << h;
          }
        } else {
          bits = 0xffffu;
        }
      }
    }
    const int w0 = 2 * c;
    const uint32_t b0 = w0 < NW ? brow[w0] : 0u, b1 = w0 + 1 < NW ? brow[w0 + 1] : 0u;
    unsigned long long mine = ~0ull;                         // lane h ends up with head h's 64 key bits (heads >= R: no spatial restriction)
    for (int h = 0; h < R; ++h) {
      const unsigned long long m = __ballot((bits >> h) & 1u);
      if (lane == h) mine = m;
    }
    for (int h = lane; h < H; h += 64) {                     // one store instruction for all heads
      uint32_t* o = out + (((int64_t)b * H + h) * N + q) * NW;
      if (w0 < NW) o[w0] = b0 & (uint32_t)mine;
      if (w0 + 1 < NW) o[w0 + 1] = b1 & (uint32_t)(mine >> 32);
    }
  }
}

// the three padding masks of a batch (int64 0 / non-zero, as the reference's collate emits them) -> the uint8 forms the kernels read, one launch:
// key_valid [B, T + No + Nc] (MMT keys), q8 [B, T] (TextBert keys), ocr8 [B, Nc] (pointer-network columns)
__global__ void pack_masks_kernel(const int64_t* q, int T, const int64_t* obj, int No, const int64_t* ocr, int Nc, int B, uint8_t* key_valid, uint8_t* q8, uint8_t* ocr8) {
  const int n = T + No + Nc;
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= B * n) return;
  const int b = idx / n, c = idx - b * n;
  uint8_t v;
  if (c < T) { v = q[(int64_t)b * T + c] != 0; q8[b * T + c] = v; }
  else if (c < T + No) v = obj[(int64_t)b * No + (c - T)] != 0;
  else { v = ocr[(int64_t)b * Nc + (c - T - No)] != 0; ocr8[b * Nc + (c - T - No)] = v; }
  key_valid[idx] = v;
}

}  // namespace

extern "C" int sam_pack_masks_u8(const int64_t* question_mask, int T, const int64_t* obj_mask, int No, const int64_t* ocr_mask, int Nc, int B, uint8_t* key_valid,
                                 uint8_t* q8, uint8_t* ocr8, void* stream) {
  SAM_REQUIRE(question_mask && obj_mask && ocr_mask && key_valid && q8 && ocr8, "sam_pack_masks_u8: null pointer");
  SAM_REQUIRE(B > 0 && T > 0 && No > 0 && Nc > 0, "sam_pack_masks_u8: bad shape");
  const int total = B * (T + No + Nc);
  pack_masks_kernel<<<dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream>>>(question_mask, T, obj_mask, No, ocr_mask, Nc, B, key_valid, q8, ocr8);
  SAM_LAUNCH_CHECK();
  return SAM_OK;
}

extern "C" int sam_mask_bits_prefix_lm(const uint8_t* key_valid, int B, int n_enc, int n_dec, int NW, uint32_t* out, void* stream) {
  SAM_REQUIRE(key_valid && out, "sam_mask_bits_prefix_lm: null pointer");
  const int N = n_enc + n_dec;
  SAM_REQUIRE(B > 0 && n_enc >= 0 && n_dec >= 0 && N > 0 && NW * 32 >= N, "sam_mask_bits_prefix_lm: bad shape B=%d n_enc=%d n_dec=%d NW=%d", B, n_enc, n_dec, NW);
  const int64_t total = (int64_t)B * N * NW;
  prefix_lm_kernel<<<dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream>>>(key_valid, B, n_enc, n_dec, NW, out);
  SAM_LAUNCH_CHECK();
  return SAM_OK;
}

extern "C" int sam_mask_bits_from_additive(const float* mask, int B, int N, int NW, uint32_t* out, void* stream) {
  SAM_REQUIRE(mask && out, "sam_mask_bits_from_additive: null pointer");
  SAM_REQUIRE(B > 0 && N > 0 && NW * 32 >= N, "sam_mask_bits_from_additive: bad shape");
  const int64_t total = (int64_t)B * N * NW;
  from_additive_kernel<<<dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream>>>(mask, B, N, NW, out);
  SAM_LAUNCH_CHECK();
  return SAM_OK;
}

extern "C" int sam_mask_bits_from_int8_bhnn(const int8_t* rel, const uint32_t* base, int B, int H, int N, int NW, uint32_t* out, void* stream) {
  SAM_REQUIRE(rel && out, "sam_mask_bits_from_int8_bhnn: null pointer");
  SAM_REQUIRE(B > 0 && H > 0 && N > 0 && NW * 32 >= N, "sam_mask_bits_from_int8_bhnn: bad shape");
  const int64_t total = (int64_t)B * H * N * NW;
  from_int8_bhnn_kernel<<<dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream>>>(rel, base, B, H, N, NW, out);
  SAM_LAUNCH_CHECK();
  return SAM_OK;
}

extern "C" int sam_mask_bits_spatial(const uint32_t* base, const int8_t* adj, int B, int N, int NW, int T, int n_oo, int R, int H,
                                     unsigned quadrant_bits, uint32_t* out, void* stream) {
  SAM_REQUIRE(base && adj && out, "sam_mask_bits_spatial: null pointer");
  SAM_REQUIRE(B > 0 && N > 0 && NW * 32 >= N && T >= 0 && n_oo > 0 && T + n_oo <= N, "sam_mask_bits_spatial: bad shape N=%d T=%d n_oo=%d", N, T, n_oo);
  SAM_REQUIRE(R >= 1 && R <= 16 && H >= R, "sam_mask_bits_spatial: need 1 <= R <= 16 and H >= R (R=%d H=%d)", R, H);
  // legal quadrant ids are 1,2,4,7,8,9 (sa_m4c.py:505-549 raises ValueError on 3,5,6)
  SAM_REQUIRE((quadrant_bits & ~((1u << 1) | (1u << 2) | (1u << 4) | (1u << 7) | (1u << 8) | (1u << 9))) == 0, "sam_mask_bits_spatial: illegal quadrant id in 0x%x", quadrant_bits);
  const int64_t total = (int64_t)B * N * NW;
  static int wave_form = -1;
  if (wave_form < 0) { const char* e = getenv("SAM_MASK_SPATIAL_WAVE"); wave_form = e ? atoi(e) : 1; }
  if (wave_form) spatial_wave_kernel<<<dim3((unsigned)((B * N + 3) / 4)), dim3(256), 0, (hipStream_t)stream>>>(base, adj, B, N, NW, T, n_oo, R, H, quadrant_bits, out);
  else spatial_kernel<<<dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream>>>(base, adj, B, N, NW, T, n_oo, R, H, quadrant_bits, out);
  SAM_LAUNCH_CHECK();
  return SAM_OK;
}
