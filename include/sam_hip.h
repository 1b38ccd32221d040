/* sam_hip.h — C ABI of libsam_hip.so: the MI355X (gfx950) hot path of SA-M4C training.
 *
 * The reference (yashkant/sam-textvqa) has no operator/FFI layer: its hot path is eager PyTorch inside
 * sam/sa_m4c.py.  Each entry point below replaces one eager-op cluster of that file (cited per function,
 * paths relative to /root/reference) and is what a maintainer would bind from Python (ctypes stub in
 * INTEGRATION.md).  Conventions:
 *   - every pointer is a DEVICE pointer owned by the caller; nothing is retained after the call returns;
 *   - bf16 tensors are raw uint16 storage, row-major, leading dimensions in ELEMENTS and multiples of 8;
 *   - calls enqueue on `stream` (a hipStream_t; NULL = default stream) and never synchronise;
 *   - return 0 on success, a negative SAM_ERR_* for rejected arguments, a positive hipError_t for launch
 *     failures; sam_last_error() gives the thread-local message.  No exceptions cross the boundary.
 *   - re-entrant; the process-global state is: one lazily-set function attribute per kernel, the CU count of each device (cached on first use),
 *     and the optional device-side dropout state installed by sam_set_rng_state (below) -- a pointer every later launch of the process reads.
 *   - dropout everywhere is COUNTER-BASED (no generator object): a 32-bit integer hash ("lowbias32" finaliser, csrc/common.h) of
 *     (seed, offset, row, column group); forward and backward regenerate identical bits from the same (seed, offset).  It is not Philox and
 *     not torch's stream: masks are compared with the oracle through the exported keep bits / regenerated masks, never bit-for-bit with torch.
 */
#ifndef SAM_HIP_H
#define SAM_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define SAM_ERR_ARG (-1)
#define SAM_ERR_UNSUPPORTED (-2)

int sam_abi_version(void);
/* sha256 of the sources + compile flags this binary was built from (the Python binding refuses a library whose digest differs from the tree's) */
const char* sam_build_digest(void);
const char* sam_last_error(void);
int sam_device_info(int* cu_count, int* lds_per_cu_bytes, char* arch, int arch_len);
/* CUs withheld from every persistent grid of the training step (GEMMs, grouped weight gradient, one-pass attention backward): n is rounded down to a multiple of
 * 8 (each XCD gives up the same number), 0 <= n <= 128; default SAM_CU_RESERVE from the environment, else 0.  Replaces nothing in the reference -- its
 * nn.DataParallel (train.py:111-112) has no kernel whose grid is sized from the device; here every hot kernel is, and RCCL's channel kernels need CUs beside them
 * (DESIGN.md section 6).  Launches already captured into a hipGraph keep the grid they were captured with. */
int sam_set_cu_reserve(int n);
int sam_get_cu_reserve(void);
/* measurement aid (tools/bench_cu_reserve.py): `blocks` workgroups that each hold one CU's LDS for `microseconds` -- a stand-in for a collective's channel kernel */
int sam_debug_cu_hog(int blocks, double microseconds, void* stream);

/* ---- allow-bit masks (built once per batch; bit k of word w of row (b,h,q) <=> key 32w+k visible) ---- */
/* words per mask row for sequence length N (= padded keys / 32); -1 if N exceeds the fused kernel (384) */
int sam_attn_words_per_row(int N);
/* MMT.forward prefix-LM + causal decoder mask, sam/sa_m4c.py:805-844; TextBert padding mask :386-387 (n_dec=0).
 * key_valid u8 [B,n_enc] -> out u32 [B,1,n_enc+n_dec,NW] */
/* question / object / OCR padding masks (int64, 0 = padded: the reference's batch_dict entries question_mask, pad_obj_mask, pad_ocr_mask) -> uint8:
 * key_valid [B, T+No+Nc] (the MMT's key-padding row, sa_m4c.py:805-812), q8 [B,T] (TextBert's, :386), ocr8 [B,Nc] (OcrPtrNet's, :889).  One launch. */
int sam_pack_masks_u8(const int64_t* question_mask, int T, const int64_t* obj_mask, int No, const int64_t* ocr_mask, int Nc, int B, uint8_t* key_valid,
                      uint8_t* q8, uint8_t* ocr8, void* stream);
int sam_mask_bits_prefix_lm(const uint8_t* key_valid, int B, int n_enc, int n_dec, int NW, uint32_t* out, void* stream);
/* any additive [B,1,N,N] fp32 mask (0 / -10000) as SpatialBertLayer.forward receives it, sam/sa_m4c.py:453-455 */
int sam_mask_bits_from_additive(const float* mask, int B, int N, int NW, uint32_t* out, void* stream);
/* SpatialBertSelfAttention mask build + min-combine, sam/sa_m4c.py:470-552,568.  adj int8 [B,n_oo,n_oo,R] multi-hot
 * (head-minor, as sam/datasets/textvqa_dataset.py:378-409 emits it); quadrant_bits: bit q set <=> quadrant id q in
 * attention_mask_quadrants (legal ids 1,2,4,7,8,9).  base u32 [B,1,N,NW] -> out u32 [B,H,N,NW] */
int sam_mask_bits_spatial(const uint32_t* base, const int8_t* adj, int B, int N, int NW, int T, int n_oo, int R, int H,
                          unsigned quadrant_bits, uint32_t* out, void* stream);

/* relation-type tensor already expanded per head over the WHOLE sequence, int8 [B,H,N,N] (non-zero = visible; the layout
 * BASELINE.json's north_star names): out = bits(rel) & base (base may be NULL).  Equivalent input to sam_mask_bits_spatial. */
int sam_mask_bits_from_int8_bhnn(const int8_t* rel, const uint32_t* base, int B, int H, int N, int NW, uint32_t* out, void* stream);

/* ---- spatial relation graph, sam/spatial_utils.py:92-218 + :33-52 + sam/datasets/textvqa_dataset.py:378-409 ----
 * boxes f64 [B,N,4] normalised xyxy (all-zero row = padding) -> multi-hot int8 [B,N,N,12] for spatial context c in {1,3,5,7,9};
 * one thread per ordered pair, float64 in the reference's operation order. */
int sam_spatial_relation_tensor(const double* boxes, int B, int N, int context, double distance_threshold, int8_t* out, void* stream);

/* ---- fused attention, sam/sa_m4c.py:563-598 (+ the plain BertSelfAttention of 'n' layers / TextBert) ----
 * qkv bf16 [B*N, 3*H*64] (q|k|v, straight out of the fused QKV projection); allow as above with element strides
 * (allow_stride_h = 0 broadcasts one mask over heads); out bf16 [B*N, H*64]; lse2 f32 [B,H,N] = log2-domain
 * logsumexp of scale*q.k (+inf for fully masked rows, whose output is exactly 0 as sa_m4c.py:574-584);
 * keep u32 [B,H,N,NW] receives the dropout keep bits when p_drop > 0 (counter hash of (seed, offset, row, 32-key word)). */
int sam_attn_fwd(const void* qkv, const uint32_t* allow, int64_t allow_stride_b, int64_t allow_stride_h, int B, int N, int H,
                 int head_dim, float scale, float p_drop, uint64_t seed, uint64_t offset, void* out, float* lse2,
                 uint32_t* keep, void* stream);
/* inference variant for the greedy decoding loop (sam/sa_m4c.py:285-302): only query rows >= q_begin (rounded down to a multiple of
 * 16) are computed, against the keys/values of ALL rows of qkv; other rows of out / lse2 are left untouched.  With the prefix-LM mask
 * encoder rows never see decoder keys, so their q/k/v (and outputs) are step-invariant and stay cached in qkv / out. */
int sam_attn_fwd_rows(const void* qkv, const uint32_t* allow, int64_t allow_stride_b, int64_t allow_stride_h, int B, int N, int H,
                      int head_dim, float scale, int q_begin, void* out, float* lse2, void* stream);
/* decoding step with the decoder rows in their own buffer: qkv_enc bf16 [B*N, 3*H*64] is the cache of a full pass (only its first N - n_dec rows per
 * sample are read: the text / object / OCR rows, step-invariant under the prefix-LM mask); qkv_dec bf16 [B*n_dec, 3*H*64] holds this step's q|k|v of
 * the decoder rows, straight out of their QKV projection (no copy into the cache); out_dec bf16 [B*n_dec, H*64] receives the attention output of
 * the decoder rows only.  Same arithmetic per row as sam_attn_fwd (the results are bit-identical to a full pass over the same values). */
int sam_attn_fwd_dec(const void* qkv_enc, const void* qkv_dec, const uint32_t* allow, int64_t allow_stride_b, int64_t allow_stride_h, int B, int N, int n_dec,
                     int H, int head_dim, float scale, void* out_dec, void* stream);
/* the same step for beam search (sam/beam_search.py:31-82 repeats every sample beam_size times; sam/sa_m4c.py:304-314 then runs full forwards over the
 * copies): the `group` beams of a sample have the SAME text / object / OCR rows in every layer, so qkv_enc [B/group * N, 3*H*64] and the allow words
 * (sample stride allow_stride_b) are kept ONCE per sample and decoder sample b reads those of sample b / group; qkv_dec / out_dec have B = samples x group
 * decoder blocks.  group = 1 is sam_attn_fwd_dec. */
int sam_attn_fwd_dec_shared(const void* qkv_enc, const void* qkv_dec, const uint32_t* allow, int64_t allow_stride_b, int64_t allow_stride_h, int B, int group,
                            int N, int n_dec, int H, int head_dim, float scale, void* out_dec, void* stream);
/* ... and for ONE decoder row: out[b, :] (bf16, row stride ldo) = the attention output of decoder row t of decoder sample b against its sample's
 * first N - n_dec cached rows and its own decoder rows 0..t (qkv_dec, which holds row t's q|k|v already).  What a beam-search step needs when the
 * earlier decoder rows of a beam are kept and re-gathered instead of recomputed; one block per (sample, head) stages the shared keys / values once for
 * all `group` beams.  fp32 arithmetic on the bf16 cache values; a row with no allowed key gives zeros. */
int sam_attn_dec_row(const void* qkv_enc, const void* qkv_dec, const uint32_t* allow, int64_t allow_stride_b, int64_t allow_stride_h, int B, int group, int N,
                     int n_dec, int t, int H, int head_dim, float scale, void* out, int64_t ldo, void* stream);
/* training forward: sam_attn_fwd plus out_lo bf16 [B*N, H*64] = bf16(out_exact - bf16(out_exact)), the rounding residual of the output.  The one-pass
 * backward takes delta = rowsum(dO * O) from out + out_lo (exact to 2^-17; the bf16 output alone costs 3e-3 of max in dQ / dK), which is what lets it
 * compute every score once instead of running a row pass for delta first.  Replaces sam/sa_m4c.py:563-598 as sam_attn_fwd does. */
int sam_attn_fwd_train(const void* qkv, const uint32_t* allow, int64_t allow_stride_b, int64_t allow_stride_h, int B, int N, int H,
                       int head_dim, float scale, float p_drop, uint64_t seed, uint64_t offset, void* out, void* out_lo, float* lse2,
                       uint32_t* keep, void* stream);
/* autograd of sam/sa_m4c.py:563-598 in ONE pass (N <= sam_attn_bwd_fused_max_n() = 384 keys: up to 192 one sub-problem per (batch, head), 193..384 as 2 x 2
 * sub-problems of 192 -- the "long" kernel; SAM_ERR_UNSUPPORTED beyond, sam_attn_bwd is the two-kernel form): one workgroup
 * per (batch, head) stages Q, K, dO once (block-scaled fp16), computes S, P, dP and dS once per score and uses them for dK, dV (in registers) and dQ
 * (through an LDS exchange of dS).  out / out_lo: the forward's output and residual (sam_attn_fwd_train); keep: its dropout bits (NULL when p_drop = 0). */
int sam_attn_bwd_fused(const void* dout, const void* qkv, const void* out, const void* out_lo, const float* lse2, const uint32_t* allow,
                       int64_t allow_stride_b, int64_t allow_stride_h, const uint32_t* keep, int B, int N, int H, int head_dim,
                       float scale, float p_drop, void* dqkv, void* stream);
int sam_attn_bwd_fused_max_n(void);
/* autograd of sam_attn_fwd, two-kernel form (any N <= 384): dout bf16 [B*N,H*64] -> dqkv bf16 [B*N,3*H*64]; delta_ws f32 [B,H,N] scratch */
int sam_attn_bwd(const void* dout, const void* qkv, const float* lse2, const uint32_t* allow, int64_t allow_stride_b,
                 int64_t allow_stride_h, const uint32_t* keep, int B, int N, int H, int head_dim, float scale, float p_drop,
                 void* dqkv, float* delta_ws, void* stream);

/* ---- bf16 MFMA GEMM with fused epilogues: the nn.Linear sites of the path ----
 * C[M,N] = epilogue( sum_k A(m,k) * B(k,n) ), fp32 accumulate.
 *   a_kcontig=1: A stored [M][lda] (k contiguous)   a_kcontig=0: A stored [K][lda] (m contiguous)
 *   b_kcontig=1: B stored [N][ldb] (nn.Linear weight) b_kcontig=0: B stored [K][ldb] (n contiguous)
 *   forward y = x W^T + b : (1,1);  dgrad dx = dy W : (1,0);  wgrad dW = dy^T x : (0,0) with A=dy, B=x swapped roles.
 * Epilogues (sam/sa_m4c.py call sites; BertSelfOutput/BertIntermediate/BertOutput are pytorch-transformers):
 *   SAM_EPI_NONE              C = acc (+ C when accumulate, fp32 C only)
 *   SAM_EPI_BIAS              C = acc + bias[n]                                   query/key/value :554-556, classifier :275
 *   SAM_EPI_BIAS_GELU         aux_out = acc + bias ; C = gelu_erf(aux_out)        BertIntermediate via :678
 *   SAM_EPI_BIAS_DROPOUT_RES  C = dropout(acc + bias) + residual[m,n]             BertSelfOutput :653 / BertOutput :680 (pre-LN)
 *   SAM_EPI_DGELU             C = acc * gelu_erf'(aux_in[m,n])                    backward of BertIntermediate
 *   SAM_EPI_BIAS_GELU_GRAD    aux_out = gelu_erf'(acc + bias) (skipped when aux_out is NULL: inference) ; C = gelu_erf(acc + bias)     BertIntermediate in TRAINING: the derivative shares the
 *                             exponential with the activation (two extra FMAs); the backward then needs no transcendental at all:
 *   SAM_EPI_MUL_AUX           C = acc * aux_in[m,n]                               backward of BertIntermediate from the stored derivative
 * bias may be NULL (treated as 0); dropout draws 8 x 16 bits per (row, col/8) from the counter hash so the backward regenerates it. */
enum { SAM_EPI_NONE = 0, SAM_EPI_BIAS = 1, SAM_EPI_BIAS_GELU = 2, SAM_EPI_BIAS_DROPOUT_RES = 3, SAM_EPI_DGELU = 4, SAM_EPI_BIAS_GELU_GRAD = 5, SAM_EPI_MUL_AUX = 6 };
/* Optional LayerNorm behind a GEMM whose epilogue is SAM_EPI_BIAS_DROPOUT_RES with a bf16 output (BertSelfOutput / BertOutput, sam/sa_m4c.py:653,680 +
 * 1016-1028: LayerNorm(dropout(x W^T + b) + residual)).  When the library runs such a GEMM split over K (skinny M, long K), the pass that sums the partials
 * and applies the epilogue owns whole rows and normalises them on the spot: C receives the pre-LayerNorm sums as ever (the backward reads them), y / mean /
 * rstd what sam_layernorm_fwd would have produced from C, bit for bit; `done` is set to 1.  In every other case `done` is set to 0 and the caller runs
 * sam_layernorm_fwd itself.  N % 4 == 0, N <= 2048. */
typedef struct sam_ln_fuse {
  const float* gamma; const float* beta; float eps;
  void* y; int64_t ldy;          /* bf16 [M, N] */
  float* mean; float* rstd;      /* fp32 [M] */
  int32_t done;                  /* OUT */
  /* Round 6 -- the LayerNorm INSIDE an MMT-size launch (SURVEY 8(b) linear_bias_dropout_residual_ln as one kernel; sa_m4c.py:653, 680): with a workspace of
   * sam_gemm_ln_ws_bytes(M, N) bytes here (ZERO-FILLED once by the caller, private to one stream; every launch leaves its counters zero again; word 0 is an error
   * word raised when a bounded wait ran out) a product that the loader-wave kernel covers in ONE round of tiles normalises its rows itself: the waves / blocks that
   * share a row exchange (mean, M2) pairs through the workspace.  y / mean / rstd then equal sam_layernorm_fwd(C) up to fp32 rounding of the statistics
   * (the split-K form above is bit-identical).  NULL: never.  OPT-IN (SAM_GEMM_LN_FUSE=1): correct, and measured 2.4x slower than sam_gemm_bf16 + sam_layernorm_fwd
   * (a row's statistics cross XCDs: memory-side round trips on an otherwise idle CU -- profiles/r6_gemm_experiments.txt #16). */
  float* xws; int64_t xws_bytes;
} sam_ln_fuse;
int64_t sam_gemm_ln_ws_bytes(int M, int N);
typedef struct sam_gemm_desc {
  int32_t M, N, K;
  int32_t a_kcontig, b_kcontig;
  int32_t c_is_f32, accumulate, epilogue;
  const void* A; int64_t lda;
  const void* B; int64_t ldb;
  void* C; int64_t ldc;
  const float* bias;
  const void* residual; int64_t ldr;
  void* aux_out; const void* aux_in; int64_t ld_aux;
  float p_drop; uint64_t seed, offset;
  int32_t split_k;    /* 0/1: none; >1: split the K loop over that many workgroups per tile; -1: auto (bounded by ws_bytes; may decide 1).
                         Partials go to `ws` and are summed in a fixed order (bit-reproducible, no atomics).  Two forms:
                         fp32 C + accumulate=1 + SAM_EPI_NONE (wgrad: the sum is added INTO C, bias_grad allowed), or any other
                         epilogue / output with accumulate=0 (skinny M, long K: the epilogue is applied by the reduction pass). */
  float* bias_grad;   /* wgrad layout (0,0) only: bias_grad[m] (+)= sum_k A(m,k), i.e. the bias gradient colsum(dy), fused into the wgrad;
                         added to the old value when `accumulate`, overwritten otherwise (like C). */
  float* ws; int64_t ws_bytes;   /* split-K scratch: split_k * (M*N + M) floats */
  int32_t force_tile; /* 0: heuristic; 64 / 128 / 160 / 192 / 256: force that block-tile height of the 4-wave kernels; 1192 / 1256 / 1448 (1128): force the 8-wave
                         persistent kernel with 192x192 / 256x256 / 192x256 (128x128, two blocks per CU) tiles; 12192 / 12448: force the 12-wave kernel with loader
                         waves (192x192 three-stage / 192x256) -- testing, tuning.  Grouped call: 0, 128 (4-wave), 1256 (8-wave pair exchange) or 12448 (loader-wave
                         kernel with the whole-tile + K-slice schedule) in descs[0] */
  int32_t defer_reduce;  /* split-K only: 1 = leave the partials in ws and let the caller run sam_gemm_splitk_reduce (separately timeable) */
  int32_t split_k_used;  /* OUT: the split factor that was launched (1 = no split, nothing to reduce) */
  sam_ln_fuse* ln;       /* optional (NULL = none): see sam_ln_fuse */
} sam_gemm_desc;
int sam_gemm_bf16(const sam_gemm_desc* d, void* stream);
/* up to 12 (20 when the 8-wave kernel takes the set: see the workspace note below) independent wgrad-layout problems (a_kcontig = b_kcontig = 0, fp32 C,
 * SAM_EPI_NONE, no split) in ONE grid -- problems of different depth K may be mixed: the tiles of the deepest ones are dispatched first, one per CU, the
 * shallow ones fill the CUs that round leaves idle (an MMT layer pair, 216 tiles x 182 k-tiles, together with TextBert's 324 tiles x 20).  E.g. the four
 * weight gradients of an encoder layer (dWqkv, dWo, dW1, dW2: 432 tiles of 128x128) fill the 512 resident block slots in a single
 * round, which makes split-K and its reduction pass unnecessary.  bias_grad is honoured per problem. */
int sam_gemm_bf16_grouped(const sam_gemm_desc* descs, int count, void* stream);
/* Workspace of the grouped call.  With descs[0].ws / ws_bytes >= this many bytes (16-byte aligned, ZERO-FILLED once by the caller, private to one
 * stream; every launch leaves its flag words zero again) and every K a multiple of 64, the call runs 256x256 tiles on the 8-wave kernel with each
 * tile's K range split over a PAIR of workgroups that exchange accumulator halves inside the launch (fixed summation order, no atomics).
 * Without it the 128x128 4-wave kernel runs.  descs[0].force_tile: 0 = choose, 128 = 4-wave kernel, 1256 = 8-wave kernel or error, 12448 = the loader-wave
 * kernel (gemm12w.hip; opt-in, also by SAM_GEMM12W=1: measured slower on the step's sets) or error; the size returned below covers whichever kernel runs.
 * The first 32-bit word of the workspace is an ERROR word: the pair wait is bounded (~1 s), a block whose partner never became resident raises it
 * and finishes with an undefined result instead of hanging the device; a caller that reads a non-zero word there must discard that step. */
int64_t sam_gemm_grouped_ws_bytes(const sam_gemm_desc* descs, int count);
/* C[m,n] += sum_s ws[s][m,n] ; bias_grad[m] += sum_s ws_bias[s][m]  (ws layout as written by sam_gemm_bf16; fixed order) */
int sam_gemm_splitk_reduce(const float* ws, int split_k, int M, int N, float* C, int64_t ldc, float* bias_grad, void* stream);

/* ---- BertLayerNorm, sam/sa_m4c.py:1016-1028 (TF style, eps inside the sqrt, biased variance) ----
 * x [M,D] bf16 or fp32 (x_is_f32) -> y bf16, plus per-row mean / rstd (fp32) for the backward. D % 4 == 0, D <= 2048. */
int sam_layernorm_fwd(const void* x, int x_is_f32, int64_t ldx, const float* gamma, const float* beta, float eps, int M, int D, void* y,
                      int64_t ldy, float* mean, float* rstd, void* stream);
/* backward of y = LN(x): dx bf16 [M,D]; when dx_dropped != NULL also writes dropout(dx) with the SAME counter-hash
 * (row, col/8) stream as SAM_EPI_BIAS_DROPOUT_RES used in the forward (the gradient of the dense in front of the
 * residual add); dgamma/dbeta/dbias fp32 [D] (dbias = column sums of the dropped dx; may be NULL); accumulate: += .
 * ws: sam_layernorm_bwd_ws_bytes(D) bytes of scratch. Deterministic two-stage reductions (no atomics). */
int64_t sam_layernorm_bwd_ws_bytes(int D);
int sam_layernorm_bwd(const void* dy, int64_t ldd, const void* x, int x_is_f32, int64_t ldx, const float* mean, const float* rstd,
                      const float* gamma, int M, int D, void* dx, void* dx_dropped, int64_t ldo, float p_drop, uint64_t seed, uint64_t offset,
                      float* dgamma, float* dbeta, float* dbias, int accumulate, float* ws, void* stream);
/* Deferred finalize: with bit 2 set in `accumulate` (accumulate | 4) sam_layernorm_bwd leaves its per-block partial sums in ws (which must then be
 * private to the call and stay alive) and launches no reduction; sam_layernorm_bwd_finalize_batch finishes up to any number of such calls in one
 * launch per 32 (dgamma / dbeta / dbias (+)= column sums of the partial rows, fixed order).  rows = sam_layernorm_bwd_partial_rows(M). */
typedef struct sam_ln_finalize_item {
  const float* ws; int32_t rows; int32_t accumulate;
  float* dgamma; float* dbeta; float* dbias;   /* dbias may be NULL */
} sam_ln_finalize_item;
int sam_layernorm_bwd_partial_rows(int M);
int sam_layernorm_bwd_finalize_batch(const sam_ln_finalize_item* items, int count, int D, void* stream);
/* bias gradients: out[n] (+)= sum_m x[m,n], x bf16 [M,N]; ws: sam_colsum_ws_bytes(N) */
int64_t sam_colsum_ws_bytes(int N);
int sam_colsum_bf16(const void* x, int64_t ldx, int M, int N, float* out, int accumulate, float* ws, void* stream);
/* ---- element-wise dropout of the object / OCR input encoders, sam/sa_m4c.py:224,263 (F.dropout on LN(feat W) + LN(bbox W)) ----
 * out = dropout(a + b), bf16 [M,D]; b may be NULL (plain dropout: the backward applies the same mask to dy).  Mask = the hidden-state dropout
 * stream on (row, col/8) under (seed, offset) [+ the device-side RNG state, sam_set_rng_state]; keep probability quantised to 16 bits as in
 * SAM_EPI_BIAS_DROPOUT_RES.  D and the row strides must be multiples of 8. */
int sam_add_dropout_bf16(const void* a, int64_t lda, const void* b, int64_t ldb, void* out, int64_t ldo, int M, int D, float p_drop, uint64_t seed,
                         uint64_t offset, void* stream);

/* ---- M4CDecodingBCEWithMaskLoss, sam/task_utils.py:19-30: forward value AND analytic gradient in one pass ----
 * scores arrive as the two blocks the model produces (classifier logits [R,V] and pointer scores [R,No], both fp32,
 * R = B*S decoding rows); loss = sum(bce * mask[r]) / max(sum(mask),1); d_fixed bf16 [R,V], d_ocr fp32 [R,No], both
 * already multiplied by grad_scale.  global_count (device scalar, may be NULL): data-parallel runs pass the all-reduced number of unmasked
 * decoding steps of the GLOBAL batch; the normaliser is then max(global_count, 1) as under the reference's nn.DataParallel (train.py:111-112
 * gathers the scores before the loss), so the SUM of the ranks' losses / gradients is the global-batch loss / gradient. */
int sam_bce_loss(const float* fixed_scores, int64_t ld_fixed, const float* ocr_scores, int64_t ld_ocr, const float* targets, int64_t ld_t,
                 const float* loss_mask, int R, int V, int No, float grad_scale, const float* global_count, float* loss, void* d_fixed,
                 int64_t ld_dfixed, float* d_ocr, int64_t ld_docr, void* stream);

/* ---- OcrPtrNet bilinear scores, sam/sa_m4c.py:891-893: out[b,s,o] = scale*<q[b,s],k[b,o]> + (1-mask[b,o])*-10000 ----
 * q bf16 [B,S,D], k bf16 [B,No,D] (already projected), mask u8 [B,No]; out fp32 with element strides (ld_out_b, ld_out_s) */
int sam_ptr_scores_fwd(const void* q, const void* k, const uint8_t* ocr_mask, int B, int S, int No, int D, float scale, float* out,
                       int64_t ld_out_b, int64_t ld_out_s, void* stream);
int sam_ptr_scores_bwd(const float* dscores, int64_t ld_b, int64_t ld_s, const void* q, const void* k, int B, int S, int No, int D, float scale,
                       void* dq, void* dk, void* stream);

/* ---- word-embedding backward (BertEmbeddings.word_embeddings of TextBert, sam/sa_m4c.py:377,383): grad[idx[t],:] += dy[t,:] ----
 * dy bf16 [T,D]; idx int64 [T]; grad fp32 [rows, ldg]; rows outside [0,rows) and row == padding_idx (nn.Embedding semantics; -1 = none)
 * are skipped.  Rows may repeat: the block of a row's FIRST occurrence is its only writer and adds the duplicates in list order -- no atomics, no sort,
 * bit-reproducible (device-scope fp32 atomics execute at the memory side of the fabric on this part: 89-200 us for 1280 rows; SAM_EMBED_BWD_ATOMIC=1
 * selects that kernel for an A/B, and it is what an unaligned table falls back to). */
int sam_embedding_bwd(const void* dy, int64_t ldd, const int64_t* idx, int T, int D, int rows, int64_t padding_idx, float* grad, int64_t ldg, uint8_t* touched,
                      void* stream);
/* the same sum in a FIXED order for an index list sorted ascending (one writer per table row, no atomics): what the data-parallel row-sparse
 * exchange uses, so that every rank adds up bit-identical gradients from the same gathered list */
int sam_embedding_bwd_sorted(const void* dy, int64_t ldd, const int64_t* idx_sorted, int T, int D, int rows, int64_t padding_idx, float* grad, int64_t ldg,
                             uint8_t* touched, void* stream);
/* `touched` (both forms; may be NULL): uint8 [rows], touched[r] := 1 for every table row that receives a gradient -- the row flags of a
 * sam_sparse_rows region (below), which let the norm and the optimizer skip rows that never have. */

/* ---- front end: feature normalisation + packing, embedding sums, previous-prediction gather (csrc/embed.hip) ----
 * sam_l2norm_pack_bf16: F.normalize(x, dim=-1) (x / max(||x||_2, eps); normalize = 0: plain cast) of fp32 rows [M, D], rounded to bf16 and
 *   written at column col0 of out [M, ldo]; columns [col0 + D, zero_upto) are zeroed.  Replaces the normalize / cat chain of
 *   SAM4C.forward_obj_encoding / forward_ocr_encoding, sam/sa_m4c.py:217-253: the OCR row (FastText 300 | PHOC 604 | FRCN 2048 | 50 zeros)
 *   is packed by three calls into the K-padded GEMM operand.  D, ldx, ldo, col0 multiples of 4.
 * sam_embed_sum_fwd: out[r,:] = table[ids[r],:] (bf16 rows; table may be NULL) + pos[r % S,:] + tt[type_ids[r],:] (type_ids NULL = type 0), fp32 out
 *   = the LayerNorm input of BertEmbeddings.forward (pytorch-transformers, used by TextBert sam/sa_m4c.py:377) and of the position/type
 *   half of PrevPredEmbeddings.forward, sam/sa_m4c.py:932-945.
 * sam_embed_sum_bwd: d_pos[s,:] += sum_b d[b*S+s,:];  d_tt[t,:] += sum_{r: type[r]==t} d[r,:]  (d bf16 [R, D]; deterministic; n_types <= 4;
 *   ws: sam_embed_sum_bwd_ws_bytes).  The table gradient is sam_embedding_bwd.
 * sam_gather2_add_fwd: out[b,s,:] = (ind < V ? ans[ind,:] : ocr[b*n_ocr + ind - V,:]) + dropout(emb[b,s,:])  -- _batch_gather over
 *   cat([ans_emb, ocr_emb]) + the embedding sum of PrevPredEmbeddings.forward, sam/sa_m4c.py:921-948, without the [B, V+n_ocr, D] table.
 *   ans bf16 [V, D], ocr bf16 [B*n_ocr, D], inds int64 [B, S] (clamped to [0, V+n_ocr)), emb bf16 [B*S, D] or NULL; counter-hash dropout on emb.
 * sam_gather2_add_bwd: d_ans[ind,:] += dy / d_ocr[...] += dy (fp32 atomics into pre-zeroed buffers: indices repeat);
 *   d_emb (bf16, may be NULL) = the same dropout mask applied to dy. */
int sam_l2norm_pack_bf16(const float* x, int64_t ldx, int M, int D, int normalize, float eps, void* out, int64_t ldo, int col0, int zero_upto,
                         void* stream);
int sam_embed_sum_fwd(const void* table, int64_t ld_table, const int64_t* ids, int table_rows, const float* pos, int64_t ld_pos, int S,
                      const float* tt, int64_t ld_tt, const uint8_t* type_ids, int n_types, int R, int D, float* out, int64_t ldo, void* stream);
int64_t sam_embed_sum_bwd_ws_bytes(int S, int n_types, int D);
int sam_embed_sum_bwd(const void* d, int64_t ldd, int R, int D, int S, const uint8_t* type_ids, int n_types, float* d_pos, int64_t ld_pos,
                      float* d_tt, int64_t ld_tt, float* ws, void* stream);
int sam_gather2_add_fwd(const void* ans, int64_t ld_ans, int V, const void* ocr, int64_t ld_ocr, int n_ocr, const int64_t* inds, int B, int S, int D,
                        const void* emb, int64_t ld_emb, float p_drop, uint64_t seed, uint64_t offset, void* out, int64_t ldo, void* stream);
int sam_gather2_add_bwd(const void* dy, int64_t ldd, int V, int n_ocr, const int64_t* inds, int B, int S, int D, float* d_ans, int64_t ld_dans,
                        float* d_ocr, int64_t ld_docr, float p_drop, uint64_t seed, uint64_t offset, void* d_emb, int64_t ld_demb, void* stream);

/* ---- tail of the object / OCR input encoders, sam/sa_m4c.py:204-224 and :226-263: out = dropout(LN_a(za) + LN_b(bbox W_b^T + b_b)) (csrc/encoder_in.hip) ----
 * za bf16 [R, D] = the wide feature projection feat W_a^T + b_a (a sam_gemm_bf16 call); bbox fp32 [R, ldbox >= 4]: the four box coordinates, read in
 * place from the batch's pad_{obj,ocr}_bboxes rows and rounded to bf16 as the GEMM operand was; wb bf16 [D, ldw] (the 4 -> D weight, rows padded), bias_b,
 * gamma / beta of the two BertLayerNorms (eps inside the sqrt), hidden-state dropout stream on (row, col/8) under (seed, offset).  ONE launch forward
 * (upstream: six eager ops; the round-2 path: box pack + box GEMM + two LayerNorms + add/dropout).  stats f32 [R, 4] = (mean_a, rstd_a, mean_b, rstd_b).
 * Backward: ONE row pass (dropout mask regenerated, both LayerNorm backwards, z_b recomputed from the boxes) writes d za bf16 [R, D] -- the operand
 * of the wide weight gradient -- and per-block partial column sums; a fixed-order finalize then (+)= d gamma_a, d beta_a, d gamma_b, d beta_b,
 * d bias_b, d wb (fp32 [D, ldgw], columns 0..3).  ws: sam_input_encoder_bwd_ws_bytes(R, D). */
int sam_input_encoder_fwd(const void* za, int64_t ldza, const float* bbox, int64_t ldbox, const void* wb, int64_t ldw, const float* bias_b,
                          const float* gamma_a, const float* beta_a, const float* gamma_b, const float* beta_b, float eps, int R, int D, float p_drop,
                          uint64_t seed, uint64_t offset, void* out, int64_t ldo, float* stats, void* stream);
int64_t sam_input_encoder_bwd_ws_bytes(int R, int D);
int sam_input_encoder_bwd(const void* dy, int64_t ldd, const void* za, int64_t ldza, const float* bbox, int64_t ldbox, const void* wb, int64_t ldw,
                          const float* bias_b, const float* gamma_a, const float* gamma_b, const float* stats, int R, int D, float p_drop, uint64_t seed,
                          uint64_t offset, void* dza, int64_t ldo, float* dgamma_a, float* dbeta_a, float* dgamma_b, float* dbeta_b, float* dbias_b, float* dwb,
                          int64_t ldgw, int accumulate, float* ws, void* stream);

/* ---- optimizer step over ONE flat fp32 parameter buffer: clip_grad_norm_ + Adam, train.py:139-142, task_utils.py:33-57 ----
 * sam_sumsq_f32: out[0] = sum g^2 (deterministic two-stage; every data-parallel rank gets the identical value).
 * sam_adam_step: torch.optim.Adam semantics (bias-corrected, eps outside the sqrt); per-segment learning rates
 * (param groups of SAM4C.get_optimizer_parameters, sa_m4c.py:349-371); gradient pre-scaled by
 * min(1, max_norm / (sqrt(gnorm_sq[0]) + 1e-6)) when gnorm_sq != NULL; also refreshes the bf16 shadow weights. */
/* Row-sparse region (optional, NULL = none): elements [lo, hi) of the flat buffers form rows of row_len elements -- the 30522 x 768 word-embedding
 * table of TextBert, a quarter of all parameters, of which one step touches at most B * 20 rows.  A row whose flag is 0 has NEVER received a gradient:
 * g = exp_avg = exp_avg_sq = 0 there, so torch.optim.Adam's update is exactly zero and its squares add nothing to the norm; such rows are skipped
 * (results bit-identical to the dense pass, 0.7 GB less traffic per step while few rows are in use).  The optimizer clears the gradient of the
 * touched rows after using it: the region is never zero-filled, untouched rows stay zero for ever.  Flags are set by sam_embedding_bwd[_sorted];
 * a caller that installs optimizer state from elsewhere (checkpoint) must set the flags of every row whose state is non-zero. */
typedef struct sam_sparse_rows { int64_t lo, hi; int32_t row_len; const uint8_t* touched; } sam_sparse_rows;
int64_t sam_sumsq_ws_bytes(void);
int sam_sumsq_f32(const float* g, int64_t n, const sam_sparse_rows* sparse, float* out, float* ws, void* stream);
int sam_adam_step(float* p, float* g, float* m, float* v, void* p_bf16, int64_t n, const int64_t* seg_end, const float* seg_lr, int nseg,
                  float beta1, float beta2, float eps, int64_t step, const float* gnorm_sq, float max_norm, const sam_sparse_rows* sparse, void* stream);
/* the same update with the step's schedule in DEVICE memory: dev_sched = [lr of segment 0..nseg-1, 1 - beta1^t, 1 - beta2^t] (fp32).  A launch
 * captured in a hipGraph freezes its by-value arguments; this form lets every replay apply the current learning rates / bias corrections. */
int sam_adam_step_dev(float* p, float* g, float* m, float* v, void* p_bf16, int64_t n, const int64_t* seg_end, int nseg, float beta1, float beta2,
                      float eps, const float* dev_sched, const float* gnorm_sq, float max_norm, const sam_sparse_rows* sparse, void* stream);
/* sam_adam_step_dev restricted to the elements [lo, hi) of the buffers (multiples of 4; the row-sparse region entirely inside or outside), for a step
 * whose update is applied in pieces on several streams: the pending update of step k at the HEAD of captured step k + 1, where the piece the next
 * forward needs first (TextBert, the input encoders) goes out before the rest and the rest runs underneath that forward (same train.py:139-142
 * arithmetic, element for element; seg_end / dev_sched describe the WHOLE buffer).  zero_grad != 0: every gradient element is cleared after use (the
 * step then needs no zero-fill and no ordering between it and the pieces).  gate (may be NULL): device word; 0 = the launch does nothing -- a replay
 * whose pending update the host has already applied (Trainer.flush_update) skips it that way.  max_blocks > 0 caps the grid (a piece that runs
 * underneath latency-bound work must leave wave slots free for it; 0 = the full-speed grid). */
int sam_adam_step_range(float* p, float* g, float* m, float* v, void* p_bf16, int64_t n, const int64_t* seg_end, int nseg, float beta1, float beta2,
                        float eps, const float* dev_sched, const float* gnorm_sq, float max_norm, const sam_sparse_rows* sparse, int64_t lo, int64_t hi,
                        int zero_grad, const int32_t* gate, int max_blocks, void* stream);
/* Per-step state of a hipGraph-captured training step, advanced ON THE DEVICE by the graph's first node (nothing the host writes is read by
 * a replay, so the host may queue replays as far ahead as it likes):
 *   rng_state[1] += offset_stride (fresh dropout masks; rng_state = the array given to sam_set_rng_state, may be NULL);
 *   t = ++step_counter[0];  dev_sched = [base_lr[s] * lambda(t - 1) for s < nseg, 1 - beta1^t, 1 - beta2^t] for sam_adam_step_dev, with
 *   lambda = the LambdaLR of sam/task_utils.py:48-54: linear warm-up from warmup_factor to 1 over warmup_iters steps, then
 *   lr_decay ^ (number of decay_iters <= step).  Arithmetic in double, rounded to fp32 once. */
typedef struct sam_lr_schedule {
  double base_lr[8]; int32_t nseg;
  int64_t warmup_iters; double warmup_factor;
  int32_t n_decay; int64_t decay_iters[4]; double lr_decay;
  double beta1, beta2;
} sam_lr_schedule;
int sam_step_advance(unsigned long long* rng_state, uint64_t offset_stride, int64_t* step_counter, const sam_lr_schedule* sched, float* dev_sched,
                     void* stream);
int sam_cast_f32_to_bf16(const float* x, void* y, int64_t n, void* stream);

/* ---- token selection of the decoding loops (csrc/decode.hip), on the two score blocks the model produces: classifier logits fixed [R*S, V] and
 * pointer scores ocr [R*S, No] (fp32, row strides in elements; the reference's `scores` is their concatenation) ----
 * sam_greedy_pick: sam/sa_m4c.py:299-302 -- prev_inds[r, s + 1] = argmax_j scores[r, s, j] for s < S - 1 (first maximum wins, as torch.argmax).
 * sam_beam_step: one BeamSearch.decode step (sam/beam_search.py:84-160) for B samples of K beams (rows b*K .. b*K+K-1), state updated IN PLACE:
 *   seqs int64 [B*K, S] (train_prev_inds), cum f32 [B*K] (topkscores), done u8 [B*K] (completed_ids as flags; zero before the first step);
 *   prev_pos int64 [B*K] (may be NULL) receives each surviving beam's source row (prev_position).  Candidate score = log(sigmoid(x)) + cum[beam];
 *   completed beams offer only EOS at log-probability 0; at step 0 only beam 0 of a sample is live; the K best of the flattened [K, V+No] axis in
 *   descending order (lower flat index first among equals); source beam = idx / (V+No) -- INTEGER division, the reference's torch <= 1.4 `/` --,
 *   token = idx % (V+No); cum[new] = cum[src] + value (the value already holds cum[src]: counted twice, as upstream).
 *   ctl = NULL: the step index is `t`.  ctl = int32[4] in device memory {t, finished, 0, 0} (zero-filled before the first step): the step index is
 *   ctl[0], advanced by the launch itself, and once every beam is complete (or the steps ran out) ctl[1] is set and later launches leave the state
 *   untouched (prev_pos, when given, then receives the identity) -- what lets captured decoding steps be replayed S - 1 times with the reference's
 *   early exit. */
int sam_greedy_pick(const float* fixed_scores, int64_t ld_fixed, const float* ocr_scores, int64_t ld_ocr, int R, int S, int V, int No, int64_t* prev_inds,
                    void* stream);
int sam_beam_step(const float* fixed_scores, int64_t ld_fixed, const float* ocr_scores, int64_t ld_ocr, int B, int K, int S, int V, int No, int eos, int t,
                  int32_t* ctl, float* cum, uint8_t* done, int64_t* seqs, int64_t* prev_pos, void* stream);
/* the same step as two launches -- the candidate scan over one block per (sample, beam) instead of one per sample, then a merge of the K lists per sample:
 * identical results (same candidates, same order, same tie rule), a quarter of the time at beam 5.  ws: sam_beam_step_ws_bytes(B, K) bytes of scratch. */
int64_t sam_beam_step_ws_bytes(int B, int K);
int sam_beam_step_split(const float* fixed_scores, int64_t ld_fixed, const float* ocr_scores, int64_t ld_ocr, int B, int K, int S, int V, int No, int eos, int t,
                        int32_t* ctl, float* cum, uint8_t* done, int64_t* seqs, int64_t* prev_pos, void* ws, void* stream);

/* ---- greedy decoding steps t_begin .. t_end-1 as ONE persistent launch (csrc/decode_steps.hip; sam/sa_m4c.py:285-302, rows of 294-302's loop) ----
 * After a full first pass (step 0) every later step only has ONE new row per sample: decoder row t, whose input token was picked by step t-1 and
 * which, under the prefix-LM mask (sa_m4c.py:834-844), sees the encoder rows and decoder rows 0..t -- so row t of the reference's last full forward
 * is what step t computes here.  Per step and sample: PrevPredEmbeddings of prev_inds[b, t] (sa_m4c.py:928-948, eval mode), every encoder layer
 * (QKV projection written into row n_enc + t of that layer's q|k|v cache, attention over cache rows 0..n_enc+t under the allow bits, output
 * projection + LayerNorm, FFN + LayerNorm), classifier logits into fixed_scores[b, t, :V] and pointer scores into ocr_scores[b, t, :No]
 * (sa_m4c.py:270-278, 866-897), prev_inds[b, t+1] = argmax over [logits | pointer scores] (first maximum wins).  The stages are phases of one
 * kernel separated by grid barriers (one block per CU); all steps run inside the launch.
 *   layers[l]: bf16 weights (wqkv = q|k|v stacked, [3D, D]) in the FRAGMENT-TILED layout [out / 16][in / 8][16][8] -- element (o, i) at
 *     ((o / 16 * (in / 8) + i / 8) * 16 + o % 16) * 8 + i % 8, `out` zero-padded to a multiple of 16: what an MFMA operand load of 16 rows reads as
 *     one contiguous kilobyte (row-major rows cost one tag lookup per lane: 6.5 us per phase) --, fp32 biases and LayerNorm vectors; qkv = that layer's
 *     bf16 [B, N, 3D] row-major cache as left by the first pass (sam_gemm_bf16 of all rows), rows n_enc + t are written by the launch; allow = the layer's bits [B, Hm, N, ceil(N/32)] with strides.
 *   ans_ln bf16 [V, D], ocr_ln bf16 [B*No, D]: the two step-invariant LayerNorms of PrevPredEmbeddings; pos_emb / type_emb fp32 rows (ld in elements);
 *   wc bf16 [V -> multiple of 16, D] and wq bf16 [D, D], both fragment-tiled as above; bc, bq fp32; ptr_k bf16 [B, No, D] the pointer network's keys; ocr_mask u8 [B, No]; ptr_scale = 1/sqrt(D);
 *   prev_inds int64 [B, S]; fixed_scores fp32 [B, S, ld_fixed], ocr_scores fp32 [B, S, No]; seq_out (may be NULL) bf16 [B, N, D] receives the final
 *     hidden state of row n_enc + t.
 * Built for D = 768, F = 3072, head_dim 64, N <= 384 (caches past 256 rows take the keys in two chunks of 192 with a running maximum), No <= 128, <= 12 layers: anything else returns SAM_ERR_UNSUPPORTED (callers fall back to the
 * per-kernel step).  ws: sam_greedy_decode_ws_bytes(B, S, n_layers) bytes (~100 MB: the activations of every (XCD, step, layer) get their own 16-row
 * slot, see csrc/decode_steps.hip on coherence), 256-byte aligned; int32 word 256 of it (byte 1024) is a sticky error flag, non-zero when
 * a grid barrier timed out.  Zero the whole workspace once after allocation and again after an error: a clean launch leaves the barrier words at zero. */
typedef struct sam_decode_layer {
  const void *wqkv, *wo, *w1, *w2;
  const float *bqkv, *bo, *b1, *b2, *ln1_g, *ln1_b, *ln2_g, *ln2_b;
  void* qkv;
  const uint32_t* allow; int64_t allow_stride_b, allow_stride_h;
} sam_decode_layer;
typedef struct sam_decode_desc {
  int32_t n_layers, B, N, n_enc, S, H, D, F, V, No, t_begin, t_end;
  float scale, ln_eps, emb_ln_eps, ptr_scale;
  const sam_decode_layer* layers;                       /* host array of n_layers entries */
  const float *pos_emb, *type_emb, *emb_ln_g, *emb_ln_b; int64_t ld_pos, ld_type;
  const void *ans_ln, *ocr_ln;
  const void* wc; const float* bc;
  const void* wq; const float* bq;
  const void* ptr_k; const uint8_t* ocr_mask;
  int64_t* prev_inds;
  float* fixed_scores; int64_t ld_fixed;
  float* ocr_scores;
  void* seq_out;
} sam_decode_desc;
int64_t sam_greedy_decode_ws_bytes(int B, int S, int n_layers);
int sam_greedy_decode_steps(const sam_decode_desc* desc, void* ws, int64_t ws_bytes, void* stream);

/* ---- `output_attentions` (sam/sa_m4c.py:600-609; BertSpatialEncoder 765-769): attention_probs [B, H, N, N] fp32, rebuilt from what the fused forward saved ----
 * P = allow ? exp2(scale * log2(e) * <q, k> - lse2) : 0, times keep / (1 - p_drop) where the forward dropped (keep: its bit planes, NULL = no dropout), times
 * head_scale[h] (head_mask as one factor per head, :591-592; NULL = none).  Fully masked rows are exact zeros (:574-584).  Off the training path. */
int sam_attn_probs(const void* qkv, const uint32_t* allow, int64_t allow_stride_b, int64_t allow_stride_h, const float* lse2, const uint32_t* keep,
                   const float* head_scale, int B, int N, int H, int head_dim, float scale, float p_drop, float* out, void* stream);

/* ---- glue (csrc/glue.hip): up to 8 strided block copies / casts / accumulations / zero-fills in ONE launch ----
 * block q: dst[b, i, :cols] (+)= src[b, i, :cols] for b < batches, i < rows; element (b, i, c) at b * batch_stride + i * row_stride + c; src NULL = zeros;
 * src_f32 / dst_f32: 1 = fp32, 0 = bf16 (converted with round-to-nearest-even); accumulate: dst += src.  cols and every stride multiples of 4 elements.
 * Replaces torch.cat of the four token groups (sam/sa_m4c.py:814-818) and its backward's four slice copies, the OCR / decoder row slices of the MMT
 * output (sa_m4c.py:270-278) with their zero-padded gradient, the casts and zero-fills around PrevPredEmbeddings' backward (sa_m4c.py:900-948), the
 * clearing of the accumulated gradient ranges.  sam_ge_u8: out[i] = x[i] >= threshold (token type of a previous prediction, sa_m4c.py:936). */
typedef struct sam_copy_desc {
  const void* src; void* dst;
  int32_t batches, rows, cols;
  int64_t src_batch_stride, src_row_stride, dst_batch_stride, dst_row_stride;
  int32_t src_f32, dst_f32, accumulate;
} sam_copy_desc;
int sam_copy_blocks(const sam_copy_desc* descs, int count, void* stream);
int sam_ge_u8(const int64_t* x, int64_t n, int64_t threshold, uint8_t* out, void* stream);
/* element-wise row kernels of the stand-alone sub-modules (bf16 rows [rows, cols], fp32 arithmetic, one rounding; cols and strides multiples of 4):
 *   mode 0: out = a * b      -- BertIntermediate's backward, dy * gelu'(pre) with the derivative the forward GEMM stored (sa_m4c.py:678, 985-991)
 *   mode 1: out = a + vec    -- SpatialBertSelfAttention's `use_bias` head biases, context_layer + biases(0) (sa_m4c.py:439-443, 600-603); vec fp32 [cols]
 *   mode 2: out = a * vec    -- head_mask as one factor per head, broadcast over the head's 64 context columns (sa_m4c.py:591-592: probs * head_mask before P V) */
int sam_rowvec_bf16(int mode, const void* a, int64_t lda, const void* b, int64_t ldb, const float* vec, void* out, int64_t ldo, int64_t rows, int cols, void* stream);

/* ---- dropout RNG state in device memory (hipGraph capture) ----
 * Every dropout site takes (seed, offset) BY VALUE (counter-based: the backward regenerates the forward's mask from the same pair).  Launches
 * captured in a hipGraph would replay the same masks for ever; with a device-side state set, kernels launched afterwards (from any thread of
 * the process: the autograd engine launches the backward from its own) use
 * key = state[0] (if non-zero, else the by-value seed) and offset = state[1] + the by-value offset, read at execution time -- the graph (or the
 * host, between replays) advances state[1].  dev_state: uint64[2] in device memory, NULL switches back to by-value only. */
void sam_set_rng_state(const unsigned long long* dev_state);

#ifdef __cplusplus
}
#endif
#endif
