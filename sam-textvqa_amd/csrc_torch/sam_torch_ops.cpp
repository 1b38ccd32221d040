// PyTorch-ROCm custom ops (namespace sam_hip) over the C ABI of libsam_hip.so.
//
// The reference has no operator layer (eager PyTorch inside sam/sa_m4c.py); BASELINE.json's north star asks for the hot path to be "exposed to
// Python through PyTorch-ROCm custom ops".  This file is that layer: TORCH_LIBRARY schemas + ROCm implementations that check their tensors
// (TORCH_CHECK -> RuntimeError), allocate outputs through ATen, enqueue on c10::hip::getCurrentHIPStream() and never synchronise.
// No arithmetic happens here: every op is one or several calls into include/sam_hip.h.
//   fine-grained ops   sam_hip::linear, spatial_attn_fwd / _bwd, layernorm_fwd / _bwd, pack_masks, mask_bits_prefix_lm, mask_bits_from_additive,
//                      pack_relations (+ _bhnn), ptr_scores (+ _bwd), bce_loss, sumsq, adam_step, step_advance   (SURVEY 8(b)'s op list; the model's
//                      masks, pointer scores, loss and optimizer go through these)
//   coarse ops         sam_hip::encoder_layer_fwd / _bwd: one SpatialBertLayer / BertLayer (sam/sa_m4c.py:660-684) = 7 launches forward,
//                      ~12 backward, enqueued from C++ -- the Python/ctypes route costs ~20 us of host time per launch, 5.7 ms per training
//                      step against 8 ms of GPU time.
// Built by sam_textvqa_amd/_build.py with g++ against the torch headers; links libsam_hip.so from its own directory.
#include <ATen/ATen.h>
#include <c10/hip/HIPStream.h>
#include <torch/library.h>

#include <tuple>
#include <string>
#include <cstdlib>
#include <map>
#include <mutex>
#include <vector>

#include "sam_hip.h"

namespace {

using at::Tensor;
using c10::optional;

void* cur_stream() { return (void*)c10::hip::getCurrentHIPStream().stream(); }

void ok(int rc, const char* what) { TORCH_CHECK(rc == 0, what, " failed (rc=", rc, "): ", sam_last_error()); }

const Tensor& need(const Tensor& t, at::ScalarType dt, const char* name) {
  TORCH_CHECK(t.defined() && t.is_cuda(), name, " must be a GPU tensor (this package has no CPU path)");
  TORCH_CHECK(t.scalar_type() == dt, name, ": expected dtype ", dt, ", got ", t.scalar_type());
  return t;
}
void need2d(const Tensor& t, const char* name) {
  need(t, at::kBFloat16, name);
  TORCH_CHECK(t.dim() == 2 && t.stride(1) == 1, name, ": need a 2-D bf16 tensor with contiguous last dim");
}
void* p(const Tensor& t) { return t.defined() ? t.data_ptr() : nullptr; }
void* p(const optional<Tensor>& t) { return t.has_value() && t->defined() ? t->data_ptr() : nullptr; }

// ---------------------------------------------------------------------------------------------------------------- GEMM (mirror of ops.gemm)
struct GemmOpt {
  int epilogue = SAM_EPI_NONE;
  const Tensor* bias = nullptr;
  const Tensor* residual = nullptr;
  const Tensor* aux_in = nullptr;
  Tensor* aux_out = nullptr;
  float p_drop = 0.f;
  int64_t seed = 0, offset = 0;
  bool out_f32 = false;
  sam_ln_fuse* ln = nullptr;       // gemm_ln: LayerNorm behind a BIAS_DROPOUT_RES epilogue, fused into the split-K reduction pass when there is one
};

Tensor gemm(const Tensor& a, const Tensor& b, bool a_kc, bool b_kc, const GemmOpt& o) {
  need2d(a, "gemm A"); need2d(b, "gemm B");
  const int64_t M = a_kc ? a.size(0) : a.size(1), K = a_kc ? a.size(1) : a.size(0), N = b_kc ? b.size(0) : b.size(1);
  Tensor out = at::empty({M, N}, a.options().dtype(o.out_f32 ? at::kFloat : at::kBFloat16));
  sam_gemm_desc d = {};
  d.M = (int32_t)M; d.N = (int32_t)N; d.K = (int32_t)K;
  d.a_kcontig = a_kc; d.b_kcontig = b_kc; d.c_is_f32 = o.out_f32; d.epilogue = o.epilogue;
  d.A = a.data_ptr(); d.lda = a.stride(0); d.B = b.data_ptr(); d.ldb = b.stride(0); d.C = out.data_ptr(); d.ldc = out.stride(0);
  d.bias = o.bias ? (const float*)o.bias->data_ptr() : nullptr;
  if (o.residual) { d.residual = o.residual->data_ptr(); d.ldr = o.residual->stride(0); }
  if (o.aux_out) { d.aux_out = o.aux_out->data_ptr(); d.ld_aux = o.aux_out->stride(0); }
  if (o.aux_in) { d.aux_in = o.aux_in->data_ptr(); d.ld_aux = o.aux_in->stride(0); }
  d.p_drop = o.p_drop; d.seed = (uint64_t)o.seed; d.offset = (uint64_t)o.offset;
  d.ln = o.ln;
  Tensor ws;
  if (M <= 4096 && K >= 1536 && M * N <= (4 << 20)) {   // skinny problem with a long K (TextBert's 20 tokens/sample): the library may split K
    d.split_k = -1;                                      // and fold the epilogue into the partial-sum reduction
    const int64_t bytes = std::min<int64_t>(8 * (M * N + M) * 4, (int64_t)96 << 20);
    ws = at::empty({(bytes + 3) / 4}, a.options().dtype(at::kFloat));
    d.ws = (float*)ws.data_ptr(); d.ws_bytes = ws.numel() * 4; d.defer_reduce = 1;
  }
  ok(sam_gemm_bf16(&d, cur_stream()), "sam_gemm_bf16");
  return out;
}

std::tuple<Tensor, Tensor, Tensor> ln_fwd(const Tensor& x, const Tensor& gamma, const Tensor& beta, double eps);
// LayerNorm(gemm(...)) -> (z, y, mean, rstd): mirror of ops.gemm_ln (sam_ln_fuse when the library splits K, sam_layernorm_fwd otherwise: same bits)
std::tuple<Tensor, Tensor, Tensor, Tensor> gemm_ln(const Tensor& a, const Tensor& b, const Tensor& gamma, const Tensor& beta, double eps, GemmOpt o) {
  const int64_t M = a.size(0), N = b.size(0);
  if (N % 4 || N > 2048 || gamma.scalar_type() != at::kFloat || beta.scalar_type() != at::kFloat) {
    Tensor z = gemm(a, b, true, true, o);
    auto [y, mean, rstd] = ln_fwd(z, gamma, beta, eps);
    return {z, y, mean, rstd};
  }
  Tensor y = at::empty({M, N}, a.options()), mean = at::empty({M}, a.options().dtype(at::kFloat)), rstd = at::empty({M}, a.options().dtype(at::kFloat));
  sam_ln_fuse ln = {(const float*)gamma.data_ptr(), (const float*)beta.data_ptr(), (float)eps, y.data_ptr(), y.stride(0), (float*)mean.data_ptr(), (float*)rstd.data_ptr(), 0, nullptr, 0};
  if (M >= 2048) {
    // MMT-size products: the exchange workspace of the LayerNorm inside the launch -- one zero-filled buffer per (device, stream), grown on demand; every launch
    // leaves its counters zero again (include/sam_hip.h: sam_ln_fuse.xws)
    static std::map<std::pair<int, void*>, Tensor> cache;
    const int64_t bytes = sam_gemm_ln_ws_bytes((int)M, (int)N);
    Tensor& xws = cache[{(int)a.get_device(), cur_stream()}];
    if (!xws.defined() || xws.numel() * 4 < bytes) xws = at::zeros({(bytes + 3) / 4}, a.options().dtype(at::kFloat));
    ln.xws = (float*)xws.data_ptr(); ln.xws_bytes = xws.numel() * 4;
  }
  o.ln = &ln;
  Tensor z = gemm(a, b, true, true, o);
  if (ln.done) return {z, y, mean, rstd};
  auto [y2, mean2, rstd2] = ln_fwd(z, gamma, beta, eps);
  return {z, y2, mean2, rstd2};
}

// dW += dy^T x (and db += colsum(dy)) for up to 8 jobs in one launch
struct WgradJob { const Tensor* dy; const Tensor* x; const Tensor* dw; const Tensor* db; };
void wgrad_grouped(const std::vector<WgradJob>& jobs, bool accumulate = true) {
  sam_gemm_desc d[20] = {};
  TORCH_CHECK(jobs.size() >= 1 && jobs.size() <= 20, "wgrad_grouped: 1..20 jobs");
  for (size_t q = 0; q < jobs.size(); ++q) {
    const WgradJob& j = jobs[q];
    d[q].M = (int32_t)j.dy->size(1); d[q].N = (int32_t)j.x->size(1); d[q].K = (int32_t)j.dy->size(0);
    d[q].c_is_f32 = 1; d[q].accumulate = accumulate ? 1 : 0; d[q].epilogue = SAM_EPI_NONE;
    d[q].A = j.dy->data_ptr(); d[q].lda = j.dy->stride(0); d[q].B = j.x->data_ptr(); d[q].ldb = j.x->stride(0);
    d[q].C = j.dw->data_ptr(); d[q].ldc = j.dw->stride(0);
    d[q].bias_grad = j.db ? (float*)j.db->data_ptr() : nullptr;
  }
  // exchange workspace of the 8-wave grouped kernel: one zero-filled buffer per (device, stream), grown on demand; the kernel leaves its flag words
  // zero and launches on one stream are ordered, so consecutive calls share it
  {
    static std::mutex mu;
    static std::map<std::pair<int, void*>, Tensor> cache;
    const int64_t bytes = sam_gemm_grouped_ws_bytes(d, (int)jobs.size());
    std::lock_guard<std::mutex> lock(mu);
    Tensor& ws = cache[{(int)jobs[0].dy->get_device(), cur_stream()}];
    if (!ws.defined() || ws.numel() * 4 < bytes) ws = at::zeros({(bytes + 3) / 4}, jobs[0].dy->options().dtype(at::kFloat));
    d[0].ws = (float*)ws.data_ptr(); d[0].ws_bytes = ws.numel() * 4;
  }
  ok(sam_gemm_bf16_grouped(d, (int)jobs.size(), cur_stream()), "sam_gemm_bf16_grouped");
}

// ---------------------------------------------------------------------------------------------------------------- attention / layernorm
std::tuple<Tensor, Tensor, Tensor> attn_fwd(const Tensor& qkv, const Tensor& allow, int64_t batch, int64_t heads, double scale, double p_drop, int64_t seed,
                                            int64_t offset) {
  need2d(qkv, "qkv"); need(allow, at::kInt, "allow");
  TORCH_CHECK(allow.dim() == 4 && allow.is_contiguous(), "allow must be a contiguous int32 [B, H|1, N, NW] tensor");
  const int64_t rows = qkv.size(0), d_model = qkv.size(1) / 3, n = rows / batch;
  Tensor out = at::empty({rows, d_model}, qkv.options());
  Tensor lse2 = at::empty({batch, heads, n}, qkv.options().dtype(at::kFloat));
  Tensor keep = p_drop > 0 ? at::empty({batch, heads, n, allow.size(3)}, allow.options()) : Tensor();
  ok(sam_attn_fwd(qkv.data_ptr(), (const uint32_t*)allow.data_ptr(), allow.stride(0), allow.size(1) == 1 ? 0 : allow.stride(1), (int)batch, (int)n, (int)heads,
                  (int)(d_model / heads), (float)scale, (float)p_drop, (uint64_t)seed, (uint64_t)offset, out.data_ptr(), (float*)lse2.data_ptr(),
                  (uint32_t*)p(keep), cur_stream()),
     "sam_attn_fwd");
  return {out, lse2, keep.defined() ? keep : at::empty({0}, allow.options())};
}

// training forward: also returns the bf16 rounding residual of the output (what the one-pass backward takes delta from)
std::tuple<Tensor, Tensor, Tensor, Tensor> attn_fwd_train(const Tensor& qkv, const Tensor& allow, int64_t batch, int64_t heads, double scale, double p_drop, int64_t seed,
                                                          int64_t offset) {
  need2d(qkv, "qkv"); need(allow, at::kInt, "allow");
  TORCH_CHECK(allow.dim() == 4 && allow.is_contiguous(), "allow must be a contiguous int32 [B, H|1, N, NW] tensor");
  const int64_t rows = qkv.size(0), d_model = qkv.size(1) / 3, n = rows / batch;
  Tensor out = at::empty({rows, d_model}, qkv.options());
  Tensor out_lo = at::empty({rows, d_model}, qkv.options());
  Tensor lse2 = at::empty({batch, heads, n}, qkv.options().dtype(at::kFloat));
  Tensor keep = p_drop > 0 ? at::empty({batch, heads, n, allow.size(3)}, allow.options()) : Tensor();
  ok(sam_attn_fwd_train(qkv.data_ptr(), (const uint32_t*)allow.data_ptr(), allow.stride(0), allow.size(1) == 1 ? 0 : allow.stride(1), (int)batch, (int)n, (int)heads,
                        (int)(d_model / heads), (float)scale, (float)p_drop, (uint64_t)seed, (uint64_t)offset, out.data_ptr(), out_lo.data_ptr(), (float*)lse2.data_ptr(),
                        (uint32_t*)p(keep), cur_stream()),
     "sam_attn_fwd_train");
  return {out, lse2, keep.defined() ? keep : at::empty({0}, allow.options()), out_lo};
}

static bool fused_attn_bwd_enabled(int64_t n) {
  // read on every call, as ops.attn_bwd does: a test or an A/B script that flips SAM_ATTN_BWD_FUSED between calls must see the same route on the
  // per-kernel (Python) path and on this coarse one
  const char* s = getenv("SAM_ATTN_BWD_FUSED");
  return !(s && s[0] == '0') && n <= sam_attn_bwd_fused_max_n();
}

Tensor attn_bwd_fused(const Tensor& dout, const Tensor& qkv, const Tensor& out, const Tensor& out_lo, const Tensor& lse2, const Tensor& allow, const Tensor& keep,
                      int64_t batch, int64_t heads, double scale, double p_drop) {
  need2d(dout, "dout"); need2d(qkv, "qkv"); need2d(out, "out"); need2d(out_lo, "out_lo"); need(lse2, at::kFloat, "lse2"); need(allow, at::kInt, "allow");
  const int64_t rows = qkv.size(0), d_model = qkv.size(1) / 3, n = rows / batch;
  Tensor dqkv = at::empty_like(qkv);
  const bool has_keep = keep.defined() && keep.numel() > 0;
  ok(sam_attn_bwd_fused(dout.data_ptr(), qkv.data_ptr(), out.data_ptr(), out_lo.data_ptr(), (const float*)lse2.data_ptr(), (const uint32_t*)allow.data_ptr(),
                        allow.stride(0), allow.size(1) == 1 ? 0 : allow.stride(1), has_keep ? (const uint32_t*)keep.data_ptr() : nullptr, (int)batch, (int)n, (int)heads,
                        (int)(d_model / heads), (float)scale, (float)p_drop, dqkv.data_ptr(), cur_stream()),
     "sam_attn_bwd_fused");
  return dqkv;
}

Tensor attn_bwd(const Tensor& dout, const Tensor& qkv, const Tensor& lse2, const Tensor& allow, const Tensor& keep, int64_t batch, int64_t heads, double scale,
                double p_drop) {
  need2d(dout, "dout"); need2d(qkv, "qkv"); need(lse2, at::kFloat, "lse2"); need(allow, at::kInt, "allow");
  const int64_t rows = qkv.size(0), d_model = qkv.size(1) / 3, n = rows / batch;
  Tensor dqkv = at::empty_like(qkv);
  Tensor delta = at::empty({batch, heads, n}, lse2.options());
  const bool has_keep = keep.defined() && keep.numel() > 0;
  ok(sam_attn_bwd(dout.data_ptr(), qkv.data_ptr(), (const float*)lse2.data_ptr(), (const uint32_t*)allow.data_ptr(), allow.stride(0),
                  allow.size(1) == 1 ? 0 : allow.stride(1), has_keep ? (const uint32_t*)keep.data_ptr() : nullptr, (int)batch, (int)n, (int)heads,
                  (int)(d_model / heads), (float)scale, (float)p_drop, dqkv.data_ptr(), (float*)delta.data_ptr(), cur_stream()),
     "sam_attn_bwd");
  return dqkv;
}

std::tuple<Tensor, Tensor, Tensor> ln_fwd(const Tensor& x, const Tensor& gamma, const Tensor& beta, double eps) {
  TORCH_CHECK(x.is_cuda() && x.dim() == 2 && x.stride(1) == 1 && (x.scalar_type() == at::kBFloat16 || x.scalar_type() == at::kFloat),
              "layernorm_fwd: need a 2-D bf16/fp32 GPU tensor with contiguous rows");
  need(gamma, at::kFloat, "gamma"); need(beta, at::kFloat, "beta");
  const int64_t m = x.size(0), d = x.size(1);
  Tensor y = at::empty({m, d}, x.options().dtype(at::kBFloat16));
  Tensor mean = at::empty({m}, x.options().dtype(at::kFloat)), rstd = at::empty({m}, x.options().dtype(at::kFloat));
  ok(sam_layernorm_fwd(x.data_ptr(), x.scalar_type() == at::kFloat, x.stride(0), (const float*)gamma.data_ptr(), (const float*)beta.data_ptr(), (float)eps, (int)m,
                       (int)d, y.data_ptr(), y.stride(0), (float*)mean.data_ptr(), (float*)rstd.data_ptr(), cur_stream()),
     "sam_layernorm_fwd");
  return {y, mean, rstd};
}

// -> (dx, dx_dropped or undefined); dgamma / dbeta / dbias accumulated in place
// Deferred LayerNorm finalizes (include/sam_hip.h: sam_layernorm_bwd_finalize_batch): when switched on (set_ln_defer, by the Trainer for the span of
// a backward pass) the LayerNorm backwards of the coarse encoder-layer op leave their partial sums in private workspaces queued here, and
// ln_finalize_flush reduces all of them in one launch.  One queue per process (the backward pass runs on one thread at a time).
struct LnQueue {
  std::mutex mu;
  bool defer = false;
  int64_t D = 0;
  std::vector<sam_ln_finalize_item> items;
  std::vector<Tensor> keep;
};
LnQueue& ln_queue() { static LnQueue q; return q; }

std::tuple<Tensor, Tensor> ln_bwd(const Tensor& dy, const Tensor& x, const Tensor& mean, const Tensor& rstd, const Tensor& gamma, const Tensor& dgamma,
                                  const Tensor& dbeta, const Tensor* dbias, bool want_dropped, double p_drop, int64_t seed, int64_t offset, bool accumulate = true,
                                  bool may_defer = false) {
  need2d(dy, "dy");
  const int64_t m = x.size(0), d = x.size(1);
  Tensor dx = at::empty({m, d}, dy.options());
  Tensor dxd = (want_dropped && p_drop > 0) ? at::empty({m, d}, dy.options()) : Tensor();
  Tensor ws = at::empty({(sam_layernorm_bwd_ws_bytes((int)d) + 3) / 4}, dy.options().dtype(at::kFloat));
  LnQueue& q = ln_queue();
  bool defer = false;
  if (may_defer) {
    std::lock_guard<std::mutex> lock(q.mu);
    defer = q.defer && (q.items.empty() || q.D == d);
    if (defer) {
      q.D = d;
      q.items.push_back({(const float*)ws.data_ptr(), (int32_t)sam_layernorm_bwd_partial_rows((int)m), accumulate ? 1 : 0, (float*)dgamma.data_ptr(),
                         (float*)dbeta.data_ptr(), dbias ? (float*)dbias->data_ptr() : nullptr});
      q.keep.push_back(ws);
    }
  }
  ok(sam_layernorm_bwd(dy.data_ptr(), dy.stride(0), x.data_ptr(), x.scalar_type() == at::kFloat, x.stride(0), (const float*)mean.data_ptr(),
                       (const float*)rstd.data_ptr(), (const float*)gamma.data_ptr(), (int)m, (int)d, dx.data_ptr(), p(dxd), dx.stride(0), (float)p_drop,
                       (uint64_t)seed, (uint64_t)offset, (float*)dgamma.data_ptr(), (float*)dbeta.data_ptr(), dbias ? (float*)dbias->data_ptr() : nullptr,
                       (accumulate ? 1 : 0) | (defer ? 4 : 0), (float*)ws.data_ptr(), cur_stream()),
     "sam_layernorm_bwd");
  return {dx, dxd.defined() ? dxd : (want_dropped ? dx : Tensor())};
}

void set_ln_defer(bool on) {
  LnQueue& q = ln_queue();
  std::lock_guard<std::mutex> lock(q.mu);
  q.defer = on;
}

void ln_finalize_flush() {
  LnQueue& q = ln_queue();
  std::lock_guard<std::mutex> lock(q.mu);
  if (q.items.empty()) return;
  ok(sam_layernorm_bwd_finalize_batch(q.items.data(), (int)q.items.size(), (int)q.D, cur_stream()), "sam_layernorm_bwd_finalize_batch");
  q.items.clear();
  q.keep.clear();          // (the flush runs on the stream the partials were written on: stream order protects the workspaces)
}

void ln_finalize_clear() {      // a backward pass that raised: drop what it queued (the partial-sum workspaces may already be gone) and stop deferring
  LnQueue& q = ln_queue();
  std::lock_guard<std::mutex> lock(q.mu);
  q.items.clear();
  q.keep.clear();
  q.defer = false;
}

// ---------------------------------------------------------------------------------------------------------------- masks (pack_relations)
int64_t words_per_row(int64_t n) {
  const int nw = sam_attn_words_per_row((int)n);
  TORCH_CHECK(nw > 0, "sequence length ", n, " exceeds the fused-attention limit (384 keys)");
  return nw;
}
// question / object / OCR padding masks (int64) -> (key_valid u8 [B, T+No+Nc], question u8 [B,T], ocr u8 [B,Nc]); sa_m4c.py:805-812, :386, :889
std::tuple<Tensor, Tensor, Tensor> pack_masks(const Tensor& q, const Tensor& o, const Tensor& c) {
  need(q, at::kLong, "question_mask"); need(o, at::kLong, "pad_obj_mask"); need(c, at::kLong, "pad_ocr_mask");
  TORCH_CHECK(q.dim() == 2 && o.dim() == 2 && c.dim() == 2 && q.is_contiguous() && o.is_contiguous() && c.is_contiguous(), "pack_masks: contiguous [B, n] masks");
  const int64_t b = q.size(0), t = q.size(1), no = o.size(1), nc = c.size(1);
  auto u8 = q.options().dtype(at::kByte);
  Tensor kv = at::empty({b, t + no + nc}, u8), q8 = at::empty({b, t}, u8), c8 = at::empty({b, nc}, u8);
  ok(sam_pack_masks_u8((const int64_t*)q.data_ptr(), (int)t, (const int64_t*)o.data_ptr(), (int)no, (const int64_t*)c.data_ptr(), (int)nc, (int)b,
                       (uint8_t*)kv.data_ptr(), (uint8_t*)q8.data_ptr(), (uint8_t*)c8.data_ptr(), cur_stream()), "sam_pack_masks_u8");
  return {kv, q8, c8};
}
Tensor mask_bits_prefix_lm(const Tensor& key_valid, int64_t n_dec) {                  // sa_m4c.py:805-844
  need(key_valid, at::kByte, "key_valid");
  TORCH_CHECK(key_valid.dim() == 2 && key_valid.is_contiguous(), "key_valid: contiguous uint8 [B, n_enc]");
  const int64_t b = key_valid.size(0), n_enc = key_valid.size(1), n = n_enc + n_dec, nw = words_per_row(n);
  Tensor out = at::empty({b, 1, n, nw}, key_valid.options().dtype(at::kInt));
  ok(sam_mask_bits_prefix_lm((const uint8_t*)key_valid.data_ptr(), (int)b, (int)n_enc, (int)n_dec, (int)nw, (uint32_t*)out.data_ptr(), cur_stream()), "sam_mask_bits_prefix_lm");
  return out;
}
Tensor mask_bits_from_additive(const Tensor& mask) {                                  // sa_m4c.py:453-455
  need(mask, at::kFloat, "attention_mask");
  TORCH_CHECK(mask.dim() == 4 && mask.size(1) == 1 && mask.size(2) == mask.size(3) && mask.is_contiguous(), "attention_mask must be a contiguous [B,1,N,N] tensor");
  const int64_t b = mask.size(0), n = mask.size(2), nw = words_per_row(n);
  Tensor out = at::empty({b, 1, n, nw}, mask.options().dtype(at::kInt));
  ok(sam_mask_bits_from_additive((const float*)mask.data_ptr(), (int)b, (int)n, (int)nw, (uint32_t*)out.data_ptr(), cur_stream()), "sam_mask_bits_from_additive");
  return out;
}
// the relation tensor of the batch (int8 [B, Noo, Noo, R] multi-hot, dataset layout) AND-ed into the base bits, per head; sa_m4c.py:470-552,568
Tensor pack_relations(const Tensor& base_bits, const Tensor& adj, int64_t n_txt, int64_t heads, int64_t quadrant_bits) {
  need(base_bits, at::kInt, "base_bits"); need(adj, at::kChar, "spatial_adj_matrix");
  TORCH_CHECK(base_bits.dim() == 4 && base_bits.is_contiguous() && adj.dim() == 4 && adj.is_contiguous(), "pack_relations: contiguous base [B,1,N,NW], adj [B,Noo,Noo,R]");
  const int64_t b = base_bits.size(0), n = base_bits.size(2), nw = base_bits.size(3);
  Tensor out = at::empty({b, heads, n, nw}, base_bits.options());
  ok(sam_mask_bits_spatial((const uint32_t*)base_bits.data_ptr(), (const int8_t*)adj.data_ptr(), (int)b, (int)n, (int)nw, (int)n_txt, (int)adj.size(1), (int)adj.size(3),
                           (int)heads, (unsigned)quadrant_bits, (uint32_t*)out.data_ptr(), cur_stream()), "sam_mask_bits_spatial");
  return out;
}
Tensor pack_relations_bhnn(const Tensor& rel, const optional<Tensor>& base_bits) {       // north-star layout int8 [B,H,N,N]
  need(rel, at::kChar, "rel");
  TORCH_CHECK(rel.dim() == 4 && rel.size(2) == rel.size(3) && rel.is_contiguous(), "rel must be a contiguous int8 [B,H,N,N] tensor");
  const int64_t b = rel.size(0), h = rel.size(1), n = rel.size(2), nw = words_per_row(n);
  Tensor out = at::empty({b, h, n, nw}, rel.options().dtype(at::kInt));
  ok(sam_mask_bits_from_int8_bhnn((const int8_t*)rel.data_ptr(), (const uint32_t*)p(base_bits), (int)b, (int)h, (int)n, (int)nw, (uint32_t*)out.data_ptr(), cur_stream()),
     "sam_mask_bits_from_int8_bhnn");
  return out;
}

// ---------------------------------------------------------------------------------------------------------------- pointer net / loss / optimizer
Tensor ptr_scores_fwd(const Tensor& q, const Tensor& k, const Tensor& ocr_mask, double scale) {      // sa_m4c.py:891-893
  need(q, at::kBFloat16, "q"); need(k, at::kBFloat16, "k"); need(ocr_mask, at::kByte, "ocr_mask");
  TORCH_CHECK(q.dim() == 3 && k.dim() == 3 && q.is_contiguous() && k.is_contiguous() && ocr_mask.is_contiguous(), "ptr_scores: contiguous q [B,S,D], k [B,No,D], mask [B,No]");
  const int64_t b = q.size(0), s = q.size(1), d = q.size(2), no = k.size(1);
  Tensor out = at::empty({b, s, no}, q.options().dtype(at::kFloat));
  ok(sam_ptr_scores_fwd(q.data_ptr(), k.data_ptr(), (const uint8_t*)ocr_mask.data_ptr(), (int)b, (int)s, (int)no, (int)d, (float)scale, (float*)out.data_ptr(), out.stride(0),
                        out.stride(1), cur_stream()), "sam_ptr_scores_fwd");
  return out;
}
std::tuple<Tensor, Tensor> ptr_scores_bwd(const Tensor& ds, const Tensor& q, const Tensor& k, double scale) {
  need(ds, at::kFloat, "dscores"); need(q, at::kBFloat16, "q"); need(k, at::kBFloat16, "k");
  TORCH_CHECK(ds.dim() == 3 && ds.stride(2) == 1 && q.is_contiguous() && k.is_contiguous(), "ptr_scores_bwd: dscores [B,S,No] with unit last stride");
  const int64_t b = q.size(0), s = q.size(1), d = q.size(2), no = k.size(1);
  Tensor dq = at::empty_like(q), dk = at::empty_like(k);
  ok(sam_ptr_scores_bwd((const float*)ds.data_ptr(), ds.stride(0), ds.stride(1), q.data_ptr(), k.data_ptr(), (int)b, (int)s, (int)no, (int)d, (float)scale, dq.data_ptr(),
                        dk.data_ptr(), cur_stream()), "sam_ptr_scores_bwd");
  return {dq, dk};
}
// M4CDecodingBCEWithMaskLoss (task_utils.py:19-30): value + analytic gradient -> (loss f32 [1], d_fixed bf16 [R,V], d_ocr f32 [R,No])
std::tuple<Tensor, Tensor, Tensor> bce_loss(const Tensor& fixed, const Tensor& ocr, const Tensor& targets, const Tensor& loss_mask, double grad_scale,
                                            const optional<Tensor>& global_count) {
  need(fixed, at::kFloat, "fixed_scores"); need(ocr, at::kFloat, "ocr_scores"); need(targets, at::kFloat, "targets"); need(loss_mask, at::kFloat, "loss_mask");
  TORCH_CHECK(fixed.dim() == 2 && ocr.dim() == 2 && targets.dim() == 2 && fixed.stride(1) == 1 && ocr.stride(1) == 1 && targets.stride(1) == 1 && loss_mask.is_contiguous(),
              "bce_loss: 2-D score / target blocks with unit last stride");
  const int64_t r = fixed.size(0), v = fixed.size(1), no = ocr.size(1);
  Tensor loss = at::empty({1}, fixed.options()), d_fixed = at::empty({r, v}, fixed.options().dtype(at::kBFloat16)), d_ocr = at::empty({r, no}, fixed.options());
  ok(sam_bce_loss((const float*)fixed.data_ptr(), fixed.stride(0), (const float*)ocr.data_ptr(), ocr.stride(0), (const float*)targets.data_ptr(), targets.stride(0),
                  (const float*)loss_mask.data_ptr(), (int)r, (int)v, (int)no, (float)grad_scale, (const float*)p(global_count), (float*)loss.data_ptr(), d_fixed.data_ptr(),
                  d_fixed.stride(0), (float*)d_ocr.data_ptr(), d_ocr.stride(0), cur_stream()), "sam_bce_loss");
  return {loss, d_fixed, d_ocr};
}
Tensor& workspace(const char* tag, int64_t bytes, const Tensor& like) {      // per (tag, device, stream) scratch that only grows
  static std::mutex mu;
  static std::map<std::tuple<std::string, int, void*>, Tensor> cache;
  std::lock_guard<std::mutex> lock(mu);
  Tensor& ws = cache[{std::string(tag), (int)like.get_device(), cur_stream()}];
  if (!ws.defined() || ws.numel() * 4 < bytes) ws = at::empty({(bytes + 3) / 4}, like.options().dtype(at::kFloat));
  return ws;
}
// row-sparse region from its schema form: region = [lo, hi, row_len] (empty: none), touched = uint8 [rows]
const sam_sparse_rows* sparse_of(sam_sparse_rows& s, at::IntArrayRef region, const optional<Tensor>& touched) {
  if (region.size() != 3 || !touched.has_value() || !touched->defined()) return nullptr;
  need(*touched, at::kByte, "touched");
  s.lo = region[0]; s.hi = region[1]; s.row_len = (int32_t)region[2]; s.touched = (const uint8_t*)touched->data_ptr();
  TORCH_CHECK(touched->numel() * region[2] >= region[1] - region[0], "row-sparse region: fewer flags than rows");
  return &s;
}
void sumsq(const Tensor& g, Tensor out, at::IntArrayRef region, const optional<Tensor>& touched) {      // clip_grad_norm_'s norm, train.py:139
  need(g, at::kFloat, "g"); need(out, at::kFloat, "out");
  Tensor& ws = workspace("sumsq", sam_sumsq_ws_bytes(), g);
  sam_sparse_rows s;
  ok(sam_sumsq_f32((const float*)g.data_ptr(), g.numel(), sparse_of(s, region, touched), (float*)out.data_ptr(), (float*)ws.data_ptr(), cur_stream()), "sam_sumsq_f32");
}
// clip + Adam + bf16 shadow refresh over the flat buffers (train.py:139-142, task_utils.py:33-57).  dev_sched given: schedule read from device memory
// (captured steps, see step_advance); else seg_lr / step by value.
void adam_step(Tensor p_, Tensor g, Tensor m, Tensor v, const optional<Tensor>& p_bf16, at::IntArrayRef seg_end, at::ArrayRef<double> seg_lr, int64_t step,
               double beta1, double beta2, double eps, const optional<Tensor>& gnorm_sq, double max_norm, const optional<Tensor>& dev_sched, at::IntArrayRef region,
               const optional<Tensor>& touched) {
  need(p_, at::kFloat, "p"); need(g, at::kFloat, "g"); need(m, at::kFloat, "exp_avg"); need(v, at::kFloat, "exp_avg_sq");
  TORCH_CHECK(seg_end.size() >= 1 && seg_end.size() <= 8, "adam_step: 1..8 segments");
  int64_t ends[8]; float lrs[8] = {};
  for (size_t s = 0; s < seg_end.size(); ++s) { ends[s] = seg_end[s]; lrs[s] = s < seg_lr.size() ? (float)seg_lr[s] : 0.f; }
  sam_sparse_rows s;
  const sam_sparse_rows* sp = sparse_of(s, region, touched);
  if (dev_sched.has_value() && dev_sched->defined())
    ok(sam_adam_step_dev((float*)p_.data_ptr(), (float*)g.data_ptr(), (float*)m.data_ptr(), (float*)v.data_ptr(), p(p_bf16), p_.numel(), ends, (int)seg_end.size(),
                         (float)beta1, (float)beta2, (float)eps, (const float*)dev_sched->data_ptr(), (const float*)p(gnorm_sq), (float)max_norm, sp, cur_stream()), "sam_adam_step_dev");
  else
    ok(sam_adam_step((float*)p_.data_ptr(), (float*)g.data_ptr(), (float*)m.data_ptr(), (float*)v.data_ptr(), p(p_bf16), p_.numel(), ends, lrs, (int)seg_end.size(),
                     (float)beta1, (float)beta2, (float)eps, step, (const float*)p(gnorm_sq), (float)max_norm, sp, cur_stream()), "sam_adam_step");
}
// head node of a captured training step (include/sam_hip.h: sam_step_advance)
void step_advance(const optional<Tensor>& rng_state, int64_t offset_stride, Tensor step_counter, at::ArrayRef<double> base_lr, int64_t warmup_iters, double warmup_factor,
                  at::IntArrayRef decay_iters, double lr_decay, double beta1, double beta2, Tensor dev_sched) {
  need(step_counter, at::kLong, "step_counter"); need(dev_sched, at::kFloat, "dev_sched");
  TORCH_CHECK(base_lr.size() >= 1 && base_lr.size() <= 8 && decay_iters.size() <= 4 && (int64_t)dev_sched.numel() >= (int64_t)base_lr.size() + 2, "step_advance: bad schedule sizes");
  sam_lr_schedule sc = {};
  sc.nseg = (int32_t)base_lr.size();
  for (size_t s = 0; s < base_lr.size(); ++s) sc.base_lr[s] = base_lr[s];
  sc.warmup_iters = warmup_iters; sc.warmup_factor = warmup_factor; sc.n_decay = (int32_t)decay_iters.size();
  for (size_t q = 0; q < decay_iters.size(); ++q) sc.decay_iters[q] = decay_iters[q];
  sc.lr_decay = lr_decay; sc.beta1 = beta1; sc.beta2 = beta2;
  ok(sam_step_advance((unsigned long long*)p(rng_state), (uint64_t)offset_stride, (int64_t*)step_counter.data_ptr(), &sc, (float*)dev_sched.data_ptr(), cur_stream()),
     "sam_step_advance");
}

// ---------------------------------------------------------------------------------------------------------------- coarse: one encoder layer
// params: wqkv bf16 [3D,D], bqkv f32 [3D], wo bf16 [D,D], bo f32, ln1_w, ln1_b, w1 bf16 [I,D], b1 f32, w2 bf16 [D,I], b2 f32, ln2_w, ln2_b
enum { P_WQKV, P_BQKV, P_WO, P_BO, P_LN1W, P_LN1B, P_W1, P_B1, P_W2, P_B2, P_LN2W, P_LN2B, P_COUNT };
// saved: x, qkv, ctx, lse2, keep, z1, mean1, rstd1, a, pre, h, z2, mean2, rstd2
// (+ ctx_lo, the output residual of the attention forward, last: empty when the sequence is too long for the one-pass attention backward)
enum { S_X, S_QKV, S_CTX, S_LSE, S_KEEP, S_Z1, S_MEAN1, S_RSTD1, S_A, S_PRE, S_H, S_Z2, S_MEAN2, S_RSTD2, S_CTXLO, S_COUNT };

std::vector<Tensor> encoder_layer_fwd(const Tensor& x, const Tensor& allow, at::TensorList params, int64_t batch, int64_t heads, double scale, double p_attn,
                                      double p_hid, at::IntArrayRef seeds, double eps1, double eps2) {
  TORCH_CHECK(params.size() == P_COUNT, "encoder_layer_fwd: expected ", (int)P_COUNT, " parameter tensors");
  TORCH_CHECK(seeds.size() == 6, "encoder_layer_fwd: seeds = (seed, offset) x 3 dropout sites");
  need2d(x, "x");
  GemmOpt o;
  o.epilogue = SAM_EPI_BIAS; o.bias = &params[P_BQKV];
  Tensor qkv = gemm(x, params[P_WQKV], true, true, o);                                                   // sa_m4c.py:554-560
  Tensor ctx, lse2, keep, ctx_lo;
  if (fused_attn_bwd_enabled(x.size(0) / batch)) std::tie(ctx, lse2, keep, ctx_lo) = attn_fwd_train(qkv, allow, batch, heads, scale, p_attn, seeds[0], seeds[1]);      // :563-598
  else { std::tie(ctx, lse2, keep) = attn_fwd(qkv, allow, batch, heads, scale, p_attn, seeds[0], seeds[1]); ctx_lo = at::empty({0}, x.options()); }
  o = GemmOpt(); o.epilogue = SAM_EPI_BIAS_DROPOUT_RES; o.bias = &params[P_BO]; o.residual = &x; o.p_drop = (float)p_hid; o.seed = seeds[2]; o.offset = seeds[3];
  auto [z1, a, mean1, rstd1] = gemm_ln(ctx, params[P_WO], params[P_LN1W], params[P_LN1B], eps1, o);      // BertSelfOutput via :653
  Tensor pre = at::empty({x.size(0), params[P_W1].size(0)}, x.options());
  o = GemmOpt(); o.epilogue = SAM_EPI_BIAS_GELU_GRAD; o.bias = &params[P_B1]; o.aux_out = &pre;      // pre := gelu'(a W1^T + b1): the backward multiplies, no erf there
  Tensor h = gemm(a, params[P_W1], true, true, o);                                                        // BertIntermediate via :678
  o = GemmOpt(); o.epilogue = SAM_EPI_BIAS_DROPOUT_RES; o.bias = &params[P_B2]; o.residual = &a; o.p_drop = (float)p_hid; o.seed = seeds[4]; o.offset = seeds[5];
  auto [z2, y, mean2, rstd2] = gemm_ln(h, params[P_W2], params[P_LN2W], params[P_LN2B], eps2, o);          // BertOutput via :680
  return {y, x, qkv, ctx, lse2, keep, z1, mean1, rstd1, a, pre, h, z2, mean2, rstd2, ctx_lo};
}

// grads: same order as params, fp32 views into the flat gradient buffer (accumulated in place).  Returns dx (undefined-size-0 when !need_dx).
static std::vector<Tensor> encoder_layer_bwd_impl(const Tensor& dy_in, at::TensorList saved, const Tensor& allow, at::TensorList params, at::TensorList grads, int64_t batch,
                                                  int64_t heads, double scale, double p_attn, double p_hid, at::IntArrayRef seeds, bool need_dx, bool accumulate, bool defer_wgrad) {
  // accumulate = false: every one of the twelve gradients is OVERWRITTEN (each is written exactly once by this call) -- the caller then need not
  // zero them before the backward pass, and the weight-gradient kernels skip the read half of their read-modify-write
  TORCH_CHECK(saved.size() == S_COUNT && params.size() == P_COUNT && grads.size() == P_COUNT, "encoder_layer_bwd: bad list sizes");
  Tensor dy = dy_in;
  if (dy.scalar_type() != at::kBFloat16 || !dy.is_contiguous()) dy = dy.to(at::kBFloat16).contiguous();
  const Tensor &x = saved[S_X], &qkv = saved[S_QKV], &ctx = saved[S_CTX], &lse2 = saved[S_LSE], &keep = saved[S_KEEP], &z1 = saved[S_Z1], &a = saved[S_A],
               &pre = saved[S_PRE], &h = saved[S_H], &z2 = saved[S_Z2];
  // ---- output block: y = LN(dropout(h W2^T + b2) + a)
  auto [dz2, dy2] = ln_bwd(dy, z2, saved[S_MEAN2], saved[S_RSTD2], params[P_LN2W], grads[P_LN2W], grads[P_LN2B], &grads[P_B2], true, p_hid, seeds[4], seeds[5], accumulate, true);
  GemmOpt o;
  o.epilogue = SAM_EPI_MUL_AUX; o.aux_in = &pre;
  Tensor dpre = gemm(dy2, params[P_W2], true, false, o);
  // ---- intermediate: h = gelu(a W1^T + b1)
  o = GemmOpt(); o.epilogue = SAM_EPI_BIAS_DROPOUT_RES; o.residual = &dz2;
  Tensor da = gemm(dpre, params[P_W1], true, false, o);
  // ---- attention output block: a = LN(dropout(ctx Wo^T + bo) + x)
  auto [dz1, dy1] = ln_bwd(da, z1, saved[S_MEAN1], saved[S_RSTD1], params[P_LN1W], grads[P_LN1W], grads[P_LN1B], &grads[P_BO], true, p_hid, seeds[2], seeds[3], accumulate, true);
  Tensor dctx = gemm(dy1, params[P_WO], true, false, GemmOpt());
  // ---- attention core + fused QKV projection
  Tensor dqkv = saved[S_CTXLO].numel() ? attn_bwd_fused(dctx, qkv, ctx, saved[S_CTXLO], lse2, allow, keep, batch, heads, scale, p_attn)
                                       : attn_bwd(dctx, qkv, lse2, allow, keep, batch, heads, scale, p_attn);
  const Tensor dw2 = grads[P_W2], dw1 = grads[P_W1], db1 = grads[P_B1], dwo = grads[P_WO], dwqkv = grads[P_WQKV], dbqkv = grads[P_BQKV];
  // defer_wgrad: the four weight gradients are left to the caller, which runs them for several layers in ONE grouped launch (TextBert's three
  // 1280-row layers: 3 x 38 us of launches that cannot fill the chip -> one); the gradient operands come back with dx
  if (!defer_wgrad) wgrad_grouped({{&dy2, &h, &dw2, nullptr}, {&dpre, &a, &dw1, &db1}, {&dy1, &ctx, &dwo, nullptr}, {&dqkv, &x, &dwqkv, &dbqkv}}, accumulate);
  Tensor dx = at::empty({0}, x.options());
  if (need_dx) {
    o = GemmOpt(); o.epilogue = SAM_EPI_BIAS_DROPOUT_RES; o.residual = &dz1;
    dx = gemm(dqkv, params[P_WQKV], true, false, o);
  }
  if (defer_wgrad) return {dx, dy2, dpre, dy1, dqkv};
  return {dx};
}
Tensor encoder_layer_bwd(const Tensor& dy_in, at::TensorList saved, const Tensor& allow, at::TensorList params, at::TensorList grads, int64_t batch, int64_t heads,
                         double scale, double p_attn, double p_hid, at::IntArrayRef seeds, bool need_dx, bool accumulate) {
  return encoder_layer_bwd_impl(dy_in, saved, allow, params, grads, batch, heads, scale, p_attn, p_hid, seeds, need_dx, accumulate, false)[0];
}
std::vector<Tensor> encoder_layer_bwd_nowgrad(const Tensor& dy_in, at::TensorList saved, const Tensor& allow, at::TensorList params, at::TensorList grads, int64_t batch,
                                              int64_t heads, double scale, double p_attn, double p_hid, at::IntArrayRef seeds, bool need_dx, bool accumulate) {
  return encoder_layer_bwd_impl(dy_in, saved, allow, params, grads, batch, heads, scale, p_attn, p_hid, seeds, need_dx, accumulate, true);
}
// dW_q (+)= dy_q^T x_q (and db_q (+)= column sums of dy_q) for up to 20 problems in one launch
void wgrad_grouped_op(at::TensorList dys, at::TensorList xs, at::TensorList dws, at::TensorList dbs, bool accumulate) {
  TORCH_CHECK(dys.size() == xs.size() && dys.size() == dws.size() && dys.size() == dbs.size(), "wgrad_grouped: list sizes differ");
  std::vector<WgradJob> jobs;
  for (size_t q = 0; q < dys.size(); ++q) jobs.push_back({&dys[q], &xs[q], &dws[q], dbs[q].numel() ? &dbs[q] : nullptr});
  wgrad_grouped(jobs, accumulate);
}

// ---------------------------------------------------------------------------------------------------------------- fine-grained op wrappers
std::tuple<Tensor, Tensor> linear_op(const Tensor& x, const Tensor& w, const optional<Tensor>& bias, int64_t epilogue, const optional<Tensor>& residual,
                                     const optional<Tensor>& aux_in, bool want_aux_out, double p_drop, int64_t seed, int64_t offset, bool b_kcontig, bool out_f32) {
  GemmOpt o;
  o.epilogue = (int)epilogue; o.p_drop = (float)p_drop; o.seed = seed; o.offset = offset; o.out_f32 = out_f32;
  Tensor b_, r_, ai_, aux;
  if (bias.has_value() && bias->defined()) { b_ = need(*bias, at::kFloat, "bias"); o.bias = &b_; }
  if (residual.has_value() && residual->defined()) { r_ = *residual; need2d(r_, "residual"); o.residual = &r_; }
  if (aux_in.has_value() && aux_in->defined()) { ai_ = *aux_in; need2d(ai_, "aux_in"); o.aux_in = &ai_; }
  const int64_t N = b_kcontig ? w.size(0) : w.size(1);
  if (want_aux_out) { aux = at::empty({x.size(0), N}, x.options()); o.aux_out = &aux; }
  Tensor y = gemm(x, w, true, b_kcontig, o);
  return {y, want_aux_out ? aux : at::empty({0}, x.options())};
}

std::tuple<Tensor, Tensor, Tensor> layernorm_fwd_op(const Tensor& x, const Tensor& gamma, const Tensor& beta, double eps) { return ln_fwd(x, gamma, beta, eps); }

std::tuple<Tensor, Tensor, Tensor> layernorm_bwd_op(const Tensor& dy, const Tensor& x, const Tensor& mean, const Tensor& rstd, const Tensor& gamma) {
  Tensor dg = at::zeros_like(gamma), db = at::zeros_like(gamma);
  auto r = ln_bwd(dy, x, mean, rstd, gamma, dg, db, nullptr, false, 0.0, 0, 0);
  return {std::get<0>(r), dg, db};
}

}  // namespace

TORCH_LIBRARY(sam_hip, m) {
  m.def("linear(Tensor x, Tensor w, Tensor? bias, int epilogue, Tensor? residual, Tensor? aux_in, bool want_aux_out, float p_drop, int seed, int offset, "
        "bool b_kcontig, bool out_f32) -> (Tensor, Tensor)");
  m.def("spatial_attn_fwd(Tensor qkv, Tensor allow, int batch, int heads, float scale, float p_drop, int seed, int offset) -> (Tensor, Tensor, Tensor)");
  m.def("spatial_attn_bwd(Tensor dout, Tensor qkv, Tensor lse2, Tensor allow, Tensor keep, int batch, int heads, float scale, float p_drop) -> Tensor");
  m.def("spatial_attn_fwd_train(Tensor qkv, Tensor allow, int batch, int heads, float scale, float p_drop, int seed, int offset) -> (Tensor, Tensor, Tensor, Tensor)");
  m.def("spatial_attn_bwd_fused(Tensor dout, Tensor qkv, Tensor out, Tensor out_lo, Tensor lse2, Tensor allow, Tensor keep, int batch, int heads, float scale, float p_drop) -> Tensor");
  m.def("layernorm_fwd(Tensor x, Tensor gamma, Tensor beta, float eps) -> (Tensor, Tensor, Tensor)");
  m.def("layernorm_bwd(Tensor dy, Tensor x, Tensor mean, Tensor rstd, Tensor gamma) -> (Tensor, Tensor, Tensor)");
  m.def("set_ln_defer(bool on) -> ()");
  m.def("ln_finalize_flush() -> ()");
  m.def("ln_finalize_clear() -> ()");
  m.def("pack_masks(Tensor question_mask, Tensor obj_mask, Tensor ocr_mask) -> (Tensor, Tensor, Tensor)");
  m.def("mask_bits_prefix_lm(Tensor key_valid, int n_dec) -> Tensor");
  m.def("mask_bits_from_additive(Tensor mask) -> Tensor");
  m.def("pack_relations(Tensor base_bits, Tensor adj, int n_txt, int heads, int quadrant_bits) -> Tensor");
  m.def("pack_relations_bhnn(Tensor rel, Tensor? base_bits) -> Tensor");
  m.def("ptr_scores(Tensor q, Tensor k, Tensor ocr_mask, float scale) -> Tensor");
  m.def("ptr_scores_bwd(Tensor dscores, Tensor q, Tensor k, float scale) -> (Tensor, Tensor)");
  m.def("bce_loss(Tensor fixed, Tensor ocr, Tensor targets, Tensor loss_mask, float grad_scale, Tensor? global_count) -> (Tensor, Tensor, Tensor)");
  m.def("sumsq(Tensor g, Tensor(a!) out, int[] sparse_region, Tensor? touched) -> ()");
  m.def("adam_step(Tensor(a!) p, Tensor(e!) g, Tensor(b!) m, Tensor(c!) v, Tensor(d!)? p_bf16, int[] seg_end, float[] seg_lr, int step, float beta1, float beta2, float eps, "
        "Tensor? gnorm_sq, float max_norm, Tensor? dev_sched, int[] sparse_region, Tensor? touched) -> ()");
  m.def("step_advance(Tensor(a!)? rng_state, int offset_stride, Tensor(b!) step_counter, float[] base_lr, int warmup_iters, float warmup_factor, int[] decay_iters, "
        "float lr_decay, float beta1, float beta2, Tensor(c!) dev_sched) -> ()");
  m.def("encoder_layer_fwd(Tensor x, Tensor allow, Tensor[] params, int batch, int heads, float scale, float p_attn, float p_hid, int[] seeds, float eps1, "
        "float eps2) -> Tensor[]");
  m.def("encoder_layer_bwd(Tensor dy, Tensor[] saved, Tensor allow, Tensor[] params, Tensor(a!)[] grads, int batch, int heads, float scale, float p_attn, "
        "float p_hid, int[] seeds, bool need_dx, bool accumulate) -> Tensor");
  m.def("encoder_layer_bwd_nowgrad(Tensor dy, Tensor[] saved, Tensor allow, Tensor[] params, Tensor(a!)[] grads, int batch, int heads, float scale, float p_attn, "
        "float p_hid, int[] seeds, bool need_dx, bool accumulate) -> Tensor[]");
  m.def("wgrad_grouped(Tensor[] dys, Tensor[] xs, Tensor(a!)[] dws, Tensor(b!)[] dbs, bool accumulate) -> ()");
}

TORCH_LIBRARY_IMPL(sam_hip, CompositeExplicitAutograd, m) {      // no tensor arguments to dispatch on
  m.impl("set_ln_defer", set_ln_defer);
  m.impl("ln_finalize_flush", ln_finalize_flush);
  m.impl("ln_finalize_clear", ln_finalize_clear);
}

TORCH_LIBRARY_IMPL(sam_hip, CUDA, m) {      // (the ROCm backend registers under the CUDA dispatch key)
  m.impl("linear", linear_op);
  m.impl("spatial_attn_fwd", attn_fwd);
  m.impl("spatial_attn_bwd", attn_bwd);
  m.impl("spatial_attn_fwd_train", attn_fwd_train);
  m.impl("spatial_attn_bwd_fused", attn_bwd_fused);
  m.impl("layernorm_fwd", layernorm_fwd_op);
  m.impl("layernorm_bwd", layernorm_bwd_op);
  m.impl("encoder_layer_fwd", encoder_layer_fwd);
  m.impl("encoder_layer_bwd", encoder_layer_bwd);
  m.impl("encoder_layer_bwd_nowgrad", encoder_layer_bwd_nowgrad);
  m.impl("wgrad_grouped", wgrad_grouped_op);
  m.impl("pack_masks", pack_masks);
  m.impl("mask_bits_prefix_lm", mask_bits_prefix_lm);
  m.impl("mask_bits_from_additive", mask_bits_from_additive);
  m.impl("pack_relations", pack_relations);
  m.impl("pack_relations_bhnn", pack_relations_bhnn);
  m.impl("ptr_scores", ptr_scores_fwd);
  m.impl("ptr_scores_bwd", ptr_scores_bwd);
  m.impl("bce_loss", bce_loss);
  m.impl("sumsq", sumsq);
  m.impl("adam_step", adam_step);
  m.impl("step_advance", step_advance);
}
