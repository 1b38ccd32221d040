// Stand-alone C++ host program against include/sam_hip.h + libsam_hip.so — no Python, no torch: the same C ABI a
// maintainer of the reference would bind.  Runs a LayerNorm, a bias GEMM and a prefix-LM-masked
// attention forward on the GPU and checks each against a plain double-precision host computation.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>
#include "sam_hip.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 2; } } while (0)
#define SAM(x) do { int r_ = (x); if (r_ != 0) { std::printf("sam error %d (%s) at %s:%d\n", r_, sam_last_error(), __FILE__, __LINE__); return 3; } } while (0)

static uint16_t f2bf(float f) { uint32_t u; std::memcpy(&u, &f, 4); u += 0x7fffu + ((u >> 16) & 1u); return (uint16_t)(u >> 16); }
static float bf2f(uint16_t h) { uint32_t u = (uint32_t)h << 16; float f; std::memcpy(&f, &u, 4); return f; }
static float rnd(uint32_t& s) { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 32768.0f - 1.0f; }
template <typename T> static T* dev(const std::vector<T>& h) { T* d = nullptr; if (hipMalloc(&d, h.size() * sizeof(T)) != hipSuccess) return nullptr; (void)hipMemcpy(d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice); return d; }
// |got - ref| <= 1e-3 * max|ref| + one bf16 ulp of ref  (the bound of tests/util.py)
static bool close(const std::vector<float>& got, const std::vector<double>& ref, const char* what) {
  double mx = 0, worst = 0;
  for (double r : ref) mx = std::fmax(mx, std::fabs(r));
  for (size_t i = 0; i < ref.size(); ++i) worst = std::fmax(worst, std::fabs(got[i] - ref[i]) - (1e-3 * mx + std::fabs(ref[i]) / 256.0));
  std::printf("%-14s max|ref| %.4f  bound excess %.3g  %s\n", what, mx, worst, worst <= 0 ? "ok" : "FAIL");
  return worst <= 0;
}

int main() {
  if (sam_abi_version() != 9) { std::printf("unexpected ABI version %d\n", sam_abi_version()); return 1; }
  int cus = 0, lds = 0; char arch[64] = {0};
  SAM(sam_device_info(&cus, &lds, arch, sizeof(arch)));
  std::printf("device: %s, %d CUs, %d B LDS/CU\n", arch, cus, lds);
  uint32_t seed = 12345u;
  bool ok = true;

  {  // ---- BertLayerNorm: x bf16 [M, D]
    const int M = 37, D = 768;
    std::vector<uint16_t> x(M * D); std::vector<float> g(D), b(D);
    for (auto& v : x) v = f2bf(3.0f * rnd(seed));
    for (int i = 0; i < D; ++i) { g[i] = 1.0f + 0.1f * rnd(seed); b[i] = 0.1f * rnd(seed); }
    uint16_t* dx = dev(x); float* dg = dev(g); float* db = dev(b);
    uint16_t* dy; float *dm, *dr; CK(hipMalloc(&dy, M * D * 2)); CK(hipMalloc(&dm, M * 4)); CK(hipMalloc(&dr, M * 4));
    SAM(sam_layernorm_fwd(dx, 0, D, dg, db, 1e-12f, M, D, dy, D, dm, dr, nullptr));
    CK(hipDeviceSynchronize());
    std::vector<uint16_t> y(M * D); CK(hipMemcpy(y.data(), dy, M * D * 2, hipMemcpyDeviceToHost));
    std::vector<double> ref(M * D); std::vector<float> got(M * D);
    for (int r = 0; r < M; ++r) {
      double mu = 0, var = 0;
      for (int c = 0; c < D; ++c) mu += bf2f(x[r * D + c]);
      mu /= D;
      for (int c = 0; c < D; ++c) { double d = bf2f(x[r * D + c]) - mu; var += d * d; }
      var /= D;
      for (int c = 0; c < D; ++c) { ref[r * D + c] = g[c] * (bf2f(x[r * D + c]) - mu) / std::sqrt(var + 1e-12) + b[c]; got[r * D + c] = bf2f(y[r * D + c]); }
    }
    ok &= close(got, ref, "layernorm");
  }

  {  // ---- y = x W^T + bias, fp32 out
    const int M = 200, N = 136, K = 192;
    std::vector<uint16_t> x(M * K), w(N * K); std::vector<float> bias(N);
    for (auto& v : x) v = f2bf(rnd(seed));
    for (auto& v : w) v = f2bf(0.1f * rnd(seed));
    for (auto& v : bias) v = rnd(seed);
    uint16_t *dx = dev(x), *dw = dev(w); float* dbias = dev(bias); float* dc; CK(hipMalloc(&dc, M * N * 4));
    sam_gemm_desc d; std::memset(&d, 0, sizeof(d));
    d.M = M; d.N = N; d.K = K; d.a_kcontig = 1; d.b_kcontig = 1; d.c_is_f32 = 1; d.epilogue = SAM_EPI_BIAS;
    d.A = dx; d.lda = K; d.B = dw; d.ldb = K; d.C = dc; d.ldc = N; d.bias = dbias;
    SAM(sam_gemm_bf16(&d, nullptr));
    CK(hipDeviceSynchronize());
    std::vector<float> got(M * N); CK(hipMemcpy(got.data(), dc, M * N * 4, hipMemcpyDeviceToHost));
    std::vector<double> ref(M * N);
    for (int m = 0; m < M; ++m) for (int n = 0; n < N; ++n) {
      double a = bias[n];
      for (int k = 0; k < K; ++k) a += (double)bf2f(x[m * K + k]) * bf2f(w[n * K + k]);
      ref[m * N + n] = a;
    }
    ok &= close(got, ref, "gemm+bias");
  }

  {  // ---- prefix-LM mask + fused attention forward: B=2, H=2, 30 encoder tokens (some padded) + 10 decoder tokens
    const int B = 2, H = 2, n_enc = 30, n_dec = 10, N = n_enc + n_dec, HD = 64, Dm = H * HD, NW = sam_attn_words_per_row(N);
    std::vector<uint8_t> valid(B * n_enc, 1);
    for (int k = 22; k < n_enc; ++k) valid[1 * n_enc + k] = 0;          // sample 1: 8 padded keys
    std::vector<uint16_t> qkv((size_t)B * N * 3 * Dm);
    for (auto& v : qkv) v = f2bf(rnd(seed));
    uint8_t* dvalid = dev(valid); uint16_t* dqkv = dev(qkv);
    uint32_t* dbits; uint16_t* dout; float* dlse;
    CK(hipMalloc(&dbits, (size_t)B * N * NW * 4)); CK(hipMalloc(&dout, (size_t)B * N * Dm * 2)); CK(hipMalloc(&dlse, (size_t)B * H * N * 4));
    SAM(sam_mask_bits_prefix_lm(dvalid, B, n_enc, n_dec, NW, dbits, nullptr));
    SAM(sam_attn_fwd(dqkv, dbits, (int64_t)N * NW, 0, B, N, H, HD, 0.125f, 0.0f, 0, 0, dout, dlse, nullptr, nullptr));
    CK(hipDeviceSynchronize());
    std::vector<uint16_t> out((size_t)B * N * Dm); CK(hipMemcpy(out.data(), dout, out.size() * 2, hipMemcpyDeviceToHost));
    std::vector<double> ref(out.size()); std::vector<float> got(out.size());
    for (int b = 0; b < B; ++b) for (int h = 0; h < H; ++h) for (int q = 0; q < N; ++q) {
      std::vector<double> s(N); double mx = -1e300;
      for (int k = 0; k < N; ++k) {
        bool allow = k < n_enc ? valid[b * n_enc + k] != 0 : (q >= n_enc && k <= q);       // encoder keys if not padded; decoder keys causally, decoder queries only
        double a = 0;
        for (int d = 0; d < HD; ++d) a += (double)bf2f(qkv[((size_t)(b * N + q)) * 3 * Dm + h * HD + d]) * bf2f(qkv[((size_t)(b * N + k)) * 3 * Dm + Dm + h * HD + d]);
        s[k] = allow ? a * 0.125 : -1e300;
        mx = std::fmax(mx, s[k]);
      }
      double sum = 0;
      for (int k = 0; k < N; ++k) { s[k] = s[k] <= -1e299 ? 0.0 : std::exp(s[k] - mx); sum += s[k]; }
      for (int d = 0; d < HD; ++d) {
        double a = 0;
        for (int k = 0; k < N; ++k) a += s[k] * bf2f(qkv[((size_t)(b * N + k)) * 3 * Dm + 2 * Dm + h * HD + d]);
        const size_t o = ((size_t)(b * N + q)) * Dm + h * HD + d;
        ref[o] = a / sum; got[o] = bf2f(out[o]);
      }
    }
    ok &= close(got, ref, "attention");
  }

  // ---- error contract: bad arguments are rejected with a message, nothing is launched
  if (sam_layernorm_fwd(nullptr, 0, 8, nullptr, nullptr, 1e-12f, 4, 8, nullptr, 8, nullptr, nullptr, nullptr) == 0 || !sam_last_error()[0]) { std::printf("null pointers were accepted\n"); ok = false; }
  std::printf(ok ? "C_ABI_OK\n" : "C_ABI_FAILED\n");
  return ok ? 0 : 4;
}
