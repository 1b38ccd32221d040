// Hardware-layout probe for gfx950 (run on the GPU box; prints PASS/FAIL per hypothesis).
// Not part of the product path: it pins the MFMA operand/result lane maps and the
// ds_read_b64_tr_b16 shuffle that the kernels in sam-textvqa_amd/csrc rely on.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <cmath>
#include <vector>
typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

static unsigned short f2bf(float f) { unsigned u; std::memcpy(&u, &f, 4); u += 0x7fff + ((u >> 16) & 1); return u >> 16; }
static float bf2f(unsigned short h) { unsigned u = ((unsigned)h) << 16; float f; std::memcpy(&f, &u, 4); return f; }

__global__ void k_mfma16(const unsigned short* A, const unsigned short* B, float* D) {
  // A: [64 lanes][8], B: [64][8] raw operand registers; D: [64][4]
  int l = threadIdx.x;
  bf16x8 a, b;
  for (int e = 0; e < 8; ++e) { a[e] = A[l * 8 + e]; b[e] = B[l * 8 + e]; }
  f32x4 c = {0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
  for (int r = 0; r < 4; ++r) D[l * 4 + r] = c[r];
}
__global__ void k_mfma32(const unsigned short* A, const unsigned short* B, float* D) {
  int l = threadIdx.x;
  bf16x8 a, b;
  for (int e = 0; e < 8; ++e) { a[e] = A[l * 8 + e]; b[e] = B[l * 8 + e]; }
  f32x16 c;
  for (int r = 0; r < 16; ++r) c[r] = 0;
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  for (int r = 0; r < 16; ++r) D[l * 16 + r] = c[r];
}
__global__ void k_tr(const int* addr_elems, unsigned short* out) {
  __shared__ __attribute__((aligned(16))) unsigned short lds[4096];
  int l = threadIdx.x;
  for (int i = l; i < 4096; i += 64) lds[i] = (unsigned short)i;
  __syncthreads();
  unsigned a = (unsigned)(size_t)(&lds[addr_elems[l]]);
  s16x4 v;
  asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a) : "memory");
  for (int e = 0; e < 4; ++e) out[l * 4 + e] = (unsigned short)v[e];
}
__global__ void k_swap(const int* in, int* out) {
  int l = threadIdx.x;
  int v = in[l];
  auto r32 = __builtin_amdgcn_permlane32_swap(v, v, false, false);
  auto r16 = __builtin_amdgcn_permlane16_swap(v, v, false, false);
  out[l * 4 + 0] = r32[0]; out[l * 4 + 1] = r32[1]; out[l * 4 + 2] = r16[0]; out[l * 4 + 3] = r16[1];
}

int main() {
  srand(1);
  // ---------------- 16x16x32 ----------------
  {
    float Am[16][32], Bm[32][16];
    for (auto& r : Am) for (auto& x : r) x = bf2f(f2bf((rand() % 200 - 100) / 64.0f));
    for (auto& r : Bm) for (auto& x : r) x = bf2f(f2bf((rand() % 200 - 100) / 64.0f));
    std::vector<unsigned short> A(512), B(512);
    for (int l = 0; l < 64; ++l) for (int e = 0; e < 8; ++e) {
      int k = 8 * (l >> 4) + e;
      A[l * 8 + e] = f2bf(Am[l & 15][k]);
      B[l * 8 + e] = f2bf(Bm[k][l & 15]);
    }
    unsigned short *dA, *dB; float* dD;
    hipMalloc(&dA, 1024); hipMalloc(&dB, 1024); hipMalloc(&dD, 64 * 4 * 4);
    hipMemcpy(dA, A.data(), 1024, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), 1024, hipMemcpyHostToDevice);
    k_mfma16<<<1, 64>>>(dA, dB, dD);
    std::vector<float> D(256);
    hipMemcpy(D.data(), dD, 1024, hipMemcpyDeviceToHost);
    double maxerr = 0;
    for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r) {
      int i = 4 * (l >> 4) + r, j = l & 15;
      double ref = 0; for (int k = 0; k < 32; ++k) ref += (double)Am[i][k] * Bm[k][j];
      maxerr = fmax(maxerr, fabs(ref - D[l * 4 + r]));
    }
    printf("MFMA16x16x32 hypothesis (A[l&15][8*(l>>4)+e], B[8*(l>>4)+e][l&15], D[4*(l>>4)+r][l&15]): maxerr=%g %s\n", maxerr, maxerr < 1e-3 ? "PASS" : "FAIL");
  }
  // ---------------- 32x32x16 ----------------
  {
    float Am[32][16], Bm[16][32];
    for (auto& r : Am) for (auto& x : r) x = bf2f(f2bf((rand() % 200 - 100) / 64.0f));
    for (auto& r : Bm) for (auto& x : r) x = bf2f(f2bf((rand() % 200 - 100) / 64.0f));
    std::vector<unsigned short> A(512), B(512);
    for (int l = 0; l < 64; ++l) for (int e = 0; e < 8; ++e) {
      int k = 8 * (l >> 5) + e;
      A[l * 8 + e] = f2bf(Am[l & 31][k]);
      B[l * 8 + e] = f2bf(Bm[k][l & 31]);
    }
    unsigned short *dA, *dB; float* dD;
    hipMalloc(&dA, 1024); hipMalloc(&dB, 1024); hipMalloc(&dD, 64 * 16 * 4);
    hipMemcpy(dA, A.data(), 1024, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), 1024, hipMemcpyHostToDevice);
    k_mfma32<<<1, 64>>>(dA, dB, dD);
    std::vector<float> D(1024);
    hipMemcpy(D.data(), dD, 4096, hipMemcpyDeviceToHost);
    double maxerr = 0;
    for (int l = 0; l < 64; ++l) for (int r = 0; r < 16; ++r) {
      int i = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5), j = l & 31;
      double ref = 0; for (int k = 0; k < 16; ++k) ref += (double)Am[i][k] * Bm[k][j];
      maxerr = fmax(maxerr, fabs(ref - D[l * 16 + r]));
    }
    printf("MFMA32x32x16 hypothesis (A[l&31][8*(l>>5)+e], B[8*(l>>5)+e][l&31], D[(r&3)+8*(r>>2)+4*(l>>5)][l&31]): maxerr=%g %s\n", maxerr, maxerr < 1e-3 ? "PASS" : "FAIL");
  }
  // ---------------- ds_read_b64_tr_b16 ----------------
  for (int pat = 0; pat < 3; ++pat) {
    std::vector<int> addr(64);
    for (int l = 0; l < 64; ++l) {
      if (pat == 0) addr[l] = 4 * l;                 // lane-linear 8-byte chunks
      else if (pat == 1) addr[l] = 4 * (63 - l);     // reversed
      else addr[l] = (l >> 2) * 100 + (l & 3) * 4;   // "row" r=l>>2 with stride 100 elems (8B aligned: 200B)
    }
    int* dAd; unsigned short* dO;
    hipMalloc(&dAd, 256); hipMalloc(&dO, 512);
    hipMemcpy(dAd, addr.data(), 256, hipMemcpyHostToDevice);
    k_tr<<<1, 64>>>(dAd, dO);
    std::vector<unsigned short> O(256);
    hipMemcpy(O.data(), dO, 512, hipMemcpyDeviceToHost);
    // hypothesis: within each 16-lane group G, out lane i elem j = element (i&3) of the chunk supplied by lane G*16 + 4*j + (i>>2)
    int bad = 0;
    for (int l = 0; l < 64; ++l) for (int j = 0; j < 4; ++j) {
      int G = l >> 4, i = l & 15;
      int src_lane = G * 16 + 4 * j + (i >> 2);
      int expect = addr[src_lane] + (i & 3);
      if (O[l * 4 + j] != expect) ++bad;
    }
    printf("TR16 pattern %d hypothesis: %s (bad=%d)\n", pat, bad ? "FAIL" : "PASS", bad);
    if (bad || pat == 0) {
      for (int l = 0; l < 64; ++l) printf("  lane %2d: %4d %4d %4d %4d\n", l, O[l * 4], O[l * 4 + 1], O[l * 4 + 2], O[l * 4 + 3]);
    }
  }
  // ---------------- permlane swaps ----------------
  {
    std::vector<int> in(64); for (int l = 0; l < 64; ++l) in[l] = l;
    int *dI, *dO; hipMalloc(&dI, 256); hipMalloc(&dO, 1024);
    hipMemcpy(dI, in.data(), 256, hipMemcpyHostToDevice);
    k_swap<<<1, 64>>>(dI, dO);
    std::vector<int> O(256); hipMemcpy(O.data(), dO, 1024, hipMemcpyDeviceToHost);
    printf("permlane32_swap(v,v): r0 / r1 ; permlane16_swap(v,v): r0 / r1\n");
    for (int l = 0; l < 64; ++l) printf("  lane %2d: %2d %2d | %2d %2d\n", l, O[l * 4], O[l * 4 + 1], O[l * 4 + 2], O[l * 4 + 3]);
  }
  hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
  printf("device %s CUs=%d clock=%d kHz lds/block=%zu\n", p.gcnArchName, p.multiProcessorCount, p.clockRate, p.sharedMemPerBlock);
  return 0;
}
