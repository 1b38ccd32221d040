// Host-side runtime bits of libsam_hip.so: error string, version, device query.
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <atomic>
#include <stdio.h>
#include <stdlib.h>

static thread_local char g_err[512] = "";

extern "C" void sam_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
extern "C" const char* sam_last_error(void) { return g_err; }
extern "C" int sam_abi_version(void) { return 9; }    // 9: sam_ln_fuse.xws / xws_bytes + sam_gemm_ln_ws_bytes (LayerNorm inside an MMT-size GEMM launch); 8: sam_set_cu_reserve / sam_get_cu_reserve (CUs withheld from persistent grids), sam_debug_cu_hog; 7: sam_adam_step_range (the update in pieces, gated); 6: sam_attn_fwd_train / sam_attn_bwd_fused (one-pass attention backward); 5: sam_greedy_decode_steps (sam_decode_desc); 4: row-sparse regions (sam_sparse_rows) in sam_sumsq_f32 / sam_adam_step[_dev], `touched` flags in sam_embedding_bwd[_sorted]
                                                      // 3: sam_step_advance; the grouped-wgrad workspace starts with an error word (layout changed)
                                                      // 2: sam_bce_loss takes global_count; sam_embedding_bwd_sorted; sam_build_digest; sam_gemm_desc.force_tile 1192/1256
#ifndef SAM_BUILD_DIGEST
#define SAM_BUILD_DIGEST "unknown"
#endif
extern "C" const char* sam_build_digest(void) { return SAM_BUILD_DIGEST; }

// process-wide, not thread-local: PyTorch's autograd engine runs the backward launches (which regenerate the forward's masks) on its own
// device thread.  One process drives one GPU in this package.
static std::atomic<const unsigned long long*> g_rng_state{nullptr};
extern "C" void sam_set_rng_state(const unsigned long long* dev_state) { g_rng_state.store(dev_state); }
extern "C" const unsigned long long* sam_get_rng_state(void) { return g_rng_state.load(); }

extern "C" int sam_device_info(int* cu_count, int* lds_per_cu_bytes, char* arch, int arch_len) {
  hipDeviceProp_t p;
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e == hipSuccess) e = hipGetDeviceProperties(&p, dev);
  if (e != hipSuccess) {
    sam_set_error("sam_device_info: %s", hipGetErrorString(e));
    return (int)e;
  }
  if (cu_count) *cu_count = p.multiProcessorCount;
  if (lds_per_cu_bytes) *lds_per_cu_bytes = (int)p.maxSharedMemoryPerMultiProcessor;
  if (arch && arch_len > 0) snprintf(arch, arch_len, "%s", p.gcnArchName);
  return 0;
}

// ---- CUs withheld from the persistent grids (gemm_common.h: grid_cu_count) ---------------------------------------------------------------------------
// Data-parallel training runs RCCL's channel kernels beside the backward pass; a channel block needs a CU's registers / LDS that the step's persistent
// blocks (one per CU) do not leave, so without a reserve the collective and the compute kernels take turns in whole launch rounds.  The reserve is a
// multiple of 8 (each XCD gives up the same number), at most half the device.  Default: SAM_CU_RESERVE from the environment (0 when unset); the
// Trainer sets it when a reducer spanning more than one rank is active (sam_textvqa_amd/trainer.py).  Grids captured into a hipGraph keep the value
// they were captured under.
static std::atomic<int> g_cu_reserve{-1};
extern "C" int sam_get_cu_reserve(void) {
  int r = g_cu_reserve.load();
  if (r < 0) {
    const char* e = getenv("SAM_CU_RESERVE");
    r = e ? atoi(e) : 0;
    if (r < 0) r = 0;
    r -= r % 8;
    g_cu_reserve.store(r);
  }
  return r;
}
extern "C" int sam_set_cu_reserve(int n) {
  if (n < 0 || n > 128) {
    sam_set_error("sam_set_cu_reserve: %d outside [0, 128]", n);
    return 1;
  }
  g_cu_reserve.store(n - n % 8);
  return 0;
}

// A stand-in for a collective's channel kernel (measurement aid, tools/bench_cu_reserve.py): `blocks` workgroups of 256 threads, each holding 64 KB of LDS
// -- no persistent block of the step (>= 112 KB) fits beside it on a CU -- until `ticks` of the 100 MHz wall clock have passed (capped at 0.5 s).
__global__ __launch_bounds__(256) void cu_hog_kernel(unsigned long long ticks, unsigned* sink) {
  extern __shared__ unsigned hog_lds[];
  const unsigned long long t0 = wall_clock64();
  unsigned acc = 0;
  while (wall_clock64() - t0 < ticks) {
    hog_lds[threadIdx.x] = acc;
    acc += hog_lds[(threadIdx.x + 1) & 255];
    __builtin_amdgcn_s_sleep(32);
  }
  if (acc == 0xdeadbeefu && sink) *sink = acc;
}
extern "C" int sam_debug_cu_hog(int blocks, double microseconds, void* stream) {
  if (blocks <= 0) return 0;
  if (blocks > 256 || microseconds < 0 || microseconds > 5e5) {
    sam_set_error("sam_debug_cu_hog: blocks %d / %.0f us outside (0, 256] / [0, 5e5]", blocks, microseconds);
    return 1;
  }
  static bool once = false;
  if (!once) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(cu_hog_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
    once = true;
  }
  cu_hog_kernel<<<dim3(blocks), dim3(256), 64 * 1024, (hipStream_t)stream>>>((unsigned long long)(microseconds * 100.0), nullptr);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    sam_set_error("sam_debug_cu_hog: %s", hipGetErrorString(e));
    return (int)e;
  }
  return 0;
}
