// Host-side runtime bits of libsam_hip.so: error string, version, device query.
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <atomic>
#include <stdio.h>

static thread_local char g_err[512] = "";

extern "C" void sam_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
extern "C" const char* sam_last_error(void) { return g_err; }
extern "C" int sam_abi_version(void) { return 7; }    // 7: sam_adam_step_range (the update in pieces, gated); 6: sam_attn_fwd_train / sam_attn_bwd_fused (one-pass attention backward); 5: sam_greedy_decode_steps (sam_decode_desc); 4: row-sparse regions (sam_sparse_rows) in sam_sumsq_f32 / sam_adam_step[_dev], `touched` flags in sam_embedding_bwd[_sorted]
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
