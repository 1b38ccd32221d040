// debugging aid (tools/debug/step_stamps.py): one-thread kernel that stores the device's constant 100 MHz clock into a slot.  Captured into the training
// step's hipGraph at chosen points, it shows when those points are reached in an UNPROFILED replay (rocprofv3's interception makes hipGraphLaunch block).
#include <hip/hip_runtime.h>
#include <cstdint>
__global__ void stamp_kernel(uint64_t* slot) { *slot = wall_clock64(); }
extern "C" int probe_stamp(uint64_t* slot, void* stream) {
  stamp_kernel<<<1, 1, 0, (hipStream_t)stream>>>(slot);
  return (int)hipGetLastError();
}
