// cc_api.hip — ABI bookkeeping entry points (version, error strings, device info).
#include <string.h>

#include "cc_common.h"

extern "C" {

int cc_abi_version(void) {
  CC_ENTRY(); return CC_ABI_VERSION; }

const char* cc_error_string(int code) {
  switch (code) {
    case CC_OK: return "ok";
    case CC_ERR_BAD_ARG: return "bad argument (null pointer, non-positive size or inconsistent shape)";
    case CC_ERR_UNSUPPORTED: return "unsupported dtype / head_dim / window for the built kernels";
    case CC_ERR_HIP: return "HIP launch error";
    case CC_ERR_WORKSPACE: return "workspace too small";
    default: return "unknown error";
  }
}

// Test hook (cc_debug_occupy): workgroups that hold `lds_bytes` of LDS each and do nothing for `microseconds` (s_memrealtime,
// 100 MHz) — placed on as many CUs as there are workgroups, they keep another kernel's workgroups from becoming resident there.
}  // extern "C"
namespace {
__global__ __launch_bounds__(64) void occupy_kernel(unsigned long long ticks, unsigned* sink) {
  extern __shared__ unsigned sm_hold[];
  sm_hold[threadIdx.x] = threadIdx.x;
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(100);
  if (sm_hold[(threadIdx.x + 1) & 63] == 0xffffffffu) sink[0] = 1;  // (keeps the allocation alive)
}
}  // namespace
extern "C" {
int cc_debug_occupy(int32_t n_workgroups, int32_t lds_bytes, int32_t microseconds, void* scratch, cc_stream_t stream) {
  CC_ENTRY();
  if (n_workgroups <= 0 || lds_bytes < 256 || lds_bytes > 160 * 1024 || microseconds <= 0 || microseconds > 5000000 || !scratch)
    return CC_ERR_BAD_ARG;
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(occupy_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes) != hipSuccess)
    return CC_ERR_HIP;
  hipLaunchKernelGGL(occupy_kernel, dim3(n_workgroups), dim3(64), (size_t)lds_bytes, (hipStream_t)stream,
                     (unsigned long long)microseconds * 100ull, reinterpret_cast<unsigned*>(scratch));
  CC_LAUNCH_CHECK();
  return CC_OK;
}

int cc_device_info(int* n_cu, int* wave_size, int* lds_bytes_per_cu, char* name, int name_len) {
  CC_ENTRY();
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return CC_ERR_HIP;
  hipDeviceProp_t p;
  if (hipGetDeviceProperties(&p, dev) != hipSuccess) return CC_ERR_HIP;
  if (n_cu) *n_cu = p.multiProcessorCount;
  if (wave_size) *wave_size = p.warpSize;
  if (lds_bytes_per_cu) *lds_bytes_per_cu = (int)p.maxSharedMemoryPerMultiProcessor;
  if (name && name_len > 0) {
    strncpy(name, p.gcnArchName, (size_t)name_len - 1);
    name[name_len - 1] = 0;
  }
  return CC_OK;
}

}  // extern "C"
