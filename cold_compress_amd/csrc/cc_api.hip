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
