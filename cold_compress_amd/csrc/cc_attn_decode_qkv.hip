// cc_attn_decode_qkv.hip — the single-launch layer step with the layer's QKV projection folded in (r5).
//
// ref: model.py:375-387 (wqkv, split, apply_rotary_emb), :389-427 (update_kv -> attention -> update_state), :452-457 (RMSNorm);
//      cache.py:725-765 (the heavy hitter's eviction), :690-723 (its state update).
//
// Why: at S = 4096 a layer step moves 17.7 MB; any stand-alone launch of that size is bounded by the launch boundary, its
// prologue and the first byte's latency (4.2 us measured for a kernel that only streams the cache, against 2.2 us of transfer at
// 8 TB/s).  The step's K / V tile does not depend on q: here ONE launch requests the weight rows of the projection, then the tile,
// computes q / k_new / v_new (cc_gemv.hip's arithmetic, bit for bit), hands them to the kv head's workgroups through tagged
// granules and runs the step — the tile streams in the shadow of the 50 MB of weights.  The kernel is the QKV = true instantiation
// of decode_attn_split_mfma_kernel (cc_attn_decode_kernels.h): this file only instantiates and launches it.
#include <atomic>
#include <mutex>

#include "cc_attn_decode_kernels.h"
#include "cc_attn_decode_qkv.h"

namespace {

typedef void (*QkvKernel)(CC_LEAD_TYPES SplitArgs);

template <typename T>
static QkvKernel qkv_kernel(int rt, int nw, bool xl2) {
#define CC_QKV_K(RT_, NW_, X_) decode_attn_split_mfma_kernel<T, RT_, NW_, false, true, false, 0, 1, 1, false, X_, true>
  if (nw == 8) {
    if (rt == 4) return xl2 ? CC_QKV_K(4, 8, true) : CC_QKV_K(4, 8, false);
    if (rt == 8) return xl2 ? CC_QKV_K(8, 8, true) : CC_QKV_K(8, 8, false);
  } else if (nw == 4) {
    if (rt == 4) return xl2 ? CC_QKV_K(4, 4, true) : CC_QKV_K(4, 4, false);
    if (rt == 8) return xl2 ? CC_QKV_K(8, 4, true) : CC_QKV_K(8, 4, false);
  }
  return nullptr;
#undef CC_QKV_K
}

static QkvKernel qkv_kernel_dt(int dtype, int rt, int nw, bool xl2) {
  return dtype == CC_DT_BF16 ? qkv_kernel<bf16_t>(rt, nw, xl2) : (dtype == CC_DT_F16 ? qkv_kernel<f16_t>(rt, nw, xl2) : nullptr);
}

}  // namespace

int cc_qkv_step_capacity(int dtype, int rt, int nw, int xl2) {
  const QkvKernel k = qkv_kernel_dt(dtype, rt, nw, xl2 != 0);
  if (!k) return 0;
  // cached per kernel (the same lock-free-reader table as cc_attn_decode.hip's one_capacity)
  struct Entry {
    QkvKernel k;
    int cap;
  };
  static Entry cache[32];
  static std::atomic<int> n_cached{0};
  static std::mutex mu;
  const int n = n_cached.load(std::memory_order_acquire);
  for (int i = 0; i < n; i++)
    if (cache[i].k == k) return cache[i].cap;
  int dev = 0, cus = 0, nb = 0;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess ||
      hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k, nw * 64, 0) != hipSuccess) {
    (void)hipGetLastError();
    return 0;
  }
  const int cap = cus * (nb > 8 ? 8 : nb);
  std::lock_guard<std::mutex> lock(mu);
  const int m = n_cached.load(std::memory_order_relaxed);
  for (int i = 0; i < m; i++)
    if (cache[i].k == k) return cache[i].cap;
  if (m < 32) {
    cache[m] = Entry{k, cap};
    n_cached.store(m + 1, std::memory_order_release);
  }
  return cap;
}

static unsigned long long* g_qkv_trace = nullptr;
// measurement hook (CC_QKV_TRACE builds; a no-op pointer otherwise): [workgroup][16] stamps, see tools/trace_qkv.py
extern "C" void cc_debug_qkv_trace(void* buf) { g_qkv_trace = reinterpret_cast<unsigned long long*>(buf); }

int cc_qkv_step_launch(const void* split_args, size_t split_args_bytes, int dtype, int rt, int nw, int xl2, int grid_x, int grid_y,
                       hipStream_t stream) {
  if (!split_args || split_args_bytes != sizeof(SplitArgs)) return CC_ERR_BAD_ARG;
  const QkvKernel k = qkv_kernel_dt(dtype, rt, nw, xl2 != 0);
  if (!k) return CC_ERR_UNSUPPORTED;
  SplitArgs a;
  __builtin_memcpy(&a, split_args, sizeof(SplitArgs));
  a.qkv.trace = g_qkv_trace;
#if CC_V_PRELOAD
  const dim3 grid = xl2 ? dim3(grid_x * grid_y, 1, 1) : dim3(grid_x, grid_y, 1);  // (XL2: a linear block index, the head count preloaded)
#else
  const dim3 grid(grid_x, grid_y, 1);
#endif
  hipLaunchKernelGGL(k, grid, dim3(nw * 64), 0, stream, CC_LEAD_ARGS(a) a);
  CC_LAUNCH_CHECK();
  return CC_OK;
}
