// When does the FIRST byte of a kernel arrive, and does it matter when it was requested?  (r6)
//
// The layer step's traces (profiles/r06_*trace*) show its K rows landing ~2.9 us after a workgroup's first instruction whether they
// were requested 0.86 or 0.46 us after it (kernel-argument preload moved the request, nothing else moved).  This probe isolates the
// effect: a graph of [writer, probe] pairs — the writer streams W bytes (read + write, so that it leaves dirty lines like a real
// predecessor), the probe runs one wave per CU; lane 0..63 of each wave waits `delay` (0 .. 2 us) after its first instruction,
// then loads 64 x 16 B from a cold address (never touched by the writer, a fresh 4 KiB-strided region per launch) and stamps the
// arrival.  Output: per delay, the mean / min / max over the waves of (request -> arrival) and (entry -> arrival).
//   hipcc --offload-arch=gfx950 -O2 -o tools/probes/first_byte_probe tools/probes/first_byte_probe.hip && tools/probes/first_byte_probe
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 2; } } while (0)

__global__ void writer(uint4* buf, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    uint4 v = buf[i];
    v.x += 1;
    buf[i] = v;
  }
}

// n_loads 16-byte loads per lane (all issued before the wait): 1 = pure latency, 8 = the step's K + V tile per lane
template <int NL>
__global__ void probe(const uint4* cold, size_t stride16, int delay_ticks, unsigned long long* out, int slot) {
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();  // 100 MHz
  unsigned long long t = t0;
  while ((long long)(t - t0) < delay_ticks) t = __builtin_amdgcn_s_memrealtime();
  const uint4* p = cold + (size_t)blockIdx.x * stride16 + threadIdx.x;
  const unsigned long long tq = __builtin_amdgcn_s_memrealtime();
  typedef unsigned int u32x4_nt __attribute__((ext_vector_type(4)));
  u32x4_nt v[NL];
#pragma unroll
  for (int i = 0; i < NL; i++) v[i] = __builtin_nontemporal_load(reinterpret_cast<const u32x4_nt*>(p + (size_t)i * 64));
  unsigned x = 0;
#pragma unroll
  for (int i = 0; i < NL; i++) x ^= v[i].x ^ v[i].y ^ v[i].z ^ v[i].w;
  asm volatile("s_waitcnt vmcnt(0)" ::"v"(x));
  const unsigned long long ta = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0) {
    unsigned long long* o = out + ((size_t)slot * gridDim.x + blockIdx.x) * 4;
    o[0] = t0;
    o[1] = tq;
    o[2] = ta;
    o[3] = x;
  }
}

int main() {
  const int n_cu = 256, reps = 12;
  const int delays_us10[] = {0, 2, 5, 10, 15, 20, 30};  // tenths of a microsecond
  const int nd = sizeof(delays_us10) / sizeof(int);
  const size_t wbytes = 16u << 20;
  uint4 *wbuf, *cold;
  const size_t cold_bytes = (size_t)2 << 30;  // 2 GiB: every launch reads a region nobody touched since the allocation's memset
  CK(hipMalloc(&wbuf, wbytes));
  CK(hipMalloc(&cold, cold_bytes));
  CK(hipMemset(wbuf, 1, wbytes));
  CK(hipMemset(cold, 1, cold_bytes));
  unsigned long long* out;
  const int slots = nd * reps * 2;
  CK(hipMalloc(&out, (size_t)slots * n_cu * 4 * 8));
  hipStream_t st;
  CK(hipStreamCreate(&st));
  for (int nl_sel = 0; nl_sel < 2; nl_sel++) {
    const int NL = nl_sel ? 8 : 1;
    hipGraph_t g;
    hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
    int slot = 0;
    size_t region = 0;
    const size_t stride16 = (NL * 1024 + 4096) / 16;  // per-wave region: its NL KiB + a page of distance
    for (int r = 0; r < reps; r++)
      for (int d = 0; d < nd; d++) {
        hipLaunchKernelGGL(writer, dim3(1024), dim3(256), 0, st, wbuf, wbytes / 16);
        const uint4* base = cold + region;
        region += stride16 * n_cu;
        if ((region + stride16 * n_cu) * 16 > cold_bytes) region = 0;
        if (NL == 1) hipLaunchKernelGGL(probe<1>, dim3(n_cu), dim3(64), 0, st, base, stride16, delays_us10[d] * 10, out, slot);
        else hipLaunchKernelGGL(probe<8>, dim3(n_cu), dim3(64), 0, st, base, stride16, delays_us10[d] * 10, out, slot);
        slot++;
      }
    CK(hipStreamEndCapture(st, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    CK(hipGraphLaunch(ge, st));
    CK(hipStreamSynchronize(st));
    std::vector<unsigned long long> h((size_t)slot * n_cu * 4);
    CK(hipMemcpy(h.data(), out, h.size() * 8, hipMemcpyDeviceToHost));
    printf("loads per lane = %d (one wave per CU, %d CUs; writer of %zu MiB in front of every probe; 100 MHz clock: 0.01 us resolution)\n", NL, n_cu, wbytes >> 20);
    printf("delay_us  request->arrival us (min mean max)   entry->arrival us (min mean max)   first wave's entry -> last wave's arrival\n");
    for (int d = 0; d < nd; d++) {
      double s1 = 0, s2 = 0, mn1 = 1e9, mx1 = 0, mn2 = 1e9, mx2 = 0, span = 0;
      int n = 0;
      for (int r = 1; r < reps; r++) {  // (the first repetition warms the code caches)
        const int sl = r * nd + d;
        unsigned long long e0 = ~0ull, a1 = 0;
        for (int b = 0; b < n_cu; b++) {
          const unsigned long long* o = &h[((size_t)sl * n_cu + b) * 4];
          const double ra = (double)(o[2] - o[1]) / 100.0, ea = (double)(o[2] - o[0]) / 100.0;
          s1 += ra; s2 += ea; n++;
          mn1 = std::min(mn1, ra); mx1 = std::max(mx1, ra); mn2 = std::min(mn2, ea); mx2 = std::max(mx2, ea);
          e0 = std::min(e0, o[0]); a1 = std::max(a1, o[2]);
        }
        span += (double)(a1 - e0) / 100.0;
      }
      printf("%6.1f    %6.2f %6.2f %6.2f                 %6.2f %6.2f %6.2f                  %6.2f\n", delays_us10[d] / 10.0, mn1, s1 / n, mx1, mn2, s2 / n, mx2,
             span / (reps - 1));
    }
    CK(hipGraphExecDestroy(ge));
    CK(hipGraphDestroy(g));
  }
  return 0;
}
