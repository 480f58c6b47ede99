// stream_occ_probe.hip — how fast can ONE launch stream B bytes from HBM as a function of its geometry (r5, not part of the product)?
// The QKV form of the layer step is bound to one 8-wave workgroup per CU (its LDS); the stand-alone GEMV runs 8 x 4 waves per CU.
// Every lane issues M 16-byte non-temporal loads up front (unrolled), xors them, loops if its share is larger.  Rotates over
// distinct buffers (> 256 MB) so that the Infinity Cache serves nothing; timed by HIP events around 32 back-to-back launches.
//   hipcc --offload-arch=gfx950 -O3 -o tools/probes/stream_occ_probe tools/probes/stream_occ_probe.hip && tools/probes/stream_occ_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CK(x)                                                                          \
  do {                                                                                 \
    hipError_t e = (x);                                                                \
    if (e != hipSuccess) {                                                             \
      printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__);     \
      exit(1);                                                                         \
    }                                                                                  \
  } while (0)

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// grid x block threads; the launch covers n16 16-byte chunks; thread t of the grid takes chunks t + k * total_threads * ... in
// groups of M consecutive "rounds" (round r of the group: chunk (g * M + r) * T + t — every load instruction of a wave is one
// contiguous 1 KiB segment)
template <int M, int LDS_KB>
__global__ void stream_kernel(const u32x4* __restrict__ src, size_t n16, unsigned* out) {
  __shared__ unsigned pad[LDS_KB * 256 + 1];
  const size_t T = (size_t)gridDim.x * blockDim.x;
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  unsigned x = 0;
  for (size_t g = 0; (g * M) * T + t < n16; g++) {
    u32x4 r[M];
#pragma unroll
    for (int i = 0; i < M; i++) {
      const size_t c = (g * M + i) * T + t;
      r[i] = __builtin_nontemporal_load(src + (c < n16 ? c : t));
    }
#pragma unroll
    for (int i = 0; i < M; i++) x ^= r[i].x ^ r[i].y ^ r[i].z ^ r[i].w;
  }
  if (LDS_KB > 0 && threadIdx.x == 0) pad[x & 255] = x;  // (keeps the allocation)
  if (x == 0x9e3779b9u) out[0] = x + (LDS_KB > 0 ? pad[1] : 0u);
}

template <int M, int LDS_KB>
static float run(const char* base, size_t bytes, int n_buf, int grid, int block, unsigned* out, hipStream_t st) {
  const size_t n16 = bytes / 16;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  float best = 1e9f;
  for (int rep = 0; rep < 5; rep++) {
    CK(hipEventRecord(e0, st));
    for (int i = 0; i < 32; i++)
      hipLaunchKernelGGL((stream_kernel<M, LDS_KB>), dim3(grid), dim3(block), 0, st,
                         reinterpret_cast<const u32x4*>(base + (size_t)((rep * 32 + i) % n_buf) * bytes), n16, out);
    CK(hipEventRecord(e1, st));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) best = ms;
  }
  return best * 1000.f / 32.f;
}

int main() {
  const size_t sizes[] = {17u << 20, 50u << 20, 67u << 20, 100u << 20, 235u << 20};
  const size_t pool = (size_t)1536 << 20;
  char* buf;
  unsigned* out;
  CK(hipMalloc(&buf, pool));
  CK(hipMemset(buf, 1, pool));
  CK(hipMalloc(&out, 256));
  hipStream_t st;
  CK(hipStreamCreate(&st));
  printf("bytes_MB grid block M lds_kb us TB/s\n");
  for (size_t bytes : sizes) {
    const int n_buf = (int)(pool / bytes);
#define RUN(M_, L_, G_, B_)                                                                                              \
  {                                                                                                                      \
    const float us = run<M_, L_>(buf, bytes, n_buf, G_, B_, out, st);                                                    \
    printf("%zu %d %d %d %d %.2f %.2f\n", bytes >> 20, G_, B_, M_, L_, us, (double)bytes / us / 1e6);                     \
  }
    RUN(8, 0, 256, 512)
    RUN(16, 0, 256, 512)
    RUN(32, 0, 256, 512)
    RUN(48, 0, 256, 512)
    RUN(16, 0, 256, 1024)
    RUN(32, 0, 256, 1024)
    RUN(8, 0, 512, 256)
    RUN(16, 0, 512, 256)
    RUN(32, 0, 512, 256)
    RUN(4, 0, 2048, 256)
    RUN(8, 0, 2048, 256)
    RUN(16, 0, 2048, 256)
    RUN(8, 0, 1024, 512)
    RUN(16, 0, 512, 512)
    RUN(16, 0, 512, 1024)
    fflush(stdout);
  }
  return 0;
}
