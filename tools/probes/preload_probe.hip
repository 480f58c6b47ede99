// Is kernel-argument PRELOAD honoured by this device's firmware / runtime?  (r6; cc_attn_decode_kernels.h, CC_V_PRELOAD.)
//
// A kernel built with -mllvm -amdgpu-kernarg-preload-count=N starts with a 256-byte compatibility prologue that s_loads the same
// arguments into the same SGPRs and branches to the real entry: firmware that preloads enters 256 bytes further down, firmware that
// does not runs the prologue — indistinguishable from inside the kernel.  This probe makes them distinguishable: it loads its own
// code object from a file, PATCHES the prologue's first s_load to read the SECOND argument's bytes (offset 8 instead of 0), and
// launches: `out[0] = a` comes back as `a` when the prologue was skipped (preload active) and as `b` when it ran.
//
//   hipcc --offload-arch=gfx950 -O2 -mllvm -amdgpu-kernarg-preload-count=4 --cuda-device-only -o /tmp/preload_kernel.co tools/probes/preload_probe.hip
//   hipcc --offload-arch=gfx950 -O2 -DHOST -o tools/probes/preload_probe tools/probes/preload_probe.hip && tools/probes/preload_probe /tmp/preload_kernel.co
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#ifndef HOST
extern "C" __global__ void preload_kernel(unsigned long long a, unsigned long long b, unsigned long long* out) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    out[0] = a;
    out[1] = b;
  }
}
#else
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 2; } } while (0)
int main(int argc, char** argv) {
  if (argc < 2) return 1;
  FILE* f = fopen(argv[1], "rb");
  if (!f) return 1;
  std::vector<unsigned char> co;
  unsigned char buf[4096];
  size_t n;
  while ((n = fread(buf, 1, sizeof buf, f)) > 0) co.insert(co.end(), buf, buf + n);
  fclose(f);
  // the prologue's first instruction: s_load_dwordx2 s[N:N+1], s[0:1], 0x0 — SMEM encoding, 8 bytes: low dword = op | sdata | sbase,
  // high dword = offset (21 bits).  Found by pattern: an SMEM s_load_dwordx2 (op 1, encoding 0xC0040000 | sdata << 6 | sbase) with
  // offset 0 followed within 32 bytes by s_waitcnt and s_branch; we simply patch the FIRST s_load_dwordx2 from s[0:1] with offset 0
  // in the file's text to offset 8.
  int patched = 0;
  for (size_t i = 0; i + 8 <= co.size() && !patched; i += 4) {
    unsigned lo, hi;
    memcpy(&lo, &co[i], 4);
    memcpy(&hi, &co[i + 4], 4);
    const bool smem = (lo >> 26) == 0x30;            // SMEM encoding 110000
    const unsigned op = (lo >> 18) & 0xff;           // 1 = s_load_dwordx2
    const unsigned sbase = lo & 0x3f;                // s[0:1] -> 0
    const bool imm = (lo >> 17) & 1;
    if (smem && op == 1 && sbase == 0 && imm && (hi & 0x1fffff) == 0) {
      hi |= 8;
      memcpy(&co[i + 4], &hi, 4);
      patched = 1;
      printf("patched s_load_dwordx2 at file offset %zu: offset 0 -> 8\n", i);
    }
  }
  if (!patched) { printf("no s_load_dwordx2 s[..], s[0:1], 0x0 found: nothing patched\n"); return 3; }
  hipModule_t mod;
  hipFunction_t fn;
  CK(hipModuleLoadData(&mod, co.data()));
  CK(hipModuleGetFunction(&fn, mod, "preload_kernel"));
  unsigned long long* out;
  CK(hipMalloc(&out, 16));
  CK(hipMemset(out, 0, 16));
  unsigned long long a = 0x1111111111111111ull, b = 0x2222222222222222ull;
  void* args[] = {&a, &b, &out};
  CK(hipModuleLaunchKernel(fn, 1, 1, 1, 64, 1, 1, 0, nullptr, args, nullptr));
  CK(hipDeviceSynchronize());
  unsigned long long h[2];
  CK(hipMemcpy(h, out, 16, hipMemcpyDeviceToHost));
  printf("out[0] = %llx (a = %llx, b = %llx), out[1] = %llx\n", h[0], a, b, h[1]);
  printf(h[0] == a ? "PRELOAD ACTIVE: the compatibility prologue was skipped\n" : (h[0] == b ? "PRELOAD INACTIVE: the compatibility prologue ran\n" : "UNEXPECTED\n"));
  // and through a captured graph
  hipStream_t st;
  CK(hipStreamCreate(&st));
  hipGraph_t g;
  hipGraphExec_t ge;
  CK(hipMemset(out, 0, 16));
  CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
  CK(hipModuleLaunchKernel(fn, 1, 1, 1, 64, 1, 1, 0, st, args, nullptr));
  CK(hipStreamEndCapture(st, &g));
  CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  CK(hipGraphLaunch(ge, st));
  CK(hipStreamSynchronize(st));
  CK(hipMemcpy(h, out, 16, hipMemcpyDeviceToHost));
  printf(h[0] == a ? "graph replay: PRELOAD ACTIVE\n" : (h[0] == b ? "graph replay: PRELOAD INACTIVE\n" : "graph replay: UNEXPECTED\n"));
  return 0;
}
#endif
