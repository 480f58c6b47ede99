// Empirical layout probe for gfx950 ds_read_b64_tr_b16 (LDS transpose read) and v_mfma_f32_16x16x16_bf16.
// Fills LDS with element index values, every lane supplies its own 8-byte-aligned address, dumps what each lane gets.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void probe(int mode, int* out) {
  __shared__ __attribute__((aligned(16))) uint16_t lds[4096];
  const int l = threadIdx.x;
  for (int i = l; i < 4096; i += 64) lds[i] = (uint16_t)i;
  __syncthreads();
  // address supplied by lane l: mode 0: lane i of each 16-group -> row (i/4), cols 4*(i%4) of a row-major [16 rows][128 cols] image,
  //                              group g = l/16 reads rows 4g .. 4g+3
  const int g = l >> 4, i = l & 15;
  int elem;
  if (mode == 0) elem = (4 * g + (i >> 2)) * 128 + 4 * (i & 3);
  else if (mode == 1) elem = (4 * g + (i & 3)) * 128 + 4 * (i >> 2);
  else elem = l * 4;
  s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(lds + elem));
  for (int j = 0; j < 4; j++) out[l * 4 + j] = (int)(uint16_t)v[j];
}

int main() {
  int* d;
  hipMalloc(&d, 64 * 4 * sizeof(int));
  int h[256];
  for (int mode = 0; mode < 3; mode++) {
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, mode, d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("mode %d\n", mode);
    for (int l = 0; l < 64; l++) {
      printf("lane %2d:", l);
      for (int j = 0; j < 4; j++) printf(" (r%d,c%d)", h[l * 4 + j] / 128, h[l * 4 + j] % 128);
      printf("\n");
      if (l == 19) { printf("...\n"); l = 47; }
    }
  }
  return 0;
}
