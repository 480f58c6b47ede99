// cc_quant.hip — quantised KV cache (--cache_bits {8,4,2}).  ref: quantization_utils.py:4-98 (axis = 2),
// KVCache.quantize_cache / dequantize_cache cache.py:283-309.
//
// The reference dequantises the WHOLE cache before every update and re-quantises it afterwards, with one
// (scale, zero point) per cache slot shared by all heads — so a slot's grid moves whenever ANY head replaces its
// entry there, and to reproduce its numbers every slot has to go through the quantise -> dequantise round trip
// every step.  cc_kv_requant does that round trip in ONE pass over the model-dtype working cache (the tensors
// the attention kernels read) and emits the quantised image the reference would hold: one workgroup per slot,
// the slot's H rows are read once (coalesced 256-byte rows), min/max through one LDS hop, written back once.
// HBM-bound: (2 + 2 + n/8) bytes per element per step.  Every elementwise op rounds to the cache dtype exactly
// where torch does (include/coldcompress.h).
#include "cc_common.h"

namespace {

constexpr int kQThreads = 256;
constexpr int kQMaxPer = 8;  // elements per thread kept in registers: H * D <= 2048 (beyond that: re-read)

template <typename T>
__device__ __forceinline__ float q_dequant(int q, int half, float scale, float zero) {
  return ElemTraits<T>::rnd(__fadd_rn(ElemTraits<T>::rnd(__fmul_rn((float)(q - half), scale)), zero));
}

template <typename T>
__global__ __launch_bounds__(kQThreads) void kv_requant_kernel(T* work, uint8_t* q_out, T* scales, T* zeros, int H, int S,
                                                              int D, int n_bit) {
  __shared__ float sm_mn[kQThreads / 64], sm_mx[kQThreads / 64];
  const int s = blockIdx.x;
  const int n = H * D;
  const int max_int = (1 << n_bit) - 1, half = 1 << (n_bit - 1);
  float x[kQMaxPer];
  float mn = INFINITY, mx = -INFINITY;
#pragma unroll
  for (int j = 0; j < kQMaxPer; j++) {
    const int e = threadIdx.x + j * kQThreads;
    if (e < n) {
      const int h = e / D, d = e - h * D;
      x[j] = ElemTraits<T>::load(work, ((size_t)h * S + s) * D + d);
      mn = fminf(mn, x[j]);
      mx = fmaxf(mx, x[j]);
    }
  }
  for (int e = threadIdx.x + kQMaxPer * kQThreads; e < n; e += kQThreads) {  // very wide slots: second read below
    const int h = e / D, d = e - h * D;
    const float v = ElemTraits<T>::load(work, ((size_t)h * S + s) * D + d);
    mn = fminf(mn, v);
    mx = fmaxf(mx, v);
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    mn = fminf(mn, __shfl_xor(mn, off, CC_WAVE));
    mx = fmaxf(mx, __shfl_xor(mx, off, CC_WAVE));
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) {
    sm_mn[wave] = mn;
    sm_mx[wave] = mx;
  }
  __syncthreads();
  mn = sm_mn[0];
  mx = sm_mx[0];
#pragma unroll
  for (int w = 1; w < kQThreads / 64; w++) {
    mn = fminf(mn, sm_mn[w]);
    mx = fmaxf(mx, sm_mx[w]);
  }
  // ref: quantization_utils.py:10-14 — every op on tensors of the cache dtype
  float range = ElemTraits<T>::rnd(__fsub_rn(mx, mn));
  const float floor_t = ElemTraits<T>::rnd(1e-6f);
  range = range < floor_t ? floor_t : range;
  const float scale = ElemTraits<T>::rnd(__fdiv_rn(range, (float)max_int));
  const float zero = ElemTraits<T>::rnd(__fadd_rn(mn, ElemTraits<T>::rnd(__fmul_rn(scale, (float)half))));
  if (threadIdx.x == 0) {
    ElemTraits<T>::store(scales, (size_t)s, scale);
    ElemTraits<T>::store(zeros, (size_t)s, zero);
  }
  const int per = 8 / n_bit;
  auto one = [&](int e, float v) -> int {  // quantises element e, writes its round trip back, returns q
    const int h = e / D, d = e - h * D;
    const size_t i = ((size_t)h * S + s) * D + d;
    float t = ElemTraits<T>::rnd(__fdiv_rn(ElemTraits<T>::rnd(__fsub_rn(v, mn)), scale));  // :17-18
    t = rintf(t);                                                                          // :19 round half to even
    t = fminf(fmaxf(t, 0.f), (float)max_int);                                              // :20
    const int q = (int)t;
    ElemTraits<T>::store(work, i, q_dequant<T>(q, half, scale, zero));                     // :41-46
    return q;
  };
  // 8/n consecutive elements of the flattened tensor share a byte: the threads of a pack group are adjacent
  // lanes (D % per == 0 and kQThreads % per == 0), so the byte is assembled with DPP-free shuffles.
  auto emit = [&](int e, int q, bool valid) {  // called by ALL threads (shuffles); a pack group is valid as a whole
    if (n_bit == 8) {
      if (valid) {
        const int h = e / D, d = e - h * D;
        q_out[((size_t)h * S + s) * D + d] = (uint8_t)q;
      }
    } else {
      unsigned b = valid ? (unsigned)q << ((e % per) * n_bit) : 0u;
      for (int off = 1; off < per; off <<= 1) b |= __shfl_xor(b, off, CC_WAVE);
      if (valid && e % per == 0) {
        const int h = e / D, d = e - h * D;
        q_out[(((size_t)h * S + s) * D + d) / per] = (uint8_t)b;
      }
    }
  };
#pragma unroll
  for (int j = 0; j < kQMaxPer; j++) {
    const int e = threadIdx.x + j * kQThreads;
    const int q = e < n ? one(e, x[j]) : 0;
    emit(e, q, e < n);
  }
  for (int e0 = kQMaxPer * kQThreads; e0 < n; e0 += kQThreads) {
    const int e = e0 + threadIdx.x;
    int q = 0;
    if (e < n) {
      const int h = e / D, d = e - h * D;
      q = one(e, ElemTraits<T>::load(work, ((size_t)h * S + s) * D + d));
    }
    emit(e, q, e < n);
  }
}

template <typename T>
__global__ __launch_bounds__(kQThreads) void kv_dequant_kernel(const uint8_t* q, const T* scales, const T* zeros, T* out,
                                                              int H, int S, int D, int n_bit) {
  const size_t total = (size_t)H * S * D;
  const int half = 1 << (n_bit - 1), per = 8 / n_bit, msk = (1 << n_bit) - 1;
  for (size_t i = (size_t)blockIdx.x * kQThreads + threadIdx.x; i < total; i += (size_t)gridDim.x * kQThreads) {
    const int s = (int)((i / D) % S);
    const int v = n_bit == 8 ? q[i] : (q[i / per] >> ((int)(i % per) * n_bit)) & msk;
    ElemTraits<T>::store(out, i, q_dequant<T>(v, half, ElemTraits<T>::load(scales, (size_t)s), ElemTraits<T>::load(zeros, (size_t)s)));
  }
}

static bool quant_args_ok(int H, int S, int D, int dtype, int n_bit) {
  return H > 0 && S > 0 && D > 0 && cc_dt_ok(dtype) && (n_bit == 8 || n_bit == 4 || n_bit == 2) && D % (8 / n_bit) == 0;
}

}  // namespace

extern "C" {

int cc_kv_requant(void* work, void* q_out, void* scales, void* zeros, int32_t H, int32_t S, int32_t D, int32_t dtype,
                  int32_t n_bit, cc_stream_t stream) {
  CC_ENTRY();
  if (!work || !q_out || !scales || !zeros || !quant_args_ok(H, S, D, dtype, n_bit)) return CC_ERR_BAD_ARG;
  hipStream_t st = (hipStream_t)stream;
  uint8_t* q = reinterpret_cast<uint8_t*>(q_out);
  switch (dtype) {
    case CC_DT_F32:
      hipLaunchKernelGGL(kv_requant_kernel<float>, dim3(S), dim3(kQThreads), 0, st, (float*)work, q, (float*)scales,
                         (float*)zeros, H, S, D, n_bit);
      break;
    case CC_DT_BF16:
      hipLaunchKernelGGL(kv_requant_kernel<bf16_t>, dim3(S), dim3(kQThreads), 0, st, (bf16_t*)work, q, (bf16_t*)scales,
                         (bf16_t*)zeros, H, S, D, n_bit);
      break;
    default:
      hipLaunchKernelGGL(kv_requant_kernel<f16_t>, dim3(S), dim3(kQThreads), 0, st, (f16_t*)work, q, (f16_t*)scales,
                         (f16_t*)zeros, H, S, D, n_bit);
      break;
  }
  CC_LAUNCH_CHECK();
  return CC_OK;
}

int cc_kv_dequant(const void* q, const void* scales, const void* zeros, void* work_out, int32_t H, int32_t S, int32_t D,
                  int32_t dtype, int32_t n_bit, cc_stream_t stream) {
  CC_ENTRY();
  if (!q || !scales || !zeros || !work_out || !quant_args_ok(H, S, D, dtype, n_bit)) return CC_ERR_BAD_ARG;
  hipStream_t st = (hipStream_t)stream;
  const size_t total = (size_t)H * S * D;
  const int blocks = (int)((total + kQThreads * 8 - 1) / (kQThreads * 8));
  const uint8_t* qq = reinterpret_cast<const uint8_t*>(q);
  switch (dtype) {
    case CC_DT_F32:
      hipLaunchKernelGGL(kv_dequant_kernel<float>, dim3(blocks), dim3(kQThreads), 0, st, qq, (const float*)scales,
                         (const float*)zeros, (float*)work_out, H, S, D, n_bit);
      break;
    case CC_DT_BF16:
      hipLaunchKernelGGL(kv_dequant_kernel<bf16_t>, dim3(blocks), dim3(kQThreads), 0, st, qq, (const bf16_t*)scales,
                         (const bf16_t*)zeros, (bf16_t*)work_out, H, S, D, n_bit);
      break;
    default:
      hipLaunchKernelGGL(kv_dequant_kernel<f16_t>, dim3(blocks), dim3(kQThreads), 0, st, qq, (const f16_t*)scales,
                         (const f16_t*)zeros, (f16_t*)work_out, H, S, D, n_bit);
      break;
  }
  CC_LAUNCH_CHECK();
  return CC_OK;
}

}  // extern "C"
