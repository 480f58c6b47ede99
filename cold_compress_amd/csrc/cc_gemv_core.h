// cc_gemv_core.h — the arithmetic of the streamed matrix-vector products (cc_gemv.hip), shared with the single-launch layer step
// that folds the layer's QKV projection in (cc_attn_decode_qkv.hip): the SAME products, chains and reduction orders, so that the
// fused launch and the stand-alone GEMV give bit-identical q / k / v.
#pragma once
#include "cc_common.h"

namespace {

typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));

template <typename T>
struct Dot16;  // acc += <16 bytes of W, 16 bytes of x>
template <>
struct Dot16<bf16_t> {
  __device__ static __forceinline__ float run(uint4 w, uint4 x, float acc) {
    acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, w.x), __builtin_bit_cast(bf16x2_t, x.x), acc, false);
    acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, w.y), __builtin_bit_cast(bf16x2_t, x.y), acc, false);
    acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, w.z), __builtin_bit_cast(bf16x2_t, x.z), acc, false);
    acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, w.w), __builtin_bit_cast(bf16x2_t, x.w), acc, false);
    return acc;
  }
};
template <>
struct Dot16<f16_t> {
  __device__ static __forceinline__ float run(uint4 w, uint4 x, float acc) {
    acc = __builtin_amdgcn_fdot2(__builtin_bit_cast(f16x2_t, w.x), __builtin_bit_cast(f16x2_t, x.x), acc, false);
    acc = __builtin_amdgcn_fdot2(__builtin_bit_cast(f16x2_t, w.y), __builtin_bit_cast(f16x2_t, x.y), acc, false);
    acc = __builtin_amdgcn_fdot2(__builtin_bit_cast(f16x2_t, w.z), __builtin_bit_cast(f16x2_t, x.z), acc, false);
    acc = __builtin_amdgcn_fdot2(__builtin_bit_cast(f16x2_t, w.w), __builtin_bit_cast(f16x2_t, x.w), acc, false);
    return acc;
  }
};
template <>
struct Dot16<float> {
  __device__ static __forceinline__ float run(uint4 w, uint4 x, float acc) {
    acc = fmaf(__uint_as_float(w.x), __uint_as_float(x.x), acc);
    acc = fmaf(__uint_as_float(w.y), __uint_as_float(x.y), acc);
    acc = fmaf(__uint_as_float(w.z), __uint_as_float(x.z), acc);
    acc = fmaf(__uint_as_float(w.w), __uint_as_float(x.w), acc);
    return acc;
  }
};

template <int CTRL>
__device__ __forceinline__ float gv_dpp(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
// whole-wave sum with a wave-uniform result (4 DPP steps per 16-lane row, 4 v_readlane); fixed order
__device__ __forceinline__ float gv_wave_sum(float v) {
  v += gv_dpp<0xB1>(v);
  v += gv_dpp<0x4E>(v);
  v += gv_dpp<0x141>(v);
  v += gv_dpp<0x140>(v);
  const int u = __builtin_bit_cast(int, v);
  const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(u, 0));
  const float r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(u, 16));
  const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(u, 32));
  const float r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(u, 48));
  return (r0 + r1) + (r2 + r3);
}

typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint4 nt_load(const uint4* p) {  // streamed once per token: do not keep it in L2 / MALL
  const u32x4_t v = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(p));
  return make_uint4(v.x, v.y, v.z, v.w);
}

template <typename T>
__device__ __forceinline__ uint4 pack16(const float* f) {
  if constexpr (sizeof(T) == 4) {
    return make_uint4(__float_as_uint(f[0]), __float_as_uint(f[1]), __float_as_uint(f[2]), __float_as_uint(f[3]));
  } else {
    T e[8];
#pragma unroll
    for (int i = 0; i < 8; i++) ElemTraits<T>::store(&e[i], 0, f[i]);
    return make_uint4((uint32_t)e[0].x | ((uint32_t)e[1].x << 16), (uint32_t)e[2].x | ((uint32_t)e[3].x << 16),
                      (uint32_t)e[4].x | ((uint32_t)e[5].x << 16), (uint32_t)e[6].x | ((uint32_t)e[7].x << 16));
  }
}

}  // namespace
