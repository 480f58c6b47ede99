// cc_common.h — device helpers shared by the gfx950 kernels (wave64, CDNA4 only; no CUDA paths).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/coldcompress.h"
#include "../../include/coldcompress_debug.h"

#define CC_WAVE 64

// ------------------------------------------------------------------ element types
struct bf16_t {
  uint16_t x;
};
struct f16_t {
  uint16_t x;
};

__device__ __forceinline__ float bf16_bits_to_f32(uint32_t h) { return __uint_as_float(h << 16); }

// round-to-nearest-even via the gfx950 v_cvt_pk_bf16_f32 instruction (one op instead of ~7); identical to
// oracle/cc_oracle.c f32_to_bf16 for every non-NaN input (NaN stays a quiet NaN).
__device__ __forceinline__ uint16_t f32_to_bf16_bits(float f) {
  const __bf16 b = (__bf16)f;
  uint16_t h;
  __builtin_memcpy(&h, &b, 2);
  return h;
}

__device__ __forceinline__ float f16_bits_to_f32(uint16_t h) {
  _Float16 v;
  __builtin_memcpy(&v, &h, 2);
  return (float)v;
}
__device__ __forceinline__ uint16_t f32_to_f16_bits(float f) {
  _Float16 v = (_Float16)f;  // v_cvt_f16_f32: round-to-nearest-even
  uint16_t h;
  __builtin_memcpy(&h, &v, 2);
  return h;
}

// An fp32 value the optimiser must treat as opaque: keeps `T(fmaf(a, b, c))` a TWO-step rounding (fp32, then T).  Without it the
// f16 instantiations fuse the pair into v_fma_mixlo_f16 (ONE rounding of the exact result) wherever the pattern is visible, and
// two kernels computing the same formula disagree by an ulp in the rare double-rounding cases.
__device__ __forceinline__ float cc_opaque_f32(float x) {
  asm("" : "+v"(x));
  return x;
}

template <typename T>
struct ElemTraits;
template <>
struct ElemTraits<float> {
  static constexpr int code = CC_DT_F32;
  __device__ static __forceinline__ float load(const float* p, size_t i) { return p[i]; }
  __device__ static __forceinline__ void store(float* p, size_t i, float v) { p[i] = v; }
  __device__ static __forceinline__ float rnd(float v) { return v; }
};
template <>
struct ElemTraits<bf16_t> {
  static constexpr int code = CC_DT_BF16;
  __device__ static __forceinline__ float load(const bf16_t* p, size_t i) { return bf16_bits_to_f32(p[i].x); }
  __device__ static __forceinline__ void store(bf16_t* p, size_t i, float v) { p[i].x = f32_to_bf16_bits(v); }
  __device__ static __forceinline__ float rnd(float v) { return bf16_bits_to_f32(f32_to_bf16_bits(v)); }
};
// two roundings in one v_cvt_pk_bf16_f32 (the conversion is a packed instruction: rounding values one at a time wastes
// half of it); `packed` = lo | hi << 16 is what a 16-bit MFMA operand wants, the floats are its two halves
__device__ __forceinline__ uint32_t bf16_round_pair(float a, float b, float& ra, float& rb) {
  typedef float cc_f32x2 __attribute__((ext_vector_type(2)));
  typedef __bf16 cc_bf16x2 __attribute__((ext_vector_type(2)));
  const cc_f32x2 v = {a, b};
  const cc_bf16x2 r = __builtin_convertvector(v, cc_bf16x2);
  const uint32_t packed = __builtin_bit_cast(uint32_t, r);
  ra = __uint_as_float(packed << 16);
  rb = __uint_as_float(packed & 0xffff0000u);
  return packed;
}

template <>
struct ElemTraits<f16_t> {
  static constexpr int code = CC_DT_F16;
  __device__ static __forceinline__ float load(const f16_t* p, size_t i) { return f16_bits_to_f32(p[i].x); }
  __device__ static __forceinline__ void store(f16_t* p, size_t i, float v) { p[i].x = f32_to_f16_bits(v); }
  __device__ static __forceinline__ float rnd(float v) { return f16_bits_to_f32(f32_to_f16_bits(v)); }
};

// 16-byte vector of T, unpacked to floats
template <typename T>
struct Vec16 {
  static constexpr int N = 16 / sizeof(T);
  uint4 raw;
  __device__ __forceinline__ void load(const T* p) { raw = *reinterpret_cast<const uint4*>(p); }
  // streamed-once data (K/V rows of a decode step, weights): non-temporal, do not displace what L2 / MALL hold
  __device__ __forceinline__ void load_nt(const T* p) {
    typedef unsigned int u32x4_nt __attribute__((ext_vector_type(4)));
    const u32x4_nt v = __builtin_nontemporal_load(reinterpret_cast<const u32x4_nt*>(p));
    raw = make_uint4(v.x, v.y, v.z, v.w);
  }
  __device__ __forceinline__ void unpack(float* out) const;
};
template <>
__device__ __forceinline__ void Vec16<float>::unpack(float* o) const {
  o[0] = __uint_as_float(raw.x);
  o[1] = __uint_as_float(raw.y);
  o[2] = __uint_as_float(raw.z);
  o[3] = __uint_as_float(raw.w);
}
template <>
__device__ __forceinline__ void Vec16<bf16_t>::unpack(float* o) const {
  const uint32_t w[4] = {raw.x, raw.y, raw.z, raw.w};
#pragma unroll
  for (int i = 0; i < 4; i++) {
    o[2 * i] = __uint_as_float(w[i] << 16);
    o[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u);
  }
}
template <>
__device__ __forceinline__ void Vec16<f16_t>::unpack(float* o) const {
  const uint32_t w[4] = {raw.x, raw.y, raw.z, raw.w};
#pragma unroll
  for (int i = 0; i < 4; i++) {
    o[2 * i] = f16_bits_to_f32((uint16_t)(w[i] & 0xffffu));
    o[2 * i + 1] = f16_bits_to_f32((uint16_t)(w[i] >> 16));
  }
}

// ------------------------------------------------------------------ torch.argmin ordering as a 64-bit key
// key = (orderable(value) << 32) | slot ; min over keys == first index of the minimum.
// NaN maps to 0 (torch.argmin returns the first NaN); -0.0 is canonicalised to +0.0 (they compare equal).
__device__ __forceinline__ uint32_t orderable_f32(float f) {
  if (f != f) return 0u;
  if (f == 0.0f) f = 0.0f;
  uint32_t u = __float_as_uint(f);
  u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
  return u == 0u ? 1u : u;  // keep 0 reserved for NaN (only ~0x ffffffff = -NaN pattern could hit it)
}
__device__ __forceinline__ uint32_t orderable_i32(int32_t v) { return (uint32_t)v ^ 0x80000000u; }

__device__ __forceinline__ unsigned long long make_key(uint32_t ord, uint32_t slot) {
  return ((unsigned long long)ord << 32) | slot;
}

// Uniform draw of KVCacheRandom's in-kernel generator (include/coldcompress.h, cc_decode_step_random_rng): stateless, 24 bits.
__host__ __device__ static inline uint64_t cc_mix64(uint64_t x) {
  x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33;
  return x;
}
__host__ __device__ static inline float cc_rng_uniform(uint64_t seed, int32_t pos, int32_t slot) {
  const uint64_t x = cc_mix64(cc_mix64(seed + (uint64_t)(uint32_t)pos * 0x9E3779B97F4A7C15ull + (uint64_t)(uint32_t)slot));
  return (float)(uint32_t)(x >> 40) * 5.9604644775390625e-08f;  // * 2^-24: exact
}

// Fused heavy-hitter decode step: the arg-min key of the next eviction is published as one partial minimum
// per 128-slot chunk of the cache (no atomics: same-address atomics from 8 XCDs serialise in the fabric and
// cost more than the whole pass); consumers take the minimum over a head's cc_next_key_slots(S) entries.
// A head's key row holds kNextKeyPerChunk entries per 128-slot chunk: the two-launch combine pass fills the first n_chunks (one
// key per block) and keeps the others at ~0; the single-launch step publishes one key per WAVE of its 64-slot workgroups
// (2 workgroups x 4 waves per chunk), so that its waves never wait for each other at the end of the launch.
constexpr int kNextKeyChunk = 128;
constexpr int kNextKeyPerChunk = 8;
// r6: behind the live entries every row carries kNextKeyTail more that no arg-min ever reads — per-head state a policy's step hands
// to the NEXT step of the same cache.  l2: entries [live + 0], [live + 1] = the head's norm record of an even / odd position (below).
constexpr int kNextKeyTail = 8;
static inline int cc_next_key_live(int S) { return kNextKeyPerChunk * ((S + kNextKeyChunk - 1) / kNextKeyChunk); }
static inline int cc_next_key_slots(int S) { return cc_next_key_live(S) + kNextKeyTail; }  // the row stride
// The l2 policy's carried norm record of kv head h, written by the step of position p into row entry [live + (p & 1)].  Bits 0-15:
// the head's largest norm over the slots it KEEPS at position p + 1 — every slot AFTER p's insert but the one p + 1 evicts, which is
// the arg-min of the keys the step of p scores — as a model-dtype bit pattern (norms are >= +0 or NaN: unsigned order == numeric
// order, NaN above everything, which is torch.max's propagation).  With the norm of p + 1's new key that is the head's term of
// cache.py:602's maximum at p + 1: no reduction over the head and no exchange between heads inside p + 1's launch.  Bits 16-31 / 32-63:
// the head's largest norm over ALL slots and a slot that holds it (not consumed: for inspection and the tests).
int cc_l2_record_launch(const void* key_norm, int H, int S, int dtype, const int32_t* input_pos, int delta, unsigned long long* next_key,
                        hipStream_t st);  // cc_evict.hip: the record of position *input_pos + delta from the norms as they stand
__host__ __device__ static inline unsigned long long cc_l2_record(unsigned kept_max, unsigned all_max, unsigned all_max_slot) {
  return (unsigned long long)(kept_max & 0xffffu) | ((unsigned long long)(all_max & 0xffffu) << 16) | ((unsigned long long)all_max_slot << 32);
}

__device__ __forceinline__ unsigned long long wave_min_u64(unsigned long long v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    unsigned long long o = __shfl_xor(v, off, CC_WAVE);
    v = o < v ? o : v;
  }
  return v;
}

__device__ __forceinline__ float wave_max_f32(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, CC_WAVE));
  return v;
}

// Block-wide min of 64-bit keys; result valid in every thread.  `sm` needs blockDim.x/64 + 1 slots.
__device__ __forceinline__ unsigned long long block_min_u64(unsigned long long v, unsigned long long* sm) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  v = wave_min_u64(v);
  if (lane == 0) sm[wave] = v;
  __syncthreads();
  if (wave == 0) {
    unsigned long long x = lane < nw ? sm[lane] : ~0ull;
    x = wave_min_u64(x);
    if (lane == 0) sm[nw] = x;
  }
  __syncthreads();
  return sm[nw];
}

// Correctly rounded fp32 square root (what torch.linalg.vector_norm's sqrt is on any IEEE backend): through fp64 —
// 53 >= 2 * 24 + 2 bits make the double rounding exact.  (__fsqrt_rn compiled to a 1-ulp v_sqrt_f32 here: fp32 key
// norms differed from the oracle in the last bit.)
__device__ __forceinline__ float cc_sqrt_rn(float x) { return (float)__dsqrt_rn((double)x); }

// Canonical sum of squares of a D-vector, evaluated by 16 cooperating lanes (lane16 = 0..15 of an aligned
// 16-lane group): a_j = sum_i x[j+16i]^2 (sequential in i; separate multiply and add, no fma), then the
// butterfly a_j += a_{j^8}, ^4, ^2, ^1.  oracle/cc_oracle.c sumsq_canonical() evaluates the same order,
// so row norms are bit-identical between device and oracle (the reference's own order is unspecified).
template <typename T>
__device__ __forceinline__ float sumsq_canonical_16(const T* x, int D, int lane16) {
  float a = 0.f;
  for (int d = lane16; d < D; d += 16) {
    float e = ElemTraits<T>::load(x, d);
    float sq = __fmul_rn(e, e);
    a = __fadd_rn(a, sq);
  }
#pragma unroll
  for (int off = 8; off > 0; off >>= 1) a = __fadd_rn(a, __shfl_xor(a, off, 16));
  return a;
}

// hipGetLastError() is per-thread state shared with every other HIP user in the process (PyTorch leaves
// benign codes such as hipErrorNotReady behind): CC_ENTRY() clears it on entry, CC_LAUNCH_CHECK() reads ours.
#define CC_ENTRY() (void)hipGetLastError()
#define CC_LAUNCH_CHECK()                               \
  do {                                                  \
    if (hipGetLastError() != hipSuccess) return CC_ERR_HIP; \
  } while (0)

static inline int cc_dt_ok(int dt) { return dt == CC_DT_F32 || dt == CC_DT_BF16 || dt == CC_DT_F16; }
static inline size_t cc_dt_size(int dt) { return dt == CC_DT_F32 ? 4 : 2; }
static inline int cc_view_ok(const cc_kv_view* c) {
  return c && c->k_cache && c->v_cache && c->pos && c->mask && c->cache_cts && c->H > 0 && c->S > 0 && c->D > 0 &&
         (c->Hp == 1 || c->Hp == c->H) && (c->Hc == 1 || c->Hc == c->H) && cc_dt_ok(c->dtype) &&
         ((size_t)c->D * cc_dt_size(c->dtype)) % 4 == 0;
}
