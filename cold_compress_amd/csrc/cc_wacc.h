// cc_wacc.h — exact window sums of the attention-history ring (shared by cc_hybrid.hip and the decode combine pass).
#pragma once
#include "cc_common.h"

namespace {
typedef unsigned long long u64;

// dtype(sum of the W history entries of one cache slot) (cache.py:855-859 `.sum(dim=-1)` on a model-dtype tensor;
// torch's own fp32 summation order is unspecified and backend-specific).  Definition used here and in the oracle:
// the EXACT sum of the W entries, rounded once (nearest-even) to the model dtype.  Exactness makes the value
// independent of summation order, which is what lets the ring policies keep it INCREMENTALLY (sum += new - old when
// one ring column is overwritten) instead of re-reading the [H, S, W] ring — 118 MB per layer per step at S = 18432,
// W = 400 — on every decode step.
//
// WAcc: 192-bit two's-complement fixed point in units of 2^-149 (the fp32 subnormal quantum; every bf16 / f16 / fp32
// value is an integer multiple of it) + a count of entries that do not fit: |v| >= 4 or non-finite (attention
// probabilities are <= 1; the window sum of a row holding such an entry is NaN).  192 bits hold 2^40 entries < 4.
struct WAcc {
  u64 w0, w1, w2, special;
};

__device__ __forceinline__ void wacc_add_words(WAcc& a, u64 b0, u64 b1, u64 b2) {
  const u64 r0 = a.w0 + b0;
  const u64 c0 = r0 < b0;
  const u64 t = a.w1 + b1;
  u64 c1 = t < b1;
  const u64 r1 = t + c0;
  c1 |= (u64)(r1 < t);
  a.w0 = r0;
  a.w1 = r1;
  a.w2 = a.w2 + b2 + c1;
}
__device__ __forceinline__ void wacc_merge(WAcc& a, const WAcc& b) {
  wacc_add_words(a, b.w0, b.w1, b.w2);
  a.special += b.special;
}
// a += v (remove == false) or a -= v (remove == true), exactly
__device__ __forceinline__ void wacc_add_value(WAcc& a, float v, bool remove) {
  const uint32_t u = __float_as_uint(v);
  const uint32_t E = (u >> 23) & 0xffu, M = u & 0x7fffffu;
  if ((u << 1) == 0) return;
  if (E >= 129) {  // |v| >= 4, inf, nan
    a.special += remove ? ~0ull : 1ull;
    return;
  }
  const u64 m = E ? (u64)(M | 0x800000u) : (u64)M;  // v = m * 2^(sh - 149)
  const int sh = E ? (int)E - 1 : 0;
  const int word = sh >> 6, bit = sh & 63;
  const u64 lo = m << bit;
  const u64 hi = bit > 40 ? m >> (64 - bit) : 0ull;
  u64 b0 = word == 0 ? lo : 0ull, b1 = word == 0 ? hi : lo, b2 = word == 0 ? 0ull : hi;
  if (((u >> 31) != 0) != remove) {  // subtract: two's complement of the 192-bit magnitude
    b0 = ~b0 + 1ull;
    const u64 k0 = b0 == 0;
    b1 = ~b1 + k0;
    const u64 k1 = k0 && b1 == 0;
    b2 = ~b2 + k1;
  }
  wacc_add_words(a, b0, b1, b2);
}
// the exact value of the accumulator, rounded once to T (nearest-even), returned as a float
template <typename T>
__device__ __forceinline__ float wacc_round(const WAcc& a) {
  if (a.special != 0) return NAN;
  u64 m0 = a.w0, m1 = a.w1, m2 = a.w2;
  const bool neg = (long long)m2 < 0;
  if (neg) {
    m0 = ~m0 + 1ull;
    const u64 k0 = m0 == 0;
    m1 = ~m1 + k0;
    const u64 k1 = k0 && m1 == 0;
    m2 = ~m2 + k1;
  }
  if ((m0 | m1 | m2) == 0) return 0.f;
  const int P = m2 ? 191 - __clzll((long long)m2) : (m1 ? 127 - __clzll((long long)m1) : 63 - __clzll((long long)m0));
  uint32_t bits;
  if (P < 24) {
    bits = (uint32_t)m0;  // below 2^-125: the accumulator IS the fp32 encoding (subnormal or first binade), nothing to round
  } else {
    // 64-bit window whose top bit is bit P; sticky = any set bit below the window
    u64 hi, sticky;
    if (P < 63) {
      hi = m0 << (63 - P);
      sticky = 0;
    } else {
      const int lb = P - 63, wi = lb >> 6, b = lb & 63;
      const u64 x0 = wi == 0 ? m0 : (wi == 1 ? m1 : m2);
      const u64 x1 = wi == 0 ? m1 : (wi == 1 ? m2 : 0ull);
      hi = b ? (x0 >> b) | (x1 << (64 - b)) : x0;
      sticky = (b ? (x0 & ((1ull << b) - 1ull)) : 0ull) | (wi >= 1 ? m0 : 0ull) | (wi >= 2 ? m1 : 0ull);
    }
    u64 mant = hi >> 40;  // 24 bits, leading one included
    const u64 rem = hi & ((1ull << 40) - 1ull);
    int Pe = P;
    if (sizeof(T) == 4) {  // the target IS fp32: nearest-even here
      const u64 half = 1ull << 39;
      if (rem > half || (rem == half && (sticky != 0 || (mant & 1ull)))) mant += 1;
      if (mant == (1ull << 24)) {
        mant >>= 1;
        Pe += 1;
      }
    } else {  // 16-bit target: round to odd at 24 bits, then the dtype's own nearest-even is exact (>= 13 spare bits)
      mant |= (u64)((rem | sticky) != 0);
    }
    bits = ((uint32_t)(Pe - 22) << 23) | ((uint32_t)mant & 0x7fffffu);
  }
  const float f = __uint_as_float(bits | (neg ? 0x80000000u : 0u));
  return ElemTraits<T>::rnd(f);
}

}  // namespace
