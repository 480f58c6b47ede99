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

constexpr int kQSlots = 4;   // slots per workgroup of the vectorised round trip

struct RequantSet {
  void* work[2];
  uint8_t* q[2];
  void* scales[2];
  void* zeros[2];
  // Exact skipping of slots the round trip no longer changes (vectorised kernel only; null = requantise everything).
  // The round trip of a slot is a pure function of that slot's rows: once a pass left a slot's rows bit-identical
  // (stable = 1) and nothing has been inserted there since (pos of every head still equals pos_seen), running it
  // again would reproduce the same rows, image, scale and zero point — so it is not run.
  const int32_t* pos;   // [Hp, S]
  int Hp;
  uint8_t* stable;      // [2, S]   (K, V)
  int32_t* pos_seen;    // [2, Hp, S]
};

template <typename T>
__global__ __launch_bounds__(kQThreads) void kv_requant_kernel(RequantSet rs, int H, int S, int D, int n_bit) {
  T* work = reinterpret_cast<T*>(rs.work[blockIdx.y]);  // blockIdx.y: 0 = K, 1 = V (one launch for both)
  uint8_t* q_out = rs.q[blockIdx.y];
  T* scales = reinterpret_cast<T*>(rs.scales[blockIdx.y]);
  T* zeros = reinterpret_cast<T*>(rs.zeros[blockIdx.y]);
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

// Vectorised round trip: one thread = one 16-byte vector of the slot (H * D / VEC threads per slot, up to 1024), so a
// slot costs one 16-byte load, one 16-byte store of the round trip and one packed store of the image per thread
// (the element-wise kernel above issues 2-byte loads: ~1 TB/s).  Same arithmetic, element for element.
// requant_vec_slot: the whole workgroup takes slot s of one tensor (K or V); returns whether THIS thread's vector changed.
template <typename T>
__device__ __forceinline__ bool requant_vec_slot(T* work, uint8_t* q_out, T* scales, T* zeros, int H, int S, int D, int n_bit, int s,
                                                 float* sm_mn, float* sm_mx) {
  constexpr int VEC = 16 / (int)sizeof(T);
  const int vpr = D / VEC;           // vectors per row
  const int nvec = H * vpr;          // == blockDim.x rounded up to a wave
  const int v = threadIdx.x;
  const bool in = v < nvec;
  const int h = in ? v / vpr : 0, dv = in ? v - (v / vpr) * vpr : 0;
  const size_t base = ((size_t)h * S + s) * D + (size_t)dv * VEC;
  const int max_int = (1 << n_bit) - 1, half = 1 << (n_bit - 1);
  float x[VEC];
  float mn = INFINITY, mx = -INFINITY;
  if (in) {
    Vec16<T> xv;
    xv.load(work + base);
    xv.unpack(x);
#pragma unroll
    for (int e = 0; e < VEC; e++) {
      mn = fminf(mn, x[e]);
      mx = fmaxf(mx, x[e]);
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    mn = fminf(mn, __shfl_xor(mn, off, CC_WAVE));
    mx = fmaxf(mx, __shfl_xor(mx, off, CC_WAVE));
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  if (lane == 0) {
    sm_mn[wave] = mn;
    sm_mx[wave] = mx;
  }
  __syncthreads();
  mn = sm_mn[0];
  mx = sm_mx[0];
  for (int w = 1; w < nw; w++) {
    mn = fminf(mn, sm_mn[w]);
    mx = fmaxf(mx, sm_mx[w]);
  }
  float range = ElemTraits<T>::rnd(__fsub_rn(mx, mn));
  const float floor_t = ElemTraits<T>::rnd(1e-6f);
  range = range < floor_t ? floor_t : range;
  const float scale = ElemTraits<T>::rnd(__fdiv_rn(range, (float)max_int));
  const float zero = ElemTraits<T>::rnd(__fadd_rn(mn, ElemTraits<T>::rnd(__fmul_rn(scale, (float)half))));
  if (threadIdx.x == 0) {
    ElemTraits<T>::store(scales, (size_t)s, scale);
    ElemTraits<T>::store(zeros, (size_t)s, zero);
  }
  bool changed = false;
  if (in) {
    float o[VEC];
    unsigned long long packed = 0ull;  // VEC values of n_bit bits, value j shifted left by j * n_bit
#pragma unroll
    for (int e = 0; e < VEC; e++) {
      float t = ElemTraits<T>::rnd(__fdiv_rn(ElemTraits<T>::rnd(__fsub_rn(x[e], mn)), scale));
      t = fminf(fmaxf(rintf(t), 0.f), (float)max_int);
      const int q = (int)t;
      o[e] = q_dequant<T>(q, half, scale, zero);
      packed |= (unsigned long long)q << (e * n_bit);
    }
    if constexpr (sizeof(T) == 4) {
#pragma unroll
      for (int e = 0; e < VEC; e++) changed |= __float_as_uint(o[e]) != __float_as_uint(x[e]);
      *reinterpret_cast<float4*>(work + base) = make_float4(o[0], o[1], o[2], o[3]);
    } else {
      T e16[VEC];
#pragma unroll
      for (int e = 0; e < VEC; e++) ElemTraits<T>::store(&e16[e], 0, o[e]);
      uint32_t w32[4];
#pragma unroll
      for (int i = 0; i < 4; i++) w32[i] = (uint32_t)e16[2 * i].x | ((uint32_t)e16[2 * i + 1].x << 16);
      T x16[VEC];
#pragma unroll
      for (int e = 0; e < VEC; e++) ElemTraits<T>::store(&x16[e], 0, x[e]);  // exact: x came from T
#pragma unroll
      for (int i = 0; i < 4; i++) changed |= w32[i] != ((uint32_t)x16[2 * i].x | ((uint32_t)x16[2 * i + 1].x << 16));
      *reinterpret_cast<uint4*>(work + base) = make_uint4(w32[0], w32[1], w32[2], w32[3]);
    }
    const size_t qbyte = base * n_bit / 8;  // VEC * n_bit is a whole number of bytes (VEC >= 4)
    const int nbytes = VEC * n_bit / 8;
    if (nbytes == 8) *reinterpret_cast<uint2*>(q_out + qbyte) = make_uint2((uint32_t)packed, (uint32_t)(packed >> 32));
    else if (nbytes == 4) *reinterpret_cast<uint32_t*>(q_out + qbyte) = (uint32_t)packed;
    else if (nbytes == 2) *reinterpret_cast<uint16_t*>(q_out + qbyte) = (uint16_t)packed;
    else q_out[qbyte] = (uint8_t)packed;
  }
  return changed;
}

template <typename T>
__global__ __launch_bounds__(1024) void kv_requant_vec_kernel(RequantSet rs, int H, int S, int D, int n_bit) {
  __shared__ float sm_mn[16], sm_mx[16];
  T* work = reinterpret_cast<T*>(rs.work[blockIdx.y]);
  uint8_t* q_out = rs.q[blockIdx.y];
  T* scales = reinterpret_cast<T*>(rs.scales[blockIdx.y]);
  T* zeros = reinterpret_cast<T*>(rs.zeros[blockIdx.y]);
  // kQSlots consecutive slots per workgroup: lane t of the first wave decides whether slot s0 + t needs the round trip
  // (in steady state ~0.3 % do: the freshly inserted slot and the few whose image oscillates), then the whole
  // workgroup walks the ones that do
  __shared__ unsigned long long sm_need;
  __shared__ int sm_changed;
  const int s0 = blockIdx.x * kQSlots;
  if (threadIdx.x < 64) {
    const int sc = s0 + (int)threadIdx.x;
    bool need = threadIdx.x < kQSlots && sc < S;
    if (need && rs.stable != nullptr) {
      bool same = rs.stable[(size_t)blockIdx.y * S + sc] != 0;
      for (int hp = 0; hp < rs.Hp; hp++)
        same &= rs.pos[(size_t)hp * S + sc] == rs.pos_seen[((size_t)blockIdx.y * rs.Hp + hp) * S + sc];
      need = !same;
    }
    const unsigned long long m = __ballot(need);
    if (threadIdx.x == 0) sm_need = m;
  }
  __syncthreads();
  unsigned long long todo = sm_need;
  while (todo) {
    const int s = s0 + __builtin_ctzll(todo);
    todo &= todo - 1;
    const bool changed = requant_vec_slot<T>(work, q_out, scales, zeros, H, S, D, n_bit, s, sm_mn, sm_mx);
    if (rs.stable != nullptr) {  // did this pass leave the slot's rows bit-identical?
      if (threadIdx.x == 0) sm_changed = 0;
      __syncthreads();
      if (changed) sm_changed = 1;
      __syncthreads();
      if (threadIdx.x == 0) rs.stable[(size_t)blockIdx.y * S + s] = sm_changed ? 0 : 1;
      if ((int)threadIdx.x < rs.Hp)
        rs.pos_seen[((size_t)blockIdx.y * rs.Hp + threadIdx.x) * S + s] = rs.pos[(size_t)threadIdx.x * S + s];
    }
    __syncthreads();  // sm_mn / sm_mx / sm_changed are reused by the next slot
  }
}

// The same round trip for SEVERAL caches in one launch (every layer of a model at the end of a token: cc_kv_requant_batch).
// One workgroup = 64 consecutive slots of one cache, K then V; lane t of the first wave decides for slot s0 + t (the
// position rows are read once for both tensors), then the workgroup walks the slots that need it.  `table`: kReqWords
// 64-bit words per cache — k_work, k_q, k_scales, k_zeros, v_work, v_q, v_scales, v_zeros, pos, stable, pos_seen, S, Hp.
constexpr int kReqWords = 16;
constexpr int kReqScan = 64;

template <typename T>
__global__ __launch_bounds__(1024) void kv_requant_batch_kernel(const long long* __restrict__ table, int H, int D, int n_bit) {
  __shared__ float sm_mn[16], sm_mx[16];
  __shared__ unsigned long long sm_need[2];
  __shared__ int sm_changed;
  const long long* e = table + (size_t)blockIdx.y * kReqWords;
  const int S = (int)e[11], Hp = (int)e[12];
  const int s0 = blockIdx.x * kReqScan;
  if (s0 >= S) return;
  const int32_t* pos = reinterpret_cast<const int32_t*>(e[8]);
  uint8_t* stable = reinterpret_cast<uint8_t*>(e[9]);
  int32_t* pos_seen = reinterpret_cast<int32_t*>(e[10]);
  if (threadIdx.x < 64) {
    const int sc = s0 + (int)threadIdx.x;
    bool need_k = sc < S, need_v = need_k;
    if (need_k) {
      bool same_k = stable[sc] != 0, same_v = stable[(size_t)S + sc] != 0;
      for (int hp = 0; hp < Hp; hp++) {
        const int32_t p = pos[(size_t)hp * S + sc];
        same_k &= p == pos_seen[(size_t)hp * S + sc];
        same_v &= p == pos_seen[((size_t)Hp + hp) * S + sc];
      }
      need_k = !same_k;
      need_v = !same_v;
    }
    const unsigned long long mk = __ballot(need_k), mv = __ballot(need_v);
    if (threadIdx.x == 0) {
      sm_need[0] = mk;
      sm_need[1] = mv;
    }
  }
  __syncthreads();
  for (int which = 0; which < 2; which++) {
    unsigned long long todo = sm_need[which];
    if (!todo) continue;
    T* work = reinterpret_cast<T*>(e[4 * which + 0]);
    uint8_t* q_out = reinterpret_cast<uint8_t*>(e[4 * which + 1]);
    T* scales = reinterpret_cast<T*>(e[4 * which + 2]);
    T* zeros = reinterpret_cast<T*>(e[4 * which + 3]);
    while (todo) {
      const int s = s0 + __builtin_ctzll(todo);
      todo &= todo - 1;
      const bool changed = requant_vec_slot<T>(work, q_out, scales, zeros, H, S, D, n_bit, s, sm_mn, sm_mx);
      if (threadIdx.x == 0) sm_changed = 0;
      __syncthreads();
      if (changed) sm_changed = 1;
      __syncthreads();
      if (threadIdx.x == 0) stable[(size_t)which * S + s] = sm_changed ? 0 : 1;
      if ((int)threadIdx.x < Hp) pos_seen[((size_t)which * Hp + threadIdx.x) * S + s] = pos[(size_t)threadIdx.x * S + s];
      __syncthreads();
    }
  }
}

static int requant_launch(const RequantSet& rs, int n, int32_t H, int32_t S, int32_t D, int32_t dtype, int32_t n_bit, hipStream_t st) {
  const int vec = 16 / (int)cc_dt_size(dtype);
  if (D % vec == 0 && H * (D / vec) <= 1024) {  // the common case: one thread per 16-byte vector of the slot
    const int threads = ((H * (D / vec) + 63) / 64) * 64;
    dim3 grid((S + kQSlots - 1) / kQSlots, n), block(threads);
    switch (dtype) {
      case CC_DT_F32: hipLaunchKernelGGL(kv_requant_vec_kernel<float>, grid, block, 0, st, rs, H, S, D, n_bit); break;
      case CC_DT_BF16: hipLaunchKernelGGL(kv_requant_vec_kernel<bf16_t>, grid, block, 0, st, rs, H, S, D, n_bit); break;
      default: hipLaunchKernelGGL(kv_requant_vec_kernel<f16_t>, grid, block, 0, st, rs, H, S, D, n_bit); break;
    }
    CC_LAUNCH_CHECK();
    return CC_OK;
  }
  dim3 grid(S, n), block(kQThreads);
  switch (dtype) {
    case CC_DT_F32: hipLaunchKernelGGL(kv_requant_kernel<float>, grid, block, 0, st, rs, H, S, D, n_bit); break;
    case CC_DT_BF16: hipLaunchKernelGGL(kv_requant_kernel<bf16_t>, grid, block, 0, st, rs, H, S, D, n_bit); break;
    default: hipLaunchKernelGGL(kv_requant_kernel<f16_t>, grid, block, 0, st, rs, H, S, D, n_bit); break;
  }
  CC_LAUNCH_CHECK();
  return CC_OK;
}


// ---------------------------------------------------------------------------------------------------------------
// The FUSED quantised cache (opt-in; NOT the reference's contract — include/coldcompress.h): one
// (scale, minimum) pair per (head, slot) ROW of K and of V, fp32, fixed when the row is written; the decode kernels read
// the uint8 images and dequantise in registers.  These two kernels convert whole caches (prefill, debugging): one wave
// per row, two passes over the row (the second one hits L1).
template <typename T>
__global__ __launch_bounds__(256) void kv_quant_rows_kernel(const T* k, const T* v, uint8_t* kq, uint8_t* vq, float* qparams,
                                                           size_t rows, int D) {
  const int lane = threadIdx.x & 63;
  const size_t wid = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (wid >= 2 * rows) return;
  const int which = wid >= rows ? 1 : 0;
  const size_t row = which ? wid - rows : wid;
  const T* src = (which ? v : k) + row * D;
  uint8_t* dst = (which ? vq : kq) + row * D;
  float mn = INFINITY, mx = -INFINITY;
  for (int e = lane; e < D; e += 64) {
    const float x = ElemTraits<T>::load(src, e);
    mn = fminf(mn, x);
    mx = fmaxf(mx, x);
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    mn = fminf(mn, __shfl_xor(mn, off, 64));
    mx = fmaxf(mx, __shfl_xor(mx, off, 64));
  }
  const float range = fmaxf(__fsub_rn(mx, mn), 1e-6f);
  const float sc = __fdiv_rn(range, 255.f), inv = __fdiv_rn(255.f, range);
  for (int e = lane; e < D; e += 64) {
    const float x = ElemTraits<T>::load(src, e);
    dst[e] = (uint8_t)fminf(fmaxf(rintf(__fmul_rn(__fsub_rn(x, mn), inv)), 0.f), 255.f);
  }
  if (lane == 0) {
    qparams[row * 4 + 2 * which] = sc;
    qparams[row * 4 + 2 * which + 1] = mn;
  }
}

template <typename T>
__global__ __launch_bounds__(256) void kv_dequant_rows_kernel(const uint8_t* kq, const uint8_t* vq, const float* qparams, T* k, T* v,
                                                             size_t rows, int D) {
  const size_t total = 2 * rows * (size_t)D;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int which = i >= rows * D ? 1 : 0;
    const size_t j = which ? i - rows * D : i;
    const size_t row = j / D;
    const float sc = qparams[row * 4 + 2 * which], mn = qparams[row * 4 + 2 * which + 1];
    const float x = cc_opaque_f32(__builtin_fmaf((float)(which ? vq : kq)[j], sc, mn));  // fp32 first, then T
    ElemTraits<T>::store(which ? v : k, j, x);
  }
}

}  // namespace

extern "C" {

int cc_kv_quant_rows(const void* k, const void* v, int32_t H, int32_t S, int32_t D, int32_t dtype, int32_t n_bit, uint8_t* k_q,
                     uint8_t* v_q, float* qparams, cc_stream_t stream) {
  CC_ENTRY();
  if (!k || !v || !k_q || !v_q || !qparams || H <= 0 || S <= 0 || D <= 0 || !cc_dt_ok(dtype)) return CC_ERR_BAD_ARG;
  if (n_bit != 8) return CC_ERR_UNSUPPORTED;
  const size_t rows = (size_t)H * S;
  const unsigned blocks = (unsigned)((2 * rows + 3) / 4);
  hipStream_t st = (hipStream_t)stream;
  switch (dtype) {
    case CC_DT_F32: hipLaunchKernelGGL(kv_quant_rows_kernel<float>, dim3(blocks), dim3(256), 0, st, (const float*)k, (const float*)v, k_q, v_q, qparams, rows, D); break;
    case CC_DT_BF16: hipLaunchKernelGGL(kv_quant_rows_kernel<bf16_t>, dim3(blocks), dim3(256), 0, st, (const bf16_t*)k, (const bf16_t*)v, k_q, v_q, qparams, rows, D); break;
    default: hipLaunchKernelGGL(kv_quant_rows_kernel<f16_t>, dim3(blocks), dim3(256), 0, st, (const f16_t*)k, (const f16_t*)v, k_q, v_q, qparams, rows, D); break;
  }
  CC_LAUNCH_CHECK();
  return CC_OK;
}

int cc_kv_dequant_rows(const uint8_t* k_q, const uint8_t* v_q, const float* qparams, int32_t H, int32_t S, int32_t D, int32_t dtype,
                       int32_t n_bit, void* k_out, void* v_out, cc_stream_t stream) {
  CC_ENTRY();
  if (!k_q || !v_q || !qparams || !k_out || !v_out || H <= 0 || S <= 0 || D <= 0 || !cc_dt_ok(dtype)) return CC_ERR_BAD_ARG;
  if (n_bit != 8) return CC_ERR_UNSUPPORTED;
  const size_t rows = (size_t)H * S;
  size_t nb = (2 * rows * D + 255) / 256;
  if (nb > 8192) nb = 8192;
  hipStream_t st = (hipStream_t)stream;
  switch (dtype) {
    case CC_DT_F32: hipLaunchKernelGGL(kv_dequant_rows_kernel<float>, dim3((unsigned)nb), dim3(256), 0, st, k_q, v_q, qparams, (float*)k_out, (float*)v_out, rows, D); break;
    case CC_DT_BF16: hipLaunchKernelGGL(kv_dequant_rows_kernel<bf16_t>, dim3((unsigned)nb), dim3(256), 0, st, k_q, v_q, qparams, (bf16_t*)k_out, (bf16_t*)v_out, rows, D); break;
    default: hipLaunchKernelGGL(kv_dequant_rows_kernel<f16_t>, dim3((unsigned)nb), dim3(256), 0, st, k_q, v_q, qparams, (f16_t*)k_out, (f16_t*)v_out, rows, D); break;
  }
  CC_LAUNCH_CHECK();
  return CC_OK;
}

int cc_kv_requant(void* work, void* q_out, void* scales, void* zeros, int32_t H, int32_t S, int32_t D, int32_t dtype,
                  int32_t n_bit, cc_stream_t stream) {
  CC_ENTRY();
  if (!work || !q_out || !scales || !zeros || !quant_args_ok(H, S, D, dtype, n_bit)) return CC_ERR_BAD_ARG;
  RequantSet rs{{work, nullptr}, {reinterpret_cast<uint8_t*>(q_out), nullptr}, {scales, nullptr}, {zeros, nullptr}, nullptr, 0, nullptr, nullptr};
  return requant_launch(rs, 1, H, S, D, dtype, n_bit, (hipStream_t)stream);
}

int cc_kv_requant_pair(void* k_work, void* k_q, void* k_scales, void* k_zeros, void* v_work, void* v_q, void* v_scales,
                       void* v_zeros, int32_t H, int32_t S, int32_t D, int32_t dtype, int32_t n_bit, const int32_t* pos,
                       int32_t Hp, uint8_t* stable, int32_t* pos_seen, cc_stream_t stream) {
  CC_ENTRY();
  if (!k_work || !k_q || !k_scales || !k_zeros || !v_work || !v_q || !v_scales || !v_zeros || !quant_args_ok(H, S, D, dtype, n_bit))
    return CC_ERR_BAD_ARG;
  if (stable && (!pos || !pos_seen || Hp <= 0)) return CC_ERR_BAD_ARG;
  if (Hp > 64) stable = nullptr;  // the stable-slot bookkeeping covers up to 64 position rows: beyond that every slot is redone
  RequantSet rs{{k_work, v_work}, {reinterpret_cast<uint8_t*>(k_q), reinterpret_cast<uint8_t*>(v_q)}, {k_scales, v_scales},
                {k_zeros, v_zeros}, pos, Hp, stable, pos_seen};
  return requant_launch(rs, 2, H, S, D, dtype, n_bit, (hipStream_t)stream);
}

int cc_kv_requant_batch(const int64_t* table, int32_t n_caches, int32_t H, int32_t S_max, int32_t D, int32_t dtype, int32_t n_bit,
                        cc_stream_t stream) {
  CC_ENTRY();
  if (!table || n_caches <= 0 || n_caches > 65535 || !quant_args_ok(H, S_max, D, dtype, n_bit)) return CC_ERR_BAD_ARG;
  const int vec = 16 / (int)cc_dt_size(dtype);
  if (D % vec != 0 || H * (D / vec) > 1024) return CC_ERR_UNSUPPORTED;  // (the single-cache call has an element-wise form)
  const int threads = ((H * (D / vec) + 63) / 64) * 64;
  dim3 grid((S_max + kReqScan - 1) / kReqScan, n_caches), block(threads);
  hipStream_t st = (hipStream_t)stream;
  const long long* tb = reinterpret_cast<const long long*>(table);
  switch (dtype) {
    case CC_DT_F32: hipLaunchKernelGGL(kv_requant_batch_kernel<float>, grid, block, 0, st, tb, H, D, n_bit); break;
    case CC_DT_BF16: hipLaunchKernelGGL(kv_requant_batch_kernel<bf16_t>, grid, block, 0, st, tb, H, D, n_bit); break;
    default: hipLaunchKernelGGL(kv_requant_batch_kernel<f16_t>, grid, block, 0, st, tb, H, D, n_bit); break;
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
