// cc_glue.hip — the caller glue around the hot path, fused: residual add + RMSNorm, QKV split + RoPE + head
// layout, SwiGLU gate.  The reference runs these as ~45 eager elementwise launches per layer
// (model.py:317-327, 375-387, 442-443, 452-457, 507-519); under hipGraph replay they cost more than the GEMVs.
// Pure HBM-bound elementwise/row-reduction work: 16-byte vector loads, wave64 shuffles, one LDS hop.
// Rounding points follow the reference's eager bf16/fp16 semantics (fp32 math inside, result rounded to the
// model dtype after each tensor op).
#include "cc_common.h"

namespace {

template <typename T>
__device__ __forceinline__ void store_vec(T* p, const float* f);
template <>
__device__ __forceinline__ void store_vec<float>(float* p, const float* f) {
  *reinterpret_cast<float4*>(p) = make_float4(f[0], f[1], f[2], f[3]);
}
template <>
__device__ __forceinline__ void store_vec<bf16_t>(bf16_t* p, const float* f) {
  uint32_t w[4];
#pragma unroll
  for (int i = 0; i < 4; i++) w[i] = (uint32_t)f32_to_bf16_bits(f[2 * i]) | ((uint32_t)f32_to_bf16_bits(f[2 * i + 1]) << 16);
  *reinterpret_cast<uint4*>(p) = make_uint4(w[0], w[1], w[2], w[3]);
}
template <>
__device__ __forceinline__ void store_vec<f16_t>(f16_t* p, const float* f) {
  uint32_t w[4];
#pragma unroll
  for (int i = 0; i < 4; i++) w[i] = (uint32_t)f32_to_f16_bits(f[2 * i]) | ((uint32_t)f32_to_f16_bits(f[2 * i + 1]) << 16);
  *reinterpret_cast<uint4*>(p) = make_uint4(w[0], w[1], w[2], w[3]);
}

constexpr int kNormThreads = 256;
constexpr int kNormMaxVec = 4;  // 16-byte vectors per thread held in registers: dim <= 256*4*VEC

// one workgroup per row; the row stays in registers between the reduction and the scaling pass
template <typename T>
__global__ __launch_bounds__(kNormThreads) void add_rmsnorm_kernel(const T* x, const T* delta, const T* weight, int dim,
                                                                  float eps, T* h_out, T* out) {
  constexpr int VEC = 16 / (int)sizeof(T);
  __shared__ float sm[kNormThreads / 64];
  const size_t row = (size_t)blockIdx.x * dim;
  const int nvec = dim / VEC;
  float h[kNormMaxVec][VEC];
  float ss = 0.f;
#pragma unroll
  for (int j = 0; j < kNormMaxVec; j++) {
    const int v = threadIdx.x + j * kNormThreads;
    if (v < nvec) {
      Vec16<T> a;
      a.load(x + row + (size_t)v * VEC);
      a.unpack(h[j]);
      if (delta) {
        Vec16<T> b;
        float d[VEC];
        b.load(delta + row + (size_t)v * VEC);
        b.unpack(d);
#pragma unroll
        for (int e = 0; e < VEC; e++) h[j][e] = ElemTraits<T>::rnd(__fadd_rn(h[j][e], d[e]));  // model-dtype add
        if (h_out) store_vec<T>(h_out + row + (size_t)v * VEC, h[j]);
      }
#pragma unroll
      for (int e = 0; e < VEC; e++) ss = fmaf(h[j][e], h[j][e], ss);
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) ss += __shfl_xor(ss, off, CC_WAVE);
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = ss;
  __syncthreads();
  float tot = 0.f;
#pragma unroll
  for (int w = 0; w < kNormThreads / 64; w++) tot += sm[w];
  const float rs = rsqrtf(tot / (float)dim + eps);
#pragma unroll
  for (int j = 0; j < kNormMaxVec; j++) {
    const int v = threadIdx.x + j * kNormThreads;
    if (v < nvec) {
      Vec16<T> wv;
      float w[VEC], o[VEC];
      wv.load(weight + (size_t)v * VEC);
      wv.unpack(w);
#pragma unroll
      for (int e = 0; e < VEC; e++) o[e] = __fmul_rn(ElemTraits<T>::rnd(__fmul_rn(h[j][e], rs)), w[e]);
      store_vec<T>(out + row + (size_t)v * VEC, o);
    }
  }
}

// one thread per (token, head, pair); q/k rotated, v copied; outputs head-major [heads, T, D]
template <typename T>
__global__ __launch_bounds__(256) void qkv_rope_kernel(const T* qkv, const T* freqs, int Tn, int HQ, int H, int D, T* q_out,
                                                       T* k_out, T* v_out) {
  const int half = D / 2;
  const int heads = HQ + 2 * H;
  const size_t total = (size_t)Tn * heads * half;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int pr = (int)(i % half);
    const int hd = (int)((i / half) % heads);
    const int t = (int)(i / ((size_t)half * heads));
    const size_t src = ((size_t)t * heads + hd) * D + 2 * pr;
    const float x0 = ElemTraits<T>::load(qkv, src), x1 = ElemTraits<T>::load(qkv, src + 1);
    T* dst;
    int hh;
    if (hd < HQ) {
      dst = q_out;
      hh = hd;
    } else if (hd < HQ + H) {
      dst = k_out;
      hh = hd - HQ;
    } else {
      dst = v_out;
      hh = hd - HQ - H;
    }
    const size_t o = ((size_t)hh * Tn + t) * D + 2 * pr;
    if (hd < HQ + H) {
      // ref: model.py:510-515 — fp32 products and sums, each a separate rounded op, result cast to x's dtype
      const float c = ElemTraits<T>::load(freqs, ((size_t)t * half + pr) * 2), s = ElemTraits<T>::load(freqs, ((size_t)t * half + pr) * 2 + 1);
      ElemTraits<T>::store(dst, o, __fsub_rn(__fmul_rn(x0, c), __fmul_rn(x1, s)));
      ElemTraits<T>::store(dst, o + 1, __fadd_rn(__fmul_rn(x1, c), __fmul_rn(x0, s)));
    } else {
      ElemTraits<T>::store(dst, o, x0);
      ElemTraits<T>::store(dst, o + 1, x1);
    }
  }
}

template <typename T>
__global__ __launch_bounds__(256) void silu_mul_kernel(const T* a, const T* b, long long n, T* out) {
  constexpr int VEC = 16 / (int)sizeof(T);
  const long long nvec = n / VEC;
  for (long long v = (long long)blockIdx.x * blockDim.x + threadIdx.x; v < nvec; v += (long long)gridDim.x * blockDim.x) {
    Vec16<T> va, vb;
    float fa[VEC], fb[VEC], o[VEC];
    va.load(a + v * VEC);
    vb.load(b + v * VEC);
    va.unpack(fa);
    vb.unpack(fb);
#pragma unroll
    for (int e = 0; e < VEC; e++) {
      const float sl = ElemTraits<T>::rnd(__fdiv_rn(fa[e], 1.0f + expf(-fa[e])));  // F.silu -> dtype
      o[e] = __fmul_rn(sl, fb[e]);
    }
    store_vec<T>(out + v * VEC, o);
  }
  for (long long i = nvec * VEC + (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const float x = ElemTraits<T>::load(a, i);
    const float sl = ElemTraits<T>::rnd(__fdiv_rn(x, 1.0f + expf(-x)));
    ElemTraits<T>::store(out, i, __fmul_rn(sl, ElemTraits<T>::load(b, i)));
  }
}

}  // namespace

extern "C" {

int cc_add_rmsnorm(const void* x, const void* delta, const void* weight, int32_t T, int32_t dim, float eps, int32_t dtype,
                   void* h_out, void* out, cc_stream_t stream) {
  CC_ENTRY();
  if (!x || !weight || !out || T <= 0 || dim <= 0 || !cc_dt_ok(dtype)) return CC_ERR_BAD_ARG;
  const int vec = 16 / (int)cc_dt_size(dtype);
  if (dim % vec || dim / vec > kNormThreads * kNormMaxVec) return CC_ERR_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  dim3 grid(T), block(kNormThreads);
  switch (dtype) {
    case CC_DT_F32: hipLaunchKernelGGL(add_rmsnorm_kernel<float>, grid, block, 0, st, (const float*)x, (const float*)delta, (const float*)weight, dim, eps, (float*)h_out, (float*)out); break;
    case CC_DT_BF16: hipLaunchKernelGGL(add_rmsnorm_kernel<bf16_t>, grid, block, 0, st, (const bf16_t*)x, (const bf16_t*)delta, (const bf16_t*)weight, dim, eps, (bf16_t*)h_out, (bf16_t*)out); break;
    default: hipLaunchKernelGGL(add_rmsnorm_kernel<f16_t>, grid, block, 0, st, (const f16_t*)x, (const f16_t*)delta, (const f16_t*)weight, dim, eps, (f16_t*)h_out, (f16_t*)out); break;
  }
  CC_LAUNCH_CHECK();
  return CC_OK;
}

int cc_qkv_rope(const void* qkv, const void* freqs, int32_t T, int32_t HQ, int32_t H, int32_t D, int32_t dtype, void* q_out,
                void* k_out, void* v_out, cc_stream_t stream) {
  CC_ENTRY();
  if (!qkv || !freqs || !q_out || !k_out || !v_out || T <= 0 || HQ <= 0 || H <= 0 || D <= 0 || (D & 1) || !cc_dt_ok(dtype))
    return CC_ERR_BAD_ARG;
  const size_t total = (size_t)T * (HQ + 2 * H) * (D / 2);
  size_t nb = (total + 255) / 256;
  if (nb > 8192) nb = 8192;
  hipStream_t st = (hipStream_t)stream;
  dim3 grid((unsigned)nb), block(256);
  switch (dtype) {
    case CC_DT_F32: hipLaunchKernelGGL(qkv_rope_kernel<float>, grid, block, 0, st, (const float*)qkv, (const float*)freqs, T, HQ, H, D, (float*)q_out, (float*)k_out, (float*)v_out); break;
    case CC_DT_BF16: hipLaunchKernelGGL(qkv_rope_kernel<bf16_t>, grid, block, 0, st, (const bf16_t*)qkv, (const bf16_t*)freqs, T, HQ, H, D, (bf16_t*)q_out, (bf16_t*)k_out, (bf16_t*)v_out); break;
    default: hipLaunchKernelGGL(qkv_rope_kernel<f16_t>, grid, block, 0, st, (const f16_t*)qkv, (const f16_t*)freqs, T, HQ, H, D, (f16_t*)q_out, (f16_t*)k_out, (f16_t*)v_out); break;
  }
  CC_LAUNCH_CHECK();
  return CC_OK;
}

int cc_silu_mul(const void* a, const void* b, int64_t n, int32_t dtype, void* out, cc_stream_t stream) {
  CC_ENTRY();
  if (!a || !b || !out || n <= 0 || !cc_dt_ok(dtype)) return CC_ERR_BAD_ARG;
  const long long nvec = n / (16 / (long long)cc_dt_size(dtype));
  long long nb = (nvec + 255) / 256;
  if (nb > 4096) nb = 4096;
  if (nb < 1) nb = 1;
  hipStream_t st = (hipStream_t)stream;
  dim3 grid((unsigned)nb), block(256);
  switch (dtype) {
    case CC_DT_F32: hipLaunchKernelGGL(silu_mul_kernel<float>, grid, block, 0, st, (const float*)a, (const float*)b, (long long)n, (float*)out); break;
    case CC_DT_BF16: hipLaunchKernelGGL(silu_mul_kernel<bf16_t>, grid, block, 0, st, (const bf16_t*)a, (const bf16_t*)b, (long long)n, (bf16_t*)out); break;
    default: hipLaunchKernelGGL(silu_mul_kernel<f16_t>, grid, block, 0, st, (const f16_t*)a, (const f16_t*)b, (long long)n, (f16_t*)out); break;
  }
  CC_LAUNCH_CHECK();
  return CC_OK;
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------------------------
// Greedy sampling tail (ref: generation_utils.py:136-142): probs = softmax(logits) in the model dtype, next token =
// first index of the largest ROUNDED probability (torch.argmax returns the first maximal element).
// Two small launches over kSmBlocks workgroups instead of torch's softmax + reduce kernels (35 + 41 us at vocab
// 128256; a single-workgroup version is VALU-bound on one CU at 39 us):
//   pass 1: per-slice (max, sum of exp) partials;
//   pass 2: every workgroup folds the partials itself (fixed order), writes its slice of probabilities and
//           min-reduces a 64-bit key (~orderable(p) << 32 | index); the last workgroup to arrive (ticket counter,
//           no spinning) publishes the token.  The arg-min over keys is order independent -> deterministic.
namespace {
constexpr int kSmThreads = 256;
constexpr int kSmBlocks = 128;

struct SmWs {  // caller-provided scratch (cc_softmax_argmax_workspace_bytes)
  float2 part[kSmBlocks];
  unsigned long long key;
  unsigned int ticket;
};

template <typename T>
__device__ __forceinline__ float sm_exp(float x) {
  // 16-bit outputs: v_exp_f32 (relative error ~1e-6, far below the 2^-9 rounding of the result); fp32: accurate expf
  if constexpr (sizeof(T) == 4) return expf(x);
  return __builtin_amdgcn_exp2f(x * 1.4426950408889634f);
}

__device__ __forceinline__ void sm_slice(int V, int& lo, int& hi) {
  const int per = ((V + kSmBlocks - 1) / kSmBlocks + 7) & ~7;  // multiples of 8 elements: slices stay 16-byte aligned
  lo = min(V, (int)blockIdx.x * per);
  hi = min(V, lo + per);
}

template <typename T>
__global__ __launch_bounds__(kSmThreads) void softmax_partial_kernel(const T* logits, int V, SmWs* ws) {
  __shared__ float sm_f[kSmThreads / 64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int lo, hi;
  sm_slice(V, lo, hi);
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    ws->key = ~0ull;
    ws->ticket = 0u;
  }
  float x[4];
  int n = 0;
  float mx = -INFINITY;
  for (int i = lo + threadIdx.x; i < hi; i += kSmThreads) {
    const float v = ElemTraits<T>::load(logits, (size_t)i);
    if (n < 4) x[n] = v;
    n++;
    mx = fmaxf(mx, v);
  }
  mx = wave_max_f32(mx);
  if (lane == 0) sm_f[wave] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(sm_f[0], sm_f[1]), fmaxf(sm_f[2], sm_f[3]));
  __syncthreads();
  float sum = 0.f;
  int k = 0;
  for (int i = lo + threadIdx.x; i < hi; i += kSmThreads, k++)
    sum += sm_exp<T>((k < 4 ? x[k] : ElemTraits<T>::load(logits, (size_t)i)) - mx);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) sum += __shfl_xor(sum, off, CC_WAVE);
  if (lane == 0) sm_f[wave] = sum;
  __syncthreads();
  if (threadIdx.x == 0) ws->part[blockIdx.x] = make_float2(mx, (sm_f[0] + sm_f[1]) + (sm_f[2] + sm_f[3]));
}

template <typename T>
__global__ __launch_bounds__(kSmThreads) void softmax_write_kernel(const T* logits, int V, T* probs, int32_t* idx_out, SmWs* ws) {
  __shared__ float sm_ml[2];
  __shared__ unsigned long long sm_k[kSmThreads / 64 + 1];
  int lo, hi;
  sm_slice(V, lo, hi);
  if (threadIdx.x < 64) {  // one wave folds the partials: M = max m_g, S = sum s_g * exp(m_g - M), fixed order
    float m = -INFINITY;
    for (int g = threadIdx.x; g < kSmBlocks; g += 64) m = fmaxf(m, ws->part[g].x);
    m = wave_max_f32(m);
    float sacc = 0.f;
    for (int g = threadIdx.x; g < kSmBlocks; g += 64) {
      const float2 p = ws->part[g];
      if (p.y > 0.f || p.y != p.y) sacc += p.y * sm_exp<T>(p.x - m);  // (a NaN partial sum makes the whole distribution NaN, as torch.softmax does)
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) sacc += __shfl_xor(sacc, off, CC_WAVE);
    if (threadIdx.x == 0) {
      sm_ml[0] = m;
      sm_ml[1] = sacc;
    }
  }
  __syncthreads();
  const float mx = sm_ml[0], sum = sm_ml[1];
  float bp = -1.f;
  int bi = 0x7fffffff, nan_i = 0x7fffffff;
  for (int i = lo + threadIdx.x; i < hi; i += kSmThreads) {  // indices increase: the first maximal element wins
    const float p = ElemTraits<T>::rnd(__fdiv_rn(sm_exp<T>(ElemTraits<T>::load(logits, (size_t)i) - mx), sum));
    ElemTraits<T>::store(probs, (size_t)i, p);
    if (p > bp) {
      bp = p;
      bi = i;
    }
    // torch.argmax: NaN counts as the maximum, the FIRST one wins (generation_utils.py:140 on a NaN distribution returns its index,
    // never an index past the vocabulary — r5: garbage logits behind a failed layer step used to yield -1 here, and the next token's
    // embedding lookup asserted on the device)
    if (p != p && nan_i == 0x7fffffff) nan_i = i;
  }
  unsigned long long best = bi == 0x7fffffff ? ~0ull : (((unsigned long long)(~orderable_f32(bp)) << 32) | (unsigned)bi);
  if (nan_i != 0x7fffffff) best = (unsigned long long)(unsigned)nan_i;  // key 0 in the upper half: beats every number; the smallest index among NaNs
  best = block_min_u64(best, sm_k);
  if (threadIdx.x == 0) {
    atomicMin(&ws->key, best);
    __threadfence();
    if (atomicAdd(&ws->ticket, 1u) == (unsigned)gridDim.x - 1) {  // last workgroup: every key is in
      __threadfence();
      const unsigned long long k = atomicMin(&ws->key, ~0ull);
      *idx_out = k == ~0ull ? 0 : (int32_t)(k & 0xffffffffull);  // (nothing compared greater than -1: cannot happen for V > 0 — a valid index anyway)
    }
  }
}
}  // namespace

extern "C" size_t cc_softmax_argmax_workspace_bytes(void) { return sizeof(SmWs); }

extern "C" int cc_softmax_argmax(const void* logits, int32_t V, int32_t dtype, void* probs, int32_t* idx_out, void* workspace,
                                 size_t workspace_bytes, cc_stream_t stream) {
  CC_ENTRY();
  if (!logits || !probs || !idx_out || !workspace || V <= 0 || !cc_dt_ok(dtype)) return CC_ERR_BAD_ARG;
  if (workspace_bytes < sizeof(SmWs)) return CC_ERR_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  SmWs* ws = reinterpret_cast<SmWs*>(workspace);
  dim3 grid(kSmBlocks), block(kSmThreads);
  switch (dtype) {
    case CC_DT_F32:
      hipLaunchKernelGGL(softmax_partial_kernel<float>, grid, block, 0, st, (const float*)logits, V, ws);
      hipLaunchKernelGGL(softmax_write_kernel<float>, grid, block, 0, st, (const float*)logits, V, (float*)probs, idx_out, ws);
      break;
    case CC_DT_BF16:
      hipLaunchKernelGGL(softmax_partial_kernel<bf16_t>, grid, block, 0, st, (const bf16_t*)logits, V, ws);
      hipLaunchKernelGGL(softmax_write_kernel<bf16_t>, grid, block, 0, st, (const bf16_t*)logits, V, (bf16_t*)probs, idx_out, ws);
      break;
    default:
      hipLaunchKernelGGL(softmax_partial_kernel<f16_t>, grid, block, 0, st, (const f16_t*)logits, V, ws);
      hipLaunchKernelGGL(softmax_write_kernel<f16_t>, grid, block, 0, st, (const f16_t*)logits, V, (f16_t*)probs, idx_out, ws);
      break;
  }
  CC_LAUNCH_CHECK();
  return CC_OK;
}
