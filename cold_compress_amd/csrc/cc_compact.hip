// cc_compact.hip — prefill-time prompt compaction for gfx950: top-K keep set per head (radix select on
// orderable keys + order-preserving compaction, so the kept indices come out ascending with no sort), row
// gathers, SnapKV pooled priority, attention column sums / column means.
//
// ref: prompt_compression.py:21-43 (_keep_idxs, __call__), :69-88 (_filter_kv), :170-194 (SnapKV).
// Integer/byte work, HBM/L2-bound; wave64 shuffles for the scans, LDS histogram for the radix digits.
#include "cc_common.h"

namespace {

constexpr int kSelThreads = 1024;

// "larger is better" orderable keys; NaN ranks above +inf (torch.topk), -0.0 == +0.0.
__device__ __forceinline__ uint32_t topk_key_f32(float f) {
  if (f != f) return 0xffffffffu;
  if (f == 0.0f) f = 0.0f;
  uint32_t u = __float_as_uint(f);
  u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
  return u == 0xffffffffu ? 0xfffffffeu : u;
}

template <int PRIO>
struct KeyOf;
template <>
struct KeyOf<CC_PRIO_F32> {
  typedef uint32_t type;
  __device__ static __forceinline__ uint32_t get(const void* p, size_t i) { return topk_key_f32(reinterpret_cast<const float*>(p)[i]); }
};
template <>
struct KeyOf<CC_PRIO_BF16> {
  typedef uint32_t type;
  __device__ static __forceinline__ uint32_t get(const void* p, size_t i) {
    return topk_key_f32(bf16_bits_to_f32(reinterpret_cast<const uint16_t*>(p)[i]));
  }
};
template <>
struct KeyOf<CC_PRIO_F16> {
  typedef uint32_t type;
  __device__ static __forceinline__ uint32_t get(const void* p, size_t i) {
    return topk_key_f32(f16_bits_to_f32(reinterpret_cast<const uint16_t*>(p)[i]));
  }
};
template <>
struct KeyOf<CC_PRIO_I64> {
  typedef unsigned long long type;
  __device__ static __forceinline__ unsigned long long get(const void* p, size_t i) {
    return (unsigned long long)reinterpret_cast<const long long*>(p)[i] ^ 0x8000000000000000ull;
  }
};

// exclusive block scan of two ints per thread (blockDim.x == kSelThreads); totals returned through refs
__device__ __forceinline__ void block_exscan2(int& a, int& b, int& tot_a, int& tot_b, int* sm /* 2*(nw+1) */) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = kSelThreads / 64;
  int ia = a, ib = b;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    int ta = __shfl_up(ia, off, CC_WAVE), tb = __shfl_up(ib, off, CC_WAVE);
    if (lane >= off) {
      ia += ta;
      ib += tb;
    }
  }
  if (lane == 63) {
    sm[wave] = ia;
    sm[nw + 1 + wave] = ib;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    int ca = 0, cb = 0;
    for (int w = 0; w < nw; w++) {
      int ta = sm[w], tb = sm[nw + 1 + w];
      sm[w] = ca;
      sm[nw + 1 + w] = cb;
      ca += ta;
      cb += tb;
    }
    sm[nw] = ca;
    sm[2 * nw + 1] = cb;
  }
  __syncthreads();
  a = ia - a + sm[wave];
  b = ib - b + sm[nw + 1 + wave];
  tot_a = sm[nw];
  tot_b = sm[2 * nw + 1];
  __syncthreads();
}

template <int PRIO>
__global__ __launch_bounds__(kSelThreads) void topk_keep_kernel(const void* prio, int L, int K, int64_t* keep_out) {
  typedef typename KeyOf<PRIO>::type key_t;
  constexpr int PASSES = (int)sizeof(key_t);
  __shared__ int hist[256];
  __shared__ int sm_scan[2 * (kSelThreads / 64 + 1)];
  __shared__ key_t sm_prefix;
  __shared__ int sm_need;
  const int row = blockIdx.x;
  const size_t base = (size_t)row * L;

  // ---- radix select: key of the K-th largest element, MSB digit first
  if (threadIdx.x == 0) {
    sm_prefix = 0;
    sm_need = K;
  }
  for (int pass = 0; pass < PASSES; pass++) {
    const int shift = 8 * (PASSES - 1 - pass);
    if (threadIdx.x < 256) hist[threadIdx.x] = 0;
    __syncthreads();
    const key_t prefix = sm_prefix;
    for (int i = threadIdx.x; i < L; i += kSelThreads) {
      const key_t k = KeyOf<PRIO>::get(prio, base + i);
      const bool match = (pass == 0) || ((k >> (shift + 8)) == prefix);
      if (match) atomicAdd(&hist[(int)((k >> shift) & 0xff)], 1);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      int need = sm_need, cum = 0, b = 255;
      for (; b > 0; b--) {
        if (cum + hist[b] >= need) break;
        cum += hist[b];
      }
      sm_need = need - cum;  // still to take among keys whose digit == b
      sm_prefix = (prefix << 8) | (key_t)b;
    }
    __syncthreads();
  }
  const key_t kth = sm_prefix;
  const int need_eq = sm_need;  // how many elements equal to the K-th key are kept (lowest index first)

  // ---- order-preserving compaction: thread t owns the contiguous index range [t*seg, (t+1)*seg)
  const int seg = (L + kSelThreads - 1) / kSelThreads;
  const int lo = threadIdx.x * seg, hi = min(L, lo + seg);
  int n_gt = 0, n_eq = 0;
  for (int i = lo; i < hi; i++) {
    const key_t k = KeyOf<PRIO>::get(prio, base + i);
    n_gt += (k > kth);
    n_eq += (k == kth);
  }
  int tot_gt, tot_eq;
  block_exscan2(n_gt, n_eq, tot_gt, tot_eq, sm_scan);  // n_gt/n_eq now = counts BEFORE this thread's range
  int64_t* out = keep_out + (size_t)row * K;
  for (int i = lo; i < hi; i++) {
    const key_t k = KeyOf<PRIO>::get(prio, base + i);
    if (k > kth) {
      out[n_gt + min(n_eq, need_eq)] = i;
      n_gt++;
    } else if (k == kth) {
      if (n_eq < need_eq) out[n_gt + n_eq] = i;
      n_eq++;
    }
  }
}

template <typename T>
__global__ __launch_bounds__(256) void gather_rows_kernel(const T* src, const int64_t* keep, int Hk, int H, int L, int K,
                                                          int D, T* dst) {
  // one 16-byte chunk per thread; rows are D*sizeof(T) bytes
  const int cpr = D * (int)sizeof(T) / 16;  // chunks per row (>= 1 checked on the host)
  const size_t total = (size_t)H * K * cpr;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const size_t r = i / cpr;
    const int ch = (int)(i - r * cpr);
    const int h = (int)(r / K), j = (int)(r - (size_t)h * K);
    const int64_t t = keep[(size_t)(Hk == 1 ? 0 : h) * K + j];
    const uint4* s = reinterpret_cast<const uint4*>(src + ((size_t)h * L + t) * D) + ch;
    reinterpret_cast<uint4*>(dst + r * D)[ch] = *s;
  }
}

template <typename T>
__global__ __launch_bounds__(256) void gather_rows_small_kernel(const T* src, const int64_t* keep, int Hk, int H, int L,
                                                                int K, int D, T* dst) {
  const size_t total = (size_t)H * K * D;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const size_t r = i / D;
    const int d = (int)(i - r * D);
    const int h = (int)(r / K), j = (int)(r - (size_t)h * K);
    const int64_t t = keep[(size_t)(Hk == 1 ? 0 : h) * K + j];
    dst[i] = src[((size_t)h * L + t) * D + d];
  }
}

// KVCacheAnalysis' decode-time bookkeeping in ONE launch.  ref: cache.py:1391-1404 — the attention row over the FULL cache is
// restricted to the slots the shadow cache still holds (unfilled slots read the last, zero, column), handed on as the shadow
// cache's attention, and the mass it lost is recorded: loss = mean_h dtype(1 - dtype(sum_s sub[h, s])), in the model dtype.
// One workgroup (a debug path: Hp * S <= a few 100 k elements); fp32 sums folded in a fixed order (the reference's own fp32
// order inside torch.sum is unspecified: compared to one rounding of the dtype).
template <typename T>
__global__ __launch_bounds__(1024) void analysis_loss_kernel(const T* attn, const int32_t* pos, int Hp, int S_full, int S, T* sub, T* losses,
                                                             int32_t* ctr, int cap) {
  __shared__ float sm_part[16];
  __shared__ float sm_head[64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int h = 0; h < Hp; h++) {
    float acc = 0.f;
    for (int s = threadIdx.x; s < S; s += 1024) {
      const int p = pos[(size_t)h * S + s];
      const float v = ElemTraits<T>::load(attn, (size_t)h * S_full + (p == -1 ? S_full - 1 : p));
      ElemTraits<T>::store(sub, (size_t)h * S + s, v);
      acc += v;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, CC_WAVE);
    if (lane == 0) sm_part[wave] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
      float t = 0.f;
      for (int w = 0; w < 16; w++) t += sm_part[w];
      sm_head[h] = ElemTraits<T>::rnd(1.0f - ElemTraits<T>::rnd(t));  // dtype(1 - dtype(sum))
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    float t = 0.f;
    for (int h = 0; h < Hp; h++) t += sm_head[h];
    const int c = *ctr;
    if (c >= 0 && c < cap) ElemTraits<T>::store(losses, c, t / (float)Hp);  // .mean() -> dtype
    *ctr = c + 1;
  }
}

template <typename T>
__global__ __launch_bounds__(256) void gather_vec_kernel(const T* src, const int64_t* keep, int Hs, int L, int K, T* dst) {
  const int total = Hs * K;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int h = i / K;
    dst[i] = src[(size_t)h * L + keep[i]];
  }
}

// ref: prompt_compression.py:174-186 — AvgPool1d(k=5, pad=2, count_include_pad=False) then forced 1.0s
template <typename T>
__global__ __launch_bounds__(256) void snapkv_priority_kernel(const T* obs, int H, int L, int obs_len, int g, T* out) {
  const int total = H * L;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int h = i / L, t = i - h * L;
    const int lo = t - 2 < 0 ? 0 : t - 2, hi = t + 2 >= L ? L - 1 : t + 2;
    float acc = 0.f;
    for (int u = lo; u <= hi; u++) acc = __fadd_rn(acc, ElemTraits<T>::load(obs, (size_t)h * L + u));
    float v = __fdiv_rn(acc, (float)(hi - lo + 1));
    if (t >= L - obs_len || t < g) v = 1.0f;
    ElemTraits<T>::store(out, i, v);
  }
}

// ref: cache.py:704 attn.sum(dim=1): sequential over the query axis (same order as the oracle)
template <typename T>
__global__ __launch_bounds__(256) void attn_colsum_kernel(const T* attn, int H, int Lq, int Lk, float* out) {
  const int total = H * Lk;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int h = i / Lk, s = i - h * Lk;
    float acc = 0.f;
    for (int qi = 0; qi < Lq; qi++) acc = __fadd_rn(acc, ElemTraits<T>::load(attn, ((size_t)h * Lq + qi) * Lk + s));
    out[i] = acc;
  }
}

// ref: cache.py:704 / prompt_compression.py:191: dtype(colsum) / (L - input_pos) -> dtype
template <typename T>
__global__ __launch_bounds__(256) void colsum_to_mean_kernel(const float* colsum, const int64_t* input_pos, int H, int L,
                                                             T* out) {
  const int total = H * L;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int t = i % L;
    const long long p = input_pos ? input_pos[t] : t;
    ElemTraits<T>::store(out, i, __fdiv_rn(ElemTraits<T>::rnd(colsum[i]), (float)(L - p)));
  }
}

static dim3 grid_for(size_t n, int cap = 2048) {
  size_t b = (n + 255) / 256;
  if (b > (size_t)cap) b = cap;
  if (b < 1) b = 1;
  return dim3((unsigned)b);
}

}  // namespace

#define CC_DISPATCH_T(dtype, KERNEL, GRID, BLOCK, ST, ...)                                                     \
  switch (dtype) {                                                                                             \
    case CC_DT_F32: hipLaunchKernelGGL(KERNEL<float>, GRID, BLOCK, 0, ST, __VA_ARGS__); break;                 \
    case CC_DT_BF16: hipLaunchKernelGGL(KERNEL<bf16_t>, GRID, BLOCK, 0, ST, __VA_ARGS__); break;               \
    default: hipLaunchKernelGGL(KERNEL<f16_t>, GRID, BLOCK, 0, ST, __VA_ARGS__); break;                        \
  }

extern "C" {

size_t cc_topk_keep_workspace_bytes(int32_t Hs, int32_t L, int32_t K) {
  (void)Hs; (void)L; (void)K;
  return 0;
}

int cc_topk_keep(const void* priority, int32_t prio_dtype, int32_t Hs, int32_t L, int32_t K, int64_t* keep_out,
                 void* workspace, size_t workspace_bytes, cc_stream_t stream) {
  CC_ENTRY();
  (void)workspace; (void)workspace_bytes;
  if (!priority || !keep_out || Hs <= 0 || L <= 0 || K <= 0 || K > L) return CC_ERR_BAD_ARG;
  dim3 grid(Hs), block(kSelThreads);
  hipStream_t st = (hipStream_t)stream;
  switch (prio_dtype) {
    case CC_PRIO_F32: hipLaunchKernelGGL(topk_keep_kernel<CC_PRIO_F32>, grid, block, 0, st, priority, L, K, keep_out); break;
    case CC_PRIO_BF16: hipLaunchKernelGGL(topk_keep_kernel<CC_PRIO_BF16>, grid, block, 0, st, priority, L, K, keep_out); break;
    case CC_PRIO_F16: hipLaunchKernelGGL(topk_keep_kernel<CC_PRIO_F16>, grid, block, 0, st, priority, L, K, keep_out); break;
    case CC_PRIO_I64: hipLaunchKernelGGL(topk_keep_kernel<CC_PRIO_I64>, grid, block, 0, st, priority, L, K, keep_out); break;
    default: return CC_ERR_BAD_ARG;
  }
  CC_LAUNCH_CHECK();
  return CC_OK;
}

int cc_gather_rows(const void* src, const int64_t* keep, int32_t Hk, int32_t H, int32_t L, int32_t K, int32_t D,
                   int32_t dtype, void* dst, cc_stream_t stream) {
  CC_ENTRY();
  if (!src || !keep || !dst || H <= 0 || L <= 0 || K <= 0 || D <= 0 || (Hk != 1 && Hk != H) || !cc_dt_ok(dtype))
    return CC_ERR_BAD_ARG;
  hipStream_t st = (hipStream_t)stream;
  const size_t row_bytes = (size_t)D * cc_dt_size(dtype);
  if (row_bytes % 16 == 0) {
    dim3 grid = grid_for((size_t)H * K * (row_bytes / 16), 4096), block(256);
    switch (dtype) {
      case CC_DT_F32: hipLaunchKernelGGL(gather_rows_kernel<float>, grid, block, 0, st, (const float*)src, keep, Hk, H, L, K, D, (float*)dst); break;
      case CC_DT_BF16: hipLaunchKernelGGL(gather_rows_kernel<bf16_t>, grid, block, 0, st, (const bf16_t*)src, keep, Hk, H, L, K, D, (bf16_t*)dst); break;
      default: hipLaunchKernelGGL(gather_rows_kernel<f16_t>, grid, block, 0, st, (const f16_t*)src, keep, Hk, H, L, K, D, (f16_t*)dst); break;
    }
  } else {
    dim3 grid = grid_for((size_t)H * K * D, 4096), block(256);
    switch (dtype) {
      case CC_DT_F32: hipLaunchKernelGGL(gather_rows_small_kernel<float>, grid, block, 0, st, (const float*)src, keep, Hk, H, L, K, D, (float*)dst); break;
      case CC_DT_BF16: hipLaunchKernelGGL(gather_rows_small_kernel<bf16_t>, grid, block, 0, st, (const bf16_t*)src, keep, Hk, H, L, K, D, (bf16_t*)dst); break;
      default: hipLaunchKernelGGL(gather_rows_small_kernel<f16_t>, grid, block, 0, st, (const f16_t*)src, keep, Hk, H, L, K, D, (f16_t*)dst); break;
    }
  }
  CC_LAUNCH_CHECK();
  return CC_OK;
}

int cc_gather_vec(const void* src, const int64_t* keep, int32_t Hs, int32_t L, int32_t K, int32_t dtype, void* dst,
                  cc_stream_t stream) {
  CC_ENTRY();
  if (!src || !keep || !dst || Hs <= 0 || L <= 0 || K <= 0 || !cc_dt_ok(dtype)) return CC_ERR_BAD_ARG;
  hipStream_t st = (hipStream_t)stream;
  dim3 grid = grid_for((size_t)Hs * K), block(256);
  switch (dtype) {
    case CC_DT_F32: hipLaunchKernelGGL(gather_vec_kernel<float>, grid, block, 0, st, (const float*)src, keep, Hs, L, K, (float*)dst); break;
    case CC_DT_BF16: hipLaunchKernelGGL(gather_vec_kernel<bf16_t>, grid, block, 0, st, (const bf16_t*)src, keep, Hs, L, K, (bf16_t*)dst); break;
    default: hipLaunchKernelGGL(gather_vec_kernel<f16_t>, grid, block, 0, st, (const f16_t*)src, keep, Hs, L, K, (f16_t*)dst); break;
  }
  CC_LAUNCH_CHECK();
  return CC_OK;
}

int cc_analysis_loss(const void* attn, const int32_t* pos, int32_t Hp, int32_t S_full, int32_t S, int32_t dtype, void* sub_out,
                     void* losses, int32_t* loss_ctr, int32_t cap, cc_stream_t stream) {
  CC_ENTRY();
  if (!attn || !pos || !sub_out || !losses || !loss_ctr || Hp <= 0 || Hp > 64 || S_full <= 0 || S <= 0 || cap <= 0 || !cc_dt_ok(dtype))
    return CC_ERR_BAD_ARG;
  hipStream_t st = (hipStream_t)stream;
  switch (dtype) {
    case CC_DT_F32: hipLaunchKernelGGL(analysis_loss_kernel<float>, dim3(1), dim3(1024), 0, st, (const float*)attn, pos, Hp, S_full, S, (float*)sub_out, (float*)losses, loss_ctr, cap); break;
    case CC_DT_BF16: hipLaunchKernelGGL(analysis_loss_kernel<bf16_t>, dim3(1), dim3(1024), 0, st, (const bf16_t*)attn, pos, Hp, S_full, S, (bf16_t*)sub_out, (bf16_t*)losses, loss_ctr, cap); break;
    default: hipLaunchKernelGGL(analysis_loss_kernel<f16_t>, dim3(1), dim3(1024), 0, st, (const f16_t*)attn, pos, Hp, S_full, S, (f16_t*)sub_out, (f16_t*)losses, loss_ctr, cap); break;
  }
  CC_LAUNCH_CHECK();
  return CC_OK;
}

int cc_snapkv_priority(const void* obs_mean, int32_t H, int32_t L, int32_t dtype, int32_t obs_len, int32_t g, void* out,
                       cc_stream_t stream) {
  CC_ENTRY();
  if (!obs_mean || !out || H <= 0 || L <= 0 || !cc_dt_ok(dtype) || obs_len < 0) return CC_ERR_BAD_ARG;
  hipStream_t st = (hipStream_t)stream;
  dim3 grid = grid_for((size_t)H * L), block(256);
  switch (dtype) {
    case CC_DT_F32: hipLaunchKernelGGL(snapkv_priority_kernel<float>, grid, block, 0, st, (const float*)obs_mean, H, L, obs_len, g, (float*)out); break;
    case CC_DT_BF16: hipLaunchKernelGGL(snapkv_priority_kernel<bf16_t>, grid, block, 0, st, (const bf16_t*)obs_mean, H, L, obs_len, g, (bf16_t*)out); break;
    default: hipLaunchKernelGGL(snapkv_priority_kernel<f16_t>, grid, block, 0, st, (const f16_t*)obs_mean, H, L, obs_len, g, (f16_t*)out); break;
  }
  CC_LAUNCH_CHECK();
  return CC_OK;
}

int cc_attn_colsum(const void* attn, int32_t H, int32_t Lq, int32_t Lk, int32_t dtype, float* out, cc_stream_t stream) {
  CC_ENTRY();
  if (!attn || !out || H <= 0 || Lq <= 0 || Lk <= 0 || !cc_dt_ok(dtype)) return CC_ERR_BAD_ARG;
  hipStream_t st = (hipStream_t)stream;
  dim3 grid = grid_for((size_t)H * Lk), block(256);
  switch (dtype) {
    case CC_DT_F32: hipLaunchKernelGGL(attn_colsum_kernel<float>, grid, block, 0, st, (const float*)attn, H, Lq, Lk, out); break;
    case CC_DT_BF16: hipLaunchKernelGGL(attn_colsum_kernel<bf16_t>, grid, block, 0, st, (const bf16_t*)attn, H, Lq, Lk, out); break;
    default: hipLaunchKernelGGL(attn_colsum_kernel<f16_t>, grid, block, 0, st, (const f16_t*)attn, H, Lq, Lk, out); break;
  }
  CC_LAUNCH_CHECK();
  return CC_OK;
}

int cc_colsum_to_mean(const float* colsum, const int64_t* input_pos, int32_t H, int32_t L, int32_t dtype, void* out,
                      cc_stream_t stream) {
  CC_ENTRY();
  if (!colsum || !out || H <= 0 || L <= 0 || !cc_dt_ok(dtype)) return CC_ERR_BAD_ARG;
  hipStream_t st = (hipStream_t)stream;
  dim3 grid = grid_for((size_t)H * L), block(256);
  switch (dtype) {
    case CC_DT_F32: hipLaunchKernelGGL(colsum_to_mean_kernel<float>, grid, block, 0, st, colsum, input_pos, H, L, (float*)out); break;
    case CC_DT_BF16: hipLaunchKernelGGL(colsum_to_mean_kernel<bf16_t>, grid, block, 0, st, colsum, input_pos, H, L, (bf16_t*)out); break;
    default: hipLaunchKernelGGL(colsum_to_mean_kernel<f16_t>, grid, block, 0, st, colsum, input_pos, H, L, (f16_t*)out); break;
  }
  CC_LAUNCH_CHECK();
  return CC_OK;
}

}  // extern "C"
