// cc_hybrid.hip — FastGen-style hybrid cache (per-head policy) and the W-slot attention-history ring.
//
// ref: KVCacheHybrid cache.py:768-1288 (decode: _decoding_update :965-1019, _select_fill_idx :896-950,
// _eviction_idx_for_head :844-894); KVCacheHeavyHitter with history_window_size > 1 (:707-736).
// The reference walks the heads in a Python loop with a device sync per head; here one workgroup per head does
// budget check, protected-slot masking, windowed-history scoring, arg-min and the insert in one launch.
#include "cc_common.h"

namespace {

enum { F_HH = 1, F_WIN = 2, F_PUNC = 4, F_SPECIAL = 8, F_FULL = 16 };

struct HybArgs {
  void* k_cache;
  void* v_cache;
  int32_t* pos;        // [H,S]
  uint8_t* mask;       // [H,S]
  int32_t* cache_cts;  // [H]
  int H, S, D, W, g;
  const void* k_new;
  const void* v_new;
  const int32_t* input_pos;
  const int64_t* strategies;  // [H] policy index per head
  const int32_t* table;       // [n_pol, 3]: flags, window slots, heavy-hitter slots
  void* num;                  // [H,S,W] T
  int32_t* denom;             // [H,S]
  const uint8_t* special_mask;  // [H,S] or null
  uint8_t* punc_mask;           // [H,S] or null
  const uint8_t* is_punc;       // device bool[1] or null
  const int32_t* num_special;   // device int[1] or null
  int32_t* num_punc;            // device int[1] or null
  int requires_hh;
  int64_t* fill_out;  // [H]
  const float* wsum;  // [H,S] window sums from the pre-pass
};

// dtype(sum of the W history entries of one cache slot) (cache.py:855-859 `.sum(dim=-1)` on a model-dtype tensor;
// torch's own fp32 order is unspecified).  Canonical order, shared with the oracle: ONE WAVE per cache slot — the
// row is cut into 16-byte chunks, lane l accumulates chunks l, l+64, ... element by element in index order, the 64
// partials meet in an xor butterfly (32 .. 1), the total is rounded to the model dtype.  A 400-entry bf16 ring row
// is 800 contiguous bytes: one fully coalesced load instruction per slot.
template <typename T>
__device__ __forceinline__ float wave_window_sum(const T* row, int W, int lane) {
  constexpr int VEC = 16 / (int)sizeof(T);
  float acc = 0.f;
  const bool aligned = (reinterpret_cast<uintptr_t>(row) & 15) == 0;
  for (int c = lane; c * VEC < W; c += 64) {
    if (aligned && c * VEC + VEC <= W) {
      Vec16<T> v;
      float f[VEC];
      v.load(row + (size_t)c * VEC);
      v.unpack(f);
#pragma unroll
      for (int e = 0; e < VEC; e++) acc = __fadd_rn(acc, f[e]);
    } else {
      for (int e = 0; e < VEC && c * VEC + e < W; e++) acc = __fadd_rn(acc, ElemTraits<T>::load(row, (size_t)c * VEC + e));
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) acc = __fadd_rn(acc, __shfl_xor(acc, off, CC_WAVE));
  return ElemTraits<T>::rnd(acc);
}

// Pre-pass of the ring policies: wsum[h, s] for every slot of every head that scores by accumulated attention.
// The [H, S, W] ring is the only large operand of these policies (118 MB at S = 18432, W = 400): it is streamed
// once, coalesced, by the whole chip — the decision kernels below (one workgroup per head) then read 4 bytes per
// slot.  (The first version summed rows inside the one-workgroup-per-head kernel: 8 CUs, 800-byte strides, 1.4 ms.)
template <typename T>
__global__ __launch_bounds__(256) void ring_window_sum_kernel(const T* num, const int64_t* strategies, const int32_t* table, int H,
                                                              int S, int W, float* out) {
  const int lane = threadIdx.x & 63;
  const size_t total = (size_t)H * S;
  const size_t nw = (size_t)gridDim.x * (blockDim.x >> 6);
  for (size_t i = (size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); i < total; i += nw) {
    if (strategies != nullptr && !(table[strategies[i / S] * 3] & 1 /* F_HH */)) continue;
    const float v = wave_window_sum<T>(num + i * (size_t)W, W, lane);
    if (lane == 0) out[i] = v;
  }
}

constexpr int kHybThreads = 1024;

template <typename T>
__global__ __launch_bounds__(kHybThreads) void hybrid_decode_kernel(HybArgs a) {
  __shared__ unsigned long long sm_key[kHybThreads / 64 + 2];
  const int h = blockIdx.x, S = a.S, W = a.W;
  const int32_t p = *a.input_pos;
  const int pol = (int)a.strategies[h];
  const int flags = a.table[pol * 3], win = a.table[pol * 3 + 1], hhs = a.table[pol * 3 + 2];
  const int cts = a.cache_cts[h];
  const bool is_punc = a.is_punc ? (*a.is_punc != 0) : false;
  const int end_idx = cts < S - 1 ? cts : S - 1;  // ref: _end_idx() :897-899
  const size_t hoff = (size_t)h * S;

  int fill = -1;       // -1 = token not kept by this head (ref: :948-950 -> dummy slot S-1)
  bool evict = false;
  if ((flags & F_PUNC) && is_punc) {  // :905-906
    fill = end_idx;
  } else if (flags & F_FULL) {  // :908-909
    fill = end_idx;
  } else {
    int budget = a.g;  // :912-925
    if (flags & F_SPECIAL) budget += a.num_special ? *a.num_special : 0;
    if (flags & F_PUNC) budget += a.num_punc ? *a.num_punc : 0;
    if (flags & F_WIN) budget += win;
    if (flags & F_HH) budget += hhs;
    if (cts < budget) {  // :927-930
      fill = end_idx;
    } else if (flags & (F_HH | F_WIN)) {  // :932-946 -> _eviction_idx_for_head :844-894
      evict = true;
      unsigned long long best = ~0ull;
      // UN slots per thread per iteration, every per-slot load issued before the first use (one workgroup scans a
      // whole head: the loop is latency-bound unless the loads of several slots are in flight together)
      constexpr int UN = 4;
      const int lim = cts < S ? cts : S;
      for (int s0 = threadIdx.x; s0 < lim; s0 += blockDim.x * UN) {
        int32_t ps[UN], dn[UN];
        float ws[UN];
        uint8_t sp[UN], pm[UN];
#pragma unroll
        for (int u = 0; u < UN; u++) {
          const int s = s0 + u * blockDim.x;
          const bool in = s < lim;
          const size_t i = hoff + (in ? s : 0);
          ps[u] = a.pos[i];
          ws[u] = (flags & F_HH) ? a.wsum[i] : 0.f;
          dn[u] = (flags & F_HH) ? a.denom[i] : 1;
          sp[u] = ((flags & F_SPECIAL) && a.special_mask) ? a.special_mask[i] : 0;
          pm[u] = ((flags & F_PUNC) && a.punc_mask) ? a.punc_mask[i] : 0;
        }
#pragma unroll
        for (int u = 0; u < UN; u++) {
          const int s = s0 + u * blockDim.x;
          if (s >= lim) continue;
          float sc;
          if (flags & F_HH) {
            const int32_t d = dn[u] > W ? W : dn[u];  // clamp_max only (:868-870): a zero count divides by zero like the reference
            sc = __fdiv_rn(ws[u], (float)d);
          } else {
            sc = (float)ps[u];  // :873
          }
          bool save = s < a.g;  // :876 first g SLOTS
          save |= sp[u] != 0;
          save |= pm[u] != 0;
          if (flags & F_WIN) save |= ps[u] > p - win;  // :885-889 strict
          if (save) sc = INFINITY;
          const unsigned long long key = make_key(orderable_f32(sc), (uint32_t)s);
          best = key < best ? key : best;
        }
      }
      best = block_min_u64(best, sm_key);
      fill = (int)(best & 0xffffffffull);
    }
  }
  const int slot = fill < 0 ? S - 1 : fill;
  if (threadIdx.x == 0) a.fill_out[h] = slot;
  if (evict && a.requires_hh) {  // :992-996
    T* num = reinterpret_cast<T*>(a.num) + (hoff + slot) * (size_t)W;
    for (int j = threadIdx.x; j < W; j += blockDim.x) ElemTraits<T>::store(num, j, 0.f);
    if (threadIdx.x == 0) a.denom[hoff + slot] = 0;
  }
  if (threadIdx.x == 0) {
    if (!evict && fill >= 0) {  // :997-1001
      a.cache_cts[h] = cts + 1;
      a.mask[hoff + slot] = 1;
    }
    a.pos[hoff + slot] = p;  // :1006-1007 _fill(update_mask=False) — every head, dropped tokens land in slot S-1
    if (is_punc && a.punc_mask) a.punc_mask[hoff + slot] = 1;  // :1011-1016
  }
  const int words = a.D * (int)sizeof(T) / 4;
  const uint32_t* ks = reinterpret_cast<const uint32_t*>(a.k_new) + (size_t)h * words;
  const uint32_t* vs = reinterpret_cast<const uint32_t*>(a.v_new) + (size_t)h * words;
  uint32_t* kd = reinterpret_cast<uint32_t*>(a.k_cache) + (hoff + slot) * words;
  uint32_t* vd = reinterpret_cast<uint32_t*>(a.v_cache) + (hoff + slot) * words;
  for (int i = threadIdx.x; i < 2 * words; i += blockDim.x) {
    if (i < words) kd[i] = ks[i];
    else vd[i - words] = vs[i - words];
  }
}

// Plain heavy hitter with a finite history window (history_window_size W > 1, model-dtype ring).
// ref: KVCacheHeavyHitter._eviction_idx cache.py:725-765: avg = float(dtype(sum_W num)) / clamp(denom, 1, W);
// (pos < g) | (pos >= p - w) -> 1.0; pos == -1 -> 0.0; arg-min; ring row and denom of the slot zeroed; insert.
struct RingArgs {
  void* k_cache;
  void* v_cache;
  int32_t* pos;
  uint8_t* mask;
  int32_t* cache_cts;
  int H, Hc, S, D, W, g, w;
  const void* k_new;
  const void* v_new;
  const int32_t* input_pos;
  void* num;
  int32_t* denom;
  int64_t* idx_out;
  const float* wsum;  // [H,S] window sums from the pre-pass
};

template <typename T>
__global__ __launch_bounds__(kHybThreads) void hh_ring_decode_kernel(RingArgs a) {
  __shared__ unsigned long long sm_key[kHybThreads / 64 + 2];
  const int h = blockIdx.x, S = a.S, W = a.W;
  const int32_t p = *a.input_pos;
  const size_t hoff = (size_t)h * S;
  T* num = reinterpret_cast<T*>(a.num);
  unsigned long long best = ~0ull;
  for (int s = threadIdx.x; s < S; s += blockDim.x) {
    const int32_t ps = a.pos[hoff + s];
    int32_t dn = a.denom[hoff + s];
    dn = dn < 1 ? 1 : (dn > W ? W : dn);
    float sc = __fdiv_rn(a.wsum[hoff + s], (float)dn);
    if (ps < a.g || ps >= p - a.w) sc = 1.0f;
    if (ps == -1) sc = 0.0f;
    const unsigned long long key = make_key(orderable_f32(sc), ((uint32_t)s << 1) | (uint32_t)(ps == -1));
    best = key < best ? key : best;
  }
  best = block_min_u64(best, sm_key);
  const int idx = (int)((best & 0xffffffffull) >> 1), ins = (int)(best & 1ull);
  for (int j = threadIdx.x; j < W; j += blockDim.x) ElemTraits<T>::store(num + (hoff + idx) * (size_t)W, j, 0.f);
  if (threadIdx.x == 0) {
    a.idx_out[h] = idx;
    a.denom[hoff + idx] = 0;
  }
  if (a.k_new == nullptr) return;
  if (threadIdx.x == 0) {
    a.pos[hoff + idx] = p;
    a.mask[hoff + idx] = 1;
    if (a.Hc == a.H) a.cache_cts[h] += ins;
    else if (h == 0) a.cache_cts[0] += ins;
  }
  const int words = a.D * (int)sizeof(T) / 4;
  const uint32_t* ks = reinterpret_cast<const uint32_t*>(a.k_new) + (size_t)h * words;
  const uint32_t* vs = reinterpret_cast<const uint32_t*>(a.v_new) + (size_t)h * words;
  uint32_t* kd = reinterpret_cast<uint32_t*>(a.k_cache) + (hoff + idx) * words;
  uint32_t* vd = reinterpret_cast<uint32_t*>(a.v_cache) + (hoff + idx) * words;
  for (int i = threadIdx.x; i < 2 * words; i += blockDim.x) {
    if (i < words) kd[i] = ks[i];
    else vd[i - words] = vs[i - words];
  }
}

// num_punc += 1 once per step (ref: :1017), after every head has read the old value
__global__ void hybrid_bump_punc_kernel(const uint8_t* is_punc, int32_t* num_punc) {
  if (threadIdx.x == 0 && blockIdx.x == 0 && *is_punc) *num_punc += 1;
}

// ref: cache.py:716-723 with W > 1: num[h,s,counter % W] = attn (zero beyond T), denom += 1, counter += 1
template <typename T>
__global__ __launch_bounds__(256) void hh_ring_update_kernel(T* num, int32_t* denom, const int64_t* counter, const T* attn,
                                                             int H, int S, int Tn, int W) {
  const int slot = (int)(*counter % W);
  const int n = H * S;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const int h = i / S, s = i - h * S;
    const float v = s < Tn ? ElemTraits<T>::load(attn, (size_t)h * Tn + s) : 0.f;
    ElemTraits<T>::store(num, (size_t)i * W + slot, v);
    denom[i] += 1;
  }
}
__global__ void bump_counter_kernel(int64_t* counter) {
  if (threadIdx.x == 0 && blockIdx.x == 0) *counter += 1;
}

// band sums of a materialised attention tensor: out[h,k] = sum_{q=k}^{min(Lq-1, k+band-1)} attn[h,q,k]
template <typename T>
__global__ __launch_bounds__(256) void attn_bandsum_kernel(const T* attn, int H, int Lq, int Lk, int band, float* out) {
  const int total = H * Lk;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int h = i / Lk, s = i - h * Lk;
    float acc = 0.f;
    const int hi = min(Lq, s + band);
    for (int qi = s; qi < hi; qi++) acc = __fadd_rn(acc, ElemTraits<T>::load(attn, ((size_t)h * Lq + qi) * Lk + s));
    out[i] = acc;
  }
}

static void launch_window_sums(const void* num, const int64_t* strategies, const int32_t* table, int H, int S, int W, int dtype,
                               float* out, hipStream_t st) {
  const size_t slots = (size_t)H * S;
  size_t nb = (slots + 3) / 4;  // 4 waves per workgroup, one slot per wave per iteration
  if (nb > 4096) nb = 4096;
  dim3 grid((unsigned)nb), block(256);
  switch (dtype) {
    case CC_DT_F32: hipLaunchKernelGGL(ring_window_sum_kernel<float>, grid, block, 0, st, (const float*)num, strategies, table, H, S, W, out); break;
    case CC_DT_BF16: hipLaunchKernelGGL(ring_window_sum_kernel<bf16_t>, grid, block, 0, st, (const bf16_t*)num, strategies, table, H, S, W, out); break;
    default: hipLaunchKernelGGL(ring_window_sum_kernel<f16_t>, grid, block, 0, st, (const f16_t*)num, strategies, table, H, S, W, out); break;
  }
}

}  // namespace

extern "C" {

int cc_hybrid_decode_update(const cc_kv_view* c, const void* k_new, const void* v_new, const int32_t* input_pos,
                            const int64_t* strategies, const int32_t* policy_table, int32_t n_policies, void* num,
                            int32_t* denom, int32_t W, const uint8_t* special_mask, uint8_t* punc_mask,
                            const uint8_t* is_punc, const int32_t* num_special, int32_t* num_punc, int32_t global_tokens,
                            int32_t requires_heavy_hitter, int64_t* fill_out, float* wsum_workspace, cc_stream_t stream) {
  CC_ENTRY();
  if (!cc_view_ok(c) || !k_new || !v_new || !input_pos || !strategies || !policy_table || n_policies <= 0 || !num ||
      !denom || W <= 0 || !fill_out || c->Hp != c->H || c->Hc != c->H)
    return CC_ERR_BAD_ARG;
  if (!wsum_workspace) return CC_ERR_WORKSPACE;
  HybArgs a{};
  a.k_cache = c->k_cache; a.v_cache = c->v_cache; a.pos = c->pos; a.mask = c->mask; a.cache_cts = c->cache_cts;
  a.H = c->H; a.S = c->S; a.D = c->D; a.W = W; a.g = global_tokens;
  a.k_new = k_new; a.v_new = v_new; a.input_pos = input_pos; a.strategies = strategies; a.table = policy_table;
  a.num = num; a.denom = denom; a.special_mask = special_mask; a.punc_mask = punc_mask; a.is_punc = is_punc;
  a.num_special = num_special; a.num_punc = num_punc; a.requires_hh = requires_heavy_hitter; a.fill_out = fill_out;
  a.wsum = wsum_workspace;
  hipStream_t st = (hipStream_t)stream;
  launch_window_sums(num, strategies, policy_table, c->H, c->S, W, c->dtype, wsum_workspace, st);
  dim3 grid(c->H), block(kHybThreads);
  switch (c->dtype) {
    case CC_DT_F32: hipLaunchKernelGGL(hybrid_decode_kernel<float>, grid, block, 0, st, a); break;
    case CC_DT_BF16: hipLaunchKernelGGL(hybrid_decode_kernel<bf16_t>, grid, block, 0, st, a); break;
    default: hipLaunchKernelGGL(hybrid_decode_kernel<f16_t>, grid, block, 0, st, a); break;
  }
  CC_LAUNCH_CHECK();
  if (is_punc && num_punc) {
    hipLaunchKernelGGL(hybrid_bump_punc_kernel, dim3(1), dim3(64), 0, st, is_punc, num_punc);
    CC_LAUNCH_CHECK();
  }
  return CC_OK;
}

int cc_decode_update_heavy_hitter_ring(const cc_kv_view* c, const void* k_new, const void* v_new,
                                       const int32_t* input_pos, void* num, int32_t* denom, int32_t W, int32_t g,
                                       int32_t w, int64_t* idx_out, float* wsum_workspace, cc_stream_t stream) {
  CC_ENTRY();
  if (!cc_view_ok(c) || !input_pos || !idx_out || !num || !denom || W <= 0 || c->Hp != c->H || (k_new && !v_new))
    return CC_ERR_BAD_ARG;
  if (!wsum_workspace) return CC_ERR_WORKSPACE;
  RingArgs a{};
  a.k_cache = c->k_cache; a.v_cache = c->v_cache; a.pos = c->pos; a.mask = c->mask; a.cache_cts = c->cache_cts;
  a.H = c->H; a.Hc = c->Hc; a.S = c->S; a.D = c->D; a.W = W; a.g = g; a.w = w;
  a.k_new = k_new; a.v_new = v_new; a.input_pos = input_pos; a.num = num; a.denom = denom; a.idx_out = idx_out;
  a.wsum = wsum_workspace;
  hipStream_t st = (hipStream_t)stream;
  launch_window_sums(num, nullptr, nullptr, c->H, c->S, W, c->dtype, wsum_workspace, st);
  dim3 grid(c->H), block(kHybThreads);
  switch (c->dtype) {
    case CC_DT_F32: hipLaunchKernelGGL(hh_ring_decode_kernel<float>, grid, block, 0, st, a); break;
    case CC_DT_BF16: hipLaunchKernelGGL(hh_ring_decode_kernel<bf16_t>, grid, block, 0, st, a); break;
    default: hipLaunchKernelGGL(hh_ring_decode_kernel<f16_t>, grid, block, 0, st, a); break;
  }
  CC_LAUNCH_CHECK();
  return CC_OK;
}

int cc_hh_ring_update(void* num, int32_t* denom, int64_t* counter, const void* attn, int32_t H, int32_t S, int32_t T,
                      int32_t W, int32_t dtype, cc_stream_t stream) {
  CC_ENTRY();
  if (!num || !denom || !counter || !attn || H <= 0 || S <= 0 || T < 0 || T > S || W <= 0 || !cc_dt_ok(dtype))
    return CC_ERR_BAD_ARG;
  const int n = H * S;
  int nb = (n + 255) / 256;
  if (nb > 2048) nb = 2048;
  hipStream_t st = (hipStream_t)stream;
  dim3 grid(nb), block(256);
  switch (dtype) {
    case CC_DT_F32: hipLaunchKernelGGL(hh_ring_update_kernel<float>, grid, block, 0, st, (float*)num, denom, counter, (const float*)attn, H, S, T, W); break;
    case CC_DT_BF16: hipLaunchKernelGGL(hh_ring_update_kernel<bf16_t>, grid, block, 0, st, (bf16_t*)num, denom, counter, (const bf16_t*)attn, H, S, T, W); break;
    default: hipLaunchKernelGGL(hh_ring_update_kernel<f16_t>, grid, block, 0, st, (f16_t*)num, denom, counter, (const f16_t*)attn, H, S, T, W); break;
  }
  CC_LAUNCH_CHECK();
  hipLaunchKernelGGL(bump_counter_kernel, dim3(1), dim3(64), 0, st, counter);
  CC_LAUNCH_CHECK();
  return CC_OK;
}

int cc_attn_bandsum(const void* attn, int32_t H, int32_t Lq, int32_t Lk, int32_t dtype, int32_t band, float* out,
                    cc_stream_t stream) {
  CC_ENTRY();
  if (!attn || !out || H <= 0 || Lq <= 0 || Lk <= 0 || band <= 0 || !cc_dt_ok(dtype)) return CC_ERR_BAD_ARG;
  hipStream_t st = (hipStream_t)stream;
  int nb = (H * Lk + 255) / 256;
  if (nb > 2048) nb = 2048;
  dim3 grid(nb), block(256);
  switch (dtype) {
    case CC_DT_F32: hipLaunchKernelGGL(attn_bandsum_kernel<float>, grid, block, 0, st, (const float*)attn, H, Lq, Lk, band, out); break;
    case CC_DT_BF16: hipLaunchKernelGGL(attn_bandsum_kernel<bf16_t>, grid, block, 0, st, (const bf16_t*)attn, H, Lq, Lk, band, out); break;
    default: hipLaunchKernelGGL(attn_bandsum_kernel<f16_t>, grid, block, 0, st, (const f16_t*)attn, H, Lq, Lk, band, out); break;
  }
  CC_LAUNCH_CHECK();
  return CC_OK;
}

}  // extern "C"
