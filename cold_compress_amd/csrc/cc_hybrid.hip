// cc_hybrid.hip — FastGen-style hybrid cache (per-head policy) and the W-slot attention-history ring.
//
// ref: KVCacheHybrid cache.py:768-1288 (decode: _decoding_update :965-1019, _select_fill_idx :896-950,
// _eviction_idx_for_head :844-894); KVCacheHeavyHitter with history_window_size > 1 (:707-736).
// The reference walks the heads in a Python loop with a device sync per head; here one launch does budget check,
// protected-slot masking, windowed-history scoring, arg-min and the insert for every head (up to 32 workgroups per
// head when the caller keeps the tracked window-sum state, one otherwise).  The window sums of the ring are exact and
// kept incrementally (cc_wacc.h); the full-ring pre-pass below remains as the stateless / rebuild path.
#include "cc_common.h"
#include "cc_wacc.h"

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
  const int64_t* token_id;      // device int64[1] + punc_ids[n_punc_ids]: the same test evaluated here (is_punc == null)
  const int64_t* punc_ids;
  int n_punc_ids;
  unsigned int* punc_ticket;    // tracked state: num_punc is bumped by the last head to have read it (null: separate launch)
  u64* head_scratch;            // tracked state, gridDim.y > 1: [H] inverted arg-min keys + [H] tickets, zero between launches
  const int32_t* num_special;   // device int[1] or null
  int32_t* num_punc;            // device int[1] or null
  int requires_hh;
  int64_t* fill_out;  // [H]
  float* wsum;        // [H,S] window sums: from the pre-pass, or the tracked state kept by cc_hh_ring_update
  u64* wacc;          // [H,S,4] tracked exact accumulators or null (stateless: pre-pass every call)
  unsigned long long* key_out;  // seed of the two-launch pipeline (cc_hybrid_next_key_init): [H][nk] candidate keys; no side effects
  int nk;
};

// exact sum of one ring row, one wave per row: 16-byte chunks strided over the lanes, integer butterfly
template <typename T>
__device__ __forceinline__ WAcc wave_window_acc(const T* row, int W, int lane) {
  constexpr int VEC = 16 / (int)sizeof(T);
  WAcc acc{0, 0, 0, 0};
  const bool aligned = (reinterpret_cast<uintptr_t>(row) & 15) == 0;
  for (int c = lane; c * VEC < W; c += 64) {
    if (aligned && c * VEC + VEC <= W) {
      Vec16<T> v;
      float f[VEC];
      v.load(row + (size_t)c * VEC);
      v.unpack(f);
#pragma unroll
      for (int e = 0; e < VEC; e++) wacc_add_value(acc, f[e], false);
    } else {
      for (int e = 0; e < VEC && c * VEC + e < W; e++) wacc_add_value(acc, ElemTraits<T>::load(row, (size_t)c * VEC + e), false);
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    WAcc o;
    o.w0 = __shfl_xor(acc.w0, off, CC_WAVE);
    o.w1 = __shfl_xor(acc.w1, off, CC_WAVE);
    o.w2 = __shfl_xor(acc.w2, off, CC_WAVE);
    o.special = __shfl_xor(acc.special, off, CC_WAVE);
    wacc_merge(acc, o);
  }
  return acc;
}

// Stateless / rebuild path of the ring policies: wsum[h, s] (and, on request, the tracked state: accumulators + the
// column-major shadow) for every slot of every head that scores by accumulated attention.  The [H, S, W] ring is the
// only large operand of these policies (118 MB at S = 18432, W = 400): it is streamed once, coalesced, by the whole
// chip, one wave per slot.  Callers that keep the tracked state never run this on the decode path.
// (The first version summed rows inside the one-workgroup-per-head kernel: 8 CUs, 800-byte strides, 1.4 ms.)
template <typename T>
__global__ __launch_bounds__(256) void ring_window_sum_kernel(const T* num, const int64_t* strategies, const int32_t* table, int H,
                                                              int S, int W, float* out, u64* acc_out, size_t scratch_off) {
  const int lane = threadIdx.x & 63;
  const size_t total = (size_t)H * S;
  const size_t nw = (size_t)gridDim.x * (blockDim.x >> 6);
  if (acc_out && blockIdx.x == 0 && threadIdx.x < 2) acc_out[total * 4 + threadIdx.x] = 0;  // the two launch tickets
  if (acc_out && blockIdx.x == 0)
    for (int i = threadIdx.x; i < 2 * H; i += blockDim.x) acc_out[scratch_off + i] = 0;  // meeting words of the split decision kernel
  for (size_t i = (size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); i < total; i += nw) {
    if (strategies != nullptr && !(table[strategies[i / S] * 3] & 1 /* F_HH */)) continue;
    const WAcc a = wave_window_acc<T>(num + i * (size_t)W, W, lane);
    if (acc_out) {  // column-major shadow of the ring, behind the accumulators and the ticket word
      T* shadow = reinterpret_cast<T*>(acc_out + total * 4 + 2);
      for (int j = lane; j < W; j += 64) shadow[(size_t)j * total + i] = num[i * (size_t)W + j];
    }
    if (lane == 0) {
      out[i] = wacc_round<T>(a);
      if (acc_out) {
        acc_out[i * 4 + 0] = a.w0; acc_out[i * 4 + 1] = a.w1; acc_out[i * 4 + 2] = a.w2; acc_out[i * 4 + 3] = a.special;
      }
    }
  }
}

constexpr int kHybThreads = 1024;

template <typename T>
__global__ __launch_bounds__(kHybThreads) void hybrid_decode_kernel(HybArgs a) {
  __shared__ unsigned long long sm_key[kHybThreads / 64 + 2];
  // gridDim.y workgroups share the slots of one head (tracked state only: they meet through head_scratch).  Every
  // workgroup reads all its inputs first; only the LAST one to arrive writes anything, so no read sees a write.
  const int h = blockIdx.x, S = a.S, W = a.W, nb = gridDim.y, bi = blockIdx.y;
  // the first batch of per-slot operands is requested before the policy chain (strategy -> table -> counts) resolves
  constexpr int UN = 4;
  const size_t hoff = (size_t)h * S;
  const int per = (S + nb - 1) / nb;
  const int lo = bi * per, hi = lo + per < S ? lo + per : S;
  int32_t ps[UN], dn[UN];
  float ws[UN];
  uint8_t sp[UN], pm[UN];
  auto load_batch = [&](int s0) {
#pragma unroll
    for (int u = 0; u < UN; u++) {
      const int s = s0 + u * (int)blockDim.x;
      const size_t i = hoff + (s < hi ? s : lo);
      ps[u] = a.pos[i];
      ws[u] = a.wsum[i];
      dn[u] = a.denom[i];
      sp[u] = a.special_mask ? a.special_mask[i] : 0;
      pm[u] = a.punc_mask ? a.punc_mask[i] : 0;
    }
  };
  load_batch(lo + threadIdx.x);
  const int32_t p = *a.input_pos;
  const int pol = (int)a.strategies[h];
  const int flags = a.table[pol * 3], win = a.table[pol * 3 + 1], hhs = a.table[pol * 3 + 2];
  const int cts = a.cache_cts[h];
  // ref: cache.py:975 torch.isin(input_ids, punc_ids): given by the caller or evaluated here; num_punc is read once
  // per head BEFORE the head takes its ticket, so the last ticket holder may bump it (ref :1017) in this launch
  __shared__ int sm_punc[2];
  unsigned int ptk = 0;
  if (threadIdx.x == 0) {
    bool f = false;
    if (a.is_punc) {
      f = *a.is_punc != 0;
    } else if (a.token_id && a.punc_ids) {
      const int64_t id = *a.token_id;
      for (int k = 0; k < a.n_punc_ids; k++) f |= a.punc_ids[k] == id;
    }
    sm_punc[0] = f;
    sm_punc[1] = a.num_punc ? *a.num_punc : 0;
  }
  __syncthreads();
  const bool is_punc = sm_punc[0] != 0;
  const int num_punc_old = sm_punc[1];
  if (threadIdx.x == 0 && a.punc_ticket && a.num_punc && !a.key_out)
    ptk = __hip_atomic_fetch_add(a.punc_ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const int end_idx = cts < S - 1 ? cts : S - 1;  // ref: _end_idx() :897-899

  int fill = -1;       // -1 = token not kept by this head (ref: :948-950 -> dummy slot S-1)
  bool evict = false;
  unsigned long long best = ~0ull;
  if (a.key_out != nullptr) {
    // Seed of the fused two-launch step: the head's eviction CANDIDATE for this position (arg-min over its live slots with
    // the protections of :876-889), whatever the budget says — whether the head appends, evicts it or drops the token is
    // decided by the streaming pass of the step that consumes it.  Key = (score, slot << 1); nothing else is touched.
    if ((flags & (F_HH | F_WIN)) && !(flags & F_FULL)) {
      const int lim = cts < S ? cts : S;
      for (int s = threadIdx.x; s < lim; s += blockDim.x) {
        const size_t i = hoff + s;
        const int32_t psv = a.pos[i];
        float sc;
        if (flags & F_HH) {
          const int32_t d0 = a.denom[i];
          sc = __fdiv_rn(a.wsum[i], (float)(d0 > W ? W : d0));
        } else {
          sc = (float)psv;
        }
        bool save = s < a.g;
        if ((flags & F_SPECIAL) && a.special_mask) save |= a.special_mask[i] != 0;
        if ((flags & F_PUNC) && a.punc_mask) save |= a.punc_mask[i] != 0;
        if (flags & F_WIN) save |= psv > p - win;
        if (save) sc = INFINITY;
        const unsigned long long key = make_key(orderable_f32(sc), (uint32_t)s << 1);
        best = key < best ? key : best;
      }
    }
    best = block_min_u64(best, sm_key);
    for (int i = threadIdx.x; i < a.nk; i += blockDim.x) a.key_out[(size_t)h * a.nk + i] = (i == 0) ? best : ~0ull;
    return;
  }
  if ((flags & F_PUNC) && is_punc) {  // :905-906
    fill = end_idx;
  } else if (flags & F_FULL) {  // :908-909
    fill = end_idx;
  } else {
    int budget = a.g;  // :912-925
    if (flags & F_SPECIAL) budget += a.num_special ? *a.num_special : 0;
    if (flags & F_PUNC) budget += num_punc_old;
    if (flags & F_WIN) budget += win;
    if (flags & F_HH) budget += hhs;
    if (cts < budget) {  // :927-930
      fill = end_idx;
    } else if (flags & (F_HH | F_WIN)) {  // :932-946 -> _eviction_idx_for_head :844-894
      evict = true;
      const int lim_all = cts < S ? cts : S;
      const int lim = hi < lim_all ? hi : lim_all;
      for (int s0 = lo + threadIdx.x; s0 < lim; s0 += blockDim.x * UN) {
        if (s0 != lo + (int)threadIdx.x) load_batch(s0);
#pragma unroll
        for (int u = 0; u < UN; u++) {
          const int s = s0 + u * blockDim.x;
          if (s >= lim) continue;
          float sc;
          if (!(flags & F_SPECIAL)) sp[u] = 0;
          if (!(flags & F_PUNC)) pm[u] = 0;
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
    }
  }
  if (threadIdx.x == 0 && a.punc_ticket && a.num_punc && ptk == gridDim.x * gridDim.y - 1) {
    if (is_punc) *a.num_punc = num_punc_old + 1;  // :1017 — every workgroup has read the old value
    __hip_atomic_store(a.punc_ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  if (nb > 1) {  // meet the other workgroups of this head; the last to arrive carries on alone
    __shared__ unsigned long long sm_meet[2];
    if (threadIdx.x == 0) {
      u64* inv_key = a.head_scratch + h;
      unsigned int* ticket = reinterpret_cast<unsigned int*>(a.head_scratch + a.H + h);
      if (evict) atomicMax(inv_key, ~best);
      __threadfence();
      const bool last = atomicAdd(ticket, 1u) == (unsigned)nb - 1;
      if (last) {
        __threadfence();
        best = ~atomicExch(inv_key, 0ull);
        atomicExch(ticket, 0u);
      }
      sm_meet[0] = last;
      sm_meet[1] = best;
    }
    __syncthreads();
    if (!sm_meet[0]) return;
    best = sm_meet[1];
  }
  if (evict) fill = (int)(best & 0xffffffffull);
  const int slot = fill < 0 ? S - 1 : fill;
  if (threadIdx.x == 0) a.fill_out[h] = slot;
  if (evict && a.requires_hh) {  // :992-996
    T* num = reinterpret_cast<T*>(a.num) + (hoff + slot) * (size_t)W;
    for (int j = threadIdx.x; j < W; j += blockDim.x) ElemTraits<T>::store(num, j, 0.f);
    if (threadIdx.x == 0) a.denom[hoff + slot] = 0;
    if (a.wacc) {  // tracked state of a zeroed row: sum, accumulator and the shadow column entries
      const size_t hs = (size_t)a.H * S;
      T* shadow = reinterpret_cast<T*>(a.wacc + hs * 4 + 2);
      for (int j = threadIdx.x; j < W; j += blockDim.x) ElemTraits<T>::store(shadow, (size_t)j * hs + hoff + slot, 0.f);
      if (threadIdx.x < 4) a.wacc[(hoff + slot) * 4 + threadIdx.x] = 0;
      if (threadIdx.x == 4) a.wsum[hoff + slot] = 0.f;
    }
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
  float* wsum;  // [H,S] window sums: from the pre-pass, or the tracked state kept by cc_hh_ring_update
  u64* wacc;    // [H,S,4] tracked exact accumulators or null
  unsigned long long* key_out;  // seed of the two-launch pipeline: [H][nk] arg-min keys, nothing else is touched
  int nk;
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
  if (a.key_out != nullptr) {
    for (int i = threadIdx.x; i < a.nk; i += blockDim.x) a.key_out[(size_t)h * a.nk + i] = (i == 0) ? best : ~0ull;
    return;
  }
  const int idx = (int)((best & 0xffffffffull) >> 1), ins = (int)(best & 1ull);
  for (int j = threadIdx.x; j < W; j += blockDim.x) ElemTraits<T>::store(num + (hoff + idx) * (size_t)W, j, 0.f);
  if (threadIdx.x == 0) {
    a.idx_out[h] = idx;
    a.denom[hoff + idx] = 0;
  }
  if (a.wacc) {  // tracked state of a zeroed row: sum, accumulator and the shadow column entries
    const size_t hs = (size_t)a.H * S;
    T* shadow = reinterpret_cast<T*>(a.wacc + hs * 4 + 2);
    for (int j = threadIdx.x; j < W; j += blockDim.x) ElemTraits<T>::store(shadow, (size_t)j * hs + hoff + idx, 0.f);
    if (threadIdx.x < 4) a.wacc[(hoff + idx) * 4 + threadIdx.x] = 0;
    if (threadIdx.x == 4) a.wsum[hoff + idx] = 0.f;
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
__global__ void hybrid_bump_punc_kernel(const uint8_t* is_punc, const int64_t* token_id, const int64_t* punc_ids, int n,
                                        int32_t* num_punc) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  bool f = false;
  if (is_punc) {
    f = *is_punc != 0;
  } else {
    const int64_t id = *token_id;
    for (int k = 0; k < n; k++) f |= punc_ids[k] == id;
  }
  if (f) *num_punc += 1;
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
// The same update keeping the exact window sums current: the overwritten ring entry leaves the accumulator, the new
// one enters, the rounded sum is republished.  The entry being overwritten is read from the column-major SHADOW of the
// ring kept inside the tracked state ([W][H*S]: the column is contiguous; the same read from the [H, S, W] ring is an
// 800-byte-stride gather, measured +7.8 us at H*S = 147k).  Per slot: 2 B attn + 2 B shadow + 32 B accumulator in,
// 2 B ring (strided, fire-and-forget) + 2 B shadow + 32 B + 4 B out — instead of re-reading W entries next step.
// Counter: thread 0 of every workgroup reads it, THEN takes a ticket; the workgroup holding the last ticket knows
// that every workgroup has read the old value and bumps it at once (the atomic's latency hides behind the slot loop).
template <typename T>
__global__ __launch_bounds__(1024) void hh_ring_update_tracked_kernel(T* num, int32_t* denom, int64_t* counter, const T* attn,
                                                                      int H, int S, int Tn, int W, u64* wacc, float* wsum) {
  __shared__ int sm_slot;
  const int n = H * S;
  int64_t c0 = 0;
  if (threadIdx.x == 0) {
    c0 = *counter;
    sm_slot = (int)(c0 % W);
  }
  __syncthreads();
  unsigned int tk = 0;
  unsigned int* ticket = reinterpret_cast<unsigned int*>(wacc + (size_t)n * 4);
  // the load above has returned (its value was used), so a relaxed atomic is ordered after it; nobody waits for the ticket
  if (threadIdx.x == 0) tk = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const int slot = sm_slot;
  T* shadow = reinterpret_cast<T*>(wacc + (size_t)n * 4 + 2) + (size_t)slot * n;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const int h = i / S, s = i - h * S;
    const float v = s < Tn ? ElemTraits<T>::load(attn, (size_t)h * Tn + s) : 0.f;
    const float old = ElemTraits<T>::load(shadow, i);
    const ulonglong2 a01 = *reinterpret_cast<const ulonglong2*>(wacc + (size_t)i * 4);
    const ulonglong2 a23 = *reinterpret_cast<const ulonglong2*>(wacc + (size_t)i * 4 + 2);
    WAcc a{a01.x, a01.y, a23.x, a23.y};
    ElemTraits<T>::store(num, (size_t)i * W + slot, v);
    ElemTraits<T>::store(shadow, i, v);
    denom[i] += 1;
    wacc_add_value(a, ElemTraits<T>::rnd(v), false);
    wacc_add_value(a, old, true);
    *reinterpret_cast<ulonglong2*>(wacc + (size_t)i * 4) = make_ulonglong2(a.w0, a.w1);
    *reinterpret_cast<ulonglong2*>(wacc + (size_t)i * 4 + 2) = make_ulonglong2(a.w2, a.special);
    wsum[i] = wacc_round<T>(a);
  }
  if (threadIdx.x == 0 && tk == gridDim.x - 1) {  // every workgroup has read the old counter: bump it, re-arm the ticket
    *counter = c0 + 1;
    __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
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
                               float* out, u64* acc_out, hipStream_t st) {
  const size_t slots = (size_t)H * S;
  const size_t scratch_off = cc_hh_ring_acc_words(H, S, W, dtype) - 2 * (size_t)H;
  size_t nb = (slots + 3) / 4;  // 4 waves per workgroup, one slot per wave per iteration
  if (nb > 4096) nb = 4096;
  dim3 grid((unsigned)nb), block(256);
  switch (dtype) {
    case CC_DT_F32: hipLaunchKernelGGL(ring_window_sum_kernel<float>, grid, block, 0, st, (const float*)num, strategies, table, H, S, W, out, acc_out, scratch_off); break;
    case CC_DT_BF16: hipLaunchKernelGGL(ring_window_sum_kernel<bf16_t>, grid, block, 0, st, (const bf16_t*)num, strategies, table, H, S, W, out, acc_out, scratch_off); break;
    default: hipLaunchKernelGGL(ring_window_sum_kernel<f16_t>, grid, block, 0, st, (const f16_t*)num, strategies, table, H, S, W, out, acc_out, scratch_off); break;
  }
}

}  // namespace

extern "C" {

int cc_hybrid_decode_update(const cc_kv_view* c, const void* k_new, const void* v_new, const int32_t* input_pos,
                            const int64_t* strategies, const int32_t* policy_table, int32_t n_policies, void* num,
                            int32_t* denom, int32_t W, const uint8_t* special_mask, uint8_t* punc_mask,
                            const uint8_t* is_punc, const int64_t* token_id, const int64_t* punc_ids, int32_t n_punc_ids,
                            const int32_t* num_special, int32_t* num_punc, int32_t global_tokens,
                            int32_t requires_heavy_hitter, int64_t* fill_out, float* wsum_workspace, uint64_t* wsum_acc,
                            cc_stream_t stream) {
  CC_ENTRY();
  if (!cc_view_ok(c) || !k_new || !v_new || !input_pos || !strategies || !policy_table || n_policies <= 0 || !num ||
      !denom || W <= 0 || !fill_out || c->Hp != c->H || c->Hc != c->H || (punc_ids && n_punc_ids < 0))
    return CC_ERR_BAD_ARG;
  if (!wsum_workspace) return CC_ERR_WORKSPACE;
  HybArgs a{};
  const bool punc_here = !is_punc && token_id && punc_ids;
  a.token_id = token_id; a.punc_ids = punc_ids; a.n_punc_ids = n_punc_ids;
  // second ticket word of the tracked state (the pad behind the ring-update ticket)
  a.punc_ticket = wsum_acc ? reinterpret_cast<unsigned int*>(wsum_acc + (size_t)c->H * c->S * 4 + 1) : nullptr;
  int nsplit = 1;
  if (wsum_acc) {  // meeting place of the workgroups that share a head: the last 2H words of the tracked state
    a.head_scratch = reinterpret_cast<u64*>(wsum_acc) + cc_hh_ring_acc_words(c->H, c->S, W, c->dtype) - 2 * (size_t)c->H;
    nsplit = (c->S + 1023) / 1024;
    if (nsplit > 32) nsplit = 32;
  }
  a.k_cache = c->k_cache; a.v_cache = c->v_cache; a.pos = c->pos; a.mask = c->mask; a.cache_cts = c->cache_cts;
  a.H = c->H; a.S = c->S; a.D = c->D; a.W = W; a.g = global_tokens;
  a.k_new = k_new; a.v_new = v_new; a.input_pos = input_pos; a.strategies = strategies; a.table = policy_table;
  a.num = num; a.denom = denom; a.special_mask = special_mask; a.punc_mask = punc_mask; a.is_punc = is_punc;
  a.num_special = num_special; a.num_punc = num_punc; a.requires_hh = requires_heavy_hitter; a.fill_out = fill_out;
  a.wsum = wsum_workspace;
  a.wacc = reinterpret_cast<u64*>(wsum_acc);
  hipStream_t st = (hipStream_t)stream;
  if (!wsum_acc) launch_window_sums(num, strategies, policy_table, c->H, c->S, W, c->dtype, wsum_workspace, nullptr, st);
  dim3 grid(c->H, nsplit), block(nsplit > 1 ? 256 : kHybThreads);
  switch (c->dtype) {
    case CC_DT_F32: hipLaunchKernelGGL(hybrid_decode_kernel<float>, grid, block, 0, st, a); break;
    case CC_DT_BF16: hipLaunchKernelGGL(hybrid_decode_kernel<bf16_t>, grid, block, 0, st, a); break;
    default: hipLaunchKernelGGL(hybrid_decode_kernel<f16_t>, grid, block, 0, st, a); break;
  }
  CC_LAUNCH_CHECK();
  if ((is_punc || punc_here) && num_punc && !a.punc_ticket) {
    hipLaunchKernelGGL(hybrid_bump_punc_kernel, dim3(1), dim3(64), 0, st, is_punc, token_id, punc_ids, n_punc_ids, num_punc);
    CC_LAUNCH_CHECK();
  }
  return CC_OK;
}

int cc_hybrid_next_key_init(const cc_kv_view* c, const int32_t* input_pos, const int64_t* strategies, const int32_t* policy_table,
                            int32_t n_policies, const int32_t* denom, int32_t W, const float* wsum, const uint8_t* special_mask,
                            const uint8_t* punc_mask, int32_t global_tokens, uint64_t* next_key, cc_stream_t stream) {
  CC_ENTRY();
  if (!cc_view_ok(c) || !input_pos || !strategies || !policy_table || n_policies <= 0 || W <= 0 || !next_key || c->Hp != c->H ||
      c->Hc != c->H)
    return CC_ERR_BAD_ARG;
  HybArgs a{};
  a.pos = c->pos; a.mask = c->mask; a.cache_cts = c->cache_cts;
  a.H = c->H; a.S = c->S; a.D = c->D; a.W = W; a.g = global_tokens;
  a.input_pos = input_pos; a.strategies = strategies; a.table = policy_table;
  a.denom = const_cast<int32_t*>(denom); a.wsum = const_cast<float*>(wsum);
  a.special_mask = special_mask; a.punc_mask = const_cast<uint8_t*>(punc_mask);
  a.key_out = reinterpret_cast<unsigned long long*>(next_key);
  a.nk = cc_next_key_slots(c->S);
  dim3 grid(c->H, 1), block(kHybThreads);
  hipStream_t st = (hipStream_t)stream;
  switch (c->dtype) {
    case CC_DT_F32: hipLaunchKernelGGL(hybrid_decode_kernel<float>, grid, block, 0, st, a); break;
    case CC_DT_BF16: hipLaunchKernelGGL(hybrid_decode_kernel<bf16_t>, grid, block, 0, st, a); break;
    default: hipLaunchKernelGGL(hybrid_decode_kernel<f16_t>, grid, block, 0, st, a); break;
  }
  CC_LAUNCH_CHECK();
  return CC_OK;
}

int cc_decode_update_heavy_hitter_ring(const cc_kv_view* c, const void* k_new, const void* v_new,
                                       const int32_t* input_pos, void* num, int32_t* denom, int32_t W, int32_t g,
                                       int32_t w, int64_t* idx_out, float* wsum_workspace, uint64_t* wsum_acc,
                                       cc_stream_t stream) {
  CC_ENTRY();
  if (!cc_view_ok(c) || !input_pos || !idx_out || !num || !denom || W <= 0 || c->Hp != c->H || (k_new && !v_new))
    return CC_ERR_BAD_ARG;
  if (!wsum_workspace) return CC_ERR_WORKSPACE;
  RingArgs a{};
  a.k_cache = c->k_cache; a.v_cache = c->v_cache; a.pos = c->pos; a.mask = c->mask; a.cache_cts = c->cache_cts;
  a.H = c->H; a.Hc = c->Hc; a.S = c->S; a.D = c->D; a.W = W; a.g = g; a.w = w;
  a.k_new = k_new; a.v_new = v_new; a.input_pos = input_pos; a.num = num; a.denom = denom; a.idx_out = idx_out;
  a.wsum = wsum_workspace;
  a.wacc = reinterpret_cast<u64*>(wsum_acc);
  hipStream_t st = (hipStream_t)stream;
  if (!wsum_acc) launch_window_sums(num, nullptr, nullptr, c->H, c->S, W, c->dtype, wsum_workspace, nullptr, st);
  dim3 grid(c->H), block(kHybThreads);
  switch (c->dtype) {
    case CC_DT_F32: hipLaunchKernelGGL(hh_ring_decode_kernel<float>, grid, block, 0, st, a); break;
    case CC_DT_BF16: hipLaunchKernelGGL(hh_ring_decode_kernel<bf16_t>, grid, block, 0, st, a); break;
    default: hipLaunchKernelGGL(hh_ring_decode_kernel<f16_t>, grid, block, 0, st, a); break;
  }
  CC_LAUNCH_CHECK();
  return CC_OK;
}

int cc_hh_ring_next_key_init(const cc_kv_view* c, const int32_t* input_pos, const int32_t* denom, int32_t W, const float* wsum,
                             int32_t g, int32_t w, uint64_t* next_key, cc_stream_t stream) {
  CC_ENTRY();
  if (!cc_view_ok(c) || !input_pos || !denom || W <= 1 || !wsum || !next_key || c->Hp != c->H) return CC_ERR_BAD_ARG;
  RingArgs a{};
  a.pos = c->pos; a.H = c->H; a.Hc = c->Hc; a.S = c->S; a.D = c->D; a.W = W; a.g = g; a.w = w;
  a.input_pos = input_pos; a.denom = const_cast<int32_t*>(denom); a.wsum = const_cast<float*>(wsum);
  a.key_out = reinterpret_cast<unsigned long long*>(next_key);
  a.nk = cc_next_key_slots(c->S);
  dim3 grid(c->H), block(kHybThreads);
  hipStream_t st = (hipStream_t)stream;
  switch (c->dtype) {
    case CC_DT_F32: hipLaunchKernelGGL(hh_ring_decode_kernel<float>, grid, block, 0, st, a); break;
    case CC_DT_BF16: hipLaunchKernelGGL(hh_ring_decode_kernel<bf16_t>, grid, block, 0, st, a); break;
    default: hipLaunchKernelGGL(hh_ring_decode_kernel<f16_t>, grid, block, 0, st, a); break;
  }
  CC_LAUNCH_CHECK();
  return CC_OK;
}

size_t cc_hh_ring_acc_words(int32_t H, int32_t S, int32_t W, int32_t dtype) {
  const size_t hs = (size_t)H * (size_t)S, es = dtype == CC_DT_F32 ? 4 : 2;
  // accumulators, two ticket words, column-major shadow of the ring, [H] keys + [H] tickets of the split decision kernel
  return hs * 4 + 2 + (hs * (size_t)W * es + 7) / 8 + 2 * (size_t)H;
}

int cc_hh_ring_window_sums(const void* num, int32_t H, int32_t S, int32_t W, int32_t dtype, float* wsum, uint64_t* wsum_acc,
                           cc_stream_t stream) {
  CC_ENTRY();
  if (!num || !wsum || H <= 0 || S <= 0 || W <= 0 || !cc_dt_ok(dtype)) return CC_ERR_BAD_ARG;
  hipStream_t st = (hipStream_t)stream;
  launch_window_sums(num, nullptr, nullptr, H, S, W, dtype, wsum, reinterpret_cast<u64*>(wsum_acc), st);
  CC_LAUNCH_CHECK();
  return CC_OK;
}

int cc_hh_ring_update(void* num, int32_t* denom, int64_t* counter, const void* attn, int32_t H, int32_t S, int32_t T,
                      int32_t W, int32_t dtype, uint64_t* wsum_acc, float* wsum, cc_stream_t stream) {
  CC_ENTRY();
  if (!num || !denom || !counter || !attn || H <= 0 || S <= 0 || T < 0 || T > S || W <= 0 || !cc_dt_ok(dtype) ||
      ((wsum_acc == nullptr) != (wsum == nullptr)))
    return CC_ERR_BAD_ARG;
  const int n = H * S;
  int nb = (n + 255) / 256;
  if (nb > 2048) nb = 2048;
  hipStream_t st = (hipStream_t)stream;
  dim3 grid(nb), block(256);
  if (wsum_acc) {
    // few, large workgroups: one same-address ticket atomic per workgroup (2048 of them serialise to 26 us)
    u64* wa = reinterpret_cast<u64*>(wsum_acc);
    block = dim3(1024);
    grid = dim3((unsigned)((n + 1023) / 1024 < 192 ? (n + 1023) / 1024 : 192));
    switch (dtype) {
      case CC_DT_F32: hipLaunchKernelGGL(hh_ring_update_tracked_kernel<float>, grid, block, 0, st, (float*)num, denom, counter, (const float*)attn, H, S, T, W, wa, wsum); break;
      case CC_DT_BF16: hipLaunchKernelGGL(hh_ring_update_tracked_kernel<bf16_t>, grid, block, 0, st, (bf16_t*)num, denom, counter, (const bf16_t*)attn, H, S, T, W, wa, wsum); break;
      default: hipLaunchKernelGGL(hh_ring_update_tracked_kernel<f16_t>, grid, block, 0, st, (f16_t*)num, denom, counter, (const f16_t*)attn, H, S, T, W, wa, wsum); break;
    }
    CC_LAUNCH_CHECK();
    return CC_OK;
  }
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
