// cc_evict.hip — decode-time update_kv for every eviction policy: per-head importance scan, arg-min with
// torch's tie rules, in-place KV insert; plus the heavy-hitter history update, the prefill fill and row
// L2 norms.  gfx950 (wave64).  HBM-bound integer/byte work: coalesced scans, wave shuffles + one LDS hop
// for the cross-wave arg-min; no MFMA.
//
// One workgroup of 1024 threads (16 waves) per "pos head" (Hp = H for head-specific policies, 1 otherwise):
// each thread scans S/1024 slots with all of its loads issued before the first use, reduces a 64-bit
// (orderable score, slot) key with __shfl_xor, one LDS hop across waves, then the same workgroup inserts
// the new token so that select+insert is a single launch (cache.py does it in ~10).
#include "cc_common.h"

namespace {

enum Policy { P_FULL = 0, P_RECENT_GLOBAL = 1, P_SCORES = 2, P_RANDOM = 3, P_L2 = 4, P_HH = 5 };

struct UpdArgs {
  void* k_cache;
  void* v_cache;
  int32_t* pos;
  uint8_t* mask;
  int32_t* cache_cts;
  int H, Hp, Hc, S, D;
  const void* k_new;
  const void* v_new;
  const int32_t* input_pos;
  int64_t* idx_out;
  int g, w;
  const void* scores;    // P_SCORES: [Hs,S] score dtype ; P_RANDOM: [S] f32
  int score_heads;       // rows in `scores`
  void* key_norm;        // P_L2: [H,S] T
  const float* gmax_in;  // P_L2: the norm maximum from l2_norm_max_kernel, or null = recompute per workgroup (select-only calls)
  double* num;           // P_HH
  int32_t* denom;        // P_HH
  unsigned long long* key_out;  // pipeline seed: [H][nk] partial arg-min keys, one row per kv head (entry 0 = the key, rest = ~0); no side effects
  int nk;
  unsigned long long rng_seed;  // P_RANDOM with scores == null: cc_rng_uniform(rng_seed, *input_pos, slot)
};

constexpr int kUpdThreads = 1024;

// The new token's K/V words this workgroup will write: word i of [n_heads_here][2][words] (K row then V row).
// `pre` = word threadIdx.x, loaded BEFORE the scan so its latency hides behind it.
template <typename T>
__device__ __forceinline__ uint32_t new_word(const UpdArgs& a, int h0, int i) {
  const int words = a.D * (int)sizeof(T) / 4;
  const int h = h0 + i / (2 * words), j = i % (2 * words);
  const uint32_t* src = reinterpret_cast<const uint32_t*>(j < words ? a.k_new : a.v_new) + (size_t)h * words;
  return src[j < words ? j : j - words];
}

// Insert the new token for kv heads [h0, h0+nh) at slot idx (all threads of the block participate).
template <typename T>
__device__ __forceinline__ void insert_rows(const UpdArgs& a, int h0, int nh, int idx, uint32_t pre) {
  const int words = a.D * (int)sizeof(T) / 4;
  const int total = nh * 2 * words;
  for (int i = threadIdx.x; i < total; i += blockDim.x) {
    const uint32_t val = (i == (int)threadIdx.x) ? pre : new_word<T>(a, h0, i);
    const int h = h0 + i / (2 * words), j = i % (2 * words);
    uint32_t* dst = reinterpret_cast<uint32_t*>(j < words ? a.k_cache : a.v_cache) + ((size_t)h * a.S + idx) * words;
    dst[j < words ? j : j - words] = val;
  }
  if ((int)threadIdx.x < nh) a.mask[(size_t)(h0 + threadIdx.x) * a.S + idx] = 1;
}

// ref: cache.py:602 `self.key_norm.max()` over ALL heads and slots, by one whole workgroup of kUpdThreads threads (the
// norms are H*S*2 bytes: L2-resident); NaN propagates like torch.max.  `sm_f` needs kUpdThreads / 64 + 1 floats.
template <typename T>
__device__ __forceinline__ float block_norm_max(const void* key_norm, int n, float* sm_f) {
  const T* kn = reinterpret_cast<const T*>(key_norm);
  float m = -INFINITY;
  int nan = 0;
  constexpr int VEC = 16 / (int)sizeof(T);
  const int nvec = ((reinterpret_cast<uintptr_t>(kn) & 15) == 0) ? n / VEC : 0;
  for (int i0 = threadIdx.x; i0 < nvec; i0 += kUpdThreads * 4) {  // 4 x 16-byte loads in flight per thread
    Vec16<T> vv[4];
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const int i = i0 + u * kUpdThreads;
      vv[u].load(kn + (size_t)(i < nvec ? i : nvec - 1) * VEC);
    }
#pragma unroll
    for (int u = 0; u < 4; u++) {
      float f[VEC];
      vv[u].unpack(f);
#pragma unroll
      for (int e = 0; e < VEC; e++) {
        nan |= (f[e] != f[e]);
        m = fmaxf(m, f[e]);
      }
    }
  }
  for (int i = nvec * VEC + threadIdx.x; i < n; i += blockDim.x) {
    float v = ElemTraits<T>::load(kn, i);
    nan |= (v != v);
    m = fmaxf(m, v);
  }
  m = wave_max_f32(m);
  nan = __any(nan);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) sm_f[wave] = nan ? NAN : m;
  __syncthreads();
  if (threadIdx.x == 0) {
    float mm = -INFINITY;
    int nn = 0;
    for (int i = 0; i < kUpdThreads / 64; i++) {
      float v = sm_f[i];
      nn |= (v != v);
      mm = fmaxf(mm, v);
    }
    sm_f[kUpdThreads / 64] = nn ? NAN : mm;
  }
  __syncthreads();
  return sm_f[kUpdThreads / 64];
}

// The maximum taken BEFORE any head inserts (cache.py:602 precedes the scatter of :592-593): with more than one head,
// a workgroup of the update launch that started late could otherwise see another head's freshly inserted norm in
// place of the evicted maximum.
template <typename T>
__global__ __launch_bounds__(kUpdThreads) void l2_norm_max_kernel(const void* key_norm, int n, float* out) {
  __shared__ float sm_f[kUpdThreads / 64 + 2];
  const float m = block_norm_max<T>(key_norm, n, sm_f);
  if (threadIdx.x == 0) *out = m;
}

// The l2 step's carried norm record (cc_common.h, cc_l2_record) of every kv head from the norms and the key row as they stand — what
// the seed of the pipeline and the two-launch step leave for the next single-launch step: the head's largest norm over the slots it
// KEEPS at the next position, i.e. all but the arg-min of the keys just scored (the row's live entries).  One workgroup per kv head;
// 16-bit norms, read as patterns (norms are >= +0 or NaN: unsigned order == numeric order, NaN on top — torch.max's propagation).
__global__ __launch_bounds__(kUpdThreads) void l2_record_kernel(const uint16_t* key_norm, int S, const int32_t* input_pos, int delta,
                                                                unsigned long long* next_key, int nk) {
  __shared__ unsigned long long sm_key[kUpdThreads / 64 + 2];
  const int h = blockIdx.x;
  const unsigned long long* row = next_key + (size_t)h * nk;
  unsigned long long kb = ~0ull;
  for (int i = threadIdx.x; i < nk - kNextKeyTail; i += kUpdThreads) kb = row[i] < kb ? row[i] : kb;
  const unsigned long long kmin = block_min_u64(kb, sm_key);
  const int e_next = (kmin == ~0ull) ? -1 : (int)((kmin & 0xffffffffull) >> 1);
  __syncthreads();  // (sm_key is reused)
  const uint16_t* kn = key_norm + (size_t)h * S;
  unsigned long long k1 = 0;  // (pattern << 32 | slot) of this thread's largest norm among the kept slots
  unsigned long long ka = 0;  // ... among all slots (kept for inspection: the record's upper fields)
  for (int s = threadIdx.x; s < S; s += kUpdThreads) {
    const unsigned long long key = ((unsigned long long)kn[s] << 32) | (unsigned)s;
    ka = key > ka ? key : ka;
    if (s != e_next) k1 = key > k1 ? key : k1;
  }
  const unsigned long long K1 = ~block_min_u64(~k1, sm_key);
  __syncthreads();
  const unsigned long long KA = ~block_min_u64(~ka, sm_key);
  if (threadIdx.x == 0)
    next_key[(size_t)h * nk + (nk - kNextKeyTail) + ((*input_pos + delta) & 1)] = cc_l2_record((unsigned)(K1 >> 32), (unsigned)(KA >> 32), (unsigned)(KA & 0xffffffffull));
}

template <int POLICY, typename T, typename ST>
__global__ __launch_bounds__(kUpdThreads) void decode_update_kernel(UpdArgs a) {
  __shared__ unsigned long long sm_key[kUpdThreads / 64 + 2];
  __shared__ float sm_f[kUpdThreads / 64 + 2];
  const int hp = blockIdx.x;
  const int S = a.S;
  const int32_t* pos = a.pos + (size_t)hp * S;
  const int32_t p = *a.input_pos;
  // heads this workgroup inserts for: all of them when pos is shared (head-constant policies), else its own
  const int h0 = a.Hp == 1 ? 0 : hp, nh = a.Hp == 1 ? a.H : 1;
  uint32_t pre = 0;
  if (a.k_new != nullptr && (int)threadIdx.x < nh * 2 * (a.D * (int)sizeof(T) / 4)) pre = new_word<T>(a, h0, threadIdx.x);

  float gmax = 0.f;
  if (POLICY == P_L2) gmax = a.gmax_in ? *a.gmax_in : block_norm_max<T>(a.key_norm, a.H * S, sm_f);

  unsigned long long best = ~0ull;
  constexpr int UN = 4;  // slots per thread per pass: all per-slot loads are issued before the first use
  const size_t hoff = (size_t)hp * S;
  for (int s0 = threadIdx.x; s0 < S; s0 += kUpdThreads * UN) {
    int32_t psv[UN];
    float fv[UN];
    double dv[UN];
    int32_t iv[UN];
#pragma unroll
    for (int u = 0; u < UN; u++) {
      const int s = s0 + u * kUpdThreads;
      const int sc = s < S ? s : S - 1;  // clamped: loads stay unconditional (no exec-masked waits)
      psv[u] = pos[sc];
      fv[u] = 0.f;
      dv[u] = 0.0;
      iv[u] = 0;
      if (POLICY == P_SCORES) fv[u] = ElemTraits<ST>::load(reinterpret_cast<const ST*>(a.scores), (a.score_heads == 1 ? 0 : hoff) + sc);
      if (POLICY == P_RANDOM && a.scores) fv[u] = reinterpret_cast<const float*>(a.scores)[sc];
      if (POLICY == P_L2) fv[u] = ElemTraits<T>::load(reinterpret_cast<const T*>(a.key_norm), hoff + sc);
      if (POLICY == P_HH) {
        dv[u] = a.num[hoff + sc];
        iv[u] = a.denom[hoff + sc];
      }
    }
#pragma unroll
    for (int u = 0; u < UN; u++) {
      const int s = s0 + u * kUpdThreads;
      const int32_t ps = psv[u];
      uint32_t ord;
      bool skip = s >= S;
      if (POLICY == P_FULL) {
        ord = orderable_i32(ps);  // ref: cache.py:502 pos.argmin()
      } else if (POLICY == P_RECENT_GLOBAL) {
        skip |= s < a.g;  // ref: cache.py:554 argmin(pos[:, :, g:]) + g
        ord = orderable_i32(ps);
      } else {
        float sc;
        if (POLICY == P_SCORES) {
          sc = fv[u];
        } else if (POLICY == P_RANDOM) {
          sc = (ps >= p - a.w) ? INFINITY : (a.scores ? fv[u] : cc_rng_uniform(a.rng_seed, p, s < S ? s : S - 1));  // ref: cache.py:523
        } else if (POLICY == P_L2) {
          // ref: cache.py:601-605 — model-dtype subtraction (fp32 op, rounded to T), recent window -> +inf
          sc = ElemTraits<T>::rnd(gmax - fv[u]);
          if (ps >= p - a.w) sc = INFINITY;
        } else {  // P_HH, ref: cache.py:727-749
          const int32_t dn = iv[u] < 1 ? 1 : iv[u];
          sc = __fdiv_rn((float)dv[u], (float)dn);  // f64->f32 RNE, int32->f32, IEEE divide
          if (ps < a.g || ps >= p - a.w) sc = 1.0f;
          if (ps == -1) sc = 0.0f;
        }
        if (POLICY != P_HH) {
          // ref: cache.py:373-376 base rules: first g SLOTS -> +inf, then empty slots -> -inf
          if (s < a.g) sc = INFINITY;
          if (ps == -1) sc = -INFINITY;
        }
        ord = orderable_f32(sc);
      }
      // low word = slot << 1 | (slot was empty): ordering by slot is preserved, and the winner's "was empty"
      // bit travels with the key so no dependent re-read of pos[idx] is needed after the reduction
      const unsigned long long key = skip ? ~0ull : make_key(ord, ((uint32_t)s << 1) | (uint32_t)(ps == -1));
      best = key < best ? key : best;
    }
  }
  best = block_min_u64(best, sm_key);
  const int idx = (int)((best & 0xffffffffull) >> 1);
  const int ins = (int)(best & 1ull);  // ref: cache.py:356-360 num_insertions = (old pos == -1)
  if (a.key_out != nullptr) {  // seed of the fused decode-step pipeline: publish the key, touch nothing else
    // (head-constant policies — one workgroup — fill every kv head's copy of the row: each head reads and rewrites its own,
    //  cc_attn_decode.hip KEY ROWS)
    for (int hh = h0; hh < h0 + nh; hh++)
      for (int i = threadIdx.x; i < a.nk - kNextKeyTail; i += blockDim.x) a.key_out[(size_t)hh * a.nk + i] = (i == 0) ? best : ~0ull;  // (the live entries)
    return;
  }
  if (threadIdx.x == 0) a.idx_out[hp] = idx;

  if (POLICY == P_HH && threadIdx.x == 0) {  // ref: cache.py:754-763 (part of _eviction_idx itself)
    a.num[(size_t)hp * S + idx] = 0.0;
    a.denom[(size_t)hp * S + idx] = 0;
  }
  if (a.k_new == nullptr) return;  // select only

  // ---- insert (ref: cache.py:356-362, 390-401, 460-490, 330)
  if (threadIdx.x == 0) {
    a.pos[(size_t)hp * S + idx] = p;
    if (a.Hp == 1) {
      for (int j = 0; j < a.Hc; j++) a.cache_cts[j] += ins;
    } else if (a.Hc == a.Hp) {
      a.cache_cts[hp] += ins;
    } else if (hp == 0) {
      a.cache_cts[0] += ins;  // num_insertions[:1]
    }
  }
  insert_rows<T>(a, h0, nh, idx, pre);
  if (POLICY == P_L2 && threadIdx.x < 16) {  // ref: cache.py:592-593
    const float ss = sumsq_canonical_16<T>(reinterpret_cast<const T*>(a.k_new) + (size_t)hp * a.D, a.D, threadIdx.x);
    if (threadIdx.x == 0) ElemTraits<T>::store(reinterpret_cast<T*>(a.key_norm), (size_t)hp * S + idx, cc_sqrt_rn(ss));
  }
}

template <int POLICY, typename ST = float>
int launch_update(const cc_kv_view* c, UpdArgs& a, hipStream_t st) {
  a.k_cache = c->k_cache;
  a.v_cache = c->v_cache;
  a.pos = c->pos;
  a.mask = c->mask;
  a.cache_cts = c->cache_cts;
  a.H = c->H;
  a.Hp = c->Hp;
  a.Hc = c->Hc;
  a.S = c->S;
  a.D = c->D;
  dim3 grid(c->Hp), block(kUpdThreads);
  switch (c->dtype) {
    case CC_DT_F32: hipLaunchKernelGGL((decode_update_kernel<POLICY, float, ST>), grid, block, 0, st, a); break;
    case CC_DT_BF16: hipLaunchKernelGGL((decode_update_kernel<POLICY, bf16_t, ST>), grid, block, 0, st, a); break;
    default: hipLaunchKernelGGL((decode_update_kernel<POLICY, f16_t, ST>), grid, block, 0, st, a); break;
  }
  CC_LAUNCH_CHECK();
  return CC_OK;
}

// ---------------------------------------------------------------- heavy-hitter history update
template <typename T>
__global__ __launch_bounds__(256) void hh_update_kernel(double* num, int32_t* denom, int64_t* counter, const T* attn,
                                                        int H, int S, int Tn) {
  const int n = H * S;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const int h = i / S, s = i - h * S;
    if (s < Tn) num[i] += (double)ElemTraits<T>::load(attn, (size_t)h * Tn + s);  // exact widening add
    denom[i] += 1;
  }
  if (counter && blockIdx.x == 0 && threadIdx.x == 0) *counter += 1;
}

// ---------------------------------------------------------------- prefill fill
template <typename T>
__global__ __launch_bounds__(256) void prefill_fill_kernel(UpdArgs a, const int64_t* pos_val, int PH, int Tn) {
  // grid.x over 16-byte chunks of [H, Tn, D]; rows keep their layout, only the slot stride changes (Tn -> S)
  const int row_words = a.D * (int)sizeof(T) / 4;
  const size_t total = (size_t)a.H * Tn * row_words;
  const uint32_t* ks = reinterpret_cast<const uint32_t*>(a.k_new);
  const uint32_t* vs = reinterpret_cast<const uint32_t*>(a.v_new);
  uint32_t* kd = reinterpret_cast<uint32_t*>(a.k_cache);
  uint32_t* vd = reinterpret_cast<uint32_t*>(a.v_cache);
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const size_t row = i / row_words;
    const int wd = (int)(i - row * row_words);
    const int h = (int)(row / Tn), t = (int)(row - (size_t)h * Tn);
    const size_t dst = ((size_t)h * a.S + t) * row_words + wd;
    kd[dst] = ks[i];
    vd[dst] = vs[i];
  }
  const int np = a.Hp * Tn;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < np; i += gridDim.x * blockDim.x) {
    const int hp = i / Tn, t = i - hp * Tn;
    a.pos[(size_t)hp * a.S + t] = (int32_t)pos_val[(size_t)(PH == 1 ? 0 : hp) * Tn + t];
  }
  const int nm = a.H * Tn;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < nm; i += gridDim.x * blockDim.x) {
    const int h = i / Tn, t = i - h * Tn;
    a.mask[(size_t)h * a.S + t] = 1;
  }
  if (blockIdx.x == 0 && threadIdx.x < a.Hc) a.cache_cts[threadIdx.x] += Tn;
}

// ---------------------------------------------------------------- row L2 norms
template <typename T>
__global__ __launch_bounds__(256) void row_l2_norm_kernel(const T* x, int rows, int D, int negate, T* out) {
  const int lane16 = threadIdx.x & 15;
  const int group = (blockIdx.x * blockDim.x + threadIdx.x) >> 4;
  const int ngroups = (gridDim.x * blockDim.x) >> 4;
  for (int r = group; r < rows; r += ngroups) {
    const float ss = sumsq_canonical_16<T>(x + (size_t)r * D, D, lane16);
    if (lane16 == 0) {
      const float n = cc_sqrt_rn(ss);
      ElemTraits<T>::store(out, r, negate ? -n : n);
    }
  }
}

}  // namespace

extern "C" int32_t cc_decode_step_l2_carry(void);  // cc_attn_decode.hip: CC_V_L2CARRY of this build
// (shared with cc_attn_decode.hip: the two-launch l2 step leaves the record of ITS position the same way; declared in cc_common.h)
int cc_l2_record_launch(const void* key_norm, int H, int S, int dtype, const int32_t* input_pos, int delta, unsigned long long* next_key,
                        hipStream_t st) {
  if (cc_dt_size(dtype) != 2) return CC_OK;  // (the single-launch l2 step serves 16-bit caches only: nobody reads a record of another)
  hipLaunchKernelGGL(l2_record_kernel, dim3(H), dim3(kUpdThreads), 0, st, reinterpret_cast<const uint16_t*>(key_norm), S, input_pos, delta,
                     next_key, cc_next_key_slots(S));
  CC_LAUNCH_CHECK();
  return CC_OK;
}

extern "C" {

int cc_decode_update_full(const cc_kv_view* c, const void* k_new, const void* v_new, const int32_t* input_pos,
                          int64_t* idx_out, cc_stream_t stream) {
  CC_ENTRY();
  if (!cc_view_ok(c) || !input_pos || !idx_out || c->Hp != 1 || (k_new && !v_new)) return CC_ERR_BAD_ARG;
  UpdArgs a{};
  a.k_new = k_new; a.v_new = v_new; a.input_pos = input_pos; a.idx_out = idx_out;
  return launch_update<P_FULL>(c, a, (hipStream_t)stream);
}

int cc_decode_update_recent_global(const cc_kv_view* c, const void* k_new, const void* v_new,
                                   const int32_t* input_pos, int32_t g, int64_t* idx_out, cc_stream_t stream) {
  CC_ENTRY();
  if (!cc_view_ok(c) || !input_pos || !idx_out || c->Hp != 1 || g < 0 || g >= c->S || (k_new && !v_new))
    return CC_ERR_BAD_ARG;
  UpdArgs a{};
  a.k_new = k_new; a.v_new = v_new; a.input_pos = input_pos; a.idx_out = idx_out; a.g = g;
  return launch_update<P_RECENT_GLOBAL>(c, a, (hipStream_t)stream);
}

int cc_decode_update_scores(const cc_kv_view* c, const void* k_new, const void* v_new, const int32_t* input_pos,
                            const void* scores, int32_t score_dtype, int32_t g, int64_t* idx_out,
                            cc_stream_t stream) {
  CC_ENTRY();
  if (!cc_view_ok(c) || !input_pos || !idx_out || !scores || !cc_dt_ok(score_dtype) || g < 0 || (k_new && !v_new))
    return CC_ERR_BAD_ARG;
  UpdArgs a{};
  a.k_new = k_new; a.v_new = v_new; a.input_pos = input_pos; a.idx_out = idx_out; a.g = g;
  a.scores = scores; a.score_heads = c->Hp;
  switch (score_dtype) {
    case CC_DT_F32: return launch_update<P_SCORES, float>(c, a, (hipStream_t)stream);
    case CC_DT_BF16: return launch_update<P_SCORES, bf16_t>(c, a, (hipStream_t)stream);
    default: return launch_update<P_SCORES, f16_t>(c, a, (hipStream_t)stream);
  }
}

int cc_decode_update_random(const cc_kv_view* c, const void* k_new, const void* v_new, const int32_t* input_pos,
                            const float* rand_u, int32_t g, int32_t w, int64_t* idx_out, cc_stream_t stream) {
  CC_ENTRY();
  if (!cc_view_ok(c) || !input_pos || !idx_out || !rand_u || c->Hp != 1 || g < 0 || (k_new && !v_new))
    return CC_ERR_BAD_ARG;
  UpdArgs a{};
  a.k_new = k_new; a.v_new = v_new; a.input_pos = input_pos; a.idx_out = idx_out; a.g = g; a.w = w;
  a.scores = rand_u; a.score_heads = 1;
  return launch_update<P_RANDOM>(c, a, (hipStream_t)stream);
}

size_t cc_decode_update_l2_workspace_bytes(int32_t H, int32_t S) {
  (void)S;
  return H > 1 ? 256 : 0;  // the norm maximum, taken by a pre-launch before any head inserts
}

int cc_decode_update_l2(const cc_kv_view* c, const void* k_new, const void* v_new, const int32_t* input_pos,
                        void* key_norm, int32_t g, int32_t w, int64_t* idx_out, void* workspace,
                        size_t workspace_bytes, cc_stream_t stream) {
  CC_ENTRY();
  if (!cc_view_ok(c) || !input_pos || !idx_out || !key_norm || c->Hp != c->H || g < 0 || (k_new && !v_new))
    return CC_ERR_BAD_ARG;
  UpdArgs a{};
  a.k_new = k_new; a.v_new = v_new; a.input_pos = input_pos; a.idx_out = idx_out; a.g = g; a.w = w;
  a.key_norm = key_norm;
  if (k_new && c->H > 1) {  // inserts of other heads must not reach this head's maximum: take it in a launch of its own
    if (!workspace || workspace_bytes < cc_decode_update_l2_workspace_bytes(c->H, c->S)) return CC_ERR_WORKSPACE;
    float* gm = reinterpret_cast<float*>(workspace);
    const int n = c->H * c->S;
    hipStream_t st = (hipStream_t)stream;
    switch (c->dtype) {
      case CC_DT_F32: hipLaunchKernelGGL(l2_norm_max_kernel<float>, dim3(1), dim3(kUpdThreads), 0, st, key_norm, n, gm); break;
      case CC_DT_BF16: hipLaunchKernelGGL(l2_norm_max_kernel<bf16_t>, dim3(1), dim3(kUpdThreads), 0, st, key_norm, n, gm); break;
      default: hipLaunchKernelGGL(l2_norm_max_kernel<f16_t>, dim3(1), dim3(kUpdThreads), 0, st, key_norm, n, gm); break;
    }
    CC_LAUNCH_CHECK();
    a.gmax_in = gm;
  }
  return launch_update<P_L2>(c, a, (hipStream_t)stream);
}

int cc_l2_next_key_init(const cc_kv_view* c, const int32_t* input_pos, void* key_norm, int32_t g, int32_t w, uint64_t* next_key,
                        cc_stream_t stream) {
  CC_ENTRY();
  if (!cc_view_ok(c) || !input_pos || !key_norm || !next_key || c->Hp != c->H || g < 0) return CC_ERR_BAD_ARG;
  UpdArgs a{};
  a.input_pos = input_pos; a.g = g; a.w = w;
  a.key_norm = key_norm;
  a.key_out = reinterpret_cast<unsigned long long*>(next_key);
  a.nk = cc_next_key_slots(c->S);
  const int rc = launch_update<P_L2>(c, a, (hipStream_t)stream);
  if (rc != CC_OK) return rc;
  // the norm record the first step (position *input_pos) takes its head's maximum from: the state behind position *input_pos - 1
  // (only the r6 A/B build's single-launch step reads it: cc_decode_step_l2_carry())
  if (!cc_decode_step_l2_carry()) return CC_OK;
  return cc_l2_record_launch(key_norm, c->H, c->S, c->dtype, input_pos, -1, reinterpret_cast<unsigned long long*>(next_key), (hipStream_t)stream);
}

int cc_decode_update_heavy_hitter(const cc_kv_view* c, const void* k_new, const void* v_new,
                                  const int32_t* input_pos, double* num, int32_t* denom, int32_t g, int32_t w,
                                  int64_t* idx_out, cc_stream_t stream) {
  CC_ENTRY();
  if (!cc_view_ok(c) || !input_pos || !idx_out || !num || !denom || c->Hp != c->H || (k_new && !v_new))
    return CC_ERR_BAD_ARG;
  UpdArgs a{};
  a.k_new = k_new; a.v_new = v_new; a.input_pos = input_pos; a.idx_out = idx_out; a.g = g; a.w = w;
  a.num = num; a.denom = denom;
  return launch_update<P_HH>(c, a, (hipStream_t)stream);
}

int32_t cc_hh_next_key_slots(int32_t S) { return S > 0 ? cc_next_key_slots(S) : 0; }

int cc_hh_next_key_init(const cc_kv_view* c, const int32_t* input_pos, const double* num, const int32_t* denom,
                        int32_t g, int32_t w, uint64_t* next_key, cc_stream_t stream) {
  CC_ENTRY();
  if (!cc_view_ok(c) || !input_pos || !num || !denom || !next_key || c->Hp != c->H) return CC_ERR_BAD_ARG;
  UpdArgs a{};
  a.input_pos = input_pos; a.g = g; a.w = w;
  a.num = const_cast<double*>(num); a.denom = const_cast<int32_t*>(denom);
  a.key_out = reinterpret_cast<unsigned long long*>(next_key);
  a.nk = cc_next_key_slots(c->S);
  return launch_update<P_HH>(c, a, (hipStream_t)stream);
}

int cc_rg_next_key_init(const cc_kv_view* c, const int32_t* input_pos, int32_t g, uint64_t* next_key, cc_stream_t stream) {
  CC_ENTRY();
  if (!cc_view_ok(c) || !input_pos || !next_key || c->Hp != 1 || g < 0 || g >= c->S) return CC_ERR_BAD_ARG;
  UpdArgs a{};
  a.input_pos = input_pos; a.g = g;
  a.key_out = reinterpret_cast<unsigned long long*>(next_key);
  a.nk = cc_next_key_slots(c->S);
  return launch_update<P_RECENT_GLOBAL>(c, a, (hipStream_t)stream);
}

int cc_random_next_key_init(const cc_kv_view* c, const int32_t* input_pos, const float* rand_u, int32_t g, int32_t w,
                            uint64_t* next_key, cc_stream_t stream) {
  CC_ENTRY();
  if (!cc_view_ok(c) || !input_pos || !rand_u || !next_key || c->Hp != 1 || g < 0) return CC_ERR_BAD_ARG;
  UpdArgs a{};
  a.input_pos = input_pos; a.g = g; a.w = w;
  a.scores = rand_u; a.score_heads = 1;
  a.key_out = reinterpret_cast<unsigned long long*>(next_key);
  a.nk = cc_next_key_slots(c->S);
  return launch_update<P_RANDOM>(c, a, (hipStream_t)stream);
}

int cc_random_next_key_init_rng(const cc_kv_view* c, const int32_t* input_pos, uint64_t seed, int32_t g, int32_t w,
                                uint64_t* next_key, cc_stream_t stream) {
  CC_ENTRY();
  if (!cc_view_ok(c) || !input_pos || !next_key || c->Hp != 1 || g < 0) return CC_ERR_BAD_ARG;
  UpdArgs a{};
  a.input_pos = input_pos; a.g = g; a.w = w;
  a.scores = nullptr; a.score_heads = 1; a.rng_seed = seed;
  a.key_out = reinterpret_cast<unsigned long long*>(next_key);
  a.nk = cc_next_key_slots(c->S);
  return launch_update<P_RANDOM>(c, a, (hipStream_t)stream);
}

int cc_hh_update(double* num, int32_t* denom, int64_t* counter, const void* attn, int32_t H, int32_t S, int32_t T,
                 int32_t dtype, cc_stream_t stream) {
  CC_ENTRY();
  if (!num || !denom || !attn || H <= 0 || S <= 0 || T < 0 || T > S || !cc_dt_ok(dtype)) return CC_ERR_BAD_ARG;
  const int n = H * S;
  dim3 grid((n + 255) / 256 < 1024 ? (n + 255) / 256 : 1024), block(256);
  hipStream_t st = (hipStream_t)stream;
  switch (dtype) {
    case CC_DT_F32: hipLaunchKernelGGL(hh_update_kernel<float>, grid, block, 0, st, num, denom, counter, (const float*)attn, H, S, T); break;
    case CC_DT_BF16: hipLaunchKernelGGL(hh_update_kernel<bf16_t>, grid, block, 0, st, num, denom, counter, (const bf16_t*)attn, H, S, T); break;
    default: hipLaunchKernelGGL(hh_update_kernel<f16_t>, grid, block, 0, st, num, denom, counter, (const f16_t*)attn, H, S, T); break;
  }
  CC_LAUNCH_CHECK();
  return CC_OK;
}

int cc_prefill_fill(const cc_kv_view* c, const void* k_val, const void* v_val, const int64_t* pos_val, int32_t PH,
                    int32_t T, cc_stream_t stream) {
  CC_ENTRY();
  if (!cc_view_ok(c) || !k_val || !v_val || !pos_val || T <= 0 || T > c->S || (PH != 1 && PH != c->Hp) || c->Hc > 256)
    return CC_ERR_BAD_ARG;
  UpdArgs a{};
  a.k_cache = c->k_cache; a.v_cache = c->v_cache; a.pos = c->pos; a.mask = c->mask; a.cache_cts = c->cache_cts;
  a.H = c->H; a.Hp = c->Hp; a.Hc = c->Hc; a.S = c->S; a.D = c->D;
  a.k_new = k_val; a.v_new = v_val;
  const size_t words = (size_t)c->H * T * c->D * cc_dt_size(c->dtype) / 4;
  size_t nb = (words + 255) / 256;
  if (nb > 4096) nb = 4096;
  if (nb < 1) nb = 1;
  dim3 grid((unsigned)nb), block(256);
  hipStream_t st = (hipStream_t)stream;
  switch (c->dtype) {
    case CC_DT_F32: hipLaunchKernelGGL(prefill_fill_kernel<float>, grid, block, 0, st, a, pos_val, PH, T); break;
    case CC_DT_BF16: hipLaunchKernelGGL(prefill_fill_kernel<bf16_t>, grid, block, 0, st, a, pos_val, PH, T); break;
    default: hipLaunchKernelGGL(prefill_fill_kernel<f16_t>, grid, block, 0, st, a, pos_val, PH, T); break;
  }
  CC_LAUNCH_CHECK();
  return CC_OK;
}

int cc_row_l2_norm(const void* x, int32_t H, int32_t N, int32_t D, int32_t dtype, int32_t negate, void* out,
                   cc_stream_t stream) {
  CC_ENTRY();
  if (!x || !out || H <= 0 || N <= 0 || D <= 0 || !cc_dt_ok(dtype)) return CC_ERR_BAD_ARG;
  const int rows = H * N;
  int nb = (rows + 15) / 16;
  if (nb > 2048) nb = 2048;
  dim3 grid(nb), block(256);
  hipStream_t st = (hipStream_t)stream;
  switch (dtype) {
    case CC_DT_F32: hipLaunchKernelGGL(row_l2_norm_kernel<float>, grid, block, 0, st, (const float*)x, rows, D, negate, (float*)out); break;
    case CC_DT_BF16: hipLaunchKernelGGL(row_l2_norm_kernel<bf16_t>, grid, block, 0, st, (const bf16_t*)x, rows, D, negate, (bf16_t*)out); break;
    default: hipLaunchKernelGGL(row_l2_norm_kernel<f16_t>, grid, block, 0, st, (const f16_t*)x, rows, D, negate, (f16_t*)out); break;
  }
  CC_LAUNCH_CHECK();
  return CC_OK;
}

}  // extern "C"
