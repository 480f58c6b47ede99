// cc_attn_decode.hip — decode attention over the pruned cache for gfx950, GQA-aware, split along the cache
// axis ("flash-decode"): K and V are read from HBM exactly once, with no repeat_interleave.
//
// ref: model.py:395-418 (GQA glue, group mean of the probabilities) + attention_utils.py:27-54.
//
// Kernel 1 (split): grid (n_split, H, R/RT) workgroups of NW waves.  A K/V row of D elements is 16-byte
//   chunks over LPR = D*sizeof(T)/16 lanes, so one wave-wide 16-byte load covers 64/LPR whole rows, fully
//   coalesced along [H, S, D].  Every lane issues its mask bytes, then ALL of its K and V loads for the
//   iteration (2*U x 16 B) before the first use, so the whole 2*H*S*D*sizeof(T) bytes of the layer are in
//   flight at once (the first version waited vmcnt(0) after every mask byte: 13.8 us -> see profiles/).
//   q.k partial dots are all-reduced across the LPR lanes with DPP moves (quad_perm / row_half_mirror /
//   row_mirror); scores leave through ONE coalesced store per iteration; softmax statistics are kept
//   online per wave; per-row-group partial outputs go through LDS (no cross-lane shuffles for the 32
//   accumulators), are merged across the workgroup, and one (m, l, O[RT][D]) partial per workgroup goes to
//   the workspace.
// Kernel 2 (combine): merges the partials into y, turns scores into probabilities with the final (M, L),
//   averages them over the R query heads of the group, and optionally applies the heavy-hitter history
//   update (cache.py:690-723) in the same pass.
// HBM-bound: ~1 flop/byte, no MFMA.
#include <atomic>
#include <cstdlib>
#include <mutex>

#include "cc_common.h"
#include "cc_wacc.h"

#include "cc_attn_decode_kernels.h"
#include "cc_attn_decode_qkv.h"

namespace {

struct CombineArgs {
  const void* scores;
  const float* part_ml;
  const float* part_o;
  void* y;
  void* attn_out;
  void* probs_out;
  double* hh_num;
  int32_t* hh_denom;
  int64_t* hh_counter;
  int S, R, D, n_split, chunk;
  // ---- fused decode step: score the NEXT step's eviction (cache.py:725-749 at position p + 1) in the same pass
  unsigned long long* next_key;  // [H][gridDim.x] or null: block c publishes the minimum over its slots
  const int32_t* input_pos;
  const int32_t* pos;  // [Hp, S]
  int H, g, w;
  int policy;  // next-eviction scoring: 1 = heavy hitter (cache.py:727-749), 2 = recent_global / full (cache.py:500-502, 552-556),
               // 3 = random (cache.py:519-524 over rand_next), 4 = l2, 5 = heavy hitter over the W > 1 history ring
  const float* rand_next;  // policy 3: [S] uniform draws for position p + 1, or null: cc_rng_uniform(rng_seed, p + 1, slot)
  unsigned long long rng_seed;
  // ---- l2 (policy 4, cache.py:597-605): score = dtype(max over all norms - norm); the maximum is folded from the
  //      per-wave partials of the streaming pass and the H freshly inserted norms
  const void* key_norm;   // [H, S] T
  const float* l2_pmax;   // [l2_np]
  const float* l2_new;    // [H]
  int l2_np;
  // ---- ring history (history_window_size W > 1, cache.py:716-723) folded into this pass, tracked window sums included
  void* ring_num;        // [H, S, W] T or null
  const int* ring_col;   // the step's ring column, published by the streaming pass of the same call
  int ring_W;
  u64* ring_acc;         // tracked state (include/coldcompress.h): accumulators, tickets, column-major shadow
  float* ring_wsum;      // [H, S]
  int Hp;
  int abl;  // measurement-only ablation bits (phases >> 8): 8 = no next-key epilogue, 16 = no y merge, 32 = no per-slot pass
  // ---- policy 6: KVCacheHybrid — score every head's eviction CANDIDATE for position p + 1 (cache.py:844-894 on the state
  //      this step leaves behind), commit the counts the streaming pass's inserts produced, bump num_punc (cache.py:1017)
  HybridStep hyb;
  int32_t* cache_cts;  // [H]
};

constexpr int kMaxR = 32;
constexpr int kMaxSplit = 512;

constexpr int kCombThreads = 128;  // == slots per block: every thread owns one cache slot of its kv head
constexpr int kPre = 8;            // query heads whose scores are prefetched into registers (R <= 8 typical)

template <typename T>
__global__ __launch_bounds__(kCombThreads) void decode_attn_combine_kernel(CombineArgs a) {
  __shared__ float sm_M[kMaxR], sm_L[kMaxR];
  extern __shared__ __attribute__((aligned(16))) float sm_wdyn[];  // [R][n_split]: exp(m_i - M)
  const int h = blockIdx.y, c = blockIdx.x, nchunks = gridDim.x;
  const int R = a.R, S = a.S, D = a.D, ns = a.n_split;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // a head's key row has kNextKeyPerChunk entries per block of this launch: this pass publishes the first, the others are kept
  // at ~0 — written HERE, at the start (their readers, the streaming pass, are done), so that these stores drain during the
  // launch instead of adding a store round trip behind its last instruction
  if (a.next_key && !(a.abl & 8) && threadIdx.x >= 1 && threadIdx.x < kNextKeyPerChunk)
    a.next_key[(size_t)h * (kNextKeyPerChunk * nchunks + kNextKeyTail) + threadIdx.x * nchunks + c] = ~0ull;
  // ---- issue this thread's per-slot loads first: their latency overlaps the (M, L) reduction below
  const T* sc = reinterpret_cast<const T*>(a.scores);
  const int s_mine = c * a.chunk + threadIdx.x;
  const bool have = threadIdx.x < a.chunk && s_mine < S;
  const int s_ld = have ? s_mine : 0;
  float xs[kPre];
#pragma unroll
  for (int r = 0; r < kPre; r++) xs[r] = ElemTraits<T>::load(sc, (size_t)(h * R + (r < R ? r : 0)) * S + s_ld);
  double num_old = 0.0;
  int32_t den_old = 0;
  if (a.hh_num) {
    num_old = a.hh_num[(size_t)h * S + s_ld];
    den_old = a.hh_denom[(size_t)h * S + s_ld];
  }
  int32_t ps_mine = 0, p_next = 0;
  if (a.next_key) {
    ps_mine = a.pos[(a.Hp == 1 ? 0 : (size_t)h * S) + s_ld];
    p_next = *a.input_pos + 1;
  }
  float rnd_mine = 0.f;
  if (a.next_key && a.policy == 3 && a.rand_next) rnd_mine = a.rand_next[s_ld];
  // hybrid (policy 6): the head's policy row, its count after this step's insert and this slot's protection masks are
  // requested here with everything else (a dependent chain strategies -> table at the tail cost ~3 us of L2 round trips)
  int hyb_flags = 0, hyb_win = 0, hyb_cts_n = 0;
  bool hyb_save = false;
  if (a.next_key && a.policy == 6) {
    const int pol = (int)a.hyb.strategies[h];
    hyb_cts_n = a.hyb.cts_next[h];
    const uint8_t spm = a.hyb.special_mask ? a.hyb.special_mask[(size_t)h * S + s_ld] : 0;
    const uint8_t pum = a.hyb.punc_mask ? a.hyb.punc_mask[(size_t)h * S + s_ld] : 0;
    hyb_flags = a.hyb.table[pol * 3];
    hyb_win = a.hyb.table[pol * 3 + 1];
    hyb_save = ((hyb_flags & HF_SPECIAL) && spm) || ((hyb_flags & HF_PUNC) && pum);
  }
  float kn_mine = 0.f, l2_part = -INFINITY;
  bool l2_nan = false;
  if (a.next_key && a.policy == 4) {
    kn_mine = ElemTraits<T>::load(reinterpret_cast<const T*>(a.key_norm), (size_t)h * S + s_ld);
    // [H, n_split, 4 waves] partial maxima: one float4 per streaming workgroup, the first four per thread in flight at once
    const float4* pm4 = reinterpret_cast<const float4*>(a.l2_pmax);
    const int n4 = a.l2_np >> 2;
    float4 pq[4];
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const int i = threadIdx.x + u * kCombThreads;
      pq[u] = i < n4 ? pm4[i] : make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
    }
    const float nw = threadIdx.x < a.H ? a.l2_new[threadIdx.x] : -INFINITY;
    auto fold = [&](float v) {
      l2_nan |= (v != v);
      l2_part = fmaxf(l2_part, v);
    };
#pragma unroll
    for (int u = 0; u < 4; u++) { fold(pq[u].x); fold(pq[u].y); fold(pq[u].z); fold(pq[u].w); }
    fold(nw);
    for (int i = threadIdx.x + 4 * kCombThreads; i < n4; i += kCombThreads) {
      const float4 q4 = pm4[i];
      fold(q4.x); fold(q4.y); fold(q4.z); fold(q4.w);
    }
    for (int i = kCombThreads + threadIdx.x; i < a.H; i += kCombThreads) fold(a.l2_new[i]);
  }
  // ring history (W > 1): the entry this step overwrites (from the column-major shadow) and the slot's exact accumulator
  int ring_col = 0;
  float ring_old = 0.f;
  WAcc racc{0, 0, 0, 0};
  T* ring_shadow = nullptr;
  if (a.ring_num) {
    const size_t hs = (size_t)gridDim.y * S, i = (size_t)h * S + s_ld;
    ring_col = *a.ring_col;
    ring_shadow = reinterpret_cast<T*>(a.ring_acc + hs * 4 + 2) + (size_t)ring_col * hs;
    ring_old = ElemTraits<T>::load(ring_shadow, i);
    const ulonglong2 a01 = *reinterpret_cast<const ulonglong2*>(a.ring_acc + i * 4);
    const ulonglong2 a23 = *reinterpret_cast<const ulonglong2*>(a.ring_acc + i * 4 + 2);
    racc = WAcc{a01.x, a01.y, a23.x, a23.y};
    den_old = a.hh_denom[i];
  }
  unsigned long long my_key = ~0ull;
  // y: the R*D outputs of this kv head are spread over the chunk blocks; inside a block the (output, split)
  // products are spread over ALL threads (G split-groups per output).  The first 8 partial-O values of every
  // thread are fetched NOW, so that their latency overlaps the (M, L) reduction instead of following it.
  const int y_total = R * D;
  const int y_per = (y_total + nchunks - 1) / nchunks;
  const int y_lo = c * y_per, y_hi = min(y_total, y_lo + y_per);
  float pv[8];
  {
    const int nout = min(y_hi - y_lo, kCombThreads);
    if (nout > 0) {
      const int G = max(1, min(kCombThreads / nout, ns));
      const int oi = threadIdx.x % nout, g = threadIdx.x / nout;
      const int t = y_lo + oi, r = t / D, d = t - r * D;
      const float* po = a.part_o + (size_t)(h * R + r) * ns * D + d;
#pragma unroll
      for (int u = 0; u < 8; u++) pv[u] = (g < G && g + u * G < ns) ? po[(size_t)(g + u * G) * D] : 0.f;
    }
  }
  // final (M, L) per query head: one wave per head, lanes stride over the splits (fixed order: deterministic).
  // The first 64 splits of up to kMLPre heads per wave are fetched before any reduction starts.
  constexpr int kWaves = kCombThreads / 64, kMLPre = 4;
  float2 mlv[kMLPre];
#pragma unroll
  for (int j = 0; j < kMLPre; j++) {
    const int r = wave + j * kWaves;
    mlv[j] = (r < R && lane < ns) ? reinterpret_cast<const float2*>(a.part_ml)[(size_t)(h * R + r) * ns + lane]
                                  : make_float2(-INFINITY, 0.f);
  }
#pragma unroll
  for (int j = 0; j < (kMaxR + kWaves - 1) / kWaves; j++) {
    const int r = wave + j * kWaves;
    if (r >= R) break;
    const float2* ml = reinterpret_cast<const float2*>(a.part_ml) + (size_t)(h * R + r) * ns;
    float2 first = make_float2(-INFINITY, 0.f);
    if (j < kMLPre) {
#pragma unroll
      for (int t = 0; t < kMLPre; t++) first = (t == j) ? mlv[t] : first;
    } else if (lane < ns) {
      first = ml[lane];
    }
    float mi = first.x;
    for (int i = lane + 64; i < ns; i += 64) mi = fmaxf(mi, ml[i].x);
    const float M = wave_max_uniform(mi);
    const float Mu = (M == -INFINITY) ? 0.f : M;
    float L = 0.f;
    if (lane < ns) {
      const float w = exp_nonpos(first.x - Mu);
      sm_wdyn[r * ns + lane] = w;
      L = first.y * w;
    }
    for (int i = lane + 64; i < ns; i += 64) {
      const float2 v = ml[i];
      const float w = exp_nonpos(v.x - Mu);
      sm_wdyn[r * ns + i] = w;
      L = fmaf(v.y, w, L);
    }
    L = wave_sum_uniform(L);
    if (lane == 0) {
      sm_M[r] = Mu;
      sm_L[r] = L;
    }
  }
  __shared__ float sm_l2[kCombThreads / 64];
  if (a.next_key && a.policy == 4) {  // l2: global maximum of the norms (torch.max propagates NaN)
    const float wm = wave_max_uniform(l2_part);
    const bool nn = __any(l2_nan) != 0;
    if (lane == 0) sm_l2[wave] = nn ? NAN : wm;
  }
  __syncthreads();

  // y (continued): weights are in LDS now.  The partial products of the first output group go to LDS, then the
  // per-slot pass and the next-key reduction run, and ONE barrier later the fixed-order final sums are taken
  // (deterministic) — the LDS round trip overlaps the per-slot work instead of adding three barriers.
  __shared__ float sm_y[kCombThreads];
  __shared__ unsigned long long sm_k[kWaves];
  auto y_partial = [&](int o0, bool prefetched, int& nout, int& G, int& oi, int& g, int& r, int& d) -> float {
    nout = min(y_hi - o0, kCombThreads);
    G = max(1, min(kCombThreads / nout, ns));
    oi = threadIdx.x % nout;
    g = threadIdx.x / nout;
    const int t = o0 + oi;
    r = t / D;
    d = t - r * D;
    float part = 0.f;
    if (g < G) {
      const float* po = a.part_o + (size_t)(h * R + r) * ns * D + d;
      int i = g;
      if (prefetched) {
#pragma unroll
        for (int u = 0; u < 8; u++)
          if (i + u * G < ns) part = fmaf(pv[u], sm_wdyn[r * ns + i + u * G], part);
        i += 8 * G;
      }
      for (; i + 7 * G < ns; i += 8 * G) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; u++) v[u] = po[(size_t)(i + u * G) * D];
#pragma unroll
        for (int u = 0; u < 8; u++) part = fmaf(v[u], sm_wdyn[r * ns + i + u * G], part);
      }
      for (; i < ns; i += G) part = fmaf(po[(size_t)i * D], sm_wdyn[r * ns + i], part);
    }
    return part;
  };
  auto y_final = [&](int nout, int G, int oi, int g, int r, int d) {
    if (g == 0) {
      float O = 0.f;
      for (int gg = 0; gg < G; gg++) O += sm_y[gg * nout + oi];
      ElemTraits<T>::store(reinterpret_cast<T*>(a.y), (size_t)(h * R + r) * D + d, O / sm_L[r]);
    }
  };
  const bool do_y = !(a.abl & 16) && y_lo < y_hi;
  int y_nout = 1, y_G = 1, y_oi = 0, y_g = 1, y_r = 0, y_d = 0;
  if (do_y) sm_y[threadIdx.x] = y_partial(y_lo, true, y_nout, y_G, y_oi, y_g, y_r, y_d);

  bool fresh = false;
  float hyb_ws = 0.f;
  int32_t hyb_dn = 1;
  // probabilities for this thread's slot
  if (have && !(a.abl & 32)) {
    const int s = s_mine;
    float sum = 0.f;
    for (int r = 0; r < R; r++) {
      const size_t j = (size_t)h * R + r;
      float x = 0.f;
      if (r < kPre) {
#pragma unroll
        for (int t = 0; t < kPre; t++) x = (t == r) ? xs[t] : x;
      } else {
        x = ElemTraits<T>::load(sc, j * S + s);
      }
      // ref: attention_utils.py:52 softmax (fp32 inside, result rounded to the model dtype)
      const float p = ElemTraits<T>::rnd(__fdiv_rn(exp_nonpos(x - sm_M[r]), sm_L[r]));
      if (a.probs_out) ElemTraits<T>::store(reinterpret_cast<T*>(a.probs_out), j * S + s, p);
      sum += p;
    }
    // ref: model.py:416-418 mean over the R query heads of the group -> model dtype
    const float av = ElemTraits<T>::rnd(__fdiv_rn(sum, (float)R));
    const size_t i = (size_t)h * S + s;
    if (a.attn_out) ElemTraits<T>::store(reinterpret_cast<T*>(a.attn_out), i, av);
    if (a.ring_num) {  // fused cache.py:716-723, W > 1: ring[h, s, counter % W] = attn; denom += 1; window sum kept exact
      if (a.next_key && a.policy == 5 && ps_mine == p_next - 1) {  // two-launch step: this slot was evicted and refilled by the streaming
        racc = WAcc{0, 0, 0, 0};                  // pass — its history starts from zero (cache.py:754-763); the rest of
        ring_old = 0.f;                           // the ring row and shadow column is cleared below, by the whole wave
        den_old = 0;
        fresh = true;
      }
      ElemTraits<T>::store(reinterpret_cast<T*>(a.ring_num), i * (size_t)a.ring_W + ring_col, av);
      ElemTraits<T>::store(ring_shadow, i, av);
      a.hh_denom[i] = den_old + 1;
      wacc_add_value(racc, av, false);
      wacc_add_value(racc, ring_old, true);
      *reinterpret_cast<ulonglong2*>(a.ring_acc + i * 4) = make_ulonglong2(racc.w0, racc.w1);
      *reinterpret_cast<ulonglong2*>(a.ring_acc + i * 4 + 2) = make_ulonglong2(racc.w2, racc.special);
      const float ws_new = wacc_round<T>(racc);
      a.ring_wsum[i] = ws_new;
      hyb_ws = ws_new;
      hyb_dn = den_old + 1;
      if (a.next_key && a.policy == 5) {  // next eviction score of the windowed history (cache.py:727-749, W > 1)
        const int32_t dn = den_old + 1;
        float scn = __fdiv_rn(ws_new, (float)(dn < 1 ? 1 : (dn > a.ring_W ? a.ring_W : dn)));
        if (ps_mine < a.g || ps_mine >= p_next - a.w) scn = 1.0f;
        if (ps_mine == -1) scn = 0.0f;
        my_key = make_key(orderable_f32(scn), ((uint32_t)s << 1) | (uint32_t)(ps_mine == -1));
      }
    }
    if (a.hh_num) {  // fused cache.py:716-722 (W == 1, attention already padded to S)
      const double num_new = num_old + (double)av;
      const int32_t den_new = den_old + 1;
      a.hh_num[i] = num_new;
      a.hh_denom[i] = den_new;
      if (a.next_key && a.policy == 1) {  // next step's eviction score from the freshly updated history (cache.py:727-749)
        float scn = __fdiv_rn((float)num_new, (float)(den_new < 1 ? 1 : den_new));
        if (ps_mine < a.g || ps_mine >= p_next - a.w) scn = 1.0f;
        if (ps_mine == -1) scn = 0.0f;
        my_key = make_key(orderable_f32(scn), ((uint32_t)s << 1) | (uint32_t)(ps_mine == -1));
      }
    }
  }
  if (a.ring_num && a.next_key) {  // clear the rest of a refilled slot's ring row and shadow column (one slot per head and step)
    unsigned long long fm = __ballot(fresh);
    while (fm) {
      const int src = __builtin_ctzll(fm);
      fm &= fm - 1;
      const int sl = __shfl(s_mine, src, CC_WAVE);
      const size_t hs = (size_t)gridDim.y * S, i = (size_t)h * S + sl;
      T* ring = reinterpret_cast<T*>(a.ring_num) + i * (size_t)a.ring_W;
      T* shadow0 = reinterpret_cast<T*>(a.ring_acc + hs * 4 + 2);
      for (int j = lane; j < a.ring_W; j += 64)
        if (j != ring_col) {
          ElemTraits<T>::store(ring, j, 0.f);
          ElemTraits<T>::store(shadow0, (size_t)j * hs + i, 0.f);
        }
    }
  }
  if (a.next_key && a.policy == 2 && have && s_mine >= a.g)  // arg-min of pos over the slots behind the sinks; -1 = empty first (every kv head: its own key row)
    my_key = make_key(orderable_i32(ps_mine), ((uint32_t)s_mine << 1) | (uint32_t)(ps_mine == -1));
  if (a.next_key && a.policy == 4 && have) {  // ref: cache.py:597-605: dtype(max - norm), recent window -> +inf, base rules
    float gm = -INFINITY;
    bool gn = false;
#pragma unroll
    for (int w2 = 0; w2 < kCombThreads / 64; w2++) {
      const float v = sm_l2[w2];
      gn |= (v != v);
      gm = fmaxf(gm, v);
    }
    float scn = ElemTraits<T>::rnd((gn ? NAN : gm) - kn_mine);
    if (ps_mine >= p_next - a.w) scn = INFINITY;
    if (s_mine < a.g) scn = INFINITY;
    if (ps_mine == -1) scn = -INFINITY;
    my_key = make_key(orderable_f32(scn), ((uint32_t)s_mine << 1) | (uint32_t)(ps_mine == -1));
  }
  if (a.next_key && a.policy == 3 && have) {  // ref: cache.py:523 recent window -> +inf, then the base rules :373-376
    float scn = a.rand_next ? rnd_mine : cc_rng_uniform(a.rng_seed, p_next, s_mine);
    if (ps_mine >= p_next - a.w) scn = INFINITY;
    if (s_mine < a.g) scn = INFINITY;
    if (ps_mine == -1) scn = -INFINITY;
    my_key = make_key(orderable_f32(scn), ((uint32_t)s_mine << 1) | (uint32_t)(ps_mine == -1));
  }
  if (a.next_key && a.policy == 6 && have) {  // ref: cache.py:844-894 _eviction_idx_for_head at position p + 1
    const int flags = hyb_flags, win = hyb_win, cts_n = hyb_cts_n;
    if ((flags & (HF_HH | HF_WIN)) && !(flags & HF_FULL) && s_mine < (cts_n < S ? cts_n : S)) {
      float scn;
      if (flags & HF_HH) {
        const int32_t d = hyb_dn > a.hyb.W ? a.hyb.W : hyb_dn;  // clamp_max only (:868-870)
        scn = __fdiv_rn(hyb_ws, (float)d);
      } else {
        scn = (float)ps_mine;  // :873
      }
      bool save = s_mine < a.g || hyb_save;  // :876 first g SLOTS, :878-883 special / punctuation slots
      if (flags & HF_WIN) save |= ps_mine > p_next - win;  // :885-889 strict
      if (save) scn = INFINITY;
      my_key = make_key(orderable_f32(scn), (uint32_t)s_mine << 1);
    }
  }
  if (a.next_key && !(a.abl & 8)) {
    const unsigned long long wk = wave_min_u64_uniform(my_key);
    if (lane == 0) sm_k[wave] = wk;
  }
  __syncthreads();
  if (do_y) y_final(y_nout, y_G, y_oi, y_g, y_r, y_d);
  if (a.next_key && !(a.abl & 8) && threadIdx.x == 0) {
    unsigned long long bk = sm_k[0];
#pragma unroll
    for (int w2 = 1; w2 < kWaves; w2++) bk = sm_k[w2] < bk ? sm_k[w2] : bk;
    // the minimum over all blocks of this head IS torch's arg-min; the next step's streaming pass takes it
    // (plain store: same-address atomics from 8 XCDs measured +4.5 us on this 5 us kernel)
    // (a head's key row has kNextKeyPerChunk * nchunks entries — one per wave of the single-launch step's 64-slot workgroups;
    //  all but the first nchunks stay ~0 here)
    {
      unsigned long long* row = a.next_key + (size_t)h * (kNextKeyPerChunk * nchunks + kNextKeyTail);  // (the row stride: cc_next_key_slots)
      row[c] = bk;
    }
  }
  if (do_y)  // further output groups (only when R*D / n_chunks > 128, i.e. very short caches)
    for (int o0 = y_lo + kCombThreads; o0 < y_hi; o0 += kCombThreads) {
      __syncthreads();
      sm_y[threadIdx.x] = y_partial(o0, false, y_nout, y_G, y_oi, y_g, y_r, y_d);
      __syncthreads();
      y_final(y_nout, y_G, y_oi, y_g, y_r, y_d);
    }
  if ((a.hh_num || a.ring_num) && a.hh_counter && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) *a.hh_counter += 1;
  if (a.next_key && a.policy == 6 && blockIdx.x == 0 && threadIdx.x == 0) {
    a.cache_cts[h] = a.hyb.cts_next[h];  // every block of this head has read cts_next, not cache_cts
    if (h == 0 && a.hyb.num_punc && a.hyb.token_id && a.hyb.punc_ids) {  // ref: cache.py:1017, once per step
      const long long id = *a.hyb.token_id;
      bool f = false;
      for (int k2 = 0; k2 < a.hyb.n_punc_ids; k2++) f |= a.hyb.punc_ids[k2] == id;
      if (f) *a.hyb.num_punc += 1;
    }
  }
}

// ---------------------------------------------------------------- launch plan
// 4-wave workgroups: ~150 VGPRs -> 3 waves/SIMD -> 3 workgroups resident per CU (33 KiB LDS each), so one
// workgroup's load phase overlaps another's math/epilogue.  (8-wave workgroups at 66 KiB LDS left ONE resident
// workgroup per CU and plateaued at ~3 TB/s even at S = 65536: profiles/r01_sweep_before.txt.)
constexpr int kNW = 4;  // waves per split workgroup
constexpr int kU = 4;   // 16-byte K loads (and V loads) in flight per lane

struct Plan {
  int n_split, rows_per_split, rt, chunk, n_chunks;
  int nw;  // waves per workgroup of the streaming pass / the single-launch step (4, or 8: see make_plan)
};

static int rows_per_iter(int D, int dtype, int nw = kNW) {
  const int vec = 16 / (int)cc_dt_size(dtype);
  const int lpr = D / vec;
  return (64 / lpr) * kU * nw;
}

// The step of ONE cache must always get the same plan, whichever form (one launch, two, three calls) and whatever rides the
// streaming pass (plain attention, l2, hybrid, the fused quantised cache): its partials — hence the last bits of (M, L) — depend on
// the geometry.
static std::atomic<int> g_wide_enabled{1};  // cc_decode_step_set_wide: 8-wave workgroups where the plan allows them (process-wide A/B switch)
static Plan make_plan_w(int HQ, int H, int S, int D, int dtype, bool wide_allowed);
static Plan make_plan(int HQ, int H, int S, int D, int dtype, bool narrow = false) {
  return make_plan_w(HQ, H, S, D, dtype, !narrow && g_wide_enabled.load(std::memory_order_relaxed) != 0);
}
static Plan make_plan_w(int HQ, int H, int S, int D, int dtype, bool wide_allowed) {
  Plan p;
  const int R = HQ / H;
  p.nw = kNW;
  p.rt = (R % 4 == 0) ? 4 : (R % 2 == 0) ? 2 : 1;
  // the matrix-core streaming pass has 16 score columns: 8 query heads per pass read K and V ONCE for a group of 8 (Llama-3
  // 70B: HQ / H = 8; with 4 per pass every K / V row was streamed twice)
  if (R % 8 == 0 && cc_dt_size(dtype) == 2 && D == 128) p.rt = 8;
  // (r3) ONE 8-wave workgroup per CU instead of two 4-wave ones, where the cache has 16-row tiles for it (>= 5 x 256 over the
  // kv heads) and every wave still gets exactly one: half the publishers, granules and polls of the in-launch hand-off, half the
  // splits for every gatherer to fold, one merge per CU — 4 or 8 query heads per kv head; EVERY policy (the hybrid cache's real
  // sizes never qualify — several tiles per wave — but a step and the plain attention of its three-call twin must fold their
  // partials alike at every size: the geometry decides the last bits of (M, L))
  // narrow: the 4-wave plan whatever the switch says (the VALU measurement pass has no 8-wave form)
  if (wide_allowed && cc_dt_size(dtype) == 2 && D == 128 && (p.rt == 4 || p.rt == 8) && R == p.rt) {
    const long tiles = (long)H * ((S + 15) / 16);
    // (measured, r3, same box: 1280 tiles (C2: S = 2560) wide 8.18 vs 8.46 us; 1024 tiles (S = 2048, or 4 kv heads at 4096) wide
    //  7.8-7.9 vs 7.35-7.75: from five tiles per CU on the 8-wave workgroup pays)
    // (r5: with a wave's partial O rows inside its K slab the workgroup holds 69.8 KB of LDS and TWO fit a CU — but up to 512 wide
    //  workgroups, i.e. caches of up to 64 x 128 rows per kv head on the single-tile form, measured WORSE than the several-tiles form
    //  they would replace: S = 5120 / 6144 / 8192 at 8 kv heads 11.35 / 11.93 / 13.6 us against 10.21 / 10.81 / 12.07, same box; the
    //  rule stays at one workgroup per CU.  CC_WIDE_MAX_WG: the A/B override, read once)
    static long wide_max = -1;
    if (wide_max < 0) {
      const char* e = getenv("CC_WIDE_MAX_WG");
      wide_max = e ? atol(e) : 256;
    }
    if (tiles >= 1280 && (long)H * ((S + 127) / 128) <= wide_max) p.nw = 8;
#if CC_V_NW16
    if (p.nw == 8 && p.rt == 4 && tiles >= 2048 && g_wide_enabled.load(std::memory_order_relaxed) == 2) p.nw = 16;  // (A/B: see CC_V_NW16)
#endif
  }
  const int rpi = rows_per_iter(D, dtype, p.nw);
  // ~512 workgroups (two per CU, all resident at once): one tile per workgroup up to S = 64 * 64 rows per kv
  // head at 8 kv heads, MORE TILES PER WORKGROUP beyond that — the loop prefetches the next tile behind the
  // current one, the per-workgroup merge and the number of partials the combine pass has to fold stay constant.
  // (512 splits at S = 65536 made every combine block re-reduce 2048 (m, l) pairs: 27 us of a 33 us launch.)
  const int zb = R / p.rt;
  int ns_cap = (512 + H * zb - 1) / (H * zb);  // (768 / 1024 workgroups at S = 18432: 30.4 instead of 26.1 us per step — measured)
  if (ns_cap < 16) ns_cap = 16;
  if (ns_cap > kMaxSplit) ns_cap = kMaxSplit;
  int ns = (S + rpi - 1) / rpi;
  if (ns > ns_cap) ns = ns_cap;
  if (ns < 1) ns = 1;
  int rps = (S + ns - 1) / ns;
  rps = ((rps + rpi - 1) / rpi) * rpi;
  ns = (S + rps - 1) / rps;
  p.n_split = ns;
  p.rows_per_split = rps;
  p.chunk = kNextKeyChunk;  // one partial arg-min key per combine block (the lower half of a head's key row)
  p.n_chunks = (S + p.chunk - 1) / p.chunk;
  return p;
}

static size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

template <typename T, int D>
static int launch_split_rt(const SplitArgs& a, const Plan& p, int H, int R, hipStream_t st) {
  dim3 grid(p.n_split, H, R / p.rt), block(kNW * 64);  // (the VALU pass: fp32 caches / other head dims never get the wide plan)
  switch (p.rt) {
    case 4: hipLaunchKernelGGL((decode_attn_split_kernel<T, D, 4, kNW, kU>), grid, block, 0, st, a); break;
    case 2: hipLaunchKernelGGL((decode_attn_split_kernel<T, D, 2, kNW, kU>), grid, block, 0, st, a); break;
    default: hipLaunchKernelGGL((decode_attn_split_kernel<T, D, 1, kNW, kU>), grid, block, 0, st, a); break;
  }
  CC_LAUNCH_CHECK();
  return CC_OK;
}

template <typename T, bool L2, bool HYB, int QB, int NSUB, int NW = kNW>
static int launch_mfma_rt(const SplitArgs& a, int rt, dim3 grid, dim3 block, hipStream_t st) {
  if constexpr (sizeof(T) == 2) {
    switch (rt) {
      case 8: hipLaunchKernelGGL((decode_attn_split_mfma_kernel<T, 8, NW, L2, false, HYB, QB, NSUB>), grid, block, 0, st, CC_LEAD_ARGS(a) a); break;
      case 4: hipLaunchKernelGGL((decode_attn_split_mfma_kernel<T, 4, NW, L2, false, HYB, QB, NSUB>), grid, block, 0, st, CC_LEAD_ARGS(a) a); break;
      case 2:
        if constexpr (QB != 0 || NW != kNW) return CC_ERR_UNSUPPORTED;
        else hipLaunchKernelGGL((decode_attn_split_mfma_kernel<T, 2, NW, L2, false, HYB, QB, NSUB>), grid, block, 0, st, CC_LEAD_ARGS(a) a);
        break;
      default:
        if constexpr (QB != 0 || NW != kNW) return CC_ERR_UNSUPPORTED;
        else hipLaunchKernelGGL((decode_attn_split_mfma_kernel<T, 1, NW, L2, false, HYB, QB, NSUB>), grid, block, 0, st, CC_LEAD_ARGS(a) a);
        break;
    }
    return CC_OK;
  } else {
    return CC_ERR_UNSUPPORTED;
  }
}

template <typename T>
static int launch_split(const SplitArgs& a, const Plan& p, int H, int R, int D, hipStream_t st) {
  if constexpr (sizeof(T) == 2) {
    if (D == 128 && !(a.abl & 32)) {  // matrix-core streaming pass (abl bit 32 = measurement: force the VALU kernel)
      static_assert(kU == 4, "the MFMA tile is 4 row groups x 4 rows per wave");
      dim3 grid(p.n_split, H, R / p.rt), block(p.nw * 64);
      // (NSUB = 2 — two tiles per wave and iteration, each with its own staging registers, the loads two half-iterations ahead of
      //  their use — was measured at S = 18432: 27.3 instead of 25.4 us per step, the uint8 instantiation 31.5 instead of 25.4; like
      //  768 / 1024 workgroups, more bytes in flight per CU make this access pattern slower, not faster.  Not instantiated.)
      int rc;
      if (a.qparams != nullptr) {  // fused quantised cache: 4 or 8 query heads per kv head
        if (p.rt != 4 && p.rt != 8) return CC_ERR_UNSUPPORTED;
        rc = p.nw == 8 ? launch_mfma_rt<T, false, false, 8, 1, 8>(a, p.rt, grid, block, st) : launch_mfma_rt<T, false, false, 8, 1>(a, p.rt, grid, block, st);
      } else if (a.hyb.strategies != nullptr) {
        rc = p.nw == 8 ? launch_mfma_rt<T, false, true, 0, 1, 8>(a, p.rt, grid, block, st) : launch_mfma_rt<T, false, true, 0, 1>(a, p.rt, grid, block, st);
      } else if (a.key_norm != nullptr) {
        rc = p.nw == 8 ? launch_mfma_rt<T, true, false, 0, 1, 8>(a, p.rt, grid, block, st) : launch_mfma_rt<T, true, false, 0, 1>(a, p.rt, grid, block, st);
      } else if (p.nw == 8) {
        rc = launch_mfma_rt<T, false, false, 0, 1, 8>(a, p.rt, grid, block, st);
      } else {
        rc = launch_mfma_rt<T, false, false, 0, 1>(a, p.rt, grid, block, st);
      }
      if (rc != CC_OK) return rc;
      CC_LAUNCH_CHECK();
      return CC_OK;
    }
  }
  if (a.qparams != nullptr) return CC_ERR_UNSUPPORTED;  // the fused quantised cache exists in the matrix-core streaming pass only
  if (p.rt > 4) return CC_ERR_UNSUPPORTED;  // 8 heads per pass exist on the matrix-core path only (reached here only by the measurement switch)
  if (a.hyb.strategies != nullptr) return CC_ERR_UNSUPPORTED;  // the hybrid decision exists in the matrix-core streaming pass only
  switch (D) {
    case 16: return launch_split_rt<T, 16>(a, p, H, R, st);
    case 32: return launch_split_rt<T, 32>(a, p, H, R, st);
    case 64: return launch_split_rt<T, 64>(a, p, H, R, st);
    case 128: return launch_split_rt<T, 128>(a, p, H, R, st);
    default: break;
  }
  return CC_ERR_UNSUPPORTED;
}

}  // namespace

extern "C" {

}  // extern "C"

namespace {
// ---- single-launch layer step: shape eligibility, workspace regions, residency
// Workspace layout: [epoch words + timeout word: 4 KiB][(m, l) + l2 norm granules: 32 heads x 9 KiB][O granules: 32 heads x 512 KiB]
//                   [QKV granules: 32 heads x 5 KiB][two-launch scratch].
// The single-launch regions sit at FIXED offsets and fixed capacities, whatever the shape: caches of different lengths
// (pyramid budgets) share one workspace and one set of epoch words, tags grow monotonically across all of them, and
// nothing but the single-launch kernel ever writes a word that could be mistaken for a tag.
// (kOneHdrBytes, kOneMlCap, kOneOCap: cc_attn_decode_kernels.h — the LDS-DMA steps derive the granule regions from the header's address)
constexpr size_t kOneQCap = (size_t)kOneMaxHeads * kOneQHead;  // r5: the fused QKV projection's granules (behind the O granules)
constexpr size_t kOneBytes = kOneHdrBytes + kOneMlCap + kOneOCap + kOneQCap;
constexpr int kOneStatusWord = 1023;  // hdr[0 .. H): epochs; hdr[1023]: timeout word
constexpr int kOneMaxTiles = 16;  // tiles per wave the single-launch step keeps scores for (NT = 4, 8 or 16 instantiations; hybrid: up to 8)
// tiles per wave of the single-launch step for this shape: 1 = the specialised single-tile form, 2 .. 8 = the multi-tile form
// (16-bit caches, 4 or 8 query heads per kv head), 0 = not eligible
static int one_tiles(const Plan& p, int HQ, int H, int D, int dtype) {
  const int R = HQ / H, rpi = rows_per_iter(D, dtype, p.nw);
  if (cc_dt_size(dtype) != 2 || D != 128 || R != p.rt || p.n_split > 64 || H > kOneMaxHeads || p.rt > 8 || p.rows_per_split % rpi) return 0;
  const int nt = p.rows_per_split / rpi;
  if (nt == 1) return 1;
  return (nt <= kOneMaxTiles && (p.rt == 4 || p.rt == 8)) ? nt : 0;
}
static bool one_shape_ok(const Plan& p, int HQ, int H, int D, int dtype) { return one_tiles(p, HQ, H, D, dtype) > 0; }
static size_t base_workspace_bytes(const Plan& p, int HQ, int H, int S, int D, int dtype) {
  return align256((size_t)HQ * S * cc_dt_size(dtype)) + align256((size_t)HQ * p.n_split * 2 * sizeof(float)) +
         align256((size_t)HQ * p.n_split * D * sizeof(float)) + 256 +  // + the ring column word of the fused W > 1 history
         align256(((size_t)H * p.n_split * kNW + H) * sizeof(float));   // + the l2 policy's partial maxima and new norms
}
// workgroups of the single-launch kernel the device keeps resident at once (0: unknown -> never use it)
// The answer is cached PER KERNEL: every instantiation has the same function type, so a cache keyed by the argument's type
// (a `static` inside a template over the type) would be ONE cache for all of them — the first kernel asked would answer for
// the fused-quant and multi-tile instantiations too, which keep fewer workgroups resident.
static int one_capacity(void (*kernel)(CC_LEAD_TYPES SplitArgs), int threads) {
  struct Entry {
    void (*k)(CC_LEAD_TYPES SplitArgs);
    int cap;
  };
  // readers take no lock: an entry is complete before the count that makes it visible is published (release / acquire); two
  // threads that miss together both ask the runtime and the second append is refused as a duplicate under the lock
  static Entry cache[64];
  static std::atomic<int> n_cached{0};
  static std::mutex mu;
  const int n = n_cached.load(std::memory_order_acquire);
  for (int i = 0; i < n; i++)
    if (cache[i].k == kernel) return cache[i].cap;
  int dev = 0, cus = 0, nb = 0;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess ||
      hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kernel, threads, 0) != hipSuccess) {
    (void)hipGetLastError();
    return 0;
  }
  const int cap = cus * (nb > 8 ? 8 : nb);
  std::lock_guard<std::mutex> lock(mu);
  const int m = n_cached.load(std::memory_order_relaxed);
  for (int i = 0; i < m; i++)
    if (cache[i].k == kernel) return cache[i].cap;
  if (m < 64) {
    cache[m] = Entry{kernel, cap};
    n_cached.store(m + 1, std::memory_order_release);
  }
  return cap;
}
typedef void (*OneKernel)(CC_LEAD_TYPES SplitArgs);
// The single-launch kernel that serves (query heads per kv head rt, tiles per wave nt, kind), or null.  kind: 0 = 16-bit cache
// (heavy hitter / head-constant policies), 8 = fused quantised cache, -1 = l2, 200 = hybrid.  full: the instantiation with the
// measurement hooks and attn_out (bf16, rt = 4 only).  ONE table for the residency check and the launch.
template <typename T>
static OneKernel one_kernel(int rt, int nt, int kind, bool full, int nw = kNW) {
#define CC_ONE_K(RT_, L2_, HYB_, QB_, NT_, FULL_) decode_attn_split_mfma_kernel<T, RT_, kNW, L2_, true, HYB_, QB_, 1, NT_, FULL_>
  if (nt < 1 || nt > kOneMaxTiles) return nullptr;
#if CC_V_NW16
  if (nw == 16) return (rt == 4 && nt == 1 && kind == 0 && !full) ? decode_attn_split_mfma_kernel<T, 4, 16, false, true, false, 0, 1, 1, false> : nullptr;
#endif
  if (nw == 8) {  // ONE 8-wave workgroup per CU: 4 or 8 query heads per kv head, one tile per wave (make_plan)
#define CC_ONE_W(RT_, L2_, QB_, FULL_) decode_attn_split_mfma_kernel<T, RT_, 8, L2_, true, false, QB_, 1, 1, FULL_>
#define CC_ONE_WH(RT_, FULL_) decode_attn_split_mfma_kernel<T, RT_, 8, false, true, true, 0, 1, 1, FULL_>
    if ((kind != 0 && kind != 8 && kind != -1 && kind != 200) || nt != 1 || (rt != 4 && rt != 8)) return nullptr;
    if (full) {
      if constexpr (ElemTraits<T>::code != CC_DT_BF16) {
        return nullptr;
      } else {
        if (rt != 4) return nullptr;
        if (kind == 200) return CC_ONE_WH(4, true);
        return kind == 0 ? CC_ONE_W(4, false, 0, true) : (kind == 8 ? CC_ONE_W(4, false, 8, true) : CC_ONE_W(4, true, 0, true));
      }
    }
    if (kind == 200) return rt == 8 ? CC_ONE_WH(8, false) : CC_ONE_WH(4, false);
    if (kind == 0) return rt == 8 ? CC_ONE_W(8, false, 0, false) : CC_ONE_W(4, false, 0, false);
    if (kind == 8) return rt == 8 ? CC_ONE_W(8, false, 8, false) : CC_ONE_W(4, false, 8, false);
    return rt == 8 ? CC_ONE_W(8, true, 0, false) : CC_ONE_W(4, true, 0, false);
#undef CC_ONE_W
#undef CC_ONE_WH
  }
  if (nw != kNW) return nullptr;
  if (full) {
    if constexpr (ElemTraits<T>::code != CC_DT_BF16) {
      return nullptr;
    } else {
      if (rt != 4) return nullptr;
      if (kind == 200) return nt == 1 ? CC_ONE_K(4, false, true, 0, 1, true) : (nt <= 8 ? CC_ONE_K(4, false, true, 0, 8, true) : nullptr);
      if (kind == -1) return nt == 1 ? CC_ONE_K(4, true, false, 0, 1, true) : nullptr;
      if (kind == 8) return nt == 1 ? CC_ONE_K(4, false, false, 8, 1, true) : nullptr;
      if (kind == 0) return nt == 1 ? CC_ONE_K(4, false, false, 0, 1, true) : (nt <= 4 ? CC_ONE_K(4, false, false, 0, 4, true) : (nt <= 8 ? CC_ONE_K(4, false, false, 0, 8, true) : nullptr));
      return nullptr;
    }
  }
  if (kind == 200 && nt > 8) return nullptr;
  if (kind == 200) {  // hybrid: 4 or 8 query heads per kv head; one tile per wave or up to eight
    if (rt == 8) return nt == 1 ? CC_ONE_K(8, false, true, 0, 1, false) : CC_ONE_K(8, false, true, 0, 8, false);
    if (rt == 4) return nt == 1 ? CC_ONE_K(4, false, true, 0, 1, false) : CC_ONE_K(4, false, true, 0, 8, false);
    return nullptr;
  }
  if (kind == -1) {  // l2: one tile per wave
    if (nt != 1) return nullptr;
    switch (rt) {
      case 8: return CC_ONE_K(8, true, false, 0, 1, false);
      case 4: return CC_ONE_K(4, true, false, 0, 1, false);
      case 2: return CC_ONE_K(2, true, false, 0, 1, false);
      case 1: return CC_ONE_K(1, true, false, 0, 1, false);
      default: return nullptr;
    }
  }
  if (kind == 8) {  // fused quantised cache: 4 or 8 query heads per kv head, one tile per wave
    if (nt != 1) return nullptr;
    if (rt == 8) return CC_ONE_K(8, false, false, 8, 1, false);
    if (rt == 4) return CC_ONE_K(4, false, false, 8, 1, false);
    return nullptr;
  }
  if (kind != 0) return nullptr;
  if (nt > 1) {  // several tiles per wave (long caches): 4 or 8 query heads per kv head
    if (rt == 8) return nt <= 4 ? CC_ONE_K(8, false, false, 0, 4, false) : (nt <= 8 ? CC_ONE_K(8, false, false, 0, 8, false) : CC_ONE_K(8, false, false, 0, 16, false));
    if (rt == 4) return nt <= 4 ? CC_ONE_K(4, false, false, 0, 4, false) : (nt <= 8 ? CC_ONE_K(4, false, false, 0, 8, false) : CC_ONE_K(4, false, false, 0, 16, false));
    return nullptr;
  }
  switch (rt) {
    case 8: return CC_ONE_K(8, false, false, 0, 1, false);
    case 4: return CC_ONE_K(4, false, false, 0, 1, false);
    case 2: return CC_ONE_K(2, false, false, 0, 1, false);
    case 1: return CC_ONE_K(1, false, false, 0, 1, false);
    default: return nullptr;
  }
#undef CC_ONE_K
}
static OneKernel one_kernel_dt(int dtype, int rt, int nt, int kind, bool full, int nw) {
  return dtype == CC_DT_BF16 ? one_kernel<bf16_t>(rt, nt, kind, full, nw) : (dtype == CC_DT_F16 ? one_kernel<f16_t>(rt, nt, kind, full, nw) : nullptr);
}
// The XL2 instantiations (placement + L2-resident hand-off): single-tile steps of the plain 16-bit cache (kind 0), the fused
// quantised cache (8) and l2 (-1), 4 or 8 query heads per kv head, 4- or 8-wave workgroups; FULL for bf16 / rt = 4 / kind 0 only.
template <typename T>
static OneKernel one_kernel_xl2(int rt, int nt, int kind, bool full, int nw) {
#define CC_ONE_X(RT_, NW_, L2_, QB_, FULL_) decode_attn_split_mfma_kernel<T, RT_, NW_, L2_, true, false, QB_, 1, 1, FULL_, true>
#define CC_ONE_XM(RT_, HYB_, NT_) decode_attn_split_mfma_kernel<T, RT_, kNW, false, true, HYB_, 0, 1, NT_, false, true>
#if CC_V_NW16
  if (nw == 16) return (rt == 4 && nt == 1 && kind == 0 && !full) ? CC_ONE_X(4, 16, false, 0, false) : nullptr;
#endif
  if ((rt != 4 && rt != 8) || (nw != 4 && nw != 8)) return nullptr;
  if (nt > 1 || kind == 200) {  // several tiles per wave (4-wave workgroups) and the hybrid cache's steps: lean instantiations only
    if (full || nt > kOneMaxTiles) return nullptr;
    if (kind == 200 && nt > 8) return nullptr;
    if (kind == 200) {
      if (nt == 1) return nw == 8 ? (rt == 8 ? decode_attn_split_mfma_kernel<T, 8, 8, false, true, true, 0, 1, 1, false, true>
                                             : decode_attn_split_mfma_kernel<T, 4, 8, false, true, true, 0, 1, 1, false, true>)
                                  : (rt == 8 ? CC_ONE_XM(8, true, 1) : CC_ONE_XM(4, true, 1));
      return nw == 4 ? (rt == 8 ? CC_ONE_XM(8, true, 8) : CC_ONE_XM(4, true, 8)) : nullptr;
    }
    if (kind != 0 || nw != 4) return nullptr;
    if (rt == 8) return nt <= 4 ? CC_ONE_XM(8, false, 4) : (nt <= 8 ? CC_ONE_XM(8, false, 8) : CC_ONE_XM(8, false, 16));
    return nt <= 4 ? CC_ONE_XM(4, false, 4) : (nt <= 8 ? CC_ONE_XM(4, false, 8) : CC_ONE_XM(4, false, 16));
  }
  if (kind != 0 && kind != 8 && kind != -1) return nullptr;
  if (full) {
    if constexpr (ElemTraits<T>::code != CC_DT_BF16) {
      return nullptr;
    } else {
      if (rt != 4 || (kind != 0 && kind != -1)) return nullptr;
      if (kind == -1) return nw == 8 ? CC_ONE_X(4, 8, true, 0, true) : nullptr;  // (l2: the traces' shape)
      return nw == 8 ? CC_ONE_X(4, 8, false, 0, true) : CC_ONE_X(4, 4, false, 0, true);
    }
  }
  if (kind == 0) {
    if (nw == 8) return rt == 8 ? CC_ONE_X(8, 8, false, 0, false) : CC_ONE_X(4, 8, false, 0, false);
    return rt == 8 ? CC_ONE_X(8, 4, false, 0, false) : CC_ONE_X(4, 4, false, 0, false);
  }
  if (kind == 8) {
    if (nw == 8) return rt == 8 ? CC_ONE_X(8, 8, false, 8, false) : CC_ONE_X(4, 8, false, 8, false);
    return rt == 8 ? CC_ONE_X(8, 4, false, 8, false) : CC_ONE_X(4, 4, false, 8, false);
  }
  if (nw == 8) return rt == 8 ? CC_ONE_X(8, 8, true, 0, false) : CC_ONE_X(4, 8, true, 0, false);
  return rt == 8 ? CC_ONE_X(8, 4, true, 0, false) : CC_ONE_X(4, 4, true, 0, false);
#undef CC_ONE_X
#undef CC_ONE_XM
}
static OneKernel one_kernel_xl2_dt(int dtype, int rt, int nt, int kind, bool full, int nw) {
  return dtype == CC_DT_BF16 ? one_kernel_xl2<bf16_t>(rt, nt, kind, full, nw) : (dtype == CC_DT_F16 ? one_kernel_xl2<f16_t>(rt, nt, kind, full, nw) : nullptr);
}

// ---- XL2 eligibility of the device: does block b of a launch run on XCD (b % 8)'s fixed XCC?  Observed once per device by
//      cc_decode_step_probe_xcd (a synchronous launch: never under stream capture — the Python layer calls it when it loads the
//      library and when it creates a decode workspace); unknown = not eligible.
constexpr int kMaxDevices = 64;
struct XccProbe {
  std::atomic<int> state{0};  // 0 unknown, 1 verified, 2 refuted / failed
  std::atomic<int> demoted{0};  // cc_decode_step_demote_l2_handoff: the caller saw a step fail with the L2-resident hand-off on THIS device
  std::atomic<int> one_off{0};  // cc_decode_step_device_single_launch(0): THIS device is shared — its steps take the two-launch forms
};
static XccProbe g_xcc_probe[kMaxDevices];
static std::atomic<int> g_l2_handoff_enabled{1};  // cc_decode_step_set_l2_handoff

__global__ void xcc_probe_kernel(unsigned* out) {
  if (threadIdx.x == 0)
    out[blockIdx.x + gridDim.x * blockIdx.y] = (unsigned)__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11)) & 15u;  // HW_REG_XCC_ID[3:0]
}
// -> true when the L2-resident hand-off may be used on the current device
static bool xl2_device_ok() {
  int dev = 0;
  if (!g_l2_handoff_enabled.load(std::memory_order_relaxed) || hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) return false;
  return g_xcc_probe[dev].state.load(std::memory_order_acquire) == 1 && g_xcc_probe[dev].demoted.load(std::memory_order_relaxed) == 0;
}
}  // namespace

extern "C" {

size_t cc_decode_attn_workspace_bytes(int32_t HQ, int32_t H, int32_t S, int32_t D, int32_t dtype) {
  if (HQ <= 0 || H <= 0 || HQ % H || S <= 0 || D <= 0 || !cc_dt_ok(dtype)) return 0;
  // the larger of the two geometries' needs: the size must not depend on the process-wide cc_decode_step_set_wide switch (a caller
  // that cached it — bench.py does — would otherwise get CC_ERR_WORKSPACE when the switch is flipped later)
  const size_t a = base_workspace_bytes(make_plan_w(HQ, H, S, D, dtype, true), HQ, H, S, D, dtype);
  const size_t b = base_workspace_bytes(make_plan_w(HQ, H, S, D, dtype, false), HQ, H, S, D, dtype);
  return kOneBytes + (a > b ? a : b);
}

// kind: see one_kernel.  Returns the kernel when the shape is eligible AND all its workgroups stay resident at once, else null.
static OneKernel one_pick(int32_t HQ, int32_t H, int32_t S, int32_t D, int32_t dtype, int kind, bool full, bool allow_xl2 = false,
                          bool* is_xl2 = nullptr) {
  if (is_xl2) *is_xl2 = false;
  if (HQ <= 0 || H <= 0 || HQ % H || S <= 0 || D <= 0 || !cc_dt_ok(dtype)) return nullptr;
  const Plan p = make_plan(HQ, H, S, D, dtype);
  const int nt = one_tiles(p, HQ, H, D, dtype);
  if (nt == 0) return nullptr;
  // l2: every thread gathers at most three workgroups' norm maxima
  if (kind == -1 && H * p.n_split > 3 * p.nw * 64) return nullptr;
  // XL2 first: a multiple of 8 kv heads (each head's workgroups on one XCD) on a device whose dispatch order was verified
  // (r6 A/B, CC_V_FEWXCD: 1, 2 or 4 kv heads on a grid of 8 virtual ones — every XCD must hold a head's n_split workgroups)
  const bool virt8 = CC_V_FEWXCD != 0 && (H == 1 || H == 2 || H == 4);
  if (allow_xl2 && ((H & 7) == 0 || virt8) && xl2_device_ok()) {
    const OneKernel kx = one_kernel_xl2_dt(dtype, p.rt, nt, kind, full, p.nw);
    if (kx && p.n_split * (virt8 ? 8 : H) <= one_capacity(kx, p.nw * 64)) {
      if (is_xl2) *is_xl2 = true;
      return kx;
    }
  }
  const OneKernel k = one_kernel_dt(dtype, p.rt, nt, kind, full, p.nw);
  if (!k) return nullptr;
  return p.n_split * H <= one_capacity(k, p.nw * 64) ? k : nullptr;
}
static std::atomic<int> g_one_enabled{1};  // cc_decode_step_set_single_launch (process-wide A/B switch, debug header)
// the single-launch forms are allowed for a launch on the CURRENT device: the process-wide switch AND the device's own (r6: co-residency
// of a launch's workgroups is a fact of one device — a rank that shares its GPU switches ITS device off, not the process)
static bool one_enabled_here() {
  if (!g_one_enabled.load(std::memory_order_relaxed)) return false;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) return true;
  return g_xcc_probe[dev].one_off.load(std::memory_order_relaxed) == 0;
}
// The QKV form of the single-launch step (cc_attn_decode_qkv.hip): the plain 16-bit caches' single-tile step, 4 or 8 query heads per
// kv head, model dim K <= 4096 (two 1 KiB input segments per wave), at most 64 projection rows per workgroup.  -> 0 = no,
// 1 = memory hand-off, 2 = the XL2 placement + L2-resident hand-off
static int qkv_pick(int32_t HQ, int32_t H, int32_t S, int32_t D, int32_t dtype, int32_t K, Plan* plan_out) {
  if (HQ <= 0 || H <= 0 || HQ % H || S <= 0 || D != 128 || cc_dt_size(dtype) != 2 || !cc_dt_ok(dtype) || K < 8 || K % 8 || K > 4096) return 0;
  if (!one_enabled_here()) return 0;
  const Plan p = make_plan(HQ, H, S, D, dtype);
  if (one_tiles(p, HQ, H, D, dtype) != 1 || (p.rt != 4 && p.rt != 8) || (p.nw != 4 && p.nw != 8)) return 0;
  const int nu = (p.rt + 2) * 32;
  if ((nu + p.n_split - 1) / p.n_split > 16) return 0;
  if (plan_out) *plan_out = p;
  if ((H & 7) == 0 && xl2_device_ok() && p.n_split * H <= cc_qkv_step_capacity(dtype, p.rt, p.nw, 1)) return 2;
  return p.n_split * H <= cc_qkv_step_capacity(dtype, p.rt, p.nw, 0) ? 1 : 0;
}
int32_t cc_decode_step_qkv_available(int32_t HQ, int32_t H, int32_t S, int32_t D, int32_t dtype, int32_t K) {
  return qkv_pick(HQ, H, S, D, dtype, K, nullptr) ? 1 : 0;
}
static int32_t one_available(int32_t HQ, int32_t H, int32_t S, int32_t D, int32_t dtype, int kind) {
  return one_pick(HQ, H, S, D, dtype, kind, false) ? 1 : 0;
}
int32_t cc_decode_step_single_launch(int32_t HQ, int32_t H, int32_t S, int32_t D, int32_t dtype) {
  return one_available(HQ, H, S, D, dtype, 0);
}
int32_t cc_decode_step_quant_single_launch(int32_t HQ, int32_t H, int32_t S, int32_t D, int32_t dtype, int32_t n_bit) {
  return n_bit == 8 ? one_available(HQ, H, S, D, dtype, 8) : 0;
}
int32_t cc_decode_step_hybrid_single_launch(int32_t HQ, int32_t H, int32_t S, int32_t D, int32_t dtype) {
  return one_available(HQ, H, S, D, dtype, 200);
}
int32_t cc_decode_step_l2_single_launch(int32_t HQ, int32_t H, int32_t S, int32_t D, int32_t dtype) {
  return one_available(HQ, H, S, D, dtype, -1);
}
// 1 while the fused decode steps may take their single-launch form at all (cc_decode_step_set_single_launch)
int32_t cc_decode_step_single_launch_enabled(void) { return one_enabled_here() ? 1 : 0; }
// Per DEVICE (the current one): enabled == 0 -> steps launched on this device take the two-launch forms from now on (a device shared
// with other processes or kernels: a launch's workgroups are not all resident together there); != 0 -> the single-launch forms
// again.  Other devices of the process are untouched.  -> the previous value.  (The user-facing knob: include/coldcompress.h.)
int32_t cc_decode_step_device_single_launch(int32_t enabled) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) return 1;
  return g_xcc_probe[dev].one_off.exchange(enabled ? 0 : 1, std::memory_order_relaxed) ? 0 : 1;
}

int32_t cc_decode_step_status_offset(void) { return kOneStatusWord * (int32_t)sizeof(unsigned); }
int32_t cc_decode_step_l2_carry(void) { return CC_V_L2CARRY != 0 ? 1 : 0; }
int32_t cc_decode_step_wait_bound_us(void) { return (int32_t)(kOneWaitTicks / 100ull); }  // (s_memrealtime: 100 ticks per microsecond)
int32_t cc_decode_step_commit_stride(void) { return kRcStride; }

static void* g_one_trace = nullptr;
void cc_decode_step_trace(void* buf) { g_one_trace = buf; }

void cc_decode_step_set_single_launch(int32_t enabled) { g_one_enabled.store(enabled ? 1 : 0, std::memory_order_relaxed); }
// 8-wave workgroups (one per CU) for the plain 16-bit caches that have the tiles for them (make_plan); 0 = 4-wave workgroups
// everywhere.  Process-wide; change it only between steps of a cache whose fused pipeline is re-seeded (prepare_decode): the
// geometry decides which entries of a head's key row are live.
void cc_decode_step_set_wide(int32_t enabled) { g_wide_enabled.store(enabled == 2 && CC_V_NW16 != 0 ? 2 : (enabled ? 1 : 0), std::memory_order_relaxed); }

// The L2-resident hand-off (XL2): on by default where cc_decode_step_probe_xcd verified the device; 0 = always the memory hand-off
// (the fallback of a step that fails with it — a kernel captured into a hipGraph keeps the form it was captured with).
void cc_decode_step_set_l2_handoff(int32_t enabled) { g_l2_handoff_enabled.store(enabled ? 1 : 0, std::memory_order_relaxed); }
int32_t cc_decode_step_l2_handoff(void) { return xl2_device_ok() ? 1 : 0; }
// Per DEVICE (the current one), unlike the A/B switch above: what the recovery path of a caller uses.  demoted != 0: steps launched on
// this device from now on take the memory hand-off whatever the probe said; 0: the probe's verdict counts again.  Other devices of the
// process are untouched.  -> the previous value.
int32_t cc_decode_step_demote_l2_handoff(int32_t demoted) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) return 0;
  return g_xcc_probe[dev].demoted.exchange(demoted ? 1 : 0, std::memory_order_relaxed);
}
// Observe where the dispatcher puts the blocks of a 2-D grid on the CURRENT device: synchronous (its own stream, one small
// allocation) — call it outside stream capture.  1 = block b always ran on the XCC of block b % 8 (two grid shapes, two launches
// each): the XL2 instantiations may be used; 0 = not so, or the probe could not run: they never are.  Cached per device.
int32_t cc_decode_step_probe_xcd(void) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) return 0;
  XccProbe& pr = g_xcc_probe[dev];
  const int st0 = pr.state.load(std::memory_order_acquire);
  if (st0 != 0) return st0 == 1 ? 1 : 0;
  static std::mutex mu;
  std::lock_guard<std::mutex> lock(mu);
  if (pr.state.load(std::memory_order_acquire) != 0) return pr.state.load() == 1 ? 1 : 0;
  constexpr int kMaxBlocks = 64 * 16;
  unsigned* dbuf = nullptr;
  hipStream_t st = nullptr;
  bool ok = hipMalloc(&dbuf, kMaxBlocks * sizeof(unsigned)) == hipSuccess && hipStreamCreateWithFlags(&st, hipStreamNonBlocking) == hipSuccess;
  unsigned host[kMaxBlocks];
  int launches = 0;
  const int shapes[4][2] = {{32, 8}, {64, 16}, {256, 1}, {1024, 1}};  // (r6: the XL2 steps are launched as 1-D grids)
  for (int rep = 0; ok && rep < 4; rep++) {
    const int gx = shapes[rep][0], gy = shapes[rep][1], nb = gx * gy;
    hipLaunchKernelGGL(xcc_probe_kernel, dim3(gx, gy), dim3(512), 0, st, dbuf);
    ok = hipGetLastError() == hipSuccess && hipMemcpyAsync(host, dbuf, nb * sizeof(unsigned), hipMemcpyDeviceToHost, st) == hipSuccess &&
         hipStreamSynchronize(st) == hipSuccess;
    // the relation the placement needs, and nothing more: blocks b and b % 8 of ONE launch share an XCD (which XCD block 0 gets
    // differs from launch to launch)
    for (int b = 8; b < nb && ok; b++) ok = host[b] == host[b & 7];
    launches += ok ? 1 : 0;
  }
  if (st) (void)hipStreamDestroy(st);
  if (dbuf) (void)hipFree(dbuf);
  (void)hipGetLastError();
  pr.state.store(ok && launches == 4 ? 1 : 2, std::memory_order_release);
  return ok && launches == 4 ? 1 : 0;
}

}  // extern "C"

namespace {
// Everything the fused decode step adds to the two launches (null = plain attention).
struct FusedStep {
  const cc_kv_view* c;
  const void* k_new;
  const void* v_new;
  const int32_t* input_pos;
  unsigned long long* next_key;
  int g, w;
  int policy;  // 1 = heavy hitter, 2 = recent_global / full, 3 = random, 4 = l2
  const float* rand_next;
  void* key_norm;  // policy 4
  const HybridStep* hyb;  // policy 6
  float* qparams;  // fused quantised cache: c->k_cache / v_cache are the uint8 images, c->dtype the model dtype
  int32_t* commit;  // recoverable hand-off: step_commit [H] or null
  unsigned long long rng_seed;  // policy 3 with rand_next == null: the in-kernel generator's seed
  int rng_on;
  const QkvIn* qkv;  // r5: the layer's QKV projection rides the step (q / k_new / v_new are null; single launch or CC_ERR_UNSUPPORTED)
};
// The W > 1 history ring folded into the combine pass (denom / counter travel as hh_denom / hh_counter).
struct RingHistory {
  void* num;
  int W;
  uint64_t* acc;
  float* wsum;
};
}  // namespace

// the single-launch tail computes probabilities only where a history consumes them: a group-mean output without one
// needs the two-launch step
static bool attn_out_needs_probs(const FusedStep* fs, const void* attn_out) {
  return attn_out != nullptr && fs && fs->policy != 1 && fs->policy != 6;
}

static int attn_impl(const void* q, const void* k, const void* v, const uint8_t* mask, int32_t HQ, int32_t H,
                     int32_t S, int32_t D, int32_t dtype, float scale, void* y, void* attn_out,
                     void* probs_out, double* hh_num, int32_t* hh_denom, int64_t* hh_counter,
                     void* workspace, size_t workspace_bytes, cc_stream_t stream, int32_t phases, const FusedStep* fs,
                     const RingHistory* rh = nullptr) {
  CC_ENTRY();
  if ((!q && !(fs && fs->qkv)) || !k || !v || !y || HQ <= 0 || H <= 0 || HQ % H || S <= 0 || D <= 0 || !cc_dt_ok(dtype) || !workspace)
    return CC_ERR_BAD_ARG;
  if (hh_num && !hh_denom) return CC_ERR_BAD_ARG;
  const int R = HQ / H;
  if (R > kMaxR) return CC_ERR_UNSUPPORTED;
  if (D != 16 && D != 32 && D != 64 && D != 128) return CC_ERR_UNSUPPORTED;
  if (workspace_bytes < cc_decode_attn_workspace_bytes(HQ, H, S, D, dtype)) return CC_ERR_WORKSPACE;
  // what rides the streaming pass decides the plan (one_kernel's kinds)
  const int kind = !fs ? 0 : (fs->qparams ? 8 : (fs->policy == 4 ? -1 : (fs->policy == 6 ? 200 : 0)));
  const Plan p = make_plan(HQ, H, S, D, dtype, ((phases >> 8) & 32) != 0);  // (measurement bit 32 forces the VALU pass: 4 waves)
  if ((size_t)R * p.n_split * sizeof(float) > 64 * 1024) return CC_ERR_UNSUPPORTED;  // combine-kernel LDS budget
  char* ws = reinterpret_cast<char*>(workspace) + kOneBytes;  // the single-launch regions come first, at fixed offsets
  SplitArgs sa{};
  sa.q = q; sa.k = k; sa.v = v; sa.mask = mask;
  sa.scores = ws;
  ws += align256((size_t)HQ * S * cc_dt_size(dtype));
  sa.part_ml = reinterpret_cast<float*>(ws);
  ws += align256((size_t)HQ * p.n_split * 2 * sizeof(float));
  sa.part_o = reinterpret_cast<float*>(ws);
  ws += align256((size_t)HQ * p.n_split * D * sizeof(float));
  if (rh) {
    if (!rh->num || rh->W <= 1 || !rh->acc || !rh->wsum || !hh_denom || !hh_counter || hh_num) return CC_ERR_BAD_ARG;
    sa.ring_counter = hh_counter; sa.ring_col = reinterpret_cast<int*>(ws); sa.ring_W = rh->W;
  }
  sa.S = S; sa.R = R; sa.n_split = p.n_split; sa.rows_per_split = p.rows_per_split; sa.scale = scale;
  sa.abl = (phases >> 8) & 0xff;
  if (fs && fs->policy == 4) {
    if (!fs->key_norm || cc_dt_size(dtype) != 2 || D != 128 || ((phases >> 8) & 32)) return CC_ERR_UNSUPPORTED;
    sa.key_norm = fs->key_norm;
    sa.l2_pmax = reinterpret_cast<float*>(ws + 256);
    sa.l2_new = sa.l2_pmax + (size_t)H * p.n_split * p.nw;
  }
  if (fs && fs->policy == 6) {
    if (!fs->hyb || !fs->hyb->strategies || !fs->hyb->table || cc_dt_size(dtype) != 2 || D != 128 || fs->c->Hp != H || fs->c->Hc != H)
      return CC_ERR_UNSUPPORTED;
    sa.hyb = *fs->hyb;
    // the counts after this step's inserts, [H] int32 behind the ring column word of the workspace
    sa.hyb.cts_next = reinterpret_cast<int32_t*>(ws + 64);
    if ((size_t)H * sizeof(int32_t) > 192) return CC_ERR_UNSUPPORTED;  // 48 kv heads per rank at most (256-byte slot)
    sa.g = fs->g;
  }
  if (fs && fs->qparams) {
    if (cc_dt_size(dtype) != 2 || D != 128 || (fs->policy != 1 && fs->policy != 2 && fs->policy != 3) || rh || probs_out ||
        ((phases >> 8) & 32) || (p.rt != 4 && p.rt != 8))
      return CC_ERR_UNSUPPORTED;
    sa.qparams = fs->qparams;
  }
  if (fs) {
    sa.next_key = fs->next_key; sa.nk = cc_next_key_slots(S);
    // entries any writer may have left non-~0: one per combine block (two-launch step), one per wave of the single-launch workgroups
    sa.nk_read = one_shape_ok(p, HQ, H, D, dtype) ? (p.n_split * p.nw > p.n_chunks ? p.n_split * p.nw : p.n_chunks) : p.n_chunks;
    if (sa.nk_read > cc_next_key_live(S)) sa.nk_read = cc_next_key_live(S);
    sa.input_pos = fs->input_pos; sa.k_new = fs->k_new; sa.v_new = fs->v_new;
    sa.pos = fs->c->pos; sa.mask_w = fs->c->mask; sa.cache_cts = fs->c->cache_cts; sa.num = hh_num; sa.denom = hh_denom;
    sa.H = H; sa.Hc = fs->c->Hc; sa.Hp = fs->c->Hp;
    sa.commit = fs->commit;
  }
#if CC_V_PRELOAD
  if (sa.nk_read > kLeadNkReadMax || H > kLeadHMax) return CC_ERR_UNSUPPORTED;  // (the packed preloaded word: 2^20 live key entries, 2047 kv heads)
#endif
  hipStream_t st = (hipStream_t)stream;
  int rc = CC_OK;
  if (fs && fs->qkv) {
    // ---- the QKV form: the projection rides the single-launch step (cc_attn_decode_qkv.hip), or nothing does
    Plan pq;
    const int form = qkv_pick(HQ, H, S, D, dtype, fs->qkv->K, &pq);
    const bool policy_ok = (fs->policy == 1 && hh_num && hh_denom && fs->c->Hp == H) ||
                           ((fs->policy == 2 || (fs->policy == 3 && (fs->rand_next || fs->rng_on))) && !hh_num && fs->c->Hp == 1);
    if (!form || !policy_ok || rh || probs_out || attn_out || sa.abl != 0 || kind != 0 || (phases & CC_PHASE_TWO_LAUNCH)) return CC_ERR_UNSUPPORTED;
    if (pq.n_split != p.n_split || pq.nw != p.nw) return CC_ERR_UNSUPPORTED;
    char* ob = reinterpret_cast<char*>(workspace);
    sa.one_hdr = reinterpret_cast<unsigned*>(ob);
    sa.one_ml = ob + kOneHdrBytes;
    sa.one_o = ob + kOneHdrBytes + kOneMlCap;
    sa.one_ml_bytes = (unsigned)kOneMlCap;
    sa.one_o_bytes = (unsigned)kOneOCap;
    sa.y = y; sa.hh_counter = hh_counter; sa.g = fs->g; sa.w = fs->w; sa.yc_chunks = p.n_chunks;
    sa.policy = fs->policy; sa.rand_next = fs->rand_next; sa.rng_seed = fs->rng_seed;
    sa.qkv = *fs->qkv;
    sa.qkv.gran = ob + kOneHdrBytes + kOneMlCap + kOneOCap;
    sa.qkv.gran_bytes = (unsigned)kOneQCap;
    sa.qkv.HQ = HQ;
    return cc_qkv_step_launch(&sa, sizeof(sa), dtype, p.rt, p.nw, form == 2 ? 1 : 0, p.n_split, H, st);
  }
  // ---- single-launch layer step (heavy hitter, W == 1): phases bit CC_PHASE_ONE_LAUNCH forces it (error if the shape or
  //      the device's residency does not allow it), CC_PHASE_TWO_LAUNCH forbids it; otherwise it is used whenever it can be
  const bool one_asked = (phases & CC_PHASE_ONE_LAUNCH) != 0;
  if (one_asked || (one_enabled_here() && (phases & 3) == 3 && !(phases & CC_PHASE_TWO_LAUNCH))) {
    const bool policy_ok = fs && ((fs->policy == 1 && hh_num && hh_denom && fs->c->Hp == H) ||
                                  ((fs->policy == 2 || (fs->policy == 3 && (fs->rand_next || fs->rng_on))) && !hh_num && fs->c->Hp == 1) ||
                                  (fs->policy == 4 && fs->key_norm && !hh_num && fs->c->Hp == H) ||
                                  (fs->policy == 6 && !hh_num && fs->c->Hp == H && fs->c->Hc == H));
    // the lean kernels carry no measurement hooks and no attn_out: a call that wants one of them runs a FULL instantiation where
    // there is one (a time stamp alone is not worth leaving the product kernel for)
    const bool want_full = attn_out != nullptr || sa.abl != 0;
    OneKernel kern = nullptr;
    bool kern_xl2 = false;
    if (policy_ok && (!rh || fs->policy == 6) && !probs_out && !attn_out_needs_probs(fs, attn_out)) {
      if (want_full || g_one_trace) kern = one_pick(HQ, H, S, D, dtype, kind, true, true, &kern_xl2);
      if (!kern && !want_full) kern = one_pick(HQ, H, S, D, dtype, kind, false, true, &kern_xl2);
    }
    const bool one_ok = kern != nullptr;
    if (one_asked && !one_ok) return CC_ERR_UNSUPPORTED;
    if (one_ok) {
      char* ob = reinterpret_cast<char*>(workspace);
      sa.one_hdr = reinterpret_cast<unsigned*>(ob);
      sa.one_ml = ob + kOneHdrBytes;
      sa.one_o = ob + kOneHdrBytes + kOneMlCap;
      sa.one_ml_bytes = (unsigned)kOneMlCap;
      sa.one_o_bytes = (unsigned)kOneOCap;
      sa.trace = reinterpret_cast<unsigned long long*>(g_one_trace);
      sa.y = y; sa.attn_out = attn_out; sa.hh_counter = hh_counter; sa.g = fs->g; sa.w = fs->w; sa.yc_chunks = p.n_chunks;
      sa.policy = fs->policy; sa.rand_next = fs->rand_next; sa.rng_seed = fs->rng_seed;
      if (fs->policy == 6) {  // hybrid: the ring state travels with the launch; every workgroup derives the ring column itself
        sa.ring_col = nullptr;
        if (rh) {
          sa.ring_num = rh->num; sa.ring_acc = reinterpret_cast<unsigned long long*>(rh->acc); sa.ring_wsum = rh->wsum;
        }
      }
#if CC_V_PRELOAD
      // (the XL2 instantiations take their (kv head, split) from a linear block index: a 1-D grid, the head count preloaded)
      sa.virt8 = (kern_xl2 && (H & 7) != 0) ? 1 : 0;
      const dim3 one_grid = kern_xl2 ? dim3(p.n_split * (sa.virt8 ? 8 : H), 1, 1) : dim3(p.n_split, H, 1);
#else
      const dim3 one_grid(p.n_split, H, 1);
      (void)kern_xl2;
#endif
      hipLaunchKernelGGL(kern, one_grid, dim3(p.nw * 64), 0, st, CC_LEAD_ARGS(sa) sa);
      CC_LAUNCH_CHECK();
      return CC_OK;
    }
  }
  if (p.nw == 16) return CC_ERR_UNSUPPORTED;  // (the 16-wave A/B geometry has no two-launch form)
  if (phases & 1) {
    switch (dtype) {
      case CC_DT_F32: rc = launch_split<float>(sa, p, H, R, D, st); break;
      case CC_DT_BF16: rc = launch_split<bf16_t>(sa, p, H, R, D, st); break;
      default: rc = launch_split<f16_t>(sa, p, H, R, D, st); break;
    }
  }
  if (rc != CC_OK) return rc;
  if (!(phases & 2)) return CC_OK;
  CombineArgs ca{};
  ca.scores = sa.scores; ca.part_ml = sa.part_ml; ca.part_o = sa.part_o;
  ca.y = y; ca.attn_out = attn_out; ca.probs_out = probs_out;
  ca.hh_num = hh_num; ca.hh_denom = hh_denom; ca.hh_counter = hh_counter;
  if (rh) {
    ca.ring_num = rh->num; ca.ring_col = sa.ring_col; ca.ring_W = rh->W;
    ca.ring_acc = reinterpret_cast<u64*>(rh->acc); ca.ring_wsum = rh->wsum;
  }
  ca.S = S; ca.R = R; ca.D = D; ca.n_split = p.n_split; ca.chunk = p.chunk;
  if (fs) {
    ca.next_key = fs->next_key; ca.input_pos = fs->input_pos; ca.pos = fs->c->pos; ca.H = H; ca.g = fs->g; ca.w = fs->w;
    ca.policy = fs->policy; ca.Hp = fs->c->Hp; ca.rand_next = fs->rand_next; ca.rng_seed = fs->rng_seed;
    ca.key_norm = sa.key_norm; ca.l2_pmax = sa.l2_pmax; ca.l2_new = sa.l2_new; ca.l2_np = H * p.n_split * p.nw;
    if (fs->policy == 6) {
      ca.hyb = sa.hyb;
      ca.cache_cts = fs->c->cache_cts;
    }
  }
  ca.abl = (phases >> 8) & 0xff;
  dim3 grid(p.n_chunks, H), block(kCombThreads);
  const size_t lds = (size_t)R * p.n_split * sizeof(float);  // <= 32 * 512 * 4 = 64 KiB
  switch (dtype) {
    case CC_DT_F32: hipLaunchKernelGGL(decode_attn_combine_kernel<float>, grid, block, lds, st, ca); break;
    case CC_DT_BF16: hipLaunchKernelGGL(decode_attn_combine_kernel<bf16_t>, grid, block, lds, st, ca); break;
    default: hipLaunchKernelGGL(decode_attn_combine_kernel<f16_t>, grid, block, lds, st, ca); break;
  }
  CC_LAUNCH_CHECK();
  // l2: the norm record the NEXT step's single-launch form starts from (cc_common.h, cc_l2_record) — the single-launch step leaves
  // it itself; behind the two launches a third, small one does (this is the slow route already)
  if (CC_V_L2CARRY != 0 && fs && fs->policy == 4 && fs->next_key)  // (only the r6 A/B build's single-launch step reads it)
    return cc_l2_record_launch(sa.key_norm, H, S, dtype, fs->input_pos, 0, fs->next_key, st);
  return CC_OK;
}

namespace {
// Measurement hook: the LAUNCH FLOOR of the layer step — a kernel with the step's grid, workgroup size and K/V access pattern
// (every lane's 16-byte non-temporal loads of its rows, all issued up front) and nothing else: no scores, no hand-off, no
// finish.  What it takes is what ANY stand-alone launch that streams this cache takes on this device (launch boundary + first
// byte + transfer); bench.py reports the step against it (roofline.launch_floor_us / frac_of_launch_floor).
template <int NW>
__global__ __launch_bounds__(NW * 64) void kv_stream_floor_kernel(const uint4* __restrict__ k, const uint4* __restrict__ v, int S,
                                                                  int rows_per_split, unsigned* out) {
  typedef unsigned int u32x4_nt __attribute__((ext_vector_type(4)));
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4, c = lane & 15;
  const int split = blockIdx.x, h = blockIdx.y;
  const int row_begin = split * rows_per_split, row_end = min(S, row_begin + rows_per_split);
  const uint4* kh = k + (size_t)h * S * 16;  // 16 x 16 B per 256-byte row (16-bit caches, head_dim 128)
  const uint4* vh = v + (size_t)h * S * 16;
  unsigned x = 0;
  for (int base = row_begin + wave * 16; base < row_end; base += NW * 16) {
    u32x4_nt r[8];
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const int row = base + 4 * g + u < row_end ? base + 4 * g + u : row_end - 1;
      r[u] = __builtin_nontemporal_load(reinterpret_cast<const u32x4_nt*>(kh + (size_t)row * 16 + c));
      r[4 + u] = __builtin_nontemporal_load(reinterpret_cast<const u32x4_nt*>(vh + (size_t)row * 16 + c));
    }
#pragma unroll
    for (int u = 0; u < 8; u++) x ^= r[u].x ^ r[u].y ^ r[u].z ^ r[u].w;
  }
  if (x == 0x9e3779b9u) out[0] = x;  // (never: keeps the loads alive)
}

}  // namespace

extern "C" {

int cc_decode_step_stream_floor_geom(const cc_kv_view* c, int32_t waves, int32_t rows_per_workgroup, void* scratch, cc_stream_t stream) {
  CC_ENTRY();
  if (!cc_view_ok(c) || !scratch || rows_per_workgroup <= 0 || rows_per_workgroup % (16 * waves)) return CC_ERR_BAD_ARG;
  if (cc_dt_size(c->dtype) != 2 || c->D != 128) return CC_ERR_UNSUPPORTED;
  const dim3 grid((c->S + rows_per_workgroup - 1) / rows_per_workgroup, c->H, 1), block(waves * 64);
  hipStream_t st = (hipStream_t)stream;
  const uint4 *k = reinterpret_cast<const uint4*>(c->k_cache), *v = reinterpret_cast<const uint4*>(c->v_cache);
  unsigned* out = reinterpret_cast<unsigned*>(scratch);
  switch (waves) {
    case 1: hipLaunchKernelGGL(kv_stream_floor_kernel<1>, grid, block, 0, st, k, v, c->S, rows_per_workgroup, out); break;
    case 2: hipLaunchKernelGGL(kv_stream_floor_kernel<2>, grid, block, 0, st, k, v, c->S, rows_per_workgroup, out); break;
    case 4: hipLaunchKernelGGL(kv_stream_floor_kernel<4>, grid, block, 0, st, k, v, c->S, rows_per_workgroup, out); break;
    case 8: hipLaunchKernelGGL(kv_stream_floor_kernel<8>, grid, block, 0, st, k, v, c->S, rows_per_workgroup, out); break;
    default: return CC_ERR_UNSUPPORTED;
  }
  CC_LAUNCH_CHECK();
  return CC_OK;
}

int cc_decode_step_stream_floor(const cc_kv_view* c, int32_t HQ, void* scratch, cc_stream_t stream) {
  CC_ENTRY();
  if (!cc_view_ok(c) || !scratch || HQ <= 0 || HQ % c->H) return CC_ERR_BAD_ARG;
  if (cc_dt_size(c->dtype) != 2 || c->D != 128) return CC_ERR_UNSUPPORTED;
  const Plan p = make_plan(HQ, c->H, c->S, c->D, c->dtype);
  const dim3 grid(p.n_split, c->H, 1), block(p.nw * 64);
  hipStream_t st = (hipStream_t)stream;
  if (p.nw == 8)
    hipLaunchKernelGGL(kv_stream_floor_kernel<8>, grid, block, 0, st, reinterpret_cast<const uint4*>(c->k_cache),
                       reinterpret_cast<const uint4*>(c->v_cache), c->S, p.rows_per_split, reinterpret_cast<unsigned*>(scratch));
  else
    hipLaunchKernelGGL(kv_stream_floor_kernel<4>, grid, block, 0, st, reinterpret_cast<const uint4*>(c->k_cache),
                       reinterpret_cast<const uint4*>(c->v_cache), c->S, p.rows_per_split, reinterpret_cast<unsigned*>(scratch));
  CC_LAUNCH_CHECK();
  return CC_OK;
}

int cc_decode_attn_gqa_phases(const void* q, const void* k, const void* v, const uint8_t* mask, int32_t HQ, int32_t H,
                              int32_t S, int32_t D, int32_t dtype, float scale, void* y, void* attn_out,
                              void* probs_out, double* hh_num, int32_t* hh_denom, int64_t* hh_counter,
                              void* workspace, size_t workspace_bytes, cc_stream_t stream, int32_t phases) {
  return attn_impl(q, k, v, mask, HQ, H, S, D, dtype, scale, y, attn_out, probs_out, hh_num, hh_denom, hh_counter, workspace,
                   workspace_bytes, stream, phases, nullptr);
}

int cc_decode_step_heavy_hitter(const cc_kv_view* c, const void* q, const void* k_new, const void* v_new,
                                const int32_t* input_pos, double* num, int32_t* denom, int64_t* counter,
                                uint64_t* next_key, int32_t global_tokens, int32_t recent_window, int32_t HQ, float scale,
                                void* y, void* attn_out, void* workspace, size_t workspace_bytes, cc_stream_t stream) {
  return cc_decode_step_heavy_hitter_phases(c, q, k_new, v_new, input_pos, num, denom, counter, next_key, global_tokens,
                                            recent_window, HQ, scale, y, attn_out, workspace, workspace_bytes, stream, 3);
}

int cc_decode_step_heavy_hitter_phases(const cc_kv_view* c, const void* q, const void* k_new, const void* v_new,
                                       const int32_t* input_pos, double* num, int32_t* denom, int64_t* counter,
                                       uint64_t* next_key, int32_t global_tokens, int32_t recent_window, int32_t HQ,
                                       float scale, void* y, void* attn_out, void* workspace, size_t workspace_bytes,
                                       cc_stream_t stream, int32_t phases) {
  if (!cc_view_ok(c) || !q || !k_new || !v_new || !input_pos || !num || !denom || !next_key || !y || c->Hp != c->H ||
      HQ <= 0 || HQ % c->H)
    return CC_ERR_BAD_ARG;
  FusedStep fs{c, k_new, v_new, input_pos, reinterpret_cast<unsigned long long*>(next_key), global_tokens, recent_window, 1, nullptr, nullptr};
  return attn_impl(q, c->k_cache, c->v_cache, c->mask, HQ, c->H, c->S, c->D, c->dtype, scale, y, attn_out, nullptr, num, denom,
                   counter, workspace, workspace_bytes, stream, phases, &fs);
}

int cc_decode_step_heavy_hitter_rc(const cc_kv_view* c, const void* q, const void* k_new, const void* v_new,
                                   const int32_t* input_pos, double* num, int32_t* denom, int64_t* counter,
                                   uint64_t* next_key, int32_t* step_commit, int32_t global_tokens, int32_t recent_window,
                                   int32_t HQ, float scale, void* y, void* workspace, size_t workspace_bytes,
                                   cc_stream_t stream, int32_t phases) {
  if (!cc_view_ok(c) || !q || !k_new || !v_new || !input_pos || !num || !denom || !next_key || !y || c->Hp != c->H ||
      HQ <= 0 || HQ % c->H)
    return CC_ERR_BAD_ARG;
  FusedStep fs{c, k_new, v_new, input_pos, reinterpret_cast<unsigned long long*>(next_key), global_tokens, recent_window, 1, nullptr, nullptr};
  fs.commit = step_commit;
  return attn_impl(q, c->k_cache, c->v_cache, c->mask, HQ, c->H, c->S, c->D, c->dtype, scale, y, nullptr, nullptr, num, denom,
                   counter, workspace, workspace_bytes, stream, phases, &fs);
}

int cc_decode_step_recent_global(const cc_kv_view* c, const void* q, const void* k_new, const void* v_new,
                                 const int32_t* input_pos, uint64_t* next_key, int32_t global_tokens, int32_t HQ, float scale,
                                 void* y, void* workspace, size_t workspace_bytes, cc_stream_t stream) {
  if (!cc_view_ok(c) || !q || !k_new || !v_new || !input_pos || !next_key || !y || c->Hp != 1 || HQ <= 0 || HQ % c->H ||
      global_tokens < 0 || global_tokens >= c->S)
    return CC_ERR_BAD_ARG;
  FusedStep fs{c, k_new, v_new, input_pos, reinterpret_cast<unsigned long long*>(next_key), global_tokens, 0, 2, nullptr, nullptr};
  return attn_impl(q, c->k_cache, c->v_cache, c->mask, HQ, c->H, c->S, c->D, c->dtype, scale, y, nullptr, nullptr, nullptr, nullptr,
                   nullptr, workspace, workspace_bytes, stream, 3, &fs);
}

int cc_decode_step_random(const cc_kv_view* c, const void* q, const void* k_new, const void* v_new, const int32_t* input_pos,
                          const float* rand_next, uint64_t* next_key, int32_t global_tokens, int32_t recent_window, int32_t HQ,
                          float scale, void* y, void* workspace, size_t workspace_bytes, cc_stream_t stream) {
  if (!cc_view_ok(c) || !q || !k_new || !v_new || !input_pos || !rand_next || !next_key || !y || c->Hp != 1 || HQ <= 0 ||
      HQ % c->H || global_tokens < 0)
    return CC_ERR_BAD_ARG;
  FusedStep fs{c, k_new, v_new, input_pos, reinterpret_cast<unsigned long long*>(next_key), global_tokens, recent_window, 3, rand_next, nullptr};
  return attn_impl(q, c->k_cache, c->v_cache, c->mask, HQ, c->H, c->S, c->D, c->dtype, scale, y, nullptr, nullptr, nullptr, nullptr,
                   nullptr, workspace, workspace_bytes, stream, 3, &fs);
}

int cc_decode_step_random_rng(const cc_kv_view* c, const void* q, const void* k_new, const void* v_new, const int32_t* input_pos,
                              uint64_t seed, uint64_t* next_key, int32_t global_tokens, int32_t recent_window, int32_t HQ,
                              float scale, void* y, void* workspace, size_t workspace_bytes, cc_stream_t stream) {
  if (!cc_view_ok(c) || !q || !k_new || !v_new || !input_pos || !next_key || !y || c->Hp != 1 || HQ <= 0 || HQ % c->H ||
      global_tokens < 0)
    return CC_ERR_BAD_ARG;
  FusedStep fs{c, k_new, v_new, input_pos, reinterpret_cast<unsigned long long*>(next_key), global_tokens, recent_window, 3, nullptr, nullptr};
  fs.rng_seed = seed;
  fs.rng_on = 1;
  return attn_impl(q, c->k_cache, c->v_cache, c->mask, HQ, c->H, c->S, c->D, c->dtype, scale, y, nullptr, nullptr, nullptr, nullptr,
                   nullptr, workspace, workspace_bytes, stream, 3, &fs);
}

int cc_decode_step_head_constant_rc(const cc_kv_view* c, int32_t policy, const void* q, const void* k_new, const void* v_new,
                                    const int32_t* input_pos, const float* rand_next, uint64_t seed, uint64_t* next_key,
                                    int32_t* step_commit, int32_t global_tokens, int32_t recent_window, int32_t HQ, float scale,
                                    void* y, void* workspace, size_t workspace_bytes, cc_stream_t stream) {
  if (!cc_view_ok(c) || !q || !k_new || !v_new || !input_pos || !next_key || !y || c->Hp != 1 || HQ <= 0 || HQ % c->H ||
      global_tokens < 0 || (policy != 2 && policy != 3) || (policy == 2 && (global_tokens >= c->S || rand_next)))
    return CC_ERR_BAD_ARG;
  FusedStep fs{c, k_new, v_new, input_pos, reinterpret_cast<unsigned long long*>(next_key), global_tokens,
               policy == 3 ? recent_window : 0, policy, policy == 3 ? rand_next : nullptr, nullptr};
  if (policy == 3 && !rand_next) {
    fs.rng_seed = seed;
    fs.rng_on = 1;
  }
  fs.commit = step_commit;
  return attn_impl(q, c->k_cache, c->v_cache, c->mask, HQ, c->H, c->S, c->D, c->dtype, scale, y, nullptr, nullptr, nullptr, nullptr,
                   nullptr, workspace, workspace_bytes, stream, 3, &fs);
}

/* The layer step with the layer's QKV projection folded in (r5): see include/coldcompress.h. */
int cc_decode_step_qkv_rc(const cc_kv_view* c, int32_t policy, const void* wqkv, const void* bias, const void* x, const void* delta,
                          const void* norm_w, float eps, void* h_out, const void* freqs, int32_t K, void* qkv_out,
                          const int32_t* input_pos, double* num, int32_t* denom, int64_t* counter, const float* rand_next,
                          uint64_t seed, uint64_t* next_key, int32_t* step_commit, int32_t global_tokens, int32_t recent_window,
                          int32_t HQ, float scale, void* y, void* workspace, size_t workspace_bytes, cc_stream_t stream) {
  if (!cc_view_ok(c) || !wqkv || !x || !norm_w || !input_pos || !next_key || !y || HQ <= 0 || HQ % c->H || global_tokens < 0 || K <= 0)
    return CC_ERR_BAD_ARG;
  switch (policy) {
    case 1: if (!num || !denom || c->Hp != c->H || rand_next) return CC_ERR_BAD_ARG; break;
    case 2: if (num || denom || c->Hp != 1 || global_tokens >= c->S || rand_next) return CC_ERR_BAD_ARG; break;
    case 3: if (num || denom || c->Hp != 1) return CC_ERR_BAD_ARG; break;
    default: return CC_ERR_BAD_ARG;
  }
  QkvIn qi{};
  qi.W = wqkv; qi.bias = bias; qi.x = x; qi.delta = delta; qi.norm_w = norm_w; qi.freqs = freqs; qi.h_out = h_out; qi.qkv_out = qkv_out;
  qi.eps = eps; qi.K = K;
  FusedStep fs{c, nullptr, nullptr, input_pos, reinterpret_cast<unsigned long long*>(next_key), global_tokens,
               policy == 2 ? 0 : recent_window, policy, policy == 3 ? rand_next : nullptr, nullptr};
  if (policy == 3 && !rand_next) {
    fs.rng_seed = seed;
    fs.rng_on = 1;
  }
  fs.commit = step_commit;
  fs.qkv = &qi;
  return attn_impl(nullptr, c->k_cache, c->v_cache, c->mask, HQ, c->H, c->S, c->D, c->dtype, scale, y, nullptr, nullptr, num, denom,
                   policy == 1 ? counter : nullptr, workspace, workspace_bytes, stream, 3, &fs);
}

static int decode_step_quant_impl(const cc_kv_view* c, float* qparams, int32_t n_bit, int32_t policy, const void* q, const void* k_new,
                                  const void* v_new, const int32_t* input_pos, double* num, int32_t* denom, int64_t* counter,
                                  const float* rand_next, bool rng_on, uint64_t seed, uint64_t* next_key, int32_t* step_commit,
                                  int32_t global_tokens, int32_t recent_window, int32_t HQ, float scale, void* y, void* attn_out,
                                  void* workspace, size_t workspace_bytes, cc_stream_t stream, int32_t phases) {
  if (!cc_view_ok(c) || !qparams || !q || !k_new || !v_new || !input_pos || !next_key || !y || HQ <= 0 || HQ % c->H ||
      global_tokens < 0)
    return CC_ERR_BAD_ARG;
  if (n_bit != 8) return CC_ERR_UNSUPPORTED;
  switch (policy) {
    case 1: if (!num || !denom || c->Hp != c->H) return CC_ERR_BAD_ARG; break;
    case 2: if (num || denom || c->Hp != 1 || global_tokens >= c->S) return CC_ERR_BAD_ARG; break;
    case 3: if (num || denom || c->Hp != 1 || (!rand_next && !rng_on)) return CC_ERR_BAD_ARG; break;
    default: return CC_ERR_UNSUPPORTED;
  }
  FusedStep fs{c, k_new, v_new, input_pos, reinterpret_cast<unsigned long long*>(next_key), global_tokens,
               policy == 2 ? 0 : recent_window, policy, policy == 3 ? rand_next : nullptr, nullptr, nullptr, qparams};
  if (policy == 3 && !rand_next) {
    fs.rng_seed = seed;
    fs.rng_on = 1;
  }
  fs.commit = step_commit;
  return attn_impl(q, c->k_cache, c->v_cache, c->mask, HQ, c->H, c->S, c->D, c->dtype, scale, y, policy == 1 ? attn_out : nullptr,
                   nullptr, num, denom, policy == 1 ? counter : nullptr, workspace, workspace_bytes, stream, phases, &fs);
}
int cc_decode_step_quant(const cc_kv_view* c, float* qparams, int32_t n_bit, int32_t policy, const void* q, const void* k_new,
                         const void* v_new, const int32_t* input_pos, double* num, int32_t* denom, int64_t* counter,
                         const float* rand_next, uint64_t* next_key, int32_t global_tokens, int32_t recent_window, int32_t HQ,
                         float scale, void* y, void* attn_out, void* workspace, size_t workspace_bytes, cc_stream_t stream,
                         int32_t phases) {
  return decode_step_quant_impl(c, qparams, n_bit, policy, q, k_new, v_new, input_pos, num, denom, counter, rand_next, false, 0, next_key,
                                nullptr, global_tokens, recent_window, HQ, scale, y, attn_out, workspace, workspace_bytes, stream, phases);
}
int cc_decode_step_quant_rc(const cc_kv_view* c, float* qparams, int32_t n_bit, int32_t policy, const void* q, const void* k_new,
                            const void* v_new, const int32_t* input_pos, double* num, int32_t* denom, int64_t* counter,
                            const float* rand_next, uint64_t seed, uint64_t* next_key, int32_t* step_commit, int32_t global_tokens,
                            int32_t recent_window, int32_t HQ, float scale, void* y, void* workspace, size_t workspace_bytes,
                            cc_stream_t stream, int32_t phases) {
  return decode_step_quant_impl(c, qparams, n_bit, policy, q, k_new, v_new, input_pos, num, denom, counter, rand_next, rand_next == nullptr,
                                seed, next_key, step_commit, global_tokens, recent_window, HQ, scale, y, nullptr, workspace, workspace_bytes,
                                stream, phases);
}
int cc_decode_step_l2_rc(const cc_kv_view* c, const void* q, const void* k_new, const void* v_new, const int32_t* input_pos,
                         void* key_norm, uint64_t* next_key, int32_t* step_commit, int32_t global_tokens, int32_t recent_window, int32_t HQ,
                         float scale, void* y, void* workspace, size_t workspace_bytes, cc_stream_t stream) {
  if (!cc_view_ok(c) || !q || !k_new || !v_new || !input_pos || !key_norm || !next_key || !y || c->Hp != c->H || HQ <= 0 ||
      HQ % c->H || global_tokens < 0)
    return CC_ERR_BAD_ARG;
  FusedStep fs{c, k_new, v_new, input_pos, reinterpret_cast<unsigned long long*>(next_key), global_tokens, recent_window, 4, nullptr,
               key_norm};
  fs.commit = step_commit;
  return attn_impl(q, c->k_cache, c->v_cache, c->mask, HQ, c->H, c->S, c->D, c->dtype, scale, y, nullptr, nullptr, nullptr, nullptr,
                   nullptr, workspace, workspace_bytes, stream, 3, &fs);
}
int cc_decode_step_l2(const cc_kv_view* c, const void* q, const void* k_new, const void* v_new, const int32_t* input_pos,
                      void* key_norm, uint64_t* next_key, int32_t global_tokens, int32_t recent_window, int32_t HQ, float scale,
                      void* y, void* workspace, size_t workspace_bytes, cc_stream_t stream) {
  return cc_decode_step_l2_rc(c, q, k_new, v_new, input_pos, key_norm, next_key, nullptr, global_tokens, recent_window, HQ, scale, y,
                              workspace, workspace_bytes, stream);
}
int cc_decode_step_heavy_hitter_ring(const cc_kv_view* c, const void* q, const void* k_new, const void* v_new, const int32_t* input_pos,
                                     void* ring_num, int32_t* denom, int64_t* counter, int32_t W, uint64_t* wsum_acc, float* wsum,
                                     uint64_t* next_key, int32_t global_tokens, int32_t recent_window, int32_t HQ, float scale, void* y,
                                     void* attn_out, void* workspace, size_t workspace_bytes, cc_stream_t stream) {
  if (!cc_view_ok(c) || !q || !k_new || !v_new || !input_pos || !next_key || !y || c->Hp != c->H || HQ <= 0 || HQ % c->H)
    return CC_ERR_BAD_ARG;
  FusedStep fs{c, k_new, v_new, input_pos, reinterpret_cast<unsigned long long*>(next_key), global_tokens, recent_window, 5, nullptr,
               nullptr};
  RingHistory rh{ring_num, W, wsum_acc, wsum};
  return attn_impl(q, c->k_cache, c->v_cache, c->mask, HQ, c->H, c->S, c->D, c->dtype, scale, y, attn_out, nullptr, nullptr, denom, counter,
                   workspace, workspace_bytes, stream, 3, &fs, &rh);
}

int cc_decode_step_hybrid_rc(const cc_kv_view* c, const void* q, const void* k_new, const void* v_new, const int32_t* input_pos,
                             const int64_t* strategies, const int32_t* policy_table, int32_t n_policies, void* ring_num,
                             int32_t* denom, int64_t* counter, int32_t W, uint64_t* wsum_acc, float* wsum, const uint8_t* special_mask,
                             uint8_t* punc_mask, const int64_t* token_id, const int64_t* punc_ids, int32_t n_punc_ids,
                             const int32_t* num_special, int32_t* num_punc, uint64_t* next_key, int32_t* step_commit,
                             int32_t global_tokens, int32_t HQ, float scale, void* y, void* attn_out, void* workspace,
                             size_t workspace_bytes, cc_stream_t stream) {
  if (!cc_view_ok(c) || !q || !k_new || !v_new || !input_pos || !strategies || !policy_table || n_policies <= 0 || !next_key || !y ||
      c->Hp != c->H || c->Hc != c->H || HQ <= 0 || HQ % c->H || W <= 0 || (punc_ids && n_punc_ids < 0) ||
      (ring_num && (!denom || !counter || !wsum_acc || !wsum)))
    return CC_ERR_BAD_ARG;
  HybridStep hs{};
  hs.strategies = strategies; hs.table = policy_table; hs.special_mask = special_mask; hs.punc_mask = punc_mask;
  hs.token_id = token_id; hs.punc_ids = punc_ids; hs.n_punc_ids = n_punc_ids; hs.num_special = num_special; hs.num_punc = num_punc;
  hs.W = W; hs.n_pol = n_policies;
  if (n_policies * 3 > 64) return CC_ERR_UNSUPPORTED;  // the streaming pass fetches the whole policy table with one vector load
  FusedStep fs{c, k_new, v_new, input_pos, reinterpret_cast<unsigned long long*>(next_key), global_tokens, 0, 6, nullptr, nullptr, &hs};
  fs.commit = step_commit;
  RingHistory rh{ring_num, W, wsum_acc, wsum};
  return attn_impl(q, c->k_cache, c->v_cache, c->mask, HQ, c->H, c->S, c->D, c->dtype, scale, y, attn_out, nullptr, nullptr,
                   ring_num ? denom : nullptr, ring_num ? counter : nullptr, workspace, workspace_bytes, stream, 3, &fs,
                   ring_num ? &rh : nullptr);
}

int cc_decode_step_hybrid(const cc_kv_view* c, const void* q, const void* k_new, const void* v_new, const int32_t* input_pos,
                          const int64_t* strategies, const int32_t* policy_table, int32_t n_policies, void* ring_num, int32_t* denom,
                          int64_t* counter, int32_t W, uint64_t* wsum_acc, float* wsum, const uint8_t* special_mask,
                          uint8_t* punc_mask, const int64_t* token_id, const int64_t* punc_ids, int32_t n_punc_ids,
                          const int32_t* num_special, int32_t* num_punc, uint64_t* next_key, int32_t global_tokens, int32_t HQ,
                          float scale, void* y, void* attn_out, void* workspace, size_t workspace_bytes, cc_stream_t stream) {
  return cc_decode_step_hybrid_rc(c, q, k_new, v_new, input_pos, strategies, policy_table, n_policies, ring_num, denom, counter, W,
                                  wsum_acc, wsum, special_mask, punc_mask, token_id, punc_ids, n_punc_ids, num_special, num_punc,
                                  next_key, nullptr, global_tokens, HQ, scale, y, attn_out, workspace, workspace_bytes, stream);
}

int cc_decode_attn_gqa_ring(const void* q, const void* k, const void* v, const uint8_t* mask, int32_t HQ, int32_t H, int32_t S,
                            int32_t D, int32_t dtype, float scale, void* y, void* attn_out, void* ring_num, int32_t* denom,
                            int64_t* counter, int32_t W, uint64_t* wsum_acc, float* wsum, void* workspace,
                            size_t workspace_bytes, cc_stream_t stream) {
  RingHistory rh{ring_num, W, wsum_acc, wsum};
  return attn_impl(q, k, v, mask, HQ, H, S, D, dtype, scale, y, attn_out, nullptr, nullptr, denom, counter, workspace,
                   workspace_bytes, stream, 3, nullptr, &rh);
}

int cc_decode_attn_gqa(const void* q, const void* k, const void* v, const uint8_t* mask, int32_t HQ, int32_t H,
                       int32_t S, int32_t D, int32_t dtype, float scale, void* y, void* attn_out, void* probs_out,
                       double* hh_num, int32_t* hh_denom, int64_t* hh_counter, void* workspace,
                       size_t workspace_bytes, cc_stream_t stream) {
  return cc_decode_attn_gqa_phases(q, k, v, mask, HQ, H, S, D, dtype, scale, y, attn_out, probs_out, hh_num, hh_denom,
                                   hh_counter, workspace, workspace_bytes, stream, 3);
}

}  // extern "C"
