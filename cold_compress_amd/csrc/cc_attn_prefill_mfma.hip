// cc_attn_prefill_mfma.hip — causal prefill attention on the gfx950 matrix cores (v_mfma_f32_32x32x16_{bf16,f16}),
// with the side outputs the eviction policies need, for the Llama-3 geometry (16-bit dtype, D = 128, R = HQ/H = 4).
// Everything else falls back to the LDS-tiled VALU kernels in cc_attn_prefill.hip.
//
// ref: attention_utils.py:36-54 (scores -> dtype, *scale -> dtype, softmax -> dtype, P@V), model.py:413-418
// (group mean), cache.py:704 / prompt_compression.py:170-194 (column sums, observation window).
//
// Structure (two passes, like the VALU path, because the reference's probabilities are rounded to the model dtype
// AFTER normalisation with the final row statistics):
//   workgroup = 4 waves = the 4 query heads of one kv head x one tile of 32 queries; persistent over query tiles
//   (w, w + NWG, ...) so the per-workgroup column-sum partials accumulate in a fixed order (no atomics).
//   K (and V^T) tiles of 32 keys are fetched ONCE per workgroup — the four heads share them — one tile ahead of the
//   math, into double-buffered XOR-swizzled LDS images (conflict-free ds_read_b128 fragment reads).
//   S^T = K . Q^T  ("swapped" product): the MFMA C tile has col = query (lane & 31) and 16 keys per lane, so every
//   softmax row statistic is lane-local + one half-wave exchange.
//   P (C layout) is converted to 16-bit and used DIRECTLY as the A operand of the PV MFMA; the B operand comes
//   from V^T stored with the matching key permutation inside each 32-key block (vt_perm), one 16-byte load per
//   operand — no LDS transpose, no cross-lane shuffles on the MFMA path.
// MFMA-bound contraction; HBM traffic is negligible (K/V tiles are re-read from L2 by the 4 waves).
#include <stdio.h>
#include <stdlib.h>

#include "cc_common.h"

#ifndef CC_PF_WPE_PV
#define CC_PF_WPE_PV 2     // waves per SIMD the compiler must leave room for (r6 A/B: 3 = 168 VGPRs)
#endif
#ifndef CC_PF_WPE_FLASH
#define CC_PF_WPE_FLASH 2
#endif

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <typename T>
struct MfmaOps;
template <>
struct MfmaOps<bf16_t> {
  __device__ static __forceinline__ f32x16 mma(uint4 a, uint4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
  }
  __device__ static __forceinline__ uint32_t pack2(float lo, float hi) {
    return (uint32_t)f32_to_bf16_bits(lo) | ((uint32_t)f32_to_bf16_bits(hi) << 16);
  }
};
template <>
struct MfmaOps<f16_t> {
  __device__ static __forceinline__ f32x16 mma(uint4 a, uint4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  }
  __device__ static __forceinline__ uint32_t pack2(float lo, float hi) {
    return (uint32_t)f32_to_f16_bits(lo) | ((uint32_t)f32_to_f16_bits(hi) << 16);
  }
};

// e^x through v_exp_f32 (2^x): relative error ~1e-6, far below the 2^-9 rounding the probabilities get next.  The
// accurate expf (~20 VALU ops) and an IEEE divide per probability cost 0.7 ms of the 3.1 ms second pass at L = 8192.
// (Double-buffering the K / V fragments across key tiles was tried and is SLOWER: 195 VGPRs, one wave per SIMD.)
__device__ __forceinline__ float pf_exp(float x) { return __builtin_amdgcn_exp2f(x * 1.4426950408889634f); }
constexpr float kLog2e = 1.4426950408889634f;

// dtype rounding of two values at once; returns the packed 16-bit pair (lo | hi << 16)
template <typename T>
__device__ __forceinline__ uint32_t pf_rnd2(float a, float b, float& ra, float& rb) {
  if constexpr (sizeof(T) == 2 && ElemTraits<T>::code == CC_DT_BF16) {
    return bf16_round_pair(a, b, ra, rb);
  } else {
    ra = ElemTraits<T>::rnd(a);
    rb = ElemTraits<T>::rnd(b);
    return MfmaOps<T>::pack2(ra, rb);
  }
}

constexpr int kD = 128;
constexpr int kTQ = 32;   // queries per tile
constexpr int kTK = 32;   // keys per tile
constexpr int kMaxBandsM = 4;

struct MArgs {
  const void* q;   // [HQ, L, D]
  const void* k;   // [H, L, D]
  const void* vt;  // [H, D, Lp] transposed + permuted V (Lp = L rounded up to 32)
  void* y;         // [HQ, L, D]
  float* stats;    // [HQ, L, 2]
  float* cpart;    // [1 + nb (+1), NWG, H, L]: column sums, band sums, observation-window sums
  int H, L, Lp, nb;
  int nwg;         // persistent workgroups per kv head (the grid is 1-D: nwg * H)
  int xcd_remap;   // H % 8 == 0: all workgroups of kv head h sit on XCD h % 8 (workgroup b is dispatched to XCD b % 8 —
                   // observed, used for speed only), so that one head's K / V^T prefix (2 * L * 256 B) is served by ONE 4 MiB L2
                   // instead of every head's by all eight
  int obs_len;     // > 0: plane 1 + nb accumulates the group-mean probabilities of the last obs_len query rows
  int band[kMaxBandsM];
  float scale;
  unsigned long long* trace;  // measurement only (CC_PREFILL_TRACE): per (kv head, wave) cycle sums of the tile phases, workgroup 0
};


// 1-D grid -> (persistent workgroup index within the head, kv head)
__device__ __forceinline__ void wg_coords(const MArgs& a, int& bx, int& h) {
  const int b = blockIdx.x;
  if (a.xcd_remap) {
    const int idx = b >> 3;
    h = (b & 7) + 8 * (idx / a.nwg);
    bx = idx % a.nwg;
  } else {
    h = b / a.nwg;
    bx = b % a.nwg;
  }
}

// Round i of a persistent workgroup's loop over query tiles: serpentine over the workgroups (0 .. nwg-1, nwg-1 .. 0, ...), so
// that every workgroup gets the same share of the causal triangle (tile qt costs qt + 1 key tiles; with the plain stride
// w, w + nwg, ... the last workgroup of a head had 25 % more work than the average at L = 8192 and the launch ended on it).
// The assignment is static: the partial planes still accumulate in a fixed order.
__device__ __forceinline__ int q_tile(int i, int bx, int nwg) { return i * nwg + ((i & 1) ? nwg - 1 - bx : bx); }

// key offset inside a 32-key tile held by accumulator register `reg` of a lane in half `hi`
__device__ __forceinline__ int c_row(int reg, int hi) { return (reg & 3) + 8 * (reg >> 2) + 4 * hi; }

// V [H, L, D] -> vt_perm [H, D, Lp]: position pp = kb*16 + 8*hi + i of each 32-key block holds key c_row(kb*8 + i, hi)
template <typename T>
__global__ __launch_bounds__(256) void vt_perm_kernel(const T* v, T* vt, int H, int L, int Lp) {
  const size_t total = (size_t)H * kD * Lp;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
    const int pp_abs = (int)(idx % Lp);
    const int d = (int)((idx / Lp) % kD);
    const int h = (int)(idx / ((size_t)Lp * kD));
    const int blk = pp_abs >> 5, pp = pp_abs & 31;
    const int kb = pp >> 4, hi = (pp >> 3) & 1, i = pp & 7;
    const int key = blk * 32 + c_row(kb * 8 + i, hi);
    T val;
    val.x = 0;
    if (key < L) val = v[((size_t)h * L + key) * kD + d];
    vt[idx] = val;
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Pass 1 (row maxima and sums), LDS-shared K tiles.  PMC counters of the first, per-wave-loading version: 65 % of the
// wave cycles parked in s_waitcnt — every wave fetched its own K fragments per 32-key tile (the four query heads of
// a group fetched the SAME tile four times) and waited out an L2 round trip per tile.  Here the workgroup fetches a
// K tile ONCE (two 16-byte loads per thread, coalesced rows), one tile ahead of the math, into a double-buffered LDS
// image whose 16-byte chunks are XOR-swizzled by (key & 15) so that the MFMA A-fragment reads (ds_read_b128, four
// non-contiguous 16-lane groups) are bank-conflict free; one barrier per tile.
template <typename T>
__global__ __launch_bounds__(256) void prefill_stats_lds_kernel(MArgs a) {
  __shared__ __attribute__((aligned(16))) uint4 sm_kt[2][kTK][16];
  const int lane = threadIdx.x & 63, r = threadIdx.x >> 6;
  const int hi = lane >> 5, lq = lane & 31;
  int bx, h;
  wg_coords(a, bx, h);
  const int L = a.L;
  const int j = h * 4 + r;
  const T* qh = reinterpret_cast<const T*>(a.q) + (size_t)j * L * kD;
  const T* kh = reinterpret_cast<const T*>(a.k) + (size_t)h * L * kD;
  const int nqt = (L + kTQ - 1) / kTQ;
  // cooperative tile load: thread t -> key row t / 8, chunks 2 * (t % 8) and + 1 (32 contiguous bytes)
  const int ld_row = threadIdx.x >> 3, ld_c0 = (threadIdx.x & 7) * 2;

  for (int i = 0; i * a.nwg < nqt; i++) {
    const int qt = q_tile(i, bx, a.nwg);
    if (qt >= nqt) continue;  // ragged last round (uniform over the workgroup)
    const int q0 = qt * kTQ;
    const int query = q0 + lq;
    const int qc = query < L ? query : L - 1;
    uint4 qb[8];
#pragma unroll
    for (int ds = 0; ds < 8; ds++) qb[ds] = *reinterpret_cast<const uint4*>(qh + (size_t)qc * kD + ds * 16 + 8 * hi);
    float m_run = -INFINITY, l_run = 0.f;
    const int last_q = min(L, q0 + kTQ) - 1;
    const int ntile = last_q / kTK + 1;
    uint4 st0, st1;
    auto fetch = [&](int t) {
      const int krow = min(t * kTK + ld_row, L - 1);
      const uint4* src = reinterpret_cast<const uint4*>(kh + (size_t)krow * kD) + ld_c0;
      st0 = src[0];
      st1 = src[1];
    };
    auto stash = [&](int buf) {
      sm_kt[buf][ld_row][(ld_c0 ^ (ld_row & 15)) & 15] = st0;
      sm_kt[buf][ld_row][((ld_c0 + 1) ^ (ld_row & 15)) & 15] = st1;
    };
    __syncthreads();  // the previous query tile's last readers are done with both buffers
    fetch(0);
    stash(0);
    __syncthreads();
    for (int t = 0; t < ntile; t++) {
      const int k0 = t * kTK;
      if (t + 1 < ntile) fetch(t + 1);  // in flight during this tile's MFMAs and softmax
      f32x16 s;
#pragma unroll
      for (int e = 0; e < 16; e++) s[e] = 0.f;
#pragma unroll
      for (int ds = 0; ds < 8; ds++) s = MfmaOps<T>::mma(sm_kt[t & 1][lq][((2 * ds + hi) ^ (lq & 15)) & 15], qb[ds], s);
      float mx = m_run;
      float x[16];
      // every tile strictly below the diagonal of a complete query tile needs no causal / bounds masking: the 16
      // compares + selects per lane are only paid on the diagonal tile (and on the ragged last query tile)
      const bool full_tile = (k0 + kTK - 1 <= q0) && (q0 + kTQ <= L);
#pragma unroll
      for (int e = 0; e < 16; e += 2) {  // dtype(dtype(q.k) * scale), two elements per conversion
        float r0, r1;
        pf_rnd2<T>(s[e], s[e + 1], r0, r1);
        pf_rnd2<T>(r0 * a.scale, r1 * a.scale, x[e], x[e + 1]);
      }
      if (!full_tile) {
#pragma unroll
        for (int e = 0; e < 16; e++) {
          const int key = k0 + c_row(e, hi);
          if (key > query || key >= L || query >= L) x[e] = -INFINITY;
        }
      }
#pragma unroll
      for (int e = 0; e < 16; e++) mx = fmaxf(mx, x[e]);
      const float mu = (mx == -INFINITY) ? 0.f : mx;
      float sum = l_run * pf_exp(m_run - mu);
#pragma unroll
      for (int e = 0; e < 16; e++) sum += pf_exp(x[e] - mu);
      m_run = mx;
      l_run = sum;
      if (t + 1 < ntile) stash((t + 1) & 1);
      __syncthreads();
    }
    const float m_o = __shfl_xor(m_run, 32, CC_WAVE), l_o = __shfl_xor(l_run, 32, CC_WAVE);
    const float mx = fmaxf(m_run, m_o);
    const float mu = (mx == -INFINITY) ? 0.f : mx;
    const float lt = l_run * pf_exp(m_run - mu) + l_o * pf_exp(m_o - mu);
    if (hi == 0 && query < L) {
      a.stats[((size_t)j * L + query) * 2] = mx;
      a.stats[((size_t)j * L + query) * 2 + 1] = lt;
    }
  }
}


// ---------------------------------------------------------------------------------------------------------------
// Pass 2 (probabilities -> P.V, group-mean column / band / observation-window sums) with the same LDS pipeline as
// pass 1: K and V^T tiles are fetched once per workgroup, one tile ahead, into double-buffered swizzled LDS images;
// the partial side sums of tile t are folded into the workgroup's planes during tile t + 1 (double-buffered
// sm_red), so a tile costs TWO barriers instead of three and no wave waits on its own global loads.
//
// The tile body is compiled four times — {below the diagonal, on it} x {observation-window rows, none} — and the number
// of band planes is a template parameter (NB = -1: run-time a.nb): the loop was ISSUE-bound on the predicate / select /
// branch instructions of the causal mask, the band tests and the plane loops (~220 of ~690 instructions per 32 x 32
// tile, against 16 MFMAs), all of which are constants for every tile but the diagonal one.
template <bool B>
struct BoolC {
  static constexpr bool value = B;
};

template <typename T, int NB>
__global__ __launch_bounds__(256, CC_PF_WPE_PV) void prefill_pv_lds_kernel(MArgs a) {
  __shared__ __attribute__((aligned(16))) uint4 sm_kt[2][kTK][16];   // [buf][key][chunk ^ (key & 15)]
  __shared__ __attribute__((aligned(16))) uint4 sm_vt[2][kD][4];     // [buf][d][chunk ^ ((d >> 2) & 3)]  (V^T, permuted keys)
  __shared__ float sm_p[4][kTK][kTQ + 1];                            // per-wave probability tiles: [r][key][query]
  constexpr int NBC = NB >= 0 ? NB : kMaxBandsM;  // band planes the compiled code iterates over
  __shared__ float sm_red[2][2 + NBC][8][kTK];     // 52.5 KiB of LDS in all without band planes: three workgroups per CU
  const int nb = NB >= 0 ? NB : a.nb;
  const int lane = threadIdx.x & 63, r = threadIdx.x >> 6;
  const int hi = lane >> 5, lq = lane & 31;
  int bx, h;
  wg_coords(a, bx, h);
  const int L = a.L;
  const int j = h * 4 + r;
  const T* qh = reinterpret_cast<const T*>(a.q) + (size_t)j * L * kD;
  const T* kh = reinterpret_cast<const T*>(a.k) + (size_t)h * L * kD;
  const T* vth = reinterpret_cast<const T*>(a.vt) + (size_t)h * kD * a.Lp;
  const int nqt = (L + kTQ - 1) / kTQ;
  const size_t plane = (size_t)a.nwg * a.H * L;
  float* cp = a.cpart + ((size_t)bx * a.H + h) * L;
  const int obs_pl = 1 + nb;
  const int npl = 1 + nb + (a.obs_len > 0 ? 1 : 0);
  for (int pl = 0; pl < npl; pl++)
    for (int s = threadIdx.x; s < L; s += 256) cp[pl * plane + s] = 0.f;
  const int kl_row = threadIdx.x >> 3, kl_c0 = (threadIdx.x & 7) * 2;   // K tile: row t/8, chunks 2(t%8), +1
  const int vl_row = threadIdx.x >> 1, vl_c0 = (threadIdx.x & 1) * 2;   // V^T tile: d row t/2, chunks 2(t%2), +1
  const int rd_key = threadIdx.x & 31, rd_qs = threadIdx.x >> 5;        // side sums: key, slice of 4 queries
  // phase clock of workgroup 0 (measurement only): acc[k] += cycles since the previous stamp
  const bool tracing = a.trace != nullptr && bx == 0;
  unsigned long long tr_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tr_last = 0;
  auto stamp = [&](int k) {
    if (tracing) {
      const unsigned long long now = __builtin_amdgcn_s_memtime();
      tr_acc[k] += now - tr_last;
      tr_last = now;
    }
  };
  if (tracing) tr_last = __builtin_amdgcn_s_memtime();

  for (int i = 0; i * a.nwg < nqt; i++) {
    const int qt = q_tile(i, bx, a.nwg);
    if (qt >= nqt) continue;  // ragged last round (uniform over the workgroup)
    const int q0 = qt * kTQ;
    const bool obs_tile = a.obs_len > 0 && q0 + kTQ > L - a.obs_len;
    const int query = q0 + lq;
    const int qc = query < L ? query : L - 1;
    uint4 qb[8];
#pragma unroll
    for (int ds = 0; ds < 8; ds++) qb[ds] = *reinterpret_cast<const uint4*>(qh + (size_t)qc * kD + ds * 16 + 8 * hi);
    const float m_fin_l2 = -a.stats[((size_t)j * L + qc) * 2] * kLog2e;
    const float inv_l = __frcp_rn(a.stats[((size_t)j * L + qc) * 2 + 1]);
    f32x16 o[4];
#pragma unroll
    for (int b = 0; b < 4; b++)
#pragma unroll
      for (int e = 0; e < 16; e++) o[b][e] = 0.f;
    const int last_q = min(L, q0 + kTQ) - 1;
    const int ntile = last_q / kTK + 1;
    uint4 sk0, sk1, sv0, sv1;
    auto fetch = [&](int t) {
      const int krow = min(t * kTK + kl_row, L - 1);
      const uint4* ks = reinterpret_cast<const uint4*>(kh + (size_t)krow * kD) + kl_c0;
      sk0 = ks[0];
      sk1 = ks[1];
      const uint4* vs = reinterpret_cast<const uint4*>(vth + (size_t)vl_row * a.Lp + t * kTK) + vl_c0;
      sv0 = vs[0];
      sv1 = vs[1];
    };
    auto stash = [&](int buf) {
      sm_kt[buf][kl_row][(kl_c0 ^ (kl_row & 15)) & 15] = sk0;
      sm_kt[buf][kl_row][((kl_c0 + 1) ^ (kl_row & 15)) & 15] = sk1;
      sm_vt[buf][vl_row][(vl_c0 ^ ((vl_row >> 2) & 3)) & 3] = sv0;
      sm_vt[buf][vl_row][((vl_c0 + 1) ^ ((vl_row >> 2) & 3)) & 3] = sv1;
    };
    // Fold the side sums tile t left in sm_red[t & 1] into this workgroup's planes (fixed order: deterministic).  The plane
    // values are REQUESTED at the top of the iteration that folds them (fold_fetch) and consumed a QK^T product and a
    // softmax later: a read-modify-write that waited out its own L2 round trip stalled wave 0 — and, at the next barrier,
    // the whole workgroup — once per tile.
    float cpv[2 + kMaxBandsM];
    auto fold_fetch = [&](int t, auto obs_c) {
      constexpr bool OBS = decltype(obs_c)::value;
      if (threadIdx.x < kTK && t * kTK + (int)threadIdx.x < L) {
        float* row = cp + t * kTK + threadIdx.x;
        cpv[0] = row[0];
#pragma unroll
        for (int b = 0; b < NBC; b++)
          if (NB >= 0 || b < nb) cpv[1 + b] = row[(size_t)(1 + b) * plane];
        if (OBS) cpv[1 + kMaxBandsM] = row[(size_t)obs_pl * plane];
      }
    };
    auto fold = [&](int t, auto obs_c) {
      constexpr bool OBS = decltype(obs_c)::value;
      if (threadIdx.x < kTK && t * kTK + (int)threadIdx.x < L) {
        const int key = threadIdx.x, pbuf = t & 1;
        float* row = cp + t * kTK + key;
        auto total = [&](int pl) {
          float tot = 0.f;
#pragma unroll
          for (int qs = 0; qs < 8; qs++) tot += sm_red[pbuf][pl][qs][key];
          return tot;
        };
        row[0] = cpv[0] + total(0);
#pragma unroll
        for (int b = 0; b < NBC; b++)
          if (NB >= 0 || b < nb) row[(size_t)(1 + b) * plane] = cpv[1 + b] + total(1 + b);
        if (OBS) row[(size_t)obs_pl * plane] = cpv[1 + kMaxBandsM] + total(obs_pl);
      }
    };
    // ---- one 32-key tile.  FULL: strictly below the diagonal of a complete query tile (no causal / bounds masking);
    //      OBS: this query tile holds observation-window rows
    auto tile = [&](int t, auto full_c, auto obs_c) {
      constexpr bool FULL = decltype(full_c)::value;
      constexpr bool OBS = decltype(obs_c)::value;
      const int k0 = t * kTK, buf = t & 1;
      stamp(6);
      if (t + 1 < ntile) fetch(t + 1);
      if (t > 0) fold_fetch(t - 1, obs_c);
      // S^T tile = K . Q^T from the LDS image
      f32x16 s;
#pragma unroll
      for (int e = 0; e < 16; e++) s[e] = 0.f;
#pragma unroll
      for (int ds = 0; ds < 8; ds++) s = MfmaOps<T>::mma(sm_kt[buf][lq][((2 * ds + hi) ^ (lq & 15)) & 15], qb[ds], s);
      uint32_t pp[8];  // the probabilities as packed 16-bit pairs: the A operand of the P.V products
#pragma unroll
      for (int e = 0; e < 16; e += 2) {  // two elements per dtype conversion (v_cvt_pk_bf16_f32)
        float r0, r1, v0, v1, p0, p1;
        pf_rnd2<T>(s[e], s[e + 1], r0, r1);
        pf_rnd2<T>(r0 * a.scale, r1 * a.scale, v0, v1);
        if (!FULL) {
          const int key0 = k0 + c_row(e, hi), key1 = k0 + c_row(e + 1, hi);
          if (key0 > query || key0 >= L || query >= L) v0 = -INFINITY;
          if (key1 > query || key1 >= L || query >= L) v1 = -INFINITY;
        }
        // exp(v - m) = 2^(v log2e - m log2e), exp(-inf) = 0
        const float e0 = __builtin_amdgcn_exp2f(__builtin_fmaf(v0, kLog2e, m_fin_l2)) * inv_l;
        const float e1 = __builtin_amdgcn_exp2f(__builtin_fmaf(v1, kLog2e, m_fin_l2)) * inv_l;
        pp[e >> 1] = pf_rnd2<T>(e0, e1, p0, p1);
        sm_p[r][c_row(e, hi)][lq] = p0;
        sm_p[r][c_row(e + 1, hi)][lq] = p1;
      }
      stamp(0);
      __syncthreads();  // B1: the four heads' probability tiles are in LDS
      stamp(1);
      {
        float cs = 0.f, os = 0.f, bs[kMaxBandsM] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int qq = 0; qq < 4; qq++) {
          const int ql = rd_qs * 4 + qq;
          const float sum = ((sm_p[0][rd_key][ql] + sm_p[1][rd_key][ql]) + sm_p[2][rd_key][ql]) + sm_p[3][rd_key][ql];
          const float av = ElemTraits<T>::rnd(sum * 0.25f);  // == sum / 4 exactly (power of two); model.py:416-418
          cs += av;
          if (OBS && q0 + ql >= L - a.obs_len && q0 + ql < L) os += av;
          if (NBC > 0) {
            const int dist = (q0 + ql) - (k0 + rd_key);
#pragma unroll
            for (int b = 0; b < NBC; b++)
              if ((NB >= 0 || b < nb) && dist < a.band[b]) bs[b] += av;
          }
        }
        sm_red[buf][0][rd_qs][rd_key] = cs;
#pragma unroll
        for (int b = 0; b < NBC; b++)
          if (NB >= 0 || b < nb) sm_red[buf][1 + b][rd_qs][rd_key] = bs[b];
        if (OBS) sm_red[buf][obs_pl][rd_qs][rd_key] = os;
      }
      if (t > 0) fold(t - 1, obs_c);  // the previous tile's sums became visible at the last barrier
      stamp(2);
      // O += P . V : A = P (C layout -> 16-bit), B = V^T fragments from the LDS image
#pragma unroll
      for (int kb = 0; kb < 2; kb++) {
        const uint4 pa = make_uint4(pp[kb * 4 + 0], pp[kb * 4 + 1], pp[kb * 4 + 2], pp[kb * 4 + 3]);
#pragma unroll
        for (int db = 0; db < 4; db++) {
          const int d = db * 32 + lq;
          o[db] = MfmaOps<T>::mma(pa, sm_vt[buf][d][((kb * 2 + hi) ^ ((d >> 2) & 3)) & 3], o[db]);
        }
      }
      stamp(3);
      if (t + 1 < ntile) stash((t + 1) & 1);
      stamp(4);
      __syncthreads();  // B2: tile t + 1 is in LDS, tile t's side sums are in sm_red[buf], sm_p may be rewritten
      stamp(5);
    };
    __syncthreads();
    fetch(0);
    stash(0);
    __syncthreads();
    // tiles [0, n_full) lie strictly below the diagonal of a complete query tile
    const int n_full = (q0 + kTQ <= L) ? min(ntile, (q0 + 1) / kTK) : 0;
    if (obs_tile) {
      for (int t = 0; t < n_full; t++) tile(t, BoolC<true>{}, BoolC<true>{});
      for (int t = n_full; t < ntile; t++) tile(t, BoolC<false>{}, BoolC<true>{});
      fold_fetch(ntile - 1, BoolC<true>{});
      fold(ntile - 1, BoolC<true>{});
    } else {
      for (int t = 0; t < n_full; t++) tile(t, BoolC<true>{}, BoolC<false>{});
      for (int t = n_full; t < ntile; t++) tile(t, BoolC<false>{}, BoolC<false>{});
      fold_fetch(ntile - 1, BoolC<false>{});
      fold(ntile - 1, BoolC<false>{});
    }
    T* yh = reinterpret_cast<T*>(a.y) + (size_t)j * L * kD;
#pragma unroll
    for (int db = 0; db < 4; db++)
#pragma unroll
      for (int e = 0; e < 16; e++) {
        const int qrow = q0 + c_row(e, hi);
        if (qrow < L) ElemTraits<T>::store(yh, (size_t)qrow * kD + db * 32 + lq, o[db][e]);
      }
    stamp(7);
  }
  if (tracing && lane == 0)
    for (int k = 0; k < 8; k++) a.trace[(h * 4 + r) * 8 + k] = tr_acc[k];
}

// ---------------------------------------------------------------------------------------------------------------
// Single pass (r3): causal attention when NOBODY wants the probabilities — the reference's fast path, attention_utils.py:27-35
// (F.scaled_dot_product_attention for recent_global / l2 / random / full: no cache state is built from the attention).  Same
// workgroup shape and LDS pipeline as pass 2 (K and V^T tiles fetched once per workgroup, one tile ahead, double-buffered
// swizzled images), but no stats pass in front, no probability tiles through LDS, no side planes, ONE barrier per tile:
// online softmax on the swapped S^T tile (every row statistic lane-local + one half-wave exchange), P unnormalised in the
// model dtype as the A operand, O rescaled only when a row's reference maximum moves.
// The reference maximum is LAZY: it is raised only when a tile's row maximum exceeds it by more than kLazy (2^kLazy-fold growth
// of the unnormalised probabilities is harmless in fp32 / bf16: relative rounding is scale-free), so that after the first few
// tiles whole waves skip the rescale — which needs the rows' factors in the O layout (rows = queries across registers), i.e. a
// trip through a wave-private LDS row and 64 multiplies.
// Numerics: scores rounded as the reference's (dtype(dtype(q.k) * scale)); P rounded to the model dtype unnormalised instead of
// normalised — a different, equally good rounding of the same quantity: y within the contract's 1e-3 + 2 roundings of the oracle.
constexpr float kLazy = 6.0f;  // in units of the scaled logits (natural log): e^6 = 403

// STATS (r6): the pass also leaves, per (query head, query), the pair (m_ref, l_exact) the K-stationary side-sum pass below
// normalises with — l_exact sums the UNROUNDED weights exp(x - m_ref) (softmax is shift-invariant: exp(x - m_ref) / l_exact is the
// reference's fp32 softmax up to fp32 rounding, attention_utils.py:52), next to l_run, which sums what P.V multiplies.
template <typename T, bool STATS = false>
__global__ __launch_bounds__(256, CC_PF_WPE_FLASH) void prefill_flash_kernel(MArgs a) {
  __shared__ __attribute__((aligned(16))) uint4 sm_kt[2][kTK][16];   // [buf][key][chunk ^ (key & 15)]
  __shared__ __attribute__((aligned(16))) uint4 sm_vt[2][kD][4];     // [buf][d][chunk ^ ((d >> 2) & 3)]  (V^T, permuted keys)
  __shared__ __attribute__((aligned(16))) float sm_row[4][kTQ];      // per wave: a factor per query row, re-read in the O layout
  const int lane = threadIdx.x & 63, r = threadIdx.x >> 6;
  const int hi = lane >> 5, lq = lane & 31;
  int bx, h;
  wg_coords(a, bx, h);
  const int L = a.L;
  const int j = h * 4 + r;
  const T* qh = reinterpret_cast<const T*>(a.q) + (size_t)j * L * kD;
  const T* kh = reinterpret_cast<const T*>(a.k) + (size_t)h * L * kD;
  const T* vth = reinterpret_cast<const T*>(a.vt) + (size_t)h * kD * a.Lp;
  const int nqt = (L + kTQ - 1) / kTQ;
  const int kl_row = threadIdx.x >> 3, kl_c0 = (threadIdx.x & 7) * 2;   // K tile: row t/8, chunks 2(t%8), +1
  const int vl_row = threadIdx.x >> 1, vl_c0 = (threadIdx.x & 1) * 2;   // V^T tile: d row t/2, chunks 2(t%2), +1
  // (the 16 query rows of this lane's O registers: c_row(e, hi) = (e & 3) + 8 (e >> 2) + 4 hi -> four runs of four floats of sm_row)

  for (int i = 0; i * a.nwg < nqt; i++) {
    const int qt = q_tile(i, bx, a.nwg);
    if (qt >= nqt) continue;  // ragged last round (uniform over the workgroup)
    const int q0 = qt * kTQ;
    const int query = q0 + lq;
    const int qc = query < L ? query : L - 1;
    uint4 qb[8];
#pragma unroll
    for (int ds = 0; ds < 8; ds++) qb[ds] = *reinterpret_cast<const uint4*>(qh + (size_t)qc * kD + ds * 16 + 8 * hi);
    float m_ref = -INFINITY, l_run = 0.f;  // the row's reference maximum (both half-wave lanes agree) / this lane's share of l
    float l_ex = 0.f;                      // STATS: this lane's share of the sum of the unrounded weights
    f32x16 o[4];
#pragma unroll
    for (int b = 0; b < 4; b++)
#pragma unroll
      for (int e = 0; e < 16; e++) o[b][e] = 0.f;
    const int last_q = min(L, q0 + kTQ) - 1;
    const int ntile = last_q / kTK + 1;
    uint4 sk0, sk1, sv0, sv1;
    auto fetch = [&](int t) {
      const int krow = min(t * kTK + kl_row, L - 1);
      const uint4* ks = reinterpret_cast<const uint4*>(kh + (size_t)krow * kD) + kl_c0;
      sk0 = ks[0];
      sk1 = ks[1];
      const uint4* vs = reinterpret_cast<const uint4*>(vth + (size_t)vl_row * a.Lp + t * kTK) + vl_c0;
      sv0 = vs[0];
      sv1 = vs[1];
    };
    auto stash = [&](int buf) {
      sm_kt[buf][kl_row][(kl_c0 ^ (kl_row & 15)) & 15] = sk0;
      sm_kt[buf][kl_row][((kl_c0 + 1) ^ (kl_row & 15)) & 15] = sk1;
      sm_vt[buf][vl_row][(vl_c0 ^ ((vl_row >> 2) & 3)) & 3] = sv0;
      sm_vt[buf][vl_row][((vl_c0 + 1) ^ ((vl_row >> 2) & 3)) & 3] = sv1;
    };
    auto tile = [&](int t, auto full_c) {
      constexpr bool FULL = decltype(full_c)::value;
      const int k0 = t * kTK, buf = t & 1;
      if (t + 1 < ntile) fetch(t + 1);
      f32x16 s;
#pragma unroll
      for (int e = 0; e < 16; e++) s[e] = 0.f;
#pragma unroll
      for (int ds = 0; ds < 8; ds++) s = MfmaOps<T>::mma(sm_kt[buf][lq][((2 * ds + hi) ^ (lq & 15)) & 15], qb[ds], s);
      float x[16];
#pragma unroll
      for (int e = 0; e < 16; e += 2) {  // ref: attention_utils.py:37 dtype(dtype(q.k) * scale), two elements per conversion
        float r0, r1;
        pf_rnd2<T>(s[e], s[e + 1], r0, r1);
        pf_rnd2<T>(r0 * a.scale, r1 * a.scale, x[e], x[e + 1]);
      }
      if (!FULL) {
#pragma unroll
        for (int e = 0; e < 16; e++) {
          const int key = k0 + c_row(e, hi);
          if (key > query || key >= L || query >= L) x[e] = -INFINITY;
        }
      }
      float mx = x[0];
#pragma unroll
      for (int e = 1; e < 16; e++) mx = fmaxf(mx, x[e]);
      {  // the row's maximum over the tile's 32 keys (both half-wave lanes hold it): one v_permlane32_swap — swap(x, x) hands back
         // {ours, the partner's} in some order, and the maximum is symmetric (a ds_bpermute round trip per tile before: found in the ISA)
        const unsigned u = __builtin_bit_cast(unsigned, mx);
        auto sw = __builtin_amdgcn_permlane32_swap(u, u, false, false);
        mx = fmaxf(__builtin_bit_cast(float, (unsigned)sw[0]), __builtin_bit_cast(float, (unsigned)sw[1]));
      }
      // lazy reference maximum: raised only when the tile's maximum leaves it more than kLazy behind (or at the first live tile)
      const bool raise = mx > m_ref + kLazy || (m_ref == -INFINITY && mx > -INFINITY);
      const float m_new = raise ? mx : m_ref;
      float alpha = 1.f;
      if (raise) alpha = m_ref == -INFINITY ? 0.f : pf_exp(m_ref - m_new);
      m_ref = m_new;
      const float mneg = m_new == -INFINITY ? 0.f : -m_new * kLog2e;
      uint32_t pp[8];  // the unnormalised probabilities as packed 16-bit pairs: the A operand of the P.V products
      float psum = 0.f, esum = 0.f;
#pragma unroll
      for (int e = 0; e < 16; e += 2) {
        float p0, p1;
        const float e0 = __builtin_amdgcn_exp2f(__builtin_fmaf(x[e], kLog2e, mneg));
        const float e1 = __builtin_amdgcn_exp2f(__builtin_fmaf(x[e + 1], kLog2e, mneg));
        pp[e >> 1] = pf_rnd2<T>(e0, e1, p0, p1);
        psum += p0 + p1;  // l sums what P.V multiplies: the ROUNDED weights (y is a proper weighted mean of the V rows)
        if constexpr (STATS) esum += e0 + e1;
      }
      l_run = l_run * alpha + psum;
      if constexpr (STATS) l_ex = l_ex * alpha + esum;
      if (__any(raise)) {  // wave-uniform: some row's reference moved — its factor travels to the O layout through LDS
        if (hi == 0) sm_row[r][lq] = alpha;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int g4 = 0; g4 < 4; g4++) {  // four rows at a time: the factors never sit in sixteen registers
          const float4 f4 = *reinterpret_cast<const float4*>(&sm_row[r][8 * g4 + 4 * hi]);
#pragma unroll
          for (int db = 0; db < 4; db++) {
            o[db][4 * g4] *= f4.x;
            o[db][4 * g4 + 1] *= f4.y;
            o[db][4 * g4 + 2] *= f4.z;
            o[db][4 * g4 + 3] *= f4.w;
          }
        }
        __builtin_amdgcn_wave_barrier();  // (the row is rewritten by the next raise)
      }
      // O += P . V : A = P (C layout -> 16-bit), B = V^T fragments from the LDS image
#pragma unroll
      for (int kb = 0; kb < 2; kb++) {
        const uint4 pa = make_uint4(pp[kb * 4 + 0], pp[kb * 4 + 1], pp[kb * 4 + 2], pp[kb * 4 + 3]);
#pragma unroll
        for (int db = 0; db < 4; db++) {
          const int d = db * 32 + lq;
          o[db] = MfmaOps<T>::mma(pa, sm_vt[buf][d][((kb * 2 + hi) ^ ((d >> 2) & 3)) & 3], o[db]);
        }
      }
      if (t + 1 < ntile) stash((t + 1) & 1);
      __syncthreads();  // tile t + 1 is in LDS; every wave is done reading tile t's images
    };
    __syncthreads();  // the previous query tile's last readers are done with both buffers
    fetch(0);
    stash(0);
    __syncthreads();
    const int n_full = (q0 + kTQ <= L) ? min(ntile, (q0 + 1) / kTK) : 0;  // tiles strictly below the diagonal of a complete query tile
    for (int t = 0; t < n_full; t++) tile(t, BoolC<true>{});
    for (int t = n_full; t < ntile; t++) tile(t, BoolC<false>{});
    if constexpr (STATS) {
      const float le = l_ex + __shfl_xor(l_ex, 32, CC_WAVE);
      if (hi == 0 && query < L) {
        a.stats[((size_t)j * L + query) * 2] = m_ref;
        a.stats[((size_t)j * L + query) * 2 + 1] = le;
      }
    }
    // y = O / l: the rows' 1 / l in the O layout, through the wave's LDS row
    const float lt = l_run + __shfl_xor(l_run, 32, CC_WAVE);
    if (hi == 0) sm_row[r][lq] = __frcp_rn(lt);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    T* yh = reinterpret_cast<T*>(a.y) + (size_t)j * L * kD;
#pragma unroll
    for (int g4 = 0; g4 < 4; g4++) {
      const float4 i4 = *reinterpret_cast<const float4*>(&sm_row[r][8 * g4 + 4 * hi]);
      const float inv[4] = {i4.x, i4.y, i4.z, i4.w};
#pragma unroll
      for (int db = 0; db < 4; db++)
#pragma unroll
        for (int u = 0; u < 4; u++) {
          const int e = 4 * g4 + u;
          const int qrow = q0 + c_row(e, hi);
          if (qrow < L) ElemTraits<T>::store(yh, (size_t)qrow * kD + db * 32 + lq, o[db][e] * inv[u]);
        }
    }
    __builtin_amdgcn_wave_barrier();
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Side sums, K-STATIONARY (r6): the column / band / observation-window sums of the group-mean probabilities as a
// QK-only sweep with the KEYS fixed — the shape of a flash backward's dK loop.  ref: model.py:416-418 (group mean -> dtype),
// cache.py:704 (column sums), prompt_compression.py:170-194 (observation window), cache.py:1093, 1155 (FastGen bands).
//
//   unit       = (kv head, block of 128 keys, segment of qseg queries); a workgroup of 4 waves, wave w owning keys
//                k0 + 32 w .. + 31 as the A operand of S^T = K . Q^T — eight 16-byte fragments per lane, loaded ONCE per unit;
//   sweep      = the segment's query tiles (32 queries) at or above the block's first key; the tile's rows of ALL FOUR query
//                heads of the group are staged once per workgroup in a double-buffered, XOR-swizzled LDS image (32 KiB per tile);
//                each wave runs the four heads one after the other against its keys: lane = query, registers = keys, so that the
//                group mean — ((p0 + p1) + p2) + p3, * 0.25, -> dtype: the two-pass kernel's order — is LANE-LOCAL (a running sum
//                across the four heads): no probability tiles through LDS, no barrier for them, no P.V, no O accumulators;
//   sums       = per lane, 16 keys x {column, bands, window} accumulators that live in registers for the whole sweep; ONE
//                cross-lane reduction per unit, one store per key and plane into the unit's own slice of the partial planes
//                [plane][segment][H][L] (every (segment, key) is written by exactly one unit: no atomics, fixed order) — folded
//                over the segments by prefill_side_kernel like the two-pass kernel's per-workgroup planes.
// Probabilities: dtype(exp2(x log2e - m log2e) * (1 / l)) with (m, l) = the flash pass's (m_ref, l_exact) — pass 2's formula.
// Against the two-pass form: the same QK^T contraction and rounding chain, but 1/4 of the group-mean work (no redundancy across
// heads, no LDS round trip), no per-tile plane fold (134 MB of writes at L = 8192), and the P.V half lives in the flash pass.
constexpr int kQSeg = 512;    // queries per unit (a multiple of 32; doubled until the segments fit the workspace's 64 partial planes)
constexpr int kKBlk = 128;    // keys per unit: 4 waves x 32

#ifndef CC_KSTAT_SGB
#define CC_KSTAT_SGB 0   // VALU instructions woven in behind each (fragment read, MFMA) pair of the side-sum pass (0: the compiler's own order —
                         // measured best: 1.514 ms per layer at L = 8192 against 1.540 (10) and 1.547 (18), same box)
#endif
template <typename T, int NB, bool OBS>
__global__ __launch_bounds__(256, 2) void prefill_colsum_kstat_kernel(MArgs a, int nseg, int qseg, int units_per_head) {
  __shared__ __attribute__((aligned(16))) uint4 sm_q[2][4][kTQ][16];  // [buf][head][query][chunk ^ (query & 15)]: 64 KiB
  constexpr int NBC = NB >= 0 ? NB : kMaxBandsM;
  const int nb = NB >= 0 ? NB : a.nb;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int hi = lane >> 5, lq = lane & 31;
  int bx, h;
  wg_coords(a, bx, h);
  const int L = a.L;
  const T* kh = reinterpret_cast<const T*>(a.k) + (size_t)h * L * kD;
  const size_t plane = (size_t)nseg * a.H * L;
  const int obs_pl = 1 + nb;
  const int nkb = (L + kKBlk - 1) / kKBlk;

  for (int u = bx; u < units_per_head; u += a.nwg) {
    // the NON-EMPTY units — segment qs holds the key blocks that start below its last query — in (segment, key block) order:
    // workgroups that run together sweep the same queries (one L2-resident segment); the slices of the empty pairs stay zero
    // (the launcher clears the planes)
    int qs = 0, kb = u;
    for (;; qs++) {
      const int n_here = min(nkb, (min(L, (qs + 1) * qseg) + kKBlk - 1) / kKBlk);
      if (kb < n_here) break;
      kb -= n_here;
    }
    const int k0 = kb * kKBlk + 32 * w;                    // this wave's first key
    const int q_hi = min(L, (qs + 1) * qseg);
    const int q_lo = max(qs * qseg, (kb * kKBlk) & ~31);   // first query tile that can see the block's first key
    float* cp = a.cpart + ((size_t)qs * a.H + h) * L;
    float acc[1 + kMaxBandsM + 1][16];
#pragma unroll
    for (int pl = 0; pl < 2 + kMaxBandsM; pl++)
#pragma unroll
      for (int e = 0; e < 16; e++) acc[pl][e] = 0.f;
    if (q_lo < q_hi) {  // (workgroup-uniform: the loop below holds barriers)
      // K fragments: lane (key k0 + lq, chunk 2 ds + hi), held for the whole sweep
      uint4 kf[8];
      const int kc = min(k0 + lq, L - 1);
#pragma unroll
      for (int ds = 0; ds < 8; ds++) kf[ds] = *reinterpret_cast<const uint4*>(kh + (size_t)kc * kD + ds * 16 + 8 * hi);
      // Q tiles by LDS-DMA (buffer_load ... lds): wave w stages the 32 rows of query head w — eight requests of four rows, lane
      // (row 4 i + lane / 16, slot lane % 16) asking for the chunk that belongs in that slot under the image's XOR swizzle — straight
      // into the image: no staging registers (a register-staged tile was SPILLED to scratch right behind its request at 237 VGPRs:
      // a wait for the prefetch and a round trip through scratch per tile, found in the ISA), no ds_write
      const auto q_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(reinterpret_cast<const T*>(a.q) + (size_t)(h * 4 + w) * L * kD), 0,
                                                          L * kD * (int)sizeof(T), 0x00020000);
      // (the destination through a generic pointer, cast inside a captureless lambda — cc_attn_decode_kernels.h's form: with the cast
      //  of the __shared__ array's address written at the call, the HOST pass silently dropped the kernel's launch stub)
      auto dma16 = [](__amdgpu_buffer_rsrc_t rs, int voff, void* lp) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)lp, 16, voff, 0, 0, 0);
      };
      auto fetch = [&](int q0, int buf) {
#pragma unroll
        for (int i4 = 0; i4 < 8; i4++) {
          const int row = 4 * i4 + (lane >> 4);
          const int qr = min(q0 + row, L - 1);
          const int voff = qr * (kD * (int)sizeof(T)) + (((lane & 15) ^ (row & 15)) & 15) * 16;
          dma16(q_rs, voff, &sm_q[buf][w][4 * i4][0]);
        }
      };
      __syncthreads();  // the previous unit's last readers are done with both buffers
      fetch(q_lo, 0);
      // one query tile (32 queries x the group's 4 heads) against this wave's 32 keys.  FULL: the tile lies strictly below the
      // diagonal of complete tiles — no causal / bounds selects (compiled twice, like the two-pass kernel's tile body)
      // the rows' statistics travel one tile ahead, like the Q tile, and are requested IN FRONT of it: loads return in order — a
      // request behind the next tile's rows (r6, first cut) made every step wait out its own prefetch, an L2 round trip per tile
      float2 st4[4], st4n[4];
      auto fetch_stats = [&](int q0) {
        const int qc = min(q0 + lq, L - 1);
#pragma unroll
        for (int r = 0; r < 4; r++) st4n[r] = *reinterpret_cast<const float2*>(a.stats + ((size_t)(h * 4 + r) * L + qc) * 2);
      };
      // One step = four heads.  The head's eight products (each with its own LDS fragment read) and the PREVIOUS head's rounding
      // chain are independent instruction streams: they are written side by side and the scheduler is told to weave them — one
      // fragment read, one MFMA, a slice of the chain, eight times (CC_KSTAT_SGB) — so that the matrix pipe, the LDS and the VALU of
      // a wave work at the same time.  Left to itself the compiler issues the step's 32 MFMAs first and the 650 VALU instructions
      // behind them, and the four waves of a workgroup — in lockstep at the tile's barrier — all wait for the LDS together and then
      // all for the VALU: 5400 cycles per step where either phase alone is ~2000 (profiles/r06_prefill_kstat.md).
      // One step = four heads, written as SLICES: slice i of a stage holds the fragment read of the NEXT product, product i of head
      // r + 1 and pair i (two of the 16 keys) of head r's rounding chain, and ends at a scheduling fence — the matrix pipe, the LDS and
      // the VALU of a wave work at the same time, by construction (hints alone — sched_group_barrier — were followed loosely and
      // bought nothing: profiles/r06_prefill_kstat.md).  Left to itself the compiler issues the step's 32 MFMAs first and the 650
      // VALU instructions behind them, and the waves of a workgroup, in lockstep at the tile's barrier, all wait for the LDS together
      // and then all for the VALU.
      auto frag = [&](int buf, int idx) {  // fragment `idx` = (head idx / 8, k-chunk idx % 8) of the staged tile
        const int r = (idx >> 3) & 3, ds = idx & 7;
        return sm_q[buf][r][lq][((2 * ds + hi) ^ (lq & 15)) & 15];
      };
      auto chain_pair = [&](int q0, int r, int e, const f32x16& sx, float (&sum)[16], auto full_c) {
        constexpr bool FULL = decltype(full_c)::value;
        const int query = q0 + lq;
        const float m_l2 = -st4[r].x * kLog2e;
        const float inv_l = __frcp_rn(st4[r].y);
        // ref: attention_utils.py:37 (dtype(dtype(q.k) * scale)), :52 (softmax -> dtype)
        float r0, r1, v0, v1, p0, p1;
        pf_rnd2<T>(sx[e], sx[e + 1], r0, r1);
        pf_rnd2<T>(r0 * a.scale, r1 * a.scale, v0, v1);
        if (!FULL) {
          const int key0 = k0 + c_row(e, hi), key1 = k0 + c_row(e + 1, hi);
          if (key0 > query || key0 >= L || query >= L) v0 = -INFINITY;
          if (key1 > query || key1 >= L || query >= L) v1 = -INFINITY;
        }
        const float e0 = __builtin_amdgcn_exp2f(__builtin_fmaf(v0, kLog2e, m_l2)) * inv_l;
        const float e1 = __builtin_amdgcn_exp2f(__builtin_fmaf(v1, kLog2e, m_l2)) * inv_l;
        pf_rnd2<T>(e0, e1, p0, p1);
        if (r == 0) {
          sum[e] = p0;
          sum[e + 1] = p1;
        } else {  // ((p0 + p1) + p2) + p3: the group mean's order (model.py:416-418; the two-pass kernel's)
          sum[e] += p0;
          sum[e + 1] += p1;
        }
      };
      auto step = [&](int q0, int buf, auto full_c) {
        constexpr bool FULL = decltype(full_c)::value;
        const int query = q0 + lq;
        float sum[16];
        f32x16 sa, sb;
#pragma unroll
        for (int e = 0; e < 16; e++) sa[e] = sb[e] = 0.f;
        uint4 fr = frag(buf, 0);
        // head 0's products, bare (reads one ahead)
#pragma unroll
        for (int ds = 0; ds < 8; ds++) {
          const uint4 nx = frag(buf, ds + 1);
          sa = MfmaOps<T>::mma(kf[ds], fr, sa);
          fr = nx;
        }
        __builtin_amdgcn_sched_barrier(0);
        // stages: products of head r + 1 woven with the chain of head r
#pragma unroll
        for (int r = 0; r < 3; r++) {
#pragma unroll
          for (int ds = 0; ds < 8; ds++) {
            const uint4 nx = frag(buf, (r + 1) * 8 + ds + 1);  // (the last one re-reads fragment 0: harmless, keeps the slices alike)
            if ((r & 1) == 0) {
              sb = MfmaOps<T>::mma(kf[ds], fr, ds == 0 ? f32x16{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f} : sb);
              chain_pair(q0, r, 2 * ds, sa, sum, full_c);
            } else {
              sa = MfmaOps<T>::mma(kf[ds], fr, ds == 0 ? f32x16{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f} : sa);
              chain_pair(q0, r, 2 * ds, sb, sum, full_c);
            }
            fr = nx;
            // (the slice's results are pinned HERE: pure VALU work carries no ordering against the fence and was otherwise sunk
            //  below all of them — the slices then held a read, a wait and an MFMA each, and the chains ran behind the lot)
            asm volatile("" : "+v"(sum[2 * ds]), "+v"(sum[2 * ds + 1]));
            __builtin_amdgcn_sched_barrier(0);
          }
        }
#pragma unroll
        for (int ds = 0; ds < 8; ds++) chain_pair(q0, 3, 2 * ds, sb, sum, full_c);
        float obs_f = 0.f;
        if constexpr (OBS) obs_f = (query >= L - a.obs_len && query < L) ? 1.f : 0.f;
        // a band plane takes the tile whole, not at all, or element by element (only the tiles the band's edge crosses)
        int band_mode[kMaxBandsM];
#pragma unroll
        for (int b = 0; b < NBC; b++) {
          const int bw = a.band[b];
          band_mode[b] = (NB < 0 && b >= nb) ? 0 : ((q0 + kTQ - 1 - k0 < bw) ? 2 : ((q0 - (k0 + kTK - 1) >= bw) ? 0 : 1));
        }
#pragma unroll
        for (int e = 0; e < 16; e += 2) {
          float a0, a1;
          pf_rnd2<T>(sum[e] * 0.25f, sum[e + 1] * 0.25f, a0, a1);  // == / 4 exactly
          acc[0][e] += a0;
          acc[0][e + 1] += a1;
          if constexpr (OBS) {
            acc[1 + kMaxBandsM][e] = __builtin_fmaf(a0, obs_f, acc[1 + kMaxBandsM][e]);
            acc[1 + kMaxBandsM][e + 1] = __builtin_fmaf(a1, obs_f, acc[1 + kMaxBandsM][e + 1]);
          }
#pragma unroll
          for (int b = 0; b < NBC; b++) {
            if (band_mode[b] == 2) {
              acc[1 + b][e] += a0;
              acc[1 + b][e + 1] += a1;
            } else if (band_mode[b] == 1) {
              if (query - (k0 + c_row(e, hi)) < a.band[b]) acc[1 + b][e] += a0;
              if (query - (k0 + c_row(e + 1, hi)) < a.band[b]) acc[1 + b][e + 1] += a1;
            }
          }
        }
        (void)FULL;
      };
      int buf = 0;
      fetch_stats(q_lo);
#pragma unroll
      for (int r = 0; r < 4; r++) st4[r] = st4n[r];
      __syncthreads();
      for (int q0 = q_lo; q0 < q_hi; q0 += kTQ, buf ^= 1) {
        const bool more = q0 + kTQ < q_hi;
        // (unconditional, clamped: a branch around loads makes every later in-order wait conservative)
        fetch_stats(more ? q0 + kTQ : q0);
        __builtin_amdgcn_sched_barrier(0);
        fetch(more ? q0 + kTQ : q0, buf ^ 1);
        __builtin_amdgcn_sched_barrier(0);
        // tiles whose last query lies below this wave's first key hold nothing for it (the diagonal block's upper waves)
        if (q0 + kTQ - 1 >= k0 && k0 < L) {
          if ((k0 + kTK - 1 <= q0) && (q0 + kTQ <= L) && (k0 + kTK <= L)) step(q0, buf, BoolC<true>{});
          else step(q0, buf, BoolC<false>{});
        }
        // (the next tile's statistics were requested ahead of its rows: their wait leaves the DMA requests in flight)
#pragma unroll
        for (int r = 0; r < 4; r++) st4[r] = st4n[r];
        __syncthreads();  // (a release at workgroup scope: waits for this wave's DMA requests) tile q0 + 32 is in LDS; every wave is done with tile q0's image
      }
    }
    // ---- the unit's sums: over the 32 query lanes of each half-wave (fixed butterfly: deterministic), one store per key and plane
    auto reduce_store = [&](float (&ac)[16], int dst_pl) {  // (the plane by reference: a run-time plane index would put acc in scratch)
#pragma unroll
      for (int e = 0; e < 16; e++) {
        float v = ac[e];
#pragma unroll
        for (int off = 16; off > 0; off >>= 1) v += __shfl_xor(v, off, 32);
        const int key = k0 + c_row(e, hi);
        if (lq == 0 && key < L) cp[(size_t)dst_pl * plane + key] = v;
      }
    };
    reduce_store(acc[0], 0);
    if constexpr (NBC > 0) {
      if (NB >= 0 || 0 < nb) reduce_store(acc[1], 1);
    }
    if constexpr (NBC > 1) {
      if (NB >= 0 || 1 < nb) reduce_store(acc[2], 2);
    }
    if constexpr (NBC > 2) {
      if (NB >= 0 || 2 < nb) reduce_store(acc[3], 3);
    }
    if constexpr (NBC > 3) {
      if (NB >= 0 || 3 < nb) reduce_store(acc[4], 4);
    }
    if constexpr (OBS) reduce_store(acc[1 + kMaxBandsM], obs_pl);
  }
}

template <typename T>
static void launch_colsum(const MArgs& a, dim3 grid, int nseg, int qseg, int units_per_head, hipStream_t st) {
#define CC_COLSUM_LAUNCH(NB_, OBS_) hipLaunchKernelGGL((prefill_colsum_kstat_kernel<T, NB_, OBS_>), grid, dim3(256), 0, st, a, nseg, qseg, units_per_head)
  const bool obs = a.obs_len > 0;
  switch (a.nb) {
    case 0: if (obs) CC_COLSUM_LAUNCH(0, true); else CC_COLSUM_LAUNCH(0, false); break;
    case 1: if (obs) CC_COLSUM_LAUNCH(1, true); else CC_COLSUM_LAUNCH(1, false); break;
    default: if (obs) CC_COLSUM_LAUNCH(-1, true); else CC_COLSUM_LAUNCH(-1, false); break;
  }
#undef CC_COLSUM_LAUNCH
}

template <typename T>
static void launch_pv(const MArgs& a, dim3 grid, hipStream_t st) {
  switch (a.nb) {
    case 0: hipLaunchKernelGGL((prefill_pv_lds_kernel<T, 0>), grid, dim3(256), 0, st, a); break;
    case 1: hipLaunchKernelGGL((prefill_pv_lds_kernel<T, 1>), grid, dim3(256), 0, st, a); break;
    default: hipLaunchKernelGGL((prefill_pv_lds_kernel<T, -1>), grid, dim3(256), 0, st, a); break;
  }
}

}  // namespace

// Single-pass entry (no side outputs).  workspace: vt_perm only.
extern "C" int cc_prefill_attn_flash_impl(const void* q, const void* k, const void* v, int HQ, int H, int L, int D, int dtype,
                                          float scale, void* y, int nwg, void* vt, hipStream_t st) {
  if (D != kD || HQ != 4 * H || (dtype != CC_DT_BF16 && dtype != CC_DT_F16)) return CC_ERR_UNSUPPORTED;
  const int Lp = (L + 31) & ~31;
  MArgs a{};
  a.q = q; a.k = k; a.vt = vt; a.y = y;
  a.H = H; a.L = L; a.Lp = Lp; a.scale = scale; a.nwg = nwg;
  static const bool no_remap = getenv("CC_PREFILL_NO_XCD_REMAP") != nullptr;  // measurement only
  a.xcd_remap = (H % 8 == 0 && !no_remap) ? 1 : 0;
  const size_t tot = (size_t)H * kD * Lp;
  size_t nbk = (tot + 255) / 256;
  if (nbk > 8192) nbk = 8192;
  const dim3 grid((unsigned)(nwg * H)), block(256);
  if (dtype == CC_DT_BF16) {
    hipLaunchKernelGGL(vt_perm_kernel<bf16_t>, dim3((unsigned)nbk), dim3(256), 0, st, (const bf16_t*)v, (bf16_t*)vt, H, L, Lp);
    hipLaunchKernelGGL(prefill_flash_kernel<bf16_t>, grid, block, 0, st, a);
  } else {
    hipLaunchKernelGGL(vt_perm_kernel<f16_t>, dim3((unsigned)nbk), dim3(256), 0, st, (const f16_t*)v, (f16_t*)vt, H, L, Lp);
    hipLaunchKernelGGL(prefill_flash_kernel<f16_t>, grid, block, 0, st, a);
  }
  if (hipGetLastError() != hipSuccess) return CC_ERR_HIP;
  return CC_OK;
}

// Entry used by cc_attn_prefill.hip's dispatcher.  Returns CC_ERR_UNSUPPORTED when the geometry is not the MFMA one.
// workspace layout is owned by the caller: stats | cpart planes | vt_perm.
extern "C" int cc_prefill_attn_mfma_impl(const void* q, const void* k, const void* v, int HQ, int H, int L, int D, int dtype,
                                         float scale, void* y, float* stats, float* cpart, int nwg, void* vt, const int* bands,
                                         int nb, int obs_len, hipStream_t st, int max_partials, int* n_partials) {
  if (D != kD || HQ != 4 * H || (dtype != CC_DT_BF16 && dtype != CC_DT_F16) || nb > kMaxBandsM) return CC_ERR_UNSUPPORTED;
  const int Lp = (L + 31) & ~31;
  MArgs a{};
  a.q = q; a.k = k; a.vt = vt; a.y = y; a.stats = stats; a.cpart = cpart;
  a.H = H; a.L = L; a.Lp = Lp; a.nb = nb; a.scale = scale; a.obs_len = obs_len;
  for (int b = 0; b < nb; b++) a.band[b] = bands[b];
  const size_t tot = (size_t)H * kD * Lp;
  size_t nbk = (tot + 255) / 256;
  if (nbk > 8192) nbk = 8192;
  dim3 block(256);
  static const bool no_remap = getenv("CC_PREFILL_NO_XCD_REMAP") != nullptr;  // measurement only
  a.xcd_remap = (H % 8 == 0 && !no_remap) ? 1 : 0;
  *n_partials = nwg;
  // r6 (built and MEASURED — profiles/r06_prefill_kstat.md): the flash pass (y, and the row statistics as a by-product)
  // + the K-stationary side-sum pass, instead of statistics pass + probabilities / P.V pass.  A tie at L = 8192 (1.51-1.55 ms against
  // 1.55), a loss at 16384 (5.60 against 5.33): the side-sum pass runs its LDS-bound fragment reads and its VALU-bound rounding chains
  // as two phases in lockstep across the workgroup (0.66 ms where either phase alone is ~0.25), and the statistics cost the flash pass
  // 12 %.  OPT-IN (CC_PREFILL_KSTAT=1); the two-pass form stays the product.
  static const char* e_kstat = getenv("CC_PREFILL_KSTAT");
  if (e_kstat && atoi(e_kstat) == 1) {
    int qseg = kQSeg;
    while ((L + qseg - 1) / qseg > max_partials) qseg *= 2;
    const int nseg = (L + qseg - 1) / qseg, nkb = (L + kKBlk - 1) / kKBlk;
    int units = 0;
    for (int qs = 0; qs < nseg; qs++) {
      const int q_end = L < (qs + 1) * qseg ? L : (qs + 1) * qseg;
      const int n_here = (q_end + kKBlk - 1) / kKBlk;
      units += n_here < nkb ? n_here : nkb;
    }
    const int nqt = (L + kTQ - 1) / kTQ;
    MArgs af = a;
    af.nwg = nqt < 128 ? nqt : 128;
    MArgs ac = a;
    ac.nwg = units < 64 ? units : 64;  // two workgroups per CU of the head's XCD
    const int npl = 1 + nb + (obs_len > 0 ? 1 : 0);
    if (hipMemsetAsync(cpart, 0, (size_t)npl * nseg * H * L * sizeof(float), st) != hipSuccess) return CC_ERR_HIP;
    const dim3 gridf((unsigned)(af.nwg * H)), gridc((unsigned)(ac.nwg * H));
    if (dtype == CC_DT_BF16) {
      hipLaunchKernelGGL(vt_perm_kernel<bf16_t>, dim3((unsigned)nbk), dim3(256), 0, st, (const bf16_t*)v, (bf16_t*)vt, H, L, Lp);
      hipLaunchKernelGGL((prefill_flash_kernel<bf16_t, true>), gridf, block, 0, st, af);
      launch_colsum<bf16_t>(ac, gridc, nseg, qseg, units, st);
    } else {
      hipLaunchKernelGGL(vt_perm_kernel<f16_t>, dim3((unsigned)nbk), dim3(256), 0, st, (const f16_t*)v, (f16_t*)vt, H, L, Lp);
      hipLaunchKernelGGL((prefill_flash_kernel<f16_t, true>), gridf, block, 0, st, af);
      launch_colsum<f16_t>(ac, gridc, nseg, qseg, units, st);
    }
    if (hipGetLastError() != hipSuccess) return CC_ERR_HIP;
    *n_partials = nseg;
    return CC_OK;
  }
  // pass 1 keeps no per-workgroup partial planes; 128 persistent workgroups per head (four per CU) measured best at L = 8192
  // (64: 0.545 ms, 128: 0.472, 256: 0.512) — with the serpentine assignment every one of them gets the same share of the triangle
  const int nqt = (L + kTQ - 1) / kTQ;
  MArgs a1 = a;
  a1.nwg = nqt < 128 ? nqt : 128;
  a.nwg = nwg;
  static unsigned long long* trace_buf = nullptr;  // measurement only: phase clocks of the LAST pass-2 launch, printed at exit
  static const bool want_trace = getenv("CC_PREFILL_TRACE") != nullptr;
  if (want_trace && !trace_buf) {
    if (hipMalloc(&trace_buf, 64 * 4 * 8 * sizeof(unsigned long long)) != hipSuccess) trace_buf = nullptr;
    static unsigned long long* tb = trace_buf;
    atexit([] {
      unsigned long long hbuf[64 * 4 * 8];
      if (hipMemcpy(hbuf, tb, sizeof(hbuf), hipMemcpyDeviceToHost) != hipSuccess) return;
      const char* names[8] = {"qk+softmax+p->lds", "wait B1", "side sums+fold", "pv issue", "stash (vmcnt)", "wait B2", "prefetch issue", "q-tile epilogue"};
      for (int hw = 0; hw < 8; hw++) {  // head 0 and 1, four waves each
        unsigned long long tot = 0;
        for (int k = 0; k < 8; k++) tot += hbuf[hw * 8 + k];
        fprintf(stderr, "[prefill trace] head %d wave %d total %llu cycles:", hw / 4, hw % 4, tot);
        for (int k = 0; k < 8; k++) fprintf(stderr, " %s %.1f%%", names[k], tot ? 100.0 * hbuf[hw * 8 + k] / tot : 0.0);
        fprintf(stderr, "\n");
      }
    });
  }
  a.trace = trace_buf;
  static const char* e_nwg1 = getenv("CC_PREFILL_NWG1");  // measurement only
  if (e_nwg1 && atoi(e_nwg1) > 0) a1.nwg = nqt < atoi(e_nwg1) ? nqt : atoi(e_nwg1);
  dim3 grid1((unsigned)(a1.nwg * H)), grid((unsigned)(nwg * H));
  if (dtype == CC_DT_BF16) {
    hipLaunchKernelGGL(vt_perm_kernel<bf16_t>, dim3((unsigned)nbk), dim3(256), 0, st, (const bf16_t*)v, (bf16_t*)vt, H, L, Lp);
    hipLaunchKernelGGL(prefill_stats_lds_kernel<bf16_t>, grid1, block, 0, st, a1);
    launch_pv<bf16_t>(a, grid, st);
  } else {
    hipLaunchKernelGGL(vt_perm_kernel<f16_t>, dim3((unsigned)nbk), dim3(256), 0, st, (const f16_t*)v, (f16_t*)vt, H, L, Lp);
    hipLaunchKernelGGL(prefill_stats_lds_kernel<f16_t>, grid1, block, 0, st, a1);
    launch_pv<f16_t>(a, grid, st);
  }
  if (hipGetLastError() != hipSuccess) return CC_ERR_HIP;
  return CC_OK;
}
