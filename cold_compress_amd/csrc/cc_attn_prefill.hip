// cc_attn_prefill.hip — causal prefill attention with the side outputs the eviction policies need
// (column sums and observation-window mean of the group-averaged probabilities) WITHOUT materialising the
// [HQ, L, L] probability tensor the reference builds (attention_utils.py:36-54, 3 x 4 GiB at L = 8192).
//
// Generic path (any dtype, head_dim <= 256): LDS-tiled, two passes.
//   P1  row statistics (m_i, l_i) per (query head, query) -> workspace            [all blocks independent]
//   P2  recompute scores, P = dtype(exp(x-m)/l), y += P.V, per-workgroup column-sum partials
//       (workgroups own query blocks w, w+NWG, ...: deterministic accumulation order, no atomics)
//   P3  column-sum reduction over the NWG partials + observation-window mean (last obs_len queries)
// Rounding points follow the reference: score -> dtype, * scale -> dtype, softmax -> dtype, group mean ->
// dtype (model.py:416-418).  Results are deterministic run to run.
#include <stdlib.h>

#include "cc_common.h"

namespace {

constexpr int kRows = 32;     // query rows (R * QB) per workgroup
constexpr int kThreads = 256;
constexpr int kNWG = 32;      // persistent workgroups per kv head in P2

struct PfArgs {
  const void* q;
  const void* k;
  const void* v;
  void* y;
  float* stats;   // [HQ, L, 2] (m, l)
  float* cpart;   // [1 + nb, kNWG, H, L]: plane 0 = column sums, plane 1+b = band b
  float* colsum;  // [H, L] or null
  float* obs;     // [H, L] or null
  int HQ, H, R, L, D, QB, TK, obs_len;
  float scale;
  int nb;           // number of band-sum side outputs (<= kMaxBands)
  int band[4];      // window widths in queries: band b sums a[h,q,k] over q in [k, k + band[b])
  float* band_out;  // [nb, H, L] or null
  int obs_plane;    // >= 0: the observation-window sums are plane obs_plane of cpart (MFMA path); -1: recompute here
};
constexpr int kMaxBands = 4;

// LDS carve (floats): Qs[kRows][D+1] | Ks[TK][D+1] | Vs[TK][D] | Ps[kRows][TK+1]
__device__ __forceinline__ int lds_floats(int D, int TK) { return kRows * (D + 1) + TK * (D + 1) + TK * D + kRows * (TK + 1); }

template <typename T>
__device__ __forceinline__ void load_q_block(const PfArgs& a, float* Qs, int h, int qb0) {
  const int D = a.D;
  for (int t = threadIdx.x; t < kRows * D; t += kThreads) {
    const int row = t / D, d = t - row * D;
    const int r = row / a.QB, il = row - r * a.QB, i = qb0 + il;
    float val = 0.f;
    if (r < a.R && i < a.L) val = ElemTraits<T>::load(reinterpret_cast<const T*>(a.q), ((size_t)(h * a.R + r) * a.L + i) * D + d);
    Qs[row * (D + 1) + d] = val;
  }
}

template <typename T>
__device__ __forceinline__ void load_kv_tile(const PfArgs& a, float* Ks, float* Vs, int h, int k0, bool want_v) {
  const int D = a.D;
  for (int t = threadIdx.x; t < a.TK * D; t += kThreads) {
    const int kk = t / D, d = t - kk * D, s = k0 + kk;
    float kv = 0.f, vv = 0.f;
    if (s < a.L) {
      kv = ElemTraits<T>::load(reinterpret_cast<const T*>(a.k), ((size_t)h * a.L + s) * D + d);
      if (want_v) vv = ElemTraits<T>::load(reinterpret_cast<const T*>(a.v), ((size_t)h * a.L + s) * D + d);
    }
    Ks[kk * (D + 1) + d] = kv;
    if (want_v) Vs[kk * D + d] = vv;
  }
}

// scores for the (rows x TK) tile -> Ps (dtype-rounded, -inf outside the causal triangle / padding)
template <typename T>
__device__ __forceinline__ void tile_scores(const PfArgs& a, const float* Qs, const float* Ks, float* Ps, int qb0, int k0) {
  const int D = a.D, TK = a.TK;
  const int row = threadIdx.x & (kRows - 1);
  const int r = row / a.QB, il = row - r * a.QB, i = qb0 + il;
  for (int kk = threadIdx.x / kRows; kk < TK; kk += kThreads / kRows) {
    const int s = k0 + kk;
    float dot = 0.f;
    const float* qr = Qs + row * (D + 1);
    const float* kr = Ks + kk * (D + 1);
    for (int d = 0; d < D; d++) dot = fmaf(qr[d], kr[d], dot);
    float x = ElemTraits<T>::rnd(ElemTraits<T>::rnd(dot) * a.scale);
    if (r >= a.R || i >= a.L || s > i) x = -INFINITY;
    Ps[row * (TK + 1) + kk] = x;
  }
}

template <typename T>
__global__ __launch_bounds__(kThreads) void prefill_stats_kernel(PfArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int D = a.D, TK = a.TK;
  float* Qs = smem;
  float* Ks = Qs + kRows * (D + 1);
  float* Vs = Ks + TK * (D + 1);
  float* Ps = Vs + TK * D;
  const int h = blockIdx.y, qb0 = blockIdx.x * a.QB;
  load_q_block<T>(a, Qs, h, qb0);
  float m = -INFINITY, l = 0.f;  // thread t < kRows owns row t
  const int last_q = min(a.L, qb0 + a.QB) - 1;
  for (int k0 = 0; k0 <= last_q; k0 += TK) {
    __syncthreads();
    load_kv_tile<T>(a, Ks, Vs, h, k0, false);
    __syncthreads();
    tile_scores<T>(a, Qs, Ks, Ps, qb0, k0);
    __syncthreads();
    if (threadIdx.x < kRows) {
      const float* pr = Ps + threadIdx.x * (TK + 1);
      float mx = m;
      for (int kk = 0; kk < TK; kk++) mx = fmaxf(mx, pr[kk]);
      const float mu = (mx == -INFINITY) ? 0.f : mx;
      float sum = l * expf(m - mu);
      for (int kk = 0; kk < TK; kk++) sum += expf(pr[kk] - mu);
      m = mx;
      l = sum;
    }
  }
  if (threadIdx.x < kRows) {
    const int row = threadIdx.x, r = row / a.QB, il = row - r * a.QB, i = qb0 + il;
    if (r < a.R && i < a.L) {
      float* st = a.stats + ((size_t)(h * a.R + r) * a.L + i) * 2;
      st[0] = m;
      st[1] = l;
    }
  }
}

template <typename T>
__global__ __launch_bounds__(kThreads) void prefill_pv_kernel(PfArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int D = a.D, TK = a.TK, L = a.L;
  float* Qs = smem;
  float* Ks = Qs + kRows * (D + 1);
  float* Vs = Ks + TK * (D + 1);
  float* Ps = Vs + TK * D;
  __shared__ float sm_m[kRows], sm_l[kRows];
  const int h = blockIdx.y, w = blockIdx.x;
  float* cp = a.cpart + ((size_t)w * a.H + h) * L;
  const size_t plane = (size_t)gridDim.x * a.H * L;  // one plane per side output
  for (int pl = 0; pl <= a.nb; pl++)
    for (int s = threadIdx.x; s < L; s += kThreads) cp[pl * plane + s] = 0.f;
  const int nqb = (L + a.QB - 1) / a.QB;
  const int n_out = kRows * D;  // y accumulators of this block, kThreads-strided
  for (int qb = w; qb < nqb; qb += gridDim.x) {
    const int qb0 = qb * a.QB;
    __syncthreads();
    load_q_block<T>(a, Qs, h, qb0);
    if (threadIdx.x < kRows) {
      const int row = threadIdx.x, r = row / a.QB, il = row - r * a.QB, i = qb0 + il;
      float m = 0.f, l = 1.f;
      if (r < a.R && i < L) {
        const float* st = a.stats + ((size_t)(h * a.R + r) * L + i) * 2;
        m = st[0];
        l = st[1];
      }
      sm_m[row] = m;
      sm_l[row] = l;
    }
    float acc[32];
#pragma unroll
    for (int c = 0; c < 32; c++) acc[c] = 0.f;
    const int last_q = min(L, qb0 + a.QB) - 1;
    for (int k0 = 0; k0 <= last_q; k0 += TK) {
      __syncthreads();
      load_kv_tile<T>(a, Ks, Vs, h, k0, true);
      __syncthreads();
      tile_scores<T>(a, Qs, Ks, Ps, qb0, k0);
      __syncthreads();
      // probabilities (ref: attention_utils.py:52), in place
      for (int t = threadIdx.x; t < kRows * TK; t += kThreads) {
        const int row = t / TK, kk = t - row * TK;
        const float x = Ps[row * (TK + 1) + kk];
        Ps[row * (TK + 1) + kk] = ElemTraits<T>::rnd(__fdiv_rn(expf(x - sm_m[row]), sm_l[row]));
      }
      __syncthreads();
      // column sums of the group mean (ref: model.py:416-418, cache.py:704), one key per thread
      if (threadIdx.x < TK && k0 + threadIdx.x < L) {
        const int kk = threadIdx.x;
        float cs = 0.f, bs[kMaxBands] = {0.f, 0.f, 0.f, 0.f};
        for (int il = 0; il < a.QB; il++) {
          float sum = 0.f;
          for (int r = 0; r < a.R; r++) sum += Ps[(r * a.QB + il) * (TK + 1) + kk];
          const float av = ElemTraits<T>::rnd(__fdiv_rn(sum, (float)a.R));
          cs += av;
          const int dist = (qb0 + il) - (k0 + kk);  // query index - key index (av == 0 when negative)
#pragma unroll
          for (int b = 0; b < kMaxBands; b++)
            if (b < a.nb && dist < a.band[b]) bs[b] += av;
        }
        cp[k0 + kk] += cs;
#pragma unroll
        for (int b = 0; b < kMaxBands; b++)
          if (b < a.nb) cp[(size_t)(1 + b) * plane + k0 + kk] += bs[b];
      }
      // y += P . V
#pragma unroll
      for (int c = 0; c < 32; c++) {
        const int o = threadIdx.x + c * kThreads;
        if (o < n_out) {
          const int row = o / D, d = o - row * D;
          const float* pr = Ps + row * (TK + 1);
          float s = acc[c];
          for (int kk = 0; kk < TK; kk++) s = fmaf(pr[kk], Vs[kk * D + d], s);
          acc[c] = s;
        }
      }
    }
#pragma unroll
    for (int c = 0; c < 32; c++) {
      const int o = threadIdx.x + c * kThreads;
      if (o < n_out) {
        const int row = o / D, d = o - row * D;
        const int r = row / a.QB, il = row - r * a.QB, i = qb0 + il;
        if (r < a.R && i < L) ElemTraits<T>::store(reinterpret_cast<T*>(a.y), ((size_t)(h * a.R + r) * L + i) * D + d, acc[c]);
      }
    }
  }
}

template <typename T>
__global__ __launch_bounds__(kThreads) void prefill_side_kernel(PfArgs a, int nwg) {
  const int L = a.L, D = a.D;
  const int total = a.H * L;
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
    const int h = idx / L, s = idx - h * L;
    if (a.colsum) {
      float cs = 0.f;
      for (int w = 0; w < nwg; w++) cs += a.cpart[((size_t)w * a.H + h) * L + s];
      a.colsum[idx] = cs;
    }
    if (a.band_out) {
      const size_t plane = (size_t)nwg * a.H * L;
      for (int b = 0; b < a.nb; b++) {
        float bs = 0.f;
        for (int w = 0; w < nwg; w++) bs += a.cpart[(size_t)(1 + b) * plane + ((size_t)w * a.H + h) * L + s];
        a.band_out[(size_t)b * a.H * L + idx] = bs;
      }
    }
    if (a.obs && a.obs_plane >= 0) {
      const size_t plane = (size_t)nwg * a.H * L;
      float os = 0.f;
      for (int w = 0; w < nwg; w++) os += a.cpart[(size_t)a.obs_plane * plane + ((size_t)w * a.H + h) * L + s];
      a.obs[idx] = a.obs_len > 0 ? __fdiv_rn(os, (float)a.obs_len) : 0.f;
    } else if (a.obs) {
      // ref: prompt_compression.py:173 attn[:, :, -obs_len:, :].mean(dim=2)
      float os = 0.f;
      const T* kr = reinterpret_cast<const T*>(a.k) + ((size_t)h * L + s) * D;
      for (int i = L - a.obs_len; i < L; i++) {
        float sum = 0.f;
        if (s <= i) {
          for (int r = 0; r < a.R; r++) {
            const size_t j = (size_t)h * a.R + r;
            const T* qr = reinterpret_cast<const T*>(a.q) + (j * L + i) * D;
            float dot = 0.f;
            for (int d = 0; d < D; d++) dot = fmaf(ElemTraits<T>::load(qr, d), ElemTraits<T>::load(kr, d), dot);
            const float x = ElemTraits<T>::rnd(ElemTraits<T>::rnd(dot) * a.scale);
            const float* st = a.stats + (j * L + i) * 2;
            sum += ElemTraits<T>::rnd(__fdiv_rn(expf(x - st[0]), st[1]));
          }
        }
        os += ElemTraits<T>::rnd(__fdiv_rn(sum, (float)a.R));
      }
      a.obs[idx] = a.obs_len > 0 ? __fdiv_rn(os, (float)a.obs_len) : 0.f;
    }
  }
}

static size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

template <typename T>
static int run_prefill(PfArgs a, hipStream_t st) {
  const size_t lds = (size_t)(kRows * (a.D + 1) + a.TK * (a.D + 1) + a.TK * a.D + kRows * (a.TK + 1)) * sizeof(float);
  if (lds > 64 * 1024) {
    if (hipFuncSetAttribute((const void*)prefill_stats_kernel<T>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess ||
        hipFuncSetAttribute((const void*)prefill_pv_kernel<T>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
      return CC_ERR_HIP;
  }
  const int nqb = (a.L + a.QB - 1) / a.QB;
  hipLaunchKernelGGL(prefill_stats_kernel<T>, dim3(nqb, a.H), dim3(kThreads), lds, st, a);
  CC_LAUNCH_CHECK();
  const int nwg = nqb < kNWG ? nqb : kNWG;
  hipLaunchKernelGGL(prefill_pv_kernel<T>, dim3(nwg, a.H), dim3(kThreads), lds, st, a);
  CC_LAUNCH_CHECK();
  if (a.colsum || a.obs || a.band_out) {
    int nb = (a.H * a.L + kThreads - 1) / kThreads;
    if (nb > 2048) nb = 2048;
    hipLaunchKernelGGL(prefill_side_kernel<T>, dim3(nb), dim3(kThreads), 0, st, a, nwg);
    CC_LAUNCH_CHECK();
  }
  return CC_OK;
}

constexpr int kNWGMfma = 64;  // persistent workgroups per kv head on the MFMA path: 512 = two per CU, all co-resident

// MFMA path (cc_attn_prefill_mfma.hip): 16-bit dtype, D == 128, HQ == 4*H.
static bool mfma_eligible(int HQ, int H, int D, int dtype) {
  return D == 128 && HQ == 4 * H && (dtype == CC_DT_BF16 || dtype == CC_DT_F16);
}

template <typename T>
static int run_side(PfArgs a, int nwg, hipStream_t st) {
  int nb = (a.H * a.L + kThreads - 1) / kThreads;
  if (nb > 2048) nb = 2048;
  hipLaunchKernelGGL(prefill_side_kernel<T>, dim3(nb), dim3(kThreads), 0, st, a, nwg);
  CC_LAUNCH_CHECK();
  return CC_OK;
}

}  // namespace

extern "C" int cc_prefill_attn_mfma_impl(const void* q, const void* k, const void* v, int HQ, int H, int L, int D, int dtype,
                                         float scale, void* y, float* stats, float* cpart, int nwg, void* vt, const int* bands,
                                         int nb, int obs_len, hipStream_t st, int max_partials, int* n_partials);

extern "C" int cc_prefill_attn_flash_impl(const void* q, const void* k, const void* v, int HQ, int H, int L, int D, int dtype,
                                          float scale, void* y, int nwg, void* vt, hipStream_t st);

extern "C" {

size_t cc_prefill_attn_workspace_bytes(int32_t HQ, int32_t H, int32_t L, int32_t D, int32_t dtype) {
  if (HQ <= 0 || H <= 0 || L <= 0) return 0;
  const size_t Lp = ((size_t)L + 31) & ~(size_t)31;
  return align256((size_t)HQ * L * 2 * sizeof(float)) + align256((size_t)(2 + kMaxBands) * kNWGMfma * H * L * sizeof(float)) +
         (mfma_eligible(HQ, H, D, dtype) ? align256((size_t)H * D * Lp * 2) : 0);
}

int cc_prefill_attn_bands(const void* q, const void* k, const void* v, int32_t HQ, int32_t H, int32_t L, int32_t D,
                          int32_t dtype, float scale, void* y, float* colsum_out, float* obs_out, int32_t obs_len,
                          const int32_t* bands, int32_t n_bands, float* band_out, void* workspace,
                          size_t workspace_bytes, cc_stream_t stream);

int cc_prefill_attn(const void* q, const void* k, const void* v, int32_t HQ, int32_t H, int32_t L, int32_t D,
                    int32_t dtype, float scale, void* y, float* colsum_out, float* obs_out, int32_t obs_len,
                    void* workspace, size_t workspace_bytes, cc_stream_t stream) {
  return cc_prefill_attn_bands(q, k, v, HQ, H, L, D, dtype, scale, y, colsum_out, obs_out, obs_len, nullptr, 0, nullptr,
                               workspace, workspace_bytes, stream);
}

int cc_prefill_attn_bands(const void* q, const void* k, const void* v, int32_t HQ, int32_t H, int32_t L, int32_t D,
                          int32_t dtype, float scale, void* y, float* colsum_out, float* obs_out, int32_t obs_len,
                          const int32_t* bands, int32_t n_bands, float* band_out, void* workspace,
                          size_t workspace_bytes, cc_stream_t stream) {
  CC_ENTRY();
  if (n_bands < 0 || n_bands > kMaxBands || (n_bands > 0 && (!bands || !band_out))) return CC_ERR_BAD_ARG;
  if (!q || !k || !v || !y || HQ <= 0 || H <= 0 || HQ % H || L <= 0 || D <= 0 || !cc_dt_ok(dtype) || !workspace)
    return CC_ERR_BAD_ARG;
  const int R = HQ / H;
  if (R > kRows || D > 256) return CC_ERR_UNSUPPORTED;
  if (workspace_bytes < cc_prefill_attn_workspace_bytes(HQ, H, L, D, dtype)) return CC_ERR_WORKSPACE;
  PfArgs a{};
  a.q = q; a.k = k; a.v = v; a.y = y;
  a.stats = reinterpret_cast<float*>(workspace);
  a.cpart = reinterpret_cast<float*>(reinterpret_cast<char*>(workspace) + align256((size_t)HQ * L * 2 * sizeof(float)));
  a.colsum = colsum_out; a.obs = obs_out;
  a.nb = n_bands; a.band_out = n_bands > 0 ? band_out : nullptr;
  for (int b = 0; b < n_bands; b++) a.band[b] = bands[b];  // HOST array: values travel as kernel arguments
  a.HQ = HQ; a.H = H; a.R = R; a.L = L; a.D = D;
  a.QB = kRows / R;
  a.TK = D <= 128 ? 32 : 16;
  a.obs_len = obs_len < 0 ? 0 : (obs_len > L ? L : obs_len);
  a.scale = scale;
  a.obs_plane = -1;
  hipStream_t st = (hipStream_t)stream;
  if (mfma_eligible(HQ, H, D, dtype) && L >= 64) {
    const int nqt = (L + 31) / 32;
    int nwg = nqt < kNWGMfma ? nqt : kNWGMfma;
    static const char* e_nwg = getenv("CC_PREFILL_NWG");  // measurement only
    if (e_nwg && atoi(e_nwg) > 0 && atoi(e_nwg) < nwg) nwg = atoi(e_nwg);
    char* vt = reinterpret_cast<char*>(a.cpart) + align256((size_t)(2 + kMaxBands) * kNWGMfma * H * L * sizeof(float));
    // nobody wants the probabilities (ref: attention_utils.py:27-35, the fused fast path of recent_global / l2 / random / full):
    // ONE pass with online softmax instead of statistics + probabilities
    static const bool two_pass_only = getenv("CC_PREFILL_TWO_PASS") != nullptr;  // measurement only
    if (!a.colsum && !a.obs && !a.band_out && !two_pass_only) {
      static const char* e_nwgf = getenv("CC_PREFILL_NWG_FLASH");  // measurement only
      int nwgf = nqt < 128 ? nqt : 128;
      if (e_nwgf && atoi(e_nwgf) > 0) nwgf = nqt < atoi(e_nwgf) ? nqt : atoi(e_nwgf);
      return cc_prefill_attn_flash_impl(q, k, v, HQ, H, L, D, dtype, scale, y, nwgf, vt, st);
    }
    const int obs_len = a.obs ? a.obs_len : 0;
    int n_partials = nwg;  // partial planes the pass left per output plane: its workgroups per head (two-pass form) or its query segments (r6)
    const int rc = cc_prefill_attn_mfma_impl(q, k, v, HQ, H, L, D, dtype, scale, y, a.stats, a.cpart, nwg, vt, a.band, a.nb, obs_len, st,
                                             kNWGMfma, &n_partials);
    if (rc != CC_OK) return rc;
    a.obs_plane = obs_len > 0 ? 1 + a.nb : -1;
    if (a.colsum || a.obs || a.band_out) return dtype == CC_DT_BF16 ? run_side<bf16_t>(a, n_partials, st) : run_side<f16_t>(a, n_partials, st);
    return CC_OK;
  }
  switch (dtype) {
    case CC_DT_F32: return run_prefill<float>(a, st);
    case CC_DT_BF16: return run_prefill<bf16_t>(a, st);
    default: return run_prefill<f16_t>(a, st);
  }
}

}  // extern "C"
