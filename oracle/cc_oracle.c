/*
 * cc_oracle.c — CPU ORACLE for the cold-compress hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * A plain-C restatement of the reference's algorithms (AnswerDotAI/cold-compress @ 2024-10-22), one
 * function per entry point of include/coldcompress.h, exported with a `_cpu` suffix and taking HOST
 * pointers.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library,
 * and only as the checker / the timed CPU baseline — never as a fallback for the product path.
 *
 * Parity pinning: every function here is checked against golden vectors captured by importing the
 * reference itself (oracle/gen_golden.py -> tests/golden/ *.npz; tests/test_oracle_golden.py).
 * The reference has no tests or vectors of its own.
 *
 * Each function cites the reference lines it restates ("ref:" = path under the reference checkout).
 * Arithmetic notes that matter for bit-exactness are called out inline.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../include/coldcompress.h"

/* ---------------------------------------------------------------- element types ------------------ */

static inline float bf16_to_f32(uint16_t h) {
  uint32_t u = (uint32_t)h << 16;
  float f;
  memcpy(&f, &u, 4);
  return f;
}

/* round-to-nearest-even, NaN preserved as quiet NaN (matches c10::BFloat16 round_to_nearest_even) */
static inline uint16_t f32_to_bf16(float f) {
  uint32_t u;
  memcpy(&u, &f, 4);
  if ((u & 0x7fffffffu) > 0x7f800000u) return 0x7fc0;
  uint32_t lsb = (u >> 16) & 1u;
  u += 0x7fffu + lsb;
  return (uint16_t)(u >> 16);
}

static inline float f16_to_f32(uint16_t h) {
  uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
  uint32_t exp = (h >> 10) & 0x1fu;
  uint32_t man = h & 0x3ffu;
  uint32_t u;
  if (exp == 0) {
    if (man == 0) {
      u = sign;
    } else { /* subnormal */
      int e = -1;
      do {
        man <<= 1;
        e++;
      } while (!(man & 0x400u));
      man &= 0x3ffu;
      u = sign | ((uint32_t)(127 - 15 - e) << 23) | (man << 13);
    }
  } else if (exp == 31) {
    u = sign | 0x7f800000u | (man << 13);
  } else {
    u = sign | ((exp + 112u) << 23) | (man << 13);
  }
  float f;
  memcpy(&f, &u, 4);
  return f;
}

static inline uint16_t f32_to_f16(float f) {
  uint32_t x;
  memcpy(&x, &f, 4);
  uint32_t sign = (x >> 16) & 0x8000u;
  x &= 0x7fffffffu;
  if (x > 0x7f800000u) return (uint16_t)(sign | 0x7e00u);
  if (x >= 0x47800000u) return (uint16_t)(sign | 0x7c00u); /* overflow -> inf (>= 65520 rounds to inf) */
  if (x >= 0x477ff000u) return (uint16_t)(sign | 0x7c00u);
  if (x < 0x33000001u) return (uint16_t)sign; /* underflow to zero (<= 2^-25) */
  uint32_t e = x >> 23;
  uint32_t m = (x & 0x7fffffu) | 0x800000u;
  uint32_t r;
  if (e < 113) { /* subnormal half */
    uint32_t shift = 126 - e; /* 14..24 */
    uint32_t halfm = m >> shift;
    uint32_t rem = m & ((1u << shift) - 1u);
    uint32_t half = 1u << (shift - 1);
    if (rem > half || (rem == half && (halfm & 1u))) halfm++;
    r = halfm;
  } else {
    uint32_t halfm = ((e - 112) << 10) | ((m & 0x7fffffu) >> 13);
    uint32_t rem = m & 0x1fffu;
    if (rem > 0x1000u || (rem == 0x1000u && (halfm & 1u))) halfm++;
    r = halfm;
  }
  return (uint16_t)(sign | r);
}

static inline size_t dt_size(int dt) { return dt == CC_DT_F32 ? 4 : 2; }

static inline float ld(const void* p, int dt, size_t i) {
  switch (dt) {
    case CC_DT_F32: return ((const float*)p)[i];
    case CC_DT_BF16: return bf16_to_f32(((const uint16_t*)p)[i]);
    default: return f16_to_f32(((const uint16_t*)p)[i]);
  }
}

/* round a float to the model dtype and back ("dtype(x)" in the header comments) */
static inline float rnd(float x, int dt) {
  switch (dt) {
    case CC_DT_F32: return x;
    case CC_DT_BF16: return bf16_to_f32(f32_to_bf16(x));
    default: return f16_to_f32(f32_to_f16(x));
  }
}

static inline void st(void* p, int dt, size_t i, float x) {
  switch (dt) {
    case CC_DT_F32: ((float*)p)[i] = x; break;
    case CC_DT_BF16: ((uint16_t*)p)[i] = f32_to_bf16(x); break;
    default: ((uint16_t*)p)[i] = f32_to_f16(x); break;
  }
}

static int dt_ok(int dt) { return dt == CC_DT_F32 || dt == CC_DT_BF16 || dt == CC_DT_F16; }

/* Canonical sum of squares (shared with the device kernels, cc_common.h sumsq_canonical_16): 16 strided
 * partials a_j = sum_i x[j+16i]^2 (sequential in i, separate multiply and add), then the butterfly
 * a_j += a_{j^8}, ^4, ^2, ^1.  torch.linalg.vector_norm's own fp32 order is unspecified (vectorised);
 * the golden tests bound the difference to 1 ulp of the model dtype. */
static float sumsq_canonical(const void* x, int dt, size_t off, int D) {
  float a[16], b[16];
  for (int j = 0; j < 16; j++) {
    float acc = 0.f;
    for (int d = j; d < D; d += 16) {
      float e = ld(x, dt, off + d);
      float sq = e * e;
      acc = acc + sq;
    }
    a[j] = acc;
  }
  for (int k = 8; k > 0; k >>= 1) {
    for (int j = 0; j < 16; j++) b[j] = a[j] + a[j ^ k];
    memcpy(a, b, sizeof(a));
  }
  return a[0];
}

/* ---------------------------------------------------------------- arg-min ------------------------ */

/* torch.argmin over a float row: first index of the minimum, NaN counts as the minimum (first NaN
 * wins) — ref: ATen reduce min with index; probed in SURVEY.md §7. */
static int64_t argmin_f32(const float* x, int n) {
  int64_t best = 0;
  float bv = x[0];
  if (bv != bv) return 0;
  for (int i = 1; i < n; i++) {
    float v = x[i];
    if (v != v) return i;
    if (v < bv) {
      bv = v;
      best = i;
    }
  }
  return best;
}

static int64_t argmin_i32(const int32_t* x, int n) {
  int64_t best = 0;
  for (int i = 1; i < n; i++)
    if (x[i] < x[best]) best = i;
  return best;
}

static int view_ok(const cc_kv_view* c) {
  return c && c->k_cache && c->v_cache && c->pos && c->mask && c->cache_cts && c->H > 0 && c->S > 0 &&
         c->D > 0 && (c->Hp == 1 || c->Hp == c->H) && (c->Hc == 1 || c->Hc == c->H) && dt_ok(c->dtype);
}

/* ref: _decoding_update cache.py:356-362 (num_insertions from the OLD pos), KVCacheHeadConstant._fill
 * :436-437 -> _fill_contiguous :390-401, KVCacheHeadSpecific._fill :460-490, update_kv :330. */
static void insert_token(const cc_kv_view* c, const void* k_new, const void* v_new, int32_t p,
                         const int64_t* idx) {
  const size_t es = dt_size(c->dtype);
  int32_t num_ins[4096];
  for (int hp = 0; hp < c->Hp; hp++) num_ins[hp] = (c->pos[(size_t)hp * c->S + idx[hp]] == -1);
  for (int hp = 0; hp < c->Hp; hp++) c->pos[(size_t)hp * c->S + idx[hp]] = p;
  for (int h = 0; h < c->H; h++) {
    int64_t i = idx[c->Hp == 1 ? 0 : h];
    memcpy((char*)c->k_cache + ((size_t)h * c->S + i) * c->D * es, (const char*)k_new + (size_t)h * c->D * es,
           c->D * es);
    memcpy((char*)c->v_cache + ((size_t)h * c->S + i) * c->D * es, (const char*)v_new + (size_t)h * c->D * es,
           c->D * es);
    c->mask[(size_t)h * c->S + i] = 1;
  }
  /* cache_cts += num_insertions[:len(cache_cts)] with broadcasting when num_insertions has 1 element */
  for (int j = 0; j < c->Hc; j++) c->cache_cts[j] += num_ins[c->Hp == 1 ? 0 : j];
}

int cc_abi_version_cpu(void) { return CC_ABI_VERSION; }

/* Thread count of the OpenMP loops (oracle_lib sets 1 at load: tests stay single-threaded; bench.py's CPU baseline
 * raises it to the host's core count).  Returns the previous maximum. */
#ifdef _OPENMP
#include <omp.h>
/* team size for a loop of n independent iterations: never more threads than iterations (an over-sized team spends its time
 * in the barrier: 256 threads on an 8- or 32-iteration loop measured 18x SLOWER than one thread) */
static int team(int n) {
  const int m = omp_get_max_threads();
  return n < m ? (n > 0 ? n : 1) : m;
}
int cc_oracle_set_threads(int n) {
  const int old = omp_get_max_threads();
  if (n > 0) omp_set_num_threads(n);
  return old;
}
#else
static int team(int n) {
  (void)n;
  return 1;
}
int cc_oracle_set_threads(int n) {
  (void)n;
  return 1;
}
#endif

/* ref: KVCacheFull._eviction_idx cache.py:500-502 */
int cc_decode_update_full_cpu(const cc_kv_view* c, const void* k_new, const void* v_new,
                              const int32_t* input_pos, int64_t* idx_out, cc_stream_t stream) {
  (void)stream;
  if (!view_ok(c) || !input_pos || !idx_out || c->Hp != 1 || c->H > 4096) return CC_ERR_BAD_ARG;
  idx_out[0] = argmin_i32(c->pos, c->S);
  if (k_new) insert_token(c, k_new, v_new, *input_pos, idx_out);
  return CC_OK;
}

/* ref: KVCacheRecentGlobal._eviction_idx cache.py:552-556 */
int cc_decode_update_recent_global_cpu(const cc_kv_view* c, const void* k_new, const void* v_new,
                                       const int32_t* input_pos, int32_t g, int64_t* idx_out,
                                       cc_stream_t stream) {
  (void)stream;
  if (!view_ok(c) || !input_pos || !idx_out || c->Hp != 1 || g < 0 || g >= c->S || c->H > 4096)
    return CC_ERR_BAD_ARG;
  idx_out[0] = argmin_i32(c->pos + g, c->S - g) + g;
  if (k_new) insert_token(c, k_new, v_new, *input_pos, idx_out);
  return CC_OK;
}

/* ref: KVCache._eviction_idx cache.py:366-379 applied to one row of scores */
static int64_t base_evict_row(float* sc, const int32_t* pos, int S, int g) {
  for (int s = 0; s < g && s < S; s++) sc[s] = INFINITY; /* :373 protects the first g SLOTS */
  for (int s = 0; s < S; s++)
    if (pos[s] == -1) sc[s] = -INFINITY; /* :376 */
  return argmin_f32(sc, S);               /* :379 */
}

int cc_decode_update_scores_cpu(const cc_kv_view* c, const void* k_new, const void* v_new,
                                const int32_t* input_pos, const void* scores, int32_t score_dtype,
                                int32_t g, int64_t* idx_out, cc_stream_t stream) {
  (void)stream;
  if (!view_ok(c) || !input_pos || !idx_out || !scores || !dt_ok(score_dtype) || g < 0 || c->H > 4096)
    return CC_ERR_BAD_ARG;
  float* sc = (float*)malloc(sizeof(float) * (size_t)c->S);
  for (int hp = 0; hp < c->Hp; hp++) {
    for (int s = 0; s < c->S; s++) sc[s] = ld(scores, score_dtype, (size_t)hp * c->S + s);
    idx_out[hp] = base_evict_row(sc, c->pos + (size_t)hp * c->S, c->S, g);
  }
  free(sc);
  if (k_new) insert_token(c, k_new, v_new, *input_pos, idx_out);
  return CC_OK;
}

/* ref: KVCacheRandom._token_importances cache.py:519-524, then base rules :366-379 */
int cc_decode_update_random_cpu(const cc_kv_view* c, const void* k_new, const void* v_new,
                                const int32_t* input_pos, const float* rand_u, int32_t g, int32_t w,
                                int64_t* idx_out, cc_stream_t stream) {
  (void)stream;
  if (!view_ok(c) || !input_pos || !idx_out || !rand_u || c->Hp != 1 || g < 0 || c->H > 4096)
    return CC_ERR_BAD_ARG;
  const int32_t p = *input_pos;
  float* sc = (float*)malloc(sizeof(float) * (size_t)c->S);
  for (int s = 0; s < c->S; s++) sc[s] = (c->pos[s] >= p - w) ? INFINITY : rand_u[s]; /* :523 */
  idx_out[0] = base_evict_row(sc, c->pos, c->S, g);
  free(sc);
  if (k_new) insert_token(c, k_new, v_new, p, idx_out);
  return CC_OK;
}

size_t cc_decode_update_l2_workspace_bytes_cpu(int32_t H, int32_t S) {
  (void)H;
  (void)S;
  return 0;
}

/* ref: KVCacheL2._token_importances cache.py:597-605, _decoding_update :580-595.
 * The score is a model-dtype tensor: (max - norm) is evaluated in fp32 and rounded to the dtype, which
 * manufactures ties; the max is over ALL heads and slots (:602). */
int cc_decode_update_l2_cpu(const cc_kv_view* c, const void* k_new, const void* v_new,
                            const int32_t* input_pos, void* key_norm, int32_t g, int32_t w,
                            int64_t* idx_out, void* workspace, size_t workspace_bytes,
                            cc_stream_t stream) {
  (void)stream;
  (void)workspace;
  (void)workspace_bytes;
  if (!view_ok(c) || !input_pos || !idx_out || !key_norm || c->Hp != c->H || g < 0 || c->H > 4096)
    return CC_ERR_BAD_ARG;
  const int32_t p = *input_pos;
  const int dt = c->dtype;
  float mx = -INFINITY;
  int has_nan = 0;
  for (size_t i = 0; i < (size_t)c->H * c->S; i++) {
    float v = ld(key_norm, dt, i);
    if (v != v) has_nan = 1;
    if (v > mx) mx = v;
  }
  if (has_nan) mx = NAN; /* torch.max propagates NaN */
  float* sc = (float*)malloc(sizeof(float) * (size_t)c->S);
  for (int h = 0; h < c->H; h++) {
    const int32_t* pos = c->pos + (size_t)h * c->S;
    for (int s = 0; s < c->S; s++) {
      float v = rnd(mx - ld(key_norm, dt, (size_t)h * c->S + s), dt);
      sc[s] = (pos[s] >= p - w) ? INFINITY : v; /* :603 */
    }
    idx_out[h] = base_evict_row(sc, pos, c->S, g);
  }
  free(sc);
  if (k_new) {
    insert_token(c, k_new, v_new, p, idx_out);
    for (int h = 0; h < c->H; h++) { /* :592-593 vector_norm in fp32 opmath, stored in model dtype */
      float acc = sumsq_canonical(k_new, dt, (size_t)h * c->D, c->D);
      st(key_norm, dt, (size_t)h * c->S + idx_out[h], sqrtf(acc));
    }
  }
  return CC_OK;
}

/* ref: KVCacheHeavyHitter._eviction_idx cache.py:725-765 with history_window_size == 1.
 *  :727 numerator = num.sum(-1).float()  -> (float)double, round-to-nearest-even
 *  :732 denominator = denom.clamp_min(1) (int32); :738 float32 / int32 -> IEEE fp32 divide
 *  :741-747 (pos < g) | (pos >= p - w) -> 1.0 ; :749 pos == -1 -> 0.0 (applied last, so it wins)
 *  :751 argmin ; :754-763 zero num/denom at idx. */
int cc_decode_update_heavy_hitter_cpu(const cc_kv_view* c, const void* k_new, const void* v_new,
                                      const int32_t* input_pos, double* num, int32_t* denom, int32_t g,
                                      int32_t w, int64_t* idx_out, cc_stream_t stream) {
  (void)stream;
  if (!view_ok(c) || !input_pos || !idx_out || !num || !denom || c->Hp != c->H || c->H > 4096)
    return CC_ERR_BAD_ARG;
  const int32_t p = *input_pos;
  /* heads are independent (cache.py:725-765): one OpenMP thread per head when built with -fopenmp (bench.py's CPU
   * baseline); the result does not depend on the thread count */
#pragma omp parallel for schedule(static) num_threads(team(c->H))
  for (int h = 0; h < c->H; h++) {
    float* sc = (float*)malloc(sizeof(float) * (size_t)c->S);
    const int32_t* pos = c->pos + (size_t)h * c->S;
    for (int s = 0; s < c->S; s++) {
      size_t i = (size_t)h * c->S + s;
      int32_t dn = denom[i] < 1 ? 1 : denom[i];
      float v = (float)num[i] / (float)dn;
      if (pos[s] < g || pos[s] >= p - w) v = 1.0f;
      if (pos[s] == -1) v = 0.0f;
      sc[s] = v;
    }
    idx_out[h] = argmin_f32(sc, c->S);
    free(sc);
  }
  for (int h = 0; h < c->H; h++) {
    num[(size_t)h * c->S + idx_out[h]] = 0.0;
    denom[(size_t)h * c->S + idx_out[h]] = 0;
  }
  if (k_new) insert_token(c, k_new, v_new, p, idx_out);
  return CC_OK;
}

/* ref: KVCacheHeavyHitter.update_state cache.py:706-723 (W == 1): zero-pad attn to S, num += attn
 * (float64 += widened model dtype: exact), denom += 1 everywhere, counter += 1. */
int cc_hh_update_cpu(double* num, int32_t* denom, int64_t* counter, const void* attn, int32_t H, int32_t S,
                     int32_t T, int32_t dtype, cc_stream_t stream) {
  (void)stream;
  if (!num || !denom || !attn || H <= 0 || S <= 0 || T < 0 || T > S || !dt_ok(dtype)) return CC_ERR_BAD_ARG;
  for (int h = 0; h < H; h++) {
    for (int s = 0; s < T; s++) num[(size_t)h * S + s] += (double)ld(attn, dtype, (size_t)h * T + s);
    for (int s = 0; s < S; s++) denom[(size_t)h * S + s] += 1;
  }
  if (counter) *counter += 1;
  return CC_OK;
}

/* ---------------------------------------------------------------- decode attention ---------------- */

size_t cc_decode_attn_workspace_bytes_cpu(int32_t HQ, int32_t H, int32_t S, int32_t D, int32_t dtype) {
  (void)HQ;
  (void)H;
  (void)S;
  (void)D;
  (void)dtype;
  return 0;
}

/* ref: model.py:395-418 + attention_utils.py:36-54 (naive path; the fused F.sdpa path :27-35 is the
 * same math without returning P).  Rounding points of the reference in a 16-bit model dtype:
 *   :37 (q @ k^T) -> dtype ; * scale_factor -> dtype ; :42-43 + (-inf) bias ; :52 softmax (fp32 inside,
 *   result -> dtype) ; :54 P @ v -> dtype ; model.py:416-418 mean over the R query heads -> dtype.
 * Contractions are accumulated in double here (the exactly-rounded value every fp32 summation order
 * approximates); comparisons against the device kernel and the reference are tolerance-based (1e-3). */
int cc_decode_attn_gqa_cpu(const void* q, const void* k, const void* v, const uint8_t* mask, int32_t HQ,
                           int32_t H, int32_t S, int32_t D, int32_t dtype, float scale, void* y,
                           void* attn_out, void* probs_out, double* hh_num, int32_t* hh_denom,
                           int64_t* hh_counter, void* workspace, size_t workspace_bytes,
                           cc_stream_t stream) {
  (void)stream;
  (void)workspace;
  (void)workspace_bytes;
  if (!q || !k || !v || !y || HQ <= 0 || H <= 0 || HQ % H || S <= 0 || D <= 0 || !dt_ok(dtype))
    return CC_ERR_BAD_ARG;
  const int R = HQ / H;
  /* Every (kv head, query head) pair is independent up to the group mean: OpenMP over the HQ query heads first (bench.py's
   * CPU baseline runs this with OMP_NUM_THREADS = nproc), then over the kv heads for the group mean / history.  Each
   * value is computed by exactly one thread with the same arithmetic: results do not depend on the thread count. */
  float* Pall = (float*)malloc(sizeof(float) * (size_t)HQ * S);
#pragma omp parallel for schedule(static) num_threads(team(HQ))
  for (int j = 0; j < HQ; j++) {
    const int h = j / R;
    float* P = Pall + (size_t)(j - h * R) * S + (size_t)h * R * S;
    float* sc = (float*)malloc(sizeof(float) * (size_t)S);
    {
      const int r = j - h * R;
      (void)r;
      float m = -INFINITY;
      for (int s = 0; s < S; s++) {
        double acc = 0.0;
        for (int d = 0; d < D; d++)
          acc += (double)ld(q, dtype, (size_t)j * D + d) * (double)ld(k, dtype, ((size_t)h * S + s) * D + d);
        float x = rnd(rnd((float)acc, dtype) * scale, dtype);
        if (mask && !mask[(size_t)h * S + s]) x = -INFINITY;
        sc[s] = x;
        if (x > m) m = x;
      }
      double sum = 0.0;
      for (int s = 0; s < S; s++) {
        sc[s] = expf(sc[s] - m);
        sum += sc[s];
      }
      const float fsum = (float)sum;
      for (int s = 0; s < S; s++) {
        float pr = rnd(sc[s] / fsum, dtype);
        P[s] = pr;
        if (probs_out) st(probs_out, dtype, (size_t)j * S + s, pr);
      }
      for (int d = 0; d < D; d++) {
        double acc = 0.0;
        for (int s = 0; s < S; s++) acc += (double)P[s] * (double)ld(v, dtype, ((size_t)h * S + s) * D + d);
        st(y, dtype, (size_t)j * D + d, (float)acc);
      }
    }
    free(sc);
  }
#pragma omp parallel for schedule(static) num_threads(team(H))
  for (int h = 0; h < H; h++) {
    const float* P = Pall + (size_t)h * R * S;
    if (attn_out || hh_num) {
      for (int s = 0; s < S; s++) {
        float acc = 0.f;
        for (int r = 0; r < R; r++) acc += P[(size_t)r * S + s];
        float a = rnd(acc / (float)R, dtype);
        if (attn_out) st(attn_out, dtype, (size_t)h * S + s, a);
        if (hh_num) {
          hh_num[(size_t)h * S + s] += (double)a;
          hh_denom[(size_t)h * S + s] += 1;
        }
      }
    }
  }
  if (hh_num && hh_counter) *hh_counter += 1;
  free(Pall);
  return CC_OK;
}

/* ---------------------------------------------------------------- prefill fill / norms ------------ */

/* ref: _prefill_update cache.py:381-388, _fill_contiguous :390-401, update_kv :330 */
int cc_prefill_fill_cpu(const cc_kv_view* c, const void* k_val, const void* v_val, const int64_t* pos_val,
                        int32_t PH, int32_t T, cc_stream_t stream) {
  (void)stream;
  if (!view_ok(c) || !k_val || !v_val || !pos_val || T <= 0 || T > c->S || (PH != 1 && PH != c->Hp))
    return CC_ERR_BAD_ARG;
  const size_t es = dt_size(c->dtype);
  for (int hp = 0; hp < c->Hp; hp++)
    for (int t = 0; t < T; t++) c->pos[(size_t)hp * c->S + t] = (int32_t)pos_val[(size_t)(PH == 1 ? 0 : hp) * T + t];
  for (int h = 0; h < c->H; h++) {
    memcpy((char*)c->k_cache + (size_t)h * c->S * c->D * es, (const char*)k_val + (size_t)h * T * c->D * es,
           (size_t)T * c->D * es);
    memcpy((char*)c->v_cache + (size_t)h * c->S * c->D * es, (const char*)v_val + (size_t)h * T * c->D * es,
           (size_t)T * c->D * es);
    memset(c->mask + (size_t)h * c->S, 1, (size_t)T);
  }
  for (int j = 0; j < c->Hc; j++) c->cache_cts[j] += T;
  return CC_OK;
}

/* ref: torch.linalg.vector_norm(x, ord=2, dim=-1) at cache.py:612 and prompt_compression.py:203 */
int cc_row_l2_norm_cpu(const void* x, int32_t H, int32_t N, int32_t D, int32_t dtype, int32_t negate,
                       void* out, cc_stream_t stream) {
  (void)stream;
  if (!x || !out || H <= 0 || N <= 0 || D <= 0 || !dt_ok(dtype)) return CC_ERR_BAD_ARG;
  for (size_t r = 0; r < (size_t)H * N; r++) {
    float n = sqrtf(sumsq_canonical(x, dtype, r * D, D));
    st(out, dtype, r, negate ? -n : n);
  }
  return CC_OK;
}

/* ---------------------------------------------------------------- prompt compaction --------------- */

typedef struct {
  double v; /* every priority dtype (f32/bf16/f16/int64 < 2^53 in practice) is exact in a double */
  int isnan;
  int64_t i;
} prio_t;

static int prio_cmp(const void* a, const void* b) {
  const prio_t* x = (const prio_t*)a;
  const prio_t* y = (const prio_t*)b;
  if (x->isnan != y->isnan) return y->isnan - x->isnan; /* NaN ranks as the largest (torch.topk) */
  if (!x->isnan) {
    if (x->v > y->v) return -1;
    if (x->v < y->v) return 1;
  }
  return (x->i > y->i) - (x->i < y->i); /* tie: lowest index first (see DESIGN.md tie contract) */
}

static int i64_cmp(const void* a, const void* b) {
  int64_t x = *(const int64_t*)a, y = *(const int64_t*)b;
  return (x > y) - (x < y);
}

size_t cc_topk_keep_workspace_bytes_cpu(int32_t Hs, int32_t L, int32_t K) {
  (void)Hs;
  (void)L;
  (void)K;
  return 0;
}

/* ref: PromptCompressor._keep_idxs prompt_compression.py:21-26: topk(K).indices.sort().values */
int cc_topk_keep_cpu(const void* priority, int32_t prio_dtype, int32_t Hs, int32_t L, int32_t K,
                     int64_t* keep_out, void* workspace, size_t workspace_bytes, cc_stream_t stream) {
  (void)stream;
  (void)workspace;
  (void)workspace_bytes;
  if (!priority || !keep_out || Hs <= 0 || L <= 0 || K <= 0 || K > L || prio_dtype < 0 || prio_dtype > 3)
    return CC_ERR_BAD_ARG;
  prio_t* a = (prio_t*)malloc(sizeof(prio_t) * (size_t)L);
  for (int h = 0; h < Hs; h++) {
    for (int t = 0; t < L; t++) {
      size_t i = (size_t)h * L + t;
      double v;
      if (prio_dtype == CC_PRIO_I64) v = (double)((const int64_t*)priority)[i];
      else v = (double)ld(priority, prio_dtype, i);
      a[t].v = v;
      a[t].isnan = (v != v);
      a[t].i = t;
    }
    qsort(a, (size_t)L, sizeof(prio_t), prio_cmp);
    for (int j = 0; j < K; j++) keep_out[(size_t)h * K + j] = a[j].i;
    qsort(keep_out + (size_t)h * K, (size_t)K, sizeof(int64_t), i64_cmp);
  }
  free(a);
  return CC_OK;
}

/* ref: _filter_kv prompt_compression.py:69-72 (head-constant index) / :82-88 (head-specific gather) */
int cc_gather_rows_cpu(const void* src, const int64_t* keep, int32_t Hk, int32_t H, int32_t L, int32_t K,
                       int32_t D, int32_t dtype, void* dst, cc_stream_t stream) {
  (void)stream;
  if (!src || !keep || !dst || H <= 0 || L <= 0 || K <= 0 || D <= 0 || (Hk != 1 && Hk != H) || !dt_ok(dtype))
    return CC_ERR_BAD_ARG;
  const size_t rb = (size_t)D * dt_size(dtype);
  for (int h = 0; h < H; h++)
    for (int j = 0; j < K; j++) {
      int64_t t = keep[(size_t)(Hk == 1 ? 0 : h) * K + j];
      if (t < 0 || t >= L) return CC_ERR_BAD_ARG;
      memcpy((char*)dst + ((size_t)h * K + j) * rb, (const char*)src + ((size_t)h * L + t) * rb, rb);
    }
  return CC_OK;
}

/* ref: cum_attn.gather(2, keep_idxs) prompt_compression.py:193 */
int cc_gather_vec_cpu(const void* src, const int64_t* keep, int32_t Hs, int32_t L, int32_t K, int32_t dtype,
                      void* dst, cc_stream_t stream) {
  (void)stream;
  if (!src || !keep || !dst || Hs <= 0 || L <= 0 || K <= 0 || !dt_ok(dtype)) return CC_ERR_BAD_ARG;
  const size_t es = dt_size(dtype);
  for (int h = 0; h < Hs; h++)
    for (int j = 0; j < K; j++) {
      int64_t t = keep[(size_t)h * K + j];
      if (t < 0 || t >= L) return CC_ERR_BAD_ARG;
      memcpy((char*)dst + ((size_t)h * K + j) * es, (const char*)src + ((size_t)h * L + t) * es, es);
    }
  return CC_OK;
}

/* ref: KVCacheAnalysis.update_state cache.py:1391-1404 (decode): indices[indices == -1] = S_full - 1;
 * attn_compressed = attn.gather(indices); loss = (1 - attn_compressed.sum(-1)).mean() in the model dtype. */
int cc_analysis_loss_cpu(const void* attn, const int32_t* pos, int32_t Hp, int32_t S_full, int32_t S, int32_t dtype, void* sub_out,
                         void* losses, int32_t* loss_ctr, int32_t cap, cc_stream_t stream) {
  (void)stream;
  if (!attn || !pos || !sub_out || !losses || !loss_ctr || Hp <= 0 || Hp > 64 || S_full <= 0 || S <= 0 || cap <= 0 || !dt_ok(dtype))
    return CC_ERR_BAD_ARG;
  float tot = 0.f;
  for (int h = 0; h < Hp; h++) {
    float acc = 0.f;
    for (int s = 0; s < S; s++) {
      const int p = pos[(size_t)h * S + s];
      const float v = ld(attn, dtype, (size_t)h * S_full + (p == -1 ? S_full - 1 : p));
      st(sub_out, dtype, (size_t)h * S + s, v);
      acc += v;
    }
    tot += rnd(1.0f - rnd(acc, dtype), dtype);
  }
  const int c = *loss_ctr;
  if (c >= 0 && c < cap) st(losses, dtype, (size_t)c, tot / (float)Hp);
  *loss_ctr = c + 1;
  return CC_OK;
}

/* ref: PromptCompressorHeavyHitter._token_importances prompt_compression.py:170-187.
 * AvgPool1d(kernel 5, stride 1, padding 2, count_include_pad=False): mean of the in-range neighbours,
 * fp32 accumulate, rounded to the model dtype; then the observation window and the global tokens are
 * forced to 1.0 (:182-186). */
int cc_snapkv_priority_cpu(const void* obs_mean, int32_t H, int32_t L, int32_t dtype, int32_t obs_len,
                           int32_t g, void* out, cc_stream_t stream) {
  (void)stream;
  if (!obs_mean || !out || H <= 0 || L <= 0 || !dt_ok(dtype) || obs_len < 0) return CC_ERR_BAD_ARG;
  for (int h = 0; h < H; h++)
    for (int t = 0; t < L; t++) {
      int lo = t - 2 < 0 ? 0 : t - 2, hi = t + 2 >= L ? L - 1 : t + 2;
      float acc = 0.f;
      for (int u = lo; u <= hi; u++) acc += ld(obs_mean, dtype, (size_t)h * L + u);
      float v = acc / (float)(hi - lo + 1);
      if (t >= L - obs_len || t < g) v = 1.0f;
      st(out, dtype, (size_t)h * L + t, v);
    }
  return CC_OK;
}

/* ---------------------------------------------------------------- prefill attention --------------- */

size_t cc_prefill_attn_workspace_bytes_cpu(int32_t HQ, int32_t H, int32_t L, int32_t D, int32_t dtype) {
  (void)HQ;
  (void)H;
  (void)L;
  (void)D;
  (void)dtype;
  return 0;
}

/* ref: attention_utils.py:36-54 with the causal mask of generation_utils.py:153-158, model.py:413-418
 * (group mean), cache.py:704 / prompt_compression.py:191 (column sums) and prompt_compression.py:173
 * (mean of the last obs_len query rows).  O(L^2 D) — small L only. */
/* Core shared by cc_prefill_attn_cpu and cc_prefill_attn_bands_cpu.  Query rows are processed in blocks of kPfBlock: the
 * rows of a block are independent (one OpenMP thread each: scores, softmax, probabilities, y, group mean -> A[row][s]),
 * then every COLUMN s accumulates its block of rows sequentially in query order — the canonical order of the column,
 * observation-window and band sums (one thread per column; the result does not depend on the thread count). */
enum { kPfBlock = 256 };
#ifdef _OPENMP
static int pf_thread(void) { return omp_get_thread_num(); }
static int pf_max_threads(void) { return omp_get_max_threads(); }
#else
static int pf_thread(void) { return 0; }
static int pf_max_threads(void) { return 1; }
#endif
static int prefill_core(const void* q, const void* k, const void* v, int HQ, int H, int L, int D, int dtype, float scale, void* y,
                        float* colsum_out, float* obs_out, int obs_len, const int32_t* bands, int n_bands, float* band_out,
                        float* attn_full) {
  const int R = HQ / H;
  if (obs_len > L) obs_len = L;
  if (colsum_out) memset(colsum_out, 0, sizeof(float) * (size_t)H * L);
  if (obs_out) memset(obs_out, 0, sizeof(float) * (size_t)H * L);
  if (band_out) memset(band_out, 0, sizeof(float) * (size_t)n_bands * H * L);
  float* kf = (float*)malloc(sizeof(float) * (size_t)L * D);
  float* vf = (float*)malloc(sizeof(float) * (size_t)L * D);
  float* A = (float*)malloc(sizeof(float) * (size_t)kPfBlock * L);
  const size_t per_thread = (size_t)R * L + L + D + 2 * (size_t)D + 2;  /* P, sc, qf, and D doubles (8-byte aligned below) for y */
  float* scratch = (float*)malloc(sizeof(float) * per_thread * (size_t)pf_max_threads());
  const int need_a = colsum_out || obs_out || band_out || attn_full;
  for (int h = 0; h < H; h++) {
    for (size_t e = 0; e < (size_t)L * D; e++) {
      kf[e] = ld(k, dtype, (size_t)h * L * D + e);
      vf[e] = ld(v, dtype, (size_t)h * L * D + e);
    }
    for (int i0 = 0; i0 < L; i0 += kPfBlock) {
      const int nb = L - i0 < kPfBlock ? L - i0 : kPfBlock;
#pragma omp parallel for schedule(dynamic, 1)
      for (int bi = 0; bi < nb; bi++) {
        const int i = i0 + bi;
        /* per-thread scratch from one allocation per call (per-row allocations of this size go through mmap and
         * serialise hundreds of threads in the kernel) */
        float* P = scratch + (size_t)pf_thread() * per_thread;
        float* sc = P + (size_t)R * L;
        float* qf = sc + L;
        double* yacc = (double*)(((uintptr_t)(qf + D) + 7) & ~(uintptr_t)7);
        for (int r = 0; r < R; r++) {
          const int j = h * R + r;
          for (int d = 0; d < D; d++) qf[d] = ld(q, dtype, ((size_t)j * L + i) * D + d);
          float m = -INFINITY;
          /* four slots at a time: four INDEPENDENT double accumulators, each adding its slot's products in d order — the value of
           * every score is what the one-slot loop gives (checked bit for bit), the four dependent add chains overlap in the pipeline */
          int s4 = 0;
          for (; s4 + 3 <= i; s4 += 4) {
            const float* k0 = kf + (size_t)s4 * D;
            const float *k1 = k0 + D, *k2 = k0 + 2 * D, *k3 = k0 + 3 * D;
            double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
            for (int d = 0; d < D; d++) {
              const double qd = (double)qf[d];
              a0 += qd * (double)k0[d];
              a1 += qd * (double)k1[d];
              a2 += qd * (double)k2[d];
              a3 += qd * (double)k3[d];
            }
            const double a[4] = {a0, a1, a2, a3};
            for (int u = 0; u < 4; u++) {
              float x = rnd(rnd((float)a[u], dtype) * scale, dtype);
              sc[s4 + u] = x;
              if (x > m) m = x;
            }
          }
          for (int s = s4; s <= i; s++) {
            double acc = 0.0;
            const float* kr = kf + (size_t)s * D;
            for (int d = 0; d < D; d++) acc += (double)qf[d] * (double)kr[d];
            float x = rnd(rnd((float)acc, dtype) * scale, dtype);
            sc[s] = x;
            if (x > m) m = x;
          }
          double sum = 0.0;
          for (int s = 0; s <= i; s++) {
            sc[s] = expf(sc[s] - m);
            sum += sc[s];
          }
          const float fsum = (float)sum;
          float* Pr = P + (size_t)r * (i + 1);
          for (int s = 0; s <= i; s++) Pr[s] = rnd(sc[s] / fsum, dtype);
          /* y[d] = sum over s (in slot order) of P[s] * V[s][d], accumulated in double: s is the OUTER loop so that V is read
           * row by row (contiguous) — every y[d] still adds the same products in the same order (bit-identical to a d-outer loop,
           * checked; 2-3x faster at L >= 8k where a column walk of V misses the cache on every element) */
          for (int d = 0; d < D; d++) yacc[d] = 0.0;
          for (int s = 0; s <= i; s++) {
            const double ps = (double)Pr[s];
            const float* vr = vf + (size_t)s * D;
            for (int d = 0; d < D; d++) yacc[d] += ps * (double)vr[d];
          }
          for (int d = 0; d < D; d++) st(y, dtype, ((size_t)j * L + i) * D + d, (float)yacc[d]);
        }
        if (need_a) {
          float* Ar = A + (size_t)bi * L;
          for (int s = 0; s <= i; s++) {
            float acc = 0.f;
            for (int r = 0; r < R; r++) acc += P[(size_t)r * (i + 1) + s];
            Ar[s] = rnd(acc / (float)R, dtype);
          }
        }
      }
      if (attn_full)  /* the group-averaged probabilities themselves, [H, L, L], zero above the diagonal */
        for (int bi = 0; bi < nb; bi++) {
          float* dst = attn_full + ((size_t)h * L + i0 + bi) * L;
          memcpy(dst, A + (size_t)bi * L, sizeof(float) * (size_t)(i0 + bi + 1));
          memset(dst + i0 + bi + 1, 0, sizeof(float) * (size_t)(L - i0 - bi - 1));
        }
      if (need_a) {
#pragma omp parallel for schedule(static)
        for (int s = 0; s < i0 + nb; s++) {
          for (int bi = (s > i0 ? s - i0 : 0); bi < nb; bi++) {  // rows i >= s only (causal)
            const int i = i0 + bi;
            const float a = A[(size_t)bi * L + s];
            if (colsum_out) colsum_out[(size_t)h * L + s] += a;
            if (obs_out && i >= L - obs_len) obs_out[(size_t)h * L + s] += a;
            for (int b = 0; b < n_bands; b++)
              if (i - s < bands[b]) band_out[((size_t)b * H + h) * L + s] += a;
          }
        }
      }
    }
  }
  if (obs_out && obs_len > 0)
    for (size_t i = 0; i < (size_t)H * L; i++) obs_out[i] /= (float)obs_len;
  free(kf);
  free(vf);
  free(A);
  free(scratch);
  return CC_OK;
}

int cc_prefill_attn_cpu(const void* q, const void* k, const void* v, int32_t HQ, int32_t H, int32_t L,
                        int32_t D, int32_t dtype, float scale, void* y, float* colsum_out, float* obs_out,
                        int32_t obs_len, void* workspace, size_t workspace_bytes, cc_stream_t stream) {
  (void)stream;
  (void)workspace;
  (void)workspace_bytes;
  if (!q || !k || !v || !y || HQ <= 0 || H <= 0 || HQ % H || L <= 0 || D <= 0 || !dt_ok(dtype))
    return CC_ERR_BAD_ARG;
  return prefill_core(q, k, v, HQ, H, L, D, dtype, scale, y, colsum_out, obs_out, obs_len, NULL, 0, NULL, NULL);
}

/* ref: attn.squeeze(0).sum(dim=1) cache.py:704 */
int cc_attn_colsum_cpu(const void* attn, int32_t H, int32_t Lq, int32_t Lk, int32_t dtype, float* out,
                       cc_stream_t stream) {
  (void)stream;
  if (!attn || !out || H <= 0 || Lq <= 0 || Lk <= 0 || !dt_ok(dtype)) return CC_ERR_BAD_ARG;
  for (int h = 0; h < H; h++)
    for (int s = 0; s < Lk; s++) {
      float acc = 0.f;
      for (int i = 0; i < Lq; i++) acc += ld(attn, dtype, ((size_t)h * Lq + i) * Lk + s);
      out[(size_t)h * Lk + s] = acc;
    }
  return CC_OK;
}

/* ref: cache.py:704, prompt_compression.py:191 */
int cc_colsum_to_mean_cpu(const float* colsum, const int64_t* input_pos, int32_t H, int32_t L, int32_t dtype,
                          void* out, cc_stream_t stream) {
  (void)stream;
  if (!colsum || !out || H <= 0 || L <= 0 || !dt_ok(dtype)) return CC_ERR_BAD_ARG;
  for (int h = 0; h < H; h++)
    for (int t = 0; t < L; t++) {
      int64_t p = input_pos ? input_pos[t] : t;
      st(out, dtype, (size_t)h * L + t, rnd(colsum[(size_t)h * L + t], dtype) / (float)(L - p));
    }
  return CC_OK;
}

/* measurement hook twin: the oracle has no phases; phase 2 (or 3) computes everything, phase 1 alone nothing */
int cc_decode_attn_gqa_phases_cpu(const void* q, const void* k, const void* v, const uint8_t* mask, int32_t HQ,
                                  int32_t H, int32_t S, int32_t D, int32_t dtype, float scale, void* y,
                                  void* attn_out, void* probs_out, double* hh_num, int32_t* hh_denom,
                                  int64_t* hh_counter, void* workspace, size_t workspace_bytes,
                                  cc_stream_t stream, int32_t phases) {
  if (!(phases & 2)) return CC_OK;
  return cc_decode_attn_gqa_cpu(q, k, v, mask, HQ, H, S, D, dtype, scale, y, attn_out, probs_out, hh_num, hh_denom,
                                hh_counter, workspace, workspace_bytes, stream);
}

/* ---------------------------------------------------------------- caller glue ---------------------- */

/* ref: model.py:452-457 RMSNorm (fp32 inside, cast, * weight) and the residual add of :325-326 */
int cc_add_rmsnorm_cpu(const void* x, const void* delta, const void* weight, int32_t T, int32_t dim, float eps,
                       int32_t dtype, void* h_out, void* out, cc_stream_t stream) {
  (void)stream;
  if (!x || !weight || !out || T <= 0 || dim <= 0 || !dt_ok(dtype)) return CC_ERR_BAD_ARG;
  float* h = (float*)malloc(sizeof(float) * (size_t)dim);
  for (int t = 0; t < T; t++) {
    double ss = 0.0;
    for (int i = 0; i < dim; i++) {
      float v = ld(x, dtype, (size_t)t * dim + i);
      if (delta) v = rnd(v + ld(delta, dtype, (size_t)t * dim + i), dtype);
      h[i] = v;
      ss += (double)v * (double)v;
    }
    if (delta && h_out)
      for (int i = 0; i < dim; i++) st(h_out, dtype, (size_t)t * dim + i, h[i]);
    const float rs = 1.0f / sqrtf((float)(ss / dim) + eps);
    for (int i = 0; i < dim; i++) st(out, dtype, (size_t)t * dim + i, rnd(h[i] * rs, dtype) * ld(weight, dtype, i));
  }
  free(h);
  return CC_OK;
}

/* ref: model.py:375-387 (split, view, transpose) + apply_rotary_emb :507-519 */
int cc_qkv_rope_cpu(const void* qkv, const void* freqs, int32_t T, int32_t HQ, int32_t H, int32_t D, int32_t dtype,
                    void* q_out, void* k_out, void* v_out, cc_stream_t stream) {
  (void)stream;
  if (!qkv || !freqs || !q_out || !k_out || !v_out || T <= 0 || HQ <= 0 || H <= 0 || D <= 0 || (D & 1) || !dt_ok(dtype))
    return CC_ERR_BAD_ARG;
  const int heads = HQ + 2 * H, half = D / 2;
  for (int t = 0; t < T; t++)
    for (int hd = 0; hd < heads; hd++) {
      void* dst = hd < HQ ? q_out : (hd < HQ + H ? k_out : v_out);
      const int hh = hd < HQ ? hd : (hd < HQ + H ? hd - HQ : hd - HQ - H);
      for (int p = 0; p < half; p++) {
        const size_t src = ((size_t)t * heads + hd) * D + 2 * p, o = ((size_t)hh * T + t) * D + 2 * p;
        const float x0 = ld(qkv, dtype, src), x1 = ld(qkv, dtype, src + 1);
        if (hd < HQ + H) {
          const float c = ld(freqs, dtype, ((size_t)t * half + p) * 2), s = ld(freqs, dtype, ((size_t)t * half + p) * 2 + 1);
          const float a0 = x0 * c, a1 = x1 * s, b0 = x1 * c, b1 = x0 * s;
          st(dst, dtype, o, a0 - a1);
          st(dst, dtype, o + 1, b0 + b1);
        } else {
          st(dst, dtype, o, x0);
          st(dst, dtype, o + 1, x1);
        }
      }
    }
  return CC_OK;
}

/* ref: model.py:442-443 F.silu(w1 x) * w3 x */
int cc_silu_mul_cpu(const void* a, const void* b, int64_t n, int32_t dtype, void* out, cc_stream_t stream) {
  (void)stream;
  if (!a || !b || !out || n <= 0 || !dt_ok(dtype)) return CC_ERR_BAD_ARG;
  for (int64_t i = 0; i < n; i++) {
    const float x = ld(a, dtype, (size_t)i);
    st(out, dtype, (size_t)i, rnd(x / (1.0f + expf(-x)), dtype) * ld(b, dtype, (size_t)i));
  }
  return CC_OK;
}

/* ---------------------------------------------------------------- hybrid (FastGen) ----------------- */

enum { HF_HH = 1, HF_WIN = 2, HF_PUNC = 4, HF_SPECIAL = 8, HF_FULL = 16 };

/* dtype(sum_W row) (cache.py:855-859 `.sum(dim=-1)` on a model-dtype tensor; torch's fp32 summation order is
 * unspecified and backend-specific).  Definition shared with the device (cc_hybrid.hip): the EXACT sum of the W ring
 * entries, rounded once, nearest-even, to the model dtype — independent of summation order, which is what allows the
 * device to keep it incrementally.  Restated here with a different mechanism than the device's 192-bit adds: signed
 * base-2^32 bins with deferred carries, then a bit-serial rounding.
 * acc[0..2]: the sum as a 192-bit two's-complement integer in units of 2^-149; acc[3]: number of entries with
 * |v| >= 4 or non-finite (such a row sums to NaN). */
static void window_acc_row(const void* num, int dt, size_t off, int W, uint64_t acc[4]) {
  int64_t bin[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  uint64_t special = 0;
  for (int j = 0; j < W; j++) {
    const float v = ld(num, dt, off + (size_t)j);
    uint32_t u;
    memcpy(&u, &v, 4);
    if ((u << 1) == 0) continue;
    const uint32_t E = (u >> 23) & 0xffu, M = u & 0x7fffffu;
    if (E >= 129) {
      special++;
      continue;
    }
    const uint64_t m = E ? (uint64_t)(M | 0x800000u) : (uint64_t)M; /* v = m * 2^(sh - 149) */
    const int sh = E ? (int)E - 1 : 0;
    const uint64_t t = m << (sh & 31); /* < 2^55 */
    const int64_t lo = (int64_t)(t & 0xffffffffu), hi = (int64_t)(t >> 32);
    if (u >> 31) {
      bin[sh >> 5] -= lo;
      bin[(sh >> 5) + 1] -= hi;
    } else {
      bin[sh >> 5] += lo;
      bin[(sh >> 5) + 1] += hi;
    }
  }
  uint32_t dig[6];
  int64_t carry = 0;
  for (int k = 0; k < 6; k++) {
    const int64_t t = bin[k] + carry;
    dig[k] = (uint32_t)((uint64_t)t & 0xffffffffu);
    carry = (t - (int64_t)dig[k]) / 4294967296LL; /* floor division: t - dig is an exact multiple of 2^32 */
  }
  acc[0] = (uint64_t)dig[0] | ((uint64_t)dig[1] << 32);
  acc[1] = (uint64_t)dig[2] | ((uint64_t)dig[3] << 32);
  acc[2] = (uint64_t)dig[4] | ((uint64_t)dig[5] << 32);
  acc[3] = special;
}
static int acc_bit(const uint64_t m[3], int i) { return i < 0 ? 0 : (int)((m[i >> 6] >> (i & 63)) & 1u); }
static float window_acc_round(const uint64_t acc[4], int dt) {
  if (acc[3] != 0) return NAN;
  uint64_t m[3] = {acc[0], acc[1], acc[2]};
  const int neg = (int)(m[2] >> 63);
  if (neg) { /* magnitude */
    m[0] = ~m[0]; m[1] = ~m[1]; m[2] = ~m[2];
    if (++m[0] == 0 && ++m[1] == 0) ++m[2];
  }
  int P = -1;
  for (int i = 191; i >= 0; i--)
    if (acc_bit(m, i)) { P = i; break; }
  if (P < 0) return 0.f;
  uint32_t bits;
  if (P < 24) {
    bits = (uint32_t)m[0]; /* the integer IS the fp32 encoding below 2^-125 */
  } else {
    uint32_t mant = 0;
    for (int i = P; i > P - 24; i--) mant = (mant << 1) | (uint32_t)acc_bit(m, i);
    const int guard = acc_bit(m, P - 24);
    int rest = 0;
    for (int i = P - 25; i >= 0 && !rest; i--) rest = acc_bit(m, i);
    int Pe = P;
    if (dt == CC_DT_F32) { /* nearest-even straight to fp32 */
      if (guard && (rest || (mant & 1u))) mant++;
      if (mant == (1u << 24)) { mant >>= 1; Pe++; }
    } else { /* round to odd at 24 bits; the 16-bit nearest-even that follows is then exact */
      mant |= (uint32_t)(guard || rest);
    }
    bits = ((uint32_t)(Pe - 22) << 23) | (mant & 0x7fffffu);
  }
  if (neg) bits |= 0x80000000u;
  float f;
  memcpy(&f, &bits, 4);
  return rnd(f, dt);
}
static float window_sum_row(const void* num, int dt, size_t off, int W) {
  uint64_t acc[4];
  window_acc_row(num, dt, off, W, acc);
  return window_acc_round(acc, dt);
}
/* the tracked state the device keeps (wsum float [H*S], acc uint64 [H*S*4 + 1]) recomputed from the ring */
static void window_state_from_ring(const void* num, int dt, int H, int S, int W, float* wsum, uint64_t* acc) {
  for (size_t i = 0; i < (size_t)H * S; i++) {
    uint64_t a[4];
    window_acc_row(num, dt, i * (size_t)W, W, a);
    if (wsum) wsum[i] = window_acc_round(a, dt);
    if (acc) memcpy(acc + i * 4, a, sizeof(a));
  }
  if (acc) { /* ticket + pad, then the column-major shadow of the ring */
    const size_t hs = (size_t)H * S, es = dt_size(dt);
    acc[hs * 4] = acc[hs * 4 + 1] = 0;
    char* shadow = (char*)(acc + hs * 4 + 2);
    for (size_t i = 0; i < hs; i++)
      for (int j = 0; j < W; j++) memcpy(shadow + ((size_t)j * hs + i) * es, (const char*)num + (i * (size_t)W + j) * es, es);
  }
}

size_t cc_hh_ring_acc_words_cpu(int32_t H, int32_t S, int32_t W, int32_t dtype) {
  const size_t hs = (size_t)H * (size_t)S, es = dtype == CC_DT_F32 ? 4 : 2;
  return hs * 4 + 2 + (hs * (size_t)W * es + 7) / 8 + 2 * (size_t)H; /* the device's layout: the last 2H words are scratch (zero) */
}

int cc_hh_ring_window_sums_cpu(const void* num, int32_t H, int32_t S, int32_t W, int32_t dtype, float* wsum, uint64_t* wsum_acc,
                               cc_stream_t stream) {
  (void)stream;
  if (!num || !wsum || H <= 0 || S <= 0 || W <= 0 || !dt_ok(dtype)) return CC_ERR_BAD_ARG;
  window_state_from_ring(num, dtype, H, S, W, wsum, wsum_acc);
  return CC_OK;
}

/* ref: KVCacheHybrid._decoding_update cache.py:965-1019, _select_fill_idx :896-950, _eviction_idx_for_head :844-894 */
int cc_hybrid_decode_update_cpu(const cc_kv_view* c, const void* k_new, const void* v_new, const int32_t* input_pos,
                                const int64_t* strategies, const int32_t* policy_table, int32_t n_policies, void* num,
                                int32_t* denom, int32_t W, const uint8_t* special_mask, uint8_t* punc_mask,
                                const uint8_t* is_punc_p, const int64_t* token_id, const int64_t* punc_ids, int32_t n_punc_ids,
                                const int32_t* num_special, int32_t* num_punc, int32_t g, int32_t requires_hh,
                                int64_t* fill_out, float* wsum_workspace, uint64_t* wsum_acc, cc_stream_t stream) {
  (void)stream;
  if (!view_ok(c) || !k_new || !v_new || !input_pos || !strategies || !policy_table || n_policies <= 0 || !num || !denom ||
      W <= 0 || !fill_out || c->Hp != c->H || c->Hc != c->H)
    return CC_ERR_BAD_ARG;
  const int S = c->S, dt = c->dtype;
  const int32_t p = *input_pos;
  int is_punc = is_punc_p ? (*is_punc_p != 0) : 0;
  if (!is_punc_p && token_id && punc_ids) /* ref: cache.py:975 torch.isin(input_ids, punc_ids) */
    for (int k = 0; k < n_punc_ids; k++) is_punc |= punc_ids[k] == *token_id;
  const size_t es = dt_size(dt);
  float* sc = (float*)malloc(sizeof(float) * (size_t)S);
  for (int h = 0; h < c->H; h++) {
    const int pol = (int)strategies[h];
    const int flags = policy_table[pol * 3], win = policy_table[pol * 3 + 1], hhs = policy_table[pol * 3 + 2];
    const int cts = c->cache_cts[h];
    const int end_idx = cts < S - 1 ? cts : S - 1;
    const size_t hoff = (size_t)h * S;
    int fill = -1, evict = 0;
    if ((flags & HF_PUNC) && is_punc) fill = end_idx;
    else if (flags & HF_FULL) fill = end_idx;
    else {
      int budget = g;
      if (flags & HF_SPECIAL) budget += num_special ? *num_special : 0;
      if (flags & HF_PUNC) budget += num_punc ? *num_punc : 0;
      if (flags & HF_WIN) budget += win;
      if (flags & HF_HH) budget += hhs;
      if (cts < budget) fill = end_idx;
      else if (flags & (HF_HH | HF_WIN)) {
        evict = 1;
        const int n = cts < S ? cts : S;
        for (int s = 0; s < n; s++) {
          const int32_t ps = c->pos[hoff + s];
          float v;
          if (flags & HF_HH) {
            int32_t dn = denom[hoff + s];
            dn = dn > W ? W : dn;
            v = window_sum_row(num, dt, (hoff + s) * (size_t)W, W) / (float)dn;
          } else v = (float)ps;
          int save = s < g;
          if ((flags & HF_SPECIAL) && special_mask) save |= special_mask[hoff + s] != 0;
          if ((flags & HF_PUNC) && punc_mask) save |= punc_mask[hoff + s] != 0;
          if (flags & HF_WIN) save |= ps > p - win;
          sc[s] = save ? INFINITY : v;
        }
        fill = (int)argmin_f32(sc, n);
      }
    }
    const int slot = fill < 0 ? S - 1 : fill;
    fill_out[h] = slot;
    if (evict && requires_hh) {
      for (int j = 0; j < W; j++) st(num, dt, (hoff + slot) * (size_t)W + j, 0.f);
      denom[hoff + slot] = 0;
      if (wsum_acc) { /* tracked state: the sum of a zeroed row (+ its shadow entries) */
        const size_t hs = (size_t)c->H * S;
        memset(wsum_acc + (hoff + slot) * 4, 0, 4 * sizeof(uint64_t));
        wsum_workspace[hoff + slot] = 0.f;
        for (int j = 0; j < W; j++) memset((char*)(wsum_acc + hs * 4 + 2) + ((size_t)j * hs + hoff + slot) * es, 0, es);
      }
    }
    if (!evict && fill >= 0) {
      c->cache_cts[h] = cts + 1;
      c->mask[hoff + slot] = 1;
    }
    c->pos[hoff + slot] = p;
    memcpy((char*)c->k_cache + (hoff + slot) * c->D * es, (const char*)k_new + (size_t)h * c->D * es, c->D * es);
    memcpy((char*)c->v_cache + (hoff + slot) * c->D * es, (const char*)v_new + (size_t)h * c->D * es, c->D * es);
    if (is_punc && punc_mask) punc_mask[hoff + slot] = 1;
  }
  if (is_punc && num_punc) *num_punc += 1;
  free(sc);
  return CC_OK;
}

/* ref: cache.py:716-723 with history_window_size > 1 */
int cc_hh_ring_update_cpu(void* num, int32_t* denom, int64_t* counter, const void* attn, int32_t H, int32_t S, int32_t T,
                          int32_t W, int32_t dtype, uint64_t* wsum_acc, float* wsum, cc_stream_t stream) {
  (void)stream;
  if (!num || !denom || !counter || !attn || H <= 0 || S <= 0 || T < 0 || T > S || W <= 0 || !dt_ok(dtype) ||
      ((wsum_acc == NULL) != (wsum == NULL)))
    return CC_ERR_BAD_ARG;
  const int slot = (int)(*counter % W);
  for (int h = 0; h < H; h++)
    for (int s = 0; s < S; s++) {
      st(num, dtype, ((size_t)h * S + s) * W + slot, s < T ? ld(attn, dtype, (size_t)h * T + s) : 0.f);
      denom[(size_t)h * S + s] += 1;
    }
  *counter += 1;
  if (wsum_acc) window_state_from_ring(num, dtype, H, S, W, wsum, wsum_acc); /* the oracle recomputes; the device tracks */
  return CC_OK;
}

/* ref: model.py:399-418 + cache.py:716-723 — attention, then the W > 1 history update from its group-mean output */
int cc_decode_attn_gqa_ring_cpu(const void* q, const void* k, const void* v, const uint8_t* mask, int32_t HQ, int32_t H, int32_t S,
                                int32_t D, int32_t dtype, float scale, void* y, void* attn_out, void* ring_num, int32_t* denom,
                                int64_t* counter, int32_t W, uint64_t* wsum_acc, float* wsum, void* workspace,
                                size_t workspace_bytes, cc_stream_t stream) {
  if (!ring_num || !denom || !counter || W <= 1 || !wsum_acc || !wsum) return CC_ERR_BAD_ARG;
  void* attn = attn_out ? attn_out : malloc((size_t)H * S * dt_size(dtype));
  int rc = cc_decode_attn_gqa_cpu(q, k, v, mask, HQ, H, S, D, dtype, scale, y, attn, NULL, NULL, NULL, NULL, workspace, workspace_bytes,
                                  stream);
  if (rc == CC_OK) rc = cc_hh_ring_update_cpu(ring_num, denom, counter, attn, H, S, S, W, dtype, wsum_acc, wsum, stream);
  if (!attn_out) free(attn);
  return rc;
}

/* Oracle twins of the hybrid two-launch step: the oracle keeps no pipeline state — the step IS the reference's sequence
 * (cache.py:965-1019 decision + insert, then attention, then the ring update cache.py:1283-1286), so the seed is a no-op
 * and next_key is ignored. */
int cc_hybrid_next_key_init_cpu(const cc_kv_view* c, const int32_t* input_pos, const int64_t* strategies, const int32_t* policy_table,
                                int32_t n_policies, const int32_t* denom, int32_t W, const float* wsum, const uint8_t* special_mask,
                                const uint8_t* punc_mask, int32_t global_tokens, uint64_t* next_key, cc_stream_t stream) {
  (void)input_pos; (void)denom; (void)W; (void)wsum; (void)special_mask; (void)punc_mask; (void)global_tokens; (void)next_key; (void)stream;
  if (!view_ok(c) || !strategies || !policy_table || n_policies <= 0) return CC_ERR_BAD_ARG;
  return CC_OK;
}

int cc_decode_step_hybrid_cpu(const cc_kv_view* c, const void* q, const void* k_new, const void* v_new, const int32_t* input_pos,
                              const int64_t* strategies, const int32_t* policy_table, int32_t n_policies, void* ring_num, int32_t* denom,
                              int64_t* counter, int32_t W, uint64_t* wsum_acc, float* wsum, const uint8_t* special_mask,
                              uint8_t* punc_mask, const int64_t* token_id, const int64_t* punc_ids, int32_t n_punc_ids,
                              const int32_t* num_special, int32_t* num_punc, uint64_t* next_key, int32_t global_tokens, int32_t HQ,
                              float scale, void* y, void* attn_out, void* workspace, size_t workspace_bytes, cc_stream_t stream) {
  (void)next_key;
  if (!view_ok(c) || !ring_num || !denom || !wsum) return CC_ERR_BAD_ARG; /* the oracle twin covers the ring-backed form */
  int64_t* fill = (int64_t*)malloc(sizeof(int64_t) * (size_t)c->H);
  int rc = cc_hybrid_decode_update_cpu(c, k_new, v_new, input_pos, strategies, policy_table, n_policies, ring_num, denom, W, special_mask,
                                       punc_mask, NULL, token_id, punc_ids, n_punc_ids, num_special, num_punc, global_tokens, 0, fill,
                                       wsum, wsum_acc, stream);
  free(fill);
  if (rc != CC_OK) return rc;
  return cc_decode_attn_gqa_ring_cpu(q, c->k_cache, c->v_cache, c->mask, HQ, c->H, c->S, c->D, c->dtype, scale, y, attn_out, ring_num, denom,
                                     counter, W, wsum_acc, wsum, workspace, workspace_bytes, stream);
}

/* cc_decode_step_hybrid_rc on the CPU: the step, then every head marked for this position (cf. cc_decode_step_heavy_hitter_rc_cpu;
 * a head already marked for it: refusal — the replay form is a device matter).  CC_RC_STRIDE is defined further down. */
int cc_decode_step_hybrid_rc_cpu(const cc_kv_view* c, const void* q, const void* k_new, const void* v_new, const int32_t* input_pos,
                                 const int64_t* strategies, const int32_t* policy_table, int32_t n_policies, void* ring_num,
                                 int32_t* denom, int64_t* counter, int32_t W, uint64_t* wsum_acc, float* wsum,
                                 const uint8_t* special_mask, uint8_t* punc_mask, const int64_t* token_id, const int64_t* punc_ids,
                                 int32_t n_punc_ids, const int32_t* num_special, int32_t* num_punc, uint64_t* next_key,
                                 int32_t* step_commit, int32_t global_tokens, int32_t HQ, float scale, void* y, void* attn_out,
                                 void* workspace, size_t workspace_bytes, cc_stream_t stream) {
  enum { RC_STRIDE = 68 };
  if (step_commit && input_pos && c)
    for (int h = 0; h < c->H; h++)
      if (step_commit[h * RC_STRIDE + 2] == *input_pos) return CC_ERR_UNSUPPORTED;
  const int rc = cc_decode_step_hybrid_cpu(c, q, k_new, v_new, input_pos, strategies, policy_table, n_policies, ring_num, denom, counter,
                                           W, wsum_acc, wsum, special_mask, punc_mask, token_id, punc_ids, n_punc_ids, num_special,
                                           num_punc, next_key, global_tokens, HQ, scale, y, attn_out, workspace, workspace_bytes, stream);
  if (rc == CC_OK && step_commit)
    for (int h = 0; h < c->H; h++) {
      step_commit[h * RC_STRIDE] = -1;
      step_commit[h * RC_STRIDE + 1] = step_commit[h * RC_STRIDE + 2] = *input_pos;
    }
  return rc;
}

int cc_attn_bandsum_cpu(const void* attn, int32_t H, int32_t Lq, int32_t Lk, int32_t dtype, int32_t band, float* out,
                        cc_stream_t stream) {
  (void)stream;
  if (!attn || !out || H <= 0 || Lq <= 0 || Lk <= 0 || band <= 0 || !dt_ok(dtype)) return CC_ERR_BAD_ARG;
  for (int h = 0; h < H; h++)
    for (int s = 0; s < Lk; s++) {
      float acc = 0.f;
      const int hi = s + band < Lq ? s + band : Lq;
      for (int i = s; i < hi; i++) acc = acc + ld(attn, dtype, ((size_t)h * Lq + i) * Lk + s);
      out[(size_t)h * Lk + s] = acc;
    }
  return CC_OK;
}

/* prefill attention with band sums (hybrid profiling): band_out[b,h,k] = sum_{q in [k, k+bands[b])} a[h,q,k] */
int cc_prefill_attn_bands_cpu(const void* q, const void* k, const void* v, int32_t HQ, int32_t H, int32_t L, int32_t D,
                              int32_t dtype, float scale, void* y, float* colsum_out, float* obs_out, int32_t obs_len,
                              const int32_t* bands, int32_t n_bands, float* band_out, void* workspace,
                              size_t workspace_bytes, cc_stream_t stream) {
  (void)stream;
  (void)workspace;
  (void)workspace_bytes;
  if (!q || !k || !v || !y || HQ <= 0 || H <= 0 || HQ % H || L <= 0 || D <= 0 || !dt_ok(dtype))
    return CC_ERR_BAD_ARG;
  if (n_bands > 0 && (!bands || !band_out || n_bands > 4)) return CC_ERR_BAD_ARG;
  return prefill_core(q, k, v, HQ, H, L, D, dtype, scale, y, colsum_out, obs_out, obs_len, bands, n_bands > 0 ? n_bands : 0,
                      n_bands > 0 ? band_out : NULL, NULL);
}

/* Oracle-only (no device twin): the materialised [H, L, L] group-averaged attention the reference's policies consume
 * (attention_utils.py:36-54 + model.py:413-418), as float32 values already rounded to the model dtype — what the
 * full-size hybrid profiling test feeds to its restatement of cache.py:1066-1187. */
int cc_prefill_attn_matrix_cpu(const void* q, const void* k, const void* v, int32_t HQ, int32_t H, int32_t L, int32_t D,
                               int32_t dtype, float scale, void* y, float* attn_full) {
  if (!q || !k || !v || !y || !attn_full || HQ <= 0 || H <= 0 || HQ % H || L <= 0 || D <= 0 || !dt_ok(dtype)) return CC_ERR_BAD_ARG;
  return prefill_core(q, k, v, HQ, H, L, D, dtype, scale, y, NULL, NULL, 0, NULL, 0, NULL, attn_full);
}

/* ref: KVCacheHeavyHitter._eviction_idx cache.py:725-765 with history_window_size W > 1 */
int cc_decode_update_heavy_hitter_ring_cpu(const cc_kv_view* c, const void* k_new, const void* v_new,
                                           const int32_t* input_pos, void* num, int32_t* denom, int32_t W, int32_t g,
                                           int32_t w, int64_t* idx_out, float* wsum_workspace, uint64_t* wsum_acc,
                                           cc_stream_t stream) {
  (void)stream;
  if (!view_ok(c) || !input_pos || !idx_out || !num || !denom || W <= 0 || c->Hp != c->H || c->H > 4096) return CC_ERR_BAD_ARG;
  const int32_t p = *input_pos;
  const int dt = c->dtype;
  float* sc = (float*)malloc(sizeof(float) * (size_t)c->S);
  for (int h = 0; h < c->H; h++) {
    const int32_t* pos = c->pos + (size_t)h * c->S;
    for (int s = 0; s < c->S; s++) {
      const size_t i = (size_t)h * c->S + s;
      int32_t dn = denom[i] < 1 ? 1 : (denom[i] > W ? W : denom[i]);
      float v = window_sum_row(num, dt, i * (size_t)W, W) / (float)dn;
      if (pos[s] < g || pos[s] >= p - w) v = 1.0f;
      if (pos[s] == -1) v = 0.0f;
      sc[s] = v;
    }
    idx_out[h] = argmin_f32(sc, c->S);
  }
  free(sc);
  for (int h = 0; h < c->H; h++) {
    const size_t i = (size_t)h * c->S + idx_out[h];
    for (int j = 0; j < W; j++) st(num, dt, i * (size_t)W + j, 0.f);
    denom[i] = 0;
    if (wsum_acc) { /* tracked state: the sum of a zeroed row (+ its shadow entries) */
      const size_t hs = (size_t)c->H * c->S, es = dt_size(dt);
      memset(wsum_acc + i * 4, 0, 4 * sizeof(uint64_t));
      wsum_workspace[i] = 0.f;
      for (int j = 0; j < W; j++) memset((char*)(wsum_acc + hs * 4 + 2) + ((size_t)j * hs + i) * es, 0, es);
    }
  }
  if (k_new) insert_token(c, k_new, v_new, p, idx_out);
  return CC_OK;
}

/* ---------------------------------------------------------------- fused decode step (pipeline form) ----- */

/* arg-min key of the heavy-hitter eviction at position p: (orderable(score) << 32) | slot << 1 | was_empty */
static uint32_t orderable_f32_host(float f) {
  uint32_t u;
  if (f != f) return 0u;
  if (f == 0.0f) f = 0.0f;
  memcpy(&u, &f, 4);
  u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
  return u == 0u ? 1u : u;
}
static uint64_t hh_key_for_head(const cc_kv_view* c, int h, int32_t p, const double* num, const int32_t* denom, int g, int w) {
  uint64_t best = ~(uint64_t)0;
  for (int s = 0; s < c->S; s++) {
    const size_t i = (size_t)h * c->S + s;
    const int32_t ps = c->pos[i];
    int32_t dn = denom[i] < 1 ? 1 : denom[i];
    float v = (float)num[i] / (float)dn;
    if (ps < g || ps >= p - w) v = 1.0f;
    if (ps == -1) v = 0.0f;
    const uint64_t key = ((uint64_t)orderable_f32_host(v) << 32) | ((uint64_t)(uint32_t)s << 1) | (uint64_t)(ps == -1);
    if (key < best) best = key;
  }
  return best;
}

/* next_key is [H][NK] partial minima (the device publishes one per 128-slot chunk); only their minimum is
 * contractual, so the oracle keeps the whole key in entry 0. */
int32_t cc_hh_next_key_slots_cpu(int32_t S) { return S > 0 ? (S + 127) / 128 : 0; }

int cc_hh_next_key_init_cpu(const cc_kv_view* c, const int32_t* input_pos, const double* num, const int32_t* denom, int32_t g,
                            int32_t w, uint64_t* next_key, cc_stream_t stream) {
  (void)stream;
  if (!view_ok(c) || !input_pos || !num || !denom || !next_key || c->Hp != c->H) return CC_ERR_BAD_ARG;
  const int32_t p = *input_pos;
  const int nk = cc_hh_next_key_slots_cpu(c->S);
  for (int h = 0; h < c->H; h++) {
    for (int i = 1; i < nk; i++) next_key[(size_t)h * nk + i] = ~(uint64_t)0;
    next_key[(size_t)h * nk] = hh_key_for_head(c, h, p, num, denom, g, w);
  }
  return CC_OK;
}

/* One whole heavy-hitter decode step in pipeline form: consume the slot chosen for this position, insert, attend
 * with the fused history update, and leave the NEXT position's arg-min key (ref: the same lines as
 * cc_decode_update_heavy_hitter + cc_decode_attn_gqa; the only difference is WHEN the arg-min is evaluated). */
int cc_decode_step_heavy_hitter_cpu(const cc_kv_view* c, const void* q, const void* k_new, const void* v_new,
                                    const int32_t* input_pos, double* num, int32_t* denom, int64_t* counter,
                                    uint64_t* next_key, int32_t g, int32_t w, int32_t HQ, float scale, void* y,
                                    void* attn_out, void* workspace, size_t workspace_bytes, cc_stream_t stream) {
  if (!view_ok(c) || !q || !k_new || !v_new || !input_pos || !num || !denom || !next_key || !y || c->Hp != c->H || c->H > 4096)
    return CC_ERR_BAD_ARG;
  const int32_t p = *input_pos;
  const int nk = cc_hh_next_key_slots_cpu(c->S);
  int64_t idx[4096];
  for (int h = 0; h < c->H; h++) {
    uint64_t key = ~(uint64_t)0;
    for (int i = 0; i < nk; i++)
      if (next_key[(size_t)h * nk + i] < key) key = next_key[(size_t)h * nk + i];
    if (key == ~(uint64_t)0) return CC_ERR_BAD_ARG;
    idx[h] = (int64_t)((key & 0xffffffffu) >> 1);
    num[(size_t)h * c->S + idx[h]] = 0.0;
    denom[(size_t)h * c->S + idx[h]] = 0;
  }
  insert_token(c, k_new, v_new, p, idx);
  int rc = cc_decode_attn_gqa_cpu(q, c->k_cache, c->v_cache, c->mask, HQ, c->H, c->S, c->D, c->dtype, scale, y, attn_out, NULL,
                                  num, denom, counter, workspace, workspace_bytes, stream);
  if (rc != CC_OK) return rc;
  for (int h = 0; h < c->H; h++) {
    for (int i = 1; i < nk; i++) next_key[(size_t)h * nk + i] = ~(uint64_t)0;
    next_key[(size_t)h * nk] = hh_key_for_head(c, h, p + 1, num, denom, g, w);
  }
  return CC_OK;
}

/* KVCacheHeavyHitter with history_window_size W > 1 in the same pipeline (ref: cache.py:725-765, 716-723): the key of
 * the slot head h evicts at position p, from the ring itself (exact window sums). */
static uint64_t hh_ring_key_for_head(const cc_kv_view* c, int h, const void* num, const int32_t* denom, int W, int32_t p, int g, int w) {
  uint64_t best = ~(uint64_t)0;
  for (int s = 0; s < c->S; s++) {
    const size_t i = (size_t)h * c->S + s;
    const int32_t ps = c->pos[i];
    int32_t dn = denom[i] < 1 ? 1 : (denom[i] > W ? W : denom[i]);
    float v = window_sum_row(num, c->dtype, i * (size_t)W, W) / (float)dn;
    if (ps < g || ps >= p - w) v = 1.0f;
    if (ps == -1) v = 0.0f;
    const uint64_t key = ((uint64_t)orderable_f32_host(v) << 32) | ((uint64_t)(uint32_t)s << 1) | (uint64_t)(ps == -1);
    if (key < best) best = key;
  }
  return best;
}

int cc_decode_step_heavy_hitter_ring_cpu(const cc_kv_view* c, const void* q, const void* k_new, const void* v_new,
                                         const int32_t* input_pos, void* ring_num, int32_t* denom, int64_t* counter, int32_t W,
                                         uint64_t* wsum_acc, float* wsum, uint64_t* next_key, int32_t g, int32_t w, int32_t HQ,
                                         float scale, void* y, void* attn_out, void* workspace, size_t workspace_bytes,
                                         cc_stream_t stream) {
  if (!view_ok(c) || !q || !k_new || !v_new || !input_pos || !ring_num || !denom || !counter || W <= 1 || !wsum_acc || !wsum ||
      !next_key || !y || c->Hp != c->H || c->H > 4096)
    return CC_ERR_BAD_ARG;
  const int nk = cc_hh_next_key_slots_cpu(c->S);
  const size_t hs = (size_t)c->H * c->S, es = dt_size(c->dtype);
  int64_t idx[4096];
  for (int h = 0; h < c->H; h++) {
    uint64_t key = ~(uint64_t)0;
    for (int i = 0; i < nk; i++)
      if (next_key[(size_t)h * nk + i] < key) key = next_key[(size_t)h * nk + i];
    if (key == ~(uint64_t)0) return CC_ERR_BAD_ARG;
    idx[h] = (int64_t)((key & 0xffffffffu) >> 1);
    const size_t i = (size_t)h * c->S + idx[h]; /* :754-763 the evicted slot's history restarts from zero */
    for (int j = 0; j < W; j++) st(ring_num, c->dtype, i * (size_t)W + j, 0.f);
    denom[i] = 0;
    memset(wsum_acc + i * 4, 0, 4 * sizeof(uint64_t));
    wsum[i] = 0.f;
    for (int j = 0; j < W; j++) memset((char*)(wsum_acc + hs * 4 + 2) + ((size_t)j * hs + i) * es, 0, es);
  }
  insert_token(c, k_new, v_new, *input_pos, idx);
  int rc = cc_decode_attn_gqa_ring_cpu(q, c->k_cache, c->v_cache, c->mask, HQ, c->H, c->S, c->D, c->dtype, scale, y, attn_out, ring_num,
                                       denom, counter, W, wsum_acc, wsum, workspace, workspace_bytes, stream);
  if (rc != CC_OK) return rc;
  for (int h = 0; h < c->H; h++) {
    for (int i = 1; i < nk; i++) next_key[(size_t)h * nk + i] = ~(uint64_t)0;
    next_key[(size_t)h * nk] = hh_ring_key_for_head(c, h, ring_num, denom, W, *input_pos + 1, g, w);
  }
  return CC_OK;
}

int cc_hh_ring_next_key_init_cpu(const cc_kv_view* c, const int32_t* input_pos, const int32_t* denom, int32_t W, const float* wsum,
                                 int32_t g, int32_t w, uint64_t* next_key, cc_stream_t stream) {
  (void)stream;
  if (!view_ok(c) || !input_pos || !denom || W <= 1 || !wsum || !next_key || c->Hp != c->H) return CC_ERR_BAD_ARG;
  const int nk = cc_hh_next_key_slots_cpu(c->S);
  for (int h = 0; h < c->H; h++) {
    uint64_t best = ~(uint64_t)0;
    for (int s = 0; s < c->S; s++) { /* from the window sums handed in (the caller's tracked state) */
      const size_t i = (size_t)h * c->S + s;
      const int32_t ps = c->pos[i];
      int32_t dn = denom[i] < 1 ? 1 : (denom[i] > W ? W : denom[i]);
      float v = wsum[i] / (float)dn;
      if (ps < g || ps >= *input_pos - w) v = 1.0f;
      if (ps == -1) v = 0.0f;
      const uint64_t key = ((uint64_t)orderable_f32_host(v) << 32) | ((uint64_t)(uint32_t)s << 1) | (uint64_t)(ps == -1);
      if (key < best) best = key;
    }
    for (int i = 1; i < nk; i++) next_key[(size_t)h * nk + i] = ~(uint64_t)0;
    next_key[(size_t)h * nk] = best;
  }
  return CC_OK;
}

/* KVCacheL2 in the same pipeline (ref: cache.py:597-605 + :373-376): the key of the slot head h evicts at position p */
static float l2_global_max(const cc_kv_view* c, const void* key_norm) {
  float mx = -INFINITY;
  int has_nan = 0;
  for (size_t i = 0; i < (size_t)c->H * c->S; i++) {
    const float v = ld(key_norm, c->dtype, i);
    if (v != v) has_nan = 1;
    if (v > mx) mx = v;
  }
  return has_nan ? NAN : mx; /* torch.max propagates NaN */
}
static uint64_t l2_key_for_head(const cc_kv_view* c, int h, const void* key_norm, float mx, int32_t p, int32_t g, int32_t w) {
  uint64_t best = ~(uint64_t)0;
  for (int s = 0; s < c->S; s++) {
    const int32_t ps = c->pos[(size_t)h * c->S + s];
    float v = rnd(mx - ld(key_norm, c->dtype, (size_t)h * c->S + s), c->dtype);
    if (ps >= p - w) v = INFINITY;
    if (s < g) v = INFINITY;
    if (ps == -1) v = -INFINITY;
    const uint64_t key = ((uint64_t)orderable_f32_host(v) << 32) | ((uint64_t)(uint32_t)s << 1) | (uint64_t)(ps == -1);
    if (key < best) best = key;
  }
  return best;
}

int cc_l2_next_key_init_cpu(const cc_kv_view* c, const int32_t* input_pos, void* key_norm, int32_t g, int32_t w, uint64_t* next_key,
                            cc_stream_t stream) {
  (void)stream;
  if (!view_ok(c) || !input_pos || !key_norm || !next_key || c->Hp != c->H || g < 0) return CC_ERR_BAD_ARG;
  const int nk = cc_hh_next_key_slots_cpu(c->S);
  const float mx = l2_global_max(c, key_norm);
  for (int h = 0; h < c->H; h++) {
    for (int i = 1; i < nk; i++) next_key[(size_t)h * nk + i] = ~(uint64_t)0;
    next_key[(size_t)h * nk] = l2_key_for_head(c, h, key_norm, mx, *input_pos, g, w);
  }
  return CC_OK;
}

int cc_decode_step_l2_cpu(const cc_kv_view* c, const void* q, const void* k_new, const void* v_new, const int32_t* input_pos,
                          void* key_norm, uint64_t* next_key, int32_t g, int32_t w, int32_t HQ, float scale, void* y, void* workspace,
                          size_t workspace_bytes, cc_stream_t stream) {
  if (!view_ok(c) || !q || !k_new || !v_new || !input_pos || !key_norm || !next_key || !y || c->Hp != c->H || g < 0 || c->H > 4096)
    return CC_ERR_BAD_ARG;
  if (dt_size(c->dtype) != 2 || c->D != 128) return CC_ERR_UNSUPPORTED;
  const int nk = cc_hh_next_key_slots_cpu(c->S);
  int64_t idx[4096];
  for (int h = 0; h < c->H; h++) {
    uint64_t key = ~(uint64_t)0;
    for (int i = 0; i < nk; i++)
      if (next_key[(size_t)h * nk + i] < key) key = next_key[(size_t)h * nk + i];
    if (key == ~(uint64_t)0) return CC_ERR_BAD_ARG;
    idx[h] = (int64_t)((key & 0xffffffffu) >> 1);
  }
  insert_token(c, k_new, v_new, *input_pos, idx);
  for (int h = 0; h < c->H; h++) /* :592-593 */
    st(key_norm, c->dtype, (size_t)h * c->S + idx[h], sqrtf(sumsq_canonical(k_new, c->dtype, (size_t)h * c->D, c->D)));
  int rc = cc_decode_attn_gqa_cpu(q, c->k_cache, c->v_cache, c->mask, HQ, c->H, c->S, c->D, c->dtype, scale, y, NULL, NULL, NULL,
                                  NULL, NULL, workspace, workspace_bytes, stream);
  if (rc != CC_OK) return rc;
  const float mx = l2_global_max(c, key_norm);
  for (int h = 0; h < c->H; h++) {
    for (int i = 1; i < nk; i++) next_key[(size_t)h * nk + i] = ~(uint64_t)0;
    next_key[(size_t)h * nk] = l2_key_for_head(c, h, key_norm, mx, *input_pos + 1, g, w);
  }
  return CC_OK;
}

/* KVCacheRandom in the same pipeline (ref: cache.py:505-524 + :373-376): the key of the slot the reference's arg-min picks
 * for position p, from the uniform draw for that position.  (The device keeps one identical key row PER KV HEAD for the
 * head-constant policies — each head's workgroups read and rewrite their own copy, include/coldcompress.h; the oracle has no
 * workgroups to keep apart and uses row 0 of whatever array it is handed.) */
static uint64_t random_key(const cc_kv_view* c, const float* rand_u, int32_t p, int32_t g, int32_t w) {
  uint64_t best = ~(uint64_t)0;
  for (int s = 0; s < c->S; s++) {
    const int32_t ps = c->pos[s];
    float v = (ps >= p - w) ? INFINITY : rand_u[s]; /* :523 */
    if (s < g) v = INFINITY;                        /* :374 */
    if (ps == -1) v = -INFINITY;                    /* :376 */
    const uint64_t key = ((uint64_t)orderable_f32_host(v) << 32) | ((uint64_t)(uint32_t)s << 1) | (uint64_t)(ps == -1);
    if (key < best) best = key;
  }
  return best;
}

int cc_random_next_key_init_cpu(const cc_kv_view* c, const int32_t* input_pos, const float* rand_u, int32_t g, int32_t w,
                                uint64_t* next_key, cc_stream_t stream) {
  (void)stream;
  if (!view_ok(c) || !input_pos || !rand_u || !next_key || c->Hp != 1 || g < 0) return CC_ERR_BAD_ARG;
  const int nk = cc_hh_next_key_slots_cpu(c->S);
  for (int i = 1; i < nk; i++) next_key[i] = ~(uint64_t)0;
  next_key[0] = random_key(c, rand_u, *input_pos, g, w);
  return CC_OK;
}

int cc_decode_step_random_cpu(const cc_kv_view* c, const void* q, const void* k_new, const void* v_new, const int32_t* input_pos,
                              const float* rand_next, uint64_t* next_key, int32_t g, int32_t w, int32_t HQ, float scale, void* y,
                              void* workspace, size_t workspace_bytes, cc_stream_t stream) {
  if (!view_ok(c) || !q || !k_new || !v_new || !input_pos || !rand_next || !next_key || !y || c->Hp != 1 || g < 0)
    return CC_ERR_BAD_ARG;
  const int nk = cc_hh_next_key_slots_cpu(c->S);
  uint64_t key = ~(uint64_t)0;
  for (int i = 0; i < nk; i++)
    if (next_key[i] < key) key = next_key[i];
  if (key == ~(uint64_t)0) return CC_ERR_BAD_ARG;
  int64_t idx = (int64_t)((key & 0xffffffffu) >> 1);
  insert_token(c, k_new, v_new, *input_pos, &idx);
  int rc = cc_decode_attn_gqa_cpu(q, c->k_cache, c->v_cache, c->mask, HQ, c->H, c->S, c->D, c->dtype, scale, y, NULL, NULL, NULL,
                                  NULL, NULL, workspace, workspace_bytes, stream);
  if (rc != CC_OK) return rc;
  for (int i = 1; i < nk; i++) next_key[i] = ~(uint64_t)0;
  next_key[0] = random_key(c, rand_next, *input_pos + 1, g, w);
  return CC_OK;
}

/* The in-kernel generator's twins (include/coldcompress.h, cc_decode_step_random_rng): the draw for slot s at position p is
 * the murmur3 64-bit finaliser applied twice to seed + p * golden + s, top 24 bits * 2^-24.  The reference draws torch.rand(S)
 * (cache.py:521), a backend stream no other implementation reproduces; parity with the reference is pinned with the vector
 * injected (cc_decode_step_random_cpu above), and these twins pin the device's own generator. */
static uint64_t rng_mix64(uint64_t x) {
  x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33;
  return x;
}
float cc_rng_uniform_cpu(uint64_t seed, int32_t pos, int32_t slot) {
  const uint64_t x = rng_mix64(rng_mix64(seed + (uint64_t)(uint32_t)pos * 0x9E3779B97F4A7C15ull + (uint64_t)(uint32_t)slot));
  return (float)(uint32_t)(x >> 40) * 5.9604644775390625e-08f;
}
static float* rng_vector(uint64_t seed, int32_t p, int S) {
  float* u = (float*)malloc((size_t)S * sizeof(float));
  for (int s = 0; u && s < S; s++) u[s] = cc_rng_uniform_cpu(seed, p, s);
  return u;
}

int cc_random_next_key_init_rng_cpu(const cc_kv_view* c, const int32_t* input_pos, uint64_t seed, int32_t g, int32_t w,
                                    uint64_t* next_key, cc_stream_t stream) {
  if (!view_ok(c) || !input_pos) return CC_ERR_BAD_ARG;
  float* u = rng_vector(seed, *input_pos, c->S);
  if (!u) return CC_ERR_BAD_ARG;
  const int rc = cc_random_next_key_init_cpu(c, input_pos, u, g, w, next_key, stream);
  free(u);
  return rc;
}

int cc_decode_step_random_rng_cpu(const cc_kv_view* c, const void* q, const void* k_new, const void* v_new,
                                  const int32_t* input_pos, uint64_t seed, uint64_t* next_key, int32_t g, int32_t w, int32_t HQ,
                                  float scale, void* y, void* workspace, size_t workspace_bytes, cc_stream_t stream) {
  if (!view_ok(c) || !input_pos) return CC_ERR_BAD_ARG;
  float* u = rng_vector(seed, *input_pos + 1, c->S);
  if (!u) return CC_ERR_BAD_ARG;
  const int rc = cc_decode_step_random_cpu(c, q, k_new, v_new, input_pos, u, next_key, g, w, HQ, scale, y, workspace,
                                           workspace_bytes, stream);
  free(u);
  return rc;
}

/* Measurement hook twin: the oracle has no launches to select; only phases == 3 (the whole step) is meaningful. */
int cc_decode_step_heavy_hitter_phases_cpu(const cc_kv_view* c, const void* q, const void* k_new, const void* v_new,
                                           const int32_t* input_pos, double* num, int32_t* denom, int64_t* counter,
                                           uint64_t* next_key, int32_t g, int32_t w, int32_t HQ, float scale, void* y,
                                           void* attn_out, void* workspace, size_t workspace_bytes, cc_stream_t stream,
                                           int32_t phases) {
  if ((phases & 3) != 3) return CC_ERR_UNSUPPORTED;
  return cc_decode_step_heavy_hitter_cpu(c, q, k_new, v_new, input_pos, num, denom, counter, next_key, g, w, HQ, scale, y,
                                         attn_out, workspace, workspace_bytes, stream);
}

/* The device entry point with the recoverable hand-off's commit words (r4 layout: CC_RC_STRIDE int32 per kv head — [0] the insert
 * word, [1] its position, [2 + split] the position workgroup `split` committed, [66], [67] the hybrid step's count and ring column): on the CPU nothing can time out — the step runs,
 * then every head is marked for this position in word [1] and in word [2] (the one workgroup a CPU has; word [0], the insert slot,
 * is the device's business and is set to -1 here); a head ALREADY marked for this position makes the whole call a refusal (the
 * replay form is a device matter). */
#define CC_RC_STRIDE 68
int cc_decode_step_heavy_hitter_rc_cpu(const cc_kv_view* c, const void* q, const void* k_new, const void* v_new,
                                       const int32_t* input_pos, double* num, int32_t* denom, int64_t* counter,
                                       uint64_t* next_key, int32_t* step_commit, int32_t g, int32_t w, int32_t HQ, float scale,
                                       void* y, void* workspace, size_t workspace_bytes, cc_stream_t stream, int32_t phases) {
  if ((phases & 3) != 3) return CC_ERR_UNSUPPORTED;
  if (step_commit && input_pos && c)
    for (int h = 0; h < c->H; h++)
      if (step_commit[h * CC_RC_STRIDE + 2] == *input_pos) return CC_ERR_UNSUPPORTED;
  const int rc = cc_decode_step_heavy_hitter_cpu(c, q, k_new, v_new, input_pos, num, denom, counter, next_key, g, w, HQ, scale, y,
                                                 NULL, workspace, workspace_bytes, stream);
  if (rc == CC_OK && step_commit)
    for (int h = 0; h < c->H; h++) {
      step_commit[h * CC_RC_STRIDE] = -1;
      step_commit[h * CC_RC_STRIDE + 1] = step_commit[h * CC_RC_STRIDE + 2] = *input_pos;
    }
  return rc;
}

/* ------------------------------------------------------------------------------------------------
 * Quantised KV cache (--cache_bits {8,4,2}).  ref: quantization_utils.py:4-98 with axis = 2 (cache.py:183):
 * ONE (scale, zero point) per cache slot s, shared by all heads and channels; every op is a torch elementwise op
 * on tensors of the cache dtype, i.e. computed in fp32 and rounded to the dtype after EACH op (rnd()):
 *   min, max over x[:, s, :];  scale = rnd(max(rnd(max - min), T(1e-6)) / max_int);
 *   zero = rnd(min + rnd(scale * 2^(n-1)));  q = clamp(roundeven(rnd(rnd(x - min) / scale)), 0, max_int);
 *   dequant = rnd(rnd((q - 2^(n-1)) * scale) + zero).
 * Storage: n = 8 -> int8 [H, S, D] holding (uint8)q (the reference's .to(int8) wraps);  n = 4 / 2 -> 8/n consecutive
 * values of the flattened [H, S, D] tensor per byte, value j of the group shifted left by j*n bits.
 * cc_kv_requant = quantize_cache() followed by dequantize_cache() (cache.py:283-309): `work` is replaced by its
 * quantise -> dequantise round trip and the quantised image is emitted.
 * ---------------------------------------------------------------------------------------------- */
static int quant_args_ok(int32_t H, int32_t S, int32_t D, int32_t dt, int32_t n_bit) {
  return H > 0 && S > 0 && D > 0 && dt_ok(dt) && (n_bit == 8 || n_bit == 4 || n_bit == 2) && (D % (8 / n_bit)) == 0;
}

static void q_store(uint8_t* q, size_t i, int n_bit, int v) {
  if (n_bit == 8) { q[i] = (uint8_t)v; return; }
  const int per = 8 / n_bit;
  const size_t byte = i / per;
  const int sh = (int)(i % per) * n_bit;
  q[byte] = (uint8_t)((q[byte] & ~(((1u << n_bit) - 1u) << sh)) | ((unsigned)v << sh));
}

static int q_load(const uint8_t* q, size_t i, int n_bit) {
  if (n_bit == 8) return q[i];
  const int per = 8 / n_bit;
  return (q[i / per] >> ((int)(i % per) * n_bit)) & ((1 << n_bit) - 1);
}

int cc_kv_requant_cpu(void* work, void* q_out, void* scales, void* zeros, int32_t H, int32_t S, int32_t D, int32_t dt,
                      int32_t n_bit, cc_stream_t stream) {
  (void)stream;
  if (!work || !q_out || !scales || !zeros || !quant_args_ok(H, S, D, dt, n_bit)) return CC_ERR_BAD_ARG;
  const int max_int = (1 << n_bit) - 1, half = 1 << (n_bit - 1);
  for (int s = 0; s < S; s++) {
    float mn = INFINITY, mx = -INFINITY;
    for (int h = 0; h < H; h++)
      for (int d = 0; d < D; d++) {
        const float x = ld(work, dt, ((size_t)h * S + s) * D + d);
        if (x < mn) mn = x;
        if (x > mx) mx = x;
      }
    float range = rnd(mx - mn, dt);
    const float floor_t = rnd(1e-6f, dt);
    if (range < floor_t) range = floor_t;
    const float scale = rnd(range / (float)max_int, dt);
    const float zero = rnd(mn + rnd(scale * (float)half, dt), dt);
    st(scales, dt, (size_t)s, scale);
    st(zeros, dt, (size_t)s, zero);
    for (int h = 0; h < H; h++)
      for (int d = 0; d < D; d++) {
        const size_t i = ((size_t)h * S + s) * D + d;
        float t = rnd(rnd(ld(work, dt, i) - mn, dt) / scale, dt);
        t = nearbyintf(t);  /* round half to even (default rounding mode), as torch.round */
        if (t < 0.f) t = 0.f;
        if (t > (float)max_int) t = (float)max_int;
        const int q = (int)t;
        q_store((uint8_t*)q_out, i, n_bit, q);
        st(work, dt, i, rnd(rnd((float)(q - half) * scale, dt) + zero, dt));
      }
  }
  return CC_OK;
}

int cc_kv_requant_pair_cpu(void* k_work, void* k_q, void* k_scales, void* k_zeros, void* v_work, void* v_q, void* v_scales,
                           void* v_zeros, int32_t H, int32_t S, int32_t D, int32_t dt, int32_t n_bit, const int32_t* pos, int32_t Hp,
                           uint8_t* stable, int32_t* pos_seen, cc_stream_t stream) {
  /* the oracle always runs the whole round trip: the device's skipping of stable slots must be invisible */
  (void)pos; (void)Hp; (void)stable; (void)pos_seen;
  int rc = cc_kv_requant_cpu(k_work, k_q, k_scales, k_zeros, H, S, D, dt, n_bit, stream);
  if (rc != CC_OK) return rc;
  return cc_kv_requant_cpu(v_work, v_q, v_scales, v_zeros, H, S, D, dt, n_bit, stream);
}

int cc_kv_dequant_cpu(const void* q, const void* scales, const void* zeros, void* work_out, int32_t H, int32_t S, int32_t D,
                      int32_t dt, int32_t n_bit, cc_stream_t stream) {
  (void)stream;
  if (!q || !scales || !zeros || !work_out || !quant_args_ok(H, S, D, dt, n_bit)) return CC_ERR_BAD_ARG;
  const int half = 1 << (n_bit - 1);
  for (int h = 0; h < H; h++)
    for (int s = 0; s < S; s++) {
      const float scale = ld(scales, dt, (size_t)s), zero = ld(zeros, dt, (size_t)s);
      for (int d = 0; d < D; d++) {
        const size_t i = ((size_t)h * S + s) * D + d;
        st(work_out, dt, i, rnd(rnd((float)(q_load((const uint8_t*)q, i, n_bit) - half) * scale, dt) + zero, dt));
      }
    }
  return CC_OK;
}

/* ------------------------------------------------------------------------------------------------
 * FUSED quantised cache (opt-in; the build's own contract, NOT the reference's — include/coldcompress.h): one
 * (scale, minimum) pair per (head, slot) row, fp32; range = max(mx - mn, 1e-6f), scale = range / 255,
 * q = clamp(rint((x - mn) * (255 / range)), 0, 255); value = T(fmaf(q, scale, mn)).
 * The step twin dequantises the whole cache, runs the policy's fused step on it with the new token replaced by the
 * round trip of its own image, and records the image — the arithmetic the device performs in registers.
 * ---------------------------------------------------------------------------------------------- */
int cc_decode_step_recent_global_cpu(const cc_kv_view* c, const void* q, const void* k_new, const void* v_new,
                                     const int32_t* input_pos, uint64_t* next_key, int32_t g, int32_t HQ, float scale, void* y,
                                     void* workspace, size_t workspace_bytes, cc_stream_t stream);

static void quant_row8(const void* src, int dt, size_t off, int D, uint8_t* dst, float* sc_out, float* mn_out) {
  float mn = INFINITY, mx = -INFINITY;
  for (int d = 0; d < D; d++) {
    const float x = ld(src, dt, off + d);
    if (x < mn) mn = x;
    if (x > mx) mx = x;
  }
  float range = mx - mn;
  if (!(range > 1e-6f)) range = 1e-6f;
  const float sc = range / 255.f, inv = 255.f / range;
  for (int d = 0; d < D; d++) {
    float t = nearbyintf((ld(src, dt, off + d) - mn) * inv);
    if (t < 0.f) t = 0.f;
    if (t > 255.f) t = 255.f;
    dst[d] = (uint8_t)t;
  }
  *sc_out = sc;
  *mn_out = mn;
}

int cc_kv_quant_rows_cpu(const void* k, const void* v, int32_t H, int32_t S, int32_t D, int32_t dt, int32_t n_bit, uint8_t* k_q,
                         uint8_t* v_q, float* qparams, cc_stream_t stream) {
  (void)stream;
  if (!k || !v || !k_q || !v_q || !qparams || H <= 0 || S <= 0 || D <= 0 || !dt_ok(dt)) return CC_ERR_BAD_ARG;
  if (n_bit != 8) return CC_ERR_UNSUPPORTED;
  for (size_t r = 0; r < (size_t)H * S; r++) {
    quant_row8(k, dt, r * D, D, k_q + r * D, &qparams[r * 4], &qparams[r * 4 + 1]);
    quant_row8(v, dt, r * D, D, v_q + r * D, &qparams[r * 4 + 2], &qparams[r * 4 + 3]);
  }
  return CC_OK;
}

int cc_kv_dequant_rows_cpu(const uint8_t* k_q, const uint8_t* v_q, const float* qparams, int32_t H, int32_t S, int32_t D, int32_t dt,
                           int32_t n_bit, void* k_out, void* v_out, cc_stream_t stream) {
  (void)stream;
  if (!k_q || !v_q || !qparams || !k_out || !v_out || H <= 0 || S <= 0 || D <= 0 || !dt_ok(dt)) return CC_ERR_BAD_ARG;
  if (n_bit != 8) return CC_ERR_UNSUPPORTED;
  for (size_t r = 0; r < (size_t)H * S; r++)
    for (int d = 0; d < D; d++) {
      st(k_out, dt, r * D + d, fmaf((float)k_q[r * D + d], qparams[r * 4], qparams[r * 4 + 1]));
      st(v_out, dt, r * D + d, fmaf((float)v_q[r * D + d], qparams[r * 4 + 2], qparams[r * 4 + 3]));
    }
  return CC_OK;
}

int cc_decode_step_quant_cpu(const cc_kv_view* c, float* qparams, int32_t n_bit, int32_t policy, const void* q, const void* k_new,
                             const void* v_new, const int32_t* input_pos, double* num, int32_t* denom, int64_t* counter,
                             const float* rand_next, uint64_t* next_key, int32_t g, int32_t w, int32_t HQ, float scale, void* y,
                             void* attn_out, void* workspace, size_t workspace_bytes, cc_stream_t stream, int32_t phases) {
  if (!view_ok(c) || !qparams || !q || !k_new || !v_new || !input_pos || !next_key || !y) return CC_ERR_BAD_ARG;
  if (n_bit != 8 || (phases & 3) != 3 || (policy != 1 && policy != 2 && policy != 3)) return CC_ERR_UNSUPPORTED;
  const int H = c->H, S = c->S, D = c->D, dt = c->dtype;
  const size_t es = dt_size(dt), n = (size_t)H * S * D;
  void* kw = malloc(n * es);
  void* vw = malloc(n * es);
  void* kn = malloc((size_t)H * D * es);
  void* vn = malloc((size_t)H * D * es);
  uint8_t* img = (uint8_t*)malloc((size_t)2 * H * D);
  float* par = (float*)malloc((size_t)H * 4 * sizeof(float));
  int64_t* idx = (int64_t*)malloc((size_t)H * sizeof(int64_t));
  int rc = CC_ERR_BAD_ARG;
  if (kw && vw && kn && vn && img && par && idx) {
    rc = cc_kv_dequant_rows_cpu((const uint8_t*)c->k_cache, (const uint8_t*)c->v_cache, qparams, H, S, D, dt, 8, kw, vw, stream);
    /* the slot every head writes this step: the minimum of its partial keys (what the step itself will take) */
    const int nk = cc_hh_next_key_slots_cpu(S);
    for (int h = 0; h < H && rc == CC_OK; h++) {
      const uint64_t* row = next_key + (size_t)(c->Hp == 1 ? 0 : h) * nk;
      uint64_t key = ~(uint64_t)0;
      for (int i = 0; i < nk; i++)
        if (row[i] < key) key = row[i];
      if (key == ~(uint64_t)0) rc = CC_ERR_BAD_ARG;
      idx[h] = (int64_t)((key & 0xffffffffu) >> 1);
    }
    if (rc == CC_OK) {
      /* the new token's image, and the values every later read (this step's attention included) sees */
      for (int h = 0; h < H; h++) {
        quant_row8(k_new, dt, (size_t)h * D, D, img + (size_t)h * D, &par[h * 4], &par[h * 4 + 1]);
        quant_row8(v_new, dt, (size_t)h * D, D, img + (size_t)(H + h) * D, &par[h * 4 + 2], &par[h * 4 + 3]);
        for (int d = 0; d < D; d++) {
          st(kn, dt, (size_t)h * D + d, fmaf((float)img[(size_t)h * D + d], par[h * 4], par[h * 4 + 1]));
          st(vn, dt, (size_t)h * D + d, fmaf((float)img[(size_t)(H + h) * D + d], par[h * 4 + 2], par[h * 4 + 3]));
        }
      }
      cc_kv_view t = *c;
      t.k_cache = kw;
      t.v_cache = vw;
      if (policy == 1)
        rc = cc_decode_step_heavy_hitter_cpu(&t, q, kn, vn, input_pos, num, denom, counter, next_key, g, w, HQ, scale, y, attn_out,
                                             workspace, workspace_bytes, stream);
      else if (policy == 2)
        rc = cc_decode_step_recent_global_cpu(&t, q, kn, vn, input_pos, next_key, g, HQ, scale, y, workspace, workspace_bytes, stream);
      else
        rc = cc_decode_step_random_cpu(&t, q, kn, vn, input_pos, rand_next, next_key, g, w, HQ, scale, y, workspace, workspace_bytes,
                                       stream);
    }
    if (rc == CC_OK)
      for (int h = 0; h < H; h++) {
        const size_t row = (size_t)h * S + (size_t)idx[h];
        memcpy((uint8_t*)c->k_cache + row * D, img + (size_t)h * D, (size_t)D);
        memcpy((uint8_t*)c->v_cache + row * D, img + (size_t)(H + h) * D, (size_t)D);
        memcpy(qparams + row * 4, par + h * 4, 4 * sizeof(float));
      }
  }
  free(kw); free(vw); free(kn); free(vn); free(img); free(par); free(idx);
  return rc;
}

/* ------------------------------------------------------------------------------------------------
 * Decode-time dense layers with the caller glue fused (see include/coldcompress.h: cc_gemv_fused).
 * Plain restatement of the eager chain (model.py:317-327, 375-387, 442-443, 452-457, 507-519): every tensor op
 * rounds to the model dtype; dot products accumulate in fp32 (summation order is unspecified -> tolerance class).
 * ---------------------------------------------------------------------------------------------- */
int cc_gemv_fused_cpu(const void* W, const void* W3, const void* x, const void* delta, const void* norm_w, float eps,
                      void* h_out, const void* bias, const void* freqs, int32_t rope_rows, int32_t head_dim, void* y, int32_t N,
                      int32_t K, int32_t dt, cc_stream_t stream) {
  (void)stream;
  if (!W || !x || !y || N <= 0 || K <= 0 || !dt_ok(dt)) return CC_ERR_BAD_ARG;
  if ((delta || h_out) && !norm_w) return CC_ERR_BAD_ARG;
  if (freqs && (W3 || rope_rows < 0 || rope_rows > N || head_dim <= 0 || (head_dim & 1) || (rope_rows % head_dim))) return CC_ERR_BAD_ARG;
  if (W3 && bias) return CC_ERR_BAD_ARG;
  float* in = (float*)malloc(sizeof(float) * (size_t)K);
  float* out = (float*)malloc(sizeof(float) * (size_t)N);
  if (!in || !out) { free(in); free(out); return CC_ERR_BAD_ARG; }
  if (norm_w) {
    double ss = 0.0;
    for (int k = 0; k < K; k++) {
      float h = ld(x, dt, (size_t)k);
      if (delta) h = rnd(h + ld(delta, dt, (size_t)k), dt);
      if (h_out) st(h_out, dt, (size_t)k, h);
      in[k] = h;
      ss += (double)h * (double)h;
    }
    const float rs = 1.0f / sqrtf((float)(ss / (double)K) + eps);
    for (int k = 0; k < K; k++) in[k] = rnd(rnd(in[k] * rs, dt) * ld(norm_w, dt, (size_t)k), dt);
  } else {
    for (int k = 0; k < K; k++) in[k] = ld(x, dt, (size_t)k);
  }
  for (int n = 0; n < N; n++) {
    double a = 0.0, a3 = 0.0;
    for (int k = 0; k < K; k++) {
      a += (double)ld(W, dt, (size_t)n * K + k) * (double)in[k];
      if (W3) a3 += (double)ld(W3, dt, (size_t)n * K + k) * (double)in[k];
    }
    float s = (float)a;
    if (bias) s += ld(bias, dt, (size_t)n);
    s = rnd(s, dt);
    if (W3) {
      const float s3 = rnd((float)a3, dt);
      const float sl = rnd(s / (1.0f + expf(-s)), dt);
      s = sl * s3;
    }
    out[n] = s;
  }
  if (freqs)
    for (int n = 0; n + 1 < rope_rows; n += 2) {
      const int pr = (n % head_dim) >> 1;
      const float c = ld(freqs, dt, (size_t)pr * 2), sn = ld(freqs, dt, (size_t)pr * 2 + 1);
      const float x0 = out[n], x1 = out[n + 1];
      out[n] = x0 * c - x1 * sn;
      out[n + 1] = x1 * c + x0 * sn;
    }
  for (int n = 0; n < N; n++) st(y, dt, (size_t)n, out[n]);
  free(in);
  free(out);
  return CC_OK;
}

/* Greedy sampling tail, ref: generation_utils.py:136-142: probs = dtype(softmax_fp32(logits)); idx = first index of
 * the largest rounded probability (torch.argmax).  The fp32 sum order is unspecified -> probabilities are a
 * tolerance class; the arg-max is exact given the probabilities. */
size_t cc_softmax_argmax_workspace_bytes_cpu(void) { return 0; }

int cc_softmax_argmax_cpu(const void* logits, int32_t V, int32_t dt, void* probs, int32_t* idx_out, void* workspace,
                          size_t workspace_bytes, cc_stream_t stream) {
  (void)stream; (void)workspace; (void)workspace_bytes;
  if (!logits || !probs || !idx_out || V <= 0 || !dt_ok(dt)) return CC_ERR_BAD_ARG;
  float mx = -INFINITY;
  for (int i = 0; i < V; i++) { const float x = ld(logits, dt, (size_t)i); if (x > mx) mx = x; }
  double sum = 0.0;
  for (int i = 0; i < V; i++) sum += (double)expf(ld(logits, dt, (size_t)i) - mx);
  float best = -INFINITY;
  int bi = 0, nan_i = -1;
  for (int i = 0; i < V; i++) {
    const float p = rnd(expf(ld(logits, dt, (size_t)i) - mx) / (float)sum, dt);
    st(probs, dt, (size_t)i, p);
    if (p > best) { best = p; bi = i; }
    if (p != p && nan_i < 0) nan_i = i;  /* torch.argmax: NaN is the maximum, the first one wins */
  }
  *idx_out = nan_i >= 0 ? nan_i : bi;
  return CC_OK;
}

/* Two-launch decode step of the head-constant ring policies (recent_global / full), pipeline form: the slot for this
 * position was chosen at the end of the previous step (or by cc_rg_next_key_init); insert, attend, choose the next.
 * ref: cache.py:493-502, 527-556 + model.py:389-418; only WHEN the arg-min runs differs from the three-call sequence. */
static uint64_t rg_key(const cc_kv_view* c, int32_t g) {
  const int64_t s = argmin_i32(c->pos + g, c->S - g) + g;
  const int32_t ps = c->pos[s];
  return ((uint64_t)((uint32_t)ps ^ 0x80000000u) << 32) | ((uint64_t)(uint32_t)s << 1) | (uint64_t)(ps == -1);
}

int cc_rg_next_key_init_cpu(const cc_kv_view* c, const int32_t* input_pos, int32_t g, uint64_t* next_key, cc_stream_t stream) {
  (void)stream;
  if (!view_ok(c) || !input_pos || !next_key || c->Hp != 1 || g < 0 || g >= c->S) return CC_ERR_BAD_ARG;
  const int nk = cc_hh_next_key_slots_cpu(c->S);
  for (int i = 1; i < nk; i++) next_key[i] = ~(uint64_t)0;
  next_key[0] = rg_key(c, g);
  return CC_OK;
}

int cc_decode_step_recent_global_cpu(const cc_kv_view* c, const void* q, const void* k_new, const void* v_new,
                                     const int32_t* input_pos, uint64_t* next_key, int32_t g, int32_t HQ, float scale, void* y,
                                     void* workspace, size_t workspace_bytes, cc_stream_t stream) {
  if (!view_ok(c) || !q || !k_new || !v_new || !input_pos || !next_key || !y || c->Hp != 1 || g < 0 || g >= c->S) return CC_ERR_BAD_ARG;
  const int nk = cc_hh_next_key_slots_cpu(c->S);
  uint64_t key = ~(uint64_t)0;
  for (int i = 0; i < nk; i++)
    if (next_key[i] < key) key = next_key[i];
  if (key == ~(uint64_t)0) return CC_ERR_BAD_ARG;
  int64_t idx = (int64_t)((key & 0xffffffffu) >> 1);
  insert_token(c, k_new, v_new, *input_pos, &idx);
  int rc = cc_decode_attn_gqa_cpu(q, c->k_cache, c->v_cache, c->mask, HQ, c->H, c->S, c->D, c->dtype, scale, y, NULL, NULL, NULL,
                                  NULL, NULL, workspace, workspace_bytes, stream);
  if (rc != CC_OK) return rc;
  for (int i = 1; i < nk; i++) next_key[i] = ~(uint64_t)0;
  next_key[0] = rg_key(c, g);
  return CC_OK;
}

/* Twin of the recoverable head-constant steps: the oracle has nothing to time out — it runs the step once and records the
 * commit; asked to REPLAY a committed position it declines (the replay is a property of the device's launch, tested there). */
int cc_decode_step_head_constant_rc_cpu(const cc_kv_view* c, int32_t policy, const void* q, const void* k_new, const void* v_new,
                                        const int32_t* input_pos, const float* rand_next, uint64_t seed, uint64_t* next_key,
                                        int32_t* step_commit, int32_t g, int32_t w, int32_t HQ, float scale, void* y,
                                        void* workspace, size_t workspace_bytes, cc_stream_t stream) {
  if (!view_ok(c) || !input_pos || (policy != 2 && policy != 3) || (policy == 2 && rand_next)) return CC_ERR_BAD_ARG;
  if (step_commit)
    for (int h = 0; h < c->H; h++)
      if (step_commit[h * CC_RC_STRIDE + 2] == *input_pos) return CC_ERR_UNSUPPORTED;
  int rc;
  if (policy == 2)
    rc = cc_decode_step_recent_global_cpu(c, q, k_new, v_new, input_pos, next_key, g, HQ, scale, y, workspace, workspace_bytes, stream);
  else if (rand_next)
    rc = cc_decode_step_random_cpu(c, q, k_new, v_new, input_pos, rand_next, next_key, g, w, HQ, scale, y, workspace, workspace_bytes,
                                   stream);
  else
    rc = cc_decode_step_random_rng_cpu(c, q, k_new, v_new, input_pos, seed, next_key, g, w, HQ, scale, y, workspace, workspace_bytes,
                                       stream);
  if (rc == CC_OK && step_commit)
    for (int h = 0; h < c->H; h++) {
      step_commit[h * CC_RC_STRIDE] = -1;
      step_commit[h * CC_RC_STRIDE + 1] = step_commit[h * CC_RC_STRIDE + 2] = *input_pos;
    }
  return rc;
}

/* The layer step with the layer's QKV projection in front of it (include/coldcompress.h: cc_decode_step_qkv_rc) — ref: model.py:375-387
 * (wqkv -> split -> apply_rotary_emb on q and k), then :389-427.  On the CPU simply the two restatements in sequence: cc_gemv_fused_cpu
 * (the projection is a tolerance class: fp32 summation order unspecified) and the recoverable step twin.  The device's single launch is
 * a scheduling matter; "available" here only restates the shape rule. */
int32_t cc_decode_step_qkv_available_cpu(int32_t HQ, int32_t H, int32_t S, int32_t D, int32_t dt, int32_t K) {
  return HQ > 0 && H > 0 && HQ % H == 0 && (HQ / H == 4 || HQ / H == 8) && S > 0 && D == 128 && (dt == CC_DT_BF16 || dt == CC_DT_F16) &&
         K >= 8 && K % 8 == 0 && K <= 4096;
}
int cc_decode_step_qkv_rc_cpu(const cc_kv_view* c, int32_t policy, const void* wqkv, const void* bias, const void* x, const void* delta,
                              const void* norm_w, float eps, void* h_out, const void* freqs, int32_t K, void* qkv_out,
                              const int32_t* input_pos, double* num, int32_t* denom, int64_t* counter, const float* rand_next,
                              uint64_t seed, uint64_t* next_key, int32_t* step_commit, int32_t g, int32_t w, int32_t HQ, float scale,
                              void* y, void* workspace, size_t workspace_bytes, cc_stream_t stream) {
  if (!view_ok(c) || !wqkv || !x || !norm_w || !input_pos || !next_key || !y || HQ <= 0 || HQ % c->H || K <= 0) return CC_ERR_BAD_ARG;
  const int N = (HQ + 2 * c->H) * c->D;
  const size_t es = c->dtype == CC_DT_F32 ? 4 : 2;
  char* qkv = (char*)malloc((size_t)N * es);
  if (!qkv) return CC_ERR_BAD_ARG;
  int rc = cc_gemv_fused_cpu(wqkv, NULL, x, delta, norm_w, eps, h_out, bias, freqs, freqs ? (HQ + c->H) * c->D : 0, c->D, qkv, N, K,
                             c->dtype, stream);
  if (rc == CC_OK) {
    if (qkv_out) memcpy(qkv_out, qkv, (size_t)N * es);
    const void* q = qkv;
    const void* kn = qkv + (size_t)HQ * c->D * es;
    const void* vn = qkv + (size_t)(HQ + c->H) * c->D * es;
    if (policy == 1)
      rc = cc_decode_step_heavy_hitter_rc_cpu(c, q, kn, vn, input_pos, num, denom, counter, next_key, step_commit, g, w, HQ, scale, y,
                                              workspace, workspace_bytes, stream, 3);
    else
      rc = cc_decode_step_head_constant_rc_cpu(c, policy, q, kn, vn, input_pos, rand_next, seed, next_key, step_commit, g, w, HQ, scale,
                                               y, workspace, workspace_bytes, stream);
  }
  free(qkv);
  return rc;
}
