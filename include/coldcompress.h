/*
 * coldcompress.h — C ABI of the MI355X-native KV-cache eviction + pruned-cache attention path.
 *
 * This is the drop-in boundary.  The reference (AnswerDotAI/cold-compress) has no
 * FFI of its own: its boundary is the Python class surface in cache.py / attention_utils.py /
 * prompt_compression.py.  The Python mirror of that surface lives in cold_compress_amd/ and reaches the
 * device ONLY through the entry points declared here (ctypes, see cold_compress_amd/_abi.py and
 * INTEGRATION.md).  Each entry point cites the reference code it replaces as
 * "ref: <file>:<lines>" (paths relative to the reference checkout).
 *
 * Conventions
 *  - Every pointer is a DEVICE pointer (HBM) unless stated otherwise.  Batch is 1 (ref: model.py:188-189).
 *  - `stream` is a hipStream_t passed as void*.  No entry point synchronises the stream, allocates or
 *    frees memory; scratch is supplied by the caller (torch's caching allocator) and sized by the
 *    matching *_workspace_bytes() query.  All entry points are hipGraph-capturable.
 *  - Return value: CC_OK (0) or a negative CC_ERR_* code.  Nothing throws.
 *  - `input_pos` for decode is a device int32[1] (the reference's decode `input_pos` tensor,
 *    generation_utils.py:482) so that captured graphs replay with an advancing position.
 *  - Arg-min rule everywhere: lowest slot index wins ties, NaN counts as the minimum (torch.argmin).
 *  - dtype codes: element type of K/V/q/probabilities ("model dtype" in the reference).
 *
 * The CPU oracle (oracle/cc_oracle.c) exports the same functions with a `_cpu` suffix and identical
 * signatures taking HOST pointers; it is test infrastructure only.
 */
#ifndef COLDCOMPRESS_H
#define COLDCOMPRESS_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CC_ABI_VERSION 1

#define CC_OK 0
#define CC_ERR_BAD_ARG (-1)      /* null pointer / non-positive size / inconsistent shape           */
#define CC_ERR_UNSUPPORTED (-2)  /* dtype / head_dim / window size the kernels are not built for     */
#define CC_ERR_HIP (-3)          /* hipGetLastError() != hipSuccess after a launch                   */
#define CC_ERR_WORKSPACE (-4)    /* workspace too small                                              */

#define CC_DT_F32 0
#define CC_DT_BF16 1
#define CC_DT_F16 2

/* priority dtypes for cc_topk_keep (prompt compaction) */
#define CC_PRIO_F32 0
#define CC_PRIO_BF16 1
#define CC_PRIO_F16 2
#define CC_PRIO_I64 3

typedef void* cc_stream_t;

/* View of one layer's cache buffers; mirrors the nn.Module buffers of ref: cache.py:178-227.
 *   k_cache, v_cache : [H, S, D] model dtype           (cache.py:185-201)
 *   pos              : [Hp, S] int32, -1 = empty slot; Hp = H if head_specific else 1 (cache.py:207-218)
 *   mask             : [H, S] bool (1 byte)            (cache.py:226-227)
 *   cache_cts        : [Hc] int32; Hc = H if variable_length else 1 (cache.py:219-222)
 */
typedef struct cc_kv_view {
  void* k_cache;
  void* v_cache;
  int32_t* pos;
  uint8_t* mask;
  int32_t* cache_cts;
  int32_t H;
  int32_t Hp;
  int32_t Hc;
  int32_t S;
  int32_t D;
  int32_t dtype;
} cc_kv_view;

int cc_abi_version(void);
const char* cc_error_string(int code);
/* Device properties of the current HIP device (host ints out); used by bench.py / DESIGN numbers. */
int cc_device_info(int* n_cu, int* wave_size, int* lds_bytes_per_cu, char* name, int name_len);

/* ------------------------------------------------------------------------------------------------
 * Decode-time update_kv: choose the slot to overwrite, then insert the new token in place.
 * ref: KVCache.update_kv cache.py:314-340 -> _decoding_update :348-364 -> _eviction_idx + _fill.
 *
 * Common behaviour (Appendix A of SURVEY.md):
 *   idx[h] chosen per policy; num_ins[h] = (pos[h or 0][idx]== -1); then
 *   pos[.,idx] <- *input_pos; K[h,idx,:] <- k_new[h,:]; V likewise; mask[h,idx] <- 1;
 *   cache_cts[j] += num_ins[j] for j < Hc                     (cache.py:330, 356-362, 390-401, 460-490)
 * k_new/v_new: [H, D] model dtype.  idx_out: int64 [Hp] (what _eviction_idx returns).
 * If k_new == NULL the call is "select only": idx_out is produced (plus the policy's own side effects,
 * e.g. heavy-hitter history zeroing, which the reference performs inside _eviction_idx) and nothing is
 * inserted.
 * ---------------------------------------------------------------------------------------------- */

/* ref: KVCacheFull._eviction_idx cache.py:500-502 — argmin over pos (first empty slot). Hp must be 1. */
int cc_decode_update_full(const cc_kv_view* c, const void* k_new, const void* v_new,
                          const int32_t* input_pos, int64_t* idx_out, cc_stream_t stream);

/* ref: KVCacheRecentGlobal._eviction_idx cache.py:552-556 — g + argmin(pos[g:]). Hp must be 1. */
int cc_decode_update_recent_global(const cc_kv_view* c, const void* k_new, const void* v_new,
                                   const int32_t* input_pos, int32_t global_tokens, int64_t* idx_out,
                                   cc_stream_t stream);

/* Generic base path for caller-supplied importances.
 * ref: KVCache._eviction_idx cache.py:366-379 — scores[:, :g]=+inf; pos==-1 -> -inf; argmin.
 * scores: [Hs, S] of score_dtype (CC_DT_*), Hs = Hp or 1 (a 1-D score vector is broadcast like
 * scores.unsqueeze(0), cache.py:369-370, and then Hp must be 1).  Scores are NOT modified. */
int cc_decode_update_scores(const cc_kv_view* c, const void* k_new, const void* v_new,
                            const int32_t* input_pos, const void* scores, int32_t score_dtype,
                            int32_t global_tokens, int64_t* idx_out, cc_stream_t stream);

/* ref: KVCacheRandom._token_importances cache.py:519-524 + base _eviction_idx :366-379.
 * rand_u: [S] f32 uniform [0,1) drawn by the caller (the RNG stream is backend-specific; parity is
 * defined given this vector).  pos[s] >= *input_pos - recent_window -> +inf. Hp must be 1. */
int cc_decode_update_random(const cc_kv_view* c, const void* k_new, const void* v_new,
                            const int32_t* input_pos, const float* rand_u, int32_t global_tokens,
                            int32_t recent_window, int64_t* idx_out, cc_stream_t stream);

/* ref: KVCacheL2 cache.py:580-605 — score = dtype(max(key_norm) - key_norm) with the max taken over
 * ALL heads and slots, recent window -> +inf, then base rules; on insert key_norm[h,idx] <- ||k_new[h]||2
 * (fp32 accumulate, rounded to model dtype).  key_norm: [H, S] model dtype.  Hp must be H.
 * workspace: cc_decode_update_l2_workspace_bytes(H,S) bytes. */
size_t cc_decode_update_l2_workspace_bytes(int32_t H, int32_t S);
int cc_decode_update_l2(const cc_kv_view* c, const void* k_new, const void* v_new,
                        const int32_t* input_pos, void* key_norm, int32_t global_tokens,
                        int32_t recent_window, int64_t* idx_out, void* workspace, size_t workspace_bytes,
                        cc_stream_t stream);

/* ref: KVCacheHeavyHitter._eviction_idx cache.py:725-765 (history_window_size W == 1, float64 history).
 *   avg = f32(num) / f32(max(denom,1));  (pos<g)|(pos>=p-w) -> 1.0;  pos==-1 -> 0.0;  argmin;
 *   num[h,idx] <- 0; denom[h,idx] <- 0; then the common insert.
 * num: [H, S] float64; denom: [H, S] int32.  Hp must be H. */
int cc_decode_update_heavy_hitter(const cc_kv_view* c, const void* k_new, const void* v_new,
                                  const int32_t* input_pos, double* num, int32_t* denom,
                                  int32_t global_tokens, int32_t recent_window, int64_t* idx_out,
                                  cc_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Heavy-hitter history update.  ref: KVCacheHeavyHitter.update_state cache.py:690-723 (W == 1).
 *   num[h,s] += (double)attn[h,s] for s < T (zero padding to S beyond T); denom[h,s] += 1 for ALL s;
 *   *counter += 1.  attn: [H, T] model dtype, already averaged over the query group.
 * ---------------------------------------------------------------------------------------------- */
int cc_hh_update(double* num, int32_t* denom, int64_t* counter, const void* attn, int32_t H, int32_t S,
                 int32_t T, int32_t dtype, cc_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Decode attention over the pruned cache, GQA-aware (K/V read once, no repeat_interleave).
 * ref: model.py:395-418 + attention_utils.py:27-54.
 *   q: [HQ, D]; k,v: [H, S, D]; mask: [H, S] bool or NULL; R = HQ/H; query head j uses kv head j/R.
 *   score = dtype(dtype(q.k) * scale) (+ -inf where mask==0); P = dtype(softmax_fp32(score));
 *   y[HQ, D] = dtype(sum_s P*v);  probs_out[HQ, S] = P (optional, may be NULL);
 *   attn_out[H, S] = dtype(mean over the R heads of the group of P) (optional, may be NULL).
 * For CC_DT_F32 the "dtype()" roundings are identities.
 * If hh_num != NULL the heavy-hitter history update (cc_hh_update with T = S) is fused into the
 * combine pass: hh_num += attn_out, hh_denom += 1, *hh_counter += 1.
 * ---------------------------------------------------------------------------------------------- */
size_t cc_decode_attn_workspace_bytes(int32_t HQ, int32_t H, int32_t S, int32_t D, int32_t dtype);
int cc_decode_attn_gqa(const void* q, const void* k, const void* v, const uint8_t* mask, int32_t HQ,
                       int32_t H, int32_t S, int32_t D, int32_t dtype, float scale, void* y,
                       void* attn_out, void* probs_out, double* hh_num, int32_t* hh_denom,
                       int64_t* hh_counter, void* workspace, size_t workspace_bytes,
                       cc_stream_t stream);

/* The same attention with the W > 1 history ring folded into the combine pass (cf. hh_num/hh_denom/hh_counter above,
 * which fold the W == 1 history): ring[h, s, *counter % W] = group-mean attention; denom += 1; *counter += 1
 * (ref: cache.py:716-723), and the tracked window-sum state (wsum_acc / wsum, see cc_hh_ring_update) kept current.
 * Equivalent to cc_decode_attn_gqa(attn_out) followed by cc_hh_ring_update(attn = attn_out, T = S, tracked). */
int cc_decode_attn_gqa_ring(const void* q, const void* k, const void* v, const uint8_t* mask, int32_t HQ, int32_t H,
                            int32_t S, int32_t D, int32_t dtype, float scale, void* y, void* attn_out, void* ring_num,
                            int32_t* denom, int64_t* counter, int32_t W, uint64_t* wsum_acc, float* wsum,
                            void* workspace, size_t workspace_bytes, cc_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Fused heavy-hitter decode step (history_window_size == 1): update_kv + attention + update_state of one layer
 * in TWO launches instead of three.  Same reference lines as cc_decode_update_heavy_hitter, cc_decode_attn_gqa
 * and cc_hh_update; the only difference is WHEN the arg-min runs:
 *   launch 1 (K/V streaming pass): the insert of this step's token is folded into its prologue — the slot is the
 *     minimum over the partial arg-min keys next_key[h][0 .. NK) left by the previous step (or by
 *     cc_hh_next_key_init);
 *   launch 2 (combine): y, group-averaged probabilities, history update, and the arg-min for position
 *     *input_pos + 1 evaluated on the freshly updated history: one partial minimum per 128-slot chunk
 *     -> next_key[h][chunk] (plain stores; no atomics, no reset pass).
 * next_key: uint64 [H, NK], NK = cc_hh_next_key_slots(S) = the row stride: eight LIVE entries per 128-slot chunk (the combine pass
 * fills the first eighth of them and keeps the others at ~0; the single-launch step below publishes one key per WAVE of its
 * 64-slot workgroups, so that no wave waits for another at the end of the launch) plus, r6, a TAIL of eight entries no arg-min
 * ever reads — per-head state a policy's step hands to the next step of the same cache (l2: the head's carried norm record,
 * written and read by the r6 A/B build only, -DCC_V_L2CARRY=1: nothing touches the tail in the product);
 * entry = (orderable(score) << 32) | slot << 1 | was_empty, ~0 = "no candidate".  Valid as long as positions advance by one and nothing else mutates pos / history in
 * between; re-seed with cc_hh_next_key_init otherwise.
 * Results are bit-identical to the three-call sequence (tests/test_gpu_fused_step.py).
 * ---------------------------------------------------------------------------------------------- */
int32_t cc_hh_next_key_slots(int32_t S);
int cc_hh_next_key_init(const cc_kv_view* c, const int32_t* input_pos, const double* num, const int32_t* denom,
                        int32_t global_tokens, int32_t recent_window, uint64_t* next_key, cc_stream_t stream);
int cc_decode_step_heavy_hitter(const cc_kv_view* c, const void* q, const void* k_new, const void* v_new,
                                const int32_t* input_pos, double* num, int32_t* denom, int64_t* counter,
                                uint64_t* next_key, int32_t global_tokens, int32_t recent_window, int32_t HQ,
                                float scale, void* y, void* attn_out, void* workspace, size_t workspace_bytes,
                                cc_stream_t stream);
/* The same two-launch step for the head-constant ring policies: KVCacheRecentGlobal (cache.py:527-556: arg-min of
 * pos over the slots behind the first `global_tokens` sinks; empty slots, pos == -1, first) and KVCacheFull
 * (cache.py:493-502: global_tokens = 0).  c->Hp must be 1; next_key: uint64 [H, NK] — one row per kv head, all holding the same
 * keys: every head reads and rewrites its own copy, so that no workgroup of one head can read a row another head's workgroups
 * have already rewritten for the next position (the single-launch form has no launch boundary between the two).  No history,
 * no group-mean output. */
int cc_rg_next_key_init(const cc_kv_view* c, const int32_t* input_pos, int32_t global_tokens, uint64_t* next_key,
                        cc_stream_t stream);
int cc_decode_step_recent_global(const cc_kv_view* c, const void* q, const void* k_new, const void* v_new,
                                 const int32_t* input_pos, uint64_t* next_key, int32_t global_tokens, int32_t HQ,
                                 float scale, void* y, void* workspace, size_t workspace_bytes, cc_stream_t stream);
/* The same two-launch step for KVCacheRandom (cache.py:505-524: uniform scores, the `recent_window` newest positions
 * -> +inf, then the base rules cache.py:373-376).  `rand_u` / `rand_next`: float32 [S] uniform draws — the init call takes
 * the draw for position *input_pos, every step the draw for position *input_pos + 1 (the reference draws one vector
 * per step, cache.py:521; the order of draws is unchanged).  c->Hp must be 1; next_key: uint64 [H, NK] (as cc_decode_step_recent_global). */
int cc_random_next_key_init(const cc_kv_view* c, const int32_t* input_pos, const float* rand_u, int32_t global_tokens,
                            int32_t recent_window, uint64_t* next_key, cc_stream_t stream);
int cc_decode_step_random(const cc_kv_view* c, const void* q, const void* k_new, const void* v_new,
                          const int32_t* input_pos, const float* rand_next, uint64_t* next_key, int32_t global_tokens,
                          int32_t recent_window, int32_t HQ, float scale, void* y, void* workspace,
                          size_t workspace_bytes, cc_stream_t stream);
/* KVCacheRandom's fused step with the uniform draws made IN the kernels (r3): ref cache.py:519-524 draws torch.rand(S) per
 * eviction — a backend-specific stream, so parity is defined given the vector (cc_decode_step_random, the injection point of the
 * tests); a product run needs no vector and no extra launch: the draw for slot s at position p is
 *   u(seed, p, s) = (mix64(seed + p * 0x9E3779B97F4A7C15 + s) >> 40) * 2^-24   (24 bits, like torch.rand's float32)
 * with mix64 = the murmur3 64-bit finaliser applied twice — stateless (a hipGraph replay advances with *input_pos), identical in
 * the single-launch step, the two-launch step, cc_random_next_key_init_rng and the oracle's twins. */
int cc_random_next_key_init_rng(const cc_kv_view* c, const int32_t* input_pos, uint64_t seed, int32_t global_tokens,
                                int32_t recent_window, uint64_t* next_key, cc_stream_t stream);
int cc_decode_step_random_rng(const cc_kv_view* c, const void* q, const void* k_new, const void* v_new, const int32_t* input_pos,
                              uint64_t seed, uint64_t* next_key, int32_t global_tokens, int32_t recent_window, int32_t HQ,
                              float scale, void* y, void* workspace, size_t workspace_bytes, cc_stream_t stream);
/* The head-constant steps with the RECOVERABLE hand-off (late r3; semantics as cc_decode_step_heavy_hitter_rc below: commit words,
 * replay of committed heads, no-op behind a set status word).  policy: 2 = recent_global / full (rand_next must be NULL), 3 = random
 * (rand_next: the uniform vector for position *input_pos + 1, or NULL: the draws are made in the kernels from `seed`, as
 * cc_decode_step_random_rng — the only form a retry may use: a retried step must score the same draw).  step_commit: int32
 * [H, cc_decode_step_commit_stride()], -1 = nothing committed; may be NULL (no replay).  The rows of next_key [H, NK] stay identical across kv heads through a
 * recovered fault: every head scores the shared positions and the same draws.  The position row (shared) is written by kv head
 * 0's inserting workgroup ahead of the hand-off — idempotent on a retry; the count by kv head 0 at its commit. */
int cc_decode_step_head_constant_rc(const cc_kv_view* c, int32_t policy, const void* q, const void* k_new, const void* v_new,
                                    const int32_t* input_pos, const float* rand_next, uint64_t seed, uint64_t* next_key,
                                    int32_t* step_commit, int32_t global_tokens, int32_t recent_window, int32_t HQ, float scale,
                                    void* y, void* workspace, size_t workspace_bytes, cc_stream_t stream);
/* The same two-launch step for KVCacheL2 (cache.py:559-612: score = dtype(max over ALL heads' and slots' key norms -
 * norm), recent window -> +inf, base rules; the inserted key's norm recorded, cache.py:592-593).  The global maximum
 * is folded across the step boundary: the streaming pass publishes per-wave maxima of the surviving norms, the
 * combine pass folds them with the freshly inserted ones.  16-bit caches with head_dim 128 only (CC_ERR_UNSUPPORTED
 * otherwise: use cc_decode_update_l2 + cc_decode_attn_gqa).  c->Hp must be H; key_norm: [H, S] model dtype;
 * next_key: uint64 [H, NK]. */
int cc_l2_next_key_init(const cc_kv_view* c, const int32_t* input_pos, void* key_norm, int32_t global_tokens,
                        int32_t recent_window, uint64_t* next_key, cc_stream_t stream);
int cc_decode_step_l2(const cc_kv_view* c, const void* q, const void* k_new, const void* v_new, const int32_t* input_pos,
                      void* key_norm, uint64_t* next_key, int32_t global_tokens, int32_t recent_window, int32_t HQ,
                      float scale, void* y, void* workspace, size_t workspace_bytes, cc_stream_t stream);
/* ... with the recoverable hand-off (r4; step_commit as in cc_decode_step_heavy_hitter_rc).  The norm maximum crosses kv heads, but
 * every workgroup republishes its own maximum on a retry and the inserted key's norm is stored with the insert (idempotent), so a
 * retried head scores against the same maximum the committed ones did. */
int cc_decode_step_l2_rc(const cc_kv_view* c, const void* q, const void* k_new, const void* v_new, const int32_t* input_pos,
                         void* key_norm, uint64_t* next_key, int32_t* step_commit, int32_t global_tokens, int32_t recent_window,
                         int32_t HQ, float scale, void* y, void* workspace, size_t workspace_bytes, cc_stream_t stream);
/* The same two-launch step for KVCacheHeavyHitter with a finite history window (history_window_size W > 1): the score is
 * dtype(sum_W ring) / clamp(denom, 1, W) over the TRACKED window sums (see cc_hh_ring_update); the evicted slot's ring
 * row, shadow column, accumulator and denom restart from zero inside the combine pass (cache.py:754-763), which also
 * records this step's attention (cache.py:716-723).  ring_num / denom / counter / wsum_acc / wsum as for
 * cc_decode_attn_gqa_ring; next_key: uint64 [H, NK]; attn_out optional. */
int cc_hh_ring_next_key_init(const cc_kv_view* c, const int32_t* input_pos, const int32_t* denom, int32_t W, const float* wsum,
                             int32_t global_tokens, int32_t recent_window, uint64_t* next_key, cc_stream_t stream);
int cc_decode_step_heavy_hitter_ring(const cc_kv_view* c, const void* q, const void* k_new, const void* v_new,
                                     const int32_t* input_pos, void* ring_num, int32_t* denom, int64_t* counter, int32_t W,
                                     uint64_t* wsum_acc, float* wsum, uint64_t* next_key, int32_t global_tokens,
                                     int32_t recent_window, int32_t HQ, float scale, void* y, void* attn_out,
                                     void* workspace, size_t workspace_bytes, cc_stream_t stream);
/* ------------------------------------------------------------------------------------------------
 * Single-launch layer step.  cc_decode_step_heavy_hitter runs the whole step — insert, K/V streaming pass, softmax
 * normalisation, group mean, history update (cache.py:716-722), the arg-min for position *input_pos + 1 (cache.py:725-749)
 * and y — in ONE launch whenever cc_decode_step_single_launch(...) returns 1: 16-bit caches, head_dim 128, HQ / H in
 * {1, 2, 4, 8}, at most 64 workgroups per kv head (the launch plan's split count: n_split <= 64), each with one 64-slot tile per
 * workgroup (S <= 4096) or — HQ / H in {4, 8}, heavy hitter / recent_global / full / random — up to eight of them (S <= 32768
 * where the plan keeps 64 splits, i.e. 8 kv heads and more per rank), at most 32 kv heads, and the n_split * H workgroups all
 * resident on the device at once.  Every workgroup publishes its partial (m, l, O) through the workspace as
 * self-validating tagged granules (write-through stores), waits — bounded — for the other workgroups of its kv head,
 * and finishes its own 64 slots from the scores still in its registers.  pos, mask, cache_cts, K/V, num, denom, counter,
 * attn_out and the next keys are bit-identical to the two-launch step; y agrees up to fp32 summation order.
 * The kernels the single launch normally runs carry no measurement hooks and no attn_out store (0.3-0.5 us of the step): a call
 * that passes attn_out, sets a measurement bit of `phases` or runs while a trace buffer is installed (cc_decode_step_trace) is
 * served by a full-featured instantiation where one exists (bf16, HQ / H = 4), by the two-launch step otherwise — with
 * CC_PHASE_ONE_LAUNCH, CC_ERR_UNSUPPORTED (a trace buffer alone never changes which step runs: no stamps are written then).
 * Workspace contract for this mode: the first 4 KiB + 288 KiB + 16 MiB of every decode workspace (sized in by
 * cc_decode_attn_workspace_bytes, at fixed offsets whatever the shape, so that caches of different lengths may share one
 * workspace) hold per-head epoch words and the granules; they must be ZERO before first use and written by nobody
 * else; a launch that could not complete its hand-off (a device that did not keep the grid resident) sets the 32-bit
 * word at byte offset cc_decode_step_status_offset() to 1 — its results are then invalid.
 * phases for cc_decode_step_heavy_hitter_phases: 1 / 2 / 3 select the launches of the two-launch step as before
 * (3 = both = let the library choose single or two launches); | CC_PHASE_TWO_LAUNCH never uses the single launch;
 * CC_PHASE_ONE_LAUNCH demands it (CC_ERR_UNSUPPORTED when not possible).
 * ---------------------------------------------------------------------------------------------- */
#define CC_PHASE_TWO_LAUNCH 0x10000
#define CC_PHASE_ONE_LAUNCH 0x20000
int32_t cc_decode_step_single_launch(int32_t HQ, int32_t H, int32_t S, int32_t D, int32_t dtype);
int32_t cc_decode_step_status_offset(void);
/* r6: every wait of the in-launch hand-off is bounded by DEVICE TIME (the 100 MHz clock all XCDs share), not by a number of poll
 * rounds: a workgroup that has waited this many microseconds for its kv head's other workgroups gives up (status word, fail word:
 * below).  The failure latency a caller's recovery path sees is this plus the kernel's own few microseconds.  (r2-r5: 2^18 poll
 * rounds — 1.6 s measured with memory polls.)  ref: the reference has no hand-off (cache.py:725-765 runs in one eager op chain). */
int32_t cc_decode_step_wait_bound_us(void);
/* The heavy-hitter layer step with the RECOVERABLE hand-off (r3; per-workgroup commit words r4): cc_decode_step_heavy_hitter_phases
 * plus `step_commit`, int32 [H, cc_decode_step_commit_stride()] on the device, all -1 = nothing committed (reset it whenever
 * positions restart).  ref: the reference has no hand-off to time out (cache.py:725-765, 716-722); this is what makes ours safe.
 * Words of kv head h: [0] = (slot << 1) | was_empty of the slot position [1]'s insert went to (-1: none), [2 + split] = the last
 * position whose step workgroup `split` of the head has committed.  Single-launch form (early (m, l) hand-off; since late r4 also
 * the several-tiles-per-wave form, whose tail keeps the memory order — one gather — and commits EVERYTHING, y included, behind it):
 *   - a launch that finds the status word set returns at once (a step of this token failed: nothing is built on its output);
 *   - a workgroup that gives up waiting sets the status word and its head's fail word; a workgroup that reads the fail word with
 *     its last gather commits nothing; one that does not commits ITS part — its slots' history, its next-eviction keys, its word
 *     [2 + split] = *input_pos (split 0: also the head's count and the step counter) — and nothing else;
 *   - a workgroup whose word equals *input_pos recomputes its scores and partials (the hand-off needs them: same values, bit for
 *     bit) and stores nothing; every workgroup of a head whose word [1] equals *input_pos takes the insert slot from word [0] (the
 *     key row may already hold the next position's keys).  Whatever the interleaving of give-ups and commits inside the failed
 *     launch, every slot's history is updated exactly once per position: a retry is idempotent per workgroup.
 * So the caller recovers by clearing the status word, advancing the workspace's epoch words (attention_utils.
 * reset_single_launch_status does both) and running the SAME token again.  Other launch forms ignore step_commit (nothing in them
 * can time out).  step_commit may be NULL: no replay, counts bumped at the insert. */
int32_t cc_decode_step_commit_stride(void);
int cc_decode_step_heavy_hitter_rc(const cc_kv_view* c, const void* q, const void* k_new, const void* v_new,
                                   const int32_t* input_pos, double* num, int32_t* denom, int64_t* counter,
                                   uint64_t* next_key, int32_t* step_commit, int32_t global_tokens, int32_t recent_window,
                                   int32_t HQ, float scale, void* y, void* workspace, size_t workspace_bytes,
                                   cc_stream_t stream, int32_t phases);
/* The layer step WITH THE LAYER'S QKV PROJECTION FOLDED IN (r5) — ref: model.py:375-387 (wqkv, split, apply_rotary_emb), :452-457
 * (RMSNorm), then :389-427 as cc_decode_step_*_rc.  One launch: every workgroup requests the weight rows of its share of its kv
 * head's projection AHEAD of its K / V tile (the tile does not depend on q), computes them with cc_gemv_fused's arithmetic — the
 * residual add x + delta in the model dtype (h_out <- it, may be NULL), RMSNorm in fp32 with two roundings, fp32 dot chains in
 * cc_gemv_fused's order, bias, the Linear's rounding, RoPE on the (even, odd) pairs of the q and k rows: q / k_new / v_new are
 * bit-identical to cc_gemv_fused(wqkv, ..., rope_rows = (HQ + H) * D) — hands them to the head's workgroups through tagged
 * granules (the step's own transports) and runs the recoverable single-launch step on them; the cache's 2 * H * S * D * 2 bytes
 * stream in the shadow of the (HQ + 2H) * D * K * 2 bytes of weights.
 *   wqkv [(HQ + 2H) * D, K] row major (q heads, k heads, v heads), bias [(HQ + 2H) * D] or NULL, x / delta (may be NULL) /
 *   norm_w [K], freqs [D / 2, 2] (cos, sin) of *input_pos in the model dtype (NULL: no RoPE), qkv_out [(HQ + 2H) * D] or NULL
 *   (a plain copy of the projection; nothing in the step reads it).
 *   policy: 1 = heavy hitter (num, denom, counter; c->Hp == H), 2 = recent_global / full, 3 = random (rand_next or, NULL, the
 *   in-kernel draws of `seed`) — the arguments of cc_decode_step_heavy_hitter_rc / cc_decode_step_head_constant_rc.
 * Shapes: cc_decode_step_qkv_available(HQ, H, S, D, dtype, K) != 0 — 16-bit caches, D = 128, HQ / H in {4, 8}, one 16-row tile
 * per wave, K % 8 == 0 and K <= 4096, all workgroups resident; CC_ERR_UNSUPPORTED otherwise (the caller runs cc_gemv_fused and
 * the plain step).  A retry of a failed token recomputes the projection (x and delta are not modified: idempotent). */
int32_t cc_decode_step_qkv_available(int32_t HQ, int32_t H, int32_t S, int32_t D, int32_t dtype, int32_t K);
int cc_decode_step_qkv_rc(const cc_kv_view* c, int32_t policy, const void* wqkv, const void* bias, const void* x, const void* delta,
                          const void* norm_w, float eps, void* h_out, const void* freqs, int32_t K, void* qkv_out,
                          const int32_t* input_pos, double* num, int32_t* denom, int64_t* counter, const float* rand_next,
                          uint64_t seed, uint64_t* next_key, int32_t* step_commit, int32_t global_tokens, int32_t recent_window,
                          int32_t HQ, float scale, void* y, void* workspace, size_t workspace_bytes, cc_stream_t stream);
/* The L2-resident hand-off of the single-launch step (r4).  ref: nothing in the reference corresponds — it is HOW the one launch that
 * replaces cache.py:725-765 + model.py:395-418 + cache.py:716-722 exchanges its partials.  On a device whose dispatcher puts block b
 * of a launch on XCD b % 8 (MI355X in SPX mode), a cache with a multiple of 8 kv heads runs its single-tile step with kv head =
 * block % H: the 16 .. 64 workgroups of a kv head share one XCD, and the head's (m, l) and partial-O granules travel through that
 * XCD's L2 (plain stores, sc1 polls) instead of through memory (write-through stores, sc0 sc1 polls): 9.4 -> 8.5 us per layer step
 * at the Llama-3-8B / cache_len 4096 shape together with the other r4 changes (profiles/r04_ab_step_variants.md).  Results are
 * bit-identical to the memory hand-off (same arithmetic, same fixed orders).  Safety: what the placement needs — blocks b and b + 8
 * of one launch share an XCD; which XCD block 0 lands on varies from launch to launch and is not relied upon — is OBSERVED per device
 * (cc_decode_step_probe_xcd) before the form is ever chosen.  Should it not hold for some launch, a head's workgroups do not see each
 * other's granules: the bounded wait ends the step as a hand-off timeout (status word, nothing committed by the recoverable kinds),
 * a workgroup that gives up sets the head's fail word and its peers stop waiting when they read it, and the harness retries — from
 * the fourth attempt on with the memory hand-off (until the next generation).
 *   cc_decode_step_probe_xcd: synchronous, call OUTSIDE stream capture (the Python layer does so when it creates a decode
 *     workspace on a device); 1 = verified for the current device (cached; two grid shapes, four launches), 0 = refuted or
 *     not probed -> memory hand-off.
 *   cc_decode_step_l2_handoff(): 1 if verified on the current device AND neither demoted there (below) nor switched off by the A/B
 *     switch of include/coldcompress_debug.h.
 *   cc_decode_step_demote_l2_handoff(demoted): state of the CURRENT DEVICE only (r5; the recovery path's lever — the process-wide
 *     switch is a debug hook): 1 = every step launched on this device from now on takes the memory hand-off whatever the probe said,
 *     0 = the probe's verdict counts again.  Returns the previous value.  The forms give bit-identical cache state, so a flip never
 *     changes results; a step captured into a hipGraph keeps the form it was captured with. */
int32_t cc_decode_step_probe_xcd(void);
int32_t cc_decode_step_l2_handoff(void);
int32_t cc_decode_step_demote_l2_handoff(int32_t demoted);
/* The two-launch step for KVCacheHybrid (FastGen per-head policies; cache.py:896-1019 + the fused ring update of
 * cc_decode_attn_gqa_ring): three launches -> two.  What a head does with the incoming token — append at the end of its
 * live slots, evict its candidate, or drop the token (slot S - 1, mask untouched) — depends on its policy, its count, the
 * budget terms and whether the token is punctuation (cache.py:905-950): decided at the top of the K/V streaming pass, from
 * device scalars.  The CANDIDATE (arg-min over the head's live slots of window-sum / min(denom, W) or of pos, with the
 * protections of cache.py:876-889 applied) depends only on the state the previous step left behind: the combine pass
 * scores it for position *input_pos + 1 from the freshly updated window sums, commits the counts the inserts produced
 * and bumps num_punc (cache.py:1017).  cc_hybrid_next_key_init seeds the candidates.  Arguments as for
 * cc_hybrid_decode_update / cc_decode_attn_gqa_ring; ring_num may be NULL when no head scores by accumulated attention
 * (then denom / counter / wsum_acc / wsum are unused).  reset-history-on-evict is not part of this step (the reference's
 * effective behaviour, see cache.py of this package); 16-bit caches with head_dim 128 and at most 21 policy rows only
 * (CC_ERR_UNSUPPORTED otherwise: use cc_hybrid_decode_update + cc_decode_attn_gqa[_ring]).  next_key: uint64 [H, NK].
 * Bit-identical to the three-launch sequence (tests/test_gpu_hybrid.py). */
int cc_hybrid_next_key_init(const cc_kv_view* c, const int32_t* input_pos, const int64_t* strategies, const int32_t* policy_table,
                            int32_t n_policies, const int32_t* denom, int32_t W, const float* wsum, const uint8_t* special_mask,
                            const uint8_t* punc_mask, int32_t global_tokens, uint64_t* next_key, cc_stream_t stream);
int cc_decode_step_hybrid(const cc_kv_view* c, const void* q, const void* k_new, const void* v_new, const int32_t* input_pos,
                          const int64_t* strategies, const int32_t* policy_table, int32_t n_policies, void* ring_num, int32_t* denom,
                          int64_t* counter, int32_t W, uint64_t* wsum_acc, float* wsum, const uint8_t* special_mask,
                          uint8_t* punc_mask, const int64_t* token_id, const int64_t* punc_ids, int32_t n_punc_ids,
                          const int32_t* num_special, int32_t* num_punc, uint64_t* next_key, int32_t global_tokens, int32_t HQ,
                          float scale, void* y, void* attn_out, void* workspace, size_t workspace_bytes, cc_stream_t stream);
/* The same step with the recoverable hand-off's commit words (step_commit: int32 [H, cc_decode_step_commit_stride()], -1 before
 * the first step; null: cc_decode_step_hybrid).  Single-launch form: a workgroup that gives up waiting, or finds that a workgroup
 * of its kv head did, stores NOTHING of the step (no y, no ring column / window sum / denominator, no keys, no counts); the others
 * commit their part and mark it in their word.  The per-head decision of the position (slot, append / evict / drop, the head's
 * count before the step, the ring column) is recorded in words [0], [1], [66], [67] before anything can be committed, and a retry
 * of the position takes it from there: committed workgroups recompute (the hand-off needs their partials, y is written again)
 * and store nothing else.  The two-launch form ignores step_commit (nothing to time out). */
int cc_decode_step_hybrid_rc(const cc_kv_view* c, const void* q, const void* k_new, const void* v_new, const int32_t* input_pos,
                             const int64_t* strategies, const int32_t* policy_table, int32_t n_policies, void* ring_num,
                             int32_t* denom, int64_t* counter, int32_t W, uint64_t* wsum_acc, float* wsum, const uint8_t* special_mask,
                             uint8_t* punc_mask, const int64_t* token_id, const int64_t* punc_ids, int32_t n_punc_ids,
                             const int32_t* num_special, int32_t* num_punc, uint64_t* next_key, int32_t* step_commit,
                             int32_t global_tokens, int32_t HQ, float scale, void* y, void* attn_out, void* workspace,
                             size_t workspace_bytes, cc_stream_t stream);
/* cc_decode_step_hybrid runs as ONE launch (the single-launch form above: in-launch hand-off, every workgroup resident) when
 * this returns 1: 16-bit caches, head_dim 128, HQ / H in {4, 8}, up to eight 64-slot tiles per workgroup (S <= 32768 at 64
 * workgroups per kv head).  The tail then also does what the combine pass did for this policy: ring column, exact window sum and
 * denominator of every slot (on all lanes), the head's candidate for the next position, the head's count (committed by the
 * head's first workgroup once all of them have published), and the step counter / num_punc (committed by the LAST head to
 * complete, through a ticket word in the workspace header: the workgroups of every head read them in their prologue).
 * Bit-identical to the two-launch step in every buffer but y (one rounding, as for the other policies).
 * cc_decode_step_set_single_launch(0) switches it off; the status word is shared with the other single-launch steps. */
int32_t cc_decode_step_hybrid_single_launch(int32_t HQ, int32_t H, int32_t S, int32_t D, int32_t dtype);
/* ------------------------------------------------------------------------------------------------
 * Quantised KV cache, --cache_bits {8, 4, 2}.  ref: quantization_utils.py:4-98 (quantize_tensor /
 * dequantize_tensor with axis = 2), KVCache.quantize_cache / dequantize_cache cache.py:283-309, called around
 * every update (cache.py:323-338).  One (scale, zero point) per cache slot, shared by all heads and channels;
 * every elementwise op rounds to the cache dtype as torch does:
 *   scale[s] = dtype(max(dtype(max_s - min_s), dtype(1e-6)) / (2^n - 1));  zero[s] = dtype(min_s + dtype(scale * 2^(n-1)));
 *   q = clamp(roundeven(dtype(dtype(x - min_s) / scale)), 0, 2^n - 1);  dequant = dtype(dtype((q - 2^(n-1)) * scale) + zero).
 * q image: n = 8 -> int8 [H, S, D] holding (uint8)q;  n = 4 / 2 -> uint8 [H*S*D*n/8], 8/n consecutive values of the
 * flattened [H, S, D] tensor per byte, value j shifted left by j*n bits (D % (8/n) == 0).
 * cc_kv_requant: work <- dequant(quant(work)) in place and the q image / scales / zeros are emitted — the
 *   reference's quantize_cache() at the end of one update followed by dequantize_cache() at the start of the next.
 * cc_kv_dequant: work_out <- dequant(q image) (loading a quantised state).
 * ---------------------------------------------------------------------------------------------- */
int cc_kv_requant(void* work, void* q_out, void* scales, void* zeros, int32_t H, int32_t S, int32_t D,
                  int32_t dtype, int32_t n_bit, cc_stream_t stream);
/* K and V in one launch (what KVCache.quantize_cache does every step).
 * Optional exact skipping (stable != NULL): the round trip of a slot is a pure function of that slot's rows, so once a
 * pass left them bit-identical (stable[kv, s] = 1) and no insert touched the slot since (pos[:, s] still equals
 * pos_seen[kv, :, s]; an insert always writes a new position), repeating it would reproduce the same rows, image,
 * scale and zero point — the slot is skipped.  stable: uint8 [2, S], pos_seen: int32 [2, Hp, S], both zero-initialised
 * (and zeroed again whenever the working caches are written by anything but the cache's own inserts); pos: [Hp, S]. */
int cc_kv_requant_pair(void* k_work, void* k_q, void* k_scales, void* k_zeros, void* v_work, void* v_q, void* v_scales,
                       void* v_zeros, int32_t H, int32_t S, int32_t D, int32_t dtype, int32_t n_bit, const int32_t* pos,
                       int32_t Hp, uint8_t* stable, int32_t* pos_seen, cc_stream_t stream);
/* The same round trip for SEVERAL caches in ONE launch: what a model runs at the end of a token for all its layers (the
 * reference does it per layer inside update_kv, cache.py:323-338; the round trip of layer l is only needed before layer l's
 * NEXT step, so one launch behind the last layer replaces one launch per layer).  table (device): n_caches x 16 int64 —
 * k_work, k_q, k_scales, k_zeros, v_work, v_q, v_scales, v_zeros, pos, stable, pos_seen (pointers as for the pair call; stable
 * and pos_seen are required here), S, Hp, 0, 0, 0.  All caches share H, D, dtype, n_bit; S may differ (S_max = the largest).
 * Slot for slot the arithmetic of the pair call.  CC_ERR_UNSUPPORTED where the pair call would take its element-wise form
 * (D not a multiple of the 16-byte vector, or H * D / vector > 1024). */
int cc_kv_requant_batch(const int64_t* table, int32_t n_caches, int32_t H, int32_t S_max, int32_t D, int32_t dtype, int32_t n_bit,
                        cc_stream_t stream);
int cc_kv_dequant(const void* q, const void* scales, const void* zeros, void* work_out, int32_t H, int32_t S,
                  int32_t D, int32_t dtype, int32_t n_bit, cc_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * FUSED quantised KV cache — opt-in (`cache_quant_mode="fused"`), a DIFFERENT numerical contract from the reference's
 * ("fuse dequant into the attention read instead of whole-cache round trips").
 * The reference (cache.py:283-338, quantization_utils.py:4-45) keeps ONE (scale, zero) per slot shared by all heads and
 * re-quantises the whole cache every step; its numbers are reproduced by cc_kv_requant above.  Here instead:
 *   - one (scale, minimum) pair per (head, slot) ROW of K and one per row of V, float32, fixed when the row is written:
 *       mn, mx = min / max over the row's D values;  range = max(mx - mn, 1e-6f);  scale = range / 255.f;
 *       q[d]   = (uint8) clamp(rintf((x[d] - mn) * (255.f / range)), 0, 255)          (IEEE fp32 ops, round-half-even)
 *       value  = T(fmaf((float)q[d], scale, mn))                                      (one rounding to the model dtype)
 *   - a row is quantised exactly once (at prefill or when the decode step inserts it) and never re-quantised: no drift;
 *     the inserted token is attended to through its image in the step that inserts it (the reference attends to the
 *     unquantised new token once and to its round trip afterwards);
 *   - the decode kernels stream the uint8 images (half the bytes of a 16-bit cache) and dequantise in registers on the way
 *     to the matrix cores; everything else of the step (scores -> dtype, softmax, P.V, history, next-eviction key) is the
 *     fused step of cc_decode_step_heavy_hitter / _recent_global / _random applied to the dequantised values.
 * qparams: float32 [H, S, 4] = (k_scale, k_min, v_scale, v_min) per (head, slot).  n_bit: 8 (others: CC_ERR_UNSUPPORTED).
 * cc_kv_quant_rows / cc_kv_dequant_rows convert whole caches ([H, S, D] model dtype <-> uint8 images + qparams).
 * cc_decode_step_quant: c->k_cache / c->v_cache are the uint8 IMAGES [H, S, D], c->dtype the MODEL dtype (of q, k_new,
 *   v_new, y, attn_out); 16-bit dtype, D == 128, HQ / H in {4, 8} (CC_ERR_UNSUPPORTED otherwise).
 *   policy: 1 = heavy hitter (num / denom / counter as in cc_decode_step_heavy_hitter, c->Hp == H),
 *           2 = recent_global / full (num = denom = NULL, c->Hp == 1), 3 = random (rand_next as in cc_decode_step_random).
 *   next_key comes from the policy's own *_next_key_init (it reads positions and history only, never K / V).
 *   phases: as cc_decode_step_heavy_hitter_phases (3 = the whole step; CC_PHASE_* select the launch form);
 *   cc_decode_step_quant_single_launch: 1 when the single-launch form is available for the shape on this device.
 * ---------------------------------------------------------------------------------------------- */
int cc_kv_quant_rows(const void* k, const void* v, int32_t H, int32_t S, int32_t D, int32_t dtype, int32_t n_bit,
                     uint8_t* k_q, uint8_t* v_q, float* qparams, cc_stream_t stream);
int cc_kv_dequant_rows(const uint8_t* k_q, const uint8_t* v_q, const float* qparams, int32_t H, int32_t S, int32_t D,
                       int32_t dtype, int32_t n_bit, void* k_out, void* v_out, cc_stream_t stream);
int cc_decode_step_quant(const cc_kv_view* c, float* qparams, int32_t n_bit, int32_t policy, const void* q,
                         const void* k_new, const void* v_new, const int32_t* input_pos, double* num, int32_t* denom,
                         int64_t* counter, const float* rand_next, uint64_t* next_key, int32_t global_tokens,
                         int32_t recent_window, int32_t HQ, float scale, void* y, void* attn_out, void* workspace,
                         size_t workspace_bytes, cc_stream_t stream, int32_t phases);
/* ... with the recoverable hand-off and, for policy 3, the in-kernel uniform draws (r4): rand_next may be NULL — the draws are then
 * the stateless hash of (seed, position, slot) that cc_decode_step_random_rng uses (the pipeline must have been seeded with
 * cc_random_next_key_init_rng on the same seed).  step_commit as in cc_decode_step_heavy_hitter_rc (may be NULL). */
int cc_decode_step_quant_rc(const cc_kv_view* c, float* qparams, int32_t n_bit, int32_t policy, const void* q, const void* k_new,
                            const void* v_new, const int32_t* input_pos, double* num, int32_t* denom, int64_t* counter,
                            const float* rand_next, uint64_t seed, uint64_t* next_key, int32_t* step_commit, int32_t global_tokens,
                            int32_t recent_window, int32_t HQ, float scale, void* y, void* workspace, size_t workspace_bytes,
                            cc_stream_t stream, int32_t phases);
int32_t cc_decode_step_quant_single_launch(int32_t HQ, int32_t H, int32_t S, int32_t D, int32_t dtype, int32_t n_bit);
/* ... and for the l2 step (cc_decode_step_l2[_rc]); cc_decode_step_single_launch_enabled: 1 while the process-wide switch
 * (cc_decode_step_set_single_launch, include/coldcompress_debug.h) AND the current device's own (cc_decode_step_device_single_launch,
 * below) allow the single-launch forms at all.  Together with the
 * per-kind queries they tell a caller which FORM a step call will take: only the single-launch forms read the status / commit words
 * of the recoverable hand-off (the two-launch and three-call forms have nothing that can time out and ignore them). */
int32_t cc_decode_step_l2_single_launch(int32_t HQ, int32_t H, int32_t S, int32_t D, int32_t dtype);
int32_t cc_decode_step_single_launch_enabled(void);
/* r6 — the knob for SHARED devices (per device, like the co-residency it is about).  The single-launch forms need all
 * workgroups of a launch resident together: on a GPU that this process shares with other processes or concurrent kernels (several TP
 * ranks on one GPU, a co-tenant job) that does not hold, and every hand-off would end in the bounded wait.  enabled == 0: steps
 * launched on the CURRENT device take the two-launch forms from now on (nothing in them can time out; the cache state they leave is
 * bit-identical, so the switch may be flipped between any two steps); != 0: the single-launch forms again.  Other devices of the
 * process are untouched; cc_decode_step_single_launch_enabled and the per-kind queries answer for the current device.  -> the previous
 * value.  (A step captured into a hipGraph keeps the form it was captured with.)  ref: the reference has no such form to switch
 * (cache.py:314-364 runs as eager ops). */
int32_t cc_decode_step_device_single_launch(int32_t enabled);

/* ------------------------------------------------------------------------------------------------
 * Prefill-time cache fill.  ref: KVCache._prefill_update / _fill_contiguous cache.py:381-401.
 *   slots 0..T-1 of every head <- k_val/v_val rows; pos[hp, t] <- pos_val[min(hp,PH-1), t] (int32 cast);
 *   mask[h, t] <- 1; cache_cts[j] += T.
 * k_val, v_val: [H, T, D] contiguous; pos_val: int64 [PH, T] with PH == 1 or PH == Hp.
 * ---------------------------------------------------------------------------------------------- */
int cc_prefill_fill(const cc_kv_view* c, const void* k_val, const void* v_val, const int64_t* pos_val,
                    int32_t PH, int32_t T, cc_stream_t stream);

/* Row L2 norms.  ref: KVCacheL2.update_state cache.py:607-612; PromptCompressorL2 prompt_compression.py:201-203.
 * x: [H, N, D] dtype -> out[H, N] dtype = dtype(sqrt(sum_d x^2)) (fp32 accumulate); negate != 0 -> -norm. */
int cc_row_l2_norm(const void* x, int32_t H, int32_t N, int32_t D, int32_t dtype, int32_t negate,
                   void* out, cc_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Prompt compaction.  ref: PromptCompressor._keep_idxs prompt_compression.py:21-26 and
 * _filter_kv :69-72 / :82-88.
 * cc_topk_keep: keep[hs, 0..K) = ascending-sorted indices of the K largest priorities of row hs.
 *   Ties at the K-th value are broken lowest-index-first (torch's CPU order there is
 *   implementation-defined; see DESIGN.md "top-k tie contract").  NaN ranks above +inf (torch.topk).
 *   priority: [Hs, L] of prio_dtype (CC_PRIO_*).  keep_out: int64 [Hs, K].
 * cc_gather_rows: dst[h, j, :] = src[h, keep[min(h,Hk-1), j], :]; src [H, L, D], dst [H, K, D].
 * ---------------------------------------------------------------------------------------------- */
size_t cc_topk_keep_workspace_bytes(int32_t Hs, int32_t L, int32_t K);
int cc_topk_keep(const void* priority, int32_t prio_dtype, int32_t Hs, int32_t L, int32_t K,
                 int64_t* keep_out, void* workspace, size_t workspace_bytes, cc_stream_t stream);
int cc_gather_rows(const void* src, const int64_t* keep, int32_t Hk, int32_t H, int32_t L, int32_t K,
                   int32_t D, int32_t dtype, void* dst, cc_stream_t stream);
/* dst[hs, j] = src[hs, keep[hs, j]] for a [Hs, L] vector of `dtype` (cumulative-attention gather,
 * ref: PromptCompressorHeavyHitter._update_state prompt_compression.py:189-194). */
int cc_gather_vec(const void* src, const int64_t* keep, int32_t Hs, int32_t L, int32_t K, int32_t dtype,
                  void* dst, cc_stream_t stream);

/* KVCacheAnalysis (`debug_<strategy>`) decode-time bookkeeping in one launch.  ref: cache.py:1391-1404.
 *   attn: [>= Hp, S_full] dtype — the group-mean attention row over the FULL cache (row h = kv head h; a head-constant shadow
 *         cache reads row 0 only, like the reference's gather with a [1, 1, S] index);
 *   pos:  [Hp, S] int32 — the shadow cache's positions, -1 = unfilled (reads the last column, which is zero);
 *   sub_out: [Hp, S] dtype = attn[h, pos[h, s]] — what the shadow cache's update_state receives;
 *   losses[*loss_ctr] = dtype(mean_h dtype(1 - dtype(sum_s sub_out[h, s]))), then *loss_ctr += 1 (device scalars; cap = length of
 *   losses; an index beyond it is counted but not stored).  Hp <= 64. */
int cc_analysis_loss(const void* attn, const int32_t* pos, int32_t Hp, int32_t S_full, int32_t S, int32_t dtype, void* sub_out,
                     void* losses, int32_t* loss_ctr, int32_t cap, cc_stream_t stream);

/* SnapKV priority.  ref: PromptCompressorHeavyHitter._token_importances prompt_compression.py:170-187.
 *   obs_mean: [H, L] dtype = mean over the last min(16,L) query rows of the group-averaged probabilities.
 *   out[h,t] = dtype(avgpool5(obs_mean)[h,t]) (pad 2, count_include_pad=False);
 *   t >= L-obs_len -> 1.0; t < global_tokens -> 1.0. */
int cc_snapkv_priority(const void* obs_mean, int32_t H, int32_t L, int32_t dtype, int32_t obs_len,
                       int32_t global_tokens, void* out, cc_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Prefill attention with the side outputs the eviction policies need, without materialising [HQ,L,L].
 * ref: attention_utils.py:36-54 (naive path, causal mask) + model.py:413-418 (group mean) +
 *      cache.py:704 / prompt_compression.py:170-173,189-192 (column mean, observation-window mean).
 *   q: [HQ, L, D]; k, v: [H, L, D]; causal.  y: [HQ, L, D] dtype.
 *   colsum_out [H, L] f32 (optional): sum over queries of the group-averaged dtype-rounded probabilities.
 *   obs_out    [H, L] f32 (optional): mean over the last obs_len queries of the same.
 * ---------------------------------------------------------------------------------------------- */
size_t cc_prefill_attn_workspace_bytes(int32_t HQ, int32_t H, int32_t L, int32_t D, int32_t dtype);
int cc_prefill_attn(const void* q, const void* k, const void* v, int32_t HQ, int32_t H, int32_t L,
                    int32_t D, int32_t dtype, float scale, void* y, float* colsum_out, float* obs_out,
                    int32_t obs_len, void* workspace, size_t workspace_bytes, cc_stream_t stream);

/* The same with band-sum side outputs for the hybrid profiling score: band_out[b, h, k] = sum over queries
 * q in [k, k + bands[b]) of the group-averaged dtype-rounded probabilities (cf. cc_attn_bandsum).
 * bands: HOST int32[n_bands], n_bands <= 4; band_out: device f32 [n_bands, H, L]. */
int cc_prefill_attn_bands(const void* q, const void* k, const void* v, int32_t HQ, int32_t H, int32_t L, int32_t D,
                          int32_t dtype, float scale, void* y, float* colsum_out, float* obs_out, int32_t obs_len,
                          const int32_t* bands, int32_t n_bands, float* band_out, void* workspace,
                          size_t workspace_bytes, cc_stream_t stream);

/* Column sums of a materialised attention tensor (API-compat path when a caller hands the reference's
 * own [H, L, L] probabilities).  ref: cache.py:704.  out[h,k] = sum_q attn[h,q,k] (fp32 accumulate). */
int cc_attn_colsum(const void* attn, int32_t H, int32_t Lq, int32_t Lk, int32_t dtype, float* out,
                   cc_stream_t stream);

/* Column mean used by the heavy-hitter prefill state and the SnapKV state.
 * ref: cache.py:704 and prompt_compression.py:191: `attn.sum(dim=2) / (seq_len - input_pos)` on a
 * model-dtype tensor: out[h,t] = dtype( dtype(colsum[h,t]) / (float)(L - input_pos[t]) ).
 * input_pos: int64 [L] or NULL (= arange(L), what prefill always passes, generation_utils.py:462). */
int cc_colsum_to_mean(const float* colsum, const int64_t* input_pos, int32_t H, int32_t L, int32_t dtype,
                      void* out, cc_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Hybrid (FastGen) cache and the W-slot attention-history ring.  ref: KVCacheHybrid cache.py:768-1288.
 * ---------------------------------------------------------------------------------------------- */

/* Decode-time update for a cache whose heads follow different policies.
 * ref: _decoding_update cache.py:965-1019, _select_fill_idx :896-950, _eviction_idx_for_head :844-894.
 *   strategies: int64 [H], policy index of each head (cache_strategies).
 *   policy_table: int32 [n_policies, 3] on the device = {flags, window slots = round(recent_window*S),
 *     heavy-hitter slots = round(heavy_hitter_frac*S)}; flags: 1 heavy_hitter, 2 window, 4 punc, 8 special, 16 full.
 *   num: [H, S, W] model dtype history ring; denom: [H, S] int32.  The heavy-hitter score of a slot is
 *     dtype(sum_W num) / min(denom, W) over the first cache_cts[h] slots; protected slots (first g slots,
 *     special/punctuation slots, pos > p - window) -> +inf; arg-min.
 *   is_punc: device bool[1] (torch.isin(input_ids, punc_ids), cache.py:975) or NULL; with is_punc == NULL and
 *     token_id (device int64[1]) + punc_ids (device int64[n_punc_ids]) given, the same membership test is evaluated
 *     inside the launch.  num_special / num_punc: device int32[1] or NULL.
 *   c must be head-specific and variable-length (Hp == Hc == H).  fill_out: int64 [H] (dropped tokens report S-1).
 * Effects: pos/K/V written at fill_out for every head; mask and cache_cts only on appends; history zeroed on
 * evictions when requires_heavy_hitter; punc_mask set and num_punc += 1 when is_punc. */
int cc_hybrid_decode_update(const cc_kv_view* c, const void* k_new, const void* v_new, const int32_t* input_pos,
                            const int64_t* strategies, const int32_t* policy_table, int32_t n_policies, void* num,
                            int32_t* denom, int32_t W, const uint8_t* special_mask, uint8_t* punc_mask,
                            const uint8_t* is_punc, const int64_t* token_id, const int64_t* punc_ids,
                            int32_t n_punc_ids, const int32_t* num_special, int32_t* num_punc,
                            int32_t global_tokens, int32_t requires_heavy_hitter, int64_t* fill_out,
                            float* wsum_workspace, uint64_t* wsum_acc, cc_stream_t stream);

/* Heavy-hitter decode update with a finite history window (history_window_size W > 1; model-dtype ring).
 * ref: KVCacheHeavyHitter._eviction_idx cache.py:725-765:
 *   avg = float(dtype(sum_W num)) / clamp(denom, 1, W); (pos<g)|(pos>=p-w) -> 1.0; pos==-1 -> 0.0; arg-min;
 *   ring row and denom of the chosen slot zeroed; then the common insert (k_new == NULL: select only).
 * num: [H, S, W] model dtype; denom: [H, S] int32; Hp must be H.
 *
 * Window sums (both ring entry points).  dtype(sum_W num) is DEFINED as the exact sum of the W ring entries rounded
 * once, nearest-even, to the model dtype (torch's own fp32 summation order is unspecified); being exact it does not
 * depend on summation order, so it can be produced two ways with identical results:
 *   wsum_acc == NULL (stateless): wsum_workspace, float [H*S], is caller scratch; a chip-wide pre-pass (one wave per
 *     slot, coalesced) re-reads the whole [H, S, W] ring on every call — 118 MB at H = 8, S = 18432, W = 400.
 *   wsum_acc != NULL (tracked): wsum_workspace / wsum_acc are PERSISTENT state kept current by cc_hh_ring_update
 *     (the overwritten ring entry leaves the sum, the new one enters) and by the evictions here (row zeroed -> sum
 *     zeroed); no pre-pass.  wsum_acc: uint64 [cc_hh_ring_acc_words(H, S, W, dtype)] = per slot a 192-bit
 *     two's-complement fixed-point accumulator in units of 2^-149 plus a count of entries with |v| >= 4 or non-finite
 *     (their window sum is NaN; attention probabilities are <= 1); then two launch-ticket words; then a column-major
 *     [W][H*S] shadow of the ring (the entry a step overwrites is read from there, coalesced); then 2H words where the
 *     workgroups sharing a head meet (zero between launches).  Zero-initialise it
 *     together with the ring, or (re)build it from an existing ring with cc_hh_ring_window_sums. */
int cc_decode_update_heavy_hitter_ring(const cc_kv_view* c, const void* k_new, const void* v_new,
                                       const int32_t* input_pos, void* num, int32_t* denom, int32_t W,
                                       int32_t global_tokens, int32_t recent_window, int64_t* idx_out,
                                       float* wsum_workspace, uint64_t* wsum_acc, cc_stream_t stream);

/* History ring update, history_window_size W > 1.  ref: cache.py:716-723:
 *   num[h, s, *counter % W] = attn[h, s] (0 beyond T); denom += 1 everywhere; *counter += 1.
 * num: [H, S, W] dtype; attn: [H, T] dtype.  wsum_acc / wsum: both NULL, or the tracked window-sum state (above). */
int cc_hh_ring_update(void* num, int32_t* denom, int64_t* counter, const void* attn, int32_t H, int32_t S, int32_t T,
                      int32_t W, int32_t dtype, uint64_t* wsum_acc, float* wsum, cc_stream_t stream);
size_t cc_hh_ring_acc_words(int32_t H, int32_t S, int32_t W, int32_t dtype);
/* Exact window sums of a whole ring: wsum float [H*S]; wsum_acc (optional) the tracked state rebuilt from the ring. */
int cc_hh_ring_window_sums(const void* num, int32_t H, int32_t S, int32_t W, int32_t dtype, float* wsum,
                           uint64_t* wsum_acc, cc_stream_t stream);

/* Band sums of a materialised attention tensor: out[h,k] = sum_{q=k}^{k+band-1} attn[h,q,k] (fp32, sequential).
 * Used by the hybrid profiling score for window policies (ref: create_window_attention_mask cache.py:142-149 +
 * profile_attn_heads :1165-1168 rewritten per key instead of per [L, L] mask). */
int cc_attn_bandsum(const void* attn, int32_t H, int32_t Lq, int32_t Lk, int32_t dtype, int32_t band, float* out,
                    cc_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Caller glue around the hot path (ref: model.py Attention.forward / TransformerBlock / RMSNorm /
 * apply_rotary_emb / FeedForward).  The reference leaves these to ~45 eager elementwise launches per layer
 * (or to torch.compile); here they are three fused launches so the decode step is GEMV-bound.
 * ---------------------------------------------------------------------------------------------- */

/* (optional residual add) + RMSNorm.  ref: model.py:452-457 and :325-326.
 *   h = delta ? x + delta : x   (model dtype add, written to h_out if non-NULL; h_out may alias x)
 *   out = dtype( dtype( float(h) * rsqrt(mean(float(h)^2) + eps) ) * weight )
 * x, delta, h_out, out: [T, dim] dtype; weight: [dim] dtype. */
int cc_add_rmsnorm(const void* x, const void* delta, const void* weight, int32_t T, int32_t dim, float eps,
                   int32_t dtype, void* h_out, void* out, cc_stream_t stream);

/* Split the fused QKV projection, apply rotary embedding to q and k, and lay the heads out for attention.
 * ref: model.py:375-387 + apply_rotary_emb :507-519 (adjacent pairs, fp32 math, table stored in model dtype).
 *   qkv: [T, (HQ + 2H) * D] dtype (q block, k block, v block); freqs: [T, D/2, 2] dtype = rows of the
 *   (cos, sin) table already gathered at the token positions.
 *   q_out: [HQ, T, D]; k_out, v_out: [H, T, D]  (== the transposes model.py:385-387 produces). */
int cc_qkv_rope(const void* qkv, const void* freqs, int32_t T, int32_t HQ, int32_t H, int32_t D, int32_t dtype,
                void* q_out, void* k_out, void* v_out, cc_stream_t stream);

/* SwiGLU gate: out = dtype( dtype(silu(a)) * b ).  ref: model.py:442-443 F.silu(w1 x) * w3 x.  a, b, out: [n]. */
int cc_silu_mul(const void* a, const void* b, int64_t n, int32_t dtype, void* out, cc_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Decode-time dense layer (one token) with the caller glue fused in.  ref (caller side): model.py:317-327 (pre-norm
 * block + residual adds), :375-387 (qkv split + apply_rotary_emb), :442-443 (FFN), :452-457 (RMSNorm), :507-519 (RoPE).
 *   prologue  (norm_w != NULL): h = dtype(x + delta) (delta may be NULL), h_out <- h (optional, the updated residual
 *             stream), in = dtype(dtype(h * rsqrt(mean(h^2) + eps)) * norm_w);  else in = x.
 *   product   t[n] = dtype(sum_k W[n,k] * in[k] (+ bias[n])), fp32 accumulation; W: [N, K] row-major (nn.Linear.weight).
 *   W3 != NULL: y[n] = dtype(dtype(silu(t[n])) * dtype(sum_k W3[n,k] * in[k]))           (SwiGLU: w1, w3 in one pass)
 *   freqs != NULL: rows [0, rope_rows) are rotated in (even, odd) pairs with freqs[(row % head_dim)/2] = (cos, sin)
 *             of the current position, fp32 math on the dtype-rounded t (q and k heads of a fused wqkv); rows beyond
 *             rope_rows (v) are copied.
 * K * sizeof(dtype) <= 64 KiB (16 input chunks per lane).  Summation order inside a dot product is the
 * kernel's own (tolerance class, like any GEMM library).
 * ---------------------------------------------------------------------------------------------- */
int cc_gemv_fused(const void* W, const void* W3, const void* x, const void* delta, const void* norm_w, float eps,
                  void* h_out, const void* bias, const void* freqs, int32_t rope_rows, int32_t head_dim, void* y,
                  int32_t N, int32_t K, int32_t dtype, cc_stream_t stream);

/* Greedy sampling tail, ref: generation_utils.py:136-142: probs[V] = dtype(softmax_fp32(logits[V])),
 * *idx_out = first index of the largest rounded probability (torch.argmax semantics).  Two small launches; `workspace`
 * (cc_softmax_argmax_workspace_bytes) needs no initialisation. */
size_t cc_softmax_argmax_workspace_bytes(void);
int cc_softmax_argmax(const void* logits, int32_t V, int32_t dtype, void* probs, int32_t* idx_out, void* workspace,
                      size_t workspace_bytes, cc_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * One-shot sum all-reduce over the GPUs of ONE node, for the decode-size messages of tensor parallelism.
 * ref: tp.py:134-138, 156-160 (`all_reduce(sum)` of the wo and FFN outputs: 2 * dim bytes, 8-16 KiB, twice per layer).
 * Every rank stores its vector straight into a slot of every peer's buffer over the xGMI mesh, raises a flag, waits for the
 * flags raised for it, and adds the `world` slots in RANK ORDER (fp32 accumulate, one rounding): one launch of one workgroup,
 * no ring hops, capturable in a hipGraph, bitwise-identical results on every rank.  RCCL (torch.distributed "nccl") remains
 * the general collective and the oracle this is tested against (tests/test_gpu_tp.py).
 *   cc_allreduce_create   allocates this rank's symmetric buffer (uncached fine-grained device memory; the ONE allocating,
 *                         synchronising entry point of this ABI: call it at start-up, not on the decode path);
 *   cc_allreduce_export   copies the buffer's hipIpcMemHandle_t (cc_allreduce_handle_bytes() bytes) into HOST memory;
 *   cc_allreduce_connect  takes the world's handles in rank order (HOST memory; the host exchanges them, e.g. with
 *                         torch.distributed.all_gather) and maps the peers' buffers;
 *   cc_allreduce_sum      in place on `data` (device, 16-byte aligned, n elements of `dtype`, n * size <= max_bytes); every
 *                         rank must call it in the same order on its stream; never synchronises;
 *   cc_allreduce_status   0, or 1 if an all-reduce gave up waiting for a peer (bounded polling) — synchronising read.
 * ---------------------------------------------------------------------------------------------- */
typedef struct cc_comm cc_comm;
size_t cc_allreduce_handle_bytes(void);
int cc_allreduce_create(int32_t rank, int32_t world, size_t max_bytes, cc_comm** out);
int cc_allreduce_export(cc_comm* comm, void* handle_out);
int cc_allreduce_connect(cc_comm* comm, const void* handles);
int cc_allreduce_sum(cc_comm* comm, void* data, int64_t n, int32_t dtype, cc_stream_t stream);
int32_t cc_allreduce_status(cc_comm* comm);
int cc_allreduce_destroy(cc_comm* comm);

#ifdef __cplusplus
}
#endif
#endif /* COLDCOMPRESS_H */
