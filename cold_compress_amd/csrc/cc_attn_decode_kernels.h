// cc_attn_decode_kernels.h — the streaming-pass / single-launch layer-step kernel templates of cc_attn_decode.hip, in a header so that
// a second translation unit (cc_attn_decode_qkv.hip: the step fused with the layer's QKV projection) can instantiate them too.
// Everything lives in an anonymous namespace: each translation unit gets its own copy of what it instantiates.
#pragma once
#include "cc_common.h"
#include "cc_wacc.h"
#include "cc_gemv_core.h"

// ---- A/B switches (tools/ab_variant.sh NAME "-DCC_V_...=0"): the defaults are what the product runs.  Round 4 measured each on one
//      box against its absence (profiles/r04_ab_step_variants.md); the switches that lost (early partial-O polls, one key per
//      workgroup, precomputed merge factors, bulk requests moved up, longer sleeps) are gone from the source (commit a08483f has them).
#ifndef CC_V_PRO
#define CC_V_PRO 1      // prologue diet: unconditional key-row loads, K rows requested ahead of the mask word
#endif
#ifndef CC_V_LDSDMA
#define CC_V_LDSDMA 1   // K / V tiles land in the wave's LDS slabs directly (buffer_load ... lds): no staging registers, no ds_write
#endif
#ifndef CC_V_EMLMT
#define CC_V_EMLMT 0    // 1: the several-tiles-per-wave steps take the early-(m, l) order too — MEASURED A LOSS (r4, one box: S = 8192 12.3 ->
                        // 12.8 us, 18432 21.0 -> 21.75, 32768 32.8 -> 33.9): two gathers in series and one more barrier, and per-slot passes
                        // of 1.5+ us that a 0.4 us partial-O round trip cannot hide; kept compilable for the record
#endif
#ifndef CC_V_ALLLANES
#define CC_V_ALLLANES 1 // the several-tiles-per-wave steps run the per-slot state pass on all 64 lanes (like the hybrid tail), not on 16 per tile
#endif
#ifndef CC_V_WORDSFIRST
#define CC_V_WORDSFIRST 1  // the step's words (epoch, position, status, commit words) are requested at the top of the kernel
#endif
#ifndef CC_V_KPIN
#define CC_V_KPIN 1  // the prologue's kernel arguments in one round of scalar loads, BEHIND the request of the step's words (0: left to the compiler, three rounds)
#endif
#ifndef CC_QKV_TILE_AT
#define CC_QKV_TILE_AT 2  // QKV step: where the K / V tile is requested — 0: ahead of the weights (20.0 us per layer at C3), 1: behind the first
                          // unit's request (20.5), 2: behind the LAST unit's request (19.1; the product) — A/B, r5, same box
#endif
#ifndef CC_QKV_DEPTH
#define CC_QKV_DEPTH 2  // QKV step: 4-row weight units (8 loads per lane each) in flight per wave: 2 (18.9 us per layer at C3) or 3 (19.6) — A/B, r5
#endif
#ifndef CC_QKV_TRACE
#define CC_QKV_TRACE 0  // 1 (measurement builds of cc_attn_decode_qkv.hip only): thread 0 of every workgroup stamps the phases of the QKV step
#endif
#ifndef CC_V_PRELOAD
#define CC_V_PRELOAD 0  // r6 A/B (VERDICT r5 #1b, "the prologue"): 1 = the operands of the step's FIRST requests (its words, the key row, the K
                        // rows) are leading scalar kernel arguments, preloaded into SGPRs at wave launch (-mllvm
                        // -amdgpu-kernarg-preload-count=14; tools/probes/preload_probe shows the firmware honours it, also under graph
                        // replay): no kernel-argument round trip in front of the first K request.  MEASURED (profiles/r06_ab_step_variants.md):
                        // with LATE + EARLYARGS below the first K request leaves 0.46-0.59 us after a workgroup's first instruction
                        // instead of 0.85 — and the K rows LAND when they always did (2.9 us at C3, 1.6-1.7 at one kv head): the step is
                        // 0.05-0.1 us SLOWER at C3 / C2, 0.05 faster at C5's rank.  What bounds the first byte is not when it is asked
                        // for (tools/probes/first_byte_probe).  Off: the r5 prologue is the product's.
#endif
#ifndef CC_V_LATE
#define CC_V_LATE 1     // r6 A/B: with the preloaded arguments, the LDS-DMA steps read the rest of their argument block BEHIND the first K request
                        // (0: left to the compiler, which puts scalar waits in front of the K request in 37 of 47 instantiations)
#endif
#ifndef CC_V_EARLYARGS
#define CC_V_EARLYARGS 1  // r6 (LATE steps): the rest of the argument block is REQUESTED at the kernel's first instruction — scalar loads spelled
                          // in assembly, which the compiler's wait-count pass does not see — and waited for once, behind the K request:
                          // the round trip runs in the shadow of the prologue instead of behind it (0: requested behind the K request)
#endif
#ifndef CC_V_ORDER
#define CC_V_ORDER 0    // r6 A/B (with PRELOAD + LATE + EARLYARGS, plain 16-bit steps): 1 = the step's SMALL requests first — words, key row, per-slot
                        // state, q — and the bulk (K rows, mask word, V rows) behind them, all within the first ~150 instructions: nothing
                        // the scores need queues behind 8 MB of rows, and the V rows leave ~0.4 us earlier than in the product's order
#endif
#ifndef CC_V_VEARLY
#define CC_V_VEARLY 0   // r6 A/B: the V rows requested right behind the K rows (1: the steps without the XL2 placement — few-head ranks, whose
                        // caches are latency-bound, not bandwidth-bound; 2: every LDS-DMA step).  At C3 this order lost 0.4 us (r4).
#endif
#ifndef CC_V_FEWXCD
#define CC_V_FEWXCD 0   // r6 A/B: ranks with 1, 2 or 4 kv heads take the XL2 instantiations on a grid of 8 VIRTUAL heads — block b works for kv head
                        // b % 8 when that is < H and exits otherwise: a head's workgroups share ONE XCD, its hand-off stays in that L2
#endif
#if CC_V_FEWXCD && !CC_V_PRELOAD
#error "the virtual-head placement takes its head counts from the preloaded arguments"
#endif
#ifndef CC_V_L2CARRY
#define CC_V_L2CARRY 0  // r6: the l2 single-launch step carries each head's norm maximum across steps (a resolved RECORD in the
                        // key row's tail: cc_common.h, cc_l2_record) — no reduction over the head's norms and NO cross-head hand-off inside
                        // the launch; see L2C in the kernel.  0 = the r4 / r5 exchange (one level through memory, or two with XL2).
#endif
#ifndef CC_V_FLATLOADS
#define CC_V_FLATLOADS 0  // r6: in the LDS-DMA steps every small load between the key row and the V rows (mask word, per-slot state, q) is issued
                          // by ALL lanes, unconditionally (lanes without a row / slot / head aim at a valid dummy or past a buffer's end):
                          // behind exec-mask branches the compiler's wait-count pass could not count them, its wait for the KEY ROW became
                          // vmcnt(8) — a wait for the K tile — and the key row's reduction, the insert decision and the commit words
                          // (~100 instructions) ran on every wave's critical path BEHIND the K rows instead of in the two microseconds the
                          // wave waits for them (found in the ISA while measuring the l2 cuts).  MEASURED A LOSS (same box, three rounds:
                          // C3 8.38 -> 8.51, C2 7.53 -> 7.84, C5's rank 7.68 -> 7.87, recent_global 8.05 -> 8.23): the wait becomes exact
                          // (vmcnt(16)) and the reduction does move up — and the V rows, no longer held back by five branches, leave
                          // earlier and compete with the K rows, the order r4 measured at +0.1..0.4 us.  0 = the r5 form (the product).
#endif
#ifndef CC_V_WFACT
#define CC_V_WFACT 0    // r6: the early-(m, l) steps keep the waves' merge factors exp(m_w - M) — computed once per (wave, head) by the lane that
                        // merges the workgroup's (m, l) pair, between the scores and the P.V products — in LDS; the partial-O publish behind
                        // the merge barrier reads them instead of recomputing NW maxima, subtractions and exponentials per thread (~50 of its
                        // ~110 instructions, on every workgroup's path between the merge barrier and its partial O).  Same function, same
                        // operands: the same bits.  0 = recompute (r5)
#endif
#ifndef CC_V_NW16
#define CC_V_NW16 0     // r6 A/B: cc_decode_step_set_wide(2) plans ONE 16-wave workgroup per 256 rows (n_split 16 at S = 4096: half the partials of
                        // the hand-off again — the move from 4 to 8 waves gained 0.4 us at C3 — on half the CUs); heavy hitter / head-constant
                        // policies, 4 query heads per kv head, single launch only, timing experiments only (no two-launch twin of this geometry)
#endif
#ifndef CC_V_VDELAY
#define CC_V_VDELAY 0   // r6 A/B: s_sleep CC_V_VDELAY (x 64 cycles) in front of the V rows' request in the LDS-DMA steps — the opposite of
                        // CC_V_VEARLY / CC_V_FLATLOADS, which both lost by letting the V rows compete with the K rows earlier
#endif
#ifndef CC_V_MLW
#define CC_V_MLW 1      // the final (M, L) fold runs on the workgroup's LAST waves (idle during the partial-O publish of the first ones)
#endif

namespace {

// ---------------------------------------------------------------- cross-lane all-reduce helpers
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}

// sum over an aligned group of W lanes (W power of two <= 64); every lane of the group gets the total
template <int W>
__device__ __forceinline__ float group_sum(float v) {
  if (W >= 2) v += dpp_mov<0xB1>(v);    // quad_perm [1,0,3,2]  : lane ^ 1
  if (W >= 4) v += dpp_mov<0x4E>(v);    // quad_perm [2,3,0,1]  : lane ^ 2
  if (W >= 8) v += dpp_mov<0x141>(v);   // row_half_mirror      : pairs the two quads of each 8
  if (W >= 16) v += dpp_mov<0x140>(v);  // row_mirror           : pairs the two halves of each 16
  if (W >= 32) v += __shfl_xor(v, 16, CC_WAVE);
  if (W >= 64) v += __shfl_xor(v, 32, CC_WAVE);
  return v;
}
// Whole-wave reductions with a wave-UNIFORM (scalar) result: four DPP steps fold each row of 16 lanes, four
// v_readlane pick up the row results.  No LDS crossbar (ds_bpermute) round trips: ~12 VALU ops instead of 6
// dependent ~100-cycle shuffles.  Fixed combination order -> deterministic.
__device__ __forceinline__ float wave_max_uniform(float v) {
  v = fmaxf(v, dpp_mov<0xB1>(v));
  v = fmaxf(v, dpp_mov<0x4E>(v));
  v = fmaxf(v, dpp_mov<0x141>(v));
  v = fmaxf(v, dpp_mov<0x140>(v));
  const int u = __builtin_bit_cast(int, v);
  const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(u, 0));
  const float r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(u, 16));
  const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(u, 32));
  const float r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(u, 48));
  return fmaxf(fmaxf(r0, r1), fmaxf(r2, r3));
}
__device__ __forceinline__ float wave_sum_uniform(float v) {
  v += dpp_mov<0xB1>(v);
  v += dpp_mov<0x4E>(v);
  v += dpp_mov<0x141>(v);
  v += dpp_mov<0x140>(v);
  const int u = __builtin_bit_cast(int, v);
  const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(u, 0));
  const float r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(u, 16));
  const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(u, 32));
  const float r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(u, 48));
  return (r0 + r1) + (r2 + r3);
}
// N (4 or 8) consecutive floats of a 16-byte aligned LDS row, read in assembly with their own wait: the compiler guards any LDS
// read it sees behind pending LDS-DMA loads with a wait for those (it cannot tell the row from the slabs they write).
template <int N>
__device__ __forceinline__ void lds_read_row_nowait(const float* row, float (&out)[N]) {
  static_assert(N == 4 || N == 8, "one or two 16-byte reads");
  typedef float f32x4_t __attribute__((ext_vector_type(4)));
  const unsigned addr = (unsigned)(uintptr_t)(__attribute__((address_space(3))) const float*)row;
  f32x4_t a, b = {0.f, 0.f, 0.f, 0.f};
  if constexpr (N == 8) {
    asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:16\n\ts_waitcnt lgkmcnt(0)" : "=&v"(a), "=&v"(b) : "v"(addr) : "memory");
  } else {
    asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(a) : "v"(addr) : "memory");
  }
#pragma unroll
  for (int i = 0; i < 4; i++) out[i] = a[i];
  if constexpr (N == 8) {
#pragma unroll
    for (int i = 0; i < 4; i++) out[4 + i] = b[i];
  }
}
template <int CTRL>
__device__ __forceinline__ unsigned long long dpp_min_u64(unsigned long long v) {
  const int lo = __builtin_amdgcn_update_dpp(0, (int)(unsigned)(v & 0xffffffffull), CTRL, 0xf, 0xf, true);
  const int hi = __builtin_amdgcn_update_dpp(0, (int)(unsigned)(v >> 32), CTRL, 0xf, 0xf, true);
  const unsigned long long o = ((unsigned long long)(unsigned)hi << 32) | (unsigned)lo;
  return o < v ? o : v;
}
__device__ __forceinline__ unsigned long long wave_min_u64_uniform(unsigned long long v) {
  v = dpp_min_u64<0xB1>(v);
  v = dpp_min_u64<0x4E>(v);
  v = dpp_min_u64<0x141>(v);
  v = dpp_min_u64<0x140>(v);
  const int lo = (int)(unsigned)(v & 0xffffffffull), hi = (int)(unsigned)(v >> 32);
  unsigned long long best = ~0ull;
#pragma unroll
  for (int row = 0; row < 4; row++) {
    const unsigned long long x = ((unsigned long long)(unsigned)__builtin_amdgcn_readlane(hi, row * 16) << 32) |
                                 (unsigned)__builtin_amdgcn_readlane(lo, row * 16);
    best = x < best ? x : best;
  }
  return best;
}
// value held by lane ^ OFF for OFF in {16, 32}: one v_permlane{16,32}_swap (VALU; no LDS crossbar traffic).
// swap(x, x) returns {rows a|a, rows b|b}: whichever differs from ours is the partner's value — but we only ever
// need sum or max with the partner, both symmetric, so combine the two returned halves directly.
template <int OFF, bool IS_MAX>
__device__ __forceinline__ float xor_combine(float v) {
  const unsigned u = __builtin_bit_cast(unsigned, v);
  float a, b;
  if (OFF == 16) {
    auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    a = __builtin_bit_cast(float, (unsigned)r[0]);
    b = __builtin_bit_cast(float, (unsigned)r[1]);
  } else {
    auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    a = __builtin_bit_cast(float, (unsigned)r[0]);
    b = __builtin_bit_cast(float, (unsigned)r[1]);
  }
  return IS_MAX ? fmaxf(a, b) : a + b;
}
// sum over an aligned group of G lanes, G a WAVE-UNIFORM power of two <= 64: every lane of the group gets the total (xor
// butterflies: DPP inside a row of 16, permlane swaps across rows — no LDS crossbar).  Fixed order -> deterministic.
__device__ __forceinline__ float seg_sum(float v, int G) {
  if (G >= 2) v += dpp_mov<0xB1>(v);
  if (G >= 4) v += dpp_mov<0x4E>(v);
  if (G >= 8) v += dpp_mov<0x141>(v);
  if (G >= 16) v += dpp_mov<0x140>(v);
  if (G >= 32) v = xor_combine<16, false>(v);
  if (G >= 64) v = xor_combine<32, false>(v);
  return v;
}
__device__ __forceinline__ float fast_exp(float x) {  // e^x via v_exp_f32 (2^x); exp(-inf) = 0
  return __builtin_amdgcn_exp2f(x * 1.4426950408889634f);
}
// e^x for x <= 0 to about one ulp in seven instructions (the library expf is ~40 with a branch, and it sat on the
// latency-critical tail of the step): x * log2(e) = t + r with t = fl(x * C_hi) and the residual r recovered by two fmas
// (log2(e) = C_hi + C_lo), then 2^(t + r) = 2^t * (1 + r ln 2 + O(r^2)), |r| < 2^-18.  Used for the probabilities that
// feed the float64 history — by the combine pass and the single-launch tail alike, so the two stay bit-identical.
__device__ __forceinline__ float exp_nonpos(float x) {
  x = fmaxf(x, -200.f);  // e^-200 = 0 in fp32; keeps -inf away from the residual (inf - inf)
  const float t = x * 1.44269502162933349609f;
  float r = fmaf(x, 1.44269502162933349609f, -t);
  r = fmaf(x, 1.92596299112661746e-08f, r);
  const float e = __builtin_amdgcn_exp2f(t);
  return fmaf(e, r * 0.6931471805599453f, e);
}

// KVCacheHybrid (FastGen) in the fused two-launch step: what the per-head decision of cache.py:896-950 needs.  The
// eviction CANDIDATE of a head (arg-min over its live slots, protections applied) is scored by the combine pass of the
// previous step; whether the head appends, evicts that candidate or drops the token is decided here, at the top of the
// streaming pass, from the head's policy, its count, the budget terms and the incoming token's punctuation flag.
enum { HF_HH = 1, HF_WIN = 2, HF_PUNC = 4, HF_SPECIAL = 8, HF_FULL = 16 };
struct HybridStep {
  const int64_t* strategies;    // [H] policy index per head
  const int32_t* table;         // [n_pol, 3]: flags, window slots, heavy-hitter slots
  const uint8_t* special_mask;  // [H, S] or null
  uint8_t* punc_mask;           // [H, S] or null
  const int64_t* token_id;      // device int64[1] or null
  const int64_t* punc_ids;      // [n_punc_ids] or null
  int n_punc_ids;
  const int32_t* num_special;   // device int[1] or null
  int32_t* num_punc;            // device int[1] or null
  int32_t* cts_next;            // [H] workspace: the head's count after this step's insert (committed by the combine pass)
  int W;                        // history window of the ring (clamp of the denominator)
  int n_pol;                    // rows of `table` (<= 21: the streaming pass fetches the whole table with one vector load)
};

// QKV instantiations (r5): the layer's QKV projection (RMSNorm prologue, bias, RoPE epilogue — cc_gemv.hip's arithmetic) folded into the
// single-launch step.  ref (caller side): model.py:375-387 (wqkv, split, apply_rotary_emb), :452-457 (RMSNorm).
struct QkvIn {
  const void* W;        // [(HQ + 2H) * D, K] row major: q heads, then k heads, then v heads (model.py:335-338)
  const void* bias;     // [(HQ + 2H) * D] or null
  const void* x;        // [K] residual stream
  const void* delta;    // [K] pending residual (added first), or null
  const void* norm_w;   // [K] RMSNorm weight
  const void* freqs;    // [D / 2, 2] (cos, sin) of this position, model dtype
  void* h_out;          // [K] x + delta, or null
  void* qkv_out;        // [(HQ + 2H) * D] plain copy of the projection, or null (nothing in the step reads it)
  void* gran;           // per-kv-head granule regions (kOneQHead bytes each): the projection's hand-off between the head's workgroups
  unsigned gran_bytes;
  float eps;
  int K;                // model dim
  int HQ;               // query heads of this rank (the k rows start at HQ * D)
  unsigned long long* trace;  // CC_QKV_TRACE builds: [workgroup][16] stamps of the 100 MHz device clock, or null
};

struct SplitArgs {
  const void* q;
  const void* k;
  const void* v;
  const uint8_t* mask;
  void* scores;    // [HQ, S] T
  float* part_ml;  // [HQ, n_split, 2]
  float* part_o;   // [HQ, n_split, D]
  int S, R, n_split, rows_per_split;
  float scale;
  int abl;  // measurement-only ablation bits (phases >> 8): 1 = no score store, 2 = no epilogue, 4 = no mask
  // ---- fused decode step (next_key != null): this step's insert is folded into the prologue.  The slot comes
  //      from the partial arg-min keys the PREVIOUS step's combine pass (or cc_hh_next_key_init) left in
  //      next_key[h][0..nk): their minimum is torch's arg-min.
  const unsigned long long* next_key;  // [H][nk]
  int nk;       // entries per key row (the row stride)
  int nk_read;  // entries any writer may have left non-~0: all of them where the single-launch step can serve the shape,
                // the first n_chunks otherwise (only the two-launch combine pass writes such rows)
  const int32_t* input_pos;
  const void* k_new;  // [H, D]
  const void* v_new;
  int32_t* pos;        // [H, S]
  uint8_t* mask_w;     // [H, S]
  int32_t* cache_cts;  // [Hc]
  double* num;         // [H, S]
  int32_t* denom;      // [H, S]
  int H, Hc, Hp;  // Hp == 1: head-constant policy (one pos row shared by every kv head; the key rows are per kv head all the same — see KEY ROWS)
  // ---- ring history folded into the combine pass (history_window_size W > 1): this launch publishes the ring column
  //      of the step, *ring_counter % W, so that the combine launch may bump the counter without a reader racing it
  const int64_t* ring_counter;
  int* ring_col;
  int ring_W;
  // ---- single-launch hybrid step (ONE + HYB): the tracked ring state the combine pass would have updated (null: no head
  //      runs a heavy-hitter policy); ring_col stays null, every workgroup derives the column from the counter itself
  void* ring_num;     // [H, S, W] T
  unsigned long long* ring_acc;  // tracked state (include/coldcompress.h): accumulators, tickets, column-major shadow
  float* ring_wsum;   // [H, S]
  // ---- l2 policy in the fused step (matrix-core kernel only): the inserted key's norm is recorded here, and every
  //      wave publishes the maximum of key_norm over its slots (the evicted slot's old norm excluded), so that the
  //      combine launch can form the global maximum of cache.py:602 without re-reading [H, S] norms per workgroup
  void* key_norm;   // [H, S] T
  float* l2_pmax;   // [H, n_split, NW]
  float* l2_new;    // [H]
  // ---- single-launch layer step (ONE): the combine pass folded into this launch.  Every workgroup publishes its
  //      partial as self-validating 16-byte granules {tag, x, tag, y} (write-through stores), waits until the
  //      n_split workgroups of its kv head have published, and then finishes ITS OWN 64 slots (probabilities, group
  //      mean, history, next-eviction key) from the scores still in its registers, plus its share of y.
  unsigned* one_hdr;  // [H] epoch words (tag = epoch + 1, bumped once per launch and head), then the spin-timeout word
  void* one_ml;       // [H][n_split][RT] granules {tag, m, tag, l}
  void* one_o;        // [H * RT][n_split][64] granules {tag, O[2p], tag, O[2p + 1]}
  unsigned one_ml_bytes, one_o_bytes;
  void* y;            // [HQ, D] T
  void* attn_out;     // [H, S] T or null
  int64_t* hh_counter;
  int g, w;           // global_tokens, recent_window of the next-eviction score
  int policy;         // 1 = heavy hitter (history in num / denom), 2 = recent_global / full, 3 = random (head-constant: no history)
  const float* rand_next;  // policy 3: [S] uniform draws for position p + 1, or null: cc_rng_uniform(rng_seed, p + 1, slot)
  unsigned long long rng_seed;
  int yc_chunks;      // grid.x of the two-launch combine pass (unused)
  unsigned long long* trace;  // measurement only (cc_decode_step_trace): [workgroup][16] time stamps and hardware ids
  // ---- recoverable hand-off (r3; the early-(m, l) single-launch steps): step_commit[h] = the last position whose step is fully
  //      committed for kv head h (-1 = none; null = the caller does not retry).  See "Recoverable hand-off" below.
  int32_t* commit;
  HybridStep hyb;     // HYB instantiation only
  // ---- fused quantised cache (QB instantiation): k / v point at the uint8 images [H, S, D]; one (scale, minimum) pair per
  //      (head, slot) row for K and one for V — dequantised in registers on the way to the LDS slabs
  float* qparams;     // [H, S, 4]: k_scale, k_min, v_scale, v_min
  QkvIn qkv;          // QKV instantiations only: q / k_new / v_new are null, the step computes them itself
  int virt8;          // XL2 instantiations (r6, CC_V_FEWXCD): the grid has 8 virtual kv heads, blocks of heads >= H exit (travels in the preloaded word)
};

// PRELOAD (r6): 14 dwords — all the hardware preloads — in front of the argument block: K base, S, rows per split, the key rows, their
// stride, (live entries | kv heads << 20), the workspace header, the position word, the commit words.  The kernel takes THESE instead of
// the block's copies (the launcher fills both alike: cc_lead_of).
#if CC_V_PRELOAD
#define CC_LEAD_PARAMS const void* pl_k, int pl_S, int pl_rps, const unsigned long long* pl_key, int pl_nk, int pl_misc, unsigned* pl_hdr, const int32_t* pl_pos, int32_t* pl_commit,
#define CC_LEAD_TYPES const void*, int, int, const unsigned long long*, int, int, unsigned*, const int32_t*, int32_t*,
#define CC_LEAD_ARGS(sa) (sa).k, (sa).S, (sa).rows_per_split, (sa).next_key, (sa).nk, (int)((unsigned)(sa).nk_read | ((unsigned)(sa).H << 20) | ((sa).virt8 ? 0x80000000u : 0u)), (sa).one_hdr, (sa).input_pos, (sa).commit,
constexpr int kLeadNkReadMax = (1 << 20) - 1, kLeadHMax = (1 << 11) - 1;
#else
#define CC_LEAD_PARAMS
#define CC_LEAD_TYPES
#define CC_LEAD_ARGS(sa)
#endif
constexpr int kLeadBytes = CC_V_PRELOAD ? 56 : 0;  // 14 dwords (pointers on 8-byte boundaries: see the parameter list)

template <typename T, int D, int RT, int NW, int U>
__global__ __launch_bounds__(NW * 64) void decode_attn_split_kernel(SplitArgs a) {
  if (a.ring_col && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x == 0)
    *a.ring_col = (int)(*a.ring_counter % a.ring_W);
  constexpr int VEC = 16 / (int)sizeof(T);
  constexpr int LPR = D / VEC;
  static_assert(LPR >= 1 && LPR <= 64 && (LPR & (LPR - 1)) == 0, "head_dim must map to a power-of-two lane count");
  constexpr int RPW = 64 / LPR;  // rows per wave-wide load
  constexpr int NG = NW * RPW;  // row groups per workgroup
  __shared__ float sm_m[NG][RT];
  __shared__ float sm_l[NG][RT];
  __shared__ __attribute__((aligned(16))) float sm_acc[NG][RT][D];

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int lr = lane / LPR, lc = lane % LPR;
  const int split = blockIdx.x, h = blockIdx.y, q0 = h * a.R + blockIdx.z * RT;
  const int S = a.S;
  const int row_begin = split * a.rows_per_split;
  const int row_end = min(S, row_begin + a.rows_per_split);
  const T* kh = reinterpret_cast<const T*>(a.k) + (size_t)h * S * D + lc * VEC;
  const T* vh = reinterpret_cast<const T*>(a.v) + (size_t)h * S * D + lc * VEC;
  const bool has_mask = a.mask != nullptr && !(a.abl & 4);
  const uint8_t* mh = has_mask ? a.mask + (size_t)h * S : reinterpret_cast<const uint8_t*>(a.k);
  T* sc_out = reinterpret_cast<T*>(a.scores);

  float m[RT], l[RT], acc[RT][VEC];
#pragma unroll
  for (int r = 0; r < RT; r++) {
    m[r] = -INFINITY;
    l[r] = 0.f;
#pragma unroll
    for (int e = 0; e < VEC; e++) acc[r][e] = 0.f;
  }

  // ---- Issue EVERY load of the first tile before anything waits: the dependent chain of this kernel is
  //      kernel args -> {partial keys, q, mask, K, V all in flight} -> math, not args -> q -> K/V.  (An earlier
  //      version converted q to fp32 first, which put a full L2 round trip in front of the K/V loads: -1 us.)
  //      vmcnt retires in issue order, so the loads are issued in the order their results are consumed.
  // fused insert: every wave reduces the head's partial arg-min keys itself (no LDS, no barrier)
  int ins_idx = -1, ins_was_empty = 0;
  bool key_pending = a.next_key != nullptr && !(a.abl & 128);
  unsigned long long key_part = ~0ull;
  if (key_pending && lane < a.nk_read) key_part = a.next_key[(size_t)h * a.nk + lane];
  // q: [RT][D] of this query group, this lane's VEC-wide column slice (L2-resident after the first workgroups)
  Vec16<T> qraw[RT];
#pragma unroll
  for (int r = 0; r < RT; r++) qraw[r].load(reinterpret_cast<const T*>(a.q) + (size_t)(q0 + r) * D + lc * VEC);

  // Row group lr of wave `wave` owns the U CONSECUTIVE rows base + lr*U + [0, U): its mask bytes are one
  // aligned 32-bit word, its scores one contiguous run, and its softmax state (m, l, acc) is private to the
  // 16-lane group — no cross-group shuffles anywhere in the loop.
  static_assert(U <= 4, "mask bytes of a row group are packed into one 32-bit word");
  uint32_t mword = 0x01010101u;
  Vec16<T> kk[U], vv[U];
  auto issue_tile = [&](int base) {
    const int row0 = base + lr * U;
    mword = 0x01010101u;
    if (has_mask) {
      if (U == 4 && row0 + 3 < S && ((reinterpret_cast<uintptr_t>(mh) + (size_t)row0) & 3) == 0) {
        mword = *reinterpret_cast<const uint32_t*>(mh + row0);  // the common case: one aligned word
      } else {  // ragged tail / unaligned head offset: assemble from bytes
        mword = 0;
#pragma unroll
        for (int u = 0; u < U; u++)
          if (row0 + u < S) mword |= (uint32_t)mh[row0 + u] << (8 * u);
      }
    }
#pragma unroll
    for (int u = 0; u < U; u++) kk[u].load(kh + (size_t)(row0 + u < row_end ? row0 + u : row_end - 1) * D);
#pragma unroll
    for (int u = 0; u < U; u++) vv[u].load(vh + (size_t)(row0 + u < row_end ? row0 + u : row_end - 1) * D);
  };
  int base = row_begin + wave * (RPW * U);
  bool more = base < row_end;
  if (more) issue_tile(base);

  float qf[RT][VEC];
#pragma unroll
  for (int r = 0; r < RT; r++) qraw[r].unpack(qf[r]);

  while (more) {
    const int row0 = base + lr * U;
    // fused insert (cache.py:356-362, 460-490, 754-763): the row group that owns the chosen slot uses the new
    // token's K/V instead of the stale cache row, writes them back, and one lane does the bookkeeping.  The new
    // rows (and the position) are fetched inside this rare branch — one row group per kv head takes it — so
    // they cost nothing on the streaming path; their latency hides behind the K/V loads already in flight.
    if (key_pending) {  // wave-uniform; first iteration only
      for (int i = lane + 64; i < a.nk_read; i += 64) {  // caches beyond 64 chunks (S > 8192)
        const unsigned long long x = a.next_key[(size_t)h * a.nk + i];
        key_part = x < key_part ? x : key_part;
      }
      const unsigned long long key = wave_min_u64_uniform(key_part);
      ins_idx = (key == ~0ull) ? -1 : (int)((key & 0xffffffffull) >> 1);
      if (a.abl & 64) ins_idx = -1;
      ins_was_empty = (int)(key & 1ull);  // the winner's "slot was empty" bit travels with the key (cache.py:356-360)
      key_pending = false;
    }
    if ((unsigned)(ins_idx - row0) < (unsigned)U) {
      Vec16<T> kn, vn;
      kn.load(reinterpret_cast<const T*>(a.k_new) + (size_t)h * D + lc * VEC);
      vn.load(reinterpret_cast<const T*>(a.v_new) + (size_t)h * D + lc * VEC);
      const int32_t p_now = *a.input_pos;
#pragma unroll
      for (int u = 0; u < U; u++)
        if (row0 + u == ins_idx) {
          kk[u].raw = kn.raw;
          vv[u].raw = vn.raw;
          mword |= 1u << (8 * u);
        }
      if (blockIdx.z == 0) {
        const size_t slot = (size_t)h * S + ins_idx;
        *reinterpret_cast<uint4*>(const_cast<T*>(kh) + (size_t)ins_idx * D) = kn.raw;
        *reinterpret_cast<uint4*>(const_cast<T*>(vh) + (size_t)ins_idx * D) = vn.raw;
        if (lc == 0) {  // stores only: nothing here waits on memory
          if (a.Hp != 1 || h == 0) a.pos[(a.Hp == 1 ? 0 : (size_t)h * S) + ins_idx] = p_now;
          a.mask_w[slot] = 1;
          if (a.num != nullptr) {  // heavy hitter: cache.py:754-763
            a.num[slot] = 0.0;
            a.denom[slot] = 0;
          }
          if (ins_was_empty && (a.Hc == a.H || h == 0)) atomicAdd(&a.cache_cts[a.Hc == a.H ? h : 0], 1);
        }
      }
    }

    float s[RT][U];
#pragma unroll
    for (int u = 0; u < U; u++) {
      float kf[VEC];
      kk[u].unpack(kf);
      const bool valid = (row0 + u < row_end) && (((mword >> (8 * u)) & 0xffu) != 0);
#pragma unroll
      for (int r = 0; r < RT; r++) {
        float d = 0.f;
#pragma unroll
        for (int e = 0; e < VEC; e++) d = fmaf(qf[r][e], kf[e], d);
        d = group_sum<LPR>(d);
        // ref: attention_utils.py:37 (q@k^T -> dtype, * scale -> dtype), :42-43 (-inf bias where masked)
        const float x = ElemTraits<T>::rnd(ElemTraits<T>::rnd(d) * a.scale);
        s[r][u] = valid ? x : -INFINITY;
      }
    }
    // ---- scores out: lane lc of each row group stores pair t = pass*LPR + lc  (t = r*U + u): one store
    //      instruction per pass with every lane active, RPW*U contiguous elements per query head
#pragma unroll
    for (int pass = 0; pass * LPR < RT * U; pass++) {
      const int t = pass * LPR + lc;
      float val = 0.f;
#pragma unroll
      for (int r = 0; r < RT; r++)
#pragma unroll
        for (int u = 0; u < U; u++) val = (t == r * U + u) ? s[r][u] : val;
      const int r = t / U, u = t - r * U;
      const int row = row0 + u;
      if (t < RT * U && row < row_end && !(a.abl & 1)) ElemTraits<T>::store(sc_out, (size_t)(q0 + r) * S + row, val);
    }
    // ---- online softmax, state private to the row group
#pragma unroll
    for (int r = 0; r < RT; r++) {
      float mx = s[r][0];
#pragma unroll
      for (int u = 1; u < U; u++) mx = fmaxf(mx, s[r][u]);
      const float m_new = fmaxf(m[r], mx);
      const float m_use = (m_new == -INFINITY) ? 0.f : m_new;
      const float alpha = fast_exp(m[r] - m_use);
      l[r] *= alpha;
#pragma unroll
      for (int e = 0; e < VEC; e++) acc[r][e] *= alpha;
      m[r] = m_new;
#pragma unroll
      for (int u = 0; u < U; u++) s[r][u] = fast_exp(s[r][u] - m_use);
    }
#pragma unroll
    for (int u = 0; u < U; u++) {
      float vf[VEC];
      vv[u].unpack(vf);
#pragma unroll
      for (int r = 0; r < RT; r++) {
        const float p = s[r][u];
        l[r] += p;
#pragma unroll
        for (int e = 0; e < VEC; e++) acc[r][e] = fmaf(p, vf[e], acc[r][e]);
      }
    }
    base += NW * RPW * U;
    more = base < row_end;
    if (more) issue_tile(base);
  }

  if (a.abl & 2) {  // measurement only: keep the accumulators live, skip the merge
    float x = 0.f;
#pragma unroll
    for (int r = 0; r < RT; r++) {
      x += l[r] + m[r];
#pragma unroll
      for (int e = 0; e < VEC; e++) x += acc[r][e];
    }
    if (x == 1.2345f) a.part_ml[0] = x;
    return;
  }
  // ---- row-group partials -> LDS.  Every lane writes its own VEC-wide slice as 16-byte pieces; piece e4 of
  //      lane lc lands at column (e4/4)*(D/ (VEC/4)) ... i.e. [piece][lc][4]: consecutive lanes are 16 B apart,
  //      so the ds_write_b128 lane groups are bank-conflict free.  (Measured: merging the row groups in
  //      registers with v_permlane16/32_swap butterflies instead is SLOWER: 8.9 vs 7.75 us at S = 4096.)
  constexpr int NP = VEC / 4 > 0 ? VEC / 4 : 1;  // 16-byte pieces per lane
  const int grp = wave * RPW + lr;
  if (lc == 0) {
#pragma unroll
    for (int r = 0; r < RT; r++) {
      sm_m[grp][r] = m[r];
      sm_l[grp][r] = l[r];
    }
  }
#pragma unroll
  for (int r = 0; r < RT; r++)
#pragma unroll
    for (int pc = 0; pc < NP; pc++)
      *reinterpret_cast<float4*>(&sm_acc[grp][r][(pc * LPR + lc) * 4]) =
          make_float4(acc[r][pc * 4], acc[r][pc * 4 + 1], acc[r][pc * 4 + 2], acc[r][pc * 4 + 3]);
  __syncthreads();
  for (int t = threadIdx.x; t < RT * D; t += NW * 64) {
    const int r = t / D, d = t - r * D;
    // column d lives in lane d / VEC, piece (d % VEC) / 4, element d % 4
    const int dl = (((d % VEC) / 4) * LPR + d / VEC) * 4 + (d & 3);
    float M = -INFINITY;
#pragma unroll
    for (int g = 0; g < NG; g++) M = fmaxf(M, sm_m[g][r]);
    const float Mu = (M == -INFINITY) ? 0.f : M;
    float L = 0.f, O = 0.f;
#pragma unroll
    for (int g = 0; g < NG; g++) {  // fixed order: deterministic
      const float f = fast_exp(sm_m[g][r] - Mu);
      L = fmaf(sm_l[g][r], f, L);
      O = fmaf(sm_acc[g][r][dl], f, O);
    }
    const size_t pj = (size_t)(q0 + r) * a.n_split + split;
    a.part_o[pj * D + d] = O;
    if (d == 0) {
      a.part_ml[pj * 2 + 0] = M;
      a.part_ml[pj * 2 + 1] = L;
    }
  }
}

// ================================================================================================================
// Matrix-core variant of the streaming pass for 16-bit caches with head_dim 128 (Llama-3 / Qwen-2 class models).
//
// The VALU kernel above spends ~560 vector instructions per 16-row tile, ~40% of them on q.k (fma + bf16 unpack +
// 16-lane DPP reductions) — with two waves per SIMD that is ~1.9 us of issue time at S = 4096, the same order as
// the 3.3 us it takes HBM to deliver the tile.  Here q.k runs as FOUR v_mfma_f32_16x16x32 per tile:
//   * global loads stay exactly as coalesced as before (16 lanes x 16 B = one whole 256-byte K row), but lane c of
//     the row with tile-local index i fetches 16-byte chunk (c ^ i) of that row instead of chunk c;
//   * every lane drops its chunk into its own wave's 4 KiB LDS slab at [i][c] (ds_write_b128, conflict free);
//   * the A operand of step j is read back as [i = lane % 16][(4j + lane / 16) ^ i] — the XOR applied at LOAD time
//     makes the 16 rows read by a quarter-wave land in 16 different bank groups (conflict free, no padding);
//   * B = q^T (query head n of the group in column n, zero for n >= RT) lives in 16 VGPRs for the whole kernel;
//   * C[i][n]: lane (g = lane / 16, n = lane % 16) gets the scores of ITS OWN row group's four rows against head n,
//     so the mask word, the online-softmax state (one m, l per lane instead of RT) and the 8-byte score store all
//     stay in the lane that already owns them; the four row groups of a wave share ONE running maximum per head
//     (two v_permlane swaps per tile);
//   * P.V runs on the matrix cores too, as O^T = V^T . P^T with v_mfma_f32_16x16x16: the QK output layout IS the
//     B operand (four probabilities per lane, converted to 16 bit), and the A operand comes from the V tile staged
//     row-major in a second wave-private LDS slab (coalesced ds_write_b128, chunks XOR-swizzled by 2*(row & 7)) and
//     read back with the gfx950 LDS transpose read ds_read_b64_tr_b16 — lane i of a 16-lane group supplies the
//     address of 4 columns of row i/4 and receives the 4 rows of column i (layout verified on the device with
//     tools/probes/tr_read_probe.hip).  The accumulators then cover all 16 rows of the tile: one partial per WAVE,
//     nothing to merge inside a wave, every accumulator of a lane belongs to one head (lane-local rescale).
// ~120 vector instructions + 12 MFMAs per tile instead of ~560 vector instructions.  No workgroup barrier in the
// loop: both LDS slabs are wave-private.
typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
template <typename T>
struct Mfma16x16x32;
template <>
struct Mfma16x16x32<bf16_t> {
  __device__ static __forceinline__ f32x4_t mma(uint4 a, uint4 b, f32x4_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
  }
};
template <>
struct Mfma16x16x32<f16_t> {
  __device__ static __forceinline__ f32x4_t mma(uint4 a, uint4 b, f32x4_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
  }
};
typedef short s16x4_t __attribute__((ext_vector_type(4)));
typedef _Float16 f16x4_t __attribute__((ext_vector_type(4)));
// P.V step: v_mfma_f32_16x16x16 (K = 16 cache rows); A = V^T fragment from the LDS transpose read, B = P^T (the QK
// MFMA's own output layout, converted to 16 bit)
template <typename T>
struct Mfma16x16x16;
template <>
struct Mfma16x16x16<bf16_t> {
  __device__ static __forceinline__ f32x4_t mma(s16x4_t a, s16x4_t b, f32x4_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, b, c, 0, 0, 0);
  }
  __device__ static __forceinline__ s16x4_t pack(const float* p) {
    s16x4_t r;
#pragma unroll
    for (int i = 0; i < 4; i++) r[i] = (short)f32_to_bf16_bits(p[i]);
    return r;
  }
};
template <>
struct Mfma16x16x16<f16_t> {
  __device__ static __forceinline__ f32x4_t mma(s16x4_t a, s16x4_t b, f32x4_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x16f16(__builtin_bit_cast(f16x4_t, a), __builtin_bit_cast(f16x4_t, b), c, 0, 0, 0);
  }
  __device__ static __forceinline__ s16x4_t pack(const float* p) {
    s16x4_t r;
#pragma unroll
    for (int i = 0; i < 4; i++) r[i] = (short)f32_to_f16_bits(p[i]);
    return r;
  }
};

typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
// Bounded waits: a launch that is not fully resident gives up instead of hanging.  r6: the bound is DEVICE TIME — kOneWaitTicks of
// s_memrealtime, the 100 MHz clock all XCDs share: 30 ms — instead of 2^18 poll rounds (1.6 s with memory polls, measured; shorter
// through an L2).  The clock is read once per 256 unsuccessful rounds (a scalar memory instruction with a wait of its own: nothing a
// wait that succeeds within microseconds ever executes), first to take the start, then to compare; 2^22 rounds stay as a cap.
constexpr unsigned kOneSpinMax = 1u << 22;
constexpr unsigned long long kOneWaitTicks = 3000000ull;
struct WaitBound {
  unsigned long long t0 = 0;
  __device__ __forceinline__ bool expired(unsigned spins) {
    if (spins > kOneSpinMax) return true;
    if ((spins & 255u) != 255u) return false;
    const unsigned long long now = __builtin_amdgcn_s_memrealtime();
    if (t0 == 0) {
      t0 = now | 1ull;
      return false;
    }
    return now - t0 > kOneWaitTicks;
  }
};
constexpr int kOneStatusWordDev = 1023;     // hdr[0 .. H): per-head epochs; hdr[1023]: timeout word (== kOneStatusWord)
constexpr int kOneTicketWord = 1022;        // hybrid: heads whose workgroups have all published (the last one commits the per-step scalars)
// Recoverable hand-off (the early-(m, l) steps).  A launch whose workgroups are not all resident cannot complete its hand-off:
// the waits are bounded, and what a timed-out workgroup leaves behind must neither corrupt state nor be half a step.
//   * hdr[kOneFailWord + h]: set to the launch's tag by any workgroup of kv head h that gives up.  Every workgroup reads it
//     with each round of its LAST gather and commits nothing when it carries its tag: a head's step is committed by all of its
//     workgroups or by none (the per-slot history, the next-eviction keys and y are stored only behind that check; a workgroup
//     that arrives late finds the word set because the one that gave up wrote it a whole streaming phase earlier).
//   * hdr[kOneStatusWordDev] != 0 (set with the fail word; cleared by the host): every later launch returns at once — nothing is
//     built on the garbage a failed step's y became.  The host clears it AND advances every epoch word (so that no granule of the
//     failed attempt can carry a later launch's tag), then simply runs the token again:
//   * step_commit (r4: per WORKGROUP, kRcStride int32 per kv head): words [2 + split] = the last position whose step THIS workgroup has
//     committed (its slots' history, its keys; split 0 also the head's count and the step counter); words [0], [1] = the slot this
//     position's insert went to ((slot << 1) | was_empty) and that position, written by whoever decides the insert — BEFORE anything
//     of the step can be committed (a workgroup commits behind the partial-O gather, i.e. after the inserting workgroup published).
//     A retry of position p: a workgroup whose word says p recomputes (same scores, same partials: the hand-off needs them) and
//     stores nothing; the others step; all of them take the insert slot from words [0], [1] when they carry p (committed
//     workgroups have overwritten their part of the key row with the NEXT position's keys: its minimum is no longer this step's).
//     Whatever the interleaving of give-ups and commits inside the failed launch — r3 left a window: a workgroup that completed its
//     gather in the round trip in which a sibling gave up committed alone, and the retry added its probabilities twice — every slot's
//     history is updated exactly once per position: the retry is idempotent per workgroup.
//   * hybrid (late r4; its tail keeps the memory order — (m, l) and O travel together, ONE gather): the same words, the insert word
//     = (slot << 2) | kind (append / evict / drop), plus [66] the head's count before the step and [67] the step's ring column:
//     on a retry the whole per-head decision comes from the record (committed workgroups have rewritten their keys, the head's
//     first workgroup may have committed the count, the LAST head to complete the step counter).  A workgroup that gives up in
//     its gather, or reads the head's fail word with it, stores nothing of the step — no y, no ring column, no keys.
constexpr int kOneFailWord = 64;
constexpr int kRcStride = 68;                  // step_commit: int32 per kv head — [0] insert word, [1] its position, [2 + split] committed position,
                                               // hybrid: [66] the head's count before the step, [67] the step's ring column
// Granule regions are PER KV HEAD at fixed strides, whatever the shape: a location is only ever written by launches of its own
// head, with tags from that head's epoch word — strictly growing per location even when caches of different head counts and
// lengths share the workspace (shape-dependent offsets let a stale granule of head 4 sit where head 1 of another shape expects
// its own, and the two heads' epochs need not be equal once launches with fewer heads have run).
constexpr int kOneMlHead = 64 * 8 * 16;        // (m, l): 64 splits x up to 8 query heads x 16 B
constexpr int kOneOHead = 8 * 64 * 64 * 16;    // O: up to 8 query heads x 64 splits x 64 pairs x 16 B
constexpr int kOneNmHead = 64 * 32;            // l2: one norm-maximum granule per split (r4 / r5 exchange), or two record granules per split (r6, L2C)
constexpr int kOneHmBytes = 32 * 32 * 16;      // l2, two-level exchange (r4): one norm-maximum granule per kv HEAD, behind the per-split regions (the first
                                               // 32 granules; the rest of the 16 KiB was the [dest head][src head] matrix of r6's first carried-record
                                               // cut — in the history — and is unused now)
constexpr int kOneMaxHeads = 32;
constexpr int kOneQHead = (8 + 2) * 128 / 4 * 16;  // QKV: up to 8 query heads + k + v of 128 values, four 16-bit values per 8-byte granule half
// the workspace's single-launch regions, at fixed offsets behind the 4 KiB header (cc_attn_decode.hip: kOneBytes)
constexpr size_t kOneHdrBytes = 4096, kOneMlCap = (size_t)kOneMaxHeads * (kOneMlHead + kOneNmHead) + kOneHmBytes,
                 kOneOCap = (size_t)kOneMaxHeads * kOneOHead;
constexpr int kOneAuxCoherent = 17;         // sc0 sc1: write-through stores / loads that bypass the non-coherent L1 (and stale L2 lines)

// ---- fused quantised cache: value = T(fma(q, scale, min)), one rounding; q in [0, 255]
template <typename T>
__device__ __forceinline__ uint32_t pack16x2(float lo, float hi) {
  if constexpr (ElemTraits<T>::code == CC_DT_BF16) {
    float ra, rb;
    return bf16_round_pair(lo, hi, ra, rb);
  } else {
    return (uint32_t)f32_to_f16_bits(lo) | ((uint32_t)f32_to_f16_bits(hi) << 16);
  }
}
template <typename T>
__device__ __forceinline__ uint4 dequant8(uint2 raw, float2 par) {
  const uint32_t w[2] = {raw.x, raw.y};
  uint32_t o[4];
#pragma unroll
  for (int i = 0; i < 2; i++) {
    float f0 = __builtin_fmaf((float)(w[i] & 0xffu), par.x, par.y);
    float f1 = __builtin_fmaf((float)((w[i] >> 8) & 0xffu), par.x, par.y);
    float f2 = __builtin_fmaf((float)((w[i] >> 16) & 0xffu), par.x, par.y);
    float f3 = __builtin_fmaf((float)(w[i] >> 24), par.x, par.y);
    if constexpr (ElemTraits<T>::code == CC_DT_F16) {  // fp32 first, then f16 (see cc_opaque_f32)
      f0 = cc_opaque_f32(f0); f1 = cc_opaque_f32(f1); f2 = cc_opaque_f32(f2); f3 = cc_opaque_f32(f3);
    }
    o[2 * i] = pack16x2<T>(f0, f1);
    o[2 * i + 1] = pack16x2<T>(f2, f3);
  }
  return make_uint4(o[0], o[1], o[2], o[3]);
}
// Quantise the 8 values of this lane on the grid of its 16-lane row group (one cache row of 128 values):
// min / max over the row, range = max(max - min, 1e-6), scale = range / 255, q = clamp(rint((x - min) * (255 / range)), 0, 255)
// (IEEE fp32 ops, no contraction)
template <typename T>
__device__ __forceinline__ uint2 quant8_row16(uint4 raw, float2& par) {
  Vec16<T> v;
  v.raw = raw;
  float x[8];
  v.unpack(x);
  float mn = x[0], mx = x[0];
#pragma unroll
  for (int i = 1; i < 8; i++) {
    mn = fminf(mn, x[i]);
    mx = fmaxf(mx, x[i]);
  }
#pragma unroll
  for (int off = 8; off > 0; off >>= 1) {
    mn = fminf(mn, __shfl_xor(mn, off, 16));
    mx = fmaxf(mx, __shfl_xor(mx, off, 16));
  }
  const float range = fmaxf(__fsub_rn(mx, mn), 1e-6f);
  const float sc = __fdiv_rn(range, 255.f), inv = __fdiv_rn(255.f, range);  // two divides per ROW, none per element
  uint32_t b[8];
#pragma unroll
  for (int i = 0; i < 8; i++) b[i] = (uint32_t)fminf(fmaxf(rintf(__fmul_rn(__fsub_rn(x[i], mn), inv)), 0.f), 255.f);
  par = make_float2(sc, mn);
  return make_uint2(b[0] | (b[1] << 8) | (b[2] << 16) | (b[3] << 24), b[4] | (b[5] << 8) | (b[6] << 16) | (b[7] << 24));
}

template <int V>
struct IntC {
  static constexpr int value = V;
};

// L2: the l2 policy's norm bookkeeping (its own instantiation: the others pay nothing).  ONE: the single-launch layer
// step (heavy hitter, recent_global / full, random; with L2: l2; with HYB: the FastGen hybrid cache) — needs R == RT, at most
// 64 workgroups per kv head, every workgroup of the grid co-resident.  HYB: the per-head decision of KVCacheHybrid at the top of
// the pass (two-launch form: candidates, ring and counts follow in the combine pass; with ONE: in the tail, on all lanes).
// QB = 8: the fused quantised cache (uint8 images + per-row (scale, minimum)), dequantised on the way to the LDS slabs.
// NSUB = 2 (multi-tile splits only): two tiles per wave and iteration, each with its own staging registers — the loads of a
// tile go out two half-iterations ahead of their use instead of one (twice the bytes in flight per wave).
// NT (ONE only): tiles per wave the single-launch step keeps scores for — 1 for caches up to 64 x 64 slots per kv head (the
// specialised form), 4 or 8 for longer ones (the wave loops over its tiles like the two-launch streaming pass, and its finish
// loops over them for the per-slot pass).
// FULL (ONE only; the two-launch pass always has them): the measurement hooks — time stamps (cc_decode_step_trace), ablation
// bits of the phases word — and the optional group-mean output attn_out are in the code.  The LEAN instantiations (FULL = false)
// are what the product runs: ~30 never-taken branches and their live ranges less is 0.25-0.5 us of the step (A/B on one box:
// heavy hitter 10.65 -> 10.38 us, the fused uint8 step 10.35 -> 9.85); a call that wants a stamp, an ablation bit or attn_out is
// routed to a FULL instantiation (bf16, four query heads per kv head) or to the two-launch step.
// QKV (r5; ONE, single tile, LDS-DMA tiles, lean): the layer's QKV projection rides the launch.  Every workgroup requests the weight
// rows of ITS share of its kv head's (RT + 2) * 128 projection rows AHEAD of its K / V tile — the tile needs no q — computes them with
// cc_gemv.hip's arithmetic (bit-identical q / k / v), publishes them as tagged granules to the head's other workgroups (the same
// transports as the step's own hand-off) and gathers the head's q, k_new, v_new into LDS; the step proper starts from there.  The
// tile's 16.8 MB stream in the shadow of the 50 MB of weights: one launch boundary, one prologue and one first-byte latency per
// attention sub-block instead of two.  One workgroup per CU (the LDS decides that anyway): 256 registers per lane.
template <typename T, int RT, int NW, bool L2, bool ONE = false, bool HYB = false, int QB = 0, int NSUB = 1, int NT = 1, bool FULL = !ONE, bool XL2 = false, bool QKV = false>
__global__ __launch_bounds__(NW * 64, NW == 16 ? 4 : (QKV ? 1 : ((ONE || QB) ? 2 : 1))) void decode_attn_split_mfma_kernel(CC_LEAD_PARAMS SplitArgs a_in) {
  // FULL (measurement instantiation): the workgroup's first instruction, on both clocks — the phases of cc_decode_step_trace count
  // from HERE (late r4; they used to count from behind the issue of the first K rows, ~0.8 us later)
  unsigned long long tr_entry = 0, rt_entry = 0;
  if constexpr (FULL) {
    tr_entry = __builtin_amdgcn_s_memtime();
    rt_entry = __builtin_amdgcn_s_memrealtime();  // 100 MHz, one clock for the whole device (s_memtime is per XCD)
  }
  // LATE (r6, with the preloaded arguments; the LDS-DMA steps): the argument block is read in TWO PHASES.  Phase 0 — entry to the
  // first K request — runs on the 14 preloaded dwords alone: the step's words, the key row and the K rows are requested without one
  // kernel-argument load having been issued, so nothing the compiler does with scalar registers can put a wait in front of them
  // (left to the compiler, 37 of 47 LDS-DMA instantiations had an `s_waitcnt lgkmcnt(0)` there: a scalar register reused while an
  // argument load into it was pending — tools/isa_first_request_audit.py).  Phase 1 — right behind the K request — fetches the rest
  // of the block in one round of scalar loads through a laundered kernarg pointer (late_phase below): one round trip, in the shadow
  // of the K rows.
  constexpr bool LATE = CC_V_PRELOAD != 0 && CC_V_LATE != 0 && CC_V_LDSDMA != 0 && ONE && NT == 1 && QB == 0 && !HYB && !QKV;
  constexpr bool SMALLFIRST = LATE && CC_V_EARLYARGS != 0 && CC_V_ORDER != 0 && !L2;
  SplitArgs a{};
  if constexpr (!LATE) a = a_in;
#if CC_V_PRELOAD
  // the preloaded copies REPLACE the argument block's (same values: CC_LEAD_ARGS): every use below reads an SGPR that was there
  // when the wave started, not a kernel-argument load
  auto apply_lead = [&]() {
    a.k = pl_k;
    a.S = pl_S;
    a.rows_per_split = pl_rps;
    a.next_key = pl_key;
    a.nk = pl_nk;
    a.nk_read = pl_misc & kLeadNkReadMax;
    a.one_hdr = pl_hdr;
    a.input_pos = pl_pos;
    a.commit = pl_commit;
  };
  apply_lead();
  const int lead_H = (int)(((unsigned)pl_misc >> 20) & (unsigned)kLeadHMax);
  // EARLY (LATE steps): phase 1's scalar loads go out HERE, as the kernel's first instructions, in assembly — the compiler does not
  // know that these registers are pending and therefore never waits for them (nor may it read them: nothing touches ea* until the
  // `s_waitcnt lgkmcnt(0)` of late_phase hands them over; tools/isa_first_request_audit.py checks that in the ISA).  Offsets: the
  // block starts kLeadBytes into the kernel-argument segment; the static_asserts pin the fields.
  typedef unsigned ea_x8 __attribute__((ext_vector_type(8)));
  typedef unsigned ea_x4 __attribute__((ext_vector_type(4)));
  typedef unsigned ea_x2 __attribute__((ext_vector_type(2)));
  constexpr bool EARLY = LATE && CC_V_EARLYARGS != 0;
  if constexpr (EARLY && FULL) asm volatile("" : "+s"(tr_entry), "+s"(rt_entry));  // (the entry stamps are scalar-memory results too: waited for HERE, ahead of the loads below)
  ea_x8 ea_q = {}, ea_kn = {}, ea_ct = {}, ea_ao = {};
  ea_x4 ea_r = {}, ea_rn = {};
  ea_x2 ea_hp = {}, ea_y = {}, ea_nrm = {}, ea_tr = {};
  unsigned ea_abl = 0;
  if constexpr (EARLY) {
    static_assert(offsetof(SplitArgs, q) == 0 && offsetof(SplitArgs, v) == 16 && offsetof(SplitArgs, mask) == 24, "ea_q");
    static_assert(offsetof(SplitArgs, R) == 60 && offsetof(SplitArgs, n_split) == 64 && offsetof(SplitArgs, scale) == 72 && offsetof(SplitArgs, abl) == 76, "ea_r");
    static_assert(offsetof(SplitArgs, k_new) == 104 && offsetof(SplitArgs, v_new) == 112 && offsetof(SplitArgs, pos) == 120 && offsetof(SplitArgs, mask_w) == 128, "ea_kn");
    static_assert(offsetof(SplitArgs, cache_cts) == 136 && offsetof(SplitArgs, num) == 144 && offsetof(SplitArgs, denom) == 152 && offsetof(SplitArgs, H) == 160 &&
                  offsetof(SplitArgs, Hc) == 164 && offsetof(SplitArgs, Hp) == 168, "ea_ct / ea_hp");
    static_assert(offsetof(SplitArgs, key_norm) == 224 && offsetof(SplitArgs, y) == 280, "ea_nrm / ea_y");
    static_assert(offsetof(SplitArgs, attn_out) == 288 && offsetof(SplitArgs, hh_counter) == 296 && offsetof(SplitArgs, g) == 304 && offsetof(SplitArgs, w) == 308 &&
                  offsetof(SplitArgs, policy) == 312, "ea_ao");
    static_assert(offsetof(SplitArgs, rand_next) == 320 && offsetof(SplitArgs, rng_seed) == 328 && offsetof(SplitArgs, trace) == 344, "ea_rn / ea_tr");
    static_assert(kLeadBytes == 56, "the immediates below are 56 + the field's offset");
    typedef const char __attribute__((address_space(4))) kernarg_b;
    kernarg_b* kp0 = (kernarg_b*)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile(
        "s_load_dwordx8 %0, %11, 0x38\n\t"    // q, k, v, mask
        "s_load_dwordx8 %1, %11, 0xa0\n\t"    // k_new, v_new, pos, mask_w
        "s_load_dwordx8 %2, %11, 0xc0\n\t"    // cache_cts, num, denom, H, Hc
        "s_load_dwordx2 %6, %11, 0xe0\n\t"    // Hp
        "s_load_dwordx4 %4, %11, 0x74\n\t"    // R, n_split, rows_per_split, scale
        "s_load_dwordx2 %7, %11, 0x150\n\t"   // y
        "s_load_dwordx8 %3, %11, 0x158\n\t"   // attn_out, hh_counter, g, w, policy
        "s_load_dwordx4 %5, %11, 0x178\n\t"   // rand_next, rng_seed
        "s_load_dwordx2 %8, %11, 0x118\n\t"   // key_norm
        "s_load_dwordx2 %9, %11, 0x190\n\t"   // trace
        "s_load_dword %10, %11, 0x84"           // abl
        : "=s"(ea_q), "=s"(ea_kn), "=s"(ea_ct), "=s"(ea_ao), "=s"(ea_r), "=s"(ea_rn), "=s"(ea_hp), "=s"(ea_y), "=s"(ea_nrm), "=s"(ea_tr), "=s"(ea_abl)
        : "s"(kp0));
  }
#else
  auto apply_lead = [&]() {};
#endif
  static_assert(!(HYB && L2), "the hybrid decision rides the plain streaming pass or the single-launch step");
  static_assert(QB == 0 || (QB == 8 && !L2 && !HYB), "fused quantised cache: 8 bits, heavy hitter / recent_global / random");
  static_assert(NSUB == 1 || (NSUB == 2 && !ONE), "two tiles per iteration: the two-launch streaming pass only");
  static_assert(NT == 1 || (ONE && !L2 && QB == 0 && NSUB == 1), "several tiles per wave in the single-launch step: 16-bit caches, heavy hitter / head-constant policies");
  constexpr bool ONE1 = ONE && NT == 1;  // the single-tile form: no loop tail, no rescale, per-slot state requested ahead of the tile
  // EML (r3): the workgroup's (m, l) pairs leave EARLY — right behind the scores, while the V rows are still in flight — so that
  // the final (M, L) of the head, the probabilities, the history update and the next-eviction keys run in the shadow of the
  // partial-O exchange; what is left behind the last O granule is the y fold alone
  // (l2 included since late r3: its norm maxima leave with the (m, l) pairs.  r4: the several-tiles-per-wave steps too — their pairs
  //  leave behind the LAST tile, ahead of the cross-wave merge: the finish's per-slot passes then run in the shadow of the partial-O
  //  exchange instead of behind it, the state stores sit behind the last gather, and the step is recoverable like the single-tile one.)
  constexpr bool EML = ONE && !HYB && (NT == 1 || CC_V_EMLMT != 0);
#ifdef CC_NO_RC  // (A/B builds)
  constexpr bool RC = false;
#else
  constexpr bool RC = EML;  // the recoverable hand-off (status / commit / fail words, state stores behind the last gather) rides the same kinds
#endif
  // NRC (late r4): the single-launch steps whose tail keeps the memory order (one gather: (m, l) and O travel together) — the
  // hybrid step and the several-tiles-per-wave steps — are recoverable too when the caller passes commit words (a.commit): the
  // whole commit (y, history / ring column / window sums, keys, counts) sits behind that gather.  HRC: the hybrid part of it (the
  // recorded decision carries the kind, the head's count and the ring column as well).
#ifdef CC_NO_HRC  // (A/B builds)
  constexpr bool NRC = false;
#else
  constexpr bool NRC = ONE && !EML;
#endif
  constexpr bool HRC = NRC && HYB;
  // L2X (r4): the l2 policy's norm maximum crosses kv heads (cache.py:602).  With the placement of XL2 it travels in two levels:
  // the workgroups' maxima inside the head's XCD (plain granules, gathered with the (m, l) pairs by one wave), then ONE granule
  // per kv head through memory (published by the head's split-0 workgroup, gathered by one wave per workgroup in the shadow of the
  // partial-O exchange) — instead of every thread of every workgroup gathering all H x n_split maxima through memory ahead of the
  // final (M, L).  max is associative: the same value, hence the same keys, bit for bit.
  // L2C (r6): the carried norm record — see CC_V_L2CARRY.  cache.py:602 takes the maximum of the norms over ALL kv heads and slots; the
  // step of position p leaves, per kv head, the maximum over the slots the head KEEPS at p + 1 (everything but the slot p + 1 evicts:
  // that slot is the arg-min of the keys this very step scores) in the head's key row tail.  The step of p + 1 then needs no reduction
  // over norms and no hand-off between heads, and — the l2 keys depend on nothing the attention computes — the workgroup's LAST wave
  // does the policy's whole per-slot pass alone, between its scores and its P.V products (the V rows are still landing):
  //   loads   (that wave only, between the K and the V rows) the new keys of all kv heads, the heads' records, position and norm of
  //           every slot of the workgroup;
  //   fold    the new keys' norms (the inserting row group's arithmetic, bit for bit: four heads per pass, one per row group) ->
  //           max over heads of max(record, new norm) = the maximum at this position; the keys of the workgroup's slots, their
  //           minimum, the slots' two largest norms after the insert (with a holder) -> two granules inside the head's XCD; the
  //           minimum key goes to the wave's entry of the key row with the step's commit (the other waves' entries stay ~0);
  //   record  the last wave of the head's split-0 workgroup gathers the workgroups' granules with the partial-O round (they left a
  //           whole streaming phase earlier), finds the slot position p + 2 will evict, resolves the head's maximum without it and
  //           stores the record of THIS position with the step's commit (entry [live + (p & 1)]: a retry of p still finds p - 1's in
  //           the other entry).
  // Nothing of the r4 / r5 exchange is left in the launch: no norm maxima with the (m, l) pairs, no level-one / level-two gathers, no
  // per-slot pass in the tail, no per-slot state in the other waves.  Seeds and two-launch steps leave the same record through
  // l2_record_kernel (cc_evict.hip).
  constexpr bool L2C = L2 && EML && CC_V_L2CARRY != 0;
  constexpr bool L2X = L2 && XL2 && EML && !L2C;
  static_assert(sizeof(T) == 2 && (RT == 1 || RT == 2 || RT == 4 || RT == 8), "16-bit caches, up to 8 query heads per pass (the MFMA has 16 columns)");
  static_assert(!ONE || NW == 4 || ((NW == 8 || (NW == 16 && !L2 && !HYB && QB == 0 && !QKV)) && NT == 1),
                "the single-launch step runs on 4-wave workgroups, or on ONE 8-wave workgroup per CU (single tile); 16 waves: the r6 A/B build (CC_V_NW16)");
  constexpr int D = 128, VEC = 8, RPW = 4, U = 4;
  static_assert(!QKV || (ONE && NT == 1 && !HYB && QB == 0 && !L2 && !FULL && (NW == 4 || NW == 8) && CC_V_LDSDMA != 0 && CC_V_WORDSFIRST != 0),
                "the fused projection rides the lean single-tile step of the plain 16-bit caches");
  static_assert(FULL || ONE, "the lean form exists for the single-launch step only");
  static_assert(!XL2 || ONE, "the L2-resident hand-off belongs to the single-launch steps");
  auto apply_consts = [&]() {
    if constexpr (!FULL) {  // constants for the optimiser: every `if (a.trace)`, `a.abl & ...`, `if (a.attn_out)` below folds away
      a.trace = nullptr;
      a.abl = 0;
      a.attn_out = nullptr;
    }
    if constexpr (ONE) a.ring_col = nullptr;  // (the single-launch steps derive the ring column themselves)
  };
  apply_consts();
  if (a.ring_col && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x == 0)
    *a.ring_col = (int)(*a.ring_counter % a.ring_W);
  __shared__ __attribute__((aligned(16))) uint4 sm_k[NW][16][16];  // [wave][tile row i][slot]: 4 KiB per wave
  __shared__ __attribute__((aligned(16))) uint4 sm_v[NW][16][16];  // [wave][tile row][chunk ^ 2*(row & 7)]: V tile, row major
  __shared__ float sm_wm[NW][RT], sm_wl[NW][RT];  // the waves' softmax state per query head (merged across the workgroup)
  __shared__ float sm_wf[NW][RT];                 // EML: the waves' merge factors exp(m_w - M) per query head (CC_V_WFACT)
  __shared__ unsigned sm_mlcnt;                    // EML: waves whose (m, l) rows are in LDS — the LAST one to arrive publishes
  __shared__ unsigned sm_fail;                     // EML: some wave of this workgroup gave up waiting (recoverable hand-off)
  __shared__ __attribute__((aligned(16))) float sm_l2w[NW];  // ONE + L2: per-wave maxima of the norms this workgroup's slots hold AFTER the step's insert
  __shared__ __attribute__((aligned(16))) T sm_l2sc[L2 ? NW : 1][L2 ? 128 : 8];  // l2: the new key, transposed for its norm (the slabs belong to the DMA loads)
  __shared__ float sm_gmax;     // L2X: the norm maximum over all kv heads (NaN propagates)
  __shared__ __attribute__((aligned(16))) T sm_l2c4[L2C ? 4 : 1][L2C ? 128 : 8];  // L2C: four new keys, transposed for their norms (one per row group of the workgroup's last wave)
  // QKV: the projection's partial sums [K quarter][row of this workgroup's share], the RMSNorm partials, and the head's gathered
  // q (RT heads), k_new, v_new
  constexpr int QNU = (RT + 2) * 32;  // granule units (4 projection rows each) per kv head
  constexpr int QNG = (QNU + 63) / 64;  // granules per lane of the gathering wave
  constexpr int QXS = 2;                // 1 KiB input segments per wave: K <= 4096 (the launcher checks)
  __shared__ float sm_qpart[QKV ? 4 : 1][QKV ? 64 : 1];
  __shared__ float sm_qred[4];  // QKV: the four K quarters' sums of squares
  __shared__ unsigned sm_qcnt;  // QKV: waves whose partial sums are in LDS — the LAST one to arrive finishes and publishes
  __shared__ __attribute__((aligned(16))) T sm_qkv[QKV ? (RT + 2) * 128 : 8];

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int g = lane >> 4, c = lane & 15;  // row group of the wave / 16-byte column chunk (and MFMA column n)
  unsigned long long qt[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  auto qstamp = [&](int i) {
    if constexpr (QKV && CC_QKV_TRACE != 0) {
      if (wave == 0) qt[i] = __builtin_amdgcn_s_memrealtime();
    }
  };
  qstamp(0);
  if constexpr (QKV) {
    // the LDS counters and the fail flag start at zero: one barrier at the very top, where the waves of a workgroup still run together
    if (threadIdx.x == 0) {
      sm_qcnt = 0u;
      sm_mlcnt = 0u;
      sm_fail = 0u;
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  }
  int split_ = blockIdx.x, h_ = blockIdx.y;
  if constexpr (XL2) {
    // XL2 — placement.  The dispatcher deals the blocks of a grid to the XCDs round-robin in block order: blocks b and b + 8 of a
    // launch always share an XCD (cc_decode_step_probe_xcd observes exactly this relation on the device before this instantiation is
    // ever chosen; WHICH XCD block 0 lands on depends on what was dispatched before — measured, r4: a table of absolute XCC ids
    // taken by a probe launch does not hold for a later launch — so nothing here depends on it).  With kv head = b % H, H a multiple
    // of 8, all workgroups of a kv head sit on ONE XCD, and the head's hand-off — its own (m, l) and partial-O granules — goes
    // through that XCD's L2 (plain stores, sc1 polls) instead of through memory.  Heads interleave in dispatch order: every head
    // needs ALL workgroups of the launch resident (the launcher checks the capacity).  Should the relation ever not hold for a
    // launch (grids of two queues dealt alternately, say), a head's workgroups do not see each other's granules: the bounded wait
    // ends the step as a recoverable failure and the host falls back to the memory hand-off (harness._recover_token).
#if CC_V_PRELOAD
    const int b = blockIdx.x;  // (a 1-D grid of n_split * H blocks: the head count comes preloaded)
#if CC_V_FEWXCD
    const int Hg = pl_misc < 0 ? 8 : lead_H;  // (bit 31: eight virtual heads)
#else
    const int Hg = lead_H;
#endif
#else
    const int b = blockIdx.x + gridDim.x * blockIdx.y;
    const int Hg = (int)gridDim.y;
#endif
    if ((Hg & (Hg - 1)) == 0) {  // (the usual case: no integer division in front of the first loads)
      h_ = b & (Hg - 1);
      split_ = b >> __builtin_ctz(Hg);
    } else {
      h_ = b % Hg;
      split_ = b / Hg;
    }
#if CC_V_FEWXCD
    if (h_ >= lead_H) return;  // a virtual head: nothing to do (before any barrier, any load)
#endif
  }
  const int split = split_, h = h_;
  const int S = a.S;
  const int row_begin = split * a.rows_per_split;
  const int row_end = min(S, row_begin + a.rows_per_split);
  const T* kb = reinterpret_cast<const T*>(a.k) + (size_t)h * S * D;
  // what hangs on the argument block's later fields (LATE: worked out again behind phase 1)
  int q0 = 0;
  const T* vh = nullptr;
  const uint8_t *kqb = nullptr, *vqh = nullptr;  // QB: byte images (element offsets are byte offsets) ...
  const float2* qpar = nullptr;                  // ... and the head's row parameters: [slot][0] = K pair, [1] = V pair
  bool has_mask = false;
  const uint8_t* mh = nullptr;
  T* sc_out = nullptr;
  auto derive = [&]() {
    q0 = h * a.R + blockIdx.z * RT;
    vh = reinterpret_cast<const T*>(a.v) + (size_t)h * S * D + c * VEC;
    kqb = reinterpret_cast<const uint8_t*>(a.k) + (size_t)h * S * D;
    vqh = reinterpret_cast<const uint8_t*>(a.v) + (size_t)h * S * D + c * VEC;
    qpar = reinterpret_cast<const float2*>(a.qparams) + (size_t)h * S * 2;
    has_mask = a.mask != nullptr && !(a.abl & 4);
    mh = has_mask ? a.mask + (size_t)h * S : reinterpret_cast<const uint8_t*>(a.k);
    sc_out = reinterpret_cast<T*>(a.scores);
  };
  if constexpr (!LATE) derive();

  float m = -INFINITY, l = 0.f;  // softmax state of (wave, query head c) / (row group g, head c); meaningful for c < RT
  // O^T accumulators of the P.V MFMAs: block b covers output columns 16b .. 16b+15; lane (g, n = c) holds
  // O[head n][16b + 4g + t] in acc[b][t] — one partial per WAVE (all 16 rows of the tile), not per row group
  f32x4_t acc[D / 16];
#pragma unroll
  for (int b = 0; b < D / 16; b++) acc[b] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  // LDS transpose read (ds_read_b64_tr_b16): lane i of a 16-lane group supplies the address of 4 contiguous columns
  // of row (i / 4); the group receives, lane n, the 4 ROWS of column n — the A fragment of v_mfma_16x16x16 without
  // any register shuffles.  Row r of the V tile is stored with its 16-byte chunks XOR-swizzled by 2*(r & 7), which
  // makes both the row-major ds_write_b128 and the transpose reads bank-conflict free.
  const int tr_row = 4 * g + (c >> 2);
  const int tr_sw = 2 * (tr_row & 7), tr_qh = (c >> 1) & 1, tr_half = c & 1;

  // l2: this thread's first norm of the block is requested here and examined only after the streaming loop
  const bool l2_here = L2 && blockIdx.z == 0;
  const int kn_row0 = row_begin + (int)threadIdx.x;
  float kn_first = -INFINITY;
  if (L2 && !ONE && l2_here && kn_row0 < row_end) kn_first = ElemTraits<T>::load(reinterpret_cast<const T*>(a.key_norm) + (size_t)h * S, kn_row0);
  // ---- every load of the first tile is issued before anything waits (partial keys, q, mask, K, V: use order)
  int ins_idx = -1, ins_was_empty = 0;
  int hyb_kind = 0, hyb_cts = 0;  // HYB: 0 = append at the end, 1 = evict the candidate, 2 = drop (slot S - 1, mask untouched)
  bool hyb_punc = false;
  bool key_pending = a.next_key != nullptr && (LATE || !(a.abl & 128));  // (LATE: the measurement bit is looked at behind phase 1)
  unsigned long long key_part = ~0ull;
  unsigned long long key_more[3] = {~0ull, ~0ull, ~0ull};  // up to 256 entries are requested at once and folded when first used
  // (late r2, measured on one box against the same build without it: these loads and the fold below cost the step 0.35 us — every
  //  wave of a kv head reads the same 2 KB — but leaving both to wave 0 and handing the key to the others through LDS and a
  //  barrier cost 0.2 us MORE: kept as is)
  struct TileRegs {            // the staging registers of one tile in flight
    uint32_t mword;
    Vec16<T> kk[U], vv[U];
    uint2 kq8[U], vq8[U];      // QB: the rows' bytes ...
    float2 kpar[U], vpar[U];   // ... and their (scale, minimum)
  };
  TileRegs tregs[NSUB];
  auto load_nt_u2 = [](const uint8_t* p) {
    typedef unsigned int u32x2_nt __attribute__((ext_vector_type(2)));
    const u32x2_nt v = __builtin_nontemporal_load(reinterpret_cast<const u32x2_nt*>(p));
    return make_uint2(v.x, v.y);
  };
  auto issue_k_rows = [&](TileRegs& R, int base) {
    const int row0 = base + g * U;
#pragma unroll
    for (int u = 0; u < U; u++) {
      const int rr = row0 + u < row_end ? row0 + u : row_end - 1;
      if constexpr (QB) {
        R.kq8[u] = load_nt_u2(kqb + (size_t)rr * D + ((c ^ (4 * g + u)) & 15) * VEC);
        R.kpar[u] = qpar[(size_t)rr * 2];
      } else {
        R.kk[u].load_nt(kb + (size_t)rr * D + ((c ^ (4 * g + u)) & 15) * VEC);
      }
    }
  };
  constexpr bool KFIRST = CC_V_PRO != 0 && ONE1 && !HYB;  // the K rows go out ahead of the (branchy) mask word
  // LDS-DMA (r4 A/B): the tile's rows land in the wave's slabs directly — load u of a wave delivers tile rows 4u .. 4u + 3 (lane
  // (g, c) asks for row 4u + g, the chunk that belongs in slot c of that row under the slab's swizzle) into one contiguous 1 KiB
  // block (lane L -> byte 16 L); no staging registers, no ds_write, and the row OWNERSHIP of the scores (row group g: rows 4g .. 4g + 3,
  // the MFMA's C layout) is untouched
  constexpr bool DMA = CC_V_LDSDMA != 0 && ONE1 && QB == 0 && !HYB;
  // (buffer_load ... lds, not global_load_lds: the compiler's wait-count pass treats the FLAT-encoded form as an access to both
  //  memories and turns every later wait into vmcnt(0) lgkmcnt(0) while one is pending; the MUBUF form is counted exactly)
  auto dma16 = [](__amdgpu_buffer_rsrc_t rs, int voff, void* lp) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)lp, 16, voff, 0, 0, 2 /* nt */);
  };
  auto issue_k_dma = [&](int base) {
    const auto krs = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(kb), 0, S * D * (int)sizeof(T), 0x00020000);
#pragma unroll
    for (int u = 0; u < U; u++) {
      const int i = 4 * u + g, rr = base + i < row_end ? base + i : row_end - 1;
      dma16(krs, rr * (D * (int)sizeof(T)) + ((c ^ i) & 15) * 16, &sm_k[wave][4 * u][0]);
    }
  };
  auto issue_v_dma = [&](int base) {
    const auto vrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(reinterpret_cast<const T*>(a.v) + (size_t)h * S * D), 0, S * D * (int)sizeof(T), 0x00020000);
#pragma unroll
    for (int u = 0; u < U; u++) {
      const int i = 4 * u + g, rr = base + i < row_end ? base + i : row_end - 1;
      dma16(vrs, rr * (D * (int)sizeof(T)) + ((c ^ (2 * (i & 7))) & 15) * 16, &sm_v[wave][4 * u][0]);
    }
  };
  unsigned long long tr_kreq = 0;
  // LATE, phase 1: the rest of the argument block through a laundered pointer — the loads cannot be issued ahead of the asm that
  // "produces" it, i.e. ahead of the K request in front of it — then the preloaded copies and the constants again
  auto late_phase = [&]() {
    if constexpr (LATE) {
      typedef const char __attribute__((address_space(4))) kernarg_bytes;
      kernarg_bytes* kp = (kernarg_bytes*)__builtin_amdgcn_kernarg_segment_ptr();
      asm volatile("" : "+s"(kp));
      static_assert(kLeadBytes % alignof(SplitArgs) == 0, "the argument block follows the 14 preloaded dwords without padding");
      // member by member THROUGH the constant address space (a memcpy through a generic pointer becomes vector loads behind the
      // DMA request — the compiler must assume it wrote what they read — with in-order waits for the K rows in front of them);
      // the hybrid / QKV sub-blocks and the two-launch scratch are not this instantiation's
#if CC_V_PRELOAD
      if constexpr (EARLY) {
        // the loads of the kernel's first instructions: ONE wait, here — they have been in flight for the whole prologue
        asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(ea_q), "+s"(ea_kn), "+s"(ea_ct), "+s"(ea_ao), "+s"(ea_r), "+s"(ea_rn), "+s"(ea_hp), "+s"(ea_y),
                     "+s"(ea_nrm), "+s"(ea_tr), "+s"(ea_abl));
        auto p64 = [](unsigned lo, unsigned hi) { return (unsigned long long)lo | ((unsigned long long)hi << 32); };
        // (pointers rebuilt from integers must be told that they point to GLOBAL memory — kernel arguments are by the ABI — or every
        //  access through them becomes a FLAT instruction, counted in vmcnt AND lgkmcnt: the in-order waits all turn into (0, 0))
        typedef __attribute__((address_space(1))) char gchar;
#define CC_GPTR(T, lo, hi) reinterpret_cast<T>((char*)(gchar*)p64(lo, hi))
        a.q = CC_GPTR(const void*, ea_q[0], ea_q[1]);
        a.v = CC_GPTR(const void*, ea_q[4], ea_q[5]);
        a.mask = CC_GPTR(const uint8_t*, ea_q[6], ea_q[7]);
        a.R = (int)ea_r[0];
        a.n_split = (int)ea_r[1];
        a.scale = __uint_as_float(ea_r[3]);
        a.abl = (int)ea_abl;
        a.k_new = CC_GPTR(const void*, ea_kn[0], ea_kn[1]);
        a.v_new = CC_GPTR(const void*, ea_kn[2], ea_kn[3]);
        a.pos = CC_GPTR(int32_t*, ea_kn[4], ea_kn[5]);
        a.mask_w = CC_GPTR(uint8_t*, ea_kn[6], ea_kn[7]);
        a.cache_cts = CC_GPTR(int32_t*, ea_ct[0], ea_ct[1]);
        a.num = CC_GPTR(double*, ea_ct[2], ea_ct[3]);
        a.denom = CC_GPTR(int32_t*, ea_ct[4], ea_ct[5]);
        a.H = (int)ea_ct[6];
        a.Hc = (int)ea_ct[7];
        a.Hp = (int)ea_hp[0];
        a.key_norm = CC_GPTR(void*, ea_nrm[0], ea_nrm[1]);
        // (the granule regions sit at fixed offsets behind the workspace's header: no loads)
        a.one_ml = reinterpret_cast<char*>(a.one_hdr) + kOneHdrBytes;
        a.one_o = reinterpret_cast<char*>(a.one_hdr) + kOneHdrBytes + kOneMlCap;
        a.one_ml_bytes = (unsigned)kOneMlCap;
        a.one_o_bytes = (unsigned)kOneOCap;
        a.y = CC_GPTR(void*, ea_y[0], ea_y[1]);
        a.attn_out = CC_GPTR(void*, ea_ao[0], ea_ao[1]);
        a.hh_counter = CC_GPTR(int64_t*, ea_ao[2], ea_ao[3]);
        a.g = (int)ea_ao[4];
        a.w = (int)ea_ao[5];
        a.policy = (int)ea_ao[6];
        a.rand_next = CC_GPTR(const float*, ea_rn[0], ea_rn[1]);
        a.rng_seed = p64(ea_rn[2], ea_rn[3]);
        a.trace = CC_GPTR(unsigned long long*, ea_tr[0], ea_tr[1]);
        apply_lead();
        apply_consts();
        derive();
        if (a.abl & 128) key_pending = false;
#undef CC_GPTR
        return;
      }
#endif
      typedef const SplitArgs __attribute__((address_space(4))) args_c;
      args_c* ap = (args_c*)(kp + kLeadBytes);
#define CC_LATE_FIELD(f) a.f = ap->f
      CC_LATE_FIELD(q); CC_LATE_FIELD(v); CC_LATE_FIELD(mask); CC_LATE_FIELD(R); CC_LATE_FIELD(n_split); CC_LATE_FIELD(scale);
      CC_LATE_FIELD(abl); CC_LATE_FIELD(k_new); CC_LATE_FIELD(v_new); CC_LATE_FIELD(pos); CC_LATE_FIELD(mask_w); CC_LATE_FIELD(cache_cts);
      CC_LATE_FIELD(num); CC_LATE_FIELD(denom); CC_LATE_FIELD(H); CC_LATE_FIELD(Hc); CC_LATE_FIELD(Hp); CC_LATE_FIELD(key_norm);
      CC_LATE_FIELD(one_ml); CC_LATE_FIELD(one_o); CC_LATE_FIELD(one_ml_bytes); CC_LATE_FIELD(one_o_bytes); CC_LATE_FIELD(y);
      CC_LATE_FIELD(attn_out); CC_LATE_FIELD(hh_counter); CC_LATE_FIELD(g); CC_LATE_FIELD(w); CC_LATE_FIELD(policy); CC_LATE_FIELD(rand_next);
      CC_LATE_FIELD(rng_seed); CC_LATE_FIELD(trace); CC_LATE_FIELD(virt8);
#undef CC_LATE_FIELD
      apply_lead();
      apply_consts();
      derive();
      if (a.abl & 128) key_pending = false;  // (measurement: the key row was read, and is ignored)
    }
  };
  auto issue_k = [&](TileRegs& R, int base) {  // mask word + the four K rows of this lane's row group (tile row i = 4g + u takes chunk c ^ i)
    const int row0 = base + g * U;
    if constexpr (DMA) {
      issue_k_dma(base);
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (LATE && FULL) tr_kreq = __builtin_amdgcn_s_memtime();  // (the trace's "K requested": the block's trace pointer arrives with phase 1, behind it)
      if constexpr (LATE && !SMALLFIRST) late_phase();  // (single tile: this is the step's one K request)
    } else
    if constexpr (KFIRST) {
      issue_k_rows(R, base);
      __builtin_amdgcn_sched_barrier(0);
    }
    if constexpr (DMA && CC_V_FLATLOADS != 0) {
      // one aligned word per lane, no branch: a lane whose four rows are not one aligned word inside the cache (a ragged end, an odd
      // base) reads the first word of the K cache instead and takes the bytes one by one where the word is consumed (mword_fixup)
      const bool okw = has_mask && row0 + 3 < S && ((reinterpret_cast<uintptr_t>(mh) + (size_t)row0) & 3) == 0;
      R.mword = *reinterpret_cast<const uint32_t*>(okw ? mh + row0 : reinterpret_cast<const uint8_t*>(a.k));
    } else {
    R.mword = 0x01010101u;
    if (has_mask) {
      if (row0 + 3 < S && ((reinterpret_cast<uintptr_t>(mh) + (size_t)row0) & 3) == 0) {
        R.mword = *reinterpret_cast<const uint32_t*>(mh + row0);
      } else {
        R.mword = 0;
#pragma unroll
        for (int u = 0; u < U; u++)
          if (row0 + u < S) R.mword |= (uint32_t)mh[row0 + u] << (8 * u);
      }
    }
    }
    if constexpr (!KFIRST && !DMA) issue_k_rows(R, base);
  };
  auto issue_v = [&](TileRegs& R, int base) {
    if constexpr (DMA) {
      issue_v_dma(base);
      return;
    }
    const int row0 = base + g * U;
#pragma unroll
    for (int u = 0; u < U; u++) {
      const int rr = row0 + u < row_end ? row0 + u : row_end - 1;
      if constexpr (QB) {
        R.vq8[u] = load_nt_u2(vqh + (size_t)rr * D);
        R.vpar[u] = qpar[(size_t)rr * 2 + 1];
      } else {
        R.vv[u].load_nt(vh + (size_t)rr * D);
      }
    }
  };
  int base = row_begin + wave * (RPW * U);
  bool more = base < row_end;
  // 16-bit caches (late r3): the (first) tile's K rows are requested right BEHIND the key row — ahead of the per-slot state, the
  // epoch / status words and q, ~150 instructions earlier than with the rest of the tile.  Same-box A/B, three runs each: heavy
  // hitter 9.13 -> 8.92 us at S = 4096 (l2 10.5 -> 10.3, recent_global 8.98 -> 8.9, random 9.2 -> 9.1; S = 2560 and one kv head
  // unchanged).  The neighbours of this placement all LOSE: K ahead of the key row +0.1 (every workgroup's first decision then
  // queues behind 16 MB of rows), K and V both here +0.2, q moved up with K +0.1 (+0.5 at one kv head), the per-slot state moved
  // behind V +0.45.
#ifdef CC_NO_KEARLY  // (A/B builds)
  constexpr bool KEARLY = false;
#else
  // (several tiles per wave — hybrid included: the first tile's K rows likewise; C4 hybrid at S = 18432 25.8 -> 25.1 us and
  //  17.15 -> 16.7 at S = 9000 on two boxes, unchanged on a third; heavy hitter unchanged: its stream is bandwidth-bound.  NOT the
  //  hybrid cache's single-tile step: 11.1 -> 11.6 us at S = 4096 with it — its decision operands want to be ahead of the rows)
  constexpr bool KEARLY = ONE && QB == 0 && !(HYB && NT == 1) && !SMALLFIRST;
#endif
  unsigned one_tag = 0;
  int32_t one_pin = 0;
  unsigned rc_status = 0;    // EML: the workspace's status word (a step failed since the host last looked: do nothing)
  int32_t rc_commit = -2;    // EML: step_commit[h][2 + split]: the last position this workgroup committed
  int32_t rc_insw = 0, rc_insp = -2;  // EML: step_commit[h][0 .. 1]: the insert slot word of position rc_insp
  int32_t rc_cts = 0, rc_col = 0;     // hybrid: step_commit[h][66 .. 67]: the head's count before that step, its ring column
  // VW (r6, with the preloaded arguments): in the LDS-DMA steps the words travel as VECTOR loads of a wave-uniform address (an
  // opaque zero in a VGPR keeps the compiler from turning them back into scalar loads), the first loads the wave issues.  As scalar
  // loads they shared lgkmcnt with the argument block's own loads: one register reused between the two kinds and the compiler put an
  // `s_waitcnt lgkmcnt(0)` — a cold round trip to memory — in front of the key row and the first K request.  Vector loads return in
  // order: the wait where the words are first consumed (words_to_sgpr, behind the issue of the whole tile) is an exact vmcnt(N).
  constexpr bool VW = CC_V_PRELOAD != 0 && DMA;
  auto load_step_words = [&]() {
    if constexpr (VW) {
      int vz;
      asm volatile("v_mov_b32 %0, 0" : "=v"(vz));
      one_tag = a.one_hdr[h + vz];
      one_pin = a.input_pos[vz];
      rc_status = a.one_hdr[kOneStatusWordDev + vz];
      // (no branch around the loads: without commit words a valid dummy — the header — is read and ignored)
      const int32_t* cw = a.commit ? a.commit + (size_t)h * kRcStride : reinterpret_cast<const int32_t*>(a.one_hdr);
      rc_commit = cw[2 + split + vz];
      rc_insw = cw[vz];
      rc_insp = cw[1 + vz];
      return;
    }
    one_tag = a.one_hdr[h] + 1u;
    one_pin = *a.input_pos;
    if constexpr (RC || NRC) {
      rc_status = a.one_hdr[kOneStatusWordDev];
      if (a.commit) {
        rc_commit = a.commit[(size_t)h * kRcStride + 2 + split];
        rc_insw = a.commit[(size_t)h * kRcStride];
        rc_insp = a.commit[(size_t)h * kRcStride + 1];
        if constexpr (HRC) {
          rc_cts = a.commit[(size_t)h * kRcStride + 66];
          rc_col = a.commit[(size_t)h * kRcStride + 67];
        }
      }
    }
  };
  auto words_to_sgpr = [&]() {  // VW: the words are back (the compiler's wait sits right here) — wave-uniform values again
    if constexpr (VW) {
      one_tag = (unsigned)__builtin_amdgcn_readfirstlane((int)one_tag) + 1u;
      one_pin = __builtin_amdgcn_readfirstlane(one_pin);
      rc_status = (RC || NRC) ? (unsigned)__builtin_amdgcn_readfirstlane((int)rc_status) : 0u;
      const bool have = (RC || NRC) && a.commit != nullptr;
      rc_commit = have ? __builtin_amdgcn_readfirstlane(rc_commit) : -2;
      rc_insw = have ? __builtin_amdgcn_readfirstlane(rc_insw) : 0;
      rc_insp = have ? __builtin_amdgcn_readfirstlane(rc_insp) : -2;
    }
  };
#if CC_V_WORDSFIRST
  if constexpr (DMA) {
    // the step's wave-uniform words are requested AHEAD of the first DMA load: behind it they could no longer travel as scalar
    // loads (the compiler must assume the DMA writes memory they read) and would become vector loads with a wait for the whole tile.
    // FIRST of all (late r4): scalar loads return out of order, so every wait for kernel arguments further down is a wait for
    // these words as well — requested here, at the top, they have arrived by then; requested right in front of the first K rows
    // (r4) they made the K request wait a memory round trip for the commit words (found with a time stamp at the kernel's entry).
    load_step_words();
    __builtin_amdgcn_sched_barrier(0);
#if CC_V_KPIN && !CC_V_PRELOAD
    // ... and behind them the kernel arguments the key row, the first K / V rows, q, the mask and the per-slot state need, in ONE
    // round of scalar loads (left to the compiler they arrive in three, each behind the previous one's wait).  Only here: behind a
    // volatile asm the compiler reads nothing from memory through scalar loads any more — the step's words are already on their way.
    asm volatile("" ::"s"(a.next_key), "s"(a.nk), "s"(a.nk_read), "s"(a.k), "s"(a.v), "s"(a.q), "s"(a.mask), "s"(a.S), "s"(a.n_split),
                 "s"(a.rows_per_split), "s"(a.num), "s"(a.denom), "s"(a.pos), "s"(a.Hp));
#endif
  }
#endif
  if (key_pending) {
    // KEY ROWS (late r3).  Every kv head reads — and at the end of the step rewrites — ITS OWN row, also under the head-constant
    // policies, whose rows all hold the same keys.  They used to share row 0, rewritten by kv head 0's waves once THEIR head's
    // workgroups had all published: nothing made a workgroup of another head read the row before that.  Found by the differential
    // fuzz at 16 workgroups (recent_global, 8 kv heads, S = 101): on 3.6 % of the steps the workgroups of two kv heads — the two
    // whose block ids fall on the same pair of XCDs, woken late on an otherwise idle chip — read the NEXT position's candidate
    // and put their row into the wrong slot, silently.  A head's own row is safe by the argument that already covers the
    // head-specific policies: its writers have gathered every workgroup of the head, i.e. every reader has published, i.e. read.
    const unsigned long long* krow = a.next_key + (size_t)h * a.nk;
    if constexpr (CC_V_PRO != 0 && ONE) {
      // no branch around the loads: a lane past the row's live entries reads the row's last live entry again (the minimum does not
      // change); four exec-mask branches less in front of the first K rows
      const int last = a.nk_read - 1;
      key_part = krow[lane < last ? lane : last];
#pragma unroll
      for (int j = 0; j < 3; j++)
        if (64 * (j + 1) < a.nk_read) key_more[j] = krow[lane + 64 * (j + 1) < last ? lane + 64 * (j + 1) : last];  // (wave-uniform test)
    } else {
      if (lane < a.nk_read) key_part = krow[lane];
#pragma unroll
      for (int j = 0; j < 3; j++)
        if (lane + 64 * (j + 1) < a.nk_read) key_more[j] = krow[lane + 64 * (j + 1)];
    }
    // rows beyond 256 live entries (two-launch step at S > 32768) — kept OUT of the streaming loop so that the waits there stay exact
    for (int i = lane + 256; i < a.nk_read; i += 64) {
      const unsigned long long x = a.next_key[(size_t)h * a.nk + i];
      key_part = x < key_part ? x : key_part;
    }
  }
#if !CC_V_WORDSFIRST
  if constexpr (DMA) {
    load_step_words();
    __builtin_amdgcn_sched_barrier(0);
  }
#endif
  // ---- QKV: the projection's operands — the input vector, then round 0 of this wave's weight rows — go out HERE, ahead of the
  //      tile (the tile is not needed before q exists; the weights are on the path to q)
  const int q_w4 = wave & 3, q_hw = wave >> 2;  // K quarter of this wave / wave half (8-wave workgroups: two halves share the rows)
  int q_u0 = 0, q_ub = 0, q_uh = 0, q_mine0 = 0, q_nmine = 0, q_nch = 0, q_nstep = 0;
  // The input vector: every wave keeps the chunks of ITS K quarter (cc_gemv.hip's layout); the RMSNorm's sum of squares crosses the
  // waves through LDS and ONE bare barrier placed behind the first weight unit's request — early, where the waves still run together
  // (r5 traces: behind ALL of a wave's requests, every wave's dots waited until the workgroup's slowest wave had ISSUED its last
  // load, 4 us later; every wave reading the whole vector instead — 18 loads per lane in front of the weights — delayed the first
  // weight request by 1 us)
  Vec16<T> q_xa[QXS], q_da[QXS], q_nv[QXS];
  constexpr int QDEPTH = CC_QKV_DEPTH;
  static_assert(QDEPTH == 2 || QDEPTH == 3, "two or three weight units in flight");
  uint4 q_w[QDEPTH][4][QXS];  // QDEPTH units (4 rows x QXS segments each) in flight per lane: see the rounds below
  uint32_t q_fr_raw = 0;  // the finishing lane's (cos, sin) pair and bias, raw (converted where used)
  uint16_t q_bi_raw = 0;
  auto q_unit_row = [&](int unit) {  // first projection row (of W) of the head's granule unit `unit`
    const int i = unit * 4;
    return i < RT * D ? (h * RT) * D + i : (i < RT * D + D ? a.qkv.HQ * D + h * D + (i - RT * D) : (a.qkv.HQ + a.H) * D + h * D + (i - RT * D - D));
  };
  // one granule unit (4 rows x QXS segments: 8 loads of 16 bytes per lane) of this wave half into buffer SLOT.  Few loads in flight
  // per lane on purpose (tools/probes/stream_occ_probe: a 256 x 512-thread launch streams 67 MB at 5.7 TB/s with 8 loads per lane
  // outstanding and at 4.8-5.0 with 32-48 — r5's first cut had 45 outstanding and streamed at 4.9)
  auto q_issue_w = [&](auto slot_c, int ul) {
    constexpr int SLOT = decltype(slot_c)::value;
    const uint4* Wv = reinterpret_cast<const uint4*>(a.qkv.W);
    const int grow0 = q_unit_row(q_u0 + q_mine0 + ul);
#pragma unroll
    for (int t = 0; t < 4; t++)
#pragma unroll
      for (int j = 0; j < QXS; j++) {
        const int cc = (j * 4 + q_w4) * 64 + lane;  // (a chunk past the row re-reads chunk 0: its input chunk is zero)
        q_w[SLOT][t][j] = nt_load(Wv + (size_t)(grow0 + t) * q_nch + ((j < q_nstep && cc < q_nch) ? cc : 0));
      }
  };
  if constexpr (QKV) {
    const int ns_ = a.n_split;
    q_u0 = (QNU * split) / ns_;
    q_ub = (QNU * (split + 1)) / ns_ - q_u0;        // units of this workgroup (<= 16: the launcher checks)
    q_uh = NW == 8 ? (q_ub + 1) >> 1 : q_ub;        // units of the first wave half
    q_mine0 = q_hw == 0 ? 0 : q_uh;
    q_nmine = q_hw == 0 ? q_uh : q_ub - q_uh;
    q_nch = a.qkv.K / VEC;
    q_nstep = (((q_nch + 63) >> 6) + 3) >> 2;
    const T* xg = reinterpret_cast<const T*>(a.qkv.x);
    const T* dg = reinterpret_cast<const T*>(a.qkv.delta);
    const T* wg = reinterpret_cast<const T*>(a.qkv.norm_w);
#pragma unroll
    for (int j = 0; j < QXS; j++) {
      const int cc = (j * 4 + q_w4) * 64 + lane;
      const size_t co = (size_t)((j < q_nstep && cc < q_nch) ? cc : 0) * VEC;  // (unconditional, like the weights; zeroed where used)
      q_xa[j].load(xg + co);
      q_da[j].load((dg != nullptr ? dg : xg) + co);
      q_nv[j].load(wg + co);
    }
    {  // the finishing lane's RoPE pair and bias (lane = row of the workgroup's share; whichever wave ends up finishing)
      const int rl = lane < q_ub * 4 ? lane : 0;
      const int grow = q_unit_row(q_u0 + (rl >> 2)) + (rl & 3);
      const uint32_t* fp = a.qkv.freqs ? reinterpret_cast<const uint32_t*>(a.qkv.freqs) + ((grow & (D - 1)) >> 1) : reinterpret_cast<const uint32_t*>(xg);
      const uint16_t* bp = a.qkv.bias ? reinterpret_cast<const uint16_t*>(a.qkv.bias) + grow : reinterpret_cast<const uint16_t*>(xg);
      q_fr_raw = *fp;
      q_bi_raw = *bp;
    }
    if constexpr (CC_QKV_TILE_AT == 0) {
      issue_k(tregs[0], base);
      issue_v(tregs[0], base);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (q_nmine > 0) q_issue_w(IntC<0>{}, 0);
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (CC_QKV_TILE_AT == 1) {
      issue_k(tregs[0], base);
      issue_v(tregs[0], base);
      __builtin_amdgcn_sched_barrier(0);
    }
    qstamp(1);
  }
  // SMALLFIRST: the rest of the argument block (requested at the kernel's first instruction) is taken HERE, behind the key row's
  // requests; the per-slot state, q, the K rows, the mask word and the V rows follow in program order below
  if constexpr (SMALLFIRST) late_phase();
  constexpr bool VEARLY = DMA && !QKV && (CC_V_VEARLY == 2 || (CC_V_VEARLY == 1 && !XL2));
  if constexpr (KEARLY && !QKV) {  // (QKV: the tile is requested behind the projection's last weight rows, below)
    issue_k(tregs[0], base);
    if constexpr (VEARLY) issue_v(tregs[0], base);
    __builtin_amdgcn_sched_barrier(0);  // (keeps the scheduler from sinking them back to the rest of the tile)
  }
  // (r4, with the DMA loads: the V rows right behind the K rows +0.4 us, the K rows ahead of the key row +0.35, both +0.8 — whatever
  //  is requested behind the bulk (q, mask word, per-slot state) is usable only when the bulk has landed: loads return in order)
  // HYB: everything the per-head decision needs besides the candidate key — the policy table (ALL rows: one vector load, the
  // head's row is picked by a lane read once its policy index has arrived), the punctuation ids (one id per lane), the head's
  // policy index and count, the budget terms, the incoming token — as FIRST-LEVEL loads issued here, ahead of q and the tile.
  // (They used to be fetched behind the tile as a dependent chain strategies[h] -> table[pol]: two serial misses to HBM — every
  // layer has its own few bytes of these, long evicted when its turn comes again — which the first tile's latency did not cover:
  // the first K rows of EVERY workgroup arrived 2 us later than in the heavy-hitter step.)
  int hy_tabv = 0, hy_pol = 0, hy_nsp = 0, hy_npu = 0;
  long long hy_pidv = 0, hy_tok = 0, hy_ctr = 0;
  int hy_flags = 0, hy_cts = 0, hy_budget = 0, hy_win = 0;
  bool hy_punc = false;
  if constexpr (HYB) {
    if (lane < a.hyb.n_pol * 3) hy_tabv = a.hyb.table[lane];
    const bool punc_on = a.hyb.token_id != nullptr && a.hyb.punc_ids != nullptr;
    if (punc_on && lane < a.hyb.n_punc_ids) hy_pidv = a.hyb.punc_ids[lane];
    hy_pol = (int)a.hyb.strategies[h];
    hy_cts = a.cache_cts[h];
    if (punc_on) hy_tok = *a.hyb.token_id;
    if (a.hyb.num_special) hy_nsp = *a.hyb.num_special;
    if (a.hyb.num_punc) hy_npu = *a.hyb.num_punc;
    if (ONE && a.ring_num) hy_ctr = *a.ring_counter;
    // more than 64 punctuation ids: the rest is compared here (a wait on the token id at the top of the kernel, on this path
    // only) — kept out of the decision so that the in-order wait counts around the tile stay exact
    if (punc_on)
      for (int k2 = lane + 64; k2 < a.hyb.n_punc_ids; k2 += 64) hy_punc |= a.hyb.punc_ids[k2] == hy_tok;
    __builtin_amdgcn_sched_barrier(0);
  }
  // ONE: this lane's slot of the per-slot pass (lane c = t * LPR of row group g finishes row 4g + t of the wave's tile): its
  // history and position are requested here, with everything else, and consumed after the hand-off
  unsigned long long tr0 = 0, tr1 = 0, tr2 = 0, tr3 = 0, tr4 = 0, rt0 = 0, rt1 = 0, trA = 0, trB = 0, trC = 0;
  if constexpr (ONE) {
    if (a.trace) {
      tr3 = (LATE && FULL) ? tr_kreq : __builtin_amdgcn_s_memtime();  // the first K rows (and the key row, the step words) are requested
      tr0 = tr_entry;
      rt0 = rt_entry;
    }
  }
  // ALL (r4): with several tiles per wave the second half of the per-slot pass — history update, next-eviction score, key: ~70
  // dependent instructions — runs on ALL 64 lanes (lane L takes the wave's slots j = L + 64 k, slot j = row (j & 15) of the wave's
  // tile (j >> 4)) instead of on the 16 score-holding lanes of one tile after the other: ceil(NT / 4) passes instead of NT; the
  // group-mean probabilities reach it through a wave-private LDS row, the per-slot state is requested per (lane, k) — coalesced
  constexpr bool ALL = CC_V_ALLLANES != 0 && ONE && NT > 1 && !HYB;
  constexpr int SN = ALL ? (NT * RPW * U + 63) / 64 : NT;  // per-slot state entries per lane
  double one_numv[SN];
  int32_t one_denv[SN], one_psv[SN];
  float one_rndv[SN];
#pragma unroll
  for (int ti = 0; ti < SN; ti++) {
    one_numv[ti] = 0.0;
    one_denv[ti] = 0;
    one_psv[ti] = 0;
    one_rndv[ti] = 0.f;
  }
  auto all_slot = [&](int k) {  // ALL: the slot of entry k of this lane
    const int j = lane + 64 * k;
    return row_begin + wave * (RPW * U) + (j & (RPW * U - 1)) + (j / (RPW * U)) * (NW * RPW * U);
  };
  auto all_valid = [&](int k) { return lane + 64 * k < NT * RPW * U && all_slot(k) < row_end; };
  constexpr int LPR = RT < 4 ? RT : 4;  // lanes per tile row in the per-slot pass; each computes KH = RT / LPR probabilities
  constexpr int KH = RT / LPR;
  // this lane's slot in tile ti of the wave: one_slot + ti * (NW * RPW * U)
  const int one_slot = row_begin + wave * (RPW * U) + g * U + c / LPR;
  const bool one_lane = ONE && c < U * LPR && (c % LPR) == 0;
  const bool one_have = one_lane && one_slot < row_end;
  // the history / position (/ uniform draw) of this lane's slot in every tile of the wave
  auto load_slot_state = [&]() {
    // A policy without a history reads a valid dummy (element 0 of the K cache) and ignores it — NO branch around these loads:
    // the path that does not load re-zeroes the registers, the compiler's wait-count pass guards that write with a vmcnt(0), and
    // with the K tile's DMA loads in flight every wave of the recent_global / full / random steps sat out its K rows here before it
    // requested q and the V tile (found in the ISA, r4: those steps were SLOWER than the heavy hitter's)
    const double* nump = a.num ? a.num : reinterpret_cast<const double*>(a.k);
    const int32_t* denp = a.num ? a.denom : reinterpret_cast<const int32_t*>(a.k);
    const size_t hist = a.num ? 1 : 0;
#pragma unroll
    for (int ti = 0; ti < SN; ti++) {
      const int sl = ALL ? all_slot(ti) : one_slot + ti * (NW * RPW * U);
      if constexpr (DMA && CC_V_FLATLOADS != 0) {  // every lane, no branch: a lane without a slot reads the split's first (valid, ignored)
        const int slc = (one_lane && sl < row_end) ? sl : row_begin;
        one_numv[ti] = nump[((size_t)h * S + slc) * hist];
        one_denv[ti] = denp[((size_t)h * S + slc) * hist];
        one_psv[ti] = a.pos[(a.Hp == 1 ? 0 : (size_t)h * S) + slc];
        if (a.policy == 3 && a.rand_next) one_rndv[ti] = a.rand_next[slc];
      } else
      if (ALL ? all_valid(ti) : (one_lane && sl < row_end)) {
        one_numv[ti] = nump[((size_t)h * S + sl) * hist];
        one_denv[ti] = denp[((size_t)h * S + sl) * hist];
        one_psv[ti] = a.pos[(a.Hp == 1 ? 0 : (size_t)h * S) + sl];
        if (a.policy == 3 && a.rand_next) one_rndv[ti] = a.rand_next[sl];
      }
    }
  };
  // ONE + HYB: the ring pass of the finish runs on ALL lanes (the 192-bit window accumulators make it ~150 instructions per slot:
  // on the 16 score-holding lanes of a tile it would be four times as long) — lane L takes the wave's slots j = L + 64 k, slot j =
  // row (j & 15) of the wave's tile (j >> 4); the group-mean probabilities reach it through a wave-private LDS row
  constexpr int HSL = (HYB && ONE) ? (NT * RPW * U + 63) / 64 : 1;
  // (RAW load results: nothing is converted, tested or combined where the loads are issued — a conversion or a `!= 0` next to
  //  its load makes the compiler wait for that load on the spot, and the state loads of a slot then go out one miss after the
  //  other: 2.6 us to ISSUE them, measured, instead of 0.1)
  uint16_t hy_old[HSL];
  ulonglong2 hy_a01[HSL], hy_a23[HSL];
  int32_t hy_den[HSL], hy_ps[HSL];
  uint8_t hy_sp[HSL], hy_pu[HSL];  // special / punctuation slot
#pragma unroll
  for (int k = 0; k < HSL; k++) {
    hy_old[k] = 0;
    hy_a01[k] = make_ulonglong2(0, 0);
    hy_a23[k] = make_ulonglong2(0, 0);
    hy_den[k] = 0;
    hy_ps[k] = 0;
    hy_sp[k] = 0;
    hy_pu[k] = 0;
  }
  int one_rcol = 0;  // ONE + HYB: the ring column of this step (every workgroup derives it from the counter: no launch ahead of this one)
  auto hyb_slot = [&](int k) {
    const int j = lane + 64 * k;
    return row_begin + wave * (RPW * U) + (j & (RPW * U - 1)) + (j / (RPW * U)) * (NW * RPW * U);
  };
  auto hyb_valid = [&](int k) { return lane + 64 * k < NT * RPW * U && hyb_slot(k) < row_end; };
  auto hyb_load_state = [&]() {
    const size_t hs = (size_t)a.H * S;
    // absent operands read a valid dummy (the cache mask / the positions) and are ignored where they would be used: no branch
    // around any of these loads
    const uint8_t* spm = a.hyb.special_mask ? a.hyb.special_mask : a.mask_w;
    const uint8_t* pum = a.hyb.punc_mask ? a.hyb.punc_mask : a.mask_w;
    const int32_t* dnp = a.ring_num ? a.denom : a.pos;
    const uint16_t* shadow = a.ring_num ? reinterpret_cast<const uint16_t*>(a.ring_acc + hs * 4 + 2) + (size_t)one_rcol * hs
                                        : reinterpret_cast<const uint16_t*>(a.pos);
    const unsigned long long* accp = a.ring_num ? a.ring_acc : reinterpret_cast<const unsigned long long*>(a.k);
#pragma unroll
    for (int k = 0; k < HSL; k++)
      if (hyb_valid(k)) {
        const size_t i = (size_t)h * S + hyb_slot(k);
        hy_old[k] = shadow[i];
        hy_a01[k] = *reinterpret_cast<const ulonglong2*>(accp + i * 4);
        hy_a23[k] = *reinterpret_cast<const ulonglong2*>(accp + i * 4 + 2);
        hy_den[k] = dnp[i];
        hy_ps[k] = a.pos[i];
        hy_sp[k] = spm[i];
        hy_pu[k] = pum[i];
      }
  };
  // ONE + L2: the key norm of this lane's slot — its RAW 16 bits (converted where it is used: a conversion next to the load makes
  // the compiler wait for the load on the spot, and behind the K tile's DMA loads that wait was vmcnt(0): every wave of the l2
  // step sat out its K rows before it requested q, the new token's rows and the V tile — found in the ISA, r4) ...
  uint16_t one_kn_raw = 0;
  auto one_kn_f = [&]() {
    T e;
    e.x = one_kn_raw;
    return ElemTraits<T>::load(&e, 0);
  };
  float l2_nv_lane = 0.f;   // ... and the inserted key's norm, in the lanes of the row group that inserted it
  unsigned l2_ep[3] = {0u, 0u, 0u};  // ONE + L2: epoch words of the kv heads whose norm granules this thread gathers (read behind the tile's loads, below)
  if constexpr (ONE) {
    if constexpr (!DMA) load_step_words();
    if constexpr (NRC) {
      if (threadIdx.x == 0) sm_fail = 0u;  // (read behind the barriers of the epilogue and the finish)
    }
    // single tile: requested AHEAD of the K/V tile (measured: behind it the step is 0.3 us slower — the tile's in-order waits
    // then end on these stragglers, and the workgroup leaves the streaming part later); several tiles: requested after the
    // publish, in the shadow of the hand-off
    if constexpr (NT == 1 && !HYB && !L2C) {  // (L2C: the workgroup's last wave reads the state of ALL its slots, between the K and the V rows)
      load_slot_state();
      if constexpr (L2)  // (unconditional: lanes without a slot read a valid neighbour and ignore it)
        one_kn_raw = reinterpret_cast<const uint16_t*>(a.key_norm)[(size_t)h * S + (one_have ? one_slot : row_begin)];
    }
  }
  float s_keep[NT][U];  // ONE: the wave's tiles of scores, kept for the per-slot pass
#pragma unroll
  for (int ti = 0; ti < NT; ti++)
#pragma unroll
    for (int t = 0; t < U; t++) s_keep[ti][t] = -INFINITY;
  // QB: the incoming token's rows are requested AHEAD of the tile (chunk c of K and of V per lane, every row group alike), so that
  // the inserting row group can quantise them while the tile is in flight — requested behind the tile, the in-order load counter
  // would hold the ~150 instructions of min / max shuffles and roundings back until the K/V rows have arrived, and the whole kv
  // head waits for its slowest workgroup.  (Quantising in EVERY wave ahead of the tile was tried: 512 workgroups of redundant
  // work cost more than the one critical path saves — 12.2 vs 11.7 us at S = 4096.)
  // The l2 policy does the same (late r2): its norm of the new key (eight more loads, a float64 square root) ran behind the tile's
  // in-order wait, and the inserting workgroup of every head left the streaming part 2.7 us after the others — with every workgroup
  // of the launch waiting for it (l2 step 13.5 -> 12.8 us).  The plain 16-bit caches keep the two loads behind the tile: requested
  // ahead of it by all 2048 waves they cost MORE than the inserting wave gains (A/B on one box: heavy hitter 10.68 -> 10.83 us,
  // recent_global 10.14 -> 10.60 — every wave of a kv head asks for the same four cache lines, one memory channel serves them one
  // by one, and the tile's rows queue behind them in the in-order return).
  constexpr bool AHEAD = QB != 0 || L2;  // the incoming token's rows are requested ahead of the tile
  Vec16<T> qb_kn, qb_vn;
  int32_t p_ins = 0;
  qb_kn.raw = make_uint4(0, 0, 0, 0);
  qb_vn.raw = make_uint4(0, 0, 0, 0);
  if (AHEAD && ((DMA && CC_V_FLATLOADS != 0) || a.k_new)) {  // (the single-launch steps always carry the new rows: no branch around the loads)
    qb_kn.load(reinterpret_cast<const T*>(a.k_new) + (size_t)h * D + c * VEC);
    qb_vn.load(reinterpret_cast<const T*>(a.v_new) + (size_t)h * D + c * VEC);
    p_ins = *a.input_pos;
  }
  unsigned hm_ep = 0;  // L2X: the epoch word of kv head `lane` (its head-maximum granule carries that tag)
  if constexpr (L2X) {
    hm_ep = a.one_hdr[lane < a.H ? lane : 0];  // (ahead of the tile and of anything this workgroup publishes, like l2_ep below)
  } else
  if constexpr (L2 && EML && !L2C) {
    // the epoch words of the kv heads whose norm granules this thread gathers — AHEAD of the tile here: this workgroup's first
    // publish (its (m, l) pairs and norm maximum, right behind the scores) must not happen before these loads have completed
    // (no head's word is bumped before every workgroup of the launch has published), and behind the tile they would make that
    // publish wait for the V rows (loads return in order).  Heads by a reciprocal multiply (exact for e < 2^16: corrected once).
    const float inv_ns = 1.0f / (float)a.n_split;
#pragma unroll
    for (int k = 0; k < 3; k++) {
      const int e = (int)threadIdx.x + k * NW * 64;
      int hh = (int)((float)e * inv_ns);
      hh += ((hh + 1) * a.n_split <= e) ? 1 : 0;
      hh -= (hh * a.n_split > e) ? 1 : 0;
      if (e < a.H * a.n_split) l2_ep[k] = a.one_hdr[hh];
    }
  }
  // B operand: lane (n = c, kb = g) of step j holds q[head n][8 * (4j + g) .. + 8]; columns n >= RT are zero
  Vec16<T> qB[4];
#pragma unroll
  for (int j = 0; j < 4; j++) {
    qB[j].raw = make_uint4(0, 0, 0, 0);
    if constexpr (!QKV) {
      if constexpr (DMA && CC_V_FLATLOADS != 0) {  // every lane, no branch: columns n >= RT aim past the buffer's end (no request, zeros back)
        const auto q_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.q), 0, (q0 + RT) * D * (int)sizeof(T), 0x00020000);
        const u32x4_t qv = __builtin_amdgcn_raw_buffer_load_b128(q_rsrc, c < RT ? ((q0 + c) * D + (4 * j + g) * VEC) * (int)sizeof(T) : 0x7ffffff0, 0, 0);
        qB[j].raw = make_uint4(qv[0], qv[1], qv[2], qv[3]);
      } else
      if (c < RT) qB[j].load(reinterpret_cast<const T*>(a.q) + (size_t)(q0 + c) * D + (4 * j + g) * VEC);
    }
  }
  // L2C: what the workgroup's last wave needs for the policy's per-slot pass — see L2C above.  A wave-uniform branch around loads:
  // harmless HERE (between the K and the V rows), where the only older loads anybody waits for are the K tile and the small ones in
  // front of it — an in-order wait that does not count these loads merely lets a few of the small ones land first.
  constexpr int L2S = L2C ? NW / 4 : 1;  // slots per lane: the workgroup holds NW * 16
  uint4 l2c_kn[2];
  ulonglong2 l2c_rc = make_ulonglong2(0ull, 0ull);
  int32_t l2c_ps[L2S];
  uint16_t l2c_nr[L2S];
  l2c_kn[0] = l2c_kn[1] = make_uint4(0, 0, 0, 0);
#pragma unroll
  for (int k = 0; k < L2S; k++) {
    l2c_ps[k] = 0;
    l2c_nr[k] = 0;
  }
  if constexpr (L2C) {
    if (wave == NW - 1) {
#pragma unroll
      for (int j = 0; j < 2; j++) {  // heads 4 j + row group (heads past the count re-read this head's: valid, ignored)
        const int hj = 4 * j + g < a.H ? 4 * j + g : h;
        l2c_kn[j] = *reinterpret_cast<const uint4*>(reinterpret_cast<const T*>(a.k_new) + (size_t)hj * D + c * VEC);
      }
      l2c_rc = *reinterpret_cast<const ulonglong2*>(a.next_key + (size_t)(lane < a.H ? lane : h) * a.nk + (a.nk - kNextKeyTail));  // lane = kv head
#pragma unroll
      for (int k = 0; k < L2S; k++) {
        const int sl = row_begin + k * 64 + lane < row_end ? row_begin + k * 64 + lane : row_end - 1;
        l2c_ps[k] = a.pos[(size_t)h * S + sl];
        l2c_nr[k] = reinterpret_cast<const uint16_t*>(a.key_norm)[(size_t)h * S + sl];
      }
    }
  }
  // UNCONDITIONAL (rows past the split's end are clamped to its last row, a valid address): behind a branch, the compiler's
  // wait-count bookkeeping merges "tile loads issued" with "none issued" and every later use of an EARLIER load (the partial
  // keys, the incoming token's rows) becomes a wait for all loads — the tile included
#pragma unroll
  for (int sub = 0; sub < NSUB; sub++) {
    if constexpr (!KEARLY) issue_k(tregs[sub], base + sub * NW * RPW * U);
    if constexpr (QB) __builtin_amdgcn_sched_barrier(0);  // K's bytes and parameters go out BEFORE V's: the K stash waits for them only
    if constexpr (DMA && CC_V_VDELAY != 0) __builtin_amdgcn_s_sleep(CC_V_VDELAY);
    if constexpr (!QKV && !VEARLY) issue_v(tregs[sub], base + sub * NW * RPW * U);
    if constexpr (QB) __builtin_amdgcn_sched_barrier(0);
  }
  // ONE + L2: the epoch words of the kv heads whose norm granules this thread will gather.  Read HERE (behind the tile's loads: three integer divisions kept out of the way of the first K rows): every workgroup has read
  // them before it publishes anything, and no head's epoch is bumped before its split-0 workgroup has gathered the granules of
  // ALL workgroups of all heads — so none of these reads can see a bumped word.
  if constexpr (ONE && L2 && !EML) {
#pragma unroll
    for (int k = 0; k < 3; k++) {
      const int e = (int)threadIdx.x + k * NW * 64;
      if (e < a.H * a.n_split) l2_ep[k] = a.one_hdr[e / a.n_split];
    }
  }
  if constexpr (QKV) {
    words_to_sgpr();
    if (rc_status != 0u) return;  // a step of this token failed before this launch: leave everything as it is (the host retries)
    // ================= the projection (cc_gemv.hip's gemv_kernel<T, false, RB, 2, 2>, operation for operation) =================
    // Four waves split K into quarters exactly like the stand-alone GEMV's workgroup (wave w: the 1 KiB segments w and w + 4 of a row,
    // one fp32 chain per lane over its segments, the DPP wave sum, then (p0 + p1) + (p2 + p3)); an 8-wave workgroup runs two such
    // groups on different rows.  No barrier: the waves meet at an LDS arrival counter, the LAST one to arrive finishes the rows and
    // publishes (the form of the early (m, l) hand-off below — a __syncthreads() would wait for the tile's DMA loads, queued BEHIND the
    // weights; even a bare barrier would hold every wave until the slowest had its weights).
    qstamp(2);
    uint4 q_xr[QXS];
    {
      const T* dg = reinterpret_cast<const T*>(a.qkv.delta);
      const auto ho_rsrc = __builtin_amdgcn_make_buffer_rsrc(a.qkv.h_out, 0, a.qkv.h_out ? a.qkv.K * (int)sizeof(T) : 0, 0x00020000);
      const bool ho_mine = a.qkv.h_out != nullptr && h == 0 && split == 0 && q_hw == 0;
      float ss = 0.f;
#pragma unroll
      for (int j = 0; j < QXS; j++) {
        const int cc = (j * 4 + q_w4) * 64 + lane;
        const bool in = j < q_nstep && cc < q_nch;
        float hh[VEC];
        q_xa[j].unpack(hh);
#pragma unroll
        for (int e = 0; e < VEC; e++) hh[e] = in ? hh[e] : 0.f;
        if (dg != nullptr) {
          float d[VEC];
          q_da[j].unpack(d);
#pragma unroll
          for (int e = 0; e < VEC; e++) hh[e] = in ? ElemTraits<T>::rnd(__fadd_rn(hh[e], d[e])) : 0.f;  // model-dtype residual add
        }
#pragma unroll
        for (int e = 0; e < VEC; e++) ss = fmaf(hh[e], hh[e], ss);  // (chunks past the row are zeros: nothing added)
        q_xr[j] = pack16<T>(hh);
        // h_out <- x + delta by ONE workgroup's first four waves; unconditional form (everybody else aims past the buffer's end)
        const u32x4_t hv = {q_xr[j].x, q_xr[j].y, q_xr[j].z, q_xr[j].w};
        __builtin_amdgcn_raw_buffer_store_b128(hv, ho_rsrc, (ho_mine && in) ? cc * 16 : 0x7ffffff0, 0, 0);
      }
      ss = gv_wave_sum(ss);
      if (lane == 0 && q_hw == 0) sm_qred[q_w4] = ss;
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      const float tot = (sm_qred[0] + sm_qred[1]) + (sm_qred[2] + sm_qred[3]);
      const float rs = rsqrtf(tot / (float)a.qkv.K + a.qkv.eps);  // ref: model.py:452-457 (fp32 inside)
#pragma unroll
      for (int j = 0; j < QXS; j++) {
        Vec16<T> hv;
        float hh[VEC], wf[VEC], o[VEC];
        hv.raw = q_xr[j];
        hv.unpack(hh);
        q_nv[j].unpack(wf);
#pragma unroll
        for (int e = 0; e < VEC; e++) o[e] = ElemTraits<T>::rnd(__fmul_rn(ElemTraits<T>::rnd(cc_opaque_f32(__fmul_rn(hh[e], rs))), wf[e]));
        q_xr[j] = pack16<T>(o);
      }
    }
    if (q_nmine > 1) q_issue_w(IntC<1>{}, 1);  // (the second unit: behind the barrier — in front of it, only 14 requests per lane are out)
    if constexpr (QDEPTH == 3) {
      if (q_nmine > 2) q_issue_w(IntC<2>{}, 2);
    }
    __builtin_amdgcn_sched_barrier(0);
    qstamp(3);
    // rounds: unit ul of this half is consumed from buffer ul & 1 while unit ul + 1 is in flight; unit ul + 2 is requested into the
    // freed buffer.  The K / V tile is requested right behind the LAST unit's request: the weights are on the path to q, the tile
    // is needed only once q is there.
    auto q_consume = [&](auto slot_c, int ul) {
      constexpr int SLOT = decltype(slot_c)::value;
#pragma unroll
      for (int t = 0; t < 4; t++) {
        float acc = 0.f;
#pragma unroll
        for (int j = 0; j < QXS; j++) acc = Dot16<T>::run(q_w[SLOT][t][j], q_xr[j], acc);
        const float sres = gv_wave_sum(acc);
        if (lane == 0) sm_qpart[q_w4][(q_mine0 + ul) * 4 + t] = sres;
      }
    };
    int q_ul = 0, q_slot = 0;  // q_slot = q_ul % QDEPTH
    for (; q_ul + QDEPTH < q_nmine; q_ul++) {
      if (q_slot == 0) {
        q_consume(IntC<0>{}, q_ul);
        q_issue_w(IntC<0>{}, q_ul + QDEPTH);
      } else if (q_slot == 1) {
        q_consume(IntC<1>{}, q_ul);
        q_issue_w(IntC<1>{}, q_ul + QDEPTH);
      } else {
        q_consume(IntC<QDEPTH - 1>{}, q_ul);
        q_issue_w(IntC<QDEPTH - 1>{}, q_ul + QDEPTH);
      }
      q_slot = q_slot + 1 == QDEPTH ? 0 : q_slot + 1;
    }
    if constexpr (CC_QKV_TILE_AT == 2) {
      issue_k(tregs[0], base);
      issue_v(tregs[0], base);
      __builtin_amdgcn_sched_barrier(0);
    }
    for (; q_ul < q_nmine; q_ul++) {
      if (q_slot == 0) q_consume(IntC<0>{}, q_ul);
      else if (q_slot == 1) q_consume(IntC<1>{}, q_ul);
      else q_consume(IntC<QDEPTH - 1>{}, q_ul);
      q_slot = q_slot + 1 == QDEPTH ? 0 : q_slot + 1;
    }
    qstamp(4);
    // (the counter was zeroed at the top of the kernel, a barrier ago)
    unsigned q_arrived = 0;
    if (lane == 0) {
      const unsigned cnt_addr = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned*)&sm_qcnt;
      asm volatile("s_waitcnt lgkmcnt(0)\n\tds_add_rtn_u32 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=v"(q_arrived) : "v"(cnt_addr), "v"(1u) : "memory");
    }
    q_arrived = (unsigned)__builtin_amdgcn_readfirstlane((int)q_arrived);
    {  // the LAST wave to arrive: lane = row of this workgroup's share — bias, the Linear's rounding, RoPE on the (even, odd) pairs, publish
      const bool fin = q_arrived == (unsigned)(NW - 1);
      const int nrows = q_ub * 4;
      const int rl = lane < nrows ? lane : 0;
      float pq[4] = {0.f, 0.f, 0.f, 0.f};
      if (fin) {
#pragma unroll
        for (int w4 = 0; w4 < 4; w4++) pq[w4] = sm_qpart[w4][rl];
      }
      float sres = (pq[0] + pq[1]) + (pq[2] + pq[3]);
      const int grow = q_unit_row(q_u0 + (rl >> 2)) + (rl & 3);
      if (a.qkv.bias != nullptr) {
        T be;
        be.x = q_bi_raw;
        sres += ElemTraits<T>::load(&be, 0);
      }
      sres = ElemTraits<T>::rnd(cc_opaque_f32(sres));  // the Linear's output in the model dtype (fp32 first, then T: see cc_gemv.hip)
      float out = sres;
      if (a.qkv.freqs != nullptr) {  // ref: model.py:507-519; the pair partner lives in lane ^ 1
        const float other = gv_dpp<0xB1>(out);
        if (grow < (a.qkv.HQ + a.H) * D) {
          T ce, se;
          ce.x = (uint16_t)(q_fr_raw & 0xffffu);
          se.x = (uint16_t)(q_fr_raw >> 16);
          const float cs_ = ElemTraits<T>::load(&ce, 0), sn_ = ElemTraits<T>::load(&se, 0);
          out = (grow & 1) ? __fadd_rn(__fmul_rn(out, cs_), __fmul_rn(other, sn_)) : __fsub_rn(__fmul_rn(out, cs_), __fmul_rn(other, sn_));
        }
      }
      T ebits;
      ElemTraits<T>::store(&ebits, 0, cc_opaque_f32(out));
      const int b0 = (int)ebits.x;
      const int b1 = __builtin_amdgcn_update_dpp(0, b0, 0x55, 0xf, 0xf, true);  // quad_perm [1, 1, 1, 1]
      const int b2 = __builtin_amdgcn_update_dpp(0, b0, 0xAA, 0xf, 0xf, true);
      const int b3 = __builtin_amdgcn_update_dpp(0, b0, 0xFF, 0xf, 0xf, true);
      const bool pub = fin && (lane & 3) == 0 && lane < nrows;
      // (unconditional stores: every lane of every wave executes them; all but the publisher's aim past the buffer's end)
      const u32x4_t gq = {one_tag, (unsigned)b0 | ((unsigned)b1 << 16), one_tag, (unsigned)b2 | ((unsigned)b3 << 16)};
      const auto qg_rsrc = __builtin_amdgcn_make_buffer_rsrc(a.qkv.gran, 0, (int)a.qkv.gran_bytes, 0x00020000);
      __builtin_amdgcn_raw_buffer_store_b128(gq, qg_rsrc, pub ? h * kOneQHead + (q_u0 + (lane >> 2)) * 16 : 0x7ffffff0, 0, XL2 ? 0 : kOneAuxCoherent);
      const auto qo_rsrc = __builtin_amdgcn_make_buffer_rsrc(a.qkv.qkv_out, 0, a.qkv.qkv_out ? (a.qkv.HQ + 2 * a.H) * D * (int)sizeof(T) : 0, 0x00020000);
      typedef unsigned int u32x2_q __attribute__((ext_vector_type(2)));
      const u32x2_q po2 = {(unsigned)b0 | ((unsigned)b1 << 16), (unsigned)b2 | ((unsigned)b3 << 16)};
      __builtin_amdgcn_raw_buffer_store_b64(po2, qo_rsrc, (pub && a.qkv.qkv_out) ? grow * (int)sizeof(T) : 0x7ffffff0, 0, 0);
    }
    qstamp(5);
  }
  int qb_ins_u = -1;  // QB: tile row (of this lane's row group) that holds the inserted token — its K chunk is UNswizzled (chunk c)
  if constexpr (EML && !QKV) {  // (QKV: zeroed behind a barrier at the top of the kernel — HERE its waves are microseconds apart, each at its own weights)
    // the arrival counter of the early (m, l) hand-off starts at zero: one barrier HERE, behind the issue of every load of the
    // tile (the waves of a workgroup reach it together; nothing waits for memory)
    if (threadIdx.x == 0) {
      sm_mlcnt = 0u;
      sm_fail = 0u;
    }
    if constexpr (DMA) {
      // a bare barrier: __syncthreads() is a workgroup-scope release fence, and with LDS-DMA loads in flight (LDS writes that count
      // in vmcnt) the fence waits for the whole tile — here, right behind its issue
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    } else {
      __syncthreads();
    }
    if constexpr (!VW) {
      if (rc_status != 0u) return;  // a step of this token failed before this launch: leave everything as it is (the host retries)
    }
    // (VW: the words are vector loads, the oldest in flight — the test sits in front of the wave's first tile half, below, so that
    //  the gather arithmetic in between still runs in the shadow of the K rows)
  }
  if constexpr (NRC) {
    // (HERE, behind the issue of the first tile's loads — at the top of the kernel the test waited for the step words, a cold
    //  scalar round trip in front of every workgroup's first K rows: +1 us on the C4 step.  Nothing has been written yet.)
    if (a.commit && rc_status != 0u) return;
  }
  bool rc_replay = false;  // this WORKGROUP committed its part of this position's step already: it recomputes and stores nothing
  if constexpr (!(VW && !QKV)) rc_replay = (EML || (NRC && a.commit != nullptr)) && rc_commit == one_pin;
  // ONE: what this thread gathers in the hand-off — up to NOG partial-O granules of the output pairs this workgroup finishes (pair
  // P = r * 64 + d / 2).  Two integer divisions by run-time values (~100 scalar and vector instructions): worked out HERE, with the
  // tile in flight, and pinned — left where they are used, they sat between the merge barrier and the first look at the (m, l)
  // pairs, on the path every workgroup of the head waits for (found in the ISA, r4).
  constexpr int NOG = (RT * 64 + 64 + NW * 64 - 1) / (NW * 64);  // O granules per thread
  int o_off[NOG], o_lds[NOG], o_use[NOG];
  int ppw = 1, pair0 = 0, n_pairs = 0;
  if constexpr (ONE) {
    const int ns_ = a.n_split;
    ppw = (RT * 64 + ns_ - 1) / ns_;      // output pairs finished per workgroup
    const int n_items = ns_ * ppw;        // (split, pair) granules this workgroup reads
    pair0 = split * ppw;
    n_pairs = RT * 64 - pair0;
    n_pairs = n_pairs < 0 ? 0 : (n_pairs > ppw ? ppw : n_pairs);
#pragma unroll
    for (int k = 0; k < NOG; k++) {
      const int item = (int)threadIdx.x + k * NW * 64;
      const int i = item / ppw, qq = item - i * ppw;
      o_use[k] = (item < n_items && qq < n_pairs) ? 1 : 0;
      const int P = o_use[k] ? pair0 + qq : 0;
      o_off[k] = h * kOneOHead + (((P >> 6) * ns_ + (o_use[k] ? i : 0)) * 64 + (P & 63)) * 16;
      o_lds[k] = (i * ppw + qq) * 2;
      asm volatile("" : "+v"(o_off[k]), "+v"(o_lds[k]), "+v"(o_use[k]));
    }
    asm volatile("" : "+s"(ppw), "+s"(pair0), "+s"(n_pairs));
  }

  if constexpr (QKV) {
    // ---- the head's q / k_new / v_new: wave 0 gathers the granules of the head's workgroups (lane = granule), bounded like every
    //      wait of the step, and drops the 16-bit values into LDS; everybody else sleeps at the barrier
    if (wave == 0) {
      const auto qg_rsrc = __builtin_amdgcn_make_buffer_rsrc(a.qkv.gran, 0, (int)a.qkv.gran_bytes, 0x00020000);
      const auto hd_rsrc = __builtin_amdgcn_make_buffer_rsrc(a.one_hdr, 0, 4096, 0x00020000);
      u32x4_t gq[QNG];
      bool q_to = false, q_pf = false;
      WaitBound q_wb;
      for (unsigned spins = 0;; spins++) {
        asm volatile("" ::: "memory");  // every round re-reads memory
#pragma unroll
        for (int k = 0; k < QNG; k++) {
          const int G = lane + 64 * k;
          gq[k] = __builtin_amdgcn_raw_buffer_load_b128(qg_rsrc, h * kOneQHead + (G < QNU ? G : 0) * 16, 0, XL2 ? 16 : kOneAuxCoherent);
        }
        bool ok = true;
#pragma unroll
        for (int k = 0; k < QNG; k++) ok = ok && (lane + 64 * k >= QNU || (gq[k][0] == one_tag && gq[k][2] == one_tag));
        if (__all(ok)) break;
        if ((spins & 7u) == 7u) {  // a workgroup of this head gave up (memory scope): nothing left to wait for
          const unsigned fq = __builtin_amdgcn_raw_buffer_load_b32(hd_rsrc, (kOneFailWord + h) * 4, 0, kOneAuxCoherent);
          if (fq == one_tag) {
            q_pf = true;
            break;
          }
        }
        if (q_wb.expired(spins)) {
          q_to = true;
          break;
        }
        __builtin_amdgcn_s_sleep(2);
      }
      if (lane == 0 && (q_to || q_pf)) {
        if (q_to) {
          __builtin_amdgcn_raw_buffer_store_b32(one_tag, hd_rsrc, (kOneFailWord + h) * 4, 0, kOneAuxCoherent);
          __builtin_amdgcn_raw_buffer_store_b32(1u, hd_rsrc, kOneStatusWordDev * 4, 0, kOneAuxCoherent);
        }
        sm_fail = 1u;  // (reset at the barrier above; read behind the finish's barriers: this head's step commits nothing)
      }
      uint32_t* qw = reinterpret_cast<uint32_t*>(&sm_qkv[0]);
      const bool q_bad = q_to || q_pf;  // (wave-uniform) the step commits nothing: ZEROS instead of whatever half-arrived granules hold —
      // y stays finite (garbage bits may be NaN: the logits behind them, the sampled token and its embedding row would follow)
#pragma unroll
      for (int k = 0; k < QNG; k++) {
        const int G = lane + 64 * k;
        if (G < QNU) *reinterpret_cast<uint2*>(qw + 2 * G) = q_bad ? make_uint2(0u, 0u) : make_uint2(gq[k][1], gq[k][3]);
      }
      qstamp(6);
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    qstamp(7);
#pragma unroll
    for (int j = 0; j < 4; j++)
      if (c < RT) qB[j].raw = *reinterpret_cast<const uint4*>(&sm_qkv[c * D + (4 * j + g) * VEC]);
  }
  float pv_p[U];  // the tile's probabilities (unnormalised), between its two halves
  auto tile_qk = [&](TileRegs& R, const int tbase, const int tbase_next, const bool more_next, auto ti_c) {
    constexpr int TI = decltype(ti_c)::value;  // ONE: which of the wave's tiles (compile time: the scores stay in registers)
    const int row0 = tbase + g * U;
    // ONE: a wave owns exactly one tile (one_shape_ok: rows_per_split == one iteration's rows) — a compile-time fact, so that the
    // next tile's address arithmetic, its loads, the loop's second body and the running-maximum rescale disappear from the code
    if (key_pending) {  // wave-uniform; first tile only
#pragma unroll
      for (int j = 0; j < 3; j++) key_part = key_more[j] < key_part ? key_more[j] : key_part;
      const unsigned long long key = wave_min_u64_uniform(key_part);
      ins_idx = (key == ~0ull) ? -1 : (int)((key & 0xffffffffull) >> 1);
      if (a.abl & 64) ins_idx = -1;
      ins_was_empty = (int)(key & 1ull);
      if constexpr (RC || (NRC && !HYB)) {
        if (rc_insp == one_pin) {  // a retry: the slot the first attempt's insert went to (the key row may hold the next position's keys by now)
          ins_idx = rc_insw >> 1;  // (arithmetic shift: -1 stays -1)
          ins_was_empty = rc_insw & 1;
        } else if (a.commit && split == 0 && threadIdx.x == 0) {  // recorded before anything of the step can be committed
          a.commit[(size_t)h * kRcStride] = ins_idx < 0 ? -1 : ((ins_idx << 1) | ins_was_empty);
          a.commit[(size_t)h * kRcStride + 1] = one_pin;
        }
      }
      key_pending = false;
      if constexpr (HYB) {  // ref: cache.py:896-950 _select_fill_idx, per head — operands requested at the top of the kernel (hy_*)
        hy_flags = __shfl(hy_tabv, hy_pol * 3, CC_WAVE);
        hy_win = __shfl(hy_tabv, hy_pol * 3 + 1, CC_WAVE);
        const int hhs = __shfl(hy_tabv, hy_pol * 3 + 2, CC_WAVE);
        if (a.hyb.token_id && a.hyb.punc_ids) {  // ref: cache.py:975 torch.isin(input_ids, punc_ids)
          const bool f = hy_punc || (lane < a.hyb.n_punc_ids && hy_pidv == hy_tok);
          hy_punc = __any(f) != 0;
        }
        hy_budget = a.g;  // ref: cache.py:912-925
        if (hy_flags & HF_SPECIAL) hy_budget += hy_nsp;
        if (hy_flags & HF_PUNC) hy_budget += hy_npu;
        if (hy_flags & HF_WIN) hy_budget += hy_win;
        if (hy_flags & HF_HH) hy_budget += hhs;
        if constexpr (ONE) {
          if (a.ring_num) {
            const unsigned long long ctr = (unsigned long long)hy_ctr;
            one_rcol = (ctr >> 32) == 0 ? (int)((unsigned)ctr % (unsigned)a.ring_W) : (int)(ctr % (unsigned long long)a.ring_W);
          }
        }
        const int flags = hy_flags, cts = hy_cts;
        hyb_punc = hy_punc;
        const int end_idx = cts < S - 1 ? cts : S - 1;  // :897-899
        hyb_cts = cts;
        if (((flags & HF_PUNC) && hyb_punc) || (flags & HF_FULL)) {  // :905-909
          ins_idx = end_idx;
          hyb_kind = 0;
        } else {
          const int budget = hy_budget;  // :912-925
          if (cts < budget) {  // :927-930 append
            ins_idx = end_idx;
            hyb_kind = 0;
          } else if (flags & (HF_HH | HF_WIN)) {  // :932-946 evict the candidate the previous step scored
            hyb_kind = 1;
            if (ins_idx < 0) ins_idx = 0;  // unreachable with a positive budget (a head over budget has live slots)
          } else {  // :948-950 the token is not kept: it lands in slot S - 1 with the mask untouched
            ins_idx = S - 1;
            hyb_kind = 2;
          }
        }
        if constexpr (HRC) {
          if (a.commit) {
            if (rc_insp == one_pin) {  // a retry: the first attempt's decision (the key row, the count, the step counter may have moved on)
              ins_idx = rc_insw >> 2;
              hyb_kind = rc_insw & 3;
              hyb_cts = rc_cts;
              one_rcol = rc_col;
            } else if (split == 0 && threadIdx.x == 0) {  // recorded before anything of the step can be committed
              a.commit[(size_t)h * kRcStride] = (ins_idx << 2) | hyb_kind;
              a.commit[(size_t)h * kRcStride + 66] = hyb_cts;
              a.commit[(size_t)h * kRcStride + 67] = one_rcol;
              a.commit[(size_t)h * kRcStride + 1] = one_pin;
            }
          }
        }
      }
    }
    if constexpr (DMA && CC_V_FLATLOADS != 0) {  // mword_fixup: see issue_k
      const bool okw = has_mask && row0 + 3 < S && ((reinterpret_cast<uintptr_t>(mh) + (size_t)row0) & 3) == 0;
      if (!has_mask) {
        R.mword = 0x01010101u;
      } else if (__any(!okw)) {  // (wave-uniform; rare: the cache's ragged end or an unaligned mask)
        uint32_t mw = 0;
#pragma unroll
        for (int u = 0; u < U; u++)
          if (!okw && row0 + u < S) mw |= (uint32_t)mh[row0 + u] << (8 * u);
        if (!okw) R.mword = mw;
      }
    }
    // fused insert (cache.py:356-362, 460-490, 754-763) — see the VALU kernel; the K chunk follows the swizzle
    if ((unsigned)(ins_idx - row0) < (unsigned)U) {
      const int um = ins_idx - row0;
      const int kcol = ((c ^ (4 * g + um)) & 15) * VEC;
      Vec16<T> kn, vn;
      if constexpr (!QB && AHEAD) {  // lane c holds chunk c of the new key (requested ahead of the tile); tile row i = 4g + um wants chunk c ^ i
        const int xm = (4 * g + um) & 15;
        kn.raw = make_uint4((uint32_t)__shfl_xor((int)qb_kn.raw.x, xm, 16), (uint32_t)__shfl_xor((int)qb_kn.raw.y, xm, 16),
                            (uint32_t)__shfl_xor((int)qb_kn.raw.z, xm, 16), (uint32_t)__shfl_xor((int)qb_kn.raw.w, xm, 16));
        vn.raw = qb_vn.raw;
      } else if constexpr (QKV) {  // gathered into LDS with q
        kn.raw = *reinterpret_cast<const uint4*>(&sm_qkv[RT * D + kcol]);
        vn.raw = *reinterpret_cast<const uint4*>(&sm_qkv[RT * D + D + c * VEC]);
      } else if constexpr (!QB) {
        kn.load(reinterpret_cast<const T*>(a.k_new) + (size_t)h * D + kcol);
        vn.load(reinterpret_cast<const T*>(a.v_new) + (size_t)h * D + c * VEC);
      }
      const int32_t p_now = ONE ? one_pin : (AHEAD ? p_ins : *a.input_pos);  // (ONE: read in the prologue already)
      // QB: the new row was quantised ahead of the tile (once per step and row), and is attended to through its image like
      // every other row; this lane holds chunk c of K, not the swizzled chunk: the LDS stash below puts it where it belongs
      uint2 knq = make_uint2(0, 0), vnq = make_uint2(0, 0);
      float2 knp = make_float2(0.f, 0.f), vnp = make_float2(0.f, 0.f);
      if constexpr (QB) {  // by the inserting row group only; its operands were requested ahead of the tile, so this runs while the tile is in flight
        knq = quant8_row16<T>(qb_kn.raw, knp);
        vnq = quant8_row16<T>(qb_vn.raw, vnp);
        qb_ins_u = um;
      }
      if constexpr (DMA) {
        // the new rows were requested BEHIND the tile's DMA loads and loads return in order: when kn / vn are here, the stale cache
        // rows have landed in the slabs and may be overwritten
        const int i = 4 * g + um;
        // (l2 requests the new rows AHEAD of the tile — their arrival says nothing about the DMA loads: wait for those explicitly)
        // (QKV: they come from LDS — likewise)
        if constexpr (AHEAD || QKV) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        sm_k[wave][i][c] = kn.raw;                           // (kn is chunk c ^ i: slot c of row i)
        sm_v[wave][i][(c ^ (2 * (i & 7))) & 15] = vn.raw;    // (vn is chunk c)
      }
#pragma unroll
      for (int u = 0; u < U; u++)
        if (u == um) {
          if constexpr (QB) {
            R.kq8[u] = knq; R.kpar[u] = knp;
            R.vq8[u] = vnq; R.vpar[u] = vnp;
          } else if constexpr (!DMA) {
            R.kk[u].raw = kn.raw;
            R.vv[u].raw = vn.raw;
          }
          if (!HYB || hyb_kind != 2) R.mword |= 1u << (8 * u);
        }
      if (HYB && blockIdx.z == 0) {  // ref: cache.py:997-1016 — bookkeeping of the hybrid decision
        const size_t slot = (size_t)h * S + ins_idx;
        *reinterpret_cast<uint4*>(const_cast<T*>(kb) + (size_t)ins_idx * D + kcol) = kn.raw;
        *reinterpret_cast<uint4*>(const_cast<T*>(vh) + (size_t)ins_idx * D) = vn.raw;
        if (c == 0) {
          a.pos[slot] = p_now;  // :1006-1007 every head, dropped tokens included
          if (hyb_kind == 0) a.mask_w[slot] = 1;  // :997-1001 appends only (an evicted slot is live already)
          if constexpr (!ONE) a.hyb.cts_next[h] = hyb_cts + (hyb_kind == 0 ? 1 : 0);  // committed to cache_cts by the combine pass (ONE: by the head's split-0 workgroup, at the end)
          if (hyb_punc && a.hyb.punc_mask) a.hyb.punc_mask[slot] = 1;  // :1011-1016
        }
      } else if (blockIdx.z == 0) {
        const size_t slot = (size_t)h * S + ins_idx;
        if constexpr (QB) {
          *reinterpret_cast<uint2*>(const_cast<uint8_t*>(kqb) + (size_t)ins_idx * D + c * VEC) = knq;
          *reinterpret_cast<uint2*>(const_cast<uint8_t*>(vqh) + (size_t)ins_idx * D) = vnq;
          if (c == 0) *reinterpret_cast<float4*>(a.qparams + slot * 4) = make_float4(knp.x, knp.y, vnp.x, vnp.y);
        } else {
          *reinterpret_cast<uint4*>(const_cast<T*>(kb) + (size_t)ins_idx * D + kcol) = kn.raw;
          *reinterpret_cast<uint4*>(const_cast<T*>(vh) + (size_t)ins_idx * D) = vn.raw;
        }
        if (c == 0) {
          if (a.Hp != 1 || h == 0) a.pos[(a.Hp == 1 ? 0 : (size_t)h * S) + ins_idx] = p_now;
          a.mask_w[slot] = 1;
          if (!ONE && a.num != nullptr) {  // heavy hitter: cache.py:754-763 (ONE: the per-slot pass restarts the history)
            a.num[slot] = 0.0;
            a.denom[slot] = 0;
          }
          // (recoverable hand-off: the count is bumped where the step is committed — a retried insert must not count twice)
          if (ins_was_empty && (a.Hc == a.H || h == 0) && !((EML || NRC) && a.commit)) atomicAdd(&a.cache_cts[a.Hc == a.H ? h : 0], 1);
        }
        if (L2) {  // l2: cache.py:592-593 — the new key's norm (sumsq_canonical_16's order), model dtype
          // the canonical order wants elements c, c + 16, ... of the key in lane c; the lanes of this row group hold chunk c (elements
          // 8c .. 8c + 7): transposed through the wave's V slab (not yet in use: the V tile is stashed after the scores) — eight more
          // loads per wave of the whole launch, ahead of the tile, cost every workgroup's first K rows ~0.4 us
          T* scratch = &sm_l2sc[wave][0];
          *reinterpret_cast<uint4*>(scratch + c * VEC) = qb_kn.raw;
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          __builtin_amdgcn_wave_barrier();
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
          float ss = 0.f;
#pragma unroll
          for (int i = 0; i < D / 16; i++) {
            const float e = ElemTraits<T>::load(scratch, c + 16 * i);
            ss = __fadd_rn(ss, __fmul_rn(e, e));
          }
          __builtin_amdgcn_wave_barrier();  // the V stash must stay behind these reads
#pragma unroll
          for (int off = 8; off > 0; off >>= 1) ss = __fadd_rn(ss, __shfl_xor(ss, off, 16));
          const float nv = ElemTraits<T>::rnd(cc_sqrt_rn(ss));  // every lane of the row group holds the full sum
          l2_nv_lane = nv;
          if (c == 0) {
            ElemTraits<T>::store(reinterpret_cast<T*>(a.key_norm), slot, nv);
            if constexpr (!ONE) a.l2_new[h] = nv;
          }
        }
      }
    }

    // ---- K tile -> wave-private LDS slab -> A operand; S^T = K q^T on the matrix core
#pragma unroll
    for (int u = 0; u < U; u++) {
      if constexpr (DMA) {
        // (already there: the DMA loads deliver into the slab; the compiler's wait for them sits in front of the fragment reads)
      } else
      if constexpr (QB) {  // slot c of tile row i holds chunk c ^ i; the inserted row's lane holds chunk c -> slot c ^ i
        const int i = 4 * g + u;
        sm_k[wave][i][u == qb_ins_u ? ((c ^ i) & 15) : c] = dequant8<T>(R.kq8[u], R.kpar[u]);
      } else {
        sm_k[wave][4 * g + u][c] = R.kk[u].raw;
      }
    }
    qb_ins_u = -1;  // later tiles of this wave hold no inserted row
    if constexpr (ONE) {
      if (a.trace && trA == 0) trA = __builtin_amdgcn_s_memtime();  // this wave's K rows have arrived and sit in its LDS slab
    }
    const uint32_t mcur = R.mword;
    if (more_next) issue_k(R, tbase_next);  // K registers are free again: the next tile streams in behind this tile's math
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    f32x4_t cs = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 4; j++) cs = Mfma16x16x32<T>::mma(sm_k[wave][c][((4 * j + g) ^ c) & 15], qB[j].raw, cs);
    __builtin_amdgcn_wave_barrier();  // the next tile's stores must stay behind these reads

    // ---- lane (g, n = c): scores of rows row0 + 0..3 against query head c
    // ref: attention_utils.py:37 (q@k^T -> dtype, * scale -> dtype), :42-43 (-inf bias where masked)
    float s[U];
#pragma unroll
    for (int t = 0; t < U; t++) {
      const bool valid = (row0 + t < row_end) && (((mcur >> (8 * t)) & 0xffu) != 0);
      const float x = ElemTraits<T>::rnd(ElemTraits<T>::rnd(cs[t]) * a.scale);
      s[t] = valid ? x : -INFINITY;
    }
    if constexpr (ONE) {
#pragma unroll
      for (int t = 0; t < U; t++) s_keep[TI][t] = s[t];
      if (a.trace && trB == 0) trB = __builtin_amdgcn_s_memtime();  // scores of the tile are in registers
    }
    if (!ONE && c < RT && !(a.abl & 1)) {
      const size_t o = (size_t)(q0 + c) * S + row0;
      if (row0 + 3 < row_end && (o & 3) == 0) {  // four consecutive 16-bit scores: one 8-byte store
        T e0, e1, e2, e3;  // s[] already holds values rounded to T: the stores below are exact
        ElemTraits<T>::store(&e0, 0, s[0]);
        ElemTraits<T>::store(&e1, 0, s[1]);
        ElemTraits<T>::store(&e2, 0, s[2]);
        ElemTraits<T>::store(&e3, 0, s[3]);
        *reinterpret_cast<uint2*>(sc_out + o) =
            make_uint2((uint32_t)e0.x | ((uint32_t)e1.x << 16), (uint32_t)e2.x | ((uint32_t)e3.x << 16));
      } else {
#pragma unroll
        for (int t = 0; t < U; t++)
          if (row0 + t < row_end) ElemTraits<T>::store(sc_out, o + t, s[t]);
      }
    }
    // ---- online softmax: ONE (m, l) per lane
    float (&p)[U] = pv_p;
    {
      float mx = fmaxf(fmaxf(s[0], s[1]), fmaxf(s[2], s[3]));
      // ONE running maximum per (wave, head): the four row groups share it (two v_permlane swaps), so their
      // accumulators carry the same scale and merge by plain addition in the epilogue — no exponentials there
      mx = xor_combine<32, true>(xor_combine<16, true>(mx));
      const float m_new = fmaxf(m, mx);
      const float m_use = (m_new == -INFINITY) ? 0.f : m_new;
      const float alpha = ONE1 ? 0.f : fast_exp(m - m_use);  // single tile: m = -inf, l = 0, acc = 0 — nothing to rescale
      m = m_new;
      if constexpr (!ONE1) l *= alpha;
#pragma unroll
      for (int t = 0; t < U; t++) {
        p[t] = fast_exp(s[t] - m_use);
        l += p[t];
      }
      if constexpr (!ONE1) {
        // every accumulator of this lane belongs to head c.  The running maximum settles after the first tiles: while no
        // head of the wave moved it, alpha is exactly 1 and the 32 multiplies — with the accumulators parked in AGPRs, 68 register
        // moves around them: a fifth of the loop's instructions — are skipped (x * 1 == x: bit-identical)
        if (__any(alpha != 1.0f)) {
#pragma unroll
          for (int b = 0; b < D / 16; b++) acc[b] *= alpha;
        }
      }
    }
  };
  auto tile_pv = [&](TileRegs& R, const int tbase_next, const bool more_next) {
    const float (&p)[U] = pv_p;
    // ---- O^T += V^T . P^T on the matrix cores: V tile -> wave-private LDS slab (row major, coalesced), A fragments
    //      back through the transpose read, B = this lane's four probabilities in 16 bit
    if constexpr (!DMA) {
#pragma unroll
      for (int u = 0; u < U; u++)
        sm_v[wave][4 * g + u][(c ^ (2 * ((4 * g + u) & 7))) & 15] = QB ? dequant8<T>(R.vq8[u], R.vpar[u]) : R.vv[u].raw;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (more_next) issue_v(R, tbase_next);  // the V registers are free once the tile sits in LDS: the next tile's rows go out before the P.V products
    {
      const s16x4_t pb = Mfma16x16x16<T>::pack(p);
      const char* vrow = reinterpret_cast<const char*>(&sm_v[wave][tr_row][0]) + tr_half * 8;
#pragma unroll
      for (int b = 0; b < D / 16; b++) {
        const int pos = ((2 * b) ^ tr_sw) | tr_qh;  // chunk 2b + q/2 of the row, swizzled
        const s16x4_t va = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (s16x4_t __attribute__((address_space(3)))*)(vrow + pos * 16));
        acc[b] = Mfma16x16x16<T>::mma(va, pb, acc[b]);
      }
    }
    __builtin_amdgcn_wave_barrier();  // the next tile's V stores must stay behind these reads
    if constexpr (ONE) {
      if (a.trace && trC == 0) trC = __builtin_amdgcn_s_memtime();  // P.V of the tile issued
    }
  };
  auto tile = [&](TileRegs& R, const int tbase, const int tbase_next, const bool more_next, auto ti_c) {
    tile_qk(R, tbase, tbase_next, more_next, ti_c);
    tile_pv(R, tbase_next, more_next);
  };
  // EML: the wave's (m, l) row goes to LDS; the LAST wave to arrive merges the workgroup's pairs and publishes them — behind the
  // scores of the wave's only tile (NT == 1: ahead of its P.V products), or behind the wave's last tile (NT > 1)
  // L2C: the policy's per-slot pass, by the workgroup's last wave alone (see L2C above).  Results: the workgroup's minimum key (into
  // l2c_wk, stored with the commit) and its two granules (stored by the caller, unconditionally).
  unsigned long long l2c_wk = ~0ull;
  u32x4_t l2c_g1 = {0u, 0u, 0u, 0u}, l2c_g2 = {0u, 0u, 0u, 0u};
  auto l2c_pass = [&]() {
    // ---- the norms of the new keys, four kv heads per pass (row group g: head 4 j + g), the inserting row group's arithmetic
    //      operation for operation.  The transposition through LDS is spelled in assembly with its own waits: a plain LDS access
    //      behind pending LDS-DMA loads gets a wait for THOSE from the compiler (it cannot tell these rows from the slabs the V rows
    //      are still landing in).
    auto norm16 = [&](const uint4& kn) -> unsigned {
      unsigned eb[8];
      const unsigned wa = (unsigned)(uintptr_t)(__attribute__((address_space(3))) T*)&sm_l2c4[g][0];
      const u32x4_t kr = {kn.x, kn.y, kn.z, kn.w};
      asm volatile(
          "ds_write_b128 %8, %9\n\t"
          "s_waitcnt lgkmcnt(0)\n\t"
          "ds_read_u16 %0, %10\n\t"
          "ds_read_u16 %1, %10 offset:32\n\t"
          "ds_read_u16 %2, %10 offset:64\n\t"
          "ds_read_u16 %3, %10 offset:96\n\t"
          "ds_read_u16 %4, %10 offset:128\n\t"
          "ds_read_u16 %5, %10 offset:160\n\t"
          "ds_read_u16 %6, %10 offset:192\n\t"
          "ds_read_u16 %7, %10 offset:224\n\t"
          "s_waitcnt lgkmcnt(0)"
          : "=&v"(eb[0]), "=&v"(eb[1]), "=&v"(eb[2]), "=&v"(eb[3]), "=&v"(eb[4]), "=&v"(eb[5]), "=&v"(eb[6]), "=&v"(eb[7])
          : "v"(wa + (unsigned)c * 16u), "v"(kr), "v"(wa + (unsigned)c * 2u)
          : "memory");
      float ss = 0.f;
#pragma unroll
      for (int i = 0; i < D / 16; i++) {
        T et;
        et.x = (uint16_t)eb[i];
        const float e = ElemTraits<T>::load(&et, 0);
        ss = __fadd_rn(ss, __fmul_rn(e, e));
      }
#pragma unroll
      for (int off = 8; off > 0; off >>= 1) ss = __fadd_rn(ss, __shfl_xor(ss, off, 16));
      T nt;
      ElemTraits<T>::store(&nt, 0, cc_sqrt_rn(ss));
      return (unsigned)nt.x;  // (every lane of the row group)
    };
    if (key_pending) {  // a last wave without rows (ragged last split) never entered its tile: the insert slot, as tile_qk derives it
#pragma unroll
      for (int j = 0; j < 3; j++) key_part = key_more[j] < key_part ? key_more[j] : key_part;
      const unsigned long long key = wave_min_u64_uniform(key_part);
      ins_idx = (key == ~0ull) ? -1 : (int)((key & 0xffffffffull) >> 1);
      ins_was_empty = (int)(key & 1ull);
      if (rc_insp == one_pin) {  // a retry: the slot the first attempt's insert went to
        ins_idx = rc_insw >> 1;
        ins_was_empty = rc_insw & 1;
      }
      key_pending = false;
    }
    // the record of position p - 1 of kv head `lane` (bits 0-15: the head's maximum over the slots it keeps at p)
    const unsigned rec16 = (unsigned)((((one_pin + 1) & 1) ? l2c_rc.y : l2c_rc.x) & 0xffffull);
    unsigned gmax16 = 0u, nv_own = 0u;
    const int npass = (a.H + 3) >> 2;
    for (int j = 0; j < npass; j++) {  // (wave-uniform trip count; the first two passes' keys were prefetched)
      const int hj = 4 * j + g;
      uint4 kn = j == 0 ? l2c_kn[0] : l2c_kn[1];
      if (j >= 2) kn = *reinterpret_cast<const uint4*>(reinterpret_cast<const T*>(a.k_new) + (size_t)(hj < a.H ? hj : h) * D + c * VEC);
      const unsigned nv = norm16(kn);
      const unsigned rh = (unsigned)__shfl((int)rec16, hj < a.H ? hj : 0, CC_WAVE);
      unsigned term = hj < a.H ? (nv > rh ? nv : rh) : 0u;  // max(record, new norm): patterns, unsigned order, NaN on top
      gmax16 = term > gmax16 ? term : gmax16;
      if (j == (h >> 2)) nv_own = (unsigned)__builtin_amdgcn_readlane((int)nv, (h & 3) << 4);  // this head's new norm (its row group's lanes all hold it)
    }
    gmax16 = (unsigned)wave_max_uniform((float)gmax16);  // (small integers: exact as floats)
    T mt;
    mt.x = (uint16_t)gmax16;
    const float gm = ElemTraits<T>::load(&mt, 0);  // NaN propagates (torch.max)
    T vt;
    vt.x = (uint16_t)nv_own;
    const float nv_own_f = ElemTraits<T>::load(&vt, 0);
    // ---- the workgroup's slots: ref cache.py:597-605 dtype(max - norm), recent window -> +inf, base rules; and the two largest
    //      norms AFTER this step's insert
    const int32_t p_next = one_pin + 1;
    unsigned long long kmin = ~0ull, k1 = 0ull;
    unsigned t2 = 0u;
#pragma unroll
    for (int k = 0; k < L2S; k++) {
      const int sl = row_begin + k * 64 + lane;
      if (sl < row_end) {
        const bool ins = sl == ins_idx;
        const int32_t ps = ins ? one_pin : l2c_ps[k];
        T et;
        et.x = l2c_nr[k];
        const float kn_eff = ins ? nv_own_f : ElemTraits<T>::load(&et, 0);
        const unsigned kp = ins ? nv_own : (unsigned)l2c_nr[k];
        float scn = ElemTraits<T>::rnd(gm - kn_eff);
        if (ps >= p_next - a.w) scn = INFINITY;
        if (sl < a.g) scn = INFINITY;
        if (ps == -1) scn = -INFINITY;
        const unsigned long long key = make_key(orderable_f32(scn), ((uint32_t)sl << 1) | (uint32_t)(ps == -1));
        kmin = key < kmin ? key : kmin;
        const unsigned long long nk64 = ((unsigned long long)kp << 32) | (unsigned)sl;
        if (nk64 > k1) {
          t2 = (unsigned)(k1 >> 32);
          k1 = nk64;
        } else if (kp > t2) {
          t2 = kp;
        }
      }
    }
    l2c_wk = wave_min_u64_uniform(kmin);
    const unsigned long long K1 = ~wave_min_u64_uniform(~k1);
    const unsigned long long x2 = (k1 == K1) ? (unsigned long long)t2 : (k1 >> 32);
    const unsigned T2w = (unsigned)~wave_min_u64_uniform(~x2);
    l2c_g1 = u32x4_t{one_tag, (unsigned)(l2c_wk >> 32), one_tag, (unsigned)(l2c_wk & 0xffffffffull)};
    l2c_g2 = u32x4_t{one_tag, (unsigned)(K1 >> 32) | (T2w << 16), one_tag, (unsigned)(K1 & 0xffffffffull)};
  };
  auto ml_block = [&]() {
      l = xor_combine<32, false>(xor_combine<16, false>(l));  // the wave's l of head c, in every row group
      if (lane < RT) {                                          // row group 0, column c = head
        sm_wm[wave][lane] = m;
        sm_wl[wave][lane] = l;
      }
      if constexpr (L2C) {
        // (nothing leaves with the pairs: the head's maximum comes from the record, see L2C)
      } else
      if constexpr (L2) {  // the wave's maximum over the norms its slots hold AFTER this step's insert (decided above)
        float kv = -INFINITY;
        if (one_have) kv = (one_slot == ins_idx) ? l2_nv_lane : one_kn_f();
        const bool nn = __any(kv != kv) != 0;
        const float wm = wave_max_uniform(kv);
        if (lane == 0) sm_l2w[wave] = nn ? NAN : wm;
        // the other heads' epoch words have ARRIVED (they were requested ahead of the tile): see above
        asm volatile("" ::"v"(l2_ep[0]), "v"(l2_ep[1]), "v"(l2_ep[2]), "v"(hm_ep) : "memory");
      }
      // No barrier: the waves' K tiles land up to 2 us apart, and a barrier here held every wave's P.V back until the workgroup's
      // LAST K tile had arrived (measured, r3: +0.8 us on the streaming part).  Each wave bumps an LDS counter behind its two
      // stores (release / acquire at workgroup scope); whoever brings it to NW has every wave's row in front of it and publishes.
      unsigned arrived = 0;
      if constexpr (DMA) {
        // (relaxed: a release at workgroup scope waits for the V rows still landing in the slab through the DMA path — they count
        //  in vmcnt.  The LDS unit serves one wave's operations in order: the two stores above are done when the add executes, and
        //  the last arriver's reads below follow its add.)
        //  The add is spelled in assembly: the compiler, which cannot tell the counter from the slabs the DMA loads write, would put
        //  a wait for those in front of any LDS atomic it sees.)
        if (lane == 0) {
          const unsigned cnt_addr = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned*)&sm_mlcnt;
          asm volatile("ds_add_rtn_u32 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=v"(arrived) : "v"(cnt_addr), "v"(1u) : "memory");
        }
      } else
      if (lane == 0) arrived = __hip_atomic_fetch_add(&sm_mlcnt, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_WORKGROUP);
      arrived = (unsigned)__builtin_amdgcn_readfirstlane((int)arrived);
      const bool ml_last = arrived == (unsigned)(NW - 1) && lane < RT;
      float ml_M = 0.f, ml_L = 0.f;
      if (ml_last) {  // the merge of the publish loop below, once per query head (LDS traffic only inside this branch)
        const int r = lane;
        float M = sm_wm[0][r];
#pragma unroll
        for (int w = 1; w < NW; w++) M = fmaxf(M, sm_wm[w][r]);
        const float Mu = (M == -INFINITY) ? 0.f : M;
        float L = 0.f;
#pragma unroll
        for (int w = 0; w < NW; w++) {
          const float f = fast_exp(sm_wm[w][r] - Mu);
          if constexpr (CC_V_WFACT != 0) sm_wf[w][r] = f;  // (read by the partial-O publish, behind the merge barrier)
          L = fmaf(sm_wl[w][r], f, L);
        }
        ml_M = M;
        ml_L = L;
      }
      {
        // The store is UNCONDITIONAL — every lane of every wave executes it; all but the publisher's RT lanes aim past the end of
        // the buffer, where the hardware drops the write.  Inside the branch above it would make the compiler's in-order wait
        // for the V rows (older loads) a wait for this write-through store's acknowledgement too (its bookkeeping merges "store
        // issued" with "not issued" at the join): a memory round trip in front of the P.V products of the workgroup's last wave.
        const u32x4_t mg = {one_tag, __float_as_uint(ml_M), one_tag, __float_as_uint(ml_L)};
        const auto ml_rsrc_e = __builtin_amdgcn_make_buffer_rsrc(a.one_ml, 0, (int)a.one_ml_bytes, 0x00020000);
        const int off = ml_last ? h * kOneMlHead + (split * RT + lane) * 16 : 0x7ffffff0;
        __builtin_amdgcn_raw_buffer_store_b128(mg, ml_rsrc_e, off, 0, XL2 ? 0 : kOneAuxCoherent);
        if constexpr (L2C) {
        } else
        if constexpr (L2) {  // l2: the workgroup's norm maximum leaves with the pairs (same unconditional form; lane RT of the publisher)
          float wm = -INFINITY;
          bool nn = false;
          if (arrived == (unsigned)(NW - 1) && lane == RT) {
            float lw[NW];
            if constexpr (DMA && (NW == 4 || NW == 8)) {
              // (read in assembly: the compiler cannot tell this row from the slabs the DMA loads write and guards a plain read
              //  with a wait for the V tile — the workgroup's norm maximum, which every workgroup of the launch waits for, then
              //  left when its publisher's V rows had landed: found in the ISA, r4)
              lds_read_row_nowait<NW>(sm_l2w, lw);
            } else {
#pragma unroll
              for (int w = 0; w < NW; w++) lw[w] = sm_l2w[w];
            }
            wm = lw[0];
            nn = wm != wm;
#pragma unroll
            for (int w = 1; w < NW; w++) {
              nn |= lw[w] != lw[w];
              wm = fmaxf(wm, lw[w]);
            }
          }
          const u32x4_t ng = {one_tag, __float_as_uint(wm), one_tag, nn ? 1u : 0u};
          const int noff = (arrived == (unsigned)(NW - 1) && lane == RT) ? kOneMaxHeads * kOneMlHead + h * kOneNmHead + split * 16 : 0x7ffffff0;
          __builtin_amdgcn_raw_buffer_store_b128(ng, ml_rsrc_e, noff, 0, L2X ? 0 : kOneAuxCoherent);  // (L2X: gathered inside the XCD)
        }
      }
      };
  if constexpr (ONE && NT > 1) {
    // the wave's tiles, unrolled: tile TI keeps its scores in s_keep[TI] for the finish (at most NT tiles: one_shape_ok)
    auto run_tiles = [&](auto self, auto ti_c) -> void {
      constexpr int TI = decltype(ti_c)::value;
      if constexpr (TI < NT) {
        const int tb = base + TI * (NW * RPW * U);
        if (tb < row_end) {
          tile(tregs[0], tb, tb + NW * RPW * U, TI + 1 < NT && tb + NW * RPW * U < row_end, ti_c);
          self(self, IntC<TI + 1>{});
        }
      }
    };
    if (more) run_tiles(run_tiles, IntC<0>{});
    if constexpr (EML) ml_block();  // (a wave without rows arrives with (-inf, 0))
  } else if constexpr (EML) {
    // single tile, early (m, l): scores -> [the workgroup's (m, l) pairs leave] -> P.V.  A wave without rows (ragged last split)
    // skips the tile halves but not the arrival counter: its pair is (-inf, 0).
    if constexpr (VW && !QKV) {
      __builtin_amdgcn_sched_barrier(0);  // (the scheduler otherwise hoists the first of these waits up to the barrier)
      words_to_sgpr();
      if (rc_status != 0u) return;  // a step of this token failed before this launch: leave everything as it is (the host retries)
      rc_replay = rc_commit == one_pin;
    }
    if (more) tile_qk(tregs[0], base, base, false, IntC<0>{});
    qstamp(8);
    ml_block();
    if constexpr (L2C) {
      // the policy's whole per-slot pass, by the workgroup's last wave, while its V rows land; its two granules leave HERE (stores
      // unconditional: the other waves aim past the buffer's end, where the hardware drops the write)
      if (wave == NW - 1) l2c_pass();
      const auto nm_rsrc_e = __builtin_amdgcn_make_buffer_rsrc(a.one_ml, 0, (int)a.one_ml_bytes, 0x00020000);
      const int goff = (wave == NW - 1 && lane == 0) ? kOneMaxHeads * kOneMlHead + h * kOneNmHead + split * 32 : 0x7ffffff0;
      __builtin_amdgcn_raw_buffer_store_b128(l2c_g1, nm_rsrc_e, goff, 0, XL2 ? 0 : kOneAuxCoherent);
      __builtin_amdgcn_raw_buffer_store_b128(l2c_g2, nm_rsrc_e, (wave == NW - 1 && lane == 0) ? goff + 16 : 0x7ffffff0, 0, XL2 ? 0 : kOneAuxCoherent);
    }
    if (more) tile_pv(tregs[0], base, false);
    qstamp(9);
  } else {
    while (more) {
      const int base_next = base + NSUB * NW * RPW * U;
      const bool more_next = ONE ? false : base_next < row_end;
#pragma unroll
      for (int sub = 0; sub < NSUB; sub++)
        tile(tregs[sub], base + sub * NW * RPW * U, base_next + sub * NW * RPW * U, more_next, IntC<0>{});
      base = base_next;
      more = more_next;
    }
  }
  if constexpr (L2 && ONE && !EML) {
    float kv = -INFINITY;
    if (one_have) kv = (one_slot == ins_idx) ? l2_nv_lane : one_kn_f();
    const bool nn = __any(kv != kv) != 0;
    const float wm = wave_max_uniform(kv);
    if (lane == 0) sm_l2w[wave] = nn ? NAN : wm;
  }
  if (L2 && !ONE && l2_here) {  // publish this wave's maximum over the norms that survive this step
    if (key_pending) {  // a wave without rows never entered the loop
#pragma unroll
      for (int j = 0; j < 3; j++) key_part = key_more[j] < key_part ? key_more[j] : key_part;
      const unsigned long long key = wave_min_u64_uniform(key_part);
      ins_idx = (key == ~0ull) ? -1 : (int)((key & 0xffffffffull) >> 1);
    }
    float v = (kn_row0 == ins_idx) ? -INFINITY : kn_first;  // the evicted slot's old norm is gone
    bool kn_nan = v != v;
    for (int r = kn_row0 + NW * 64; r < row_end; r += NW * 64) {  // only beyond 256 slots per workgroup
      const float x = ElemTraits<T>::load(reinterpret_cast<const T*>(a.key_norm) + (size_t)h * S, r);
      if (r != ins_idx) {
        kn_nan |= (x != x);
        v = fmaxf(v, x);
      }
    }
    v = wave_max_uniform(v);
    const bool nn = __any(kn_nan) != 0;
    if (lane == 0) a.l2_pmax[((size_t)h * a.n_split + split) * NW + wave] = nn ? NAN : v;
  }

  if (a.abl & 2) {  // measurement only
    float x = l + m;
#pragma unroll
    for (int b = 0; b < D / 16; b++) x += acc[b][0] + acc[b][1] + acc[b][2] + acc[b][3];
    if (x == 1.2345f) a.part_ml[0] = x;
    return;
  }
  // ---- Epilogue.  The P.V accumulators already cover all 16 rows of the wave's tiles (one running maximum per
  //      (wave, head)), so there is nothing to merge inside a wave: lanes n < RT drop their [128] partial into LDS
  //      (8 x ds_write_b128) and the four waves of the workgroup meet there.
  // The waves' partial O rows live IN their K slabs (r5): a wave's slab (16 rows x 256 B = 4 KiB) is read by that wave alone, for the
  // last time by the Q.K products of its last tile — fences and this wave's own program order separate those reads from the stores
  // below — and RT x 128 floats fit it exactly at RT = 8.  16 KiB of LDS less per 8-wave workgroup: 69.8 instead of 85.8 KB (two wide
  // workgroups would fit a CU; the plan does not use that — make_plan_w has the measurement — but a co-tenant's LDS has more room).
  static_assert(RT * D * sizeof(float) <= sizeof(sm_k[0]), "a wave's partial O rows must fit its K slab");
  typedef float __attribute__((may_alias)) wacc_f32;
  auto sm_wacc = [&](int w, int r, int d) -> wacc_f32& { return reinterpret_cast<wacc_f32*>(&sm_k[w][0][0])[r * D + d]; };
  {
    if constexpr (!EML) {  // (EML: done between the tile's halves)
      l = xor_combine<32, false>(xor_combine<16, false>(l));  // the wave's l of head c, in every row group
      if (lane < RT) {                                          // row group 0, column c = head
        sm_wm[wave][lane] = m;
        sm_wl[wave][lane] = l;
      }
    }
    if (c < RT) {
#pragma unroll
      for (int b = 0; b < D / 16; b++)
        *reinterpret_cast<float4 __attribute__((may_alias))*>(&sm_wacc(wave, c, 16 * b + 4 * g)) = make_float4(acc[b][0], acc[b][1], acc[b][2], acc[b][3]);
    }
  }
  if constexpr (L2X) {
    // L2X, level one — EARLY: the workgroups' norm maxima left with their (m, l) pairs, behind their scores; the last wave of the
    // head's split-0 workgroup (the first blocks of the grid: among the first to start) gathers them through the XCD's L2 right here,
    // ahead of the merge barrier, and publishes the head's maximum through memory for the other kv heads.  (After the merge barrier
    // and the (m, l) gather it was 1.5 us later, and every workgroup of the launch ended behind it; every workgroup of the head
    // storing the same granule was far worse: 32 write-through stores per address, 14.4 us per step.)
    if (split == 0 && wave == NW - 1) {
      const auto ml_rsrc_p = __builtin_amdgcn_make_buffer_rsrc(a.one_ml, 0, (int)a.one_ml_bytes, 0x00020000);
      const int off1 = kOneMaxHeads * kOneMlHead + h * kOneNmHead + (lane < a.n_split ? lane : 0) * 16;
      u32x4_t nmx = {0u, 0u, 0u, 0u};
      bool got = false;
      WaitBound nm_wb;
      for (unsigned spins = 0; !nm_wb.expired(spins); spins++) {
        asm volatile("" ::: "memory");  // every round re-reads memory
        nmx = __builtin_amdgcn_raw_buffer_load_b128(ml_rsrc_p, off1, 0, 16 /* sc1: the XCD's L2 */);
        if (__all(lane >= a.n_split || (nmx[0] == one_tag && nmx[2] == one_tag))) {
          got = true;
          break;
        }
        __builtin_amdgcn_s_sleep(1);
      }
      if (got) {  // (not: nothing is published, and every workgroup of the launch times out on level two — reported like any hand-off timeout)
        const float v = lane < a.n_split ? __uint_as_float(nmx[1]) : -INFINITY;
        const bool hn = __any(lane < a.n_split && (nmx[3] != 0u || v != v)) != 0;
        const float hmx = wave_max_uniform(v);
        const u32x4_t hg = {one_tag, __float_as_uint(hmx), one_tag, hn ? 1u : 0u};
        __builtin_amdgcn_raw_buffer_store_b128(hg, ml_rsrc_p, lane == 0 ? kOneMaxHeads * (kOneMlHead + kOneNmHead) + h * 16 : 0x7ffffff0, 0, kOneAuxCoherent);
      }
    }
  }
  __syncthreads();
  qstamp(10);
  if constexpr (ONE) {
    // ============================================================================================================
    // Single-launch layer step: what the combine launch did, done here behind an in-launch hand-off.
    //   publish   the workgroup's (m, l, O[RT][128]) partial as 16-byte granules {tag, x, tag, y}: write-through
    //             (sc0 sc1) stores; each 8-byte half validates itself, so no flag, no fence and no store ordering;
    //   gather    wave r collects the n_split (m, l) pairs of query head r (lane = split) and every thread its share
    //             of the O granules of the output columns this workgroup finishes — coherent loads, re-read (s_sleep
    //             between rounds) until every tag is this launch's.  Tags only ever grow (the epoch words live in the workspace and are bumped once
    //             per launch and head), so a granule left by any earlier launch, layer or shape can never match;
    //   finish    the final (M, L) per head in the combine kernel's exact order; then this workgroup's 64 slots —
    //             probabilities from the scores still in registers, group mean, history update, next-eviction key —
    //             and its columns of y.
    // Probabilities, history and keys repeat decode_attn_combine_kernel's operations in its order: bit-identical to the
    // two-launch step (y sums its partials in a different fixed order: equal up to fp32 rounding).  Needs all
    // n_split * H workgroups co-resident (the launcher checks); every wait is bounded and reports through the
    // timeout word should that ever not hold.
    const int ns = a.n_split;
    const unsigned tag = one_tag;
    // XL2: a kv head's own granules never leave its XCD — plain stores (they stay in the L2), sc1 polls (past the L1, served by the
    // L2).  The l2 policy's norm maxima cross kv heads, hence XCDs: they keep the write-through / memory-scope form.
    constexpr int kGatherAux = XL2 ? 16 : kOneAuxCoherent;
    constexpr int kPublishAux = XL2 ? 0 : kOneAuxCoherent;
    if constexpr (L2 && !EML) {
      // the epoch words of the other heads (requested behind the tile's loads) must have ARRIVED before this workgroup publishes:
      // whoever bumps a word does so only after every workgroup of the launch has published
      asm volatile("" ::"v"(l2_ep[0]), "v"(l2_ep[1]), "v"(l2_ep[2]) : "memory");
    }
    if constexpr (HYB) {
      // the step counter, num_punc / num_special and the head's count are overwritten at the END of this launch by whoever sees
      // that every workgroup has published: their prologue loads must have COMPLETED (not merely been issued) before this
      // workgroup publishes — the values are pinned into registers here
      asm volatile("" ::"s"(one_rcol), "s"(hy_budget), "s"(hy_cts), "s"((int)hy_punc), "s"(hy_nsp), "s"(hy_npu) : "memory");
    }
    if (a.trace) {
      tr1 = __builtin_amdgcn_s_memtime();
      rt1 = __builtin_amdgcn_s_memrealtime();
    }
    __shared__ float sm_w1[RT][64];  // exp(m_i - M_r)
    __shared__ float sm_M1[RT], sm_L1[RT];
    __shared__ __attribute__((aligned(16))) float sm_o1[2 * (RT * 64 + 64)];  // [split][pair of this workgroup][2]: raw partial O
    const auto ml_rsrc = __builtin_amdgcn_make_buffer_rsrc(a.one_ml, 0, (int)a.one_ml_bytes, 0x00020000);
    const auto o_rsrc = __builtin_amdgcn_make_buffer_rsrc(a.one_o, 0, (int)a.one_o_bytes, 0x00020000);
    constexpr int MLN = (RT + NW - 1) / NW;            // (m, l) granules per thread: wave w collects heads w, w + NW, ...
    int ml_off[MLN];
    // the waves that fold a head's (m, l) pairs: the first RT — or (MLW) the LAST RT of a workgroup with at least 2 RT waves: the first
    // RT waves merge and publish the partial O meanwhile (thread t: outputs 2t, 2t + 1), the others would idle
    constexpr int ML_W0 = (CC_V_MLW != 0 && EML && NW >= 2 * RT) ? NW - RT : 0;
    const int ml_w = wave - ML_W0;  // this wave's first head (negative: none)
#pragma unroll
    for (int k = 0; k < MLN; k++) ml_off[k] = h * kOneMlHead + ((lane < ns ? lane : 0) * RT + (ml_w >= 0 && ml_w + k * NW < RT ? ml_w + k * NW : 0)) * 16;
    u32x4_t mlq[MLN];
    const int nm_base = kOneMaxHeads * kOneMlHead;  // l2: the norm-maximum granules sit behind the (m, l) regions of all heads
    constexpr int NLG = (L2 && !L2X && !L2C) ? 3 : 0;  // l2, one-level exchange: norm-maximum granules per thread (H * n_split <= 768 workgroups are ever co-resident)
    int nm_off[NLG > 0 ? NLG : 1];
    bool nm_use[NLG > 0 ? NLG : 1];
    unsigned nm_tag[NLG > 0 ? NLG : 1];  // a granule of kv head h' carries h''s tag (its epoch word was read in the prologue: l2_ep)
#pragma unroll
    for (int k = 0; k < NLG; k++) {
      const int e = (int)threadIdx.x + k * NW * 64;
      nm_use[k] = e < a.H * ns;
      const int hh = nm_use[k] ? e / ns : 0, ss = nm_use[k] ? e - hh * ns : 0;
      nm_off[k] = nm_base + hh * kOneNmHead + ss * 16;
      nm_tag[k] = l2_ep[k] + 1u;
    }
    u32x4_t nq[NLG > 0 ? NLG : 1];
    u32x4_t hmq = {0u, 0u, 0u, 0u};  // L2X: the kv heads' maxima (lane = kv head)
    const int hm_base = kOneMaxHeads * (kOneMlHead + kOneNmHead);
    if constexpr (EML) {
      // the (m, l) pairs left behind the scores, long ago: their first round of loads goes out AHEAD of this workgroup's partial-O
      // stores (loads return in order: behind the stores they would also wait for the stores' acknowledgements, a round trip)
      if (ml_w >= 0 && ml_w < RT) {
#pragma unroll
        for (int k = 0; k < MLN; k++) mlq[k] = __builtin_amdgcn_raw_buffer_load_b128(ml_rsrc, ml_off[k], 0, kGatherAux);
      }

      // l2: every workgroup's norm maximum (they left with the pairs), gathered by every thread of every workgroup; a thread
      // without a granule aims past the buffer's end (no request, zeros back): no branch around the loads
#pragma unroll
      for (int k = 0; k < NLG; k++) nq[k] = __builtin_amdgcn_raw_buffer_load_b128(ml_rsrc, nm_use[k] ? nm_off[k] : 0x7ffffff0, 0, kOneAuxCoherent);
      __builtin_amdgcn_sched_barrier(0);
    }
    // ---- publish: thread t merges output columns 2t, 2t + 1 of the workgroup's partial (the arithmetic of the
    //      two-launch epilogue below) and stores them as one granule
    for (int o2 = (int)threadIdx.x * 2; o2 < RT * D; o2 += 2 * NW * 64) {
      {
        const int r = o2 / D, d = o2 - r * D;
        float M = 0.f, L = 0.f, O0 = 0.f, O1 = 0.f;
        if constexpr (EML && CC_V_WFACT != 0) {
#pragma unroll
          for (int w = 0; w < NW; w++) {  // fixed order: deterministic (the factors: ml_block, the (m, l) merge)
            const float f = sm_wf[w][r];
            O0 = fmaf(sm_wacc(w, r, d), f, O0);
            O1 = fmaf(sm_wacc(w, r, d + 1), f, O1);
          }
        } else {
        M = sm_wm[0][r];
#pragma unroll
        for (int w = 1; w < NW; w++) M = fmaxf(M, sm_wm[w][r]);
        const float Mu = (M == -INFINITY) ? 0.f : M;
#pragma unroll
        for (int w = 0; w < NW; w++) {  // fixed order: deterministic
          const float f = fast_exp(sm_wm[w][r] - Mu);
          L = fmaf(sm_wl[w][r], f, L);
          O0 = fmaf(sm_wacc(w, r, d), f, O0);
          O1 = fmaf(sm_wacc(w, r, d + 1), f, O1);
        }
        }
        const u32x4_t og = {tag, __float_as_uint(O0), tag, __float_as_uint(O1)};
        __builtin_amdgcn_raw_buffer_store_b128(og, o_rsrc, h * kOneOHead + ((r * ns + split) * 64 + (d >> 1)) * 16, 0, kPublishAux);
        if (!EML && d == 0) {
          const u32x4_t mg = {tag, __float_as_uint(M), tag, __float_as_uint(L)};
          __builtin_amdgcn_raw_buffer_store_b128(mg, ml_rsrc, h * kOneMlHead + (split * RT + r) * 16, 0, kPublishAux);
        }
      }
    }
    // l2: the workgroup's norm maximum travels the same way — one granule {tag, max, tag, nan} per workgroup in the upper half
    // of the (m, l) region; every workgroup of EVERY kv head gathers all of them (cache.py:602 takes the maximum over all heads)
    if constexpr (L2 && !EML) {
      if (threadIdx.x == NW * 64 - 1) {
        float wm = sm_l2w[0];
        bool nn = wm != wm;
#pragma unroll
        for (int w = 1; w < NW; w++) {
          nn |= sm_l2w[w] != sm_l2w[w];
          wm = fmaxf(wm, sm_l2w[w]);
        }
        const u32x4_t ng = {tag, __float_as_uint(wm), tag, nn ? 1u : 0u};
        __builtin_amdgcn_raw_buffer_store_b128(ng, ml_rsrc, nm_base + h * kOneNmHead + split * 16, 0, kOneAuxCoherent);
      }
    }
    if constexpr (HYB) hyb_load_state();  // hybrid: the slots' ring state / positions / protection masks arrive during the hand-off
    else if constexpr (NT > 1) load_slot_state();  // several tiles per wave: the slots' history / positions arrive during the hand-off
    if (a.trace) tr2 = __builtin_amdgcn_s_memtime();
    // ---- what this thread gathers: the (m, l) granule of (head = wave, split = lane) and up to two O granules
    // (ppw, pair0, n_pairs, o_off / o_lds / o_use: worked out in the prologue, while the tile was in flight)
    bool timed_out = false;
    bool hrc_peer_failed = false, hrc_failed = false;  // hybrid, recoverable form: a peer gave up / this workgroup commits nothing
    // The first poll waits until this wave's OWN publish stores are acknowledged (vmcnt counts stores on this chip).  Polls issued
    // right behind the write-through stores cost 0.8 us at S = 4096 (11.1 vs 10.3 us; found by accident: a never-taken measurement
    // branch with loads of its own made the compiler put this wait at the join): the stragglers' K/V rows and everybody's granules
    // then queue behind 6 MB of polls per round that cannot succeed yet.
    if constexpr (!EML) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (EML: the first rounds of both gathers are not issued behind these stores)
    u32x4_t oq[NOG];
    // one round of loads of each kind (coherent: they bypass the L1 and stale L2 lines), and whether every granule of the round
    // carries this launch's tag
    auto load_ml = [&]() {
#pragma unroll
      for (int k = 0; k < MLN; k++) mlq[k] = __builtin_amdgcn_raw_buffer_load_b128(ml_rsrc, ml_off[k], 0, kGatherAux);
    };
    unsigned failq = 0;  // EML: the head's fail word, read with every round of the partial-O gather (recoverable hand-off)
    const auto hdr_rsrc = __builtin_amdgcn_make_buffer_rsrc(a.one_hdr, 0, 4096, 0x00020000);
    auto load_o = [&]() {
#pragma unroll
      for (int k = 0; k < NOG; k++) oq[k] = __builtin_amdgcn_raw_buffer_load_b128(o_rsrc, o_off[k], 0, kGatherAux);
      if constexpr (RC) failq = __builtin_amdgcn_raw_buffer_load_b32(hdr_rsrc, (kOneFailWord + h) * 4, 0, kOneAuxCoherent);
    };
    // a wave that gives up: the head's fail word (this launch's tag) and the workspace's status word, write-through
    auto give_up = [&]() {
      if (lane == 0) {
        __builtin_amdgcn_raw_buffer_store_b32(tag, hdr_rsrc, (kOneFailWord + h) * 4, 0, kOneAuxCoherent);
        __builtin_amdgcn_raw_buffer_store_b32(1u, hdr_rsrc, kOneStatusWordDev * 4, 0, kOneAuxCoherent);
        sm_fail = 1u;
      }
    };
    auto ok_ml = [&]() {
      bool ok = true;
#pragma unroll
      for (int k = 0; k < MLN; k++) ok = ok && mlq[k][0] == tag && mlq[k][2] == tag;
      return ok;
    };
    auto ok_o = [&]() {
      bool ok = true;
#pragma unroll
      for (int k = 0; k < NOG; k++) ok = ok && (!o_use[k] || (oq[k][0] == tag && oq[k][2] == tag));
      return ok;
    };
    // ---- final (M, L) of query heads r = wave, wave + NW, ...: decode_attn_combine_kernel's order (lane = split, n_split <= 64)
    auto final_ml = [&]() {
#pragma unroll
      for (int k = 0; k < MLN; k++) {
        const int rr = ml_w + k * NW;
        if (rr >= 0 && rr < RT) {
          const float mi = lane < ns ? __uint_as_float(mlq[k][1]) : -INFINITY;
          const float M = wave_max_uniform(mi);
          const float Mu = (M == -INFINITY) ? 0.f : M;
          float L = 0.f;
          if (lane < ns) {
            const float wgt = exp_nonpos(mi - Mu);
            sm_w1[rr][lane] = wgt;
            L = __uint_as_float(mlq[k][3]) * wgt;
          }
          L = wave_sum_uniform(L);
          if (lane == 0) {
            sm_M1[rr] = Mu;
            sm_L1[rr] = L;
          }
        }
      }
    };
    auto stash_o = [&]() {
#pragma unroll
      for (int k = 0; k < NOG; k++)
        if (o_use[k]) *reinterpret_cast<float2*>(&sm_o1[o_lds[k]]) = make_float2(__uint_as_float(oq[k][1]), __uint_as_float(oq[k][3]));
    };
    // ---- y: per output column, G1 strided chains over the splits.  The G1 chains of an output sit in G1 adjacent lanes: folded by
    //      DPP butterflies and stored at once — no partial sums through LDS, no second barrier (tasks go to the LAST threads first).
    auto y_fold = [&]() {
      const int n_out = 2 * n_pairs;
      const int sh = ns >= 8 ? 3 : (ns >= 4 ? 2 : (ns >= 2 ? 1 : 0)), G1 = 1 << sh;
      for (int task = NW * 64 - 1 - (int)threadIdx.x; task < (n_out << sh); task += NW * 64) {
        const int ol = task >> sh, gg = task & (G1 - 1);
        const int to = 2 * pair0 + ol, r = to / D;
        float part = 0.f;
        for (int i = gg; i < ns; i += G1) part = fmaf(sm_o1[i * ppw * 2 + ol], sm_w1[r][i], part);
        part = seg_sum(part, G1);  // whole groups of G1 lanes are in or out of this loop together
        if (gg == 0) ElemTraits<T>::store(reinterpret_cast<T*>(a.y), (size_t)(h * RT + r) * D + (to - r * D), part / sm_L1[r]);
      }
    };
    __shared__ float sm_l2g[NW];  // l2: per-wave fold of the gathered norm maxima (NaN propagates: torch.max)
    unsigned long long trD = 0, trE = 0;
    if constexpr (EML) {
      // ---- the (m, l) pairs left behind the scores: most of them are there by now.  Only the waves that fold a head poll.
      const bool ml_mine = ml_w >= 0 && ml_w < RT;  // (MLN == 1 whenever RT <= NW; with RT = 8 on four waves every wave folds two heads)
      // Round 0 (issued ahead of the partial-O stores) is examined in STRAIGHT-LINE code: the compiler then waits for exactly
      // these loads — the oldest in flight — and not for the acknowledgements of the stores behind them (at a loop header its
      // in-order wait counts merge with the back edge's and become a wait for everything)
      auto ok_nm = [&]() {
        bool ok = true;
#pragma unroll
        for (int k = 0; k < NLG; k++) ok = ok && (!nm_use[k] || (nq[k][0] == nm_tag[k] && nq[k][2] == nm_tag[k]));
        return ok;
      };
      bool ml_ok = (!ml_mine || __all(ok_ml())) && (!L2 || __all(ok_nm()));
      bool peer_failed = false;
      WaitBound ml_wb;
      for (unsigned spins = 0; !ml_ok; spins++) {
        if (ml_wb.expired(spins)) {
          timed_out = true;
          break;
        }
        __builtin_amdgcn_s_sleep(1);
        asm volatile("" ::: "memory");  // every round re-reads memory
        if constexpr (RC) {  // a workgroup of this head gave up (memory scope: it may sit on another XCD): nothing left to wait for
          failq = __builtin_amdgcn_raw_buffer_load_b32(hdr_rsrc, (kOneFailWord + h) * 4, 0, kOneAuxCoherent);
          if (failq == tag) {
            peer_failed = true;
            break;
          }
        }
        if (ml_mine) load_ml();
#pragma unroll
        for (int k = 0; k < NLG; k++)
          if (nm_use[k]) nq[k] = __builtin_amdgcn_raw_buffer_load_b128(ml_rsrc, nm_off[k], 0, kOneAuxCoherent);
        ml_ok = (!ml_mine || __all(ok_ml())) && (!L2 || __all(ok_nm()));
      }
      if (timed_out) give_up();
      else if (peer_failed && lane == 0) sm_fail = 1u;
      if (a.trace) tr4 = __builtin_amdgcn_s_memtime();
      if (ml_mine) final_ml();
      if constexpr (L2C) {
        // (the kv heads' terms sit in sm_l2m since the merge barrier: folded where the keys are scored, below)
      } else
      if constexpr (L2) {  // the wave's fold of the gathered norm maxima (NaN propagates: torch.max)
        float gm = -INFINITY;
        bool gn = false;
#pragma unroll
        for (int k = 0; k < NLG; k++)
          if (nm_use[k]) {
            const float v = __uint_as_float(nq[k][1]);
            gn |= nq[k][3] != 0u || v != v;
            gm = fmaxf(gm, v);
          }
        const bool nn = __any(gn) != 0;
        const float wm = wave_max_uniform(gm);
        if (lane == 0) sm_l2g[wave] = nn ? NAN : wm;
      }
      __syncthreads();
      if (a.trace) trD = __builtin_amdgcn_s_memtime();
    } else {
      WaitBound mo_wb;
      for (unsigned spins = 0;; spins++) {
        asm volatile("" ::: "memory");  // every round re-reads memory
        load_ml();
        load_o();
#pragma unroll
        for (int k = 0; k < NLG; k++) nq[k] = __builtin_amdgcn_raw_buffer_load_b128(ml_rsrc, nm_off[k], 0, kOneAuxCoherent);
        bool ok = true;
#pragma unroll
        for (int k = 0; k < NLG; k++) ok = ok && (!nm_use[k] || (nq[k][0] == nm_tag[k] && nq[k][2] == nm_tag[k]));
        ok = ok && ok_ml() && ok_o();
        if (__all(ok)) break;
        if constexpr (NRC) {
          // the head's fail word, with one round in eight (it travels through memory: a round that waits for it lasts longer —
          // +1 us on the C4 step when it was read with every round) — a workgroup of this head gave up: nothing left to wait for
          if (a.commit && (spins & 7u) == 7u) {
            failq = __builtin_amdgcn_raw_buffer_load_b32(hdr_rsrc, (kOneFailWord + h) * 4, 0, kOneAuxCoherent);
            if (failq == tag) {
              hrc_peer_failed = true;
              break;
            }
          }
        }
        if (mo_wb.expired(spins)) {
          timed_out = true;
          break;
        }
        __builtin_amdgcn_s_sleep(1);
      }
      if (a.trace) tr4 = __builtin_amdgcn_s_memtime();
      if (NRC && a.commit) {
        if (timed_out) give_up();
        else if (hrc_peer_failed && lane == 0) sm_fail = 1u;  // (every workgroup commits or repeats ITS part: per-workgroup commit words)
      } else if (timed_out && lane == 0) {
        a.one_hdr[kOneStatusWordDev] = 1u;  // this launch's results are invalid; the host reads the word
      }
      final_ml();
      stash_o();
      if constexpr (L2) {
        float gm = -INFINITY;
        bool gn = false;
#pragma unroll
        for (int k = 0; k < NLG; k++)
          if (nm_use[k]) {
            const float v = __uint_as_float(nq[k][1]);
            gn |= nq[k][3] != 0u || v != v;
            gm = fmaxf(gm, v);
          }
        const bool nn = __any(gn) != 0;
        const float wm = wave_max_uniform(gm);
        if (lane == 0) sm_l2g[wave] = nn ? NAN : wm;
      }
      __syncthreads();
      if (a.trace) trD = __builtin_amdgcn_s_memtime();
      if constexpr (NRC) hrc_failed = a.commit != nullptr && sm_fail != 0u;  // (workgroup-uniform behind the barrier)
      if (!hrc_failed) y_fold();
    }
    // hybrid, recoverable form: nothing of the step is stored by a workgroup that failed or that committed this position before
    const bool hrc_skip = NRC && a.commit != nullptr && (hrc_failed || rc_replay);
    if constexpr (EML) {
      // the first round of the partial-O gather goes out HERE and flies while the per-slot pass runs: what is left behind the last O
      // granule of the head is the y fold
      asm volatile("" ::: "memory");
      load_o();
      __builtin_amdgcn_sched_barrier(0);
    }
    // ---- this workgroup's slots.  ref: attention_utils.py:52 softmax -> model dtype; model.py:416-418 group mean -> model
    //      dtype; cache.py:716-722 history; cache.py:727-749 the next position's eviction score
    unsigned long long my_key = ~0ull;
    double def_num[NT];  // EML: this lane's deferred history stores, one per tile of the wave (def_i < 0: none)
    int32_t def_den[NT];
    long long def_i[NT];
#pragma unroll
    for (int ti = 0; ti < NT; ti++) {
      def_num[ti] = 0.0;
      def_den[ti] = 0;
      def_i[ti] = -1;
    }
    __shared__ float sm_hav[(HYB || ALL) ? NW : 1][(HYB || ALL) ? NT * RPW * U : 1];  // hybrid / ALL: group-mean probabilities, [wave][tile * 16 + row]
    auto slot_pass = [&](auto ti_c) {
      constexpr int TI = decltype(ti_c)::value;
      const int slot_ti = one_slot + TI * (NW * RPW * U);
      const bool have_ti = one_lane && slot_ti < row_end;
      unsigned long long key_ti = ~0ull;
      float av = 0.f;
      const bool want_probs = HYB ? (a.ring_num != nullptr || a.attn_out != nullptr) : (a.num != nullptr);
      if (want_probs) {  // heavy hitter: the group-mean probability of this lane's row (the head-constant policies keep no history)
        // lane c of a row group computes the probabilities of row t = c / LPR for heads (c % LPR) * KH + [0, KH) — one exp and one
        // IEEE divide per (row, head), spread over the 16 lanes; the score of (row t, head r) sits in lane r of the group as
        // s_keep[TI][t].  The group mean adds the heads in order r = 0 .. RT - 1, like the combine pass.
        const int t_me = c / LPR, j_me = c % LPR;
        float pr[KH];
#pragma unroll
        for (int kk = 0; kk < KH; kk++) {
          const int r_me = j_me * KH + kk;
          float x;
          if constexpr (RT == 4) {
            // lane c = 4t + r of the 16-lane row takes s_keep[TI][t] of lane r: row shifts by 4t (DPP, no LDS crossbar round trips)
            const float x1 = dpp_mov<0x114>(s_keep[TI][1]);  // row_shr:4
            const float x2 = dpp_mov<0x118>(s_keep[TI][2]);  // row_shr:8
            const float x3 = dpp_mov<0x11C>(s_keep[TI][3]);  // row_shr:12
            x = t_me == 0 ? s_keep[TI][0] : (t_me == 1 ? x1 : (t_me == 2 ? x2 : x3));
          } else {
            const int src = (lane & ~15) | r_me;
            x = -INFINITY;
#pragma unroll
            for (int t = 0; t < U; t++) {
              const float v = __shfl(s_keep[TI][t], src, CC_WAVE);
              x = (t == t_me) ? v : x;
            }
          }
          pr[kk] = ElemTraits<T>::rnd(__fdiv_rn(exp_nonpos(x - sm_M1[r_me]), sm_L1[r_me]));
        }
        float sum = 0.f;
        if constexpr (LPR == 4) {
#pragma unroll
          for (int kk = 0; kk < KH; kk++) sum += dpp_mov<0x00>(pr[kk]);
#pragma unroll
          for (int kk = 0; kk < KH; kk++) sum += dpp_mov<0x55>(pr[kk]);
#pragma unroll
          for (int kk = 0; kk < KH; kk++) sum += dpp_mov<0xAA>(pr[kk]);
#pragma unroll
          for (int kk = 0; kk < KH; kk++) sum += dpp_mov<0xFF>(pr[kk]);
        } else if constexpr (LPR == 2) {
          sum += dpp_mov<0xA0>(pr[0]);  // quad_perm [0, 0, 2, 2]
          sum += dpp_mov<0xF5>(pr[0]);  // quad_perm [1, 1, 3, 3]
        } else {
          sum += pr[0];
        }
        av = ElemTraits<T>::rnd(sum * (1.0f / (float)RT));  // RT is a power of two: bit-identical to the IEEE divide of the combine pass
      }
      if constexpr (HYB || ALL) {  // hybrid / ALL: the pass below consumes the probabilities on all lanes
        if (have_ti) sm_hav[wave][TI * (RPW * U) + g * U + c / LPR] = av;
      } else if (have_ti) {
        const size_t i = (size_t)h * S + slot_ti;
        int32_t ps = one_psv[TI];
        double num_old = one_numv[TI];
        int32_t den_old = one_denv[TI];
        if (slot_ti == ins_idx) {  // refilled by this launch's insert: position p, history from zero (cache.py:754-763)
          ps = one_pin;
          num_old = 0.0;
          den_old = 0;
        }
        const int32_t p_next = one_pin + 1;
        const uint32_t low = ((uint32_t)slot_ti << 1) | (uint32_t)(ps == -1);
        if constexpr (L2) {  // ref: cache.py:597-605: dtype(max over ALL heads and slots - norm), recent window -> +inf, base rules
          float gm = -INFINITY;
          bool gn = false;
          if constexpr (L2X) {
            gm = sm_gmax;
            gn = gm != gm;
          } else {
#pragma unroll
            for (int w2 = 0; w2 < NW; w2++) {
              const float v = sm_l2g[w2];
              gn |= (v != v);
              gm = fmaxf(gm, v);
            }
          }
          const float kn_eff = (slot_ti == ins_idx) ? l2_nv_lane : one_kn_f();
          float scn = ElemTraits<T>::rnd((gn ? NAN : gm) - kn_eff);
          if (ps >= p_next - a.w) scn = INFINITY;
          if (slot_ti < a.g) scn = INFINITY;
          if (ps == -1) scn = -INFINITY;
          key_ti = make_key(orderable_f32(scn), low);
        } else if (a.num) {
          if (a.attn_out) ElemTraits<T>::store(reinterpret_cast<T*>(a.attn_out), i, av);
          const double num_new = num_old + (double)av;
          const int32_t den_new = den_old + 1;
          if constexpr (RC) {  // stored behind the LAST gather and the head's fail word: a head's step is committed whole or not at all
            def_num[TI] = num_new;
            def_den[TI] = den_new;
            def_i[TI] = (long long)i;
          } else {
            a.num[i] = num_new;
            a.denom[i] = den_new;
          }
          float scn = __fdiv_rn((float)num_new, (float)(den_new < 1 ? 1 : den_new));
          if (ps < a.g || ps >= p_next - a.w) scn = 1.0f;
          if (ps == -1) scn = 0.0f;
          key_ti = make_key(orderable_f32(scn), low);
        } else {  // head-constant policies: every kv head scores the shared positions for its own copy of the key row (KEY ROWS)
          if (a.policy == 2) {  // ref: cache.py:500-502, 552-556 — arg-min of pos behind the sinks; -1 = empty first
            if (slot_ti >= a.g) key_ti = make_key(orderable_i32(ps), low);
          } else {  // random, ref: cache.py:523 recent window -> +inf, then the base rules :373-376
            float scn = a.rand_next ? one_rndv[TI] : cc_rng_uniform(a.rng_seed, p_next, slot_ti);
            if (ps >= p_next - a.w) scn = INFINITY;
            if (slot_ti < a.g) scn = INFINITY;
            if (ps == -1) scn = -INFINITY;
            key_ti = make_key(orderable_f32(scn), low);
          }
        }
      }
      my_key = key_ti < my_key ? key_ti : my_key;
    };
    // every tile of the wave (compile-time unrolled; tiles past the split's end hold no slot)
    auto all_tiles = [&](auto self, auto ti_c) -> void {
      constexpr int TI = decltype(ti_c)::value;
      if constexpr (TI < NT) {
        if (row_begin + wave * (RPW * U) + TI * (NW * RPW * U) < row_end) {  // wave-uniform: the wave has a tile TI
          slot_pass(ti_c);
          self(self, IntC<TI + 1>{});
        }
      }
    };
    if constexpr (!L2X && !L2C) {
      if (!hrc_skip) all_tiles(all_tiles, IntC<0>{});  // (L2X: behind the partial-O gather — the heads' maxima arrive with it; L2C: done long ago by one wave)
    }
    if constexpr (ALL) {
      // ---- the second half of the per-slot pass on all lanes: the slot_pass branches above, operation for operation (heavy hitter:
      //      cache.py:716-722, 727-749; head-constant policies: cache.py:500-502, 519-524, 552-556)
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      const int32_t p_next = one_pin + 1;
#pragma unroll
      for (int k = 0; k < SN; k++)
        if (all_valid(k) && !hrc_skip) {
          const int sl = all_slot(k);
          const size_t i = (size_t)h * S + sl;
          const float av = sm_hav[wave][lane + 64 * k];
          int32_t ps = one_psv[k];
          double num_old = one_numv[k];
          int32_t den_old = one_denv[k];
          if (sl == ins_idx) {  // refilled by this launch's insert: position p, history from zero (cache.py:754-763)
            ps = one_pin;
            num_old = 0.0;
            den_old = 0;
          }
          const uint32_t low = ((uint32_t)sl << 1) | (uint32_t)(ps == -1);
          unsigned long long key_k = ~0ull;
          if (a.num) {
            if (a.attn_out) ElemTraits<T>::store(reinterpret_cast<T*>(a.attn_out), i, av);
            const double num_new = num_old + (double)av;
            const int32_t den_new = den_old + 1;
            a.num[i] = num_new;
            a.denom[i] = den_new;
            float scn = __fdiv_rn((float)num_new, (float)(den_new < 1 ? 1 : den_new));
            if (ps < a.g || ps >= p_next - a.w) scn = 1.0f;
            if (ps == -1) scn = 0.0f;
            key_k = make_key(orderable_f32(scn), low);
          } else if (a.policy == 2) {
            if (sl >= a.g) key_k = make_key(orderable_i32(ps), low);
          } else {
            float scn = a.rand_next ? one_rndv[k] : cc_rng_uniform(a.rng_seed, p_next, sl);
            if (ps >= p_next - a.w) scn = INFINITY;
            if (sl < a.g) scn = INFINITY;
            if (ps == -1) scn = -INFINITY;
            key_k = make_key(orderable_f32(scn), low);
          }
          my_key = key_k < my_key ? key_k : my_key;
        }
    }
    const int hyb_cts_n = hyb_cts + (hyb_kind == 0 ? 1 : 0);  // hybrid: the head's count after this step's insert
    if constexpr (HYB) {
      // ---- hybrid: ring column / exact window sum / denominator of every slot (cache.py:716-723 with W = 400), then the head's
      //      candidate for position p + 1 (cache.py:844-894) — decode_attn_combine_kernel's operations in its order, on all lanes
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      const int32_t p_next = one_pin + 1;
      const size_t hs = (size_t)a.H * S;
      // opaque constants born HERE: the compiler otherwise hoists `denom + 1` and the mask tests up into the block that issues the
      // state loads (same condition, operands defined there) — with a wait for each load right behind its issue
      int late_one = 1, late_ff = 0xff;
      asm volatile("" : "+s"(late_one), "+s"(late_ff));
#pragma unroll
      for (int k = 0; k < HSL; k++)
        if (hyb_valid(k) && !hrc_skip) {
          const int sl = hyb_slot(k);
          const size_t i = (size_t)h * S + sl;
          int32_t ps = hy_ps[k];
          uint32_t msk = (a.hyb.special_mask && (hy_sp[k] & late_ff) ? 1u : 0u) | (a.hyb.punc_mask && (hy_pu[k] & late_ff) ? 2u : 0u);
          if (sl == ins_idx) {  // this launch's insert: position p (dropped tokens too, :1006-1007), punctuation flag :1011-1016
            ps = one_pin;
            if (hyb_punc && a.hyb.punc_mask) msk |= 2u;
          }
          const float av = sm_hav[wave][lane + 64 * k];
          if (a.attn_out) ElemTraits<T>::store(reinterpret_cast<T*>(a.attn_out), i, av);
          float ws = 0.f;
          int32_t dn = 1;
          if (a.ring_num) {
            WAcc racc{hy_a01[k].x, hy_a01[k].y, hy_a23[k].x, hy_a23[k].y};
            T old_e;
            old_e.x = hy_old[k];
            const float old_v = ElemTraits<T>::load(&old_e, 0);
            T* shadow = reinterpret_cast<T*>(a.ring_acc + hs * 4 + 2) + (size_t)one_rcol * hs;
#ifndef CC_HYB_NO_RING_STORE  // (A/B builds: the price of the reference-layout column — one 2-byte store per slot, W * 2 bytes apart)
            ElemTraits<T>::store(reinterpret_cast<T*>(a.ring_num), i * (size_t)a.ring_W + one_rcol, av);
#endif
            ElemTraits<T>::store(shadow, i, av);
            dn = hy_den[k] + late_one;
            a.denom[i] = dn;
            wacc_add_value(racc, av, false);
            wacc_add_value(racc, old_v, true);
            *reinterpret_cast<ulonglong2*>(a.ring_acc + i * 4) = make_ulonglong2(racc.w0, racc.w1);
            *reinterpret_cast<ulonglong2*>(a.ring_acc + i * 4 + 2) = make_ulonglong2(racc.w2, racc.special);
            ws = wacc_round<T>(racc);
            a.ring_wsum[i] = ws;
          }
          const int flags = hy_flags;
          if ((flags & (HF_HH | HF_WIN)) && !(flags & HF_FULL) && sl < (hyb_cts_n < S ? hyb_cts_n : S)) {
            float scn;
            if (flags & HF_HH) {
              const int32_t d = dn > a.hyb.W ? a.hyb.W : dn;  // clamp_max only (:868-870)
              scn = __fdiv_rn(ws, (float)d);
            } else {
              scn = (float)ps;  // :873
            }
            bool save = sl < a.g || ((flags & HF_SPECIAL) && (msk & 1u)) || ((flags & HF_PUNC) && (msk & 2u));  // :876-883
            if (flags & HF_WIN) save |= ps > p_next - hy_win;  // :885-889 strict
            if (save) scn = INFINITY;
            const unsigned long long key = make_key(orderable_f32(scn), (uint32_t)sl << 1);
            my_key = key < my_key ? key : my_key;
          }
        }
    }
    // one key per WAVE (a head's key row has room for NW per 64-slot workgroup): nothing crosses the waves after the last
    // barrier of the finish, so no wave's stores wait for another wave (the launch ends a store round trip after the LAST
    // store is issued: every store that can go out early shortens it)
    // (L2C: the workgroup's minimum in its last wave's entry, ~0 in the others': the row's minimum is all a reader takes)
    unsigned long long wk = L2C ? l2c_wk : wave_min_u64_uniform(my_key);
    auto store_key = [&]() {
      if (lane == 0) {
        unsigned long long* nk_row = const_cast<unsigned long long*>(a.next_key) + (size_t)h * a.nk;
        const int e0 = split * NW + wave;
        nk_row[e0] = wk;  // every key of this row was consumed before its readers published: no reader is left
        for (int s2 = e0 + ns * NW; s2 < a.nk_read; s2 += ns * NW) nk_row[s2] = ~0ull;  // entries beyond nk_read are never read
      }
    };
    bool failed = false;  // EML: the head's step is not committed by this launch (somebody gave up)
    if constexpr (EML) {
      __builtin_amdgcn_sched_barrier(0);
      // L2X, level two: the kv heads' maxima (lane = kv head), one wave per workgroup, through memory — requested here, behind the
      // partial-O round (issued above), polled with it
      const bool hm_mine = L2X && wave == NW - 1;
      const int hm_off = hm_base + (lane < a.H ? lane : 0) * 16;
      auto ok_hm = [&]() { return !hm_mine || lane >= a.H || (hmq[0] == hm_ep + 1u && hmq[2] == hm_ep + 1u); };
      if constexpr (L2X) {
        if (hm_mine) hmq = __builtin_amdgcn_raw_buffer_load_b128(ml_rsrc, hm_off, 0, kOneAuxCoherent);
      }
      // L2C, tail: the workgroups' two granules of this head (lane = split), gathered by the last wave of the head's split-0 workgroup
      // with the partial-O round — the record of this position is resolved from them
      const bool rec_mine = L2C && split == 0 && wave == NW - 1;
      const int rec_off = nm_base + h * kOneNmHead + (lane < ns ? lane : 0) * 32;
      u32x4_t recq = {0u, 0u, 0u, 0u}, recq2 = {0u, 0u, 0u, 0u};
      auto ok_rec = [&]() { return !rec_mine || lane >= ns || (recq[0] == tag && recq[2] == tag && recq2[0] == tag && recq2[2] == tag); };
      auto load_rec = [&]() {
        recq = __builtin_amdgcn_raw_buffer_load_b128(ml_rsrc, rec_mine ? rec_off : 0x7ffffff0, 0, kGatherAux);
        recq2 = __builtin_amdgcn_raw_buffer_load_b128(ml_rsrc, rec_mine ? rec_off + 16 : 0x7ffffff0, 0, kGatherAux);
      };
      if constexpr (L2C) load_rec();  // (unconditional: every other wave aims past the buffer's end — no request, zeros back)
      WaitBound o_wb;
      for (unsigned spins = 0;; spins++) {
        if (__all(ok_o()) && __all(ok_hm()) && __all(ok_rec())) break;
        if (RC && failq == tag) break;  // a workgroup of this head gave up: the head's step is not committed, nothing left to wait for
        if (o_wb.expired(spins)) {
          timed_out = true;
          break;
        }
        __builtin_amdgcn_s_sleep(1);
        asm volatile("" ::: "memory");  // every round re-reads memory
        load_o();
        if constexpr (L2X) {
          if (hm_mine) hmq = __builtin_amdgcn_raw_buffer_load_b128(ml_rsrc, hm_off, 0, kOneAuxCoherent);
        }
        if constexpr (L2C) load_rec();
      }
      unsigned long long rec_new = 0ull;  // L2C: the head's record of this position (wave-uniform; meaningful in rec_mine's wave)
      if constexpr (L2C) {
        if (rec_mine) {
          // the slot position p + 2 will evict: the arg-min of the keys this step scored (the minimum over the workgroups' minima)
          const unsigned long long kq = lane < ns ? (((unsigned long long)recq[1] << 32) | recq[3]) : ~0ull;
          const unsigned long long kmin = wave_min_u64_uniform(kq);
          const int e_next = (kmin == ~0ull) ? -1 : (int)((kmin & 0xffffffffull) >> 1);
          // the head's two largest norms, a holder of the largest — and the head's maximum over the slots it keeps
          const unsigned a1 = lane < ns ? (recq2[1] & 0xffffu) : 0u, a2 = lane < ns ? (recq2[1] >> 16) : 0u, ai = lane < ns ? recq2[3] : 0u;
          const unsigned long long k64 = ((unsigned long long)a1 << 32) | ai;
          const unsigned long long K1 = ~wave_min_u64_uniform(~k64);
          const unsigned long long win = __ballot(k64 == K1);                         // (slots are distinct across workgroups; lanes
          const int wl = (int)__builtin_ctzll(win ? win : 1ull);                      //  without one tie at 0 only when every norm is +0)
          const unsigned long long x2 = (lane == wl) ? (unsigned long long)a2 : (unsigned long long)a1;
          const unsigned T2h = (unsigned)~wave_min_u64_uniform(~x2);
          const unsigned T1h = (unsigned)(K1 >> 32), i1h = (unsigned)(K1 & 0xffffffffull);
          rec_new = cc_l2_record((e_next >= 0 && i1h == (unsigned)e_next) ? T2h : T1h, T1h, i1h);
        }
      }
      if (a.trace) trE = __builtin_amdgcn_s_memtime();
      if (timed_out) give_up();
      else if (failq == tag && lane == 0) sm_fail = 1u;  // another workgroup of this head gave up: nothing of the head's step is committed
      stash_o();
      if constexpr (L2X) {
        if (hm_mine) {  // the maximum over all kv heads (NaN propagates: torch.max)
          const float v = lane < a.H ? __uint_as_float(hmq[1]) : -INFINITY;
          const bool gn = __any(lane < a.H && (hmq[3] != 0u || v != v)) != 0;
          const float gm = wave_max_uniform(v);
          if (lane == 0) sm_gmax = gn ? NAN : gm;
        }
      }
      __syncthreads();  // (also makes the verdict workgroup-uniform)
      failed = sm_fail != 0u;
      if (!failed) {
        y_fold();
        if constexpr (L2X) {  // the slots' next-eviction keys, now that the maximum is here (no history to update: l2 keeps none)
          all_tiles(all_tiles, IntC<0>{});
          wk = wave_min_u64_uniform(my_key);
        }
        if (!rc_replay) {  // the step's state, behind the last gather and the fail word
          if constexpr (RC) {
#pragma unroll
            for (int ti = 0; ti < NT; ti++)
              if (def_i[ti] >= 0) {
                a.num[def_i[ti]] = def_num[ti];
                a.denom[def_i[ti]] = def_den[ti];
              }
          }
          store_key();
          if constexpr (L2C) {  // the record the step of position p + 1 starts from (a retry of p reads p - 1's: the other entry)
            if (rec_mine && lane == 0)
              const_cast<unsigned long long*>(a.next_key)[(size_t)h * a.nk + (a.nk - kNextKeyTail) + (one_pin & 1)] = rec_new;
          }
        }
      }
    } else {
      failed = hrc_failed;
      if (!hrc_skip) store_key();
      if (a.trace) trE = __builtin_amdgcn_s_memtime();
    }
    if (threadIdx.x == 0) {
      if (split == 0 && !failed) {
        a.one_hdr[h] = tag;  // all n_split workgroups of this head have published, hence read the old epoch
        if constexpr (HYB) {
          if (!rc_replay) {  // (recoverable form: a head that committed this position before does not count twice)
            a.cache_cts[h] = hyb_cts_n;  // every workgroup of this head has read the old count (it decided before it published)
            // the step counter and num_punc are read by the workgroups of EVERY head: the head that completes the set commits them
            const unsigned t = atomicAdd(&a.one_hdr[kOneTicketWord], 1u);
            if (t == (unsigned)a.H - 1u) {
              a.one_hdr[kOneTicketWord] = 0u;
              if (a.ring_num && a.hh_counter) *a.hh_counter += 1;               // cache.py:723
              if (hy_punc && a.hyb.num_punc) *a.hyb.num_punc += 1;              // cache.py:1017, once per step
            }
          }
        } else if (!rc_replay) {
          if (h == 0 && a.hh_counter) *a.hh_counter += 1;
          if ((EML || NRC) && a.commit) {  // recoverable hand-off: the head's count travels with split 0's commit
            if (ins_was_empty && (a.Hc == a.H || h == 0)) a.cache_cts[a.Hc == a.H ? h : 0] += 1;  // (one writer per count)
          }
        }
      }
      if constexpr (RC || NRC) {  // this workgroup's part of the step (its slots' history, its keys) is committed
        if (!failed && !rc_replay && a.commit) a.commit[(size_t)h * kRcStride + 2 + split] = one_pin;
      }
      if constexpr (QKV && CC_QKV_TRACE != 0) {
        if (a.qkv.trace) {
          unsigned long long* tr = a.qkv.trace + (size_t)(h * ns + split) * 16;
          qt[11] = __builtin_amdgcn_s_memrealtime();
#pragma unroll
          for (int i = 0; i < 12; i++) tr[i] = qt[i];
          tr[12] = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11));  // HW_REG_XCC_ID[3:0]
        }
      }
      if (a.trace) {
        unsigned long long* tr = a.trace + (size_t)(h * ns + split) * 16;
        tr[0] = tr0; tr[1] = tr1; tr[2] = tr2; tr[3] = tr3; tr[4] = tr4; tr[5] = __builtin_amdgcn_s_memtime();
        tr[6] = rt0;
        tr[7] = rt1;
        tr[8] = __builtin_amdgcn_s_memrealtime();
        tr[9] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));   // HW_REG_HW_ID
        tr[10] = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11));  // HW_REG_XCC_ID[3:0]
        tr[11] = trA; tr[12] = trB; tr[13] = trC;  // wave 0: K arrived / scores ready / P.V issued
        tr[14] = trD; tr[15] = trE;                // finish: final (M, L) + weights in LDS / slots, y chains and keys done
      }
    }
    return;
  }
  for (int t = threadIdx.x; t < RT * D; t += NW * 64) {
    const int r = t / D, d = t - r * D;
    float M = sm_wm[0][r];
#pragma unroll
    for (int w = 1; w < NW; w++) M = fmaxf(M, sm_wm[w][r]);
    const float Mu = (M == -INFINITY) ? 0.f : M;
    float L = 0.f, O = 0.f;
#pragma unroll
    for (int w = 0; w < NW; w++) {  // fixed order: deterministic
      const float f = fast_exp(sm_wm[w][r] - Mu);
      L = fmaf(sm_wl[w][r], f, L);
      O = fmaf(sm_wacc(w, r, d), f, O);
    }
    if (a.abl & 16) {  // measurement only: merge but do not store the partials
      if (O + L == 1.2345f) a.part_ml[0] = L;
      continue;
    }
    const size_t pj = (size_t)(q0 + r) * a.n_split + split;
    a.part_o[pj * D + d] = O;
    if (d == 0) *reinterpret_cast<float2*>(a.part_ml + pj * 2) = make_float2(M, L);
  }
}

}  // namespace
