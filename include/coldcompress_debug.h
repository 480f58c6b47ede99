/* coldcompress_debug.h — measurement hooks, test hooks and process-wide A/B switches of libcoldcompress_hip.so (r5: split out of
 * coldcompress.h).  NOTHING here is part of the drop-in boundary: a caller that replaces the reference's hot path
 * needs include/coldcompress.h only.  bench.py, tools/ and tests/ use these; the Python product layer uses NONE of them (its recovery
 * path demotes the L2-resident hand-off per device: cc_decode_step_demote_l2_handoff, include/coldcompress.h).
 *
 * The three switches are process-wide (std::atomic<int> inside the library: reads and writes are safe from any thread, but a flip is
 * seen by every stream's NEXT call) — they select between forms that give bit-identical cache state, so flipping one never changes
 * results, only which kernels run; the entry points of coldcompress.h stay reentrant across streams. */
#ifndef COLDCOMPRESS_DEBUG_H
#define COLDCOMPRESS_DEBUG_H
#include "coldcompress.h"
#ifdef __cplusplus
extern "C" {
#endif

/* Process-wide switch (default 1): 0 makes every fused decode step (heavy hitter, recent_global / full, random, l2) use the
 * two-launch form — what tests compare the single launch against.  The head-constant policies (recent_global, full,
 * random: cc_decode_step_recent_global / cc_decode_step_random) and l2 (cc_decode_step_l2: every workgroup also gathers every
 * workgroup's norm maximum) take the single launch under the same conditions (l2: at most 768 workgroups, 32 kv heads). */
void cc_decode_step_set_single_launch(int32_t enabled);
/* Test hook: n_workgroups one-wave workgroups that hold lds_bytes (256 .. 163840) of LDS each and idle for `microseconds`
 * (<= 5 s) — what a co-tenant kernel does to the residency of a single-launch step (tests/test_gpu_recovery.py).  scratch: >= 4
 * bytes of device memory. */
int cc_debug_occupy(int32_t n_workgroups, int32_t lds_bytes, int32_t microseconds, void* scratch, cc_stream_t stream);
/* Wide geometry (r3): ONE 8-wave workgroup per CU (128 cache rows each) instead of two 4-wave ones, for the plain 16-bit
 * caches (heavy hitter / recent_global / full / random, 4 or 8 query heads per kv head, head_dim 128) that have 16-row
 * tiles for it (H * S / 16 >= 1280 and H * ceil(S / 128) <= 256: Llama-3-8B at cache_len 2560 .. 4096).  On by default;
 * process-wide.  The geometry decides the split partials (hence the last bits of y and of the probabilities) and which
 * entries of a head's key row are live: all forms of one cache's step (one launch, two, three calls) follow the switch
 * together; flip it only where the fused pipeline is re-seeded (prepare_decode / cc_hh_next_key_init). */
void cc_decode_step_set_wide(int32_t enabled);
/* 1 if this build's single-launch l2 step takes its norm maximum from the record carried in the key rows' tail (the r6 A/B build,
 * -DCC_V_L2CARRY=1) and writes that record itself; 0 = the r4 / r5 exchange (the product).  In that build the seed
 * (cc_l2_next_key_init) and the two-launch l2 step write the record too (tests/test_gpu_fused_step.py checks it against the state). */
int32_t cc_decode_step_l2_carry(void);
/* Process-wide off switch of the L2-resident hand-off (include/coldcompress.h, cc_decode_step_probe_xcd): 0 = always the memory
 * hand-off, on every device (A/B measurements; the harness's recovery path uses the per-device cc_decode_step_demote_l2_handoff
 * instead).  A step captured into a hipGraph keeps the form it was captured with. */
void cc_decode_step_set_l2_handoff(int32_t enabled);
/* Measurement hook: buf = device buffer of [workgroups][16] uint64, or NULL (default).  While set, thread 0 of every
 * workgroup of a single-launch step records [0..5] s_memtime stamps (start, streaming done, published, sentinel seen,
 * gathered, end), [6..8] s_memrealtime at start / streaming done / end, [9] HW_ID, [10] XCC_ID, [11..13] s_memtime of wave 0
 * when its K rows have arrived / its scores are in registers / its P.V products are issued, [14..15] s_memtime of thread 0 behind the
 * two barriers of the finish. */
void cc_decode_step_trace(void* buf);
/* Measurement hook: the launch floor of the layer step over cache `c` — a kernel with the step's grid, workgroup size and
 * K/V access pattern (every row read once with the step's 16-byte non-temporal loads) and NOTHING else.  Its duration is what
 * any stand-alone launch streaming this cache costs on the device (launch boundary + first byte + transfer): bench.py times
 * it beside the step (roofline.launch_floor_us, frac_of_launch_floor).  16-bit caches with head_dim 128; scratch: >= 4 bytes
 * of device memory (never written in practice). */
int cc_decode_step_stream_floor(const cc_kv_view* c, int32_t HQ, void* scratch, cc_stream_t stream);
/* ... with the geometry given (measurement hook, r4: profiles/r04_step_geometry_H1.jsonl): `waves` in {1, 2, 4, 8} waves per workgroup,
 * `rows_per_workgroup` cache rows each (a multiple of 16 * waves) — what a launch costs that streams the cache in smaller pieces on
 * more CUs (few kv heads per rank: tp.py:151-154). */
int cc_decode_step_stream_floor_geom(const cc_kv_view* c, int32_t waves, int32_t rows_per_workgroup, void* scratch, cc_stream_t stream);
/* Measurement hook (cf. cc_decode_attn_gqa_phases): the same step with its launches selectable. */
int cc_decode_step_heavy_hitter_phases(const cc_kv_view* c, const void* q, const void* k_new, const void* v_new,
                                const int32_t* input_pos, double* num, int32_t* denom, int64_t* counter,
                                uint64_t* next_key, int32_t global_tokens, int32_t recent_window, int32_t HQ,
                                float scale, void* y, void* attn_out, void* workspace, size_t workspace_bytes,
                                cc_stream_t stream, int32_t phases);
/* Measurement hook: the same operation with its two launches selectable, so that bench.py can bracket the
 * dominant kernel alone with HIP events.  phases: 1 = split kernel only (K/V streaming pass),
 * 2 = combine kernel only (needs a prior phase-1 call on the same workspace), 3 = both (== cc_decode_attn_gqa). */
int cc_decode_attn_gqa_phases(const void* q, const void* k, const void* v, const uint8_t* mask, int32_t HQ,
                              int32_t H, int32_t S, int32_t D, int32_t dtype, float scale, void* y,
                              void* attn_out, void* probs_out, double* hh_num, int32_t* hh_denom,
                              int64_t* hh_counter, void* workspace, size_t workspace_bytes,
                              cc_stream_t stream, int32_t phases);
/* Measurement hook of the QKV form of the layer step (cc_decode_step_qkv_rc): buf = device buffer of [workgroups][16] uint64 or NULL.
 * Only a -DCC_QKV_TRACE=1 build of cc_attn_decode_qkv.hip writes stamps (tools/ab_variant_qkv.sh, tools/trace_qkv.py): the 100 MHz
 * device clock at entry, first weight unit requested, projection started, input normalised, dots done, published, head gathered,
 * q in LDS, scores, P.V issued, merge barrier, end. */
void cc_debug_qkv_trace(void* buf);

#ifdef __cplusplus
}
#endif
#endif /* COLDCOMPRESS_DEBUG_H */
