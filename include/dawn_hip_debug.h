/* dawn_hip_debug.h — test hooks, measurement and tuning of libdawn_hip.so.
 *
 * NOT part of the drop-in boundary: the reference-side binding binds include/dawn_hip.h only.  Everything
 * here is exported by the same library for this repository's tests (tests/), bench.py and the tools under tools/: synthetic
 * index contents, rows read back, kernel timing, the filters' intermediate results, and the catalogue of option names
 * dawn_index_set_option / dawn_embedder_set_option accept.  Results never depend on an option. */
#ifndef DAWN_HIP_DEBUG_H
#define DAWN_HIP_DEBUG_H
#include "dawn_hip.h"
#ifdef __cplusplus
extern "C" {
#endif
#if defined(__GNUC__)
#pragma GCC visibility push(default) /* the library is built with -fvisibility=hidden: only what these headers declare is exported */
#endif

/* Fill rows [size, size+n) with the synthetic unit rows of DESIGN.md §5 (stream `seed`, rows
 * first_row..) generated on the GPU, ids = first_id + i.  Bench / test input only. */
int dawn_index_fill_synthetic(dawn_index *idx, uint64_t seed, uint64_t first_row, size_t n, uint64_t first_id);
/* Copy rows [first, first+n) back to the host (tests: generator parity, save/load). */
int dawn_index_get_rows(dawn_index *idx, size_t first, size_t n, float *out_rows, uint64_t *out_ids);

/* Kernel-level timing of the dominant (scan) kernel with HIP events recorded on the launch
 * stream.  enable=1 starts recording (at most 4096 launches are kept), read returns the launch
 * count and summed milliseconds since the last reset and resets. Synchronises the device. */
int dawn_index_profile_enable(dawn_index *idx, int enable);
int dawn_index_profile_read(dawn_index *idx, uint64_t *launches, double *total_ms);
/* Test hook: the matrix-core FILTER scores (f16 MFMA, before the exact rescore) of B <= 256 queries against
 * rows [0, n), n = min(size, 8192): out [B][n].  Lets a test check the bound the certificate relies on. */
int dawn_index_debug_filter_scores(dawn_index *idx, const float *queries, size_t B, float *out, size_t *n_out);
/* Test hook: the same for the FP6 (e2m3) shadow of the rows (scan_f6.hip; options "f6_shadow" = 1, "f6_min_rows"): upper bounds. */
int dawn_index_debug_f6_scores(dawn_index *idx, const float *queries, size_t B, float *out, size_t *n_out);
/* What the feedback of the batched paths did: batches (of <= 256 queries) the FP6 first filter took; batches its feedback handed
 * to the int8 pass instead (an index whose FP6-filtered queries end in the ladder more than 30 % of the time — topical rows —
 * suspends it for 16 .. 1024 batches at a time); batches the int8 pass ran with thresholds four times as deep ("mfma_target" 4096
 * instead of 1024: an index that sent more than 10 % of a window of 1024 batched queries to the ladder keeps them until its rows
 * change); rerun_answers: queries of such batches whose failed certificate was settled by a SECOND matrix-core pass with the
 * threshold their own k-th exact distance gives (option "batch_rerun"; the bounded pass keeps the rest).  Option "ladder_feedback" = 0
 * switches all of it off.  A sharded handle reports the sums over its shards. */
int dawn_index_stats_batch_feedback(dawn_index *idx, uint64_t *f6_batches, uint64_t *f6_suspended, uint64_t *deepened_batches,
                                    uint64_t *rerun_answers);
/* The device-side counters of the index as they are: out8[8], indexed by a query's final flag (1 exact pass over all rows, 2 second
 * chance, 3 deeper round, 4 bounded exact pass, 6 second matrix-core pass); [5] single queries whose packed-stream certificate failed;
 * [7] (row, query) pairs of the bounded pass that got past its int8 bound (mod 2^32); [0] queries the wide batch form of that pass answered.  A sharded handle reports the sums over its shards. */
int dawn_index_debug_raw_stats(dawn_index *idx, uint64_t *out8);
/* The packed stream's list sizing (scan_i6.hip: i6_refine_count): *n_refine = entries of its coarse list a wave refines for a search of
 * `count` results on this index (0: the packed stream would not certify, the int8 stream is used; -1: no packed shadow), frac64[64]
 * (may be NULL) = the measured histogram of the shadow's error bounds E in bins of 0.004 (all zero: not measured, option
 * "i6_slack_model" = 0).  Single-device handles. */
int dawn_index_debug_i6_refine(dawn_index *idx, size_t count, int *n_refine, float *frac64);
/* Diagnostic: per-wave phase cycle sums ([blocks][8 waves][8 phases]) of the last batched full pass run with the
 * "mfma_sched" option = 2 (s_memtime-stamped build of the kernel; tools/batch_phases.py prints the shares). */
int dawn_index_debug_read_diag(dawn_index *idx, unsigned long long *out, size_t blocks);
/* Timing hook: mean ms of the matrix-core full pass alone (thresholds of the last batched search; results discarded). */
int dawn_index_debug_time_full_pass(dawn_index *idx, size_t B, int iters, double *mean_ms);
/* Test hook: per-workgroup candidate lists of the batch-1 streaming filter (scores descending, rows; [blocks][64]). */
int dawn_index_debug_stream_lists(dawn_index *idx, const float *query, float *out_scores, uint32_t *out_rows,
                                  size_t cap_blocks, size_t *n_blocks);
/* Test hook: the certificate bound T of the packed-shadow stream (scan_i6.hip) for the query of the last
 * dawn_index_debug_stream_lists call: every row that is in no list scores <= T. */
int dawn_index_debug_stream_bound(dawn_index *idx, float *bound);
/* Option catalogue of dawn_index_set_option (dawn_hip.h) — tuning knobs (tests and tools sweep them; the defaults are the tuned values):
 *   "mfma_min_batch"   batches of at least this many queries take the matrix-core path; 0 (default): 2 on small indexes, up to 5 / 7
 *                      on large ones, where up to 4 / 6 queries are cheaper as one stream of the int8 shadow
 *   "mfma_blocks"      workgroups of the matrix-core kernels (default: one per CU)
 *   "mfma_sched"       4 = default kernel choice, 5 = pipelined 4-wave kernel for every pass, 1 = 8-wave kernel only,
 *                      0 = lockstep converting kernel on the f32 rows, 32 = the int8 filter on v_mfma_i32_32x32x32_i8 (default:
 *                      16x16x64, the shape the chip clocks higher under load; same results).  (The timing experiments 2 / 41..55 — parts of the
 *                      pipelined kernels switched off, wrong results by design — only exist in `make EXPERIMENTS=1`
 *                      builds; the release library rejects them.)
 *   "stream_dynamic_tail" 0: the single-query streams (packed shadow, f32 rows) assign every unit of work statically (default 1:
 *                      the last eighth of a long stream is handed out on demand; same results)
 *   "mfma_target"      candidates per query the sampled thresholds of the matrix-core path aim for (1024; twice that for count > 32)
 *   "i8_shadow"        0: no integer shadows (int8: 384 B/row, scan_i8.hip; 6-bit: 288 B/row, scan_i6.hip) of the index rows: the
 *                      filters read the f16 shadow of an f32 index / the rows of a bf16 index themselves.  Default 1, or env
 *                      DAWN_I8_SHADOW at creation
 *   "i6_shadow"        0: no packed shadow: single queries stream the int8 shadow (its memory is released; 1 rebuilds it).
 *                      Default 1, or env DAWN_I6_SHADOW at creation
 *   "i6_refine"        entries of its coarse list a wave of the packed stream keeps and refines: 1..64, or 0 = chosen from the
 *                      index size, k and the shadow's measured error bounds (default; dawn_index_debug_i6_refine reads the choice);
 *                      too few cost a failed certificate (the bounded pass answers), never a result;
 *                      -1 (tests): full lists that are NOT refined — dawn_index_debug_stream_lists then returns the packed
 *                      shadow's own bounds
 *   "zero_copy_batch"  host API: batches of up to this many queries get their results by zero-copy stores into pinned host memory
 *                      (default 256 = all; 0: by copy commands)
 *   "i6_slack_model"   1 (default): that choice uses the histogram of the shadow's own error bounds E, re-read whenever the shadow
 *                      changes; 0: the constants of rounds 3-4 (deeper lists: A/B)
 *   "i6_bits"          bits per component of the packed shadow: 5 (240 B/row, default; env DAWN_I6_BITS) or 6 (288 B/row)
 *   "i6_min_rows"      single queries of an index of at least this many rows stream the packed shadow (default 768 Ki, or env
 *                      DAWN_I6_MIN_ROWS at creation; below it the fixed costs of a search dominate and the shadow is not kept)
 *   "i6_scan_blocks" / "i6_scan_threads" / "i6_scan_ring"   geometry of the packed stream: workgroups, 64..512 threads, loads in
 *                      flight per wave (6 bits: 12 / 6 / 4 / 3 / 2 fragments of 768 B; 5 bits: 8 or 4 loads of 768 B - 1 KiB); same
 *                      results whatever the geometry
 *   "i8_batched"       0: only batches below mfma_min_batch filter on the int8 shadow
 *   "f16_shadow"       0: an f32 index keeps no f16 shadow either (filters read / convert the f32 rows)
 *   "f16_shadow_b1"    0: batches below mfma_min_batch stream the f32 rows instead of a shadow
 *   "scan_blocks" / "scan_threads" / "scan_unroll"                  geometry of the f32-row stream
 *   "shadow_scan_blocks" / "shadow_scan_threads" / "shadow_scan_unroll"   geometry of the shadow fragment streams; the int8
 *                      stream's unroll code picks the kernel: 1-4 the round-2 kernel with rings of 3 / 4 / 12 / 6 fragments, 5 the
 *                      same with plain loads, 6 / 8 / 9 / 10 the software-pipelined kernel with rings of 12 / 6 / 4 / 3 (8 = default
 *                      at 4 waves per CU), 7 pipelined + per-XCD address ranges; same results whatever the code
 *   "force_fallback"   1: every query also takes the exact pass (tests); 2: every certificate is made to fail and the ladder
 *                      behind it answers (bounded exact pass first)
 *   "bounded_pass"     0: a failed certificate goes straight to the exact pass over all rows (A/B of the ladder); default 1
 *   "ladder_feedback"  0: single queries of a large index always try the packed stream first, however often its certificate
 *                      fails (default 1: full refinement lists above 5 % failures, the bounded pass directly above 35 %);
 *                      2: never — the bounded pass is their whole search (what a demoted index does; A/B, tests)
 *   "debug_bad_threshold" test hook: a demoted search starts its bounded pass from an impossible threshold; the pass notices and
 *                      its last workgroup scans all rows exactly (counted as a fallback)
 *   "bounded_packed"   the bounded pass of a SINGLE query streams the packed 5-bit shadow (240 B/row) instead of the int8 one: 0 never,
 *                      1 (default) wherever the packed shadow is live (>= 2 Mi rows) and the pass has a first threshold — a failed
 *                      packed stream's, or the seed's; unseeded only from 40 Mi rows —, 2 always (tests)
 *   "bounded_seed"     1 (default): a demoted single query's bounded pass on the packed shadow starts from the k-th exact distance of a
 *                      packed-stream search over the first 1/32 of the rows (topical rows, mean ms per query, int8 form -> packed
 *                      seeded: 12.5 M rows 0.88 -> 0.68, 100 M 5.79 -> 4.17); 0: no seed; 2: also on indexes below 2 Mi rows (tests)
 *   "bounded_seed_shift" the seed searches the first n >> shift rows, 2..8 (default 5 = 1/32: a flat optimum — 100 M topical rows, mean
 *                      ms per query at shift 3 .. 7: 4.43 / 4.26 / 4.17 / 4.15 / 4.13 with the p95 rising again from 6,
 *                      profiles/r04/bounded_seed_fraction_sweep_*.log)
 *   "batch_rerun"      a second matrix-core pass for the flagged queries of a batch, each with the threshold its own k-th exact distance
 *                      gives, before the bounded pass takes what is left: 0 never (default), 1 on indexes whose batch feedback has
 *                      deepened the thresholds, 2 every batch.  100 M topical rows: settles 18-30 % of a batch at the default depth
 *                      (74 -> 69 ms), 4-19 % at the deepened one, where it no longer pays for its 12 ms (63 -> 71 ms)
 *   "bounded_wide"     1 (default): the flagged queries of a BATCH go through the wide form of the bounded pass first — 64 queries per
 *                      stream of the int8 shadow, no lists: pairs past the int8 bound are re-tested on the f32 row, the few that can still
 *                      matter are scored in the reference's order and appended, a finish kernel sorts them; a query whose buffer
 *                      (2048 results) overflows is answered by the 16-query form behind it; 0: the 16-query form only (A/B, tests)
 *   "bounded_multi_waves"  waves per workgroup of the 16-query batch form of the bounded pass, 8 (default) or 4 (one workgroup per CU
 *                      either way; 74.0 against 77.4 ms per topical batch of 256 at 100 M rows)
 *   "bounded_multi_packed" 1: the 16-query batch form streams the packed 5-bit shadow too.  Slower (100 M topical
 *                      rows: 80.9 against 73.9 ms per batch of 256: sixteen queries per stream turn the looser bound into several
 *                      times the hits to queue and score); default 0
 *   "bounded_ring"     16-B fragments a wave of the bounded pass (int8 shadow) keeps in flight, 6 (default) or 12 — no
 *                      measurable difference (profiles/r04/bounded_ring_ab_100M.log)
 *   "f6_shadow"        batches of an index of at least "f6_min_rows" rows (default 64 Mi: below ~50 M rows the survivors' re-scoring costs more than
 *                      the pass saves) filter on an FP6 (e2m3) shadow of the rows first (288 B/row; v_mfma_scale_f32_16x16x128_f8f6f4: 1.5 x the
 *                      int8 matrix rate under the chip's power envelope): 0 never, 1 whenever it can be allocated, 2 (DEFAULT, or env
 *                      DAWN_F6_SHADOW at creation) auto — only where it leaves 24 GiB of HBM free once built (it is an optional accelerator:
 *                      the first shadow to go when HBM runs out), and its feedback suspends it on an index whose FP6-filtered queries end in
 *                      the ladder (topical rows).  Its survivors are re-scored on the f32 rows ("f6_refine_rows" 1, default) or on the int8
 *                      shadow (0; always for a bf16 index); "f6_target" = survivors per query its threshold aims for (12288; twice
 *                      that for count > 32); "f6_stagger" -1 (default): the pass staged through LDS, >= 0: the register-ring pass
 *                      with its waves that many tiles apart (A/B), -2 / -4: timing experiments of the LDS-staged pass in a `make EXPERIMENTS=1`
 *                      build.  100 M rows x 256 queries: 8.4-8.5 against 9.3-9.6 ms per batch for 28.8 GB more HBM
 *   "i6_central_tail"  1: the packed stream's workgroups do not rescore their own 64 rows exactly; merge_rescore_kernel rescores
 *                      the index's 64 best by the refined score (measured: a wash; default 0)
 *   "i6_dyn_chunk" / "i6_dyn_share"   the packed stream's dynamically assigned tail: sub-tiles per chunk (default 16; 8 below 32 Mi
 *                      rows) and sixteenths of the index it covers (2; 3)
 *   "debug_i8_levels"  experiment hook: quantise this index's int8 shadow to +-N levels, 3..127 (127 = normal), bytes unchanged —
 *                      what a coarser shadow would cost the certificates (tools/coarse_shadow_probe.py); results stay exact
 *   "debug_fail_alloc" test hook for the out-of-HBM order of the filter sources (FP6 shadow -> 6-bit shadow -> int8 shadow -> f16 shadow ->
 *                      the rows themselves): bit 0 / bit 1 / bit 2 / bit 3 make the int8 / f16 / 6-bit / FP6 shadow allocation fail as if
 *                      the card were full; 0 = normal
 *   "synth_dist"       rows made by dawn_index_fill_synthetic: 0 the spec's uniform rows (default), 1 Gaussian, 2 heavy-tailed
 *                      (4 fixed dimensions x5), 3 heavy-tailed (4 dimensions per row x5) — bench legs on realistic tails; 4 topical mixture
 *                      (Zipf-sized clusters, cosine 0.5 .. 0.95 inside a cluster; restated on the CPU: dawnsearch_amd/synth.py),
 *                      5 the same with runs of 256 consecutive rows per cluster (one site's pages inserted back to back)
 * (dawn_index_set_option itself is declared in dawn_hip.h.) */

/* Option catalogue of dawn_embedder_set_option (dawn_hip.h; defaults are the tuned values): "gemm_bf16x3" 0 = batches above the latency form run their dense layers on
 * the f32-MFMA tile kernel instead of the f32-accurate 3-way bf16 split on the bf16 matrix cores (default 1;
 * "gemm3_big_min_tiles" = number of 128 x 128 tiles from which that form is used, "gemm3_stages" = ring depth of its 64 x 64 form,
 * "gemm3_pingpong" 0 = the 128 x 128 form's waves in lockstep, "gemm3_persistent" = its workgroups (default 256, one per CU, walking
 * the tile list; 0 = one per tile)); "attention_wave" 1 = sequences of up to 64 tokens always take the wave-per-sequence
 * attention kernel (default 0: only where the dense layers read planes), 2 = sequences of up to 32 tokens of a latency-form call take the
 * three-phase block kernel instead of the register form (attention_regs_kernel: one wave per head, no LDS; one 27-token text 0.183 -> 0.169 ms);
 * "ffn2_split" 0 = the FFN-down layer of a latency-form call (<= 64 tokens) in one piece (default 1: four K-slices, their partial sums added
 * up by the next LayerNorm: 0.169 -> 0.158 ms);
 * "skinny_max_rows" = total tokens up to which the GEMMs use the split-K latency form; "graphs" 0 = never replay hipGraphs (default 1: forwards of up to "graph_max_tokens" = 512
 * tokens are captured at the second sighting of their (B, tokens, longest sequence, buffers) shape and replayed);
 * "host_io" 0 = dawn_embedder_forward moves ids / offsets / vectors with three copy commands (default 1: offsets | ids staged in one
 * pinned block and copied once, the vectors stored by the last kernel straight into pinned host memory). */

/* BertModel::forward hidden states (model.rs:565-570) for tests: out [total_tokens][384]. */
int dawn_embedder_hidden_states(dawn_embedder *e, const uint32_t *token_ids, const int32_t *seq_offsets,
                                int B, float *out);
/* Test hook: one kernel of the forward in isolation — op 0 BertEmbeddings (model.rs:266-281: in = T token ids of one
 * sequence, out [T][384]); 1 LayerNorm(a + r) with layer 0's attention-output LayerNorm (:86-104,378: in = a | r, each
 * [T][384]); 2 / 3 layer 0's intermediate dense + activation (:425-430, :28-37: in [T][384], out [T][1536]; 2 = the form
 * the forward would take for T rows, 3 = the 64x64 f32-MFMA tile kernel); 4 / 5 the same layer through the bf16x3 kernel
 * (4: its f32 output, 5: its three-plane output summed). */
int dawn_embedder_debug_op(dawn_embedder *e, int op, const void *in, int T, float *out);
/* Timing hook: mean ms of one dense-layer shape of the model ([T x K] . [N x K]^T) over `iters` launches; variant 0 = the
 * f32-MFMA tile kernel, 1 = the bf16x3 kernel (f32-accurate 3-way bf16 split on the bf16 matrix cores). */
int dawn_embedder_debug_gemm_time(dawn_embedder *e, int T, int N, int K, int variant, int iters, double *mean_ms);

#if defined(__GNUC__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif
