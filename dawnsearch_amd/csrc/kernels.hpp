// kernels.hpp — launcher declarations for the gfx950 kernels of libdawn_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <mutex>

namespace dawn {

constexpr int EM = 384;        // src/search/vector.rs:26
constexpr int ROW_F4 = 96;     // float4 per row
constexpr int LIST = 64;       // shortlist length: one entry per lane of a wavefront
constexpr int ROW_PAD = 64;    // index allocations are padded to a multiple of this many rows (+ROW_PAD)

// Rigorous bound on |filter score - exact sequential-f32 dot| for vectors passing is_normalized
// (DESIGN.md §4.3): gamma_384 * 1.0201 (sequential, un-fused reference order) + gamma_16 * 1.0201
// (the filter's 8-deep FMA chain + 6-level tree) < 2.6e-5.
constexpr float FILTER_EPS_F32 = 2.6e-5f;
// f16 matrix-core filter (scan_batched.hip).  Rows and queries are scaled by 2^8 and rounded to f16 (relative
// error 2^-11 each, RNE): |sum(q~x~)/2^16 - sum(qx)| <= (2*2^-11 + 2^-22) * sum|q_i x_i| <= 9.97e-4 with
// sum|q_i x_i| <= 1.0201 (is_normalized gate).  Products of two f16 are exact in f32; the 384-term f32
// accumulation inside/between MFMAs adds <= 384 * 2^-23 * 1.0201 = 4.7e-5 (one ulp per add, any order);
// components below 2^-22 flushed as f16 subnormals (if the hardware does) add < 1e-5; the exact side's own
// gamma_384 * 1.0201 = 2.4e-5.  Total < 1.08e-3; 1.25e-3 is used.
constexpr float FILTER_EPS_F16 = 1.25e-3f;
// bf16 index, bf16 matrix cores.  The rows ARE bf16 (no row-side rounding); products of two bf16 are exact in f32.
// bf16 keeps 8 significant bits: round-to-nearest-even errs by <= 2^-8 relative.
//  * streaming filter: the query enters as hi + lo (hi = bf16(q), lo = bf16(q - hi)): |q - hi - lo| <= 2^-16 |q|, so the
//    representation error is <= 2^-16 * sum|q_i x_i| <= 1.6e-5 (sum|q_i x_i| <= 1.0201 * 1.004: is_normalized gate, row
//    norms moved by at most 2^-8 by their rounding); f32 accumulation of 384 exact products, any order: <= 2.4e-5; the
//    exact side's own gamma_384: 2.4e-5.  Total < 6.4e-5; 7e-5 is used.
//  * matrix-core filter (one bf16 image of the query): 2^-8 * 1.0201 * 1.004 = 4.0e-3 + the same 4.8e-5; 4.1e-3 is used
//    (measured worst case on planted near-duplicates: 1.65e-3, tests/test_scan_bf16_gpu.py).
constexpr float FILTER_EPS_BF16_STREAM = 7.0e-5f;
constexpr float FILTER_EPS_BF16_MFMA = 4.1e-3f;

constexpr int BATCH_TILE_ROWS = 64;   // rows per LDS tile of the batched scan
constexpr int BATCH_QT = 256;         // queries per batched pass (8 waves x 32)
constexpr int BATCH_CAP = 8192;       // candidate slots per query (and dense sample size)
constexpr int BATCH_CAND_SEGS = 16;   // segments (and counters) per query candidate buffer, scan_batched.hip

constexpr int ROW_F32 = 0;   // DAWN_DTYPE_F32: rows are 384 x f32 (1536 B)
// 16-bit rows are stored TILE BY TILE (64 rows = 48 KiB) in the operand order of v_mfma_f32_32x32x16_{f16,bf16}:
// tile T = rows 64T..64T+63; inside it fragment f = sub*24 + s (sub = 32-row half, s = k-step of 16) is 1 KiB =
// 64 lanes x 16 B, lane L = h*32 + r holding elements k = 16s + 8h .. +7 of row 64T + 32*sub + r.  A wave-wide 16-B load
// of a fragment IS the MFMA A operand (streaming filter, scan_kernels.hip) and an LDS-DMA of the tile needs no
// permutation (matrix-core filter, scan_batched.hip).  frag_chunk() in wave_topk.hpp maps (row, 16-B chunk) to its slot.
//   ROW_BF16: DAWN_DTYPE_BF16, the index itself: bf16 values (rounded to nearest even on add), scored exactly as their
//             f32 widening by the tail
//   ROW_F16S: filter-only shadow of an f32 index: f16(2^8 * x)
constexpr int ROW_BF16 = 1;
constexpr int ROW_F16S = 2;
// ROW_I8S: int8 filter-only shadow of an f32 index, quantised per 32-row sub-tile (12 fragments of 1 KiB in the operand
// order of v_mfma_i32_32x32x32_i8 + {scale, error bound} per sub-tile): scan_i8.hip.  Its filter score is an UPPER BOUND
// of the real dot product; FILTER_EPS_I8 covers the rounding of that bound's evaluation (1e-6), the f32 rounding of the
// rotation the shadow and the query images go through (2 x 1.1e-6 x 1.0201, scan_i8.hip) and the reference's own
// sequential-sum error (gamma_384 * 1.0201 = 2.34e-5): 2.67e-5.
constexpr int ROW_I8S = 3;
// packed 5- / 6-bit shadow of the rows (scan_i6.hip): 32-row sub-tiles of 7680 / 9216 B (240 / 288 B/row) + {1 / s, E} per sub-tile
constexpr int ROW_I6S = 4;
constexpr float FILTER_EPS_I8 = 2.9e-5f;

constexpr uint32_t FLAG_OK = 0;        // certificate holds: result is exact
constexpr uint32_t FLAG_FALLBACK = 1;  // certificate failed: the exact pass must (and will) run
constexpr uint32_t FLAG_SECOND = 2;    // first certificate failed, the 1024-deep second one held: result is exact
constexpr uint32_t FLAG_DEEP = 3;      // first certificate failed, a deeper round (128 .. 256 rows) held: result is exact
constexpr uint32_t FLAG_BOUNDED = 4;   // every certificate failed, the bounded exact pass (scan_bounded.hip) answered: result is exact
constexpr uint32_t FLAG_RERUN = 6;     // a batch's certificate failed at the sampled threshold; a second pass of the flagged queries with the
                                       // threshold their own k-th exact distance gives answered: result is exact (scan_i8.hip: launch_i8_rerun)
constexpr int N_STAT_SLOTS = 8;        // device-side counters per index, indexed by the final flag of a query ...
constexpr int STAT_PACKED_FAIL = 5;    // ... and [5]: single-query searches whose packed-stream certificate failed (merge_exact_kernel)
constexpr int STAT_BOUNDED_WIDE = 0;   // ... [0] (no final flag is 0): queries the WIDE batch form of the bounded pass answered (a subset of [4])
constexpr int STAT_BOUNDED_EXACT = 7;  // ... and [7]: (row, query) pairs the bounded pass scored exactly (mod 2^32; scan_bounded.hip)

// Function attributes (hipFuncSetAttribute: the dynamic-LDS limit of a kernel) belong to the CURRENT device's copy of the
// kernel: a process that drives several devices (dawn_sharded.cpp) has to set them on each.  once_per_device(state, fn) runs fn
// the first time it is called with a given device current; `state` is a per-call-site OncePerDevice.
struct OncePerDevice {
    std::mutex mu;
    uint64_t done = 0;  // bit d: device d has been served (devices >= 64 are served every time)
};
template <class Fn>
inline void once_per_device(OncePerDevice& st, Fn fn) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::lock_guard<std::mutex> g(st.mu);
    const uint64_t bit = dev >= 0 && dev < 64 ? (1ull << dev) : 0;
    if (bit && (st.done & bit)) return;
    fn();
    st.done |= bit;
}

struct BatchWorkspace {
    _Float16* qh;    // [BATCH_QT][384] scaled f16 queries (int8 path: int8 images [256][384] | {s_q, K2}[256])
    float* tau;      // [BATCH_QT]
    uint32_t* cnt;   // [BATCH_QT][BATCH_CAND_SEGS] candidates appended per segment
    void* cand;      // [BATCH_QT][BATCH_CAND_SEGS][BATCH_CAP / BATCH_CAND_SEGS] uint2 (score bits, row); first half doubles
                     // as the dense f32 score matrix [BATCH_QT][BATCH_CAP]
    // per-index tuning (dawn_index_set_option "mfma_sched" / "mfma_target")
    int sched = 4;    // 4 = default (16-bit rows: pipelined 4-wave LDS-DMA kernel for long passes, 8-wave kernel for short
                      // ones), 5 = pipelined kernel always, 1 = 8-wave kernel always, 0 = lockstep converting kernel on the
                      // f32 rows; builds with -DDAWN_EXPERIMENTS only: 2 = + stamps, 41..55 = timing experiments
    // candidates per query the sampled thresholds aim for (twice that for k > 32).  Rows that never become candidates are
    // only bounded by tau in the pass's own bound, so the certificates need tau a few hundred ranks below the k-th best
    // score even when the sampled estimate comes out low (plan_batched_tiles); a candidate costs the pass ~80 clk.
    int target = 1024;
    uint32_t* pool = nullptr;  // [32] chunk counters of the int8 append pass's dynamic tail (tau_select leaves them at zero)
    unsigned long long* diag = nullptr;  // DAWN_EXPERIMENTS, sched 2: [grid][8 waves][8] phase stamps
};
struct ScanGeom {
    int blocks;           // scan grid (== number of candidate lists per query)
    int threads;          // 1024 / 512 / 256
    int unroll = 2;       // row pairs per wave iteration (batch-1 kernel)
    int refine = 0;       // packed-shadow stream only: entries of its list a wave keeps and refines (0: chosen from N and k)
    int chunk = 16;       // packed-shadow stream only: sub-tiles per chunk of the dynamically assigned tail ...
    int dyn_share = 2;    // ... and the sixteenths of the index that tail covers
};

// Filter pass: approximate scores for all rows, per-block top-64 lists.
//   cand_s/cand_p: [B][geom.blocks][64]
void launch_scan_filter(const void* d_x, int dtype, uint32_t n_rows, const float* d_q, int B, float* cand_s,
                        uint32_t* cand_p, const ScanGeom& geom, hipStream_t stream, hipEvent_t ev0,
                        hipEvent_t ev1, uint32_t* pool = nullptr);
// The same over fragment-ordered 16-bit rows — rt = ROW_F16S: the scaled-f16 shadow of an f32 index (filter error
// FILTER_EPS_F16), ROW_BF16: a bf16 index (FILTER_EPS_BF16_STREAM) —, 8 queries per pass; d_q = the f32 queries
// (converted in the kernel).
void launch_scan_filter_f16s(const void* d_rows, int rt, uint32_t n_rows, const float* d_q, int B, float* cand_s,
                             uint32_t* cand_p, const ScanGeom& geom, hipStream_t stream, hipEvent_t ev0, hipEvent_t ev1);
// int8 shadow (scan_i8.hip): d_meta = float2 {scale, error bound} per 32-row sub-tile
void launch_scan_filter_i8s(const void* d_shadow, const void* d_meta, uint32_t n_rows, const float* d_q, int B, float* cand_s,
                            uint32_t* cand_p, const ScanGeom& geom, hipStream_t stream, hipEvent_t ev0, hipEvent_t ev1);
// levels: quantiser levels of the int8 shadow (127; index option "debug_i8_levels": fewer, experiments on coarser shadows)
void launch_rows_to_i8s(const void* d_rows, int rt, void* d_shadow, void* d_meta, size_t first_row, size_t n_valid,
                        hipStream_t stream, float levels = 127.0f);
// Packed shadow of 5 or 6 bits per component (scan_i6.hip): conversion, and the whole single-query search on it (stream with
// exact rescoring of every workgroup's shortlist(s) in its epilogue, merge of the exact lists + certificate; merge = false:
// the stream alone).  d_i8 / d_i8meta: the int8 shadow, which refines the bounds of the listed rows; tb [blocks].
void launch_rows_to_i6s(const void* d_rows, int rt, int bits, void* d_shadow, void* d_meta, size_t first_row, size_t n_valid,
                        hipStream_t stream);
// entries of its coarse list a wave keeps (8 .. 64), chosen from the index size and k; 0: the packed stream cannot certify this
// search (too many rows above the bound for a 64-entry list: large k on a very large index) — stream the int8 shadow instead
// slack: the packed shadow's own error bounds, measured (I6Slack below); nullptr: the constants of an unmeasured shadow
struct I6Slack {
    // share of the shadow's sub-tiles whose E (their bound on ||x' - s X||_2, the coarse score's offset above the true one) lies in
    // [b, b + 1) x I6_SLACK_STEP; the last bin holds everything above.  version: bumped whenever the histogram is re-read.
    uint32_t version = 0;
    float frac[64] = {};
};
constexpr float I6_SLACK_STEP = 0.004f;
// histogram of E over the n_sub sub-tiles of a packed shadow's meta array -> hist [64] (zeroed here)
void launch_i6_slack_hist(const void* d_meta, uint32_t n_sub, uint32_t* d_hist, hipStream_t stream);
int i6_refine_count(uint32_t n_rows, uint32_t k, int bits, int waves, const I6Slack* slack = nullptr);
void launch_scan_i6(const void* d_i6, const void* d_meta, int bits, const void* d_i8, const void* d_i8meta, const void* d_rows,
                    int dtype, const uint64_t* d_ids, uint32_t n_rows, const float* d_q, float* ub_s, uint32_t* ub_p, float* ex_s,
                    uint32_t* ex_p, float* tb, uint32_t* pool, const ScanGeom& g,
                    uint32_t k, uint64_t* d_labels, float* d_dist, uint32_t* d_found, uint32_t* d_flags, int force_fallback,
                    bool merge, hipStream_t stream, hipEvent_t ev0, hipEvent_t ev1, uint32_t* d_stats = nullptr,
                    bool central_tail = false);
void launch_prep_queries(const float* d_q, int B, const BatchWorkspace& ws, hipStream_t stream);
// Merge the lists, rescore the 64 survivors exactly (reference order), certify, write results.  list_bounds [n_lists] (B == 1
// only; the packed stream's tail): list l's own bound on the rows it does not hold, where that is not its 64th entry;
// d_stats_packed: counters whose [STAT_PACKED_FAIL] is bumped when the query ends flagged for the ladder.
void launch_merge_rescore(const void* d_x, int dtype, const uint64_t* d_ids, uint32_t n_rows, const float* d_q, int B,
                          const float* cand_s, const uint32_t* cand_p, int n_lists, uint32_t k,
                          uint64_t* d_labels, float* d_dist, uint32_t* d_found, uint32_t* d_flags,
                          int force_fallback, float eps, hipStream_t stream, uint32_t* pool = nullptr,
                          const float* list_bounds = nullptr, uint32_t* d_stats_packed = nullptr);
// Batched search on the matrix cores (mfma_min_batch <= B <= BATCH_QT) over 16-bit rows: f16 / bf16 MFMA filter with sampled thresholds,
// candidate append, exact rescore + certificate (scan_batched.hip).  ev0/ev1 bracket the full pass.
struct BatchPlan {
    bool dense_only;          // n_rows <= BATCH_CAP: one dense pass, no thresholds
    uint32_t n_tiles_total;
    uint32_t s1_tiles, s1_stride, m1;   // dense sample, tau = m1-th largest
    uint32_t s2_tiles, s2_stride, m2;   // appended sample (0 = skipped), tau = m2-th largest
};
BatchPlan plan_batched(uint32_t n_rows, int target, uint32_t k);
BatchPlan plan_batched_tiles(uint32_t n_rows, uint32_t tile_rows, int target, uint32_t k);
// pieces of the batched tail shared with the int8 path (scan_i8.hip)
void launch_tau_select(bool dense_pass, int B, const BatchWorkspace& ws, uint32_t dense_count, uint32_t m, hipStream_t stream);
void launch_select_rescore_eps(bool dense_pass, const void* d_x, int dtype, const uint64_t* d_ids, uint32_t n_rows,
                               const float* d_q, int B, uint32_t k, const BatchWorkspace& ws, uint64_t* d_labels, float* d_dist,
                               uint32_t* d_found, uint32_t* d_flags, int force_fallback, float eps, hipStream_t stream, int rerun = 0);
// (rerun = 1: only queries flagged FLAG_FALLBACK are looked at; the ones this tail settles become FLAG_RERUN)
// The int8 matrix-core pass once more for the FLAGGED queries of a batch, each with the threshold its own k-th exact distance gives
// (1 - d_k - margin: the bounded pass's arithmetic) instead of the sampled one; d_go: 4 + BATCH_QT device words (scratch).
void launch_i8_rerun(const void* d_x, int dtype, const void* d_i8, const void* d_meta, const uint64_t* d_ids, uint32_t n_rows,
                     const float* d_q, int B, uint32_t k, const BatchWorkspace& ws, int grid, uint64_t* d_labels, float* d_dist,
                     uint32_t* d_found, uint32_t* d_flags, uint32_t* d_go, hipStream_t stream);
void launch_batched_full_pass_i8(const void* d_i8, const void* d_meta, uint32_t n_rows, int B, const BatchWorkspace& ws, int grid,
                                 int iters, hipStream_t stream, hipEvent_t ev0, hipEvent_t ev1);
void launch_batched_dense_scores_i8(const void* d_i8, const void* d_meta, uint32_t n_rows, const float* d_q, int B,
                                    const BatchWorkspace& ws, int grid, hipStream_t stream);
// Batched search on the int8 shadow (scan_i8.hip): the sequence of launch_scan_batched over 128-row int8 tiles
void launch_scan_batched_i8(const void* d_x, int dtype, const void* d_i8, const void* d_meta, const uint64_t* d_ids,
                            uint32_t n_rows, const float* d_q, int B, uint32_t k, const BatchWorkspace& ws, int grid,
                            uint64_t* d_labels, float* d_dist, uint32_t* d_found, uint32_t* d_flags, int force_fallback,
                            hipStream_t stream, hipEvent_t ev0, hipEvent_t ev1);
// FP6 (e2m3) shadow of the rows and the batched search that uses it as its FIRST filter (scan_f6.hip)
struct F6Workspace {
    void* qf6 = nullptr;         // [16 groups][3 k-steps][64 lanes][6 dwords] e2m3 query images (B-operand order)
    void* qmeta = nullptr;       // [BATCH_QT] float2 {s_q, ||dq||_2}
    float* tau6 = nullptr;       // [BATCH_QT]
    void* cand_big = nullptr;    // [BATCH_QT][BATCH_CAND_SEGS][seg_cap_big] uint2 (bound bits, row): survivors of the FP6 pass
    uint32_t* cnt_big = nullptr; // [BATCH_QT][BATCH_CAND_SEGS], zero between searches
    uint32_t seg_cap_big = 2048;
    int target = 12288;          // survivors per query the FP6 threshold aims for (twice that for k > 32); option "f6_target"
    int refine_rows = 1;         // f32 index: survivors are re-scored on the rows themselves (0: on the int8 shadow); option "f6_refine_rows"
    int stagger = -1;            // < 0: the LDS-staged pass; >= 0: the register-ring pass, its waves this many tiles apart; option "f6_stagger"
};
void launch_rows_to_f6s(const void* d_rows, int rt, void* d_shadow, void* d_meta, size_t first_row, size_t n_valid, hipStream_t stream);
void launch_f6_dense_scores(const void* d_f6, const void* d_meta, uint32_t n_rows, const float* d_q, int B, void* d_qf6, void* d_qmeta,
                            float* d_out, hipStream_t stream);
void launch_scan_batched_f6(const void* d_x, int dtype, const void* d_i8, const void* d_i8meta, const void* d_f6, const void* d_f6meta,
                            const uint64_t* d_ids, uint32_t n_rows, const float* d_q, int B, uint32_t k, const BatchWorkspace& ws,
                            const F6Workspace& f6, int grid, uint64_t* d_labels, float* d_dist, uint32_t* d_found, uint32_t* d_flags,
                            int force_fallback, hipStream_t stream, hipEvent_t ev0, hipEvent_t ev1);
// the int8 batched path's sampling passes alone: queries -> int8 images, ws.tau = its sampled thresholds, counters at zero
void launch_i8_sample_thresholds(const void* d_i8, const void* d_meta, uint32_t n_rows, const float* d_q, int B, uint32_t k,
                                 const BatchWorkspace& ws, int grid, hipStream_t stream);
// Test hook: dense filter scores of rows [0, min(n_rows, BATCH_CAP)) -> ws.cand viewed as float [BATCH_QT][BATCH_CAP].
void launch_batched_dense_scores(const void* d_x, int dtype, uint32_t n_rows, const float* d_q, int B,
                                 const BatchWorkspace& ws, int grid, hipStream_t stream);
void launch_batched_full_pass(const void* d_frows, int frt, uint32_t n_rows, int B, const BatchWorkspace& ws, int grid,
                              int iters, hipStream_t stream, hipEvent_t ev0, hipEvent_t ev1);
int batched_init();  // raises the dynamic-LDS limit of the scan kernels; 0 or a hipError_t
void launch_scan_batched(const void* d_x, int dtype, const void* d_frows, int frt, const uint64_t* d_ids, uint32_t n_rows,
                         const float* d_q, int B, uint32_t k, const BatchWorkspace& ws, int grid, uint64_t* d_labels,
                         float* d_dist, uint32_t* d_found, uint32_t* d_flags, int force_fallback, hipStream_t stream,
                         hipEvent_t ev0, hipEvent_t ev1);
void launch_rows_f32_to_f16s(const float* d_rows, void* d_shadow, size_t first_row, size_t n_valid, hipStream_t stream);
// Exact fallback (predicated per query on d_flags[b] == FLAG_FALLBACK): per-workgroup exact lists, merged and written out
// by the last workgroup to arrive (d_done[B]: arrival counters, zero before and after).
// stats_mirror (may be NULL): device-visible host memory [N_STAT_SLOTS] that receives a copy of d_stats at the end of the search
void launch_scan_exact(const void* d_x, int dtype, const uint64_t* d_ids, uint32_t n_rows, const float* d_q, int B,
                       const uint32_t* d_flags, uint32_t* d_done, uint32_t* d_stats, float* cand_s, uint32_t* cand_p,
                       int n_lists, uint32_t k, uint64_t* d_labels, float* d_dist, uint32_t* d_found, hipStream_t stream,
                       uint32_t* stats_mirror = nullptr);

// Bounded exact pass (scan_bounded.hip; predicated per query on d_flags[b] == FLAG_FALLBACK): streams
// the int8 shadow, scores exactly every row whose upper bound can still reach the k-th best distance known so far (d_dist of
// the failed stage), sets FLAG_BOUNDED.  cand_s / cand_p [B][n_lists][64]; d_done [B] arrival counters (zero before and after).
// It is the LAST launch of a search (no exact pass behind it): d_stats / stats_mirror as for launch_scan_exact.
// Per-index choices of the pass (dawn_index::bounded; options "bounded_ring" / "bounded_multi_waves" / "bounded_multi_packed" /
// "bounded_wide") and the workspace of its wide batch form.
constexpr uint32_t BOUNDED_WIDE_CAP = 2048;  // exact results a query of the wide form may collect (more: the 16-query form answers it)
struct BoundedOpts {
    int ring = 6;          // 16-B fragments a wave keeps in flight ahead of its MFMAs: 6 (half a sub-tile) or 12
    int multi_waves = 8;   // waves per workgroup of the 16-query batch form: 4 or 8 (one workgroup per CU either way: its LDS)
    int multi_packed = 0;  // 1: the 16-query batch form streams the packed 5-bit shadow where one is passed
    int wide = 1;          // 1: a batch's flagged queries go through the WIDE form first — 64 queries per stream of the int8 shadow
    uint2* wide_res = nullptr;     // [kBoundedMaxFlags = 256][BOUNDED_WIDE_CAP] {distance bits, row}: exact results appended by the wide form
    uint32_t* wide_cnt = nullptr;  // [256] entries appended per query; zero between searches
};
void launch_scan_bounded(const void* d_i8, const void* d_i8meta, const void* d_x, int dtype, const uint64_t* d_ids, uint32_t n_rows,
                         const float* d_q, int B, uint32_t* d_flags, uint32_t* d_done, float* cand_s, uint32_t* cand_p,
                         int n_lists, uint32_t k, uint64_t* d_labels, float* d_dist, uint32_t* d_found, hipStream_t stream,
                         uint32_t* d_stats, uint32_t* stats_mirror, const BoundedOpts& opts, const void* d_i5 = nullptr,
                         const void* d_i5meta = nullptr);
// (d_i5 / d_i5meta: the packed 5-bit shadow of the same rows, or NULL — a single query (B = 1) then streams it, 240 B per row,
// instead of the int8 shadow)
// ... as the WHOLE search of one query (a demoted index, dawn_index.cpp: ladder feedback): the flag is raised and the threshold
// starts at +inf — the waves' own k-th best distances are the thresholds.  ev0 / ev1 bracket the pass.
void launch_scan_bounded_direct(const void* d_i8, const void* d_i8meta, const void* d_x, int dtype, const uint64_t* d_ids,
                                uint32_t n_rows, const float* d_q, uint32_t* d_flags, uint32_t* d_done, float* cand_s,
                                uint32_t* cand_p, int n_lists, uint32_t k, uint64_t* d_labels, float* d_dist, uint32_t* d_found,
                                hipStream_t stream, hipEvent_t ev0, hipEvent_t ev1, uint32_t* d_stats, uint32_t* stats_mirror,
                                const BoundedOpts& opts, float first_threshold = __builtin_inff(), const void* d_i5 = nullptr,
                                const void* d_i5meta = nullptr, bool seeded = false);
// (seeded: d_dist[k - 1] already holds the k-th exact distance of a search over PART of the rows — a valid first threshold; it is kept)

// Stable G-way merge of per-shard results (multi-GPU).  pos_to_label != NULL: the incoming labels are global insertion
// positions — ties go to the lower position and the winners are translated through the table.
constexpr size_t kMaxMergeCands = 64 * 64;  // G * k a merge launch accepts (48 KiB of LDS at most)
void launch_shard_merge(size_t G, size_t B, size_t k, const uint64_t* in_labels, const float* in_dist,
                        const uint32_t* in_found, size_t sl, size_t sd, size_t sf, const uint64_t* pos_to_label,
                        uint64_t* out_labels, float* out_dist, uint32_t* out_found, hipStream_t stream);
// PageEntry records (1568 B, vector at byte 16: src/index/warc.rs:35-43) -> packed f32 rows
void launch_page_entries_to_rows(const void* d_records, uint32_t n, float* d_rows, hipStream_t stream);

// is_normalized (vector.rs:185-192) over n rows; *d_bad_count += number of failing rows.
void launch_validate_rows(const float* d_rows, uint32_t n, uint32_t* d_bad_count, hipStream_t stream);
// Synthetic unit rows: rows first_row.. of stream seed -> d_out[n][384]; d_len scratch [n].
// dist: 0 = the spec's uniform rows; 1 / 2 / 3 = Gaussian / heavy-tailed bench distributions (scan_kernels.hip)
void launch_fill_synth(uint64_t seed, uint64_t first_row, uint32_t n, float* d_out, float* d_len, int dist,
                       hipStream_t stream);
void launch_iota_u64(uint64_t* d_out, uint64_t first, uint32_t n, hipStream_t stream);
// bf16 index rows <-> f32 rows: d_in[n][384] f32 -> rows first_row.. of the fragment-ordered index d_x (round to nearest
// even), and back (exact widening).
void launch_rows_f32_to_bf16(const float* d_in, void* d_x, size_t first_row, size_t n_rows, hipStream_t stream);
void launch_rows_bf16_to_f32(const void* d_x, size_t first_row, float* d_out, size_t n_rows, hipStream_t stream);

}  // namespace dawn
