// index_internal.hpp — the index object behind the dawn_index_* C ABI and the pieces dawn_index.cpp (one device) and
// dawn_sharded.cpp (one process, several devices) share.  Not part of the ABI.
#pragma once
#include <algorithm>
#include <cstdint>
#include <exception>
#include <new>
#include <utility>
#include <vector>

#include "common.hpp"
#include "kernels.hpp"

namespace dawn {
struct ShardSet;  // dawn_sharded.cpp

constexpr size_t kMaxBatch = 256;        // queries per internal pass of the host API
constexpr size_t kMaxProfile = 4096;     // kept event pairs
constexpr size_t kZeroCopyBatch = 256;   // host API: up to this many queries get their results by zero-copy stores (option "zero_copy_batch";
                                         // 8 until round 5: three copy commands behind the last kernel cost a 256-batch ~20 us)
constexpr size_t kShadowSmallRows = 6u << 20;  // below this the shadow stream uses geom_h_small
constexpr size_t kI6MinRows = 3u << 18;        // indexes of at least this many rows keep the packed shadow for their single queries
                                               // (768 Ki; round 5, lists of 24: 0.5 M rows 0.076 ms per search against the int8 stream's 0.071,
                                               // 1 M 0.094 / 0.100, 2 M 0.133 / 0.156, 3 M 0.164 / 0.211: profiles/r05/i6_min_rows_probe.log;
                                               // rounds 3-4, lists of 40-64: 2 Mi)
constexpr size_t kI6SmallRows = 32u << 20;     // below this the packed stream uses geom_i6_small
constexpr size_t kAddStageRows = 1024;   // single-row adds staged on the host before they travel together
constexpr size_t kStageChunk = 1u << 18;  // rows of device staging at most (bf16 adds / PageEntry records / get_rows)
constexpr uint32_t kFbWindow = 32;        // ladder feedback: packed single-query searches per window
constexpr double kFbBoost = 0.05, kFbDemote = 0.35;  // failure rates that make the waves refine full lists / demote the index
// ... an index of >= 40 Mi rows (its bounded pass runs on the packed shadow, seeded): packed stream 3.5 ms + 4.2 ms behind a failure
// against 4.17 ms for the seeded pass alone on 100 M rows — break-even at a failure rate of 0.16 (12.5 M rows: 0.30, 5 M: 0.37)
constexpr double kFbDemotePacked = 0.18;
constexpr uint32_t kFbDemoteMin = 256, kFbDemoteMax = 8192;
constexpr uint32_t kBatchFbWindow = 1024;  // int8 batched pass: queries per feedback window
constexpr double kBatchFbBoost = 0.10;
constexpr int kBatchBoostTarget = 4096, kBatchBoostTargetWide = 3072;  // (k > 32 doubles the target: 2 x 3072 stays inside the 8192 slots)
constexpr size_t kF6AutoHeadroom = (size_t)24 << 30;  // "f6_shadow" = auto: HBM that must stay free once the FP6 shadow is built
constexpr uint32_t kF6FbWindow = 1024;  // FP6 feedback: queries per window
constexpr double kF6FbSuspend = 0.30;
constexpr uint32_t kF6FbSuspendMin = 16, kF6FbSuspendMax = 1024;
}  // namespace dawn

struct dawn_index {
    int device = 0;
    size_t dims = DAWN_EM_LEN;
    hipStream_t stream = nullptr;

    int dtype = DAWN_DTYPE_F32;  // row storage: f32 (1536 B/row) or bf16 (768 B/row)
    char* d_x = nullptr;         // [(cap_phys + ROW_PAD)][384] of dtype
    // f32 index only: scaled-f16 shadow copy of the rows (f16(2^8 x), 768 B/row, tiles in MFMA-fragment order: ROW_F16S
    // in kernels.hpp) read by the 16-bit matrix-core FILTER instead of the f32 rows.  Only kept when the int8 shadow is
    // switched off or does not fit; results stay exact (the rescore reads the f32 rows).
    char* d_shadow = nullptr;
    size_t shadow_cap = 0;       // rows allocated
    size_t shadow_rows = 0;      // rows converted so far (prefix)
    int use_shadow = 1;          // option "f16_shadow"
    int shadow_small_batches = 1;  // option "f16_shadow_b1": batches of 1..8 queries also filter on a shadow
    // geometry of the shadow stream (MFMA from registers): one 2-wave block per CU, `unroll` picks the load schedule
    // (launch_filter_f16s_qb: 3 = ring of 12 fragments = 12 KiB in flight per wave).  tools/scan_sweep_shadow.py,
    // 80M rows: 7.02-7.07 TB/s; every schedule with 2-4 waves per CU lands within 1 % of it
    dawn::ScanGeom geom_h{256, 128, 3};
    // ... and below kShadowSmallRows rows (a few dozen sub-tiles per wave: start-up, tail and load balance count)
    // one 8-wave block per CU: 1M rows 154 -> 130 us.  Setting any shadow_scan_* option pins geom_h for every size.
    dawn::ScanGeom geom_h_small{256, 512, 3};
    bool geom_h_pinned = false;
    const dawn::ScanGeom& shadow_geom() const {
        return (!geom_h_pinned && size < dawn::kShadowSmallRows) ? geom_h_small : geom_h;
    }
    // int8 shadow stream: the software-pipelined kernel (scan_filter_i8s_pipe_kernel), 4 waves per CU x a ring of 6
    // fragments = 24 KiB in flight per CU (unroll code 8).  tools/stream_pipe_ab.py, 100 M rows, interleaved rounds
    // (profiles/r03/stream_pipe_ab_100M*.log): 5.40 ms = 7.12 TB/s = 0.890 of spec; 2 waves x 12: 5.44-5.45 (0.883); the round-2
    // kernel (code 3, 4 waves x 12 KiB): 5.48-5.55 (0.865-0.876); 8 waves x 6: 5.49-5.52; rings of 3 / 4: 5.54-5.58.  The bare read
    // of the same stream (tools/probes/hbm_read.hip) tops out at 7.17-7.23 TB/s: fewer bytes in flight are FASTER on this
    // chip as long as the request stream never pauses.
    dawn::ScanGeom geom_i8{256, 256, 8};
    // ... and the round-2 kernel at 8 waves per CU x 12 KiB below 6 M rows (a few dozen sub-tiles per wave: start-up and
    // balance count; 1 M rows: 75 us against 82 for the pipelined form at 4 waves).  12.5 M rows — one shard of 100 M on 8
    // GPUs —: pipelined 4 x 6 694 us, round-2 8 x 12 702 us (profiles/r03/stream_pipe_ab_12p5M.log)
    const dawn::ScanGeom& i8_geom() const {
        return geom_h_pinned ? geom_h : size < dawn::kShadowSmallRows ? geom_h_small : geom_i8;
    }
    bool shadow_failed = false;  // allocation failed once: do not retry until the index is re-created
    // int8 shadow of the index rows (ROW_I8S, scan_i8.hip: 384 B/row + 8 B per 32 rows; f32 and bf16 indexes alike) read by
    // every filter — the streaming one of single queries and the matrix-core pass — instead of the rows: a quarter of the
    // f32 bytes.  Kept current by every mutation (add / add_batch / fill / load / reserve: the last sub-tile is
    // re-quantised on add, everything on growth), so a search is launches only; if it cannot be allocated (or
    // "i8_shadow" = 0) the filters fall back to the f16 shadow / the rows.
    char* d_i8 = nullptr;
    float* d_i8meta = nullptr;
    size_t i8_cap = 0, i8_rows = 0;
    int use_i8 = 1;              // option "i8_shadow"
    int i8_batched = 1;          // option "i8_batched": batches of mfma_min_batch and more also filter on it
    bool i8_failed = false;
    float i8_levels = 127.0f;    // option "debug_i8_levels" (experiments on coarser shadows)
    // Packed shadow (ROW_I6S, scan_i6.hip: 5 bits per component = 240 B/row, or 6 = 288 B/row, + 8 B per 32 rows) read by the
    // single-query stream of an index of at least i6_min_rows rows instead of the int8 shadow (which the matrix-core pass keeps
    // using): the stream is HBM-bound, fewer bytes per row are the only thing that makes it faster.  Kept current by the
    // mutations like the int8 shadow; if it cannot be allocated (or "i6_shadow" = 0) single queries stream the int8 shadow.
    char* d_i6 = nullptr;
    float* d_i6meta = nullptr;
    size_t i6_cap = 0, i6_rows = 0;
    int use_i6 = 1;              // option "i6_shadow"
    int i6_bits = 5;             // option "i6_bits": 5 (default) or 6
    size_t i6_row_bytes() const { return i6_bits == 6 ? 288 : 240; }
    size_t i6_min_rows = dawn::kI6MinRows;  // option "i6_min_rows" (tests: 0)
    bool i6_failed = false;
    size_t zero_copy_batch = dawn::kZeroCopyBatch;  // option "zero_copy_batch" (0: results always come back by copy commands)
    // The shadow's own error bounds, measured: histogram of its sub-tiles' E, re-read whenever the shadow changes (i6_shadow_sync);
    // i6_refine_count sizes the waves' lists from it.  Option "i6_slack_model" = 0: the constants of rounds 3-4 (A/B).
    dawn::I6Slack i6_slack;
    uint32_t* d_i6hist = nullptr;
    bool i6_slack_dirty = true;
    int i6_slack_model = 1;
    const dawn::I6Slack* i6_slack_ptr() const { return i6_slack_model && i6_slack.version ? &i6_slack : nullptr; }
    // its stream: `threads` / 64 waves per CU, `unroll` = loads in flight per wave (options "i6_scan_threads", "i6_scan_ring");
    // exact lists of the workgroups' epilogues [lists][64].  6-bit form, tools/stream_i6_ab.py, 100 M rows, three boxes
    // (profiles/r03/stream_i6_ab_100M_*.log, stream_i6_parts_off_100M.log; us per launch, the int8 stream on the same box
    // 5542 / 5585 / -): 8 waves x 4 fragments 4237 / - / 4130, 8 x 3 4194 / - / 4196, 8 x 6 4288 / - / 4149, 4 x 12 4302 / - / 4146,
    // 3 x 12 4304 / - / 4152, 4 x 6 4417, 2 x 12 4677: everything with >= 24 KiB in flight per CU lands within 2 %.  Below
    // kI6SmallRows rows 8 waves x 12 fragments (72 KiB in flight per CU: a short stream is start-up and tail, it wants its requests
    // out at once): 1 M rows 79 us against 88 (8 x 4), 3 M 158 / 172, 12.5 M 566 / 574 — and against the int8 stream's
    // 78 / 186 / 725 us, whose one-workgroup tail costs ~8 us more per search than merge_exact_kernel
    // (profiles/r03/stream_i6_ab_small_sizes.log).  5-bit form (the ring codes 4 / anything else = 4 / 8 loads = half a sub-tile
    // / a whole one ahead; profiles/r03/stream_i5_ab_*.log): 100 M rows 8 waves x 4 loads 3551 us, 4 x 8 3523 (but half the waves
    // = half the depth of the coarse lists), 8 x 8 3622, 6 x 8 3670, int8 stream 5607; 12.5 M: 8 x 8 511, 8 x 4 516, int8 714;
    // 3 M: 159 / 174 / 186; 1 M: 90 / 95 / 75.
    // The dynamically assigned tail (scan_i6.hip): chunks of 16 sub-tiles over the last 2/16 of a long stream; below kI6SmallRows
    // chunks of 8 over the last 3/16 (12.5 M rows: stream kernel 496 -> 485 us; 100 M rows: +0.3 %).  Smaller chunks LOSE: a
    // chunk costs a same-address scalar atomic, which this chip retires at ~8 M/s per address — 12.5 M rows with chunks of 4 / 2 / 1:
    // +1 / +12 / +37 %, 100 M rows: +5 / +20 / +50 % (tools/stream_chunk_ab.py, profiles/r04/stream_dyn_chunk_ab.log).
    dawn::ScanGeom geom_i6{256, 512, 4, 0, 16, 2};
    dawn::ScanGeom geom_i6_small{256, 512, 12, 0, 8, 3};
    bool geom_i6_pinned = false;
    const dawn::ScanGeom& i6_geom() const { return (!geom_i6_pinned && size < dawn::kI6SmallRows) ? geom_i6_small : geom_i6; }
    // FP6 (e2m3) shadow (ROW_F6S, scan_f6.hip: 288 B/row + 8 B per 16 rows): the FIRST filter of batches on a large index —
    // 1.5 x the int8 matrix rate under the chip's power envelope; its survivors are re-scored on the int8 shadow.  Option
    // "f6_shadow" (default 2 = auto: built from f6_min_rows rows where it leaves kF6AutoHeadroom of HBM free; its feedback suspends it on an
    // index whose FP6-filtered queries end in the ladder); kept current by the mutations like the others.
    char* d_f6 = nullptr;
    float* d_f6meta = nullptr;
    size_t f6_cap = 0, f6_rows = 0;
    int use_f6 = 2;              // option "f6_shadow": 0 never, 1 always, 2 auto (default): from f6_min_rows rows, where it fits with kF6AutoHeadroom to spare
    // option "f6_min_rows": batches of smaller indexes take the int8 pass.  The re-scoring of the filter's ~12 k survivors per
    // query costs 0.8 ms per batch of 256 whatever the index size: a tie at 50 M rows (4.79 ms both), -10 % at 100 M, + 55 % at
    // 12.5 M (2.00 against 1.29 ms; profiles/r04/f6_ab_12p5M_v8.log)
    size_t f6_min_rows = 64u << 20;
    bool f6_failed = false;
    dawn::F6Workspace f6ws{};
    float* d_cand_es = nullptr;
    float* d_cand_tb = nullptr;   // [blocks] the workgroups' bounds on their unlisted rows
    uint32_t* d_i6_pool = nullptr;  // [32] chunk counters of the dynamically assigned tail of the packed and the f32-row streams
    int stream_dyn_tail = 1;        // option "stream_dynamic_tail"
    int i6_central_tail = 0;        // option "i6_central_tail": the packed stream hands its refined lists to merge_rescore_kernel
    uint32_t* d_cand_ep = nullptr;
    int debug_fail_alloc = 0;    // option "debug_fail_alloc" (tests): bit 0 / 1 / 2 = the int8 / f16 / 6-bit shadow allocation fails
    float* d_stage = nullptr;    // device staging ([stage_bytes]): bf16 adds / get_rows / fill, PageEntry records
    size_t stage_bytes = 0;
    size_t row_bytes() const { return dtype == DAWN_DTYPE_BF16 ? dawn::EM * 2 : dawn::EM * 4; }
    uint64_t* d_ids = nullptr;  // [cap_phys]
    size_t size = 0;
    size_t pending = 0;       // rows enqueued by index_append_async past `size`, not committed yet
    size_t cap_reported = 0;  // what reserve() promised (usearch semantics)
    size_t cap_phys = 0;      // rows actually allocated (geometric growth)

    // search workspaces (allocated at creation for batches of up to kMaxBatch queries: a search is launches only)
    // batch-1..8 streaming scan: one 4-wave block per CU, 3 row pairs (9 KiB) in flight per wave.  Measured on
    // MI355X (tools/scan_sweep.py, 40M rows): 36 KiB in flight per CU reads 7.17 TB/s; the full-occupancy
    // geometry (32 waves, 192 KiB per CU) only 6.55 TB/s.
    dawn::ScanGeom geom{256, 256, 3};
    size_t ws_B = 0;
    size_t ws_lists = 0;      // candidate lists per query the workspace was sized for
    float* d_cand_s = nullptr;
    uint32_t* d_cand_p = nullptr;
    uint32_t* d_flags = nullptr;   // [ws_B] certificate flags | [ws_B] arrival counters of the exact pass
    uint32_t* d_stats = nullptr;   // [4]: queries that ended with FLAG_FALLBACK ([1]) / FLAG_SECOND ([2]) / FLAG_DEEP ([3]), counted on the device
    dawn::BatchWorkspace bws{};    // matrix-core batched path (+ per-index "mfma_sched" / "mfma_target")
    int mfma_blocks = 256;   // one 8-wave workgroup per CU
    // B >= this goes to the matrix-core filter (sampled thresholds, one candidate buffer per query); below it the
    // streaming filter keeps per-wave top-64 lists, whose warm-up grows with every extra query
    // int8 shadow (tools/small_batch_paths.py, stream / matrix-core ms): 1M rows B=1 0.136 / 0.166, B=2 0.199 / 0.170,
    // B=3 0.251 / 0.169; 40M rows B=1 2.27 / 2.33, B=2 2.36 / 2.35, B=3 2.42 / 2.34: two queries and more take the pass
    // (round 5: 0 = by index size, index_search_on_device; the table above is round 2's)
    int mfma_min_batch = 0;
    // host-API staging
    float* d_q = nullptr;
    uint64_t* d_labels = nullptr;
    float* d_dist = nullptr;
    uint32_t* d_found = nullptr;
    uint32_t* d_bad = nullptr;
    void* h_pinned = nullptr;  // kMaxBatch * (384*4 + 64*8 + 64*4 + 4 + 4)
    size_t h_pinned_bytes = 0;

    bool profiling = false;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> events;
    size_t events_used = 0;
    uint64_t n_searches = 0;
    int force_fallback = 0;      // option "force_fallback": 1 = every query takes the exact pass, 2 = every certificate fails (ladder)
    // option "batch_rerun": a second matrix-core pass for a batch's flagged queries with exact-derived thresholds (scan_i8.hip:
    // launch_i8_rerun): 0 never (default), 1 on ladder-heavy indexes, 2 always.  Measured on 100 M topical rows
    // (profiles/r04/batch_rerun_ab_100M.log): it settles 18-30 % of a batch at the default threshold depth (74 -> 69, 84 -> 72 ms) but
    // only 4-19 % once the batch feedback has deepened the thresholds, which alone gets 63-72 ms — the second pass costs 12 ms, two
    // bounded streams' worth, and ~35 % of such a batch (the near-tie shells) overflows any candidate buffer.  Kept for indexes where
    // the split is different; off by default.
    int batch_rerun = 0;
    int bounded_seed_shift = 5;  // option "bounded_seed_shift": the seed searches the first n >> shift rows
    int bounded_seed = 1;        // option "bounded_seed": a demoted query's packed bounded pass is seeded by a search over 1/32 of the rows
    int bounded_packed = 1;      // option "bounded_packed": the bounded pass of a single query streams the packed 5-bit shadow
                                 // (240 B/row): 0 never, 1 from 2 Mi rows, 2 always (dawn_index.cpp: bounded_packed_wanted)
    int bounded_pass = 1;        // option "bounded_pass": a failed certificate is answered from the int8 shadow (scan_bounded.hip)
    // the pass's per-index choices and the result buffers of its wide batch form (options "bounded_ring", "bounded_multi_waves",
    // "bounded_multi_packed", "bounded_wide"; buffers owned by ensure_workspace)
    dawn::BoundedOpts bounded{};
    // Ladder feedback.  The packed stream (240 B/row) followed by the bounded pass (384 B/row) costs 2.6 x the packed stream
    // when its certificate fails; the bounded pass ALONE, started without a threshold, costs 1.6 x and cannot fail.  The index
    // therefore watches how often the packed certificate fails — the device mirrors its counters into h_stats at the end of
    // every search (scan_exact_kernel; pinned memory, no synchronisation, the host reads whatever has landed) — in windows of
    // kFbWindow single-query searches: above kFbBoost the waves refine full lists (n_refine = 64), above kFbDemote the next
    // demote_len single queries go to the bounded pass directly (doubling while the next probe window fails again, back to
    // kFbDemoteMin once one passes).  Results never depend on it.  Reset by options, load / clear and growth by an eighth (index_prepare_search).
    uint32_t* h_stats = nullptr;   // [N_STAT_SLOTS] pinned, device-visible
    int ladder_feedback = 1;       // option "ladder_feedback": 0 off, 1 adaptive, 2 every single query goes to the bounded pass directly
    int debug_bad_threshold = 0;   // option "debug_bad_threshold" (tests)
    struct LadderFeedback {
        uint64_t issued = 0, win_issued0 = 0;  // packed single-query searches issued / at the start of the window
        uint32_t win_fail0 = 0;                // h_stats[STAT_PACKED_FAIL] at the start of the window
        uint32_t demote_left = 0, demote_len = 256;
        bool boosted = false;
    } fb;
    size_t fb_rows0 = 0;           // rows when the feedback last started over (appends keep it until the index has grown by an eighth)
    uint64_t n_demoted = 0;        // single queries answered by the bounded pass directly
    // The same for the FP6 first filter of batches ("f6_shadow"): its looser bound sends MORE queries of a batch to the ladder than
    // the int8 pass on topical rows (81-90 % against 53-68 % at 100 M rows: 133 against 80 ms per batch, profiles/r04/
    // f6_ab_100M_v7_topical*.log) while it wins 10 % where certificates hold.  Over windows of kF6FbWindow queries filtered that way:
    // above kF6FbSuspend of them ended in the ladder (h_stats, the mirrored counters: an over-estimate when single queries run in
    // between) the next suspend_len batches take the int8 pass (doubling up to kF6FbSuspendMax while the probes keep failing).
    struct F6Feedback {
        uint64_t issued = 0;      // queries filtered through the FP6 pass in this window
        uint32_t ladder0 = 0;     // h_stats[FLAG_BOUNDED] + h_stats[FLAG_FALLBACK] at its start
        uint32_t suspend_left = 0, suspend_len = 16;
    } f6fb;
    // ... and for the int8 matrix-core pass of batches: its sampled thresholds aim at ~1024 candidates per query ("mfma_target":
    // the fastest pass on well-spread rows, 8.82 ms per 100 M x 256 against 9.21 at 4096).  On topical rows a threshold that shallow
    // often sits ABOVE the k-th score inside a shell of near-ties and no certificate can hold: 53 % of a batch ended in the ladder
    // at 1024, 39 % at 4096 (79 -> 67 ms per batch at 100 M rows, profiles/r04/topical_target_sweep_100M.log).  Above
    // kBatchFbBoost of a window's queries in the ladder, the index TRIES four times as deep for a window, and keeps that depth only if it
    // saves the ladder a whole stream: with the wide form of the bounded pass (round 5: 64 flagged queries per stream of the int8 shadow)
    // a batch's ladder costs ceil(flagged / 64) streams — 135 -> 100 flagged is 3 -> 2 streams at 100 M rows (33 -> 28 ms per batch), but
    // 102 -> 74 at 12.5 M rows is 2 -> 2 and the deeper pass only costs (4.14 -> 5.19 ms: profiles/r05/operating_point_sweep.log).  An index
    // whose batches never reach the ladder aims half as deep instead (-3 % per batch on well-spread rows), and goes back at the first
    // sign of one.
    struct BatchFeedback {
        uint64_t issued = 0, passes = 0;  // batched queries / passes (of <= 256 queries) of this window
        uint32_t ladder0 = 0;             // h_stats[FLAG_BOUNDED] + h_stats[FLAG_FALLBACK] at its start
        int level = 1;                    // 0: half as deep (a quiet index), 1: "mfma_target", 2: four times as deep
        bool tried_deep = false, no_shallow = false;
        double base_per_pass = 0.0;       // queries per pass the ladder answered at level 1 (the window before a trial of level 2)
    } bfb;
    uint64_t n_f6_batches = 0, n_f6_suspended = 0, n_deepened_batches = 0;  // batches (of <= 256 queries) the FP6 filter took / handed to the int8 pass
    int synth_dist = 0;  // option "synth_dist": distribution of dawn_index_fill_synthetic rows (bench / tests)

    // bulk transfers (load / load_page_entries): one event per pinned host buffer of the caller's double buffer,
    // recorded behind the last copy out of it
    hipEvent_t ev_slot[2] = {nullptr, nullptr};
    bool ev_slot_used[2] = {false, false};

    // Single-row adds (dawn_index_add — the reference's insert path and its rebuild loop call index.add once per row,
    // search_provider.rs:149,284) are STAGED in pinned host memory: the row has passed the is_normalized gate on the host, so
    // nothing on the device can reject it; it joins the index with the next flush — when the stage is full, or in front of
    // whatever call looks at the rows next (search, save, get_rows, add_batch, ...): one transfer, one validation kernel, one
    // shadow update and one synchronisation per kAddStageRows rows instead of per row.  size() counts staged rows.
    float* h_add_rows = nullptr;     // [kAddStageRows][384], pinned (allocated at the first add)
    uint64_t* h_add_ids = nullptr;   // [kAddStageRows]
    size_t staged = 0;

    // one process, several devices (dawn_index_create_sharded): this handle owns no rows itself and routes every call
    dawn::ShardSet* shards = nullptr;
    bool pos_ids = false;  // this index is a shard of such a handle: d_ids hold global insertion positions
};

namespace dawn {

// Staged single-row adds (dawn_index::h_add_rows) join the index: called in front of everything that looks at the rows.
int index_flush_adds(dawn_index* idx);

// ---- single-device pieces used by the sharded router -------------------------------------------------------------
int index_create_single(int dtype, int device, dawn_index** out);
void index_destroy_single(dawn_index* idx);
int index_reserve_single(dawn_index* idx, size_t capacity);
// The whole search as a fixed launch sequence on `stream` (no allocation, no host decisions for B <= kMaxBatch).
int index_search_on_device(dawn_index* idx, const float* d_q, size_t B, size_t k, uint64_t* d_labels, float* d_dist,
                           uint32_t* d_found, hipStream_t stream);
// Bulk append in two phases.  async: enqueue, on idx->stream, the transfer of m rows into the slots behind size + pending
// (invisible to searches) and their is_normalized check; nothing is synchronised, the host buffers must stay valid
// until the stream has passed (bulk callers hand pinned buffers and wait on an event).  commit: wait, and either make
// all pending rows live (shadows brought up to date) or — a row failed the gate — drop them all.
enum class RowSrc { HostRows, HostPageEntries };
// h_src: m rows of 384 f32, or m PageEntry records of 1568 B.  h_ids: m labels; NULL = labels are first_label + i.
// slot 0/1: record the index's event of that slot behind the copies (index_append_wait(slot) then tells when the host
// buffer may be overwritten); -1: none.
int index_append_async(dawn_index* idx, RowSrc kind, const void* h_src, const uint64_t* h_ids, uint64_t first_label, size_t m,
                       int slot);
int index_append_wait(dawn_index* idx, int slot);
int index_append_check(dawn_index* idx, uint32_t* bad);  // wait for the pending rows; *bad = rows that failed the gate
int index_append_finish(dawn_index* idx, bool keep);     // make the pending rows live (shadows updated) or drop them
int index_append_commit(dawn_index* idx);   // check + finish: DAWN_OK / DAWN_ERR_NOT_NORMALIZED (nothing added) / error
void index_append_abort(dawn_index* idx);   // drop the pending rows (I/O error half way)
int index_clear(dawn_index* idx);           // size = 0 (load replaces the contents)
int index_fill_async(dawn_index* idx, uint64_t seed, uint64_t first_row, size_t n, uint64_t first_id, bool ids_are_positions,
                     uint64_t first_pos);
// Workspaces + filter shadows for the current rows and options, on idx->stream (callers synchronise).
int index_prepare_search(dawn_index* idx, bool appended = false);  // appended: rows were added, nothing else changed
int index_set_option_single(dawn_index* idx, const char* name, int64_t value);
int index_get_rows_single(dawn_index* idx, size_t first, size_t n, float* out_rows, uint64_t* out_ids);
int index_memory_single(dawn_index* idx, uint64_t* rows_bytes, uint64_t* shadow_bytes, uint64_t* other_bytes);
int index_stats_single(dawn_index* idx, uint64_t* searches, uint64_t* second, uint64_t* fallbacks, uint64_t* deepened,
                       uint64_t* bounded = nullptr, uint64_t* packed_failures = nullptr, uint64_t* demoted = nullptr);
int index_profile_read_single(dawn_index* idx, uint64_t* launches, double* total_ms);
int index_profile_enable_single(dawn_index* idx, int enable);

// ---- the router (dawn_sharded.cpp) -------------------------------------------------------------------------------
void sharded_destroy(dawn_index* idx);
int sharded_reserve(dawn_index* idx, size_t capacity);
size_t sharded_size(const dawn_index* idx);
size_t sharded_capacity(const dawn_index* idx);
int sharded_append_async(dawn_index* idx, RowSrc kind, const void* h_src, const uint64_t* h_ids, uint64_t first_label, size_t m,
                         int slot);
int sharded_append_wait(dawn_index* idx, int slot);
int sharded_append_commit(dawn_index* idx);
void sharded_append_abort(dawn_index* idx);
int sharded_clear(dawn_index* idx);
int sharded_search_device(dawn_index* idx, const float* d_q, size_t B, size_t k, uint64_t* d_labels, float* d_dist,
                          uint32_t* d_found, hipStream_t stream);
int sharded_search_batch(dawn_index* idx, const float* queries, size_t B, size_t count, uint64_t* labels, float* distances,
                         size_t* found);
int sharded_fill_synthetic(dawn_index* idx, uint64_t seed, uint64_t first_row, size_t n, uint64_t first_id);
int sharded_get_rows(dawn_index* idx, size_t first, size_t n, float* out_rows, uint64_t* out_ids);
int sharded_set_option(dawn_index* idx, const char* name, int64_t value);
int sharded_memory(dawn_index* idx, uint64_t* rows_bytes, uint64_t* shadow_bytes, uint64_t* other_bytes);
int sharded_stats(dawn_index* idx, uint64_t* searches, uint64_t* second, uint64_t* fallbacks, uint64_t* deepened,
                  uint64_t* bounded = nullptr, uint64_t* packed_failures = nullptr, uint64_t* demoted = nullptr);
void sharded_collect(dawn_index* idx, std::vector<dawn_index*>& out);  // the shards' single-device indexes
void sharded_batch_feedback(dawn_index* idx, uint64_t* f6_batches, uint64_t* f6_suspended, uint64_t* deepened_batches);
int sharded_profile_enable(dawn_index* idx, int enable);
int sharded_profile_read(dawn_index* idx, uint64_t* launches, double* total_ms);
int sharded_dtype(const dawn_index* idx);
int sharded_root_device(const dawn_index* idx);

}  // namespace dawn
