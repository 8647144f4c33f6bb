// dawn_index.cpp — the HBM-resident packed vector index behind the dawn_index_* C ABI.
//
// Replaces usearch::ffi::Index as used by src/search/search_provider.rs (new_index :102, reserve
// :133/:282, add :149/:284, search :214, size/capacity :246/:280, save/load :115-117/:178) with an
// exact brute-force index: rows [N][384] f32 + ids [N] u64 live in one HBM allocation each, appended in
// insertion order; search = scan_kernels.hip.
#include <algorithm>
#include <cstdlib>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "common.hpp"
#include "kernels.hpp"

using dawn::fail;

namespace {
constexpr size_t kMaxBatch = 256;        // queries per internal pass of the host API
constexpr size_t kMaxProfile = 4096;     // kept event pairs
constexpr size_t kZeroCopyBatch = 8;      // host API: up to this many queries get their results by zero-copy stores
constexpr size_t kShadowSmallRows = 6u << 20;  // below this the shadow stream uses geom_h_small
constexpr char kMagic[8] = {'D', 'A', 'W', 'N', 'I', 'D', 'X', '1'};
}  // namespace

struct dawn_index {
    int device = 0;
    size_t dims = DAWN_EM_LEN;
    hipStream_t stream = nullptr;

    int dtype = DAWN_DTYPE_F32;  // row storage: f32 (1536 B/row) or bf16 (768 B/row)
    char* d_x = nullptr;         // [(cap_phys + ROW_PAD)][384] of dtype
    // f32 index only: scaled-f16 shadow copy of the rows (f16(2^8 x), 768 B/row, tiles in MFMA-fragment order: ROW_F16S
    // in kernels.hpp) read by the matrix-core
    // FILTER instead of the f32 rows: half the bytes and no conversion work in the scan.  Built lazily at the first
    // batched search, extended on add; results stay exact (the rescore reads the f32 rows).  Costs +50 % HBM; if
    // the allocation fails the filter converts f32 rows on the fly as before.
    char* d_shadow = nullptr;
    size_t shadow_cap = 0;       // rows allocated
    size_t shadow_rows = 0;      // rows converted so far (prefix)
    int use_shadow = 1;          // option "f16_shadow"
    int shadow_small_batches = 1;  // option "f16_shadow_b1": batches of 1..8 queries also filter on the shadow
    // geometry of the shadow stream (MFMA from registers): one 2-wave block per CU, `unroll` picks the load schedule
    // (launch_filter_f16s_qb: 3 = ring of 12 fragments = 12 KiB in flight per wave).  tools/scan_sweep_shadow.py,
    // 80M rows: 7.02-7.07 TB/s; every schedule with 2-4 waves per CU lands within 1 % of it
    dawn::ScanGeom geom_h{256, 128, 3};
    // ... and below kShadowSmallRows rows (a few dozen sub-tiles per wave: start-up, tail and load balance count)
    // one 8-wave block per CU: 1M rows 154 -> 130 us.  Setting any shadow_scan_* option pins geom_h for every size.
    dawn::ScanGeom geom_h_small{256, 512, 3};
    bool geom_h_pinned = false;
    const dawn::ScanGeom& shadow_geom() const {
        return (!geom_h_pinned && size < kShadowSmallRows) ? geom_h_small : geom_h;
    }
    // int8 shadow stream: 4 waves per CU, whole sub-tiles (12 KiB) in flight per wave (tools/scan_sweep_shadow.py, 80M
    // rows: 7.01 TB/s against 6.97 with 2 waves; everything with >= 24 KiB in flight per CU lands within 2 %)
    dawn::ScanGeom geom_i8{256, 256, 3};
    // ... and 8 waves per CU below 16 M rows (12.5 M rows — one shard of 100 M on 8 GPUs —: 723 vs 731 us; 25 M: a tie)
    const dawn::ScanGeom& i8_geom() const {
        return geom_h_pinned ? geom_h : size < ((size_t)16 << 20) ? geom_h_small : geom_i8;
    }
    bool shadow_failed = false;  // allocation failed once: do not retry until the index is re-created
    // int8 shadow of the index rows (ROW_I8S, scan_i8.hip: 384 B/row + 8 B per 32 rows; f32 and bf16 indexes alike) read by
    // every filter — the streaming one of single queries and the matrix-core pass — instead of the rows: a quarter of the
    // f32 bytes.  Built lazily at the first search, extended on add, re-quantised on growth; if it cannot be allocated
    // (or "i8_shadow" = 0) the filters fall back to the f16 shadow / the rows.
    char* d_i8 = nullptr;
    float* d_i8meta = nullptr;
    size_t i8_cap = 0, i8_rows = 0;
    int use_i8 = 1;              // option "i8_shadow"
    int i8_batched = 1;          // option "i8_batched": batches of mfma_min_batch and more also filter on it
    bool i8_failed = false;
    float* d_stage = nullptr;    // bf16 index: f32 staging rows for add / get_rows / fill ([stage_rows][384])
    size_t stage_rows = 0;
    size_t row_bytes() const { return dtype == DAWN_DTYPE_BF16 ? dawn::EM * 2 : dawn::EM * 4; }
    uint64_t* d_ids = nullptr;  // [cap_phys]
    size_t size = 0;
    size_t cap_reported = 0;  // what reserve() promised (usearch semantics)
    size_t cap_phys = 0;      // rows actually allocated (geometric growth)

    // search workspaces
    // batch-1..8 streaming scan: one 4-wave block per CU, 3 row pairs (9 KiB) in flight per wave.  Measured on
    // MI355X (tools/scan_sweep.py, 40M rows): 36 KiB in flight per CU reads 7.17 TB/s; the full-occupancy
    // geometry (32 waves, 192 KiB per CU) only 6.55 TB/s.
    dawn::ScanGeom geom{256, 256, 3};
    size_t ws_B = 0;
    float* d_cand_s = nullptr;
    uint32_t* d_cand_p = nullptr;
    uint32_t* d_flags = nullptr;
    dawn::BatchWorkspace bws{nullptr, nullptr, nullptr, nullptr};  // matrix-core batched path
    int mfma_blocks = 256;   // one 8-wave workgroup per CU
    // B >= this goes to the matrix-core filter (sampled thresholds, one candidate buffer per query); below it the
    // streaming filter keeps per-wave top-64 lists, whose warm-up grows with every extra query
    // (tools/small_batch_paths.py, stream vs matrix-core ms — 1M rows: B=2 0.22 / 0.21, B=4 0.33 / 0.22, B=8 0.74 / 0.23;
    // 100M rows: B=2 11.00 / 11.15, B=4 11.20 / 11.19, B=8 11.96 / 11.18; B=1 0.176 / 0.199 and 10.93 / 11.15)
    // int8 shadow (tools/small_batch_paths.py, stream / matrix-core ms): 1M rows B=1 0.136 / 0.166, B=2 0.199 / 0.170,
    // B=3 0.251 / 0.169; 40M rows B=1 2.27 / 2.33, B=2 2.36 / 2.35, B=3 2.42 / 2.34: two queries and more take the pass
    int mfma_min_batch = 2;
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
    uint64_t n_searches = 0, n_fallbacks = 0, n_second = 0;
    int force_fallback = 0;
};

namespace {

int set_device(const dawn_index* idx) {
    DAWN_HIP_TRY(hipSetDevice(idx->device));
    return DAWN_OK;
}

size_t padded_rows(size_t rows) { return ((rows + dawn::ROW_PAD - 1) / dawn::ROW_PAD) * dawn::ROW_PAD + dawn::ROW_PAD; }

// Make room for at least `rows` rows (physical).  Existing rows are preserved.
int grow_phys(dawn_index* idx, size_t rows) {
    if (rows <= idx->cap_phys) return DAWN_OK;
    if (rows >= 0xFFFFFF00ull) return fail(DAWN_ERR_UNSUPPORTED, "index limited to 2^32-256 rows per device");
    char* nx = nullptr;
    uint64_t* nid = nullptr;
    const size_t prow = padded_rows(rows);
    const size_t rb = idx->row_bytes();
    DAWN_HIP_TRY(hipMalloc((void**)&nx, prow * rb));
    hipError_t e = hipMalloc((void**)&nid, std::max<size_t>(rows, 1) * sizeof(uint64_t));
    if (e != hipSuccess) {
        (void)hipFree(nx);
        return fail(DAWN_ERR_OOM, "hipMalloc(ids): %s", hipGetErrorString(e));
    }
    // (a bf16 index is stored in 64-row tiles: copy and clear whole tiles)
    const size_t live = idx->dtype == DAWN_DTYPE_BF16 ? (idx->size + dawn::ROW_PAD - 1) / dawn::ROW_PAD * dawn::ROW_PAD : idx->size;
    if (idx->size) {
        DAWN_HIP_TRY(hipMemcpyAsync(nx, idx->d_x, live * rb, hipMemcpyDeviceToDevice, idx->stream));
        DAWN_HIP_TRY(hipMemcpyAsync(nid, idx->d_ids, idx->size * sizeof(uint64_t), hipMemcpyDeviceToDevice,
                                    idx->stream));
    }
    // zero everything past the live rows: the scan may read (never use) up to ROW_PAD rows past size
    DAWN_HIP_TRY(hipMemsetAsync(nx + live * rb, 0, (prow - live) * rb, idx->stream));
    DAWN_HIP_TRY(hipStreamSynchronize(idx->stream));
    if (idx->d_x) (void)hipFree(idx->d_x);
    if (idx->d_ids) (void)hipFree(idx->d_ids);
    idx->d_x = nx;
    idx->d_ids = nid;
    idx->cap_phys = rows;
    return DAWN_OK;
}

int ensure_room(dawn_index* idx, size_t extra) {
    const size_t need = idx->size + extra;
    if (need > idx->cap_phys) {
        size_t target = std::max(need, idx->cap_phys + idx->cap_phys / 2);
        target = std::max<size_t>(target, 1024);
        DAWN_TRY(grow_phys(idx, target));
    }
    if (need > idx->cap_reported) idx->cap_reported = need;  // usearch would have required reserve(); we grow
    return DAWN_OK;
}

int ensure_workspace(dawn_index* idx, size_t B) {
    // the batched workspace also serves the f16-shadow streaming filter of small batches (scaled query images)
    const bool shadow_possible = idx->dtype == DAWN_DTYPE_F32 && idx->use_shadow && !idx->shadow_failed;
    if ((B >= (size_t)idx->mfma_min_batch || shadow_possible) && !idx->bws.cand) {
        if (int e = dawn::batched_init()) return fail(DAWN_ERR_HIP, "hipFuncSetAttribute(LDS): %s", hipGetErrorString((hipError_t)e));
        DAWN_HIP_TRY(hipMalloc((void**)&idx->bws.qh, (size_t)dawn::BATCH_QT * dawn::EM * sizeof(_Float16)));
        DAWN_HIP_TRY(hipMalloc((void**)&idx->bws.tau, dawn::BATCH_QT * sizeof(float)));
        DAWN_HIP_TRY(hipMalloc((void**)&idx->bws.cnt, dawn::BATCH_QT * dawn::BATCH_CAND_SEGS * sizeof(uint32_t)));
        DAWN_HIP_TRY(hipMalloc(&idx->bws.cand, (size_t)dawn::BATCH_QT * dawn::BATCH_CAP * 8));
    }
    if (B <= idx->ws_B) return DAWN_OK;
    if (idx->d_cand_s) (void)hipFree(idx->d_cand_s);
    if (idx->d_cand_p) (void)hipFree(idx->d_cand_p);
    if (idx->d_flags) (void)hipFree(idx->d_flags);
    idx->d_cand_s = nullptr;
    idx->d_cand_p = nullptr;
    idx->d_flags = nullptr;
    idx->ws_B = 0;
    const size_t n = B * (size_t)std::max({idx->geom.blocks, idx->geom_h.blocks, idx->geom_h_small.blocks, idx->geom_i8.blocks}) * dawn::LIST;
    DAWN_HIP_TRY(hipMalloc((void**)&idx->d_cand_s, n * sizeof(float)));
    DAWN_HIP_TRY(hipMalloc((void**)&idx->d_cand_p, n * sizeof(uint32_t)));
    DAWN_HIP_TRY(hipMalloc((void**)&idx->d_flags, 2 * B * sizeof(uint32_t)));  // flags[B] | arrival counters of the exact pass[B]
    DAWN_HIP_TRY(hipMemset(idx->d_flags, 0, 2 * B * sizeof(uint32_t)));
    DAWN_HIP_TRY(hipDeviceSynchronize());  // (searches run on non-blocking streams)
    idx->ws_B = B;
    return DAWN_OK;
}

// Bring the f16 shadow up to date with the f32 rows (no-op when disabled / not an f32 index / out of memory).
// Returns the filter's row source.
const void* filter_rows(dawn_index* idx, int* frt, hipStream_t stream) {
    *frt = idx->dtype;
    if (idx->dtype != DAWN_DTYPE_F32 || !idx->use_shadow || idx->shadow_failed) return idx->d_x;
    if (idx->shadow_cap < idx->cap_phys) {
        char* ns = nullptr;
        const size_t prow = padded_rows(idx->cap_phys);
        if (hipMalloc((void**)&ns, prow * dawn::EM * 2) != hipSuccess) {
            (void)hipGetLastError();
            idx->shadow_failed = true;
            return idx->d_x;
        }
        (void)hipMemsetAsync(ns, 0, prow * dawn::EM * 2, stream);
        if (idx->d_shadow) {  // keep what is converted already (the old buffer is idle: searches are serialised)
            (void)hipMemcpyAsync(ns, idx->d_shadow, padded_rows(idx->shadow_rows) * dawn::EM * 2, hipMemcpyDeviceToDevice,
                                 stream);  // whole tiles
            (void)hipStreamSynchronize(stream);
            (void)hipFree(idx->d_shadow);
        }
        idx->d_shadow = ns;
        idx->shadow_cap = idx->cap_phys;
    }
    if (idx->shadow_rows < idx->size) {
        dawn::launch_rows_f32_to_f16s(reinterpret_cast<const float*>(idx->d_x), idx->d_shadow, idx->shadow_rows, idx->size,
                                      stream);
        idx->shadow_rows = idx->size;
    }
    *frt = dawn::ROW_F16S;
    return idx->d_shadow;
}

// Bring the int8 shadow up to date; false when it is disabled or does not fit.
bool i8_rows_ready(dawn_index* idx, hipStream_t stream) {
    if (!idx->use_i8 || idx->i8_failed) return false;  // (a bf16 index: the int8 copy shadows its bf16 rows)
    if (idx->i8_cap < idx->cap_phys) {
        const size_t prow = padded_rows(idx->cap_phys) + 128;  // (the batched kernel moves 128-row tiles)
        const size_t bytes = prow * dawn::EM, mbytes = (prow / 32 + 1) * 8;
        // leave room for the f16 shadow of the matrix-core path (allocated at the first batch of mfma_min_batch queries)
        (void)hipStreamSynchronize(stream);  // the old buffers are idle: searches are serialised
        if (idx->d_i8) (void)hipFree(idx->d_i8);
        if (idx->d_i8meta) (void)hipFree(idx->d_i8meta);
        idx->d_i8 = nullptr;
        idx->d_i8meta = nullptr;
        idx->i8_cap = 0;
        size_t fr = 0, tot = 0;
        size_t need = bytes + mbytes + ((size_t)2 << 30);
        if (idx->dtype == DAWN_DTYPE_F32 && !idx->i8_batched && idx->use_shadow && !idx->shadow_failed &&
            idx->shadow_cap < idx->cap_phys)
            need += prow * dawn::EM * 2;
        char* ns = nullptr;
        float* nm = nullptr;
        if (hipMemGetInfo(&fr, &tot) != hipSuccess || fr < need || hipMalloc((void**)&ns, bytes) != hipSuccess ||
            hipMalloc((void**)&nm, mbytes) != hipSuccess) {
            (void)hipGetLastError();
            if (ns) (void)hipFree(ns);
            idx->i8_failed = true;
            return false;
        }
        (void)hipMemsetAsync(ns, 0, bytes, stream);  // sub-tiles past the last row: zeros, scale 0
        (void)hipMemsetAsync(nm, 0, mbytes, stream);
        idx->d_i8 = ns;
        idx->d_i8meta = nm;
        idx->i8_cap = idx->cap_phys;
        idx->i8_rows = 0;  // (re-quantised from the f32 rows: 0.03 ms per million rows)
    }
    if (idx->i8_rows < idx->size) {
        dawn::launch_rows_to_i8s(idx->d_x, idx->dtype, idx->d_i8, idx->d_i8meta, idx->i8_rows, idx->size, stream);
        idx->i8_rows = idx->size;
    }
    return true;
}

// The whole search as a fixed launch sequence on `stream` (no host decisions in between).
int search_on_device(dawn_index* idx, const float* d_q, size_t B, size_t k, uint64_t* d_labels, float* d_dist,
                     uint32_t* d_found, hipStream_t stream) {
    DAWN_TRY(ensure_workspace(idx, B));
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (idx->profiling && idx->events_used < kMaxProfile) {
        if (idx->events_used == idx->events.size()) {
            hipEvent_t a, b;
            DAWN_HIP_TRY(hipEventCreate(&a));
            DAWN_HIP_TRY(hipEventCreate(&b));
            idx->events.emplace_back(a, b);
        }
        e0 = idx->events[idx->events_used].first;
        e1 = idx->events[idx->events_used].second;
        idx->events_used++;
    }
    const uint32_t n = (uint32_t)idx->size;
    if ((int)B >= idx->mfma_min_batch && idx->i8_batched && i8_rows_ready(idx, stream)) {
        // matrix-core path on the int8 shadow (v_mfma_i32_32x32x32_i8, upper-bound scores), BATCH_QT queries per pass
        for (size_t b0 = 0; b0 < B; b0 += dawn::BATCH_QT) {
            const size_t nb = std::min<size_t>(dawn::BATCH_QT, B - b0);
            dawn::launch_scan_batched_i8(idx->d_x, idx->dtype, idx->d_i8, idx->d_i8meta, idx->d_ids, n, d_q + b0 * dawn::EM, (int)nb,
                                         (uint32_t)k, idx->bws, idx->mfma_blocks, d_labels + b0 * k, d_dist + b0 * k,
                                         d_found + b0, idx->d_flags + b0, idx->force_fallback, stream, b0 == 0 ? e0 : nullptr,
                                         b0 == 0 ? e1 : nullptr);
        }
    } else if ((int)B >= idx->mfma_min_batch) {
        // matrix-core path, BATCH_QT queries per pass over the index
        int frt = idx->dtype;
        const void* frows = idx->d_x;  // f32 index with mfma_sched 0 / 2: the lockstep kernel converts the f32 rows
        if (idx->dtype == DAWN_DTYPE_BF16 || (dawn::g_batched_sched != 0 && dawn::g_batched_sched != 2))
            frows = filter_rows(idx, &frt, stream);  // (a bf16 index is its own fragment-ordered filter source)
        for (size_t b0 = 0; b0 < B; b0 += dawn::BATCH_QT) {
            const size_t nb = std::min<size_t>(dawn::BATCH_QT, B - b0);
            dawn::launch_scan_batched(idx->d_x, idx->dtype, frows, frt, idx->d_ids, n, d_q + b0 * dawn::EM, (int)nb, (uint32_t)k, idx->bws,
                                      idx->mfma_blocks, d_labels + b0 * k, d_dist + b0 * k, d_found + b0,
                                      idx->d_flags + b0, idx->force_fallback, stream, b0 == 0 ? e0 : nullptr,
                                      b0 == 0 ? e1 : nullptr);
        }
    } else if (idx->shadow_small_batches && i8_rows_ready(idx, stream)) {
        // 1..8 queries on the int8 shadow (384 B/row): the filter scores are upper bounds of the exact ones
        const dawn::ScanGeom& gh = idx->i8_geom();
        dawn::launch_scan_filter_i8s(idx->d_i8, idx->d_i8meta, n, d_q, (int)B, idx->d_cand_s, idx->d_cand_p, gh, stream, e0, e1);
        dawn::launch_merge_rescore(idx->d_x, idx->dtype, idx->d_ids, n, d_q, (int)B, idx->d_cand_s, idx->d_cand_p, gh.blocks,
                                   (uint32_t)k, d_labels, d_dist, d_found, idx->d_flags, idx->force_fallback,
                                   dawn::FILTER_EPS_I8, stream);
    } else {
        int frt = idx->dtype;
        const void* frows = filter_rows(idx, &frt, stream);
        if (frt == dawn::ROW_BF16 || (frt == dawn::ROW_F16S && idx->shadow_small_batches)) {
            // 1..8 queries: stream the 16-bit fragments (768 B/row: the f16 shadow instead of the f32 rows, or the
            // bf16 index itself) through the matrix cores
            const dawn::ScanGeom& gh = idx->shadow_geom();
            dawn::launch_scan_filter_f16s(frows, frt, n, d_q, (int)B, idx->d_cand_s, idx->d_cand_p, gh, stream, e0, e1);
            dawn::launch_merge_rescore(idx->d_x, idx->dtype, idx->d_ids, n, d_q, (int)B, idx->d_cand_s, idx->d_cand_p,
                                       gh.blocks, (uint32_t)k, d_labels, d_dist, d_found, idx->d_flags, idx->force_fallback,
                                       frt == dawn::ROW_BF16 ? dawn::FILTER_EPS_BF16_STREAM : dawn::FILTER_EPS_F16, stream);
        } else {
            dawn::launch_scan_filter(idx->d_x, idx->dtype, n, d_q, (int)B, idx->d_cand_s, idx->d_cand_p, idx->geom, stream,
                                     e0, e1);
            dawn::launch_merge_rescore(idx->d_x, idx->dtype, idx->d_ids, n, d_q, (int)B, idx->d_cand_s, idx->d_cand_p,
                                       idx->geom.blocks, (uint32_t)k, d_labels, d_dist, d_found, idx->d_flags,
                                       idx->force_fallback, dawn::FILTER_EPS_F32, stream);
        }
    }
    dawn::launch_scan_exact(idx->d_x, idx->dtype, idx->d_ids, n, d_q, (int)B, idx->d_flags, idx->d_flags + idx->ws_B, idx->d_cand_s,
                            idx->d_cand_p, idx->geom.blocks, (uint32_t)k, d_labels, d_dist, d_found, stream);
    DAWN_HIP_TRY(hipGetLastError());
    return DAWN_OK;
}

}  // namespace

extern "C" {

int dawn_index_create(size_t dims, int dtype, int device, dawn_index** out) {
    if (!out) return fail(DAWN_ERR_INVALID_ARG, "out is NULL");
    *out = nullptr;
    if (dims != DAWN_EM_LEN) return fail(DAWN_ERR_UNSUPPORTED, "dims must be %d (EM_LEN)", DAWN_EM_LEN);
    if (dtype != DAWN_DTYPE_F32 && dtype != DAWN_DTYPE_BF16)
        return fail(DAWN_ERR_UNSUPPORTED, "dtype %d not supported", dtype);
    DAWN_TRY(dawn::require_device(device));
    DAWN_HIP_TRY(hipSetDevice(device));
    auto* idx = new dawn_index();
    idx->device = device;
    idx->dtype = dtype;
    if (const char* e = getenv("DAWN_I8_SHADOW")) idx->use_i8 = atoi(e) != 0;  // default of the "i8_shadow" option
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0)
        idx->geom.blocks = prop.multiProcessorCount;  // one block per CU
    if (prop.multiProcessorCount > 0) {
        idx->geom_h.blocks = prop.multiProcessorCount;
        idx->geom_h_small.blocks = prop.multiProcessorCount;
        idx->geom_i8.blocks = prop.multiProcessorCount;
    }
    if (prop.multiProcessorCount > 0) idx->mfma_blocks = prop.multiProcessorCount;
    hipError_t e = hipStreamCreateWithFlags(&idx->stream, hipStreamNonBlocking);
    if (e != hipSuccess) {
        delete idx;
        return fail(DAWN_ERR_HIP, "hipStreamCreate: %s", hipGetErrorString(e));
    }
    const size_t hb = kMaxBatch * (dawn::EM * 4 + DAWN_MAX_K * 8 + DAWN_MAX_K * 4 + 8);
    if (hipHostMalloc(&idx->h_pinned, hb, hipHostMallocDefault) != hipSuccess ||
        hipMalloc((void**)&idx->d_q, kMaxBatch * dawn::EM * sizeof(float)) != hipSuccess ||
        hipMalloc((void**)&idx->d_labels, kMaxBatch * DAWN_MAX_K * sizeof(uint64_t)) != hipSuccess ||
        hipMalloc((void**)&idx->d_dist, kMaxBatch * DAWN_MAX_K * sizeof(float)) != hipSuccess ||
        hipMalloc((void**)&idx->d_found, kMaxBatch * sizeof(uint32_t)) != hipSuccess ||
        hipMalloc((void**)&idx->d_bad, sizeof(uint32_t)) != hipSuccess) {
        dawn_index_destroy(idx);
        return fail(DAWN_ERR_OOM, "allocating index staging buffers failed");
    }
    idx->h_pinned_bytes = hb;
    *out = idx;
    return DAWN_OK;
}

void dawn_index_destroy(dawn_index* idx) {
    if (!idx) return;
    (void)hipSetDevice(idx->device);
    if (idx->stream) (void)hipStreamSynchronize(idx->stream);
    for (auto& ev : idx->events) {
        (void)hipEventDestroy(ev.first);
        (void)hipEventDestroy(ev.second);
    }
    void* ptrs[] = {idx->d_x, idx->d_shadow, idx->d_i8, idx->d_i8meta, idx->d_stage, idx->d_ids, idx->d_cand_s, idx->d_cand_p, idx->d_flags, idx->bws.qh, idx->bws.tau,
                    idx->bws.cnt, idx->bws.cand, idx->d_q, idx->d_labels, idx->d_dist, idx->d_found, idx->d_bad};
    for (void* p : ptrs)
        if (p) (void)hipFree(p);
    if (idx->h_pinned) (void)hipHostFree(idx->h_pinned);
    if (idx->stream) (void)hipStreamDestroy(idx->stream);
    delete idx;
}

int dawn_index_reserve(dawn_index* idx, size_t capacity) {
    if (!idx) return fail(DAWN_ERR_INVALID_ARG, "idx is NULL");
    DAWN_TRY(set_device(idx));
    if (capacity > idx->cap_phys) DAWN_TRY(grow_phys(idx, capacity));
    if (capacity > idx->cap_reported) idx->cap_reported = capacity;
    return DAWN_OK;
}

size_t dawn_index_size(const dawn_index* idx) { return idx ? idx->size : 0; }
size_t dawn_index_capacity(const dawn_index* idx) { return idx ? idx->cap_reported : 0; }

// bf16 index: f32 rows pass through a device staging buffer (validated there, then rounded into the index)
static int ensure_stage(dawn_index* idx, size_t rows) {
    if (rows <= idx->stage_rows) return DAWN_OK;
    if (idx->d_stage) (void)hipFree(idx->d_stage);
    idx->d_stage = nullptr;
    idx->stage_rows = 0;
    DAWN_HIP_TRY(hipMalloc((void**)&idx->d_stage, rows * dawn::EM * sizeof(float)));
    idx->stage_rows = rows;
    return DAWN_OK;
}
constexpr size_t kStageChunk = 1u << 18;  // 256 Ki rows = 384 MiB of f32 staging at most

int dawn_index_add_batch(dawn_index* idx, size_t n, const uint64_t* ids, const float* v) {
    if (!idx || (!ids && n) || (!v && n)) return fail(DAWN_ERR_INVALID_ARG, "NULL argument");
    if (n == 0) return DAWN_OK;
    DAWN_TRY(set_device(idx));
    DAWN_TRY(ensure_room(idx, n));
    const bool bf16 = idx->dtype == DAWN_DTYPE_BF16;
    const size_t rb = idx->row_bytes();
    // rows land past `size` (invisible to searches) and become live only after validation
    DAWN_HIP_TRY(hipMemsetAsync(idx->d_bad, 0, sizeof(uint32_t), idx->stream));
    if (!bf16) {
        float* dst = reinterpret_cast<float*>(idx->d_x + idx->size * rb);
        DAWN_HIP_TRY(hipMemcpyAsync(dst, v, n * rb, hipMemcpyHostToDevice, idx->stream));
        dawn::launch_validate_rows(dst, (uint32_t)n, idx->d_bad, idx->stream);
    } else {
        DAWN_TRY(ensure_stage(idx, std::min(n, kStageChunk)));
        for (size_t o = 0; o < n; o += kStageChunk) {
            const size_t m = std::min(kStageChunk, n - o);
            DAWN_HIP_TRY(hipMemcpyAsync(idx->d_stage, v + o * dawn::EM, m * dawn::EM * sizeof(float),
                                        hipMemcpyHostToDevice, idx->stream));
            dawn::launch_validate_rows(idx->d_stage, (uint32_t)m, idx->d_bad, idx->stream);  // gate on the f32 input
            dawn::launch_rows_f32_to_bf16(idx->d_stage, idx->d_x, idx->size + o, m, idx->stream);
            if (o + kStageChunk < n) DAWN_HIP_TRY(hipStreamSynchronize(idx->stream));  // staging buffer reuse
        }
    }
    DAWN_HIP_TRY(hipMemcpyAsync(idx->d_ids + idx->size, ids, n * sizeof(uint64_t), hipMemcpyHostToDevice,
                                idx->stream));
    uint32_t bad = 0;
    DAWN_HIP_TRY(hipMemcpyAsync(&bad, idx->d_bad, sizeof(uint32_t), hipMemcpyDeviceToHost, idx->stream));
    DAWN_HIP_TRY(hipStreamSynchronize(idx->stream));
    if (bad) {
        // (a bf16 index keeps the rejected rows' fragments: rows >= size are masked by position in every kernel)
        if (!bf16) (void)hipMemsetAsync(idx->d_x + idx->size * rb, 0, n * rb, idx->stream);
        (void)hipStreamSynchronize(idx->stream);
        return fail(DAWN_ERR_NOT_NORMALIZED, "Insert embedding is not normalized (%u of %zu rows)", bad, n);
    }
    idx->size += n;
    return DAWN_OK;
}

int dawn_index_add(dawn_index* idx, uint64_t id, const float* v) {
    if (!idx || !v) return fail(DAWN_ERR_INVALID_ARG, "NULL argument");
    if (!dawn::host_is_normalized(v)) return fail(DAWN_ERR_NOT_NORMALIZED, "Insert embedding is not normalized");
    DAWN_TRY(set_device(idx));
    DAWN_TRY(ensure_room(idx, 1));
    float* hp = (float*)idx->h_pinned;
    std::memcpy(hp, v, dawn::EM * sizeof(float));
    uint64_t* hid = (uint64_t*)(hp + dawn::EM);
    *hid = id;
    const size_t rb = idx->row_bytes();
    if (idx->dtype == DAWN_DTYPE_BF16) {
        DAWN_TRY(ensure_stage(idx, 1));
        DAWN_HIP_TRY(hipMemcpyAsync(idx->d_stage, hp, dawn::EM * sizeof(float), hipMemcpyHostToDevice, idx->stream));
        dawn::launch_rows_f32_to_bf16(idx->d_stage, idx->d_x, idx->size, 1, idx->stream);
    } else {
        DAWN_HIP_TRY(hipMemcpyAsync(idx->d_x + idx->size * rb, hp, rb, hipMemcpyHostToDevice, idx->stream));
    }
    DAWN_HIP_TRY(hipMemcpyAsync(idx->d_ids + idx->size, hid, sizeof(uint64_t), hipMemcpyHostToDevice, idx->stream));
    DAWN_HIP_TRY(hipStreamSynchronize(idx->stream));
    idx->size += 1;
    return DAWN_OK;
}

int dawn_index_search_device(dawn_index* idx, const float* d_queries, size_t B, size_t count, uint64_t* d_labels,
                             float* d_distances, uint32_t* d_found, void* stream) {
    if (!idx || !d_queries || !d_labels || !d_distances || !d_found) return fail(DAWN_ERR_INVALID_ARG, "NULL argument");
    if (count == 0 || count > DAWN_MAX_K) return fail(DAWN_ERR_UNSUPPORTED, "count must be 1..%d", DAWN_MAX_K);
    if (B == 0) return DAWN_OK;
    DAWN_TRY(set_device(idx));
    idx->n_searches += B;
    return search_on_device(idx, d_queries, B, count, d_labels, d_distances, d_found, (hipStream_t)stream);
}

int dawn_index_search_batch(dawn_index* idx, const float* queries, size_t B, size_t count, uint64_t* labels,
                            float* distances, size_t* found) {
    if (!idx || !queries || !labels || !distances || !found) return fail(DAWN_ERR_INVALID_ARG, "NULL argument");
    if (count == 0 || count > DAWN_MAX_K) return fail(DAWN_ERR_UNSUPPORTED, "count must be 1..%d", DAWN_MAX_K);
    for (size_t b = 0; b < B; ++b)  // search_provider.rs:206-208
        if (!dawn::host_is_normalized(queries + b * dawn::EM))
            return fail(DAWN_ERR_NOT_NORMALIZED, "Search vector is not normalized");
    DAWN_TRY(set_device(idx));
    for (size_t b0 = 0; b0 < B; b0 += kMaxBatch) {
        const size_t nb = std::min(kMaxBatch, B - b0);
        char* hp = (char*)idx->h_pinned;
        float* hq = (float*)hp;
        uint64_t* hl = (uint64_t*)(hp + kMaxBatch * dawn::EM * 4);
        float* hd = (float*)(hp + kMaxBatch * (dawn::EM * 4 + DAWN_MAX_K * 8));
        uint32_t* hf = (uint32_t*)(hp + kMaxBatch * (dawn::EM * 4 + DAWN_MAX_K * 8 + DAWN_MAX_K * 4));
        uint32_t* hflag = hf + kMaxBatch;
        std::memcpy(hq, queries + b0 * dawn::EM, nb * dawn::EM * sizeof(float));
        DAWN_HIP_TRY(hipMemcpyAsync(idx->d_q, hq, nb * dawn::EM * sizeof(float), hipMemcpyHostToDevice, idx->stream));
        idx->n_searches += nb;
        if (nb <= kZeroCopyBatch) {
            // few queries: the tail kernels store the results straight into the pinned host block (coherent,
            // device-visible): no copy commands between the last kernel and the host's wake-up
            DAWN_TRY(search_on_device(idx, idx->d_q, nb, count, hl, hd, hf, idx->stream));
        } else {
            DAWN_TRY(search_on_device(idx, idx->d_q, nb, count, idx->d_labels, idx->d_dist, idx->d_found, idx->stream));
            DAWN_HIP_TRY(hipMemcpyAsync(hl, idx->d_labels, nb * count * sizeof(uint64_t), hipMemcpyDeviceToHost, idx->stream));
            DAWN_HIP_TRY(hipMemcpyAsync(hd, idx->d_dist, nb * count * sizeof(float), hipMemcpyDeviceToHost, idx->stream));
            DAWN_HIP_TRY(hipMemcpyAsync(hf, idx->d_found, nb * sizeof(uint32_t), hipMemcpyDeviceToHost, idx->stream));
        }
        DAWN_HIP_TRY(hipMemcpyAsync(hflag, idx->d_flags, nb * sizeof(uint32_t), hipMemcpyDeviceToHost, idx->stream));
        DAWN_HIP_TRY(hipStreamSynchronize(idx->stream));
        std::memcpy(labels + b0 * count, hl, nb * count * sizeof(uint64_t));
        std::memcpy(distances + b0 * count, hd, nb * count * sizeof(float));
        for (size_t b = 0; b < nb; ++b) {
            found[b0 + b] = hf[b];
            if (hflag[b] == dawn::FLAG_FALLBACK) idx->n_fallbacks++;
            else if (hflag[b] == dawn::FLAG_SECOND) idx->n_second++;
        }
    }
    return DAWN_OK;
}

int dawn_index_search(dawn_index* idx, const float* query, size_t count, uint64_t* labels, float* distances,
                      size_t* found) {
    return dawn_index_search_batch(idx, query, 1, count, labels, distances, found);
}

int dawn_index_search_limited(dawn_index* idx, const float* query, size_t count, float distance_limit, uint64_t* labels,
                              float* distances, size_t* found) {
    DAWN_TRY(dawn_index_search_batch(idx, query, 1, count, labels, distances, found));
    size_t keep = 0;  // hits are ascending: the reported ones are a prefix
    while (keep < *found && !(distances[keep] >= distance_limit)) ++keep;
    *found = keep;
    return DAWN_OK;
}

int dawn_topk_merge_device(int device, size_t G, size_t B, size_t count, const uint64_t* d_in_labels,
                           const float* d_in_distances, const uint32_t* d_in_found, uint64_t* d_labels,
                           float* d_distances, uint32_t* d_found, void* stream) {
    if (!d_in_labels || !d_in_distances || !d_in_found || !d_labels || !d_distances || !d_found)
        return fail(DAWN_ERR_INVALID_ARG, "NULL argument");
    if (G == 0 || count == 0 || G * count > 512) return fail(DAWN_ERR_UNSUPPORTED, "G*count must be 1..512");
    DAWN_TRY(dawn::require_device(device));
    DAWN_HIP_TRY(hipSetDevice(device));
    if (B == 0) return DAWN_OK;
    dawn::launch_shard_merge(G, B, count, d_in_labels, d_in_distances, d_in_found, B * count, B * count, B, d_labels,
                             d_distances, d_found, (hipStream_t)stream);
    DAWN_HIP_TRY(hipGetLastError());
    return DAWN_OK;
}

size_t dawn_result_blob_bytes(size_t B, size_t count) { return (B * count * 12 + B * 4 + 15) / 16 * 16; }

int dawn_topk_merge_packed_device(int device, size_t G, size_t B, size_t count, const void* d_blobs,
                                  uint64_t* d_labels, float* d_distances, uint32_t* d_found, void* stream) {
    if (!d_blobs || !d_labels || !d_distances || !d_found) return fail(DAWN_ERR_INVALID_ARG, "NULL argument");
    if (G == 0 || count == 0 || G * count > 512) return fail(DAWN_ERR_UNSUPPORTED, "G*count must be 1..512");
    DAWN_TRY(dawn::require_device(device));
    DAWN_HIP_TRY(hipSetDevice(device));
    if (B == 0) return DAWN_OK;
    const size_t stride = dawn_result_blob_bytes(B, count);
    const char* base = (const char*)d_blobs;
    dawn::launch_shard_merge(G, B, count, (const uint64_t*)base, (const float*)(base + B * count * 8),
                             (const uint32_t*)(base + B * count * 12), stride / 8, stride / 4, stride / 4, d_labels,
                             d_distances, d_found, (hipStream_t)stream);
    DAWN_HIP_TRY(hipGetLastError());
    return DAWN_OK;
}

int dawn_index_fill_synthetic(dawn_index* idx, uint64_t seed, uint64_t first_row, size_t n, uint64_t first_id) {
    if (!idx) return fail(DAWN_ERR_INVALID_ARG, "idx is NULL");
    if (n == 0) return DAWN_OK;
    DAWN_TRY(set_device(idx));
    DAWN_TRY(ensure_room(idx, n));
    const bool bf16 = idx->dtype == DAWN_DTYPE_BF16;
    const size_t rb = idx->row_bytes();
    const size_t chunk = bf16 ? kStageChunk : (size_t)1u << 22;  // rows per generator launch
    if (bf16) DAWN_TRY(ensure_stage(idx, std::min(n, chunk)));
    float* d_len = nullptr;
    DAWN_HIP_TRY(hipMalloc((void**)&d_len, std::min(n, chunk) * sizeof(float)));
    for (size_t o = 0; o < n; o += chunk) {
        const size_t m = std::min(chunk, n - o);
        char* dst = idx->d_x + (idx->size + o) * rb;
        if (bf16) {  // f32 unit rows of the spec, then rounded: the bf16 index holds round_bf16(spec row)
            dawn::launch_fill_synth(seed, first_row + o, (uint32_t)m, idx->d_stage, d_len, idx->stream);
            dawn::launch_rows_f32_to_bf16(idx->d_stage, idx->d_x, idx->size + o, m, idx->stream);
        } else {
            dawn::launch_fill_synth(seed, first_row + o, (uint32_t)m, reinterpret_cast<float*>(dst), d_len, idx->stream);
        }
        dawn::launch_iota_u64(idx->d_ids + idx->size + o, first_id + o, (uint32_t)m, idx->stream);
    }
    hipError_t e = hipStreamSynchronize(idx->stream);
    (void)hipFree(d_len);
    if (e != hipSuccess) return fail(DAWN_ERR_HIP, "fill_synthetic: %s", hipGetErrorString(e));
    idx->size += n;
    return DAWN_OK;
}

// Rows come back as f32 whatever the storage type (bf16 rows widened exactly).
int dawn_index_get_rows(dawn_index* idx, size_t first, size_t n, float* out_rows, uint64_t* out_ids) {
    if (!idx) return fail(DAWN_ERR_INVALID_ARG, "idx is NULL");
    if (first + n > idx->size) return fail(DAWN_ERR_INVALID_ARG, "rows [%zu, %zu) out of range (size %zu)", first, first + n, idx->size);
    if (n == 0) return DAWN_OK;
    DAWN_TRY(set_device(idx));
    const size_t rb = idx->row_bytes();
    if (out_rows && idx->dtype == DAWN_DTYPE_BF16) {
        DAWN_TRY(ensure_stage(idx, std::min(n, kStageChunk)));
        for (size_t o = 0; o < n; o += kStageChunk) {
            const size_t m = std::min(kStageChunk, n - o);
            dawn::launch_rows_bf16_to_f32(idx->d_x, first + o, idx->d_stage, m, idx->stream);
            DAWN_HIP_TRY(hipMemcpyAsync(out_rows + o * dawn::EM, idx->d_stage, m * dawn::EM * sizeof(float),
                                        hipMemcpyDeviceToHost, idx->stream));
            DAWN_HIP_TRY(hipStreamSynchronize(idx->stream));
        }
    } else if (out_rows) {
        DAWN_HIP_TRY(hipMemcpy(out_rows, idx->d_x + first * rb, n * rb, hipMemcpyDeviceToHost));
    }
    if (out_ids) DAWN_HIP_TRY(hipMemcpy(out_ids, idx->d_ids + first, n * sizeof(uint64_t), hipMemcpyDeviceToHost));
    return DAWN_OK;
}

// File layout: "DAWNIDX1" | u32 dims | u32 dtype | u64 n | ids[n] u64 | rows[n][384] f32 (little endian).  Rows are
// written as f32 for both storage types (a bf16 index widens exactly and re-rounds to the same bits on load).
int dawn_index_save(dawn_index* idx, const char* path) {
    if (!idx || !path) return fail(DAWN_ERR_INVALID_ARG, "NULL argument");
    DAWN_TRY(set_device(idx));
    FILE* f = std::fopen(path, "wb");
    if (!f) return fail(DAWN_ERR_IO, "cannot open %s for writing", path);
    const uint32_t dims = DAWN_EM_LEN, dtype = (uint32_t)idx->dtype;
    const uint64_t n = idx->size;
    bool ok = std::fwrite(kMagic, 1, 8, f) == 8 && std::fwrite(&dims, 4, 1, f) == 1 &&
              std::fwrite(&dtype, 4, 1, f) == 1 && std::fwrite(&n, 8, 1, f) == 1;
    const size_t chunk = 1u << 16;
    std::vector<char> buf(chunk * dawn::EM * sizeof(float));
    for (size_t o = 0; ok && o < n; o += chunk) {
        const size_t m = std::min<size_t>(chunk, n - o);
        if (hipMemcpy(buf.data(), idx->d_ids + o, m * 8, hipMemcpyDeviceToHost) != hipSuccess) ok = false;
        else ok = std::fwrite(buf.data(), 8, m, f) == m;
    }
    for (size_t o = 0; ok && o < n; o += chunk) {
        const size_t m = std::min<size_t>(chunk, n - o);
        if (dawn_index_get_rows(idx, o, m, reinterpret_cast<float*>(buf.data()), nullptr) != DAWN_OK) ok = false;
        else ok = std::fwrite(buf.data(), dawn::EM * 4, m, f) == m;
    }
    if (std::fclose(f) != 0) ok = false;
    if (!ok) return fail(DAWN_ERR_IO, "writing %s failed", path);
    return DAWN_OK;
}

int dawn_index_load(dawn_index* idx, const char* path) {
    if (!idx || !path) return fail(DAWN_ERR_INVALID_ARG, "NULL argument");
    DAWN_TRY(set_device(idx));
    FILE* f = std::fopen(path, "rb");
    if (!f) return fail(DAWN_ERR_IO, "cannot open %s", path);
    char magic[8];
    uint32_t dims = 0, dtype = 0;
    uint64_t n = 0;
    if (std::fread(magic, 1, 8, f) != 8 || std::memcmp(magic, kMagic, 8) != 0 || std::fread(&dims, 4, 1, f) != 1 ||
        std::fread(&dtype, 4, 1, f) != 1 || std::fread(&n, 8, 1, f) != 1 || dims != DAWN_EM_LEN ||
        (dtype != DAWN_DTYPE_F32 && dtype != DAWN_DTYPE_BF16)) {
        std::fclose(f);
        return fail(DAWN_ERR_IO, "%s is not a dawn index file", path);
    }
    std::vector<uint64_t> ids(n);
    if (n && std::fread(ids.data(), 8, n, f) != n) {
        std::fclose(f);
        return fail(DAWN_ERR_IO, "%s: truncated id table", path);
    }
    idx->size = 0;  // load replaces the contents (usearch load semantics)
    idx->shadow_rows = 0;
    idx->i8_rows = 0;
    const size_t chunk = 1u << 16;
    std::vector<float> buf(chunk * dawn::EM);
    for (size_t o = 0; o < n; o += chunk) {
        const size_t m = std::min<size_t>(chunk, n - o);
        if (std::fread(buf.data(), dawn::EM * 4, m, f) != m) {
            std::fclose(f);
            return fail(DAWN_ERR_IO, "%s: truncated row data", path);
        }
        int rc = dawn_index_add_batch(idx, m, ids.data() + o, buf.data());
        if (rc != DAWN_OK) {
            std::fclose(f);
            return rc;
        }
    }
    std::fclose(f);
    return DAWN_OK;
}

// src/index/warc.rs:35-43 PageEntry (repr(C)): u64 url_pos, u64 title_pos, f32 vector[384], u64 url_len,
// u64 title_len = 1568 bytes; read as examples_old/document_embeddings.rs:60-71 does.
int dawn_index_load_page_entries(dawn_index* idx, const char* emb_path, uint64_t first_id) {
    if (!idx || !emb_path) return fail(DAWN_ERR_INVALID_ARG, "NULL argument");
    FILE* f = std::fopen(emb_path, "rb");
    if (!f) return fail(DAWN_ERR_IO, "cannot open %s", emb_path);
    constexpr size_t REC = 1568, OFF = 16;
    const size_t chunk = 1u << 14;
    std::vector<unsigned char> raw(chunk * REC);
    std::vector<float> rows(chunk * dawn::EM);
    std::vector<uint64_t> ids(chunk);
    uint64_t next = first_id;
    for (;;) {
        const size_t m = std::fread(raw.data(), REC, chunk, f);  // entries() = len / size_of::<PageEntry>()
        if (m == 0) break;
        for (size_t i = 0; i < m; ++i) {
            std::memcpy(rows.data() + i * dawn::EM, raw.data() + i * REC + OFF, dawn::EM * 4);
            ids[i] = next++;
        }
        int rc = dawn_index_add_batch(idx, m, ids.data(), rows.data());
        if (rc != DAWN_OK) {
            std::fclose(f);
            return rc;
        }
    }
    std::fclose(f);
    return DAWN_OK;
}

int dawn_index_profile_enable(dawn_index* idx, int enable) {
    if (!idx) return fail(DAWN_ERR_INVALID_ARG, "idx is NULL");
    idx->profiling = enable != 0;
    idx->events_used = 0;
    return DAWN_OK;
}

int dawn_index_profile_read(dawn_index* idx, uint64_t* launches, double* total_ms) {
    if (!idx || !launches || !total_ms) return fail(DAWN_ERR_INVALID_ARG, "NULL argument");
    DAWN_TRY(set_device(idx));
    DAWN_HIP_TRY(hipDeviceSynchronize());
    double sum = 0.0;
    for (size_t i = 0; i < idx->events_used; ++i) {
        float ms = 0.f;
        DAWN_HIP_TRY(hipEventElapsedTime(&ms, idx->events[i].first, idx->events[i].second));
        sum += ms;
    }
    *launches = idx->events_used;
    *total_ms = sum;
    idx->events_used = 0;
    return DAWN_OK;
}

int dawn_index_stats(dawn_index* idx, uint64_t* searches, uint64_t* fallbacks) {
    if (!idx) return fail(DAWN_ERR_INVALID_ARG, "idx is NULL");
    if (searches) *searches = idx->n_searches;
    if (fallbacks) *fallbacks = idx->n_fallbacks;
    return DAWN_OK;
}

// ... plus the queries whose first certificate failed and whose 1024-deep second one held (no exact pass needed)
int dawn_index_stats_ext(dawn_index* idx, uint64_t* searches, uint64_t* second_chances, uint64_t* fallbacks) {
    if (!idx) return fail(DAWN_ERR_INVALID_ARG, "idx is NULL");
    if (searches) *searches = idx->n_searches;
    if (second_chances) *second_chances = idx->n_second;
    if (fallbacks) *fallbacks = idx->n_fallbacks;
    return DAWN_OK;
}

int dawn_index_memory(dawn_index* idx, uint64_t* rows_bytes, uint64_t* shadow_bytes, uint64_t* other_bytes) {
    if (!idx) return fail(DAWN_ERR_INVALID_ARG, "idx is NULL");
    const uint64_t rows = idx->d_x ? (uint64_t)padded_rows(idx->cap_phys) * idx->row_bytes() : 0;
    uint64_t shadows = 0;
    if (idx->d_shadow) shadows += (uint64_t)padded_rows(idx->shadow_cap) * dawn::EM * 2;
    if (idx->d_i8) {
        const uint64_t prow = padded_rows(idx->i8_cap) + 128;
        shadows += prow * dawn::EM + (prow / 32 + 1) * 8;
    }
    uint64_t other = (uint64_t)std::max<size_t>(idx->cap_phys, idx->d_ids ? 1 : 0) * sizeof(uint64_t);  // ids
    if (idx->d_cand_s)
        other += (uint64_t)idx->ws_B * std::max({idx->geom.blocks, idx->geom_h.blocks, idx->geom_h_small.blocks, idx->geom_i8.blocks}) *
                     dawn::LIST * 8 + 2 * idx->ws_B * 4;
    if (idx->bws.cand)
        other += (uint64_t)dawn::BATCH_QT * (dawn::EM * 2 + 4 + dawn::BATCH_CAND_SEGS * 4 + (uint64_t)dawn::BATCH_CAP * 8);
    if (idx->d_stage) other += (uint64_t)idx->stage_rows * dawn::EM * 4;
    other += kMaxBatch * (dawn::EM * 4 + DAWN_MAX_K * 12 + 4) + 4;  // host-API staging
    if (rows_bytes) *rows_bytes = rows;
    if (shadow_bytes) *shadow_bytes = shadows;
    if (other_bytes) *other_bytes = other;
    return DAWN_OK;
}

int dawn_index_debug_filter_scores(dawn_index* idx, const float* queries, size_t B, float* out, size_t* n_out) {
    if (!idx || !queries || !out || !n_out) return fail(DAWN_ERR_INVALID_ARG, "NULL argument");
    if (B == 0 || B > (size_t)dawn::BATCH_QT) return fail(DAWN_ERR_INVALID_ARG, "B must be 1..%d", dawn::BATCH_QT);
    DAWN_TRY(set_device(idx));
    DAWN_TRY(ensure_workspace(idx, std::max<size_t>(B, idx->mfma_min_batch)));
    const size_t n = std::min<size_t>(idx->size, dawn::BATCH_CAP);
    *n_out = n;
    if (n == 0) return DAWN_OK;
    DAWN_HIP_TRY(hipMemcpyAsync(idx->d_q, queries, B * dawn::EM * sizeof(float), hipMemcpyHostToDevice, idx->stream));
    if (idx->i8_batched && i8_rows_ready(idx, idx->stream)) {  // (upper-bound scores: scan_i8.hip)
        dawn::launch_batched_dense_scores_i8(idx->d_i8, idx->d_i8meta, (uint32_t)idx->size, idx->d_q, (int)B, idx->bws,
                                             idx->mfma_blocks, idx->stream);
    } else {
        int frt = idx->dtype;
        const void* frows = filter_rows(idx, &frt, idx->stream);
        dawn::launch_batched_dense_scores(frows, frt, (uint32_t)idx->size, idx->d_q, (int)B, idx->bws, idx->mfma_blocks,
                                          idx->stream);
    }
    DAWN_HIP_TRY(hipGetLastError());
    DAWN_HIP_TRY(hipMemcpy2DAsync(out, n * sizeof(float), idx->bws.cand, dawn::BATCH_CAP * sizeof(float),
                                  n * sizeof(float), B, hipMemcpyDeviceToHost, idx->stream));
    DAWN_HIP_TRY(hipStreamSynchronize(idx->stream));
    return DAWN_OK;
}

// Test hook: the per-workgroup candidate lists of the streaming filter for ONE query (what merge_rescore consumes):
// out_scores / out_rows [blocks][64] descending, fillers (-inf, 0xFFFFFFFF); *n_blocks = lists written.
int dawn_index_debug_stream_lists(dawn_index* idx, const float* query, float* out_scores, uint32_t* out_rows,
                                  size_t cap_blocks, size_t* n_blocks) {
    if (!idx || !query || !out_scores || !out_rows || !n_blocks) return fail(DAWN_ERR_INVALID_ARG, "NULL argument");
    DAWN_TRY(set_device(idx));
    DAWN_TRY(ensure_workspace(idx, 1));
    hipStream_t stream = idx->stream;
    DAWN_HIP_TRY(hipMemcpyAsync(idx->d_q, query, dawn::EM * sizeof(float), hipMemcpyHostToDevice, stream));
    int frt = idx->dtype;
    const void* frows = nullptr;
    size_t blocks;
    if (idx->shadow_small_batches && i8_rows_ready(idx, stream)) {
        const dawn::ScanGeom& gh = idx->i8_geom();
        blocks = gh.blocks;
        dawn::launch_scan_filter_i8s(idx->d_i8, idx->d_i8meta, (uint32_t)idx->size, idx->d_q, 1, idx->d_cand_s, idx->d_cand_p, gh,
                                     stream, nullptr, nullptr);
    } else if ((frows = filter_rows(idx, &frt, stream)), frt == dawn::ROW_BF16 || (frt == dawn::ROW_F16S && idx->shadow_small_batches)) {
        const dawn::ScanGeom& gh = idx->shadow_geom();
        blocks = gh.blocks;
        dawn::launch_scan_filter_f16s(frows, frt, (uint32_t)idx->size, idx->d_q, 1, idx->d_cand_s, idx->d_cand_p, gh,
                                      stream, nullptr, nullptr);
    } else {
        blocks = idx->geom.blocks;
        dawn::launch_scan_filter(idx->d_x, idx->dtype, (uint32_t)idx->size, idx->d_q, 1, idx->d_cand_s, idx->d_cand_p,
                                 idx->geom, stream, nullptr, nullptr);
    }
    DAWN_HIP_TRY(hipGetLastError());
    if (blocks > cap_blocks) return fail(DAWN_ERR_INVALID_ARG, "need room for %zu lists", blocks);
    *n_blocks = blocks;
    DAWN_HIP_TRY(hipMemcpyAsync(out_scores, idx->d_cand_s, blocks * dawn::LIST * sizeof(float), hipMemcpyDeviceToHost, stream));
    DAWN_HIP_TRY(hipMemcpyAsync(out_rows, idx->d_cand_p, blocks * dawn::LIST * sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
    DAWN_HIP_TRY(hipStreamSynchronize(stream));
    return DAWN_OK;
}

// Timing hook: mean duration (ms) of the matrix-core full pass alone over `iters` launches for B queries, with the
// thresholds of the last batched search (run one first) and the current mfma_sched variant; results are discarded.
int dawn_index_debug_time_full_pass(dawn_index* idx, size_t B, int iters, double* mean_ms) {
    if (!idx || !mean_ms || iters < 1) return fail(DAWN_ERR_INVALID_ARG, "bad argument");
    if (B == 0 || B > (size_t)dawn::BATCH_QT || !idx->bws.cand) return fail(DAWN_ERR_INVALID_ARG, "run a batched search first");
    DAWN_TRY(set_device(idx));
    hipEvent_t e0, e1;
    DAWN_HIP_TRY(hipEventCreate(&e0));
    DAWN_HIP_TRY(hipEventCreate(&e1));
    if (idx->i8_batched && i8_rows_ready(idx, idx->stream)) {
        dawn::launch_batched_full_pass_i8(idx->d_i8, idx->d_i8meta, (uint32_t)idx->size, (int)B, idx->bws, idx->mfma_blocks, iters,
                                          idx->stream, e0, e1);
    } else {
        int frt = idx->dtype;
        const void* frows = filter_rows(idx, &frt, idx->stream);
        dawn::launch_batched_full_pass(frows, frt, (uint32_t)idx->size, (int)B, idx->bws, idx->mfma_blocks, iters, idx->stream, e0, e1);
    }
    DAWN_HIP_TRY(hipStreamSynchronize(idx->stream));
    float ms = 0.f;
    DAWN_HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    *mean_ms = ms / iters;
    return DAWN_OK;
}

// Diagnostic: per-wave phase cycle sums of the last batched full pass run with mfma_sched = 2: out [blocks][8][8].
int dawn_index_debug_read_diag(dawn_index* idx, unsigned long long* out, size_t blocks) {
    if (!idx || !out) return fail(DAWN_ERR_INVALID_ARG, "NULL argument");
    if (!dawn::g_batched_diag || blocks > 4096) return fail(DAWN_ERR_INVALID_ARG, "no diagnostic buffer");
    DAWN_TRY(set_device(idx));
    DAWN_HIP_TRY(hipDeviceSynchronize());
    DAWN_HIP_TRY(hipMemcpy(out, dawn::g_batched_diag, blocks * 64 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    return DAWN_OK;
}

int dawn_index_set_option(dawn_index* idx, const char* name, int64_t value) {
    if (!idx || !name) return fail(DAWN_ERR_INVALID_ARG, "NULL argument");
    const std::string n(name);
    if (n == "force_fallback") {
        idx->force_fallback = value != 0;
        return DAWN_OK;
    }
    if (n == "scan_blocks") {
        if (value < 1 || value > 65535) return fail(DAWN_ERR_INVALID_ARG, "scan_blocks out of range");
        idx->geom.blocks = (int)value;
        idx->ws_B = 0;  // candidate buffers are sized by the grid
        return DAWN_OK;
    }
    if (n == "mfma_min_batch") {
        if (value < 1) return fail(DAWN_ERR_INVALID_ARG, "mfma_min_batch must be >= 1");
        idx->mfma_min_batch = (int)value;
        return DAWN_OK;
    }
    if (n == "mfma_blocks") {
        if (value < 1 || value > 4096) return fail(DAWN_ERR_INVALID_ARG, "mfma_blocks out of range");
        idx->mfma_blocks = (int)value;
        return DAWN_OK;
    }
    if (n == "scan_unroll") {
        if (value < 1 || value > 4) return fail(DAWN_ERR_INVALID_ARG, "scan_unroll must be 1..4");
        idx->geom.unroll = (int)value;
        return DAWN_OK;
    }
    if (n == "f16_shadow") {
        idx->use_shadow = value != 0;
        return DAWN_OK;
    }
    if (n == "i8_shadow") {
        idx->use_i8 = value != 0;
        if (value) idx->i8_failed = false;
        return DAWN_OK;
    }
    if (n == "i8_batched") {
        idx->i8_batched = value != 0;
        return DAWN_OK;
    }
    if (n == "f16_shadow_b1") {
        idx->shadow_small_batches = value != 0;
        return DAWN_OK;
    }
    if (n == "shadow_scan_blocks" || n == "shadow_scan_threads" || n == "shadow_scan_unroll") {
        idx->geom_h_pinned = true;
        if (n == "shadow_scan_blocks") idx->geom_h.blocks = (int)value, idx->ws_B = 0;
        else if (n == "shadow_scan_threads") {
            if (value != 64 && value != 128 && value != 256 && value != 512)
                return fail(DAWN_ERR_INVALID_ARG, "shadow_scan_threads must be 64/128/256/512");
            idx->geom_h.threads = (int)value;
        } else idx->geom_h.unroll = (int)value;
        return DAWN_OK;
    }
    if (n == "mfma_sched") {
        if (value < 0 || value == 3 || (value > 5 && value < 41) || value > 55)
            return fail(DAWN_ERR_INVALID_ARG, "mfma_sched must be 0, 1, 2, 4 or 5 (41..55: timing experiments)");
        if (value == 2 && !dawn::g_batched_diag) {
            DAWN_HIP_TRY(hipMalloc((void**)&dawn::g_batched_diag, 4096 * 8 * 8 * sizeof(unsigned long long)));
            DAWN_HIP_TRY(hipMemset(dawn::g_batched_diag, 0, 4096 * 8 * 8 * sizeof(unsigned long long)));
        }
        dawn::g_batched_sched = (int)value;
        return DAWN_OK;
    }
    if (n == "mfma_target") {
        if (value < 64 || value > 4096) return fail(DAWN_ERR_INVALID_ARG, "mfma_target must be 64..4096");
        dawn::g_batched_target = (int)value;
        return DAWN_OK;
    }
    if (n == "scan_threads") {
        if (value != 64 && value != 128 && value != 256 && value != 512 && value != 1024)
            return fail(DAWN_ERR_INVALID_ARG, "scan_threads must be 64/128/256/512/1024");
        idx->geom.threads = (int)value;
        return DAWN_OK;
    }
    return fail(DAWN_ERR_INVALID_ARG, "unknown option %s", name);
}

}  // extern "C"
