// dawn_index.cpp — the HBM-resident packed vector index behind the dawn_index_* C ABI.
//
// Replaces usearch::ffi::Index as used by src/search/search_provider.rs (new_index :102, reserve
// :133/:282, add :149/:284, search :214, size/capacity :246/:280, save/load :115-117/:178) with an
// exact brute-force index: rows [N][384] f32 + ids [N] u64 live in one HBM allocation each, appended in
// insertion order; search = scan_i8.hip / scan_kernels.hip / scan_batched.hip.
//
// Division of labour: every MUTATION (create, reserve, add*, fill, load*, set_option) leaves the index ready to be
// searched — workspaces allocated, the filter shadow of the current rows built — and synchronises its own stream; a
// SEARCH of up to 256 queries is then a fixed sequence of kernel launches on the caller's stream: no allocation, no
// synchronisation, no host decision that depends on device data (graph-capturable).  Callers that search on a stream
// of their own must have that stream idle before they mutate the index (the reference's actor never overlaps the two).
#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <cerrno>
#include <cmath>
#include <cstdlib>
#include <cstdio>
#include <cstring>
#include <string>
#include <thread>

#include "index_internal.hpp"

using dawn::fail;

namespace {
constexpr char kMagic[8] = {'D', 'A', 'W', 'N', 'I', 'D', 'X', '1'};
constexpr size_t kPageEntryBytes = 1568;  // src/index/warc.rs:35-43 (repr(C)): 8 + 8 + 384*4 + 8 + 8
constexpr size_t kBulkChunkRows = 32768;  // rows per pinned staging buffer of the bulk loaders (48 MiB)

int set_device(const dawn_index* idx) {
    DAWN_HIP_TRY(hipSetDevice(idx->device));
    return DAWN_OK;
}

size_t padded_rows(size_t rows) { return ((rows + dawn::ROW_PAD - 1) / dawn::ROW_PAD) * dawn::ROW_PAD + dawn::ROW_PAD; }

// Make room for at least `rows` rows (physical).  Live and pending rows are preserved.
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
    const size_t keep = idx->size + idx->pending;
    // (a bf16 index is stored in 64-row tiles: copy and clear whole tiles)
    const size_t live = idx->dtype == DAWN_DTYPE_BF16 ? (keep + dawn::ROW_PAD - 1) / dawn::ROW_PAD * dawn::ROW_PAD : keep;
    if (keep) {
        DAWN_HIP_TRY(hipMemcpyAsync(nx, idx->d_x, live * rb, hipMemcpyDeviceToDevice, idx->stream));
        DAWN_HIP_TRY(hipMemcpyAsync(nid, idx->d_ids, keep * sizeof(uint64_t), hipMemcpyDeviceToDevice, idx->stream));
    }
    // zero everything past the live rows: the scan may read (never use) up to ROW_PAD rows past size
    DAWN_HIP_TRY(hipMemsetAsync(nx + live * rb, 0, (prow - live) * rb, idx->stream));
    // the old rows may still be read by a search a caller issued on a stream of its own
    DAWN_HIP_TRY(hipDeviceSynchronize());
    if (idx->d_x) (void)hipFree(idx->d_x);
    if (idx->d_ids) (void)hipFree(idx->d_ids);
    idx->d_x = nx;
    idx->d_ids = nid;
    idx->cap_phys = rows;
    return DAWN_OK;
}

int ensure_room(dawn_index* idx, size_t extra) {
    const size_t need = idx->size + idx->pending + extra;
    if (need > idx->cap_phys) {
        size_t target = std::max(need, idx->cap_phys + idx->cap_phys / 2);
        target = std::max<size_t>(target, 1024);
        DAWN_TRY(grow_phys(idx, target));
    }
    if (need > idx->cap_reported) idx->cap_reported = need;  // usearch would have required reserve(); we grow
    return DAWN_OK;
}

size_t ws_lists_needed(const dawn_index* idx) {
    return (size_t)std::max({idx->geom.blocks, idx->geom_h.blocks, idx->geom_h_small.blocks, idx->geom_i8.blocks, idx->geom_i6.blocks, idx->geom_i6_small.blocks});
}

// Search workspaces for batches of up to B queries (creation: kMaxBatch; a search_device call with more queries in one
// call grows them — the one case where a search allocates).
int ensure_workspace(dawn_index* idx, size_t B) {
    if (!idx->bws.cand) {
        if (int e = dawn::batched_init()) return fail(DAWN_ERR_HIP, "hipFuncSetAttribute(LDS): %s", hipGetErrorString((hipError_t)e));
        // (16-bit paths: f16 / bf16 query images; int8 path: H images | {s_q, K2} | L images)
        DAWN_HIP_TRY(hipMalloc((void**)&idx->bws.qh, (size_t)dawn::BATCH_QT * (dawn::EM * sizeof(_Float16) + 16)));
        DAWN_HIP_TRY(hipMalloc((void**)&idx->bws.tau, dawn::BATCH_QT * sizeof(float)));
        DAWN_HIP_TRY(hipMalloc((void**)&idx->bws.cnt, dawn::BATCH_QT * dawn::BATCH_CAND_SEGS * sizeof(uint32_t)));
        DAWN_HIP_TRY(hipMalloc(&idx->bws.cand, (size_t)dawn::BATCH_QT * dawn::BATCH_CAP * 8));
        DAWN_HIP_TRY(hipMemset(idx->bws.cnt, 0, dawn::BATCH_QT * dawn::BATCH_CAND_SEGS * sizeof(uint32_t)));
        DAWN_HIP_TRY(hipMalloc((void**)&idx->bws.pool, 32 * sizeof(uint32_t)));
        DAWN_HIP_TRY(hipMemset(idx->bws.pool, 0, 32 * sizeof(uint32_t)));
    }
    if (!idx->d_stats) {
        DAWN_HIP_TRY(hipMalloc((void**)&idx->d_stats, dawn::N_STAT_SLOTS * sizeof(uint32_t)));
        DAWN_HIP_TRY(hipMemset(idx->d_stats, 0, dawn::N_STAT_SLOTS * sizeof(uint32_t)));
    }
    if (!idx->h_stats) {  // the counters' mirror in pinned host memory (ladder feedback)
        DAWN_HIP_TRY(hipHostMalloc((void**)&idx->h_stats, dawn::N_STAT_SLOTS * sizeof(uint32_t), hipHostMallocDefault));
        std::memset(idx->h_stats, 0, dawn::N_STAT_SLOTS * sizeof(uint32_t));
    }
    if (!idx->bounded.wide_res) {  // the wide batch form of the bounded pass: appended exact results per query + their counters (zero between searches)
        DAWN_HIP_TRY(hipMalloc((void**)&idx->bounded.wide_res, (size_t)dawn::BATCH_QT * dawn::BOUNDED_WIDE_CAP * sizeof(uint2)));
        DAWN_HIP_TRY(hipMalloc((void**)&idx->bounded.wide_cnt, dawn::BATCH_QT * sizeof(uint32_t)));
        DAWN_HIP_TRY(hipMemset(idx->bounded.wide_cnt, 0, dawn::BATCH_QT * sizeof(uint32_t)));
    }
    if (!idx->d_i6_pool) {  // chunk counters of the packed stream: zero between searches (merge_exact_kernel resets them)
        DAWN_HIP_TRY(hipMalloc((void**)&idx->d_i6_pool, 32 * sizeof(uint32_t)));
        DAWN_HIP_TRY(hipMemset(idx->d_i6_pool, 0, 32 * sizeof(uint32_t)));
    }
    const size_t lists = ws_lists_needed(idx);
    if (B <= idx->ws_B && lists <= idx->ws_lists) return DAWN_OK;
    B = std::max(B, idx->ws_B);
    DAWN_HIP_TRY(hipDeviceSynchronize());  // nothing may still be using the old buffers
    if (idx->d_cand_s) (void)hipFree(idx->d_cand_s);
    if (idx->d_cand_p) (void)hipFree(idx->d_cand_p);
    if (idx->d_flags) (void)hipFree(idx->d_flags);
    if (idx->d_cand_es) (void)hipFree(idx->d_cand_es);
    if (idx->d_cand_ep) (void)hipFree(idx->d_cand_ep);
    if (idx->d_cand_tb) (void)hipFree(idx->d_cand_tb);
    idx->d_cand_tb = nullptr;
    idx->d_cand_s = nullptr;
    idx->d_cand_p = nullptr;
    idx->d_flags = nullptr;
    idx->d_cand_es = nullptr;
    idx->d_cand_ep = nullptr;
    idx->ws_B = 0;
    const size_t n = B * lists * dawn::LIST;
    DAWN_HIP_TRY(hipMalloc((void**)&idx->d_cand_es, lists * dawn::LIST * sizeof(float)));  // (6-bit stream: one query)
    DAWN_HIP_TRY(hipMalloc((void**)&idx->d_cand_ep, lists * dawn::LIST * sizeof(uint32_t)));
    DAWN_HIP_TRY(hipMalloc((void**)&idx->d_cand_tb, lists * sizeof(float)));
    DAWN_HIP_TRY(hipMalloc((void**)&idx->d_cand_s, n * sizeof(float)));
    DAWN_HIP_TRY(hipMalloc((void**)&idx->d_cand_p, n * sizeof(uint32_t)));
    // flags[B] | arrival counters of the exact pass[B] | launch_i8_rerun's scratch: go word (+ 3 unused) + 256 "lost" marks
    DAWN_HIP_TRY(hipMalloc((void**)&idx->d_flags, (2 * B + 4 + dawn::BATCH_QT) * sizeof(uint32_t)));
    DAWN_HIP_TRY(hipMemset(idx->d_flags, 0, (2 * B + 4 + dawn::BATCH_QT) * sizeof(uint32_t)));
    DAWN_HIP_TRY(hipDeviceSynchronize());  // (searches run on non-blocking streams)
    idx->ws_B = B;
    idx->ws_lists = lists;
    return DAWN_OK;
}

bool enough_free(size_t bytes) {
    size_t fr = 0, tot = 0;
    return hipMemGetInfo(&fr, &tot) == hipSuccess && fr >= bytes + ((size_t)2 << 30);
}

// Bring the f16 shadow up to date with the f32 rows (no-op when it cannot be allocated).
void f16_shadow_sync(dawn_index* idx) {
    hipStream_t stream = idx->stream;
    if (idx->shadow_cap < idx->cap_phys) {
        char* ns = nullptr;
        const size_t prow = padded_rows(idx->cap_phys);
        if ((idx->debug_fail_alloc & 2) || !enough_free(prow * dawn::EM * 2) ||
            hipMalloc((void**)&ns, prow * dawn::EM * 2) != hipSuccess) {
            (void)hipGetLastError();
            idx->shadow_failed = true;
            return;
        }
        (void)hipMemsetAsync(ns, 0, prow * dawn::EM * 2, stream);
        if (idx->d_shadow) {  // keep what is converted already
            (void)hipMemcpyAsync(ns, idx->d_shadow, padded_rows(idx->shadow_rows) * dawn::EM * 2, hipMemcpyDeviceToDevice,
                                 stream);  // whole tiles
            (void)hipDeviceSynchronize();
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
}

// The f16 shadow's memory goes back when no search can read it any more (both integer shadows live again, or "f16_shadow" = 0):
// 76.8 GB per 100 M rows — next to the rows and the two integer shadows a card holds nothing else otherwise.
void f16_shadow_release(dawn_index* idx) {
    if (!idx->d_shadow) return;
    (void)hipDeviceSynchronize();
    (void)hipFree(idx->d_shadow);
    idx->d_shadow = nullptr;
    idx->shadow_cap = idx->shadow_rows = 0;
}

// Bring the int8 shadow up to date; false when it does not fit.
bool i8_shadow_sync(dawn_index* idx) {
    hipStream_t stream = idx->stream;
    if (idx->i8_cap < idx->cap_phys) {
        const size_t prow = padded_rows(idx->cap_phys) + 128;  // (the batched kernel moves 128-row tiles)
        const size_t bytes = prow * dawn::EM, mbytes = (prow / 32 + 1) * 8;
        (void)hipDeviceSynchronize();  // nothing reads the old buffers any more
        if (idx->d_i8) (void)hipFree(idx->d_i8);
        if (idx->d_i8meta) (void)hipFree(idx->d_i8meta);
        idx->d_i8 = nullptr;
        idx->d_i8meta = nullptr;
        idx->i8_cap = 0;
        idx->i8_rows = 0;
        char* ns = nullptr;
        float* nm = nullptr;
        if ((idx->debug_fail_alloc & 1) || !enough_free(bytes + mbytes) || hipMalloc((void**)&ns, bytes) != hipSuccess ||
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
        idx->i8_cap = idx->cap_phys;  // (re-quantised from the rows: 0.03 ms per million rows)
    }
    if (idx->i8_rows < idx->size) {
        dawn::launch_rows_to_i8s(idx->d_x, idx->dtype, idx->d_i8, idx->d_i8meta, idx->i8_rows, idx->size, stream, idx->i8_levels);
        idx->i8_rows = idx->size;
    }
    return true;
}

// The 6-bit shadow is wanted by single-query searches of an index of at least i6_min_rows rows ("i8_shadow" = 0 switches
// both integer shadows off: the 16-bit filters are then what the caller asked for).
bool i6_wanted(const dawn_index* idx) {
    return idx->use_i6 && idx->use_i8 && !idx->i6_failed && idx->shadow_small_batches && idx->size > 0 &&
           idx->size >= idx->i6_min_rows;
}
void i6_release(dawn_index* idx) {
    if (!idx->d_i6 && !idx->d_i6meta) return;
    (void)hipDeviceSynchronize();  // nothing reads the buffers any more
    if (idx->d_i6) (void)hipFree(idx->d_i6);
    if (idx->d_i6meta) (void)hipFree(idx->d_i6meta);
    idx->d_i6 = nullptr;
    idx->d_i6meta = nullptr;
    idx->i6_cap = idx->i6_rows = 0;
}
// Bring the 6-bit shadow up to date; when it is not wanted (any more) its memory goes back (28.8 GB per 100 M rows: next to
// the int8 shadow there is no room for the f16 shadow of 100 M rows otherwise).  false: not live.
bool i6_shadow_sync(dawn_index* idx) {
    hipStream_t stream = idx->stream;
    if (!i6_wanted(idx)) {
        i6_release(idx);
        return false;
    }
    if (idx->i6_cap < idx->cap_phys) {
        const size_t prow = padded_rows(idx->cap_phys) + 128;
        const size_t bytes = prow * idx->i6_row_bytes(), mbytes = (prow / 32 + 1) * 8;
        i6_release(idx);
        char* ns = nullptr;
        float* nm = nullptr;
        if ((idx->debug_fail_alloc & 4) || !enough_free(bytes + mbytes) || hipMalloc((void**)&ns, bytes) != hipSuccess ||
            hipMalloc((void**)&nm, mbytes) != hipSuccess) {
            (void)hipGetLastError();
            if (ns) (void)hipFree(ns);
            idx->i6_failed = true;
            return false;
        }
        (void)hipMemsetAsync(ns, 0, bytes, stream);
        (void)hipMemsetAsync(nm, 0, mbytes, stream);
        idx->d_i6 = ns;
        idx->d_i6meta = nm;
        idx->i6_cap = idx->cap_phys;
    }
    if (idx->i6_rows < idx->size) {
        dawn::launch_rows_to_i6s(idx->d_x, idx->dtype, idx->i6_bits, idx->d_i6, idx->d_i6meta, idx->i6_rows, idx->size, stream);
        idx->i6_rows = idx->size;
        idx->i6_slack_dirty = true;
    }
    if (idx->i6_slack_dirty && idx->i6_slack_model) {
        // what the conversion measured (E per sub-tile) sizes the waves' lists: 256 B back to the host, one wait per change of the shadow
        uint32_t h[64] = {};
        if ((idx->d_i6hist || hipMalloc((void**)&idx->d_i6hist, sizeof(h)) == hipSuccess)) {
            dawn::launch_i6_slack_hist(idx->d_i6meta, (uint32_t)((idx->size + 31) / 32), idx->d_i6hist, stream);
            if (hipMemcpyAsync(h, idx->d_i6hist, sizeof(h), hipMemcpyDeviceToHost, stream) == hipSuccess &&
                hipStreamSynchronize(stream) == hipSuccess) {
                double total = 0.0;
                for (uint32_t v : h) total += v;
                if (total > 0.0) {
                    for (int b = 0; b < 64; ++b) idx->i6_slack.frac[b] = (float)(h[b] / total);
                    // (unique across indexes: i6_refine_count remembers its last answer by this number)
                    static std::atomic<uint32_t> next_version{0};
                    idx->i6_slack.version = ++next_version;
                    idx->i6_slack_dirty = false;
                }
            }
        }
        (void)hipGetLastError();  // (a failure leaves the model on its constants)
    }
    return true;
}

// The FP6 shadow of batches (scan_f6.hip) and its workspaces; released when switched off.  false: not live.
void f6_release(dawn_index* idx) {
    if (!idx->d_f6 && !idx->d_f6meta && !idx->f6ws.cand_big) return;
    (void)hipDeviceSynchronize();
    void* p[] = {idx->d_f6, idx->d_f6meta, idx->f6ws.qf6, idx->f6ws.qmeta, idx->f6ws.tau6, idx->f6ws.cand_big, idx->f6ws.cnt_big};
    for (void* q : p)
        if (q) (void)hipFree(q);
    idx->d_f6 = nullptr;
    idx->d_f6meta = nullptr;
    const int target = idx->f6ws.target, stagger = idx->f6ws.stagger, refine_rows = idx->f6ws.refine_rows;
    idx->f6ws = dawn::F6Workspace{};
    idx->f6ws.target = target;
    idx->f6ws.stagger = stagger;
    idx->f6ws.refine_rows = refine_rows;
    idx->f6_cap = idx->f6_rows = 0;
}
bool f6_shadow_sync(dawn_index* idx) {
    hipStream_t stream = idx->stream;
    const bool wanted = idx->use_f6 && idx->use_i8 && idx->i8_batched && !idx->f6_failed && idx->size >= idx->f6_min_rows && idx->size > 0;
    if (!wanted) {
        f6_release(idx);
        return false;
    }
    if (!idx->f6ws.cand_big) {
        const size_t big = (size_t)dawn::BATCH_QT * dawn::BATCH_CAND_SEGS * idx->f6ws.seg_cap_big * 8;
        if (hipMalloc(&idx->f6ws.qf6, 16 * 3 * 64 * 6 * 4) != hipSuccess || hipMalloc(&idx->f6ws.qmeta, dawn::BATCH_QT * 8) != hipSuccess ||
            hipMalloc((void**)&idx->f6ws.tau6, dawn::BATCH_QT * 4) != hipSuccess || hipMalloc(&idx->f6ws.cand_big, big) != hipSuccess ||
            hipMalloc((void**)&idx->f6ws.cnt_big, dawn::BATCH_QT * dawn::BATCH_CAND_SEGS * 4) != hipSuccess) {
            (void)hipGetLastError();
            idx->f6_failed = true;
            f6_release(idx);
            return false;
        }
        (void)hipMemsetAsync(idx->f6ws.cnt_big, 0, dawn::BATCH_QT * dawn::BATCH_CAND_SEGS * 4, stream);
    }
    if (idx->f6_cap < idx->cap_phys) {
        const size_t prow = padded_rows(idx->cap_phys) + 128;
        const size_t bytes = prow * 288, mbytes = (prow / 16 + 1) * 8;
        (void)hipDeviceSynchronize();
        if (idx->d_f6) (void)hipFree(idx->d_f6);
        if (idx->d_f6meta) (void)hipFree(idx->d_f6meta);
        idx->d_f6 = nullptr;
        idx->d_f6meta = nullptr;
        idx->f6_cap = idx->f6_rows = 0;
        char* ns = nullptr;
        float* nm = nullptr;
        // auto (the default): the FP6 shadow is an OPTIONAL accelerator of batches (+ 288 B per row for ~ -10 % per batch) — it is only built
        // where it leaves dawn::kF6AutoHeadroom of HBM free for whatever else the caller keeps on the card (other indexes, the embedder);
        // the order when HBM runs out is therefore: FP6 shadow first to go, then packed shadow -> int8 shadow -> f16 shadow -> the rows
        const size_t spare = idx->use_f6 == 2 ? dawn::kF6AutoHeadroom : 0;
        if ((idx->debug_fail_alloc & 8) || !enough_free(bytes + mbytes + spare) || hipMalloc((void**)&ns, bytes) != hipSuccess ||
            hipMalloc((void**)&nm, mbytes) != hipSuccess) {
            (void)hipGetLastError();
            if (ns) (void)hipFree(ns);
            idx->f6_failed = true;
            f6_release(idx);
            return false;
        }
        (void)hipMemsetAsync(ns, 0, bytes, stream);
        (void)hipMemsetAsync(nm, 0, mbytes, stream);
        idx->d_f6 = ns;
        idx->d_f6meta = nm;
        idx->f6_cap = idx->cap_phys;
    }
    if (idx->f6_rows < idx->size) {
        dawn::launch_rows_to_f6s(idx->d_x, idx->dtype, idx->d_f6, idx->d_f6meta, idx->f6_rows, idx->size, stream);
        idx->f6_rows = idx->size;
    }
    return true;
}
bool f6_live(const dawn_index* idx) {
    return idx->use_f6 && idx->d_f6 && idx->f6_rows == idx->size && idx->size >= idx->f6_min_rows && idx->f6ws.cand_big;
}

bool i8_live(const dawn_index* idx);
// (its stream refines the listed rows on the int8 shadow: no int8 shadow, no packed stream)
// The bounded pass of a single query on the packed 5-bit shadow instead of the int8 one: 240 instead of 384 B per row, but a bound
// seven times as loose — more rows reach the exact scores, and a pass that starts without a threshold needs much longer to find
// one.  With a first threshold (the failed packed stream's k-th distance, or the seed below) it wins at every size that keeps a
// packed shadow; without one it loses below ~40 M rows.  Measured on topical rows, mean ms per demoted query (profiles/r04/
// bounded_packed_ab_*.log; int8 / packed unseeded / packed seeded): 2.5 M rows 0.314 / 0.436 / 0.273, 5 M 0.463 / 0.608 / 0.382,
// 12.5 M 0.881 / 1.055 / 0.684, 25 M 1.58 / 1.67 / 1.18, 100 M 5.79 / 4.97 / 4.17 (k = 20: 5.85 / 5.35 / 4.20).
// Option "bounded_packed": 0 never, 1 (default) wherever the packed shadow is live from 2 Mi rows, 2 always (tests).
static bool bounded_packed_wanted(const dawn_index* idx, uint32_t n) {
    if (idx->i6_bits != 5) return false;
    return idx->bounded_packed == 2 || (idx->bounded_packed == 1 && n >= (2u << 20));
}
// ... and the seed of a pass that would otherwise start without a threshold (a demoted query): the packed stream over the first 1/32
// of the rows — its k-th exact distance bounds the final one from above whatever its own certificate says.  Option "bounded_seed":
// 0 never (the packed form is then only used from 40 Mi rows), 1 (default) from 2 Mi rows, 2 from 32 Ki rows (tests).
static bool bounded_seed_wanted(const dawn_index* idx, uint32_t n) {
    if (!idx->bounded_seed || idx->debug_bad_threshold) return false;
    return (n >> idx->bounded_seed_shift) >= (idx->bounded_seed == 2 ? 1024u : (64u << 10));
}

bool i6_live(const dawn_index* idx) { return i6_wanted(idx) && idx->d_i6 && idx->i6_rows == idx->size && i8_live(idx); }
bool i8_live(const dawn_index* idx) {
    return idx->use_i8 && !idx->i8_failed && idx->i8_rows == idx->size && (idx->d_i8 || idx->size == 0);
}
bool f16_live(const dawn_index* idx) {
    return idx->dtype == DAWN_DTYPE_F32 && idx->use_shadow && !idx->shadow_failed && idx->d_shadow &&
           idx->shadow_rows == idx->size;
}

int ensure_stage(dawn_index* idx, size_t bytes) {
    if (bytes <= idx->stage_bytes) return DAWN_OK;
    DAWN_HIP_TRY(hipStreamSynchronize(idx->stream));  // earlier chunks may still be passing through it
    if (idx->d_stage) (void)hipFree(idx->d_stage);
    idx->d_stage = nullptr;
    idx->stage_bytes = 0;
    DAWN_HIP_TRY(hipMalloc((void**)&idx->d_stage, bytes));
    idx->stage_bytes = bytes;
    return DAWN_OK;
}

}  // namespace

namespace dawn {

// Which filter source the searches of this index will read, given its options (mirrors index_search_on_device), built on
// idx->stream.  An option that needs a shadow the index does not hold yet builds it here — never inside a search.
int index_prepare_search(dawn_index* idx, bool appended) {
    DAWN_TRY(ensure_workspace(idx, std::max(idx->ws_B, kMaxBatch)));
    // The feedback (what the certificates of this index did lately) starts over when the options change or the contents are replaced.
    // Appends keep it — the reference crawls while it serves (src/search/search_service.rs:158-171: an insert per extracted page between
    // searches), and an index whose windows are reset by every flush of staged adds never demotes or deepens — until the
    // index has grown by an eighth since the feedback last started over: by then the rows it was learnt on are not the index any more.
    if (!appended || idx->size < 1024 || idx->size - idx->fb_rows0 > idx->fb_rows0 / 8) {
        idx->fb = dawn_index::LadderFeedback{};
        idx->f6fb = dawn_index::F6Feedback{};
        idx->bfb = dawn_index::BatchFeedback{};
        idx->fb_rows0 = idx->size;
        if (idx->h_stats) idx->fb.win_fail0 = reinterpret_cast<volatile uint32_t*>(idx->h_stats)[STAT_PACKED_FAIL];
    }
    if (idx->size == 0) return DAWN_OK;
    bool i8_ok = false;
    if (idx->use_i8 && !idx->i8_failed && (idx->i8_batched || idx->shadow_small_batches)) i8_ok = i8_shadow_sync(idx);
    // what is not wanted any more gives its memory back BEFORE anything new is built (100 M rows: the rows and all three shadows
    // do not fit one card together)
    bool f16_needed = false;
    if (idx->dtype == DAWN_DTYPE_F32 && idx->use_shadow && !idx->shadow_failed) {
        const bool batched_needs = !(i8_ok && idx->i8_batched);
        const bool small_needs = idx->shadow_small_batches && !i8_ok;
        f16_needed = batched_needs || small_needs;
    }
    if (!f16_needed) f16_shadow_release(idx);
    (void)i6_shadow_sync(idx);  // single queries stream the 6-bit shadow (released when not wanted)
    (void)f6_shadow_sync(idx);  // batches of a large index filter on the FP6 shadow first (option "f6_shadow")
    if (f16_needed) f16_shadow_sync(idx);
    DAWN_HIP_TRY(hipGetLastError());
    return DAWN_OK;
}

// The whole search as a fixed launch sequence on `stream` (no host decisions in between).
int index_search_on_device(dawn_index* idx, const float* d_q, size_t B, size_t k, uint64_t* d_labels, float* d_dist,
                           uint32_t* d_found, hipStream_t stream) {
    if (B > idx->ws_B) DAWN_TRY(ensure_workspace(idx, B));  // more than kMaxBatch queries in one device call
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (idx->profiling && idx->events_used < idx->events.size()) {
        e0 = idx->events[idx->events_used].first;
        e1 = idx->events[idx->events_used].second;
        idx->events_used++;
    }
    const uint32_t n = (uint32_t)idx->size;
    // (mfma_min_batch = 0, the default: by size — the matrix-core pass costs the same for 2 queries as for 256, the streaming filter
    // takes up to 8 queries through ONE stream of the int8 shadow but keeps per-wave lists per query: on large indexes it wins up
    // to 4 queries — 12.5 M rows x 2: 0.84 against 1.23 ms, 100 M x 4: 5.7 against 6.5 —, on small ones the pass does from 2:
    // profiles/r05/small_batch_paths.log)
    bool batched = idx->mfma_min_batch > 0 ? (int)B >= idx->mfma_min_batch : B >= 2;
    if (idx->mfma_min_batch == 0 && batched && idx->shadow_small_batches && i8_live(idx) &&
        ((B <= 4 && (size_t)n >= (B - 1) * (size_t)1250000) || (B <= 6 && n >= (64u << 20))))
        batched = false;
    bool use_f6 = batched && idx->i8_batched && i8_live(idx) && f6_live(idx) && n > (uint32_t)BATCH_CAP;
    if (use_f6 && idx->ladder_feedback && idx->h_stats && !idx->force_fallback) {
        // FP6 feedback (index_internal.hpp): does this index send too many of its FP6-filtered queries to the ladder?
        auto& fb = idx->f6fb;
        const volatile uint32_t* hs = reinterpret_cast<volatile uint32_t*>(idx->h_stats);
        // (a window of kF6FbWindow queries; or, from 256 queries on, as soon as MORE THAN HALF of them ended in the ladder — a topical
        // index: an FP6-filtered batch there costs several times an int8-filtered one, the verdict need not wait for four of them)
        if (fb.suspend_left == 0 && fb.issued >= 256) {
            const uint32_t now = hs[FLAG_BOUNDED] + hs[FLAG_FALLBACK];
            const double in_ladder = (double)(now - fb.ladder0);
            if (in_ladder > 0.5 * (double)fb.issued || (fb.issued >= kF6FbWindow && in_ladder > kF6FbSuspend * (double)fb.issued)) {
                fb.suspend_left = fb.suspend_len;
                fb.suspend_len = std::min(fb.suspend_len * 2u, kF6FbSuspendMax);
                fb.issued = 0;
            } else if (fb.issued >= kF6FbWindow) {
                fb.suspend_len = kF6FbSuspendMin;
                fb.issued = 0;
            }
        }
        // ... and no probe while the int8 pass's own batches are ladder-heavy: the FP6 bound is the looser one by construction, it can
        // only send more of a batch there (the batch feedback's last full window at its base depth, measured while this filter is suspended)
        if (fb.suspend_left == 1 && idx->bfb.base_per_pass > kBatchFbBoost * (double)BATCH_QT) fb.suspend_left = fb.suspend_len;
        if (fb.suspend_left > 0) {
            --fb.suspend_left;
            ++idx->n_f6_suspended;
            use_f6 = false;
        } else {
            if (fb.issued == 0) fb.ladder0 = hs[FLAG_BOUNDED] + hs[FLAG_FALLBACK];
            fb.issued += B;
        }
    }
    if (use_f6) {
        // matrix-core path with the FP6 shadow as first filter (scan_f6.hip), its survivors re-scored on the f32 rows / the int8 shadow
        idx->n_f6_batches += (B + BATCH_QT - 1) / BATCH_QT;
        for (size_t b0 = 0; b0 < B; b0 += BATCH_QT) {
            const size_t nb = std::min<size_t>(BATCH_QT, B - b0);
            launch_scan_batched_f6(idx->d_x, idx->dtype, idx->d_i8, idx->d_i8meta, idx->d_f6, idx->d_f6meta, idx->d_ids, n,
                                   d_q + b0 * EM, (int)nb, (uint32_t)k, idx->bws, idx->f6ws, idx->mfma_blocks, d_labels + b0 * k,
                                   d_dist + b0 * k, d_found + b0, idx->d_flags + b0, idx->force_fallback, stream,
                                   b0 == 0 ? e0 : nullptr, b0 == 0 ? e1 : nullptr);
        }
    } else if (batched && idx->i8_batched && i8_live(idx)) {
        // matrix-core path on the int8 shadow (v_mfma_i32_32x32x32_i8, upper-bound scores), BATCH_QT queries per pass
        dawn::BatchWorkspace bw = idx->bws;
        bool rerun = false;  // the flagged queries of a batch get a second pass with exact-derived thresholds (scan_i8.hip: launch_i8_rerun)
        if (idx->ladder_feedback && idx->h_stats && !idx->force_fallback && n > (uint32_t)BATCH_CAP) {
            // batch feedback (index_internal.hpp): too many of this index's batched queries in the ladder -> deeper thresholds
            auto& fb = idx->bfb;
            const volatile uint32_t* hs = reinterpret_cast<volatile uint32_t*>(idx->h_stats);
            if (fb.issued >= kBatchFbWindow) {
                const uint32_t now = hs[FLAG_BOUNDED] + hs[FLAG_FALLBACK];
                const double per_pass = (double)(now - fb.ladder0) / (double)std::max<uint64_t>(fb.passes, 1);
                auto streams = [](double flagged) { return flagged <= 0.0 ? 0.0 : std::ceil(flagged / 64.0 - 0.02); };  // (wide form: 64 per stream)
                if (fb.level == 1) {
                    fb.base_per_pass = per_pass;
                    if ((double)(now - fb.ladder0) > kBatchFbBoost * (double)fb.issued && !fb.tried_deep) fb.level = 2;  // a trial window
                    else if (now == fb.ladder0 && !fb.no_shallow) fb.level = 0;
                } else if (fb.level == 2 && !fb.tried_deep) {
                    fb.tried_deep = true;  // keep the deeper thresholds only where they save the ladder a stream
                    if (!(streams(per_pass) < streams(fb.base_per_pass))) fb.level = 1;
                } else if (fb.level == 0 && now != fb.ladder0) {
                    fb.level = 1;
                    fb.no_shallow = true;
                }
                fb.issued = 0;
                fb.passes = 0;
            }
            if (fb.issued == 0) fb.ladder0 = hs[FLAG_BOUNDED] + hs[FLAG_FALLBACK];
            fb.issued += B;
            fb.passes += (B + BATCH_QT - 1) / BATCH_QT;
            if (fb.level == 2) {
                bw.target = std::max(bw.target, k > 32 ? kBatchBoostTargetWide : kBatchBoostTarget);
                idx->n_deepened_batches += (B + BATCH_QT - 1) / BATCH_QT;
                rerun = idx->batch_rerun != 0 && idx->bounded_pass;
            } else if (fb.level == 0) {
                bw.target = std::max(bw.target / 2, 256);
            }
        }
        if (idx->batch_rerun == 2 && n > (uint32_t)BATCH_CAP && idx->force_fallback != 1) rerun = true;  // (tests, A/B: every batch)
        for (size_t b0 = 0; b0 < B; b0 += BATCH_QT) {
            const size_t nb = std::min<size_t>(BATCH_QT, B - b0);
            launch_scan_batched_i8(idx->d_x, idx->dtype, idx->d_i8, idx->d_i8meta, idx->d_ids, n, d_q + b0 * EM, (int)nb,
                                   (uint32_t)k, bw, idx->mfma_blocks, d_labels + b0 * k, d_dist + b0 * k, d_found + b0,
                                   idx->d_flags + b0, idx->force_fallback, stream, b0 == 0 ? e0 : nullptr,
                                   b0 == 0 ? e1 : nullptr);
            if (rerun)
                launch_i8_rerun(idx->d_x, idx->dtype, idx->d_i8, idx->d_i8meta, idx->d_ids, n, d_q + b0 * EM, (int)nb, (uint32_t)k, bw,
                                idx->mfma_blocks, d_labels + b0 * k, d_dist + b0 * k, d_found + b0, idx->d_flags + b0,
                                idx->d_flags + 2 * idx->ws_B, stream);
        }
    } else if (batched) {
        // matrix-core path on 16-bit rows: the f16 shadow of an f32 index, a bf16 index itself; an f32 index without a
        // shadow (or mfma_sched 0 / 2): the lockstep kernel converts the f32 rows on the fly
        int frt = idx->dtype;
        const void* frows = idx->d_x;
        if (idx->dtype == DAWN_DTYPE_F32 && idx->bws.sched != 0 && idx->bws.sched != 2 && f16_live(idx)) {
            frows = idx->d_shadow;
            frt = ROW_F16S;
        }
        for (size_t b0 = 0; b0 < B; b0 += BATCH_QT) {
            const size_t nb = std::min<size_t>(BATCH_QT, B - b0);
            launch_scan_batched(idx->d_x, idx->dtype, frows, frt, idx->d_ids, n, d_q + b0 * EM, (int)nb, (uint32_t)k, idx->bws,
                                idx->mfma_blocks, d_labels + b0 * k, d_dist + b0 * k, d_found + b0, idx->d_flags + b0,
                                idx->force_fallback, stream, b0 == 0 ? e0 : nullptr, b0 == 0 ? e1 : nullptr);
        }
    } else if (B == 1 && i6_live(idx) &&
               (idx->i6_geom().refine > 0 ||
                i6_refine_count(n, (uint32_t)k, idx->i6_bits, idx->i6_geom().blocks * (idx->i6_geom().threads / 64), idx->i6_slack_ptr()) > 0)) {
        // ladder feedback (index_internal.hpp): how often did the packed certificate fail lately?
        bool demoted = false;
        if (idx->ladder_feedback && idx->bounded_pass && !idx->force_fallback && idx->h_stats) {
            auto& fb = idx->fb;
            if (fb.demote_left == 0 && fb.issued - fb.win_issued0 >= kFbWindow) {
                const uint32_t now = reinterpret_cast<volatile uint32_t*>(idx->h_stats)[STAT_PACKED_FAIL];
                const double rate = (double)(now - fb.win_fail0) / (double)(fb.issued - fb.win_issued0);
                if (rate > (bounded_packed_wanted(idx, n) && idx->bounded_seed && n >= (40u << 20) ? kFbDemotePacked : kFbDemote)) {
                    fb.demote_left = fb.demote_len;
                    fb.demote_len = std::min(fb.demote_len * 2u, kFbDemoteMax);
                } else {
                    fb.demote_len = kFbDemoteMin;
                }
                if (rate > kFbBoost) fb.boosted = true;
                fb.win_issued0 = fb.issued;
                fb.win_fail0 = now;
            }
            if (fb.demote_left > 0) {
                --fb.demote_left;
                demoted = true;
            } else {
                ++fb.issued;
            }
            if (idx->ladder_feedback == 2) demoted = true;  // (tests / A-B: the bounded pass is the whole search of every single query)
        }
        if (demoted) {
            // a demoted index: the bounded exact pass is the whole search (240 B/row on the packed shadow, no certificate to fail)
            ++idx->n_demoted;
            // (unseeded, the packed form only pays on very large indexes)
            const bool seed_ok = bounded_seed_wanted(idx, n);
            const bool p5 = bounded_packed_wanted(idx, n) && (seed_ok || idx->bounded_packed == 2 || n >= (40u << 20));
            // A pass that starts without a threshold scores rows exactly until its waves have found k good ones each — with the
            // packed shadow's loose bound that costs ~0.75 ms of a 5-ms pass at 100 M rows (a pass behind a failed packed stream,
            // which hands over its k-th distance: 4.2 ms).  The packed stream over the first 1/32 of the rows (0.15 ms) finds a
            // k-th exact distance that bounds the final one from above just as well (option "bounded_seed").
            const bool seed = p5 && seed_ok;
            if (seed) {
                ScanGeom g6 = idx->i6_geom();
                if (g6.refine == 0)
                    g6.refine = std::max(0, i6_refine_count(n >> idx->bounded_seed_shift, (uint32_t)k, idx->i6_bits, g6.blocks * (g6.threads / 64),
                                                            idx->i6_slack_ptr()));
                launch_scan_i6(idx->d_i6, idx->d_i6meta, idx->i6_bits, idx->d_i8, idx->d_i8meta, idx->d_x, idx->dtype, idx->d_ids,
                               n >> idx->bounded_seed_shift, d_q, idx->d_cand_s, idx->d_cand_p, idx->d_cand_es, idx->d_cand_ep, idx->d_cand_tb,
                               idx->stream_dyn_tail ? idx->d_i6_pool : nullptr, g6, (uint32_t)k, d_labels, d_dist, d_found, idx->d_flags,
                               0, true, stream, e0, nullptr, nullptr, idx->i6_central_tail != 0);  // (the profile of a demoted search includes its seed)
            }
            launch_scan_bounded_direct(idx->d_i8, idx->d_i8meta, idx->d_x, idx->dtype, idx->d_ids, n, d_q, idx->d_flags,
                                       idx->d_flags + idx->ws_B, idx->d_cand_s, idx->d_cand_p, idx->geom_i8.blocks, (uint32_t)k,
                                       d_labels, d_dist, d_found, stream, seed ? nullptr : e0, e1, idx->d_stats, idx->h_stats, idx->bounded,
                                       idx->debug_bad_threshold ? -1.0f : __builtin_inff(), p5 ? idx->d_i6 : nullptr,
                                       p5 ? idx->d_i6meta : nullptr, seed);
            DAWN_HIP_TRY(hipGetLastError());
            return DAWN_OK;
        }
        // one query on the packed shadow (240 B/row): upper-bound scores, every workgroup's shortlist rescored exactly in the
        // stream's epilogue, one merge + certificate (scan_i6.hip)
        ScanGeom g6 = idx->i6_geom();
        if (idx->fb.boosted && g6.refine == 0) g6.refine = LIST;
        if (g6.refine == 0) g6.refine = i6_refine_count(n, (uint32_t)k, idx->i6_bits, g6.blocks * (g6.threads / 64), idx->i6_slack_ptr());
        launch_scan_i6(idx->d_i6, idx->d_i6meta, idx->i6_bits, idx->d_i8, idx->d_i8meta, idx->d_x, idx->dtype, idx->d_ids, n, d_q,
                       idx->d_cand_s, idx->d_cand_p, idx->d_cand_es, idx->d_cand_ep, idx->d_cand_tb, idx->stream_dyn_tail ? idx->d_i6_pool : nullptr,
                       g6, (uint32_t)k, d_labels, d_dist, d_found, idx->d_flags,
                       idx->force_fallback, true, stream, e0, e1, idx->d_stats, idx->i6_central_tail != 0);
    } else if (idx->shadow_small_batches && i8_live(idx)) {
        // 1..8 queries on the int8 shadow (384 B/row): the filter scores are upper bounds of the exact ones
        const ScanGeom& gh = idx->i8_geom();
        launch_scan_filter_i8s(idx->d_i8, idx->d_i8meta, n, d_q, (int)B, idx->d_cand_s, idx->d_cand_p, gh, stream, e0, e1);
        launch_merge_rescore(idx->d_x, idx->dtype, idx->d_ids, n, d_q, (int)B, idx->d_cand_s, idx->d_cand_p, gh.blocks,
                             (uint32_t)k, d_labels, d_dist, d_found, idx->d_flags, idx->force_fallback, FILTER_EPS_I8, stream);
    } else if (idx->dtype == DAWN_DTYPE_BF16 || (idx->shadow_small_batches && f16_live(idx))) {
        // 1..8 queries: stream the 16-bit fragments (768 B/row: the f16 shadow instead of the f32 rows, or the bf16
        // index itself) through the matrix cores
        const bool own = idx->dtype == DAWN_DTYPE_BF16;
        const ScanGeom& gh = idx->shadow_geom();
        launch_scan_filter_f16s(own ? idx->d_x : idx->d_shadow, own ? ROW_BF16 : ROW_F16S, n, d_q, (int)B, idx->d_cand_s,
                                idx->d_cand_p, gh, stream, e0, e1);
        launch_merge_rescore(idx->d_x, idx->dtype, idx->d_ids, n, d_q, (int)B, idx->d_cand_s, idx->d_cand_p, gh.blocks,
                             (uint32_t)k, d_labels, d_dist, d_found, idx->d_flags, idx->force_fallback,
                             own ? FILTER_EPS_BF16_STREAM : FILTER_EPS_F16, stream);
    } else {
        // the f32 rows themselves (1536 B/row)
        uint32_t* pool = idx->stream_dyn_tail ? idx->d_i6_pool : nullptr;
        launch_scan_filter(idx->d_x, idx->dtype, n, d_q, (int)B, idx->d_cand_s, idx->d_cand_p, idx->geom, stream, e0, e1, pool);
        launch_merge_rescore(idx->d_x, idx->dtype, idx->d_ids, n, d_q, (int)B, idx->d_cand_s, idx->d_cand_p, idx->geom.blocks,
                             (uint32_t)k, d_labels, d_dist, d_found, idx->d_flags, idx->force_fallback, FILTER_EPS_F32, stream,
                             pool);
    }
    // The ladder behind a failed certificate, predicated on the query's flag (launch-only): an index that keeps an int8 shadow
    // closes every search with the bounded exact pass (scan_bounded.hip: 384 B per row + the rows that can still matter; it
    // cannot fail, keeps the certificate counters and mirrors them to the host), the others with the exact pass over all rows.
    // force_fallback = 1 (tests of the exact pass) takes the second form; 2 forces the flags only.
    // (a single query of an index with a live packed 5-bit shadow streams that one: 240 instead of 384 B per row; option "bounded_packed")
    // (a batch: the packed form of the multi kernel, option "bounded_multi_packed")
    const bool packed5 = bounded_packed_wanted(idx, n) && i6_live(idx) && idx->i6_bits == 5;
    if (idx->bounded_pass && idx->force_fallback != 1 && i8_live(idx) && n > 0)
        launch_scan_bounded(idx->d_i8, idx->d_i8meta, idx->d_x, idx->dtype, idx->d_ids, n, d_q, (int)B, idx->d_flags,
                            idx->d_flags + idx->ws_B, idx->d_cand_s, idx->d_cand_p, idx->geom_i8.blocks, (uint32_t)k, d_labels,
                            d_dist, d_found, stream, idx->d_stats, idx->h_stats, idx->bounded, packed5 ? idx->d_i6 : nullptr,
                            packed5 ? idx->d_i6meta : nullptr);
    else
        launch_scan_exact(idx->d_x, idx->dtype, idx->d_ids, n, d_q, (int)B, idx->d_flags, idx->d_flags + idx->ws_B, idx->d_stats,
                          idx->d_cand_s, idx->d_cand_p, idx->geom.blocks, (uint32_t)k, d_labels, d_dist, d_found, stream,
                          idx->h_stats);
    DAWN_HIP_TRY(hipGetLastError());
    return DAWN_OK;
}

int index_create_single(int dtype, int device, dawn_index** out) {
    *out = nullptr;
    DAWN_TRY(require_device(device));
    DAWN_HIP_TRY(hipSetDevice(device));
    auto* idx = new dawn_index();
    idx->device = device;
    idx->dtype = dtype;
    if (const char* e = getenv("DAWN_I8_SHADOW")) idx->use_i8 = atoi(e) != 0;  // default of the "i8_shadow" option
    if (const char* e = getenv("DAWN_I6_SHADOW")) idx->use_i6 = atoi(e) != 0;  // default of the "i6_shadow" option
    if (const char* e = getenv("DAWN_I6_BITS")) idx->i6_bits = atoi(e) == 6 ? 6 : 5;
    if (const char* e = getenv("DAWN_I6_MIN_ROWS")) idx->i6_min_rows = (size_t)std::max(0ll, atoll(e));
    if (const char* e = getenv("DAWN_F6_SHADOW")) idx->use_f6 = std::min(2, std::max(0, atoi(e)));  // default of the "f6_shadow" option
    hipDeviceProp_t prop{};
    if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0) {
        idx->geom.blocks = prop.multiProcessorCount;  // one block per CU
        idx->geom_h.blocks = prop.multiProcessorCount;
        idx->geom_h_small.blocks = prop.multiProcessorCount;
        idx->geom_i8.blocks = prop.multiProcessorCount;
        idx->geom_i6.blocks = prop.multiProcessorCount;
        idx->geom_i6_small.blocks = prop.multiProcessorCount;
        idx->mfma_blocks = prop.multiProcessorCount;
    }
    hipError_t e = hipStreamCreateWithFlags(&idx->stream, hipStreamNonBlocking);
    if (e != hipSuccess) {
        delete idx;
        return fail(DAWN_ERR_HIP, "hipStreamCreate: %s", hipGetErrorString(e));
    }
    const size_t hb = kMaxBatch * (EM * 4 + DAWN_MAX_K * 8 + DAWN_MAX_K * 4 + 8);
    if (hipHostMalloc(&idx->h_pinned, hb, hipHostMallocDefault) != hipSuccess ||
        hipMalloc((void**)&idx->d_q, kMaxBatch * EM * sizeof(float)) != hipSuccess ||
        hipMalloc((void**)&idx->d_labels, kMaxBatch * DAWN_MAX_K * sizeof(uint64_t)) != hipSuccess ||
        hipMalloc((void**)&idx->d_dist, kMaxBatch * DAWN_MAX_K * sizeof(float)) != hipSuccess ||
        hipMalloc((void**)&idx->d_found, kMaxBatch * sizeof(uint32_t)) != hipSuccess ||
        hipMalloc((void**)&idx->d_bad, sizeof(uint32_t)) != hipSuccess ||
        hipEventCreateWithFlags(&idx->ev_slot[0], hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&idx->ev_slot[1], hipEventDisableTiming) != hipSuccess) {
        index_destroy_single(idx);
        return fail(DAWN_ERR_OOM, "allocating index staging buffers failed");
    }
    idx->h_pinned_bytes = hb;
    int rc = index_prepare_search(idx);
    if (rc == DAWN_OK && hipStreamSynchronize(idx->stream) != hipSuccess) rc = fail(DAWN_ERR_HIP, "index creation failed");
    if (rc != DAWN_OK) {
        index_destroy_single(idx);
        return rc;
    }
    *out = idx;
    return DAWN_OK;
}

void index_destroy_single(dawn_index* idx) {
    if (!idx) return;
    (void)hipSetDevice(idx->device);
    (void)hipDeviceSynchronize();
    for (auto& ev : idx->events) {
        (void)hipEventDestroy(ev.first);
        (void)hipEventDestroy(ev.second);
    }
    for (hipEvent_t ev : idx->ev_slot)
        if (ev) (void)hipEventDestroy(ev);
    void* ptrs[] = {idx->d_x, idx->d_shadow, idx->d_i8, idx->d_i8meta, idx->d_i6, idx->d_i6meta, idx->d_cand_es, idx->d_cand_ep, idx->d_cand_tb, idx->d_i6_pool, idx->d_i6hist, idx->d_stage, idx->d_ids, idx->d_cand_s, idx->d_cand_p,
                    idx->d_flags, idx->d_stats, idx->bounded.wide_res, idx->bounded.wide_cnt, idx->bws.qh, idx->bws.tau, idx->bws.cnt, idx->bws.cand, idx->bws.diag, idx->bws.pool, idx->d_q,
                    idx->d_labels, idx->d_dist, idx->d_found, idx->d_bad};
    for (void* p : ptrs)
        if (p) (void)hipFree(p);
    if (idx->h_pinned) (void)hipHostFree(idx->h_pinned);
    if (idx->h_stats) (void)hipHostFree(idx->h_stats);
    f6_release(idx);
    if (idx->stream) (void)hipStreamDestroy(idx->stream);
    delete idx;
}

int index_reserve_single(dawn_index* idx, size_t capacity) {
    DAWN_TRY(set_device(idx));
    if (capacity > idx->cap_phys) {
        DAWN_TRY(grow_phys(idx, capacity));
        DAWN_TRY(index_prepare_search(idx));  // the shadows move with the rows
        DAWN_HIP_TRY(hipStreamSynchronize(idx->stream));
    }
    if (capacity > idx->cap_reported) idx->cap_reported = capacity;
    return DAWN_OK;
}

int index_append_async(dawn_index* idx, RowSrc kind, const void* h_src, const uint64_t* h_ids, uint64_t first_label, size_t m,
                       int slot) {
    if (idx->shards) return sharded_append_async(idx, kind, h_src, h_ids, first_label, m, slot);
    if (m == 0) return DAWN_OK;
    DAWN_TRY(set_device(idx));
    DAWN_TRY(ensure_room(idx, m));
    hipStream_t st = idx->stream;
    if (idx->pending == 0) DAWN_HIP_TRY(hipMemsetAsync(idx->d_bad, 0, sizeof(uint32_t), st));
    const bool bf16 = idx->dtype == DAWN_DTYPE_BF16;
    const size_t at = idx->size + idx->pending;
    const size_t rb = idx->row_bytes();
    // rows land past `size` (invisible to searches) and become live only after validation
    if (kind == RowSrc::HostRows && !bf16) {
        float* dst = reinterpret_cast<float*>(idx->d_x + at * rb);
        DAWN_HIP_TRY(hipMemcpyAsync(dst, h_src, m * rb, hipMemcpyHostToDevice, st));
        launch_validate_rows(dst, (uint32_t)m, idx->d_bad, st);
    } else {
        // through device staging: PageEntry records are cut down to their vectors on the GPU (the host only moves
        // bytes), a bf16 index gates the f32 input and rounds it into its tiles
        const size_t rec = kind == RowSrc::HostPageEntries ? kPageEntryBytes : EM * sizeof(float);
        const size_t sub_rows = std::min(m, kStageChunk);
        const size_t raw_bytes = kind == RowSrc::HostPageEntries ? (sub_rows * rec + 255) / 256 * 256 : 0;
        const size_t f32_bytes = bf16 ? sub_rows * EM * sizeof(float) : 0;
        DAWN_TRY(ensure_stage(idx, raw_bytes + f32_bytes));
        for (size_t o = 0; o < m; o += sub_rows) {
            const size_t mm = std::min(sub_rows, m - o);
            char* d_raw = reinterpret_cast<char*>(idx->d_stage);
            float* d_f32 = bf16 ? reinterpret_cast<float*>(d_raw + raw_bytes) : reinterpret_cast<float*>(idx->d_x + (at + o) * rb);
            const char* src = reinterpret_cast<const char*>(h_src) + o * rec;
            if (kind == RowSrc::HostPageEntries) {
                DAWN_HIP_TRY(hipMemcpyAsync(d_raw, src, mm * rec, hipMemcpyHostToDevice, st));
                launch_page_entries_to_rows(d_raw, (uint32_t)mm, d_f32, st);
            } else {
                DAWN_HIP_TRY(hipMemcpyAsync(d_f32, src, mm * rec, hipMemcpyHostToDevice, st));
            }
            launch_validate_rows(d_f32, (uint32_t)mm, idx->d_bad, st);  // gate on the f32 input
            if (bf16) launch_rows_f32_to_bf16(d_f32, idx->d_x, at + o, mm, st);
        }
    }
    if (h_ids) DAWN_HIP_TRY(hipMemcpyAsync(idx->d_ids + at, h_ids, m * sizeof(uint64_t), hipMemcpyHostToDevice, st));
    else launch_iota_u64(idx->d_ids + at, first_label, (uint32_t)m, st);
    DAWN_HIP_TRY(hipGetLastError());
    idx->pending += m;
    if (slot >= 0) {
        DAWN_HIP_TRY(hipEventRecord(idx->ev_slot[slot], st));
        idx->ev_slot_used[slot] = true;
    }
    return DAWN_OK;
}

int index_append_wait(dawn_index* idx, int slot) {
    if (idx->shards) return sharded_append_wait(idx, slot);
    if (!idx->ev_slot_used[slot]) return DAWN_OK;
    DAWN_HIP_TRY(hipEventSynchronize(idx->ev_slot[slot]));
    idx->ev_slot_used[slot] = false;
    return DAWN_OK;
}

int index_append_check(dawn_index* idx, uint32_t* bad) {
    *bad = 0;
    if (idx->pending == 0) return DAWN_OK;
    DAWN_TRY(set_device(idx));
    uint32_t* hb = reinterpret_cast<uint32_t*>(idx->h_pinned);
    DAWN_HIP_TRY(hipMemcpyAsync(hb, idx->d_bad, sizeof(uint32_t), hipMemcpyDeviceToHost, idx->stream));
    DAWN_HIP_TRY(hipStreamSynchronize(idx->stream));
    idx->ev_slot_used[0] = idx->ev_slot_used[1] = false;
    *bad = *hb;
    return DAWN_OK;
}

int index_append_finish(dawn_index* idx, bool keep) {
    if (idx->pending == 0) return DAWN_OK;
    DAWN_TRY(set_device(idx));
    if (!keep) {
        // (a bf16 index keeps the rejected rows' fragments: rows >= size are masked by position in every kernel)
        if (idx->dtype != DAWN_DTYPE_BF16)
            (void)hipMemsetAsync(idx->d_x + idx->size * idx->row_bytes(), 0, idx->pending * idx->row_bytes(), idx->stream);
        idx->pending = 0;
        DAWN_HIP_TRY(hipStreamSynchronize(idx->stream));
        return DAWN_OK;
    }
    idx->size += idx->pending;
    idx->pending = 0;
    DAWN_TRY(index_prepare_search(idx, true));
    DAWN_HIP_TRY(hipStreamSynchronize(idx->stream));
    return DAWN_OK;
}

int index_append_commit(dawn_index* idx) {
    if (idx->shards) return sharded_append_commit(idx);
    uint32_t bad = 0;
    const size_t n = idx->pending;
    DAWN_TRY(index_append_check(idx, &bad));
    DAWN_TRY(index_append_finish(idx, bad == 0));
    if (bad) return fail(DAWN_ERR_NOT_NORMALIZED, "Insert embedding is not normalized (%u of %zu rows)", bad, n);
    return DAWN_OK;
}

void index_append_abort(dawn_index* idx) {
    if (idx->shards) return sharded_append_abort(idx);
    (void)hipSetDevice(idx->device);
    (void)hipStreamSynchronize(idx->stream);
    idx->ev_slot_used[0] = idx->ev_slot_used[1] = false;
    (void)index_append_finish(idx, false);
}

int index_clear(dawn_index* idx) {
    if (idx->shards) return sharded_clear(idx);
    DAWN_TRY(set_device(idx));
    DAWN_HIP_TRY(hipDeviceSynchronize());
    idx->size = 0;
    idx->pending = 0;
    idx->shadow_rows = 0;
    idx->i8_rows = 0;
    idx->i6_rows = 0;
    idx->f6_rows = 0;
    idx->fb_rows0 = 0;  // (whatever comes next is a new index: its feedback starts over)
    return DAWN_OK;
}

// Synthetic unit rows appended as pending rows: rows first_row.. of stream `seed`; labels first_id + i, or
// (a shard of a sharded index) the global positions first_pos + i.
int index_fill_async(dawn_index* idx, uint64_t seed, uint64_t first_row, size_t n, uint64_t first_id, bool ids_are_positions,
                     uint64_t first_pos) {
    if (n == 0) return DAWN_OK;
    DAWN_TRY(set_device(idx));
    DAWN_TRY(ensure_room(idx, n));
    const bool bf16 = idx->dtype == DAWN_DTYPE_BF16;
    const size_t rb = idx->row_bytes();
    const size_t chunk = bf16 ? kStageChunk : (size_t)1u << 22;  // rows per generator launch
    const size_t m0 = std::min(n, chunk);
    // staging: [m0] f32 lengths (+ [m0][384] f32 rows for a bf16 index)
    const size_t len_bytes = (m0 * sizeof(float) + 255) / 256 * 256;
    DAWN_TRY(ensure_stage(idx, len_bytes + (bf16 ? m0 * EM * sizeof(float) : 0)));
    float* d_len = idx->d_stage;
    float* d_f32 = reinterpret_cast<float*>(reinterpret_cast<char*>(idx->d_stage) + len_bytes);
    const size_t at = idx->size + idx->pending;
    for (size_t o = 0; o < n; o += chunk) {
        const size_t m = std::min(chunk, n - o);
        char* dst = idx->d_x + (at + o) * rb;
        if (bf16) {  // f32 unit rows of the spec, then rounded: the bf16 index holds round_bf16(spec row)
            launch_fill_synth(seed, first_row + o, (uint32_t)m, d_f32, d_len, idx->synth_dist, idx->stream);
            launch_rows_f32_to_bf16(d_f32, idx->d_x, at + o, m, idx->stream);
        } else {
            launch_fill_synth(seed, first_row + o, (uint32_t)m, reinterpret_cast<float*>(dst), d_len, idx->synth_dist, idx->stream);
        }
        launch_iota_u64(idx->d_ids + at + o, (ids_are_positions ? first_pos : first_id) + o, (uint32_t)m, idx->stream);
    }
    DAWN_HIP_TRY(hipGetLastError());
    if (idx->pending == 0) DAWN_HIP_TRY(hipMemsetAsync(idx->d_bad, 0, sizeof(uint32_t), idx->stream));
    idx->pending += n;
    return DAWN_OK;
}

// Rows come back as f32 whatever the storage type (bf16 rows widened exactly).
int index_get_rows_single(dawn_index* idx, size_t first, size_t n, float* out_rows, uint64_t* out_ids) {
    if (first + n > idx->size) return fail(DAWN_ERR_INVALID_ARG, "rows [%zu, %zu) out of range (size %zu)", first, first + n, idx->size);
    if (n == 0) return DAWN_OK;
    DAWN_TRY(set_device(idx));
    const size_t rb = idx->row_bytes();
    if (out_rows && idx->dtype == DAWN_DTYPE_BF16) {
        DAWN_TRY(ensure_stage(idx, std::min(n, kStageChunk) * EM * sizeof(float)));
        for (size_t o = 0; o < n; o += kStageChunk) {
            const size_t m = std::min(kStageChunk, n - o);
            launch_rows_bf16_to_f32(idx->d_x, first + o, idx->d_stage, m, idx->stream);
            DAWN_HIP_TRY(hipMemcpyAsync(out_rows + o * EM, idx->d_stage, m * EM * sizeof(float), hipMemcpyDeviceToHost,
                                        idx->stream));
            DAWN_HIP_TRY(hipStreamSynchronize(idx->stream));
        }
    } else if (out_rows) {
        DAWN_HIP_TRY(hipMemcpy(out_rows, idx->d_x + first * rb, n * rb, hipMemcpyDeviceToHost));
    }
    if (out_ids) DAWN_HIP_TRY(hipMemcpy(out_ids, idx->d_ids + first, n * sizeof(uint64_t), hipMemcpyDeviceToHost));
    return DAWN_OK;
}

int index_profile_enable_single(dawn_index* idx, int enable) {
    DAWN_TRY(set_device(idx));
    if (enable) {  // the event pairs exist before the first profiled search: a search creates nothing
        idx->events.reserve(kMaxProfile);
        while (idx->events.size() < kMaxProfile) {
            hipEvent_t a, b;
            DAWN_HIP_TRY(hipEventCreate(&a));
            if (hipEventCreate(&b) != hipSuccess) {
                (void)hipEventDestroy(a);
                return fail(DAWN_ERR_HIP, "hipEventCreate failed");
            }
            idx->events.emplace_back(a, b);
        }
    }
    idx->profiling = enable != 0;
    idx->events_used = 0;
    return DAWN_OK;
}

int index_profile_read_single(dawn_index* idx, uint64_t* launches, double* total_ms) {
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

// The certificate counters live on the device (scan_exact_kernel bumps them at the end of every search, whichever entry
// point issued it); reading them synchronises the device.
int index_stats_single(dawn_index* idx, uint64_t* searches, uint64_t* second, uint64_t* fallbacks, uint64_t* deepened,
                       uint64_t* bounded, uint64_t* packed_failures, uint64_t* demoted) {
    DAWN_TRY(set_device(idx));
    uint32_t st[N_STAT_SLOTS] = {};
    DAWN_HIP_TRY(hipDeviceSynchronize());
    if (idx->d_stats) DAWN_HIP_TRY(hipMemcpy(st, idx->d_stats, sizeof(st), hipMemcpyDeviceToHost));
    if (searches) *searches = idx->n_searches;
    if (second) *second = (uint64_t)st[FLAG_SECOND] + st[FLAG_DEEP];
    if (fallbacks) *fallbacks = st[FLAG_FALLBACK];
    if (deepened) *deepened = st[FLAG_DEEP];
    if (bounded) *bounded = st[FLAG_BOUNDED];
    if (packed_failures) *packed_failures = st[STAT_PACKED_FAIL];
    if (demoted) *demoted = idx->n_demoted;
    return DAWN_OK;
}

int index_memory_single(dawn_index* idx, uint64_t* rows_bytes, uint64_t* shadow_bytes, uint64_t* other_bytes) {
    const uint64_t rows = idx->d_x ? (uint64_t)padded_rows(idx->cap_phys) * idx->row_bytes() : 0;
    uint64_t shadows = 0;
    if (idx->d_shadow) shadows += (uint64_t)padded_rows(idx->shadow_cap) * EM * 2;
    if (idx->d_i8) {
        const uint64_t prow = padded_rows(idx->i8_cap) + 128;
        shadows += prow * EM + (prow / 32 + 1) * 8;
    }
    if (idx->d_i6) {
        const uint64_t prow = padded_rows(idx->i6_cap) + 128;
        shadows += prow * idx->i6_row_bytes() + (prow / 32 + 1) * 8;
    }
    if (idx->d_f6) {
        const uint64_t prow = padded_rows(idx->f6_cap) + 128;
        shadows += prow * 288 + (prow / 16 + 1) * 8;
    }
    uint64_t other = (uint64_t)std::max<size_t>(idx->cap_phys, idx->d_ids ? 1 : 0) * sizeof(uint64_t);  // ids
    if (idx->d_cand_s) other += (uint64_t)idx->ws_B * idx->ws_lists * LIST * 8 + 2 * idx->ws_B * 4 + 16;
    if (idx->bws.cand) other += (uint64_t)BATCH_QT * (EM * 2 + 4 + BATCH_CAND_SEGS * 4 + (uint64_t)BATCH_CAP * 8);
    other += idx->stage_bytes;
    other += kMaxBatch * (EM * 4 + DAWN_MAX_K * 12 + 4) + 4;  // host-API staging
    if (idx->f6ws.cand_big)  // the FP6 first filter's workspaces: survivors, their counters, query images, thresholds
        other += (uint64_t)BATCH_QT * BATCH_CAND_SEGS * idx->f6ws.seg_cap_big * 8 + (uint64_t)BATCH_QT * BATCH_CAND_SEGS * 4 + 16 * 3 * 64 * 6 * 4 +
                 (uint64_t)BATCH_QT * 12;
    if (idx->bounded.wide_res) other += (uint64_t)BATCH_QT * BOUNDED_WIDE_CAP * 8 + BATCH_QT * 4;  // the bounded pass's wide form
    if (rows_bytes) *rows_bytes = rows;
    if (shadow_bytes) *shadow_bytes = shadows;
    if (other_bytes) *other_bytes = other;
    return DAWN_OK;
}

int index_set_option_single(dawn_index* idx, const char* name, int64_t value) {
    const std::string n(name);
    DAWN_TRY(set_device(idx));
    auto reprepare = [&]() -> int {  // options that change which shadow / workspace the searches need
        DAWN_TRY(index_prepare_search(idx));
        DAWN_HIP_TRY(hipStreamSynchronize(idx->stream));
        return DAWN_OK;
    };
    if (n == "force_fallback") {  // 1: every query takes the exact pass; 2: every certificate is made to fail, the ladder answers
        if (value < 0 || value > 2) return fail(DAWN_ERR_INVALID_ARG, "force_fallback must be 0, 1 or 2");
        idx->force_fallback = (int)value;
        return DAWN_OK;
    }
    if (n == "ladder_feedback") {  // 0: single queries of a large index always try the packed stream first (A/B, tests);
                                   // 2: never — the bounded pass is their whole search (what a demoted index does)
        if (value < 0 || value > 2) return fail(DAWN_ERR_INVALID_ARG, "ladder_feedback must be 0, 1 or 2");
        idx->ladder_feedback = (int)value;
        idx->fb = dawn_index::LadderFeedback{};
        idx->f6fb = dawn_index::F6Feedback{};
        idx->bfb = dawn_index::BatchFeedback{};
        if (idx->h_stats) idx->fb.win_fail0 = reinterpret_cast<volatile uint32_t*>(idx->h_stats)[STAT_PACKED_FAIL];
        return DAWN_OK;
    }
    if (n == "debug_bad_threshold") {  // test hook: a demoted search starts its bounded pass from an impossible threshold (-1): the
                                       // pass must notice and its last workgroup scan all rows exactly instead
        idx->debug_bad_threshold = value != 0;
        return DAWN_OK;
    }
    if (n == "bounded_pass") {  // 0: a failed certificate goes straight to the exact pass over all rows (round 3; A/B)
        idx->bounded_pass = value != 0;
        return DAWN_OK;
    }
    if (n == "synth_dist") {  // dawn_index_fill_synthetic: 0 the spec's uniform rows, 1 Gaussian, 2 / 3 heavy-tailed, 4 / 5 topical mixture
        if (value < 0 || value > 5) return fail(DAWN_ERR_INVALID_ARG, "synth_dist must be 0..5");
        idx->synth_dist = (int)value;
        return DAWN_OK;
    }
    if (n == "scan_blocks") {
        if (value < 1 || value > 65535) return fail(DAWN_ERR_INVALID_ARG, "scan_blocks out of range");
        idx->geom.blocks = (int)value;
        return reprepare();  // candidate buffers are sized by the grid
    }
    if (n == "mfma_min_batch") {
        if (value < 0) return fail(DAWN_ERR_INVALID_ARG, "mfma_min_batch must be >= 0");
        idx->mfma_min_batch = (int)std::min<int64_t>(value, 1 << 30);
        return DAWN_OK;
    }
    if (n == "f6_shadow") {  // batches of an index of >= f6_min_rows rows filter on the FP6 shadow first (scan_f6.hip): 0 never, 1 whenever
                             // it can be allocated, 2 (default) auto: when it fits with kF6AutoHeadroom of HBM to spare
        if (value < 0 || value > 2) return fail(DAWN_ERR_INVALID_ARG, "f6_shadow must be 0, 1 or 2");
        idx->use_f6 = (int)value;
        if (value) idx->f6_failed = false;
        return reprepare();
    }
    if (n == "f6_min_rows") {
        if (value < 0) return fail(DAWN_ERR_INVALID_ARG, "f6_min_rows must be >= 0");
        idx->f6_min_rows = (size_t)value;
        return reprepare();
    }
    if (n == "f6_stagger") {  // -1: the LDS-staged FP6 pass (default); >= 0: the register-ring pass, its waves this many tiles apart
        if (value < -4 || value > 4096) return fail(DAWN_ERR_INVALID_ARG, "f6_stagger must be -4..4096");
        idx->f6ws.stagger = (int)value;
        return DAWN_OK;
    }
    if (n == "bounded_packed") {  // the bounded pass of a single query streams the packed 5-bit shadow: 0 never, 1 from 2 Mi rows, 2 always
        if (value < 0 || value > 2) return fail(DAWN_ERR_INVALID_ARG, "bounded_packed must be 0, 1 or 2");
        idx->bounded_packed = (int)value;
        return DAWN_OK;
    }
    if (n == "bounded_seed_shift") {  // the seed searches the first n >> shift rows (default 5: 1/32)
        if (value < 2 || value > 8) return fail(DAWN_ERR_INVALID_ARG, "bounded_seed_shift must be 2..8");
        idx->bounded_seed_shift = (int)value;
        return DAWN_OK;
    }
    if (n == "bounded_seed") {  // 1 (default): a demoted query's bounded pass on the packed shadow starts from the k-th distance of a
                                // packed-stream search over the first 1/32 of the rows
        if (value < 0 || value > 2) return fail(DAWN_ERR_INVALID_ARG, "bounded_seed must be 0, 1 or 2");
        idx->bounded_seed = (int)value;
        return DAWN_OK;
    }
    if (n == "batch_rerun") {  // the second pass of a batch's flagged queries with exact-derived thresholds: 0 never (default), 1 on an
                               // index whose batch feedback has deepened its thresholds, 2 every batch (tests, A/B)
        if (value < 0 || value > 2) return fail(DAWN_ERR_INVALID_ARG, "batch_rerun must be 0, 1 or 2");
        idx->batch_rerun = (int)value;
        return DAWN_OK;
    }
    if (n == "bounded_multi_packed") {  // the 16-query batch form of the bounded pass streams the packed 5-bit shadow too (0 / 1)
        if (value != 0 && value != 1) return fail(DAWN_ERR_INVALID_ARG, "bounded_multi_packed must be 0 or 1");
        idx->bounded.multi_packed = (int)value;
        return DAWN_OK;
    }
    if (n == "bounded_multi_waves") {  // waves per workgroup of the 16-query batch form (4 or 8)
        if (value != 4 && value != 8) return fail(DAWN_ERR_INVALID_ARG, "bounded_multi_waves must be 4 or 8");
        idx->bounded.multi_waves = (int)value;
        return DAWN_OK;
    }
    if (n == "bounded_ring") {  // 16-B fragments a wave of the bounded pass keeps in flight (6 or 12)
        if (value != 6 && value != 12) return fail(DAWN_ERR_INVALID_ARG, "bounded_ring must be 6 or 12");
        idx->bounded.ring = (int)value;
        return DAWN_OK;
    }
    if (n == "bounded_wide") {  // a batch's flagged queries go through the wide form (64 per stream of the int8 shadow) first (0 / 1)
        if (value != 0 && value != 1) return fail(DAWN_ERR_INVALID_ARG, "bounded_wide must be 0 or 1");
        idx->bounded.wide = (int)value;
        return DAWN_OK;
    }
    if (n == "f6_refine_rows") {  // f32 index: re-score the FP6 survivors on the rows themselves (1, default) or on the int8 shadow (0)
        if (value != 0 && value != 1) return fail(DAWN_ERR_INVALID_ARG, "f6_refine_rows must be 0 or 1");
        idx->f6ws.refine_rows = (int)value;
        return DAWN_OK;
    }
    if (n == "f6_target") {  // survivors per query the FP6 threshold aims for (twice that for count > 32)
        if (value < 256 || value > 24576) return fail(DAWN_ERR_INVALID_ARG, "f6_target must be 256..24576");
        idx->f6ws.target = (int)value;
        return DAWN_OK;
    }
    if (n == "i6_dyn_chunk") {  // sub-tiles per chunk of the packed stream's dynamically assigned tail (default 16)
        if (value < 1 || value > 256) return fail(DAWN_ERR_INVALID_ARG, "i6_dyn_chunk must be 1..256");
        idx->geom_i6.chunk = idx->geom_i6_small.chunk = (int)value;
        return DAWN_OK;
    }
    if (n == "i6_dyn_share") {  // sixteenths of the index the packed stream hands out dynamically (default 2)
        if (value < 1 || value > 15) return fail(DAWN_ERR_INVALID_ARG, "i6_dyn_share must be 1..15");
        idx->geom_i6.dyn_share = idx->geom_i6_small.dyn_share = (int)value;
        return DAWN_OK;
    }
    if (n == "zero_copy_batch") {  // host API: batches of up to this many queries have their results stored straight into pinned host memory
        if (value < 0 || value > (long long)dawn::kMaxBatch) return fail(DAWN_ERR_INVALID_ARG, "zero_copy_batch must be 0..%zu", dawn::kMaxBatch);
        idx->zero_copy_batch = (size_t)value;
        return DAWN_OK;
    }
    if (n == "i6_slack_model") {  // 1 (default): the packed stream's lists are sized from the shadow's measured error bounds; 0: from constants
        if (value != 0 && value != 1) return fail(DAWN_ERR_INVALID_ARG, "i6_slack_model must be 0 or 1");
        idx->i6_slack_model = (int)value;
        idx->i6_slack_dirty = true;
        return index_prepare_search(idx);
    }
    if (n == "i6_central_tail") {  // 1: the packed stream's workgroups do not rescore their own 64 rows exactly; one merge_rescore_kernel
                                   // rescores the index's 64 best by the refined score (deeper rounds, second chance behind it)
        idx->i6_central_tail = value != 0;
        return DAWN_OK;
    }
    if (n == "stream_dynamic_tail") {  // 0: the single-query streams assign every sub-tile statically (A/B of the dynamic tails)
        idx->stream_dyn_tail = value != 0;
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
        return reprepare();
    }
    if (n == "i8_shadow") {
        idx->use_i8 = value != 0;
        if (value) idx->i8_failed = false;
        return reprepare();
    }
    if (n == "i6_shadow") {  // single queries of a large index stream the 6-bit shadow (scan_i6.hip); 0: the int8 shadow
        idx->use_i6 = value != 0;
        if (value) idx->i6_failed = false;
        return reprepare();  // (0: the shadow's memory goes back — i6_shadow_sync)
    }
    if (n == "i6_bits") {  // 5 (240 B/row, default) or 6 (288 B/row) bits per component of the packed shadow; rebuilt
        if (value != 5 && value != 6) return fail(DAWN_ERR_INVALID_ARG, "i6_bits must be 5 or 6");
        if ((int)value != idx->i6_bits) {
            i6_release(idx);
            idx->i6_bits = (int)value;
            idx->i6_failed = false;
        }
        return reprepare();
    }
    if (n == "i6_min_rows") {  // indexes of at least this many rows keep the 6-bit shadow (default 2 Mi; tests: 0)
        if (value < 0) return fail(DAWN_ERR_INVALID_ARG, "i6_min_rows must be >= 0");
        idx->i6_min_rows = (size_t)value;
        return reprepare();
    }
    if (n == "i6_refine") {  // entries of its coarse list a wave of the packed stream refines: 1..64; 0 = from N and k (default);
                             // -1 (tests): full lists, NOT refined — they keep the packed shadow's own bounds
        if (value < -1 || value > 64) return fail(DAWN_ERR_INVALID_ARG, "i6_refine must be -1..64");
        idx->geom_i6.refine = idx->geom_i6_small.refine = (int)value;
        return DAWN_OK;
    }
    if (n == "i6_scan_threads" || n == "i6_scan_ring" || n == "i6_scan_blocks") {
        if (n == "i6_scan_threads") {
            if (value < 64 || value > 512 || value % 64) return fail(DAWN_ERR_INVALID_ARG, "i6_scan_threads must be 64..512, a multiple of 64");
            idx->geom_i6.threads = (int)value;
        } else if (n == "i6_scan_ring") {
            if (value != 12 && value != 8 && value != 6 && value != 4 && value != 3 && value != 2)
                return fail(DAWN_ERR_INVALID_ARG, "i6_scan_ring must be 12, 8, 6, 4, 3 or 2");
            idx->geom_i6.unroll = (int)value;
        } else {
            if (value < 1 || value > 65535) return fail(DAWN_ERR_INVALID_ARG, "i6_scan_blocks out of range");
            idx->geom_i6.blocks = (int)value;
        }
        idx->geom_i6_pinned = true;
        return reprepare();
    }
    if (n == "i8_batched") {
        idx->i8_batched = value != 0;
        return reprepare();
    }
    if (n == "debug_i8_levels") {  // experiment hook: quantise this index's int8 shadow to +-value levels (127 = normal)
        if (value < 3 || value > 127) return fail(DAWN_ERR_INVALID_ARG, "debug_i8_levels must be 3..127");
        DAWN_HIP_TRY(hipDeviceSynchronize());
        idx->i8_levels = (float)value;
        idx->i8_rows = 0;  // re-quantise everything
        return reprepare();
    }
    if (n == "debug_fail_alloc") {
        // test hook for the out-of-memory order (int8 shadow -> f16 shadow -> the f32 rows themselves): bit 0 makes the
        // next int8-shadow allocation fail, bit 1 the next f16-shadow allocation; the shadows held now are dropped so that
        // the allocation is attempted again.  0 restores normal behaviour (and retries).
        // bit 3: the FP6 shadow's allocation
        if (value < 0 || value > 15) return fail(DAWN_ERR_INVALID_ARG, "debug_fail_alloc is a 4-bit mask");
        DAWN_HIP_TRY(hipDeviceSynchronize());
        f6_release(idx);
        idx->f6_failed = false;
        void* drop[] = {idx->d_i8, idx->d_i8meta, idx->d_shadow, idx->d_i6, idx->d_i6meta};
        for (void* p : drop)
            if (p) (void)hipFree(p);
        idx->d_i8 = nullptr;
        idx->d_i8meta = nullptr;
        idx->d_shadow = nullptr;
        idx->d_i6 = nullptr;
        idx->d_i6meta = nullptr;
        idx->i8_cap = idx->i8_rows = idx->shadow_cap = idx->shadow_rows = idx->i6_cap = idx->i6_rows = 0;
        idx->i8_failed = idx->shadow_failed = idx->i6_failed = false;
        idx->debug_fail_alloc = (int)value;
        return reprepare();
    }
    if (n == "f16_shadow_b1") {
        idx->shadow_small_batches = value != 0;
        return reprepare();
    }
    if (n == "shadow_scan_blocks" || n == "shadow_scan_threads" || n == "shadow_scan_unroll") {
        if (n == "shadow_scan_blocks") {
            if (value < 1 || value > 65535) return fail(DAWN_ERR_INVALID_ARG, "shadow_scan_blocks out of range");
            idx->geom_h.blocks = (int)value;
        } else if (n == "shadow_scan_threads") {
            if (value != 64 && value != 128 && value != 256 && value != 512)
                return fail(DAWN_ERR_INVALID_ARG, "shadow_scan_threads must be 64/128/256/512");
            idx->geom_h.threads = (int)value;
        } else {
            if (value < 1 || value > 10) return fail(DAWN_ERR_INVALID_ARG, "shadow_scan_unroll must be 1..10");
            idx->geom_h.unroll = (int)value;
        }
        idx->geom_h_pinned = true;
        return reprepare();
    }
    if (n == "mfma_sched") {
#ifdef DAWN_EXPERIMENTS
        const bool ok = value == 0 || value == 1 || value == 2 || value == 4 || value == 5 || value == 32 || (value >= 41 && value <= 55);
#else
        const bool ok = value == 0 || value == 1 || value == 4 || value == 5 || value == 32;
#endif
        if (!ok) return fail(DAWN_ERR_INVALID_ARG, "mfma_sched must be 0, 1, 4, 5 or 32");
#ifdef DAWN_EXPERIMENTS
        if (value == 2 && !idx->bws.diag) {
            DAWN_HIP_TRY(hipMalloc((void**)&idx->bws.diag, 4096 * 8 * 8 * sizeof(unsigned long long)));
            DAWN_HIP_TRY(hipMemset(idx->bws.diag, 0, 4096 * 8 * 8 * sizeof(unsigned long long)));
        }
#endif
        idx->bws.sched = (int)value;
        return reprepare();
    }
    if (n == "mfma_target") {
        if (value < 64 || value > 4096) return fail(DAWN_ERR_INVALID_ARG, "mfma_target must be 64..4096");
        idx->bws.target = (int)value;
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

}  // namespace dawn

// ---------------------------------------------------------------------------------------------------------------------
// bulk file I/O: parallel pread / pwrite of large extents (the page cache copies at a few GB/s per thread)
// ---------------------------------------------------------------------------------------------------------------------
namespace {

bool rw_all(int fd, char* buf, size_t bytes, off_t off, bool write) {
    while (bytes) {
        const ssize_t r = write ? ::pwrite(fd, buf, bytes, off) : ::pread(fd, buf, bytes, off);
        if (r < 0 && errno == EINTR) continue;
        if (r <= 0) return false;  // error, or a file shorter than its header promised
        buf += r;
        off += r;
        bytes -= (size_t)r;
    }
    return true;
}

bool parallel_rw(int fd, void* buf, size_t bytes, off_t off, bool write) {
    unsigned T = std::thread::hardware_concurrency();
    T = std::max(1u, std::min(T ? T : 1u, 8u));
    if (bytes < ((size_t)8 << 20)) T = 1;
    if (T == 1) return rw_all(fd, (char*)buf, bytes, off, write);
    const size_t per = ((bytes + T - 1) / T + 4095) / 4096 * 4096;
    std::vector<std::thread> th;
    std::vector<char> ok(T, 1);
    for (unsigned t = 0; t < T; ++t) {
        const size_t b0 = std::min(bytes, (size_t)t * per), b1 = std::min(bytes, b0 + per);
        if (b0 == b1) break;
        th.emplace_back([=, &ok]() { ok[t] = rw_all(fd, (char*)buf + b0, b1 - b0, off + (off_t)b0, write) ? 1 : 0; });
    }
    for (auto& x : th) x.join();
    for (char c : ok)
        if (!c) return false;
    return true;
}

struct PinnedPair {
    void* buf[2] = {nullptr, nullptr};
    int init(size_t bytes) {
        for (auto& b : buf)
            if (hipHostMalloc(&b, bytes, hipHostMallocDefault) != hipSuccess) {
                b = nullptr;
                return fail(DAWN_ERR_OOM, "hipHostMalloc(%zu) for the bulk staging buffers failed", bytes);
            }
        return DAWN_OK;
    }
    ~PinnedPair() {
        for (void* b : buf)
            if (b) (void)hipHostFree(b);
    }
};

// n_rows records of `kind` at rows_off of fd (+ their labels at ids_off, or first_label + i) -> pending rows of idx,
// through two pinned buffers: the parallel read of chunk c+1 overlaps the DMA (and the GPU-side unpacking /
// validation) of chunk c.  The caller commits or aborts.
int ingest_file_rows(dawn_index* idx, int fd, dawn::RowSrc kind, size_t n_rows, off_t rows_off, off_t ids_off,
                     uint64_t first_label, const char* path) {
    const size_t rec = kind == dawn::RowSrc::HostPageEntries ? kPageEntryBytes : dawn::EM * sizeof(float);
    const size_t ch = std::min(kBulkChunkRows, std::max<size_t>(n_rows, 1));
    const size_t rows_bytes = (ch * rec + 255) / 256 * 256;
    PinnedPair pp;
    DAWN_TRY(pp.init(rows_bytes + (ids_off >= 0 ? ch * sizeof(uint64_t) : 0)));
    auto run = [&]() -> int {
        size_t c = 0;
        for (size_t o = 0; o < n_rows; o += ch, ++c) {
            const int slot = (int)(c & 1);
            const size_t m = std::min(ch, n_rows - o);
            DAWN_TRY(dawn::index_append_wait(idx, slot));  // the copies out of this buffer two chunks ago
            char* hb = reinterpret_cast<char*>(pp.buf[slot]);
            if (!parallel_rw(fd, hb, m * rec, rows_off + (off_t)(o * rec), false))
                return fail(DAWN_ERR_IO, "%s: truncated row data", path);
            uint64_t* hid = nullptr;
            if (ids_off >= 0) {
                hid = reinterpret_cast<uint64_t*>(hb + rows_bytes);
                if (!rw_all(fd, (char*)hid, m * 8, ids_off + (off_t)(o * 8), false))
                    return fail(DAWN_ERR_IO, "%s: truncated id table", path);
            }
            DAWN_TRY(dawn::index_append_async(idx, kind, hb, hid, first_label + o, m, slot));
        }
        return DAWN_OK;
    };
    const int rc = run();
    // the pinned buffers go away with this frame: every copy out of them must have finished, error or not
    const std::string msg = dawn::last_error();
    (void)dawn::index_append_wait(idx, 0);
    (void)dawn::index_append_wait(idx, 1);
    if (rc != DAWN_OK) dawn::last_error() = msg;
    return rc;
}

int add_batch_now(dawn_index* idx, size_t n, const uint64_t* ids, const float* v);
}  // namespace

int dawn::index_flush_adds(dawn_index* idx) {
    if (idx->staged == 0) return DAWN_OK;
    const size_t n = idx->staged;
    idx->staged = 0;  // (whatever happens: the rows are either in or dropped with the error)
    return add_batch_now(idx, n, idx->h_add_ids, idx->h_add_rows);
}

namespace {
int add_batch_now(dawn_index* idx, size_t n, const uint64_t* ids, const float* v) {
    // (pageable caller buffers: the runtime stages them; every call ends synchronised in commit)
    for (size_t o = 0; o < n; o += dawn::kStageChunk) {
        const size_t m = std::min(dawn::kStageChunk, n - o);
        int rc = dawn::index_append_async(idx, dawn::RowSrc::HostRows, v + o * dawn::EM, ids + o, 0, m, -1);
        if (rc != DAWN_OK) {
            dawn::index_append_abort(idx);
            return rc;
        }
    }
    return dawn::index_append_commit(idx);
}

// The staged single-row adds join the index (index_internal.hpp: h_add_rows).  Called in front of everything that looks at
// the rows; an error (growth failed: out of HBM) belongs to the adds and is reported by the call that flushes them.
int flush_adds(dawn_index* idx) { return dawn::index_flush_adds(idx); }

int root_device(const dawn_index* idx) { return idx->shards ? dawn::sharded_root_device(idx) : idx->device; }
int index_dtype(const dawn_index* idx) { return idx->shards ? dawn::sharded_dtype(idx) : idx->dtype; }

}  // namespace

extern "C" {

int dawn_index_create(size_t dims, int dtype, int device, dawn_index** out) {
    if (!out) return fail(DAWN_ERR_INVALID_ARG, "out is NULL");
    *out = nullptr;
    if (dims != DAWN_EM_LEN) return fail(DAWN_ERR_UNSUPPORTED, "dims must be %d (EM_LEN)", DAWN_EM_LEN);
    if (dtype != DAWN_DTYPE_F32 && dtype != DAWN_DTYPE_BF16)
        return fail(DAWN_ERR_UNSUPPORTED, "dtype %d not supported", dtype);
    return dawn::guarded([&] { return dawn::index_create_single(dtype, device, out); });
}

void dawn_index_destroy(dawn_index* idx) {
    if (!idx) return;
    if (idx->h_add_rows) (void)hipHostFree(idx->h_add_rows);  // (staged adds that were never looked at go with the index)
    idx->h_add_rows = nullptr;
    if (idx->shards) return dawn::sharded_destroy(idx);
    dawn::index_destroy_single(idx);
}

int dawn_index_reserve(dawn_index* idx, size_t capacity) {
    if (!idx) return fail(DAWN_ERR_INVALID_ARG, "idx is NULL");
    return dawn::guarded([&] {
        DAWN_TRY(flush_adds(idx));
        return idx->shards ? dawn::sharded_reserve(idx, capacity) : dawn::index_reserve_single(idx, capacity);
    });
}

size_t dawn_index_size(const dawn_index* idx) {
    return !idx ? 0 : (idx->shards ? dawn::sharded_size(idx) : idx->size) + idx->staged;
}
size_t dawn_index_capacity(const dawn_index* idx) {
    return !idx ? 0 : idx->shards ? dawn::sharded_capacity(idx) : idx->cap_reported;
}

int dawn_index_add_batch(dawn_index* idx, size_t n, const uint64_t* ids, const float* v) {
    if (!idx || (!ids && n) || (!v && n)) return fail(DAWN_ERR_INVALID_ARG, "NULL argument");
    if (n == 0) return DAWN_OK;
    return dawn::guarded([&] {
        DAWN_TRY(flush_adds(idx));  // (insertion order)
        return add_batch_now(idx, n, ids, v);
    });
}

// One row: gated on the host (vector.rs:185-192, as the reference does right before index.add, search_provider.rs:265-267),
// then staged; see index_internal.hpp.  ~0.3 us per call instead of a transfer + two synchronisations.
int dawn_index_add(dawn_index* idx, uint64_t id, const float* v) {
    if (!idx || !v) return fail(DAWN_ERR_INVALID_ARG, "NULL argument");
    if (!dawn::host_is_normalized(v)) return fail(DAWN_ERR_NOT_NORMALIZED, "Insert embedding is not normalized");
    return dawn::guarded([&] {
        if (!idx->h_add_rows) {
            DAWN_HIP_TRY(hipSetDevice(root_device(idx)));
            void* hp = nullptr;
            DAWN_HIP_TRY(hipHostMalloc(&hp, dawn::kAddStageRows * (dawn::EM * sizeof(float) + sizeof(uint64_t)), hipHostMallocDefault));
            idx->h_add_rows = reinterpret_cast<float*>(hp);
            idx->h_add_ids = reinterpret_cast<uint64_t*>(idx->h_add_rows + dawn::kAddStageRows * dawn::EM);
        }
        if (idx->staged == dawn::kAddStageRows) DAWN_TRY(flush_adds(idx));
        std::memcpy(idx->h_add_rows + idx->staged * dawn::EM, v, dawn::EM * sizeof(float));
        idx->h_add_ids[idx->staged] = id;
        ++idx->staged;
        return DAWN_OK;
    });
}

int dawn_index_search_device(dawn_index* idx, const float* d_queries, size_t B, size_t count, uint64_t* d_labels,
                             float* d_distances, uint32_t* d_found, void* stream) {
    if (!idx || !d_queries || !d_labels || !d_distances || !d_found) return fail(DAWN_ERR_INVALID_ARG, "NULL argument");
    if (count == 0 || count > DAWN_MAX_K) return fail(DAWN_ERR_UNSUPPORTED, "count must be 1..%d", DAWN_MAX_K);
    if (B == 0) return DAWN_OK;
    if (idx->staged) {  // (rows added one by one since the last call: they join the index now — this one search synchronises)
        const int rc = dawn::guarded([&] { return flush_adds(idx); });
        if (rc != DAWN_OK) return rc;
    }
    if (idx->shards)
        return dawn::guarded(
            [&] { return dawn::sharded_search_device(idx, d_queries, B, count, d_labels, d_distances, d_found, (hipStream_t)stream); });
    DAWN_TRY(set_device(idx));
    idx->n_searches += B;
    return dawn::index_search_on_device(idx, d_queries, B, count, d_labels, d_distances, d_found, (hipStream_t)stream);
}

int dawn_index_search_batch(dawn_index* idx, const float* queries, size_t B, size_t count, uint64_t* labels,
                            float* distances, size_t* found) {
    if (!idx || !queries || !labels || !distances || !found) return fail(DAWN_ERR_INVALID_ARG, "NULL argument");
    if (count == 0 || count > DAWN_MAX_K) return fail(DAWN_ERR_UNSUPPORTED, "count must be 1..%d", DAWN_MAX_K);
    if (dawn::host_first_not_normalized(queries, B) != B)  // search_provider.rs:206-208
        return fail(DAWN_ERR_NOT_NORMALIZED, "Search vector is not normalized");
    if (idx->staged) {
        const int rc = dawn::guarded([&] { return flush_adds(idx); });
        if (rc != DAWN_OK) return rc;
    }
    if (idx->shards)
        return dawn::guarded([&] { return dawn::sharded_search_batch(idx, queries, B, count, labels, distances, found); });
    DAWN_TRY(set_device(idx));
    using dawn::kMaxBatch;
    for (size_t b0 = 0; b0 < B; b0 += kMaxBatch) {
        const size_t nb = std::min(kMaxBatch, B - b0);
        char* hp = (char*)idx->h_pinned;
        float* hq = (float*)hp;
        uint64_t* hl = (uint64_t*)(hp + kMaxBatch * dawn::EM * 4);
        float* hd = (float*)(hp + kMaxBatch * (dawn::EM * 4 + DAWN_MAX_K * 8));
        uint32_t* hf = (uint32_t*)(hp + kMaxBatch * (dawn::EM * 4 + DAWN_MAX_K * 8 + DAWN_MAX_K * 4));
        std::memcpy(hq, queries + b0 * dawn::EM, nb * dawn::EM * sizeof(float));
        DAWN_HIP_TRY(hipMemcpyAsync(idx->d_q, hq, nb * dawn::EM * sizeof(float), hipMemcpyHostToDevice, idx->stream));
        idx->n_searches += nb;
        if (nb <= idx->zero_copy_batch) {
            // few queries: the tail kernels store the results straight into the pinned host block (coherent,
            // device-visible): no copy commands between the last kernel and the host's wake-up
            DAWN_TRY(dawn::index_search_on_device(idx, idx->d_q, nb, count, hl, hd, hf, idx->stream));
        } else {
            DAWN_TRY(dawn::index_search_on_device(idx, idx->d_q, nb, count, idx->d_labels, idx->d_dist, idx->d_found, idx->stream));
            DAWN_HIP_TRY(hipMemcpyAsync(hl, idx->d_labels, nb * count * sizeof(uint64_t), hipMemcpyDeviceToHost, idx->stream));
            DAWN_HIP_TRY(hipMemcpyAsync(hd, idx->d_dist, nb * count * sizeof(float), hipMemcpyDeviceToHost, idx->stream));
            DAWN_HIP_TRY(hipMemcpyAsync(hf, idx->d_found, nb * sizeof(uint32_t), hipMemcpyDeviceToHost, idx->stream));
        }
        DAWN_HIP_TRY(hipStreamSynchronize(idx->stream));
        std::memcpy(labels + b0 * count, hl, nb * count * sizeof(uint64_t));
        std::memcpy(distances + b0 * count, hd, nb * count * sizeof(float));
        for (size_t b = 0; b < nb; ++b) found[b0 + b] = hf[b];
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
    if (G == 0 || count == 0 || count > DAWN_MAX_K || G > dawn::kMaxMergeCands / DAWN_MAX_K)
        return fail(DAWN_ERR_UNSUPPORTED, "G must be 1..64 and count 1..%d", DAWN_MAX_K);
    DAWN_TRY(dawn::require_device(device));
    DAWN_HIP_TRY(hipSetDevice(device));
    if (B == 0) return DAWN_OK;
    dawn::launch_shard_merge(G, B, count, d_in_labels, d_in_distances, d_in_found, B * count, B * count, B, nullptr, d_labels,
                             d_distances, d_found, (hipStream_t)stream);
    DAWN_HIP_TRY(hipGetLastError());
    return DAWN_OK;
}

size_t dawn_result_blob_bytes(size_t B, size_t count) { return (B * count * 12 + B * 4 + 15) / 16 * 16; }

int dawn_topk_merge_packed_device(int device, size_t G, size_t B, size_t count, const void* d_blobs,
                                  uint64_t* d_labels, float* d_distances, uint32_t* d_found, void* stream) {
    if (!d_blobs || !d_labels || !d_distances || !d_found) return fail(DAWN_ERR_INVALID_ARG, "NULL argument");
    if (G == 0 || count == 0 || count > DAWN_MAX_K || G > dawn::kMaxMergeCands / DAWN_MAX_K)
        return fail(DAWN_ERR_UNSUPPORTED, "G must be 1..64 and count 1..%d", DAWN_MAX_K);
    DAWN_TRY(dawn::require_device(device));
    DAWN_HIP_TRY(hipSetDevice(device));
    if (B == 0) return DAWN_OK;
    const size_t stride = dawn_result_blob_bytes(B, count);
    const char* base = (const char*)d_blobs;
    dawn::launch_shard_merge(G, B, count, (const uint64_t*)base, (const float*)(base + B * count * 8),
                             (const uint32_t*)(base + B * count * 12), stride / 8, stride / 4, stride / 4, nullptr, d_labels,
                             d_distances, d_found, (hipStream_t)stream);
    DAWN_HIP_TRY(hipGetLastError());
    return DAWN_OK;
}

int dawn_index_fill_synthetic(dawn_index* idx, uint64_t seed, uint64_t first_row, size_t n, uint64_t first_id) {
    if (!idx) return fail(DAWN_ERR_INVALID_ARG, "idx is NULL");
    if (n == 0) return DAWN_OK;
    return dawn::guarded([&] {
        DAWN_TRY(flush_adds(idx));
        if (idx->shards) return dawn::sharded_fill_synthetic(idx, seed, first_row, n, first_id);
        int rc = dawn::index_fill_async(idx, seed, first_row, n, first_id, false, 0);
        if (rc != DAWN_OK) {
            dawn::index_append_abort(idx);
            return rc;
        }
        return dawn::index_append_finish(idx, true);  // (generated rows are unit rows by construction: no gate)
    });
}

int dawn_index_get_rows(dawn_index* idx, size_t first, size_t n, float* out_rows, uint64_t* out_ids) {
    if (!idx) return fail(DAWN_ERR_INVALID_ARG, "idx is NULL");
    return dawn::guarded([&] {
        DAWN_TRY(flush_adds(idx));
        return idx->shards ? dawn::sharded_get_rows(idx, first, n, out_rows, out_ids)
                           : dawn::index_get_rows_single(idx, first, n, out_rows, out_ids);
    });
}

// File layout: "DAWNIDX1" | u32 dims | u32 dtype | u64 n | ids[n] u64 | rows[n][384] f32 (little endian).  Rows are
// written as f32 for both storage types (a bf16 index widens exactly and re-rounds to the same bits on load); a sharded
// index writes its rows in insertion order: the file does not depend on how many devices hold the index.
// Written to `path`.tmp, flushed to disk and renamed over `path`: an interrupted save (the reference saves on shutdown,
// src/bin/dawnsearch.rs:151) leaves the previous file, never a truncated one.
int dawn_index_save(dawn_index* idx, const char* path) {
    if (!idx || !path) return fail(DAWN_ERR_INVALID_ARG, "NULL argument");
    return dawn::guarded([&] {
        DAWN_HIP_TRY(hipSetDevice(root_device(idx)));
        DAWN_TRY(flush_adds(idx));
        const std::string tmp = std::string(path) + ".tmp";
        const int fd = ::open(tmp.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0644);
        if (fd < 0) return fail(DAWN_ERR_IO, "cannot open %s for writing", tmp.c_str());
        const uint64_t n = dawn_index_size(idx);
        char header[24];
        const uint32_t dims = DAWN_EM_LEN, dtype = (uint32_t)index_dtype(idx);
        std::memcpy(header, kMagic, 8);
        std::memcpy(header + 8, &dims, 4);
        std::memcpy(header + 12, &dtype, 4);
        std::memcpy(header + 16, &n, 8);
        bool ok = rw_all(fd, header, 24, 0, true);
        const size_t ch = std::min<size_t>(kBulkChunkRows, std::max<uint64_t>(n, 1));
        PinnedPair pp;
        int rc = ok ? pp.init(ch * dawn::EM * sizeof(float)) : DAWN_OK;
        const off_t rows_off = 24 + (off_t)(n * 8);
        for (size_t o = 0; ok && rc == DAWN_OK && o < n; o += ch) {
            const size_t m = std::min<size_t>(ch, n - o);
            // D2H into pinned memory runs at the link rate; the parallel pwrite behind it is the slower half
            rc = dawn_index_get_rows(idx, o, m, reinterpret_cast<float*>(pp.buf[0]), reinterpret_cast<uint64_t*>(pp.buf[1]));
            if (rc != DAWN_OK) break;
            ok = rw_all(fd, (char*)pp.buf[1], m * 8, 24 + (off_t)(o * 8), true) &&
                 parallel_rw(fd, pp.buf[0], m * dawn::EM * sizeof(float), rows_off + (off_t)(o * dawn::EM * sizeof(float)), true);
        }
        if (ok && rc == DAWN_OK && ::fsync(fd) != 0) ok = false;
        if (::close(fd) != 0) ok = false;
        if (rc == DAWN_OK && ok && ::rename(tmp.c_str(), path) != 0) ok = false;
        if (rc != DAWN_OK || !ok) {
            ::unlink(tmp.c_str());
            return rc != DAWN_OK ? rc : fail(DAWN_ERR_IO, "writing %s failed", path);
        }
        return DAWN_OK;
    });
}

// index.load(path): replaces the contents.  All or nothing: the header is checked against the file's size before the
// index is touched, and any later failure (I/O error, a row that fails the is_normalized gate) leaves the index EMPTY,
// never partially filled — the reference's start-up flow then rebuilds from the database
// (`if !load(path).is_ok() { fill_index_from_db(); save() }`, search_provider.rs:115-117) onto a clean index.
int dawn_index_load(dawn_index* idx, const char* path) {
    if (!idx || !path) return fail(DAWN_ERR_INVALID_ARG, "NULL argument");
    return dawn::guarded([&] {
        DAWN_HIP_TRY(hipSetDevice(root_device(idx)));
        idx->staged = 0;  // (load replaces the contents, staged single-row adds included)
        // EVERY failure below leaves the index empty (usearch's load resets the index before it reads): the reference's
        // `if !load(path).is_ok() { fill_index_from_db() }` (search_provider.rs:115-117) must never append to old rows
        auto failed_empty = [&](int rc) {
            const std::string msg = dawn::last_error();
            (void)dawn::index_clear(idx);
            dawn::last_error() = msg;
            return rc;
        };
        const int fd = ::open(path, O_RDONLY);
        if (fd < 0) return failed_empty(fail(DAWN_ERR_IO, "cannot open %s", path));
        struct stat stt;
        char header[24];
        uint32_t dims = 0, dtype = 0;
        uint64_t n = 0;
        int rc = DAWN_OK;
        if (::fstat(fd, &stt) != 0 || !rw_all(fd, header, 24, 0, false) || std::memcmp(header, kMagic, 8) != 0) {
            rc = fail(DAWN_ERR_IO, "%s is not a dawn index file", path);
        } else {
            std::memcpy(&dims, header + 8, 4);
            std::memcpy(&dtype, header + 12, 4);
            std::memcpy(&n, header + 16, 8);
            if (dims != DAWN_EM_LEN || (dtype != DAWN_DTYPE_F32 && dtype != DAWN_DTYPE_BF16))
                rc = fail(DAWN_ERR_IO, "%s is not a dawn index file", path);
            else if (n > ((uint64_t)stt.st_size - 24) / (8 + dawn::EM * sizeof(float)))
                rc = fail(DAWN_ERR_IO, "%s: truncated (header promises %llu rows, the file holds %llu bytes)", path,
                          (unsigned long long)n, (unsigned long long)stt.st_size);
        }
        if (rc != DAWN_OK) {
            ::close(fd);
            return failed_empty(rc);
        }
        rc = dawn::index_clear(idx);  // load replaces the contents (usearch load semantics)
        if (rc == DAWN_OK && n) {
            rc = dawn_index_reserve(idx, n);
            if (rc == DAWN_OK)
                rc = ingest_file_rows(idx, fd, dawn::RowSrc::HostRows, n, 24 + (off_t)(n * 8), 24, 0, path);
            if (rc == DAWN_OK) rc = dawn::index_append_commit(idx);
            else dawn::index_append_abort(idx);
        }
        ::close(fd);
        return rc == DAWN_OK ? rc : failed_empty(rc);
    });
}

// src/index/warc.rs:35-43 PageEntry (repr(C)): u64 url_pos, u64 title_pos, f32 vector[384], u64 url_len,
// u64 title_len = 1568 bytes; read as examples_old/document_embeddings.rs:60-71 does (entries = len / 1568).  Appends;
// on any failure nothing is added.  The records go to the GPU as they are on disk (pinned staging, DMA overlapped with
// the next read) and are cut down to their vectors there.
int dawn_index_load_page_entries(dawn_index* idx, const char* emb_path, uint64_t first_id) {
    if (!idx || !emb_path) return fail(DAWN_ERR_INVALID_ARG, "NULL argument");
    return dawn::guarded([&] {
        DAWN_HIP_TRY(hipSetDevice(root_device(idx)));
        DAWN_TRY(flush_adds(idx));
        const int fd = ::open(emb_path, O_RDONLY);
        if (fd < 0) return fail(DAWN_ERR_IO, "cannot open %s", emb_path);
        struct stat stt;
        int rc = DAWN_OK;
        if (::fstat(fd, &stt) != 0) rc = fail(DAWN_ERR_IO, "cannot stat %s", emb_path);
        const size_t n = rc == DAWN_OK ? (size_t)stt.st_size / kPageEntryBytes : 0;
        if (rc == DAWN_OK && n) {
            rc = dawn_index_reserve(idx, dawn_index_size(idx) + n);
            if (rc == DAWN_OK) rc = ingest_file_rows(idx, fd, dawn::RowSrc::HostPageEntries, n, 0, -1, first_id, emb_path);
            if (rc == DAWN_OK) rc = dawn::index_append_commit(idx);
            else dawn::index_append_abort(idx);
        }
        ::close(fd);
        return rc;
    });
}

int dawn_index_profile_enable(dawn_index* idx, int enable) {
    if (!idx) return fail(DAWN_ERR_INVALID_ARG, "idx is NULL");
    return dawn::guarded(
        [&] { return idx->shards ? dawn::sharded_profile_enable(idx, enable) : dawn::index_profile_enable_single(idx, enable); });
}

int dawn_index_profile_read(dawn_index* idx, uint64_t* launches, double* total_ms) {
    if (!idx || !launches || !total_ms) return fail(DAWN_ERR_INVALID_ARG, "NULL argument");
    return idx->shards ? dawn::sharded_profile_read(idx, launches, total_ms)
                       : dawn::index_profile_read_single(idx, launches, total_ms);
}

int dawn_index_stats(dawn_index* idx, uint64_t* searches, uint64_t* fallbacks) {
    return dawn_index_stats_ext(idx, searches, nullptr, fallbacks);
}

// ... plus the queries whose first certificate failed and whose 1024-deep second one held (no exact pass needed)
int dawn_index_stats_ext(dawn_index* idx, uint64_t* searches, uint64_t* second_chances, uint64_t* fallbacks) {
    if (!idx) return fail(DAWN_ERR_INVALID_ARG, "idx is NULL");
    return idx->shards ? dawn::sharded_stats(idx, searches, second_chances, fallbacks, nullptr)
                       : dawn::index_stats_single(idx, searches, second_chances, fallbacks, nullptr);
}

// ... the ladder behind the certificates: queries the bounded exact pass answered (scan_bounded.hip; NOT in `fallbacks`), single
// queries whose packed-stream certificate failed, single queries a demoted index sent to the bounded pass directly
int dawn_index_stats_ladder(dawn_index* idx, uint64_t* bounded, uint64_t* packed_failures, uint64_t* demoted) {
    if (!idx) return fail(DAWN_ERR_INVALID_ARG, "idx is NULL");
    return idx->shards ? dawn::sharded_stats(idx, nullptr, nullptr, nullptr, nullptr, bounded, packed_failures, demoted)
                       : dawn::index_stats_single(idx, nullptr, nullptr, nullptr, nullptr, bounded, packed_failures, demoted);
}

// ... what the feedback of the batched paths did (debug header): batches (of <= 256 queries) the FP6 first filter took, batches
// its feedback handed to the int8 pass instead, batches the int8 pass ran with the deeper thresholds of a ladder-heavy index
int dawn_index_debug_raw_stats(dawn_index* idx, uint64_t* out8);
int dawn_index_stats_batch_feedback(dawn_index* idx, uint64_t* f6_batches, uint64_t* f6_suspended, uint64_t* deepened_batches,
                                    uint64_t* rerun_answers) {
    if (!idx) return fail(DAWN_ERR_INVALID_ARG, "idx is NULL");
    if (f6_batches) *f6_batches = 0;
    if (f6_suspended) *f6_suspended = 0;
    if (deepened_batches) *deepened_batches = 0;
    if (idx->shards) {  // a sharded handle: the sums over its shards
        dawn::sharded_batch_feedback(idx, f6_batches, f6_suspended, deepened_batches);
    } else {
        if (f6_batches) *f6_batches = idx->n_f6_batches;
        if (f6_suspended) *f6_suspended = idx->n_f6_suspended;
        if (deepened_batches) *deepened_batches = idx->n_deepened_batches;
    }
    if (rerun_answers) {  // queries of batches answered by the second pass with exact-derived thresholds (FLAG_RERUN)
        uint64_t raw[dawn::N_STAT_SLOTS] = {};
        DAWN_TRY(dawn_index_debug_raw_stats(idx, raw));
        *rerun_answers = raw[dawn::FLAG_RERUN];
    }
    return DAWN_OK;
}

// ... the device-side counters as they are (debug header): out[N_STAT_SLOTS = 8], indexed by final flag; [5] packed-stream failures,
// [7] (row, query) pairs the bounded pass scored exactly.  A sharded handle sums its shards.
int dawn_index_debug_i6_refine(dawn_index* idx, size_t count, int* n_refine, float* frac64) {
    if (!idx || !n_refine) return fail(DAWN_ERR_INVALID_ARG, "NULL argument");
    if (idx->shards) return fail(DAWN_ERR_UNSUPPORTED, "single-device handles only");
    return dawn::guarded([&] {
        DAWN_TRY(flush_adds(idx));
        const dawn::I6Slack* sl = idx->i6_slack_ptr();
        if (frac64)
            for (int b = 0; b < 64; ++b) frac64[b] = sl ? sl->frac[b] : 0.f;
        if (!i6_live(idx)) {
            *n_refine = -1;
            return (int)DAWN_OK;
        }
        const dawn::ScanGeom g6 = idx->i6_geom();
        *n_refine = g6.refine > 0 ? g6.refine
                                  : dawn::i6_refine_count((uint32_t)idx->size, (uint32_t)count, idx->i6_bits, g6.blocks * (g6.threads / 64), sl);
        return (int)DAWN_OK;
    });
}

int dawn_index_debug_raw_stats(dawn_index* idx, uint64_t* out8) {
    if (!idx || !out8) return fail(DAWN_ERR_INVALID_ARG, "NULL argument");
    return dawn::guarded([&] {
        for (int i = 0; i < dawn::N_STAT_SLOTS; ++i) out8[i] = 0;
        std::vector<dawn_index*> parts;
        if (idx->shards) dawn::sharded_collect(idx, parts);
        else parts.push_back(idx);
        for (dawn_index* s : parts) {
            uint32_t st[dawn::N_STAT_SLOTS] = {};
            DAWN_TRY(set_device(s));
            DAWN_HIP_TRY(hipDeviceSynchronize());
            if (s->d_stats) DAWN_HIP_TRY(hipMemcpy(st, s->d_stats, sizeof(st), hipMemcpyDeviceToHost));
            for (int i = 0; i < dawn::N_STAT_SLOTS; ++i) out8[i] += st[i];
        }
        return (int)DAWN_OK;
    });
}

// ... and, of the second chances, the ones a deeper round of the same certificate (128 .. 256 rows, ~10 us each) settled
int dawn_index_stats_deep(dawn_index* idx, uint64_t* deepened) {
    if (!idx) return fail(DAWN_ERR_INVALID_ARG, "idx is NULL");
    return idx->shards ? dawn::sharded_stats(idx, nullptr, nullptr, nullptr, deepened)
                       : dawn::index_stats_single(idx, nullptr, nullptr, nullptr, deepened);
}

int dawn_index_memory(dawn_index* idx, uint64_t* rows_bytes, uint64_t* shadow_bytes, uint64_t* other_bytes) {
    if (!idx) return fail(DAWN_ERR_INVALID_ARG, "idx is NULL");
    return idx->shards ? dawn::sharded_memory(idx, rows_bytes, shadow_bytes, other_bytes)
                       : dawn::index_memory_single(idx, rows_bytes, shadow_bytes, other_bytes);
}

int dawn_index_set_option(dawn_index* idx, const char* name, int64_t value) {
    if (!idx || !name) return fail(DAWN_ERR_INVALID_ARG, "NULL argument");
    return dawn::guarded([&] {
        DAWN_TRY(flush_adds(idx));
        return idx->shards ? dawn::sharded_set_option(idx, name, value) : dawn::index_set_option_single(idx, name, value);
    });
}

// ---- test / timing hooks (single-device indexes only) ------------------------------------------------------------------

int dawn_index_debug_filter_scores(dawn_index* idx, const float* queries, size_t B, float* out, size_t* n_out) {
    if (!idx || !queries || !out || !n_out) return fail(DAWN_ERR_INVALID_ARG, "NULL argument");
    if (idx->shards) return fail(DAWN_ERR_UNSUPPORTED, "debug hooks take a single-device index");
    if (B == 0 || B > (size_t)dawn::BATCH_QT) return fail(DAWN_ERR_INVALID_ARG, "B must be 1..%d", dawn::BATCH_QT);
    DAWN_TRY(flush_adds(idx));
    DAWN_TRY(set_device(idx));
    const size_t n = std::min<size_t>(idx->size, dawn::BATCH_CAP);
    *n_out = n;
    if (n == 0) return DAWN_OK;
    DAWN_HIP_TRY(hipMemcpyAsync(idx->d_q, queries, B * dawn::EM * sizeof(float), hipMemcpyHostToDevice, idx->stream));
    if (idx->i8_batched && i8_live(idx)) {  // (upper-bound scores: scan_i8.hip)
        dawn::launch_batched_dense_scores_i8(idx->d_i8, idx->d_i8meta, (uint32_t)idx->size, idx->d_q, (int)B, idx->bws,
                                             idx->mfma_blocks, idx->stream);
    } else {
        const bool sh = f16_live(idx);
        dawn::launch_batched_dense_scores(sh ? idx->d_shadow : idx->d_x, sh ? dawn::ROW_F16S : idx->dtype, (uint32_t)idx->size,
                                          idx->d_q, (int)B, idx->bws, idx->mfma_blocks, idx->stream);
    }
    DAWN_HIP_TRY(hipGetLastError());
    DAWN_HIP_TRY(hipMemcpy2DAsync(out, n * sizeof(float), idx->bws.cand, dawn::BATCH_CAP * sizeof(float),
                                  n * sizeof(float), B, hipMemcpyDeviceToHost, idx->stream));
    DAWN_HIP_TRY(hipStreamSynchronize(idx->stream));
    return DAWN_OK;
}

// Test hook: the FP6 shadow's upper-bound scores (scan_f6.hip) of B <= 256 queries against rows [0, n), n = min(size, 8192):
// out [B][n].  Needs option "f6_shadow" = 1 (and "f6_min_rows" <= size).
int dawn_index_debug_f6_scores(dawn_index* idx, const float* queries, size_t B, float* out, size_t* n_out) {
    if (!idx || !queries || !out || !n_out) return fail(DAWN_ERR_INVALID_ARG, "NULL argument");
    if (idx->shards) return fail(DAWN_ERR_UNSUPPORTED, "debug hooks take a single-device index");
    if (B == 0 || B > (size_t)dawn::BATCH_QT) return fail(DAWN_ERR_INVALID_ARG, "B must be 1..%d", dawn::BATCH_QT);
    DAWN_TRY(flush_adds(idx));
    DAWN_TRY(set_device(idx));
    if (!f6_live(idx)) return fail(DAWN_ERR_UNSUPPORTED, "the index keeps no FP6 shadow (options f6_shadow / f6_min_rows)");
    const size_t n = std::min<size_t>(idx->size, dawn::BATCH_CAP);
    *n_out = n;
    if (n == 0) return DAWN_OK;
    DAWN_HIP_TRY(hipMemcpyAsync(idx->d_q, queries, B * dawn::EM * sizeof(float), hipMemcpyHostToDevice, idx->stream));
    dawn::launch_f6_dense_scores(idx->d_f6, idx->d_f6meta, (uint32_t)idx->size, idx->d_q, (int)B, idx->f6ws.qf6, idx->f6ws.qmeta,
                                 reinterpret_cast<float*>(idx->bws.cand), idx->stream);
    DAWN_HIP_TRY(hipGetLastError());
    DAWN_HIP_TRY(hipMemcpy2DAsync(out, n * sizeof(float), idx->bws.cand, dawn::BATCH_CAP * sizeof(float), n * sizeof(float), B,
                                  hipMemcpyDeviceToHost, idx->stream));
    DAWN_HIP_TRY(hipStreamSynchronize(idx->stream));
    return DAWN_OK;
}

// Test hook: the per-workgroup candidate lists of the streaming filter for ONE query (what merge_rescore consumes):
// out_scores / out_rows [blocks][64] descending, fillers (-inf, 0xFFFFFFFF); *n_blocks = lists written.
int dawn_index_debug_stream_lists(dawn_index* idx, const float* query, float* out_scores, uint32_t* out_rows,
                                  size_t cap_blocks, size_t* n_blocks) {
    if (!idx || !query || !out_scores || !out_rows || !n_blocks) return fail(DAWN_ERR_INVALID_ARG, "NULL argument");
    if (idx->shards) return fail(DAWN_ERR_UNSUPPORTED, "debug hooks take a single-device index");
    DAWN_TRY(flush_adds(idx));
    DAWN_TRY(set_device(idx));
    hipStream_t stream = idx->stream;
    DAWN_HIP_TRY(hipMemcpyAsync(idx->d_q, query, dawn::EM * sizeof(float), hipMemcpyHostToDevice, stream));
    size_t blocks;
    if (i6_live(idx)) {
        blocks = (size_t)idx->i6_geom().blocks;
        dawn::launch_scan_i6(idx->d_i6, idx->d_i6meta, idx->i6_bits, idx->d_i8, idx->d_i8meta, idx->d_x, idx->dtype, idx->d_ids,
                             (uint32_t)idx->size, idx->d_q, idx->d_cand_s, idx->d_cand_p, idx->d_cand_es, idx->d_cand_ep, idx->d_cand_tb,
                             idx->d_i6_pool, idx->i6_geom(), 0, nullptr, nullptr, nullptr, nullptr, 0,
                             false, stream, nullptr, nullptr);
    } else if (idx->shadow_small_batches && i8_live(idx)) {
        const dawn::ScanGeom& gh = idx->i8_geom();
        blocks = gh.blocks;
        dawn::launch_scan_filter_i8s(idx->d_i8, idx->d_i8meta, (uint32_t)idx->size, idx->d_q, 1, idx->d_cand_s, idx->d_cand_p, gh,
                                     stream, nullptr, nullptr);
    } else if (idx->dtype == DAWN_DTYPE_BF16 || (idx->shadow_small_batches && f16_live(idx))) {
        const bool own = idx->dtype == DAWN_DTYPE_BF16;
        const dawn::ScanGeom& gh = idx->shadow_geom();
        blocks = gh.blocks;
        dawn::launch_scan_filter_f16s(own ? idx->d_x : idx->d_shadow, own ? dawn::ROW_BF16 : dawn::ROW_F16S, (uint32_t)idx->size,
                                      idx->d_q, 1, idx->d_cand_s, idx->d_cand_p, gh, stream, nullptr, nullptr);
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

// Test hook: the certificate bound of the packed-shadow stream for the query of the last dawn_index_debug_stream_lists call —
// T = the largest of the workgroups' bounds on their unlisted rows (scan_i6.hip: every row in no list scores <= T).
int dawn_index_debug_stream_bound(dawn_index* idx, float* bound) {
    if (!idx || !bound) return fail(DAWN_ERR_INVALID_ARG, "NULL argument");
    if (idx->shards) return fail(DAWN_ERR_UNSUPPORTED, "debug hooks take a single-device index");
    return dawn::guarded([&] {
        DAWN_TRY(set_device(idx));
        if (!i6_live(idx)) return fail(DAWN_ERR_UNSUPPORTED, "the packed shadow is not live on this index");
        const size_t blocks = (size_t)idx->i6_geom().blocks;
        std::vector<float> tb(blocks);
        DAWN_HIP_TRY(hipDeviceSynchronize());
        DAWN_HIP_TRY(hipMemcpy(tb.data(), idx->d_cand_tb, blocks * sizeof(float), hipMemcpyDeviceToHost));
        float m = -__builtin_inff();
        for (float v : tb) m = std::max(m, v);
        *bound = m;
        return DAWN_OK;
    });
}

// Timing hook: mean duration (ms) of the matrix-core full pass alone over `iters` launches for B queries, with the
// thresholds of the last batched search (run one first) and the current mfma_sched variant; results are discarded.
int dawn_index_debug_time_full_pass(dawn_index* idx, size_t B, int iters, double* mean_ms) {
    if (!idx || !mean_ms || iters < 1) return fail(DAWN_ERR_INVALID_ARG, "bad argument");
    if (idx->shards) return fail(DAWN_ERR_UNSUPPORTED, "debug hooks take a single-device index");
    if (B == 0 || B > (size_t)dawn::BATCH_QT || !idx->bws.cand) return fail(DAWN_ERR_INVALID_ARG, "run a batched search first");
    DAWN_TRY(flush_adds(idx));
    DAWN_TRY(set_device(idx));
    hipEvent_t e0, e1;
    DAWN_HIP_TRY(hipEventCreate(&e0));
    DAWN_HIP_TRY(hipEventCreate(&e1));
    if (idx->i8_batched && i8_live(idx)) {
        dawn::launch_batched_full_pass_i8(idx->d_i8, idx->d_i8meta, (uint32_t)idx->size, (int)B, idx->bws, idx->mfma_blocks, iters,
                                          idx->stream, e0, e1);
    } else {
        const bool sh = f16_live(idx);
        dawn::launch_batched_full_pass(sh ? idx->d_shadow : idx->d_x, sh ? dawn::ROW_F16S : idx->dtype, (uint32_t)idx->size, (int)B,
                                       idx->bws, idx->mfma_blocks, iters, idx->stream, e0, e1);
    }
    DAWN_HIP_TRY(hipStreamSynchronize(idx->stream));
    float ms = 0.f;
    DAWN_HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    *mean_ms = ms / iters;
    return DAWN_OK;
}

// Diagnostic (builds with -DDAWN_EXPERIMENTS): per-wave phase cycle sums of the last batched full pass run with
// mfma_sched = 2: out [blocks][8][8].
int dawn_index_debug_read_diag(dawn_index* idx, unsigned long long* out, size_t blocks) {
    if (!idx || !out) return fail(DAWN_ERR_INVALID_ARG, "NULL argument");
    if (idx->shards || !idx->bws.diag || blocks > 4096) return fail(DAWN_ERR_INVALID_ARG, "no diagnostic buffer");
    DAWN_TRY(set_device(idx));
    DAWN_HIP_TRY(hipDeviceSynchronize());
    DAWN_HIP_TRY(hipMemcpy(out, idx->bws.diag, blocks * 64 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    return DAWN_OK;
}

}  // extern "C"
