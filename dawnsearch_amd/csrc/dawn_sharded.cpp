// dawn_sharded.cpp — one index, its rows dealt over the GPUs of one node, driven by ONE process through the unchanged
// dawn_index_* calls (dawn_index_create_sharded): the multi-GPU form of the drop-in behind SearchProvider.
//
// Reference analogue: SearchService::search_remote (src/search/search_service.rs:201-277) fans a query out to peer
// instances and merges their top-20 lists with BestResults.  Here the "peers" are the devices of one node:
//   * rows are dealt to the shards in chunks of `shard_chunk` (4096) consecutive insertion positions, round robin — an
//     index that grows by `add` stays balanced to within one chunk, whatever its final size;
//   * every shard is a complete single-device index (dawn_index.cpp) whose labels are the rows' GLOBAL INSERTION
//     POSITIONS; the caller's u64 labels live in one table on the root device;
//   * a search copies the queries to every device over xGMI (peer copies, <= 393 KB), runs the single-device launch
//     sequence on each device's stream, brings the per-shard (position, distance)[B][k] blobs (<= 62 KB each at B = 256,
//     k = 20: latency-bound) together — ONE grouped ncclAllGather over RCCL, or G peer copies into the root's buffer
//     (option "shard_gather") —, and merges them on the root device: ties go to the lower insertion position, exactly
//     the single index's order, and the winners are translated to labels.  The answer equals the single-device answer
//     bit for bit (tests/test_sharded_capi_gpu.py).
// One host thread issues everything (the reference drives its index from one thread); the devices run concurrently.
#include <dlfcn.h>
#include <rccl/rccl.h>  // types only: the library is loaded on first use (a single-GPU deployment never needs it)

#include <condition_variable>
#include <cstring>
#include <functional>
#include <mutex>
#include <string>
#include <thread>

#include "index_internal.hpp"

namespace dawn {

namespace {

struct Rccl {
    void* lib = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    bool load() {
        if (lib) return true;
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (lib) break;
        }
        if (!lib) return false;
        CommInitAll = reinterpret_cast<decltype(CommInitAll)>(dlsym(lib, "ncclCommInitAll"));
        CommDestroy = reinterpret_cast<decltype(CommDestroy)>(dlsym(lib, "ncclCommDestroy"));
        AllGather = reinterpret_cast<decltype(AllGather)>(dlsym(lib, "ncclAllGather"));
        GroupStart = reinterpret_cast<decltype(GroupStart)>(dlsym(lib, "ncclGroupStart"));
        GroupEnd = reinterpret_cast<decltype(GroupEnd)>(dlsym(lib, "ncclGroupEnd"));
        GetErrorString = reinterpret_cast<decltype(GetErrorString)>(dlsym(lib, "ncclGetErrorString"));
        return CommInitAll && CommDestroy && AllGather && GroupStart && GroupEnd && GetErrorString;
    }
};
Rccl g_rccl;  // process-wide handle of the library; communicators are per index

constexpr int GATHER_AUTO = 0, GATHER_RCCL = 1;  // (2 = peer copies: whatever is not RCCL)

// One issuing thread per shard (shard 0: the caller's own).  A search is ~6 launches per device; issued one device after
// the other the last of 8 shards starts ~0.2 ms after the first — a quarter of a 12.5 M-row shard scan.  The workers only
// ISSUE (hipSetDevice is per thread; each touches its own shard and stream); ordering between devices stays with the
// events.  run() returns when every shard's calls have been made.
struct ShardWorkers {
    std::vector<std::thread> th;
    std::mutex mu;
    std::condition_variable cv_go, cv_done;
    std::function<int(int)> job;
    uint64_t gen = 0;
    int pending = 0;
    bool stop = false;
    std::vector<int> rc;
    std::vector<std::string> err;

    void start(int G) {
        rc.assign(G, DAWN_OK);
        err.assign(G, std::string());
        const uint64_t gen0 = gen;  // (a restart after "shard_threads" 0 -> 1 must not replay the last job)
        for (int g = 1; g < G; ++g)
            th.emplace_back([this, g, gen0] {
                uint64_t seen = gen0;
                for (;;) {
                    std::function<int(int)> f;
                    {
                        std::unique_lock<std::mutex> lk(mu);
                        cv_go.wait(lk, [&] { return stop || gen != seen; });
                        if (stop) return;
                        seen = gen;
                        f = job;
                    }
                    int r;
                    try {
                        r = f(g);
                    } catch (...) {
                        r = fail(DAWN_ERR_HIP, "exception in a shard worker");
                    }
                    {
                        std::lock_guard<std::mutex> lk(mu);
                        rc[g] = r;
                        if (r != DAWN_OK) err[g] = last_error();
                        if (--pending == 0) cv_done.notify_one();
                    }
                }
            });
    }
    // f(g) for every shard: g = 0 on the calling thread, the others on their workers; the first failure is returned (its
    // message becomes the caller's last error)
    int run(int G, const std::function<int(int)>& f) {
        if (th.empty()) {
            for (int g = 0; g < G; ++g) DAWN_TRY(f(g));
            return DAWN_OK;
        }
        {
            std::lock_guard<std::mutex> lk(mu);
            job = f;
            pending = G - 1;
            ++gen;
        }
        cv_go.notify_all();
        const int r0 = f(0);
        {
            std::unique_lock<std::mutex> lk(mu);
            cv_done.wait(lk, [&] { return pending == 0; });
        }
        if (r0 != DAWN_OK) return r0;
        for (int g = 1; g < G; ++g)
            if (rc[g] != DAWN_OK) {
                last_error() = err[g];
                return rc[g];
            }
        return DAWN_OK;
    }
    void shutdown() {
        {
            std::lock_guard<std::mutex> lk(mu);
            stop = true;
        }
        cv_go.notify_all();
        for (auto& t : th) t.join();
        th.clear();
    }
};

}  // namespace

struct ShardSet {
    int G = 0;
    int dtype = DAWN_DTYPE_F32;
    std::vector<dawn_index*> sh;
    std::vector<int> dev;
    size_t chunk = 4096;            // consecutive insertion positions per deal (option "shard_chunk", while empty)
    size_t size = 0, pending = 0, cap_reported = 0;
    uint64_t* d_gids = nullptr;     // root device: label of insertion position p
    size_t gids_cap = 0;
    std::vector<char*> d_blob;      // per shard, on its device: its result blob of the search in flight
    std::vector<char*> d_gather;    // per shard device: the G blobs back to back ([0]: what the root merges)
    std::vector<hipEvent_t> ev_done;
    hipEvent_t ev_q = nullptr;      // root: the queries of the search in flight are ready
    hipEvent_t ev_merged = nullptr; // root: the previous search's merge has read d_gather[0] (whatever stream it ran on)
    bool merged_valid = false;
    int gather = GATHER_AUTO;       // option "shard_gather"
    bool distinct = true;           // no device holds two shards (RCCL needs that)
    std::vector<ncclComm_t> comms;  // RCCL communicators (created at the first search that uses them)
    bool rccl_failed = false;
    std::string rccl_error;
    uint64_t n_searches = 0;
    ShardWorkers workers;           // option "shard_threads" (default 1: one issuing thread per shard beyond the first)

    void locate(size_t p, int* s, size_t* local) const {
        const size_t c = p / chunk;
        *s = (int)(c % (size_t)G);
        *local = (c / (size_t)G) * chunk + p % chunk;
    }
    size_t blob_cap() const { return dawn_result_blob_bytes(kMaxBatch, DAWN_MAX_K); }
    // auto: RCCL when there is more than one device; "shard_gather" = 1 forces the collective even for one shard
    bool use_rccl() const { return distinct && !rccl_failed && (gather == GATHER_RCCL || (gather == GATHER_AUTO && G > 1)); }
};

namespace {

int root_dev(const ShardSet& S) { return S.dev[0]; }

int grow_gids(ShardSet& S, size_t need) {
    if (need <= S.gids_cap) return DAWN_OK;
    DAWN_HIP_TRY(hipSetDevice(root_dev(S)));
    size_t cap = std::max({need, S.gids_cap + S.gids_cap / 2, (size_t)1024});
    uint64_t* n = nullptr;
    DAWN_HIP_TRY(hipMalloc((void**)&n, cap * sizeof(uint64_t)));
    hipStream_t st = S.sh[0]->stream;
    const size_t keep = S.size + S.pending;
    if (keep) DAWN_HIP_TRY(hipMemcpyAsync(n, S.d_gids, keep * sizeof(uint64_t), hipMemcpyDeviceToDevice, st));
    DAWN_HIP_TRY(hipDeviceSynchronize());  // (a search in flight on a caller's stream may still read the old table)
    if (S.d_gids) (void)hipFree(S.d_gids);
    S.d_gids = n;
    S.gids_cap = cap;
    return DAWN_OK;
}

int init_rccl(ShardSet& S) {
    if (!S.comms.empty()) return DAWN_OK;
    if (!g_rccl.load()) {
        S.rccl_failed = true;
        S.rccl_error = "librccl.so could not be loaded";
        return DAWN_ERR_UNSUPPORTED;
    }
    S.comms.assign(S.G, nullptr);
    const ncclResult_t r = g_rccl.CommInitAll(S.comms.data(), S.G, S.dev.data());
    if (r != ncclSuccess) {
        S.comms.clear();
        S.rccl_failed = true;
        S.rccl_error = std::string("ncclCommInitAll: ") + g_rccl.GetErrorString(r);
        return DAWN_ERR_UNSUPPORTED;
    }
    return DAWN_OK;
}

// One search of nb <= kMaxBatch queries (d_q on the root device) issued on the caller's stream `cs` of the root device.
int search_chunk(ShardSet& S, const float* d_q, size_t nb, size_t k, uint64_t* d_labels, float* d_dist, uint32_t* d_found,
                 hipStream_t cs) {
    const size_t nbytes = dawn_result_blob_bytes(nb, k);
    const size_t off_d = nb * k * 8, off_f = nb * k * 12;
    S.n_searches += nb;
    DAWN_HIP_TRY(hipSetDevice(root_dev(S)));
    DAWN_HIP_TRY(hipEventRecord(S.ev_q, cs));
    const bool rccl = S.use_rccl() && init_rccl(S) == DAWN_OK;
    // every shard's part, issued by its own thread: wait for the queries, fetch them over xGMI, the launch-only search, and
    // (peer-copy gather) the copy of its blob into the root's buffer
    DAWN_TRY(S.workers.run(S.G, [&](int g) -> int {
        dawn_index* sh = S.sh[g];
        DAWN_HIP_TRY(hipSetDevice(S.dev[g]));
        DAWN_HIP_TRY(hipStreamWaitEvent(sh->stream, S.ev_q, 0));
        // the previous search's merge may sit on ANOTHER caller stream: its blobs / gather buffer are reused from here on
        if (S.merged_valid) DAWN_HIP_TRY(hipStreamWaitEvent(sh->stream, S.ev_merged, 0));
        const float* q = d_q;
        if (S.dev[g] != root_dev(S)) {
            DAWN_HIP_TRY(hipMemcpyPeerAsync(sh->d_q, S.dev[g], d_q, root_dev(S), nb * EM * sizeof(float), sh->stream));
            q = sh->d_q;
        }
        char* blob = S.d_blob[g];
        DAWN_TRY(index_search_on_device(sh, q, nb, k, reinterpret_cast<uint64_t*>(blob), reinterpret_cast<float*>(blob + off_d),
                                        reinterpret_cast<uint32_t*>(blob + off_f), sh->stream));
        if (!rccl) {
            DAWN_HIP_TRY(hipMemcpyPeerAsync(S.d_gather[0] + (size_t)g * nbytes, root_dev(S), blob, S.dev[g], nbytes, sh->stream));
            DAWN_HIP_TRY(hipEventRecord(S.ev_done[g], sh->stream));
        }
        return DAWN_OK;
    }));
    bool gathered = false;
    if (rccl) {
        // ONE collective: every device contributes its blob and receives all G (the root's copy feeds the merge)
        ncclResult_t r = g_rccl.GroupStart();
        for (int g = 0; g < S.G && r == ncclSuccess; ++g) {
            (void)hipSetDevice(S.dev[g]);
            r = g_rccl.AllGather(S.d_blob[g], S.d_gather[g], nbytes, ncclUint8, S.comms[g], S.sh[g]->stream);
        }
        const ncclResult_t re = g_rccl.GroupEnd();
        if (r != ncclSuccess || re != ncclSuccess)
            return fail(DAWN_ERR_HIP, "ncclAllGather: %s", g_rccl.GetErrorString(r != ncclSuccess ? r : re));
        DAWN_HIP_TRY(hipSetDevice(root_dev(S)));
        DAWN_HIP_TRY(hipEventRecord(S.ev_done[0], S.sh[0]->stream));
        DAWN_HIP_TRY(hipStreamWaitEvent(cs, S.ev_done[0], 0));
        gathered = true;
    }
    if (!gathered) {
        // (peer copies over xGMI into the root's buffer were issued behind each shard's search, on that shard's stream)
        DAWN_HIP_TRY(hipSetDevice(root_dev(S)));
        for (int g = 0; g < S.G; ++g) DAWN_HIP_TRY(hipStreamWaitEvent(cs, S.ev_done[g], 0));
    }
    const char* base = S.d_gather[0];
    launch_shard_merge((size_t)S.G, nb, k, reinterpret_cast<const uint64_t*>(base), reinterpret_cast<const float*>(base + off_d),
                       reinterpret_cast<const uint32_t*>(base + off_f), nbytes / 8, nbytes / 4, nbytes / 4, S.d_gids, d_labels,
                       d_dist, d_found, cs);
    DAWN_HIP_TRY(hipGetLastError());
    DAWN_HIP_TRY(hipEventRecord(S.ev_merged, cs));
    S.merged_valid = true;
    return DAWN_OK;
}

}  // namespace

void sharded_destroy(dawn_index* idx) {
    ShardSet* S = idx->shards;
    S->workers.shutdown();
    for (int g = 0; g < (int)S->sh.size(); ++g) {
        (void)hipSetDevice(S->dev[g]);
        (void)hipDeviceSynchronize();
    }
    for (ncclComm_t c : S->comms)
        if (c) (void)g_rccl.CommDestroy(c);
    for (int g = 0; g < (int)S->sh.size(); ++g) {
        (void)hipSetDevice(S->dev[g]);
        if (g < (int)S->d_blob.size() && S->d_blob[g]) (void)hipFree(S->d_blob[g]);
        if (g < (int)S->d_gather.size() && S->d_gather[g]) (void)hipFree(S->d_gather[g]);
        if (g < (int)S->ev_done.size() && S->ev_done[g]) (void)hipEventDestroy(S->ev_done[g]);
        index_destroy_single(S->sh[g]);
    }
    if (!S->dev.empty()) (void)hipSetDevice(S->dev[0]);
    if (S->d_gids) (void)hipFree(S->d_gids);
    if (S->ev_q) (void)hipEventDestroy(S->ev_q);
    if (S->ev_merged) (void)hipEventDestroy(S->ev_merged);
    delete S;
    delete idx;
}

int sharded_root_device(const dawn_index* idx) { return idx->shards->dev[0]; }
int sharded_dtype(const dawn_index* idx) { return idx->shards->dtype; }
size_t sharded_size(const dawn_index* idx) { return idx->shards->size; }
size_t sharded_capacity(const dawn_index* idx) { return idx->shards->cap_reported; }

int sharded_reserve(dawn_index* idx, size_t capacity) {
    ShardSet& S = *idx->shards;
    const size_t chunks = (capacity + S.chunk - 1) / S.chunk;
    const size_t per = (chunks + S.G - 1) / S.G * S.chunk;
    for (dawn_index* sh : S.sh) DAWN_TRY(index_reserve_single(sh, per));
    DAWN_TRY(grow_gids(S, capacity));
    if (capacity > S.cap_reported) S.cap_reported = capacity;
    return DAWN_OK;
}

int sharded_append_async(dawn_index* idx, RowSrc kind, const void* h_src, const uint64_t* h_ids, uint64_t first_label, size_t m,
                         int slot) {
    ShardSet& S = *idx->shards;
    if (m == 0) return DAWN_OK;
    const size_t p0 = S.size + S.pending;
    DAWN_TRY(grow_gids(S, p0 + m));
    // labels: the root's table, indexed by insertion position
    DAWN_HIP_TRY(hipSetDevice(root_dev(S)));
    hipStream_t rs = S.sh[0]->stream;
    if (h_ids) {
        DAWN_HIP_TRY(hipMemcpyAsync(S.d_gids + p0, h_ids, m * sizeof(uint64_t), hipMemcpyHostToDevice, rs));
    } else {
        for (size_t o = 0; o < m; o += (size_t)1 << 30)
            launch_iota_u64(S.d_gids + p0 + o, first_label + o, (uint32_t)std::min<size_t>((size_t)1 << 30, m - o), rs);
    }
    // rows: whole chunks of insertion positions, dealt round robin; a shard labels its rows with their positions
    const size_t rec = kind == RowSrc::HostPageEntries ? 1568 : EM * sizeof(float);
    for (size_t o = 0; o < m;) {
        const size_t p = p0 + o;
        const size_t run = std::min(m - o, S.chunk - p % S.chunk);
        int s;
        size_t local;
        S.locate(p, &s, &local);
        dawn_index* sh = S.sh[s];
        if (sh->size + sh->pending != local) return fail(DAWN_ERR_INVALID_ARG, "sharded index: shard %d is out of step", s);
        DAWN_TRY(index_append_async(sh, kind, reinterpret_cast<const char*>(h_src) + o * rec, nullptr, (uint64_t)p, run, slot));
        o += run;
    }
    if (slot >= 0 && h_ids) {  // the label copy read the caller's buffer too
        DAWN_HIP_TRY(hipSetDevice(root_dev(S)));
        DAWN_HIP_TRY(hipEventRecord(S.sh[0]->ev_slot[slot], rs));
        S.sh[0]->ev_slot_used[slot] = true;
    }
    S.pending += m;
    return DAWN_OK;
}

int sharded_append_wait(dawn_index* idx, int slot) {
    for (dawn_index* sh : idx->shards->sh) DAWN_TRY(index_append_wait(sh, slot));
    return DAWN_OK;
}

int sharded_append_commit(dawn_index* idx) {
    ShardSet& S = *idx->shards;
    if (S.pending == 0) return DAWN_OK;
    uint32_t bad = 0;
    for (dawn_index* sh : S.sh) {  // all shards first: either every pending row becomes live, or none
        uint32_t b = 0;
        DAWN_TRY(index_append_check(sh, &b));
        bad += b;
    }
    DAWN_HIP_TRY(hipSetDevice(root_dev(S)));
    DAWN_HIP_TRY(hipStreamSynchronize(S.sh[0]->stream));  // the label table
    S.sh[0]->ev_slot_used[0] = S.sh[0]->ev_slot_used[1] = false;
    for (dawn_index* sh : S.sh) DAWN_TRY(index_append_finish(sh, bad == 0));
    const size_t n = S.pending;
    S.pending = 0;
    if (bad) return fail(DAWN_ERR_NOT_NORMALIZED, "Insert embedding is not normalized (%u of %zu rows)", bad, n);
    S.size += n;
    if (S.size > S.cap_reported) S.cap_reported = S.size;
    return DAWN_OK;
}

void sharded_append_abort(dawn_index* idx) {
    ShardSet& S = *idx->shards;
    for (dawn_index* sh : S.sh) index_append_abort(sh);
    (void)hipSetDevice(root_dev(S));
    (void)hipStreamSynchronize(S.sh[0]->stream);
    S.pending = 0;
}

int sharded_clear(dawn_index* idx) {
    ShardSet& S = *idx->shards;
    for (dawn_index* sh : S.sh) DAWN_TRY(index_clear(sh));
    S.size = 0;
    S.pending = 0;
    return DAWN_OK;
}

int sharded_search_device(dawn_index* idx, const float* d_q, size_t B, size_t k, uint64_t* d_labels, float* d_dist,
                          uint32_t* d_found, hipStream_t stream) {
    ShardSet& S = *idx->shards;
    for (size_t b0 = 0; b0 < B; b0 += kMaxBatch) {
        const size_t nb = std::min(kMaxBatch, B - b0);
        DAWN_TRY(search_chunk(S, d_q + b0 * EM, nb, k, d_labels + b0 * k, d_dist + b0 * k, d_found + b0, stream));
    }
    return DAWN_OK;
}

int sharded_search_batch(dawn_index* idx, const float* queries, size_t B, size_t count, uint64_t* labels, float* distances,
                         size_t* found) {
    ShardSet& S = *idx->shards;
    dawn_index* r = S.sh[0];  // the root shard's host-API staging carries queries in and results out
    for (size_t b0 = 0; b0 < B; b0 += kMaxBatch) {
        const size_t nb = std::min(kMaxBatch, B - b0);
        char* hp = (char*)r->h_pinned;
        float* hq = (float*)hp;
        uint64_t* hl = (uint64_t*)(hp + kMaxBatch * EM * 4);
        float* hd = (float*)(hp + kMaxBatch * (EM * 4 + DAWN_MAX_K * 8));
        uint32_t* hf = (uint32_t*)(hp + kMaxBatch * (EM * 4 + DAWN_MAX_K * 8 + DAWN_MAX_K * 4));
        std::memcpy(hq, queries + b0 * EM, nb * EM * sizeof(float));
        DAWN_HIP_TRY(hipSetDevice(root_dev(S)));
        DAWN_HIP_TRY(hipMemcpyAsync(r->d_q, hq, nb * EM * sizeof(float), hipMemcpyHostToDevice, r->stream));
        DAWN_TRY(search_chunk(S, r->d_q, nb, count, r->d_labels, r->d_dist, r->d_found, r->stream));
        DAWN_HIP_TRY(hipSetDevice(root_dev(S)));
        DAWN_HIP_TRY(hipMemcpyAsync(hl, r->d_labels, nb * count * sizeof(uint64_t), hipMemcpyDeviceToHost, r->stream));
        DAWN_HIP_TRY(hipMemcpyAsync(hd, r->d_dist, nb * count * sizeof(float), hipMemcpyDeviceToHost, r->stream));
        DAWN_HIP_TRY(hipMemcpyAsync(hf, r->d_found, nb * sizeof(uint32_t), hipMemcpyDeviceToHost, r->stream));
        DAWN_HIP_TRY(hipStreamSynchronize(r->stream));
        std::memcpy(labels + b0 * count, hl, nb * count * sizeof(uint64_t));
        std::memcpy(distances + b0 * count, hd, nb * count * sizeof(float));
        for (size_t b = 0; b < nb; ++b) found[b0 + b] = hf[b];
    }
    return DAWN_OK;
}

int sharded_fill_synthetic(dawn_index* idx, uint64_t seed, uint64_t first_row, size_t n, uint64_t first_id) {
    ShardSet& S = *idx->shards;
    DAWN_TRY(sharded_reserve(idx, S.size + n));
    const size_t p0 = S.size;
    DAWN_HIP_TRY(hipSetDevice(root_dev(S)));
    for (size_t o = 0; o < n; o += (size_t)1 << 30)
        launch_iota_u64(S.d_gids + p0 + o, first_id + o, (uint32_t)std::min<size_t>((size_t)1 << 30, n - o), S.sh[0]->stream);
    int rc = DAWN_OK;
    for (size_t o = 0; o < n && rc == DAWN_OK;) {
        const size_t p = p0 + o;
        const size_t run = std::min(n - o, S.chunk - p % S.chunk);
        int s;
        size_t local;
        S.locate(p, &s, &local);
        rc = index_fill_async(S.sh[s], seed, first_row + o, run, 0, true, (uint64_t)p);
        o += run;
    }
    for (dawn_index* sh : S.sh) {
        if (rc == DAWN_OK) rc = index_append_finish(sh, true);
        else index_append_abort(sh);
    }
    DAWN_HIP_TRY(hipSetDevice(root_dev(S)));
    DAWN_HIP_TRY(hipStreamSynchronize(S.sh[0]->stream));
    if (rc != DAWN_OK) return rc;
    S.size += n;
    if (S.size > S.cap_reported) S.cap_reported = S.size;
    return DAWN_OK;
}

int sharded_get_rows(dawn_index* idx, size_t first, size_t n, float* out_rows, uint64_t* out_ids) {
    ShardSet& S = *idx->shards;
    if (first + n > S.size) return fail(DAWN_ERR_INVALID_ARG, "rows [%zu, %zu) out of range (size %zu)", first, first + n, S.size);
    if (n == 0) return DAWN_OK;
    if (out_rows) {
        for (size_t o = 0; o < n;) {
            const size_t p = first + o;
            const size_t run = std::min(n - o, S.chunk - p % S.chunk);
            int s;
            size_t local;
            S.locate(p, &s, &local);
            DAWN_TRY(index_get_rows_single(S.sh[s], local, run, out_rows + o * EM, nullptr));
            o += run;
        }
    }
    if (out_ids) {
        DAWN_HIP_TRY(hipSetDevice(root_dev(S)));
        DAWN_HIP_TRY(hipMemcpy(out_ids, S.d_gids + first, n * sizeof(uint64_t), hipMemcpyDeviceToHost));
    }
    return DAWN_OK;
}

int sharded_set_option(dawn_index* idx, const char* name, int64_t value) {
    ShardSet& S = *idx->shards;
    const std::string n(name);
    if (n == "shard_chunk") {
        if (S.size || S.pending) return fail(DAWN_ERR_INVALID_ARG, "shard_chunk can only be set on an empty index");
        if (value < 64 || value > (1 << 24) || value % 64) return fail(DAWN_ERR_INVALID_ARG, "shard_chunk must be a multiple of 64 in 64..2^24");
        S.chunk = (size_t)value;
        return DAWN_OK;
    }
    if (n == "shard_threads") {  // 1: one issuing thread per shard beyond the first (default); 0: the caller issues everything
        S.workers.shutdown();
        S.workers.stop = false;
        if (value && S.G > 1) S.workers.start(S.G);
        return DAWN_OK;
    }
    if (n == "shard_gather") {
        if (value < 0 || value > 2) return fail(DAWN_ERR_INVALID_ARG, "shard_gather: 0 auto, 1 RCCL all-gather, 2 peer copies");
        if (value == GATHER_RCCL && !S.distinct)
            return fail(DAWN_ERR_UNSUPPORTED, "RCCL needs one shard per device (this index has devices holding several)");
        S.gather = (int)value;
        if (value == GATHER_RCCL) {
            S.rccl_failed = false;
            if (init_rccl(S) != DAWN_OK) return fail(DAWN_ERR_UNSUPPORTED, "RCCL is not usable: %s", S.rccl_error.c_str());
        }
        return DAWN_OK;
    }
    for (dawn_index* sh : S.sh) DAWN_TRY(index_set_option_single(sh, name, value));
    return DAWN_OK;
}

int sharded_memory(dawn_index* idx, uint64_t* rows_bytes, uint64_t* shadow_bytes, uint64_t* other_bytes) {
    ShardSet& S = *idx->shards;
    uint64_t r = 0, s = 0, o = 0;
    for (dawn_index* sh : S.sh) {
        uint64_t a = 0, b = 0, c = 0;
        DAWN_TRY(index_memory_single(sh, &a, &b, &c));
        r += a;
        s += b;
        o += c;
    }
    o += S.gids_cap * sizeof(uint64_t) + (uint64_t)S.G * (S.blob_cap() + (uint64_t)S.G * S.blob_cap());
    if (rows_bytes) *rows_bytes = r;
    if (shadow_bytes) *shadow_bytes = s;
    if (other_bytes) *other_bytes = o;
    return DAWN_OK;
}

int sharded_stats(dawn_index* idx, uint64_t* searches, uint64_t* second, uint64_t* fallbacks, uint64_t* deepened,
                  uint64_t* bounded, uint64_t* packed_failures, uint64_t* demoted) {
    ShardSet& S = *idx->shards;
    uint64_t s2 = 0, fb = 0, dp = 0, bd = 0, pf = 0, dm = 0;
    for (dawn_index* sh : S.sh) {  // (per-shard events: one query can fall back on one shard and not on another)
        uint64_t a = 0, b = 0, c = 0, d = 0, e = 0, f = 0;
        DAWN_TRY(index_stats_single(sh, nullptr, &a, &b, &c, &d, &e, &f));
        s2 += a;
        fb += b;
        dp += c;
        bd += d;
        pf += e;
        dm += f;
    }
    if (bounded) *bounded = bd;
    if (packed_failures) *packed_failures = pf;
    if (demoted) *demoted = dm;
    if (searches) *searches = S.n_searches;
    if (second) *second = s2;
    if (fallbacks) *fallbacks = fb;
    if (deepened) *deepened = dp;
    return DAWN_OK;
}

void sharded_collect(dawn_index* idx, std::vector<dawn_index*>& out) {
    for (dawn_index* sh : idx->shards->sh) out.push_back(sh);
}

// the batch feedback's host-side counters, summed over the shards (the sharded handle used to refuse)
void sharded_batch_feedback(dawn_index* idx, uint64_t* f6_batches, uint64_t* f6_suspended, uint64_t* deepened_batches) {
    for (dawn_index* sh : idx->shards->sh) {
        if (f6_batches) *f6_batches += sh->n_f6_batches;
        if (f6_suspended) *f6_suspended += sh->n_f6_suspended;
        if (deepened_batches) *deepened_batches += sh->n_deepened_batches;
    }
}

int sharded_profile_enable(dawn_index* idx, int enable) {
    for (dawn_index* sh : idx->shards->sh) DAWN_TRY(index_profile_enable_single(sh, enable));
    return DAWN_OK;
}

// launches / summed ms over ALL shards: total_ms / launches is the mean duration of one shard's dominant kernel
int sharded_profile_read(dawn_index* idx, uint64_t* launches, double* total_ms) {
    uint64_t n = 0;
    double ms = 0;
    for (dawn_index* sh : idx->shards->sh) {
        uint64_t a = 0;
        double b = 0;
        DAWN_TRY(index_profile_read_single(sh, &a, &b));
        n += a;
        ms += b;
    }
    *launches = n;
    *total_ms = ms;
    return DAWN_OK;
}

}  // namespace dawn

using dawn::fail;

extern "C" {

int dawn_index_create_sharded(size_t dims, int dtype, int n_gpus, const int* devices, dawn_index** out) {
    if (!out) return fail(DAWN_ERR_INVALID_ARG, "out is NULL");
    *out = nullptr;
    if (dims != DAWN_EM_LEN) return fail(DAWN_ERR_UNSUPPORTED, "dims must be %d (EM_LEN)", DAWN_EM_LEN);
    if (dtype != DAWN_DTYPE_F32 && dtype != DAWN_DTYPE_BF16) return fail(DAWN_ERR_UNSUPPORTED, "dtype %d not supported", dtype);
    if (n_gpus < 1 || n_gpus > 64) return fail(DAWN_ERR_INVALID_ARG, "n_gpus must be 1..64");
    return dawn::guarded([&]() -> int {
        auto* idx = new dawn_index();
        auto* S = new dawn::ShardSet();
        idx->shards = S;
        idx->dtype = dtype;
        S->G = n_gpus;
        S->dtype = dtype;
        auto bail = [&](int rc) {
            const std::string msg = dawn::last_error();
            dawn::sharded_destroy(idx);
            dawn::last_error() = msg;
            return rc;
        };
        for (int g = 0; g < n_gpus; ++g) {
            const int d = devices ? devices[g] : g;
            for (int o : S->dev)
                if (o == d) S->distinct = false;
            dawn_index* sh = nullptr;
            int rc = dawn::index_create_single(dtype, d, &sh);
            if (rc != DAWN_OK) return bail(rc);
            sh->pos_ids = true;
            S->sh.push_back(sh);
            S->dev.push_back(d);
        }
        idx->device = S->dev[0];
        const size_t cap = S->blob_cap();
        S->d_blob.assign(n_gpus, nullptr);
        S->d_gather.assign(n_gpus, nullptr);
        S->ev_done.assign(n_gpus, nullptr);
        for (int g = 0; g < n_gpus; ++g) {
            if (hipSetDevice(S->dev[g]) != hipSuccess || hipMalloc((void**)&S->d_blob[g], cap) != hipSuccess ||
                hipMalloc((void**)&S->d_gather[g], cap * n_gpus) != hipSuccess ||
                hipEventCreateWithFlags(&S->ev_done[g], hipEventDisableTiming) != hipSuccess)
                return bail(fail(DAWN_ERR_OOM, "allocating the result blobs of shard %d failed", g));
            // peer mappings between the root and every other device (the query / blob copies then go straight over xGMI)
            if (S->dev[g] != S->dev[0]) {
                (void)hipDeviceEnablePeerAccess(S->dev[0], 0);
                (void)hipSetDevice(S->dev[0]);
                (void)hipDeviceEnablePeerAccess(S->dev[g], 0);
                (void)hipGetLastError();  // (already enabled / not supported: the copies still work, staged)
            }
        }
        if (hipSetDevice(S->dev[0]) != hipSuccess || hipEventCreateWithFlags(&S->ev_q, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&S->ev_merged, hipEventDisableTiming) != hipSuccess)
            return bail(fail(DAWN_ERR_HIP, "creating the query event failed"));
        // issuing threads pay off when the shards sit on different devices (per-device queues); logical shards on one
        // device contend for the same queue: measured 0.302 -> 0.359 ms per search for 8 shards of 25 k rows
        // (tools/shard_issue_time.py) — there the caller issues everything unless "shard_threads" = 1 asks otherwise
        if (n_gpus > 1 && S->distinct) S->workers.start(n_gpus);
        *out = idx;
        return DAWN_OK;
    });
}

int dawn_index_shard_info(dawn_index* idx, int* n_shards, int* gather, size_t* shard_sizes, size_t cap) {
    if (!idx) return fail(DAWN_ERR_INVALID_ARG, "idx is NULL");
    {
        const int rc = dawn::guarded([&] { return dawn::index_flush_adds(idx); });
        if (rc != DAWN_OK) return rc;
    }
    if (!idx->shards) {
        if (n_shards) *n_shards = 1;
        if (gather) *gather = 0;
        if (shard_sizes && cap >= 1) shard_sizes[0] = idx->size;
        return DAWN_OK;
    }
    dawn::ShardSet& S = *idx->shards;
    if (n_shards) *n_shards = S.G;
    if (gather) *gather = (S.use_rccl() && !S.comms.empty()) ? 1 : S.use_rccl() ? -1 : S.G == 1 ? 0 : 2;
    for (int g = 0; shard_sizes && g < S.G && (size_t)g < cap; ++g) shard_sizes[g] = S.sh[g]->size;
    return DAWN_OK;
}

}  // extern "C"
