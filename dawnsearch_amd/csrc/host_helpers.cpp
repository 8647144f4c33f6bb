// host_helpers.cpp — pure-host pieces of the C ABI (no HIP header: also part of the CPU sanitizer build of tests/native): error text and the
// bit-compatible restatements of src/search/vector.rs and src/search/best_results.rs that the
// reference's callers use right next to the index (normalisation gate, i24 wire codec, local+remote
// result merge).  Compiled with -ffp-contract=off: Rust never fuses a*b+c.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "host_common.hpp"

namespace dawn {

std::string& last_error() {
    static thread_local std::string e;
    return e;
}

// vector.rs:181-192
bool host_is_normalized(const float* v) {
    float s = 0.0f;
    for (int i = 0; i < DAWN_EM_LEN; ++i) {
        const float d = v[i] - 0.0f;
        s += d * d;
    }
    const float l = std::sqrt(s);
    if (!std::isfinite(l)) return false;
    return l > 1.0f - 0.01f && l < 1.0f + 0.01f;
}

// The same predicate for B vectors: index of the first one that fails it, or B.  Eight vectors at a time — every vector's sum is the
// sequential one of vector.rs:181-192 (the predicate's bits do not change), but eight independent chains hide the 4-cycle latency
// of the dependent adds: 256 queries 128 -> 20 us of a host call.
size_t host_first_not_normalized(const float* v, size_t B) {
    size_t b0 = 0;
    for (; b0 + 8 <= B; b0 += 8) {
        float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        const float* p = v + b0 * DAWN_EM_LEN;
        for (int i = 0; i < DAWN_EM_LEN; ++i)
            for (int j = 0; j < 8; ++j) {
                const float d = p[(size_t)j * DAWN_EM_LEN + i] - 0.0f;
                s[j] += d * d;
            }
        for (int j = 0; j < 8; ++j) {
            const float l = std::sqrt(s[j]);
            if (!std::isfinite(l) || !(l > 1.0f - 0.01f && l < 1.0f + 0.01f)) return b0 + j;
        }
    }
    for (; b0 < B; ++b0)
        if (!host_is_normalized(v + b0 * DAWN_EM_LEN)) return b0;
    return B;
}

}  // namespace dawn

using dawn::fail;
using dawn::guarded;

extern "C" {

const char* dawn_last_error(void) { return dawn::last_error().c_str(); }

int dawn_version(void) { return 100; }  // 0.1.0

// ---- src/search/vector.rs --------------------------------------------------------------------

int dawn_vec_is_normalized(const float* v) { return (v && dawn::host_is_normalized(v)) ? 1 : 0; }
size_t dawn_vec_first_not_normalized(const float* v, size_t n) { return v ? dawn::host_first_not_normalized(v, n) : 0; }

// vector.rs:194-197
void dawn_vec_normalize(float* v, size_t n) {
    if (!v) return;
    float s = 0.0f;
    for (size_t i = 0; i < n; ++i) s += v[i] * v[i];
    const float len = std::sqrt(s);
    for (size_t i = 0; i < n; ++i) v[i] /= len;
}

// vector.rs:74-86: (((x as f64 + 1.0) / 2.0) * I24_MAX as f64) as i32 -> 3 little-endian bytes
void dawn_vec_to24(const float* v, uint8_t* out) {
    if (!v || !out) return;
    for (int i = 0; i < DAWN_EM_LEN; ++i) {
        const double t = (((double)v[i] + 1.0) / 2.0) * (double)0x7FFFFF;
        int32_t iv;
        if (t != t) iv = 0;  // Rust float->int `as`: NaN -> 0, saturating
        else if (t >= 2147483647.0) iv = INT32_MAX;
        else if (t <= -2147483648.0) iv = INT32_MIN;
        else iv = (int32_t)t;
        out[i * 3 + 0] = (uint8_t)(iv & 0xFF);
        out[i * 3 + 1] = (uint8_t)((iv >> 8) & 0xFF);
        out[i * 3 + 2] = (uint8_t)((iv >> 16) & 0xFF);
    }
}

// vector.rs:57-72 (the `v |= 0xFF` "sign extend" of the low byte is reproduced as written)
int dawn_vec_from24(const uint8_t* in, float* out) {
    if (!in || !out) return fail(DAWN_ERR_INVALID_ARG, "NULL argument");
    for (int i = 0; i < DAWN_EM_LEN; ++i) {
        int32_t v = 0;
        v |= (int32_t)in[i * 3];
        v |= ((int32_t)in[i * 3 + 1]) << 8;
        v |= ((int32_t)in[i * 3 + 2]) << 16;
        if (in[i * 3 + 2] & 0x80) v |= 0xFF;
        out[i] = (float)((double)v / (double)0x7FFFFF * 2.0 - 1.0);
    }
    if (!dawn::host_is_normalized(out)) return fail(DAWN_ERR_NOT_NORMALIZED, "Embedding is not normalized");
    return DAWN_OK;
}

// Stable G-way merge of per-shard ascending lists: order by (distance, shard, position-in-shard).
int dawn_topk_merge_host(size_t G, size_t B, size_t count, const uint64_t* in_labels, const float* in_distances,
                         const uint32_t* in_found, uint64_t* labels, float* distances, uint32_t* found) {
    if (!in_labels || !in_distances || !in_found || !labels || !distances || !found)
        return fail(DAWN_ERR_INVALID_ARG, "NULL argument");
    if (G > 65536) return fail(DAWN_ERR_INVALID_ARG, "%zu shards", G);
    return guarded([&] {
    std::vector<size_t> cur(G);
    for (size_t b = 0; b < B; ++b) {
        std::fill(cur.begin(), cur.end(), 0);
        size_t n = 0;
        while (n < count) {
            size_t best = G;
            float bd = 0.f;
            for (size_t g = 0; g < G; ++g) {
                if (cur[g] >= in_found[g * B + b]) continue;
                const float d = in_distances[(g * B + b) * count + cur[g]];
                if (best == G || d < bd) {  // strict: ties stay with the lower shard
                    best = g;
                    bd = d;
                }
            }
            if (best == G) break;
            labels[b * count + n] = in_labels[(best * B + b) * count + cur[best]];
            distances[b * count + n] = bd;
            ++cur[best];
            ++n;
        }
        found[b] = (uint32_t)n;
    }
    return DAWN_OK;
    });
}

// ---- src/search/best_results.rs ----------------------------------------------------------------

struct dawn_best_results {
    struct Node {
        size_t id;
        float distance;
    };
    std::vector<Node> results;
    size_t worst_result_index = 0;
    float worst_distance = 0.0f;  // T::zero() until full (:40)
    size_t size = 0;

    bool contains_id(size_t id) const {  // :67-69
        for (const Node& n : results)
            if (n.id == id) return true;
        return false;
    }
    void update_worst() {  // :97-107
        worst_result_index = 0;
        worst_distance = results[0].distance;
        for (size_t i = 1; i < results.size(); ++i) {
            if (results[i].distance > worst_distance) {
                worst_distance = results[i].distance;
                worst_result_index = i;
            }
        }
    }
};

int dawn_best_new(size_t size, dawn_best_results** out) {
    if (!out) return fail(DAWN_ERR_INVALID_ARG, "out is NULL");
    *out = nullptr;
    return guarded([&] {
        auto* b = new dawn_best_results();
        b->size = size;
        b->results.reserve(std::min<size_t>(size, 4096));  // (Vec::with_capacity(size) :37; a huge size must not throw here)
        *out = b;
        return DAWN_OK;
    });
}

void dawn_best_free(dawn_best_results* b) { delete b; }

// 1 inserted / 0 not / negative error code.  (size 0: the reference would index an empty Vec and panic, :58; here nothing
// is ever inserted.)
int dawn_best_insert(dawn_best_results* b, size_t id, float distance) {  // :44-65
    if (!b) return fail(DAWN_ERR_INVALID_ARG, "best_results is NULL");
    return guarded([&] {
        if (b->results.size() < b->size) {
            if (b->contains_id(id)) return 0;
            b->results.push_back({id, distance});
            if (b->results.size() == b->size) b->update_worst();
            return 1;
        }
        if (b->size != 0 && distance < b->worst_distance) {
            if (b->contains_id(id)) return 0;
            b->results[b->worst_result_index] = {id, distance};
            b->update_worst();
            return 1;
        }
        return 0;
    });
}

void dawn_best_sort(dawn_best_results* b) {  // :71-79, stable like Vec::sort_by
    if (!b || b->results.empty()) return;
    for (size_t i = 1; i < b->results.size(); ++i) {
        auto t = b->results[i];
        size_t j = i;
        while (j > 0 && b->results[j - 1].distance > t.distance) {
            b->results[j] = b->results[j - 1];
            --j;
        }
        b->results[j] = t;
    }
    b->worst_result_index = b->results.size() - 1;
    b->worst_distance = b->results.back().distance;
}

float dawn_best_worst_distance(const dawn_best_results* b) { return b ? b->worst_distance : 0.0f; }
size_t dawn_best_len(const dawn_best_results* b) { return b ? b->results.size() : 0; }

int dawn_best_get(const dawn_best_results* b, size_t i, size_t* id, float* distance) {
    if (!b) return fail(DAWN_ERR_INVALID_ARG, "best_results is NULL");
    if (i >= b->results.size()) return fail(DAWN_ERR_INVALID_ARG, "index %zu out of range", i);
    if (id) *id = b->results[i].id;
    if (distance) *distance = b->results[i].distance;
    return DAWN_OK;
}

}  // extern "C"
