// kernels.hpp — launcher declarations for the gfx950 kernels of libdawn_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace dawn {

constexpr int EM = 384;        // src/search/vector.rs:26
constexpr int ROW_F4 = 96;     // float4 per row
constexpr int LIST = 64;       // shortlist length: one entry per lane of a wavefront
constexpr int ROW_PAD = 64;    // index allocations are padded to a multiple of this many rows (+ROW_PAD)

// Rigorous bound on |filter score - exact sequential-f32 dot| for vectors passing is_normalized
// (DESIGN.md §4.3): gamma_384 * 1.0201 (sequential, un-fused reference order) + gamma_16 * 1.0201
// (the filter's 8-deep FMA chain + 6-level tree) < 2.6e-5.
constexpr float FILTER_EPS_F32 = 2.6e-5f;
// MFMA filter (scan_mfma.hip): a 384-deep k-ordered FMA chain -> gamma_384 * 1.0201 on its own side.
constexpr float FILTER_EPS_MFMA = 4.8e-5f;

constexpr uint32_t FLAG_OK = 0;        // certificate holds: result is exact
constexpr uint32_t FLAG_FALLBACK = 1;  // certificate failed: the exact pass must (and will) run

struct ScanGeom {
    int blocks;           // scan grid (== number of candidate lists per query)
    int threads;          // 1024 / 512 / 256
};

// Filter pass: approximate scores for all rows, per-block top-64 lists.
//   cand_s/cand_p: [B][geom.blocks][64]
void launch_scan_filter(const float* d_x, uint32_t n_rows, const float* d_q, int B, float* cand_s,
                        uint32_t* cand_p, const ScanGeom& geom, hipStream_t stream, hipEvent_t ev0,
                        hipEvent_t ev1);
// Merge the lists, rescore the 64 survivors exactly (reference order), certify, write results.
void launch_merge_rescore(const float* d_x, const uint64_t* d_ids, uint32_t n_rows, const float* d_q, int B,
                          const float* cand_s, const uint32_t* cand_p, int n_lists, uint32_t k,
                          uint64_t* d_labels, float* d_dist, uint32_t* d_found, uint32_t* d_flags,
                          int force_fallback, float eps, const int* d_gtau, hipStream_t stream);
// Batched filter on the matrix cores (B > 8): per-wave lists [B][blocks*4][64]; d_gtau [roundup(B,64)] must be
// INT_MIN-initialised (launch_fill_i32) before each search.
void launch_scan_mfma(const float* d_x, uint32_t n_rows, const float* d_q, int B, int* d_gtau, float* cand_s,
                      uint32_t* cand_p, int blocks, hipStream_t stream, hipEvent_t ev0, hipEvent_t ev1);
void launch_fill_i32(int* d, int value, uint32_t n, hipStream_t stream);
// Exact fallback (predicated per query on d_flags[b] == FLAG_FALLBACK).
void launch_scan_exact(const float* d_x, uint32_t n_rows, const float* d_q, int B, const uint32_t* d_flags,
                       float* cand_s, uint32_t* cand_p, int n_lists, hipStream_t stream);
void launch_merge_exact(const uint64_t* d_ids, uint32_t n_rows, int B, const uint32_t* d_flags,
                        const float* cand_s, const uint32_t* cand_p, int n_lists, uint32_t k,
                        uint64_t* d_labels, float* d_dist, uint32_t* d_found, hipStream_t stream);

// Stable G-way merge of per-shard results (multi-GPU).
void launch_shard_merge(size_t G, size_t B, size_t k, const uint64_t* in_labels, const float* in_dist,
                        const uint32_t* in_found, uint64_t* out_labels, float* out_dist, uint32_t* out_found,
                        hipStream_t stream);

// is_normalized (vector.rs:185-192) over n rows; *d_bad_count += number of failing rows.
void launch_validate_rows(const float* d_rows, uint32_t n, uint32_t* d_bad_count, hipStream_t stream);
// Synthetic unit rows: rows first_row.. of stream seed -> d_out[n][384]; d_len scratch [n].
void launch_fill_synth(uint64_t seed, uint64_t first_row, uint32_t n, float* d_out, float* d_len,
                       hipStream_t stream);
void launch_iota_u64(uint64_t* d_out, uint64_t first, uint32_t n, hipStream_t stream);

}  // namespace dawn
