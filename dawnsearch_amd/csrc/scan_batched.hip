// scan_batched.hip — batched query x index contraction on the matrix cores (gfx950), B = 4..256 per pass.
//
// One pass over the f32 index serves up to 256 queries: the row stream stays HBM-bound (1536 B/row read
// once), the contraction runs on v_mfma_f32_32x32x16_f16 at 1/3 of its peak.  The f16 scores are a FILTER:
// a rigorous bound FILTER_EPS_F16 on |filter - exact| lets the tail (select_rescore_kernel) rescore a
// 64-row shortlist in the reference's sequential f32 order (src/search/vector.rs:128-134) and certify that
// no other row can reach the top-k; results are therefore bit-identical to the exact scan.
//
// Structure per search of <= 256 queries (all launches on one stream, no host decisions in between):
//   prep_queries_kernel     q f32 -> f16(256*q), zero rows for b >= B                      [256][384] f16
//   scan_f16_kernel<DENSE>  sample pass: scores of 8192 strided rows, stored densely       [256][8192] f32
//   tau_select_kernel       per query: the m-th largest sample score -> threshold tau
//   scan_f16_kernel<APPEND> (large N only: a second, larger sample, then tau again)
//   scan_f16_kernel<APPEND> full pass: every row with score > tau is appended to the query's candidate
//                           buffer (score, row) through an atomic counter; ~1.5k of N rows qualify
//   select_rescore_kernel   top-64 of the candidates, exact rescore, certificate, output / fallback flag
//
// scan_f16_kernel: one 512-thread workgroup (8 waves, 2 per SIMD) per CU, grid-strided over 64-row tiles so
// that the chip reads one moving contiguous window of HBM.  Wave w owns queries 32w..32w+31: their B-operand
// fragments (24 k-steps x 8 f16) live in 96 VGPRs for the whole kernel.  A row tile is loaded once per
// workgroup with fully coalesced 16-B/lane non-temporal loads (12 per lane, issued one tile ahead), scaled,
// converted to f16 and written to LDS in MFMA-fragment order ([k-group g][row] 16-B slots, rows rotated by
// g&7: conflict-free ds_write_b64 and ds_read_b128); every wave then reads the A fragments back (one
// ds_read_b128 per MFMA).  D[row][query]: a lane holds 16 rows of ONE query, so the threshold is one VGPR.
#include <type_traits>

#include "kernels.hpp"
#include "wave_topk.hpp"

namespace dawn {

typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
// one MFMA on fragments held as half8 registers (bf16 data is reinterpreted)
template <bool BF16>
__device__ __forceinline__ f32x16 mfma16(const half8& a, const half8& b, const f32x16& c) {
    if (BF16)
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}

constexpr int TILE_ROWS = BATCH_TILE_ROWS;       // 64
constexpr int G_STRIDE = 40 * 16;                // LDS bytes per k-group: 32 row slots + 8 (skew room)
constexpr int SUB_BYTES = 48 * G_STRIDE;         // one 32-row f16 sub-tile: 30 KiB
constexpr int TILE_BYTES = 2 * SUB_BYTES;        // 60 KiB
constexpr float ROW_SCALE = 256.0f;              // rows and queries are scaled by 2^8 before f16 conversion:
constexpr float SCORE_SCALE = 65536.0f;          // keeps small components out of the f16 subnormal range
constexpr uint32_t STAGE_CAP = 2048;             // LDS-staged candidates per workgroup (q, score, row)
constexpr uint32_t STAGE_FLUSH_AT = 1024;        // flush to the global per-query buffers beyond this fill
constexpr int LDS_BYTES = 2 * TILE_BYTES + (int)STAGE_CAP * 12 + 16;

// A query's candidate buffer is cut into CAND_SEGS segments with a counter each; a workgroup (or wave) appends to the
// segment picked by its index.  One counter per query would take ~1500 same-address atomics per pass; on a short pass
// (1 M rows: 150 us) they arrive faster than the memory-side atomic unit retires them (measured +130 us).
constexpr uint32_t SEG_CAP = (uint32_t)BATCH_CAP / (uint32_t)BATCH_CAND_SEGS;  // 512

__device__ __forceinline__ void append_candidate(uint32_t* __restrict__ cnt, uint2* __restrict__ cand, uint32_t q,
                                                 uint32_t seg, uint32_t score_bits, uint32_t row) {
    const uint32_t slot = atomicAdd(&cnt[q * BATCH_CAND_SEGS + seg], 1u);
    if (slot < SEG_CAP) cand[(size_t)q * BATCH_CAP + seg * SEG_CAP + slot] = make_uint2(score_bits, row);
}

// BF16: a bf16 index — the queries become plain bf16 (round to nearest even, no scaling)
template <bool BF16>
__global__ __launch_bounds__(256) void prep_queries_kernel(const float* __restrict__ q, int n_q,
                                                          unsigned short* __restrict__ qh) {
    const int i = blockIdx.x * 256 + threadIdx.x;  // over BATCH_QT * EM
    if (i >= BATCH_QT * EM) return;
    const int b = i / EM;
    const float v = (b < n_q) ? q[i] : 0.0f;
    qh[i] = BF16 ? (unsigned short)f32_to_bf16_rne(v) : __builtin_bit_cast(unsigned short, (_Float16)(v * ROW_SCALE));
}

__device__ __forceinline__ half4 to_half4_scaled(const f32x4& v) {
    half4 r;
    r.x = (_Float16)(v.x * ROW_SCALE);  // round-to-nearest-even conversions
    r.y = (_Float16)(v.y * ROW_SCALE);
    r.z = (_Float16)(v.z * ROW_SCALE);
    r.w = (_Float16)(v.w * ROW_SCALE);
    return r;
}

// f32 chunk (4 values) -> 8 B at +32 B per step
__device__ __forceinline__ void store_converted(unsigned char* base, int step, const f32x4& v) {
    *reinterpret_cast<half4*>(base + step * 32) = to_half4_scaled(v);
}

// LDS byte offset (inside one tile buffer) of the 16-B slot holding f16 elements k = 8g..8g+7 of tile row `row`:
// [sub-tile][g][row + (g & 7)].  The skew by g & 7 spreads one row's consecutive k-groups over all banks
// (ds_write_b64 of 16 consecutive chunks is conflict-free); a k-group's 32 rows stay contiguous (ds_read_b128
// of one MFMA operand is conflict-free).  G_STRIDE = 640 B = 5 x 128 B keeps the bank phase of every group equal.
__device__ __forceinline__ uint32_t slot_off(uint32_t row, uint32_t g) {
    return (row >> 5) * SUB_BYTES + g * G_STRIDE + ((row & 31u) + (g & 7u)) * 16u;
}

// Tiles visited by a pass: first_tile + i*tile_stride, i < n_tiles.  DENSE: score of sample row (i*64 + r) of
// query b goes to dense[b][i*64 + r] (-inf past the end of the index); n_tiles*64 <= BATCH_CAP.
// NW waves per workgroup (one workgroup per CU); a wave owns QG = 8/NW groups of 32 queries and converts
// 64/NW rows of every tile; PF tiles are in flight per wave (register-staged: 96/NW 16-B loads per lane and tile).
//   NW = 4: one wave per SIMD, 512 registers: 2 x 24 loads in flight per lane = 192 KiB per CU
//   NW = 8: two waves per SIMD, 256 registers: 12 loads in flight per lane     =  96 KiB per CU
template <bool DENSE, int NW, int SCHED>
__global__ __launch_bounds__(NW * 64) void scan_f16_kernel(const void* __restrict__ xv, uint32_t n_rows,
                                                          uint32_t first_tile, uint32_t tile_stride,
                                                          uint32_t n_tiles, const half8* __restrict__ qh, int n_q,
                                                          const float* __restrict__ tau, uint32_t* __restrict__ cnt,
                                                          uint2* __restrict__ cand, float* __restrict__ dense,
                                                          unsigned long long* __restrict__ diag) {
    // SCHED == 2: diagnostic build of the lockstep schedule — s_memtime stamps around the phases of every tile,
    // per-wave sums written to diag[block][wave][8] (shares only: the stamps' waits forbid overlaps the real kernel has)
    unsigned long long seg[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long t_prev = 0;
    auto stamp = [&](int k) {
        if (SCHED == 2) {
            unsigned long long t;
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
            __builtin_amdgcn_sched_barrier(0);
            if (k >= 0) seg[k] += t - t_prev;
            t_prev = t;
        }
    };
    constexpr int QG = 8 / NW;          // 32-query groups per wave
    constexpr int CPR = ROW_F4;         // 16-B chunks per index row (f32: 4 values each)
    constexpr int LPL = CPR / NW;       // loads per lane per tile
    constexpr int RPW = TILE_ROWS / NW; // tile rows converted by one wave
    constexpr int PF = NW == 4 ? 2 : 1; // tiles in flight
    constexpr int NT = NW * 64;
    // 2 x TILE_BYTES | stage_q[STAGE_CAP] | stage_s[STAGE_CAP] | stage_r[STAGE_CAP] | count, latch[2]
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    uint32_t* stage_q = reinterpret_cast<uint32_t*>(lds + 2 * TILE_BYTES);
    float* stage_s = reinterpret_cast<float*>(stage_q + STAGE_CAP);
    uint32_t* stage_r = stage_q + 2 * STAGE_CAP;
    uint32_t* stage_n = stage_q + 3 * STAGE_CAP;  // [0] fill count, [1..2] per-iteration snapshot of it
    if (!DENSE && threadIdx.x < 3) stage_n[threadIdx.x] = 0;
    // Staged candidates -> global per-query buffers.  Called by every thread between two barriers during
    // which no wave appends.
    auto flush = [&]() {
        uint32_t n = stage_n[0];
        if (n > STAGE_CAP) n = STAGE_CAP;
        for (uint32_t e = threadIdx.x; e < n; e += NT) {
            append_candidate(cnt, cand, stage_q[e], blockIdx.x % BATCH_CAND_SEGS, __builtin_bit_cast(uint32_t, stage_s[e]),
                             stage_r[e]);
        }
    };
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t r = lane & 31, h = lane >> 5;
    const int q0 = wave * (32 * QG) + (int)r;  // this lane's queries: q0 + 32*g
    const bool wave_has_queries = wave * (32 * QG) < n_q;  // wave-uniform

    // B operand: B[k = 16s + 8h + j][col = query r], j = 0..7  ->  qh[query][2s + h]
    half8 qf[QG][24];
    float tau_s[QG];
#pragma unroll
    for (int g = 0; g < QG; ++g) {
#pragma unroll
        for (int s = 0; s < 24; ++s) qf[g][s] = qh[(size_t)(q0 + 32 * g) * 48 + 2 * s + h];
        tau_s[g] = __builtin_inff();
        if (!DENSE && q0 + 32 * g < n_q) tau_s[g] = tau[q0 + 32 * g] * SCORE_SCALE;
    }

    // producer map: wave w converts rows RPW*w.. of the tile = RPW*CPR consecutive 16-B chunks; load j = 3a + b of
    // this lane is chunk (b*64 + lane) + 192a of the f32 rows (96 chunks of 4 values each): row RPW*w + 2a +
    // (b*64+lane)/96, chunk c -> k-group c/2, half c&1, 8-B store, +32 B per a.  Three LDS addresses.
    uint32_t wr_off[3];
#pragma unroll
    for (int b = 0; b < 3; ++b) {
        const uint32_t Lb = (uint32_t)b * 64u + (uint32_t)lane;
        const uint32_t row = (uint32_t)RPW * wave + Lb / CPR, c = Lb % CPR;
        wr_off[b] = slot_off(row, c >> 1) + (c & 1u) * 8u;
    }
    // consumer map: k-step s reads slot g = 2s + h of row r: s*2*G_STRIDE + (s&3)*32 + [h*(G_STRIDE+16) + r*16]
    const uint32_t rd_off = h * (G_STRIDE + 16u) + r * 16u;

    typedef f32x4 chunk_t;
    const chunk_t* x = reinterpret_cast<const chunk_t*>(xv);
    chunk_t st[PF][LPL];
    auto issue = [&](chunk_t(&dst)[LPL], uint32_t i) {
        const chunk_t* p = x + ((size_t)first_tile + (size_t)i * tile_stride) * (TILE_ROWS * CPR) +
                           wave * (RPW * CPR) + lane;
#pragma unroll
        for (int j = 0; j < LPL; ++j) dst[j] = __builtin_nontemporal_load(p + j * 64);
    };

    // convert: the staged rows -> f16 fragments in LDS buffer `buf`; then refill the staging registers with tile `nxt`
    auto convert = [&](chunk_t(&src)[LPL], uint32_t buf, uint32_t nxt) {
        unsigned char* tb = lds + buf * TILE_BYTES;
#pragma unroll
        for (int j = 0; j < LPL; ++j) store_converted(tb + wr_off[j % 3], j / 3, src[j]);
        if (nxt < n_tiles) issue(src, nxt);
    };
    // barrier: LDS writes visible to the workgroup (the prefetch loads stay in flight across it); when the
    // candidate stage is half full (snapshot taken by thread 0 before the barrier: the same value in every wave)
    // all waves flush it — nobody appends between this barrier and the end of the flush
    auto barrier_and_flush = [&](uint32_t par) {
        if (!DENSE && threadIdx.x == 0) stage_n[1 + par] = stage_n[0];
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (!DENSE && stage_n[1 + par] >= STAGE_FLUSH_AT) {
            flush();
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            if (threadIdx.x == 0) stage_n[0] = 0;
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        }
    };
    // contract: the 64 rows of tile i (LDS buffer `buf`) against this wave's queries, threshold test / dense store
    auto contract = [&](uint32_t i, uint32_t buf) {
        const unsigned char* tb = lds + buf * TILE_BYTES;
        if (!wave_has_queries) return;  // every query of this wave is padding: it only converts rows
        const uint32_t row_base = (first_tile + i * tile_stride) * TILE_ROWS;
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
            f32x16 acc[QG];
#pragma unroll
            for (int g = 0; g < QG; ++g)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[g][e] = 0.f;
            // A fragments through a register ring PD k-steps deep: an LDS read has ~100+ cycles of latency with eight
            // waves reading, an MFMA issues every 32 — reads issued one or two MFMAs ahead leave the matrix pipe waiting
            constexpr int PD = 8;
            const unsigned char* ab = tb + sub * SUB_BYTES + rd_off;
            half8 a[PD];
#pragma unroll
            for (int d = 0; d < PD; ++d)
                a[d] = *reinterpret_cast<const half8*>(ab + d * (2 * G_STRIDE) + (d & 3) * 32);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int s = 0; s < 24; ++s) {
#pragma unroll
                for (int g = 0; g < QG; ++g)
                    acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[s % PD], qf[g][s], acc[g], 0, 0, 0);
                if (s + PD < 24)
                    a[s % PD] = *reinterpret_cast<const half8*>(ab + (s + PD) * (2 * G_STRIDE) + ((s + PD) & 3) * 32);
                __builtin_amdgcn_sched_barrier(0);  // keep the read here, PD steps ahead of its use (hipcc sinks it otherwise)
            }
            stamp(2 + 2 * sub);  // contraction of sub-tile `sub`
            // C/D map: this lane holds D[row = (e&3) + 8*(e>>2) + 4*h][query r]
            const uint32_t row0 = row_base + sub * 32 + 4 * h;
#pragma unroll
            for (int g = 0; g < QG; ++g) {
                const int qi = q0 + 32 * g;
                if (DENSE) {
                    if (qi < n_q) {
#pragma unroll
                        for (int e4 = 0; e4 < 4; ++e4) {
                            f32x4 o;
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                const uint32_t row = row0 + e + 8 * e4;
                                o[e] = row < n_rows ? acc[g][e4 * 4 + e] * (1.0f / SCORE_SCALE) : NEG_INF;
                            }
                            *reinterpret_cast<f32x4*>(dense + (size_t)qi * BATCH_CAP + (size_t)i * TILE_ROWS +
                                                      sub * 32 + 4 * h + 8 * e4) = o;
                        }
                    }
                } else {
                    float mx = acc[g][0];
#pragma unroll
                    for (int e = 1; e < 16; ++e) mx = fmaxf(mx, acc[g][e]);
                    if (__any(mx > tau_s[g])) {
                        uint32_t mask = 0;
#pragma unroll
                        for (int e = 0; e < 16; ++e) {
                            const uint32_t row = row0 + (e & 3) + 8 * (e >> 2);
                            mask |= (acc[g][e] > tau_s[g] && row < n_rows) ? (1u << e) : 0u;
                        }
                        if (mask) {
                            uint32_t pos = atomicAdd(&stage_n[0], (uint32_t)__popc(mask));  // LDS atomic
#pragma unroll
                            for (int e = 0; e < 16; ++e) {
                                if (mask & (1u << e)) {
                                    const float sc = acc[g][e] * (1.0f / SCORE_SCALE);
                                    const uint32_t row = row0 + (e & 3) + 8 * (e >> 2);
                                    if (pos < STAGE_CAP) {
                                        stage_q[pos] = (uint32_t)qi;
                                        stage_s[pos] = sc;
                                        stage_r[pos] = row;
                                    } else {  // stage full (a burst: many queries hitting the same rows): append directly
                                        append_candidate(cnt, cand, (uint32_t)qi, blockIdx.x % BATCH_CAND_SEGS,
                                                         __builtin_bit_cast(uint32_t, sc), row);
                                    }
                                    ++pos;
                                }
                            }
                        }
                    }
                }
            }
            stamp(3 + 2 * sub);  // threshold test / append of sub-tile `sub`
        }
    };

    const uint32_t G = gridDim.x;
    uint32_t i = blockIdx.x;
    {
        // lockstep: every wave converts tile i, barrier, every wave contracts tile i
#pragma unroll
        for (int f = 0; f < PF; ++f)
            if (i + f * G < n_tiles) issue(st[f], i + f * G);
        uint32_t buf = 0;
        while (i < n_tiles) {
#pragma unroll
            for (int f = 0; f < PF; ++f) {
                if (i < n_tiles) {
                    stamp(-1);
                    convert(st[f], buf, i + PF * G);
                    stamp(0);  // wait for the staged loads, convert, LDS stores, issue the next loads
                    barrier_and_flush(buf);
                    stamp(1);  // barrier (arrival skew of the workgroup's waves)
                    contract(i, buf);
                    buf ^= 1u;
                    i += G;
                }
            }
        }
    }
    if (!DENSE) {
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        flush();
    }
    if (SCHED == 2 && diag && lane == 0) {
#pragma unroll
        for (int k = 0; k < 8; ++k) diag[((size_t)blockIdx.x * NW + wave) * 8 + k] = seg[k];
    }
}

// ------------------------------------------------------------------------------------------------
// scan_f16_dma_kernel — the matrix-core filter over the scaled-f16 SHADOW rows of an f32 index (ROW_F16S): no
// conversion, no staging registers, no LDS stores.  Row tiles go HBM -> LDS by LDS-DMA (global_load_lds_dwordx4, 1 KiB
// per wave-instruction, two tiles ahead in a ring of three 48-KiB images), the waves only read fragments and issue
// MFMAs.
//   The shadow is stored tile by tile in MFMA-fragment order (ROW_F16S, kernels.hpp): the LDS image of a tile is a
//   linear copy of its 48 KiB (an LDS-DMA writes lane-linear bytes; the source is one contiguous 1 KiB per
//   instruction), and the A-operand read of fragment (sub, s) is 64 consecutive 16-B slots (conflict-free ds_read_b128).
//   Per tile and wave: 6 DMA instructions, 48 ds_read_b128, 48 MFMAs, two threshold tests, ONE barrier
//   (s_waitcnt vmcnt(6): the tile for the next iteration has landed, the one after stays in flight).
// ------------------------------------------------------------------------------------------------
constexpr int DMA_TILE_BYTES = TILE_ROWS * EM * 2;  // 49152
constexpr uint32_t DMA_STAGE_CAP = 1024;            // candidates staged per workgroup (12 KiB beside the 144-KiB ring)
constexpr uint32_t DMA_STAGE_FLUSH_AT = 384;

// NW waves per workgroup (one workgroup per CU): 8 (two per SIMD, 32 queries each) or 4 (one per SIMD, 64 queries each:
// every A fragment read from LDS feeds two MFMAs, half the LDS traffic, no sharing of the matrix pipe)
// BF16: the tiles are a bf16 INDEX (ROW_BF16), the queries bf16, the contraction v_mfma_f32_32x32x16_bf16, no score scale
template <bool DENSE, int NW, bool BF16 = false>
__global__ __launch_bounds__(NW * 64) void scan_f16_dma_kernel(const unsigned char* __restrict__ xs, uint32_t n_rows,
                                                          uint32_t first_tile, uint32_t tile_stride, uint32_t n_tiles,
                                                          const half8* __restrict__ qh, int n_q,
                                                          const float* __restrict__ tau, uint32_t* __restrict__ cnt,
                                                          uint2* __restrict__ cand, float* __restrict__ dense) {
    constexpr int NT = NW * 64, PD = 8;
    constexpr float SC = BF16 ? 1.0f : SCORE_SCALE;
    constexpr int QG = 8 / NW;     // 32-query groups per wave
    constexpr int DPW = 48 / NW;   // DMA instructions per wave and tile
    // three SEPARATE LDS objects: the module-LDS lowering then tags their accesses with alias scopes and hipcc's
    // waitcnt insertion does not drain the in-flight LDS-DMA (vmcnt(0)) in front of reads of ANOTHER image
    __shared__ __attribute__((aligned(16))) unsigned char img0[DMA_TILE_BYTES];
    __shared__ __attribute__((aligned(16))) unsigned char img1[DMA_TILE_BYTES];
    __shared__ __attribute__((aligned(16))) unsigned char img2[DMA_TILE_BYTES];
    __shared__ uint32_t stage_q[DMA_STAGE_CAP];
    __shared__ float stage_s[DMA_STAGE_CAP];
    __shared__ uint32_t stage_r[DMA_STAGE_CAP];
    __shared__ uint32_t stage_n[4];
    if (!DENSE && threadIdx.x < 3) stage_n[threadIdx.x] = 0;
    auto flush = [&]() {
        uint32_t n = stage_n[0];
        if (n > DMA_STAGE_CAP) n = DMA_STAGE_CAP;
        for (uint32_t e = threadIdx.x; e < n; e += NT) {
            append_candidate(cnt, cand, stage_q[e], blockIdx.x % BATCH_CAND_SEGS, __builtin_bit_cast(uint32_t, stage_s[e]),
                             stage_r[e]);
        }
    };
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t r = lane & 31, h = lane >> 5;
    const int q0 = wave * (32 * QG) + (int)r;  // this lane's queries: q0 + 32*g
    const bool wave_has_queries = wave * (32 * QG) < n_q;

    half8 qf[QG][24];
    float tau_s[QG];
#pragma unroll
    for (int g = 0; g < QG; ++g) {
#pragma unroll
        for (int s = 0; s < 24; ++s) qf[g][s] = qh[(size_t)(q0 + 32 * g) * 48 + 2 * s + h];
        tau_s[g] = __builtin_inff();
        if (!DENSE && q0 + 32 * g < n_q) tau_s[g] = tau[q0 + 32 * g] * SC;
    }

    // DMA map: the shadow tile is stored in fragment order already (kernels.hpp, ROW_F16S), so the image is a linear
    // copy: instruction i of this wave moves bytes [(DPW*w+i)*1024, +1024) of the tile, 16 B per lane
    const uint32_t src_off0 = (uint32_t)(DPW * wave) * 1024u + (uint32_t)lane * 16u;

    const uint32_t G = gridDim.x;
    const uint32_t n_units = (n_tiles - blockIdx.x + G - 1) / G;
    const uint32_t last = n_units - 1;
    auto unit_row0 = [&](uint32_t t) { return (first_tile + (blockIdx.x + t * G) * tile_stride) * TILE_ROWS; };
    auto unit_slot0 = [&](uint32_t t) { return (blockIdx.x + t * G) * TILE_ROWS; };
    // The six per-lane source addresses of a tile are kept in registers of their own until the end of the iteration
    // (fake use below): hipcc otherwise recycles the address temporaries for the A ring / accumulator and then waits
    // for the whole DMA (vmcnt(0)) before the first MFMA overwrites them.
    auto dma = [&](uint32_t t, unsigned char* img, const unsigned char* (&gp)[DPW]) {  // tile t -> LDS image
        const unsigned char* base = xs + (size_t)unit_row0(t) * (EM * 2);
        unsigned char* dst = img + wave * (DPW * 1024);
#pragma unroll
        for (int i = 0; i < DPW; ++i) gp[i] = base + src_off0 + i * 1024;
#pragma unroll
        for (int i = 0; i < DPW; ++i)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gp[i],
                                             (__attribute__((address_space(3))) void*)(dst + i * 1024), 16, 0, 2 /* nt */);
    };
    auto keep = [&](const unsigned char* (&gp)[DPW]) {
#pragma unroll
        for (int i = 0; i < DPW; ++i) asm volatile("" ::"v"(gp[i]));
    };

    auto tail_slow = [&](const f32x16& acc, uint32_t row0, int qi, float tau_q) {
        uint32_t mask = 0;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const uint32_t row = row0 + (e & 3) + 8 * (e >> 2);
            mask |= (acc[e] > tau_q && row < n_rows) ? (1u << e) : 0u;
        }
        if (mask) {
            uint32_t pos = atomicAdd(&stage_n[0], (uint32_t)__popc(mask));  // LDS atomic
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                if (mask & (1u << e)) {
                    const float sc = acc[e] * (1.0f / SC);
                    const uint32_t row = row0 + (e & 3) + 8 * (e >> 2);
                    if (pos < DMA_STAGE_CAP) {
                        stage_q[pos] = (uint32_t)qi;
                        stage_s[pos] = sc;
                        stage_r[pos] = row;
                    } else {  // stage full (a burst: many queries hitting the same rows): append directly
                        append_candidate(cnt, cand, (uint32_t)qi, blockIdx.x % BATCH_CAND_SEGS,
                                         __builtin_bit_cast(uint32_t, sc), row);
                    }
                    ++pos;
                }
            }
        }
    };

    // prologue: tiles 0 and 1 on their way (a workgroup with one tile fetches it twice: no branch around the loads)
    // The query fragments must be complete BEFORE the loop as far as hipcc can tell (a use it can see): their first
    // real use is the first MFMA inside the loop, and a wait placed there is re-executed every iteration as vmcnt(0),
    // i.e. it would also drain the row DMA that is meant to stay in flight.
#pragma unroll
    for (int g = 0; g < QG; ++g) {
#pragma unroll
        for (int s = 0; s < 24; ++s) asm volatile("" ::"v"(qf[g][s]));
        asm volatile("" ::"v"(tau_s[g]));
    }
    const unsigned char* gp0[DPW];
    const unsigned char* gp1[DPW];
    dma(0, img0, gp0);
    dma(last < 1u ? last : 1u, img1, gp1);
    asm volatile("s_waitcnt vmcnt(%0)\n\ts_waitcnt lgkmcnt(0)\n\ts_barrier" ::"n"(DPW) : "memory");  // tile 0 landed
    keep(gp0);
    keep(gp1);

    uint32_t t = 0;
    // one tile: contract `rd` (tile t), DMA tile t+2 into `wr` (last read in iteration t-1, which every wave left
    // through the barrier)
    auto step = [&](uint32_t rd_off, unsigned char* wr) -> bool {
        if (t >= n_units) return false;
        const unsigned char* gp[DPW];
        dma(t + 2 < n_units ? t + 2 : last, wr, gp);
        if (wave_has_queries) {
#pragma unroll
            for (int sub = 0; sub < 2; ++sub) {
                // A fragments by inline-asm ds_read_b128 with hand-counted lgkmcnt: hipcc drains every in-flight
                // LDS-DMA (s_waitcnt vmcnt(0)) in front of any LDS read it can see, which would serialise the row stream
                // behind each contraction.  Ring of PD reads; before MFMA s at most PD-1 younger reads are outstanding.
                // fragment (sub, s) = bytes [(sub*24+s)*1024, +1024) of the image, lane-linear: conflict-free reads
                const uint32_t ad = rd_off + (uint32_t)lane * 16u;
                half8 a[PD];
#pragma unroll
                for (int d = 0; d < PD; ++d)
                    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(a[d]) : "v"(ad), "n"((sub * 24 + d) * 1024));
                f32x16 acc[QG];
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int s = 0; s < 24; ++s) {
                    // reads s .. min(s+PD,24)-1 are outstanding; the oldest must have landed
                    // (no "memory" clobber: hipcc treats such an asm as an LDS access and drains the DMA in front of it)
                    if (s + PD <= 24) asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(PD - 1));
                    else asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(0));
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int g = 0; g < QG; ++g) {
                        if (s == 0) {
                            f32x16 z;
#pragma unroll
                            for (int e = 0; e < 16; ++e) z[e] = 0.f;
                            acc[g] = mfma16<BF16>(a[0], qf[g][0], z);
                        } else {
                            acc[g] = mfma16<BF16>(a[s % PD], qf[g][s], acc[g]);
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    if (s + PD < 24)
                        asm volatile("ds_read_b128 %0, %1 offset:%2"
                                     : "=v"(a[s % PD])
                                     : "v"(ad), "n"((sub * 24 + s + PD) * 1024));
                    __builtin_amdgcn_sched_barrier(0);
                }
                const uint32_t row0 = unit_row0(t) + sub * 32 + 4 * h;
#pragma unroll
                for (int g = 0; g < QG; ++g) {
                    const int qi = q0 + 32 * g;
                    if (DENSE) {
                        if (qi < n_q) {
#pragma unroll
                            for (int e4 = 0; e4 < 4; ++e4) {
                                f32x4 o;
#pragma unroll
                                for (int e = 0; e < 4; ++e)
                                    o[e] = (row0 + e + 8 * e4) < n_rows ? acc[g][e4 * 4 + e] * (1.0f / SC) : NEG_INF;
                                *reinterpret_cast<f32x4*>(dense + (size_t)qi * BATCH_CAP + unit_slot0(t) + sub * 32 +
                                                          4 * h + 8 * e4) = o;
                            }
                        }
                    } else {
                        float mx = acc[g][0];
#pragma unroll
                        for (int e = 1; e < 16; ++e) mx = fmaxf(mx, acc[g][e]);
                        if (__any(mx > tau_s[g])) tail_slow(acc[g], row0, qi, tau_s[g]);
                    }
                }
            }
        }
        keep(gp);
        // publish: tile t+1 has landed (this wave's share), tile t+2 may stay in flight; candidate stage check
        // (the stage fill is read by asm as well: a visible LDS read would make hipcc drain the DMA of tile t+2)
        uint32_t fill = 0;
        if (!DENSE) {
            const uint32_t sn = (uint32_t)(size_t)(__attribute__((address_space(3))) uint32_t*)stage_n;
            asm volatile("s_waitcnt vmcnt(%2)\n\ts_waitcnt lgkmcnt(0)\n\ts_barrier\n\tds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)"
                         : "=v"(fill)
                         : "v"(sn), "n"(DPW)
                         : "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(%0)\n\ts_waitcnt lgkmcnt(0)\n\ts_barrier" ::"n"(DPW) : "memory");
        }
        if (!DENSE && __builtin_amdgcn_readfirstlane(fill) >= DMA_STAGE_FLUSH_AT) {
            flush();
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            if (threadIdx.x == 0) stage_n[0] = 0;
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        }
        ++t;
        return true;
    };
    const uint32_t o0 = (uint32_t)(size_t)(__attribute__((address_space(3))) unsigned char*)img0;
    const uint32_t o1 = (uint32_t)(size_t)(__attribute__((address_space(3))) unsigned char*)img1;
    const uint32_t o2 = (uint32_t)(size_t)(__attribute__((address_space(3))) unsigned char*)img2;
    for (;;) {
        if (!step(o0, img2)) break;
        if (!step(o1, img0)) break;
        if (!step(o2, img1)) break;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the clamped tail DMAs must not outlive the workgroup's LDS
    if (!DENSE) {
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        flush();
    }
}

// ------------------------------------------------------------------------------------------------
// scan_f16_pipe_kernel — the same filter with ONE wave per SIMD (256 threads), 64 queries per wave: every A fragment
// read from LDS feeds two MFMAs, so the LDS pipe (128 B/clk per CU, exactly the rate at which eight 32-query waves
// consume fragments) runs at half load and the matrix pipe is shared with nobody.  With one wave per SIMD nothing hides
// a bubble, so the tile loop is one software pipeline:
//   * the ring of PD fragment reads never drains: the last PD k-steps of a sub-tile already fetch the first fragments
//     of the next one (across the tile boundary: from the next LDS image);
//   * two accumulator sets: the threshold test of a sub-tile runs after the first MFMAs of the next one are issued;
//   * two barriers per tile, neither waits for LDS reads: P1 (k-step 2: every wave has left the previous tile, its
//     image may be overwritten -> issue the DMA of tile t+2) and P2 (k-step 39: s_waitcnt vmcnt -> this wave's share
//     of tile t+1 has landed; after the barrier all of it has, and k-step 40 starts reading it).  A DMA has 1.7 tile
//     times to land;
//   * candidates are staged per WAVE (private LDS region and counter, flushed by the wave itself): no LDS atomics,
//     no fill snapshot, no flush barriers.
// Query groups are dealt round-robin (wave w: groups w and w+4), so up to 128 queries use one group per wave.
// ------------------------------------------------------------------------------------------------
constexpr uint32_t PIPE_WCAP = 256;  // staged candidates per wave (4 x 3 KiB beside the 144-KiB ring)

// DBG (timing experiments only, results are wrong): 1 = no DMA, 2 = no barriers, 4 = no threshold tests, 8 = no LDS reads
template <bool DENSE, int DBG = 0, bool BF16 = false>
__global__ __launch_bounds__(256) void scan_f16_pipe_kernel(const unsigned char* __restrict__ xs, uint32_t n_rows,
                                                           uint32_t first_tile, uint32_t tile_stride, uint32_t n_tiles,
                                                           const half8* __restrict__ qh, int n_q,
                                                           const float* __restrict__ tau, uint32_t* __restrict__ cnt,
                                                           uint2* __restrict__ cand, float* __restrict__ dense) {
    constexpr int NW = 4, PD = 8, DPW = 48 / NW;
    constexpr float SC = BF16 ? 1.0f : 65536.0f;  // score scale of the f16 path (rows and queries x 2^8)
    // ring of three tile images; which one is read / filled rotates at run time (the fast path has no LDS access
    // hipcc can see — the fragment reads are inline asm — so it has no reason to drain the DMA)
    __shared__ __attribute__((aligned(16))) unsigned char img[3 * DMA_TILE_BYTES];
    // candidate stage: [query | score bits | row] planes of NW * PIPE_WCAP words; written by inline asm (a store hipcc
    // can see makes it wait for every DMA in flight first: vmcnt(0), ~2 us, on ~8 % of the sub-tiles)
    constexpr uint32_t PLANE = NW * PIPE_WCAP * 4;
    __shared__ uint32_t stage[3 * NW * PIPE_WCAP];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t r = lane & 31, h = lane >> 5;
    const int qg0 = wave * 32 + (int)r, qg1 = (wave + NW) * 32 + (int)r;  // this lane's query in group 0 / 1
    const bool live0 = wave * 32 < n_q, live1 = (wave + NW) * 32 < n_q;    // wave-uniform

    half8 qf[2][24];
    float tau_s[2];
#pragma unroll
    for (int s = 0; s < 24; ++s) {
        qf[0][s] = qh[(size_t)qg0 * 48 + 2 * s + h];
        qf[1][s] = qh[(size_t)qg1 * 48 + 2 * s + h];
    }
    tau_s[0] = tau_s[1] = __builtin_inff();
    if (!DENSE && qg0 < n_q) tau_s[0] = tau[qg0] * SC;
    if (!DENSE && qg1 < n_q) tau_s[1] = tau[qg1] * SC;

    const uint32_t src_off0 = (uint32_t)(DPW * wave) * 1024u + (uint32_t)lane * 16u;
    const uint32_t G = gridDim.x;
    const uint32_t n_units = (n_tiles - blockIdx.x + G - 1) / G;
    const uint32_t last = n_units - 1;
    auto unit_row0 = [&](uint32_t t) { return (first_tile + (blockIdx.x + t * G) * tile_stride) * TILE_ROWS; };
    auto unit_slot0 = [&](uint32_t t) { return (blockIdx.x + t * G) * TILE_ROWS; };
    // 12 DMA instructions per wave and tile = 3 address registers x 4 immediate offsets (the instruction offset moves
    // the global and the LDS address alike)
    constexpr int NGP = DPW / 4;
    auto dma = [&](uint32_t t, uint32_t image, const unsigned char* (&gp)[NGP]) __attribute__((always_inline)) {
        const unsigned char* base = xs + (size_t)unit_row0(t) * (EM * 2) + src_off0;
        unsigned char* dst = img + image * DMA_TILE_BYTES + wave * (DPW * 1024);
#pragma unroll
        for (int j = 0; j < NGP; ++j) gp[j] = base + j * 4096;
#pragma unroll
        for (int j = 0; j < NGP; ++j) {
            const __attribute__((address_space(1))) void* g = (const __attribute__((address_space(1))) void*)gp[j];
            __attribute__((address_space(3))) void* l = (__attribute__((address_space(3))) void*)(dst + j * 4096);
            __builtin_amdgcn_global_load_lds(g, l, 16, 0, 2 /* nt */);
            __builtin_amdgcn_global_load_lds(g, l, 16, 1024, 2);
            __builtin_amdgcn_global_load_lds(g, l, 16, 2048, 2);
            __builtin_amdgcn_global_load_lds(g, l, 16, 3072, 2);
        }
    };
    // the same, one instruction at a time (spread over the k-steps of a tile so that the matrix pipe never waits for a
    // block of address arithmetic and 12 back-to-back issues)
    auto dma_setup = [&](uint32_t t, const unsigned char* (&gp)[NGP]) __attribute__((always_inline)) {
        const unsigned char* base = xs + (size_t)unit_row0(t) * (EM * 2) + src_off0;
#pragma unroll
        for (int j = 0; j < NGP; ++j) gp[j] = base + j * 4096;
    };
    auto dma_one = [&](auto i_c, uint32_t image, const unsigned char* (&gp)[NGP]) __attribute__((always_inline)) {
        constexpr int I = decltype(i_c)::value, J = I / 4, O = (I % 4) * 1024;
        unsigned char* dst = img + image * DMA_TILE_BYTES + wave * (DPW * 1024) + J * 4096;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gp[J],
                                         (__attribute__((address_space(3))) void*)dst, 16, O, 2 /* nt */);
    };
    auto keep = [&](const unsigned char* (&gp)[NGP]) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < NGP; ++j) asm volatile("" ::"v"(gp[j]));
    };

    // ---- candidate staging, private to the wave ----
    uint32_t wpos = 0;  // wave-uniform fill of this wave's region
    auto flush_wave = [&]() __attribute__((always_inline)) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the asm stores below
        // the region is 4 entries per lane: all four slot reservations in flight together (each is a round trip to the
        // memory-side atomic unit, ~2 us when 1024 waves hammer the 256 counters of a small index)
        uint32_t q_[PIPE_WCAP / 64], slot[PIPE_WCAP / 64];
        const uint32_t seg = blockIdx.x % BATCH_CAND_SEGS;  // (a query belongs to ONE wave of every workgroup)
#pragma unroll
        for (int j = 0; j < (int)(PIPE_WCAP / 64); ++j) {
            const uint32_t e = lane + 64u * j;
            q_[j] = e < wpos ? stage[wave * PIPE_WCAP + e] : 0u;
        }
#pragma unroll
        for (int j = 0; j < (int)(PIPE_WCAP / 64); ++j) {
            const uint32_t e = lane + 64u * j;
            slot[j] = 0xFFFFFFFFu;
            if (e < wpos) slot[j] = (DBG & 16) ? e : atomicAdd(&cnt[q_[j] * BATCH_CAND_SEGS + seg], 1u);
        }
#pragma unroll
        for (int j = 0; j < (int)(PIPE_WCAP / 64); ++j) {
            const uint32_t e = lane + 64u * j;
            if (slot[j] < SEG_CAP)
                cand[(size_t)q_[j] * BATCH_CAP + seg * SEG_CAP + slot[j]] =
                    make_uint2(stage[NW * PIPE_WCAP + wave * PIPE_WCAP + e], stage[2 * NW * PIPE_WCAP + wave * PIPE_WCAP + e]);
        }
        wpos = 0;
    };
    const uint32_t stage_base = (uint32_t)(size_t)(__attribute__((address_space(3))) uint32_t*)stage;
    // Slow path of the threshold test (some lane of the wave holds a score above its query's threshold): one ballot
    // per accumulator element — most find nothing and cost two compares — and a scalar walk over the hits: read the
    // score from its lane, append (query, score, row) to the wave's stage.  row0: first row of the sub-tile;
    // q_first: query of lane 0 in this group; tau_lane: the lane's threshold (x 2^16).
    auto tail_slow = [&](const f32x16& acc, uint32_t row0, uint32_t q_first, float tau_lane) __attribute__((always_inline)) {
        const uint32_t lim = n_rows > row0 + 4 * h ? n_rows - row0 - 4 * h : 0u;  // rows of this lane's column that exist
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const float ae = acc[e];  // (a scalar copy first: see the bit_cast note in scan_kernels.hip)
            const uint32_t roff = (uint32_t)((e & 3) + 8 * (e >> 2));
            unsigned long long m = __ballot(ae > tau_lane && roff < lim);
            while (m) {
                const int l = __builtin_ctzll(m);
                m &= m - 1;
                const float sc = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, ae), l)) *
                                 (1.0f / SC);
                const uint32_t qi = q_first + (uint32_t)(l & 31);
                const uint32_t row = row0 + roff + 4u * (uint32_t)(l >> 5);
                if (wpos >= PIPE_WCAP) flush_wave();
                if (lane == 0) {
                    const uint32_t pa = stage_base + (wave * PIPE_WCAP + wpos) * 4u;
                    asm volatile("ds_write_b32 %0, %1\n\tds_write_b32 %0, %2 offset:%4\n\tds_write_b32 %0, %3 offset:%5"
                                 :
                                 : "v"(pa), "v"(qi), "v"(__builtin_bit_cast(uint32_t, sc)), "v"(row), "n"(PLANE), "n"(2 * PLANE));
                }
                ++wpos;
            }
        }
    };
    // two accumulator sets (even / odd sub-tiles) x two query groups; mx: running maxima of the set under test
    f32x16 acc[2][2];
    float mx[2] = {0.f, 0.f};
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[1][0][e] = acc[1][1][e] = 0.f;  // (read by the max slices of the first tile)
    // threshold test / dense store of one finished 32-row sub-tile (accumulator set `set`, NL live query groups)
    auto tail = [&](auto set_c, auto nl_c, uint32_t row_base, uint32_t slot_base) __attribute__((always_inline)) {
        constexpr int SET = decltype(set_c)::value, NL = decltype(nl_c)::value;
        const uint32_t row0 = row_base + 4 * h;
#pragma unroll
        for (int g = 0; g < NL; ++g) {
            const int qi = g == 0 ? qg0 : qg1;
            if (DENSE) {
                if (qi < n_q) {
#pragma unroll
                    for (int e4 = 0; e4 < 4; ++e4) {
                        f32x4 o;
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            o[e] = (row0 + e + 8 * e4) < n_rows ? acc[SET][g][e4 * 4 + e] * (1.0f / SC) : NEG_INF;
                        *reinterpret_cast<f32x4*>(dense + (size_t)qi * BATCH_CAP + slot_base + 4 * h + 8 * e4) = o;
                    }
                }
            } else {
                float mx = acc[SET][g][0];
#pragma unroll
                for (int e = 1; e < 16; ++e) mx = fmaxf(mx, acc[SET][g][e]);
                if (__any(mx > tau_s[g])) tail_slow(acc[SET][g], row_base, (uint32_t)((wave + NW * g) * 32), tau_s[g]);
            }
        }
    };

#pragma unroll
    for (int g = 0; g < 2; ++g) {
#pragma unroll
        for (int s = 0; s < 24; ++s) asm volatile("" ::"v"(qf[g][s]));
        asm volatile("" ::"v"(tau_s[g]));
    }
    const uint32_t o0 = (uint32_t)(size_t)(__attribute__((address_space(3))) unsigned char*)img;

    // prologue: tiles 0 and 1 on their way, tile 0 landed everywhere, ring primed with its first PD fragments
    const unsigned char* gp0[NGP];
    const unsigned char* gp1[NGP];
    dma(0, 0, gp0);
    dma(last < 1u ? last : 1u, 1, gp1);
    asm volatile("s_waitcnt vmcnt(%0)\n\ts_waitcnt lgkmcnt(0)\n\ts_barrier" ::"n"(DPW) : "memory");
    keep(gp0);
    keep(gp1);
    half8 a[PD];
    {
        const uint32_t ad = o0 + (uint32_t)lane * 16u;
#pragma unroll
        for (int d = 0; d < PD; ++d)
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(a[d]) : "v"(ad), "n"(d * 1024));
    }

    uint32_t t = 0;
    typedef std::integral_constant<int, 0> C0;
    typedef std::integral_constant<int, 1> C1;
    typedef std::integral_constant<int, 2> C2;
    // one tile: image `rd` holds tile t (its first PD fragments are already in the ring), `nx` tile t+1, `wr` gets t+2;
    // NL = live query groups of this wave (0: the wave only moves rows)
    auto step = [&](auto nl_c, uint32_t rd, uint32_t nx, uint32_t wr) __attribute__((always_inline)) {
        constexpr int NL = decltype(nl_c)::value;
        const uint32_t ad = o0 + rd * DMA_TILE_BYTES + (uint32_t)lane * 16u;
        const uint32_t adn = o0 + nx * DMA_TILE_BYTES + (uint32_t)lane * 16u;
        const unsigned char* gp[NGP];
#pragma unroll
        for (int f = 0; f < 48; ++f) {
            const int sub = f / 24, s = f % 24;
            if (NL > 0) {
                if (!(DBG & 8)) asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(PD - 1));  // fragment f has landed (younger reads may be out)
                __builtin_amdgcn_sched_barrier(0);
                // MFMAs by inline asm so that the operands stay where they are: accumulators in VGPRs (the test reads
                // them with plain VALU ops), group 0's query fragments in VGPRs, group 1's in AGPRs read directly as
                // SrcB.  (Left to hipcc, group 1 lives in AGPRs and is COPIED out before every MFMA — and a
                // v_accvgpr_read waits for the MFMA in flight: measured 127 clk per k-step instead of 64.)
#define DAWN_MFMA_ZERO(T, D, A, B, CB) asm volatile("v_mfma_f32_32x32x16_" T " %0, %1, %2, 0" : "=&v"(D) : "v"(A), CB(B))
#define DAWN_MFMA_ACC(T, D, A, B, CB) asm volatile("v_mfma_f32_32x32x16_" T " %0, %1, %2, %0" : "+v"(D) : "v"(A), CB(B))
                if (s == 0) {
                    if (BF16) DAWN_MFMA_ZERO("bf16", acc[sub][0], a[f % PD], qf[0][0], "v");
                    else DAWN_MFMA_ZERO("f16", acc[sub][0], a[f % PD], qf[0][0], "v");
                    if (NL > 1) {
                        if (BF16) DAWN_MFMA_ZERO("bf16", acc[sub][1], a[f % PD], qf[1][0], "a");
                        else DAWN_MFMA_ZERO("f16", acc[sub][1], a[f % PD], qf[1][0], "a");
                    }
                } else {
                    if (BF16) DAWN_MFMA_ACC("bf16", acc[sub][0], a[f % PD], qf[0][s], "v");
                    else DAWN_MFMA_ACC("f16", acc[sub][0], a[f % PD], qf[0][s], "v");
                    if (NL > 1) {
                        if (BF16) DAWN_MFMA_ACC("bf16", acc[sub][1], a[f % PD], qf[1][s], "a");
                        else DAWN_MFMA_ACC("f16", acc[sub][1], a[f % PD], qf[1][s], "a");
                    }
                }
#undef DAWN_MFMA_ZERO
#undef DAWN_MFMA_ACC
                __builtin_amdgcn_sched_barrier(0);
                // The threshold test of the sub-tile finished 24 k-steps ago (the other accumulator set), two or three
                // VALU instructions per k-step so that they issue in the shadow of this step's MFMAs — a wave issues
                // in order, a block of 40 VALU instructions is 150 clk without an MFMA in the queue.  Starts 2 k-steps
                // after that set's last MFMA (results complete; hipcc pads no hazards for asm MFMAs: s_nop).
                if (!DENSE && !(DBG & 4)) {
                    constexpr int FIRST = 2;
                    if (s == FIRST) asm volatile("s_nop 7");
                    if (s >= FIRST && s < FIRST + 8) {
                        const int j = s - FIRST;
#pragma unroll
                        for (int g = 0; g < NL; ++g) {
                            mx[g] = j == 0 ? fmaxf(acc[1 - sub][g][0], acc[1 - sub][g][1])
                                           : fmaxf(fmaxf(mx[g], acc[1 - sub][g][2 * j]), acc[1 - sub][g][2 * j + 1]);
                            asm volatile("" : "+v"(mx[g]));  // computed HERE (hipcc otherwise sinks the slices to the test)
                        }
                    }
                    if (s == FIRST + 8 && (sub == 1 || t > 0)) {
                        bool hit = mx[0] > tau_s[0];
                        if (NL > 1) hit = hit || mx[1] > tau_s[1];
                        if (__any(hit)) {
                            const uint32_t rb = sub == 1 ? unit_row0(t) : unit_row0(t - 1) + 32;
#pragma unroll
                            for (int g = 0; g < NL; ++g)
                                if (__any(mx[g] > tau_s[g]))
                                    tail_slow(acc[1 - sub][g], rb, (uint32_t)((wave + NW * g) * 32), tau_s[g]);
                        }
                    }
                }
                if (DENSE && s == 2 && (sub == 1 || t > 0)) {
                    asm volatile("s_nop 7");
                    if (sub == 1) tail(C0(), nl_c, unit_row0(t), unit_slot0(t));
                    else tail(C1(), nl_c, unit_row0(t - 1) + 32, unit_slot0(t - 1) + 32);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            if (f == 12) {  // P1: every wave has left tile t-1, its image may be overwritten
                if (!(DBG & 2)) asm volatile("s_barrier");
                __builtin_amdgcn_sched_barrier(0);
                dma_setup(t + 2 < n_units ? t + 2 : last, gp);
            }
            if (!(DBG & 1)) {  // DMA of tile t+2: one instruction per k-step
                if (f == 13) dma_one(std::integral_constant<int, 0>(), wr, gp);
                if (f == 14) dma_one(std::integral_constant<int, 1>(), wr, gp);
                if (f == 15) dma_one(std::integral_constant<int, 2>(), wr, gp);
                if (f == 16) dma_one(std::integral_constant<int, 3>(), wr, gp);
                if (f == 17) dma_one(std::integral_constant<int, 4>(), wr, gp);
                if (f == 18) dma_one(std::integral_constant<int, 5>(), wr, gp);
                if (f == 19) dma_one(std::integral_constant<int, 6>(), wr, gp);
                if (f == 20) dma_one(std::integral_constant<int, 7>(), wr, gp);
                if (f == 21) dma_one(std::integral_constant<int, 8>(), wr, gp);
                if (f == 22) dma_one(std::integral_constant<int, 9>(), wr, gp);
                if (f == 23) dma_one(std::integral_constant<int, 10>(), wr, gp);
                if (f == 24) dma_one(std::integral_constant<int, 11>(), wr, gp);
            }
            if (f == 39) {  // P2: tile t+1 has landed (this wave's share; after the barrier all of it)
                __builtin_amdgcn_sched_barrier(0);
                if (DBG & 2) {
                    if (!(DBG & 1)) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DPW));
                } else if (DENSE) asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier");  // (dense stores share vmcnt)
                else if (DBG & 1) asm volatile("s_barrier");
                else asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(DPW));
            }
            if (NL > 0 && !(DBG & 8)) {
                __builtin_amdgcn_sched_barrier(0);
                if (f + PD < 48)
                    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(a[f % PD]) : "v"(ad), "n"((f + PD) * 1024));
                else
                    asm volatile("ds_read_b128 %0, %1 offset:%2"
                                 : "=v"(a[f % PD])
                                 : "v"(adn), "n"((f + PD < 48 ? 0 : f + PD - 48) * 1024));
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        keep(gp);
        ++t;
    };
    auto run = [&](auto nl_c) __attribute__((always_inline)) {
        uint32_t rd = 0, nx = 1, wr = 2;  // image indices: read, next, fill
        while (t < n_units) {
            step(nl_c, rd, nx, wr);
            const uint32_t o = rd;
            rd = nx;
            nx = wr;
            wr = o;
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_nop 15\n\ts_nop 15");  // the ring's look-ahead reads; the last MFMAs
        __builtin_amdgcn_sched_barrier(0);
        if (decltype(nl_c)::value > 0 && !(DBG & 4)) tail(C1(), nl_c, unit_row0(last) + 32, unit_slot0(last) + 32);
    };
    if (live1) run(C2());
    else if (live0) run(C1());
    else run(C0());
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the clamped tail DMAs must not outlive the workgroup's LDS
    if (!DENSE) flush_wave();
}

// ------------------------------------------------------------------------------------------------
// per-query top-64 of an unsorted candidate set (block of 1024 threads); result in wave 0, descending
// ------------------------------------------------------------------------------------------------
// DENSE: `count` scores dense_q[0..count); else: wave w takes segment w of the query's candidate buffer (seg_cnt_q[w]
// entries, clamped to SEG_CAP) — the kernels run 16 waves = BATCH_CAND_SEGS.
// keep: only the best `keep` (<= 64) entries of the result are needed (tau_select wants the m-th largest): after its
// first chunk a wave then inserts just the elements that beat its keep-th best (a handful per chunk) instead of
// sorting and merging every chunk of 64.
// excl: only entries strictly worse than (ex_s, ex_p) in (score desc, row asc) order count (the certificate's deeper rounds)
template <bool DENSE>
__device__ __forceinline__ void block_top64(const float* __restrict__ dense_q, const uint2* __restrict__ cand_q,
                                            const uint32_t* __restrict__ seg_cnt_q, uint32_t count, float& s, uint32_t& p,
                                            float (*sh_s)[LIST], uint32_t (*sh_p)[LIST], int wave, int lane, int nwaves,
                                            uint32_t keep = LIST, bool excl = false, float ex_s = 0.f, uint32_t ex_p = 0u) {
    s = NEG_INF;
    p = NO_POS;
    bool first = true;
    if (!DENSE) {
        count = seg_cnt_q[wave];
        if (count > SEG_CAP) count = SEG_CAP;
        cand_q += (size_t)wave * SEG_CAP;
    }
    const uint32_t n_chunks = (count + 63u) >> 6;
    // a wave's chunks are loaded eight at a time, all in flight together (one memory round trip per eight chunks: a loop of
    // load -> process per chunk was eight dependent round trips, 12 of tau_select's 17 us on a sample of 8192 scores)
    constexpr int GRP = 8;
    const uint32_t cstep = DENSE ? (uint32_t)nwaves : 1u;
    for (uint32_t c0 = DENSE ? wave : 0; c0 < n_chunks; c0 += GRP * cstep) {
        float dv[GRP];
        uint32_t rv[GRP];
#pragma unroll
        for (int j = 0; j < GRP; ++j) {
            const uint32_t e = (c0 + j * cstep) * 64u + lane;
            dv[j] = POS_INF;  // key = -score: ascending sort = descending score, ties -> lower row
            rv[j] = NO_POS;
            if (c0 + j * cstep < n_chunks && e < count) {
                if (DENSE) {
                    const float sc = dense_q[e];
                    if (sc > NEG_INF) {
                        dv[j] = -sc;
                        rv[j] = e;
                    }
                } else {
                    const uint2 v = cand_q[e];
                    dv[j] = -__builtin_bit_cast(float, v.x);
                    rv[j] = v.y;
                }
            }
        }
#pragma unroll
        for (int j = 0; j < GRP; ++j) {
            if (c0 + j * cstep >= n_chunks) break;  // wave-uniform
            float d = dv[j];
            uint32_t row = rv[j];
            if (excl && row != NO_POS && !better(ex_s, ex_p, -d, row)) {
                d = POS_INF;
                row = NO_POS;
            }
            if (!first && keep <= 32) {
                const float bar = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, s), (int)keep - 1));
                unsigned long long hits = __ballot(-d > bar);  // (fillers: d = +inf -> never)
                while (hits) {
                    const int l = __builtin_ctzll(hits);
                    hits &= hits - 1;
                    const float sc = -__builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, d), l));
                    const uint32_t rw = (uint32_t)__builtin_amdgcn_readlane((int)row, l);
                    wave_insert(s, p, sc, rw, lane);
                }
                continue;
            }
            first = false;
            sort64_asc(d, row, lane);
            // merge64 wants the other list reversed: lane i <- other[63 - i]
            const float os = -__shfl(d, 63 - lane);
            const uint32_t op = __shfl(row, 63 - lane);
            merge64(s, p, os, op, lane);
        }
    }
    block_merge(s, p, sh_s, sh_p, wave, lane, nwaves);
}

// tau[b] = m-th largest score of query b's sample (dense scores or appended candidates); fewer than m
// samples -> the smallest one; none -> -inf.  Rows scoring <= tau are NOT appended by the next pass.
// Leaves the query's segment counters at zero for the append pass that follows (no memset launch in between).
template <bool DENSE>
__global__ __launch_bounds__(1024) void tau_select_kernel(const float* __restrict__ dense,
                                                         const uint2* __restrict__ cand,
                                                         uint32_t* __restrict__ cnt, uint32_t dense_count,
                                                         uint32_t m, float* __restrict__ tau, uint32_t* __restrict__ pool) {
    // The m-th largest of a query's <= 8192 sample scores by radix selection on the order-preserving key (three digits of
    // 11 + 11 + 10 bits, LDS histograms, one block scan per digit): every thread keeps its <= 8 values in registers and
    // touches memory once.  (Round 2 kept a sorted 64-entry list per wave — a bitonic sort, up to eight insert rounds and
    // a four-level block merge, ~1200 instructions per wave, sixteen waves per CU: 15-20 us; this is the same value.)
    __shared__ uint32_t hist[2048];
    __shared__ uint32_t misc[32];  // [0] crossing digit [1] entries above it; [16..31] scan scratch
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t tid = threadIdx.x;
    const int b = blockIdx.x;
    constexpr int PER = BATCH_CAP / 1024;  // 8
    uint32_t key[PER];
    uint32_t have = 0;
    if (DENSE) {
        const uint32_t count = dense_count < (uint32_t)BATCH_CAP ? dense_count : (uint32_t)BATCH_CAP;
        const float* dq = dense + (size_t)b * BATCH_CAP;
#pragma unroll
        for (int j = 0; j < PER; ++j) {
            const uint32_t e = tid + 1024u * j;
            const float v = e < count ? dq[e] : NEG_INF;
            key[j] = v > NEG_INF ? order_key(v) : 0u;  // 0 = no entry (order_key of a score is never 0: that is -NaN)
        }
    } else {
        const uint2* cq = cand + (size_t)b * BATCH_CAP;
        const uint32_t* cn = cnt + (size_t)b * BATCH_CAND_SEGS;
        // (all sixteen loads requested before any is used — a count, then the entry it guards, per j was `global_load; s_waitcnt
        // vmcnt(0)` sixteen times in the ISA; they hit L2: 8.8 -> 8.4 us per launch.  Entries past a segment's count are read — the
        // buffer holds BATCH_CAP of them per query — and discarded)
        uint32_t c8[PER], v8[PER];
#pragma unroll
        for (int j = 0; j < PER; ++j) c8[j] = cn[(tid + 1024u * j) / SEG_CAP];
#pragma unroll
        for (int j = 0; j < PER; ++j) v8[j] = cq[tid + 1024u * j].x;
#pragma unroll
        for (int j = 0; j < PER; ++j) {
            const uint32_t jj = (tid + 1024u * j) % SEG_CAP;
            const uint32_t c = c8[j] > SEG_CAP ? SEG_CAP : c8[j];
            key[j] = jj < c ? order_key(__builtin_bit_cast(float, v8[j])) : 0u;
        }
    }
#pragma unroll
    for (int j = 0; j < PER; ++j) have += key[j] != 0u;
    // total number of entries (block sum)
    const uint32_t incl_have = block_incl_scan_u32(have, misc + 16, wave, lane);
    if (tid == 1023) misc[2] = incl_have;
    __syncthreads();
    const uint32_t total = misc[2];
    __syncthreads();
    float t = NEG_INF;
    if (total > 0) {
        uint32_t rank = m <= total ? m : total;  // 1-based from the top
        uint32_t prefix = 0;                      // the digits fixed so far, right-aligned
        // digit p covers key bits [shift, shift + bits)
        const int shifts[3] = {21, 10, 0}, nbits[3] = {11, 11, 10};
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            const int sh = shifts[p];
            const uint32_t mask = (1u << nbits[p]) - 1u;
            hist[2 * tid] = 0;
            hist[2 * tid + 1] = 0;
            __syncthreads();
#pragma unroll
            for (int j = 0; j < PER; ++j) {
                const uint32_t kk = key[j];
                const bool in = kk != 0u && (p == 0 || (kk >> (sh + nbits[p])) == prefix);
                if (in) atomicAdd(&hist[(kk >> sh) & mask], 1u);
            }
            __syncthreads();
            // digits in descending order: thread t owns digits 2047 - 2t and 2046 - 2t (zero counts beyond the digit's range)
            const uint32_t c0 = hist[2047 - 2 * tid], c1 = hist[2046 - 2 * tid];
            const uint32_t incl = block_incl_scan_u32(c0 + c1, misc + 16, wave, lane);
            const uint32_t excl = incl - (c0 + c1);
            if (excl < rank && excl + c0 >= rank) {
                misc[0] = 2047 - 2 * tid;
                misc[1] = excl;
            } else if (excl + c0 < rank && incl >= rank) {
                misc[0] = 2046 - 2 * tid;
                misc[1] = excl + c0;
            }
            __syncthreads();
            prefix = (prefix << nbits[p]) | misc[0];
            rank -= misc[1];
            __syncthreads();
        }
        t = key_to_float(prefix);
    }
    if (tid == 0) tau[b] = t;
    // (every thread read its segments' counts at the top)
    if (tid < BATCH_CAND_SEGS) cnt[(size_t)b * BATCH_CAND_SEGS + tid] = 0u;
    // ... and the chunk counters of the int8 append pass's dynamically assigned tail (scan_i8_pipe16_kernel)
    if (b == 0 && tid < 32 && pool != nullptr) pool[tid] = 0u;
}

// Final: shortlist = top-64 candidates by filter score; exact rescore in the reference order; certificate.
// Rows outside the shortlist scored <= m: the 64th candidate score if there are >= 64 candidates (every
// candidate beat tau), else tau itself (DENSE: every row is a candidate, m = 64th score).
template <bool DENSE, int RT>
__global__ __launch_bounds__(1024) void select_rescore_kernel(
    const void* __restrict__ x, const uint64_t* __restrict__ ids, uint32_t n_rows, const float* __restrict__ q,
    const float* __restrict__ dense, const uint2* __restrict__ cand, const uint32_t* __restrict__ cnt,
    const float* __restrict__ tau, uint32_t k, uint64_t* __restrict__ out_labels, float* __restrict__ out_dist,
    uint32_t* __restrict__ out_found, uint32_t* __restrict__ out_flags, int force_fallback, float eps, int rerun) {
    if (rerun && out_flags[blockIdx.x] != FLAG_FALLBACK) return;  // (block-uniform: a second pass looks at the flagged queries only)
    __shared__ float sh_s[16][LIST];
    __shared__ uint32_t sh_p[16][LIST];
    __shared__ uint32_t sh_rows[LIST];
    extern __shared__ __attribute__((aligned(16))) unsigned char rescore_stage[];  // RescoreStage<RT>::BYTES
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int b = blockIdx.x;
    const float q_val = threadIdx.x < EM ? q[(size_t)b * EM + threadIdx.x] : 0.f;  // (used after the selection: block_exact_dots)
    // candidates of this query: DENSE every row; else the segments' counts (a segment past its capacity dropped
    // candidates: the query goes to the exact pass)
    uint32_t count = n_rows;
    bool overflow = false;
    if (!DENSE) {
        count = 0;
        // (no short circuit — `overflow || ...` made every count's load wait for the one before it —: one s_load_dwordx16)
        uint32_t cs[BATCH_CAND_SEGS];
#pragma unroll
        for (int sg = 0; sg < BATCH_CAND_SEGS; ++sg) cs[sg] = cnt[(size_t)b * BATCH_CAND_SEGS + sg];
#pragma unroll
        for (int sg = 0; sg < BATCH_CAND_SEGS; ++sg) {
            overflow |= cs[sg] > SEG_CAP;
            count += cs[sg] > SEG_CAP ? SEG_CAP : cs[sg];
        }
    }
    __shared__ uint32_t sh_ctl[4];
    auto select = [&](bool first, float ex_s, uint32_t ex_p, float& s, uint32_t& p) {
        block_top64<DENSE>(dense + (size_t)b * BATCH_CAP, cand + (size_t)b * BATCH_CAP, cnt + (size_t)b * BATCH_CAND_SEGS, count,
                           s, p, sh_s, sh_p, wave, lane, 16, LIST, !first, ex_s, ex_p);
    };
    // rows that never became candidates scored <= tau in the pass (DENSE: every row is a candidate)
    const float base = DENSE ? NEG_INF : tau[b];
    const uint32_t found = n_rows < k ? n_rows : k;
    float bs;
    uint32_t bp;
    bool heavy;
    const uint32_t flag = certify_rounds<RT>(select, base, !overflow, n_rows, found, eps, force_fallback, q_val, x,
                                             rescore_stage, sh_rows, sh_ctl, wave, lane, bs, bp, heavy);
    if (wave == 0) {
        if ((uint32_t)lane < found) {
            // (slots without a candidate read as "no threshold" to the ladder behind a flag: scan_bounded.hip)
            out_labels[(size_t)b * k + lane] = bp != NO_POS ? ids[bp] : 0ull;
            out_dist[(size_t)b * k + lane] = bp != NO_POS ? -bs : POS_INF;
        }
        if (lane == 0) {
            out_found[b] = found;
            out_flags[b] = (rerun && flag != FLAG_FALLBACK) ? FLAG_RERUN : flag;
        }
    }
    if (!heavy) return;
    __syncthreads();

    // ---- second chance (wave_topk.hpp): every row scoring above tau is a candidate (DENSE: every row is one); ALL of them
    // (up to the 8192 the buffer holds) are rescored exactly, 1024 per round
    const float* dense_q = dense + (size_t)b * BATCH_CAP;
    const uint2* cand_q = cand + (size_t)b * BATCH_CAP;
    const uint32_t* cnt_q = cnt + (size_t)b * BATCH_CAND_SEGS;
    auto load = [&](uint32_t e, float& sc, uint32_t& row) {
        if (DENSE) {
            if (e >= n_rows) return false;
            sc = dense_q[e];
            row = e;
            return sc > NEG_INF;
        }
        const uint32_t sg = e / SEG_CAP, j = e % SEG_CAP;
        if (j >= cnt_q[sg]) return false;
        const uint2 v = cand_q[e];
        sc = __builtin_bit_cast(float, v.x);
        row = v.y;
        return true;
    };
    float s2;
    uint32_t p2;
    static_assert(RescoreStage<RT>::BYTES >= (2048 + SECOND_CHANCE_K_ALL + 32) * 4, "second-chance scratch");
    const bool ok = second_chance<RT>(load, DENSE ? (n_rows < (uint32_t)BATCH_CAP ? n_rows : (uint32_t)BATCH_CAP) : (uint32_t)BATCH_CAP,
                                      DENSE ? NEG_INF : tau[b], q + (size_t)b * EM, x, found, eps, rescore_stage, sh_s, sh_p,
                                      wave, lane, s2, p2, SECOND_CHANCE_K_ALL);
    if (wave == 0 && ok) {
        if ((uint32_t)lane < found) {
            out_labels[(size_t)b * k + lane] = ids[p2];
            out_dist[(size_t)b * k + lane] = -s2;
        }
        if (lane == 0) out_flags[b] = rerun ? FLAG_RERUN : FLAG_SECOND;
    }
}

// ------------------------------------------------------------------------------------------------
// host side: pass planning + launch sequence
// ------------------------------------------------------------------------------------------------
BatchPlan plan_batched(uint32_t n_rows, int target, uint32_t k) { return plan_batched_tiles(n_rows, TILE_ROWS, target, k); }

// tile_rows: 64 (16-bit tiles) or 128 (int8 tiles, scan_i8.hip): the samples cover the same numbers of ROWS
BatchPlan plan_batched_tiles(uint32_t n_rows, uint32_t tile_rows, int target_per_query, uint32_t k) {
    BatchPlan pl{};
    const uint32_t TILE_ROWS = tile_rows;  // (shadows the constant)
    const uint32_t n_tiles = (n_rows + TILE_ROWS - 1) / TILE_ROWS;
    pl.n_tiles_total = n_tiles;
    if (n_rows <= (uint32_t)BATCH_CAP) {
        pl.dense_only = true;
        return pl;
    }
    // sample 1: up to 8192 strided rows, dense; tau = the m1-th largest sample score, m1 <= 64 (tau_select reads it off a
    // top-64 list).  A small index (m_full > 64) samples fewer rows instead, so that the 64th largest of the sample still
    // sits at the target depth: the candidates' depth is what the certificates rest on (rows that never became candidates
    // are only known to score <= tau in the pass's own bound — int8: E + K2 ~ 0.017 of slack on unit vectors — so tau has
    // to lie c_min ~ k exp(slack z / sigma) ranks deep: 75 .. 120 candidates for k = 20 on 1 M .. 100 M rows, 250 .. 390
    // for k = 64; the sampled estimate of that depth scatters like target x Gamma(m) / m, m = the rank tau is read from).
    // expected candidates per query in the full pass (option "mfma_target"; twice that for k > 32)
    const double target = (double)target_per_query * (k > 32 ? 2.0 : 1.0);
    const double m_full = target * BATCH_CAP / (double)n_rows;
    uint32_t s1_rows = BATCH_CAP;
    if (m_full > (double)LIST) {
        s1_rows = (uint32_t)((double)BATCH_CAP * (double)LIST / m_full);
        if (s1_rows < 8u * TILE_ROWS) s1_rows = 8u * TILE_ROWS;
    }
    pl.s1_tiles = (s1_rows + TILE_ROWS - 1) / TILE_ROWS;
    if (pl.s1_tiles > n_tiles) pl.s1_tiles = n_tiles;
    pl.s1_stride = n_tiles / pl.s1_tiles;  // >= 1
    // one sample is enough while the threshold can be read from at least the 8th largest sample score (n <= 1 M rows at
    // target 1024): the candidate count of the full pass then follows target/8 x Gamma(8) — below c_min once in > 10^5
    if (m_full >= 8.0) {
        const double m_s = target * (double)(pl.s1_tiles * TILE_ROWS) / (double)n_rows;
        pl.m1 = (uint32_t)(m_s + 0.999);
        if (pl.m1 > (uint32_t)LIST) pl.m1 = LIST;
        if (pl.m1 < 1) pl.m1 = 1;
        pl.s2_tiles = 0;
        return pl;
    }
    // sample 2: n/64 rows (so that the threshold is the ~24th largest of it: the count of candidates of the full pass then
    // scatters by ~20 %), at least 16 Ki and at most 1.5M rows, appended above tau1
    uint32_t t2 = n_tiles / 64;
    if (t2 < 16384u / TILE_ROWS) t2 = 16384u / TILE_ROWS;
    if (t2 > 1500000u / TILE_ROWS) t2 = 1500000u / TILE_ROWS;
    pl.s2_tiles = t2;
    pl.s2_stride = n_tiles / t2;
    const double n2 = (double)t2 * TILE_ROWS;
    double m1 = 2048.0 * BATCH_CAP / n2;
    pl.m1 = (uint32_t)(m1 + 0.999);
    if (pl.m1 < 8) pl.m1 = 8;
    // <= 32: tau_select takes its insertion path (the m-th largest of 8192 scores in 15 instead of 33 us); sample 2 then still
    // appends >= 4x the m2 entries its threshold is read from
    if (pl.m1 > 32u) pl.m1 = 32u;
    double m2 = target * n2 / (double)n_rows;
    pl.m2 = (uint32_t)(m2 + 0.999);
    if (pl.m2 < 8) pl.m2 = 8;
    if (pl.m2 > (uint32_t)LIST) pl.m2 = LIST;  // (target 1024: m2 = 16 while sample 2 is n / 64 rows)
    return pl;
}

static OncePerDevice g_lds_attr_once;


// Kernel choice and threshold target travel in the index's BatchWorkspace (ws.sched / ws.target / ws.diag: per index, not
// per process).  The timing experiments (parts of the pipelined kernels switched off: wrong results by design) and the
// stamped diagnostic kernel only exist in builds with -DDAWN_EXPERIMENTS (make EXPERIMENTS=1); the release library does
// not contain them.

template <bool DENSE, int NW, int SCHED>
static void launch_pass_nw(const void* d_x, uint32_t n_rows, uint32_t first, uint32_t stride, uint32_t n_tiles,
                           const BatchWorkspace& ws, int n_q, int grid, hipStream_t stream) {
    const uint32_t blocks = n_tiles < (uint32_t)grid ? n_tiles : (uint32_t)grid;
    hipLaunchKernelGGL((scan_f16_kernel<DENSE, NW, SCHED>), dim3(blocks), dim3(NW * 64), LDS_BYTES, stream, d_x, n_rows,
                       first, stride, n_tiles, reinterpret_cast<const half8*>(ws.qh), n_q, ws.tau, ws.cnt,
                       reinterpret_cast<uint2*>(ws.cand), reinterpret_cast<float*>(ws.cand), ws.diag);
}

template <bool DENSE>
static void launch_pass_rt(const void* d_rows, uint32_t n_rows, uint32_t first, uint32_t stride, uint32_t n_tiles,
                           const BatchWorkspace& ws, int n_q, int grid, hipStream_t stream) {
#ifdef DAWN_EXPERIMENTS
    if (ws.sched == 2 && !DENSE) {  // lockstep kernel with diagnostic stamps (full append pass only)
        launch_pass_nw<false, 8, 2>(d_rows, n_rows, first, stride, n_tiles, ws, n_q, grid, stream);
        return;
    }
#endif
    launch_pass_nw<DENSE, 8, 0>(d_rows, n_rows, first, stride, n_tiles, ws, n_q, grid, stream);
}

// Fragment-ordered 16-bit tiles (f16 shadow of an f32 index, or a bf16 index): the LDS-DMA kernels.
template <bool DENSE, bool BF16>
static void launch_pass_dma(const void* d_rows, uint32_t n_rows, uint32_t first, uint32_t stride, uint32_t n_tiles,
                            const BatchWorkspace& ws, int n_q, int grid, hipStream_t stream) {
    const uint32_t blocks = n_tiles < (uint32_t)grid ? n_tiles : (uint32_t)grid;
#define DAWN_PIPE_LAUNCH(DBG_)                                                                                          \
    hipLaunchKernelGGL((scan_f16_pipe_kernel<DENSE, DBG_, BF16>), dim3(blocks), dim3(256), 0, stream,                   \
                       reinterpret_cast<const unsigned char*>(d_rows), n_rows, first, stride, n_tiles,                  \
                       reinterpret_cast<const half8*>(ws.qh), n_q, ws.tau, ws.cnt, reinterpret_cast<uint2*>(ws.cand),   \
                       reinterpret_cast<float*>(ws.cand))
    // mfma_sched 4 (default; also 0 / 2 on a bf16 index): the pipelined kernel for long append passes; short ones (below
    // ~500 tiles per CU = 8M rows) are dominated by candidate handling, where two waves per SIMD hide each other's slow
    // paths, and go to the 8-wave kernel like the sample passes (measured full pass, 256 queries: 40M rows 7.43 vs
    // 7.82 ms, 12.5M 2.47 vs 2.53, 4M 0.91 vs 0.87, 1M 0.34 vs 0.28 ms).  5: pipelined kernel for every pass (tests).
    // 1: 8-wave kernel only.
    const int v = ws.sched;
    const bool pipe = v >= 5 || (v != 1 && !DENSE && n_tiles >= (1u << 17));
    if (!pipe)
        hipLaunchKernelGGL((scan_f16_dma_kernel<DENSE, 8, BF16>), dim3(blocks), dim3(512), 0, stream,
                           reinterpret_cast<const unsigned char*>(d_rows), n_rows, first, stride, n_tiles,
                           reinterpret_cast<const half8*>(ws.qh), n_q, ws.tau, ws.cnt, reinterpret_cast<uint2*>(ws.cand),
                           reinterpret_cast<float*>(ws.cand));
#ifdef DAWN_EXPERIMENTS
    else if (!BF16 && !DENSE && v == 41) DAWN_PIPE_LAUNCH(1);   // 41..55: timing experiments, parts switched off
    else if (!BF16 && !DENSE && v == 42) DAWN_PIPE_LAUNCH(2);
    else if (!BF16 && !DENSE && v == 43) DAWN_PIPE_LAUNCH(3);
    else if (!BF16 && !DENSE && v == 44) DAWN_PIPE_LAUNCH(4);
    else if (!BF16 && !DENSE && v == 47) DAWN_PIPE_LAUNCH(7);
    else if (!BF16 && !DENSE && v == 48) DAWN_PIPE_LAUNCH(8);
    else if (!BF16 && !DENSE && v == 55) DAWN_PIPE_LAUNCH(15);
    else if (!BF16 && !DENSE && v == 54) DAWN_PIPE_LAUNCH(16);
#endif
    else DAWN_PIPE_LAUNCH(0);
#undef DAWN_PIPE_LAUNCH
}

// rows: the filter's row source; rt: its row type (ROW_F32: converted on the fly by the lockstep kernel; ROW_F16S /
// ROW_BF16: fragment-ordered tiles)
template <bool DENSE>
static void launch_pass(const void* d_rows, int rt, uint32_t n_rows, uint32_t first, uint32_t stride, uint32_t n_tiles,
                        const BatchWorkspace& ws, int n_q, int grid, hipStream_t stream) {
    if (n_tiles == 0) return;
    if (rt == ROW_F16S) launch_pass_dma<DENSE, false>(d_rows, n_rows, first, stride, n_tiles, ws, n_q, grid, stream);
    else if (rt == ROW_BF16) launch_pass_dma<DENSE, true>(d_rows, n_rows, first, stride, n_tiles, ws, n_q, grid, stream);
    else launch_pass_rt<DENSE>(d_rows, n_rows, first, stride, n_tiles, ws, n_q, grid, stream);
}

static hipError_t set_lds_attr_rt() {
    const void* fns[] = {reinterpret_cast<const void*>(scan_f16_kernel<true, 8, 0>),
#ifdef DAWN_EXPERIMENTS
                         reinterpret_cast<const void*>(scan_f16_kernel<false, 8, 2>),
#endif
                         reinterpret_cast<const void*>(scan_f16_kernel<false, 8, 0>)};
    for (const void* f : fns) {
        hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

int batched_init() {  // (on the current device: every shard of a multi-device handle calls it with its own device set)
    hipError_t e = hipSuccess;
    once_per_device(g_lds_attr_once, [&e] {
        e = set_lds_attr_rt();
        const void* tails[] = {reinterpret_cast<const void*>(select_rescore_kernel<true, 0>),
                               reinterpret_cast<const void*>(select_rescore_kernel<false, 0>),
                               reinterpret_cast<const void*>(select_rescore_kernel<true, 1>),
                               reinterpret_cast<const void*>(select_rescore_kernel<false, 1>)};
        for (int i = 0; i < 4 && e == hipSuccess; ++i)
            e = hipFuncSetAttribute(tails[i], hipFuncAttributeMaxDynamicSharedMemorySize,
                                    i < 2 ? RescoreStage<0>::BYTES : RescoreStage<1>::BYTES);
    });
    return (int)e;
}

// f32 rows -> the filter's shadow copy (ROW_F16S, kernels.hpp): value * 2^8 rounded to nearest even, stored tile by
// tile (64 rows, 48 KiB) in MFMA-fragment order.  One workgroup per tile: coalesced f32 reads, fragment-order
// image assembled in LDS, linear coalesced write.  Rows >= n_valid of the last tile are written as zeros.
__global__ __launch_bounds__(256) void rows_f32_to_f16s_kernel(const f32x4* __restrict__ x, half8* __restrict__ shadow,
                                                              uint32_t first_tile, uint32_t n_valid) {
    __shared__ __attribute__((aligned(16))) half4 img[TILE_ROWS * ROW_F4];  // 48 KiB, half4 units
    const uint32_t tile = first_tile + blockIdx.x;
    const f32x4* src = x + (size_t)tile * (TILE_ROWS * ROW_F4);
    for (uint32_t c = threadIdx.x; c < TILE_ROWS * ROW_F4; c += 256) {
        const uint32_t row = c / ROW_F4, c4 = c % ROW_F4;  // f32x4 chunk c4 of tile row `row`: k = 4*c4..4*c4+3
        half4 v = {(_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f};
        if (tile * TILE_ROWS + row < n_valid) v = to_half4_scaled(src[c]);
        const uint32_t g = c4 >> 1;  // k-group (8 values) -> fragment (sub, s = g/2), lane (h = g&1, r)
        const uint32_t slot = ((row >> 5) * 24u + (g >> 1)) * 64u + (g & 1u) * 32u + (row & 31u);
        img[slot * 2u + (c4 & 1u)] = v;
    }
    __syncthreads();
    const half8* im8 = reinterpret_cast<const half8*>(img);
    half8* dst = shadow + (size_t)tile * (TILE_ROWS * ROW_C8);
    for (uint32_t c = threadIdx.x; c < TILE_ROWS * ROW_C8; c += 256) dst[c] = im8[c];
}

// Converts the tiles that hold rows [first_row, n_valid) of the f32 rows d_rows (row 0 = index row 0; the capacity
// is padded to whole tiles) into d_shadow; rows below first_row in the first tile are converted again (same values).
void launch_rows_f32_to_f16s(const float* d_rows, void* d_shadow, size_t first_row, size_t n_valid, hipStream_t stream) {
    if (n_valid <= first_row) return;
    const uint32_t t0 = (uint32_t)(first_row / TILE_ROWS), t1 = (uint32_t)((n_valid + TILE_ROWS - 1) / TILE_ROWS);
    hipLaunchKernelGGL(rows_f32_to_f16s_kernel, dim3(t1 - t0), dim3(256), 0, stream, reinterpret_cast<const f32x4*>(d_rows),
                       reinterpret_cast<half8*>(d_shadow), t0, (uint32_t)n_valid);
}

// queries -> the filter's 16-bit operand images: bf16 for a bf16 index (frt = ROW_BF16), scaled f16 otherwise
static void prep_queries(const float* d_q, int B, const BatchWorkspace& ws, int frt, hipStream_t stream) {
    unsigned short* qh = reinterpret_cast<unsigned short*>(ws.qh);
    if (frt == ROW_BF16)
        hipLaunchKernelGGL(prep_queries_kernel<true>, dim3(BATCH_QT * EM / 256), dim3(256), 0, stream, d_q, B, qh);
    else
        hipLaunchKernelGGL(prep_queries_kernel<false>, dim3(BATCH_QT * EM / 256), dim3(256), 0, stream, d_q, B, qh);
}

void launch_prep_queries(const float* d_q, int B, const BatchWorkspace& ws, hipStream_t stream) {
    prep_queries(d_q, B, ws, ROW_F16S, stream);
}

// Timing hook: the full append pass alone (thresholds ws.tau as left by the last search), `iters` times.
void launch_batched_full_pass(const void* d_frows, int frt, uint32_t n_rows, int B, const BatchWorkspace& ws, int grid,
                              int iters, hipStream_t stream, hipEvent_t ev0, hipEvent_t ev1) {
    const BatchPlan pl = plan_batched(n_rows, ws.target, 10);
    (void)hipEventRecord(ev0, stream);
    for (int i = 0; i < iters; ++i) {
        (void)hipMemsetAsync(ws.cnt, 0, BATCH_QT * BATCH_CAND_SEGS * sizeof(uint32_t), stream);
        launch_pass<false>(d_frows, frt, n_rows, 0, 1, pl.n_tiles_total, ws, B, grid, stream);
    }
    (void)hipEventRecord(ev1, stream);
}

void launch_batched_dense_scores(const void* d_frows, int frt, uint32_t n_rows, const float* d_q, int B,
                                 const BatchWorkspace& ws, int grid, hipStream_t stream) {
    const void* d_x = d_frows;
    const int dtype = frt;
    prep_queries(d_q, B, ws, frt, stream);
    const uint32_t n = n_rows < (uint32_t)BATCH_CAP ? n_rows : (uint32_t)BATCH_CAP;
    launch_pass<true>(d_x, dtype, n_rows, 0, 1, (n + TILE_ROWS - 1) / TILE_ROWS, ws, B, grid, stream);
}

// eps_override > 0: the filter's bound when it is not the 16-bit matrix-core one (int8 upper-bound scores)
template <bool DENSE>
static void launch_select_rescore(const void* d_x, int dtype, const uint64_t* d_ids, uint32_t n_rows, const float* d_q,
                                  int B, uint32_t k, const BatchWorkspace& ws, uint64_t* d_labels, float* d_dist,
                                  uint32_t* d_found, uint32_t* d_flags, int force_fallback, hipStream_t stream,
                                  float eps_override = 0.f, int rerun = 0) {
    const float* dense = reinterpret_cast<const float*>(ws.cand);
    const uint2* cand = reinterpret_cast<const uint2*>(ws.cand);
    // the bf16-rounded rows may exceed the is_normalized band by 2^-8: scale the bound on sum|q_i x_i| accordingly
    const float eps = eps_override > 0.f ? eps_override : dtype == ROW_BF16 ? FILTER_EPS_BF16_MFMA : FILTER_EPS_F16;
    if (dtype == ROW_BF16)
        hipLaunchKernelGGL((select_rescore_kernel<DENSE, 1>), dim3(B), dim3(1024), RescoreStage<1>::BYTES, stream, d_x,
                           d_ids, n_rows, d_q, dense, cand, ws.cnt, ws.tau, k, d_labels, d_dist, d_found, d_flags,
                           force_fallback, eps, rerun);
    else
        hipLaunchKernelGGL((select_rescore_kernel<DENSE, 0>), dim3(B), dim3(1024), RescoreStage<0>::BYTES, stream, d_x,
                           d_ids, n_rows, d_q, dense, cand, ws.cnt, ws.tau, k, d_labels, d_dist, d_found, d_flags,
                           force_fallback, eps, rerun);
}

void launch_select_rescore_eps(bool dense_pass, const void* d_x, int dtype, const uint64_t* d_ids, uint32_t n_rows,
                               const float* d_q, int B, uint32_t k, const BatchWorkspace& ws, uint64_t* d_labels, float* d_dist,
                               uint32_t* d_found, uint32_t* d_flags, int force_fallback, float eps, hipStream_t stream, int rerun) {
    if (dense_pass)
        launch_select_rescore<true>(d_x, dtype, d_ids, n_rows, d_q, B, k, ws, d_labels, d_dist, d_found, d_flags, force_fallback,
                                    stream, eps, rerun);
    else
        launch_select_rescore<false>(d_x, dtype, d_ids, n_rows, d_q, B, k, ws, d_labels, d_dist, d_found, d_flags, force_fallback,
                                     stream, eps, rerun);
}

void launch_tau_select(bool dense_pass, int B, const BatchWorkspace& ws, uint32_t dense_count, uint32_t m, hipStream_t stream) {
    const float* dense = reinterpret_cast<const float*>(ws.cand);
    const uint2* cand = reinterpret_cast<const uint2*>(ws.cand);
    if (dense_pass)
        hipLaunchKernelGGL((tau_select_kernel<true>), dim3(B), dim3(1024), 0, stream, dense, cand, ws.cnt, dense_count, m, ws.tau, ws.pool);
    else
        hipLaunchKernelGGL((tau_select_kernel<false>), dim3(B), dim3(1024), 0, stream, dense, cand, ws.cnt, 0u, m, ws.tau, ws.pool);
}

// d_x/dtype: the index rows (exact rescore); d_frows/frt: the filter's row source (the same rows, or the scaled f16
// shadow copy ROW_F16S of an f32 index)
void launch_scan_batched(const void* d_x, int dtype, const void* d_frows, int frt, const uint64_t* d_ids, uint32_t n_rows,
                         const float* d_q, int B, uint32_t k, const BatchWorkspace& ws, int grid, uint64_t* d_labels,
                         float* d_dist, uint32_t* d_found, uint32_t* d_flags, int force_fallback, hipStream_t stream,
                         hipEvent_t ev0, hipEvent_t ev1) {
    const BatchPlan pl = plan_batched(n_rows, ws.target, k);
    const float* dense = reinterpret_cast<const float*>(ws.cand);
    const uint2* cand = reinterpret_cast<const uint2*>(ws.cand);
    prep_queries(d_q, B, ws, frt, stream);
    if (pl.dense_only) {
        if (ev0) (void)hipEventRecord(ev0, stream);
        launch_pass<true>(d_frows, frt, n_rows, 0, 1, pl.n_tiles_total, ws, B, grid, stream);
        if (ev1) (void)hipEventRecord(ev1, stream);
        launch_select_rescore<true>(d_x, dtype, d_ids, n_rows, d_q, B, k, ws, d_labels, d_dist, d_found, d_flags,
                                    force_fallback, stream);
        return;
    }
    launch_pass<true>(d_frows, frt, n_rows, 0, pl.s1_stride, pl.s1_tiles, ws, B, grid, stream);
    hipLaunchKernelGGL((tau_select_kernel<true>), dim3(B), dim3(1024), 0, stream, dense, cand, ws.cnt,
                       pl.s1_tiles * TILE_ROWS, pl.m1, ws.tau, ws.pool);
    if (pl.s2_tiles) {  // (the counters were left at zero by tau_select)
        launch_pass<false>(d_frows, frt, n_rows, 0, pl.s2_stride, pl.s2_tiles, ws, B, grid, stream);
        hipLaunchKernelGGL((tau_select_kernel<false>), dim3(B), dim3(1024), 0, stream, dense, cand, ws.cnt, 0u, pl.m2,
                           ws.tau, ws.pool);
    }
    if (ev0) (void)hipEventRecord(ev0, stream);
    launch_pass<false>(d_frows, frt, n_rows, 0, 1, pl.n_tiles_total, ws, B, grid, stream);
    if (ev1) (void)hipEventRecord(ev1, stream);
    launch_select_rescore<false>(d_x, dtype, d_ids, n_rows, d_q, B, k, ws, d_labels, d_dist, d_found, d_flags,
                                 force_fallback, stream);
}

}  // namespace dawn
