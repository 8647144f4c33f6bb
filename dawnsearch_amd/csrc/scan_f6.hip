// scan_f6.hip — 6-bit FLOATING-POINT filter shadow of the index rows (ROW_F6S) for the matrix-core pass of a batch.
//
// The batched pass (scan_i8.hip: 2..256 queries against every row on the integer matrix cores) is bound by the chip's power
// envelope, not by its structure: 2.2 Pop/s of int8 matrix work next to a 4.3-TB/s stream, 8.9 ms per 100 M x 256, whatever the
// kernel looks like.  The block-scaled v_mfma_scale_f32_16x16x128_f8f6f4 sustains 1.5 x the int8 rate on FP6
// (e2m3) operands under the same envelope — measured WITH the row stream beside it (tools/probes/mfma_hbm_mix.hip,
// profiles/r04/mfma_hbm_mix_probe.log: 5.14 against 7.75 ms per 100 M x 256 of raw mix) — and its rows are 288 B instead of 384.
//
// Construction (the int8 / packed shadows' with another grid):
//   * rows are rotated by R (rotate384.hpp) and quantised per 16-row TILE: s = c max|x'| / 7.5 (c: the candidate of
//     {1, .92, .84, .76, .68, .60} with the smallest resulting error), X = e2m3(x' / s) in +-{0, 1/8 .. 7/8, 1 .. 15/8, 2 .. 15/4,
//     4 .. 15/2}, round to nearest, saturating; the tile stores {1 / s, E}, E >= 1.0101 max_r ||x'_r - s X_r||_2 MEASURED;
//   * a tile is 3 k-steps of 1536 B in the A-operand order of v_mfma_scale_f32_16x16x128_f8f6f4: lane l = (kb = l >> 4, r = l & 15)
//     holds the 32 values k = 128 ks + 32 kb + i of row r as a little-endian string of 32 x 6 bits = 6 dwords
//     (tools/probes/fp6_layout_check.hip verifies the layout on the device), stored [64 lanes x 16 B | 64 lanes x 8 B] so that a
//     wave loads a fragment with one dwordx4 and one dwordx2 per lane; block scales are all 1 (E8M0 0x7F);
//   * a query enters as ONE e2m3 image V = e2m3(q' / s_q), s_q = max|q'| / 7.5, with ||dq||_2 = ||q' - s_q V||_2 measured;
//   * products of two e2m3 values are multiples of 1/64 below 57, a sum of 384 of them stays below 2^24 / 64: the f32
//     accumulation of the MFMA is EXACT, so with x' = s X + dx, q' = s_q V + dq:
//         x.q = s s_q acc + dx.q' + (s X).dq,   |dx.q'| <= E,   |(s X).dq| <= (1.015 + E) ||dq||_2 =: K2
//     and ub = fma(acc, s s_q, E + K2) >= x.q up to FILTER_EPS_I8 (the same rounding budget as the integer shadows').
// E + K2 ~ 0.07 on unit vectors — six times the int8 pass's slack: this shadow is a FIRST filter.  Its pass keeps every row
// whose bound exceeds a threshold tau6 that sits ~12 k ranks deep (so that tau6 stays below the k-th best score), the survivors
// are re-scored on the int8 shadow (a lane per candidate, v_dot4_i32_i8: the packed stream's refinement) and the usual tail takes
// over with base = tau6.
#include <type_traits>

#include "kernels.hpp"
#include "rotate384.hpp"
#include "wave_topk.hpp"

namespace dawn {

typedef int i32x8_t __attribute__((ext_vector_type(8)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

// (16 rows per tile)
constexpr uint32_t F6_KS_DW = 384;            // dwords per k-step fragment (1536 B)
constexpr uint32_t F6_TILE_DW = 3 * F6_KS_DW;  // 4608 B per 16 rows = 288 B per row
constexpr float F6_MAX = 7.5f;

// e2m3 code (5 bits of magnitude: e = c >> 3, m = c & 7) of t >= 0, round to nearest, saturating at 7.5; *deq = its value
__device__ __forceinline__ uint32_t f6_encode_mag(float t, float* deq) {
    t = fminf(t, F6_MAX);
    const int e = t < 1.0f ? 0 : t < 2.0f ? 1 : t < 4.0f ? 2 : 3;
    const float step = e <= 1 ? 0.125f : e == 2 ? 0.25f : 0.5f;
    float qn = rintf(t / step);  // e = 0: 0..8; e >= 1: 8..16
    if (e == 3) qn = fminf(qn, 15.0f);
    *deq = qn * step;
    const int q = (int)qn;
    return e == 0 ? (uint32_t)q : (uint32_t)((e << 3) + (q - 8));  // (q = 16 carries into the next exponent: the same value)
}
__device__ __forceinline__ uint32_t f6_encode(float v, float inv_s, float* deq) {
    float d;
    const uint32_t c = f6_encode_mag(fabsf(v) * inv_s, &d);
    *deq = v < 0.f ? -d : d;
    return c | (v < 0.f ? 32u : 0u);
}

// ------------------------------------------------------------------------------------------------
// conversion: rows -> FP6 tiles + {1 / s, E} per 16-row tile.  One 256-thread block per 32 rows (two tiles); thread = (row r =
// tid / 8, part = tid % 8) holds v[j] = elements 32 j + 4 part + {0..3}: the 8 threads of a row hold the 32 values of k-block j.
// ------------------------------------------------------------------------------------------------
template <int RT>
__global__ __launch_bounds__(256) void rows_to_f6s_kernel(const void* __restrict__ xv, uint32_t* __restrict__ out,
                                                           float2* __restrict__ meta, uint32_t first_pair, uint32_t n_valid) {
    constexpr int NC = 6;
    __shared__ float sh[4];
    __shared__ float sh_e[4][NC];
    const uint32_t pair = first_pair + blockIdx.x;  // rows 32 pair .. +31 = tiles 2 pair, 2 pair + 1
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t r = tid >> 3, part = tid & 7;
    const uint32_t row = pair * 32u + r;
    f32x4 v[12];
#pragma unroll
    for (int j = 0; j < 12; ++j) {
        v[j] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (row < n_valid) {
            const uint32_t c4 = part + 8 * j;
            if (RT == 0) {
                v[j] = reinterpret_cast<const f32x4*>(xv)[(size_t)row * ROW_F4 + c4];
            } else {
                const u32x4 w = reinterpret_cast<const u32x4*>(xv)[frag_chunk(row, (int)(c4 >> 1))];
                const uint32_t w0 = (c4 & 1u) ? w.z : w.x, w1 = (c4 & 1u) ? w.w : w.y;
                v[j] = f32x4{bf16_lo(w0), bf16_hi(w0), bf16_lo(w1), bf16_hi(w1)};
            }
        }
    }
    rotate384_rowpart(v, part, lane);
    // tile = waves {0, 1} (rows 0-15) / {2, 3} (rows 16-31)
    float amax = 0.f;
#pragma unroll
    for (int j = 0; j < 12; ++j)
        amax = fmaxf(amax, fmaxf(fmaxf(fabsf(v[j].x), fabsf(v[j].y)), fmaxf(fabsf(v[j].z), fabsf(v[j].w))));
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) amax = fmaxf(amax, __shfl_xor(amax, o));
    if (lane == 0) sh[wave] = amax;
    __syncthreads();
    const int w0 = wave & ~1;
    amax = fmaxf(sh[w0], sh[w0 + 1]);
    const float s0 = fmaxf(amax, 1e-20f) / F6_MAX;
    float e2c[NC];
#pragma unroll
    for (int ci = 0; ci < NC; ++ci) {
        const float inv_s = 1.0f / (s0 * (1.0f - 0.08f * ci)), s = s0 * (1.0f - 0.08f * ci);
        float e2 = 0.f;
#pragma unroll
        for (int j = 0; j < 12; ++j) {
            const float vv[4] = {v[j].x, v[j].y, v[j].z, v[j].w};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float dq;
                (void)f6_encode(vv[i], inv_s, &dq);
                const float dx = vv[i] - s * dq;
                e2 = __builtin_fmaf(dx, dx, e2);
            }
        }
        e2 += __shfl_xor(e2, 1);  // row sum over its 8 threads, then the maximum over the wave's 8 rows
        e2 += __shfl_xor(e2, 2);
        e2 += __shfl_xor(e2, 4);
#pragma unroll
        for (int o2 = 32; o2 >= 8; o2 >>= 1) e2 = fmaxf(e2, __shfl_xor(e2, o2));
        e2c[ci] = e2;
    }
    if (lane == 0) {
#pragma unroll
        for (int ci = 0; ci < NC; ++ci) sh_e[wave][ci] = e2c[ci];
    }
    __syncthreads();
    float best_e2 = 0.f, s = s0;
#pragma unroll
    for (int ci = 0; ci < NC; ++ci) {  // (both waves of a tile take the same decision)
        const float m = fmaxf(sh_e[w0][ci], sh_e[w0 + 1][ci]);
        if (ci == 0 || m < best_e2) {
            best_e2 = m;
            s = s0 * (1.0f - 0.08f * ci);
        }
    }
    const float inv_s = 1.0f / s;
    const uint32_t tile = pair * 2u + (r >> 4), r16 = r & 15u;
    uint32_t* o = out + (size_t)tile * F6_TILE_DW;
    const int base = lane & ~7;  // first lane of this row's 8 threads
#pragma unroll
    for (int j = 0; j < 12; ++j) {  // k-block j = 4 ks + kb: this thread's 4 values are i = 4 part .. +3 of the block's 32
        const float vv[4] = {v[j].x, v[j].y, v[j].z, v[j].w};
        uint32_t c24 = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float dq;
            c24 |= f6_encode(vv[i], inv_s, &dq) << (6 * i);
        }
        // chunk p of the row's 8 threads occupies bits [24 p, 24 p + 24) of the block's 192-bit string
        const uint32_t c0 = __shfl(c24, base + 0), c1 = __shfl(c24, base + 1), c2 = __shfl(c24, base + 2), c3 = __shfl(c24, base + 3);
        const uint32_t c4 = __shfl(c24, base + 4), c5 = __shfl(c24, base + 5), c6 = __shfl(c24, base + 6), c7 = __shfl(c24, base + 7);
        uint32_t d = 0;
        switch (part) {
            case 0: d = c0 | (c1 << 24); break;
            case 1: d = (c1 >> 8) | (c2 << 16); break;
            case 2: d = (c2 >> 16) | (c3 << 8); break;
            case 3: d = c4 | (c5 << 24); break;
            case 4: d = (c5 >> 8) | (c6 << 16); break;
            case 5: d = (c6 >> 16) | (c7 << 8); break;
            default: break;
        }
        const uint32_t ks = (uint32_t)j >> 2, kb = (uint32_t)j & 3u, l = kb * 16u + r16;
        if (part < 4u) o[ks * F6_KS_DW + l * 4u + part] = d;
        else if (part < 6u) o[ks * F6_KS_DW + 256u + l * 2u + (part - 4u)] = d;
    }
    if ((tid & 127) == 0) {
        // 1.0101: ||q||_2 < 1.01 (gate); 1.001 + 1e-9: the f32 evaluation of dx, the sum and the square root
        meta[tile] = float2{inv_s, sqrtf(best_e2) * 1.0101f * 1.001f + 1e-9f};
    }
}

void launch_rows_to_f6s(const void* d_rows, int rt, void* d_shadow, void* d_meta, size_t first_row, size_t n_valid, hipStream_t stream) {
    const uint32_t first_pair = (uint32_t)(first_row / 32);  // the 32-row pair holding first_row is re-quantised whole
    const uint32_t end_pair = (uint32_t)((n_valid + 31) / 32);
    if (end_pair <= first_pair) return;
    if (rt == ROW_BF16)
        hipLaunchKernelGGL(rows_to_f6s_kernel<1>, dim3(end_pair - first_pair), dim3(256), 0, stream, d_rows,
                           reinterpret_cast<uint32_t*>(d_shadow), reinterpret_cast<float2*>(d_meta), first_pair, (uint32_t)n_valid);
    else
        hipLaunchKernelGGL(rows_to_f6s_kernel<0>, dim3(end_pair - first_pair), dim3(256), 0, stream, d_rows,
                           reinterpret_cast<uint32_t*>(d_shadow), reinterpret_cast<float2*>(d_meta), first_pair, (uint32_t)n_valid);
}

// ------------------------------------------------------------------------------------------------
// queries -> e2m3 images in B-operand order + {s_q, ||dq||_2} per query.  One wave per query (grid = BATCH_QT: the images of
// the slots beyond n_q are zero).  qf6: [16 groups][3 k-steps][64 lanes][6 dwords]; lane l = (kb, c16): query 16 g + c16.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void prep_queries_f6_kernel(const float* __restrict__ q, int n_q, uint32_t* __restrict__ qf6,
                                                              float2* __restrict__ qmeta) {
    __shared__ unsigned char codes[EM];
    const int b = blockIdx.x, lane = threadIdx.x;
    float v[6];
#pragma unroll
    for (int j = 0; j < 6; ++j) v[j] = b < n_q ? q[(size_t)b * EM + lane + 64 * j] : 0.f;
    rotate384_wave(v, lane);
    float amax = 0.f;
#pragma unroll
    for (int j = 0; j < 6; ++j) amax = fmaxf(amax, fabsf(v[j]));
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) amax = fmaxf(amax, __shfl_xor(amax, o));
    const float sq = fmaxf(amax, 1e-20f) / F6_MAX, inv = 1.0f / sq;
    float e2 = 0.f;
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        float dq;
        const uint32_t c = f6_encode(v[j], inv, &dq);
        codes[lane + 64 * j] = b < n_q ? (unsigned char)c : (unsigned char)0;
        const float dx = v[j] - sq * dq;
        e2 = __builtin_fmaf(dx, dx, e2);
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) e2 += __shfl_xor(e2, o);
    __syncthreads();
    const int g = b >> 4, c16 = b & 15;
    for (int w = lane; w < 72; w += 64) {  // dword d of k-block kbk = 4 ks + kb
        const int kbk = w / 6, d = w % 6;
        uint32_t word = 0;
        const int first = (32 * d) / 6, last = (32 * d + 31) / 6;  // codes overlapping bits [32 d, 32 d + 32)
        for (int i = first; i <= last && i < 32; ++i) {
            const int sh = 6 * i - 32 * d;
            const uint32_t c = codes[32 * kbk + i];
            word |= sh >= 0 ? (c << sh) : (c >> (-sh));
        }
        const int ks = kbk >> 2, kb = kbk & 3;
        qf6[(((size_t)g * 3 + ks) * 64 + kb * 16 + c16) * 6 + d] = word;
    }
    if (lane == 0) qmeta[b] = float2{sq, b < n_q ? sqrtf(e2) * 1.001f + 1e-9f : 0.f};
}

void launch_prep_queries_f6(const float* d_q, int n_q, void* d_qf6, void* d_qmeta, hipStream_t stream) {
    hipLaunchKernelGGL(prep_queries_f6_kernel, dim3(BATCH_QT), dim3(64), 0, stream, d_q, n_q, reinterpret_cast<uint32_t*>(d_qf6),
                       reinterpret_cast<float2*>(d_qmeta));
}

// ------------------------------------------------------------------------------------------------
// Test hook: dense upper-bound scores of rows [0, n) (n <= BATCH_CAP) for B queries -> out [BATCH_QT][BATCH_CAP] f32.  One wave
// per 16-row tile; nothing here is tuned.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ i32x8_t f6_load_a(const uint32_t* tile, int ks, int lane) {
    const u32x4 lo = *reinterpret_cast<const u32x4*>(tile + ks * F6_KS_DW + lane * 4);
    const u32x2 hi = *reinterpret_cast<const u32x2*>(tile + ks * F6_KS_DW + 256 + lane * 2);
    return i32x8_t{(int)lo.x, (int)lo.y, (int)lo.z, (int)lo.w, (int)hi.x, (int)hi.y, 0, 0};
}

__global__ __launch_bounds__(64) void f6_dense_scores_kernel(const uint32_t* __restrict__ x, const float2* __restrict__ meta,
                                                              uint32_t n_rows, const uint32_t* __restrict__ qf6,
                                                              const float2* __restrict__ qmeta, int n_q, float* __restrict__ out) {
    const int lane = threadIdx.x;
    const uint32_t tile = blockIdx.x;
    const uint32_t* t = x + (size_t)tile * F6_TILE_DW;
    const float2 mt = meta[tile];
    const float s = __builtin_amdgcn_rcpf(mt.x);
    i32x8_t a[3];
#pragma unroll
    for (int ks = 0; ks < 3; ++ks) a[ks] = f6_load_a(t, ks, lane);
    for (int g = 0; g * 16 < n_q; ++g) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 3; ++ks) {
            const uint32_t* bp = qf6 + (((size_t)g * 3 + ks) * 64 + lane) * 6;
            const i32x8_t bv = {(int)bp[0], (int)bp[1], (int)bp[2], (int)bp[3], (int)bp[4], (int)bp[5], 0, 0};
            acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a[ks], bv, acc, 2, 2, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
        }
        const int qi = 16 * g + (lane & 15);
        const float2 qm = qmeta[qi];
        const float k2 = (1.015f + mt.y) * qm.y;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const uint32_t row = tile * 16u + 4u * (uint32_t)(lane >> 4) + (uint32_t)r;
            if (qi < n_q && row < n_rows && row < (uint32_t)BATCH_CAP)
                out[(size_t)qi * BATCH_CAP + row] = __builtin_fmaf(acc[r], s * qm.x, mt.y + k2);
        }
    }
}

void launch_f6_dense_scores(const void* d_f6, const void* d_meta, uint32_t n_rows, const float* d_q, int B, void* d_qf6, void* d_qmeta,
                            float* d_out, hipStream_t stream) {
    launch_prep_queries_f6(d_q, B, d_qf6, d_qmeta, stream);
    const uint32_t n = n_rows < (uint32_t)BATCH_CAP ? n_rows : (uint32_t)BATCH_CAP;
    const uint32_t tiles = (n + 15u) / 16u;
    if (tiles == 0) return;
    hipLaunchKernelGGL(f6_dense_scores_kernel, dim3(tiles), dim3(64), 0, stream, reinterpret_cast<const uint32_t*>(d_f6),
                       reinterpret_cast<const float2*>(d_meta), n_rows, reinterpret_cast<const uint32_t*>(d_qf6),
                       reinterpret_cast<const float2*>(d_qmeta), B, d_out);
}


// ------------------------------------------------------------------------------------------------
// The pass: 256 queries against every visited 16-row tile.  One wave per SIMD; wave w keeps the e2m3 images of queries
// 64 w .. 64 w + 63 in registers (4 groups x 3 k-steps x 6 dwords) for the whole kernel; a tile's three A fragments come straight
// from global memory (dwordx4 + dwordx2 per lane and k-step; the four waves of a workgroup read the same lines: L1 / L2), a ring
// of RING tiles ahead; 12 MFMAs per tile and wave.  D[row][query]: a lane holds 4 rows of ONE query per group, so the threshold
// test is a register compare against 4 per-lane thresholds made once per tile.
//   DENSE: every bound of the visited tiles -> dense[query][visit * 16 + row]  (the strided sample)
//   else:  (bound, row) of every pair above tau[query] -> the query's candidate buffer (segment blockIdx % 16), staged per wave in LDS
// ------------------------------------------------------------------------------------------------
constexpr int F6_RING = 3;
constexpr uint32_t F6_STAGE = 1344;  // staged hits per wave (a tile adds at most 16 x 64 = 1024: one flush check per tile)

template <bool DENSE>
__global__ __launch_bounds__(256) void scan_f6_pass_kernel(const uint32_t* __restrict__ x, const float2* __restrict__ meta, uint32_t n_rows,
                                                            uint32_t stride, uint32_t n_visits, const uint32_t* __restrict__ qf6,
                                                            const float2* __restrict__ qmeta, int n_q, const float* __restrict__ tau,
                                                            uint32_t* __restrict__ cnt, uint2* __restrict__ cand, uint32_t seg_cap,
                                                            float* __restrict__ dense, uint32_t stagger, int mark_lost) {
    __shared__ uint32_t stage[4][F6_STAGE][3];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t G = gridDim.x;
    // This workgroup's visits: blockIdx.x, + G, ... (n_mine of them); wave w walks them starting `stagger` x w positions in, so
    // that the four waves have DIFFERENT tiles in flight (four waves on one tile at a time keep 18 KB in flight per CU: the pass is
    // then bound by memory latency, 15 ms per 100 M x 256) while a tile still comes from HBM once and from L2 / MALL three times.
    const uint32_t n_mine = n_visits > blockIdx.x ? (n_visits - blockIdx.x + G - 1u) / G : 0u;
    const uint32_t p0 = n_mine ? ((uint32_t)wave * stagger) % n_mine : 0u;
    auto visit_of = [&](uint32_t pos) __attribute__((always_inline)) -> uint32_t {  // pos may run past n_mine: wraps (valid memory)
        if (n_mine == 0u) return 0u;
        uint32_t q = p0 + pos;
        q = q >= n_mine ? q - n_mine : q;
        q = q >= n_mine ? q % n_mine : q;
        return blockIdx.x + q * G;
    };
    uint32_t vis = 0;  // position in the walk
    // The rows first: a ring of F6_RING tiles (3 x (16 + 8) B per lane each), asm loads under a hand-counted vmcnt.  hipcc cannot
    // count this ring itself: at the head of the unrolled loop its waitcnt pass merges the states of the two ways in and waits for
    // EVERYTHING (vmcnt(0)) before the first tile of every iteration — the ring drained every four tiles and a pass over 100 M
    // rows took 14-16 ms against the 5.1 ms its mix sustains bare (profiles/r04/f6_ab_*.log).  The MFMA's A operand is SIX
    // consecutive registers filled by two loads; the slot's registers are therefore moved into the operand by the same asm
    // statement that waits for them (six v_mov in the shadow of the MFMAs) — with the wait as a separate statement, hipcc copied the
    // slot in FRONT of it (stale data, wrong answers).  Loads return in order: when tile v is consumed, the 3 x 6 loads of the tiles
    // v + 1 .. v + 3 may still be in flight.
    u32x4 alo[F6_RING][3];
    u32x2 ahi[F6_RING][3];
    // (plain loads, not nt: the four waves of a workgroup read the same lines one after the other — the first from HBM, the others
    // should find them in L2)
    // (scalar base + per-lane 32-bit offset: the tile's address costs no vector instruction)
#define DAWN_F6_LD4(DST, PTR, OFF) asm volatile("global_load_dwordx4 %0, %1, %2 offset:" #OFF : "=v"(DST) : "v"(off4), "s"(PTR))
#define DAWN_F6_LD2(DST, PTR, OFF) asm volatile("global_load_dwordx2 %0, %1, %2 offset:" #OFF : "=v"(DST) : "v"(off2), "s"(PTR))
    const uint32_t off4 = (uint32_t)lane * 16u, off2 = 1024u + (uint32_t)lane * 8u;  // byte offsets inside k-step 0
    auto load_tile = [&](int slot, uint32_t v) __attribute__((always_inline)) {
        const uint32_t* p4 = x + (size_t)(v < n_visits ? v * stride : 0u) * F6_TILE_DW;  // (wave-uniform)
        const uint32_t* p2 = p4;
        DAWN_F6_LD4(alo[slot][0], p4, 0);
        DAWN_F6_LD2(ahi[slot][0], p2, 0);
        DAWN_F6_LD4(alo[slot][1], p4, 1536);
        DAWN_F6_LD2(ahi[slot][1], p2, 1536);
        DAWN_F6_LD4(alo[slot][2], p4, 3072);
        DAWN_F6_LD2(ahi[slot][2], p2, 3072);
    };
    // wait for the slot's loads (N younger loads may stay in flight) and move fragment ks into six registers of their own
    auto take = [&](int slot, int ks, auto n_tag) __attribute__((always_inline)) -> i32x8_t {
        constexpr int N = decltype(n_tag)::value;
        int o0, o1, o2, o3, o4, o5;
        if constexpr (N >= 0)
            asm volatile("s_waitcnt vmcnt(%12)\n\tv_mov_b32 %0, %6\n\tv_mov_b32 %1, %7\n\tv_mov_b32 %2, %8\n\tv_mov_b32 %3, %9\n\t"
                         "v_mov_b32 %4, %10\n\tv_mov_b32 %5, %11"
                         : "=&v"(o0), "=&v"(o1), "=&v"(o2), "=&v"(o3), "=&v"(o4), "=&v"(o5)
                         : "v"(alo[slot][ks].x), "v"(alo[slot][ks].y), "v"(alo[slot][ks].z), "v"(alo[slot][ks].w), "v"(ahi[slot][ks].x),
                           "v"(ahi[slot][ks].y), "n"(N));
        else
            asm volatile("v_mov_b32 %0, %6\n\tv_mov_b32 %1, %7\n\tv_mov_b32 %2, %8\n\tv_mov_b32 %3, %9\n\tv_mov_b32 %4, %10\n\t"
                         "v_mov_b32 %5, %11"
                         : "=&v"(o0), "=&v"(o1), "=&v"(o2), "=&v"(o3), "=&v"(o4), "=&v"(o5)
                         : "v"(alo[slot][ks].x), "v"(alo[slot][ks].y), "v"(alo[slot][ks].z), "v"(alo[slot][ks].w), "v"(ahi[slot][ks].x),
                           "v"(ahi[slot][ks].y));
        return i32x8_t{o0, o1, o2, o3, o4, o5, 0, 0};
    };
#pragma unroll
    for (int d = 0; d < F6_RING; ++d) load_tile(d, visit_of((uint32_t)d));
    // the wave's query images and per-lane constants
    i32x8_t bq[4][3];
    float thrA[4], thrB[4], thrC[4], sq_l[4], dqn_l[4], tau_l[4];
#pragma unroll
    for (int gg = 0; gg < 4; ++gg) {
        const int g = 4 * wave + gg;
        const int qi = 16 * g + (lane & 15);
#pragma unroll
        for (int ks = 0; ks < 3; ++ks) {
            const uint32_t* bp = qf6 + (((size_t)g * 3 + ks) * 64 + lane) * 6;
            const u32x2 b01 = *reinterpret_cast<const u32x2*>(bp), b23 = *reinterpret_cast<const u32x2*>(bp + 2),
                        b45 = *reinterpret_cast<const u32x2*>(bp + 4);
            bq[gg][ks] = i32x8_t{(int)b01.x, (int)b01.y, (int)b23.x, (int)b23.y, (int)b45.x, (int)b45.y, 0, 0};
        }
        const float2 qm = qmeta[qi];
        sq_l[gg] = qm.x;
        dqn_l[gg] = qm.y;
        tau_l[gg] = (!DENSE && qi < n_q) ? tau[qi] : POS_INF;
        thrA[gg] = 1.0f / qm.x;
        thrB[gg] = tau_l[gg] - 1.015f * qm.y;
        thrC[gg] = -(1.0f + qm.y);
    }
    uint32_t n_stage = 0;  // wave-uniform
    uint32_t(*st)[3] = stage[wave];
    const uint32_t seg = blockIdx.x % (uint32_t)BATCH_CAND_SEGS;
    auto flush = [&]() __attribute__((always_inline)) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        for (uint32_t i = lane; i < n_stage; i += 64u) {
            const uint32_t ub = st[i][0], row = st[i][1], qi = st[i][2];
            const uint32_t pos = atomicAdd(&cnt[(size_t)qi * BATCH_CAND_SEGS + seg], 1u);
            if (pos < seg_cap) cand[((size_t)qi * BATCH_CAND_SEGS + seg) * seg_cap + pos] = uint2{ub, row};
        }
        n_stage = 0;
    };

    // Two loops.  The INNER one — up to 16 x F6_RING tiles — touches no memory but the ring's loads and the wave's LDS stage, so
    // that hipcc can count its waits (a global atomic anywhere in a loop makes its waitcnt pass drain the ring, vmcnt(0), at
    // every tile of that loop); the stage is flushed between two runs of it.  A tile whose hits do not fit the stage any more
    // (> 1000 pairs within a few tiles: no data does that) drops them and marks the wave's queries as overflowed — the tail then
    // sends them to the ladder.
    bool lost = false;
    while (vis < n_mine) {
      for (int it = 0; it < 16 && vis < n_mine && (DENSE || n_stage <= F6_STAGE - 1024u); ++it, vis += (uint32_t)F6_RING) {
#pragma unroll
        for (int d = 0; d < F6_RING; ++d) {
            if (vis + (uint32_t)d >= n_mine) break;  // (workgroup-uniform)
            const uint32_t v = visit_of(vis + (uint32_t)d);
            const uint32_t tile = v * stride;
            const float2 mt = meta[tile];
            i32x8_t av[3];
            av[0] = take(d, 0, std::integral_constant<int, 6 * (F6_RING - 1)>());  // (in-order returns: the whole tile has landed)
            av[1] = take(d, 1, std::integral_constant<int, -1>());
            av[2] = take(d, 2, std::integral_constant<int, -1>());
            load_tile(d, visit_of(vis + (uint32_t)d + (uint32_t)F6_RING));
            f32x4 acc[4];
#pragma unroll
            for (int gg = 0; gg < 4; ++gg) {
                acc[gg] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < 3; ++ks)
                    acc[gg] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(av[ks], bq[gg][ks], acc[gg], 2, 2, 0, 0x7F7F7F7F, 0,
                                                                                0x7F7F7F7F);
            }
            const float s = __builtin_amdgcn_rcpf(mt.x);
            const uint32_t row0 = tile * 16u + 4u * (uint32_t)(lane >> 4);
            if constexpr (DENSE) {
#pragma unroll
                for (int gg = 0; gg < 4; ++gg) {
                    const int qi = 16 * (4 * wave + gg) + (lane & 15);
                    const float k2 = (1.015f + mt.y) * dqn_l[gg];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const uint32_t si = v * 16u + 4u * (uint32_t)(lane >> 4) + (uint32_t)r;
                        if (qi < n_q && si < (uint32_t)BATCH_CAP)
                            dense[(size_t)qi * BATCH_CAP + si] =
                                row0 + (uint32_t)r < n_rows ? __builtin_fmaf(acc[gg][r], s * sq_l[gg], mt.y + k2) : NEG_INF;
                    }
                }
            } else {
                // acc > thr  <=  ub = acc s s_q + E + (1.015 + E) dqn > tau   (thr a little low: hits are re-tested on ub itself)
                bool any = false;
                float thr[4];
#pragma unroll
                for (int gg = 0; gg < 4; ++gg) {
                    // (acc is a multiple of 1/64; the f32 rounding of this expression is < 1e-4 at |thr| < 1000: 0.02 below is safe)
                    thr[gg] = __builtin_fmaf(__builtin_fmaf(mt.y, thrC[gg], thrB[gg]), mt.x * thrA[gg], -0.02f);
                    any = any || fmaxf(fmaxf(acc[gg][0], acc[gg][1]), fmaxf(acc[gg][2], acc[gg][3])) > thr[gg];
                }
                if (__any(any)) {
#pragma unroll
                    for (int gg = 0; gg < 4; ++gg) {
                        const uint32_t qi = (uint32_t)(16 * (4 * wave + gg) + (lane & 15));
                        const float g1 = s * sq_l[gg], g0 = mt.y + (1.015f + mt.y) * dqn_l[gg];
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const float ub = __builtin_fmaf(acc[gg][r], g1, g0);
                            const bool hit = acc[gg][r] > thr[gg] && ub > tau_l[gg] && row0 + (uint32_t)r < n_rows;
                            const unsigned long long m = __ballot(hit);
                            if (m) {
                                if (hit) {
                                    const uint32_t slot = n_stage + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
                                    if (slot < F6_STAGE) {
                                        st[slot][0] = __builtin_bit_cast(uint32_t, ub);
                                        st[slot][1] = row0 + (uint32_t)r;
                                        st[slot][2] = qi;
                                    } else {
                                        lost = true;
                                    }
                                }
                                n_stage += (uint32_t)__popcll(m);
                                if (n_stage > F6_STAGE) n_stage = F6_STAGE;
                            }
                        }
                    }
                }
            }
        }
      }
      if (!DENSE && n_stage > 0u) flush();
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (the ring's last loads: nothing may be in flight when the wave ends)
#undef DAWN_F6_LD4
#undef DAWN_F6_LD2
    // (a SAMPLING pass does not mark: hits dropped from a sample only move the threshold read from it, and an inflated counter
    // would make tau_select read entries nobody wrote)
    if (!DENSE && mark_lost && __any(lost)) {
        const uint32_t qi = (uint32_t)(64 * wave + lane);
        if ((int)qi < n_q) atomicAdd(&cnt[(size_t)qi * BATCH_CAND_SEGS + seg], seg_cap + 1u);
    }
}

// ------------------------------------------------------------------------------------------------
// The full pass, staged through LDS.  The kernel above has each of its four waves load every tile itself: the rows then cross
// L2 -> L1 -> registers four times (115 GB per 100 M rows) with one wave per SIMD and nothing to hide a stall behind — 16-17 ms per
// 100 M x 256 whatever the ring, the load form or the hit rate (profiles/r04/f6_ab_*.log); it stays for the strided samples.
// Here a workgroup of EIGHT waves takes groups of 8 consecutive tiles (36864 B, contiguous): every thread copies 4 x 16 B + 8 B of
// a group global -> registers two groups ahead, registers -> LDS (double buffer) in the MIDDLE of the work on the group before it,
// one barrier per group; a tile's fragments are read from LDS by the four waves that hold the four quarters of the batch: waves
// 0-3 take the even tiles of a group, waves 4-7 the odd ones, two waves per SIMD.  Every byte of the shadow is read from HBM once
// and crosses L2 once.  What each step was worth, per 100 M rows x 256 queries (profiles/r04/f6_ab_*_v5..v7*.log):
//     16.0 ms  the register-ring kernel above
//      9.8     through LDS, one group in flight
//      9.1     two groups in flight (hand-counted vmcnt)
//      8.0     per-query-group ballots in the hit path
//      7.45    tiles software-pipelined inside a wave, landing moved into the middle of a group, compares into scalar masks
// against 8.9-9.3 ms of the int8 pass on the same index.  The copy alone (no MFMA, no tests) runs at 7 TB/s (4.1 ms); with the
// MFMAs 6.5 ms; a wave's cycles (MODE 4): 5-6 % waiting for its loads, 13-23 % at the barrier (the older wave of a SIMD gets there
// first), the rest ~470 cycles per tile and wave for 12 MFMAs (192 cycles of matrix pipe) and ~46 vector instructions — two waves
// per SIMD do not cover each other's dependency stalls completely, and 226 VGPRs do not leave room for a third.  The barrier share
// is slack, not loss: two workgroups of four waves per CU (groups of four tiles, two barrier domains) take exactly as long
// (profiles/r04/f6_ab_100M_v9_two_workgroups_per_cu_no_gain.log); so do scheduling hints that interleave MFMAs and tests.
// ------------------------------------------------------------------------------------------------
constexpr int F6L_GROUP = 8;                                       // tiles per group
constexpr uint32_t F6L_GROUP_BYTES = F6L_GROUP * F6_TILE_DW * 4u;  // 36864
constexpr uint32_t F6L_STAGE = 448;                                // staged hits per wave
constexpr uint32_t F6L_FLUSH_AT = 192;                             // flush between groups above this
struct F6PassLds {
    uint32_t tiles[2][F6L_GROUP_BYTES / 4];
    uint32_t stage[8][F6L_STAGE][3];
};

template <int MODE>  // 0: the pass; timing experiments: 1: no threshold tests (wrong results), 4: a few waves print where their cycles went
__global__ __launch_bounds__(512) void scan_f6_pass_lds_kernel(const uint32_t* __restrict__ x, const float2* __restrict__ meta,
                                                                uint32_t n_rows, uint32_t n_tiles, const uint32_t* __restrict__ qf6,
                                                                const float2* __restrict__ qmeta, int n_q, const float* __restrict__ tau,
                                                                uint32_t* __restrict__ cnt, uint2* __restrict__ cand, uint32_t seg_cap) {
    extern __shared__ __attribute__((aligned(16))) unsigned char f6l_raw[];
    F6PassLds& L = *reinterpret_cast<F6PassLds*>(f6l_raw);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int qw = wave & 3, half = wave >> 2;
    const uint32_t G = gridDim.x;
    const uint32_t n_groups = (n_tiles + (uint32_t)F6L_GROUP - 1u) / (uint32_t)F6L_GROUP;
    // the wave's query images and per-lane constants
    i32x8_t bq[4][3];
    float thrP[4], thrQ[4], sq_l[4], dqn_l[4];
#pragma unroll
    for (int gg = 0; gg < 4; ++gg) {
        const int g = 4 * qw + gg;
        const int qi = 16 * g + (lane & 15);
#pragma unroll
        for (int ks = 0; ks < 3; ++ks) {
            const uint32_t* bp = qf6 + (((size_t)g * 3 + ks) * 64 + lane) * 6;
            const u32x2 b01 = *reinterpret_cast<const u32x2*>(bp), b23 = *reinterpret_cast<const u32x2*>(bp + 2),
                        b45 = *reinterpret_cast<const u32x2*>(bp + 4);
            bq[gg][ks] = i32x8_t{(int)b01.x, (int)b01.y, (int)b23.x, (int)b23.y, (int)b45.x, (int)b45.y, 0, 0};
        }
        const float2 qm = qmeta[qi];
        sq_l[gg] = qm.x;
        dqn_l[gg] = qm.y;
        const float tau_q = qi < n_q ? tau[qi] : POS_INF;
        thrP[gg] = (tau_q - 1.015f * qm.y) / qm.x;
        thrQ[gg] = -(1.0f + qm.y) / qm.x;
    }
    uint32_t n_stage = 0;  // wave-uniform
    uint32_t(*st)[3] = L.stage[wave];
    const uint32_t seg = blockIdx.x % (uint32_t)BATCH_CAND_SEGS;
    bool lost = false;

    // The copy of a group: thread t moves the 16-B pieces t, t + 512, t + 1024, t + 1536 and the 8-B piece 4096 + t / 2 (the shadow
    // is padded by 8 tiles of zeros: a group that starts inside the index never reads outside the allocation), and lane l of every
    // wave the metadata of tile l & 7.  TWO groups are in flight per workgroup (72 KB per CU, 18 MB on the chip: with one group the
    // pass waits for memory latency, 9.8 ms per 100 M x 256): two register sets, asm loads under a hand-counted vmcnt — hipcc's
    // waitcnt pass drains a ring like this at the loop header (see the kernel above).  The asm statement that waits for a set also
    // writes it to LDS and copies the metadata out, so nothing the compiler schedules can read a register before its data is there.
    const uint32_t lds0 = (uint32_t)(size_t)(__attribute__((address_space(3))) unsigned char*)f6l_raw;
    const uint32_t o4 = (uint32_t)threadIdx.x * 16u, o2 = 32768u + (uint32_t)threadIdx.x * 8u, om = (uint32_t)(lane & 7) * 8u;
    const uint32_t o4b = o4 + 8192u, o4c = o4 + 16384u, o4d = o4 + 24576u;
    u32x4 pa0, pa1, pa2, pa3, pb0, pb1, pb2, pb3;
    u32x2 pa4, pb4, pam, pbm;
#define DAWN_F6L_PREFETCH(P0, P1, P2, P3, P4, PM, GRP)                                                                         \
    do {                                                                                                                      \
        const uint32_t g_ = (GRP) < n_groups ? (GRP) : n_groups - 1u; /* (past the end: any valid group, never used) */        \
        const unsigned char* xb_ = reinterpret_cast<const unsigned char*>(x) + (size_t)g_ * F6L_GROUP_BYTES;                  \
        const float2* mb_ = meta + (size_t)g_ * F6L_GROUP;                                                                    \
        asm volatile("global_load_dwordx4 %0, %6, %11 nt\n\tglobal_load_dwordx4 %1, %7, %11 nt\n\t"                           \
                     "global_load_dwordx4 %2, %8, %11 nt\n\tglobal_load_dwordx4 %3, %9, %11 nt\n\t"                           \
                     "global_load_dwordx2 %4, %10, %11 nt\n\tglobal_load_dwordx2 %5, %12, %13"                                \
                     : "=&v"(P0), "=&v"(P1), "=&v"(P2), "=&v"(P3), "=&v"(P4), "=&v"(PM)                                       \
                     : "v"(o4), "v"(o4b), "v"(o4c), "v"(o4d), "v"(o2), "s"(xb_), "v"(om), "s"(mb_)                            \
                     : "memory");                                                                                             \
    } while (0)
    // wait until only the OTHER set's six loads are in flight, set -> LDS buffer at byte offset BUF, metadata -> (MX, MY)
#define DAWN_F6L_LAND(P0, P1, P2, P3, P4, PM, BUF, MX, MY)                                                                     \
    do {                                                                                                                      \
        const uint32_t a4_ = lds0 + (BUF) + o4, a2_ = lds0 + (BUF) + (uint32_t)threadIdx.x * 8u;                              \
        asm volatile("s_waitcnt vmcnt(6)\n\tds_write_b128 %2, %4\n\tds_write_b128 %2, %5 offset:8192\n\t"                    \
                     "ds_write_b128 %2, %6 offset:16384\n\tds_write_b128 %2, %7 offset:24576\n\t"                            \
                     "ds_write_b64 %3, %8 offset:32768\n\tv_mov_b32 %0, %9\n\tv_mov_b32 %1, %10"                             \
                     : "=&v"(MX), "=&v"(MY)                                                                                   \
                     : "v"(a4_), "v"(a2_), "v"(P0), "v"(P1), "v"(P2), "v"(P3), "v"(P4), "v"(PM.x), "v"(PM.y)                  \
                     : "memory");                                                                                             \
    } while (0)
    // the workgroup's barrier, one per group: every wave has written its share of the NEXT group (in the middle of its work on the
    // current one, where the wait for the data and the LDS writes cost nothing) and is done reading the current one
#define DAWN_F6L_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

    // A group's work for this wave: its four tiles, software-pipelined — the fragments of tile k + 1 are read from LDS and the
    // thresholds of tile k - 1 tested while the 12 MFMAs of tile k run (the matrix pipe takes one every 16 cycles: three or four
    // vector instructions fit between two of them; in program order tests-after-MFMAs a wave leaves the pipe idle for the ~180
    // cycles of its tests, and with two waves per SIMD nothing else fills them).  `mid` runs after the second tile: the landing of
    // the next group.
    auto compute = [&](const unsigned char* tb, uint32_t grp, int cmx, int cmy, auto mid) __attribute__((always_inline)) {
        auto read_tile = [&](int t, i32x8_t (&av)[3]) __attribute__((always_inline)) {
#pragma unroll
            for (int ks = 0; ks < 3; ++ks) {
                const unsigned char* f = tb + (uint32_t)t * (F6_TILE_DW * 4u) + (uint32_t)ks * (F6_KS_DW * 4u);
                const u32x4 lo = *reinterpret_cast<const u32x4*>(f + lane * 16);
                const u32x2 hi = *reinterpret_cast<const u32x2*>(f + 1024 + lane * 8);
                av[ks] = i32x8_t{(int)lo.x, (int)lo.y, (int)lo.z, (int)lo.w, (int)hi.x, (int)hi.y, 0, 0};
            }
        };
        constexpr int NT = F6L_GROUP / 2;
        i32x8_t av[2][3];
        f32x4 acc[2][4];
        read_tile(half, av[0]);
#pragma unroll
        for (int k = 0; k <= NT; ++k) {
            if (k + 1 < NT) read_tile(half + 2 * (k + 1), av[(k + 1) & 1]);
            if (k < NT) {
#pragma unroll
                for (int gg = 0; gg < 4; ++gg) acc[k & 1][gg] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < 3; ++ks)  // (k-step outside: four independent accumulators back to back)
#pragma unroll
                    for (int gg = 0; gg < 4; ++gg)
                        acc[k & 1][gg] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(av[k & 1][ks], bq[gg][ks], acc[k & 1][gg], 2, 2, 0,
                                                                                           0x7F7F7F7F, 0, 0x7F7F7F7F);
            }
            if (k >= 1) {
                const int t = half + 2 * (k - 1);
                const f32x4(&ac)[4] = acc[(k - 1) & 1];
                const uint32_t tile = grp * (uint32_t)F6L_GROUP + (uint32_t)t;
                const float2 mt = float2{__builtin_bit_cast(float, __builtin_amdgcn_readlane(cmx, t)),
                                         __builtin_bit_cast(float, __builtin_amdgcn_readlane(cmy, t))};
                // acc > thr  <=>  ub = acc s s_q + E + (1.015 + E) dqn > tau:  thr = (tau - E - (1.015 + E) dqn) / (s s_q) - 0.02
                //   = (1 / s) P + (E / s) Q - 0.02 with P, Q per query (acc is a multiple of 1/64 and |thr| < 1e4: the f32 rounding of
                // this expression is < 1e-3, 0.02 is safe).  Sixteen compares straight into scalar masks — no maxima, no ballots:
                // the fast path is 8 + 16 vector instructions per tile.  (The zero padding behind the last tile has thr = -0.02 <
                // acc = 0: its rows fail the row < n_rows test below.)
                const float sx = mt.x, sxy = mt.x * mt.y;
                unsigned long long mk[4][4], any_m = 0ull;
                float thr[4];
#pragma unroll
                for (int gg = 0; gg < 4; ++gg) {
                    thr[gg] = __builtin_fmaf(sx, thrP[gg], __builtin_fmaf(sxy, thrQ[gg], -0.02f));
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        mk[gg][r] = __ballot(ac[gg][r] > thr[gg]);
                        any_m |= mk[gg][r];
                    }
                }
                if (MODE != 1 && any_m != 0ull) {  // (one tile in four or five: ~0.25 pairs per tile and wave at the default target)
                    const float s = __builtin_amdgcn_rcpf(mt.x);
                    const uint32_t row0 = tile * 16u + 4u * (uint32_t)(lane >> 4);
#pragma unroll
                    for (int gg = 0; gg < 4; ++gg) {
                        if ((mk[gg][0] | mk[gg][1] | mk[gg][2] | mk[gg][3]) == 0ull) continue;
                        const uint32_t qi = (uint32_t)(16 * (4 * qw + gg) + (lane & 15));
                        const float g1 = s * sq_l[gg], g0 = mt.y + (1.015f + mt.y) * dqn_l[gg];
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            if (mk[gg][r] == 0ull) continue;
                            const bool hit = ac[gg][r] > thr[gg] && row0 + (uint32_t)r < n_rows;
                            const unsigned long long m = __ballot(hit);
                            if (hit) {
                                const uint32_t slot = n_stage + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
                                if (slot < F6L_STAGE) {
                                    st[slot][0] = __builtin_bit_cast(uint32_t, __builtin_fmaf(ac[gg][r], g1, g0));
                                    st[slot][1] = row0 + (uint32_t)r;
                                    st[slot][2] = qi;
                                } else {
                                    lost = true;
                                }
                            }
                            n_stage += (uint32_t)__popcll(m);
                            if (n_stage > F6L_STAGE) n_stage = F6L_STAGE;
                        }
                    }
                }
                if constexpr (MODE == 1) {
                    const float sum = (ac[0][0] + ac[1][1]) + (ac[2][2] + ac[3][3]);
                    if (sum == 1.2345e30f) lost = true;
                }
            }
            if (k == 1) mid();
        }
    };
    auto flush = [&]() __attribute__((always_inline)) {
        for (uint32_t i = lane; i < n_stage; i += 64u) {
            const uint32_t ub = st[i][0], row = st[i][1], qi = st[i][2];
            const uint32_t pos = atomicAdd(&cnt[(size_t)qi * BATCH_CAND_SEGS + seg], 1u);
            if (pos < seg_cap) cand[((size_t)qi * BATCH_CAND_SEGS + seg) * seg_cap + pos] = uint2{ub, row};
        }
        n_stage = 0;
        // (stores and atomics share the loads' counter and return out of order with them: start the hand count from zero again)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    };

    uint32_t grp = blockIdx.x;
    if (grp < n_groups) {  // (workgroup-uniform)
        const unsigned char* tb0 = reinterpret_cast<const unsigned char*>(L.tiles[0]);
        const unsigned char* tb1 = reinterpret_cast<const unsigned char*>(L.tiles[1]);
        int mx0, my0, mx1, my1;  // metadata of the group in buffer 0 / 1
        DAWN_F6L_PREFETCH(pa0, pa1, pa2, pa3, pa4, pam, grp);
        DAWN_F6L_PREFETCH(pb0, pb1, pb2, pb3, pb4, pbm, grp + G);
        DAWN_F6L_LAND(pa0, pa1, pa2, pa3, pa4, pam, 0u, mx0, my0);
        DAWN_F6L_PREFETCH(pa0, pa1, pa2, pa3, pa4, pam, grp + 2u * G);
        DAWN_F6L_BARRIER();
        // MODE 4 (timing experiment): cycles this wave spends waiting for its loads (the landing) and at the barrier
        unsigned long long t_land = 0, t_bar = 0, t_all = 0, t0 = 0, n_it = 0;
#define DAWN_F6L_T() (MODE == 4 ? __builtin_readcyclecounter() : 0ull)
        t_all = DAWN_F6L_T();
        while (true) {
            compute(tb0, grp, mx0, my0, [&]() __attribute__((always_inline)) {
                t0 = DAWN_F6L_T();
                DAWN_F6L_LAND(pb0, pb1, pb2, pb3, pb4, pbm, F6L_GROUP_BYTES, mx1, my1);
                t_land += DAWN_F6L_T() - t0;
                DAWN_F6L_PREFETCH(pb0, pb1, pb2, pb3, pb4, pbm, grp + 3u * G);
            });
            t0 = DAWN_F6L_T();
            DAWN_F6L_BARRIER();
            t_bar += DAWN_F6L_T() - t0;
            ++n_it;
            grp += G;
            if (grp >= n_groups) break;
            compute(tb1, grp, mx1, my1, [&]() __attribute__((always_inline)) {
                t0 = DAWN_F6L_T();
                DAWN_F6L_LAND(pa0, pa1, pa2, pa3, pa4, pam, 0u, mx0, my0);
                t_land += DAWN_F6L_T() - t0;
                DAWN_F6L_PREFETCH(pa0, pa1, pa2, pa3, pa4, pam, grp + 3u * G);
            });
            t0 = DAWN_F6L_T();
            DAWN_F6L_BARRIER();
            t_bar += DAWN_F6L_T() - t0;
            ++n_it;
            grp += G;
            if (grp >= n_groups) break;
            if (n_stage > F6L_FLUSH_AT) flush();  // (a wave's own decision: no barrier inside)
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (the ring's last loads: nothing may be in flight when the wave ends)
        if constexpr (MODE == 4) {
            t_all = DAWN_F6L_T() - t_all;
            if (lane == 0 && (blockIdx.x == 0 || blockIdx.x == 131) && (wave == 0 || wave == 5))
                printf("f6 pass timing: block %d wave %d: %llu groups, shader cycles total %llu, landing %llu, barrier %llu\n",
                       (int)blockIdx.x, wave, n_it, t_all, t_land, t_bar);
        }
#undef DAWN_F6L_T
        if (n_stage > 0u) flush();
    }
#undef DAWN_F6L_PREFETCH
#undef DAWN_F6L_LAND
#undef DAWN_F6L_BARRIER
    if (__any(lost)) {  // (a burst of > 256 pairs within four tiles: the tail sends the wave's queries to the ladder)
        const uint32_t qi = (uint32_t)(64 * qw + lane);
        if ((int)qi < n_q) atomicAdd(&cnt[(size_t)qi * BATCH_CAND_SEGS + seg], seg_cap + 1u);
    }
}

// ------------------------------------------------------------------------------------------------
// Second stage: every survivor of the FP6 pass gets the int8 shadow's (one-image) bound as well — a lane per candidate, its 24
// 16-B pieces of the int8 sub-tile against the query's int8 image in LDS, v_dot4_i32_i8 — and the ones whose tighter bound
// still exceeds the threshold go to the query's ordinary candidate buffer, where select_rescore_kernel finds them.
// grid (B, F6_REFINE_BLOCKS); big: [B][16][seg_cap_big] with counters cnt_big (zeroed here for the next search).
// ------------------------------------------------------------------------------------------------


__global__ __launch_bounds__(256) void f6_refine_kernel(const uint2* __restrict__ big, uint32_t* __restrict__ cnt_big, uint32_t seg_cap_big,
                                                         const unsigned char* __restrict__ x8, const float2* __restrict__ meta8,
                                                         const signed char* __restrict__ qi8, const float2* __restrict__ qm8,
                                                         float* __restrict__ tau, uint32_t* __restrict__ cnt, uint2* __restrict__ cand) {
    __shared__ __attribute__((aligned(16))) signed char sh_img[EM];
    __shared__ uint32_t sh_cnt;
    const int b = blockIdx.x, seg = blockIdx.y;  // one block per (query, segment of the big buffer)
    for (int i = threadIdx.x; i < EM / 4; i += blockDim.x)
        reinterpret_cast<int*>(sh_img)[i] = reinterpret_cast<const int*>(qi8 + (size_t)b * EM)[i];
    if (threadIdx.x == 0) {
        // (a counter past the capacity: survivors were dropped — or a wave of the pass marked the query as lost, in which case the
        // entries behind the ones it wrote are NOT survivors of this search: nothing of such a segment is read)
        const uint32_t c = cnt_big[(size_t)b * BATCH_CAND_SEGS + seg];
        sh_cnt = c <= seg_cap_big ? c : 0u;
    }
    __syncthreads();
    const uint32_t n = sh_cnt;
    const float2 qm = qm8[b];  // {s_q, K2}
    const float t = tau[b];
    const uint2* src = big + ((size_t)b * BATCH_CAND_SEGS + seg) * seg_cap_big;
    const i32x4_t* img = reinterpret_cast<const i32x4_t*>(sh_img);
    for (uint32_t e = threadIdx.x; e < n; e += blockDim.x) {
        const uint2 v = src[e];
        const float ub6 = __builtin_bit_cast(float, v.x);
        const uint32_t row = v.y;
        const uint32_t t8 = row >> 5, r8 = row & 31u;
        const i32x4_t* xs = reinterpret_cast<const i32x4_t*>(x8 + (size_t)t8 * 12288u) + r8;
        int c = 0;
#pragma unroll 6
        for (int j = 0; j < 24; ++j) {  // piece j = fragment j / 2, half j % 2: k = 16 j .. 16 j + 15
            const i32x4_t xv = xs[(j >> 1) * 64 + (j & 1) * 32];
            const i32x4_t hv = img[j];
            c = __builtin_amdgcn_sdot4(xv[0], hv[0], c, false);
            c = __builtin_amdgcn_sdot4(xv[1], hv[1], c, false);
            c = __builtin_amdgcn_sdot4(xv[2], hv[2], c, false);
            c = __builtin_amdgcn_sdot4(xv[3], hv[3], c, false);
        }
        const float2 m8 = meta8[t8];
        const float ub8 = __builtin_fmaf((float)c, __builtin_amdgcn_rcpf(m8.x) * qm.x, m8.y + qm.y);
        const float ub = fminf(ub6, ub8);
        if (ub > t) {
            const uint32_t pos = atomicAdd(&cnt[(size_t)b * BATCH_CAND_SEGS + seg], 1u);
            // (beyond the segment: dropped, and the counter tells select_rescore_kernel that candidates were lost)
            if (pos < (uint32_t)BATCH_CAP / (uint32_t)BATCH_CAND_SEGS)
                cand[((size_t)b * BATCH_CAND_SEGS + seg) * ((uint32_t)BATCH_CAP / (uint32_t)BATCH_CAND_SEGS) + pos] =
                    uint2{__builtin_bit_cast(uint32_t, ub), row};
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        // a big segment that overflowed lost candidates: no certificate may hold for this query — its threshold becomes +inf ("rows that
        // never became candidates score <= tau" is then never enough), the tail flags it and the ladder answers.  (Not by inflating
        // the ordinary counter: the tail would read entries nobody wrote.)
        if (cnt_big[(size_t)b * BATCH_CAND_SEGS + seg] > seg_cap_big) tau[b] = POS_INF;
        cnt_big[(size_t)b * BATCH_CAND_SEGS + seg] = 0u;
    }
}

// The same on an f32 index, from the rows themselves.  A survivor's int8 image is 24 pieces of 16 B in 24 different 128-B lines
// (3 KB of traffic per candidate: 1.4-2 ms per 100 M x 256 at ~12 k survivors per query); its f32 row is 1536 contiguous bytes.
// Sixteen lanes per candidate, six 16-B pieces each against the query's in registers, a butterfly over the sixteen lanes; the
// f32 dot product is within gamma_384 x 1.0201 = 2.34e-5 of the real one whatever the order of the sum: + 3e-5 makes it a bound.
__global__ __launch_bounds__(256) void f6_refine_rows_kernel(const uint2* __restrict__ big, uint32_t* __restrict__ cnt_big,
                                                              uint32_t seg_cap_big, const f32x4* __restrict__ x,
                                                              const float* __restrict__ q, float* __restrict__ tau,
                                                              uint32_t* __restrict__ cnt, uint2* __restrict__ cand) {
    const int b = blockIdx.x, seg = blockIdx.y;
    const int part = threadIdx.x & 15, grp = threadIdx.x >> 4;
    f32x4 qv[6];
#pragma unroll
    for (int j = 0; j < 6; ++j) qv[j] = reinterpret_cast<const f32x4*>(q + (size_t)b * EM)[part + 16 * j];
    const uint32_t c0 = cnt_big[(size_t)b * BATCH_CAND_SEGS + seg];
    const uint32_t n = c0 <= seg_cap_big ? c0 : 0u;  // (past the capacity: dropped survivors or a "lost" mark — see f6_refine_kernel)
    const float t = tau[b];
    const uint2* src = big + ((size_t)b * BATCH_CAND_SEGS + seg) * seg_cap_big;
    constexpr uint32_t seg_small = (uint32_t)BATCH_CAP / (uint32_t)BATCH_CAND_SEGS;
    for (uint32_t e0 = 0; e0 < n; e0 += 32u) {  // two candidates per 16 lanes and step: 12 loads in flight per lane
        const uint32_t ea = e0 + (uint32_t)grp, eb = ea + 16u;
        const uint2 va = ea < n ? src[ea] : uint2{0u, 0u}, vb = eb < n ? src[eb] : uint2{0u, 0u};
        const f32x4* ra = x + (size_t)va.y * ROW_F4 + part;
        const f32x4* rb = x + (size_t)vb.y * ROW_F4 + part;
        f32x4 xa[6], xb[6];
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            xa[j] = __builtin_nontemporal_load(ra + 16 * j);
            xb[j] = __builtin_nontemporal_load(rb + 16 * j);
        }
        float da = 0.f, db = 0.f;
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            da = __builtin_fmaf(xa[j].x, qv[j].x, da);
            da = __builtin_fmaf(xa[j].y, qv[j].y, da);
            da = __builtin_fmaf(xa[j].z, qv[j].z, da);
            da = __builtin_fmaf(xa[j].w, qv[j].w, da);
            db = __builtin_fmaf(xb[j].x, qv[j].x, db);
            db = __builtin_fmaf(xb[j].y, qv[j].y, db);
            db = __builtin_fmaf(xb[j].z, qv[j].z, db);
            db = __builtin_fmaf(xb[j].w, qv[j].w, db);
        }
#pragma unroll
        for (int o = 8; o >= 1; o >>= 1) {
            da += __shfl_xor(da, o);
            db += __shfl_xor(db, o);
        }
        if (part == 0) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const uint32_t e = h ? eb : ea;
                const uint2 v = h ? vb : va;
                const float ub = fminf(__builtin_bit_cast(float, v.x), (h ? db : da) + 3e-5f);
                if (e < n && ub > t) {
                    const uint32_t pos = atomicAdd(&cnt[(size_t)b * BATCH_CAND_SEGS + seg], 1u);
                    if (pos < seg_small) cand[((size_t)b * BATCH_CAND_SEGS + seg) * seg_small + pos] = uint2{__builtin_bit_cast(uint32_t, ub), v.y};
                }
            }
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        if (c0 > seg_cap_big) tau[b] = POS_INF;  // (the query is flagged by the tail: see f6_refine_kernel)
        cnt_big[(size_t)b * BATCH_CAND_SEGS + seg] = 0u;
    }
}

// tau <- max(tau, tau6): one threshold for both filters (a row dropped by either scores <= it)
__global__ void f6_merge_tau_kernel(float* __restrict__ tau, float* __restrict__ tau6, int n, int have_tau8) {
    const int i = threadIdx.x;
    if (i < n) {
        const float t = have_tau8 ? fmaxf(tau[i], tau6[i]) : tau6[i];
        tau[i] = t;
        tau6[i] = t;
    }
}

// The FP6 plan: a dense sample of 8192 strided rows -> a preliminary threshold -> an appended sample of n / 64 rows -> tau6 at
// rank `target` of the whole index.
struct F6Plan {
    uint32_t n_tiles, s1_tiles, s1_stride, m1, s2_tiles, s2_stride, m2;
};
static F6Plan plan_f6(uint32_t n_rows, uint32_t target) {
    F6Plan pl{};
    pl.n_tiles = (n_rows + 15u) / 16u;
    pl.s1_tiles = (uint32_t)BATCH_CAP / 16u;
    if (pl.s1_tiles > pl.n_tiles) pl.s1_tiles = pl.n_tiles;
    pl.s1_stride = pl.n_tiles / pl.s1_tiles;
    uint32_t t2 = pl.n_tiles / 64u;
    if (t2 > 1500000u / 16u) t2 = 1500000u / 16u;
    if (t2 < 1024u) t2 = 1024u;
    if (t2 > pl.n_tiles) t2 = pl.n_tiles;
    pl.s2_tiles = t2;
    pl.s2_stride = pl.n_tiles / t2;
    const double n2 = (double)t2 * 16.0;
    double m2 = (double)target * n2 / (double)n_rows;
    if (m2 < 16.0) m2 = 16.0;
    if (m2 > 2048.0) m2 = 2048.0;
    pl.m2 = (uint32_t)(m2 + 0.5);
    // the appended sample should hold ~4 m2 entries: its preliminary threshold is read from the sample-1 rank that deep
    double m1 = 4.0 * m2 * (double)(pl.s1_tiles * 16u) / n2;
    if (m1 < 4.0) m1 = 4.0;
    if (m1 > 256.0) m1 = 256.0;
    pl.m1 = (uint32_t)(m1 + 0.999);
    return pl;
}

// Batched search with the FP6 first filter (n_rows above ~50 M: below that the re-scoring of its survivors costs more than its pass saves).  ws: the int8 path's
// workspace (its query images refine the survivors; its sampled threshold tau8 and its candidate buffer feed the common tail);
// f6: the FP6 path's own buffers.
void launch_scan_batched_f6(const void* d_x, int dtype, const void* d_i8, const void* d_i8meta, const void* d_f6, const void* d_f6meta,
                            const uint64_t* d_ids, uint32_t n_rows, const float* d_q, int B, uint32_t k, const BatchWorkspace& ws,
                            const F6Workspace& f6, int grid, uint64_t* d_labels, float* d_dist, uint32_t* d_found, uint32_t* d_flags,
                            int force_fallback, hipStream_t stream, hipEvent_t ev0, hipEvent_t ev1) {
    // 1. survivors re-scored on the int8 shadow: the int8 path's query images and thresholds (tau8 at rank ~1024) — its sampling
    //    passes, not its full pass (0.41 ms per 100 M rows).  Re-scored on the f32 rows themselves, nothing of the int8 path is needed.
    const bool rows_refine = dtype == ROW_F32 && f6.refine_rows;
    if (!rows_refine) launch_i8_sample_thresholds(d_i8, d_i8meta, n_rows, d_q, B, k, ws, grid, stream);
    // 2. the FP6 thresholds
    const uint32_t target = (uint32_t)f6.target * (k > 32 ? 2u : 1u);
    const F6Plan pl = plan_f6(n_rows, target);
    const uint32_t* xs = reinterpret_cast<const uint32_t*>(d_f6);
    const float2* mt = reinterpret_cast<const float2*>(d_f6meta);
    const uint32_t* qf6 = reinterpret_cast<const uint32_t*>(f6.qf6);
    const float2* qm6 = reinterpret_cast<const float2*>(f6.qmeta);
    launch_prep_queries_f6(d_q, B, f6.qf6, f6.qmeta, stream);
    BatchWorkspace w6 = ws;  // (the int8 sampling is done with ws.cand / ws.cnt: free again, counters at zero)
    w6.tau = f6.tau6;
    const uint32_t seg_small = (uint32_t)BATCH_CAP / (uint32_t)BATCH_CAND_SEGS;
    {
        const uint32_t blocks = pl.s1_tiles < (uint32_t)grid ? pl.s1_tiles : (uint32_t)grid;
        hipLaunchKernelGGL(scan_f6_pass_kernel<true>, dim3(blocks), dim3(256), 0, stream, xs, mt, n_rows, pl.s1_stride, pl.s1_tiles, qf6,
                           qm6, B, f6.tau6, ws.cnt, reinterpret_cast<uint2*>(ws.cand), seg_small, reinterpret_cast<float*>(ws.cand), 0u, 0);
        launch_tau_select(true, B, w6, pl.s1_tiles * 16u, pl.m1, stream);
    }
    {
        // (two workgroups per CU: the strided sample is bound by memory latency, 0.36 -> 0.2 ms per 100 M rows)
        const uint32_t blocks = pl.s2_tiles < 2u * (uint32_t)grid ? pl.s2_tiles : 2u * (uint32_t)grid;
        hipLaunchKernelGGL(scan_f6_pass_kernel<false>, dim3(blocks), dim3(256), 0, stream, xs, mt, n_rows, pl.s2_stride, pl.s2_tiles, qf6,
                           qm6, B, f6.tau6, ws.cnt, reinterpret_cast<uint2*>(ws.cand), seg_small, reinterpret_cast<float*>(ws.cand), 0u, 0);
        launch_tau_select(false, B, w6, 0u, pl.m2, stream);
    }
    hipLaunchKernelGGL(f6_merge_tau_kernel, dim3(1), dim3(256), 0, stream, ws.tau, f6.tau6, B, rows_refine ? 0 : 1);
    // 3. the pass
    if (ev0) (void)hipEventRecord(ev0, stream);
    if (f6.stagger < 0) {  // the LDS-staged pass (the default)
        static OncePerDevice attr_once;
        once_per_device(attr_once, [] {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(scan_f6_pass_lds_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                      (int)sizeof(F6PassLds));
#ifdef DAWN_EXPERIMENTS  // (make EXPERIMENTS=1: the timing forms are not in the release library)
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(scan_f6_pass_lds_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                      (int)sizeof(F6PassLds));
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(scan_f6_pass_lds_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                      (int)sizeof(F6PassLds));
#endif
        });
        const uint32_t n_groups = (pl.n_tiles + (uint32_t)F6L_GROUP - 1u) / (uint32_t)F6L_GROUP;
        const uint32_t blocks = n_groups < (uint32_t)grid ? n_groups : (uint32_t)grid;
#ifdef DAWN_EXPERIMENTS
        auto kern = f6.stagger == -2 ? scan_f6_pass_lds_kernel<1> : f6.stagger == -4 ? scan_f6_pass_lds_kernel<4> : scan_f6_pass_lds_kernel<0>;
#else
        auto kern = scan_f6_pass_lds_kernel<0>;  // (f6_stagger -2 / -4: the timing forms of an EXPERIMENTS build; here the pass itself)
#endif
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(512), sizeof(F6PassLds), stream, xs, mt, n_rows, pl.n_tiles, qf6, qm6, B, f6.tau6,
                           f6.cnt_big, reinterpret_cast<uint2*>(f6.cand_big), f6.seg_cap_big);
    } else {
        const uint32_t blocks = pl.n_tiles < (uint32_t)grid ? pl.n_tiles : (uint32_t)grid;
        hipLaunchKernelGGL(scan_f6_pass_kernel<false>, dim3(blocks), dim3(256), 0, stream, xs, mt, n_rows, 1u, pl.n_tiles, qf6, qm6, B,
                           f6.tau6, f6.cnt_big, reinterpret_cast<uint2*>(f6.cand_big), f6.seg_cap_big, nullptr, (uint32_t)f6.stagger, 1);
    }
    if (ev1) (void)hipEventRecord(ev1, stream);
    // 4. survivors -> int8 bound -> the ordinary candidate buffers
    const signed char* qi8 = reinterpret_cast<const signed char*>(ws.qh);
    const float2* qm8 = reinterpret_cast<const float2*>(qi8 + (size_t)BATCH_QT * EM);
    if (rows_refine)
        hipLaunchKernelGGL(f6_refine_rows_kernel, dim3(B, BATCH_CAND_SEGS), dim3(256), 0, stream,
                           reinterpret_cast<const uint2*>(f6.cand_big), f6.cnt_big, f6.seg_cap_big, reinterpret_cast<const f32x4*>(d_x),
                           d_q, ws.tau, ws.cnt, reinterpret_cast<uint2*>(ws.cand));
    else
        hipLaunchKernelGGL(f6_refine_kernel, dim3(B, BATCH_CAND_SEGS), dim3(256), 0, stream, reinterpret_cast<const uint2*>(f6.cand_big),
                           f6.cnt_big, f6.seg_cap_big, reinterpret_cast<const unsigned char*>(d_i8),
                           reinterpret_cast<const float2*>(d_i8meta), qi8, qm8, ws.tau, ws.cnt, reinterpret_cast<uint2*>(ws.cand));
    // 5. the common tail
    launch_select_rescore_eps(false, d_x, dtype, d_ids, n_rows, d_q, B, k, ws, d_labels, d_dist, d_found, d_flags, force_fallback,
                              FILTER_EPS_I8, stream);
}

}  // namespace dawn
