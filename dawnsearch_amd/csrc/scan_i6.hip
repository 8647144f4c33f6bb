// scan_i6.hip — 6-bit FILTER shadow of the index rows (ROW_I6S) and the single-query stream over it: 288 B per row instead
// of the int8 shadow's 384 B.  The single-query search is a pure HBM stream (scan_i8.hip: 0.89 of the 8 TB/s spec, 0.99 of
// what a bare read reaches on this chip), so the only thing left that makes it faster is fewer bytes per row.
//
// Same construction as the int8 shadow (header of scan_i8.hip; every bound there carries over with 31 levels for 127):
//   * rows are rotated by R (rotate384.hpp) and quantised per 32-row sub-tile: s = max|x'| / 31, X = rint(x' / s) in
//     [-31, 31]; the sub-tile stores {1 / s, E}, E >= 1.0101 * max_r ||x'_r - s X_r||_2 measured at conversion time;
//   * a value is stored as the 6-bit code X + 32 in [1, 63].  A fragment is the A operand of one v_mfma_i32_32x32x32_i8 —
//     lane (h, r): the 16 values k = 32 f + 16 h .. +15 of row r — packed into THREE dwords per lane (768 B per fragment,
//     one global_load_dwordx3 per lane; 12 fragments = 9 KiB per sub-tile):  byte b of dword j < 3 holds code(4 j + b) in
//     its low six bits and two bits of code(12 + b) on top: D0 bits 0-1, D1 bits 2-3, D2 bits 4-5 of it.  Unpacking is
//     nine VALU instructions per fragment (three v_and, three v_lshrrev, one v_and, two v_and_or) in the shadow of the
//     previous fragment's MFMA (64 clocks of matrix pipe);
//   * codes are unsigned, so the accumulators start at -32 * sum_k Q_k (a constant per query image) instead of zero:
//     sum_k (code_k - 32) Q_k is exact in the integers, and everything from C = 254 acc_H + acc_L on is the int8 stream's;
//   * ub = fma(float(C), s * s_q / 254, E + K2) >= x.q as before, with K2 >= |(s X).dq| taken from the sub-tile's own MEASURED
//     error: ||s X||_2 <= ||x'||_2 + ||dx||_2 <= 1.015 + E (||x'||_2 < 1.01 by the gate, x 1.004 for a bf16 index's rounding;
//     ||dx||_2 <= E / 1.0101 by construction of E), ||dq||_2 <= sqrt(384) * 2.0e-3 s_q:  K2 = (1.015 + E) * 19.6 * 2.0e-3 * s_q.
//     (Round 3 used the constant 1.35 = 1.01 + sqrt(384) * s / 2 at s <= 1.01 / 31: right for 6 bits, 0.3 short for 5 — a
//     sub-tile that holds a one-hot row next to rows of components (n + 1/2) s reaches ||s X||_2 = 1.4 - 1.6; the measured
//     form holds whatever the rows and the chosen scale are: tests/test_scan_i6_gpu.py::test_k2_covers_the_worst_sub_tile.)
// E is four times the int8 shadow's (~0.037 on unit vectors; eight times at 5 bits), which a 64-row shortlist cannot absorb (the
// gap between the 10th and the 64th best score of 100 M rows is 0.017; tools/coarse_shadow_probe.py).  The stream therefore
// does not hand a 64-row shortlist to a one-workgroup tail.  Every WAVE keeps the best 24-64 rows of its share by the packed
// bound and re-scores them tightly (f32 index: on the f32 rows, a wave per row; bf16 index: on the int8 shadow), every
// WORKGROUP merges its waves' lists by that score and rescores its own 64 best rows exactly (reference order,
// src/search/vector.rs:128-134) in its epilogue — a shortlist up to 131 072 rows deep by the packed bound, 16 384 by the tight
// score —, and merge_exact_kernel merges the exact lists and checks ONE certificate: rows in no list score <= T = the largest
// of the workgroups' bounds on what they dropped, i.e. lie at distance >= fl(1 - up(T + eps)); if that exceeds the k-th exact
// distance strictly the result is exact, otherwise the query takes the exact pass (scan_exact_kernel) like any failed
// certificate.
#include <cmath>
#include <type_traits>

#include "kernels.hpp"
#include "rotate384.hpp"
#include "packed_shadow.hpp"
#include "wave_topk.hpp"

// FIVE bits (240 B per row, BITS = 5 below — the default; the 6-bit form stays selectable): the same with 15 levels, codes X + 16.
//   * a sub-tile is 7680 B in consumption order: [H0 | N0 N1 N2 | H1 | N3 N4 N5]; N_p (64 lanes x 16 B, one global_load_dwordx4)
//     holds the low nibbles of fragment PAIR p (fragments 2p, 2p + 1): dwords {0, 1} = fragment 2p, {2, 3} = fragment 2p + 1, a
//     dword = two unpacked dwords interleaved (low nibble of byte b: value b of A_{2i}, high nibble: value b of A_{2i+1});
//     H_g (64 lanes x 12 B, one global_load_dwordx3) holds the fifth bits of pairs 3g .. 3g + 2, one dword per pair: byte b,
//     bit 4 + j = bit 4 of value b of A_j of fragment 2p, bit j = the same of fragment 2p + 1 — so that
//     A_j = lo_j | ((H >> j) & 0x10101010) and A'_j = lo'_j | ((H << (4 - j)) & 0x10101010): 13-14 VALU per fragment;
//   * E doubles again (~0.06 on unit vectors with the clipped scale below): the certificate needs T (the largest 64th bound of
//     any list) a further 0.03 below the k-th score, i.e. lists twice as deep: a workgroup of eight waves keeps TWO lists (waves
//     0-3 and 4-7) and rescores both, 512 lists = 32 768 rows in all.
// Both forms choose the scale of a sub-tile by MEASUREMENT: s = c * max|x'| / levels for c in {1, .92, .84, .76, .68, .60}, the c
// with the smallest resulting E wins — clipping the few largest components (they saturate; their error is part of the measured
// E) buys a finer step for all the others: E -14 % at 6 bits, -23 % at 5 (the optimum of a uniform quantiser on near-Gaussian
// data lies at 3.0-3.3 sigma, the sub-tile maximum at ~4 sigma).
#ifdef DAWN_EXPERIMENTS
// timestamps (100-MHz counter) of every wave of scan_filter_i6s_kernel: [0] entry, [1] stream starts, [2] stream done,
// [3] refined, [4] workgroup done (dawn_debug_read_ts_i6; tools/stream_i5_ts.py)
static __device__ unsigned long long dawn_ts_i6[2048 * 5];
#define DAWN_TS6(i)                                                                                      \
    do {                                                                                                 \
        if ((threadIdx.x & 63) == 0 && blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6) < 2048)       \
            dawn_ts_i6[(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * 5 + (i)] = __builtin_amdgcn_s_memrealtime(); \
    } while (0)
#else
#define DAWN_TS6(i)
#endif

namespace dawn {

// ------------------------------------------------------------------------------------------------
// conversion: rows -> 6-bit sub-tiles + {1 / s, E} per sub-tile (rows_to_i8s_kernel with the packing above)
// ------------------------------------------------------------------------------------------------
template <int RT, int BITS>
__global__ __launch_bounds__(256) void rows_to_i6s_kernel(const void* __restrict__ xv, uint32_t* __restrict__ out,
                                                           float2* __restrict__ meta, uint32_t first_sub, uint32_t n_valid) {
    typedef PackedShadow<BITS> PS;
    constexpr int NC = 6;  // candidate scales
    __shared__ float sh[4];
    __shared__ float sh_e[4][NC];
    const uint32_t sub = first_sub + blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t r = tid >> 3, part = tid & 7;
    const uint32_t row = sub * 32u + r;
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
    float amax = 0.f;
#pragma unroll
    for (int j = 0; j < 12; ++j)
        amax = fmaxf(amax, fmaxf(fmaxf(fabsf(v[j].x), fabsf(v[j].y)), fmaxf(fabsf(v[j].z), fabsf(v[j].w))));
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) amax = fmaxf(amax, __shfl_xor(amax, o));
    if (lane == 0) sh[wave] = amax;
    __syncthreads();
    amax = fmaxf(fmaxf(sh[0], sh[1]), fmaxf(sh[2], sh[3]));
    // the scale: the candidate with the smallest measured E = max over the rows of ||x' - s X||_2 (clipped components included)
    const float s0 = fmaxf(amax, 1e-20f) / PS::LEVELS;
    float e2c[NC];
#pragma unroll
    for (int ci = 0; ci < NC; ++ci) {
        const float s = s0 * (1.0f - 0.08f * ci);
        float e2 = 0.f;
#pragma unroll
        for (int j = 0; j < 12; ++j) {
            const float vv[4] = {v[j].x, v[j].y, v[j].z, v[j].w};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float t = rintf(vv[i] / s);
                t = fminf(fmaxf(t, -PS::LEVELS), PS::LEVELS);
                const float dx = vv[i] - s * t;
                e2 = __builtin_fmaf(dx, dx, e2);
            }
        }
        e2 += __shfl_xor(e2, 1);  // row sum over its 8 threads (fixed order), then the maximum over the wave's 8 rows
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
    for (int ci = 0; ci < NC; ++ci) {  // (every thread takes the same decision)
        const float m = fmaxf(fmaxf(sh_e[0][ci], sh_e[1][ci]), fmaxf(sh_e[2][ci], sh_e[3][ci]));
        if (ci == 0 || m < best_e2) {
            best_e2 = m;
            s = s0 * (1.0f - 0.08f * ci);
        }
    }
    uint32_t* o = out + (size_t)sub * PS::SUB_DW;
    const uint32_t h = (part >> 2) & 1u, dj = part & 3u;
    const uint32_t ln = h * 32u + r;  // the lane of the stream's wave that owns these 16 values
    [[maybe_unused]] uint32_t hb_prev = 0;
#pragma unroll
    for (int j = 0; j < 12; ++j) {  // float4 chunk part + 8 j = dword dj of lane (h, r) of fragment j
        uint32_t w = 0;
        const float vv[4] = {v[j].x, v[j].y, v[j].z, v[j].w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float t = rintf(vv[i] / s);
            t = fminf(fmaxf(t, -PS::LEVELS), PS::LEVELS);
            w |= (uint32_t)((int)t + PS::OFFSET) << (8 * i);
        }
        const int base = lane & ~3;
        if constexpr (BITS == 6) {
            // the fourth dword of the lane (values 12..15) is spread over the top two bits of the other three
            const uint32_t w3 = __shfl(w, base + 3);
            if (dj < 3u) o[j * I6_FRAG_DW + ln * 3u + dj] = w | (((w3 >> (2u * dj)) & 0x03030303u) << 6);
        } else {
            const uint32_t w0 = __shfl(w, base), w1 = __shfl(w, base + 1), w2 = __shfl(w, base + 2), w3 = __shfl(w, base + 3);
            const uint32_t nd0 = (w0 & 0x0F0F0F0Fu) | ((w1 & 0x0F0F0F0Fu) << 4);
            const uint32_t nd1 = (w2 & 0x0F0F0F0Fu) | ((w3 & 0x0F0F0F0Fu) << 4);
            const uint32_t hb = ((w0 >> 4) & 0x01010101u) | (((w1 >> 4) & 0x01010101u) << 1) | (((w2 >> 4) & 0x01010101u) << 2) |
                                (((w3 >> 4) & 0x01010101u) << 3);  // bit d of byte b: the fifth bit of value b of A_d
            const uint32_t pr = (uint32_t)j >> 1, hf = pr / 3u, m = pr % 3u;
            uint32_t* half = o + hf * I5_HALF_DW;
            if (dj == 0u) half[192u + m * 256u + ln * 4u + 2u * (j & 1)] = nd0;
            if (dj == 1u) half[192u + m * 256u + ln * 4u + 2u * (j & 1) + 1u] = nd1;
            if ((j & 1) && dj == 2u) half[ln * 3u + m] = (hb_prev << 4) | hb;
            hb_prev = hb;
        }
    }
    if (tid == 0) {
        // 1.0101: ||q||_2 < 1.01 (gate); 1.001 + 1e-9: the f32 evaluation of dx, the sum and the square root
        meta[sub] = float2{1.0f / s, sqrtf(best_e2) * 1.0101f * 1.001f + 1e-9f};
    }
}

void launch_rows_to_i6s(const void* d_rows, int rt, int bits, void* d_shadow, void* d_meta, size_t first_row, size_t n_valid,
                        hipStream_t stream) {
    const uint32_t first_sub = (uint32_t)(first_row / 32);  // the sub-tile holding first_row is re-quantised whole
    const uint32_t end_sub = (uint32_t)((n_valid + 31) / 32);
    if (end_sub <= first_sub) return;
    uint32_t* o = reinterpret_cast<uint32_t*>(d_shadow);
    float2* mt = reinterpret_cast<float2*>(d_meta);
#define DAWN_I6_BUILD(RT_, BITS_)                                                                                          \
    hipLaunchKernelGGL((rows_to_i6s_kernel<RT_, BITS_>), dim3(end_sub - first_sub), dim3(256), 0, stream, d_rows, o, mt, first_sub, \
                       (uint32_t)n_valid)
    if (rt == ROW_BF16) {
        if (bits == 6) DAWN_I6_BUILD(1, 6);
        else DAWN_I6_BUILD(1, 5);
    } else {
        if (bits == 6) DAWN_I6_BUILD(0, 6);
        else DAWN_I6_BUILD(0, 5);
    }
#undef DAWN_I6_BUILD
}

// ------------------------------------------------------------------------------------------------
// the stream: scan_filter_i8s_pipe_kernel (software-pipelined test, two accumulator sets) for ONE query on 6-bit
// fragments, + the exact rescore of the workgroup's shortlist as its epilogue
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float dot4_f(const f32x4& a, const f32x4& b, float acc) {
    acc = __builtin_fmaf(a.x, b.x, acc);
    acc = __builtin_fmaf(a.y, b.y, acc);
    acc = __builtin_fmaf(a.z, b.z, acc);
    acc = __builtin_fmaf(a.w, b.w, acc);
    return acc;
}


// block_merge (wave_topk.hpp) for any number of waves
__device__ __forceinline__ void block_merge_any(float& s, uint32_t& p, float (*sh_s)[LIST], uint32_t (*sh_p)[LIST], int wave,
                                                int lane, int nwaves) {
    int top = 1;
    while (top < nwaves) top <<= 1;
    for (int stride = top >> 1; stride >= 1; stride >>= 1) {
        if (wave >= stride && wave < 2 * stride && wave < nwaves) {
            sh_s[wave][lane] = s;
            sh_p[wave][lane] = p;
        }
        __syncthreads();
        if (wave < stride && wave + stride < nwaves) {
            const float os = sh_s[wave + stride][63 - lane];
            const uint32_t op = sh_p[wave + stride][63 - lane];
            merge64(s, p, os, op, lane);
        }
        __syncthreads();
    }
}

// RT: row type of the index (0 f32, 1 bf16) for the exact rescore; BITS: 6 or 5; PD: loads in flight per wave — 6 bits: 12 / 6 /
// 4 / 3 / 2 fragments of 768 B; 5 bits: 8 (a whole sub-tile ahead, 7.5 KiB) or 4 (half a sub-tile: one H and three N loads)
template <int RT, int BITS, int PD>
__global__ __launch_bounds__(512) void scan_filter_i6s_kernel(const uint32_t* __restrict__ x, const float2* __restrict__ meta,
                                                               uint32_t n_rows, const float* __restrict__ q,
                                                               const void* __restrict__ rows, const unsigned char* __restrict__ x8,
                                                               const float2* __restrict__ meta8, float* __restrict__ out_s,
                                                               uint32_t* __restrict__ out_p, float* __restrict__ out_es,
                                                               uint32_t* __restrict__ out_ep, float* __restrict__ out_t,
                                                               int n_refine_arg, uint32_t* __restrict__ pool, int exact_epilogue,
                                                               uint32_t chunk_arg) {
    static_assert(BITS == 6 ? 12 % PD == 0 : (PD == 8 || PD == 4), "ring depth");
    typedef PackedShadow<BITS> PS;
    __shared__ float sh_s[8][LIST];
    __shared__ uint32_t sh_p[8][LIST];
    __shared__ uint32_t sh_rows[LIST];
    __shared__ float sh_strip[8][32];  // a sub-tile's 32 upper bounds, one strip per wave (slow_path)
    __shared__ float sh_tw[8];         // the waves' bounds on their unlisted rows
    extern __shared__ __attribute__((aligned(16))) unsigned char rescore_stage[];  // RescoreStage<RT>::BYTES
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nwaves = blockDim.x >> 6;
    const uint32_t n_sub = (n_rows + 31u) >> 5;
    const uint32_t c = lane & 31, h = lane >> 5;
    uint32_t t = blockIdx.x * nwaves + wave;
    const uint32_t t_stride = gridDim.x * nwaves, t_end = n_sub;
    const int n_refine = n_refine_arg < 0 ? LIST : n_refine_arg;
    DAWN_TS6(0);
    // Work assignment (indexes of ~3.7 M rows and more; smaller ones: static throughout).  Static and interleaved — wave w takes
    // sub-tiles w, w + W, w + 2W, ... — for the first 7/8 of the index;
    // the last eighth is handed out in CHUNKS of 16 sub-tiles on demand.  The waves' shares of a static assignment are equal,
    // their speeds are not: the first wave of a 100 M-row launch is done 260-340 us before the last (tools/stream_i5_ts.py,
    // profiles/r03/stream_i5_wave_timestamps_100M_static.log) — 4 % of the kernel during which the memory system runs half empty.
    //   * a chunk is STRIDED: chunk j = sub-tiles D0 + j + i C, i < 16 (C = number of chunks): never two consecutive sub-tiles
    //     in one wave — a run of similar rows in one wave is what the list length is sized against;
    //   * a chunk is fetched with a SCALAR atomic (s_atomic_add, returns through lgkmcnt): a returning vector atomic in this
    //     loop makes hipcc drain the load ring — with one sub-tile per atomic that cost 2 x the kernel (7.4 ms,
    //     stream_i5_wave_timestamps_100M_atomic_pool_experiment.log), and same-address atomics retire at only ~13 M/s on
    //     this chip (512 of them took 40 us): hence chunks, and 32 counters — counter (b / 8) % 32 serves the chunks of that
    //     residue: its eight workgroups run on the eight XCDs (workgroup b: XCD b % 8), whose memory speeds differ too;
    //   * a rotated static interleave (round i of wave w = sub-tile i W + ((w + 37 i) mod W)) changes nothing: the speed
    //     differences belong to the waves' places on the chip, not to their addresses (..._rotated_interleave_experiment.log).
    constexpr uint32_t NONE = 0xFFFFFFFFu;
    constexpr uint32_t POOLS = 32;
    const uint32_t CHUNK = chunk_arg & 0xFFFFu;  // sub-tiles per chunk of the dynamic tail (16; option "i6_dyn_chunk")
    const uint32_t SHARE16 = chunk_arg >> 16;     // sixteenths of the index handed out dynamically (2; option "i6_dyn_share")
    uint32_t static_left = NONE;  // further static sub-tiles of this wave (NONE: static to the end)
    uint32_t dyn0 = 0, n_chunks = 0, chunk_base = 0, chunk_i = CHUNK;
    // counters in use: one per group of eight workgroups, at most POOLS (a counter nobody reads would strand its chunks)
    const uint32_t n_pools = ((gridDim.x + 7u) >> 3) < POOLS ? ((gridDim.x + 7u) >> 3) : POOLS;
    const uint32_t pool_id = (blockIdx.x >> 3) % n_pools;
    uint32_t* my_pool = nullptr;
    // (from 56 rounds of the grid on, ~3.7 M rows: below that the static interleave alone is faster — 1.5 M rows 0.111 against 0.116 ms,
    // 2 M 0.129 / 0.131, a tie at 4 M: profiles/r05/dyn_small_probe.log; 16 rounds until round 5)
    if (pool != nullptr && n_sub / t_stride >= 56u) {
        const uint32_t rounds = n_sub / t_stride, i_static = rounds - rounds * SHARE16 / 16u;
        dyn0 = t_stride * i_static;
        n_chunks = (n_sub - dyn0 + CHUNK - 1u) / CHUNK;
        my_pool = pool + pool_id;
        static_left = i_static - 1u;
    }
    auto fetch_chunk = [&]() __attribute__((always_inline)) -> uint32_t {  // the next chunk of this wave's pool, or NONE
        uint32_t v = 1u;
        asm volatile("s_atomic_add %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "+s"(v) : "s"(my_pool));
        const uint32_t j = pool_id + n_pools * v;
        return j < n_chunks ? j : NONE;
    };
    // the first loads fly while the query images are made.  6 bits: a[] = ring of fragments (lane slot: 3 dwords);
    // 5 bits: hq[] = ring of H loads (lane slot: 3 dwords), nq[] = ring of N loads (lane slot: 4 dwords)
    constexpr int NA = BITS == 6 ? PD : 1, NH = BITS == 5 ? PD / 4 : 1, NN = BITS == 5 ? 3 * PD / 4 : 1;
    const uint32_t* p = x + (size_t)(t < t_end ? t : 0) * PS::SUB_DW;
    [[maybe_unused]] u32x3 a[NA];
    [[maybe_unused]] u32x3 hq[NH];
    [[maybe_unused]] u32x4 nq[NN];
    auto load_h = [&](const uint32_t* sub, int g) __attribute__((always_inline)) {
        return frag_load(sub + g * I5_HALF_DW + lane * 3);
    };
    auto load_n = [&](const uint32_t* sub, int pr) __attribute__((always_inline)) {  // pr = pair 0..5
        return __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(sub + (pr / 3) * I5_HALF_DW + 192 + (pr % 3) * 256 + lane * 4));
    };
    float2 mt = {0.f, 0.f};
    if (t < t_end) {
        if constexpr (BITS == 6) {
#pragma unroll
            for (int d = 0; d < PD; ++d) a[d] = frag_load(p + d * I6_FRAG_DW + lane * 3);
        } else {
#pragma unroll
            for (int g = 0; g < NH; ++g) {
                hq[g] = load_h(p, g);
#pragma unroll
                for (int m = 0; m < 3; ++m) nq[3 * g + m] = load_n(p, 3 * g + m);
            }
        }
        mt = meta[t];
    }

    // the query's two int8 images (column 0: H = rint(q' / s_q), column 8: L = rint(254 (q' / s_q - H))), built once per
    // workgroup by wave 0 — as in scan_filter_i8s_kernel — together with the sums the accumulators start from
    __shared__ __attribute__((aligned(16))) signed char sh_img[2][EM];
    __shared__ float sh_sq;
    __shared__ int sh_sum[2];
    if (wave == 0) {
        float v[6];
#pragma unroll
        for (int j = 0; j < 6; ++j) v[j] = q[lane + 64 * j];
        rotate384_wave(v, lane);
        float amax = 0.f;
#pragma unroll
        for (int j = 0; j < 6; ++j) amax = fmaxf(amax, fabsf(v[j]));
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) amax = fmaxf(amax, __shfl_xor(amax, o));
        const float sq = fmaxf(amax, 1e-20f) / 127.0f;
        int sumH = 0, sumL = 0;
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            const float tt = v[j] / sq;
            const float H = fminf(fmaxf(rintf(tt), -127.f), 127.f);
            const float L = fminf(fmaxf(rintf((tt - H) * 254.0f), -127.f), 127.f);
            sh_img[0][lane + 64 * j] = (signed char)(int)H;
            sh_img[1][lane + 64 * j] = (signed char)(int)L;
            sumH += (int)H;
            sumL += (int)L;
        }
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) {
            sumH += __shfl_xor(sumH, o);
            sumL += __shfl_xor(sumL, o);
        }
        if (lane == 0) {
            sh_sq = sq;
            sh_sum[0] = sumH;
            sh_sum[1] = sumL;
        }
    }
    __syncthreads();
    i32x4_t qf[12];
    const bool col_live = c == 0 || c == 8;
    {
        const i32x4_t* img = reinterpret_cast<const i32x4_t*>(&sh_img[c == 8 ? 1 : 0][0]);
#pragma unroll
        for (int f = 0; f < 12; ++f) qf[f] = col_live ? img[2 * f + h] : i32x4_t{0, 0, 0, 0};
    }
    const float sq = sh_sq;
    // ub = C g1 + E + K2(E) = C g1 + E (1 + k2u) + k2c
    const float sq254 = sq / 254.0f, rsq254 = 254.0f / sq, k2u = I6_K2U_PER_SQ * sq, k2c = I6_XNORM * k2u;
    const float emul = 1.0f + k2u, emul_thr = emul * 1.000001f;
    const int acc0 = col_live ? -PS::OFFSET * sh_sum[c == 8 ? 1 : 0] : 0;  // codes are value + OFFSET
    float ls = NEG_INF, tau = NEG_INF;
    uint32_t lp = NO_POS;
    const bool tested = c == 0;
    float tau_m = tested ? NEG_INF : __builtin_inff();

    DAWN_TS6(1);
    if (t < t_end) {
        i32x16_t accs[2];
        float2 pmt = mt;
        uint32_t prow = 0;
        int C[16];
        int thr = 0, mx = 0;

        // The wave's threshold: only its best n_refine rows are refined, and the bound on everything else is the best row it
        // does NOT keep — entry n_refine of the list.  Rows below that entry can neither enter the kept part nor raise the bound,
        // so they are not inserted at all (entries behind it go stale and are discarded with it): a list of 40 takes a third
        // fewer insertions than one of 64 on a 12.5 M-row index (profiles/r03/stream_i5_threshold_ab.log).
        const int tau_lane = n_refine < LIST ? n_refine : LIST - 1;
        auto list_tau = [&]() __attribute__((always_inline)) -> float {
            return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, ls), tau_lane));
        };
        auto slow_path = [&]() __attribute__((always_inline)) {
            // Many hits in one sub-tile — the first sub-tiles of every wave, while its list fills: the j-th still contributes
            // 64 / (j + 1) rows — are merged as ONE sorted batch (scan_filter_i8s_kernel: the 32 upper bounds, held by the two
            // lanes of column 0, go through a 128-B strip of LDS to one per lane, a bitonic sort and one merge64) instead of one
            // ballot-and-shift insertion each.  Same list afterwards (rows arrive in ascending order).
            int nh = 0;
#pragma unroll
            for (int e = 0; e < 16; ++e)
                nh += (C[e] > thr && prow + (uint32_t)((e & 3) + 8 * (e >> 2)) + 4u * h < n_rows) ? 1 : 0;
            const int total = __builtin_amdgcn_readlane(nh, 0) + __builtin_amdgcn_readlane(nh, 32);
            if (total > 6) {
                float* strip = &sh_strip[wave][0];
                if (c == 0) {  // lanes 0 (h = 0) and 32 (h = 1)
                    const float g1 = __builtin_amdgcn_rcpf(pmt.x) * sq254, g0 = __builtin_fmaf(pmt.y, emul, k2c);
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const uint32_t roff = (uint32_t)((e & 3) + 8 * (e >> 2)) + 4u * h;
                        const bool ok = C[e] > thr && prow + roff < n_rows;
                        strip[roff] = ok ? __builtin_fmaf((float)C[e], g1, g0) : NEG_INF;
                    }
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // (wave-private strip: LDS operations of a wave are in order)
                float d = POS_INF;
                uint32_t row = NO_POS;
                if (lane < 32) {
                    const float sc = strip[lane];
                    if (sc > tau) {
                        d = -sc;
                        row = prow + (uint32_t)lane;
                    }
                }
                asm volatile("" ::: "memory");
                sort64_asc(d, row, lane);
                const float os = -__shfl(d, 63 - lane);
                const uint32_t op = __shfl(row, 63 - lane);
                merge64(ls, lp, os, op, lane);
                tau = list_tau();
            } else {
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const uint32_t roff = (uint32_t)((e & 3) + 8 * (e >> 2));
                    unsigned long long m = __ballot(C[e] > thr && prow + roff + 4u * h < n_rows);
                    while (m) {
                        const int l = __builtin_ctzll(m);
                        m &= m - 1;
                        const float cf = (float)__builtin_amdgcn_readlane(C[e], l);
                        const uint32_t row = prow + roff + 4u * (uint32_t)(l >> 5);
                        const float sc = __builtin_fmaf(cf, __builtin_amdgcn_rcpf(pmt.x) * sq254, __builtin_fmaf(pmt.y, emul, k2c));
                        if (sc > tau) {
                            wave_insert(ls, lp, sc, row, lane);
                            tau = list_tau();
                        }
                    }
                }
            }
            if (tested) {
                const float tk = tau - k2c;
                tau_m = tk - fabsf(tk) * 1e-6f;
            }
        };
        auto test_slice = [&](int s, const i32x16_t& pacc) __attribute__((always_inline)) {
            if (s == 0) {
                const float u = __builtin_fmaf(-pmt.y, emul_thr, tau_m);
                float thr_f = __builtin_fmaf(u, pmt.x * rsq254, -2.0f);
                thr_f = fminf(fmaxf(thr_f, -2.0e9f), 2.0e9f);
                if (!tested) thr_f = 2.0e9f;
                thr = (int)floorf(thr_f);
            } else if (s <= 8) {
#pragma unroll
                for (int e = 2 * (s - 1); e < 2 * s; ++e) {
                    const int ae = pacc[e];
                    C[e] = __mul24(ae, 254) + __builtin_amdgcn_update_dpp(0, ae, 0x108, 0xf, 0xf, true);
                    mx = e == 0 ? C[0] : max(mx, C[e]);
                }
            }
        };

        bool more;
        auto round = [&](auto with_test, auto parity) __attribute__((always_inline)) {
            constexpr int P = decltype(parity)::value;
            constexpr bool TEST = decltype(with_test)::value;
            i32x16_t& acc = accs[P];
            uint32_t tn;
            if (static_left != 0u) {
                tn = t + t_stride;
                if (static_left != NONE) --static_left;
            } else {
                tn = chunk_i + 1u < CHUNK ? chunk_base + (chunk_i + 1u) * n_chunks : NONE;
                if (tn < t_end) {
                    ++chunk_i;
                } else {
                    const uint32_t j = fetch_chunk();
                    chunk_base = dyn0 + j;
                    chunk_i = 0u;
                    tn = j != NONE ? chunk_base : NONE;
                }
            }
            more = tn < t_end;
            // (the last sub-tile re-reads its own first bytes: no branch in the stream)
            const uint32_t* pn = more ? x + (size_t)tn * PS::SUB_DW : p;
            const float2 mtn = meta[more ? tn : t];
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[e] = acc0;
            if constexpr (BITS == 6) {
#pragma unroll
                for (int f = 0; f < 12; ++f) {
                    const u32x3 w = a[f % PD];
                    const uint32_t w0 = w.x, w1 = w.y, w2 = w.z;
                    i32x4_t av;
                    av[0] = (int)(w0 & 0x3F3F3F3Fu);
                    av[1] = (int)(w1 & 0x3F3F3F3Fu);
                    av[2] = (int)(w2 & 0x3F3F3F3Fu);
                    av[3] = (int)(((w0 >> 6) & 0x03030303u) | ((w1 >> 4) & 0x0C0C0C0Cu) | ((w2 >> 2) & 0x30303030u));
                    acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(av, qf[f], acc, 0, 0, 0);
                    // (the reload BEHIND the MFMA: in front of it — the slot is free once unpacked — rings of 3-4 fragments lose
                    // 1-2 %, rings of 12 gain nothing: profiles/r03/stream_i6_parts_off_100M.log, DBG = 8)
                    if (f + PD < 12) a[f % PD] = frag_load(p + (f + PD) * I6_FRAG_DW + lane * 3);
                    else a[f % PD] = frag_load(pn + (f + PD - 12) * I6_FRAG_DW + lane * 3);
                    if constexpr (TEST) test_slice(f, accs[1 - P]);
                    __builtin_amdgcn_sched_barrier(0);
                }
            } else {
#pragma unroll
                for (int pr = 0; pr < 6; ++pr) {  // fragment pairs
                    const int g = pr / 3, m = pr % 3;
                    const u32x4 nw = nq[pr % NN];
                    const u32x3 hw = hq[g % NH];
                    const uint32_t H = m == 0 ? hw.x : m == 1 ? hw.y : hw.z;
                    const uint32_t n0 = nw.x, n1 = nw.y, n2 = nw.z, n3 = nw.w;
                    i32x4_t av, bv;
                    av[0] = (int)((n0 & 0x0F0F0F0Fu) | (H & 0x10101010u));
                    av[1] = (int)(((n0 >> 4) & 0x0F0F0F0Fu) | ((H >> 1) & 0x10101010u));
                    av[2] = (int)((n1 & 0x0F0F0F0Fu) | ((H >> 2) & 0x10101010u));
                    av[3] = (int)(((n1 >> 4) & 0x0F0F0F0Fu) | ((H >> 3) & 0x10101010u));
                    acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(av, qf[2 * pr], acc, 0, 0, 0);
                    if constexpr (TEST) test_slice(2 * pr, accs[1 - P]);
                    __builtin_amdgcn_sched_barrier(0);
                    bv[0] = (int)((n2 & 0x0F0F0F0Fu) | ((H << 4) & 0x10101010u));
                    bv[1] = (int)(((n2 >> 4) & 0x0F0F0F0Fu) | ((H << 3) & 0x10101010u));
                    bv[2] = (int)((n3 & 0x0F0F0F0Fu) | ((H << 2) & 0x10101010u));
                    bv[3] = (int)(((n3 >> 4) & 0x0F0F0F0Fu) | ((H << 1) & 0x10101010u));
                    acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(bv, qf[2 * pr + 1], acc, 0, 0, 0);
                    // reloads: the same slot NN pairs / NH halves further down the stream
                    if (pr + NN < 6) nq[pr % NN] = load_n(p, pr + NN);
                    else nq[pr % NN] = load_n(pn, pr + NN - 6);
                    if (m == 2) {
                        if (g + NH < 2) hq[g % NH] = load_h(p, g + NH);
                        else hq[g % NH] = load_h(pn, g + NH - 2);
                    }
                    if constexpr (TEST) test_slice(2 * pr + 1, accs[1 - P]);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            if constexpr (TEST)
                if (__any(mx > thr)) slow_path();
            pmt = mt;
            prow = t * 32u;
            t = tn;
            p = pn;
            mt = mtn;
        };
        using P0 = std::integral_constant<int, 0>;
        using P1 = std::integral_constant<int, 1>;
        round(std::false_type{}, P0{});
        int last = 0;
        while (more) {
            round(std::true_type{}, P1{});
            last = 1;
            if (!more) break;
            round(std::true_type{}, P0{});
            last = 0;
        }
        if (last) {
#pragma unroll
            for (int s = 0; s < 12; ++s) test_slice(s, accs[1]);
        } else {
#pragma unroll
            for (int s = 0; s < 12; ++s) test_slice(s, accs[0]);
        }
        if (__any(mx > thr)) slow_path();
    }

    DAWN_TS6(2);
    // ---- epilogue 1: REFINEMENT.  The wave's list holds its best rows by the packed shadow's bound; every row of the wave
    // that is not listed is bounded by the first entry it drops (tw; -inf while the list is not full).  The listed rows get a
    // TIGHT score in its place and the list is re-sorted by it: what the workgroup then merges and rescores are its 64 best rows
    // by the tight score, chosen among nwaves x n_refine candidates of the coarse one.  f32 index: the fast f32 dot of the row
    // itself; bf16 index (rows in fragment order: no contiguous row to read): the int8 shadow's bound (scan_i8.hip: an eighth of
    // the slack) — a lane per row, its 24 16-B pieces of the int8 sub-tile against the query's two int8 images in LDS,
    // v_dot4_i32_i8.
    // Only the wave's best n_refine entries are kept (1 .. 64, chosen by the host from the index size and k: i6_refine_count):
    // a short list is a shallow one — its bound tw sits higher —, but every kept row costs a 3-KB gather, and the depth the
    // certificate needs grows with the index.
    const bool coarse_only = n_refine_arg < 0;  // test hook (option "i6_refine" = -1): the lists keep the packed shadow's own bounds
    float tw;
    if (n_refine < LIST) {
        const bool more_rows = __builtin_amdgcn_readlane((int)lp, n_refine) != (int)NO_POS;
        tw = more_rows ? __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, ls), n_refine)) : NEG_INF;
        if (lane >= n_refine) {
            ls = NEG_INF;
            lp = NO_POS;
        }
    } else {
        const bool full = __builtin_amdgcn_readlane((int)lp, 63) != (int)NO_POS;
        tw = full ? read_lane63(ls) : NEG_INF;
    }
    if (coarse_only) {
        // (nothing: descending by the coarse bound already)
    } else if constexpr (RT == 0) {
        // f32 index: the kept rows are scored on the f32 rows themselves, a wave per row — one coalesced 1.5-KB read per row
        // (the int8 sub-tile holds a row as 24 pieces 512 B apart: a 3-KB scatter in 64-B sectors), 8 rows in flight, 24 FMAs
        // per lane and a DPP sum per row.  The score errs like scan_filter_kernel's (f32 FMAs in another order than the
        // reference's sequential sum: FILTER_EPS_F32 = 2.6e-5 <= the eps of the certificate), i.e. it bounds the exact score
        // as tightly as anything short of the exact rescore.
        const f32x4* q4 = reinterpret_cast<const f32x4*>(q);
        const f32x4* x4 = reinterpret_cast<const f32x4*>(rows);
        const f32x4 qa = q4[lane];
        const f32x4 qb = lane < 32 ? q4[64 + lane] : f32x4{0.f, 0.f, 0.f, 0.f};
        float mine = NEG_INF;
        for (int i0 = 0; i0 < n_refine; i0 += 8) {
            f32x4 xa[8], xb[8];
            uint32_t rr[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                rr[j] = (uint32_t)__builtin_amdgcn_readlane((int)lp, i0 + j);  // (i0 + j < 64: n_refine is a multiple of 8)
                const f32x4* rp = x4 + (size_t)(rr[j] != NO_POS ? rr[j] : 0u) * ROW_F4;
                xa[j] = rp[lane];
                xb[j] = lane < 32 ? rp[64 + lane] : f32x4{0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float part = dot4_f(xa[j], qa, dot4_f(xb[j], qb, 0.f));
                const float tot = read_lane63(wave_sum_lane63(part));
                if (lane == i0 + j && rr[j] != NO_POS) mine = tot;
            }
        }
        float d = lp != NO_POS ? -mine : POS_INF;
        uint32_t pr = lp;
        sort64_asc(d, pr, lane);
        ls = -d;  // descending by the refined score, ties -> lower row; fillers (-inf, NO_POS) last
        lp = pr;
    } else {
        float ub8 = NEG_INF;
        if (lp != NO_POS) {
            const uint32_t t8 = lp >> 5, r8 = lp & 31u;
            const i32x4_t* xs = reinterpret_cast<const i32x4_t*>(x8 + (size_t)t8 * 12288u) + r8;
            const i32x4_t* ih = reinterpret_cast<const i32x4_t*>(&sh_img[0][0]);
            const i32x4_t* il = reinterpret_cast<const i32x4_t*>(&sh_img[1][0]);
            int ch = 0, cl = 0;
#pragma unroll 6
            for (int j = 0; j < 24; ++j) {  // piece j = fragment j / 2, half j % 2: k = 16 j .. 16 j + 15
                const i32x4_t xv = xs[(j >> 1) * 64 + (j & 1) * 32];
                const i32x4_t hv = ih[j], lv = il[j];
                ch = __builtin_amdgcn_sdot4(xv[0], hv[0], ch, false);
                ch = __builtin_amdgcn_sdot4(xv[1], hv[1], ch, false);
                ch = __builtin_amdgcn_sdot4(xv[2], hv[2], ch, false);
                ch = __builtin_amdgcn_sdot4(xv[3], hv[3], ch, false);
                cl = __builtin_amdgcn_sdot4(xv[0], lv[0], cl, false);
                cl = __builtin_amdgcn_sdot4(xv[1], lv[1], cl, false);
                cl = __builtin_amdgcn_sdot4(xv[2], lv[2], cl, false);
                cl = __builtin_amdgcn_sdot4(xv[3], lv[3], cl, false);
            }
            const float2 m8 = meta8[t8];  // {1 / s, E} of the int8 sub-tile
            const int c8 = ch * 254 + cl;
            ub8 = __builtin_fmaf((float)c8, __builtin_amdgcn_rcpf(m8.x) * sq254, m8.y + I8_K2_PER_SQ * sq);
        }
        float d = lp != NO_POS ? -ub8 : POS_INF;
        uint32_t pr = lp;
        sort64_asc(d, pr, lane);
        ls = -d;  // descending by the int8 bound, ties -> lower row; fillers (-inf, NO_POS) last
        lp = pr;
    }
    if (lane == 0) sh_tw[wave] = tw;  // (visible after block_merge's barriers)
    DAWN_TS6(3);

    // ---- epilogue 2: the workgroup's list — upper bounds (descending: lane 63 bounds every refined row that is not listed) —
    // and the same 64 rows rescored exactly (block_exact_dots of wave_topk.hpp for any block size) as (-distance descending, row)
    block_merge_any(ls, lp, sh_s, sh_p, wave, lane, nwaves);
    typedef RescoreStage<RT> S;
    float* sh_q = reinterpret_cast<float*>(rescore_stage + S::ROWS_BYTES);
    for (int i = threadIdx.x; i < EM; i += blockDim.x) sh_q[i] = q[i];
    const u32x4* xr = reinterpret_cast<const u32x4*>(rows);
    const size_t o = (size_t)blockIdx.x * LIST + lane;
    if (wave == 0) {
        out_s[o] = ls;
        out_p[o] = lp;
        sh_rows[lane] = lp;
        if (lane == 0) {
            // the bound on every row of this workgroup that is in no list: the coarse bound of the rows its waves dropped, the
            // tight one of the rows the merge dropped
            float tb = (__builtin_amdgcn_readlane((int)lp, 63) != (int)NO_POS) ? read_lane63(ls) : NEG_INF;
            for (int w = 0; w < nwaves; ++w) tb = fmaxf(tb, sh_tw[w]);
            out_t[blockIdx.x] = tb;
        }
    }
    // (central tail — option "i6_central_tail": the lists and the bound go to merge_rescore_kernel, which rescores the 64 best
    // rows of the whole index exactly instead of every workgroup its own 64)
    if (!exact_epilogue) return;
    __syncthreads();
    for (int i = threadIdx.x; i < LIST * S::CH; i += blockDim.x) {
        const int r = i / S::CH, chn = i % S::CH;
        const uint32_t row = sh_rows[r];
        if (row != NO_POS)
            *reinterpret_cast<u32x4*>(rescore_stage + r * S::STRIDE + chn * 16) =
                RT == 1 ? xr[frag_chunk(row, chn)] : xr[(size_t)row * S::CH + chn];
    }
    __syncthreads();
    if (RT == 0) {  // products in place by everyone, the sequential adds by wave 0 (block_exact_dots)
        for (int i = threadIdx.x; i < LIST * S::CH; i += blockDim.x) {
            const int r = i / S::CH, chn = i % S::CH;
            f32x4* px = reinterpret_cast<f32x4*>(rescore_stage + r * S::STRIDE + chn * 16);
            const f32x4 xv = *px, qq = reinterpret_cast<const f32x4*>(sh_q)[chn];
            *px = f32x4{__fmul_rn(qq.x, xv.x), __fmul_rn(qq.y, xv.y), __fmul_rn(qq.z, xv.z), __fmul_rn(qq.w, xv.w)};
        }
        __syncthreads();
    }
    if (wave == 0) {
        float d = POS_INF;
        uint32_t pr = lp;
        if (lp != NO_POS) {
            float dot;
            if (RT == 1) dot = exact_dot_seq_bf16<8>(sh_q, reinterpret_cast<const u32x4*>(rescore_stage + lane * S::STRIDE));
            else dot = sum_seq(reinterpret_cast<const f32x4*>(rescore_stage + lane * S::STRIDE));
            d = __fsub_rn(1.0f, dot);  // vector.rs:133  1.0 - result
            if (!(d == d)) {
                d = POS_INF;
                pr = NO_POS;
            }
        }
        sort64_asc(d, pr, lane);
        out_es[o] = -d;  // (-distance descending, row ascending among equals)
        out_ep[o] = pr;
    }
    DAWN_TS6(4);
}

// ------------------------------------------------------------------------------------------------
// merge of the exact per-workgroup lists + the certificate
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void merge_exact_kernel(const uint64_t* __restrict__ ids, uint32_t n_rows,
                                                            const float* __restrict__ tb, const float* __restrict__ ex_s,
                                                            const uint32_t* __restrict__ ex_p, int n_lists, uint32_t k,
                                                            uint64_t* __restrict__ out_labels, float* __restrict__ out_dist,
                                                            uint32_t* __restrict__ out_found, uint32_t* __restrict__ out_flags,
                                                            int force_fallback, float eps, uint32_t* __restrict__ pool,
                                                            uint32_t* __restrict__ stats) {
    __shared__ float sh_s[16][LIST];
    __shared__ uint32_t sh_p[16][LIST];
    __shared__ float sh_t[16];
    if (pool != nullptr && threadIdx.x < 32) pool[threadIdx.x] = 0;  // the stream's chunk counters, for the next search
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nwaves = blockDim.x >> 6;
    float s = NEG_INF, t0 = NEG_INF;
    uint32_t p = NO_POS;
    constexpr int INF = 16;  // lists in flight per wave: the usual 256 lists are one round of loads
    for (int l0 = wave; l0 < n_lists; l0 += INF * nwaves) {
        float os[INF], tl[INF];
        uint32_t op[INF];
#pragma unroll
        for (int j = 0; j < INF; ++j) {
            const int l = l0 + j * nwaves;
            os[j] = l < n_lists ? ex_s[(size_t)l * LIST + 63 - lane] : NEG_INF;
            op[j] = l < n_lists ? ex_p[(size_t)l * LIST + 63 - lane] : NO_POS;
            tl[j] = l < n_lists ? tb[l] : NEG_INF;  // the workgroup's bound on its unlisted rows (-inf: nothing was left out)
        }
#pragma unroll
        for (int j = 0; j < INF; ++j) {
            if (l0 + j * nwaves >= n_lists) continue;  // wave-uniform
            t0 = fmaxf(t0, tl[j]);
            merge64(s, p, os[j], op[j], lane);
        }
    }
    if (lane == 0) sh_t[wave] = t0;  // (visible after block_merge's barriers)
    block_merge(s, p, sh_s, sh_p, wave, lane, nwaves);
    if (wave != 0) return;
    float T = NEG_INF;
    for (int w = 0; w < nwaves; ++w) T = fmaxf(T, sh_t[w]);
    const uint32_t found = n_rows < k ? n_rows : k;
    uint32_t flag = FLAG_OK;
    if (found > 0) {
        const uint32_t have = __popcll(__ballot(p != NO_POS));
        if (have < found) {
            flag = FLAG_FALLBACK;
        } else if (T > NEG_INF) {
            const float t = round_up_f32((double)T + (double)eps);
            const float d_bound = __fsub_rn(1.0f, t);
            const float dk = -__builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, s), (int)found - 1));
            if (!(d_bound > dk)) flag = FLAG_FALLBACK;
        }
    }
    // (the index's ladder feedback watches how often this certificate fails on its own: dawn_index.cpp)
    if (stats && lane == 0 && flag == FLAG_FALLBACK) atomicAdd(&stats[STAT_PACKED_FAIL], 1u);
    if (force_fallback && n_rows > 0) flag = FLAG_FALLBACK;
    if ((uint32_t)lane < found) {
        // (slots without a candidate — fewer rows found than asked for: the ladder takes over — read as "no threshold")
        out_labels[lane] = p != NO_POS ? ids[p] : 0ull;
        out_dist[lane] = p != NO_POS ? -s : POS_INF;
    }
    if (lane == 0) {
        out_found[0] = found;
        out_flags[0] = flag;
    }
}

// histogram of the sub-tiles' error bounds E (kernels.hpp: I6Slack) for the sizing below
__global__ __launch_bounds__(256) void i6_slack_hist_kernel(const float2* __restrict__ meta, uint32_t n_sub, uint32_t* __restrict__ hist) {
    __shared__ uint32_t sh[64];
    if (threadIdx.x < 64) sh[threadIdx.x] = 0u;
    __syncthreads();
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n_sub; i += gridDim.x * blockDim.x) {
        const float e = meta[i].y;
        int b = e == e ? (int)(e / I6_SLACK_STEP) : 63;  // (a NaN bound: the last bin)
        b = b < 0 ? 0 : (b > 63 ? 63 : b);
        atomicAdd(&sh[b], 1u);
    }
    __syncthreads();
    if (threadIdx.x < 64 && sh[threadIdx.x]) atomicAdd(&hist[threadIdx.x], sh[threadIdx.x]);
}
void launch_i6_slack_hist(const void* d_meta, uint32_t n_sub, uint32_t* d_hist, hipStream_t stream) {
    (void)hipMemsetAsync(d_hist, 0, 64 * sizeof(uint32_t), stream);
    if (n_sub == 0) return;
    const uint32_t blocks = (n_sub + 255u) / 256u < 1024u ? (n_sub + 255u) / 256u : 1024u;
    hipLaunchKernelGGL(i6_slack_hist_kernel, dim3(blocks), dim3(256), 0, stream, reinterpret_cast<const float2*>(d_meta), n_sub, d_hist);
}

// How many entries of its coarse list a wave keeps and refines.  The certificate needs the bound on every unlisted row — an
// exact score plus the packed shadow's slack E — below the k-th best score s_k, i.e. no wave may drop a row scoring above
// s_k - E - margin.  With scores ~ N(0, 1/384) (the thinnest top an index can have: uniform or Gaussian rows; clustered data has
// wider gaps) a wave holds Poisson(lambda) such rows, lambda = (N / waves) * (1 - Phi(z_k - (E + margin) sqrt(384))); n is the
// smallest list length that all waves together exceed with probability < 1 % (0: no list length does).  A wrong guess costs
// time (the certificate fails, the exact pass answers), never correctness.
int i6_refine_count(uint32_t n_rows, uint32_t k, int bits, int waves, const I6Slack* sl) {
    // (the last answer is kept per thread: a sharded handle issues its shards' searches from worker threads)
    thread_local uint32_t c_n = 0, c_k = 0, c_ver = 0;
    thread_local int c_bits = 0, c_waves = 0, c_out = LIST;
    thread_local const I6Slack* c_sl = nullptr;
    if (n_rows == c_n && k == c_k && bits == c_bits && waves == c_waves && sl == c_sl && (sl == nullptr || sl->version == c_ver)) return c_out;
    auto tail = [](double z) { return 0.5 * std::erfc(z / 1.4142135623730951); };
    const double kk = k < 1 ? 1.0 : (double)k;
    int n = LIST;
    if ((double)n_rows > 4.0 * kk) {
        double lo = 0.0, hi = 8.0;  // z_k: tail(z_k) = k / N
        for (int i = 0; i < 60; ++i) {
            const double mid = 0.5 * (lo + hi);
            if (tail(mid) * n_rows > kk) lo = mid;
            else hi = mid;
        }
        const double n_sub = std::ceil(n_rows / 32.0);  // (a small index does not reach every wave)
        const double holders = n_sub < (double)waves ? n_sub : (double)(waves > 0 ? waves : 1);
        // rows of a wave's share whose coarse bound reaches the k-th score: true score within the sub-tile's slack E + K2 of it.
        // A measured shadow (I6Slack: the histogram of its sub-tiles' E, upper bin edges, + 0.004 for K2 and the packed score's own
        // scatter, ~||dx|| / sqrt(384)) — or the constants of rounds 3-4, which predate the clipped scales and sit ~0.015 above
        // what uniform rows measure (three times the lambda: profiles/r05/refine_sweep.log).
        double share = 0.0;
        if (sl != nullptr) {
            for (int b = 0; b < 64; ++b)
                if (sl->frac[b] > 0.f) share += (double)sl->frac[b] * tail(lo - ((double)(b + 1) * I6_SLACK_STEP + 0.004) * 19.5959);
        } else {
            share = tail(lo - ((bits == 6 ? 0.040 : 0.082) + 0.012) * 19.5959);
        }
        const double lambda = share * n_rows / holders;
        const double allowed = 0.01 / holders;
        double term = std::exp(-lambda), cdf = term;  // P(X <= 0)
        n = 0;
        while (n < LIST && 1.0 - cdf > allowed) {  // smallest n with P(X > n) <= allowed
            ++n;
            term *= lambda / n;
            cdf += term;
        }
        // not even a full list is deep enough (k = 64 on 100 M rows at 5 bits): 0 — the caller streams the int8 shadow, whose
        // 64-row certificate has deeper rounds to fall back on, instead of paying an exact pass for a certificate that must fail
        if (1.0 - cdf > 0.05) n = -1;
    }
    // + 8, and a floor: real rows cluster — the pages of one site arrive together and fill a 32-row sub-tile, which is ONE wave's —
    // and a wave should be able to keep most of such a sub-tile on top of its ordinary share.  A refined entry costs ~0.8 us per
    // list position of the whole grid (a 1.5-KB row per entry and wave: 3 MB per position), so the floor is what the small indexes
    // pay: 40 on an unmeasured shadow; 24 on a measured one — 12.5 M rows, k = 10: 0.487 instead of 0.500 ms, no certificate lost
    // on uniform rows, 30 instead of 27 of 64 on the topical mixture (profiles/r05/refine_sweep.log) — an index that loses more
    // than one certificate in twenty has its lists raised to 64 by the ladder feedback (dawn_index.cpp: kFbBoost).
    const int floor_n = sl != nullptr ? 24 : 40;
    if (n < 0) n = 0;
    else n = (n + 8 < floor_n) ? floor_n : ((n + 8 + 7) & ~7);
    if (n > LIST) n = LIST;
    c_n = n_rows, c_k = k, c_bits = bits, c_waves = waves, c_out = n, c_sl = sl, c_ver = sl ? sl->version : 0u;
    return n;
}

// One query: stream + epilogue, merge + certificate.  ub_s / ub_p: the upper-bound lists [blocks][64] (what the other streams
// hand to merge_rescore_kernel; kept for the test hooks), ex_s / ex_p: the exact lists, tb [blocks]: the workgroups' bounds on
// their unlisted rows; d_i8 / d_i8meta: the int8 shadow (refinement); pool [32]: the stream's chunk counters (zero between searches).  geom.unroll: loads in flight per wave —
// 6 bits: 12 / 6 / 4 / 3 / 2 fragments of 768 B; 5 bits: 8 or 4 (anything else: 8) loads of 768 B - 1 KiB.
void launch_scan_i6(const void* d_i6, const void* d_meta, int bits, const void* d_i8, const void* d_i8meta, const void* d_rows,
                    int dtype, const uint64_t* d_ids, uint32_t n_rows, const float* d_q, float* ub_s, uint32_t* ub_p, float* ex_s,
                    uint32_t* ex_p, float* tb, uint32_t* pool, const ScanGeom& g,
                    uint32_t k, uint64_t* d_labels, float* d_dist, uint32_t* d_found, uint32_t* d_flags, int force_fallback,
                    bool merge, hipStream_t stream, hipEvent_t ev0, hipEvent_t ev1, uint32_t* d_stats, bool central_tail) {
    static OncePerDevice attr_once;
    const uint32_t* x = reinterpret_cast<const uint32_t*>(d_i6);
    const float2* mt = reinterpret_cast<const float2*>(d_meta);
#define DAWN_I6_EACH(F)                                                                                              \
    F(0, 6, 12) F(0, 6, 6) F(0, 6, 4) F(0, 6, 3) F(0, 6, 2) F(1, 6, 12) F(1, 6, 6) F(1, 6, 4) F(1, 6, 3) F(1, 6, 2) \
    F(0, 5, 8) F(0, 5, 4) F(1, 5, 8) F(1, 5, 4)
    once_per_device(attr_once, [] {
#define DAWN_I6_ATTR(RT_, BITS_, PD_)                                                                          \
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(scan_filter_i6s_kernel<RT_, BITS_, PD_>),         \
                              hipFuncAttributeMaxDynamicSharedMemorySize, RescoreStage<RT_>::BYTES);
        DAWN_I6_EACH(DAWN_I6_ATTR)
#undef DAWN_I6_ATTR
    });
    if (ev0) (void)hipEventRecord(ev0, stream);
    const int rt = dtype == ROW_BF16 ? 1 : 0;
    int n_refine = g.refine > 0 ? g.refine : i6_refine_count(n_rows, k, bits, g.blocks * (g.threads / 64));
    if (n_refine < 1 || n_refine > LIST) n_refine = LIST;  // (callers ask i6_refine_count first and go elsewhere on 0)
    n_refine = (n_refine + 7) & ~7;                         // (the f32 refinement takes its rows eight at a time)
    if (g.refine < 0) n_refine = -1;                        // (test hook: coarse lists, no refinement)
    int pd;
    if (bits == 6) pd = g.unroll == 6 || g.unroll == 4 || g.unroll == 3 || g.unroll == 2 ? g.unroll : 12;
    else pd = g.unroll == 4 ? 4 : 8;
#define DAWN_I6_LAUNCH(RT_, BITS_, PD_)                                                                                  \
    if (rt == RT_ && bits == BITS_ && pd == PD_)                                                                         \
        hipLaunchKernelGGL((scan_filter_i6s_kernel<RT_, BITS_, PD_>), dim3(g.blocks), dim3(g.threads),                  \
                           central_tail ? 0 : RescoreStage<RT_>::BYTES, stream, x, mt, n_rows, d_q, d_rows,              \
                           reinterpret_cast<const unsigned char*>(d_i8), reinterpret_cast<const float2*>(d_i8meta), ub_s, ub_p,  \
                           ex_s, ex_p, tb, n_refine, pool, central_tail ? 0 : 1, (uint32_t)(g.chunk > 0 ? g.chunk : 16) | ((uint32_t)(g.dyn_share > 0 ? g.dyn_share : 2) << 16));
    DAWN_I6_EACH(DAWN_I6_LAUNCH)
#undef DAWN_I6_LAUNCH
#undef DAWN_I6_EACH
    if (ev1) (void)hipEventRecord(ev1, stream);
    if (merge && central_tail)
        // the tail of the other streams on the refined lists: the 64 best rows of the index by the tight score rescored exactly, the
        // certificate against T = the largest of the workgroups' bounds (tb), deeper rounds and the 1024-row second chance behind it
        launch_merge_rescore(d_rows, dtype, d_ids, n_rows, d_q, 1, ub_s, ub_p, g.blocks, k, d_labels, d_dist, d_found, d_flags,
                             force_fallback, FILTER_EPS_I8, stream, pool, tb, d_stats);
    else if (merge)
        hipLaunchKernelGGL(merge_exact_kernel, dim3(1), dim3(1024), 0, stream, d_ids, n_rows, tb, ex_s, ex_p, g.blocks, k,
                           d_labels, d_dist, d_found, d_flags, force_fallback, FILTER_EPS_I8, pool, d_stats);
    else if (pool != nullptr)
        (void)hipMemsetAsync(pool, 0, 32 * sizeof(uint32_t), stream);  // (the test hook's stream-only launch)
}

}  // namespace dawn

#ifdef DAWN_EXPERIMENTS
extern "C" __attribute__((visibility("default"))) int dawn_debug_read_ts_i6(unsigned long long* out, int n) {
    static unsigned long long h[2048 * 5];
    if (hipMemcpyFromSymbol(h, HIP_SYMBOL(dawn_ts_i6), sizeof(h)) != hipSuccess) return -1;
    for (int i = 0; i < n && i < 2048 * 5; ++i) out[i] = h[i];
    return 0;
}
#endif
