// scan_i8.hip — int8 FILTER shadow of the index rows (ROW_I8S, kernels.hpp), its streaming filter (1..8 queries per pass;
// single queries by default) and its matrix-core pass (up to 256 queries; second half of this file).
//
// The filter stage of a search only has to produce, for every row, an UPPER BOUND on its exact score that is tight
// enough for the 64-row shortlist to contain the true top-k (the exact rescore and the certificate of
// merge_rescore_kernel do the rest, src/search/vector.rs:128-134 order).  Quarter the bytes of the f32 rows:
//
//   * rows are quantised per 32-row SUB-TILE: s = max|x| / 127 over the sub-tile, X = rint(x / s) in [-127, 127];
//     the sub-tile also stores E >= 1.0101 * max_r ||x_r - s X_r||_2 (measured at conversion time, not assumed);
//   * a sub-tile is 12 fragments of 1 KiB in the operand order of v_mfma_i32_32x32x32_i8: fragment f, lane
//     L = h*32 + r holds the 16 bytes k = 32f + 16h .. +15 of row r — a wave-wide 16-B load IS the A operand;
//   * the query enters as two int8 columns: q ~ s_q (H + L/254), H = rint(q / s_q), L = rint(254 (q/s_q - H)),
//     column c = H, column 8 + c = L; the integer accumulators combine exactly: C = 254 acc_H + acc_L
//     (|C| < 2^31: |acc_H| <= 384 * 127 * 127 < 2^23);
//   * integer MFMA accumulation is EXACT, so with x = sX + dx, q = q~ + dq:
//         x.q = s (s_q/254) C  +  dx.q  +  (sX).dq
//         |dx.q|    <= ||dx||_2 ||q||_2 <= E            (Cauchy-Schwarz; ||q||_2 < 1.01 by the is_normalized gate)
//         |(sX).dq| <= ||sX||_2 ||dq||_2 <= 1.1 * sqrt(384) * 2.0e-3 s_q =: K2(q)
//     (|dq_i| <= s_q (0.5/254 + 127 * 2^-23) < 2.0e-3 s_q; ||sX||_2 <= ||x||_2 + ||dx||_2 < 1.01 + 0.09).
//     ub = fma(float(C), s * s_q/254, E + K2) therefore bounds the real dot product from above up to the rounding of
//     this expression (< 1e-6: the product s * s_q comes from a stored 1 / s through v_rcp_f32, 1 ulp each) and the
//     reference's own sequential-sum error gamma_384 * 1.0201 = 2.34e-5; with the f32 rounding of the rotation below on both
//     operands (2 x 1.1e-6 x 1.0201) the budget is 2.67e-5: FILTER_EPS_I8 = 2.9e-5 (kernels.hpp) on top of ub.
//   * a bf16 index gets the same shadow of its (bf16-rounded) rows: ||x||_2 <= 1.01 (1 + 2^-8) keeps ||sX||_2 < 1.1, and
//     the exact side's gamma_384 * 1.0201 * 1.004 = 2.35e-5 stays inside FILTER_EPS_I8.
//   * the hot loop never leaves the integers: the lane's list threshold tau is turned into an integer threshold once
//     per sub-tile (5 VALU), conservatively (rows it passes are re-tested on ub itself).
//   * THE SHADOW LIVES IN A ROTATED BASIS.  E grows with max|x| of a sub-tile (s = max|x| / 127), and sentence embeddings
//     are heavy-tailed: a few dimensions are several times larger than the rest.  Measured on 100 M synthetic rows with 4
//     dimensions x5 (tools/cert_stats.py, profiles/r02/cert_stats_*.log): without the rotation 73-94 % of the queries of a
//     256-batch failed both certificates and took the exact pass — 7-9 SECONDS per batch instead of 10 ms.  Rows and
//     queries are therefore multiplied by one fixed orthogonal matrix R before they are quantised (the filter only needs
//     x.q = (Rx).(Rq)):  R = (H_128 / sqrt(128) (x) I_3) . (M_3 (x) I_128) . diag(+-1)  — pseudo-random signs, the
//     orthogonal 3x3 mix M_3 = (2/3) J - I across the three 128-blocks, a 128-point Walsh-Hadamard transform in each block:
//     every output is a signed sum of all 384 inputs, so one large dimension is spread evenly (x a / sqrt(384)) and
//     whatever the data looked like, the quantiser sees near-Gaussian components with max|x'| ~ 4 / sqrt(384).  13 adds
//     per element, in registers (DPP / permlane butterflies across lanes), at conversion and query-preparation time; the
//     scans themselves are unchanged.  Its f32 rounding (<= 1.1e-6 ||x|| per vector: 7 butterfly stages, the 3x3 mix, the
//     scale) is part of FILTER_EPS_I8.  Correctness never depends on what the rotation achieves: E and K2 are measured on
//     the rotated values.
#include <type_traits>

#include "kernels.hpp"
#include "rotate384.hpp"
#include "wave_topk.hpp"

#ifdef DAWN_EXPERIMENTS
// timestamp probes (100-MHz counter) of scan_filter_i8s_kernel: workgroups 0 / 85 / 170 / 255, first and last wave, 8 points
// (dawn_debug_read_ts_i8; tools/stream_ts.py)
static __device__ unsigned long long dawn_ts_i8[4 * 2 * 8];
// ... and the time every wave of scan_i8_pipe16_kernel leaves its last launch (dawn_debug_read_ts_pass; tools/pass_ts.py)
static __device__ unsigned long long dawn_ts_pass[256 * 4];
#define DAWN_TSI(i)                                                                                            \
    do {                                                                                                       \
        if ((threadIdx.x & 63) == 0 && blockIdx.x % 85 == 0 && ((threadIdx.x >> 6) == 0 || (threadIdx.x >> 6) == (blockDim.x >> 6) - 1)) \
            dawn_ts_i8[((blockIdx.x / 85) * 2 + ((threadIdx.x >> 6) != 0)) * 8 + (i)] = __builtin_amdgcn_s_memrealtime();               \
    } while (0)
#else
#define DAWN_TSI(i)
#endif

namespace dawn {


// ------------------------------------------------------------------------------------------------
// conversion: f32 rows -> int8 sub-tiles + {s, E} per sub-tile
// ------------------------------------------------------------------------------------------------
// One 256-thread block per sub-tile; thread = (row r = tid / 8, part = tid % 8) handles float4 chunks part + 8j.
// RT = 0: f32 rows; RT = 1: the fragment-ordered bf16 rows of a bf16 index (the int8 copy then shadows THOSE values).
template <int RT>
__global__ __launch_bounds__(256) void rows_to_i8s_kernel(const void* __restrict__ xv, uint32_t* __restrict__ out,
                                                           float2* __restrict__ meta, uint32_t first_sub,
                                                           uint32_t n_valid, float levels) {
    __shared__ float sh[4];
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
            } else {  // values 4 c4 .. +3 = half (c4 & 1) of 16-B chunk c4 / 2
                const u32x4 w = reinterpret_cast<const u32x4*>(xv)[frag_chunk(row, (int)(c4 >> 1))];
                const uint32_t w0 = (c4 & 1u) ? w.z : w.x, w1 = (c4 & 1u) ? w.w : w.y;
                v[j] = f32x4{bf16_lo(w0), bf16_hi(w0), bf16_lo(w1), bf16_hi(w1)};
            }
        }
    }
    rotate384_rowpart(v, part, lane);  // (rows past n_valid stay zero)
    float amax = 0.f;
#pragma unroll
    for (int j = 0; j < 12; ++j)
        amax = fmaxf(amax, fmaxf(fmaxf(fabsf(v[j].x), fabsf(v[j].y)), fmaxf(fabsf(v[j].z), fabsf(v[j].w))));
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) amax = fmaxf(amax, __shfl_xor(amax, o));
    if (lane == 0) sh[wave] = amax;
    __syncthreads();
    amax = fmaxf(fmaxf(sh[0], sh[1]), fmaxf(sh[2], sh[3]));
    __syncthreads();
    const float s = fmaxf(amax, 1e-20f) / levels;  // levels = 127 (option "debug_i8_levels": fewer, to study coarser shadows)
    float e2 = 0.f;
    uint32_t* o = out + (size_t)sub * (12 * 256);  // 12 KiB per sub-tile, in dwords
#pragma unroll
    for (int j = 0; j < 12; ++j) {
        const uint32_t c4 = part + 8 * j;  // k = 4 c4 .. +3
        int X[4];
        const float vv[4] = {v[j].x, v[j].y, v[j].z, v[j].w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float t = rintf(vv[i] / s);
            t = fminf(fmaxf(t, -levels), levels);
            X[i] = (int)t;
            const float dx = vv[i] - s * t;
            e2 = __builtin_fmaf(dx, dx, e2);
        }
        const uint32_t w = (uint32_t)(X[0] & 255) | ((uint32_t)(X[1] & 255) << 8) | ((uint32_t)(X[2] & 255) << 16) |
                           ((uint32_t)(X[3] & 255) << 24);
        // fragment c4 / 8, half (c4 % 8) / 4, dword c4 % 4 of lane (h, r)
        o[(c4 >> 3) * 256u + (((c4 >> 2) & 1u) * 32u + r) * 4u + (c4 & 3u)] = w;
    }
    // row sum over its 8 threads (fixed order), then the block maximum
    e2 += __shfl_xor(e2, 1);
    e2 += __shfl_xor(e2, 2);
    e2 += __shfl_xor(e2, 4);
#pragma unroll
    for (int o2 = 32; o2 >= 8; o2 >>= 1) e2 = fmaxf(e2, __shfl_xor(e2, o2));
    if (lane == 0) sh[wave] = e2;
    __syncthreads();
    if (tid == 0) {
        const float m = fmaxf(fmaxf(sh[0], sh[1]), fmaxf(sh[2], sh[3]));
        // 1.0101: ||q||_2 < 1.01 (gate); 1.001 + 1e-9: the f32 evaluation of dx, the sum and the square root
        // stored: {1 / s, E}: the filters' integer thresholds need the reciprocal once per sub-tile, s itself only on a hit
        meta[sub] = float2{1.0f / s, sqrtf(m) * 1.0101f * 1.001f + 1e-9f};
    }
}


// rt: ROW_F32 (f32 rows) or ROW_BF16 (fragment-ordered bf16 rows)
void launch_rows_to_i8s(const void* d_rows, int rt, void* d_shadow, void* d_meta, size_t first_row, size_t n_valid,
                        hipStream_t stream, float levels) {
    const uint32_t first_sub = (uint32_t)(first_row / 32);  // the sub-tile holding first_row is re-quantised whole
    const uint32_t end_sub = (uint32_t)((n_valid + 31) / 32);
    if (end_sub <= first_sub) return;
    if (rt == ROW_BF16)
        hipLaunchKernelGGL(rows_to_i8s_kernel<1>, dim3(end_sub - first_sub), dim3(256), 0, stream, d_rows,
                           reinterpret_cast<uint32_t*>(d_shadow), reinterpret_cast<float2*>(d_meta), first_sub, (uint32_t)n_valid, levels);
    else
        hipLaunchKernelGGL(rows_to_i8s_kernel<0>, dim3(end_sub - first_sub), dim3(256), 0, stream, d_rows,
                           reinterpret_cast<uint32_t*>(d_shadow), reinterpret_cast<float2*>(d_meta), first_sub, (uint32_t)n_valid, levels);
}

// ------------------------------------------------------------------------------------------------
// streaming filter
// ------------------------------------------------------------------------------------------------
// Lists hold ub (see the header) as the score.  q: the n_q <= QB <= 8 queries, f32 [n_q][384].
template <bool NT>
__device__ __forceinline__ u32x4 row_load(const u32x4* p) {
    if (NT) return __builtin_nontemporal_load(p);
    return *p;
}

// NT: non-temporal row loads (default) or plain ones (geometry experiments)
template <int QB, int PD, bool NT = true>
__global__ __launch_bounds__(512) void scan_filter_i8s_kernel(const u32x4* __restrict__ x, const float2* __restrict__ meta,
                                                               uint32_t n_rows, const float* __restrict__ q, int n_q,
                                                               float* __restrict__ out_s, uint32_t* __restrict__ out_p,
                                                               uint32_t q_stride_lists) {
    static_assert(12 % PD == 0, "the ring must divide the 12 k-steps of a sub-tile");
    __shared__ float sh_s[16][LIST];
    __shared__ uint32_t sh_p[16][LIST];
    __shared__ float sh_strip[16][32];  // (QB == 1) a sub-tile's 32 upper bounds, one strip per wave
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nwaves = blockDim.x >> 6;
    const uint32_t gwave = blockIdx.x * nwaves + wave;
    const uint32_t total_waves = gridDim.x * nwaves;
    const uint32_t n_sub = (n_rows + 31u) >> 5;
    const uint32_t c = lane & 31, h = lane >> 5;
    DAWN_TSI(0);
    // the wave's first fragments are requested before anything else: they fly while the query images are made (~3 us of
    // dependent work at the head of a kernel that lasts 75 us on 1 M rows)
    uint32_t t = gwave;
    const u32x4* p = x + (size_t)(t < n_sub ? t : 0) * (12 * 64) + lane;
    u32x4 a[PD];
    float2 mt = {0.f, 0.f};
    if (t < n_sub) {
#pragma unroll
        for (int d = 0; d < PD; ++d) a[d] = row_load<NT>(p + d * 64);
        mt = meta[t];
    }

    // B operand: column c < 8 = H of query c, column 8 + c = L; lane (h, c) holds k = 32f + 16h .. +15 of k-step f.
    // The two int8 images of a query are built ONCE per workgroup, by the wave that rotates it (lane l holds elements
    // l + 64 j: wave maximum -> s_q, six H and six L bytes per lane into LDS), and every wave reads its twelve fragments
    // back as 16-B chunks.  (Each lane of columns 0 / 8 quantising all 384 elements itself — 192 divisions and a 96-step
    // maximum per lane, in every wave — made the kernel take 30 us on an index of 4096 rows.)
    __shared__ __attribute__((aligned(16))) signed char sh_img[2][QB][EM];  // [H | L][query][k]
    __shared__ float sh_sq[QB];
    for (int b = wave; b < QB; b += nwaves) {
        float v[6];
#pragma unroll
        for (int j = 0; j < 6; ++j) v[j] = b < n_q ? q[(size_t)b * EM + lane + 64 * j] : 0.f;
        rotate384_wave(v, lane);  // the shadow's basis (header of this file)
        float amax = 0.f;
#pragma unroll
        for (int j = 0; j < 6; ++j) amax = fmaxf(amax, fabsf(v[j]));
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) amax = fmaxf(amax, __shfl_xor(amax, o));
        const float sq = fmaxf(amax, 1e-20f) / 127.0f;
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            const float t = v[j] / sq;
            const float H = fminf(fmaxf(rintf(t), -127.f), 127.f);
            const float L = fminf(fmaxf(rintf((t - H) * 254.0f), -127.f), 127.f);
            sh_img[0][b][lane + 64 * j] = b < n_q ? (signed char)(int)H : (signed char)0;
            sh_img[1][b][lane + 64 * j] = b < n_q ? (signed char)(int)L : (signed char)0;
        }
        if (lane == 0) sh_sq[b] = sq;
    }
    __syncthreads();
    DAWN_TSI(1);
    i32x4_t qf[12];
    const int qcol = (int)(c & 7u);
    float sq254_l = 0.f, k2_l = 0.f, rsq254_l = 0.f;  // this lane's query: s_q / 254, K2, 254 / s_q
    {
        const bool col_live = qcol < n_q && qcol < QB && c < 16;
        const i32x4_t* img = reinterpret_cast<const i32x4_t*>(&sh_img[c >= 8 ? 1 : 0][col_live ? qcol : 0][0]);
#pragma unroll
        for (int f = 0; f < 12; ++f) qf[f] = col_live ? img[2 * f + h] : i32x4_t{0, 0, 0, 0};
        if (col_live) {
            const float sq = sh_sq[qcol];
            sq254_l = sq / 254.0f;
            rsq254_l = 254.0f / sq;
            k2_l = I8_K2_PER_SQ * sq;
        }
    }
    float ls[QB], tau[QB], sq254[QB], k2[QB];
    uint32_t lp[QB];
#pragma unroll
    for (int b = 0; b < QB; ++b) {
        ls[b] = NEG_INF;
        lp[b] = NO_POS;
        tau[b] = NEG_INF;
        // wave-uniform copies of the per-query constants (lane b holds query b's)
        sq254[b] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, sq254_l), b));
        k2[b] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, k2_l), b));
    }
    // the lane's threshold, lowered by its own rounding allowance and K2: a row is tested further iff
    // C > floor(((tau_m - E (1 + 1e-6)) / (s s_q/254)) - 2); +inf in the columns that hold no H
    const bool tested = (int)c < n_q && c < 8;
    float tau_m = tested ? NEG_INF : __builtin_inff();

    DAWN_TSI(2);
    [[maybe_unused]] int ts_iter = 0;
    if (t < n_sub) {
        for (;;) {
            if (ts_iter == 1) DAWN_TSI(3);
            if (ts_iter == 2) DAWN_TSI(4);
            ++ts_iter;
            const uint32_t tn = t + total_waves;
            const bool more = tn < n_sub;
            // the ring runs into the wave's next sub-tile (the last one re-reads its own first fragments: no branch)
            const u32x4* pn = more ? x + (size_t)tn * (12 * 64) + lane : p;
            const float2 mtn = meta[more ? tn : t];
            i32x16_t acc;
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[e] = 0;
#pragma unroll
            for (int f = 0; f < 12; ++f) {
                acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(__builtin_bit_cast(i32x4_t, a[f % PD]), qf[f], acc, 0, 0, 0);
                if (f + PD < 12) a[f % PD] = row_load<NT>(p + (f + PD) * 64);
                else a[f % PD] = row_load<NT>(pn + (f + PD - 12) * 64);
                __builtin_amdgcn_sched_barrier(0);  // keep every load PD steps ahead of its use
            }
            // this lane: D[row = 32t + (e&3) + 8*(e>>2) + 4h][column c];  C = 254 acc_H + acc_L (lane c <- lane c + 8)
            // mt = {1 / s, E};  units of C per unit of score = (1 / s) * (254 / s_q)
            const float u = __builtin_fmaf(-mt.y, 1.000001f, tau_m);
            float thr_f = __builtin_fmaf(u, mt.x * rsq254_l, -2.0f);
            thr_f = fminf(fmaxf(thr_f, -2.0e9f), 2.0e9f);
            if (!tested) thr_f = 2.0e9f;  // (0 * inf above)
            const int thr = (int)floorf(thr_f);
            int C[16];
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int ae = acc[e];
                C[e] = __mul24(ae, 254) + __builtin_amdgcn_update_dpp(0, ae, 0x108, 0xf, 0xf, true);
            }
            int mx = C[0];
#pragma unroll
            for (int e = 1; e < 16; ++e) mx = max(mx, C[e]);
            if (__any(mx > thr)) {
                const uint32_t row_base = t * 32u;
                bool merged = false;
                if constexpr (QB == 1) {
                    // Many hits in one sub-tile — the first sub-tiles of every wave, while its list fills: a wave's first 64
                    // rows all enter, the j-th sub-tile still contributes 64 / (j + 1) — are merged as ONE sorted batch instead
                    // of one ballot-and-shift insertion each: the 32 upper bounds, held by the two lanes of column 0, go through
                    // a 128-B strip of LDS to one per lane, a bitonic sort and one merge64.  Same list afterwards (rows arrive in
                    // ascending order: a later row never displaces an equal score).
                    int nh = 0;
#pragma unroll
                    for (int e = 0; e < 16; ++e)
                        nh += (C[e] > thr && row_base + (uint32_t)((e & 3) + 8 * (e >> 2)) + 4u * h < n_rows) ? 1 : 0;
                    const int total = __builtin_amdgcn_readlane(nh, 0) + __builtin_amdgcn_readlane(nh, 32);
                    if (total > 6) {
                        float* strip = &sh_strip[wave][0];
                        if (c == 0) {  // lanes 0 (h = 0) and 32 (h = 1)
                            const float g1 = __builtin_amdgcn_rcpf(mt.x) * sq254[0], g0 = mt.y + k2[0];
#pragma unroll
                            for (int e = 0; e < 16; ++e) {
                                const uint32_t roff = (uint32_t)((e & 3) + 8 * (e >> 2)) + 4u * h;
                                const bool ok = C[e] > thr && row_base + roff < n_rows;
                                strip[roff] = ok ? __builtin_fmaf((float)C[e], g1, g0) : NEG_INF;
                            }
                        }
                        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // (wave-private strip: LDS operations of a wave are in order)
                        float d = POS_INF;
                        uint32_t row = NO_POS;
                        if (lane < 32) {
                            const float sc = strip[lane];
                            if (sc > tau[0]) {
                                d = -sc;
                                row = row_base + (uint32_t)lane;
                            }
                        }
                        asm volatile("" ::: "memory");
                        sort64_asc(d, row, lane);
                        const float os = -__shfl(d, 63 - lane);
                        const uint32_t op = __shfl(row, 63 - lane);
                        merge64(ls[0], lp[0], os, op, lane);
                        tau[0] = read_lane63(ls[0]);
                        merged = true;
                    }
                }
                if (!merged) {
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const uint32_t roff = (uint32_t)((e & 3) + 8 * (e >> 2));
                    // rows past the end (zero padding) never enter a list
                    unsigned long long m = __ballot(C[e] > thr && row_base + roff + 4u * h < n_rows);
                    while (m) {
                        const int l = __builtin_ctzll(m);
                        m &= m - 1;
                        const float cf = (float)__builtin_amdgcn_readlane(C[e], l);
                        const int qb = l & 31;
                        const uint32_t row = row_base + roff + 4u * (uint32_t)(l >> 5);
#pragma unroll
                        for (int b = 0; b < QB; ++b) {
                            if (b == qb) {
                                const float sc = __builtin_fmaf(cf, __builtin_amdgcn_rcpf(mt.x) * sq254[b], mt.y + k2[b]);
                                if (sc > tau[b]) {
                                    wave_insert(ls[b], lp[b], sc, row, lane);
                                    tau[b] = read_lane63(ls[b]);
                                }
                            }
                        }
                    }
                }
                }
#pragma unroll
                for (int b = 0; b < QB; ++b)
                    if ((int)c == b && tested) {
                        const float tk = tau[b] - k2[b];
                        tau_m = tk - fabsf(tk) * 1e-6f;
                    }
            }
            if (!more) break;
            t = tn;
            p = pn;
            mt = mtn;
        }
    }

    DAWN_TSI(5);
#pragma unroll
    for (int b = 0; b < QB; ++b) {
        if (b < n_q) {  // uniform
            block_merge(ls[b], lp[b], sh_s, sh_p, wave, lane, nwaves);
            DAWN_TSI(6);
            if (wave == 0) {
                const size_t o = ((size_t)b * q_stride_lists + blockIdx.x) * LIST + lane;
                out_s[o] = ls[b];
                out_p[o] = lp[b];
            }
        }
    }
    DAWN_TSI(7);
}

// ---- the same filter, software-pipelined (round 3; the default) ------------------------------------------------------
// The kernel above tests a sub-tile right after its twelve MFMAs: ~60 VALU instructions during which the wave issues no
// load, every sub-tile — a 150-clock hole in the wave's request stream per 770 clocks of matrix work.  The bare-read
// probe (tools/probes/hbm_read.hip, profiles/r03/hbm_read_probe.log) shows what that costs on this chip: with loads
// re-issued the moment a fragment lands, TWO waves per CU x 12 KiB in flight read 7.18-7.23 TB/s (0.90 of the 8 TB/s
// spec), four waves 7.03-7.08 — fewer bytes in flight are faster, provided the stream of requests never pauses.
// Here the test of sub-tile t - 1 is cut into slices that sit in the shadow of sub-tile t's MFMAs (a 32x32x32 int8 MFMA
// keeps the matrix pipe busy for 64 clocks; the slices are <= 6 VALU instructions each), two accumulator sets, so loads
// are re-issued at the steady cadence of the MFMAs and nothing else is on the wave's critical path.  Results are
// identical to the kernel above by construction (same scores, same tests in the same order).
// XCD: workgroup b runs on XCD b % 8 (round-robin dispatch); each XCD's workgroups sweep their own contiguous eighth of
// the shadow instead of the chip-wide window (probe: +0.6 % at two waves per CU).
template <int QB, bool XCD, int PD = 12>
__global__ __launch_bounds__(512) void scan_filter_i8s_pipe_kernel(const u32x4* __restrict__ x, const float2* __restrict__ meta,
                                                                    uint32_t n_rows, const float* __restrict__ q, int n_q,
                                                                    float* __restrict__ out_s, uint32_t* __restrict__ out_p,
                                                                    uint32_t q_stride_lists) {
    __shared__ float sh_s[16][LIST];
    __shared__ uint32_t sh_p[16][LIST];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nwaves = blockDim.x >> 6;
    const uint32_t n_sub = (n_rows + 31u) >> 5;
    const uint32_t c = lane & 31, h = lane >> 5;
    // this wave's sub-tiles: first, first + stride, ... < end
    uint32_t t, t_stride, t_end;
    if (XCD) {
        const uint32_t xcd = blockIdx.x & 7u, j = blockIdx.x >> 3, per = (gridDim.x + 7u) >> 3;
        const uint32_t lo = (uint32_t)((uint64_t)n_sub * xcd / 8), hi = (uint32_t)((uint64_t)n_sub * (xcd + 1) / 8);
        t = lo + j * nwaves + wave;
        t_stride = per * nwaves;
        t_end = hi;
    } else {
        t = blockIdx.x * nwaves + wave;
        t_stride = gridDim.x * nwaves;
        t_end = n_sub;
    }
    // the first fragments fly while the query images are made (see scan_filter_i8s_kernel)
    static_assert(12 % PD == 0, "the ring must divide the 12 k-steps of a sub-tile");
    const u32x4* p = x + (size_t)(t < t_end ? t : 0) * (12 * 64) + lane;
    u32x4 a[PD];
    float2 mt = {0.f, 0.f};
    if (t < t_end) {
#pragma unroll
        for (int d = 0; d < PD; ++d) a[d] = row_load<true>(p + d * 64);
        mt = meta[t];
    }

    // query images: as in scan_filter_i8s_kernel
    __shared__ __attribute__((aligned(16))) signed char sh_img[2][QB][EM];
    __shared__ float sh_sq[QB];
    for (int b = wave; b < QB; b += nwaves) {
        float v[6];
#pragma unroll
        for (int j = 0; j < 6; ++j) v[j] = b < n_q ? q[(size_t)b * EM + lane + 64 * j] : 0.f;
        rotate384_wave(v, lane);
        float amax = 0.f;
#pragma unroll
        for (int j = 0; j < 6; ++j) amax = fmaxf(amax, fabsf(v[j]));
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) amax = fmaxf(amax, __shfl_xor(amax, o));
        const float sq = fmaxf(amax, 1e-20f) / 127.0f;
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            const float tt = v[j] / sq;
            const float H = fminf(fmaxf(rintf(tt), -127.f), 127.f);
            const float L = fminf(fmaxf(rintf((tt - H) * 254.0f), -127.f), 127.f);
            sh_img[0][b][lane + 64 * j] = b < n_q ? (signed char)(int)H : (signed char)0;
            sh_img[1][b][lane + 64 * j] = b < n_q ? (signed char)(int)L : (signed char)0;
        }
        if (lane == 0) sh_sq[b] = sq;
    }
    __syncthreads();
    i32x4_t qf[12];
    const int qcol = (int)(c & 7u);
    float sq254_l = 0.f, k2_l = 0.f, rsq254_l = 0.f;
    {
        const bool col_live = qcol < n_q && qcol < QB && c < 16;
        const i32x4_t* img = reinterpret_cast<const i32x4_t*>(&sh_img[c >= 8 ? 1 : 0][col_live ? qcol : 0][0]);
#pragma unroll
        for (int f = 0; f < 12; ++f) qf[f] = col_live ? img[2 * f + h] : i32x4_t{0, 0, 0, 0};
        if (col_live) {
            const float sq = sh_sq[qcol];
            sq254_l = sq / 254.0f;
            rsq254_l = 254.0f / sq;
            k2_l = I8_K2_PER_SQ * sq;
        }
    }
    float ls[QB], tau[QB], sq254[QB], k2[QB];
    uint32_t lp[QB];
#pragma unroll
    for (int b = 0; b < QB; ++b) {
        ls[b] = NEG_INF;
        lp[b] = NO_POS;
        tau[b] = NEG_INF;
        sq254[b] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, sq254_l), b));
        k2[b] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, k2_l), b));
    }
    const bool tested = (int)c < n_q && c < 8;
    float tau_m = tested ? NEG_INF : __builtin_inff();

    if (t < t_end) {
        // state of the sub-tile under test (the previous one)
        i32x16_t accs[2];  // ping-pong: one being accumulated, the other under test (no copies)
        float2 pmt = mt;
        uint32_t prow = 0;
        int C[16];
        int thr = 0, mx = 0;

        // the slow path of a tested sub-tile: exactly the kernel above's
        auto slow_path = [&]() __attribute__((always_inline)) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const uint32_t roff = (uint32_t)((e & 3) + 8 * (e >> 2));
                unsigned long long m = __ballot(C[e] > thr && prow + roff + 4u * h < n_rows);
                while (m) {
                    const int l = __builtin_ctzll(m);
                    m &= m - 1;
                    const float cf = (float)__builtin_amdgcn_readlane(C[e], l);
                    const int qb = l & 31;
                    const uint32_t row = prow + roff + 4u * (uint32_t)(l >> 5);
#pragma unroll
                    for (int b = 0; b < QB; ++b) {
                        if (b == qb) {
                            const float sc = __builtin_fmaf(cf, __builtin_amdgcn_rcpf(pmt.x) * sq254[b], pmt.y + k2[b]);
                            if (sc > tau[b]) {
                                wave_insert(ls[b], lp[b], sc, row, lane);
                                tau[b] = read_lane63(ls[b]);
                            }
                        }
                    }
                }
            }
#pragma unroll
            for (int b = 0; b < QB; ++b)
                if ((int)c == b && tested) {
                    const float tk = tau[b] - k2[b];
                    tau_m = tk - fabsf(tk) * 1e-6f;
                }
        };
        // slice s (0..11) of the test of the previous sub-tile; slices 1..8 turn two accumulators each into C and fold them
        // into the maximum, slice 0 makes the integer threshold
        auto test_slice = [&](int s, const i32x16_t& pacc) __attribute__((always_inline)) {
            if (s == 0) {
                const float u = __builtin_fmaf(-pmt.y, 1.000001f, tau_m);
                float thr_f = __builtin_fmaf(u, pmt.x * rsq254_l, -2.0f);
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

        // one round: the twelve MFMAs of sub-tile t (its fragments are in the ring), the ring refilled from the wave's next
        // sub-tile, and — TEST — the slices of the previous sub-tile's test in the MFMAs' shadow.  The first round has no
        // previous sub-tile and is a copy of its own without the slices (a flag tested in the loop would let the compiler
        // sink the slices behind the MFMAs, into the branch that uses them).
        bool more;
        auto round = [&](auto with_test, auto parity) __attribute__((always_inline)) {
            constexpr int P = decltype(parity)::value;
            i32x16_t& acc = accs[P];
            const uint32_t tn = t + t_stride;
            more = tn < t_end;
            // (the last sub-tile re-reads its own first fragments: no branch in the stream)
            const u32x4* pn = more ? x + (size_t)tn * (12 * 64) + lane : p;
            const float2 mtn = meta[more ? tn : t];
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[e] = 0;
#pragma unroll
            for (int f = 0; f < 12; ++f) {
                acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(__builtin_bit_cast(i32x4_t, a[f % PD]), qf[f], acc, 0, 0, 0);
                if (f + PD < 12) a[f % PD] = row_load<true>(p + (f + PD) * 64);
                else a[f % PD] = row_load<true>(pn + (f + PD - 12) * 64);
                if constexpr (decltype(with_test)::value) test_slice(f, accs[1 - P]);
                __builtin_amdgcn_sched_barrier(0);
            }
            if constexpr (decltype(with_test)::value)
                if (__any(mx > thr)) slow_path();
            // this sub-tile is the next round's test
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
        // the last sub-tile's test, unpipelined
        if (last) {
#pragma unroll
            for (int s = 0; s < 12; ++s) test_slice(s, accs[1]);
        } else {
#pragma unroll
            for (int s = 0; s < 12; ++s) test_slice(s, accs[0]);
        }
        if (__any(mx > thr)) slow_path();
    }

#pragma unroll
    for (int b = 0; b < QB; ++b) {
        if (b < n_q) {
            block_merge(ls[b], lp[b], sh_s, sh_p, wave, lane, nwaves);
            if (wave == 0) {
                const size_t o = ((size_t)b * q_stride_lists + blockIdx.x) * LIST + lane;
                out_s[o] = ls[b];
                out_p[o] = lp[b];
            }
        }
    }
}

template <int QB>
static void launch_filter_i8s_qb(const void* d_shadow, const void* d_meta, uint32_t n_rows, const float* q8, int n_q,
                                 float* cand_s, uint32_t* cand_p, const ScanGeom& g, hipStream_t stream) {
    const u32x4* x8 = reinterpret_cast<const u32x4*>(d_shadow);
    const float2* mt = reinterpret_cast<const float2*>(d_meta);
    // geom.unroll picks the ring: 3 (default) -> 12 fragments (KiB per wave) ahead of the MFMAs; 1 / 2 / 4 -> 3 / 4 / 6
#define DAWN_I8S_LAUNCH(PD_)                                                                                        \
    hipLaunchKernelGGL((scan_filter_i8s_kernel<QB, PD_>), dim3(g.blocks), dim3(g.threads), 0, stream, x8, mt, n_rows, q8, \
                       n_q, cand_s, cand_p, (uint32_t)g.blocks)
    switch (g.unroll) {
        case 1: DAWN_I8S_LAUNCH(3); break;
        case 2: DAWN_I8S_LAUNCH(4); break;
        case 4: DAWN_I8S_LAUNCH(6); break;
        case 6:  // software-pipelined test (two accumulator sets), chip-wide window
            hipLaunchKernelGGL((scan_filter_i8s_pipe_kernel<QB, false>), dim3(g.blocks), dim3(g.threads), 0, stream, x8, mt, n_rows,
                               q8, n_q, cand_s, cand_p, (uint32_t)g.blocks);
            break;
        case 8:  // ... with a ring of 6 fragments
            hipLaunchKernelGGL((scan_filter_i8s_pipe_kernel<QB, false, 6>), dim3(g.blocks), dim3(g.threads), 0, stream, x8, mt, n_rows,
                               q8, n_q, cand_s, cand_p, (uint32_t)g.blocks);
            break;
        case 9:  // ... of 4
            hipLaunchKernelGGL((scan_filter_i8s_pipe_kernel<QB, false, 4>), dim3(g.blocks), dim3(g.threads), 0, stream, x8, mt, n_rows,
                               q8, n_q, cand_s, cand_p, (uint32_t)g.blocks);
            break;
        case 10:  // ... of 3
            hipLaunchKernelGGL((scan_filter_i8s_pipe_kernel<QB, false, 3>), dim3(g.blocks), dim3(g.threads), 0, stream, x8, mt, n_rows,
                               q8, n_q, cand_s, cand_p, (uint32_t)g.blocks);
            break;
        case 7:  // ... and per-XCD contiguous ranges
            hipLaunchKernelGGL((scan_filter_i8s_pipe_kernel<QB, true>), dim3(g.blocks), dim3(g.threads), 0, stream, x8, mt, n_rows,
                               q8, n_q, cand_s, cand_p, (uint32_t)g.blocks);
            break;
        case 5:  // 12 fragments, plain (temporal) loads
            hipLaunchKernelGGL((scan_filter_i8s_kernel<QB, 12, false>), dim3(g.blocks), dim3(g.threads), 0, stream, x8, mt, n_rows,
                               q8, n_q, cand_s, cand_p, (uint32_t)g.blocks);
            break;
        default: DAWN_I8S_LAUNCH(12); break;
    }
#undef DAWN_I8S_LAUNCH
}

void launch_scan_filter_i8s(const void* d_shadow, const void* d_meta, uint32_t n_rows, const float* d_q, int B, float* cand_s,
                            uint32_t* cand_p, const ScanGeom& g, hipStream_t stream, hipEvent_t ev0, hipEvent_t ev1) {
    if (ev0) (void)hipEventRecord(ev0, stream);
    const size_t per_q = (size_t)g.blocks * LIST;
    for (int b = 0; b < B; b += 8) {  // 8 queries per pass over the index
        const int nb = B - b < 8 ? B - b : 8;
        const float* q = d_q + (size_t)b * EM;
        float* cs = cand_s + (size_t)b * per_q;
        uint32_t* cp = cand_p + (size_t)b * per_q;
        if (nb <= 1) launch_filter_i8s_qb<1>(d_shadow, d_meta, n_rows, q, nb, cs, cp, g, stream);
        else if (nb <= 4) launch_filter_i8s_qb<4>(d_shadow, d_meta, n_rows, q, nb, cs, cp, g, stream);
        else launch_filter_i8s_qb<8>(d_shadow, d_meta, n_rows, q, nb, cs, cp, g, stream);
    }
    if (ev1) (void)hipEventRecord(ev1, stream);
}


// ------------------------------------------------------------------------------------------------
// matrix-core filter for 4..256 queries per pass on the int8 shadow
// ------------------------------------------------------------------------------------------------
// The software pipeline of scan_f16_pipe_kernel (scan_batched.hip: one wave per SIMD, 64 queries per wave, three
// 48-KiB tile images filled by LDS-DMA, never-draining fragment ring, two accumulator sets, per-wave candidate stage)
// on v_mfma_i32_32x32x32_i8: a 48-KiB image is FOUR 32-row sub-tiles of 12 fragments (128 rows), every MFMA covers 32
// k instead of 16, so the matrix pipe needs half the instructions per row.  The query enters as ONE int8 image
// (X_q = rint(q / s_q)); with the measured ||q - s_q X_q||_2 the score
//     ub = fma(float(acc), s * s_q, E + K2),    K2 = 1.1 * ||dq||_2   (>= |(sX).dq|, see the header of this file)
// is an upper bound of the real dot product.  The threshold test stays in the integers: per sub-tile and query group
// one integer threshold from {s, E} of the sub-tile (6 VALU), then v_max3_i32 slices in the shadow of the MFMAs.
// {s, E} of a tile's four sub-tiles (32 B) travel beside the DMA: two global_load_dwordx4 issued in front of the DMA
// of tile t+2 (vmcnt is in order: the wait that publishes tile t+1 covers them), copied into the registers the tests
// read once the last test of tile t-1 is done.
constexpr int I8_TILE_ROWS = 128;
constexpr int I8_TILE_BYTES = I8_TILE_ROWS * EM;  // 49152
constexpr uint32_t I8_ECAP = 40;                  // staged hit entries per wave ...
constexpr uint32_t I8_EDW = 24;                   // ... of 24 dwords: 16 accumulators, query, first row, threshold, s s_q, E + K2
constexpr uint32_t I8_SEG_CAP = (uint32_t)BATCH_CAP / (uint32_t)BATCH_CAND_SEGS;

// queries -> int8 images [BATCH_QT][384] + {s_q, K2} per query; rows b >= n_q: zeros.  One wave per query.
__global__ __launch_bounds__(64) void prep_queries_i8_kernel(const float* __restrict__ q, int n_q, signed char* __restrict__ qi,
                                                             float2* __restrict__ qmeta) {
    const int b = blockIdx.x, lane = threadIdx.x;
    float v[6];
#pragma unroll
    for (int j = 0; j < 6; ++j) v[j] = b < n_q ? q[(size_t)b * EM + lane + 64 * j] : 0.f;
    rotate384_wave(v, lane);  // the shadow's basis (header of this file)
    float amax = 0.f;
#pragma unroll
    for (int j = 0; j < 6; ++j) amax = fmaxf(amax, fabsf(v[j]));
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) amax = fmaxf(amax, __shfl_xor(amax, o));
    const float sq = fmaxf(amax, 1e-20f) / 127.0f;
    float e2 = 0.f;
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        const float t = fminf(fmaxf(rintf(v[j] / sq), -127.f), 127.f);
        const float d = v[j] - sq * t;
        e2 = __builtin_fmaf(d, d, e2);
        qi[(size_t)b * EM + lane + 64 * j] = (signed char)(int)t;
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) e2 += __shfl_xor(e2, o);
    if (lane == 0) qmeta[b] = b < n_q ? float2{sq, 1.1f * 1.001f * sqrtf(e2) + 1e-9f} : float2{0.f, 0.f};
}

// DBG (timing experiments only, results are wrong): 1 = no DMA, 2 = no barriers, 4 = no threshold tests; of the tests only:
// 8 = no thresholds (nothing hits), 16 = no max slices, 32 = no compare / slow path
template <bool DENSE, int DBG = 0>
__global__ __launch_bounds__(256) void scan_i8_pipe_kernel(const unsigned char* __restrict__ xs, const float2* __restrict__ meta,
                                                          uint32_t n_rows, uint32_t first_tile, uint32_t tile_stride,
                                                          uint32_t n_tiles, const i32x4_t* __restrict__ qi,
                                                          const float2* __restrict__ qmeta, int n_q,
                                                          const float* __restrict__ tau, uint32_t* __restrict__ cnt,
                                                          uint2* __restrict__ cand, float* __restrict__ dense) {
    constexpr int NW = 4, PD = 8, DPW = 48 / NW;
    __shared__ __attribute__((aligned(16))) unsigned char img[3 * I8_TILE_BYTES];
    __shared__ __attribute__((aligned(16))) uint32_t stage[NW * I8_ECAP * I8_EDW];  // 15 KiB beside the 144-KiB ring
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t r = lane & 31, h = lane >> 5;
    const int qg0 = wave * 32 + (int)r, qg1 = (wave + NW) * 32 + (int)r;  // this lane's query in group 0 / 1
    const bool live0 = wave * 32 < n_q, live1 = (wave + NW) * 32 < n_q;    // wave-uniform

    // B operand: lane (h, query) holds k = 32s + 16h .. +15 of k-step s = 16-B chunk 2s + h of the query's image
    i32x4_t qf[2][12];
#pragma unroll
    for (int s = 0; s < 12; ++s) {
        qf[0][s] = qi[(size_t)qg0 * 24 + 2 * s + h];
        qf[1][s] = qi[(size_t)qg1 * 24 + 2 * s + h];
    }
    // per lane and group: s_q, K2, 1/s_q and the threshold lowered by K2 and by its own rounding allowance
    float sq_l[2], k2_l[2], c1[2], c2[2];  // c1 = taum / s_q, c2 = 1.000001 / s_q (thresholds: set_thr)
    bool pad[2];
#pragma unroll
    for (int g = 0; g < 2; ++g) {
        const int qi_ = g == 0 ? qg0 : qg1;
        const float2 qm = qmeta[qi_];
        sq_l[g] = qm.x;
        k2_l[g] = qm.y;
        pad[g] = DENSE || qi_ >= n_q;
        c1[g] = 3.0e38f;  // padding column: threshold +2e9
        c2[g] = 0.f;
        if (!pad[g]) {
            const float tk = tau[qi_] - qm.y;
            const float rsq = 1.0f / qm.x;
            c1[g] = (tk - fabsf(tk) * 1e-6f) * rsq;  // (tau = -inf: -inf, every row is a candidate)
            c2[g] = 1.000001f * rsq;
        }
    }

    const uint32_t src_off0 = (uint32_t)(DPW * wave) * 1024u + (uint32_t)lane * 16u;
    const uint32_t G = gridDim.x;
    const uint32_t n_units = (n_tiles - blockIdx.x + G - 1) / G;
    const uint32_t last = n_units - 1;
    auto unit_tile = [&](uint32_t t) { return first_tile + (blockIdx.x + t * G) * tile_stride; };
    auto unit_row0 = [&](uint32_t t) { return unit_tile(t) * I8_TILE_ROWS; };
    auto unit_slot0 = [&](uint32_t t) { return (blockIdx.x + t * G) * I8_TILE_ROWS; };
    constexpr int NGP = DPW / 4;
    auto dma = [&](uint32_t t, uint32_t image, const unsigned char* (&gp)[NGP]) __attribute__((always_inline)) {
        const unsigned char* base = xs + (size_t)unit_tile(t) * I8_TILE_BYTES + src_off0;
        unsigned char* dst = img + image * I8_TILE_BYTES + wave * (DPW * 1024);
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
    auto dma_setup = [&](uint32_t t, const unsigned char* (&gp)[NGP]) __attribute__((always_inline)) {
        const unsigned char* base = xs + (size_t)unit_tile(t) * I8_TILE_BYTES + src_off0;
#pragma unroll
        for (int j = 0; j < NGP; ++j) gp[j] = base + j * 4096;
    };
    auto dma_one = [&](auto i_c, uint32_t image, const unsigned char* (&gp)[NGP]) __attribute__((always_inline)) {
        constexpr int I = decltype(i_c)::value, J = I / 4, O = (I % 4) * 1024;
        unsigned char* dst = img + image * I8_TILE_BYTES + wave * (DPW * 1024) + J * 4096;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gp[J],
                                         (__attribute__((address_space(3))) void*)dst, 16, O, 2 /* nt */);
    };
    auto keep = [&](const unsigned char* (&gp)[NGP]) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < NGP; ++j) asm volatile("" ::"v"(gp[j]));
    };

    // ---- candidate staging, private to the wave, LANE-PARALLEL: a lane whose running maximum beats its threshold dumps
    // its 16 accumulators + {query, first row, threshold, s s_q, E + K2} as one 96-B entry (six predicated ds_write_b128,
    // slot = popcount of the hit lanes below it: no ballot per element, no scalar loop — a hit costs ~80 clk instead of
    // ~700, and every clock of it is a clock the other three waves wait at the next barrier); the entries are expanded
    // into (score, row) candidates when the stage is flushed, one entry per lane.
    uint32_t wpos = 0;  // wave-uniform fill of this wave's region
    auto flush_wave = [&]() __attribute__((always_inline)) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the asm stores below
        const uint32_t seg = blockIdx.x % BATCH_CAND_SEGS;  // (a query belongs to ONE wave of every workgroup)
        if ((uint32_t)lane < wpos) {
            const uint32_t* en = stage + (wave * I8_ECAP + lane) * I8_EDW;
            const uint32_t qidx = en[16], row0 = en[17];
            const int th = (int)en[18];
            const float gl = __builtin_bit_cast(float, en[19]), ek = __builtin_bit_cast(float, en[20]);
            // (rolled loops, accumulators read again below rather than kept: the kernel has no registers to spare, and what
            // hipcc parks in AGPRs around a slow path must never be an accumulator — see the MFMA asm in step())
            uint32_t hits = 0;
#pragma unroll 1
            for (int i = 0; i < 16; ++i) {
                const uint32_t row = row0 + (uint32_t)((i & 3) + 8 * (i >> 2));
                hits |= ((int)en[i] > th && row < n_rows) ? (1u << i) : 0u;
            }
            // ONE slot reservation per entry (all its hits belong to one query): a single round trip to the memory-side
            // atomic unit for the whole flush
            uint32_t slot = hits ? atomicAdd(&cnt[qidx * BATCH_CAND_SEGS + seg], (uint32_t)__popc(hits)) : 0u;
#pragma unroll 1
            for (int i = 0; i < 16; ++i) {
                if (hits & (1u << i)) {
                    if (slot < I8_SEG_CAP)
                        cand[(size_t)qidx * BATCH_CAP + seg * I8_SEG_CAP + slot] =
                            make_uint2(__builtin_bit_cast(uint32_t, __builtin_fmaf((float)(int)en[i], gl, ek)),
                                       row0 + (uint32_t)((i & 3) + 8 * (i >> 2)));
                    ++slot;
                }
            }
        }
        wpos = 0;
    };
    const uint32_t stage_base = (uint32_t)(size_t)(__attribute__((address_space(3))) uint32_t*)stage;
    // slow path of one query group: mxl = the lane's maximum over its 16 accumulators, thr_lane its integer threshold,
    // gl / ek its s * s_q and E + K2; row_base: first row of the sub-tile, q_first: query of lane 0
    auto stage_hits = [&](const i32x16_t& acc, int mxl, uint32_t row_base, uint32_t q_first, int thr_lane, float gl, float ek)
        __attribute__((always_inline)) {
        const bool hitl = mxl > thr_lane;
        const unsigned long long hm = __ballot(hitl);
        const uint32_t n = (uint32_t)__popcll(hm);
        if (n > I8_ECAP) {
            // a burst (tau = -inf, or every query of the group next to the same row): element by element, straight to the
            // candidate buffers
            const uint32_t lim = n_rows > row_base + 4 * h ? n_rows - row_base - 4 * h : 0u;
            const uint32_t seg = blockIdx.x % BATCH_CAND_SEGS;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int ae = acc[e];
                const uint32_t roff = (uint32_t)((e & 3) + 8 * (e >> 2));
                if (ae > thr_lane && roff < lim) {
                    const uint32_t qidx = q_first + r;
                    const float sc = __builtin_fmaf((float)ae, gl, ek);
                    const uint32_t slot = atomicAdd(&cnt[qidx * BATCH_CAND_SEGS + seg], 1u);
                    if (slot < I8_SEG_CAP)
                        cand[(size_t)qidx * BATCH_CAP + seg * I8_SEG_CAP + slot] =
                            make_uint2(__builtin_bit_cast(uint32_t, sc), row_base + 4 * h + roff);
                }
            }
            return;
        }
        if (wpos + n > I8_ECAP) {
            flush_wave();
            asm volatile("" ::: "memory");  // the flush's reads of the stage stay in front of the stores below
        }
        if (hitl) {
            const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(hm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)hm, 0u));
            const uint32_t pa = stage_base + (wave * I8_ECAP + wpos + rank) * (I8_EDW * 4u);
            // (sub-registers of the accumulator tuple and single registers: no copies, no extra live values)
            const i32x4_t a0 = __builtin_shufflevector(acc, acc, 0, 1, 2, 3), a1 = __builtin_shufflevector(acc, acc, 4, 5, 6, 7);
            const i32x4_t a2 = __builtin_shufflevector(acc, acc, 8, 9, 10, 11), a3 = __builtin_shufflevector(acc, acc, 12, 13, 14, 15);
            asm volatile(
                "ds_write_b128 %0, %1\n\tds_write_b128 %0, %2 offset:16\n\tds_write_b128 %0, %3 offset:32\n\t"
                "ds_write_b128 %0, %4 offset:48\n\tds_write_b32 %0, %5 offset:64\n\tds_write_b32 %0, %6 offset:68\n\t"
                "ds_write_b32 %0, %7 offset:72\n\tds_write_b32 %0, %8 offset:76\n\tds_write_b32 %0, %9 offset:80"
                :
                : "v"(pa), "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(q_first + r), "v"(row_base + 4 * h), "v"(thr_lane), "v"(gl),
                  "v"(ek));
        }
        wpos += n;
    };
    i32x16_t acc[2][2];
    int mx[2] = {0, 0}, thr[2] = {0x7fffffff, 0x7fffffff};
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[1][0][e] = acc[1][1][e] = 0;
    // {s, E} of the four sub-tiles: mu = the tile whose sub-tiles are under test, ml = the next one (in flight / landed)
    f32x4 mu0, mu1, ml0, ml1;
    // {1 / s, E} of sub-tile j
    auto sub_meta = [&](int j, float& rs_, float& e_) __attribute__((always_inline)) {
        rs_ = j == 0 ? mu0.x : j == 1 ? mu0.z : j == 2 ? mu1.x : mu1.z;
        e_ = j == 0 ? mu0.y : j == 1 ? mu0.w : j == 2 ? mu1.y : mu1.w;
    };
    // dense store of one finished sub-tile (accumulator set SET, sub-tile J of the tile in mu)
    auto tail_dense = [&](auto set_c, auto nl_c, int J, uint32_t row_base, uint32_t slot_base) __attribute__((always_inline)) {
        constexpr int SET = decltype(set_c)::value, NL = decltype(nl_c)::value;
        const uint32_t row0 = row_base + 4 * h;
        float rs_, e_;
        sub_meta(J, rs_, e_);
        const float s_ = __builtin_amdgcn_rcpf(rs_);
#pragma unroll
        for (int g = 0; g < NL; ++g) {
            const int qidx = g == 0 ? qg0 : qg1;
            const float gl = s_ * sq_l[g], ek = e_ + k2_l[g];
            if (qidx < n_q) {
#pragma unroll
                for (int e4 = 0; e4 < 4; ++e4) {
                    f32x4 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int ae = acc[SET][g][e4 * 4 + e];
                        o[e] = (row0 + e + 8 * e4) < n_rows ? __builtin_fmaf((float)ae, gl, ek) : NEG_INF;
                    }
                    *reinterpret_cast<f32x4*>(dense + (size_t)qidx * BATCH_CAP + slot_base + 4 * h + 8 * e4) = o;
                }
            }
        }
    };
    // integer thresholds of sub-tile J (of the tile in mu) for the NL live groups
    // thr = floor(((taum - 1.000001 E) / (s s_q)) - 2), both groups at once: two packed fmas (v_pk_fma_f32), then med3 +
    // floor-convert per group.  Padding columns carry c1 = 3e38, c2 = 0: their threshold saturates at +2e9 (never hit).
    typedef float f32x2_t __attribute__((ext_vector_type(2)));
    auto set_thr = [&](auto nl_c, int J) __attribute__((always_inline)) {
        constexpr int NL = decltype(nl_c)::value;
        float rs_, e_;
        sub_meta(J, rs_, e_);
        const f32x2_t u = __builtin_elementwise_fma(f32x2_t{-e_, -e_}, f32x2_t{c2[0], c2[1]}, f32x2_t{c1[0], c1[1]});
        const f32x2_t t2 = __builtin_elementwise_fma(u, f32x2_t{rs_, rs_}, f32x2_t{-2.0f, -2.0f});
#pragma unroll
        for (int g = 0; g < NL; ++g) {
            // (NaN -- 0 * inf on a sub-tile past the end of the index -- comes out as -2e9: everything is tested further,
            // and the rows do not exist)
            const float tf = __builtin_amdgcn_fmed3f(g == 0 ? t2.x : t2.y, -2.0e9f, 2.0e9f);
            int ti;
            asm volatile("v_cvt_flr_i32_f32 %0, %1" : "=v"(ti) : "v"(tf));
            thr[g] = ti;
        }
    };
    auto slow = [&](auto set_c, auto nl_c, int J, uint32_t row_base) __attribute__((always_inline)) {
        constexpr int SET = decltype(set_c)::value, NL = decltype(nl_c)::value;
        float rs_, e_;
        sub_meta(J, rs_, e_);
        const float s_ = __builtin_amdgcn_rcpf(rs_);
#pragma unroll
        for (int g = 0; g < NL; ++g)
            if (__any(mx[g] > thr[g]))
                stage_hits(acc[SET][g], mx[g], row_base, (uint32_t)((wave + NW * g) * 32), thr[g], s_ * sq_l[g], e_ + k2_l[g]);
    };

#pragma unroll
    for (int g = 0; g < 2; ++g) {
#pragma unroll
        for (int s = 0; s < 12; ++s) asm volatile("" ::"v"(qf[g][s]));
        asm volatile("" ::"v"(c1[g]), "v"(sq_l[g]), "v"(k2_l[g]), "v"(c2[g]));
    }
    const uint32_t o0 = (uint32_t)(size_t)(__attribute__((address_space(3))) unsigned char*)img;
    const f32x4* meta4 = reinterpret_cast<const f32x4*>(meta);  // 2 per tile

    // prologue: {s, E} of tile 0 (complete before the loop as far as hipcc can tell), tiles 0 and 1 on their way
    ml0 = meta4[(size_t)unit_tile(0) * 2];
    ml1 = meta4[(size_t)unit_tile(0) * 2 + 1];
    asm volatile("" : "+v"(ml0), "+v"(ml1));
    mu0 = ml0;
    mu1 = ml1;
    const unsigned char* gp0[NGP];
    const unsigned char* gp1[NGP];
    dma(0, 0, gp0);
    dma(last < 1u ? last : 1u, 1, gp1);
    asm volatile("s_waitcnt vmcnt(%0)\n\ts_waitcnt lgkmcnt(0)\n\ts_barrier" ::"n"(DPW) : "memory");
    keep(gp0);
    keep(gp1);
    i32x4_t a[PD];
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
    auto step = [&](auto nl_c, uint32_t rd, uint32_t nx, uint32_t wr) __attribute__((always_inline)) {
        constexpr int NL = decltype(nl_c)::value;
        const uint32_t ad = o0 + rd * I8_TILE_BYTES + (uint32_t)lane * 16u;
        const uint32_t adn = o0 + nx * I8_TILE_BYTES + (uint32_t)lane * 16u;
        const unsigned char* gp[NGP];
#pragma unroll
        for (int f = 0; f < 48; ++f) {
            const int sub = f / 12, s = f % 12, set = sub & 1;
            // the sub-tile whose test runs during this one: (t, sub - 1), or (t - 1, 3) while sub == 0
            const int J = sub == 0 ? 3 : sub - 1;
            if (NL > 0) {
                asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(PD - 1));  // fragment f has landed
                __builtin_amdgcn_sched_barrier(0);
                // asm MFMAs: accumulators and group 0's query fragments in VGPRs, group 1's in AGPRs read directly as SrcB.
                // The kernel must stay below 256 VGPRs: hipcc would park live values in AGPRs around the slow paths, and a
                // copy of an accumulator it cannot know to be an MFMA result in flight reads a stale register.
#define DAWN_I8_ZERO(D, A, B, CB) asm volatile("v_mfma_i32_32x32x32_i8 %0, %1, %2, 0" : "=&v"(D) : "v"(A), CB(B))
#define DAWN_I8_ACC(D, A, B, CB) asm volatile("v_mfma_i32_32x32x32_i8 %0, %1, %2, %0" : "+v"(D) : "v"(A), CB(B))
                if (s == 0) {
                    DAWN_I8_ZERO(acc[set][0], a[f % PD], qf[0][0], "v");
                    if (NL > 1) DAWN_I8_ZERO(acc[set][1], a[f % PD], qf[1][0], "a");
                } else {
                    DAWN_I8_ACC(acc[set][0], a[f % PD], qf[0][s], "v");
                    if (NL > 1) DAWN_I8_ACC(acc[set][1], a[f % PD], qf[1][s], "a");
                }
#undef DAWN_I8_ZERO
#undef DAWN_I8_ACC
                __builtin_amdgcn_sched_barrier(0);
                const bool have_prev = sub > 0 || t > 0;  // (sub is a constant: folds to `t > 0` or true)
                if (!DENSE && !(DBG & 4)) {
                    constexpr int FIRST = 2;
                    if (s == FIRST - 1 && !(DBG & 8)) set_thr(nl_c, J);
                    if (s == FIRST) asm volatile("s_nop 7");
                    if (s >= FIRST && s < FIRST + 8 && !(DBG & 16)) {
                        const int j = s - FIRST;
#pragma unroll
                        for (int g = 0; g < NL; ++g) {
                            mx[g] = j == 0 ? max(acc[1 - set][g][0], acc[1 - set][g][1])
                                           : max(max(mx[g], acc[1 - set][g][2 * j]), acc[1 - set][g][2 * j + 1]);
                            asm volatile("" : "+v"(mx[g]));
                        }
                    }
                    if (s == FIRST + 8 && have_prev && !(DBG & 32)) {
                        bool hit = mx[0] > thr[0];
                        if (NL > 1) hit = hit || mx[1] > thr[1];
                        if (__builtin_expect(__any(hit), 0)) {  // (unlikely: keeps the slow paths out of the hot instruction stream)
                            const uint32_t rb = sub == 0 ? unit_row0(t - 1) + 96 : unit_row0(t) + 32 * (sub - 1);
                            if (set == 0) slow(C1(), nl_c, J, rb);
                            else slow(C0(), nl_c, J, rb);
                        }
                    }
                }
                if (DENSE && s == 2 && have_prev) {
                    asm volatile("s_nop 7");
                    const uint32_t rb = sub == 0 ? unit_row0(t - 1) + 96 : unit_row0(t) + 32 * (sub - 1);
                    const uint32_t sb = sub == 0 ? unit_slot0(t - 1) + 96 : unit_slot0(t) + 32 * (sub - 1);
                    if (set == 0) tail_dense(C1(), nl_c, J, rb, sb);
                    else tail_dense(C0(), nl_c, J, rb, sb);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            if (f == 11) {  // the last test of tile t-1 is done: its {s, E} make room for tile t's (landed since P2 of t-1)
                mu0 = ml0;
                mu1 = ml1;
                asm volatile("" : "+v"(mu0), "+v"(mu1));
            }
            if (f == 12) {  // P1: every wave has left tile t-1, its image may be overwritten
                if (!(DBG & 2)) asm volatile("s_barrier");
                __builtin_amdgcn_sched_barrier(0);
                // {s, E} of tile t+1, IN FRONT of the DMA of tile t+2: P2's vmcnt(DPW) covers them
                const f32x4* mp = meta4 + (size_t)unit_tile(t + 1 < n_units ? t + 1 : last) * 2;
                asm volatile("global_load_dwordx4 %0, %2, off\n\tglobal_load_dwordx4 %1, %2, off offset:16"
                             : "=&v"(ml0), "=&v"(ml1)
                             : "v"(mp));
                dma_setup(t + 2 < n_units ? t + 2 : last, gp);
            }
            if (!(DBG & 1)) {
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
            if (f == 39) {  // P2: tile t+1 and the {s, E} loads in front of tile t+2's DMA have landed
                __builtin_amdgcn_sched_barrier(0);
                if (DENSE) asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier");  // (dense stores share vmcnt)
                else if (DBG & 1) asm volatile("s_waitcnt vmcnt(0)");
                else if (DBG & 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DPW));
                else asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(DPW));
                asm volatile("" : "+v"(ml0), "+v"(ml1));
            }
            if (NL > 0) {
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
        constexpr int NL = decltype(nl_c)::value;
        uint32_t rd = 0, nx = 1, wr = 2;
        while (t < n_units) {
            step(nl_c, rd, nx, wr);
            const uint32_t o = rd;
            rd = nx;
            nx = wr;
            wr = o;
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_nop 15\n\ts_nop 15");  // the ring's look-ahead reads; the last MFMAs
        __builtin_amdgcn_sched_barrier(0);
        if (NL > 0) {  // the last sub-tile (t = last, sub 3, accumulator set 1); mu holds tile `last` since its f == 11
            const uint32_t rb = unit_row0(last) + 96, sb = unit_slot0(last) + 96;
            if (DENSE) {
                tail_dense(C1(), nl_c, 3, rb, sb);
            } else if (!(DBG & 4)) {
                set_thr(nl_c, 3);
#pragma unroll
                for (int g = 0; g < NL; ++g) {
                    mx[g] = acc[1][g][0];
#pragma unroll
                    for (int e = 1; e < 16; ++e) mx[g] = max(mx[g], acc[1][g][e]);
                }
                slow(C1(), nl_c, 3, rb);
            }
        }
    };
    if (live1) run(C2());
    else if (live0) run(C1());
    else run(C0());
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the clamped tail DMAs must not outlive the workgroup's LDS
    if (!DENSE) flush_wave();
}

// ------------------------------------------------------------------------------------------------
// The same pipeline on v_mfma_i32_16x16x64_i8.  With every CU busy the chip holds a higher clock on the small shape:
// operands in registers, pseudo-random data, 256 CUs: 32x32x32 3.38 Pop/s, 16x16x64 3.98 Pop/s (+18 %; on 32 CUs, no power
// limit, both run at the 4.97 Pop/s peak — tools/probes/mfma_i8_shapes.hip).  Same images, same DMA, same barriers, same
// fragment ring; what changes:
//   a wave's 64 queries are FOUR groups of 16 (B operand: lane (query n = lane & 15, q4 = lane >> 4) holds the 16 bytes
//   k = 64 s + 16 q4 ..+15 of k-step s = 0..5 — chunk 4 s + q4 of the query's image);
//   a fragment step f = (sub16 = f / 6, s = f % 6) is FOUR MFMAs (one per group) on 16 rows x 64 k.  The A operand — lane
//   (row r16 = lane & 15, q4) wants k = 64 s + 16 q4 ..+15 of row 16 sub16 + r16 — is read from the unchanged image (built
//   for the 32x32x32 operand: 1-KiB fragment (sub, s32) = lane (r, h) -> 16 bytes k = 32 s32 + 16 h of row 32 sub + r) at
//   byte ((sub16 / 2) * 12 + 2 s) * 1024 + 256 (sub16 & 1) + 512 q4 + 16 r16: per lane group still 16 distinct 16-B slots;
//   the accumulators of a (16-row, 16-query) tile are 4 registers per lane (rows 4 q4 ..+3 of the lane's query): a lane's
//   test is a max over 4 values, a staged hit entry is 4 accumulators + 5 words (48 B, one entry per lane at the flush).
// ------------------------------------------------------------------------------------------------
constexpr uint32_t I16_ECAP = 64;  // staged hit entries per wave (the flush gives an entry to a lane) ...
constexpr uint32_t I16_EDW = 12;   // ... of 12 dwords: 4 accumulators, query, first row, threshold, s s_q, E + K2, 3 unused

template <bool DENSE>
__global__ __launch_bounds__(256) void scan_i8_pipe16_kernel(const unsigned char* __restrict__ xs, const float2* __restrict__ meta,
                                                            uint32_t n_rows, uint32_t first_tile, uint32_t tile_stride,
                                                            uint32_t n_tiles, const i32x4_t* __restrict__ qi,
                                                            const float2* __restrict__ qmeta, int n_q,
                                                            const float* __restrict__ tau, uint32_t* __restrict__ cnt,
                                                            uint2* __restrict__ cand, float* __restrict__ dense,
                                                            uint32_t* __restrict__ go) {
    if (go != nullptr && go[0] == 0u) return;  // (launch_i8_rerun: no query of the batch is flagged — grid-uniform)
    constexpr int NW = 4, PD = 8, DPW = 48 / NW;
    __shared__ __attribute__((aligned(16))) unsigned char img[3 * I8_TILE_BYTES];
    __shared__ __attribute__((aligned(16))) uint32_t stage[NW * I16_ECAP * I16_EDW];  // 12 KiB beside the 144-KiB ring
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t r16 = lane & 15, q4 = lane >> 4;
    // group g of this wave: queries 32 wave + 16 g ..+15 (g = 0, 1) and 128 + 32 wave + 16 (g - 2) ..+15 (g = 2, 3): a query
    // belongs to the same wave as in scan_i8_pipe_kernel
    auto group_first = [&](int g) { return (g < 2 ? wave * 32 + 16 * g : (wave + NW) * 32 + 16 * (g - 2)); };
    const bool live0 = wave * 32 < n_q, live1 = (wave + NW) * 32 < n_q;  // wave-uniform: groups {0, 1} / {2, 3}

    i32x4_t qf[4][6];
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int s = 0; s < 6; ++s) qf[g][s] = qi[(size_t)(group_first(g) + (int)r16) * 24 + 4 * s + q4];
    float sq_l[4], k2_l[4], c1[4], c2[4];  // (as in scan_i8_pipe_kernel)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int qi_ = group_first(g) + (int)r16;
        const float2 qm = qmeta[qi_];
        sq_l[g] = qm.x;
        k2_l[g] = qm.y;
        c1[g] = 3.0e38f;  // padding column: threshold +2e9
        c2[g] = 0.f;
        if (!(DENSE || qi_ >= n_q)) {
            const float tk = tau[qi_] - qm.y;
            const float rsq = 1.0f / qm.x;
            c1[g] = (tk - fabsf(tk) * 1e-6f) * rsq;  // (tau = -inf: -inf, every row is a candidate)
            c2[g] = 1.000001f * rsq;
        }
    }

    const uint32_t src_off0 = (uint32_t)(DPW * wave) * 1024u + (uint32_t)lane * 16u;
    const uint32_t G = gridDim.x;
    const uint32_t n_units = (n_tiles - blockIdx.x + G - 1) / G;
    const uint32_t last = n_units - 1;
    auto unit_tile = [&](uint32_t t) { return first_tile + (blockIdx.x + t * G) * tile_stride; };
    auto unit_row0 = [&](uint32_t t) { return unit_tile(t) * I8_TILE_ROWS; };
    auto unit_slot0 = [&](uint32_t t) { return (blockIdx.x + t * G) * I8_TILE_ROWS; };
    constexpr int NGP = DPW / 4;
    auto dma = [&](uint32_t t, uint32_t image, const unsigned char* (&gp)[NGP]) __attribute__((always_inline)) {
        const unsigned char* base = xs + (size_t)unit_tile(t) * I8_TILE_BYTES + src_off0;
        unsigned char* dst = img + image * I8_TILE_BYTES + wave * (DPW * 1024);
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
    auto dma_setup = [&](uint32_t t, const unsigned char* (&gp)[NGP]) __attribute__((always_inline)) {
        const unsigned char* base = xs + (size_t)unit_tile(t) * I8_TILE_BYTES + src_off0;
#pragma unroll
        for (int j = 0; j < NGP; ++j) gp[j] = base + j * 4096;
    };
    auto dma_one = [&](auto i_c, uint32_t image, const unsigned char* (&gp)[NGP]) __attribute__((always_inline)) {
        constexpr int I = decltype(i_c)::value, J = I / 4, O = (I % 4) * 1024;
        unsigned char* dst = img + image * I8_TILE_BYTES + wave * (DPW * 1024) + J * 4096;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gp[J],
                                         (__attribute__((address_space(3))) void*)dst, 16, O, 2 /* nt */);
    };
    auto keep = [&](const unsigned char* (&gp)[NGP]) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < NGP; ++j) asm volatile("" ::"v"(gp[j]));
    };

    // ---- candidate staging, private to the wave, lane-parallel (see scan_i8_pipe_kernel): entry = 4 accumulators + {query,
    // first row, threshold, s s_q, E + K2}
    uint32_t wpos = 0;  // wave-uniform fill of this wave's region
    auto flush_wave = [&]() __attribute__((always_inline)) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the asm stores below
        const uint32_t seg = blockIdx.x % BATCH_CAND_SEGS;
        if ((uint32_t)lane < wpos) {
            const uint32_t* en = stage + (wave * I16_ECAP + lane) * I16_EDW;
            const uint32_t qidx = en[4], row0 = en[5];
            const int th = (int)en[6];
            const float gl = __builtin_bit_cast(float, en[7]), ek = __builtin_bit_cast(float, en[8]);
            uint32_t hits = 0;
#pragma unroll
            for (int i = 0; i < 4; ++i) hits |= ((int)en[i] > th && row0 + (uint32_t)i < n_rows) ? (1u << i) : 0u;
            // (second pass, launch_i8_rerun: a segment that has overflowed stays overflowed — its query goes to the bounded pass
            // whatever else is appended, and a query inside a shell of a million near-ties would otherwise send a million atomics to
            // its sixteen counters, ~8 M/s per address: 0.5 s per batch before this test.  Not in the first pass: the extra load in
            // front of every atomic cost 1 M rows x 256 queries 0.150 -> 0.204 ms.)
            if (go != nullptr && hits && cnt[qidx * BATCH_CAND_SEGS + seg] > I8_SEG_CAP) {
                hits = 0;
                go[4 + qidx] = 1u;  // (the query is lost to the bounded pass: every wave may stop looking)
            }
            uint32_t slot = hits ? atomicAdd(&cnt[qidx * BATCH_CAND_SEGS + seg], (uint32_t)__popc(hits)) : 0u;
#pragma unroll 1
            for (int i = 0; i < 4; ++i) {
                if (hits & (1u << i)) {
                    if (slot < I8_SEG_CAP)
                        cand[(size_t)qidx * BATCH_CAP + seg * I8_SEG_CAP + slot] =
                            make_uint2(__builtin_bit_cast(uint32_t, __builtin_fmaf((float)(int)en[i], gl, ek)), row0 + (uint32_t)i);
                    ++slot;
                }
            }
        }
        wpos = 0;
        if (go != nullptr) {
            // second pass (launch_i8_rerun): a query one of whose segments has overflowed gets the padding columns' threshold in this
            // wave from here on — inside a shell of a million near-ties every tile would otherwise stage hits for it to the end
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int qi_ = group_first(g) + (int)r16;
                if (qi_ < n_q && __atomic_load_n(&go[4 + qi_], __ATOMIC_RELAXED) != 0u) c1[g] = 3.0e38f;
            }
        }
    };
    const uint32_t stage_base = (uint32_t)(size_t)(__attribute__((address_space(3))) uint32_t*)stage;
    // slow path of one query group: mxl = the lane's maximum over its 4 accumulators, thr_lane its integer threshold,
    // gl / ek its s * s_q and E + K2; row_base: first row of the 16-row tile, q_first: query of lane 0
    auto stage_hits = [&](const i32x4_t& acc, int mxl, uint32_t row_base, uint32_t q_first, int thr_lane, float gl, float ek)
        __attribute__((always_inline)) {
        const bool hitl = mxl > thr_lane;
        const unsigned long long hm = __ballot(hitl);
        const uint32_t n = (uint32_t)__popcll(hm);
        if (wpos + n > I16_ECAP) {
            flush_wave();
            asm volatile("" ::: "memory");  // the flush's reads of the stage stay in front of the stores below
        }
        if (hitl) {
            const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(hm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)hm, 0u));
            const uint32_t pa = stage_base + (wave * I16_ECAP + wpos + rank) * (I16_EDW * 4u);
            asm volatile(
                "ds_write_b128 %0, %1\n\tds_write_b32 %0, %2 offset:16\n\tds_write_b32 %0, %3 offset:20\n\t"
                "ds_write_b32 %0, %4 offset:24\n\tds_write_b32 %0, %5 offset:28\n\tds_write_b32 %0, %6 offset:32"
                :
                : "v"(pa), "v"(acc), "v"(q_first + r16), "v"(row_base + 4 * q4), "v"(thr_lane), "v"(gl), "v"(ek));
        }
        wpos += n;  // (n <= 64 = I16_ECAP: after a flush the entries always fit)
    };
    i32x4_t acc[2][4];
    int mx[4] = {0, 0, 0, 0}, thr[4] = {0x7fffffff, 0x7fffffff, 0x7fffffff, 0x7fffffff};
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[1][g][e] = 0;
    // {s, E} of the four 32-row sub-tiles: mu = the tile whose sub-tiles are under test, ml = the next one
    f32x4 mu0, mu1, ml0, ml1;
    auto sub_meta = [&](int j, float& rs_, float& e_) __attribute__((always_inline)) {  // {1 / s, E} of 32-row sub-tile j
        rs_ = j == 0 ? mu0.x : j == 1 ? mu0.z : j == 2 ? mu1.x : mu1.z;
        e_ = j == 0 ? mu0.y : j == 1 ? mu0.w : j == 2 ? mu1.y : mu1.w;
    };
    // dense store of one finished 16-row tile (accumulator set SET, 16-row tile J16 of the tile in mu)
    auto tail_dense = [&](auto set_c, auto nl_c, int J16, uint32_t row_base, uint32_t slot_base) __attribute__((always_inline)) {
        constexpr int SET = decltype(set_c)::value, NL = decltype(nl_c)::value;
        const uint32_t row0 = row_base + 4 * q4;
        float rs_, e_;
        sub_meta(J16 >> 1, rs_, e_);
        const float s_ = __builtin_amdgcn_rcpf(rs_);
#pragma unroll
        for (int g = 0; g < 2 * NL; ++g) {
            const int qidx = group_first(g) + (int)r16;
            const float gl = s_ * sq_l[g], ek = e_ + k2_l[g];
            if (qidx < n_q) {
                f32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = (row0 + e) < n_rows ? __builtin_fmaf((float)acc[SET][g][e], gl, ek) : NEG_INF;
                *reinterpret_cast<f32x4*>(dense + (size_t)qidx * BATCH_CAP + slot_base + 4 * q4) = o;
            }
        }
    };
    // integer thresholds of 32-row sub-tile J for the live groups (see scan_i8_pipe_kernel::set_thr), two groups per packed fma
    typedef float f32x2_t __attribute__((ext_vector_type(2)));
    auto set_thr = [&](auto nl_c, int J) __attribute__((always_inline)) {
        constexpr int NL = decltype(nl_c)::value;
        float rs_, e_;
        sub_meta(J, rs_, e_);
#pragma unroll
        for (int p = 0; p < NL; ++p) {
            const f32x2_t u = __builtin_elementwise_fma(f32x2_t{-e_, -e_}, f32x2_t{c2[2 * p], c2[2 * p + 1]}, f32x2_t{c1[2 * p], c1[2 * p + 1]});
            const f32x2_t t2 = __builtin_elementwise_fma(u, f32x2_t{rs_, rs_}, f32x2_t{-2.0f, -2.0f});
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const float tf = __builtin_amdgcn_fmed3f(i == 0 ? t2.x : t2.y, -2.0e9f, 2.0e9f);
                int ti;
                asm volatile("v_cvt_flr_i32_f32 %0, %1" : "=v"(ti) : "v"(tf));
                thr[2 * p + i] = ti;
            }
        }
    };
    auto slow = [&](auto set_c, auto nl_c, int J16, uint32_t row_base) __attribute__((always_inline)) {
        constexpr int SET = decltype(set_c)::value, NL = decltype(nl_c)::value;
        float rs_, e_;
        sub_meta(J16 >> 1, rs_, e_);
        const float s_ = __builtin_amdgcn_rcpf(rs_);
#pragma unroll
        for (int g = 0; g < 2 * NL; ++g)
            if (__any(mx[g] > thr[g]))
                stage_hits(acc[SET][g], mx[g], row_base, (uint32_t)group_first(g), thr[g], s_ * sq_l[g], e_ + k2_l[g]);
    };

#pragma unroll
    for (int g = 0; g < 4; ++g) {
#pragma unroll
        for (int s = 0; s < 6; ++s) asm volatile("" ::"v"(qf[g][s]));
        asm volatile("" ::"v"(c1[g]), "v"(sq_l[g]), "v"(k2_l[g]), "v"(c2[g]));
    }
    const uint32_t o0 = (uint32_t)(size_t)(__attribute__((address_space(3))) unsigned char*)img;
    const f32x4* meta4 = reinterpret_cast<const f32x4*>(meta);  // 2 per tile

    ml0 = meta4[(size_t)unit_tile(0) * 2];
    ml1 = meta4[(size_t)unit_tile(0) * 2 + 1];
    asm volatile("" : "+v"(ml0), "+v"(ml1));
    mu0 = ml0;
    mu1 = ml1;
    const unsigned char* gp0[NGP];
    const unsigned char* gp1[NGP];
    dma(0, 0, gp0);
    dma(last < 1u ? last : 1u, 1, gp1);
    asm volatile("s_waitcnt vmcnt(%0)\n\ts_waitcnt lgkmcnt(0)\n\ts_barrier" ::"n"(DPW) : "memory");
    keep(gp0);
    keep(gp1);
    // byte offset of this lane's 16 bytes of fragment step f within a tile image
    const uint32_t lane_off = q4 * 512u + r16 * 16u;
#define DAWN_I16_FOFF(F) ((((F) / 12) * 12 + 2 * ((F) % 6)) * 1024 + 256 * (((F) / 6) & 1))
    i32x4_t a[PD];
    {
        const uint32_t ad = o0 + lane_off;
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(a[0]) : "v"(ad), "n"(DAWN_I16_FOFF(0)));
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(a[1]) : "v"(ad), "n"(DAWN_I16_FOFF(1)));
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(a[2]) : "v"(ad), "n"(DAWN_I16_FOFF(2)));
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(a[3]) : "v"(ad), "n"(DAWN_I16_FOFF(3)));
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(a[4]) : "v"(ad), "n"(DAWN_I16_FOFF(4)));
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(a[5]) : "v"(ad), "n"(DAWN_I16_FOFF(5)));
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(a[6]) : "v"(ad), "n"(DAWN_I16_FOFF(6)));
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(a[7]) : "v"(ad), "n"(DAWN_I16_FOFF(7)));
    }

    uint32_t t = 0;
    typedef std::integral_constant<int, 0> C0;
    typedef std::integral_constant<int, 1> C1;
    typedef std::integral_constant<int, 2> C2;
    auto step = [&](auto nl_c, uint32_t rd, uint32_t nx, uint32_t wr) __attribute__((always_inline)) {
        constexpr int NL = decltype(nl_c)::value;  // live group PAIRS
        const uint32_t ad = o0 + rd * I8_TILE_BYTES + lane_off;
        const uint32_t adn = o0 + nx * I8_TILE_BYTES + lane_off;
        const unsigned char* gp[NGP];
#pragma unroll
        for (int f = 0; f < 48; ++f) {
            const int sub16 = f / 6, s = f % 6, set = sub16 & 1;
            // the 16-row tile whose test runs during this one: (t, sub16 - 1), or (t - 1, 7) while sub16 == 0
            const int J16 = sub16 == 0 ? 7 : sub16 - 1;
            if (NL > 0) {
                asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(PD - 1));  // fragment f has landed
                __builtin_amdgcn_sched_barrier(0);
                // asm MFMAs (see scan_i8_pipe_kernel): accumulators and the query fragments of groups 0, 1 in VGPRs, those of
                // groups 2, 3 in AGPRs read directly as SrcB
#define DAWN_I16_ZERO(D, A, B, CB) asm volatile("v_mfma_i32_16x16x64_i8 %0, %1, %2, 0" : "=&v"(D) : "v"(A), CB(B))
#define DAWN_I16_ACC(D, A, B, CB) asm volatile("v_mfma_i32_16x16x64_i8 %0, %1, %2, %0" : "+v"(D) : "v"(A), CB(B))
                if (s == 0) {
                    DAWN_I16_ZERO(acc[set][0], a[f % PD], qf[0][0], "v");
                    DAWN_I16_ZERO(acc[set][1], a[f % PD], qf[1][0], "v");
                    if (NL > 1) {
                        DAWN_I16_ZERO(acc[set][2], a[f % PD], qf[2][0], "a");
                        DAWN_I16_ZERO(acc[set][3], a[f % PD], qf[3][0], "a");
                    }
                } else {
                    DAWN_I16_ACC(acc[set][0], a[f % PD], qf[0][s], "v");
                    DAWN_I16_ACC(acc[set][1], a[f % PD], qf[1][s], "v");
                    if (NL > 1) {
                        DAWN_I16_ACC(acc[set][2], a[f % PD], qf[2][s], "a");
                        DAWN_I16_ACC(acc[set][3], a[f % PD], qf[3][s], "a");
                    }
                }
#undef DAWN_I16_ZERO
#undef DAWN_I16_ACC
                __builtin_amdgcn_sched_barrier(0);
                const bool have_prev = sub16 > 0 || t > 0;  // (sub16 is a constant: folds to `t > 0` or true)
                if (!DENSE) {
                    // thresholds belong to a 32-row sub-tile: computed for its first 16-row tile, kept for the second.  (No s_nop in
                    // front of the slices below: the accumulators they read were last written eight MFMAs ago.)
                    if (s == 0 && (J16 & 1) == 0) set_thr(nl_c, J16 >> 1);
                    if (s == 1 || s == 2) {
#pragma unroll
                        for (int g = 2 * (s - 1); g < 2 * (s - 1) + 2; ++g) {
                            if (g < 2 * NL) {
                                mx[g] = max(max(acc[1 - set][g][0], acc[1 - set][g][1]), max(acc[1 - set][g][2], acc[1 - set][g][3]));
                                asm volatile("" : "+v"(mx[g]));
                            }
                        }
                    }
                    if (s == 3 && have_prev) {
                        bool hit = mx[0] > thr[0] || mx[1] > thr[1];
                        if (NL > 1) hit = hit || mx[2] > thr[2] || mx[3] > thr[3];
                        if (__builtin_expect(__any(hit), 0)) {  // (unlikely: keeps the slow paths out of the hot instruction stream)
                            const uint32_t rb = sub16 == 0 ? unit_row0(t - 1) + 112 : unit_row0(t) + 16 * (sub16 - 1);
                            if (set == 0) slow(C1(), nl_c, J16, rb);
                            else slow(C0(), nl_c, J16, rb);
                        }
                    }
                }
                if (DENSE && s == 1 && have_prev) {
                    asm volatile("s_nop 7");
                    const uint32_t rb = sub16 == 0 ? unit_row0(t - 1) + 112 : unit_row0(t) + 16 * (sub16 - 1);
                    const uint32_t sb = sub16 == 0 ? unit_slot0(t - 1) + 112 : unit_slot0(t) + 16 * (sub16 - 1);
                    if (set == 0) tail_dense(C1(), nl_c, J16, rb, sb);
                    else tail_dense(C0(), nl_c, J16, rb, sb);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            if (f == 5) {  // the last test of tile t-1 is done: its {s, E} make room for tile t's (landed since P2 of t-1)
                mu0 = ml0;
                mu1 = ml1;
                asm volatile("" : "+v"(mu0), "+v"(mu1));
            }
            if (f == 12) {  // P1: every wave has left tile t-1, its image may be overwritten
                asm volatile("s_barrier");
                __builtin_amdgcn_sched_barrier(0);
                // {s, E} of tile t+1, IN FRONT of the DMA of tile t+2: P2's vmcnt(DPW) covers them
                const f32x4* mp = meta4 + (size_t)unit_tile(t + 1 < n_units ? t + 1 : last) * 2;
                asm volatile("global_load_dwordx4 %0, %2, off\n\tglobal_load_dwordx4 %1, %2, off offset:16"
                             : "=&v"(ml0), "=&v"(ml1)
                             : "v"(mp));
                dma_setup(t + 2 < n_units ? t + 2 : last, gp);
            }
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
            if (f == 39) {  // P2: tile t+1 and the {s, E} loads in front of tile t+2's DMA have landed
                __builtin_amdgcn_sched_barrier(0);
                if (DENSE) asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier");  // (dense stores share vmcnt)
                else asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(DPW));
                asm volatile("" : "+v"(ml0), "+v"(ml1));
            }
            if (NL > 0) {
                __builtin_amdgcn_sched_barrier(0);
                if (f + PD < 48)
                    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(a[f % PD]) : "v"(ad), "n"(DAWN_I16_FOFF(f + PD)));
                else
                    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(a[f % PD]) : "v"(adn), "n"(DAWN_I16_FOFF(f + PD < 48 ? 0 : f + PD - 48)));
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        keep(gp);
        ++t;
    };
    auto run = [&](auto nl_c) __attribute__((always_inline)) {
        constexpr int NL = decltype(nl_c)::value;
        uint32_t rd = 0, nx = 1, wr = 2;
        while (t < n_units) {
            step(nl_c, rd, nx, wr);
            const uint32_t o = rd;
            rd = nx;
            nx = wr;
            wr = o;
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_nop 15\n\ts_nop 15");  // the ring's look-ahead reads; the last MFMAs
        __builtin_amdgcn_sched_barrier(0);
        if (NL > 0) {  // the last 16-row tile (t = last, sub16 7, accumulator set 1); mu holds tile `last` since its f == 5
            const uint32_t rb = unit_row0(last) + 112, sb = unit_slot0(last) + 112;
            if (DENSE) {
                tail_dense(C1(), nl_c, 7, rb, sb);
            } else {
                set_thr(nl_c, 3);
#pragma unroll
                for (int g = 0; g < 2 * NL; ++g) mx[g] = max(max(acc[1][g][0], acc[1][g][1]), max(acc[1][g][2], acc[1][g][3]));
                slow(C1(), nl_c, 7, rb);
            }
        }
    };
    if (live1) run(C2());
    else if (live0) run(C1());
    else run(C0());
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the clamped tail DMAs must not outlive the workgroup's LDS
    if (!DENSE) flush_wave();
#ifdef DAWN_EXPERIMENTS
    if (lane == 0 && blockIdx.x < 256) dawn_ts_pass[blockIdx.x * 4 + wave] = __builtin_amdgcn_s_memrealtime();  // tools/pass_ts.py
#endif
#undef DAWN_I16_FOFF
}

// append pass; mfma_sched 41 / 42 / 44 / 47: timing experiments with parts switched off (results are wrong)
static void launch_i8_append(const unsigned char* xs, const float2* mt, uint32_t n_rows, uint32_t stride, uint32_t n_tiles,
                             const i32x4_t* qi, const float2* qm, int B, const BatchWorkspace& ws, uint32_t blocks,
                             hipStream_t stream, uint32_t* go = nullptr) {
#define DAWN_I8_PIPE(DBG_)                                                                                                   \
    hipLaunchKernelGGL((scan_i8_pipe_kernel<false, DBG_>), dim3(blocks), dim3(256), 0, stream, xs, mt, n_rows, 0u, stride, n_tiles, \
                       qi, qm, B, ws.tau, ws.cnt, reinterpret_cast<uint2*>(ws.cand), reinterpret_cast<float*>(ws.cand))
    switch (ws.sched) {
#ifdef DAWN_EXPERIMENTS  // (make EXPERIMENTS=1: not in the release library)
        case 41: DAWN_I8_PIPE(1); break;
        case 42: DAWN_I8_PIPE(2); break;
        case 44: DAWN_I8_PIPE(4); break;
        case 47: DAWN_I8_PIPE(7); break;
        case 48: DAWN_I8_PIPE(8); break;    // 48 / 49 / 50: the threshold test without its thresholds / slices / branch
        case 49: DAWN_I8_PIPE(16); break;
        case 50: DAWN_I8_PIPE(32); break;
#endif
        case 32: DAWN_I8_PIPE(0); break;  // the 32x32x32 form (option "mfma_sched" = 32)
        default:
            hipLaunchKernelGGL(scan_i8_pipe16_kernel<false>, dim3(blocks), dim3(256), 0, stream, xs, mt, n_rows, 0u, stride, n_tiles,
                               qi, qm, B, ws.tau, ws.cnt, reinterpret_cast<uint2*>(ws.cand), reinterpret_cast<float*>(ws.cand), go);
            break;
    }
#undef DAWN_I8_PIPE
}

// dense pass (every score of the tiles it visits)
static void launch_i8_dense(const unsigned char* xs, const float2* mt, uint32_t n_rows, uint32_t stride, uint32_t n_tiles,
                            const i32x4_t* qi, const float2* qm, int B, const BatchWorkspace& ws, uint32_t blocks,
                            hipStream_t stream) {
    if (ws.sched == 32)
        hipLaunchKernelGGL(scan_i8_pipe_kernel<true>, dim3(blocks), dim3(256), 0, stream, xs, mt, n_rows, 0u, stride, n_tiles, qi, qm,
                           B, ws.tau, ws.cnt, reinterpret_cast<uint2*>(ws.cand), reinterpret_cast<float*>(ws.cand));
    else
        hipLaunchKernelGGL(scan_i8_pipe16_kernel<true>, dim3(blocks), dim3(256), 0, stream, xs, mt, n_rows, 0u, stride, n_tiles, qi,
                           qm, B, ws.tau, ws.cnt, reinterpret_cast<uint2*>(ws.cand), reinterpret_cast<float*>(ws.cand), nullptr);
}

// Timing hook: the full append pass alone (thresholds ws.tau and query images as left by the last search), `iters` times
void launch_batched_full_pass_i8(const void* d_i8, const void* d_meta, uint32_t n_rows, int B, const BatchWorkspace& ws, int grid,
                                 int iters, hipStream_t stream, hipEvent_t ev0, hipEvent_t ev1) {
    const BatchPlan pl = plan_batched_tiles(n_rows, I8_TILE_ROWS, ws.target, 10);
    const signed char* qi = reinterpret_cast<const signed char*>(ws.qh);
    const float2* qm = reinterpret_cast<const float2*>(qi + (size_t)BATCH_QT * EM);
    const uint32_t blocks = pl.n_tiles_total < (uint32_t)grid ? pl.n_tiles_total : (uint32_t)grid;
    (void)hipEventRecord(ev0, stream);
    for (int i = 0; i < iters; ++i) {
        (void)hipMemsetAsync(ws.cnt, 0, BATCH_QT * BATCH_CAND_SEGS * sizeof(uint32_t), stream);
        launch_i8_append(reinterpret_cast<const unsigned char*>(d_i8), reinterpret_cast<const float2*>(d_meta), n_rows, 1u,
                         pl.n_tiles_total, reinterpret_cast<const i32x4_t*>(qi), qm, B, ws, blocks, stream);
    }
    (void)hipEventRecord(ev1, stream);
}

// Test hook: dense upper-bound scores of rows [0, min(n_rows, BATCH_CAP)) -> ws.cand viewed as float [BATCH_QT][BATCH_CAP]
void launch_batched_dense_scores_i8(const void* d_i8, const void* d_meta, uint32_t n_rows, const float* d_q, int B,
                                    const BatchWorkspace& ws, int grid, hipStream_t stream) {
    signed char* qi = reinterpret_cast<signed char*>(ws.qh);
    float2* qm = reinterpret_cast<float2*>(qi + (size_t)BATCH_QT * EM);
    hipLaunchKernelGGL(prep_queries_i8_kernel, dim3(BATCH_QT), dim3(64), 0, stream, d_q, B, qi, qm);
    const uint32_t n = n_rows < (uint32_t)BATCH_CAP ? n_rows : (uint32_t)BATCH_CAP;
    const uint32_t n_tiles = (n + I8_TILE_ROWS - 1) / I8_TILE_ROWS;
    if (n_tiles == 0) return;
    launch_i8_dense(reinterpret_cast<const unsigned char*>(d_i8), reinterpret_cast<const float2*>(d_meta), n_rows, 1u, n_tiles,
                    reinterpret_cast<const i32x4_t*>(qi), qm, B, ws, n_tiles < (uint32_t)grid ? n_tiles : (uint32_t)grid, stream);
}

// The sampling passes of the batched int8 search alone (scan_f6.hip runs its own full pass behind them): queries -> int8 images,
// ws.tau = the sampled thresholds (rank ~ws.target), segment counters left at zero.  n_rows > BATCH_CAP.
void launch_i8_sample_thresholds(const void* d_i8, const void* d_meta, uint32_t n_rows, const float* d_q, int B, uint32_t k,
                                 const BatchWorkspace& ws, int grid, hipStream_t stream) {
    const BatchPlan pl = plan_batched_tiles(n_rows, I8_TILE_ROWS, ws.target, k);
    signed char* qi = reinterpret_cast<signed char*>(ws.qh);
    float2* qm = reinterpret_cast<float2*>(qi + (size_t)BATCH_QT * EM);
    hipLaunchKernelGGL(prep_queries_i8_kernel, dim3(BATCH_QT), dim3(64), 0, stream, d_q, B, qi, qm);
    const unsigned char* xs = reinterpret_cast<const unsigned char*>(d_i8);
    const float2* mt = reinterpret_cast<const float2*>(d_meta);
    const uint32_t b1 = pl.s1_tiles < (uint32_t)grid ? pl.s1_tiles : (uint32_t)grid;
    launch_i8_dense(xs, mt, n_rows, pl.s1_stride, pl.s1_tiles, reinterpret_cast<const i32x4_t*>(qi), qm, B, ws, b1, stream);
    launch_tau_select(true, B, ws, pl.s1_tiles * I8_TILE_ROWS, pl.m1, stream);
    if (pl.s2_tiles) {
        const uint32_t b2 = pl.s2_tiles < (uint32_t)grid ? pl.s2_tiles : (uint32_t)grid;
        launch_i8_append(xs, mt, n_rows, pl.s2_stride, pl.s2_tiles, reinterpret_cast<const i32x4_t*>(qi), qm, B, ws, b2, stream);
        launch_tau_select(false, B, ws, 0u, pl.m2, stream);
    }
}

// ------------------------------------------------------------------------------------------------
// A second pass for the FLAGGED queries of a batch.  The sampled threshold of a query aims at ~1024 (4096) candidates; on topical
// rows it often lands ABOVE the query's k-th score and no certificate can hold — but the tail has rescored what it had, and the
// k-th exact distance d_k it found bounds the final one from above: every row that can still matter has ub > (1 - d_k) - 1e-4 (the
// bounded pass's arithmetic, scan_bounded.hip).  That is a threshold like any other: the matrix-core pass runs once more with it
// (every other query: +inf, nothing appended), and the same tail decides again with ALL rows above it as candidates — exact by the
// usual certificate when they fit the 8192 slots.  One 9-ms stream for all flagged queries of the batch instead of 7 ms per
// sixteen of them in the bounded pass, which keeps the ones that overflow (the near-tie shells of the largest clusters).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void i8_rerun_prepare_kernel(const uint32_t* __restrict__ flags, const float* __restrict__ out_dist,
                                                               const uint32_t* __restrict__ out_found, uint32_t k, int n_q,
                                                               float* __restrict__ tau, uint32_t* __restrict__ cnt,
                                                               uint32_t* __restrict__ go) {
    const int b = threadIdx.x;  // BATCH_QT = 256 threads
    bool fl = false;
    float t = POS_INF;
    if (b < n_q && flags[b] == FLAG_FALLBACK) {
        const uint32_t found = out_found[b];
        const float dk = found > 0 ? out_dist[(size_t)b * k + found - 1] : POS_INF;
        if (found == k && dk < POS_INF) {  // (k real rows with exact distances: the first result of a flagged query)
            t = __fsub_rn(__fsub_rn(1.0f, dk), 1.0e-4f);
            fl = true;
        }
    }
    tau[b] = t;
    go[4 + b] = 0u;  // (the pass's "lost" marks)
#pragma unroll
    for (int sg = 0; sg < BATCH_CAND_SEGS; ++sg) cnt[(size_t)b * BATCH_CAND_SEGS + sg] = 0u;
    const int any = __syncthreads_or(fl ? 1 : 0);
    if (b == 0) go[0] = any ? 1u : 0u;
}

void launch_i8_rerun(const void* d_x, int dtype, const void* d_i8, const void* d_meta, const uint64_t* d_ids, uint32_t n_rows,
                     const float* d_q, int B, uint32_t k, const BatchWorkspace& ws, int grid, uint64_t* d_labels, float* d_dist,
                     uint32_t* d_found, uint32_t* d_flags, uint32_t* d_go, hipStream_t stream) {
    // (the default 16x16x64 pass only — the other forms, option "mfma_sched", do not take the go word; a dense-only index has no threshold)
    if (n_rows <= (uint32_t)BATCH_CAP || ws.sched == 32 || (ws.sched >= 41 && ws.sched <= 50)) return;
    const BatchPlan pl = plan_batched_tiles(n_rows, I8_TILE_ROWS, ws.target, k);
    const signed char* qi = reinterpret_cast<const signed char*>(ws.qh);  // (the images of the first pass)
    const float2* qm = reinterpret_cast<const float2*>(qi + (size_t)BATCH_QT * EM);
    hipLaunchKernelGGL(i8_rerun_prepare_kernel, dim3(1), dim3(BATCH_QT), 0, stream, d_flags, d_dist, d_found, k, B, ws.tau, ws.cnt, d_go);
    const uint32_t blocks = pl.n_tiles_total < (uint32_t)grid ? pl.n_tiles_total : (uint32_t)grid;
    launch_i8_append(reinterpret_cast<const unsigned char*>(d_i8), reinterpret_cast<const float2*>(d_meta), n_rows, 1u, pl.n_tiles_total,
                     reinterpret_cast<const i32x4_t*>(qi), qm, B, ws, blocks, stream, d_go);
    launch_select_rescore_eps(false, d_x, dtype, d_ids, n_rows, d_q, B, k, ws, d_labels, d_dist, d_found, d_flags, 0, FILTER_EPS_I8,
                              stream, 1);
}

void launch_scan_batched_i8(const void* d_x, int dtype, const void* d_i8, const void* d_meta, const uint64_t* d_ids,
                            uint32_t n_rows, const float* d_q, int B, uint32_t k, const BatchWorkspace& ws, int grid,
                            uint64_t* d_labels, float* d_dist, uint32_t* d_found, uint32_t* d_flags, int force_fallback,
                            hipStream_t stream, hipEvent_t ev0, hipEvent_t ev1) {
    const BatchPlan pl = plan_batched_tiles(n_rows, I8_TILE_ROWS, ws.target, k);
    signed char* qi = reinterpret_cast<signed char*>(ws.qh);               // [256][384] int8
    float2* qm = reinterpret_cast<float2*>(qi + (size_t)BATCH_QT * EM);     // [256] {s_q, K2}
    hipLaunchKernelGGL(prep_queries_i8_kernel, dim3(BATCH_QT), dim3(64), 0, stream, d_q, B, qi, qm);
    const unsigned char* xs = reinterpret_cast<const unsigned char*>(d_i8);
    const float2* mt = reinterpret_cast<const float2*>(d_meta);
    auto pass = [&](bool dense_pass, uint32_t stride, uint32_t n_tiles) {
        if (n_tiles == 0) return;
        const uint32_t blocks = n_tiles < (uint32_t)grid ? n_tiles : (uint32_t)grid;
        if (dense_pass)
            launch_i8_dense(xs, mt, n_rows, stride, n_tiles, reinterpret_cast<const i32x4_t*>(qi), qm, B, ws, blocks, stream);
        else
            launch_i8_append(xs, mt, n_rows, stride, n_tiles, reinterpret_cast<const i32x4_t*>(qi), qm, B, ws, blocks, stream);
    };
    if (pl.dense_only) {
        if (ev0) (void)hipEventRecord(ev0, stream);
        pass(true, 1, pl.n_tiles_total);
        if (ev1) (void)hipEventRecord(ev1, stream);
        launch_select_rescore_eps(true, d_x, dtype, d_ids, n_rows, d_q, B, k, ws, d_labels, d_dist, d_found, d_flags,
                                  force_fallback, FILTER_EPS_I8, stream);
        return;
    }
    pass(true, pl.s1_stride, pl.s1_tiles);
    launch_tau_select(true, B, ws, pl.s1_tiles * I8_TILE_ROWS, pl.m1, stream);
    if (pl.s2_tiles) {  // (tau_select leaves the segment counters at zero)
        pass(false, pl.s2_stride, pl.s2_tiles);
        launch_tau_select(false, B, ws, 0u, pl.m2, stream);
    }
    if (ev0) (void)hipEventRecord(ev0, stream);
    pass(false, 1, pl.n_tiles_total);
    if (ev1) (void)hipEventRecord(ev1, stream);
    launch_select_rescore_eps(false, d_x, dtype, d_ids, n_rows, d_q, B, k, ws, d_labels, d_dist, d_found, d_flags,
                              force_fallback, FILTER_EPS_I8, stream);
}

}  // namespace dawn

#ifdef DAWN_EXPERIMENTS
extern "C" __attribute__((visibility("default"))) int dawn_debug_read_ts_pass(unsigned long long* out, int n) {
    unsigned long long h[1024];
    if (hipMemcpyFromSymbol(h, HIP_SYMBOL(dawn_ts_pass), sizeof(h)) != hipSuccess) return -1;
    for (int i = 0; i < n && i < 1024; ++i) out[i] = h[i];
    return 0;
}
extern "C" __attribute__((visibility("default"))) int dawn_debug_read_ts_i8(unsigned long long* out, int n) {
    unsigned long long h[64];
    if (hipMemcpyFromSymbol(h, HIP_SYMBOL(dawn_ts_i8), sizeof(h)) != hipSuccess) return -1;
    for (int i = 0; i < n && i < 64; ++i) out[i] = h[i];
    return 0;
}
#endif
