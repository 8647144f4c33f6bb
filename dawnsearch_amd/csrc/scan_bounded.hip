// scan_bounded.hip — the BOUNDED EXACT PASS: the rung of the search ladder between a failed certificate and the exact pass
// over all rows (scan_kernels.hip: scan_exact_kernel).
//
// Every filter of this library ends in a certificate: "no row outside the shortlist can beat the k-th exact distance".  It fails
// when more rows crowd the top of a query — within the filter's slack — than the fixed-size lists hold: 64 rows per workgroup on the
// int8 shadow (slack E + K2 ~ 0.012), 40-64 rows per WAVE on the packed 5-bit shadow (slack ~ 0.09).  On isotropic synthetic rows
// that never happens; on topical data it is ordinary: a query inside a cluster of a million pages whose mutual cosine is 0.9 has
// the whole cluster within 0.09 of its 10th best score (tools/clustered_probe.py).  Round 3 sent such a query to the exact pass:
// 1536 B per row, 22-33 ms per 100 M rows behind a 3.5-ms search.
//
// This pass answers it from the int8 shadow instead, without lists and therefore without a way to fail:
//   * it streams the int8 shadow exactly like the single-query filter (scan_i8.hip: global load -> v_mfma_i32_32x32x32_i8, the
//     query as two int8 images, threshold test of sub-tile t - 1 in the shadow of sub-tile t's MFMAs): 384 B per row;
//   * a row's upper bound ub >= x.q is compared with a THRESHOLD instead of a list: D = the k-th best exact distance known so far
//     — the failed stage's own result is the first one: its rows are real rows with exact distances, so its k-th distance bounds
//     the final one from above —; a row with ub <= (1 - D) - 1e-4 lies at distance > D (the certificate's arithmetic: dot <= ub +
//     FILTER_EPS_I8, d = fl(1 - dot) >= fl(1 - up(ub + eps)); 1e-4 covers eps = 2.9e-5 and every rounding on the way) and is
//     skipped; every other row is scored EXACTLY, in the reference's order (src/search/vector.rs:128-134), a lane per row, 64
//     rows per wave at a time, and enters the wave's exact top-64;
//   * D tightens as the wave finds better rows (its own k-th best distance is an upper bound of the final one too);
//   * the last workgroup to finish merges the exact lists and checks the ONE assumption it made: that the threshold it was given
//     was a valid upper bound — its own k-th distance must not exceed it.  It cannot, unless the failed stage's result was not
//     k distinct rows (no producer of this library writes such a result); should it happen all the same, that workgroup scans
//     all rows exactly by itself (block_exact_scan: slow, correct) and the query counts as a fallback.
//   * this pass is the LAST launch of a search on an index that keeps an int8 shadow (the exact pass over all rows closes the
//     searches of the others): it keeps the index's certificate counters and mirrors them to the host (see scan_exact_kernel).
// Cost: one int8 stream (100 M rows: 5.5 ms) + 1536 B for every row within the int8 slack of the k-th score — a few hundred rows on
// isotropic data, the dense part of a cluster on topical data — read at the exact pass's rate (lane-per-row walks, 4.6 TB/s with
// every wave of the chip at it): it degrades towards the exact pass's cost as the data gets denser, never beyond it + 5.5 ms.
// Predicated per query on d_flags[b] == FLAG_FALLBACK like the exact pass (launch-only searches: no host decision).
// A SINGLE query of an index with a live packed 5-bit shadow (scan_i6.hip; >= 2 Mi rows) streams THAT shadow instead (template
// parameter SH = 5: 240 B per row, the packed stream's loads and unpacking, its bound E (1 + k2u) + k2c).  Its bound is seven times
// as loose, so more rows reach the exact scores and a pass that starts without a threshold takes much longer to find one: a demoted
// query's pass is therefore SEEDED by a packed-stream search over 1/32 of the rows (dawn_index.cpp: bounded_packed_wanted /
// bounded_seed_wanted; 100 M topical rows 5.79 -> 4.17 ms per query, 12.5 M 0.88 -> 0.68; profiles/r04/bounded_packed_ab_*.log).
#include <type_traits>

#include "kernels.hpp"
#include "rotate384.hpp"
#include "packed_shadow.hpp"
#include "wave_topk.hpp"

namespace dawn {

constexpr float BOUNDED_MARGIN = 1.0e-4f;
constexpr int kBoundedMaxFlags = 256;

// The safety net of the bounded pass: exact top-64 of query qv over ALL rows by ONE workgroup (scan_exact_kernel's loop with the
// workgroup's own waves as the whole grid).  Result in wave 0.  ~N / 256 row walks per lane: seconds on 100 M rows — it exists so
// that an impossible threshold can never become a wrong answer, not to be fast.
template <int RT>
__device__ __noinline__ void block_exact_scan(const float* __restrict__ qv, const void* __restrict__ rows, uint32_t n_rows,
                                              float (*sh_s)[LIST], uint32_t (*sh_p)[LIST], float& s, uint32_t& p) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nwaves = blockDim.x >> 6;
    float ls = NEG_INF, tau = NEG_INF;
    uint32_t lp = NO_POS;
    const uint32_t n_groups = (n_rows + 63u) >> 6;
    for (uint32_t g = wave; g < n_groups; g += nwaves) {
        const uint32_t r = g * 64u + lane;
        float key = NEG_INF;
        if (r < n_rows) {
            const float d = __fsub_rn(1.0f, exact_dot_row<RT>(qv, rows, r));
            key = (d == d) ? -d : NEG_INF;
        }
        unsigned long long hits = __ballot(key > tau);  // (rows arrive in ascending order)
        while (hits) {
            const int src = __builtin_ctzll(hits);
            hits &= hits - 1;
            const float ks = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, key), src));
            if (ks > tau) {
                wave_insert(ls, lp, ks, g * 64u + (uint32_t)src, lane);
                tau = read_lane63(ls);
            }
        }
    }
    block_merge(ls, lp, sh_s, sh_p, wave, lane, nwaves);
    s = ls;
    p = lp;
}

// What scan_exact_kernel does for the searches it closes: the queries' final flags into the index's counters (the flagged ones
// are counted when this pass has answered them), and a copy of the counters to the host.  Block 0, after the flags were read.
__device__ __forceinline__ void bounded_count_and_mirror(uint32_t myflag, uint32_t* __restrict__ stats, uint32_t* __restrict__ mirror) {
    if (!stats || blockIdx.x != 0) return;
    if (myflag != FLAG_OK && myflag != FLAG_FALLBACK) atomicAdd(&stats[myflag], 1u);
    __syncthreads();
    if (mirror && threadIdx.x < (unsigned)N_STAT_SLOTS) mirror[threadIdx.x] = atomicAdd(&stats[threadIdx.x], 0u);
}

// SH = 8: the int8 shadow (384 B per row; PD fragments of 1 KB in flight per wave).  SH = 5: the packed 5-bit shadow of scan_i6.hip
// (240 B per row; its looser bound — E ~ 0.075 instead of ~ 0.01 — lets more rows through to the exact scores, 1536 B each: worth it
// while those stay well below the 144 B per row the stream saves; PD = 4 or 8 as there).  One query per launch either way.
template <int RT, int PD, int SH = 8>
__global__ __launch_bounds__(256) void scan_bounded_i8_kernel(const void* __restrict__ xv, const float2* __restrict__ meta,
                                                               const void* __restrict__ rows, const uint64_t* __restrict__ ids,
                                                               uint32_t n_rows, const float* __restrict__ q, int n_q,
                                                               uint32_t* __restrict__ flags, uint32_t* __restrict__ done,
                                                               float* __restrict__ out_s, uint32_t* __restrict__ out_p,
                                                               uint32_t n_lists, uint32_t k, uint64_t* __restrict__ out_labels,
                                                               float* __restrict__ out_dist, uint32_t* __restrict__ out_found,
                                                               uint32_t* __restrict__ stats, uint32_t* __restrict__ mirror) {
    static_assert(SH == 5 ? (PD == 4 || PD == 8) : 12 % PD == 0, "the ring must divide the 12 k-steps of a sub-tile");
    const u32x4* x = reinterpret_cast<const u32x4*>(xv);        // SH = 8
    const uint32_t* x5 = reinterpret_cast<const uint32_t*>(xv);  // SH = 5
    constexpr int NH = SH == 5 ? PD / 4 : 1, NN = SH == 5 ? 3 * PD / 4 : 1, NA = SH == 5 ? 1 : PD;
    __shared__ int sh_sum[2];
    __shared__ float sh_s[4][LIST];
    __shared__ uint32_t sh_p[4][LIST];
    __shared__ uint32_t sh_queue[4][LIST];  // rows waiting for their exact score, per wave
    __shared__ float sh_strip[4][32];       // a sub-tile's 32 upper bounds, one strip per wave
    __shared__ __attribute__((aligned(16))) signed char sh_img[2][EM];
    __shared__ __attribute__((aligned(16))) float sh_q[EM];
    __shared__ float sh_sq;
    __shared__ uint32_t sh_last;
    __shared__ unsigned long long sh_mask[kBoundedMaxFlags / 64];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nwaves = blockDim.x >> 6;
    const uint32_t n_sub = (n_rows + 31u) >> 5;
    const uint32_t c = lane & 31, h = lane >> 5;
    const uint32_t t_stride = gridDim.x * nwaves;
    {
        const uint32_t myflag = (int)threadIdx.x < n_q ? flags[threadIdx.x] : FLAG_OK;
        const unsigned long long m = __ballot(myflag == FLAG_FALLBACK);
        if (lane == 0) sh_mask[wave] = m;
        bounded_count_and_mirror(myflag, stats, mirror);
    }
    __syncthreads();
    for (int w = 0; w < kBoundedMaxFlags / 64; ++w) {
      unsigned long long todo = sh_mask[w];  // block-uniform
      while (todo) {
        const int b = w * 64 + __builtin_ctzll(todo);
        todo &= todo - 1;
        const float* qv = q + (size_t)b * EM;
        const uint32_t found = n_rows < k ? n_rows : k;
        // the failed stage's k-th distance (what it wrote for a query it flagged: the best rows it found, exact distances)
        const float d_in = found > 0 ? out_dist[(size_t)b * k + found - 1] : POS_INF;

        uint32_t t = blockIdx.x * nwaves + wave;
        const u32x4* p = x + (size_t)(t < n_sub ? t : 0) * (12 * 64) + lane;
        const uint32_t* p5 = x5 + (size_t)(t < n_sub ? t : 0) * I5_SUB_DW;
        [[maybe_unused]] u32x4 a[NA];
        [[maybe_unused]] u32x3 hq[NH];
        [[maybe_unused]] u32x4 nq[NN];
        auto load_h = [&](const uint32_t* sub, int g) __attribute__((always_inline)) { return frag_load(sub + g * I5_HALF_DW + lane * 3); };
        auto load_n = [&](const uint32_t* sub, int pr) __attribute__((always_inline)) {  // pr = fragment pair 0..5
            return __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(sub + (pr / 3) * I5_HALF_DW + 192 + (pr % 3) * 256 + lane * 4));
        };
        float2 mt = {0.f, 0.f};
        if (t < n_sub) {
            if constexpr (SH == 5) {
#pragma unroll
                for (int g = 0; g < NH; ++g) {
                    hq[g] = load_h(p5, g);
#pragma unroll
                    for (int m = 0; m < 3; ++m) nq[3 * g + m] = load_n(p5, 3 * g + m);
                }
            } else {
#pragma unroll
                for (int d = 0; d < PD; ++d) a[d] = __builtin_nontemporal_load(p + d * 64);
            }
            mt = meta[t];
        }
        // the query: f32 copy for the exact scores, two int8 images for the bounds (scan_filter_i8s_kernel)
        for (int i = threadIdx.x; i < EM; i += blockDim.x) sh_q[i] = qv[i];
        if (wave == 0) {
            float v[6];
#pragma unroll
            for (int j = 0; j < 6; ++j) v[j] = qv[lane + 64 * j];
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
            for (int o = 32; o >= 1; o >>= 1) {  // (the packed codes are value + 16: the accumulators start from -16 x these sums)
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
        // ub = C g1 + g0(E):  int8 shadow: E + K2, K2 = I8_K2_PER_SQ s_q;  packed shadow: E + (I6_XNORM + E) k2u = E emul + k2c (scan_i6.hip)
        const float sq254 = sq / 254.0f, rsq254 = 254.0f / sq;
        const float k2u = I6_K2U_PER_SQ * sq;
        const float k2 = SH == 5 ? I6_XNORM * k2u : I8_K2_PER_SQ * sq;  // the part of the slack that does not depend on the sub-tile
        const float emul = SH == 5 ? 1.0f + k2u : 1.0f, emul_thr = emul * 1.000001f;
        const int acc0 = (SH == 5 && col_live) ? -PackedShadow<5>::OFFSET * sh_sum[c == 8 ? 1 : 0] : 0;
        const bool tested = c == 0;

        // the wave's exact list (key = -distance, descending) and the score threshold below which a row is skipped
        float ls = NEG_INF;
        uint32_t lp = NO_POS;
        float tau = d_in < POS_INF ? __fsub_rn(__fsub_rn(1.0f, d_in), BOUNDED_MARGIN) : NEG_INF;  // (NaN: nothing passes, the check fails)
        float tau_m = POS_INF;
        auto set_tau_m = [&]() __attribute__((always_inline)) {
            if (tested) {
                const float tk = tau - k2;
                tau_m = tk - fabsf(tk) * 1e-6f;
            }
        };
        set_tau_m();
        uint32_t n_wait = 0;  // rows in the wave's queue (wave-uniform)
        uint32_t n_exact = 0;  // rows this wave scored exactly (-> stats[STAT_BOUNDED_EXACT])
        uint32_t* queue = &sh_queue[wave][0];

        auto flush = [&]() __attribute__((always_inline)) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // (wave-private queue: LDS operations of a wave are in order)
            float key = NEG_INF;
            uint32_t row = NO_POS;
            if ((uint32_t)lane < n_wait) {
                row = queue[lane];
                const float dot = exact_dot_row<RT>(sh_q, rows, row);
                const float d = __fsub_rn(1.0f, dot);  // vector.rs:133  1.0 - result
                if (d == d) key = -d;
                else row = NO_POS;
            }
            n_exact += n_wait;
            n_wait = 0;
            // rows arrive in ascending order (queue order = stream order), so a later row never displaces an equal key
            const float t64 = read_lane63(ls);
            unsigned long long hits = __ballot(row != NO_POS && key > t64);
            if (__popcll(hits) > 8) {
                float d = (row != NO_POS && key > t64) ? -key : POS_INF;
                uint32_t pr = (row != NO_POS && key > t64) ? row : NO_POS;
                sort64_asc(d, pr, lane);
                const float os = -__shfl(d, 63 - lane);
                const uint32_t op = __shfl(pr, 63 - lane);
                merge64(ls, lp, os, op, lane);
            } else {
                float tl = t64;
                while (hits) {
                    const int src = __builtin_ctzll(hits);
                    hits &= hits - 1;
                    const float ks = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, key), src));
                    const uint32_t kr = (uint32_t)__builtin_amdgcn_readlane((int)row, src);
                    if (ks > tl) {
                        wave_insert(ls, lp, ks, kr, lane);
                        tl = read_lane63(ls);
                    }
                }
            }
            // the wave's own k-th best distance bounds the final one from above as well
            if (found > 0 && (uint32_t)__builtin_amdgcn_readlane((int)lp, (int)found - 1) != NO_POS) {
                const float dkw = -__builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, ls), (int)found - 1));
                const float tw = __fsub_rn(__fsub_rn(1.0f, dkw), BOUNDED_MARGIN);
                if (tw > tau) {
                    tau = tw;
                    set_tau_m();
                }
            }
        };

        if (t < n_sub) {
            i32x16_t accs[2];
            float2 pmt = mt;
            uint32_t prow = 0;
            int C[16];
            int thr = 0, mx = 0;

            auto slow_path = [&]() __attribute__((always_inline)) {
                float* strip = &sh_strip[wave][0];
                if (c == 0) {  // lanes 0 (h = 0) and 32 (h = 1) hold the sub-tile's 32 sums
                    const float g1 = __builtin_amdgcn_rcpf(pmt.x) * sq254, g0 = __builtin_fmaf(pmt.y, emul, k2);
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const uint32_t roff = (uint32_t)((e & 3) + 8 * (e >> 2)) + 4u * h;
                        const bool ok = C[e] > thr && prow + roff < n_rows;
                        strip[roff] = ok ? __builtin_fmaf((float)C[e], g1, g0) : NEG_INF;
                    }
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                const bool hit = lane < 32 && strip[lane & 31] > tau;
                asm volatile("" ::: "memory");
                const unsigned long long m = __ballot(hit);
                if (hit) queue[n_wait + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = prow + (uint32_t)lane;
                n_wait += (uint32_t)__popcll(m);
                if (n_wait > 32u) flush();  // (a sub-tile adds at most 32)
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
                i32x16_t& acc = accs[P];
                const uint32_t tn = t + t_stride;
                more = tn < n_sub;
                const u32x4* pn = more ? x + (size_t)tn * (12 * 64) + lane : p;
                const uint32_t* pn5 = more ? x5 + (size_t)tn * I5_SUB_DW : p5;
                const float2 mtn = meta[more ? tn : t];
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[e] = acc0;
                if constexpr (SH == 5) {
                    // (scan_filter_i6s_kernel's 5-bit loop: a fragment pair = one 16-B load of nibbles + a third of a 12-B load of
                    // fifth bits -> two MFMA operands)
#pragma unroll
                    for (int pr = 0; pr < 6; ++pr) {
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
                        if constexpr (decltype(with_test)::value) test_slice(2 * pr, accs[1 - P]);
                        __builtin_amdgcn_sched_barrier(0);
                        bv[0] = (int)((n2 & 0x0F0F0F0Fu) | ((H << 4) & 0x10101010u));
                        bv[1] = (int)(((n2 >> 4) & 0x0F0F0F0Fu) | ((H << 3) & 0x10101010u));
                        bv[2] = (int)((n3 & 0x0F0F0F0Fu) | ((H << 2) & 0x10101010u));
                        bv[3] = (int)(((n3 >> 4) & 0x0F0F0F0Fu) | ((H << 1) & 0x10101010u));
                        acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(bv, qf[2 * pr + 1], acc, 0, 0, 0);
                        if (pr + NN < 6) nq[pr % NN] = load_n(p5, pr + NN);
                        else nq[pr % NN] = load_n(pn5, pr + NN - 6);
                        if (m == 2) {
                            if (g + NH < 2) hq[g % NH] = load_h(p5, g + NH);
                            else hq[g % NH] = load_h(pn5, g + NH - 2);
                        }
                        if constexpr (decltype(with_test)::value) test_slice(2 * pr + 1, accs[1 - P]);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                } else {
#pragma unroll
                    for (int f = 0; f < 12; ++f) {
                        acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(__builtin_bit_cast(i32x4_t, a[f % NA]), qf[f], acc, 0, 0, 0);
                        if (f + PD < 12) a[f % NA] = __builtin_nontemporal_load(p + (f + PD) * 64);
                        else a[f % NA] = __builtin_nontemporal_load(pn + (f + PD - 12) * 64);
                        if constexpr (decltype(with_test)::value) test_slice(f, accs[1 - P]);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
                if constexpr (decltype(with_test)::value)
                    if (__any(mx > thr)) slow_path();
                pmt = mt;
                prow = t * 32u;
                t = tn;
                p = pn;
                p5 = pn5;
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
        if (n_wait > 0u) flush();
        if (stats && lane == 0 && n_exact) atomicAdd(&stats[STAT_BOUNDED_EXACT], n_exact);

        block_merge(ls, lp, sh_s, sh_p, wave, lane, nwaves);
        if (wave == 0) {
            const size_t o = ((size_t)b * n_lists + blockIdx.x) * LIST + lane;
            out_s[o] = ls;
            out_p[o] = lp;
        }
        __threadfence();  // this workgroup's list is visible device-wide before it counts itself in
        __syncthreads();
        if (threadIdx.x == 0) sh_last = atomicAdd(&done[b], 1u) == gridDim.x - 1u ? 1u : 0u;
        __syncthreads();
        if (sh_last) {  // block-uniform: every other workgroup's list of query b is complete
            __threadfence();
            float s = NEG_INF;
            uint32_t pp = NO_POS;
            const float* cs = out_s + (size_t)b * n_lists * LIST;
            const uint32_t* cp = out_p + (size_t)b * n_lists * LIST;
            for (uint32_t l = wave; l < gridDim.x; l += nwaves)
                merge64(s, pp, cs[(size_t)l * LIST + 63 - lane], cp[(size_t)l * LIST + 63 - lane], lane);
            block_merge(s, pp, sh_s, sh_p, wave, lane, nwaves);
            if (wave == 0) {
                const uint32_t have = __popcll(__ballot(pp != NO_POS));
                bool valid = have >= found;
                if (valid && found > 0) {
                    const float dk = -__builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, s), (int)found - 1));
                    valid = dk <= d_in;  // the threshold this pass started from was an upper bound of the k-th distance
                }
                if (lane == 0) sh_last = valid ? 1u : 2u;
            }
            __syncthreads();
            const bool valid = sh_last == 1u;
            if (!valid) block_exact_scan<RT>(qv, rows, n_rows, sh_s, sh_p, s, pp);  // (never, see the header)
            if (wave == 0) {
                if ((uint32_t)lane < found) {
                    out_labels[(size_t)b * k + lane] = ids[pp];
                    out_dist[(size_t)b * k + lane] = -s;
                }
                if (lane == 0) {
                    out_found[b] = found;
                    if (valid) flags[b] = FLAG_BOUNDED;
                    if (stats) atomicAdd(&stats[valid ? FLAG_BOUNDED : FLAG_FALLBACK], 1u);
                    done[b] = 0u;
                }
            }
        }
        __syncthreads();  // the shared state is reused by the next query
      }
    }
}

// ------------------------------------------------------------------------------------------------
// The same pass for the flagged queries of a BATCH, sixteen per stream of the shadow.  A batch of 256 topical queries can
// leave a hundred certificates open (tools/clustered_probe.py: 108 of 256 on 12.5 M rows); one stream per query would cost
// 100 x 0.7 ms behind a 1.2-ms batch.  The 32 columns of the integer MFMA hold the two int8 images of 16 queries — columns
// c and c + 8 of each 16-lane row: H and L of one query, so that C = 254 acc_H + acc_L still comes from one DPP shift —, every
// tested lane carries its own query's threshold, hits are queued as (row, query slot) and scored a lane per entry; the waves'
// exact lists (16 x 64 entries per wave) live in LDS.
// ------------------------------------------------------------------------------------------------
constexpr int BQ = 16;          // queries per stream
constexpr int BQ_QSTRIDE = 388; // floats per staged query (odd multiple of 4: lanes reading different queries hit different banks)
constexpr int BQ_QCAP = 640;    // queue entries per wave (< 64 waiting + <= 32 rows x 16 queries of one sub-tile)
template <int NW>  // waves per workgroup: 4 (93 KB) or 8 (149 KB)
struct BoundedMultiLds {
    float q[BQ][BQ_QSTRIDE];
    signed char img[2][BQ][EM];
    float lists_s[NW][BQ][LIST];
    uint32_t lists_p[NW][BQ][LIST];
    uint2 queue[NW][BQ_QCAP];
    float merge_s[NW][LIST];
    uint32_t merge_p[NW][LIST];
    float sq[BQ];
    int sum[2][BQ];  // sums of the queries' int8 images (SH = 5: accumulator offsets)
    float d_in[BQ];
    uint32_t flagged[kBoundedMaxFlags];
    unsigned long long mask[kBoundedMaxFlags / 64];
    uint32_t last;
};

// SH = 5: the packed 5-bit shadow instead of the int8 one (as in scan_bounded_i8_kernel; PD = 4 or 8): every query of a batch's ladder
// comes with a first threshold, which is what the looser bound needs
template <int RT, int PD, int NW = 4, int SH = 8>
__global__ __launch_bounds__(64 * NW) void scan_bounded_i8_multi_kernel(const void* __restrict__ xv, const float2* __restrict__ meta,
                                                                     const void* __restrict__ rows, const uint64_t* __restrict__ ids,
                                                                     uint32_t n_rows, const float* __restrict__ q, int n_q,
                                                                     uint32_t* __restrict__ flags, uint32_t* __restrict__ done,
                                                                     float* __restrict__ out_s, uint32_t* __restrict__ out_p,
                                                                     uint32_t n_lists, uint32_t k, uint64_t* __restrict__ out_labels,
                                                                     float* __restrict__ out_dist, uint32_t* __restrict__ out_found,
                                                                     uint32_t* __restrict__ stats, uint32_t* __restrict__ mirror) {
    static_assert(SH == 5 ? (PD == 4 || PD == 8) : 12 % PD == 0, "the ring must divide the 12 k-steps of a sub-tile");
    const u32x4* x = reinterpret_cast<const u32x4*>(xv);        // SH = 8
    const uint32_t* x5 = reinterpret_cast<const uint32_t*>(xv);  // SH = 5
    constexpr int NH = SH == 5 ? PD / 4 : 1, NN = SH == 5 ? 3 * PD / 4 : 1, NA = SH == 5 ? 1 : PD;
    extern __shared__ __attribute__((aligned(16))) unsigned char bounded_lds[];
    BoundedMultiLds<NW>& S = *reinterpret_cast<BoundedMultiLds<NW>*>(bounded_lds);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nwaves = NW;
    const uint32_t n_sub = (n_rows + 31u) >> 5;
    const uint32_t c = lane & 31, h = lane >> 5;
    const uint32_t t_stride = gridDim.x * nwaves;
    const uint32_t found = n_rows < k ? n_rows : k;
    // the flagged queries, in order (every workgroup reads the flags before any of them is rewritten: a query is finished by
    // the last workgroup to arrive, after all of them have been here)
    {
        const uint32_t myflag = (int)threadIdx.x < n_q ? flags[threadIdx.x] : FLAG_OK;
        const bool fl = myflag == FLAG_FALLBACK;
        const unsigned long long m = __ballot(fl);
        if (lane == 0 && wave < kBoundedMaxFlags / 64) S.mask[wave] = m;  // (the flags live in the first four waves)
        bounded_count_and_mirror(myflag, stats, mirror);
        __syncthreads();
        uint32_t rank = (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
        for (int w = 0; w < wave && w < kBoundedMaxFlags / 64; ++w) rank += (uint32_t)__popcll(S.mask[w]);
        if (fl) S.flagged[rank] = threadIdx.x;
        __syncthreads();
    }
    uint32_t n_flagged = 0;
    for (int w = 0; w < kBoundedMaxFlags / 64; ++w) n_flagged += (uint32_t)__popcll(S.mask[w]);

    const uint32_t slot = (c & 7u) + ((c >> 4) << 3);  // the query slot of this lane's column
    for (uint32_t g0 = 0; g0 < n_flagged; g0 += BQ) {
        const uint32_t ng = n_flagged - g0 < (uint32_t)BQ ? n_flagged - g0 : (uint32_t)BQ;
        uint32_t t = blockIdx.x * nwaves + wave;
        const u32x4* p = x + (size_t)(t < n_sub ? t : 0) * (12 * 64) + lane;
        const uint32_t* p5 = x5 + (size_t)(t < n_sub ? t : 0) * I5_SUB_DW;
        [[maybe_unused]] u32x4 a[NA];
        [[maybe_unused]] u32x3 hq[NH];
        [[maybe_unused]] u32x4 nq[NN];
        auto load_h = [&](const uint32_t* sub, int g) __attribute__((always_inline)) { return frag_load(sub + g * I5_HALF_DW + lane * 3); };
        auto load_n = [&](const uint32_t* sub, int pr) __attribute__((always_inline)) {  // pr = fragment pair 0..5
            return __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(sub + (pr / 3) * I5_HALF_DW + 192 + (pr % 3) * 256 + lane * 4));
        };
        float2 mt = {0.f, 0.f};
        if (t < n_sub) {
            if constexpr (SH == 5) {
#pragma unroll
                for (int g = 0; g < NH; ++g) {
                    hq[g] = load_h(p5, g);
#pragma unroll
                    for (int m = 0; m < 3; ++m) nq[3 * g + m] = load_n(p5, 3 * g + m);
                }
            } else {
#pragma unroll
                for (int d = 0; d < PD; ++d) a[d] = __builtin_nontemporal_load(p + d * 64);
            }
            mt = meta[t];
        }
        // the group's queries: f32 copies, int8 images (a wave per query), the failed stages' k-th distances
        for (uint32_t i = threadIdx.x; i < ng * EM; i += blockDim.x) {
            const uint32_t sidx = i / EM, e = i % EM;
            S.q[sidx][e] = q[(size_t)S.flagged[g0 + sidx] * EM + e];
        }
        if (threadIdx.x < (uint32_t)BQ)
            S.d_in[threadIdx.x] = (threadIdx.x < ng && found > 0) ? out_dist[(size_t)S.flagged[g0 + threadIdx.x] * k + found - 1] : POS_INF;
        for (uint32_t sidx = wave; sidx < (uint32_t)BQ; sidx += nwaves) {
            float v[6];
            const float* qv = q + (size_t)S.flagged[g0 + (sidx < ng ? sidx : 0)] * EM;
#pragma unroll
            for (int j = 0; j < 6; ++j) v[j] = sidx < ng ? qv[lane + 64 * j] : 0.f;
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
                S.img[0][sidx][lane + 64 * j] = sidx < ng ? (signed char)(int)H : (signed char)0;
                S.img[1][sidx][lane + 64 * j] = sidx < ng ? (signed char)(int)L : (signed char)0;
                sumH += sidx < ng ? (int)H : 0;
                sumL += sidx < ng ? (int)L : 0;
            }
#pragma unroll
            for (int o = 32; o >= 1; o >>= 1) {  // (the packed codes are value + 16: the accumulators start from -16 x these sums)
                sumH += __shfl_xor(sumH, o);
                sumL += __shfl_xor(sumL, o);
            }
            if (lane == 0) {
                S.sq[sidx] = sq;
                S.sum[0][sidx] = sumH;
                S.sum[1][sidx] = sumL;
            }
        }
        for (int i = lane; i < BQ * LIST; i += 64) {  // the wave's exact lists start empty
            (&S.lists_s[wave][0][0])[i] = NEG_INF;
            (&S.lists_p[wave][0][0])[i] = NO_POS;
        }
        __syncthreads();
        i32x4_t qf[12];
        const bool tested = (c & 8u) == 0u && slot < ng;
        const unsigned long long tested_mask = __ballot(tested);
        {
            const i32x4_t* img = reinterpret_cast<const i32x4_t*>(&S.img[(c & 8u) ? 1 : 0][slot][0]);
#pragma unroll
            for (int f = 0; f < 12; ++f) qf[f] = slot < ng ? img[2 * f + h] : i32x4_t{0, 0, 0, 0};
        }
        const float sq_l = S.sq[slot];
        // ub = C g1 + g0(E): int8 shadow E + K2; packed shadow E (1 + k2u) + k2c (scan_bounded_i8_kernel)
        const float sq254_l = sq_l / 254.0f, rsq254_l = 254.0f / sq_l;
        const float k2u_l = I6_K2U_PER_SQ * sq_l;
        const float k2_l = SH == 5 ? I6_XNORM * k2u_l : I8_K2_PER_SQ * sq_l;
        const float emul_l = SH == 5 ? 1.0f + k2u_l : 1.0f, emul_thr_l = emul_l * 1.000001f;
        const int acc0 = (SH == 5 && slot < ng) ? -PackedShadow<5>::OFFSET * S.sum[(c & 8u) ? 1 : 0][slot] : 0;
        const float d_in_l = S.d_in[slot];
        float tau_l = (tested && d_in_l < POS_INF) ? __fsub_rn(__fsub_rn(1.0f, d_in_l), BOUNDED_MARGIN) : NEG_INF;
        float tau_m = POS_INF;
        auto set_tau_m = [&]() __attribute__((always_inline)) {
            if (tested) {
                const float tk = tau_l - k2_l;
                tau_m = tk - fabsf(tk) * 1e-6f;
            }
        };
        set_tau_m();
        uint32_t n_wait = 0;  // entries in the wave's queue (wave-uniform)
        uint32_t n_exact = 0;  // (row, query) pairs this wave scored exactly (-> stats[STAT_BOUNDED_EXACT])
        uint2* queue = &S.queue[wave][0];

        // the top (up to) 64 entries of the queue: exact scores, a lane per entry, into the lists of their queries
        auto flush64 = [&]() __attribute__((always_inline)) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // (wave-private queue: LDS operations of a wave are in order)
            const uint32_t n = n_wait < 64u ? n_wait : 64u;
            const uint32_t base = n_wait - n;
            n_wait = base;
            n_exact += n;
            float key = NEG_INF;
            uint32_t row = NO_POS, es = 0;
            if ((uint32_t)lane < n) {
                const uint2 ent = queue[base + lane];
                row = ent.x;
                es = ent.y;
                const float dot = exact_dot_row<RT>(&S.q[es][0], rows, row);
                const float d = __fsub_rn(1.0f, dot);  // vector.rs:133  1.0 - result
                if (d == d) key = -d;
                else row = NO_POS;
            }
            unsigned long long left = __ballot(row != NO_POS);
            while (left) {
                const int first = __builtin_ctzll(left);
                const uint32_t s_u = (uint32_t)__builtin_amdgcn_readlane((int)es, first);  // wave-uniform slot
                const bool mine = row != NO_POS && es == s_u;
                left &= ~__ballot(mine);
                float ls = S.lists_s[wave][s_u][lane];
                uint32_t lp = S.lists_p[wave][s_u][lane];
                const float t64s = read_lane63(ls);
                const uint32_t t64p = (uint32_t)__builtin_amdgcn_readlane((int)lp, 63);
                const bool cand = mine && better(key, row, t64s, t64p);
                unsigned long long hits = __ballot(cand);
                if (!hits) continue;
                if (__popcll(hits) > 8) {
                    float d = cand ? -key : POS_INF;
                    uint32_t pr = cand ? row : NO_POS;
                    sort64_asc(d, pr, lane);
                    const float os = -__shfl(d, 63 - lane);
                    const uint32_t op = __shfl(pr, 63 - lane);
                    merge64(ls, lp, os, op, lane);
                } else {
                    while (hits) {
                        const int src = __builtin_ctzll(hits);
                        hits &= hits - 1;
                        const float ks = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, key), src));
                        const uint32_t kr = (uint32_t)__builtin_amdgcn_readlane((int)row, src);
                        if (better(ks, kr, read_lane63(ls), (uint32_t)__builtin_amdgcn_readlane((int)lp, 63)))
                            wave_insert(ls, lp, ks, kr, lane);
                    }
                }
                S.lists_s[wave][s_u][lane] = ls;
                S.lists_p[wave][s_u][lane] = lp;
                // the wave's own k-th best distance of this query bounds the final one from above as well
                if (found > 0 && (uint32_t)__builtin_amdgcn_readlane((int)lp, (int)found - 1) != NO_POS) {
                    const float dkw = -__builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, ls), (int)found - 1));
                    const float tw = __fsub_rn(__fsub_rn(1.0f, dkw), BOUNDED_MARGIN);
                    if (tested && slot == s_u && tw > tau_l) {
                        tau_l = tw;
                        set_tau_m();
                    }
                }
            }
        };

        if (t < n_sub) {
            i32x16_t accs[2];
            float2 pmt = mt;
            uint32_t prow = 0;
            int C[16];
            int thr = 0, mx = 0;

            auto slow_path = [&]() __attribute__((always_inline)) {
                const float g1 = __builtin_amdgcn_rcpf(pmt.x) * sq254_l, g0 = __builtin_fmaf(pmt.y, emul_l, k2_l);
                // (sixteen compares into scalar masks first: with 16 queries per stream some lane is over its threshold in most
                // sub-tiles of a topical index, but in one or two of the sixteen accumulator registers only — the rest of the work
                // is skipped by scalar branches)
                unsigned long long mk[16];
#pragma unroll
                for (int e = 0; e < 16; ++e) mk[e] = __ballot(C[e] > thr) & tested_mask;
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    if (mk[e] == 0ull) continue;
                    const uint32_t row = prow + (uint32_t)((e & 3) + 8 * (e >> 2)) + 4u * h;
                    const bool hit = tested && C[e] > thr && row < n_rows && __builtin_fmaf((float)C[e], g1, g0) > tau_l;
                    const unsigned long long m = __ballot(hit);
                    if (m) {
                        if (hit) queue[n_wait + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = uint2{row, slot};
                        n_wait += (uint32_t)__popcll(m);
                    }
                }
                while (n_wait >= 64u) flush64();
            };
            auto test_slice = [&](int s, const i32x16_t& pacc) __attribute__((always_inline)) {
                if (s == 0) {
                    const float u = __builtin_fmaf(-pmt.y, emul_thr_l, tau_m);
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
            bool more;
            auto round = [&](auto with_test, auto parity) __attribute__((always_inline)) {
                constexpr int P = decltype(parity)::value;
                i32x16_t& acc = accs[P];
                const uint32_t tn = t + t_stride;
                more = tn < n_sub;
                const u32x4* pn = more ? x + (size_t)tn * (12 * 64) + lane : p;
                const uint32_t* pn5 = more ? x5 + (size_t)tn * I5_SUB_DW : p5;
                const float2 mtn = meta[more ? tn : t];
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[e] = acc0;
                if constexpr (SH == 5) {
#pragma unroll
                    for (int pr = 0; pr < 6; ++pr) {  // (scan_filter_i6s_kernel's 5-bit loop)
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
                        if constexpr (decltype(with_test)::value) test_slice(2 * pr, accs[1 - P]);
                        __builtin_amdgcn_sched_barrier(0);
                        bv[0] = (int)((n2 & 0x0F0F0F0Fu) | ((H << 4) & 0x10101010u));
                        bv[1] = (int)(((n2 >> 4) & 0x0F0F0F0Fu) | ((H << 3) & 0x10101010u));
                        bv[2] = (int)((n3 & 0x0F0F0F0Fu) | ((H << 2) & 0x10101010u));
                        bv[3] = (int)(((n3 >> 4) & 0x0F0F0F0Fu) | ((H << 1) & 0x10101010u));
                        acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(bv, qf[2 * pr + 1], acc, 0, 0, 0);
                        if (pr + NN < 6) nq[pr % NN] = load_n(p5, pr + NN);
                        else nq[pr % NN] = load_n(pn5, pr + NN - 6);
                        if (m == 2) {
                            if (g + NH < 2) hq[g % NH] = load_h(p5, g + NH);
                            else hq[g % NH] = load_h(pn5, g + NH - 2);
                        }
                        if constexpr (decltype(with_test)::value) test_slice(2 * pr + 1, accs[1 - P]);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                } else {
#pragma unroll
                    for (int f = 0; f < 12; ++f) {
                        acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(__builtin_bit_cast(i32x4_t, a[f % NA]), qf[f], acc, 0, 0, 0);
                        if (f + PD < 12) a[f % NA] = __builtin_nontemporal_load(p + (f + PD) * 64);
                        else a[f % NA] = __builtin_nontemporal_load(pn + (f + PD - 12) * 64);
                        if constexpr (decltype(with_test)::value) test_slice(f, accs[1 - P]);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
                if constexpr (decltype(with_test)::value)
                    if (__any(mx > thr)) slow_path();
                pmt = mt;
                prow = t * 32u;
                t = tn;
                p = pn;
                p5 = pn5;
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
        while (n_wait > 0u) flush64();
        if (stats && lane == 0 && n_exact) atomicAdd(&stats[STAT_BOUNDED_EXACT], n_exact);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");

        // the workgroup's list of every query of the group
        for (uint32_t sidx = 0; sidx < ng; ++sidx) {
            float ls = S.lists_s[wave][sidx][lane];
            uint32_t lp = S.lists_p[wave][sidx][lane];
            block_merge(ls, lp, S.merge_s, S.merge_p, wave, lane, nwaves);
            if (wave == 0) {
                const size_t o = ((size_t)S.flagged[g0 + sidx] * n_lists + blockIdx.x) * LIST + lane;
                out_s[o] = ls;
                out_p[o] = lp;
            }
        }
        __threadfence();  // this workgroup's lists are visible device-wide before it counts itself in
        __syncthreads();
        const uint32_t b_first = S.flagged[g0];
        if (threadIdx.x == 0) S.last = atomicAdd(&done[b_first], 1u) == gridDim.x - 1u ? 1u : 0u;
        __syncthreads();
        if (S.last) {  // block-uniform: every other workgroup's lists of this group are complete
            __threadfence();
            for (uint32_t sidx = 0; sidx < ng; ++sidx) {
                const uint32_t b = S.flagged[g0 + sidx];
                float s = NEG_INF;
                uint32_t pp = NO_POS;
                const float* cs = out_s + (size_t)b * n_lists * LIST;
                const uint32_t* cp = out_p + (size_t)b * n_lists * LIST;
                for (uint32_t l = wave; l < gridDim.x; l += nwaves)
                    merge64(s, pp, cs[(size_t)l * LIST + 63 - lane], cp[(size_t)l * LIST + 63 - lane], lane);
                block_merge(s, pp, S.merge_s, S.merge_p, wave, lane, nwaves);
                if (wave == 0) {
                    const uint32_t have = __popcll(__ballot(pp != NO_POS));
                    bool valid = have >= found;
                    if (valid && found > 0) {
                        const float dk = -__builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, s), (int)found - 1));
                        valid = dk <= S.d_in[sidx];
                    }
                    if (lane == 0) S.last = valid ? 1u : 2u;
                }
                __syncthreads();
                const bool valid = S.last == 1u;
                if (!valid) block_exact_scan<RT>(q + (size_t)b * EM, rows, n_rows, S.merge_s, S.merge_p, s, pp);  // (never)
                if (wave == 0) {
                    if ((uint32_t)lane < found) {
                        out_labels[(size_t)b * k + lane] = ids[pp];
                        out_dist[(size_t)b * k + lane] = -s;
                    }
                    if (lane == 0) {
                        out_found[b] = found;
                        if (valid) flags[b] = FLAG_BOUNDED;
                        if (stats) atomicAdd(&stats[valid ? FLAG_BOUNDED : FLAG_FALLBACK], 1u);
                    }
                }
                __syncthreads();  // (S.last is rewritten for the next query)
            }
            if (threadIdx.x == 0) done[b_first] = 0u;
        }
        __syncthreads();  // the shared state is reused by the next group
    }
}

// Per query b < B with d_flags[b] == FLAG_FALLBACK: the exact top-k through the int8 shadow, flag -> FLAG_BOUNDED; every other
// query is left alone (one nearly empty launch when no flag is set).  cand_s / cand_p: the per-workgroup lists [B][n_lists][64]
// (the filter's own, free by now); d_done [B]: arrival counters, zero before and after.  B <= 256 per launch.
// fragments a wave keeps in flight ahead of its MFMAs (6: half a sub-tile, 12: a whole one); process-wide, option "bounded_ring"
static int g_bounded_ring = 6;
void set_bounded_ring(int pd) { g_bounded_ring = pd == 12 ? 12 : 6; }
// waves per workgroup of the batch form (one workgroup per CU either way: its LDS); process-wide, option "bounded_multi_waves"
static int g_bounded_multi_packed = 0;  // the batch form on the packed 5-bit shadow (process-wide; option "bounded_multi_packed")
void set_bounded_multi_packed(int v) { g_bounded_multi_packed = v ? 1 : 0; }
static int g_bounded_multi_waves = 8;  // (74.0 against 77.4 ms per topical batch of 256 at 100 M rows: profiles/r04/bounded_multi_waves_100M.log)
void set_bounded_multi_waves(int nw) { g_bounded_multi_waves = nw == 8 ? 8 : 4; }

void launch_scan_bounded(const void* d_i8, const void* d_i8meta, const void* d_x, int dtype, const uint64_t* d_ids, uint32_t n_rows,
                         const float* d_q, int B, uint32_t* d_flags, uint32_t* d_done, float* cand_s, uint32_t* cand_p,
                         int n_lists, uint32_t k, uint64_t* d_labels, float* d_dist, uint32_t* d_found, hipStream_t stream,
                         uint32_t* d_stats, uint32_t* stats_mirror, const void* d_i5, const void* d_i5meta) {
    static OncePerDevice attr_once;
    once_per_device(attr_once, [] {
#define DAWN_BM_ATTR(RT_, PD_, NW_)                                                                      \
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(scan_bounded_i8_multi_kernel<RT_, PD_, NW_>), \
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(BoundedMultiLds<NW_>));
        DAWN_BM_ATTR(0, 6, 4) DAWN_BM_ATTR(1, 6, 4) DAWN_BM_ATTR(0, 12, 4) DAWN_BM_ATTR(1, 12, 4) DAWN_BM_ATTR(0, 6, 8) DAWN_BM_ATTR(1, 6, 8)
#undef DAWN_BM_ATTR
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(scan_bounded_i8_multi_kernel<0, 8, 8, 5>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(BoundedMultiLds<8>));
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(scan_bounded_i8_multi_kernel<1, 8, 8, 5>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(BoundedMultiLds<8>));
    });
    const u32x4* x8 = reinterpret_cast<const u32x4*>(d_i8);
    const float2* mt = reinterpret_cast<const float2*>(d_i8meta);
    for (int b0 = 0; b0 < B; b0 += kBoundedMaxFlags) {
        const int nb = B - b0 < kBoundedMaxFlags ? B - b0 : kBoundedMaxFlags;
#define DAWN_BOUNDED_ARGS                                                                                                      \
    x8, mt, d_x, d_ids, n_rows, d_q + (size_t)b0 * EM, nb, d_flags + b0, d_done + b0, cand_s + (size_t)b0 * n_lists * LIST,       \
        cand_p + (size_t)b0 * n_lists * LIST, (uint32_t)n_lists, k, d_labels + (size_t)b0 * k, d_dist + (size_t)b0 * k, d_found + b0,   \
        d_stats, stats_mirror
        const bool deep = g_bounded_ring == 12;
        if (B == 1 && d_i5 && d_i5meta) {  // one query, an index with a packed shadow: 240 B per row
#define DAWN_BOUNDED_ARGS5                                                                                                     \
    d_i5, reinterpret_cast<const float2*>(d_i5meta), d_x, d_ids, n_rows, d_q + (size_t)b0 * EM, nb, d_flags + b0, d_done + b0,   \
        cand_s + (size_t)b0 * n_lists * LIST, cand_p + (size_t)b0 * n_lists * LIST, (uint32_t)n_lists, k, d_labels + (size_t)b0 * k, \
        d_dist + (size_t)b0 * k, d_found + b0, d_stats, stats_mirror
            if (dtype == ROW_BF16)
                hipLaunchKernelGGL((scan_bounded_i8_kernel<1, 8, 5>), dim3(n_lists), dim3(256), 0, stream, DAWN_BOUNDED_ARGS5);
            else
                hipLaunchKernelGGL((scan_bounded_i8_kernel<0, 8, 5>), dim3(n_lists), dim3(256), 0, stream, DAWN_BOUNDED_ARGS5);
#undef DAWN_BOUNDED_ARGS5
        } else if (B == 1) {  // one query: its list stays in registers
            if (dtype == ROW_BF16) {
                if (deep) hipLaunchKernelGGL((scan_bounded_i8_kernel<1, 12>), dim3(n_lists), dim3(256), 0, stream, DAWN_BOUNDED_ARGS);
                else hipLaunchKernelGGL((scan_bounded_i8_kernel<1, 6>), dim3(n_lists), dim3(256), 0, stream, DAWN_BOUNDED_ARGS);
            } else {
                if (deep) hipLaunchKernelGGL((scan_bounded_i8_kernel<0, 12>), dim3(n_lists), dim3(256), 0, stream, DAWN_BOUNDED_ARGS);
                else hipLaunchKernelGGL((scan_bounded_i8_kernel<0, 6>), dim3(n_lists), dim3(256), 0, stream, DAWN_BOUNDED_ARGS);
            }
        } else {       // a batch: its flagged queries, sixteen per stream of the shadow
#define DAWN_BM_LAUNCH(RT_, PD_, NW_)                                                                                                  \
    hipLaunchKernelGGL((scan_bounded_i8_multi_kernel<RT_, PD_, NW_>), dim3(n_lists), dim3(64 * NW_), sizeof(BoundedMultiLds<NW_>), stream, \
                       DAWN_BOUNDED_ARGS)
            const int rt = dtype == ROW_BF16 ? 1 : 0;
            if (d_i5 && d_i5meta && g_bounded_multi_packed) {  // the packed 5-bit shadow (eight waves)
#define DAWN_BOUNDED_ARGS5M                                                                                                    \
    d_i5, reinterpret_cast<const float2*>(d_i5meta), d_x, d_ids, n_rows, d_q + (size_t)b0 * EM, nb, d_flags + b0, d_done + b0,   \
        cand_s + (size_t)b0 * n_lists * LIST, cand_p + (size_t)b0 * n_lists * LIST, (uint32_t)n_lists, k, d_labels + (size_t)b0 * k, \
        d_dist + (size_t)b0 * k, d_found + b0, d_stats, stats_mirror
                if (rt)
                    hipLaunchKernelGGL((scan_bounded_i8_multi_kernel<1, 8, 8, 5>), dim3(n_lists), dim3(512), sizeof(BoundedMultiLds<8>), stream,
                                       DAWN_BOUNDED_ARGS5M);
                else
                    hipLaunchKernelGGL((scan_bounded_i8_multi_kernel<0, 8, 8, 5>), dim3(n_lists), dim3(512), sizeof(BoundedMultiLds<8>), stream,
                                       DAWN_BOUNDED_ARGS5M);
#undef DAWN_BOUNDED_ARGS5M
            } else if (g_bounded_multi_waves == 8) {
                if (rt) DAWN_BM_LAUNCH(1, 6, 8); else DAWN_BM_LAUNCH(0, 6, 8);
            } else if (deep) {
                if (rt) DAWN_BM_LAUNCH(1, 12, 4); else DAWN_BM_LAUNCH(0, 12, 4);
            } else {
                if (rt) DAWN_BM_LAUNCH(1, 6, 4); else DAWN_BM_LAUNCH(0, 6, 4);
            }
#undef DAWN_BM_LAUNCH
        }
#undef DAWN_BOUNDED_ARGS
    }
}

// keep = 1: out_dist[found - 1] already holds a valid first threshold (the k-th exact distance of a search over a PART of the rows:
// launch_scan_bounded_direct's seed) — only the flag is raised
__global__ void bounded_prime_kernel(uint32_t* __restrict__ flags, float* __restrict__ out_dist, uint32_t found, float threshold,
                                     int keep) {
    if (threadIdx.x == 0) {
        flags[0] = FLAG_FALLBACK;
        if (found > 0 && !keep) out_dist[found - 1] = threshold;
    }
}

void launch_scan_bounded_direct(const void* d_i8, const void* d_i8meta, const void* d_x, int dtype, const uint64_t* d_ids,
                                uint32_t n_rows, const float* d_q, uint32_t* d_flags, uint32_t* d_done, float* cand_s,
                                uint32_t* cand_p, int n_lists, uint32_t k, uint64_t* d_labels, float* d_dist, uint32_t* d_found,
                                hipStream_t stream, hipEvent_t ev0, hipEvent_t ev1, uint32_t* d_stats, uint32_t* stats_mirror,
                                float first_threshold, const void* d_i5, const void* d_i5meta, bool seeded) {
    const uint32_t found = n_rows < k ? n_rows : k;
    hipLaunchKernelGGL(bounded_prime_kernel, dim3(1), dim3(64), 0, stream, d_flags, d_dist, found, first_threshold, seeded ? 1 : 0);
    if (ev0) (void)hipEventRecord(ev0, stream);
    launch_scan_bounded(d_i8, d_i8meta, d_x, dtype, d_ids, n_rows, d_q, 1, d_flags, d_done, cand_s, cand_p, n_lists, k, d_labels,
                        d_dist, d_found, stream, d_stats, stats_mirror, d_i5, d_i5meta);
    if (ev1) (void)hipEventRecord(ev1, stream);
}

}  // namespace dawn
