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

// ------------------------------------------------------------------------------------------------
// The WIDE batch form (round 5): SIXTY-FOUR flagged queries per stream of the int8 shadow.  The 16-query form above is one HBM stream
// per sixteen queries with the matrix pipe nearly idle: a topical batch that leaves 100-135 certificates open paid 7-9 streams
// (profiles/r04: 348 GB read per batch = 9.07 x the shadow).  Here the H and L images of 32 queries each fill the 32 columns of
// their own MFMAs — four MFMAs per k-step, 48 per 32-row sub-tile, C = 254 acc_H + acc_L —
// with all 192 query fragments resident in registers (one wave per SIMD, 512 registers),
// and nothing per query lives in LDS but its f32 copy:
//   * no lists.  A pair (row, query) whose int8 bound passes the query's threshold tau = (1 - D) - 1e-4 (D = the failed stage's
//     k-th exact distance; fixed for the pass) is queued BY ROW — (row, 32-query mask) — and re-tested on the f32 row itself: one
//     coalesced 1.5-KB read per row, an any-order f32 dot per hitting query (FILTER_EPS_F32 from the reference-order sum: the same
//     tau and the same 1e-4 cover it).  Only what passes that too — the rows within 1e-4 of the k-th score and the better ones —
//     is scored in the reference's order (vector.rs:128-134), and every pair at distance <= D is APPENDED to the query's result
//     buffer (global, BOUNDED_WIDE_CAP entries); bounded_wide_finish_kernel sorts them by (distance, row) and writes the top k.
//   * a query that overflows its buffer, or whose D is unusable, keeps its flag: the 16-query form, launched behind this one,
//     answers it with its lists (and closes the search: counters, mirror).
// ------------------------------------------------------------------------------------------------
constexpr uint32_t kBoundedWideMinRows = 4096;  // smaller indexes: the 16-query form (launch_scan_bounded)
constexpr int WQ = 64;            // queries per stream
constexpr int WQ_QSTRIDE = 388;   // floats per staged query (lanes reading different queries hit different banks)
constexpr int WQ_HITCAP = 128;    // (row, mask, group) entries per wave: drained from WQ_DRAIN_AT on, a sub-tile adds at most 64
constexpr int WQ_DRAIN_AT = 24;   // entries that make a drain worth its row reads (8 rows in flight per round)
constexpr int WQ_EXCAP = 1024;    // pairs per wave and group that may await their reference-order score (scored when the stream is through)
struct BoundedWideLds {
    float q[WQ][WQ_QSTRIDE];
    union {
        signed char img[2][WQ][EM];  // set-up only: the int8 images the waves load their fragments from
        struct {
            uint4 hit[4][WQ_HITCAP];  // {row, query mask, group, -}
            uint2 ex[4][WQ_EXCAP];    // {row, query slot}: pairs awaiting their reference-order score
        } w;
    } u;
    float sq[WQ];
    float d_in[WQ];
    float tau[WQ];
    uint32_t lost[WQ];  // != 0: a wave ran out of room for this query's pairs — the query keeps its flag
    uint32_t flagged[kBoundedMaxFlags];
    unsigned long long mask[kBoundedMaxFlags / 64];
};
static_assert(sizeof(BoundedWideLds) <= 160 * 1024, "one workgroup per CU: all of its LDS");

template <int RT>
__global__ __launch_bounds__(256) void scan_bounded_i8_wide_kernel(const u32x4* __restrict__ x, const float2* __restrict__ meta,
                                                                    const void* __restrict__ rows, uint32_t n_rows,
                                                                    const float* __restrict__ q, int n_q,
                                                                    const uint32_t* __restrict__ flags, uint32_t k,
                                                                    const float* __restrict__ out_dist, uint2* __restrict__ res,
                                                                    uint32_t* __restrict__ res_cnt, uint32_t* __restrict__ stats) {
    extern __shared__ __attribute__((aligned(16))) unsigned char bounded_lds[];
    BoundedWideLds& S = *reinterpret_cast<BoundedWideLds*>(bounded_lds);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t n_sub = (n_rows + 31u) >> 5;
    const uint32_t c = lane & 31, h = lane >> 5;
    const uint32_t t_stride = gridDim.x * 4u;
    const uint32_t found = n_rows < k ? n_rows : k;
    {
        const uint32_t myflag = (int)threadIdx.x < n_q ? flags[threadIdx.x] : FLAG_OK;
        const bool fl = myflag == FLAG_FALLBACK;
        const unsigned long long m = __ballot(fl);
        if (lane == 0) S.mask[wave] = m;
        __syncthreads();
        uint32_t rank = (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
        for (int w = 0; w < wave; ++w) rank += (uint32_t)__popcll(S.mask[w]);
        if (fl) S.flagged[rank] = threadIdx.x;
        __syncthreads();
    }
    uint32_t n_flagged = 0;
    for (int w = 0; w < kBoundedMaxFlags / 64; ++w) n_flagged += (uint32_t)__popcll(S.mask[w]);
    if (n_flagged == 0 || found == 0) return;  // (block-uniform)

    uint32_t n_pairs = 0;  // (row, query) pairs past the int8 bound (-> stats[STAT_BOUNDED_EXACT])
    for (uint32_t g0 = 0; g0 < n_flagged; g0 += WQ) {
        const uint32_t ng = n_flagged - g0 < (uint32_t)WQ ? n_flagged - g0 : (uint32_t)WQ;
        uint32_t t = blockIdx.x * 4u + wave;
        // two sub-tiles of fragments in flight per wave (24 KiB): the pass is 48 MFMAs per sub-tile, a whole one ahead hides HBM
        // The fragments are loaded by inline asm under hand-counted vmcnt: the wave keeps two sub-tiles (2 x 12 fragments) in flight, and
        // vmcnt retires loads in order.  hipcc's own waits came out as vmcnt(0) at the head of every sub-tile (registers shared with the
        // drain path joined as "pending" at the loop header): 7.0 ms per stream of 100 M rows, the MFMAs idle half the time.
        // Step f of a sub-tile needs its fragment f: the 11 - f behind it and the other sub-tile's 12 may stay out.  The fragments live
        // in AGPRs that nothing but these loads and the MFMAs touch (tests/test_abi_cpu.py checks the ISA: a register copy between
        // a load and its wait would read stale data).  The sub-tile's {scale, error bound} is a SCALAR load (its address is
        // wave-uniform): lgkmcnt, not vmcnt.
        u32x4 A0[12], A1[12];
        auto load_tile = [&](u32x4 (&A)[12], uint32_t tt) __attribute__((always_inline)) {
            const u32x4* pb = x + (size_t)(tt < n_sub ? tt : 0u) * (12 * 64) + lane;
#pragma unroll
            for (int f = 0; f < 12; ++f)
                asm volatile("global_load_dwordx4 %0, %1, off offset:%2 nt" : "=a"(A[f]) : "v"(pb + (f & ~3) * 64), "n"((f & 3) * 1024));
        };
        load_tile(A0, t);
        load_tile(A1, t + t_stride);
        // the group's queries: f32 copies, thresholds, int8 images (a wave per query)
        for (uint32_t i = threadIdx.x; i < (uint32_t)WQ * EM; i += blockDim.x) {
            const uint32_t sidx = i / EM, e = i % EM;
            S.q[sidx][e] = sidx < ng ? q[(size_t)S.flagged[g0 + sidx] * EM + e] : 0.f;
        }
        if (threadIdx.x < (uint32_t)WQ) {
            const uint32_t sidx = threadIdx.x;
            float d = POS_INF;
            if (sidx < ng) d = out_dist[(size_t)S.flagged[g0 + sidx] * k + found - 1];
            // a usable D is the k-th distance of k real rows: finite, and no wider than anything a unit-vector index holds.  Otherwise
            // nothing passes for this query, it collects nothing and keeps its flag (the 16-query form takes it)
            const bool ok = sidx < ng && d == d && d < 2.5f;
            S.d_in[sidx] = ok ? d : NEG_INF;
            S.tau[sidx] = ok ? __fsub_rn(__fsub_rn(1.0f, d), BOUNDED_MARGIN) : POS_INF;
            S.lost[sidx] = 0u;
        }
        for (uint32_t sidx = wave; sidx < (uint32_t)WQ; sidx += 4u) {
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
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                const float tt = v[j] / sq;
                const float H = fminf(fmaxf(rintf(tt), -127.f), 127.f);
                const float L = fminf(fmaxf(rintf((tt - H) * 254.0f), -127.f), 127.f);
                S.u.img[0][sidx][lane + 64 * j] = sidx < ng ? (signed char)(int)H : (signed char)0;
                S.u.img[1][sidx][lane + 64 * j] = sidx < ng ? (signed char)(int)L : (signed char)0;
            }
            if (lane == 0) S.sq[sidx] = sq;
        }
        __syncthreads();
        // this lane's columns: query slots c (group 0) and 32 + c (group 1); fragments H0 | L0 | H1 | L1
        i32x4_t qf[4][12];
#pragma unroll
        for (int gi = 0; gi < 4; ++gi) {
            const i32x4_t* img = reinterpret_cast<const i32x4_t*>(&S.u.img[gi & 1][32 * (gi >> 1) + c][0]);
#pragma unroll
            for (int f = 0; f < 12; ++f) qf[gi][f] = img[2 * f + h];
        }
        bool tested[2];
        float rsq254_l[2], k2_l[2], tau_l[2], tau_m[2];
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            const uint32_t slot = 32u * g + c;
            const float sq_l = S.sq[slot];
            tau_l[g] = S.tau[slot];
            tested[g] = slot < ng && tau_l[g] < POS_INF;
            rsq254_l[g] = 254.0f / sq_l;
            k2_l[g] = I8_K2_PER_SQ * sq_l;
            const float tk = tau_l[g] - k2_l[g];
            tau_m[g] = tested[g] ? tk - fabsf(tk) * 1e-6f : POS_INF;
        }
        __syncthreads();  // the images are in registers: their LDS becomes the waves' queues
        uint4* hitq = &S.u.w.hit[wave][0];
        uint2* exq = &S.u.w.ex[wave][0];
        uint32_t n_hit = 0, n_ex = 0;  // wave-uniform

        // pairs that passed both bounds: reference-order scores, a lane per pair; what lies at distance <= D is appended.  Runs when the
        // stream is through (hipcc-visible loads inside the stream loop would bring back its vmcnt(0) waits: see the fragment loads)
        auto flush_exact = [&]() __attribute__((always_inline)) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            for (uint32_t i0 = 0; i0 < n_ex; i0 += 64u) {
                if (i0 + (uint32_t)lane < n_ex) {
                    const uint2 ent = exq[i0 + lane];
                    const float dot = exact_dot_row<RT>(&S.q[ent.y][0], rows, ent.x);
                    const float d = __fsub_rn(1.0f, dot);  // vector.rs:133  1.0 - result
                    if (d == d && d <= S.d_in[ent.y]) {
                        const uint32_t b = S.flagged[g0 + ent.y];
                        const uint32_t pos = atomicAdd(&res_cnt[b], 1u);
                        if (pos < BOUNDED_WIDE_CAP) res[(size_t)b * BOUNDED_WIDE_CAP + pos] = uint2{__builtin_bit_cast(uint32_t, d), ent.x};
                    }
                }
            }
            n_ex = 0;
        };
        // the queued rows against the queries that hit them: the f32 row read once, coalesced; an any-order dot per query
        auto drain_hits = [&]() __attribute__((always_inline)) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            constexpr int RB = RT == 1 ? 8 : 6;  // rows in flight per round
            for (uint32_t i0 = 0; i0 < n_hit; i0 += RB) {
                float xr[RT == 1 ? 8 : 6];
                [[maybe_unused]] float2 x2[RB][3];
                [[maybe_unused]] u32x4 xw[RB];
                uint32_t erow[RB], emask[RB], egrp[RB];
#pragma unroll
                for (int j = 0; j < RB; ++j) {
                    const uint4 ent = hitq[i0 + j < n_hit ? i0 + j : i0];
                    erow[j] = (uint32_t)__builtin_amdgcn_readfirstlane((int)ent.x);
                    emask[j] = i0 + j < n_hit ? (uint32_t)__builtin_amdgcn_readfirstlane((int)ent.y) : 0u;
                    egrp[j] = (uint32_t)__builtin_amdgcn_readfirstlane((int)ent.z);
                }
                // The round's row reads and their wait are ONE asm statement: hipcc must not see a loaded register before its data is there
                // (hand-counted vmcnt as for the fragments; a copy it inserted between a load and the wait would read stale data).
                if constexpr (RT == 1) {  // bf16 rows in fragment order: 48 chunks of 8 values (lanes >= 48 re-read chunk 47)
                    const u32x4* xb = reinterpret_cast<const u32x4*>(rows);
                    const int cl = lane < ROW_C8 ? lane : ROW_C8 - 1;
                    const u32x4 *p0 = xb + frag_chunk(erow[0], cl), *p1 = xb + frag_chunk(erow[1], cl), *p2 = xb + frag_chunk(erow[2], cl),
                                *p3 = xb + frag_chunk(erow[3], cl), *p4 = xb + frag_chunk(erow[4], cl), *p5 = xb + frag_chunk(erow[5], cl),
                                *p6 = xb + frag_chunk(erow[6], cl), *p7 = xb + frag_chunk(erow[7], cl);
                    asm volatile(
                        "global_load_dwordx4 %0, %8, off\n\tglobal_load_dwordx4 %1, %9, off\n\tglobal_load_dwordx4 %2, %10, off\n\t"
                        "global_load_dwordx4 %3, %11, off\n\tglobal_load_dwordx4 %4, %12, off\n\tglobal_load_dwordx4 %5, %13, off\n\t"
                        "global_load_dwordx4 %6, %14, off\n\tglobal_load_dwordx4 %7, %15, off\n\ts_waitcnt vmcnt(0)"
                        : "=&v"(xw[0]), "=&v"(xw[1]), "=&v"(xw[2]), "=&v"(xw[3]), "=&v"(xw[4]), "=&v"(xw[5]), "=&v"(xw[6]), "=&v"(xw[7])
                        : "v"(p0), "v"(p1), "v"(p2), "v"(p3), "v"(p4), "v"(p5), "v"(p6), "v"(p7));
                } else {  // f32 rows: lane l holds elements 2l + 128 m, + 1 (three coalesced 512-B reads per row)
                    const float2* xb = reinterpret_cast<const float2*>(rows) + lane;
                    const float2 *p0 = xb + (size_t)erow[0] * (EM / 2), *p1 = xb + (size_t)erow[1] * (EM / 2), *p2 = xb + (size_t)erow[2] * (EM / 2),
                                 *p3 = xb + (size_t)erow[3] * (EM / 2), *p4 = xb + (size_t)erow[4] * (EM / 2), *p5 = xb + (size_t)erow[5] * (EM / 2);
                    asm volatile(
                        "global_load_dwordx2 %0, %18, off\n\tglobal_load_dwordx2 %1, %18, off offset:512\n\tglobal_load_dwordx2 %2, %18, off offset:1024\n\t"
                        "global_load_dwordx2 %3, %19, off\n\tglobal_load_dwordx2 %4, %19, off offset:512\n\tglobal_load_dwordx2 %5, %19, off offset:1024\n\t"
                        "global_load_dwordx2 %6, %20, off\n\tglobal_load_dwordx2 %7, %20, off offset:512\n\tglobal_load_dwordx2 %8, %20, off offset:1024\n\t"
                        "global_load_dwordx2 %9, %21, off\n\tglobal_load_dwordx2 %10, %21, off offset:512\n\tglobal_load_dwordx2 %11, %21, off offset:1024\n\t"
                        "global_load_dwordx2 %12, %22, off\n\tglobal_load_dwordx2 %13, %22, off offset:512\n\tglobal_load_dwordx2 %14, %22, off offset:1024\n\t"
                        "global_load_dwordx2 %15, %23, off\n\tglobal_load_dwordx2 %16, %23, off offset:512\n\tglobal_load_dwordx2 %17, %23, off offset:1024\n\t"
                        "s_waitcnt vmcnt(0)"
                        : "=&v"(x2[0][0]), "=&v"(x2[0][1]), "=&v"(x2[0][2]), "=&v"(x2[1][0]), "=&v"(x2[1][1]), "=&v"(x2[1][2]), "=&v"(x2[2][0]),
                          "=&v"(x2[2][1]), "=&v"(x2[2][2]), "=&v"(x2[3][0]), "=&v"(x2[3][1]), "=&v"(x2[3][2]), "=&v"(x2[4][0]), "=&v"(x2[4][1]),
                          "=&v"(x2[4][2]), "=&v"(x2[5][0]), "=&v"(x2[5][1]), "=&v"(x2[5][2])
                        : "v"(p0), "v"(p1), "v"(p2), "v"(p3), "v"(p4), "v"(p5));
                }
#pragma unroll
                for (int j = 0; j < RB; ++j) {
                    if constexpr (RT == 1) {
                        const u32x4 w = xw[j];
                        xr[0] = bf16_lo(w.x); xr[1] = bf16_hi(w.x); xr[2] = bf16_lo(w.y); xr[3] = bf16_hi(w.y);
                        xr[4] = bf16_lo(w.z); xr[5] = bf16_hi(w.z); xr[6] = bf16_lo(w.w); xr[7] = bf16_hi(w.w);
                    } else {
                        xr[0] = x2[j][0].x; xr[1] = x2[j][0].y; xr[2] = x2[j][1].x; xr[3] = x2[j][1].y; xr[4] = x2[j][2].x; xr[5] = x2[j][2].y;
                    }
                    uint32_t m = emask[j];
                    while (m) {
                        const uint32_t b = (uint32_t)__builtin_ctz(m);
                        m &= m - 1u;
                        const uint32_t s_u = 32u * egrp[j] + b;  // wave-uniform query slot
                        float p = 0.f;
                        if constexpr (RT == 1) {
                            if (lane < ROW_C8) {
                                const f32x4 qa = *reinterpret_cast<const f32x4*>(&S.q[s_u][8 * lane]);
                                const f32x4 qb = *reinterpret_cast<const f32x4*>(&S.q[s_u][8 * lane + 4]);
                                p = xr[0] * qa.x;
                                p = __builtin_fmaf(xr[1], qa.y, p);
                                p = __builtin_fmaf(xr[2], qa.z, p);
                                p = __builtin_fmaf(xr[3], qa.w, p);
                                p = __builtin_fmaf(xr[4], qb.x, p);
                                p = __builtin_fmaf(xr[5], qb.y, p);
                                p = __builtin_fmaf(xr[6], qb.z, p);
                                p = __builtin_fmaf(xr[7], qb.w, p);
                            }
                        } else {
                            const float2* qs = reinterpret_cast<const float2*>(&S.q[s_u][0]) + lane;
                            const float2 q0 = qs[0], q1 = qs[64], q2 = qs[128];
                            p = xr[0] * q0.x;
                            p = __builtin_fmaf(xr[1], q0.y, p);
                            p = __builtin_fmaf(xr[2], q1.x, p);
                            p = __builtin_fmaf(xr[3], q1.y, p);
                            p = __builtin_fmaf(xr[4], q2.x, p);
                            p = __builtin_fmaf(xr[5], q2.y, p);
                        }
                        const float tot = read_lane63(wave_sum_lane63(p));
                        ++n_pairs;
                        if (tot > S.tau[s_u]) {  // (a NaN passes nothing; the exact score would drop it as well)
                            if (n_ex < (uint32_t)WQ_EXCAP) {
                                if (lane == 0) exq[n_ex] = uint2{erow[j], s_u};
                                ++n_ex;
                            } else if (lane == 0) {
                                S.lost[s_u] = 1u;  // no room: the query is left to the 16-query form
                            }
                        }
                    }
                }
            }
            n_hit = 0;
        };

        if (t < n_sub) {
            // The hits of one group of a sub-tile: `bits` = this lane's 16 accumulator elements (rows (e & 3) + 8 (e >> 2) + 4 h of the
            // sub-tile, column = the lane's query) over the integer threshold.  Queued BY ROW: one ballot per element any lane has set.
            auto emit = [&](int g, uint32_t bits, uint32_t prow) __attribute__((always_inline)) {
                if (prow + 32u > n_rows) {  // the last sub-tile: rows past the end never hit
                    uint32_t vm = 0u;
#pragma unroll
                    for (int e = 0; e < 16; ++e)
                        if (prow + (uint32_t)((e & 3) + 8 * (e >> 2)) + 4u * h < n_rows) vm |= 1u << e;
                    bits &= vm;
                }
                if (!tested[g]) bits = 0u;
                if (!__any(bits != 0u)) return;
                uint32_t u = bits;  // OR over the wave (DPP; the total lands in lane 63)
                u |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)u, 0xB1, 0xf, 0xf, true);
                u |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)u, 0x4E, 0xf, 0xf, true);
                u |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)u, 0x141, 0xf, 0xf, true);
                u |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)u, 0x140, 0xf, 0xf, true);
                u |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)u, 0x142, 0xa, 0xf, false);
                u |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)u, 0x143, 0xc, 0xf, false);
                uint32_t U = (uint32_t)__builtin_amdgcn_readlane((int)u, 63);
                while (U) {
                    const uint32_t e = (uint32_t)__builtin_ctz(U);
                    U &= U - 1u;
                    const unsigned long long m = __ballot(((bits >> e) & 1u) != 0u);
                    const uint32_t r0 = prow + (e & 3u) + 8u * (e >> 2);
                    const uint32_t lo = (uint32_t)m, hi = (uint32_t)(m >> 32);
                    if (lo) {
                        if (lane == 0) hitq[n_hit] = uint4{r0, lo, (uint32_t)g, 0u};
                        ++n_hit;
                    }
                    if (hi) {
                        if (lane == 0) hitq[n_hit] = uint4{r0 + 4u, hi, (uint32_t)g, 0u};
                        ++n_hit;
                    }
                }
            };
            // C = 254 acc_H + acc_L (|acc_H| <= 384 x 127^2 < 2^23) of elements e, e + 1 against the integer threshold.  The test is the
            // conservative one of the streaming filters (thr sits 2 units below the bound's own crossing); what it lets through is
            // re-tested on the f32 row anyway
            auto post2 = [&](const i32x16_t& ahh, const i32x16_t& all_, int thr, uint32_t& bits, int e) __attribute__((always_inline)) {
                const int c0 = __mul24(ahh[e], 254) + all_[e];
                const int c1 = __mul24(ahh[e + 1], 254) + all_[e + 1];
                bits |= (c0 > thr ? 1u : 0u) << e;
                bits |= (c1 > thr ? 1u : 0u) << (e + 1);
            };
            auto thr_of = [&](int g, const float2 pmt) __attribute__((always_inline)) {
                const float u = __builtin_fmaf(-pmt.y, 1.000001f, tau_m[g]);
                float thr_f = __builtin_fmaf(u, pmt.x * rsq254_l[g], -2.0f);
                thr_f = fminf(fmaxf(thr_f, -2.0e9f), 2.0e9f);
                if (!tested[g]) thr_f = 2.0e9f;
                return (int)floorf(thr_f);
            };
            // One sub-tile held in A, in two phases of 24 MFMAs: group 0 (H0 | L0 chains) with the PREVIOUS sub-tile's group-1 sums tested
            // in its shadow, then group 1 (H1 | L1) with this sub-tile's group-0 sums tested in its shadow; a fragment is free after its
            // group-1 MFMAs and refilled at once, two sub-tiles ahead.  (One wave per SIMD: what is not interleaved with the MFMAs is
            // paid in full — a version that tested after all 48 MFMAs, one scalar branch per accumulator element, took 8.3 ms per
            // stream of 100 M rows against the 5.6 ms the stream itself needs.)
            i32x16_t ah[2], al[2];
#pragma unroll
            for (int e = 0; e < 16; ++e) ah[1][e] = al[1][e] = 0;
            int prev_thr1 = 0x7fffffff;
            uint32_t prev_prow = 0;
            auto tile = [&](u32x4 (&A)[12]) __attribute__((always_inline)) {
                const float2 pmt = meta[t];  // (wave-uniform address: s_load_dwordx2)
                const uint32_t prow = t * 32u;
                const uint32_t t2 = t + 2u * t_stride;
                const uint32_t tn = t2 < n_sub ? t2 : t;
                const u32x4* pn = x + (size_t)tn * (12 * 64) + lane;
                const int thr0 = thr_of(0, pmt), thr1 = thr_of(1, pmt);
                uint32_t bits1p = 0u, bits0 = 0u;
#pragma unroll
                for (int e = 0; e < 16; ++e) ah[0][e] = al[0][e] = 0;
#pragma unroll
                for (int f = 0; f < 12; ++f) {
                    asm volatile("s_waitcnt vmcnt(%1)" : "+a"(A[f]) : "n"(23 - f));
                    const i32x4_t af = __builtin_bit_cast(i32x4_t, A[f]);
                    ah[0] = __builtin_amdgcn_mfma_i32_32x32x32_i8(af, qf[0][f], ah[0], 0, 0, 0);
                    al[0] = __builtin_amdgcn_mfma_i32_32x32x32_i8(af, qf[1][f], al[0], 0, 0, 0);
                    if (f >= 2 && f < 10) post2(ah[1], al[1], prev_thr1, bits1p, 2 * (f - 2));
                    __builtin_amdgcn_sched_barrier(0);
                }
                emit(1, bits1p, prev_prow);
#pragma unroll
                for (int e = 0; e < 16; ++e) ah[1][e] = al[1][e] = 0;
#pragma unroll
                for (int f = 0; f < 12; ++f) {
                    const i32x4_t af = __builtin_bit_cast(i32x4_t, A[f]);
                    ah[1] = __builtin_amdgcn_mfma_i32_32x32x32_i8(af, qf[2][f], ah[1], 0, 0, 0);
                    al[1] = __builtin_amdgcn_mfma_i32_32x32x32_i8(af, qf[3][f], al[1], 0, 0, 0);
                    // (the MFMAs above have read A[f] by the time the data of this load can arrive)
                    asm volatile("global_load_dwordx4 %0, %1, off offset:%2 nt" : "=a"(A[f]) : "v"(pn + (f & ~3) * 64), "n"((f & 3) * 1024));
                    if (f >= 2 && f < 10) post2(ah[0], al[0], thr0, bits0, 2 * (f - 2));
                    __builtin_amdgcn_sched_barrier(0);
                }
                emit(0, bits0, prow);
                prev_thr1 = thr1;
                prev_prow = prow;
                if (n_hit >= (uint32_t)WQ_DRAIN_AT) drain_hits();
                t += t_stride;
            };
            while (true) {
                tile(A0);
                if (t >= n_sub) break;
                tile(A1);
                if (t >= n_sub) break;
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (the refills past the end: nothing of this group stays in flight)
            {  // the last sub-tile's group 1
                uint32_t bits1p = 0u;
#pragma unroll
                for (int e = 0; e < 16; e += 2) post2(ah[1], al[1], prev_thr1, bits1p, e);
                emit(1, bits1p, prev_prow);
            }
        }
        if (n_hit > 0u) drain_hits();
        if (n_ex > 0u) flush_exact();
        __syncthreads();
        // a query some wave ran out of room for: its count goes past the buffer's size, which makes the finish kernel leave it flagged
        if (threadIdx.x < ng && S.lost[threadIdx.x] != 0u) atomicAdd(&res_cnt[S.flagged[g0 + threadIdx.x]], BOUNDED_WIDE_CAP + 1u);
        __syncthreads();  // the shared state is reused by the next group
    }
    if (stats && lane == 0 && n_pairs) atomicAdd(&stats[STAT_BOUNDED_EXACT], n_pairs);
}

// One wave per query of the batch: the pairs the wide form appended -> the k best by (distance, row).  A query that collected fewer
// than k pairs or more than the buffer holds keeps FLAG_FALLBACK (the 16-query form behind this launch answers it); the counters
// go back to zero either way.  The final flag is COUNTED by the launch that closes the search (bounded_count_and_mirror).
template <int DUMMY = 0>
__global__ __launch_bounds__(64) void bounded_wide_finish_kernel(const uint64_t* __restrict__ ids, uint32_t n_rows, int n_q,
                                                                 uint32_t* __restrict__ flags, uint32_t k,
                                                                 const uint2* __restrict__ res, uint32_t* __restrict__ res_cnt,
                                                                 uint64_t* __restrict__ out_labels, float* __restrict__ out_dist,
                                                                 uint32_t* __restrict__ out_found, uint32_t* __restrict__ stats) {
    const uint32_t b = blockIdx.x;
    const int lane = threadIdx.x;
    if ((int)b >= n_q) return;
    const uint32_t cnt = res_cnt[b];
    if (cnt == 0u) return;  // (nothing appended: not a query of the wide form, or one without a usable threshold)
    if (lane == 0) res_cnt[b] = 0u;
    const uint32_t found = n_rows < k ? n_rows : k;
    if (flags[b] != FLAG_FALLBACK || cnt > BOUNDED_WIDE_CAP || cnt < found) return;
    float ls = NEG_INF;
    uint32_t lp = NO_POS;
    const uint2* r = res + (size_t)b * BOUNDED_WIDE_CAP;
    for (uint32_t i0 = 0; i0 < cnt; i0 += 64u) {
        float d = POS_INF;
        uint32_t pr = NO_POS;
        if (i0 + lane < cnt) {
            const uint2 e = r[i0 + lane];
            d = __builtin_bit_cast(float, e.x);
            pr = e.y;
        }
        sort64_asc(d, pr, lane);
        const float os = -__shfl(d, 63 - lane);
        const uint32_t op = __shfl(pr, 63 - lane);
        merge64(ls, lp, os, op, lane);
    }
    if ((uint32_t)lane < found) {
        out_labels[(size_t)b * k + lane] = ids[lp];
        out_dist[(size_t)b * k + lane] = -ls;
    }
    if (lane == 0) {
        out_found[b] = found;
        flags[b] = FLAG_BOUNDED;
        if (stats) atomicAdd(&stats[STAT_BOUNDED_WIDE], 1u);  // (of the FLAG_BOUNDED answers, the ones this form gave)
    }
}

// Per query b < B with d_flags[b] == FLAG_FALLBACK: the exact top-k through the int8 shadow, flag -> FLAG_BOUNDED; every other
// query is left alone (one nearly empty launch when no flag is set).  cand_s / cand_p: the per-workgroup lists [B][n_lists][64]
// (the filter's own, free by now); d_done [B]: arrival counters, zero before and after.  B <= 256 per launch.
// opts (per index): fragments a wave keeps in flight ahead of its MFMAs (6: half a sub-tile, 12: a whole one), waves per workgroup of
// the 16-query batch form (8: 74.0 against 77.4 ms per topical batch of 256 at 100 M rows with 4, profiles/r04/
// bounded_multi_waves_100M.log), that form on the packed 5-bit shadow, and the wide form in front of it (round 5).
void launch_scan_bounded(const void* d_i8, const void* d_i8meta, const void* d_x, int dtype, const uint64_t* d_ids, uint32_t n_rows,
                         const float* d_q, int B, uint32_t* d_flags, uint32_t* d_done, float* cand_s, uint32_t* cand_p,
                         int n_lists, uint32_t k, uint64_t* d_labels, float* d_dist, uint32_t* d_found, hipStream_t stream,
                         uint32_t* d_stats, uint32_t* stats_mirror, const BoundedOpts& opts, const void* d_i5, const void* d_i5meta) {
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
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(scan_bounded_i8_wide_kernel<0>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(BoundedWideLds));
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(scan_bounded_i8_wide_kernel<1>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(BoundedWideLds));
    });
    const u32x4* x8 = reinterpret_cast<const u32x4*>(d_i8);
    const float2* mt = reinterpret_cast<const float2*>(d_i8meta);
    for (int b0 = 0; b0 < B; b0 += kBoundedMaxFlags) {
        const int nb = B - b0 < kBoundedMaxFlags ? B - b0 : kBoundedMaxFlags;
#define DAWN_BOUNDED_ARGS                                                                                                      \
    x8, mt, d_x, d_ids, n_rows, d_q + (size_t)b0 * EM, nb, d_flags + b0, d_done + b0, cand_s + (size_t)b0 * n_lists * LIST,       \
        cand_p + (size_t)b0 * n_lists * LIST, (uint32_t)n_lists, k, d_labels + (size_t)b0 * k, d_dist + (size_t)b0 * k, d_found + b0,   \
        d_stats, stats_mirror
        const bool deep = opts.ring == 12;
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
        } else {       // a batch: its flagged queries, sixty-four per stream of the int8 shadow (wide form), then sixteen per stream
            // (not on a tiny index: with k >= n every row of every query is a result — 64 queries x 32 rows of pairs per wave, more than
            // its queue of pairs awaiting their exact score holds; nothing to win there either)
            if (opts.wide && opts.wide_res && opts.wide_cnt && k <= (uint32_t)LIST && n_rows >= kBoundedWideMinRows) {
                static_assert(sizeof(BoundedWideLds) <= 160 * 1024, "one workgroup per CU: all of its LDS");
                const float* qb = d_q + (size_t)b0 * EM;
                if (dtype == ROW_BF16)
                    hipLaunchKernelGGL((scan_bounded_i8_wide_kernel<1>), dim3(n_lists), dim3(256), sizeof(BoundedWideLds), stream, x8, mt, d_x,
                                       n_rows, qb, nb, d_flags + b0, k, d_dist + (size_t)b0 * k, opts.wide_res, opts.wide_cnt, d_stats);
                else
                    hipLaunchKernelGGL((scan_bounded_i8_wide_kernel<0>), dim3(n_lists), dim3(256), sizeof(BoundedWideLds), stream, x8, mt, d_x,
                                       n_rows, qb, nb, d_flags + b0, k, d_dist + (size_t)b0 * k, opts.wide_res, opts.wide_cnt, d_stats);
                hipLaunchKernelGGL((bounded_wide_finish_kernel<0>), dim3(nb), dim3(64), 0, stream, d_ids, n_rows, nb, d_flags + b0, k,
                                   opts.wide_res, opts.wide_cnt, d_labels + (size_t)b0 * k, d_dist + (size_t)b0 * k, d_found + b0, d_stats);
            }
#define DAWN_BM_LAUNCH(RT_, PD_, NW_)                                                                                                  \
    hipLaunchKernelGGL((scan_bounded_i8_multi_kernel<RT_, PD_, NW_>), dim3(n_lists), dim3(64 * NW_), sizeof(BoundedMultiLds<NW_>), stream, \
                       DAWN_BOUNDED_ARGS)
            const int rt = dtype == ROW_BF16 ? 1 : 0;
            if (d_i5 && d_i5meta && opts.multi_packed) {  // the packed 5-bit shadow (eight waves)
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
            } else if (opts.multi_waves == 8) {
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
                                const BoundedOpts& opts, float first_threshold, const void* d_i5, const void* d_i5meta, bool seeded) {
    const uint32_t found = n_rows < k ? n_rows : k;
    hipLaunchKernelGGL(bounded_prime_kernel, dim3(1), dim3(64), 0, stream, d_flags, d_dist, found, first_threshold, seeded ? 1 : 0);
    if (ev0) (void)hipEventRecord(ev0, stream);
    launch_scan_bounded(d_i8, d_i8meta, d_x, dtype, d_ids, n_rows, d_q, 1, d_flags, d_done, cand_s, cand_p, n_lists, k, d_labels,
                        d_dist, d_found, stream, d_stats, stats_mirror, opts, d_i5, d_i5meta);
    if (ev1) (void)hipEventRecord(ev1, stream);
}

}  // namespace dawn
