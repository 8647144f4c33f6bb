// scan_kernels.hip — brute-force cosine scan + wavefront top-k for gfx950 (MI355X).
//
// Replaces the vector-index search the reference delegates to usearch (src/search/search_provider.rs:214)
// with the EXACT computation that search approximates: distance = 1 - sum(q_i*x_i)
// (src/search/vector.rs:128-134), ascending, ties -> earlier-added row.
//
// Structure:
//   1. scan_filter_kernel   HBM-bound stream over the packed [N][384] f32 index.  Coalesced 16-B/lane
//                           non-temporal loads, 2 rows per 3 wave-loads, DPP wave reduction, per-wave
//                           sorted top-64 list held one entry per lane (insertion only on a threshold
//                           hit), per-block bitonic merge.  Scores here are f32 in a tree order.
//   2. merge_rescore_kernel merges the per-block lists, recomputes the 64 survivors' distances in the
//                           reference's sequential un-fused f32 order (bit-for-bit), sorts by
//                           (distance, row), and certifies that no row outside the shortlist can reach
//                           the top-k (rigorous bound FILTER_EPS_F32).
//   3. scan_exact_kernel (its last workgroup merges): the always-exact (slower, lane-per-row) pass, run only for
//                           queries whose certificate failed (predicated on a device flag: no host
//                           round trip).
//
// Compiled with -ffp-contract=off: every fused multiply-add below is an explicit __builtin_fmaf, and
// the exact paths use separate multiply and add as the reference (Rust) does.
#include "kernels.hpp"
#ifdef DAWN_EXPERIMENTS
// timestamp probes (100-MHz counter) of workgroup 0, thread 0: merge_rescore_kernel's phases (dawn_debug_read_ts)
static __device__ unsigned long long dawn_ts[16];
#define DAWN_TS(i)                                                                       \
    do {                                                                                 \
        if (threadIdx.x == 0 && blockIdx.x == 0) dawn_ts[i] = __builtin_amdgcn_s_memrealtime(); \
    } while (0)
#endif
#include "wave_topk.hpp"

namespace dawn {

__device__ __forceinline__ float dot4_fma(const f32x4& a, const f32x4& b, float acc) {
    acc = __builtin_fmaf(a.x, b.x, acc);
    acc = __builtin_fmaf(a.y, b.y, acc);
    acc = __builtin_fmaf(a.z, b.z, acc);
    acc = __builtin_fmaf(a.w, b.w, acc);
    return acc;
}


// ------------------------------------------------------------------------------------------------
// 1. filter scan
// ------------------------------------------------------------------------------------------------
// Work unit of a wave: a "pair" = 2 consecutive rows = 3072 contiguous bytes = 3 wave-wide 16-B loads.
//   load 0: row 0, f32x4 chunks 0..63         load 1: lanes 0..31 -> row 0 chunks 64..95,
//   load 2: row 1, chunks 32..95                        lanes 32..63 -> row 1 chunks 0..31
// A wave handles U pairs per iteration (U*3 KiB in flight) and strides over the index with all other
// waves of the grid, so that at any moment the chip reads one contiguous window of HBM.
template <int QB, int U>
__global__ __launch_bounds__(1024) void scan_filter_kernel(const f32x4* __restrict__ x, uint32_t n_rows,
                                                           const float* __restrict__ q,
                                                           float* __restrict__ out_s,
                                                           uint32_t* __restrict__ out_p, uint32_t q_stride_lists,
                                                           uint32_t* __restrict__ pool) {
    __shared__ float sh_s[16][LIST];
    __shared__ uint32_t sh_p[16][LIST];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nwaves = blockDim.x >> 6;
    const uint32_t gwave = blockIdx.x * nwaves + wave;
    const uint32_t total_waves = gridDim.x * nwaves;
    const uint32_t n_pairs = (n_rows + 1u) >> 1;
    const uint32_t n_chunks = (n_pairs + U - 1) / U;  // iterations ("chunks" of U row pairs) in all

    f32x4 qa[QB], qb[QB], qc[QB];
    float ls[QB], tau[QB];
    uint32_t lp[QB];
#pragma unroll
    for (int b = 0; b < QB; ++b) {
        const f32x4* qq = reinterpret_cast<const f32x4*>(q + b * EM);
        qa[b] = qq[lane];
        qb[b] = qq[lane < 32 ? 64 + lane : lane - 32];
        qc[b] = qq[32 + lane];
        ls[b] = NEG_INF;
        lp[b] = NO_POS;
        tau[b] = NEG_INF;
    }
    const bool lo_half = lane < 32;

    auto process = [&](uint32_t c) __attribute__((always_inline)) {
        const f32x4* p = x + (size_t)c * (U * 192) + lane;
        f32x4 v[U][3];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            v[u][0] = nt_load(p + u * 192);
            v[u][1] = nt_load(p + u * 192 + 64);
            v[u][2] = nt_load(p + u * 192 + 128);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint32_t r0 = (c * U + u) * 2u;
            const uint32_t r1 = r0 + 1u;
#pragma unroll
            for (int b = 0; b < QB; ++b) {
                const float d1 = dot4_fma(v[u][1], qb[b], 0.f);
                float p0 = dot4_fma(v[u][0], qa[b], lo_half ? d1 : 0.f);
                float p1 = dot4_fma(v[u][2], qc[b], lo_half ? 0.f : d1);
                float s0 = read_lane63(wave_sum_lane63(p0));
                float s1 = read_lane63(wave_sum_lane63(p1));
                // rows past the end (padding) and non-finite garbage never enter a list
                s0 = (r0 < n_rows && s0 == s0) ? s0 : NEG_INF;
                s1 = (r1 < n_rows && s1 == s1) ? s1 : NEG_INF;
                if (s0 > tau[b]) {
                    wave_insert(ls[b], lp[b], s0, r0, lane);
                    tau[b] = read_lane63(ls[b]);
                }
                if (s1 > tau[b]) {
                    wave_insert(ls[b], lp[b], s1, r1, lane);
                    tau[b] = read_lane63(ls[b]);
                }
            }
        }
    };
    // Static and interleaved (iteration gwave, gwave + W, ...) — and, in a long single-query launch (pool != NULL), only for the
    // first 7/8 of the index: the waves' shares are equal, their speeds are not (scan_i6.hip: the first wave of a 100 M-row stream
    // is done ~300 us before the last), so the last eighth is handed out on demand in blocks of 16 consecutive iterations, each
    // fetched with a scalar atomic (returns through lgkmcnt: the loads in flight are not drained) from one of 32 counters —
    // one per group of eight workgroups, i.e. per set of one CU on every XCD; merge_rescore_kernel leaves them at zero.
    constexpr uint32_t DCH = 16, NONE = 0xFFFFFFFFu;
    const uint32_t rounds = n_chunks / total_waves;
    if (pool != nullptr && rounds >= 32u) {
        const uint32_t i_static = rounds - rounds / 8u, d0 = total_waves * i_static;
        const uint32_t n_blocks = (n_chunks - d0 + DCH - 1u) / DCH;
        const uint32_t n_pools = ((gridDim.x + 7u) >> 3) < 32u ? ((gridDim.x + 7u) >> 3) : 32u;
        const uint32_t pool_id = (blockIdx.x >> 3) % n_pools;
        uint32_t* my_pool = pool + pool_id;
        for (uint32_t i = 0; i < i_static; ++i) process(gwave + i * total_waves);
        for (;;) {
            uint32_t v = 1u;
            asm volatile("s_atomic_add %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "+s"(v) : "s"(my_pool));
            const uint32_t j = pool_id + n_pools * v;
            if (j >= n_blocks) break;
            const uint32_t c0 = d0 + j * DCH;
            for (uint32_t i = 0; i < DCH && c0 + i < n_chunks; ++i) process(c0 + i);
        }
        (void)NONE;
    } else {
        for (uint32_t c = gwave; c < n_chunks; c += total_waves) process(c);
    }

#pragma unroll
    for (int b = 0; b < QB; ++b) {
        block_merge(ls[b], lp[b], sh_s, sh_p, wave, lane, nwaves);
        if (wave == 0) {
            const size_t o = ((size_t)b * q_stride_lists + blockIdx.x) * LIST + lane;
            out_s[o] = ls[b];
            out_p[o] = lp[b];
        }
    }
}

// f16 SHADOW of an f32 index (ROW_F16S, kernels.hpp: f16(2^8 x), 768 B per row, tiles in MFMA-fragment order) — the
// streaming filter for 1..8 queries.  A wave-wide 16-B load of a 1-KiB fragment IS the A operand of
// v_mfma_f32_32x32x16_f16 (32 rows x 16 k), so the rows go HBM -> VGPR -> matrix core with no LDS, no cross-lane
// reduction and almost no VALU work: the queries sit in the B operand's first columns (96 VGPRs for the whole kernel,
// zero columns beyond B), and after 24 k-steps lane (h, c) holds the scores of query c against 16 rows of the
// 32-row sub-tile.  One compare against the lane's threshold (the 64th best score of query c in this wave so far)
// decides whether the slow path runs: ballot, read the hit, insert it into the wave's sorted 64-entry list.
// Loads run PD fragments (PD KiB per wave) ahead of the MFMAs in a register ring that continues across sub-tiles.
// Half the bytes of the f32 rows; exactness comes from the rescore tail on the f32 rows (error bound FILTER_EPS_F16).
//
// BF16 = true: the same kernel over a bf16 INDEX (ROW_BF16, rows unscaled) on v_mfma_f32_32x32x16_bf16.  The rows are
// exact; the query enters twice — column c holds hi = bf16(q), column 8 + c holds lo = bf16(q - hi) — and a lane adds
// the two accumulators (DPP row_shl:8) before the test: the filter then errs like an f32 one (FILTER_EPS_BF16_STREAM),
// for 16 extra VALU instructions per 32 rows.
typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x16_t __attribute__((ext_vector_type(16)));

template <bool BF16>
__device__ __forceinline__ f32x16_t mfma_16bit(const u32x4& a, const u32x4& b, const f32x16_t& c) {
    if (BF16)
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8_t, a), __builtin_bit_cast(half8_t, b), c, 0, 0, 0);
}

// f32 -> the 16-bit operand element: f16(2^8 v) for the shadow, bf16 hi / lo part for a bf16 index
__device__ __forceinline__ uint32_t f16s_bits(float v) {
    return (uint32_t)__builtin_bit_cast(unsigned short, (_Float16)(v * 256.0f));
}
__device__ __forceinline__ uint32_t bf16_part(float v, bool lo_part) {
    const uint32_t hi = f32_to_bf16_rne(v);
    if (!lo_part) return hi;
    return f32_to_bf16_rne(v - __builtin_bit_cast(float, hi << 16));  // exact difference, then rounded
}

// q: the n_q <= QB queries, f32 [n_q][384], converted here (f16 path: exactly as prep_queries_kernel of
// scan_batched.hip does for the matrix-core path)
template <int QB, int PD, bool BURST, bool BF16>
__global__ __launch_bounds__(512) void scan_filter_f16s_kernel(const u32x4* __restrict__ x, uint32_t n_rows,
                                                                const float* __restrict__ q, int n_q,
                                                                float* __restrict__ out_s,
                                                                uint32_t* __restrict__ out_p,
                                                                uint32_t q_stride_lists) {
    static_assert(24 % PD == 0, "the ring must divide the 24 k-steps of a sub-tile");
    __shared__ float sh_s[16][LIST];
    __shared__ uint32_t sh_p[16][LIST];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nwaves = blockDim.x >> 6;
    const uint32_t gwave = blockIdx.x * nwaves + wave;
    const uint32_t total_waves = gridDim.x * nwaves;
    const uint32_t n_sub = (n_rows + 31u) >> 5;  // 32-row sub-tiles = 24 fragments = 24 KiB each
    const uint32_t c = lane & 31, h = lane >> 5;

    // B operand: column c = query c (zero beyond n_q; BF16: columns 8.. = the lo parts), lane (h, c) holds
    // k = 16s + 8h .. +7 of k-step s
    u32x4 qf[24];
#pragma unroll
    for (int s = 0; s < 24; ++s) qf[s] = u32x4{0u, 0u, 0u, 0u};
    const int qcol = BF16 ? (int)(c & 7u) : (int)c;       // the query this column belongs to
    const bool lo_part = BF16 && c >= 8;
    if (qcol < n_q && (BF16 ? c < 16 : true)) {
        const f32x4* qc = reinterpret_cast<const f32x4*>(q + (size_t)qcol * EM);
#pragma unroll
        for (int s = 0; s < 24; ++s) {
            const f32x4 v0 = qc[4 * s + 2 * h], v1 = qc[4 * s + 2 * h + 1];
            if (BF16) {
                qf[s].x = bf16_part(v0.x, lo_part) | (bf16_part(v0.y, lo_part) << 16);
                qf[s].y = bf16_part(v0.z, lo_part) | (bf16_part(v0.w, lo_part) << 16);
                qf[s].z = bf16_part(v1.x, lo_part) | (bf16_part(v1.y, lo_part) << 16);
                qf[s].w = bf16_part(v1.z, lo_part) | (bf16_part(v1.w, lo_part) << 16);
            } else {
                qf[s].x = f16s_bits(v0.x) | (f16s_bits(v0.y) << 16);
                qf[s].y = f16s_bits(v0.z) | (f16s_bits(v0.w) << 16);
                qf[s].z = f16s_bits(v1.x) | (f16s_bits(v1.y) << 16);
                qf[s].w = f16s_bits(v1.z) | (f16s_bits(v1.w) << 16);
            }
        }
    }
    float ls[QB], tau[QB];
    uint32_t lp[QB];
#pragma unroll
    for (int b = 0; b < QB; ++b) {
        ls[b] = NEG_INF;
        lp[b] = NO_POS;
        tau[b] = NEG_INF;
    }
    const float unscale = BF16 ? 1.0f : 1.0f / 65536.0f;
    // the lane's threshold in the units of the accumulator (f16: scores x 2^16); +inf in the padding columns
    float tau_l = (int)c < n_q ? NEG_INF : __builtin_inff();

    uint32_t t = gwave;
    if (t < n_sub) {
        const u32x4* p = x + (size_t)t * (24 * 64) + lane;
        u32x4 a[PD];
#pragma unroll
        for (int d = 0; d < PD; ++d) a[d] = __builtin_nontemporal_load(p + d * 64);
        for (;;) {
            const uint32_t tn = t + total_waves;
            // the ring runs into the wave's next sub-tile (the last one re-reads its own first fragments: no branch)
            const u32x4* pn = tn < n_sub ? x + (size_t)tn * (24 * 64) + lane : p;
            f32x16_t acc;
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[e] = 0.f;
            if (BURST) {
                // PD fragments (PD KiB, contiguous) requested back to back, consumed as they arrive, then the next PD
#pragma unroll
                for (int s0 = 0; s0 < 24; s0 += PD) {
                    if (s0 > 0) {
#pragma unroll
                        for (int d = 0; d < PD; ++d) a[d] = __builtin_nontemporal_load(p + (s0 + d) * 64);
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int d = 0; d < PD; ++d)
                        acc = mfma_16bit<BF16>(a[d], qf[s0 + d], acc);
                    __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
                for (int d = 0; d < PD; ++d) a[d] = __builtin_nontemporal_load(pn + d * 64);
                __builtin_amdgcn_sched_barrier(0);
            } else {
#pragma unroll
                for (int s = 0; s < 24; ++s) {
                    acc = mfma_16bit<BF16>(a[s % PD], qf[s], acc);
                    if (s + PD < 24) a[s % PD] = __builtin_nontemporal_load(p + (s + PD) * 64);
                    else a[s % PD] = __builtin_nontemporal_load(pn + (s + PD - 24) * 64);
                    __builtin_amdgcn_sched_barrier(0);  // keep every load PD steps ahead of its use
                }
            }
            // this lane: D[row = 32t + (e&3) + 8*(e>>2) + 4h][query c]
            if (BF16) {  // hi + lo columns: lane c <- lane c + 8 (same 16-lane row)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const float ae = acc[e];
                    acc[e] = ae + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, ae), 0x108,
                                                                                        0xf, 0xf, true));
                }
            }
            float mx = acc[0];
#pragma unroll
            for (int e = 1; e < 16; ++e) mx = fmaxf(mx, acc[e]);
            if (__any(mx > tau_l)) {
                const uint32_t row_base = t * 32u;
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const uint32_t roff = (uint32_t)((e & 3) + 8 * (e >> 2));
                    // (copied to a scalar first: __builtin_bit_cast straight on the vector element acc[e] makes
                    // hipcc 7.2 read element 0 — the same miscompile as in dot8 of the bf16 kernel's sibling)
                    const float ae = acc[e];
                    // rows past the end (zero padding) and non-finite garbage never enter a list
                    unsigned long long m = __ballot(ae > tau_l && row_base + roff + 4u * h < n_rows);
                    while (m) {
                        const int l = __builtin_ctzll(m);
                        m &= m - 1;
                        const float sc = __builtin_bit_cast(
                            float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, ae), l)) * unscale;
                        const int qb = l & 31;
                        const uint32_t row = row_base + roff + 4u * (uint32_t)(l >> 5);
#pragma unroll
                        for (int b = 0; b < QB; ++b) {
                            if (b == qb && sc > tau[b]) {
                                wave_insert(ls[b], lp[b], sc, row, lane);
                                tau[b] = read_lane63(ls[b]);
                            }
                        }
                    }
                }
#pragma unroll
                for (int b = 0; b < QB; ++b)
                    if ((int)c == b && b < n_q) tau_l = tau[b] * (BF16 ? 1.0f : 65536.0f);
            }
            if (tn >= n_sub) break;
            t = tn;
            p = pn;
        }
    }

#pragma unroll
    for (int b = 0; b < QB; ++b) {
        if (b < n_q) {  // uniform
            block_merge(ls[b], lp[b], sh_s, sh_p, wave, lane, nwaves);
            if (wave == 0) {
                const size_t o = ((size_t)b * q_stride_lists + blockIdx.x) * LIST + lane;
                out_s[o] = ls[b];
                out_p[o] = lp[b];
            }
        }
    }
}

template <int QB, bool BF16>
static void launch_filter_f16s_qb(const void* d_shadow, uint32_t n_rows, const float* q8, int n_q, float* cand_s,
                                  uint32_t* cand_p, const ScanGeom& g, hipStream_t stream) {
    const u32x4* x8 = reinterpret_cast<const u32x4*>(d_shadow);
    // geom.unroll picks the load schedule: 1..4 -> ring of 6 / 8 / 12 / 24 fragments running ahead of the MFMAs;
    // 11..14 -> bursts of 6 / 8 / 12 / 24 fragments (KiB per wave) requested back to back
#define DAWN_F16S_LAUNCH(PD_, BURST_)                                                                              \
    hipLaunchKernelGGL((scan_filter_f16s_kernel<QB, PD_, BURST_, BF16>), dim3(g.blocks), dim3(g.threads), 0, stream, x8, \
                       n_rows, q8, n_q, cand_s, cand_p, (uint32_t)g.blocks)
    switch (g.unroll) {
        case 1: DAWN_F16S_LAUNCH(6, false); break;
        case 2: DAWN_F16S_LAUNCH(8, false); break;
        case 4: DAWN_F16S_LAUNCH(24, false); break;
        case 11: DAWN_F16S_LAUNCH(6, true); break;
        case 12: DAWN_F16S_LAUNCH(8, true); break;
        case 13: DAWN_F16S_LAUNCH(12, true); break;
        case 14: DAWN_F16S_LAUNCH(24, true); break;
        default: DAWN_F16S_LAUNCH(12, false); break;
    }
#undef DAWN_F16S_LAUNCH
}

// Streaming filter over fragment-ordered 16-bit rows (rt = ROW_F16S: the f16 shadow of an f32 index; ROW_BF16: a bf16
// index), 8 queries per pass; d_q = the f32 queries [B][384].
void launch_scan_filter_f16s(const void* d_shadow, int rt, uint32_t n_rows, const float* d_q, int B, float* cand_s,
                             uint32_t* cand_p, const ScanGeom& g, hipStream_t stream, hipEvent_t ev0, hipEvent_t ev1) {
    if (ev0) (void)hipEventRecord(ev0, stream);
    const size_t per_q = (size_t)g.blocks * LIST;
    for (int b = 0; b < B; b += 8) {  // 8 queries per pass over the index
        const int nb = B - b < 8 ? B - b : 8;
        const float* q = d_q + (size_t)b * EM;
        float* cs = cand_s + (size_t)b * per_q;
        uint32_t* cp = cand_p + (size_t)b * per_q;
        if (rt == ROW_BF16) {
            if (nb <= 1) launch_filter_f16s_qb<1, true>(d_shadow, n_rows, q, nb, cs, cp, g, stream);
            else if (nb <= 4) launch_filter_f16s_qb<4, true>(d_shadow, n_rows, q, nb, cs, cp, g, stream);
            else launch_filter_f16s_qb<8, true>(d_shadow, n_rows, q, nb, cs, cp, g, stream);
        } else {
            if (nb <= 1) launch_filter_f16s_qb<1, false>(d_shadow, n_rows, q, nb, cs, cp, g, stream);
            else if (nb <= 4) launch_filter_f16s_qb<4, false>(d_shadow, n_rows, q, nb, cs, cp, g, stream);
            else launch_filter_f16s_qb<8, false>(d_shadow, n_rows, q, nb, cs, cp, g, stream);
        }
    }
    if (ev1) (void)hipEventRecord(ev1, stream);
}

template <int QB, int U>
static void launch_filter_qbu(const float* d_x, uint32_t n_rows, const float* d_q, float* cand_s, uint32_t* cand_p,
                              const ScanGeom& g, hipStream_t stream, uint32_t* pool) {
    hipLaunchKernelGGL((scan_filter_kernel<QB, U>), dim3(g.blocks), dim3(g.threads), 0, stream,
                       reinterpret_cast<const f32x4*>(d_x), n_rows, d_q, cand_s, cand_p, (uint32_t)g.blocks, pool);
}

template <int QB>
static void launch_filter_qb(const float* d_x, uint32_t n_rows, const float* d_q, float* cand_s, uint32_t* cand_p,
                             const ScanGeom& g, hipStream_t stream, uint32_t* pool) {
    if (QB == 1 && g.unroll == 1) launch_filter_qbu<1, 1>(d_x, n_rows, d_q, cand_s, cand_p, g, stream, pool);
    else if (QB == 1 && g.unroll == 3) launch_filter_qbu<1, 3>(d_x, n_rows, d_q, cand_s, cand_p, g, stream, pool);
    else if (QB == 1 && g.unroll == 4) launch_filter_qbu<1, 4>(d_x, n_rows, d_q, cand_s, cand_p, g, stream, pool);
    else launch_filter_qbu<QB, 2>(d_x, n_rows, d_q, cand_s, cand_p, g, stream, pool);
}

// f32 rows streamed directly (no shadow): 1 / 2 / 4 queries per pass
// pool: 32 zeroed counters for the dynamically assigned tail of a single-query launch (left at zero again by
// merge_rescore_kernel), or NULL: every iteration statically assigned; batches of several launches are always static
void launch_scan_filter(const void* d_xv, int dtype, uint32_t n_rows, const float* d_q, int B, float* cand_s,
                        uint32_t* cand_p, const ScanGeom& g, hipStream_t stream, hipEvent_t ev0, hipEvent_t ev1, uint32_t* pool) {
    if (B != 1) pool = nullptr;
    (void)dtype;  // ROW_F32 (a bf16 index streams through launch_scan_filter_f16s)
    const float* d_x = reinterpret_cast<const float*>(d_xv);
    if (ev0) (void)hipEventRecord(ev0, stream);
    int b = 0;
    const size_t per_q = (size_t)g.blocks * LIST;
    while (b < B) {
        const int rem = B - b;
        const float* q = d_q + (size_t)b * EM;
        float* cs = cand_s + (size_t)b * per_q;
        uint32_t* cp = cand_p + (size_t)b * per_q;
        if (rem >= 4) {
            launch_filter_qb<4>(d_x, n_rows, q, cs, cp, g, stream, pool);
            b += 4;
        } else if (rem >= 2) {
            launch_filter_qb<2>(d_x, n_rows, q, cs, cp, g, stream, pool);
            b += 2;
        } else {
            launch_filter_qb<1>(d_x, n_rows, q, cs, cp, g, stream, pool);
            b += 1;
        }
    }
    if (ev1) (void)hipEventRecord(ev1, stream);
}

// ------------------------------------------------------------------------------------------------
// 2. merge + exact rescore + certificate
// ------------------------------------------------------------------------------------------------
template <int RT>
__global__ __launch_bounds__(1024) void merge_rescore_kernel(
    const void* __restrict__ x, const uint64_t* __restrict__ ids, uint32_t n_rows, const float* __restrict__ q,
    const float* __restrict__ cand_s, const uint32_t* __restrict__ cand_p, int n_lists, uint32_t k,
    uint64_t* __restrict__ out_labels, float* __restrict__ out_dist, uint32_t* __restrict__ out_found,
    uint32_t* __restrict__ out_flags, int force_fallback, float eps, uint32_t* __restrict__ pool,
    const float* __restrict__ list_bounds, uint32_t* __restrict__ stats_packed) {
    __shared__ float sh_s[16][LIST];
    __shared__ uint32_t sh_p[16][LIST];
    __shared__ uint32_t sh_rows[LIST];
    extern __shared__ __attribute__((aligned(16))) unsigned char rescore_stage[];  // RescoreStage<RT>::BYTES
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nwaves = blockDim.x >> 6;
    const int b = blockIdx.x;
    if (pool != nullptr && b == 0 && threadIdx.x < 32) pool[threadIdx.x] = 0;  // the stream's work counters, for the next search

    const float* cs = cand_s + (size_t)b * n_lists * LIST;
    const uint32_t* cp = cand_p + (size_t)b * n_lists * LIST;
    __shared__ uint32_t sh_ctl[4];
    __shared__ float sh_t[16];
    DAWN_TS(0);
    const float q_val = threadIdx.x < EM ? q[(size_t)b * EM + threadIdx.x] : 0.f;  // (used after the selection: see block_exact_dots)
    // rows in no list scored <= T = the largest 64th entry of any list (round 1: the merged 64th entry is >= T anyway); found
    // by the first selection from the entries it loads anyway (lane 0 of a reversed list holds its 64th entry)
    float T = NEG_INF;
    // the next 64 candidates by filter score among the per-workgroup lists
    auto select = [&](bool first, float ex_s, uint32_t ex_p, float& s, uint32_t& p) {
        s = NEG_INF;
        p = NO_POS;
        float t0 = NEG_INF;
        constexpr int INF = 16;  // lists in flight per wave: the usual 256 lists are ONE round of loads
        for (int l0 = wave; l0 < n_lists; l0 += INF * nwaves) {
            float os[INF];
            uint32_t op[INF];
#pragma unroll
            for (int j = 0; j < INF; ++j) {
                const int l = l0 + j * nwaves;
                os[j] = l < n_lists ? cs[(size_t)l * LIST + 63 - lane] : NEG_INF;
                op[j] = l < n_lists ? cp[(size_t)l * LIST + 63 - lane] : NO_POS;
            }
#pragma unroll
            for (int j = 0; j < INF; ++j) {
                if (l0 + j * nwaves >= n_lists) continue;  // wave-uniform
                if (first) t0 = fmaxf(t0, __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, os[j]))));
                if (first && list_bounds) t0 = fmaxf(t0, list_bounds[l0 + j * nwaves]);
                if (!first) {
                    // entries down to the last one already rescored drop out: they are a prefix of the (descending) list,
                    // so the survivors are sorted again with the fillers moved behind them
                    float d = POS_INF;
                    uint32_t row = NO_POS;
                    if (op[j] != NO_POS && better(ex_s, ex_p, os[j], op[j])) {
                        d = -os[j];
                        row = op[j];
                    }
                    sort64_asc(d, row, lane);
                    os[j] = -__shfl(d, 63 - lane);
                    op[j] = __shfl(row, 63 - lane);
                }
                merge64(s, p, os[j], op[j], lane);
            }
        }
        if (first && lane == 0) sh_t[wave] = t0;  // (visible after block_merge's barriers)
        DAWN_TS(1);
        block_merge(s, p, sh_s, sh_p, wave, lane, nwaves);
        DAWN_TS(2);
        if (first)
            for (int w = 0; w < nwaves; ++w) T = fmaxf(T, sh_t[w]);
    };
    const uint32_t found = n_rows < k ? n_rows : k;
    float bs;
    uint32_t bp;
    bool heavy;
    const uint32_t flag = certify_rounds<RT>(select, T, true, n_rows, found, eps, force_fallback, q_val, x,
                                             rescore_stage, sh_rows, sh_ctl, wave, lane, bs, bp, heavy);
    // (the ladder feedback of an index watches how often the packed stream's queries end up flagged: dawn_index.cpp; a forced
    // flag is not a failure; a flag the second chance lifts below is taken back there)
    if (stats_packed && threadIdx.x == 0 && flag == FLAG_FALLBACK && !force_fallback) atomicAdd(&stats_packed[STAT_PACKED_FAIL], 1u);
    if (wave == 0) {
        if ((uint32_t)lane < found) {
            // (slots without a candidate read as "no threshold" to the ladder behind a flag: scan_bounded.hip)
            out_labels[(size_t)b * k + lane] = bp != NO_POS ? ids[bp] : 0ull;
            out_dist[(size_t)b * k + lane] = bp != NO_POS ? -bs : POS_INF;
        }
        if (lane == 0) {
            out_found[b] = found;
            out_flags[b] = flag;
        }
    }
    DAWN_TS(7);
    if (!heavy) return;

    // ---- second chance (wave_topk.hpp): the union of the workgroup lists holds every row whose filter score exceeds T;
    // its 1024 best are rescored exactly
    __syncthreads();
    auto load = [&](uint32_t e, float& sc, uint32_t& row) {
        sc = cs[e];
        row = cp[e];
        return row != NO_POS;
    };
    float s2;
    uint32_t p2;
    const bool ok = second_chance<RT>(load, (uint32_t)n_lists * LIST, T, q + (size_t)b * EM, x, found, eps, rescore_stage, sh_s,
                                      sh_p, wave, lane, s2, p2);
    if (wave == 0 && ok) {
        if ((uint32_t)lane < found) {
            out_labels[(size_t)b * k + lane] = ids[p2];
            out_dist[(size_t)b * k + lane] = -s2;
        }
        if (lane == 0) {
            out_flags[b] = FLAG_SECOND;
            if (stats_packed && !force_fallback) atomicSub(&stats_packed[STAT_PACKED_FAIL], 1u);
        }
    }
}

void launch_merge_rescore(const void* d_x, int dtype, const uint64_t* d_ids, uint32_t n_rows, const float* d_q, int B,
                          const float* cand_s, const uint32_t* cand_p, int n_lists, uint32_t k, uint64_t* d_labels,
                          float* d_dist, uint32_t* d_found, uint32_t* d_flags, int force_fallback, float eps,
                          hipStream_t stream, uint32_t* pool, const float* list_bounds, uint32_t* d_stats_packed) {
    static OncePerDevice attr_once;  // the f32 stage (97 KiB) is above the default dynamic-LDS limit
    once_per_device(attr_once, [] {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(merge_rescore_kernel<0>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, RescoreStage<0>::BYTES);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(merge_rescore_kernel<1>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, RescoreStage<1>::BYTES);
    });
    if (dtype == ROW_BF16)
        hipLaunchKernelGGL(merge_rescore_kernel<1>, dim3(B), dim3(1024), RescoreStage<1>::BYTES, stream, d_x, d_ids, n_rows,
                           d_q, cand_s, cand_p, n_lists, k, d_labels, d_dist, d_found, d_flags, force_fallback, eps, pool,
                           list_bounds, d_stats_packed);
    else
        hipLaunchKernelGGL(merge_rescore_kernel<0>, dim3(B), dim3(1024), RescoreStage<0>::BYTES, stream, d_x, d_ids, n_rows,
                           d_q, cand_s, cand_p, n_lists, k, d_labels, d_dist, d_found, d_flags, force_fallback, eps, pool,
                           list_bounds, d_stats_packed);
}

// ------------------------------------------------------------------------------------------------
// 3. exact fallback: lane-per-row, reference summation order, key = -distance
// ------------------------------------------------------------------------------------------------
// The last workgroup to finish a query merges the per-workgroup lists and writes the result (done[b]: arrival counter,
// zero between searches): the exact pass is ONE launch, and when no flag is set — every search on ordinary data — it
// costs one empty kernel instead of two.
constexpr int kMaxFlags = 256;
template <int RT>
__global__ __launch_bounds__(256) void scan_exact_kernel(const void* __restrict__ x, const uint64_t* __restrict__ ids,
                                                        uint32_t n_rows, const float* __restrict__ q, int n_q,
                                                        const uint32_t* __restrict__ flags, uint32_t* __restrict__ done,
                                                        uint32_t* __restrict__ stats,
                                                        float* __restrict__ out_s, uint32_t* __restrict__ out_p,
                                                        uint32_t n_lists, uint32_t k, uint64_t* __restrict__ out_labels,
                                                        float* __restrict__ out_dist, uint32_t* __restrict__ out_found,
                                                        uint32_t* __restrict__ mirror) {
    __shared__ float sh_s[4][LIST];
    __shared__ uint32_t sh_p[4][LIST];
    __shared__ uint32_t sh_last;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nwaves = blockDim.x >> 6;
    const uint32_t gwave = blockIdx.x * nwaves + wave;
    const uint32_t total_waves = gridDim.x * nwaves;
    // The batch's flags, one per thread, in ONE round trip (n_q <= kMaxFlags = blockDim: the search paths pass at most 256
    // queries per launch); the flagged queries come out of wave ballots as bit masks and only those are visited.  (Round 2
    // read a slot's flags one after the other — 16 dependent loads per workgroup, 4096 workgroups: 12.8 us behind every
    // 256-query search; a serial walk over 256 LDS flags was worse, 44 us.)
    __shared__ unsigned long long sh_mask[kMaxFlags / 64];
    {
        const uint32_t myflag = (int)threadIdx.x < n_q ? flags[threadIdx.x] : FLAG_OK;
        // certificate statistics of the index (dawn_index_stats*): this kernel closes every search and sees every query's
        // final flag, so the counters also cover searches issued through dawn_index_search_device
        if (stats && blockIdx.x == 0 && myflag != FLAG_OK) atomicAdd(&stats[myflag], 1u);
        const unsigned long long m = __ballot(myflag == FLAG_FALLBACK);
        if (lane == 0) sh_mask[wave] = m;
    }
    __syncthreads();
    // ... and a copy of the counters goes to the host (pinned, device-visible memory: posted stores, nobody waits) — what the
    // index's ladder feedback reads without ever synchronising (dawn_index.cpp)
    if (mirror && stats && blockIdx.x == 0 && threadIdx.x < (unsigned)N_STAT_SLOTS) mirror[threadIdx.x] = atomicAdd(&stats[threadIdx.x], 0u);
    for (int w = 0; w < kMaxFlags / 64; ++w) {
      unsigned long long todo = sh_mask[w];  // block-uniform
      while (todo) {
        const int b = w * 64 + __builtin_ctzll(todo);
        todo &= todo - 1;
        const float* qv = q + (size_t)b * EM;

        float ls = NEG_INF, tau = NEG_INF;
        uint32_t lp = NO_POS;
        const uint32_t n_groups = (n_rows + 63u) >> 6;
        for (uint32_t g = gwave; g < n_groups; g += total_waves) {
            const uint32_t r = g * 64u + lane;
            float key = NEG_INF;
            if (r < n_rows) {
                const float dot = exact_dot_row<RT>(qv, x, r);
                const float d = __fsub_rn(1.0f, dot);
                key = (d == d) ? -d : NEG_INF;
            }
            // rows of this group arrive in ascending row order; insert the lanes that beat the threshold
            unsigned long long hits = __ballot(key > tau);
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
            uint32_t p = NO_POS;
            const float* cs = out_s + (size_t)b * n_lists * LIST;
            const uint32_t* cp = out_p + (size_t)b * n_lists * LIST;
            for (uint32_t l = wave; l < n_lists; l += nwaves)
                merge64(s, p, cs[(size_t)l * LIST + 63 - lane], cp[(size_t)l * LIST + 63 - lane], lane);
            block_merge(s, p, sh_s, sh_p, wave, lane, nwaves);
            if (wave == 0) {
                const uint32_t found = n_rows < k ? n_rows : k;
                if ((uint32_t)lane < found) {
                    out_labels[(size_t)b * k + lane] = ids[p];
                    out_dist[(size_t)b * k + lane] = -s;
                }
                if (lane == 0) {
                    out_found[b] = found;
                    done[b] = 0u;
                }
            }
        }
        __syncthreads();  // sh_s / sh_p / sh_last are reused by the next query
      }
    }
}

void launch_scan_exact(const void* d_x, int dtype, const uint64_t* d_ids, uint32_t n_rows, const float* d_q, int B,
                       const uint32_t* d_flags, uint32_t* d_done, uint32_t* d_stats, float* cand_s, uint32_t* cand_p,
                       int n_lists, uint32_t k, uint64_t* d_labels, float* d_dist, uint32_t* d_found, hipStream_t stream,
                       uint32_t* stats_mirror) {
    // one workgroup per list; it walks the batch's flags (LDS, one round trip) and scans for the flagged queries one after the
    // other.  (Round 2 launched 16 query slots side by side: 4096 workgroups whose launch alone took 12 us behind every 256-query
    // search, flags set or not; a flagged query's scan is HBM-bound with 256 workgroups, so the slots bought nothing.)
    const dim3 grid(n_lists, 1);
    if (dtype == ROW_BF16)
        hipLaunchKernelGGL(scan_exact_kernel<1>, grid, dim3(256), 0, stream, d_x, d_ids, n_rows, d_q, B, d_flags, d_done, d_stats,
                           cand_s, cand_p, (uint32_t)n_lists, k, d_labels, d_dist, d_found, stats_mirror);
    else
        hipLaunchKernelGGL(scan_exact_kernel<0>, grid, dim3(256), 0, stream, d_x, d_ids, n_rows, d_q, B, d_flags, d_done, d_stats,
                           cand_s, cand_p, (uint32_t)n_lists, k, d_labels, d_dist, d_found, stats_mirror);
}

// ------------------------------------------------------------------------------------------------
// multi-GPU: stable G-way merge of per-shard (distance-ascending) results
// ------------------------------------------------------------------------------------------------
// Shard g's lists start at in_labels + g*sl, in_dist + g*sd, in_found + g*sf (element strides): separate
// [G][B][k] arrays (sl = sd = B*k, sf = B) or one packed blob per shard as all-gathered by the ranks.
// Ties: lower shard, then shard-local order (contiguous row shards: that IS insertion order).  POS = true (the
// single-process sharded index of dawn_sharded.cpp, whose rows are dealt to the shards chunk by chunk): the "labels" coming
// in are global insertion positions, ties go to the lower position, and pos_to_label[] turns the winners into labels.
// Any G * k fits: a thread owns candidates t, t + 512, ... (one each up to G * k = 512, the usual case — 8 GPUs x k <= 64);
// LDS is sized by the launch (G * k * 4 B of distances, + G * k * 8 B of positions when POS).
template <bool POS>
__global__ __launch_bounds__(512) void shard_merge_kernel(uint32_t G, uint32_t B, uint32_t k,
                                                         const uint64_t* __restrict__ in_labels,
                                                         const float* __restrict__ in_dist,
                                                         const uint32_t* __restrict__ in_found, size_t sl, size_t sd,
                                                         size_t sf, const uint64_t* __restrict__ pos_to_label,
                                                         uint64_t* __restrict__ out_labels,
                                                         float* __restrict__ out_dist,
                                                         uint32_t* __restrict__ out_found) {
    extern __shared__ __attribute__((aligned(16))) unsigned char merge_lds[];
    const uint32_t total = G * k;
    uint64_t* sh_l = reinterpret_cast<uint64_t*>(merge_lds);                      // [POS ? total : 0]
    float* sh_d = reinterpret_cast<float*>(merge_lds + (POS ? (size_t)total * 8 : 0));  // [total]
    const uint32_t b = blockIdx.x;
    for (uint32_t c = threadIdx.x; c < total; c += blockDim.x) {
        const uint32_t g = c / k, i = c % k;
        const bool valid = i < in_found[g * sf + b];
        sh_d[c] = valid ? in_dist[g * sd + (size_t)b * k + i] : POS_INF;
        if (POS) sh_l[c] = valid ? in_labels[g * sl + (size_t)b * k + i] : ~0ull;
    }
    __syncthreads();
    for (uint32_t c = threadIdx.x; c < total; c += blockDim.x) {
        const uint32_t g = c / k, i = c % k;
        if (i >= in_found[g * sf + b]) continue;
        const float d = sh_d[c];
        const uint64_t label = POS ? sh_l[c] : in_labels[g * sl + (size_t)b * k + i];
        // rank = number of candidates ordered before (d, g, i); c = g*k+i is that lexicographic index
        uint32_t rank = 0;
        for (uint32_t o = 0; o < total; ++o) {
            const float od = sh_d[o];
            const bool before = POS ? sh_l[o] < label : o < c;
            rank += (od < d || (od == d && before)) ? 1u : 0u;
        }
        if (rank < k) {
            out_labels[(size_t)b * k + rank] = POS ? pos_to_label[label] : label;
            out_dist[(size_t)b * k + rank] = d;
        }
    }
    if (threadIdx.x == 0) {
        uint32_t sum = 0;
        for (uint32_t gg = 0; gg < G; ++gg) sum += in_found[gg * sf + b];
        out_found[b] = sum < k ? sum : k;
    }
}

void launch_shard_merge(size_t G, size_t B, size_t k, const uint64_t* in_labels, const float* in_dist,
                        const uint32_t* in_found, size_t sl, size_t sd, size_t sf, const uint64_t* pos_to_label,
                        uint64_t* out_labels, float* out_dist, uint32_t* out_found, hipStream_t stream) {
    const size_t total = G * k;  // callers bound it: <= kMaxMergeCands (64 shards x DAWN_MAX_K)
    if (pos_to_label)
        hipLaunchKernelGGL(shard_merge_kernel<true>, dim3((unsigned)B), dim3(512), total * 12, stream, (uint32_t)G, (uint32_t)B,
                           (uint32_t)k, in_labels, in_dist, in_found, sl, sd, sf, pos_to_label, out_labels, out_dist, out_found);
    else
        hipLaunchKernelGGL(shard_merge_kernel<false>, dim3((unsigned)B), dim3(512), total * 4, stream, (uint32_t)G, (uint32_t)B,
                           (uint32_t)k, in_labels, in_dist, in_found, sl, sd, sf, pos_to_label, out_labels, out_dist, out_found);
}

// ------------------------------------------------------------------------------------------------
// validation + synthetic generator
// ------------------------------------------------------------------------------------------------
// vector.rs:181-192: l = sqrt(sum((v_i - 0)^2)) sequential; finite and 0.99 < l < 1.01
__global__ void validate_rows_kernel(const float* __restrict__ rows, uint32_t n, uint32_t* __restrict__ bad) {
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    const float* v = rows + (size_t)r * EM;
    float s = 0.0f;
    for (int i = 0; i < EM; ++i) {
        const float dlt = __fsub_rn(v[i], 0.0f);
        s = __fadd_rn(s, __fmul_rn(dlt, dlt));
    }
    const float l = sqrtf(s);  // correctly rounded (HIP default)
    const bool ok = __builtin_isfinite(l) && l > (1.0f - 0.01f) && l < (1.0f + 0.01f);
    if (!ok) atomicAdd(bad, 1u);
}

// src/index/warc.rs:35-43 PageEntry records (1568 B: u64 url_pos, u64 title_pos, f32 vector[384], u64 url_len, u64
// title_len) as they sit in an .emb file -> packed f32 rows.  The records arrive by DMA exactly as on disk; the vector of
// record r is the 96 16-B chunks starting at byte 1568 r + 16 (16-B aligned: 1568 = 98 x 16).
__global__ void page_entries_to_rows_kernel(const f32x4* __restrict__ rec, uint32_t n, f32x4* __restrict__ rows) {
    const size_t total = (size_t)n * ROW_F4;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t r = i / ROW_F4, c = i % ROW_F4;
        rows[i] = rec[r * 98 + 1 + c];
    }
}

void launch_page_entries_to_rows(const void* d_records, uint32_t n, float* d_rows, hipStream_t stream) {
    if (n == 0) return;
    size_t blocks = ((size_t)n * ROW_F4 + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(page_entries_to_rows_kernel, dim3((unsigned)blocks), dim3(256), 0, stream,
                       reinterpret_cast<const f32x4*>(d_records), n, reinterpret_cast<f32x4*>(d_rows));
}

void launch_validate_rows(const float* d_rows, uint32_t n, uint32_t* d_bad_count, hipStream_t stream) {
    if (n == 0) return;
    hipLaunchKernelGGL(validate_rows_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, d_rows, n, d_bad_count);
}

__device__ __forceinline__ uint64_t splitmix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ULL;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

__device__ __forceinline__ float synth_uniform(uint64_t key, uint64_t idx) {
    const uint64_t h = splitmix64(key + idx * 0x9E3779B97F4A7C15ULL);
    const int u = (int)(h >> 40);
    const int nn = 2 * u + 1 - (1 << 24);
    return __fmul_rn((float)nn, 1.0f / 16777216.0f);
}

// Bench / test distributions other than the spec's uniform rows (option "synth_dist"; device-generated, NOT restated on the
// CPU: the bench legs that use them check planted rows fetched back from the index, not an oracle scan):
//   1  Gaussian components (Box-Muller over two uniforms of the stream): what a random rotation makes of any embedding
//   2  heavy-tailed, fixed dimensions: Gaussian with dimensions {7, 101, 213, 340} scaled x5 — sentence-embedding models
//      have a handful of dimensions that are large in every vector
//   3  heavy-tailed, per-row dimensions: Gaussian with 4 pseudo-random dimensions per row scaled x5
//   4  topical mixture — the one distribution besides 0 that IS restated on the CPU (dawnsearch_amd/synth.py:
//      unit_rows_topical, oracle: orc_synth_topical_row; integer hashing and single f32 operations in a fixed order):
//      Zipf-sized clusters (12 octaves, 4095 clusters: cluster j of octave o holds 1 / (12 * 2^o) of the rows) around
//      bell-shaped centroids, row = normalise(centroid + t_j * noise) with the cosine between two rows of a cluster
//      0.5 ... 0.95 — dense semantic neighbourhoods, what the filters' slack is measured against
//   5  the same with runs of 256 consecutive rows per cluster: the pages of one site are inserted back to back
//      (src/index/warc.rs:75-86, src/search/search_provider.rs:250-286)
__device__ __forceinline__ float synth_g4(uint64_t key, uint64_t i) {
    float g = __fadd_rn(synth_uniform(key, 4 * i), synth_uniform(key, 4 * i + 1));
    g = __fadd_rn(g, synth_uniform(key, 4 * i + 2));
    g = __fadd_rn(g, synth_uniform(key, 4 * i + 3));
    return __fmul_rn(g, 0.8660254f);
}
template <int DIST>
__device__ __forceinline__ float synth_value(uint64_t key, uint64_t row, uint32_t col) {
    const uint64_t idx = row * (uint64_t)EM + col;
    if (DIST == 0) return synth_uniform(key, idx);
    if (DIST == 4 || DIST == 5) {
        const uint64_t unit = DIST == 5 ? row >> 8 : row;
        const uint64_t h = splitmix64(key ^ (unit * 0xD1B54A32D192ED03ULL) ^ 0x746F706963730001ULL);
        const uint32_t o = (uint32_t)((h >> 32) % 12u);
        const uint32_t j = ((1u << o) - 1u) + ((uint32_t)h & ((1u << o) - 1u));
        const uint64_t hj = splitmix64(key ^ ((uint64_t)j * 0xD6E8FEB86659FD93ULL) ^ 0x746F706963730002ULL);
        const uint32_t lv = (uint32_t)((hj >> 20) % 6u);
        const float t = lv == 0 ? 1.0f : lv == 1 ? 0.8164966f : lv == 2 ? 0.6546537f : lv == 3 ? 0.5f : lv == 4 ? 0.33333334f : 0.22941573f;
        const uint64_t ckey = splitmix64(key ^ 0x746F706963730003ULL);
        const float cen = synth_g4(ckey, (uint64_t)j * EM + col);
        return __fadd_rn(cen, __fmul_rn(t, synth_g4(key, idx)));
    }
    const float u1 = synth_uniform(key, 2 * idx), u2 = synth_uniform(key, 2 * idx + 1);
    const float a = 0.5f * u1 + 0.5f;  // (0, 1)
    float g = sqrtf(-2.0f * __logf(a)) * __cosf(3.14159265f * u2);
    if (DIST == 2 && (col == 7u || col == 101u || col == 213u || col == 340u)) g *= 5.0f;
    if (DIST == 3) {
        const uint64_t hsh = splitmix64(key ^ (row * 0xD1B54A32D192ED03ULL));
        if (col == (uint32_t)(hsh % EM) || col == (uint32_t)((hsh >> 16) % EM) || col == (uint32_t)((hsh >> 32) % EM) ||
            col == (uint32_t)((hsh >> 48) % EM))
            g *= 5.0f;
    }
    return g;
}

// pass 1: per-row length, sequential sum of squares (vector.rs:195)
template <int DIST>
__global__ void synth_len_kernel(uint64_t key, uint64_t first_row, uint32_t n, float* __restrict__ len) {
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    float s = 0.0f;
    for (int c = 0; c < EM; ++c) {
        const float v = synth_value<DIST>(key, first_row + r, (uint32_t)c);
        s = __fadd_rn(s, __fmul_rn(v, v));
    }
    len[r] = sqrtf(s);
}

// pass 2: coalesced write of v / len (vector.rs:196)
template <int DIST>
__global__ void synth_write_kernel(uint64_t key, uint64_t first_row, uint32_t n, const float* __restrict__ len,
                                   f32x4* __restrict__ out) {
    const size_t total = (size_t)n * ROW_F4;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const uint32_t r = (uint32_t)(i / ROW_F4);
        const uint32_t c = (uint32_t)(i % ROW_F4);
        const float l = len[r];
        f32x4 v;
        v.x = synth_value<DIST>(key, first_row + r, c * 4u + 0u) / l;
        v.y = synth_value<DIST>(key, first_row + r, c * 4u + 1u) / l;
        v.z = synth_value<DIST>(key, first_row + r, c * 4u + 2u) / l;
        v.w = synth_value<DIST>(key, first_row + r, c * 4u + 3u) / l;
        out[i] = v;
    }
}

// f32 rows -> bf16 rows (round to nearest even), 8 values per thread; and back (exact widening)
// staged f32 rows [n][384] -> rows first_row.. of the fragment-ordered bf16 index (round to nearest even).  One block
// per 64 staged rows; consecutive threads take consecutive rows of one 16-B chunk column: 512-B contiguous writes.
__global__ __launch_bounds__(256) void rows_f32_to_bf16_kernel(const f32x4* __restrict__ in, u32x4* __restrict__ x,
                                                              size_t first_row, size_t n_rows) {
    const size_t r0 = (size_t)blockIdx.x * 64;
    for (int idx = threadIdx.x; idx < 64 * ROW_C8; idx += 256) {
        const size_t r = r0 + (idx & 63);
        const int c = idx >> 6;
        if (r >= n_rows) continue;
        const f32x4 a = in[r * ROW_F4 + 2 * c], b = in[r * ROW_F4 + 2 * c + 1];
        u32x4 w;
        w.x = f32_to_bf16_rne(a.x) | (f32_to_bf16_rne(a.y) << 16);
        w.y = f32_to_bf16_rne(a.z) | (f32_to_bf16_rne(a.w) << 16);
        w.z = f32_to_bf16_rne(b.x) | (f32_to_bf16_rne(b.y) << 16);
        w.w = f32_to_bf16_rne(b.z) | (f32_to_bf16_rne(b.w) << 16);
        x[frag_chunk(first_row + r, c)] = w;
    }
}

// ... and back: rows first_row.. of the index -> f32 rows [n][384] (exact widening)
__global__ __launch_bounds__(256) void rows_bf16_to_f32_kernel(const u32x4* __restrict__ x, size_t first_row,
                                                              f32x4* __restrict__ out, size_t n_rows) {
    const size_t r0 = (size_t)blockIdx.x * 64;
    for (int idx = threadIdx.x; idx < 64 * ROW_C8; idx += 256) {
        const size_t r = r0 + (idx & 63);
        const int c = idx >> 6;
        if (r >= n_rows) continue;
        const u32x4 w = x[frag_chunk(first_row + r, c)];
        out[r * ROW_F4 + 2 * c] = f32x4{bf16_lo(w.x), bf16_hi(w.x), bf16_lo(w.y), bf16_hi(w.y)};
        out[r * ROW_F4 + 2 * c + 1] = f32x4{bf16_lo(w.z), bf16_hi(w.z), bf16_lo(w.w), bf16_hi(w.w)};
    }
}

void launch_rows_f32_to_bf16(const float* d_in, void* d_x, size_t first_row, size_t n_rows, hipStream_t stream) {
    if (n_rows == 0) return;
    hipLaunchKernelGGL(rows_f32_to_bf16_kernel, dim3((unsigned)((n_rows + 63) / 64)), dim3(256), 0, stream,
                       reinterpret_cast<const f32x4*>(d_in), reinterpret_cast<u32x4*>(d_x), first_row, n_rows);
}

void launch_rows_bf16_to_f32(const void* d_x, size_t first_row, float* d_out, size_t n_rows, hipStream_t stream) {
    if (n_rows == 0) return;
    hipLaunchKernelGGL(rows_bf16_to_f32_kernel, dim3((unsigned)((n_rows + 63) / 64)), dim3(256), 0, stream,
                       reinterpret_cast<const u32x4*>(d_x), first_row, reinterpret_cast<f32x4*>(d_out), n_rows);
}

static uint64_t host_splitmix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ULL;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

template <int DIST>
static void fill_synth_dist(uint64_t key, uint64_t first_row, uint32_t n, float* d_out, float* d_len, hipStream_t stream) {
    hipLaunchKernelGGL(synth_len_kernel<DIST>, dim3((n + 255) / 256), dim3(256), 0, stream, key, first_row, n, d_len);
    const size_t total = (size_t)n * ROW_F4;
    size_t blocks = (total + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(synth_write_kernel<DIST>, dim3((unsigned)blocks), dim3(256), 0, stream, key, first_row, n, d_len,
                       reinterpret_cast<f32x4*>(d_out));
}

void launch_fill_synth(uint64_t seed, uint64_t first_row, uint32_t n, float* d_out, float* d_len, int dist,
                       hipStream_t stream) {
    if (n == 0) return;
    const uint64_t key = host_splitmix64(seed);
    switch (dist) {
        case 1: fill_synth_dist<1>(key, first_row, n, d_out, d_len, stream); break;
        case 2: fill_synth_dist<2>(key, first_row, n, d_out, d_len, stream); break;
        case 3: fill_synth_dist<3>(key, first_row, n, d_out, d_len, stream); break;
        case 4: fill_synth_dist<4>(key, first_row, n, d_out, d_len, stream); break;
        case 5: fill_synth_dist<5>(key, first_row, n, d_out, d_len, stream); break;
        default: fill_synth_dist<0>(key, first_row, n, d_out, d_len, stream); break;
    }
}

__global__ void iota_u64_kernel(uint64_t* out, uint64_t first, uint32_t n) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = first + i;
}

void launch_iota_u64(uint64_t* d_out, uint64_t first, uint32_t n, hipStream_t stream) {
    if (n == 0) return;
    hipLaunchKernelGGL(iota_u64_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, d_out, first, n);
}

}  // namespace dawn

#ifdef DAWN_EXPERIMENTS
extern "C" __attribute__((visibility("default"))) int dawn_debug_read_ts(unsigned long long* out, int n) {
    unsigned long long h[16];
    if (hipMemcpyFromSymbol(h, HIP_SYMBOL(dawn_ts), sizeof(h)) != hipSuccess) return -1;
    for (int i = 0; i < n && i < 16; ++i) out[i] = h[i];
    return 0;
}
#endif
