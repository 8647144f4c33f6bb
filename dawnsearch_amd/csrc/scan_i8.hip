// scan_i8.hip — int8 FILTER shadow of an f32 index (ROW_I8S, kernels.hpp) and its streaming filter for 1..8 queries.
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
//     this expression (< 5e-7) and the reference's own sequential-sum error gamma_384 * 1.0201 = 2.4e-5:
//     FILTER_EPS_I8 = 2.6e-5 on top of ub.
//   * the hot loop never leaves the integers: the lane's list threshold tau is turned into an integer threshold once
//     per sub-tile (5 VALU), conservatively (rows it passes are re-tested on ub itself).
#include "kernels.hpp"
#include "wave_topk.hpp"

namespace dawn {

typedef int i32x4_t __attribute__((ext_vector_type(4)));
typedef int i32x16_t __attribute__((ext_vector_type(16)));

constexpr float I8_QRES = 2.0e-3f;                   // |dq_i| <= I8_QRES * s_q
constexpr float I8_K2_PER_SQ = 1.1f * 19.6f * I8_QRES;  // K2 = I8_K2_PER_SQ * s_q  (sqrt(384) < 19.6)

// ------------------------------------------------------------------------------------------------
// conversion: f32 rows -> int8 sub-tiles + {s, E} per sub-tile
// ------------------------------------------------------------------------------------------------
// One 256-thread block per sub-tile; thread = (row r = tid / 8, part = tid % 8) handles float4 chunks part + 8j.
__global__ __launch_bounds__(256) void rows_f32_to_i8s_kernel(const f32x4* __restrict__ x, uint32_t* __restrict__ out,
                                                               float2* __restrict__ meta, uint32_t first_sub,
                                                               uint32_t n_valid) {
    __shared__ float sh[4];
    const uint32_t sub = first_sub + blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t r = tid >> 3, part = tid & 7;
    const uint32_t row = sub * 32u + r;
    f32x4 v[12];
#pragma unroll
    for (int j = 0; j < 12; ++j)
        v[j] = row < n_valid ? x[(size_t)row * ROW_F4 + part + 8 * j] : f32x4{0.f, 0.f, 0.f, 0.f};
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
    const float s = fmaxf(amax, 1e-20f) / 127.0f;
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
            t = fminf(fmaxf(t, -127.f), 127.f);
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
        meta[sub] = float2{s, sqrtf(m) * 1.0101f * 1.001f + 1e-9f};
    }
}

void launch_rows_f32_to_i8s(const float* d_rows, void* d_shadow, void* d_meta, size_t first_row, size_t n_valid,
                            hipStream_t stream) {
    const uint32_t first_sub = (uint32_t)(first_row / 32);  // the sub-tile holding first_row is re-quantised whole
    const uint32_t end_sub = (uint32_t)((n_valid + 31) / 32);
    if (end_sub <= first_sub) return;
    hipLaunchKernelGGL(rows_f32_to_i8s_kernel, dim3(end_sub - first_sub), dim3(256), 0, stream,
                       reinterpret_cast<const f32x4*>(d_rows), reinterpret_cast<uint32_t*>(d_shadow),
                       reinterpret_cast<float2*>(d_meta), first_sub, (uint32_t)n_valid);
}

// ------------------------------------------------------------------------------------------------
// streaming filter
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int pack4_i8(int a, int b, int c, int d) {
    return (a & 255) | ((b & 255) << 8) | ((c & 255) << 16) | (int)((uint32_t)(d & 255) << 24);
}

// Lists hold ub (see the header) as the score.  q: the n_q <= QB <= 8 queries, f32 [n_q][384].
template <int QB, int PD>
__global__ __launch_bounds__(512) void scan_filter_i8s_kernel(const u32x4* __restrict__ x, const float2* __restrict__ meta,
                                                               uint32_t n_rows, const float* __restrict__ q, int n_q,
                                                               float* __restrict__ out_s, uint32_t* __restrict__ out_p,
                                                               uint32_t q_stride_lists) {
    static_assert(12 % PD == 0, "the ring must divide the 12 k-steps of a sub-tile");
    __shared__ float sh_s[16][LIST];
    __shared__ uint32_t sh_p[16][LIST];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nwaves = blockDim.x >> 6;
    const uint32_t gwave = blockIdx.x * nwaves + wave;
    const uint32_t total_waves = gridDim.x * nwaves;
    const uint32_t n_sub = (n_rows + 31u) >> 5;
    const uint32_t c = lane & 31, h = lane >> 5;

    // B operand: column c < 8 = H of query c, column 8 + c = L; lane (h, c) holds k = 32f + 16h .. +15 of k-step f
    i32x4_t qf[12];
#pragma unroll
    for (int f = 0; f < 12; ++f) qf[f] = i32x4_t{0, 0, 0, 0};
    const int qcol = (int)(c & 7u);
    const bool lo_part = c >= 8;
    float sq254_l = 0.f, k2_l = 0.f;  // this lane's query: s_q / 254 and K2
    if (qcol < n_q && c < 16) {
        const f32x4* qc = reinterpret_cast<const f32x4*>(q + (size_t)qcol * EM);
        float amax = 0.f;
        for (int i = 0; i < ROW_F4; ++i) {
            const f32x4 v = qc[i];
            amax = fmaxf(amax, fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
        }
        const float sq = fmaxf(amax, 1e-20f) / 127.0f;
        sq254_l = sq / 254.0f;
        k2_l = I8_K2_PER_SQ * sq;
#pragma unroll
        for (int f = 0; f < 12; ++f) {
            int w[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const f32x4 v = qc[8 * f + 4 * h + j];
                const float vv[4] = {v.x, v.y, v.z, v.w};
                int b4[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float t = vv[i] / sq;
                    const float H = fminf(fmaxf(rintf(t), -127.f), 127.f);
                    const float L = fminf(fmaxf(rintf((t - H) * 254.0f), -127.f), 127.f);
                    b4[i] = (int)(lo_part ? L : H);
                }
                w[j] = pack4_i8(b4[0], b4[1], b4[2], b4[3]);
            }
            qf[f] = i32x4_t{w[0], w[1], w[2], w[3]};
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

    uint32_t t = gwave;
    if (t < n_sub) {
        const u32x4* p = x + (size_t)t * (12 * 64) + lane;
        u32x4 a[PD];
#pragma unroll
        for (int d = 0; d < PD; ++d) a[d] = __builtin_nontemporal_load(p + d * 64);
        float2 mt = meta[t];
        for (;;) {
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
                if (f + PD < 12) a[f % PD] = __builtin_nontemporal_load(p + (f + PD) * 64);
                else a[f % PD] = __builtin_nontemporal_load(pn + (f + PD - 12) * 64);
                __builtin_amdgcn_sched_barrier(0);  // keep every load PD steps ahead of its use
            }
            // this lane: D[row = 32t + (e&3) + 8*(e>>2) + 4h][column c];  C = 254 acc_H + acc_L (lane c <- lane c + 8)
            const float g_l = mt.x * sq254_l;                       // score units per unit of C
            const float u = __builtin_fmaf(-mt.y, 1.000001f, tau_m);
            float thr_f = __builtin_fmaf(u, __builtin_amdgcn_rcpf(g_l), -2.0f);
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
                                const float sc = __builtin_fmaf(cf, mt.x * sq254[b], mt.y + k2[b]);
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
            }
            if (!more) break;
            t = tn;
            p = pn;
            mt = mtn;
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

}  // namespace dawn
