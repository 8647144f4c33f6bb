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
//   * ub = fma(float(C), s * s_q / 254, E + K2) >= x.q as before; K2 uses ||s X||_2 <= ||x||_2 + ||dx||_2 < 1.01 + 0.32
//     (||dx||_2 <= sqrt(384) * s / 2, s <= 1.01 / 31: the worst case of a sub-tile that holds a one-hot row in the rotated
//     basis; the int8 shadow's 0.09 does not carry over).
// E is four times the int8 shadow's (~0.037 on unit vectors), which a 64-row shortlist cannot absorb (the gap between the
// 10th and the 64th best score of 100 M rows is 0.017; tools/coarse_shadow_probe.py).  The stream therefore does not hand a
// 64-row shortlist to a one-workgroup tail: EVERY workgroup rescores its own 64 best rows exactly (reference order,
// src/search/vector.rs:128-134) in its epilogue — 256 workgroups x 64 rows = a shortlist 16 384 rows deep for ~8 us —, and
// merge_exact_kernel merges the exact lists and checks ONE certificate: rows in no list have ub <= T = the largest 64th
// upper bound of any workgroup, i.e. distance >= fl(1 - up(T + eps)); if that exceeds the k-th exact distance strictly
// the result is exact, otherwise the query takes the exact pass (scan_exact_kernel) like any failed certificate.
#include <type_traits>

#include "kernels.hpp"
#include "rotate384.hpp"
#include "wave_topk.hpp"

namespace dawn {

typedef int i32x16_t __attribute__((ext_vector_type(16)));
typedef uint32_t u32x3 __attribute__((ext_vector_type(3)));
typedef u32x3 u32x3_u __attribute__((aligned(4)));  // (fragments are 12-B lane slots: 4-B aligned)

constexpr float I6_LEVELS = 31.0f;
constexpr uint32_t I6_FRAG_DW = 64 * 3;            // dwords per fragment
constexpr uint32_t I6_SUB_DW = 12 * I6_FRAG_DW;    // dwords per sub-tile (9216 B)
constexpr float I6_K2_PER_SQ = 1.35f * 19.6f * I8_QRES;

// ------------------------------------------------------------------------------------------------
// conversion: rows -> 6-bit sub-tiles + {1 / s, E} per sub-tile (rows_to_i8s_kernel with the packing above)
// ------------------------------------------------------------------------------------------------
template <int RT>
__global__ __launch_bounds__(256) void rows_to_i6s_kernel(const void* __restrict__ xv, uint32_t* __restrict__ out,
                                                           float2* __restrict__ meta, uint32_t first_sub, uint32_t n_valid) {
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
    __syncthreads();
    const float s = fmaxf(amax, 1e-20f) / I6_LEVELS;
    float e2 = 0.f;
    uint32_t* o = out + (size_t)sub * I6_SUB_DW;
    const uint32_t h = (part >> 2) & 1u, dj = part & 3u;
#pragma unroll
    for (int j = 0; j < 12; ++j) {  // float4 chunk part + 8 j = dword dj of lane (h, r) of fragment j
        uint32_t w = 0;
        const float vv[4] = {v[j].x, v[j].y, v[j].z, v[j].w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float t = rintf(vv[i] / s);
            t = fminf(fmaxf(t, -I6_LEVELS), I6_LEVELS);
            const float dx = vv[i] - s * t;
            e2 = __builtin_fmaf(dx, dx, e2);
            w |= (uint32_t)((int)t + 32) << (8 * i);
        }
        // the fourth dword of the lane (values 12..15) is spread over the top two bits of the other three
        const uint32_t w3 = __shfl(w, lane | 3);
        if (dj < 3u) o[j * I6_FRAG_DW + (h * 32u + r) * 3u + dj] = w | (((w3 >> (2u * dj)) & 0x03030303u) << 6);
    }
    e2 += __shfl_xor(e2, 1);
    e2 += __shfl_xor(e2, 2);
    e2 += __shfl_xor(e2, 4);
#pragma unroll
    for (int o2 = 32; o2 >= 8; o2 >>= 1) e2 = fmaxf(e2, __shfl_xor(e2, o2));
    if (lane == 0) sh[wave] = e2;
    __syncthreads();
    if (tid == 0) {
        const float m = fmaxf(fmaxf(sh[0], sh[1]), fmaxf(sh[2], sh[3]));
        meta[sub] = float2{1.0f / s, sqrtf(m) * 1.0101f * 1.001f + 1e-9f};  // as rows_to_i8s_kernel
    }
}

void launch_rows_to_i6s(const void* d_rows, int rt, void* d_shadow, void* d_meta, size_t first_row, size_t n_valid,
                        hipStream_t stream) {
    const uint32_t first_sub = (uint32_t)(first_row / 32);  // the sub-tile holding first_row is re-quantised whole
    const uint32_t end_sub = (uint32_t)((n_valid + 31) / 32);
    if (end_sub <= first_sub) return;
    if (rt == ROW_BF16)
        hipLaunchKernelGGL(rows_to_i6s_kernel<1>, dim3(end_sub - first_sub), dim3(256), 0, stream, d_rows,
                           reinterpret_cast<uint32_t*>(d_shadow), reinterpret_cast<float2*>(d_meta), first_sub, (uint32_t)n_valid);
    else
        hipLaunchKernelGGL(rows_to_i6s_kernel<0>, dim3(end_sub - first_sub), dim3(256), 0, stream, d_rows,
                           reinterpret_cast<uint32_t*>(d_shadow), reinterpret_cast<float2*>(d_meta), first_sub, (uint32_t)n_valid);
}

// ------------------------------------------------------------------------------------------------
// the stream: scan_filter_i8s_pipe_kernel (software-pipelined test, two accumulator sets) for ONE query on 6-bit
// fragments, + the exact rescore of the workgroup's shortlist as its epilogue
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ u32x3 frag_load(const uint32_t* p) {
    return __builtin_nontemporal_load(reinterpret_cast<const u32x3_u*>(p));
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

// RT: row type of the index (0 f32, 1 bf16) for the exact rescore; PD: fragments in flight per wave (ring)
template <int RT, int PD>
__global__ __launch_bounds__(512) void scan_filter_i6s_kernel(const uint32_t* __restrict__ x, const float2* __restrict__ meta,
                                                               uint32_t n_rows, const float* __restrict__ q,
                                                               const void* __restrict__ rows, float* __restrict__ out_s,
                                                               uint32_t* __restrict__ out_p, float* __restrict__ out_es,
                                                               uint32_t* __restrict__ out_ep) {
    static_assert(12 % PD == 0, "the ring must divide the 12 k-steps of a sub-tile");
    __shared__ float sh_s[8][LIST];
    __shared__ uint32_t sh_p[8][LIST];
    __shared__ uint32_t sh_rows[LIST];
    extern __shared__ __attribute__((aligned(16))) unsigned char rescore_stage[];  // RescoreStage<RT>::BYTES
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nwaves = blockDim.x >> 6;
    const uint32_t n_sub = (n_rows + 31u) >> 5;
    const uint32_t c = lane & 31, h = lane >> 5;
    uint32_t t = blockIdx.x * nwaves + wave;
    const uint32_t t_stride = gridDim.x * nwaves, t_end = n_sub;
    // the first fragments fly while the query images are made
    const uint32_t* p = x + (size_t)(t < t_end ? t : 0) * I6_SUB_DW + lane * 3;
    u32x3 a[PD];
    float2 mt = {0.f, 0.f};
    if (t < t_end) {
#pragma unroll
        for (int d = 0; d < PD; ++d) a[d] = frag_load(p + d * I6_FRAG_DW);
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
    const float sq254 = sq / 254.0f, rsq254 = 254.0f / sq, k2 = I6_K2_PER_SQ * sq;
    const int acc0 = col_live ? -32 * sh_sum[c == 8 ? 1 : 0] : 0;  // codes are value + 32
    float ls = NEG_INF, tau = NEG_INF;
    uint32_t lp = NO_POS;
    const bool tested = c == 0;
    float tau_m = tested ? NEG_INF : __builtin_inff();

    if (t < t_end) {
        i32x16_t accs[2];
        float2 pmt = mt;
        uint32_t prow = 0;
        int C[16];
        int thr = 0, mx = 0;

        auto slow_path = [&]() __attribute__((always_inline)) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const uint32_t roff = (uint32_t)((e & 3) + 8 * (e >> 2));
                unsigned long long m = __ballot(C[e] > thr && prow + roff + 4u * h < n_rows);
                while (m) {
                    const int l = __builtin_ctzll(m);
                    m &= m - 1;
                    const float cf = (float)__builtin_amdgcn_readlane(C[e], l);
                    const uint32_t row = prow + roff + 4u * (uint32_t)(l >> 5);
                    const float sc = __builtin_fmaf(cf, __builtin_amdgcn_rcpf(pmt.x) * sq254, pmt.y + k2);
                    if (sc > tau) {
                        wave_insert(ls, lp, sc, row, lane);
                        tau = read_lane63(ls);
                    }
                }
            }
            if (tested) {
                const float tk = tau - k2;
                tau_m = tk - fabsf(tk) * 1e-6f;
            }
        };
        auto test_slice = [&](int s, const i32x16_t& pacc) __attribute__((always_inline)) {
            if (s == 0) {
                const float u = __builtin_fmaf(-pmt.y, 1.000001f, tau_m);
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
            more = tn < t_end;
            const uint32_t* pn = more ? x + (size_t)tn * I6_SUB_DW + lane * 3 : p;
            const float2 mtn = meta[more ? tn : t];
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[e] = acc0;
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
                // (the reload BEHIND the MFMA: in front of it — the slot is free once unpacked — rings of 3-4 fragments lose 1-2 %,
                // rings of 12 gain nothing: profiles/r03/stream_i6_parts_off_100M.log, DBG = 8)
                if (f + PD < 12) a[f % PD] = frag_load(p + (f + PD) * I6_FRAG_DW);
                else a[f % PD] = frag_load(pn + (f + PD - 12) * I6_FRAG_DW);
                if constexpr (decltype(with_test)::value) test_slice(f, accs[1 - P]);
                __builtin_amdgcn_sched_barrier(0);
            }
            if constexpr (decltype(with_test)::value)
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

    block_merge_any(ls, lp, sh_s, sh_p, wave, lane, nwaves);
    const size_t o = (size_t)blockIdx.x * LIST + lane;
    if (wave == 0) {
        out_s[o] = ls;  // upper bounds, descending: lane 63 = the bound on every row of this workgroup that is not listed
        out_p[o] = lp;
        sh_rows[lane] = lp;
    }
    // ---- epilogue: the workgroup's 64 rows, exactly (block_exact_dots of wave_topk.hpp for any block size)
    typedef RescoreStage<RT> S;
    float* sh_q = reinterpret_cast<float*>(rescore_stage + S::ROWS_BYTES);
    for (int i = threadIdx.x; i < EM; i += blockDim.x) sh_q[i] = q[i];
    __syncthreads();
    const u32x4* xr = reinterpret_cast<const u32x4*>(rows);
    for (int i = threadIdx.x; i < LIST * S::CH; i += blockDim.x) {
        const int r = i / S::CH, ch = i % S::CH;
        const uint32_t row = sh_rows[r];
        if (row != NO_POS)
            *reinterpret_cast<u32x4*>(rescore_stage + r * S::STRIDE + ch * 16) =
                RT == 1 ? xr[frag_chunk(row, ch)] : xr[(size_t)row * S::CH + ch];
    }
    __syncthreads();
    if (wave == 0) {
        float d = POS_INF;
        uint32_t pr = lp;
        if (lp != NO_POS) {
            float dot;
            if (RT == 1) dot = exact_dot_seq_bf16<8>(sh_q, reinterpret_cast<const u32x4*>(rescore_stage + lane * S::STRIDE));
            else dot = exact_dot_seq<16>(sh_q, reinterpret_cast<const f32x4*>(rescore_stage + lane * S::STRIDE));
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
}

// ------------------------------------------------------------------------------------------------
// merge of the exact per-workgroup lists + the certificate
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void merge_exact_kernel(const uint64_t* __restrict__ ids, uint32_t n_rows,
                                                            const float* __restrict__ ub_s, const float* __restrict__ ex_s,
                                                            const uint32_t* __restrict__ ex_p, int n_lists, uint32_t k,
                                                            uint64_t* __restrict__ out_labels, float* __restrict__ out_dist,
                                                            uint32_t* __restrict__ out_found, uint32_t* __restrict__ out_flags,
                                                            int force_fallback, float eps) {
    __shared__ float sh_s[16][LIST];
    __shared__ uint32_t sh_p[16][LIST];
    __shared__ float sh_t[16];
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
            tl[j] = l < n_lists ? ub_s[(size_t)l * LIST + 63] : NEG_INF;  // (-inf: the list is not full, nothing was left out)
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
    if (force_fallback && n_rows > 0) flag = FLAG_FALLBACK;
    if ((uint32_t)lane < found && p != NO_POS) {
        out_labels[lane] = ids[p];
        out_dist[lane] = -s;
    }
    if (lane == 0) {
        out_found[0] = found;
        out_flags[0] = flag;
    }
}

// One query: stream + epilogue, merge + certificate.  ub_s / ub_p: the upper-bound lists [blocks][64] (what the other
// streams hand to merge_rescore_kernel; kept for the test hooks), ex_s / ex_p: the exact lists.  geom.unroll: 12 / 6 / 4 / 3 / 2
// fragments of 768 B in flight per wave.
void launch_scan_i6(const void* d_i6, const void* d_meta, const void* d_rows, int dtype, const uint64_t* d_ids, uint32_t n_rows,
                    const float* d_q, float* ub_s, uint32_t* ub_p, float* ex_s, uint32_t* ex_p, const ScanGeom& g, uint32_t k,
                    uint64_t* d_labels, float* d_dist, uint32_t* d_found, uint32_t* d_flags, int force_fallback, bool merge,
                    hipStream_t stream, hipEvent_t ev0, hipEvent_t ev1) {
    static bool attr_set = false;
    const uint32_t* x = reinterpret_cast<const uint32_t*>(d_i6);
    const float2* mt = reinterpret_cast<const float2*>(d_meta);
#define DAWN_I6_EACH(F) F(0, 12) F(0, 6) F(0, 4) F(0, 3) F(0, 2) F(1, 12) F(1, 6) F(1, 4) F(1, 3) F(1, 2)
    if (!attr_set) {
#define DAWN_I6_ATTR(RT_, PD_)                                                                                 \
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(scan_filter_i6s_kernel<RT_, PD_>),                \
                              hipFuncAttributeMaxDynamicSharedMemorySize, RescoreStage<RT_>::BYTES);
        DAWN_I6_EACH(DAWN_I6_ATTR)
#undef DAWN_I6_ATTR
        attr_set = true;
    }
    if (ev0) (void)hipEventRecord(ev0, stream);
    const int rt = dtype == ROW_BF16 ? 1 : 0;
    const int pd = g.unroll == 6 || g.unroll == 4 || g.unroll == 3 || g.unroll == 2 ? g.unroll : 12;
#define DAWN_I6_LAUNCH(RT_, PD_)                                                                                          \
    if (rt == RT_ && pd == PD_)                                                                                           \
        hipLaunchKernelGGL((scan_filter_i6s_kernel<RT_, PD_>), dim3(g.blocks), dim3(g.threads), RescoreStage<RT_>::BYTES, \
                           stream, x, mt, n_rows, d_q, d_rows, ub_s, ub_p, ex_s, ex_p);
    DAWN_I6_EACH(DAWN_I6_LAUNCH)
#undef DAWN_I6_LAUNCH
#undef DAWN_I6_EACH
    if (ev1) (void)hipEventRecord(ev1, stream);
    if (merge)
        hipLaunchKernelGGL(merge_exact_kernel, dim3(1), dim3(1024), 0, stream, d_ids, n_rows, ub_s, ex_s, ex_p, g.blocks, k,
                           d_labels, d_dist, d_found, d_flags, force_fallback, FILTER_EPS_I8);
}

}  // namespace dawn
