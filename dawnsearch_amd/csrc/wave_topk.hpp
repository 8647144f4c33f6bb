// wave_topk.hpp — device-side primitives shared by the scan kernels: 64-wide wavefront reductions,
// sorted 64-entry lists held one entry per lane, bitonic merge/sort, and the reference-order exact dot.
#pragma once
#include "kernels.hpp"

// Timestamp probes of the experiments build (scan_kernels.hip defines DAWN_TS before including this file); nothing otherwise.
#ifndef DAWN_TS
#define DAWN_TS(i)
#endif

namespace dawn {

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define NEG_INF (-__builtin_inff())
#define POS_INF (__builtin_inff())
constexpr uint32_t NO_POS = 0xFFFFFFFFu;

// once-read index stream: non-temporal 16-B loads
__device__ __forceinline__ f32x4 nt_load(const f32x4* p) { return __builtin_nontemporal_load(p); }

// ------------------------------------------------------------------------------------------------
// wave-level primitives (64-wide wavefront)
// ------------------------------------------------------------------------------------------------
template <int CTRL, int ROW_MASK, int BANK_MASK, bool BOUND>
__device__ __forceinline__ float dpp_mov(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK,
                                                                 BANK_MASK, BOUND));
}

// Sum over the 64 lanes; the total is valid in lane 63 (rows 3's lanes).  6 DPP adds, no LDS.
__device__ __forceinline__ float wave_sum_lane63(float v) {
    v += dpp_mov<0xB1, 0xf, 0xf, true>(v);    // quad_perm [1,0,3,2]
    v += dpp_mov<0x4E, 0xf, 0xf, true>(v);    // quad_perm [2,3,0,1]
    v += dpp_mov<0x141, 0xf, 0xf, true>(v);   // row_half_mirror
    v += dpp_mov<0x140, 0xf, 0xf, true>(v);   // row_mirror       -> every lane of a 16-row holds the row sum
    v += dpp_mov<0x142, 0xa, 0xf, false>(v);  // row_bcast15 into rows 1,3
    v += dpp_mov<0x143, 0xc, 0xf, false>(v);  // row_bcast31 into rows 2,3 -> row 3 holds the total
    return v;
}

__device__ __forceinline__ float read_lane63(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

// Lane i <- lane i ^ STRIDE without the LDS crossbar (ds_bpermute costs ~100+ cycles of latency per dependent
// step; the bitonic networks below are chains of them): quad_perm / row_shl+row_shr / row_ror DPP inside a 16-lane
// row, v_permlane16_swap / v_permlane32_swap (gfx950) across rows.
template <int STRIDE>
__device__ __forceinline__ uint32_t lane_xor_u32(uint32_t v, int lane) {
    const int iv = (int)v;
    if (STRIDE == 1) return (uint32_t)__builtin_amdgcn_update_dpp(iv, iv, 0xB1, 0xf, 0xf, false);  // quad_perm [1,0,3,2]
    if (STRIDE == 2) return (uint32_t)__builtin_amdgcn_update_dpp(iv, iv, 0x4E, 0xf, 0xf, false);  // quad_perm [2,3,0,1]
    if (STRIDE == 4) {
        const int up = __builtin_amdgcn_update_dpp(iv, iv, 0x104, 0xf, 0xf, false);  // row_shl:4: lane i <- i + 4
        const int dn = __builtin_amdgcn_update_dpp(iv, iv, 0x114, 0xf, 0xf, false);  // row_shr:4: lane i <- i - 4
        return (uint32_t)((lane & 4) ? dn : up);
    }
    if (STRIDE == 8) return (uint32_t)__builtin_amdgcn_update_dpp(iv, iv, 0x128, 0xf, 0xf, false);  // row_ror:8
    if (STRIDE == 16) {
        // rows (r0 r1 r2 r3) x2 -> [0] = (r0 r0 r2 r2), [1] = (r1 r1 r3 r3)
        const auto r = __builtin_amdgcn_permlane16_swap(v, v, false, false);
        return (lane & 16) ? r[0] : r[1];
    }
    // halves (lo hi) x2 -> [0] = (lo lo), [1] = (hi hi)
    const auto r = __builtin_amdgcn_permlane32_swap(v, v, false, false);
    return (lane & 32) ? r[0] : r[1];
}
template <int STRIDE>
__device__ __forceinline__ float lane_xor_f32(float v, int lane) {
    return __builtin_bit_cast(float, lane_xor_u32<STRIDE>(__builtin_bit_cast(uint32_t, v), lane));
}
// lane i <- lane i - 1 (lane 0 keeps its own value): DPP wave_shr:1
__device__ __forceinline__ uint32_t lane_up1_u32(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x138, 0xf, 0xf, false);
}

// (score desc, row asc) strict order; fillers are (-inf, NO_POS).
__device__ __forceinline__ bool better(float s, uint32_t p, float s2, uint32_t p2) {
    return s > s2 || (s == s2 && p < p2);
}

// Insert a wave-uniform candidate into the wave's descending list (one entry per lane).
__device__ __forceinline__ void wave_insert(float& ls, uint32_t& lp, float s, uint32_t p, int lane) {
    const bool ahead = better(ls, lp, s, p);
    const int pos = __popcll(__ballot(ahead));
    const float ps = __builtin_bit_cast(float, lane_up1_u32(__builtin_bit_cast(uint32_t, ls)));
    const uint32_t pp = lane_up1_u32(lp);
    if (lane > pos) {
        ls = ps;
        lp = pp;
    } else if (lane == pos) {
        ls = s;
        lp = p;
    }
}

template <int STRIDE>
__device__ __forceinline__ void merge64_stage(float& s, uint32_t& p, int lane) {
    const float s2 = lane_xor_f32<STRIDE>(s, lane);
    const uint32_t p2 = lane_xor_u32<STRIDE>(p, lane);
    const bool lower = (lane & STRIDE) == 0;
    const bool other_better = better(s2, p2, s, p);
    if (lower == other_better) {
        s = s2;
        p = p2;
    }
}

// Merge another descending list (given REVERSED: lane i holds other[63-i]) into mine; result = top 64 of
// the union, descending.  Bitonic half-cleaner + 6 compare-exchange stages.
__device__ __forceinline__ void merge64(float& s, uint32_t& p, float os_rev, uint32_t op_rev, int lane) {
    if (better(os_rev, op_rev, s, p)) {
        s = os_rev;
        p = op_rev;
    }
    merge64_stage<32>(s, p, lane);
    merge64_stage<16>(s, p, lane);
    merge64_stage<8>(s, p, lane);
    merge64_stage<4>(s, p, lane);
    merge64_stage<2>(s, p, lane);
    merge64_stage<1>(s, p, lane);
}

// Block-level tree merge of per-wave lists through LDS; result in wave 0.  nwaves is a power of two.
// (A flat form — four leader waves merge three lists each, wave 0 merges the leaders': three barrier pairs instead of four —
// measured the same: in merge_rescore_kernel the ~6 us up to the end of this merge are the sixteen waves' own 256 bitonic
// merges sharing one CU's vector ALUs, not the barriers: tools/merge_ts.py.)
__device__ __forceinline__ void block_merge(float& s, uint32_t& p, float (*sh_s)[LIST], uint32_t (*sh_p)[LIST],
                                            int wave, int lane, int nwaves) {
    for (int stride = nwaves >> 1; stride >= 1; stride >>= 1) {
        if (wave >= stride && wave < 2 * stride) {
            sh_s[wave][lane] = s;
            sh_p[wave][lane] = p;
        }
        __syncthreads();
        if (wave < stride) {
            const float os = sh_s[wave + stride][63 - lane];
            const uint32_t op = sh_p[wave + stride][63 - lane];
            merge64(s, p, os, op, lane);
        }
        __syncthreads();
    }
}

// Sequential, un-fused f32 dot in the reference's order (vector.rs:128-134): result += a[i]*b[i].
// UNROLL: chunks fetched ahead of the dependent add chain (4 for rows in global memory: registers; 16 for rows staged in
// LDS, where an un-prefetched read costs more than the four adds it feeds)
template <int UNROLL = 4>
__device__ __forceinline__ float exact_dot_seq(const float* __restrict__ qv, const f32x4* __restrict__ row) {
    float acc = 0.0f;
#pragma unroll UNROLL
    for (int c = 0; c < ROW_F4; ++c) {
        const f32x4 xv = row[c];
        const f32x4 qq = reinterpret_cast<const f32x4*>(qv)[c];
        acc = __fadd_rn(acc, __fmul_rn(qq.x, xv.x));
        acc = __fadd_rn(acc, __fmul_rn(qq.y, xv.y));
        acc = __fadd_rn(acc, __fmul_rn(qq.z, xv.z));
        acc = __fadd_rn(acc, __fmul_rn(qq.w, xv.w));
    }
    return acc;
}

// The adds of the same sum over products already rounded one by one (block_exact_dots): acc = fl(acc + p[i]), i ascending.
__device__ __forceinline__ float sum_seq(const f32x4* __restrict__ prod) {
    float acc = 0.0f;
#pragma unroll 16
    for (int c = 0; c < ROW_F4; ++c) {
        const f32x4 pv = prod[c];
        acc = __fadd_rn(acc, pv.x);
        acc = __fadd_rn(acc, pv.y);
        acc = __fadd_rn(acc, pv.z);
        acc = __fadd_rn(acc, pv.w);
    }
    return acc;
}

// ---- bf16 rows (DAWN_DTYPE_BF16): 384 x bf16 = 768 B = 48 16-B chunks of 8 values; element k of a chunk word w:
// even k in the low half, odd k in the high half (little endian).  Widening to f32 is exact.
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
constexpr int ROW_C8 = 48;  // 16-B chunks per bf16 row

__device__ __forceinline__ float bf16_lo(uint32_t w) { return __builtin_bit_cast(float, w << 16); }
__device__ __forceinline__ float bf16_hi(uint32_t w) { return __builtin_bit_cast(float, w & 0xFFFF0000u); }

// f32 -> bf16 bits, round to nearest even (inputs are finite: every row passed is_normalized)
__device__ __forceinline__ uint32_t f32_to_bf16_rne(float f) {
    const uint32_t u = __builtin_bit_cast(uint32_t, f);
    return (u + 0x7FFFu + ((u >> 16) & 1u)) >> 16;
}

// Slot (in 16-B units from the start of the array) of chunk c (elements 8c..8c+7) of row `row` in a fragment-ordered
// 16-bit row array (ROW_BF16 / ROW_F16S, kernels.hpp).  Consecutive rows of a 32-row sub-tile are consecutive slots.
__device__ __forceinline__ size_t frag_chunk(size_t row, int c) {
    return (row >> 6) * 3072 + (((row >> 5) & 1) * 24 + (c >> 1)) * 64 + (c & 1) * 32 + (row & 31);
}

// bf16 row given as 48 16-B chunks, chunk c at row[c * stride] (stride in 16-B units: 1 = a contiguous copy of the row)
template <int UNROLL = 2>
__device__ __forceinline__ float exact_dot_seq_bf16(const float* __restrict__ qv, const u32x4* __restrict__ row,
                                                    int stride = 1) {
    float acc = 0.0f;
#pragma unroll UNROLL
    for (int c = 0; c < ROW_C8; ++c) {
        const u32x4 w = row[(size_t)c * stride];
        const f32x4 q0 = reinterpret_cast<const f32x4*>(qv)[2 * c];
        const f32x4 q1 = reinterpret_cast<const f32x4*>(qv)[2 * c + 1];
        acc = __fadd_rn(acc, __fmul_rn(q0.x, bf16_lo(w.x)));
        acc = __fadd_rn(acc, __fmul_rn(q0.y, bf16_hi(w.x)));
        acc = __fadd_rn(acc, __fmul_rn(q0.z, bf16_lo(w.y)));
        acc = __fadd_rn(acc, __fmul_rn(q0.w, bf16_hi(w.y)));
        acc = __fadd_rn(acc, __fmul_rn(q1.x, bf16_lo(w.z)));
        acc = __fadd_rn(acc, __fmul_rn(q1.y, bf16_hi(w.z)));
        acc = __fadd_rn(acc, __fmul_rn(q1.z, bf16_lo(w.w)));
        acc = __fadd_rn(acc, __fmul_rn(q1.w, bf16_hi(w.w)));
    }
    return acc;
}

// Reference-order exact dot of query qv with row `p` of an index of row type RT (0 = f32, 1 = bf16).
template <int RT>
__device__ __forceinline__ float exact_dot_row(const float* __restrict__ qv, const void* __restrict__ x, size_t p) {
    if (RT == 1) return exact_dot_seq_bf16(qv, reinterpret_cast<const u32x4*>(x) + frag_chunk(p, 0), 32);
    return exact_dot_seq(qv, reinterpret_cast<const f32x4*>(x) + p * ROW_F4);
}

// Exact dots of ONE query with the 64 shortlisted rows held by wave 0 (row index p per lane, NO_POS = none), for a
// whole workgroup: every wave copies its share of the rows into LDS with independent coalesced 16-B loads (all in
// flight together — a lane walking its own row through HBM pays the memory latency 24 times over), then wave 0 runs
// the reference-order sums out of LDS.  stage: RescoreStage<RT>::BYTES of LDS (row stride + 16 B: lanes of a
// 16-lane group read different banks); sh_rows: 64 words; blockDim.x >= 384.  Returns the dot in wave 0 (0 for
// NO_POS lanes).
template <int RT>
struct RescoreStage {
    static constexpr int CH = RT == 1 ? ROW_C8 : ROW_F4;  // 16-B chunks per row
    static constexpr int STRIDE = CH * 16 + 16;
    static constexpr int ROWS_BYTES = LIST * STRIDE;
    static constexpr int BYTES = ROWS_BYTES + EM * 4;  // + the query
};

template <int RT>
// q_val: element threadIdx.x of the query (threads < 384), loaded by the caller at the START of its kernel — a load issued
// here would be one more exposed round trip in front of the row gather
__device__ __forceinline__ float block_exact_dots(float q_val, const void* __restrict__ x, uint32_t p,
                                                  unsigned char* stage, uint32_t* sh_rows, int wave, int lane) {
    typedef RescoreStage<RT> S;
    float* sh_q = reinterpret_cast<float*>(stage + S::ROWS_BYTES);  // the query too: 96 dependent global reads otherwise
    if (wave == 0) sh_rows[lane] = p;
    if (threadIdx.x < EM) sh_q[threadIdx.x] = q_val;
    __syncthreads();
    DAWN_TS(3);
    const u32x4* xr = reinterpret_cast<const u32x4*>(x);
    for (int i = threadIdx.x; i < LIST * S::CH; i += blockDim.x) {
        const int r = i / S::CH, c = i % S::CH;
        const uint32_t row = sh_rows[r];
        if (row != NO_POS)
            *reinterpret_cast<u32x4*>(stage + r * S::STRIDE + c * 16) =
                RT == 1 ? xr[frag_chunk(row, c)] : xr[(size_t)row * S::CH + c];
    }
    __syncthreads();
    DAWN_TS(4);
    float dot = 0.0f;
    if (RT == 0) {
        // The reference's sum is sequential in the ADDS only: result += a[i] * b[i] rounds every product on its own.  So all
        // threads turn the staged rows into their products in place (fl(q_i x_i), the same __fmul_rn the chain would do) and
        // wave 0 is left with 384 dependent adds per row instead of 384 multiply-add pairs fed by two LDS reads each: 4.4 -> ~1.5 us.
        for (int i = threadIdx.x; i < LIST * S::CH; i += blockDim.x) {
            const int r = i / S::CH, c = i % S::CH;
            f32x4* px = reinterpret_cast<f32x4*>(stage + r * S::STRIDE + c * 16);
            const f32x4 xv = *px, qq = reinterpret_cast<const f32x4*>(sh_q)[c];
            *px = f32x4{__fmul_rn(qq.x, xv.x), __fmul_rn(qq.y, xv.y), __fmul_rn(qq.z, xv.z), __fmul_rn(qq.w, xv.w)};
        }
        __syncthreads();
        if (wave == 0 && p != NO_POS) dot = sum_seq(reinterpret_cast<const f32x4*>(stage + lane * S::STRIDE));
    } else if (wave == 0 && p != NO_POS) {
        dot = exact_dot_seq_bf16<8>(sh_q, reinterpret_cast<const u32x4*>(stage + lane * S::STRIDE));
    }
    DAWN_TS(5);
    return dot;
}

// (distance asc, row asc)
__device__ __forceinline__ bool less_dp(float d, uint32_t p, float d2, uint32_t p2) {
    return d < d2 || (d == d2 && p < p2);
}

template <int K2, int J>
__device__ __forceinline__ void sort64_stage(float& d, uint32_t& p, int lane) {
    const float d2 = lane_xor_f32<J>(d, lane);
    const uint32_t p2 = lane_xor_u32<J>(p, lane);
    const bool asc = (lane & K2) == 0;
    const bool lower = (lane & J) == 0;
    const bool keep_small = (lower == asc);
    const bool other_less = less_dp(d2, p2, d, p);
    const bool other_greater = less_dp(d, p, d2, p2);
    if (keep_small ? other_less : other_greater) {
        d = d2;
        p = p2;
    }
}
template <int K2, int J>
struct Sort64Run {
    static __device__ __forceinline__ void run(float& d, uint32_t& p, int lane) {
        sort64_stage<K2, J>(d, p, lane);
        Sort64Run<K2, J / 2>::run(d, p, lane);
    }
};
template <int K2>
struct Sort64Run<K2, 0> {
    static __device__ __forceinline__ void run(float&, uint32_t&, int) {}
};

// Full bitonic sort of one (d, p) per lane, ascending.
__device__ __forceinline__ void sort64_asc(float& d, uint32_t& p, int lane) {
    Sort64Run<2, 1>::run(d, p, lane);
    Sort64Run<4, 2>::run(d, p, lane);
    Sort64Run<8, 4>::run(d, p, lane);
    Sort64Run<16, 8>::run(d, p, lane);
    Sort64Run<32, 16>::run(d, p, lane);
    Sort64Run<64, 32>::run(d, p, lane);
}

// smallest float >= t (t finite, double)
__device__ __forceinline__ float round_up_f32(double t) {
    float f = (float)t;
    if ((double)f < t) {
        if (f == 0.0f) return 1.0e-45f;
        int bits = __builtin_bit_cast(int, f);
        bits += (f > 0.0f) ? 1 : -1;
        f = __builtin_bit_cast(float, bits);
    }
    return f;
}



// ------------------------------------------------------------------------------------------------
// The certificate loop shared by the search tails (merge_rescore_kernel: streaming filters; select_rescore_kernel:
// matrix-core passes).
//
// Round 1: the 64 best candidates by FILTER score are rescored exactly (reference order); every other row is known to
// score <= m = the 64th filter score (or `base`, the bound the filter pass itself gives for rows that never became
// candidates), hence to lie at distance >= fl(1 - up(m + eps)); if that exceeds the k-th exact distance strictly, the
// exact top-k is inside the shortlist.  The int8 filters' scores are upper bounds with E (+ K2) ~ 0.009 .. 0.017 of slack
// on unit vectors — the order of the gap between the 20th and the 64th best score of a large index — so a marginal
// failure is ordinary there (tools/cert_stats.py: k = 20 on 100 M rows fails round 1 for 10 .. 100 % of the queries).
// Instead of the 1024-row second chance below (a radix selection and 1024 uncoalesced row walks: 0.1 - 0.2 ms) the loop
// DEEPENS: round r + 1 takes the next 64 candidates by filter score (selection repeated with everything down to the last
// entry of round r excluded), rescores them the same cooperative way (block_exact_dots: ~4 us), merges the exact results
// and tries the certificate with the bound 64 ranks further down — up to CERT_ROUNDS x 64 rows.  Same mathematics as
// round 1, each round ~10 us.
//   base: read AFTER the first select() — the caller may compute it there (merge_rescore_kernel does);
//   q_val: element threadIdx.x of the query (see block_exact_dots);
//   select(first, ex_s, ex_p, s, p): called by every thread; leaves in wave 0 the (up to) 64 best candidates by filter
//   score (descending, ties -> lower row; fillers (-inf, NO_POS) last) among those strictly worse than (ex_s, ex_p)
//   (first: among all).
// Returns the flag; wave 0 holds the result in (bs = -distance descending, bp = row).  heavy: the rounds are used up (or
// the candidates ran out above `base`) — the caller may still try second_chance().
// ------------------------------------------------------------------------------------------------
constexpr int CERT_ROUNDS = 4;

template <int RT, class SelectFn>
__device__ __forceinline__ uint32_t certify_rounds(SelectFn select, const float& base, bool complete, uint32_t n_rows, uint32_t found,
                                                   float eps, int force_fallback, float q_val,
                                                   const void* __restrict__ x, unsigned char* rescore_stage, uint32_t* sh_rows,
                                                   uint32_t* sh_ctl, int wave, int lane, float& bs, uint32_t& bp, bool& heavy) {
    float s;
    uint32_t p;
    select(true, 0.f, 0u, s, p);
    bs = NEG_INF;
    bp = NO_POS;
    uint32_t flag = FLAG_OK;
    heavy = false;
    for (int r = 0;; ++r) {
        const float dot = block_exact_dots<RT>(q_val, x, p, rescore_stage, sh_rows, wave, lane);
        if (wave == 0) {
            const uint32_t have_r = __popcll(__ballot(p != NO_POS));
            const float s_last = read_lane63(s);
            const uint32_t p_last = (uint32_t)__builtin_amdgcn_readlane((int)p, 63);
            float d = POS_INF;
            uint32_t pr = p;
            if (p != NO_POS) {
                d = __fsub_rn(1.0f, dot);  // vector.rs:133  1.0 - result
                if (!(d == d)) {
                    d = POS_INF;
                    pr = NO_POS;
                }
            }
            sort64_asc(d, pr, lane);
            if (r == 0) {
                bs = -d;
                bp = pr;
            } else {
                const float os = -__shfl(d, 63 - lane);
                const uint32_t op = __shfl(pr, 63 - lane);
                merge64(bs, bp, os, op, lane);
            }
            // bound on the filter score of everything not rescored so far
            const float m = have_r == (uint32_t)LIST ? fmaxf(s_last, base) : base;
            uint32_t next = 0;
            flag = FLAG_OK;
            if (n_rows > (uint32_t)LIST && found > 0) {
                const uint32_t have = __popcll(__ballot(bp != NO_POS));
                if (have < found || !complete) {
                    flag = FLAG_FALLBACK;  // not enough candidates / candidates were dropped: only the exact pass can tell
                } else if (m > NEG_INF) {
                    const float t = round_up_f32((double)m + (double)eps);
                    const float d_bound = __fsub_rn(1.0f, t);
                    const float dk = -__builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, bs), (int)found - 1));
                    if (!(d_bound > dk)) {
                        flag = FLAG_FALLBACK;
                        // worth another round: there are candidates left above `base`
                        if (!force_fallback && have_r == (uint32_t)LIST && s_last > base) next = (r + 1 < CERT_ROUNDS) ? 1u : 2u;
                        else if (!force_fallback) next = 2u;
                    }
                }
            }
            if (flag == FLAG_OK && r > 0) flag = FLAG_DEEP;
            if (lane == 0) {
                sh_ctl[0] = next;
                sh_ctl[1] = __builtin_bit_cast(uint32_t, s_last);
                sh_ctl[2] = p_last;
                sh_ctl[3] = flag;
            }
        }
        DAWN_TS(6);
        __syncthreads();
        const uint32_t next = sh_ctl[0];
        const float ex_s = __builtin_bit_cast(float, sh_ctl[1]);
        const uint32_t ex_p = sh_ctl[2];
        flag = sh_ctl[3];
        __syncthreads();
        if (next != 1u) {
            heavy = next == 2u;
            break;
        }
        select(false, ex_s, ex_p, s, p);
    }
    if (force_fallback && n_rows > 0) flag = FLAG_FALLBACK;
    return flag;
}

// ------------------------------------------------------------------------------------------------
// Second chance of a failed certificate.  The first certificate bounds everything outside a 64-row shortlist by the
// 64th filter score; it fails when more than ~54 rows crowd the top of a query within the filter's error band (near-
// duplicate pages, k = 64).  Before the whole index is rescanned exactly, the candidates the filter pass left behind are
// used once more: the (up to) 1024 best of them by filter score are ALL rescored exactly, and everything else is
// bounded by m2 = max(theta, base) — theta: an upper bound on the filter score of every entry not taken, base: the bound
// the filter pass itself gives for rows that are not entries at all.  Same certificate, 1024 rows deep instead of 64.
//   load(e, score, row) -> bool: entry e of n_entries (false: empty slot); every row appears at most once.
// Called by all 1024 threads of the block; scratch: >= 12.5 KB of LDS; the result (score = -distance descending, row)
// and the verdict are valid in wave 0.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t order_key(float s) {  // ascending in s (finite or -inf)
    const uint32_t u = __builtin_bit_cast(uint32_t, s);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float key_to_float(uint32_t k) {
    const uint32_t u = (k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k;
    return __builtin_bit_cast(float, u);
}

// inclusive prefix sum over the 1024 threads of the block (thread order); lds16: 16 words
__device__ __forceinline__ uint32_t block_incl_scan_u32(uint32_t v, uint32_t* lds16, int wave, int lane) {
    int x = (int)v;
    x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, true);   // row_shr:1
    x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, true);   // row_shr:2
    x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, true);   // row_shr:4
    x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xf, true);   // row_shr:8
    x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xa, 0xf, false);  // row_bcast15 -> rows 1, 3
    x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xc, 0xf, false);  // row_bcast31 -> rows 2, 3
    if (lane == 63) lds16[wave] = (uint32_t)x;
    __syncthreads();
    uint32_t off = 0;
    for (int w = 0; w < wave; ++w) off += lds16[w];
    __syncthreads();
    return (uint32_t)x + off;
}

// K: how many entries are rescored at most — 1024 (the streaming tails: the best 1024 of the workgroups' lists) or 8192 (the
// matrix-core tails: EVERY candidate the pass appended, i.e. every row whose bound exceeds the pass's threshold — on topical
// data a query inside a cluster of a few thousand near rows is settled here, 1024 rows per round, instead of by a stream of
// the bounded pass).  scratch: (2048 + K + 32) * 4 bytes.
constexpr uint32_t SECOND_CHANCE_K = 1024;
constexpr uint32_t SECOND_CHANCE_K_ALL = 8192;

template <int RT, class LoadFn>
__device__ __forceinline__ bool second_chance(LoadFn load, uint32_t n_entries, float base, const float* __restrict__ qv,
                                              const void* __restrict__ x, uint32_t found, float eps,
                                              unsigned char* scratch, float (*sh_s)[LIST], uint32_t (*sh_p)[LIST], int wave,
                                              int lane, float& out_s, uint32_t& out_p, const uint32_t K = SECOND_CHANCE_K) {
    uint32_t* hist = reinterpret_cast<uint32_t*>(scratch);
    uint32_t* sel = hist + 2048;
    uint32_t* misc = sel + K;  // [0] crossing bin (or ~0) [1] entries above it [2] selected count; [16..31] scan
    const uint32_t tid = threadIdx.x;
    uint32_t prefix = 0;       // bits fixed so far (pass 2: the crossing 11-bit bin of pass 1)
    uint32_t above_total = 0;  // entries strictly above the crossing bin(s)
    bool take_all = false;
    uint32_t edge22 = 0;       // entries with key >> 10 > edge22 are taken
    for (int pass = 0; pass < 2 && !take_all; ++pass) {
        for (uint32_t i = tid; i < 2048; i += 1024) hist[i] = 0;
        if (tid < 3) misc[tid] = 0xFFFFFFFFu * (tid == 0);
        __syncthreads();
        for (uint32_t e = tid; e < n_entries; e += 1024) {
            float sc;
            uint32_t row;
            if (!load(e, sc, row)) continue;
            const uint32_t key = order_key(sc);
            if (pass == 0) atomicAdd(&hist[key >> 21], 1u);
            else if ((key >> 21) == prefix) atomicAdd(&hist[(key >> 10) & 2047u], 1u);
        }
        __syncthreads();
        // bins in descending order: thread t owns bins 2047 - 2t and 2046 - 2t
        const uint32_t c0 = hist[2047 - 2 * tid], c1 = hist[2046 - 2 * tid];
        const uint32_t incl = block_incl_scan_u32(c0 + c1, misc + 16, wave, lane);
        const uint32_t excl = incl - (c0 + c1);
        const uint32_t budget = K - above_total;  // entries that may still be taken
        if (excl <= budget && excl + c0 > budget) {
            misc[0] = 2047 - 2 * tid;
            misc[1] = excl;
        } else if (excl + c0 <= budget && incl > budget) {
            misc[0] = 2046 - 2 * tid;
            misc[1] = excl + c0;
        }
        __syncthreads();
        const uint32_t bin = misc[0];
        if (bin == 0xFFFFFFFFu) {  // everything (left) fits
            if (pass == 0) take_all = true;
            else edge22 = (prefix << 11);  // whole crossing bin of pass 1 fits after all: cannot happen (kept for safety)
            if (pass == 1) edge22 = (prefix << 11) - 1;
        } else {
            above_total += misc[1];
            if (pass == 0) prefix = bin;
            else edge22 = (prefix << 11) | bin;
        }
        __syncthreads();
    }
    const float theta = take_all ? NEG_INF : key_to_float((edge22 << 10) | 0x3FFu);  // >= score of every entry not taken
    if (tid == 0) misc[2] = 0;
    __syncthreads();
    for (uint32_t e = tid; e < n_entries; e += 1024) {
        float sc;
        uint32_t row;
        if (!load(e, sc, row)) continue;
        if (take_all || (order_key(sc) >> 10) > edge22) {
            const uint32_t pos = atomicAdd(&misc[2], 1u);
            if (pos < K) sel[pos] = row;
        }
    }
    __syncthreads();
    const uint32_t count = misc[2];
    if (count > K || count < found) return false;  // (the first cannot happen)
    // exact scores, 1024 rows per round (a thread per row), every round merged into the running top-64 held by wave 0
    out_s = NEG_INF;
    out_p = NO_POS;
    for (uint32_t r0 = 0; r0 < count; r0 += 1024u) {
        float d = POS_INF;
        uint32_t row = NO_POS;
        if (r0 + tid < count) {
            row = sel[r0 + tid];
            const float dd = __fsub_rn(1.0f, exact_dot_row<RT>(qv, x, row));
            if (dd == dd) d = dd;
            else row = NO_POS;
        }
        sort64_asc(d, row, lane);
        float s = -d;
        uint32_t pr = row;
        if (wave == 0 && r0 > 0) {
            const float os = __shfl(out_s, 63 - lane);
            const uint32_t op = __shfl(out_p, 63 - lane);
            merge64(s, pr, os, op, lane);
        }
        block_merge(s, pr, sh_s, sh_p, wave, lane, 16);
        if (wave == 0) {
            out_s = s;
            out_p = pr;
        }
    }
    bool ok = false;
    if (wave == 0) {
        const uint32_t have = __popcll(__ballot(out_p != NO_POS));
        const float m2 = fmaxf(theta, base);
        const float t = round_up_f32((double)m2 + (double)eps);
        const float d_bound = __fsub_rn(1.0f, t);
        const float dk = -__builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, out_s), (int)found - 1));
        ok = have >= found && (m2 == NEG_INF || d_bound > dk);
    }
    return ok;
}

}  // namespace dawn
