// wave_topk.hpp — device-side primitives shared by the scan kernels: 64-wide wavefront reductions,
// sorted 64-entry lists held one entry per lane, bitonic merge/sort, and the reference-order exact dot.
#pragma once
#include "kernels.hpp"

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

// (score desc, row asc) strict order; fillers are (-inf, NO_POS).
__device__ __forceinline__ bool better(float s, uint32_t p, float s2, uint32_t p2) {
    return s > s2 || (s == s2 && p < p2);
}

// Insert a wave-uniform candidate into the wave's descending list (one entry per lane).
__device__ __forceinline__ void wave_insert(float& ls, uint32_t& lp, float s, uint32_t p, int lane) {
    const bool ahead = better(ls, lp, s, p);
    const int pos = __popcll(__ballot(ahead));
    const float ps = __shfl_up(ls, 1);
    const uint32_t pp = __shfl_up(lp, 1);
    if (lane > pos) {
        ls = ps;
        lp = pp;
    } else if (lane == pos) {
        ls = s;
        lp = p;
    }
}

// Merge another descending list (given REVERSED: lane i holds other[63-i]) into mine; result = top 64 of
// the union, descending.  Bitonic half-cleaner + 6 compare-exchange stages.
__device__ __forceinline__ void merge64(float& s, uint32_t& p, float os_rev, uint32_t op_rev, int lane) {
    if (better(os_rev, op_rev, s, p)) {
        s = os_rev;
        p = op_rev;
    }
#pragma unroll
    for (int stride = 32; stride >= 1; stride >>= 1) {
        const float s2 = __shfl_xor(s, stride);
        const uint32_t p2 = __shfl_xor(p, stride);
        const bool lower = (lane & stride) == 0;
        const bool other_better = better(s2, p2, s, p);
        if (lower == other_better) {
            s = s2;
            p = p2;
        }
    }
}

// Block-level tree merge of per-wave lists through LDS; result in wave 0.  nwaves is a power of two.
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
__device__ __forceinline__ float exact_dot_seq(const float* __restrict__ qv, const f32x4* __restrict__ row) {
    float acc = 0.0f;
#pragma unroll 4
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

__device__ __forceinline__ float exact_dot_seq_bf16(const float* __restrict__ qv, const u32x4* __restrict__ row) {
    float acc = 0.0f;
#pragma unroll 2
    for (int c = 0; c < ROW_C8; ++c) {
        const u32x4 w = row[c];
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
    if (RT == 1) return exact_dot_seq_bf16(qv, reinterpret_cast<const u32x4*>(x) + p * ROW_C8);
    return exact_dot_seq(qv, reinterpret_cast<const f32x4*>(x) + p * ROW_F4);
}

// (distance asc, row asc)
__device__ __forceinline__ bool less_dp(float d, uint32_t p, float d2, uint32_t p2) {
    return d < d2 || (d == d2 && p < p2);
}

// Full bitonic sort of one (d, p) per lane, ascending.
__device__ __forceinline__ void sort64_asc(float& d, uint32_t& p, int lane) {
#pragma unroll
    for (int k2 = 2; k2 <= 64; k2 <<= 1) {
#pragma unroll
        for (int j = k2 >> 1; j >= 1; j >>= 1) {
            const float d2 = __shfl_xor(d, j);
            const uint32_t p2 = __shfl_xor(p, j);
            const bool asc = (lane & k2) == 0;
            const bool lower = (lane & j) == 0;
            const bool keep_small = (lower == asc);
            const bool other_less = less_dp(d2, p2, d, p);
            const bool other_greater = less_dp(d, p, d2, p2);
            if (keep_small ? other_less : other_greater) {
                d = d2;
                p = p2;
            }
        }
    }
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


}  // namespace dawn
