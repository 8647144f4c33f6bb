// rotate384.hpp — the fixed orthogonal rotation R of the quantised filter shadows (int8: scan_i8.hip, 6-bit: scan_i6.hip) and the
// constants of their query images.  See the header of scan_i8.hip for why the shadows live in a rotated basis.
#pragma once
#include <type_traits>

#include "kernels.hpp"
#include "wave_topk.hpp"

namespace dawn {

typedef int i32x4_t __attribute__((ext_vector_type(4)));
typedef int i32x16_t __attribute__((ext_vector_type(16)));

// ---- the rotation R -------------------------------------------------------------------------------------------------
__device__ __forceinline__ bool rot_neg(uint32_t k) { return ((k * 0x9E3779B1u) >> 19) & 1u; }  // sign of element k
constexpr float ROT_SCALE = 0.08838834764831845f;  // 1 / sqrt(128)

// One wavefront, lane l holds v[j] = element l + 64 j (j = 0..5) of a 384-vector: v <- R v.  Block b = elements
// 128 b .. 128 b + 127 = v[2b], v[2b+1]; index inside the block = l + 64 (j & 1): bit 6 in-thread, bits 0..5 across lanes.
__device__ __forceinline__ void rotate384_wave(float (&v)[6], int lane) {
#pragma unroll
    for (int j = 0; j < 6; ++j)
        if (rot_neg((uint32_t)(lane + 64 * j))) v[j] = -v[j];
#pragma unroll
    for (int t = 0; t < 2; ++t) {  // M_3 = (2/3) J - I on (block 0, block 1, block 2), element by element
        const float m = (v[t] + v[t + 2] + v[t + 4]) * (2.0f / 3.0f);
        v[t] = m - v[t];
        v[t + 2] = m - v[t + 2];
        v[t + 4] = m - v[t + 4];
    }
#pragma unroll
    for (int b = 0; b < 3; ++b) {
        const float a0 = v[2 * b], a1 = v[2 * b + 1];
        v[2 * b] = a0 + a1;
        v[2 * b + 1] = a0 - a1;
    }
    auto stage = [&](auto st) {
        constexpr int O = decltype(st)::value;
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            const float o = lane_xor_f32<O>(v[j], lane);
            v[j] = (lane & O) ? o - v[j] : v[j] + o;
        }
    };
    stage(std::integral_constant<int, 1>());
    stage(std::integral_constant<int, 2>());
    stage(std::integral_constant<int, 4>());
    stage(std::integral_constant<int, 8>());
    stage(std::integral_constant<int, 16>());
    stage(std::integral_constant<int, 32>());
#pragma unroll
    for (int j = 0; j < 6; ++j) v[j] *= ROT_SCALE;
}

// The conversion kernel's layout: thread (row r, part = tid & 7) holds v[j] = elements 32 j + 4 part + {0,1,2,3}
// (j = 0..11).  Block b = v[4b .. 4b+3]; index inside the block = 32 (j & 3) + 4 part + i: bits 0-1 in the float4, bits 2-4
// across the 8 neighbouring lanes, bits 5-6 in-thread.
__device__ __forceinline__ void rotate384_rowpart(f32x4 (&v)[12], uint32_t part, int lane) {
#pragma unroll
    for (int j = 0; j < 12; ++j) {
        const uint32_t k = 32u * j + 4u * part;
        if (rot_neg(k + 0)) v[j].x = -v[j].x;
        if (rot_neg(k + 1)) v[j].y = -v[j].y;
        if (rot_neg(k + 2)) v[j].z = -v[j].z;
        if (rot_neg(k + 3)) v[j].w = -v[j].w;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const f32x4 m = (v[j] + v[j + 4] + v[j + 8]) * (2.0f / 3.0f);
        v[j] = m - v[j];
        v[j + 4] = m - v[j + 4];
        v[j + 8] = m - v[j + 8];
    }
#pragma unroll
    for (int j = 0; j < 12; ++j) {  // bits 0, 1: inside the float4
        f32x4 a = v[j];
        a = f32x4{a.x + a.y, a.x - a.y, a.z + a.w, a.z - a.w};
        v[j] = f32x4{a.x + a.z, a.y + a.w, a.x - a.z, a.y - a.w};
    }
    auto stage = [&](auto st) {  // bits 2..4: lanes part ^ 1, 2, 4
        constexpr int O = decltype(st)::value;
#pragma unroll
        for (int j = 0; j < 12; ++j) {
            f32x4 o;
            o.x = lane_xor_f32<O>(v[j].x, lane);
            o.y = lane_xor_f32<O>(v[j].y, lane);
            o.z = lane_xor_f32<O>(v[j].z, lane);
            o.w = lane_xor_f32<O>(v[j].w, lane);
            v[j] = (lane & O) ? o - v[j] : v[j] + o;
        }
    };
    stage(std::integral_constant<int, 1>());
    stage(std::integral_constant<int, 2>());
    stage(std::integral_constant<int, 4>());
#pragma unroll
    for (int b = 0; b < 3; ++b) {  // bits 5, 6: v[4b + (0..3)]
        f32x4 a0 = v[4 * b], a1 = v[4 * b + 1], a2 = v[4 * b + 2], a3 = v[4 * b + 3];
        const f32x4 b0 = a0 + a1, b1 = a0 - a1, b2 = a2 + a3, b3 = a2 - a3;
        v[4 * b] = (b0 + b2) * ROT_SCALE;
        v[4 * b + 1] = (b1 + b3) * ROT_SCALE;
        v[4 * b + 2] = (b0 - b2) * ROT_SCALE;
        v[4 * b + 3] = (b1 - b3) * ROT_SCALE;
    }
}

constexpr float I8_QRES = 2.0e-3f;                   // |dq_i| <= I8_QRES * s_q
constexpr float I8_K2_PER_SQ = 1.1f * 19.6f * I8_QRES;  // K2 = I8_K2_PER_SQ * s_q  (sqrt(384) < 19.6)

}  // namespace dawn
