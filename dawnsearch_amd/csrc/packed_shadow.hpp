// packed_shadow.hpp — layout constants and bound terms of the packed 5- / 6-bit shadow of the rows (scan_i6.hip: construction and
// the single-query stream; scan_bounded.hip: the bounded exact pass of single queries streams the same shadow).
#pragma once
#include <cstdint>

#include "kernels.hpp"
#include "rotate384.hpp"

namespace dawn {

typedef uint32_t u32x3 __attribute__((ext_vector_type(3)));
typedef u32x3 u32x3_u __attribute__((aligned(4)));  // (fragments are 12-B lane slots: 4-B aligned)

constexpr uint32_t I6_FRAG_DW = 64 * 3;            // 6-bit form: dwords per fragment
constexpr uint32_t I6_SUB_DW = 12 * I6_FRAG_DW;    // ... per sub-tile (9216 B)
constexpr uint32_t I5_SUB_DW = 1920;               // 5-bit form: dwords per sub-tile (7680 B)
constexpr uint32_t I5_HALF_DW = 960;               // [H (192 dwords) | N N N (256 dwords each)]
template <int BITS> struct PackedShadow {
    static constexpr float LEVELS = BITS == 6 ? 31.0f : 15.0f;
    static constexpr int OFFSET = BITS == 6 ? 32 : 16;  // code = value + OFFSET
    static constexpr uint32_t SUB_DW = BITS == 6 ? I6_SUB_DW : I5_SUB_DW;
};
// K2 of a sub-tile with measured error E and a query of scale s_q: (I6_XNORM + E) * I6_K2U_PER_SQ * s_q  (header)
constexpr float I6_XNORM = 1.015f;
constexpr float I6_K2U_PER_SQ = 19.6f * I8_QRES;


__device__ __forceinline__ u32x3 frag_load(const uint32_t* p) {
    return __builtin_nontemporal_load(reinterpret_cast<const u32x3_u*>(p));
}

}  // namespace dawn
