// Dev check: the operand layout of v_mfma_scale_f32_16x16x128_f8f6f4 with FP6 (e2m3) operands and E8M0 block scales, as the
// FP6 filter shadow (csrc/scan_f6.hip) assumes it:
//   A: lane l holds row (l & 15), k-block (l >> 4): k = 32 (l >> 4) + i, i = 0..31, as a little-endian bit string of 32 x 6 bits
//      in dwords 0..5 of the 8-dword operand (element i at bits [6 i, 6 i + 6)); B likewise with column (l & 15);
//   D: lane l, register r: row = 4 (l >> 4) + r, column = l & 15;
//   e2m3 code c: sign = c >> 5, e = (c >> 3) & 3, m = c & 7: value = e ? (1 + m / 8) 2^(e - 1) : m / 8;
//   scale operand: byte 0 of the lane's dword (opsel 0) = E8M0 exponent (2^(byte - 127)) of the lane's 32-element block.
// Prints the largest deviation from a host reference in double: 0 (products of two 4-bit significands and f32 accumulation of 128
// of them are exact at these magnitudes) if the assumptions hold.
// build: hipcc --offload-arch=gfx950 -O2 -o fp6_layout_check fp6_layout_check.hip
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <vector>
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

static double f6_value(uint32_t c) {
    const int s = (c >> 5) & 1, e = (c >> 3) & 3, m = c & 7;
    const double v = e ? (1.0 + m / 8.0) * std::ldexp(1.0, e - 1) : m / 8.0;
    return s ? -v : v;
}

__global__ void k(const uint32_t* a, const uint32_t* b, const uint32_t* sa, const uint32_t* sb, float* d) {
    const int l = threadIdx.x;
    i32x8 av, bv;
    for (int j = 0; j < 8; ++j) {
        av[j] = j < 6 ? (int)a[l * 6 + j] : 0;
        bv[j] = j < 6 ? (int)b[l * 6 + j] : 0;
    }
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(av, bv, acc, 2, 2, 0, (int)sa[l], 0, (int)sb[l]);
    for (int r = 0; r < 4; ++r) d[l * 4 + r] = acc[r];
}

int main() {
    std::vector<uint32_t> codeA(16 * 128), codeB(16 * 128), ha(64 * 6, 0), hb(64 * 6, 0), hsa(64), hsb(64);
    uint32_t x = 12345;
    auto rnd = [&]() { x = x * 1664525u + 1013904223u; return x >> 8; };
    for (auto& c : codeA) c = rnd() & 63;
    for (auto& c : codeB) c = rnd() & 63;
    for (int l = 0; l < 64; ++l) {
        hsa[l] = 127 - 3 + (rnd() % 5) + ((rnd() & 0xFFFFFF) << 8);  // byte 0 = exponent; the other bytes are garbage on purpose
        hsb[l] = 127 - 2 + (rnd() % 4) + ((rnd() & 0xFFFFFF) << 8);
        for (int i = 0; i < 32; ++i) {
            const uint32_t ca = codeA[(l & 15) * 128 + 32 * (l >> 4) + i], cb = codeB[(l & 15) * 128 + 32 * (l >> 4) + i];
            const int bit = 6 * i, w = bit >> 5, o = bit & 31;
            ha[l * 6 + w] |= ca << o;
            hb[l * 6 + w] |= cb << o;
            if (o > 26) {
                ha[l * 6 + w + 1] |= ca >> (32 - o);
                hb[l * 6 + w + 1] |= cb >> (32 - o);
            }
        }
    }
    uint32_t *da, *db, *dsa, *dsb;
    float* dd;
    hipMalloc(&da, ha.size() * 4);
    hipMalloc(&db, hb.size() * 4);
    hipMalloc(&dsa, 256);
    hipMalloc(&dsb, 256);
    hipMalloc(&dd, 1024);
    hipMemcpy(da, ha.data(), ha.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(db, hb.data(), hb.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dsa, hsa.data(), 256, hipMemcpyHostToDevice);
    hipMemcpy(dsb, hsb.data(), 256, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, da, db, dsa, dsb, dd);
    std::vector<float> hd(256);
    hipMemcpy(hd.data(), dd, 1024, hipMemcpyDeviceToHost);
    double worst = 0;
    for (int l = 0; l < 64; ++l)
        for (int r = 0; r < 4; ++r) {
            const int row = 4 * (l >> 4) + r, col = l & 15;
            double ref = 0;
            for (int kb = 0; kb < 4; ++kb) {
                const double sA = std::ldexp(1.0, (int)(hsa[kb * 16 + row] & 255) - 127), sB = std::ldexp(1.0, (int)(hsb[kb * 16 + col] & 255) - 127);
                for (int i = 0; i < 32; ++i) ref += sA * f6_value(codeA[row * 128 + 32 * kb + i]) * sB * f6_value(codeB[col * 128 + 32 * kb + i]);
            }
            worst = std::fmax(worst, std::fabs(ref - hd[l * 4 + r]));
        }
    printf("fp6 16x16x128 layout check: max |device - reference| = %g  (%s)\n", worst, worst == 0 ? "layout as assumed" : "MISMATCH");
    return worst == 0 ? 0 : 1;
}
