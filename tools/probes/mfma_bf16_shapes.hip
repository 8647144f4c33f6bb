// Microbenchmark (dev tool): chip-wide bf16 matrix-core rate by MFMA shape, operands in registers, pseudo-random data:
// v_mfma_f32_32x32x16_bf16 against v_mfma_f32_16x16x32_bf16 (see mfma_i8_shapes.hip).
// build: hipcc --offload-arch=gfx950 -O3 -o mfma_bf16_shapes mfma_bf16_shapes.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint32_t mix(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

template <int SHAPE>
__global__ __launch_bounds__(512) void probe(float* out, int iters) {
    const uint32_t id = blockIdx.x * blockDim.x + threadIdx.x;
    bf16x8 a[4], b[4];
    for (int c = 0; c < 4; ++c) {
        u32x4 ua, ub;
        for (int j = 0; j < 4; ++j) {  // bf16 pairs with exponents near 1.0: finite, varied mantissas
            ua[j] = (mix(id * 131u + c * 17u + j) & 0x807F807Fu) | 0x3F003F00u;
            ub[j] = (mix(id * 257u + c * 29u + j + 7u) & 0x807F807Fu) | 0x3F003F00u;
        }
        a[c] = __builtin_bit_cast(bf16x8, ua);
        b[c] = __builtin_bit_cast(bf16x8, ub);
    }
    float s = 0;
    if constexpr (SHAPE == 0) {
        f32x16 acc[4];
        for (int c = 0; c < 4; ++c)
            for (int e = 0; e < 16; ++e) acc[c][e] = 0;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int c = 0; c < 4; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[(c + r) & 3], b[c], acc[c], 0, 0, 0);
        }
        for (int c = 0; c < 4; ++c)
            for (int e = 0; e < 16; ++e) s += acc[c][e];
    } else {
        f32x4 acc[4];
        for (int c = 0; c < 4; ++c)
            for (int e = 0; e < 4; ++e) acc[c][e] = 0;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int r = 0; r < 8; ++r)
#pragma unroll
                for (int c = 0; c < 4; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[(c + r) & 3], b[c], acc[c], 0, 0, 0);
        }
        for (int c = 0; c < 4; ++c)
            for (int e = 0; e < 4; ++e) s += acc[c][e];
    }
    out[id] = s;
}

template <int SHAPE>
void run(int threads, int blocks) {
    float* out;
    (void)hipMalloc(&out, (size_t)blocks * 512 * 4);
    const int iters = 40000;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((probe<SHAPE>), dim3(blocks), dim3(threads), 0, 0, out, iters);
    (void)hipEventRecord(e0, 0);
    hipLaunchKernelGGL((probe<SHAPE>), dim3(blocks), dim3(threads), 0, 0, out, iters);
    (void)hipEventRecord(e1, 0);
    (void)hipDeviceSynchronize();
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double waves = (double)blocks * threads / 64.0;
    const double macs = waves * iters * 16.0 * 16384.0;  // both bodies: 16 x 16384 MACs per iteration and wave
    printf("%s blocks=%3d waves/SIMD=%d : %8.2f ms -> %7.1f TFLOP/s (dense bf16 peak 2500)\n", SHAPE == 0 ? "32x32x16_bf16" : "16x16x32_bf16",
           blocks, threads / 256, ms, 2.0 * macs / (ms * 1e-3) / 1e12);
    (void)hipFree(out);
}

int main() {
    for (int rep = 0; rep < 2; ++rep)
        for (int threads : {256, 512}) {
            run<0>(threads, 256);
            run<1>(threads, 256);
        }
    run<0>(256, 32);
    run<1>(256, 32);
    return 0;
}
