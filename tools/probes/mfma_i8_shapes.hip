// Microbenchmark (dev tool): chip-wide int8 matrix-core rate by MFMA shape, operands in registers, pseudo-random data:
// v_mfma_i32_32x32x32_i8 (32768 MACs) against v_mfma_i32_16x16x64_i8 (16384 MACs) — does the smaller shape hold a higher
// clock under the power limit, as the bf16 pair does?  One or two waves per SIMD, four independent accumulator chains.
// build: hipcc --offload-arch=gfx950 -O3 -o mfma_i8_shapes mfma_i8_shapes.hip ; run: ./mfma_i8_shapes
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ uint32_t mix(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

template <int SHAPE>  // 0: 32x32x32, 1: 16x16x64
__global__ __launch_bounds__(512) void probe(int* out, int iters) {
    const uint32_t id = blockIdx.x * blockDim.x + threadIdx.x;
    i32x4 a[4], b[4];
    for (int c = 0; c < 4; ++c)
        for (int j = 0; j < 4; ++j) {
            a[c][j] = (int)mix(id * 131u + c * 17u + j);
            b[c][j] = (int)mix(id * 257u + c * 29u + j + 7u);
        }
    int s = 0;
    if constexpr (SHAPE == 0) {
        i32x16 acc[4];
        for (int c = 0; c < 4; ++c)
            for (int e = 0; e < 16; ++e) acc[c][e] = 0;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int c = 0; c < 4; ++c) acc[c] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[(c + r) & 3], b[c], acc[c], 0, 0, 0);
        }
        for (int c = 0; c < 4; ++c)
            for (int e = 0; e < 16; ++e) s += acc[c][e];
    } else {
        i32x4 acc[4];
        for (int c = 0; c < 4; ++c)
            for (int e = 0; e < 4; ++e) acc[c][e] = 0;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int r = 0; r < 8; ++r)
#pragma unroll
                for (int c = 0; c < 4; ++c) acc[c] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[(c + r) & 3], b[c], acc[c], 0, 0, 0);
        }
        for (int c = 0; c < 4; ++c)
            for (int e = 0; e < 4; ++e) s += acc[c][e];
    }
    out[id] = s;
}

template <int SHAPE>
void run(int threads, int blocks) {
    int* out;
    hipMalloc(&out, (size_t)blocks * 512 * 4);
    const int iters = 40000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL((probe<SHAPE>), dim3(blocks), dim3(threads), 0, 0, out, iters);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((probe<SHAPE>), dim3(blocks), dim3(threads), 0, 0, out, iters);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double waves = (double)blocks * threads / 64.0;
    const double macs = waves * iters * 16.0 * 32768.0;  // both bodies: 16 x 32768 MACs per iteration and wave
    printf("%s blocks=%3d waves/SIMD=%d : %8.2f ms -> %7.1f Top/s (dense int8 peak 5000)\n", SHAPE == 0 ? "32x32x32_i8" : "16x16x64_i8", blocks,
           threads / 256, ms, 2.0 * macs / (ms * 1e-3) / 1e12);
    hipFree(out);
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
