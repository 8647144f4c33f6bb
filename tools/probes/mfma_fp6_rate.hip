// Microbenchmark (dev tool): chip-wide sustained rate of the block-scaled f8f6f4 matrix-core instruction with FP6 (e2m3), FP4 and
// FP8 operands in registers, next to v_mfma_i32_16x16x64_i8 — what a 6-bit first filter of the batched pass could count on under
// the chip's power limit (the int8 pass of scan_i8.hip sustains 2.2 Pop/s next to its HBM stream, 3.98 Pop/s bare).
// build: hipcc --offload-arch=gfx950 -O3 -o mfma_fp6_rate mfma_fp6_rate.hip ; run: ./mfma_fp6_rate
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint32_t mix(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

// FMT: 0 fp8 e4m3, 2 fp6 e2m3, 4 fp4 e2m1 (cbsz / blgp of the instruction); -1: v_mfma_i32_16x16x64_i8
template <int FMT>
__global__ __launch_bounds__(512) void probe(float* out, int iters) {
    const uint32_t id = blockIdx.x * blockDim.x + threadIdx.x;
    float s = 0.f;
    if constexpr (FMT >= 0) {
        i32x8 a[4], b[4];
        for (int c = 0; c < 4; ++c)
            for (int j = 0; j < 8; ++j) {
                // (keep exponents small: fp8 e4m3 bytes with the top exponent bits cleared, so nothing overflows to inf / NaN)
                a[c][j] = (int)(mix(id * 131u + c * 17u + j) & (FMT == 0 ? 0xB7B7B7B7u : 0xFFFFFFFFu));
                b[c][j] = (int)(mix(id * 257u + c * 29u + j + 7u) & (FMT == 0 ? 0xB7B7B7B7u : 0xFFFFFFFFu));
            }
        f32x4 acc[4];
        for (int c = 0; c < 4; ++c) acc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int c = 0; c < 4; ++c)
                    acc[c] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a[(c + r) & 3], b[c], acc[c], FMT, FMT, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
        }
        for (int c = 0; c < 4; ++c) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
    } else {
        i32x4 a[4], b[4];
        for (int c = 0; c < 4; ++c)
            for (int j = 0; j < 4; ++j) {
                a[c][j] = (int)mix(id * 131u + c * 17u + j);
                b[c][j] = (int)mix(id * 257u + c * 29u + j + 7u);
            }
        i32x4 acc[4];
        for (int c = 0; c < 4; ++c) acc[c] = i32x4{0, 0, 0, 0};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int r = 0; r < 8; ++r)  // 8 x 4 x 16384 MACs = the 4 x 4 x 32768 of the scaled form
#pragma unroll
                for (int c = 0; c < 4; ++c) acc[c] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[(c + r) & 3], b[c], acc[c], 0, 0, 0);
        }
        for (int c = 0; c < 4; ++c) s += (float)(acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3]);
    }
    out[id] = s;
}

template <int FMT>
void run(int threads, int blocks, const char* name) {
    float* out;
    hipMalloc(&out, (size_t)blocks * 512 * 4);
    const int iters = 20000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL((probe<FMT>), dim3(blocks), dim3(threads), 0, 0, out, iters);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((probe<FMT>), dim3(blocks), dim3(threads), 0, 0, out, iters);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double waves = (double)blocks * threads / 64.0;
    const double macs = waves * iters * 16.0 * 32768.0;  // 16 x (16 x 16 x 128) MACs per iteration and wave
    printf("%-22s blocks=%3d waves/SIMD=%d : %8.2f ms -> %7.1f Top/s\n", name, blocks, threads / 256, ms, 2.0 * macs / (ms * 1e-3) / 1e12);
    fflush(stdout);
    hipFree(out);
}

int main() {
    for (int rep = 0; rep < 2; ++rep)
        for (int threads : {256, 512}) {
            run<-1>(threads, 256, "i8 16x16x64");
            run<0>(threads, 256, "fp8 e4m3 16x16x128");
            run<2>(threads, 256, "fp6 e2m3 16x16x128");
            run<4>(threads, 256, "fp4 e2m1 16x16x128");
        }
    run<-1>(256, 32, "i8 16x16x64");
    run<2>(256, 32, "fp6 e2m3 16x16x128");
    run<4>(256, 32, "fp4 e2m1 16x16x128");
    return 0;
}
