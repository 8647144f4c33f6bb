// Microbenchmark (dev tool): how fast does ONE SIMD retire v_mfma_f32_32x32x16_f16, measured in wall time with the
// whole chip busy (power-limited clock included)?  Patterns: 1 or 2 waves per SIMD x 1, 2 or 4 independent accumulator
// chains per wave, operands in registers.  Prints ns per MFMA per SIMD and the chip-wide TFLOP/s it implies.
// build: hipcc --offload-arch=gfx950 -O3 -o mfma_rate mfma_rate.hip ; run: ./mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int CHAINS>
__global__ __launch_bounds__(512) void probe(float* out, int iters) {
    const int lane = threadIdx.x & 63;
    half8 a, b[CHAINS];
    for (int j = 0; j < 8; ++j) a[j] = (_Float16)(0.001f * (lane + j));
    for (int c = 0; c < CHAINS; ++c)
        for (int j = 0; j < 8; ++j) b[c][j] = (_Float16)(0.002f * (lane + j + c));
    f32x16 acc[CHAINS];
    for (int c = 0; c < CHAINS; ++c)
        for (int e = 0; e < 16; ++e) acc[c][e] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 0; s < 16; ++s) {
#pragma unroll
            for (int c = 0; c < CHAINS; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b[c], acc[c], 0, 0, 0);
        }
    }
    float s = 0;
    for (int c = 0; c < CHAINS; ++c)
        for (int e = 0; e < 16; ++e) s += acc[c][e];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// the same for v_mfma_f32_32x32x2_f32 (the embedder's GEMMs): 4096 flop per instruction
template <int CHAINS>
__global__ __launch_bounds__(512) void probe_f32(float* out, int iters) {
    const int lane = threadIdx.x & 63;
    float a = 0.001f * lane, b[CHAINS];
    for (int c = 0; c < CHAINS; ++c) b[c] = 0.002f * (lane + c);
    f32x16 acc[CHAINS];
    for (int c = 0; c < CHAINS; ++c)
        for (int e = 0; e < 16; ++e) acc[c][e] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 0; s < 16; ++s) {
#pragma unroll
            for (int c = 0; c < CHAINS; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b[c], acc[c], 0, 0, 0);
        }
    }
    float s = 0;
    for (int c = 0; c < CHAINS; ++c)
        for (int e = 0; e < 16; ++e) s += acc[c][e];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

void run_f32(int threads, int blocks) {
    float* out;
    hipMalloc(&out, (size_t)blocks * 512 * 4);
    const int iters = 5000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL((probe_f32<2>), dim3(blocks), dim3(threads), 0, 0, out, iters);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((probe_f32<2>), dim3(blocks), dim3(threads), 0, 0, out, iters);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double mfma_per_simd = (double)iters * 16 * 2 * (threads / 256.0);
    printf("f32 32x32x2: blocks=%3d waves/SIMD=%d : %.1f ns per MFMA per SIMD -> %.1f TFLOP/s on %d CUs\n", blocks, threads / 256,
           ms * 1e6 / mfma_per_simd, 4096.0 * mfma_per_simd * 4 * blocks / (ms * 1e-3) / 1e12, blocks);
    hipFree(out);
}

template <int CHAINS>
void run(int threads, int blocks) {
    float* out;
    hipMalloc(&out, (size_t)blocks * 512 * 4);
    const int iters = 20000 / CHAINS;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL((probe<CHAINS>), dim3(blocks), dim3(threads), 0, 0, out, iters);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((probe<CHAINS>), dim3(blocks), dim3(threads), 0, 0, out, iters);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double waves_per_simd = threads / 256.0;
    const double mfma_per_simd = (double)iters * 16 * CHAINS * waves_per_simd;
    const double ns = ms * 1e6 / mfma_per_simd;
    const double tf = 32768.0 * mfma_per_simd * 4 * blocks / (ms * 1e-3) / 1e12;
    printf("blocks=%3d waves/SIMD=%d chains/wave=%d : %.2f ms  %.1f ns per MFMA per SIMD  -> %.0f TFLOP/s on %d CUs\n", blocks,
           (int)waves_per_simd, CHAINS, ms, ns, tf, blocks);
    hipFree(out);
}

int main() {
    for (int blocks : {256, 32}) {
        run<1>(256, blocks);
        run<2>(256, blocks);
        run<4>(256, blocks);
        run<1>(512, blocks);
        run<2>(512, blocks);
    }
    run_f32(256, 256);
    run_f32(512, 256);
    run_f32(256, 32);
    return 0;
}
