// Microbenchmark (dev tool): cycles per v_mfma_f32_32x32x16_f16 for a dependent chain, 1 or 2 waves per SIMD,
// with/without one ds_read_b128 (operand from LDS) per MFMA and with a ring of PD reads in flight.
// build: hipcc --offload-arch=gfx950 -O3 -o mfma_probe mfma_probe.hip ; run: ./mfma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int MODE, int PD>
__global__ __launch_bounds__(512) void probe(float* out, unsigned long long* cyc, int iters) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 30720 / 4; i += blockDim.x) reinterpret_cast<float*>(lds)[i] = 0.001f * (i & 255);
    __syncthreads();
    half8 b;
    for (int j = 0; j < 8; ++j) b[j] = (_Float16)(0.01f * (lane + j));
    f32x16 acc;
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    const unsigned char* ab = lds + (lane >> 5) * 656 + (lane & 31) * 16;
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {  // registers only
            half8 a = b;
#pragma unroll
            for (int s = 0; s < 24; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
        } else {  // operand ring from LDS
            half8 a[PD];
#pragma unroll
            for (int d = 0; d < PD; ++d) a[d] = *reinterpret_cast<const half8*>(ab + d * 1280 + (d & 3) * 32);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int s = 0; s < 24; ++s) {
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[s % PD], b, acc, 0, 0, 0);
                if (s + PD < 24) a[s % PD] = *reinterpret_cast<const half8*>(ab + (s + PD) * 1280 + ((s + PD) & 3) * 32);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    float s = 0;
    for (int e = 0; e < 16; ++e) s += acc[e];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (lane == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}

template <int MODE, int PD>
void run(const char* name, int threads) {
    float* out;
    unsigned long long* cyc;
    hipMalloc(&out, 256 * 512 * 4);
    hipMalloc(&cyc, 256 * 8 * 8);
    hipMemset(cyc, 0, 256 * 8 * 8);
    const int iters = 2000;
    hipFuncSetAttribute((const void*)probe<MODE, PD>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((probe<MODE, PD>), dim3(256), dim3(threads), 65536, 0, out, cyc, iters);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(256 * 8);
    hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
    double s0 = 0, s1 = 0;
    int nw = threads / 64;
    for (int b = 0; b < 256; ++b)
        for (int w = 0; w < nw; ++w) (w < 4 ? s0 : s1) += (double)h[b * 8 + w];
    printf("%-34s waves/SIMD=%d  cycles per MFMA per wave: waves0-3 %.1f", name, nw / 4, s0 / (256.0 * 4) / iters / 24);
    if (nw > 4) printf("  waves4-7 %.1f", s1 / (256.0 * 4) / iters / 24);
    printf("\n");
    hipFree(out);
    hipFree(cyc);
}

int main() {
    run<0, 4>("registers only", 256);
    run<0, 4>("registers only", 512);
    run<1, 4>("LDS operand ring PD=4", 256);
    run<1, 4>("LDS operand ring PD=4", 512);
    run<1, 8>("LDS operand ring PD=8", 256);
    run<1, 8>("LDS operand ring PD=8", 512);
    return 0;
}
