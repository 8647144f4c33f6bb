// Microbenchmark (dev tool): what the chip sustains for the MIX of the batched pass — an HBM row stream feeding matrix-core work at
// the pass's arithmetic intensity — with the operands the pass uses today (int8 rows of 384 B, v_mfma_i32_16x16x64_i8) against
// the 6-bit floating-point operands a coarser first filter would use (rows of 288 B, v_mfma_scale_f32_16x16x128_f8f6f4, e2m3).
// The batched pass of scan_i8.hip is power-bound: 2.2 Pop/s of int8 matrix work next to 4.3 TB/s of HBM traffic, whatever the
// kernel's structure (the LDS-DMA pipeline and a register-operand kernel without LDS take the same 8.9-9.2 ms per
// 100 M x 256).  The register-only probe (mfma_fp6_rate.hip) says FP6 matrix work is 1.5 x cheaper; this one asks what is left
// of that next to the stream: every wave streams its own fragments (nt loads, ring of 8 in flight) and runs M MFMAs on each —
// M = 16 is a 256-query pass (16 query groups of 16 per 16-row fragment), 8 a 128-query pass.
//   int8: fragment = 16 rows x 64 k  = 1 KiB  (16 B per lane),  384-B rows -> 6 fragments per 16 rows
//   fp6:  fragment = 16 rows x 128 k = 1.5 KiB (24 B per lane), 288-B rows -> 3 fragments per 16 rows
// Reported per configuration: ms per 100 M rows, TB/s, Pop/s.  No results are computed that mean anything.
// build: hipcc --offload-arch=gfx950 -O3 -o mfma_hbm_mix mfma_hbm_mix.hip ; run: ./mfma_hbm_mix [rows_millions=100]
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ uint32_t mix(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

__global__ void fill_kernel(uint32_t* p, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        p[i] = mix((uint32_t)i * 2654435761u + (uint32_t)(i >> 32));
}

// MODE 0: int8, MODE 1: fp6.  M: MFMAs per fragment.  Wave w takes fragments w, w + W, ...
// SHARE: that many consecutive waves of a workgroup stream the SAME fragments (a pass whose waves split the queries, not the rows:
// one read from HBM, the others from L2); NT: non-temporal loads
template <int MODE, int M, int SHARE = 1, bool NT = true>
__global__ __launch_bounds__(256) void mix_kernel(const uint32_t* __restrict__ buf, uint32_t n_frag, float* __restrict__ out) {
    constexpr int PD = 8;
    constexpr uint32_t FRAG_DW = MODE == 0 ? 256 : 384;  // dwords per fragment
    const int lane = threadIdx.x & 63;
    const uint32_t w = ((blockIdx.x * blockDim.x + threadIdx.x) >> 6) / SHARE, W = ((gridDim.x * blockDim.x) >> 6) / SHARE;
    i32x8 b8[4];
    for (int c = 0; c < 4; ++c)
        for (int j = 0; j < 8; ++j) b8[c][j] = (int)mix((w * 64 + lane) * 257u + c * 29u + j + 7u);
    f32x4 accf[4];
    i32x4 acci[4];
    for (int c = 0; c < 4; ++c) {
        accf[c] = f32x4{0.f, 0.f, 0.f, 0.f};
        acci[c] = i32x4{0, 0, 0, 0};
    }
    u32x4 ra[PD];
    u32x2 rb[PD];
    auto load = [&](int d, uint32_t f) __attribute__((always_inline)) {
        const uint32_t* p = buf + (size_t)(f < n_frag ? f : w) * FRAG_DW;
        if constexpr (NT) {
            ra[d] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p) + lane);
            if constexpr (MODE == 1) rb[d] = __builtin_nontemporal_load(reinterpret_cast<const u32x2*>(p + 256) + lane);
        } else {
            ra[d] = reinterpret_cast<const u32x4*>(p)[lane];
            if constexpr (MODE == 1) rb[d] = reinterpret_cast<const u32x2*>(p + 256)[lane];
        }
    };
    uint32_t f = w;
#pragma unroll
    for (int d = 0; d < PD; ++d) load(d, f + d * W);
    for (; f < n_frag; f += PD * W) {
#pragma unroll
        for (int d = 0; d < PD; ++d) {
            const u32x4 a = ra[d];
            [[maybe_unused]] const u32x2 a2 = rb[d];
            load(d, f + (PD + d) * W);
            if constexpr (MODE == 0) {
                const i32x4 av = {(int)a.x, (int)a.y, (int)a.z, (int)a.w};
#pragma unroll
                for (int m = 0; m < M; ++m) {
                    const i32x4 bv = {b8[m & 3][0], b8[m & 3][1], b8[m & 3][2], b8[m & 3][3]};
                    acci[m & 3] = __builtin_amdgcn_mfma_i32_16x16x64_i8(av, bv, acci[m & 3], 0, 0, 0);
                }
            } else {
                // (fp6 operands occupy 6 of the 8 dwords of the register operand)
                const i32x8 av = {(int)a.x, (int)a.y, (int)a.z, (int)a.w, (int)a2.x, (int)a2.y, 0, 0};
#pragma unroll
                for (int m = 0; m < M; ++m)
                    accf[m & 3] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(av, b8[m & 3], accf[m & 3], 2, 2, 0, 0x7F7F7F7F, 0,
                                                                                    0x7F7F7F7F);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float s = 0.f;
    for (int c = 0; c < 4; ++c) s += accf[c][0] + accf[c][1] + accf[c][2] + accf[c][3] + (float)(acci[c][0] + acci[c][1] + acci[c][2] + acci[c][3]);
    for (int d = 0; d < PD; ++d) s += (float)(ra[d].x & 1u);
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// The register-resident form of a 256-query FP6 pass: every wave holds ALL 48 query operands (16 groups x 3 k-steps x 6 dwords = 288
// registers), streams its own tiles (3 fragments = 16 rows) and runs 16 MFMAs per fragment into 16 accumulators that are folded
// (a max per group: the pass's threshold test costs about that) and cleared per tile.  One wave per SIMD.
template <int PDT>
__global__ __launch_bounds__(256) void fp6_resident_kernel(const uint32_t* __restrict__ buf, uint32_t n_tiles, float* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const uint32_t w = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, W = (gridDim.x * blockDim.x) >> 6;
    i32x8 bq[16][3];
#pragma unroll
    for (int g = 0; g < 16; ++g)
#pragma unroll
        for (int ks = 0; ks < 3; ++ks)
#pragma unroll
            for (int j = 0; j < 8; ++j) bq[g][ks][j] = j < 6 ? (int)mix((w * 64 + lane) * 257u + g * 29u + ks * 7u + j + 7u) : 0;
    u32x4 ra[PDT][3];
    u32x2 rb[PDT][3];
    auto load = [&](int d, uint32_t t) __attribute__((always_inline)) {
        const uint32_t* p = buf + (size_t)(t < n_tiles ? t : w) * (3 * 384);
#pragma unroll
        for (int ks = 0; ks < 3; ++ks) {
            ra[d][ks] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p + ks * 384) + lane);
            rb[d][ks] = __builtin_nontemporal_load(reinterpret_cast<const u32x2*>(p + ks * 384 + 256) + lane);
        }
    };
    float best = 0.f;
    uint32_t t = w;
#pragma unroll
    for (int d = 0; d < PDT; ++d) load(d, t + d * W);
    for (; t < n_tiles; t += PDT * W) {
#pragma unroll
        for (int d = 0; d < PDT; ++d) {
            f32x4 acc[16];
#pragma unroll
            for (int ks = 0; ks < 3; ++ks) {
                const i32x8 av = {(int)ra[d][ks].x, (int)ra[d][ks].y, (int)ra[d][ks].z, (int)ra[d][ks].w, (int)rb[d][ks].x, (int)rb[d][ks].y, 0, 0};
#pragma unroll
                for (int g = 0; g < 16; ++g)
                    acc[g] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(av, bq[g][ks], ks == 0 ? f32x4{0.f, 0.f, 0.f, 0.f} : acc[g], 2, 2, 0,
                                                                               0x7F7F7F7F, 0, 0x7F7F7F7F);
            }
            load(d, t + (PDT + d) * W);
#pragma unroll
            for (int g = 0; g < 16; ++g) best = fmaxf(best, fmaxf(fmaxf(acc[g][0], acc[g][1]), fmaxf(acc[g][2], acc[g][3])));
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float s = best;
    for (int d = 0; d < PDT; ++d) s += (float)(ra[d][0].x & 1u);
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int PDT>
void run_resident(const uint32_t* buf, double rows, float* out, const char* name) {
    const uint32_t n_tiles = (uint32_t)(rows / 16.0);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    double best = 1e30, sum = 0;
    const int reps = 4;
    for (int r = 0; r < reps + 1; ++r) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL((fp6_resident_kernel<PDT>), dim3(256), dim3(256), 0, 0, buf, n_tiles, out);
        hipEventRecord(e1, 0);
        hipDeviceSynchronize();
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        if (r == 0) continue;
        best = ms < best ? ms : best;
        sum += ms;
    }
    const double bytes = (double)n_tiles * 4608.0, ops = (double)n_tiles * 48 * 2.0 * 16 * 16 * 128;
    printf("%-34s PDT=%d: best %7.3f mean %7.3f ms per %.0f M rows -> %5.2f TB/s, %5.2f Pop/s\n", name, PDT, best, sum / reps, rows / 1e6,
           bytes / (best * 1e-3) / 1e12, ops / (best * 1e-3) / 1e15);
    fflush(stdout);
}

template <int MODE, int M, int SHARE = 1, bool NT = true>
void run(const uint32_t* buf, double rows, float* out, const char* name) {
    const double row_bytes = MODE == 0 ? 384.0 : 288.0;
    const uint32_t n_frag = (uint32_t)(rows * row_bytes / (MODE == 0 ? 1024.0 : 1536.0));
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    double best = 1e30, sum = 0;
    const int reps = 4;
    for (int r = 0; r < reps + 1; ++r) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL((mix_kernel<MODE, M, SHARE, NT>), dim3(256), dim3(256), 0, 0, buf, n_frag, out);
        hipEventRecord(e1, 0);
        hipDeviceSynchronize();
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        if (r == 0) continue;
        best = ms < best ? ms : best;
        sum += ms;
    }
    const double bytes = (double)n_frag * (MODE == 0 ? 1024.0 : 1536.0);
    const double ops = (double)n_frag * M * SHARE * 2.0 * 16 * 16 * (MODE == 0 ? 64 : 128);
    printf("%-34s M=%2d : best %7.3f mean %7.3f ms per %.0f M rows -> %5.2f TB/s, %5.2f Pop/s\n", name, M, best, sum / reps, rows / 1e6,
           bytes / (best * 1e-3) / 1e12, ops / (best * 1e-3) / 1e15);
    fflush(stdout);
}

int main(int argc, char** argv) {
    const double rows = (argc > 1 ? atof(argv[1]) : 100.0) * 1e6;
    uint32_t* buf;
    float* out;
    const size_t bytes = (size_t)(rows * 384.0) + (1 << 20);
    if (hipMalloc(&buf, bytes) != hipSuccess || hipMalloc(&out, 256 * 256 * 4) != hipSuccess) {
        printf("allocation failed\n");
        return 1;
    }
    hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, 0, buf, bytes / 4);
    hipDeviceSynchronize();
    for (int rep = 0; rep < 2; ++rep) {
        run<0, 0>(buf, rows, out, "int8 rows (384 B), stream only");
        run<0, 8>(buf, rows, out, "int8 16x16x64, 128 queries");
        run<0, 16>(buf, rows, out, "int8 16x16x64, 256 queries");
        run<1, 0>(buf, rows, out, "fp6 rows (288 B), stream only");
        run<1, 8>(buf, rows, out, "fp6 16x16x128, 128 queries");
        run<1, 16>(buf, rows, out, "fp6 16x16x128, 256 queries");
        run_resident<2>(buf, rows, out, "fp6 256 q resident operands");
        run_resident<3>(buf, rows, out, "fp6 256 q resident operands");
        run_resident<4>(buf, rows, out, "fp6 256 q resident operands");
        run<1, 8, 2, true>(buf, rows, out, "fp6 256 q, 2 waves share, nt");
        run<1, 8, 2, false>(buf, rows, out, "fp6 256 q, 2 waves share");
        run<1, 4, 4, true>(buf, rows, out, "fp6 256 q, 4 waves share, nt");
        run<1, 4, 4, false>(buf, rows, out, "fp6 256 q, 4 waves share");
        run<0, 8, 2, false>(buf, rows, out, "int8 256 q, 2 waves share");
        run<0, 4, 4, false>(buf, rows, out, "int8 256 q, 4 waves share");
    }
    return 0;
}
