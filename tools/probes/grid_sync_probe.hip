// Microbenchmark (dev tool): what an all-to-all hand-off between the phases of a small dependent chain costs on this chip —
// (a) a kernel boundary (the phases as separate launches, replayed from a hipGraph: what one text's forward pass pays 32 times),
// (b) a grid barrier inside ONE launch (atomic arrive + spin, bounded) over W workgroups, all on one XCD (workgroup b runs on XCD
//     b % 8: only b % 8 == 0 take part) or spread over the eight XCDs.
// Every phase reads what ALL workgroups wrote in the phase before (a real all-to-all dependency), adds `work` FMAs per thread.
// build: hipcc --offload-arch=gfx950 -O3 -o grid_sync_probe grid_sync_probe.hip ; run: ./grid_sync_probe
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

constexpr int MAXW = 256;

__device__ __forceinline__ float phase_work(const float* prev, int n_active, int work, int wg) {
    // read every workgroup's value of the previous phase (device-scope loads: another CU wrote them)
    float s = 0.f;
    for (int i = threadIdx.x; i < n_active; i += blockDim.x) s += __hip_atomic_load(prev + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    for (int o = 32; o >= 1; o >>= 1) s += __shfl_xor(s, o);
    float x = s * 1e-3f + (float)wg;
    for (int i = 0; i < work; ++i) x = __builtin_fmaf(x, 1.0000001f, 1e-7f);
    return x;
}

// one phase as a kernel of its own
__global__ __launch_bounds__(256) void phase_kernel(const float* prev, float* cur, int n_active, int work, int one_xcd) {
    int wg = blockIdx.x;
    if (one_xcd) {
        if (wg % 8) return;
        wg /= 8;
    }
    if (wg >= n_active) return;
    const float x = phase_work(prev, n_active, work, wg);
    if (threadIdx.x == 0) __hip_atomic_store(cur + wg, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// all phases in one launch, a bounded spin barrier between them; err != 0: a barrier timed out
__global__ __launch_bounds__(256) void chain_kernel(float* buf, uint32_t* counters, int n_phase, int n_active, int work, int one_xcd, uint32_t* err) {
    int wg = blockIdx.x;
    if (one_xcd) {
        if (wg % 8) return;
        wg /= 8;
    }
    if (wg >= n_active) return;
    for (int p = 0; p < n_phase; ++p) {
        const float x = phase_work(buf + (size_t)(p & 1) * MAXW, n_active, work, wg);
        if (threadIdx.x == 0) {
            __hip_atomic_store(buf + (size_t)((p + 1) & 1) * MAXW + wg, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __threadfence();
            __hip_atomic_fetch_add(counters + p, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            uint32_t spins = 0;
            while (__hip_atomic_load(counters + p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < (uint32_t)n_active) {
                if (++spins > 20000000u) {  // (every workgroup is resident: this only trips on a bug)
                    *err = 1u;
                    break;
                }
            }
        }
        __syncthreads();
    }
}

int main() {
    float* buf;
    uint32_t *counters, *err;
    hipMalloc(&buf, 2 * MAXW * sizeof(float));
    hipMalloc(&counters, 4096 * sizeof(uint32_t));
    hipMalloc(&err, 4);
    hipMemset(buf, 0, 2 * MAXW * sizeof(float));
    hipMemset(err, 0, 4);
    hipStream_t s;
    hipStreamCreate(&s);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int n_phase = 32;
    for (int work : {0, 2000}) {
        for (int one_xcd : {1, 0}) {
            for (int n_active : {12, 32, 96, 256}) {
                if (one_xcd && n_active > 32) continue;
                const int grid = one_xcd ? n_active * 8 : n_active;
                // (a) a graph of n_phase launches
                hipGraph_t g;
                hipGraphExec_t ge;
                hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
                for (int p = 0; p < n_phase; ++p)
                    hipLaunchKernelGGL(phase_kernel, dim3(grid), dim3(256), 0, s, buf + (size_t)(p & 1) * MAXW, buf + (size_t)((p + 1) & 1) * MAXW,
                                       n_active, work, one_xcd);
                hipStreamEndCapture(s, &g);
                hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
                float best_a = 1e30f, best_b = 1e30f;
                for (int r = 0; r < 12; ++r) {
                    hipEventRecord(e0, s);
                    hipGraphLaunch(ge, s);
                    hipEventRecord(e1, s);
                    hipStreamSynchronize(s);
                    float ms;
                    hipEventElapsedTime(&ms, e0, e1);
                    if (r >= 2 && ms < best_a) best_a = ms;
                }
                // (b) one launch with barriers
                for (int r = 0; r < 12; ++r) {
                    hipMemsetAsync(counters, 0, n_phase * sizeof(uint32_t), s);
                    hipEventRecord(e0, s);
                    hipLaunchKernelGGL(chain_kernel, dim3(grid), dim3(256), 0, s, buf, counters, n_phase, n_active, work, one_xcd, err);
                    hipEventRecord(e1, s);
                    hipStreamSynchronize(s);
                    float ms;
                    hipEventElapsedTime(&ms, e0, e1);
                    if (r >= 2 && ms < best_b) best_b = ms;
                }
                uint32_t h_err = 0;
                hipMemcpy(&h_err, err, 4, hipMemcpyDeviceToHost);
                printf("work=%4d FMAs  %3d workgroups %-10s: %d phases as launches of a graph %7.2f us (%.2f per phase); as one launch with barriers %7.2f us (%.2f per phase)%s\n",
                       work, n_active, one_xcd ? "on one XCD" : "on 8 XCDs", n_phase, best_a * 1e3f, best_a * 1e3f / n_phase, best_b * 1e3f,
                       best_b * 1e3f / n_phase, h_err ? "  BARRIER TIMED OUT" : "");
                fflush(stdout);
                hipGraphExecDestroy(ge);
                hipGraphDestroy(g);
            }
        }
    }
    return 0;
}
