// Probe (dev tool): v_dot2_f32_f16 numerics on gfx950 against a float reference.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
__global__ void k(const unsigned* a, const unsigned* b, float* out, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float acc = 0.f;
    for (int j = 0; j < 4; ++j)
        acc = __builtin_amdgcn_fdot2(__builtin_bit_cast(half2_t, a[i * 4 + j]), __builtin_bit_cast(half2_t, b[i * 4 + j]), acc, false);
    out[i] = acc;
}
static unsigned short f2h(float f) { _Float16 h = (_Float16)f; unsigned short u; __builtin_memcpy(&u, &h, 2); return u; }
static float h2f(unsigned short u) { _Float16 h; __builtin_memcpy(&h, &u, 2); return (float)h; }
int main() {
    const int n = 4096;
    std::vector<unsigned> a(n * 4), b(n * 4);
    std::vector<float> ref(n);
    srand(1);
    for (int i = 0; i < n; ++i) {
        double s = 0;
        for (int j = 0; j < 4; ++j) {
            float x0 = 256.f * 0.2f * (rand() / (float)RAND_MAX - 0.5f), x1 = 256.f * 0.2f * (rand() / (float)RAND_MAX - 0.5f);
            float y0 = 256.f * 0.2f * (rand() / (float)RAND_MAX - 0.5f), y1 = 256.f * 0.2f * (rand() / (float)RAND_MAX - 0.5f);
            if (i % 7 == 0) x0 *= 1e-4f;  // small values
            unsigned short hx0 = f2h(x0), hx1 = f2h(x1), hy0 = f2h(y0), hy1 = f2h(y1);
            a[i * 4 + j] = hx0 | (hx1 << 16);
            b[i * 4 + j] = hy0 | (hy1 << 16);
            s += (double)h2f(hx0) * h2f(hy0) + (double)h2f(hx1) * h2f(hy1);
        }
        ref[i] = (float)s;
    }
    unsigned *da, *db; float* dout;
    hipMalloc(&da, n * 16); hipMalloc(&db, n * 16); hipMalloc(&dout, n * 4);
    hipMemcpy(da, a.data(), n * 16, hipMemcpyHostToDevice); hipMemcpy(db, b.data(), n * 16, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, da, db, dout, n);
    std::vector<float> out(n);
    hipMemcpy(out.data(), dout, n * 4, hipMemcpyDeviceToHost);
    double maxrel = 0; int bad = 0;
    for (int i = 0; i < n; ++i) {
        double e = fabs(out[i] - ref[i]) / (fabs(ref[i]) + 1.0);
        if (e > maxrel) maxrel = e;
        if (e > 1e-3 && bad < 5) { printf("i=%d got %g ref %g\n", i, out[i], ref[i]); ++bad; }
    }
    printf("max relative error %.3g\n", maxrel);
    return 0;
}
