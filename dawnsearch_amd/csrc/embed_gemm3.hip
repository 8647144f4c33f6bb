// embed_gemm3.hip — f32-accurate dense layers on the bf16 matrix cores (3-way split, 6 products).
//
// model.rs:53-64  y = x . W^T + b  is an f32 contraction.  v_mfma_f32_32x32x2_f32 does it at 157 TFLOP/s peak; the bf16
// matrix cores run 16x that.  Every f32 value is the exact sum of three bf16 values up to 2^-24 of its magnitude:
//     a = a1 + a2 + a3,   a1 = bf16(a), a2 = bf16(a - a1), a3 = bf16(a - a1 - a2)        (the subtractions are exact)
// so  a b = sum_{i,j} a_i b_j, and the six products with i + j <= 4 carry everything down to 3 x 2^-24 |a b| — the size
// of one f32 rounding.  Each of them is an EXACT f32 number (8-bit x 8-bit significands), accumulated in f32 by
// v_mfma_f32_32x32x16_bf16: six MFMAs of 8 passes (192 cycles) do the work of eight f32 MFMAs of 16 passes (512 cycles)
// for one 32x32 tile and 16 values of k.  The result differs from the f32-MFMA kernel (embed_kernels.hip) by the order of
// the f32 additions and the three dropped product classes: ~1e-7 relative — the parity bar is 1e-5 on the unit embeddings.
//
// Operands travel as PLANES: three bf16 arrays per matrix, K-blocked (plane_index in embed_kernels.hpp) (weights are split once at load time; activations are
// written as planes by the kernel that produces them — a consumer tile would otherwise re-split every element N/64 times).
// A 64x64 block tile, four waves of 32x32; per K-step of 32 each operand is three 4-KiB plane tiles that go HBM/L2 -> LDS by
// LDS-DMA (global_load_lds_dwordx4: no staging registers, no ds_write; 6 instructions per wave and step) into a double
// buffer; 16-B chunk c of row r sits at slot c ^ ((r >> 2) & 3) of its 64-B row, so that the ds_read_b128 of an MFMA
// operand (32 rows x one chunk) is conflict-free: the instruction is served in four groups of 16 lanes — lanes {0-3, 12-15,
// 20-27}, {4-11, 16-19, 28-31} and the same + 32 — and the four rows of a group that share r & 3 (one 64-B quarter of the
// 256-B bank row) differ in (r >> 2) & 3.  (With (r >> 1) & 3 every read was a 2-way conflict: SQ_LDS_BANK_CONFLICT was 12 %
// of the kernel's cycles.)
#include "embed_kernels.hpp"
#include "wave_topk.hpp"

namespace dawn {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// f32 [rows][K] row-major -> three K-blocked bf16 planes (plane_index)
__global__ void split_planes_kernel(const float* __restrict__ in, uint16_t* __restrict__ planes, int rows, int K, size_t rows_alloc) {
    const size_t n2 = (size_t)rows * K / 2, plane_stride = rows_alloc * K;  // K is even: a pair never straddles a row
    for (size_t j = blockIdx.x * (size_t)blockDim.x + threadIdx.x; j < n2; j += (size_t)gridDim.x * blockDim.x) {
        const size_t i = 2 * j;
        const float2 f = *reinterpret_cast<const float2*>(in + i);
        uint32_t w[3];
        split3_bf16_pair(f.x, f.y, w[0], w[1], w[2]);
        const size_t o = plane_index(i / K, (int)(i % K), rows_alloc);  // (k even: the pair is adjacent in the plane too)
#pragma unroll
        for (int p = 0; p < 3; ++p) *reinterpret_cast<uint32_t*>(planes + p * plane_stride + o) = w[p];
    }
}

void launch_split_planes(const float* in, uint16_t* planes, int rows, int K, size_t rows_alloc, hipStream_t s) {
    const size_t n = (size_t)rows * K / 2;
    if (n == 0) return;
    size_t blocks = (n + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(split_planes_kernel, dim3((unsigned)blocks), dim3(256), 0, s, in, planes, rows, K, rows_alloc);
}

constexpr int G3T = 64;                 // block tile (rows of A, rows of W)
constexpr int G3K = 32;                 // K-step
constexpr int G3_PLANE = G3T * G3K * 2;  // 4096 B: one plane tile
constexpr int G3_BUF = 6 * G3_PLANE;    // A planes | W planes

__device__ __forceinline__ float act_apply3(float v, int act) {
    if (act == 1) {  // tanh-GELU, model.rs:31-35 (see act_apply in embed_kernels.hip)
        const float k = 0.7978845608028654f;
        const float u = k * v * (1.0f + 0.044715f * v * v);
        const float t = fminf(fmaxf(-2.0f * u, -80.0f), 80.0f);
        return v / (1.0f + __expf(t));
    }
    if (act == 2) return v > 0.f ? v : 0.f;
    return v;
}

// Y[M,N] (f32, may be NULL) and / or Yp (K-blocked planes of a [y_plane / N x N] operand, may be NULL) = act(A . W^T + bias)
//   Ap: K-blocked planes of a [a_plane / K x K] operand (a_plane = elements per plane), Wp: of [N x K] (embed_kernels.hpp)
// M arbitrary (rows past M are clamped on load and not stored), N % 64 == 0, K % 32 == 0.
// A ring of STAGES K-step images: the DMA of step i + STAGES - 1 is issued at step i (after the barrier that says
// everyone left step i - 1, whose image it overwrites), so a transfer has STAGES - 1 steps of matrix time to land — one
// step (12 MFMAs = 384 cycles) is shorter than an L2 round trip.  hipcc waits for EVERY LDS-DMA in flight in front of an
// LDS read it can see, so the fragment reads are inline asm under a hand-counted vmcnt.
template <int ACT, int STAGES>
__global__ __launch_bounds__(256) void gemm_bf16x3_kernel(const uint16_t* __restrict__ Ap, size_t a_plane,
                                                         const uint16_t* __restrict__ Wp, size_t w_plane,
                                                         const float* __restrict__ bias, float* __restrict__ Y,
                                                         uint16_t* __restrict__ Yp, size_t y_plane, int M, int N, int K) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];  // STAGES x 24 KiB
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // XCD-aware tile order (see gemm_nt_kernel): consecutive tiles of one strip of A on one XCD
    const int per_xcd = gridDim.x >> 3;
    const int tile = (int)(blockIdx.x & 7) * per_xcd + (int)(blockIdx.x >> 3);
    const int tiles_n = N / G3T;
    if (tile >= tiles_n * ((M + G3T - 1) / G3T)) return;
    const int m0 = (tile / tiles_n) * G3T, n0 = (tile % tiles_n) * G3T;
    const int wm = (wave >> 1) * 32, wn = (wave & 1) * 32;

    // DMA: this lane fills slot s = wave * 64 + lane of every plane tile: row s >> 2, physical chunk s & 3
    const int s_row = (wave * 64 + lane) >> 2, s_cp = lane & 3;
    const int s_c = s_cp ^ ((s_row >> 2) & 3);  // logical 16-B chunk (8 values of k) it has to fetch
    const int a_row = m0 + s_row < M ? m0 + s_row : M - 1;
    // K-blocked planes (plane_index): the 64 rows x 32 k of a step are 4 KiB contiguous; step k0 starts k0 * rows_alloc on
    const size_t a_rows = a_plane / K, w_rows = w_plane / K, y_rows = Yp ? y_plane / N : 0;
    const uint16_t* ga = Ap + (size_t)a_row * 32 + s_c * 8;
    const uint16_t* gw = Wp + (size_t)(n0 + s_row) * 32 + s_c * 8;
    auto dma = [&](int k0, int stage) __attribute__((always_inline)) {
        unsigned char* base = lds + stage * G3_BUF + wave * 1024;
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(ga + p * a_plane + (size_t)k0 * a_rows),
                                             (__attribute__((address_space(3))) void*)(base + p * G3_PLANE), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gw + p * w_plane + (size_t)k0 * w_rows),
                                             (__attribute__((address_space(3))) void*)(base + (3 + p) * G3_PLANE), 16, 0, 0);
        }
    };
    // operand reads: lane (r = lane & 31, kh = lane >> 5) wants chunk 2 j + kh of row wm + r (A) / wn + r (W)
    const int ra = wm + (lane & 31), rb = wn + (lane & 31), kh = lane >> 5;
    const uint32_t lds0 = (uint32_t)(size_t)(__attribute__((address_space(3))) unsigned char*)lds;
    uint32_t a_ad[2], b_ad[2];  // LDS byte offsets (stage 0) of this lane's chunk in k16 half j, plane 0
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        a_ad[j] = lds0 + (uint32_t)(ra * 64 + (((2 * j + kh) ^ ((ra >> 2) & 3)) << 4));
        b_ad[j] = lds0 + (uint32_t)(3 * G3_PLANE + rb * 64 + (((2 * j + kh) ^ ((rb >> 2) & 3)) << 4));
    }

    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;

    const int n_steps = K / G3K;
#pragma unroll
    for (int st = 0; st < STAGES - 1; ++st)
        if (st < n_steps) dma(st * G3K, st);
    int stage = 0;
    for (int i = 0; i < n_steps; ++i) {
        // this wave's share of step i has landed: at most the DMAs of the (up to STAGES - 2) younger steps are in flight
        const int younger = n_steps - 1 - i < STAGES - 2 ? n_steps - 1 - i : STAGES - 2;
        if (younger >= 2) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
        else if (younger == 1) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_barrier" ::: "memory");  // ... everyone's; and everyone has left step i - 1
        if (i + STAGES - 1 < n_steps) {
            int ws = stage + STAGES - 1;
            if (ws >= STAGES) ws -= STAGES;
            dma((i + STAGES - 1) * G3K, ws);  // into the image of step i - 1
        }
        const uint32_t so = (uint32_t)(stage * G3_BUF);
        u32x4 fa[2][3], fb[2][3];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int p = 0; p < 3; ++p) {
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fa[j][p]) : "v"(a_ad[j] + so), "n"(p * G3_PLANE));
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fb[j][p]) : "v"(b_ad[j] + so), "n"(p * G3_PLANE));
            }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            // (reads are issued in j order, 6 per k16 half: the first half is complete when 6 are still outstanding)
            // (the fragments are operands of the wait: the MFMAs that read them cannot be scheduled in front of it)
            if (j == 0)
                asm volatile("s_waitcnt lgkmcnt(6)"
                             : "+v"(fa[0][0]), "+v"(fa[0][1]), "+v"(fa[0][2]), "+v"(fb[0][0]), "+v"(fb[0][1]), "+v"(fb[0][2]));
            else
                asm volatile("s_waitcnt lgkmcnt(0)"
                             : "+v"(fa[1][0]), "+v"(fa[1][1]), "+v"(fa[1][2]), "+v"(fb[1][0]), "+v"(fb[1][1]), "+v"(fb[1][2]));
            auto mm = [&](int pa, int pb) __attribute__((always_inline)) {
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa[j][pa]),
                                                              __builtin_bit_cast(bf16x8, fb[j][pb]), acc, 0, 0, 0);
            };
            mm(2, 0);  // smallest classes first
            mm(0, 2);
            mm(1, 1);
            mm(1, 0);
            mm(0, 1);
            mm(0, 0);
        }
        if (++stage == STAGES) stage = 0;
    }
    // C/D map of the 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
    asm volatile("s_barrier" ::: "memory");  // every wave has left the K loop: the images are free (plane staging below)
    const int n = n0 + wn + (lane & 31);
    const float bvv = bias[n];
    float vv[16];
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
        const int row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
        const int m = m0 + wm + row;
        vv[reg] = act_apply3(acc[reg] + bvv, ACT);
        if (Y && m < M) Y[(size_t)m * N + n] = vv[reg];
    }
    if (Yp) {
        // plane output through the wave's share of the idle LDS (see gemm_bf16x3_big_kernel): the 32 x 32 tile in f32, 128 B per
        // row; 16-B slot q of row r sits at (q + ((r >> 2) & 1)) & 7 (a ds_read_b128 lane group then covers all 16 slots of the
        // 256-B bank row); read back as 8 consecutive floats per lane: 32 rows x 4 chunks = 2 per lane
        float* st = reinterpret_cast<float*>(lds + wave * (32 * 128));
        const int rot = 4 * (lane >> 5);  // ((row >> 2) & 1) in dwords: row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const int row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
            st[row * 32 + (((lane & 31) + rot) & 31)] = vv[reg];
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int ch = it * 64 + lane, row = ch >> 2, c = ch & 3;
            const int m = m0 + wm + row;
            const int r1 = (row >> 2) & 1;
            const float4 f0 = *reinterpret_cast<const float4*>(st + row * 32 + (((2 * c + r1) & 7) << 2));
            const float4 f1 = *reinterpret_cast<const float4*>(st + row * 32 + (((2 * c + 1 + r1) & 7) << 2));
            const float f[8] = {f0.x, f0.y, f0.z, f0.w, f1.x, f1.y, f1.z, f1.w};
            uint32_t w[3][4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                split3_bf16_pair(f[2 * e], f[2 * e + 1], w[0][e], w[1][e], w[2][e]);
            }
            if (m < M) {  // (the wave's 32 columns are one k-block of the next layer: 16 rows x 64 B contiguous per store)
                uint16_t* dst = Yp + plane_index(m, n0 + wn + c * 8, y_rows);
#pragma unroll
                for (int p = 0; p < 3; ++p)
                    *reinterpret_cast<u32x4*>(dst + p * y_plane) = u32x4{w[p][0], w[p][1], w[p][2], w[p][3]};
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// 128 x 128 block tile, 8 waves (two per SIMD), wave tile 32 x 64: an A fragment feeds two MFMA columns, so a k16 step is 9
// fragment reads for 12 MFMAs instead of 6 for 6 — the 64x64 kernel above is bound by the LDS read port (4 SIMDs x 6 KiB per
// 192 cycles = the port's 128 B/clk), this one needs 73 % of it.  K-step 32, double buffer of 48-KiB images (one workgroup
// per CU; the second wave of a SIMD covers the first one's waits).  Large M only (M >= 2048: below that there are not
// enough 128 x 128 tiles for the chip).
// ------------------------------------------------------------------------------------------------------------------
constexpr int G3B = 128;
constexpr int G3B_PLANE = G3B * G3K * 2;  // 8192 B
constexpr int G3B_BUF = 6 * G3B_PLANE;    // 48 KiB
constexpr int G3B_STAGE = 8 * 32 * 256;                  // epilogue staging: 8 waves x 32 rows x 64 floats
constexpr int G3B_STAGES = 3;
constexpr int G3B_LDS = G3B_STAGES * G3B_BUF > G3B_STAGE ? G3B_STAGES * G3B_BUF : G3B_STAGE;  // 144 KiB

template <int ACT, int STAGES, int PP>
__global__ __launch_bounds__(512) void gemm_bf16x3_big_kernel(const uint16_t* __restrict__ Ap, size_t a_plane,
                                                             const uint16_t* __restrict__ Wp, size_t w_plane,
                                                             const float* __restrict__ bias, float* __restrict__ Y,
                                                             uint16_t* __restrict__ Yp, size_t y_plane, int M, int N, int K) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];  // STAGES x 48 KiB
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // Tiles in XCD-aware order (see gemm_nt_kernel): the blocks of one XCD (blockIdx & 7) walk one eighth of the tile list,
    // consecutive tiles — one strip of A — at the same time.  With fewer blocks than tiles (launch_g3_big: one per CU) a block
    // loops: no block launch between two tiles, and a tile's plane stores drain under the next tile's first DMA.
    const int tiles_n = N / G3B;
    const int n_tiles = tiles_n * ((M + G3B - 1) / G3B);
    const int per_xcd = (n_tiles + 7) >> 3, slots = gridDim.x >> 3;
    const int wm = (wave >> 1) * 32, wn = (wave & 1) * 64;  // 4 x 2 waves
    for (int local = (int)(blockIdx.x >> 3); local < per_xcd; local += slots) {
        const int tile = (int)(blockIdx.x & 7) * per_xcd + local;
        if (tile >= n_tiles) break;
        const int m0 = (tile / tiles_n) * G3B, n0 = (tile % tiles_n) * G3B;

        // DMA: 512 lanes fill the 512 16-B slots of every 128-row plane tile: slot s = wave * 64 + lane: row s >> 2, chunk s & 3
        const int s_row = (wave * 64 + lane) >> 2, s_cp = lane & 3;
        const int s_c = s_cp ^ ((s_row >> 2) & 3);
        const int a_row = m0 + s_row < M ? m0 + s_row : M - 1;
        // K-blocked planes (plane_index): the 128 rows x 32 k of a step are 8 KiB contiguous — a wave's piece is 1 KiB of it
        const size_t a_rows = a_plane / K, w_rows = w_plane / K, y_rows = Yp ? y_plane / N : 0;
        const uint16_t* ga = Ap + (size_t)a_row * 32 + s_c * 8;
        const uint16_t* gw = Wp + (size_t)(n0 + s_row) * 32 + s_c * 8;
        auto dma = [&](int k0, int stage) __attribute__((always_inline)) {
            unsigned char* base = lds + stage * G3B_BUF + wave * 1024;
#pragma unroll
            for (int p = 0; p < 3; ++p) {
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(ga + p * a_plane + (size_t)k0 * a_rows),
                                                 (__attribute__((address_space(3))) void*)(base + p * G3B_PLANE), 16, 0, 0);
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gw + p * w_plane + (size_t)k0 * w_rows),
                                                 (__attribute__((address_space(3))) void*)(base + (3 + p) * G3B_PLANE), 16, 0, 0);
            }
        };
        const int ra = wm + (lane & 31), rb0 = wn + (lane & 31), rb1 = rb0 + 32, kh = lane >> 5;
        const uint32_t lds0 = (uint32_t)(size_t)(__attribute__((address_space(3))) unsigned char*)lds;
        uint32_t a_ad[2], b_ad[2][2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            a_ad[j] = lds0 + (uint32_t)(ra * 64 + (((2 * j + kh) ^ ((ra >> 2) & 3)) << 4));
            b_ad[0][j] = lds0 + (uint32_t)(3 * G3B_PLANE + rb0 * 64 + (((2 * j + kh) ^ ((rb0 >> 2) & 3)) << 4));
            b_ad[1][j] = lds0 + (uint32_t)(3 * G3B_PLANE + rb1 * 64 + (((2 * j + kh) ^ ((rb1 >> 2) & 3)) << 4));
        }
        f32x16 acc0, acc1;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc0[i] = acc1[i] = 0.f;

        // ring of STAGES images: a step is 24 MFMAs per wave (0.3 us of matrix time, two waves per SIMD) and an L2 round trip
        // under load is ~1 us: with one image in flight the K loop runs at the DMA's latency (measured: 190 us for
        // 32768 x 1152 x 384, whose MFMAs, LDS reads and L2 traffic each need ~70 us); two in flight cover it
        const int n_steps = K / G3K;
        if constexpr (PP != 0) {
            // Ping-pong: the two waves of a SIMD (w and w + 4) run half a step apart.  A wave's step is a memory phase M(i) — the
            // 18 fragment reads of step i, its six DMA pieces of step i + STAGES - 1, the wait for its share of step i + 1 — and
            // a matrix phase C(i) — the 24 MFMAs — with a block barrier after each; waves 4..7 start one barrier late, so that
            // M(i) of one group runs under C(i) (or C(i - 1)) of the other: while a SIMD's matrix pipe works for one wave, the
            // other wave's loads are being issued.  (In lockstep all eight waves issue DMA at the same time, ~100-180 cycles a
            // piece, and nobody feeds the matrix pipe.  Measured on 32768 rows, lockstep / ping-pong: K = 1536: 229 / 214 us,
            // the K = 384 shapes 1-2 %; with the K loop's parts removed (K = 1536): no DMA 154 us, no fragment reads 178 us,
            // no MFMAs 125 us, the MFMAs alone would take 97 us.)
            //   image i is read by group 0 in its M(i) and by group 1 one phase later; it is overwritten by the DMA of
            //   M(i + 1), which both groups start after the barrier that ends group 1's M(i);
            //   image i + 1 is complete when every wave has waited for its share: at the end of its M(i), in front of the
            //   barrier that group 0's M(i + 1) follows.
            const int grp = wave >> 2;
#pragma unroll
            for (int st = 0; st < STAGES - 1; ++st)
                if (st < n_steps) dma(st * G3K, st);
            if (n_steps - 1 >= STAGES - 2 && STAGES >= 3) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            asm volatile("s_barrier" ::: "memory");  // image 0 is complete
            if (grp == 1) asm volatile("s_barrier" ::: "memory");  // the stagger
            int stage = 0;
            for (int i = 0; i < n_steps; ++i) {
                const uint32_t so = (uint32_t)(stage * G3B_BUF);
                u32x4 fa[2][3], fb0[2][3], fb1[2][3];
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int p = 0; p < 3; ++p) {
                        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fa[j][p]) : "v"(a_ad[j] + so), "n"(p * G3B_PLANE));
                        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fb0[j][p]) : "v"(b_ad[0][j] + so), "n"(p * G3B_PLANE));
                        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fb1[j][p]) : "v"(b_ad[1][j] + so), "n"(p * G3B_PLANE));
                    }
                if (i + STAGES - 1 < n_steps) {
                    int ws = stage + STAGES - 1;
                    if (ws >= STAGES) ws -= STAGES;
                    dma((i + STAGES - 1) * G3K, ws);  // into the image of step i - 1
                }
                // my share of step i + 1: at most the younger steps' pieces may still be in flight
                if (i + 1 < n_steps) {
                    const int younger = n_steps - 2 - i < STAGES - 2 ? n_steps - 2 - i : STAGES - 2;
                    if (younger >= 1) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
                    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
                asm volatile("s_waitcnt lgkmcnt(0)"
                             : "+v"(fa[0][0]), "+v"(fa[0][1]), "+v"(fa[0][2]), "+v"(fb0[0][0]), "+v"(fb0[0][1]), "+v"(fb0[0][2]),
                               "+v"(fb1[0][0]), "+v"(fb1[0][1]), "+v"(fb1[0][2]));
                asm volatile("" : "+v"(fa[1][0]), "+v"(fa[1][1]), "+v"(fa[1][2]), "+v"(fb0[1][0]), "+v"(fb0[1][1]), "+v"(fb0[1][2]),
                             "+v"(fb1[1][0]), "+v"(fb1[1][1]), "+v"(fb1[1][2]));
                asm volatile("s_barrier" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    auto mm = [&](int pa, int pb) __attribute__((always_inline)) {
                        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa[j][pa]),
                                                                       __builtin_bit_cast(bf16x8, fb0[j][pb]), acc0, 0, 0, 0);
                        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa[j][pa]),
                                                                       __builtin_bit_cast(bf16x8, fb1[j][pb]), acc1, 0, 0, 0);
                    };
                    mm(2, 0);
                    mm(0, 2);
                    mm(1, 1);
                    mm(1, 0);
                    mm(0, 1);
                    mm(0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
                asm volatile("s_barrier" ::: "memory");
                if (++stage == STAGES) stage = 0;
            }
            if (grp == 0) asm volatile("s_barrier" ::: "memory");  // (as many barriers as group 1)
        } else {
#pragma unroll
        for (int st = 0; st < STAGES - 1; ++st)
            if (st < n_steps) dma(st * G3K, st);
        int stage = 0;
        for (int i = 0; i < n_steps; ++i) {
            const int younger = n_steps - 1 - i < STAGES - 2 ? n_steps - 1 - i : STAGES - 2;
            if (younger >= 1) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            asm volatile("s_barrier" ::: "memory");
            if (i + STAGES - 1 < n_steps) {
                int ws = stage + STAGES - 1;
                if (ws >= STAGES) ws -= STAGES;
                dma((i + STAGES - 1) * G3K, ws);
            }
            const uint32_t so = (uint32_t)(stage * G3B_BUF);
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                u32x4 fa[3], fb0[3], fb1[3];
#pragma unroll
                for (int p = 0; p < 3; ++p) {
                    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fa[p]) : "v"(a_ad[j] + so), "n"(p * G3B_PLANE));
                    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fb0[p]) : "v"(b_ad[0][j] + so), "n"(p * G3B_PLANE));
                    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fb1[p]) : "v"(b_ad[1][j] + so), "n"(p * G3B_PLANE));
                }
                asm volatile("s_waitcnt lgkmcnt(0)"
                             : "+v"(fa[0]), "+v"(fa[1]), "+v"(fa[2]), "+v"(fb0[0]), "+v"(fb0[1]), "+v"(fb0[2]), "+v"(fb1[0]),
                               "+v"(fb1[1]), "+v"(fb1[2]));
                auto mm = [&](int pa, int pb) __attribute__((always_inline)) {
                    acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa[pa]), __builtin_bit_cast(bf16x8, fb0[pb]),
                                                                   acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa[pa]), __builtin_bit_cast(bf16x8, fb1[pb]),
                                                                   acc1, 0, 0, 0);
                };
                mm(2, 0);
                mm(0, 2);
                mm(1, 1);
                mm(1, 0);
                mm(0, 1);
                mm(0, 0);
            }
            if (++stage == STAGES) stage = 0;
        }
        }
        // ---- epilogue.  f32 output: straight from the accumulators (a lane holds one column: 32 lanes = 128 contiguous bytes
        // per row).  Plane output: a lane's values are 2 bytes each — 2-byte stores are read-modify-writes of 32-B sectors in
        // L2 (measured: +116 us on the 262-us FFN1 of 32 k tokens) — so the wave's 32 x 64 tile goes through its share of the
        // (now idle) LDS in f32, is read back by row, split there, and stored as 16-B chunks (full sectors).  (Splitting first
        // and staging bf16 triples costs 96 two-byte LDS stores per lane instead of 32 dword stores: +5 % on the whole kernel.)
        asm volatile("s_barrier" ::: "memory");  // every wave has left the K loop: the images are free
        float vv[2][16];
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const int n = n0 + wn + 32 * half + (lane & 31);
            const float bvv = bias[n];
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
                const int m = m0 + wm + row;
                const float v = act_apply3((half ? acc1[reg] : acc0[reg]) + bvv, ACT);
                vv[half][reg] = v;
                if (Y && m < M) Y[(size_t)m * N + n] = v;
            }
        }
        if (Yp) {
            // staging in f32: the wave's 32 x 64 tile as [row][64 floats] (256 B per row, 8 KiB per wave), one 4-byte LDS
            // store per value (a 32-lane group writes 32 consecutive banks); read back as 8 consecutive floats of a row per
            // lane (two ds_read_b128), split there, and stored as one 16-B chunk per plane.  The tile's 64 columns are two
            // k-blocks of the next layer's operand: lane q (+ 64 per round) takes chunk q & 3 of row (q >> 2) & 31 in k-block
            // q >> 7, so that a store instruction writes 16 rows x 64 B = 1 KiB contiguous.  16-B slot s of row r sits at
            // (s + (r & 1) + 8 ((r >> 1) & 1)) & 15: a b128 lane group (rows {0, 3, 5, 6} or {1, 2, 4, 7} + 8 n, four even or
            // four odd slots each) then covers all 16 slots of the 256-B bank row.
            float* st = reinterpret_cast<float*>(lds + wave * (32 * 256));
#pragma unroll
            for (int half = 0; half < 2; ++half)
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) {
                    const int row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
                    const int rot = 4 * ((reg & 1) + 8 * ((reg >> 1) & 1));  // slot rotation of the row, in dwords
                    st[row * 64 + ((32 * half + (lane & 31) + rot) & 63)] = vv[half][reg];
                }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int q = it * 64 + lane, c4 = q & 3, row = (q >> 2) & 31, kb = q >> 7;
                const int m = m0 + wm + row;
                const int rot = (row & 1) + 8 * ((row >> 1) & 1);
                const int s0 = 2 * (kb * 4 + c4);
                const float4 f0 = *reinterpret_cast<const float4*>(st + row * 64 + (((s0 + rot) & 15) << 2));
                const float4 f1 = *reinterpret_cast<const float4*>(st + row * 64 + (((s0 + 1 + rot) & 15) << 2));
                const float f[8] = {f0.x, f0.y, f0.z, f0.w, f1.x, f1.y, f1.z, f1.w};
                uint32_t w[3][4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    split3_bf16_pair(f[2 * e], f[2 * e + 1], w[0][e], w[1][e], w[2][e]);
                }
                if (m < M) {
                    uint16_t* dst = Yp + plane_index(m, n0 + wn + kb * 32 + c4 * 8, y_rows);
#pragma unroll
                    for (int p = 0; p < 3; ++p)
                        *reinterpret_cast<u32x4*>(dst + p * y_plane) = u32x4{w[p][0], w[p][1], w[p][2], w[p][3]};
                }
            }
        }
        // (the staging reads are done — their data is in registers — before the next tile's DMA lands on the same LDS)
        if (local + slots < per_xcd) asm volatile("s_barrier" ::: "memory");
    }
}


template <int PP>
static void launch_g3_big_v(const uint16_t* Ap, size_t a_plane, const uint16_t* Wp, size_t w_plane, const float* bias, float* Y,
                            uint16_t* Yp, size_t y_plane, int M, int N, int K, int act, hipStream_t s, const Gemm3Opts& o) {
    static OncePerDevice attr;  // (the attribute belongs to the current device's copy of the kernel: kernels.hpp)
    once_per_device(attr, [] {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_bf16x3_big_kernel<0, G3B_STAGES, PP>), hipFuncAttributeMaxDynamicSharedMemorySize, G3B_LDS);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_bf16x3_big_kernel<1, G3B_STAGES, PP>), hipFuncAttributeMaxDynamicSharedMemorySize, G3B_LDS);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_bf16x3_big_kernel<2, G3B_STAGES, PP>), hipFuncAttributeMaxDynamicSharedMemorySize, G3B_LDS);
    });
    const int n_tiles = (N / G3B) * ((M + G3B - 1) / G3B);
    int blocks = (n_tiles + 7) / 8 * 8;
    if (o.persistent > 0 && blocks > o.persistent) blocks = o.persistent;  // one block per CU walks the tiles
    dim3 grid(blocks), block(512);
    const size_t lds = G3B_LDS;
    if (act == 1) hipLaunchKernelGGL((gemm_bf16x3_big_kernel<1, G3B_STAGES, PP>), grid, block, lds, s, Ap, a_plane, Wp, w_plane, bias, Y, Yp, y_plane, M, N, K);
    else if (act == 2) hipLaunchKernelGGL((gemm_bf16x3_big_kernel<2, G3B_STAGES, PP>), grid, block, lds, s, Ap, a_plane, Wp, w_plane, bias, Y, Yp, y_plane, M, N, K);
    else hipLaunchKernelGGL((gemm_bf16x3_big_kernel<0, G3B_STAGES, PP>), grid, block, lds, s, Ap, a_plane, Wp, w_plane, bias, Y, Yp, y_plane, M, N, K);
}

static void launch_g3_big(const uint16_t* Ap, size_t a_plane, const uint16_t* Wp, size_t w_plane, const float* bias, float* Y,
                          uint16_t* Yp, size_t y_plane, int M, int N, int K, int act, hipStream_t s, const Gemm3Opts& o) {
    if (o.pingpong) return launch_g3_big_v<1>(Ap, a_plane, Wp, w_plane, bias, Y, Yp, y_plane, M, N, K, act, s, o);
    return launch_g3_big_v<0>(Ap, a_plane, Wp, w_plane, bias, Y, Yp, y_plane, M, N, K, act, s, o);
}

// The 128 x 128 kernel is used when the GEMM has at least this many of its tiles: below that the chip is better filled by four
// times as many 64 x 64 tiles (4708 rows, 64 / 128 tiles: N = 384 — 111 tiles —: 15.8 / 21.4 us, K = 1536: 44 / 68 us;
// N = 1536 — 444 tiles —: 51 / 46 us; 32768 rows: 857 / 761 us for the four shapes of a layer — tools/gemm3_probe.py).
// Round 3, whole forwards (tools/embed_tile_sweep.py, profiles/r03/embed_tile_sweep.log; results are bit-identical whatever the
// form): the round-2 threshold of 512 tiles (two per CU) kept every GEMM of a 4708-token batch and the N = 384 GEMMs of a 9174-token
// batch on the small form; from ~190 tiles (three quarters of the CUs busy with one persistent workgroup each) the big form
// wins: 256 queries / 4708 tokens 0.965 -> 0.90 ms, 512 queries / 9174 tokens 1.71 -> 1.50 ms, 64 pages 1.44 -> 1.40 ms;
// at 111 tiles it loses (1.00 ms).  0 = always.  (Gemm3Opts::big_min_tiles, embed_kernels.hpp)

template <int STAGES>
static void launch_g3(const uint16_t* Ap, size_t a_plane, const uint16_t* Wp, size_t w_plane, const float* bias, float* Y,
                      uint16_t* Yp, size_t y_plane, int M, int N, int K, int act, hipStream_t s) {
    static OncePerDevice attr;
    once_per_device(attr, [] {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_bf16x3_kernel<0, STAGES>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, STAGES * G3_BUF);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_bf16x3_kernel<1, STAGES>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, STAGES * G3_BUF);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_bf16x3_kernel<2, STAGES>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, STAGES * G3_BUF);
    });
    const int n_tiles = (N / G3T) * ((M + G3T - 1) / G3T);
    dim3 grid((n_tiles + 7) / 8 * 8), block(256);
    const size_t lds = (size_t)STAGES * G3_BUF;
    if (act == 1)
        hipLaunchKernelGGL((gemm_bf16x3_kernel<1, STAGES>), grid, block, lds, s, Ap, a_plane, Wp, w_plane, bias, Y, Yp, y_plane, M, N, K);
    else if (act == 2)
        hipLaunchKernelGGL((gemm_bf16x3_kernel<2, STAGES>), grid, block, lds, s, Ap, a_plane, Wp, w_plane, bias, Y, Yp, y_plane, M, N, K);
    else
        hipLaunchKernelGGL((gemm_bf16x3_kernel<0, STAGES>), grid, block, lds, s, Ap, a_plane, Wp, w_plane, bias, Y, Yp, y_plane, M, N, K);
}

void launch_gemm_bf16x3(const uint16_t* Ap, size_t a_plane, const uint16_t* Wp, size_t w_plane, const float* bias, float* Y,
                        uint16_t* Yp, size_t y_plane, int M, int N, int K, int act, hipStream_t s, const Gemm3Opts& o) {
    if (M <= 0) return;
    if (N % G3B == 0 && (long long)((M + G3B - 1) / G3B) * (N / G3B) >= o.big_min_tiles) return launch_g3_big(Ap, a_plane, Wp, w_plane, bias, Y, Yp, y_plane, M, N, K, act, s, o);
    if (o.stages == 2) launch_g3<2>(Ap, a_plane, Wp, w_plane, bias, Y, Yp, y_plane, M, N, K, act, s);
    else if (o.stages == 4) launch_g3<4>(Ap, a_plane, Wp, w_plane, bias, Y, Yp, y_plane, M, N, K, act, s);
    else launch_g3<3>(Ap, a_plane, Wp, w_plane, bias, Y, Yp, y_plane, M, N, K, act, s);
}

}  // namespace dawn
