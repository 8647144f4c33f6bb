// scan_mfma.hip — batched query x index contraction on the matrix cores (gfx950), fused with top-k.
//
// For B > 8 queries the streaming filter (scan_kernels.hip) would re-read the index B/4 times; here one
// pass over the rows serves a tile of QT = 32*G queries (G = 1..3) with v_mfma_f32_32x32x2_f32 — f32 in,
// f32 accumulate: exact products, k-ordered FMA chain — so the filter error bound stays at the 1e-5 level
// and the same merge/exact-rescore/certificate tail (scan_kernels.hip) applies unchanged.
//
// Block = 4 waves (one per SIMD; the kernel spends its registers, not occupancy).  The query tile sits in
// LDS ([QT][388] f32, 16-B padded rows: conflict-free ds_read_b128 for the A operand).  Each wave owns
// 32-row blocks of the index (grid-strided): the B operand comes straight from global memory in fragment
// shape (lane = (row, 16-B chunk parity)), three 64-dim slabs in flight.  D[i][j] = score(query i, row j).
// The k index inside one MFMA is permuted (dims 8t+4h+m for lane-half h, component m) identically for both
// operands.
//
// Top-k: every wave keeps, for each of its QT queries, a sorted 64-entry list with ONE ENTRY PER LANE in
// registers (2 VGPRs per query).  A 32x32 score tile is tested against per-query thresholds with 16*G
// compares; insertion (ballot + shift) happens only on a hit.  Thresholds are shared chip-wide through
// gtau[q] (atomicMax of any wave's full-list 64th score — a valid lower bound of the final 64th score), so
// the number of insertions is ~64*ln(N/64) per query for the whole grid, not per wave.
#include "kernels.hpp"

namespace dawn {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define NEG_INF (-__builtin_inff())
constexpr uint32_t NO_POS = 0xFFFFFFFFu;
constexpr int QROW = 388;  // floats per staged query row (1552 B)

__device__ __forceinline__ float read_lane(float v, int l) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l));
}

__device__ __forceinline__ bool better(float s, uint32_t p, float s2, uint32_t p2) {
    return s > s2 || (s == s2 && p < p2);
}

__device__ __forceinline__ void wave_insert(float& ls, uint32_t& lp, float s, uint32_t p, int lane) {
    const bool ahead = better(ls, lp, s, p);
    const int pos = __popcll(__ballot(ahead));
    const float ps = __shfl_up(ls, 1);
    const uint32_t pp = __shfl_up(lp, 1);
    if (lane > pos) {
        ls = ps;
        lp = pp;
    } else if (lane == pos) {
        ls = s;
        lp = p;
    }
}

// monotone float -> int map so that atomicMax(int) orders floats
__device__ __forceinline__ int f2ord(float f) {
    const int i = __builtin_bit_cast(int, f);
    return i ^ ((i >> 31) & 0x7FFFFFFF);
}
__device__ __forceinline__ float ord2f(int i) { return __builtin_bit_cast(float, i ^ ((i >> 31) & 0x7FFFFFFF)); }

template <int G>
__global__ __launch_bounds__(256, 1) void scan_mfma_kernel(const f32x4* __restrict__ x, uint32_t n_rows,
                                                          const float* __restrict__ q, int n_q,
                                                          int* __restrict__ gtau, float* __restrict__ out_s,
                                                          uint32_t* __restrict__ out_p, uint32_t n_lists) {
    extern __shared__ __attribute__((aligned(16))) float Qs[];  // [32*G][QROW]
    constexpr int QT = 32 * G;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5, j = lane & 31;
    const int q0 = blockIdx.y * QT;  // first query of this tile

    // stage the query tile (zero rows past n_q)
    for (int i = tid; i < QT * 96; i += 256) {
        const int r = i / 96, c = i % 96;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (q0 + r < n_q) v = *reinterpret_cast<const f32x4*>(q + (size_t)(q0 + r) * EM + c * 4);
        *reinterpret_cast<f32x4*>(Qs + r * QROW + c * 4) = v;
    }
    __syncthreads();

    float lsA[G][16], lsB[G][16], tauv[G][16], gtv[G][16];
    uint32_t lpA[G][16], lpB[G][16];
#pragma unroll
    for (int g = 0; g < G; ++g)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            lsA[g][r] = NEG_INF;
            lsB[g][r] = NEG_INF;
            lpA[g][r] = NO_POS;
            lpB[g][r] = NO_POS;
            tauv[g][r] = NEG_INF;
            gtv[g][r] = NEG_INF;
        }

    const uint32_t n_rb = (n_rows + 31u) >> 5;
    const uint32_t W = gridDim.x * 4u;
    const uint32_t gw = blockIdx.x * 4u + wave;

    f32x4 b0[8], b1[8], b2[8];
    auto load_slab = [&](f32x4(&buf)[8], uint32_t rb, int s) {
        const f32x4* p = x + ((size_t)rb * 32 + j) * ROW_F4 + s * 16 + h;
#pragma unroll
        for (int tt = 0; tt < 8; ++tt) buf[tt] = p[2 * tt];
    };
    const float* qbase0 = Qs + j * QROW + 4 * h;
    const float* qbase = qbase0;
    f32x16 acc[G];
    auto compute_slab = [&](const f32x4(&buf)[8], int s) {
#pragma unroll
        for (int tt = 0; tt < 8; ++tt) {
            const int t = s * 8 + tt;
#pragma unroll
            for (int g = 0; g < G; ++g) {
                const f32x4 qv = *reinterpret_cast<const f32x4*>(qbase + g * 32 * QROW + 8 * t);
                acc[g] = __builtin_amdgcn_mfma_f32_32x32x2f32(qv.x, buf[tt].x, acc[g], 0, 0, 0);
                acc[g] = __builtin_amdgcn_mfma_f32_32x32x2f32(qv.y, buf[tt].y, acc[g], 0, 0, 0);
                acc[g] = __builtin_amdgcn_mfma_f32_32x32x2f32(qv.z, buf[tt].z, acc[g], 0, 0, 0);
                acc[g] = __builtin_amdgcn_mfma_f32_32x32x2f32(qv.w, buf[tt].w, acc[g], 0, 0, 0);
            }
        }
    };

    if (gw < n_rb) {
        load_slab(b0, gw, 0);
        load_slab(b1, gw, 1);
    }
    for (uint32_t rb = gw; rb < n_rb; rb += W) {
        const uint32_t nxt = (rb + W < n_rb) ? rb + W : rb;  // clamp: the last prefetch re-reads this block
        {   // The query fragments are loop-invariant per lane; left alone, LICM hoists all 48*G ds_read_b128
            // results (192*G VGPRs) out of this loop and spills them.  Keep them in LDS: re-derive the base
            // through an opaque zero every iteration.
            int zero = 0;
            asm volatile("" : "+v"(zero));
            qbase = qbase0 + zero;
        }
#pragma unroll
        for (int g = 0; g < G; ++g)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[g][r] = 0.f;
        load_slab(b2, rb, 2);
        compute_slab(b0, 0);
        load_slab(b0, rb, 3);
        compute_slab(b1, 1);
        load_slab(b1, rb, 4);
        compute_slab(b2, 2);
        load_slab(b2, rb, 5);
        compute_slab(b0, 3);
        load_slab(b0, nxt, 0);
        compute_slab(b1, 4);
        load_slab(b1, nxt, 1);
        compute_slab(b2, 5);

        // ---- epilogue: threshold test of the 32 x QT score tile -----------------------------------
        const uint32_t row = rb * 32u + (uint32_t)j;
        const bool row_ok = row < n_rows;
#pragma unroll
        for (int g = 0; g < G; ++g) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                // C/D map: this lane holds D[query = g*32 + (r&3) + 8*(r>>2) + 4*h][row j]
                const int qa = q0 + g * 32 + (r & 3) + 8 * (r >> 2);  // lanes 0..31; lanes 32..63: qa + 4
                float sc = acc[g][r];
                sc = (row_ok && sc == sc) ? sc : NEG_INF;
                // chip-wide threshold fetched at the end of the previous row block (staleness only costs insertions)
                const float gt = gtv[g][r];
                float tv = tauv[g][r];
                tv = gt > tv ? gt : tv;
                unsigned long long hits = __ballot(sc > tv);
                if (hits) {
                    float tA = read_lane(tv, 0), tB = read_lane(tv, 32);
                    while (hits) {
                        const int src = __builtin_ctzll(hits);
                        hits &= hits - 1;
                        const float s = read_lane(sc, src);
                        const uint32_t prow = rb * 32u + (uint32_t)(src & 31);
                        if (src < 32) {
                            if (s > tA) {
                                wave_insert(lsA[g][r], lpA[g][r], s, prow, lane);
                                const float nt = read_lane(lsA[g][r], 63);
                                if (nt > tA) {
                                    tA = nt;
                                    if (lane == 0) atomicMax(&gtau[qa], f2ord(nt));
                                }
                            }
                        } else {
                            if (s > tB) {
                                wave_insert(lsB[g][r], lpB[g][r], s, prow, lane);
                                const float nt = read_lane(lsB[g][r], 63);
                                if (nt > tB) {
                                    tB = nt;
                                    if (lane == 0) atomicMax(&gtau[qa + 4], f2ord(nt));
                                }
                            }
                        }
                    }
                    tv = h ? tB : tA;
                }
                tauv[g][r] = tv;
            }
        }
        // Fetch the shared thresholds for the NEXT row block now; the loads fly under the next 384*G MFMAs.
        // Agent-scope relaxed loads (sc1): a plain load would be served from this CU's L1 forever.
#pragma unroll
        for (int g = 0; g < G; ++g)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int qa = q0 + g * 32 + (r & 3) + 8 * (r >> 2);
                gtv[g][r] = ord2f(__hip_atomic_load(&gtau[qa + 4 * h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
            }
    }

    // ---- write the per-wave lists: [query][list = blockIdx.x*4 + wave][64] ---------------------------
#pragma unroll
    for (int g = 0; g < G; ++g)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int qa = q0 + g * 32 + (r & 3) + 8 * (r >> 2);
            if (qa < n_q) {
                const size_t o = ((size_t)qa * n_lists + gw) * LIST + lane;
                out_s[o] = lsA[g][r];
                out_p[o] = lpA[g][r];
            }
            if (qa + 4 < n_q) {
                const size_t o = ((size_t)(qa + 4) * n_lists + gw) * LIST + lane;
                out_s[o] = lsB[g][r];
                out_p[o] = lpB[g][r];
            }
        }
}

template <int G>
static void launch_g(const float* d_x, uint32_t n_rows, const float* d_q, int n_q, int tiles, int* gtau, float* cs,
                     uint32_t* cp, int blocks, hipStream_t stream) {
    const size_t lds = (size_t)32 * G * QROW * sizeof(float);
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(scan_mfma_kernel<G>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    hipLaunchKernelGGL((scan_mfma_kernel<G>), dim3(blocks, tiles), dim3(256), lds, stream,
                       reinterpret_cast<const f32x4*>(d_x), n_rows, d_q, n_q, gtau, cs, cp, (uint32_t)(blocks * 4));
}

// d_gtau: [B rounded up to 32*G tiles] ints, initialised to INT_MIN by the caller per search.
// Lists come out as [B][blocks*4][64].  All B queries are covered by ceil(B/96) tiles of G=3 ... or fewer
// queries per tile for small B (G = 1 for B <= 32, 2 for B <= 64).
void launch_scan_mfma(const float* d_x, uint32_t n_rows, const float* d_q, int B, int* d_gtau, float* cand_s,
                      uint32_t* cand_p, int blocks, hipStream_t stream, hipEvent_t ev0, hipEvent_t ev1) {
    if (ev0) (void)hipEventRecord(ev0, stream);
    if (B <= 32) launch_g<1>(d_x, n_rows, d_q, B, 1, d_gtau, cand_s, cand_p, blocks, stream);
    else if (B <= 64) launch_g<2>(d_x, n_rows, d_q, B, 1, d_gtau, cand_s, cand_p, blocks, stream);
    else launch_g<2>(d_x, n_rows, d_q, B, (B + 63) / 64, d_gtau, cand_s, cand_p, blocks, stream);
    if (ev1) (void)hipEventRecord(ev1, stream);
}

}  // namespace dawn
