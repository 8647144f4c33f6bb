// embed_kernels.hip — all-MiniLM-L6-v2 forward for gfx950, f32 end to end.
//
// Restates BertModel::forward (src/embedding/model.rs:565-570) + the pool/normalise tail of
// EmbeddingProvider::calculate_embedding (src/embedding/embedding_service.rs:124-136) on PACKED
// variable-length sequences: token t of sequence b lives at row seq_offsets[b] + t, there are no padding
// tokens, attention never crosses a sequence and the mean is over the sequence's own tokens — so every text
// gets exactly its batch-1 result (the reference only ever embeds one text per call, :161-163).
//
//   embed_ln_kernel       word[id] + type[0] + pos[t]  -> LayerNorm            (model.rs:266-281, 86-104)
//   gemm_nt_kernel<ACT>   Y = X·Wᵀ + b (+ tanh-GELU / ReLU), W stored [out,in] (model.rs:53-64, 28-37)
//                         v_mfma_f32_32x32x2_f32: exact f32, k-ordered FMA chain
//   attention_kernel      per (sequence, head): softmax(QKᵀ/√32)·V, no mask     (model.rs:325-347)
//   add_ln_kernel         LayerNorm(dense_out + residual)                       (model.rs:374-379, 458-463)
//   pool_norm_kernel      mean over the sequence's tokens, then x/√Σx²          (embedding_service.rs:126-136)
#include "embed_kernels.hpp"
#include "wave_topk.hpp"

namespace dawn {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int H = 384;   // hidden_size
constexpr int DH = 32;   // attention_head_size
constexpr int NH = 12;   // num_attention_heads

// xor-butterfly all-reduce (the summation order of the classic __shfl_xor loop) on DPP / permlane-swap lane
// exchanges (wave_topk.hpp): a ds_bpermute per step costs ~100 clk of dependent latency, 12 of them per LayerNorm row
__device__ __forceinline__ float wave_allreduce_sum(float v) {
    const int lane = threadIdx.x & 63;
    v += lane_xor_f32<32>(v, lane);
    v += lane_xor_f32<16>(v, lane);
    v += lane_xor_f32<8>(v, lane);
    v += lane_xor_f32<4>(v, lane);
    v += lane_xor_f32<2>(v, lane);
    v += lane_xor_f32<1>(v, lane);
    return v;
}
__device__ __forceinline__ float wave_allreduce_max(float v) {
    const int lane = threadIdx.x & 63;
    v = fmaxf(v, lane_xor_f32<32>(v, lane));
    v = fmaxf(v, lane_xor_f32<16>(v, lane));
    v = fmaxf(v, lane_xor_f32<8>(v, lane));
    v = fmaxf(v, lane_xor_f32<4>(v, lane));
    v = fmaxf(v, lane_xor_f32<2>(v, lane));
    v = fmaxf(v, lane_xor_f32<1>(v, lane));
    return v;
}

// LayerNorm of one 384-wide row held as 6 values per lane (element d = lane + 64*i). model.rs:86-104:
// mean = sum/H ; xc = x - mean ; var = sum(xc^2)/H (biased) ; xc / sqrt(var + eps) * gamma + beta
// planes != NULL: row t is also written into three K-blocked bf16 planes (embed_gemm3.hip: the dense layer that follows
// reads those; plane_stride = rows_alloc * H)
__device__ __forceinline__ void row_layer_norm(float (&v)[6], const float* __restrict__ g,
                                               const float* __restrict__ b, float eps, int lane,
                                               float* __restrict__ out, uint16_t* __restrict__ planes = nullptr,
                                               size_t plane_stride = 0, size_t t = 0) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 6; ++i) s += v[i];
    const float inv_h = 1.0f / (float)H;  // candle: Tensor / f64 == affine(1/rhs, 0)
    const float mean = wave_allreduce_sum(s) * inv_h;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        v[i] -= mean;
        q += v[i] * v[i];
    }
    const float var = wave_allreduce_sum(q) * inv_h;
    const float den = sqrtf(var + eps);
    float y[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        const int d = lane + 64 * i;
        y[i] = (v[i] / den) * g[d] + b[d];
        out[d] = y[i];
    }
    if (planes) {
#pragma unroll
        for (int i = 0; i < 6; i += 2) {
            uint32_t w[3];
            split3_bf16_pair(y[i], y[i + 1], w[0], w[1], w[2]);
            const size_t o0 = plane_index(t, lane + 64 * i, plane_stride / H);  // (32 lanes: 64 contiguous bytes)
            const size_t o1 = plane_index(t, lane + 64 * (i + 1), plane_stride / H);
#pragma unroll
            for (int p = 0; p < 3; ++p) {
                planes[p * plane_stride + o0] = (uint16_t)(w[p] & 0xFFFFu);
                planes[p * plane_stride + o1] = (uint16_t)(w[p] >> 16);
            }
        }
    }
}

// one wave per token
__global__ __launch_bounds__(256) void embed_ln_kernel(const uint32_t* __restrict__ ids,
                                                      const int* __restrict__ tok_pos, int T,
                                                      const float* __restrict__ word,
                                                      const float* __restrict__ pos,
                                                      const float* __restrict__ type0,
                                                      const float* __restrict__ g, const float* __restrict__ b,
                                                      float eps, float* __restrict__ x, uint16_t* __restrict__ xp,
                                                      size_t plane_stride, const int* __restrict__ seq_offsets, int B) {
    const int lane = threadIdx.x & 63;
    const int t = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (t >= T) return;
    int p;
    if (tok_pos) {
        p = tok_pos[t];
    } else {  // a few sequences (one text per call): find this token's sequence here instead of in a launch of its own
        int b = 0;
        while (b + 1 < B && seq_offsets[b + 1] <= t) ++b;
        p = t - seq_offsets[b];
    }
    const float* we = word + (size_t)ids[t] * H;
    const float* pe = pos + (size_t)p * H;
    float v[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        const int d = lane + 64 * i;
        v[i] = (we[d] + type0[d]) + pe[d];  // model.rs:269-276 order
    }
    row_layer_norm(v, g, b, eps, lane, x + (size_t)t * H, xp, plane_stride, (size_t)t);
}

__global__ __launch_bounds__(256) void add_ln_kernel(const float* __restrict__ a, const float* __restrict__ r,
                                                    int T, const float* __restrict__ g,
                                                    const float* __restrict__ b, float eps,
                                                    float* __restrict__ out, uint16_t* __restrict__ outp,
                                                    size_t plane_stride, int a_parts, size_t a_part_stride) {
    const int lane = threadIdx.x & 63;
    const int t = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (t >= T) return;
    float v[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        const int d = lane + 64 * i;
        float av = a[(size_t)t * H + d];
        if (a_parts == 4) {  // (a split-K producer's partial sums, in order; all requested up front)
            const float p1 = a[a_part_stride + (size_t)t * H + d], p2 = a[2 * a_part_stride + (size_t)t * H + d],
                        p3 = a[3 * a_part_stride + (size_t)t * H + d];
            av = ((av + p1) + p2) + p3;
        } else if (a_parts == 2) {
            av += a[a_part_stride + (size_t)t * H + d];
        }
        v[i] = av + r[(size_t)t * H + d];
    }
    row_layer_norm(v, g, b, eps, lane, out + (size_t)t * H, outp, plane_stride, (size_t)t);
}

// ------------------------------------------------------------------------------------------------
// Y[M,N] = X[M,K] · W[N,K]ᵀ + bias (+ activation).  64x64 block tile, 4 waves of 32x32 (one
// v_mfma_f32_32x32x2_f32 accumulator each), K-step 32, double-buffered LDS with one barrier per K-step.
// Tiles are staged with 16-B global loads and ds_write_b128 into [row][36] images (144-B rows: the
// ds_read_b128 of 16 rows x one 16-B column covers every bank once).  Inside a K-step the 32 k-values are
// consumed in the order (j, 16+j), j = 0..15: lane half kh reads its 16 values k = 16kh..16kh+15 as four
// ds_read_b128, identically for both operands (a sum over k in a fixed, permuted order — still exact f32 FMAs).
// N % 64 == 0 and K % 32 == 0 for every MiniLM shape (384, 1152, 1536).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float act_apply(float v, int act) {
    if (act == 1) {  // candle gelu = tanh form (model.rs:31-35)
        // 0.5 v (1 + tanh u) = v / (1 + exp(-2u)): one v_exp_f32 and one division instead of ocml's tanhf (~40
        // instructions; 16 of them per lane sat in the epilogue of every FFN1 tile, a fifth of that tile's matrix time).
        // |error| <= 3 ulp of the result (exp2 and rcp are 1-ulp instructions; the argument is clamped so that exp stays
        // finite) — three orders of magnitude inside the 1e-5 bar on the unit embeddings.
        const float k = 0.7978845608028654f;
        const float u = k * v * (1.0f + 0.044715f * v * v);
        const float t = fminf(fmaxf(-2.0f * u, -80.0f), 80.0f);
        return v / (1.0f + __expf(t));
    }
    if (act == 2) return v > 0.f ? v : 0.f;  // HiddenAct::Relu
    return v;
}

constexpr int GT = 64;    // block tile (rows of X, rows of W)
constexpr int GK = 32;    // K-step
constexpr int GLD = 36;   // LDS row stride in floats

template <int ACT>
__global__ __launch_bounds__(256) void gemm_nt_kernel(const float* __restrict__ A, const float* __restrict__ W,
                                                     const float* __restrict__ bias, float* __restrict__ Y,
                                                     int M, int N, int K) {
    __shared__ __attribute__((aligned(16))) float As[2][GT * GLD];
    __shared__ __attribute__((aligned(16))) float Bs[2][GT * GLD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // XCD-aware tile order.  The grid is 1-D and a multiple of 8: the hardware deals consecutive workgroups to the 8
    // XCDs round-robin, each with its own L2.  Workgroup b works on tile L = (b % 8) * (grid / 8) + b / 8, so one XCD gets
    // a contiguous run of tiles — the N/64 tiles that share a 64-row strip of X sit on the SAME XCD back to back and the
    // strip is fetched into one L2 once (dealt in plain order every strip was pulled into up to 8 L2s, and the MFMA
    // phase of a K-step is about as long as an L2 miss).
    const int per_xcd = gridDim.x >> 3;
    const int tile = (int)(blockIdx.x & 7) * per_xcd + (int)(blockIdx.x >> 3);
    const int tiles_n = N / GT;
    if (tile >= tiles_n * ((M + GT - 1) / GT)) return;  // padding of the grid
    const int m0 = (tile / tiles_n) * GT, n0 = (tile % tiles_n) * GT;
    const int wm = (wave >> 1) * 32, wn = (wave & 1) * 32;
    // staging: thread -> rows lr, lr+32 ; 16-B column lc of the 32-wide K-step
    const int lr = tid >> 3, lc = (tid & 7) * 4;

    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;

    f32x4 ra[2], rb[2];
    auto gload = [&](int k0) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int r = lr + 32 * j;
            const int m = m0 + r;
            ra[j] = (m < M) ? *reinterpret_cast<const f32x4*>(A + (size_t)m * K + k0 + lc) : f32x4{0.f, 0.f, 0.f, 0.f};
            rb[j] = *reinterpret_cast<const f32x4*>(W + (size_t)(n0 + r) * K + k0 + lc);
        }
    };
    auto lstore = [&](int buf) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int r = lr + 32 * j;
            *reinterpret_cast<f32x4*>(&As[buf][r * GLD + lc]) = ra[j];
            *reinterpret_cast<f32x4*>(&Bs[buf][r * GLD + lc]) = rb[j];
        }
    };
    const int kh = lane >> 5;
    const int a_off = (wm + (lane & 31)) * GLD + 16 * kh;
    const int b_off = (wn + (lane & 31)) * GLD + 16 * kh;

    gload(0);
    lstore(0);
    __syncthreads();
    int buf = 0;
    for (int k0 = 0; k0 < K; k0 += GK) {
        const bool more = k0 + GK < K;
        if (more) gload(k0 + GK);
        f32x4 av[4], bv[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            av[c] = *reinterpret_cast<const f32x4*>(&As[buf][a_off + 4 * c]);
            bv[c] = *reinterpret_cast<const f32x4*>(&Bs[buf][b_off + 4 * c]);
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[c].x, bv[c].x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[c].y, bv[c].y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[c].z, bv[c].z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[c].w, bv[c].w, acc, 0, 0, 0);
        }
        if (more) lstore(buf ^ 1);  // the other buffer was last read before the previous barrier
        __syncthreads();
        buf ^= 1;
    }
    // C/D map of the 32x32 MFMA: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
    const int n = n0 + wn + (lane & 31);
    const float bvv = bias[n];
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
        const int row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
        const int m = m0 + wm + row;
        if (m < M) Y[(size_t)m * N + n] = act_apply(acc[reg] + bvv, ACT);
    }
}

// ------------------------------------------------------------------------------------------------
// Skinny form for few rows (the reference's call shape: ONE text per call, embedding_service.rs:161-163).
// The 64x64 kernel above is latency-bound there (N/64 blocks walking K in 12..48 dependent steps).  Here a block
// owns a 16-row x 16-column tile of Y, its NWV waves split K (K/NWV = 48 or 96 values each), every wave loads
// its operand fragments straight from global memory up front (no LDS staging, one latency), runs 12..24
// v_mfma_f32_16x16x4_f32, and the partial tiles are summed through LDS in wave order.
// Lane (r = l&15, kq = l>>4) loads A[m0+r][kb + 4kq .. +3] and W[n0+r][same] as one 16-B load per
// 16-wide k-step; MFMA j consumes element j of every lane (k = kb + 4kq + j).
// ------------------------------------------------------------------------------------------------

template <int ACT, int NWV>
__global__ __launch_bounds__(NWV * 64) void gemm_skinny16_kernel(const float* __restrict__ A,
                                                                const float* __restrict__ W,
                                                                const float* __restrict__ bias,
                                                                float* __restrict__ Y, int M, int N, int K) {
    // gridDim.z > 1: split K — block z takes K / gridDim.z values of k and writes its partial sums to Y + z M N (the bias goes
    // with z = 0, no activation); the consumer adds the parts up in order (launch_gemm_ln_nt / launch_add_ln*: a_parts).  The
    // FFN-down layer of a one-text forward (K = 1536) took 7.6 us as 24 workgroups of 16 waves; as 4 x 24 of 8 it is a 4.8-us launch
    // like the K = 384 layers.
    extern __shared__ __attribute__((aligned(16))) float part[];  // [NWV][4][64]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n0 = blockIdx.x * 16, m0 = blockIdx.y * 16;
    const int r = lane & 15, kq = lane >> 4;
    const int kz = K / gridDim.z;
    const int kw = kz / NWV;
    const int k_begin = blockIdx.z * kz + wave * kw;
    constexpr int MAXS = 6;
    const int steps = kw / 16;
    f32x4 av[MAXS], bv[MAXS];
    const float* arow = A + (size_t)(m0 + r) * K + k_begin + 4 * kq;
    const float* wrow = W + (size_t)(n0 + r) * K + k_begin + 4 * kq;
    const bool a_ok = m0 + r < M;
    // (requested with the operands: behind the barrier it is one more dependent round trip of a 4-us launch)
    const float bias_v = bias[n0 + r];
#pragma unroll
    for (int s = 0; s < MAXS; ++s) {
        if (s < steps) {
            av[s] = a_ok ? *reinterpret_cast<const f32x4*>(arow + 16 * s) : f32x4{0.f, 0.f, 0.f, 0.f};
            bv[s] = *reinterpret_cast<const f32x4*>(wrow + 16 * s);
        }
    }
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < MAXS; ++s) {
        if (s < steps) {
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s].x, bv[s].x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s].y, bv[s].y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s].z, bv[s].z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s].w, bv[s].w, acc, 0, 0, 0);
        }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) part[(wave * 4 + e) * 64 + lane] = acc[e];
    __syncthreads();
    // 256 tile elements (reg e, lane l): row = 4*(l>>4) + e, col = l&15
    if (tid < 256) {
        const int e = tid >> 6, l = tid & 63;
        float sum = part[e * 64 + l];
#pragma unroll
        for (int w = 1; w < NWV; ++w) sum += part[(w * 4 + e) * 64 + l];
        const int row = m0 + 4 * (l >> 4) + e;
        const int n = n0 + (l & 15);
        if (row < M) Y[blockIdx.z * (size_t)M * N + (size_t)row * N + n] = act_apply(blockIdx.z == 0 ? sum + bias_v : sum, ACT);
    }
}

// The same with LayerNorm(a + r) as a PROLOGUE (K = 384 = the hidden size, 8 waves): the two dense layers that follow a
// residual LayerNorm (FFN1 after attention-output, Q|K|V of the next layer after output) normalise the 16 rows they load
// anyway — every block recomputes the statistics of its strip (16 x 384 values it reads in any case) and the blocks of
// column 0 also write the normalised rows out (they are the next residual).  One launch less per LayerNorm: 11 of the 45
// dependent launches of a one-text forward (model.rs:374-379, 458-463 + :86-104 arithmetic: mean = sum / H; xc = x - mean;
// var = sum xc^2 / H; xc / sqrt(var + eps) * gamma + beta — the sums run over the block's lanes and waves in a fixed
// order of their own, f32 throughout).
template <int ACT, int PARTS>
__global__ __launch_bounds__(512) void gemm_skinny16_ln_kernel(const float* __restrict__ Aa, const float* __restrict__ Ar,
                                                              const float* __restrict__ gam, const float* __restrict__ bet,
                                                              float eps, float* __restrict__ Xout,
                                                              const float* __restrict__ W, const float* __restrict__ bias,
                                                              float* __restrict__ Y, int M, int N, size_t a_part_stride, EmbSrc emb) {
    // PARTS > 1: `a` is a split-K producer's partial sums, PARTS arrays a_part_stride apart, added up in order (all requested up front).
    // PARTS == 0: the rows are BertEmbeddings (model.rs:266-281: word[ids] + type[0] + pos, the sum LayerNorm'ed with Aa / Ar unused) —
    // the first layer's Q|K|V of a one-text forward takes the embedding launch with it
    constexpr int NWV = 8, K = H, STEPS = K / NWV / 16;  // 48 k-values per wave and row: 3 steps of 16
    __shared__ __attribute__((aligned(16))) float part[NWV * 4 * 64];
    __shared__ float red[NWV][16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n0 = blockIdx.x * 16, m0 = blockIdx.y * 16;
    const int r = lane & 15, kq = lane >> 4;
    const int k_begin = wave * (K / NWV) + 4 * kq;
    const bool a_ok = m0 + r < M;
    f32x4 av[STEPS], bv[STEPS], g4[STEPS], b4[STEPS];
    const float* wrow = W + (size_t)(n0 + r) * K + k_begin;
    // gamma, beta and the bias are requested with the operands: loaded where they are used — behind the reductions' barriers —
    // they were three more dependent round trips (hipcc does not move a load across a barrier), ~2 us of a 6-8-us launch
    const float bias_v = bias[n0 + r];
#pragma unroll
    for (int st = 0; st < STEPS; ++st) {
        g4[st] = *reinterpret_cast<const f32x4*>(gam + k_begin + 16 * st);
        b4[st] = *reinterpret_cast<const f32x4*>(bet + k_begin + 16 * st);
    }
#pragma unroll
    for (int st = 0; st < STEPS; ++st) {
        const size_t o = (size_t)(m0 + r) * K + k_begin + 16 * st;
        const size_t oc = a_ok ? o : (size_t)(k_begin + 16 * st);  // (rows past M: row 0, discarded)
        if constexpr (PARTS == 0) {
            const int t = a_ok ? m0 + r : 0;
            int b = 0;
            while (b + 1 < emb.B && emb.seq_offsets[b + 1] <= t) ++b;
            const int kk = k_begin + 16 * st;
            const f32x4 we = *reinterpret_cast<const f32x4*>(emb.word + (size_t)emb.ids[t] * K + kk);
            const f32x4 ty = *reinterpret_cast<const f32x4*>(emb.type0 + kk);
            const f32x4 pe = *reinterpret_cast<const f32x4*>(emb.pos + (size_t)(t - emb.seq_offsets[b]) * K + kk);
            av[st] = a_ok ? (we + ty) + pe : f32x4{0.f, 0.f, 0.f, 0.f};  // model.rs:269-276 order
        } else {
            f32x4 ap[PARTS > 0 ? PARTS : 1];
#pragma unroll
            for (int z = 0; z < PARTS; ++z) ap[z] = *reinterpret_cast<const f32x4*>(Aa + z * a_part_stride + oc);
            const f32x4 rr = *reinterpret_cast<const f32x4*>(Ar + oc);
            f32x4 asum = ap[0];
#pragma unroll
            for (int z = 1; z < PARTS; ++z) asum = asum + ap[z];
            av[st] = a_ok ? asum + rr : f32x4{0.f, 0.f, 0.f, 0.f};
        }
        bv[st] = *reinterpret_cast<const f32x4*>(wrow + 16 * st);
    }
    // row statistics: this lane holds 12 of row r's 384 values; lanes r, r+16, r+32, r+48 of the 8 waves hold the rest
    auto row_total = [&](float v) {
        v += lane_xor_f32<16>(v, lane);
        v += lane_xor_f32<32>(v, lane);
        if (kq == 0) red[wave][r] = v;
        __syncthreads();
        float t = red[0][r];
#pragma unroll
        for (int w = 1; w < NWV; ++w) t += red[w][r];
        __syncthreads();
        return t;
    };
    float s1 = 0.f;
#pragma unroll
    for (int st = 0; st < STEPS; ++st) s1 += (av[st].x + av[st].y) + (av[st].z + av[st].w);
    const float inv_h = 1.0f / (float)H;
    const float mean = row_total(s1) * inv_h;
    float s2 = 0.f;
#pragma unroll
    for (int st = 0; st < STEPS; ++st) {
        av[st] = av[st] - mean;
        s2 += (av[st].x * av[st].x + av[st].y * av[st].y) + (av[st].z * av[st].z + av[st].w * av[st].w);
    }
    const float den = sqrtf(row_total(s2) * inv_h + eps);
#pragma unroll
    for (int st = 0; st < STEPS; ++st) {
        const int k = k_begin + 16 * st;
        av[st] = (av[st] / den) * g4[st] + b4[st];
        if (blockIdx.x == 0 && a_ok) *reinterpret_cast<f32x4*>(Xout + (size_t)(m0 + r) * K + k) = av[st];
    }
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int st = 0; st < STEPS; ++st) {
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[st].x, bv[st].x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[st].y, bv[st].y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[st].z, bv[st].z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[st].w, bv[st].w, acc, 0, 0, 0);
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) part[(wave * 4 + e) * 64 + lane] = acc[e];
    __syncthreads();
    if (tid < 256) {
        const int e = tid >> 6, l = tid & 63;
        float sum = part[e * 64 + l];
#pragma unroll
        for (int w = 1; w < NWV; ++w) sum += part[(w * 4 + e) * 64 + l];
        const int row = m0 + 4 * (l >> 4) + e;
        const int n = n0 + (l & 15);
        if (row < M) Y[(size_t)row * N + n] = act_apply(sum + bias_v, ACT);
    }
}

// Y = act(LN(a + r) . W^T + bias), x_out = LN(a + r); false: this shape does not take the fused form (the caller then
// runs add_ln + gemm)
bool launch_gemm_ln_nt(const float* a, const float* r, const float* g, const float* b, float eps, float* x_out,
                       const float* W, const float* bias, float* Y, int M, int N, int K, int act, hipStream_t s, int skinny_max_m,
                       int a_parts, size_t a_part_stride, const EmbSrc* emb) {
    if (M <= 0) return true;
    // measured (tools/embed_latency.py, device-resident loop): 12 tokens 0.170 -> 0.164 ms, 27 tokens 0.194 -> 0.189 ms per
    // forward; at 128 tokens the N/16 blocks of a strip each redoing its statistics cost more than the launch saves
    // (0.290 -> 0.310 ms): fused up to 64 rows only
    if (K != H || M > 64 || M > skinny_max_m || N % 16 != 0) return false;
    dim3 grid(N / 16, (M + 15) / 16), block(512);
    if (a_parts != 1 && a_parts != 2 && a_parts != 4) return false;
    const EmbSrc es = emb ? *emb : EmbSrc{};
#define DAWN_SKINNY_LN(ACT_, PARTS_) \
    hipLaunchKernelGGL((gemm_skinny16_ln_kernel<ACT_, PARTS_>), grid, block, 0, s, a, r, g, b, eps, x_out, W, bias, Y, M, N, a_part_stride, es)
    if (emb) {  // BertEmbeddings as the prologue (act 0: the first layer's Q|K|V)
        if (act != 0 || emb->B > 16) return false;
        DAWN_SKINNY_LN(0, 0);
    } else if (a_parts == 4) {
        if (act == 1) DAWN_SKINNY_LN(1, 4); else if (act == 2) DAWN_SKINNY_LN(2, 4); else DAWN_SKINNY_LN(0, 4);
    } else if (a_parts == 2) {
        if (act == 1) DAWN_SKINNY_LN(1, 2); else if (act == 2) DAWN_SKINNY_LN(2, 2); else DAWN_SKINNY_LN(0, 2);
    } else {
        if (act == 1) DAWN_SKINNY_LN(1, 1); else if (act == 2) DAWN_SKINNY_LN(2, 1); else DAWN_SKINNY_LN(0, 1);
    }
#undef DAWN_SKINNY_LN
    return true;
}

template <int NWV>
static void launch_skinny16(const float* A, const float* W, const float* bias, float* Y, int M, int N, int K, int act,
                            hipStream_t s, int splits = 1) {
    dim3 grid(N / 16, (M + 15) / 16, splits), block(NWV * 64);
    const size_t lds = (size_t)NWV * 4 * 64 * sizeof(float);
    if (act == 1) hipLaunchKernelGGL((gemm_skinny16_kernel<1, NWV>), grid, block, lds, s, A, W, bias, Y, M, N, K);
    else if (act == 2) hipLaunchKernelGGL((gemm_skinny16_kernel<2, NWV>), grid, block, lds, s, A, W, bias, Y, M, N, K);
    else hipLaunchKernelGGL((gemm_skinny16_kernel<0, NWV>), grid, block, lds, s, A, W, bias, Y, M, N, K);
}

void launch_gemm_nt(const float* A, const float* W, const float* bias, float* Y, int M, int N, int K, int act,
                    hipStream_t s, bool tile_only, int skinny_max_m, int splits) {
    if (M <= 0) return;
    if (!tile_only && M <= skinny_max_m && N % 16 == 0) {  // one text per call: latency form (16-row strips x split K)
        if (K == 384) return launch_skinny16<8>(A, W, bias, Y, M, N, K, act, s);
        if (K == 1536 && (splits == 4 || splits == 2) && act == 0) return launch_skinny16<8>(A, W, bias, Y, M, N, K, 0, s, splits);  // parts at Y + z M N
        if (K == 1536) return launch_skinny16<16>(A, W, bias, Y, M, N, K, act, s);
    }
    const int n_tiles = (N / GT) * ((M + GT - 1) / GT);
    dim3 grid((n_tiles + 7) / 8 * 8), block(256);  // 1-D, padded to the 8 XCDs (see the kernel)
    if (act == 1) hipLaunchKernelGGL(gemm_nt_kernel<1>, grid, block, 0, s, A, W, bias, Y, M, N, K);
    else if (act == 2) hipLaunchKernelGGL(gemm_nt_kernel<2>, grid, block, 0, s, A, W, bias, Y, M, N, K);
    else hipLaunchKernelGGL(gemm_nt_kernel<0>, grid, block, 0, s, A, W, bias, Y, M, N, K);
}

// ------------------------------------------------------------------------------------------------
// attention, sequences of up to 64 tokens (queries): one block (4 waves) per (head, sequence), everything in LDS,
// three phases that each spread over all 256 threads — a 27-token text has 729 scores and 864 outputs, a thread per
// query row kept 27 lanes busy for 15 us:
//   1. scores  S[i][j] = (q_i . k_j) / sqrt(32): one thread per (i, j), the 32-long FMA chain in d order
//   2. softmax per row (wave per row, lanes over the keys): max, exp(s - max), sum — the max-subtract softmax of
//      candle_nn::ops::softmax
//   3. out[i][d] = (sum_j P[i][j] V[j][d]) / sum_i: one thread per (i, d)
// No mask inside a sequence; nothing outside it exists.  K rows are stored with stride 33 (phase 1 reads one key row
// per lane), scores with stride 65.
// ------------------------------------------------------------------------------------------------
constexpr int ATT_LD = DH + 1;

// ATT_SMAX = 32 or 64: longest sequence of the call (the LDS footprint, 17 / 42 KB, sets how many of the B x 12
// blocks a CU holds at once)
template <int ATT_SMAX>
__global__ __launch_bounds__(256) void attention_kernel(const float* __restrict__ qkv /*[T][1152]*/,
                                                       const int* __restrict__ seq_offsets,
                                                       float* __restrict__ ctx /*[T][384]*/) {
    __shared__ __attribute__((aligned(16))) float Qs[ATT_SMAX * DH];
    __shared__ float Ks[ATT_SMAX * ATT_LD];
    __shared__ __attribute__((aligned(16))) float Vs[ATT_SMAX * DH];
    __shared__ float Sc[ATT_SMAX * (ATT_SMAX + 1)];
    __shared__ float Sum[ATT_SMAX];
    const int h = blockIdx.x, b = blockIdx.y;
    const int start = seq_offsets[b];
    const int S = seq_offsets[b + 1] - start;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < S * (DH / 4); i += 256) {
        const int j = i >> 3, c = (i & 7) * 4;
        const float* row = qkv + (size_t)(start + j) * (3 * H) + h * DH + c;
        const f32x4 qq = *reinterpret_cast<const f32x4*>(row);
        const f32x4 kk = *reinterpret_cast<const f32x4*>(row + H);
        const f32x4 vv = *reinterpret_cast<const f32x4*>(row + 2 * H);
        *reinterpret_cast<f32x4*>(Qs + j * DH + c) = qq;
        *reinterpret_cast<f32x4*>(Vs + j * DH + c) = vv;
#pragma unroll
        for (int e = 0; e < 4; ++e) Ks[j * ATT_LD + c + e] = kk[e];
    }
    __syncthreads();
    const float inv_scale = (float)(1.0 / 5.656854249492381);  // 1/sqrt(32) as f32 (affine(1/rhs, 0))
    for (int idx = tid; idx < S * S; idx += 256) {
        const int i = idx / S, j = idx - i * S;
        float sc = 0.f;
#pragma unroll
        for (int d = 0; d < DH; ++d) sc = __builtin_fmaf(Qs[i * DH + d], Ks[j * ATT_LD + d], sc);
        Sc[i * (ATT_SMAX + 1) + j] = sc * inv_scale;
    }
    __syncthreads();
    for (int i = wave; i < S; i += 4) {
        const float sc = lane < S ? Sc[i * (ATT_SMAX + 1) + lane] : -__builtin_inff();
        const float mx = wave_allreduce_max(sc);
        const float p = lane < S ? expf(sc - mx) : 0.f;
        if (lane < S) Sc[i * (ATT_SMAX + 1) + lane] = p;
        const float sum = wave_allreduce_sum(p);
        if (lane == 0) Sum[i] = sum;
    }
    __syncthreads();
    for (int idx = tid; idx < S * DH; idx += 256) {
        const int i = idx >> 5, d = idx & 31;
        float acc = 0.f;
        for (int j = 0; j < S; ++j) acc = __builtin_fmaf(Sc[i * (ATT_SMAX + 1) + j], Vs[j * DH + d], acc);
        ctx[(size_t)(start + i) * H + h * DH + d] = acc / Sum[i];
    }
}

// ------------------------------------------------------------------------------------------------
// attention, sequences of 65..128 tokens (pages; the indexer embeds one page per call): one block (4 waves) per (head,
// sequence), both contractions on v_mfma_f32_32x32x2_f32 (exact f32 FMAs).  Wave w owns query rows 32w..32w+31:
//   scores: 4 key tiles x 16 MFMAs (A = Q rows, straight from global memory into registers; B = K rows from LDS, row
//   stride 33: conflict-free);
//   softmax in the accumulator layout (a lane holds one key column of 16 rows per tile: row max / row sum are a
//   5-step xor butterfly over the 32 lanes of its half-wave);
//   P.V one key tile at a time: the tile's 32 x 32 probabilities go through the wave's 4-KiB LDS strip and come back as the
//   A operand (16 MFMAs per tile, B = V rows).
// 50 KiB of LDS (K, V, four strips): three blocks per CU, so that one block's loads run under another's matrix work.
// (With Q and the whole 32 x 128 P strip of every wave in LDS — 117 KiB, one block per CU — a layer of 256 pages took
// 154 us; a thread per query row took 51 us per layer for ONE 128-token page.)
// ctx (f32) and / or ctxp (three bf16 planes, embed_gemm3.hip) are written; either may be NULL.
// ------------------------------------------------------------------------------------------------
constexpr int ATM_S = 128;            // keys / query rows covered
constexpr int ATM_LD = DH + 1;        // K, V, P row stride
constexpr int ATM_LDS_FLOATS = 2 * ATM_S * ATM_LD + 4 * 32 * ATM_LD;

__global__ __launch_bounds__(256, 3) void attention_mfma_kernel(const float* __restrict__ qkv /*[T][1152]*/,
                                                            const int* __restrict__ seq_offsets,
                                                            float* __restrict__ ctx /*[T][384]*/,
                                                            uint16_t* __restrict__ ctxp, size_t plane_stride) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* Ks = sm;
    float* Vs = sm + ATM_S * ATM_LD;
    const int h = blockIdx.x, b = blockIdx.y;
    const int start = seq_offsets[b];
    const int S = seq_offsets[b + 1] - start;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float* Ps = sm + 2 * ATM_S * ATM_LD + wave * (32 * ATM_LD);
    const int r = lane & 31, kh = lane >> 5;
    // this lane's Q fragment: Q[row 32w + r][2 kk + kh], kk = 0..15 (the two lanes of a row share its 128-B line)
    float qa[16];
    {
        const int qrow = wave * 32 + r;
        const float* q = qkv + (size_t)(start + (qrow < S ? qrow : 0)) * (3 * H) + h * DH;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            f32x4 t = *reinterpret_cast<const f32x4*>(q + 4 * c);
            if (qrow >= S) t = f32x4{0.f, 0.f, 0.f, 0.f};
            qa[2 * c] = kh ? t[1] : t[0];
            qa[2 * c + 1] = kh ? t[3] : t[2];
        }
    }
    for (int i = tid; i < ATM_S * (DH / 4); i += 256) {
        const int j = i >> 3, c = (i & 7) * 4;
        f32x4 kk = {0.f, 0.f, 0.f, 0.f}, vv = kk;  // rows past the sequence: zeros (masked below)
        if (j < S) {
            const float* row = qkv + (size_t)(start + j) * (3 * H) + h * DH + c;
            kk = *reinterpret_cast<const f32x4*>(row + H);
            vv = *reinterpret_cast<const f32x4*>(row + 2 * H);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            Ks[j * ATM_LD + c + e] = kk[e];
            Vs[j * ATM_LD + c + e] = vv[e];
        }
    }
    __syncthreads();
    if (wave * 32 >= S) return;  // none of this wave's query rows exists (no block barrier below)
    // ---- scores: acc[jt][e] = S[row 32w + 8(e>>2) + (e&3) + 4kh][key 32jt + r]
    // (kk outer, key tile inner: four independent accumulator chains — a dependent MFMA waits out its predecessor's 16 passes)
    f32x16 acc[4];
#pragma unroll
    for (int jt = 0; jt < 4; ++jt)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[jt][e] = 0.f;
#pragma unroll
    for (int kk = 0; kk < 16; ++kk)
#pragma unroll
        for (int jt = 0; jt < 4; ++jt)
            acc[jt] = __builtin_amdgcn_mfma_f32_32x32x2f32(qa[kk], Ks[(jt * 32 + r) * ATM_LD + 2 * kk + kh], acc[jt], 0, 0, 0);
    const float inv_scale = (float)(1.0 / 5.656854249492381);  // 1/sqrt(32) as f32 (affine(1/rhs, 0))
    float mx[16], sum[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) mx[e] = -__builtin_inff();
#pragma unroll
    for (int jt = 0; jt < 4; ++jt) {
        const bool key_ok = jt * 32 + r < S;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            acc[jt][e] = key_ok ? acc[jt][e] * inv_scale : -__builtin_inff();
            mx[e] = fmaxf(mx[e], acc[jt][e]);
        }
    }
#pragma unroll
    for (int e = 0; e < 16; ++e) {  // over the 32 lanes of this half-wave (the row's keys)
        mx[e] = fmaxf(mx[e], lane_xor_f32<16>(mx[e], lane));
        mx[e] = fmaxf(mx[e], lane_xor_f32<8>(mx[e], lane));
        mx[e] = fmaxf(mx[e], lane_xor_f32<4>(mx[e], lane));
        mx[e] = fmaxf(mx[e], lane_xor_f32<2>(mx[e], lane));
        mx[e] = fmaxf(mx[e], lane_xor_f32<1>(mx[e], lane));
        sum[e] = 0.f;
    }
    // ---- out = P.V, key tile by key tile: A = P[row r][key 2kk + kh] (this wave's strip), B = V[key][dim r].  The strip is
    // private to the wave and LDS operations of one wave complete in order: no barrier between its writes and reads.
    // (two accumulator chains: even and odd key pairs, added at the end)
    f32x16 o, o1;
#pragma unroll
    for (int e = 0; e < 16; ++e) o[e] = o1[e] = 0.f;
    const int n_jt = (S + 31) >> 5;  // key tiles beyond S have P = 0 and V = 0: skipped
#pragma unroll
    for (int jt = 0; jt < 4; ++jt) {
        if (jt < n_jt) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const float p = __expf(acc[jt][e] - mx[e]);  // exp(-inf) = 0 for the masked keys
                sum[e] += p;
                Ps[(8 * (e >> 2) + (e & 3) + 4 * kh) * ATM_LD + r] = p;
            }
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int kk = 0; kk < 16; kk += 2) {
                o = __builtin_amdgcn_mfma_f32_32x32x2f32(Ps[r * ATM_LD + 2 * kk + kh], Vs[(jt * 32 + 2 * kk + kh) * ATM_LD + r], o, 0, 0, 0);
                o1 = __builtin_amdgcn_mfma_f32_32x32x2f32(Ps[r * ATM_LD + 2 * kk + 2 + kh], Vs[(jt * 32 + 2 * kk + 2 + kh) * ATM_LD + r], o1, 0, 0, 0);
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
#pragma unroll
    for (int e = 0; e < 16; ++e) o[e] += o1[e];
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        sum[e] += lane_xor_f32<16>(sum[e], lane);
        sum[e] += lane_xor_f32<8>(sum[e], lane);
        sum[e] += lane_xor_f32<4>(sum[e], lane);
        sum[e] += lane_xor_f32<2>(sum[e], lane);
        sum[e] += lane_xor_f32<1>(sum[e], lane);
    }
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int lrow = 8 * (e >> 2) + (e & 3) + 4 * kh;
        const int row = wave * 32 + lrow;
        const float v = o[e] / sum[e];
        if (ctx && row < S) ctx[(size_t)(start + row) * H + h * DH + r] = v;
        if (ctxp) Ps[lrow * ATM_LD + r] = v;
    }
    if (ctxp) {
        // the 32 x 32 tile back by row: 8 consecutive dims per lane, split into the three bf16 planes, one 16-B store each
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int ch = it * 64 + lane, lrow = ch >> 2, c = ch & 3;
            const int row = wave * 32 + lrow;
            uint32_t w[3][4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                split3_bf16_pair(Ps[lrow * ATM_LD + c * 8 + 2 * e], Ps[lrow * ATM_LD + c * 8 + 2 * e + 1], w[0][e], w[1][e], w[2][e]);
            }
            if (row < S) {
                uint16_t* dst = ctxp + plane_index((size_t)(start + row), h * DH + c * 8, plane_stride / H);  // head h = k-block h
#pragma unroll
                for (int p = 0; p < 3; ++p)
                    *reinterpret_cast<u32x4*>(dst + p * plane_stride) = u32x4{w[p][0], w[p][1], w[p][2], w[p][3]};
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// attention for BATCHES of short sequences (up to 32 NT tokens, NT = 1 or 2; the query batch of the search path): one WAVE
// per (head, sequence, tile of 32 query rows), four of them per block, no block barrier.  Same contractions and
// accumulator-layout softmax as attention_mfma_kernel, but every MFMA operand except P comes straight from global memory
// into registers (Q, K: this lane's row, 8 x 16 B; V: 16 keys x one dim, 32 lanes = one 128-B line) — a sequence's rows are
// read by this one wave (and its NT - 1 siblings), so there is nothing to share through LDS.  (The three-phase block
// kernel above, built for ONE text per call, took 28 us per layer for 256 queries: 3072 blocks x four barrier phases.)
// Single-tile scores and P.V run as two accumulator chains (even / odd k pairs), summed at the end.
// Writes the context as K-blocked bf16 planes (embed_gemm3.hip).
// ------------------------------------------------------------------------------------------------
template <int NT>
__global__ __launch_bounds__(256) void attention_wave_kernel(const float* __restrict__ qkv /*[T][1152]*/,
                                                            const int* __restrict__ seq_offsets, int B,
                                                            float* __restrict__ ctx /*[T][384], used when ctxp == NULL*/,
                                                            uint16_t* __restrict__ ctxp, size_t plane_stride) {
    __shared__ float strips[4 * 32 * ATM_LD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int item = blockIdx.y * 4 + wave;
    const int b = item / NT, rt = item % NT;
    if (b >= B) return;
    const int h = blockIdx.x;
    const int start = seq_offsets[b];
    const int S = seq_offsets[b + 1] - start;
    if (rt * 32 >= S) return;
    float* Ps = strips + wave * (32 * ATM_LD);
    const int r = lane & 31, kh = lane >> 5;
    const int n_jt = (S + 31) >> 5;
    // fragment of row `row` of Q (which = 0) or K (which = 1): element 2 kk + kh, kk = 0..15; zeros past the sequence
    auto row_frag = [&](int row, int which, float (&f)[16]) __attribute__((always_inline)) {
        const float* q = qkv + (size_t)(start + (row < S ? row : 0)) * (3 * H) + which * H + h * DH;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            f32x4 t = *reinterpret_cast<const f32x4*>(q + 4 * c);
            if (row >= S) t = f32x4{0.f, 0.f, 0.f, 0.f};
            f[2 * c] = kh ? t[1] : t[0];
            f[2 * c + 1] = kh ? t[3] : t[2];
        }
    };
    // every operand is requested up front, unconditionally (clamped addresses, zeros selected afterwards): one round trip
    float qa[16], kb[NT][16], vb[NT][16];
    row_frag(rt * 32 + r, 0, qa);
#pragma unroll
    for (int jt = 0; jt < NT; ++jt) row_frag(jt * 32 + r, 1, kb[jt]);
    // V fragments: vb[jt][kk] = V[key 32 jt + 2 kk + kh][dim r]
#pragma unroll
    for (int jt = 0; jt < NT; ++jt)
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) {
            const int key = jt * 32 + 2 * kk + kh;
            const float v = qkv[(size_t)(start + (key < S ? key : 0)) * (3 * H) + 2 * H + h * DH + r];
            vb[jt][kk] = key < S ? v : 0.f;
        }
    // ---- scores: acc[jt][e] = S[row 32 rt + 8(e>>2) + (e&3) + 4kh][key 32jt + r]
    f32x16 acc[NT];
#pragma unroll
    for (int jt = 0; jt < NT; ++jt) {
        f32x16 a0, a1;
#pragma unroll
        for (int e = 0; e < 16; ++e) a0[e] = a1[e] = 0.f;
#pragma unroll
        for (int kk = 0; kk < 16; kk += 2) {
            a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(qa[kk], kb[jt][kk], a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(qa[kk + 1], kb[jt][kk + 1], a1, 0, 0, 0);
        }
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[jt][e] = a0[e] + a1[e];
    }
    const float inv_scale = (float)(1.0 / 5.656854249492381);  // 1/sqrt(32) as f32 (affine(1/rhs, 0))
    float mx[16], sum[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) mx[e] = -__builtin_inff();
#pragma unroll
    for (int jt = 0; jt < NT; ++jt) {
        const bool key_ok = jt * 32 + r < S;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            acc[jt][e] = key_ok ? acc[jt][e] * inv_scale : -__builtin_inff();
            mx[e] = fmaxf(mx[e], acc[jt][e]);
        }
    }
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        mx[e] = fmaxf(mx[e], lane_xor_f32<16>(mx[e], lane));
        mx[e] = fmaxf(mx[e], lane_xor_f32<8>(mx[e], lane));
        mx[e] = fmaxf(mx[e], lane_xor_f32<4>(mx[e], lane));
        mx[e] = fmaxf(mx[e], lane_xor_f32<2>(mx[e], lane));
        mx[e] = fmaxf(mx[e], lane_xor_f32<1>(mx[e], lane));
        sum[e] = 0.f;
    }
    f32x16 o, o1;
#pragma unroll
    for (int e = 0; e < 16; ++e) o[e] = o1[e] = 0.f;
#pragma unroll
    for (int jt = 0; jt < NT; ++jt) {
        if (jt < n_jt) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const float p = __expf(acc[jt][e] - mx[e]);  // exp(-inf) = 0 for the masked keys
                sum[e] += p;
                Ps[(8 * (e >> 2) + (e & 3) + 4 * kh) * ATM_LD + r] = p;
            }
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int kk = 0; kk < 16; kk += 2) {
                o = __builtin_amdgcn_mfma_f32_32x32x2f32(Ps[r * ATM_LD + 2 * kk + kh], vb[jt][kk], o, 0, 0, 0);
                o1 = __builtin_amdgcn_mfma_f32_32x32x2f32(Ps[r * ATM_LD + 2 * kk + 2 + kh], vb[jt][kk + 1], o1, 0, 0, 0);
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        sum[e] += lane_xor_f32<16>(sum[e], lane);
        sum[e] += lane_xor_f32<8>(sum[e], lane);
        sum[e] += lane_xor_f32<4>(sum[e], lane);
        sum[e] += lane_xor_f32<2>(sum[e], lane);
        sum[e] += lane_xor_f32<1>(sum[e], lane);
        Ps[(8 * (e >> 2) + (e & 3) + 4 * kh) * ATM_LD + r] = (o[e] + o1[e]) / sum[e];
    }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int ch = it * 64 + lane, lrow = ch >> 2, c = ch & 3;
        const int row = rt * 32 + lrow;
        uint32_t w[3][4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            split3_bf16_pair(Ps[lrow * ATM_LD + c * 8 + 2 * e], Ps[lrow * ATM_LD + c * 8 + 2 * e + 1], w[0][e], w[1][e], w[2][e]);
        }
        if (row < S && !ctxp) {  // f32 context (the latency path): the same 8 values as two 16-B stores
            float* dst = ctx + (size_t)(start + row) * H + h * DH + c * 8;
            f32x4 v0, v1;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                v0[e] = Ps[lrow * ATM_LD + c * 8 + e];
                v1[e] = Ps[lrow * ATM_LD + c * 8 + 4 + e];
            }
            *reinterpret_cast<f32x4*>(dst) = v0;
            *reinterpret_cast<f32x4*>(dst + 4) = v1;
        } else if (row < S) {
            uint16_t* dst = ctxp + plane_index((size_t)(start + row), h * DH + c * 8, plane_stride / H);
#pragma unroll
            for (int p = 0; p < 3; ++p)
                *reinterpret_cast<u32x4*>(dst + p * plane_stride) = u32x4{w[p][0], w[p][1], w[p][2], w[p][3]};
        }
    }
}

// ------------------------------------------------------------------------------------------------
// ONE TEXT (round 5): attention of sequences of up to 32 tokens in REGISTERS — one wave per (head, sequence), no LDS, no barrier.
// model.rs:325-347: softmax(Q K^T / sqrt(32)) V.  The block kernel above spends 7.3 us on one 27-token text (twelve workgroups, three
// barrier-separated phases through LDS) next to dense layers of 4.7 us.  Here both products run TRANSPOSED so that nothing has to change
// lanes in between:
//   S^T = K Q^T   (v_mfma_f32_32x32x2_f32; accumulator: lane = QUERY ROW (+ 32: the other half of the keys), registers = 16 keys)
//   softmax over a row's keys = over the lane's 16 registers + one exchange with lane ^ 32
//   O^T = V^T P^T (the MFMA's k index is summed over: step e feeds key kappa(e, lane >> 5) = 8 (e >> 2) + (e & 3) + 4 (lane >> 5) — the
//                  key whose probability the lane holds in register e; the V fragment is loaded from global memory in that order)
// and a lane ends up with 16 context values of ITS row in four runs of four consecutive dimensions: four 16-B stores.
// (Also built and dropped: this + the attention-output dense layer in one launch, wave = head, the dense layer's K slice = the head's
// dimensions, twelve partial tiles summed through LDS — correct, and exactly as long as the two launches it replaced, 12.2 us: the
// workgroup's 576 f32 MFMAs of 64 cycles share one CU's four matrix pipes, 3.8 us that twelve CUs otherwise do side by side.)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void attention_regs_kernel(const float* __restrict__ qkv /*[T][1152]*/, const int* __restrict__ seq_offsets,
                                                            float* __restrict__ ctx /*[T][384]*/) {
    const int lane = threadIdx.x;
    const int c = lane & 31, hh = lane >> 5;
    const int w = blockIdx.x, b = blockIdx.y;  // head, sequence
    const int start = seq_offsets[b];
    const int S = seq_offsets[b + 1] - start;  // <= 32 (the launcher's condition)
    // row c of Q and of K (clamped: rows past the sequence are masked below), elements 2 kk + hh
    // Every load of the launch is requested before anything is computed: left to itself hipcc sinks the loads between the MFMAs
    // (request two, wait, multiply, request two ...) — eight memory round trips in a row instead of one, 4.7-5.2 us for a launch
    // whose arithmetic is 32 MFMAs.
    float qb[16], ka[16];
    f32x4 tq[8], tk[8];
    float va[16];  // V^T fragment of step e: V[key kappa(e, hh)][dim c]
    {
        const float* qr = qkv + (size_t)(start + (c < S ? c : 0)) * (3 * H) + w * DH;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            tq[j] = *reinterpret_cast<const f32x4*>(qr + 4 * j);
            tk[j] = *reinterpret_cast<const f32x4*>(qr + H + 4 * j);
        }
    }
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int key = 8 * (e >> 2) + (e & 3) + 4 * hh;
        va[e] = qkv[(size_t)(start + (key < S ? key : 0)) * (3 * H) + 2 * H + w * DH + c];
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        qb[2 * j] = hh ? tq[j][1] : tq[j][0];
        qb[2 * j + 1] = hh ? tq[j][3] : tq[j][2];
        ka[2 * j] = hh ? tk[j][1] : tk[j][0];
        ka[2 * j + 1] = hh ? tk[j][3] : tk[j][2];
    }
    // S^T[key][row]: A = K (m = key), B = Q (n = row)
    f32x16 st, st1;
#pragma unroll
    for (int e = 0; e < 16; ++e) st[e] = st1[e] = 0.f;
#pragma unroll
    for (int kk = 0; kk < 16; kk += 2) {
        st = __builtin_amdgcn_mfma_f32_32x32x2f32(ka[kk], qb[kk], st, 0, 0, 0);
        st1 = __builtin_amdgcn_mfma_f32_32x32x2f32(ka[kk + 1], qb[kk + 1], st1, 0, 0, 0);
    }
    const float inv_scale = (float)(1.0 / 5.656854249492381);  // 1/sqrt(32) as f32 (affine(1/rhs, 0))
    float p[16];
    float mx = -__builtin_inff();
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int key = 8 * (e >> 2) + (e & 3) + 4 * hh;
        p[e] = key < S ? (st[e] + st1[e]) * inv_scale : -__builtin_inff();
        mx = fmaxf(mx, p[e]);
    }
    mx = fmaxf(mx, lane_xor_f32<32>(mx, lane));
    float sum = 0.f;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        p[e] = __expf(p[e] - mx);  // exp(-inf) = 0 for the masked keys
        sum += p[e];
    }
    sum += lane_xor_f32<32>(sum, lane);
    // O^T[dim][row]: A = V^T (m = dim, k = key kappa(e, .)), B = P^T (n = row)
    f32x16 o, o1;
#pragma unroll
    for (int e = 0; e < 16; ++e) o[e] = o1[e] = 0.f;
#pragma unroll
    for (int e = 0; e < 16; e += 2) {
        o = __builtin_amdgcn_mfma_f32_32x32x2f32(va[e], p[e], o, 0, 0, 0);
        o1 = __builtin_amdgcn_mfma_f32_32x32x2f32(va[e + 1], p[e + 1], o1, 0, 0, 0);
    }
    // register e = context[row c][dim kappa(e, hh)]: dims 4 hh + 8 j + {0, 1, 2, 3}
    if (c < S) {
        float* dst = ctx + (size_t)(start + c) * H + w * DH + 4 * hh;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            *reinterpret_cast<f32x4*>(dst + 8 * j) = f32x4{(o[4 * j] + o1[4 * j]) / sum, (o[4 * j + 1] + o1[4 * j + 1]) / sum,
                                                            (o[4 * j + 2] + o1[4 * j + 2]) / sum, (o[4 * j + 3] + o1[4 * j + 3]) / sum};
    }
}

// Long sequences (max_len > 64: pages): one THREAD per query row, two passes over the keys (max, then exp/sum/PV);
// K and V unpadded in LDS (every thread reads the same key row: broadcasts).  At S = 128 this keeps two full waves
// busy per block; up to 64 tokens the three-phase kernel above is used.
__global__ __launch_bounds__(128) void attention_rows_kernel(const float* __restrict__ qkv /*[T][1152]*/,
                                                       const int* __restrict__ seq_offsets,
                                                       float* __restrict__ ctx /*[T][384]*/) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int h = blockIdx.x, b = blockIdx.y;
    const int start = seq_offsets[b];
    const int S = seq_offsets[b + 1] - start;
    float* Ks = sm;
    float* Vs = sm + (size_t)S * DH;
    for (int i = threadIdx.x; i < S * (DH / 4); i += blockDim.x) {
        const int j = i >> 3, c = (i & 7) * 4;
        const float* row = qkv + (size_t)(start + j) * (3 * H) + h * DH + c;
        *reinterpret_cast<f32x4*>(Ks + j * DH + c) = *reinterpret_cast<const f32x4*>(row + H);
        *reinterpret_cast<f32x4*>(Vs + j * DH + c) = *reinterpret_cast<const f32x4*>(row + 2 * H);
    }
    __syncthreads();
    const float inv_scale = (float)(1.0 / 5.656854249492381);  // 1/sqrt(32) as f32 (affine(1/rhs, 0))
    for (int i = threadIdx.x; i < S; i += blockDim.x) {
        float q[DH];
        const float* qr = qkv + (size_t)(start + i) * (3 * H) + h * DH;
#pragma unroll
        for (int d = 0; d < DH; d += 4) {
            const f32x4 t = *reinterpret_cast<const f32x4*>(qr + d);
            q[d] = t.x;
            q[d + 1] = t.y;
            q[d + 2] = t.z;
            q[d + 3] = t.w;
        }
        float mx = -__builtin_inff();
        for (int j = 0; j < S; ++j) {
            float s = 0.f;
#pragma unroll
            for (int d = 0; d < DH; ++d) s = __builtin_fmaf(q[d], Ks[j * DH + d], s);
            s *= inv_scale;
            mx = s > mx ? s : mx;
        }
        float sum = 0.f;
        float acc[DH];
#pragma unroll
        for (int d = 0; d < DH; ++d) acc[d] = 0.f;
        for (int j = 0; j < S; ++j) {
            float s = 0.f;
#pragma unroll
            for (int d = 0; d < DH; ++d) s = __builtin_fmaf(q[d], Ks[j * DH + d], s);
            const float p = expf(s * inv_scale - mx);
            sum += p;
#pragma unroll
            for (int d = 0; d < DH; ++d) acc[d] = __builtin_fmaf(p, Vs[j * DH + d], acc[d]);
        }
        float* o = ctx + (size_t)(start + i) * H + h * DH;
#pragma unroll
        for (int d = 0; d < DH; d += 4) {
            f32x4 t = {acc[d] / sum, acc[d + 1] / sum, acc[d + 2] / sum, acc[d + 3] / sum};
            *reinterpret_cast<f32x4*>(o + d) = t;
        }
    }
}

// one block (384 threads) per sequence: mean over its tokens, then L2 normalise (vector.rs:194-197)
__global__ __launch_bounds__(384) void pool_norm_kernel(const float* __restrict__ x, const int* __restrict__ seq_offsets,
                                                       float* __restrict__ out) {
    __shared__ float red[6];
    const int b = blockIdx.x, d = threadIdx.x;
    const int start = seq_offsets[b];
    const int S = seq_offsets[b + 1] - start;
    float s = 0.f;
    for (int t = 0; t < S; ++t) s += x[(size_t)(start + t) * H + d];
    const float m = s * (float)(1.0 / (double)S);  // sum(1) / (n_tokens as f64): affine(1/S, 0)
    float sq = wave_allreduce_sum(m * m);
    if ((d & 63) == 0) red[d >> 6] = sq;
    __syncthreads();
    const float tot = ((red[0] + red[1]) + (red[2] + red[3])) + (red[4] + red[5]);
    out[(size_t)b * H + d] = m / sqrtf(tot);
}

// The last LayerNorm of the encoder and the pooling in one launch (the latency form: one text per call): one block of 16 waves
// per sequence; a wave per token normalises a + r into x (the hidden states), then — after the block's barrier — thread d
// sums column d over the tokens in order, exactly as pool_norm_kernel does.
__global__ __launch_bounds__(1024) void add_ln_pool_norm_kernel(const float* __restrict__ a, const float* __restrict__ r,
                                                                const int* __restrict__ seq_offsets, const float* __restrict__ g,
                                                                const float* __restrict__ bta, float eps, float* __restrict__ x,
                                                                float* __restrict__ out, int a_parts, size_t a_part_stride) {
    __shared__ float red[6];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int start = seq_offsets[b];
    const int S = seq_offsets[b + 1] - start;
    for (int i = wave; i < S; i += 16) {
        const size_t t = (size_t)(start + i);
        float v[6];
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            const int d = lane + 64 * j;
            float av = a[t * H + d];
            if (a_parts == 4) {  // (all requested up front)
                const float p1 = a[a_part_stride + t * H + d], p2 = a[2 * a_part_stride + t * H + d], p3 = a[3 * a_part_stride + t * H + d];
                av = ((av + p1) + p2) + p3;
            } else if (a_parts == 2) {
                av += a[a_part_stride + t * H + d];
            }
            v[j] = av + r[t * H + d];
        }
        row_layer_norm(v, g, bta, eps, lane, x + t * H);
    }
    __syncthreads();  // (a workgroup's global writes are visible to it after its barrier)
    const int d = tid;
    float sq = 0.f, m = 0.f;
    if (d < H) {
        float s = 0.f;
        for (int t = 0; t < S; ++t) s += x[(size_t)(start + t) * H + d];
        m = s * (float)(1.0 / (double)S);
        sq = wave_allreduce_sum(m * m);
        if ((d & 63) == 0) red[d >> 6] = sq;
    }
    __syncthreads();
    if (d < H) {
        const float tot = ((red[0] + red[1]) + (red[2] + red[3])) + (red[4] + red[5]);
        out[(size_t)b * H + d] = m / sqrtf(tot);
    }
}

void launch_embed_ln(const uint32_t* ids, const int* tok_pos, int T, const float* word, const float* pos,
                     const float* type0, const float* g, const float* b, float eps, float* x, hipStream_t s,
                     uint16_t* xp, size_t plane_stride, const int* seq_offsets, int B) {
    if (T <= 0) return;
    hipLaunchKernelGGL(embed_ln_kernel, dim3((T + 3) / 4), dim3(256), 0, s, ids, tok_pos, T, word, pos, type0, g, b,
                       eps, x, xp, plane_stride, seq_offsets, B);
}

void launch_add_ln_pool_norm(const float* a, const float* r, const int* seq_offsets, int B, const float* g, const float* b,
                             float eps, float* x, float* out, hipStream_t s, int a_parts, size_t a_part_stride) {
    if (B <= 0) return;
    hipLaunchKernelGGL(add_ln_pool_norm_kernel, dim3(B), dim3(1024), 0, s, a, r, seq_offsets, g, b, eps, x, out, a_parts, a_part_stride);
}

void launch_add_ln(const float* a, const float* r, int T, const float* g, const float* b, float eps, float* out,
                   hipStream_t s, uint16_t* outp, size_t plane_stride, int a_parts, size_t a_part_stride) {
    if (T <= 0) return;
    hipLaunchKernelGGL(add_ln_kernel, dim3((T + 3) / 4), dim3(256), 0, s, a, r, T, g, b, eps, out, outp, plane_stride, a_parts, a_part_stride);
}

// attn_wave (embedder option "attention_wave"): 2 = sequences of up to 32 tokens take the three-phase block kernel instead of the register
// form (A/B, tests); 1 = sequences of up to 64 tokens always take attention_wave_kernel (0: only with planes
// — for ONE text the block kernel is as fast: 12 tokens 0.160 vs 0.173 ms per forward, 27 tokens 0.184 vs 0.181)
bool launch_attention(const float* qkv, const int* seq_offsets, int B, int max_len, float* ctx, hipStream_t s, uint16_t* ctxp,
                      size_t plane_stride, int attn_wave) {
    if (B <= 0) return false;
    if (max_len > 64 && max_len <= ATM_S) {
        // (with planes asked for, only the planes are written: the dense layer that follows reads nothing else)
        hipLaunchKernelGGL(attention_mfma_kernel, dim3(NH, B), dim3(256), ATM_LDS_FLOATS * sizeof(float), s, qkv, seq_offsets,
                           ctxp ? nullptr : ctx, ctxp, plane_stride);
        return ctxp != nullptr;
    }
    if (max_len > 64) {
        const size_t lds = (size_t)max_len * DH * 2 * sizeof(float);
        hipLaunchKernelGGL(attention_rows_kernel, dim3(NH, B), dim3(128), lds, s, qkv, seq_offsets, ctx);
        return false;
    }
    if (ctxp || attn_wave) {  // the wave-per-sequence form (planes asked for: the throughput path)
        if (max_len <= 32)
            hipLaunchKernelGGL(attention_wave_kernel<1>, dim3(NH, (B + 3) / 4), dim3(256), 0, s, qkv, seq_offsets, B, ctx, ctxp, plane_stride);
        else
            hipLaunchKernelGGL(attention_wave_kernel<2>, dim3(NH, (2 * B + 3) / 4), dim3(256), 0, s, qkv, seq_offsets, B, ctx, ctxp, plane_stride);
        return ctxp != nullptr;
    }
    if (max_len <= 32 && attn_wave != 2) hipLaunchKernelGGL(attention_regs_kernel, dim3(NH, B), dim3(64), 0, s, qkv, seq_offsets, ctx);
    else if (max_len <= 32) hipLaunchKernelGGL(attention_kernel<32>, dim3(NH, B), dim3(256), 0, s, qkv, seq_offsets, ctx);
    else hipLaunchKernelGGL(attention_kernel<64>, dim3(NH, B), dim3(256), 0, s, qkv, seq_offsets, ctx);
    return false;
}

void launch_pool_norm(const float* x, const int* seq_offsets, int B, float* out, hipStream_t s) {
    if (B <= 0) return;
    hipLaunchKernelGGL(pool_norm_kernel, dim3(B), dim3(384), 0, s, x, seq_offsets, out);
}

// token -> position-in-sequence (position_ids restart at 0 per sequence, model.rs:274)
__global__ void tok_pos_kernel(const int* __restrict__ seq_offsets, int B, int* __restrict__ tok_pos) {
    const int b = blockIdx.x;
    const int start = seq_offsets[b], end = seq_offsets[b + 1];
    for (int t = start + threadIdx.x; t < end; t += blockDim.x) tok_pos[t] = t - start;
}

void launch_tok_pos(const int* seq_offsets, int B, int* tok_pos, hipStream_t s) {
    if (B <= 0) return;
    hipLaunchKernelGGL(tok_pos_kernel, dim3(B), dim3(128), 0, s, seq_offsets, B, tok_pos);
}

int attention_set_max_lds() {
    // S = 512 needs 128 KiB of dynamic LDS: raise the kernel's limit once
    const hipError_t e1 = hipFuncSetAttribute(reinterpret_cast<const void*>(attention_mfma_kernel),
                                              hipFuncAttributeMaxDynamicSharedMemorySize,
                                              ATM_LDS_FLOATS * (int)sizeof(float));
    if (e1 != hipSuccess) return (int)e1;
    return (int)hipFuncSetAttribute(reinterpret_cast<const void*>(attention_rows_kernel),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 512 * DH * 2 * (int)sizeof(float));
}

}  // namespace dawn
