// embed_kernels.hpp — launchers of the MiniLM forward kernels (embed_kernels.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace dawn {
// rows up to which launch_gemm_nt takes the skinny (split-K latency) form by default (measured crossover with the 64x64 tile kernel:
// ~768 tokens); per embedder: option "skinny_max_rows"
constexpr int kSkinnyMaxM = 640;
void launch_tok_pos(const int* seq_offsets, int B, int* tok_pos, hipStream_t s);
// xp / outp != NULL: the rows are also written as three bf16 planes (K-blocked, see plane_index; plane_stride = rows_alloc * 384)
void launch_embed_ln(const uint32_t* ids, const int* tok_pos, int T, const float* word, const float* pos,
                     const float* type0, const float* g, const float* b, float eps, float* x, hipStream_t s,
                     uint16_t* xp = nullptr, size_t plane_stride = 0, const int* seq_offsets = nullptr, int B = 0);
// x = LayerNorm(a + r) and out[b] = normalise(mean over the tokens of sequence b of x) in one launch (a block per sequence)
void launch_add_ln_pool_norm(const float* a, const float* r, const int* seq_offsets, int B, const float* g, const float* b,
                             float eps, float* x, float* out, hipStream_t s, int a_parts = 1, size_t a_part_stride = 0);
void launch_add_ln(const float* a, const float* r, int T, const float* g, const float* b, float eps, float* out,
                   hipStream_t s, uint16_t* outp = nullptr, size_t plane_stride = 0, int a_parts = 1, size_t a_part_stride = 0);
// Y[M,N] = X[M,K]·W[N,K]^T + bias ; act: 0 none, 1 tanh-GELU, 2 ReLU.  N % 64 == 0, K % 32 == 0.  tile_only: never the
// split-K latency form (test / timing hooks ask for the 64x64 tile kernel whatever M is)
void launch_gemm_nt(const float* A, const float* W, const float* bias, float* Y, int M, int N, int K, int act,
                    hipStream_t s, bool tile_only = false, int skinny_max_m = kSkinnyMaxM, int splits = 1);
// (splits = 4, K = 1536, act = 0, latency form only: four partial sums at Y + z M N, to be added up by the consumer — a_parts below)
// Y = act(LN(a + r) . W^T + bias) and x_out = LN(a + r) in ONE launch (latency form: M <= skinny limit, K = 384); false =
// not applicable to this shape
// emb != NULL: the rows are BertEmbeddings of the token ids (a, r unused; <= 16 sequences; act 0) — g / b the embeddings' LayerNorm
struct EmbSrc {
    const uint32_t* ids = nullptr;
    const int* seq_offsets = nullptr;
    int B = 0;
    const float* word = nullptr;
    const float* pos = nullptr;
    const float* type0 = nullptr;
};
bool launch_gemm_ln_nt(const float* a, const float* r, const float* g, const float* b, float eps, float* x_out,
                       const float* W, const float* bias, float* Y, int M, int N, int K, int act, hipStream_t s,
                       int skinny_max_m = kSkinnyMaxM, int a_parts = 1, size_t a_part_stride = 0, const EmbSrc* emb = nullptr);
// Planes (embed_gemm3.hip) are K-BLOCKED: element (row, k) of one plane of a [rows_alloc x width] operand sits at
// ((k / 32) * rows_alloc + row) * 32 + k % 32 — the 32 values of k that one K-step of the dense kernels consumes are 64
// contiguous bytes, and the rows of a tile follow each other: a tile's K-step slice of a plane is ONE contiguous run (8 KiB
// for 128 rows), fetched in whole 128-B lines.  (Row-major planes gave 64-B pieces, every line requested twice: the
// L2 -> LDS stream of the 128 x 128 kernel ran at 17 B/clk/CU and set its time.)  plane p of an operand starts at
// p * rows_alloc * width.
#if defined(__HIPCC__)
__host__ __device__ __forceinline__ size_t plane_index(size_t row, int k, size_t rows_alloc) {
    return ((size_t)(k >> 5) * rows_alloc + row) * 32 + (size_t)(k & 31);
}
// f32 -> three bf16 values (bit patterns) with a = b1 + b2 + b3 up to 2^-24 |a|: b1 = bf16(a), b2 = bf16(a - b1),
// b3 = bf16(a - b1 - b2), round to nearest even, the subtractions exact (finite inputs)
__device__ __forceinline__ uint32_t bf16_rne_bits(float f) {
    const uint32_t u = __builtin_bit_cast(uint32_t, f);
    return (u + 0x7FFFu + ((u >> 16) & 1u)) >> 16;
}
__device__ __forceinline__ void split3_bf16(float a, uint32_t& b1, uint32_t& b2, uint32_t& b3) {
    b1 = bf16_rne_bits(a);
    const float r1 = a - __builtin_bit_cast(float, b1 << 16);
    b2 = bf16_rne_bits(r1);
    const float r2 = r1 - __builtin_bit_cast(float, b2 << 16);
    b3 = bf16_rne_bits(r2);
}
// the same for two values at once, on v_cvt_pk_bf16_f32 (round to nearest even in hardware; ~4.5 vector instructions per
// value instead of ~18): dword p = (part p of a) | (part p of c) << 16 — the order two neighbours have in a plane
typedef __bf16 dawn_bf16x2 __attribute__((ext_vector_type(2)));
typedef float dawn_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t bf16_pack2(float a, float c) {
    const dawn_f32x2 f = {a, c};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(f, dawn_bf16x2));
}
__device__ __forceinline__ void split3_bf16_pair(float a, float c, uint32_t& w1, uint32_t& w2, uint32_t& w3) {
    w1 = bf16_pack2(a, c);
    const float ra = a - __builtin_bit_cast(float, w1 << 16), rc = c - __builtin_bit_cast(float, w1 & 0xFFFF0000u);
    w2 = bf16_pack2(ra, rc);
    const float sa = ra - __builtin_bit_cast(float, w2 << 16), sc = rc - __builtin_bit_cast(float, w2 & 0xFFFF0000u);
    w3 = bf16_pack2(sa, sc);
}
#endif

// f32-accurate dense layer on the bf16 matrix cores (embed_gemm3.hip): operands as three bf16 planes each
// in: [rows][K] f32 row-major -> planes of a [rows_alloc x K] operand (K % 32 == 0, rows <= rows_alloc)
void launch_split_planes(const float* in, uint16_t* planes, int rows, int K, size_t rows_alloc, hipStream_t s);
// per-embedder tuning of the bf16x3 kernels (options "gemm3_stages", "gemm3_big_min_tiles", "gemm3_pingpong", "gemm3_persistent")
struct Gemm3Opts {
    int stages = 2;           // ring depth of the 64 x 64 kernel (2 .. 4)
    int big_min_tiles = 190;  // the 128 x 128 kernel is used from this many of its tiles (embed_gemm3.hip: launch_gemm_bf16x3)
    int pingpong = 1;         // 1 = the two waves of a SIMD of the 128 x 128 kernel run half a step apart
    int persistent = 256;     // workgroups of the 128 x 128 kernel (a multiple of 8; 0 = one per tile)
};
void launch_gemm_bf16x3(const uint16_t* Ap, size_t a_plane, const uint16_t* Wp, size_t w_plane, const float* bias, float* Y,
                        uint16_t* Yp, size_t y_plane, int M, int N, int K, int act, hipStream_t s, const Gemm3Opts& o = Gemm3Opts{});
// ctxp != NULL asks for the context as three bf16 planes (embed_gemm3.hip): true = written (and ctx is NOT), false = this
// sequence length has no plane-writing kernel: ctx was written, split it
bool launch_attention(const float* qkv, const int* seq_offsets, int B, int max_len, float* ctx, hipStream_t s,
                      uint16_t* ctxp = nullptr, size_t plane_stride = 0, int attn_wave = 0);
void launch_pool_norm(const float* x, const int* seq_offsets, int B, float* out, hipStream_t s);
int attention_set_max_lds();
}  // namespace dawn
