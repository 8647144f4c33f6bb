// embedder.cpp — dawn_embedder_* C ABI: the MI355X replacement of EmbeddingProvider
// (src/embedding/embedding_service.rs:49-139) minus the tokenizer.
//
// create  = EmbeddingProvider::new (:55-95) without the hub download: read config.json + model.safetensors
//           from disk, resolve the tensor names BertModel::load asks for (src/embedding/model.rs:235-255,
//           301-303,359-363,417,443-447,510,538-546 incl. the "bert." prefix retry :543-547 and the
//           LayerNorm gamma/beta fallback :210-222), upload to HBM.
// forward = calculate_embedding (:97-139) for packed token-id sequences.
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <exception>
#include <map>
#include <new>
#include <string>
#include <tuple>
#include <vector>

#include "common.hpp"
#include "embed_kernels.hpp"
#include "model_files.hpp"

using dawn::fail;

namespace {

using Config = dawn::BertConfig;

struct LayerW {
    float *qkv_w, *qkv_b, *ao_w, *ao_b, *ao_g, *ao_beta, *i_w, *i_b, *o_w, *o_b, *o_g, *o_beta;
    // the four dense weights as three bf16 planes each (embed_gemm3.hip), split once at load time
    uint16_t *qkv_p = nullptr, *ao_p = nullptr, *i_p = nullptr, *o_p = nullptr;
};

struct DevBuf {  // scratch device memory of the test / timing hooks
    void* p = nullptr;
    hipError_t alloc(size_t bytes) { return hipMalloc(&p, bytes); }
    ~DevBuf() {
        if (p) (void)hipFree(p);
    }
};
struct DevEvent {
    hipEvent_t e = nullptr;
    hipError_t create() { return hipEventCreate(&e); }
    ~DevEvent() {
        if (e) (void)hipEventDestroy(e);
    }
};

}  // namespace

struct dawn_embedder {
    int device = 0;
    Config cfg;
    hipStream_t stream = nullptr;
    float* d_weights = nullptr;  // one block
    float *word = nullptr, *pos = nullptr, *type0 = nullptr, *emb_g = nullptr, *emb_b = nullptr;
    std::vector<LayerW> layers;
    // workspaces (grown on demand)
    int cap_T = 0, cap_B = 0;
    float *x = nullptr, *qkv = nullptr, *ctx = nullptr, *tmp = nullptr, *tmp2 = nullptr, *attn = nullptr, *ff = nullptr;
    // bf16x3 path (batches above the skinny limit): activations that feed a dense layer, as planes [3][cap_T][width]
    uint16_t* d_wplanes = nullptr;
    uint16_t *xp = nullptr, *ctxp = nullptr, *attnp = nullptr, *ffp = nullptr;
    int use_bf16x3 = 1;  // option "gemm_bf16x3"
    int skinny_max_m = dawn::kSkinnyMaxM;  // option "skinny_max_rows": tokens up to which the GEMMs take the split-K latency form
    int attn_wave = 0;                     // option "attention_wave"
    dawn::Gemm3Opts g3{};                  // options "gemm3_*"
    int fused_embed = 1;                   // option "fused_embed": one text — BertEmbeddings as the prologue of the first Q|K|V launch
    int ffn2_split = 4;                    // option "ffn2_split": one text — the FFN-down layer as 4 (or 2) K-slices (partials summed by the next LayerNorm); 0 / 1: whole
    uint32_t* d_ids = nullptr;
    int *d_off = nullptr, *d_pos = nullptr;
    float* d_out = nullptr;
    // host API (dawn_embedder_forward): offsets | ids staged in ONE pinned block and copied with one command, the vectors stored by
    // the last kernel straight into pinned host memory (zero-copy): one text 0.210 -> 0.19 ms per call (two pageable copies in, one
    // out before).  Option "host_io" = 0: the three copy commands.
    int32_t* h_in = nullptr;   // pinned [offsets (B + 1, padded to 4) | ids (T)]
    int32_t* d_in = nullptr;   // its device copy
    size_t io_cap = 0;         // ints
    float* h_out = nullptr;    // pinned [B][hidden]
    size_t out_cap = 0;        // vectors
    int host_io = 1;
    // hipGraph replay of launch-bound forwards (one text = 45 kernels of 3-6 us): the launch sequence depends only on
    // (B, total tokens, longest sequence) and on the buffer addresses, so an instantiated graph is kept per such key and
    // replayed; the token ids / offsets are read on the device at run time.  Larger batches are GPU-bound: no graphs.
    struct GraphKey {
        int B, T, max_len;
        const void *ids, *off, *out;
        bool operator<(const GraphKey& o) const {
            return std::tie(B, T, max_len, ids, off, out) < std::tie(o.B, o.T, o.max_len, o.ids, o.off, o.out);
        }
    };
    std::map<GraphKey, hipGraphExec_t> graphs;
    std::map<GraphKey, int> graph_seen;  // a shape is captured the second time it shows up
    int use_graphs = 1;                  // option "graphs"
    int graph_max_tokens = 512;
    void drop_graphs() {
        for (auto& kv : graphs) (void)hipGraphExecDestroy(kv.second);
        graphs.clear();
        graph_seen.clear();
    }
};

namespace {

int ensure_ws(dawn_embedder* e, int T, int B) {
    if (T > e->cap_T || B > e->cap_B) e->drop_graphs();  // (they hold the old buffer addresses)
    if (T > e->cap_T) {
        float** bufs[] = {&e->x, &e->qkv, &e->ctx, &e->tmp, &e->tmp2, &e->attn, &e->ff};
        for (float** b : bufs)
            if (*b) {
                (void)hipFree(*b);
                *b = nullptr;
            }
        uint16_t** pbufs[] = {&e->xp, &e->ctxp, &e->attnp, &e->ffp};
        for (uint16_t** b : pbufs)
            if (*b) {
                (void)hipFree(*b);
                *b = nullptr;
            }
        if (e->d_ids) (void)hipFree(e->d_ids);
        if (e->d_pos) (void)hipFree(e->d_pos);
        e->d_ids = nullptr;
        e->d_pos = nullptr;
        e->cap_T = 0;
        const int cap = std::max(T, 256);
        const size_t H = e->cfg.hidden_size, I = e->cfg.intermediate_size;
        DAWN_HIP_TRY(hipMalloc((void**)&e->x, cap * H * 4));
        DAWN_HIP_TRY(hipMalloc((void**)&e->qkv, cap * 3 * H * 4));
        DAWN_HIP_TRY(hipMalloc((void**)&e->ctx, cap * H * 4));
        DAWN_HIP_TRY(hipMalloc((void**)&e->tmp, cap * H * 4));
        DAWN_HIP_TRY(hipMalloc((void**)&e->tmp2, 4 * cap * H * 4));  // (up to four split-K partial sums of the FFN-down layer)
        DAWN_HIP_TRY(hipMalloc((void**)&e->attn, cap * H * 4));
        DAWN_HIP_TRY(hipMalloc((void**)&e->ff, cap * I * 4));
        if (cap > e->skinny_max_m) {  // planes are only used by the tile path
            DAWN_HIP_TRY(hipMalloc((void**)&e->xp, (size_t)3 * cap * H * 2));
            DAWN_HIP_TRY(hipMalloc((void**)&e->ctxp, (size_t)3 * cap * H * 2));
            DAWN_HIP_TRY(hipMalloc((void**)&e->attnp, (size_t)3 * cap * H * 2));
            DAWN_HIP_TRY(hipMalloc((void**)&e->ffp, (size_t)3 * cap * I * 2));
        }
        DAWN_HIP_TRY(hipMalloc((void**)&e->d_ids, cap * sizeof(uint32_t)));
        DAWN_HIP_TRY(hipMalloc((void**)&e->d_pos, cap * sizeof(int)));
        e->cap_T = cap;
    }
    if (B > e->cap_B) {
        if (e->d_off) (void)hipFree(e->d_off);
        if (e->d_out) (void)hipFree(e->d_out);
        e->d_off = nullptr;
        e->d_out = nullptr;
        e->cap_B = 0;
        const int cap = std::max(B, 64);
        DAWN_HIP_TRY(hipMalloc((void**)&e->d_off, (cap + 1) * sizeof(int)));
        DAWN_HIP_TRY(hipMalloc((void**)&e->d_out, (size_t)cap * e->cfg.hidden_size * 4));
        e->cap_B = cap;
    }
    return DAWN_OK;
}

// BertModel::forward on packed tokens already on the device; result (last hidden states) in e->x.
// d_pool_out != NULL: the caller wants the pooled unit vectors too; true = this call produced them (the latency form fuses
// the pooling into its last LayerNorm launch), false = launch_pool_norm on e->x is still to be done.
bool encoder_forward(dawn_embedder* e, const uint32_t* d_ids, const int* d_off, int B, int T, int max_len,
                     hipStream_t s, float* d_pool_out = nullptr) {
    const Config& c = e->cfg;
    const int H = c.hidden_size, I = c.intermediate_size;
    const float eps = (float)c.layer_norm_eps;
    // token positions: a launch of their own for batches; found inside embed_ln for a few sequences (one text per call)
    const int* d_pos = nullptr;
    if (B > 16) {
        dawn::launch_tok_pos(d_off, B, e->d_pos, s);
        d_pos = e->d_pos;
    }
    if (e->use_bf16x3 && T > e->skinny_max_m && e->xp && e->d_wplanes) {
        // Throughput form: the dense layers run f32-accurately on the bf16 matrix cores (embed_gemm3.hip: 3-way bf16 split,
        // 6 products).  Whatever feeds a dense layer is produced as three bf16 planes by the kernel that computes it (the
        // LayerNorms beside their f32 output — the residual —, FFN1's GELU epilogue and the page attention instead of it; the
        // attention kernels of other sequence lengths write f32, split by one extra pass).
        const size_t ps = (size_t)e->cap_T * H, psi = (size_t)e->cap_T * I;  // plane strides
        dawn::launch_embed_ln(d_ids, d_pos, T, e->word, e->pos, e->type0, e->emb_g, e->emb_b, eps, e->x, s, e->xp, ps, d_off, B);
        for (const LayerW& L : e->layers) {
            dawn::launch_gemm_bf16x3(e->xp, ps, L.qkv_p, (size_t)3 * H * H, L.qkv_b, e->qkv, nullptr, 0, T, 3 * H, H, 0, s, e->g3);
            if (!dawn::launch_attention(e->qkv, d_off, B, max_len, e->ctx, s, e->ctxp, ps, e->attn_wave))
                dawn::launch_split_planes(e->ctx, e->ctxp, T, H, (size_t)e->cap_T, s);
            dawn::launch_gemm_bf16x3(e->ctxp, ps, L.ao_p, (size_t)H * H, L.ao_b, e->tmp, nullptr, 0, T, H, H, 0, s, e->g3);
            dawn::launch_add_ln(e->tmp, e->x, T, L.ao_g, L.ao_beta, eps, e->attn, s, e->attnp, ps);
            dawn::launch_gemm_bf16x3(e->attnp, ps, L.i_p, (size_t)I * H, L.i_b, nullptr, e->ffp, psi, T, I, H, c.act, s, e->g3);
            dawn::launch_gemm_bf16x3(e->ffp, psi, L.o_p, (size_t)H * I, L.o_b, e->tmp2, nullptr, 0, T, H, I, 0, s, e->g3);
            dawn::launch_add_ln(e->tmp2, e->attn, T, L.o_g, L.o_beta, eps, e->x, s, e->xp, ps);
        }
        return false;
    }
    // one text (<= 64 tokens, <= 16 sequences): BertEmbeddings is the prologue of the first layer's Q|K|V (option "fused_embed")
    bool emb_pending = e->fused_embed && T <= 64 && T <= e->skinny_max_m && B <= 16 && !e->layers.empty();
    if (!emb_pending) dawn::launch_embed_ln(d_ids, d_pos, T, e->word, e->pos, e->type0, e->emb_g, e->emb_b, eps, e->x, s, nullptr, 0, d_off, B);
    // Latency form (few tokens: the reference's one text per call): the residual LayerNorms run as the prologue of the dense
    // layer that consumes them (launch_gemm_ln_nt) — `pending` = the output LayerNorm of the previous layer not applied
    // yet: x = LN(tmp2 + attn) is produced by this layer's Q|K|V launch.
    const LayerW* pending = nullptr;
    // one text: the FFN-down layer (K = 1536) as four K-slices whose partial sums the next LayerNorm adds up (option "ffn2_split")
    const int parts = (e->ffn2_split && T <= 64 && T <= e->skinny_max_m) ? e->ffn2_split : 1;
    const size_t part_stride = (size_t)T * H;
    for (const LayerW& L : e->layers) {  // BertLayer::forward model.rs:487-498
        // :327-329 (Q|K|V fused)
        if (emb_pending) {
            const dawn::EmbSrc es{d_ids, d_off, B, e->word, e->pos, e->type0};
            emb_pending = false;
            if (!dawn::launch_gemm_ln_nt(nullptr, nullptr, e->emb_g, e->emb_b, eps, e->x, L.qkv_w, L.qkv_b, e->qkv, T, 3 * H, H, 0, s,
                                         e->skinny_max_m, 1, 0, &es)) {
                dawn::launch_embed_ln(d_ids, d_pos, T, e->word, e->pos, e->type0, e->emb_g, e->emb_b, eps, e->x, s, nullptr, 0, d_off, B);
                dawn::launch_gemm_nt(e->x, L.qkv_w, L.qkv_b, e->qkv, T, 3 * H, H, 0, s, false, e->skinny_max_m);
            }
        } else if (!(pending && dawn::launch_gemm_ln_nt(e->tmp2, e->attn, pending->o_g, pending->o_beta, eps, e->x, L.qkv_w, L.qkv_b,
                                                 e->qkv, T, 3 * H, H, 0, s, e->skinny_max_m, parts, part_stride))) {
            if (pending) dawn::launch_add_ln(e->tmp2, e->attn, T, pending->o_g, pending->o_beta, eps, e->x, s, nullptr, 0, parts, part_stride);
            dawn::launch_gemm_nt(e->x, L.qkv_w, L.qkv_b, e->qkv, T, 3 * H, H, 0, s, false, e->skinny_max_m);
        }
        dawn::launch_attention(e->qkv, d_off, B, max_len, e->ctx, s, nullptr, 0, e->attn_wave);                      // :331-346
        dawn::launch_gemm_nt(e->ctx, L.ao_w, L.ao_b, e->tmp, T, H, H, 0, s, false, e->skinny_max_m);               // :376
        // :378 LayerNorm(dense + x) -> attn, then :427-428 intermediate dense + activation
        if (!dawn::launch_gemm_ln_nt(e->tmp, e->x, L.ao_g, L.ao_beta, eps, e->attn, L.i_w, L.i_b, e->ff, T, I, H, c.act, s, e->skinny_max_m)) {
            dawn::launch_add_ln(e->tmp, e->x, T, L.ao_g, L.ao_beta, eps, e->attn, s);
            dawn::launch_gemm_nt(e->attn, L.i_w, L.i_b, e->ff, T, I, H, c.act, s, false, e->skinny_max_m);
        }
        dawn::launch_gemm_nt(e->ff, L.o_w, L.o_b, e->tmp2, T, H, I, 0, s, false, e->skinny_max_m, parts);          // :460
        pending = &L;                                                                      // :462 LayerNorm(dense + attn)
    }
    if (pending && d_pool_out && B <= 64) {  // (a block per sequence: few sequences only)
        dawn::launch_add_ln_pool_norm(e->tmp2, e->attn, d_off, B, pending->o_g, pending->o_beta, eps, e->x, d_pool_out, s, parts, part_stride);
        return true;
    }
    if (pending) dawn::launch_add_ln(e->tmp2, e->attn, T, pending->o_g, pending->o_beta, eps, e->x, s, nullptr, 0, parts, part_stride);
    return false;
}

int check_sequences(const dawn_embedder* e, const uint32_t* ids, const int32_t* off, int B, int* T_out, int* max_len) {
    if (off[0] != 0) return fail(DAWN_ERR_INVALID_ARG, "seq_offsets[0] must be 0");
    int mx = 0;
    for (int b = 0; b < B; ++b) {
        const int len = off[b + 1] - off[b];
        if (len < 1) return fail(DAWN_ERR_INVALID_ARG, "sequence %d is empty", b);
        if (len > e->cfg.max_position_embeddings)
            return fail(DAWN_ERR_INVALID_ARG, "sequence %d has %d tokens > max_position_embeddings %d", b, len,
                        e->cfg.max_position_embeddings);
        mx = std::max(mx, len);
    }
    const int T = off[B];
    for (int t = 0; t < T; ++t)
        if (ids[t] >= (uint32_t)e->cfg.vocab_size)
            return fail(DAWN_ERR_INVALID_ARG, "token id %u at %d >= vocab_size %d", ids[t], t, e->cfg.vocab_size);
    *T_out = T;
    *max_len = mx;
    return DAWN_OK;
}

// the pinned staging of the host API, grown on demand (graphs hold the old addresses)
int ensure_host_io(dawn_embedder* e, int T, int B) {
    const size_t need = (((size_t)B + 1 + 3) & ~(size_t)3) + (size_t)T;
    if (need > e->io_cap) {
        e->drop_graphs();
        if (e->h_in) (void)hipHostFree(e->h_in);
        if (e->d_in) (void)hipFree(e->d_in);
        e->h_in = nullptr;
        e->d_in = nullptr;
        e->io_cap = 0;
        const size_t cap = std::max<size_t>(need, 1024);
        DAWN_HIP_TRY(hipHostMalloc((void**)&e->h_in, cap * 4, hipHostMallocDefault));
        DAWN_HIP_TRY(hipMalloc((void**)&e->d_in, cap * 4));
        e->io_cap = cap;
    }
    if ((size_t)B > e->out_cap) {
        e->drop_graphs();
        if (e->h_out) (void)hipHostFree(e->h_out);
        e->h_out = nullptr;
        e->out_cap = 0;
        const size_t cap = std::max<size_t>((size_t)B, 64);
        DAWN_HIP_TRY(hipHostMalloc((void**)&e->h_out, cap * e->cfg.hidden_size * 4, hipHostMallocDefault));
        e->out_cap = cap;
    }
    return DAWN_OK;
}

int upload_inputs(dawn_embedder* e, const uint32_t* ids, const int32_t* off, int B, int T) {
    DAWN_TRY(ensure_ws(e, T, B));
    DAWN_HIP_TRY(hipMemcpyAsync(e->d_ids, ids, (size_t)T * 4, hipMemcpyHostToDevice, e->stream));
    DAWN_HIP_TRY(hipMemcpyAsync(e->d_off, off, (size_t)(B + 1) * 4, hipMemcpyHostToDevice, e->stream));
    return DAWN_OK;
}

}  // namespace

extern "C" {

static int embedder_create_impl(const char* safetensors_path, const char* config_json_path, int device, dawn_embedder** out);

int dawn_embedder_create(const char* safetensors_path, const char* config_json_path, int device,
                         dawn_embedder** out) {
    if (!out || !safetensors_path) return fail(DAWN_ERR_INVALID_ARG, "NULL argument");
    *out = nullptr;
    try {  // (std::vector / std::map / std::string allocations on file-sized data)
        return embedder_create_impl(safetensors_path, config_json_path, device, out);
    } catch (const std::bad_alloc&) {
        return fail(DAWN_ERR_OOM, "out of host memory while loading %s", safetensors_path);
    } catch (const std::exception& ex) {
        return fail(DAWN_ERR_IO, "%s: %s", safetensors_path, ex.what());
    } catch (...) {
        return fail(DAWN_ERR_IO, "%s: unexpected exception", safetensors_path);
    }
}

static int embedder_create_impl(const char* safetensors_path, const char* config_json_path, int device, dawn_embedder** out) {
    DAWN_TRY(dawn::require_device(device));
    DAWN_HIP_TRY(hipSetDevice(device));

    dawn::ModelHost m;
    DAWN_TRY(dawn::load_model_files(safetensors_path, config_json_path, m));
    const Config& cfg = m.cfg;
    const int H = cfg.hidden_size, I = cfg.intermediate_size, NL = cfg.num_hidden_layers;
    const size_t total = m.weights.size();
    const std::vector<float>& host = m.weights;
    const std::vector<dawn::LayerOffsets>& lo = m.layers;
    const size_t o_word = m.o_word, o_pos = m.o_pos, o_type = m.o_type, o_eg = m.o_eg, o_eb = m.o_eb;

    auto* e = new dawn_embedder();
    e->device = device;
    e->cfg = cfg;
    if (hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking) != hipSuccess ||
        hipMalloc((void**)&e->d_weights, total * 4) != hipSuccess ||
        hipMemcpy(e->d_weights, host.data(), total * 4, hipMemcpyHostToDevice) != hipSuccess) {
        dawn_embedder_destroy(e);
        return fail(DAWN_ERR_OOM, "uploading %zu weight bytes failed", total * 4);
    }
    (void)dawn::attention_set_max_lds();
    float* W = e->d_weights;
    e->word = W + o_word;
    e->pos = W + o_pos;
    e->type0 = W + o_type;  // token_type_ids are all zero (embedding_service.rs:123): row 0
    e->emb_g = W + o_eg;
    e->emb_b = W + o_eb;
    for (int L = 0; L < NL; ++L)
        e->layers.push_back({W + lo[L].qw, W + lo[L].qb, W + lo[L].aow, W + lo[L].aob, W + lo[L].aog, W + lo[L].aobeta,
                             W + lo[L].iw, W + lo[L].ib, W + lo[L].ow, W + lo[L].ob, W + lo[L].og, W + lo[L].obeta});
    // the dense weights once more as bf16 planes (6 B per weight: 64 MB for MiniLM-L6); without them the f32 path is used
    {
        const size_t per_layer = (size_t)3 * H * H + (size_t)H * H + (size_t)2 * I * H;
        if (hipMalloc((void**)&e->d_wplanes, per_layer * NL * 3 * sizeof(uint16_t)) == hipSuccess) {
            uint16_t* p = e->d_wplanes;
            for (LayerW& Lw : e->layers) {
                auto planes = [&](const float* w, int rows, int K) {
                    const size_t n = (size_t)rows * K;
                    uint16_t* at = p;
                    dawn::launch_split_planes(w, at, rows, K, (size_t)rows, e->stream);
                    p += 3 * n;
                    return at;
                };
                Lw.qkv_p = planes(Lw.qkv_w, 3 * H, H);
                Lw.ao_p = planes(Lw.ao_w, H, H);
                Lw.i_p = planes(Lw.i_w, I, H);
                Lw.o_p = planes(Lw.o_w, H, I);
            }
            if (hipStreamSynchronize(e->stream) != hipSuccess) {
                (void)hipFree(e->d_wplanes);
                e->d_wplanes = nullptr;
            }
        } else {
            (void)hipGetLastError();
            e->d_wplanes = nullptr;
        }
    }
    *out = e;
    return DAWN_OK;
}

void dawn_embedder_destroy(dawn_embedder* e) {
    if (!e) return;
    (void)hipSetDevice(e->device);
    if (e->stream) (void)hipStreamSynchronize(e->stream);
    e->drop_graphs();
    void* ptrs[] = {e->d_weights, e->d_wplanes, e->x, e->qkv, e->ctx, e->tmp, e->tmp2, e->attn, e->ff, e->xp, e->ctxp, e->attnp,
                    e->ffp, e->d_ids, e->d_off, e->d_pos, e->d_out, e->d_in};
    for (void* p : ptrs)
        if (p) (void)hipFree(p);
    if (e->h_in) (void)hipHostFree(e->h_in);
    if (e->h_out) (void)hipHostFree(e->h_out);
    if (e->stream) (void)hipStreamDestroy(e->stream);
    delete e;
}

static int embedder_set_option_impl(dawn_embedder* e, const char* name, int64_t value);
int dawn_embedder_set_option(dawn_embedder* e, const char* name, int64_t value) {
    return dawn::guarded([&] { return embedder_set_option_impl(e, name, value); });
}
static int embedder_set_option_impl(dawn_embedder* e, const char* name, int64_t value) {
    if (!e || !name) return fail(DAWN_ERR_INVALID_ARG, "NULL argument");
    if (std::string(name) == "skinny_max_rows") {  // token count up to which the GEMMs take the split-K skinny form
        if (value < 0 || value > 4096) return fail(DAWN_ERR_INVALID_ARG, "skinny_max_rows out of range");
        e->skinny_max_m = (int)value;
        return DAWN_OK;
    }
    if (std::string(name) == "host_io") {  // 0: dawn_embedder_forward moves its inputs and outputs with three copy commands (A/B, tests)
        e->host_io = value != 0;
        return DAWN_OK;
    }
    if (std::string(name) == "fused_embed") {  // 0: BertEmbeddings as a launch of its own (A/B, tests)
        e->fused_embed = value != 0;
        e->drop_graphs();
        return DAWN_OK;
    }
    if (std::string(name) == "ffn2_split") {  // 0: the FFN-down layer of a one-text forward in one piece (A/B, tests)
        if (value != 0 && value != 1 && value != 2 && value != 4) return fail(DAWN_ERR_INVALID_ARG, "ffn2_split must be 0, 1, 2 or 4");
        e->ffn2_split = value == 1 ? 4 : (int)value;  // (1: the default split)
        e->drop_graphs();
        return DAWN_OK;
    }
    if (std::string(name) == "gemm_bf16x3") {  // 0: batches use the f32-MFMA tile kernel instead of the bf16x3 kernels
        e->use_bf16x3 = value != 0;
        e->drop_graphs();
        return DAWN_OK;
    }
    if (std::string(name) == "gemm3_big_min_tiles") {  // 128 x 128 tiles from which that form of the bf16x3 kernel is used
        if (value < 0) return fail(DAWN_ERR_INVALID_ARG, "gemm3_big_min_tiles out of range");
        e->g3.big_min_tiles = (int)std::min<int64_t>(value, 1 << 30);
        return DAWN_OK;
    }
    if (std::string(name) == "attention_wave") {  // sequences of up to 64 tokens: 1 = always the wave-per-sequence kernel (tuning)
        if (value < 0 || value > 2) return fail(DAWN_ERR_INVALID_ARG, "attention_wave must be 0, 1 or 2");
        e->attn_wave = (int)value;
        e->drop_graphs();
        return DAWN_OK;
    }
    if (std::string(name) == "gemm3_persistent") {  // 128 x 128 kernel: blocks that walk the tile list (0 = a block per tile)
        if (value < 0 || value > 4096 || value % 8) return fail(DAWN_ERR_INVALID_ARG, "gemm3_persistent must be a multiple of 8 in 0..4096");
        e->g3.persistent = (int)value;
        return DAWN_OK;
    }
    if (std::string(name) == "gemm3_pingpong") {  // 128 x 128 kernel: waves of a SIMD half a step apart (tuning; default 1)
        if (value < 0 || value > 1) return fail(DAWN_ERR_INVALID_ARG, "gemm3_pingpong must be 0 or 1");
        e->g3.pingpong = (int)value;
        return DAWN_OK;
    }
    if (std::string(name) == "gemm3_stages") {  // ring depth of the bf16x3 kernel (tuning)
        if (value < 2 || value > 4) return fail(DAWN_ERR_INVALID_ARG, "gemm3_stages must be 2..4");
        e->g3.stages = (int)value;
        return DAWN_OK;
    }
    if (std::string(name) == "graphs") {  // 0: never replay hipGraphs (every forward is ~45 plain launches)
        e->use_graphs = value != 0;
        if (!value) e->drop_graphs();
        return DAWN_OK;
    }
    if (std::string(name) == "graph_max_tokens") {
        if (value < 0 || value > 65536) return fail(DAWN_ERR_INVALID_ARG, "graph_max_tokens out of range");
        e->graph_max_tokens = (int)value;
        return DAWN_OK;
    }
    return fail(DAWN_ERR_INVALID_ARG, "unknown option %s", name);
}

static int embedder_forward_device_impl(dawn_embedder* e, const uint32_t* d_token_ids, const int32_t* d_seq_offsets, int B,
                                 int total_tokens, int max_len, float* d_out, void* stream);
int dawn_embedder_forward_device(dawn_embedder* e, const uint32_t* d_token_ids, const int32_t* d_seq_offsets, int B,
                                 int total_tokens, int max_len, float* d_out, void* stream) {
    return dawn::guarded([&] { return embedder_forward_device_impl(e, d_token_ids, d_seq_offsets, B, total_tokens, max_len, d_out, stream); });
}
static int embedder_forward_device_impl(dawn_embedder* e, const uint32_t* d_token_ids, const int32_t* d_seq_offsets, int B,
                                 int total_tokens, int max_len, float* d_out, void* stream) {
    if (!e || !d_token_ids || !d_seq_offsets || !d_out) return fail(DAWN_ERR_INVALID_ARG, "NULL argument");
    if (B <= 0 || total_tokens <= 0) return DAWN_OK;
    if (max_len < 1 || max_len > e->cfg.max_position_embeddings) return fail(DAWN_ERR_INVALID_ARG, "max_len out of range");
    DAWN_HIP_TRY(hipSetDevice(e->device));
    DAWN_TRY(ensure_ws(e, total_tokens, B));
    hipStream_t s = (hipStream_t)stream;
    if (e->use_graphs && total_tokens <= e->graph_max_tokens) {
        const dawn_embedder::GraphKey key{B, total_tokens, max_len, d_token_ids, d_seq_offsets, d_out};
        auto it = e->graphs.find(key);
        if (it != e->graphs.end()) {
            DAWN_HIP_TRY(hipGraphLaunch(it->second, s));
            return DAWN_OK;
        }
        if (++e->graph_seen[key] >= 2 && e->graphs.size() < 512) {
            // second sighting (the first, plain, run did every one-time initialisation): capture, instantiate, replay
            hipGraph_t g = nullptr;
            hipGraphExec_t ge = nullptr;
            if (hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal) == hipSuccess) {
                if (!encoder_forward(e, d_token_ids, d_seq_offsets, B, total_tokens, max_len, s, d_out))
                    dawn::launch_pool_norm(e->x, d_seq_offsets, B, d_out, s);
                const hipError_t ce = hipStreamEndCapture(s, &g);
                if (ce == hipSuccess && g && hipGraphInstantiate(&ge, g, nullptr, nullptr, 0) == hipSuccess) {
                    (void)hipGraphDestroy(g);
                    e->graphs[key] = ge;
                    DAWN_HIP_TRY(hipGraphLaunch(ge, s));
                    return DAWN_OK;
                }
                if (g) (void)hipGraphDestroy(g);
            }
            (void)hipGetLastError();
            e->use_graphs = 0;  // capture is not available here: plain launches from now on
        }
    }
    if (!encoder_forward(e, d_token_ids, d_seq_offsets, B, total_tokens, max_len, s, d_out))
        dawn::launch_pool_norm(e->x, d_seq_offsets, B, d_out, s);  // embedding_service.rs:126-136
    DAWN_HIP_TRY(hipGetLastError());
    return DAWN_OK;
}

static int embedder_forward_impl(dawn_embedder* e, const uint32_t* token_ids, const int32_t* seq_offsets, int B, float* out);
int dawn_embedder_forward(dawn_embedder* e, const uint32_t* token_ids, const int32_t* seq_offsets, int B, float* out) {
    return dawn::guarded([&] { return embedder_forward_impl(e, token_ids, seq_offsets, B, out); });
}
static int embedder_forward_impl(dawn_embedder* e, const uint32_t* token_ids, const int32_t* seq_offsets, int B, float* out) {
    if (!e || !token_ids || !seq_offsets || !out) return fail(DAWN_ERR_INVALID_ARG, "NULL argument");
    if (B <= 0) return DAWN_OK;
    int T = 0, max_len = 0;
    DAWN_TRY(check_sequences(e, token_ids, seq_offsets, B, &T, &max_len));
    DAWN_HIP_TRY(hipSetDevice(e->device));
    if (e->host_io) {
        DAWN_TRY(ensure_ws(e, T, B));
        DAWN_TRY(ensure_host_io(e, T, B));
        const size_t ro = ((size_t)B + 1 + 3) & ~(size_t)3;
        std::memcpy(e->h_in, seq_offsets, ((size_t)B + 1) * 4);
        std::memcpy(e->h_in + ro, token_ids, (size_t)T * 4);
        DAWN_HIP_TRY(hipMemcpyAsync(e->d_in, e->h_in, (ro + (size_t)T) * 4, hipMemcpyHostToDevice, e->stream));
        DAWN_TRY(dawn_embedder_forward_device(e, reinterpret_cast<const uint32_t*>(e->d_in + ro), e->d_in, B, T, max_len, e->h_out, e->stream));
        DAWN_HIP_TRY(hipStreamSynchronize(e->stream));
        std::memcpy(out, e->h_out, (size_t)B * e->cfg.hidden_size * 4);
        return DAWN_OK;
    }
    DAWN_TRY(upload_inputs(e, token_ids, seq_offsets, B, T));
    DAWN_TRY(dawn_embedder_forward_device(e, e->d_ids, e->d_off, B, T, max_len, e->d_out, e->stream));
    DAWN_HIP_TRY(hipMemcpyAsync(out, e->d_out, (size_t)B * e->cfg.hidden_size * 4, hipMemcpyDeviceToHost, e->stream));
    DAWN_HIP_TRY(hipStreamSynchronize(e->stream));
    return DAWN_OK;
}

static int embedder_hidden_states_impl(dawn_embedder* e, const uint32_t* token_ids, const int32_t* seq_offsets, int B,
                                float* out);
int dawn_embedder_hidden_states(dawn_embedder* e, const uint32_t* token_ids, const int32_t* seq_offsets, int B,
                                float* out) {
    return dawn::guarded([&] { return embedder_hidden_states_impl(e, token_ids, seq_offsets, B, out); });
}
static int embedder_hidden_states_impl(dawn_embedder* e, const uint32_t* token_ids, const int32_t* seq_offsets, int B,
                                float* out) {
    if (!e || !token_ids || !seq_offsets || !out) return fail(DAWN_ERR_INVALID_ARG, "NULL argument");
    if (B <= 0) return DAWN_OK;
    int T = 0, max_len = 0;
    DAWN_TRY(check_sequences(e, token_ids, seq_offsets, B, &T, &max_len));
    DAWN_HIP_TRY(hipSetDevice(e->device));
    DAWN_TRY(upload_inputs(e, token_ids, seq_offsets, B, T));
    encoder_forward(e, e->d_ids, e->d_off, B, T, max_len, e->stream);
    DAWN_HIP_TRY(hipGetLastError());
    DAWN_HIP_TRY(hipMemcpyAsync(out, e->x, (size_t)T * e->cfg.hidden_size * 4, hipMemcpyDeviceToHost, e->stream));
    DAWN_HIP_TRY(hipStreamSynchronize(e->stream));
    return DAWN_OK;
}

// Test hook: ONE kernel of the forward in isolation, so that a3 / a5 / a7 of SURVEY 8(a) have their own parity tests:
//   op 0  BertEmbeddings (model.rs:266-281): token_ids [T] (one sequence, positions 0..T-1) -> out [T][384]
//   op 1  LayerNorm(a + r) with layer 0's attention-output LayerNorm (model.rs:86-104,378): in = a [T][384] | r [T][384]
//   op 2  layer 0's intermediate dense + activation (model.rs:425-430, HiddenActLayer :28-37): in [T][384] -> [T][1536]
//   op 3  the same GEMM through the 64x64 tile kernel regardless of T (op 2 takes the skinny form for T <= 640)
static int debug_op_impl(dawn_embedder* e, int op, const void* in, int T, float* out);
int dawn_embedder_debug_op(dawn_embedder* e, int op, const void* in, int T, float* out) {
    return dawn::guarded([&] { return debug_op_impl(e, op, in, T, out); });
}
static int debug_op_impl(dawn_embedder* e, int op, const void* in, int T, float* out) {
    if (!e || !in || !out || T <= 0) return fail(DAWN_ERR_INVALID_ARG, "bad argument");
    const Config& c = e->cfg;
    if (op == 0 && T > c.max_position_embeddings) return fail(DAWN_ERR_INVALID_ARG, "T exceeds max_position_embeddings");
    if (T > 65536) return fail(DAWN_ERR_INVALID_ARG, "T too large");
    DAWN_HIP_TRY(hipSetDevice(e->device));
    DAWN_TRY(ensure_ws(e, 2 * T, 1));
    hipStream_t s = e->stream;
    const size_t H = c.hidden_size, I = c.intermediate_size;
    const LayerW& L = e->layers[0];
    const float eps = (float)c.layer_norm_eps;
    size_t out_elems = (size_t)T * H;
    if (op == 0) {
        const int off[2] = {0, T};
        DAWN_HIP_TRY(hipMemcpyAsync(e->d_ids, in, (size_t)T * 4, hipMemcpyHostToDevice, s));
        DAWN_HIP_TRY(hipMemcpyAsync(e->d_off, off, sizeof(off), hipMemcpyHostToDevice, s));
        dawn::launch_tok_pos(e->d_off, 1, e->d_pos, s);
        dawn::launch_embed_ln(e->d_ids, e->d_pos, T, e->word, e->pos, e->type0, e->emb_g, e->emb_b, eps, e->x, s);
    } else if (op == 1) {
        DAWN_HIP_TRY(hipMemcpyAsync(e->tmp, in, (size_t)T * H * 4, hipMemcpyHostToDevice, s));
        DAWN_HIP_TRY(hipMemcpyAsync(e->attn, (const float*)in + (size_t)T * H, (size_t)T * H * 4, hipMemcpyHostToDevice, s));
        dawn::launch_add_ln(e->tmp, e->attn, T, L.ao_g, L.ao_beta, eps, e->x, s);
    } else if (op == 2 || op == 3) {
        DAWN_HIP_TRY(hipMemcpyAsync(e->attn, in, (size_t)T * H * 4, hipMemcpyHostToDevice, s));
        dawn::launch_gemm_nt(e->attn, L.i_w, L.i_b, e->ff, T, (int)I, (int)H, c.act, s, op == 3, e->skinny_max_m);
        out_elems = (size_t)T * I;
    } else if (op == 4 || op == 5) {
        // 4: the same dense layer through the bf16x3 kernel (embed_gemm3.hip): planes made here from the f32 input / weights
        // 5: ... its planes OUTPUT, re-assembled (p1 + p2 + p3) on the host side of this hook
        DevBuf b_ap, b_wp, b_yp;  // (freed on every return path)
        DAWN_HIP_TRY(b_ap.alloc((size_t)3 * T * H * 2));
        DAWN_HIP_TRY(b_wp.alloc((size_t)3 * I * H * 2));
        DAWN_HIP_TRY(b_yp.alloc((size_t)3 * T * I * 2));
        uint16_t *ap = (uint16_t*)b_ap.p, *wp = (uint16_t*)b_wp.p, *yp = (uint16_t*)b_yp.p;
        DAWN_HIP_TRY(hipMemcpyAsync(e->attn, in, (size_t)T * H * 4, hipMemcpyHostToDevice, s));
        dawn::launch_split_planes(e->attn, ap, T, (int)H, (size_t)T, s);
        dawn::launch_split_planes(L.i_w, wp, (int)I, (int)H, (size_t)I, s);
        dawn::launch_gemm_bf16x3(ap, (size_t)T * H, wp, (size_t)I * H, L.i_b, e->ff, yp, (size_t)T * I, T, (int)I, (int)H, c.act, s, e->g3);
        out_elems = (size_t)T * I;
        if (op == 5) {
            std::vector<uint16_t> hp((size_t)3 * T * I);
            DAWN_HIP_TRY(hipMemcpyAsync(hp.data(), yp, hp.size() * 2, hipMemcpyDeviceToHost, s));
            DAWN_HIP_TRY(hipStreamSynchronize(s));
            for (size_t i = 0; i < out_elems; ++i) {
                auto f = [&](size_t j) { uint32_t b = (uint32_t)hp[j] << 16; float v; std::memcpy(&v, &b, 4); return v; };
                const size_t o = dawn::plane_index(i / I, (int)(i % I), (size_t)T);  // K-blocked planes
                out[i] = (f(o) + f(out_elems + o)) + f(2 * out_elems + o);
            }
        }
        DAWN_HIP_TRY(hipStreamSynchronize(s));
        if (op == 5) return DAWN_OK;
        DAWN_HIP_TRY(hipMemcpy(out, e->ff, out_elems * 4, hipMemcpyDeviceToHost));
        return DAWN_OK;
    } else {
        return fail(DAWN_ERR_INVALID_ARG, "unknown op %d", op);
    }
    DAWN_HIP_TRY(hipGetLastError());
    DAWN_HIP_TRY(hipMemcpyAsync(out, op >= 2 ? e->ff : e->x, out_elems * 4, hipMemcpyDeviceToHost, s));
    DAWN_HIP_TRY(hipStreamSynchronize(s));
    return DAWN_OK;
}

// Timing hook: mean ms of one dense layer shape [T x K] . [N x K]^T over `iters` launches: variant 0 = f32 MFMA tile kernel,
// 1 = bf16x3 kernel (planes prepared outside the timed region), 2 = ... writing planes instead of f32, 3 = ... with GELU.  Layer-0 weights are reused for every shape (K, N) in
// {(384, 1152), (384, 384), (384, 1536), (1536, 384)}.
static int debug_gemm_time_impl(dawn_embedder* e, int T, int N, int K, int variant, int iters, double* mean_ms);
int dawn_embedder_debug_gemm_time(dawn_embedder* e, int T, int N, int K, int variant, int iters, double* mean_ms) {
    return dawn::guarded([&] { return debug_gemm_time_impl(e, T, N, K, variant, iters, mean_ms); });
}
static int debug_gemm_time_impl(dawn_embedder* e, int T, int N, int K, int variant, int iters, double* mean_ms) {
    if (!e || !mean_ms || T <= 0 || T > (1 << 20) || iters <= 0) return fail(DAWN_ERR_INVALID_ARG, "bad argument");
    const Config& c = e->cfg;
    const int H = c.hidden_size, I = c.intermediate_size;
    const LayerW& L = e->layers[0];
    const float *W = nullptr, *bias = nullptr;
    if (K == H && N == 3 * H) W = L.qkv_w, bias = L.qkv_b;
    else if (K == H && N == H) W = L.ao_w, bias = L.ao_b;
    else if (K == H && N == I) W = L.i_w, bias = L.i_b;
    else if (K == I && N == H) W = L.o_w, bias = L.o_b;
    else return fail(DAWN_ERR_INVALID_ARG, "shape not in the model");
    DAWN_HIP_TRY(hipSetDevice(e->device));
    hipStream_t s = e->stream;
    DevBuf b_a, b_y, b_ap, b_wp, b_yp;  // (freed on every return path)
    DAWN_HIP_TRY(b_a.alloc((size_t)T * K * 4));
    DAWN_HIP_TRY(b_y.alloc((size_t)T * N * 4));
    DAWN_HIP_TRY(b_ap.alloc((size_t)3 * T * K * 2));
    DAWN_HIP_TRY(b_wp.alloc((size_t)3 * N * K * 2));
    float *a = (float*)b_a.p, *y = (float*)b_y.p;
    uint16_t *ap = (uint16_t*)b_ap.p, *wp = (uint16_t*)b_wp.p;
    // activations: the model's own (random) weights, repeated — zero operands would let the chip clock higher than real data
    for (size_t off = 0, n = (size_t)T * K; off < n;) {
        const size_t take = std::min(n - off, (size_t)3 * H * H);
        DAWN_HIP_TRY(hipMemcpyAsync(a + off, L.qkv_w, take * 4, hipMemcpyDeviceToDevice, s));
        off += take;
    }
    if (variant >= 2) DAWN_HIP_TRY(b_yp.alloc((size_t)3 * T * N * 2));
    uint16_t* yp = (uint16_t*)b_yp.p;
    dawn::launch_split_planes(a, ap, T, K, (size_t)T, s);
    dawn::launch_split_planes(W, wp, N, K, (size_t)N, s);
    DevEvent ev0, ev1;
    DAWN_HIP_TRY(ev0.create());
    DAWN_HIP_TRY(ev1.create());
    const hipEvent_t e0 = ev0.e, e1 = ev1.e;
    for (int it = -2; it < iters; ++it) {
        if (it == 0) DAWN_HIP_TRY(hipEventRecord(e0, s));
        if (variant == 0) dawn::launch_gemm_nt(a, W, bias, y, T, N, K, 0, s, true);
        else if (variant == 1) dawn::launch_gemm_bf16x3(ap, (size_t)T * K, wp, (size_t)N * K, bias, y, nullptr, 0, T, N, K, 0, s, e->g3);
        else dawn::launch_gemm_bf16x3(ap, (size_t)T * K, wp, (size_t)N * K, bias, nullptr, yp, (size_t)T * N, T, N, K, variant == 3 ? 1 : 0, s, e->g3);
    }
    DAWN_HIP_TRY(hipEventRecord(e1, s));
    DAWN_HIP_TRY(hipStreamSynchronize(s));
    float ms = 0.f;
    DAWN_HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
    *mean_ms = ms / iters;
    return DAWN_OK;
}

}  // extern "C"
