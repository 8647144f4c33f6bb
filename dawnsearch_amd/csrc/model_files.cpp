// model_files.cpp — see model_files.hpp.  Tensor names as BertModel::load asks for them (src/embedding/model.rs:235-255,
// 301-303,359-363,417,443-447,510,538-546 incl. the "bert." prefix retry :543-547 and the LayerNorm gamma/beta fallback
// :210-222).  Every number that comes out of a file is range-checked before it is cast or sizes anything.
#include "model_files.hpp"

#include <cstdint>
#include <cstring>
#include <map>

#include "host_common.hpp"
#include "mini_json.hpp"

namespace dawn {

namespace {

struct TensorRef {
    std::vector<int64_t> shape;
    size_t begin = 0, end = 0;
};

constexpr size_t kMaxConfigBytes = (size_t)1 << 20;
constexpr uint64_t kMaxHeaderBytes = (uint64_t)100 << 20;  // the safetensors crate's own limit

int parse_config(const char* path, BertConfig& cfg) {
    std::vector<char> txt;
    if (!read_file(path, txt, kMaxConfigBytes)) return fail(DAWN_ERR_IO, "cannot read %s", path);
    JParser jp{txt.data(), txt.data() + txt.size()};
    JVal j = jp.parse();
    if (!jp.ok || j.kind != JVal::Obj) return fail(DAWN_ERR_IO, "%s: invalid JSON", path);
    bool bad = false;
    auto geti = [&](const char* k, int& dst) {
        const JVal* v = j.get(k);
        if (!v) return;
        int64_t i;
        if (jint(v, -(1 << 30), 1 << 30, i)) dst = (int)i;
        else bad = true;
    };
    geti("vocab_size", cfg.vocab_size);
    geti("hidden_size", cfg.hidden_size);
    geti("num_hidden_layers", cfg.num_hidden_layers);
    geti("num_attention_heads", cfg.num_attention_heads);
    geti("intermediate_size", cfg.intermediate_size);
    geti("max_position_embeddings", cfg.max_position_embeddings);
    geti("type_vocab_size", cfg.type_vocab_size);
    if (bad) return fail(DAWN_ERR_IO, "%s: a size field is not an integer", path);
    if (const JVal* v = j.get("layer_norm_eps"))
        if (v->kind == JVal::Num) cfg.layer_norm_eps = v->num;
    if (const JVal* v = j.get("model_type"))
        if (v->kind == JVal::Str) cfg.model_type = v->str;
    if (const JVal* v = j.get("hidden_act")) {  // enum HiddenAct { Gelu, Relu } model.rs:10-15
        if (v->kind == JVal::Str && v->str == "gelu") cfg.act = 1;
        else if (v->kind == JVal::Str && v->str == "relu") cfg.act = 2;
        else return fail(DAWN_ERR_UNSUPPORTED, "hidden_act must be \"gelu\" or \"relu\"");
    }
    return DAWN_OK;
}

}  // namespace

int load_model_files(const char* safetensors_path, const char* config_json_path, ModelHost& out) {
    BertConfig cfg;
    if (config_json_path) DAWN_TRY(parse_config(config_json_path, cfg));
    if (cfg.hidden_size != 384 || cfg.num_attention_heads != 12 || cfg.intermediate_size % 64 != 0 ||
        cfg.num_hidden_layers < 1 || cfg.num_hidden_layers > 48 || cfg.max_position_embeddings > 512)
        return fail(DAWN_ERR_UNSUPPORTED,
                    "kernels are built for hidden 384 / 12 heads / intermediate %%64 / <=512 positions (all-MiniLM-L6-v2)");
    // every size below comes from a file: bound it before it sizes an allocation (a C ABI must not throw / terminate)
    if (cfg.vocab_size < 1 || cfg.vocab_size > (1 << 22) || cfg.type_vocab_size < 1 || cfg.type_vocab_size > 1024 ||
        cfg.intermediate_size < 64 || cfg.intermediate_size > 65536 || cfg.max_position_embeddings < 1 ||
        !(cfg.layer_norm_eps >= 0.0) || cfg.layer_norm_eps > 1.0 || cfg.model_type.size() > 64)
        return fail(DAWN_ERR_UNSUPPORTED, "config.json: vocab_size / type_vocab_size / intermediate_size / "
                                          "max_position_embeddings / layer_norm_eps out of range");

    std::vector<char> file;
    if (!read_file(safetensors_path, file) || file.size() < 8) return fail(DAWN_ERR_IO, "cannot read %s", safetensors_path);
    uint64_t hlen = 0;
    std::memcpy(&hlen, file.data(), 8);
    if (hlen > file.size() - 8 || hlen > kMaxHeaderBytes)
        return fail(DAWN_ERR_IO, "%s: bad safetensors header length", safetensors_path);
    JParser jp{file.data() + 8, file.data() + 8 + hlen};
    JVal hdr = jp.parse();
    if (!jp.ok || hdr.kind != JVal::Obj) return fail(DAWN_ERR_IO, "%s: invalid safetensors header", safetensors_path);
    const char* data = file.data() + 8 + hlen;
    const size_t data_len = file.size() - 8 - hlen;
    std::map<std::string, TensorRef> tensors;
    for (auto& kv : hdr.obj) {
        if (kv.first == "__metadata__") continue;
        const JVal* dt = kv.second.get("dtype");
        const JVal* sh = kv.second.get("shape");
        const JVal* off = kv.second.get("data_offsets");
        if (!dt || !sh || !off || dt->kind != JVal::Str || sh->kind != JVal::Arr || off->kind != JVal::Arr ||
            off->arr.size() != 2 || sh->arr.size() > 8)
            return fail(DAWN_ERR_IO, "tensor %.200s: malformed entry", kv.first.c_str());
        if (dt->str != "F32") continue;  // DTYPE = F32 (model.rs:8); other dtypes are not used by this model
        TensorRef t;
        for (auto& d : sh->arr) {
            int64_t v;
            if (!jint(&d, 0, (int64_t)1 << 31, v)) return fail(DAWN_ERR_IO, "tensor %.200s: bad shape", kv.first.c_str());
            t.shape.push_back(v);
        }
        int64_t b, e;
        if (!jint(&off->arr[0], 0, (int64_t)1 << 52, b) || !jint(&off->arr[1], 0, (int64_t)1 << 52, e) || b > e ||
            (uint64_t)e > data_len)
            return fail(DAWN_ERR_IO, "tensor %.200s: data out of range", kv.first.c_str());
        t.begin = (size_t)b;
        t.end = (size_t)e;
        tensors[kv.first] = t;
    }

    // name resolution: plain, then "{model_type}." prefix (model.rs:538-556)
    std::string prefix;
    auto has = [&](const std::string& n) { return tensors.count(n) != 0; };
    if (!has("embeddings.word_embeddings.weight")) {
        prefix = cfg.model_type + ".";
        if (!has(prefix + "embeddings.word_embeddings.weight"))
            return fail(DAWN_ERR_IO, "cannot find tensor embeddings.word_embeddings.weight (also tried prefix %s)", prefix.c_str());
    }
    const int H = cfg.hidden_size, I = cfg.intermediate_size, NL = cfg.num_hidden_layers;
    size_t total = (size_t)cfg.vocab_size * H + (size_t)cfg.max_position_embeddings * H + (size_t)cfg.type_vocab_size * H + 2 * H;
    total += (size_t)NL * ((size_t)3 * H * H + 3 * H + (size_t)H * H + H + 2 * H + (size_t)I * H + I + (size_t)H * I + H + 2 * H);
    // resolve first, allocate after: a file is only worth `total` floats of host memory once every tensor is in it
    struct Copy {
        size_t at, begin, bytes;
    };
    std::vector<Copy> copies;
    size_t cur = 0;
    std::string err;
    auto take = [&](const std::string& name, std::vector<int64_t> shape, const std::string& alt = "") -> size_t {
        std::string full = prefix + name;
        auto it = tensors.find(full);
        if (it == tensors.end() && !alt.empty()) it = tensors.find(prefix + alt);
        size_t n = 1;
        for (auto d : shape) n *= (size_t)d;
        const size_t at = cur;
        cur += n;
        if (it == tensors.end()) {
            if (err.empty()) err = "cannot find tensor " + full;
            return at;
        }
        if (it->second.shape != shape || it->second.end - it->second.begin != n * 4) {
            if (err.empty()) err = "shape mismatch for tensor " + full;
            return at;
        }
        copies.push_back({at, it->second.begin, n * 4});
        return at;
    };
    auto ln = [&](const std::string& base, size_t& g, size_t& b) {  // weight/bias, fallback gamma/beta (:210-222)
        g = take(base + ".weight", {H}, base + ".gamma");
        b = take(base + ".bias", {H}, base + ".beta");
    };
    out.o_word = take("embeddings.word_embeddings.weight", {cfg.vocab_size, H});
    out.o_pos = take("embeddings.position_embeddings.weight", {cfg.max_position_embeddings, H});
    out.o_type = take("embeddings.token_type_embeddings.weight", {cfg.type_vocab_size, H});
    ln("embeddings.LayerNorm", out.o_eg, out.o_eb);
    std::vector<LayerOffsets>& lo = out.layers;
    lo.assign(NL, LayerOffsets{});
    for (int L = 0; L < NL; ++L) {
        const std::string p = "encoder.layer." + std::to_string(L) + ".";
        // Q|K|V weights and biases are laid out back to back so one GEMM produces [T][1152]
        lo[L].qw = take(p + "attention.self.query.weight", {H, H});
        lo[L].kw = take(p + "attention.self.key.weight", {H, H});
        lo[L].vw = take(p + "attention.self.value.weight", {H, H});
        lo[L].qb = take(p + "attention.self.query.bias", {H});
        lo[L].kb = take(p + "attention.self.key.bias", {H});
        lo[L].vb = take(p + "attention.self.value.bias", {H});
        lo[L].aow = take(p + "attention.output.dense.weight", {H, H});
        lo[L].aob = take(p + "attention.output.dense.bias", {H});
        ln(p + "attention.output.LayerNorm", lo[L].aog, lo[L].aobeta);
        lo[L].iw = take(p + "intermediate.dense.weight", {I, H});
        lo[L].ib = take(p + "intermediate.dense.bias", {I});
        lo[L].ow = take(p + "output.dense.weight", {H, I});
        lo[L].ob = take(p + "output.dense.bias", {H});
        ln(p + "output.LayerNorm", lo[L].og, lo[L].obeta);
    }
    if (!err.empty()) return fail(DAWN_ERR_IO, "%s: %.300s", safetensors_path, err.c_str());
    if (cur != total) return fail(DAWN_ERR_IO, "%s: internal size mismatch", safetensors_path);
    out.weights.assign(total, 0.0f);
    for (const Copy& c : copies) std::memcpy(out.weights.data() + c.at, data + c.begin, c.bytes);
    out.cfg = cfg;
    return DAWN_OK;
}

}  // namespace dawn

// Host-only check of the model files dawn_embedder_create would load: same parsing, same errors, no device needed
// (a deployment can validate its files before it claims a GPU; the CPU fuzz test drives the parsers through it).
extern "C" int dawn_embedder_check_files(const char* safetensors_path, const char* config_json_path) {
    if (!safetensors_path) return dawn::fail(DAWN_ERR_INVALID_ARG, "NULL argument");
    const int rc = dawn::guarded([&] {
        dawn::ModelHost m;
        return dawn::load_model_files(safetensors_path, config_json_path, m);
    });
    return rc == DAWN_ERR_INVALID_ARG ? DAWN_ERR_IO : rc;
}
