// model_files.hpp — the on-disk side of EmbeddingProvider::new (src/embedding/embedding_service.rs:55-95) as pure host code:
// config.json (model.rs:115-133) and model.safetensors (safetensors 0.3.1 layout: u64 header length, JSON header, raw
// little-endian data) are read, validated and resolved into ONE block of f32 weights in upload order.  No HIP here: the
// same code is compiled into the CPU sanitizer build of tests/native and fuzzed on the CPU (tests/test_fuzz_cpu.py).
#pragma once
#include <cstddef>
#include <string>
#include <vector>

namespace dawn {

struct BertConfig {  // model.rs:115-133 ; defaults = Config::_all_mini_lm_l6_v2 (:160-180)
    int vocab_size = 30522, hidden_size = 384, num_hidden_layers = 6, num_attention_heads = 12;
    int intermediate_size = 1536, max_position_embeddings = 512, type_vocab_size = 2;
    int act = 1;  // 1 = gelu (tanh form), 2 = relu
    double layer_norm_eps = 1e-12;
    std::string model_type = "bert";
};

struct LayerOffsets {  // element offsets into ModelHost::weights
    size_t qw, kw, vw, qb, kb, vb, aow, aob, aog, aobeta, iw, ib, ow, ob, og, obeta;
};

struct ModelHost {
    BertConfig cfg;
    std::vector<float> weights;
    size_t o_word = 0, o_pos = 0, o_type = 0, o_eg = 0, o_eb = 0;
    std::vector<LayerOffsets> layers;
};

// DAWN_OK, or DAWN_ERR_IO (unreadable / malformed / missing tensors) / DAWN_ERR_UNSUPPORTED (a model the kernels are not
// built for) with dawn_last_error() set.  May throw std::bad_alloc (callers sit inside guarded()).
int load_model_files(const char* safetensors_path, const char* config_json_path, ModelHost& out);

}  // namespace dawn
