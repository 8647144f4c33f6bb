// fuzz_host.cpp — driver of the CPU sanitizer / fuzz test (tests/test_fuzz_cpu.py).  Test infrastructure, not product.
//
// Built by `make -C tests/native asan` together with the pure-host translation units of the library (tokenizer.cpp,
// model_files.cpp, host_helpers.cpp — which include mini_json.hpp) and the C oracle, all under
// -fsanitize=address,undefined -fno-sanitize-recover.  It walks a directory of (mutated) input files and pushes each one
// through the C-ABI entry point that would read it in production; whatever the file holds, the call has to come back with
// a return code.  Any sanitizer report aborts the process, which the Python side sees as a non-zero exit.
//
//   fuzz_host <dir>      tok_*.json / tok_*.txt  -> dawn_tokenizer_create (+ encode of the texts in <dir>/texts.bin)
//                        cfg_*.json              -> dawn_embedder_check_files(<dir>/good.safetensors, file)
//                        st_*.safetensors        -> dawn_embedder_check_files(file, NULL) and (file, <dir>/good_config.json)
//                        texts.bin               -> NUL-separated byte strings, encoded with every tokenizer that loaded
//                        helpers.bin             -> random bytes through dawn_vec_*, dawn_best_*, dawn_topk_merge_host and
//                                                   the oracle's counterparts (same answers expected)
// Prints one line per file: "<name> rc=<code>"; exit status 0 unless an invariant below is violated.
#include <dirent.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/dawn_hip.h"
#include "../../oracle/dawn_oracle.h"

extern "C" int dawn_embedder_check_files(const char* safetensors_path, const char* config_json_path);

static std::vector<char> slurp(const std::string& p) {
    std::vector<char> out;
    if (FILE* f = std::fopen(p.c_str(), "rb")) {
        char buf[65536];
        size_t n;
        while ((n = std::fread(buf, 1, sizeof(buf), f)) > 0) out.insert(out.end(), buf, buf + n);
        std::fclose(f);
    }
    return out;
}

static bool starts(const std::string& s, const char* pre) { return s.rfind(pre, 0) == 0; }
static bool ends(const std::string& s, const char* suf) {
    const size_t n = std::strlen(suf);
    return s.size() >= n && s.compare(s.size() - n, n, suf) == 0;
}

static int failures = 0;
#define CHECK(cond)                                                            \
    do {                                                                       \
        if (!(cond)) {                                                         \
            std::fprintf(stderr, "CHECK failed %s:%d: %s\n", __FILE__, __LINE__, #cond); \
            ++failures;                                                        \
        }                                                                      \
    } while (0)

static void run_texts(dawn_tokenizer* t, const std::vector<std::string>& texts) {
    std::vector<uint32_t> ids(1 << 16);
    for (const std::string& s : texts) {
        size_t n = 0;
        const int rc = dawn_tokenizer_encode(t, s.c_str(), ids.data(), ids.size(), &n);
        CHECK(rc == DAWN_OK || rc == DAWN_ERR_INVALID_ARG);
        if (rc == DAWN_OK) CHECK(n >= 2 && n <= ids.size());
        size_t n2 = 0;  // too small a buffer: an error, never an overrun
        (void)dawn_tokenizer_encode(t, s.c_str(), ids.data(), 1, &n2);
    }
    std::vector<const char*> ptrs;
    for (const std::string& s : texts) ptrs.push_back(s.c_str());
    std::vector<int32_t> off(texts.size() + 1);
    const int rc = dawn_tokenizer_encode_batch(t, ptrs.data(), ptrs.size(), ids.data(), ids.size(), off.data());
    CHECK(rc == DAWN_OK || rc == DAWN_ERR_INVALID_ARG);
    if (rc == DAWN_OK)
        for (size_t b = 0; b < texts.size(); ++b) CHECK(off[b + 1] - off[b] >= 2);
}

// random bytes through the vector / BestResults / merge helpers, against the oracle's restatements
static void run_helpers(const std::vector<char>& bytes) {
    if (bytes.size() < 384 * 4) return;
    const size_t nvec = bytes.size() / (384 * 4);
    for (size_t v = 0; v < nvec; ++v) {
        float x[384], y[384], z[384];
        std::memcpy(x, bytes.data() + v * 384 * 4, sizeof(x));
        CHECK(dawn_vec_is_normalized(x) == orc_is_normalized(x));
        uint8_t w1[1152], w2[1152];
        dawn_vec_to24(x, w1);
        orc_to24(x, w2);
        CHECK(std::memcmp(w1, w2, sizeof(w1)) == 0);
        const int r1 = dawn_vec_from24((const uint8_t*)bytes.data() + v * 1152, y);
        const int r2 = orc_from24((const uint8_t*)bytes.data() + v * 1152, z);
        CHECK((r1 == DAWN_OK) == (r2 == 0));
        CHECK(std::memcmp(y, z, sizeof(y)) == 0);
        std::memcpy(y, x, sizeof(x));
        std::memcpy(z, x, sizeof(x));
        dawn_vec_normalize(y, 384);
        orc_normalize(z, 384);
        CHECK(std::memcmp(y, z, sizeof(y)) == 0);
    }
    // BestResults: a stream of (id, distance) pairs cut from the bytes, sizes 0..20
    const size_t pairs = bytes.size() / 8;
    for (size_t size : {(size_t)0, (size_t)1, (size_t)3, (size_t)20}) {
        dawn_best_results* b = nullptr;
        CHECK(dawn_best_new(size, &b) == DAWN_OK && b);
        orc_best_results* o = size ? orc_best_new(size) : nullptr;
        for (size_t i = 0; i < std::min<size_t>(pairs, 400); ++i) {
            uint32_t id;
            float d;
            std::memcpy(&id, bytes.data() + i * 8, 4);
            std::memcpy(&d, bytes.data() + i * 8 + 4, 4);
            if (d != d) d = 0.5f;  // (NaN ordering is not specified by either side)
            const int r = dawn_best_insert(b, id % 64, d);
            if (o) CHECK(r == orc_best_insert(o, id % 64, d));
            else CHECK(r == 0);
        }
        dawn_best_sort(b);
        if (o) {
            orc_best_sort(o);
            CHECK(dawn_best_len(b) == o->len);
            for (size_t i = 0; i < o->len; ++i) {
                size_t id;
                float d;
                CHECK(dawn_best_get(b, i, &id, &d) == DAWN_OK);
                CHECK(id == o->results[i].id && std::memcmp(&d, &o->results[i].distance, 4) == 0);
            }
            orc_best_free(o);
        }
        CHECK(dawn_best_get(b, 1000, nullptr, nullptr) == DAWN_ERR_INVALID_ARG);
        dawn_best_free(b);
    }
    CHECK(dawn_best_insert(nullptr, 0, 0.f) < 0);
    dawn_best_results* huge = nullptr;
    const int rh = dawn_best_new((size_t)-1, &huge);  // must not throw through the ABI
    CHECK(rh == DAWN_OK || rh == DAWN_ERR_OOM);
    dawn_best_free(huge);
    // host merge of G sorted lists
    const size_t G = 3, B = 2, k = 5;
    std::vector<uint64_t> il(G * B * k), ol(B * k);
    std::vector<float> id(G * B * k), od(B * k);
    std::vector<uint32_t> ifd(G * B), ofd(B);
    for (size_t i = 0; i < il.size(); ++i) il[i] = i;
    for (size_t g = 0; g < G * B; ++g) {
        ifd[g] = (uint8_t)bytes[g] % (k + 1);
        for (size_t i = 0; i < k; ++i) id[g * k + i] = (float)((uint8_t)bytes[64 + g * k + i] % 7) + (float)i * 8.f;
    }
    CHECK(dawn_topk_merge_host(G, B, k, il.data(), id.data(), ifd.data(), ol.data(), od.data(), ofd.data()) == DAWN_OK);
    for (size_t b = 0; b < B; ++b)
        for (uint32_t i = 1; i < ofd[b]; ++i) CHECK(od[b * k + i - 1] <= od[b * k + i]);
}

int main(int argc, char** argv) {
    if (argc < 2) {
        std::fprintf(stderr, "usage: fuzz_host <dir>\n");
        return 2;
    }
    const std::string dir = argv[1];
    std::vector<std::string> names;
    if (DIR* d = opendir(dir.c_str())) {
        while (dirent* e = readdir(d))
            if (e->d_name[0] != '.') names.emplace_back(e->d_name);
        closedir(d);
    }
    std::sort(names.begin(), names.end());
    std::vector<std::string> texts;
    {
        const std::vector<char> tb = slurp(dir + "/texts.bin");
        size_t i = 0;
        while (i < tb.size()) {
            size_t e = i;
            while (e < tb.size() && tb[e]) ++e;
            texts.emplace_back(tb.data() + i, e - i);
            i = e + 1;
        }
    }
    const std::string good_st = dir + "/good.safetensors", good_cfg = dir + "/good_config.json";
    for (const std::string& n : names) {
        const std::string path = dir + "/" + n;
        int rc = 1;
        if (starts(n, "tok_")) {
            dawn_tokenizer* t = nullptr;
            rc = dawn_tokenizer_create(path.c_str(), &t);
            CHECK((rc == DAWN_OK) == (t != nullptr));
            CHECK(rc == DAWN_OK || rc == DAWN_ERR_IO || rc == DAWN_ERR_OOM);
            if (t) {
                (void)dawn_tokenizer_vocab_size(t);
                run_texts(t, texts);
                CHECK(dawn_tokenizer_set_max_length(t, 1) == DAWN_ERR_INVALID_ARG);
                CHECK(dawn_tokenizer_set_max_length(t, 8) == DAWN_OK);
                run_texts(t, texts);
                dawn_tokenizer_destroy(t);
            }
        } else if (starts(n, "cfg_")) {
            rc = dawn_embedder_check_files(good_st.c_str(), path.c_str());
            CHECK(rc == DAWN_OK || rc == DAWN_ERR_IO || rc == DAWN_ERR_UNSUPPORTED || rc == DAWN_ERR_OOM);
        } else if (starts(n, "st_") && ends(n, ".safetensors")) {
            const int rc2 = dawn_embedder_check_files(path.c_str(), nullptr);  // against the built-in MiniLM-L6 config
            CHECK(rc2 == DAWN_OK || rc2 == DAWN_ERR_IO || rc2 == DAWN_ERR_UNSUPPORTED || rc2 == DAWN_ERR_OOM);
            rc = dawn_embedder_check_files(path.c_str(), good_cfg.c_str());
            CHECK(rc == DAWN_OK || rc == DAWN_ERR_IO || rc == DAWN_ERR_UNSUPPORTED || rc == DAWN_ERR_OOM);
        } else if (n == "helpers.bin") {
            run_helpers(slurp(path));
            rc = 0;
        } else {
            continue;
        }
        std::printf("%s rc=%d %s\n", n.c_str(), rc, rc ? dawn_last_error() : "");
    }
    return failures ? 1 : 0;
}
