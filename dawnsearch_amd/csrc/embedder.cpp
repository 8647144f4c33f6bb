// embedder.cpp — dawn_embedder_* C ABI (src/embedding/embedding_service.rs:49-139).
#include "common.hpp"
using dawn::fail;
struct dawn_embedder { int device; };
extern "C" {
int dawn_embedder_create(const char*, const char*, int, dawn_embedder** out) {
    if (out) *out = nullptr;
    return fail(DAWN_ERR_UNSUPPORTED, "embedder not built yet");
}
void dawn_embedder_destroy(dawn_embedder* e) { delete e; }
int dawn_embedder_forward(dawn_embedder*, const uint32_t*, const int32_t*, int, float*) { return fail(DAWN_ERR_UNSUPPORTED, "embedder not built yet"); }
int dawn_embedder_forward_device(dawn_embedder*, const uint32_t*, const int32_t*, int, int, int, float*, void*) { return fail(DAWN_ERR_UNSUPPORTED, "embedder not built yet"); }
int dawn_embedder_hidden_states(dawn_embedder*, const uint32_t*, const int32_t*, int, float*) { return fail(DAWN_ERR_UNSUPPORTED, "embedder not built yet"); }
}
