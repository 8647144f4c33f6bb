// host_common.hpp — error plumbing of the C ABI that needs no HIP header: shared by every translation unit, and all
// the pure-host ones (tokenizer.cpp, model_files.cpp, host_helpers.cpp — the sanitizer build of tests/native compiles exactly those
// with g++ -fsanitize=address,undefined) include nothing else of the library.
#pragma once
#include <cstdarg>
#include <cstdio>
#include <exception>
#include <new>
#include <string>

#include "../../include/dawn_hip_debug.h"  // (dawn_hip.h + the test / measurement hooks this library also exports)

namespace dawn {

std::string& last_error();  // thread-local

inline int fail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    last_error() = buf;
    return code;
}

#define DAWN_TRY(expr)          \
    do {                        \
        int _rc = (expr);       \
        if (_rc != DAWN_OK) return _rc; \
    } while (0)

// A C ABI must not let C++ exceptions through (the callers are Rust / C): every entry point that can allocate runs
// inside guarded().
template <class F>
int guarded(F&& f) noexcept {
    try {
        return f();
    } catch (const std::bad_alloc&) {
        return fail(DAWN_ERR_OOM, "out of host memory");
    } catch (const std::exception& e) {
        return fail(DAWN_ERR_INVALID_ARG, "%s", e.what());
    } catch (...) {
        return fail(DAWN_ERR_HIP, "unexpected exception");
    }
}

// vector.rs host restatements used by the ABI-side validation
bool host_is_normalized(const float* v);
size_t host_first_not_normalized(const float* v, size_t B);  // the same predicate, B vectors: the first that fails it, or B

}  // namespace dawn
