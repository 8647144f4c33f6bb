// mini_json.hpp — a small JSON reader (objects / arrays / strings / numbers / literals): enough for safetensors
// headers, HF config.json and tokenizer.json.  The input is an untrusted file: nesting is bounded (kMaxDepth — parse()
// recurses once per level, and so does ~JVal), the number of values is bounded (kMaxNodes — a JVal is ~100 B, so a file
// of "0,0,0,..." would otherwise cost 50 x its size in host memory), numbers are scanned inside [p, end) only (the
// buffer is not NUL-terminated), and any violation just clears `ok`.
#pragma once
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <utility>
#include <vector>

namespace dawn {

inline void append_utf8(std::string& out, uint32_t cp) {
    if (cp < 0x80) out += (char)cp;
    else if (cp < 0x800) {
        out += (char)(0xC0 | (cp >> 6));
        out += (char)(0x80 | (cp & 0x3F));
    } else if (cp < 0x10000) {
        out += (char)(0xE0 | (cp >> 12));
        out += (char)(0x80 | ((cp >> 6) & 0x3F));
        out += (char)(0x80 | (cp & 0x3F));
    } else {
        out += (char)(0xF0 | (cp >> 18));
        out += (char)(0x80 | ((cp >> 12) & 0x3F));
        out += (char)(0x80 | ((cp >> 6) & 0x3F));
        out += (char)(0x80 | (cp & 0x3F));
    }
}

struct JVal {
    enum Kind { Null, Bool, Num, Str, Arr, Obj } kind = Null;
    double num = 0;
    bool b = false;
    std::string str;
    std::vector<JVal> arr;
    std::vector<std::pair<std::string, JVal>> obj;
    const JVal* get(const std::string& k) const {
        for (auto& kv : obj)
            if (kv.first == k) return &kv.second;
        return nullptr;
    }
};

struct JParser {
    static constexpr int kMaxDepth = 64;
    static constexpr size_t kMaxNodes = (size_t)4 << 20;
    const char* p;
    const char* end;
    bool ok = true;
    int depth = 0;
    size_t nodes = 0;
    void ws() {
        while (p < end && (*p == ' ' || *p == '\n' || *p == '\t' || *p == '\r')) ++p;
    }
    bool lit(const char* s) {
        size_t n = std::strlen(s);
        if ((size_t)(end - p) >= n && std::memcmp(p, s, n) == 0) {
            p += n;
            return true;
        }
        return false;
    }
    std::string parse_string() {
        std::string out;
        if (p >= end || *p != '"') {
            ok = false;
            return out;
        }
        ++p;
        while (p < end && *p != '"') {
            if (*p == '\\' && p + 1 < end) {
                ++p;
                switch (*p) {
                    case 'n': out += '\n'; break;
                    case 't': out += '\t'; break;
                    case 'r': out += '\r'; break;
                    case 'b': out += '\b'; break;
                    case 'f': out += '\f'; break;
                    case 'u': {  // \uXXXX (+ surrogate pair) -> UTF-8
                        auto hex4 = [&](const char* q, uint32_t& v) {
                            v = 0;
                            if (end - q < 4) return false;
                            for (int i = 0; i < 4; ++i) {
                                const char c = q[i];
                                uint32_t d;
                                if (c >= '0' && c <= '9') d = c - '0';
                                else if (c >= 'a' && c <= 'f') d = 10 + c - 'a';
                                else if (c >= 'A' && c <= 'F') d = 10 + c - 'A';
                                else return false;
                                v = v * 16 + d;
                            }
                            return true;
                        };
                        uint32_t cp = 0;
                        if (!hex4(p + 1, cp)) {
                            ok = false;
                            return out;
                        }
                        p += 4;
                        if (cp >= 0xD800 && cp <= 0xDBFF && end - p > 6 && p[1] == '\\' && p[2] == 'u') {
                            uint32_t lo = 0;
                            if (hex4(p + 3, lo) && lo >= 0xDC00 && lo <= 0xDFFF) {
                                cp = 0x10000 + ((cp - 0xD800) << 10) + (lo - 0xDC00);
                                p += 6;
                            }
                        }
                        append_utf8(out, cp);
                        break;
                    }
                    default: out += *p;
                }
                ++p;
            } else {
                out += *p++;
            }
        }
        if (p >= end) ok = false;
        else ++p;
        return out;
    }
    // JSON number grammar only (-? digits [. digits] [e[+-] digits]), at most 63 characters, converted from a local
    // NUL-terminated copy
    bool parse_number(double& out) {
        char buf[64];
        size_t n = 0;
        const char* q = p;
        auto take = [&](bool cond) {
            if (!cond || n + 1 >= sizeof(buf)) return false;
            buf[n++] = *q++;
            return true;
        };
        auto digits = [&]() {
            size_t k = 0;
            while (q < end && *q >= '0' && *q <= '9' && take(true)) ++k;
            return k;
        };
        if (q < end && *q == '-') take(true);
        if (!digits()) return false;
        if (q < end && *q == '.') {
            take(true);
            if (!digits()) return false;
        }
        if (q < end && (*q == 'e' || *q == 'E')) {
            take(true);
            if (q < end && (*q == '+' || *q == '-')) take(true);
            if (!digits()) return false;
        }
        if (q < end && ((*q >= '0' && *q <= '9') || *q == '.')) return false;  // longer than the local buffer
        buf[n] = 0;
        out = std::strtod(buf, nullptr);
        p = q;
        return true;
    }
    JVal parse() {
        JVal v;
        if (++nodes > kMaxNodes || depth >= kMaxDepth) {
            ok = false;
            return v;
        }
        struct Level {
            int& d;
            explicit Level(int& x) : d(x) { ++d; }
            ~Level() { --d; }
        } level(depth);
        ws();
        if (p >= end) {
            ok = false;
            return v;
        }
        if (*p == '{') {
            v.kind = JVal::Obj;
            ++p;
            ws();
            if (p < end && *p == '}') {
                ++p;
                return v;
            }
            while (ok) {
                ws();
                std::string k = parse_string();
                ws();
                if (p >= end || *p != ':') {
                    ok = false;
                    break;
                }
                ++p;
                v.obj.emplace_back(k, parse());
                ws();
                if (p < end && *p == ',') {
                    ++p;
                    continue;
                }
                if (p < end && *p == '}') {
                    ++p;
                    break;
                }
                ok = false;
            }
        } else if (*p == '[') {
            v.kind = JVal::Arr;
            ++p;
            ws();
            if (p < end && *p == ']') {
                ++p;
                return v;
            }
            while (ok) {
                v.arr.push_back(parse());
                ws();
                if (p < end && *p == ',') {
                    ++p;
                    continue;
                }
                if (p < end && *p == ']') {
                    ++p;
                    break;
                }
                ok = false;
            }
        } else if (*p == '"') {
            v.kind = JVal::Str;
            v.str = parse_string();
        } else if (lit("true")) {
            v.kind = JVal::Bool;
            v.b = true;
        } else if (lit("false")) {
            v.kind = JVal::Bool;
        } else if (lit("null")) {
            v.kind = JVal::Null;
        } else {
            v.kind = JVal::Num;
            if (!parse_number(v.num)) ok = false;
        }
        return v;
    }
};


// A JSON number as an integer in [lo, hi]: false for anything else (strings, fractions, NaN, out of range) — a
// double -> integer cast of an unchecked value is undefined behaviour.
inline bool jint(const JVal* v, int64_t lo, int64_t hi, int64_t& out) {
    if (!v || v->kind != JVal::Num) return false;
    const double d = v->num;
    if (!(d >= (double)lo && d <= (double)hi)) return false;
    const int64_t i = (int64_t)d;
    if ((double)i != d) return false;
    out = i;
    return true;
}

inline bool read_file(const char* path, std::vector<char>& out, size_t max_bytes = (size_t)-1) {
    FILE* f = std::fopen(path, "rb");
    if (!f) return false;
    std::fseek(f, 0, SEEK_END);
    long n = std::ftell(f);
    std::fseek(f, 0, SEEK_SET);
    if (n < 0 || (size_t)n > max_bytes) {  // (a directory reports LONG_MAX or -1)
        std::fclose(f);
        return false;
    }
    out.resize((size_t)n);
    bool ok = n == 0 || std::fread(out.data(), 1, (size_t)n, f) == (size_t)n;
    std::fclose(f);
    return ok;
}

}  // namespace dawn
