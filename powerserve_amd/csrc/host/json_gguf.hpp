// Minimal JSON value parser (model.json, src/core/config.cpp:68-120) and GGUF v2/v3 reader
// (what gguf_init_from_file gives the reference, libs/ggml/src/ggml.c:23249): the file is mmap'ed and tensors
// are handed out as (type, ne, pointer) without copies.
#pragma once
#include "core.hpp"

#include <map>

namespace powerserve {

struct JsonValue {
    enum Kind { NUL, NUM, STR, BOOL, OBJ, ARR } kind = NUL;
    double num = 0;
    uint64_t u64 = 0;      // NUM written as a plain integer: its exact value (negative ones wrapped), which a double cannot hold above 2^53
    bool is_int = false;
    bool b = false;
    std::string str;
    std::map<std::string, JsonValue> obj;
    std::vector<JsonValue> arr;
    const JsonValue &at(const std::string &k) const;
    bool contains(const std::string &k) const { return kind == OBJ && obj.count(k); }
    static JsonValue parse_file(const std::string &path);
};

struct GGUFTensor {
    std::string name;
    int type = 0; // ggml_type
    std::vector<int64_t> ne;
    const void *data = nullptr;
    size_t nbytes = 0;
};

struct GGUFFile {
    explicit GGUFFile(const std::string &path);
    ~GGUFFile();
    GGUFFile(const GGUFFile &) = delete;
    const GGUFTensor *find(const std::string &name) const;
    std::vector<GGUFTensor> tensors;
    std::map<std::string, std::string> kv_str;
    std::map<std::string, double> kv_num;

private:
    void *m_map = nullptr;
    size_t m_size = 0;
};

size_t ggml_row_size_host(int type, int64_t k);

} // namespace powerserve
