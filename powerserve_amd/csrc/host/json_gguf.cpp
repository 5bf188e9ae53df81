#include "json_gguf.hpp"

#include <cctype>
#include <cerrno>
#include <cstring>
#include <fcntl.h>
#include <algorithm>
#include <filesystem>
#include <fstream>
#include <thread>
#include <sstream>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

namespace powerserve {

// ---------------------------------------------------------------- JSON
namespace {
struct JP {
    const std::string &s;
    size_t p = 0;
    void ws() { while (p < s.size() && std::isspace((unsigned char)s[p])) p++; }
    [[noreturn]] void bad(const char *m) { throw std::runtime_error(std::string("json: ") + m + " at offset " + std::to_string(p)); }
    JsonValue val() {
        ws();
        if (p >= s.size()) bad("unexpected end");
        JsonValue v;
        char c = s[p];
        if (c == '{') {
            v.kind = JsonValue::OBJ; p++; ws();
            if (s[p] == '}') { p++; return v; }
            for (;;) {
                ws(); JsonValue k = val(); if (k.kind != JsonValue::STR) bad("key");
                ws(); if (s[p++] != ':') bad("':'");
                v.obj[k.str] = val(); ws();
                if (s[p] == ',') { p++; continue; }
                if (s[p] == '}') { p++; break; }
                bad("',' or '}'");
            }
        } else if (c == '[') {
            v.kind = JsonValue::ARR; p++; ws();
            if (s[p] == ']') { p++; return v; }
            for (;;) {
                v.arr.push_back(val()); ws();
                if (s[p] == ',') { p++; continue; }
                if (s[p] == ']') { p++; break; }
                bad("',' or ']'");
            }
        } else if (c == '"') {
            v.kind = JsonValue::STR; p++;
            while (p < s.size() && s[p] != '"') { if (s[p] == '\\' && p + 1 < s.size()) p++; v.str.push_back(s[p++]); }
            p++;
        } else if (!s.compare(p, 4, "true")) { v.kind = JsonValue::BOOL; v.b = true; p += 4; }
        else if (!s.compare(p, 5, "false")) { v.kind = JsonValue::BOOL; p += 5; }
        else if (!s.compare(p, 4, "null")) { p += 4; }
        else {
            size_t e = p;
            while (e < s.size() && (std::isdigit((unsigned char)s[e]) || strchr("+-.eE", s[e]))) e++;
            if (e == p) bad("value");
            const std::string lit = s.substr(p, e - p);
            v.kind = JsonValue::NUM; v.num = std::stod(lit); p = e;
            if (lit.find_first_of(".eE") == std::string::npos) { // integer literal: keep all 64 bits (seed 18446744073709551615 = "pick one")
                errno = 0;
                const bool neg = lit[0] == '-';
                const unsigned long long mag = std::strtoull(lit.c_str() + (neg || lit[0] == '+'), nullptr, 10);
                if (errno == 0) { v.is_int = true; v.u64 = neg ? (uint64_t)0 - mag : (uint64_t)mag; }
            }
        }
        return v;
    }
};
} // namespace

const JsonValue &JsonValue::at(const std::string &k) const {
    auto it = obj.find(k);
    if (kind != OBJ || it == obj.end()) throw std::runtime_error("json: missing key '" + k + "'");
    return it->second;
}
JsonValue JsonValue::parse_file(const std::string &path) {
    std::ifstream f(path);
    if (!f.good()) throw std::runtime_error("cannot open " + path);
    std::stringstream ss; ss << f.rdbuf();
    std::string s = ss.str();
    JP p{s};
    return p.val();
}

HyperParams::HyperParams(const std::string &params_file) {
    try {
        const JsonValue j = JsonValue::parse_file(params_file);
        auto num = [](const JsonValue &o, const char *k, double dflt) { return o.contains(k) ? o.at(k).num : dflt; };
        auto flag = [](const JsonValue &o, const char *k, bool dflt) { return o.contains(k) ? (o.at(k).kind == JsonValue::BOOL ? o.at(k).b : o.at(k).num != 0) : dflt; };
        n_threads  = (size_t)num(j, "n_threads", (double)n_threads);
        batch_size = (size_t)num(j, "batch_size", (double)batch_size);
        const unsigned hw = std::thread::hardware_concurrency();
        if (hw != 0) n_threads = std::min<size_t>(n_threads, hw);
        if (j.contains("sampler") && j.at("sampler").kind == JsonValue::OBJ && !j.at("sampler").obj.empty()) {
            const JsonValue &s = j.at("sampler");
            auto &c = sampler_config;
            if (s.contains("seed")) { const JsonValue &sd = s.at("seed"); c.seed = sd.is_int ? sd.u64 : (sd.num < 0 ? (uint64_t)(int64_t)sd.num : (uint64_t)sd.num); }
            c.temperature = (float)num(s, "temperature", c.temperature);
            c.top_p = (float)num(s, "top_p", c.top_p);
            c.top_k = (size_t)num(s, "top_k", (double)c.top_k);
            c.min_keep = (size_t)num(s, "min_keep", (double)c.min_keep);
            c.penalty_last_n = (int)num(s, "penalty_last_n", c.penalty_last_n);
            c.penalty_repeat = (float)num(s, "penalty_repeat", c.penalty_repeat);
            c.penalty_freq = (float)num(s, "penalty_freq", c.penalty_freq);
            c.penalty_present = (float)num(s, "penalty_present", c.penalty_present);
            c.penalize_nl = flag(s, "penalize_nl", c.penalize_nl);
            c.ignore_eos = flag(s, "ignore_eos", c.ignore_eos);
        }
    } catch (const std::exception &e) {
        POWERSERVE_ABORT("failed parsing hyper param config file " + params_file + ": " + e.what());
    }
}

Config::Config(const std::string &work_folder, const std::string &workspace_config_path) {
    struct stat st;
    if (stat(work_folder.c_str(), &st) != 0 || !S_ISDIR(st.st_mode)) POWERSERVE_ABORT("work folder " + work_folder + " is not a directory");
    try {
        const JsonValue j = JsonValue::parse_file(workspace_config_path);
        auto join = [&](const std::string &rel) { return (std::filesystem::path(work_folder) / rel).string(); };
        if (j.contains("hparams_config")) hyper_params = HyperParams(join(j.at("hparams_config").str));
        if (j.contains("model_main")) main_model_dir = join(j.at("model_main").str);
        if (j.contains("model_draft")) draft_model_dir = join(j.at("model_draft").str);
    } catch (const std::exception &e) {
        POWERSERVE_ABORT("failed parsing artifact config file " + workspace_config_path + ": " + e.what());
    }
}

ModelConfig::ModelConfig(const std::string &path) {
    try {
        JsonValue j = JsonValue::parse_file(path);
        version = (uint32_t)j.at("version").num; arch = j.at("model_arch").str; model_id = j.at("model_id").str;
        const auto &l = j.at("llm_config");
        llm.dim = (uint32_t)l.at("embed_dim").num; llm.hidden_dim = (uint32_t)l.at("ffn_dim").num; llm.n_layers = (uint32_t)l.at("n_layers").num;
        llm.n_heads = (uint32_t)l.at("n_attn_heads").num; llm.n_kv_heads = (uint32_t)l.at("n_attn_kv_heads").num;
        llm.seq_len = (uint32_t)l.at("n_ctx").num; llm.vocab_size = (uint32_t)l.at("vocab_size").num; llm.kv_dim = (uint32_t)l.at("kv_dim").num;
        llm.head_size = (uint32_t)l.at("head_size").num; llm.norm_eps = (float)l.at("norm_eps").num;
        const auto &r = l.at("rope_config");
        auto &rc = llm.rope_config;
        rc.n_dims = (int)r.at("rope_dim").num; rc.n_ctx_orig = (int)r.at("n_rope_ctx_orig").num; rc.freq_base = (float)r.at("rope_freq_base").num;
        rc.freq_scale = (float)r.at("rope_freq_scale").num; rc.attn_factor = (float)r.at("rope_attn_factor").num; rc.rope_type = (int)r.at("rope_type").num;
        rc.ext_factor = 0.0f; rc.beta_fast = 32.0f; rc.beta_slow = 0.0f; // forced exactly like src/core/config.cpp:97-101
    } catch (const std::exception &e) {
        POWERSERVE_ABORT(std::string("failed parsing model config file ") + path + ": " + e.what());
    }
}

// ---------------------------------------------------------------- GGUF
size_t ggml_row_size_host(int type, int64_t k) {
    DataType t = from_ggml_type(type);
    return get_type_size(t) * (size_t)k / get_block_size(t);
}

namespace {
struct Rd {
    const uint8_t *b; size_t n, p = 0;
    template <typename T> T get() { if (p + sizeof(T) > n) throw std::runtime_error("gguf: truncated"); T v; memcpy(&v, b + p, sizeof(T)); p += sizeof(T); return v; }
    std::string str() { uint64_t l = get<uint64_t>(); if (p + l > n) throw std::runtime_error("gguf: truncated string"); std::string s((const char *)b + p, l); p += l; return s; }
};
}

GGUFFile::GGUFFile(const std::string &path) {
    int fd = open(path.c_str(), O_RDONLY);
    if (fd < 0) POWERSERVE_ABORT("cannot open " + path);
    struct stat st; fstat(fd, &st);
    m_size = (size_t)st.st_size;
    m_map = mmap(nullptr, m_size, PROT_READ, MAP_PRIVATE, fd, 0);
    close(fd);
    if (m_map == MAP_FAILED) POWERSERVE_ABORT("mmap failed for " + path);
    try {
        Rd r{(const uint8_t *)m_map, m_size};
        if (memcmp(r.b, "GGUF", 4)) throw std::runtime_error("gguf: bad magic");
        r.p = 4;
        uint32_t ver = r.get<uint32_t>();
        if (ver != 2 && ver != 3) throw std::runtime_error("gguf: unsupported version");
        uint64_t nt = r.get<uint64_t>(), nkv = r.get<uint64_t>();
        size_t align = 32;
        std::function<void(uint32_t, const std::string &)> skip = [&](uint32_t t, const std::string &key) {
            static const size_t sz[] = {1, 1, 2, 2, 4, 4, 4, 1, 0, 0, 8, 8, 8};
            if (t == 8) { std::string s = r.str(); if (!key.empty()) kv_str[key] = s; }
            else if (t == 9) { uint32_t et = r.get<uint32_t>(); uint64_t n = r.get<uint64_t>(); for (uint64_t i = 0; i < n; i++) skip(et, ""); }
            else if (t < 13) {
                double v = 0;
                switch (t) {
                case 0: v = r.get<uint8_t>(); break; case 1: v = r.get<int8_t>(); break; case 2: v = r.get<uint16_t>(); break;
                case 3: v = r.get<int16_t>(); break; case 4: v = r.get<uint32_t>(); break; case 5: v = r.get<int32_t>(); break;
                case 6: v = r.get<float>(); break; case 7: v = r.get<uint8_t>(); break; case 10: v = (double)r.get<uint64_t>(); break;
                case 11: v = (double)r.get<int64_t>(); break; case 12: v = r.get<double>(); break;
                }
                (void)sz;
                if (!key.empty()) kv_num[key] = v;
            } else throw std::runtime_error("gguf: bad kv type");
        };
        for (uint64_t i = 0; i < nkv; i++) { std::string k = r.str(); uint32_t t = r.get<uint32_t>(); skip(t, k); }
        if (kv_num.count("general.alignment")) align = (size_t)kv_num["general.alignment"];
        if (align == 0 || (align & (align - 1))) throw std::runtime_error("gguf: general.alignment must be a power of two");
        std::vector<uint64_t> offs;
        for (uint64_t i = 0; i < nt; i++) {
            GGUFTensor t; t.name = r.str();
            uint32_t nd = r.get<uint32_t>();
            for (uint32_t d = 0; d < nd; d++) t.ne.push_back((int64_t)r.get<uint64_t>());
            t.type = (int)r.get<uint32_t>();
            offs.push_back(r.get<uint64_t>());
            size_t rows = 1; for (size_t d = 1; d < t.ne.size(); d++) rows *= (size_t)t.ne[d];
            t.nbytes = ggml_row_size_host(t.type, t.ne[0]) * rows;
            tensors.push_back(std::move(t));
        }
        size_t data0 = (r.p + align - 1) / align * align;
        for (size_t i = 0; i < tensors.size(); i++) {
            if (data0 > m_size || offs[i] > m_size - data0 || tensors[i].nbytes > m_size - data0 - offs[i]) // (no wrap-around on hostile offsets)
                throw std::runtime_error("gguf: tensor data out of range: " + tensors[i].name);
            tensors[i].data = (const uint8_t *)m_map + data0 + offs[i];
        }
    } catch (const std::exception &e) {
        POWERSERVE_ABORT(path + ": " + e.what());
    }
}
GGUFFile::~GGUFFile() { if (m_map && m_map != MAP_FAILED) munmap(m_map, m_size); }
const GGUFTensor *GGUFFile::find(const std::string &name) const {
    for (auto &t : tensors) if (t.name == name) return &t;
    return nullptr;
}

} // namespace powerserve

// ---------------------------------------------------------------- C driver API: what the reader saw in a file
extern "C" {
void psh_set_error(const char *msg);
// One line per item into buf: "T <name> <type> <nbytes> <fnv1a-64 of the data, hex> <ne...>" per tensor (file order),
// "S <key> <value>" / "N <key> <value %.17g>" per scalar metadata key.  Returns the length needed (> cap: truncated), -1 on error.
int64_t psh_gguf_summary(const char *path, char *buf, size_t cap) {
    try {
        powerserve::GGUFFile f(path);
        std::string out;
        char line[512];
        for (const auto &t : f.tensors) {
            uint64_t h = 0xcbf29ce484222325ull;
            const uint8_t *p = (const uint8_t *)t.data;
            for (size_t i = 0; i < t.nbytes; i++) { h ^= p[i]; h *= 0x100000001b3ull; }
            snprintf(line, sizeof line, "T %s %d %zu %016llx", t.name.c_str(), t.type, t.nbytes, (unsigned long long)h);
            out += line;
            for (int64_t d : t.ne) out += " " + std::to_string(d);
            out += "\n";
        }
        for (const auto &kv : f.kv_str) out += "S " + kv.first + " " + kv.second + "\n";
        for (const auto &kv : f.kv_num) { snprintf(line, sizeof line, "N %s %.17g\n", kv.first.c_str(), kv.second); out += line; }
        if (cap) { const size_t n = std::min(cap - 1, out.size()); memcpy(buf, out.data(), n); buf[n] = 0; }
        return (int64_t)out.size() + 1;
    } catch (const std::exception &e) { psh_set_error(e.what()); return -1; }
}
}

extern "C" {
// "key=value" lines of what Config(work_folder, work_folder/workspace.json) parsed; returns the length needed, -1 on error
int64_t psh_config_summary(const char *work_folder, char *buf, size_t cap) {
    try {
        const std::string wf(work_folder);
        powerserve::Config c(wf, (std::filesystem::path(wf) / "workspace.json").string());
        const auto &h = c.hyper_params; const auto &s = h.sampler_config;
        char line[1024];
        snprintf(line, sizeof line, "n_threads=%zu\nbatch_size=%zu\nseed=%llu\ntemperature=%.9g\ntop_p=%.9g\ntop_k=%zu\nmin_keep=%zu\npenalty_last_n=%d\npenalty_repeat=%.9g\n"
                 "penalty_freq=%.9g\npenalty_present=%.9g\npenalize_nl=%d\nignore_eos=%d\n", h.n_threads, h.batch_size, (unsigned long long)s.seed, (double)s.temperature,
                 (double)s.top_p, s.top_k, s.min_keep, s.penalty_last_n, (double)s.penalty_repeat, (double)s.penalty_freq, (double)s.penalty_present, (int)s.penalize_nl, (int)s.ignore_eos);
        const std::string out = std::string(line) + "model_main=" + c.main_model_dir + "\nmodel_draft=" + c.draft_model_dir + "\n";
        if (cap) { const size_t n = std::min(cap - 1, out.size()); memcpy(buf, out.data(), n); buf[n] = 0; }
        return (int64_t)out.size() + 1;
    } catch (const std::exception &e) { psh_set_error(e.what()); return -1; }
}
}
