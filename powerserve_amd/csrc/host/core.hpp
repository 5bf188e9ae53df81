// Host-side core types of the MI355X backend, mirroring PowerServe's src/core (same names, same meaning) so
// that graph-builder code written against the reference compiles against this tree:
//   Tensor / Shape / Stride      src/core/tensor.hpp:27-91, src/core/typedefs.hpp:27-29
//   DataType (+ GGML_Q4_K/Q5_K/Q6_K)  src/core/data_type.hpp:24-35  (the reference stops at Q8_0 and aborts on K-quants)
//   BaseBuffer / HIPBuffer       src/core/buffer.hpp:21-26, src/backend/cpu_buffer.hpp:23-64 (stride in BYTES)
//   ModelConfig / HyperParams    src/core/config.hpp:33-147
//   POWERSERVE_ASSERT / ABORT    src/core/logger.hpp:56-82 (log + abort, or throw under POWERSERVE_EXCEPTION_ABORT)
#pragma once
#include <algorithm>
#include <array>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <memory>
#include <numeric>
#include <stdexcept>
#include <string>
#include <vector>

namespace powerserve {

struct AbortException : std::runtime_error { using std::runtime_error::runtime_error; };

[[noreturn]] inline void ps_fail(const char *kind, const char *expr, const char *file, int line, const std::string &msg) {
    std::string full = std::string("[") + kind + "] " + file + ":" + std::to_string(line) + ": " + expr + (msg.empty() ? "" : (": " + msg));
    std::fprintf(stderr, "%s\n", full.c_str());
#if defined(POWERSERVE_EXCEPTION_ABORT)
    throw AbortException(full);
#else
    std::abort();
#endif
}
#define POWERSERVE_ASSERT(expr, ...) do { if (!(expr)) ::powerserve::ps_fail("ASSERT", #expr, __FILE__, __LINE__, std::string("" __VA_ARGS__)); } while (0)
#define POWERSERVE_ABORT(msg) ::powerserve::ps_fail("ABORT", "abort", __FILE__, __LINE__, std::string(msg))
#define POWERSERVE_UNUSED(x) ((void)(x))

using Token = int32_t;
static constexpr size_t max_n_dims = 4;
using Shape  = std::array<size_t, max_n_dims>;
using Stride = std::array<size_t, max_n_dims>;

// values chosen so that static_cast<int> is NOT the ggml enum: use to_ps_dtype()/from ggml for the C-ABI
enum class DataType { UNKNOWN, FP32, FP16, INT32, INT64, GGML_Q4_0, GGML_Q8_0, GGML_Q4_K, GGML_Q5_K, GGML_Q6_K, COUNT };

// ggml_type values used by the C-ABI (include/ps_hip.h)
inline int to_ggml_type(DataType t) {
    switch (t) {
    case DataType::FP32: return 0; case DataType::FP16: return 1; case DataType::GGML_Q4_0: return 2;
    case DataType::GGML_Q8_0: return 8; case DataType::GGML_Q4_K: return 12; case DataType::GGML_Q5_K: return 13; case DataType::GGML_Q6_K: return 14;
    case DataType::INT32: return 26; case DataType::INT64: return 27;
    default: POWERSERVE_ABORT("unsupported data type");
    }
}
inline DataType from_ggml_type(int t) {
    switch (t) {
    case 0: return DataType::FP32; case 1: return DataType::FP16; case 2: return DataType::GGML_Q4_0;
    case 8: return DataType::GGML_Q8_0; case 12: return DataType::GGML_Q4_K; case 13: return DataType::GGML_Q5_K; case 14: return DataType::GGML_Q6_K;
    case 26: return DataType::INT32; case 27: return DataType::INT64;
    default: POWERSERVE_ABORT("unsupported ggml data type " + std::to_string(t));
    }
}
inline size_t get_type_size(DataType t) {
    switch (t) {
    case DataType::FP32: case DataType::INT32: return 4; case DataType::FP16: return 2; case DataType::INT64: return 8;
    case DataType::GGML_Q4_0: return 18; case DataType::GGML_Q8_0: return 34; case DataType::GGML_Q4_K: return 144; case DataType::GGML_Q5_K: return 176; case DataType::GGML_Q6_K: return 210;
    default: POWERSERVE_ABORT("get_type_size");
    }
}
inline size_t get_block_size(DataType t) {
    switch (t) {
    case DataType::GGML_Q4_0: case DataType::GGML_Q8_0: return 32;
    case DataType::GGML_Q4_K: case DataType::GGML_Q5_K: case DataType::GGML_Q6_K: return 256;
    default: return 1;
    }
}

struct BaseBuffer { virtual ~BaseBuffer() = default; };
using BufferPtr = std::shared_ptr<BaseBuffer>;

// Device buffer view: never owns memory (weights belong to the backend, intermediates to its arena).
// For quantized weights m_data is the ps_weight handle, exactly what ps_tensor.data carries.
struct HIPBuffer : BaseBuffer {
    Stride m_stride; // bytes
    void *m_data;
    HIPBuffer(Stride stride, void *data) : m_stride(stride), m_data(data) {}
};
// host buffer for results handed back to the caller (LogitsVector)
struct CPUBuffer : BaseBuffer {
    Stride m_stride;
    void *m_data;
    std::vector<char> m_storage;
    CPUBuffer(Stride stride, size_t bytes) : m_stride(stride), m_storage(bytes) { m_data = m_storage.data(); }
};

struct Tensor {
    DataType m_dtype = DataType::UNKNOWN;
    Shape m_shape    = {0};
    BufferPtr m_data = nullptr;

    Tensor() = default;
    Tensor(DataType dtype, const Shape &shape) : m_dtype(dtype) {
        for (size_t i = 0; i < shape.size(); i++) m_shape[i] = std::max(shape[i], size_t(1));
    }
    size_t n_dims() const { for (size_t i = max_n_dims - 1; i > 0; i--) if (m_shape[i] > 1) return i + 1; return 1; }
    size_t n_elements() const { return m_shape[0] * m_shape[1] * m_shape[2] * m_shape[3]; }
    template <typename Buffer> auto get() const -> Buffer & { return dynamic_cast<Buffer &>(*m_data); }
    bool is_quantized() const { return m_dtype != DataType::FP32; }
    int64_t nrows() const { return (int64_t)(m_shape[1] * m_shape[2] * m_shape[3]); }
    size_t element_size() const { return get_type_size(m_dtype); }
    size_t row_size(int64_t ne) const {
        POWERSERVE_ASSERT(ne % (int64_t)get_block_size(m_dtype) == 0);
        return element_size() * (size_t)ne / get_block_size(m_dtype);
    }
};
inline bool tensor_can_mul_mat(const Tensor *t0, const Tensor *t1) {
    return t0->m_shape[0] == t1->m_shape[0] && t1->m_shape[2] % t0->m_shape[2] == 0 && t1->m_shape[3] % t0->m_shape[3] == 0;
}
inline bool tensor_can_repeat(const Tensor *t0, const Tensor *t1) {
    return t1->m_shape[0] % t0->m_shape[0] == 0 && t1->m_shape[1] % t0->m_shape[1] == 0 && t1->m_shape[2] % t0->m_shape[2] == 0 &&
           t1->m_shape[3] % t0->m_shape[3] == 0;
}

struct SamplerConfig { // HyperParams::SamplerConfig, src/core/config.hpp:34-47
    uint64_t seed     = (uint64_t)-1; // -1: take one from std::random_device
    float temperature = 0.80f;
    float top_p       = 0.95f;
    size_t top_k      = 40;
    size_t min_keep   = 0;
    int penalty_last_n    = 64;
    float penalty_repeat  = 1.00f;
    float penalty_freq    = 0.00f;
    float penalty_present = 0.00f;
    bool penalize_nl      = false;
    bool ignore_eos       = false;
};

struct HyperParams { // hparams.json (src/core/config.cpp:30-67): every key optional, defaults as the reference's
    SamplerConfig sampler_config;
    size_t n_threads  = 4;   // unused on the GPU (kept for config compatibility; clamped to the host's cores like the reference)
    size_t batch_size = 128; // prefill chunk
    HyperParams() = default;
    explicit HyperParams(const std::string &params_file);
};

// workspace.json (src/core/config.cpp:121-152, key names config.hpp:24-26): which hparams file, main and draft model directories,
// all relative to the work folder
struct Config {
    HyperParams hyper_params;
    std::string main_model_dir, draft_model_dir;
    Config(const std::string &work_folder, const std::string &workspace_config_path);
};

struct ModelConfig {
    uint32_t version = 0;
    std::string arch, model_id;
    struct LLMConfig {
        struct RopeConfig {
            int n_dims = 128, n_ctx_orig = 2048;
            float freq_base = 10000.0f, freq_scale = 1.0f, ext_factor = 0.0f, attn_factor = 1.0f, beta_fast = 32.0f, beta_slow = 0.0f;
            int rope_type = -1;
        } rope_config;
        uint32_t dim = 0, hidden_dim = 0, n_layers = 0, n_heads = 0, n_kv_heads = 0, seq_len = 0, vocab_size = 0, kv_dim = 0, head_size = 0;
        float norm_eps = 1e-5f;
    } llm;
    ModelConfig() = default;
    explicit ModelConfig(const std::string &model_json_path); // same schema as src/core/config.cpp:68-104
};

} // namespace powerserve
