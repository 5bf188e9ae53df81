// oracle/ref_gguf.cpp — TEST INFRASTRUCTURE ONLY (part of oracle/_ref/libps_ref.so).
//
// Writes a GGUF file with the REFERENCE's own writer (gguf_init_empty / gguf_set_val_* / gguf_add_tensor /
// gguf_write_to_file, libs/ggml/src/ggml.c) so that the product's readers (powerserve_amd/csrc/host/json_gguf.cpp,
// powerserve_amd/gguf.py) are tested on a file this repository's own writer did not produce.  Besides the caller's
// tensors the file carries one key of every GGUF value type and three arrays (the readers must step over all of them) and
// a caller-chosen general.alignment.  Used by oracle/gen_golden_gguf.py to make tests/golden/ref_written_model/.
#include "ggml.h"

#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

extern "C" int ref_gguf_write(const char *path, const char *arch, const char *model_name, int alignment, int n_tensors, const char **names,
                              const int32_t *types, const int64_t *ne /* [n_tensors][4] */, const void **data) {
    ggml_init_params ip{ggml_tensor_overhead() * (size_t)(n_tensors + 4), nullptr, true};
    ggml_context *ctx = ggml_init(ip);
    gguf_context *g   = gguf_init_empty();
    if (!ctx || !g) return 1;
    gguf_set_val_str(g, "general.architecture", arch);
    gguf_set_val_str(g, "general.name", model_name);
    gguf_set_val_u32(g, "general.alignment", (uint32_t)alignment);
    gguf_set_val_u32(g, "general.file_type", 15); // LLAMA_FTYPE_MOSTLY_Q4_K_M
    // one key of every scalar type ...
    gguf_set_val_u8(g, "test.u8", 200);
    gguf_set_val_i8(g, "test.i8", -100);
    gguf_set_val_u16(g, "test.u16", 60000);
    gguf_set_val_i16(g, "test.i16", -30000);
    gguf_set_val_i32(g, "test.i32", -2000000000);
    gguf_set_val_f32(g, "test.f32", 0.15625f);
    gguf_set_val_u64(g, "test.u64", 1ull << 40);
    gguf_set_val_i64(g, "test.i64", -(1ll << 40));
    gguf_set_val_f64(g, "test.f64", 1.0 / 3.0);
    gguf_set_val_bool(g, "test.bool", true);
    // ... and arrays, as a real checkpoint's tokenizer section has them
    const char *pieces[] = {"<s>", "</s>", "hello", "", "\xe4\xb8\x96\xe7\x95\x8c"};
    gguf_set_arr_str(g, "tokenizer.ggml.tokens", pieces, 5);
    const float scores[] = {0.f, -1.f, -2.5f, -3.f, -1e9f};
    gguf_set_arr_data(g, "tokenizer.ggml.scores", GGUF_TYPE_FLOAT32, scores, 5);
    const int32_t kinds[] = {3, 3, 1, 1, 1};
    gguf_set_arr_data(g, "tokenizer.ggml.token_type", GGUF_TYPE_INT32, kinds, 5);
    gguf_set_val_str(g, "tokenizer.ggml.model", "llama");

    for (int i = 0; i < n_tensors; i++) {
        ggml_tensor *t = ggml_new_tensor(ctx, (ggml_type)types[i], 4, ne + 4 * i);
        ggml_set_name(t, names[i]);
        t->data = const_cast<void *>(data[i]);
        gguf_add_tensor(g, t);
    }
    gguf_write_to_file(g, path, false);
    gguf_free(g);
    ggml_free(ctx);
    return 0;
}
