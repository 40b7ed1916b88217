// gguf_check — walks a GGUF file with the native reader (gl3_gguf.cpp) and touches everything it exposes: header, every metadata lookup the
// loader performs, the tensor table, the first and last byte of every tensor, the K-quant -> Q8_0 conversion of every K-quant tensor.
// Built with -fsanitize=address,undefined by `make -C gpullama3.java_amd/csrc asan` (sanitizer builds of the host-side
// code); scripts/sanitize.sh runs it over valid files and over truncated / corrupted copies, which must be REJECTED, not crash.
//   gguf_check file.gguf        exit 0 = parsed, 3 = rejected with an error message, anything else = a bug
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

#include "../include/gpullama3_hip.h"

int main(int argc, char** argv) {
    if (argc < 2) { fprintf(stderr, "usage: gguf_check file.gguf\n"); return 2; }
    gl3_gguf* g = nullptr;
    if (gl3_gguf_open(argv[1], &g) != GL3_OK) { printf("rejected: %s\n", gl3_gguf_last_error(g)); return 3; }
    gl3_model_desc d{};
    float theta = 0;
    const int32_t rd = gl3_gguf_model_desc(g, &d, &theta);
    if (rd != GL3_OK) { printf("rejected (model description): %s\n", gl3_gguf_last_error(g)); gl3_gguf_close(g); return 3; }
    const char* name = "?";
    gl3_gguf_meta_string(g, "general.name", &name);
    double v = 0;
    gl3_gguf_meta_number(g, "general.file_type", &v);
    unsigned long long sum = 0, kq = 0;
    const int32_t n = gl3_gguf_tensor_count(g);
    for (int32_t i = 0; i < n; ++i) {
        const char* tn = nullptr;
        int32_t type = 0;
        uint64_t ne[4] = {0, 0, 0, 0}, bytes = 0;
        const void* data = nullptr;
        if (gl3_gguf_tensor_info(g, i, &tn, &type, ne, &data, &bytes) != GL3_OK) { printf("rejected (tensor %d): %s\n", i, gl3_gguf_last_error(g)); gl3_gguf_close(g); return 3; }
        const uint8_t* p = static_cast<const uint8_t*>(data);
        if (bytes) sum += p[0] + p[bytes - 1] + strlen(tn);
        if (type == GL3_TYPE_Q4_K || type == GL3_TYPE_Q5_K || type == GL3_TYPE_Q6_K) {
            const uint64_t elems = ne[0] * (ne[1] ? ne[1] : 1) * (ne[2] ? ne[2] : 1) * (ne[3] ? ne[3] : 1);
            std::vector<uint8_t> q8((size_t)(elems / 32) * 34);
            if (gl3_kquant_to_q8_0(type, data, elems, q8.data()) != GL3_OK) { printf("rejected (K-quant tensor %s)\n", tn); gl3_gguf_close(g); return 3; }
            sum += q8.empty() ? 0 : q8[0] + q8.back();
            ++kq;
        }
    }
    printf("ok: %s arch %d dim %d layers %d vocab %d type %d tensors %d (K-quant converted: %llu) checksum %llu\n", name, d.arch, d.dim, d.n_layers, d.vocab,
           d.weight_type, n, kq, sum);
    gl3_gguf_close(g);
    return 0;
}
