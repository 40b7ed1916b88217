// gl3_bench — native llama-bench twin over the C-ABI of libgpullama_hip.so (no Python, no torch): what a compiled host such
// as the Java FFM shim of INTEGRATION.md does, written in C++.  Protocol = J/bench/LlamaBench.java:172-273: token ids from
// java.util.Random(42).nextInt(vocab) (:188-193), 1 untimed warm-up repetition + -r timed repetitions, pp = prompt tokens
// from position 0 in chunks of -b, tg = single-token forwards with logits D2H from position 0; tok/s mean +- sample stddev.
//   gl3_bench -m model.gguf [-p 512] [-n 128] [-b 512] [-r 5] [--ids]
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../include/gpullama3_hip.h"

// java.util.Random (48-bit LCG), nextInt(bound) with the power-of-two fast path and the rejection loop
struct JavaRandom {
    uint64_t seed;
    explicit JavaRandom(uint64_t s) : seed((s ^ 0x5DEECE66DULL) & ((1ULL << 48) - 1)) {}
    int32_t next(int bits) {
        seed = (seed * 0x5DEECE66DULL + 0xBULL) & ((1ULL << 48) - 1);
        return (int32_t)((int64_t)seed >> (48 - bits));
    }
    int32_t nextInt(int32_t bound) {
        int32_t r = next(31);
        const int32_t m = bound - 1;
        if ((bound & m) == 0) return (int32_t)(((int64_t)bound * (int64_t)r) >> 31);
        for (int32_t u = r; u - (r = u % bound) + m < 0; u = next(31)) {}
        return r;
    }
};

static double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

#define CK(call)                                                                                        \
    do {                                                                                                \
        const int32_t r_ = (call);                                                                      \
        if (r_ != GL3_OK) { fprintf(stderr, "%s failed (%d): %s\n", #call, r_, ctx ? gl3_last_error(ctx) : gl3_gguf_last_error(nullptr)); return 1; } \
    } while (0)

int main(int argc, char** argv) {
    std::string path;
    int n_prompt = 512, n_gen = 128, batch = 512, reps = 5;
    bool print_ids = false;
    uint32_t flags = 0;              // the reference's arithmetic switches: --scalar-dot, --f32-activation
    for (int i = 1; i < argc; ++i) {
        const std::string a = argv[i];
        auto val = [&]() { return i + 1 < argc ? argv[++i] : (char*)"0"; };
        if (a == "-m") path = val();
        else if (a == "-p") n_prompt = atoi(val());
        else if (a == "-n") n_gen = atoi(val());
        else if (a == "-b") batch = atoi(val());
        else if (a == "-r") reps = atoi(val());
        else if (a == "--ids") print_ids = true;
        else if (a == "--scalar-dot") flags |= GL3_FLAG_SCALAR_DOT;          // -Dllama.VectorBitSize=0 (F16 / Q4_0)
        else if (a == "--f32-activation") flags |= GL3_FLAG_F32_ACTIVATION;  // -Dllama.quantizeActivation=false (Q8_0)
        else { fprintf(stderr, "usage: gl3_bench -m model.gguf [-p N] [-n N] [-b N] [-r N] [--ids] [--scalar-dot] [--f32-activation]\n"); return 2; }
    }
    if (path.empty()) { fprintf(stderr, "gl3_bench: -m model.gguf is required\n"); return 2; }
    gl3_ctx* ctx = nullptr;
    gl3_model_desc opts{};
    opts.struct_size = sizeof(opts);
    opts.ctx = (n_prompt > n_gen ? n_prompt : n_gen) + n_gen + 8;       // LlamaBench: max(depth + tokens) + 8
    opts.max_batch = batch;
    opts.tp_size = 1;
    opts.flags = flags;
    const double t_load = now_s();
    CK(gl3_load_gguf(path.c_str(), &opts, &ctx));
    double plan_ms = 0, copy_ms = 0;
    gl3_get_init_ms(ctx, &plan_ms, &copy_ms);
    gl3_gguf* g = nullptr;
    gl3_model_desc d{};
    if (gl3_gguf_open(path.c_str(), &g) != GL3_OK || gl3_gguf_model_desc(g, &d, nullptr) != GL3_OK) { fprintf(stderr, "cannot re-read %s\n", path.c_str()); return 1; }
    const char* name = "?";
    gl3_gguf_meta_string(g, "general.name", &name);
    const std::string model_name = name;
    gl3_gguf_close(g);
    fprintf(stderr, "loaded %s: dim %d, layers %d, vocab %d, type %d in %.2f s (plan %.0f ms, weights %.0f ms)\n", model_name.c_str(), d.dim,
            d.n_layers, d.vocab, d.weight_type, now_s() - t_load, plan_ms, copy_ms);

    JavaRandom rng(42);
    std::vector<int32_t> toks((size_t)(n_prompt > n_gen ? n_prompt : n_gen));
    for (auto& t : toks) t = rng.nextInt(d.vocab);
    std::vector<float> logits((size_t)d.vocab);

    auto stats = [&](const std::vector<double>& tps, double* mean, double* sd) {
        double m = 0; for (double v : tps) m += v; m /= tps.size();
        double s = 0; for (double v : tps) s += (v - m) * (v - m);
        *mean = m; *sd = tps.size() > 1 ? sqrt(s / (tps.size() - 1)) : 0.0;
    };
    printf("| model | test | t/s |\n| --- | --- | --- |\n");
    if (n_prompt > 0) {
        std::vector<double> tps;
        for (int rep = -1; rep < reps; ++rep) {           // rep -1 = warm-up
            const double t0 = now_s();
            if (batch > 1) {
                for (int off = 0; off < n_prompt; off += batch)
                    CK(gl3_forward_prefill(ctx, toks.data() + off, n_prompt - off < batch ? n_prompt - off : batch, off));
            } else {
                for (int i = 0; i < n_prompt; ++i) CK(gl3_forward_decode(ctx, toks[i], i, logits.data(), nullptr));
            }
            if (rep >= 0) tps.push_back(n_prompt / (now_s() - t0));
        }
        double m, s; stats(tps, &m, &s);
        printf("| %s | pp%d -b %d | %.2f +- %.2f |\n", model_name.c_str(), n_prompt, batch, m, s);
    }
    if (n_gen > 0) {
        std::vector<double> tps;
        std::vector<int32_t> ids;
        for (int rep = -1; rep < reps; ++rep) {
            const double t0 = now_s();
            for (int i = 0; i < n_gen; ++i) {
                int32_t id = 0;
                CK(gl3_forward_decode(ctx, toks[i], i, logits.data(), print_ids && rep == 0 ? &id : nullptr));
                if (print_ids && rep == 0) ids.push_back(id);
            }
            if (rep >= 0) tps.push_back(n_gen / (now_s() - t0));
        }
        double m, s; stats(tps, &m, &s);
        printf("| %s | tg%d | %.2f +- %.2f |\n", model_name.c_str(), n_gen, m, s);
        if (print_ids) { printf("greedy ids:"); for (int32_t v : ids) printf(" %d", v); printf("\n"); }
    }
    gl3_destroy(ctx);
    return 0;
}
