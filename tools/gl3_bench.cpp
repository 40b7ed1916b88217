// gl3_bench — native llama-bench twin over the C-ABI of libgpullama_hip.so (no Python, no torch): what a compiled host such
// as the Java FFM shim of INTEGRATION.md does, written in C++.  Protocol and command line = J/bench/LlamaBench.java
// (J = /root/reference/src/main/java/org/beehive/gpullama3):
//   * options :99-109 — -m (repeatable / comma list), -p, -n (comma lists), -pg P,G (repeatable), -b, -d (comma list of context
//     depths), -r, -o / -oe md|csv|json|jsonl|sql, --delay, --no-warmup; defaults pp512 + tg128 at depth 0, 5 repetitions;
//   * test list :130-147 — for every depth: every pp, every tg, every pp+tg pair; name "pp512", "tg128", "pp512+tg128", "...@d4096",
//     with " b<batch>" appended when -b > 1 (:213);
//   * one repetition (runTest :233-254): d positions prefilled UNTIMED from position 0, then the timed window = nPrompt prompt tokens
//     (chunks of -b through the batched prefill, or single-token forwards with logits when -b 1) + nGen single-token forwards with
//     the logits copied to the host, at positions d ..; tokens / wall seconds;
//   * token ids java.util.Random(42).nextInt(vocab), indexed by ABSOLUTE position (:188-193); context = max(depth + tokens) + 8 (:173);
//   * mean and sample standard deviation over the repetitions (:205-212); the five output formats :309-372 column for column.
// Not in the reference: --ids (greedy ids of the first timed tg repetition), --scalar-dot / --f32-activation (the reference's
// -Dllama.VectorBitSize=0 / -Dllama.quantizeActivation=false arithmetic switches).
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include <sys/stat.h>

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

struct TestSpec {
    int n_prompt, n_gen, depth;
    int tokens() const { return n_prompt + n_gen; }
    std::string name() const {
        std::string base = (n_prompt > 0 && n_gen > 0) ? "pp" + std::to_string(n_prompt) + "+tg" + std::to_string(n_gen)
                           : n_prompt > 0 ? "pp" + std::to_string(n_prompt) : "tg" + std::to_string(n_gen);
        return depth > 0 ? base + "@d" + std::to_string(depth) : base;
    }
};

struct Result {
    std::string model, quant, backend, test;
    double size_gib, params_b, avg, stddev;
    std::vector<double> samples;
};

static std::vector<int> int_list(const char* s) {
    std::vector<int> v;
    for (const char* p = s; *p;) {
        v.push_back(atoi(p));
        while (*p && *p != ',' && *p != '+') ++p;
        if (*p) ++p;
    }
    return v;
}

static std::string samples_str(const Result& r, const char* sep) {
    std::string s;
    char b[32];
    for (size_t j = 0; j < r.samples.size(); ++j) { snprintf(b, sizeof b, "%.2f", r.samples[j]); if (j) s += sep; s += b; }
    return s;
}
static std::string json_row(const Result& r) {
    char b[1024];
    snprintf(b, sizeof b, "{\"model\": \"%s\", \"quant\": \"%s\", \"size_gib\": %.3f, \"params_b\": %.3f, \"backend\": \"%s\", \"test\": \"%s\", \"avg_ts\": %.2f, \"stddev_ts\": %.2f, \"samples_ts\": [",
             r.model.c_str(), r.quant.c_str(), r.size_gib, r.params_b, r.backend.c_str(), r.test.c_str(), r.avg, r.stddev);
    return std::string(b) + samples_str(r, ", ") + "]}";
}
static void print_results(const std::string& fmt, const std::vector<Result>& rs, FILE* f) {
    if (fmt == "csv") {
        fprintf(f, "model,quant,size_gib,params_b,backend,test,avg_ts,stddev_ts,samples\n");
        for (const Result& r : rs)
            fprintf(f, "%s,%s,%.3f,%.3f,%s,%s,%.2f,%.2f,%s\n", r.model.c_str(), r.quant.c_str(), r.size_gib, r.params_b, r.backend.c_str(), r.test.c_str(), r.avg, r.stddev,
                    samples_str(r, ";").c_str());
    } else if (fmt == "json") {
        fprintf(f, "[\n");
        for (size_t i = 0; i < rs.size(); ++i) fprintf(f, "  %s%s\n", json_row(rs[i]).c_str(), i + 1 < rs.size() ? "," : "");
        fprintf(f, "]\n");
    } else if (fmt == "jsonl") {
        for (const Result& r : rs) fprintf(f, "%s\n", json_row(r).c_str());
    } else if (fmt == "sql") {
        fprintf(f, "CREATE TABLE IF NOT EXISTS llama_bench (model TEXT, quant TEXT, size_gib REAL, params_b REAL, backend TEXT, test TEXT, avg_ts REAL, stddev_ts REAL);\n");
        for (const Result& r : rs)
            fprintf(f, "INSERT INTO llama_bench VALUES ('%s', '%s', %.3f, %.3f, '%s', '%s', %.2f, %.2f);\n", r.model.c_str(), r.quant.c_str(), r.size_gib, r.params_b,
                    r.backend.c_str(), r.test.c_str(), r.avg, r.stddev);
    } else {
        fprintf(f, "\n| model | quant | size | params | backend | test | t/s |\n| ----- | ----- | ---: | -----: | ------- | ---- | --: |\n");
        for (const Result& r : rs)
            fprintf(f, "| %s | %s | %.2f GiB | %.2f B | %s | %s | %.2f ± %.2f |\n", r.model.c_str(), r.quant.c_str(), r.size_gib, r.params_b, r.backend.c_str(), r.test.c_str(),
                    r.avg, r.stddev);
    }
    fflush(f);
}

struct BenchOpts {
    int batch = 1, reps = 5, delay = 0;
    bool warmup = true, print_ids = false;
    uint32_t flags = 0;
};

#define CK(call)                                                                                        \
    do {                                                                                                \
        const int32_t r_ = (call);                                                                      \
        if (r_ != GL3_OK) { fprintf(stderr, "%s failed (%d): %s\n", #call, r_, ctx ? gl3_last_error(ctx) : gl3_gguf_last_error(nullptr)); return r_; } \
    } while (0)

// prefill(...) :257-273 — count tokens from toks[start..] at positions start..: chunks of -b, or single-token forwards with logits
static int32_t feed(gl3_ctx* ctx, const std::vector<int32_t>& toks, int start, int count, int batch, float* logits) {
    if (count <= 0) return GL3_OK;
    if (batch > 1) {
        for (int off = 0; off < count; off += batch) CK(gl3_forward_prefill(ctx, toks.data() + start + off, count - off < batch ? count - off : batch, start + off));
    } else {
        for (int i = 0; i < count; ++i) CK(gl3_forward_decode(ctx, toks[start + i], start + i, logits, nullptr));
    }
    return GL3_OK;
}

static int32_t bench_model(const std::string& path, const std::vector<TestSpec>& tests, const BenchOpts& o, std::vector<Result>* out) {
    gl3_ctx* ctx = nullptr;
    int max_tokens = 0;
    for (const TestSpec& t : tests) max_tokens = t.depth + t.tokens() > max_tokens ? t.depth + t.tokens() : max_tokens;
    gl3_model_desc opts{};
    opts.struct_size = sizeof(opts);
    opts.ctx = (tests.empty() ? 1024 : max_tokens) + 8;
    opts.max_batch = o.batch;
    opts.tp_size = 1;
    opts.flags = o.flags;
    const double t_load = now_s();
    CK(gl3_load_gguf(path.c_str(), &opts, &ctx));
    double plan_ms = 0, copy_ms = 0;
    gl3_get_init_ms(ctx, &plan_ms, &copy_ms);
    gl3_gguf* g = nullptr;
    gl3_model_desc d{};
    if (gl3_gguf_open(path.c_str(), &g) != GL3_OK || gl3_gguf_model_desc(g, &d, nullptr) != GL3_OK) { fprintf(stderr, "cannot re-read %s\n", path.c_str()); gl3_destroy(ctx); return GL3_E_ARG; }
    gl3_gguf_close(g);
    // Result columns as LlamaBench.benchModel :180-185: file name without .gguf, the matrices' quantisation, file size, parameter estimate
    std::string name = path.substr(path.find_last_of('/') == std::string::npos ? 0 : path.find_last_of('/') + 1);
    if (name.size() > 5 && name.compare(name.size() - 5, 5, ".gguf") == 0) name.resize(name.size() - 5);
    const std::string quant = d.weight_type == GL3_TYPE_F16 ? "FP16" : d.weight_type == GL3_TYPE_Q8_0 ? "Q8_0" : d.weight_type == GL3_TYPE_Q4_0 ? "Q4_0" : "?";
    struct stat st{};
    stat(path.c_str(), &st);
    const double size_gib = (double)st.st_size / (1024.0 * 1024.0 * 1024.0);
    const double params_b = (double)st.st_size / (quant == "Q8_0" ? 34.0 / 32.0 : 2.0) / 1e9;          // estimateParamsB :276-283
    fprintf(stderr, "loaded %s: dim %d, layers %d, vocab %d, %s in %.2f s (plan %.0f ms, weights %.0f ms)\n", name.c_str(), d.dim, d.n_layers, d.vocab, quant.c_str(),
            now_s() - t_load, plan_ms, copy_ms);
    JavaRandom rng(42);
    std::vector<int32_t> toks((size_t)max_tokens);
    for (auto& t : toks) t = rng.nextInt(d.vocab);
    std::vector<float> logits((size_t)d.vocab);
    for (const TestSpec& t : tests) {
        if (o.delay > 0) std::this_thread::sleep_for(std::chrono::seconds(o.delay));
        std::vector<double> samples;
        std::vector<int32_t> ids;
        for (int rep = o.warmup ? -1 : 0; rep < o.reps; ++rep) {           // rep -1 = the untimed warm-up repetition
            CK(feed(ctx, toks, 0, t.depth, o.batch, logits.data()));      // untimed depth prefill
            const double t0 = now_s();
            CK(feed(ctx, toks, t.depth, t.n_prompt, o.batch, logits.data()));
            for (int i = 0; i < t.n_gen; ++i) {
                const int pos = t.depth + t.n_prompt + i;
                int32_t id = 0;
                const bool want_id = o.print_ids && rep == 0;
                CK(gl3_forward_decode(ctx, toks[pos], pos, logits.data(), want_id ? &id : nullptr));
                if (want_id) ids.push_back(id);
            }
            const double dt = now_s() - t0;
            if (rep >= 0) samples.push_back(t.tokens() / dt);
        }
        double avg = 0, var = 0;
        for (double s : samples) avg += s;
        avg /= (double)samples.size();
        for (double s : samples) var += (s - avg) * (s - avg);
        const double sd = samples.size() > 1 ? sqrt(var / (double)(samples.size() - 1)) : 0.0;
        const std::string test = o.batch > 1 ? t.name() + " b" + std::to_string(o.batch) : t.name();
        out->push_back(Result{name, quant, "HIP gfx950", test, size_gib, params_b, avg, sd, samples});
        fprintf(stderr, "[bench] %-28s %-14s %8.2f ± %.2f t/s\n", name.c_str(), test.c_str(), avg, sd);
        if (!ids.empty()) { fprintf(stderr, "greedy ids:"); for (int32_t v : ids) fprintf(stderr, " %d", v); fprintf(stderr, "\n"); }
    }
    gl3_destroy(ctx);
    return GL3_OK;
}

int main(int argc, char** argv) {
    std::vector<std::string> models;
    std::vector<int> pps, tgs, depths;
    std::vector<std::pair<int, int>> pgs;
    BenchOpts o;
    std::string out = "md", out_err;
    for (int i = 1; i < argc; ++i) {
        const std::string a = argv[i];
        auto val = [&]() -> const char* { return i + 1 < argc ? argv[++i] : "0"; };
        if (a == "-m" || a == "--model") { std::string v = val(); for (size_t p = 0; p <= v.size();) { const size_t q = v.find(',', p); models.push_back(v.substr(p, q == std::string::npos ? q : q - p)); if (q == std::string::npos) break; p = q + 1; } }
        else if (a == "-p" || a == "--n-prompt") { for (int v : int_list(val())) pps.push_back(v); }
        else if (a == "-n" || a == "--n-gen") { for (int v : int_list(val())) tgs.push_back(v); }
        else if (a == "-pg") { const std::vector<int> v = int_list(val()); if (v.size() != 2) { fprintf(stderr, "-pg wants P,G\n"); return 2; } pgs.emplace_back(v[0], v[1]); }
        else if (a == "-b" || a == "--batch-size") o.batch = atoi(val());
        else if (a == "-d" || a == "--n-depth") { for (int v : int_list(val())) depths.push_back(v); }
        else if (a == "-r" || a == "--repetitions") o.reps = atoi(val());
        else if (a == "-o" || a == "--output") out = val();
        else if (a == "-oe" || a == "--output-err") out_err = val();
        else if (a == "--delay") o.delay = atoi(val());
        else if (a == "--no-warmup") o.warmup = false;
        else if (a == "--ids") o.print_ids = true;
        else if (a == "--scalar-dot") o.flags |= GL3_FLAG_SCALAR_DOT;          // -Dllama.VectorBitSize=0 (F16 / Q4_0)
        else if (a == "--f32-activation") o.flags |= GL3_FLAG_F32_ACTIVATION;  // -Dllama.quantizeActivation=false (Q8_0)
        else { fprintf(stderr, "gl3_bench: unknown option %s\n", a.c_str()); return 2; }
    }
    if (models.empty()) {
        fprintf(stderr, "usage: gl3_bench -m model.gguf [-m model2.gguf] [-p 512] [-n 128] [-pg 512,128] [-b 1] [-d 0] [-r 5] [-o md|csv|json|jsonl|sql] [-oe fmt] [--delay s] [--no-warmup]\n");
        return 1;
    }
    if (o.batch < 1 || o.reps < 1) { fprintf(stderr, "gl3_bench: -b and -r must be >= 1\n"); return 2; }
    if (pps.empty() && tgs.empty() && pgs.empty()) { pps.push_back(512); tgs.push_back(128); }
    if (depths.empty()) depths.push_back(0);
    std::vector<TestSpec> tests;
    for (int d : depths) {
        for (int p : pps) if (p > 0) tests.push_back({p, 0, d});
        for (int n : tgs) if (n > 0) tests.push_back({0, n, d});
        for (auto& pg : pgs) tests.push_back({pg.first, pg.second, d});
    }
    std::vector<Result> results;
    int failed = 0;
    for (const std::string& m : models)
        if (bench_model(m, tests, o, &results) != GL3_OK) { fprintf(stderr, "[bench] %s FAILED (batch=%d unsupported for this model?)\n", m.c_str(), o.batch); ++failed; }
    print_results(out, results, stdout);
    if (!out_err.empty()) print_results(out_err, results, stderr);
    return failed ? 1 : 0;
}
