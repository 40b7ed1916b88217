// gl3_run — native generation loop over the C-ABI of libgpullama_hip.so (no Python, no torch): the token / position protocol of the
// reference's GPU engines, written as a compiled host would write it (the Java FFM shim of INTEGRATION.md §2 does the same calls).
//
// Protocols (J = /root/reference/src/main/java/org/beehive/gpullama3):
//   llama   InferenceEngine.generateTokensGPULlama            J/inference/InferenceEngine.java:293-382
//           the state's latestToken (begin-of-text, Llama.createNewState J/model/llama/Llama.java:49-53) is forwarded at startPosition,
//           then every prompt token (the chat format's prompt starts with begin-of-text again — it IS forwarded twice), then the
//           sampled tokens; the stop token is part of the result; loop while pos < min(maxTokens > 0 ? maxTokens : ctx, ctx).
//           With -b > 1: InferenceEngineWithBatchPrefillDecode.generateTokensGPULlama (:163-250) — the same (token, position)
//           sequence, its first N entries through gl3_forward_prefill in chunks of -b, no logits.
//   qwen3   InferenceEngine.generateTokensGPUQwen3             :383-475
//           no begin-of-text; prompt token k at startPosition + k; after the LAST prompt token the position is advanced twice
//           (`position++` and the loop's `++position`, :431,:413), so the first sampled token is forwarded at startPosition + N + 1
//           and KV row startPosition + N stays as State left it (zero).  Mirrored exactly: a drop-in must produce the ids the
//           reference produces, quirks included.  ONE deliberate deviation at the loop bound: the reference's qwen3 loop runs
//           `position < maxTokens` with the raw maxTokens (:413), i.e. it generates nothing for maxTokens <= 0 and would index
//           past the KV cache for maxTokens > contextLength; this tool needs a cache to size, so without -n it substitutes the
//           file's context length and always stops at it.  Pass -n to get the reference's bound (for -n <= contextLength the
//           id sequences are identical).
// Sampling: temperature 0 = greedy (Sampler.TENSOR_ARGMAX, first index of the maximum); else gl3_forward_decode_sample with the
// caller-side coin rng.nextFloat(1f) — RandomGeneratorFactory.getDefault().create(seed) = L32X64MixRandom in the reference
// (J/inference/sampler/Sampler.java:76-123); --rng lcg selects java.util.Random instead.  One coin per sampled token.
//
//   gl3_run -m model.gguf --ids 1,2,3 [--protocol llama|qwen3] [--bos ID] [-n maxTokens] [-b prefillBatch] [--start-pos P]
//           [--temperature T] [--top-p P] [--seed S] [--rng l32x64|lcg] [--stop id,id] [--scalar-dot] [--f32-activation]
// prints "generated: id id ..." (stdout) and the reference's metric lines (stderr).
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <set>
#include <string>
#include <vector>

#include "../include/gpullama3_hip.h"

// java.util.Random.nextFloat(): next(24) / (float)(1 << 24)
struct JavaRandom {
    uint64_t seed;
    explicit JavaRandom(uint64_t s) : seed((s ^ 0x5DEECE66DULL) & ((1ULL << 48) - 1)) {}
    int32_t next(int bits) {
        seed = (seed * 0x5DEECE66DULL + 0xBULL) & ((1ULL << 48) - 1);
        return (int32_t)((int64_t)seed >> (48 - bits));
    }
    float nextFloat() { return (float)next(24) / (float)(1 << 24); }
};
// jdk.random.L32X64MixRandom (the JDK's default RandomGenerator), written from the JDK's published algorithm: 32-bit LCG s = M s + a,
// xoroshiro64 (x0, x1), mixer lea32(s + x0).  UNPINNED like its Python twin (gpullama3.java_amd/javarand.py): no JVM here.
struct L32X64MixRandom {
    uint32_t a, s, x0, x1;
    static uint32_t murmur32(uint32_t z) { z = (z ^ (z >> 16)) * 0x85EBCA6Bu; z = (z ^ (z >> 13)) * 0xC2B2AE35u; return z ^ (z >> 16); }
    static uint32_t lea32(uint32_t z) { z = (z ^ (z >> 16)) * 0xD36D884Bu; z = (z ^ (z >> 16)) * 0xD36D884Bu; return z ^ (z >> 16); }
    static uint32_t rotl(uint32_t x, int k) { return (x << k) | (x >> (32 - k)); }
    explicit L32X64MixRandom(uint64_t seed) {
        seed ^= 0x6A09E667F3BCC909ULL;
        a = murmur32((uint32_t)(seed >> 32)) | 1u;
        s = 1;
        x0 = lea32((uint32_t)seed);
        x1 = lea32((uint32_t)seed + 0x9E3779B9u);
        if ((x0 | x1) == 0) { const uint32_t v = s + 0x9E3779B9u; x0 = murmur32(v); x1 = murmur32(v + 0x9E3779B9u); }
    }
    uint32_t nextInt() {
        const uint32_t r = lea32(s + x0);
        s = 0xADB4A92Du * s + a;
        uint32_t q0 = x0, q1 = x1;
        q1 ^= q0; q0 = rotl(q0, 26); q0 = q0 ^ q1 ^ (q1 << 9); q1 = rotl(q1, 13);
        x0 = q0; x1 = q1;
        return r;
    }
    float nextFloat() { return (float)(nextInt() >> 8) * (1.0f / (float)(1 << 24)); }      // RandomSupport.boundedNextFloat(rng, 1f)
};

static double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static std::vector<int32_t> parse_ids(const char* s) {
    std::vector<int32_t> v;
    for (const char* p = s; *p;) {
        char* e;
        const long x = strtol(p, &e, 10);
        if (e == p) break;
        v.push_back((int32_t)x);
        p = *e == ',' ? e + 1 : e;
    }
    return v;
}

#define CK(call)                                                                                        \
    do {                                                                                                \
        const int32_t r_ = (call);                                                                      \
        if (r_ != GL3_OK) { fprintf(stderr, "%s failed (%d): %s\n", #call, r_, ctx ? gl3_last_error(ctx) : gl3_gguf_last_error(nullptr)); return 1; } \
    } while (0)

int main(int argc, char** argv) {
    std::string path, protocol, rng_kind = "l32x64";
    std::vector<int32_t> prompt, stop;
    int max_tokens = 0, batch = 1, start_pos = 0, bos = -1;
    float temperature = 0.f, topp = 0.95f;
    uint64_t seed = 1234;                                  // the reference's default (Options.java)
    uint32_t flags = 0;
    for (int i = 1; i < argc; ++i) {
        const std::string a = argv[i];
        auto val = [&]() { return i + 1 < argc ? argv[++i] : (char*)""; };
        if (a == "-m") path = val();
        else if (a == "--ids") prompt = parse_ids(val());
        else if (a == "--protocol") protocol = val();
        else if (a == "--bos") bos = atoi(val());
        else if (a == "-n") max_tokens = atoi(val());
        else if (a == "-b") batch = atoi(val());
        else if (a == "--start-pos") start_pos = atoi(val());
        else if (a == "--temperature") temperature = (float)atof(val());
        else if (a == "--top-p") topp = (float)atof(val());
        else if (a == "--seed") seed = strtoull(val(), nullptr, 10);
        else if (a == "--rng") rng_kind = val();
        else if (a == "--stop") stop = parse_ids(val());
        else if (a == "--scalar-dot") flags |= GL3_FLAG_SCALAR_DOT;
        else if (a == "--f32-activation") flags |= GL3_FLAG_F32_ACTIVATION;
        else { fprintf(stderr, "gl3_run: unknown argument %s (see the header of tools/gl3_run.cpp)\n", a.c_str()); return 2; }
    }
    if (path.empty() || prompt.empty()) { fprintf(stderr, "gl3_run: -m model.gguf and --ids a,b,c are required\n"); return 2; }

    gl3_gguf* g = nullptr;
    gl3_model_desc d{};
    if (gl3_gguf_open(path.c_str(), &g) != GL3_OK || gl3_gguf_model_desc(g, &d, nullptr) != GL3_OK) { fprintf(stderr, "cannot read %s: %s\n", path.c_str(), gl3_gguf_last_error(g)); return 1; }
    // Qwen2.java:115, Qwen2MoE.java:98, Qwen3.java:98 all run generateTokensGPUQwen3; Llama / Mistral / Devstral generateTokensGPULlama
    if (protocol.empty()) protocol = (d.arch == GL3_ARCH_QWEN3 || d.arch == GL3_ARCH_QWEN2 || d.arch == GL3_ARCH_QWEN2MOE) ? "qwen3" : "llama";
    if (protocol != "llama" && protocol != "qwen3") { fprintf(stderr, "gl3_run: --protocol llama|qwen3\n"); return 2; }
    if (bos < 0) {
        double v = 0;
        bos = gl3_gguf_meta_number(g, "tokenizer.ggml.bos_token_id", &v) == GL3_OK ? (int)v : (d.vocab > 128000 ? 128000 : 1);   // <|begin_of_text|>
    }
    gl3_gguf_close(g);
    for (int32_t t : prompt) if (t < 0 || t >= d.vocab) { fprintf(stderr, "gl3_run: prompt id %d outside the vocabulary\n", t); return 2; }

    gl3_ctx* ctx = nullptr;
    gl3_model_desc opts{};
    opts.struct_size = sizeof(opts);
    opts.ctx = max_tokens > 0 ? max_tokens + 2 : 0;       // 0: the loader's default (min(context_length, 4096))
    opts.max_batch = batch;
    opts.tp_size = 1;
    opts.flags = flags;
    CK(gl3_load_gguf(path.c_str(), &opts, &ctx));
    gl3_model_desc live{};
    {   // the plan's context length bounds the loop like config.contextLength() does
        gl3_gguf* g2 = nullptr;
        if (gl3_gguf_open(path.c_str(), &g2) == GL3_OK) { live.ctx = opts.ctx; gl3_gguf_model_desc(g2, &live, nullptr); gl3_gguf_close(g2); }
    }
    const int ctx_len = live.ctx > 0 ? live.ctx : (opts.ctx > 0 ? opts.ctx : 4096);
    const int actual_max = max_tokens > 0 && max_tokens < ctx_len ? max_tokens : ctx_len;
    const std::set<int32_t> stop_set(stop.begin(), stop.end());
    JavaRandom lcg(seed);
    L32X64MixRandom mix(seed);
    auto sample = [&](int32_t token, int pos, int32_t* out) -> int32_t {      // forward + Sampler.sampleToken
        if (temperature == 0.f) return gl3_forward_decode(ctx, token, pos, nullptr, out);
        const float coin = rng_kind == "lcg" ? lcg.nextFloat() : mix.nextFloat();
        return gl3_forward_decode_sample(ctx, token, pos, temperature, topp, coin, out);
    };

    std::vector<int32_t> generated;
    const int N = (int)prompt.size();
    const double t_start = now_s();
    double t_decode = 0;
    int prompt_done = 0;
    if (protocol == "llama" && batch > 1) {
        // InferenceEngineWithBatchPrefillDecode: positions start .. start + N - 1 hold [latestToken, prompt[0 .. N-2]]
        std::vector<int32_t> seq((size_t)N);
        seq[0] = bos;
        for (int i = 1; i < N; ++i) seq[i] = prompt[i - 1];
        int pos = start_pos;
        for (int c0 = 0; c0 < N && pos + c0 < actual_max; c0 += batch) {
            int c1 = c0 + batch < N ? c0 + batch : N;
            if (c1 > actual_max - pos) c1 = actual_max - pos;
            CK(gl3_forward_prefill(ctx, seq.data() + c0, c1 - c0, pos + c0));
        }
        int32_t cur = prompt[N - 1];
        pos = start_pos + N;
        prompt_done = N;
        t_decode = now_s();
        while (pos < actual_max) {
            int32_t next = 0;
            CK(sample(cur, pos, &next));
            generated.push_back(next);
            if (stop_set.count(next)) break;
            cur = next;
            ++pos;
        }
    } else if (protocol == "llama") {
        // generateTokensGPULlama: forward first, then either take the next prompt token or sample
        int32_t cur = bos;
        int pos = start_pos, pi = 0;
        while (pos < actual_max) {
            int32_t next = 0;
            if (pi < N) {
                CK(gl3_forward_decode(ctx, cur, pos, nullptr, nullptr));       // logits unused while the prompt is ingested
                next = prompt[pi++];
            } else {
                if (t_decode == 0) t_decode = now_s();
                CK(sample(cur, pos, &next));
                generated.push_back(next);
                if (stop_set.count(next)) break;
            }
            cur = next;
            ++pos;
        }
        prompt_done = pi;
    } else {
        // generateTokensGPUQwen3 (note the position skipped after the last prompt token)
        int32_t cur = 0;
        int pi = 0;
        const int limit = max_tokens > 0 ? max_tokens : ctx_len;             // the reference loops to maxTokens here (:413)
        for (int position = start_pos; position < limit && position < ctx_len; ++position) {
            int32_t next = 0;
            if (pi < N) {
                const int32_t token = prompt[pi];
                ++pi;
                if (pi < N) { CK(gl3_forward_decode(ctx, token, position, nullptr, nullptr)); continue; }
                CK(sample(token, position, &next));                            // last prompt token: its logits give the first response token
                ++position;
            } else {
                if (t_decode == 0) t_decode = now_s();
                CK(sample(cur, position, &next));
            }
            generated.push_back(next);
            if (stop_set.count(next)) break;
            cur = next;
        }
        prompt_done = pi;
    }
    const double t_end = now_s();
    if (t_decode == 0) t_decode = t_end;
    printf("generated:");
    for (int32_t v : generated) printf(" %d", v);
    printf("\n");
    // RunMetrics.setInferenceMetrics(promptTokens, prefillNanos, generatedTokens, decodeNanos, totalNanos)
    fprintf(stderr, "protocol %s, bos %d, prompt tokens %d in %.1f ms (%.1f tok/s), generated %zu in %.1f ms (%.1f tok/s), total %.1f ms\n", protocol.c_str(), bos, prompt_done,
            (t_decode - t_start) * 1e3, prompt_done / (t_decode - t_start + 1e-12), generated.size(), (t_end - t_decode) * 1e3, generated.size() / (t_end - t_decode + 1e-12),
            (t_end - t_start) * 1e3);
    gl3_destroy(ctx);
    return 0;
}
