// gl3_gguf.cpp — native GGUF v2/v3 reader + weight upload (the on-disk format feeding every kernel).
// Replaces, for the HIP path, the Java host's loader chain
//   J/tensor/GGUF.java:43-92 (header, metadata KV, tensor infos), :105-137 (alignment, tensor data offset), :217-311 (value
//   types); J/model/loader/LlamaModelLoader.java:47-69 / Qwen3ModelLoader.java:48-79 (config keys), :83-98 (tensor names);
//   J/inference/operation/RoPE.java:6-37 (frequency table)
// with one mmap and one gl3_upload_tensor per tensor: no 16-byte-header private mapping (GGUF.java:157-194 is a TornadoVM
// device-buffer trick) and no 2 GiB per-tensor limit (GGMLType.java:70-74 Math.toIntExact).  Host-only C++.
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <cmath>
#include <cstdint>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "../../include/gpullama3_hip.h"

namespace {

enum { GT_U8 = 0, GT_I8, GT_U16, GT_I16, GT_U32, GT_I32, GT_F32, GT_BOOL, GT_STRING, GT_ARRAY, GT_U64, GT_I64, GT_F64 };

struct MetaValue {
    int type = -1;
    double num = 0;            // every scalar numeric / bool value
    std::string str;           // GT_STRING
    uint64_t array_len = 0;    // GT_ARRAY (elements are skipped, only the length is kept: vocabulary size)
    int array_type = -1;
    const uint8_t* array_data = nullptr;   // fixed-size element types: the elements inside the mapping
};

struct TensorInfo {
    std::string name;
    int n_dims = 0;
    uint64_t ne[4] = {1, 1, 1, 1};
    int type = 0;
    uint64_t offset = 0;       // relative to the tensor-data section
    uint64_t bytes = 0;
};

uint64_t type_bytes(int type, uint64_t n) {       // GGMLType.java:5-21 (type size / block size)
    switch (type) {
    case GL3_TYPE_F32: return n * 4;
    case GL3_TYPE_F16: return n * 2;
    case GL3_TYPE_Q4_0: return n / 32 * 18;
    case GL3_TYPE_Q8_0: return n / 32 * 34;
    case GL3_TYPE_Q4_K: return n / 256 * 144;
    case GL3_TYPE_Q5_K: return n / 256 * 176;
    case GL3_TYPE_Q6_K: return n / 256 * 210;
    default: return 0;
    }
}

}  // namespace

struct gl3_gguf {
    int fd = -1;
    const uint8_t* base = nullptr;
    size_t size = 0;
    uint32_t version = 0;
    uint64_t alignment = 32, data_off = 0;
    std::map<std::string, MetaValue> meta;
    std::vector<TensorInfo> tensors;
    std::map<std::string, int> by_name;
    std::string err;
};

namespace {

struct Cursor {
    const uint8_t* p; const uint8_t* end; bool ok = true;
    template <class T> T get() {
        T v{};
        if (p + sizeof(T) > end) { ok = false; return v; }
        memcpy(&v, p, sizeof(T)); p += sizeof(T);
        return v;
    }
    std::string str(bool v1_len32 = false) {
        const uint64_t n = v1_len32 ? get<uint32_t>() : get<uint64_t>();
        if (!ok || n > (uint64_t)(end - p)) { ok = false; return {}; }
        std::string s((const char*)p, (size_t)n); p += n;
        return s;
    }
};

bool read_scalar(Cursor& c, int ty, MetaValue& v) {
    switch (ty) {
    case GT_U8: v.num = c.get<uint8_t>(); break;
    case GT_I8: v.num = c.get<int8_t>(); break;
    case GT_U16: v.num = c.get<uint16_t>(); break;
    case GT_I16: v.num = c.get<int16_t>(); break;
    case GT_U32: v.num = c.get<uint32_t>(); break;
    case GT_I32: v.num = c.get<int32_t>(); break;
    case GT_F32: v.num = c.get<float>(); break;
    case GT_BOOL: v.num = c.get<uint8_t>() != 0; break;
    case GT_U64: v.num = (double)c.get<uint64_t>(); break;
    case GT_I64: v.num = (double)c.get<int64_t>(); break;
    case GT_F64: v.num = c.get<double>(); break;
    case GT_STRING: v.str = c.str(); break;
    default: return false;
    }
    return c.ok;
}

bool read_value(Cursor& c, int ty, MetaValue& v) {
    v.type = ty;
    if (ty != GT_ARRAY) return read_scalar(c, ty, v);
    v.array_type = (int)c.get<uint32_t>();
    v.array_len = c.get<uint64_t>();
    if (!c.ok || v.array_type == GT_ARRAY) return false;      // nested arrays do not occur in model files
    static const int fixed[] = {1, 1, 2, 2, 4, 4, 4, 1, 0, 0, 8, 8, 8};
    if (v.array_type == GT_STRING) {
        for (uint64_t i = 0; i < v.array_len && c.ok; ++i) {
            const uint64_t n = c.get<uint64_t>();
            if (!c.ok || n > (uint64_t)(c.end - c.p)) { c.ok = false; break; }
            c.p += n;
        }
    } else {
        if (v.array_type < 0 || v.array_type > GT_F64) return false;
        const uint64_t nb = v.array_len * fixed[v.array_type];
        if (nb > (uint64_t)(c.end - c.p)) return false;
        v.array_data = c.p;
        c.p += nb;
    }
    return c.ok;
}

int32_t fail(gl3_gguf* g, int32_t code, const std::string& m) { g->err = m; return code; }

// ---- K-quants: FloatTensor.getFloat of the reference's CPU tensors (f32 arithmetic, left to right) ----------------------
inline float f16_at(const uint8_t* p) { uint16_t h; memcpy(&h, p, 2); return (float)__builtin_bit_cast(_Float16, h); }

// Q4_KFloatTensor.getScaleK4 / getMinK4 (J/tensor/standard/Q4_KFloatTensor.java:63-83; shared by Q5_K :67-83)
inline int k4_scale(int j, const uint8_t* sc) { return j < 4 ? (sc[j] & 63) : ((sc[j + 4] & 0xF) | ((sc[j - 4] >> 6) << 4)); }
inline int k4_min(int j, const uint8_t* sc) { return j < 4 ? (sc[j + 4] & 63) : ((sc[j + 4] >> 4) | ((sc[j] >> 6) << 4)); }

// Q4_KFloatTensor.getFloat :86-114 (block = f16 d | f16 dmin | 12 scale bytes | 128 nibble bytes = 144 B per 256 elements)
inline float q4k_get(const uint8_t* base, size_t i) {
    const uint8_t* b = base + (i / 256) * 144;
    const int w = (int)(i % 256), pair = w / 64, pos = w % 64;
    const float d = f16_at(b), dmin = f16_at(b + 2);
    int sub, q;
    if (pos < 32) { sub = pair * 2; q = b[16 + pair * 32 + pos] & 0xF; }
    else { sub = pair * 2 + 1; q = (b[16 + pair * 32 + pos - 32] >> 4) & 0xF; }
    return d * (float)k4_scale(sub, b + 4) * (float)q - dmin * (float)k4_min(sub, b + 4);
}
// Q5_KFloatTensor.getFloat :86-120 (f16 d | f16 dmin | 12 scale bytes | 32 high-bit bytes | 128 nibble bytes = 176 B)
inline float q5k_get(const uint8_t* base, size_t i) {
    const uint8_t* b = base + (i / 256) * 176;
    const int w = (int)(i % 256), pair = w / 64, pos = w % 64;
    const float d = f16_at(b), dmin = f16_at(b + 2);
    int sub, q, hb;
    if (pos < 32) { sub = pair * 2; q = b[48 + pair * 32 + pos] & 0xF; hb = (b[16 + pos] >> (pair * 2)) & 1; }
    else { sub = pair * 2 + 1; q = (b[48 + pair * 32 + pos - 32] >> 4) & 0xF; hb = (b[16 + pos - 32] >> (pair * 2 + 1)) & 1; }
    q += hb * 16;
    return d * (float)k4_scale(sub, b + 4) * (float)q - dmin * (float)k4_min(sub, b + 4);
}
// Q6_KFloatTensor.getFloat (128 low-nibble bytes | 64 high-2-bit bytes | 16 int8 scales | f16 d = 210 B)
inline float q6k_get(const uint8_t* base, size_t i) {
    const uint8_t* b = base + (i / 256) * 210;
    const int w = (int)(i % 256), half = w / 128, ph = w % 128, grp = ph / 32, pg = ph % 32, is = pg / 16;
    const float d = f16_at(b + 208);
    const uint8_t* ql = b + half * 64;
    const uint8_t* qh = b + 128 + half * 32;
    const int8_t* sc = reinterpret_cast<const int8_t*>(b + 192 + half * 8);
    int qv;
    switch (grp) {
    case 0: qv = ((ql[pg] & 0xF) | (((qh[pg] >> 0) & 3) << 4)) - 32; return d * (float)sc[is] * (float)qv;
    case 1: qv = ((ql[32 + pg] & 0xF) | (((qh[pg] >> 2) & 3) << 4)) - 32; return d * (float)sc[is + 2] * (float)qv;
    case 2: qv = ((ql[pg] >> 4) | (((qh[pg] >> 4) & 3) << 4)) - 32; return d * (float)sc[is + 4] * (float)qv;
    default: qv = ((ql[32 + pg] >> 4) | (((qh[pg] >> 6) & 3) << 4)) - 32; return d * (float)sc[is + 6] * (float)qv;
    }
}

bool meta_num(const gl3_gguf* g, const std::string& key, double* out) {
    auto it = g->meta.find(key);
    if (it == g->meta.end() || it->second.type == GT_STRING || it->second.type == GT_ARRAY) return false;
    *out = it->second.num;
    return true;
}

thread_local std::string g_open_err;

}  // namespace

extern "C" {

// ModelLoader.dequantizeToQ8_0TornadoTensor (J/model/loader/ModelLoader.java:173-224): what the reference's GPU path does with
// Q4_K / Q5_K / Q6_K tensors at load time — dequantise element by element with the CPU tensor's getFloat and re-quantise to
// Q8_0 blocks: scale = maxAbs / 127 (stored as f16, RNE), q = clamp(Math.round(x * (1 / scale)), -128, 127), with the
// UNROUNDED scale in the reciprocal.  n = number of elements (multiple of 256); dst receives n / 32 * 34 bytes.
int32_t gl3_kquant_to_q8_0(int32_t src_type, const void* src, uint64_t n, void* dst) {
    if (!src || !dst || (n % 256)) return GL3_E_ARG;
    if (src_type != GL3_TYPE_Q4_K && src_type != GL3_TYPE_Q5_K && src_type != GL3_TYPE_Q6_K) return GL3_E_UNSUPPORTED;
    const uint8_t* s = (const uint8_t*)src;
    uint8_t* out = (uint8_t*)dst;
    const uint64_t nblk = n / 32;
#pragma omp parallel for schedule(static)
    for (long long b = 0; b < (long long)nblk; ++b) {
        float x[32];
        float max_abs = 0.f;
        for (int i = 0; i < 32; ++i) {
            const size_t e = (size_t)b * 32 + i;
            x[i] = src_type == GL3_TYPE_Q4_K ? q4k_get(s, e) : src_type == GL3_TYPE_Q5_K ? q5k_get(s, e) : q6k_get(s, e);
            max_abs = std::fmax(max_abs, std::fabs(x[i]));
        }
        const float scale = max_abs / 127.0f;
        const uint16_t h = __builtin_bit_cast(uint16_t, (_Float16)scale);          // Float.floatToFloat16: round to nearest even
        uint8_t* o = out + (size_t)b * 34;
        o[0] = (uint8_t)(h & 0xFF); o[1] = (uint8_t)(h >> 8);
        const float inv = scale != 0.f ? 1.0f / scale : 0.f;
        for (int i = 0; i < 32; ++i) {
            // Math.round(float) = floor(a + 1/2) with the sum taken EXACTLY (ties toward +infinity): the f32 product is widened
            // before the addition (a = 0.49999997f: a + 0.5f rounds to 1.0f in f32, Java gives 0)
            int q = (int)std::floor((double)(x[i] * inv) + 0.5);
            q = q < -128 ? -128 : q > 127 ? 127 : q;
            o[2 + i] = (uint8_t)(int8_t)q;
        }
    }
    return GL3_OK;
}

const char* gl3_gguf_last_error(const gl3_gguf* g) { return g ? g->err.c_str() : g_open_err.c_str(); }

void gl3_gguf_close(gl3_gguf* g) {
    if (!g) return;
    if (g->base) munmap((void*)g->base, g->size);
    if (g->fd >= 0) close(g->fd);
    delete g;
}

int32_t gl3_gguf_open(const char* path, gl3_gguf** out) {
    if (!path || !out) return GL3_E_ARG;
    *out = nullptr;
    gl3_gguf* g = new gl3_gguf();
    auto bail = [&](int32_t code, const std::string& m) { g_open_err = m; gl3_gguf_close(g); return code; };
    g->fd = open(path, O_RDONLY);
    if (g->fd < 0) return bail(GL3_E_ARG, std::string("cannot open ") + path);
    struct stat st;
    if (fstat(g->fd, &st) != 0 || st.st_size < 24) return bail(GL3_E_ARG, "not a GGUF file (too short)");
    g->size = (size_t)st.st_size;
    void* m = mmap(nullptr, g->size, PROT_READ, MAP_PRIVATE, g->fd, 0);
    if (m == MAP_FAILED) return bail(GL3_E_OOM, "mmap failed");
    g->base = (const uint8_t*)m;
    Cursor c{g->base, g->base + g->size};
    if (c.get<uint32_t>() != 0x46554747u) return bail(GL3_E_ARG, "bad magic: not a GGUF file");    // "GGUF" (GGUF.java:43)
    g->version = c.get<uint32_t>();
    if (g->version != 2 && g->version != 3) return bail(GL3_E_UNSUPPORTED, "unsupported GGUF version " + std::to_string(g->version));
    const uint64_t n_tensors = c.get<uint64_t>(), n_kv = c.get<uint64_t>();
    if (!c.ok || n_tensors > (1u << 20) || n_kv > (1u << 20)) return bail(GL3_E_ARG, "corrupt GGUF header");
    for (uint64_t i = 0; i < n_kv; ++i) {
        std::string key = c.str();
        const int ty = (int)c.get<uint32_t>();
        MetaValue v;
        if (!c.ok || !read_value(c, ty, v)) return bail(GL3_E_ARG, "corrupt GGUF metadata near key '" + key + "'");
        g->meta[key] = v;
    }
    double al;
    if (meta_num(g, "general.alignment", &al)) {
        // a power of two between 1 and 1 MiB (GGUF default 32); anything else is a corrupt or hostile file
        if (!(al >= 1 && al <= 1048576.0) || ((uint64_t)al & ((uint64_t)al - 1))) return bail(GL3_E_ARG, "general.alignment is not a sane power of two");
        g->alignment = (uint64_t)al;
    }
    g->tensors.resize((size_t)n_tensors);
    for (auto& t : g->tensors) {
        t.name = c.str();
        t.n_dims = (int)c.get<uint32_t>();
        if (!c.ok || t.n_dims < 1 || t.n_dims > 4) return bail(GL3_E_ARG, "corrupt GGUF tensor info");
        uint64_t n = 1;
        bool overflow = false;
        for (int d = 0; d < t.n_dims; ++d) {
            t.ne[d] = c.get<uint64_t>();
            if (t.ne[d] != 0 && n > (UINT64_MAX / 8) / t.ne[d]) overflow = true;      // n * 4 bytes must not wrap either
            else n *= t.ne[d];
        }
        t.type = (int)c.get<uint32_t>();
        t.offset = c.get<uint64_t>();
        if (!c.ok || overflow) return bail(GL3_E_ARG, "corrupt GGUF tensor info");
        t.bytes = type_bytes(t.type, n);
    }
    const uint64_t pos = (uint64_t)(c.p - g->base);
    g->data_off = (pos + g->alignment - 1) / g->alignment * g->alignment;        // GGUF.java:105-137
    if (g->data_off > g->size) return bail(GL3_E_ARG, "GGUF tensor-data section starts past the end of the file");
    const uint64_t data_size = g->size - g->data_off;
    for (size_t i = 0; i < g->tensors.size(); ++i) {
        const TensorInfo& t = g->tensors[i];
        // overflow-safe: offset and offset + bytes are compared against the remaining size, never added to data_off
        if (t.bytes && (t.offset > data_size || t.bytes > data_size - t.offset || t.offset % g->alignment))
            return bail(GL3_E_ARG, "tensor '" + t.name + "' lies outside the file or is misaligned");
        g->by_name[t.name] = (int)i;
    }
    *out = g;
    return GL3_OK;
}

int32_t gl3_gguf_tensor_count(const gl3_gguf* g) { return g ? (int32_t)g->tensors.size() : 0; }

int32_t gl3_gguf_tensor_info(const gl3_gguf* g, int32_t i, const char** name, int32_t* type, uint64_t* ne /*[4]*/, const void** data,
                             uint64_t* bytes) {
    if (!g || i < 0 || i >= (int32_t)g->tensors.size()) return GL3_E_ARG;
    const TensorInfo& t = g->tensors[i];
    if (name) *name = t.name.c_str();
    if (type) *type = t.type;
    if (ne) for (int d = 0; d < 4; ++d) ne[d] = t.ne[d];
    if (data) *data = g->base + g->data_off + t.offset;
    if (bytes) *bytes = t.bytes;
    return GL3_OK;
}

int32_t gl3_gguf_meta_number(const gl3_gguf* g, const char* key, double* out) {
    if (!g || !key || !out) return GL3_E_ARG;
    return meta_num(g, key, out) ? GL3_OK : GL3_E_ARG;
}

int32_t gl3_gguf_meta_string(const gl3_gguf* g, const char* key, const char** out) {
    if (!g || !key || !out) return GL3_E_ARG;
    auto it = g->meta.find(key);
    if (it == g->meta.end() || it->second.type != GT_STRING) return GL3_E_ARG;
    *out = it->second.str.c_str();
    return GL3_OK;
}

// Fills the shape fields of desc from the metadata (LlamaModelLoader.java:47-63, Qwen3ModelLoader.java:48-74);
// max_batch / device / tp_* / flags / n_seqs are left as the caller set them.  rope_theta is returned separately.
int32_t gl3_gguf_model_desc(gl3_gguf* g, gl3_model_desc* d, float* rope_theta) {
    if (!g || !d) return GL3_E_ARG;
    auto it = g->meta.find("general.architecture");
    if (it == g->meta.end()) return fail(g, GL3_E_ARG, "general.architecture missing");
    const std::string a = it->second.str;
    if (a == "llama") d->arch = GL3_ARCH_LLAMA;
    // Devstral 2 (DevstralModelLoader.java:45-70, metadata prefix "mistral3"): the Llama graph with an independent head
    // dimension (attention.key_length; forwardJavaDevstral InferenceCore.java:178-261) and a YaRN RoPE table
    else if (a == "mistral3") d->arch = GL3_ARCH_LLAMA;
    else if (a == "qwen3") d->arch = GL3_ARCH_QWEN3;
    else if (a == "qwen2") d->arch = GL3_ARCH_QWEN2;
    else if (a == "granite") d->arch = GL3_ARCH_GRANITE;
    else if (a == "phi3") d->arch = GL3_ARCH_PHI3;
    else if (a == "qwen2moe") d->arch = GL3_ARCH_QWEN2MOE;          // ModelLoader.detectModelType :50-52: the architecture key decides
    else return fail(g, GL3_E_UNSUPPORTED, "architecture '" + a + "' is not implemented (llama, mistral3, qwen3, qwen2, qwen2moe, granite, phi3)");
    auto need = [&](const char* k, double* v) { return meta_num(g, a + "." + k, v); };
    // defaults as the reference loaders: rms epsilon 1e-5, rope theta 10000 (LlamaModelLoader.java:62-63)
    double dim, hid, nl, nh, nkv, eps = 1e-5, theta = 10000.0, ctx, kl;
    if (!need("embedding_length", &dim) || !need("feed_forward_length", &hid) || !need("block_count", &nl) ||
        !need("attention.head_count", &nh) || !need("context_length", &ctx))
        return fail(g, GL3_E_ARG, "model shape keys missing from the metadata");
    need("attention.layer_norm_rms_epsilon", &eps);
    if (!need("attention.head_count_kv", &nkv)) {
        // Granite 4.0 stores a per-layer int array; GraniteLoader.java:61-71 takes element 0 ("assuming uniform").  A uniform array
        // is accepted the same way; a non-uniform one would give a wrong kv_dim for some layer, so it is refused.
        nkv = nh;
        auto kv = g->meta.find(a + ".attention.head_count_kv");
        if (kv != g->meta.end() && kv->second.type == GT_ARRAY && kv->second.array_data && kv->second.array_len > 0 &&
            (kv->second.array_type == GT_U32 || kv->second.array_type == GT_I32)) {
            int32_t first = 0;
            memcpy(&first, kv->second.array_data, 4);
            for (uint64_t i = 1; i < kv->second.array_len; ++i) {
                int32_t e = 0;
                memcpy(&e, kv->second.array_data + 4 * i, 4);
                if (e != first) return fail(g, GL3_E_UNSUPPORTED, "attention.head_count_kv differs between layers (per-layer kv heads are not implemented)");
            }
            nkv = first;
        }
    }
    need("rope.freq_base", &theta);
    if (!(dim >= 1 && dim <= (1 << 20)) || !(hid >= 1 && hid <= (1 << 24)) || !(nl >= 1 && nl <= 4096) || !(nh >= 1 && nh <= 4096) ||
        !(nkv >= 1 && nkv <= nh) || !(ctx >= 1 && ctx <= 2147483647.0))
        return fail(g, GL3_E_ARG, "model shape keys out of range (head_count / block_count / lengths must be positive)");
    auto te = g->by_name.find("token_embd.weight");
    if (te == g->by_name.end()) return fail(g, GL3_E_ARG, "token_embd.weight missing");
    const TensorInfo& emb = g->tensors[te->second];
    d->struct_size = sizeof(gl3_model_desc);
    d->dim = (int32_t)dim; d->hidden = (int32_t)hid; d->n_layers = (int32_t)nl; d->n_heads = (int32_t)nh; d->n_kv_heads = (int32_t)nkv;
    d->head_size = need("attention.key_length", &kl) ? (int32_t)kl : d->dim / d->n_heads;
    d->vocab = (int32_t)emb.ne[1];
    // The caller may ask for a shorter KV cache.  ctx = 0 means "default": min(context_length, 4096) — the reference always
    // clamps through Configuration.withContextLength(maxTokens); <arch>.context_length itself (131072 for Llama-3.1/3.2)
    // would allocate tens of GB of f32 KV cache and is beyond the decode attention kernel's limit.
    if (d->ctx <= 0) d->ctx = ctx < 4096 ? (int32_t)ctx : 4096;
    else if (d->ctx > (int32_t)ctx) d->ctx = (int32_t)ctx;
    d->rms_eps = (float)eps;
    if (d->arch == GL3_ARCH_GRANITE) {          // GraniteLoader.java:55-58 (same defaults)
        double es = 12.0, rs = 0.22, as = 0.0078125, ls = 16.0;
        need("embedding_scale", &es); need("residual_scale", &rs); need("attention.scale", &as); need("logit_scale", &ls);
        d->embedding_scale = (float)es; d->residual_scale = (float)rs; d->attention_scale = (float)as; d->logit_scale = (float)ls;
    }
    d->n_experts = d->n_experts_used = d->moe_hidden = 0;
    if (d->arch == GL3_ARCH_QWEN2MOE) {         // Qwen2MoEModelLoader.java:56-84
        double ne, nu;
        if (!need("expert_count", &ne) || !need("expert_used_count", &nu)) return fail(g, GL3_E_ARG, "qwen2moe: expert_count / expert_used_count missing");
        auto de = g->by_name.find("blk.0.ffn_down_exps.weight");
        if (de == g->by_name.end() || g->tensors[de->second].n_dims != 3) return fail(g, GL3_E_ARG, "qwen2moe: blk.0.ffn_down_exps.weight missing or not 3-D");
        const uint64_t mh = g->tensors[de->second].ne[0];
        if (!(ne >= 1 && ne <= 4096) || !(nu >= 1 && nu <= ne) || mh < 32 || mh > (1u << 24))       // range-checked BEFORE the int32 casts (untrusted file)
            return fail(g, GL3_E_ARG, "qwen2moe: expert_count / expert_used_count / expert hidden size out of range");
        d->n_experts = (int32_t)ne; d->n_experts_used = (int32_t)nu;
        d->moe_hidden = (int32_t)mh;                  // dimensions()[0] of the down stack = experts' hidden size
    }
    // K-quant files run as Q8_0 after the load-time conversion (ModelLoader.loadTornadoTensor :163-164)
    d->weight_type = (emb.type == GL3_TYPE_Q4_K || emb.type == GL3_TYPE_Q5_K || emb.type == GL3_TYPE_Q6_K) ? GL3_TYPE_Q8_0 : emb.type;
    if (rope_theta) *rope_theta = (float)theta;
    return GL3_OK;
}

// RoPE.precomputeFreqsCis (J/inference/operation/RoPE.java:6-37, ropeScaling = false): freq = (float)(1 / pow(theta, i / hs))
// in double, val = pos * freq in f32, cos / sin evaluated in double and cast.  cr / ci: f32[ctx * hs/2].
void gl3_rope_table(int32_t ctx, int32_t head_size, float theta, float* cr, float* ci) {
    const int half = head_size / 2;
    for (int pos = 0; pos < ctx; ++pos)
        for (int i = 0; i < half; ++i) {
            const float freq = (float)(1.0 / pow((double)theta, (double)(2 * i) / (double)head_size));
            const float val = (float)pos * freq;
            cr[(size_t)pos * half + i] = (float)cos((double)val);
            ci[(size_t)pos * half + i] = (float)sin((double)val);
        }
}

// RoPE.precomputeFreqsCisYaRN (J/inference/operation/RoPE.java:39-83, Devstral 2): per pair i0 = i/2 the frequency is a ramp
// between the extrapolated (plain) and the interpolated (/ factor) one, and cos / sin carry the attention scale mscale.
// Every intermediate is rounded to f32 where the reference's is a Java float.
void gl3_rope_table_yarn(int32_t ctx, int32_t head_size, float theta, float factor, float beta_fast, float beta_slow,
                         float log_multiplier, int32_t original_ctx, float* cr, float* ci) {
    const int half = head_size / 2;
    const float freq_scale = 1.0f / factor;
    auto corr_dim = [&](float n_rot) {           // yarnCorrDim :76-78
        const float ratio = (float)original_ctx / (n_rot * 2.0f * (float)M_PI);
        return (float)head_size * (float)log((double)ratio) / (2.0f * (float)log((double)theta));
    };
    const float low = corr_dim(beta_fast), high = corr_dim(beta_slow);
    const float mscale = log_multiplier > 0 ? 1.0f + 0.1f * log_multiplier * (float)log((double)(1.0f / freq_scale)) : 1.0f;
    for (int pos = 0; pos < ctx; ++pos)
        for (int i = 0; i < half; ++i) {
            const float extrap = (float)(1.0 / pow((double)theta, (double)(2 * i) / (double)head_size));
            const float interp = freq_scale * extrap;
            const float span = high - low;
            const float y = ((float)i - low) / (0.001f > span ? 0.001f : span);                  // yarnRamp :80-83
            const float ramp = 1.0f - fminf(1.0f, fmaxf(0.0f, y));
            const float freq = interp * (1.0f - ramp) + extrap * ramp;
            const float val = (float)pos * freq;
            cr[(size_t)pos * half + i] = (float)cos((double)val) * mscale;
            ci[(size_t)pos * half + i] = (float)sin((double)val) * mscale;
        }
}

// <arch>.rope.scaling.* of a YaRN file (DevstralModelLoader.java:80-86).  Returns 1 and fills the five parameters when the file's
// architecture is "mistral3", rope.scaling.type == "yarn" and the required keys are present and usable; 0 when the plain table
// applies.  ONLY the Devstral loader of the reference reads these keys: the llama / qwen2 / qwen3 / ... loaders ignore them and
// build the plain table even for a long-context file that carries yarn metadata, so the same is done here (r4 advisor finding).
// factor <= 0 or original_context_length <= 0 would put inf / NaN into the table: such a file gets -1 (rejected by gl3_load_gguf).
int32_t gl3_gguf_yarn_params(gl3_gguf* g, float* factor, float* beta_fast, float* beta_slow, float* log_multiplier, int32_t* original_ctx) {
    if (!g) return 0;
    auto it = g->meta.find("general.architecture");
    if (it == g->meta.end()) return 0;
    const std::string a = it->second.str;
    if (a != "mistral3") return 0;
    auto ty = g->meta.find(a + ".rope.scaling.type");
    if (ty == g->meta.end() || ty->second.type != GT_STRING || ty->second.str != "yarn") return 0;
    double f, bf, bs, lm = 0.0, oc;
    if (!meta_num(g, a + ".rope.scaling.factor", &f) || !meta_num(g, a + ".rope.scaling.yarn_beta_fast", &bf) ||
        !meta_num(g, a + ".rope.scaling.yarn_beta_slow", &bs) || !meta_num(g, a + ".rope.scaling.original_context_length", &oc))
        return 0;
    meta_num(g, a + ".rope.scaling.yarn_log_multiplier", &lm);
    if (!(f > 0.0) || !(oc >= 1.0) || oc > 2147483647.0 || !std::isfinite(f) || !std::isfinite(bf) || !std::isfinite(bs) || !std::isfinite(lm)) return -1;
    if (factor) *factor = (float)f;
    if (beta_fast) *beta_fast = (float)bf;
    if (beta_slow) *beta_slow = (float)bs;
    if (log_multiplier) *log_multiplier = (float)lm;
    if (original_ctx) *original_ctx = (int32_t)oc;
    return 1;
}

// Opens the file, builds the plan and uploads every tensor straight from the mapping (tensor names as
// LlamaModelLoader.java:83-98).  opts (nullable) supplies ctx / max_batch / device / tp_* / flags / n_seqs; shape fields
// are overwritten from the metadata.  Tensor-parallel plans are returned un-finalized when tp_size > 1 (the caller must
// gl3_tp_init / gl3_tp_attach_local first); otherwise the plan is finalized.
int32_t gl3_load_gguf(const char* path, const gl3_model_desc* opts, gl3_ctx** out) {
    if (!out) return GL3_E_ARG;
    *out = nullptr;
    gl3_gguf* g = nullptr;
    int32_t r = gl3_gguf_open(path, &g);
    if (r != GL3_OK) return r;
    gl3_model_desc d{};
    if (opts) d = *opts;
    float theta = 10000.f;
    if ((r = gl3_gguf_model_desc(g, &d, &theta)) != GL3_OK) { g_open_err = g->err; gl3_gguf_close(g); return r; }
    float yf = 0, ybf = 0, ybs = 0, ylm = 0;
    int32_t yoc = 0;
    const int32_t yarn = gl3_gguf_yarn_params(g, &yf, &ybf, &ybs, &ylm, &yoc);
    if (yarn < 0) { g_open_err = "mistral3.rope.scaling: factor and original_context_length must be finite and > 0"; gl3_gguf_close(g); return GL3_E_ARG; }
    gl3_ctx* ctx = nullptr;
    if ((r = gl3_create(&d, &ctx)) != GL3_OK) { g_open_err = gl3_last_error(nullptr); gl3_gguf_close(g); return r; }
    std::vector<uint8_t> kq;          // Q8_0 image of the K-quant tensor being uploaded
    auto up = [&](const std::string& name, int id, int layer, bool required) -> int32_t {
        auto it = g->by_name.find(name);
        if (it == g->by_name.end()) return required ? GL3_E_STATE : GL3_OK;
        const TensorInfo& t = g->tensors[it->second];
        const uint8_t* data = g->base + g->data_off + t.offset;
        if (t.type == GL3_TYPE_Q4_K || t.type == GL3_TYPE_Q5_K || t.type == GL3_TYPE_Q6_K) {     // K-quant -> Q8_0 at load
            uint64_t n = 1;
            for (int dd = 0; dd < t.n_dims; ++dd) n *= t.ne[dd];
            if (n % 256) return GL3_E_ARG;
            kq.resize((size_t)(n / 32 * 34));
            int32_t rr = gl3_kquant_to_q8_0(t.type, data, n, kq.data());
            if (rr != GL3_OK) return rr;
            return gl3_upload_tensor(ctx, id, layer, kq.data(), kq.size(), GL3_TYPE_Q8_0);
        }
        return gl3_upload_tensor(ctx, id, layer, data, t.bytes, t.type);
    };
    r = up("token_embd.weight", GL3_T_TOKEN_EMBD, 0, true);
    if (r == GL3_OK) r = up("output_norm.weight", GL3_T_OUTPUT_NORM, 0, true);
    if (r == GL3_OK) r = up("output.weight", GL3_T_OUTPUT, 0, false);                       // absent: tied embeddings
    static const struct { const char* name; int id; int arch_only; } per_layer[] = {     // arch_only: -1 = every architecture
        {"attn_norm.weight", GL3_T_ATTN_NORM, -1}, {"attn_q.weight", GL3_T_WQ, -1}, {"attn_k.weight", GL3_T_WK, -1},
        {"attn_v.weight", GL3_T_WV, -1}, {"attn_output.weight", GL3_T_WO, -1}, {"ffn_norm.weight", GL3_T_FFN_NORM, -1},
        {"ffn_gate.weight", GL3_T_W1, -1}, {"ffn_down.weight", GL3_T_W2, -1}, {"ffn_up.weight", GL3_T_W3, -1},
        {"attn_q_norm.weight", GL3_T_ATTN_Q_NORM, GL3_ARCH_QWEN3}, {"attn_k_norm.weight", GL3_T_ATTN_K_NORM, GL3_ARCH_QWEN3},
        {"attn_q.bias", GL3_T_BQ, GL3_ARCH_QWEN2}, {"attn_k.bias", GL3_T_BK, GL3_ARCH_QWEN2}, {"attn_v.bias", GL3_T_BV, GL3_ARCH_QWEN2}};
    // Qwen2-MoE (Qwen2MoEModelLoader.java:86-110): qwen2 attention tensors, no dense FFN; the shared expert takes the W1 / W3 / W2 slots
    static const struct { const char* name; int id; } moe_layer[] = {
        {"attn_norm.weight", GL3_T_ATTN_NORM}, {"attn_q.weight", GL3_T_WQ}, {"attn_k.weight", GL3_T_WK}, {"attn_v.weight", GL3_T_WV},
        {"attn_q.bias", GL3_T_BQ}, {"attn_k.bias", GL3_T_BK}, {"attn_v.bias", GL3_T_BV}, {"attn_output.weight", GL3_T_WO},
        {"ffn_norm.weight", GL3_T_FFN_NORM}, {"ffn_gate_inp.weight", GL3_T_FFN_GATE_INP}, {"ffn_gate_exps.weight", GL3_T_FFN_GATE_EXPS},
        {"ffn_up_exps.weight", GL3_T_FFN_UP_EXPS}, {"ffn_down_exps.weight", GL3_T_FFN_DOWN_EXPS}, {"ffn_gate_shexp.weight", GL3_T_W1},
        {"ffn_up_shexp.weight", GL3_T_W3}, {"ffn_down_shexp.weight", GL3_T_W2}, {"ffn_gate_inp_shexp.weight", GL3_T_FFN_GATE_INP_SHEXP}};
    if (d.arch == GL3_ARCH_QWEN2MOE) {
        for (int l = 0; l < d.n_layers && r == GL3_OK; ++l)
            for (const auto& t : moe_layer) {
                r = up("blk." + std::to_string(l) + "." + t.name, t.id, l, true);
                if (r != GL3_OK) break;
            }
    }
    // Phi-3 (Phi3ModelLoader.java:111-116): attn_qkv and ffn_up (= gate | up) are fused tensors
    static const struct { const char* name; int id; } phi3_layer[] = {
        {"attn_norm.weight", GL3_T_ATTN_NORM}, {"attn_qkv.weight", GL3_T_WQKV}, {"attn_output.weight", GL3_T_WO},
        {"ffn_norm.weight", GL3_T_FFN_NORM}, {"ffn_up.weight", GL3_T_W13}, {"ffn_down.weight", GL3_T_W2}};
    if (d.arch == GL3_ARCH_PHI3) {
        for (int l = 0; l < d.n_layers && r == GL3_OK; ++l)
            for (const auto& t : phi3_layer) {
                r = up("blk." + std::to_string(l) + "." + t.name, t.id, l, true);
                if (r != GL3_OK) break;
            }
    }
    for (int l = 0; l < d.n_layers && r == GL3_OK && d.arch != GL3_ARCH_PHI3 && d.arch != GL3_ARCH_QWEN2MOE; ++l)
        for (const auto& t : per_layer) {
            if (t.arch_only >= 0 && t.arch_only != d.arch) continue;
            r = up("blk." + std::to_string(l) + "." + t.name, t.id, l, true);
            if (r != GL3_OK) break;
        }
    if (r == GL3_OK) {
        const size_t n = (size_t)d.ctx * (d.head_size / 2);
        std::vector<float> cr(n), ci(n);
        if (yarn > 0) gl3_rope_table_yarn(d.ctx, d.head_size, theta, yf, ybf, ybs, ylm, yoc, cr.data(), ci.data());
        else gl3_rope_table(d.ctx, d.head_size, theta, cr.data(), ci.data());
        r = gl3_upload_rope(ctx, cr.data(), ci.data(), n);
    }
    if (r == GL3_OK && d.tp_size <= 1 && !(d.flags & GL3_FLAG_FORCE_RCCL)) r = gl3_finalize(ctx);
    if (r != GL3_OK) {
        g_open_err = r == GL3_E_STATE ? std::string("a required tensor is missing from the GGUF file") : std::string(gl3_last_error(ctx));
        gl3_destroy(ctx);
        gl3_gguf_close(g);
        return r;
    }
    gl3_gguf_close(g);        // weights are resident in HBM; the mapping is no longer needed
    *out = ctx;
    return GL3_OK;
}

}  // extern "C"
