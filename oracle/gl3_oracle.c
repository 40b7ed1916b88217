/*
 * gl3_oracle.c — CPU restatement of GPULlama3.java's pure-Java forward pass.
 *
 * TEST INFRASTRUCTURE ONLY.  This file is the parity oracle and the timed CPU
 * baseline ("port") for the HIP path.  Nothing in the product path
 * (gpullama3.java_amd/, include/, libgpullama_hip.so) links, imports or calls
 * it; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg do.
 *
 * PARITY UNPINNED: the reference ships no golden vectors, known-answer tests
 * or fixtures for the forward pass and cannot be compiled
 * here (no JDK).  This restatement is pinned instead by (1) hand-derived KATs
 * (tests/test_oracle_kat.py) and (2) bit-for-bit agreement with an
 * independent NumPy restatement (oracle/oracle_np.py).
 *
 * All citations are relative to /root/reference/src/main/java/org/beehive/gpullama3/
 * (abbreviated J/).  Java numeric rules honoured here:
 *   - float ops round to binary32 at every step, no FMA contraction
 *     (compile with -ffp-contract=off, never -ffast-math);
 *   - Math.sqrt/exp/pow/cos/sin are evaluated in double and then cast;
 *   - Float.float16ToFloat / floatToFloat16 are IEEE (RNE, subnormals kept);
 *   - (int) of a float truncates toward zero.
 * Threading mirrors Parallel.parallelFor (J/auxiliary/Parallel.java:9-11):
 * matmul rows and attention heads are fanned out, no cross-thread reductions,
 * so results are independent of the thread count.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define ORC_API __attribute__((visibility("default")))

/* ggml type ids used by the wire format — J/tensor/GGMLType.java:5-21 */
enum { ORC_F32 = 0, ORC_F16 = 1, ORC_Q4_0 = 2, ORC_Q8_0 = 8 };

/* tensor ids: same numbering as include/gpullama3_hip.h (gl3_tensor_id) */
enum {
    ORC_T_TOKEN_EMBD = 0, ORC_T_OUTPUT_NORM = 1, ORC_T_OUTPUT = 2,
    ORC_T_ATTN_NORM = 3, ORC_T_WQ = 4, ORC_T_WK = 5, ORC_T_WV = 6, ORC_T_WO = 7,
    ORC_T_FFN_NORM = 8, ORC_T_W1 = 9, ORC_T_W2 = 10, ORC_T_W3 = 11,
    ORC_T_ATTN_Q_NORM = 12, ORC_T_ATTN_K_NORM = 13,
    ORC_T_BQ = 14, ORC_T_BK = 15, ORC_T_BV = 16,       /* qwen2: blk.L.attn_{q,k,v}.bias, F32 */
    ORC_T_WQKV = 17, ORC_T_W13 = 18,                   /* phi3 fused tensors: not used here (row views are handed over instead) */
    /* qwen2moe (Qwen2MoEModelLoader.java:97-105): router, stacked routed experts, shared-expert gate; the shared expert's
     * gate / up / down matrices are W1 / W3 / W2 with hidden = sharedExpertHiddenDim */
    ORC_T_GATE_INP = 19,        /* blk.L.ffn_gate_inp.weight        [n_experts x dim]              F32 */
    ORC_T_GATE_EXPS = 20,       /* blk.L.ffn_gate_exps.weight       [n_experts x moe_hidden x dim]     */
    ORC_T_UP_EXPS = 21,         /* blk.L.ffn_up_exps.weight         [n_experts x moe_hidden x dim]     */
    ORC_T_DOWN_EXPS = 22,       /* blk.L.ffn_down_exps.weight       [n_experts x dim x moe_hidden]     */
    ORC_T_GATE_INP_SHEXP = 23,  /* blk.L.ffn_gate_inp_shexp.weight  [dim]                          F32 */
    ORC_T_COUNT = 24
};

typedef struct {
    int32_t arch;       /* 0 = llama (InferenceCore.forwardJava), 1 = qwen3 (forwardJavaQwen3), 2 = qwen2 (forwardJavaQwen2),
                           3 = granite (forwardGranite :814-924: the llama graph + the four scalars below),
                           4 = phi3 (forwardJavaPhi3 :699-800: NeoX RoPE pairs, no bias, no per-head norm; the fused attn_qkv / gate|up
                               tensors are handed to this oracle as row views wq | wk | wv and w1 | w3 of the same bytes),
                           5 = qwen2moe (forwardJavaQwen2MoE :263-422: the qwen2 attention + router / top-k routed experts /
                               gated shared expert instead of the dense FFN) */
    int32_t dim, hidden, n_layers, n_heads, n_kv_heads, head_size, vocab, ctx;
    float   rms_eps;
    float   embedding_scale, attention_scale, residual_scale, logit_scale;   /* granite only (GraniteLoader.java:55-58) */
    int32_t n_experts, n_experts_used, moe_hidden;   /* qwen2moe only (Qwen2MoEModelLoader.java:56-84); hidden = sharedExpertHiddenDim */
} orc_config;

typedef struct { const void* p; int type; } orc_tensor;

typedef struct {
    orc_config c;
    int vector_bits;                      /* 0: scalar dots (-Dllama.VectorBitSize=0); 128 / 256 / 512: the Vector-API dots of F16 / Q4_0 (and Q8_0 with f32 activation) of that species */
    int species_error;                    /* set by a matmul the reference throws UnsupportedOperationException for (Q4_0 / Q8_0-f32act on a 512-bit species) */
    int f32_activation;                   /* 1: -Dllama.quantizeActivation=false — Q8_0 matrices take the f32 activation (vectorDot / scalarDot) */
    orc_tensor global[3];
    orc_tensor* layer[ORC_T_COUNT];       /* per-layer tensors, [id][layer] */
    const float *rope_cr, *rope_ci;       /* freq_cis_real / freq_cis_imag, [ctx * head_size/2] */
    /* State — J/inference/state/LlamaState.java:28-81 */
    float *x, *xb, *xb2, *q, *k, *v, *hb, *hb2, *att, *logits;
    float *key_cache, *value_cache;       /* [L][ctx][kvDim] */
    int8_t* aq; float* ascale;            /* hoisted activation quantisation scratch */
    int q_dim, kv_dim;
    float *router, *hbe, *hbe2, *ytmp;    /* Qwen2MoEState.java:15-32: routerLogits, hbE, hbE2, yTmp */
    int moe_sel[64]; float moe_w[64];     /* experts selected by the last layer of the last step and their routing weights (parity tap) */
    float moe_shared_w;
} orc_ctx;

/* ---- Float.float16ToFloat / Float.floatToFloat16 (IEEE binary16, RNE) ---- */
ORC_API float orc_f16_to_f32(uint16_t h) {
    uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
    uint32_t exp = (h >> 10) & 0x1Fu, man = h & 0x3FFu, bits;
    if (exp == 0) {
        if (man == 0) bits = sign;
        else { /* subnormal: value = man * 2^-24 */
            float f = (float)man * 5.9604644775390625e-08f;
            memcpy(&bits, &f, 4); bits |= sign;
        }
    } else if (exp == 31) bits = sign | 0x7F800000u | (man << 13);
    else bits = sign | ((exp + 112u) << 23) | (man << 13);
    float out; memcpy(&out, &bits, 4); return out;
}

ORC_API uint16_t orc_f32_to_f16(float f) {
    uint32_t x; memcpy(&x, &f, 4);
    uint32_t sign = (x >> 16) & 0x8000u;
    uint32_t ax = x & 0x7FFFFFFFu;
    if (ax >= 0x7F800000u) return (uint16_t)(sign | 0x7C00u | ((ax > 0x7F800000u) ? (0x200u | ((ax >> 13) & 0x3FFu)) : 0));
    if (ax >= 0x477FF000u) return (uint16_t)(sign | 0x7C00u);        /* rounds to inf (>= 65520) */
    if (ax < 0x33000001u) return (uint16_t)sign;                     /* <= 2^-25 rounds to 0 (tie to even) */
    int e = (int)(ax >> 23) - 127;
    uint32_t m = (ax & 0x7FFFFFu) | 0x800000u;
    int shift = (e < -14) ? (13 + (-14 - e)) : 13;                  /* subnormal halves shift further */
    uint32_t half_m = m >> shift;
    uint32_t rem = m & ((1u << shift) - 1u), halfway = 1u << (shift - 1);
    if (rem > halfway || (rem == halfway && (half_m & 1u))) half_m++;
    uint32_t he = (e < -14) ? 0u : (uint32_t)(e + 15);
    /* half_m carries the implicit bit for normals: adding it to (he-1)<<10 also handles mantissa carry */
    uint32_t out = (e < -14) ? half_m : (((he - 1u) << 10) + half_m);
    return (uint16_t)(sign | out);
}

static inline uint16_t rd16(const uint8_t* p) { return (uint16_t)(p[0] | (p[1] << 8)); }

/* ---- element access: FloatTensor.getFloat per ggml type ------------------
 * Q8_0FloatTensor.getFloat  J/tensor/standard/Q8_0FloatTensor.java:55-63
 * Q4_0FloatTensor.getFloat  J/tensor/standard/Q4_0FloatTensor.java:57-71
 * FP16FloatTensor.getFloat  J/tensor/standard/FP16FloatTensor.java:48-51      */
static inline float t_get(const orc_tensor* t, size_t i) {
    const uint8_t* p = (const uint8_t*)t->p;
    switch (t->type) {
    case ORC_F32: { float f; memcpy(&f, p + 4 * i, 4); return f; }
    case ORC_F16: return orc_f16_to_f32(rd16(p + 2 * i));
    case ORC_Q8_0: {
        const uint8_t* b = p + (i / 32) * 34;
        return (float)(int8_t)b[2 + (i % 32)] * orc_f16_to_f32(rd16(b));
    }
    case ORC_Q4_0: {
        const uint8_t* b = p + (i / 32) * 18;
        size_t m = i % 32; int qv;
        if (m < 16) qv = b[2 + m] & 0x0F; else qv = (b[2 + m - 16] >> 4) & 0x0F;
        return (float)(int8_t)(qv - 8) * orc_f16_to_f32(rd16(b));
    }
    }
    return 0.f;
}

/* ---- Q8_0 activation quantisation, hoisted out of dotQ8Activation --------
 * J/tensor/standard/Q8_0FloatTensor.java:90-123.  The Java code re-quantises
 * the activation block inside every row's dot; the values depend only on the
 * activation, so computing them once per matmul is the same arithmetic.      */
static void quantize_act(const float* x, int n, int8_t* aq, float* ascale) {
    for (int b = 0; b < n / 32; b++) {
        float amax = 0.f;
        for (int i = 0; i < 32; i++) { float av = fabsf(x[b * 32 + i]); if (av > amax) amax = av; }
        float qs = amax / 127.f;
        ascale[b] = orc_f16_to_f32(orc_f32_to_f16(qs));
        float ainv = qs != 0.f ? 1.f / qs : 0.f;
        for (int i = 0; i < 32; i++) {
            float s = x[b * 32 + i] * ainv;
            aq[b * 32 + i] = (int8_t)(int)(s + copysignf(0.5f, s));
        }
    }
}

static inline float dot_q8(const uint8_t* wrow, const int8_t* aq, const float* ascale, int n) {
    float result = 0.f;
    for (int b = 0; b < n / 32; b++) {
        const uint8_t* blk = wrow + b * 34;
        float wscale = orc_f16_to_f32(rd16(blk));
        const int8_t* wq = (const int8_t*)(blk + 2);
        int isum = 0;
        for (int i = 0; i < 32; i++) isum += (int)aq[b * 32 + i] * (int)wq[i];
        result += (float)isum * (wscale * ascale[b]);
    }
    return result;
}

/* FloatTensor.scalarDot  J/tensor/standard/FloatTensor.java:86-92 */
static inline float dot_scalar(const orc_tensor* w, size_t off, const float* x, int n) {
    float result = 0.f;
    for (int j = 0; j < n; j++) result += t_get(w, off + j) * x[j];
    return result;
}

/* ---- the Vector-API dots of F16 and Q4_0 with a 256-bit species (8 float lanes) --------------------------------------
 * These are what the reference computes for F16 / Q4_0 weights unless -Dllama.VectorBitSize=0 (FloatTensor.java:21-22,
 * LlamaApp.java:14); 256 bits is the preferred shape of an AVX2 host and the widest one Q4_0FloatTensor.vectorDot accepts
 * (:98-116 throws for 512).  Lane l accumulates elements l, l+8, l+16, ... with FUSED multiply-adds (FloatVector.fma);
 * reduceLanes(ADD) is evaluated in lane order from 0 (HotSpot's strictly ordered float add reduction on x86 = the Java
 * fall-back FloatVector.reduceLanesTemplate: ((((0 + v0) + v1) + ...) + v7)); sizes here are multiples of the lane count /
 * block size, so the scalar tails are empty.  Q8_0 is unaffected (dotQ8Activation is scalar, Q8_0FloatTensor.java:73-76),
 * and so is attention (ArrayFloatTensor inherits FloatTensor.dot = scalarDot).                                          */

/* FP16FloatTensor.vectorDot  J/tensor/standard/FP16FloatTensor.java:63-110: the f16 -> f32 bit trick flushes subnormal
 * weights to (signed) zero ("emulate DAZ") and is exact for normal values; infinities / NaNs are not supported there. */
static inline float f16_to_f32_daz(uint16_t h) {
    uint32_t b = h;
    uint32_t mask = (b & 0x7C00u) ? 0xFFFFFFFFu : 0u;
    uint32_t bits = ((b & 0x8000u) << 16) | ((((b & 0x7FFFu) + 0x1C000u) << 13) & mask);
    float f; memcpy(&f, &bits, 4); return f;
}

static inline float reduce_lanes(const float* v, int L) {
    float r = 0.f;
    for (int l = 0; l < L; l++) r = r + v[l];
    return r;
}

/* Species: FloatTensor.java:21 takes VectorShape.preferredShape() — 256 bits on an AVX2 host, 512 on AVX-512 (the GPU box's EPYC 9575F),
 * 128 on NEON / SSE — unless -Dllama.VectorBitSize says otherwise; L = bits / 32 float lanes.  FP16FloatTensor.vectorDot is
 * species-generic; the Q8_0 / Q4_0 vector dots have a 256-bit and a 128-bit branch and THROW for anything else
 * (Q8_0FloatTensor.java:165-167, Q4_0FloatTensor.java:118-120): restated as ORC_E_SPECIES from the forward calls.                      */
static float dot_f16_vec(const uint8_t* wrow, const float* x, int n, int L) {
    float val[16] = {0};
    int ub = n / L * L;
    for (int i = 0; i < ub; i += L)
        for (int l = 0; l < L; l++) val[l] = fmaf(f16_to_f32_daz(rd16(wrow + 2 * (i + l))), x[i + l], val[l]);     /* thizVector.fma(thatVector, val) */
    float result = reduce_lanes(val, L);
    if (ub < n) { float t = 0.f; for (int j = ub; j < n; j++) t += orc_f16_to_f32(rd16(wrow + 2 * j)) * x[j]; result += t; }
    return result;
}

/* One block of the Q8_0 / Q4_0 vector dots, q[32] = the block's dequantised integers in ELEMENT order (Q4_0: lo nibbles = elements
 * 0..15, hi nibbles = 16..31).  256 bits (:145-152 / :101-106): val = sum0.add(sum1).add(sum2).add(sum3).fma(wScale, val) with
 * sum_i = x[8 i .. 8 i + 7] * q[8 i .. 8 i + 7].  128 bits (:154-163 / :107-117): the same with 4-lane vectors over elements 0..15,
 * then again over 16..31 — two fmas per block.                                                                                         */
static inline void block_vec(const float* x, const float* q, float ws, float* val, int L) {
    for (int h = 0; h < 32; h += 4 * L)
        for (int l = 0; l < L; l++) {
            float sum0 = x[h + l] * q[h + l], sum1 = x[h + L + l] * q[h + L + l];
            float sum2 = x[h + 2 * L + l] * q[h + 2 * L + l], sum3 = x[h + 3 * L + l] * q[h + 3 * L + l];
            float s = ((sum0 + sum1) + sum2) + sum3;
            val[l] = fmaf(s, ws, val[l]);
        }
}

/* Q4_0FloatTensor.vectorDot  J/tensor/standard/Q4_0FloatTensor.java:82-133 */
static float dot_q4_0_vec(const uint8_t* wrow, const float* x, int n, int L) {
    float val[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int ub = n / 32 * 32;
    for (int j = 0; j < ub; j += 32) {
        const uint8_t* blk = wrow + (j / 32) * 18;
        float ws = orc_f16_to_f32(rd16(blk)), q[32];
        for (int e = 0; e < 16; e++) { q[e] = (float)(int8_t)((blk[2 + e] & 0x0F) - 8); q[16 + e] = (float)(int8_t)((blk[2 + e] >> 4) - 8); }
        block_vec(x + j, q, ws, val, L);
    }
    float result = 0.f;
    result += reduce_lanes(val, L);
    if (ub < n) {
        orc_tensor t = {wrow, ORC_Q4_0};
        float tl = 0.f;
        for (int j = ub; j < n; j++) tl += t_get(&t, j) * x[j];
        result += tl;
    }
    return result;
}

/* Q8_0FloatTensor.vectorDot  J/tensor/standard/Q8_0FloatTensor.java:125-175 (taken by dot() when llama.quantizeActivation=false and the
 * Vector API is on); scalar tail for n % 32 (never taken: K % 32 == 0) */
static float dot_q8_0_vec(const uint8_t* wrow, const float* x, int n, int L) {
    float val[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int ub = n / 32 * 32;
    for (int j = 0; j < ub; j += 32) {
        const uint8_t* blk = wrow + (j / 32) * 34;
        float ws = orc_f16_to_f32(rd16(blk)), q[32];
        const int8_t* qi = (const int8_t*)(blk + 2);
        for (int e = 0; e < 32; e++) q[e] = (float)qi[e];
        block_vec(x + j, q, ws, val, L);
    }
    float result = 0.f;
    result += reduce_lanes(val, L);
    if (ub < n) {
        orc_tensor t = {wrow, ORC_Q8_0};
        float tl = 0.f;
        for (int j = ub; j < n; j++) tl += t_get(&t, j) * x[j];
        result += tl;
    }
    return result;
}

/* FloatTensor.matmul  J/tensor/standard/FloatTensor.java:98-100 (rows in parallel) */
static void matmul(orc_ctx* o, const orc_tensor* w, const float* x, float* out, int d0, int d1) {
    const int L = o->vector_bits / 32;
    if (o->vector_bits && (w->type == ORC_F16 || w->type == ORC_Q4_0)) {
        const uint8_t* base = (const uint8_t*)w->p;
        size_t rb = w->type == ORC_F16 ? (size_t)d1 * 2 : (size_t)(d1 / 32) * 18;
        if (w->type == ORC_Q4_0 && L > 8) { o->species_error = 1; return; }         /* throw new UnsupportedOperationException(F_SPECIES.toString()) */
#pragma omp parallel for schedule(static)
        for (int i = 0; i < d0; i++)
            out[i] = w->type == ORC_F16 ? dot_f16_vec(base + (size_t)i * rb, x, d1, L) : dot_q4_0_vec(base + (size_t)i * rb, x, d1, L);
        return;
    }
    if (w->type == ORC_Q8_0 && o->f32_activation && o->vector_bits) {     /* Q8_0FloatTensor.dot :73-83 with QUANTIZE_ACTIVATION = false */
        const uint8_t* base = (const uint8_t*)w->p;
        size_t rb = (size_t)(d1 / 32) * 34;
        if (L > 8) { o->species_error = 1; return; }
#pragma omp parallel for schedule(static)
        for (int i = 0; i < d0; i++) out[i] = dot_q8_0_vec(base + (size_t)i * rb, x, d1, L);
        return;
    }
    if (w->type == ORC_Q8_0 && !o->f32_activation) {
        quantize_act(x, d1, o->aq, o->ascale);
        const uint8_t* base = (const uint8_t*)w->p;
        size_t rb = (size_t)(d1 / 32) * 34;
#pragma omp parallel for schedule(static)
        for (int i = 0; i < d0; i++) out[i] = dot_q8(base + (size_t)i * rb, o->aq, o->ascale, d1);
    } else {
#pragma omp parallel for schedule(static)
        for (int i = 0; i < d0; i++) out[i] = dot_scalar(w, (size_t)i * d1, x, d1);
    }
}

/* InferenceCore.rmsnorm  J/inference/InferenceCore.java:39-48 */
static void rmsnorm(float* out, const float* x, const orc_tensor* w, int offset, int size, float eps) {
    float ss = 0.f;
    for (int i = 0; i < size; i++) { float xi = x[offset + i]; ss = ss + xi * xi; }
    ss /= (float)size;
    ss += eps;
    ss = (float)(1.0 / sqrt((double)ss));
    for (int i = 0; i < size; i++) out[offset + i] = t_get(w, i) * (ss * x[offset + i]);
}

/* FloatTensor.softmaxInPlace  J/tensor/standard/FloatTensor.java:211-219 */
static void softmax(float* a, int n) {
    float maxv = -INFINITY;
    for (int i = 0; i < n; i++) maxv = fmaxf(maxv, a[i]);
    for (int i = 0; i < n; i++) a[i] = (float)exp((double)(a[i] - maxv));
    float sum = 0.f;
    for (int i = 0; i < n; i++) sum += a[i];
    for (int i = 0; i < n; i++) a[i] = a[i] / sum;
}

/* Attention over the KV cache — J/inference/InferenceCore.java:98-137 (Llama)
 * and :631-663 (Qwen3); identical arithmetic, head_size taken from config.   */
static void attention(orc_ctx* o, int l, int pos) {
    const orc_config* c = &o->c;
    int hs = c->head_size, kvd = o->kv_dim, kvmul = c->n_heads / c->n_kv_heads;
    float sqrt_hs = (float)sqrt((double)hs);
    const float* kc = o->key_cache + (size_t)l * c->ctx * kvd;
    const float* vc = o->value_cache + (size_t)l * c->ctx * kvd;
#pragma omp parallel for schedule(static)
    for (int h = 0; h < c->n_heads; h++) {
        const float* q = o->q + h * hs;
        float* att = o->att + (size_t)h * c->ctx;
        for (int t = 0; t <= pos; t++) {
            const float* kk = kc + (size_t)t * kvd + (h / kvmul) * hs;
            float score = 0.f;
            for (int j = 0; j < hs; j++) score += q[j] * kk[j];
            att[t] = o->c.arch == 3 ? score * o->c.attention_scale : score / sqrt_hs;        /* granite :870-872 */
        }
        softmax(att, pos + 1);
        float* xb = o->xb + h * hs;
        for (int j = 0; j < hs; j++) xb[j] = 0.f;
        for (int t = 0; t <= pos; t++) {
            const float* vv = vc + (size_t)t * kvd + (h / kvmul) * hs;
            float a = att[t];
            for (int j = 0; j < hs; j++) xb[j] = a * vv[j] + xb[j];   /* saxpyInPlace :221-227 */
        }
    }
}

/* dense SwiGLU on hb / hb2 — InferenceCore.java:155-158 (and :397-398, :408-409 of the MoE forward): exp in double */
static void swiglu(float* hb, const float* hb2, int n) {
    for (int i = 0; i < n; i++) {
        float v = hb[i];
        v = v / (float)(1.0 + exp(-(double)v));
        hb[i] = v * hb2[i];
    }
}

/* FloatTensor.matmul / InferenceCore.matmulExpert :430-432 on rows [row0, row0 + d0) of a stacked tensor: the same per-row dot */
static void matmul_rows(orc_ctx* o, const orc_tensor* w, size_t row0, const float* x, float* out, int d0, int d1) {
    orc_tensor sub = *w;
    size_t rb = w->type == ORC_Q8_0 ? (size_t)(d1 / 32) * 34 : w->type == ORC_Q4_0 ? (size_t)(d1 / 32) * 18 : w->type == ORC_F16 ? (size_t)d1 * 2 : (size_t)d1 * 4;
    sub.p = (const uint8_t*)w->p + row0 * rb;
    matmul(o, &sub, x, out, d0, d1);
}

/* Router probabilities and expert selection — InferenceCore.java:374-390: softmaxInPlace over ALL experts, then top-k by
 * repeated strict-> scans (the first index wins a tie); the selected probabilities are used as they are, not renormalised. */
static void moe_route(float* router, int E, int topk, int* sel, float* w) {
    softmax(router, E);
    for (int i = 0; i < topk; i++) {
        float best = -INFINITY; int index = -1;
        for (int j = 0; j < E; j++) if (router[j] > best) { best = router[j]; index = j; }
        sel[i] = index; w[i] = best;
        router[index] = -INFINITY;
    }
}
ORC_API void orc_moe_route(float* router_logits, int n_experts, int topk, int32_t* sel, float* w) {
    int s[64];
    moe_route(router_logits, n_experts, topk > 64 ? 64 : topk, s, w);
    for (int i = 0; i < topk && i < 64; i++) sel[i] = s[i];
}

/* The MoE feed-forward block of forwardJavaQwen2MoE — InferenceCore.java:363-415.  o->xb holds rmsnorm(x) on entry. */
static void moe_ffn(orc_ctx* o, int l) {
    const orc_config* c = &o->c;
    int dim = c->dim, E = c->n_experts, topk = c->n_experts_used, mh = c->moe_hidden;
    /* router: routerGate[l].matmul(xb, routerLogits, E, dim) (:373) then softmaxInPlace over ALL experts (:374) */
    matmul(o, &o->layer[ORC_T_GATE_INP][l], o->xb, o->router, E, dim);
    moe_route(o->router, E, topk, o->moe_sel, o->moe_w);
    /* routed experts in selection order, each accumulated into x with saxpy (:392-402) */
    for (int j = 0; j < topk; j++) {
        int e = o->moe_sel[j];
        matmul_rows(o, &o->layer[ORC_T_GATE_EXPS][l], (size_t)e * mh, o->xb, o->hbe, mh, dim);
        matmul_rows(o, &o->layer[ORC_T_UP_EXPS][l], (size_t)e * mh, o->xb, o->hbe2, mh, dim);
        swiglu(o->hbe, o->hbe2, mh);
        matmul_rows(o, &o->layer[ORC_T_DOWN_EXPS][l], (size_t)e * dim, o->hbe, o->ytmp, dim, mh);
        for (int i = 0; i < dim; i++) o->x[i] = o->moe_w[j] * o->ytmp[i] + o->x[i];
    }
    /* the always-on shared expert (:405-410) gated by sigmoid(sharedGateInp . xb) (:413-415) */
    matmul(o, &o->layer[ORC_T_W1][l], o->xb, o->hb, c->hidden, dim);
    matmul(o, &o->layer[ORC_T_W3][l], o->xb, o->hb2, c->hidden, dim);
    swiglu(o->hb, o->hb2, c->hidden);
    matmul(o, &o->layer[ORC_T_W2][l], o->hb, o->ytmp, dim, c->hidden);
    float gate_score = dot_scalar(&o->layer[ORC_T_GATE_INP_SHEXP][l], 0, o->xb, dim);
    float sw = 1.f / (1.f + (float)exp(-(double)gate_score));
    o->moe_shared_w = sw;
    for (int i = 0; i < dim; i++) o->x[i] = sw * o->ytmp[i] + o->x[i];
}

/* One transformer step.  arch 0: InferenceCore.forwardJava :50-172;
 * arch 1: forwardJavaQwen3 :565-697; arch 2: forwardJavaQwen2 :434-563 (q/k/v bias
 * :456-459, NeoX RoPE :461-478, no per-head norm).  want_logits=0 reproduces the prefill
 * variants (InferenceCoreWithPrefillDecode.forwardJavaPrefill :47-132 and the
 * per-token arithmetic of batchForwardJavaPrefill,
 * InferenceCoreBatchPrefillDecode.java:62-168 — same dot, same sequential
 * attention, so the KV cache written is bit-identical).  layer_x, if not NULL,
 * receives x after every layer ([L][dim], the parity tap).                    */
static void forward(orc_ctx* o, int token, int pos, int want_logits, float* layer_x) {
    const orc_config* c = &o->c;
    int dim = c->dim, hs = c->head_size, kvd = o->kv_dim, qd = o->q_dim;
    /* weights.token_embedding_table.copyTo(token * dim, state.x, 0, dim) */
    for (int i = 0; i < dim; i++) o->x[i] = t_get(&o->global[ORC_T_TOKEN_EMBD], (size_t)token * dim + i);
    if (c->arch == 3) for (int i = 0; i < dim; i++) o->x[i] = o->x[i] * c->embedding_scale;       /* forwardGranite :829 */

    for (int l = 0; l < c->n_layers; l++) {
        rmsnorm(o->xb, o->x, &o->layer[ORC_T_ATTN_NORM][l], 0, dim, c->rms_eps);
        matmul(o, &o->layer[ORC_T_WQ][l], o->xb, o->q, qd, dim);
        matmul(o, &o->layer[ORC_T_WK][l], o->xb, o->k, kvd, dim);
        matmul(o, &o->layer[ORC_T_WV][l], o->xb, o->v, kvd, dim);

        if (c->arch == 2 || c->arch == 5) {
            /* state.q.addInPlace(weights.q_bias[l]) ... — InferenceCore.java:456-459 */
            for (int i = 0; i < qd; i++) o->q[i] = o->q[i] + t_get(&o->layer[ORC_T_BQ][l], i);
            for (int i = 0; i < kvd; i++) o->k[i] = o->k[i] + t_get(&o->layer[ORC_T_BK][l], i);
            for (int i = 0; i < kvd; i++) o->v[i] = o->v[i] + t_get(&o->layer[ORC_T_BV][l], i);
        }
        if (c->arch == 0 || c->arch == 3) {
            /* adjacent-pair RoPE, q for i<dim and k for i<kvDim — InferenceCore.java:75-87; a Llama / Granite file has
             * qDim == dim (headSize = dim / heads), Devstral 2 runs the same loop over qDim — forwardJavaDevstral :198-212 */
            for (int i = 0; i < qd; i += 2) {
                int head_dim = i % hs;
                float fcr = o->rope_cr[(size_t)pos * (hs / 2) + head_dim / 2];
                float fci = o->rope_ci[(size_t)pos * (hs / 2) + head_dim / 2];
                int rotn = i < kvd ? 2 : 1;
                for (int v = 0; v < rotn; v++) {
                    float* vec = v == 0 ? o->q : o->k;
                    float v0 = vec[i], v1 = vec[i + 1];
                    vec[i] = v0 * fcr - v1 * fci;
                    vec[i + 1] = v0 * fci + v1 * fcr;
                }
            }
        } else {
            /* (qwen3: per-head Q/K RMSNorm, InferenceCore.java:594-600) then NeoX RoPE — :604-619 / qwen2 :461-478 */
            if (c->arch == 1) {
                for (int i = 0; i < c->n_heads; i++) rmsnorm(o->q, o->q, &o->layer[ORC_T_ATTN_Q_NORM][l], i * hs, hs, c->rms_eps);
                for (int i = 0; i < c->n_kv_heads; i++) rmsnorm(o->k, o->k, &o->layer[ORC_T_ATTN_K_NORM][l], i * hs, hs, c->rms_eps);
            }
            int half = hs / 2;
            for (int h = 0; h < c->n_heads; h++) {
                int rotn = h < c->n_kv_heads ? 2 : 1, poff = h * hs;
                for (int ic = 0; ic < half; ic++) {
                    float fcr = o->rope_cr[(size_t)pos * half + ic];
                    float fci = o->rope_ci[(size_t)pos * half + ic];
                    for (int vi = 0; vi < rotn; vi++) {
                        float* vec = vi == 0 ? o->q : o->k;
                        float v0 = vec[poff + ic], v1 = vec[poff + ic + half];
                        vec[poff + ic] = v0 * fcr - v1 * fci;
                        vec[poff + ic + half] = v0 * fci + v1 * fcr;
                    }
                }
            }
        }
        /* KV write — InferenceCore.java:92-93 */
        memcpy(o->key_cache + ((size_t)l * c->ctx + pos) * kvd, o->k, sizeof(float) * kvd);
        memcpy(o->value_cache + ((size_t)l * c->ctx + pos) * kvd, o->v, sizeof(float) * kvd);

        attention(o, l, pos);

        matmul(o, &o->layer[ORC_T_WO][l], o->xb, o->xb2, dim, qd);
        if (c->arch == 3) for (int i = 0; i < dim; i++) o->xb2[i] = o->xb2[i] * c->residual_scale;   /* :893 */
        for (int i = 0; i < dim; i++) o->x[i] = o->x[i] + o->xb2[i];

        rmsnorm(o->xb, o->x, &o->layer[ORC_T_FFN_NORM][l], 0, dim, c->rms_eps);
        if (c->arch == 5) { moe_ffn(o, l); if (layer_x) memcpy(layer_x + (size_t)l * dim, o->x, sizeof(float) * dim); continue; }
        matmul(o, &o->layer[ORC_T_W1][l], o->xb, o->hb, c->hidden, dim);
        matmul(o, &o->layer[ORC_T_W3][l], o->xb, o->hb2, c->hidden, dim);
        /* SwiGLU — InferenceCore.java:155-158: exp in double */
        for (int i = 0; i < c->hidden; i++) {
            float v = o->hb[i];
            v = v / (float)(1.0 + exp(-(double)v));
            o->hb[i] = v * o->hb2[i];
        }
        matmul(o, &o->layer[ORC_T_W2][l], o->hb, o->xb, dim, c->hidden);
        if (c->arch == 3) for (int i = 0; i < dim; i++) o->xb[i] = o->xb[i] * c->residual_scale;     /* :911 */
        for (int i = 0; i < dim; i++) o->x[i] = o->x[i] + o->xb[i];
        if (layer_x) memcpy(layer_x + (size_t)l * dim, o->x, sizeof(float) * dim);
    }
    if (!want_logits) return;
    rmsnorm(o->x, o->x, &o->global[ORC_T_OUTPUT_NORM], 0, dim, c->rms_eps);
    matmul(o, &o->global[ORC_T_OUTPUT], o->x, o->logits, c->vocab, dim);
    if (c->arch == 3) for (int i = 0; i < c->vocab; i++) o->logits[i] = o->logits[i] * c->logit_scale;   /* :921 */
}

/* ------------------------------- C API ----------------------------------- */
ORC_API orc_ctx* orc_create(const orc_config* cfg) {
    orc_ctx* o = (orc_ctx*)calloc(1, sizeof(orc_ctx));
    o->c = *cfg;
    const orc_config* c = &o->c;
    o->q_dim = c->n_heads * c->head_size;
    o->kv_dim = c->n_kv_heads * c->head_size;
    for (int i = 0; i < ORC_T_COUNT; i++) o->layer[i] = (orc_tensor*)calloc(c->n_layers, sizeof(orc_tensor));
    int big = c->dim > o->q_dim ? c->dim : o->q_dim;
    int maxk = big > c->hidden ? big : c->hidden;
    o->x = calloc(c->dim, 4); o->xb = calloc(big, 4); o->xb2 = calloc(c->dim, 4);
    o->q = calloc(o->q_dim, 4); o->k = calloc(o->kv_dim, 4); o->v = calloc(o->kv_dim, 4);
    o->hb = calloc(c->hidden, 4); o->hb2 = calloc(c->hidden, 4);
    o->att = calloc((size_t)c->n_heads * c->ctx, 4); o->logits = calloc(c->vocab, 4);
    o->key_cache = calloc((size_t)c->n_layers * c->ctx * o->kv_dim, 4);
    o->value_cache = calloc((size_t)c->n_layers * c->ctx * o->kv_dim, 4);
    if (c->moe_hidden > maxk) maxk = c->moe_hidden;
    o->aq = malloc(maxk); o->ascale = malloc(sizeof(float) * (maxk / 32 + 1));
    if (c->arch == 5) {
        o->router = calloc(c->n_experts, 4); o->hbe = calloc(c->moe_hidden, 4); o->hbe2 = calloc(c->moe_hidden, 4); o->ytmp = calloc(c->dim, 4);
    }
    return o;
}

ORC_API void orc_destroy(orc_ctx* o) {
    if (!o) return;
    for (int i = 0; i < ORC_T_COUNT; i++) free(o->layer[i]);
    free(o->x); free(o->xb); free(o->xb2); free(o->q); free(o->k); free(o->v); free(o->hb); free(o->hb2);
    free(o->att); free(o->logits); free(o->key_cache); free(o->value_cache); free(o->aq); free(o->ascale);
    free(o->router); free(o->hbe); free(o->hbe2); free(o->ytmp);
    free(o);
}

/* host pointer stays owned by the caller (the mmap'd GGUF tensor-data section) */
ORC_API int orc_set_tensor(orc_ctx* o, int id, int layer, const void* p, int ggml_type) {
    if (id < 0 || id >= ORC_T_COUNT) return -1;
    if (id <= ORC_T_OUTPUT) { o->global[id].p = p; o->global[id].type = ggml_type; return 0; }
    if (layer < 0 || layer >= o->c.n_layers) return -1;
    o->layer[id][layer].p = p; o->layer[id][layer].type = ggml_type; return 0;
}

ORC_API void orc_set_rope(orc_ctx* o, const float* cr, const float* ci) { o->rope_cr = cr; o->rope_ci = ci; }

ORC_API void orc_forward(orc_ctx* o, int token, int pos, float* logits_out, float* layer_x) {
    forward(o, token, pos, 1, layer_x);
    if (logits_out) memcpy(logits_out, o->logits, sizeof(float) * o->c.vocab);
}

ORC_API void orc_prefill(orc_ctx* o, const int32_t* tokens, int n, int start_pos) {
    for (int b = 0; b < n; b++) forward(o, tokens[b], start_pos + b, 0, NULL);
}

/* routing decision of the last layer of the last step: expert ids, their weights, the shared expert's gate */
ORC_API void orc_get_moe_routing(orc_ctx* o, int32_t* sel, float* w, float* shared_w) {
    for (int i = 0; i < o->c.n_experts_used; i++) { if (sel) sel[i] = o->moe_sel[i]; if (w) w[i] = o->moe_w[i]; }
    if (shared_w) *shared_w = o->moe_shared_w;
}

ORC_API void orc_get_x(orc_ctx* o, float* out) { memcpy(out, o->x, sizeof(float) * o->c.dim); }

ORC_API void orc_get_kv(orc_ctx* o, int layer, int pos, float* k_out, float* v_out) {
    size_t off = ((size_t)layer * o->c.ctx + pos) * o->kv_dim;
    memcpy(k_out, o->key_cache + off, sizeof(float) * o->kv_dim);
    memcpy(v_out, o->value_cache + off, sizeof(float) * o->kv_dim);
}

/* Sampler.TENSOR_ARGMAX → FloatTensor.argmax  J/tensor/standard/FloatTensor.java:138-151:
 * first index of the maximum (strict >), NaN never selected.                   */
ORC_API int orc_argmax(const float* v, int n) {
    int mi = 0; float mv = v[0];
    for (int i = 0; i < n; i++) if (v[i] > mv) { mv = v[i]; mi = i; }
    return mi;
}

/* RoPE.precomputeFreqsCis (ropeScaling=false branch)  J/inference/operation/RoPE.java:6-37 */
ORC_API void orc_rope_table(int ctx, int head_size, double theta, float* cr, float* ci) {
    size_t n = 0;
    for (int pos = 0; pos < ctx; ++pos)
        for (int i = 0; i < head_size; i += 2) {
            float freq = (float)(1.0 / pow(theta, i / (double)head_size));
            float val = pos * freq;
            cr[n] = (float)cos((double)val);
            ci[n] = (float)sin((double)val);
            n++;
        }
}

/* RoPE.precomputeFreqsCisYaRN  J/inference/operation/RoPE.java:39-74 with yarnCorrDim :76-78 and yarnRamp :80-83 (Devstral 2) */
static float yarn_corr_dim(int n_dims, int n_ctx_orig, float n_rot, float base) {
    return n_dims * (float)log((double)(n_ctx_orig / (n_rot * 2.0f * (float)M_PI))) / (2.0f * (float)log((double)base));
}
static float yarn_ramp(float low, float high, int i0) {
    float span = high - low;
    float y = (i0 - low) / (0.001f > span ? 0.001f : span);
    float c = 0.0f > y ? 0.0f : y;
    return 1.0f - (1.0f < c ? 1.0f : c);
}
ORC_API void orc_rope_table_yarn(int ctx, int head_size, double theta, float factor, float beta_fast, float beta_slow,
                                 float log_multiplier, int original_ctx, float* cr, float* ci) {
    float freq_scale = 1.0f / factor;
    float d0 = yarn_corr_dim(head_size, original_ctx, beta_fast, (float)theta);
    float d1 = yarn_corr_dim(head_size, original_ctx, beta_slow, (float)theta);
    float mscale = log_multiplier > 0 ? 1.0f + 0.1f * log_multiplier * (float)log((double)(1.0f / freq_scale)) : 1.0f;
    size_t n = 0;
    for (int pos = 0; pos < ctx; ++pos)
        for (int i = 0; i < head_size; i += 2) {
            float extrap = (float)(1.0 / pow(theta, i / (double)head_size));
            float interp = freq_scale * extrap;
            float mix = yarn_ramp(d0, d1, i / 2);
            float freq = interp * (1.0f - mix) + extrap * mix;
            float val = pos * freq;
            cr[n] = (float)cos((double)val) * mscale;
            ci[n] = (float)sin((double)val) * mscale;
            n++;
        }
}

/* stand-alone pieces exported for known-answer tests */
ORC_API void orc_quantize_act(const float* x, int n, int8_t* aq, float* ascale) { quantize_act(x, n, aq, ascale); }
ORC_API float orc_dot_q8(const uint8_t* wrow, const float* x, int n) {
    int8_t* aq = malloc(n); float* as = malloc(sizeof(float) * (n / 32 + 1));
    quantize_act(x, n, aq, as);
    float r = dot_q8(wrow, aq, as, n); free(aq); free(as); return r;
}
ORC_API float orc_get_float(const void* p, int type, long i) { orc_tensor t = {p, type}; return t_get(&t, (size_t)i); }
ORC_API void orc_rmsnorm(float* out, const float* x, const float* w, int size, float eps) {
    orc_tensor t = {w, ORC_F32}; rmsnorm(out, x, &t, 0, size, eps);
}
ORC_API void orc_softmax(float* a, int n) { softmax(a, n); }
/* ---- sampling: Sampler.selectSampler's lambda + CategoricalSampler / ToppSampler -------------------------------------
 * J/inference/sampler/Sampler.java:76-123 (temperature scaling, softmax, inner sampler), CategoricalSampler.java:33-44,
 * ToppSampler.java:24-160.  coin = rng.nextFloat(1f) of the caller's RandomGenerator.  probs_out (nullable) receives the
 * softmax output the sampler drew from.                                                                                 */
static int topp_cmp(const float* v, int a, int b) {      /* Comparator.comparingDouble(logits::getFloat).reversed() */
    double da = v[a], db = v[b];
    return (db < da) ? -1 : (db > da) ? 1 : 0;
}
static void topp_sift_down(int* array, int from, int n, const float* v) {            /* ToppSampler.siftDown :30-44 */
    int prev = from, next;
    while ((next = 2 * prev + 1) < n) {
        int r = 2 * prev + 2;
        if (r < n && topp_cmp(v, array[r], array[next]) < 0) next = r;
        if (topp_cmp(v, array[next], array[prev]) < 0) { int tmp = array[prev]; array[prev] = array[next]; array[next] = tmp; prev = next; }
        else break;
    }
}
ORC_API int orc_sample(const float* logits, int n, float temperature, float topp, float coin, float* probs_out) {
    if (temperature == 0.0f) return orc_argmax(logits, n);                           /* Sampler.java:79-81 */
    float* p = (float*)malloc(sizeof(float) * (size_t)n);
    for (int i = 0; i < n; i++) p[i] = logits[i] / temperature;                      /* divideInPlace(temperature) */
    softmax(p, n);
    if (probs_out) memcpy(probs_out, p, sizeof(float) * (size_t)n);
    int result;
    if (topp <= 0 || topp >= 1) {                                                    /* CategoricalSampler.sampleFromFloatTensor */
        float cdf = 0.0f;
        result = n - 1;
        for (int i = 0; i < n; i++) { cdf += p[i]; if (coin < cdf) { result = i; break; } }
    } else {                                                                         /* ToppSampler.sampleFromFloatTensor + processTopP */
        int* indices = (int*)malloc(sizeof(int) * (size_t)n);
        int head = 0, tail = n - 1;
        float cutoff = (1.0f - topp) / (n - 1);
        for (int i = 0; i < n; i++) { if (p[i] >= cutoff) indices[head++] = i; else indices[tail--] = i; }
        int n0 = head;
        for (int i = n0 / 2 - 1; i >= 0; --i) topp_sift_down(indices, i, n0, p);
        float cumulativeProb = 0.0f;
        int lastIndex = 0;
        for (int i = n0 - 1; i >= 0; i--) {
            int tmp = indices[0]; indices[0] = indices[i]; indices[i] = tmp;
            cumulativeProb += p[indices[i]];
            if (cumulativeProb > topp) { lastIndex = i; break; }
            topp_sift_down(indices, 0, i - 1, p);
        }
        float r = coin * cumulativeProb;
        float cdf = 0.0f;
        result = indices[lastIndex];
        for (int i = n0 - 1; i >= lastIndex; i--) { cdf += p[indices[i]]; if (r < cdf) { result = indices[i]; break; } }
        free(indices);
    }
    free(p);
    return result;
}

/* 0 = scalar dots (the default of this oracle, -Dllama.VectorBitSize=0); 128 / 256 / 512 = Vector-API dots of that species for F16 / Q4_0
 * matrices (and Q8_0 with the f32 activation) */
ORC_API int orc_set_vector_bits(orc_ctx* o, int bits) {
    if (bits != 0 && bits != 128 && bits != 256 && bits != 512) return -1;
    o->vector_bits = bits;
    o->species_error = 0;
    return 0;
}
/* 1 after a forward / prefill whose matmul the reference refuses for this species (UnsupportedOperationException); reading clears it */
ORC_API int orc_species_error(orc_ctx* o) { const int e = o->species_error; o->species_error = 0; return e; }
/* one row's Vector-API dot of a `bits`-wide species (KAT entry point); NaN for the combinations the reference throws for */
ORC_API float orc_dot_vec(const void* wrow, int type, const float* x, int n, int bits) {
    const int L = bits / 32;
    if (type == ORC_F16) return dot_f16_vec((const uint8_t*)wrow, x, n, L);
    if (L > 8) return NAN;
    return type == ORC_Q8_0 ? dot_q8_0_vec((const uint8_t*)wrow, x, n, L) : dot_q4_0_vec((const uint8_t*)wrow, x, n, L);
}
ORC_API float orc_dot_v256(const void* wrow, int type, const float* x, int n) { return orc_dot_vec(wrow, type, x, n, 256); }
/* 1 = -Dllama.quantizeActivation=false: Q8_0 matrices multiply the f32 activation (vector_bits 256: vectorDot, 0: scalarDot) */
ORC_API int orc_set_f32_activation(orc_ctx* o, int on) { o->f32_activation = on ? 1 : 0; return 0; }
/* thread pool size of the following forwards (tiny test models run faster on a few threads than on every logical CPU); returns the old value */
ORC_API int orc_set_num_threads(int n) {
#ifdef _OPENMP
    const int old = omp_get_max_threads();
    if (n > 0) omp_set_num_threads(n);
    return old;
#else
    (void)n;
    return 1;
#endif
}
ORC_API int orc_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
