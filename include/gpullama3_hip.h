/*
 * gpullama3_hip.h — C-ABI of libgpullama_hip.so, the MI355X-native replacement for the
 * TornadoVM TaskGraph layer of beehive-lab/GPULlama3.java.
 *
 * What it replaces (all paths relative to /root/reference/src/main/java/org/beehive/gpullama3/, "J/"):
 *   interface TornadoVMMasterPlan          J/tornadovm/TornadoVMMasterPlan.java:30-85
 *     initializeTornadoVMPlan(state,model) :55-70   -> gl3_create + gl3_upload_* + gl3_finalize
 *     forceCopyInReadOnlyData()            :79      -> gl3_finalize
 *     tornadoVMForwardDecode(position)     :81      -> gl3_forward_decode
 *     freeTornadoExecutionPlan()           :84      -> gl3_destroy
 *   TornadoVMMasterPlanBatchPrefillDecode.tornadoVMForwardBatchPrefill()
 *                                          J/tornadovm/TornadoVMMasterPlanBatchPrefillDecode.java:107-123
 *                                                   -> gl3_forward_prefill
 *   TornadoVMMasterPlanPrefillDecode.tornadoVMForwardPrefill(position)
 *                                          J/tornadovm/TornadoVMMasterPlanPrefillDecode.java:116
 *                                                   -> gl3_forward_prefill(n = 1)
 * In the reference, inputs cross this seam by side effect through the shared State object
 * (J/inference/state/State.java:28-100: embeddingX, wrapXBatch, batchStartPosHolder, wrapLogits) and
 * TornadoWeights (J/inference/weights/tornado/TornadoWeights.java:20-48).  Here every argument is
 * explicit: token ids in, logits (or the greedy argmax, Sampler.java:21-28) out; the embedding row
 * gather that InferenceCore.forwardTornadoVM does on the host (J/inference/InferenceCore.java:956-980)
 * happens on the device.
 *
 * Ownership: the library owns all device memory for the ctx lifetime.  Every host pointer passed in
 * stays owned by the caller and only has to be valid during the call (weights: the caller's mmap'd
 * GGUF tensor-data segment, J/tensor/GGUF.java:105-137; raw ggml block layout, J/tensor/GGMLType.java:5-21).
 * Errors: every function returns a gl3_status (0 = OK, negative = error); no C++ exception crosses
 * the ABI; gl3_last_error() gives the message (the Java shim maps non-zero to RuntimeException, as the
 * reference throws UnsupportedOperationException / TornadoOutOfMemoryException —
 * J/tornadovm/plan/ForwardPlanFactory.java:84-86).
 * Threading: a gl3_ctx is not re-entrant but may be called from any thread, never concurrently
 * (the reference serialises inference with a lock, J/server/InferenceService.java:59).  forward_*
 * return only after the requested outputs are in host memory.
 */
#ifndef GPULLAMA3_HIP_H
#define GPULLAMA3_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GL3_API __attribute__((visibility("default")))

typedef struct gl3_ctx gl3_ctx;

typedef enum {
    GL3_OK = 0,
    GL3_E_ARG = -1,          /* bad argument / tensor shape mismatch */
    GL3_E_UNSUPPORTED = -2,  /* model x quantisation x mode not implemented (ForwardPlanFactory.java:80-121) */
    GL3_E_OOM = -3,          /* hipMalloc failed (TornadoOutOfMemoryException) */
    GL3_E_HIP = -4,          /* any other HIP runtime error */
    GL3_E_RCCL = -5,         /* collective failed */
    GL3_E_STATE = -6         /* call order violated (e.g. forward before finalize) */
} gl3_status;

/* model families with all three plan modes in the reference (ForwardPlanFactory.java:123-141) */
/* LLAMA: InferenceCore.forwardJava (also Mistral GGUFs, architecture "llama") and forwardJavaDevstral :178-261 (Devstral 2,
 * architecture "mistral3": the same graph with head_size = attention.key_length != dim / n_heads and a YaRN RoPE table); QWEN3: forwardJavaQwen3 (per-head q/k RMSNorm,
 * NeoX RoPE); QWEN2: forwardJavaQwen2 :434-563 (q/k/v bias, NeoX RoPE; Qwen2.5, DeepSeek-R1-Distill-Qwen). */
enum { GL3_ARCH_LLAMA = 0, GL3_ARCH_QWEN3 = 1, GL3_ARCH_QWEN2 = 2,
       GL3_ARCH_GRANITE = 3, /* InferenceCore.forwardGranite :814-924: the Llama graph + embedding / attention / residual / logit scalars */
       GL3_ARCH_PHI3 = 4,    /* InferenceCore.forwardJavaPhi3 :699-800: fused attn_qkv and gate|up tensors (GL3_T_WQKV, GL3_T_W13), NeoX RoPE */
       GL3_ARCH_QWEN2MOE = 5 /* InferenceCore.forwardJavaQwen2MoE :263-422 (Qwen1.5-MoE / Qwen2-MoE): the Qwen2 attention; the FFN is an F32
                              * router over n_experts (softmax over all, top n_experts_used by strict >, no renormalisation), the
                              * selected experts' SwiGLU FFNs accumulated into x in selection order, then the always-on shared
                              * expert scaled by sigmoid(ffn_gate_inp_shexp . xb).  Q8_0 matrices, one rank, one sequence; a prefill chunk runs token by token. */ };

/* ggml tensor types of the wire format (J/tensor/GGMLType.java:5-21) */
enum { GL3_TYPE_F32 = 0, GL3_TYPE_F16 = 1, GL3_TYPE_Q4_0 = 2, GL3_TYPE_Q8_0 = 8,
       GL3_TYPE_Q4_K = 12, GL3_TYPE_Q5_K = 13, GL3_TYPE_Q6_K = 14 };   /* K-quants: converted to Q8_0 at load (gl3_kquant_to_q8_0) */

/* weight set (J/inference/weights/tornado/TornadoWeights.java:20-48; GGUF names in
 * J/model/loader/LlamaModelLoader.java:83-98, Qwen3ModelLoader.java:96-118) */
enum {
    GL3_T_TOKEN_EMBD = 0,   /* token_embd.weight        [vocab x dim]  quantised */
    GL3_T_OUTPUT_NORM = 1,  /* output_norm.weight       [dim]          F32       */
    GL3_T_OUTPUT = 2,       /* output.weight (wcls)     [vocab x dim]  quantised; omit when tied */
    GL3_T_ATTN_NORM = 3,    /* blk.L.attn_norm.weight   [dim]          F32       */
    GL3_T_WQ = 4,           /* blk.L.attn_q.weight      [qDim x dim]             */
    GL3_T_WK = 5,           /* blk.L.attn_k.weight      [kvDim x dim]            */
    GL3_T_WV = 6,           /* blk.L.attn_v.weight      [kvDim x dim]            */
    GL3_T_WO = 7,           /* blk.L.attn_output.weight [dim x qDim]             */
    GL3_T_FFN_NORM = 8,     /* blk.L.ffn_norm.weight    [dim]          F32       */
    GL3_T_W1 = 9,           /* blk.L.ffn_gate.weight    [hidden x dim]           */
    GL3_T_W2 = 10,          /* blk.L.ffn_down.weight    [dim x hidden]           */
    GL3_T_W3 = 11,          /* blk.L.ffn_up.weight      [hidden x dim]           */
    GL3_T_ATTN_Q_NORM = 12, /* blk.L.attn_q_norm.weight [head_size]    F32, qwen3 */
    GL3_T_ATTN_K_NORM = 13, /* blk.L.attn_k_norm.weight [head_size]    F32, qwen3 */
    GL3_T_BQ = 14,          /* blk.L.attn_q.bias        [q_dim]        F32, qwen2 (Qwen2StandardWeights q_bias) */
    GL3_T_BK = 15,          /* blk.L.attn_k.bias        [kv_dim]       F32, qwen2 */
    GL3_T_BV = 16,          /* blk.L.attn_v.bias        [kv_dim]       F32, qwen2 */
    GL3_T_WQKV = 17,        /* blk.L.attn_qkv.weight    [(qDim + 2 kvDim) x dim]  phi3: rows q | k | v (Phi3ModelLoader.java:112)          */
    GL3_T_W13 = 18,         /* blk.L.ffn_up.weight      [2 hidden x dim]          phi3: rows gate | up (forwardJavaPhi3 :778-780)          */
    /* qwen2moe (Qwen2MoEModelLoader.java:97-105).  The shared expert's ffn_{gate,up,down}_shexp.weight are uploaded as
     * GL3_T_W1 / GL3_T_W3 / GL3_T_W2 with gl3_model_desc.hidden = qwen2moe.feed_forward_length (sharedExpertHiddenDim). */
    GL3_T_FFN_GATE_INP = 19,       /* blk.L.ffn_gate_inp.weight        [n_experts x dim]               F32 (router)            */
    GL3_T_FFN_GATE_EXPS = 20,      /* blk.L.ffn_gate_exps.weight       [n_experts x moe_hidden x dim]  stacked routed experts  */
    GL3_T_FFN_UP_EXPS = 21,        /* blk.L.ffn_up_exps.weight         [n_experts x moe_hidden x dim]                          */
    GL3_T_FFN_DOWN_EXPS = 22,      /* blk.L.ffn_down_exps.weight       [n_experts x dim x moe_hidden]                          */
    GL3_T_FFN_GATE_INP_SHEXP = 23, /* blk.L.ffn_gate_inp_shexp.weight  [dim]                           F32 (shared-expert gate) */
    GL3_T_COUNT = 24
};

/* gl3_model_desc.flags */
#define GL3_FLAG_NO_GRAPH   0x1u  /* launch the decode step kernel by kernel instead of one hipGraph replay */
#define GL3_FLAG_LAYER_TAPS 0x2u  /* keep x after every layer for gl3_get_layer_x (parity tap)            */
#define GL3_FLAG_FORCE_RCCL 0x4u  /* run the tensor-parallel gathers even when tp_size == 1 (test hook)     */
#define GL3_FLAG_SCALAR_DOT 0x8u  /* F16 / Q4_0 matrices: the reference's SCALAR dot order (-Dllama.VectorBitSize=0,
                                     FloatTensor.scalarDot) instead of its default Vector-API order with a 256-bit species
                                     (FP16FloatTensor.vectorDot / Q4_0FloatTensor.vectorDot: 8 fused accumulator lanes).  Q8_0 is
                                     not affected: dotQ8Activation is scalar in both modes. */
#define GL3_FLAG_F32_ACTIVATION 0x10u /* Q8_0 matrices: -Dllama.quantizeActivation=false — f32 activation x dequantised weights in the
                                       * Vector-API order of Q8_0FloatTensor.vectorDot (256-bit species) instead of the int8 activation */
/* The species of the Vector-API order = -Dllama.VectorBitSize, default VectorShape.preferredShape() of the JVM's host
 * (J/tensor/standard/FloatTensor.java:21): 256 on an AVX2 host (this library's default), 512 on AVX-512 (e.g. the EPYC 9575F of an
 * MI355X node), 128 on SSE / NEON.  Only F16 / Q4_0 matrices and Q8_0 with GL3_FLAG_F32_ACTIVATION depend on it (the Q8_0 int8 path is
 * scalar).  GL3_FLAG_VECTOR_512: F16 matrices in the 16-accumulator order of FP16FloatTensor.vectorDot (decode kernels; a prefill chunk
 * runs token by token); Q4_0 / Q8_0-f32act plans are refused with GL3_E_UNSUPPORTED — the reference throws UnsupportedOperationException
 * for them (Q4_0FloatTensor.java:118-120, Q8_0FloatTensor.java:165-167; run such a JVM with -Dllama.VectorBitSize=256).
 * GL3_FLAG_VECTOR_128 (4 accumulators; two fmas per block for Q8_0 / Q4_0): restated in the oracles, refused here (GL3_E_UNSUPPORTED). */
#define GL3_FLAG_VECTOR_512 0x20u
#define GL3_FLAG_VECTOR_128 0x40u

/* Configuration (J/model/Configuration.java via LlamaModelLoader.java:47-63 / Qwen3ModelLoader.java:48-74)
 * plus the plan-selection knobs the reference reads from system properties
 * (llama.prefillBatchSize -> max_batch, TornadoVMMasterPlan.java:39-41). */
typedef struct {
    uint32_t struct_size;   /* sizeof(gl3_model_desc), for forward compatibility */
    int32_t arch;           /* GL3_ARCH_* */
    int32_t dim;            /* embedding_length */
    int32_t hidden;         /* feed_forward_length */
    int32_t n_layers;       /* block_count */
    int32_t n_heads;        /* attention.head_count */
    int32_t n_kv_heads;     /* attention.head_count_kv */
    int32_t head_size;      /* dim / n_heads (llama, phi3) or attention.key_length (qwen3); a multiple of 32 in 32 .. 256 */
    int32_t vocab;
    int32_t ctx;            /* context length = KV-cache rows per layer (no upper bound besides memory: f32 K and V rows) */
    float   rms_eps;        /* attention.layer_norm_rms_epsilon */
    int32_t weight_type;    /* GL3_TYPE_* of the matrices */
    int32_t max_batch;      /* largest prefill chunk (llama.prefillBatchSize); <= 1: no batched prefill buffers */
    int32_t device;         /* HIP device ordinal */
    int32_t tp_rank;        /* tensor-parallel rank of this process (0 when tp_size == 1) */
    int32_t tp_size;        /* tensor-parallel degree: heads / hidden units / dim rows of W2 / vocab rows are split tp_size ways
                             * (Wo is replicated: a rank computes the whole projection from the gathered attention output) */
    uint32_t flags;         /* GL3_FLAG_* */
    int32_t n_seqs;         /* independent sequences with their own KV cache (static batched decode); 0 or 1 = one */
    /* GL3_ARCH_GRANITE only (GraniteLoader.java:55-58: granite.embedding_scale, granite.attention.scale, granite.residual_scale,
     * granite.logit_scale): x *= embedding_scale after the embedding lookup, score *= attention_scale (instead of / sqrt(head_size)),
     * block output *= residual_scale before the residual add, logits *= logit_scale */
    float embedding_scale, attention_scale, residual_scale, logit_scale;
    /* GL3_ARCH_QWEN2MOE only (Qwen2MoEModelLoader.java:56-84: qwen2moe.expert_count, qwen2moe.expert_used_count, and the first
     * dimension of blk.0.ffn_down_exps.weight); 0 for every other architecture */
    int32_t n_experts, n_experts_used, moe_hidden;
} gl3_model_desc;

/* Per-kernel-class device time of one instrumented decode step (HIP events around every launch). */
enum { GL3_K_MATVEC_QKV = 0, GL3_K_MATVEC_WO = 1, GL3_K_MATVEC_GATEUP = 2, GL3_K_MATVEC_DOWN = 3,
       GL3_K_MATVEC_LOGITS = 4, GL3_K_ATTENTION = 5, GL3_K_OTHER = 6, GL3_K_COLLECTIVE = 7, GL3_K_COUNT = 8 };
typedef struct {
    double   ms[GL3_K_COUNT];        /* summed event-to-event time per class */
    uint32_t launches[GL3_K_COUNT];
    uint64_t bytes[GL3_K_COUNT];     /* algorithmic HBM bytes per class (weights + vectors), SURVEY.md §8d */
} gl3_kernel_times;

GL3_API const char* gl3_version(void);

/* Build the plan: allocates weights/KV/scratch in HBM (TornadoVMMasterPlan ctor: graph build + JIT are
 * replaced by precompiled gfx950 code objects). */
GL3_API int32_t gl3_create(const gl3_model_desc* desc, gl3_ctx** out);

/* Copy one tensor to HBM (FIRST_EXECUTION copy-in of the reference).  `host` is raw ggml blocks for the
 * FULL tensor, also under tensor parallelism (the library keeps only this rank's slice).  The library may
 * repack (split scale/quant planes); bytes moved per token are unchanged. */
GL3_API int32_t gl3_upload_tensor(gl3_ctx* ctx, int32_t tensor_id, int32_t layer, const void* host,
                                  uint64_t bytes, int32_t ggml_type);

/* freq_cis_real / freq_cis_imag as the host precomputes them (J/inference/operation/RoPE.java:6-37);
 * n = rows * head_size/2 floats, rows >= ctx. */
GL3_API int32_t gl3_upload_rope(gl3_ctx* ctx, const float* cr, const float* ci, uint64_t n);

/* Join a tensor-parallel group (one process per GPU; must precede gl3_finalize).  Two transports for the in-place
 * all-gathers of the row-split plan:
 *   peer-write over xGMI (default): every rank exports the IPC handle of its arena of gathered buffers
 *     (gl3_tp_p2p_handle, 64 bytes), the host exchanges the handles through any side channel (torch.distributed
 *     all_gather_object, a file, a socket) and hands all tp_size of them, in rank order, to gl3_tp_p2p_attach; one small
 *     kernel per gather then stores this rank's slice straight into the peers' buffers (csrc/gl3_tp.hip);
 *   RCCL (fall-back): `unique_id` is the id made by gl3_tp_unique_id on rank 0 and broadcast by the host. */
/* Transport choice BEFORE any plan exists: *reachable = how many of the n `peer_devices` (HIP ordinals) `device` can address
 * directly (hipDeviceCanAccessPeer; a device reaches itself).  reachable < n: use the RCCL transport (gl3_tp_init). */
GL3_API int32_t gl3_tp_peer_access(int32_t device, const int32_t* peer_devices, int32_t n, int32_t* reachable);
GL3_API int32_t gl3_tp_p2p_handle(gl3_ctx* ctx, void* out, uint64_t bytes);                  /* bytes >= 64 */
GL3_API int32_t gl3_tp_p2p_attach(gl3_ctx* ctx, const void* handles, uint64_t bytes);        /* tp_size x 64 bytes */
GL3_API int32_t gl3_tp_unique_id(void* out, uint64_t bytes);   /* bytes >= 128 */
GL3_API int32_t gl3_tp_init(gl3_ctx* ctx, const void* unique_id, uint64_t bytes);

/* In-process tensor-parallel group for TESTS on one GPU: `n` plans (one host thread each, same device) run the same
 * peer-write gather kernel, row split, kernel arguments and gather points as `n` processes on `n` GPUs; only the peers'
 * arena addresses are exchanged as plain pointers instead of IPC handles.  Create once, pass to every rank's
 * gl3_tp_attach_local before gl3_finalize (which waits for all ranks), destroy after the plans. */
typedef struct gl3_local_group gl3_local_group;
GL3_API int32_t gl3_local_group_create(int32_t n, gl3_local_group** out);
GL3_API void gl3_local_group_destroy(gl3_local_group* g);
GL3_API int32_t gl3_tp_attach_local(gl3_ctx* ctx, gl3_local_group* g);
/* After gl3_finalize: how this plan's decode step hands the gathered activations over (gl3_api.hip tp_fold_setup).  *mode = 0: one
 * gather launch per hand-over (every type but the Q8_0 int8 path, the RCCL transport, GL3_TP_FOLD=0); 1: the producing kernels
 * write their results into the peers' arenas themselves.  *consumer_mask: bit set = that consumer waits for the peers in its own
 * prologue (1 wo, 2 down, 4 qkv, 8 logits, 16 embedding; GL3_TP_FOLD=2 sets all), clear = a one-wavefront wait launch precedes it. */
GL3_API int32_t gl3_tp_fold_mode(gl3_ctx* ctx, int32_t* mode, int32_t* consumer_mask);
/* The per-rank arena is UNCACHED device memory and is never returned to the allocator when a plan is destroyed: it waits in a process-wide pool and
 * is re-used by the next tensor-parallel plan of the device that fits into it (best fit, capacity >= request), so a process retains at most as many
 * arenas as it had tensor-parallel plans alive at the same time.  gl3_tp_pool_stats reports what is retained (device -1: all devices);
 * gl3_tp_pool_trim releases it — only for a quiescent process (model reload in a server): freed uncached pages that come back from a later
 * hipMalloc under the cached policy were the root cause of stale activation rows (tensor parallelism). */
GL3_API int32_t gl3_tp_pool_stats(int32_t device, uint64_t* arenas, uint64_t* bytes);
GL3_API int32_t gl3_tp_pool_trim(int32_t device);

/* forceCopyInReadOnlyData(): checks that every tensor arrived, ties wcls to token_embd when
 * GL3_T_OUTPUT was not uploaded (AbstractModelLoader.java:194), captures the decode hipGraph. */
GL3_API int32_t gl3_finalize(gl3_ctx* ctx);

/* One decode step (InferenceCore.forwardTornadoVM + plan.tornadoVMForwardDecode(position)).
 * logits_out: caller-allocated f32[vocab] or NULL; argmax_out: first index of the maximum or NULL. */
GL3_API int32_t gl3_forward_decode(gl3_ctx* ctx, int32_t token, int32_t position, float* logits_out,
                                   int32_t* argmax_out);

/* One decode step + Sampler.selectSampler(vocab, temperature, topp, seed).sampleToken(logits)
 * (J/inference/sampler/Sampler.java:76-123): temperature == 0 -> greedy argmax; else logits / temperature, softmax (max,
 * exp in double, strictly sequential f32 sum, divide) on the device, then CategoricalSampler (topp outside (0,1); on the
 * device, 4 bytes come back) or ToppSampler (probabilities to the host, the reference's heap selection in native code).
 * `coin` is rng.nextFloat(1f) drawn by the CALLER from its own RandomGenerator — RandomGeneratorFactory.getDefault()
 * .create(seed) in the reference — one per sampled token, exactly where the reference draws it; the library holds no RNG, so
 * the random stream and the sampled ids are the reference's by construction. */
GL3_API int32_t gl3_forward_decode_sample(gl3_ctx* ctx, int32_t token, int32_t position, float temperature, float topp,
                                          float coin, int32_t* token_out);
/* Parity tap: the probabilities (f32[vocab]) the last gl3_forward_decode_sample sampled from. */
GL3_API int32_t gl3_get_sample_probs(gl3_ctx* ctx, float* out);
/* Parity / measurement tap: top-p draws of this plan answered by the device path (8 bytes back) and by the host heap (a tie between equal
 * probabilities at the sampled rank: the reference's heap order decides, gl3_sample.hip). */
GL3_API int32_t gl3_get_topp_counts(gl3_ctx* ctx, int64_t* on_device, int64_t* on_host);

/* Optional: page-lock a caller-owned host buffer (e.g. the MemorySegment the Java shim passes as logits_out on every step) so
 * that gl3_forward_decode copies the logits straight into it instead of going through the plan's pinned staging buffer and a
 * host memcpy.  The buffer must stay allocated until gl3_unpin_host_buffer / gl3_destroy.  `ptr` and `bytes` must be multiples
 * of 4096: registration pins and maps whole pages, so the buffer has to own its pages (mmap / posix_memalign /
 * Arena.allocate(bytes, 4096) with the size rounded up); anything else returns GL3_E_ARG and the plan keeps staging. */
GL3_API int32_t gl3_pin_host_buffer(gl3_ctx* ctx, void* ptr, uint64_t bytes);
GL3_API int32_t gl3_unpin_host_buffer(gl3_ctx* ctx, void* ptr);

/* Batched prefill of tokens[0..n) at positions start_pos.. (no logits, as the reference skips them).
 * n <= max_batch. */
GL3_API int32_t gl3_forward_prefill(gl3_ctx* ctx, const int32_t* tokens, int32_t n, int32_t start_pos);

/* Same for sequence `seq` (0 <= seq < n_seqs); gl3_forward_prefill is seq = 0. */
GL3_API int32_t gl3_forward_prefill_seq(gl3_ctx* ctx, int32_t seq, const int32_t* tokens, int32_t n, int32_t start_pos);

/* Static batched decode (BASELINE.json config 5; announced for the reference as PR #129, README.md:74): one decode
 * step for n independent sequences: token tokens[i] of sequence seq_ids[i] at position positions[i].  The n matvecs
 * become one int8-MFMA GEMM over the shared weights (weights are streamed once per step).  n <= max_batch, distinct
 * seq_ids.  logits_out: f32[n][vocab] or NULL; argmax_out: int32[n] (greedy id per sequence, sampled on the device) or NULL. */
GL3_API int32_t gl3_forward_decode_batch(gl3_ctx* ctx, const int32_t* tokens, const int32_t* seq_ids, const int32_t* positions,
                                         int32_t n, float* logits_out, int32_t* argmax_out);

/* Parity taps. */
GL3_API int32_t gl3_get_x(gl3_ctx* ctx, float* out /* f32[dim] */);
GL3_API int32_t gl3_get_layer_x(gl3_ctx* ctx, int32_t layer, float* out /* f32[dim], needs GL3_FLAG_LAYER_TAPS */);
GL3_API int32_t gl3_get_kv(gl3_ctx* ctx, int32_t layer, int32_t position, float* k_out, float* v_out /* f32[kvDim/tp] */);
GL3_API int32_t gl3_get_kv_seq(gl3_ctx* ctx, int32_t seq, int32_t layer, int32_t position, float* k_out, float* v_out);

/* Debug/parity tap: copy a scratch buffer of the LAST executed layer to the host.
 * which: 0 = raw q|k|v of the qkv projection, 1 = attention output xb, 2 = hb (SwiGLU output), 3 = logits. */
GL3_API int32_t gl3_get_buffer(gl3_ctx* ctx, int32_t which, float* out, uint64_t n_floats);

/* Test hook: the strictly sequential f32 sum of squares of x[0..n) (InferenceCore.rmsnorm's reduce), evaluated
 * by the same exact parallel device routine the kernels use (1024 <= n <= 5120, multiple of 4; device = ordinal). */
GL3_API int32_t gl3_debug_sumsq(int32_t device, const float* x, int32_t n, float* out);

GL3_API int32_t gl3_reset_kv(gl3_ctx* ctx);

/* One decode step launched kernel by kernel with HIP events around every launch. */
GL3_API int32_t gl3_profile_decode(gl3_ctx* ctx, int32_t token, int32_t position, gl3_kernel_times* out);

/* Average device time of ONE kernel class, measured with a single HIP event pair around `iters` sweeps over all
 * layers (back-to-back launches of that kernel on every layer's own weights, so nothing is cache-resident).
 * klass: GL3_K_MATVEC_*; out_us = mean per launch (includes the ~1.5 us inter-kernel boundary). */
GL3_API int32_t gl3_profile_kernel(gl3_ctx* ctx, int32_t klass, int32_t iters, double* out_us, uint64_t* bytes_per_launch);

/* The batched-prefill GEMM of one class (GL3_K_MATVEC_QKV / WO / GATEUP / DOWN) at n_tokens <= max_batch tokens: mean device
 * time per launch over `iters` sweeps of all layers, and the int8 multiply-add work of one launch (2 * rows * K * n_tokens
 * operations; the activations are whatever the last prefill left in the plan's buffers — timing only, the residual stream
 * is clobbered). */
GL3_API int32_t gl3_profile_prefill_kernel(gl3_ctx* ctx, int32_t klass, int32_t n_tokens, int32_t iters, double* out_us,
                                           uint64_t* int8_ops_per_launch);

/* Measured-achievable peaks of THIS device, for the roofline denominators beside the spec-sheet figures: a streaming
 * read of 1 GiB (float4 non-temporal loads, every CU busy; three regions read in rotation so that nothing is still in the Infinity Cache), a device-to-device copy of the same size (bytes counted = read + written) and
 * a loop of independent v_mfma_i32_32x32x32_i8 on every SIMD (4 accumulators per wavefront).  Best of 3 timed repetitions each,
 * HIP events on a private stream; any out pointer may be NULL.  Needs no plan. */
GL3_API int32_t gl3_probe_peaks(int32_t device, double* hbm_read_gbs, double* hbm_copy_gbs, double* int8_mfma_tops);

/* RunMetrics slots (TornadoVMMasterPlanSingleToken.java:40-54): plan creation and weight copy-in, ms. */
GL3_API int32_t gl3_get_init_ms(gl3_ctx* ctx, double* plan_ms, double* copy_in_ms);

GL3_API void gl3_destroy(gl3_ctx* ctx);
GL3_API const char* gl3_last_error(gl3_ctx* ctx);  /* ctx may be NULL: last create error */

/* ---- native GGUF loader.  Replaces, for this path, J/tensor/GGUF.java:43-137,217-311 (header,
 * metadata, tensor infos, alignment), the config extraction of LlamaModelLoader.java:47-69 / Qwen3ModelLoader.java:48-79, the
 * tensor-name map :83-98 and RoPE.precomputeFreqsCis (J/inference/operation/RoPE.java:6-37).  The file is mmap'd; tensor
 * bytes go from the mapping straight to gl3_upload_tensor (no 2 GiB limit, no TornadoVM 16-byte header remapping). */
typedef struct gl3_gguf gl3_gguf;
GL3_API int32_t gl3_gguf_open(const char* path, gl3_gguf** out);            /* GGUF v2 / v3 */
GL3_API void gl3_gguf_close(gl3_gguf* g);
GL3_API const char* gl3_gguf_last_error(const gl3_gguf* g);                 /* g may be NULL: last open / load error */
GL3_API int32_t gl3_gguf_tensor_count(const gl3_gguf* g);
/* ne[4]: GGUF dimension order (ne[0] = row length); data points into the mapping (valid until gl3_gguf_close). */
GL3_API int32_t gl3_gguf_tensor_info(const gl3_gguf* g, int32_t i, const char** name, int32_t* type, uint64_t* ne, const void** data,
                                     uint64_t* bytes);
GL3_API int32_t gl3_gguf_meta_number(const gl3_gguf* g, const char* key, double* out);   /* any scalar numeric / bool key */
GL3_API int32_t gl3_gguf_meta_string(const gl3_gguf* g, const char* key, const char** out);
/* Shape fields of desc (arch, dim, hidden, layers, heads, head_size, vocab, rms_eps, weight_type) from the metadata; desc->ctx is
 * kept if it is > 0 and not larger than <arch>.context_length, desc->ctx = 0 selects min(context_length, 4096) (the reference
 * always clamps with Configuration.withContextLength(maxTokens)); the other fields are left as the caller set them. */
GL3_API int32_t gl3_gguf_model_desc(gl3_gguf* g, gl3_model_desc* desc, float* rope_theta);
/* RoPE.precomputeFreqsCis with ropeScaling = false: cr / ci are f32[ctx * head_size/2]. */
GL3_API void gl3_rope_table(int32_t ctx, int32_t head_size, float theta, float* cr, float* ci);
/* RoPE.precomputeFreqsCisYaRN (J/inference/operation/RoPE.java:39-83; Devstral 2, DevstralModelLoader.java:76-110): ramp between the
 * plain and the / factor frequency per pair, cos / sin scaled by mscale = 1 + 0.1 * log_multiplier * ln(factor) (1 when
 * log_multiplier <= 0).  gl3_load_gguf builds this table when <arch>.rope.scaling.type == "yarn". */
GL3_API void gl3_rope_table_yarn(int32_t ctx, int32_t head_size, float theta, float factor, float beta_fast, float beta_slow,
                                 float log_multiplier, int32_t original_ctx, float* cr, float* ci);
/* <arch>.rope.scaling.{factor, yarn_beta_fast, yarn_beta_slow, yarn_log_multiplier (default 0), original_context_length}
 * (DevstralModelLoader.java:80-86): returns 1 and fills them for a "yarn" file, 0 for a file with the plain table. */
GL3_API int32_t gl3_gguf_yarn_params(gl3_gguf* g, float* factor, float* beta_fast, float* beta_slow, float* log_multiplier,
                                     int32_t* original_ctx);
/* ModelLoader.dequantizeToQ8_0TornadoTensor (J/model/loader/ModelLoader.java:173-224): the load-time conversion the reference's GPU
 * path applies to Q4_K / Q5_K / Q6_K tensors — element-wise getFloat of the CPU tensor classes (Q4_KFloatTensor.java:86-114,
 * Q5_KFloatTensor.java:86-120, Q6_KFloatTensor.java) re-quantised to Q8_0 blocks.  n_elements % 256 == 0; dst: n / 32 * 34 bytes.
 * gl3_load_gguf applies it to every K-quant tensor; a host that uploads tensors itself can call it first. */
GL3_API int32_t gl3_kquant_to_q8_0(int32_t src_type, const void* src, uint64_t n_elements, void* dst);

/* open + gl3_create + one gl3_upload_tensor per tensor + RoPE table + gl3_finalize (not finalized when opts->tp_size > 1 or
 * GL3_FLAG_FORCE_RCCL is set: call gl3_tp_init / gl3_tp_attach_local and gl3_finalize).  opts may be NULL; it supplies
 * ctx / max_batch / device / tp_rank / tp_size / flags / n_seqs. */
GL3_API int32_t gl3_load_gguf(const char* path, const gl3_model_desc* opts, gl3_ctx** out);

#ifdef __cplusplus
}
#endif
#endif
