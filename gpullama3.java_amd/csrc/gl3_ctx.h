// gl3_ctx.h — plan state shared by the translation units of libgpullama_hip.so.
// Mirrors what the reference keeps in State (J/inference/state/State.java:28-100) + TornadoWeights
// (J/inference/weights/tornado/TornadoWeights.java:20-48), but resident in HBM for the ctx lifetime.
#pragma once
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/gpullama3_hip.h"

// Slack behind every weight matrix (and the small-batch activation buffers): bdw_gemm_kernel's load rings run up to 16 tiles
// past the end of a strip instead of clamping or guarding their addresses (gl3_bd_gemm.h).
constexpr size_t GL3_TAIL_PAD = 256 * 1024;

// Internal format code of Q8_0 matrices run with an f32 activation (GL3_FLAG_F32_ACTIVATION): VL layout, gl3_veclane_kernels.h
// diagnostic switches: environment variable as a flag (unset / empty = default)
static inline bool env_flag(const char* name, bool dflt) {
    const char* v = getenv(name);
    if (!v || !*v) return dflt;
    return atoi(v) != 0;
}

// roctx ranges (SURVEY.md §5 "tracing"): GL3_ROCTX=1 brackets the decode step, every layer and every kernel class of an EAGER step
// (graphs are switched off: a replayed graph has no per-layer host calls) with roctxRangePush / Pop, resolved at run time from
// rocprofiler-sdk's roctx library (or roctracer's) so that the library has no link-time dependency on a profiler.
//     GL3_ROCTX=1 rocprofv3 --marker-trace --kernel-trace --output-format csv -- python bench.py --steps 1 --no-pp
#include <dlfcn.h>
struct Gl3Roctx { int (*push)(const char*) = nullptr; int (*pop)() = nullptr; };
static inline Gl3Roctx& gl3_roctx() {
    static Gl3Roctx r = [] {
        Gl3Roctx x;
        if (!env_flag("GL3_ROCTX", false)) return x;
        void* h = dlopen("librocprofiler-sdk-roctx.so", RTLD_NOW | RTLD_GLOBAL);
        if (!h) h = dlopen("libroctx64.so", RTLD_NOW | RTLD_GLOBAL);
        if (h) {
            x.push = reinterpret_cast<int (*)(const char*)>(dlsym(h, "roctxRangePushA"));
            x.pop = reinterpret_cast<int (*)()>(dlsym(h, "roctxRangePop"));
            if (!x.push || !x.pop) x = Gl3Roctx{};
        }
        return x;
    }();
    return r;
}
static inline bool gl3_roctx_on() { return gl3_roctx().push != nullptr; }
struct Gl3Range {
    bool on;
    explicit Gl3Range(const char* name) : on(gl3_roctx_on()) { if (on) gl3_roctx().push(name); }
    Gl3Range(const char* what, int i) : on(gl3_roctx_on()) { if (on) { char b[48]; snprintf(b, sizeof b, "%s %d", what, i); gl3_roctx().push(b); } }
    ~Gl3Range() { if (on) gl3_roctx().pop(); }
    Gl3Range(const Gl3Range&) = delete;
    Gl3Range& operator=(const Gl3Range&) = delete;
};

constexpr int GL3_FMT_Q8V = 108;

struct Q8Mat {               // one repacked matrix: Q8T tiles (gl3_decode_kernels.h), VL groups (gl3_veclane_kernels.h) or row-lane groups
    uint8_t* w = nullptr;
    int rows = 0, k = 0;
    int ng = 0;              // Q8T: tile groups per strip = ceil(k/32 / 4)
    int nstrips = 0;         // Q8T: ceil(rows / 16)
    int fmt = 8;             // 8 = Q8_0 as Q8T tiles (int8 activation), 1 = F16, 2 = Q4_0, GL3_FMT_Q8V = Q8_0 with f32 activation
    bool vl = false;         // "VL" layout of the Vector-API-order kernels; false (F16 / Q4_0 only) = row-lane, scalar order
    size_t rl_group_bytes() const { return fmt == 1 ? (size_t)(k / 8) * 1024 : (size_t)(k / 32) * 1152; }
    size_t vl_group_bytes() const { return fmt == 1 ? (size_t)(k / 64) * 1024 : fmt == 2 ? (size_t)(k / 256) * 1152 : (size_t)(k / 128) * 1088; }
    size_t bytes() const {
        if (fmt != 8 && vl) return (size_t)((rows + 7) / 8) * vl_group_bytes();
        if (fmt != 8) return (size_t)((rows + 63) / 64) * rl_group_bytes();
        return (size_t)(nstrips + (nstrips & 1)) * ng * 2176;            // even #strips: the prefill GEMM reads 32-row groups
    }
    size_t algo_bytes() const {                                          // GGUF bytes (no padding)
        return (fmt == 8 || fmt == GL3_FMT_Q8V) ? (size_t)rows * (k / 32) * 34 : fmt == 1 ? (size_t)rows * k * 2 : (size_t)rows * (k / 32) * 18;
    }
};

struct gl3_layer {
    Q8Mat wqkv, wo, w1, w3, w2;
    float *attn_norm = nullptr, *ffn_norm = nullptr, *qnorm = nullptr, *knorm = nullptr;
    float *bq = nullptr, *bk = nullptr, *bv = nullptr;      // qwen2: this rank's rows of the q / k / v bias
    // qwen2moe: stacked routed experts ([n_experts * moe_hidden x dim] x 2, [n_experts * dim x moe_hidden]), the F32 router rows
    // [n_experts x dim] and the F32 shared-expert gate row [dim]; the shared expert itself is w1 / w3 / w2
    Q8Mat gate_exps, up_exps, down_exps;
    float *gate_inp = nullptr, *gate_inp_shexp = nullptr;
    uint32_t have = 0;       // bit per tensor id
};

// In-process tensor-parallel group (tests): ranks are host threads of one process sharing one device.  The ranks run the
// SAME peer-write gather kernel as separate processes do over xGMI; only the way a rank learns its peers' arena addresses
// differs (plain pointers here, hipIpcOpenMemHandle there).
struct gl3_local_group {
    int n = 0;
    std::mutex mu;
    std::condition_variable cv;
    int arrived = 0, generation = 0;
    std::vector<struct gl3_ctx*> ranks;
    void barrier() {
        std::unique_lock<std::mutex> lk(mu);
        const int gen = generation;
        if (++arrived == n) { arrived = 0; ++generation; cv.notify_all(); }
        else cv.wait(lk, [&] { return generation != gen; });
    }
};

constexpr int GL3_MAX_TP = 16;
enum { GL3_TP_NONE = 0, GL3_TP_RCCL = 1, GL3_TP_P2P = 2 };

// Tensor-parallel arena: ONE device allocation per rank that holds every gathered activation buffer plus the transport's
// header, at the same offsets on every rank, so a peer addresses "rank p's copy of buffer b" as peer_base[p] + off[b].
//   header: u32 flags[GL3_MAX_TP] (flags[p] = number of gathers rank p has pushed into this arena; written by the peers),
//           u32 seq (gathers completed by this rank), u32 arrive (workgroup ticket of the running gather kernel)
struct gl3_tp_arena {
    uint8_t* base = nullptr;
    size_t bytes = 0;                             // bytes this plan uses
    size_t cap = 0;                               // capacity of the allocation (>= bytes when it came from the pool)
    size_t off[8] = {};                           // byte offset of buffer GB_*; 0 = not allocated
    int pf_logits_rows = 0;                       // capacity of the batched-decode logits buffer (rows)
    int kind = 0;                                 // 0 uncached (default), 1 cached, 2 fine-grained, 3 uncached + hipFree (GL3_TP_ARENA experiments)
    bool pooled = false;                          // base goes back to the process-wide arena pool, not to hipFree (gl3_tp.hip)
};
constexpr size_t GL3_ARENA_HDR = 1024, GL3_ARENA_SEQ = 64, GL3_ARENA_ARRIVE = 68;      // bytes 256..511: checksums of GL3_TP_DEBUG (gl3_tp.hip)
// folded gathers (decode, Q8_0 int8 path): u32 flags[3][GL3_MAX_TP] per buffer GB_XB / GB_X / GB_HB (flags[b][p] = gathers of buffer b
// rank p has completed into this arena), u32 ticket[3] (finished producer wavefronts of the running launch), u32 step (decode steps)
constexpr size_t GL3_ARENA_FOLD = 512, GL3_ARENA_FOLD_TICKET = 704, GL3_ARENA_STEP = 720;

struct gl3_ctx {
    gl3_model_desc d{};
    // derived (local = this tensor-parallel rank's share)
    int q_dim = 0, kv_dim = 0, heads_l = 0, kv_heads_l = 0, q_dim_l = 0, kv_dim_l = 0, hidden_l = 0, vocab_l = 0, dim_l = 0;
    int n_tsplit = 1;        // ceil(ctx / 64) score tiles
    int attn_win = 0;        // floats of a softmax row kept in LDS by the long-context attention kernels (longer rows: windows)
    bool wo_replicated = false;   // tensor parallel: every rank holds all rows of Wo (3 gathers per layer instead of 4)
    int wo_rows = 0;              // rows of Wo held by this rank: dim (replicated) or dim / tp
    // Granite (InferenceCore.forwardGranite): embedding / residual / logit multipliers (1 otherwise) and the score multiplier
    // (0 = divide by sqrt(head_size)); rope_arch = RoPE / per-head-norm flavour of the kernels (0 adjacent pairs, 1 qwen3, 2 NeoX)
    float emb_scale = 1.f, resid_scale = 1.f, logit_scale = 1.f, att_mul = 0.f;
    int rope_arch = 0;
    hipStream_t stream = nullptr;
    // weights
    Q8Mat emb, wcls;
    bool wcls_owned = false;
    float* out_norm = nullptr;
    uint32_t have_global = 0;
    std::vector<gl3_layer> layers;
    float *rope_cr = nullptr, *rope_ci = nullptr;
    uint64_t rope_n = 0;
    // state
    float *kcache = nullptr, *vcache = nullptr;   // [n_seqs][L][ctx][kv_dim_l]
    int n_seqs = 1;
    size_t kv_seq_stride = 0;                     // floats per sequence
    float *x = nullptr, *qkv = nullptr, *xb = nullptr, *hb = nullptr, *logits = nullptr, *att = nullptr;
    float *att_t = nullptr, *att_tmax = nullptr, *att_sums = nullptr;   // long-context decode: softmax numerators in attn_pv_kernel's operand order, per-tile score maxima, denominators
    float* xn = nullptr;                          // RMS-normalised activation (F16 / Q4_0 path only)
    float* taps = nullptr;                        // [L][dim] when GL3_FLAG_LAYER_TAPS
    // qwen2moe scratch (Qwen2MoEState.java:15-32): router logits [n_experts], routing weights [topk + 1] (last = shared-expert
    // gate), selected expert ids [topk], the selected experts' SwiGLU outputs [topk][moe_hidden], the down-projected outputs
    // [topk + 1][dim] (last = shared expert)
    float *moe_logits = nullptr, *moe_w = nullptr, *moe_hb = nullptr, *moe_y = nullptr;
    int* moe_sel = nullptr;                       // [topk] expert ids, then the router kernel's arrival ticket
    void* moe_slots = nullptr;                    // MoeSlots[n_layers][3]: gate/up with the shared expert merged, gate/up alone, down
    int *dyn = nullptr, *argmax = nullptr;        // dyn[0] = token, dyn[1] = position
    int* h_dyn = nullptr;                         // pinned
    const int* dyn_cur = nullptr;                 // (token, position) pair the next launches read: dyn, or an entry of dyn_seq
    int* dyn_seq = nullptr;                       // [2 * dyn_seq_cap] pairs of a sequential (token-by-token) prefill chunk
    int dyn_seq_cap = 0;
    float* h_logits = nullptr;                    // pinned f32[vocab]
    int* h_argmax = nullptr;
    float *sm_probs = nullptr, *sm_aux = nullptr, *h_probs = nullptr;   // sampling (gl3_sample.hip): probabilities, scratch, pinned copy
    int64_t topp_device = 0, topp_host = 0;       // top-p draws answered on the device / by the host heap (gl3_get_topp_counts)
    void* sm_sort = nullptr;                      // top-p on the device: (key, index) x 2, radix histogram, result words
    std::vector<int> topp_indices;
    std::vector<std::pair<void*, size_t>> pinned;     // caller buffers registered with gl3_pin_host_buffer (logits land there directly)
    // upload staging
    uint8_t* staging = nullptr;
    size_t staging_bytes = 0;
    // batched prefill (gl3_prefill.hip)
    struct gl3_prefill_state* pf = nullptr;
    // execution
    bool finalized = false;
    hipGraph_t graph = nullptr;
    hipGraphExec_t graph_exec = nullptr;          // decode step incl. logits
    hipGraph_t graph_s = nullptr;
    hipGraphExec_t graph_exec_s = nullptr;        // same step with the fused short-context attention (pos < AF_MAXN)
    hipGraph_t graph_m = nullptr;
    hipGraphExec_t graph_exec_m = nullptr;        // same step with the two-launch attention (AF_MAXN <= pos < attn_mid)
    int attn_mid = 0;                             // see attn_mode (gl3_api.hip)
    bool fused_attn_ok = false;                   // shape admits attn_head_kernel (one launch per layer for positions < AF_MAXN)
    ncclComm_t comm = nullptr;
    gl3_local_group* lgrp = nullptr;
    bool use_rccl = false;                        // tensor-parallel gathers are active (any transport)
    int transport = GL3_TP_NONE;                  // GL3_TP_RCCL (gl3_tp_init) or GL3_TP_P2P (gl3_tp_p2p_attach / gl3_tp_attach_local)
    gl3_tp_arena arena;
    uint8_t* peer_base[GL3_MAX_TP] = {};          // arena of every rank as mapped into this process (peer_base[tp_rank] = arena.base)
    void* ipc_opened[GL3_MAX_TP] = {};            // mappings to close (hipIpcCloseMemHandle)
    int tp_fold_mask = 0;                         // TF_* consumers that wait in their own prologue
    bool tp_quiet = false;                        // gl3_profile_kernel: launch without the folded hand-overs (no pushes into the peers' arenas)
    int tp_fold = 0;                              // decode gathers folded into producers / consumers (gl3_api.hip tp_fold_setup): 0, 1, 2
    void* tp_recs = nullptr;                      // device TpRec[n_layers * 8 + 2]
    uint32_t* h_tp_err = nullptr;                 // pinned, device-visible: set by a gather kernel whose peers never arrived
    size_t tp_dbg_prev_off = 0, tp_dbg_prev_n4 = 0;   // GL3_TP_DEBUG: the previous gather of this forward call (re-checked by the next one)
    std::vector<hipEvent_t> ev;
    hipEvent_t prof_ev0 = nullptr, prof_ev1 = nullptr;   // non-null: the next matvec launch carries them as start / stop events
    // metrics
    double plan_ms = 0, copy_in_ms = 0;
    std::string err;
};

#define GL3_HIP(call)                                                                                  \
    do {                                                                                               \
        hipError_t e_ = (call);                                                                        \
        if (e_ != hipSuccess) {                                                                        \
            ctx->err = std::string(#call) + ": " + hipGetErrorString(e_);                              \
            return e_ == hipErrorOutOfMemory ? GL3_E_OOM : GL3_E_HIP;                                  \
        }                                                                                              \
    } while (0)

#define GL3_NCCL(call)                                                                                 \
    do {                                                                                               \
        ncclResult_t r_ = (call);                                                                      \
        if (r_ != ncclSuccess) {                                                                       \
            ctx->err = std::string(#call) + ": " + ncclGetErrorString(r_);                             \
            return GL3_E_RCCL;                                                                         \
        }                                                                                              \
    } while (0)

#define GL3_FAIL(code, msg)                                                                            \
    do {                                                                                               \
        ctx->err = (msg);                                                                              \
        return (code);                                                                                 \
    } while (0)

// gl3_api.hip: in-place all-gather of buf = [tp][count_per_rank] over the plan's transport (RCCL or the local test group);
// which = GB_* id of the buffer (the local transport resolves the peers' pointers through gl3_gather_buf)
enum { GB_XB = 0, GB_X = 1, GB_HB = 2, GB_LOGITS = 3, GB_PF_X = 4, GB_PF_AO = 5, GB_PF_HB = 6, GB_PF_LOGITS = 7 };
float* gl3_gather_buf(gl3_ctx* c, int which);
int32_t gl3_all_gather(gl3_ctx* ctx, int which, size_t count_per_rank);
int32_t gl3_tp_arena_alloc(gl3_ctx* ctx);
void gl3_tp_arena_free(gl3_ctx* ctx);
int32_t gl3_tp_local_resolve(gl3_ctx* ctx);
int32_t gl3_tp_check(gl3_ctx* ctx);      // after a stream sync: GL3_E_RCCL if a gather timed out

// gl3_sample.hip
int32_t gl3_sample_run(gl3_ctx* ctx, const float* logits_dev, float temperature, float topp, float coin, int32_t* token_out);
int32_t gl3_sample_probs(gl3_ctx* ctx, float* out);
void gl3_sample_free(gl3_ctx* ctx);

// gl3_prefill.hip
float* gl3_prefill_buf(gl3_ctx* ctx, int which);
int32_t gl3_prefill_alloc(gl3_ctx* ctx);
void gl3_prefill_free(gl3_ctx* ctx);
int32_t gl3_prefill_run(gl3_ctx* ctx, int32_t seq, const int32_t* tokens, int32_t n, int32_t start_pos);
int32_t gl3_prefill_profile(gl3_ctx* ctx, int klass, int n, int iters, double* out_us, uint64_t* int8_ops);
int32_t gl3_decode_batch_run(gl3_ctx* ctx, const int32_t* tokens, const int32_t* seq_ids, const int32_t* positions, int32_t n,
                             float* logits_out, int32_t* argmax_out);
