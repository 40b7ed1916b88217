// gl3_api.hip — C-ABI of libgpullama_hip.so (include/gpullama3_hip.h): plan lifetime, weight upload +
// repack, the decode-step launch sequence (captured once into a hipGraph), parity taps and profiling.
//
// The launch sequence follows the task list of the reference's single-token plan
// (J/tornadovm/layers/type/q8_0/LlamaQ8_0FFNLayers.java:111-230, LogitsQ8_0Layer.java:60-95) fused down to
// six launches per layer:
//   1 qkv matvec   = attn_rms_reduce + attn_rms_apply + qkv_projection        (RMSNorm + act-quant in prologue)
//   2 scores       = rope_and_kv_cache + q.k scores (tiles of 64 timesteps)
//   3 softmax + PV = softmax and weighted V sum, t ascending
//   4 wo matvec    = attn_output_proj (+ residual)
//   5 gate/up      = ffn_rms_reduce + rms_ffn_gate_up (SwiGLU epilogue)
//   6 down matvec  = ffn_down_proj (+ residual)
// and per token: embedding gather/dequant, final RMSNorm fused into the vocab projection, optional argmax.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "gl3_ctx.h"
#include "gl3_decode_kernels.h"
#include "gl3_rowlane_kernels.h"
#include "gl3_veclane_kernels.h"
#include "gl3_moe_kernels.h"

using namespace gl3;

static std::string g_create_err;

static double now_ms() {
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}


// ------------------------------------------------------------------------------------------------ matvec launch
static size_t matvec_smem(int pro, int epi, const Q8Mat& w) {
    const int nm = epi == EPI_SWIGLU ? 2 : 1;
    return (size_t)w.ng * 4 * 32 + (size_t)w.ng * 4 * 4 + (pro == PRO_RMS ? (size_t)(w.k + 32) * 4 : 0) + (size_t)2 * nm * w.ng * 64 * 4 + 128;
}

// Instrumented steps (gl3_profile_decode) pass a start / stop event pair INTO the dispatch (hipExtLaunchKernel): the events
// then carry the kernel's own begin / end timestamps — the quantity rocprofv3 --kernel-trace reports — instead of the
// stream-order time between two recorded events, which also contains the host's launch latency in eager mode.
#include <hip/hip_ext.h>
template <typename K>
static void launch_mv(gl3_ctx* ctx, K kernel, int wgs, int threads, size_t smem, const MatvecArgs& a) {
    if (ctx->prof_ev0) hipExtLaunchKernelGGL(kernel, dim3(wgs), dim3(threads), (uint32_t)smem, ctx->stream, ctx->prof_ev0, ctx->prof_ev1, 0, a);
    else hipLaunchKernelGGL(kernel, dim3(wgs), dim3(threads), smem, ctx->stream, a);
}

template <int PRO, int EPI>
static void launch_matvec_t(gl3_ctx* ctx, const MatvecArgs& a, int wgs, size_t smem, bool nt) {
    if (a.tp) { launch_mv(ctx, matvec_q8t_kernel<PRO, EPI, true, 4, false, true>, wgs, mv_threads(4), smem, a); return; }   // folded gathers
    if (nt) launch_mv(ctx, matvec_q8t_kernel<PRO, EPI, true>, wgs, mv_threads(4), smem, a);
    else launch_mv(ctx, matvec_q8t_kernel<PRO, EPI, false>, wgs, mv_threads(4), smem, a);
}

template <int PRO, int EPI>
static hipError_t allow_big_lds() {
    hipError_t e = hipFuncSetAttribute((const void*)matvec_q8t_kernel<PRO, EPI, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute((const void*)matvec_q8t_kernel<PRO, EPI, true, 4, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256);
    if (e != hipSuccess) return e;
    return hipFuncSetAttribute((const void*)matvec_q8t_kernel<PRO, EPI, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256);
}

// rows_valid / out / resid_in may address a slice (tensor parallel row split); w holds exactly that slice.
static void launch_matvec(gl3_ctx* ctx, int pro, int epi, const Q8Mat& w, const Q8Mat* w2, const float* x,
                          const float* norm_w, float* out, const float* resid_in, float out_scale = 1.0f, const TpRec* tp = nullptr) {
    if (w.fmt != GL3_TYPE_Q8_0) {      // F16 / Q4_0: element-wise chains, one output row per lane (gl3_rowlane_kernels.h)
        hipStream_t s = ctx->stream;
        // Vector-order kernels normalise in their own prologue (matvec_vl_kernel<.., RMS>); GL3_VL_RMS=0: separate rmsnorm launch
        static const bool vl_rms = env_flag("GL3_VL_RMS", true);
        // ... except for the K-split types when the launch has few 8-row groups (a tensor-parallel rank's slice): the fused kernel
        // keeps one wavefront per group whatever the row count, so a 224-group gate/up slice would take as long as the whole
        // matrix; there the one-workgroup rmsnorm launch + the K-split kernel (16 wavefronts per group) is the faster pair.
        static const bool ksplit_on = env_flag("GL3_VLQ", true);
        const bool ksplit_type = ksplit_on && w.vl && (w.fmt == GL3_FMT_Q8V || w.fmt == GL3_TYPE_Q4_0);
        const long vl_waves = (long)((w.rows + 7) / 8) * (epi == EPI_SWIGLU ? 2 : 1);
        const bool fuse_rms = pro == PRO_RMS && w.vl && vl_rms && vl_rms_fusable(w.k) && epi != EPI_RESID && !(ksplit_type && vl_waves <= 512);
        if (pro == PRO_RMS && !fuse_rms) {
            const size_t sm = (size_t)(w.k + 32) * 4 + ss_scratch_bytes(w.k) + 64;
            hipLaunchKernelGGL(rmsnorm_f32_kernel, dim3(1), dim3(256), sm, s, x, w.k, norm_w, ctx->d.rms_eps, ctx->xn);
            x = ctx->xn;
        }
        if (w.vl) {                   // Vector-API order (8 accumulator lanes per row): lane = (row, accumulator), HBM-bound
            VlArgs v{};
            v.w = w.w; v.w2 = w2 ? w2->w : nullptr; v.rows = w.rows; v.k = w.k; v.x = x; v.out = out; v.resid_in = resid_in; v.out_scale = out_scale;
            v.tp = tp;      // folded gathers (gate/up -> hb, down -> x): producer side only, the consumers get wait launches
            const dim3 vg(((w.rows + 7) / 8 + VL_WAVES - 1) / VL_WAVES), vb(64 * VL_WAVES);
            const size_t vs = vl_smem_bytes(w.k);
            if (fuse_rms) {
                v.norm_w = norm_w; v.eps = ctx->d.rms_eps;
                const dim3 rg(((w.rows + 7) / 8 + VL_RMS_WAVES - 1) / VL_RMS_WAVES), rb(64 * VL_RMS_WAVES);
                const size_t rs = vl_rms_smem_bytes(w.k);
#define GL3_VLR(WT_) \
                do { \
                    if (epi == EPI_STORE) hipLaunchKernelGGL((matvec_vl_kernel<WT_, EPI_STORE, true, VL_RMS_WAVES>), rg, rb, rs, s, v); \
                    else hipLaunchKernelGGL((matvec_vl_kernel<WT_, EPI_SWIGLU, true, VL_RMS_WAVES>), rg, rb, rs, s, v); \
                } while (0)
                if (w.fmt == GL3_TYPE_F16 && (ctx->d.flags & GL3_FLAG_VECTOR_512)) {          // 16 accumulator lanes (FP16FloatTensor.vectorDot on a 512-bit species)
                    if (epi == EPI_STORE) hipLaunchKernelGGL((matvec_vl_kernel<WT_F16, EPI_STORE, true, VL_RMS_WAVES, 512>), rg, rb, rs, s, v);
                    else hipLaunchKernelGGL((matvec_vl_kernel<WT_F16, EPI_SWIGLU, true, VL_RMS_WAVES, 512>), rg, rb, rs, s, v);
                }
                else if (w.fmt == GL3_TYPE_F16) GL3_VLR(WT_F16); else if (w.fmt == GL3_FMT_Q8V) GL3_VLR(WT_Q8_0); else GL3_VLR(WT_Q4_0);
#undef GL3_VLR
                return;
            }
#define GL3_VL(WT_) \
            do { \
                if (epi == EPI_STORE) hipLaunchKernelGGL((matvec_vl_kernel<WT_, EPI_STORE>), vg, vb, vs, s, v); \
                else if (epi == EPI_RESID) hipLaunchKernelGGL((matvec_vl_kernel<WT_, EPI_RESID>), vg, vb, vs, s, v); \
                else hipLaunchKernelGGL((matvec_vl_kernel<WT_, EPI_SWIGLU>), vg, vb, vs, s, v); \
            } while (0)
            // Q4_0 / Q8_0-f32act: four wavefronts split K of one 8-row group (matvec_vlq_kernel); GL3_VLQ=0 keeps the one-wavefront kernel
            static const bool ksplit = env_flag("GL3_VLQ", true);
#define GL3_VLQ_E(WT_, MAXW_) \
            do { \
                const size_t qs = vq_smem_bytes<WT_, MAXW_>(w.k, epi == EPI_SWIGLU ? 2 : 1, nw); \
                if (epi == EPI_STORE) hipLaunchKernelGGL((matvec_vlq_kernel<WT_, EPI_STORE, MAXW_>), qg, qb, qs, s, v); \
                else if (epi == EPI_RESID) hipLaunchKernelGGL((matvec_vlq_kernel<WT_, EPI_RESID, MAXW_>), qg, qb, qs, s, v); \
                else hipLaunchKernelGGL((matvec_vlq_kernel<WT_, EPI_SWIGLU, MAXW_>), qg, qb, qs, s, v); \
            } while (0)
#define GL3_VLQ(WT_) \
            do { \
                const int ngroups = (w.rows + 7) / 8, nw = vq_waves<WT_>(w.k, epi == EPI_SWIGLU ? 2 : 1, ngroups); \
                const dim3 qg(ngroups), qb(64 * nw); \
                if (nw == 16) GL3_VLQ_E(WT_, 16); else GL3_VLQ_E(WT_, 8); \
            } while (0)
            if (w.fmt == GL3_TYPE_F16 && (ctx->d.flags & GL3_FLAG_VECTOR_512)) {
                if (epi == EPI_STORE) hipLaunchKernelGGL((matvec_vl_kernel<WT_F16, EPI_STORE, false, VL_WAVES, 512>), vg, vb, vs, s, v);
                else if (epi == EPI_RESID) hipLaunchKernelGGL((matvec_vl_kernel<WT_F16, EPI_RESID, false, VL_WAVES, 512>), vg, vb, vs, s, v);
                else hipLaunchKernelGGL((matvec_vl_kernel<WT_F16, EPI_SWIGLU, false, VL_WAVES, 512>), vg, vb, vs, s, v);
            }
            else if (w.fmt == GL3_TYPE_F16) GL3_VL(WT_F16);
            else if (w.fmt == GL3_FMT_Q8V) { if (ksplit) GL3_VLQ(WT_Q8_0); else GL3_VL(WT_Q8_0); }
            else { if (ksplit) GL3_VLQ(WT_Q4_0); else GL3_VL(WT_Q4_0); }
#undef GL3_VLQ
#undef GL3_VLQ_E
#undef GL3_VL
            return;
        }
        RlArgs a{};
        a.w = w.w; a.w2 = w2 ? w2->w : nullptr; a.rows = w.rows; a.k = w.k; a.x = x; a.out = out; a.resid_in = resid_in; a.out_scale = out_scale;
        const dim3 grid((w.rows + 63) / 64);
        const size_t sm = rl_smem_bytes(w.k, epi);
#define GL3_RL(WT_) \
        do { \
            if (epi == EPI_STORE) hipLaunchKernelGGL((matvec_rl_kernel<WT_, EPI_STORE>), grid, dim3(RL_THREADS), sm, s, a); \
            else if (epi == EPI_RESID) hipLaunchKernelGGL((matvec_rl_kernel<WT_, EPI_RESID>), grid, dim3(RL_THREADS), sm, s, a); \
            else hipLaunchKernelGGL((matvec_rl_kernel<WT_, EPI_SWIGLU>), grid, dim3(RL_THREADS), sm, s, a); \
        } while (0)
        if (w.fmt == GL3_TYPE_F16) GL3_RL(WT_F16); else GL3_RL(WT_Q4_0);
#undef GL3_RL
        return;
    }
    static const bool nt = env_flag("GL3_NT", true);
    static const int max_wgs = getenv("GL3_WGS") ? atoi(getenv("GL3_WGS")) : 512;   // 2 resident workgroups per CU
    MatvecArgs a{};
    a.w = w.w; a.w2 = w2 ? w2->w : nullptr; a.rows = w.rows; a.k = w.k; a.ng = w.ng; a.nstrips = w.nstrips;
    a.x = x; a.norm_w = norm_w; a.eps = ctx->d.rms_eps; a.out = out; a.resid_in = resid_in; a.out_scale = out_scale;
    a.tp = tp;
    const int wgs = w.nstrips < max_wgs ? w.nstrips : max_wgs;
    const size_t smem = matvec_smem(pro, epi, w);
    if (pro == PRO_RMS && epi == EPI_STORE) launch_matvec_t<PRO_RMS, EPI_STORE>(ctx, a, wgs, smem, nt);
    else if (pro == PRO_QUANT && epi == EPI_RESID) {
        static const int wide_max = getenv("GL3_WIDE_STRIPS") ? atoi(getenv("GL3_WIDE_STRIPS")) : 256;
        if (w.nstrips <= wide_max && nt && tp)
            launch_mv(ctx, matvec_q8t_kernel<PRO_QUANT, EPI_RESID, true, 8, false, true>, wgs, mv_threads(8), smem, a);
        else if (w.nstrips <= wide_max && nt)      // one workgroup per CU: 8 producer wavefronts
            launch_mv(ctx, matvec_q8t_kernel<PRO_QUANT, EPI_RESID, true, 8>, wgs, mv_threads(8), smem, a);
        else launch_matvec_t<PRO_QUANT, EPI_RESID>(ctx, a, wgs, smem, nt);
    }
    else launch_matvec_t<PRO_RMS, EPI_SWIGLU>(ctx, a, wgs, smem, nt);
}

// Routed experts of a Qwen2-MoE layer: one launch, blockIdx.y = slot of the top-k selection (matvec_q8t_kernel<.., SEL>).
// stack holds n_experts sub-matrices of `rows` rows each; x / out advance by x_slot / out_slot floats per slot.
// which: record of this layer (0 = gate/up with the shared expert's chunks as extra slots, 1 = gate/up alone, 2 = down).
static void launch_matvec_sel(gl3_ctx* ctx, int layer, int which, const Q8Mat& stack, const Q8Mat* stack2, int rows, const float* x,
                              const float* norm_w, float* out, int shared_rows) {
    const gl3_model_desc& d = ctx->d;
    static const int max_wgs = getenv("GL3_WGS") ? atoi(getenv("GL3_WGS")) : 512;
    Q8Mat sub = stack;
    sub.rows = rows; sub.nstrips = rows / 16;
    MatvecArgs a{};
    a.w = stack.w; a.w2 = stack2 ? stack2->w : nullptr; a.rows = rows; a.k = stack.k; a.ng = stack.ng; a.nstrips = sub.nstrips;
    a.x = x; a.norm_w = norm_w; a.eps = d.rms_eps; a.out = out; a.resid_in = nullptr; a.out_scale = 1.0f;
    a.moe = reinterpret_cast<const MoeSlots*>(ctx->moe_slots) + (size_t)layer * 3 + which;
    const int slots = d.n_experts_used + (which == 0 ? (shared_rows + rows - 1) / rows : 0);
    const int wgs = sub.nstrips < max_wgs ? sub.nstrips : max_wgs;
    const dim3 grid(wgs, slots);
    if (which != 2) hipLaunchKernelGGL((matvec_q8t_kernel<PRO_RMS, EPI_SWIGLU, true, 4, true>), grid, dim3(mv_threads(4)), matvec_smem(PRO_RMS, EPI_SWIGLU, sub), ctx->stream, a);
    else hipLaunchKernelGGL((matvec_q8t_kernel<PRO_QUANT, EPI_RESID, true, 4, true>), grid, dim3(mv_threads(4)), matvec_smem(PRO_QUANT, EPI_RESID, sub), ctx->stream, a);
}

// The slot records of every layer's two routed-expert launches (MoeSlots, gl3_decode_kernels.h); pointers are final after gl3_create.
static int32_t moe_build_slots(gl3_ctx* ctx) {
    const gl3_model_desc& d = ctx->d;
    std::vector<MoeSlots> h((size_t)d.n_layers * 3);
    for (int l = 0; l < d.n_layers; ++l) {
        const gl3_layer& L = ctx->layers[l];
        MoeSlots gu{};
        gu.sel = ctx->moe_sel; gu.sel_stride = (size_t)(d.moe_hidden / 16) * L.gate_exps.ng * TILE_BYTES;
        gu.x_slot_stride = 0; gu.out_slot_stride = d.moe_hidden; gu.n_sel = d.n_experts_used;
        gu.sh_rows = L.w1.rows; gu.sh_w = L.w1.w; gu.sh_w2 = L.w3.w; gu.sh_out = ctx->hb;
        h[(size_t)l * 3 + 0] = gu;
        gu.sh_rows = 0; gu.sh_w = gu.sh_w2 = nullptr; gu.sh_out = nullptr;
        h[(size_t)l * 3 + 1] = gu;
        MoeSlots dn{};
        dn.sel = ctx->moe_sel; dn.sel_stride = (size_t)(d.dim / 16) * L.down_exps.ng * TILE_BYTES;
        dn.x_slot_stride = d.moe_hidden; dn.out_slot_stride = d.dim; dn.n_sel = d.n_experts_used;
        h[(size_t)l * 3 + 2] = dn;
    }
    GL3_HIP(hipMalloc(&ctx->moe_slots, h.size() * sizeof(MoeSlots)));
    GL3_HIP(hipMemcpy(ctx->moe_slots, h.data(), h.size() * sizeof(MoeSlots), hipMemcpyHostToDevice));
    return GL3_OK;
}

// ------------------------------------------------------------------------------------------------ decode step
struct Prof {
    gl3_ctx* ctx; gl3_kernel_times* kt; size_t n = 0; std::vector<int> klass;
    bool ext = false;
    // single_kernel: the class is exactly one Q8_0 matvec launch -> kernel begin / end timestamps (see launch_mv)
    void begin(int k, uint64_t bytes, bool single_kernel = false) {
        if (!kt) return;
        if (ctx->ev.size() < 2 * (n + 1)) {
            ctx->ev.resize(2 * (n + 1));
            hipEventCreate(&ctx->ev[2 * n]); hipEventCreate(&ctx->ev[2 * n + 1]);
        }
        ext = single_kernel;
        if (ext) { ctx->prof_ev0 = ctx->ev[2 * n]; ctx->prof_ev1 = ctx->ev[2 * n + 1]; }
        else hipEventRecord(ctx->ev[2 * n], ctx->stream);
        klass.push_back(k); kt->launches[k]++; kt->bytes[k] += bytes;
    }
    void end() {
        if (!kt) return;
        if (ext) ctx->prof_ev0 = ctx->prof_ev1 = nullptr;
        else hipEventRecord(ctx->ev[2 * n + 1], ctx->stream);
        ++n;
    }
    void collect() {
        if (!kt) return;
        hipStreamSynchronize(ctx->stream);
        for (size_t i = 0; i < n; ++i) {
            float ms = 0; hipEventElapsedTime(&ms, ctx->ev[2 * i], ctx->ev[2 * i + 1]);
            kt->ms[klass[i]] += ms;
        }
    }
};

static uint64_t mv_bytes(const Q8Mat& w) { return w.algo_bytes() + (uint64_t)w.k * 4 + (uint64_t)w.rows * 4; }

// The MoE feed-forward block of one decode step — InferenceCore.forwardJavaQwen2MoE :363-415 (kernels: gl3_moe_kernels.h).
// x is read by every projection (each normalises it in its own prologue) and only rewritten by the final combine launch, so the
// order of the launches in between is free; the accumulation order into x is the reference's (selection order, then shared).
static void enqueue_moe_ffn(gl3_ctx* ctx, int l, Prof& pr) {
    gl3_layer& L = ctx->layers[l];
    const gl3_model_desc& d = ctx->d;
    hipStream_t s = ctx->stream;
    const int E = d.n_experts, topk = d.n_experts_used, mh = d.moe_hidden;
    const uint64_t row34 = (uint64_t)(d.dim / 32) * 34;
    // GL3_MOE_MERGE_SHARED=0: the shared expert's gate/up as its own launch instead of extra slots of the routed experts' launch
    static const bool merge_shared = env_flag("GL3_MOE_MERGE_SHARED", true);
    pr.begin(GL3_K_OTHER, (uint64_t)(E + 1) * d.dim * 4 + d.dim * 8);
    {   Gl3Range g("moe: rmsnorm + router + top-k");
        MoeRouterArgs ra{};
        ra.x = ctx->x; ra.norm_w = L.ffn_norm; ra.eps = d.rms_eps; ra.gate_inp = L.gate_inp; ra.gate_inp_shexp = L.gate_inp_shexp;
        ra.dim = d.dim; ra.n_experts = E; ra.topk = topk; ra.logits = ctx->moe_logits; ra.w_out = ctx->moe_w; ra.sel = ctx->moe_sel;
        ra.ticket = ctx->moe_sel + topk;
        hipLaunchKernelGGL(moe_router_kernel, dim3(moe_router_wgs(E)), dim3(256), moe_router_smem(d.dim, E), s, ra); }
    pr.end();
    pr.begin(GL3_K_MATVEC_GATEUP, (uint64_t)2 * topk * mh * row34 + mv_bytes(L.w1) + L.w3.algo_bytes() + d.dim * 4);
    {   Gl3Range g("moe: expert + shared gate/up + swiglu");
        if (merge_shared) launch_matvec_sel(ctx, l, 0, L.gate_exps, &L.up_exps, mh, ctx->x, L.ffn_norm, ctx->moe_hb, L.w1.rows);
        else {
            launch_matvec_sel(ctx, l, 1, L.gate_exps, &L.up_exps, mh, ctx->x, L.ffn_norm, ctx->moe_hb, 0);
            launch_matvec(ctx, PRO_RMS, EPI_SWIGLU, L.w1, &L.w3, ctx->x, L.ffn_norm, ctx->hb, nullptr);
        } }
    pr.end();
    pr.begin(GL3_K_MATVEC_DOWN, (uint64_t)topk * d.dim * (mh / 32) * 34 + mv_bytes(L.w2));
    {   Gl3Range g("moe: expert + shared down");
        launch_matvec_sel(ctx, l, 2, L.down_exps, nullptr, d.dim, ctx->moe_hb, nullptr, ctx->moe_y, 0);
        launch_matvec(ctx, PRO_QUANT, EPI_RESID, L.w2, nullptr, ctx->hb, nullptr, ctx->moe_y + (size_t)topk * d.dim, nullptr); }
    pr.end();
    pr.begin(GL3_K_OTHER, (uint64_t)(topk + 3) * d.dim * 4);
    {   Gl3Range g("moe: weighted accumulation into x");
        hipLaunchKernelGGL(moe_combine_kernel, dim3((d.dim + 255) / 256), dim3(256), 0, s, ctx->x, ctx->moe_y, ctx->moe_w, d.dim, topk + 1); }
    pr.end();
}

// ---- folded gathers (TpRec, gl3_decode_kernels.h; protocol notes in gl3_tp.hip).  Records per layer + embedding + logits, built
// once the peers' arenas are known (gl3_finalize).  ctx->tp_fold: 0 = gather kernels (every type but the Q8_0 int8 path, RCCL
// transport, GL3_TP_FOLD=0), 1 = producers push; the consumer side is a one-wavefront wait launch (GL3_TP_FOLD=1) or the consumer's
// own prologue (GL3_TP_FOLD=2: no launch is left between producer and consumer; ranks must not share a GPU once the models are
// large enough for a polling consumer to fill it), per consumer kind in ctx->tp_fold_mask.
enum { TR_ATTN = 0, TR_WO, TR_GATEUP, TR_DOWN, TR_QKV, TR_WAIT_XB, TR_WAIT_HB, TR_WAIT_X, TR_PER_LAYER };
enum { TF_WO = 1, TF_DOWN = 2, TF_QKV = 4, TF_LOGITS = 8, TF_EMBED = 16 };
static int32_t tp_fold_setup(gl3_ctx* ctx) {
    const gl3_model_desc& d = ctx->d;
    ctx->tp_fold = 0;
    const int mode = getenv("GL3_TP_FOLD") ? atoi(getenv("GL3_TP_FOLD")) : 1;
    if (!ctx->use_rccl || ctx->transport != GL3_TP_P2P || d.tp_size < 2 || mode <= 0) return GL3_OK;
    // Q8_0 int8 path, and (r6) the vector-order types F16 / Q4_0 / Q8_0-f32act (matvec_vl_kernel / matvec_vlq_kernel push and publish; their consumers
    // always wait in a wait launch: mode 1).  The scalar-order kernels (GL3_FLAG_SCALAR_DOT) keep the gather kernels.
    const bool vl_types = ctx->emb.vl;
    if ((ctx->emb.fmt != GL3_TYPE_Q8_0 && !vl_types) || !ctx->wo_replicated || d.arch == GL3_ARCH_QWEN2MOE || env_flag("GL3_TP_DEBUG", false)) return GL3_OK;
    // the producers publish without a per-wavefront release fence: that is only sound on an UNCACHED arena (advisor finding): any other arena kind
    // (GL3_TP_ARENA=cached | finegrained | unpooled experiments) keeps the gather kernels
    if (ctx->arena.kind != 0) return GL3_OK;
    const int L = d.n_layers, tp = d.tp_size, me = d.tp_rank;
    // which consumers wait in their own prologue (the others get a wait launch in front): GL3_TP_FOLD=2 all, GL3_TP_FOLD_MASK picks
    int mask = mode >= 2 ? (TF_WO | TF_DOWN | TF_QKV | TF_LOGITS | TF_EMBED) : 0;
    if (getenv("GL3_TP_FOLD_MASK")) mask = atoi(getenv("GL3_TP_FOLD_MASK"));
    if (vl_types) mask = 0;
    uint8_t* own = ctx->arena.base;
    static const unsigned limit = getenv("GL3_TP_SPIN_LIMIT") ? (unsigned)atol(getenv("GL3_TP_SPIN_LIMIT")) : 20000000u;
    uint32_t* step = reinterpret_cast<uint32_t*>(own + GL3_ARENA_STEP);
    auto flags_of = [&](uint8_t* base, int buf) { return reinterpret_cast<uint32_t*>(base + GL3_ARENA_FOLD + (size_t)buf * GL3_MAX_TP * 4); };
    auto mk_wait = [&](int buf, int add) {
        TpWait w{};
        w.flags = flags_of(own, buf); w.step = step; w.err = ctx->h_tp_err; w.mul = L; w.add = add; w.tp = tp; w.me = me; w.spin_limit = limit;
        return w;
    };
    auto mk_push = [&](int buf, int add) {
        TpPush p{};
        p.ticket = reinterpret_cast<uint32_t*>(own + GL3_ARENA_FOLD_TICKET) + buf; p.step = step; p.mul = L; p.add = add; p.npeers = tp - 1;
        for (int j = 0; j < tp - 1; ++j) {
            uint8_t* peer = ctx->peer_base[(me + 1 + j) % tp];
            p.delta[j] = (long)(peer - own);
            p.flag[j] = flags_of(peer, buf) + me;
        }
        return p;
    };
    std::vector<TpRec> h((size_t)L * TR_PER_LAYER + 2);
    for (int l = 0; l < L; ++l) {
        TpRec* r = h.data() + (size_t)l * TR_PER_LAYER;
        r[TR_ATTN].p = mk_push(GB_XB, l + 1);
        r[TR_GATEUP].p = mk_push(GB_HB, l + 1);
        r[TR_DOWN].p = mk_push(GB_X, l + 1);
        if (mask & TF_WO) r[TR_WO].w = mk_wait(GB_XB, l + 1);
        if (mask & TF_DOWN) r[TR_DOWN].w = mk_wait(GB_HB, l + 1);
        if ((mask & TF_QKV) && l > 0) r[TR_QKV].w = mk_wait(GB_X, l);
        r[TR_WAIT_XB].w = mk_wait(GB_XB, l + 1); r[TR_WAIT_HB].w = mk_wait(GB_HB, l + 1); r[TR_WAIT_X].w = mk_wait(GB_X, l + 1);
    }
    h[(size_t)L * TR_PER_LAYER].w = mk_wait(GB_X, 0);          // embedding: the previous step's last pushes into x
    if (mask & TF_LOGITS) h[(size_t)L * TR_PER_LAYER + 1].w = mk_wait(GB_X, L);
    GL3_HIP(hipMalloc((void**)&ctx->tp_recs, h.size() * sizeof(TpRec)));
    // on the plan's own stream: a legacy-stream copy would have to wait for every blocking stream of the process, and another rank of an
    // in-process group may already be CAPTURING its step graph on one (hipErrorStreamCaptureImplicit, seen as a flaky 8-rank test)
    GL3_HIP(hipMemcpyAsync(ctx->tp_recs, h.data(), h.size() * sizeof(TpRec), hipMemcpyHostToDevice, ctx->stream));
    GL3_HIP(hipStreamSynchronize(ctx->stream));
    ctx->tp_fold = 1; ctx->tp_fold_mask = mask;
    return GL3_OK;
}
static const TpRec* tp_rec(const gl3_ctx* ctx, int l, int which) {
    if (ctx->tp_quiet) return nullptr;      // gl3_profile_kernel: launches outside the step protocol must not push into the peers' arenas
    return ctx->tp_fold ? reinterpret_cast<const TpRec*>(ctx->tp_recs) + (size_t)l * TR_PER_LAYER + which : nullptr;
}
// mode 1: the consumer side of a folded gather as its own one-wavefront launch
static void fold_wait(gl3_ctx* ctx, int l, int which, Prof& pr) {
    const int m = ctx->tp_fold_mask, last = l == ctx->d.n_layers - 1;
    if (which == TR_WAIT_XB && (m & TF_WO)) return;
    if (which == TR_WAIT_HB && (m & TF_DOWN)) return;
    if (which == TR_WAIT_X && !last && (m & TF_QKV)) return;
    if (which == TR_WAIT_X && last && (m & TF_LOGITS) && (m & TF_EMBED)) return;
    pr.begin(GL3_K_COLLECTIVE, 0);
    hipLaunchKernelGGL(tp_wait_kernel, dim3(1), dim3(64), 0, ctx->stream, tp_rec(ctx, l, which));
    pr.end();
}

static int32_t all_gather(gl3_ctx* ctx, int which, int count_per_rank, Prof& pr) {
    if (!ctx->use_rccl) return GL3_OK;
    pr.begin(GL3_K_COLLECTIVE, 0);
    const int32_t r = gl3_all_gather(ctx, which, (size_t)count_per_rank);
    if (r != GL3_OK) return r;
    pr.end();
    return GL3_OK;
}

// Decode attention by context depth (the host knows the position): ATT_SHORT = one launch (attn_head_kernel, positions < AF_MAXN),
// ATT_MID = scores + the r2 softmax-and-PV kernel (two launches; below ~768 positions two more ~5 us launches cost more than the chains
// they shorten: tg128@d256 18.5 vs 21.8 us per 8B layer, profiles/r05_tg_depth.md), ATT_LONG = scores, exp, sum, PV (gl3_decode_kernels.h).
enum { ATT_LONG = 0, ATT_SHORT = 1, ATT_MID = 2 };
static int attn_mode(const gl3_ctx* ctx, int pos) {
    return (ctx->fused_attn_ok && pos < AF_MAXN) ? ATT_SHORT : pos < ctx->attn_mid ? ATT_MID : ATT_LONG;
}

static void launch_attention(gl3_ctx* ctx, int l, int which /* 0 both, 1 scores, 2 softmax+pv */, int amode = ATT_LONG) {
    const gl3_model_desc& d = ctx->d;
    gl3_layer& L = ctx->layers[l];
    const size_t kv_layer = (size_t)d.ctx * ctx->kv_dim_l;
    const int kvmul = d.n_heads / d.n_kv_heads;
    AttnArgs aa{};
    aa.qkv = ctx->qkv; aa.kcache = ctx->kcache + l * kv_layer; aa.vcache = ctx->vcache + l * kv_layer;
    aa.rope_cr = ctx->rope_cr; aa.rope_ci = ctx->rope_ci; aa.qnorm = L.qnorm; aa.knorm = L.knorm; aa.bq = L.bq; aa.bk = L.bk; aa.bv = L.bv;
    aa.dyn = ctx->dyn_cur; aa.att = ctx->att; aa.xb = ctx->xb + (size_t)d.tp_rank * ctx->q_dim_l;
    aa.n_heads = ctx->heads_l; aa.n_kv_heads = ctx->kv_heads_l;
    aa.hs = d.head_size; aa.q_dim = ctx->q_dim_l; aa.kv_dim = ctx->kv_dim_l; aa.ctx = d.ctx;
    aa.eps = d.rms_eps; aa.arch = ctx->rope_arch; aa.att_mul = ctx->att_mul;
    const size_t sm1 = ((size_t)kvmul * d.head_size + (size_t)ATT_TT * (d.head_size + 4) + d.head_size) * 4;
    const int pv_rows = d.ctx < PV_ROWS ? d.ctx : PV_ROWS;
    aa.win = ctx->attn_win;
    aa.att_stride = (d.ctx + 3) & ~3;
    aa.att_t = ctx->att_t; aa.tmax = ctx->att_tmax; aa.sums = ctx->att_sums;
    aa.tp = tp_rec(ctx, l, TR_ATTN);
    const size_t sm2 = ((size_t)ctx->attn_win + (size_t)pv_rows * PV_COLS) * 4;
    if (which == 0 && amode == ATT_SHORT && ctx->fused_attn_ok) {      // positions < AF_MAXN: one launch
        attn_head_dispatch(d.head_size, [&](auto kern) { hipLaunchKernelGGL(kern, dim3(ctx->heads_l), dim3(256), attn_head_smem(d.head_size), ctx->stream, aa); });
        return;
    }
    // r6 experiment: positions AF_MAXN .. 767 in one launch (attn_mid_kernel): measured slower than the pair (gl3_decode_kernels.h), off unless GL3_ATTN_FUSED_MID=1
    static const bool fused_mid = env_flag("GL3_ATTN_FUSED_MID", false);
    if (which == 0 && amode == ATT_MID && fused_mid && kvmul <= 4 && ctx->attn_mid <= AM_MAXN && (d.head_size == 128 || d.head_size == 64)) {
        const dim3 mg(d.head_size / 16, ctx->kv_heads_l);
        if (d.head_size == 128) hipLaunchKernelGGL(attn_mid_kernel<128>, mg, dim3(256), attn_mid_smem<128>(), ctx->stream, aa);
        else hipLaunchKernelGGL(attn_mid_kernel<64>, mg, dim3(256), attn_mid_smem<64>(), ctx->stream, aa);
        return;
    }
    // GL3_ATTN_SCORES_LOOP=0: the one-tile-per-workgroup scores kernel at every depth (A/B switch)
    static const bool scores_loop = env_flag("GL3_ATTN_SCORES_LOOP", true);
    if (which != 2 && amode == ATT_LONG && scores_loop && kvmul <= 4 && (d.head_size == 128 || d.head_size == 64)) {
        const dim3 sg(ctx->n_tsplit < SCL_WGS ? ctx->n_tsplit : SCL_WGS, ctx->kv_heads_l), sb(64 * (kvmul + SCL_LOADERS));
        const size_t sml = ((size_t)kvmul * d.head_size + 2 * (size_t)ATT_TT * (d.head_size + 4) + 2 * d.head_size) * 4;
        if (d.head_size == 128) hipLaunchKernelGGL(attn_scores_loop_kernel<128>, sg, sb, sml, ctx->stream, aa, ctx->n_tsplit);
        else hipLaunchKernelGGL(attn_scores_loop_kernel<64>, sg, sb, sml, ctx->stream, aa, ctx->n_tsplit);
    }
    else if (which != 2) hipLaunchKernelGGL(attn_scores_kernel, dim3(ctx->n_tsplit, ctx->kv_heads_l), dim3(64 * kvmul), sm1, ctx->stream, aa);
    if (which != 1 && amode == ATT_LONG) {
        const int exp_rows = (d.ctx + EXP_ROW - 1) / EXP_ROW;
        hipLaunchKernelGGL(attn_exp_kernel, dim3(exp_rows < EXP_GRID_MAX ? exp_rows : EXP_GRID_MAX, ctx->heads_l), dim3(256), 0, ctx->stream, aa, ctx->n_tsplit);
        hipLaunchKernelGGL(attn_sum_kernel, dim3(ctx->heads_l), dim3(256), attn_sum_smem(), ctx->stream, aa);
        hipLaunchKernelGGL(attn_pv_kernel, dim3(ctx->kv_heads_l * attn_pv_hq(kvmul) * (d.head_size / PV_COLS16)), dim3(64 * PV_WAVES), attn_pv_smem(), ctx->stream, aa);
    } else if (which != 1)
        hipLaunchKernelGGL(attn_softmax_pv_kernel, dim3(ctx->heads_l * (d.head_size / PV_COLS)), dim3(256), sm2, ctx->stream, aa);
}

// amode: attn_mode(position) — the host knows the position of the step
static int32_t enqueue_decode(gl3_ctx* ctx, bool want_logits, gl3_kernel_times* kt, int amode) {
    const gl3_model_desc& d = ctx->d;
    hipStream_t s = ctx->stream;
    Prof pr{ctx, kt};
    const int rank = d.tp_rank;
    const bool q8 = ctx->emb.fmt == GL3_TYPE_Q8_0;       // Q8T path (int8 activation)
    int32_t r;
    Gl3Range step_range("gl3 decode step");

    pr.begin(GL3_K_OTHER, (uint64_t)d.dim / 32 * 34);
    const int fold = ctx->tp_fold, fmask = fold ? ctx->tp_fold_mask : 0, L_ = d.n_layers;
    uint32_t* vl_step = fold ? reinterpret_cast<uint32_t*>(ctx->arena.base + GL3_ARENA_STEP) : (uint32_t*)nullptr;
    if (ctx->emb.fmt == GL3_TYPE_Q8_0)
        hipLaunchKernelGGL(embed_q8t_kernel, dim3(1), dim3(256), 0, s, ctx->emb.w, ctx->emb.ng, d.dim, ctx->dyn_cur, ctx->x, ctx->emb_scale,
                           (fmask & TF_EMBED) ? tp_rec(ctx, L_, 0) : (const TpRec*)nullptr,
                           fold ? reinterpret_cast<uint32_t*>(ctx->arena.base + GL3_ARENA_STEP) : (uint32_t*)nullptr);
    else if (ctx->emb.vl && ctx->emb.fmt == GL3_TYPE_F16) hipLaunchKernelGGL((embed_vl_kernel<WT_F16>), dim3(1), dim3(256), 0, s, ctx->emb.w, d.dim, ctx->dyn_cur, ctx->x, ctx->emb_scale, vl_step);
    else if (ctx->emb.vl && ctx->emb.fmt == GL3_FMT_Q8V) hipLaunchKernelGGL((embed_vl_kernel<WT_Q8_0>), dim3(1), dim3(256), 0, s, ctx->emb.w, d.dim, ctx->dyn_cur, ctx->x, ctx->emb_scale, vl_step);
    else if (ctx->emb.vl) hipLaunchKernelGGL((embed_vl_kernel<WT_Q4_0>), dim3(1), dim3(256), 0, s, ctx->emb.w, d.dim, ctx->dyn_cur, ctx->x, ctx->emb_scale, vl_step);
    else if (ctx->emb.fmt == GL3_TYPE_F16) hipLaunchKernelGGL((embed_rl_kernel<WT_F16>), dim3(1), dim3(256), 0, s, ctx->emb.w, d.dim, ctx->dyn_cur, ctx->x, ctx->emb_scale);
    else hipLaunchKernelGGL((embed_rl_kernel<WT_Q4_0>), dim3(1), dim3(256), 0, s, ctx->emb.w, d.dim, ctx->dyn_cur, ctx->x, ctx->emb_scale);
    pr.end();

    for (int l = 0; l < d.n_layers; ++l) {
        gl3_layer& L = ctx->layers[l];
        Gl3Range layer_range("layer", l);
        pr.begin(GL3_K_MATVEC_QKV, mv_bytes(L.wqkv) + d.dim * 4, q8);
        { Gl3Range g("rmsnorm + qkv"); launch_matvec(ctx, PRO_RMS, EPI_STORE, L.wqkv, nullptr, ctx->x, L.attn_norm, ctx->qkv, nullptr, 1.0f, (fmask & TF_QKV) && l > 0 ? tp_rec(ctx, l, TR_QKV) : nullptr); }
        pr.end();

        pr.begin(GL3_K_ATTENTION, 0);
        { Gl3Range g("rope + kv write + attention"); launch_attention(ctx, l, 0, amode); }
        pr.end();
        if (fold) fold_wait(ctx, l, TR_WAIT_XB, pr);
        else if ((r = all_gather(ctx, GB_XB, ctx->q_dim_l, pr)) != GL3_OK) return r;

        // x[rows of this rank] += Wo[rows, :] . xb — all rows on every rank when Wo is replicated (no gather behind it)
        const size_t wo_off = ctx->wo_replicated ? 0 : (size_t)rank * ctx->dim_l;
        pr.begin(GL3_K_MATVEC_WO, mv_bytes(L.wo), q8);
        { Gl3Range g("wo + residual"); launch_matvec(ctx, PRO_QUANT, EPI_RESID, L.wo, nullptr, ctx->xb, nullptr, ctx->x + wo_off, ctx->x + wo_off, ctx->resid_scale,
                                                        (fmask & TF_WO) ? tp_rec(ctx, l, TR_WO) : nullptr); }
        pr.end();
        if (!ctx->wo_replicated && (r = all_gather(ctx, GB_X, ctx->dim_l, pr)) != GL3_OK) return r;

        if (d.arch == GL3_ARCH_QWEN2MOE) {
            enqueue_moe_ffn(ctx, l, pr);
            if (ctx->taps) hipMemcpyAsync(ctx->taps + (size_t)l * d.dim, ctx->x, sizeof(float) * d.dim, hipMemcpyDeviceToDevice, s);
            continue;
        }
        pr.begin(GL3_K_MATVEC_GATEUP, mv_bytes(L.w1) + L.w3.algo_bytes() + d.dim * 4, q8);
        { Gl3Range g("rmsnorm + gate/up + swiglu"); launch_matvec(ctx, PRO_RMS, EPI_SWIGLU, L.w1, &L.w3, ctx->x, L.ffn_norm, ctx->hb + (size_t)rank * ctx->hidden_l, nullptr, 1.0f,
                                                                  tp_rec(ctx, l, TR_GATEUP)); }
        pr.end();
        if (fold) fold_wait(ctx, l, TR_WAIT_HB, pr);
        else if ((r = all_gather(ctx, GB_HB, ctx->hidden_l, pr)) != GL3_OK) return r;

        pr.begin(GL3_K_MATVEC_DOWN, mv_bytes(L.w2), q8);
        { Gl3Range g("down + residual");
          launch_matvec(ctx, PRO_QUANT, EPI_RESID, L.w2, nullptr, ctx->hb, nullptr, ctx->x + (size_t)rank * ctx->dim_l,
                        ctx->x + (size_t)rank * ctx->dim_l, ctx->resid_scale, tp_rec(ctx, l, TR_DOWN)); }
        pr.end();
        if (fold) fold_wait(ctx, l, TR_WAIT_X, pr);
        else if ((r = all_gather(ctx, GB_X, ctx->dim_l, pr)) != GL3_OK) return r;
        if (ctx->taps) hipMemcpyAsync(ctx->taps + (size_t)l * d.dim, ctx->x, sizeof(float) * d.dim, hipMemcpyDeviceToDevice, s);
    }
    if (want_logits) {
        pr.begin(GL3_K_MATVEC_LOGITS, mv_bytes(ctx->wcls) + d.dim * 4, q8);
        { Gl3Range g("final rmsnorm + logits");
          launch_matvec(ctx, PRO_RMS, EPI_STORE, ctx->wcls, nullptr, ctx->x, ctx->out_norm,
                        ctx->logits + (size_t)rank * ctx->vocab_l, nullptr, ctx->logit_scale, (fmask & TF_LOGITS) ? tp_rec(ctx, L_, 1) : nullptr); }
        pr.end();
        if ((r = all_gather(ctx, GB_LOGITS, ctx->vocab_l, pr)) != GL3_OK) return r;
    }
    GL3_HIP(hipGetLastError());
    pr.collect();
    return GL3_OK;
}

// ------------------------------------------------------------------------------------------------ create / destroy
template <typename T>
static int32_t dmalloc(gl3_ctx* ctx, T** p, size_t n) {
    GL3_HIP(hipMalloc((void**)p, n * sizeof(T)));
    return GL3_OK;
}

static int32_t alloc_mat(gl3_ctx* ctx, Q8Mat& m, int rows, int k) {
    m.rows = rows; m.k = k; m.ng = ((k / 32) + 3) / 4; m.nstrips = (rows + 15) / 16; m.fmt = ctx->d.weight_type;
    if (m.fmt == GL3_TYPE_Q8_0 && (ctx->d.flags & GL3_FLAG_F32_ACTIVATION)) m.fmt = GL3_FMT_Q8V;
    m.vl = m.fmt == GL3_FMT_Q8V || (m.fmt != GL3_TYPE_Q8_0 && !(ctx->d.flags & GL3_FLAG_SCALAR_DOT));
    GL3_HIP(hipMalloc((void**)&m.w, m.bytes() + GL3_TAIL_PAD));
    if (m.fmt != GL3_TYPE_Q8_0) GL3_HIP(hipMemset(m.w, 0, m.bytes()));      // padded rows of the last 64-row group
    return GL3_OK;
}

__global__ __launch_bounds__(256) void debug_sumsq_kernel(const float* __restrict__ x, int n, float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    float* xf = reinterpret_cast<float*>(smem);
    uint8_t* scratch = smem + (size_t)(n + 32) * 4;
    for (int i = threadIdx.x; i < n + 32; i += 256) xf[i] = i < n ? x[i] : 0.f;
    __syncthreads();
    BlockBarrier bb;
    const float s = exact_sumsq_lds(xf, n, scratch, threadIdx.x, bb);
    if (threadIdx.x == 0) *out = s;
}

extern "C" {

int32_t gl3_debug_sumsq(int32_t device, const float* x, int32_t n, float* out) {
    if (!x || !out || n < 1024 || n > 5120 || (n & 3)) return GL3_E_ARG;
    if (hipSetDevice(device) != hipSuccess) return GL3_E_HIP;
    float *dx = nullptr, *dout = nullptr;
    if (hipMalloc((void**)&dx, (size_t)n * 4) != hipSuccess || hipMalloc((void**)&dout, 4) != hipSuccess) return GL3_E_OOM;
    hipMemcpy(dx, x, (size_t)n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(debug_sumsq_kernel, dim3(1), dim3(256), (size_t)(n + 32) * 4 + ss_scratch_bytes(n), 0, dx, n, dout);
    const hipError_t e = hipMemcpy(out, dout, 4, hipMemcpyDeviceToHost);
    hipFree(dx); hipFree(dout);
    return e == hipSuccess ? GL3_OK : GL3_E_HIP;
}

const char* gl3_version(void) { return "gpullama3-hip 0.1 (gfx950)"; }

const char* gl3_last_error(gl3_ctx* ctx) { return ctx ? ctx->err.c_str() : g_create_err.c_str(); }

int32_t gl3_create(const gl3_model_desc* desc, gl3_ctx** out) {
    if (!desc || !out) { g_create_err = "null argument"; return GL3_E_ARG; }
    if (desc->struct_size != sizeof(gl3_model_desc)) { g_create_err = "gl3_model_desc.struct_size mismatch"; return GL3_E_ARG; }
    gl3_ctx* ctx = new gl3_ctx();
    ctx->d = *desc;
    ctx->n_seqs = desc->n_seqs < 1 ? 1 : desc->n_seqs;
    const gl3_model_desc& d = ctx->d;
    auto bail = [&](int32_t code, const std::string& msg) { g_create_err = msg.empty() ? ctx->err : msg; gl3_destroy(ctx); return code; };
    const double t0 = now_ms();
    if (d.arch != GL3_ARCH_LLAMA && d.arch != GL3_ARCH_QWEN3 && d.arch != GL3_ARCH_QWEN2 && d.arch != GL3_ARCH_GRANITE && d.arch != GL3_ARCH_PHI3 &&
        d.arch != GL3_ARCH_QWEN2MOE)
        return bail(GL3_E_UNSUPPORTED, "unsupported architecture");
    const bool moe = d.arch == GL3_ARCH_QWEN2MOE;
    if (moe) {      // the reference's GPU plan for this family is Q8_0, single token (Qwen2MoEQ8_0PlanComponents); so is this one
        if (d.weight_type != GL3_TYPE_Q8_0 || (d.flags & GL3_FLAG_F32_ACTIVATION))
            return bail(GL3_E_UNSUPPORTED, "qwen2moe: Q8_0 matrices with the int8 activation only");
        if (d.tp_size > 1 || (d.flags & GL3_FLAG_FORCE_RCCL)) return bail(GL3_E_UNSUPPORTED, "qwen2moe: tensor parallelism is not built");
        // max_batch > 1 is accepted: a prefill chunk then runs token by token, as the reference prefills this family (and as the
        // scalar-dot plans do); there are no batched buffers, so static-batched decode is refused
        if (d.n_seqs > 1) return bail(GL3_E_UNSUPPORTED, "qwen2moe: static-batched decode is not built (n_seqs <= 1)");
        if (d.n_experts < 1 || d.n_experts > 4096 || d.n_experts_used < 1 || d.n_experts_used > d.n_experts || d.n_experts_used > 64)
            return bail(GL3_E_ARG, "qwen2moe: need 1 <= n_experts_used <= min(n_experts, 64), n_experts <= 4096");
        if (d.moe_hidden < 32 || d.moe_hidden % 32) return bail(GL3_E_ARG, "qwen2moe: moe_hidden must be a positive multiple of 32");
        // the stacked expert tensors are addressed with int row counts (alloc_mat / upload_q8): n_experts * moe_hidden and
        // n_experts * dim must fit, and moe_hidden comes from an untrusted GGUF's ne[0] (r4 advisor finding)
        if (d.moe_hidden > (1 << 24) || (int64_t)d.n_experts * d.moe_hidden > INT32_MAX || (int64_t)d.n_experts * d.dim > INT32_MAX)
            return bail(GL3_E_ARG, "qwen2moe: n_experts * moe_hidden / n_experts * dim exceed the supported range");
        // the router keeps the products of MOE_RR rows in LDS (gl3_moe_kernels.h): dim <= ~4500 with 60 experts
        if (d.dim > 0 && moe_router_smem(d.dim, d.n_experts) > (size_t)160 * 1024 - 256)
            return bail(GL3_E_UNSUPPORTED, "qwen2moe: dim too large for the router kernel's LDS staging");
    } else if (d.n_experts || d.n_experts_used || d.moe_hidden)
        return bail(GL3_E_ARG, "n_experts / n_experts_used / moe_hidden belong to GL3_ARCH_QWEN2MOE");
    // Granite: the Llama graph (adjacent-pair RoPE) + four scalars; Phi-3: NeoX pairs like Qwen2, without biases
    ctx->rope_arch = d.arch == GL3_ARCH_GRANITE ? 0 : (d.arch == GL3_ARCH_PHI3 || moe) ? 2 : d.arch;
    if (d.arch == GL3_ARCH_GRANITE) {
        if (!(d.attention_scale > 0.f)) return bail(GL3_E_ARG, "granite: attention_scale must be > 0");
        ctx->emb_scale = d.embedding_scale; ctx->resid_scale = d.residual_scale; ctx->logit_scale = d.logit_scale; ctx->att_mul = d.attention_scale;
    }
    if (d.weight_type != GL3_TYPE_Q8_0 && d.weight_type != GL3_TYPE_F16 && d.weight_type != GL3_TYPE_Q4_0)
        return bail(GL3_E_UNSUPPORTED, "matrix weight type must be Q8_0, F16 or Q4_0");
    const bool q8v = d.weight_type == GL3_TYPE_Q8_0 && (d.flags & GL3_FLAG_F32_ACTIVATION);
    {   // the Vector-API species (-Dllama.VectorBitSize; include/gpullama3_hip.h): 256 by default, 512 for F16 decode, the rest refused
        const bool species_type = d.weight_type == GL3_TYPE_F16 || d.weight_type == GL3_TYPE_Q4_0 || q8v;
        if ((d.flags & GL3_FLAG_VECTOR_512) && (d.flags & GL3_FLAG_VECTOR_128)) return bail(GL3_E_ARG, "GL3_FLAG_VECTOR_512 and GL3_FLAG_VECTOR_128 exclude each other");
        if ((d.flags & (GL3_FLAG_VECTOR_512 | GL3_FLAG_VECTOR_128)) && (d.flags & GL3_FLAG_SCALAR_DOT))
            return bail(GL3_E_ARG, "a vector species and GL3_FLAG_SCALAR_DOT (-Dllama.VectorBitSize=0) exclude each other");
        if ((d.flags & GL3_FLAG_VECTOR_128) && species_type)
            return bail(GL3_E_UNSUPPORTED, "the 128-bit Vector-API species (4 accumulator lanes) is restated in the oracles only; run the JVM with -Dllama.VectorBitSize=256");
        if ((d.flags & GL3_FLAG_VECTOR_512) && species_type && d.weight_type != GL3_TYPE_F16)
            return bail(GL3_E_UNSUPPORTED, "Q4_0 / Q8_0 (f32 activation) have no 512-bit vector dot: the reference throws UnsupportedOperationException (Q4_0FloatTensor.java:118-120, "
                                           "Q8_0FloatTensor.java:165-167); run the JVM with -Dllama.VectorBitSize=256");
        if ((d.flags & GL3_FLAG_VECTOR_512) && d.weight_type == GL3_TYPE_F16 && d.tp_size > 1)
            return bail(GL3_E_UNSUPPORTED, "the 512-bit F16 species is built for one rank");
    }
    if ((d.flags & GL3_FLAG_F32_ACTIVATION) && d.weight_type != GL3_TYPE_Q8_0)
        return bail(GL3_E_ARG, "GL3_FLAG_F32_ACTIVATION applies to Q8_0 matrices");
    if (q8v && (d.flags & GL3_FLAG_SCALAR_DOT))
        return bail(GL3_E_UNSUPPORTED, "Q8_0 with f32 activation is built in the Vector-API order only");
    if (q8v && (d.dim % 128 || d.hidden % 128 || (d.n_heads * d.head_size) % 128))
        return bail(GL3_E_UNSUPPORTED, "Q8_0 with f32 activation needs inner dimensions that are multiples of 128");
    if (q8v && d.tp_size > 1 && (d.vocab / d.tp_size) % 8)
        return bail(GL3_E_UNSUPPORTED, "Q8_0 with f32 activation: vocab/tp_size must be a multiple of 8");
    if (d.weight_type != GL3_TYPE_Q8_0 && (d.dim % 64 || d.hidden % 64 || (d.n_heads * d.head_size) % 64))
        return bail(GL3_E_UNSUPPORTED, "F16 / Q4_0 matrices need inner dimensions that are multiples of 64");
    if (d.weight_type != GL3_TYPE_Q8_0 && rl_smem_bytes(d.hidden > d.dim ? d.hidden : d.dim, EPI_STORE) > 150 * 1024)
        return bail(GL3_E_UNSUPPORTED, "F16 / Q4_0: activation vector does not fit in LDS");
    {   // vector-order matvecs (the default for F16 / Q4_0, and Q8_0 with the f32 activation) stage the whole activation in LDS under the
        // 64 KB dynamic default: K <= 16384.  Reported here, not as a launch failure at the first decode step (r3 review).
        const bool vector_order = q8v || (d.weight_type != GL3_TYPE_Q8_0 && !(d.flags & GL3_FLAG_SCALAR_DOT));
        const int qd_ = d.n_heads * d.head_size, kmax = d.hidden > d.dim ? (d.hidden > qd_ ? d.hidden : qd_) : (d.dim > qd_ ? d.dim : qd_);
        if (vector_order && vl_smem_bytes(kmax) > 64 * 1024)
            return bail(GL3_E_UNSUPPORTED, "vector-order matvec: an inner dimension above 16384 does not fit its LDS staging (Q8_0 with the int8 activation has no such limit)");
    }
    if (d.weight_type == GL3_TYPE_Q4_0 && !(d.flags & GL3_FLAG_SCALAR_DOT) && (d.dim % 256 || d.hidden % 256 || (d.n_heads * d.head_size) % 256))
        return bail(GL3_E_UNSUPPORTED, "Q4_0 in Vector-API order needs inner dimensions that are multiples of 256 (or GL3_FLAG_SCALAR_DOT)");
    // a rank's vocabulary slice is whole weight groups: 8 rows in the vector order (128256 / 8 = 16032 = 2004 groups: BASELINE configs[3]), 64 in the scalar order
    if (d.weight_type != GL3_TYPE_Q8_0 && d.tp_size > 1 && (d.vocab / d.tp_size) % ((d.flags & GL3_FLAG_SCALAR_DOT) ? 64 : 8))
        return bail(GL3_E_UNSUPPORTED, "F16 / Q4_0 tensor parallel needs vocab/tp_size to be a multiple of 8 (64 with GL3_FLAG_SCALAR_DOT)");
    if (d.dim <= 0 || d.dim % 32 || d.hidden % 32 || d.n_layers <= 0 || d.n_heads <= 0 || d.n_kv_heads <= 0 ||
        d.n_heads % d.n_kv_heads || d.vocab <= 0 || d.ctx <= 0)
        return bail(GL3_E_ARG, "bad model dimensions");
    if (d.head_size < 32 || d.head_size > 256 || d.head_size % 32)       // 96: Phi-3-mini; 160 / 192 / 224 work the same way
        return bail(GL3_E_UNSUPPORTED, "head_size must be a multiple of 32 between 32 and 256");
    const int tp = d.tp_size < 1 ? 1 : d.tp_size;
    ctx->d.tp_size = tp;
    if (d.tp_rank < 0 || d.tp_rank >= tp) return bail(GL3_E_ARG, "tp_rank out of range");
    if (d.n_heads % tp || d.n_kv_heads % tp || d.hidden % (16 * tp) || d.vocab % (16 * tp) || d.dim % (16 * tp))
        return bail(GL3_E_UNSUPPORTED, "tp_size must divide n_heads, n_kv_heads, hidden/16, dim/16 and vocab/16");
    if (d.n_heads / d.n_kv_heads > 16) return bail(GL3_E_UNSUPPORTED, "more than 16 query heads per kv head");
    ctx->q_dim = d.n_heads * d.head_size; ctx->kv_dim = d.n_kv_heads * d.head_size;
    ctx->heads_l = d.n_heads / tp; ctx->kv_heads_l = d.n_kv_heads / tp;
    ctx->q_dim_l = ctx->heads_l * d.head_size; ctx->kv_dim_l = ctx->kv_heads_l * d.head_size;
    ctx->hidden_l = d.hidden / tp; ctx->vocab_l = d.vocab / tp; ctx->dim_l = d.dim / tp;
    ctx->n_tsplit = (d.ctx + ATT_TT - 1) / ATT_TT;
    // Tensor parallel: Wo is REPLICATED (every rank holds all dim rows) — each rank computes the whole attention output projection
    // from the gathered xb, so the residual stream is complete on every rank without a gather behind it: 3 gathers per layer
    // (xb, hb, x) instead of 4.  Streaming all of Wo (17.9 MB for the 8B model: ~6 us) costs less than a gather hop over xGMI, and
    // the dot products stay whole and in order, so the results stay bit-identical.  GL3_TP_SPLIT_WO=1 restores the row split.
    // (the batched prefill runs the replicated Wo per rank chunk of dim / tp rows and reads 32-row strip pairs: a chunk must start on an
    // even strip, i.e. dim % (32 tp) == 0 — otherwise the row split with its padded per-rank allocation is used)
    ctx->wo_replicated = tp > 1 && !env_flag("GL3_TP_SPLIT_WO", false) && (d.max_batch <= 1 || d.dim % (32 * tp) == 0);
    ctx->wo_rows = ctx->wo_replicated ? d.dim : ctx->dim_l;
    ctx->use_rccl = tp > 1 || (d.flags & GL3_FLAG_FORCE_RCCL);

#define TRY(x) do { int32_t r_ = (x); if (r_ != GL3_OK) return bail(r_, ""); } while (0)
#define TRYHIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return bail(e_ == hipErrorOutOfMemory ? GL3_E_OOM : GL3_E_HIP, std::string(#x) + ": " + hipGetErrorString(e_)); } while (0)
    TRYHIP(hipSetDevice(d.device));
    TRYHIP(hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
    TRY(alloc_mat(ctx, ctx->emb, d.vocab, d.dim));
    ctx->layers.resize(d.n_layers);
    for (auto& L : ctx->layers) {
        TRY(alloc_mat(ctx, L.wqkv, ctx->q_dim_l + 2 * ctx->kv_dim_l, d.dim));
        TRY(alloc_mat(ctx, L.wo, ctx->wo_rows, ctx->q_dim));
        TRY(alloc_mat(ctx, L.w1, ctx->hidden_l, d.dim));
        TRY(alloc_mat(ctx, L.w3, ctx->hidden_l, d.dim));
        TRY(alloc_mat(ctx, L.w2, ctx->dim_l, d.hidden));
        TRY(dmalloc(ctx, &L.attn_norm, d.dim));
        TRY(dmalloc(ctx, &L.ffn_norm, d.dim));
        if (d.arch == GL3_ARCH_QWEN3) { TRY(dmalloc(ctx, &L.qnorm, d.head_size)); TRY(dmalloc(ctx, &L.knorm, d.head_size)); }
        if (d.arch == GL3_ARCH_QWEN2 || moe) { TRY(dmalloc(ctx, &L.bq, ctx->q_dim_l)); TRY(dmalloc(ctx, &L.bk, ctx->kv_dim_l)); TRY(dmalloc(ctx, &L.bv, ctx->kv_dim_l)); }
        if (moe) {
            TRY(alloc_mat(ctx, L.gate_exps, d.n_experts * d.moe_hidden, d.dim));
            TRY(alloc_mat(ctx, L.up_exps, d.n_experts * d.moe_hidden, d.dim));
            TRY(alloc_mat(ctx, L.down_exps, d.n_experts * d.dim, d.moe_hidden));
            TRY(dmalloc(ctx, &L.gate_inp, (size_t)d.n_experts * d.dim));
            TRY(dmalloc(ctx, &L.gate_inp_shexp, d.dim));
        }
    }
    if (moe) {
        TRY(dmalloc(ctx, &ctx->moe_logits, d.n_experts));
        TRY(dmalloc(ctx, &ctx->moe_w, d.n_experts_used + 1));
        TRY(dmalloc(ctx, &ctx->moe_sel, d.n_experts_used + 1));          // + the router kernel's arrival ticket
        TRY(dmalloc(ctx, &ctx->moe_hb, (size_t)d.n_experts_used * d.moe_hidden));
        TRY(dmalloc(ctx, &ctx->moe_y, (size_t)(d.n_experts_used + 1) * d.dim));
        TRYHIP(hipMemset(ctx->moe_sel, 0, sizeof(int) * (d.n_experts_used + 1)));
        TRYHIP(hipFuncSetAttribute((const void*)matvec_q8t_kernel<PRO_RMS, EPI_SWIGLU, true, 4, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256));
        TRYHIP(hipFuncSetAttribute((const void*)matvec_q8t_kernel<PRO_QUANT, EPI_RESID, true, 4, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256));
        TRYHIP(hipFuncSetAttribute((const void*)moe_router_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256));
    }
    TRY(dmalloc(ctx, &ctx->out_norm, d.dim));
    ctx->kv_seq_stride = (size_t)d.n_layers * d.ctx * ctx->kv_dim_l;
    const size_t kvn = ctx->kv_seq_stride * ctx->n_seqs;
    TRY(dmalloc(ctx, &ctx->kcache, kvn));
    TRY(dmalloc(ctx, &ctx->vcache, kvn + (size_t)PVT * ctx->kv_dim_l));      // + PVT rows: attn_pv_kernel's last tile reads past position n - 1 unclamped (masked)
    TRYHIP(hipMemset(ctx->kcache, 0, kvn * 4));
    TRYHIP(hipMemset(ctx->vcache, 0, (kvn + (size_t)PVT * ctx->kv_dim_l) * 4));
    TRY(dmalloc(ctx, &ctx->xn, d.dim));
    TRY(dmalloc(ctx, &ctx->qkv, ctx->q_dim_l + 2 * ctx->kv_dim_l));
    if (ctx->use_rccl) {          // gathered buffers live in the tensor-parallel arena (one allocation the peers map, gl3_tp.hip)
        if (tp > GL3_MAX_TP) return bail(GL3_E_UNSUPPORTED, "tp_size above 16");
        TRY(gl3_tp_arena_alloc(ctx));
        ctx->x = (float*)(ctx->arena.base + ctx->arena.off[GB_X]); ctx->xb = (float*)(ctx->arena.base + ctx->arena.off[GB_XB]);
        ctx->hb = (float*)(ctx->arena.base + ctx->arena.off[GB_HB]); ctx->logits = (float*)(ctx->arena.base + ctx->arena.off[GB_LOGITS]);
    } else {
        TRY(dmalloc(ctx, &ctx->x, d.dim));
        TRY(dmalloc(ctx, &ctx->xb, ctx->q_dim));
        TRY(dmalloc(ctx, &ctx->hb, d.hidden));
        TRY(dmalloc(ctx, &ctx->logits, d.vocab));
    }
    TRY(dmalloc(ctx, &ctx->att, (size_t)ctx->heads_l * ((d.ctx + 3) & ~3)));      // score rows padded to float4 (AttnArgs.att_stride)
    TRY(dmalloc(ctx, &ctx->att_t, attn_att_t_floats(ctx->kv_heads_l, d.n_heads / d.n_kv_heads, (d.ctx + 3) & ~3)));      // normalised weights, attn_pv_kernel's order (+ PVT rows: see vcache)
    TRYHIP(hipMemset(ctx->att_t, 0, attn_att_t_floats(ctx->kv_heads_l, d.n_heads / d.n_kv_heads, (d.ctx + 3) & ~3) * 4));
    TRY(dmalloc(ctx, &ctx->att_tmax, (size_t)ctx->heads_l * ctx->n_tsplit));
    TRY(dmalloc(ctx, &ctx->att_sums, (size_t)ctx->heads_l));
    if (moe) TRY(moe_build_slots(ctx));          // after hb: the merged gate/up launch writes the shared expert's SwiGLU output there
    TRYHIP((allow_big_lds<PRO_RMS, EPI_STORE>()));
    TRYHIP((allow_big_lds<PRO_QUANT, EPI_RESID>()));
    TRYHIP((allow_big_lds<PRO_RMS, EPI_SWIGLU>()));
    {
        // one launch per layer for positions < AF_MAXN (attn_head_kernel); head_size 256 does not fit its K / V tiles in LDS
        ctx->fused_attn_ok = d.head_size % 4 == 0 && attn_head_smem(d.head_size) <= 150 * 1024 && !env_flag("GL3_NO_FUSED_ATTN", false);
        if (ctx->fused_attn_ok) {
            hipError_t ae = hipSuccess;
            attn_head_dispatch(d.head_size, [&](auto kern) { ae = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256); });
            TRYHIP(ae);
        }
    }
    TRYHIP(hipFuncSetAttribute((const void*)matvec_q8t_kernel<PRO_QUANT, EPI_RESID, true, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256));
#define GL3_RL_LDS(...) TRYHIP(hipFuncSetAttribute((const void*)matvec_rl_kernel<__VA_ARGS__>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256))
    GL3_RL_LDS(WT_F16, EPI_STORE); GL3_RL_LDS(WT_F16, EPI_RESID); GL3_RL_LDS(WT_F16, EPI_SWIGLU);
    GL3_RL_LDS(WT_Q4_0, EPI_STORE); GL3_RL_LDS(WT_Q4_0, EPI_RESID); GL3_RL_LDS(WT_Q4_0, EPI_SWIGLU);
#undef GL3_RL_LDS
    TRYHIP(hipFuncSetAttribute((const void*)rmsnorm_f32_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256));
    TRYHIP(hipFuncSetAttribute((const void*)attn_scores_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
    TRYHIP(hipFuncSetAttribute((const void*)attn_scores_loop_kernel<128>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
    TRYHIP(hipFuncSetAttribute((const void*)attn_scores_loop_kernel<64>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
    TRYHIP(hipFuncSetAttribute((const void*)attn_softmax_pv_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
    TRYHIP(hipFuncSetAttribute((const void*)attn_mid_kernel<128>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)attn_mid_smem<128>()));
    TRYHIP(hipFuncSetAttribute((const void*)attn_mid_kernel<64>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)attn_mid_smem<64>()));
    TRYHIP(hipFuncSetAttribute((const void*)attn_sum_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)attn_sum_smem()));
    TRYHIP(hipFuncSetAttribute((const void*)attn_pv_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)attn_pv_smem()));
    // softmax rows longer than the LDS window (16384 positions; GL3_ATTN_WINDOW, a multiple of 1024, shrinks it for tests) run in
    // windows with the sequential sum carried across them: no cap on the context length (r2 rejected contexts above ~20 k)
    {
        const char* wv = getenv("GL3_ATTN_WINDOW");
        int wmax = wv && *wv ? atoi(wv) : 16384;
        if (wmax < PV_ROWS || wmax % PV_ROWS != 0 || wmax > 16384) return bail(GL3_E_ARG, "GL3_ATTN_WINDOW must be a multiple of 1024 between 1024 and 16384");
        ctx->attn_win = d.ctx <= wmax ? ((d.ctx + 3) & ~3) : wmax;
        // depth below which the two-launch attention is used (GL3_ATTN_MID: 0 = always the four-launch long-context path)
        const char* mv = getenv("GL3_ATTN_MID");
        ctx->attn_mid = mv && *mv ? atoi(mv) : 768;
    }
    TRY(dmalloc(ctx, &ctx->dyn, 4));
    ctx->dyn_cur = ctx->dyn;
    TRY(dmalloc(ctx, &ctx->argmax, 2 + 2 * AMX_WGS));          // result, ticket, (value, index) pairs of argmax_kernel
    TRYHIP(hipMemset(ctx->argmax, 0, (2 + 2 * AMX_WGS) * sizeof(int)));
    if (d.flags & GL3_FLAG_LAYER_TAPS) TRY(dmalloc(ctx, &ctx->taps, (size_t)d.n_layers * d.dim));
    TRYHIP(hipHostMalloc((void**)&ctx->h_dyn, 4 * sizeof(int)));
    TRYHIP(hipHostMalloc((void**)&ctx->h_logits, (size_t)d.vocab * 4));
    TRYHIP(hipHostMalloc((void**)&ctx->h_argmax, sizeof(int)));
    // batched prefill / static-batched decode: int8 MFMA GEMMs for Q8_0 (any tensor-parallel degree); f32-MFMA / VALU GEMMs in the
    // Vector-API order for F16, Q4_0 and Q8_0 with the f32 activation (gl3_prefill_vl.h).  The scalar dot order
    // (GL3_FLAG_SCALAR_DOT) prefills token by token.
    {
        const bool int8_path = d.weight_type == GL3_TYPE_Q8_0 && !(d.flags & GL3_FLAG_F32_ACTIVATION);
        // (the 512-bit F16 species has decode kernels only: its prefill chunks run token by token, like the scalar order)
        const bool vl_path = !int8_path && !(d.flags & GL3_FLAG_SCALAR_DOT) && !(d.flags & GL3_FLAG_VECTOR_512);      // r4: tensor-parallel ranks too (rank-chunked activations)
        if (d.max_batch > 1 && (int8_path || vl_path) && !moe) TRY(gl3_prefill_alloc(ctx));
    }
    if (getenv("GL3_DEBUG_ALLOC")) {
        fprintf(stderr, "[gl3 alloc] ctx %p emb %p (+%zu) kcache %p vcache %p (%zu floats) x %p qkv %p xb %p hb %p logits %p att %p xn %p\n", (void*)ctx, (void*)ctx->emb.w,
                ctx->emb.bytes(), (void*)ctx->kcache, (void*)ctx->vcache, kvn, (void*)ctx->x, (void*)ctx->qkv, (void*)ctx->xb, (void*)ctx->hb, (void*)ctx->logits, (void*)ctx->att, (void*)ctx->xn);
        for (size_t l = 0; l < ctx->layers.size(); ++l) {
            const gl3_layer& L = ctx->layers[l];
            fprintf(stderr, "[gl3 alloc]   layer %zu wqkv %p (+%zu) wo %p (+%zu) w1 %p (+%zu) w3 %p w2 %p (+%zu)\n", l, (void*)L.wqkv.w, L.wqkv.bytes(), (void*)L.wo.w, L.wo.bytes(),
                    (void*)L.w1.w, L.w1.bytes(), (void*)L.w3.w, (void*)L.w2.w, L.w2.bytes());
        }
    }
    TRYHIP(hipDeviceSynchronize());
#undef TRY
#undef TRYHIP
    ctx->plan_ms = now_ms() - t0;
    *out = ctx;
    return GL3_OK;
}

void gl3_destroy(gl3_ctx* ctx) {
    if (!ctx) return;
    hipSetDevice(ctx->d.device);
    if (ctx->stream) hipStreamSynchronize(ctx->stream);
    if (ctx->graph_exec) hipGraphExecDestroy(ctx->graph_exec);
    if (ctx->graph_exec_s) hipGraphExecDestroy(ctx->graph_exec_s);
    if (ctx->graph_exec_m) hipGraphExecDestroy(ctx->graph_exec_m);
    if (ctx->graph_m) hipGraphDestroy(ctx->graph_m);
    if (ctx->graph_s) hipGraphDestroy(ctx->graph_s);
    if (ctx->graph) hipGraphDestroy(ctx->graph);
    if (ctx->comm) ncclCommDestroy(ctx->comm);
    gl3_prefill_free(ctx);
    gl3_sample_free(ctx);
    for (auto e : ctx->ev) hipEventDestroy(e);
    auto f = [](void* p) { if (p) hipFree(p); };
    f(ctx->emb.w);
    if (ctx->wcls_owned) f(ctx->wcls.w);
    for (auto& L : ctx->layers) {
        f(L.wqkv.w); f(L.wo.w); f(L.w1.w); f(L.w3.w); f(L.w2.w);
        f(L.attn_norm); f(L.ffn_norm); f(L.qnorm); f(L.knorm); f(L.bq); f(L.bk); f(L.bv);
        f(L.gate_exps.w); f(L.up_exps.w); f(L.down_exps.w); f(L.gate_inp); f(L.gate_inp_shexp);
    }
    f(ctx->moe_logits); f(ctx->moe_w); f(ctx->moe_sel); f(ctx->moe_hb); f(ctx->moe_y); f(ctx->moe_slots); f(ctx->tp_recs);
    f(ctx->out_norm); f(ctx->rope_cr); f(ctx->rope_ci); f(ctx->kcache); f(ctx->vcache); f(ctx->xn); f(ctx->qkv);
    if (!ctx->arena.base) { f(ctx->x); f(ctx->xb); f(ctx->hb); f(ctx->logits); }
    gl3_tp_arena_free(ctx);
    f(ctx->att); f(ctx->att_t); f(ctx->att_tmax); f(ctx->att_sums); f(ctx->dyn); f(ctx->dyn_seq); f(ctx->argmax); f(ctx->taps); f(ctx->staging);
    for (auto& pr : ctx->pinned) hipHostUnregister(pr.first);
    ctx->pinned.clear();
    if (ctx->h_dyn) hipHostFree(ctx->h_dyn);
    if (ctx->h_logits) hipHostFree(ctx->h_logits);
    if (ctx->h_argmax) hipHostFree(ctx->h_argmax);
    if (ctx->stream) hipStreamDestroy(ctx->stream);
    delete ctx;
}

// ------------------------------------------------------------------------------------------------ upload
static int32_t stage(gl3_ctx* ctx, const void* host, size_t bytes) {
    if (ctx->staging_bytes < bytes) {
        if (ctx->staging) hipFree(ctx->staging);
        ctx->staging = nullptr; ctx->staging_bytes = 0;
        GL3_HIP(hipMalloc((void**)&ctx->staging, bytes));
        ctx->staging_bytes = bytes;
    }
    GL3_HIP(hipMemcpy(ctx->staging, host, bytes, hipMemcpyHostToDevice));
    return GL3_OK;
}

// src: full [rows_full x k_full] GGUF tensor on the host (Q8_0 / F16 / Q4_0 row-major blocks); keeps rows
// [r0, r0+sub_rows), placed at row dst_row0 of m.  Only the kept rows are staged (tensor-parallel ranks upload 1/tp of
// the bytes).
static int32_t upload_q8(gl3_ctx* ctx, Q8Mat& m, int dst_row0, int sub_rows, const void* host, uint64_t bytes, int rows_full,
                         int k_full, long r0) {
    const int nb_full = k_full / 32;
    if (k_full != m.k) GL3_FAIL(GL3_E_ARG, "tensor inner dimension mismatch");
    const size_t row_bytes = (m.fmt == GL3_TYPE_Q8_0 || m.fmt == GL3_FMT_Q8V) ? (size_t)nb_full * 34 : m.fmt == GL3_TYPE_F16 ? (size_t)k_full * 2 : (size_t)nb_full * 18;
    if (bytes != (uint64_t)rows_full * row_bytes) GL3_FAIL(GL3_E_ARG, "tensor byte size does not match its shape");
    const uint8_t* h = (const uint8_t*)host + (size_t)r0 * row_bytes;
    if (m.fmt != GL3_TYPE_Q8_0) {
        int32_t r = stage(ctx, h, (size_t)sub_rows * row_bytes);
        if (r != GL3_OK) return r;
        if (m.vl) {
            const long total = (long)sub_rows * (m.fmt == GL3_TYPE_F16 ? k_full / 64 : m.fmt == GL3_FMT_Q8V ? k_full / 128 : k_full / 256) * 8;
            const dim3 grid((unsigned)((total + 255) / 256));
            if (m.fmt == GL3_TYPE_F16) hipLaunchKernelGGL((repack_vl_kernel<WT_F16>), grid, dim3(256), 0, ctx->stream, ctx->staging, m.w, sub_rows, k_full, dst_row0);
            else if (m.fmt == GL3_FMT_Q8V) hipLaunchKernelGGL((repack_vl_kernel<WT_Q8_0>), grid, dim3(256), 0, ctx->stream, ctx->staging, m.w, sub_rows, k_full, dst_row0);
            else hipLaunchKernelGGL((repack_vl_kernel<WT_Q4_0>), grid, dim3(256), 0, ctx->stream, ctx->staging, m.w, sub_rows, k_full, dst_row0);
            GL3_HIP(hipStreamSynchronize(ctx->stream));
            return GL3_OK;
        }
        const long total = (long)sub_rows * (m.fmt == GL3_TYPE_F16 ? k_full / 8 : nb_full);
        const dim3 grid((unsigned)((total + 255) / 256));
        if (m.fmt == GL3_TYPE_F16) hipLaunchKernelGGL((repack_rl_kernel<WT_F16>), grid, dim3(256), 0, ctx->stream, ctx->staging, m.w, sub_rows, k_full, dst_row0);
        else hipLaunchKernelGGL((repack_rl_kernel<WT_Q4_0>), grid, dim3(256), 0, ctx->stream, ctx->staging, m.w, sub_rows, k_full, dst_row0);
        GL3_HIP(hipStreamSynchronize(ctx->stream));
        return GL3_OK;
    }
    if (dst_row0 % 16) GL3_FAIL(GL3_E_UNSUPPORTED, "sub-matrix row offset must be a multiple of 16");
    int32_t r = stage(ctx, h, (size_t)sub_rows * nb_full * 34);
    if (r != GL3_OK) return r;
    const long total = (long)((sub_rows + 15) & ~15) * m.ng * 4;
    hipLaunchKernelGGL(repack_q8t_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ctx->stream, ctx->staging,
                       m.w + (size_t)(dst_row0 / 16) * m.ng * TILE_BYTES, sub_rows, nb_full, m.ng, 0L, 0, nb_full);
    GL3_HIP(hipStreamSynchronize(ctx->stream));
    return GL3_OK;
}

static int32_t upload_f32(gl3_ctx* ctx, float* dst, int n, const void* host, uint64_t bytes, int type) {
    if (type != GL3_TYPE_F32) GL3_FAIL(GL3_E_UNSUPPORTED, "norm weights must be F32");
    if (!dst) GL3_FAIL(GL3_E_ARG, "tensor not part of this architecture");
    if (bytes != (uint64_t)n * 4) GL3_FAIL(GL3_E_ARG, "norm tensor byte size mismatch");
    GL3_HIP(hipMemcpy(dst, host, bytes, hipMemcpyHostToDevice));
    return GL3_OK;
}

// f32 vector split like the rows of its matrix: rank keeps elements [rank * n_local, (rank + 1) * n_local)
static int32_t upload_f32_slice(gl3_ctx* ctx, float* dst, int n_full, int n_local, int rank, const void* host, uint64_t bytes, int type) {
    if (type != GL3_TYPE_F32) GL3_FAIL(GL3_E_UNSUPPORTED, "bias vectors must be F32");
    if (!dst) GL3_FAIL(GL3_E_ARG, "tensor not part of this architecture");
    if (bytes != (uint64_t)n_full * 4) GL3_FAIL(GL3_E_ARG, "bias tensor byte size mismatch");
    GL3_HIP(hipMemcpy(dst, (const float*)host + (size_t)rank * n_local, (size_t)n_local * 4, hipMemcpyHostToDevice));
    return GL3_OK;
}

int32_t gl3_upload_tensor(gl3_ctx* ctx, int32_t id, int32_t layer, const void* host, uint64_t bytes, int32_t type) {
    if (!ctx) return GL3_E_ARG;
    if (!host || id < 0 || id >= GL3_T_COUNT) GL3_FAIL(GL3_E_ARG, "bad tensor id / null host pointer");
    if (ctx->finalized) GL3_FAIL(GL3_E_STATE, "upload after finalize");
    GL3_HIP(hipSetDevice(ctx->d.device));
    const gl3_model_desc& d = ctx->d;
    const double t0 = now_ms();
    int32_t r = GL3_OK;
    const int rank = d.tp_rank;
    if ((id == GL3_T_WQKV || id == GL3_T_W13) && d.arch != GL3_ARCH_PHI3) GL3_FAIL(GL3_E_ARG, "fused attn_qkv / gate|up tensors belong to GL3_ARCH_PHI3");
    const bool is_mat = !(id == GL3_T_OUTPUT_NORM || id == GL3_T_ATTN_NORM || id == GL3_T_FFN_NORM || id == GL3_T_ATTN_Q_NORM ||
                          id == GL3_T_ATTN_K_NORM || id == GL3_T_BQ || id == GL3_T_BK || id == GL3_T_BV || id == GL3_T_FFN_GATE_INP ||
                          id == GL3_T_FFN_GATE_INP_SHEXP);
    if (id >= GL3_T_FFN_GATE_INP && d.arch != GL3_ARCH_QWEN2MOE) GL3_FAIL(GL3_E_ARG, "router / expert tensors belong to GL3_ARCH_QWEN2MOE");
    if (is_mat && type != d.weight_type) GL3_FAIL(GL3_E_UNSUPPORTED, "matrix ggml type differs from gl3_model_desc.weight_type");
    if (id > GL3_T_OUTPUT && (layer < 0 || layer >= d.n_layers)) GL3_FAIL(GL3_E_ARG, "layer out of range");
    gl3_layer* L = id > GL3_T_OUTPUT ? &ctx->layers[layer] : nullptr;
    switch (id) {
    case GL3_T_TOKEN_EMBD: r = upload_q8(ctx, ctx->emb, 0, d.vocab, host, bytes, d.vocab, d.dim, 0); break;
    case GL3_T_OUTPUT:
        if (!ctx->wcls_owned) { r = alloc_mat(ctx, ctx->wcls, ctx->vocab_l, d.dim); ctx->wcls_owned = (r == GL3_OK); }
        if (r == GL3_OK) r = upload_q8(ctx, ctx->wcls, 0, ctx->vocab_l, host, bytes, d.vocab, d.dim, (long)rank * ctx->vocab_l);
        break;
    case GL3_T_OUTPUT_NORM: r = upload_f32(ctx, ctx->out_norm, d.dim, host, bytes, type); break;
    case GL3_T_ATTN_NORM: r = upload_f32(ctx, L->attn_norm, d.dim, host, bytes, type); break;
    case GL3_T_FFN_NORM: r = upload_f32(ctx, L->ffn_norm, d.dim, host, bytes, type); break;
    case GL3_T_ATTN_Q_NORM: r = upload_f32(ctx, L->qnorm, d.head_size, host, bytes, type); break;
    case GL3_T_ATTN_K_NORM: r = upload_f32(ctx, L->knorm, d.head_size, host, bytes, type); break;
    case GL3_T_BQ: r = upload_f32_slice(ctx, L->bq, ctx->q_dim, ctx->q_dim_l, rank, host, bytes, type); break;
    case GL3_T_BK: r = upload_f32_slice(ctx, L->bk, ctx->kv_dim, ctx->kv_dim_l, rank, host, bytes, type); break;
    case GL3_T_BV: r = upload_f32_slice(ctx, L->bv, ctx->kv_dim, ctx->kv_dim_l, rank, host, bytes, type); break;
    case GL3_T_WQ: r = upload_q8(ctx, L->wqkv, 0, ctx->q_dim_l, host, bytes, ctx->q_dim, d.dim, (long)rank * ctx->q_dim_l); break;
    case GL3_T_WK: r = upload_q8(ctx, L->wqkv, ctx->q_dim_l, ctx->kv_dim_l, host, bytes, ctx->kv_dim, d.dim, (long)rank * ctx->kv_dim_l); break;
    case GL3_T_WV: r = upload_q8(ctx, L->wqkv, ctx->q_dim_l + ctx->kv_dim_l, ctx->kv_dim_l, host, bytes, ctx->kv_dim, d.dim, (long)rank * ctx->kv_dim_l); break;
    case GL3_T_WO: r = upload_q8(ctx, L->wo, 0, ctx->wo_rows, host, bytes, d.dim, ctx->q_dim, ctx->wo_replicated ? 0L : (long)rank * ctx->dim_l); break;
    case GL3_T_W1: r = upload_q8(ctx, L->w1, 0, ctx->hidden_l, host, bytes, d.hidden, d.dim, (long)rank * ctx->hidden_l); break;
    case GL3_T_W3: r = upload_q8(ctx, L->w3, 0, ctx->hidden_l, host, bytes, d.hidden, d.dim, (long)rank * ctx->hidden_l); break;
    case GL3_T_W2: r = upload_q8(ctx, L->w2, 0, ctx->dim_l, host, bytes, d.dim, d.hidden, (long)rank * ctx->dim_l); break;
    case GL3_T_WQKV: {        // phi3: rows q | k | v of one tensor; each rank keeps its heads of all three parts
        const int full = ctx->q_dim + 2 * ctx->kv_dim;
        r = upload_q8(ctx, L->wqkv, 0, ctx->q_dim_l, host, bytes, full, d.dim, (long)rank * ctx->q_dim_l);
        if (r == GL3_OK) r = upload_q8(ctx, L->wqkv, ctx->q_dim_l, ctx->kv_dim_l, host, bytes, full, d.dim, (long)ctx->q_dim + (long)rank * ctx->kv_dim_l);
        if (r == GL3_OK) r = upload_q8(ctx, L->wqkv, ctx->q_dim_l + ctx->kv_dim_l, ctx->kv_dim_l, host, bytes, full, d.dim,
                                       (long)ctx->q_dim + ctx->kv_dim + (long)rank * ctx->kv_dim_l);
        if (r == GL3_OK) L->have |= (1u << GL3_T_WQ) | (1u << GL3_T_WK) | (1u << GL3_T_WV);
        break;
    }
    case GL3_T_W13:           // phi3: rows gate | up of one tensor
        r = upload_q8(ctx, L->w1, 0, ctx->hidden_l, host, bytes, 2 * d.hidden, d.dim, (long)rank * ctx->hidden_l);
        if (r == GL3_OK) r = upload_q8(ctx, L->w3, 0, ctx->hidden_l, host, bytes, 2 * d.hidden, d.dim, (long)d.hidden + (long)rank * ctx->hidden_l);
        if (r == GL3_OK) L->have |= (1u << GL3_T_W1) | (1u << GL3_T_W3);
        break;
    // qwen2moe: the stacked experts are plain row ranges of one matrix (expert e = rows [e * rows_per_expert, (e + 1) * ...))
    case GL3_T_FFN_GATE_EXPS: r = upload_q8(ctx, L->gate_exps, 0, d.n_experts * d.moe_hidden, host, bytes, d.n_experts * d.moe_hidden, d.dim, 0); break;
    case GL3_T_FFN_UP_EXPS: r = upload_q8(ctx, L->up_exps, 0, d.n_experts * d.moe_hidden, host, bytes, d.n_experts * d.moe_hidden, d.dim, 0); break;
    case GL3_T_FFN_DOWN_EXPS: r = upload_q8(ctx, L->down_exps, 0, d.n_experts * d.dim, host, bytes, d.n_experts * d.dim, d.moe_hidden, 0); break;
    case GL3_T_FFN_GATE_INP: r = upload_f32(ctx, L->gate_inp, d.n_experts * d.dim, host, bytes, type); break;
    case GL3_T_FFN_GATE_INP_SHEXP: r = upload_f32(ctx, L->gate_inp_shexp, d.dim, host, bytes, type); break;
    default: GL3_FAIL(GL3_E_ARG, "unknown tensor id");
    }
    if (r != GL3_OK) return r;
    if (L) L->have |= 1u << id; else ctx->have_global |= 1u << id;
    ctx->copy_in_ms += now_ms() - t0;
    return GL3_OK;
}

int32_t gl3_upload_rope(gl3_ctx* ctx, const float* cr, const float* ci, uint64_t n) {
    if (!ctx) return GL3_E_ARG;
    if (!cr || !ci) GL3_FAIL(GL3_E_ARG, "null rope table");
    const uint64_t need = (uint64_t)ctx->d.ctx * (ctx->d.head_size / 2);
    if (n < need) GL3_FAIL(GL3_E_ARG, "rope table shorter than ctx * head_size/2");
    GL3_HIP(hipSetDevice(ctx->d.device));
    if (ctx->rope_cr) { hipFree(ctx->rope_cr); hipFree(ctx->rope_ci); ctx->rope_cr = ctx->rope_ci = nullptr; }
    GL3_HIP(hipMalloc((void**)&ctx->rope_cr, need * 4));
    GL3_HIP(hipMalloc((void**)&ctx->rope_ci, need * 4));
    GL3_HIP(hipMemcpy(ctx->rope_cr, cr, need * 4, hipMemcpyHostToDevice));
    GL3_HIP(hipMemcpy(ctx->rope_ci, ci, need * 4, hipMemcpyHostToDevice));
    ctx->rope_n = need;
    return GL3_OK;
}

static int32_t capture(gl3_ctx* ctx, bool want_logits, int amode, hipGraph_t* g, hipGraphExec_t* ge) {
    ctx->tp_dbg_prev_n4 = 0;
    GL3_HIP(hipStreamBeginCapture(ctx->stream, hipStreamCaptureModeThreadLocal));
    int32_t r = enqueue_decode(ctx, want_logits, nullptr, amode);
    hipError_t e = hipStreamEndCapture(ctx->stream, g);
    if (r != GL3_OK) return r;
    GL3_HIP(e);
    GL3_HIP(hipGraphInstantiate(ge, *g, nullptr, nullptr, 0));
    return GL3_OK;
}

int32_t gl3_finalize(gl3_ctx* ctx) {
    if (!ctx) return GL3_E_ARG;
    if (ctx->finalized) return GL3_OK;
    GL3_HIP(hipSetDevice(ctx->d.device));
    const gl3_model_desc& d = ctx->d;
    const uint32_t need_g = (1u << GL3_T_TOKEN_EMBD) | (1u << GL3_T_OUTPUT_NORM);
    if ((ctx->have_global & need_g) != need_g) GL3_FAIL(GL3_E_STATE, "token_embd / output_norm not uploaded");
    uint32_t need_l = (1u << GL3_T_ATTN_NORM) | (1u << GL3_T_WQ) | (1u << GL3_T_WK) | (1u << GL3_T_WV) | (1u << GL3_T_WO) |
                      (1u << GL3_T_FFN_NORM) | (1u << GL3_T_W1) | (1u << GL3_T_W2) | (1u << GL3_T_W3);
    if (d.arch == GL3_ARCH_QWEN3) need_l |= (1u << GL3_T_ATTN_Q_NORM) | (1u << GL3_T_ATTN_K_NORM);
    if (d.arch == GL3_ARCH_QWEN2 || d.arch == GL3_ARCH_QWEN2MOE) need_l |= (1u << GL3_T_BQ) | (1u << GL3_T_BK) | (1u << GL3_T_BV);
    if (d.arch == GL3_ARCH_QWEN2MOE)
        need_l |= (1u << GL3_T_FFN_GATE_INP) | (1u << GL3_T_FFN_GATE_EXPS) | (1u << GL3_T_FFN_UP_EXPS) | (1u << GL3_T_FFN_DOWN_EXPS) | (1u << GL3_T_FFN_GATE_INP_SHEXP);
    for (int l = 0; l < d.n_layers; ++l)
        if ((ctx->layers[l].have & need_l) != need_l) GL3_FAIL(GL3_E_STATE, "layer " + std::to_string(l) + ": tensors missing");
    if (!ctx->rope_cr) GL3_FAIL(GL3_E_STATE, "rope tables not uploaded");
    if (ctx->use_rccl && ctx->transport == GL3_TP_NONE)
        GL3_FAIL(GL3_E_STATE, "tensor parallel plan without gl3_tp_p2p_attach / gl3_tp_init / gl3_tp_attach_local");
    { int32_t r = gl3_tp_local_resolve(ctx); if (r != GL3_OK) return r; }
    { int32_t r = tp_fold_setup(ctx); if (r != GL3_OK) return r; }
    if (!(ctx->have_global & (1u << GL3_T_OUTPUT))) {   // tied: wcls = this rank's vocab rows of token_embd
        ctx->wcls = ctx->emb;
        ctx->wcls.rows = ctx->vocab_l;
        ctx->wcls.nstrips = (ctx->vocab_l + 15) / 16;
        if (ctx->emb.fmt == GL3_TYPE_Q8_0) ctx->wcls.w = ctx->emb.w + (size_t)(d.tp_rank * ctx->vocab_l / 16) * ctx->emb.ng * TILE_BYTES;
        else if (ctx->emb.vl) ctx->wcls.w = ctx->emb.w + (size_t)(d.tp_rank * ctx->vocab_l / 8) * ctx->emb.vl_group_bytes();
        else ctx->wcls.w = ctx->emb.w + (size_t)(d.tp_rank * ctx->vocab_l / 64) * ctx->emb.rl_group_bytes();
        ctx->wcls_owned = false;
    }
    if (ctx->staging) { hipFree(ctx->staging); ctx->staging = nullptr; ctx->staging_bytes = 0; }
    ctx->finalized = true;
    if (!(d.flags & GL3_FLAG_NO_GRAPH) && !env_flag("GL3_NO_GRAPH", false) && !gl3_roctx_on()) {
        const double t0 = now_ms();
        int32_t r = capture(ctx, true, ATT_LONG, &ctx->graph, &ctx->graph_exec);
        if (r == GL3_OK && ctx->fused_attn_ok) r = capture(ctx, true, ATT_SHORT, &ctx->graph_s, &ctx->graph_exec_s);
        if (r == GL3_OK && ctx->attn_mid > (ctx->fused_attn_ok ? AF_MAXN : 0)) r = capture(ctx, true, ATT_MID, &ctx->graph_m, &ctx->graph_exec_m);
        if (r != GL3_OK) {   // e.g. a collective that cannot be captured: run eagerly instead
            ctx->graph_exec = nullptr; ctx->graph_exec_s = nullptr; ctx->graph_exec_m = nullptr;
            (void)hipGetLastError();
            fprintf(stderr, "[gl3] hipGraph capture failed (%s); falling back to eager launches\n", ctx->err.c_str());
        }
        ctx->plan_ms += now_ms() - t0;
    }
    GL3_HIP(hipStreamSynchronize(ctx->stream));
    return GL3_OK;
}

// ------------------------------------------------------------------------------------------------ forward
static int32_t set_dyn(gl3_ctx* ctx, int32_t token, int32_t pos) {
    if (!ctx->finalized) GL3_FAIL(GL3_E_STATE, "forward before gl3_finalize");
    if (token < 0 || token >= ctx->d.vocab) GL3_FAIL(GL3_E_ARG, "token id out of range");
    if (pos < 0 || pos >= ctx->d.ctx) GL3_FAIL(GL3_E_ARG, "position outside the KV cache (context length)");
    GL3_HIP(hipSetDevice(ctx->d.device));
    ctx->h_dyn[0] = token; ctx->h_dyn[1] = pos;
    GL3_HIP(hipMemcpyAsync(ctx->dyn, ctx->h_dyn, 2 * sizeof(int), hipMemcpyHostToDevice, ctx->stream));
    return GL3_OK;
}

static hipGraphExec_t step_graph(gl3_ctx* ctx, int amode) {
    if (amode == ATT_SHORT && ctx->graph_exec_s) return ctx->graph_exec_s;
    if (amode == ATT_MID && ctx->graph_exec_m) return ctx->graph_exec_m;
    return ctx->graph_exec;          // the ATT_LONG step is correct at every position
}

int32_t gl3_forward_decode(gl3_ctx* ctx, int32_t token, int32_t pos, float* logits_out, int32_t* argmax_out) {
    if (!ctx) return GL3_E_ARG;
    int32_t r = set_dyn(ctx, token, pos);
    if (r != GL3_OK) return r;
    const bool want_logits = logits_out || argmax_out;
    const int amode = attn_mode(ctx, pos);
    if (ctx->graph_exec && want_logits) GL3_HIP(hipGraphLaunch(step_graph(ctx, amode), ctx->stream));
    else if ((r = enqueue_decode(ctx, want_logits, nullptr, amode)) != GL3_OK) return r;
    if (argmax_out) {
        hipLaunchKernelGGL(argmax_kernel, dim3(AMX_WGS), dim3(256), 0, ctx->stream, ctx->logits, ctx->d.vocab, ctx->argmax);
        GL3_HIP(hipMemcpyAsync(ctx->h_argmax, ctx->argmax, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    }
    // logits D2H: straight into the caller's buffer when it is page-locked (gl3_pin_host_buffer), else through the plan's pinned
    // staging buffer + one host memcpy (513 KB for a 128 k vocabulary: ~40 us per token on the host)
    bool direct = false;
    if (logits_out)
        for (const auto& pr : ctx->pinned)
            if ((uint8_t*)logits_out >= (uint8_t*)pr.first && (uint8_t*)logits_out + (size_t)ctx->d.vocab * 4 <= (uint8_t*)pr.first + pr.second) direct = true;
    if (logits_out) GL3_HIP(hipMemcpyAsync(direct ? logits_out : ctx->h_logits, ctx->logits, (size_t)ctx->d.vocab * 4, hipMemcpyDeviceToHost, ctx->stream));
    GL3_HIP(hipStreamSynchronize(ctx->stream));
    if ((r = gl3_tp_check(ctx)) != GL3_OK) return r;
    if (logits_out && !direct) memcpy(logits_out, ctx->h_logits, (size_t)ctx->d.vocab * 4);
    if (argmax_out) *argmax_out = *ctx->h_argmax;
    return GL3_OK;
}

int32_t gl3_forward_decode_sample(gl3_ctx* ctx, int32_t token, int32_t pos, float temperature, float topp, float coin, int32_t* token_out) {
    if (!ctx) return GL3_E_ARG;
    if (!token_out) GL3_FAIL(GL3_E_ARG, "null token_out");
    if (!(temperature >= 0.f)) GL3_FAIL(GL3_E_ARG, "temperature must be >= 0");
    if (temperature == 0.f) return gl3_forward_decode(ctx, token, pos, nullptr, token_out);      // Sampler.java:79-81: greedy argmax
    if (!(coin >= 0.f && coin < 1.f)) GL3_FAIL(GL3_E_ARG, "coin must be rng.nextFloat(1f): in [0, 1)");
    int32_t r = set_dyn(ctx, token, pos);
    if (r != GL3_OK) return r;
    const int amode = attn_mode(ctx, pos);
    if (ctx->graph_exec) GL3_HIP(hipGraphLaunch(step_graph(ctx, amode), ctx->stream));
    else if ((r = enqueue_decode(ctx, true, nullptr, amode)) != GL3_OK) return r;
    if ((r = gl3_sample_run(ctx, ctx->logits, temperature, topp, coin, token_out)) != GL3_OK) return r;
    return gl3_tp_check(ctx);
}

int32_t gl3_tp_fold_mode(gl3_ctx* ctx, int32_t* mode, int32_t* consumer_mask) {
    if (!ctx || !mode || !consumer_mask) return GL3_E_ARG;
    if (!ctx->finalized) GL3_FAIL(GL3_E_STATE, "gl3_tp_fold_mode before gl3_finalize");
    *mode = ctx->tp_fold; *consumer_mask = ctx->tp_fold ? ctx->tp_fold_mask : 0;
    return GL3_OK;
}

int32_t gl3_get_topp_counts(gl3_ctx* ctx, int64_t* on_device, int64_t* on_host) {
    if (!ctx || !on_device || !on_host) return GL3_E_ARG;
    *on_device = ctx->topp_device; *on_host = ctx->topp_host;
    return GL3_OK;
}

int32_t gl3_get_sample_probs(gl3_ctx* ctx, float* out) {
    if (!ctx || !out) return GL3_E_ARG;
    GL3_HIP(hipSetDevice(ctx->d.device));
    return gl3_sample_probs(ctx, out);
}

int32_t gl3_forward_prefill_seq(gl3_ctx* ctx, int32_t seq, const int32_t* tokens, int32_t n, int32_t start_pos) {
    if (!ctx) return GL3_E_ARG;
    if (!tokens || n < 0) GL3_FAIL(GL3_E_ARG, "bad token array");
    if (!ctx->finalized) GL3_FAIL(GL3_E_STATE, "forward before gl3_finalize");
    if (seq < 0 || seq >= ctx->n_seqs) GL3_FAIL(GL3_E_ARG, "sequence id out of range");
    if (start_pos < 0 || start_pos + n > ctx->d.ctx) GL3_FAIL(GL3_E_ARG, "prefill range outside the KV cache");
    if (n == 0) return GL3_OK;
    if (ctx->pf) {
        if (n > ctx->d.max_batch) GL3_FAIL(GL3_E_ARG, "prefill chunk larger than max_batch");
        return gl3_prefill_run(ctx, seq, tokens, n, start_pos);
    }
    if (seq != 0) GL3_FAIL(GL3_E_UNSUPPORTED, "sequences other than 0 need max_batch > 1");
    // max_batch <= 1: sequential single-token prefill without logits
    // (TornadoVMMasterPlanPrefillDecode.tornadoVMForwardPrefill, J/tornadovm/TornadoVMMasterPlanPrefillDecode.java:116).
    // All (token, position) pairs go to the device once; every token's launches read their own pair, so the host
    // enqueues the whole chunk without waiting for the stream.
    for (int i = 0; i < n; ++i)
        if (tokens[i] < 0 || tokens[i] >= ctx->d.vocab) GL3_FAIL(GL3_E_ARG, "token id out of range");
    GL3_HIP(hipSetDevice(ctx->d.device));
    if (ctx->dyn_seq_cap < n) {
        if (ctx->dyn_seq) hipFree(ctx->dyn_seq);
        ctx->dyn_seq = nullptr; ctx->dyn_seq_cap = 0;
        GL3_HIP(hipMalloc((void**)&ctx->dyn_seq, (size_t)2 * n * sizeof(int)));
        ctx->dyn_seq_cap = n;
    }
    std::vector<int> pairs((size_t)2 * n);
    for (int i = 0; i < n; ++i) { pairs[2 * i] = tokens[i]; pairs[2 * i + 1] = start_pos + i; }
    GL3_HIP(hipStreamSynchronize(ctx->stream));          // a previous chunk may still read dyn_seq
    GL3_HIP(hipMemcpy(ctx->dyn_seq, pairs.data(), pairs.size() * sizeof(int), hipMemcpyHostToDevice));
    int32_t r = GL3_OK;
    for (int i = 0; i < n && r == GL3_OK; ++i) {
        ctx->dyn_cur = ctx->dyn_seq + 2 * i;
        r = enqueue_decode(ctx, false, nullptr, attn_mode(ctx, start_pos + i));
    }
    ctx->dyn_cur = ctx->dyn;
    if (r != GL3_OK) return r;
    GL3_HIP(hipStreamSynchronize(ctx->stream));
    return gl3_tp_check(ctx);
}

int32_t gl3_forward_prefill(gl3_ctx* ctx, const int32_t* tokens, int32_t n, int32_t start_pos) {
    return gl3_forward_prefill_seq(ctx, 0, tokens, n, start_pos);
}

int32_t gl3_forward_decode_batch(gl3_ctx* ctx, const int32_t* tokens, const int32_t* seq_ids, const int32_t* positions, int32_t n,
                                 float* logits_out, int32_t* argmax_out) {
    if (!ctx) return GL3_E_ARG;
    if (!tokens || !seq_ids || !positions || n <= 0) GL3_FAIL(GL3_E_ARG, "bad batch arrays");
    if (!ctx->finalized) GL3_FAIL(GL3_E_STATE, "forward before gl3_finalize");
    if (!ctx->pf) GL3_FAIL(GL3_E_UNSUPPORTED, "batched decode needs max_batch > 1 (and, for F16 / Q4_0 / f32-activation Q8_0, one rank in the Vector-API order)");
    if (n > ctx->d.max_batch) GL3_FAIL(GL3_E_ARG, "batch larger than max_batch");
    for (int i = 0; i < n; ++i) {
        if (tokens[i] < 0 || tokens[i] >= ctx->d.vocab) GL3_FAIL(GL3_E_ARG, "token id out of range");
        if (seq_ids[i] < 0 || seq_ids[i] >= ctx->n_seqs) GL3_FAIL(GL3_E_ARG, "sequence id out of range");
        if (positions[i] < 0 || positions[i] >= ctx->d.ctx) GL3_FAIL(GL3_E_ARG, "position outside the KV cache (context length)");
        for (int j = 0; j < i; ++j) if (seq_ids[j] == seq_ids[i]) GL3_FAIL(GL3_E_ARG, "duplicate sequence id in one batched step");
    }
    return gl3_decode_batch_run(ctx, tokens, seq_ids, positions, n, logits_out, argmax_out);
}

int32_t gl3_profile_decode(gl3_ctx* ctx, int32_t token, int32_t pos, gl3_kernel_times* out) {
    if (!ctx || !out) return GL3_E_ARG;
    memset(out, 0, sizeof(*out));
    int32_t r = set_dyn(ctx, token, pos);
    if (r != GL3_OK) return r;
    return enqueue_decode(ctx, true, out, attn_mode(ctx, pos));
}

int32_t gl3_profile_kernel(gl3_ctx* ctx, int32_t klass, int32_t iters, double* out_us, uint64_t* bytes_per_launch) {
    if (!ctx || !out_us || iters <= 0) return GL3_E_ARG;
    if (!ctx->finalized) GL3_FAIL(GL3_E_STATE, "profile before gl3_finalize");
    if (klass < GL3_K_MATVEC_QKV || klass > GL3_K_OTHER) GL3_FAIL(GL3_E_ARG, "kernel class cannot be profiled");
    GL3_HIP(hipSetDevice(ctx->d.device));
    const gl3_model_desc& d = ctx->d;
    const int rank = d.tp_rank;
    hipEvent_t e0, e1;
    GL3_HIP(hipEventCreate(&e0)); GL3_HIP(hipEventCreate(&e1));
    auto sweep = [&]() {
        const int nl = klass == GL3_K_MATVEC_LOGITS ? 1 : d.n_layers;
        for (int l = 0; l < nl; ++l) {
            gl3_layer& L = ctx->layers[l];
            switch (klass) {
            case GL3_K_MATVEC_QKV: launch_matvec(ctx, PRO_RMS, EPI_STORE, L.wqkv, nullptr, ctx->x, L.attn_norm, ctx->qkv, nullptr); break;
            case GL3_K_MATVEC_WO: launch_matvec(ctx, PRO_QUANT, EPI_RESID, L.wo, nullptr, ctx->xb, nullptr, ctx->qkv, nullptr); break;
            case GL3_K_MATVEC_GATEUP: launch_matvec(ctx, PRO_RMS, EPI_SWIGLU, L.w1, &L.w3, ctx->x, L.ffn_norm, ctx->hb + (size_t)rank * ctx->hidden_l, nullptr); break;
            case GL3_K_MATVEC_DOWN: launch_matvec(ctx, PRO_QUANT, EPI_RESID, L.w2, nullptr, ctx->hb, nullptr, ctx->qkv, nullptr); break;
            case GL3_K_ATTENTION: launch_attention(ctx, l, 0, attn_mode(ctx, ctx->h_dyn[1])); break;   // whole attention at the last position
            case GL3_K_OTHER: launch_attention(ctx, l, 2, attn_mode(ctx, ctx->h_dyn[1]) == ATT_LONG ? ATT_LONG : ATT_MID); break;          // behind the scores only
            default: launch_matvec(ctx, PRO_RMS, EPI_STORE, ctx->wcls, nullptr, ctx->x, ctx->out_norm, ctx->logits + (size_t)rank * ctx->vocab_l, nullptr); break;
            }
        }
        return nl;
    };
    struct Quiet { gl3_ctx* c; ~Quiet() { c->tp_quiet = false; } } quiet{ctx};
    ctx->tp_quiet = true;                               // these launches are outside the step protocol (advisor finding)
    sweep();                                            // warm-up
    GL3_HIP(hipEventRecord(e0, ctx->stream));
    long n = 0;
    for (int i = 0; i < iters; ++i) n += sweep();
    GL3_HIP(hipEventRecord(e1, ctx->stream));
    GL3_HIP(hipEventSynchronize(e1));
    float ms = 0;
    GL3_HIP(hipEventElapsedTime(&ms, e0, e1));
    hipEventDestroy(e0); hipEventDestroy(e1);
    *out_us = (double)ms * 1e3 / (double)n;
    if (bytes_per_launch) {
        const gl3_layer& L = ctx->layers[0];
        switch (klass) {
        case GL3_K_MATVEC_QKV: *bytes_per_launch = mv_bytes(L.wqkv) + d.dim * 4; break;
        case GL3_K_MATVEC_WO: *bytes_per_launch = mv_bytes(L.wo); break;
        case GL3_K_MATVEC_GATEUP: *bytes_per_launch = mv_bytes(L.w1) + L.w3.algo_bytes() + d.dim * 4; break;
        case GL3_K_MATVEC_DOWN: *bytes_per_launch = mv_bytes(L.w2); break;
        case GL3_K_ATTENTION: case GL3_K_OTHER: *bytes_per_launch = 0; break;
        default: *bytes_per_launch = mv_bytes(ctx->wcls) + d.dim * 4; break;
        }
    }
    return GL3_OK;
}

int32_t gl3_pin_host_buffer(gl3_ctx* ctx, void* ptr, uint64_t bytes) {
    if (!ctx) return GL3_E_ARG;
    if (!ptr || !bytes) GL3_FAIL(GL3_E_ARG, "null buffer");
    // Whole pages only.  hipHostRegister pins and GPU-maps the PAGES of the range and hipHostUnregister unmaps them again: a
    // buffer in the middle of the C heap (a small numpy array, a malloc block) shares its pages with whatever the allocator
    // puts beside it, and unmapping them under another registered range gave sporadic "Memory access fault by GPU" aborts in
    // long test runs (round 3).  A page-aligned, page-padded buffer (mmap, posix_memalign, Arena.allocate(n, 4096)) owns its pages.
    if (((uintptr_t)ptr & 4095) || (bytes & 4095)) GL3_FAIL(GL3_E_ARG, "gl3_pin_host_buffer: pointer and size must be multiples of 4096 (the buffer must own its pages)");
    GL3_HIP(hipSetDevice(ctx->d.device));
    GL3_HIP(hipHostRegister(ptr, bytes, hipHostRegisterDefault));
    ctx->pinned.emplace_back(ptr, (size_t)bytes);
    return GL3_OK;
}

int32_t gl3_unpin_host_buffer(gl3_ctx* ctx, void* ptr) {
    if (!ctx) return GL3_E_ARG;
    for (size_t i = 0; i < ctx->pinned.size(); ++i)
        if (ctx->pinned[i].first == ptr) {
            GL3_HIP(hipSetDevice(ctx->d.device));
            GL3_HIP(hipStreamSynchronize(ctx->stream));
            hipHostUnregister(ptr);
            ctx->pinned.erase(ctx->pinned.begin() + i);
            return GL3_OK;
        }
    GL3_FAIL(GL3_E_ARG, "buffer was not pinned through this plan");
}

// ---------------------------------------------------------------------------------------------------
// gl3_probe_peaks: what this device actually sustains ("re-verify on the box, report both spec and measured")
typedef int probe_v4i __attribute__((ext_vector_type(4)));
typedef int probe_v16i __attribute__((ext_vector_type(16)));
typedef float probe_v4f __attribute__((ext_vector_type(4)));
static __global__ __launch_bounds__(256) void probe_read_kernel(const float4* __restrict__ src4, size_t n4, float* __restrict__ sink) {
    const probe_v4f* src = reinterpret_cast<const probe_v4f*>(src4);
    // every workgroup streams its own contiguous slice, eight independent non-temporal 16-byte loads per thread in flight
    const size_t per = n4 / gridDim.x, base = (size_t)blockIdx.x * per;
    float4 a = {0.f, 0.f, 0.f, 0.f};
    size_t i = threadIdx.x;
    for (; i + 7 * 256 < per; i += 8 * 256) {
        probe_v4f v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = __builtin_nontemporal_load(src + base + i + u * 256);
#pragma unroll
        for (int u = 0; u < 8; ++u) { a.x += v[u].x; a.y += v[u].y; a.z += v[u].z; a.w += v[u].w; }
    }
    for (; i < per; i += 256) { const probe_v4f v = src[base + i]; a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w; }
    if (a.x + a.y + a.z + a.w == 12345.678f) sink[0] = a.x;        // keeps the loads alive
}
static __global__ __launch_bounds__(256) void probe_copy_kernel(const float4* __restrict__ src, float4* __restrict__ dst, size_t n4) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) dst[i] = src[i];
}
static __global__ __launch_bounds__(256) void probe_mfma_kernel(int iters, int* __restrict__ sink) {
    probe_v16i c0, c1, c2, c3;
    for (int r = 0; r < 16; ++r) { c0[r] = 0; c1[r] = 1; c2[r] = 2; c3[r] = 3; }
    const probe_v4i a = {0x01020304, 0x05060708, (int)threadIdx.x, 0x01010101};
    for (int i = 0; i < iters; ++i) {
        c0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, a, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, a, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, a, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, a, c3, 0, 0, 0);
    }
    int s = 0;
    for (int r = 0; r < 16; ++r) s += c0[r] + c1[r] + c2[r] + c3[r];
    if (s == 0x7fffffff) sink[0] = s;
}

int32_t gl3_probe_peaks(int32_t device, double* hbm_read_gbs, double* hbm_copy_gbs, double* int8_mfma_tops) {
    if (hipSetDevice(device) != hipSuccess) return GL3_E_HIP;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) return GL3_E_HIP;
    const int cus = prop.multiProcessorCount;
    const size_t bytes = (size_t)1 << 30, n4 = bytes / 16;
    float4 *src = nullptr, *dst = nullptr;
    float* sink = nullptr;
    hipStream_t s = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    int32_t rc = GL3_E_HIP;
    do {
        if (hipMalloc((void**)&src, 3 * bytes) != hipSuccess) { rc = GL3_E_OOM; break; }      // three regions, read in rotation: nothing of a
                                                                                               // repetition's 1 GiB is still in the 256 MB Infinity Cache
        if (hipMalloc((void**)&dst, bytes) != hipSuccess) { rc = GL3_E_OOM; break; }
        if (hipMalloc((void**)&sink, 256) != hipSuccess) break;
        if (hipStreamCreate(&s) != hipSuccess || hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) break;
        if (hipMemsetAsync(src, 1, 3 * bytes, s) != hipSuccess || hipMemsetAsync(dst, 0, bytes, s) != hipSuccess) break;
        int region = 0;
        auto timed = [&](auto&& launch) -> double {                 // best of 3 after one warm-up, ms
            double best = 1e30;
            for (int rep = 0; rep < 4; ++rep) {
                region = (region + 1) % 3;
                hipEventRecord(e0, s);
                launch();
                hipEventRecord(e1, s);
                hipEventSynchronize(e1);
                float ms = 0;
                hipEventElapsedTime(&ms, e0, e1);
                if (rep > 0 && ms < best) best = ms;
            }
            return best;
        };
        const dim3 grid(cus * 8), block(256);
        if (hbm_read_gbs) {                                         // best over a few grid sizes (2 ... 16 workgroups per CU)
            double best = 0;
            for (int wpc = 2; wpc <= 16; wpc *= 2) {
                const double gbs = bytes / (timed([&] { hipLaunchKernelGGL(probe_read_kernel, dim3(cus * wpc), block, 0, s, src + (size_t)region * n4, n4, sink); }) * 1e6);
                if (gbs > best) best = gbs;
            }
            *hbm_read_gbs = best;
        }
        if (hbm_copy_gbs) *hbm_copy_gbs = 2.0 * bytes / (timed([&] { hipLaunchKernelGGL(probe_copy_kernel, grid, block, 0, s, src + (size_t)region * n4, dst, n4); }) * 1e6);
        if (int8_mfma_tops) {
            const int iters = 4096;
            const double ms = timed([&] { hipLaunchKernelGGL(probe_mfma_kernel, dim3(cus), block, 0, s, iters, (int*)sink); });
            // 4 wavefronts per workgroup x 4 MFMAs per trip x 2 * 32 * 32 * 32 int8 operations
            *int8_mfma_tops = (double)cus * 4.0 * iters * 4.0 * 65536.0 / (ms * 1e9);
        }
        rc = hipGetLastError() == hipSuccess ? GL3_OK : GL3_E_HIP;
    } while (0);
    if (e0) hipEventDestroy(e0);
    if (e1) hipEventDestroy(e1);
    if (s) hipStreamDestroy(s);
    if (src) hipFree(src);
    if (dst) hipFree(dst);
    if (sink) hipFree(sink);
    return rc;
}

int32_t gl3_profile_prefill_kernel(gl3_ctx* ctx, int32_t klass, int32_t n_tokens, int32_t iters, double* out_us, uint64_t* int8_ops_per_launch) {
    if (!ctx || !out_us || iters <= 0) return GL3_E_ARG;
    if (!ctx->finalized) GL3_FAIL(GL3_E_STATE, "profile before gl3_finalize");
    if (klass < GL3_K_MATVEC_QKV || klass > GL3_K_MATVEC_DOWN) GL3_FAIL(GL3_E_ARG, "kernel class is not a batched-prefill GEMM");
    return gl3_prefill_profile(ctx, klass, n_tokens, iters, out_us, int8_ops_per_launch);
}

int32_t gl3_get_x(gl3_ctx* ctx, float* out) {
    if (!ctx || !out) return GL3_E_ARG;
    GL3_HIP(hipSetDevice(ctx->d.device));
    GL3_HIP(hipMemcpy(out, ctx->x, sizeof(float) * ctx->d.dim, hipMemcpyDeviceToHost));
    return GL3_OK;
}

int32_t gl3_get_layer_x(gl3_ctx* ctx, int32_t layer, float* out) {
    if (!ctx || !out) return GL3_E_ARG;
    if (!ctx->taps) GL3_FAIL(GL3_E_STATE, "plan was created without GL3_FLAG_LAYER_TAPS");
    if (layer < 0 || layer >= ctx->d.n_layers) GL3_FAIL(GL3_E_ARG, "layer out of range");
    GL3_HIP(hipSetDevice(ctx->d.device));
    GL3_HIP(hipMemcpy(out, ctx->taps + (size_t)layer * ctx->d.dim, sizeof(float) * ctx->d.dim, hipMemcpyDeviceToHost));
    return GL3_OK;
}

int32_t gl3_get_kv_seq(gl3_ctx* ctx, int32_t seq, int32_t layer, int32_t pos, float* k_out, float* v_out) {
    if (!ctx || !k_out || !v_out) return GL3_E_ARG;
    if (seq < 0 || seq >= ctx->n_seqs || layer < 0 || layer >= ctx->d.n_layers || pos < 0 || pos >= ctx->d.ctx)
        GL3_FAIL(GL3_E_ARG, "sequence/layer/position out of range");
    GL3_HIP(hipSetDevice(ctx->d.device));
    const size_t off = (size_t)seq * ctx->kv_seq_stride + ((size_t)layer * ctx->d.ctx + pos) * ctx->kv_dim_l;
    GL3_HIP(hipMemcpy(k_out, ctx->kcache + off, sizeof(float) * ctx->kv_dim_l, hipMemcpyDeviceToHost));
    GL3_HIP(hipMemcpy(v_out, ctx->vcache + off, sizeof(float) * ctx->kv_dim_l, hipMemcpyDeviceToHost));
    return GL3_OK;
}

int32_t gl3_get_kv(gl3_ctx* ctx, int32_t layer, int32_t pos, float* k_out, float* v_out) {
    return gl3_get_kv_seq(ctx, 0, layer, pos, k_out, v_out);
}

int32_t gl3_get_buffer(gl3_ctx* ctx, int32_t which, float* out, uint64_t n) {
    if (!ctx || !out) return GL3_E_ARG;
    const float* src = nullptr;
    uint64_t cap = 0;
    switch (which) {
    case 0: src = ctx->qkv; cap = ctx->q_dim_l + 2 * ctx->kv_dim_l; break;
    case 1: src = ctx->xb; cap = ctx->q_dim; break;
    case 2: src = ctx->hb; cap = ctx->d.hidden; break;
    case 3: src = ctx->logits; cap = ctx->d.vocab; break;
    case 4: case 5: case 6:                    // batched step: rank-chunked X / AO / HB of the last chunk (GB_PF_X / _AO / _HB), max_batch rows
        if (!ctx->pf) GL3_FAIL(GL3_E_STATE, "no batched-prefill buffers (max_batch <= 1)");
        src = gl3_prefill_buf(ctx, which);
        cap = (uint64_t)ctx->d.max_batch * (which == 4 ? ctx->d.dim : which == 5 ? ctx->q_dim : ctx->d.hidden);
        break;
    case 7:                                     // qwen2moe: routing weights of the last layer [n_experts_used] + the shared-expert gate
        if (!ctx->moe_w) GL3_FAIL(GL3_E_STATE, "not a qwen2moe plan");
        src = ctx->moe_w; cap = ctx->d.n_experts_used + 1;
        break;
    case 8: {                                   // qwen2moe: selected expert ids of the last layer, as floats
        if (!ctx->moe_sel) GL3_FAIL(GL3_E_STATE, "not a qwen2moe plan");
        if (n > (uint64_t)ctx->d.n_experts_used) GL3_FAIL(GL3_E_ARG, "buffer shorter than requested");
        GL3_HIP(hipSetDevice(ctx->d.device));
        GL3_HIP(hipStreamSynchronize(ctx->stream));
        std::vector<int> ids(n);
        GL3_HIP(hipMemcpy(ids.data(), ctx->moe_sel, n * sizeof(int), hipMemcpyDeviceToHost));
        for (uint64_t i = 0; i < n; ++i) out[i] = (float)ids[i];
        return GL3_OK;
    }
    default: GL3_FAIL(GL3_E_ARG, "unknown buffer id");
    }
    if (n > cap) GL3_FAIL(GL3_E_ARG, "buffer shorter than requested");
    GL3_HIP(hipSetDevice(ctx->d.device));
    GL3_HIP(hipMemcpy(out, src, n * sizeof(float), hipMemcpyDeviceToHost));
    return GL3_OK;
}

int32_t gl3_reset_kv(gl3_ctx* ctx) {
    if (!ctx) return GL3_E_ARG;
    GL3_HIP(hipSetDevice(ctx->d.device));
    const size_t kvn = ctx->kv_seq_stride * ctx->n_seqs;
    GL3_HIP(hipMemsetAsync(ctx->kcache, 0, kvn * 4, ctx->stream));
    GL3_HIP(hipMemsetAsync(ctx->vcache, 0, kvn * 4, ctx->stream));
    GL3_HIP(hipStreamSynchronize(ctx->stream));
    return GL3_OK;
}

int32_t gl3_get_init_ms(gl3_ctx* ctx, double* plan_ms, double* copy_in_ms) {
    if (!ctx) return GL3_E_ARG;
    if (plan_ms) *plan_ms = ctx->plan_ms;
    if (copy_in_ms) *copy_in_ms = ctx->copy_in_ms;
    return GL3_OK;
}

}  // extern "C"
