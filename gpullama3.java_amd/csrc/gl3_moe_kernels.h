// gl3_moe_kernels.h — the routing side of the Qwen2-MoE feed-forward block (decode step).
//
// Replaces Qwen2MoEKernels.softmaxAndTopK and the router / shared-gate tasks of Qwen2MoEQ8_0FFNLayers
// (J/tornadovm/kernels/Qwen2MoEKernels.java:37-100, J/tornadovm/layers/type/q8_0/Qwen2MoEQ8_0FFNLayers.java) with the ARITHMETIC
// of the CPU path InferenceCore.forwardJavaQwen2MoE (J/inference/InferenceCore.java:363-415), bit for bit:
//   * router logits and the shared-expert gate score are FP32FloatTensor dots = FloatTensor.scalarDot (:86-92): products rounded
//     to f32 and added strictly left to right (the terms are signed, so the monotone-sum shortcut of gl3_seqsum.h does not apply;
//     one workgroup per expert row normalises x itself, writes the products to LDS and runs the chain there);
//   * softmaxInPlace over ALL experts (:374; FloatTensor.java:211-219: max, (float)exp(x - max) in double, sequential sum, divide);
//   * top-k by repeated strict-> scans, first index wins, the probabilities are NOT renormalised (:376-390);
//   * x = w_j * y_j + x per selected expert in selection order, then the shared expert with w = 1 / (1 + (float)exp(-score))
//     (:392-415, saxpyInPlace = a * that + this with two roundings).
// The expert matrices themselves run on matvec_q8t_kernel<.., SEL = true> (gl3_decode_kernels.h), which reads the expert id a
// launch slot works on from the selection buffer written here — no host round trip, so the whole step stays one hipGraph.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "gl3_decode_kernels.h"

namespace gl3 {

struct MoeRouterArgs {
    const float* x;               // residual stream f32[dim]
    const float* norm_w;          // ffn_norm.weight
    float eps;
    const float* gate_inp;        // router rows f32[n_experts][dim]
    const float* gate_inp_shexp;  // shared-expert gate row f32[dim]
    int dim, n_experts, topk;
    float* logits;                // [n_experts] router logits (scratch between the workgroups)
    float* w_out;                 // [topk + 1]: probabilities of the selected experts, then the shared expert's sigmoid gate
    int* sel;                     // [topk] selected expert ids
    int* ticket;                  // workgroup arrival counter (0 between launches)
};

// Row r < n_experts: logits[r] = gate_inp[r] . xb; row n_experts: the shared-expert gate -> w_out[topk] = sigmoid.  A workgroup owns
// MOE_RR consecutive rows.  It computes xb = rmsnorm(x) itself (exact in-order sum of squares, as the matvec prologues do — no separate
// normalisation launch), writes the products of its rows to LDS in parallel, and then ONE wavefront adds them up with lane = row: the
// dim dependent f32 adds of scalarDot are the critical path of the whole MoE block, and a lane-private chain over 16-byte LDS reads
// runs at the VALU's dependent-issue rate (measured: 24.5 us -> see DESIGN 5d for the one-row-per-workgroup version it replaces).
// The LAST workgroup to arrive (ticket) runs softmax + top-k over the logits of all of them.
//   LDS: xf[dim + 32] | exact_sumsq scratch | e[n_experts + 4] | red[4] | P[MOE_RR][dim + 4] products
constexpr int MOE_RR = 8;      // rows per workgroup; P rows are dim + 4 floats apart, so the 16-byte reads of up to 16 lanes hit distinct banks
constexpr int MOE_QJ = 2;      // quads per thread and row held in registers ahead of the sum of squares (dim <= 2048; longer rows load late)

__host__ __device__ inline size_t moe_router_smem(int dim, int n_experts) {
    return (size_t)(dim + 32) * 4 + ss_scratch_bytes(dim) + (size_t)((n_experts + 7) & ~3) * 4 + 64 + (size_t)MOE_RR * (dim + 4) * 4;
}
__host__ __device__ inline int moe_router_wgs(int n_experts) { return (n_experts + 1 + MOE_RR - 1) / MOE_RR; }

static __global__ __launch_bounds__(256) void moe_router_kernel(const MoeRouterArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    __shared__ int last;
    float* xf = reinterpret_cast<float*>(smem);
    uint8_t* scratch = smem + (size_t)(a.dim + 32) * 4;
    float* e = reinterpret_cast<float*>(scratch + ss_scratch_bytes(a.dim));
    float* red = e + ((a.n_experts + 7) & ~3);
    float* P = red + 16;
    const int t = threadIdx.x, k = a.dim, E = a.n_experts, nq = k >> 2, pitch = k + 4;
    const int row0 = blockIdx.x * MOE_RR;
    auto row_ptr = [&](int r) {                            // rows past the shared-gate row (padding of the last workgroup) alias it
        const int rr = row0 + r;
        return rr < E ? a.gate_inp + (size_t)rr * k : a.gate_inp_shexp;
    };
    // the router rows are cold in HBM: request this thread's quads of every row (and the norm weights) before the sum of squares
    float4 wv[MOE_RR][MOE_QJ], nwv[MOE_QJ];
#pragma unroll
    for (int j = 0; j < MOE_QJ; ++j) {
        const int q = min(t + 256 * j, nq - 1);            // clamped, unconditional (see the matvec prologue)
        nwv[j] = *reinterpret_cast<const float4*>(a.norm_w + 4 * q);
#pragma unroll
        for (int r = 0; r < MOE_RR; ++r) wv[r][j] = *reinterpret_cast<const float4*>(row_ptr(r) + 4 * q);
    }
    for (int i = t; i < k + 32; i += 256) xf[i] = i < k ? a.x[i] : 0.f;
    __syncthreads();
    float ss;                                              // InferenceCore.rmsnorm :39-48, as rmsnorm_f32_kernel
    if (k >= 1024 && k <= 5120 && (k & 3) == 0) {
        BlockBarrier bb;
        ss = exact_sumsq_lds(xf, k, scratch, t, bb);
    } else {
        if (t < 64) { const float s1 = seq_sum_lds<true>(xf, k); if (t == 0) red[0] = s1; }
        __syncthreads();
        ss = red[0];
    }
    ss /= (float)k;
    ss += a.eps;
    const float scale = (float)(1.0 / sqrt((double)ss));
    // scalarDot's products row[i] * xb[i] with xb[i] = w[i] * (scale * x[i]) (rmsnorm :47)
    auto products = [&](int q, const float4& nw, auto&& wq) {
        const float4 xv = *reinterpret_cast<const float4*>(xf + 4 * q);
        float4 xb;
        xb.x = nw.x * (scale * xv.x); xb.y = nw.y * (scale * xv.y); xb.z = nw.z * (scale * xv.z); xb.w = nw.w * (scale * xv.w);
#pragma unroll
        for (int r = 0; r < MOE_RR; ++r) {
            const float4 w = wq(r);
            float4 pv;
            pv.x = w.x * xb.x; pv.y = w.y * xb.y; pv.z = w.z * xb.z; pv.w = w.w * xb.w;
            *reinterpret_cast<float4*>(P + (size_t)r * pitch + 4 * q) = pv;
        }
    };
#pragma unroll
    for (int j = 0; j < MOE_QJ; ++j) {
        const int q = t + 256 * j;
        if (q < nq) products(q, nwv[j], [&](int r) { return wv[r][j]; });
    }
    for (int q = t + 256 * MOE_QJ; q < nq; q += 256)
        products(q, *reinterpret_cast<const float4*>(a.norm_w + 4 * q), [&](int r) { return *reinterpret_cast<const float4*>(row_ptr(r) + 4 * q); });
    __syncthreads();
    if (t < 64) {
        float s = 0.f;
        if (t < MOE_RR) {                                  // lane = row: result += product, element order; 16 elements in flight ahead
            // Three groups of 16 products are in flight ahead of the adds (ring A -> B -> C): an LDS read returns after ~100+
            // cycles, a group of 16 dependent adds takes ~80.  The sched_barrier after every refill keeps the program order
            // "add group A, refill A, add group B, refill B, ..." — left alone, the scheduler sinks the reads next to their uses
            // and every group waits for LDS (seen in the ISA of the first version: s_waitcnt lgkmcnt(0) two adds after the read).
            const float* pr = P + (size_t)t * pitch;
            const int G = k >> 4;                               // groups of 16 (dim is a multiple of 32)
            auto ld = [&](int g, float4& v0, float4& v1, float4& v2, float4& v3) {
                const float* q = pr + 16 * min(g, G - 1);       // past the end: re-read the last group, never added
                v0 = *reinterpret_cast<const float4*>(q); v1 = *reinterpret_cast<const float4*>(q + 4);
                v2 = *reinterpret_cast<const float4*>(q + 8); v3 = *reinterpret_cast<const float4*>(q + 12);
            };
#define MOE_ADD16(v0, v1, v2, v3) do { s = seq_add4<false>(s, v0); s = seq_add4<false>(s, v1); s = seq_add4<false>(s, v2); s = seq_add4<false>(s, v3); } while (0)
#define MOE_ORDER() __builtin_amdgcn_sched_barrier(0)
            float4 a0, a1, a2, a3, b0, b1, b2, b3, c0, c1, c2, c3;
            ld(0, a0, a1, a2, a3); ld(1, b0, b1, b2, b3); ld(2, c0, c1, c2, c3);
            int g = 0;
            for (; g + 3 <= G; g += 3) {
                MOE_ADD16(a0, a1, a2, a3); ld(g + 3, a0, a1, a2, a3); MOE_ORDER();
                MOE_ADD16(b0, b1, b2, b3); ld(g + 4, b0, b1, b2, b3); MOE_ORDER();
                MOE_ADD16(c0, c1, c2, c3); ld(g + 5, c0, c1, c2, c3); MOE_ORDER();
            }
            if (g < G) { MOE_ADD16(a0, a1, a2, a3); ++g; }
            if (g < G) { MOE_ADD16(b0, b1, b2, b3); ++g; }
#undef MOE_ORDER
#undef MOE_ADD16
            const int rr = row0 + t;
            if (rr < E) __hip_atomic_store(a.logits + rr, s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else if (rr == E) __hip_atomic_store(a.w_out + a.topk, 1.0f / (1.0f + (float)exp(-(double)s)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // :414
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        if (t == 0) {
            const int tk = __hip_atomic_fetch_add(a.ticket, 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
            last = tk == (int)gridDim.x - 1;
        }
    }
    __syncthreads();
    if (!last || t >= 64) return;
    // ---- the last workgroup: softmaxInPlace over all experts (:374), then the top-k scan (:376-390)
    float m = -INFINITY;
    for (int i = t; i < E; i += 64) {
        const float v = __hip_atomic_load(a.logits + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        e[i] = v;
        m = fmaxf(m, v);
    }
    m = wave_max(m);
    for (int i = t; i < E; i += 64) e[i] = (float)exp((double)(e[i] - m));
    __syncthreads();                                       // only this wavefront is left (retired wavefronts do not count)
    const float sum = seq_sum_lds<false>(e, E);
    __syncthreads();
    for (int i = t; i < E; i += 64) e[i] = e[i] / sum;
    __syncthreads();
    // top-k: k rounds of "first index of the maximum" (strict > in ascending index order = the lowest index wins a tie), each a
    // per-lane scan of its strided elements and a wavefront reduction
    for (int kk = 0; kk < a.topk; ++kk) {
        float best = -INFINITY;
        int index = 0x7FFFFFFF;
        for (int j = t; j < E; j += 64) amx_take(best, index, e[j], j);
        for (int mk = 32; mk >= 1; mk >>= 1) {
            const float ob = __shfl_xor(best, mk, 64);
            const int oi = __shfl_xor(index, mk, 64);
            amx_take(best, index, ob, oi);
        }
        if (index == 0x7FFFFFFF) index = 0;                // nothing above -inf (NaN logits): the reference would throw; stay in bounds
        if (t == 0) { a.sel[kk] = index; a.w_out[kk] = best; }
        if ((index & 63) == t) e[index] = -INFINITY;
        __syncthreads();
    }
    if (t == 0) __hip_atomic_store(a.ticket, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// x[i] = w[j] * y[j][i] + x[i] for j = 0 .. n_terms - 1 in order (the selected experts, then the shared expert)
static __global__ __launch_bounds__(256) void moe_combine_kernel(float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ w,
                                                                 int dim, int n_terms) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= dim) return;
    float v = x[i];
    for (int j = 0; j < n_terms; ++j) {
        const float prod = w[j] * y[(size_t)j * dim + i];
        v = prod + v;
    }
    x[i] = v;
}

}  // namespace gl3
