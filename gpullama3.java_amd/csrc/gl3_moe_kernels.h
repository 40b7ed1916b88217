// gl3_moe_kernels.h — the routing side of the Qwen2-MoE feed-forward block (decode step).
//
// Replaces Qwen2MoEKernels.softmaxAndTopK and the router / shared-gate tasks of Qwen2MoEQ8_0FFNLayers
// (J/tornadovm/kernels/Qwen2MoEKernels.java:37-100, J/tornadovm/layers/type/q8_0/Qwen2MoEQ8_0FFNLayers.java) with the ARITHMETIC
// of the CPU path InferenceCore.forwardJavaQwen2MoE (J/inference/InferenceCore.java:363-415), bit for bit:
//   * router logits and the shared-expert gate score are FP32FloatTensor dots = FloatTensor.scalarDot (:86-92): products rounded
//     to f32 and added strictly left to right (the terms are signed, so the monotone-sum shortcut of gl3_seqsum.h does not apply;
//     one workgroup per expert row runs the chain from LDS);
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

// logits[r] = gate_inp[r] . xn for r < n_experts (workgroup r); workgroup n_experts: shared-expert gate -> w_out[topk] = sigmoid.
// xn = rmsnorm(x) in f32 (rmsnorm_f32_kernel).  LDS: p[dim] products.
static __global__ __launch_bounds__(256) void moe_router_kernel(const float* __restrict__ gate_inp, const float* __restrict__ gate_inp_shexp,
                                                                const float* __restrict__ xn, int dim, int n_experts, int topk,
                                                                float* __restrict__ logits, float* __restrict__ w_out) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    float* p = reinterpret_cast<float*>(smem);
    const int r = blockIdx.x, t = threadIdx.x;
    const float* row = r < n_experts ? gate_inp + (size_t)r * dim : gate_inp_shexp;
    for (int i = t; i < dim; i += 256) p[i] = row[i] * xn[i];
    __syncthreads();
    if (t < 64) {
        const float s = seq_sum_lds<false>(p, dim);
        if (t == 0) {
            if (r < n_experts) logits[r] = s;
            else w_out[topk] = 1.0f / (1.0f + (float)exp(-(double)s));          // :414
        }
    }
}

// One wavefront: softmax over the n_experts logits, then the top-k scan.  sel[i] / w_out[i] = i-th selected expert and its
// probability.  LDS: e[n_experts + 4].
static __global__ __launch_bounds__(64) void moe_select_kernel(const float* __restrict__ logits, int n_experts, int topk,
                                                               int* __restrict__ sel, float* __restrict__ w_out) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    float* e = reinterpret_cast<float*>(smem);
    const int t = threadIdx.x;
    float m = -INFINITY;
    for (int i = t; i < n_experts; i += 64) m = fmaxf(m, logits[i]);
    m = wave_max(m);
    for (int i = t; i < n_experts; i += 64) e[i] = (float)exp((double)(logits[i] - m));
    __syncthreads();
    const float sum = seq_sum_lds<false>(e, n_experts);
    __syncthreads();
    for (int i = t; i < n_experts; i += 64) e[i] = e[i] / sum;
    __syncthreads();
    if (t == 0) {
        for (int k = 0; k < topk; ++k) {
            float best = -INFINITY;
            int index = -1;
            for (int j = 0; j < n_experts; ++j) {
                const float v = e[j];
                if (v > best) { best = v; index = j; }
            }
            if (index < 0) index = 0;                      // all NaN: the reference would throw; keep the launch in bounds
            sel[k] = index;
            w_out[k] = best;
            e[index] = -INFINITY;
        }
    }
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
