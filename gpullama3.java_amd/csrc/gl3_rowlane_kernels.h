// gl3_rowlane_kernels.h — decode matvec for the weight types whose reference dot product is an ELEMENT-wise f32 chain:
//   F16  : FP16FloatTensor.dot, scalar mode   (J/tensor/standard/FP16FloatTensor.java:54-60, getFloat :48-51):
//              result += float16ToFloat(w[j]) * x[j],               j ascending
//   Q4_0 : Q4_0FloatTensor.getFloat + FloatTensor.scalarDot (J/tensor/standard/Q4_0FloatTensor.java:57-71,
//          J/tensor/standard/FloatTensor.java:86-92):
//              result += ((float)(nibble - 8) * float16ToFloat(d)) * x[j],   j ascending; element j < 16 of a block is the
//              LOW nibble of byte j, element j >= 16 the HIGH nibble of byte j - 16
// Neither quantises the activation.  A row's K-long chain cannot be split without changing the f32 result, so the
// parallelism is across rows: lane = one output row ("row-lane"), 64 rows per wavefront, and the weights are repacked
// at upload so that a wavefront's loads are contiguous:
//   F16  "RL": [row group g = row/64][chunk c = 8 elements][lane = row%64][16 B]              (1024 B per (g, c))
//   Q4_0 "RL": [row group g][block b][64 x f16 d (128 B)][lane][16 B of nibbles (1024 B)]     (1152 B per (g, b))
// The activation vector (already RMS-normalised by rmsnorm_f32_kernel when the reference normalises first) sits in LDS
// and is read as wavefront-uniform float4 broadcasts.  The kernels are latency-bound by the chain itself
// (K dependent adds per row); they exist for parity on §8 rows a5 / a6, the HBM-roofline target is the Q8_0 path.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "gl3_decode_kernels.h"

namespace gl3 {

enum { WT_F16 = 1, WT_Q4_0 = 2 };

__host__ __device__ inline size_t rl_group_bytes(int wt, int k) {       // bytes of one 64-row group
    return wt == WT_F16 ? (size_t)(k / 8) * 1024 : (size_t)(k / 32) * 1152;
}

// GGUF row-major -> RL.  src holds `rows` rows of this matrix slice; they land at rows dst_row0.. of dst.
// One thread per (row, 16-byte unit): F16 unit = 8 halfs, Q4_0 unit = one block (scale + 16 bytes).
template <int WT>
static __global__ __launch_bounds__(256) void repack_rl_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, int rows,
                                                               int k, int dst_row0) {
    const int units = WT == WT_F16 ? k / 8 : k / 32;
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long)rows * units) return;
    const int r = (int)(i / units), u = (int)(i % units);
    const int row = dst_row0 + r, g = row >> 6, lane = row & 63;
    uint8_t* gb = dst + (size_t)g * rl_group_bytes(WT, k);
    if (WT == WT_F16) {
        const uint8_t* s = src + ((size_t)r * units + u) * 16;
        uint8_t* d = gb + (size_t)u * 1024 + lane * 16;
        for (int b = 0; b < 16; ++b) d[b] = s[b];
    } else {
        const uint8_t* s = src + ((size_t)r * units + u) * 18;
        uint8_t* d = gb + (size_t)u * 1152;
        d[2 * lane] = s[0]; d[2 * lane + 1] = s[1];
        for (int b = 0; b < 16; ++b) d[128 + lane * 16 + b] = s[2 + b];
    }
}

// token_embedding_table.copyTo -> getFloat per element (InferenceCore.java:61)
template <int WT>
static __global__ __launch_bounds__(256) void embed_rl_kernel(const uint8_t* __restrict__ emb, int dim, const int* __restrict__ dyn,
                                                              float* __restrict__ x) {
    const int token = dyn[0], g = token >> 6, lane = token & 63;
    const uint8_t* gb = emb + (size_t)g * rl_group_bytes(WT, dim);
    for (int i = threadIdx.x; i < dim; i += 256) {
        if (WT == WT_F16) {
            x[i] = h2f(*reinterpret_cast<const uint16_t*>(gb + (size_t)(i >> 3) * 1024 + lane * 16 + 2 * (i & 7)));
        } else {
            const uint8_t* b = gb + (size_t)(i >> 5) * 1152;
            const int j = i & 31;
            const uint8_t byte = b[128 + lane * 16 + (j & 15)];
            const int q = j < 16 ? (byte & 0x0F) : (byte >> 4);
            x[i] = (float)(q - 8) * h2f(*reinterpret_cast<const uint16_t*>(b + 2 * lane));
        }
    }
}

// out[i] = w[i] * (ss * x[i]) with the exact in-order sum of squares (InferenceCore.rmsnorm :39-48).  One workgroup.
static __global__ __launch_bounds__(256) void rmsnorm_f32_kernel(const float* __restrict__ x, int k, const float* __restrict__ w, float eps,
                                                                 float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    float* xf = reinterpret_cast<float*>(smem);                 // [k + 32]
    uint8_t* scratch = smem + (size_t)(k + 32) * 4;             // ss_scratch_bytes(k)
    float* red = reinterpret_cast<float*>(scratch + ss_scratch_bytes(k));
    const int t = threadIdx.x;
    for (int i = t; i < k + 32; i += 256) xf[i] = i < k ? x[i] : 0.f;
    __syncthreads();
    float ss;
    if (k >= 1024 && k <= 5120 && (k & 3) == 0) {
        BlockBarrier bb;
        ss = exact_sumsq_lds(xf, k, scratch, t, bb);
    } else {
        if (t < 64) { const float s1 = seq_sum_lds<true>(xf, k); if (t == 0) red[0] = s1; }
        __syncthreads();
        ss = red[0];
    }
    ss /= (float)k;
    ss += eps;
    const float scale = (float)(1.0 / sqrt((double)ss));
    for (int i = t; i < k; i += 256) out[i] = w[i] * (scale * xf[i]);
}

struct RlArgs {
    const uint8_t* w; const uint8_t* w2;    // RL matrices (w2: the "up" matrix of the SwiGLU pair)
    int rows, k;
    const float* x;                         // f32[k] activation (normalised where the reference normalises)
    float* out; const float* resid_in;      // EPI_RESID: out[i] = resid_in[i] + result
};

// Workgroup = 4 wavefronts = 4 row groups; the activation is staged in LDS once per workgroup.
template <int WT, int EPI>
static __global__ __launch_bounds__(256) void matvec_rl_kernel(const RlArgs a) {
    extern __shared__ __attribute__((aligned(16))) float xs[];   // [k]
    constexpr int NM = EPI == EPI_SWIGLU ? 2 : 1;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    for (int i = t; i < (a.k >> 2); i += 256) *reinterpret_cast<float4*>(xs + 4 * i) = *reinterpret_cast<const float4*>(a.x + 4 * i);
    __syncthreads();
    const int g = blockIdx.x * 4 + wave;
    if (g * 64 >= a.rows) return;
    const size_t gbytes = rl_group_bytes(WT, a.k);
    const uint8_t* wp[NM];
    wp[0] = a.w + (size_t)g * gbytes;
    if (NM == 2) wp[NM - 1] = a.w2 + (size_t)g * gbytes;
    float res[NM];
#pragma unroll
    for (int m = 0; m < NM; ++m) res[m] = 0.f;
    typedef int v4i __attribute__((ext_vector_type(4)));
    if (WT == WT_F16) {
        const int nch = a.k >> 3;
        constexpr int U = 4;                                     // chunks in flight
        int c = 0;
        for (; c + U <= nch; c += U) {
            v4i wv[NM][U];
#pragma unroll
            for (int m = 0; m < NM; ++m)
#pragma unroll
                for (int u = 0; u < U; ++u) wv[m][u] = __builtin_nontemporal_load(reinterpret_cast<const v4i*>(wp[m] + (size_t)(c + u) * 1024 + lane * 16));
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const float4 x0 = *reinterpret_cast<const float4*>(xs + 8 * (c + u)), x1 = *reinterpret_cast<const float4*>(xs + 8 * (c + u) + 4);
                const float xv[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
#pragma unroll
                for (int m = 0; m < NM; ++m)
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const uint32_t word = (uint32_t)wv[m][u][i >> 1];
                        res[m] = res[m] + h2f((uint16_t)((i & 1) ? word >> 16 : word & 0xFFFF)) * xv[i];
                    }
            }
        }
        for (; c < nch; ++c)
#pragma unroll
            for (int m = 0; m < NM; ++m) {
                const v4i wv = *reinterpret_cast<const v4i*>(wp[m] + (size_t)c * 1024 + lane * 16);
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const uint32_t word = (uint32_t)wv[i >> 1];
                    res[m] = res[m] + h2f((uint16_t)((i & 1) ? word >> 16 : word & 0xFFFF)) * xs[8 * c + i];
                }
            }
    } else {
        const int nb = a.k >> 5;
        constexpr int U = 2;                                     // blocks in flight
        auto block = [&](int m, int b, const v4i& qv, uint16_t dh) {
            const float d = h2f(dh);
            const float* xb = xs + 32 * b;
#pragma unroll
            for (int half = 0; half < 2; ++half)                 // elements 0..15 = low nibbles, 16..31 = high nibbles
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    const uint32_t word = (uint32_t)qv[j >> 2];
                    const int q = (int)((word >> (8 * (j & 3) + 4 * half)) & 0xF);
                    res[m] = res[m] + ((float)(q - 8) * d) * xb[16 * half + j];
                }
        };
        int b = 0;
        for (; b + U <= nb; b += U) {
            v4i qv[NM][U];
            uint16_t dh[NM][U];
#pragma unroll
            for (int m = 0; m < NM; ++m)
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const uint8_t* p = wp[m] + (size_t)(b + u) * 1152;
                    dh[m][u] = *reinterpret_cast<const uint16_t*>(p + 2 * lane);
                    qv[m][u] = __builtin_nontemporal_load(reinterpret_cast<const v4i*>(p + 128 + lane * 16));
                }
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int m = 0; m < NM; ++m) block(m, b + u, qv[m][u], dh[m][u]);
        }
        for (; b < nb; ++b)
#pragma unroll
            for (int m = 0; m < NM; ++m) {
                const uint8_t* p = wp[m] + (size_t)b * 1152;
                block(m, b, *reinterpret_cast<const v4i*>(p + 128 + lane * 16), *reinterpret_cast<const uint16_t*>(p + 2 * lane));
            }
    }
    const int row = g * 64 + lane;
    if (row >= a.rows) return;
    if (EPI == EPI_STORE) a.out[row] = res[0];
    else if (EPI == EPI_RESID) a.out[row] = a.resid_in ? a.resid_in[row] + res[0] : res[0];
    else {
        float gte = res[0];
        gte = gte / (float)(1.0 + exp(-(double)gte));            // InferenceCore.java:155-158
        a.out[row] = gte * res[NM - 1];
    }
}

}  // namespace gl3
