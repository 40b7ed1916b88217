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
// and is read as wavefront-uniform float4 broadcasts.  The products are computed in parallel by producer wavefronts;
// the kernels stay bound by the K dependent adds of the chain itself (about 5-6 cycles per element per row).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "gl3_decode_kernels.h"

namespace gl3 {

enum { WT_F16 = 1, WT_Q4_0 = 2, WT_Q8_0 = 3 };      // WT_Q8_0: Q8_0 with f32 activation (veclane kernels only)

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
                                                              float* __restrict__ x, float emb_scale) {
    const int token = dyn[0], g = token >> 6, lane = token & 63;
    const uint8_t* gb = emb + (size_t)g * rl_group_bytes(WT, dim);
    for (int i = threadIdx.x; i < dim; i += 256) {
        if (WT == WT_F16) {
            x[i] = h2f(*reinterpret_cast<const uint16_t*>(gb + (size_t)(i >> 3) * 1024 + lane * 16 + 2 * (i & 7))) * emb_scale;
        } else {
            const uint8_t* b = gb + (size_t)(i >> 5) * 1152;
            const int j = i & 31;
            const uint8_t byte = b[128 + lane * 16 + (j & 15)];
            const int q = j < 16 ? (byte & 0x0F) : (byte >> 4);
            x[i] = ((float)(q - 8) * h2f(*reinterpret_cast<const uint16_t*>(b + 2 * lane))) * emb_scale;
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
    float out_scale;                        // result *= out_scale first (Granite residual / logit scaling; 1 otherwise)
};

// Workgroup = one 64-row group, 9 wavefronts in two roles (as in matvec_q8t_kernel, register pressure = max of the roles):
//   producers (waves 1-8): lane = (row, 8 consecutive elements of the current 64-element tile): one 16-byte (F16) or
//       8-byte (Q4_0) weight load, kept RL_D tiles ahead in registers, p = w * x (Q4_0: ((q - 8) * d) * x) written to a
//       double-buffered LDS tile P[row][64] (pitch 68: conflict-free 16-byte rows);
//   chain (wave 0, raised priority): lane = row, result += p in element order — the only serial part of the reference's dot product —
//       and the epilogue.
// One workgroup barrier per tile: producers fill tile i+1 while the chain consumes tile i.
constexpr int RL_T = 64, RL_PITCH = 68, RL_NP = 8, RL_THREADS = 64 * (RL_NP + 1), RL_D = 8;

__host__ __device__ inline size_t rl_smem_bytes(int k, int epi) {
    return ((size_t)k + (size_t)2 * (epi == EPI_SWIGLU ? 2 : 1) * 64 * RL_PITCH) * 4;
}

template <int WT, int EPI>
static __global__ __launch_bounds__(RL_THREADS) void matvec_rl_kernel(const RlArgs a) {
    extern __shared__ __attribute__((aligned(16))) float xs[];   // [k] | P[2][NM][64][RL_PITCH]
    constexpr int NM = EPI == EPI_SWIGLU ? 2 : 1;
    typedef int v4i __attribute__((ext_vector_type(4)));
    typedef int v2i __attribute__((ext_vector_type(2)));
    const int t = threadIdx.x, lane = t & 63;
    // wavefront 0 (the oldest: wins issue arbitration on its SIMD) runs the chain; producers are role-waves 0..7
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6) - 1;
#ifdef GL3_RL_TIMING
    const unsigned long long tstart = __builtin_readcyclecounter();
#endif
    float* P = xs + a.k;
    for (int i = t; i < (a.k >> 2); i += RL_THREADS) *reinterpret_cast<float4*>(xs + 4 * i) = *reinterpret_cast<const float4*>(a.x + 4 * i);
    __syncthreads();
    const int g = blockIdx.x, ntiles = a.k / RL_T;
    const size_t gbytes = rl_group_bytes(WT, a.k);

    if (wave >= 0) {
        // ------------------------------------------------------------------ producers
        const uint8_t* wp[NM];
        wp[0] = a.w + (size_t)g * gbytes;
        if (NM == 2) wp[NM - 1] = a.w2 + (size_t)g * gbytes;
        // Q4_0 unit: block-in-tile bt, nibble half h, byte quarter q8 -> elements 32*bt + 16*h + 8*q8 + i
        const int bt = wave >> 2, h = (wave >> 1) & 1, q8 = wave & 1;
        v4i raw16[NM][RL_D];      // F16: 8 halfs
        v2i raw8[NM][RL_D];       // Q4_0: 8 bytes of nibbles
        uint16_t dsc[NM][RL_D];   // Q4_0: block scale
        auto load = [&](int m, int u, int tile) {
            if (WT == WT_F16) {
                raw16[m][u] = __builtin_nontemporal_load(reinterpret_cast<const v4i*>(wp[m] + (size_t)(tile * 8 + wave) * 1024 + lane * 16));
            } else {
                const uint8_t* pb = wp[m] + (size_t)(tile * 2 + bt) * 1152;
                raw8[m][u] = *reinterpret_cast<const v2i*>(pb + 128 + lane * 16 + 8 * q8);
                dsc[m][u] = *reinterpret_cast<const uint16_t*>(pb + 2 * lane);
            }
        };
#pragma unroll
        for (int u = 0; u < RL_D; ++u)
            if (u < ntiles)
#pragma unroll
                for (int m = 0; m < NM; ++m) load(m, u, u);
        for (int base = 0; base < ntiles; base += RL_D) {
#pragma unroll
            for (int u = 0; u < RL_D; ++u) {
                const int tile = base + u;
                if (tile < ntiles) {
                    const int e0 = WT == WT_F16 ? wave * 8 : 32 * bt + 16 * h + 8 * q8;     // first element within the tile
                    const float4 x0 = *reinterpret_cast<const float4*>(xs + tile * RL_T + e0);
                    const float4 x1 = *reinterpret_cast<const float4*>(xs + tile * RL_T + e0 + 4);
                    const float xv[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
#pragma unroll
                    for (int m = 0; m < NM; ++m) {
                        float pv[8];
                        if (WT == WT_F16) {
#pragma unroll
                            for (int i = 0; i < 8; ++i) {
                                const uint32_t word = (uint32_t)raw16[m][u][i >> 1];
                                pv[i] = h2f((uint16_t)((i & 1) ? word >> 16 : word & 0xFFFF)) * xv[i];
                            }
                        } else {
                            const float d = h2f(dsc[m][u]);
#pragma unroll
                            for (int i = 0; i < 8; ++i) {
                                const uint32_t word = (uint32_t)raw8[m][u][i >> 2];
                                const int q = (int)((word >> (8 * (i & 3) + 4 * h)) & 0xF);
                                pv[i] = ((float)(q - 8) * d) * xv[i];
                            }
                        }
                        float* dst = P + ((size_t)((tile & 1) * NM + m) * 64 + lane) * RL_PITCH + e0;
                        *reinterpret_cast<float4*>(dst) = make_float4(pv[0], pv[1], pv[2], pv[3]);
                        *reinterpret_cast<float4*>(dst + 4) = make_float4(pv[4], pv[5], pv[6], pv[7]);
                        if (tile + RL_D < ntiles) load(m, u, tile + RL_D);
                    }
                    __syncthreads();
                }
            }
        }
        __syncthreads();
        return;
    }
    // ---------------------------------------------------------------------- chain wavefront (lane = row)
    __builtin_amdgcn_s_setprio(3);
    float res[NM];
#pragma unroll
    for (int m = 0; m < NM; ++m) res[m] = 0.f;
#ifdef GL3_RL_TIMING
    unsigned long long ts[4] = {0, 0, 0, 0};
#endif
    for (int tile = 0; tile < ntiles; ++tile) {
#ifdef GL3_RL_TIMING
        if (tile == 8) ts[0] = __builtin_readcyclecounter();
#endif
        __syncthreads();
#ifdef GL3_RL_TIMING
        if (tile == 8) ts[1] = __builtin_readcyclecounter();
        if (tile == 0) ts[3] = __builtin_readcyclecounter();
#endif
        float4 pv[NM][RL_T / 4];
#pragma unroll
        for (int m = 0; m < NM; ++m)
#pragma unroll
            for (int c = 0; c < RL_T / 4; ++c)
                pv[m][c] = *reinterpret_cast<const float4*>(P + ((size_t)((tile & 1) * NM + m) * 64 + lane) * RL_PITCH + 4 * c);
        // a dependent f32 add issues every ~8 cycles: the two chains of the SwiGLU pair alternate instruction by instruction
#pragma unroll
        for (int c = 0; c < RL_T / 4; ++c) {
#pragma unroll
            for (int m = 0; m < NM; ++m) res[m] = res[m] + pv[m][c].x;
#pragma unroll
            for (int m = 0; m < NM; ++m) res[m] = res[m] + pv[m][c].y;
#pragma unroll
            for (int m = 0; m < NM; ++m) res[m] = res[m] + pv[m][c].z;
#pragma unroll
            for (int m = 0; m < NM; ++m) res[m] = res[m] + pv[m][c].w;
        }
#ifdef GL3_RL_TIMING
        if (tile == 8) { asm volatile("" :: "v"(res[0])); ts[2] = __builtin_readcyclecounter(); }
        if (tile == 9 && lane == 0 && blockIdx.x == 0) printf("rl chain k=%d NM=%d: barrier wait %llu, reads+adds %llu, tile period %llu\n", a.k, NM, ts[1] - ts[0], ts[2] - ts[1], __builtin_readcyclecounter() - ts[2] + ts[2] - ts[0]);
#endif
    }
    __syncthreads();
#ifdef GL3_RL_TIMING
    if (lane == 0 && blockIdx.x == 0) printf("rl total k=%d NM=%d: first tile ready at %llu, tile8 start %llu, end %llu\n", a.k, NM, ts[3] - tstart, ts[0] - tstart, __builtin_readcyclecounter() - tstart);
#endif
    const int row = g * 64 + lane;
    if (row >= a.rows) return;
    if (EPI == EPI_STORE) a.out[row] = res[0] * a.out_scale;
    else if (EPI == EPI_RESID) a.out[row] = a.resid_in ? a.resid_in[row] + res[0] * a.out_scale : res[0] * a.out_scale;
    else {
        float gte = res[0];
        gte = gte / (float)(1.0 + exp(-(double)gte));            // InferenceCore.java:155-158
        a.out[row] = gte * res[NM - 1];
    }
}

}  // namespace gl3
