// gl3_decode_kernels.h — hand-written gfx950 kernels for the single-token decode step.
//
// Replaces the TornadoVM-JITted Java kernels of
//   J/tornadovm/kernels/TransformerComputeKernelsLayered.java (fusedQKVMatmulQ8 :3038-3223,
//   matrixVectorGenericWithResidualQ8_0Byte :2888-2906, fullyFusedRmsNormFFNGateUpQ8 :3386-3549,
//   processHeadsFlashAttention :784-906, ropeRotationWithCacheCopy :495-542) and
//   J/tornadovm/kernels/TransformerComputeKernels.java (convertQ8_0toFP32 :95-126, reductionOneBlock* :149-208)
// but follows the ARITHMETIC of the pure-Java CPU path (the parity oracle), not of those GPU kernels:
//   * Q8_0 matvec = dotQ8Activation (J/tensor/standard/Q8_0FloatTensor.java:90-123): the activation is
//     quantised to int8 per 32-block (amax/127, f16-rounded scale, round-half-away) and the dot is
//     int8 x int8 -> int32 (v_dot4_i32_i8), scaled by wScale*aScale in f32;
//   * RMSNorm eps / RoPE tables come from the configuration, not from literals (SURVEY.md §7 hard parts);
//   * exp / sqrt are evaluated in double and cast, as java.lang.Math does.
// Only the ORDER of f32 reductions differs from the oracle (wave/tree reductions instead of a strictly
// sequential sum); every element-wise operation is bit-identical (compiled with -ffp-contract=off).
//
// Weight layout in HBM ("Q8R", built once at upload by repack_q8_kernel): a row of nb 32-element blocks is
// padded to a multiple of 8 blocks and cut into chunks of <= 64 blocks; a chunk of n blocks is stored as
//   [n x f16 scale][n x 16 B quants 0..15][n x 16 B quants 16..31]          (34 n bytes, as in GGUF)
// so one wavefront reads a chunk with three fully coalesced loads (2 B, 16 B, 16 B per lane; lane = block).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace gl3 {

constexpr int WG = 256;            // 4 wavefronts of 64
constexpr int WAVES = WG / 64;
constexpr int CHUNK_BLOCKS = 64;   // blocks per chunk = lanes per wave
constexpr int CHUNK_BYTES = 34 * CHUNK_BLOCKS;

enum { PRO_RMS = 0, PRO_QUANT = 1 };
enum { EPI_STORE = 0, EPI_RESID = 1, EPI_SWIGLU = 2 };

struct MatvecArgs {
    const uint8_t* w;        // Q8R rows
    const uint8_t* w2;       // second matrix (EPI_SWIGLU: w = gate W1, w2 = up W3)
    int rows;                // output rows
    int k;                   // input length (multiple of 32)
    int nbp;                 // padded blocks per row (multiple of 8)
    int rows_per_wave;       // contiguous rows owned by one wave
    const float* x;          // input vector f32[k]
    const float* norm_w;     // PRO_RMS: RMSNorm weight f32[k]
    float eps;
    float* out;              // EPI_STORE: out[row]; EPI_SWIGLU: hb[row]
    const float* resid_in;   // EPI_RESID: out[row] = resid_in[row] + acc  (resid_in may be NULL: out[row] = acc)
};

__device__ __forceinline__ float h2f(uint16_t h) { return (float)__builtin_bit_cast(_Float16, h); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v = fmaxf(v, __shfl_xor(v, m, 64));
    return v;
}

// ---------------------------------------------------------------------------------------------------
// Activation prologue: (optional RMSNorm) + Q8_0 activation quantisation into LDS.
//   InferenceCore.rmsnorm  J/inference/InferenceCore.java:39-48
//   quantisation           J/tensor/standard/Q8_0FloatTensor.java:96-118
// LDS image per chunk c: xq[c*2048 + bl*16] = quants 0..15 of block bl, xq[c*2048 + 1024 + bl*16] = 16..31;
// xs[c*64 + bl] = f16-rounded activation scale.  Every workgroup builds its own copy (deterministic).
template <int PRO>
__device__ __forceinline__ void quantize_to_lds(const float* __restrict__ x, const float* __restrict__ nw, float eps,
                                                int k, int nct, uint8_t* xq, float* xs, float* red) {
    const int t = threadIdx.x;
    const int nquads = k >> 2;
    float scale = 1.0f;
    if (PRO == PRO_RMS) {
        float ss = 0.f;
        for (int qd = t; qd < nquads; qd += WG) {
            const float4 v = *reinterpret_cast<const float4*>(x + 4 * qd);
            ss += v.x * v.x; ss += v.y * v.y; ss += v.z * v.z; ss += v.w * v.w;
        }
        ss = wave_sum(ss);
        if ((t & 63) == 0) red[t >> 6] = ss;
        __syncthreads();
        float tot = ((red[0] + red[1]) + red[2]) + red[3];
        tot /= (float)k;
        tot += eps;
        scale = (float)(1.0 / sqrt((double)tot));
    }
    // zero the padded tail (blocks k/32 .. nct*64) so ragged chunks read finite data
    for (int b = (k >> 5) + t; b < nct * CHUNK_BLOCKS; b += WG) {
        xs[b] = 0.f;
        int4 z = {0, 0, 0, 0};
        *reinterpret_cast<int4*>(xq + (b >> 6) * 2048 + (b & 63) * 16) = z;
        *reinterpret_cast<int4*>(xq + (b >> 6) * 2048 + 1024 + (b & 63) * 16) = z;
    }
    for (int qd = t; qd < nquads; qd += WG) {   // 8 consecutive threads own one 32-element block
        float4 v = *reinterpret_cast<const float4*>(x + 4 * qd);
        if (PRO == PRO_RMS) {
            const float4 w = *reinterpret_cast<const float4*>(nw + 4 * qd);
            v.x = w.x * (scale * v.x); v.y = w.y * (scale * v.y); v.z = w.z * (scale * v.z); v.w = w.w * (scale * v.w);
        }
        float amax = fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w)));
        amax = fmaxf(amax, __shfl_xor(amax, 1, 64));
        amax = fmaxf(amax, __shfl_xor(amax, 2, 64));
        amax = fmaxf(amax, __shfl_xor(amax, 4, 64));
        const float qs = amax / 127.0f;
        const float ainv = qs != 0.f ? 1.0f / qs : 0.f;
        float s0 = v.x * ainv, s1 = v.y * ainv, s2 = v.z * ainv, s3 = v.w * ainv;
        const int q0 = (int)(s0 + copysignf(0.5f, s0)), q1 = (int)(s1 + copysignf(0.5f, s1));
        const int q2 = (int)(s2 + copysignf(0.5f, s2)), q3 = (int)(s3 + copysignf(0.5f, s3));
        const uint32_t packed = (uint32_t)(q0 & 0xFF) | ((uint32_t)(q1 & 0xFF) << 8) | ((uint32_t)(q2 & 0xFF) << 16) |
                                ((uint32_t)(q3 & 0xFF) << 24);
        const int b = qd >> 3, wd = qd & 7;
        uint8_t* dst = xq + (b >> 6) * 2048 + (wd >= 4 ? 1024 : 0) + (b & 63) * 16 + (wd & 3) * 4;
        *reinterpret_cast<uint32_t*>(dst) = packed;
        if (wd == 0) xs[b] = (float)(_Float16)qs;   // Float.float16ToFloat(Float.floatToFloat16(qs))
    }
    __syncthreads();
}

__device__ __forceinline__ int dot32(const int4& a0, const int4& a1, const int4& b0, const int4& b1) {
    int s = 0;
    s = __builtin_amdgcn_sdot4(a0.x, b0.x, s, false); s = __builtin_amdgcn_sdot4(a0.y, b0.y, s, false);
    s = __builtin_amdgcn_sdot4(a0.z, b0.z, s, false); s = __builtin_amdgcn_sdot4(a0.w, b0.w, s, false);
    s = __builtin_amdgcn_sdot4(a1.x, b1.x, s, false); s = __builtin_amdgcn_sdot4(a1.y, b1.y, s, false);
    s = __builtin_amdgcn_sdot4(a1.z, b1.z, s, false); s = __builtin_amdgcn_sdot4(a1.w, b1.w, s, false);
    return s;
}

template <bool NT>
__device__ __forceinline__ int4 ld16(const uint8_t* p) {
    typedef int v4i __attribute__((ext_vector_type(4)));
    v4i v;
    if (NT) v = __builtin_nontemporal_load(reinterpret_cast<const v4i*>(p));
    else v = *reinterpret_cast<const v4i*>(p);
    int4 r; r.x = v.x; r.y = v.y; r.z = v.z; r.w = v.w;
    return r;
}
template <bool NT>
__device__ __forceinline__ uint16_t ld2(const uint8_t* p) {
    if (NT) return __builtin_nontemporal_load(reinterpret_cast<const uint16_t*>(p));
    return *reinterpret_cast<const uint16_t*>(p);
}

// ---------------------------------------------------------------------------------------------------
// Dequant-fused Q8_0 matvec.  One wavefront owns `rows_per_wave` contiguous rows and walks them G at a time
// (NM = 2 for the fused gate/up pair); lane = block within a 64-block chunk.  HBM-bound: per row-chunk a wave
// issues 3 coalesced loads (128 B + 1 KiB + 1 KiB), straight to VGPRs, non-temporal (each byte is read once).
template <int PRO, int EPI, int G, bool NT>
__global__ __launch_bounds__(WG) void matvec_q8_kernel(const MatvecArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    constexpr int NM = (EPI == EPI_SWIGLU) ? 2 : 1;
    const int nct = (a.nbp + CHUNK_BLOCKS - 1) / CHUNK_BLOCKS;
    uint8_t* xq = smem;
    float* xs = reinterpret_cast<float*>(smem + nct * 2048);
    float* red = xs + nct * CHUNK_BLOCKS;

    const int lane = threadIdx.x & 63;
    const int gw = blockIdx.x * WAVES + (threadIdx.x >> 6);
    const int row_begin = gw * a.rows_per_wave;
    const int row_end = min(a.rows, row_begin + a.rows_per_wave);
    const size_t stride = (size_t)a.nbp * 34;

    quantize_to_lds<PRO>(a.x, a.norm_w, a.eps, a.k, nct, xq, xs, red);

    for (int row0 = row_begin; row0 < row_end; row0 += G) {
        float acc[NM][G];
#pragma unroll
        for (int m = 0; m < NM; ++m)
#pragma unroll
            for (int r = 0; r < G; ++r) acc[m][r] = 0.f;

        for (int c = 0; c < nct; ++c) {
            const int n = min(CHUNK_BLOCKS, a.nbp - c * CHUNK_BLOCKS);   // wave-uniform
            if (lane < n) {
                uint16_t sc[NM][G];
                int4 lo[NM][G], hi[NM][G];
#pragma unroll
                for (int m = 0; m < NM; ++m)
#pragma unroll
                    for (int r = 0; r < G; ++r) {
                        const int row = min(row0 + r, row_end - 1);      // ragged last group re-reads a valid row
                        const uint8_t* p = (m == 0 ? a.w : a.w2) + (size_t)row * stride + (size_t)c * CHUNK_BYTES;
                        sc[m][r] = ld2<NT>(p + 2 * lane);
                        lo[m][r] = ld16<NT>(p + 2 * n + 16 * lane);
                        hi[m][r] = ld16<NT>(p + 18 * n + 16 * lane);
                    }
                const int4 xlo = *reinterpret_cast<const int4*>(xq + c * 2048 + lane * 16);
                const int4 xhi = *reinterpret_cast<const int4*>(xq + c * 2048 + 1024 + lane * 16);
                const float xsc = xs[c * CHUNK_BLOCKS + lane];
#pragma unroll
                for (int m = 0; m < NM; ++m)
#pragma unroll
                    for (int r = 0; r < G; ++r) {
                        const int isum = dot32(lo[m][r], hi[m][r], xlo, xhi);
                        acc[m][r] += (float)isum * (h2f(sc[m][r]) * xsc);   // result += isum * (wScale * aScale)
                    }
            }
        }
#pragma unroll
        for (int m = 0; m < NM; ++m)
#pragma unroll
            for (int r = 0; r < G; ++r) acc[m][r] = wave_sum(acc[m][r]);

        if (lane == 0) {
#pragma unroll
            for (int r = 0; r < G; ++r) {
                const int row = row0 + r;
                if (row < row_end) {
                    if (EPI == EPI_STORE) a.out[row] = acc[0][r];
                    if (EPI == EPI_RESID) a.out[row] = a.resid_in ? a.resid_in[row] + acc[0][r] : acc[0][r];
                    if (EPI == EPI_SWIGLU) {   // InferenceCore.java:155-158, exp in double
                        float g = acc[0][r];
                        g = g / (float)(1.0 + exp(-(double)g));
                        a.out[row] = g * acc[NM - 1][r];
                    }
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// Embedding row gather + dequant: x[i] = q * d  (token_embedding_table.copyTo, InferenceCore.java:61;
// replaces the host row copy of forwardTornadoVM :956-980 + convertQ8_0toFP32).
__global__ __launch_bounds__(WG) void embed_q8_kernel(const uint8_t* __restrict__ emb, int nbp, int dim,
                                                       const int* __restrict__ dyn, float* __restrict__ x) {
    const int token = dyn[0];
    const uint8_t* row = emb + (size_t)token * nbp * 34;
    for (int i = threadIdx.x; i < dim; i += WG) {
        const int b = i >> 5, c = b >> 6, bl = b & 63, j = i & 31;
        const int n = min(CHUNK_BLOCKS, nbp - c * CHUNK_BLOCKS);
        const uint8_t* p = row + (size_t)c * CHUNK_BYTES;
        const float d = h2f(*reinterpret_cast<const uint16_t*>(p + 2 * bl));
        const int8_t q = (int8_t)p[(j < 16 ? 2 * n : 18 * n) + 16 * bl + (j & 15)];
        x[i] = (float)q * d;
    }
}

// ---------------------------------------------------------------------------------------------------
// Decode attention, split over the sequence.  Grid = n_heads x n_split workgroups.
//   scores / softmax / weighted V sum: InferenceCore.java:98-137 (Qwen3 :631-663)
//   RoPE (adjacent pairs) :75-87, Qwen3 per-head RMSNorm + NeoX RoPE :594-619, KV write :92-93
// The qkv matvec leaves RAW q|k|v in `qkv`; this kernel rotates q (and the new k) on the fly, the split
// that owns `pos` of head h with h % kvMul == 0 writes the rotated k and v into the cache, and every
// split takes the row for t == pos from registers instead of the cache (no intra-launch read-after-write).
struct AttnArgs {
    const float* qkv;        // raw [qDim | kvDim | kvDim]
    float* kcache;           // [ctx][kvDim] of this layer
    float* vcache;
    const float* rope_cr;    // [ctx][hs/2]
    const float* rope_ci;
    const float* qnorm;      // qwen3: f32[hs] (else NULL)
    const float* knorm;
    const int* dyn;          // dyn[1] = position
    float* part;             // [H][S][hs + 2]  (m, l, o[hs])
    int n_heads, n_kv_heads, hs, q_dim, kv_dim, n_split;
    float eps;
    int arch;
};

// rotate one head vector held in LDS: v[hs]; Llama pairs (2i,2i+1), NeoX pairs (i, i+hs/2)
__device__ __forceinline__ void rope_head(float* v, int hs, const float* cr, const float* ci, int arch, int t, int nthreads) {
    const int half = hs >> 1;
    for (int i = t; i < half; i += nthreads) {
        const float fcr = cr[i], fci = ci[i];
        const int i0 = arch == 0 ? 2 * i : i, i1 = arch == 0 ? 2 * i + 1 : i + half;
        const float v0 = v[i0], v1 = v[i1];
        v[i0] = v0 * fcr - v1 * fci;
        v[i1] = v0 * fci + v1 * fcr;
    }
}

// per-head RMSNorm in LDS by one wavefront-sized group (hs <= 256): out = w * (ss * x)
__device__ __forceinline__ void head_rmsnorm(float* v, const float* w, int hs, float eps, float* red) {
    const int t = threadIdx.x;
    float ss = 0.f;
    for (int i = t; i < hs; i += WG) ss += v[i] * v[i];
    ss = wave_sum(ss);
    if ((t & 63) == 0) red[t >> 6] = ss;
    __syncthreads();
    float tot = ((red[0] + red[1]) + red[2]) + red[3];
    __syncthreads();
    tot /= (float)hs;
    tot += eps;
    const float sc = (float)(1.0 / sqrt((double)tot));
    for (int i = t; i < hs; i += WG) v[i] = w[i] * (sc * v[i]);
    __syncthreads();
}

constexpr int ATT_MAX_T = 1024;   // timesteps one split can hold in LDS

__global__ __launch_bounds__(WG) void attn_partial_kernel(const AttnArgs a) {
    __shared__ float q_s[256], k_s[256], sc_s[ATT_MAX_T], red[WAVES], o_s[WG];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int h = blockIdx.x / a.n_split, sp = blockIdx.x % a.n_split;
    const int hs = a.hs, kvmul = a.n_heads / a.n_kv_heads, kvh = h / kvmul;
    const int pos = a.dyn[1];
    const int span = (pos + 1 + a.n_split - 1) / a.n_split;
    const int t0 = sp * span, t1 = min(pos + 1, t0 + span);
    float* part = a.part + ((size_t)h * a.n_split + sp) * (hs + 2);
    if (t0 >= t1) {               // empty split
        if (t == 0) { part[0] = -INFINITY; part[1] = 0.f; }
        return;
    }
    const bool owns_pos = (t1 == pos + 1);
    for (int i = t; i < hs; i += WG) {
        q_s[i] = a.qkv[h * hs + i];
        if (owns_pos) k_s[i] = a.qkv[a.q_dim + kvh * hs + i];
    }
    __syncthreads();
    if (a.arch == 1) {
        head_rmsnorm(q_s, a.qnorm, hs, a.eps, red);
        if (owns_pos) head_rmsnorm(k_s, a.knorm, hs, a.eps, red);
    }
    const float* cr = a.rope_cr + (size_t)pos * (hs >> 1);
    const float* ci = a.rope_ci + (size_t)pos * (hs >> 1);
    rope_head(q_s, hs, cr, ci, a.arch, t, WG);
    if (owns_pos) rope_head(k_s, hs, cr, ci, a.arch, t, WG);
    __syncthreads();
    if (owns_pos && (h % kvmul) == 0) {          // KV write, InferenceCore.java:92-93
        for (int i = t; i < hs; i += WG) {
            a.kcache[(size_t)pos * a.kv_dim + kvh * hs + i] = k_s[i];
            a.vcache[(size_t)pos * a.kv_dim + kvh * hs + i] = a.qkv[a.q_dim + a.kv_dim + kvh * hs + i];
        }
    }
    // ---- scores: lpt = hs/4 lanes per timestep, float4 each
    const int lpt = hs >> 2, tpi = WG / lpt;      // timesteps per iteration
    const int sub = t % lpt, tslot = t / lpt;
    const float sqrt_hs = (float)sqrt((double)hs);
    const float4 qv = *reinterpret_cast<const float4*>(&q_s[4 * sub]);
    for (int tb = t0; tb < t1; tb += tpi) {
        const int tt = tb + tslot;
        float s = 0.f;
        if (tt < t1) {
            float4 kv;
            if (tt == pos) kv = *reinterpret_cast<const float4*>(&k_s[4 * sub]);
            else kv = *reinterpret_cast<const float4*>(a.kcache + (size_t)tt * a.kv_dim + kvh * hs + 4 * sub);
            s = qv.x * kv.x; s += qv.y * kv.y; s += qv.z * kv.z; s += qv.w * kv.w;
        }
        for (int m = lpt >> 1; m >= 1; m >>= 1) s += __shfl_xor(s, m, 64);   // lpt <= 64 lanes, aligned groups
        if (tt < t1 && sub == 0) sc_s[tt - t0] = s / sqrt_hs;
    }
    __syncthreads();
    // ---- local softmax numerators (max, exp in double, sum)
    const int nt = t1 - t0;
    float mx = -INFINITY;
    for (int i = t; i < nt; i += WG) mx = fmaxf(mx, sc_s[i]);
    mx = wave_max(mx);
    if (lane == 0) red[wave] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
    float ls = 0.f;
    for (int i = t; i < nt; i += WG) {
        const float p = (float)exp((double)(sc_s[i] - mx));
        sc_s[i] = p;
        ls += p;
    }
    ls = wave_sum(ls);
    if (lane == 0) red[wave] = ls;
    __syncthreads();
    ls = ((red[0] + red[1]) + red[2]) + red[3];
    // ---- o[j] = sum_t p_t * V[t][j]; hs lanes over j, WG/hs groups over t
    const int groups = max(1, WG / hs);
    const int j = t % hs, grp = t / hs;
    float o = 0.f;
    if (grp < groups && j < hs) {
        for (int tt = t0 + grp; tt < t1; tt += groups) {
            const float v = (tt == pos) ? a.qkv[a.q_dim + a.kv_dim + kvh * hs + j]
                                        : a.vcache[(size_t)tt * a.kv_dim + kvh * hs + j];
            o = sc_s[tt - t0] * v + o;
        }
        o_s[grp * hs + j] = o;
    }
    __syncthreads();
    if (t < hs) {
        float tot = o_s[t];
        for (int g = 1; g < groups; ++g) tot += o_s[g * hs + t];
        part[2 + t] = tot;
    }
    if (t == 0) { part[0] = mx; part[1] = ls; }
}

// Combine the splits of one head: xb[h*hs + j] = sum_s w_s o_s[j] / sum_s w_s l_s, w_s = exp(m_s - M).
__global__ __launch_bounds__(WG) void attn_combine_kernel(const float* __restrict__ part, float* __restrict__ xb,
                                                           int hs, int n_split) {
    const int h = blockIdx.x, t = threadIdx.x;
    const float* p = part + (size_t)h * n_split * (hs + 2);
    float M = -INFINITY;
    for (int s = 0; s < n_split; ++s) M = fmaxf(M, p[s * (hs + 2)]);
    float L = 0.f;
    for (int j = t; j < hs; j += WG) {
        float o = 0.f;
        L = 0.f;
        for (int s = 0; s < n_split; ++s) {
            const float m = p[s * (hs + 2)];
            if (m == -INFINITY) continue;
            const float w = (float)exp((double)(m - M));
            L += w * p[s * (hs + 2) + 1];
            o += w * p[s * (hs + 2) + 2 + j];
        }
        xb[h * hs + j] = o / L;
    }
}

// ---------------------------------------------------------------------------------------------------
// Greedy sampling on the device: first index of the maximum (strict >), FloatTensor.argmax
// J/tensor/standard/FloatTensor.java:138-151 — NOT the strided-scan tie-break of the reference's
// argmaxLogits (TransformerComputeKernels.java:25-59).
__global__ __launch_bounds__(1024) void argmax_kernel(const float* __restrict__ v, int n, int* __restrict__ out) {
    __shared__ float bv[16];
    __shared__ int bi[16];
    const int t = threadIdx.x;
    float best = -INFINITY;
    int idx = 0x7FFFFFFF;
    for (int i = t; i < n; i += 1024) {
        const float f = v[i];
        if (f > best || (f == best && i < idx)) { best = f; idx = i; }
    }
    for (int m = 32; m >= 1; m >>= 1) {
        const float ob = __shfl_xor(best, m, 64);
        const int oi = __shfl_xor(idx, m, 64);
        if (ob > best || (ob == best && oi < idx)) { best = ob; idx = oi; }
    }
    if ((t & 63) == 0) { bv[t >> 6] = best; bi[t >> 6] = idx; }
    __syncthreads();
    if (t == 0) {
        for (int w = 1; w < 16; ++w)
            if (bv[w] > best || (bv[w] == best && bi[w] < idx)) { best = bv[w]; idx = bi[w]; }
        *out = idx == 0x7FFFFFFF ? 0 : idx;
    }
}

// ---------------------------------------------------------------------------------------------------
// One-time layout transform at upload: GGUF Q8_0 blocks (34 B: f16 d + 32 x int8, GGMLType.java:13) of the
// row range [r0, r0+rows) and block range [b0, b0+nb) of a [*, nb_full*32] matrix -> Q8R chunks.
__global__ void repack_q8_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, int rows, int nb, int nbp,
                                 long r0, int b0, int nb_full) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)rows * nbp) return;
    const int row = (int)(idx / nbp), pb = (int)(idx % nbp);
    const int c = pb >> 6, bl = pb & 63;
    const int n = min(CHUNK_BLOCKS, nbp - c * CHUNK_BLOCKS);
    uint8_t* base = dst + (size_t)row * nbp * 34 + (size_t)c * CHUNK_BYTES;
    uint16_t h[17];
    if (pb < nb) {
        const uint16_t* s = reinterpret_cast<const uint16_t*>(src + ((size_t)(r0 + row) * nb_full + b0 + pb) * 34);
#pragma unroll
        for (int i = 0; i < 17; ++i) h[i] = s[i];
    } else {
#pragma unroll
        for (int i = 0; i < 17; ++i) h[i] = 0;
    }
    *reinterpret_cast<uint16_t*>(base + 2 * bl) = h[0];
    uint16_t* lo = reinterpret_cast<uint16_t*>(base + 2 * n + 16 * bl);
    uint16_t* hi = reinterpret_cast<uint16_t*>(base + 18 * n + 16 * bl);
#pragma unroll
    for (int i = 0; i < 8; ++i) { lo[i] = h[1 + i]; hi[i] = h[9 + i]; }
}

}  // namespace gl3
