// gl3_veclane_kernels.h — decode matvec for F16 and Q4_0 weights in the order of the reference's VECTOR-API dot products
// (the reference's default for these types: FloatTensor.USE_VECTOR_API, J/tensor/standard/FloatTensor.java:21-22) with a
// 256-bit species = 8 float lanes:
//   F16  : FP16FloatTensor.vectorDot (J/tensor/standard/FP16FloatTensor.java:63-110)
//              val[l] = fma(w[8 i + l], x[8 i + l], val[l]),  i ascending;  w via the DAZ bit trick (subnormals -> +-0)
//   Q4_0 : Q4_0FloatTensor.vectorDot, 256-bit branch (J/tensor/standard/Q4_0FloatTensor.java:82-133), per 32-element block j:
//              s[l]   = ((x[j+l] * lo[l] + x[j+8+l] * lo[8+l]) + x[j+16+l] * hi[l]) + x[j+24+l] * hi[8+l]
//              val[l] = fma(s[l], wScale, val[l])        lo / hi = (low / high nibbles of the 16 quant bytes) - 8 as floats
//   Q8_0 with f32 activation (GL3_FLAG_F32_ACTIVATION = -Dllama.quantizeActivation=false): Q8_0FloatTensor.vectorDot, 256-bit
//          branch (J/tensor/standard/Q8_0FloatTensor.java:125-175), per 32-element block j with int8 quants q:
//              s[l]   = ((x[j+l] * q[l] + x[j+8+l] * q[8+l]) + x[j+16+l] * q[16+l]) + x[j+24+l] * q[24+l]
//              val[l] = fma(s[l], wScale, val[l])
//   result = reduceLanes(ADD) = ((((0 + val[0]) + val[1]) + ...) + val[7])     (lane order from 0: HotSpot evaluates float add reductions strictly in order on x86)
// Eight independent accumulator chains per row make the row's K-long sum 8-way parallel, and that is also the natural GPU
// mapping: lane = (row, accumulator lane), a wavefront = 8 rows x 8 accumulators.  Every lane runs a K/8-long chain of
// FMAs, all lanes of the chip in parallel -> the kernel is bound by the weight stream (HBM), not by a serial chain as the
// scalar-order kernels in gl3_rowlane_kernels.h are (those stay available behind GL3_FLAG_SCALAR_DOT).
//
// Weight layout in HBM ("VL", built once at upload, same byte count as GGUF): rows in groups of 8; per group and chunk a
// wavefront's loads are contiguous and each lane's 16 bytes are exactly what its chain consumes next:
//   F16  chunk = 64 elements : [lane (r, l)][8 halfs w[r][64 c + 8 k + l], k = 0..7]                       1024 B
//   Q8_0 chunk = 4 blocks    : [lane (r, l)][4 x (q[l], q[8+l], q[16+l], q[24+l]) of blocks 4c .. 4c+3] then [row r][4 x f16 d]   1024 + 64 B
//   Q4_0 chunk = 8 blocks    : [lane (r, l)][A_0..A_7 | B_0..B_7]  (A_k = quant byte l of block 8c+k: nibbles of elements
//                              l and 16+l; B_k = byte 8+l: elements 8+l and 24+l)  then [row r][8 x f16 d]   1024 + 128 B
// The activation is staged in LDS transposed the same way (F16: xT[c][l][k]; Q4_0: xT[block][l][4]) so a lane reads its
// operands with one or two conflict-free ds_read_b128.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "gl3_rowlane_kernels.h"

namespace gl3 {

__host__ __device__ inline size_t vl_group_bytes(int wt, int k) {       // bytes of one 8-row group
    return wt == WT_F16 ? (size_t)(k / 64) * 1024 : wt == WT_Q4_0 ? (size_t)(k / 256) * 1152 : (size_t)(k / 128) * 1088;
}
__host__ __device__ inline int vl_chunk_elems(int wt) { return wt == WT_F16 ? 64 : wt == WT_Q4_0 ? 256 : 128; }
__host__ __device__ inline int vl_chunk_bytes(int wt) { return wt == WT_F16 ? 1024 : wt == WT_Q4_0 ? 1152 : 1088; }

// GGUF row-major -> VL.  One thread per destination lane slot (row, chunk, l).
template <int WT>
static __global__ __launch_bounds__(256) void repack_vl_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, int rows, int k,
                                                               int dst_row0) {
    const int nch = k / vl_chunk_elems(WT);
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long)rows * nch * 8) return;
    const int l = (int)(i & 7), c = (int)((i >> 3) % nch), r = (int)((i >> 3) / nch);
    const int row = dst_row0 + r, g = row >> 3, rr = row & 7;
    uint8_t* gb = dst + (size_t)g * vl_group_bytes(WT, k);
    if (WT == WT_F16) {
        const uint16_t* s = reinterpret_cast<const uint16_t*>(src + (size_t)r * k * 2) + 64 * c + l;
        uint16_t* d = reinterpret_cast<uint16_t*>(gb + (size_t)c * 1024 + (rr * 8 + l) * 16);
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) d[kk] = s[8 * kk];
    } else if (WT == WT_Q8_0) {
        const uint8_t* s = src + ((size_t)r * (k / 32) + 4 * c) * 34;        // 4 consecutive blocks of this row
        uint8_t* d = gb + (size_t)c * 1088 + (rr * 8 + l) * 16;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
#pragma unroll
            for (int q = 0; q < 4; ++q) d[4 * kk + q] = s[kk * 34 + 2 + 8 * q + l];
        if (l < 4) reinterpret_cast<uint16_t*>(gb + (size_t)c * 1088 + 1024 + rr * 8)[l] = *reinterpret_cast<const uint16_t*>(s + l * 34);
    } else {
        const uint8_t* s = src + ((size_t)r * (k / 32) + 8 * c) * 18;        // 8 consecutive blocks of this row
        uint8_t* d = gb + (size_t)c * 1152 + (rr * 8 + l) * 16;
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) { d[kk] = s[kk * 18 + 2 + l]; d[8 + kk] = s[kk * 18 + 2 + 8 + l]; }
        // the row's 8 block scales of this chunk: written by its l-th lane (one f16 each)
        reinterpret_cast<uint16_t*>(gb + (size_t)c * 1152 + 1024 + rr * 16)[l] = *reinterpret_cast<const uint16_t*>(s + l * 18);
    }
}

// token_embedding_table.copyTo -> getFloat per element (InferenceCore.java:61): SCALAR semantics (IEEE f16 -> f32)
template <int WT>
static __global__ __launch_bounds__(256) void embed_vl_kernel(const uint8_t* __restrict__ emb, int dim, const int* __restrict__ dyn,
                                                              float* __restrict__ x, float emb_scale, uint32_t* step = nullptr) {
    if (step) {                              // folded gathers: the kernel opens decode step *step + 1 of the plan (as embed_q8t_kernel; the wait for the
        const uint32_t s0 = *step;           // previous step's last pushes into x is the wait launch behind the last down projection)
        __syncthreads();
        if (threadIdx.x == 0) *step = s0 + 1;
    }
    const int token = dyn[0], g = token >> 3, rr = token & 7;
    const uint8_t* gb = emb + (size_t)g * vl_group_bytes(WT, dim);
    for (int i = threadIdx.x; i < dim; i += 256) {
        if (WT == WT_F16) {
            const int c = i >> 6, e = i & 63, l = e & 7, kk = e >> 3;
            x[i] = h2f(reinterpret_cast<const uint16_t*>(gb + (size_t)c * 1024 + (rr * 8 + l) * 16)[kk]) * emb_scale;
        } else if (WT == WT_Q8_0) {                      // getFloat: quant * scale (Q8_0FloatTensor.java:55-63)
            const int b = i >> 5, j = i & 31, c = b >> 2, kk = b & 3, l = j & 7;
            const uint8_t* cb = gb + (size_t)c * 1088;
            const int q = (int8_t)cb[(rr * 8 + l) * 16 + 4 * kk + (j >> 3)];
            x[i] = ((float)q * h2f(reinterpret_cast<const uint16_t*>(cb + 1024 + rr * 8)[kk])) * emb_scale;
        } else {
            const int b = i >> 5, j = i & 31, c = b >> 3, kk = b & 7, l = j & 7;
            const uint8_t* cb = gb + (size_t)c * 1152;
            const uint8_t byte = cb[(rr * 8 + l) * 16 + ((j & 8) ? 8 : 0) + kk];       // j in [8,16) or [24,32): byte 8 + l
            const int q = j < 16 ? (byte & 0x0F) : (byte >> 4);
            x[i] = ((float)(q - 8) * h2f(reinterpret_cast<const uint16_t*>(cb + 1024 + rr * 16)[kk])) * emb_scale;
        }
    }
}

struct VlArgs {
    const uint8_t* w; const uint8_t* w2;    // VL matrices (w2: the "up" matrix of the SwiGLU pair)
    int rows, k;
    const float* x;                         // f32[k] activation (normalised where the reference normalises)
    float* out; const float* resid_in;      // EPI_RESID: out[i] = resid_in[i] + result
    float out_scale;                        // result *= out_scale first (Granite residual / logit scaling; 1 otherwise)
    const float* norm_w; float eps;         // RMS variant: x is the raw residual stream, normalised in the kernel's prologue
    const TpRec* tp;                        // tensor parallel, folded gathers (r6): results also go to the peers' arenas, the last wavefront publishes (NULL: no)
};

constexpr int VL_WAVES = 2;                 // 16 rows per workgroup: 4096-row matrices still give one workgroup per CU
constexpr int VL_RMS_WAVES = 4;             // RMS variant: the exact sum of squares runs on 256 threads
__host__ __device__ inline size_t vl_smem_bytes(int k) { return (size_t)k * 4; }
// RMS variant: xT[k] | xf[k + 32] (natural order, zero padded) | exact-sum scratch | red
__host__ __device__ inline size_t vl_rms_smem_bytes(int k) { return (size_t)k * 4 + (size_t)(k + 32) * 4 + ss_scratch_bytes(k) + 16; }
__host__ __device__ inline bool vl_rms_fusable(int k) { return k >= 1024 && k <= 5120 && (k & 3) == 0; }    // exact_sumsq_lds range, LDS <= 64 KB

// F16 -> f32 with subnormal inputs flushed to signed zero = the reference's bit trick (FP16FloatTensor.java:72-100, "emulate
// DAZ"): the wavefront runs its main loop with MODE.FP_DENORM[3:2] (the f16 / f64 field) = 0 (flush), so one v_cvt_f32_f16
// per weight is the whole conversion.  hwreg(HW_REG_MODE = 1, offset 6, size 2).
__device__ __forceinline__ void set_f16_denorm_flush(bool flush) {
    if (flush) __builtin_amdgcn_s_setreg(1 | (6 << 6) | (1 << 11), 0);
    else __builtin_amdgcn_s_setreg(1 | (6 << 6) | (1 << 11), 3);
}
__device__ __forceinline__ float cvt_lo(uint32_t w) { return (float)__builtin_bit_cast(_Float16, (uint16_t)(w & 0xFFFFu)); }
__device__ __forceinline__ float cvt_hi(uint32_t w) { return (float)__builtin_bit_cast(_Float16, (uint16_t)(w >> 16)); }

// Q4_0 nibbles -> (nibble - 8) without integer arithmetic or conversions: (w & 0x000F000F) | 0x64006400 is the f16 pair
// (1024 + n_lo, 1024 + n_hi) (f16 counts in units of 1 from 1024 to 2047), one v_pk_add_f16 of -1032 makes it the exact pair
// (n - 8), and v_fma_mix_f32 takes an f16 half as a multiplicand: fma(x, (float)h, -0) = fl(x * (n - 8)), the same single rounding
// as the reference's float multiply (the -0 addend keeps the sign of a zero product).  Per dword of four quant bytes: 3 shifts,
// 4 and-or, 4 packed adds for 8 values (the integer path took and / shift, add, convert = 3 per value).
typedef _Float16 h2_t __attribute__((ext_vector_type(2)));
struct Q4Consts { uint32_t magic; float negzero; };      // VGPR-resident operands (v_and_or_b32 takes one literal only)
__device__ __forceinline__ Q4Consts q4_consts() {
    Q4Consts c{0x64006400u, -0.0f};
    asm volatile("" : "+v"(c.magic), "+v"(c.negzero));
    return c;
}
__device__ __forceinline__ uint32_t q4_plane(uint32_t w_shifted, uint32_t magic) {
    uint32_t bits;
    asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(bits) : "v"(w_shifted), "s"(0x000F000Fu), "v"(magic));    // VOP3 on gfx9: no literal, one SGPR
    const h2_t v = __builtin_bit_cast(h2_t, bits) + h2_t{(_Float16)-1032.0f, (_Float16)-1032.0f};
    return __builtin_bit_cast(uint32_t, v);
}
__device__ __forceinline__ float mul_f16lo(float x, uint32_t hpair, float negzero) {
    float r;
    asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel_hi:[0,1,0]" : "=v"(r) : "v"(x), "v"(hpair), "v"(negzero));
    return r;
}
__device__ __forceinline__ float mul_f16hi(float x, uint32_t hpair, float negzero) {
    float r;
    asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[0,1,0]" : "=v"(r) : "v"(x), "v"(hpair), "v"(negzero));
    return r;
}
// the four planes of a dword of quant bytes b0..b3: [0] = low nibbles of (b0, b2), [1] = high nibbles of (b0, b2),
// [2] = low nibbles of (b1, b3), [3] = high nibbles of (b1, b3); byte j's low / high nibble = plane[2 (j & 1) + 0 / 1], half j >> 1
struct Q4Planes { uint32_t p[4]; };
__device__ __forceinline__ Q4Planes q4_planes(uint32_t w, uint32_t magic) {
    Q4Planes r;
    r.p[0] = q4_plane(w, magic); r.p[1] = q4_plane(w >> 4, magic); r.p[2] = q4_plane(w >> 8, magic); r.p[3] = q4_plane(w >> 12, magic);
    return r;
}
template <int J, int HI>      // (nibble - 8) of byte J (0..3), low (HI = 0) or high nibble, times x
__device__ __forceinline__ float q4_mul(float x, const Q4Planes& pl, float negzero) {
    const uint32_t hp = pl.p[2 * (J & 1) + HI];
    return (J >> 1) ? mul_f16hi(x, hp, negzero) : mul_f16lo(x, hp, negzero);
}

// Q4_0 is VALU-heavy (~5 instructions per weight: nibble extract, -8, convert, multiply, 3/4 add, 1/4 fma).  Two wavefronts
// per SIMD (<= 256 VGPRs); capping the registers at 128 for four made the compiler spill 119 VGPRs and was 30 % slower.
// RMS = true (qkv, gate / up, logits): InferenceCore.rmsnorm (:39-48) runs in the prologue — every workgroup computes the exact
// in-order sum of squares of the residual stream itself (gl3_seqsum.h, 256 threads) while its first weight chunks are in flight,
// instead of a one-workgroup rmsnorm_f32_kernel launch in front of the matvec (9.3 us + a launch boundary, twice per layer).
// SPECIES = 512 (F16 only, GL3_FLAG_VECTOR_512): FP16FloatTensor.vectorDot is species-generic — on an AVX-512 host it keeps 16 accumulator
// lanes, lane j = elements j, j + 16, ....  The VL layout gives lane l of a row the elements l + 8 kk of every 64-element chunk: even kk
// are accumulator lane l of the 16, odd kk lane l + 8, each in ascending element order — so the same loads feed TWO chains per thread,
// and reduceLanes adds the eight even chains in lane order, then the eight odd ones.
template <int WT, int EPI, bool RMS = false, int VW = VL_WAVES, int SPECIES = 256>
static __global__ __launch_bounds__(64 * VW, WT == WT_F16 ? 1 : 2) void matvec_vl_kernel(const VlArgs a) {
    extern __shared__ __attribute__((aligned(16))) float xT[];
    static_assert(!RMS || VW == 4, "exact_sumsq_lds: 256 threads");
    static_assert(SPECIES == 256 || (SPECIES == 512 && WT == WT_F16), "the Q8_0 / Q4_0 vector dots have no 512-bit form (the reference throws)");
    constexpr int VL_WAVES = VW;
    constexpr int NM = EPI == EPI_SWIGLU ? 2 : 1;
    // Chunks in flight per wavefront (4 or 8 VGPRs each; 16 KB / 9 KB of the weight stream per wavefront and matrix)
    constexpr int D = WT == WT_F16 ? (NM == 1 ? 16 : 8) : WT == WT_Q8_0 ? (NM == 1 ? 12 : 6) : (NM == 1 ? 8 : 4);
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, l = lane & 7, rr = lane >> 3;
    const int g = blockIdx.x * VL_WAVES + wave;              // 8-row group of this wavefront
    const int ngroups = (a.rows + 7) >> 3;
    const int nch = a.k / vl_chunk_elems(WT);
    const size_t gbytes = vl_group_bytes(WT, a.k);
    const bool live = g < ngroups;
    const uint8_t* wb[NM];
    wb[0] = a.w + (size_t)(live ? g : 0) * gbytes;
    if (NM == 2) wb[NM - 1] = a.w2 + (size_t)(live ? g : 0) * gbytes;

    // ---- weights first: D chunks per matrix are in flight before the activation is even staged
    int4 wq[NM][D];
    int4 wsc[NM][WT == WT_Q4_0 ? D : 1];
    uint2 wsc8[NM][WT == WT_Q8_0 ? D : 1];              // Q8_0: the row's 4 block scales of a chunk
    auto issue = [&](int u, int c) {
#pragma unroll
        for (int m = 0; m < NM; ++m) {
            const uint8_t* cb = wb[m] + (size_t)c * vl_chunk_bytes(WT);
            wq[m][u] = ld16<true>(cb + 16 * lane);
            if (WT == WT_Q4_0) wsc[m][u] = ld16<true>(cb + 1024 + 16 * rr);
            if (WT == WT_Q8_0) wsc8[m][u] = *reinterpret_cast<const uint2*>(cb + 1024 + 8 * rr);
        }
    };
#pragma unroll
    for (int u = 0; u < D; ++u) {
#pragma unroll
        for (int m = 0; m < NM; ++m) { wq[m][u] = make_int4(0, 0, 0, 0); if (WT == WT_Q4_0) wsc[m][u] = make_int4(0, 0, 0, 0); if (WT == WT_Q8_0) wsc8[m][u] = make_uint2(0u, 0u); }
        if (live && u < nch) issue(u, u);
    }
    // ---- activation -> LDS, transposed: F16 xT[64 c + 8 l + k] = x[64 c + 8 k + l]; Q4_0 / Q8_0 xT[32 b + 4 l + q] = x[32 b + 8 q + l]
    if constexpr (RMS) {
        constexpr int NT = 64 * VL_WAVES, RQ = 5;               // k <= 5120: at most 5 quads per thread
        float* xf = xT + a.k;                                    // [k + 32] natural order
        uint8_t* scratch = reinterpret_cast<uint8_t*>(xf + a.k + 32);
        const int nq = a.k >> 2;
        float4 xr[RQ], nw[RQ];
#pragma unroll
        for (int u = 0; u < RQ; ++u) xr[u] = *reinterpret_cast<const float4*>(a.x + 4 * min(u * NT + t, nq - 1));
#pragma unroll
        for (int u = 0; u < RQ; ++u) nw[u] = *reinterpret_cast<const float4*>(a.norm_w + 4 * min(u * NT + t, nq - 1));
#pragma unroll
        for (int u = 0; u < RQ; ++u) if (u * NT + t < nq) *reinterpret_cast<float4*>(xf + 4 * (u * NT + t)) = xr[u];
        if (t < 32) xf[a.k + t] = 0.f;
        __syncthreads();
        BlockBarrier bb;
        float ss = exact_sumsq_lds(xf, a.k, scratch, t, bb);
        ss /= (float)a.k;
        ss += a.eps;
        const float scale = (float)(1.0 / sqrt((double)ss));
#pragma unroll
        for (int u = 0; u < RQ; ++u) {
            const int qd = u * NT + t;
            if (qd < nq) {
                const float vv[4] = {nw[u].x * (scale * xr[u].x), nw[u].y * (scale * xr[u].y), nw[u].z * (scale * xr[u].z), nw[u].w * (scale * xr[u].w)};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int i = 4 * qd + e;
                    if (WT == WT_F16) xT[(i & ~63) + 8 * (i & 7) + ((i >> 3) & 7)] = vv[e];
                    else xT[(i & ~31) + 4 * (i & 7) + ((i >> 3) & 3)] = vv[e];
                }
            }
        }
    } else {
        // 16 float4 per thread are requested at once (one L2 round trip per 32 KB of activation), then scattered
        constexpr int XB = 16, NT = 64 * VL_WAVES;
        const int nq = a.k >> 2;
        for (int base = 0; base < nq; base += XB * NT) {
            float4 xr[XB];
#pragma unroll
            for (int u = 0; u < XB; ++u) {
                xr[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (base + u * NT < nq) xr[u] = *reinterpret_cast<const float4*>(a.x + 4 * min(base + u * NT + t, nq - 1));
            }
#pragma unroll
            for (int u = 0; u < XB; ++u) {
                const int qd = base + u * NT + t;
                if (qd < nq) {
                    const float vv[4] = {xr[u].x, xr[u].y, xr[u].z, xr[u].w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int i = 4 * qd + e;
                        if (WT == WT_F16) xT[(i & ~63) + 8 * (i & 7) + ((i >> 3) & 7)] = vv[e];
                        else xT[(i & ~31) + 4 * (i & 7) + ((i >> 3) & 3)] = vv[e];
                    }
                }
            }
        }
    }
    __syncthreads();
    if (!live) return;

    float acc[NM], acc2[NM];                          // acc2: accumulator lanes 8..15 of a 512-bit species
#pragma unroll
    for (int m = 0; m < NM; ++m) { acc[m] = 0.f; acc2[m] = 0.f; }
    const Q4Consts qc = q4_consts();
    // One chunk of the 8 accumulator chains.  The chunk's activation operands (xa) were fetched from LDS while the previous
    // chunk was computed; this call fetches the next chunk's (xb).
    constexpr int XV = 2;                             // F16: float4 operands per chunk and lane (Q4_0 reads its operands per block)
    auto xload = [&](float4 (&xv)[XV], int c) {
        if (WT != WT_F16) return;
#pragma unroll
        for (int q = 0; q < XV; ++q) xv[q] = *reinterpret_cast<const float4*>(xT + 64 * c + 8 * l + 4 * q);
    };
    auto consume = [&](int u, const float4 (&xv)[XV], int cq = 0) {      // cq: chunk index (Q4_0)
        if (WT == WT_F16) {
            const float xs[8] = {xv[0].x, xv[0].y, xv[0].z, xv[0].w, xv[1].x, xv[1].y, xv[1].z, xv[1].w};
#pragma unroll
            for (int m = 0; m < NM; ++m) {
                const uint32_t wd[4] = {(uint32_t)wq[m][u].x, (uint32_t)wq[m][u].y, (uint32_t)wq[m][u].z, (uint32_t)wq[m][u].w};
#pragma unroll
                for (int kk = 0; kk < 8; ++kk) {     // thizVector.fma(thatVector, val); the conversion flushes subnormals (MODE)
                    const float wv = (kk & 1) ? cvt_hi(wd[kk >> 1]) : cvt_lo(wd[kk >> 1]);
                    if (SPECIES == 512 && (kk & 1)) acc2[m] = __builtin_fmaf(wv, xs[kk], acc2[m]);
                    else acc[m] = __builtin_fmaf(wv, xs[kk], acc[m]);
                }
            }
        } else if (WT == WT_Q8_0) {
#pragma unroll
            for (int m = 0; m < NM; ++m) {
                const uint32_t qw[4] = {(uint32_t)wq[m][u].x, (uint32_t)wq[m][u].y, (uint32_t)wq[m][u].z, (uint32_t)wq[m][u].w};   // one block each
                const uint32_t sw[2] = {wsc8[m][u].x, wsc8[m][u].y};
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    const float q0 = (float)(int8_t)(qw[kk] & 0xFFu), q1 = (float)(int8_t)((qw[kk] >> 8) & 0xFFu);      // castShape(F_SPECIES, i): exact
                    const float q2 = (float)(int8_t)((qw[kk] >> 16) & 0xFFu), q3 = (float)(int8_t)(qw[kk] >> 24);
                    const float ws = h2f((uint16_t)((kk & 1) ? (sw[kk >> 1] >> 16) : (sw[kk >> 1] & 0xFFFFu)));          // Float.float16ToFloat: IEEE, subnormals kept
                    const float4 xk = *reinterpret_cast<const float4*>(xT + 32 * (4 * cq + kk) + 4 * l);   // x[j+l], x[j+8+l], x[j+16+l], x[j+24+l]
                    const float s0 = xk.x * q0, s1 = xk.y * q1, s2 = xk.z * q2, s3 = xk.w * q3;
                    const float sm = ((s0 + s1) + s2) + s3;                     // sum0.add(sum1).add(sum2).add(sum3)
                    acc[m] = __builtin_fmaf(sm, ws, acc[m]);                    // .fma(wScale, val)
                }
            }
        } else {
#pragma unroll
            for (int m = 0; m < NM; ++m) {
                // quant bytes of the chunk's 8 blocks: A_0..3 | A_4..7 | B_0..3 | B_4..7; nibble planes of four bytes at once
                const Q4Planes pa[2] = {q4_planes((uint32_t)wq[m][u].x, qc.magic), q4_planes((uint32_t)wq[m][u].y, qc.magic)};
                const Q4Planes pb[2] = {q4_planes((uint32_t)wq[m][u].z, qc.magic), q4_planes((uint32_t)wq[m][u].w, qc.magic)};
                const uint32_t sw[4] = {(uint32_t)wsc[m][u].x, (uint32_t)wsc[m][u].y, (uint32_t)wsc[m][u].z, (uint32_t)wsc[m][u].w};
#define VL_Q4_BLOCK(KK)                                                                                                            \
                {                                                                                                                  \
                    const float ws = ((KK) & 1) ? cvt_hi(sw[(KK) >> 1]) : cvt_lo(sw[(KK) >> 1]);                                   \
                    const float4 xk = *reinterpret_cast<const float4*>(xT + 32 * (8 * cq + (KK)) + 4 * l);   /* x[j+l], x[j+8+l], x[j+16+l], x[j+24+l] */ \
                    const float s0 = q4_mul<(KK) & 3, 0>(xk.x, pa[(KK) >> 2], qc.negzero), s1 = q4_mul<(KK) & 3, 0>(xk.y, pb[(KK) >> 2], qc.negzero);   \
                    const float s2 = q4_mul<(KK) & 3, 1>(xk.z, pa[(KK) >> 2], qc.negzero), s3 = q4_mul<(KK) & 3, 1>(xk.w, pb[(KK) >> 2], qc.negzero);   \
                    const float sm = ((s0 + s1) + s2) + s3;                     /* sum0.add(sum1).add(sum2).add(sum3) */          \
                    acc[m] = __builtin_fmaf(sm, ws, acc[m]);                    /* .fma(wScale, val) */                            \
                }
                VL_Q4_BLOCK(0) VL_Q4_BLOCK(1) VL_Q4_BLOCK(2) VL_Q4_BLOCK(3) VL_Q4_BLOCK(4) VL_Q4_BLOCK(5) VL_Q4_BLOCK(6) VL_Q4_BLOCK(7)
#undef VL_Q4_BLOCK
            }
        }
    };
    if (WT == WT_F16) set_f16_denorm_flush(true);
    float4 xa[XV], xb[XV];
    xload(xa, 0);
    int c0 = 0;
    for (; c0 + 2 * D <= nch; c0 += D) {           // branch-free: the whole next group of D chunks exists
#pragma unroll
        for (int u = 0; u < D; u += 2) {
            xload(xb, c0 + u + 1);
            consume(u, xa, c0 + u);
            issue(u, c0 + u + D);
            xload(xa, c0 + u + 2);                 // c0 + u + 2 <= c0 + D < nch
            consume(u + 1, xb, c0 + u + 1);
            issue(u + 1, c0 + u + 1 + D);
        }
    }
    for (; c0 < nch; c0 += D) {                    // last groups: guarded
#pragma unroll
        for (int u = 0; u < D; ++u) {
            const int c = c0 + u;
            if (c < nch) {
                if (u > 0 || c0 + 2 * D > nch) xload(xa, c);      // (re)load: the fast loop left xa at chunk c0 only
                consume(u, xa, c);
                if (c + D < nch) issue(u, c + D);
            }
        }
    }
    if (WT == WT_F16) set_f16_denorm_flush(false);
    // ---- reduceLanes(ADD) in lane order from 0, then the epilogue on the row's first lane
    float res[NM];
#pragma unroll
    for (int m = 0; m < NM; ++m) {
        float r = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) r = r + __shfl(acc[m], (lane & ~7) + j, 64);
        if (SPECIES == 512) {
#pragma unroll
            for (int j = 0; j < 8; ++j) r = r + __shfl(acc2[m], (lane & ~7) + j, 64);
        }
        res[m] = r;
    }
    const int row = g * 8 + rr;
    if (l == 0 && row < a.rows) {
        float o = 0.f;
        if (EPI == EPI_STORE) o = res[0] * a.out_scale;
        if (EPI == EPI_RESID) o = a.resid_in ? a.resid_in[row] + res[0] * a.out_scale : res[0] * a.out_scale;
        if (EPI == EPI_SWIGLU) {                               // InferenceCore.java:155-158, exp in double
            const float gte = res[0] / (float)(1.0 + exp(-(double)res[0]));
            o = gte * res[NM - 1];
        }
        a.out[row] = o;
        if (a.tp) tp_push_store(a.tp->p, a.out + row, o);
    }
    if (a.tp && a.tp->p.npeers) tp_publish(a.tp->p, ngroups);      // every live wavefront of the launch reaches its epilogue
}

// ---- Q4_0 / Q8_0-with-f32-activation, K split over the wavefronts of a workgroup.
// In these two dot products only ONE operation per 32-element block is on the row's dependent chain: val[l] = fma(s[l], wScale,
// val[l]).  The block sum s[l] (nibble / byte extraction, conversions, 4 multiplies, 3 adds: ~17 of the ~18 VALU instructions
// per block and lane) depends on the block alone.  matvec_vl_kernel gives a whole 8-row group to one wavefront, so a 4096-row
// matrix runs 512 wavefronts on 1024 SIMDs, each issuing alone on its SIMD (~2.4 ns per instruction instead of ~1.1 with two
// wavefronts interleaved): the Llama-3-8B Q4_0 down projection took 33 us for 33 MB.  Here a workgroup of 4 / 8 / 16 wavefronts
// owns one 8-row group; per round wavefront w computes the block sums of its contiguous K range (<= 64 blocks, kept in registers),
// then the wavefronts run their part of the fma chain one after the other, passing the 64 chain values (8 rows x 8 accumulator
// lanes) through LDS.  Same arithmetic in the same order -> same bits.  Each wavefront stages only its own slice of the activation
// (wave-private LDS, no workgroup barrier before the compute phase).  Loads are unconditional with clamped indices so that the
// waits are exact vmcnt counts (slice scattered while the weights fly, chunk u consumed while chunks u + 1 .. are in flight).
// Wavefronts per group (launcher, vq_waves): enough for >= 2 wavefronts per SIMD over the launch — a rank of a tensor-parallel
// group holds rows / tp rows — and for one round to cover K.  MAXW = 8: <= 256 VGPRs, 64 (Q4_0) / 48 (Q8_0) blocks of sums per
// wavefront and round; MAXW = 16: <= 128 VGPRs, half of that.
// Used for the residual projections (wo, down).  Variants that were measured and dropped (8B Q4_0, 8 layers + logits, us / token;
// this kernel: 734): a persistent version looping over groups with the RMSNorm in its prologue for qkv / gate-up / logits (1198:
// one exposed load latency per group); whole-workgroup staging of x (755); loads under wavefront-uniform branches (809).  This
// kernel's register allocation is fragile (hipcc, ROCm 7.2): equivalent restructurings of the chain loop spilled 130 - 330 bytes.
template <int WT, int MAXW>
__host__ __device__ constexpr int vq_round_chunks(int nm) { return (WT == WT_Q4_0 ? 8 : 12) / (MAXW == 16 ? 2 : 1) / nm; }
template <int WT, int MAXW>
__host__ __device__ inline int vq_rc(int k, int nm, int nw) {
    const int nch = k / (WT == WT_Q4_0 ? 256 : 128), cap = nw * vq_round_chunks<WT, MAXW>(nm);
    const int nrounds = (nch + cap - 1) / cap;
    return (nch + nw * nrounds - 1) / (nw * nrounds);
}
template <int WT, int MAXW>
__host__ __device__ inline size_t vq_smem_bytes(int k, int nm, int nw) {
    return (size_t)nm * 64 * 4 + (size_t)nw * vq_rc<WT, MAXW>(k, nm, nw) * (WT == WT_Q4_0 ? 256 : 128) * 4;
}
template <int WT>
__host__ inline int vq_waves(int k, int nm, int ngroups) {
    const int nch = k / (WT == WT_Q4_0 ? 256 : 128);
    int nw = 4;
    while (nw < 16 && (long)ngroups * nw < 2048) nw *= 2;
    if (nw == 4 && nch > 4 * vq_round_chunks<WT, 8>(nm)) nw = 8;
    while (nw > 4 && nch < nw) nw /= 2;
    return nw;
}

template <int WT, int EPI, int MAXW>
static __global__ __launch_bounds__(64 * MAXW) void matvec_vlq_kernel(const VlArgs a) {
    static_assert(WT == WT_Q4_0 || WT == WT_Q8_0, "K-split kernel: block-sum types only");
    extern __shared__ __attribute__((aligned(16))) float vq_smem[];
    constexpr int NM = EPI == EPI_SWIGLU ? 2 : 1;
    constexpr int CB = WT == WT_Q4_0 ? 8 : 4;                  // blocks per chunk
    constexpr int CE = CB * 32;                                // elements per chunk
    constexpr int RCM = vq_round_chunks<WT, MAXW>(NM);         // chunks per wavefront and round
    constexpr int CBYTES = WT == WT_Q4_0 ? 1152 : 1088;
    const int t = threadIdx.x, lane = t & 63, l = lane & 7, rr = lane >> 3;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int VQ_WAVES = blockDim.x >> 6;
    const int g = blockIdx.x;
    const int nch = a.k / CE;
    const int nrounds = (nch + VQ_WAVES * RCM - 1) / (VQ_WAVES * RCM);
    const int rc = (nch + VQ_WAVES * nrounds - 1) / (VQ_WAVES * nrounds);
    float* accx = vq_smem;
    float* xT = vq_smem + NM * 64 + wave * (rc * CE);
    const size_t gbytes = vl_group_bytes(WT, a.k);
    const uint8_t* wb[NM];
    wb[0] = a.w + (size_t)g * gbytes;
    if (NM == 2) wb[NM - 1] = a.w2 + (size_t)g * gbytes;

    int4 wq[NM][RCM];
    int4 wsc4[NM][WT == WT_Q4_0 ? RCM : 1];
    uint2 wsc8[NM][WT == WT_Q8_0 ? RCM : 1];
    // float4 loads per lane that stage one round's activation slice (RCM chunks of CE floats).  Rounded UP: Q8_0 / SWIGLU / 16
    // wavefronts has RCM * CE = 384, and a truncated 384 / 256 = 1 left the third chunk's slice unstaged (r3 advisor finding).
    constexpr int XQ = (RCM * CE + 255) / 256;
    static_assert(XQ * 256 >= RCM * CE, "the x staging registers must cover a round's chunks");
    float4 xr[XQ];
    float sm[NM][RCM * CB];

#define VQ_ISSUE(R_)                                                                                                     \
    do {                                                                                                                 \
        const int c_lo_ = min(((R_) * VQ_WAVES + wave) * rc, nch - 1);                                                   \
        const int nc_ = max(1, min(rc, nch - c_lo_));                                                                    \
        const int nq_ = nc_ * (CE / 4);                                                                                  \
        _Pragma("unroll") for (int u = 0; u < XQ; ++u)                                                                   \
            xr[u] = *reinterpret_cast<const float4*>(a.x + (size_t)c_lo_ * CE + 4 * min(u * 64 + lane, nq_ - 1));       \
        _Pragma("unroll") for (int u = 0; u < RCM; ++u) {                                                                \
            _Pragma("unroll") for (int m = 0; m < NM; ++m) {                                                             \
                const uint8_t* cb = wb[m] + (size_t)(c_lo_ + min(u, nc_ - 1)) * CBYTES;                                  \
                wq[m][u] = ld16<true>(cb + 16 * lane);                                                                   \
                if (WT == WT_Q4_0) wsc4[m][u] = ld16<true>(cb + 1024 + 16 * rr);                                         \
                if (WT == WT_Q8_0) wsc8[m][u] = *reinterpret_cast<const uint2*>(cb + 1024 + 8 * rr);                     \
            }                                                                                                            \
        }                                                                                                                \
    } while (0)

    VQ_ISSUE(0);
    const Q4Consts qc = q4_consts();
    float acc[NM];
#pragma unroll
    for (int m = 0; m < NM; ++m) acc[m] = 0.f;
    for (int r = 0; r < nrounds; ++r) {
        const int c_lo = (r * VQ_WAVES + wave) * rc;
        const int nc = max(0, min(rc, nch - c_lo));
        const int nq = nc * (CE / 4);
        // ---- activation slice -> wave-private LDS, transposed: xT[32 b + 4 l + q] = x[32 b + 8 q + l]
#pragma unroll
        for (int u = 0; u < XQ; ++u) {
            const int qd = u * 64 + lane;
            if (qd < nq) {
                const float vv[4] = {xr[u].x, xr[u].y, xr[u].z, xr[u].w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int i = 4 * qd + e;
                    xT[(i & ~31) + 4 * (i & 7) + ((i >> 3) & 3)] = vv[e];
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // ---- block sums of this wavefront's chunks
#pragma unroll
        for (int u = 0; u < RCM; ++u) {
            if (u < nc) {
#pragma unroll
                for (int m = 0; m < NM; ++m) {
                    if (WT == WT_Q8_0) {
                        const uint32_t qw[4] = {(uint32_t)wq[m][u].x, (uint32_t)wq[m][u].y, (uint32_t)wq[m][u].z, (uint32_t)wq[m][u].w};   // one block each
#pragma unroll
                        for (int kk = 0; kk < 4; ++kk) {
                            const float q0 = (float)(int8_t)(qw[kk] & 0xFFu), q1 = (float)(int8_t)((qw[kk] >> 8) & 0xFFu);      // castShape(F_SPECIES, i): exact
                            const float q2 = (float)(int8_t)((qw[kk] >> 16) & 0xFFu), q3 = (float)(int8_t)(qw[kk] >> 24);
                            const float4 xk = *reinterpret_cast<const float4*>(xT + 32 * (4 * u + kk) + 4 * l);   // x[j+l], x[j+8+l], x[j+16+l], x[j+24+l]
                            const float s0 = xk.x * q0, s1 = xk.y * q1, s2 = xk.z * q2, s3 = xk.w * q3;
                            sm[m][4 * u + kk] = ((s0 + s1) + s2) + s3;          // sum0.add(sum1).add(sum2).add(sum3)
                        }
                    } else {
                        // quant bytes of the chunk's 8 blocks: A_0..3 | A_4..7 | B_0..3 | B_4..7; nibble planes of four bytes at once
                        const Q4Planes pa[2] = {q4_planes((uint32_t)wq[m][u].x, qc.magic), q4_planes((uint32_t)wq[m][u].y, qc.magic)};
                        const Q4Planes pb[2] = {q4_planes((uint32_t)wq[m][u].z, qc.magic), q4_planes((uint32_t)wq[m][u].w, qc.magic)};
#define VQ_Q4_BLOCK(KK)                                                                                                            \
                        {                                                                                                          \
                            const float4 xk = *reinterpret_cast<const float4*>(xT + 32 * (8 * u + (KK)) + 4 * l);   /* x[j+l], x[j+8+l], x[j+16+l], x[j+24+l] */ \
                            const float s0 = q4_mul<(KK) & 3, 0>(xk.x, pa[(KK) >> 2], qc.negzero), s1 = q4_mul<(KK) & 3, 0>(xk.y, pb[(KK) >> 2], qc.negzero);   \
                            const float s2 = q4_mul<(KK) & 3, 1>(xk.z, pa[(KK) >> 2], qc.negzero), s3 = q4_mul<(KK) & 3, 1>(xk.w, pb[(KK) >> 2], qc.negzero);   \
                            sm[m][8 * u + (KK)] = ((s0 + s1) + s2) + s3;       /* sum0.add(sum1).add(sum2).add(sum3) */           \
                        }
                        VQ_Q4_BLOCK(0) VQ_Q4_BLOCK(1) VQ_Q4_BLOCK(2) VQ_Q4_BLOCK(3) VQ_Q4_BLOCK(4) VQ_Q4_BLOCK(5) VQ_Q4_BLOCK(6) VQ_Q4_BLOCK(7)
#undef VQ_Q4_BLOCK
                    }
                }
            }
        }
        // ---- the chain, wavefront after wavefront (K order); the scales stay packed until their fma
        for (int w = 0; w < VQ_WAVES; ++w) {
            if (wave == w && nc > 0) {
                if (!(r == 0 && w == 0)) {
#pragma unroll
                    for (int m = 0; m < NM; ++m) acc[m] = accx[m * 64 + lane];
                }
#pragma unroll
                for (int u = 0; u < RCM; ++u) {
                    if (u < nc) {
#pragma unroll
                        for (int m = 0; m < NM; ++m) {
#pragma unroll
                            for (int kk = 0; kk < CB; ++kk) {
                                float ws;
                                if (WT == WT_Q8_0) {
                                    const uint32_t sw = (kk >> 1) ? wsc8[m][u].y : wsc8[m][u].x;
                                    ws = h2f((uint16_t)((kk & 1) ? (sw >> 16) : (sw & 0xFFFFu)));     // Float.float16ToFloat: IEEE, subnormals kept
                                } else {
                                    const uint32_t sw4[4] = {(uint32_t)wsc4[m][u].x, (uint32_t)wsc4[m][u].y, (uint32_t)wsc4[m][u].z, (uint32_t)wsc4[m][u].w};
                                    ws = (kk & 1) ? cvt_hi(sw4[kk >> 1]) : cvt_lo(sw4[kk >> 1]);
                                }
                                acc[m] = __builtin_fmaf(sm[m][CB * u + kk], ws, acc[m]);          // .fma(wScale, val)
                            }
                        }
                    }
                }
#pragma unroll
                for (int m = 0; m < NM; ++m) accx[m * 64 + lane] = acc[m];
            }
            // next round's loads once the wavefront's own scales are used up: they fly during the remaining chain phases
            if (wave == w && r + 1 < nrounds) VQ_ISSUE(r + 1);
            __syncthreads();
        }
    }
#undef VQ_ISSUE
    if (wave != 0) return;
    // ---- reduceLanes(ADD) in lane order from 0, then the epilogue on the row's first lane
    float res[NM];
#pragma unroll
    for (int m = 0; m < NM; ++m) {
        const float v = accx[m * 64 + lane];
        float rsum = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) rsum = rsum + __shfl(v, (lane & ~7) + j, 64);
        res[m] = rsum;
    }
    const int row = g * 8 + rr;
    if (l == 0 && row < a.rows) {
        float o = 0.f;
        if (EPI == EPI_STORE) o = res[0] * a.out_scale;
        if (EPI == EPI_RESID) o = a.resid_in ? a.resid_in[row] + res[0] * a.out_scale : res[0] * a.out_scale;
        if (EPI == EPI_SWIGLU) {                               // InferenceCore.java:155-158, exp in double
            const float gte = res[0] / (float)(1.0 + exp(-(double)res[0]));
            o = gte * res[NM - 1];
        }
        a.out[row] = o;
        if (a.tp) tp_push_store(a.tp->p, a.out + row, o);
    }
    if (a.tp && a.tp->p.npeers) tp_publish(a.tp->p, gridDim.x);      // wavefront 0 of every workgroup
}

}  // namespace gl3
