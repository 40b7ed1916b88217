// gl3_prefill_vl.h — batched prefill / static-batched decode for the weight types whose reference dot product runs on an f32
// activation in the Vector-API order (256-bit species, 8 accumulator lanes per row): F16, Q4_0 and Q8_0 with
// -Dllama.quantizeActivation=false.  Replaces, for these types, the token-by-token fall-back of rounds 1-2; the reference's
// batched graphs are J/tornadovm/layers/type/fp16/prefill/LlamaFP16LayersBatchPrefillMMA.java / Qwen3FP16LayersBatchPrefillMMA.java
// (f16 tensor-core MMA), the arithmetic reproduced here is the CPU path's: InferenceCoreBatchPrefillDecode.batchForwardJavaPrefill
// (J/inference/InferenceCoreBatchPrefillDecode.java:62-168) = FloatTensor.matmul(context, ...) :102-111 over
//   FP16FloatTensor.vectorDot  (J/tensor/standard/FP16FloatTensor.java:63-110),
//   Q4_0FloatTensor.vectorDot  (J/tensor/standard/Q4_0FloatTensor.java:82-133),
//   Q8_0FloatTensor.vectorDot  (J/tensor/standard/Q8_0FloatTensor.java:125-175).
// The order of a dot product is per (row, accumulator lane): val[l] = fma(., ., val[l]) over the row's chunks, then
// reduceLanes in lane order.  Tokens are independent, so a token tile only adds a loop over accumulators.
//
//   * F16: val[l] = fma(w[8 i + l], x[8 i + l], val[l]) is an i-ordered f32 FMA chain per (row, l, token) = exactly what
//     v_mfma_f32_32x32x2_f32 computes (D = fma(a_k1, b_k1, fma(a_k0, b_k0, C)), one rounding per step — MI355X_MICROARCH.md
//     "f32-input MFMA"; the same property the decode kernels use for their ordered adds).  Eight GEMMs, one per accumulator
//     lane l, with K' = K / 8 steps each: C_l[row][token]; A_l[row][i] = DAZ-converted w[row][8 i + l], B_l[i][token] =
//     x[token][8 i + l].  The weights stay in the decode path's VL layout (no second copy): a workgroup stages the VL chunks
//     of its 64 rows in LDS and every lane picks the halfs of its (row, k) A slot.
//   * Q4_0 / Q8_0 (f32 activation): per block s[l] = ((x0*q0 + x1*q1) + x2*q2) + x3*q3 has ROUNDED products (no FMA), so it
//     stays on the VALU: lane = (row, l) as in matvec_vl_kernel, two 8-row groups per wavefront, the dequantised quants of a
//     block are computed once and reused for all 32 tokens of the tile; x comes from LDS in the decode kernel's transposed
//     layout (one broadcast ds_read_b128 per block and token).  2 VALU lane-operations per weight and token: VALU-bound.
#pragma once
#include "gl3_veclane_kernels.h"

namespace gl3 {

typedef float v16f_vl __attribute__((ext_vector_type(16)));

struct VlGemmArgs {
    const uint8_t* w;            // VL matrix
    int rows, k;
    const float* X; int x_stride;    // f32 activations [ntok][x_stride]
    int ntok;
    float* out; int out_stride;      // EPI_STORE: out[b][row] = r * out_scale; EPI_RESID: out[b][row] += r * out_scale
    float out_scale;
    int nrt, ntt;                    // row tiles, token tiles (grid = 8 * ceil(nrt * ntt / 8), XCD-aware mapping)
    int xcd_tokens;                  // gemm_vlq_mfma_kernel: 1 = vqm_tile_of (token tiles pinned to XCDs), 0 = vl_tile_of
};

// XCD-aware tile mapping shared by both kernels: workgroups are dealt round-robin to the 8 XCDs, so the token tiles that share
// a weight row tile get consecutive slots of ONE XCD (one L2 streams the weights once).
__device__ __forceinline__ bool vl_tile_of(const VlGemmArgs& a, int& rt, int& tt) {
    const int per_xcd = (a.ntt * a.nrt + 7) >> 3;
    const int lin = blockIdx.x, J = (lin & 7) * per_xcd + (lin >> 3);
    if (J >= a.ntt * a.nrt) return false;
    rt = J / a.ntt; tt = J % a.ntt;
    return true;
}

// ---------------------------------------------------------------------------------------------------
// F16 on the f32 matrix cores.  Workgroup = 4 wavefronts in a 2 x 2 grid, tile = 64 rows x 64 tokens, every wavefront a
// 32 x 32 sub-tile for all 8 accumulator lanes (8 x 16 accumulator registers).  K advances one VL chunk (64 elements = 8 chain
// steps per accumulator lane = 4 MFMAs per lane) per stage; the next stage travels HBM/L2 -> registers while this one is
// consumed from LDS.
//   LDS per stage: A[8 l][65][16 B] (slot l * 65 + row: the 8 halfs w[row][64 c + 8 k + l], k = 0..7; pitch 65 keeps the
//   row-wise reads and the l-wise writes conflict-free) | B[64 tokens][68 floats]
constexpr int F16G_ROWS = 64, F16G_TOK = 64;
constexpr int F16G_A_BYTES = 8 * 65 * 16, F16G_B_PITCH = 68, F16G_B_BYTES = F16G_TOK * F16G_B_PITCH * 4;
constexpr int F16G_STAGE = F16G_A_BYTES + F16G_B_BYTES;

template <int EPI>
__global__ __launch_bounds__(256, 2) void gemm_f16_mfma_kernel(const VlGemmArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6), wr = wave >> 1, wc = wave & 1;
    int rt, tt;
    if (!vl_tile_of(a, rt, tt)) return;
    const int row0 = rt * F16G_ROWS, tok0 = tt * F16G_TOK;
    const int nch = a.k >> 6, ngroups = (a.rows + 7) >> 3;
    const size_t gbytes = (size_t)nch * 1024;
    // ---- global -> registers of one K stage: A = 8 row groups x 64 lanes x 16 B (2 pieces per thread), B = 64 tokens x 16
    // float4 (4 pieces per thread); rows / tokens past the end re-read the last valid one (never stored)
    int4 ra[2]; float4 rb[4];
    const uint8_t* pa[2]; const float* pb[4];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int p = t + 256 * j, g = min(ngroups - 1, (row0 >> 3) + (p >> 6));
        pa[j] = a.w + (size_t)g * gbytes + 16 * (p & 63);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int p = t + 256 * j, tk = min(a.ntok - 1, tok0 + (p >> 4));
        pb[j] = a.X + (size_t)tk * a.x_stride + 4 * (p & 15);
    }
#define F16G_GLOAD(c_) do { \
        _Pragma("unroll") for (int j = 0; j < 2; ++j) ra[j] = ld16<false>(pa[j] + (size_t)(c_) * 1024); \
        _Pragma("unroll") for (int j = 0; j < 4; ++j) rb[j] = *reinterpret_cast<const float4*>(pb[j] + 64 * (c_)); \
    } while (0)
#define F16G_LSTORE(stage_) do { \
        uint8_t* A_ = smem + (size_t)(stage_) * F16G_STAGE; \
        float* B_ = reinterpret_cast<float*>(A_ + F16G_A_BYTES); \
        _Pragma("unroll") for (int j = 0; j < 2; ++j) { \
            const int p = t + 256 * j, grp = p >> 6, r = (p >> 3) & 7, l = p & 7; \
            *reinterpret_cast<int4*>(A_ + ((size_t)l * 65 + grp * 8 + r) * 16) = ra[j]; \
        } \
        _Pragma("unroll") for (int j = 0; j < 4; ++j) { \
            const int p = t + 256 * j; \
            *reinterpret_cast<float4*>(B_ + (p >> 4) * F16G_B_PITCH + 4 * (p & 15)) = rb[j]; \
        } \
    } while (0)
    v16f_vl acc[8];
#pragma unroll
    for (int l = 0; l < 8; ++l)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[l][r] = 0.f;
    const int mi = lane & 31, kk = lane >> 5;          // MFMA operand slot: A[row mi][k kk], B[k kk][token mi]
    const uint32_t hsh = 16 * kk;                      // half 2 s + kk of a 32-bit word pair: word s, shifted by 16 kk
    set_f16_denorm_flush(true);                        // v_cvt_f32_f16 flushes subnormal weights: the reference's DAZ bit trick
    F16G_GLOAD(0);
    F16G_LSTORE(0);
    __syncthreads();
    for (int c = 0; c < nch; ++c) {
        F16G_GLOAD(min(c + 1, nch - 1));               // unconditional (a load under a condition is spilled / drained): the last trip re-reads its own chunk
        const uint8_t* A = smem + (size_t)(c & 1) * F16G_STAGE;
        const float* B = reinterpret_cast<const float*>(A + F16G_A_BYTES) + (wc * 32 + mi) * F16G_B_PITCH + 8 * kk;
        int4 aw[8];
#pragma unroll
        for (int l = 0; l < 8; ++l) aw[l] = *reinterpret_cast<const int4*>(A + ((size_t)l * 65 + wr * 32 + mi) * 16);
#pragma unroll
        for (int s = 0; s < 4; ++s) {                  // chain steps 2 s (k = 0) and 2 s + 1 (k = 1) of every accumulator lane
            const float4 b0 = *reinterpret_cast<const float4*>(B + 16 * s), b1 = *reinterpret_cast<const float4*>(B + 16 * s + 4);
            const float bx[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
#pragma unroll
            for (int l = 0; l < 8; ++l) {
                const uint32_t wd = (uint32_t)(s == 0 ? aw[l].x : s == 1 ? aw[l].y : s == 2 ? aw[l].z : aw[l].w);
                const float af = cvt_lo(wd >> hsh);
                acc[l] = __builtin_amdgcn_mfma_f32_32x32x2f32(af, bx[l], acc[l], 0, 0, 0);
            }
        }
        F16G_LSTORE((c + 1) & 1);
        __syncthreads();
    }
#undef F16G_GLOAD
#undef F16G_LSTORE
    set_f16_denorm_flush(false);
    // ---- reduceLanes(ADD) in lane order from 0, then the epilogue.  C layout: token = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 kk
    const int b = tok0 + wc * 32 + mi;
    if (b >= a.ntok) return;
    float* o = a.out + (size_t)b * a.out_stride;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = row0 + wr * 32 + (r & 3) + 8 * (r >> 2) + 4 * kk;
        if (row >= a.rows) continue;
        float v = 0.f;
#pragma unroll
        for (int l = 0; l < 8; ++l) v = v + acc[l][r];
        if (EPI == EPI_RESID) o[row] = o[row] + v * a.out_scale;
        else o[row] = v * a.out_scale;
    }
}

// ---------------------------------------------------------------------------------------------------
// Q4_0 / Q8_0 with f32 activation on the VALU.  Workgroup = 4 wavefronts = 64 rows (wavefront = two 8-row VL groups, lane =
// (row, accumulator lane l)) x 16 tokens; K advances one VL chunk per stage (Q4_0: 8 blocks, Q8_0: 4 blocks).  x of the
// token tile is staged in LDS in the decode kernel's transposed order (xT[32 b + 4 l + q] = x[32 b + 8 q + l]: a lane's four
// operands of a block are one ds_read_b128, the same address for the 8 lanes of equal l -> broadcast); the weights go
// straight from global memory to the registers of the wavefront that owns the rows.
constexpr int VLQ_TOK = 16;
template <int WT>
__host__ __device__ constexpr int vlq_stage_floats() { return VLQ_TOK * (vl_chunk_elems(WT) + 4); }     // row pitch + 4: token rows on different banks

template <int WT, int EPI>
__global__ __launch_bounds__(256, 2) void gemm_vlq_kernel(const VlGemmArgs a) {
    static_assert(WT == WT_Q4_0 || WT == WT_Q8_0, "F16 runs on gemm_f16_mfma_kernel");
    extern __shared__ __attribute__((aligned(16))) float xs_[];
    constexpr int CE = WT == WT_Q4_0 ? 256 : 128, NB = CE / 32, CB = WT == WT_Q4_0 ? 1152 : 1088, PITCH = CE + 4;
    constexpr int XP = VLQ_TOK * CE / 4 / 256;          // float4 pieces of the x tile per thread and stage
    const int t = threadIdx.x, lane = t & 63, l = lane & 7, rr = lane >> 3;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    int rt, tt;
    if (!vl_tile_of(a, rt, tt)) return;
    const int tok0 = tt * VLQ_TOK;
    const int nch = a.k / CE, ngroups = (a.rows + 7) >> 3;
    const size_t gbytes = (size_t)nch * CB;
    const int g0 = rt * 8 + wave * 2;                   // first of this wavefront's two row groups
    const uint8_t* wb[2];
#pragma unroll
    for (int gi = 0; gi < 2; ++gi) wb[gi] = a.w + (size_t)min(ngroups - 1, g0 + gi) * gbytes;
    const float* px[XP];
#pragma unroll
    for (int j = 0; j < XP; ++j) {
        const int p = t + 256 * j, tk = min(a.ntok - 1, tok0 + p / (CE / 4));
        px[j] = a.X + (size_t)tk * a.x_stride + 4 * (p % (CE / 4));
    }
    int4 wq[2], wsc[2]; float4 rx[XP];
    // (macros, not reference-capturing lambdas: the staging arrays must stay in registers)
#define VLQ_GLOAD(c_) do { \
        _Pragma("unroll") for (int gi = 0; gi < 2; ++gi) { \
            const uint8_t* cb = wb[gi] + (size_t)(c_) * CB; \
            wq[gi] = ld16<false>(cb + 16 * lane); \
            if (WT == WT_Q4_0) wsc[gi] = ld16<false>(cb + 1024 + 16 * rr); \
            else { const uint2 s2 = *reinterpret_cast<const uint2*>(cb + 1024 + 8 * rr); wsc[gi] = make_int4((int)s2.x, (int)s2.y, 0, 0); } \
        } \
        _Pragma("unroll") for (int j = 0; j < XP; ++j) rx[j] = *reinterpret_cast<const float4*>(px[j] + (size_t)(c_) * CE); \
    } while (0)
#define VLQ_LSTORE(stage_) do { \
        float* xT_ = xs_ + (size_t)(stage_) * vlq_stage_floats<WT>(); \
        _Pragma("unroll") for (int j = 0; j < XP; ++j) { \
            const int p = t + 256 * j, tk = p / (CE / 4), i0 = 4 * (p % (CE / 4)); \
            const float vv[4] = {rx[j].x, rx[j].y, rx[j].z, rx[j].w}; \
            _Pragma("unroll") for (int e = 0; e < 4; ++e) { \
                const int i = i0 + e; \
                xT_[tk * PITCH + (i & ~31) + 4 * (i & 7) + ((i >> 3) & 3)] = vv[e]; \
            } \
        } \
    } while (0)
    float acc[2][VLQ_TOK];
#pragma unroll
    for (int gi = 0; gi < 2; ++gi)
#pragma unroll
        for (int b = 0; b < VLQ_TOK; ++b) acc[gi][b] = 0.f;
    VLQ_GLOAD(0);
    VLQ_LSTORE(0);
    __syncthreads();
    for (int c = 0; c < nch; ++c) {
        const int4 cq[2] = {wq[0], wq[1]}, cs[2] = {wsc[0], wsc[1]};       // this stage's weights (the loads below reuse the registers)
        VLQ_GLOAD(min(c + 1, nch - 1));                // unconditional; the last trip re-reads its own chunk
        const float* xT = xs_ + (size_t)(c & 1) * vlq_stage_floats<WT>() + 4 * l;
        // Blocks in a ROLLED loop (the unrolled nest of 8 blocks x 16 tokens x 2 groups made the scheduler keep hundreds of
        // values live: 256 VGPRs + 1.8 KB of spills); the block's words are picked with wave-uniform selects and shifts.
#pragma unroll 1
        for (int kb = 0; kb < NB; ++kb) {
            // the block's four dequantised quants of this lane (exact small integers) and its scale, for both row groups
            float q0[2], q1[2], q2[2], q3[2], ws[2];
#pragma unroll
            for (int gi = 0; gi < 2; ++gi) {
                if (WT == WT_Q4_0) {
                    const uint32_t aw = (uint32_t)(kb < 4 ? cq[gi].x : cq[gi].y), bw = (uint32_t)(kb < 4 ? cq[gi].z : cq[gi].w);
                    const int sh = 8 * (kb & 3);
                    const uint32_t ab = (aw >> sh) & 0xFFu, bb = (bw >> sh) & 0xFFu;        // byte l / byte 8 + l of block kb
                    q0[gi] = (float)(ab & 0xFu) - 8.0f; q2[gi] = (float)(ab >> 4) - 8.0f;   // elements l, 16 + l
                    q1[gi] = (float)(bb & 0xFu) - 8.0f; q3[gi] = (float)(bb >> 4) - 8.0f;   // elements 8 + l, 24 + l
                    const uint32_t sw = (uint32_t)(kb < 4 ? (kb < 2 ? cs[gi].x : cs[gi].y) : (kb < 6 ? cs[gi].z : cs[gi].w));
                    ws[gi] = cvt_lo(sw >> (16 * (kb & 1)));
                } else {
                    const uint32_t qw = (uint32_t)(kb < 2 ? (kb == 0 ? cq[gi].x : cq[gi].y) : (kb == 2 ? cq[gi].z : cq[gi].w));
                    q0[gi] = (float)(int8_t)(qw & 0xFFu); q1[gi] = (float)(int8_t)((qw >> 8) & 0xFFu);
                    q2[gi] = (float)(int8_t)((qw >> 16) & 0xFFu); q3[gi] = (float)(int8_t)(qw >> 24);
                    const uint32_t sw = (uint32_t)(kb < 2 ? cs[gi].x : cs[gi].y);
                    ws[gi] = h2f((uint16_t)((sw >> (16 * (kb & 1))) & 0xFFFFu));
                }
            }
            const float* xb = xT + 32 * kb;
#pragma unroll
            for (int b0 = 0; b0 < VLQ_TOK; b0 += 4) {       // four tokens' operands per LDS round trip
                float4 xk[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) xk[j] = *reinterpret_cast<const float4*>(xb + (b0 + j) * PITCH);   // x[j+l], x[j+8+l], x[j+16+l], x[j+24+l]
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int gi = 0; gi < 2; ++gi) {
                        const float s0 = xk[j].x * q0[gi], s1 = xk[j].y * q1[gi], s2 = xk[j].z * q2[gi], s3 = xk[j].w * q3[gi];
                        const float sm = ((s0 + s1) + s2) + s3;                     // sum0.add(sum1).add(sum2).add(sum3)
                        acc[gi][b0 + j] = __builtin_fmaf(sm, ws[gi], acc[gi][b0 + j]);    // .fma(wScale, val)
                    }
            }
        }
        VLQ_LSTORE((c + 1) & 1);
        __syncthreads();
    }
#undef VLQ_GLOAD
#undef VLQ_LSTORE
    // ---- reduceLanes(ADD) in lane order from 0; the row's first lane stores
#pragma unroll
    for (int gi = 0; gi < 2; ++gi) {
        const int row = (g0 + gi) * 8 + rr;
#pragma unroll
        for (int b = 0; b < VLQ_TOK; ++b) {
            float r = 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) r = r + __shfl(acc[gi][b], (lane & ~7) + j, 64);
            if (l == 0 && row < a.rows && tok0 + b < a.ntok) {
                float* o = a.out + (size_t)(tok0 + b) * a.out_stride + row;
                if (EPI == EPI_RESID) *o = *o + r * a.out_scale;
                else *o = r * a.out_scale;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// Q4_0 / Q8_0 with f32 activation, round 5: the ROUNDED PRODUCTS on the f32 matrix cores, the ordered adds on the VALU.
//   s[l] = ((x0*q0 + x1*q1) + x2*q2) + x3*q3 needs every product rounded to f32 on its own, which no k-accumulating MFMA gives —
//   but a K = 1 MFMA with C = 0 is exactly that: v_mfma_f32_16x16x1_4b_f32 computes four independent 16 x 16 outer products
//   D_q[i][j] = fma(a_q[i], b_q[j], 0) = fl(a_q[i] * b_q[j]) (one rounding per product, subnormals kept: cdna_hip_programming.md
//   "f32-input MFMA"; a zero product of either sign plus +0 leaves the sums' and the fma's values unchanged).  With block q = the
//   four elements 8 q + l of a weight block, A = x of 16 tokens and B = the dequantised quants of 16 rows, ONE instruction delivers
//   the four product tiles P0..P3 of (16 tokens x 16 rows, block, accumulator lane l); the VALU then does the reference's three
//   adds and the fma with the row's scale on the 4 registers of the tile (v_pk_add_f32 / v_pk_fma_f32): 2 packed VALU
//   instructions per MFMA-delivered product quad instead of 4, and the matrix pipe (64 FLOP/clk/SIMD, the f32 VALU's own rate)
//   runs beside them.  Both pipes are ~balanced: 8 packed VALU instructions per 32-cycle MFMA.
// Workgroup = 8 wavefronts = 64 rows x 64 tokens; wavefront (rt, th) = 16 rows x 32 tokens (two tiles) for all 8 accumulator
// lanes (64 accumulator registers).  K advances 64 elements (two blocks) per stage through double-buffered LDS:
//   XF[64 tokens][68]                 x, natural order (A operand of lane (q, token): 8 consecutive floats = l 0..7)
//   QF[2 blocks][8 l][4 q][..][row]   dequantised quants as f32, index b * SB + l * SL + q * SQ + row (SL = 68, SQ = 560: the
//                                     staging thread (row, l) writes and the operand lane (q, row) reads without bank conflicts)
//   WS[2 blocks][64 rows]             block scales as f32
// The weights stay in the decode path's VL layout: thread (group, row, l) of the workgroup holds its 16 bytes of a VL chunk
// (8 / 4 blocks) in registers and dequantises two blocks per stage.
constexpr int VQM_ROWS = 64, VQM_TOK = 64, VQM_XP = 68, VQM_SL = 68, VQM_SQ = 560, VQM_SB = 4 * VQM_SQ;
constexpr int VQM_STAGE_FLOATS = VQM_TOK * VQM_XP + 2 * VQM_SB + 2 * VQM_ROWS;

// Tile mapping: the f32 x tile of a workgroup (64 tokens x K x 4 B: 1 MB at K = 4096) is 7x the bytes of its weight tile, so each
// XCD keeps a FIXED set of token tiles (its x stays in that XCD's 4 MB L2 while the row tiles stream past) and the weights are
// re-read once per XCD — against vl_tile_of's "token tiles of one row tile share an XCD", which re-reads all of x (8.4 MB at 512
// tokens: more than one L2) for every row tile: 7 GB of x per 8B layer.  G = token groups (a power of two <= 8), P = 8 / G row
// parts: XCD x = blockIdx.x & 7 works on token tiles g + G i (g = x % G) and row tiles part + P j (part = x / G), rows fastest.
__host__ __device__ inline int vqm_groups(int ntt) { return ntt >= 8 ? 8 : ntt >= 4 ? 4 : ntt >= 2 ? 2 : 1; }
__host__ __device__ inline int vqm_grid(int nrt, int ntt) {
    const int G = vqm_groups(ntt), P = 8 / G;
    return 8 * ((nrt + P - 1) / P) * ((ntt + G - 1) / G);
}
__device__ __forceinline__ bool vqm_tile_of(const VlGemmArgs& a, int& rt, int& tt) {
    const int G = vqm_groups(a.ntt), P = 8 / G, x = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int nrp = (a.nrt + P - 1) / P;
    rt = (x / G) + P * (slot % nrp);
    tt = (x % G) + G * (slot / nrp);
    return rt < a.nrt && tt < a.ntt;
}

template <int WT, int EPI, int OCC = 2>
__global__ __launch_bounds__(512, OCC) void gemm_vlq_mfma_kernel(const VlGemmArgs a) {
    static_assert(WT == WT_Q4_0 || WT == WT_Q8_0, "F16 runs on gemm_f16_mfma_kernel");
    extern __shared__ __attribute__((aligned(16))) float sm_[];
    constexpr int CE = WT == WT_Q4_0 ? 256 : 128, CB = WT == WT_Q4_0 ? 1152 : 1088, SPC = CE / 64;      // stages per VL chunk
    const int t = threadIdx.x, lane = t & 63, l = lane & 7, rr = lane >> 3;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6), rt = wave >> 1, th = wave & 1;
    int rtile, ttile;
    if (!(a.xcd_tokens ? vqm_tile_of(a, rtile, ttile) : vl_tile_of(a, rtile, ttile))) return;
    const int row0 = rtile * VQM_ROWS, tok0 = ttile * VQM_TOK;
    const int nst = a.k >> 6, ngroups = (a.rows + 7) >> 3;
    const size_t gbytes = (size_t)(a.k / CE) * CB;
    // staging roles: weights — thread = (group wave, lane) of the VL layout; x — two float4 per thread and stage
    const uint8_t* wb = a.w + (size_t)min(ngroups - 1, rtile * 8 + wave) * gbytes;
    const float* px[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int p = t + 512 * j, tk = min(a.ntok - 1, tok0 + (p >> 4));
        px[j] = a.X + (size_t)tk * a.x_stride + 4 * (p & 15);
    }
    int4 wq, wsc; float4 rx[2];
#define VQM_WLOAD(c_) do { \
        const uint8_t* cb_ = wb + (size_t)(c_) * CB; \
        wq = ld16<false>(cb_ + 16 * lane); \
        if (WT == WT_Q4_0) wsc = ld16<false>(cb_ + 1024 + 16 * rr); \
        else { const uint2 s2_ = *reinterpret_cast<const uint2*>(cb_ + 1024 + 8 * rr); wsc = make_int4((int)s2_.x, (int)s2_.y, 0, 0); } \
    } while (0)
#define VQM_XLOAD(s_) do { \
        _Pragma("unroll") for (int j = 0; j < 2; ++j) rx[j] = *reinterpret_cast<const float4*>(px[j] + (size_t)(s_) * 64); \
    } while (0)
    // stage s_ of chunk registers cq_ / cs_ -> LDS buffer buf_: x as loaded, the two blocks' quants as f32, their scales
#define VQM_LSTORE(buf_, s_, cq_, cs_) do { \
        float* XF_ = sm_ + (size_t)(buf_) * VQM_STAGE_FLOATS; \
        float* QF_ = XF_ + VQM_TOK * VQM_XP; \
        float* WS_ = QF_ + 2 * VQM_SB; \
        _Pragma("unroll") for (int j = 0; j < 2; ++j) { \
            const int p = t + 512 * j; \
            *reinterpret_cast<float4*>(XF_ + (p >> 4) * VQM_XP + 4 * (p & 15)) = rx[j]; \
        } \
        const int sub_ = (s_) % SPC; \
        float* qd_ = QF_ + l * VQM_SL + wave * 8 + rr; \
        if (WT == WT_Q4_0) { \
            const uint32_t aw_ = (uint32_t)(sub_ < 2 ? (cq_).x : (cq_).y) >> (16 * (sub_ & 1)); \
            const uint32_t bw_ = (uint32_t)(sub_ < 2 ? (cq_).z : (cq_).w) >> (16 * (sub_ & 1)); \
            _Pragma("unroll") for (int b = 0; b < 2; ++b) { \
                const uint32_t ab_ = (aw_ >> (8 * b)) & 0xFFu, bb_ = (bw_ >> (8 * b)) & 0xFFu; \
                qd_[b * VQM_SB + 0 * VQM_SQ] = (float)(ab_ & 0xFu) - 8.0f;      /* element l */      \
                qd_[b * VQM_SB + 1 * VQM_SQ] = (float)(bb_ & 0xFu) - 8.0f;      /* element 8 + l */  \
                qd_[b * VQM_SB + 2 * VQM_SQ] = (float)(ab_ >> 4) - 8.0f;        /* element 16 + l */ \
                qd_[b * VQM_SB + 3 * VQM_SQ] = (float)(bb_ >> 4) - 8.0f;        /* element 24 + l */ \
            } \
        } else { \
            _Pragma("unroll") for (int b = 0; b < 2; ++b) { \
                const uint32_t qw_ = (uint32_t)(sub_ == 0 ? (b == 0 ? (cq_).x : (cq_).y) : (b == 0 ? (cq_).z : (cq_).w)); \
                qd_[b * VQM_SB + 0 * VQM_SQ] = (float)(int8_t)(qw_ & 0xFFu); \
                qd_[b * VQM_SB + 1 * VQM_SQ] = (float)(int8_t)((qw_ >> 8) & 0xFFu); \
                qd_[b * VQM_SB + 2 * VQM_SQ] = (float)(int8_t)((qw_ >> 16) & 0xFFu); \
                qd_[b * VQM_SB + 3 * VQM_SQ] = (float)(int8_t)(qw_ >> 24); \
            } \
        } \
        if (l == 0) { \
            const uint32_t sw_ = (uint32_t)(sub_ == 0 ? (cs_).x : sub_ == 1 ? (cs_).y : sub_ == 2 ? (cs_).z : (cs_).w); \
            WS_[wave * 8 + rr] = cvt_lo(sw_); \
            WS_[VQM_ROWS + wave * 8 + rr] = cvt_hi(sw_); \
        } \
    } while (0)

    typedef float v2f_q __attribute__((ext_vector_type(2)));
    v2f_q val[2][8][2];                              // [token tile][accumulator lane][register pair]: tokens 4 (lane >> 4) + 0..3, row lane & 15
#pragma unroll
    for (int tt = 0; tt < 2; ++tt)
#pragma unroll
        for (int ll = 0; ll < 8; ++ll) { val[tt][ll][0] = (v2f_q){0.f, 0.f}; val[tt][ll][1] = (v2f_q){0.f, 0.f}; }
    const bool live = tok0 + th * 32 < a.ntok;       // a token half past the end only helps staging (static-batched decode, B <= 32)
    const int mq = lane >> 4, mi = lane & 15;        // MFMA operand slot: block q, token / row index
    VQM_WLOAD(0);
    VQM_XLOAD(0);
    VQM_LSTORE(0, 0, wq, wsc);
    int4 cq = wq, cs = wsc;                          // chunk registers the stages of the current chunk read
    __syncthreads();
    for (int s = 0; s < nst; ++s) {
        const int sn = min(s + 1, nst - 1);          // unconditional loads; the last trip re-reads its own stage
        if ((sn % SPC) == 0) VQM_WLOAD(sn / SPC);    // wave-uniform: the next stage opens a new VL chunk
        VQM_XLOAD(sn);
        if (live) {
            const float* XF = sm_ + (size_t)(s & 1) * VQM_STAGE_FLOATS;
            const float* QF = XF + VQM_TOK * VQM_XP;
            const float* WS = QF + 2 * VQM_SB;
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                float av[2][8], bv[8];
#pragma unroll
                for (int tt = 0; tt < 2; ++tt) {
                    const float* xa = XF + (th * 32 + tt * 16 + mi) * VQM_XP + 32 * b + 8 * mq;
                    const float4 lo = *reinterpret_cast<const float4*>(xa), hi = *reinterpret_cast<const float4*>(xa + 4);
                    av[tt][0] = lo.x; av[tt][1] = lo.y; av[tt][2] = lo.z; av[tt][3] = lo.w;
                    av[tt][4] = hi.x; av[tt][5] = hi.y; av[tt][6] = hi.z; av[tt][7] = hi.w;
                }
#pragma unroll
                for (int ll = 0; ll < 8; ++ll) bv[ll] = QF[b * VQM_SB + ll * VQM_SL + mq * VQM_SQ + rt * 16 + mi];
                const float ws = WS[b * VQM_ROWS + rt * 16 + mi];
                const v2f_q ws2 = {ws, ws};
                // One MFMA, then its 8 packed VALU instructions, tile after tile.  Issuing the MFMAs ahead of the VALU chain (two or
                // three P buffers, VALU chain pinned in inline assembly) bought nothing — 1380 vs 1290 us for the 8B gate GEMM:
                // on gfx950 an MFMA (f32 or bf16) and f32 VALU work of the same SIMD do not overlap, not even between two
                // wavefronts (scripts/probes/mfma_overlap_probe.hip: 4 MFMA 123 ns + 32 v_pk_fma_f32 153 ns = 258 ns together), so
                // the matrix core only REPLACES the 8 product instructions of a tile at the same cost (32 cycles) — the gain
                // over gemm_vlq_kernel comes from the operand traffic and the packed adds, not from a second pipe.
#pragma unroll
                for (int ll = 0; ll < 8; ++ll)
#pragma unroll
                    for (int tt = 0; tt < 2; ++tt) {
                        const v16f_vl z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                        const v16f_vl P = __builtin_amdgcn_mfma_f32_16x16x1f32(av[tt][ll], bv[ll], z, 0, 0, 0);
#pragma unroll
                        for (int h = 0; h < 2; ++h) {
                            const v2f_q p0 = {P[2 * h], P[2 * h + 1]}, p1 = {P[4 + 2 * h], P[5 + 2 * h]};
                            const v2f_q p2 = {P[8 + 2 * h], P[9 + 2 * h]}, p3 = {P[12 + 2 * h], P[13 + 2 * h]};
                            const v2f_q sm2 = ((p0 + p1) + p2) + p3;                                   // sum0.add(sum1).add(sum2).add(sum3)
                            val[tt][ll][h] = __builtin_elementwise_fma(sm2, ws2, val[tt][ll][h]);     // .fma(wScale, val)
                        }
                    }
            }
        }
        if ((sn % SPC) == 0) { cq = wq; cs = wsc; }
        VQM_LSTORE((s + 1) & 1, sn, cq, cs);
        __syncthreads();
    }
#undef VQM_WLOAD
#undef VQM_XLOAD
#undef VQM_LSTORE
    // ---- reduceLanes(ADD) in lane order from 0, then the epilogue: lane = row, register = token
    const int row = row0 + rt * 16 + mi;
    if (!live || row >= a.rows) return;
#pragma unroll
    for (int tt = 0; tt < 2; ++tt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int b = tok0 + th * 32 + tt * 16 + 4 * mq + r;
            if (b >= a.ntok) continue;
            float v = 0.f;
#pragma unroll
            for (int ll = 0; ll < 8; ++ll) v = v + val[tt][ll][r >> 1][r & 1];
            float* o = a.out + (size_t)b * a.out_stride + row;
            if (EPI == EPI_RESID) *o = *o + v * a.out_scale;
            else *o = v * a.out_scale;
        }
}

// hb = silu(gate) * up per element (InferenceCore.java:155-158: exp in double), in place on the gate buffer
static __global__ __launch_bounds__(256) void pf_swiglu_kernel(float* __restrict__ g, const float* __restrict__ u, size_t n) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float v = g[i];
    v = v / (float)(1.0 + exp(-(double)v));
    g[i] = v * u[i];
}

// token_embedding_table.copyTo per token of the batch (batchForwardJavaPrefill :96), VL layouts
template <int WT>
static __global__ __launch_bounds__(256) void pf_embed_vl_kernel(const uint8_t* __restrict__ emb, int dim, const int32_t* __restrict__ tokens,
                                                                 float* __restrict__ X, float emb_scale, int cc) {
    // X is rank-chunked [dim / cc][ntok][cc] (cc = dim: the plain [ntok][dim]); element i of token b at ((i / cc) ntok + b) cc + i % cc
    const int token = tokens[blockIdx.x], g = token >> 3, rr = token & 7;
    const uint8_t* gb = emb + (size_t)g * vl_group_bytes(WT, dim);
    const int bt = blockIdx.x, nt = gridDim.x;
#define x(I_) X[((size_t)((I_) / cc) * nt + bt) * cc + (I_) % cc]
    for (int i = threadIdx.x; i < dim; i += 256) {
        if (WT == WT_F16) {
            const int c = i >> 6, e = i & 63, l = e & 7, kk = e >> 3;
            x(i) = h2f(reinterpret_cast<const uint16_t*>(gb + (size_t)c * 1024 + (rr * 8 + l) * 16)[kk]) * emb_scale;
        } else if (WT == WT_Q8_0) {
            const int b = i >> 5, j = i & 31, c = b >> 2, kk = b & 3, l = j & 7;
            const uint8_t* cb = gb + (size_t)c * 1088;
            const int q = (int8_t)cb[(rr * 8 + l) * 16 + 4 * kk + (j >> 3)];
            x(i) = ((float)q * h2f(reinterpret_cast<const uint16_t*>(cb + 1024 + rr * 8)[kk])) * emb_scale;
        } else {
            const int b = i >> 5, j = i & 31, c = b >> 3, kk = b & 7, l = j & 7;
            const uint8_t* cb = gb + (size_t)c * 1152;
            const uint8_t byte = cb[(rr * 8 + l) * 16 + ((j & 8) ? 8 : 0) + kk];
            const int q = j < 16 ? (byte & 0x0F) : (byte >> 4);
            x(i) = ((float)(q - 8) * h2f(reinterpret_cast<const uint16_t*>(cb + 1024 + rr * 16)[kk])) * emb_scale;
        }
    }
#undef x
}

}  // namespace gl3
