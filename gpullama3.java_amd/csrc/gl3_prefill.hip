// gl3_prefill.hip — batched prefill (gl3_forward_prefill with max_batch > 1) and static-batched decode
// (gl3_forward_decode_batch); tensor-parallel ranks keep gathered activations rank-chunked (see `chunked`).
//
// Replaces the reference's batched-prefill task graphs
//   J/tornadovm/layers/type/q8_0/prefill/LlamaQ8_0LayersBatchPrefillMMA.java:84-219 (tensor-core path, CUDA only) and
//   ...LlamaQ8_0LayersBatchPrefill.java (scalar path), kernels in J/tornadovm/kernels/TransformerBatchPrefillKernels.java
// but computes what the CPU path computes, bit for bit: InferenceCoreBatchPrefillDecode.batchForwardJavaPrefill
// (J/inference/InferenceCoreBatchPrefillDecode.java:62-168) = per token exactly forwardJava without the logits.
// The reference's MMA path is W8A16 (f16 activations); the CPU oracle is W8A8 with per-32-block int8 activations
// (Q8_0FloatTensor.java:90-123), and that is what the GEMM below does on CDNA4's int8 matrix cores:
//   one v_mfma_i32_32x32x32_i8 = the int32 dot of one Q8_0 block for a 32-row x 32-token tile (exact), then
//   acc = acc + float(isum) * (wScale * aScale) on the VALU, blocks ascending — the reference's f32 order.
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "gl3_ctx.h"
#include <type_traits>
#include "gl3_decode_kernels.h"

#include "gl3_prefill_vl.h"

using namespace gl3;
#include "gl3_bd_gemm.h"      // GemmArgs, bdw_gemm_kernel (expects the gl3 names in scope)
#include "gl3_bdk_gemm.h"     // r6: the same GEMM with K split over producer wavefronts + an ordered chain wavefront
// gl3_prefill_gemm2.hip (own translation unit, -fno-slp-vectorize): the > 64-token GEMM with the scale products on the matrix pipe (r4)
void gl3_gemm2_launch(int epi, const GemmArgs& a, int rows, int ntok, int mode, hipStream_t s);
hipError_t gl3_gemm2_allow_lds();
// r6: the same arithmetic behind a mid-stage barrier / partial-vmcnt operand ring, every GEMM class (gl3_prefill_gemm3.h)
void gl3_gemm3_launch(int epi, const GemmArgs& a, int rows, int ntok, hipStream_t s);
bool gl3_gemm3_swiglu_quantises(int rows, int ntok);
hipError_t gl3_gemm3_allow_lds();

struct gl3_prefill_state {
    int max_batch = 0;
    int32_t* tokens = nullptr;          // [M]
    float* X = nullptr;                 // [M][dim] residual stream (rank-chunked [tp][n][dim/tp] under tensor parallelism)
    uint8_t* XQ = nullptr;              // [M][maxk] int8 activations
    float* XS = nullptr;                // [M][maxk/32] activation scales
    uint8_t* XP = nullptr;              // [maxk/32 + 4][2 lane halves][xp_tok][16 B] the activation scales as bf16 MFMA operands (pf_gemm3_kernel)
    int xp_tok = 0;                     //   token slots per block: max_batch rounded up to the GEMM's 128-token tile
    uint8_t* XQh = nullptr;             // second operand set of the > 64-token path: hb quantised by the gate + up GEMM's own epilogue (pf_gemm3t_kernel<.., QOUT>)
    uint8_t* XPh = nullptr;             //   while other workgroups still read XQ / XP
    uint8_t* XQb = nullptr;             // second small-batch operand buffer: hb quantised by the gate/up kernel's own epilogue
    float* XSb = nullptr;               //   (its input still being read by other workgroups)
    float* QKV = nullptr;               // [M][q_dim + 2 kv_dim]
    float* AO = nullptr;                // [M][q_dim] attention output (rank-chunked)
    float* HB = nullptr;                // [M][hidden] (rank-chunked)
    float* ATT = nullptr;               // [M][n_heads][ctx] scores
    float* TMX = nullptr;               // [M][n_heads][tmx_tiles] per-64-timestep-tile maxima of the score rows (pf_scores_tiled_kernel -> pf_softmax_rows_kernel)
    float* SUMS = nullptr;              // [M][n_heads] softmax denominators (pf_softmax_rows_kernel -> pf_pv_tiled_kernel)
    int tmx_tiles = 0;
    int32_t* seqpos = nullptr;          // [2][M]: sequence id, position of every token of the step
    float* LOGITS = nullptr;            // [rows][vocab], grown on demand (batched decode)
    int logits_rows = 0;
    std::vector<hipGraphExec_t> step_graphs;   // static-batched decode: one captured step per batch size (positions < AF_MAXN)
    bool in_arena = false;              // X / AO / HB / LOGITS are slices of the tensor-parallel arena (not freed here)
    int32_t* amax = nullptr;            // [M]
    float* amx_v = nullptr;             // [M][AMX_SPLIT] partial maxima of the greedy scan
    int* amx_i = nullptr;
    int maxk = 0;
    // F16 / Q4_0 / Q8_0-with-f32-activation plans (gl3_prefill_vl.h): the GEMMs read f32 activations
    bool vl = false;
    float* XN = nullptr;                // [M][dim] RMS-normalised activations
    float* HB2 = nullptr;               // [M][hidden] up projection (hb = silu(HB) * HB2)
};


// ---------------------------------------------------------------------------------------------------
// token_embedding_table.copyTo per token (batchForwardJavaPrefill :96)
// Activation layout under tensor parallelism ("rank-chunked"): a [ntok][cols] activation that is produced by row-split
// matrices is kept as [tp][ntok][cols / tp], so every rank's output is one contiguous chunk and the all-gather is in
// place; element j of token b sits at (j / cc) * ntok * cc + b * cc + j % cc with cc = cols / tp (tp = 1: the plain layout).
__device__ __forceinline__ size_t chunked(int b, int j, int cc, int ntok) { return ((size_t)(j / cc) * ntok + b) * cc + (j % cc); }

__global__ __launch_bounds__(256) void pf_embed_kernel(const uint8_t* __restrict__ emb, int ng, int dim,
                                                        const int32_t* __restrict__ tokens, float* __restrict__ X, int cc, float emb_scale) {
    const int token = tokens[blockIdx.x];
    const uint8_t* strip = emb + (size_t)(token >> 4) * ng * TILE_BYTES;
    const int i16 = token & 15;
    const int bt = blockIdx.x, nt = gridDim.x;
    for (int i = threadIdx.x; i < dim; i += 256) {
        const int b = i >> 5, j = i & 31;
        const uint8_t* p = strip + (size_t)(b >> 2) * TILE_BYTES;
        const int l = i16 + 16 * (b & 3);
        const float d = h2f(*reinterpret_cast<const uint16_t*>(p + 2 * l));
        const int8_t q = (int8_t)p[(j < 16 ? 128 : 1152) + 16 * l + (j & 15)];
        X[chunked(bt, i, cc, nt)] = ((float)q * d) * emb_scale;
    }
}

// ---------------------------------------------------------------------------------------------------
// Per token: (RMSNorm with the exact in-order sum of squares) + Q8_0 activation quantisation.
// One workgroup of 256 threads per token (PQ_NORM) or per (token, 1024-element chunk) (the other modes: the blocks are
// independent, and one workgroup per token left 32 tokens on 32 CUs).
//   PQ_PLAIN:  quantise an f32 row;  PQ_NORM: RMSNorm, then quantise
//   PQ_NORM_F32: RMSNorm only, f32 out (XS = [ntok][k] floats; the f32-activation weight types, gl3_prefill_vl.h)
enum { PQ_PLAIN = 0, PQ_NORM = 1, PQ_NORM_F32 = 2, PQ_PLAIN_F32 = 3 };     // PQ_PLAIN_F32: rank-chunked f32 row -> plain f32 row (XS)
template <int MODE>
__global__ __launch_bounds__(256) void pf_norm_quant_kernel(const float* __restrict__ in, int k, int in_stride,
                                                             const float* __restrict__ norm_w, float eps,
                                                             uint8_t* __restrict__ XQ, float* __restrict__ XS, int maxk, int tslots,
                                                             uint2* __restrict__ XP = nullptr, int xp_tok = 0) {
    constexpr bool NORM = MODE == PQ_NORM || MODE == PQ_NORM_F32;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    float* xf = reinterpret_cast<float*>(smem);                 // [k + 32]
    uint8_t* scratch = smem + (size_t)(k + 32) * 4;             // ss_scratch_bytes(k)
    float* red = reinterpret_cast<float*>(scratch + ss_scratch_bytes(k));
    const int t = threadIdx.x, b = blockIdx.x;
    const int cc = in_stride, nt = gridDim.x;                    // in_stride = chunk columns (k / tp); cc % 4 == 0
    auto xquad = [&](int qd) { return *reinterpret_cast<const float4*>(in + chunked(b, 4 * qd, cc, nt)); };
    const int nquads = k >> 2;
    float scale = 1.0f;
    if (NORM) {
        for (int qd = t; qd < nquads; qd += 256)
            *reinterpret_cast<float4*>(xf + 4 * qd) = xquad(qd);
        if (t < 32) xf[k + t] = 0.f;
        __syncthreads();
        float ss;
        if (k >= 1024 && k <= 5120) {
            BlockBarrier bb;
            ss = exact_sumsq_lds(xf, k, scratch, t, bb);
        } else {
            if (t < 64) { const float s1 = seq_sum_lds<true>(xf, k); if (t == 0) red[0] = s1; }
            __syncthreads();
            ss = red[0];
        }
        ss /= (float)k;
        ss += eps;
        scale = (float)(1.0 / sqrt((double)ss));
    }
    uint8_t* xq = XQ + (size_t)b * maxk;
    float* xs = XS + (size_t)b * (maxk >> 5);
    for (int qd = t + 256 * blockIdx.y; qd < nquads; qd += 256 * gridDim.y) {
        float4 v;
        if (NORM) {
            v = *reinterpret_cast<const float4*>(xf + 4 * qd);
            const float4 w = *reinterpret_cast<const float4*>(norm_w + 4 * qd);
            v.x = w.x * (scale * v.x); v.y = w.y * (scale * v.y); v.z = w.z * (scale * v.z); v.w = w.w * (scale * v.w);
        } else {
            v = xquad(qd);
        }
        if (MODE == PQ_NORM_F32 || MODE == PQ_PLAIN_F32) { *reinterpret_cast<float4*>(XS + (size_t)b * k + 4 * qd) = v; continue; }
        if (tslots == 0) {
            float qs;
            const uint32_t packed = quantize_quad_pack(v, qs);
            // XP set (pf_gemm3_kernel): chunk-major int8 operand XQ3[k / 16][xp_tok token slots][16 B]; otherwise the row layout XQ[token][k]
            if (XP) *reinterpret_cast<uint32_t*>(XQ + ((size_t)(qd >> 2) * xp_tok + b) * 16 + 4 * (qd & 3)) = packed;
            else *reinterpret_cast<uint32_t*>(xq + 4 * qd) = packed;
            if ((qd & 7) == 0) {
                xs[qd >> 3] = qs;
                if (XP) {
                    // the scale as the bf16 operands the s / -B s MFMAs read (gl3_prefill_gemm3.h): P = {a_hi, a_lo} (hi = top 8 significand bits, lo = the
                    // rest: exact, qs is an f16 value), Q = P * -2^23 for the k slots of lane half 0, P * -2^22 for half 1 (their sum is -B = -3 * 2^22)
                    const float ahi = __uint_as_float(__float_as_uint(qs) & 0xFFFF0000u), alo = qs - ahi;
                    auto pk = [](float h, float l) { return (__float_as_uint(h) >> 16) | (__float_as_uint(l) & 0xFFFF0000u); };
                    const uint32_t pr = pk(ahi, alo), q0 = pk(ahi * -8388608.f, alo * -8388608.f), q1 = pk(ahi * -4194304.f, alo * -4194304.f);
                    const int blk = qd >> 3;
                    uint4* xp = reinterpret_cast<uint4*>(XP);      // XP[block][half][xp_tok][16 B]
                    xp[((size_t)blk * 2 + 0) * xp_tok + b] = make_uint4(pr, pr, q0, q0);
                    xp[((size_t)blk * 2 + 1) * xp_tok + b] = make_uint4(0u, 0u, q1, q1);
                    // ragged K: the padded blocks of the last tile group carry zero weights; give them zero activation operands too
                    if (blk == (k >> 5) - 1)
                        for (int pb = blk + 1; pb < ((blk + 4) & ~3); ++pb) {
                            xp[((size_t)pb * 2 + 0) * xp_tok + b] = make_uint4(0u, 0u, 0u, 0u);
                            xp[((size_t)pb * 2 + 1) * xp_tok + b] = make_uint4(0u, 0u, 0u, 0u);
                        }
                }
            }
        } else {                                           // the wave-owned small-batch GEMM's operand layout (gl3_bd_gemm.h)
            float qs;
            const uint32_t packed = quantize_quad_pack(v, qs);
            *reinterpret_cast<uint32_t*>(XQ + bdq_offset(qd, b, tslots)) = packed;
            if ((qd & 7) == 0) XS[bds_offset(qd >> 3, b, tslots)] = qs;
        }
    }
}


// LDS-tiled version: workgroup = 128 weight rows (64 gate + 64 up rows for the SwiGLU epilogue) x 128 tokens, 4
// wavefronts in a 2 x 2 grid, each owning 64 rows x 64 tokens = four 32x32 int8 MFMA tiles.  K advances 4 blocks
// (= one Q8T tile per 16-row strip) per stage; the next stage's operands travel HBM/L2 -> registers while the current
// stage is consumed from LDS (double buffer).  LDS image per stage (36 KB):
//   Aq[blk][half][128 rows][16 B] | As[blk][128 rows] f32 | Bq[blk][half][128 tokens][16 B] | Bs[blk][128 tokens] f32
// so an MFMA fragment is one conflict-free ds_read_b128 and the 16 weight scales of a lane are four broadcast b128 reads.
// RF = 32-row fragments per wavefront (2: 128-row workgroup tile for the wide matrices; 1: 64-row tile so that the
// 4096-row wo / down projections still launch >= 256 workgroups at 512 tokens).  The SwiGLU epilogue always runs
// RF = 1 over two matrices.
constexpr int GM_TOK = 128, GM_KB = 4;
typedef float v16f_t __attribute__((ext_vector_type(16)));
__host__ __device__ constexpr int gm_stage_bytes(int arows, int toks = 128) {
    return GM_KB * 2 * arows * 16 + GM_KB * arows * 4 + GM_KB * 2 * toks * 16 + GM_KB * toks * 4;
}

// NW = wavefronts per workgroup: 4 (2 x 2, two token fragments each) or 8 (2 x 4, one token fragment each; used for the
// 4096-row matrices where only one workgroup fits a CU, so that every SIMD still interleaves two wavefronts).
// TOK = tokens per workgroup: 128, or 32 for static-batched decode (few tokens: four wavefronts side by side along the
// rows, one token fragment each, so that a batch of 32 does not pay for 128 padded columns).
template <int EPI, int RF, int NW, int TOK = GM_TOK>
__global__ __launch_bounds__(64 * NW, NW == 4 ? 2 : 1) void pf_gemm_kernel(const GemmArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    constexpr int NM = (EPI == EPI_SWIGLU) ? 2 : 1;
    static_assert(NM * RF <= 2, "accumulator budget");
    static_assert(NW == 4 || (NW == 8 && NM * RF == 1), "8-wavefront layout is for the single-fragment variant");
    static_assert(TOK == GM_TOK || (TOK == 32 && NW == 4 && RF == 1), "32-token tiles: 4 wavefronts x one fragment");
    constexpr int NT = 64 * NW;                        // threads
    constexpr int TF = TOK == 32 ? 1 : 8 / NW;         // 32-token fragments per wavefront
    constexpr int WR = TOK == 32 ? 4 : 2;              // wavefronts along the rows
    constexpr int AROWS = NM * RF * 32 * WR;           // weight rows staged per K stage (both matrices together)
    constexpr int RPM = AROWS / NM;                    // output rows per matrix covered by this workgroup
    constexpr int STAGE = gm_stage_bytes(AROWS, TOK);
    constexpr int NAP = (AROWS * 4 + NT - 1) / NT;     // (strip, lane) pairs per thread
    constexpr int NBP = (TOK * 8 + NT - 1) / NT;       // 16-byte activation pieces per thread
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int tl = lane & 31, hi = lane >> 5;
    const int wr = TOK == 32 ? wave : NW == 4 ? wave >> 1 : wave >> 2;    // wavefront grid: row part wr,
    const int wc = TOK == 32 ? 0 : NW == 4 ? wave & 1 : wave & 3;         // tokens wc * 32 * TF ..
    // XCD-aware tile mapping: workgroups are dealt round-robin to the 8 XCDs (private L2 each), so the token tiles that
    // share a weight row tile are given consecutive slots of ONE XCD — the weights cross the fabric once, not once per
    // token tile (rocprofv3 FETCH_SIZE of the gate/up GEMM at 512 tokens: 517 MB -> see profiles/).
    const int ntt_g = a.ntt, per_xcd = (a.ntt * a.nrt + 7) >> 3;
    const int lin = blockIdx.x, J = (lin & 7) * per_xcd + (lin >> 3);
    if (J >= ntt_g * a.nrt) return;
    const int row0 = (J / ntt_g) * RPM;                // first output row (per matrix)
    const int tok0 = (J % ntt_g) * TOK;
    const size_t strip_bytes = (size_t)a.ng * TILE_BYTES;
    const int nkb = a.ng;                              // K stages = tile groups per strip
    const int nstrips = (a.rows + 15) >> 4;

    // ---- global -> register staging of one K stage
    // A: AROWS/16 strips x one tile (64 x f16 scales, 64 x 16 B lo, 64 x 16 B hi).  Thread t: lo/hi pieces
    // t + 256 i of the (strip, lane-in-tile) pairs; scales: threads < AROWS/2 take 8 f16 = (strip t>>3, lanes 8*(t&7)..+7)
    v4i_t ra_lo[NAP], ra_hi[NAP], ra_sc, rb[NBP];
    float4 rb_s;
    auto tile_of = [&](int sl, int kb) -> const uint8_t* {   // sl: local strip 0..AROWS/16-1
        constexpr int SPM = RPM / 16;                      // strips per matrix in this workgroup
        const int m = NM == 2 ? (sl / SPM) : 0;
        const int strip = min(nstrips - 1, (row0 >> 4) + (NM == 2 ? (sl % SPM) : sl));
        return (m == 0 ? a.w : a.w2) + (size_t)strip * strip_bytes + (size_t)kb * TILE_BYTES;
    };
    auto gload = [&](int kb) {
#pragma unroll
        for (int i = 0; i < NAP; ++i) {
            const int pr = t + NT * i, lt = pr & 63;
            if (pr < AROWS * 4) {
                const uint8_t* tile = tile_of(pr >> 6, kb);
                ra_lo[i] = *reinterpret_cast<const v4i_t*>(tile + 128 + 16 * lt);
                ra_hi[i] = *reinterpret_cast<const v4i_t*>(tile + 1152 + 16 * lt);
            }
        }
        if (t < AROWS / 2) ra_sc = *reinterpret_cast<const v4i_t*>(tile_of(t >> 3, kb) + 16 * (t & 7));
        // B: 128 tokens x 128 B of int8 (4 blocks) -> 1024 16-byte pieces, 4 per thread: piece = t + 256*i ->
        // token = piece >> 3, 16-byte chunk c = piece & 7 (block c>>1, half c&1)
#pragma unroll
        for (int i = 0; i < NBP; ++i) {
            const int pc = t + NT * i, tk = min(a.ntok - 1, tok0 + (pc >> 3)), c = pc & 7;
            if (pc < TOK * 8) rb[i] = *reinterpret_cast<const v4i_t*>(a.XQ + (size_t)tk * a.maxk + (size_t)kb * 128 + 16 * c);
        }
        if (t < TOK) {
            const int tk = min(a.ntok - 1, tok0 + t);
            rb_s = *reinterpret_cast<const float4*>(a.XS + (size_t)tk * (a.maxk >> 5) + kb * 4);
        }
    };
    auto lstore = [&](int stage, int kbs) {        // kbs = K stage held in the staging registers
        uint8_t* base = smem + (size_t)stage * STAGE;
        uint8_t* Aq = base;
        float* As = reinterpret_cast<float*>(base + GM_KB * 2 * AROWS * 16);
        uint8_t* Bq = base + GM_KB * 2 * AROWS * 16 + GM_KB * AROWS * 4;
        float* Bs = reinterpret_cast<float*>(Bq + GM_KB * 2 * TOK * 16);
#pragma unroll
        for (int i = 0; i < NAP; ++i) {
            const int pr = t + NT * i, sl = pr >> 6, lt = pr & 63;
            const int row = sl * 16 + (lt & 15), blk = lt >> 4;
            if (pr < AROWS * 4) {
                *reinterpret_cast<v4i_t*>(Aq + ((size_t)(blk * 2 + 0) * AROWS + row) * 16) = ra_lo[i];
                *reinterpret_cast<v4i_t*>(Aq + ((size_t)(blk * 2 + 1) * AROWS + row) * 16) = ra_hi[i];
            }
        }
        if (t < AROWS / 2) {
            const int sl = t >> 3;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int lt = 8 * (t & 7) + j;
                const uint32_t w = (uint32_t)ra_sc[j >> 1];
                As[(lt >> 4) * AROWS + sl * 16 + (lt & 15)] = h2f((uint16_t)((j & 1) ? (w >> 16) : (w & 0xFFFF)));
            }
        }
#pragma unroll
        for (int i = 0; i < NBP; ++i) {
            const int pc = t + NT * i, tk = pc >> 3, c = pc & 7;
            if (pc < TOK * 8) *reinterpret_cast<v4i_t*>(Bq + ((size_t)c * TOK + (tk ^ c)) * 16) = rb[i]; // c = blk*2 + half; xor: bank spread
        }
        if (t < TOK) {
            // ragged K (k % 128 != 0): the padded blocks carry zero weights; zero their activation scale too.  (Done
            // here, not at load time, so that the global loads stay in flight across the compute phase.)
            Bs[0 * TOK + t] = rb_s.x;
            Bs[1 * TOK + t] = kbs * 4 + 1 < a.nb ? rb_s.y : 0.f;
            Bs[2 * TOK + t] = kbs * 4 + 2 < a.nb ? rb_s.z : 0.f;
            Bs[3 * TOK + t] = kbs * 4 + 3 < a.nb ? rb_s.w : 0.f;
        }
    };

    // accumulators: [fragment f][token frag][8 x 2]; fragment f = matrix (SwiGLU) or row fragment
    constexpr int NF = NM * RF;
    v2f_t acc[NF][TF][8];
#pragma unroll
    for (int i = 0; i < NF; ++i)
#pragma unroll
        for (int j = 0; j < TF; ++j)
#pragma unroll
            for (int r = 0; r < 8; ++r) acc[i][j][r] = v2f_t{0.f, 0.f};

#ifdef GL3_GEMM_TIMING
    const unsigned long long tk0 = __builtin_readcyclecounter(), rt0 = __builtin_amdgcn_s_memrealtime();
#endif
    v16i_t cbias;
#pragma unroll
    for (int r = 0; r < 16; ++r) cbias[r] = 0x4B400000;
    if constexpr (NM * RF == 1) asm volatile("" : "+v"(cbias));   // keep the splat in VGPRs (else 8 v_mov_b64 per MFMA pair)
    gload(0);
    lstore(0, 0);
    __syncthreads();
    for (int kb = 0; kb < nkb; ++kb) {
        if (kb + 1 < nkb) gload(kb + 1);
        const uint8_t* base = smem + (size_t)(kb & 1) * STAGE;
        const uint8_t* Aq = base;
        const float* As = reinterpret_cast<const float*>(base + GM_KB * 2 * AROWS * 16);
        const uint8_t* Bq = base + GM_KB * 2 * AROWS * 16 + GM_KB * AROWS * 4;
        const float* Bs = reinterpret_cast<const float*>(Bq + GM_KB * 2 * TOK * 16);
        // operand fragments of one Q8_0 block for this wavefront
        // MFMA_SCALES (single-fragment variants: wo / down at 512 tokens): the scale products wScale * aScale of the 32 x 32 tile
        // come from the MATRIX pipe: one v_mfma_f32_32x32x2_f32 with A = wScale (k = 0 | 0), B = aScale (k = 0 | 0), C = 0 gives
        // D[i][j] = fl(wScale_i * aScale_j) in the layout of the int8 tile — the f32 MFMA rounds once per step like fmaf
        // (MI355X_MICROARCH.md), fma(0, 0, fl(w a)) = fl(w a), and both scales are >= +0 so no -0 arises.  The VALU keeps three
        // operations per output and block (subtract, multiply, add) instead of four: down 145 -> 132 us, wo 46 -> 43 us at 512
        // tokens.  With two or four fragments per wavefront the extra 64-cycle MFMAs sit in front of the dependent VALU work and
        // the same change LOST time (gate/up 236 -> 275 us, qkv 73 -> 79 us), so those variants multiply on the VALU.
        constexpr bool MFMA_SCALES = NF * TF == 1;
        struct Frag { v4i_t bf[TF]; v2f_t xsc[TF]; v4i_t af[NF]; v2f_t wsf[NF][MFMA_SCALES ? 1 : 8]; };
        auto fload_a = [&](Frag& fr, int blk, int f) {
            const int lrow = NM == 2 ? f * RPM + wr * 32 : wr * (32 * RF) + f * 32;    // local row of this fragment
            fr.af[f] = *reinterpret_cast<const v4i_t*>(Aq + ((size_t)(blk * 2 + hi) * AROWS + lrow + tl) * 16);
            if constexpr (MFMA_SCALES) {
                const float w = As[blk * AROWS + lrow + tl];
                fr.wsf[f][0] = v2f_t{hi ? 0.f : w, 0.f};        // A operand: lane = row in k = 0, zeros in k = 1
            } else {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 w4 = *reinterpret_cast<const float4*>(As + blk * AROWS + lrow + 8 * q + 4 * hi);
                    fr.wsf[f][2 * q] = v2f_t{w4.x, w4.y};
                    fr.wsf[f][2 * q + 1] = v2f_t{w4.z, w4.w};
                }
            }
        };
        auto fload = [&](Frag& fr, int blk, bool with_a) {
#pragma unroll
            for (int tf = 0; tf < TF; ++tf) {
                const int tk = wc * (32 * TF) + tf * 32 + tl;
                fr.bf[tf] = *reinterpret_cast<const v4i_t*>(Bq + ((size_t)(blk * 2 + hi) * TOK + (tk ^ (blk * 2 + hi))) * 16);
                const float x = Bs[blk * TOK + tk];
                fr.xsc[tf] = MFMA_SCALES ? v2f_t{hi ? 0.f : x, 0.f} : v2f_t{x, x};
            }
            if (with_a) {
#pragma unroll
                for (int f = 0; f < NF; ++f) fload_a(fr, blk, f);
            }
        };
        // The int32 block sums come out of the MFMA already biased by 0x4B400000: reinterpreted as f32 that is
        // 12582912 + isum exactly (|isum| <= 32*127*127 < 2^22), so (float)isum = bits - 12582912.0f is one packed
        // subtract per two values instead of two v_cvt_f32_i32.
        auto fcompute = [&](Frag& fr, int blk, bool load_a) {
#pragma unroll
            for (int f = 0; f < NF; ++f) {
                if (load_a) fload_a(fr, blk, f);
                v16i_t c[TF];
#pragma unroll
                for (int tf = 0; tf < TF; ++tf) c[tf] = __builtin_amdgcn_mfma_i32_32x32x32_i8(fr.af[f], fr.bf[tf], cbias, 0, 0, 0);
#pragma unroll
                for (int tf = 0; tf < TF; ++tf) {
                    v2f_t cf[8], pr[8];
#pragma unroll
                    for (int r = 0; r < 8; ++r)
                        cf[r] = v2f_t{__int_as_float(c[tf][2 * r]), __int_as_float(c[tf][2 * r + 1])} - v2f_t{12582912.f, 12582912.f};
                    if constexpr (MFMA_SCALES) {
                        const v16f_t zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                        const v16f_t p16 = __builtin_amdgcn_mfma_f32_32x32x2f32(fr.wsf[f][0][0], fr.xsc[tf][0], zero16, 0, 0, 0);
#pragma unroll
                        for (int r = 0; r < 8; ++r) pr[r] = v2f_t{p16[2 * r], p16[2 * r + 1]};
                    } else {
#pragma unroll
                        for (int r = 0; r < 8; ++r) pr[r] = fr.wsf[f][r] * fr.xsc[tf];
                    }
#pragma unroll
                    for (int r = 0; r < 8; ++r) cf[r] = cf[r] * pr[r];       // isum * (wScale * aScale)
#pragma unroll
                    for (int r = 0; r < 8; ++r) acc[f][tf][r] = acc[f][tf][r] + cf[r];   // result +=, blocks ascending
                }
            }
        };
        if constexpr (NF == 1) {
            // one wavefront per SIMD in the narrow-matrix launches: fetch block b+1's fragments while block b computes
            Frag fa, fb;
            fload(fa, 0, true);
            if constexpr (TF == 1) {          // small fragments: static registers for the whole stage, no loop-carried copies
#pragma unroll
                for (int blk = 0; blk < GM_KB; blk += 2) {
                    fload(fb, blk + 1, true);
                    fcompute(fa, 0, false);
                    if (blk + 2 < GM_KB) fload(fa, blk + 2, true);
                    fcompute(fb, 0, false);
                }
            } else {
#pragma unroll 1
                for (int blk = 0; blk < GM_KB; blk += 2) {
                    fload(fb, blk + 1, true);
                    fcompute(fa, 0, false);
                    if (blk + 2 < GM_KB) fload(fa, blk + 2, true);
                    fcompute(fb, 0, false);
                }
            }
        } else {
#pragma unroll 1
            for (int blk = 0; blk < GM_KB; ++blk) {
                Frag fr;
                fload(fr, blk, false);
                fcompute(fr, blk, true);
            }
        }
        if (kb + 1 < nkb) lstore((kb + 1) & 1, kb + 1);
        __syncthreads();
    }
#ifdef GL3_GEMM_TIMING
    if (t == 0 && (blockIdx.y % 97) == 0 && blockIdx.x == 0) {
        const unsigned long long tk1 = __builtin_readcyclecounter(), rt1 = __builtin_amdgcn_s_memrealtime();
        printf("gemm epi=%d wg=(%d,%d) nkb=%d ticks=%llu realtime=%llu start_rt=%llu\n", EPI, blockIdx.x, blockIdx.y, nkb, tk1 - tk0, rt1 - rt0, rt0);
    }
#endif
    // ---- epilogue.  C layout: token = lane & 31 (column), weight row = (r & 3) + 8 * (r >> 2) + 4 * hi
#pragma unroll
    for (int tf = 0; tf < TF; ++tf) {
        const int b = tok0 + wc * (32 * TF) + tf * 32 + tl;
        if (b >= a.ntok) continue;
        if (EPI == EPI_SWIGLU) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = row0 + wr * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                if (row >= a.rows) continue;
                float g = acc[0][tf][r >> 1][r & 1];
                g = g / (float)(1.0 + exp(-(double)g));
                a.out[(size_t)b * a.out_stride + row] = g * acc[NF - 1][tf][r >> 1][r & 1];
            }
        } else {
#pragma unroll
            for (int f = 0; f < NF; ++f) {
                // rows (r & 3) + 8 * (r >> 2) + 4 * hi: four runs of four consecutive rows -> float4 accesses
                float* o = a.out + (size_t)b * a.out_stride + row0 + wr * (32 * RF) + f * 32 + 4 * hi;
                const int rbase = row0 + wr * (32 * RF) + f * 32 + 4 * hi;
                float4 old[4];
                if (EPI == EPI_RESID) {
                    // unconditional loads (a guarded load into an array is followed by s_waitcnt vmcnt(0): four serialised round trips
                    // per fragment); rows past the end read the buffer's first element instead and are not used
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        old[q] = *reinterpret_cast<const float4*>(rbase + 8 * q + 3 < a.rows ? o + 8 * q : a.out);
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float4 v = {acc[f][tf][2 * q][0] * a.out_scale, acc[f][tf][2 * q][1] * a.out_scale, acc[f][tf][2 * q + 1][0] * a.out_scale, acc[f][tf][2 * q + 1][1] * a.out_scale};
                    if (rbase + 8 * q + 3 < a.rows) {
                        if (EPI == EPI_RESID) { v.x = old[q].x + v.x; v.y = old[q].y + v.y; v.z = old[q].z + v.z; v.w = old[q].w + v.w; }
                        *reinterpret_cast<float4*>(o + 8 * q) = v;
                    } else {
                        const float vv[4] = {v.x, v.y, v.z, v.w};
                        for (int i = 0; i < 4; ++i)
                            if (rbase + 8 * q + i < a.rows) o[8 * q + i] = EPI == EPI_RESID ? o[8 * q + i] + vv[i] : vv[i];
                    }
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// RoPE on q and k of every token + KV-cache write (batchForwardJavaPrefill :106-121; Qwen3 adds the per-head
// RMSNorm, InferenceCore.java:594-600).  Grid = (n_heads + n_kv_heads, ntok), block = 64.
struct RopeArgs {
    float* QKV; int qkv_stride; float* kcache; float* vcache; const float* cr; const float* ci;
    const float* qnorm; const float* knorm; const float* bq; const float* bk; const float* bv;   // bias: qwen2 (else NULL)
    int n_heads, n_kv_heads, hs, q_dim, kv_dim, arch; float eps;
    const int32_t* seq; const int32_t* pos; size_t seq_stride;   // per-token sequence id / position; floats between sequences' caches
};

__global__ __launch_bounds__(64) void pf_rope_kv_kernel(const RopeArgs a) {
    __shared__ __attribute__((aligned(16))) float v[256];
    const int h = blockIdx.x, b = blockIdx.y, t = threadIdx.x, hs = a.hs;
    const int pos = a.pos[b];
    const size_t soff = (size_t)a.seq[b] * a.seq_stride;
    const bool is_k = h >= a.n_heads;
    const int hk = is_k ? h - a.n_heads : h;
    float* src = a.QKV + (size_t)b * a.qkv_stride + (is_k ? a.q_dim + hk * hs : hk * hs);
    const float* bias = is_k ? a.bk : a.bq;              // qwen2: q / k / v bias before RoPE (InferenceCore.java:456-459)
    for (int i = t; i < hs; i += 64) v[i] = bias ? src[i] + bias[hk * hs + i] : src[i];
    __syncthreads();
    if (a.arch == 1) {
        head_rmsnorm_wave(v, is_k ? a.knorm : a.qnorm, hs, a.eps, t);      // the workgroup is one wavefront
        __syncthreads();
    }
    rope_head(v, hs, a.cr + (size_t)pos * (hs >> 1), a.ci + (size_t)pos * (hs >> 1), a.arch, t, 64);
    __syncthreads();
    if (!is_k) {
        for (int i = t; i < hs; i += 64) src[i] = v[i];
    } else {
        const float* vsrc = a.QKV + (size_t)b * a.qkv_stride + a.q_dim + a.kv_dim + hk * hs;
        for (int i = t; i < hs; i += 64) {
            a.kcache[soff + (size_t)pos * a.kv_dim + hk * hs + i] = v[i];
            a.vcache[soff + (size_t)pos * a.kv_dim + hk * hs + i] = a.bv ? vsrc[i] + a.bv[hk * hs + i] : vsrc[i];
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// Batched attention (batchForwardJavaPrefill :123-145: sequential per token, identical arithmetic to decode).
// Scores: grid = (n_tsplit, n_kv_heads, ntok), block = 64 x kvMul; all K rows come from the cache.
struct PfAttnArgs {
    const float* Q; int q_stride;       // roped q rows [ntok][...]
    const float* kcache; const float* vcache;
    float* att;                          // [ntok][n_heads][ctx]
    float* out; int out_stride;          // [ntok][q_dim]
    int n_heads, n_kv_heads, hs, kv_dim, ctx;
    const int32_t* seq; const int32_t* pos; size_t seq_stride;
    float att_mul;                       // 0: score / sqrt(head_size); Granite: score * attentionScale
    int win;                             // pf_attn_softmax_pv_kernel: floats of a softmax row held in LDS (longer rows: windows)
};

__global__ void pf_attn_scores_kernel(const PfAttnArgs a) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int hs = a.hs, kvmul = a.n_heads / a.n_kv_heads, pitch = hs + 1;
    float* q_s = sm;
    float* kt = q_s + kvmul * hs;
    const int t = threadIdx.x, nthr = blockDim.x;
    const int sp = blockIdx.x, kvh = blockIdx.y, b = blockIdx.z;
    const int pos = a.pos[b];
    const float* kc = a.kcache + (size_t)a.seq[b] * a.seq_stride;
    const int t0 = sp * ATT_TT;
    if (t0 > pos) return;
    const int t1 = min(pos + 1, t0 + ATT_TT);
    for (int i = t; i < kvmul * hs; i += nthr) q_s[i] = a.Q[(size_t)b * a.q_stride + (kvh * kvmul) * hs + i];
    const int q4 = hs >> 2;
    for (int i = t; i < (t1 - t0) * q4; i += nthr) {
        const int r = i / q4, c = i % q4;
        const float4 v = *reinterpret_cast<const float4*>(kc + (size_t)(t0 + r) * a.kv_dim + kvh * hs + 4 * c);
        float* d = kt + r * pitch + 4 * c;
        d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
    }
    __syncthreads();
    const int hq = t >> 6, r = t & 63;
    if (hq < kvmul && t0 + r < t1) {
        const float* q = q_s + hq * hs;
        const float* kk = kt + r * pitch;
        float score = 0.f;
        for (int j = 0; j < hs; ++j) score = score + q[j] * kk[j];
        const float sqrt_hs = (float)sqrt((double)hs);
        a.att[((size_t)b * a.n_heads + kvh * kvmul + hq) * a.ctx + t0 + r] = a.att_mul != 0.f ? score * a.att_mul : score / sqrt_hs;
    }
}

// softmax + weighted V sum: grid = (n_heads * ceil(hs/64), ntok), block = 64.  The row sits in LDS when it fits the window
// (a.win floats); longer rows (contexts beyond ~16 k positions) run in windows: the sequential sum carries its running value
// across them and the numerators are recomputed per window for the weighted V sum (same exp of the same argument -> same bits).
__global__ __launch_bounds__(64) void pf_attn_softmax_pv_kernel(const PfAttnArgs a) {
    extern __shared__ __attribute__((aligned(16))) float e_s[];
    const int hs = a.hs, kvmul = a.n_heads / a.n_kv_heads;
    const int nj = (hs + 63) / 64;
    const int h = blockIdx.x / nj, j = (blockIdx.x % nj) * 64 + threadIdx.x, b = blockIdx.y;
    const int lane = threadIdx.x, kvh = h / kvmul;
    const int n = a.pos[b] + 1, W = a.win;
    const float* sc = a.att + ((size_t)b * a.n_heads + h) * a.ctx;
    float mx = -INFINITY;
    if (n <= W) { for (int i = lane; i < n; i += 64) { const float s = sc[i]; e_s[i] = s; mx = fmaxf(mx, s); } }
    else { for (int i = lane; i < n; i += 64) mx = fmaxf(mx, sc[i]); }
    mx = wave_max(mx);
    __syncthreads();
    float sum = 0.f;
    if (n <= W) {
        for (int i = lane; i < n; i += 64) e_s[i] = (float)exp((double)(e_s[i] - mx));
        __syncthreads();
        sum = seq_sum_lds<false>(e_s, n);
        __syncthreads();
        for (int i = lane; i < n; i += 64) e_s[i] = e_s[i] / sum;
        __syncthreads();
    } else {
        for (int c0 = 0; c0 < n; c0 += W) {
            const int len = min(W, n - c0);
            __syncthreads();
            for (int i = lane; i < len; i += 64) e_s[i] = (float)exp((double)(sc[c0 + i] - mx));
            __syncthreads();
            sum = seq_sum_lds<false>(e_s, len, sum);
        }
    }
    const float* v = a.vcache + (size_t)a.seq[b] * a.seq_stride + kvh * hs + min(j, hs - 1);
    float acc = 0.f;
    for (int c0 = 0; c0 < n; c0 += W) {                 // one trip unless the row is longer than the window
        const int clen = min(W, n - c0);
        if (n > W) {
            __syncthreads();
            for (int i = lane; i < clen; i += 64) e_s[i] = (float)exp((double)(sc[c0 + i] - mx)) / sum;
            __syncthreads();
        }
        const float* vw = v + (size_t)c0 * a.kv_dim;
        int tt = 0;
        for (; tt + 8 <= clen; tt += 8) {
            float vv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) vv[u] = vw[(size_t)(tt + u) * a.kv_dim];
#pragma unroll
            for (int u = 0; u < 8; ++u) acc = e_s[tt + u] * vv[u] + acc;
        }
        for (; tt < clen; ++tt) acc = e_s[tt] * vw[(size_t)tt * a.kv_dim] + acc;
    }
    if (j < hs) a.out[(size_t)b * a.out_stride + h * hs + j] = acc;
}

// ---------------------------------------------------------------------------------------------------
// Prefill of ONE sequence (token b sits at position pos0 + b): the K / V tiles are shared by a tile of PA_TB tokens
// instead of being re-read for every token.  Per-element arithmetic and order are those of the per-token kernels above.
//
// Scores: grid = (64-timestep K tiles, n_kv_heads, token tiles), block = 64 x kvMul (kvMul <= 4).  Thread (head hq,
// timestep r) keeps its K row in registers and walks the PA_TB query rows of its head (four chains in flight).
template <int I, int N, int STEP, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + STEP, N, STEP>(f);
    }
}
// maximum over the 64 lanes, uniform result: four DPP steps inside the rows of 16 lanes, then one readlane per row (VALU only; wave_max's
// six ds_bpermute round trips would sit on the score chains' critical path)
#define GL3_DPP_MAX(V_, CTRL_) V_ = fmaxf(V_, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, V_), CTRL_, 0xf, 0xf, false)))
__device__ __forceinline__ float row8_max(float v) {        // every lane: maximum over its aligned group of 8 lanes
    GL3_DPP_MAX(v, 0xB1); GL3_DPP_MAX(v, 0x4E); GL3_DPP_MAX(v, 0x141);      // quad_perm [1,0,3,2], quad_perm [2,3,0,1], row_half_mirror
    return v;
}
__device__ __forceinline__ float wave_max_uniform(float v) {
    v = row8_max(v); GL3_DPP_MAX(v, 0x140);                                  // row_mirror: the row of 16
    const float a = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 0));
    const float b = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 16));
    const float c = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 32));
    const float d = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 48));
    return fmaxf(fmaxf(a, b), fmaxf(c, d));
}
// two score chains advance 16 elements: score = score + q[j] * k[j], j ascending (no FMA)
__device__ __forceinline__ void score_step16(float& s0, float& s1, const v16f_t& qa, const v16f_t& qb, const float4* k) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        s0 = s0 + qa[4 * i] * k[i].x;     s1 = s1 + qb[4 * i] * k[i].x;
        s0 = s0 + qa[4 * i + 1] * k[i].y; s1 = s1 + qb[4 * i + 1] * k[i].y;
        s0 = s0 + qa[4 * i + 2] * k[i].z; s1 = s1 + qb[4 * i + 2] * k[i].z;
        s0 = s0 + qa[4 * i + 3] * k[i].w; s1 = s1 + qb[4 * i + 3] * k[i].w;
    }
}
constexpr int PA_TB = 16;
template <int HS>
__global__ __launch_bounds__(256) void pf_scores_tiled_kernel(const float* __restrict__ Q, int q_stride, const float* __restrict__ kc,
                                                              float* __restrict__ att, int n_heads, int kvmul, int kv_dim, int ctx,
                                                              int pos0, int ntok, float att_mul, float* __restrict__ tmx, int tmx_tiles) {
    extern __shared__ __attribute__((aligned(16))) float kt[];       // [64][PITCH]
    constexpr int PITCH = HS + 4, H4 = HS / 4;
    const int t = threadIdx.x, nthr = blockDim.x;
    const int t0 = blockIdx.x * 64, kvh = blockIdx.y, b0 = blockIdx.z * PA_TB;
    const int nb = min(PA_TB, ntok - b0);
    const int tmax = pos0 + b0 + nb - 1;              // last timestep any token of this tile attends to
    if (tmax < t0) return;
    const int t1 = min(tmax + 1, t0 + 64);
    for (int i = t; i < (t1 - t0) * H4; i += nthr) {
        const int r = i / H4, c = i % H4;
        *reinterpret_cast<float4*>(kt + r * PITCH + 4 * c) =
            *reinterpret_cast<const float4*>(kc + (size_t)(t0 + r) * kv_dim + kvh * HS + 4 * c);
    }
    __syncthreads();
    const int hq = __builtin_amdgcn_readfirstlane(t >> 6), r = t & 63;
    float4 kr[H4];
#pragma unroll
    for (int c = 0; c < H4; ++c) kr[c] = *reinterpret_cast<const float4*>(kt + min(r, t1 - t0 - 1) * PITCH + 4 * c);
    const float sqrt_hs = (float)sqrt((double)HS);
    const int head = kvh * kvmul + hq;
    for (int tb = 0; tb < nb; tb += 2) {
        // the two query rows are wavefront-uniform: scalar loads (8 floats per row per step, double-buffered), SGPR
        // operands in the multiplies.  Explicit s_load: the compiler would hoist every load and spill SGPRs.
        const float* q0 = Q + (size_t)(b0 + tb) * q_stride + (size_t)head * HS;
        const float* q1 = Q + (size_t)(b0 + min(tb + 1, nb - 1)) * q_stride + (size_t)head * HS;
        float s0 = 0.f, s1 = 0.f;
        if constexpr (HS >= 64) {
            v16f_t a0, a1, c0, c1;                    // 16 q values per row per step (64 SGPRs for the double buffer)
            asm volatile("s_load_dwordx16 %0, %2, 0x0\n\ts_load_dwordx16 %1, %3, 0x0" : "=&s"(a0), "=&s"(a1) : "s"(q0), "s"(q1));
            static_for<0, H4 / 4, 2>([&](auto ic) {
                constexpr int c4 = decltype(ic)::value;
                asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(a0), "+s"(a1), "+v"(s0), "+v"(s1));   // s0/s1 pin the VALU chain between the asm statements
                asm volatile("s_load_dwordx16 %0, %2, %4\n\ts_load_dwordx16 %1, %3, %4" : "=&s"(c0), "=&s"(c1) : "s"(q0), "s"(q1), "n"((c4 + 1) * 64));
                score_step16(s0, s1, a0, a1, &kr[4 * c4]);
                asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(c0), "+s"(c1), "+v"(s0), "+v"(s1));
                if constexpr (c4 + 2 < H4 / 4)
                    asm volatile("s_load_dwordx16 %0, %2, %4\n\ts_load_dwordx16 %1, %3, %4" : "=&s"(a0), "=&s"(a1) : "s"(q0), "s"(q1), "n"((c4 + 2) * 64));
                score_step16(s0, s1, c0, c1, &kr[4 * c4 + 4]);
            });
        } else {
            v16f_t a0, a1, c0, c1;
            asm volatile("s_load_dwordx16 %0, %2, 0x0\n\ts_load_dwordx16 %1, %3, 0x0" : "=&s"(a0), "=&s"(a1) : "s"(q0), "s"(q1));
            asm volatile("s_load_dwordx16 %0, %2, 64\n\ts_load_dwordx16 %1, %3, 64" : "=&s"(c0), "=&s"(c1) : "s"(q0), "s"(q1));
            asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(a0), "+s"(a1), "+s"(c0), "+s"(c1), "+v"(s0), "+v"(s1));
            score_step16(s0, s1, a0, a1, &kr[0]);
            score_step16(s0, s1, c0, c1, &kr[4]);
        }
        const int b = b0 + tb;
        const float v0 = att_mul != 0.f ? s0 * att_mul : s0 / sqrt_hs, v1 = att_mul != 0.f ? s1 * att_mul : s1 / sqrt_hs;
        const bool ok0 = t0 + r <= pos0 + b, ok1 = tb + 1 < nb && t0 + r <= pos0 + b + 1;       // (tmax >= pos0 + b: both imply t0 + r < t1)
        if (ok0) att[((size_t)b * n_heads + head) * ctx + t0 + r] = v0;
        if (ok1) att[((size_t)(b + 1) * n_heads + head) * ctx + t0 + r] = v1;
        if (tmx) {      // r6: the tile's maximum per (token, head) row for pf_softmax_rows_kernel (max is order-independent)
            const float m0 = wave_max_uniform(ok0 ? v0 : -INFINITY), m1 = wave_max_uniform(ok1 ? v1 : -INFINITY);
            if (r == 0) {
                if (t0 <= pos0 + b) tmx[((size_t)b * n_heads + head) * tmx_tiles + blockIdx.x] = m0;
                if (tb + 1 < nb && t0 <= pos0 + b + 1) tmx[((size_t)(b + 1) * n_heads + head) * tmx_tiles + blockIdx.x] = m1;
            }
        }
    }
}

// r6 — pf_scores_tiled_kernel on packed f32: two tokens' chains advance in ONE register pair, {s_a, s_b} = {s_a, s_b} + {q_a[j], q_b[j]} * k[j]
// (v_pk_mul_f32 + v_pk_add_f32: every product and every sum rounded as before, j ascending), two pairs side by side per wavefront.  The query
// rows of the tile's 16 tokens reach LDS once per workgroup, interleaved by token pairs ([head][pair][j][2]), so a 16-byte broadcast read is
// two steps of a pair; the reads of the next group of 8 steps are pinned under the current group's 32 packed instructions.  (The scalar-load
// form feeds q through SGPRs: its lead is bounded by the SGPR file — one group of 16 steps — and the scalar cache misses to L2 at 32 KB of
// query rows per workgroup.)  K rows in registers as before, lane = timestep.  Two wavefronts per SIMD (~210 VGPRs, 65 KB of LDS).
typedef float v2f_native __attribute__((ext_vector_type(2)));
typedef float v4f_native_s __attribute__((ext_vector_type(4)));
template <int HS, int KVM>
__global__ __launch_bounds__(64 * KVM) __attribute__((amdgpu_waves_per_eu(2, 2))) void pf_scores_pk_kernel(const float* __restrict__ Q, int q_stride, const float* __restrict__ kc,
        float* __restrict__ att, int n_heads, int kvmul_, int kv_dim, int ctx, int pos0, int ntok, float att_mul, float* __restrict__ tmx, int tmx_tiles) {
    extern __shared__ __attribute__((aligned(16))) float kt[];       // [64][PITCH] K rows, then [KVM][8 pairs][HS][2] query rows
    constexpr int PITCH = HS + 4, H4 = HS / 4, NGR = HS / 8, NT = 64 * KVM, KPT = 64 * H4 / NT, QPT = H4 / 8, kvmul = KVM;
    static_assert(KPT >= 1 && QPT >= 1, "staging slots per thread");
    float* qs = kt + 64 * PITCH;
    const int t = threadIdx.x;
    const int t0 = blockIdx.x * 64, kvh = blockIdx.y, b0 = blockIdx.z * PA_TB;
    const int nb = min(PA_TB, ntok - b0);
    const int tmax = pos0 + b0 + nb - 1;
    if (tmax < t0) return;
    const int t1 = min(tmax + 1, t0 + 64);
    {   // staging: EVERY global load of the workgroup's K tile and query rows is in flight before the first LDS write (a loop with a run-time
        // trip count keeps one load per thread in flight: 8 + 8 L2 round trips per workgroup against ~8 us of arithmetic)
        v4f_native_s kreg[KPT], qra[QPT], qrb[QPT];
#pragma unroll
        for (int j = 0; j < KPT; ++j) {                               // rows past the tile's last timestep repeat it (their scores are never stored)
            const int i = t + NT * j, r = i / H4, c = i % H4;
            kreg[j] = *reinterpret_cast<const v4f_native_s*>(kc + (size_t)(t0 + min(r, t1 - t0 - 1)) * kv_dim + kvh * HS + 4 * c);
        }
#pragma unroll
        for (int j = 0; j < QPT; ++j) {                               // slot = (head, token pair, 4 columns): both tokens' float4
            const int i = t + NT * j, c = i % H4, pair = (i / H4) % (PA_TB / 2), hq = i / (H4 * (PA_TB / 2));
            const float* qp = Q + (size_t)(kvh * kvmul + hq) * HS + 4 * c;
            qra[j] = *reinterpret_cast<const v4f_native_s*>(qp + (size_t)(b0 + min(2 * pair, nb - 1)) * q_stride);
            qrb[j] = *reinterpret_cast<const v4f_native_s*>(qp + (size_t)(b0 + min(2 * pair + 1, nb - 1)) * q_stride);
        }
#pragma unroll
        for (int j = 0; j < KPT; ++j) {
            const int i = t + NT * j, r = i / H4, c = i % H4;
            *reinterpret_cast<v4f_native_s*>(kt + r * PITCH + 4 * c) = kreg[j];
        }
#pragma unroll
        for (int j = 0; j < QPT; ++j) {
            const int i = t + NT * j, c = i % H4, pair = (i / H4) % (PA_TB / 2), hq = i / (H4 * (PA_TB / 2));
            float* d = qs + ((size_t)(hq * (PA_TB / 2) + pair) * HS + 4 * c) * 2;
            *reinterpret_cast<v4f_native_s*>(d) = (v4f_native_s){qra[j].x, qrb[j].x, qra[j].y, qrb[j].y};
            *reinterpret_cast<v4f_native_s*>(d + 4) = (v4f_native_s){qra[j].z, qrb[j].z, qra[j].w, qrb[j].w};
        }
    }
    __syncthreads();
    const int hq = __builtin_amdgcn_readfirstlane(t >> 6), r = t & 63;
    v2f_native kr[HS / 2];                                             // this lane's K row as 64-bit operands {k[2 i], k[2 i + 1]}
#pragma unroll
    for (int c = 0; c < H4; ++c) {
        const v4f_native_s x = *reinterpret_cast<const v4f_native_s*>(kt + r * PITCH + 4 * c);
        kr[2 * c] = x.xy; kr[2 * c + 1] = x.zw;
    }
    const float sqrt_hs = (float)sqrt((double)HS);
    const int head = kvh * kvmul + hq;
    for (int pp = 0; 4 * pp < nb; ++pp) {
        const float* q01 = qs + (size_t)(hq * (PA_TB / 2) + 2 * pp) * HS * 2;      // pairs (4 pp, 4 pp + 1) and (4 pp + 2, 4 pp + 3)
        const float* q23 = q01 + HS * 2;
        v2f_native c0 = {0.f, 0.f}, c1 = {0.f, 0.f};
        v4f_native_s qa[8], qb[8];                                     // two groups of 8 steps: [0..3] pair 0, [4..7] pair 1
#define SPK_LD(G_, R_) do { _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_) { \
            R_[i_] = *reinterpret_cast<const v4f_native_s*>(q01 + 16 * (G_) + 4 * i_); R_[4 + i_] = *reinterpret_cast<const v4f_native_s*>(q23 + 16 * (G_) + 4 * i_); } } while (0)
        // two steps of both pairs in one block: the K value is broadcast out of its register pair by op_sel (the compiler materialises {k, k}
        // pairs instead: twice the K registers), and every dependent instruction has an independent one in front of it (packed f32 needs a wait
        // state between a result and its use)
#define SPK_STEP2(QA_, QB_, K_) do { v2f_native p0_, p1_; \
            asm("v_pk_mul_f32 %[p0], %[qa0], %[k] op_sel_hi:[1,0]\n\tv_pk_mul_f32 %[p1], %[qb0], %[k] op_sel_hi:[1,0]\n\t" \
                "v_pk_add_f32 %[c0], %[c0], %[p0]\n\tv_pk_add_f32 %[c1], %[c1], %[p1]\n\t" \
                "v_pk_mul_f32 %[p0], %[qa1], %[k] op_sel:[0,1]\n\tv_pk_mul_f32 %[p1], %[qb1], %[k] op_sel:[0,1]\n\t" \
                "v_pk_add_f32 %[c0], %[c0], %[p0]\n\tv_pk_add_f32 %[c1], %[c1], %[p1]" \
                : [c0] "+v"(c0), [c1] "+v"(c1), [p0] "=&v"(p0_), [p1] "=&v"(p1_) \
                : [qa0] "v"(QA_.xy), [qa1] "v"(QA_.zw), [qb0] "v"(QB_.xy), [qb1] "v"(QB_.zw), [k] "v"(K_)); } while (0)
#define SPK_ACC(G_, R_) do { SPK_STEP2(R_[0], R_[4], kr[4 * (G_)]); SPK_STEP2(R_[1], R_[5], kr[4 * (G_) + 1]); \
            SPK_STEP2(R_[2], R_[6], kr[4 * (G_) + 2]); SPK_STEP2(R_[3], R_[7], kr[4 * (G_) + 3]); } while (0)
        SPK_LD(0, qa); SPK_LD(1, qb); __builtin_amdgcn_sched_barrier(0);
        static_for<0, NGR, 2>([&](auto gc) {
            constexpr int g = decltype(gc)::value;
            SPK_ACC(g, qa);
            SPK_LD((g + 2 < NGR ? g + 2 : NGR - 1), qa); __builtin_amdgcn_sched_barrier(0);
            SPK_ACC(g + 1, qb);
            SPK_LD((g + 3 < NGR ? g + 3 : NGR - 1), qb); __builtin_amdgcn_sched_barrier(0);
        });
        const float sv[4] = {c0.x, c0.y, c1.x, c1.y};
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int tb = 4 * pp + u, b = b0 + tb;
            const float v = att_mul != 0.f ? sv[u] * att_mul : sv[u] / sqrt_hs;
            const bool ok = tb < nb && t0 + r <= pos0 + b;
            if (ok) att[((size_t)b * n_heads + head) * ctx + t0 + r] = v;
            if (tmx) {
                const float m = wave_max_uniform(ok ? v : -INFINITY);
                if (r == 0 && tb < nb && t0 <= pos0 + b) tmx[((size_t)b * n_heads + head) * tmx_tiles + blockIdx.x] = m;
            }
        }
    }
}

// Softmax of every (token, head) score row, in place: one wavefront per row, wpw rows per workgroup.
// max -> exp in double -> sequential f32 sum -> divide (InferenceCore.java softmax via FloatTensor.softmaxInPlace).
__global__ __launch_bounds__(256) void pf_softmax_kernel(const PfAttnArgs a, int ntok, int wpw, int npad) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int pair = blockIdx.x * wpw + w;
    const bool live = w < wpw && pair < ntok * a.n_heads;
    const int b = live ? pair / a.n_heads : 0;
    const int n = live ? a.pos[b] + 1 : 0;
    float* sc = a.att + (size_t)(live ? pair : 0) * a.ctx;
    float* e_s = sm + (size_t)(w < wpw ? w : 0) * npad;
    float mx = -INFINITY;
    for (int i = lane; i < n; i += 64) { const float v = sc[i]; e_s[i] = v; mx = fmaxf(mx, v); }
    mx = wave_max(mx);
    for (int i = lane; i < n; i += 64) e_s[i] = (float)exp((double)(e_s[i] - mx));   // lane-private slots so far
    __syncthreads();
    const float sum = seq_sum_lds_ring(e_s, n);      // LDS reads pinned three groups ahead of the adds (~6 instead of ~11 cycles per element)
    for (int i = lane; i < n; i += 64) sc[i] = e_s[i] / sum;
}

// r6 — softmax of the score rows at depth, R rows per workgroup: 8 worker wavefronts stream the rows' 64-timestep tiles (loads a tile ahead,
// e_t = (float) exp((double) (s_t - max)) written back in place and into a double-buffered LDS tile), a ninth wavefront runs the strictly
// sequential sums with lane = row — R chains side by side, LDS reads pinned ahead of the adds (seq_sum_lds_ring).  The row maxima come from
// the per-tile maxima pf_scores_tiled_kernel leaves in tmx (max is order-independent); the denominators go to `sums` and the division
// e_t / sum happens where the weights are staged (pf_pv_tiled_kernel) — same operands, same rounding as FloatTensor.softmaxInPlace
// (J/tensor/standard/FloatTensor.java:196-219: max, exp, sum, divide).  pf_softmax_kernel keeps ONE row per wavefront in LDS: its row loads
// are one HBM round trip per 64 scores, its sums one chain per wavefront and at most three rows per workgroup fit at 4608 positions:
// 640 us per 8B layer at pp512 @ d4096 against ~150 us here.  Needs ctx % 4 == 0 (16-byte row starts).
constexpr int SR_PITCH = 68;
template <int R>
__global__ __launch_bounds__(576) void pf_softmax_rows_kernel(const PfAttnArgs a, int nrows_total, const float* __restrict__ tmx, int tmx_tiles, float* __restrict__ sums) {
    __shared__ __attribute__((aligned(16))) float E[2][R * SR_PITCH];
    __shared__ float mx_s[R];
    __shared__ int n_s[R];
    __shared__ int nmax_s;
    const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int row0 = blockIdx.x * R;
    if (t == 0) nmax_s = 0;
    if (t < R * 8) {                                  // 8 lanes fold a row's tile maxima
        const int r = t >> 3, sub = t & 7, row = row0 + r;
        const int n = row < nrows_total ? a.pos[row / a.n_heads] + 1 : 0;
        const int nt = (n + 63) >> 6;
        float m = -INFINITY;
        for (int i = sub; i < nt; i += 8) m = fmaxf(m, tmx[(size_t)row * tmx_tiles + i]);
        m = row8_max(m);
        if (sub == 0) { mx_s[r] = m; n_s[r] = n; }
    }
    __syncthreads();
    if (t < R) atomicMax(&nmax_s, n_s[t]);
    __syncthreads();
    const int ntile = (nmax_s + 63) >> 6;
    if (wave == 8) {                                  // the chains: lane = row
        const int r = min(lane, R - 1);
        float s = 0.f;
        for (int k = 0; k < ntile; ++k) {
            __syncthreads();                          // tile k has landed in E[k & 1]; the workers refill it behind the NEXT barrier
            s = seq_sum_lds_ring(&E[k & 1][r * SR_PITCH], 64, s);
        }
        if (lane < R && row0 + lane < nrows_total) sums[row0 + lane] = s;
        return;
    }
    constexpr int NS = (R * 16 + 511) / 512;          // 16-byte slots per worker thread and tile
    float* rowp[NS]; float mrow[NS]; int nrow[NS], ldsoff[NS];
#pragma unroll
    for (int u = 0; u < NS; ++u) {
        const int q = min(t + 512 * u, R * 16 - 1), r = q >> 4, c4 = q & 15;
        rowp[u] = a.att + (size_t)min(row0 + r, nrows_total - 1) * a.ctx + 4 * c4;
        mrow[u] = mx_s[r];
        nrow[u] = (t + 512 * u < R * 16) ? n_s[r] - 4 * c4 : 0;       // elements of the row at and behind this slot's first column of tile 0
        ldsoff[u] = r * SR_PITCH + 4 * c4;
    }
    float4 cur[NS], nxt[NS];
#pragma unroll
    for (int u = 0; u < NS; ++u) cur[u] = *reinterpret_cast<const float4*>(rowp[u]);               // tile 0 (column 4 c4 < 64 <= ctx)
    for (int k = 0; k < ntile; ++k) {
        const int kn = min(k + 1, ntile - 1);
#pragma unroll
        for (int u = 0; u < NS; ++u) {                // unconditional loads: slots past the row's end re-read tile 0 (masked below)
            const float* src = 64 * kn < nrow[u] ? rowp[u] + 64 * kn : rowp[u];
            nxt[u] = *reinterpret_cast<const float4*>(src);
        }
#pragma unroll
        for (int u = 0; u < NS; ++u) {
            const int left = nrow[u] - 64 * k;        // valid elements of this slot: min(left, 4)
            float4 e;
            e.x = left > 0 ? (float)exp((double)(cur[u].x - mrow[u])) : 0.f;
            e.y = left > 1 ? (float)exp((double)(cur[u].y - mrow[u])) : 0.f;
            e.z = left > 2 ? (float)exp((double)(cur[u].z - mrow[u])) : 0.f;
            e.w = left > 3 ? (float)exp((double)(cur[u].w - mrow[u])) : 0.f;
            if (t + 512 * u < R * 16) *reinterpret_cast<float4*>(&E[k & 1][ldsoff[u]]) = e;
            if (left > 0) *reinterpret_cast<float4*>(rowp[u] + 64 * k) = e;      // in place (columns past the row's end stay inside the row: ctx % 4 == 0)
        }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < NS; ++u) cur[u] = nxt[u];
    }
}

// Weighted V sum: grid = (n_heads, token tiles), block 256.  V tiles of 64 timesteps are staged in LDS once per
// workgroup; wavefront w carries tokens 4w..4w+3 of the tile, lane j the output columns j (+64): acc = a_t * v + acc,
// t ascending.  The softmax weights are wavefront-uniform loads.
template <int NCOL>
__global__ __launch_bounds__(256) void pf_pv_tiled_kernel(const PfAttnArgs a, int seq, int pos0, int ntok, const float* __restrict__ sums) {
    extern __shared__ __attribute__((aligned(16))) float vt[];        // [64][hs]
    const int hs = a.hs, h4 = hs >> 2, kvmul = a.n_heads / a.n_kv_heads;
    const int t = threadIdx.x, lane = t & 63, w = __builtin_amdgcn_readfirstlane(t >> 6);
    const int h = blockIdx.x, kvh = h / kvmul, b0 = blockIdx.y * PA_TB;
    const int nb = min(PA_TB, ntok - b0);
    const int tmax = pos0 + b0 + nb - 1;
    const float* vc = a.vcache + (size_t)seq * a.seq_stride;
    float* as = vt + 64 * hs;                                         // [PA_TB][64] softmax weights of the current tile
    int posu[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) posu[u] = 4 * w + u < nb ? pos0 + b0 + 4 * w + u : -1;   // -1: no timestep qualifies
    const int wmax = 4 * w < nb ? pos0 + b0 + min(4 * w + 3, nb - 1) : -1;
    float acc[4][NCOL];
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int c = 0; c < NCOL; ++c) acc[u][c] = 0.f;
    for (int t0 = 0; t0 <= tmax; t0 += 64) {
        const int tt = min(64, tmax + 1 - t0);
        __syncthreads();
        for (int i = t; i < tt * h4; i += 256) {
            const int r = i / h4, c = i % h4;
            *reinterpret_cast<float4*>(vt + r * hs + 4 * c) =
                *reinterpret_cast<const float4*>(vc + (size_t)(t0 + r) * a.kv_dim + kvh * hs + 4 * c);
        }
        for (int i = t; i < PA_TB * 64; i += 256) {                   // entries past a token's position are never used
            const int tb = i >> 6, r = i & 63;
            float wv = (tb < nb && t0 + r <= pos0 + b0 + tb) ? a.att[((size_t)(b0 + tb) * a.n_heads + h) * a.ctx + t0 + r] : 0.f;
            if (sums) wv = wv / sums[(size_t)(b0 + min(tb, nb - 1)) * a.n_heads + h];      // r6: att holds the numerators (pf_softmax_rows_kernel)
            as[i] = wv;
        }
        __syncthreads();
        const int ttw = min(tt, wmax + 1 - t0);                       // this wavefront's tokens stop at wmax
        // timesteps every one of the four tokens attends to (t <= position of the first token): no conditions
        const int rfull = (4 * w + 3 < nb) ? max(0, min(tt, posu[0] + 1 - t0)) & ~3 : 0;
        auto vload = [&](int r, float (&v)[NCOL]) {
            if (NCOL == 2) {                                          // lane owns columns 2*lane, 2*lane + 1
                const float2 v2 = (2 * lane < hs) ? *reinterpret_cast<const float2*>(vt + r * hs + 2 * lane) : make_float2(0.f, 0.f);
                v[0] = v2.x; v[NCOL - 1] = v2.y;
            } else {
                v[0] = lane < hs ? vt[r * hs + lane] : 0.f;
            }
        };
        for (int r = 0; r < rfull; r += 4) {
            float4 a4[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) a4[u] = *reinterpret_cast<const float4*>(as + (4 * w + u) * 64 + r);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float v[NCOL];
                vload(r + i, v);
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const float at = i == 0 ? a4[u].x : i == 1 ? a4[u].y : i == 2 ? a4[u].z : a4[u].w;
#pragma unroll
                    for (int c = 0; c < NCOL; ++c) acc[u][c] = at * v[c] + acc[u][c];
                }
            }
        }
        for (int r = rfull; r < ttw; ++r) {                           // the diagonal: per-token conditions (uniform)
            float v[NCOL];
            vload(r, v);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (t0 + r <= posu[u]) {
                    const float at = as[(4 * w + u) * 64 + r];
#pragma unroll
                    for (int c = 0; c < NCOL; ++c) acc[u][c] = at * v[c] + acc[u][c];
                }
            }
        }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int tb = 4 * w + u;
        if (tb >= nb) continue;
#pragma unroll
        for (int c = 0; c < NCOL; ++c) {
            const int j = NCOL == 2 ? 2 * lane + c : lane;
            if (j < hs) a.out[(size_t)(b0 + tb) * a.out_stride + h * hs + j] = acc[u][c];
        }
    }
}

// r6 — pf_pv_tiled_kernel with the two latencies taken off its critical path (same arithmetic: acc = a_t * v + acc, t ascending; token tiles of 32):
//   * the NEXT tile's V rows and softmax numerators are requested into registers before the current tile is consumed and reach LDS behind it
//     (pf_pv_tiled_kernel loads, waits and stores between two barriers, 72 times per workgroup at 4608 positions);
//   * inside a tile the LDS reads of timestep group g + 1 (four weights per token, four V rows) are in flight under the 64 multiply-add pairs
//     of group g, pinned there with sched_barrier (left alone the scheduler sinks every read next to its use).
// Masking is by weight: timesteps behind a token's position get the weight 0 (0 * v + acc = acc exactly: acc is never -0 and every staged
// V row is a written row <= the tile's last position), so the inner loop has no per-token conditions.  The division e_t / sum happens
// where the weights are staged (sums from pf_softmax_rows_kernel).
constexpr int PVR_NW = 8, PVR_TB = 4 * PVR_NW;     // 8 wavefronts of 4 tokens: a V tile serves 32 tokens, 512 workgroups for 512 tokens x 32 heads (two per CU)
template <int HS>
__global__ __launch_bounds__(64 * PVR_NW) void pf_pv_ring_kernel(const PfAttnArgs a, int seq, int pos0, int ntok, const float* __restrict__ sums) {
    constexpr int NCOL = HS > 64 ? 2 : 1, H4 = HS / 4, NT = 64 * PVR_NW, VPT = 64 * H4 / NT;
    static_assert(VPT >= 1, "a V tile is at least one 16-byte slot per thread");
    extern __shared__ __attribute__((aligned(16))) float vt[];        // [64][HS] V rows, then [PVR_TB][64] weights
    float* as = vt + 64 * HS;
    const int t = threadIdx.x, lane = t & 63, w = __builtin_amdgcn_readfirstlane(t >> 6);
    const int kvmul = a.n_heads / a.n_kv_heads, h = blockIdx.x, kvh = h / kvmul, b0 = blockIdx.y * PVR_TB;
    const int nb = min(PVR_TB, ntok - b0);
    const int tmax = pos0 + b0 + nb - 1, ntile = tmax / 64 + 1;
    const int wmax = 4 * w < nb ? pos0 + b0 + min(4 * w + 3, nb - 1) : -1;     // last position any of this wavefront's four tokens attends to
    const float* vc = a.vcache + (size_t)seq * a.seq_stride + kvh * HS;
    // staging roles: thread = (token w + PVR_NW j, timestep lane) of the weights; 16-byte slots t + NT j of the V tile
    const float* arow[4]; float rsum[4]; int apos[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int tb = w + PVR_NW * j;
        apos[j] = tb < nb ? pos0 + b0 + tb : -1;
        arow[j] = a.att + ((size_t)(b0 + min(tb, nb - 1)) * a.n_heads + h) * a.ctx;
        rsum[j] = sums[(size_t)(b0 + min(tb, nb - 1)) * a.n_heads + h];
    }
    typedef float v4f_native __attribute__((ext_vector_type(4)));     // typed loads / stores (a float4 array that is only copied in and out stays a stack object)
    v4f_native vreg[VPT]; float areg[4];
#define PVR_GLOAD(K_) do { const int t0_ = 64 * (K_); \
        static_for<0, VPT, 1>([&](auto jc) { constexpr int j = decltype(jc)::value; const int i = t + NT * j, r = i / H4, c = i % H4; \
            vreg[j] = *reinterpret_cast<const v4f_native*>(vc + (size_t)min(t0_ + r, tmax) * a.kv_dim + 4 * c); }); \
        _Pragma("unroll") for (int j = 0; j < 4; ++j) areg[j] = arow[j][max(min(t0_ + lane, apos[j]), 0)]; } while (0)
#define PVR_LSTORE(K_) do { const int t0_ = 64 * (K_); \
        static_for<0, VPT, 1>([&](auto jc) { constexpr int j = decltype(jc)::value; const int i = t + NT * j, r = i / H4, c = i % H4; \
            *reinterpret_cast<v4f_native*>(vt + r * HS + 4 * c) = vreg[j]; }); \
        _Pragma("unroll") for (int j = 0; j < 4; ++j) as[(w + PVR_NW * j) * 64 + lane] = t0_ + lane <= apos[j] ? areg[j] / rsum[j] : 0.f; } while (0)
    float acc[4][NCOL];
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int c = 0; c < NCOL; ++c) acc[u][c] = 0.f;
    const float* vp = vt + (NCOL == 2 ? 2 * lane : min(lane, HS - 1));
    const float* ap = as + 4 * w * 64;
    PVR_GLOAD(0);
    PVR_LSTORE(0);
    __syncthreads();
    for (int k = 0; k < ntile; ++k) {
        PVR_GLOAD(min(k + 1, ntile - 1));                            // unconditional (a condition around the loads makes the compiler drain them)
        const int ng = max(0, min(64, wmax + 1 - 64 * k) + 3) >> 2;   // groups of four timesteps this wavefront's tokens reach in the tile
        float4 wa0, wa1, wa2, wa3, wb0, wb1, wb2, wb3;
        float va[4][NCOL], vb[4][NCOL];
#define PVR_LD(G_, W0_, W1_, W2_, W3_, V_) do { const int r_ = 4 * min((G_), 15); \
            W0_ = *reinterpret_cast<const float4*>(ap + r_); W1_ = *reinterpret_cast<const float4*>(ap + 64 + r_); \
            W2_ = *reinterpret_cast<const float4*>(ap + 128 + r_); W3_ = *reinterpret_cast<const float4*>(ap + 192 + r_); \
            _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_) { \
                if (NCOL == 2) { const float2 x_ = *reinterpret_cast<const float2*>(vp + (r_ + i_) * HS); V_[i_][0] = x_.x; V_[i_][NCOL - 1] = x_.y; } \
                else V_[i_][0] = vp[(r_ + i_) * HS]; } } while (0)
#define PVR_STEP(I_, WX_, V_) do { const float w_[4] = {W0X_.WX_, W1X_.WX_, W2X_.WX_, W3X_.WX_}; \
            _Pragma("unroll") for (int u_ = 0; u_ < 4; ++u_) _Pragma("unroll") for (int c_ = 0; c_ < NCOL; ++c_) acc[u_][c_] = w_[u_] * V_[I_][c_] + acc[u_][c_]; } while (0)
        PVR_LD(0, wa0, wa1, wa2, wa3, va); PVR_LD(1, wb0, wb1, wb2, wb3, vb); __builtin_amdgcn_sched_barrier(0);
        int g = 0;
        for (; g + 2 <= ng; g += 2) {
#define W0X_ wa0
#define W1X_ wa1
#define W2X_ wa2
#define W3X_ wa3
            PVR_STEP(0, x, va); PVR_STEP(1, y, va); PVR_STEP(2, z, va); PVR_STEP(3, w, va);
#undef W0X_
#undef W1X_
#undef W2X_
#undef W3X_
            PVR_LD(g + 2, wa0, wa1, wa2, wa3, va); __builtin_amdgcn_sched_barrier(0);
#define W0X_ wb0
#define W1X_ wb1
#define W2X_ wb2
#define W3X_ wb3
            PVR_STEP(0, x, vb); PVR_STEP(1, y, vb); PVR_STEP(2, z, vb); PVR_STEP(3, w, vb);
#undef W0X_
#undef W1X_
#undef W2X_
#undef W3X_
            PVR_LD(g + 3, wb0, wb1, wb2, wb3, vb); __builtin_amdgcn_sched_barrier(0);
        }
        if (g < ng) {
#define W0X_ wa0
#define W1X_ wa1
#define W2X_ wa2
#define W3X_ wa3
            PVR_STEP(0, x, va); PVR_STEP(1, y, va); PVR_STEP(2, z, va); PVR_STEP(3, w, va);
#undef W0X_
#undef W1X_
#undef W2X_
#undef W3X_
        }
#undef PVR_LD
#undef PVR_STEP
        __syncthreads();
        PVR_LSTORE(min(k + 1, ntile - 1));
        __syncthreads();
    }
#undef PVR_GLOAD
#undef PVR_LSTORE
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int tb = 4 * w + u;
        if (tb >= nb) continue;
#pragma unroll
        for (int c = 0; c < NCOL; ++c) {
            const int j = NCOL == 2 ? 2 * lane + c : lane;
            if (j < HS) a.out[(size_t)(b0 + tb) * a.out_stride + h * HS + j] = acc[u][c];
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// Round 4: scores + softmax + weighted V sum of ONE sequence's prefill chunk in one launch (the three kernels above stay for long
// contexts and odd shapes).  Workgroup = (kv head, tile of FA_TB = 8 tokens), 2 * kvMul wavefronts; the score rows of the tile's
// kvMul x 8 (query head, token) pairs live in LDS from the first q.k to the last a.v — no [token][head][ctx] round trip through
// HBM / L2 (33 MB written, read, rewritten and read again per 8B layer at 512 tokens) and one launch instead of three.
//   phase 1  scores: the K tiles of 64 timesteps alternate between the two wavefront groups; thread = (query head, timestep), K row
//            in registers, the query rows as SGPR operands (pf_scores_tiled_kernel's inner loop: j ascending, product rounded, no FMA)
//   phase 2  softmax rows (FloatTensor.softmaxInPlace): max, exp in double, the strictly sequential sum of ALL rows at once
//            (lane = row: 32 chains side by side instead of one row per wavefront), divide
//   phase 3  weighted V sum: V tiles of 64 timesteps through LDS, wavefront = (query head, 4 tokens), lane = 2 columns,
//            acc = a_t * v + acc with t ascending (pf_pv_tiled_kernel's inner loop)
// Tiles are dealt heaviest (latest positions) first, so the triangular work profile does not leave a tail.
constexpr int FA_TB = 8;
__host__ __device__ constexpr size_t fa_smem_bytes(int hs, int kvmul, int sstride) {
    return ((size_t)kvmul * FA_TB * sstride + 2 * 64 * (hs + 4) + 64) * 4;
}
template <int HS>
__global__ __launch_bounds__(512) void pf_attn_fused_kernel(const float* __restrict__ Q, int q_stride, const float* __restrict__ kc, const float* __restrict__ vc,
                                                            float* __restrict__ out, int out_stride, int n_kv_heads, int kvmul, int kv_dim,
                                                            int pos0, int ntok, float att_mul, int sstride,
                                                            uint8_t* __restrict__ xq_out = nullptr, uint4* __restrict__ xp_out = nullptr, int xp_tok = 0) {
    extern __shared__ __attribute__((aligned(16))) float fa_sm[];
    constexpr int PITCH = HS + 4, H4 = HS / 4, NCOL = HS > 64 ? 2 : 1;
    float* Ssc = fa_sm;                                             // [kvmul][FA_TB][sstride] score -> softmax rows
    float* kt = fa_sm + (size_t)kvmul * FA_TB * sstride;            // [2][64][PITCH] K (phase 1) / V (phase 3) tiles
    float* sums = kt + 2 * 64 * PITCH;                              // [kvmul * FA_TB]
    const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int nthr = blockDim.x, gthreads = 64 * kvmul;
    const int grp = wave / kvmul, hq = wave % kvmul, gt = t - grp * gthreads;
    const int ntile = (ntok + FA_TB - 1) / FA_TB;
    const int kvh = blockIdx.x % n_kv_heads, tile = ntile - 1 - blockIdx.x / n_kv_heads;
    const int b0 = tile * FA_TB, nb = min(FA_TB, ntok - b0), tmax = pos0 + b0 + nb - 1;
    const int head = kvh * kvmul + hq;
    const float sqrt_hs = (float)sqrt((double)HS);

    // K / V tiles travel global -> registers -> LDS; the next tile's loads are in flight while the current one is consumed (clamped
    // rows: every address is inside the cache, the surplus rows are never read).  8 float4 per thread cover a 64-row tile (host check).
    // (written out at every site: a register array captured by a lambda, or filled in a macro loop, ends up in scratch with this compiler)
    constexpr int NPK = 8;
#ifdef FA_TIMING
    unsigned long long fa_t0 = __builtin_readcyclecounter(), fa_t1, fa_t2, fa_t3;
#endif
    // ---- phase 1: scores
    const int nkt = tmax / 64 + 1;
    float4 pk0, pk1, pk2, pk3, pk4, pk5, pk6, pk7;           // named registers: an array here is not promoted out of scratch
#define FA_REP8(X_) X_(0) X_(1) X_(2) X_(3) X_(4) X_(5) X_(6) X_(7)
    static_assert(NPK == 8, "FA_REP8");
    {
        const int ft0 = min(grp * 64, tmax), frows = max(1, min(64, tmax + 1 - grp * 64));
#define FA_F(U_) { const int fi = min(gt + U_ * gthreads, 64 * H4 - 1), fr = min(fi / H4, frows - 1), fc = fi % H4; \
                        pk##U_ = *reinterpret_cast<const float4*>(kc + (size_t)(ft0 + fr) * kv_dim + kvh * HS + 4 * fc); }
        FA_REP8(FA_F)
#undef FA_F
    }
    for (int trip = 0; 2 * trip < nkt; ++trip) {
        const int t0 = (2 * trip + grp) * 64;
        const bool live = t0 <= tmax;
        const int t1 = min(tmax + 1, t0 + 64);
        float* ktg = kt + grp * 64 * PITCH;
        if (live) {
#define FA_P(U_) { const int fi = gt + U_ * gthreads; if (fi < 64 * H4) *reinterpret_cast<float4*>(ktg + (fi / H4) * PITCH + 4 * (fi % H4)) = pk##U_; }
            FA_REP8(FA_P)
#undef FA_P
        }
        __syncthreads();
        {
            const int tn = t0 + 128;                                 // this group's next tile (clamped: fetched even if it is not used)
            {
                const int ft0 = min(tn, tmax), frows = max(1, min(64, tmax + 1 - tn));
#define FA_F(U_) { const int fi = min(gt + U_ * gthreads, 64 * H4 - 1), fr = min(fi / H4, frows - 1), fc = fi % H4; \
                                pk##U_ = *reinterpret_cast<const float4*>(kc + (size_t)(ft0 + fr) * kv_dim + kvh * HS + 4 * fc); }
                FA_REP8(FA_F)
#undef FA_F
            }
        }
        if (live) {
            float4 kr[H4];
#pragma unroll
            for (int c = 0; c < H4; ++c) kr[c] = *reinterpret_cast<const float4*>(ktg + min(lane, t1 - t0 - 1) * PITCH + 4 * c);
            for (int tb = 0; tb < nb; tb += 2) {
                // the two query rows are wavefront-uniform: scalar loads, SGPR operands in the multiplies (as pf_scores_tiled_kernel)
                const float* q0 = Q + (size_t)(b0 + tb) * q_stride + (size_t)head * HS;
                const float* q1 = Q + (size_t)(b0 + min(tb + 1, nb - 1)) * q_stride + (size_t)head * HS;
                float s0 = 0.f, s1 = 0.f;
                if constexpr (HS >= 64) {
                    v16f_t a0, a1, c0, c1;
                    asm volatile("s_load_dwordx16 %0, %2, 0x0\n\ts_load_dwordx16 %1, %3, 0x0" : "=&s"(a0), "=&s"(a1) : "s"(q0), "s"(q1));
                    static_for<0, H4 / 4, 2>([&](auto ic) {
                        constexpr int c4 = decltype(ic)::value;
                        asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(a0), "+s"(a1), "+v"(s0), "+v"(s1));
                        asm volatile("s_load_dwordx16 %0, %2, %4\n\ts_load_dwordx16 %1, %3, %4" : "=&s"(c0), "=&s"(c1) : "s"(q0), "s"(q1), "n"((c4 + 1) * 64));
                        score_step16(s0, s1, a0, a1, &kr[4 * c4]);
                        asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(c0), "+s"(c1), "+v"(s0), "+v"(s1));
                        if constexpr (c4 + 2 < H4 / 4)
                            asm volatile("s_load_dwordx16 %0, %2, %4\n\ts_load_dwordx16 %1, %3, %4" : "=&s"(a0), "=&s"(a1) : "s"(q0), "s"(q1), "n"((c4 + 2) * 64));
                        score_step16(s0, s1, c0, c1, &kr[4 * c4 + 4]);
                    });
                } else {
                    v16f_t a0, a1, c0, c1;
                    asm volatile("s_load_dwordx16 %0, %2, 0x0\n\ts_load_dwordx16 %1, %3, 0x0" : "=&s"(a0), "=&s"(a1) : "s"(q0), "s"(q1));
                    asm volatile("s_load_dwordx16 %0, %2, 64\n\ts_load_dwordx16 %1, %3, 64" : "=&s"(c0), "=&s"(c1) : "s"(q0), "s"(q1));
                    asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(a0), "+s"(a1), "+s"(c0), "+s"(c1), "+v"(s0), "+v"(s1));
                    score_step16(s0, s1, a0, a1, &kr[0]);
                    score_step16(s0, s1, c0, c1, &kr[4]);
                }
                const int ts = t0 + lane;
                if (ts < t1) {
                    if (ts <= pos0 + b0 + tb) Ssc[(size_t)(hq * FA_TB + tb) * sstride + ts] = att_mul != 0.f ? s0 * att_mul : s0 / sqrt_hs;
                    if (tb + 1 < nb && ts <= pos0 + b0 + tb + 1) Ssc[(size_t)(hq * FA_TB + tb + 1) * sstride + ts] = att_mul != 0.f ? s1 * att_mul : s1 / sqrt_hs;
                }
            }
        }
        __syncthreads();
    }

#ifdef FA_TIMING
    fa_t1 = __builtin_readcyclecounter();
#endif
    // the first V tile travels while the softmax runs
    {
        const int ft0 = 0, frows = min(64, tmax + 1);
#define FA_F(U_) { const int fi = min(t + U_ * nthr, 64 * H4 - 1), fr = min(fi / H4, frows - 1), fc = fi % H4; \
                        pk##U_ = *reinterpret_cast<const float4*>(vc + (size_t)(ft0 + fr) * kv_dim + kvh * HS + 4 * fc); }
        FA_REP8(FA_F)
#undef FA_F
    }
    // ---- phase 2: softmax of the kvmul * nb rows
    const int nrows = kvmul * nb, nwaves = nthr >> 6;
    for (int row = wave; row < nrows; row += nwaves) {
        const int tb = row % nb, n = pos0 + b0 + tb + 1;
        float* e = Ssc + (size_t)((row / nb) * FA_TB + tb) * sstride;
        float mx = -INFINITY;
        for (int i = lane; i < n; i += 64) mx = fmaxf(mx, e[i]);
        mx = wave_max(mx);
        for (int i = lane; i < n; i += 64) e[i] = (float)exp((double)(e[i] - mx));      // lane-private slots
    }
    __syncthreads();
    if (wave == 0 && lane < nrows) {                                 // lane = row: the strictly sequential sums, side by side
        const int tb = lane % nb, n = pos0 + b0 + tb + 1;
        const float* e = Ssc + (size_t)((lane / nb) * FA_TB + tb) * sstride;
        sums[lane] = seq_sum_lds_ring(e, n);                         // reads pinned three groups ahead (gl3_decode_kernels.h)
    }
    __syncthreads();
    for (int row = wave; row < nrows; row += nwaves) {
        const int tb = row % nb, n = pos0 + b0 + tb + 1;
        float* e = Ssc + (size_t)((row / nb) * FA_TB + tb) * sstride;
        const float sum = sums[row];
        for (int i = lane; i < n; i += 64) e[i] = e[i] / sum;
    }

#ifdef FA_TIMING
    fa_t2 = __builtin_readcyclecounter();
#endif
    // ---- phase 3: weighted V sum; wavefront = (query head hq, tokens 4 * grp .. + 3)
    int posu[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) posu[u] = 4 * grp + u < nb ? pos0 + b0 + 4 * grp + u : -1;       // -1: no timestep qualifies
    const int wmax = 4 * grp < nb ? pos0 + b0 + min(4 * grp + 3, nb - 1) : -1;
    const float* as = Ssc + (size_t)(hq * FA_TB + 4 * grp) * sstride;                             // rows of this wavefront's four tokens
    float acc[4][NCOL];
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int c = 0; c < NCOL; ++c) acc[u][c] = 0.f;
    int vb = 0;
    for (int t0 = 0; t0 <= tmax; t0 += 64, vb ^= 1) {
        const int tt = min(64, tmax + 1 - t0);
        float* vt = kt + vb * 64 * PITCH;
        {
#define FA_P(U_) { const int fi = t + U_ * nthr; if (fi < 64 * H4) *reinterpret_cast<float4*>(vt + (fi / H4) * PITCH + 4 * (fi % H4)) = pk##U_; }
            FA_REP8(FA_P)
#undef FA_P
        }
        __syncthreads();                                             // (also orders phase 2's writes before the first reads of `as`)
        {
            const int ft0 = min(t0 + 64, tmax), frows = max(1, min(64, tmax + 1 - (t0 + 64)));
#define FA_F(U_) { const int fi = min(t + U_ * nthr, 64 * H4 - 1), fr = min(fi / H4, frows - 1), fc = fi % H4; \
                            pk##U_ = *reinterpret_cast<const float4*>(vc + (size_t)(ft0 + fr) * kv_dim + kvh * HS + 4 * fc); }
            FA_REP8(FA_F)
#undef FA_F
        }
        const int ttw = min(tt, wmax + 1 - t0);
        const int rfull = (4 * grp + 3 < nb) ? max(0, min(tt, posu[0] + 1 - t0)) & ~3 : 0;       // timesteps all four tokens attend to
        auto vload = [&](int r, float (&v)[NCOL]) {
            if (NCOL == 2) {
                const float2 v2 = *reinterpret_cast<const float2*>(vt + r * PITCH + 2 * lane);
                v[0] = v2.x; v[NCOL - 1] = v2.y;
            } else {
                v[0] = lane < HS ? vt[r * PITCH + lane] : 0.f;
            }
        };
        // four timesteps per group; the next group's softmax weights and V rows are read from LDS while the current group is
        // accumulated (two named register sets: the un-pipelined loop spent ~2/3 of its time waiting for LDS, in-kernel stamps)
#define FA_LD(A_, V_, R_)                                                                                                    \
        do {                                                                                                                 \
            _Pragma("unroll") for (int u = 0; u < 4; ++u) A_[u] = *reinterpret_cast<const float4*>(as + (size_t)u * sstride + t0 + (R_)); \
            _Pragma("unroll") for (int i = 0; i < 4; ++i) vload((R_) + i, V_[i]);                                             \
        } while (0)
#define FA_ACC(A_, V_)                                                                                                       \
        do {                                                                                                                 \
            _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                                    \
                _Pragma("unroll") for (int u = 0; u < 4; ++u) {                                                              \
                    const float at = i == 0 ? A_[u].x : i == 1 ? A_[u].y : i == 2 ? A_[u].z : A_[u].w;                        \
                    _Pragma("unroll") for (int c = 0; c < NCOL; ++c) acc[u][c] = at * V_[i][c] + acc[u][c];                   \
                }                                                                                                            \
        } while (0)
        float4 aA[4], aB[4];
        float vA[4][NCOL], vB[4][NCOL];
        if (rfull > 0) FA_LD(aA, vA, 0);
        int r = 0;
        for (; r + 8 <= rfull; r += 8) {
            FA_LD(aB, vB, r + 4);
            FA_ACC(aA, vA);
            if (r + 8 < rfull) FA_LD(aA, vA, r + 8);
            FA_ACC(aB, vB);
        }
        if (r < rfull) FA_ACC(aA, vA);                               // rfull is a multiple of 4: one group left
#undef FA_LD
#undef FA_ACC
        for (int r = rfull; r < ttw; ++r) {                          // the diagonal: per-token conditions (uniform)
            float v[NCOL];
            vload(r, v);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (t0 + r <= posu[u]) {
                    const float at = as[(size_t)u * sstride + t0 + r];
#pragma unroll
                    for (int c = 0; c < NCOL; ++c) acc[u][c] = at * v[c] + acc[u][c];
                }
            }
        }
    }
#ifdef FA_TIMING
    fa_t3 = __builtin_readcyclecounter();
    if (lane == 0 && kvh == 0 && (tile % 9) == 0) printf("fa tile %d wave %d: scores %llu softmax %llu pv %llu\n", tile, wave, fa_t1 - fa_t0, fa_t2 - fa_t1, fa_t3 - fa_t2);
#endif
    if (NCOL == 2 && xq_out) {
        // r6: the attention output leaves the kernel as the wo projection's operand (int8 chunks XQ3[k / 16][token slot][16 B] + the scale-operand table
        // of gl3_prefill_gemm3.h) instead of f32 + a quantise launch.  A 32-element block of a token's row = 32 consecutive columns = the 16 lanes of a
        // DPP row, two columns each; Q8_0FloatTensor.java:96-118 arithmetic as quantize_quad_pack.
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int tb = 4 * grp + u;
            float amax = fmaxf(fabsf(acc[u][0]), fabsf(acc[u][NCOL - 1]));
#pragma unroll
            for (int m = 1; m < 16; m <<= 1) amax = fmaxf(amax, __shfl_xor(amax, m, 64));
            const float qs = amax / 127.0f;
            const float ainv = qs != 0.f ? 1.0f / qs : 0.f;
            const float s0 = acc[u][0] * ainv, s1 = acc[u][NCOL - 1] * ainv;
            const uint32_t q0 = (uint32_t)((int)(s0 + copysignf(0.5f, s0)) & 0xFF), q1 = (uint32_t)((int)(s1 + copysignf(0.5f, s1)) & 0xFF);
            if (tb >= nb) continue;
            const int col = head * HS + 2 * lane, b = b0 + tb;
            *reinterpret_cast<uint16_t*>(xq_out + ((size_t)(col >> 4) * xp_tok + b) * 16 + (col & 15)) = (uint16_t)(q0 | (q1 << 8));
            if ((lane & 15) == 0) {
                const float qf = (float)(_Float16)qs;
                const float ahi = __uint_as_float(__float_as_uint(qf) & 0xFFFF0000u), alo = qf - ahi;
                auto pk = [](float h, float l) { return (__float_as_uint(h) >> 16) | (__float_as_uint(l) & 0xFFFF0000u); };
                const uint32_t pr = pk(ahi, alo), n0 = pk(ahi * -8388608.f, alo * -8388608.f), n1 = pk(ahi * -4194304.f, alo * -4194304.f);
                const int blk = col >> 5;
                xp_out[((size_t)blk * 2 + 0) * xp_tok + b] = make_uint4(pr, pr, n0, n0);
                xp_out[((size_t)blk * 2 + 1) * xp_tok + b] = make_uint4(0u, 0u, n1, n1);
            }
        }
        return;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int tb = 4 * grp + u;
        if (tb >= nb) continue;
#pragma unroll
        for (int c = 0; c < NCOL; ++c) {
            const int j = NCOL == 2 ? 2 * lane + c : lane;
            if (j < HS) out[(size_t)(b0 + tb) * out_stride + head * HS + j] = acc[u][c];
        }
    }
}

// r6 — pf_attn_fused_kernel with the inner loops of pf_scores_pk_kernel (phase 1: query rows interleaved in LDS, two tokens' chains per register pair
// on packed f32, reads of the next 8 steps pinned under the current group) and pf_pv_ring_kernel (phase 3: masking by zero weights, the next
// timestep group's LDS reads pinned under the current group's arithmetic).  Same arithmetic in the same order; 16 KB more LDS (query rows).
template <int HS>
__global__ __launch_bounds__(512) void pf_attn_fused2_kernel(const float* __restrict__ Q, int q_stride, const float* __restrict__ kc, const float* __restrict__ vc,
                                                            float* __restrict__ out, int out_stride, int n_kv_heads, int kvmul, int kv_dim,
                                                            int pos0, int ntok, float att_mul, int sstride,
                                                            uint8_t* __restrict__ xq_out = nullptr, uint4* __restrict__ xp_out = nullptr, int xp_tok = 0) {
    extern __shared__ __attribute__((aligned(16))) float fa_sm[];
    constexpr int PITCH = HS + 4, H4 = HS / 4, NCOL = HS > 64 ? 2 : 1;
    float* Ssc = fa_sm;                                             // [kvmul][FA_TB][sstride] score -> softmax rows
    float* kt = fa_sm + (size_t)kvmul * FA_TB * sstride;            // [2][64][PITCH] K (phase 1) / V (phase 3) tiles
    float* sums = kt + 2 * 64 * PITCH;                              // [kvmul * FA_TB] (+ padding to 64 floats)
    float* qs = sums + 64;                                          // [kvmul][FA_TB / 2 pairs][HS][2] query rows, interleaved by token pairs
    const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int nthr = blockDim.x, gthreads = 64 * kvmul;
    const int grp = wave / kvmul, hq = wave % kvmul, gt = t - grp * gthreads;
    const int ntile = (ntok + FA_TB - 1) / FA_TB;
    const int kvh = blockIdx.x % n_kv_heads, tile = ntile - 1 - blockIdx.x / n_kv_heads;
    const int b0 = tile * FA_TB, nb = min(FA_TB, ntok - b0), tmax = pos0 + b0 + nb - 1;
    const int head = kvh * kvmul + hq;
    const float sqrt_hs = (float)sqrt((double)HS);

    // K / V tiles travel global -> registers -> LDS; the next tile's loads are in flight while the current one is consumed (clamped
    // rows: every address is inside the cache, the surplus rows are never read).  8 float4 per thread cover a 64-row tile (host check).
    // (written out at every site: a register array captured by a lambda, or filled in a macro loop, ends up in scratch with this compiler)
    constexpr int NPK = 8;
#ifdef FA_TIMING
    unsigned long long fa_t0 = __builtin_readcyclecounter(), fa_t1, fa_t2, fa_t3;
#endif
    // ---- phase 1: scores
    const int nkt = tmax / 64 + 1;
    float4 pk0, pk1, pk2, pk3, pk4, pk5, pk6, pk7;           // named registers: an array here is not promoted out of scratch
#define FA_REP8(X_) X_(0) X_(1) X_(2) X_(3) X_(4) X_(5) X_(6) X_(7)
    static_assert(NPK == 8, "FA_REP8");
    {
        const int ft0 = min(grp * 64, tmax), frows = max(1, min(64, tmax + 1 - grp * 64));
#define FA_F(U_) { const int fi = min(gt + U_ * gthreads, 64 * H4 - 1), fr = min(fi / H4, frows - 1), fc = fi % H4; \
                        pk##U_ = *reinterpret_cast<const float4*>(kc + (size_t)(ft0 + fr) * kv_dim + kvh * HS + 4 * fc); }
        FA_REP8(FA_F)
#undef FA_F
    }
    // query rows of the tile's tokens -> LDS, interleaved by token pairs: {q_a[j], q_b[j], q_a[j + 1], q_b[j + 1]} is one 16-byte broadcast read
    for (int i = t; i < kvmul * (FA_TB / 2) * H4; i += nthr) {
        const int c = i % H4, pair = (i / H4) % (FA_TB / 2), qh = i / (H4 * (FA_TB / 2));
        const float* qp = Q + (size_t)(kvh * kvmul + qh) * HS + 4 * c;
        const v4f_native_s xa = *reinterpret_cast<const v4f_native_s*>(qp + (size_t)(b0 + min(2 * pair, nb - 1)) * q_stride);
        const v4f_native_s xb = *reinterpret_cast<const v4f_native_s*>(qp + (size_t)(b0 + min(2 * pair + 1, nb - 1)) * q_stride);
        float* d = qs + ((size_t)(qh * (FA_TB / 2) + pair) * HS + 4 * c) * 2;
        *reinterpret_cast<v4f_native_s*>(d) = (v4f_native_s){xa.x, xb.x, xa.y, xb.y};
        *reinterpret_cast<v4f_native_s*>(d + 4) = (v4f_native_s){xa.z, xb.z, xa.w, xb.w};
    }
    for (int trip = 0; 2 * trip < nkt; ++trip) {
        const int t0 = (2 * trip + grp) * 64;
        const bool live = t0 <= tmax;
        const int t1 = min(tmax + 1, t0 + 64);
        float* ktg = kt + grp * 64 * PITCH;
        if (live) {
#define FA_P(U_) { const int fi = gt + U_ * gthreads; if (fi < 64 * H4) *reinterpret_cast<float4*>(ktg + (fi / H4) * PITCH + 4 * (fi % H4)) = pk##U_; }
            FA_REP8(FA_P)
#undef FA_P
        }
        __syncthreads();
        {
            const int tn = t0 + 128;                                 // this group's next tile (clamped: fetched even if it is not used)
            {
                const int ft0 = min(tn, tmax), frows = max(1, min(64, tmax + 1 - tn));
#define FA_F(U_) { const int fi = min(gt + U_ * gthreads, 64 * H4 - 1), fr = min(fi / H4, frows - 1), fc = fi % H4; \
                                pk##U_ = *reinterpret_cast<const float4*>(kc + (size_t)(ft0 + fr) * kv_dim + kvh * HS + 4 * fc); }
                FA_REP8(FA_F)
#undef FA_F
            }
        }
        if (live) {
            v2f_native kr[HS / 2];
#pragma unroll
            for (int c = 0; c < H4; ++c) {
                const v4f_native_s x = *reinterpret_cast<const v4f_native_s*>(ktg + min(lane, t1 - t0 - 1) * PITCH + 4 * c);
                kr[2 * c] = x.xy; kr[2 * c + 1] = x.zw;
            }
            for (int pp = 0; 4 * pp < nb; ++pp) {                    // four tokens = two packed chains per pass (pf_scores_pk_kernel's inner loop)
                const float* q01 = qs + (size_t)(hq * (FA_TB / 2) + 2 * pp) * HS * 2;
                const float* q23 = q01 + HS * 2;
                v2f_native c0 = {0.f, 0.f}, c1 = {0.f, 0.f};
                v4f_native_s qa[8], qb[8];
                SPK_LD(0, qa); SPK_LD(1, qb); __builtin_amdgcn_sched_barrier(0);
                static_for<0, HS / 8, 2>([&](auto gc) {
                    constexpr int g = decltype(gc)::value;
                    SPK_ACC(g, qa);
                    SPK_LD((g + 2 < HS / 8 ? g + 2 : HS / 8 - 1), qa); __builtin_amdgcn_sched_barrier(0);
                    SPK_ACC(g + 1, qb);
                    SPK_LD((g + 3 < HS / 8 ? g + 3 : HS / 8 - 1), qb); __builtin_amdgcn_sched_barrier(0);
                });
                const float sv[4] = {c0.x, c0.y, c1.x, c1.y};
                const int ts = t0 + lane;
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int tb = 4 * pp + u;
                    if (tb < nb && ts <= pos0 + b0 + tb) Ssc[(size_t)(hq * FA_TB + tb) * sstride + ts] = att_mul != 0.f ? sv[u] * att_mul : sv[u] / sqrt_hs;
                }
            }
        }
        __syncthreads();
    }

#ifdef FA_TIMING
    fa_t1 = __builtin_readcyclecounter();
#endif
    // the first V tile travels while the softmax runs
    {
        const int ft0 = 0, frows = min(64, tmax + 1);
#define FA_F(U_) { const int fi = min(t + U_ * nthr, 64 * H4 - 1), fr = min(fi / H4, frows - 1), fc = fi % H4; \
                        pk##U_ = *reinterpret_cast<const float4*>(vc + (size_t)(ft0 + fr) * kv_dim + kvh * HS + 4 * fc); }
        FA_REP8(FA_F)
#undef FA_F
    }
    // ---- phase 2: softmax of the kvmul * nb rows
    const int nrows = kvmul * nb, nwaves = nthr >> 6;
    for (int row = wave; row < nrows; row += nwaves) {
        const int tb = row % nb, n = pos0 + b0 + tb + 1;
        float* e = Ssc + (size_t)((row / nb) * FA_TB + tb) * sstride;
        float mx = -INFINITY;
        for (int i = lane; i < n; i += 64) mx = fmaxf(mx, e[i]);
        mx = wave_max(mx);
        for (int i = lane; i < n; i += 64) e[i] = (float)exp((double)(e[i] - mx));      // lane-private slots
    }
    __syncthreads();
    if (wave == 0 && lane < nrows) {                                 // lane = row: the strictly sequential sums, side by side
        const int tb = lane % nb, n = pos0 + b0 + tb + 1;
        const float* e = Ssc + (size_t)((lane / nb) * FA_TB + tb) * sstride;
        sums[lane] = seq_sum_lds_ring(e, n);                         // reads pinned three groups ahead (gl3_decode_kernels.h)
    }
    __syncthreads();
    for (int row = wave; row < nrows; row += nwaves) {
        const int tb = row % nb, n = pos0 + b0 + tb + 1;
        float* e = Ssc + (size_t)((row / nb) * FA_TB + tb) * sstride;
        const float sum = sums[row];
        // weight 0 behind the token's position (phase 3 masks by weight): up to where the wavefront that carries this token can read — its last
        // token sits at most 3 positions further, rounded up to a group of four, inside the tile's last 64-timestep block
        const int zend = min((n + 4 + 63) & ~63, (tmax + 1 + 63) & ~63);
        for (int i = lane; i < zend; i += 64) e[i] = i < n ? e[i] / sum : 0.f;
    }

#ifdef FA_TIMING
    fa_t2 = __builtin_readcyclecounter();
#endif
    // ---- phase 3: weighted V sum; wavefront = (query head hq, tokens 4 * grp .. + 3)
    const int wmax = 4 * grp < nb ? pos0 + b0 + min(4 * grp + 3, nb - 1) : -1;
    const float* as = Ssc + (size_t)(hq * FA_TB + 4 * grp) * sstride;                             // rows of this wavefront's four tokens
    float acc[4][NCOL];
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int c = 0; c < NCOL; ++c) acc[u][c] = 0.f;
    int vb = 0;
    for (int t0 = 0; t0 <= tmax; t0 += 64, vb ^= 1) {
        const int tt = min(64, tmax + 1 - t0);
        float* vt = kt + vb * 64 * PITCH;
        {
#define FA_P(U_) { const int fi = t + U_ * nthr; if (fi < 64 * H4) *reinterpret_cast<float4*>(vt + (fi / H4) * PITCH + 4 * (fi % H4)) = pk##U_; }
            FA_REP8(FA_P)
#undef FA_P
        }
        __syncthreads();                                             // (also orders phase 2's writes before the first reads of `as`)
        {
            const int ft0 = min(t0 + 64, tmax), frows = max(1, min(64, tmax + 1 - (t0 + 64)));
#define FA_F(U_) { const int fi = min(t + U_ * nthr, 64 * H4 - 1), fr = min(fi / H4, frows - 1), fc = fi % H4; \
                            pk##U_ = *reinterpret_cast<const float4*>(vc + (size_t)(ft0 + fr) * kv_dim + kvh * HS + 4 * fc); }
            FA_REP8(FA_F)
#undef FA_F
        }
        // groups of four timesteps; the next group's weights and V rows are read from LDS under the current group's arithmetic (pinned).
        // Timesteps behind a token's position carry the weight 0 (0 * v + acc = acc exactly; every staged V row is a written row).
        const int ng = max(0, min(64, wmax + 1 - t0) + 3) >> 2;
        const float* ap = as + t0;
        const float* vp = vt + (NCOL == 2 ? 2 * lane : min(lane, HS - 1));
        float4 wa0, wa1, wa2, wa3, wb0, wb1, wb2, wb3;
        float va[4][NCOL], vb4[4][NCOL];
#define FA2_LD(G_, W0_, W1_, W2_, W3_, V_) do { const int r_ = 4 * min((G_), 15); \
            W0_ = *reinterpret_cast<const float4*>(ap + r_); W1_ = *reinterpret_cast<const float4*>(ap + (size_t)sstride + r_); \
            W2_ = *reinterpret_cast<const float4*>(ap + 2 * (size_t)sstride + r_); W3_ = *reinterpret_cast<const float4*>(ap + 3 * (size_t)sstride + r_); \
            _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_) { \
                if (NCOL == 2) { const float2 x_ = *reinterpret_cast<const float2*>(vp + (r_ + i_) * PITCH); V_[i_][0] = x_.x; V_[i_][NCOL - 1] = x_.y; } \
                else V_[i_][0] = vp[(r_ + i_) * PITCH]; } } while (0)
#define FA2_STEP(I_, WX_, V_, W0_, W1_, W2_, W3_) do { const float w_[4] = {W0_.WX_, W1_.WX_, W2_.WX_, W3_.WX_}; \
            _Pragma("unroll") for (int u_ = 0; u_ < 4; ++u_) _Pragma("unroll") for (int c_ = 0; c_ < NCOL; ++c_) acc[u_][c_] = w_[u_] * V_[I_][c_] + acc[u_][c_]; } while (0)
#define FA2_ACC(V_, W0_, W1_, W2_, W3_) do { FA2_STEP(0, x, V_, W0_, W1_, W2_, W3_); FA2_STEP(1, y, V_, W0_, W1_, W2_, W3_); \
            FA2_STEP(2, z, V_, W0_, W1_, W2_, W3_); FA2_STEP(3, w, V_, W0_, W1_, W2_, W3_); } while (0)
        FA2_LD(0, wa0, wa1, wa2, wa3, va); FA2_LD(1, wb0, wb1, wb2, wb3, vb4); __builtin_amdgcn_sched_barrier(0);
        int g = 0;
        for (; g + 2 <= ng; g += 2) {
            FA2_ACC(va, wa0, wa1, wa2, wa3);
            FA2_LD(g + 2, wa0, wa1, wa2, wa3, va); __builtin_amdgcn_sched_barrier(0);
            FA2_ACC(vb4, wb0, wb1, wb2, wb3);
            FA2_LD(g + 3, wb0, wb1, wb2, wb3, vb4); __builtin_amdgcn_sched_barrier(0);
        }
        if (g < ng) FA2_ACC(va, wa0, wa1, wa2, wa3);
#undef FA2_LD
#undef FA2_STEP
#undef FA2_ACC
    }
#ifdef FA_TIMING
    fa_t3 = __builtin_readcyclecounter();
    if (lane == 0 && kvh == 0 && (tile % 9) == 0) printf("fa tile %d wave %d: scores %llu softmax %llu pv %llu\n", tile, wave, fa_t1 - fa_t0, fa_t2 - fa_t1, fa_t3 - fa_t2);
#endif
    if (NCOL == 2 && xq_out) {
        // r6: the attention output leaves the kernel as the wo projection's operand (int8 chunks XQ3[k / 16][token slot][16 B] + the scale-operand table
        // of gl3_prefill_gemm3.h) instead of f32 + a quantise launch.  A 32-element block of a token's row = 32 consecutive columns = the 16 lanes of a
        // DPP row, two columns each; Q8_0FloatTensor.java:96-118 arithmetic as quantize_quad_pack.
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int tb = 4 * grp + u;
            float amax = fmaxf(fabsf(acc[u][0]), fabsf(acc[u][NCOL - 1]));
#pragma unroll
            for (int m = 1; m < 16; m <<= 1) amax = fmaxf(amax, __shfl_xor(amax, m, 64));
            const float qs = amax / 127.0f;
            const float ainv = qs != 0.f ? 1.0f / qs : 0.f;
            const float s0 = acc[u][0] * ainv, s1 = acc[u][NCOL - 1] * ainv;
            const uint32_t q0 = (uint32_t)((int)(s0 + copysignf(0.5f, s0)) & 0xFF), q1 = (uint32_t)((int)(s1 + copysignf(0.5f, s1)) & 0xFF);
            if (tb >= nb) continue;
            const int col = head * HS + 2 * lane, b = b0 + tb;
            *reinterpret_cast<uint16_t*>(xq_out + ((size_t)(col >> 4) * xp_tok + b) * 16 + (col & 15)) = (uint16_t)(q0 | (q1 << 8));
            if ((lane & 15) == 0) {
                const float qf = (float)(_Float16)qs;
                const float ahi = __uint_as_float(__float_as_uint(qf) & 0xFFFF0000u), alo = qf - ahi;
                auto pk = [](float h, float l) { return (__float_as_uint(h) >> 16) | (__float_as_uint(l) & 0xFFFF0000u); };
                const uint32_t pr = pk(ahi, alo), n0 = pk(ahi * -8388608.f, alo * -8388608.f), n1 = pk(ahi * -4194304.f, alo * -4194304.f);
                const int blk = col >> 5;
                xp_out[((size_t)blk * 2 + 0) * xp_tok + b] = make_uint4(pr, pr, n0, n0);
                xp_out[((size_t)blk * 2 + 1) * xp_tok + b] = make_uint4(0u, 0u, n1, n1);
            }
        }
        return;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int tb = 4 * grp + u;
        if (tb >= nb) continue;
#pragma unroll
        for (int c = 0; c < NCOL; ++c) {
            const int j = NCOL == 2 ? 2 * lane + c : lane;
            if (j < HS) out[(size_t)(b0 + tb) * out_stride + head * HS + j] = acc[u][c];
        }
    }
}

#undef FA_REP8

// Greedy id per sequence: first index of the maximum of each logits row (FloatTensor.argmax :138-151).
// logits: rank-chunked [tp][rows][n / tp] (cc = n / tp, a multiple of 4; tp = 1: plain rows).
// Two launches: (AMX_SPLIT segments x rows) workgroups scan their segment with float4 loads -> one (value, index) pair each; one
// wavefront per row folds the pairs.  (Round 2: one 1024-thread workgroup per row with scalar strided loads, 75 us per step for a
// 19 MB scan at B = 32 — 30x its HBM time.)
constexpr int AMX_SPLIT = 32;
__device__ __forceinline__ void amx_fold(float& best, int& idx, float ob, int oi) {
    if (ob > best || (ob == best && oi < idx)) { best = ob; idx = oi; }
}
__global__ __launch_bounds__(256) void pf_argmax_part_kernel(const float* __restrict__ logits, int n, int cc, float* __restrict__ pv, int* __restrict__ pi) {
    __shared__ float bv[4];
    __shared__ int bi[4];
    const int t = threadIdx.x, seg = blockIdx.x, row = blockIdx.y, nrows = gridDim.y;
    const int nq = n >> 2, per = (nq + AMX_SPLIT - 1) / AMX_SPLIT;
    const int q0 = seg * per, q1 = min(nq, q0 + per);
    float best = -INFINITY;
    int idx = 0x7FFFFFFF;
    for (int q = q0 + t; q < q1; q += 256) {
        const int i = 4 * q;
        const float4 f = *reinterpret_cast<const float4*>(logits + chunked(row, i, cc, nrows));
        if (f.x > best) { best = f.x; idx = i; }            // ascending i inside a thread: strict > keeps the first maximum
        if (f.y > best) { best = f.y; idx = i + 1; }
        if (f.z > best) { best = f.z; idx = i + 2; }
        if (f.w > best) { best = f.w; idx = i + 3; }
    }
    if (seg == AMX_SPLIT - 1)                               // n is a multiple of 16 in every supported shape; kept for safety
        for (int i = 4 * nq + t; i < n; i += 256) { const float f = logits[chunked(row, i, cc, nrows)]; if (f > best) { best = f; idx = i; } }
    for (int m = 32; m >= 1; m >>= 1) amx_fold(best, idx, __shfl_xor(best, m, 64), __shfl_xor(idx, m, 64));
    if ((t & 63) == 0) { bv[t >> 6] = best; bi[t >> 6] = idx; }
    __syncthreads();
    if (t == 0) {
        for (int w = 1; w < 4; ++w) amx_fold(best, idx, bv[w], bi[w]);
        pv[row * AMX_SPLIT + seg] = best; pi[row * AMX_SPLIT + seg] = idx;
    }
}
__global__ __launch_bounds__(64) void pf_argmax_fold_kernel(const float* __restrict__ pv, const int* __restrict__ pi, int32_t* __restrict__ out) {
    const int row = blockIdx.x, t = threadIdx.x;
    float best = t < AMX_SPLIT ? pv[row * AMX_SPLIT + t] : -INFINITY;
    int idx = t < AMX_SPLIT ? pi[row * AMX_SPLIT + t] : 0x7FFFFFFF;
    for (int m = 32; m >= 1; m >>= 1) amx_fold(best, idx, __shfl_xor(best, m, 64), __shfl_xor(idx, m, 64));
    if (t == 0) out[row] = idx == 0x7FFFFFFF ? 0 : idx;
}

// ------------------------------------------------------------------------------------------------ host side
// r6 — the one-launch prefill attention with the PRODUCTS of phases 1 and 3 on the matrix pipe (kvMul 4: 32 (head, token) rows per workgroup).
// pf_attn_fused_kernel / fused2 feed one operand of every multiply from a wavefront-uniform place (SGPRs: lead bounded by the SGPR file; LDS:
// a uniform-address ds_read_b128 costs 9.2 cycles of the CU's LDS pipe against 4.9 for 64 distinct addresses, scripts/probes/lds_bcast_probe.hip)
// and both phases end up bound by that delivery.  A K = 1 f32 MFMA with C = 0 is an outer product of two LANE-DISTINCT vectors whose every
// element is rounded once (D = fma(a, b, 0) = fl(a * b): the property gemm_vlq_mfma_kernel uses): v_mfma_f32_16x16x1_4b_f32 = four independent
// 16 x 16 blocks per instruction, 1024 rounded products in 32 cycles, no broadcast anywhere; the VALU keeps the ordered adds (packed).
//   phase 1  wavefront = 16 timesteps of the K tile x all 32 rows.  Block q = (row group q & 1, step parity q >> 1): A = k[t][2 m + parity]
//            (the lane's K row, every second element, in registers), B = q[row][2 m + parity] (one ds_read_b32 of 64 distinct addresses).
//            Per MFMA the chains advance two steps: s = (s + P_even) + P_odd, j ascending.  64 MFMAs + 512 packed adds per tile and wavefront.
//   phase 3  wavefront = 16 rows x 32 columns.  Block q = (column group q & 1, timestep parity q >> 1): A = w[row][t + parity] (softmax weight,
//            0 behind the row's position), B = v[t + parity][column]; acc = (acc + P_t) + P_t+1, t ascending.  8 accumulator registers.
// Same arithmetic, same order, same roundings as InferenceCore.java:98-137; phase 2 is pf_attn_fused_kernel's.
__host__ __device__ constexpr size_t fa3_smem_bytes(int hs, int sstride) {
    return ((size_t)4 * FA_TB * sstride + 2 * 64 * (hs + 4) + 64 + (size_t)4 * FA_TB * (hs + 2)) * 4;
}
typedef float v8f_native __attribute__((ext_vector_type(8)));
template <int HS>
__global__ __launch_bounds__(512) void pf_attn_fused3_kernel(const float* __restrict__ Q, int q_stride, const float* __restrict__ kc, const float* __restrict__ vc,
                                                             float* __restrict__ out, int out_stride, int n_kv_heads, int kv_dim,
                                                             int pos0, int ntok, float att_mul, int sstride,
                                                             uint8_t* __restrict__ xq_out, uint4* __restrict__ xp_out, int xp_tok) {
    extern __shared__ __attribute__((aligned(16))) float fa_sm[];
    constexpr int KVM = 4, ROWS = KVM * FA_TB, PITCH = HS + 4, H4 = HS / 4, QP = HS + 2, NM = HS / 2, kvmul = KVM;
    static_assert(ROWS == 32 && NM % 16 == 0, "two row groups of 16; operand ring of 8 MFMAs");
    float* Ssc = fa_sm;                                             // [ROWS][sstride] score -> softmax rows, row = head * FA_TB + token
    float* kt = fa_sm + (size_t)ROWS * sstride;                     // [2][64][PITCH] K (phase 1) / V (phase 3) tiles
    float* sums = kt + 2 * 64 * PITCH;                              // [ROWS] (+ padding to 64 floats)
    float* qs = sums + 64;                                          // [ROWS][QP] query rows
    const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
    constexpr int nthr = 512, gthreads = 256;
    const int grp = wave >> 2, wg = wave & 3, gt = t - grp * gthreads;
    const int lq = lane >> 4, li = lane & 15, par = lq >> 1;        // MFMA block of this lane's operands, index inside it, step parity of the block
    const int ntile = (ntok + FA_TB - 1) / FA_TB;
    const int kvh = blockIdx.x % n_kv_heads, tile = ntile - 1 - blockIdx.x / n_kv_heads;
    const int b0 = tile * FA_TB, nb = min(FA_TB, ntok - b0), tmax = pos0 + b0 + nb - 1;
    const float sqrt_hs = (float)sqrt((double)HS);
    const v16f_t zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    constexpr int NPK = 8;
    const int nkt = tmax / 64 + 1;
    float4 pk0, pk1, pk2, pk3, pk4, pk5, pk6, pk7;           // named registers: an array here is not promoted out of scratch
#define FA_REP8(X_) X_(0) X_(1) X_(2) X_(3) X_(4) X_(5) X_(6) X_(7)
    static_assert(NPK == 8, "FA_REP8");
    {
        const int ft0 = min(grp * 64, tmax), frows = max(1, min(64, tmax + 1 - grp * 64));
#define FA_F(U_) { const int fi = min(gt + U_ * gthreads, 64 * H4 - 1), fr = min(fi / H4, frows - 1), fc = fi % H4; \
                        pk##U_ = *reinterpret_cast<const float4*>(kc + (size_t)(ft0 + fr) * kv_dim + kvh * HS + 4 * fc); }
        FA_REP8(FA_F)
#undef FA_F
    }
    for (int i = t; i < ROWS * H4; i += nthr) {                      // query rows (tokens past the chunk's end repeat its last token: never stored)
        const int row = i / H4, c = i % H4;
        const float4 x = *reinterpret_cast<const float4*>(Q + (size_t)(b0 + min(row & (FA_TB - 1), nb - 1)) * q_stride + (size_t)(kvh * kvmul + (row >> 3)) * HS + 4 * c);
        float* d = qs + row * QP + 4 * c;
        *reinterpret_cast<float2*>(d) = make_float2(x.x, x.y);
        *reinterpret_cast<float2*>(d + 2) = make_float2(x.z, x.w);
    }
#ifdef FA_TIMING
    unsigned long long fa_t0 = __builtin_readcyclecounter(), fa_t1, fa_t2, fa_t3;
#endif
    // ---- phase 1: scores
    for (int trip = 0; 2 * trip < nkt; ++trip) {
        const int t0 = (2 * trip + grp) * 64;
        const bool live = t0 <= tmax;
        const int t1 = min(tmax + 1, t0 + 64);
        float* ktg = kt + grp * 64 * PITCH;
        if (live) {
#define FA_P(U_) { const int fi = gt + U_ * gthreads; if (fi < 64 * H4) *reinterpret_cast<float4*>(ktg + (fi / H4) * PITCH + 4 * (fi % H4)) = pk##U_; }
            FA_REP8(FA_P)
#undef FA_P
        }
        __syncthreads();
        {
            const int tn = t0 + 128;                                 // this group's next tile (clamped: fetched even if it is not used)
            const int ft0 = min(tn, tmax), frows = max(1, min(64, tmax + 1 - tn));
#define FA_F(U_) { const int fi = min(gt + U_ * gthreads, 64 * H4 - 1), fr = min(fi / H4, frows - 1), fc = fi % H4; \
                            pk##U_ = *reinterpret_cast<const float4*>(kc + (size_t)(ft0 + fr) * kv_dim + kvh * HS + 4 * fc); }
            FA_REP8(FA_F)
#undef FA_F
        }
        if (live && t0 + 16 * wg <= tmax) {                          // this wavefront's 16 timesteps hold at least one attended position
            const float* krow = ktg + min(16 * wg + li, t1 - t0 - 1) * PITCH;
            float kreg[NM];                                          // k[t][2 m + parity], m ascending
#pragma unroll
            for (int c = 0; c < H4; ++c) {
                const v4f_native_s x = *reinterpret_cast<const v4f_native_s*>(krow + 4 * c);
                kreg[2 * c] = par ? x.y : x.x; kreg[2 * c + 1] = par ? x.w : x.z;
            }
            const float* qrow = qs + (16 * (lq & 1) + li) * QP + par;
            v8f_native sc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};      // [0..3] row li, [4..7] row 16 + li; the four timesteps 16 wg + 4 lq + r
            float qa[8], qb[8];
#define F3_LDQ(G_, R_) do { _Pragma("unroll") for (int u_ = 0; u_ < 8; ++u_) R_[u_] = qrow[2 * (8 * (G_) + u_)]; } while (0)
            // blocks 0 / 1 = row groups 0 / 1 at the even step, blocks 2 / 3 at the odd step: two dependent adds per chain and MFMA
#define F3_ADD(P_, S_) do { S_ = S_ + __builtin_shufflevector(P_, P_, 0, 1, 2, 3, 4, 5, 6, 7); S_ = S_ + __builtin_shufflevector(P_, P_, 8, 9, 10, 11, 12, 13, 14, 15); } while (0)
            // one MFMA ahead of the adds that consume the previous one (two product registers, order pinned: left alone the scheduler issues a whole
            // group's MFMAs first and spills their 8 x 16 result registers)
#define F3_MF8(G_, R_, RN_) do { _Pragma("unroll") for (int u_ = 0; u_ < 8; ++u_) { \
                const int mn_ = 8 * (G_) + u_ + 1;      /* a constant after unrolling */ \
                v16f_t Pn_ = Pc; \
                if (mn_ < NM) Pn_ = __builtin_amdgcn_mfma_f32_16x16x1f32(kreg[mn_ < NM ? mn_ : 0], u_ < 7 ? R_[u_ < 7 ? u_ + 1 : 0] : RN_[0], zero16, 0, 0, 0); \
 __builtin_amdgcn_sched_barrier(0); \
                F3_ADD(Pc, sc); asm volatile("" : "+v"(sc)); \
                __builtin_amdgcn_sched_barrier(0); \
                Pc = Pn_; } } while (0)
            F3_LDQ(0, qa); F3_LDQ(1, qb); __builtin_amdgcn_sched_barrier(0);
            v16f_t Pc = __builtin_amdgcn_mfma_f32_16x16x1f32(kreg[0], qa[0], zero16, 0, 0, 0);
            static_for<0, NM / 8, 2>([&](auto gc) {
                constexpr int g = decltype(gc)::value;
                F3_MF8(g, qa, qb);
                F3_LDQ((g + 2 < NM / 8 ? g + 2 : NM / 8 - 1), qa); __builtin_amdgcn_sched_barrier(0);
                F3_MF8(g + 1, qb, qa);
                F3_LDQ((g + 3 < NM / 8 ? g + 3 : NM / 8 - 1), qb); __builtin_amdgcn_sched_barrier(0);
            });
#undef F3_LDQ
#undef F3_MF8
            // lane (lq, li) holds rows li and 16 + li at the timesteps 16 wg + 4 lq + r
            const float sv[2][4] = {{sc[0], sc[1], sc[2], sc[3]}, {sc[4], sc[5], sc[6], sc[7]}};
#pragma unroll
            for (int rgp = 0; rgp < 2; ++rgp) {
                const int row = 16 * rgp + li, tb = row & (FA_TB - 1);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int ts = t0 + 16 * wg + 4 * lq + r;
                    if (tb < nb && ts <= pos0 + b0 + tb) Ssc[(size_t)row * sstride + ts] = att_mul != 0.f ? sv[rgp][r] * att_mul : sv[rgp][r] / sqrt_hs;
                }
            }
        }
        __syncthreads();
    }
#ifdef FA_TIMING
    fa_t1 = __builtin_readcyclecounter();
#endif
    // the first V tile travels while the softmax runs
    {
        const int frows = min(64, tmax + 1);
#define FA_F(U_) { const int fi = min(t + U_ * nthr, 64 * H4 - 1), fr = min(fi / H4, frows - 1), fc = fi % H4; \
                        pk##U_ = *reinterpret_cast<const float4*>(vc + (size_t)fr * kv_dim + kvh * HS + 4 * fc); }
        FA_REP8(FA_F)
#undef FA_F
    }
    // ---- phase 2: softmax of the kvmul * nb rows (pf_attn_fused_kernel's; weights behind a row's position are written as 0: phase 3 masks by weight)
    const int nrows = kvmul * nb, nwaves = nthr >> 6;
    for (int row = wave; row < nrows; row += nwaves) {
        const int tb = row % nb, n = pos0 + b0 + tb + 1;
        float* e = Ssc + (size_t)((row / nb) * FA_TB + tb) * sstride;
        float mx = -INFINITY;
        for (int i = lane; i < n; i += 64) mx = fmaxf(mx, e[i]);
        mx = wave_max(mx);
        for (int i = lane; i < n; i += 64) e[i] = (float)exp((double)(e[i] - mx));      // lane-private slots
    }
    __syncthreads();
    if (wave == 0 && lane < nrows) {                                 // lane = row: the strictly sequential sums, side by side
        const int tb = lane % nb, n = pos0 + b0 + tb + 1;
        const float* e = Ssc + (size_t)((lane / nb) * FA_TB + tb) * sstride;
        sums[lane] = seq_sum_lds_ring(e, n);
    }
    __syncthreads();
    const int tend = (tmax + 1 + 63) & ~63;
    for (int row = wave; row < ROWS; row += nwaves) {                // every row of the tile: rows of tokens past the chunk's end become all-zero weights
        const int tb = row & (FA_TB - 1), hqr = row >> 3;
        float* e = Ssc + (size_t)row * sstride;
        const int n = tb < nb ? pos0 + b0 + tb + 1 : 0;
        const float sum = tb < nb ? sums[hqr * nb + tb] : 1.f;
        for (int i = lane; i < tend; i += 64) e[i] = i < n ? e[i] / sum : 0.f;
    }
#ifdef FA_TIMING
    fa_t2 = __builtin_readcyclecounter();
#endif
    // ---- phase 3: weighted V sum; wavefront = (row group rg, 32 columns cq)
    const int rg = wave & 1, cq = wave >> 1, cg = lq & 1;
    const bool pv_live = cq < HS / 32;
    v8f_native ac = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};      // [0..3] column 32 cq + li, [4..7] column 32 cq + 16 + li; rows 16 rg + 4 lq + r
    const float* wrow = Ssc + (size_t)(16 * rg + li) * sstride + par;
    int vb = 0;
    for (int t0 = 0; t0 <= tmax; t0 += 64, vb ^= 1) {
        float* vt = kt + vb * 64 * PITCH;
        {
#define FA_P(U_) { const int fi = t + U_ * nthr; if (fi < 64 * H4) *reinterpret_cast<float4*>(vt + (fi / H4) * PITCH + 4 * (fi % H4)) = pk##U_; }
            FA_REP8(FA_P)
#undef FA_P
        }
        __syncthreads();                                             // (also orders phase 2's writes before the first reads of the weights)
        {
            const int ft0 = min(t0 + 64, tmax), frows = max(1, min(64, tmax + 1 - (t0 + 64)));
#define FA_F(U_) { const int fi = min(t + U_ * nthr, 64 * H4 - 1), fr = min(fi / H4, frows - 1), fc = fi % H4; \
                            pk##U_ = *reinterpret_cast<const float4*>(vc + (size_t)(ft0 + fr) * kv_dim + kvh * HS + 4 * fc); }
            FA_REP8(FA_F)
#undef FA_F
        }
        if (pv_live) {
            const int npair = (min(64, tmax + 1 - t0) + 1) >> 1;     // timestep pairs of the tile that hold an attended position (weights behind: 0)
            const float* wp = wrow + t0;
            const float* vp = vt + par * PITCH + 32 * cq + 16 * cg + li;
            float wa[8], va[8], wb[8], vb8[8];
#define F3_LDV(G_, W_, V_) do { _Pragma("unroll") for (int u_ = 0; u_ < 8; ++u_) { const int m_ = min(8 * (G_) + u_, 31); W_[u_] = wp[2 * m_]; V_[u_] = vp[2 * m_ * PITCH]; } } while (0)
#define F3_PV8(W_, V_) do { v16f_t Pc_ = __builtin_amdgcn_mfma_f32_16x16x1f32(W_[0], V_[0], zero16, 0, 0, 0); \
                _Pragma("unroll") for (int u_ = 0; u_ < 8; ++u_) { \
                    v16f_t Pn_ = Pc_; \
                    if (u_ < 7) Pn_ = __builtin_amdgcn_mfma_f32_16x16x1f32(W_[u_ < 7 ? u_ + 1 : 0], V_[u_ < 7 ? u_ + 1 : 0], zero16, 0, 0, 0); \
 __builtin_amdgcn_sched_barrier(0); \
                    F3_ADD(Pc_, ac); asm volatile("" : "+v"(ac)); \
                    __builtin_amdgcn_sched_barrier(0); \
                    Pc_ = Pn_; } } while (0)
            // groups of 8 MFMAs = 16 timesteps; a group past the tile's last attended pair multiplies zero weights (exact: acc + 0), so the trip
            // count is rounded up to whole groups only where needed
            const int ngr = (npair + 7) >> 3;
            F3_LDV(0, wa, va); F3_LDV(1, wb, vb8); __builtin_amdgcn_sched_barrier(0);
            int g = 0;
            for (; g + 2 <= ngr; g += 2) {
                F3_PV8(wa, va);
                F3_LDV(g + 2, wa, va); __builtin_amdgcn_sched_barrier(0);
                F3_PV8(wb, vb8);
                F3_LDV(g + 3, wb, vb8); __builtin_amdgcn_sched_barrier(0);
            }
            if (g < ngr) F3_PV8(wa, va);
#undef F3_LDV
#undef F3_PV8
        }
    }
#undef F3_ADD
#undef FA_REP8
#ifdef FA_TIMING
    fa_t3 = __builtin_readcyclecounter();
    if (lane == 0 && kvh == 0 && (tile % 9) == 0) printf("fa tile %d wave %d: scores %llu softmax %llu pv %llu\n", tile, wave, fa_t1 - fa_t0, fa_t2 - fa_t1, fa_t3 - fa_t2);
#endif
    if (!pv_live) return;
    const float av[2][4] = {{ac[0], ac[1], ac[2], ac[3]}, {ac[4], ac[5], ac[6], ac[7]}};
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = 16 * rg + 4 * lq + r, tb = row & (FA_TB - 1), head = kvh * kvmul + (row >> 3), b = b0 + tb;
        const int col = head * HS + 32 * cq + li;                    // av[0][r]; av[1][r] sits 16 columns further
        if (xq_out) {
            // the wo projection's operand (int8 chunks XQ3[k / 16][token slot][16 B] + the scale-operand table of gl3_prefill_gemm3.h): this wavefront's
            // 32 columns of a row are one Q8_0 block = the 16 lanes of a DPP row x 2 registers (Q8_0FloatTensor.java:96-118 arithmetic)
            float amax = fmaxf(fabsf(av[0][r]), fabsf(av[1][r]));
            amax = row8_max(amax); GL3_DPP_MAX(amax, 0x140);
            const float qsc = amax / 127.0f;
            const float ainv = qsc != 0.f ? 1.0f / qsc : 0.f;
            const float s0 = av[0][r] * ainv, s1 = av[1][r] * ainv;
            const uint8_t q0 = (uint8_t)((int)(s0 + copysignf(0.5f, s0)) & 0xFF), q1 = (uint8_t)((int)(s1 + copysignf(0.5f, s1)) & 0xFF);
            if (tb >= nb) continue;
            xq_out[((size_t)(col >> 4) * xp_tok + b) * 16 + li] = q0;
            xq_out[((size_t)((col >> 4) + 1) * xp_tok + b) * 16 + li] = q1;
            if (li == 0) {
                const float qf = (float)(_Float16)qsc;
                const float ahi = __uint_as_float(__float_as_uint(qf) & 0xFFFF0000u), alo = qf - ahi;
                auto pk = [](float h, float l) { return (__float_as_uint(h) >> 16) | (__float_as_uint(l) & 0xFFFF0000u); };
                const uint32_t pr = pk(ahi, alo), n0 = pk(ahi * -8388608.f, alo * -8388608.f), n1 = pk(ahi * -4194304.f, alo * -4194304.f);
                const int blk = col >> 5;
                xp_out[((size_t)blk * 2 + 0) * xp_tok + b] = make_uint4(pr, pr, n0, n0);
                xp_out[((size_t)blk * 2 + 1) * xp_tok + b] = make_uint4(0u, 0u, n1, n1);
            }
        } else if (tb < nb) {
            out[(size_t)b * out_stride + col] = av[0][r];
            out[(size_t)b * out_stride + col + 16] = av[1][r];
        }
    }
}

// r6 — the weighted V sum behind a long context with its products on the matrix pipe (phase 3 of pf_attn_fused3_kernel as a kernel of its own; kvMul 4).
// Workgroup = (kv head, 16 tokens) = four row groups (one per query head) x HS / 32 column slices = 16 wavefronts; a wavefront advances its
// 16 rows x 32 columns two timesteps per MFMA: A = w[row][t + parity] (numerator / sum, 0 behind the row's position: staged that way), B = v[t + parity]
// [column], acc = (acc + P_t) + P_t+1.  Operands are lane-distinct 4-byte LDS reads (pf_pv_ring_kernel's uniform-address weight reads kept the
// LDS pipe 68 % busy and bound it).  Staging as pf_pv_ring_kernel: the next tile's V rows and numerators travel in registers under the current
// tile's arithmetic.
constexpr int PVM_TB = 16, PVM_WP = 68;
template <int HS>
__global__ __launch_bounds__(1024) void pf_pv_mfma_kernel(const PfAttnArgs a, int seq, int pos0, int ntok, const float* __restrict__ sums) {
    constexpr int PITCH = HS + 4, H4 = HS / 4, NT = 1024, VPT = 64 * H4 / NT, KVM = 4;
    static_assert(VPT >= 1, "a V tile is at least one 16-byte slot per thread");
    extern __shared__ __attribute__((aligned(16))) float vt[];        // [64][PITCH] V rows, then [4 heads x 16 tokens][PVM_WP] weights
    float* ws = vt + 64 * PITCH;
    const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int rg = wave & 3, cq = wave >> 2;                          // query head of the kv group, 32-column slice
    const int lq = lane >> 4, li = lane & 15, par = lq >> 1, cg = lq & 1;
    const int kvh = blockIdx.x, b0 = blockIdx.y * PVM_TB;
    const int nb = min(PVM_TB, ntok - b0);
    const int tmax = pos0 + b0 + nb - 1, ntile = tmax / 64 + 1;
    const float* vc = a.vcache + (size_t)seq * a.seq_stride + kvh * HS;
    const v16f_t zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    // staging roles: thread = (row wave + 16 j, timestep lane) of the weights; 16-byte slots t + NT j of the V tile
    const float* arow[4]; float rsum[4]; int apos[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int row = wave + 16 * j, hq = row >> 4, tb = row & 15, b = b0 + min(tb, nb - 1);
        apos[j] = tb < nb ? pos0 + b0 + tb : -1;
        arow[j] = a.att + ((size_t)b * a.n_heads + kvh * KVM + hq) * a.ctx;
        rsum[j] = sums[(size_t)b * a.n_heads + kvh * KVM + hq];
    }
    typedef float v4f_native __attribute__((ext_vector_type(4)));
    v4f_native vreg[VPT]; float areg[4];
#define PVM_GLOAD(K_) do { const int t0_ = 64 * (K_); \
        static_for<0, VPT, 1>([&](auto jc) { constexpr int j = decltype(jc)::value; const int i = t + NT * j, r = i / H4, c = i % H4; \
            vreg[j] = *reinterpret_cast<const v4f_native*>(vc + (size_t)min(t0_ + r, tmax) * a.kv_dim + 4 * c); }); \
        _Pragma("unroll") for (int j = 0; j < 4; ++j) areg[j] = arow[j][max(min(t0_ + lane, apos[j]), 0)]; } while (0)
#define PVM_LSTORE(K_) do { const int t0_ = 64 * (K_); \
        static_for<0, VPT, 1>([&](auto jc) { constexpr int j = decltype(jc)::value; const int i = t + NT * j, r = i / H4, c = i % H4; \
            *reinterpret_cast<v4f_native*>(vt + r * PITCH + 4 * c) = vreg[j]; }); \
        _Pragma("unroll") for (int j = 0; j < 4; ++j) ws[(wave + 16 * j) * PVM_WP + lane] = t0_ + lane <= apos[j] ? areg[j] / rsum[j] : 0.f; } while (0)
    v8f_native ac = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};          // [0..3] column 32 cq + li, [4..7] column 32 cq + 16 + li; tokens 4 lq + r of head rg
    const bool live = cq < HS / 32;
    const float* wp = ws + (16 * rg + li) * PVM_WP + par;
    const float* vp = vt + par * PITCH + 32 * cq + 16 * cg + li;
    PVM_GLOAD(0);
    PVM_LSTORE(0);
    __syncthreads();
    for (int k = 0; k < ntile; ++k) {
        PVM_GLOAD(min(k + 1, ntile - 1));                            // unconditional (a condition around the loads makes the compiler drain them)
        if (live) {
            const int npair = (min(64, tmax + 1 - 64 * k) + 1) >> 1, ngr = (npair + 7) >> 3;
            float wa[8], va[8], wb[8], vb8[8];
#define PVM_ADD(P_, S_) do { S_ = S_ + __builtin_shufflevector(P_, P_, 0, 1, 2, 3, 4, 5, 6, 7); S_ = S_ + __builtin_shufflevector(P_, P_, 8, 9, 10, 11, 12, 13, 14, 15); } while (0)
#define PVM_LDV(G_, W_, V_) do { _Pragma("unroll") for (int u_ = 0; u_ < 8; ++u_) { const int m_ = min(8 * (G_) + u_, 31); W_[u_] = wp[2 * m_]; V_[u_] = vp[2 * m_ * PITCH]; } } while (0)
#define PVM_PV8(W_, V_) do { v16f_t Pc_ = __builtin_amdgcn_mfma_f32_16x16x1f32(W_[0], V_[0], zero16, 0, 0, 0); \
                _Pragma("unroll") for (int u_ = 0; u_ < 8; ++u_) { \
                    v16f_t Pn_ = Pc_; \
                    if (u_ < 7) Pn_ = __builtin_amdgcn_mfma_f32_16x16x1f32(W_[u_ < 7 ? u_ + 1 : 0], V_[u_ < 7 ? u_ + 1 : 0], zero16, 0, 0, 0); \
 __builtin_amdgcn_sched_barrier(0); \
                    PVM_ADD(Pc_, ac); asm volatile("" : "+v"(ac)); \
                    __builtin_amdgcn_sched_barrier(0); \
                    Pc_ = Pn_; } } while (0)
            PVM_LDV(0, wa, va); PVM_LDV(1, wb, vb8); __builtin_amdgcn_sched_barrier(0);
            int g = 0;
            for (; g + 2 <= ngr; g += 2) {
                PVM_PV8(wa, va);
                PVM_LDV(g + 2, wa, va); __builtin_amdgcn_sched_barrier(0);
                PVM_PV8(wb, vb8);
                PVM_LDV(g + 3, wb, vb8); __builtin_amdgcn_sched_barrier(0);
            }
            if (g < ngr) PVM_PV8(wa, va);
#undef PVM_ADD
#undef PVM_LDV
#undef PVM_PV8
        }
        __syncthreads();
        PVM_LSTORE(min(k + 1, ntile - 1));
        __syncthreads();
    }
#undef PVM_GLOAD
#undef PVM_LSTORE
    if (!live) return;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int tb = 4 * lq + r;
        if (tb >= nb) continue;
        float* o = a.out + (size_t)(b0 + tb) * a.out_stride + (size_t)(kvh * KVM + rg) * HS + 32 * cq + li;
        o[0] = ac[r]; o[16] = ac[4 + r];
    }
}

// r6 — the scores behind a long context with their products on the matrix pipe (phase 1 of pf_attn_fused3_kernel as a kernel of its own; kvMul 4).
// Workgroup = (kv head, 16 tokens, every S-th K tile): 64 (head, token) rows whose query rows stay in LDS for the workgroup's whole life; 8 wavefronts
// = 4 quarters of a 64-timestep K tile x 2 pairs of row groups.  Block q of an MFMA = (row group of the pair q & 1, step parity q >> 1): A = k[t][2 m +
// parity] (the lane's K row, every second element, in registers), B = q[row][2 m + parity] (64 distinct LDS addresses); the chains advance two steps
// per MFMA, s = (s + P_even) + P_odd, j ascending.  The next K tile travels in registers under the current tile's arithmetic.  Per-tile row maxima
// for pf_softmax_rows_kernel: registers -> two cross-row exchanges -> one LDS slot per (quarter, row) -> 64 threads fold the quarters.
constexpr int SCM_TB = 16, SCM_SPLIT = 4;
template <int HS>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(4, 4))) void pf_scores_mfma_kernel(const float* __restrict__ Q, int q_stride, const float* __restrict__ kc, float* __restrict__ att,
                                                             int n_heads, int kv_dim, int ctx, int pos0, int ntok, float att_mul, float* __restrict__ tmx, int tmx_tiles) {
    extern __shared__ __attribute__((aligned(16))) float kt[];       // [64][PITCH] K rows, then [64 rows][QP] query rows, then [4][64] quarter maxima
    constexpr int KVM = 4, ROWS = KVM * SCM_TB, PITCH = HS + 4, H4 = HS / 4, QP = HS + 2, NM = HS / 2, NT = 512, KPT = 64 * H4 / NT;
    static_assert(KPT >= 1 && NM % 16 == 0, "staging slots per thread; operand ring of 8 MFMAs");
    float* qs = kt + 64 * PITCH;
    float* mxs = qs + ROWS * QP;
    const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int tq = wave & 3, rp = wave >> 2;
    const int lq = lane >> 4, li = lane & 15, par = lq >> 1, rsel = lq & 1;
    const int split = blockIdx.x, nsplit = gridDim.x, kvh = blockIdx.y, b0 = blockIdx.z * SCM_TB;
    const int nb = min(SCM_TB, ntok - b0);
    const int tmax = pos0 + b0 + nb - 1, ntile = tmax / 64 + 1;
    if (split >= ntile) return;
    const float sqrt_hs = (float)sqrt((double)HS);
    const v16f_t zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    typedef float v4f_native __attribute__((ext_vector_type(4)));
    v4f_native kp[KPT];
#define SCM_KLOAD(TILE_) do { const int t0_ = 64 * (TILE_), rows_ = max(1, min(64, tmax + 1 - t0_)); \
        static_for<0, KPT, 1>([&](auto jc) { constexpr int j = decltype(jc)::value; const int i = t + NT * j, r = i / H4, c = i % H4; \
            kp[j] = *reinterpret_cast<const v4f_native*>(kc + (size_t)(min(t0_, tmax) + min(r, rows_ - 1)) * kv_dim + kvh * HS + 4 * c); }); } while (0)
    SCM_KLOAD(split);
    for (int i = t; i < ROWS * H4; i += NT) {                        // query rows (tokens past the chunk's end repeat its last token: never stored)
        const int row = i / H4, c = i % H4;
        const float4 x = *reinterpret_cast<const float4*>(Q + (size_t)(b0 + min(row & (SCM_TB - 1), nb - 1)) * q_stride + (size_t)(kvh * KVM + (row >> 4)) * HS + 4 * c);
        float* d = qs + row * QP + 4 * c;
        *reinterpret_cast<float2*>(d) = make_float2(x.x, x.y);
        *reinterpret_cast<float2*>(d + 2) = make_float2(x.z, x.w);
    }
    const float* qrow = qs + (16 * (2 * rp + rsel) + li) * QP + par;
    for (int tile = split; tile < ntile; tile += nsplit) {
        const int t0 = 64 * tile, t1 = min(tmax + 1, t0 + 64);
        static_for<0, KPT, 1>([&](auto jc) { constexpr int j = decltype(jc)::value; const int i = t + NT * j, r = i / H4, c = i % H4;
            *reinterpret_cast<v4f_native*>(kt + r * PITCH + 4 * c) = kp[j]; });
        __syncthreads();
        SCM_KLOAD(min(tile + nsplit, ntile - 1));                    // unconditional; the last trip re-reads a tile it does not use
        float mrow[2] = {-INFINITY, -INFINITY};
        if (t0 + 16 * tq <= tmax) {                                  // this wavefront's 16 timesteps hold at least one attended position
            const float* krow = kt + min(16 * tq + li, t1 - t0 - 1) * PITCH;
            float kreg[NM];
#pragma unroll
            for (int c = 0; c < H4; ++c) {
                const v4f_native_s x = *reinterpret_cast<const v4f_native_s*>(krow + 4 * c);
                kreg[2 * c] = par ? x.y : x.x; kreg[2 * c + 1] = par ? x.w : x.z;
            }
            v8f_native sc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};      // [0..3] row group 2 rp, [4..7] row group 2 rp + 1 (row li of each); timesteps 16 tq + 4 lq + r
            float qa[8], qb[8];
#define SCM_LDQ(G_, R_) do { _Pragma("unroll") for (int u_ = 0; u_ < 8; ++u_) R_[u_] = qrow[2 * (8 * (G_) + u_)]; } while (0)
#define SCM_ADD(P_, S_) do { S_ = S_ + __builtin_shufflevector(P_, P_, 0, 1, 2, 3, 4, 5, 6, 7); S_ = S_ + __builtin_shufflevector(P_, P_, 8, 9, 10, 11, 12, 13, 14, 15); } while (0)
#define SCM_MF8(G_, R_, RN_) do { _Pragma("unroll") for (int u_ = 0; u_ < 8; ++u_) { \
                const int mn_ = 8 * (G_) + u_ + 1;      /* a constant after unrolling */ \
                v16f_t Pn_ = Pc; \
                if (mn_ < NM) Pn_ = __builtin_amdgcn_mfma_f32_16x16x1f32(kreg[mn_ < NM ? mn_ : 0], u_ < 7 ? R_[u_ < 7 ? u_ + 1 : 0] : RN_[0], zero16, 0, 0, 0); \
                SCM_ADD(Pc, sc); asm volatile("" : "+v"(sc));      /* every element's adds stay with their MFMA (left alone the chains are scalarised, re-vectorised pair by pair and the products spilled) */ \
                __builtin_amdgcn_sched_barrier(0); \
                Pc = Pn_; } } while (0)
            SCM_LDQ(0, qa); SCM_LDQ(1, qb); __builtin_amdgcn_sched_barrier(0);
            v16f_t Pc = __builtin_amdgcn_mfma_f32_16x16x1f32(kreg[0], qa[0], zero16, 0, 0, 0);
            static_for<0, NM / 8, 2>([&](auto gc) {
                constexpr int g = decltype(gc)::value;
                SCM_MF8(g, qa, qb);
                SCM_LDQ((g + 2 < NM / 8 ? g + 2 : NM / 8 - 1), qa); __builtin_amdgcn_sched_barrier(0);
                SCM_MF8(g + 1, qb, qa);
                SCM_LDQ((g + 3 < NM / 8 ? g + 3 : NM / 8 - 1), qb); __builtin_amdgcn_sched_barrier(0);
            });
#undef SCM_LDQ
#undef SCM_ADD
#undef SCM_MF8
#pragma unroll
            for (int rgp = 0; rgp < 2; ++rgp) {
                const int tb = li, head = kvh * KVM + 2 * rp + rgp, b = b0 + tb;
                const int ts0 = t0 + 16 * tq + 4 * lq, lim = tb < nb ? pos0 + b : -1;       // attended: ts <= lim
                float v[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    v[r] = att_mul != 0.f ? sc[4 * rgp + r] * att_mul : sc[4 * rgp + r] / sqrt_hs;
                    if (ts0 + r <= lim) mrow[rgp] = fmaxf(mrow[rgp], v[r]);
                }
                float* o = att + ((size_t)b * n_heads + head) * ctx + ts0;
                if (ts0 + 3 <= lim) *reinterpret_cast<float4*>(o) = make_float4(v[0], v[1], v[2], v[3]);
                else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) if (ts0 + r <= lim) o[r] = v[r];
                }
            }
        }
        if (tmx) {                                                   // the quarter's maximum per row: fold the four 16-lane rows, one slot per (quarter, row)
#pragma unroll
            for (int rgp = 0; rgp < 2; ++rgp) {
                float m = mrow[rgp];
                m = fmaxf(m, __shfl_xor(m, 16, 64)); m = fmaxf(m, __shfl_xor(m, 32, 64));
                if (lq == 0) mxs[tq * ROWS + 16 * (2 * rp + rgp) + li] = m;
            }
        }
        __syncthreads();
        if (tmx && t < ROWS) {
            const int tb = t & (SCM_TB - 1), b = b0 + tb;
            if (tb < nb && t0 <= pos0 + b) {
                const float m = fmaxf(fmaxf(mxs[t], mxs[ROWS + t]), fmaxf(mxs[2 * ROWS + t], mxs[3 * ROWS + t]));
                tmx[((size_t)b * n_heads + kvh * KVM + (t >> 4)) * tmx_tiles + tile] = m;
            }
        }
    }
#undef SCM_KLOAD
}

// LDS attributes of the prefill attention kernels (both plan kinds)
static int32_t pf_attention_attributes(gl3_ctx* ctx) {
#define GL3_ATTR150(K_) GL3_HIP(hipFuncSetAttribute((const void*)K_, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024))
    GL3_ATTR150(pf_attn_fused_kernel<128>); GL3_ATTR150(pf_attn_fused_kernel<64>); GL3_ATTR150(pf_attn_fused_kernel<32>);
    GL3_ATTR150(pf_attn_fused2_kernel<128>); GL3_ATTR150(pf_attn_fused2_kernel<64>); GL3_ATTR150(pf_attn_fused2_kernel<32>);
    GL3_ATTR150(pf_attn_fused3_kernel<128>); GL3_ATTR150(pf_attn_fused3_kernel<64>);
    GL3_ATTR150(pf_scores_mfma_kernel<128>); GL3_ATTR150(pf_scores_mfma_kernel<64>); GL3_ATTR150(pf_pv_mfma_kernel<128>); GL3_ATTR150(pf_pv_mfma_kernel<64>);
    GL3_ATTR150((pf_scores_pk_kernel<128, 4>)); GL3_ATTR150((pf_scores_pk_kernel<128, 2>)); GL3_ATTR150((pf_scores_pk_kernel<128, 1>));
    GL3_ATTR150((pf_scores_pk_kernel<64, 4>)); GL3_ATTR150((pf_scores_pk_kernel<64, 2>)); GL3_ATTR150((pf_scores_pk_kernel<64, 1>));
    GL3_ATTR150((pf_scores_pk_kernel<32, 4>)); GL3_ATTR150((pf_scores_pk_kernel<32, 2>)); GL3_ATTR150((pf_scores_pk_kernel<32, 1>));
#undef GL3_ATTR150
    return GL3_OK;
}

int32_t gl3_prefill_alloc(gl3_ctx* ctx) {
    const gl3_model_desc& d = ctx->d;
    gl3_prefill_state* p = new gl3_prefill_state();
    ctx->pf = p;
    p->max_batch = d.max_batch;
    const size_t M = d.max_batch;
    p->vl = ctx->emb.vl;
    if (p->vl) {       // f32-activation weight types: f32 buffers; the gathered ones (X, AO, HB, LOGITS) in the arena under tensor parallelism
        GL3_HIP(hipMalloc((void**)&p->tokens, M * sizeof(int32_t)));
        if (ctx->arena.base && ctx->arena.off[GB_PF_X]) {
            uint8_t* b = ctx->arena.base;
            p->X = (float*)(b + ctx->arena.off[GB_PF_X]); p->AO = (float*)(b + ctx->arena.off[GB_PF_AO]); p->HB = (float*)(b + ctx->arena.off[GB_PF_HB]);
            p->LOGITS = (float*)(b + ctx->arena.off[GB_PF_LOGITS]); p->logits_rows = ctx->arena.pf_logits_rows;
            p->in_arena = true;
        } else {
            GL3_HIP(hipMalloc((void**)&p->X, M * d.dim * 4));
            GL3_HIP(hipMalloc((void**)&p->AO, M * ctx->q_dim * 4));
            GL3_HIP(hipMalloc((void**)&p->HB, M * d.hidden * 4));
        }
        // XN: normalised / un-chunked f32 operand of the next GEMM (K up to max(dim, q_dim, hidden)); HB2: this rank's up projection
        const size_t kmax = (size_t)(d.hidden > ctx->q_dim ? (d.hidden > d.dim ? d.hidden : d.dim) : (ctx->q_dim > d.dim ? ctx->q_dim : d.dim));
        GL3_HIP(hipMalloc((void**)&p->XN, M * kmax * 4));
        GL3_HIP(hipMalloc((void**)&p->HB2, M * ctx->hidden_l * 4));
        GL3_HIP(hipMalloc((void**)&p->QKV, M * (ctx->q_dim + 2 * ctx->kv_dim) * 4));
        GL3_HIP(hipMalloc((void**)&p->ATT, M * d.n_heads * (size_t)d.ctx * 4));
        p->tmx_tiles = (d.ctx + 63) / 64;
        GL3_HIP(hipMalloc((void**)&p->TMX, M * d.n_heads * (size_t)p->tmx_tiles * 4));
        GL3_HIP(hipMalloc((void**)&p->SUMS, M * d.n_heads * 4));
        GL3_HIP(hipMalloc((void**)&p->seqpos, 2 * M * sizeof(int32_t)));
        GL3_HIP(hipMalloc((void**)&p->amax, M * sizeof(int32_t)));
        GL3_HIP(hipMalloc((void**)&p->amx_v, M * AMX_SPLIT * sizeof(float)));
        GL3_HIP(hipMalloc((void**)&p->amx_i, M * AMX_SPLIT * sizeof(int)));
        if (getenv("GL3_DEBUG_ALLOC"))
            fprintf(stderr, "[gl3 alloc vl] M %zu tokens %p X %p XN %p AO %p HB %p HB2 %p QKV %p ATT %p seqpos %p amax %p (dim %d hidden %d qdim %d ctx %d)\n", M, (void*)p->tokens,
                    (void*)p->X, (void*)p->XN, (void*)p->AO, (void*)p->HB, (void*)p->HB2, (void*)p->QKV, (void*)p->ATT, (void*)p->seqpos, (void*)p->amax, d.dim, d.hidden, ctx->q_dim, d.ctx);
        GL3_HIP(hipFuncSetAttribute((const void*)gemm_f16_mfma_kernel<EPI_STORE>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * F16G_STAGE));
        GL3_HIP(hipFuncSetAttribute((const void*)gemm_f16_mfma_kernel<EPI_RESID>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * F16G_STAGE));
        GL3_HIP(hipFuncSetAttribute((const void*)gemm_vlq_kernel<WT_Q4_0, EPI_STORE>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * vlq_stage_floats<WT_Q4_0>() * 4));
        GL3_HIP(hipFuncSetAttribute((const void*)gemm_vlq_kernel<WT_Q4_0, EPI_RESID>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * vlq_stage_floats<WT_Q4_0>() * 4));
        GL3_HIP(hipFuncSetAttribute((const void*)gemm_vlq_kernel<WT_Q8_0, EPI_STORE>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * vlq_stage_floats<WT_Q8_0>() * 4));
        GL3_HIP(hipFuncSetAttribute((const void*)gemm_vlq_kernel<WT_Q8_0, EPI_RESID>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * vlq_stage_floats<WT_Q8_0>() * 4));
#define GL3_VQM_ATTR(WT_, EPI_, OCC_) GL3_HIP(hipFuncSetAttribute((const void*)gemm_vlq_mfma_kernel<WT_, EPI_, OCC_>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * VQM_STAGE_FLOATS * 4))
        GL3_VQM_ATTR(WT_Q4_0, EPI_STORE, 2); GL3_VQM_ATTR(WT_Q4_0, EPI_RESID, 2); GL3_VQM_ATTR(WT_Q4_0, EPI_STORE, 4); GL3_VQM_ATTR(WT_Q4_0, EPI_RESID, 4);
        GL3_VQM_ATTR(WT_Q8_0, EPI_STORE, 2); GL3_VQM_ATTR(WT_Q8_0, EPI_RESID, 2); GL3_VQM_ATTR(WT_Q8_0, EPI_STORE, 4); GL3_VQM_ATTR(WT_Q8_0, EPI_RESID, 4);
#undef GL3_VQM_ATTR
        GL3_HIP(hipFuncSetAttribute((const void*)pf_attn_scores_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
        GL3_HIP(hipFuncSetAttribute((const void*)pf_attn_softmax_pv_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
    { const int32_t ar = pf_attention_attributes(ctx); if (ar != GL3_OK) return ar; }
        return GL3_OK;
    }
    p->maxk = d.hidden > ctx->q_dim ? d.hidden : ctx->q_dim;
    if (d.dim > p->maxk) p->maxk = d.dim;
    p->maxk = (p->maxk + 127) & ~127;
    GL3_HIP(hipMalloc((void**)&p->tokens, M * sizeof(int32_t)));
    if (ctx->arena.base && ctx->arena.off[GB_PF_X]) {      // tensor parallel: the gathered activations live in the arena the peers map
        uint8_t* b = ctx->arena.base;
        p->X = (float*)(b + ctx->arena.off[GB_PF_X]); p->AO = (float*)(b + ctx->arena.off[GB_PF_AO]); p->HB = (float*)(b + ctx->arena.off[GB_PF_HB]);
        p->LOGITS = (float*)(b + ctx->arena.off[GB_PF_LOGITS]); p->logits_rows = ctx->arena.pf_logits_rows;
        p->in_arena = true;
    } else {
        GL3_HIP(hipMalloc((void**)&p->X, M * d.dim * 4));
        GL3_HIP(hipMalloc((void**)&p->AO, M * ctx->q_dim * 4));
        GL3_HIP(hipMalloc((void**)&p->HB, M * d.hidden * 4));
    }
    const size_t MQ = M < BD_TS_MAX ? BD_TS_MAX : M;      // the small-batch operand layout (bd_tslots) always spans its 32 / 64 token slots
    const size_t MQP = (MQ + 127) & ~(size_t)127;      // token slots of the chunk-major layouts (whole 128-token GEMM tiles)
    GL3_HIP(hipMalloc((void**)&p->XQ, MQP * p->maxk + GL3_TAIL_PAD));
    GL3_HIP(hipMalloc((void**)&p->XS, MQ * (p->maxk / 32) * 4 + GL3_TAIL_PAD));
    p->xp_tok = (int)MQP;
    GL3_HIP(hipMalloc((void**)&p->XP, (size_t)(p->maxk / 32 + 4) * p->xp_tok * 32 + GL3_TAIL_PAD));
    GL3_HIP(hipMemsetAsync(p->XP, 0, (size_t)(p->maxk / 32 + 4) * p->xp_tok * 32 + GL3_TAIL_PAD, ctx->stream));
    if (M > 64 && d.tp_size == 1) {
        GL3_HIP(hipMalloc((void**)&p->XQh, MQP * p->maxk + GL3_TAIL_PAD));
        GL3_HIP(hipMalloc((void**)&p->XPh, (size_t)(p->maxk / 32 + 4) * p->xp_tok * 32 + GL3_TAIL_PAD));
        GL3_HIP(hipMemsetAsync(p->XQh, 0, MQP * p->maxk + GL3_TAIL_PAD, ctx->stream));
        GL3_HIP(hipMemsetAsync(p->XPh, 0, (size_t)(p->maxk / 32 + 4) * p->xp_tok * 32 + GL3_TAIL_PAD, ctx->stream));
    }
    GL3_HIP(hipMalloc((void**)&p->XQb, (size_t)BD_TS_MAX * p->maxk + GL3_TAIL_PAD));
    GL3_HIP(hipMalloc((void**)&p->XSb, (size_t)BD_TS_MAX * (p->maxk / 32) * 4 + GL3_TAIL_PAD));
    GL3_HIP(hipMemsetAsync(p->XQb, 0, (size_t)BD_TS_MAX * p->maxk, ctx->stream));
    GL3_HIP(hipMemsetAsync(p->XSb, 0, (size_t)BD_TS_MAX * (p->maxk / 32) * 4, ctx->stream));
    GL3_HIP(hipMemsetAsync(p->XQ, 0, MQP * p->maxk, ctx->stream));
    GL3_HIP(hipMemsetAsync(p->XS, 0, MQ * (p->maxk / 32) * 4, ctx->stream));
    GL3_HIP(hipMalloc((void**)&p->QKV, M * (ctx->q_dim + 2 * ctx->kv_dim) * 4));
    GL3_HIP(hipMalloc((void**)&p->ATT, M * d.n_heads * (size_t)d.ctx * 4));
    p->tmx_tiles = (d.ctx + 63) / 64;
    GL3_HIP(hipMalloc((void**)&p->TMX, M * d.n_heads * (size_t)p->tmx_tiles * 4));
    GL3_HIP(hipMalloc((void**)&p->SUMS, M * d.n_heads * 4));
    GL3_HIP(hipMalloc((void**)&p->seqpos, 2 * M * sizeof(int32_t)));
    GL3_HIP(hipMalloc((void**)&p->amax, M * sizeof(int32_t)));
    GL3_HIP(hipMalloc((void**)&p->amx_v, M * AMX_SPLIT * sizeof(float)));
    GL3_HIP(hipMalloc((void**)&p->amx_i, M * AMX_SPLIT * sizeof(int)));
#define GL3_GEMM_LDS(...) GL3_HIP(hipFuncSetAttribute((const void*)pf_gemm_kernel<__VA_ARGS__>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * gm_stage_bytes(128)))
    GL3_GEMM_LDS(EPI_STORE, 1, 4); GL3_GEMM_LDS(EPI_STORE, 2, 4); GL3_GEMM_LDS(EPI_STORE, 1, 8);
    GL3_GEMM_LDS(EPI_RESID, 1, 4); GL3_GEMM_LDS(EPI_RESID, 2, 4); GL3_GEMM_LDS(EPI_RESID, 1, 8);
    GL3_GEMM_LDS(EPI_SWIGLU, 1, 4);
    GL3_GEMM_LDS(EPI_STORE, 1, 4, 32); GL3_GEMM_LDS(EPI_RESID, 1, 4, 32); GL3_GEMM_LDS(EPI_SWIGLU, 1, 4, 32);
#undef GL3_GEMM_LDS
    GL3_HIP(gl3_gemm2_allow_lds());                      // pf_gemm2_kernel instantiations (own translation unit)
    GL3_HIP(gl3_gemm3_allow_lds());
#define GL3_BDK_ATTR(EPI_, P_, DA_, Q_, TS_) GL3_HIP(hipFuncSetAttribute((const void*)bdk_gemm_kernel<EPI_, P_, DA_, Q_, TS_>, hipFuncAttributeMaxDynamicSharedMemorySize, bdk_lds_bytes<EPI_, P_, Q_>()))
#define GL3_BDK_ATTRS(P_, TS_) GL3_BDK_ATTR(EPI_STORE, P_, 4, false, TS_); GL3_BDK_ATTR(EPI_RESID, P_, 4, false, TS_); GL3_BDK_ATTR(EPI_SWIGLU, P_, 4, false, TS_); GL3_BDK_ATTR(EPI_SWIGLU, (P_ > 2 ? 2 : P_), 4, true, TS_)
    GL3_BDK_ATTRS(2, BD_TS); GL3_BDK_ATTRS(3, BD_TS); GL3_BDK_ATTRS(4, BD_TS); GL3_BDK_ATTRS(2, BD_TS_MAX); GL3_BDK_ATTRS(3, BD_TS_MAX); GL3_BDK_ATTRS(4, BD_TS_MAX);
#undef GL3_BDK_ATTRS
    // ring of 8 tiles per producer (GL3_BDK_DA=8): single-matrix classes only
#define GL3_BDK_ATTRS8(P_, TS_) GL3_BDK_ATTR(EPI_STORE, P_, 8, false, TS_); GL3_BDK_ATTR(EPI_RESID, P_, 8, false, TS_)
    GL3_BDK_ATTRS8(2, BD_TS); GL3_BDK_ATTRS8(3, BD_TS); GL3_BDK_ATTRS8(2, BD_TS_MAX); GL3_BDK_ATTRS8(3, BD_TS_MAX);
#undef GL3_BDK_ATTRS8
#undef GL3_BDK_ATTR
    GL3_HIP(hipFuncSetAttribute((const void*)pf_attn_scores_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
    GL3_HIP(hipFuncSetAttribute((const void*)pf_attn_softmax_pv_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
    { const int32_t ar = pf_attention_attributes(ctx); if (ar != GL3_OK) return ar; }
    return GL3_OK;
}

void gl3_prefill_free(gl3_ctx* ctx) {
    gl3_prefill_state* p = ctx->pf;
    if (!p) return;
    for (auto ge : p->step_graphs) if (ge) hipGraphExecDestroy(ge);
    auto f = [](void* q) { if (q) hipFree(q); };
    f(p->tokens); f(p->XQ); f(p->XS); f(p->XP); f(p->XQh); f(p->XPh); f(p->XQb); f(p->XSb); f(p->QKV); f(p->ATT); f(p->TMX); f(p->SUMS); f(p->seqpos); f(p->amax); f(p->amx_v); f(p->amx_i); f(p->XN); f(p->HB2);
    if (!p->in_arena) { f(p->X); f(p->AO); f(p->HB); f(p->LOGITS); }
    delete p;
    ctx->pf = nullptr;
}

// > 64-token GEMMs on pf_gemm3_kernel (default) or on the r3 / r4 kernels (GL3_PF_GEMM3=0): the quantiser and the GEMM of a step must agree on the
// activation layout, so both ask here
static bool pf_use_gemm3() {
    static const bool on = !(getenv("GL3_PF_GEMM3") && atoi(getenv("GL3_PF_GEMM3")) == 0);
    return on;
}

// Token slots of the XQ2 / XS2 operand layout when a step of n tokens runs on the wave-owned small-batch GEMM
// (bdw_gemm_kernel), 0 = row layout + the tiled GEMMs.  The quantiser and the GEMM of a step must agree, so both ask here.
static int bd_tslots(int n) {
    static const bool off = getenv("GL3_BD_GEMM") && atoi(getenv("GL3_BD_GEMM")) == 0;      // A/B switch: tiled GEMMs for small batches too
    return off ? 0 : n <= BD_TS ? BD_TS : n <= BD_TS_MAX ? BD_TS_MAX : 0;
}

template <int EPI>
static void launch_gemm(gl3_ctx* ctx, const Q8Mat& w, const Q8Mat* w2, int ntok, float* out, int out_stride, float out_scale = 1.0f,
                        bool second_operand = false, bool quantised_out = false) {
    gl3_prefill_state* p = ctx->pf;
    GemmArgs a{};
    a.w = w.w; a.w2 = w2 ? w2->w : nullptr; a.rows = w.rows; a.ng = w.ng; a.nb = w.k / 32;
    a.XQ = second_operand ? p->XQb : p->XQ; a.XS = second_operand ? p->XSb : p->XS;
    a.maxk = p->maxk; a.ntok = ntok; a.out = out; a.out_stride = out_stride; a.out_scale = out_scale;
    a.XQo = p->XQb; a.XSo = p->XSb;
    a.XP = p->XP; a.xp_tok = p->xp_tok;
    // 128-row tiles only when they still give >= 2 workgroups per CU; otherwise 64-row tiles, and 8 wavefronts per
    // workgroup when even those leave a single workgroup per CU
    if (const int ts = bd_tslots(ntok)) {      // static-batched decode / small chunks: one wavefront per (16-row strip, 16 tokens), all of K
        a.tslots = ts;
        const dim3 grid(bdw_grid((w.rows + 15) / 16, (ntok + 15) / 16));
        const dim3 gridq(bdw_grid((w.rows + 31) / 32, (ntok + 15) / 16));      // quantised output: two strips per workgroup
        // r6: K split over P producer wavefronts + one chain wavefront per (strip, token tile) (gl3_bdk_gemm.h); GL3_BDK=0: one wavefront per (strip, token tile)
        // Measured slower than one wavefront per (strip, token tile) on every class (Qwen3-4B B = 32: 3.53 - 4.0 ms per step against 3.14;
        // profiles/r06_bd32_kslice.md has the per-round stamps), so it is OFF by default and kept as the bit-exact record of that experiment.
        static const int bdk = getenv("GL3_BDK") ? atoi(getenv("GL3_BDK")) : 0;
        static const int bdk_p = getenv("GL3_BDK_P") ? atoi(getenv("GL3_BDK_P")) : 3;
#define GL3_BDK_L(P_, TS_) \
        do { \
            if constexpr (EPI == EPI_SWIGLU) { \
                if (quantised_out) hipLaunchKernelGGL((bdk_gemm_kernel<EPI, (P_ > 2 ? 2 : P_), 4, true, TS_>), gridq, dim3(128 * ((P_ > 2 ? 2 : P_) + 1)), (bdk_lds_bytes<EPI, (P_ > 2 ? 2 : P_), true>()), ctx->stream, a); \
                else hipLaunchKernelGGL((bdk_gemm_kernel<EPI, P_, 4, false, TS_>), grid, dim3(64 * (P_ + 1)), (bdk_lds_bytes<EPI, P_, false>()), ctx->stream, a); \
            } else hipLaunchKernelGGL((bdk_gemm_kernel<EPI, P_, 4, false, TS_>), grid, dim3(64 * (P_ + 1)), (bdk_lds_bytes<EPI, P_, false>()), ctx->stream, a); \
        } while (0)
#define GL3_BDK(TS_) do { if (bdk_p == 2) GL3_BDK_L(2, TS_); else if (bdk_p == 4) GL3_BDK_L(4, TS_); else GL3_BDK_L(3, TS_); } while (0)
        static const int bdk_gu = getenv("GL3_BDK_GU") ? atoi(getenv("GL3_BDK_GU")) : 1;      // 0: the gate + up launch stays on bdw_gemm_kernel
        static const int bdk_da = getenv("GL3_BDK_DA") ? atoi(getenv("GL3_BDK_DA")) : 4;      // tiles in flight per producer (8: single-matrix classes, P = 2 / 3)
        if constexpr (EPI != EPI_SWIGLU) {
            if (bdk && bdk_da == 8 && bdk_p <= 3) {
#define GL3_BDK8(P_, TS_) hipLaunchKernelGGL((bdk_gemm_kernel<EPI, P_, 8, false, TS_>), grid, dim3(64 * (P_ + 1)), (bdk_lds_bytes<EPI, P_, false>()), ctx->stream, a)
                if (ts == BD_TS) { if (bdk_p == 2) GL3_BDK8(2, BD_TS); else GL3_BDK8(3, BD_TS); }
                else { if (bdk_p == 2) GL3_BDK8(2, BD_TS_MAX); else GL3_BDK8(3, BD_TS_MAX); }
#undef GL3_BDK8
                return;
            }
        }
        if (bdk && (EPI != EPI_SWIGLU || bdk_gu)) { if (ts == BD_TS) GL3_BDK(BD_TS); else GL3_BDK(BD_TS_MAX); return; }
#undef GL3_BDK
#undef GL3_BDK_L
#define GL3_BDW(TS_) \
        do { \
            if constexpr (EPI == EPI_SWIGLU) { \
                if (quantised_out) hipLaunchKernelGGL((bdw_gemm_kernel<EPI, 4, 2, true, TS_>), gridq, dim3(128), 0, ctx->stream, a); \
                else hipLaunchKernelGGL((bdw_gemm_kernel<EPI, 4, 2, false, TS_>), grid, dim3(64), 0, ctx->stream, a); \
            } else hipLaunchKernelGGL((bdw_gemm_kernel<EPI, 8, 2, false, TS_>), grid, dim3(64), 0, ctx->stream, a); \
        } while (0)
        if (ts == BD_TS) GL3_BDW(BD_TS); else GL3_BDW(BD_TS_MAX);
#undef GL3_BDW
        return;
    }
    if (ntok <= 64) {      // 32-token tiles, 128 rows (x2 matrices for SwiGLU) per workgroup
        const int ntt = (ntok + 31) / 32, nrt = (w.rows + 127) / 128;
        a.ntt = ntt; a.nrt = nrt;
        constexpr int AR = EPI == EPI_SWIGLU ? 256 : 128;
        hipLaunchKernelGGL((pf_gemm_kernel<EPI, 1, 4, 32>), dim3(8 * ((ntt * nrt + 7) / 8)), dim3(256), 2 * gm_stage_bytes(AR, 32), ctx->stream, a);
        return;
    }
    const int ntt = (ntok + GM_TOK - 1) / GM_TOK;
    a.ntt = ntt;
    auto grid = [&](int nrt) { a.nrt = nrt; return dim3(8 * ((ntt * nrt + 7) / 8)); };
    // r4: scale products on the matrix pipe (gl3_prefill_gemm2.h).  GL3_PF_GEMM2=0: the r3 kernel; 1: -B s on the VALU (A/B switches)
    // Default: the gate/up GEMM (two matrices per workgroup, the dominant launch) on the r4 kernel — measured 246-250 us against 266-272
    // for the r3 kernel at 512 tokens of the 8B layer; the other shapes stay on the r3 kernel (qkv 79 vs 80, wo 57 vs 45, down 222 vs 139:
    // profiles/r04_gemm_experiments.md).  GL3_PF_GEMM2=0: r3 kernel everywhere; GL3_PF_GEMM2_ALL=1: r4 kernel for every shape;
    // GL3_PF_GEMM2=1: A/B form (-B s on the VALU).
    // r6: every class on pf_gemm3_kernel (mid-stage barrier, partial vmcnt, scale-operand side table, chunk-major activations);
    // GL3_PF_GEMM3=0 restores the r5 choice below (the quantiser then writes the row layout those kernels read)
    if (pf_use_gemm3()) {
        if (quantised_out) { a.XQo = p->XQh; a.XPo = p->XPh; }          // gate + up: hb as the down projection's operand (tall tiling, one rank)
        if (second_operand) { a.XQ = p->XQh; a.XP = p->XPh; }            // down: reads it
        gl3_gemm3_launch(EPI, a, w.rows, ntok, ctx->stream);
        return;
    }
    static const int g2 = getenv("GL3_PF_GEMM2") ? atoi(getenv("GL3_PF_GEMM2")) : 2;
    static const bool g2_all = getenv("GL3_PF_GEMM2_ALL") && atoi(getenv("GL3_PF_GEMM2_ALL"));
    if (g2 && (EPI == EPI_SWIGLU || g2_all)) { gl3_gemm2_launch(EPI, a, w.rows, ntok, g2, ctx->stream); return; }
    if constexpr (EPI == EPI_SWIGLU) {
        const dim3 g = grid((w.rows + 63) / 64);
        hipLaunchKernelGGL((pf_gemm_kernel<EPI, 1, 4>), g, dim3(256), 2 * gm_stage_bytes(128), ctx->stream, a);
    } else if ((size_t)ntt * ((w.rows + 127) / 128) >= 512) {
        const dim3 g = grid((w.rows + 127) / 128);
        hipLaunchKernelGGL((pf_gemm_kernel<EPI, 2, 4>), g, dim3(256), 2 * gm_stage_bytes(128), ctx->stream, a);
    } else if ((size_t)ntt * ((w.rows + 63) / 64) > 256) {
        const dim3 g = grid((w.rows + 63) / 64);
        hipLaunchKernelGGL((pf_gemm_kernel<EPI, 1, 4>), g, dim3(256), 2 * gm_stage_bytes(64), ctx->stream, a);
    } else {
        const dim3 g = grid((w.rows + 63) / 64);
        hipLaunchKernelGGL((pf_gemm_kernel<EPI, 1, 8>), g, dim3(512), 2 * gm_stage_bytes(64), ctx->stream, a);
    }
}

// RoPE + KV write + attention of layer l for the n tokens whose raw q | k | v rows are in p->QKV -> AOr (this rank's chunk of the
// attention output).  fuse_q: static-batched decode on one rank writes the output as the wo projection's int8 operand instead.
// returns true when the attention output was written as the wo projection's int8 operand (no quantise launch needed)
static bool pf_attention(gl3_ctx* ctx, int l, int n, int max_pos, int one_seq, float* AOr, bool fuse_q) {
    gl3_prefill_state* p = ctx->pf;
    const gl3_model_desc& d = ctx->d;
    hipStream_t s = ctx->stream;
    gl3_layer& L = ctx->layers[l];
    const int32_t* seq = p->seqpos;
    const int32_t* pos = p->seqpos + p->max_batch;
    const int kvmul = d.n_heads / d.n_kv_heads;
    const int H = ctx->heads_l, KVH = ctx->kv_heads_l, qd = ctx->q_dim_l, kvd = ctx->kv_dim_l;
    const int qkv_dim = qd + 2 * kvd;
    const size_t kv_layer = (size_t)d.ctx * kvd;
    // one workgroup per (kv head, token) serves the kv head's whole group of query heads when its LDS image fits
    const int bd_group = (kvmul <= 8 && attn_head_smem(d.head_size, kvmul) <= 150 * 1024) ? kvmul : 1;
    const bool fused_decode = ctx->fused_attn_ok && max_pos < AF_MAXN && !(getenv("GL3_NO_FUSED_BD_ATTN") && atoi(getenv("GL3_NO_FUSED_BD_ATTN")));
    RopeArgs ra{};
    ra.QKV = p->QKV; ra.qkv_stride = qkv_dim; ra.kcache = ctx->kcache + l * kv_layer; ra.vcache = ctx->vcache + l * kv_layer;
    ra.cr = ctx->rope_cr; ra.ci = ctx->rope_ci; ra.qnorm = L.qnorm; ra.knorm = L.knorm; ra.bq = L.bq; ra.bk = L.bk; ra.bv = L.bv; ra.n_heads = H;
    ra.n_kv_heads = KVH; ra.hs = d.head_size; ra.q_dim = qd; ra.kv_dim = kvd;
    ra.arch = ctx->rope_arch; ra.eps = d.rms_eps; ra.seq = seq; ra.pos = pos; ra.seq_stride = ctx->kv_seq_stride;
    PfAttnArgs aa{};
    aa.Q = p->QKV; aa.q_stride = qkv_dim; aa.kcache = ra.kcache; aa.vcache = ra.vcache; aa.att = p->ATT; aa.out = AOr;
    aa.out_stride = qd; aa.n_heads = H; aa.n_kv_heads = KVH; aa.hs = d.head_size; aa.kv_dim = kvd;
    aa.ctx = d.ctx; aa.seq = seq; aa.pos = pos; aa.seq_stride = ctx->kv_seq_stride; aa.att_mul = ctx->att_mul;
    const int nsplit = (max_pos + 1 + ATT_TT - 1) / ATT_TT;
    const int hs = d.head_size;
    if (one_seq < 0 && fused_decode) {
        // static-batched decode at positions < AF_MAXN: RoPE + KV write + scores + softmax + weighted V sum of every
        // (token, head) in ONE launch (attn_head_kernel, grid = heads x tokens) instead of three per-token-grid kernels
        AttnArgs ha{};
        ha.qkv = p->QKV; ha.qkv_stride = qkv_dim; ha.kcache = ra.kcache; ha.vcache = ra.vcache; ha.rope_cr = ctx->rope_cr; ha.rope_ci = ctx->rope_ci;
        ha.qnorm = L.qnorm; ha.knorm = L.knorm; ha.bq = L.bq; ha.bk = L.bk; ha.bv = L.bv; ha.dyn = ctx->dyn; ha.att = nullptr;
        ha.xb = AOr; ha.xb_stride = qd; ha.n_heads = H; ha.n_kv_heads = KVH; ha.hs = hs; ha.q_dim = qd; ha.kv_dim = kvd; ha.ctx = d.ctx;
        ha.eps = d.rms_eps; ha.arch = ctx->rope_arch; ha.att_mul = ctx->att_mul; ha.seqv = seq; ha.posv = pos; ha.seq_stride = ctx->kv_seq_stride;
        ha.group = bd_group;
        if (fuse_q) { ha.xq_out = p->XQ; ha.xs_out = p->XS; ha.xq_slots = bd_tslots(n); }
        attn_head_dispatch(hs, [&](auto kern) { hipLaunchKernelGGL(kern, dim3(H / bd_group, n), dim3(256), attn_head_smem(hs, bd_group), s, ha); });
        return fuse_q;
    }
    hipLaunchKernelGGL(pf_rope_kv_kernel, dim3(H + KVH, n), dim3(64), 0, s, ra);
    // r6: pf_softmax_rows_kernel streams the score rows (no row-fits-LDS limit); GL3_PF_SOFTMAX_ROWS=0: the one-row-per-wavefront kernel
    static const bool rows_off = getenv("GL3_PF_SOFTMAX_ROWS") && atoi(getenv("GL3_PF_SOFTMAX_ROWS")) == 0;
    const bool rows_softmax = !rows_off && d.ctx % 4 == 0 && d.ctx >= 64 && p->TMX && p->SUMS;
    const bool tiled = one_seq >= 0 && kvmul <= 4 && (hs == 32 || hs == 64 || hs == 128) && (rows_softmax || (size_t)(max_pos + 1) * 4 <= 60 * 1024);
    // r4: one launch for scores + softmax + weighted V sum when a tile's score rows fit LDS (GL3_PF_FUSED_ATTN=0: the three kernels)
    static const bool fused_off = getenv("GL3_PF_FUSED_ATTN") && atoi(getenv("GL3_PF_FUSED_ATTN")) == 0;
    const int fa_sstride = ((max_pos + 1 + 63) & ~63) + 4;
    if (tiled && !fused_off && 64 * (hs / 4) <= 8 * 64 * kvmul && fa_smem_bytes(hs, kvmul, fa_sstride) <= 150 * 1024) {      // 8 float4 per thread stage a tile
        const int pos0 = max_pos + 1 - n, ntile = (n + FA_TB - 1) / FA_TB;
        const size_t sms = fa_smem_bytes(hs, kvmul, fa_sstride);
        const float* kc1 = aa.kcache + (size_t)one_seq * ctx->kv_seq_stride;
        const float* vc1 = aa.vcache + (size_t)one_seq * ctx->kv_seq_stride;
        // r6: > 64 tokens on one rank with head size 128: the output is written quantised for the wo GEMM (pf_gemm3_kernel's operand layout)
        static const bool qao_off = getenv("GL3_PF_ATTN_QOUT") && atoi(getenv("GL3_PF_ATTN_QOUT")) == 0;
        const bool qao = !qao_off && hs == 128 && n > 64 && d.tp_size == 1 && pf_use_gemm3() && p->XP && (getenv("GL3_NO_FUSED_QUANT") == nullptr || atoi(getenv("GL3_NO_FUSED_QUANT")) == 0);
        uint8_t* xqo = qao ? p->XQ : nullptr;
        uint4* xpo = qao ? reinterpret_cast<uint4*>(p->XP) : nullptr;
        // r6: packed-f32 scores + pinned weighted V sum (pf_attn_fused2_kernel) when its 16 KB of query rows still fit; GL3_PF_FUSED_V1=1: the r4 kernel
        static const bool v1_only = getenv("GL3_PF_FUSED_V1") && atoi(getenv("GL3_PF_FUSED_V1")) != 0;
        const size_t sms2 = sms + (size_t)kvmul * FA_TB * hs * 4;
        const bool v2 = !v1_only && sms2 <= 150 * 1024;
#define GL3_FA(HS_) do { if (v2) hipLaunchKernelGGL((pf_attn_fused2_kernel<HS_>), dim3(KVH * ntile), dim3(128 * kvmul), sms2, s, aa.Q, aa.q_stride, kc1, vc1, aa.out, aa.out_stride, \
                                       KVH, kvmul, aa.kv_dim, pos0, n, aa.att_mul, fa_sstride, xqo, xpo, p->xp_tok); \
        else hipLaunchKernelGGL((pf_attn_fused_kernel<HS_>), dim3(KVH * ntile), dim3(128 * kvmul), sms, s, aa.Q, aa.q_stride, kc1, vc1, aa.out, aa.out_stride, \
                                       KVH, kvmul, aa.kv_dim, pos0, n, aa.att_mul, fa_sstride, xqo, xpo, p->xp_tok); } while (0)
        // r6: products of both phases on the matrix pipe (pf_attn_fused3_kernel: kvMul 4, head size 128 / 64); GL3_PF_FUSED_MFMA=0: the VALU kernels
        static const bool mfma_off = getenv("GL3_PF_FUSED_MFMA") && atoi(getenv("GL3_PF_FUSED_MFMA")) == 0;
        const size_t sms3 = fa3_smem_bytes(hs, fa_sstride);
        if (!mfma_off && !v1_only && kvmul == 4 && (hs == 128 || hs == 64) && sms3 <= 150 * 1024) {
            if (hs == 128) hipLaunchKernelGGL((pf_attn_fused3_kernel<128>), dim3(KVH * ntile), dim3(512), sms3, s, aa.Q, aa.q_stride, kc1, vc1, aa.out, aa.out_stride,
                                              KVH, aa.kv_dim, pos0, n, aa.att_mul, fa_sstride, xqo, xpo, p->xp_tok);
            else hipLaunchKernelGGL((pf_attn_fused3_kernel<64>), dim3(KVH * ntile), dim3(512), sms3, s, aa.Q, aa.q_stride, kc1, vc1, aa.out, aa.out_stride,
                                    KVH, aa.kv_dim, pos0, n, aa.att_mul, fa_sstride, xqo, xpo, p->xp_tok);
        }
        else if (hs == 128) GL3_FA(128);
        else if (hs == 64) GL3_FA(64);
        else GL3_FA(32);
#undef GL3_FA
        return qao;
    }
    if (tiled) {
        const int pos0 = max_pos + 1 - n, ntt = (n + PA_TB - 1) / PA_TB;
        const size_t sms = (size_t)64 * (hs + 4) * 4;
        const dim3 g1(nsplit, KVH, ntt), b1(64 * kvmul);
        const float* kc1 = aa.kcache + (size_t)one_seq * ctx->kv_seq_stride;
        static const bool pk_off = getenv("GL3_PF_SCORES_PK") && atoi(getenv("GL3_PF_SCORES_PK")) == 0;
        const size_t sms_pk = sms + (size_t)kvmul * PA_TB * hs * 4;
        const bool pk = !pk_off && sms_pk <= 150 * 1024 && (kvmul == 4 || kvmul == 2 || kvmul == 1);
#define GL3_SCORES_PK(HS_, KVM_) hipLaunchKernelGGL((pf_scores_pk_kernel<HS_, KVM_>), g1, b1, sms_pk, s, aa.Q, aa.q_stride, kc1, aa.att, aa.n_heads, kvmul, aa.kv_dim, aa.ctx, pos0, n, aa.att_mul, \
                                           rows_softmax ? p->TMX : nullptr, p->tmx_tiles)
#define GL3_SCORES(HS_) do { if (pk) { if (kvmul == 4) GL3_SCORES_PK(HS_, 4); else if (kvmul == 2) GL3_SCORES_PK(HS_, 2); else GL3_SCORES_PK(HS_, 1); } \
        else hipLaunchKernelGGL((pf_scores_tiled_kernel<HS_>), g1, b1, sms, s, aa.Q, aa.q_stride, kc1, aa.att, aa.n_heads, kvmul, aa.kv_dim, aa.ctx, pos0, n, aa.att_mul, \
                                           rows_softmax ? p->TMX : nullptr, p->tmx_tiles); } while (0)
        static const bool scm_off = getenv("GL3_PF_SCORES_MFMA") && atoi(getenv("GL3_PF_SCORES_MFMA")) == 0;
        if (rows_softmax && !scm_off && kvmul == 4 && (hs == 128 || hs == 64)) {      // r6: products on the matrix pipe, query rows resident, K tiles prefetched
            static const int scm_split = getenv("GL3_SCM_SPLIT") ? atoi(getenv("GL3_SCM_SPLIT")) : SCM_SPLIT;      // workgroups that share a (kv head, token tile)'s K tiles
            const dim3 g(nsplit < scm_split ? nsplit : scm_split, KVH, (n + SCM_TB - 1) / SCM_TB);
            const size_t sm = ((size_t)64 * (hs + 4) + 4 * SCM_TB * (hs + 2) + 4 * 4 * SCM_TB) * 4;
            if (hs == 128) hipLaunchKernelGGL((pf_scores_mfma_kernel<128>), g, dim3(512), sm, s, aa.Q, aa.q_stride, kc1, aa.att, aa.n_heads, aa.kv_dim, aa.ctx, pos0, n, aa.att_mul, p->TMX, p->tmx_tiles);
            else hipLaunchKernelGGL((pf_scores_mfma_kernel<64>), g, dim3(512), sm, s, aa.Q, aa.q_stride, kc1, aa.att, aa.n_heads, aa.kv_dim, aa.ctx, pos0, n, aa.att_mul, p->TMX, p->tmx_tiles);
        }
        else if (hs == 128) GL3_SCORES(128);
        else if (hs == 64) GL3_SCORES(64);
        else GL3_SCORES(32);
#undef GL3_SCORES
#undef GL3_SCORES_PK
        const float* sums = nullptr;
        if (rows_softmax) {
            // r6: R rows per workgroup, the sums as R chains of one wavefront; at least one workgroup per CU when the chunk has the rows
            const int rows = n * H;
            sums = p->SUMS;
            if (rows >= 64 * 256) hipLaunchKernelGGL((pf_softmax_rows_kernel<64>), dim3((rows + 63) / 64), dim3(576), 0, s, aa, rows, p->TMX, p->tmx_tiles, p->SUMS);
            else if (rows >= 32 * 256) hipLaunchKernelGGL((pf_softmax_rows_kernel<32>), dim3((rows + 31) / 32), dim3(576), 0, s, aa, rows, p->TMX, p->tmx_tiles, p->SUMS);
            else hipLaunchKernelGGL((pf_softmax_rows_kernel<16>), dim3((rows + 15) / 16), dim3(576), 0, s, aa, rows, p->TMX, p->tmx_tiles, p->SUMS);
        } else {
            const int npad = (max_pos + 1 + 63) & ~63;
            int wpw = (int)((60 * 1024) / ((size_t)npad * 4));
            wpw = wpw > 4 ? 4 : wpw;
            hipLaunchKernelGGL(pf_softmax_kernel, dim3((n * H + wpw - 1) / wpw), dim3(256), (size_t)wpw * npad * 4, s, aa, n, wpw, npad);
        }
        static const bool ring_off = getenv("GL3_PF_PV_RING") && atoi(getenv("GL3_PF_PV_RING")) == 0;
        const size_t pv_sm = (size_t)64 * (hs + PA_TB) * 4, pvr_sm = (size_t)64 * (hs + PVR_TB) * 4;
        const dim3 pvr_grid(H, (n + PVR_TB - 1) / PVR_TB), pvr_block(64 * PVR_NW);
        static const bool pvm_off = getenv("GL3_PF_PV_MFMA") && atoi(getenv("GL3_PF_PV_MFMA")) == 0;
        if (sums && !pvm_off && kvmul == 4 && (hs == 128 || hs == 64)) {      // r6: products on the matrix pipe (no uniform-address LDS reads)
            const dim3 g(KVH, (n + PVM_TB - 1) / PVM_TB);
            const size_t sm = ((size_t)64 * (hs + 4) + 4 * PVM_TB * PVM_WP) * 4;
            if (hs == 128) hipLaunchKernelGGL((pf_pv_mfma_kernel<128>), g, dim3(1024), sm, s, aa, one_seq, pos0, n, sums);
            else hipLaunchKernelGGL((pf_pv_mfma_kernel<64>), g, dim3(1024), sm, s, aa, one_seq, pos0, n, sums);
        } else if (sums && !ring_off) {
            if (hs == 128) hipLaunchKernelGGL((pf_pv_ring_kernel<128>), pvr_grid, pvr_block, pvr_sm, s, aa, one_seq, pos0, n, sums);
            else if (hs == 64) hipLaunchKernelGGL((pf_pv_ring_kernel<64>), pvr_grid, pvr_block, pvr_sm, s, aa, one_seq, pos0, n, sums);
            else hipLaunchKernelGGL((pf_pv_ring_kernel<32>), pvr_grid, pvr_block, pvr_sm, s, aa, one_seq, pos0, n, sums);
        } else if (hs > 64) hipLaunchKernelGGL((pf_pv_tiled_kernel<2>), dim3(H, ntt), dim3(256), pv_sm, s, aa, one_seq, pos0, n, sums);
        else hipLaunchKernelGGL((pf_pv_tiled_kernel<1>), dim3(H, ntt), dim3(256), pv_sm, s, aa, one_seq, pos0, n, sums);
    } else {
        const size_t sm1 = ((size_t)kvmul * d.head_size + (size_t)ATT_TT * (d.head_size + 1)) * 4;
        hipLaunchKernelGGL(pf_attn_scores_kernel, dim3(nsplit, KVH, n), dim3(64 * kvmul), sm1, s, aa);
        aa.win = ctx->attn_win;
        hipLaunchKernelGGL(pf_attn_softmax_pv_kernel, dim3(H * ((d.head_size + 63) / 64), n), dim3(64), (size_t)ctx->attn_win * 4 + 16, s, aa);
    }
    return false;
}

static bool env_flag_cached_vlq_mfma() { static const bool on = env_flag("GL3_VLQ_MFMA", true); return on; }

// Batched matmul of the f32-activation weight types (gl3_prefill_vl.h): out[b][row] (+)= dot(W[row], act[b]) in the Vector-API order
template <int EPI>
static void launch_gemm_vl(gl3_ctx* ctx, const Q8Mat& w, int ntok, const float* act, int act_stride, float* out, int out_stride, float out_scale = 1.0f) {
    VlGemmArgs a{};
    a.w = w.w; a.rows = w.rows; a.k = w.k; a.X = act; a.x_stride = act_stride; a.ntok = ntok; a.out = out; a.out_stride = out_stride; a.out_scale = out_scale;
    a.nrt = (w.rows + 63) / 64;
    if (w.fmt == GL3_TYPE_F16) {
        a.ntt = (ntok + F16G_TOK - 1) / F16G_TOK;
        hipLaunchKernelGGL((gemm_f16_mfma_kernel<EPI>), dim3(8 * ((a.nrt * a.ntt + 7) / 8)), dim3(256), 2 * F16G_STAGE, ctx->stream, a);
    } else if (ntok > VLQ_TOK && env_flag_cached_vlq_mfma() && a.nrt * ((ntok + VQM_TOK - 1) / VQM_TOK) >= 192) {
        // enough 64 x 64 tiles to fill the chip: products on the f32 matrix cores (gemm_vlq_mfma_kernel; 8B Q4_0 pp512 2.53 k -> 3.4 k
        // tok/s).  Fewer tiles (short chunks, the 4096-row projections at 128 tokens) keep the 16-token VALU kernel; GL3_VLQ_MFMA=0: always.
        a.ntt = (ntok + VQM_TOK - 1) / VQM_TOK;
        static const bool by_tokens = env_flag("GL3_VQM_XCD_TOKENS", true);      // A/B switch of the tile mapping (vqm_tile_of)
        a.xcd_tokens = by_tokens ? 1 : 0;
        const dim3 g(by_tokens ? vqm_grid(a.nrt, a.ntt) : 8 * ((a.nrt * a.ntt + 7) / 8));
        const size_t sm = (size_t)2 * VQM_STAGE_FLOATS * 4;
        static const bool occ2 = env_flag("GL3_VQM_OCC2", false);                // A/B: the 143-register build, one workgroup per CU (5 % slower)
        if (w.fmt == GL3_TYPE_Q4_0) {
            if (occ2) hipLaunchKernelGGL((gemm_vlq_mfma_kernel<WT_Q4_0, EPI, 2>), g, dim3(512), sm, ctx->stream, a);
            else hipLaunchKernelGGL((gemm_vlq_mfma_kernel<WT_Q4_0, EPI, 4>), g, dim3(512), sm, ctx->stream, a);
        } else {
            if (occ2) hipLaunchKernelGGL((gemm_vlq_mfma_kernel<WT_Q8_0, EPI, 2>), g, dim3(512), sm, ctx->stream, a);
            else hipLaunchKernelGGL((gemm_vlq_mfma_kernel<WT_Q8_0, EPI, 4>), g, dim3(512), sm, ctx->stream, a);
        }
    } else {
        a.ntt = (ntok + VLQ_TOK - 1) / VLQ_TOK;
        const dim3 g(8 * ((a.nrt * a.ntt + 7) / 8));
        if (w.fmt == GL3_TYPE_Q4_0) hipLaunchKernelGGL((gemm_vlq_kernel<WT_Q4_0, EPI>), g, dim3(256), 2 * vlq_stage_floats<WT_Q4_0>() * 4, ctx->stream, a);
        else hipLaunchKernelGGL((gemm_vlq_kernel<WT_Q8_0, EPI>), g, dim3(256), 2 * vlq_stage_floats<WT_Q8_0>() * 4, ctx->stream, a);
    }
}

// The same layers for F16 / Q4_0 / Q8_0-with-f32-activation matrices: RMSNorm to f32 (exact sum of squares), GEMMs on the f32
// activations, gate and up as two GEMMs + an element-wise SwiGLU.  Tensor parallel (r4): the row-split matrices write this rank's
// chunk of the rank-chunked X / AO / HB (as the int8 path does), the gathers are in place, and the next GEMM's operand is the
// gathered activation un-chunked (or normalised) into XN.
static int32_t pf_layers_vl(gl3_ctx* ctx, int n, int max_pos, int one_seq) {
    Gl3Range chunk_range("gl3 batched step, tokens", n);
    gl3_prefill_state* p = ctx->pf;
    const gl3_model_desc& d = ctx->d;
    hipStream_t s = ctx->stream;
    const int rank = d.tp_rank, tp = d.tp_size, qd = ctx->q_dim_l, kvd = ctx->kv_dim_l, hid = ctx->hidden_l, dml = ctx->dim_l;
    const int qkv_dim = qd + 2 * kvd;
    float* Xr = p->X + (size_t)rank * n * dml;
    float* AOr = p->AO + (size_t)rank * n * qd;
    float* HBr = p->HB + (size_t)rank * n * hid;
    int32_t r;
    // every rank holds the whole embedding table: the full X, written in the rank-chunked layout
    if (ctx->emb.fmt == GL3_TYPE_F16) hipLaunchKernelGGL((pf_embed_vl_kernel<WT_F16>), dim3(n), dim3(256), 0, s, ctx->emb.w, d.dim, p->tokens, p->X, ctx->emb_scale, dml);
    else if (ctx->emb.fmt == GL3_TYPE_Q4_0) hipLaunchKernelGGL((pf_embed_vl_kernel<WT_Q4_0>), dim3(n), dim3(256), 0, s, ctx->emb.w, d.dim, p->tokens, p->X, ctx->emb_scale, dml);
    else hipLaunchKernelGGL((pf_embed_vl_kernel<WT_Q8_0>), dim3(n), dim3(256), 0, s, ctx->emb.w, d.dim, p->tokens, p->X, ctx->emb_scale, dml);
    const size_t nq = (size_t)(d.dim + 32) * 4 + ss_scratch_bytes(d.dim) + 64;
    auto unchunk = [&](const float* src, int k, int cc) {          // rank-chunked [tp][n][cc] -> plain XN[n][k] (tp = 1: a copy the GEMM could skip, kept for one code path)
        hipLaunchKernelGGL((pf_norm_quant_kernel<PQ_PLAIN_F32>), dim3(n, (k / 4 + 255) / 256), dim3(256), 0, s, src, k, cc, (const float*)nullptr, 0.f, (uint8_t*)nullptr, p->XN, 0, 0);
    };
    for (int l = 0; l < d.n_layers; ++l) {
        gl3_layer& L = ctx->layers[l];
        Gl3Range layer_range("layer", l);
        hipLaunchKernelGGL((pf_norm_quant_kernel<PQ_NORM_F32>), dim3(n), dim3(256), nq, s, p->X, d.dim, dml, L.attn_norm, d.rms_eps, (uint8_t*)nullptr, p->XN, 0, 0);
        launch_gemm_vl<EPI_STORE>(ctx, L.wqkv, n, p->XN, d.dim, p->QKV, qkv_dim);
        pf_attention(ctx, l, n, max_pos, one_seq, AOr, false);
        if ((r = gl3_all_gather(ctx, GB_PF_AO, (size_t)n * qd)) != GL3_OK) return r;
        const float* ao = p->AO;
        if (tp > 1) { unchunk(p->AO, ctx->q_dim, qd); ao = p->XN; }
        if (ctx->wo_replicated) {
            // every rank holds all of Wo: one GEMM per rank chunk of the rank-chunked X (rows [c dml, (c + 1) dml) -> chunk c), no gather
            for (int c = 0; c < tp; ++c) {
                Q8Mat sub = L.wo;
                sub.rows = dml;
                sub.w = L.wo.w + (size_t)(c * dml / 8) * L.wo.vl_group_bytes();
                launch_gemm_vl<EPI_RESID>(ctx, sub, n, ao, ctx->q_dim, p->X + (size_t)c * n * dml, dml, ctx->resid_scale);
            }
        } else {
            launch_gemm_vl<EPI_RESID>(ctx, L.wo, n, ao, ctx->q_dim, Xr, dml, ctx->resid_scale);
            if ((r = gl3_all_gather(ctx, GB_PF_X, (size_t)n * dml)) != GL3_OK) return r;
        }
        hipLaunchKernelGGL((pf_norm_quant_kernel<PQ_NORM_F32>), dim3(n), dim3(256), nq, s, p->X, d.dim, dml, L.ffn_norm, d.rms_eps, (uint8_t*)nullptr, p->XN, 0, 0);
        launch_gemm_vl<EPI_STORE>(ctx, L.w1, n, p->XN, d.dim, HBr, hid);
        launch_gemm_vl<EPI_STORE>(ctx, L.w3, n, p->XN, d.dim, p->HB2, hid);
        const size_t ne = (size_t)n * hid;
        hipLaunchKernelGGL(pf_swiglu_kernel, dim3((unsigned)((ne + 255) / 256)), dim3(256), 0, s, HBr, p->HB2, ne);
        if ((r = gl3_all_gather(ctx, GB_PF_HB, (size_t)n * hid)) != GL3_OK) return r;
        const float* hb = p->HB;
        if (tp > 1) { unchunk(p->HB, d.hidden, hid); hb = p->XN; }
        launch_gemm_vl<EPI_RESID>(ctx, L.w2, n, hb, d.hidden, Xr, dml, ctx->resid_scale);
        if ((r = gl3_all_gather(ctx, GB_PF_X, (size_t)n * dml)) != GL3_OK) return r;
    }
    GL3_HIP(hipGetLastError());
    return GL3_OK;
}

// All layers for n tokens whose (token, sequence, position) are already on the device.  max_pos = largest position.
// one_seq >= 0: all n tokens belong to that sequence at consecutive positions ending at max_pos (prefill).
static int32_t pf_layers(gl3_ctx* ctx, int n, int max_pos, int one_seq) {
    Gl3Range chunk_range("gl3 batched step, tokens", n);
    gl3_prefill_state* p = ctx->pf;
    if (p->vl) return pf_layers_vl(ctx, n, max_pos, one_seq);
    const gl3_model_desc& d = ctx->d;
    hipStream_t s = ctx->stream;
    // tensor parallel: this rank's heads / hidden units / dim rows; activations that are gathered use the rank-chunked layout
    const int rank = d.tp_rank, qd = ctx->q_dim_l, kvd = ctx->kv_dim_l;
    const int hid = ctx->hidden_l, dml = ctx->dim_l;
    const int qkv_dim = qd + 2 * kvd;
    const bool fused_decode = ctx->fused_attn_ok && max_pos < AF_MAXN && !(getenv("GL3_NO_FUSED_BD_ATTN") && atoi(getenv("GL3_NO_FUSED_BD_ATTN")));
    // small batch on one rank: the attention output and hb leave their kernels already quantised for the next GEMM (no
    // separate quantise launches; under tensor parallelism the f32 vectors are gathered first, so the launches stay)
    static const bool fuse_off = getenv("GL3_NO_FUSED_QUANT") && atoi(getenv("GL3_NO_FUSED_QUANT"));
    const bool fuse_q = !fuse_off && bd_tslots(n) != 0 && d.tp_size == 1 && (d.hidden % 32) == 0 && (d.head_size % 32) == 0;
    float* Xr = p->X + (size_t)rank * n * dml;           // this rank's chunk of X / AO / HB
    float* AOr = p->AO + (size_t)rank * n * qd;
    float* HBr = p->HB + (size_t)rank * n * hid;
    int32_t r;
    hipLaunchKernelGGL(pf_embed_kernel, dim3(n), dim3(256), 0, s, ctx->emb.w, ctx->emb.ng, d.dim, p->tokens, p->X, dml, ctx->emb_scale);
    auto nq_smem = [&](int k) { return (size_t)(k + 32) * 4 + ss_scratch_bytes(k) + 64; };
    for (int l = 0; l < d.n_layers; ++l) {
        gl3_layer& L = ctx->layers[l];
        Gl3Range layer_range("layer", l);
        hipLaunchKernelGGL((pf_norm_quant_kernel<PQ_NORM>), dim3(n), dim3(256), nq_smem(d.dim), s, p->X, d.dim, dml, L.attn_norm, d.rms_eps,
                           p->XQ, p->XS, p->maxk, bd_tslots(n), (pf_use_gemm3() && n > 64) ? (uint2*)p->XP : nullptr, p->xp_tok);
        launch_gemm<EPI_STORE>(ctx, L.wqkv, nullptr, n, p->QKV, qkv_dim);
        const bool quantised_ao = pf_attention(ctx, l, n, max_pos, one_seq, AOr, fuse_q && one_seq < 0 && fused_decode);
        if ((r = gl3_all_gather(ctx, GB_PF_AO, (size_t)n * qd)) != GL3_OK) return r;
        if (!quantised_ao)
            hipLaunchKernelGGL((pf_norm_quant_kernel<PQ_PLAIN>), dim3(n, (ctx->q_dim / 4 + 255) / 256), dim3(256), 0, s, p->AO, ctx->q_dim, qd, (const float*)nullptr,
                               0.f, p->XQ, p->XS, p->maxk, bd_tslots(n), (pf_use_gemm3() && n > 64) ? (uint2*)p->XP : nullptr, p->xp_tok);
        if (ctx->wo_replicated) {
            // every rank holds all of Wo: one GEMM per rank chunk of the rank-chunked X (rows [c dml, (c + 1) dml) -> chunk c), no gather
            for (int c = 0; c < d.tp_size; ++c) {
                Q8Mat sub = L.wo;
                sub.rows = dml; sub.nstrips = (dml + 15) / 16;
                sub.w = L.wo.w + (size_t)(c * dml / 16) * L.wo.ng * TILE_BYTES;
                launch_gemm<EPI_RESID>(ctx, sub, nullptr, n, p->X + (size_t)c * n * dml, dml, ctx->resid_scale);
            }
        } else {
            launch_gemm<EPI_RESID>(ctx, L.wo, nullptr, n, Xr, dml, ctx->resid_scale);
            if ((r = gl3_all_gather(ctx, GB_PF_X, (size_t)n * dml)) != GL3_OK) return r;
        }
        hipLaunchKernelGGL((pf_norm_quant_kernel<PQ_NORM>), dim3(n), dim3(256), nq_smem(d.dim), s, p->X, d.dim, dml, L.ffn_norm, d.rms_eps,
                           p->XQ, p->XS, p->maxk, bd_tslots(n), (pf_use_gemm3() && n > 64) ? (uint2*)p->XP : nullptr, p->xp_tok);
        // > 64 tokens on one rank: the tall gate + up tiling writes hb quantised (no f32 round trip, no quantise launch)
        const bool fuse_big = !fuse_off && n > 64 && d.tp_size == 1 && p->XQh && pf_use_gemm3() && gl3_gemm3_swiglu_quantises(L.w1.rows, n);
        if (fuse_q || fuse_big) {
            launch_gemm<EPI_SWIGLU>(ctx, L.w1, &L.w3, n, HBr, hid, 1.0f, false, true);
            launch_gemm<EPI_RESID>(ctx, L.w2, nullptr, n, Xr, dml, ctx->resid_scale, true);
        } else {
            launch_gemm<EPI_SWIGLU>(ctx, L.w1, &L.w3, n, HBr, hid);
            if ((r = gl3_all_gather(ctx, GB_PF_HB, (size_t)n * hid)) != GL3_OK) return r;
            hipLaunchKernelGGL((pf_norm_quant_kernel<PQ_PLAIN>), dim3(n, (d.hidden / 4 + 255) / 256), dim3(256), 0, s, p->HB, d.hidden, hid, (const float*)nullptr, 0.f,
                               p->XQ, p->XS, p->maxk, bd_tslots(n), (pf_use_gemm3() && n > 64) ? (uint2*)p->XP : nullptr, p->xp_tok);
            launch_gemm<EPI_RESID>(ctx, L.w2, nullptr, n, Xr, dml, ctx->resid_scale);
        }
        if ((r = gl3_all_gather(ctx, GB_PF_X, (size_t)n * dml)) != GL3_OK) return r;
    }
    GL3_HIP(hipGetLastError());
    return GL3_OK;
}

float* gl3_prefill_buf(gl3_ctx* ctx, int which) {
    gl3_prefill_state* p = ctx->pf;
    return which == GB_PF_X ? p->X : which == GB_PF_AO ? p->AO : which == GB_PF_HB ? p->HB : p->LOGITS;
}

// x of token b from the rank-chunked X into the decode path's plain ctx->x (parity tap gl3_get_x)
static __global__ void pf_unchunk_row_kernel(const float* __restrict__ X, int b, int dim, int cc, int ntok, float* __restrict__ out) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < dim; i += gridDim.x * blockDim.x) out[i] = X[chunked(b, i, cc, ntok)];
}

static int32_t pf_stage_tokens(gl3_ctx* ctx, const int32_t* tokens, const int32_t* seqs, const int32_t* poss, int n) {
    gl3_prefill_state* p = ctx->pf;
    GL3_HIP(hipMemcpyAsync(p->tokens, tokens, (size_t)n * sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream));
    GL3_HIP(hipMemcpyAsync(p->seqpos, seqs, (size_t)n * sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream));
    GL3_HIP(hipMemcpyAsync(p->seqpos + p->max_batch, poss, (size_t)n * sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream));
    return GL3_OK;
}

int32_t gl3_prefill_run(gl3_ctx* ctx, int32_t seq, const int32_t* tokens, int32_t n, int32_t start_pos) {
    gl3_prefill_state* p = ctx->pf;
    const gl3_model_desc& d = ctx->d;
    for (int i = 0; i < n; ++i)
        if (tokens[i] < 0 || tokens[i] >= d.vocab) GL3_FAIL(GL3_E_ARG, "token id out of range");
    GL3_HIP(hipSetDevice(d.device));
    std::vector<int32_t> seqs(n, seq), poss(n);
    for (int i = 0; i < n; ++i) poss[i] = start_pos + i;
    int32_t r = pf_stage_tokens(ctx, tokens, seqs.data(), poss.data(), n);
    if (r != GL3_OK) return r;
    if ((r = pf_layers(ctx, n, start_pos + n - 1, seq)) != GL3_OK) return r;
    // keep the decode path's x in step with the last prefilled token (parity tap gl3_get_x)
    hipLaunchKernelGGL(pf_unchunk_row_kernel, dim3(4), dim3(256), 0, ctx->stream, p->X, n - 1, d.dim, ctx->dim_l, n, ctx->x);
    GL3_HIP(hipStreamSynchronize(ctx->stream));
    return gl3_tp_check(ctx);
}

// One decode step of n independent sequences = the prefill machinery over (token, sequence, position) triples +
// final RMSNorm + vocab projection for every row (the n matvecs become one GEMM over the shared weights).
int32_t gl3_decode_batch_run(gl3_ctx* ctx, const int32_t* tokens, const int32_t* seq_ids, const int32_t* positions, int32_t n,
                             float* logits_out, int32_t* argmax_out) {
    gl3_prefill_state* p = ctx->pf;
    const gl3_model_desc& d = ctx->d;
    GL3_HIP(hipSetDevice(d.device));
    if (p->logits_rows < n && p->in_arena) GL3_FAIL(GL3_E_UNSUPPORTED, "tensor-parallel static-batched decode is limited to 64 sequences per step");
    if (p->logits_rows < n) {
        for (auto& ge : p->step_graphs) if (ge) { hipGraphExecDestroy(ge); ge = nullptr; }      // captured steps point at the old buffer
        if (p->LOGITS) hipFree(p->LOGITS);
        p->LOGITS = nullptr; p->logits_rows = 0;
        GL3_HIP(hipMalloc((void**)&p->LOGITS, (size_t)n * d.vocab * 4));
        p->logits_rows = n;
    }
    int max_pos = 0;
    for (int i = 0; i < n; ++i) max_pos = positions[i] > max_pos ? positions[i] : max_pos;
    int32_t r = pf_stage_tokens(ctx, tokens, seq_ids, positions, n);
    if (r != GL3_OK) return r;
    hipStream_t s = ctx->stream;
    const int vl = ctx->vocab_l;
    // the whole step: layers, final RMSNorm + vocabulary projection of every row, greedy ids
    auto enqueue_step = [&](int mp) -> int32_t {
        int32_t rr = pf_layers(ctx, n, mp, -1);
        if (rr != GL3_OK) return rr;
        const size_t nq = (size_t)(d.dim + 32) * 4 + ss_scratch_bytes(d.dim) + 64;
        if (p->vl) {
            hipLaunchKernelGGL((pf_norm_quant_kernel<PQ_NORM_F32>), dim3(n), dim3(256), nq, s, p->X, d.dim, ctx->dim_l, ctx->out_norm, d.rms_eps, (uint8_t*)nullptr, p->XN, 0, 0);
            launch_gemm_vl<EPI_STORE>(ctx, ctx->wcls, n, p->XN, d.dim, p->LOGITS + (size_t)d.tp_rank * n * vl, vl, ctx->logit_scale);
        } else {
        hipLaunchKernelGGL((pf_norm_quant_kernel<PQ_NORM>), dim3(n), dim3(256), nq, s, p->X, d.dim, ctx->dim_l, ctx->out_norm, d.rms_eps, p->XQ, p->XS, p->maxk, bd_tslots(n), (pf_use_gemm3() && n > 64) ? (uint2*)p->XP : nullptr, p->xp_tok);
        // vocab rows are split across ranks: this rank's logits are the chunk [n][vocab / tp] of the rank-chunked buffer
        launch_gemm<EPI_STORE>(ctx, ctx->wcls, nullptr, n, p->LOGITS + (size_t)d.tp_rank * n * vl, vl, ctx->logit_scale);
        }
        if ((rr = gl3_all_gather(ctx, GB_PF_LOGITS, (size_t)n * vl)) != GL3_OK) return rr;
        hipLaunchKernelGGL(pf_argmax_part_kernel, dim3(AMX_SPLIT, n), dim3(256), 0, s, p->LOGITS, d.vocab, vl, p->amx_v, p->amx_i);
        hipLaunchKernelGGL(pf_argmax_fold_kernel, dim3(n), dim3(64), 0, s, p->amx_v, p->amx_i, p->amax);
        return GL3_OK;
    };
    // ~400 launches per step: replay them as one hipGraph per batch size.  Nothing position-dependent is baked in when every
    // position is below AF_MAXN (the one-launch attention reads sequence ids / positions from device memory).
    static const bool graphs_off = getenv("GL3_NO_GRAPH") && atoi(getenv("GL3_NO_GRAPH"));
    const bool graphable = !graphs_off && !gl3_roctx_on() && !(d.flags & GL3_FLAG_NO_GRAPH) && ctx->fused_attn_ok && max_pos < AF_MAXN && ctx->transport != GL3_TP_RCCL;
    if (graphable) {
        if ((int)p->step_graphs.size() <= n) p->step_graphs.resize(n + 1, nullptr);
        if (!p->step_graphs[n]) {
            hipGraph_t g = nullptr;
            GL3_HIP(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
            r = enqueue_step(0);
            const hipError_t e = hipStreamEndCapture(s, &g);
            if (r != GL3_OK) return r;
            GL3_HIP(e);
            GL3_HIP(hipGraphInstantiate(&p->step_graphs[n], g, nullptr, nullptr, 0));
            hipGraphDestroy(g);
        }
        GL3_HIP(hipGraphLaunch(p->step_graphs[n], s));
    } else if ((r = enqueue_step(max_pos)) != GL3_OK) return r;
    if (argmax_out) GL3_HIP(hipMemcpyAsync(argmax_out, p->amax, (size_t)n * sizeof(int32_t), hipMemcpyDeviceToHost, s));
    if (logits_out) {      // un-chunk on the way out: [tp][n][vl] -> [n][vocab]
        for (int c = 0; c < d.tp_size; ++c)
            GL3_HIP(hipMemcpy2DAsync(logits_out + (size_t)c * vl, (size_t)d.vocab * 4, p->LOGITS + (size_t)c * n * vl, (size_t)vl * 4, (size_t)vl * 4, n,
                                     hipMemcpyDeviceToHost, s));
    }
    GL3_HIP(hipGetLastError());
    GL3_HIP(hipStreamSynchronize(s));
    return gl3_tp_check(ctx);
}


// Average device time of one batched-prefill GEMM class at n tokens: one HIP event pair around `iters` sweeps over every
// layer's weights (GEMMs are 70-250 us, the ~1.5 us boundary is noise).  int8_ops = 2 * rows * K * n per launch (MFMA work
// only; the f32 scale-and-accumulate epilogue the reference arithmetic needs is not counted).
int32_t gl3_prefill_profile(gl3_ctx* ctx, int klass, int n, int iters, double* out_us, uint64_t* int8_ops) {
    gl3_prefill_state* p = ctx->pf;
    const gl3_model_desc& d = ctx->d;
    if (!p || p->vl) GL3_FAIL(GL3_E_UNSUPPORTED, "the int8 GEMM profile needs max_batch > 1 and Q8_0 weights with the int8 activation");
    if (n < 1 || n > p->max_batch) GL3_FAIL(GL3_E_ARG, "token count outside 1..max_batch");
    GL3_HIP(hipSetDevice(d.device));
    const int qkv_dim = ctx->q_dim_l + 2 * ctx->kv_dim_l;
    auto sweep = [&]() {
        for (int l = 0; l < d.n_layers; ++l) {
            gl3_layer& L = ctx->layers[l];
            switch (klass) {
            case GL3_K_MATVEC_QKV: launch_gemm<EPI_STORE>(ctx, L.wqkv, nullptr, n, p->QKV, qkv_dim); break;
            case GL3_K_MATVEC_WO: launch_gemm<EPI_RESID>(ctx, L.wo, nullptr, n, p->X, ctx->wo_rows); break;
            case GL3_K_MATVEC_GATEUP: launch_gemm<EPI_SWIGLU>(ctx, L.w1, &L.w3, n, p->HB, ctx->hidden_l); break;
            default: launch_gemm<EPI_RESID>(ctx, L.w2, nullptr, n, p->X, ctx->dim_l); break;
            }
        }
    };
    hipEvent_t e0, e1;
    GL3_HIP(hipEventCreate(&e0)); GL3_HIP(hipEventCreate(&e1));
    sweep();
    GL3_HIP(hipEventRecord(e0, ctx->stream));
    for (int i = 0; i < iters; ++i) sweep();
    GL3_HIP(hipEventRecord(e1, ctx->stream));
    GL3_HIP(hipEventSynchronize(e1));
    float ms = 0;
    GL3_HIP(hipEventElapsedTime(&ms, e0, e1));
    hipEventDestroy(e0); hipEventDestroy(e1);
    *out_us = (double)ms * 1e3 / ((double)iters * d.n_layers);
    if (int8_ops) {
        const gl3_layer& L = ctx->layers[0];
        const Q8Mat& w = klass == GL3_K_MATVEC_QKV ? L.wqkv : klass == GL3_K_MATVEC_WO ? L.wo : klass == GL3_K_MATVEC_GATEUP ? L.w1 : L.w2;
        *int8_ops = (uint64_t)2 * w.rows * w.k * n * (klass == GL3_K_MATVEC_GATEUP ? 2 : 1);
    }
    return GL3_OK;
}
