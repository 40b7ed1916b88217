// gl3_prefill_gemm4.h — batched-prefill Q8_0 GEMM (> 64 tokens), round 4, high-occupancy form of gl3_prefill_gemm2.h.
//
// Same arithmetic (see gl3_prefill_gemm2.h for the three exact identities): per (32 x 32 tile, block) one
// v_mfma_i32_32x32x32_i8 (biased int8 dot D), two v_mfma_f32_32x32x8_bf16_1k (s = wScale aScale and -B s as exact bf16-split
// outer products), then acc += fma(D, s, -B s) — 16 v_fma_f32 + 16 v_add_f32 per lane.
//
// What the in-kernel stamps of pf_gemm2_kernel showed (profiles/r04_gemm_experiments.md): with four tiles per wavefront the
// kernel needs ~250 VGPRs, so only two wavefronts share a SIMD, and every VMEM instruction (LDS-DMA piece or plain load) costs
// the issuing wavefront ~80 cycles that the one other wavefront cannot fill: arithmetic alone 123 cycles per (tile, block) and
// SIMD, with the data movement 200.  Here a wavefront owns ONE tile (~120 VGPRs): a workgroup of 4 WR wavefronts (WR row
// fragments x 4 token fragments = 32 WR weight rows x 128 tokens) puts FOUR wavefronts on every SIMD (WR = 4: one workgroup per
// CU; WR = 2: two), which hide each other's LDS-DMA issue, LDS latency and MFMA result latency without a software pipeline.
// Price: the LDS operand reads per MFMA double (one A and one B fragment per tile instead of 2 + 2 per four) — 45 % of the LDS
// read rate at the matrix pipe's speed.
#pragma once
#include "gl3_prefill_gemm2.h"

template <int EPI, int WR, int OCCW>
__global__ __launch_bounds__(256 * WR, OCCW) void pf_gemm4_kernel(const GemmArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    constexpr int KB = G2_KB, TOK = G2_TOK, RING = G2_RING;
    constexpr int NM = (EPI == EPI_SWIGLU) ? 2 : 1;
    constexpr int NW = 4 * WR, NT = 64 * NW;
    constexpr int AROWS = 32 * WR;                     // weight rows staged per K stage (all matrices)
    constexpr int RPM = AROWS / NM;                    // output rows per matrix covered by this workgroup
    static_assert(WR == 2 || WR == 4, "8 or 16 wavefronts");
    constexpr int OFF_AT = KB * 2 * AROWS * 16, OFF_BQ = 2 * OFF_AT, OFF_BS = OFF_BQ + KB * 2 * TOK * 16;
    constexpr int STAGE = g2_stage_bytes(AROWS);
    constexpr int NLA = KB * 2 * AROWS / 64, NLB = KB * 2 * TOK / 64;      // LDS-DMA wave-loads per stage: weights, activations
    constexpr int NDMA = (NLA + NLB + NW - 1) / NW;
    constexpr int NAT = AROWS * KB, NBS = TOK * KB;                        // scale-operand entries per stage
    static_assert(NAT <= 256 && NBS == 256 && NT >= 512, "threads [0, NAT) build the weight entries, [256, 512) the token entries");
    const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int tl = lane & 31, hi = lane >> 5;
    const int wr = wave >> 2, wc = wave & 3;           // this wavefront's tile: row fragment wr, token fragment wc
    const int ntt_g = a.ntt, per_xcd = (a.ntt * a.nrt + 7) >> 3;
    const int lin = blockIdx.x, J = (lin & 7) * per_xcd + (lin >> 3);       // XCD-aware: token tiles of a row tile on one XCD
    if (J >= ntt_g * a.nrt) return;
    const int row0 = (J / ntt_g) * RPM;
    const int tok0 = (J % ntt_g) * TOK;
    const uint32_t strip_bytes = (uint32_t)a.ng * TILE_BYTES;
    const int nkb = (a.nb + KB - 1) / KB;
    const int nstrips = (a.rows + 15) >> 4;
    // local row lr of the staged image -> (matrix, row inside the matrix' RPM rows)
    auto row_ptr = [&](int lr) -> const uint8_t* {     // start of local row lr's strip (tile group 0)
        const int m = NM == 2 ? lr / RPM : 0, r = NM == 2 ? lr % RPM : lr;
        return (m ? a.w2 : a.w) + (size_t)min(nstrips - 1, (row0 + r) >> 4) * strip_bytes;
    };

    // ---- this wavefront's LDS-DMA pieces: per-lane source pointers (stage-independent part); destination = piece id * 1 KB
    const uint8_t* dma_src[NDMA];
#pragma unroll
    for (int u = 0; u < NDMA; ++u) {
        const int j = wave + NW * u;                   // piece id: [0, NLA) weights, [NLA, NLA + NLB) activations
        if (j < NLA) {
            const int e = 64 * j + lane, c = e / AROWS, row = e % AROWS;       // c = blk * 2 + half
            dma_src[u] = row_ptr(row) + ((c & 1) ? 1152 : 128) + 16 * ((c >> 1) * 16 + (row & 15));
        } else {
            const int jb = min(j - NLA, NLB - 1), c = jb / (TOK / 64), tk = (((jb % (TOK / 64)) * 64 + lane) ^ c);   // LDS slot p holds token p ^ c
            dma_src[u] = a.XQ + (size_t)min(a.ntok - 1, tok0 + tk) * a.maxk + 16 * c;
        }
    }
    auto dma_stage = [&](int kb, int slot) {
        const uint32_t aoff = (uint32_t)(kb >> 1) * TILE_BYTES + (kb & 1) * 512, boff = (uint32_t)kb * (32 * KB);
        uint8_t* base = smem + slot * STAGE;
#pragma unroll
        for (int u = 0; u < NDMA; ++u) {
            const int j = wave + NW * u;
            if (j < NLA) g2_dma16(dma_src[u] + aoff, base + 1024 * j);
            else if (j < NLA + NLB) g2_dma16(dma_src[u] + boff, base + OFF_BQ + 1024 * (j - NLA));
        }
    };
    // ---- scale operands (gl3_prefill_gemm2.h): one entry per thread
    const int s_row = t % AROWS, s_blk = (t / AROWS) % KB, x_tok = t % TOK, x_blk = (t / TOK) % KB;
    const uint8_t* s_wp = row_ptr(s_row) + 2 * (s_blk * 16 + (s_row & 15));
    const float* s_xp = a.XS + (size_t)min(a.ntok - 1, tok0 + x_tok) * (a.maxk >> 5) + x_blk;
    const bool do_at = t < NAT, do_bs = t >= 256 && t < 512;
    uint16_t r_ws = 0;
    float r_xs = 0.f;
    auto scale_load = [&](int kb) {
        if (do_at) r_ws = *reinterpret_cast<const uint16_t*>(s_wp + (size_t)(kb >> 1) * TILE_BYTES + (kb & 1) * 64);
        if (do_bs) r_xs = s_xp[kb * KB];
    };
    auto scale_store = [&](int kb, int slot) {
        uint8_t* base = smem + slot * STAGE;
        if (do_at) {
            const float wf = h2f(r_ws);
            const float whi = __uint_as_float(__float_as_uint(wf) & 0xFFFF0000u), wlo = wf - whi;
            const v4i_t lo = {(int)g2_bf16_dup(whi), (int)g2_bf16_dup(wlo), (int)g2_bf16_dup(whi * -8388608.f), (int)g2_bf16_dup(wlo * -8388608.f)};
            const v4i_t hh = {0, 0, (int)g2_bf16_dup(whi * -4194304.f), (int)g2_bf16_dup(wlo * -4194304.f)};
            *reinterpret_cast<v4i_t*>(base + OFF_AT + ((size_t)(s_blk * 2 + 0) * AROWS + s_row) * 16) = lo;
            *reinterpret_cast<v4i_t*>(base + OFF_AT + ((size_t)(s_blk * 2 + 1) * AROWS + s_row) * 16) = hh;
        }
        if (do_bs) {
            const float av = kb * KB + x_blk < a.nb ? r_xs : 0.f;       // ragged K: zero scale for the padded blocks
            const float ahi = __uint_as_float(__float_as_uint(av) & 0xFFFF0000u), alo = av - ahi;
            const uint32_t p = (__float_as_uint(ahi) >> 16) | (__float_as_uint(alo) & 0xFFFF0000u);
            *reinterpret_cast<uint2*>(base + OFF_BS + ((size_t)x_blk * TOK + x_tok) * 8) = make_uint2(p, p);
        }
    };

    float acc[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    v16i_t cbias;
#pragma unroll
    for (int r = 0; r < 16; ++r) cbias[r] = 0x4B400000;
    asm volatile("" : "+v"(cbias));
    const v16f2_t zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

    // operand fragments of a block, double-buffered by block parity
    v4i_t af[2], at[2], bf[2];
    v4s_t bp[2];
    const int tk = wc * 32 + tl;
    const uint32_t la = (uint32_t)((hi * AROWS + wr * 32 + tl) * 16), lp = (uint32_t)(OFF_BS + tk * 8);
    uint32_t lb[KB];
#pragma unroll
    for (int b = 0; b < KB; ++b) lb[b] = (uint32_t)(OFF_BQ + ((b * 2 + hi) * TOK + (tk ^ (b * 2 + hi))) * 16);
    auto load_ops = [&](const uint8_t* sb, int blk, int buf) {
        af[buf] = *reinterpret_cast<const v4i_t*>(sb + la + blk * (2 * AROWS * 16));
        at[buf] = *reinterpret_cast<const v4i_t*>(sb + OFF_AT + la + blk * (2 * AROWS * 16));
        bf[buf] = *reinterpret_cast<const v4i_t*>(sb + lb[blk]);
        bp[buf] = *reinterpret_cast<const v4s_t*>(sb + lp + blk * (TOK * 8));
    };

    // ---- prologue: stages 0 and 1 in the ring
    dma_stage(0, 0);
    scale_load(0);
    scale_store(0, 0);
    if (nkb > 1) {
        dma_stage(1, 1);
        scale_load(1);
        scale_store(1, 1);
    }
    __syncthreads();                                   // (waits for the LDS-DMA too)
    load_ops(smem, 0, 0);
    int cur = 0;
    for (int kb = 0; kb < nkb; ++kb) {
        const int nxt = cur == RING - 1 ? 0 : cur + 1, fil = nxt == RING - 1 ? 0 : nxt + 1;
        const int kf = min(kb + 2, nkb - 1);           // past the last stage: refill a slot nobody reads (no branches)
#if defined(G4_SKIP_MEM)
        (void)kf;
#elif defined(G4_SAME_STAGE)
        scale_load(0);
        dma_stage(0, fil);
#elif defined(G4_NO_DMA_ONLY)
        scale_load(kf);
#elif defined(G4_ASYNC_EXPERIMENT)
        dma_stage(kf, fil);
#else
        scale_load(kf);
        dma_stage(kf, fil);
#endif
        const uint8_t* sb_cur = smem + cur * STAGE;
        const uint8_t* sb_nxt = smem + nxt * STAGE;
#pragma unroll
        for (int blk = 0; blk < KB; ++blk) {
            const int buf = blk & 1;
            const v16i_t D = __builtin_amdgcn_mfma_i32_32x32x32_i8(af[buf], bf[buf], cbias, 0, 0, 0);
            const v16f2_t S = __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(__builtin_bit_cast(v4s_t, v2i_t{at[buf][0], at[buf][1]}), bp[buf], zero16, 0, 0, 0);
            const v16f2_t N = __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(__builtin_bit_cast(v4s_t, v2i_t{at[buf][2], at[buf][3]}), bp[buf], zero16, 0, 0, 0);
            // next block's fragments (the next stage's first block at the end: complete since the previous barrier)
            if (blk + 1 < KB) load_ops(sb_cur, blk + 1, buf ^ 1);
            else load_ops(sb_nxt, 0, buf ^ 1);
            __builtin_amdgcn_sched_barrier(0);
            float cf[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) cf[r] = __builtin_fmaf(__int_as_float(D[r]), S[r], N[r]);     // = fl(float(isum) * (wScale * aScale))
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = acc[r] + cf[r];                                        // result +=, blocks ascending
            // pin the block's arithmetic here: the compiler would otherwise issue both blocks' MFMAs first (six result tiles: spills)
            asm volatile("" : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(acc[4]), "+v"(acc[5]), "+v"(acc[6]), "+v"(acc[7]),
                              "+v"(acc[8]), "+v"(acc[9]), "+v"(acc[10]), "+v"(acc[11]), "+v"(acc[12]), "+v"(acc[13]), "+v"(acc[14]), "+v"(acc[15]));
            __builtin_amdgcn_sched_barrier(0);
        }
        static_assert(KB % 2 == 0, "operand buffer parity restarts with every stage");
#if !defined(G4_SKIP_MEM) && !defined(G4_ASYNC_EXPERIMENT)
        scale_store(kf, fil);
#endif
#ifdef G4_ASYNC_EXPERIMENT
        // timing experiment (wrong results: the scale operands are never refreshed): only LDS-DMA in the loop, and the barrier
        // leaves the pieces issued during THIS stage in flight (they are needed one stage later)
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" :: "n"(NDMA) : "memory");
#else
        __syncthreads();                               // slot fil complete for every wavefront; slot cur free
#endif
        cur = nxt;
    }

    // ---- epilogue.  C layout: token = lane & 31 (column), weight row = (r & 3) + 8 * (r >> 2) + 4 * hi
    const int b = tok0 + tk;
    if constexpr (EPI == EPI_SWIGLU) {
        // the up-projection wavefronts hand their tile to the gate wavefront of the same rows and tokens through LDS
        constexpr int HALF = WR / 2;
        float4* xch = reinterpret_cast<float4*>(smem);
        const int pair = (wr % HALF) * 4 + wc;
        if (wr >= HALF) {
#pragma unroll
            for (int q = 0; q < 4; ++q) xch[(pair * 4 + q) * 64 + lane] = make_float4(acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]);
        }
        __syncthreads();
        if (wr < HALF && b < a.ntok) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 up = xch[(pair * 4 + q) * 64 + lane];
                const float uv[4] = {up.x, up.y, up.z, up.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int row = row0 + wr * 32 + e + 8 * q + 4 * hi;
                    float g = acc[4 * q + e];
                    g = g / (float)(1.0 + exp(-(double)g));
                    if (row < a.rows) a.out[(size_t)b * a.out_stride + row] = g * uv[e];
                }
            }
        }
    } else if (b < a.ntok) {
        float* o = a.out + (size_t)b * a.out_stride + row0 + wr * 32 + 4 * hi;
        const int rbase = row0 + wr * 32 + 4 * hi;
        float4 old[4];
        if (EPI == EPI_RESID) {
#pragma unroll
            for (int q = 0; q < 4; ++q) old[q] = *reinterpret_cast<const float4*>(rbase + 8 * q + 3 < a.rows ? o + 8 * q : a.out);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float4 v = {acc[4 * q] * a.out_scale, acc[4 * q + 1] * a.out_scale, acc[4 * q + 2] * a.out_scale, acc[4 * q + 3] * a.out_scale};
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
