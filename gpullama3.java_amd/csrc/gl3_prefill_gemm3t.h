// gl3_prefill_gemm3t.h — the gate + up GEMM of the batched prefill (SwiGLU epilogue) on "tall" workgroup tiles, round 6.
// Same arithmetic, ring and barrier discipline as pf_gemm3_kernel (gl3_prefill_gemm3.h); what differs is the tile shape, chosen so that the launch is
// ONE round of workgroups.  The 128 x 128 tiling gives 896 workgroups for the 8B layer at 512 tokens on 512 slots: 1.75 rounds that cost two
// (SQ_BUSY_CYCLES 437 k per launch against 196 k wave-cycles per workgroup: 22 % of the launch is the half-empty second round).  14336 rows x 2
// matrices x 512 tokens = 14336 result tiles of 32 x 32 = exactly 56 per CU, so a workgroup here is
//     8 wavefronts = {gate, up} x 4 token fragments, each owning NFR row fragments x ONE token fragment    (NFR = 7: 224 rows per matrix x 128 tokens)
// one workgroup per CU, two wavefronts per SIMD, 64 x 4 = 256 workgroups for the 8B layer.  NFR = 4 .. 7 covers the other hidden sizes
// (8192 = 64 x 128 rows: NFR 4; 9728 = 60.8 x 160: NFR 5 -> 244 workgroups); the host picks the shape with the fewest tile-steps per CU.
//   * a K stage is ONE block (the weight side of a stage is 28 KB at NFR = 7): ring = 3 x 33 KB.  NFR may be odd, so the loop body is two
//     stages (the result-tile register sets alternate with the global tile count).
//   * the token fragment's B operands are read once per block and serve NFR tiles; the A operands stream through two register sets, tile
//     q + 2's fragment is fetched when tile q + 1's MFMAs have been issued.
//   * gate and up tiles of an output element sit in different wavefronts: after the K loop each wavefront parks half of its tiles in LDS (the ring is
//     dead by then), takes the partner's other half, and applies SwiGLU to its share; 16-byte stores (a lane owns 4 consecutive hidden units).
#pragma once
#include "gl3_prefill_gemm3.h"

__host__ __device__ constexpr int g3t_stage_bytes(int nfr) { return 2 * (64 * nfr) * 16 * 2 + 2 * 128 * 16 + 128 * 8; }
__host__ __device__ constexpr int g3t_lds_bytes(int nfr) { return G3_RING * g3t_stage_bytes(nfr) > 16384 * nfr ? G3_RING * g3t_stage_bytes(nfr) : 16384 * nfr; }

template <int NFR>
__global__ __launch_bounds__(512, 2) void pf_gemm3t_kernel(const GemmArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    constexpr int NW = 8, RPM = 32 * NFR, AROWS = 2 * RPM, TOK = 128;
    constexpr int OFF_AT = 2 * AROWS * 16, OFF_BQ = 2 * OFF_AT, OFF_BS = OFF_BQ + 2 * TOK * 16, STAGE = g3t_stage_bytes(NFR);
    static_assert(STAGE == OFF_BS + TOK * 8, "stage layout");
    constexpr int NLA = 2 * AROWS / 64, NLB = 2 * TOK / 64, NPIECE = NLA + NLB + 1;      // pieces per stage: weights, int8 activations, activation scale operands
    constexpr int NDMA = (NPIECE + NW - 1) / NW;
    constexpr int BSTEP = NFR - 2, NLATE = NFR - BSTEP;                                  // first step that fetches an A fragment of the next stage
    static_assert(NFR >= 3 && NDMA <= NFR && AROWS <= 512, "shape");
    const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int tl = lane & 31, hi = lane >> 5;
    const int wm = wave >> 2, wc = wave & 3;           // matrix (0 gate, 1 up), token fragment
    const int ntt_g = a.ntt, per_xcd = (a.ntt * a.nrt + 7) >> 3;
    const int lin = blockIdx.x, J = (lin & 7) * per_xcd + (lin >> 3);
    if (J >= ntt_g * a.nrt) return;
    const int row0 = (J / ntt_g) * RPM;
    const int tok0 = (J % ntt_g) * TOK;
    const uint32_t strip_bytes = (uint32_t)a.ng * TILE_BYTES;
    const int nkb = a.nb;                              // one block per stage
    const int nstrips = (a.rows + 15) >> 4;
    auto strip_off = [&](int lr) -> uint32_t { return (uint32_t)min(nstrips - 1, (row0 >> 4) + (lr >> 4)) * strip_bytes; };      // lr = row inside the matrix' RPM rows

    // ---- LDS-DMA pieces of this wavefront (per-lane 64-bit source: a 64-row weight piece may straddle the gate / up boundary)
    const uint8_t* p_src[NDMA];
    uint32_t p_dst[NDMA], p_mul[NDMA], p_odd[NDMA];
    int p_sh[NDMA];                                    // per-stage offset = (kf >> p_sh) * p_mul + (kf & ((1 << p_sh) - 1)) * p_odd
#pragma unroll
    for (int u = 0; u < NDMA; ++u) {
        int j = wave + NW * u;
        if (j >= NPIECE) j -= NPIECE;                  // surplus slot: re-load a piece (every wavefront issues exactly NDMA pieces per window)
        if (j < NLA) {
            const int e = 64 * j + lane, c = e / AROWS, row = e % AROWS, lr = row % RPM;       // c = half
            p_src[u] = (row >= RPM ? a.w2 : a.w) + strip_off(lr) + (c ? 1152 : 128) + 16 * (lr & 15);
            p_dst[u] = 1024 * j; p_sh[u] = 2; p_mul[u] = TILE_BYTES; p_odd[u] = 256;
        } else if (j < NLA + NLB) {
            const int jb = j - NLA, e = 64 * jb + lane, c = e / TOK, tk = (e % TOK) ^ c;       // LDS slot p holds token p ^ c (bank spread)
            p_src[u] = a.XQ + ((size_t)c * a.xp_tok + (size_t)(tok0 + tk)) * 16;
            p_dst[u] = OFF_BQ + 1024 * jb; p_sh[u] = 0; p_mul[u] = 2u * (uint32_t)a.xp_tok * 16; p_odd[u] = 0;
        } else {
            p_src[u] = a.XP + (size_t)tok0 * 8 + 16 * lane;
            p_dst[u] = OFF_BS; p_sh[u] = 0; p_mul[u] = (uint32_t)a.xp_tok * 8; p_odd[u] = 0;
        }
    }
    auto dma_one = [&](int kf, int slot, int u) {
        const uint32_t off = ((uint32_t)kf >> p_sh[u]) * p_mul[u] + ((uint32_t)kf & ((1u << p_sh[u]) - 1)) * p_odd[u];
        g2_dma16(p_src[u] + off, smem + slot * STAGE + p_dst[u]);
    };
    // ---- weight scale operands: thread t < AROWS owns row t's entry of every stage
    const int s_lr = t % RPM;
    const uint8_t* s_wp = (t >= RPM ? a.w2 : a.w) + strip_off(s_lr) + 2 * (s_lr & 15);
    uint32_t r_ws = 0;
    auto scale_load = [&](int kf) {
        const uint8_t* p = s_wp + (size_t)(kf >> 2) * TILE_BYTES + (kf & 3) * 32;
        if (t < AROWS) asm volatile("global_load_ushort %0, %1, off" : "=v"(r_ws) : "v"(p) : "memory");
    };
    auto scale_store = [&](int slot) {
        uint8_t* base = smem + slot * STAGE;
        if (t < AROWS) {
            const float wf = h2f((uint16_t)r_ws);
            const float whi = __uint_as_float(__float_as_uint(wf) & 0xFFFF0000u), wlo = wf - whi;
            const v4i_t lo = {(int)g2_bf16_dup(whi), (int)g2_bf16_dup(wlo), (int)g2_bf16_dup(whi * -8388608.f), (int)g2_bf16_dup(wlo * -8388608.f)};
            const v4i_t hh = {0, 0, (int)g2_bf16_dup(whi * -4194304.f), (int)g2_bf16_dup(wlo * -4194304.f)};
            *reinterpret_cast<v4i_t*>(base + OFF_AT + (size_t)t * 16) = lo;
            *reinterpret_cast<v4i_t*>(base + OFF_AT + ((size_t)AROWS + t) * 16) = hh;
        }
    };

    float acc[NFR][16];
#pragma unroll
    for (int i = 0; i < NFR; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    v16i_t cbias;
#pragma unroll
    for (int r = 0; r < 16; ++r) cbias[r] = 0x4B400000;
    asm volatile("" : "+v"(cbias));
    const v16f2_t zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

    v4i_t af[2], at[2], bf[2];
    v4s_t bp[2];
    v16i_t D[2];
    constexpr bool SSB = NFR >= 6;
    v16f2_t S[SSB ? 1 : 2], N[1];
    const uint32_t la = (uint32_t)((hi * AROWS + wm * RPM + tl) * 16);
    const uint32_t lb = (uint32_t)(OFF_BQ + (hi * TOK + ((wc * 32 + tl) ^ hi)) * 16), lp = (uint32_t)(OFF_BS + (wc * 32 + tl) * 8);
    auto load_a = [&](const uint8_t* sb, int f, int set) {
        af[set] = *reinterpret_cast<const v4i_t*>(sb + la + f * 512);
        at[set] = *reinterpret_cast<const v4i_t*>(sb + OFF_AT + la + f * 512);
    };
    auto load_b = [&](const uint8_t* sb, int set) {
        bf[set] = *reinterpret_cast<const v4i_t*>(sb + lb);
        bp[set] = *reinterpret_cast<const v4s_t*>(sb + lp);
    };

    // ---- prologue: stages 0 .. 2 complete in the ring, the weight scales of stage 3 in flight, tile 0's MFMAs issued
#pragma unroll
    for (int s = 0; s < G3_RING; ++s) {
        const int ks = min(s, nkb - 1);
#pragma unroll
        for (int u = 0; u < NDMA; ++u) dma_one(ks, s, u);
        scale_load(ks);
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(r_ws) :: "memory");
        scale_store(s);
    }
    scale_load(min(G3_RING, nkb - 1));
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" : "+v"(r_ws) :: "memory");
    load_b(smem, 0);
    load_a(smem, 0, 0);
    load_a(smem, 1, 1);
    D[0] = __builtin_amdgcn_mfma_i32_32x32x32_i8(af[0], bf[0], cbias, 0, 0, 0);
    S[0] = __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(__builtin_bit_cast(v4s_t, v2i_t{at[0][0], at[0][1]}), bp[0], zero16, 0, 0, 0);
    N[0] = __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(__builtin_bit_cast(v4s_t, v2i_t{at[0][2], at[0][3]}), bp[0], zero16, 0, 0, 0);
    int cur = 0;
    // one K stage; PS = parity of the stage inside the two-stage loop body (all register-set indices are static)
    auto stage = [&](auto psc, int kb) {
        constexpr int PS = decltype(psc)::value;
        const int nxt = cur == G3_RING - 1 ? 0 : cur + 1, prv = nxt == G3_RING - 1 ? 0 : nxt + 1;
        const int kf_late = min(kb + 3, nkb - 1), kf_early = min(kb + 2, nkb - 1);
        const uint8_t* sb_cur = smem + cur * STAGE;
        const uint8_t* sb_nxt = smem + nxt * STAGE;
        g2_static_for<0, NFR>([&](auto ic) {
            constexpr int i = decltype(ic)::value, q = PS * NFR + i, qb = q & 1, qn = (q + 1) & 1;
            constexpr bool last = i + 1 == NFR;
            constexpr int bn = last ? (PS ^ 1) : PS;                   // B register set of the next tile's stage
            float cf[16];
            auto fma8 = [&](int r0) {
#pragma unroll
                for (int r = r0; r < r0 + 8; ++r) cf[r] = __builtin_fmaf(__int_as_float(D[qb][r]), S[SSB ? 0 : qb][r], N[0][r]);
                asm volatile("" : "+v"(cf[r0]), "+v"(cf[r0 + 1]), "+v"(cf[r0 + 2]), "+v"(cf[r0 + 3]), "+v"(cf[r0 + 4]), "+v"(cf[r0 + 5]), "+v"(cf[r0 + 6]), "+v"(cf[r0 + 7]));
            };
            // SSB (NFR >= 6, register budget): the s tile single-buffered — both fma halves first, then the s and -B s MFMAs back to back under the adds
            constexpr int sb = SSB ? 0 : qb, sn = SSB ? 0 : qn;
            D[qn] = __builtin_amdgcn_mfma_i32_32x32x32_i8(af[qn], bf[bn], cbias, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            fma8(0);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (!SSB) {
                S[sn] = __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(__builtin_bit_cast(v4s_t, v2i_t{at[qn][0], at[qn][1]}), bp[bn], zero16, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            fma8(8);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (SSB) {
                S[sn] = __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(__builtin_bit_cast(v4s_t, v2i_t{at[qn][0], at[qn][1]}), bp[bn], zero16, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            N[0] = __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(__builtin_bit_cast(v4s_t, v2i_t{at[qn][2], at[qn][3]}), bp[bn], zero16, 0, 0, 0);
            if constexpr (i == BSTEP) {
                // barrier kb (see gl3_prefill_gemm3.h): slot cur has been read for the last time, stage kb + 1 has landed in slot nxt
                __builtin_amdgcn_sched_barrier(0);
                asm volatile("s_waitcnt vmcnt(%1) lgkmcnt(0)\n\ts_barrier" : "+v"(r_ws) : "n"(NDMA) : "memory");
                scale_store(cur);
                scale_load(min(kb + 4, nkb - 1));
                load_b(sb_nxt, PS ^ 1);
                __builtin_amdgcn_sched_barrier(0);
            }
            if constexpr (i + 2 < NFR) load_a(sb_cur, i + 2, qb);      // tile q + 2's fragment into the set tile q has released
            else load_a(sb_nxt, i + 2 - NFR, qb);
            if constexpr (i >= BSTEP) {
                if constexpr (i - BSTEP < NDMA) dma_one(kf_late, cur, i - BSTEP);
            } else {
                if constexpr (i + NLATE < NDMA) dma_one(kf_early, prv, i + NLATE);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][r] = acc[i][r] + cf[r];               // result +=, blocks ascending
            asm volatile("" : "+v"(acc[i][0]), "+v"(acc[i][1]), "+v"(acc[i][2]), "+v"(acc[i][3]), "+v"(acc[i][4]), "+v"(acc[i][5]), "+v"(acc[i][6]), "+v"(acc[i][7]),
                              "+v"(acc[i][8]), "+v"(acc[i][9]), "+v"(acc[i][10]), "+v"(acc[i][11]), "+v"(acc[i][12]), "+v"(acc[i][13]), "+v"(acc[i][14]), "+v"(acc[i][15]));
            __builtin_amdgcn_sched_barrier(0);
        });
        cur = nxt;
    };
    for (int kb = 0; kb < nkb; kb += 2) {
        stage(std::integral_constant<int, 0>{}, kb);
        if (kb + 1 < nkb) stage(std::integral_constant<int, 1>{}, kb + 1);
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" : "+v"(r_ws) :: "memory");      // ring dead: LDS becomes the gate / up exchange
    // ---- epilogue.  C layout of a tile: token = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * hi.  Exchange X[token fragment][row fragment][r][lane]:
    // the gate wavefront parks its fragments >= FH, the up wavefront its fragments < FH; each applies SwiGLU to the fragments it kept.
    constexpr int FH = (NFR + 1) / 2;
    float* X = reinterpret_cast<float*>(smem);
    g2_static_for<0, NFR>([&](auto fc) {
        constexpr int f = decltype(fc)::value;
        const bool give = wm == 0 ? f >= FH : f < FH;
        if (give) {
#pragma unroll
            for (int r = 0; r < 16; ++r) X[((wc * NFR + f) * 16 + r) * 64 + lane] = acc[f][r];
        }
    });
    __syncthreads();
    const int b = tok0 + wc * 32 + tl;
    g2_static_for<0, NFR>([&](auto fc) {
        constexpr int f = decltype(fc)::value;
        const bool mine = wm == 0 ? f < FH : f >= FH;
        if (mine && b < a.ntok) {
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) {
                const int row = row0 + f * 32 + 8 * q4 + 4 * hi;
                if (row >= a.rows) continue;
                float o[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int r = 4 * q4 + e;
                    const float other = X[((wc * NFR + f) * 16 + r) * 64 + lane];
                    float g = wm == 0 ? acc[f][r] : other;
                    const float up = wm == 0 ? other : acc[f][r];
                    g = g / (float)(1.0 + exp(-(double)g));
                    o[e] = g * up;
                }
                *reinterpret_cast<float4*>(a.out + (size_t)b * a.out_stride + row) = make_float4(o[0], o[1], o[2], o[3]);
            }
        }
    });
}
