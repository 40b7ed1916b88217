// gl3_prefill_gemm3.h — batched-prefill Q8_0 GEMM (> 64 tokens), round 6.  Arithmetic of gl3_prefill_gemm2.h, unchanged and bit-exact:
//     result += (float) isum * (wScale * aScale)                      (Q8_0FloatTensor.java:119, blocks ascending)
// as   D = 0x4B400000 + isum (int8 MFMA),  s = wScale * aScale and -B s (two bf16 MFMAs, exact),  p = fma(D, s, -B s),  result += p.
// What changed is how the operands reach the matrix pipe.  r4 / r5 stamps of pf_gemm2_kernel: arithmetic alone 123 cycles per (tile, block) and
// SIMD, the whole kernel ~265 — the rest was the stage structure: a __syncthreads() per K stage that carries vmcnt(0) and so waits for the LDS-DMA
// pieces the stage itself issued (issued -> landed is ~1-2 k cycles, about one stage), scale operands built by every thread inside the loop
// (two VGPR-returning loads, ~25 VALU, three LDS stores), and a branchy piece issue.  Here:
//   * THE BARRIER SITS IN THE MIDDLE OF A STAGE.  The last LDS read of ring slot k is issued long before the slot's last arithmetic (operands
//     are fetched one block ahead), so "slot k is free" and "slot k + 1 has landed" are both checked at the step whose operand refill first
//     reaches into slot k + 1 — not at the stage's end.  The operand prefetch across the stage boundary (MFMAs of the next stage's first tile
//     issued under the last tile's VALU work) survives, and with THREE ring slots the LDS-DMA lead is still 1.5-2 stages:
//         window k = [barrier k, barrier k + 1): pieces of stage k + 3 -> slot k % 3 (just freed);
//         barrier k + 1 waits s_waitcnt vmcnt(NDMA) = everything but window k's own pieces, i.e. stage k + 2 has landed.
//     Raw s_barrier + partial vmcnt from an asm block: nothing ever waits for a piece it has just issued.
//   * the lane-half constants of the -B s product (-2^23 in k slots 0-3, -2^22 in slots 4-7) ride on the ACTIVATION side: the side table
//     XP[block][half][token slot][16 B] = {P, P, Q, Q} with P = {bf16(a_hi), bf16(a_lo)}, Q = the same pair times -2^23 (half 0) / -2^22 (half 1), and
//     P = 0 in half 1, is written by pf_norm_quant_kernel next to XQ / XS — once per activation instead of once per row tile — and travels by
//     LDS-DMA like the int8 operands.  The weight side is then ONE 8-byte operand {w_hi, w_hi, w_lo, w_lo} per (row, block) for both scale MFMAs,
//     converted in the kernel (7 VALU + one 8-byte LDS store per entry; r4's form: 19 VALU + two 16-byte stores) by the threads of window k right
//     behind barrier k from a load issued a whole window earlier (inline asm, so that its wait is the barrier's).
//   * the int8 activations are stored chunk-major, XQ3[k / 16][token slot][16 B] (pf_norm_quant_kernel writes that layout when the side table is
//     on): an activation piece (64 tokens x 16 B of one k chunk) is 1 KB of CONSECUTIVE bytes.  In the row layout XQ[token][k] the same piece was
//     64 separate 16-byte accesses on 64 cache lines — 8 of a stage's 18 pieces went through the texture addresser one line at a time.
//   * piece issue is branch-free: per piece a uniform base pointer + a per-lane offset + one of three per-stage scalar offsets (s_cselect);
//     every wavefront issues exactly NDMA pieces per window (surplus slots re-load a piece: same bytes to the same place), which is what makes
//     the vmcnt count exact.
//   * generic wave grid WR x WC with NF x TF fragments per wavefront and KB blocks per stage, so that the 4096-row projections (wo / down: 256
//     workgroups of 64 rows x 128 tokens) run 8 one-tile wavefronts per workgroup with FOUR blocks per stage (a barrier per 4 steps, not 2) and
//     the qkv projection (6144 rows) 12 one-tile wavefronts on 96-row tiles = exactly one workgroup per CU.
// Ring slot image: Aq[blk][half][AROWS][16 B] | At[blk][AROWS][8 B] | Bq[blk][half][TOK][16 B] | Bs[blk][half][TOK][16 B].
#pragma once
#include "gl3_prefill_gemm2.h"

constexpr int G3_RING = 3;
__host__ __device__ constexpr int g3_stage_bytes(int arows, int tok, int kb) { return kb * (2 * arows * 16 + arows * 8 + 2 * tok * 16 + 2 * tok * 16); }

// first tile of a stage whose operand refill reads the NEXT stage's slot (tiles run block-major, row fragment, token fragment)
__host__ __device__ constexpr int g3_first_wrap_tile(int kb, int nf, int tf) {
    const int tpb = nf * tf;
    for (int it = (kb - 1) * tpb; it < kb * tpb; ++it)
        if (it % tf == tf - 1 || (it % tpb) / tf == nf - 1) return it;
    return kb * tpb - 1;
}

template <int EPI, int RF, int TF, int WR, int WC, int KB, int OCC>
__global__ __launch_bounds__(64 * WR * WC, (OCC * WR * WC + 3) / 4) void pf_gemm3_kernel(const GemmArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    constexpr int NM = (EPI == EPI_SWIGLU) ? 2 : 1;
    constexpr int NF = NM * RF;                        // 32-row fragments per wavefront
    constexpr int NW = WR * WC, NT = 64 * NW;
    constexpr int AROWS = NF * 32 * WR;                // weight rows staged per K stage
    constexpr int RPM = AROWS / NM;                    // output rows per matrix covered by this workgroup
    constexpr int TOK = WC * TF * 32;
    static_assert(NF <= 2 && TF <= 2, "accumulator budget");
    static_assert(NM == 1 || RF == 1, "SwiGLU: fragment index = matrix");
    static_assert(KB == 2 || KB == 4, "a stage is half a Q8T tile group or a whole one");
    static_assert(TOK == 128 || TOK == 64, "token tile");
    static_assert(NM == 1 || RPM % 64 == 0, "a weight piece (64 rows) belongs to one matrix");
    constexpr int OFF_AT = KB * 2 * AROWS * 16, OFF_BQ = OFF_AT + KB * AROWS * 8, OFF_BS = OFF_BQ + KB * 2 * TOK * 16;
    constexpr int STAGE = g3_stage_bytes(AROWS, TOK, KB);
    static_assert(STAGE == OFF_BS + KB * 2 * TOK * 16, "stage layout");
    constexpr int NLA = KB * 2 * AROWS / 64, NLB = KB * 2 * TOK / 64, NLP = KB * 2 * TOK / 64;      // LDS-DMA pieces per stage: weights, activations, activation scale operands
    constexpr int NPIECE = NLA + NLB + NLP;
    constexpr int NDMA = (NPIECE + NW - 1) / NW;                                       // ... per wavefront
    constexpr int NAT = AROWS * KB;                                                    // weight scale entries per stage (one thread each)
    static_assert(NAT <= NT, "one weight scale entry per thread");
    static_assert((KB * 2 * AROWS) % 64 == 0, "whole weight pieces");
    constexpr int TPB = NF * TF, NTILE = KB * TPB;
    static_assert(NTILE % 2 == 0, "result tiles alternate between two register sets");
    constexpr int FWT = g3_first_wrap_tile(KB, NF, TF), BSTEP = FWT - 1;              // the barrier sits in step BSTEP, before the refill behind tile FWT
    static_assert(BSTEP >= 0 && BSTEP < NTILE, "barrier step");
    constexpr int NLATE = NTILE - BSTEP;                                               // steps BSTEP .. NTILE - 1 follow the barrier inside the iteration
    constexpr int PPS = (NDMA + NTILE - 1) / NTILE;                                    // LDS-DMA pieces per step

    const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int tl = lane & 31, hi = lane >> 5;
    const int wr = wave / WC, wc = wave % WC;
    // XCD-aware tile mapping (as pf_gemm_kernel): the token tiles that share a weight row tile sit on ONE XCD
    const int ntt_g = a.ntt, per_xcd = (a.ntt * a.nrt + 7) >> 3;
    const int lin = blockIdx.x, J = (lin & 7) * per_xcd + (lin >> 3);
    if (J >= ntt_g * a.nrt) return;
    const int row0 = (J / ntt_g) * RPM;
    const int tok0 = (J % ntt_g) * TOK;
    const uint32_t strip_bytes = (uint32_t)a.ng * TILE_BYTES;
    const int nkb = (a.nb + KB - 1) / KB;              // stages that hold at least one real block
    const int nstrips = (a.rows + 15) >> 4;
    constexpr int SPM = RPM / 16;                      // strips per matrix in this workgroup (RPM = 96: six)
    auto strip_off = [&](int lrow) -> uint32_t {       // byte offset of local row lrow's strip inside its matrix
        const int sl = lrow >> 4;
        return (uint32_t)min(nstrips - 1, (row0 >> 4) + (NM == 2 ? sl % SPM : sl)) * strip_bytes;
    };

    // ---- LDS-DMA pieces of this wavefront: uniform {kind, base pointer, LDS destination} + per-lane source offset, stage-independent
    uint32_t p_lane[NDMA];
    uint32_t p_dst[NDMA];
    uint32_t p_mul[NDMA], p_odd[NDMA];                 // per-stage source offset = (kf >> p_sh) * p_mul + (kf & p_sh) * p_odd (p_sh = 0 | 1, uniform)
    int p_sh[NDMA];
    const uint8_t* p_base[NDMA];
#pragma unroll
    for (int u = 0; u < NDMA; ++u) {
        int j = wave + NW * u;
        if (j >= NPIECE) j -= NPIECE;                  // surplus slot: re-load a piece (keeps every wavefront at NDMA pieces per window)
        if (j < NLA) {                                 // weights: stage kf starts at block kf * KB = tile group (kf * KB) >> 2, block (kf * KB) & 3 inside it
            const int e = 64 * j + lane, c = e / AROWS, row = e % AROWS;       // c = blk * 2 + half
            p_lane[u] = strip_off(row) + ((c & 1) ? 1152 : 128) + 16 * ((c >> 1) * 16 + (row & 15));
            p_dst[u] = 1024 * j;
            p_base[u] = (NM == 2 && ((64 * j) % AROWS) / RPM) ? a.w2 : a.w;
            p_sh[u] = KB == 2 ? 1 : 0; p_mul[u] = TILE_BYTES; p_odd[u] = 512;
        } else if (j < NLA + NLB) {                    // int8 activations, chunk-major XQ3[k / 16][token slot][16 B]: a piece is 1 KB of consecutive bytes
            const int jb = j - NLA, e = 64 * jb + lane, c = e / TOK, tk = (e % TOK) ^ c;   // LDS slot p holds token p ^ c (bank spread)
            p_lane[u] = ((uint32_t)c * (uint32_t)a.xp_tok + (uint32_t)(tok0 + tk)) * 16;
            p_dst[u] = OFF_BQ + 1024 * jb;
            p_base[u] = a.XQ;
            p_sh[u] = 0; p_mul[u] = (uint32_t)(2 * KB) * (uint32_t)a.xp_tok * 16; p_odd[u] = 0;
        } else {                                       // activation scale operands XP[block][half][token slot][16 B], the image of Bs
            const int jp = j - NLA - NLB, e = 64 * jp + lane, c = e / TOK;      // c = blk * 2 + half
            p_lane[u] = ((uint32_t)c * (uint32_t)a.xp_tok + (uint32_t)(tok0 + e % TOK)) * 16;
            p_dst[u] = OFF_BS + 1024 * jp;
            p_base[u] = a.XP;
            p_sh[u] = 0; p_mul[u] = (uint32_t)(2 * KB) * (uint32_t)a.xp_tok * 16; p_odd[u] = 0;
        }
    }
    auto dma_one = [&](int kf, int slot, int u) {      // piece u of K stage kf -> ring slot; branch-free scalar address arithmetic
        const uint32_t off = ((uint32_t)kf >> p_sh[u]) * p_mul[u] + ((uint32_t)kf & (uint32_t)p_sh[u]) * p_odd[u];
        g2_dma16(p_base[u] + off + p_lane[u], smem + slot * STAGE + p_dst[u]);
    };
    // ---- weight scale operands: thread t < NAT owns entry (row t % AROWS, block t / AROWS) of every stage
    const int s_row = t % AROWS, s_blk = (t / AROWS) % KB;
    const uint8_t* s_wp = ((NM == 2 && s_row / RPM) ? a.w2 : a.w) + strip_off(s_row) + 2 * (s_row & 15);
    uint32_t r_ws = 0;
    auto scale_load = [&](int kf) {                    // inline asm: the wait for it is the window barrier's, not the compiler's
        const uint32_t ba = (uint32_t)kf * KB + s_blk;
        const uint8_t* p = s_wp + (size_t)(ba >> 2) * TILE_BYTES + (ba & 3) * 32;
        if (t < NAT) asm volatile("global_load_ushort %0, %1, off" : "=v"(r_ws) : "v"(p) : "memory");
    };
    auto scale_store = [&](int slot) {
        uint8_t* base = smem + slot * STAGE;
        if (t < NAT) {
            const float wf = h2f((uint16_t)r_ws);
            const float whi = __uint_as_float(__float_as_uint(wf) & 0xFFFF0000u), wlo = wf - whi;     // 8 + <= 3 significand bits
            *reinterpret_cast<uint2*>(base + OFF_AT + ((size_t)s_blk * AROWS + s_row) * 8) = make_uint2(g2_bf16_dup(whi), g2_bf16_dup(wlo));
        }
    };

    float acc[NF][TF][16];
#pragma unroll
    for (int i = 0; i < NF; ++i)
#pragma unroll
        for (int j = 0; j < TF; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    v16i_t cbias;
#pragma unroll
    for (int r = 0; r < 16; ++r) cbias[r] = 0x4B400000;
    asm volatile("" : "+v"(cbias));                    // keep the splat in VGPRs
    const v16f2_t zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

    // ---- operand fetch of one block from a ring slot; the per-lane LDS offsets are stage-independent
    v4i_t bf[TF], af[NF], bp[TF];                      // bp = {s operand (2 dwords), -B s operand (2 dwords)} of the lane's half
    v4s_t at[NF];                                      // {w_hi, w_hi, w_lo, w_lo}: the A operand of both scale MFMAs
    v16i_t D[2];
    v16f2_t S[2], N[1];
    uint32_t la[NF], lb[TF][KB], lp[TF];
#pragma unroll
    for (int f = 0; f < NF; ++f) la[f] = (uint32_t)(NM == 2 ? f * RPM + wr * 32 : wr * (32 * RF) + f * 32) + tl;      // local weight row
#pragma unroll
    for (int tf = 0; tf < TF; ++tf) {
        const int tk = wc * (32 * TF) + tf * 32 + tl;
        lp[tf] = (uint32_t)(OFF_BS + (hi * TOK + tk) * 16);
#pragma unroll
        for (int b = 0; b < KB; ++b) lb[tf][b] = (uint32_t)(OFF_BQ + ((b * 2 + hi) * TOK + (tk ^ (b * 2 + hi))) * 16);
    }
    auto load_a = [&](const uint8_t* sb, int blk, int f) {
        af[f] = *reinterpret_cast<const v4i_t*>(sb + (hi * AROWS + la[f]) * 16 + blk * (2 * AROWS * 16));
        at[f] = *reinterpret_cast<const v4s_t*>(sb + OFF_AT + la[f] * 8 + blk * (AROWS * 8));
    };
    auto load_b = [&](const uint8_t* sb, int blk, int tf) {
        bf[tf] = *reinterpret_cast<const v4i_t*>(sb + lb[tf][blk]);
        bp[tf] = *reinterpret_cast<const v4i_t*>(sb + lp[tf] + blk * (2 * TOK * 16));
    };
    auto issue_d = [&](int f, int tf, int buf) { D[buf] = __builtin_amdgcn_mfma_i32_32x32x32_i8(af[f], bf[tf], cbias, 0, 0, 0); };
    auto issue_s = [&](int f, int tf, int buf) {
        S[buf] = __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(at[f], __builtin_bit_cast(v4s_t, v2i_t{bp[tf][0], bp[tf][1]}), zero16, 0, 0, 0);
    };
    auto issue_n = [&](int f, int tf, int buf) {
        N[buf] = __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(at[f], __builtin_bit_cast(v4s_t, v2i_t{bp[tf][2], bp[tf][3]}), zero16, 0, 0, 0);
    };
    // After the three MFMAs of tile `it` (index inside its stage, whose ring slot is sb_t; sb_f = the slot of the stage after it) have been
    // issued: fragment registers that no later tile of the block reads are refilled with the next block's.
    auto refill = [&](auto itc, const uint8_t* sb_t, const uint8_t* sb_f) {
        constexpr int it = decltype(itc)::value, fi = (it % TPB) / TF, tfi = it % TF, bi = it / TPB;
        constexpr bool wrap = bi + 1 >= KB;
        const uint8_t* sb_o = wrap ? sb_f : sb_t;
        constexpr int bo = wrap ? 0 : bi + 1;
        if constexpr (tfi == TF - 1) load_a(sb_o, bo, fi);       // last token fragment of row fragment fi
        if constexpr (fi == NF - 1) load_b(sb_o, bo, tfi);       // last row fragment of token fragment tfi
    };

    // ---- prologue: stages 0 .. 2 complete in the ring, the weight scales of stage 3 in flight, the first tile's MFMAs issued
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
#pragma unroll
    for (int tf = 0; tf < TF; ++tf) load_b(smem, 0, tf);
#pragma unroll
    for (int f = 0; f < NF; ++f) load_a(smem, 0, f);
    issue_d(0, 0, 0);
    issue_s(0, 0, 0);
    issue_n(0, 0, 0);
    refill(std::integral_constant<int, 0>{}, smem, smem + STAGE);
    int cur = 0;                                       // ring slot of stage kb
#ifdef G3_TIMING
    unsigned long long tm_bar = 0, tm_scale = 0, tm_dma = 0, tm_t, tm_begin = __builtin_readcyclecounter();
#define G3_T0() tm_t = __builtin_readcyclecounter()
#define G3_T1(acc_) acc_ += __builtin_readcyclecounter() - tm_t
#else
#define G3_T0() do {} while (0)
#define G3_T1(acc_) do {} while (0)
#endif
    for (int kb = 0; kb < nkb; ++kb) {
        const int nxt = cur == G3_RING - 1 ? 0 : cur + 1, prv = nxt == G3_RING - 1 ? 0 : nxt + 1;
        const int kf_late = min(kb + 3, nkb - 1);      // window kb (after this iteration's barrier): stage kb + 3 -> slot cur
        const int kf_early = min(kb + 2, nkb - 1);     // window kb - 1 (before it): stage kb + 2 -> slot prv = (cur + 2) % 3
        const uint8_t* sb_cur = smem + cur * STAGE;
        const uint8_t* sb_nxt = smem + nxt * STAGE;    // past the last stage: a stale slot, results unused
        // step i finishes (tile, block) i of this stage and issues the MFMAs of step i + 1 (step 0 of the next stage at the end):
        //   [int8 MFMA i+1] [8 fma i] [s MFMA i+1] [8 fma i] [-B s MFMA i+1] (barrier) [operand refill] [LDS-DMA piece] [16 adds i]
        // The schedule is pinned (sched_barrier + value pins), see gl3_prefill_gemm2.h.
        g2_static_for<0, NTILE>([&](auto ic) {
            constexpr int i = decltype(ic)::value, f = (i % TPB) / TF, tf = i % TF;
            constexpr int in = (i + 1) % NTILE, fn = (in % TPB) / TF, tfn = in % TF;
            constexpr bool next_stage = i + 1 == NTILE;
            float cf[16];
            auto fma8 = [&](int r0) {
#pragma unroll
                for (int r = r0; r < r0 + 8; ++r) cf[r] = __builtin_fmaf(__int_as_float(D[i & 1][r]), S[i & 1][r], N[0][r]);   // = fl(float(isum) * (wScale * aScale))
                asm volatile("" : "+v"(cf[r0]), "+v"(cf[r0 + 1]), "+v"(cf[r0 + 2]), "+v"(cf[r0 + 3]), "+v"(cf[r0 + 4]), "+v"(cf[r0 + 5]), "+v"(cf[r0 + 6]), "+v"(cf[r0 + 7]));
            };
            issue_d(fn, tfn, (i + 1) & 1);
            __builtin_amdgcn_sched_barrier(0);
            fma8(0);
            __builtin_amdgcn_sched_barrier(0);
            issue_s(fn, tfn, (i + 1) & 1);
            __builtin_amdgcn_sched_barrier(0);
            fma8(8);
            __builtin_amdgcn_sched_barrier(0);
            issue_n(fn, tfn, 0);
            if constexpr (i == BSTEP) {
                // barrier kb: every read of slot cur has been issued (and is waited for here), so the slot is free; everything older than
                // window kb - 1's pieces has landed, i.e. stage kb + 1 is complete in slot nxt and the weight scales of stage kb + 3 are in r_ws
                __builtin_amdgcn_sched_barrier(0);
                G3_T0();
                asm volatile("s_waitcnt vmcnt(%1) lgkmcnt(0)\n\ts_barrier" : "+v"(r_ws) : "n"(NDMA) : "memory");
                G3_T1(tm_bar);
                G3_T0();
                scale_store(cur);
                scale_load(min(kb + 4, nkb - 1));
                G3_T1(tm_scale);
                __builtin_amdgcn_sched_barrier(0);
            }
            refill(std::integral_constant<int, in>{}, next_stage ? sb_nxt : sb_cur, sb_nxt);
#if defined(G3_TIMING) && G3_TIMING >= 2
            G3_T0();
#endif
            g2_static_for<0, PPS>([&](auto pc) {       // window-relative step i' = i - BSTEP behind the barrier, i + NLATE in front of it
                constexpr int pi = decltype(pc)::value;
                if constexpr (i >= BSTEP) {
                    if constexpr (PPS * (i - BSTEP) + pi < NDMA) dma_one(kf_late, cur, PPS * (i - BSTEP) + pi);
                } else {
                    if constexpr (PPS * (i + NLATE) + pi < NDMA) dma_one(kf_early, prv, PPS * (i + NLATE) + pi);
                }
            });
#if defined(G3_TIMING) && G3_TIMING >= 2
            G3_T1(tm_dma);
#endif
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[f][tf][r] = acc[f][tf][r] + cf[r];       // result +=, blocks ascending
            asm volatile("" : "+v"(acc[f][tf][0]), "+v"(acc[f][tf][1]), "+v"(acc[f][tf][2]), "+v"(acc[f][tf][3]), "+v"(acc[f][tf][4]), "+v"(acc[f][tf][5]),
                              "+v"(acc[f][tf][6]), "+v"(acc[f][tf][7]), "+v"(acc[f][tf][8]), "+v"(acc[f][tf][9]), "+v"(acc[f][tf][10]), "+v"(acc[f][tf][11]),
                              "+v"(acc[f][tf][12]), "+v"(acc[f][tf][13]), "+v"(acc[f][tf][14]), "+v"(acc[f][tf][15]));
            __builtin_amdgcn_sched_barrier(0);
        });
        cur = nxt;
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" : "+v"(r_ws) :: "memory");      // the loads hidden from the compiler's counters end here
#ifdef G3_TIMING
    if (lane == 0 && (J % 61) == 0)
        printf("g3 EPI %d NW %d KB %d J %d wave %d stages %d: barrier %llu scale %llu dma %llu total %llu cycles\n", EPI, NW, KB, J, wave, nkb, tm_bar, tm_scale, tm_dma,
               __builtin_readcyclecounter() - tm_begin);
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
                float g = acc[0][tf][r];
                g = g / (float)(1.0 + exp(-(double)g));
                a.out[(size_t)b * a.out_stride + row] = g * acc[NF - 1][tf][r];
            }
        } else {
#pragma unroll
            for (int f = 0; f < NF; ++f) {
                float* o = a.out + (size_t)b * a.out_stride + row0 + wr * (32 * RF) + f * 32 + 4 * hi;
                const int rbase = row0 + wr * (32 * RF) + f * 32 + 4 * hi;
                float4 old[4];
                if (EPI == EPI_RESID) {
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        old[q] = *reinterpret_cast<const float4*>(rbase + 8 * q + 3 < a.rows ? o + 8 * q : a.out);
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float4 v = {acc[f][tf][4 * q] * a.out_scale, acc[f][tf][4 * q + 1] * a.out_scale, acc[f][tf][4 * q + 2] * a.out_scale, acc[f][tf][4 * q + 3] * a.out_scale};
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
