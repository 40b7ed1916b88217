// gl3_prefill_gemm3t.h — the gate + up GEMM of the batched prefill (SwiGLU epilogue) on "tall" workgroup tiles, round 6.
// Same arithmetic, ring and barrier discipline as pf_gemm3_kernel (gl3_prefill_gemm3.h); what differs is the tile shape, chosen so that the launch is
// ONE round of workgroups.  The 128 x 128 tiling gives 896 workgroups for the 8B layer at 512 tokens on 512 slots: 1.75 rounds that cost two
// (SQ_BUSY_CYCLES 437 k per launch against 196 k wave-cycles per workgroup: 22 % of the launch is the half-empty second round).  14336 rows x 2
// matrices x 512 tokens = 14336 result tiles of 32 x 32 = exactly 56 per CU, so a workgroup here is
//     8 wavefronts = {gate, up} x 4 token fragments, each owning NFR row fragments x ONE token fragment    (NFR = 7: 224 rows per matrix x 128 tokens)
// one workgroup per CU, two wavefronts per SIMD, 64 x 4 = 256 workgroups for the 8B layer.  NFR = 4 .. 7 covers the other hidden sizes
// (8192 = 64 x 128 rows: NFR 4; 9728 = 60.8 x 160: NFR 5 -> 244 workgroups); the host picks the shape with the fewest tile-steps per CU.
//   * a K stage is KBT blocks (1 or 2): the weight side of a block is 17.5 KB at NFR = 7, a stage of two 51 KB, the ring 3 x 51 = 153 KB.  All 8
//     wavefronts of a CU share ONE barrier domain here, so both wavefronts of every SIMD sit in the barrier / scale-conversion phase together and the
//     matrix pipe idles through it (stamps: 429 + 228 of 2685 cycles per one-block stage): two blocks per stage halve those phases.  With KBT = 1 and
//     NFR odd the loop body is two stages (the result-tile register sets alternate with the global tile count).
//   * the token fragment's B operands are read once per block and serve NFR tiles; the A operands stream through two register sets, tile
//     q + 2's fragment is fetched when tile q + 1's MFMAs have been issued.
//   * gate and up tiles of an output element sit in different wavefronts: after the K loop each wavefront parks half of its tiles in LDS (the ring is
//     dead by then), takes the partner's other half, and applies SwiGLU to its share; 16-byte stores (a lane owns 4 consecutive hidden units).
// Slot image: Aq[blk][row fragment][half][32 rows][16 B] | At[blk][row][8 B] | Bq[blk][half][TOK][16 B] | Bs[blk][half][TOK][16 B].
// (A form with two 4-wavefront workgroups per CU and a ring of two slots was measured at 217 us against 200 for this one and removed.)
#pragma once
#include "gl3_prefill_gemm3.h"

__host__ __device__ constexpr int g3t_stage_bytes(int nfr, int kbt) { return kbt * (2 * (64 * nfr) * 16 + (64 * nfr) * 8 + 2 * 128 * 16 + 2 * 128 * 16); }
__host__ __device__ constexpr int g3t_lds_bytes(int nfr, int kbt) {
    return G3_RING * g3t_stage_bytes(nfr, kbt) > 16384 * nfr ? G3_RING * g3t_stage_bytes(nfr, kbt) : 16384 * nfr;
}

// QOUT: hb = silu(gate) * up leaves the kernel as the down projection's operand — int8 chunks (XQ3 layout), block scales and the scale-operand table
// entries of gl3_prefill_gemm3.h — instead of f32: a result tile's 32 rows ARE one 32-element activation block of a token (16 values in the lane,
// 16 in lane ^ 32), so the block maximum is a register maximum and one cross-lane exchange.  Saves the f32 round trip of hb and the quantise launch
// (pf_norm_quant_kernel<PQ_PLAIN>, 15 us per 8B layer at 512 tokens).  One rank only (under tensor parallelism the f32 hb is gathered first).
template <int NFR, int KBT, bool QOUT = false>
__global__ __launch_bounds__(512, 2) void pf_gemm3t_kernel(const GemmArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    constexpr int NW = 8, NT = 512, RPM = 32 * NFR, AROWS = 2 * RPM, TOK = 128, RING = G3_RING;
    constexpr int ABLK = 2 * AROWS * 16, BBLK = 2 * TOK * 16;                            // bytes of one block's int8 weights / activations (and scale operands)
    constexpr int OFF_AT = KBT * ABLK, OFF_BQ = OFF_AT + KBT * AROWS * 8, OFF_BS = OFF_BQ + KBT * BBLK, STAGE = g3t_stage_bytes(NFR, KBT);
    static_assert(STAGE == OFF_BS + KBT * BBLK, "stage layout");
    static_assert(KBT == 1 || KBT == 2, "blocks per stage");
    constexpr int NLA = KBT * 2 * NFR, NLB = KBT * BBLK / 1024, NPIECE = NLA + 2 * NLB;  // pieces per stage: weights, int8 activations, activation scale operands
    constexpr int NDMA = (NPIECE + NW - 1) / NW;
    constexpr int NSTEP = KBT * NFR;                                                     // tile steps per stage
    constexpr int SPB = (NSTEP & 1) ? 2 : 1;                                             // stages per loop body (register-set parity)
    constexpr int BSTEP = NSTEP - 2, NLATE = 2;                                          // first step that fetches an A fragment of the next stage
    constexpr int NSC = (KBT * AROWS + NT - 1) / NT;                                     // weight scale entries per thread and stage
    static_assert(NFR >= 3 && NDMA <= NSTEP && NSC <= 2, "shape");
    const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int tl = lane & 31, hi = lane >> 5;
    const int wm = wave >> 2, wc = wave & 3;           // matrix (0 gate, 1 up), token fragment
    const int ntt_g = a.ntt, per_xcd = (a.ntt * a.nrt + 7) >> 3;
    const int lin = blockIdx.x, J = (lin & 7) * per_xcd + (lin >> 3);
    if (J >= ntt_g * a.nrt) return;
    const int row0 = (J / ntt_g) * RPM;
    const int tok0 = (J % ntt_g) * TOK;
    const uint32_t strip_bytes = (uint32_t)a.ng * TILE_BYTES;
    const int nkb = (a.nb + KBT - 1) / KBT;            // stages that hold at least one real block
    const int nstrips = (a.rows + 15) >> 4;
    auto strip_off = [&](int lr) -> uint32_t { return (uint32_t)min(nstrips - 1, (row0 >> 4) + (lr >> 4)) * strip_bytes; };      // lr = row inside the matrix' RPM rows

    // ---- LDS-DMA pieces of this wavefront: uniform base pointer + 32-bit per-lane offset.  A weight piece = one 32-row fragment (rows of ONE
    // matrix) x both 16-byte halves of one block.  Per-stage source offset = (kf >> p_sh) * p_mul + (kf & ((1 << p_sh) - 1)) * p_odd: stage kf starts at
    // block kf * KBT = tile group (kf * KBT) >> 2, block (kf * KBT) & 3 inside it (KBT = 2: even, so both blocks sit in one tile group).
    const uint8_t* p_base[NDMA];
    uint32_t p_lane[NDMA], p_dst[NDMA], p_mul[NDMA], p_odd[NDMA];
    int p_sh[NDMA];
#pragma unroll
    for (int u = 0; u < NDMA; ++u) {
        int j = wave + NW * u;
        if (j >= NPIECE) j -= NPIECE;                  // surplus slot: re-load a piece (every wavefront issues exactly NDMA pieces per window)
        if (j < NLA) {
            const int blk = j / (2 * NFR), g = j % (2 * NFR), lr = (g % NFR) * 32 + tl;       // fragment g = matrix g / NFR, fragment g % NFR
            p_base[u] = g >= NFR ? a.w2 : a.w;
            p_lane[u] = strip_off(lr) + (hi ? 1152 : 128) + 16 * (lr & 15) + blk * 256;
            p_dst[u] = blk * ABLK + 1024 * g; p_sh[u] = KBT == 1 ? 2 : 1; p_mul[u] = TILE_BYTES; p_odd[u] = 256 * KBT;
        } else if (j < NLA + NLB) {
            const int jb = j - NLA, e = 64 * jb + lane, c = e / TOK, tk = (e % TOK) ^ (c & 1);      // c = blk * 2 + half; LDS slot p holds token p ^ half
            p_base[u] = a.XQ;
            p_lane[u] = ((uint32_t)c * (uint32_t)a.xp_tok + (uint32_t)(tok0 + tk)) * 16;
            p_dst[u] = OFF_BQ + 1024 * jb; p_sh[u] = 0; p_mul[u] = (uint32_t)(2 * KBT) * (uint32_t)a.xp_tok * 16; p_odd[u] = 0;
        } else {                                       // activation scale operands XP[block][half][token slot][16 B], the image of Bs
            const int jp = j - NLA - NLB, e = 64 * jp + lane, c = e / TOK;
            p_base[u] = a.XP;
            p_lane[u] = ((uint32_t)c * (uint32_t)a.xp_tok + (uint32_t)(tok0 + e % TOK)) * 16;
            p_dst[u] = OFF_BS + 1024 * jp; p_sh[u] = 0; p_mul[u] = (uint32_t)(2 * KBT) * (uint32_t)a.xp_tok * 16; p_odd[u] = 0;
        }
    }
    auto dma_one = [&](int kf, int slot, int u) {
        const uint32_t off = ((uint32_t)kf >> p_sh[u]) * p_mul[u] + ((uint32_t)kf & ((1u << p_sh[u]) - 1)) * p_odd[u];
        g2_dma16(p_base[u] + (p_lane[u] + off), smem + slot * STAGE + p_dst[u]);
    };
    // ---- weight scale operands: thread t owns the entries t, t + NT (< KBT * AROWS) of every stage; entry = (block e / AROWS, row e % AROWS)
    const uint8_t* s_wp[NSC];
    uint32_t r_ws[NSC];
#pragma unroll
    for (int k = 0; k < NSC; ++k) {
        const int e = t + NT * k, blk = e / AROWS, row = e % AROWS, lr = row % RPM;
        s_wp[k] = (row >= RPM ? a.w2 : a.w) + strip_off(lr) + 2 * (lr & 15) + blk * 32;
        r_ws[k] = 0;
    }
    auto scale_load = [&](int kf) {
        const uint32_t b0 = (uint32_t)kf * KBT;
#pragma unroll
        for (int k = 0; k < NSC; ++k) {
            const uint8_t* p = s_wp[k] + (size_t)(b0 >> 2) * TILE_BYTES + (b0 & 3) * 32;
            if (t + NT * k < KBT * AROWS) asm volatile("global_load_ushort %0, %1, off" : "=v"(r_ws[k]) : "v"(p) : "memory");
        }
    };
    auto scale_store = [&](int slot) {
        uint8_t* base = smem + slot * STAGE;
#pragma unroll
        for (int k = 0; k < NSC; ++k) {
            const int e = t + NT * k;
            if (e < KBT * AROWS) {
                const float wf = h2f((uint16_t)r_ws[k]);
                const float whi = __uint_as_float(__float_as_uint(wf) & 0xFFFF0000u), wlo = wf - whi;
                *reinterpret_cast<uint2*>(base + OFF_AT + e * 8) = make_uint2(g2_bf16_dup(whi), g2_bf16_dup(wlo));      // At[blk][row]: e = blk * AROWS + row
            }
        }
    };
    // every asm block that waits for the scale loads carries exactly their registers (the same variable twice would make the compiler copy a register
    // whose load is still in flight)
#define G3T_WAIT(text_, ...)                                                                                  \
    do {                                                                                                      \
        if constexpr (NSC == 1) asm volatile(text_ : "+v"(r_ws[0]) : __VA_ARGS__ : "memory");               \
        else asm volatile(text_ : "+v"(r_ws[0]), "+v"(r_ws[NSC - 1]) : __VA_ARGS__ : "memory");             \
    } while (0)

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

    v4i_t af[2], bf[2], bp[2];                         // bp = {s operand, -B s operand} of the lane's half
    v4s_t at[2];                                       // {w_hi, w_hi, w_lo, w_lo}
    v16i_t D[2];
    constexpr bool SSB = NFR >= 6;                     // register budget: the s tile single-buffered
    v16f2_t S[SSB ? 1 : 2], N[1];
    const uint32_t la = (uint32_t)(wm * NFR * 1024 + hi * 512 + tl * 16);
    const uint32_t lat = (uint32_t)(OFF_AT + (wm * RPM + tl) * 8);
    const uint32_t lb = (uint32_t)(OFF_BQ + (hi * TOK + ((wc * 32 + tl) ^ hi)) * 16), lp = (uint32_t)(OFF_BS + (hi * TOK + wc * 32 + tl) * 16);
    auto load_a = [&](const uint8_t* sb, int blk, int f, int set) {
        af[set] = *reinterpret_cast<const v4i_t*>(sb + blk * ABLK + la + f * 1024);
        at[set] = *reinterpret_cast<const v4s_t*>(sb + blk * (AROWS * 8) + lat + f * 256);
    };
    auto load_b = [&](const uint8_t* sb, int blk, int set) {
        bf[set] = *reinterpret_cast<const v4i_t*>(sb + blk * BBLK + lb);
        bp[set] = *reinterpret_cast<const v4i_t*>(sb + blk * BBLK + lp);
    };

    // ---- prologue: the ring full (stages 0 .. RING - 1), the weight scales of stage RING in flight, tile 0's MFMAs issued
#pragma unroll
    for (int s = 0; s < RING; ++s) {
        const int ks = min(s, nkb - 1);
#pragma unroll
        for (int u = 0; u < NDMA; ++u) dma_one(ks, s, u);
        scale_load(ks);
        G3T_WAIT("s_waitcnt vmcnt(0)", "n"(0));
        scale_store(s);
    }
    scale_load(min(RING, nkb - 1));
    G3T_WAIT("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier", "n"(0));
    load_b(smem, 0, 0);
    load_a(smem, 0, 0, 0);
    load_a(smem, 0, 1, 1);
    D[0] = __builtin_amdgcn_mfma_i32_32x32x32_i8(af[0], bf[0], cbias, 0, 0, 0);
    S[0] = __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(at[0], __builtin_bit_cast(v4s_t, v2i_t{bp[0][0], bp[0][1]}), zero16, 0, 0, 0);
    N[0] = __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(at[0], __builtin_bit_cast(v4s_t, v2i_t{bp[0][2], bp[0][3]}), zero16, 0, 0, 0);
    int cur = 0;
#ifdef G3_TIMING
    unsigned long long tm_bar = 0, tm_scale = 0, tm_t, tm_begin = __builtin_readcyclecounter();
#define G3T_T0() tm_t = __builtin_readcyclecounter()
#define G3T_T1(acc_) acc_ += __builtin_readcyclecounter() - tm_t
#else
#define G3T_T0() do {} while (0)
#define G3T_T1(acc_) do {} while (0)
#endif
    // one K stage; PS = index of the stage inside the loop body (all register-set indices are static)
    auto stage = [&](auto psc, int kb) {
        constexpr int PS = decltype(psc)::value;
        // window kb (behind this stage's barrier): stage kb + RING -> slot cur; the steps in front of the barrier still belong to window kb - 1:
        // stage kb + RING - 1 -> slot erl = (cur + RING - 1) % RING
        const int nxt = cur == RING - 1 ? 0 : cur + 1, erl = cur == 0 ? RING - 1 : cur - 1;
        const int kf_late = min(kb + RING, nkb - 1), kf_early = min(kb + RING - 1, nkb - 1);
        const uint8_t* sb_cur = smem + cur * STAGE;
        const uint8_t* sb_nxt = smem + nxt * STAGE;
        g2_static_for<0, NSTEP>([&](auto ic) {
            constexpr int i = decltype(ic)::value, q = PS * NSTEP + i, qb = q & 1, qn = (q + 1) & 1;      // q: tile count inside the body (register-set parity)
            constexpr int blk = i / NFR, f = i % NFR;
            constexpr int gb = PS * KBT + blk;                                     // block count inside the body: B register set gb & 1
            constexpr int bn = (gb + (f == NFR - 1 ? 1 : 0)) & 1;                  // B set of the next tile's block
            float cf[16];
            auto fma8 = [&](int r0) {
#pragma unroll
                for (int r = r0; r < r0 + 8; ++r) cf[r] = __builtin_fmaf(__int_as_float(D[qb][r]), S[SSB ? 0 : qb][r], N[0][r]);
                asm volatile("" : "+v"(cf[r0]), "+v"(cf[r0 + 1]), "+v"(cf[r0 + 2]), "+v"(cf[r0 + 3]), "+v"(cf[r0 + 4]), "+v"(cf[r0 + 5]), "+v"(cf[r0 + 6]), "+v"(cf[r0 + 7]));
            };
            // SSB: both fma halves first, then the s and -B s MFMAs back to back under the adds
            constexpr int sn = SSB ? 0 : qn;
            D[qn] = __builtin_amdgcn_mfma_i32_32x32x32_i8(af[qn], bf[bn], cbias, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            fma8(0);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (!SSB) {
                S[sn] = __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(at[qn], __builtin_bit_cast(v4s_t, v2i_t{bp[bn][0], bp[bn][1]}), zero16, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            fma8(8);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (SSB) {
                S[sn] = __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(at[qn], __builtin_bit_cast(v4s_t, v2i_t{bp[bn][0], bp[bn][1]}), zero16, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            N[0] = __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(at[qn], __builtin_bit_cast(v4s_t, v2i_t{bp[bn][2], bp[bn][3]}), zero16, 0, 0, 0);
            if constexpr (i == BSTEP) {
                // barrier kb (see gl3_prefill_gemm3.h): slot cur has been read for the last time, stage kb + 1 has landed in slot nxt
                __builtin_amdgcn_sched_barrier(0);
                G3T_T0();
                G3T_WAIT("s_waitcnt vmcnt(%[n]) lgkmcnt(0)\n\ts_barrier", [n] "n"(NDMA));
                G3T_T1(tm_bar);
                G3T_T0();
                scale_store(cur);
                scale_load(min(kb + RING + 1, nkb - 1));
                G3T_T1(tm_scale);
                __builtin_amdgcn_sched_barrier(0);
            }
            // the B operands of the next block, two steps before its first tile
            if constexpr (f == NFR - 2) {
                if constexpr (blk + 1 < KBT) load_b(sb_cur, blk + 1, (gb + 1) & 1);
                else load_b(sb_nxt, 0, (gb + 1) & 1);
            }
            // tile q + 2's A fragment into the set tile q has released
            if constexpr (i + 2 < NSTEP) load_a(sb_cur, (i + 2) / NFR, (i + 2) % NFR, qb);
            else load_a(sb_nxt, (i + 2 - NSTEP) / NFR, (i + 2 - NSTEP) % NFR, qb);
            if constexpr (i >= BSTEP) {
                if constexpr (i - BSTEP < NDMA) dma_one(kf_late, cur, i - BSTEP);
            } else {
                if constexpr (i + NLATE < NDMA) dma_one(kf_early, erl, i + NLATE);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[f][r] = acc[f][r] + cf[r];               // result +=, blocks ascending
            asm volatile("" : "+v"(acc[f][0]), "+v"(acc[f][1]), "+v"(acc[f][2]), "+v"(acc[f][3]), "+v"(acc[f][4]), "+v"(acc[f][5]), "+v"(acc[f][6]), "+v"(acc[f][7]),
                              "+v"(acc[f][8]), "+v"(acc[f][9]), "+v"(acc[f][10]), "+v"(acc[f][11]), "+v"(acc[f][12]), "+v"(acc[f][13]), "+v"(acc[f][14]), "+v"(acc[f][15]));
            __builtin_amdgcn_sched_barrier(0);
        });
        cur = nxt;
    };
    for (int kb = 0; kb < nkb; kb += SPB) {
        stage(std::integral_constant<int, 0>{}, kb);
        if constexpr (SPB == 2) { if (kb + 1 < nkb) stage(std::integral_constant<int, 1>{}, kb + 1); }
    }
#ifdef G3_TIMING
    if (lane == 0 && (J % 31) == 0)
        printf("g3t NFR %d KBT %d J %d wave %d stages %d: barrier %llu scale %llu total %llu cycles\n", NFR, KBT, J, wave, nkb, tm_bar, tm_scale, __builtin_readcyclecounter() - tm_begin);
#endif
    G3T_WAIT("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier", "n"(0));      // ring dead: LDS becomes the gate / up exchange
#undef G3T_WAIT
    // ---- epilogue.  C layout of a tile: token = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * hi.  Exchange X[token fragment][row fragment][r][lane]:
    // the gate wavefront parks its fragments >= FH, the up wavefront its fragments < FH; each applies SwiGLU to the fragments it kept.
    constexpr int FH = (NFR + 1) / 2;
    float* X = reinterpret_cast<float*>(smem);
    const int b = tok0 + wc * 32 + tl;
    auto park = [&](auto fc) {
        constexpr int f = decltype(fc)::value;
#pragma unroll
        for (int r = 0; r < 16; ++r) X[((wc * NFR + f) * 16 + r) * 64 + lane] = acc[f][r];
    };
    auto finish = [&](auto fc, auto gatec) {       // fragment f: this wavefront holds the gate tile (gatec) or the up tile; the other comes from X
        constexpr int f = decltype(fc)::value;
        constexpr bool have_gate = decltype(gatec)::value;
        if (!QOUT && b >= a.ntok) return;
        float o[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float other = X[((wc * NFR + f) * 16 + r) * 64 + lane];
            float g = have_gate ? acc[f][r] : other;
            const float up = have_gate ? other : acc[f][r];
            g = g / (float)(1.0 + exp(-(double)g));
            o[r] = g * up;
        }
        if constexpr (!QOUT) {
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) {
                const int row = row0 + f * 32 + 8 * q4 + 4 * hi;
                if (row < a.rows) *reinterpret_cast<float4*>(a.out + (size_t)b * a.out_stride + row) = make_float4(o[4 * q4], o[4 * q4 + 1], o[4 * q4 + 2], o[4 * q4 + 3]);
            }
        } else {
            // Q8_0 activation quantisation of the block (Q8_0FloatTensor.java:96-118; quantize_quad_pack's arithmetic): rows of a block = k index
            float amax = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) amax = fmaxf(amax, fabsf(o[r]));
            amax = fmaxf(amax, __shfl_xor(amax, 32, 64));
            const float qs = amax / 127.0f;
            const float ainv = qs != 0.f ? 1.0f / qs : 0.f;
            const int blk = (row0 + f * 32) >> 5;
            if (row0 + f * 32 < a.rows && b < a.ntok) {
#pragma unroll
                for (int q4 = 0; q4 < 4; ++q4) {           // rows 8 q4 + 4 hi .. + 3 = bytes (8 q4 + 4 hi) & 15 of chunk 2 blk + (q4 >> 1)
                    uint32_t packed = 0;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float sv = o[4 * q4 + e] * ainv;
                        packed |= (uint32_t)((int)(sv + copysignf(0.5f, sv)) & 0xFF) << (8 * e);
                    }
                    *reinterpret_cast<uint32_t*>(a.XQo + ((size_t)(2 * blk + (q4 >> 1)) * a.xp_tok + b) * 16 + ((8 * q4 + 4 * hi) & 15)) = packed;
                }
                if (hi == 0) {
                    const float qf = (float)(_Float16)qs;                              // float16ToFloat(floatToFloat16(qs))
                    const float ahi = __uint_as_float(__float_as_uint(qf) & 0xFFFF0000u), alo = qf - ahi;
                    auto pk = [](float h, float l) { return (__float_as_uint(h) >> 16) | (__float_as_uint(l) & 0xFFFF0000u); };
                    const uint32_t pr = pk(ahi, alo), q0 = pk(ahi * -8388608.f, alo * -8388608.f), q1 = pk(ahi * -4194304.f, alo * -4194304.f);
                    uint4* xp = reinterpret_cast<uint4*>(a.XPo);
                    xp[((size_t)blk * 2 + 0) * a.xp_tok + b] = make_uint4(pr, pr, q0, q0);
                    xp[((size_t)blk * 2 + 1) * a.xp_tok + b] = make_uint4(0u, 0u, q1, q1);
                    if (blk == (a.rows >> 5) - 1)                                        // ragged K of the consumer: zero operands for the padded blocks
                        for (int pb = blk + 1; pb < ((blk + 4) & ~3); ++pb) {
                            xp[((size_t)pb * 2 + 0) * a.xp_tok + b] = make_uint4(0u, 0u, 0u, 0u);
                            xp[((size_t)pb * 2 + 1) * a.xp_tok + b] = make_uint4(0u, 0u, 0u, 0u);
                        }
                }
            }
        }
    };
    // two straight-line branches (a shared loop with a run-time fragment index would move the accumulators to scratch)
    if (wm == 0) g2_static_for<FH, NFR>(park); else g2_static_for<0, FH>(park);
    __syncthreads();
    if (wm == 0) g2_static_for<0, FH>([&](auto fc) { finish(fc, std::true_type{}); });
    else g2_static_for<FH, NFR>([&](auto fc) { finish(fc, std::false_type{}); });
}
