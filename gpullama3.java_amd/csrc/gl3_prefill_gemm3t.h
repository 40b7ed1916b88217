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
// WCN = token fragments per workgroup.  WCN 4: the 8-wavefront form above (ring of 3, partial vmcnt).  WCN 2: 4 wavefronts = {gate, up} x 2 token
// fragments, 64 tokens, a ring of TWO slots (61 KB at NFR = 7), so that two INDEPENDENT workgroups share a CU as in the 128 x 128 tiling: with all 8
// wavefronts of a CU in one barrier domain, both wavefronts of every SIMD sit in the barrier / scale-conversion phase at the same time and the matrix pipe
// idles; two domains fill each other's gaps.  With two slots the pieces of window k (stage k + 2 -> the slot barrier k freed) must have landed by barrier
// k + 1, one whole stage later: they are issued in a burst right behind the barrier and the wait is a plain vmcnt(0).
#pragma once
#include "gl3_prefill_gemm3.h"

__host__ __device__ constexpr int g3t_stage_bytes(int nfr, int wcn) { return 2 * (64 * nfr) * 16 + (64 * nfr) * 8 + 2 * (32 * wcn) * 16 + 2 * (32 * wcn) * 16; }
__host__ __device__ constexpr int g3t_ring(int wcn) { return wcn == 4 ? 3 : 2; }
__host__ __device__ constexpr int g3t_lds_bytes(int nfr, int wcn) {
    return g3t_ring(wcn) * g3t_stage_bytes(nfr, wcn) > 4096 * wcn * nfr ? g3t_ring(wcn) * g3t_stage_bytes(nfr, wcn) : 4096 * wcn * nfr;
}

template <int NFR, int WCN>
__global__ __launch_bounds__(128 * WCN, 2) void pf_gemm3t_kernel(const GemmArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    constexpr int NW = 2 * WCN, NT = 64 * NW, RPM = 32 * NFR, AROWS = 2 * RPM, TOK = 32 * WCN, RING = g3t_ring(WCN);
    constexpr int OFF_AT = 2 * AROWS * 16, OFF_BQ = OFF_AT + AROWS * 8, OFF_BS = OFF_BQ + 2 * TOK * 16, STAGE = g3t_stage_bytes(NFR, WCN);
    static_assert(STAGE == OFF_BS + 2 * TOK * 16, "stage layout");
    constexpr int NLA = 2 * AROWS / 64, NLB = 2 * TOK / 64, NPIECE = NLA + 2 * NLB;      // pieces per stage: weights, int8 activations, activation scale operands
    constexpr int NDMA = (NPIECE + NW - 1) / NW;
    constexpr int BSTEP = NFR - 2, NLATE = NFR - BSTEP;                                  // first step that fetches an A fragment of the next stage
    constexpr int PPS = RING == 3 ? 1 : 2;                                               // pieces per step behind the barrier (ring of 2: a burst)
    constexpr int NSC = (AROWS + NT - 1) / NT;                                           // weight scale entries per thread and stage
    static_assert(WCN == 2 || WCN == 4, "token fragments per workgroup");
    static_assert(NFR >= 3 && NDMA <= PPS * NFR, "shape");
    const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int tl = lane & 31, hi = lane >> 5;
    const int wm = wave / WCN, wc = wave % WCN;        // matrix (0 gate, 1 up), token fragment
    const int ntt_g = a.ntt, per_xcd = (a.ntt * a.nrt + 7) >> 3;
    const int lin = blockIdx.x, J = (lin & 7) * per_xcd + (lin >> 3);
    if (J >= ntt_g * a.nrt) return;
    const int row0 = (J / ntt_g) * RPM;
    const int tok0 = (J % ntt_g) * TOK;
    const uint32_t strip_bytes = (uint32_t)a.ng * TILE_BYTES;
    const int nkb = a.nb;                              // one block per stage
    const int nstrips = (a.rows + 15) >> 4;
    auto strip_off = [&](int lr) -> uint32_t { return (uint32_t)min(nstrips - 1, (row0 >> 4) + (lr >> 4)) * strip_bytes; };      // lr = row inside the matrix' RPM rows

    // ---- LDS-DMA pieces of this wavefront: uniform base pointer + 32-bit per-lane offset.  A weight piece = one 32-row fragment (rows of ONE
    // matrix) x both 16-byte halves of the block: slot image Aq[row fragment][half][32 rows][16 B] | At[row][8 B] | Bq[half][TOK][16 B] | Bs[half][TOK][16 B]
    const uint8_t* p_base[NDMA];
    uint32_t p_lane[NDMA], p_dst[NDMA], p_mul[NDMA], p_odd[NDMA];
    int p_sh[NDMA];                                    // per-stage offset = (kf >> p_sh) * p_mul + (kf & ((1 << p_sh) - 1)) * p_odd
#pragma unroll
    for (int u = 0; u < NDMA; ++u) {
        int j = wave + NW * u;
        if (j >= NPIECE) j -= NPIECE;                  // surplus slot: re-load a piece (every wavefront issues exactly NDMA pieces per window)
        if (j < NLA) {
            const int lr = (j % NFR) * 32 + tl;        // row inside the matrix' RPM rows; fragment j = matrix j / NFR, fragment j % NFR
            p_base[u] = j >= NFR ? a.w2 : a.w;
            p_lane[u] = strip_off(lr) + (hi ? 1152 : 128) + 16 * (lr & 15);
            p_dst[u] = 1024 * j; p_sh[u] = 2; p_mul[u] = TILE_BYTES; p_odd[u] = 256;
        } else if (j < NLA + NLB) {
            const int jb = j - NLA, e = 64 * jb + lane, c = e / TOK, tk = (e % TOK) ^ c;       // LDS slot p holds token p ^ c (bank spread)
            p_base[u] = a.XQ;
            p_lane[u] = ((uint32_t)c * (uint32_t)a.xp_tok + (uint32_t)(tok0 + tk)) * 16;
            p_dst[u] = OFF_BQ + 1024 * jb; p_sh[u] = 0; p_mul[u] = 2u * (uint32_t)a.xp_tok * 16; p_odd[u] = 0;
        } else {                                       // activation scale operands XP[block][half][token slot][16 B], the image of Bs
            const int jp = j - NLA - NLB, e = 64 * jp + lane, c = e / TOK;
            p_base[u] = a.XP;
            p_lane[u] = ((uint32_t)c * (uint32_t)a.xp_tok + (uint32_t)(tok0 + e % TOK)) * 16;
            p_dst[u] = OFF_BS + 1024 * jp; p_sh[u] = 0; p_mul[u] = 2u * (uint32_t)a.xp_tok * 16; p_odd[u] = 0;
        }
    }
    auto dma_one = [&](int kf, int slot, int u) {
        const uint32_t off = ((uint32_t)kf >> p_sh[u]) * p_mul[u] + ((uint32_t)kf & ((1u << p_sh[u]) - 1)) * p_odd[u];
        g2_dma16(p_base[u] + (p_lane[u] + off), smem + slot * STAGE + p_dst[u]);
    };
    // ---- weight scale operands: thread t owns the entries of rows t, t + NT, .. (< AROWS) of every stage
    const uint8_t* s_wp[NSC];
    uint32_t r_ws[NSC];
#pragma unroll
    for (int k = 0; k < NSC; ++k) {
        const int row = t + NT * k, lr = row % RPM;
        s_wp[k] = (row >= RPM ? a.w2 : a.w) + strip_off(lr) + 2 * (lr & 15);
        r_ws[k] = 0;
    }
    auto scale_load = [&](int kf) {
#pragma unroll
        for (int k = 0; k < NSC; ++k) {
            const uint8_t* p = s_wp[k] + (size_t)(kf >> 2) * TILE_BYTES + (kf & 3) * 32;
            if (t + NT * k < AROWS) asm volatile("global_load_ushort %0, %1, off" : "=v"(r_ws[k]) : "v"(p) : "memory");
        }
    };
    auto scale_store = [&](int slot) {
        uint8_t* base = smem + slot * STAGE;
#pragma unroll
        for (int k = 0; k < NSC; ++k) {
            const int row = t + NT * k;
            if (row < AROWS) {
                const float wf = h2f((uint16_t)r_ws[k]);
                const float whi = __uint_as_float(__float_as_uint(wf) & 0xFFFF0000u), wlo = wf - whi;
                *reinterpret_cast<uint2*>(base + OFF_AT + row * 8) = make_uint2(g2_bf16_dup(whi), g2_bf16_dup(wlo));
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
    static_assert(NSC <= 2, "scale entries per thread");

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
    constexpr bool SSB = NFR >= 6;
    v16f2_t S[SSB ? 1 : 2], N[1];
    const uint32_t la = (uint32_t)(wm * NFR * 1024 + hi * 512 + tl * 16);
    const uint32_t lat = (uint32_t)(OFF_AT + (wm * RPM + tl) * 8);
    const uint32_t lb = (uint32_t)(OFF_BQ + (hi * TOK + ((wc * 32 + tl) ^ hi)) * 16), lp = (uint32_t)(OFF_BS + (hi * TOK + wc * 32 + tl) * 16);
    auto load_a = [&](const uint8_t* sb, int f, int set) {
        af[set] = *reinterpret_cast<const v4i_t*>(sb + la + f * 1024);
        at[set] = *reinterpret_cast<const v4s_t*>(sb + lat + f * 256);
    };
    auto load_b = [&](const uint8_t* sb, int set) {
        bf[set] = *reinterpret_cast<const v4i_t*>(sb + lb);
        bp[set] = *reinterpret_cast<const v4i_t*>(sb + lp);
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
    load_b(smem, 0);
    load_a(smem, 0, 0);
    load_a(smem, 1, 1);
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
    // one K stage; PS = parity of the stage inside the two-stage loop body (all register-set indices are static)
    auto stage = [&](auto psc, int kb) {
        constexpr int PS = decltype(psc)::value;
        // window kb (behind this stage's barrier): stage kb + RING -> slot cur; the steps in front of the barrier still belong to window kb - 1:
        // stage kb + RING - 1 -> slot erl = (cur + RING - 1) % RING
        const int nxt = cur == RING - 1 ? 0 : cur + 1, erl = cur == 0 ? RING - 1 : cur - 1;
        const int kf_late = min(kb + RING, nkb - 1), kf_early = min(kb + RING - 1, nkb - 1);
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
                if constexpr (RING == 3) G3T_WAIT("s_waitcnt vmcnt(%[n]) lgkmcnt(0)\n\ts_barrier", [n] "n"(NDMA));
                else G3T_WAIT("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier", "n"(0));
                G3T_T1(tm_bar);
                G3T_T0();
                scale_store(cur);
                scale_load(min(kb + RING + 1, nkb - 1));
                G3T_T1(tm_scale);
                load_b(sb_nxt, PS ^ 1);
                __builtin_amdgcn_sched_barrier(0);
            }
            if constexpr (i + 2 < NFR) load_a(sb_cur, i + 2, qb);      // tile q + 2's fragment into the set tile q has released
            else load_a(sb_nxt, i + 2 - NFR, qb);
            g2_static_for<0, PPS>([&](auto pc) {
                constexpr int pi = decltype(pc)::value;
                if constexpr (i >= BSTEP) {
                    if constexpr (PPS * (i - BSTEP) + pi < NDMA) dma_one(kf_late, cur, PPS * (i - BSTEP) + pi);
                } else {
                    if constexpr (PPS * (i + NLATE) + pi < NDMA) dma_one(kf_early, erl, PPS * (i + NLATE) + pi);
                }
            });
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
#ifdef G3_TIMING
    if (lane == 0 && (J % 31) == 0)
        printf("g3t NFR %d WCN %d J %d wave %d stages %d: barrier %llu scale %llu total %llu cycles\n", NFR, WCN, J, wave, nkb, tm_bar, tm_scale, __builtin_readcyclecounter() - tm_begin);
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
        if (b >= a.ntok) return;
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) {
            const int row = row0 + f * 32 + 8 * q4 + 4 * hi;
            float o[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int r = 4 * q4 + e;
                const float other = X[((wc * NFR + f) * 16 + r) * 64 + lane];
                float g = have_gate ? acc[f][r] : other;
                const float up = have_gate ? other : acc[f][r];
                g = g / (float)(1.0 + exp(-(double)g));
                o[e] = g * up;
            }
            if (row < a.rows) *reinterpret_cast<float4*>(a.out + (size_t)b * a.out_stride + row) = make_float4(o[0], o[1], o[2], o[3]);
        }
    };
    // two straight-line branches (a shared loop with a run-time fragment index would move the accumulators to scratch)
    if (wm == 0) g2_static_for<FH, NFR>(park); else g2_static_for<0, FH>(park);
    __syncthreads();
    if (wm == 0) g2_static_for<0, FH>([&](auto fc) { finish(fc, std::true_type{}); });
    else g2_static_for<FH, NFR>([&](auto fc) { finish(fc, std::false_type{}); });
}
