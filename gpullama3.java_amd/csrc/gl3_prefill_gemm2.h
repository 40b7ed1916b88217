// gl3_prefill_gemm2.h — batched-prefill Q8_0 GEMM (> 64 tokens), round 4: the per-block f32 arithmetic of the reference
//     result += (float) isum * (wScale * aScale)                      (Q8_0FloatTensor.java:119, blocks ascending)
// with only TWO VALU operations per output element and block instead of four.  (Compiled in gl3_prefill_gemm2.hip.)
//
// The r1-r3 kernel (pf_gemm_kernel) issued, per (32 x 32 tile, block), one 32-cycle v_mfma_i32_32x32x32_i8 and 32 packed f32
// instructions (bias subtract, scale product, multiply, add).  Three exact identities move two of the four operations onto the
// matrix pipe (checked bit for bit incl. f16 subnormals, zeros, the f16 maximum and negative block scales by
// scripts/probes/scale_mfma_probe.hip, profiles/r04_scale_mfma_probe.txt):
//   (1) wScale and aScale are f16 values, so s = wScale * aScale is EXACT in f32 (11 + 11 significand bits <= 24);
//   (2) the int8 MFMA's accumulator starts at the integer 0x4B400000, so its output read as f32 is D = B + isum exactly with
//       B = 12582912 = 3 * 2^22, and B * s is exact too (2 + 22 bits), hence
//           fl(float(isum) * s) = fl(D * s - B * s) = fma(D, s, -B s)          one rounding, the reference's;
//   (3) an outer product of exactly representable 16-bit operands is what a 16-bit MFMA delivers exactly: with the bf16 splits
//       w = w_hi + w_lo, a = a_hi + a_lo (hi = the top 8 significand bits, lo = the remaining <= 3),
//           s    = sum over {w_hi, w_hi, w_lo, w_lo} x {a_hi, a_lo, a_hi, a_lo}                            (4 k-slots)
//           -B s = the same four terms scaled by -2^23 (k-slots of lanes 0-31) plus by -2^22 (lanes 32-63)   (8 k-slots)
//       every partial sum is a same-sign multiple of one ulp(w) ulp(a) below 2^24 of them, so the accumulation order and the
//       internal precision of the MFMA cannot matter.  v_mfma_f32_32x32x8_bf16_1k carries 4 k-slots per lane half in two
//       VGPRs: exactly these operands.
// Per (tile, block): three 32-cycle MFMAs (int8 dot, s, -B s) and 16 v_fma_f32 + 16 v_add_f32 — SCALAR on purpose (this
// translation unit is built with -fno-slp-vectorize): packed f32 instructions issued beside MFMAs cost ~13 extra cycles each
// (MI355X_MICROARCH.md, "price of one filler beside MFMAs").
//
// Data movement (r4, after in-kernel A/B runs: LDS stores through VGPRs 50 us, the global loads 80 us and the arithmetic 177 us
// of a 262 us gate/up launch did NOT overlap):
//   * K advances 2 blocks per stage through a THREE-deep LDS ring
//       Aq[blk][half][AROWS][16 B] int8 weights  | At[blk][half][AROWS][16 B] scale operands {s0, s1, n0, n1} of a weight row
//       Bq[blk][half][128 tokens][16 B] int8 activations | Bs[blk][128 tokens][8 B] {p, p}, p = bf16 pair (a_hi, a_lo)
//   * the int8 operands travel HBM/L2 -> LDS by LDS-DMA (global_load_lds_dwordx4: no staging VGPRs, no ds_write pass); a wave's
//     64 lanes land in 1 KB of consecutive LDS, so the LDS image order is chosen per lane on the SOURCE side (the xor bank
//     spread of the token pieces too);
//   * the scale operands are built once per (row, block) / (token, block) by all threads (one entry each);
//   * during stage k the workgroup computes from ring slot k % 3, prefetches the first operands of slot (k + 1) % 3 and fills
//     slot (k + 2) % 3, so the MFMA / VALU pipeline runs across the stage barrier without draining.
#pragma once

typedef short v4s_t __attribute__((ext_vector_type(4)));
typedef int v2i_t __attribute__((ext_vector_type(2)));
typedef float v16f2_t __attribute__((ext_vector_type(16)));

constexpr int G2_KB = 2, G2_TOK = 128, G2_RING = 3;
__host__ __device__ constexpr int g2_stage_bytes(int arows) { return G2_KB * (2 * arows * 16 + 2 * arows * 16 + 2 * G2_TOK * 16 + G2_TOK * 8); }

template <int I, int N, class F>
__device__ __forceinline__ void g2_static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        g2_static_for<I + 1, N>(f);
    }
}

__device__ __forceinline__ uint32_t g2_bf16_dup(float x) {             // {bf16(x), bf16(x)} of an x with <= 8 significand bits
    const uint32_t b = __float_as_uint(x);
    return (b >> 16) | (b & 0xFFFF0000u);
}
__device__ __forceinline__ void g2_dma16(const uint8_t* src, uint8_t* lds_wave_base) {   // lane l's 16 bytes -> lds_wave_base + 16 l
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// MODE 2: s and -B s from the matrix pipe; MODE 1: s from the matrix pipe, -B s = s * (-B) on the VALU (A/B switch)
template <int EPI, int RF, int NW, int OCC, int MODE>
__global__ __launch_bounds__(64 * NW, OCC) void pf_gemm2_kernel(const GemmArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    constexpr int KB = G2_KB, TOK = G2_TOK;
    constexpr int NM = (EPI == EPI_SWIGLU) ? 2 : 1;
    constexpr int NF = NM * RF;                        // 32-row fragments per wavefront (matrix or row fragment)
    static_assert(NF <= 2, "accumulator budget");
    static_assert(NW == 4 || (NW == 8 && NF == 1), "8-wavefront layout is for the single-fragment variant");
    constexpr int NT = 64 * NW;
    constexpr int TF = 8 / NW;                         // 32-token fragments per wavefront
    constexpr int AROWS = NF * 64;                     // weight rows staged per K stage (two wavefronts along the rows)
    constexpr int RPM = AROWS / NM;                    // output rows per matrix covered by this workgroup
    constexpr int OFF_AT = KB * 2 * AROWS * 16, OFF_BQ = 2 * OFF_AT, OFF_BS = OFF_BQ + KB * 2 * TOK * 16;
    constexpr int STAGE = g2_stage_bytes(AROWS);
    static_assert(STAGE == OFF_BS + KB * TOK * 8, "stage layout");
    constexpr int NLA = KB * 2 * AROWS / 64, NLB = KB * 2 * TOK / 64;      // LDS-DMA wave-loads per stage: weights, activations
    constexpr int NDMA = (NLA + NLB + NW - 1) / NW;                        // ... per wavefront
    constexpr int NAT = AROWS * KB, NBS = TOK * KB;                        // scale-operand entries per stage (one thread each)
    static_assert(NAT <= NT && NBS <= NT, "one scale entry per staging thread");
    static_assert(64 % 1 == 0 && AROWS % 64 == 0, "a weight wave-load covers 64 rows of one (block, half)");
    const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int tl = lane & 31, hi = lane >> 5;
    const int wr = NW == 4 ? wave >> 1 : wave >> 2;    // wavefront grid: row part wr,
    const int wc = NW == 4 ? wave & 1 : wave & 3;      // tokens wc * 32 * TF ..
    // XCD-aware tile mapping (as pf_gemm_kernel): the token tiles that share a weight row tile sit on ONE XCD
    const int ntt_g = a.ntt, per_xcd = (a.ntt * a.nrt + 7) >> 3;
    const int lin = blockIdx.x, J = (lin & 7) * per_xcd + (lin >> 3);
    if (J >= ntt_g * a.nrt) return;
    const int row0 = (J / ntt_g) * RPM;
    const int tok0 = (J % ntt_g) * TOK;
    const uint32_t strip_bytes = (uint32_t)a.ng * TILE_BYTES;
    const int nkb = (a.nb + KB - 1) / KB;              // stages that hold at least one real block
    const int nstrips = (a.rows + 15) >> 4;
    constexpr int SPM = RPM / 16;                      // strips per matrix in this workgroup
    auto strip_off = [&](int lrow) -> uint32_t {       // byte offset of local row lrow's strip inside its matrix
        const int sl = lrow >> 4;
        return (uint32_t)min(nstrips - 1, (row0 >> 4) + (NM == 2 ? sl % SPM : sl)) * strip_bytes;
    };
    auto mat_of = [&](int lrow) -> const uint8_t* { return (NM == 2 && (lrow >> 4) / SPM) ? a.w2 : a.w; };

    // ---- per-lane source offsets of this wavefront's LDS-DMA loads (stage-independent part), LDS destination = load id * 1 KB
    uint32_t dma_off[NDMA];
#pragma unroll
    for (int u = 0; u < NDMA; ++u) {
        const int j = wave + NW * u;                   // load id: [0, NLA) weights, [NLA, NLA + NLB) activations
        if (j < NLA) {
            const int e = 64 * j + lane, c = e / AROWS, row = e % AROWS;       // c = blk * 2 + half
            dma_off[u] = strip_off(row) + ((c & 1) ? 1152 : 128) + 16 * ((c >> 1) * 16 + (row & 15));
        } else {
            const int jb = j - NLA, c = jb / (TOK / 64), tk = (((jb % (TOK / 64)) * 64 + lane) ^ c);   // LDS slot p holds token p ^ c (bank spread)
            dma_off[u] = (uint32_t)min(a.ntok - 1, tok0 + tk) * (uint32_t)a.maxk + 16 * c;
        }
    }
#ifdef G2_NO_DMA
    v4i_t stg[NDMA];                                   // A/B switch: the same pieces through VGPRs (global_load_dwordx4 now, ds_write_b128 after the stage's arithmetic)
#endif
    auto dma_one = [&](int kb, int slot, int u) {      // this wavefront's u-th piece of K stage kb's int8 operands -> ring slot
        const uint32_t aoff = (uint32_t)(kb >> 1) * TILE_BYTES + (kb & 1) * 512;     // tile group, lane half of the stage's two blocks
        uint8_t* base = smem + slot * STAGE;
        {
            const int j = wave + NW * u;
#ifdef G2_NO_DMA
            (void)base;
            if (j < NLA) stg[u] = *reinterpret_cast<const v4i_t*>(mat_of((64 * j) % AROWS) + aoff + dma_off[u]);
            else if (j < NLA + NLB) stg[u] = *reinterpret_cast<const v4i_t*>(a.XQ + (size_t)kb * (32 * KB) + dma_off[u]);
#else
            if (j < NLA) g2_dma16(mat_of((64 * j) % AROWS) + aoff + dma_off[u], base + 1024 * j);
            else if (j < NLA + NLB) g2_dma16(a.XQ + (size_t)kb * (32 * KB) + dma_off[u], base + OFF_BQ + 1024 * (j - NLA));
#endif
        }
    };
    auto dma_stage = [&](int kb, int slot) {
#pragma unroll
        for (int u = 0; u < NDMA; ++u) dma_one(kb, slot, u);
    };
    auto stage_store = [&](int slot) {                 // G2_NO_DMA: the staged pieces -> ring slot (same image as the LDS-DMA writes)
#ifdef G2_NO_DMA
        uint8_t* base = smem + slot * STAGE;
#pragma unroll
        for (int u = 0; u < NDMA; ++u) {
            const int j = wave + NW * u;
            if (j < NLA) *reinterpret_cast<v4i_t*>(base + 1024 * j + 16 * lane) = stg[u];
            else if (j < NLA + NLB) *reinterpret_cast<v4i_t*>(base + OFF_BQ + 1024 * (j - NLA) + 16 * lane) = stg[u];
        }
#endif
    };
    // ---- scale operands: thread t < NAT builds the entry of (row t % AROWS, block t / AROWS), t < NBS that of (token t % 128, block t / 128)
    const int s_row = t % AROWS, s_blk = (t / AROWS) % KB, x_blk = (t / TOK) % KB;
    const uint8_t* s_wp = mat_of(s_row) + strip_off(s_row) + 2 * (s_blk * 16 + (s_row & 15));
    const float* s_xp = a.XS + (size_t)min(a.ntok - 1, tok0 + t % TOK) * (a.maxk >> 5) + x_blk;
    uint16_t r_ws = 0;
    float r_xs = 0.f;
    auto scale_load = [&](int kb) {
        if (t < NAT) r_ws = *reinterpret_cast<const uint16_t*>(s_wp + (size_t)(kb >> 1) * TILE_BYTES + (kb & 1) * 64);
        if (t < NBS) r_xs = s_xp[kb * KB];
    };
    auto scale_store = [&](int kb, int slot) {
        uint8_t* base = smem + slot * STAGE;
        if (t < NAT) {
            const float wf = h2f(r_ws);
            const float whi = __uint_as_float(__float_as_uint(wf) & 0xFFFF0000u), wlo = wf - whi;     // 8 + <= 3 significand bits
            const v4i_t lo = {(int)g2_bf16_dup(whi), (int)g2_bf16_dup(wlo), (int)g2_bf16_dup(whi * -8388608.f), (int)g2_bf16_dup(wlo * -8388608.f)};
            const v4i_t hh = {0, 0, (int)g2_bf16_dup(whi * -4194304.f), (int)g2_bf16_dup(wlo * -4194304.f)};
            *reinterpret_cast<v4i_t*>(base + OFF_AT + ((size_t)(s_blk * 2 + 0) * AROWS + s_row) * 16) = lo;
            *reinterpret_cast<v4i_t*>(base + OFF_AT + ((size_t)(s_blk * 2 + 1) * AROWS + s_row) * 16) = hh;
        }
        if (t < NBS) {
            // ragged K: the padded blocks of the last tile group carry zero weights; zero their activation scale too
            const float av = kb * KB + x_blk < a.nb ? r_xs : 0.f;
            const float ahi = __uint_as_float(__float_as_uint(av) & 0xFFFF0000u), alo = av - ahi;
            const uint32_t p = (__float_as_uint(ahi) >> 16) | (__float_as_uint(alo) & 0xFFFF0000u);
            *reinterpret_cast<uint2*>(base + OFF_BS + ((size_t)x_blk * TOK + t % TOK) * 8) = make_uint2(p, p);
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
    v4i_t bf[TF], af[NF], at[NF];
    v4s_t bp[TF];
    v16i_t D[2];
    v16f2_t S[2], N[1];
    uint32_t la[NF], lb[TF][KB], lp[TF];
#pragma unroll
    for (int f = 0; f < NF; ++f) la[f] = (uint32_t)((hi * AROWS + (NM == 2 ? f * RPM + wr * 32 : wr * (32 * RF) + f * 32) + tl) * 16);
#pragma unroll
    for (int tf = 0; tf < TF; ++tf) {
        const int tk = wc * (32 * TF) + tf * 32 + tl;
        lp[tf] = (uint32_t)(OFF_BS + tk * 8);
#pragma unroll
        for (int b = 0; b < KB; ++b) lb[tf][b] = (uint32_t)(OFF_BQ + ((b * 2 + hi) * TOK + (tk ^ (b * 2 + hi))) * 16);
    }
    auto load_a = [&](const uint8_t* sb, int blk, int f) {
        af[f] = *reinterpret_cast<const v4i_t*>(sb + la[f] + blk * (2 * AROWS * 16));
        at[f] = *reinterpret_cast<const v4i_t*>(sb + OFF_AT + la[f] + blk * (2 * AROWS * 16));
    };
    auto load_b = [&](const uint8_t* sb, int blk, int tf) {
        bf[tf] = *reinterpret_cast<const v4i_t*>(sb + lb[tf][blk]);
        bp[tf] = *reinterpret_cast<const v4s_t*>(sb + lp[tf] + blk * (TOK * 8));
    };
    constexpr int NTILE = KB * NF * TF, TPB = NF * TF;
#ifdef G2_NO_MFMA
    auto issue_d = [&](int f, int tf, int buf) { D[buf][0] = af[f][0] + bf[tf][0]; };
#else
    auto issue_d = [&](int f, int tf, int buf) { D[buf] = __builtin_amdgcn_mfma_i32_32x32x32_i8(af[f], bf[tf], cbias, 0, 0, 0); };
#endif
    auto issue_s = [&](int f, int tf, int buf) {
#ifdef G2_NO_MFMA
        S[buf][0] = __int_as_float(at[f][0]); return;
#endif
        S[buf] = __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(__builtin_bit_cast(v4s_t, v2i_t{at[f][0], at[f][1]}), bp[tf], zero16, 0, 0, 0);
    };
    auto issue_n = [&](int f, int tf) {
#ifdef G2_NO_MFMA
        N[0][0] = __int_as_float(at[f][2]); return;
#endif
        if constexpr (MODE == 2) N[0] = __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(__builtin_bit_cast(v4s_t, v2i_t{at[f][2], at[f][3]}), bp[tf], zero16, 0, 0, 0);
    };
    // After the three MFMAs of tile `it` (index inside its stage, whose ring slot is sb_t; sb_f = the slot of the stage after
    // it) have been issued: fragment registers that no later tile of the block reads are refilled with the next block's.
    auto refill = [&](auto itc, const uint8_t* sb_t, const uint8_t* sb_f) {
        constexpr int it = decltype(itc)::value, fi = (it % TPB) / TF, tfi = it % TF, bi = it / TPB;
        constexpr bool wrap = bi + 1 >= KB;
        const uint8_t* sb_o = wrap ? sb_f : sb_t;
        constexpr int bo = wrap ? 0 : bi + 1;
        if constexpr (tfi == TF - 1) load_a(sb_o, bo, fi);       // last token fragment of row fragment fi
        if constexpr (fi == NF - 1) load_b(sb_o, bo, tfi);       // last row fragment of token fragment tfi
    };

    // ---- prologue: stages 0 and 1 complete in the ring, the first tile's MFMAs in flight
    dma_stage(0, 0);
    scale_load(0);
    stage_store(0);
    scale_store(0, 0);
    if (nkb > 1) {
        dma_stage(1, 1);
        scale_load(1);
        stage_store(1);
        scale_store(1, 1);
    }
    __syncthreads();                                   // (waits for the LDS-DMA too)
#pragma unroll
    for (int tf = 0; tf < TF; ++tf) load_b(smem, 0, tf);
#pragma unroll
    for (int f = 0; f < NF; ++f) load_a(smem, 0, f);
    issue_d(0, 0, 0);
    issue_s(0, 0, 0);
    issue_n(0, 0);
    refill(std::integral_constant<int, 0>{}, smem, smem + STAGE);
    int cur = 0;                                       // ring slot of stage kb
#ifdef G2_TIMING
    unsigned long long tm_issue = 0, tm_comp = 0, tm_store = 0, tm_bar = 0, tm0 = __builtin_readcyclecounter(), tm_begin = tm0;
#define G2_STAMP(acc_) do { const unsigned long long n_ = __builtin_readcyclecounter(); acc_ += n_ - tm0; tm0 = n_; } while (0)
#else
#define G2_STAMP(acc_) do {} while (0)
#endif
    for (int kb = 0; kb < nkb; ++kb) {
        const int nxt = cur == G2_RING - 1 ? 0 : cur + 1, fil = nxt == G2_RING - 1 ? 0 : nxt + 1;
        // slot fil is filled during this stage: scale loads now, one LDS-DMA piece after each of the first NDMA steps (all the
        // wavefronts of a CU issuing their pieces at once queue behind each other in the texture-address unit: ~700 cycles per
        // stage measured).  Past the last stage the fill repeats stage nkb - 1 into a slot nobody reads: no branches.
        const int kf = min(kb + 2, nkb - 1);
#if !defined(G2_SKIP_MEM) && !defined(G2_ASYNC_EXPERIMENT)
        scale_load(kf);
#endif
        G2_STAMP(tm_issue);
        const uint8_t* sb_cur = smem + cur * STAGE;
        const uint8_t* sb_nxt = smem + nxt * STAGE;    // past the last stage: a stale slot, results unused
        // step i finishes (tile, block) i of this stage and issues the MFMAs of step i + 1 (step 0 of the next stage at the end):
        //   [int8 MFMA i+1] [8 fma i] [s MFMA i+1] [8 fma i] [-B s MFMA i+1, operand refill] [16 adds i]
        // The schedule is pinned (sched_barrier + value pins): left alone, the compiler issues every MFMA of a stage first and
        // spills the result tiles, and its IR-level sinking moves the arithmetic behind the stage's LDS stores.
        g2_static_for<0, NTILE>([&](auto ic) {
            constexpr int i = decltype(ic)::value, f = (i % TPB) / TF, tf = i % TF;
            constexpr int in = (i + 1) % NTILE, fn = (in % TPB) / TF, tfn = in % TF;
            constexpr bool next_stage = i + 1 == NTILE;
            static_assert(!next_stage || KB > 1, "the refill after the next stage's first tile stays inside that stage");
#ifdef G2_PACKED
            v2f_t cf[8];
            auto fma8 = [&](int r0) {
#pragma unroll
                for (int r = r0 / 2; r < r0 / 2 + 4; ++r) {
                    const v2f_t d = {__int_as_float(D[i & 1][2 * r]), __int_as_float(D[i & 1][2 * r + 1])}, sv = {S[i & 1][2 * r], S[i & 1][2 * r + 1]};
                    const v2f_t nv = MODE == 2 ? v2f_t{N[0][2 * r], N[0][2 * r + 1]} : sv * v2f_t{-12582912.f, -12582912.f};
                    cf[r] = __builtin_elementwise_fma(d, sv, nv);
                }
                asm volatile("" : "+v"(cf[r0 / 2]), "+v"(cf[r0 / 2 + 1]), "+v"(cf[r0 / 2 + 2]), "+v"(cf[r0 / 2 + 3]));
            };
#else
            float cf[16];
            auto fma8 = [&](int r0) {
#pragma unroll
                for (int r = r0; r < r0 + 8; ++r) {
                    const float d = __int_as_float(D[i & 1][r]), sv = S[i & 1][r];
                    const float nv = MODE == 2 ? N[0][r] : sv * -12582912.f;
                    cf[r] = __builtin_fmaf(d, sv, nv);                             // = fl(float(isum) * (wScale * aScale))
                }
                asm volatile("" : "+v"(cf[r0]), "+v"(cf[r0 + 1]), "+v"(cf[r0 + 2]), "+v"(cf[r0 + 3]), "+v"(cf[r0 + 4]), "+v"(cf[r0 + 5]), "+v"(cf[r0 + 6]), "+v"(cf[r0 + 7]));
            };
#endif
            issue_d(fn, tfn, (i + 1) & 1);
            __builtin_amdgcn_sched_barrier(0);
#ifndef G2_NO_VALU
            fma8(0);
#endif
            __builtin_amdgcn_sched_barrier(0);
            issue_s(fn, tfn, (i + 1) & 1);
            __builtin_amdgcn_sched_barrier(0);
#ifndef G2_NO_VALU
            fma8(8);
#endif
            __builtin_amdgcn_sched_barrier(0);
            issue_n(fn, tfn);
            refill(std::integral_constant<int, in>{}, next_stage ? sb_nxt : sb_cur, sb_nxt);
            constexpr int DSTRIDE = NTILE >= 2 * NDMA ? 1 : 1;
#ifndef G2_SKIP_MEM
            if constexpr (i % DSTRIDE == 0 && i / DSTRIDE < NDMA) dma_one(kf, fil, i / DSTRIDE);
#endif
            __builtin_amdgcn_sched_barrier(0);
#ifdef G2_NO_VALU
            acc[f][tf][0] += __int_as_float(D[i & 1][0]) + S[i & 1][0] + N[0][0];
#elif defined(G2_PACKED)
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const v2f_t sum = v2f_t{acc[f][tf][2 * r], acc[f][tf][2 * r + 1]} + cf[r];
                acc[f][tf][2 * r] = sum[0]; acc[f][tf][2 * r + 1] = sum[1];
            }
#else
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[f][tf][r] = acc[f][tf][r] + cf[r];       // result +=, blocks ascending
#endif
            asm volatile("" : "+v"(acc[f][tf][0]), "+v"(acc[f][tf][1]), "+v"(acc[f][tf][2]), "+v"(acc[f][tf][3]), "+v"(acc[f][tf][4]), "+v"(acc[f][tf][5]),
                              "+v"(acc[f][tf][6]), "+v"(acc[f][tf][7]), "+v"(acc[f][tf][8]), "+v"(acc[f][tf][9]), "+v"(acc[f][tf][10]), "+v"(acc[f][tf][11]),
                              "+v"(acc[f][tf][12]), "+v"(acc[f][tf][13]), "+v"(acc[f][tf][14]), "+v"(acc[f][tf][15]));
            __builtin_amdgcn_sched_barrier(0);
        });
        G2_STAMP(tm_comp);
        static_assert(NDMA <= NTILE, "one LDS-DMA piece per step");
#if !defined(G2_SKIP_MEM) && !defined(G2_ASYNC_EXPERIMENT)
        stage_store(fil);
        scale_store(kf, fil);
#endif
#ifdef G2_TIMING
        __builtin_amdgcn_s_waitcnt(0);                 // vmcnt(0) lgkmcnt(0): the LDS-DMA wait is charged to "store", not to the barrier
#endif
        G2_STAMP(tm_store);
#ifdef G2_ASYNC_EXPERIMENT
        // timing experiment (wrong results: the scale operands are never refreshed): only LDS-DMA in the loop, and the barrier
        // leaves the pieces issued during THIS stage in flight (they are needed one stage later)
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" :: "n"(NDMA) : "memory");
#else
        __syncthreads();                               // slot fil complete for every wavefront; slot cur free
#endif
        G2_STAMP(tm_bar);
        cur = nxt;
    }
#ifdef G2_TIMING
    if (lane == 0 && (J % 97) == 0)
        printf("g2 EPI %d NW %d J %d wave %d stages %d: issue %llu compute %llu store+wait %llu barrier %llu total %llu cycles\n", EPI, NW, J, wave, nkb, tm_issue, tm_comp,
               tm_store, tm_bar, __builtin_readcyclecounter() - tm_begin);
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
