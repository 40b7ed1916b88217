// gl3_bdk_gemm.h — static-batched decode / small-chunk GEMM (<= 64 tokens) with the K dimension split over PRODUCER wavefronts and one ordered
// CHAIN wavefront per (16-row strip, 16-token tile), round 6.  Same operands, layouts (Q8T weights, XQ2 / XS2 activations) and arithmetic as
// bdw_gemm_kernel (gl3_bd_gemm.h), bit for bit:
//     result += (float) isum * (wScale * aScale), blocks ascending                      (Q8_0FloatTensor.java:119)
// bdw_gemm_kernel runs ONE wavefront per (strip, token tile) over all of K: 320 wavefronts on 1024 SIMDs for the 2560-row matrices of Qwen3-4B at
// B = 32, each bound by a lone wavefront's issue rate (~105 cycles per block against ~50 of instruction time; wo 13.3 us for 11 MB).  The only
// parallelism left is K — and the reference's block order allows it: the products p_b = fl(isum_b * s_b) are independent, only `result += p_b` is a chain.
//   * workgroup = P producers + 1 chain wavefront per strip (QOUT: two strips = 2 (P + 1) wavefronts).  Producer w takes the Q8T tiles w, w + P, ..
//     (4 blocks each) with bdw's register ring, swaps and MFMAs, and stores p (one float4 = the lane's 4 rows per block) into a double-buffered LDS
//     array; the chain wavefront adds the P tiles of the previous round in block order — a ds_read_b128 and four adds per block — and owns the epilogue.
//   * one raw `s_waitcnt lgkmcnt(0); s_barrier` per round (P tiles): a __syncthreads() would carry vmcnt(0) and drain the producers' load rings
//     (profiles/r05_tg_depth.md, the lesson of attn_pv_kernel).  r2's k-slice experiments were measured with __syncthreads().
//   * loads are unconditional and unclamped (uniform bases + running 32-bit lane offsets; the reads past the end of a strip land in the next strip or
//     in GL3_TAIL_PAD); tiles / blocks past the end are skipped by the chain, whose full rounds add without per-block conditions.
//   * two workgroups per CU in the launch bounds: a register budget of 256 makes the compiler pick the VGPR form of the MFMAs (with 512 it
//     parked the results in AGPRs: 16 v_accvgpr_read per tile on a wavefront that is bound by its own issue rate).
#pragma once
#include "gl3_bd_gemm.h"

constexpr int BDK_T = 2;          // tiles per producer and round (a barrier per P * BDK_T tiles)
template <int NM, int P>
__host__ __device__ constexpr int bdk_group_floats() { return P * (4 * NM * 64 + 4 * 128) + 2 * P * BDK_T * 4 * NM * 256; }
template <int EPI, int P, bool QOUT>
__host__ __device__ constexpr int bdk_lds_bytes() { return (QOUT ? 2 : 1) * bdk_group_floats<(EPI == EPI_SWIGLU ? 2 : 1), P>() * 4; }

template <int EPI, int P, int DA, bool QOUT = false, int TS = BD_TS>
__global__ __launch_bounds__(64 * (QOUT ? 2 : 1) * (P + 1), 2) void bdk_gemm_kernel(const GemmArgs a) {
    static_assert(!QOUT || EPI == EPI_SWIGLU, "the quantising epilogue is the SwiGLU one");
    static_assert(DA % 2 == 0 && DA % BDK_T == 0, "ring slots are static under the unroll; a round is a whole number of ring slots");
    constexpr int T = BDK_T;
    constexpr int NWV = QOUT ? 2 : 1, NM = (EPI == EPI_SWIGLU) ? 2 : 1, NR = P + 1;
    extern __shared__ __attribute__((aligned(16))) float bdk_smem[];
    __shared__ float amax_s[NWV][64];
    const int lane = threadIdx.x & 63, t = lane & 15, g = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int sg = wave / NR, role = wave % NR;                                   // strip of the workgroup, producer 0 .. P - 1 or chain (P)
    float* grp = bdk_smem + sg * bdk_group_floats<NM, P>();
    float* pr = grp + P * (4 * NM * 64 + 4 * 128);                                // p ring [2][P][T tiles][4 blocks][NM][64 lanes][4]
    float* wsl = grp + (role < P ? role : 0) * (4 * NM * 64 + 4 * 128);           // producer-private scale parking [tile & 3][matrix][block][row]
    float* xsl = wsl + 4 * NM * 64;                                               // [tile & 3][blocks 01 | 23][token][2 blocks][2]
    const int NTG = (a.ntok + 15) >> 4;
    const int nstrips = (a.rows + 15) >> 4;
    const int h = (blockIdx.x >> 3) % NTG;
    const int unit = (blockIdx.x / (8 * NTG)) * 8 + (blockIdx.x & 7);
    if (unit * NWV >= nstrips) return;
    const int strip = unit * NWV + sg;
    const int ntiles = a.ng;
    const int nrounds = (ntiles + P * T - 1) / (P * T);            // rounds of P * T tiles
    const int nrp = (nrounds + 1 + DA / T - 1) / (DA / T) * (DA / T);   // + the chain's last adds, rounded up to whole trips of DA / T rounds
    v2f_t acc[NM][2];
#pragma unroll
    for (int m = 0; m < NM; ++m) { acc[m][0] = v2f_t{0.f, 0.f}; acc[m][1] = v2f_t{0.f, 0.f}; }
#ifdef BDK_TIMING
    unsigned long long tm_work = 0, tm_wait = 0, tm_a = __builtin_readcyclecounter(), tm_begin = tm_a;
#define BDK_BARRIER() do { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); const unsigned long long b_ = __builtin_readcyclecounter(); tm_work += b_ - tm_a; \
                           asm volatile("s_barrier" ::: "memory"); tm_a = __builtin_readcyclecounter(); tm_wait += tm_a - b_; } while (0)
#else
#define BDK_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
#endif

    if (role < P) {
        // ------------------------------------------------------------------------------------------------ producer
        // addresses: uniform bases (SGPRs) + running 32-bit lane offsets, advanced by compile-time steps — a producer's tiles are role * T + j of
        // every round of P * T tiles — and nothing is clamped: the rings read up to 2 * DA * P tiles past the end of a strip / of the
        // activations (GL3_TAIL_PAD covers it for P <= 3; those products are never added)
        const size_t strip_bytes = (size_t)a.ng * TILE_BYTES;
        const uint8_t* pa[NM];
#pragma unroll
        for (int m = 0; m < NM; ++m) pa[m] = (m == 0 ? a.w : a.w2) + (size_t)strip * strip_bytes;
        const uint8_t* pb = a.XQ;
        const uint8_t* px = reinterpret_cast<const uint8_t*>(a.XS);
        uint32_t oa = (uint32_t)(role * T) * TILE_BYTES + ((g & 1) ? 1152 : 128) + 16 * (t + 16 * (g >> 1));
        uint32_t oh = (uint32_t)(role * T) * TILE_BYTES + 4 * (lane & 31);
        uint32_t ob = (uint32_t)(role * T) * (2 * TS * 64) + (16 * h + t) * 64 + 16 * g;
        uint32_t ox = (uint32_t)(role * T) * (TS * 16) + ((16 * h + t) * 4 + g) * 4;
        v4i_t Ar[NM][DA][2]; uint32_t Hr[NM][DA]; v2l_t Br[DA][2]; float Xr[DA];
        auto fetch = [&](int u) {                        // this producer's next tile -> ring slot u (u % T = its index in the round's share)
#pragma unroll
            for (int m = 0; m < NM; ++m) {
#pragma unroll
                for (int jj = 0; jj < 2; ++jj) Ar[m][u][jj] = *reinterpret_cast<const v4i_t*>(pa[m] + oa + 512 * jj);
                Hr[m][u] = *reinterpret_cast<const uint32_t*>(pa[m] + oh);
            }
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) Br[u][jj] = *reinterpret_cast<const v2l_t*>(pb + ob + jj * (TS * 64));
            Xr[u] = *reinterpret_cast<const float*>(px + ox);
            const uint32_t step = (u % T == T - 1) ? (uint32_t)(P * T - T + 1) : 1u;      // tiles to this producer's next one
            oa += step * TILE_BYTES; oh += step * TILE_BYTES; ob += step * (2 * TS * 64); ox += step * (TS * 16);
        };
        auto park = [&](int u, int k) {                  // scales of the tile in ring slot u (local tile k) -> LDS slot k & 3
#pragma unroll
            for (int m = 0; m < NM; ++m) {
                const uint32_t hh = Hr[m][u];
                *reinterpret_cast<float2*>(&wsl[((k & 3) * NM + m) * 64 + 2 * (lane & 31)]) = make_float2(h2f((uint16_t)(hh & 0xffff)), h2f((uint16_t)(hh >> 16)));
            }
            *reinterpret_cast<float2*>(&xsl[(k & 3) * 128 + (g >> 1) * 64 + 4 * t + 2 * (g & 1)]) = make_float2(Xr[u], Xr[u]);
        };
        auto split = [](const v4i_t v, long& lo, long& hi) {
            const auto r0 = __builtin_amdgcn_permlane32_swap((unsigned)v[0], (unsigned)v[2], false, false);
            const auto r1 = __builtin_amdgcn_permlane32_swap((unsigned)v[1], (unsigned)v[3], false, false);
            lo = (long)(((unsigned long)r1[0] << 32) | r0[0]);
            hi = (long)(((unsigned long)r1[1] << 32) | r0[1]);
        };
        const v4i_t cbias = {0x4B400000, 0x4B400000, 0x4B400000, 0x4B400000};
        const v2f_t fbias = {12582912.f, 12582912.f};
        v4i_t Cb[2][NM][4]; float4 Wc[NM][4]; v2f_t Xc[4];
        auto scales = [&](int k) {
            const float4 x01 = *reinterpret_cast<const float4*>(&xsl[(k & 3) * 128 + 4 * t]), x23 = *reinterpret_cast<const float4*>(&xsl[(k & 3) * 128 + 64 + 4 * t]);
            Xc[0] = v2f_t{x01.x, x01.y}; Xc[1] = v2f_t{x01.z, x01.w}; Xc[2] = v2f_t{x23.x, x23.y}; Xc[3] = v2f_t{x23.z, x23.w};
#pragma unroll
            for (int m = 0; m < NM; ++m)
#pragma unroll
                for (int bi = 0; bi < 4; ++bi) Wc[m][bi] = *reinterpret_cast<const float4*>(&wsl[((k & 3) * NM + m) * 64 + 16 * bi + 4 * g]);
        };
        auto front = [&](int u) {
            const int q = u & 1;
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) {
                const long blo = Br[u][jj][0], bhi = Br[u][jj][1];
#pragma unroll
                for (int m = 0; m < NM; ++m) {
                    long alo, ahi;
                    split(Ar[m][u][jj], alo, ahi);
                    Cb[q][m][2 * jj] = __builtin_amdgcn_mfma_i32_16x16x32_i8(alo, blo, cbias, 0, 0, 0);
                    Cb[q][m][2 * jj + 1] = __builtin_amdgcn_mfma_i32_16x16x32_i8(ahi, bhi, cbias, 0, 0, 0);
                }
            }
        };
        auto back = [&](int u, int buf, int tt) {        // p of the tile in slot u -> ring buffer buf, tile tt of this producer's round share
            const int q = u & 1;
            float* dst = pr + ((size_t)((buf * P + role) * T + tt) * 4) * NM * 256 + lane * 4;
#pragma unroll
            for (int bi = 0; bi < 4; ++bi) {
#pragma unroll
                for (int m = 0; m < NM; ++m) {
                    const v4i_t c = Cb[q][m][bi];
                    const float4 w4 = Wc[m][bi];
                    const v2f_t ca = v2f_t{__int_as_float(c[0]), __int_as_float(c[1])} - fbias;
                    const v2f_t cb = v2f_t{__int_as_float(c[2]), __int_as_float(c[3])} - fbias;
                    const v2f_t p0 = ca * (v2f_t{w4.x, w4.y} * Xc[bi]), p1 = cb * (v2f_t{w4.z, w4.w} * Xc[bi]);      // isum * (wScale * aScale)
                    *reinterpret_cast<float4*>(dst + (bi * NM + m) * 256) = make_float4(p0[0], p0[1], p1[0], p1[1]);
                }
            }
        };
#pragma unroll
        for (int u = 0; u < DA; ++u) { fetch(u); __builtin_amdgcn_sched_barrier(0); }
        park(0, 0);
        front(0);
        // round k: this producer's tile k.  nrp rounds = nrounds + 1 (the chain's last adds) rounded up to whole trips: every trip is branch-free (a
        // condition around the ring's loads makes the compiler drain it, DESIGN.md "conditional loads"); the surplus rounds recompute the last tile
        for (int base = 0; base < nrp * T; base += DA) {     // base, k: local tile index; round k / T
#pragma unroll
            for (int u = 0; u < DA; ++u) {
                const int k = base + u;
                scales(k);
                park((u + 1) % DA, k + 1);
                front((u + 1) % DA);
                back(u, (k / T) & 1, u % T);
                fetch(u);
                if (u % T == T - 1) BDK_BARRIER();
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    } else {
        // ------------------------------------------------------------------------------------------------ chain
        for (int k = 0; k < nrp; ++k) {
            if (k > 0) {
                const float* src = pr + ((size_t)(((k - 1) & 1) * P * T) * 4) * NM * 256 + lane * 4;
                // one tile's products (4 blocks x NM float4) are read a tile ahead of their adds and pinned there: left alone the scheduler sinks every read
                // next to its add and each of the 24 dependent steps of a round waits out an LDS round trip (stamps: 1.6 - 2.2 k cycles per round)
                float4 pv[2][4 * NM];
                auto rd = [&](int w, int set) {
#pragma unroll
                    for (int e = 0; e < 4 * NM; ++e) pv[set][e] = *reinterpret_cast<const float4*>(src + (w * 4 * NM + e) * 256);
                };
                rd(0, 0);
                auto adds = [&](auto full_tag) {
                    constexpr bool FULL = decltype(full_tag)::value;
#pragma unroll
                    for (int w = 0; w < P * T; ++w) {                         // producer w / T, its tile w % T: ascending tile order
                        if (w + 1 < P * T) rd(w + 1, (w + 1) & 1);
                        __builtin_amdgcn_sched_barrier(0);
                        const int nv = FULL ? 4 : a.nb - 4 * ((k - 1) * P * T + w);      // real blocks of this tile (<= 0: past the end)
#pragma unroll
                        for (int bi = 0; bi < 4; ++bi) {
                            if (!FULL && bi >= nv) continue;
#pragma unroll
                            for (int m = 0; m < NM; ++m) {
                                const float4 p = pv[w & 1][bi * NM + m];
                                acc[m][0] = acc[m][0] + v2f_t{p.x, p.y};     // result +=, blocks ascending
                                acc[m][1] = acc[m][1] + v2f_t{p.z, p.w};
                            }
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                };
                if (4 * k * P * T <= a.nb) adds(std::true_type{});            // every block of the round is a real one: no per-block conditions
                else adds(std::false_type{});
            }
            BDK_BARRIER();
        }
    }
#ifdef BDK_TIMING
    if (lane == 0 && (blockIdx.x % 37) == 0)
        printf("bdk EPI %d P %d rows %d nb %d role %d rounds %d: work %llu barrier %llu total %llu cycles\n", EPI, P, a.rows, a.nb, role, nrp, tm_work, tm_wait, __builtin_readcyclecounter() - tm_begin);
#endif
    // ---------------------------------------------------------------------------------------------------- epilogue (chain wavefronts)
    const int b = 16 * h + t;
    if constexpr (QOUT) {
        float hv[4] = {0.f, 0.f, 0.f, 0.f};
        if (role == P) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float gt = acc[0][i >> 1][i & 1];
                gt = gt / (float)(1.0 + exp(-(double)gt));
                hv[i] = gt * acc[NM - 1][i >> 1][i & 1];
            }
            amax_s[sg][lane] = fmaxf(fmaxf(fabsf(hv[0]), fabsf(hv[1])), fmaxf(fabsf(hv[2]), fabsf(hv[3])));
        }
        BDK_BARRIER();
        if (role != P) return;
        float amax = 0.f;
#pragma unroll
        for (int w = 0; w < 2; ++w)
#pragma unroll
            for (int gg = 0; gg < 4; ++gg) amax = fmaxf(amax, amax_s[w][16 * gg + t]);
        if (b >= a.ntok) return;
        const float qs = amax / 127.0f;
        const float ainv = qs != 0.f ? 1.0f / qs : 0.f;
        uint32_t packed = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float sv = hv[i] * ainv;
            packed |= (uint32_t)((int)(sv + copysignf(0.5f, sv)) & 0xFF) << (8 * i);
        }
        const int blk = strip >> 1;
        *reinterpret_cast<uint32_t*>(a.XQo + bdq_offset(blk * 8 + (strip & 1) * 4 + g, b, TS)) = packed;
        if (sg == 0 && g == 0) a.XSo[bds_offset(blk, b, TS)] = (float)(_Float16)qs;
        return;
    }
    if (role != P || b >= a.ntok) return;
    const int rbase = strip * 16 + 4 * g;
    float* o = a.out + (size_t)b * a.out_stride + rbase;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if (rbase + i >= a.rows) continue;
        const float v0 = acc[0][i >> 1][i & 1];
        if (EPI == EPI_SWIGLU) {
            float gt = v0;
            gt = gt / (float)(1.0 + exp(-(double)gt));
            o[i] = gt * acc[NM - 1][i >> 1][i & 1];
        } else if (EPI == EPI_STORE) o[i] = v0 * a.out_scale;
        else o[i] = o[i] + v0 * a.out_scale;
    }
#undef BDK_BARRIER
}
