// gl3_prefill_gemm2.hip — translation unit of pf_gemm2_kernel (gl3_prefill_gemm2.h), the batched-prefill Q8_0 GEMM for > 64 tokens.
// Separate from gl3_prefill.hip because it is compiled with -fno-slp-vectorize: its per-block arithmetic must stay scalar
// (v_fma_f32 / v_add_f32) — packed f32 instructions beside MFMAs are slow on gfx950, see the header.
#include "gl3_ctx.h"
#include <type_traits>
#include "gl3_decode_kernels.h"
using namespace gl3;
#include "gl3_bd_gemm.h"          // GemmArgs
#include "gl3_prefill_gemm2.h"
#include "gl3_prefill_gemm3.h"
#include "gl3_prefill_gemm3t.h"

template <int EPI, int RF>
constexpr int g2_lds_bytes() { return G2_RING * g2_stage_bytes((EPI == EPI_SWIGLU ? 2 : 1) * RF * 64); }

template <int EPI, int RF, int NW, int OCC, int MODE>
static void g2_launch(const GemmArgs& a, dim3 grid, hipStream_t s) {
    hipLaunchKernelGGL((pf_gemm2_kernel<EPI, RF, NW, OCC, MODE>), grid, dim3(64 * NW), (g2_lds_bytes<EPI, RF>()), s, a);
}

// Dynamic LDS above 64 KB needs hipFuncAttributeMaxDynamicSharedMemorySize (below it the call is harmless).  Called by
// gl3_prefill_alloc for every plan, after hipSetDevice, like the attributes of the other kernels: the attribute belongs to the
// (function, device) pair, and a once-per-process flag would leave a second device's plan — or a second in-process rank racing on
// the flag — without it (r4 advisor finding).
template <int EPI, int RF, int NW, int OCC>
static hipError_t g2_allow_lds() {
    hipError_t e = hipFuncSetAttribute((const void*)pf_gemm2_kernel<EPI, RF, NW, OCC, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, g2_lds_bytes<EPI, RF>());
    if (e != hipSuccess) return e;
    return hipFuncSetAttribute((const void*)pf_gemm2_kernel<EPI, RF, NW, OCC, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, g2_lds_bytes<EPI, RF>());
}
template <int EPI>
static hipError_t g2_allow_lds_epi() {
    hipError_t e = g2_allow_lds<EPI, 1, 4, 2>();
    if constexpr (EPI != EPI_SWIGLU) {
        if (e == hipSuccess) e = g2_allow_lds<EPI, 2, 4, 2>();
        if (e == hipSuccess) e = g2_allow_lds<EPI, 1, 8, 1>();
    }
    return e;
}
hipError_t gl3_gemm2_allow_lds() {
    hipError_t e = g2_allow_lds_epi<EPI_SWIGLU>();
    if (e == hipSuccess) e = g2_allow_lds_epi<EPI_RESID>();
    if (e == hipSuccess) e = g2_allow_lds_epi<EPI_STORE>();
    return e;
}

// out[b][row] (=, +=, SwiGLU) for ntok > 64 tokens; tile shape by the matrix's row count so that the grid fills the chip:
// 128-row tiles (two fragments per wavefront) when they still give >= 512 workgroups, 64-row tiles otherwise, and 8 wavefronts
// per workgroup when even those leave one workgroup per CU.  mode: GL3_PF_GEMM2 (2 = default; 1 = -B s on the VALU, A/B switch).
template <int EPI>
static void g2_dispatch(GemmArgs a, int rows, int ntok, int mode, hipStream_t s) {
    const int ntt = (ntok + 127) / 128;
    a.ntt = ntt;

    auto grid = [&](int nrt) { a.nrt = nrt; return dim3(8 * ((ntt * nrt + 7) / 8)); };
#define GL3_G2(RF_, NW_, OCC_, NRT_)                                                        \
    do {                                                                                   \
        const dim3 g = grid(NRT_);                                                         \
        if (mode == 1) g2_launch<EPI, RF_, NW_, OCC_, 1>(a, g, s);                       \
        else g2_launch<EPI, RF_, NW_, OCC_, 2>(a, g, s);                                 \
    } while (0)
    if constexpr (EPI == EPI_SWIGLU) GL3_G2(1, 4, 2, (rows + 63) / 64);
    else if ((size_t)ntt * ((rows + 127) / 128) >= 512) GL3_G2(2, 4, 2, (rows + 127) / 128);
    else if ((size_t)ntt * ((rows + 63) / 64) > 256) GL3_G2(1, 4, 2, (rows + 63) / 64);
    else GL3_G2(1, 8, 1, (rows + 63) / 64);
#undef GL3_G2
}

void gl3_gemm2_launch(int epi, const GemmArgs& a, int rows, int ntok, int mode, hipStream_t s) {
    if (epi == EPI_SWIGLU) g2_dispatch<EPI_SWIGLU>(a, rows, ntok, mode, s);
    else if (epi == EPI_RESID) g2_dispatch<EPI_RESID>(a, rows, ntok, mode, s);
    else g2_dispatch<EPI_STORE>(a, rows, ntok, mode, s);
}

// ---------------------------------------------------------------------------------------------------------------- r6: pf_gemm3_kernel
template <int EPI, int RF, int TF, int WR, int WC, int KB>
constexpr int g3_lds_bytes() { return G3_RING * g3_stage_bytes((EPI == EPI_SWIGLU ? 2 : 1) * RF * 32 * WR, WC * TF * 32, KB); }
template <int EPI, int RF, int TF, int WR, int WC, int KB, int OCC>
static void g3_launch(const GemmArgs& a, dim3 grid, hipStream_t s) {
    hipLaunchKernelGGL((pf_gemm3_kernel<EPI, RF, TF, WR, WC, KB, OCC>), grid, dim3(64 * WR * WC), (g3_lds_bytes<EPI, RF, TF, WR, WC, KB>()), s, a);
}
template <int EPI, int RF, int TF, int WR, int WC, int KB, int OCC>
static hipError_t g3_allow() {
    return hipFuncSetAttribute((const void*)pf_gemm3_kernel<EPI, RF, TF, WR, WC, KB, OCC>, hipFuncAttributeMaxDynamicSharedMemorySize, g3_lds_bytes<EPI, RF, TF, WR, WC, KB>());
}
// tile shapes (rows x tokens per workgroup; wavefront grid; blocks per stage):
//   BIG   128 x 128, 2 x 2 wavefronts of 2 x 2 fragments, KB 2, two workgroups per CU   (gate + up as 64 + 64 rows; matrices with >= 512 such tiles)
//   QKV    96 x 128, 3 x 4 one-tile wavefronts, KB 4, one workgroup per CU              (row counts that are a multiple of 96 with 128..384 tiles: 6144-row qkv)
//   SMALL  64 x 128, 2 x 4 one-tile wavefronts, KB 4, one workgroup per CU              (everything else: the 4096-row wo / down projections)
//   SMALL2 64 x 64,  2 x 2 one-tile wavefronts, KB 4, two workgroups per CU             (A/B: GL3_PF_GEMM3_SHAPE=4 — two independent barrier domains per CU)
template <int EPI>
static hipError_t g3_allow_epi() {
    hipError_t e = g3_allow<EPI, EPI == EPI_SWIGLU ? 1 : 2, 2, 2, 2, 2, 2>();
    if constexpr (EPI != EPI_SWIGLU) {
        if (e == hipSuccess) e = g3_allow<EPI, 1, 1, 3, 4, 4, 1>();
        if (e == hipSuccess) e = g3_allow<EPI, 1, 1, 2, 4, 4, 1>();
        if (e == hipSuccess) e = g3_allow<EPI, 1, 1, 2, 2, 4, 2>();
    }
    return e;
}
template <int NFR, int KBT>
static hipError_t g3t_allow() {
    hipError_t e = hipFuncSetAttribute((const void*)pf_gemm3t_kernel<NFR, KBT, false>, hipFuncAttributeMaxDynamicSharedMemorySize, g3t_lds_bytes(NFR, KBT));
    if (e == hipSuccess && KBT == 2) e = hipFuncSetAttribute((const void*)pf_gemm3t_kernel<NFR, 2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, g3t_lds_bytes(NFR, 2));
    return e;
}
template <int NFR, int KBT>
static void g3t_launch(GemmArgs a, int rows, int ntok, hipStream_t s) {
    a.ntt = (ntok + 127) / 128; a.nrt = (rows + 32 * NFR - 1) / (32 * NFR);
    const dim3 g(8 * ((a.ntt * a.nrt + 7) / 8));
    if (KBT == 2 && a.XPo) hipLaunchKernelGGL((pf_gemm3t_kernel<NFR, 2, true>), g, dim3(512), g3t_lds_bytes(NFR, 2), s, a);      // hb leaves the kernel quantised
    else hipLaunchKernelGGL((pf_gemm3t_kernel<NFR, KBT, false>), g, dim3(512), g3t_lds_bytes(NFR, KBT), s, a);
}
// the tiling the gate + up launch of (rows, ntok) takes: 0 = 128 x 128, 4 .. 7 = tall with that many row fragments (gl3_prefill_gemm3t.h)
static int g3_tall_choice(int rows, int ntok, int* kb) {
    static const int tall_env = getenv("GL3_PF_GEMM3_TALL") ? atoi(getenv("GL3_PF_GEMM3_TALL")) : 0;
    static const int tall_kb = getenv("GL3_PF_GEMM3_TALL_KB") ? atoi(getenv("GL3_PF_GEMM3_TALL_KB")) : 2;      // blocks per K stage of the tall tiling (1 | 2)
    if (kb) *kb = tall_kb;
    const int ntt = (ntok + 127) / 128;
    int best = 0, cost = ((ntt * ((rows + 63) / 64) + 511) / 512) * 8;      // 128 x 128: two workgroups per CU, 8 result tiles per SIMD and round
    for (int nfr = 4; nfr <= 7 && tall_env == 0; ++nfr) {                   // tall: one workgroup per CU, 2 NFR tiles per SIMD and round
        const int c = ((ntt * ((rows + 32 * nfr - 1) / (32 * nfr)) + 255) / 256) * 2 * nfr;
        if (c < cost) { cost = c; best = nfr; }
    }
    if (tall_env >= 4 && tall_env <= 7) best = tall_env;
    return best;
}
// true when the gate + up launch can write hb quantised (pf_gemm3t_kernel<.., QOUT>): the caller then passes XQo / XPo and skips the quantise launch
bool gl3_gemm3_swiglu_quantises(int rows, int ntok) {
    static const bool off = getenv("GL3_PF_GEMM3_QOUT") && atoi(getenv("GL3_PF_GEMM3_QOUT")) == 0;
    int kb = 0;
    return !off && g3_tall_choice(rows, ntok, &kb) != 0 && kb == 2 && rows % 32 == 0;
}
hipError_t gl3_gemm3_allow_lds() {
    hipError_t e = g3_allow_epi<EPI_SWIGLU>();
    if (e == hipSuccess) e = g3t_allow<4, 1>();
    if (e == hipSuccess) e = g3t_allow<5, 1>();
    if (e == hipSuccess) e = g3t_allow<6, 1>();
    if (e == hipSuccess) e = g3t_allow<7, 1>();
    if (e == hipSuccess) e = g3t_allow<4, 2>();
    if (e == hipSuccess) e = g3t_allow<5, 2>();
    if (e == hipSuccess) e = g3t_allow<6, 2>();
    if (e == hipSuccess) e = g3t_allow<7, 2>();
    if (e == hipSuccess) e = g3_allow_epi<EPI_RESID>();
    if (e == hipSuccess) e = g3_allow_epi<EPI_STORE>();
    return e;
}
template <int EPI>
static void g3_dispatch(GemmArgs a, int rows, int ntok, hipStream_t s) {
    const int ntt = (ntok + 127) / 128;
    a.ntt = ntt;
    auto grid = [&](int nrt) { a.nrt = nrt; return dim3(8 * ((ntt * nrt + 7) / 8)); };
    if constexpr (EPI == EPI_SWIGLU) {
        // gate + up: the 128 x 128 tiling or a tall tiling (gl3_prefill_gemm3t.h) — whichever leaves a SIMD fewer tile-steps (g3_tall_choice).
        // GL3_PF_GEMM3_TALL: -1 never, 4 .. 7 that shape always.
        int tall_kb = 2;
        const int best = g3_tall_choice(rows, ntok, &tall_kb);
#define GL3_G3T(N_) do { if (tall_kb == 1) g3t_launch<N_, 1>(a, rows, ntok, s); else g3t_launch<N_, 2>(a, rows, ntok, s); } while (0)
        switch (best) {
        case 4: GL3_G3T(4); break;
        case 5: GL3_G3T(5); break;
        case 6: GL3_G3T(6); break;
        case 7: GL3_G3T(7); break;
        default: g3_launch<EPI, 1, 2, 2, 2, 2, 2>(a, grid((rows + 63) / 64), s);
        }
#undef GL3_G3T
    } else {
        const int t128 = ntt * ((rows + 127) / 128), t96 = ntt * ((rows + 95) / 96);
        static const int force = getenv("GL3_PF_GEMM3_SHAPE") ? atoi(getenv("GL3_PF_GEMM3_SHAPE")) : 0;      // A/B: 1 BIG, 2 QKV, 3 SMALL
        const int shape = force ? force : t128 >= 512 ? 1 : (rows % 96 == 0 && t96 > 128 && t96 <= 384) ? 2 : 3;
        if (shape == 1) g3_launch<EPI, 2, 2, 2, 2, 2, 2>(a, grid((rows + 127) / 128), s);
        else if (shape == 2) g3_launch<EPI, 1, 1, 3, 4, 4, 1>(a, grid((rows + 95) / 96), s);
        else if (shape == 4) { a.ntt = (ntok + 63) / 64; const int nrt = (rows + 63) / 64; a.nrt = nrt; g3_launch<EPI, 1, 1, 2, 2, 4, 2>(a, dim3(8 * ((a.ntt * nrt + 7) / 8)), s); }
        else g3_launch<EPI, 1, 1, 2, 4, 4, 1>(a, grid((rows + 63) / 64), s);
    }
}
void gl3_gemm3_launch(int epi, const GemmArgs& a, int rows, int ntok, hipStream_t s) {
    if (epi == EPI_SWIGLU) g3_dispatch<EPI_SWIGLU>(a, rows, ntok, s);
    else if (epi == EPI_RESID) g3_dispatch<EPI_RESID>(a, rows, ntok, s);
    else g3_dispatch<EPI_STORE>(a, rows, ntok, s);
}
