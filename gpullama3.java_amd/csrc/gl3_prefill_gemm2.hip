// gl3_prefill_gemm2.hip — translation unit of pf_gemm2_kernel (gl3_prefill_gemm2.h), the batched-prefill Q8_0 GEMM for > 64 tokens.
// Separate from gl3_prefill.hip because it is compiled with -fno-slp-vectorize: its per-block arithmetic must stay scalar
// (v_fma_f32 / v_add_f32) — packed f32 instructions beside MFMAs are slow on gfx950, see the header.
#include "gl3_ctx.h"
#include <type_traits>
#include "gl3_decode_kernels.h"
using namespace gl3;
#include "gl3_bd_gemm.h"          // GemmArgs
#include "gl3_prefill_gemm2.h"
#include "gl3_prefill_gemm4.h"

// LDS request of a variant; set once (dynamic LDS above 64 KB needs the attribute, below it is harmless)
template <int EPI, int RF, int NW, int OCC, int MODE>
static void g2_launch(const GemmArgs& a, dim3 grid, hipStream_t s) {
    constexpr int AROWS = (EPI == EPI_SWIGLU ? 2 : 1) * RF * 64;
    constexpr int LDS = G2_RING * g2_stage_bytes(AROWS);
    static bool attr = false;
    if (!attr) { (void)hipFuncSetAttribute((const void*)pf_gemm2_kernel<EPI, RF, NW, OCC, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS); attr = true; }
    hipLaunchKernelGGL((pf_gemm2_kernel<EPI, RF, NW, OCC, MODE>), grid, dim3(64 * NW), LDS, s, a);
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

// One tile per wavefront, four wavefronts per SIMD (gl3_prefill_gemm4.h): 128-row workgroup tiles (16 wavefronts, one workgroup per
// CU) when they give every CU a workgroup, else 64-row tiles (8 wavefronts, two workgroups per CU).
template <int EPI, int WR>
static void g4_launch(GemmArgs a, int rows, int ntok, hipStream_t s) {
    constexpr int NM = EPI == EPI_SWIGLU ? 2 : 1, RPM = 32 * WR / NM;
    constexpr int LDS = G2_RING * g2_stage_bytes(32 * WR);
    static bool attr = false;
    if (!attr) { (void)hipFuncSetAttribute((const void*)pf_gemm4_kernel<EPI, WR, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS); attr = true; }
    a.ntt = (ntok + 127) / 128;
    a.nrt = (rows + RPM - 1) / RPM;
    hipLaunchKernelGGL((pf_gemm4_kernel<EPI, WR, 4>), dim3(8 * ((a.ntt * a.nrt + 7) / 8)), dim3(256 * WR), LDS, s, a);
}
template <int EPI>
static void g4_dispatch(const GemmArgs& a, int rows, int ntok, hipStream_t s) {
    constexpr int NM = EPI == EPI_SWIGLU ? 2 : 1;
    const int ntt = (ntok + 127) / 128;
    if ((size_t)ntt * ((rows + 128 / NM - 1) / (128 / NM)) >= 256) g4_launch<EPI, 4>(a, rows, ntok, s);
    else g4_launch<EPI, 2>(a, rows, ntok, s);
}

void gl3_gemm2_launch(int epi, const GemmArgs& a, int rows, int ntok, int mode, hipStream_t s) {
    if (mode >= 4) {
        if (epi == EPI_SWIGLU) g4_dispatch<EPI_SWIGLU>(a, rows, ntok, s);
        else if (epi == EPI_RESID) g4_dispatch<EPI_RESID>(a, rows, ntok, s);
        else g4_dispatch<EPI_STORE>(a, rows, ntok, s);
        return;
    }
    if (epi == EPI_SWIGLU) g2_dispatch<EPI_SWIGLU>(a, rows, ntok, mode, s);
    else if (epi == EPI_RESID) g2_dispatch<EPI_RESID>(a, rows, ntok, mode, s);
    else g2_dispatch<EPI_STORE>(a, rows, ntok, mode, s);
}
