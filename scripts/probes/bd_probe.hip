// Stand-alone timing harness for bdw_gemm_kernel (static-batched decode GEMM; no parity check: random bytes as weights).
// hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I../../gpullama3.java_amd/csrc -I../../include bd_probe.hip -o bd_probe
// ./bd_probe [tokens]   — Qwen3-4B and Llama-3-8B shapes, ring depth / wavefronts-per-SIMD variants, weights rotated through
// 600 MB so that neither L2 nor the 256 MB infinity cache holds them.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include "gl3_ctx.h"
#include "gl3_decode_kernels.h"
using namespace gl3;
#include "gl3_bd_gemm.h"

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

struct Shape { const char* name; int epi, rows, k; };

template <int EPI, int DA, int WPE>
static float run(GemmArgs a, std::vector<uint8_t*>& w, std::vector<uint8_t*>& w2, int iters) {
    a.tslots = BD_TS;
    const dim3 grid(bdw_grid((a.rows + 15) / 16, (a.ntok + 15) / 16)), block(64);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = -3; i < iters; ++i) {
        if (i == 0) { CK(hipDeviceSynchronize()); CK(hipEventRecord(e0, 0)); }
        const int j = (i + 3) % (int)w.size();
        a.w = w[j]; a.w2 = w2.empty() ? nullptr : w2[j];
        hipLaunchKernelGGL((bdw_gemm_kernel<EPI, DA, WPE>), grid, block, 0, 0, a);
    }
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / iters * 1e3f;
}

int main(int argc, char** argv) {
    const int ntok = argc > 1 ? atoi(argv[1]) : 32;
    const Shape shapes[] = {{"qkv    q3-4b", EPI_STORE, 6144, 2560}, {"wo     q3-4b", EPI_RESID, 2560, 4096},
                            {"gateup q3-4b", EPI_SWIGLU, 9728, 2560}, {"down   q3-4b", EPI_RESID, 2560, 9728},
                            {"qkv    8b", EPI_STORE, 6144, 4096}, {"gateup 8b", EPI_SWIGLU, 14336, 4096}, {"down   8b", EPI_RESID, 4096, 14336}};
    for (const Shape& sh : shapes) {
        const int ng = (sh.k + 127) / 128, nstrips = (sh.rows + 15) / 16;
        const size_t wbytes = (size_t)nstrips * ng * TILE_BYTES;
        const int ncopy = (int)((size_t)600 * 1024 * 1024 / wbytes / (sh.epi == EPI_SWIGLU ? 2 : 1)) + 1;
        std::vector<uint8_t*> w, w2;
        std::vector<uint8_t> h(wbytes);
        srand(1);
        for (size_t i = 0; i < wbytes; ++i) h[i] = (uint8_t)(rand() & 0x3f);     // f16 scales stay small finite numbers
        for (int c = 0; c < ncopy; ++c) {
            uint8_t* d; CK(hipMalloc(&d, wbytes + GL3_TAIL_PAD)); CK(hipMemcpy(d, h.data(), wbytes, hipMemcpyHostToDevice)); w.push_back(d);
            if (sh.epi == EPI_SWIGLU) { CK(hipMalloc(&d, wbytes + GL3_TAIL_PAD)); CK(hipMemcpy(d, h.data(), wbytes, hipMemcpyHostToDevice)); w2.push_back(d); }
        }
        const int maxk = 16384;
        uint8_t* XQ; float* XS; float* out;
        CK(hipMalloc(&XQ, (size_t)BD_TS * maxk + GL3_TAIL_PAD)); CK(hipMemset(XQ, 3, (size_t)BD_TS * maxk));
        CK(hipMalloc(&XS, (size_t)BD_TS * (maxk / 32) * 4 + GL3_TAIL_PAD)); CK(hipMemset(XS, 0, (size_t)BD_TS * (maxk / 32) * 4));
        CK(hipMalloc(&out, (size_t)BD_TS * 16384 * 4)); CK(hipMemset(out, 0, (size_t)BD_TS * 16384 * 4));
        GemmArgs a{};
        a.rows = sh.rows; a.ng = ng; a.nb = sh.k / 32; a.XQ = XQ; a.XS = XS; a.maxk = maxk; a.ntok = ntok;
        a.out = out; a.out_stride = 16384; a.out_scale = 1.0f;
        float us[3];
        const char* what = "ring/waves-per-SIMD 8/2 6/2 4/3";
        if (sh.epi == EPI_STORE) { us[0] = run<EPI_STORE, 8, 2>(a, w, w2, 50); us[1] = run<EPI_STORE, 6, 2>(a, w, w2, 50); us[2] = run<EPI_STORE, 4, 3>(a, w, w2, 50); }
        else if (sh.epi == EPI_RESID) { us[0] = run<EPI_RESID, 8, 2>(a, w, w2, 50); us[1] = run<EPI_RESID, 6, 2>(a, w, w2, 50); us[2] = run<EPI_RESID, 4, 3>(a, w, w2, 50); }
        else { what = "ring/waves-per-SIMD 4/2 2/2 2/3"; us[0] = run<EPI_SWIGLU, 4, 2>(a, w, w2, 50); us[1] = run<EPI_SWIGLU, 2, 2>(a, w, w2, 50); us[2] = run<EPI_SWIGLU, 2, 3>(a, w, w2, 50); }
        const double bytes = (double)sh.rows * sh.k * 34.0 / 32.0 * (sh.epi == EPI_SWIGLU ? 2 : 1);
        const float best = fminf(us[0], fminf(us[1], us[2]));
        printf("%-14s rows %6d k %6d  B=%d (%s): %7.2f %7.2f %7.2f us   best %5.2f TB/s\n", sh.name, sh.rows, sh.k, ntok, what, us[0], us[1], us[2], bytes / best * 1e-6);
        for (auto p : w) CK(hipFree(p)); for (auto p : w2) CK(hipFree(p));
        CK(hipFree(XQ)); CK(hipFree(XS)); CK(hipFree(out));
    }
    return 0;
}
