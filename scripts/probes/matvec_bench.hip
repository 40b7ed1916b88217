// Stand-alone timing harness for matvec_q8t_kernel (no parity check: random bytes as weights).
// hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I../../gpullama3.java_amd/csrc matvec_bench.hip -o matvec_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <string>
#define GL3_MV_TIMING 1
#define GL3_SS_TIMING 1
#include "gl3_decode_kernels.h"
using namespace gl3;

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

struct Shape { const char* name; int pro, epi, rows, k; };

template <int PRO, int EPI, int NPW = 4>
static float run(const MatvecArgs& a, int wgs, size_t smem, int iters, std::vector<uint8_t*>& wbufs, std::vector<uint8_t*>& w2bufs) {
    CK(hipFuncSetAttribute((const void*)matvec_q8t_kernel<PRO, EPI, true, NPW>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    MatvecArgs b = a;
    for (int i = 0; i < 3; ++i) { b.w = wbufs[i % wbufs.size()]; b.w2 = w2bufs.empty() ? nullptr : w2bufs[i % w2bufs.size()];
        hipLaunchKernelGGL((matvec_q8t_kernel<PRO, EPI, true, NPW>), dim3(wgs), dim3(mv_threads(NPW)), smem, 0, b); }
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < iters; ++i) { b.w = wbufs[i % wbufs.size()]; b.w2 = w2bufs.empty() ? nullptr : w2bufs[i % w2bufs.size()];
        hipLaunchKernelGGL((matvec_q8t_kernel<PRO, EPI, true, NPW>), dim3(wgs), dim3(mv_threads(NPW)), smem, 0, b); }
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / iters * 1e3f;
}

int main(int argc, char** argv) {
    const Shape shapes[] = {
        {"qkv    8B", PRO_RMS, EPI_STORE, 6144, 4096}, {"wo     8B", PRO_QUANT, EPI_RESID, 4096, 4096},
        {"gateup 8B", PRO_RMS, EPI_SWIGLU, 14336, 4096}, {"down   8B", PRO_QUANT, EPI_RESID, 4096, 14336},
        {"logits 8B", PRO_RMS, EPI_STORE, 128256, 4096},
        {"qkv    1B", PRO_RMS, EPI_STORE, 3072, 2048}, {"wo     1B", PRO_QUANT, EPI_RESID, 2048, 2048},
        {"gateup 1B", PRO_RMS, EPI_SWIGLU, 8192, 2048}, {"down   1B", PRO_QUANT, EPI_RESID, 2048, 8192},
    };
    std::vector<int> wg_list = {256, 384, 512, 768, 1024};
    if (argc > 1) { wg_list.clear(); for (int i = 1; i < argc; ++i) wg_list.push_back(atoi(argv[i])); }
    float *x, *nw, *out;
    CK(hipMalloc(&x, 16384 * 4)); CK(hipMalloc(&nw, 16384 * 4)); CK(hipMalloc(&out, 131072 * 4));
    std::vector<float> hx(16384);
    for (auto& v : hx) v = (rand() / (float)RAND_MAX - 0.5f);
    CK(hipMemcpy(x, hx.data(), 16384 * 4, hipMemcpyHostToDevice));
    for (auto& v : hx) v = 1.0f + 0.02f * (rand() / (float)RAND_MAX - 0.5f);
    CK(hipMemcpy(nw, hx.data(), 16384 * 4, hipMemcpyHostToDevice));
    for (const Shape& sh : shapes) {
        MatvecArgs a{};
        a.rows = sh.rows; a.k = sh.k; a.ng = (sh.k / 32 + 3) / 4; a.nstrips = (sh.rows + 15) / 16;
        a.x = x; a.norm_w = nw; a.eps = 1e-5f; a.out = out; a.resid_in = sh.epi == EPI_RESID ? out : nullptr;
        const size_t bytes = (size_t)a.nstrips * a.ng * TILE_BYTES;
        const int nm = sh.epi == EPI_SWIGLU ? 2 : 1;
        // enough distinct buffers to defeat the 256 MiB infinity cache
        const int nbuf = getenv("MB_WARM") ? 1 : (int)std::max<size_t>(2, (600ull << 20) / (bytes * nm) + 1);   // MB_WARM=1: same buffer every launch (Infinity-Cache resident)
        std::vector<uint8_t*> wb, w2b;
        for (int i = 0; i < nbuf; ++i) { uint8_t* p; CK(hipMalloc(&p, bytes)); CK(hipMemset(p, 0x11 + i, bytes)); wb.push_back(p);
            if (nm == 2) { CK(hipMalloc(&p, bytes)); CK(hipMemset(p, 0x23 + i, bytes)); w2b.push_back(p); } }
        const size_t smem = (size_t)a.ng * 4 * 32 + (size_t)a.ng * 4 * 4 + (sh.pro == PRO_RMS ? (size_t)(a.k + 32) * 4 : 0) + (size_t)2 * nm * a.ng * 64 * 4 + 128;
        const double algo = (double)sh.rows * (sh.k / 32) * 34 * nm + sh.k * 4 + sh.rows * 4;
        for (int wgs : wg_list) {
            const int g = std::min(wgs, a.nstrips);
            const int iters = 200;
            float us;
            const bool wide = getenv("MB_WIDE") != nullptr;     // 8 producer wavefronts, one workgroup per CU, for every shape
            if (wide && sh.pro == PRO_RMS && sh.epi == EPI_STORE) us = run<PRO_RMS, EPI_STORE, 8>(a, std::min(g, 256), smem, iters, wb, w2b);
            else if (wide && sh.pro == PRO_RMS) us = run<PRO_RMS, EPI_SWIGLU, 8>(a, std::min(g, 256), smem, iters, wb, w2b);
            else if (sh.pro == PRO_RMS && sh.epi == EPI_STORE) us = run<PRO_RMS, EPI_STORE>(a, g, smem, iters, wb, w2b);
            else if (sh.pro == PRO_QUANT && a.nstrips <= 256) us = run<PRO_QUANT, EPI_RESID, 8>(a, g, smem, iters, wb, w2b);
            else if (sh.pro == PRO_QUANT) us = run<PRO_QUANT, EPI_RESID>(a, g, smem, iters, wb, w2b);
            else us = run<PRO_RMS, EPI_SWIGLU>(a, g, smem, iters, wb, w2b);
            { long long st[32]; hipMemcpyFromSymbol(st, HIP_SYMBOL(gl3::gl3_mv_stamp), sizeof(st)); printf("   WG0 cycles: xload %lld sumsq %lld quant %lld | prod0 %lld bar %lld | chain0 wait->start %lld chain %lld\n", st[1]-st[0], st[2]-st[1], st[3]-st[2], st[4]-st[3], st[5]-st[4], st[16]-st[3], st[17]-st[16]); }
            if (sh.pro == PRO_RMS) { long long ss[16]; hipMemcpyFromSymbol(ss, HIP_SYMBOL(gl3::gl3_ss_stamp), sizeof(ss)); printf("   exact-sum phases (cycles): predictor %lld | translations %lld | scan %lld | lists %lld | replay-gather %lld | replay %lld | bcast %lld\n", ss[1]-ss[0], ss[2]-ss[1], ss[3]-ss[2], ss[4]-ss[3], ss[5]-ss[4], ss[6]-ss[5], ss[7]-ss[6]); }
            printf("%s rows %6d k %5d wgs %4d : %8.2f us  %7.1f GB/s (%.1f%% of 8 TB/s)\n", sh.name, sh.rows, sh.k, g, us, algo / us / 1e3, algo / us / 1e3 / 80.0);
        }
        for (auto p : wb) hipFree(p);
        for (auto p : w2b) hipFree(p);
    }
    return 0;
}
