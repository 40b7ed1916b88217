// Phase stamps of attn_head_kernel (decode attention, positions < 128) for the Llama-3-8B head shape.
// hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I../../gpullama3.java_amd/csrc attn_head_probe.hip -o attn_head_probe
#define GL3_MV_TIMING 1
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "gl3_decode_kernels.h"
using namespace gl3;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
int main(int argc, char** argv) {
    const int arch = argc > 1 ? atoi(argv[1]) : 0;
    const int ctx = 648, H = 32, KVH = 8, hs = 128, kvd = KVH * hs, qd = H * hs;
    std::vector<float> qkv(qd + 2 * kvd), kc((size_t)ctx * kvd), cr((size_t)ctx * hs / 2), nw(hs, 1.0f);
    for (auto& x : qkv) x = rand() / (float)RAND_MAX - 0.5f;
    for (auto& x : kc) x = rand() / (float)RAND_MAX - 0.5f;
    for (auto& x : cr) x = rand() / (float)RAND_MAX;
    float *dqkv, *dk, *dv, *dcr, *dci, *dxb, *dnw; int* ddyn;
    CK(hipMalloc(&dqkv, qkv.size() * 4)); CK(hipMalloc(&dk, kc.size() * 4)); CK(hipMalloc(&dv, kc.size() * 4)); CK(hipMalloc(&dcr, cr.size() * 4));
    CK(hipMalloc(&dci, cr.size() * 4)); CK(hipMalloc(&dxb, qd * 4)); CK(hipMalloc(&ddyn, 16)); CK(hipMalloc(&dnw, hs * 4));
    CK(hipMemcpy(dqkv, qkv.data(), qkv.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dk, kc.data(), kc.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dv, kc.data(), kc.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dcr, cr.data(), cr.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dci, cr.data(), cr.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dnw, nw.data(), hs * 4, hipMemcpyHostToDevice));
    CK(hipFuncSetAttribute((const void*)attn_head_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
    for (int pos : {0, 31, 63, 127}) {
        int dyn[4] = {0, pos, 0, 0}; CK(hipMemcpy(ddyn, dyn, 16, hipMemcpyHostToDevice));
        AttnArgs a{};
        a.qkv = dqkv; a.qkv_stride = qd + 2 * kvd; a.kcache = dk; a.vcache = dv; a.rope_cr = dcr; a.rope_ci = dci; a.dyn = ddyn; a.xb = dxb; a.xb_stride = qd;
        a.qnorm = dnw; a.knorm = dnw;
        a.n_heads = H; a.n_kv_heads = KVH; a.hs = hs; a.q_dim = qd; a.kv_dim = kvd; a.ctx = ctx; a.eps = 1e-5f; a.arch = arch; a.group = 1;
        const size_t sm = attn_head_smem(hs, 1);
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(attn_head_kernel, dim3(H, 1), dim3(256), sm, 0, a);
        CK(hipEventRecord(e0)); for (int i = 0; i < 50; ++i) hipLaunchKernelGGL(attn_head_kernel, dim3(H, 1), dim3(256), sm, 0, a);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        long long st[32]; CK(hipMemcpyFromSymbol(st, HIP_SYMBOL(gl3::gl3_mv_stamp), sizeof(st)));
        printf("arch %d pos %3d: %.2f us/launch; WG0 phases (10 ns ticks): load+stage %lld, norm+rope %lld, scores %lld, max+exp %lld, sum+div %lld, pv %lld\n", arch, pos, ms * 20,
               st[1] - st[0], st[2] - st[1], st[3] - st[2], st[4] - st[3], st[5] - st[4], st[6] - st[5]);
    }
    return 0;
}
