#define GL3_MV_TIMING 1
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "gl3_decode_kernels.h"
using namespace gl3;
int main() {
    const int ctx = 648, H = 32, KVH = 8, hs = 128, kvd = KVH * hs;
    for (int pos : {63, 639}) {
        std::vector<float> att(H * ctx), v(ctx * kvd);
        for (auto& x : att) x = rand() / (float)RAND_MAX * 4 - 2;
        for (auto& x : v) x = rand() / (float)RAND_MAX - 0.5f;
        float *datt, *dv, *dxb; int* ddyn;
        hipMalloc(&datt, att.size() * 4); hipMalloc(&dv, v.size() * 4); hipMalloc(&dxb, H * hs * 4); hipMalloc(&ddyn, 8);
        hipMemcpy(datt, att.data(), att.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dv, v.data(), v.size() * 4, hipMemcpyHostToDevice);
        int dyn[2] = {0, pos}; hipMemcpy(ddyn, dyn, 8, hipMemcpyHostToDevice);
        AttnArgs a{}; a.vcache = dv; a.att = datt; a.xb = dxb; a.dyn = ddyn; a.n_heads = H; a.n_kv_heads = KVH; a.hs = hs; a.kv_dim = kvd; a.ctx = ctx;
        const size_t sm2 = ((size_t)((ctx + 3) & ~3) + (size_t)ctx * PV_COLS) * 4;
        hipFuncSetAttribute((const void*)attn_softmax_pv_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        for (int i = 0; i < 3; ++i) attn_softmax_pv_kernel<<<H * (hs / PV_COLS), 256, sm2>>>(a);
        hipEventRecord(e0); for (int i = 0; i < 50; ++i) attn_softmax_pv_kernel<<<H * (hs / PV_COLS), 256, sm2>>>(a); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        long long st[32]; hipMemcpyFromSymbol(st, HIP_SYMBOL(gl3::gl3_mv_stamp), sizeof(st));
        printf("pos %d: %.2f us/launch; WG0 phases (10 ns ticks): stage-issue %lld, max %lld, exp %lld, sum %lld, div %lld, pv %lld\n", pos, ms * 20,
               st[1] - st[0], st[2] - st[1], st[3] - st[2], st[4] - st[3], st[5] - st[4], st[6] - st[5]);
    }
    return 0;
}
