// attn_long_probe.hip — the long-context decode attention kernels behind the scores (attn_exp_kernel, attn_sum_kernel, attn_pv_kernel; gl3_decode_kernels.h) alone on
// synthetic scores / V of the 8B shape: time per launch, and where the chain wavefront and one helper wavefront of workgroup 0 spend it
// (clock64 cycles: waiting at the tile barrier vs working).
#define GL3_MV_TIMING 1
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <cmath>
#include "gl3_decode_kernels.h"
using namespace gl3;
int main() {
    const int H = 32, KVH = 8, hs = 128, kvd = KVH * hs;
    for (int n : {1024, 4096, 16384}) {
        const int ctx = n + 8, stride = (ctx + 3) & ~3;
        std::vector<float> att((size_t)H * stride + PVT), v((size_t)(ctx + PVT) * kvd);
        for (auto& x : att) x = rand() / (float)RAND_MAX * 8 - 4;
        for (auto& x : v) x = rand() / (float)RAND_MAX - 0.5f;
        float *datt, *datt0, *dv, *dxb; int* ddyn;
        hipMalloc(&datt, att.size() * 4); hipMalloc(&datt0, att.size() * 4); hipMalloc(&dv, v.size() * 4); hipMalloc(&dxb, H * hs * 4); hipMalloc(&ddyn, 8);
        hipMemcpy(datt0, att.data(), att.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dv, v.data(), v.size() * 4, hipMemcpyHostToDevice);
        int dyn[2] = {0, n - 1}; hipMemcpy(ddyn, dyn, 8, hipMemcpyHostToDevice);
        AttnArgs a{}; a.vcache = dv; a.att = datt; a.xb = dxb; a.dyn = ddyn; a.n_heads = H; a.n_kv_heads = KVH; a.hs = hs; a.kv_dim = kvd; a.ctx = ctx;
        a.att_stride = stride; float* datt_t; hipMalloc(&datt_t, attn_att_t_floats(KVH, H / KVH, stride) * 4); hipMemset(datt_t, 0, attn_att_t_floats(KVH, H / KVH, stride) * 4); a.att_t = datt_t;
        const int nts = (ctx + ATT_TT - 1) / ATT_TT; std::vector<float> tm((size_t)H * nts, -INFINITY);
        for (int h = 0; h < H; ++h) for (int i = 0; i < n; ++i) tm[(size_t)h * nts + i / ATT_TT] = std::max(tm[(size_t)h * nts + i / ATT_TT], att[(size_t)h * stride + i]);
        float *dtm, *dsum; hipMalloc(&dtm, tm.size() * 4); hipMalloc(&dsum, H * 4); hipMemcpy(dtm, tm.data(), tm.size() * 4, hipMemcpyHostToDevice); a.tmax = dtm; a.sums = dsum;
        hipFuncSetAttribute((const void*)attn_sum_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)attn_sum_smem());
        hipFuncSetAttribute((const void*)attn_pv_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)attn_pv_smem());
        // scores: random K cache / q of the 8B shape (RoPE table of ones and zeros: the rotation is not what is timed)
        std::vector<float> kc((size_t)(ctx + PVT) * kvd), qkvh((size_t)H * hs + 2 * kvd), cr((size_t)ctx * hs / 2, 1.f), ci((size_t)ctx * hs / 2, 0.f);
        for (auto& x : kc) x = rand() / (float)RAND_MAX - 0.5f;
        for (auto& x : qkvh) x = rand() / (float)RAND_MAX - 0.5f;
        float *dk, *dqkv, *dcr, *dci;
        hipMalloc(&dk, kc.size() * 4); hipMalloc(&dqkv, qkvh.size() * 4); hipMalloc(&dcr, cr.size() * 4); hipMalloc(&dci, ci.size() * 4);
        hipMemcpy(dk, kc.data(), kc.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dqkv, qkvh.data(), qkvh.size() * 4, hipMemcpyHostToDevice);
        hipMemcpy(dcr, cr.data(), cr.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dci, ci.data(), ci.size() * 4, hipMemcpyHostToDevice);
        a.kcache = dk; a.qkv = dqkv; a.rope_cr = dcr; a.rope_ci = dci; a.q_dim = H * hs; a.eps = 1e-5f;
        {
            const int kvmul = H / KVH;
            const size_t sml = ((size_t)kvmul * hs + 2 * (size_t)ATT_TT * (hs + 4) + 2 * hs) * 4, sm1 = ((size_t)kvmul * hs + (size_t)ATT_TT * (hs + 4) + hs) * 4;
            hipFuncSetAttribute((const void*)attn_scores_loop_kernel<128>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
            hipFuncSetAttribute((const void*)attn_scores_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
            hipEvent_t s0, s1, s2; hipEventCreate(&s0); hipEventCreate(&s1); hipEventCreate(&s2);
            float ml = 0, mo = 0;
            for (int i = 0; i < 12; ++i) {
                hipEventRecord(s0);
                attn_scores_loop_kernel<128><<<dim3(nts < SCL_WGS ? nts : SCL_WGS, KVH), 64 * (kvmul + SCL_LOADERS), sml>>>(a, nts);
                hipEventRecord(s1);
                attn_scores_kernel<<<dim3(nts, KVH), 64 * kvmul, sm1>>>(a);
                hipEventRecord(s2); hipEventSynchronize(s2);
                float m1, m2; hipEventElapsedTime(&m1, s0, s1); hipEventElapsedTime(&m2, s1, s2);
                if (i >= 2) { ml += m1; mo += m2; }
            }
            long long st[32]; hipMemcpyFromSymbol(st, HIP_SYMBOL(gl3::gl3_mv_stamp), sizeof(st));
            printf("n %5d: scores loop kernel %.2f us (one-tile kernel %.2f us) | chain wave 0 of workgroup (0, 0): %lld tiles, %lld cycles, %lld waiting at barriers | loader thread 0: "
                   "%lld cycles, storing %lld, waiting at barriers %lld\n", n, ml * 100, mo * 100, st[7], st[6], st[5], st[10], st[8], st[9]);
        }
        hipEvent_t e0, e1, e2, e3; hipEventCreate(&e0); hipEventCreate(&e1); hipEventCreate(&e2); hipEventCreate(&e3);
        float ms_s = 0, ms_p = 0, ms_e = 0;
        const int reps = 10;
        for (int i = 0; i < reps + 2; ++i) {
            hipMemcpy(datt, datt0, att.size() * 4, hipMemcpyDeviceToDevice);
            hipEventRecord(e3);
            attn_exp_kernel<<<dim3((ctx + EXP_ROW - 1) / EXP_ROW, H), 256>>>(a, nts);
            hipEventRecord(e0);
            attn_sum_kernel<<<H, 256, attn_sum_smem()>>>(a);
            hipEventRecord(e1);
            attn_pv_kernel<<<KVH * attn_pv_hq(H / KVH) * (hs / PV_COLS16), 64 * PV_WAVES, attn_pv_smem()>>>(a);
            hipEventRecord(e2); hipEventSynchronize(e2);
            float m1, m2, m3; hipEventElapsedTime(&m1, e0, e1); hipEventElapsedTime(&m2, e1, e2); hipEventElapsedTime(&m3, e3, e0);
            if (i >= 2) { ms_s += m1; ms_p += m2; ms_e += m3; }
        }
        long long st[32]; hipMemcpyFromSymbol(st, HIP_SYMBOL(gl3::gl3_mv_stamp), sizeof(st));
        const int ntiles = (n + PVT - 1) / PVT;
        printf("n %5d: exp %.2f us, sum %.2f us, pv %.2f us (%d workgroups of %d threads, cols %d) | chain wave: %lld cycles total = %.1f per timestep, %lld waiting at barriers (%.0f per tile)"
               " | helper wave 0: total %lld, barrier wait %lld, store+load issue %lld\n", n, ms_e * 1e3 / reps, ms_s * 1e3 / reps, ms_p * 1e3 / reps, KVH * attn_pv_hq(H / KVH) * (hs / PV_COLS16), 64 * PV_WAVES,
               PV_COLS16, st[1], (double)st[1] / n, st[0], (double)st[0] / ntiles, st[4], st[2], st[3]);
        hipFree(dk); hipFree(dqkv); hipFree(dcr); hipFree(dci); hipFree(dtm); hipFree(dsum); hipFree(datt_t); hipFree(datt); hipFree(datt0); hipFree(dv); hipFree(dxb); hipFree(ddyn);
    }
    return 0;
}
