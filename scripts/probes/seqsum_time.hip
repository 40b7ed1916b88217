// times exact_sumsq_lds vs the naive chain inside one workgroup (clock64), n = 4096 / 2048
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <cmath>
#define GL3_SS_TIMING 1
#include "gl3_seqsum.h"
using namespace gl3;
__global__ __launch_bounds__(256) void k(const float* x, int n, float* out, long long* cyc, int mode) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    float* xf = (float*)smem; uint8_t* scratch = smem + (n + 32) * 4;
    for (int i = threadIdx.x; i < n + 32; i += 256) xf[i] = i < n ? x[i] : 0.f;
    __syncthreads();
    long long t0 = wall_clock64();
    float s;
    if (mode == 0) { BlockBarrier bb; s = exact_sumsq_lds(xf, n, scratch, threadIdx.x, bb); }
    else { s = 0; if (threadIdx.x < 64) s = naive_sumsq_lds(xf, 0, n, 0.f); __syncthreads(); }
    long long t1 = wall_clock64();
    if (threadIdx.x == 0) { *out = s; *cyc = t1 - t0; }
}
int main() {
    for (int n : {2048, 4096}) {
        std::vector<float> x(n); srand(3);
        for (auto& v : x) { float u1 = rand() / (float)RAND_MAX + 1e-9f, u2 = rand() / (float)RAND_MAX; v = sqrtf(-2 * logf(u1)) * cosf(6.2831853f * u2); }
        float* dx; float* dout; long long* dc;
        hipMalloc(&dx, n * 4); hipMalloc(&dout, 4); hipMalloc(&dc, 8);
        hipMemcpy(dx, x.data(), n * 4, hipMemcpyHostToDevice);
        for (int mode = 0; mode < 2; ++mode) {
            float r = 0; long long c = 0;
            for (int rep = 0; rep < 3; ++rep) k<<<1, 256, (n + 32) * 4 + 20000>>>(dx, n, dout, dc, mode);
            hipMemcpy(&r, dout, 4, hipMemcpyDeviceToHost); hipMemcpy(&c, dc, 8, hipMemcpyDeviceToHost);
            if (mode == 0) { long long st[16]; hipMemcpyFromSymbol(st, HIP_SYMBOL(gl3::gl3_ss_stamp), sizeof(st)); printf("   phase cycles:"); for (int i = 1; i < 8; ++i) printf(" %lld", st[i] - st[i-1]); printf("\n"); }
            printf("n %d %s: sum %.9g  wall_clock ticks %lld (100 MHz -> %.2f us)\n", n, mode ? "naive" : "exact-parallel", r, c, c / 100.0);
        }
    }
    return 0;
}
