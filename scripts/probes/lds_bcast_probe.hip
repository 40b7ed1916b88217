// LDS read issue cost on gfx950: ds_read_b128 / b64 / b32, per-lane distinct (conflict-free) vs all lanes one address
// (broadcast).  4 wavefronts per CU hammer the LDS; cycles per instruction per CU.  (scripts/probes: measurement only)
#include <hip/hip_runtime.h>
#include <cstdio>
#define N 2048
template <int MODE>
__global__ void k(float* out, long long* cyc, int zero) {
    __shared__ __attribute__((aligned(16))) float lds[8192];
    for (int i = threadIdx.x; i < 8192; i += blockDim.x) lds[i] = i;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    float4 acc = {0, 0, 0, 0};
    int off = MODE == 0 ? lane * 4 : MODE == 1 ? zero : MODE == 2 ? lane * 2 : MODE == 3 ? zero : MODE == 4 ? lane : zero;
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < N; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int o = (off + 256 * u + (it & 7) * 4) & 8191;
            if (MODE < 2) { float4 v = *reinterpret_cast<const float4*>(lds + o); acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w; }
            else if (MODE < 4) { float2 v = *reinterpret_cast<const float2*>(lds + o); acc.x += v.x; acc.y += v.y; }
            else { acc.x += lds[o]; }
        }
    }
    long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc.x + acc.y + acc.z + acc.w;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
template <int MODE> void run(const char* name, int threads) {
    float* out; long long* cyc; hipMalloc(&out, 1 << 22); hipMalloc(&cyc, 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<256, threads>>>(out, cyc, 0);
    hipEventRecord(e0); k<MODE><<<256, threads>>>(out, cyc, 0); hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double ns_per_instr_cu = ms * 1e6 / ((double)N * 8 * (threads / 64));
    printf("%-22s threads %4d: %.2f ns per wave-instruction per CU (~%.1f cycles @2.1GHz)\n", name, threads, ns_per_instr_cu, ns_per_instr_cu * 2.1);
}
int main() {
    for (int th : {256, 512}) {
        run<0>("b128 distinct", th); run<1>("b128 broadcast", th); run<2>("b64 distinct", th); run<3>("b64 broadcast", th);
        run<4>("b32 distinct", th); run<5>("b32 broadcast", th);
    }
    return 0;
}
