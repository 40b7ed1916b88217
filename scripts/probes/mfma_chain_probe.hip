// Probe: is v_mfma_f32_16x16x4_f32 with B = 1.0 bit-identical to a sequential f32 add chain (k ascending)?
// Also times dependent VALU add chains and LDS-broadcast chains.  Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <cmath>
typedef float v4f __attribute__((ext_vector_type(4)));

__global__ void chain_mfma(const float* __restrict__ p /*[16 rows][nb]*/, int nb, float* __restrict__ out /*[16]*/) {
    const int lane = threadIdx.x;
    v4f acc = {0.f, 0.f, 0.f, 0.f};
    for (int g = 0; g < nb / 4; ++g) {
        const float a = p[(lane & 15) * nb + 4 * g + (lane >> 4)];
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, 1.0f, acc, 0, 0, 0);
    }
    if ((lane & 15) == 0) for (int r = 0; r < 4; ++r) out[4 * (lane >> 4) + r] = acc[r];
}

__global__ void chain_valu(const float* __restrict__ x, int n, float* out, long long* cycles) {
    extern __shared__ float xs[];
    for (int i = threadIdx.x; i < n; i += blockDim.x) xs[i] = x[i];
    __syncthreads();
    long long t0 = clock64();
    float ss = 0.f;
    for (int i = 0; i < n; i += 4) {
        const float4 v = *reinterpret_cast<const float4*>(&xs[i]);
        ss = ss + v.x * v.x; ss = ss + v.y * v.y; ss = ss + v.z * v.z; ss = ss + v.w * v.w;
    }
    long long t1 = clock64();
    if (threadIdx.x == 0) { *out = ss; *cycles = t1 - t0; }
}

int main() {
    const int nb = 448;
    std::vector<float> p(16 * nb);
    srand(1);
    for (auto& v : p) { int e = rand() % 40 - 20; v = ((rand() / (float)RAND_MAX) - 0.5f) * ldexpf(1.f, e); }
    std::vector<float> ref(16);
    for (int r = 0; r < 16; ++r) { volatile float s = 0.f; for (int b = 0; b < nb; ++b) s = s + p[r * nb + b]; ref[r] = s; }
    float *dp, *dout; long long* dc;
    hipMalloc(&dp, p.size() * 4); hipMalloc(&dout, 64 * 4); hipMalloc(&dc, 8);
    hipMemcpy(dp, p.data(), p.size() * 4, hipMemcpyHostToDevice);
    chain_mfma<<<1, 64>>>(dp, nb, dout);
    std::vector<float> got(16);
    hipMemcpy(got.data(), dout, 64, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int r = 0; r < 16; ++r) if (memcmp(&got[r], &ref[r], 4)) { ++bad; printf("row %d: mfma %.9g seq %.9g\n", r, got[r], ref[r]); }
    printf("MFMA_CHAIN_BITEXACT %s (%d/16 rows differ)\n", bad ? "NO" : "YES", bad);
    // VALU chain timing
    const int n = 4096;
    std::vector<float> x(n);
    for (auto& v : x) v = (rand() / (float)RAND_MAX) - 0.5f;
    volatile float s = 0.f; for (int i = 0; i < n; ++i) s = s + x[i] * x[i];
    float* dx; hipMalloc(&dx, n * 4); hipMemcpy(dx, x.data(), n * 4, hipMemcpyHostToDevice);
    for (int rep = 0; rep < 3; ++rep) chain_valu<<<1, 64, n * 4>>>(dx, n, dout, dc);
    float g; long long cyc; hipMemcpy(&g, dout, 4, hipMemcpyDeviceToHost); hipMemcpy(&cyc, dc, 8, hipMemcpyDeviceToHost);
    float sf = s;
    printf("VALU_CHAIN n=%d cycles=%lld (%.2f cyc/elem) bitexact=%d\n", n, cyc, (double)cyc / n, !memcmp(&g, &sf, 4));
    return 0;
}
