// Does v_mfma_f32_16x16x4_f32 round each product a*b to f32 before adding (mul-then-add, k ascending), or does it
// fuse (fma), or something else?  Compares the device result with host models.  (scripts/probes: measurement only)
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
typedef float v4f __attribute__((ext_vector_type(4)));
__global__ void k(const float* A, const float* B, const float* C, float* D) {
    const int l = threadIdx.x;
    v4f c;
    for (int r = 0; r < 4; ++r) c[r] = C[(4 * (l >> 4) + r) * 16 + (l & 15)];
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(A[(l & 15) * 4 + (l >> 4)], B[(l >> 4) * 16 + (l & 15)], c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) D[(4 * (l >> 4) + r) * 16 + (l & 15)] = c[r];
}
int main() {
    float hA[64], hB[64], hC[256], hD[256];
    float *A, *B, *C, *D;
    hipMalloc(&A, 256); hipMalloc(&B, 256); hipMalloc(&C, 1024); hipMalloc(&D, 1024);
    int n_mul_add = 0, n_fma = 0, n_mul_add_rev = 0, n_dbl = 0, total = 0;
    srand(1);
    for (int trial = 0; trial < 200; ++trial) {
        for (int i = 0; i < 64; ++i) { hA[i] = (float)rand() / RAND_MAX * 2 - 1; hB[i] = (float)rand() / RAND_MAX * 2 - 1; }
        for (int i = 0; i < 256; ++i) hC[i] = ((float)rand() / RAND_MAX * 2 - 1) * (trial & 1 ? 1.f : 8.f);
        hipMemcpy(A, hA, 256, hipMemcpyHostToDevice); hipMemcpy(B, hB, 256, hipMemcpyHostToDevice); hipMemcpy(C, hC, 1024, hipMemcpyHostToDevice);
        k<<<1, 64>>>(A, B, C, D);
        hipMemcpy(hD, D, 1024, hipMemcpyDeviceToHost);
        for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) {
            volatile float s1 = hC[i * 16 + j]; float s2 = hC[i * 16 + j]; volatile float s3 = hC[i * 16 + j]; double s4 = hC[i * 16 + j];
            for (int kk = 0; kk < 4; ++kk) { volatile float p = hA[i * 4 + kk] * hB[kk * 16 + j]; s1 = s1 + p; s2 = fmaf(hA[i * 4 + kk], hB[kk * 16 + j], s2); s4 += (double)hA[i * 4 + kk] * hB[kk * 16 + j]; }
            for (int kk = 3; kk >= 0; --kk) { volatile float p = hA[i * 4 + kk] * hB[kk * 16 + j]; s3 = s3 + p; }
            const float d = hD[i * 16 + j];
            float f1 = s1, f3 = s3, f4 = (float)s4;
            n_mul_add += !memcmp(&d, &f1, 4); n_fma += !memcmp(&d, &s2, 4); n_mul_add_rev += !memcmp(&d, &f3, 4); n_dbl += !memcmp(&d, &f4, 4); ++total;
        }
    }
    printf("total %d: mul-then-add(k asc) %d, fma chain %d, mul-then-add(k desc) %d, exact-then-round %d\n", total, n_mul_add, n_fma, n_mul_add_rev, n_dbl);
    return 0;
}
