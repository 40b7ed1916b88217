// Issue-rate probe for gfx950: cycles per wave64 instruction for v_mul_f32, v_pk_mul_f32, v_cvt_f32_i32 and the
// int8 32x32x32 MFMA, alone and interleaved, with 1 or 2 wavefronts per SIMD.  (scripts/probes: measurement only)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef float v2f __attribute__((ext_vector_type(2)));
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
#define N 4096
template <int MODE>
__global__ void k(float* out, long long* cyc, float seed) {
    float a[8]; v2f p[8]; v16i c = {0}; v16i c2 = {0};
    v4i af = {1, 2, 3, 4}, bf = {5, 6, 7, 8};
    for (int i = 0; i < 8; ++i) { a[i] = seed + i + threadIdx.x; p[i] = v2f{a[i], a[i] + 1}; }
    __syncthreads();
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < N; ++it) {
        if (MODE == 0) {
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(seed));
        } else if (MODE == 1) {
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(p[(i + 1) & 7]));
        } else if (MODE == 2) {
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("v_cvt_f32_i32 %0, %0" : "+v"(a[i]));
        } else if (MODE == 3) {
            c = __builtin_amdgcn_mfma_i32_32x32x32_i8(af, bf, c, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_i32_32x32x32_i8(af, bf, c2, 0, 0, 0);
        } else if (MODE == 4) {   // 2 MFMA + 32 pk
            c = __builtin_amdgcn_mfma_i32_32x32x32_i8(af, bf, c, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_i32_32x32x32_i8(af, bf, c2, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(p[(i + 1) & 7]));
        } else if (MODE == 6) {   // 2 MFMA + 64 plain mul
            c = __builtin_amdgcn_mfma_i32_32x32x32_i8(af, bf, c, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_i32_32x32x32_i8(af, bf, c2, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 8; ++r)
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(seed));
        } else if (MODE == 7) {   // 64 plain mul
#pragma unroll
            for (int r = 0; r < 8; ++r)
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(seed));
        } else if (MODE == 8) {   // interleaved: mfma, 32 mul, mfma, 32 mul
            c = __builtin_amdgcn_mfma_i32_32x32x32_i8(af, bf, c, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(seed));
            c2 = __builtin_amdgcn_mfma_i32_32x32x32_i8(af, bf, c2, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(seed));
        } else if (MODE == 5) {
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(p[(i + 1) & 7]));
        }
    }
    long long t1 = __builtin_readcyclecounter();
    float s = 0; for (int i = 0; i < 8; ++i) s += a[i] + p[i][0] + p[i][1];
    for (int i = 0; i < 16; ++i) s += c[i] + c2[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
template <int MODE> void run(const char* name, int threads, int per_iter) {
    float* out; long long* cyc; hipMalloc(&out, 1 << 20); hipMalloc(&cyc, 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<256, threads>>>(out, cyc, 1.0f);
    hipEventRecord(e0); k<MODE><<<256, threads>>>(out, cyc, 1.0f); hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long h; hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    printf("%-26s threads=%4d  ticks/iter %.1f  wall ns/iter %.2f  (%d instr/iter) -> ns/instr/wave %.3f\n", name, threads, (double)h / N, ms * 1e6 / N, per_iter, ms * 1e6 / N / per_iter);
}
int main() {
    for (int th : {256, 512, 1024}) {
        run<0>("v_mul_f32 x8", th, 8); run<1>("v_pk_mul_f32 x8", th, 8); run<5>("v_pk_add_f32 x8", th, 8); run<2>("v_cvt_f32_i32 x8", th, 8);
        run<3>("mfma_i8_32x32x32 x2", th, 2); run<4>("2 mfma + 32 pk_mul", th, 34);
        run<7>("64 mul", th, 64); run<6>("2 mfma + 64 mul", th, 66); run<8>("mfma,32mul,mfma,32mul", th, 66);
    }
    return 0;
}
