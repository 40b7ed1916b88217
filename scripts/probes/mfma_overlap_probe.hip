// mfma_overlap_probe.hip — two questions behind the Q4_0 / Q8_0-f32act batched GEMM (gl3_prefill_vl.h gemm_vlq_mfma_kernel):
//  1. Does an f32-input MFMA (v_mfma_f32_16x16x1_4b_f32) execute BESIDE v_pk_*_f32 VALU work of the same SIMD, or do the two
//     share the FP32 datapath?  Cycles per iteration of {4 MFMA}, {32 v_pk_fma_f32}, {both}, one and two wavefronts per SIMD;
//     the same with a bf16 MFMA (v_mfma_f32_32x32x16_bf16, 8 passes).
//  2. Is a bf16 MFMA exact enough to deliver fl(x * q) for an f32 x split into three bf16 pieces (x = xh + xm + xl exactly)
//     and a small integer q: D = xh q + xm q + xl q summed inside ONE v_mfma_f32_32x32x16_bf16 (13 of 16 k slots zero)?
//     The exact sum has up to 28 significant bits; fl(x q) needs it rounded once, to nearest even.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef __bf16 v8bf __attribute__((ext_vector_type(8)));
typedef short v8s __attribute__((ext_vector_type(8)));

template <int MODE, int KIND, bool SCALAR = false>      // MODE bit 0: MFMA, bit 1: VALU; KIND 0: f32 16x16x1_4b, 1: bf16 32x32x16
__global__ __launch_bounds__(512) void rate_kernel(float* out, long long* cyc, int iters) {
    const int lane = threadIdx.x & 63;
    float a = 1.0f + lane * 1e-3f, b = 0.5f + lane * 2e-3f;
    v8bf ab, bb;
    for (int i = 0; i < 8; ++i) { ab[i] = (__bf16)(1.0f + i); bb[i] = (__bf16)(0.5f * i); }
    const v16f z = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    v16f P0 = z, P1 = z, P2 = z, P3 = z;
    v2f acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = (v2f){1.0f + i, 2.0f + i};
    const v2f m = {1.0000001f, 0.9999999f};
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
        // the GEMM's pattern: one MFMA, then the 8 packed VALU instructions of the previous tile, four times per iteration
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (MODE & 1) {
                v16f p;
                if (KIND == 0) p = __builtin_amdgcn_mfma_f32_16x16x1f32(r & 1 ? a : b, r & 2 ? a : b, z, 0, 0, 0);
                else if (KIND == 1) p = __builtin_amdgcn_mfma_f32_32x32x16_bf16(r & 1 ? ab : bb, r & 2 ? ab : bb, z, 0, 0, 0);
                else {
                    typedef int v4i __attribute__((ext_vector_type(4)));
                    typedef int v16i __attribute__((ext_vector_type(16)));
                    const v4i ia = {lane, lane + r, 3, 4}, ib = {r, 2, lane, 7};
                    const v16i zi = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
                    p = __builtin_bit_cast(v16f, __builtin_amdgcn_mfma_i32_32x32x32_i8(ia, ib, zi, 0, 0, 0));
                }
                if (r == 0) P0 = p; else if (r == 1) P1 = p; else if (r == 2) P2 = p; else P3 = p;
            }
            __builtin_amdgcn_sched_barrier(0);
            if ((MODE & 2) && SCALAR) {
                float* f = reinterpret_cast<float*>(acc);
                asm volatile("v_fma_f32 %0, %0, %8, %0\n\tv_fma_f32 %1, %1, %8, %1\n\tv_fma_f32 %2, %2, %8, %2\n\tv_fma_f32 %3, %3, %8, %3\n\t"
                             "v_fma_f32 %4, %4, %8, %4\n\tv_fma_f32 %5, %5, %8, %5\n\tv_fma_f32 %6, %6, %8, %6\n\tv_fma_f32 %7, %7, %8, %7"
                             : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]), "+v"(f[4]), "+v"(f[5]), "+v"(f[6]), "+v"(f[7]) : "v"(m[0]));
                asm volatile("v_fma_f32 %0, %0, %8, %0\n\tv_fma_f32 %1, %1, %8, %1\n\tv_fma_f32 %2, %2, %8, %2\n\tv_fma_f32 %3, %3, %8, %3\n\t"
                             "v_fma_f32 %4, %4, %8, %4\n\tv_fma_f32 %5, %5, %8, %5\n\tv_fma_f32 %6, %6, %8, %6\n\tv_fma_f32 %7, %7, %8, %7"
                             : "+v"(f[8]), "+v"(f[9]), "+v"(f[10]), "+v"(f[11]), "+v"(f[12]), "+v"(f[13]), "+v"(f[14]), "+v"(f[15]) : "v"(m[1]));
            } else if (MODE & 2)
                asm volatile("v_pk_fma_f32 %0, %0, %8, %0\n\tv_pk_fma_f32 %1, %1, %8, %1\n\tv_pk_fma_f32 %2, %2, %8, %2\n\tv_pk_fma_f32 %3, %3, %8, %3\n\t"
                             "v_pk_fma_f32 %4, %4, %8, %4\n\tv_pk_fma_f32 %5, %5, %8, %5\n\tv_pk_fma_f32 %6, %6, %8, %6\n\tv_pk_fma_f32 %7, %7, %8, %7"
                             : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(acc[4]), "+v"(acc[5]), "+v"(acc[6]), "+v"(acc[7]) : "v"(m));
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const long long t1 = clock64();
    float s = P0[0] + P1[1] + P2[2] + P3[3];
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

// exactness: D[i][j] = sum_k A[i][k] B[k][j] with A[i][k0..k2] = bf16 pieces of x_i, B[k0..k2][j] = q_j
__global__ void exact_kernel(const float* x, const float* q, float* D, int k0, int k1, int k2) {
    const int lane = threadIdx.x, i = lane & 31, kg = lane >> 5;
    const float xv = x[i];
    const uint32_t u = __builtin_bit_cast(uint32_t, xv);
    const float xh = __builtin_bit_cast(float, u & 0xFFFF0000u);
    const float r1 = xv - xh;
    const float xm = __builtin_bit_cast(float, __builtin_bit_cast(uint32_t, r1) & 0xFFFF0000u);
    const float xl = r1 - xm;                                   // <= 8 significant bits left: exact as bf16
    v8s av = {0, 0, 0, 0, 0, 0, 0, 0}, bv = {0, 0, 0, 0, 0, 0, 0, 0};
    const short qb = (short)(__builtin_bit_cast(uint32_t, q[i]) >> 16);
    const int ks[3] = {k0, k1, k2};
    const float pc[3] = {xh, xm, xl};
    for (int p = 0; p < 3; ++p)
        if ((ks[p] >> 3) == kg) { av[ks[p] & 7] = (short)(__builtin_bit_cast(uint32_t, pc[p]) >> 16); bv[ks[p] & 7] = qb; }
    const v16f z = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    const v16f d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(v8bf, av), __builtin_bit_cast(v8bf, bv), z, 0, 0, 0);
    for (int r = 0; r < 16; ++r) D[((r & 3) + 8 * (r >> 2) + 4 * kg) * 32 + i] = d[r];       // D[row = A index][col = B index]: lane = column
}

template <int MODE, int KIND, bool SCALAR = false>
static void run_rate(const char* name, int wgs, float* out, long long* cyc) {
    const int iters = 20000;
    rate_kernel<MODE, KIND, SCALAR><<<wgs, 512>>>(out, cyc, iters);
    hipDeviceSynchronize();
    rate_kernel<MODE, KIND, SCALAR><<<wgs, 512>>>(out, cyc, iters);
    hipDeviceSynchronize();
    long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0); rate_kernel<MODE, KIND, SCALAR><<<wgs, 512>>>(out, cyc, iters); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    printf("%-34s %7.1f clock64 ticks, %7.1f ns per iteration of 4 x (MFMA, 8 v_pk_fma_f32); 2 wavefronts per SIMD\n", name, (double)c / iters, ms * 1e6 / iters);
}

int main() {
    float* out; long long* cyc;
    hipMalloc(&out, 1 << 22); hipMalloc(&cyc, 8);
    run_rate<1, 0>("f32 MFMA 16x16x1_4b alone", 256, out, cyc);
    run_rate<2, 0>("v_pk_fma_f32 alone", 256, out, cyc);
    run_rate<3, 0>("f32 MFMA + v_pk_fma_f32", 256, out, cyc);
    run_rate<2, 0, true>("16 v_fma_f32 alone", 256, out, cyc);
    run_rate<3, 0, true>("f32 MFMA + 16 v_fma_f32", 256, out, cyc);
    run_rate<1, 2>("int8 MFMA 32x32x32 alone", 256, out, cyc);
    run_rate<3, 2>("int8 MFMA + v_pk_fma_f32", 256, out, cyc);
    run_rate<3, 2, true>("int8 MFMA + 16 v_fma_f32", 256, out, cyc);
    run_rate<1, 1>("bf16 MFMA 32x32x16 alone", 256, out, cyc);
    run_rate<3, 1, true>("bf16 MFMA + 16 v_fma_f32", 256, out, cyc);
    run_rate<3, 1>("bf16 MFMA + v_pk_fma_f32", 256, out, cyc);
    // ---- exactness
    float *dx, *dq, *dD;
    hipMalloc(&dx, 128); hipMalloc(&dq, 128); hipMalloc(&dD, 4096);
    float hx[32], hq[32], hD[1024];
    srand(7);
    const int kpos[4][3] = {{0, 1, 2}, {2, 1, 0}, {0, 5, 11}, {15, 8, 3}};
    for (int kp = 0; kp < 4; ++kp) {
        long total = 0, bad = 0, bad_rz = 0; double worst = 0;
        for (int trial = 0; trial < 400; ++trial) {
            for (int i = 0; i < 32; ++i) {
                uint32_t m = ((uint32_t)rand() << 8 ^ (uint32_t)rand()) & 0x7FFFFFu;
                uint32_t e = 100 + rand() % 60, sg = rand() & 1;
                uint32_t u = (sg << 31) | (e << 23) | m;
                memcpy(&hx[i], &u, 4);
                hq[i] = (float)((trial & 1) ? (rand() % 16) - 8 : (rand() % 256) - 128);
            }
            hipMemcpy(dx, hx, 128, hipMemcpyHostToDevice); hipMemcpy(dq, hq, 128, hipMemcpyHostToDevice);
            exact_kernel<<<1, 64>>>(dx, dq, dD, kpos[kp][0], kpos[kp][1], kpos[kp][2]);
            hipMemcpy(hD, dD, 4096, hipMemcpyDeviceToHost);
            for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) {
                volatile float want = hx[i] * hq[j];
                const float got = hD[i * 32 + j];
                ++total;
                if (memcmp((const void*)&want, &got, 4) != 0 && !(want == 0.f && got == 0.f)) {
                    ++bad;
                    const double e = fabs((double)got - (double)hx[i] * hq[j]) / fabs((double)want);
                    if (e > worst) worst = e;
                }
            }
        }
        printf("bf16 split product, pieces at k = %2d %2d %2d: %ld of %ld differ from fl(x*q); worst relative error of a differing one %.3g\n",
               kpos[kp][0], kpos[kp][1], kpos[kp][2], bad, total, worst);
    }
    return 0;
}
