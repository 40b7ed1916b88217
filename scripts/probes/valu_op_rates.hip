// Issue rate of single VALU opcodes on gfx950 (8 independent dependency chains per wavefront, 1 / 2 / 4 wavefronts per SIMD):
// which of the instructions of the Q4_0 / Q8_0 unpack paths are full rate.  (scripts/probes: measurement only)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define N 4096
#define OPS(X) \
    X(0, "v_mul_f32 %0, %0, %1") X(1, "v_add_f32 %0, %0, %1") X(2, "v_fma_f32 %0, %0, %1, %1") \
    X(3, "v_fma_mix_f32 %0, %1, %0, %1 op_sel_hi:[0,1,0]") X(4, "v_pk_add_f16 %0, %0, %1") X(5, "v_and_or_b32 %0, %0, %1, %1") \
    X(6, "v_and_b32 %0, %0, %1") X(7, "v_lshrrev_b32 %0, 4, %0") X(8, "v_add_u32 %0, %0, %1") X(9, "v_cvt_f32_i32 %0, %0") \
    X(10, "v_cvt_f32_ubyte0 %0, %0") X(11, "v_bfe_u32 %0, %0, 4, 4") X(12, "v_cvt_f32_f16 %0, %0") X(13, "v_perm_b32 %0, %0, %1, %1") \
    X(14, "v_cvt_f32_ubyte2 %0, %0") X(15, "v_fma_mix_f32 %0, %0, %1, %1") X(16, "v_mad_u32_u24 %0, %0, %1, %1") X(17, "v_dot4_i32_i8 %0, %0, %1, %0") \
    X(18, "v_cvt_f32_i32_sdwa %0, sext(%0) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1") X(19, "v_sub_f32 %0, %0, %1") \
    X(20, "v_fmac_f32_e32 %0, %1, %1") X(21, "v_fma_f32 %0, %1, %1, %0") X(22, "v_fmac_f32_e64 %0, %1, %1")
template <int MODE>
__global__ void k(float* out, long long* cyc, float seed) {
    float a[8];
    for (int i = 0; i < 8; ++i) a[i] = seed + i + threadIdx.x;
    __syncthreads();
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < N; ++it) {
#define X(M, S) if (MODE == M) { _Pragma("unroll") for (int i = 0; i < 8; ++i) asm volatile(S : "+v"(a[i]) : "v"(seed)); }
        OPS(X)
#undef X
    }
    long long t1 = __builtin_readcyclecounter();
    float s = 0; for (int i = 0; i < 8; ++i) s += a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
template <int MODE> void run(const char* name) {
    float* out; long long* cyc; hipMalloc(&out, 1 << 20); hipMalloc(&cyc, 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    printf("%-90s", name);
    for (int threads : {256, 512, 1024}) {
        k<MODE><<<256, threads>>>(out, cyc, 1.0f);
        hipEventRecord(e0); k<MODE><<<256, threads>>>(out, cyc, 1.0f); hipEventRecord(e1); hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("  %d waves/SIMD: %.3f ns/instr/SIMD", threads / 256, ms * 1e6 / N / 8 / (threads / 256));
    }
    printf("\n");
    hipFree(out); hipFree(cyc);
}
int main() {
#define X(M, S) run<M>(S);
    OPS(X)
#undef X
    return 0;
}
