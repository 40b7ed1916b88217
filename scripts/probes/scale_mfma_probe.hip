// Can the per-block scale products of the batched-prefill GEMM leave the VALU?  (scripts/probes: measurement only)
//   acc = acc + float(isum) * (wScale * aScale)           (Q8_0FloatTensor.java:119, one f32 rounding per operation)
// wScale and aScale are f16 values, so s = wScale * aScale is EXACT in f32 (11 + 11 significand bits), and so is
// B * s for the conversion bias B = 12582912 = 3 * 2^22 (2 + 22 bits).  Hence
//   fl(float(isum) * s) = fma(D, s, -B s)   with D = the int8 MFMA's biased output read as f32 (= B + isum exactly),
// and both s and -B s are outer products that a 16-bit MFMA can deliver exactly:
//   s    = v_mfma_f32_32x32x16_f16  with A = {w, 0 ...}, B = {a, 0 ...}
//   -B s = v_mfma_f32_32x32x16_bf16 with the 8 terms (-2^23 | -2^22) x (w_hi | w_lo) x (a_hi | a_lo), hi = the top 8
//          significand bits, lo = the remaining <= 3 (every partial sum is a same-sign multiple of one ulp below 2^24 ulps)
// This probe checks the three identities bit for bit (f16 subnormals, zeros and the f16 maximum included) and times
// back-to-back issue of the candidate instructions.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
typedef float v16f __attribute__((ext_vector_type(16)));
typedef _Float16 v8h __attribute__((ext_vector_type(8)));
typedef _Float16 v4h __attribute__((ext_vector_type(4)));
typedef __bf16 v8b __attribute__((ext_vector_type(8)));
typedef short v8s __attribute__((ext_vector_type(8)));
typedef short v4s __attribute__((ext_vector_type(4)));
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

__device__ __forceinline__ float h2f_dev(uint16_t h) { return (float)__builtin_bit_cast(_Float16, h); }

// w[32], a[32]: f16 bit patterns; isum[32][32]; out: s, nBs, cf (each [32][32], row-major [row][token])
__global__ void check_kernel(const uint16_t* w, const uint16_t* a, const int* isum, float* s_out, float* s8_out, float* n_out, float* cf_out) {
    const int l = threadIdx.x, tl = l & 31, hi = l >> 5;
    v8s as = {0, 0, 0, 0, 0, 0, 0, 0}, bs = as;
    if (!hi) { as[0] = (short)w[tl]; bs[0] = (short)a[tl]; }
    v16f z;
    for (int r = 0; r < 16; ++r) z[r] = 0.f;
    const v16f s = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(v8h, as), __builtin_bit_cast(v8h, bs), z, 0, 0, 0);
    v4s a4 = {0, 0, 0, 0}, b4 = a4;
    if (!hi) { a4[0] = (short)w[tl]; b4[0] = (short)a[tl]; }
    const v16f s8 = __builtin_amdgcn_mfma_f32_32x32x8f16(__builtin_bit_cast(v4h, a4), __builtin_bit_cast(v4h, b4), z, 0, 0, 0);
    // bf16 split
    const float wf = h2f_dev(w[tl]), af = h2f_dev(a[tl]);
    const float whi = __uint_as_float(__float_as_uint(wf) & 0xFFFF0000u), wlo = wf - whi;
    const float ahi = __uint_as_float(__float_as_uint(af) & 0xFFFF0000u), alo = af - ahi;
    auto bf = [](float x) { return (short)(__float_as_uint(x) >> 16); };
    v8s an, bn;
    // k = 8 * hi + e: lanes of half 0 carry the 2^23 terms, half 1 the 2^22 terms; elements 4..7 are zero
    const float sc = hi ? -4194304.f : -8388608.f;
    an[0] = bf(sc * whi); an[1] = bf(sc * whi); an[2] = bf(sc * wlo); an[3] = bf(sc * wlo);
    bn[0] = bf(ahi); bn[1] = bf(alo); bn[2] = bf(ahi); bn[3] = bf(alo);
    for (int e = 4; e < 8; ++e) { an[e] = 0; bn[e] = 0; }
    const v16f nb = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(v8b, an), __builtin_bit_cast(v8b, bn), z, 0, 0, 0);
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * hi;
        const float D = __int_as_float(0x4B400000 + isum[row * 32 + tl]);
        s_out[row * 32 + tl] = s[r];
        s8_out[row * 32 + tl] = s8[r];
        n_out[row * 32 + tl] = nb[r];
        cf_out[row * 32 + tl] = __builtin_fmaf(D, s[r], nb[r]);
    }
}

template <int WHICH>
__global__ void rate_kernel(int iters, float* sink, unsigned long long* cyc) {
    v16f c0, c1, c2, c3;
    for (int r = 0; r < 16; ++r) { c0[r] = 0.f; c1[r] = 1.f; c2[r] = 2.f; c3[r] = 3.f; }
    v8s a = {1, 2, 3, 4, 5, 6, 7, (short)threadIdx.x};
    v4i ai = {1, 2, 3, (int)threadIdx.x};
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
        if (WHICH == 0) {
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(v8h, a), __builtin_bit_cast(v8h, a), c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(v8h, a), __builtin_bit_cast(v8h, a), c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(v8h, a), __builtin_bit_cast(v8h, a), c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(v8h, a), __builtin_bit_cast(v8h, a), c3, 0, 0, 0);
        } else if (WHICH == 1) {
            v4s a4 = {a[0], a[1], a[2], a[7]};
            c0 = __builtin_amdgcn_mfma_f32_32x32x8f16(__builtin_bit_cast(v4h, a4), __builtin_bit_cast(v4h, a4), c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_32x32x8f16(__builtin_bit_cast(v4h, a4), __builtin_bit_cast(v4h, a4), c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_32x32x8f16(__builtin_bit_cast(v4h, a4), __builtin_bit_cast(v4h, a4), c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f32_32x32x8f16(__builtin_bit_cast(v4h, a4), __builtin_bit_cast(v4h, a4), c3, 0, 0, 0);
        } else if (WHICH == 2) {
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(v8b, a), __builtin_bit_cast(v8b, a), c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(v8b, a), __builtin_bit_cast(v8b, a), c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(v8b, a), __builtin_bit_cast(v8b, a), c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(v8b, a), __builtin_bit_cast(v8b, a), c3, 0, 0, 0);
        } else {
            v16i d0 = __builtin_bit_cast(v16i, c0), d1 = __builtin_bit_cast(v16i, c1), d2 = __builtin_bit_cast(v16i, c2), d3 = __builtin_bit_cast(v16i, c3);
            d0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(ai, ai, d0, 0, 0, 0);
            d1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(ai, ai, d1, 0, 0, 0);
            d2 = __builtin_amdgcn_mfma_i32_32x32x32_i8(ai, ai, d2, 0, 0, 0);
            d3 = __builtin_amdgcn_mfma_i32_32x32x32_i8(ai, ai, d3, 0, 0, 0);
            c0 = __builtin_bit_cast(v16f, d0); c1 = __builtin_bit_cast(v16f, d1); c2 = __builtin_bit_cast(v16f, d2); c3 = __builtin_bit_cast(v16f, d3);
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float acc = 0.f;
    for (int r = 0; r < 16; ++r) acc += c0[r] + c1[r] + c2[r] + c3[r];
    sink[blockIdx.x * blockDim.x + threadIdx.x] = acc;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

static float h2f_host(uint16_t h) {
    const int s = h >> 15, e = (h >> 10) & 31, m = h & 1023;
    float v = e == 0 ? ldexpf((float)m, -24) : ldexpf((float)(m | 1024), e - 25);
    return s ? -v : v;
}

int main() {
    uint16_t hw[32], ha[32];
    int hi_[1024];
    float hs[1024], hs8[1024], hn[1024], hcf[1024];
    uint16_t *w, *a; int* isum; float *s, *s8, *n, *cf;
    hipMalloc(&w, 64); hipMalloc(&a, 64); hipMalloc(&isum, 4096); hipMalloc(&s, 4096); hipMalloc(&s8, 4096); hipMalloc(&n, 4096); hipMalloc(&cf, 4096);
    srand(7);
    long bad_s = 0, bad_s8 = 0, bad_n = 0, bad_cf = 0, total = 0;
    for (int trial = 0; trial < 400; ++trial) {
        for (int i = 0; i < 32; ++i) {
            // positive f16 (scales are >= +0): all exponents incl. subnormals (e = 0), zero, the maximum 0x7BFF
            auto pick = [&]() -> uint16_t {
                const int m = rand() % 8;
                if (m == 0) return (uint16_t)(rand() % 1024);                 // subnormal or zero
                if (m == 1) return 0x7BFF;
                if (m == 2) return 0;
                return (uint16_t)(rand() % 0x7C00);                           // any finite positive
            };
            hw[i] = pick(); ha[i] = pick();
            if ((rand() & 3) == 0) hw[i] |= 0x8000;                             // a GGUF may carry a negative block scale
        }
        for (int i = 0; i < 1024; ++i) hi_[i] = (trial & 1) ? (rand() % (2 * 516128 + 1)) - 516128 : (rand() % 8001) - 4000;
        hipMemcpy(w, hw, 64, hipMemcpyHostToDevice); hipMemcpy(a, ha, 64, hipMemcpyHostToDevice); hipMemcpy(isum, hi_, 4096, hipMemcpyHostToDevice);
        check_kernel<<<1, 64>>>(w, a, isum, s, s8, n, cf);
        hipMemcpy(hs, s, 4096, hipMemcpyDeviceToHost); hipMemcpy(hs8, s8, 4096, hipMemcpyDeviceToHost);
        hipMemcpy(hn, n, 4096, hipMemcpyDeviceToHost); hipMemcpy(hcf, cf, 4096, hipMemcpyDeviceToHost);
        for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) {
            volatile float sp = h2f_host(hw[i]) * h2f_host(ha[j]);
            volatile float np = -12582912.f * sp;
            volatile float cp = (float)hi_[i * 32 + j] * sp;
            const float s_ref = sp, n_ref = np, c_ref = cp;
            bad_s += memcmp(&hs[i * 32 + j], &s_ref, 4) != 0;
            bad_s8 += memcmp(&hs8[i * 32 + j], &s_ref, 4) != 0;
            bad_n += memcmp(&hn[i * 32 + j], &n_ref, 4) != 0;
            // -0 vs +0: isum = 0 or s = 0 gives +-0 in the reference; acc + (+-0) is the same value unless acc is -0 (acc starts at +0)
            const float got = hcf[i * 32 + j];
            bad_cf += !(memcmp(&got, &c_ref, 4) == 0 || (got == 0.f && c_ref == 0.f));
            ++total;
        }
    }
    printf("exactness over %ld (row, token) pairs: s(32x32x16 f16) mismatches %ld, s(32x32x8 f16) %ld, -B s (bf16 8-term) %ld, fma(D, s, -B s) vs float(isum) * s %ld\n",
           total, bad_s, bad_s8, bad_n, bad_cf);
    float* sink; unsigned long long* cyc; unsigned long long hc;
    hipMalloc(&sink, 256 * 4 * 1024); hipMalloc(&cyc, 8);
    const int iters = 2000;
    const char* names[4] = {"v_mfma_f32_32x32x16_f16", "v_mfma_f32_32x32x8_f16", "v_mfma_f32_32x32x16_bf16", "v_mfma_i32_32x32x32_i8"};
    for (int which = 0; which < 4; ++which) {
        for (int rep = 0; rep < 2; ++rep) {
            if (which == 0) rate_kernel<0><<<256, 256>>>(iters, sink, cyc);
            if (which == 1) rate_kernel<1><<<256, 256>>>(iters, sink, cyc);
            if (which == 2) rate_kernel<2><<<256, 256>>>(iters, sink, cyc);
            if (which == 3) rate_kernel<3><<<256, 256>>>(iters, sink, cyc);
            hipDeviceSynchronize();
        }
        hipMemcpy(&hc, cyc, 8, hipMemcpyDeviceToHost);
        printf("%-28s %.1f cycles per instruction (one wavefront per SIMD, 4 independent accumulators)\n", names[which], (double)hc / (4.0 * iters));
    }
    return 0;
}
