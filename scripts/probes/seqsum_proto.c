// CPU prototype of the exact parallel evaluation of s = (((0 + a0) + a1) + ...) in binary32, a_k >= 0.
// Emulates the thread-parallel structure of the HIP device function (T threads x m elements) sequentially and
// checks it bit for bit against the naive chain on random and adversarial inputs.
// gcc -O2 -ffp-contract=off seqsum_proto.c -o seqsum_proto -lm
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static inline uint32_t fbits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float bfloat(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

static float naive(const float* a, int n) { volatile float s = 0.f; for (int i = 0; i < n; ++i) s = s + a[i]; return s; }

static long g_fallbacks = 0, g_hard = 0, g_calls = 0;

#define T 256
static float fast_seqsum(const float* a, int n) {
    const int m = (n + T - 1) / T;
    double Q[T], P[T + 1];
    float D[T];
    int hard[T];
    uint32_t ebits[T];
    ++g_calls;
    for (int t = 0; t < T; ++t) {
        double q = 0;
        for (int k = t * m; k < n && k < (t + 1) * m; ++k) q += (double)a[k];
        Q[t] = q;
    }
    P[0] = 0;
    for (int t = 0; t < T; ++t) P[t + 1] = P[t] + Q[t];          // (a parallel scan on the GPU)
    for (int t = 0; t < T; ++t) {
        const int k0 = t * m, k1 = (t + 1) * m < n ? (t + 1) * m : n;
        hard[t] = 0; D[t] = 0.f; ebits[t] = 0;
        if (k0 >= n) continue;
        const float r = (float)P[t];
        const uint32_t rb = fbits(r) & ~1u;
        const uint32_t e = rb >> 23;
        ebits[t] = e;
        if (t == 0 || e <= 40 || e >= 250) { hard[t] = 1; continue; }
        const float R0 = bfloat(rb), R1 = bfloat(rb | 1u);
        float E0 = R0, E1 = R1;
        for (int k = k0; k < k1; ++k) { E0 = E0 + a[k]; E1 = E1 + a[k]; }
        const float D0 = E0 - R0, D1 = E1 - R1;
        const float margin = bfloat((e - 23 + 13) << 23);          // 8192 ulp
        if (!(D0 == D1) || (fbits(E0) >> 23) != e || (fbits(E1) >> 23) != e || (fbits(R0 - margin) >> 23) != e ||
            (fbits(E0 + margin) >> 23) != e) { hard[t] = 1; continue; }
        D[t] = D0;
    }
    // resolution: sequential over hard segments, easy runs summed exactly in double
    float s = 0.f;              // value at the start of segment t
    double run = 0;             // sum of easy D since the last hard segment
    float base = 0.f;           // chain value right after the last hard segment
    int ok = 1;
    for (int t = 0; t < T; ++t) {
        const int k0 = t * m, k1 = (t + 1) * m < n ? (t + 1) * m : n;
        if (k0 >= n) break;
        const double sv = (double)base + run;
        s = (float)sv;
        if ((double)s != sv) ok = 0;
        if (hard[t]) {
            ++g_hard;
            for (int k = k0; k < k1; ++k) s = s + a[k];
            base = s; run = 0;
        } else {
            // verification of the translation property's premises for the TRUE start
            if ((fbits(s) >> 23) != ebits[t] || (fbits(s + D[t]) >> 23) != ebits[t]) ok = 0;
            run += (double)D[t];
        }
    }
    const double fv = (double)base + run;
    float res = (float)fv;
    if ((double)res != fv) ok = 0;
    if (!ok) { ++g_fallbacks; return naive(a, n); }
    return res;
}

static float rnd01(void) { return (float)rand() / (float)RAND_MAX; }

int main(void) {
    srand(12345);
    static float a[16384];
    long bad = 0, total = 0;
    for (int trial = 0; trial < 60000; ++trial) {
        const int kind = trial % 12;
        int n = (trial % 7 == 0) ? 2048 : (trial % 7 == 1) ? 2560 : (trial % 7 == 2) ? 5120 : (trial % 7 == 3) ? 4000 : 4096;
        if (kind == 11) n = 1 + rand() % 9000;
        for (int i = 0; i < n; ++i) {
            float x;
            switch (kind) {
            case 0: x = (rnd01() - 0.5f) * 2.f; break;                                  // uniform
            case 1: { float u1 = rnd01() + 1e-9f, u2 = rnd01(); x = sqrtf(-2.f * logf(u1)) * cosf(6.2831853f * u2) * 0.02f; } break;
            case 2: x = (float)(rand() % 8);  break;                                    // small ints: ties + exact adds
            case 3: x = ldexpf(1.f, rand() % 12 - 6); break;                            // powers of two
            case 4: x = (rand() % 50 == 0) ? 100.f * rnd01() : 1e-3f * rnd01(); break;  // outliers
            case 5: x = 1.0f; break;                                                    // constant
            case 6: x = (i < 5) ? 1e-12f : rnd01(); break;                              // tiny head
            case 7: x = (i % 97 == 0) ? 0.f : (float)(rand() % 3) * 0.5f; break;        // zeros and halves
            case 8: x = ldexpf(rnd01(), rand() % 40 - 20); break;                       // wide dynamic range
            case 9: x = (i == n / 2) ? 3000.f : rnd01() * 0.01f; break;                 // one huge element
            case 10: x = (float)(1 + rand() % 4) * 0.25f; break;                        // quarter steps: many ties
            default: x = rnd01() * 3.f; break;
            }
            a[i] = x * x;       // squares, as in rmsnorm
        }
        if (kind == 3 && trial % 24 == 3) for (int i = 0; i < n; ++i) a[i] = 0.f;       // all zero
        const float ref = naive(a, n), got = fast_seqsum(a, n);
        ++total;
        if (fbits(ref) != fbits(got)) { if (++bad < 10) printf("MISMATCH trial %d kind %d n %d: %.9g vs %.9g\n", trial, kind, n, got, ref); }
    }
    printf("trials %ld mismatches %ld fallbacks %ld avg hard segments %.2f\n", total, bad, g_fallbacks, (double)g_hard / g_calls);
    return bad != 0;
}
