// CPU mirror of csrc/gl3_seqsum.h (v2: float predictor, uint32 run sums, run-boundary verification).
// gcc -O2 -ffp-contract=off seqsum_proto2.c -o seqsum_proto2 -lm
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
static inline uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static float naive(const float* a, int n) { volatile float s = 0.f; for (int i = 0; i < n; ++i) s = s + a[i]; return s; }
static long g_fallbacks = 0, g_hard = 0, g_calls = 0;
#define T 256
static float fast(const float* a, int n) {
    const int m = ((n + T - 1) / T + 3) & ~3;         // multiple of 4 (aligned LDS reads)
    const int nseg = (n + m - 1) / m;
    float q[T], P[T + 1]; uint32_t es[T], nd[T], pre[T]; int hard[T];
    ++g_calls;
    for (int t = 0; t < nseg; ++t) { float s = 0; for (int k = t * m; k < n && k < (t + 1) * m; ++k) s += a[k]; q[t] = s; }
    // tree-ish float prefix (order irrelevant: it is only a predictor)
    P[0] = 0; for (int t = 0; t < nseg; ++t) P[t + 1] = P[t] + q[t];
    for (int t = 0; t < nseg; ++t) {
        const int k0 = t * m, k1 = (t + 1) * m < n ? (t + 1) * m : n;
        const uint32_t rb = f2u(P[t]) & ~1u, e = rb >> 23;
        hard[t] = 0; nd[t] = 0; es[t] = e;
        if (t == 0 || e <= 40 || e >= 250) { hard[t] = 1; continue; }
        const float R0 = u2f(rb), R1 = u2f(rb | 1u);
        float E0 = R0, E1 = R1;
        for (int k = k0; k < k1; ++k) { E0 = E0 + a[k]; E1 = E1 + a[k]; }
        const float D0 = E0 - R0, D1 = E1 - R1;
        const float margin = u2f((e - 23 + 13) << 23);
        if (!(D0 == D1) || (f2u(E0) >> 23) != e || (f2u(E1) >> 23) != e || (f2u(R0 - margin) >> 23) != e || (f2u(E0 + margin) >> 23) != e) { hard[t] = 1; continue; }
        nd[t] = (uint32_t)(D0 * u2f((277 - e) << 23));
    }
    int ok = 1;
    uint32_t acc = 0;
    for (int t = 0; t < nseg; ++t) { acc += nd[t]; pre[t] = acc; if (!hard[t] && t > 0 && !hard[t - 1] && es[t - 1] != es[t]) ok = 0; }
    float base = 0.f; int ph = -1;
    for (int t = 0; t <= nseg; ++t) {
        if (t < nseg && !hard[t]) continue;
        float s = base;
        if (t - 1 > ph) {                                 // easy run ph+1 .. t-1
            const uint32_t R = pre[t - 1] - (ph >= 0 ? pre[ph] : 0u);
            const uint32_t er = es[ph + 1];
            if ((f2u(base) >> 23) != er) ok = 0;
            s = base + (float)R * u2f((er - 23) << 23);
            if ((f2u(s) >> 23) != er) ok = 0;
        }
        if (t == nseg) { base = s; break; }
        ++g_hard;
        for (int k = t * m; k < n && k < (t + 1) * m; ++k) s = s + a[k];
        base = s; ph = t;
    }
    if (!ok) { ++g_fallbacks; return naive(a, n); }
    return base;
}
static float rnd01(void) { return (float)rand() / (float)RAND_MAX; }
int main(void) {
    srand(12345);
    static float a[16384];
    long bad = 0, total = 0; long hk[12] = {0}, ck[12] = {0}, fk[12] = {0};
    for (int trial = 0; trial < 120000; ++trial) {
        const int kind = trial % 12;
        int n = (trial % 7 == 0) ? 2048 : (trial % 7 == 1) ? 2560 : (trial % 7 == 2) ? 5120 : (trial % 7 == 3) ? 4000 : 4096;
        if (kind == 11) n = 4 * (1 + rand() % 2040);
        for (int i = 0; i < n; ++i) {
            float x;
            switch (kind) {
            case 0: x = (rnd01() - 0.5f) * 2.f; break;
            case 1: { float u1 = rnd01() + 1e-9f, u2 = rnd01(); x = sqrtf(-2.f * logf(u1)) * cosf(6.2831853f * u2) * 0.02f; } break;
            case 2: x = (float)(rand() % 8); break;
            case 3: x = ldexpf(1.f, rand() % 12 - 6); break;
            case 4: x = (rand() % 50 == 0) ? 100.f * rnd01() : 1e-3f * rnd01(); break;
            case 5: x = 1.0f; break;
            case 6: x = (i < 5) ? 1e-12f : rnd01(); break;
            case 7: x = (i % 97 == 0) ? 0.f : (float)(rand() % 3) * 0.5f; break;
            case 8: x = ldexpf(rnd01(), rand() % 40 - 20); break;
            case 9: x = (i == n / 2) ? 3000.f : rnd01() * 0.01f; break;
            case 10: x = (float)(1 + rand() % 4) * 0.25f; break;
            default: x = rnd01() * 3.f; break;
            }
            a[i] = x * x;
        }
        if (kind == 3 && trial % 24 == 3) for (int i = 0; i < n; ++i) a[i] = 0.f;
        long h0 = g_hard, f0 = g_fallbacks;
        const float ref = naive(a, n), got = fast(a, n);
        hk[kind] += g_hard - h0; ck[kind]++; fk[kind] += g_fallbacks - f0;
        ++total;
        if (f2u(ref) != f2u(got)) { if (++bad < 10) printf("MISMATCH trial %d kind %d n %d: %.9g vs %.9g\n", trial, kind, n, got, ref); }
    }
    printf("trials %ld mismatches %ld fallbacks %ld\n", total, bad, g_fallbacks);
    for (int k = 0; k < 12; ++k) printf("kind %2d: avg hard %.1f fallbacks %ld/%ld\n", k, (double)hk[k] / ck[k], fk[k], ck[k]);
    return bad != 0;
}
