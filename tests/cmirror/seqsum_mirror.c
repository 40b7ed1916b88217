/* CPU mirror of gpullama3.java_amd/csrc/gl3_seqsum.h (exact_sumsq_lds, 4-element segments): the same predictor, trial,
 * composition, replay and verification steps in plain C, so that the algorithm's exactness claim is tested without a GPU
 * (tests/test_seqsum_mirror.py).  Thread t of the kernel = loop index t here; wavefront scans = plain prefix sums (the
 * predictor is order-insensitive by design, the integer prefix is exact mod 2^32).
 *   gcc -O2 -ffp-contract=off -fno-fast-math seqsum_mirror.c -o seqsum_mirror -lm ; ./seqsum_mirror <trials> <seed>
 * exit status 0 = every trial bit-identical to the sequential chain. */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static inline uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static float naive_sumsq(const float* x, int n) { volatile float s = 0.f; for (int i = 0; i < n; ++i) { volatile float p = x[i] * x[i]; s = s + p; } return s; }

#define SS_T 256
static long g_fallback = 0, g_hard = 0, g_calls = 0;

static float exact_sumsq(const float* x, int n) {        /* n % 4 == 0, n <= 5120, x padded with zeros is not needed here */
    const int nseg = n >> 2, spt = (nseg + SS_T - 1) / SS_T;
    static uint32_t es[1280], pre[1280]; static int hlist[1280];
    static float qs[1280]; static uint32_t nd[1280];
    int fail_flag = 0, nhard = 0;
    volatile float a[4];
    ++g_calls;
    /* predictor: approximate prefix of the squares before each segment (any order) */
    for (int s = 0; s < nseg; ++s) {
        volatile float a0 = x[4 * s] * x[4 * s], a1 = x[4 * s + 1] * x[4 * s + 1], a2 = x[4 * s + 2] * x[4 * s + 2], a3 = x[4 * s + 3] * x[4 * s + 3];
        volatile float l = a0 + a1, r = a2 + a3;
        qs[s] = l + r;
    }
    /* kernel order: per-thread totals, wave scan, wave totals; here a prefix over threads, then within the thread */
    static float Pth[SS_T + 1];
    Pth[0] = 0.f;
    for (int t = 0; t < SS_T; ++t) {
        volatile float qt = 0.f;
        for (int i = 0; i < spt; ++i) { const int s = t * spt + i; if (s < nseg) qt = qt + qs[s]; }
        volatile float acc = Pth[t] + qt;
        Pth[t + 1] = acc;
    }
    for (int t = 0; t < SS_T; ++t) {
        volatile float P = Pth[t];
        for (int i = 0; i < spt; ++i) {
            const int s = t * spt + i;
            if (s >= nseg) break;
            for (int k = 0; k < 4; ++k) a[k] = x[4 * s + k] * x[4 * s + k];
            const uint32_t rb = f2u(P) & ~1u, e = rb >> 23;
            int hard = 0; uint32_t ndv = 0;
            if (s == 0 || e <= 40u || e >= 250u) hard = 1;
            else {
                const float R0 = u2f(rb), R1 = u2f(rb | 1u);
                volatile float E0 = R0, E1 = R1;
                for (int k = 0; k < 4; ++k) { E0 = E0 + a[k]; E1 = E1 + a[k]; }
                volatile float D0 = E0 - R0, D1 = E1 - R1;
                const float margin = u2f((e - 23u + 13u) << 23);
                volatile float lo = R0 - margin, hi = E0 + margin;
                if (!(D0 == D1) || (f2u(E0) >> 23) != e || (f2u(E1) >> 23) != e || (f2u(lo) >> 23) != e || (f2u(hi) >> 23) != e) hard = 1;
                else { volatile float q = D0 * u2f((277u - e) << 23); ndv = (uint32_t)q; }
            }
            es[s] = e | (hard ? 0x80000000u : 0u);
            nd[s] = ndv;
            P = P + qs[s];
        }
    }
    /* inclusive prefix of nd (mod 2^32), hard list in order, neighbour-exponent check of easy segments */
    uint32_t run = 0;
    for (int s = 0; s < nseg; ++s) {
        run += nd[s]; pre[s] = run;
        if (es[s] >> 31) hlist[nhard++] = s;
        else if (s > 0 && !(es[s - 1] >> 31) && (es[s - 1] & 0x7FFFFFFFu) != (es[s] & 0x7FFFFFFFu)) fail_flag = 1;
    }
    g_hard += nhard;
    /* replay: event j < nhard = easy run since the previous hard segment, then hard segment hlist[j]; event nhard = trailing run */
    volatile float base = 0.f;
    for (int j = 0; j <= nhard; ++j) {
        const int h = j < nhard ? hlist[j] : nseg, ph = j > 0 ? hlist[j - 1] : -1;
        if (h - 1 > ph) {
            const uint32_t R = pre[h - 1] - (ph >= 0 ? pre[ph] : 0u);
            const uint32_t er = es[ph + 1] & 0x7FFFFFFFu;
            const float runadd = (float)R * u2f((er - 23u) << 23);
            if ((f2u(base) >> 23) != er) fail_flag = 1;
            base = base + runadd;
            if ((f2u(base) >> 23) != er) fail_flag = 1;
        }
        if (h < nseg) for (int k = 0; k < 4; ++k) { volatile float p = x[4 * h + k] * x[4 * h + k]; base = base + p; }
    }
    if (fail_flag) { ++g_fallback; return naive_sumsq(x, n); }
    return base;
}

static float rnd01(void) { return (float)rand() / (float)RAND_MAX; }

int main(int argc, char** argv) {
    const int trials = argc > 1 ? atoi(argv[1]) : 12000;
    srand(argc > 2 ? atoi(argv[2]) : 12345);
    static float x[5120];
    long bad = 0;
    for (int trial = 0; trial < trials; ++trial) {
        const int kind = trial % 12;
        int n = (trial % 7 == 0) ? 2048 : (trial % 7 == 1) ? 2560 : (trial % 7 == 2) ? 5120 : (trial % 7 == 3) ? 4000 : 4096;
        if (kind == 11) n = 4 * (256 + rand() % 1024);
        for (int i = 0; i < n; ++i) {
            float v;
            switch (kind) {
            case 0: v = (rnd01() - 0.5f) * 2.f; break;
            case 1: { float u1 = rnd01() + 1e-9f, u2 = rnd01(); v = sqrtf(-2.f * logf(u1)) * cosf(6.2831853f * u2) * 0.02f; } break;
            case 2: v = (float)(rand() % 8); break;                                  /* exact ties everywhere */
            case 3: v = ldexpf(1.f, rand() % 12 - 6); break;
            case 4: v = (rand() % 50 == 0) ? 100.f * rnd01() : 1e-3f * rnd01(); break;
            case 5: v = 1.0f; break;
            case 6: v = (i < 5) ? 1e-12f : rnd01(); break;
            case 7: v = (i % 97 == 0) ? 0.f : (float)(rand() % 3) * 0.5f; break;
            case 8: v = ldexpf(rnd01(), rand() % 40 - 20); break;                    /* wide dynamic range */
            case 9: v = (i == n / 2) ? 3000.f : rnd01() * 0.01f; break;              /* one giant element */
            case 10: v = (float)(1 + rand() % 4) * 0.25f; break;
            default: v = rnd01() * 3.f; break;
            }
            x[i] = v;
        }
        if (kind == 3 && trial % 24 == 3) for (int i = 0; i < n; ++i) x[i] = 0.f;
        const float ref = naive_sumsq(x, n), got = exact_sumsq(x, n);
        if (f2u(ref) != f2u(got)) { if (bad < 5) printf("MISMATCH trial %d kind %d n %d: ref %.9g got %.9g\n", trial, kind, n, ref, got); ++bad; }
    }
    printf("trials %d mismatches %ld fallbacks %ld hard/call %.1f\n", trials, bad, g_fallback, (double)g_hard / (double)g_calls);
    return bad != 0;
}
