// gl3_seqsum.h — exact parallel evaluation of the strictly sequential binary32 sum
//     s = (((0 + a_0) + a_1) + ... + a_{n-1}),   a_k = x_k * x_k >= 0,
// i.e. InferenceCore.rmsnorm's `x.reduce(0f, (acc, xi) -> acc + xi * xi)` (J/inference/InferenceCore.java:41),
// bit for bit, without a 4096-step dependent chain (13 cycles / element on gfx950 = 22 us for dim 4096).
//
// Idea (CPU mirror with 120k adversarial trials: scripts/probes/seqsum_proto2.c): while the running sum stays
// inside one binade it is N*u with an integer N, and adding a_k adds an integer that depends only on a_k
// (and, on an exact rounding tie, on the parity of N).  So a segment of m elements that (i) starts in the
// binade predicted by an approximate prefix sum, (ii) does not leave it and (iii) gives the same increment D
// from an even and from an odd start is a pure translation s -> s + D, and translations compose exactly
// (integer sums of D/u).  Each of 256 threads finds its segment's D by running the real f32 chain from two
// representative starts; the few "hard" segments (a binade crossing, a parity-dependent tie, the first one;
// ~13 of 256 for Gaussian data) are replayed with real f32 adds from their true start = previous hard
// segment's end + (integer run sum) * u.  The premises are re-checked on the true values at every run
// boundary (values are monotone, so boundary checks cover the interior); on any violation the naive chain
// runs instead, so the result is always exactly the sequential sum.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace gl3 {

#ifdef GL3_SS_TIMING
__device__ long long gl3_ss_stamp[16];
#define SS_STAMP(i) do { if ((threadIdx.x & 255) == 0) gl3_ss_stamp[i] = clock64(); } while (0)
#else
#define SS_STAMP(i)
#endif

constexpr int SS_T = 256;                          // threads that own a segment
constexpr int SS_SCRATCH_BYTES = 4 * 1024;         // LDS scratch the caller must provide (16-byte aligned)

__device__ __forceinline__ uint32_t f2u(float f) { return __builtin_bit_cast(uint32_t, f); }
__device__ __forceinline__ float u2f(uint32_t u) { return __builtin_bit_cast(float, u); }

// wave64 inclusive scan with DPP row shifts + row broadcasts (the sequence LLVM's atomic optimizer emits on gfx9)
__device__ __forceinline__ uint32_t wave_incl_scan_u32(uint32_t v) {
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);   // row_shr:1
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);   // row_shr:2
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false);   // row_shr:4
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false);   // row_shr:8
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);   // row_bcast15 -> rows 1,3
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);   // row_bcast31 -> rows 2,3
    return v;
}
__device__ __forceinline__ float wave_incl_scan_f32(float v) {                        // predictor only (any order)
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x111, 0xf, 0xf, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x112, 0xf, 0xf, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x114, 0xf, 0xf, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x118, 0xf, 0xf, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x142, 0xa, 0xf, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x143, 0xc, 0xf, false));
    return v;
}

// naive chain over x[k0..k1), executed redundantly by all lanes of one wavefront (LDS broadcast reads)
__device__ __forceinline__ float naive_sumsq_lds(const float* x, int k0, int k1, float s) {
    int k = k0;
    for (; k + 4 <= k1; k += 4) {
        const float4 v = *reinterpret_cast<const float4*>(x + k);
        s = s + v.x * v.x; s = s + v.y * v.y; s = s + v.z * v.z; s = s + v.w * v.w;
    }
    for (; k < k1; ++k) { const float v = x[k]; s = s + v * v; }
    return s;
}

// Sequential part of exact_sumsq_lds, run by one wavefront.  Event j < nhard = hard segment hlist[j] preceded by
// the easy run since the previous hard segment; event nhard = the trailing easy run.  Lane j gathers event j's
// run translation (exact: integer run sum * ulp); the M4*4 squares of a hard segment are fetched with uniform
// (broadcast) LDS reads one event ahead (double buffer A/B), so the dependent f32 add chain never waits on LDS.
template <int M4>
__device__ __forceinline__ void ss_load(float4 (&buf)[M4], const float* x, int h, int m, int nseg) {
    if (h < nseg) {
#pragma unroll
        for (int g = 0; g < M4; ++g) buf[g] = *reinterpret_cast<const float4*>(x + h * m + 4 * g);
    }
}
template <int M4>
__device__ __forceinline__ float ss_chain(const float4 (&buf)[M4], float base) {
#pragma unroll
    for (int g = 0; g < M4; ++g) {
        const float a0 = buf[g].x * buf[g].x, a1 = buf[g].y * buf[g].y, a2 = buf[g].z * buf[g].z, a3 = buf[g].w * buf[g].w;
        base = base + a0; base = base + a1; base = base + a2; base = base + a3;
    }
    return base;
}
template <int M4>
__device__ __forceinline__ void replay_events(const float* x, int m, int nseg, int nhard, const uint32_t* es, const uint32_t* pre,
                                              const int* hlist, float& base, int& fail) {
    const int lane = threadIdx.x & 63;
    for (int c0 = 0; c0 <= nhard; c0 += 64) {
        const int j = c0 + lane;
        const bool valid = j <= nhard;
        const int h = (valid && j < nhard) ? hlist[j] : nseg;
        const int ph = (valid && j > 0) ? hlist[j - 1] : -1;
        const int hasrun = (valid && (h - 1 > ph)) ? 1 : 0;
        uint32_t er = 0;
        float runadd = 0.f;
        if (hasrun) {
            const uint32_t R = pre[h - 1] - (ph >= 0 ? pre[ph] : 0u);
            er = es[ph + 1] & 0x7FFFFFFFu;
            runadd = (float)R * u2f((er - 23u) << 23);
        }
        SS_STAMP(5);
        const int nev = min(64, nhard + 1 - c0);
        float4 A[M4], B[M4];
        ss_load<M4>(A, x, __builtin_amdgcn_readlane(h, 0), m, nseg);
        for (int jj = 0; jj < nev; jj += 2) {
            ss_load<M4>(B, x, jj + 1 < nev ? __builtin_amdgcn_readlane(h, (jj + 1) & 63) : nseg, m, nseg);
            {
                const uint32_t e_r = (uint32_t)__builtin_amdgcn_readlane((int)er, jj & 63);
                if (__builtin_amdgcn_readlane(hasrun, jj & 63)) {
                    fail |= (f2u(base) >> 23) != e_r;
                    base = base + u2f((uint32_t)__builtin_amdgcn_readlane((int)f2u(runadd), jj & 63));
                    fail |= (f2u(base) >> 23) != e_r;
                }
                if (c0 + jj < nhard) base = ss_chain<M4>(A, base);
            }
            if (jj + 1 >= nev) break;
            ss_load<M4>(A, x, jj + 2 < nev ? __builtin_amdgcn_readlane(h, (jj + 2) & 63) : nseg, m, nseg);
            {
                const uint32_t e_r = (uint32_t)__builtin_amdgcn_readlane((int)er, (jj + 1) & 63);
                if (__builtin_amdgcn_readlane(hasrun, (jj + 1) & 63)) {
                    fail |= (f2u(base) >> 23) != e_r;
                    base = base + u2f((uint32_t)__builtin_amdgcn_readlane((int)f2u(runadd), (jj + 1) & 63));
                    fail |= (f2u(base) >> 23) != e_r;
                }
                if (c0 + jj + 1 < nhard) base = ss_chain<M4>(B, base);
            }
        }
    }
}

// Called by exactly 256 threads (4 wavefronts) of the workgroup.  x: n floats in LDS, 16-byte
// aligned, n a multiple of 4, 1024 <= n <= 5120 (callers use the plain chain outside that range), followed by at least 32 readable ZERO floats (segment padding).  Returns the sequential sum of squares in every thread.
// `t` = index of the calling thread among the 256 participating threads; `sync()` is a barrier over exactly
// those 4 wavefronts (the whole workgroup's __syncthreads, or a sub-group barrier — see SubBarrier).
struct BlockBarrier { __device__ __forceinline__ void operator()() const { __syncthreads(); } };

// Barrier over a subset of the workgroup's wavefronts through an LDS counter (gfx950 has no named barriers).
// ctr must be zero before first use; every participating wavefront calls operator() the same number of times.
struct SubBarrier {
    int* ctr; int nwaves; int target;
    __device__ __forceinline__ void operator()() {
        target += nwaves;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        if ((threadIdx.x & 63) == 0) __hip_atomic_fetch_add(ctr, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < target) __builtin_amdgcn_s_sleep(1);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    }
};

template <typename Sync>
__device__ float exact_sumsq_lds(const float* x, int n, uint8_t* scratch, const int t, Sync& sync) {
    const int lane = t & 63, wave = t >> 6;
    float* w_tot = reinterpret_cast<float*>(scratch);                    // [8]  predictor wave totals
    uint32_t* w_nd = reinterpret_cast<uint32_t*>(scratch + 32);          // [8]  run-sum wave totals
    int* w_cnt = reinterpret_cast<int*>(scratch + 64);                   // [8]  hard segments per wave
    int* misc = reinterpret_cast<int*>(scratch + 96);                    // [0] fail flag
    float* result = reinterpret_cast<float*>(scratch + 112);             // [0]
    uint32_t* es = reinterpret_cast<uint32_t*>(scratch + 128);           // [256] predicted exponent | hard<<31
    uint32_t* pre = es + SS_T;                                           // [256] inclusive prefix of D/u (mod 2^32)
    int* hlist = reinterpret_cast<int*>(pre + SS_T);                     // [256] hard segment ids in order

    const int m = ((n + SS_T - 1) / SS_T + 3) & ~3;
    const int nseg = (n + m - 1) / m;
    const bool own = t < nseg;
    const int k0 = own ? t * m : n, k1 = own ? min(n, k0 + m) : n;
    if (t == 0) misc[0] = 0;
    SS_STAMP(0);

    // ---- predictor: approximate (any-order) f32 prefix of the squares before segment t
    float q = 0.f;
    for (int k = k0; k < k1; k += 4) {
        const float4 v = *reinterpret_cast<const float4*>(x + k);
        q += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
    }
    const float incl = wave_incl_scan_f32(q);
    if (lane == 63 && wave < 8) w_tot[wave] = incl;
    sync();
    SS_STAMP(1);
    float P = incl - q;
    for (int w = 0; w < wave && w < 8; ++w) P += w_tot[w];

    // ---- translation D of the segment from two representative starts (even / odd mantissa)
    bool hard = false;
    uint32_t nd = 0, e = 0;
    if (own) {
        const uint32_t rb = f2u(P) & ~1u;
        e = rb >> 23;
        if (t == 0 || e <= 40u || e >= 250u) hard = true;
        else {
            const float R0 = u2f(rb), R1 = u2f(rb | 1u);
            float E0 = R0, E1 = R1;
            for (int k = k0; k < k1; k += 4) {
                const float4 v = *reinterpret_cast<const float4*>(x + k);
                const float a0 = v.x * v.x, a1 = v.y * v.y, a2 = v.z * v.z, a3 = v.w * v.w;
                E0 = E0 + a0; E1 = E1 + a0; E0 = E0 + a1; E1 = E1 + a1;
                E0 = E0 + a2; E1 = E1 + a2; E0 = E0 + a3; E1 = E1 + a3;
            }
            const float D0 = E0 - R0, D1 = E1 - R1;
            const float margin = u2f((e - 23u + 13u) << 23);             // 8192 ulp: keeps the fallback rare
            if (!(D0 == D1) || (f2u(E0) >> 23) != e || (f2u(E1) >> 23) != e || (f2u(R0 - margin) >> 23) != e ||
                (f2u(E0 + margin) >> 23) != e)
                hard = true;
            else nd = (uint32_t)(D0 * u2f((277u - e) << 23));            // D / ulp, exact integer < 2^24
        }
        es[t] = e | (hard ? 0x80000000u : 0u);
    }
    SS_STAMP(2);
    // ---- inclusive prefix of nd (mod 2^32: only differences inside one run are used) + hard list
    const uint32_t pin = wave_incl_scan_u32(nd);
    const unsigned long long hb = __ballot(hard);
    if (lane == 63 && wave < 8) { w_nd[wave] = pin; w_cnt[wave] = __popcll(hb); }
    sync();
    SS_STAMP(3);
    uint32_t nbase = 0; int hbase = 0;
    for (int w = 0; w < wave && w < 8; ++w) { nbase += w_nd[w]; hbase += w_cnt[w]; }
    if (own) {
        pre[t] = pin + nbase;
        if (hard) hlist[hbase + __popcll(hb & ((1ull << lane) - 1ull))] = t;
        else if (t > 0) { const uint32_t ep = es[t - 1]; if (!(ep >> 31) && ep != e) misc[0] = 1; }   // one binade per run
    }
    sync();
    SS_STAMP(4);
    // ---- replay the hard segments in order (one wavefront), see replay_events
    if (wave == 0) {
        int nhard = 0;
        for (int w = 0; w < 8 && w * 64 < SS_T; ++w) nhard += w_cnt[w];
        float base = 0.f;
        int fail = 0;
        switch (m >> 2) {
        case 2: replay_events<2>(x, m, nseg, nhard, es, pre, hlist, base, fail); break;
        case 3: replay_events<3>(x, m, nseg, nhard, es, pre, hlist, base, fail); break;
        case 4: replay_events<4>(x, m, nseg, nhard, es, pre, hlist, base, fail); break;
        case 5: replay_events<5>(x, m, nseg, nhard, es, pre, hlist, base, fail); break;
        default: fail = 1; break;
        }
        if (fail || misc[0]) base = naive_sumsq_lds(x, 0, n, 0.f);       // never expected: plain chain
        if (lane == 0) result[0] = base;
        SS_STAMP(6);
    }
    sync();
    SS_STAMP(7);
    return result[0];
}

}  // namespace gl3
