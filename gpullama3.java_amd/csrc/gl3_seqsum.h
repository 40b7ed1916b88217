// gl3_seqsum.h — exact parallel evaluation of the strictly sequential binary32 sum
//     s = (((0 + a_0) + a_1) + ... + a_{n-1}),   a_k = x_k * x_k >= 0,
// i.e. InferenceCore.rmsnorm's `x.reduce(0f, (acc, xi) -> acc + xi * xi)` (J/inference/InferenceCore.java:41),
// bit for bit, without a 4096-step dependent chain (13 cycles / element on gfx950 = 22 us for dim 4096).
//
// Idea (CPU mirror with adversarial trials: tests/test_seqsum_mirror.py): while the running sum stays
// inside one binade it is N*u with an integer N, and adding a_k adds an integer that depends only on a_k
// (and, on an exact rounding tie, on the parity of N).  So a segment of m elements that (i) starts in the
// binade predicted by an approximate prefix sum, (ii) does not leave it and (iii) gives the same increment D
// from an even and from an odd start is a pure translation s -> s + D, and translations compose exactly
// (integer sums of D/u).  Each of 256 threads finds the D of its 4-element segments by running the real f32 chain from two
// representative starts; the few "hard" segments (a binade crossing, a parity-dependent tie, the first one;
// ~18 of 1024 four-element segments for Gaussian data) are replayed with real f32 adds from their true start = previous hard
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
template <bool SQ = true>
__device__ __forceinline__ float naive_sumsq_lds(const float* x, int k0, int k1, float s) {
    int k = k0;
    for (; k + 4 <= k1; k += 4) {
        const float4 v = *reinterpret_cast<const float4*>(x + k);
        if (SQ) { s = s + v.x * v.x; s = s + v.y * v.y; s = s + v.z * v.z; s = s + v.w * v.w; }
        else { s = s + v.x; s = s + v.y; s = s + v.z; s = s + v.w; }
    }
    for (; k < k1; ++k) { const float v = x[k]; s = SQ ? s + v * v : s + v; }
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
template <int M4, bool SQ = true>
__device__ __forceinline__ void replay_events(const float* x, int m, int nseg, int nhard, const uint32_t* es, const uint32_t* pre,
                                              const int* hlist, float& base, int& fail) {
    static_assert(M4 == 1, "segments are one float4");
    const int lane = threadIdx.x & 63;
    for (int c0 = 0; c0 <= nhard; c0 += 64) {
        // lane j gathers everything event c0 + j needs in ONE parallel LDS round trip
        const int j = c0 + lane;
        const bool valid = j <= nhard;
        const int h = (valid && j < nhard) ? hlist[j] : nseg;
        const int ph = (valid && j > 0) ? hlist[j - 1] : -1;
        uint32_t er = 0;                                   // 0 = no easy run before this event
        float runadd = 0.f;
        if (valid && (h - 1 > ph)) {
            const uint32_t R = pre[h - 1] - (ph >= 0 ? pre[ph] : 0u);
            er = es[ph + 1] & 0x7FFFFFFFu;
            runadd = (float)R * u2f((er - 23u) << 23);     // exact: integer run sum * ulp
        }
        float4 sq = {0.f, 0.f, 0.f, 0.f};                  // +0 adds nothing for the trailing event
        if (h < nseg) {
            const float4 v = *reinterpret_cast<const float4*>(x + 4 * h);
            if (SQ) { sq.x = v.x * v.x; sq.y = v.y * v.y; sq.z = v.z * v.z; sq.w = v.w * v.w; }
            else sq = v;
        }
        SS_STAMP(5);
        const int nev = __builtin_amdgcn_readfirstlane(min(64, nhard + 1 - c0));   // scalar loop control (s_cmp, not a VALU compare + vcc branch)
        // The sequential part.  Lane j already holds event j's operands, so instead of feeding one chain through v_readlane
        // (5 per event: ~150 cycles per event for a lone wavefront) EVERY lane runs the event's five adds on its own operands and
        // the result moves one lane up (v_mov_dpp wave_shr:1): at step k lane k holds the true incoming value — lane 0 from the
        // start, lane k from lane k - 1's step k - 1 — and keeps it from then on (lanes <= k ignore the shifted value).  After nev
        // steps lane j < nev still has the value before its event for the deferred binade checks, and lane nev - 1's last result
        // is the chain value after the chunk.  Per step: five adds, one DPP move, one select.
        float cur = base, t2 = base;
        for (int k = 0; k < nev; ++k) {
            const float t1 = cur + runadd;                  // +0 when there is no easy run before the event
            t2 = t1 + sq.x; t2 = t2 + sq.y; t2 = t2 + sq.z; t2 = t2 + sq.w;
            const float up = u2f((uint32_t)__builtin_amdgcn_update_dpp(0, (int)f2u(t2), 0x138, 0xf, 0xf, false));   // wave_shr:1
            cur = lane > k ? up : cur;
        }
        if (lane < nev && er != 0u) {
            const float after_run = cur + runadd;
            fail |= (int)((f2u(cur) >> 23) != er) | (int)((f2u(after_run) >> 23) != er);
        }
        base = u2f((uint32_t)__builtin_amdgcn_readlane((int)f2u(t2), nev - 1));
    }
    fail = __builtin_amdgcn_ballot_w64(fail != 0) != 0;          // any lane's violation fails the wavefront
}

// Barrier functors for exact_sumsq_lds: `sync()` must be a barrier over exactly the 4 participating wavefronts.
struct BlockBarrier { __device__ __forceinline__ void operator()() const { __syncthreads(); } };

// Barrier over a subset of the workgroup's wavefronts through an LDS counter (gfx950 has no named barriers).
// ctr must be zero before first use; every participating wavefront calls operator() the same number of times.
struct SubBarrier {
    int* ctr; int nwaves; int target;
    __device__ __forceinline__ void operator()() {
        target += nwaves;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        if ((threadIdx.x & 63) == 0) __hip_atomic_fetch_add(ctr, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < target) {}     // busy poll: an s_sleep(1) quantum is 64 cycles
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    }
};

// scratch bytes needed for n elements (16-byte aligned): header + es/pre/hlist per 4-element segment
__host__ __device__ constexpr size_t ss_scratch_bytes(int n) { return 128 + (size_t)3 * n; }
constexpr int SS_MAX_SPT = 5;                      // segments per thread: n <= 4 * 256 * 5 = 5120

// SQ = true: a_k = x_k * x_k (RMSNorm); SQ = false: a_k = x_k >= 0 (FloatTensor.sum of softmax numerators / probabilities,
// J/tensor/standard/FloatTensor.java:211-219).  start = value of the running sum before x[0] (chunked sums of long vectors).
template <bool SQ, typename Sync>
__device__ float exact_seqsum_lds(const float* x, int n, uint8_t* scratch, const int t, Sync& sync, const float start) {
    const int lane = t & 63, wave = t >> 6;
    float* w_tot = reinterpret_cast<float*>(scratch);                    // [4]  predictor wave totals
    uint32_t* w_nd = reinterpret_cast<uint32_t*>(scratch + 16);          // [4]  run-sum wave totals
    int* w_cnt = reinterpret_cast<int*>(scratch + 32);                   // [4]  hard segments per wave
    int* misc = reinterpret_cast<int*>(scratch + 48);                    // [0] fail flag
    float* result = reinterpret_cast<float*>(scratch + 64);              // [0]
    const int nseg = n >> 2;                                             // 4-element segments (one float4 each)
    uint32_t* es = reinterpret_cast<uint32_t*>(scratch + 128);           // [nseg] predicted exponent | hard<<31
    uint32_t* pre = es + nseg;                                           // [nseg] inclusive prefix of D/ulp (mod 2^32)
    int* hlist = reinterpret_cast<int*>(pre + nseg);                     // [nseg] hard segment ids in order
    const int spt = (nseg + SS_T - 1) / SS_T;
    const int seg0 = t * spt;
    if (t == 0) misc[0] = 0;
    SS_STAMP(0);

    // ---- predictor: approximate (any-order) f32 prefix of the squares before each of my segments
    float4 a[SS_MAX_SPT];
    float qs[SS_MAX_SPT];
    float qt = 0.f;
    // all LDS reads are issued unconditionally (segments past the end read the zero padding behind x[n]): reads under a
    // lane-dependent condition are waited for one by one
#pragma unroll
    for (int i = 0; i < SS_MAX_SPT; ++i) {
        const float4 v = *reinterpret_cast<const float4*>(x + 4 * min(seg0 + i, nseg));
        if (SQ) { a[i].x = v.x * v.x; a[i].y = v.y * v.y; a[i].z = v.z * v.z; a[i].w = v.w * v.w; }
        else a[i] = v;
    }
#pragma unroll
    for (int i = 0; i < SS_MAX_SPT; ++i) {
        qs[i] = 0.f;
        if (i < spt) {                                   // wave-uniform; the zero padding makes segments >= nseg contribute 0
            qs[i] = (a[i].x + a[i].y) + (a[i].z + a[i].w);
            qt += qs[i];
        }
    }
    const float incl = wave_incl_scan_f32(qt);
    if (lane == 63) w_tot[wave] = incl;
    sync();
    SS_STAMP(1);
    float P = start + (incl - qt);
    for (int w = 0; w < wave; ++w) P += w_tot[w];

    // ---- translation D of each segment from two representative starts (even / odd mantissa)
    uint32_t ev[SS_MAX_SPT], ndp[SS_MAX_SPT];
    uint32_t hmask = 0, ndt = 0;
#pragma unroll
    for (int i = 0; i < SS_MAX_SPT; ++i) {
        ev[i] = 0; ndp[i] = ndt;
        if (i < spt && seg0 + i < nseg) {
            const uint32_t rb = f2u(P) & ~1u, e = rb >> 23;
            bool hard = false;
            uint32_t nd = 0;
            if (seg0 + i == 0 || e <= 40u || e >= 250u) hard = true;
            else {
                const float R0 = u2f(rb), R1 = u2f(rb | 1u);
                float E0 = R0, E1 = R1;
                E0 = E0 + a[i].x; E1 = E1 + a[i].x; E0 = E0 + a[i].y; E1 = E1 + a[i].y;
                E0 = E0 + a[i].z; E1 = E1 + a[i].z; E0 = E0 + a[i].w; E1 = E1 + a[i].w;
                const float D0 = E0 - R0, D1 = E1 - R1;
                const float margin = u2f((e - 23u + 13u) << 23);         // 8192 ulp: keeps the fallback rare
                if (!(D0 == D1) || (f2u(E0) >> 23) != e || (f2u(E1) >> 23) != e || (f2u(R0 - margin) >> 23) != e ||
                    (f2u(E0 + margin) >> 23) != e)
                    hard = true;
                else nd = (uint32_t)(D0 * u2f((277u - e) << 23));        // D / ulp, exact integer < 2^24
            }
            ev[i] = e | (hard ? 0x80000000u : 0u);
            es[seg0 + i] = ev[i];
            hmask |= (hard ? 1u : 0u) << i;
            ndt += nd;
            ndp[i] = ndt;
            P += qs[i];
        }
    }
    SS_STAMP(2);
    // ---- inclusive prefix of nd (mod 2^32: only differences inside one run are used) + hard list
    const uint32_t pin = wave_incl_scan_u32(ndt);
    const uint32_t hc = (uint32_t)__popc(hmask);
    const uint32_t hin = wave_incl_scan_u32(hc);
    if (lane == 63) { w_nd[wave] = pin; w_cnt[wave] = (int)hin; }
    sync();
    SS_STAMP(3);
    uint32_t nbase = 0; int hbase = 0;
    for (int w = 0; w < wave; ++w) { nbase += w_nd[w]; hbase += w_cnt[w]; }
    {
        const uint32_t nb0 = nbase + pin - ndt;
        int hpos = hbase + (int)(hin - hc);
#pragma unroll
        for (int i = 0; i < SS_MAX_SPT; ++i) {
            if (i < spt && seg0 + i < nseg) {
                pre[seg0 + i] = nb0 + ndp[i];
                if ((hmask >> i) & 1u) hlist[hpos++] = seg0 + i;
                else if (i > 0) { if (!(ev[i - 1] >> 31) && (ev[i - 1] & 0x7FFFFFFFu) != (ev[i] & 0x7FFFFFFFu)) misc[0] = 1; }
            }
        }
    }
    // first segment of the thread vs its left neighbour: es[] was complete at the previous sync(), and the flag must be
    // set BEFORE the next one — wave 0 reads misc[0] right after it
    if (seg0 > 0 && seg0 < nseg && !(ev[0] >> 31)) {
        const uint32_t ep = es[seg0 - 1];
        if (!(ep >> 31) && ep != ev[0]) misc[0] = 1;
    }
    sync();
    SS_STAMP(4);
    // ---- replay the hard segments in order (one wavefront), see replay_events
    if (wave == 0) {
        const int nhard = w_cnt[0] + w_cnt[1] + w_cnt[2] + w_cnt[3];
        float base = start;
        int fail = 0;
        replay_events<1, SQ>(x, 4, nseg, nhard, es, pre, hlist, base, fail);
        if (fail || misc[0]) base = naive_sumsq_lds<SQ>(x, 0, n, start);       // never expected: plain chain
        if (lane == 0) result[0] = base;
        SS_STAMP(6);
    }
    sync();
    SS_STAMP(7);
    return result[0];
}

template <typename Sync>
__device__ __forceinline__ float exact_sumsq_lds(const float* x, int n, uint8_t* scratch, const int t, Sync& sync) {
    return exact_seqsum_lds<true>(x, n, scratch, t, sync, 0.f);
}

}  // namespace gl3
