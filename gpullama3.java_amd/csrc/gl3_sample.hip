// gl3_sample.hip — temperature / top-p sampling behind the decode step.
//
// Replaces, for the HIP path, Sampler.selectSampler's lambda (J/inference/sampler/Sampler.java:76-123):
//     logits.divideInPlace(temperature); logits.softmaxInPlace();                 FloatTensor.java:203-219
//     CategoricalSampler.sampleToken (J/inference/sampler/CategoricalSampler.java:33-44)   or
//     ToppSampler.sampleToken       (J/inference/sampler/ToppSampler.java:57-160)
// with the reference's arithmetic: f / temperature, max, (float)Math.exp(f - max) in double, the STRICTLY SEQUENTIAL f32 sum
// of all vocab numerators (FloatTensor.sum = reduce(0f, Float::sum)), f / sum, and the sequential f32 cdf of the sampler.
// The 128 k-long sequential sums run on the device with the exact parallel evaluation of gl3_seqsum.h, 4096 elements at a
// time, each chunk starting from the exact running value of the previous one.
//
// The random number stays the CALLER's: `coin` is rng.nextFloat(1f) drawn from the host's RandomGenerator
// (RandomGeneratorFactory.getDefault().create(seed), Sampler.java:84) exactly where the reference draws it — one per sampled
// token — so the stream of random numbers, and with it the sampled ids, are the reference's by construction.
//
// Categorical sampling is entirely on the device (4 bytes come back instead of vocab * 4).  Top-p (r5) too, whenever the answer
// does not hinge on a tie: the reference's heap selection (ToppSampler.java:118-160) emits the candidates in non-increasing VALUE order —
// its one quirk, siftDown(indices, 0, i - 1, ..) leaving the last leaf out of every sift, never lets a larger value wait behind a
// smaller one (the leaf left out is <= the value its parent had, and that value is still in the heap) — so the truncation point, the
// renormalised coin and the RANK the coin lands on are those of a descending sort, evaluated here with a radix sort + the exact chunked
// f32 prefix sums; only WHICH index stands at a rank shared by equal probabilities is decided by the heap's sift history, which no sort
// order reproduces.  The device path returns the token and a tie flag (8 bytes); on a tie at the sampled rank (~2 % of the draws on the
// near-uniform distributions of random-weight test models, rarer on real ones) the probabilities are copied out and the reference's heap
// runs on the host (same sift order, same choice).
#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>

#include "gl3_ctx.h"
#include "gl3_decode_kernels.h"

using namespace gl3;

constexpr int SM_BLOCKS = 256, SM_CHUNK = 4096;

__global__ __launch_bounds__(256) void smp_scale_max_kernel(const float* __restrict__ logits, int n, float temperature, float* __restrict__ p,
                                                            float* __restrict__ blockmax) {
    __shared__ float red[4];
    float mx = -INFINITY;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        const float v = logits[i] / temperature;                       // divideInPlace(temperature)
        p[i] = v;
        mx = fmaxf(mx, v);
    }
    mx = wave_max(mx);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
    __syncthreads();
    if (threadIdx.x == 0) blockmax[blockIdx.x] = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}

__global__ __launch_bounds__(256) void smp_exp_kernel(float* __restrict__ p, int n, const float* __restrict__ blockmax, int nblocks) {
    __shared__ float red[4];
    float mx = -INFINITY;
    for (int i = threadIdx.x; i < nblocks; i += 256) mx = fmaxf(mx, blockmax[i]);
    mx = wave_max(mx);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) p[i] = (float)exp((double)(p[i] - mx));   // (float) Math.exp(f - maxVal)
}

// Strictly sequential f32 sum of p[0..n) (all >= 0), one workgroup: chunks of SM_CHUNK through LDS, exact parallel evaluation
// per chunk (gl3_seqsum.h) continued from the exact running value.  chunk_end[c] = running sum after chunk c (the cdf at the
// chunk boundaries).  With pick = true the kernel then samples: first index whose cdf exceeds coin (CategoricalSampler).
template <bool PICK>
__global__ __launch_bounds__(256) void smp_seqsum_kernel(const float* __restrict__ p, int n, float* __restrict__ total, float* __restrict__ chunk_end,
                                                         float coin, int* __restrict__ picked) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    float* xf = reinterpret_cast<float*>(smem);                      // [SM_CHUNK + 32]
    uint8_t* scratch = smem + (size_t)(SM_CHUNK + 32) * 4;
    __shared__ float run_s;
    __shared__ int hit_s;
    const int t = threadIdx.x;
    if (t == 0) { run_s = 0.f; hit_s = -1; }
    __syncthreads();
    const int nchunks = (n + SM_CHUNK - 1) / SM_CHUNK;
    for (int c = 0; c < nchunks; ++c) {
        const int base = c * SM_CHUNK, len = min(SM_CHUNK, n - base);
        for (int i = t; i < SM_CHUNK + 32; i += 256) xf[i] = i < len ? p[base + i] : 0.f;
        __syncthreads();
        float run = run_s;
        const int n4 = len & ~3;
        if (n4 >= 1024) {
            BlockBarrier bb;
            run = exact_seqsum_lds<false>(xf, n4, scratch, t, bb, run);
            if (n4 < len && t < 64) run = naive_sumsq_lds<false>(xf, n4, len, run);      // at most 3 trailing elements
        } else if (t < 64) {
            run = naive_sumsq_lds<false>(xf, 0, len, run);
        }
        __syncthreads();
        if (t == 0) {
            if (PICK && hit_s < 0 && coin < run) hit_s = c;           // the cdf is non-decreasing: the first chunk whose end exceeds coin
            run_s = run;
            chunk_end[c] = run;
        }
        __syncthreads();
        if (PICK && hit_s == c) {
            // cdf += p[i]; if (coin < cdf) return i   (CategoricalSampler.java:37-42), continued inside the chunk from its exact start
            if (t == 0) {
                float cdf = c ? chunk_end[c - 1] : 0.f;
                int idx = -1;
                for (int i = 0; i < len; ++i) { cdf = cdf + xf[i]; if (coin < cdf) { idx = base + i; break; } }
                *picked = idx >= 0 ? idx : base + len - 1;
            }
            break;
        }
    }
    __syncthreads();
    if (t == 0) {
        if (total) *total = run_s;
        if (PICK && hit_s < 0) *picked = n - 1;                       // "in case of rounding errors"
    }
}

__global__ __launch_bounds__(256) void smp_div_kernel(float* __restrict__ p, int n, const float* __restrict__ total) {
    const float s = *total;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) p[i] = p[i] / s;     // divideInPlace(sum)
}

// ToppSampler (J/inference/sampler/ToppSampler.java:57-160) on host probabilities — same cutoff, same heap build / sift order,
// same cumulative f32 sums, so the same index also when probabilities tie.
static void sift_down(int* array, int from, int n, const float* v) {
    auto cmp = [&](int a, int b) {          // Comparator.comparingDouble(getFloat).reversed(): negative when a's value is LARGER
        const double da = v[a], db = v[b];
        return db < da ? -1 : db > da ? 1 : 0;
    };
    int prev = from, next;
    while ((next = 2 * prev + 1) < n) {
        const int r = 2 * prev + 2;
        if (r < n && cmp(array[r], array[next]) < 0) next = r;
        if (cmp(array[next], array[prev]) < 0) { std::swap(array[prev], array[next]); prev = next; }
        else break;
    }
}

static int topp_sample(const float* p, int n, float topp, float coin, std::vector<int>& indices) {
    indices.resize(n);
    int head = 0, tail = n - 1;
    const float cutoff = (1.0f - topp) / (float)(n - 1);
    for (int i = 0; i < n; ++i) {
        if (p[i] >= cutoff) indices[head++] = i;
        else indices[tail--] = i;
    }
    const int n0 = head;
    int* idx = indices.data();
    for (int i = n0 / 2 - 1; i >= 0; --i) sift_down(idx, i, n0, p);
    float cumulative = 0.0f;
    int last = 0;
    for (int i = n0 - 1; i >= 0; --i) {
        std::swap(idx[0], idx[i]);
        cumulative += p[idx[i]];
        if (cumulative > topp) { last = i; break; }
        sift_down(idx, 0, i - 1, p);
    }
    const float r = coin * cumulative;
    float cdf = 0.0f;
    for (int i = n0 - 1; i >= last; --i) {
        cdf += p[idx[i]];
        if (r < cdf) return idx[i];
    }
    return idx[last];
}

// ------------------------------------------------------------------------------------------------ top-p on the device
// Keys: candidates (p >= cutoff, ToppSampler.java:74-81) get ~bits(p) (positive floats order like their bit patterns, so ascending keys
// = descending probabilities), everything else the maximal key; a stable LSD radix sort (4 x 8 bits) of (key, index) pairs puts the n0
// candidates first in descending order.
constexpr int RS_THREADS = 256, RS_PER = 4, RS_TILE = RS_THREADS * RS_PER;

__global__ __launch_bounds__(RS_THREADS) void topp_keys_kernel(const float* __restrict__ p, int n, float cutoff, uint32_t* __restrict__ keys, int* __restrict__ idx,
                                                                int* __restrict__ n0) {
    __shared__ int cnt_s;
    if (threadIdx.x == 0) cnt_s = 0;
    __syncthreads();
    int c = 0;
    for (int i = blockIdx.x * RS_THREADS + threadIdx.x; i < n; i += gridDim.x * RS_THREADS) {
        const float v = p[i];
        const bool cand = v >= cutoff;
        keys[i] = cand ? ~__builtin_bit_cast(uint32_t, v) : 0xFFFFFFFFu;
        idx[i] = i;
        c += cand ? 1 : 0;
    }
    atomicAdd(&cnt_s, c);
    __syncthreads();
    if (threadIdx.x == 0 && cnt_s) atomicAdd(n0, cnt_s);
}

// hist[d * nblocks + b] = elements of tile b whose digit is d
__global__ __launch_bounds__(RS_THREADS) void radix_hist_kernel(const uint32_t* __restrict__ keys, int n, int shift, int* __restrict__ hist, int nblocks) {
    __shared__ int h[256];
    h[threadIdx.x] = 0;
    __syncthreads();
#pragma unroll
    for (int r = 0; r < RS_PER; ++r) {
        const int e = blockIdx.x * RS_TILE + r * RS_THREADS + threadIdx.x;
        if (e < n) atomicAdd(&h[(keys[e] >> shift) & 255u], 1);
    }
    __syncthreads();
    hist[threadIdx.x * nblocks + blockIdx.x] = h[threadIdx.x];
}

// exclusive prefix over hist in (digit, block) order, in place; one workgroup of 1024 threads
__global__ __launch_bounds__(1024) void radix_scan_kernel(int* __restrict__ hist, int total) {
    __shared__ int part[1024];
    const int t = threadIdx.x, per = (total + 1023) / 1024, lo = t * per, hi = min(total, lo + per);
    int s = 0;
    for (int i = lo; i < hi; ++i) s += hist[i];
    part[t] = s;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {
        const int v = t >= d ? part[t - d] : 0;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    int run = part[t] - s;
    for (int i = lo; i < hi; ++i) { const int v = hist[i]; hist[i] = run; run += v; }
}

// stable scatter of tile blockIdx.x: element order inside a tile is (round, wavefront, lane) = ascending index
__global__ __launch_bounds__(RS_THREADS) void radix_scatter_kernel(const uint32_t* __restrict__ kin, const int* __restrict__ iin, uint32_t* __restrict__ kout,
                                                                    int* __restrict__ iout, int n, int shift, const int* __restrict__ hist, int nblocks) {
    __shared__ int base[256];                  // next output slot of digit d for this tile
    __shared__ int wcnt[4][256];               // per wavefront: elements of digit d in the current round
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    base[t] = hist[t * nblocks + blockIdx.x];
#pragma unroll
    for (int r = 0; r < RS_PER; ++r) {
#pragma unroll
        for (int w = 0; w < 4; ++w) wcnt[w][t] = 0;
        __syncthreads();
        const int e = blockIdx.x * RS_TILE + r * RS_THREADS + t;
        const bool valid = e < n;
        const uint32_t key = valid ? kin[e] : 0u;
        const int id = valid ? iin[e] : 0;
        const uint32_t d = (key >> shift) & 255u;
        // lanes of my wavefront holding the same digit (8 ballots); invalid lanes match nobody
        unsigned long long same = __ballot(valid);
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            const unsigned long long vote = __ballot(valid && ((d >> b) & 1u));
            same &= ((d >> b) & 1u) ? vote : ~vote;
        }
        const int before = __popcll(same & ((1ull << lane) - 1ull));
        if (valid && before == 0) wcnt[wave][d] = __popcll(same);
        __syncthreads();
        if (valid) {
            int off = base[d] + before;
            for (int w = 0; w < wave; ++w) off += wcnt[w][d];
            kout[off] = key;
            iout[off] = id;
        }
        __syncthreads();
        base[t] += wcnt[0][t] + wcnt[1][t] + wcnt[2][t] + wcnt[3][t];
        __syncthreads();
    }
}

// ToppSampler.processTopP :118-160 on the sorted candidates sv[0 .. n0) (sv[r] = probability at rank r, descending):
//   cumulativeProb += value (f32, in order) until it EXCEEDS topp -> last rank (rank n0 - 1 if it never does);
//   r = coin * cumulativeProb;  cdf += value from rank 0: the first rank with r < cdf, bounded by the last rank.
// The two strictly sequential prefix scans run 4096 ranks at a time with the exact parallel sum (gl3_seqsum.h) and walk only the
// chunk in which the threshold falls.  out[0] = index at the chosen rank, out[1] = 1 if another candidate has the same probability
// (then the reference's heap order, not this sort order, names the token: the host re-runs it).
__global__ __launch_bounds__(256) void topp_pick_kernel(const uint32_t* __restrict__ skeys, const int* __restrict__ sidx, const int* __restrict__ n0p, float topp, float coin,
                                                        int* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    float* xf = reinterpret_cast<float*>(smem);                      // [SM_CHUNK + 32]
    uint8_t* scratch = smem + (size_t)(SM_CHUNK + 32) * 4;
    __shared__ float run_s, thr_s, cum_s;
    __shared__ int rank_s, last_s;
    const int t = threadIdx.x;
    const int n0 = *n0p;
    if (n0 <= 0) { if (t == 0) { out[0] = 0; out[1] = 1; } return; }      // no candidate (cannot happen for a normalised row): let the host decide
    if (t == 0) { thr_s = topp; last_s = n0 - 1; cum_s = 0.f; }
    for (int phase = 0; phase < 2; ++phase) {
        if (t == 0) { run_s = 0.f; rank_s = -1; }
        __syncthreads();
        const float thr = thr_s;
        const int limit = phase == 0 ? n0 : last_s + 1;              // phase 1 never looks past the truncation point
        for (int base = 0; base < limit; base += SM_CHUNK) {
            const int len = min(SM_CHUNK, limit - base);
            for (int i = t; i < SM_CHUNK + 32; i += 256) xf[i] = i < len ? __builtin_bit_cast(float, ~skeys[base + i]) : 0.f;
            __syncthreads();
            const float start = run_s;
            float run = start;
            const int n4 = len & ~3;
            if (n4 >= 1024) {
                BlockBarrier bb;
                run = exact_seqsum_lds<false>(xf, n4, scratch, t, bb, run);
                if (n4 < len && t < 64) run = naive_sumsq_lds<false>(xf, n4, len, run);
            } else if (t < 64) {
                run = naive_sumsq_lds<false>(xf, 0, len, run);
            }
            __syncthreads();
            if (t == 0) {
                if (thr < run) {                                       // the prefix is non-decreasing: the threshold falls in this chunk
                    float cdf = start;
                    int hit = len - 1;
                    for (int i = 0; i < len; ++i) { cdf = cdf + xf[i]; if (thr < cdf) { hit = i; break; } }
                    rank_s = base + hit;
                    run = cdf;
                }
                run_s = run;
            }
            __syncthreads();
            if (rank_s >= 0) break;
        }
        if (t == 0) {
            if (phase == 0) {
                if (rank_s >= 0) last_s = rank_s;                      // cumulativeProb > topp at this rank (its value included)
                cum_s = run_s;                                         // else: every candidate, lastIndex = 0 in the reference
                thr_s = coin * cum_s;                                  // rng.nextFloat(1f) * cumulativeProb
            } else if (rank_s < 0) rank_s = last_s;                    // "in case of rounding errors"
        }
        __syncthreads();
    }
    if (t == 0) {
        const int r = rank_s;
        const uint32_t k = skeys[r];
        const bool tie = (r > 0 && skeys[r - 1] == k) || (r + 1 < n0 && skeys[r + 1] == k);
        out[0] = sidx[r];
        out[1] = tie ? 1 : 0;
    }
}

static int32_t sample_alloc_all(gl3_ctx* ctx) {
    const int nchunks = (ctx->d.vocab + SM_CHUNK - 1) / SM_CHUNK;
    GL3_HIP(hipMalloc((void**)&ctx->sm_probs, (size_t)ctx->d.vocab * 4));
    GL3_HIP(hipMalloc((void**)&ctx->sm_aux, (size_t)(SM_BLOCKS + nchunks + 8) * 4));
    GL3_HIP(hipHostMalloc((void**)&ctx->h_probs, (size_t)ctx->d.vocab * 4));
    {   // top-p on the device: two (key, index) buffers, the radix histogram, {n0, token, tie}
        const int nb = (ctx->d.vocab + RS_TILE - 1) / RS_TILE;
        GL3_HIP(hipMalloc((void**)&ctx->sm_sort, ((size_t)4 * ctx->d.vocab + (size_t)256 * nb + 8) * 4));
    }
    GL3_HIP(hipFuncSetAttribute((const void*)topp_pick_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
    GL3_HIP(hipFuncSetAttribute((const void*)smp_seqsum_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
    GL3_HIP(hipFuncSetAttribute((const void*)smp_seqsum_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
    return GL3_OK;
}

// All three buffers or none: a partial allocation (out of memory half way) is released, so the next call starts over instead of
// launching kernels on null pointers.
int32_t gl3_sample_alloc(gl3_ctx* ctx) {
    if (ctx->sm_probs && ctx->sm_aux && ctx->h_probs && ctx->sm_sort) return GL3_OK;
    gl3_sample_free(ctx);
    const int32_t r = sample_alloc_all(ctx);
    if (r != GL3_OK) gl3_sample_free(ctx);
    return r;
}

void gl3_sample_free(gl3_ctx* ctx) {
    if (ctx->sm_probs) hipFree(ctx->sm_probs);
    if (ctx->sm_aux) hipFree(ctx->sm_aux);
    if (ctx->h_probs) hipHostFree(ctx->h_probs);
    if (ctx->sm_sort) hipFree(ctx->sm_sort);
    ctx->sm_probs = nullptr; ctx->sm_aux = nullptr; ctx->h_probs = nullptr; ctx->sm_sort = nullptr;
}

// logits (device, f32[vocab], complete on this rank) -> sampled id.  temperature > 0.
int32_t gl3_sample_run(gl3_ctx* ctx, const float* logits_dev, float temperature, float topp, float coin, int32_t* token_out) {
    int32_t r = gl3_sample_alloc(ctx);
    if (r != GL3_OK) return r;
    const int n = ctx->d.vocab;
    hipStream_t s = ctx->stream;
    float* blockmax = ctx->sm_aux;
    float* total = ctx->sm_aux + SM_BLOCKS;
    int* picked = reinterpret_cast<int*>(ctx->sm_aux + SM_BLOCKS + 1);
    float* chunk_end = ctx->sm_aux + SM_BLOCKS + 8;
    const size_t smem = (size_t)(SM_CHUNK + 32) * 4 + ss_scratch_bytes(SM_CHUNK);
    hipLaunchKernelGGL(smp_scale_max_kernel, dim3(SM_BLOCKS), dim3(256), 0, s, logits_dev, n, temperature, ctx->sm_probs, blockmax);
    hipLaunchKernelGGL(smp_exp_kernel, dim3(SM_BLOCKS), dim3(256), 0, s, ctx->sm_probs, n, blockmax, SM_BLOCKS);
    hipLaunchKernelGGL(smp_seqsum_kernel<false>, dim3(1), dim3(256), smem, s, ctx->sm_probs, n, total, chunk_end, 0.f, picked);
    hipLaunchKernelGGL(smp_div_kernel, dim3(SM_BLOCKS), dim3(256), 0, s, ctx->sm_probs, n, total);
    GL3_HIP(hipGetLastError());
    const bool use_topp = topp > 0.f && topp < 1.f;                  // Sampler.java:88-98
    if (!use_topp) {
        hipLaunchKernelGGL(smp_seqsum_kernel<true>, dim3(1), dim3(256), smem, s, ctx->sm_probs, n, (float*)nullptr, chunk_end, coin, picked);
        GL3_HIP(hipGetLastError());
        GL3_HIP(hipMemcpyAsync(ctx->h_argmax, picked, sizeof(int), hipMemcpyDeviceToHost, s));
        GL3_HIP(hipStreamSynchronize(s));
        *token_out = *ctx->h_argmax;
        return GL3_OK;
    }
    // ---- top-p on the device: candidates -> descending radix sort -> truncation, renormalised coin, rank (8 bytes come back)
    static const bool host_topp = env_flag("GL3_TOPP_HOST", false);              // A/B switch: the r4 path (probabilities to the host)
    if (!host_topp) {
        const int nb = (n + RS_TILE - 1) / RS_TILE;
        uint32_t* ka = reinterpret_cast<uint32_t*>(ctx->sm_sort);
        int* ia = reinterpret_cast<int*>(ka + n);
        uint32_t* kb = reinterpret_cast<uint32_t*>(ia + n);
        int* ib = reinterpret_cast<int*>(kb + n);
        int* hist = ib + n;
        int* res = hist + (size_t)256 * nb;                               // [0] n0, [1] token, [2] tie
        const float cutoff = (1.0f - topp) / (float)(n - 1);             // ToppSampler.java:73
        GL3_HIP(hipMemsetAsync(res, 0, 3 * sizeof(int), s));
        hipLaunchKernelGGL(topp_keys_kernel, dim3(SM_BLOCKS), dim3(RS_THREADS), 0, s, ctx->sm_probs, n, cutoff, ka, ia, res);
        for (int pass = 0; pass < 4; ++pass) {
            uint32_t* kin = pass & 1 ? kb : ka; int* iin = pass & 1 ? ib : ia;
            uint32_t* kout = pass & 1 ? ka : kb; int* iout = pass & 1 ? ia : ib;
            hipLaunchKernelGGL(radix_hist_kernel, dim3(nb), dim3(RS_THREADS), 0, s, kin, n, 8 * pass, hist, nb);
            hipLaunchKernelGGL(radix_scan_kernel, dim3(1), dim3(1024), 0, s, hist, 256 * nb);
            hipLaunchKernelGGL(radix_scatter_kernel, dim3(nb), dim3(RS_THREADS), 0, s, kin, iin, kout, iout, n, 8 * pass, hist, nb);
        }
        hipLaunchKernelGGL(topp_pick_kernel, dim3(1), dim3(256), smem, s, ka, ia, res, topp, coin, res + 1);       // 4 passes: the result is back in (ka, ia)
        GL3_HIP(hipGetLastError());
        GL3_HIP(hipMemcpyAsync(ctx->h_dyn + 2, res + 1, 2 * sizeof(int), hipMemcpyDeviceToHost, s));
        GL3_HIP(hipStreamSynchronize(s));
        if (!ctx->h_dyn[3]) { *token_out = ctx->h_dyn[2]; ++ctx->topp_device; return GL3_OK; }
        // a tie at the sampled rank: the reference's heap history decides between equal probabilities — run it
    }
    GL3_HIP(hipMemcpyAsync(ctx->h_probs, ctx->sm_probs, (size_t)n * 4, hipMemcpyDeviceToHost, s));
    GL3_HIP(hipStreamSynchronize(s));
    *token_out = topp_sample(ctx->h_probs, n, topp, coin, ctx->topp_indices);
    ++ctx->topp_host;
    return GL3_OK;
}

int32_t gl3_sample_probs(gl3_ctx* ctx, float* out) {       // parity tap: the probabilities of the last sampled step
    if (!ctx->sm_probs) GL3_FAIL(GL3_E_STATE, "no sampled step yet");
    GL3_HIP(hipMemcpy(out, ctx->sm_probs, (size_t)ctx->d.vocab * 4, hipMemcpyDeviceToHost));
    return GL3_OK;
}
