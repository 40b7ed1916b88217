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
// Categorical sampling is entirely on the device (4 bytes come back instead of vocab * 4).  Top-p copies the probabilities
// to the host and runs the reference's heap selection there (native, same sift order and therefore the same choice among
// equal probabilities); it is the softmax — the expensive part on the Java side — that moves to the device.
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

static int32_t sample_alloc_all(gl3_ctx* ctx) {
    const int nchunks = (ctx->d.vocab + SM_CHUNK - 1) / SM_CHUNK;
    GL3_HIP(hipMalloc((void**)&ctx->sm_probs, (size_t)ctx->d.vocab * 4));
    GL3_HIP(hipMalloc((void**)&ctx->sm_aux, (size_t)(SM_BLOCKS + nchunks + 8) * 4));
    GL3_HIP(hipHostMalloc((void**)&ctx->h_probs, (size_t)ctx->d.vocab * 4));
    GL3_HIP(hipFuncSetAttribute((const void*)smp_seqsum_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
    GL3_HIP(hipFuncSetAttribute((const void*)smp_seqsum_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
    return GL3_OK;
}

// All three buffers or none: a partial allocation (out of memory half way) is released, so the next call starts over instead of
// launching kernels on null pointers.
int32_t gl3_sample_alloc(gl3_ctx* ctx) {
    if (ctx->sm_probs && ctx->sm_aux && ctx->h_probs) return GL3_OK;
    gl3_sample_free(ctx);
    const int32_t r = sample_alloc_all(ctx);
    if (r != GL3_OK) gl3_sample_free(ctx);
    return r;
}

void gl3_sample_free(gl3_ctx* ctx) {
    if (ctx->sm_probs) hipFree(ctx->sm_probs);
    if (ctx->sm_aux) hipFree(ctx->sm_aux);
    if (ctx->h_probs) hipHostFree(ctx->h_probs);
    ctx->sm_probs = nullptr; ctx->sm_aux = nullptr; ctx->h_probs = nullptr;
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
    GL3_HIP(hipMemcpyAsync(ctx->h_probs, ctx->sm_probs, (size_t)n * 4, hipMemcpyDeviceToHost, s));
    GL3_HIP(hipStreamSynchronize(s));
    *token_out = topp_sample(ctx->h_probs, n, topp, coin, ctx->topp_indices);
    return GL3_OK;
}

int32_t gl3_sample_probs(gl3_ctx* ctx, float* out) {       // parity tap: the probabilities of the last sampled step
    if (!ctx->sm_probs) GL3_FAIL(GL3_E_STATE, "no sampled step yet");
    GL3_HIP(hipMemcpy(out, ctx->sm_probs, (size_t)ctx->d.vocab * 4, hipMemcpyDeviceToHost));
    return GL3_OK;
}
