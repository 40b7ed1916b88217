// anyorder_probe.hip — does hipExtLaunchKernel(..., hipExtAnyOrderLaunch) clear the AQL barrier bit on gfx950 (so that the
// command processor starts dispatching kernel k + 1 while kernel k's last workgroups are still running), and what does a
// dependent-launch boundary cost (a) with the queue barrier and (b) with the barrier bit cleared + an in-kernel wait on a
// device counter that the workgroups of the previous kernel bump when they finish?
//   1. overlap test: A = 512 workgroups that each spin ~40 us, B launched any-order right behind it; B's first start stamp
//      against A's last end stamp (s_memrealtime, 100 MHz).
//   2. boundary cost: a chain of N kernels of 512 x 512 threads, each workgroup does ~2 us of dependent work:
//      plain launches vs any-order launches with the counter wait (the pattern gl3's decode step would use).
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <chrono>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void spin_kernel(long long* stamps, int ticks) {
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(4);
    if (threadIdx.x == 0) { stamps[2 * blockIdx.x] = t0; stamps[2 * blockIdx.x + 1] = wall_clock64(); }
}

// one link of a dependent chain: wait until `done` has reached wait_for (all workgroups of every earlier link have finished),
// do a little work on buf, publish, bump the counter.  wait_for < 0: no wait (the queue barrier orders the links).
__global__ __launch_bounds__(512) void link_kernel(float* buf, unsigned* done, int wait_for, int spin_limit, unsigned* err, int work) {
    __shared__ float red[8];
    if (wait_for >= 0) {
        if (threadIdx.x == 0) {
            int spins = 0;
            while ((int)(__hip_atomic_load(done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - (unsigned)wait_for) < 0) {
                __builtin_amdgcn_s_sleep(2);
                if (++spins > spin_limit) { *err = 1; break; }
            }
        }
        __syncthreads();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    float v = buf[(blockIdx.x * 512 + threadIdx.x) & 4095];
    for (int i = 0; i < work; ++i) v = v * 1.0000001f + 1e-7f;
    v += __shfl_xor(v, 1, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        float s = 0;
        for (int i = 0; i < 8; ++i) s += red[i];
        buf[4096 + blockIdx.x] = s;
    }
    if (wait_for >= 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_fetch_add(done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// r6 (review item 7): the hand-over the r5 table argued away on paper — per-XCD counters instead of ONE word.  Every workgroup of link k bumps
// done8[16 * xcc_id] (its own XCD's word, 64 bytes apart); lanes 0..7 of the next link's workgroups poll the 8 words.  FENCE = 0 leaves out the agent-scope
// release / acquire fences (NOT a valid hand-over for data: it isolates the price of the counter traffic from the price of the fences).
template <int FENCE>
__global__ __launch_bounds__(512) void link8_kernel(float* buf, unsigned* done8, int wait_for /* per XCD */, int spin_limit, unsigned* err, int work) {
    __shared__ float red[8];
    if (wait_for >= 0) {
        if (threadIdx.x < 64) {
            const int l = threadIdx.x;
            int spins = 0;
            for (;;) {
                const bool ok = l >= 8 || (int)(__hip_atomic_load(done8 + 16 * l, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - (unsigned)wait_for) >= 0;
                if (__builtin_amdgcn_ballot_w64(!ok) == 0) break;
                __builtin_amdgcn_s_sleep(2);
                if (++spins > spin_limit) { *err = 1; break; }
            }
        }
        __syncthreads();
        if (FENCE) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    float v = buf[(blockIdx.x * 512 + threadIdx.x) & 4095];
    for (int i = 0; i < work; ++i) v = v * 1.0000001f + 1e-7f;
    v += __shfl_xor(v, 1, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        float s = 0;
        for (int i = 0; i < 8; ++i) s += red[i];
        buf[4096 + blockIdx.x] = s;
    }
    if (wait_for >= 0) {
        if (FENCE) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __syncthreads();
        if (threadIdx.x == 0) {
            unsigned xcc;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            __hip_atomic_fetch_add(done8 + 16 * (xcc & 7), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

int main() {
    hipStream_t s;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    // ---------------------------------------------------------------- 1. overlap
    for (int nA : {512, 2048, 64}) {
        const int nB = 512;
        long long *sa, *sb;
        CK(hipMalloc((void**)&sa, 16 * nA)); CK(hipMalloc((void**)&sb, 16 * nB));
        std::vector<long long> ha(2 * nA), hb(2 * nB);
        for (int flags : {0, (int)hipExtAnyOrderLaunch}) {
            int ticksA = 4000, ticksB = 200;          // 40 us, 2 us
            void* argsA[] = {&sa, &ticksA};
            void* argsB[] = {&sb, &ticksB};
            for (int rep = 0; rep < 2; ++rep) {
                CK(hipExtLaunchKernel((const void*)spin_kernel, dim3(nA), dim3(256), argsA, 0, s, nullptr, nullptr, 0));
                CK(hipExtLaunchKernel((const void*)spin_kernel, dim3(nB), dim3(256), argsB, 0, s, nullptr, nullptr, flags));
                CK(hipStreamSynchronize(s));
            }
            CK(hipMemcpy(ha.data(), sa, 16 * nA, hipMemcpyDeviceToHost)); CK(hipMemcpy(hb.data(), sb, 16 * nB, hipMemcpyDeviceToHost));
            long long a0 = ha[0], a1 = ha[1], b0 = hb[0];
            for (int i = 0; i < nA; ++i) { a0 = std::min(a0, ha[2 * i]); a1 = std::max(a1, ha[2 * i + 1]); }
            for (int i = 0; i < nB; ++i) b0 = std::min(b0, hb[2 * i]);
            printf("overlap A=%4d wgs x 40us, B=512 wgs, flags=%d: A runs %.2f us; B's first workgroup starts %+.2f us relative to A's last end  -> %s\n", nA, flags,
                   (a1 - a0) / 100.0, (b0 - a1) / 100.0, b0 < a1 ? "OVERLAP" : "serial");
        }
        hipFree(sa); hipFree(sb);
    }
    // ---------------------------------------------------------------- 2. boundary cost of a dependent chain
    float* buf; unsigned *done, *err;
    CK(hipMalloc((void**)&buf, 8192 * 4)); CK(hipMalloc((void**)&done, 4)); CK(hipHostMalloc((void**)&err, 4));
    CK(hipMemset(buf, 0, 8192 * 4)); CK(hipMemset(done, 0, 4)); *err = 0;
    const int N = 2000, WG = 512;
    for (int work : {0, 2000}) {
        for (int mode = 0; mode < 3; ++mode) {        // 0 plain launches, 1 any-order + counter, 2 plain launches + counter (cost of the counter alone)
            unsigned base = 0;
            CK(hipMemset(done, 0, 4));
            CK(hipDeviceSynchronize());
            hipEvent_t e0, e1;
            CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
            int limit = 4000000;
            const auto h0 = std::chrono::steady_clock::now();
            CK(hipEventRecord(e0, s));
            for (int i = 0; i < N; ++i) {
                int wait_for = mode == 0 ? -1 : (int)base;
                void* args[] = {&buf, &done, &wait_for, &limit, &err, (void*)&work};
                CK(hipExtLaunchKernel((const void*)link_kernel, dim3(WG), dim3(512), args, 0, s, nullptr, nullptr, (mode == 1 && i > 0) ? hipExtAnyOrderLaunch : 0));
                base += WG;
            }
            const auto h1 = std::chrono::steady_clock::now();
            CK(hipEventRecord(e1, s));
            CK(hipEventSynchronize(e1));
            float ms = 0;
            CK(hipEventElapsedTime(&ms, e0, e1));
            printf("chain of %d links (512 wgs x 512 thr, work %d): mode %d (%s): %.2f us per link on the device, host enqueue %.2f us per launch, err %u\n", N, work, mode,
                   mode == 0 ? "queue barrier" : mode == 1 ? "any-order + counter wait" : "queue barrier + counter wait", ms * 1e3 / N,
                   std::chrono::duration<double, std::micro>(h1 - h0).count() / N, *err);
        }
    }
    // ---------------------------------------------------------------- 3. r6: per-XCD counters (workgroup b runs on XCD b % 8: 64 arrivals per word and link)
    unsigned* done8;
    CK(hipMalloc((void**)&done8, 8 * 64));
    for (int work : {0, 2000}) {
        for (int fence = 1; fence >= 0; --fence) {
            CK(hipMemset(done8, 0, 8 * 64));
            CK(hipDeviceSynchronize());
            hipEvent_t e0, e1;
            CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
            int limit = 4000000;
            unsigned base = 0;
            CK(hipEventRecord(e0, s));
            for (int i = 0; i < N; ++i) {
                int wait_for = (int)base;
                void* args[] = {&buf, &done8, &wait_for, &limit, &err, (void*)&work};
                CK(hipExtLaunchKernel(fence ? (const void*)link8_kernel<1> : (const void*)link8_kernel<0>, dim3(WG), dim3(512), args, 0, s, nullptr, nullptr, 0));
                base += WG / 8;
            }
            CK(hipEventRecord(e1, s));
            CK(hipEventSynchronize(e1));
            float ms = 0;
            CK(hipEventElapsedTime(&ms, e0, e1));
            printf("chain of %d links (512 wgs x 512 thr, work %d): per-XCD counters (8 words), %s: %.2f us per link on the device, err %u\n", N, work,
                   fence ? "release + acquire fences" : "NO fences (counter traffic alone; not a valid data hand-over)", ms * 1e3 / N, *err);
        }
    }
    return 0;
}
