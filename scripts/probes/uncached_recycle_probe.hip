// uncached_recycle_probe.hip — minimal form of the round-4 "stale activation rows" fault: inside ONE process, memory that
// was used through the cached mapping (hipMalloc), freed, and handed out again by hipExtMallocWithFlags(hipDeviceMallocUncached)
// (or the other way round) — do kernels then see stale data?  Writes pattern P1 through the first mapping, frees, allocates with the
// other policy until the same address comes back, writes P2, thrashes L2, and counts words that still read P1 / anything but P2.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void fill(unsigned* p, size_t n, unsigned tag) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = tag ^ (unsigned)i;
}
__global__ void check(const unsigned* p, size_t n, unsigned tag, unsigned old_tag, unsigned* bad, unsigned* stale) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const unsigned v = p[i];
        if (v != (tag ^ (unsigned)i)) { atomicAdd(bad, 1u); if (v == (old_tag ^ (unsigned)i)) atomicAdd(stale, 1u); }
    }
}
__global__ void thrash(const float4* p, size_t n, float* sink) {
    float s = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { const float4 v = p[i]; s += v.x + v.w; }
    if (s == 12345.678f) *sink = s;
}

static hipError_t alloc(void** p, size_t bytes, bool uncached) {
    return uncached ? hipExtMallocWithFlags(p, bytes, hipDeviceMallocUncached) : hipMalloc(p, bytes);
}

int main() {
    const size_t bytes = 1 << 20, n = bytes / 4;
    unsigned *bad, *stale; float* sink; float4* big;
    CK(hipMalloc((void**)&bad, 4)); CK(hipMalloc((void**)&stale, 4)); CK(hipMalloc((void**)&sink, 4));
    const size_t bign = (size_t)64 << 20;                     // 1 GiB of float4
    CK(hipMalloc((void**)&big, bign * 16));
    CK(hipMemset(big, 0, bign * 16));
    for (int first_uncached = 0; first_uncached < 2; ++first_uncached) {
        int same = 0, faults = 0;
        for (int rep = 0; rep < 40; ++rep) {
            void* a = nullptr;
            CK(alloc(&a, bytes, first_uncached));
            fill<<<256, 256>>>((unsigned*)a, n, 0x11110000u + rep);
            CK(hipDeviceSynchronize());
            CK(hipFree(a));
            void* b = nullptr;
            CK(alloc(&b, bytes, !first_uncached));
            if (b == a) ++same;
            fill<<<256, 256>>>((unsigned*)b, n, 0x22220000u + rep);
            thrash<<<2048, 256>>>(big, bign, sink);
            CK(hipMemset(bad, 0, 4)); CK(hipMemset(stale, 0, 4));
            check<<<256, 256>>>((const unsigned*)b, n, 0x22220000u + rep, 0x11110000u + rep, bad, stale);
            unsigned hb = 0, hs = 0;
            CK(hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(&hs, stale, 4, hipMemcpyDeviceToHost));
            if (hb) { ++faults; if (faults <= 3) printf("   rep %d: %u of %zu words wrong after re-allocation (%u of them = the FREED buffer's pattern), same address %d\n", rep, hb, n, hs, b == a); }
            CK(hipFree(b));
        }
        printf("%s -> free -> %s: %d of 40 rounds got the same address back, %d rounds read wrong data\n", first_uncached ? "uncached" : "cached",
               first_uncached ? "cached" : "uncached", same, faults);
    }
    return 0;
}
