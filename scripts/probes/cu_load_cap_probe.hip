// How many bytes per clock ONE CU pulls through its vector memory path (global_load_dwordx4 -> VGPRs) on gfx950, by wavefronts per CU, loads in
// flight per wavefront and where the lines come from (L2-resident slices vs an HBM stream).  One workgroup per CU (100 KB of dynamic LDS keeps a second
// one out), W wavefronts each; every wavefront runs U independent 16-byte-per-lane loads per trip over its CU's slice.  The figure the static-batched
// decode GEMMs run into (profiles/r06_bd32_kslice.md).  (scripts/probes: measurement only)
//   hipcc --offload-arch=gfx950 -O3 -o cu_load_cap_probe cu_load_cap_probe.hip && ./cu_load_cap_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <initializer_list>

template <int U>
__global__ void k(const uint4* __restrict__ buf, size_t slice_vec, int trips, unsigned long long* cyc, uint32_t* sink, int shared) {
    extern __shared__ uint8_t pad[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const uint4* base = buf + (shared ? 0 : (size_t)blockIdx.x * slice_vec);      // shared: every CU reads the SAME slice (hot lines, as the activations of a GEMM)
    uint4 acc = {0, 0, 0, 0};
    const uint32_t mask = (uint32_t)slice_vec - 1;                 // slices are powers of two
    uint32_t pos = (uint32_t)wave * 64 * U + lane;                 // wavefronts interleave: trip = nw * U * 1 KB of the slice
    const uint32_t step = (uint32_t)nw * 64 * U;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < trips; ++it) {
        uint4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = base[(pos + 64 * u) & mask];
#pragma unroll
        for (int u = 0; u < U; ++u) { acc.x ^= v[u].x; acc.y ^= v[u].y; acc.z ^= v[u].z; acc.w ^= v[u].w; }
        pos += step;
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[0] = 1;      // keeps the loads
    if (lane == 0) cyc[blockIdx.x * 16 + wave] = t1 - t0;
    if (pad[0] == 77 && trips < 0) sink[1] = 1;
}

template <int U>
static void run(const uint4* buf, size_t total_bytes, size_t slice_bytes, int waves, const char* what, unsigned long long* cyc, uint32_t* sink, int shared = 0) {
    const int ncu = 256;
    const size_t slice_vec = slice_bytes / 16;
    const size_t per_trip = (size_t)waves * U * 1024;
    // L2-resident: many passes over the slice; stream: one pass
    const size_t want = slice_bytes <= (256u << 10) ? (size_t)32 << 20 : slice_bytes;      // bytes per CU
    const int trips = (int)(want / per_trip);
    hipFuncSetAttribute((const void*)k<U>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    hipLaunchKernelGGL(k<U>, dim3(ncu), dim3(64 * waves), 100 * 1024, 0, buf, slice_vec, trips / 8 + 1, cyc, sink, shared);      // warm (L2 / TLB)
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<U>, dim3(ncu), dim3(64 * waves), 100 * 1024, 0, buf, slice_vec, trips, cyc, sink, shared);
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[256 * 16];
    hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    double mean = 0; unsigned long long mx = 0;
    for (int b = 0; b < ncu; ++b) for (int w = 0; w < waves; ++w) { mean += h[b * 16 + w]; if (h[b * 16 + w] > mx) mx = h[b * 16 + w]; }
    mean /= (double)ncu * waves;
    const double bytes_cu = (double)trips * per_trip;
    printf("%-10s waves/CU %2d loads in flight/wave %2d: %6.1f B/clk/CU (mean wavefront), %6.1f (slowest); chip %.2f TB/s by events (%.0f us)\n", what, waves, U,
           bytes_cu / mean, bytes_cu / (double)mx, bytes_cu * ncu / (ms * 1e-3) / 1e12, ms * 1e3);
}

int main() {
    const size_t total = (size_t)4 << 30;
    uint4* buf; hipMalloc(&buf, total); hipMemset(buf, 1, total);
    unsigned long long* cyc; hipMalloc(&cyc, 256 * 16 * 8); uint32_t* sink; hipMalloc(&sink, 8);
    for (int waves : {1, 2, 3, 4, 8, 16}) {
        run<4>(buf, total, 64 << 10, waves, "L2 slice", cyc, sink);
        run<8>(buf, total, 64 << 10, waves, "L2 slice", cyc, sink);
        run<16>(buf, total, 64 << 10, waves, "L2 slice", cyc, sink);
    }
    for (int waves : {1, 2, 3, 4, 8}) {
        run<8>(buf, total, 64 << 10, waves, "L2 shared", cyc, sink, 1);
        run<8>(buf, total, 2 << 10, waves, "2KB shared", cyc, sink, 1);
    }
    for (int waves : {1, 2, 4, 8, 16}) {
        run<8>(buf, total, (size_t)16 << 20, waves, "HBM stream", cyc, sink);
        run<16>(buf, total, (size_t)16 << 20, waves, "HBM stream", cyc, sink);
    }
    return 0;
}
