// gl3_decode_kernels.h — hand-written gfx950 kernels for the single-token decode step.
//
// Replaces the TornadoVM-JITted Java kernels of
//   J/tornadovm/kernels/TransformerComputeKernelsLayered.java (fusedQKVMatmulQ8 :3038-3223,
//   matrixVectorGenericWithResidualQ8_0Byte :2888-2906, fullyFusedRmsNormFFNGateUpQ8 :3386-3549,
//   processHeadsFlashAttention :784-906, ropeRotationWithCacheCopy :495-542) and
//   J/tornadovm/kernels/TransformerComputeKernels.java (convertQ8_0toFP32 :95-126, reductionOneBlock* :149-208)
// but reproduces the ARITHMETIC of the pure-Java CPU path (the parity oracle) BIT FOR BIT:
//   * Q8_0 matvec = dotQ8Activation (J/tensor/standard/Q8_0FloatTensor.java:90-123): activation quantised to
//     int8 per 32-block (amax/127, f16-rounded scale, round-half-away), int8 x int8 -> int32 (v_dot4_i32_i8),
//     p_b = isum * (wScale * aScale) in f32, and result += p_b STRICTLY IN BLOCK ORDER;
//   * every other reduction (RMSNorm sum of squares, q.k dot, softmax denominator, weighted V sum) is also
//     evaluated in the reference's left-to-right order with one f32 rounding per step, no FMA contraction
//     (-ffp-contract=off); exp / sqrt are evaluated in double and cast, as java.lang.Math does.
// Why bit-exact and not "close": the reference re-quantises activations to int8 before every matmul, so a
// 1-ulp difference in any f32 sum flips an int8 somewhere and grows to ~1e-2 in the logits within a layer
// (measured).  Only the same summation order meets the 1e-3 north-star tolerance.
//
// In-order sums at HBM speed: v_mfma_f32_16x16x4_f32 with B = 1.0 is bit-identical to four sequential f32
// adds per row (D = fl(A[k] + D), k ascending; verified on MI355X by scripts/probes/mfma_chain_probe.hip), so
// a wavefront advances 16 rows x 4 blocks of the strictly ordered block sum per instruction on the otherwise
// idle matrix pipe.
//
// Weight layout in HBM ("Q8T", built once at upload by repack_q8t_kernel): rows are grouped in strips of 16
// and 32-element blocks in groups of 4; tile (strip, g) holds 64 blocks, lane l = (row l&15, block 4g + l>>4):
//   [64 x f16 scale][64 x 16 B quants 0..15][64 x 16 B quants 16..31]      (2176 B = 64 GGUF blocks)
// = exactly the A-operand layout of the 16x16x4 MFMA, read with three fully coalesced loads per wavefront.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "gl3_seqsum.h"

namespace gl3 {

constexpr int TILE_BYTES = 2176;       // 64 Q8_0 blocks
constexpr int MV_AUX = 4;               // wavefronts running the prologue; the first one then runs the ordered sums
// NPW = wavefronts streaming weights: 4, or 8 for matrices with <= 256 strips (wo / down), where only one workgroup
// lands on a CU and four wavefronts cannot keep enough bytes in flight (each alternates issue -> wait -> dot).
__host__ __device__ constexpr int mv_threads(int npw) { return 64 * (npw + MV_AUX); }

enum { PRO_RMS = 0, PRO_QUANT = 1 };
enum { EPI_STORE = 0, EPI_RESID = 1, EPI_SWIGLU = 2 };

typedef float v4f __attribute__((ext_vector_type(4)));

#ifdef GL3_MV_TIMING
__device__ long long gl3_mv_stamp[32];
#define ATT_STAMP(i) do { if (blockIdx.x == 0 && threadIdx.x == 0) gl3_mv_stamp[i] = wall_clock64(); } while (0)
#define MV_STAMP(i, cond) do { if (blockIdx.x == 0 && (cond)) gl3_mv_stamp[i] = clock64(); } while (0)
#define PV_T(...) __VA_ARGS__
#else
#define PV_T(...)
#define MV_STAMP(i, cond)
#define ATT_STAMP(i)
#endif

// ---------------------------------------------------------------------------------------------------
// Tensor-parallel hand-over folded into the kernels that produce / consume a gathered buffer (protocol: gl3_tp.hip, "folded
// gathers").  A producer (attention -> xb, gate/up -> hb, down -> x) stores every result element into its own arena AND into each
// peer's arena (the same arena offset: own address + delta[j]); the wavefront that finishes last publishes "gather k of this buffer
// from rank me is complete" into every peer's flag word.  A consumer waits until every peer's flag for the buffer has reached k —
// in a one-wavefront wait kernel in front of it, or in its own prologue (TpRec.w inside matvec_q8t_kernel<.., TPF = true>).
// k = (step - 1) * mul + add: step = decode steps taken by this plan (device word, bumped by the embedding kernel), mul = gathers of
// the buffer per step, add = index of this gather inside the step: nothing of it is baked into a captured graph.
constexpr int TPF_MAX_PEERS = 15;
struct TpWait {
    const uint32_t* flags;       // own arena: flags[rank] of the awaited buffer; NULL = nothing to wait for
    const uint32_t* step;
    uint32_t* err;               // host-pinned word: set to 1 on a timeout
    int mul, add, tp, me;
    unsigned spin_limit;
};
struct TpPush {
    uint32_t* ticket;            // own arena: wavefronts of the producing launch that have finished
    const uint32_t* step;
    int mul, add, npeers;        // npeers = 0: nothing to push
    long delta[TPF_MAX_PEERS];   // peer arena base - own arena base (bytes), peers in rotated order (me + 1, me + 2, ...)
    uint32_t* flag[TPF_MAX_PEERS];   // peer j's flags[me] of the buffer
};
struct TpRec { TpWait w; TpPush p; };

// every lane of the calling wavefront returns once all peers have published gather k (lane p polls the flag of rank p)
__device__ __forceinline__ void tp_wait(const TpWait& w) {
    const uint32_t need = (__hip_atomic_load(w.step, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - 1u) * (uint32_t)w.mul + (uint32_t)w.add;
    const int lane = threadIdx.x & 63;
    const bool mine = lane < w.tp && lane != w.me;
    unsigned spins = 0;
    for (;;) {
        // signed distance: a peer may already be a step ahead
        const bool ok = !mine || (int32_t)(__hip_atomic_load(w.flags + (mine ? lane : 0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) - need) >= 0;
        if (__builtin_amdgcn_ballot_w64(!ok) == 0) break;
        __builtin_amdgcn_s_sleep(8);
        if (++spins > w.spin_limit) { if (lane == 0) __hip_atomic_store(w.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); break; }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");      // system scope: nothing read before the peers' data landed survives
}
// one result element into every peer's copy of the buffer
__device__ __forceinline__ void tp_push_store(const TpPush& p, float* own, float v) {
    for (int j = 0; j < p.npeers; ++j)
        *reinterpret_cast<float*>(reinterpret_cast<uint8_t*>(own) + p.delta[j]) = v;
}
// called by every storing wavefront of the launch after its last tp_push_store (all lanes); total = such wavefronts in the launch
__device__ __forceinline__ void tp_publish(const TpPush& p, unsigned total) {
    // This wavefront's stores have been acknowledged by the peers' memory: the arenas are uncached on both sides, so a completed
    // store is a visible one and no cache needs writing back here.  (A system-scope fence per wavefront — hundreds per launch, each
    // an L2 write-back — made the folded decode step 20 % slower than the gather kernels between two ranks on one GPU.)  The one
    // wavefront that publishes fences once.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if ((threadIdx.x & 63) != 0) return;
    const unsigned ticket = __hip_atomic_fetch_add(p.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (ticket != total - 1) return;
    __hip_atomic_store(p.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const uint32_t k = (__hip_atomic_load(p.step, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - 1u) * (uint32_t)p.mul + (uint32_t)p.add;
    __threadfence_system();
    for (int j = 0; j < p.npeers; ++j) __hip_atomic_store(p.flag[j], k, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// The wait as its own launch (one wavefront): the default consumer side — it cannot starve a peer rank's producer of compute units
// when ranks share one GPU (CI), which a consumer that fills the chip while it polls can.
static __global__ __launch_bounds__(64) void tp_wait_kernel(const TpRec* r) { tp_wait(r->w); }

struct MatvecArgs {
    const uint8_t* w;        // Q8T tiles
    const uint8_t* w2;       // second matrix (EPI_SWIGLU: w = gate W1, w2 = up W3)
    int rows;                // valid output rows
    int k;                   // input length (multiple of 32)
    int ng;                  // tile groups per strip = padded blocks / 4
    int nstrips;             // 16-row strips
    const float* x;          // input vector f32[k]
    const float* norm_w;     // PRO_RMS: RMSNorm weight f32[k]
    float eps;
    float* out;              // EPI_STORE: out[row]; EPI_SWIGLU: hb[row]; EPI_RESID: out[row] = resid_in[row] + acc
    const float* resid_in;   // may be NULL (tensor-parallel ranks > 0)
    float out_scale;         // EPI_STORE / EPI_RESID: the row result is multiplied by this first (1 except Granite: residualScale after wo /
                             // down, logitScale on the logits — InferenceCore.forwardGranite :893-894, :911-912, :921; x * 1.0f is exact)
    const struct MoeSlots* moe;   // SEL instantiations only (Qwen2-MoE): what each blockIdx.y slot of the launch works on
    const TpRec* tp;         // tensor parallel, folded gathers: what this launch pushes to the peers / waits for (NULL: nothing)
};

// Qwen2-MoE launches of matvec_q8t_kernel<.., SEL = true> (InferenceCore.matmulExpert :430-432): blockIdx.y = slot.  Slots j < n_sel
// are the top-k selection: the launch works on expert sel[j] of a stacked tensor — w / w2 += sel[j] * sel_stride bytes,
// x += j * x_slot_stride floats (the slot's own hbE for the down projection), out += j * out_slot_stride floats.  Slots >= n_sel
// (gate/up launch only) are chunk c = slot - n_sel of the SHARED expert's dense matrices sh_w / sh_w2: `rows` rows per chunk out of
// sh_rows, result to sh_out + c * rows — the shared expert rides in the routed experts' launch.  One record per (layer, launch) in
// device memory, written once at plan creation.
struct MoeSlots {
    const int* sel;
    size_t sel_stride;
    int x_slot_stride, out_slot_stride;
    int n_sel, sh_rows;
    const uint8_t* sh_w;
    const uint8_t* sh_w2;
    float* sh_out;
};

// Workgroup barrier for LDS hand-overs: s_waitcnt lgkmcnt(0) + s_barrier.  __syncthreads() also waits for vmcnt(0) — every global load
// a wavefront has in flight — which turns a prefetch issued in front of it into a full memory round trip per barrier.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

struct LdsBarrier { __device__ __forceinline__ void operator()() const { lds_barrier(); } };      // Sync functor of exact_seqsum_lds

__device__ __forceinline__ float h2f(uint16_t h) { return (float)__builtin_bit_cast(_Float16, h); }

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v = fmaxf(v, __shfl_xor(v, m, 64));
    return v;
}

// Strict left-to-right f32 sum of n floats held in LDS, executed redundantly by every lane of the calling
// wavefront (uniform addresses -> LDS broadcast).  SQ = true sums x*x (InferenceCore.rmsnorm :41).
template <bool SQ>
__device__ __forceinline__ float seq_add4(float s, const float4& a) {
    if (SQ) { s = s + a.x * a.x; s = s + a.y * a.y; s = s + a.z * a.z; s = s + a.w * a.w; }
    else { s = s + a.x; s = s + a.y; s = s + a.z; s = s + a.w; }
    return s;
}
template <bool SQ>
__device__ __forceinline__ float seq_sum_lds(const float* v, int n, float start = 0.f) {     // start: running sum before v[0] (chunked rows)
    float s = start;
    int i = 0;
    if (n >= 16) {                                     // software pipeline: the next 8 elements are in flight
        float4 a0 = *reinterpret_cast<const float4*>(v), a1 = *reinterpret_cast<const float4*>(v + 4);
        for (; i + 16 <= n; i += 8) {
            const float4 b0 = *reinterpret_cast<const float4*>(v + i + 8), b1 = *reinterpret_cast<const float4*>(v + i + 12);
            s = seq_add4<SQ>(s, a0); s = seq_add4<SQ>(s, a1);
            a0 = b0; a1 = b1;
        }
        s = seq_add4<SQ>(s, a0); s = seq_add4<SQ>(s, a1);
        i += 8;
    }
    for (; i + 4 <= n; i += 4) s = seq_add4<SQ>(s, *reinterpret_cast<const float4*>(v + i));
    for (; i < n; ++i) s = SQ ? s + v[i] * v[i] : s + v[i];
    return s;
}

// a * b as one v_mul_f32 the SLP vectoriser cannot pair: v_pk_mul_f32 wants its operands in aligned register pairs, and the
// v_mov shuffles it inserts for that read freshly issued LDS results, i.e. wait for them — which is what a read ring is there to avoid.
__device__ __forceinline__ float mul_f32_scalar(float a, float b) {
    float r;
    asm("v_mul_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// seq_sum_lds<false> with the LDS reads three 16-element groups ahead of the adds, pinned there with sched_barrier (the scheduler
// otherwise sinks every read next to its use and the chain waits an LDS round trip per group: 11 cycles per element instead of ~6,
// scripts/probes/seqsum_time.hip).  Same adds in the same order.  v may differ per lane (lane = row) or be uniform.
__device__ __forceinline__ float seq_sum_lds_ring(const float* v, int n, float start = 0.f) {
    float s = start;
    const int G = n >> 4;
    int i = 0;
    if (G >= 3) {
        float4 a0, a1, a2, a3, b0, b1, b2, b3, c0, c1, c2, c3;
#define SSR_LD(G_, V0_, V1_, V2_, V3_) do { const float* q_ = v + 16 * min((G_), G - 1); V0_ = *reinterpret_cast<const float4*>(q_); \
        V1_ = *reinterpret_cast<const float4*>(q_ + 4); V2_ = *reinterpret_cast<const float4*>(q_ + 8); V3_ = *reinterpret_cast<const float4*>(q_ + 12); } while (0)
#define SSR_ADD(V0_, V1_, V2_, V3_) do { s = seq_add4<false>(s, V0_); s = seq_add4<false>(s, V1_); s = seq_add4<false>(s, V2_); s = seq_add4<false>(s, V3_); } while (0)
        SSR_LD(0, a0, a1, a2, a3); SSR_LD(1, b0, b1, b2, b3); SSR_LD(2, c0, c1, c2, c3);
        int g = 0;
        for (; g + 3 <= G; g += 3) {
            SSR_ADD(a0, a1, a2, a3); SSR_LD(g + 3, a0, a1, a2, a3); __builtin_amdgcn_sched_barrier(0);
            SSR_ADD(b0, b1, b2, b3); SSR_LD(g + 4, b0, b1, b2, b3); __builtin_amdgcn_sched_barrier(0);
            SSR_ADD(c0, c1, c2, c3); SSR_LD(g + 5, c0, c1, c2, c3); __builtin_amdgcn_sched_barrier(0);
        }
        if (g < G) { SSR_ADD(a0, a1, a2, a3); ++g; }
        if (g < G) { SSR_ADD(b0, b1, b2, b3); ++g; }
#undef SSR_ADD
#undef SSR_LD
        i = 16 * G;
    }
    for (; i + 4 <= n; i += 4) s = seq_add4<false>(s, *reinterpret_cast<const float4*>(v + i));
    for (; i < n; ++i) s = s + v[i];
    return s;
}

// ---------------------------------------------------------------------------------------------------
// Activation quantisation into LDS (Q8_0FloatTensor.java:96-118): 8 consecutive threads own one 32-block.
// xq[32*b ..] = int8 quants of block b, xs[b] = f16-rounded activation scale.  v = value already normalised.
// the 4 packed quants of a quad and its block's scale (valid in all 8 lanes of the block)
__device__ __forceinline__ uint32_t quantize_quad_pack(float4 v, float& qs_out) {
    float amax = fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w)));
    // maximum over the block's 8 lanes with DPP moves (quad_perm [1,0,3,2], quad_perm [2,3,0,1], row_half_mirror) instead of
    // three dependent ds_bpermute round trips through LDS (__shfl_xor): ~300 cycles less per quad on the prologue's critical path
    amax = fmaxf(amax, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, amax), 0xB1, 0xf, 0xf, false)));
    amax = fmaxf(amax, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, amax), 0x4E, 0xf, 0xf, false)));
    amax = fmaxf(amax, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, amax), 0x141, 0xf, 0xf, false)));
    const float qs = amax / 127.0f;
    const float ainv = qs != 0.f ? 1.0f / qs : 0.f;
    const float s0 = v.x * ainv, s1 = v.y * ainv, s2 = v.z * ainv, s3 = v.w * ainv;
    const int q0 = (int)(s0 + copysignf(0.5f, s0)), q1 = (int)(s1 + copysignf(0.5f, s1));
    const int q2 = (int)(s2 + copysignf(0.5f, s2)), q3 = (int)(s3 + copysignf(0.5f, s3));
    qs_out = (float)(_Float16)qs;                                        // float16ToFloat(floatToFloat16(qs))
    return (uint32_t)(q0 & 0xFF) | ((uint32_t)(q1 & 0xFF) << 8) | ((uint32_t)(q2 & 0xFF) << 16) | ((uint32_t)(q3 & 0xFF) << 24);
}

__device__ __forceinline__ void quantize_quad(float4 v, int qd, uint8_t* xq, float* xs) {
    float qs;
    const uint32_t packed = quantize_quad_pack(v, qs);
    *reinterpret_cast<uint32_t*>(xq + 4 * qd) = packed;                  // byte 32*b + 4*(qd&7)
    if ((qd & 7) == 0) xs[qd >> 3] = qs;
}

// Small-batch operand layout (gl3_bd_gemm.h): byte offset of quad qd (elements 4qd .. 4qd + 3 of a token's row) in XQ2, and
// float offset of block blk's scale in XS2; tslots token slots.
__device__ __forceinline__ size_t bdq_offset(int qd, int tok, int tslots) {
    const int blk = qd >> 3, qi = qd & 7, c = qi >> 1;
    const int g = ((c & 1) << 1) | (c >> 1);                 // k-chunk 0, 1, 2, 3 -> k-group 0, 2, 1, 3
    return ((size_t)(blk >> 1) * tslots + tok) * 64 + 16 * g + 8 * (blk & 1) + 4 * (qi & 1);
}
__device__ __forceinline__ size_t bds_offset(int blk, int tok, int tslots) { return ((size_t)(blk >> 2) * tslots + tok) * 4 + (blk & 3); }

__device__ __forceinline__ int dot32(const int4& a0, const int4& a1, const int4& b0, const int4& b1) {
    int s = 0;
    s = __builtin_amdgcn_sdot4(a0.x, b0.x, s, false); s = __builtin_amdgcn_sdot4(a0.y, b0.y, s, false);
    s = __builtin_amdgcn_sdot4(a0.z, b0.z, s, false); s = __builtin_amdgcn_sdot4(a0.w, b0.w, s, false);
    s = __builtin_amdgcn_sdot4(a1.x, b1.x, s, false); s = __builtin_amdgcn_sdot4(a1.y, b1.y, s, false);
    s = __builtin_amdgcn_sdot4(a1.z, b1.z, s, false); s = __builtin_amdgcn_sdot4(a1.w, b1.w, s, false);
    return s;
}

template <bool NT>
__device__ __forceinline__ int4 ld16(const uint8_t* p) {
    typedef int v4i __attribute__((ext_vector_type(4)));
    v4i v;
    if (NT) v = __builtin_nontemporal_load(reinterpret_cast<const v4i*>(p));
    else v = *reinterpret_cast<const v4i*>(p);
    int4 r; r.x = v.x; r.y = v.y; r.z = v.z; r.w = v.w;
    return r;
}
template <bool NT>
__device__ __forceinline__ uint16_t ld2(const uint8_t* p) {
    if (NT) return __builtin_nontemporal_load(reinterpret_cast<const uint16_t*>(p));
    return *reinterpret_cast<const uint16_t*>(p);
}

// ---------------------------------------------------------------------------------------------------
// Dequant-fused Q8_0 matvec, bit-exact to FloatTensor.matmul + dotQ8Activation.
// Workgroup = 8 wavefronts with three roles:
//   producers (waves 0-3): stream the weight tiles of the workgroup's 16-row strips (lane = one 32-block of one
//       row: 3 coalesced non-temporal loads, 8 v_dot4, p = isum*(wScale*aScale)) into a double-buffered LDS
//       array.  Their first loads are issued BEFORE the activation is ready, so HBM latency hides the prologue;
//   aux (waves 4-7): the prologue — RMSNorm with the exact in-order sum of squares (gl3_seqsum.h) and the
//       Q8_0 activation quantisation into LDS; waves 5-7 then retire;
//   chain (wave 4): adds the p's of each row in block order with v_mfma_f32_16x16x4_f32 (B = 1.0) and applies
//       the epilogue while the producers already stream the next strip.
// Register pressure is the maximum of the roles, not their sum (wave-uniform branches).
//   LDS: xq[ng*128] | xs[ng*4] f32 | xf[k + 32] f32 (PRO_RMS) | pbuf[2][NM][ng*64] f32 | red[4] | sync[4]
// TPF (tensor parallel, folded gathers): the aux wavefronts wait for the peers' slices of x before they read it (after the barrier
// that releases the producers: the weight stream starts while the wait polls), and the chain wavefront stores every result into
// the peers' arenas as well and publishes the gather (TpRec above).
template <int PRO, int EPI, bool NT, int NPW = 4, bool SEL = false, bool TPF = false>
__global__ __launch_bounds__(mv_threads(NPW), NPW == 4 ? 4 : 2) void matvec_q8t_kernel(const MatvecArgs a_in) {
    MatvecArgs a = a_in;
    if (SEL) {                                          // wave-uniform: one scalar load of the expert id
        const int slot = blockIdx.y;
        const MoeSlots m = *a.moe;
        if (slot < m.n_sel) {
            const size_t woff = (size_t)m.sel[slot] * m.sel_stride;
            a.w += woff;
            if (a.w2) a.w2 += woff;
            a.x += (size_t)slot * m.x_slot_stride;
            a.out += (size_t)slot * m.out_slot_stride;
        } else {
            const int c = slot - m.n_sel;
            const size_t woff = (size_t)c * m.sel_stride;
            a.w = m.sh_w + woff;
            a.w2 = m.sh_w2 ? m.sh_w2 + woff : nullptr;
            a.out = m.sh_out + (size_t)c * a.rows;
            const int left = m.sh_rows - c * a.rows;
            if (left < a.rows) { a.rows = left; a.nstrips = (left + 15) / 16; }
        }
    }
    constexpr int MV_PRODUCERS = NPW;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    constexpr int NM = (EPI == EPI_SWIGLU) ? 2 : 1;
    constexpr int CH = (NM == 1) ? 8 : 4;             // tiles in flight per producer wave
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int nb4 = a.ng * 4;
    uint8_t* xq = smem;
    float* xs = reinterpret_cast<float*>(smem + (size_t)nb4 * 32);
    float* xf = xs + nb4;
    float* pbuf = xf + (PRO == PRO_RMS ? a.k + 32 : 0);     // 32 zero floats pad xf for exact_sumsq_lds
    float* red = pbuf + (size_t)2 * NM * a.ng * 64;         // [0] plain-chain result
    int* sync_w = reinterpret_cast<int*>(red + 4);          // [0] aux sub-barrier counter, [1] activation-ready flag
    const size_t strip_bytes = (size_t)a.ng * TILE_BYTES;
    MV_STAMP(0, t == 0);
    if (t < 2) sync_w[t] = 0;

    if (wave < MV_PRODUCERS) {
        // ------------------------------------------------------------------ producers
        const int nt = (a.ng - wave + MV_PRODUCERS - 1) / MV_PRODUCERS;      // tiles g = wave + 4*i of a strip
        uint16_t sc[NM][CH];
        int4 lo[NM][CH], hi[NM][CH];
        auto issue = [&](int strip, int i0) {
#pragma unroll
            for (int u = 0; u < CH; ++u) {
                if (i0 + u < nt) {
                    const int g = wave + (i0 + u) * MV_PRODUCERS;
#pragma unroll
                    for (int m = 0; m < NM; ++m) {
                        const uint8_t* p = (m == 0 ? a.w : a.w2) + (size_t)strip * strip_bytes + (size_t)g * TILE_BYTES;
                        sc[m][u] = ld2<NT>(p + 2 * lane);
                        lo[m][u] = ld16<NT>(p + 128 + 16 * lane);
                        hi[m][u] = ld16<NT>(p + 1152 + 16 * lane);
                    }
                }
            }
        };
        auto compute = [&](int i0, float* pb) {
#pragma unroll
            for (int u = 0; u < CH; ++u) {
                if (i0 + u < nt) {
                    const int g = wave + (i0 + u) * MV_PRODUCERS;
                    const int xb = 4 * g + (lane >> 4);
                    const int4 xlo = *reinterpret_cast<const int4*>(xq + 32 * xb);
                    const int4 xhi = *reinterpret_cast<const int4*>(xq + 32 * xb + 16);
                    const float xsc = xs[xb];
#pragma unroll
                    for (int m = 0; m < NM; ++m) {
                        const int isum = dot32(lo[m][u], hi[m][u], xlo, xhi);
                        pb[(size_t)m * a.ng * 64 + g * 64 + lane] = (float)isum * (h2f(sc[m][u]) * xsc);
                    }
                }
            }
        };
        int strip = blockIdx.x;
        __syncthreads();                                         // the only prologue barrier the producers join
        if (strip < a.nstrips) issue(strip, 0);                  // HBM latency hides the activation prologue
        while (__hip_atomic_load(&sync_w[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == 0) __builtin_amdgcn_s_sleep(2);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        MV_STAMP(3, t == 0);
        int it = 0;
        for (; strip < a.nstrips; strip += gridDim.x, ++it) {
            float* pb = pbuf + (size_t)(it & 1) * NM * a.ng * 64;
            for (int i0 = 0; i0 < nt; i0 += CH) {
                if (it != 0 || i0 != 0) issue(strip, i0);
                compute(i0, pb);
            }
            MV_STAMP(4 + 2 * it, t == 0 && it < 4);
            __syncthreads();
            MV_STAMP(5 + 2 * it, t == 0 && it < 4);
        }
        return;
    }

    // ---------------------------------------------------------------------- aux waves: prologue
    // the prologue is the critical path of the kernel; the producers only issue their loads and poll
    __builtin_amdgcn_s_setprio(3);
    const int ta = t - 64 * MV_PRODUCERS;                // 0..255
    const int nquads = a.k >> 2;
    float scale = 1.0f;
    constexpr int NXV = (PRO == PRO_RMS) ? 5 : 14;       // quads of the activation held in registers per aux thread
    float4 nwv[5], xv[NXV];                              // this thread's RMSNorm weights and activations
    // all activation loads first: they come from L2, the norm weights behind them from HBM (vmcnt retires in order, and the
    // activation is needed 5 us before the weights)
    // UNCONDITIONAL loads with clamped indices (quads past the end re-read the last one: an L1 hit that is never used).  With a
    // lane-predicated `if (qd < nquads) xv[i] = load`, and even with a wave-uniform condition, the compiler waited
    // (s_waitcnt vmcnt(0)) after every single load: 10 to 14 serial L2 round trips in front of every matvec (seen in the ISA).
    if (!TPF) {
#pragma unroll
        for (int i = 0; i < NXV; ++i) xv[i] = *reinterpret_cast<const float4*>(a.x + 4 * min(ta + 256 * i, nquads - 1));
    }
    if (PRO == PRO_RMS) {
#pragma unroll
        for (int i = 0; i < 5; ++i) nwv[i] = *reinterpret_cast<const float4*>(a.norm_w + 4 * min(ta + 256 * i, nquads - 1));
    }
    __syncthreads();                                     // activation loads are queued ahead of the weight stream
    if (TPF) {
        if (a.tp->w.flags) tp_wait(a.tp->w);
#pragma unroll
        for (int i = 0; i < NXV; ++i) xv[i] = *reinterpret_cast<const float4*>(a.x + 4 * min(ta + 256 * i, nquads - 1));
    }
    SubBarrier aux_sync{&sync_w[0], MV_AUX, 0};
    if (PRO == PRO_RMS) {
        if (nquads <= NXV * 256) {
#pragma unroll
            for (int i = 0; i < NXV; ++i) { const int qd = ta + 256 * i; if (qd < nquads) *reinterpret_cast<float4*>(xf + 4 * qd) = xv[i]; }
        } else {
            for (int qd = ta; qd < nquads; qd += 256)
                *reinterpret_cast<float4*>(xf + 4 * qd) = *reinterpret_cast<const float4*>(a.x + 4 * qd);
        }
        if (ta < 32) xf[a.k + ta] = 0.f;
        aux_sync();
        MV_STAMP(1, ta == 0);
        // InferenceCore.rmsnorm :41 — the strict left-to-right sum of squares, evaluated exactly in parallel
        // (gl3_seqsum.h); pbuf is free during the prologue and serves as its scratch.
        float ss;
        if ((size_t)2 * NM * a.ng * 64 * 4 >= ss_scratch_bytes(a.k) && a.k >= 1024 && a.k <= 5120) {
            ss = exact_sumsq_lds(xf, a.k, reinterpret_cast<uint8_t*>(pbuf), ta, aux_sync);
        } else {
            if (wave == MV_PRODUCERS) {
                const float s1 = seq_sum_lds<true>(xf, a.k);
                if (lane == 0) red[0] = s1;
            }
            aux_sync();
            ss = red[0];
        }
        MV_STAMP(2, ta == 0);
        ss /= (float)a.k;
        ss += a.eps;
        scale = (float)(1.0 / sqrt((double)ss));
    }
    for (int b = (a.k >> 5) + ta; b < nb4; b += 256) {   // zero-padded blocks
        xs[b] = 0.f;
        const int4 z = {0, 0, 0, 0};
        *reinterpret_cast<int4*>(xq + 32 * b) = z;
        *reinterpret_cast<int4*>(xq + 32 * b + 16) = z;
    }
    if (nquads <= NXV * 256) {
#pragma unroll
        for (int i = 0; i < NXV; ++i) {
            const int qd = ta + 256 * i;
            if (qd < nquads) {
                float4 v = xv[i];
                if (PRO == PRO_RMS) {
                    const float4 w = nwv[i];
                    v.x = w.x * (scale * v.x); v.y = w.y * (scale * v.y); v.z = w.z * (scale * v.z); v.w = w.w * (scale * v.w);
                }
                quantize_quad(v, qd, xq, xs);
            }
        }
    } else {
        for (int qd = ta; qd < nquads; qd += 256) {
            float4 v = *reinterpret_cast<const float4*>(a.x + 4 * qd);
            if (PRO == PRO_RMS) {
                const float4 w = *reinterpret_cast<const float4*>(a.norm_w + 4 * qd);
                v.x = w.x * (scale * v.x); v.y = w.y * (scale * v.y); v.z = w.z * (scale * v.z); v.w = w.w * (scale * v.w);
            }
            quantize_quad(v, qd, xq, xs);
        }
    }
    aux_sync();                                          // xq / xs complete
    if (ta == 0) __hip_atomic_store(&sync_w[1], 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (wave != MV_PRODUCERS) return;                    // waves 5-7 retire; wave 4 becomes the chain wavefront

    // ---------------------------------------------------------------------- chain wavefront
    int it = 0;
    for (int strip = blockIdx.x; strip < a.nstrips; strip += gridDim.x, ++it) {
        __syncthreads();
        MV_STAMP(16 + 2 * it, lane == 0 && it < 4);
        const float* pb = pbuf + (size_t)(it & 1) * NM * a.ng * 64;
        v4f acc[NM];
#pragma unroll
        for (int m = 0; m < NM; ++m) acc[m] = (v4f){0.f, 0.f, 0.f, 0.f};
        int g = 0;
        for (; g + 8 <= a.ng; g += 8) {               // result += p_b, b ascending: 4 blocks x 16 rows per MFMA
            float av[NM][8];
#pragma unroll
            for (int m = 0; m < NM; ++m)
#pragma unroll
                for (int u = 0; u < 8; ++u) av[m][u] = pb[(size_t)m * a.ng * 64 + (g + u) * 64 + lane];
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
                for (int m = 0; m < NM; ++m) acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[m][u], 1.0f, acc[m], 0, 0, 0);
        }
        for (; g < a.ng; ++g)
#pragma unroll
            for (int m = 0; m < NM; ++m)
                acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(pb[(size_t)m * a.ng * 64 + g * 64 + lane], 1.0f, acc[m], 0, 0, 0);
        MV_STAMP(17 + 2 * it, lane == 0 && it < 4);
        // D[i][j]: row i = 4*(lane>>4) + reg, all 16 columns identical.  Lane l takes row (l & 3) of its own group
        // when (l & 15) < 4, so the 16 rows of the strip are finished by 16 lanes in parallel (one double exp each).
        if ((lane & 15) < 4) {
            const int r = lane & 3;
            const int row = strip * 16 + 4 * (lane >> 4) + r;
            if (row < a.rows) {
                const float v0 = r == 0 ? acc[0][0] : r == 1 ? acc[0][1] : r == 2 ? acc[0][2] : acc[0][3];
                float res;
                if (EPI == EPI_STORE) res = v0 * a.out_scale;
                if (EPI == EPI_RESID) res = a.resid_in ? a.resid_in[row] + v0 * a.out_scale : v0 * a.out_scale;
                if (EPI == EPI_SWIGLU) {                  // InferenceCore.java:155-158, exp in double
                    const float v1 = r == 0 ? acc[NM - 1][0] : r == 1 ? acc[NM - 1][1] : r == 2 ? acc[NM - 1][2] : acc[NM - 1][3];
                    const float gte = v0 / (float)(1.0 + exp(-(double)v0));
                    res = gte * v1;
                }
                a.out[row] = res;
                if (TPF) tp_push_store(a.tp->p, a.out + row, res);
            }
        }
    }
    if (TPF && a.tp->p.npeers) tp_publish(a.tp->p, gridDim.x);
}

// ---------------------------------------------------------------------------------------------------
// Embedding row gather + dequant: x[i] = q * d  (token_embedding_table.copyTo, InferenceCore.java:61;
// replaces the host row copy of forwardTornadoVM :956-980 + convertQ8_0toFP32).
// tp (folded gathers): the kernel opens decode step number *step + 1 of the plan — it first waits until the peers' last pushes of
// the previous step into x have landed (a later arrival would overwrite this step's embedding), then bumps the step word.
static __global__ __launch_bounds__(256) void embed_q8t_kernel(const uint8_t* __restrict__ emb, int ng, int dim,
                                                         const int* __restrict__ dyn, float* __restrict__ x, float emb_scale,
                                                         const TpRec* tp = nullptr, uint32_t* step = nullptr) {
    if (step) {
        const uint32_t s = *step;
        __syncthreads();
        if (threadIdx.x == 0) *step = s + 1;
        if (tp) {                          // as tp_wait with "need" spelled out: the step word is being rewritten
            TpWait w = tp->w;
            const uint32_t need = s * (uint32_t)w.mul;
            const int lane = threadIdx.x & 63;
            const bool mine = lane < w.tp && lane != w.me;
            unsigned spins = 0;
            for (;;) {
                const bool ok = !mine || (int32_t)(__hip_atomic_load(w.flags + (mine ? lane : 0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) - need) >= 0;
                if (__builtin_amdgcn_ballot_w64(!ok) == 0) break;
                __builtin_amdgcn_s_sleep(8);
                if (++spins > w.spin_limit) { if (lane == 0) __hip_atomic_store(w.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); break; }
            }
        }
    }
    const int token = dyn[0];
    const uint8_t* strip = emb + (size_t)(token >> 4) * ng * TILE_BYTES;
    const int i16 = token & 15;
    for (int i = threadIdx.x; i < dim; i += 256) {
        const int b = i >> 5, j = i & 31;
        const uint8_t* p = strip + (size_t)(b >> 2) * TILE_BYTES;
        const int l = i16 + 16 * (b & 3);
        const float d = h2f(*reinterpret_cast<const uint16_t*>(p + 2 * l));
        const int8_t q = (int8_t)p[(j < 16 ? 128 : 1152) + 16 * l + (j & 15)];
        x[i] = ((float)q * d) * emb_scale;        // Granite: embeddingScale (forwardGranite :829); 1 otherwise (exact)
    }
}

// ---------------------------------------------------------------------------------------------------
// Decode attention, part 1: RoPE + KV-cache write + scores.   Grid = (n_tsplit, n_kv_heads), block = 64 x kvMul.
//   RoPE (adjacent pairs) InferenceCore.java:75-87; Qwen3 per-head RMSNorm + NeoX RoPE :594-619;
//   KV write :92-93; score = scalarDot(q_h, K[t]) / sqrt(head_size) :108-116 (strict j order, mul then add).
// One workgroup owns a kv head and a tile of 64 timesteps: thread = (query head of the group, timestep).
// The K tile is staged through LDS with coalesced loads (row pitch hs+1: conflict-free per-lane rows).
struct AttnArgs {
    const float* qkv;        // raw [qDim | kvDim | kvDim] from the qkv projection
    float* kcache;           // [ctx][kvDim] of this layer
    float* vcache;
    const float* rope_cr;    // [ctx][hs/2]
    const float* rope_ci;
    const float* qnorm;      // qwen3: f32[hs] (else NULL)
    const float* knorm;
    const float* bq;         // qwen2: q / k / v bias of this rank's heads (else NULL), added before RoPE (InferenceCore.java:456-459)
    const float* bk;
    const float* bv;
    const int* dyn;          // dyn[1] = position
    float* att;              // [n_heads][ctx] scores
    float* xb;               // [qDim] attention output
    int n_heads, n_kv_heads, hs, q_dim, kv_dim, ctx;
    float eps;
    int arch;
    // static-batched decode (attn_head_kernel with gridDim.y = tokens): token b belongs to sequence seqv[b] at position posv[b]
    const int* seqv;         // NULL: single-token decode (position = dyn[1], one KV cache)
    const int* posv;
    size_t seq_stride;       // floats between the KV caches of consecutive sequences
    int qkv_stride, xb_stride;   // floats between consecutive tokens' rows of qkv / xb
    float att_mul;           // 0: score / sqrt(head_size); Granite: score * attentionScale (forwardGranite :870-872)
    int win;                 // attn_softmax_pv_kernel: floats of the softmax row held in LDS (a multiple of PV_ROWS); longer rows run in windows
    int att_stride;          // floats between the score rows of consecutive heads in att (a multiple of 4, >= ctx)
    float* att_t;            // softmax numerators e_t in attn_pv_kernel's operand order (attn_att_t_floats)
    float* tmax;             // [n_heads][score tiles]: per-tile maxima of the scores (attn_scores_kernel -> attn_exp_kernel); NULL = not wanted
    float* sums;             // [n_heads]: softmax denominators (attn_sum_kernel -> attn_pv_kernel)
    int group;               // attn_head_kernel: query heads per workgroup (0 / 1: one; kvMul: the whole group of a kv head)
    // attn_head_kernel, static-batched decode on one rank: the output leaves the kernel as the wo projection's int8 operand in the
    // small-batch layout (gl3_bd_gemm.h: XQ2 / XS2, xq_slots token slots) instead of f32 xb; NULL = write xb
    uint8_t* xq_out; float* xs_out; int xq_slots;
    const TpRec* tp;         // tensor parallel, folded gathers: xb also goes to the peers' arenas (NULL: not folded)
};

__device__ __forceinline__ void rope_head(float* v, int hs, const float* cr, const float* ci, int arch, int t0, int nthreads) {
    const int half = hs >> 1;
    for (int i = t0; i < half; i += nthreads) {
        const float fcr = cr[i], fci = ci[i];
        const int i0 = arch == 0 ? 2 * i : i, i1 = arch == 0 ? 2 * i + 1 : i + half;
        const float v0 = v[i0], v1 = v[i1];
        v[i0] = v0 * fcr - v1 * fci;
        v[i1] = v0 * fci + v1 * fcr;
    }
}

// rmsnorm(v, v, w, hs) of one head held in LDS (InferenceCore.rmsnorm applied per head, :594-600) by one WAVEFRONT: every lane runs
// the strict sum of squares (uniform addresses: LDS broadcast reads, pipelined float4s), then the lanes normalise their elements in
// parallel.  (One thread per head did both loops alone: 2 x hs dependent LDS / global round trips, ~4 k cycles on the kernel's
// critical path for head_size 128.)  v must be 16-byte aligned; the wavefront's reads for the sum precede its writes (LDS is in order).
__device__ __forceinline__ void head_rmsnorm_wave(float* v, const float* __restrict__ w, int hs, float eps, int lane) {
    float ss = seq_sum_lds<true>(v, hs);
    ss /= (float)hs;
    ss += eps;
    ss = (float)(1.0 / sqrt((double)ss));
    for (int i = lane; i < hs; i += 64) v[i] = w[i] * (ss * v[i]);
}

constexpr int ATT_TT = 64;   // timesteps per score workgroup

static __global__ void attn_scores_kernel(const AttnArgs a) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int hs = a.hs, kvmul = a.n_heads / a.n_kv_heads, half = hs >> 1;
    float* q_s = sm;                         // [kvmul][hs]
    float* kt = q_s + kvmul * hs;            // [ATT_TT][hs + 4]: float4 rows, conflict-free for 16-lane b128 groups
    float* cr_s = kt + ATT_TT * (hs + 4);    // [hs/2]
    float* ci_s = cr_s + half;               // [hs/2]
    const int t = threadIdx.x, nthr = blockDim.x;
    const int sp = blockIdx.x, kvh = blockIdx.y;
    const int pos = a.dyn[1];
    const int t0 = sp * ATT_TT;
    if (t0 > pos) return;
    const int t1 = min(pos + 1, t0 + ATT_TT);
    const bool owns_pos = (t1 == pos + 1);
    const int pitch = hs + 4;
    float* krow = kt + (pos - t0) * pitch;
    // ---- ONE global round trip: raw q of the group's heads, raw k/v of this kv head (owner), the RoPE row of
    // `pos` and the K tile all go to registers first, then to LDS.
    const int q4 = hs >> 2;
    const int nrows_cache = owns_pos ? (t1 - 1 - t0) : (t1 - t0);
    const int nk4 = nrows_cache * q4;                         // float4s of the K tile that come from the cache
    // Staging registers: every element is initialised, the load condition is wave-uniform and the index is clamped — a
    // lane-predicated `if (i < nk4) kreg[u] = load` leaves the array in scratch memory and serialises the loads behind
    // one s_waitcnt each (seen in the ISA: private_segment 272 B, 14 us instead of 5 for a 128-row tile).
    constexpr int KMAX = 16;
    float4 kreg[KMAX];
    const int per = (nk4 + nthr - 1) / nthr;
#pragma unroll
    for (int u = 0; u < KMAX; ++u) {
        kreg[u] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (u < per) {
            const int i = min(t + u * nthr, nk4 - 1);
            kreg[u] = *reinterpret_cast<const float4*>(a.kcache + (size_t)(t0 + i / q4) * a.kv_dim + kvh * hs + 4 * (i % q4));
        }
    }
    for (int i = t; i < kvmul * hs; i += nthr) q_s[i] = a.bq ? a.qkv[(kvh * kvmul) * hs + i] + a.bq[(kvh * kvmul) * hs + i] : a.qkv[(kvh * kvmul) * hs + i];
    for (int i = t; i < half; i += nthr) { cr_s[i] = a.rope_cr[(size_t)pos * half + i]; ci_s[i] = a.rope_ci[(size_t)pos * half + i]; }
    float vraw[4] = {0.f, 0.f, 0.f, 0.f};                 // element t + u * nthr of the v row (hs <= 256, nthr >= 64)
    if (owns_pos) {
        for (int i = t; i < hs; i += nthr) krow[i] = a.bk ? a.qkv[a.q_dim + kvh * hs + i] + a.bk[kvh * hs + i] : a.qkv[a.q_dim + kvh * hs + i];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = t + u * nthr;
            if (i < hs) vraw[u] = a.bv ? a.qkv[a.q_dim + a.kv_dim + kvh * hs + i] + a.bv[kvh * hs + i] : a.qkv[a.q_dim + a.kv_dim + kvh * hs + i];
        }
    }
    if (per > KMAX) {                                         // generic fallback (few threads): straight to LDS
        for (int i = t; i < nk4; i += nthr) {
            const float4 v = *reinterpret_cast<const float4*>(a.kcache + (size_t)(t0 + i / q4) * a.kv_dim + kvh * hs + 4 * (i % q4));
            *reinterpret_cast<float4*>(kt + (i / q4) * pitch + 4 * (i % q4)) = v;
        }
    } else {
#pragma unroll
        for (int u = 0; u < KMAX; ++u) {
            const int i = t + u * nthr;
            if (u < per && i < nk4) *reinterpret_cast<float4*>(kt + (i / q4) * pitch + 4 * (i % q4)) = kreg[u];
        }
    }
    __syncthreads();
    if (a.arch == 1) {
        const int nvec = kvmul + (owns_pos ? 1 : 0);              // the group's query heads (+ this position's key): one wavefront each
        for (int vec = t >> 6; vec < nvec; vec += nthr >> 6)
            head_rmsnorm_wave(vec < kvmul ? q_s + vec * hs : krow, vec < kvmul ? a.qnorm : a.knorm, hs, a.eps, t & 63);
        __syncthreads();
    }
    for (int h = 0; h < kvmul; ++h) rope_head(q_s + h * hs, hs, cr_s, ci_s, a.arch, t, nthr);
    if (owns_pos) rope_head(krow, hs, cr_s, ci_s, a.arch, t, nthr);
    __syncthreads();
    if (owns_pos) {                              // KV write, InferenceCore.java:92-93 (strided: head_size may exceed 64 * kvMul)
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = t + u * nthr;
            if (i < hs) {
                a.kcache[(size_t)pos * a.kv_dim + kvh * hs + i] = krow[i];
                a.vcache[(size_t)pos * a.kv_dim + kvh * hs + i] = vraw[u];
            }
        }
    }
    const int hq = t >> 6, r = t & 63;           // wavefront = query head of the group, lane = timestep
    float my_score = -INFINITY;
    if (hq < kvmul && t0 + r < t1) {
        const float* q = q_s + hq * hs;
        const float* kk = kt + r * pitch;
        float score = 0.f;                           // strict j order, mul then add (FloatTensor.scalarDot)
        float4 qv = *reinterpret_cast<const float4*>(q), kv = *reinterpret_cast<const float4*>(kk);
        for (int j = 4; j < hs; j += 4) {            // software pipeline: next float4 pair in flight
            const float4 qn = *reinterpret_cast<const float4*>(q + j), kn = *reinterpret_cast<const float4*>(kk + j);
            score = score + qv.x * kv.x; score = score + qv.y * kv.y; score = score + qv.z * kv.z; score = score + qv.w * kv.w;
            qv = qn; kv = kn;
        }
        score = score + qv.x * kv.x; score = score + qv.y * kv.y; score = score + qv.z * kv.z; score = score + qv.w * kv.w;
        const float sqrt_hs = (float)sqrt((double)hs);
        my_score = a.att_mul != 0.f ? score * a.att_mul : score / sqrt_hs;
        a.att[(size_t)(kvh * kvmul + hq) * a.att_stride + t0 + r] = my_score;
    }
    // the tile's maximum per head (attn_exp_kernel folds the tiles: max is order-independent)
    if (a.tmax && hq < kvmul) {
        const float m = wave_max(my_score);
        if (r == 0) a.tmax[(size_t)(kvh * kvmul + hq) * gridDim.x + sp] = m;
    }
}

// Long-context form of attn_scores_kernel (r5, positions >= attn_mid; head_size 64 / 128, kvMul <= 4): grid = (SCL_WGS, n_kv_heads), every
// workgroup LOOPS over the score tiles w, w + SCL_WGS, ... of its kv head, with two kinds of wavefronts:
//   chain wavefronts (one per query head of the group, lane = timestep of the tile): q (broadcast) and the lane's K row come from LDS
//       in groups of four 16-byte quads, the next group's reads pinned in flight under the current group's 16 multiply-add pairs;
//   loader wavefronts (4): K rows of the tile THREE trips ahead are requested into named registers (rows of 512 / 256 bytes, fully
//       coalesced), the tile one trip ahead is written into the other half of a double-buffered LDS image.
// The one-tile-per-workgroup kernel above pays, per tile, an exposed HBM round trip for K, the q staging and RoPE (2064 workgroups at
// depth 16384, each redoing them) and chains that read q AND k from LDS right in front of their use (~130 cycles per element when a CU
// holds only four wavefronts; it gets by on 16 resident wavefronts per CU): 27.9 us per 8B layer for 67 MB of K.  Same arithmetic, same order.
constexpr int SCL_WGS = 32, SCL_LOADERS = 4;
template <int HS>
static __global__ __launch_bounds__(64 * (4 + SCL_LOADERS), 1) void attn_scores_loop_kernel(const AttnArgs a, int n_tiles_max) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    constexpr int hs = HS, half = HS / 2, pitch = HS + 4, q4 = HS / 4;
    const int kvmul = a.n_heads / a.n_kv_heads;
    float* q_s = sm;                               // [kvmul][hs]
    float* kt = q_s + kvmul * hs;                  // [2][ATT_TT][hs + 4]
    float* krow_s = kt + 2 * ATT_TT * pitch;       // [hs] this position's rotated key (owner workgroup)
    float* cr_s = krow_s + hs;                     // [hs/2]
    float* ci_s = cr_s + half;                     // [hs/2]
    const int t = threadIdx.x, nthr = blockDim.x;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int w = blockIdx.x, kvh = blockIdx.y;
    const int pos = a.dyn[1];
    const int ntile = pos / ATT_TT + 1;
    if (w >= ntile) return;
    const int cnt = (ntile - w + SCL_WGS - 1) / SCL_WGS;             // my tiles: w + i SCL_WGS, i < cnt
    const bool owner = ((ntile - 1) % SCL_WGS) == w;                 // the tile that holds `pos` is mine
    // ---- once per workgroup: q of the group's heads (+ bias), the RoPE row of `pos`, per-head RMSNorm (qwen3), RoPE; the owner also
    // rotates this position's key and writes the KV row (InferenceCore.java:75-93)
    for (int i = t; i < kvmul * hs; i += nthr) q_s[i] = a.bq ? a.qkv[(kvh * kvmul) * hs + i] + a.bq[(kvh * kvmul) * hs + i] : a.qkv[(kvh * kvmul) * hs + i];
    for (int i = t; i < half; i += nthr) { cr_s[i] = a.rope_cr[(size_t)pos * half + i]; ci_s[i] = a.rope_ci[(size_t)pos * half + i]; }
    float vraw = 0.f;
    if (owner && t < hs) {
        krow_s[t] = a.bk ? a.qkv[a.q_dim + kvh * hs + t] + a.bk[kvh * hs + t] : a.qkv[a.q_dim + kvh * hs + t];
        vraw = a.bv ? a.qkv[a.q_dim + a.kv_dim + kvh * hs + t] + a.bv[kvh * hs + t] : a.qkv[a.q_dim + a.kv_dim + kvh * hs + t];
    }
    __syncthreads();
    if (a.arch == 1) {
        const int nvec = kvmul + (owner ? 1 : 0);
        for (int vec = wave; vec < nvec; vec += nthr >> 6) head_rmsnorm_wave(vec < kvmul ? q_s + vec * hs : krow_s, vec < kvmul ? a.qnorm : a.knorm, hs, a.eps, t & 63);
        __syncthreads();
    }
    for (int h = 0; h < kvmul; ++h) rope_head(q_s + h * hs, hs, cr_s, ci_s, a.arch, t, nthr);
    if (owner) rope_head(krow_s, hs, cr_s, ci_s, a.arch, t, nthr);
    __syncthreads();
    if (owner && t < hs) {
        a.kcache[(size_t)pos * a.kv_dim + kvh * hs + t] = krow_s[t];
        a.vcache[(size_t)pos * a.kv_dim + kvh * hs + t] = vraw;
    }
    if (wave >= kvmul) {
        // ------------------------------------------------------------------ loaders
        if (wave >= kvmul + SCL_LOADERS) return;                   // (blocks are launched with kvmul + SCL_LOADERS wavefronts: none)
        const int lt = t - 64 * kvmul;                              // 0 .. 255
        constexpr int PER = ATT_TT * q4 / (64 * SCL_LOADERS);      // float4 pieces of a tile per loader thread: 8 (hs 128) / 4 (hs 64)
        // three tiles in flight in NAMED registers (an array carried across the barriers of a loop is kept in scratch by hipcc)
#define SCL_DECL(S_) float4 S_##0 = {0.f, 0.f, 0.f, 0.f}, S_##1 = S_##0, S_##2 = S_##0, S_##3 = S_##0, S_##4 = S_##0, S_##5 = S_##0, S_##6 = S_##0, S_##7 = S_##0
        SCL_DECL(k0_); SCL_DECL(k1_); SCL_DECL(k2_);
#undef SCL_DECL
        const float* kb = a.kcache + (size_t)kvh * hs;
        // rows past pos - 1 do not exist in the cache yet: clamped (row pos is taken from krow_s, later rows are never stored as scores)
#define SCL_LOAD1(S_, U_, T0_) do { if ((U_) < PER) { const int p_ = lt + 256 * (U_), row_ = max(min((T0_) + p_ / q4, pos - 1), 0); \
            S_##U_ = *reinterpret_cast<const float4*>(kb + (size_t)row_ * a.kv_dim + 4 * (p_ % q4)); } } while (0)
#define SCL_LOAD(S_, I_) do { \
            const int t0_ = (w + min((I_), cnt - 1) * SCL_WGS) * ATT_TT; \
            SCL_LOAD1(S_, 0, t0_); SCL_LOAD1(S_, 1, t0_); SCL_LOAD1(S_, 2, t0_); SCL_LOAD1(S_, 3, t0_); \
            SCL_LOAD1(S_, 4, t0_); SCL_LOAD1(S_, 5, t0_); SCL_LOAD1(S_, 6, t0_); SCL_LOAD1(S_, 7, t0_); \
        } while (0)
        // (only rows that exist in the cache are written: in the tile that holds pos another wavefront writes row pos from krow_s, and a
        // clamped copy of row pos - 1 landing there afterwards was a write-write race — position 130 = row 2 of its tile failed, 128 / 129 not)
#define SCL_STORE1(S_, U_, KD_) do { if ((U_) < PER) { const int p_ = lt + 256 * (U_); \
            if (st0_ + p_ / q4 < pos) *reinterpret_cast<float4*>((KD_) + (p_ / q4) * pitch + 4 * (p_ % q4)) = S_##U_; } } while (0)
#define SCL_STORE(S_, I_) do { \
            float* kd_ = kt + ((I_) & 1) * ATT_TT * pitch; \
            const int st0_ = (w + (I_) * SCL_WGS) * ATT_TT; \
            SCL_STORE1(S_, 0, kd_); SCL_STORE1(S_, 1, kd_); SCL_STORE1(S_, 2, kd_); SCL_STORE1(S_, 3, kd_); \
            SCL_STORE1(S_, 4, kd_); SCL_STORE1(S_, 5, kd_); SCL_STORE1(S_, 6, kd_); SCL_STORE1(S_, 7, kd_); \
            if (owner && (I_) == cnt - 1 && lt < q4)        /* the tile that holds pos: its row is this step's own key */ \
                *reinterpret_cast<float4*>(kd_ + (pos & (ATT_TT - 1)) * pitch + 4 * lt) = *reinterpret_cast<const float4*>(krow_s + 4 * lt); \
        } while (0)
#define SCL_STEP(S_, I_) do { \
            if ((I_) < cnt) {                               /* workgroup-uniform: every wavefront runs exactly cnt barriers */ \
                PV_T(const long long s0_ = clock64();) \
                SCL_STORE(S_, (I_)); \
                PV_T(const long long b0_ = clock64(); lst_ += b0_ - s0_;) \
                lds_barrier(); \
                PV_T(lbw_ += clock64() - b0_;) \
            } \
            SCL_LOAD(S_, (I_) + 3);                         /* unconditional (clamped): no wait at the end of a branch */ \
        } while (0)
        PV_T(long long lst_ = 0, lbw_ = 0; const long long lts_ = clock64();)
        SCL_LOAD(k0_, 0); SCL_LOAD(k1_, 1); SCL_LOAD(k2_, 2);
        for (int i = 0; i < cnt; i += 3) { SCL_STEP(k0_, i); SCL_STEP(k1_, i + 1); SCL_STEP(k2_, i + 2); }
        PV_T(if (blockIdx.x == 0 && blockIdx.y == 0 && lt == 0) { gl3_mv_stamp[8] = lst_; gl3_mv_stamp[9] = lbw_; gl3_mv_stamp[10] = clock64() - lts_; })
#undef SCL_LOAD
#undef SCL_LOAD1
#undef SCL_STORE
#undef SCL_STORE1
#undef SCL_STEP
        return;
    }
    // ---------------------------------------------------------------------- chains: wavefront = query head, lane = timestep of the tile
    const int hq = wave, r = t & 63;
    const float sqrt_hs = (float)sqrt((double)hs);
    const float4* q4p = reinterpret_cast<const float4*>(q_s + hq * hs);      // wave-uniform addresses: LDS broadcast reads
    PV_T(long long cbw_ = 0; const long long cts_ = clock64();)
    for (int i = 0; i < cnt; ++i) {
        const int sp = w + i * SCL_WGS, t0 = sp * ATT_TT, t1 = min(pos + 1, t0 + ATT_TT);
        PV_T(const long long cb0_ = clock64();)
        lds_barrier();                                      // tile i is in buffer i & 1 (and buffer (i + 1) & 1 may be overwritten)
        PV_T(cbw_ += clock64() - cb0_;)
        // every lane runs the chain (rows past the tile's end read stale LDS and are not stored): no divergence around the pinned reads
        const float4* kk4 = reinterpret_cast<const float4*>(kt + (i & 1) * ATT_TT * pitch + r * pitch);
        float score = 0.f;                                  // strict j order, mul then add (FloatTensor.scalarDot)
        // four quads of q and of k a group, the next group's eight reads in flight under the current group's 16 multiply-add pairs; the
        // sched_barriers keep that order (left alone the scheduler sinks every read next to its use and the chain waits an LDS round
        // trip per quad).  (q in HS registers instead of LDS reads needs more than the 256 registers of two wavefronts per SIMD: the
        // spilled part was re-read from scratch every tile, 44 cycles per element.)
        float4 qa[4], ka[4], qb[4], kb[4];
#define SCL_RD(Q_, K_, Q0_) do { _Pragma("unroll") for (int u = 0; u < 4; ++u) { Q_[u] = q4p[(Q0_) + u]; K_[u] = kk4[(Q0_) + u]; } } while (0)
#define SCL_USE(Q_, K_) do { _Pragma("unroll") for (int u = 0; u < 4; ++u) { \
            score = score + mul_f32_scalar(Q_[u].x, K_[u].x); score = score + mul_f32_scalar(Q_[u].y, K_[u].y); \
            score = score + mul_f32_scalar(Q_[u].z, K_[u].z); score = score + mul_f32_scalar(Q_[u].w, K_[u].w); } } while (0)
        SCL_RD(qa, ka, 0);
#pragma unroll
        for (int g = 0; g < HS / 16; g += 2) {
            SCL_RD(qb, kb, 4 * (g + 1));
            __builtin_amdgcn_sched_barrier(0);
            SCL_USE(qa, ka);
            __builtin_amdgcn_sched_barrier(0);
            if (g + 2 < HS / 16) SCL_RD(qa, ka, 4 * (g + 2));
            __builtin_amdgcn_sched_barrier(0);
            SCL_USE(qb, kb);
            __builtin_amdgcn_sched_barrier(0);
        }
#undef SCL_RD
#undef SCL_USE
        float my_score = -INFINITY;
        if (t0 + r < t1) {
            my_score = a.att_mul != 0.f ? score * a.att_mul : score / sqrt_hs;
            a.att[(size_t)(kvh * kvmul + hq) * a.att_stride + t0 + r] = my_score;
        }
        const float m = wave_max(my_score);                 // the tile's maximum per head (attn_exp_kernel folds the tiles)
        if (r == 0) a.tmax[(size_t)(kvh * kvmul + hq) * n_tiles_max + sp] = m;
    }
    PV_T(if (blockIdx.x == 0 && blockIdx.y == 0 && t == 0) { gl3_mv_stamp[5] = cbw_; gl3_mv_stamp[6] = clock64() - cts_; gl3_mv_stamp[7] = cnt; })
}

// ---------------------------------------------------------------------------------------------------
// Short-context decode attention in ONE launch (positions < AF_MAXN): RoPE + KV write + scores + softmax + weighted V
// sum.  Same arithmetic and order as the two-kernel path above/below; what it saves is a launch, two kernel boundaries
// and three dependent global round trips (scores -> HBM -> softmax, V after the softmax), which is most of the decode
// attention time at tg128 depth 0.
// Grid = n_heads (one workgroup per QUERY head), block = 256.  The kvMul query heads of a group each stage their own
// copy of the group's K / V rows (<= 64 KB each at 128 positions, L2 / Infinity-Cache hits after the first): a workgroup's
// load rate is bounded per CU (~10 B/clk), so 32 workgroups x 128 KB beat 8 workgroups x 135 KB + 4 heads each (the first
// version of this kernel, one workgroup per kv head, lost to the two-kernel path for head_size 128 for that reason).
// RoPE of k is recomputed by every head of the group (128 elements); only the group's first head writes the KV row.
//   LDS: q[hs] | K[AF_MAXN][hs+4] | V[AF_MAXN][hs] | e[AF_MAXN] | rope row | red[16]
constexpr int AF_MAXN = 128;
// group = query heads per workgroup: 1 (single-token decode: one workgroup per query head) or kvMul (static-batched decode:
// one workgroup per (kv head, token) serves the whole group from one staged K / V tile, so that 32 tokens x 8 kv heads are 256
// workgroups = one wave of workgroups on the chip instead of four)
__host__ __device__ inline size_t attn_head_smem(int hs, int group = 1) {
    return ((size_t)group * hs + (size_t)AF_MAXN * (hs + 4) + (size_t)AF_MAXN * hs + (size_t)group * AF_MAXN + hs + 16) * 4;
}

// HS: head size as a compile-time constant (64 / 128: LDS row strides become immediate offsets and the index arithmetic folds), 0 = a.hs
template <int HS>
static __global__ __launch_bounds__(256, 1) void attn_head_kernel(const AttnArgs a) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int hs = HS > 0 ? HS : a.hs, kvmul = a.n_heads / a.n_kv_heads, half = hs >> 1, pitch = hs + 4, q4 = hs >> 2;
    // float4 index -> (row, column quad) of a K / V tile: shift / mask for the power-of-two head sizes, division otherwise
    // (head_size 96: Phi-3-mini / Phi-3.5-mini, forwardJavaPhi3 with headSize = dim / heads)
    const int q4sh = (q4 & (q4 - 1)) == 0 ? __ffs(q4) - 1 : -1;
    auto row_of = [&](int i) { return q4sh >= 0 ? i >> q4sh : i / q4; };
    auto col_of = [&](int i, int row) { return q4sh >= 0 ? i & (q4 - 1) : i - row * q4; };
    const int G = a.group > 1 ? a.group : 1;
    float* q_s = sm;                                 // [G][hs]
    float* kt = q_s + G * hs;
    float* vt = kt + AF_MAXN * pitch;
    float* e_s = vt + AF_MAXN * hs;                  // [G][AF_MAXN]
    float* cr_s = e_s + G * AF_MAXN;
    float* ci_s = cr_s + half;
    float* red = ci_s + half;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int h0 = blockIdx.x * G, kvh = h0 / kvmul;          // first query head of this workgroup
    const bool owner = (h0 % kvmul) == 0;
    // static-batched decode: one grid row per token, each with its own sequence (KV cache) and position
    const int bt = blockIdx.y;
    const int pos = a.seqv ? a.posv[bt] : a.dyn[1], n = pos + 1;
    const size_t kvoff = a.seqv ? (size_t)a.seqv[bt] * a.seq_stride : 0;
    const float* qkv = a.qkv + (size_t)bt * a.qkv_stride;
    float* kcache = a.kcache + kvoff;
    float* vcache = a.vcache + kvoff;
    ATT_STAMP(0);
    // ---- one global round trip: cached K / V rows (both in flight at once), raw q / k / v of this token, the RoPE row
    const int nk4 = pos * q4;
    constexpr int KMAX = 16;                         // 127 rows x 32 float4 / 256 threads (head_size 128)
    float4 kreg[KMAX], vreg[KMAX];                   // initialised + uniform condition + clamped index: stays in VGPRs (see attn_scores_kernel)
    const bool in_regs = nk4 <= KMAX * 256;
    const int per = in_regs ? (nk4 + 255) >> 8 : 0;
#pragma unroll
    for (int u = 0; u < KMAX; ++u) {
        kreg[u] = make_float4(0.f, 0.f, 0.f, 0.f);
        vreg[u] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (u < per) {
            const int i = min(t + u * 256, nk4 - 1), ri = row_of(i);
            const size_t off = (size_t)ri * a.kv_dim + kvh * hs + 4 * col_of(i, ri);
            kreg[u] = *reinterpret_cast<const float4*>(kcache + off);
            vreg[u] = *reinterpret_cast<const float4*>(vcache + off);
        }
    }
    for (int i = t; i < G * hs; i += 256) q_s[i] = a.bq ? qkv[h0 * hs + i] + a.bq[h0 * hs + i] : qkv[h0 * hs + i];
    for (int i = t; i < half; i += 256) { cr_s[i] = a.rope_cr[(size_t)pos * half + i]; ci_s[i] = a.rope_ci[(size_t)pos * half + i]; }
    float* krow = kt + pos * pitch;
    for (int i = t; i < hs; i += 256) {
        krow[i] = a.bk ? qkv[a.q_dim + kvh * hs + i] + a.bk[kvh * hs + i] : qkv[a.q_dim + kvh * hs + i];
        vt[pos * hs + i] = a.bv ? qkv[a.q_dim + a.kv_dim + kvh * hs + i] + a.bv[kvh * hs + i] : qkv[a.q_dim + a.kv_dim + kvh * hs + i];
    }
    if (in_regs) {
#pragma unroll
        for (int u = 0; u < KMAX; ++u) {
            const int i = t + u * 256;
            if (i < nk4) {
                const int ri = row_of(i), ci = col_of(i, ri);
                *reinterpret_cast<float4*>(kt + ri * pitch + 4 * ci) = kreg[u];
                *reinterpret_cast<float4*>(vt + ri * hs + 4 * ci) = vreg[u];
            }
        }
    } else {                                         // head_size 256: straight to LDS
        for (int i = t; i < nk4; i += 256) {
            const int ri = row_of(i), ci = col_of(i, ri);
            const size_t off = (size_t)ri * a.kv_dim + kvh * hs + 4 * ci;
            *reinterpret_cast<float4*>(kt + ri * pitch + 4 * ci) = *reinterpret_cast<const float4*>(kcache + off);
            *reinterpret_cast<float4*>(vt + ri * hs + 4 * ci) = *reinterpret_cast<const float4*>(vcache + off);
        }
    }
    __syncthreads();
    ATT_STAMP(1);
    if (a.arch == 1) {                               // qwen3: per-head RMSNorm of q and k (strict-order sum, one wavefront per head)
        for (int vec = wave; vec <= G; vec += 4)                   // G query heads + the key: one wavefront each
            head_rmsnorm_wave(vec < G ? q_s + vec * hs : krow, vec < G ? a.qnorm : a.knorm, hs, a.eps, lane);
        __syncthreads();
    }
    for (int g = 0; g < G; ++g) rope_head(q_s + g * hs, hs, cr_s, ci_s, a.arch, t, 256);
    rope_head(krow, hs, cr_s, ci_s, a.arch, t, 256);
    __syncthreads();
    if (owner) {                                     // KV write, InferenceCore.java:92-93
        for (int i = t; i < hs; i += 256) {
            kcache[(size_t)pos * a.kv_dim + kvh * hs + i] = krow[i];
            vcache[(size_t)pos * a.kv_dim + kvh * hs + i] = vt[pos * hs + i];
        }
    }
    ATT_STAMP(2);
    // ---- scores: one (query head, timestep) pair per thread and pass; strict j order, mul then add (FloatTensor.scalarDot)
    const float sqrt_hs = (float)sqrt((double)hs);
    float sc = -INFINITY;                            // group == 1: this thread's score (timestep t)
    for (int idx = t; idx < G * n; idx += 256) {
        const int g = idx / n, tt = idx - g * n;
        const float* q = q_s + g * hs;
        const float* kk = kt + tt * pitch;
        float score = 0.f;
        float4 qv = *reinterpret_cast<const float4*>(q), kv = *reinterpret_cast<const float4*>(kk);
        for (int j = 4; j < hs; j += 4) {
            const float4 qn = *reinterpret_cast<const float4*>(q + j), kn = *reinterpret_cast<const float4*>(kk + j);
            score = score + qv.x * kv.x; score = score + qv.y * kv.y; score = score + qv.z * kv.z; score = score + qv.w * kv.w;
            qv = qn; kv = kn;
        }
        score = score + qv.x * kv.x; score = score + qv.y * kv.y; score = score + qv.z * kv.z; score = score + qv.w * kv.w;
        sc = a.att_mul != 0.f ? score * a.att_mul : score / sqrt_hs;
        if (G > 1) e_s[g * AF_MAXN + tt] = sc;
    }
    ATT_STAMP(3);
    // ---- softmax (FloatTensor.softmaxInPlace :211-219): max, exp in double, strict sum, divide
    if (G == 1) {                                    // one head: all timesteps in parallel across the workgroup
        {
            const float wm = wave_max(t < n ? sc : -INFINITY);
            if (lane == 0) red[wave] = wm;
        }
        __syncthreads();
        const float mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
        float ex = 0.f;
        if (t < n) { ex = (float)exp((double)(sc - mx)); e_s[t] = ex; }
        __syncthreads();
        ATT_STAMP(4);
        if (wave == 0) { const float sum = seq_sum_lds_ring(e_s, n); if (lane == 0) red[4] = sum; }
        __syncthreads();
        if (t < n) e_s[t] = ex / red[4];
        __syncthreads();
    } else {                                         // a group: one wavefront per head, everything wave-local (LDS runs a wavefront in order)
        __syncthreads();
        for (int g = wave; g < G; g += 4) {
            float* e = e_s + g * AF_MAXN;
            const float s0 = lane < n ? e[lane] : -INFINITY, s1 = lane + 64 < n ? e[lane + 64] : -INFINITY;
            const float mx = wave_max(fmaxf(s0, s1));
            const float e0 = lane < n ? (float)exp((double)(s0 - mx)) : 0.f, e1 = lane + 64 < n ? (float)exp((double)(s1 - mx)) : 0.f;
            if (lane < n) e[lane] = e0;
            if (lane + 64 < n) e[lane + 64] = e1;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            const float sum = seq_sum_lds_ring(e, n);
            __builtin_amdgcn_wave_barrier();
            if (lane < n) e[lane] = e0 / sum;
            if (lane + 64 < n) e[lane + 64] = e1 / sum;
        }
        __syncthreads();
    }
    ATT_STAMP(5);
    // ---- weighted V sum, t ascending: xb[j] = a_t * v[t][j] + xb[j] (saxpyInPlace :221-227); one (head, output column) per thread and pass
    for (int idx = t; idx < G * hs; idx += 256) {
        const int g = idx / hs, j = idx - g * hs;
        const float* e = e_s + g * AF_MAXN;
        const float* vj = vt + j;
        float acc = 0.f;
        int tt = 0;
        // Groups of 4 timesteps (one 16-byte read of the weights, four 4-byte reads of this column of V), three groups = 15 LDS
        // reads in flight ahead of the dependent mul / add chain.  The sched_barrier after every refill keeps the order "use group A,
        // refill A, use group B, ..."; left alone the scheduler puts the five reads of a group right in front of their use and the
        // wavefront (the only one on its SIMD) waits a full LDS round trip per 4 timesteps (seen in the ISA: s_waitcnt lgkmcnt(0) two
        // instructions after the reads, ~150 cycles per iteration for 40 cycles of arithmetic).
        const int NG = n >> 2;
        if (NG >= 3) {
            float4 ea, eb, ec;
            float a0, a1, a2, a3, b0, b1, b2, b3, c0, c1, c2, c3;
#define PV_LD(G_, E_, V0_, V1_, V2_, V3_) do { const int r_ = 4 * min((G_), NG - 1); E_ = *reinterpret_cast<const float4*>(e + r_); \
            const float* vr_ = vj + r_ * hs; V0_ = vr_[0]; V1_ = vr_[hs]; V2_ = vr_[2 * hs]; V3_ = vr_[3 * hs]; } while (0)
#define PV_ACC(E_, V0_, V1_, V2_, V3_) do { acc = mul_f32_scalar(E_.x, V0_) + acc; acc = mul_f32_scalar(E_.y, V1_) + acc; \
            acc = mul_f32_scalar(E_.z, V2_) + acc; acc = mul_f32_scalar(E_.w, V3_) + acc; } while (0)
            PV_LD(0, ea, a0, a1, a2, a3); PV_LD(1, eb, b0, b1, b2, b3); PV_LD(2, ec, c0, c1, c2, c3);
            int gq = 0;
            for (; gq + 3 <= NG; gq += 3) {
                PV_ACC(ea, a0, a1, a2, a3); PV_LD(gq + 3, ea, a0, a1, a2, a3); __builtin_amdgcn_sched_barrier(0);
                PV_ACC(eb, b0, b1, b2, b3); PV_LD(gq + 4, eb, b0, b1, b2, b3); __builtin_amdgcn_sched_barrier(0);
                PV_ACC(ec, c0, c1, c2, c3); PV_LD(gq + 5, ec, c0, c1, c2, c3); __builtin_amdgcn_sched_barrier(0);
            }
            if (gq < NG) { PV_ACC(ea, a0, a1, a2, a3); ++gq; }
            if (gq < NG) { PV_ACC(eb, b0, b1, b2, b3); ++gq; }
#undef PV_ACC
#undef PV_LD
            tt = 4 * NG;
        }
        for (; tt < n; ++tt) acc = e[tt] * vj[tt * hs] + acc;
        if (a.xq_out) {
            // Q8_0 activation quantisation of the 32-element block this half-wavefront holds (Q8_0FloatTensor.java:96-118; head
            // sizes are multiples of 32, so a block never straddles heads or passes): block maximum over 32 lanes, round half
            // away from zero, four lanes' bytes packed by the quad's first lane.
            float am = fabsf(acc);
            am = fmaxf(am, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, am), 0xB1, 0xf, 0xf, false)));
            am = fmaxf(am, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, am), 0x4E, 0xf, 0xf, false)));
            am = fmaxf(am, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, am), 0x141, 0xf, 0xf, false)));
            am = fmaxf(am, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, am), 0x140, 0xf, 0xf, false)));
            am = fmaxf(am, __shfl_xor(am, 16, 64));
            const float qs = am / 127.0f;
            const float ainv = qs != 0.f ? 1.0f / qs : 0.f;
            const float sv = acc * ainv;
            const int q0 = (int)(sv + copysignf(0.5f, sv)) & 0xFF;
            const int q1 = __builtin_amdgcn_update_dpp(0, q0, 0x55, 0xf, 0xf, false);      // quad_perm [1,1,1,1]
            const int q2 = __builtin_amdgcn_update_dpp(0, q0, 0xAA, 0xf, 0xf, false);      // [2,2,2,2]
            const int q3 = __builtin_amdgcn_update_dpp(0, q0, 0xFF, 0xf, 0xf, false);      // [3,3,3,3]
            const int el = (h0 + g) * hs + j;                                                // element of the token's q_dim row
            if ((lane & 3) == 0)
                *reinterpret_cast<uint32_t*>(a.xq_out + bdq_offset(el >> 2, bt, a.xq_slots)) = (uint32_t)q0 | ((uint32_t)q1 << 8) | ((uint32_t)q2 << 16) | ((uint32_t)q3 << 24);
            if ((lane & 31) == 0) a.xs_out[bds_offset(el >> 5, bt, a.xq_slots)] = (float)(_Float16)qs;
        } else {
            float* o = a.xb + (size_t)bt * a.xb_stride + (size_t)(h0 + g) * hs + j;
            *o = acc;
            if (a.tp) tp_push_store(a.tp->p, o, acc);
        }
    }
    ATT_STAMP(6);
    if (a.tp) tp_publish(a.tp->p, gridDim.x * gridDim.y * 4);
}

// Picks the instantiation for the head size (host side).
template <typename F>
static inline void attn_head_dispatch(int hs, F&& f) {
    if (hs == 128) f(attn_head_kernel<128>);
    else if (hs == 64) f(attn_head_kernel<64>);
    else f(attn_head_kernel<0>);
}

// Decode attention, part 2: softmax + weighted V sum.   Grid = n_heads x hs/16, block = 256.
//   FloatTensor.softmaxInPlace :211-219 (max, exp in double, strict sum, divide); saxpyInPlace :221-227 with
//   t ascending: xb[j] = a_t * v[t][j] + xb[j].
// A workgroup owns 16 output columns of one head.  All 256 threads put the V slab [n][16] in flight at once while the
// head's softmax is computed (elementwise parts on all threads, the strict sum on one wavefront); the products
// a_t * v_tj are rounded on the VALU and added in order on the MFMA pipe (16x16x4, B = 1.0).
//   Dynamic LDS: e[ctx] | vbuf[PV_ROWS][16]; contexts longer than PV_ROWS are processed in slabs of PV_ROWS rows.
constexpr int PV_COLS = 16;
constexpr int PV_ROWS = 1024;

static __global__ __launch_bounds__(256) void attn_softmax_pv_kernel(const AttnArgs a) {
    extern __shared__ __attribute__((aligned(16))) float e_s[];
    const int hs = a.hs, kvmul = a.n_heads / a.n_kv_heads;
    const int nj = hs / PV_COLS;
    const int h = blockIdx.x / nj, j0 = (blockIdx.x % nj) * PV_COLS;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, kvh = h / kvmul;
    const int n = a.dyn[1] + 1;
    float* vbuf = e_s + a.win;                        // win = min(ctx rounded up to 4, window)
    const float* vbase = a.vcache + kvh * hs + j0;
    constexpr int SU = PV_ROWS * 4 / 256;         // thread = (row t>>2 + 64u, column quad t&3)
    float4 sreg[SU];
#pragma unroll
    for (int u = 0; u < SU; ++u) sreg[u] = make_float4(0.f, 0.f, 0.f, 0.f);
    auto stage_issue = [&](int r0, int nr) {         // uniform condition + clamped row: sreg stays in VGPRs, loads issue back to back
#pragma unroll
        for (int u = 0; u < SU; ++u) {
            if (64 * u < nr) {
                const int row = min((t >> 2) + 64 * u, nr - 1);
                sreg[u] = *reinterpret_cast<const float4*>(vbase + (size_t)(r0 + row) * a.kv_dim + 4 * (t & 3));
            }
        }
    };
    auto stage_commit = [&](int nr) {
#pragma unroll
        for (int u = 0; u < SU; ++u)
            if ((t >> 2) + 64 * u < nr) *reinterpret_cast<float4*>(vbuf + 4 * (t + 256 * u)) = sreg[u];
    };
    __shared__ float red_s[8];
    ATT_STAMP(0);
    stage_issue(0, min(n, PV_ROWS));
    ATT_STAMP(1);
    const float* sc = a.att + (size_t)h * a.att_stride;
    const int W = a.win;                              // LDS window of the row
    float mx = -INFINITY, sum = 0.f;
    {   // softmax of the head: max, exp in double, strictly sequential f32 sum, divide (FloatTensor.softmaxInPlace :195-219)
        if (n <= W) {
            for (int i = t; i < n; i += 256) { const float s = sc[i]; e_s[i] = s; mx = fmaxf(mx, s); }
        } else {
            for (int i = t; i < n; i += 256) mx = fmaxf(mx, sc[i]);
        }
        mx = wave_max(mx);
        if (lane == 0) red_s[wave] = mx;
        __syncthreads();
        mx = fmaxf(fmaxf(red_s[0], red_s[1]), fmaxf(red_s[2], red_s[3]));
        ATT_STAMP(2);
        if (n <= W) {
            for (int i = t; i < n; i += 256) e_s[i] = (float)exp((double)(e_s[i] - mx));
            __syncthreads();
            ATT_STAMP(3);
            if (wave == 0) { const float sm = seq_sum_lds_ring(e_s, n); if (lane == 0) red_s[4] = sm; }      // reads pinned three groups ahead: ~6 instead of ~11 cycles per element
            __syncthreads();
            ATT_STAMP(4);
            sum = red_s[4];
            for (int i = t; i < n; i += 256) e_s[i] = e_s[i] / sum;
        } else {
            // Row longer than the window (contexts beyond ~16 k positions): the sum runs over the windows in order, carrying the
            // running value; the numerators are recomputed per window below (same exp of the same argument -> same bits).
            for (int c0 = 0; c0 < n; c0 += W) {
                const int len = min(W, n - c0);
                __syncthreads();
                for (int i = t; i < len; i += 256) e_s[i] = (float)exp((double)(sc[c0 + i] - mx));
                __syncthreads();
                if (wave == 0) { const float sm = seq_sum_lds_ring(e_s, len, c0 == 0 ? 0.f : red_s[4]); if (lane == 0) red_s[4] = sm; }
            }
            __syncthreads();
            sum = red_s[4];
        }
    }
    stage_commit(min(n, PV_ROWS));
    __syncthreads();
    ATT_STAMP(5);
    v4f acc = {0.f, 0.f, 0.f, 0.f};
    for (int c0 = 0; c0 < n; c0 += W) {                // one trip unless the row is longer than the window
        const int clen = min(W, n - c0);
        if (n > W) {
            __syncthreads();
            for (int i = t; i < clen; i += 256) e_s[i] = (float)exp((double)(sc[c0 + i] - mx)) / sum;
            __syncthreads();
        }
        for (int r0 = c0; r0 < c0 + clen; r0 += PV_ROWS) {
            const int nr = min(PV_ROWS, c0 + clen - r0);
            if (r0 > 0) { __syncthreads(); stage_issue(r0, nr); stage_commit(nr); __syncthreads(); }
            if (wave == 0) {                               // lane l = column l&15, timestep 4g + (l>>4)
                const int col = lane & 15, k = lane >> 4;
                const float* ap = e_s + (r0 - c0);
                int g = 0;
                for (; 4 * g + 16 <= nr; g += 4) {          // 4 MFMAs per iteration, operands fetched first
                    float p[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) { const int r = 4 * (g + u) + k; p[u] = ap[r] * vbuf[r * PV_COLS + col]; }
#pragma unroll
                    for (int u = 0; u < 4; ++u) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(p[u], 1.0f, acc, 0, 0, 0);
                }
                for (; 4 * g < nr; ++g) {
                    const int r = 4 * g + k;
                    const float p = r < nr ? ap[r] * vbuf[r * PV_COLS + col] : 0.f;   // +0 pads the last group
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(p, 1.0f, acc, 0, 0, 0);
                }
            }
        }
    }
    ATT_STAMP(6);
    // D[row][col]: row = column index of the slab = 4*(lane>>4) + reg; every MFMA column holds the same chain
    if (wave == 0 && (lane & 15) == 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float* o = a.xb + h * hs + j0 + 4 * (lane >> 4) + r;
            *o = acc[r];
            if (a.tp) tp_push_store(a.tp->p, o, acc[r]);
        }
    }
    if (a.tp && wave == 0) tp_publish(a.tp->p, gridDim.x);
}

// ---------------------------------------------------------------------------------------------------
// MEASURED SLOWER than the pair and therefore OFF by default (GL3_ATTN_FUSED_MID=1 selects it): 19.0 us per 8B layer at depth 256 and 28.9 us at 512
// against 18.8 / 23.3 for the pair (tg128@d512 352 vs 397 tok/s).  64 workgroups of one wavefront per SIMD expose every LDS and memory latency that
// the pair hides behind 256 column workgroups; kept as the bit-exact record of the experiment (profiles/r06_mid_attention.md).
// Decode attention for positions AF_MAXN .. AM_MAXN - 1 in ONE launch (r6; head_size 64 / 128, kvMul <= 4).  The r4 pair (attn_scores_kernel +
// attn_softmax_pv_kernel) spent 18.8 us per 8B layer at depth 256 and 23.3 us at 512 against 0.3 - 0.6 us of KV read: two launches, a round trip of
// the scores through memory, a softmax recomputed by each of a head's hs / 16 column workgroups with its strict sum on ONE wavefront, and three of the
// four wavefronts idle during the weighted V sum.  Here a workgroup owns (kv head, 16 output columns) for ALL query heads of the group, wavefront =
// query head:
//   * the group's K rows stream through LDS in tiles of AM_TT timesteps (registers one tile ahead), every wavefront runs the strict q . k chains
//     of its head for two timesteps per lane; the scores never leave LDS;
//   * the V slab [n][16] is in flight from the first instruction and lands in LDS behind the scores;
//   * max, (float)exp((double)(s - max)), the strictly sequential sum (LDS reads pinned three groups ahead) and the division run per wavefront, the
//     four heads side by side; a_t * v rounded on the VALU, added in order on the matrix pipe (16x16x4, B = 1.0) — four chains side by side.
//   * RoPE (+ Qwen3 per-head RMSNorm, Qwen2 bias) of q and of this position's key in every workgroup (the key row is a score operand); the KV write by
//     the workgroups of column slab 0.
// Same arithmetic and order as the pair: positions 128 .. 767 of every decode parity test compare np.array_equal.
constexpr int AM_TT = 128;       // timesteps per K tile
constexpr int AM_MAXN = 768;     // rows of the score / V slab buffers: positions < AM_MAXN
template <int HS>
__host__ __device__ constexpr size_t attn_mid_smem() { return ((size_t)4 * HS + HS + HS + 4 * AM_MAXN + (size_t)AM_MAXN * 16 + (size_t)AM_TT * (HS + 4)) * 4; }

template <int HS>
static __global__ __launch_bounds__(256, 1) void attn_mid_kernel(const AttnArgs a) {
    extern __shared__ __attribute__((aligned(16))) float am_sm[];
    constexpr int PITCH = HS + 4, HALF = HS / 2, Q4 = HS / 4;
    constexpr int KREG = AM_TT * Q4 / 256;             // float4s of a K tile per thread
    constexpr int SU = AM_MAXN * 4 / 256;              // float4s of the V slab per thread: thread = (row t >> 2 + 64 u, column quad t & 3)
    float* q_s = am_sm;                                // [4][HS]
    float* kcur = q_s + 4 * HS;                        // [HS] this position's key
    float* cr_s = kcur + HS;
    float* ci_s = cr_s + HALF;
    float* e_s = ci_s + HALF;                          // [4][AM_MAXN] scores, then softmax weights
    float* vbuf = e_s + 4 * AM_MAXN;                   // [AM_MAXN][16]
    float* kt = vbuf + AM_MAXN * 16;                   // [AM_TT][PITCH]
    const int kvmul = a.n_heads / a.n_kv_heads;
    const int t = threadIdx.x, lane = t & 63, hq = t >> 6;
    const int j0 = blockIdx.x * 16, kvh = blockIdx.y;
    const int pos = a.dyn[1], n = pos + 1;
    const bool writer = blockIdx.x == 0;               // this workgroup writes the position's K / V rows into the cache
    // ---- V slab rows 0 .. pos - 1 from the cache (row pos comes from the raw qkv below)
    float4 sreg[SU];
#pragma unroll
    for (int u = 0; u < SU; ++u) {
        sreg[u] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (64 * u < pos) {                            // uniform condition + clamped row ("conditional loads")
            const int row = min((t >> 2) + 64 * u, pos - 1);
            sreg[u] = *reinterpret_cast<const float4*>(a.vcache + (size_t)row * a.kv_dim + kvh * HS + j0 + 4 * (t & 3));
        }
    }
    // ---- first K tile (rows < pos) to registers
    float4 kreg[KREG];
    auto k_issue = [&](int t0) {
        const int nr = min(AM_TT, pos - t0);           // cache rows of the tile (<= 0: none)
#pragma unroll
        for (int u = 0; u < KREG; ++u) {
            kreg[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (nr > 0) {
                const int i = min(t + 256 * u, nr * Q4 - 1);
                kreg[u] = *reinterpret_cast<const float4*>(a.kcache + (size_t)(t0 + i / Q4) * a.kv_dim + kvh * HS + 4 * (i % Q4));
            }
        }
    };
    k_issue(0);
    // ---- raw q of the group's heads, raw k / v of this kv head, the RoPE row of pos
    for (int i = t; i < kvmul * HS; i += 256) q_s[i] = a.bq ? a.qkv[(kvh * kvmul) * HS + i] + a.bq[(kvh * kvmul) * HS + i] : a.qkv[(kvh * kvmul) * HS + i];
    for (int i = t; i < HALF; i += 256) { cr_s[i] = a.rope_cr[(size_t)pos * HALF + i]; ci_s[i] = a.rope_ci[(size_t)pos * HALF + i]; }
    for (int i = t; i < HS; i += 256) kcur[i] = a.bk ? a.qkv[a.q_dim + kvh * HS + i] + a.bk[kvh * HS + i] : a.qkv[a.q_dim + kvh * HS + i];
    float vraw = 0.f;                                  // writer: element t of the v row (HS <= 256); others: column j0 + t for t < 16
    {
        const int i = writer ? t : j0 + t;
        if (writer ? t < HS : t < 16) vraw = a.bv ? a.qkv[a.q_dim + a.kv_dim + kvh * HS + i] + a.bv[kvh * HS + i] : a.qkv[a.q_dim + a.kv_dim + kvh * HS + i];
    }
    __syncthreads();
    if (a.arch == 1) {                                 // Qwen3: per-head RMSNorm of q and k before RoPE, one wavefront per vector
        for (int vec = hq; vec < kvmul + 1; vec += 4) head_rmsnorm_wave(vec < kvmul ? q_s + vec * HS : kcur, vec < kvmul ? a.qnorm : a.knorm, HS, a.eps, lane);
        __syncthreads();
    }
    for (int h = 0; h < kvmul; ++h) rope_head(q_s + h * HS, HS, cr_s, ci_s, a.arch, t, 256);
    rope_head(kcur, HS, cr_s, ci_s, a.arch, t, 256);
    __syncthreads();
    if (writer && t < HS) {                            // KV write, InferenceCore.java:92-93
        a.kcache[(size_t)pos * a.kv_dim + kvh * HS + t] = kcur[t];
        a.vcache[(size_t)pos * a.kv_dim + kvh * HS + t] = vraw;
    }
    if (writer ? (t >= j0 && t < j0 + 16) : t < 16) vbuf[pos * 16 + (writer ? t - j0 : t)] = vraw;      // row pos of the slab (j0 = 0 for the writer)
    // ---- scores: tiles of AM_TT timesteps, lane = timesteps r and r + 64 of the tile, strict j order, mul then add (FloatTensor.scalarDot)
    const float sqrt_hs = (float)sqrt((double)HS);
    float mx = -INFINITY;
    for (int t0 = 0; t0 < n; t0 += AM_TT) {
        const int nr = min(AM_TT, pos - t0);
#pragma unroll
        for (int u = 0; u < KREG; ++u) {
            const int i = t + 256 * u;
            if (nr > 0 && i < nr * Q4) *reinterpret_cast<float4*>(kt + (i / Q4) * PITCH + 4 * (i % Q4)) = kreg[u];
        }
        if (pos >= t0 && pos < t0 + AM_TT && t < Q4) *reinterpret_cast<float4*>(kt + (pos - t0) * PITCH + 4 * t) = *reinterpret_cast<const float4*>(kcur + 4 * t);
        __syncthreads();
        if (t0 + AM_TT < n) k_issue(t0 + AM_TT);       // next tile in flight under this tile's chains
        if (hq < kvmul) {
            const float* q = q_s + hq * HS;
            const int r0 = min(lane, n - 1 - t0), r1 = min(lane + 64, n - 1 - t0);      // clamped rows: lanes past the end recompute the last timestep
            const float* k0 = kt + r0 * PITCH;
            const float* k1 = kt + r1 * PITCH;
            float s0 = 0.f, s1 = 0.f;
            float4 qv = *reinterpret_cast<const float4*>(q), ka = *reinterpret_cast<const float4*>(k0), kb = *reinterpret_cast<const float4*>(k1);
            for (int j = 4; j < HS; j += 4) {
                const float4 qn = *reinterpret_cast<const float4*>(q + j), kan = *reinterpret_cast<const float4*>(k0 + j), kbn = *reinterpret_cast<const float4*>(k1 + j);
                s0 = s0 + qv.x * ka.x; s1 = s1 + qv.x * kb.x; s0 = s0 + qv.y * ka.y; s1 = s1 + qv.y * kb.y;
                s0 = s0 + qv.z * ka.z; s1 = s1 + qv.z * kb.z; s0 = s0 + qv.w * ka.w; s1 = s1 + qv.w * kb.w;
                qv = qn; ka = kan; kb = kbn;
            }
            s0 = s0 + qv.x * ka.x; s1 = s1 + qv.x * kb.x; s0 = s0 + qv.y * ka.y; s1 = s1 + qv.y * kb.y;
            s0 = s0 + qv.z * ka.z; s1 = s1 + qv.z * kb.z; s0 = s0 + qv.w * ka.w; s1 = s1 + qv.w * kb.w;
            s0 = a.att_mul != 0.f ? s0 * a.att_mul : s0 / sqrt_hs;
            s1 = a.att_mul != 0.f ? s1 * a.att_mul : s1 / sqrt_hs;
            if (t0 + lane < n) { e_s[hq * AM_MAXN + t0 + lane] = s0; mx = fmaxf(mx, s0); }
            if (t0 + lane + 64 < n) { e_s[hq * AM_MAXN + t0 + lane + 64] = s1; mx = fmaxf(mx, s1); }
        }
        __syncthreads();
    }
    // ---- softmax of the wavefront's head (FloatTensor.softmaxInPlace :211-219): max, exp in double, strictly sequential sum, divide
    if (hq < kvmul) {
        float* e = e_s + hq * AM_MAXN;
        mx = wave_max(mx);
        for (int i = lane; i < n; i += 64) e[i] = (float)exp((double)(e[i] - mx));
        const float sum = seq_sum_lds_ring(e, n);
        for (int i = lane; i < n; i += 64) e[i] = e[i] / sum;
    }
#pragma unroll
    for (int u = 0; u < SU; ++u)
        if ((t >> 2) + 64 * u < pos) *reinterpret_cast<float4*>(vbuf + 4 * (t + 256 * u)) = sreg[u];
    __syncthreads();
    // ---- weighted V sum of the wavefront's head: xb[j] = a_t * v[t][j] + xb[j], t ascending (saxpyInPlace :221-227); lane = (column l & 15, timestep 4 g + (l >> 4))
    if (hq >= kvmul) return;
    v4f acc = {0.f, 0.f, 0.f, 0.f};
    {
        const int col = lane & 15, k = lane >> 4;
        const float* ap = e_s + hq * AM_MAXN;
        int g = 0;
        for (; 4 * g + 16 <= n; g += 4) {                  // 4 MFMAs per iteration, operands fetched first
            float p[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) { const int r = 4 * (g + u) + k; p[u] = ap[r] * vbuf[r * 16 + col]; }
#pragma unroll
            for (int u = 0; u < 4; ++u) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(p[u], 1.0f, acc, 0, 0, 0);
        }
        for (; 4 * g < n; ++g) {
            const int r = 4 * g + k;
            const float p = r < n ? ap[r] * vbuf[r * 16 + col] : 0.f;      // +0 pads the last group
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(p, 1.0f, acc, 0, 0, 0);
        }
    }
    if ((lane & 15) == 0) {                            // D[row][col]: row = slab column 4 * (lane >> 4) + reg; every MFMA column holds the same chain
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float* o = a.xb + (kvh * kvmul + hq) * HS + j0 + 4 * (lane >> 4) + r;
            *o = acc[r];
            if (a.tp) tp_push_store(a.tp->p, o, acc[r]);
        }
    }
    if (a.tp) tp_publish(a.tp->p, gridDim.x * gridDim.y * kvmul);
}

// ---------------------------------------------------------------------------------------------------
// Long-context decode attention, round 5: the softmax and the weighted V sum as TWO launches shaped after what bounds them.
// At depth (llama-bench -d) the one-launch pair above spends its time in two strictly sequential chains per workgroup: the
// softmax denominator (n dependent adds at ~11 cycles each, recomputed by each of the hs / 16 workgroups of a head) and the
// MFMA-fed weighted V sum (n / 4 dependent 16x16x4 steps of ~40 cycles): 87 us per 8B layer at depth 4096, 312 us at 16384
// (profiles/r05_tg_depth.md).  Same arithmetic, same order, different evaluation:
//   attn_softmax_kernel  one workgroup per query head: max; (float)exp((double)(s - max)); the strictly sequential f32 sum of
//       the n non-negative numerators evaluated EXACTLY in parallel, 4096 at a time, each chunk continued from the exact running
//       value of the previous one (gl3_seqsum.h, as the samplers do over the vocabulary); a_t = e_t / sum back into att[].
//   attn_pv_kernel       one workgroup per (kv head, 16 output columns) serving four query heads of the group
//       from one pass over V: lane (g, col) of the chain wavefront owns the chain xb[g][col] = a_t * v[t][col] + xb[g][col],
//       t ascending, and does nothing but `acc = p + acc` on products the eight helper wavefronts rounded for it (one v_mul_f32
//       each, straight from global operands) and parked in LDS in the chain's read order: one ds_read_b128 + four v_add_f32 per
//       four timesteps = 5 issue slots per 4 steps for the wavefront everything waits for.  (A lone wavefront issues one
//       instruction per 4 cycles, so computing the products itself would cost 8 cycles per timestep, through LDS-staged operands
//       10; every product crosses LDS twice, 64 KB per 128-timestep tile = 80 % of what LDS moves in the chain's 640 cycles.)
//       Workgroup ids are dealt so that the slabs of one kv head share an XCD (blockIdx % kv_heads = kv head): a V line is
//       fetched into one L2, once.
constexpr int PVT = 128;                   // timesteps per product tile
constexpr int PV_G = 4, PV_COLS16 = 16;    // a workgroup serves PV_G query heads of one kv head x 16 output columns: 64 chains = one wavefront
constexpr int PV_HELPERS = 8, PV_RING = 4;  // helper wavefronts (wavefront 0 is the chain); tiles their operands are requested ahead
constexpr int PV_WAVES = 11;               // wavefronts per workgroup: chain, 8 helpers, and wavefronts 4 and 8 — the chain's SIMD neighbours — which retire at once
constexpr int PV_QP = 64 * 4 + 4;          // floats per timestep quad in LDS: 64 lanes x 4 steps + 4 (rows of consecutive quads 4 banks apart: see the helpers' store)
__host__ __device__ constexpr size_t attn_pv_smem() { return (size_t)2 * (PVT / 4) * PV_QP * 4; }
__host__ __device__ inline int attn_pv_hq(int kvmul) { return (kvmul + PV_G - 1) / PV_G; }       // groups of PV_G query heads per kv head
// floats of the transposed weight buffer att_t[kv head][head quad][t][PV_G] (written by attn_softmax_kernel, read by attn_pv_kernel)
__host__ __device__ inline size_t attn_att_t_floats(int kv_heads, int kvmul, int att_stride) { return (size_t)kv_heads * attn_pv_hq(kvmul) * att_stride * PV_G + (size_t)PVT * PV_G; }


constexpr int SMX_CHUNK = 4096;
__host__ __device__ constexpr size_t attn_sum_smem() { return (size_t)(SMX_CHUNK + 32) * 4 + ss_scratch_bytes(SMX_CHUNK); }
constexpr int EXP_ROW = 1024;              // scores per attn_exp_kernel workgroup

// e_t = (float)exp((double)(s_t - max)) for rows of EXP_ROW scores of one head: grid = (min(ceil(ctx / EXP_ROW), EXP_GRID_MAX), heads), 256 threads,
// workgroup x takes the rows x, x + gridDim.x, ... (r6: a grid of ceil(ctx / EXP_ROW) left 128 x heads workgroups per layer at a 128 k context of
// which all but ceil((pos + 1) / EXP_ROW) rows exit at once).  One workgroup per head (the first form of this path) needed 7 us per 4096 scores
// for the exp alone — the VALU of one CU — so it is spread over the chip.  max = fold of the per-tile maxima attn_scores_kernel left in tmax.  The
// numerators go to att_t in attn_pv_kernel's operand order [kv head][head quad][t][PV_G] (the PV_G heads of a timestep are one 16-byte load there).
constexpr int EXP_GRID_MAX = 16;      // 16 k positions in one pass; the 21 k-position test takes the second trip
static __global__ __launch_bounds__(256) void attn_exp_kernel(const AttnArgs a, int n_tiles_max) {
    __shared__ float red_s[4];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int h = blockIdx.y, n = a.dyn[1] + 1;
    if (blockIdx.x * EXP_ROW >= n) return;
    const int ntile = (n + ATT_TT - 1) / ATT_TT;
    float mx = -INFINITY;
    for (int i = t; i < ntile; i += 256) mx = fmaxf(mx, a.tmax[(size_t)h * n_tiles_max + i]);
    mx = wave_max(mx);
    if (lane == 0) red_s[wave] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red_s[0], red_s[1]), fmaxf(red_s[2], red_s[3]));
    const int kvmul = a.n_heads / a.n_kv_heads, kvh = h / kvmul, gq = h % kvmul;
    const float* sc = a.att + (size_t)h * a.att_stride;
    float* at = a.att_t + (size_t)(kvh * attn_pv_hq(kvmul) + gq / PV_G) * a.att_stride * PV_G + (gq % PV_G);
    for (int i0 = blockIdx.x * EXP_ROW; i0 < n; i0 += gridDim.x * EXP_ROW) {
#pragma unroll
        for (int u = 0; u < EXP_ROW / 256; ++u) {
            const int i = i0 + t + 256 * u;
            if (i < n) at[(size_t)i * PV_G] = (float)exp((double)(sc[i] - mx));
        }
    }
}

// The softmax denominator of one head: the strictly sequential f32 sum of its n numerators (FloatTensor.sum, J/tensor/standard/FloatTensor.java
// :211-219), evaluated exactly in parallel 4096 at a time, each chunk continued from the exact running value of the previous one
// (gl3_seqsum.h).  grid = heads, 256 threads.  The division e_t / sum happens where the weights are used (attn_pv_kernel's helpers).
static __global__ __launch_bounds__(256) void attn_sum_kernel(const AttnArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smx[];
    float* xf = reinterpret_cast<float*>(smx);                        // [SMX_CHUNK + 32]
    uint8_t* scratch = smx + (size_t)(SMX_CHUNK + 32) * 4;
    __shared__ float run_s;
    const int t = threadIdx.x;
    const int h = blockIdx.x, n = a.dyn[1] + 1;
    const int kvmul = a.n_heads / a.n_kv_heads, kvh = h / kvmul, gq = h % kvmul;
    const float* at = a.att_t + (size_t)(kvh * attn_pv_hq(kvmul) + gq / PV_G) * a.att_stride * PV_G + (gq % PV_G);
    if (t == 0) run_s = 0.f;
    if (t < 32) xf[SMX_CHUNK + t] = 0.f;                              // zero padding behind the chunk (exact_seqsum_lds reads past the end)
    // the chunk's numerators travel global -> registers -> LDS; the NEXT chunk's loads are in flight while this one is summed
    // (unconditional loads, clamped index: elements past the end are zeroed on their way to LDS)
    constexpr int PT = SMX_CHUNK / 256;
    float cur[PT];
#pragma unroll
    for (int u = 0; u < PT; ++u) cur[u] = at[(size_t)min(t + 256 * u, n - 1) * PV_G];
    for (int base = 0; base < n; base += SMX_CHUNK) {
        const int len = min(SMX_CHUNK, n - base);
#pragma unroll
        for (int u = 0; u < PT; ++u) xf[t + 256 * u] = t + 256 * u < len ? cur[u] : 0.f;
        const int nb = base + SMX_CHUNK;
#pragma unroll
        for (int u = 0; u < PT; ++u) cur[u] = at[(size_t)min(nb + t + 256 * u, n - 1) * PV_G];
        lds_barrier();
        float run = run_s;
        const int n4 = len & ~3;
        if (n4 >= 1024) {
            LdsBarrier bb;
            run = exact_seqsum_lds<false>(xf, n4, scratch, t, bb, run);
            if (n4 < len && t < 64) run = naive_sumsq_lds<false>(xf, n4, len, run);      // at most 3 trailing elements
        } else if (t < 64) {
            run = seq_sum_lds_ring(xf, len, run);
        }
        lds_barrier();
        if (t == 0) run_s = run;
        lds_barrier();
    }
    if (t == 0) a.sums[h] = run_s;
}

// attn_pv_kernel: grid = kv heads x head quads x (head_size / 16) slabs, block = 64 x (1 + PV_HELPERS).
//   chain (wavefront 0): lane (g, col) = (lane >> 4, lane & 15) owns xb[head quad * 4 + g][slab * 16 + col]; per 128-timestep tile
//       32 ds_read_b128 (four consecutive timesteps of its chain) + 128 dependent v_add_f32.
//   helpers: every helper wavefront prepares 16 rows of every tile.  Loads are shaped for the memory pipe, not for the chain: lane
//       (r, c4) = (lane >> 2, lane & 3) takes v[row r][4 c4 .. 4 c4 + 3] and the PV_G numerators of that row with ONE 16-byte load
//       each — 16 wave-level loads per tile instead of 192 (the first version of this kernel loaded per (chain lane, timestep): the
//       texture addresser needs ~16 cycles per 64-lane instruction and the whole workgroup waited for it: 12.9 cycles per
//       timestep, profiles/r05_tg_depth.md) — divides by the heads' denominators, multiplies out (16 products, each one rounding) and
//       scatters them with 4-byte LDS stores into the chain's order.
static __global__ __launch_bounds__(64 * PV_WAVES) void attn_pv_kernel(const AttnArgs a) {
    extern __shared__ __attribute__((aligned(16))) float pbuf[];              // [2][PVT / 4][PV_QP]
    const int hs = a.hs, kvmul = a.n_heads / a.n_kv_heads, hq_n = attn_pv_hq(kvmul);
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int kvh = blockIdx.x % a.n_kv_heads, rest = blockIdx.x / a.n_kv_heads, hq = rest % hq_n, slab = rest / hq_n;
    const int n = a.dyn[1] + 1;
    const int ntiles = (n + PVT - 1) / PVT;
    constexpr int QPT = PVT / 4;                                             // timestep quads per tile
    if (wave == 0) {
        // ------------------------------------------------------------------ chain: acc = p_t + acc, t ascending
        __builtin_amdgcn_s_setprio(3);
        // chain lane -> (head, column): the helpers' layout below puts product (g, 4 c4 + jj) into chain lane 16 g + 8 (jj >> 1) + 2 c4 + (jj & 1)
        const int g = hq * PV_G + (lane >> 4), col = 4 * ((lane >> 1) & 3) + 2 * ((lane >> 3) & 1) + (lane & 1);
        float acc = 0.f;
        PV_T(long long tw_ = 0; const long long ts_ = clock64();)
        for (int k = 0; k < ntiles; ++k) {
            PV_T(const long long b0_ = clock64();)
            lds_barrier();                                                 // tile k is complete (and tile k - 1 may be overwritten)
            PV_T(tw_ += clock64() - b0_;)
            const float* pb = pbuf + (size_t)(k & 1) * QPT * PV_QP + 4 * lane;
            // eight quads per group, the next group's reads in flight under this group's 32 dependent adds
            float4 ra[8], rb[8], rc[8];
#define PV_RD8(R_, Q0_) do { _Pragma("unroll") for (int u = 0; u < 8; ++u) R_[u] = *reinterpret_cast<const float4*>(pb + ((Q0_) + u) * PV_QP); } while (0)
            // sixteen adds as ONE asm statement: the compiler then waits once for the four quads (s_waitcnt lgkmcnt(n) in front of the
            // statement) instead of once per quad — every s_waitcnt is an issue slot of the wavefront the whole workgroup waits for
#define PV_ADD4Q(A_, B_, C_, D_) asm volatile( \
                "v_add_f32 %0, %1, %0\n\tv_add_f32 %0, %2, %0\n\tv_add_f32 %0, %3, %0\n\tv_add_f32 %0, %4, %0\n\t" \
                "v_add_f32 %0, %5, %0\n\tv_add_f32 %0, %6, %0\n\tv_add_f32 %0, %7, %0\n\tv_add_f32 %0, %8, %0\n\t" \
                "v_add_f32 %0, %9, %0\n\tv_add_f32 %0, %10, %0\n\tv_add_f32 %0, %11, %0\n\tv_add_f32 %0, %12, %0\n\t" \
                "v_add_f32 %0, %13, %0\n\tv_add_f32 %0, %14, %0\n\tv_add_f32 %0, %15, %0\n\tv_add_f32 %0, %16, %0" \
                : "+v"(acc) : "v"((A_).x), "v"((A_).y), "v"((A_).z), "v"((A_).w), "v"((B_).x), "v"((B_).y), "v"((B_).z), "v"((B_).w), \
                              "v"((C_).x), "v"((C_).y), "v"((C_).z), "v"((C_).w), "v"((D_).x), "v"((D_).y), "v"((D_).z), "v"((D_).w))
#define PV_ADD8(R_) do { PV_ADD4Q(R_[0], R_[1], R_[2], R_[3]); PV_ADD4Q(R_[4], R_[5], R_[6], R_[7]); } while (0)
            // three groups of eight quads: two groups (the helpers' stores share the LDS with these reads) are in flight under a group's adds
            PV_RD8(ra, 0);
            PV_RD8(rb, 8);
            PV_RD8(rc, 16);
            __builtin_amdgcn_sched_barrier(0);
            PV_ADD8(ra);
            __builtin_amdgcn_sched_barrier(0);
            PV_RD8(ra, 24);
            __builtin_amdgcn_sched_barrier(0);
            PV_ADD8(rb);
            __builtin_amdgcn_sched_barrier(0);
            PV_ADD8(rc);
            __builtin_amdgcn_sched_barrier(0);
            PV_ADD8(ra);
#undef PV_ADD4Q
#undef PV_ADD8
#undef PV_RD8
        }
        if (g < kvmul) {
            float* o = a.xb + (size_t)(kvh * kvmul + g) * hs + slab * PV_COLS16 + col;
            *o = acc;
            if (a.tp) tp_push_store(a.tp->p, o, acc);
        }
        if (a.tp) tp_publish(a.tp->p, gridDim.x);
        PV_T(if (blockIdx.x == 0 && lane == 0) { gl3_mv_stamp[0] = tw_; gl3_mv_stamp[1] = clock64() - ts_; })
        return;
    }
    // ---------------------------------------------------------------------- helpers
    // Helper wavefront hw prepares the 16 rows hw 16 .. hw 16 + 15 of EVERY tile: two 16-byte loads (v, the PV_G numerators) requested
    // PV_RING tiles ahead into a ring of named registers, then per tile 4 divisions (a = e / sum: FloatTensor.divideInPlace of the
    // softmax), 16 products and 16 scattered LDS stores — short enough to hide behind the chain's tile.
    // Wavefronts are dealt to the four SIMDs of a CU in turn: 4 and 8 would share the chain's SIMD and its issue slots — they retire
    // (a finished wavefront no longer counts at s_barrier), so the chain has its SIMD to itself.
    if (wave == 4 || wave == 8) return;
    const int hw = wave - 1 - (wave > 4) - (wave > 8);
    static_assert(PV_HELPERS * 16 == PVT && PV_WAVES == PV_HELPERS + 3, "one 16-row group per helper wavefront and tile");
    const int r = lane >> 2, c4 = lane & 3;
    // wave-uniform bases (scalar registers) + one 32-bit lane offset per stream
    const float* vbase = a.vcache + (size_t)kvh * hs + slab * PV_COLS16 + (size_t)(hw * 16) * a.kv_dim;
    const float* abase = a.att_t + ((size_t)(kvh * hq_n + hq) * a.att_stride + hw * 16) * PV_G;
    const unsigned vlane = (unsigned)r * a.kv_dim + 4 * c4;
    // LDS: row (in tile) rt = hw 16 + r -> quad rt >> 2, step rt & 3; product (g, 4 c4 + jj) -> chain lane 16 g + 8 (jj >> 1) + 2 c4 + (jj & 1),
    // i.e. float 4 (chain lane) + step of the quad's row.  One store instruction (fixed g, jj) then writes bank
    // 4 (r >> 2) + (r & 3) + 8 c4 (+ const) mod 32: the 32 lanes of a half wavefront (8 rows x 4 column quads) hit 32 different banks.
    // (With chain lane 16 g + 4 c4 + jj the column quads 0 / 2 and 1 / 3 shared their banks — LDS has 32, not 64: SQ_LDS_BANK_CONFLICT was
    // as large as SQ_ACTIVE_INST_LDS for this kernel, profiles/r05_tg_depth.md.)
    float* pst = pbuf + (size_t)((hw * 16 + r) >> 2) * PV_QP + (r & 3) + 8 * c4;
    // Lane (r, c4) divides ONE numerator per tile — head c4 of row r: the 16 rows x PV_G heads of the wavefront are exactly its 64 lanes —
    // and reads the other three heads' weights of its row from its quad neighbours as DPP operands of the multiplies (quad = the four
    // lanes of a row).  (Four divisions per lane, each repeated by the row's four lanes, were a third of the helpers' instructions; with
    // three helper wavefronts on a SIMD their instruction count, not the chain, set the tile time: profiles/r05_tg_depth.md.)
    const float smq = a.sums[kvh * kvmul + min(hq * PV_G + c4, kvmul - 1)];   // denominator of head c4 (heads past kvMul: clamped, unused)
    float4 v0, v1, v2, v3;                                                   // the ring (named: an array carried across the barriers would live in scratch)
    float e0, e1, e2, e3;
    // The last tile's loads run up to PVT - 1 rows past position n - 1: vcache and att_t are allocated with that much slack, and what
    // they return is masked in PVH_STORE.  Loads are unconditional (tile index clamped): a load under a branch is waited for at its end.
#define PVH_LOAD(S_, K_) do { \
        const int kc_ = min((K_), ntiles - 1); \
        v##S_ = *reinterpret_cast<const float4*>(vbase + (size_t)kc_ * PVT * a.kv_dim + vlane); \
        e##S_ = (abase + (size_t)kc_ * PVT * PV_G)[(unsigned)lane]; \
    } while (0)
#define PVH_QUAD(X_, G_) __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, X_), 0x55 * (G_), 0xf, 0xf, false))      /* lane G_ of my quad */
#define PVH_STORE_G(P_, A_, V_, G_) do { const float ag_ = PVH_QUAD(A_, G_); \
        (P_)[64 * (G_) + 0] = ag_ * (V_).x; (P_)[64 * (G_) + 4] = ag_ * (V_).y; (P_)[64 * (G_) + 32] = ag_ * (V_).z; (P_)[64 * (G_) + 36] = ag_ * (V_).w; } while (0)
#define PVH_STORE(S_, K_) do { \
        float* pb_ = pst + (size_t)((K_) & 1) * QPT * PV_QP; \
        float4 v_ = v##S_; \
        float a_ = e##S_ / smq;                             /* softmaxInPlace's divideInPlace(sum) */ \
        if ((K_) == ntiles - 1) {                           /* +0 pads the timesteps past the end: acc + 0 = acc (acc is never -0) */ \
            const bool in_ = (K_) * PVT + hw * 16 + r < n; \
            a_ = in_ ? a_ : 0.f; \
            v_.x = in_ ? v_.x : 0.f; v_.y = in_ ? v_.y : 0.f; v_.z = in_ ? v_.z : 0.f; v_.w = in_ ? v_.w : 0.f; \
        } \
        PVH_STORE_G(pb_, a_, v_, 0); PVH_STORE_G(pb_, a_, v_, 1); PVH_STORE_G(pb_, a_, v_, 2); PVH_STORE_G(pb_, a_, v_, 3); \
    } while (0)
#ifdef PV_NO_STORE                 /* probe build: chain speed without the helpers' LDS traffic (wrong results) */
#define PVH_STORE_X(S_, K_) do {} while (0)
#else
#define PVH_STORE_X(S_, K_) PVH_STORE(S_, K_)
#endif
#define PVH_STEP(S_, K_) do { \
        if ((K_) < ntiles) {                                /* wave-uniform; the chain runs exactly ntiles barriers */ \
            PV_T(const long long s0_ = clock64();) \
            PVH_STORE_X(S_, (K_));                          /* buffer K & 1: the chain finished tile K - 2 before the last barrier */ \
            PV_T(const long long b0_ = clock64(); tst_ += b0_ - s0_;) \
            lds_barrier(); \
            PV_T(tw_ += clock64() - b0_;) \
        } \
        PVH_LOAD(S_, (K_) + PV_RING);                       /* behind the barrier, PV_RING tiles ahead */ \
    } while (0)
    PV_T(long long tw_ = 0, tst_ = 0; const long long ts_ = clock64();)
    PVH_LOAD(0, 0); PVH_LOAD(1, 1); PVH_LOAD(2, 2); PVH_LOAD(3, 3);
    for (int k = 0; k < ntiles; k += PV_RING) {
        PVH_STEP(0, k); PVH_STEP(1, k + 1); PVH_STEP(2, k + 2); PVH_STEP(3, k + 3);
    }
    PV_T(if (blockIdx.x == 0 && hw == 0 && lane == 0) { gl3_mv_stamp[2] = tw_; gl3_mv_stamp[3] = tst_; gl3_mv_stamp[4] = clock64() - ts_; })
#undef PVH_LOAD
#undef PVH_STORE
#undef PVH_STORE_G
#undef PVH_STORE_X
#undef PVH_QUAD
#undef PVH_STEP
}

// ---------------------------------------------------------------------------------------------------
// Greedy sampling on the device: first index of the maximum (strict >), FloatTensor.argmax
// J/tensor/standard/FloatTensor.java:138-151 — NOT the strided-scan tie-break of the reference's
// argmaxLogits (TransformerComputeKernels.java:25-59).
// One launch, AMX_WGS workgroups: each scans a contiguous slice (float4 loads), publishes its (value, index) pair and takes a
// ticket; the LAST workgroup to arrive (no spinning: whoever draws the final ticket) folds the pairs in slice order and resets the
// ticket for the next token.  ws = int[2 + 2 * AMX_WGS]: [0] result, [1] ticket (zero at allocation), then the pairs.
// (One 1024-thread workgroup took 45 us for the 128256 logits of Llama-3.)
constexpr int AMX_WGS = 64;
__device__ __forceinline__ void amx_take(float& best, int& idx, float f, int i) {
    if (f > best || (f == best && i < idx)) { best = f; idx = i; }
}
static __global__ __launch_bounds__(256) void argmax_kernel(const float* __restrict__ v, int n, int* __restrict__ ws) {
    __shared__ float bv[4];
    __shared__ int bi[4];
    __shared__ int last;
    const int t = threadIdx.x, b = blockIdx.x;
    const int per = (((n + AMX_WGS - 1) / AMX_WGS) + 3) & ~3;          // slice length, multiple of 4 (n % 4 == 0 is not required)
    const int lo = b * per, hi = min(n, lo + per);
    float best = -INFINITY;
    int idx = 0x7FFFFFFF;
    const bool vec = (reinterpret_cast<uintptr_t>(v) & 15) == 0;
    if (vec) {
        for (int i = lo + 4 * t; i < hi; i += 1024) {
            if (i + 4 <= hi) {
                const float4 f = *reinterpret_cast<const float4*>(v + i);
                amx_take(best, idx, f.x, i); amx_take(best, idx, f.y, i + 1); amx_take(best, idx, f.z, i + 2); amx_take(best, idx, f.w, i + 3);
            } else {
                for (int j = i; j < hi; ++j) amx_take(best, idx, v[j], j);
            }
        }
    } else {
        for (int i = lo + t; i < hi; i += 256) amx_take(best, idx, v[i], i);
    }
    for (int m = 32; m >= 1; m >>= 1) {
        const float ob = __shfl_xor(best, m, 64);
        const int oi = __shfl_xor(idx, m, 64);
        amx_take(best, idx, ob, oi);
    }
    if ((t & 63) == 0) { bv[t >> 6] = best; bi[t >> 6] = idx; }
    __syncthreads();
    if (t == 0) {
        for (int w = 1; w < 4; ++w) amx_take(best, idx, bv[w], bi[w]);
        __hip_atomic_store(ws + 2 + 2 * b, __builtin_bit_cast(int, best), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(ws + 3 + 2 * b, idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int ticket = __hip_atomic_fetch_add(ws + 1, 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        last = ticket == AMX_WGS - 1;
    }
    __syncthreads();
    if (!last) return;
    if (t < 64) {                                   // AMX_WGS = 64 pairs: one per lane
        best = __builtin_bit_cast(float, __hip_atomic_load(ws + 2 + 2 * t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        idx = __hip_atomic_load(ws + 3 + 2 * t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        for (int m = 32; m >= 1; m >>= 1) {
            const float ob = __shfl_xor(best, m, 64);
            const int oi = __shfl_xor(idx, m, 64);
            amx_take(best, idx, ob, oi);
        }
        if (t == 0) {
            ws[0] = idx == 0x7FFFFFFF ? 0 : idx;
            __hip_atomic_store(ws + 1, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// One-time layout transform at upload: GGUF Q8_0 blocks (34 B: f16 d + 32 x int8, GGMLType.java:13) of rows
// [r0, r0+rows) and blocks [b0, b0+nb) of a [*, nb_full*32] matrix -> Q8T tiles.  One thread per destination
// (row, block) slot incl. zero padding (rows -> x16, blocks -> x4).
static __global__ void repack_q8t_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, int rows, int nb, int ng,
                                  long r0, int b0, int nb_full) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int rows16 = (rows + 15) & ~15, nb4 = ng * 4;
    if (idx >= (long)rows16 * nb4) return;
    const int row = (int)(idx / nb4), pb = (int)(idx % nb4);
    uint8_t* tile = dst + ((size_t)(row >> 4) * ng + (pb >> 2)) * TILE_BYTES;
    const int l = (row & 15) + 16 * (pb & 3);
    uint16_t h[17];
    if (row < rows && pb < nb) {
        const uint16_t* s = reinterpret_cast<const uint16_t*>(src + ((size_t)(r0 + row) * nb_full + b0 + pb) * 34);
#pragma unroll
        for (int i = 0; i < 17; ++i) h[i] = s[i];
    } else {
#pragma unroll
        for (int i = 0; i < 17; ++i) h[i] = 0;
    }
    *reinterpret_cast<uint16_t*>(tile + 2 * l) = h[0];
    uint16_t* lo = reinterpret_cast<uint16_t*>(tile + 128 + 16 * l);
    uint16_t* hi = reinterpret_cast<uint16_t*>(tile + 1152 + 16 * l);
#pragma unroll
    for (int i = 0; i < 8; ++i) { lo[i] = h[1 + i]; hi[i] = h[9 + i]; }
}

}  // namespace gl3
