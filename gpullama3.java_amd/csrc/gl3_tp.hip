// gl3_tp.hip — tensor parallelism over the 8 GPUs of one node: the per-rank arena of gathered buffers, the
// peer-write all-gather over xGMI, its set-up (IPC handles between processes, plain pointers inside one process) and the RCCL
// fall-back.
//
// The reference has no multi-GPU path (docs/TORNADOVM_TRANSFORMER_OPTIMIZATIONS.md:70 lists it as open), so this is the
// north star's extension.  Every matrix is split by OUTPUT rows (heads / hidden units / dim rows / vocab rows): each dot
// product stays whole and in the reference's order on one rank, so tensor-parallel results are bit-identical to the
// single-GPU / CPU result; the activations are re-assembled by in-place all-gathers (buf = [tp][count_per_rank], rank r owns
// chunk r): xb, x, hb, x per layer (+ the logits), the same points in batched prefill and static-batched decode.
//
// Transport (GL3_TP_P2P).  xGMI is point-to-point (7 links per GPU), and at decode the gathered slices are 2-7 KB per rank:
// the cost is latency, not bandwidth.  RCCL's all-gather costs 10-20 us per call at this size (protocol hand-shakes, one
// launch per collective, channel set-up); here one small kernel per gather
//   1. PUSHES this rank's slice straight into every peer's copy of the buffer (16-byte stores through the peer mapping,
//      targets rotated so the 7 links are used at once),
//   2. makes the stores visible system-wide (__threadfence_system) and bumps a per-source flag in every peer's arena
//      (one 4-byte system-scope store per peer: "rank me has completed its k-th gather"),
//   3. polls its OWN arena's flags until all peers have completed their k-th gather, then ends — the consumer is the next
//      kernel on the stream, so the kernel boundary orders the peers' data before its loads.
// k counts gathers since plan creation in device memory (replay-safe: nothing about the step is baked into a captured
// graph).  No acknowledgement is needed: a rank can run at most one gather ahead of a peer's push, and consecutive gathers
// never target the same buffer (xb, x, hb, x, ...), so a slice is never overwritten while a slower rank still reads it.
// The kernel is an ordinary launch: it is captured into the decode hipGraph like any other node.
// Expected cost: ~1.5 us boundary + ~3-5 us (store round trip over xGMI + flag propagation), 4 per layer + 1 per token.
// No hardware curve exists until the driver's 8-GPU run (SCALE_rNN.json); in CI the SAME kernel runs between ranks that are
// host threads of one process on one GPU (gl3_local_group) and between processes sharing one GPU over IPC handles.
//
// Folded gathers (decode step of the Q8_0 int8 path; gl3_api.hip tp_fold_setup, device side TpRec in gl3_decode_kernels.h).
// The three per-layer hand-overs of a decode step need no launch of their own on the producing side: the attention kernels
// (xb), the gate/up matvec (hb) and the down matvec (x) store every result element into the peers' arenas as they store it
// into their own, and the launch's last wavefront publishes "gather k of buffer b from rank me" into a per-buffer flag word
// of every peer (header bytes 512..: flags[3][16], ticket[3], step).  k = (step - 1) * n_layers + layer + 1 with step counted
// in device memory by the embedding kernel, so a captured graph replays unchanged.  The consumer side is either a
// one-wavefront wait launch in front of wo / down / the next qkv (default: measured +5 % decode tokens/s over the gather
// kernels between two processes on ONE GPU, 994 vs 943 tok/s Llama-3.2-1B) or the consumer's own prologue (GL3_TP_FOLD=2:
// the aux wavefronts of matvec_q8t_kernel<.., TPF> poll while the producer wavefronts already stream weights; nothing is
// launched between producer and consumer).  A polling consumer holds its compute units: between ranks that share one GPU it
// can keep a peer's producer from ever being scheduled (four in-process ranks, 2048-row wo: 384 polling workgroups on 256
// CUs, the fourth rank's attention kernel — whole-SIMD register budget — never starts), so mode 2 is for one rank per GPU
// and CI runs it with capped grids (GL3_WGS=16); there it is 11 % slower than the wait launches (843 tok/s) for the same reason.
// Write-after-read safety (no acknowledgements): rank A pushes buffer b of layer l only after it has consumed a flag that the
// slowest peer B publishes AFTER B's last read of its previous copy of b —
//   xb(l) is pushed by A's attention(l), behind A's wait for x(l-1) = end of B's down(l-1), which follows B's wo(l-1), the reader;
//   hb(l) by A's gate/up(l), behind the same wait; B's reader of hb was down(l-1) itself;
//   x(l)  by A's down(l), behind A's wait for hb(l) = end of B's gate/up(l), the last reader of x in B's layer l;
//   across steps the embedding kernel (mode 2) / the wait launch behind the last down (mode 1) waits for x(L) of the previous
//   step before x is overwritten, and the logits gather (a gather kernel in every mode) joins the ranks once per sampled token.
// The batched paths (prefill, static-batched decode) and the F16 / Q4_0 plans keep the gather kernels.
#include <cstring>
#include <tuple>

#include "gl3_ctx.h"

static size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

// ------------------------------------------------------------------------------------------------ arena pool
// An UNCACHED allocation is never handed back to the HIP allocator while the process lives: freed arenas wait here, keyed by
// (device, size), for the next plan of the same shape.  Root cause of round 4's "stale activation rows" (profiles/r05_tp_flake.md):
// with hipFree, the pages of a freed uncached arena were recycled by later hipMalloc calls of the same process under the CACHED
// policy (and freed cached buffers came back as the next arena); kernels of the next plans then computed on stale lines — in-process
// test ranks failed 8-9 runs of 10 from the second plan of a process on, and 0 of 20 with the arena never freed, with a cached or
// with a fine-grained arena; the gather's own checksums (GL3_TP_DEBUG) were clean in every failing run, i.e. the transport
// protocol was never at fault.  One process per GPU with one plan per process (production) never recycles; a server that
// re-creates tensor-parallel plans does, so the pool is always on.
namespace {
struct ArenaPool {
    std::mutex mu;
    std::vector<std::tuple<int, size_t, uint8_t*>> free_list;      // (device, capacity in bytes, base)
};
ArenaPool& arena_pool() { static ArenaPool* p = new ArenaPool(); return *p; }      // never destroyed: outlives every plan
// Best fit: the smallest pooled arena of the device that holds `bytes` (r5 matched the exact size only, so a server that re-created plans with a
// different ctx / max_batch / vocab kept one dead arena per shape for ever — advisor finding).  With >= matching the shapes converge: the pool never
// holds more arenas than were alive at the same time.  `cap` returns the arena's real capacity, which is what goes back into the pool.
uint8_t* arena_pool_take(int device, size_t bytes, size_t* cap) {
    ArenaPool& P = arena_pool();
    std::lock_guard<std::mutex> lk(P.mu);
    long best = -1;
    for (size_t i = 0; i < P.free_list.size(); ++i)
        if (std::get<0>(P.free_list[i]) == device && std::get<1>(P.free_list[i]) >= bytes &&
            (best < 0 || std::get<1>(P.free_list[i]) < std::get<1>(P.free_list[(size_t)best]))) best = (long)i;
    if (best < 0) return nullptr;
    uint8_t* b = std::get<2>(P.free_list[(size_t)best]);
    *cap = std::get<1>(P.free_list[(size_t)best]);
    P.free_list.erase(P.free_list.begin() + best);
    return b;
}
void arena_pool_give(int device, size_t cap, uint8_t* base) {
    ArenaPool& P = arena_pool();
    std::lock_guard<std::mutex> lk(P.mu);
    P.free_list.emplace_back(device, cap, base);
}
}  // namespace

// Pool inspection / release for long-lived hosts (include/gpullama3_hip.h).  gl3_tp_pool_trim hands the pooled arenas of a device (-1: all) back to
// the allocator — the one call that can re-create the recycling fault described above, so it is for a quiescent process (no plan alive on that
// device that was created after an arena was freed, nothing else allocating): servers call it between model reloads, tests never.
extern "C" GL3_API int32_t gl3_tp_pool_stats(int32_t device, uint64_t* arenas, uint64_t* bytes) {
    ArenaPool& P = arena_pool();
    std::lock_guard<std::mutex> lk(P.mu);
    uint64_t n = 0, b = 0;
    for (auto& e : P.free_list)
        if (device < 0 || std::get<0>(e) == device) { ++n; b += std::get<1>(e); }
    if (arenas) *arenas = n;
    if (bytes) *bytes = b;
    return GL3_OK;
}
extern "C" GL3_API int32_t gl3_tp_pool_trim(int32_t device) {
    ArenaPool& P = arena_pool();
    std::lock_guard<std::mutex> lk(P.mu);
    int cur = 0;
    if (hipGetDevice(&cur) != hipSuccess) return GL3_E_HIP;
    for (size_t i = 0; i < P.free_list.size();) {
        if (device >= 0 && std::get<0>(P.free_list[i]) != device) { ++i; continue; }
        if (hipSetDevice(std::get<0>(P.free_list[i])) != hipSuccess || hipFree(std::get<2>(P.free_list[i])) != hipSuccess) { hipSetDevice(cur); return GL3_E_HIP; }
        P.free_list.erase(P.free_list.begin() + (long)i);
    }
    hipSetDevice(cur);
    return GL3_OK;
}

// ------------------------------------------------------------------------------------------------ arena
int32_t gl3_tp_arena_alloc(gl3_ctx* ctx) {
    const gl3_model_desc& d = ctx->d;
    gl3_tp_arena& A = ctx->arena;
    size_t o = GL3_ARENA_HDR;
    auto put = [&](int which, size_t floats) { A.off[which] = o; o += align256(floats * 4); };
    put(GB_X, d.dim); put(GB_XB, (size_t)d.n_heads * d.head_size); put(GB_HB, d.hidden); put(GB_LOGITS, d.vocab);
    const bool int8_path = d.weight_type == GL3_TYPE_Q8_0 && !(d.flags & GL3_FLAG_F32_ACTIVATION);
    if (d.max_batch > 1 && (int8_path || !(d.flags & GL3_FLAG_SCALAR_DOT))) {      // every type with a batched path (gl3_prefill.hip)
        const size_t M = d.max_batch;
        A.pf_logits_rows = d.max_batch < 64 ? d.max_batch : 64;
        put(GB_PF_X, M * d.dim); put(GB_PF_AO, M * d.n_heads * d.head_size); put(GB_PF_HB, M * d.hidden);
        put(GB_PF_LOGITS, (size_t)A.pf_logits_rows * d.vocab);
    }
    A.bytes = o;
    // Uncached device memory (what RCCL uses for its own peer buffers): a peer's writes arrive over xGMI behind the owner's
    // L2, so the owner must never hold a stale line.  Experiments: GL3_TP_ARENA=cached selects plain hipMalloc (kernel-boundary
    // invalidation only), =finegrained hipDeviceMallocFinegrained, =unpooled the uncached arena returned with hipFree (reproduces
    // the recycling fault described at the pool above).
    const char* mode = getenv("GL3_TP_ARENA");
    const int kind = (mode && !strcmp(mode, "cached")) ? 1 : (mode && !strcmp(mode, "finegrained")) ? 2 : (mode && !strcmp(mode, "unpooled")) ? 3 : 0;
    A.cap = A.bytes;
    A.base = kind == 0 ? arena_pool_take(d.device, A.bytes, &A.cap) : nullptr;
    A.pooled = kind == 0;
    A.kind = kind;
    if (!A.base) {
        hipError_t e = kind == 1 ? hipMalloc((void**)&A.base, A.bytes)
                       : kind == 2 ? hipExtMallocWithFlags((void**)&A.base, A.bytes, hipDeviceMallocFinegrained)
                                   : hipExtMallocWithFlags((void**)&A.base, A.bytes, hipDeviceMallocUncached);
        if (e != hipSuccess) { A.base = nullptr; ctx->err = std::string("tensor-parallel arena: ") + hipGetErrorString(e); return e == hipErrorOutOfMemory ? GL3_E_OOM : GL3_E_HIP; }
    }
    GL3_HIP(hipMemset(A.base, 0, A.bytes));
    GL3_HIP(hipHostMalloc((void**)&ctx->h_tp_err, sizeof(uint32_t)));
    *ctx->h_tp_err = 0;
    ctx->peer_base[d.tp_rank] = A.base;
    return GL3_OK;
}

void gl3_tp_arena_free(gl3_ctx* ctx) {
    for (int p = 0; p < GL3_MAX_TP; ++p)
        if (ctx->ipc_opened[p]) { hipIpcCloseMemHandle(ctx->ipc_opened[p]); ctx->ipc_opened[p] = nullptr; }
    if (ctx->arena.base) {
        if (ctx->arena.pooled) arena_pool_give(ctx->d.device, ctx->arena.cap, ctx->arena.base);
        else hipFree(ctx->arena.base);
    }
    ctx->arena.base = nullptr;
    if (ctx->h_tp_err) hipHostFree(ctx->h_tp_err);
    ctx->h_tp_err = nullptr;
}

float* gl3_gather_buf(gl3_ctx* c, int which) {
    if (which >= GB_PF_X) return gl3_prefill_buf(c, which);
    return which == GB_XB ? c->xb : which == GB_X ? c->x : which == GB_HB ? c->hb : c->logits;
}

// ------------------------------------------------------------------------------------------------ peer-write all-gather
struct TpGatherArgs {
    uint8_t* peer[GL3_MAX_TP];   // arena of every rank as mapped here; peer[me] = own arena
    size_t buf_off;              // byte offset of the gathered buffer inside every arena
    size_t n4;                   // float4s per rank slice
    int me, tp;
    unsigned spin_limit;         // polls before giving up
    uint32_t* err;               // host-pinned word: set to 1 on a timeout
};

typedef float v4f_tp __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(1024) void tp_gather_kernel(const TpGatherArgs a) {
    const int t = threadIdx.x;
    uint8_t* own = a.peer[a.me];
    uint32_t* seq = reinterpret_cast<uint32_t*>(own + GL3_ARENA_SEQ);
    uint32_t* arrive = reinterpret_cast<uint32_t*>(own + GL3_ARENA_ARRIVE);
    // ---- 1. push my slice into every peer's buffer (peer order rotated by rank: all links busy at once)
    const v4f_tp* src = reinterpret_cast<const v4f_tp*>(own + a.buf_off) + (size_t)a.me * a.n4;
    for (size_t i = (size_t)blockIdx.x * 1024 + t; i < a.n4; i += (size_t)gridDim.x * 1024) {
        const v4f_tp v = src[i];
        for (int j = 1; j < a.tp; ++j) {
            const int p = (a.me + j) % a.tp;
            reinterpret_cast<v4f_tp*>(a.peer[p] + a.buf_off)[(size_t)a.me * a.n4 + i] = v;
        }
    }
    __threadfence_system();                  // my stores have reached the peers' memory
    __syncthreads();
    if (t != 0) return;
    // ---- 2. the last workgroup to finish publishes the flag, 3. and waits for the peers
    const unsigned ticket = __hip_atomic_fetch_add(arrive, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    if (ticket != gridDim.x - 1) return;
    __hip_atomic_store(arrive, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const uint32_t k = __hip_atomic_load(seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1;
    __threadfence_system();
    for (int j = 1; j < a.tp; ++j) {
        const int p = (a.me + j) % a.tp;
        __hip_atomic_store(reinterpret_cast<uint32_t*>(a.peer[p]) + a.me, k, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    const uint32_t* flags = reinterpret_cast<const uint32_t*>(own);
    bool ok = true;
    for (int j = 1; j < a.tp && ok; ++j) {
        const int p = (a.me + j) % a.tp;
        unsigned spins = 0;
        // signed distance: a peer may already be one gather ahead
        while ((int32_t)(__hip_atomic_load(flags + p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) - k) < 0) {
            __builtin_amdgcn_s_sleep(16);
            if (++spins > a.spin_limit) { ok = false; break; }
        }
    }
    if (!ok) __hip_atomic_store(a.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(seq, k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");      // system scope: drop anything cached before the peers' data landed
}

// GL3_TP_DEBUG=1 — diagnostic twin of tp_gather_kernel (ONE workgroup): every pushed slice travels with a position-weighted
// checksum (written into the peers' arena headers before the flag), and the receiver
//   phase 0: re-computes the checksum of every peer's slice right after its flag wait (a mismatch = the flag was visible before
//            the data, i.e. a visibility / ordering fault of the transport), and
//   phase 1: at the START of its next gather re-checks the previous gather's peer slices (nothing on this rank writes them in
//            between; a mismatch = a peer's later push landed while this rank could still be reading: a write-after-read fault).
// Mismatches are printed from the device.  Neither firing while results are wrong = the fault is not in the transport.
struct TpGatherDbg { size_t prev_off, prev_n4; };
constexpr size_t GL3_ARENA_DBG = 256;      // u32 sums[4][GL3_MAX_TP]: sums[k & 3][p] = checksum of rank p's slice of gather k

__device__ __forceinline__ unsigned tp_dbg_sum(const v4f_tp* p, size_t n4, int t, unsigned* sh) {
    unsigned h = 0;
    for (size_t i = t; i < n4; i += 1024) {
        const v4f_tp v = p[i];
        const unsigned w0 = __builtin_bit_cast(unsigned, v.x), w1 = __builtin_bit_cast(unsigned, v.y), w2 = __builtin_bit_cast(unsigned, v.z), w3 = __builtin_bit_cast(unsigned, v.w);
        const unsigned m = (unsigned)(4 * i) * 2654435761u;
        h += w0 * (m | 1u) + w1 * ((m + 0x9E3779B9u) | 1u) + w2 * ((m + 0x3C6EF372u) | 1u) + w3 * ((m + 0xDAA66D2Bu) | 1u);
    }
    __syncthreads();
    if (t == 0) *sh = 0;
    __syncthreads();
    atomicAdd(sh, h);
    __syncthreads();
    return *sh;
}

__global__ __launch_bounds__(1024) void tp_gather_dbg_kernel(const TpGatherArgs a, const TpGatherDbg g) {
    __shared__ unsigned sh, kk;
    const int t = threadIdx.x;
    uint8_t* own = a.peer[a.me];
    uint32_t* seq = reinterpret_cast<uint32_t*>(own + GL3_ARENA_SEQ);
    unsigned* sums = reinterpret_cast<unsigned*>(own + GL3_ARENA_DBG);
    if (t == 0) kk = __hip_atomic_load(seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1;
    __syncthreads();
    const uint32_t k = kk;
    // ---- phase 1: the previous gather's peer slices must still be what their owners pushed
    if (g.prev_n4) {
        for (int j = 1; j < a.tp; ++j) {
            const int p = (a.me + j) % a.tp;
            const unsigned got = tp_dbg_sum(reinterpret_cast<const v4f_tp*>(own + g.prev_off) + (size_t)p * g.prev_n4, g.prev_n4, t, &sh);
            const unsigned want = __hip_atomic_load(sums + ((k - 1) & 3) * GL3_MAX_TP + p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            if (t == 0 && got != want)
                printf("[gl3 tp dbg] rank %d gather %u: slice of rank %d from gather %u (arena offset %zu, %zu float4) CHANGED before the next gather: %08x != %08x (write-after-read)\n",
                       a.me, k, p, k - 1, g.prev_off, g.prev_n4, got, want);
        }
    }
    // ---- push (as tp_gather_kernel) + checksum of my slice into the peers' headers
    const v4f_tp* src = reinterpret_cast<const v4f_tp*>(own + a.buf_off) + (size_t)a.me * a.n4;
    for (size_t i = t; i < a.n4; i += 1024) {
        const v4f_tp v = src[i];
        for (int j = 1; j < a.tp; ++j) {
            const int p = (a.me + j) % a.tp;
            reinterpret_cast<v4f_tp*>(a.peer[p] + a.buf_off)[(size_t)a.me * a.n4 + i] = v;
        }
    }
    const unsigned mine = tp_dbg_sum(src, a.n4, t, &sh);
    __threadfence_system();
    __syncthreads();
    if (t == 0) {
        for (int j = 1; j < a.tp; ++j) {
            const int p = (a.me + j) % a.tp;
            __hip_atomic_store(reinterpret_cast<unsigned*>(a.peer[p] + GL3_ARENA_DBG) + (k & 3) * GL3_MAX_TP + a.me, mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        __threadfence_system();
        for (int j = 1; j < a.tp; ++j) {
            const int p = (a.me + j) % a.tp;
            __hip_atomic_store(reinterpret_cast<uint32_t*>(a.peer[p]) + a.me, k, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        const uint32_t* flags = reinterpret_cast<const uint32_t*>(own);
        bool ok = true;
        for (int j = 1; j < a.tp && ok; ++j) {
            const int p = (a.me + j) % a.tp;
            unsigned spins = 0;
            while ((int32_t)(__hip_atomic_load(flags + p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) - k) < 0) {
                __builtin_amdgcn_s_sleep(16);
                if (++spins > a.spin_limit) { ok = false; break; }
            }
        }
        if (!ok) __hip_atomic_store(a.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(seq, k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
    }
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
    // ---- phase 0: what the peers pushed is what is visible here now
    for (int j = 1; j < a.tp; ++j) {
        const int p = (a.me + j) % a.tp;
        const unsigned got = tp_dbg_sum(reinterpret_cast<const v4f_tp*>(own + a.buf_off) + (size_t)p * a.n4, a.n4, t, &sh);
        const unsigned want = __hip_atomic_load(sums + (k & 3) * GL3_MAX_TP + p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if (t == 0 && got != want)
            printf("[gl3 tp dbg] rank %d gather %u: slice of rank %d (arena offset %zu, %zu float4) NOT COMPLETE when its flag was seen: %08x != %08x (visibility)\n",
                   a.me, k, p, a.buf_off, a.n4, got, want);
    }
}

static int32_t all_gather_p2p(gl3_ctx* ctx, int which, size_t count_per_rank) {
    const gl3_model_desc& d = ctx->d;
    if (!ctx->arena.off[which]) GL3_FAIL(GL3_E_STATE, "gathered buffer is not part of the tensor-parallel arena");
    if (count_per_rank % 4) GL3_FAIL(GL3_E_UNSUPPORTED, "tensor-parallel slice must be a multiple of 4 floats");
    TpGatherArgs a{};
    for (int p = 0; p < d.tp_size; ++p) a.peer[p] = ctx->peer_base[p];
    a.buf_off = ctx->arena.off[which]; a.n4 = count_per_rank / 4; a.me = d.tp_rank; a.tp = d.tp_size;
    static const unsigned limit = getenv("GL3_TP_SPIN_LIMIT") ? (unsigned)atol(getenv("GL3_TP_SPIN_LIMIT")) : 20000000u;   // x ~0.5 us: ~10 s
    a.spin_limit = limit; a.err = ctx->h_tp_err;
    size_t wgs = (a.n4 * 16 + 65535) / 65536;           // 64 KB of slice per workgroup
    wgs = wgs < 1 ? 1 : wgs > 64 ? 64 : wgs;
    static const bool dbg = env_flag("GL3_TP_DEBUG", false);
    if (dbg) {
        TpGatherDbg g{ctx->tp_dbg_prev_off, ctx->tp_dbg_prev_n4};
        hipLaunchKernelGGL(tp_gather_dbg_kernel, dim3(1), dim3(1024), 0, ctx->stream, a, g);
        ctx->tp_dbg_prev_off = a.buf_off; ctx->tp_dbg_prev_n4 = a.n4;
        return GL3_OK;
    }
    hipLaunchKernelGGL(tp_gather_kernel, dim3((unsigned)wgs), dim3(1024), 0, ctx->stream, a);
    return GL3_OK;
}

int32_t gl3_all_gather(gl3_ctx* ctx, int which, size_t count_per_rank) {
    if (!ctx->use_rccl) return GL3_OK;
    if (ctx->transport == GL3_TP_P2P) return all_gather_p2p(ctx, which, count_per_rank);
    float* buf = gl3_gather_buf(ctx, which);
    GL3_NCCL(ncclAllGather(buf + (size_t)ctx->d.tp_rank * count_per_rank, buf, count_per_rank, ncclFloat, ctx->comm, ctx->stream));
    return GL3_OK;
}

int32_t gl3_tp_check(gl3_ctx* ctx) {
    ctx->tp_dbg_prev_n4 = 0;                 // GL3_TP_DEBUG: the next forward call's first gather has no predecessor to re-check
    if (ctx->h_tp_err && *ctx->h_tp_err) {
        *ctx->h_tp_err = 0;
        GL3_FAIL(GL3_E_RCCL, "tensor-parallel all-gather timed out waiting for a peer rank");
    }
    return GL3_OK;
}

// ------------------------------------------------------------------------------------------------ set-up (C-ABI)
extern "C" {

int32_t gl3_tp_unique_id(void* out, uint64_t bytes) {
    if (!out || bytes < sizeof(ncclUniqueId)) return GL3_E_ARG;
    ncclUniqueId id;
    if (ncclGetUniqueId(&id) != ncclSuccess) return GL3_E_RCCL;
    memset(out, 0, bytes);
    memcpy(out, &id, sizeof(id));
    return GL3_OK;
}

int32_t gl3_tp_init(gl3_ctx* ctx, const void* unique_id, uint64_t bytes) {
    if (!ctx) return GL3_E_ARG;
    if (!unique_id || bytes < sizeof(ncclUniqueId)) GL3_FAIL(GL3_E_ARG, "bad RCCL unique id");
    if (ctx->finalized) GL3_FAIL(GL3_E_STATE, "tp_init after finalize");
    GL3_HIP(hipSetDevice(ctx->d.device));
    ncclUniqueId id;
    memcpy(&id, unique_id, sizeof(id));
    GL3_NCCL(ncclCommInitRank(&ctx->comm, ctx->d.tp_size, id, ctx->d.tp_rank));
    ctx->transport = GL3_TP_RCCL;
    return GL3_OK;
}

int32_t gl3_tp_peer_access(int32_t device, const int32_t* peer_devices, int32_t n, int32_t* reachable) {
    if (!peer_devices || !reachable || n < 0) return GL3_E_ARG;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess) return GL3_E_HIP;
    if (device < 0 || device >= count) return GL3_E_ARG;
    *reachable = 0;
    for (int i = 0; i < n; ++i) {
        const int p = peer_devices[i];
        if (p < 0 || p >= count) return GL3_E_ARG;
        int can = 1;                                     // a device reaches itself (ranks sharing one GPU in tests)
        if (p != device && hipDeviceCanAccessPeer(&can, device, p) != hipSuccess) return GL3_E_HIP;
        *reachable += can ? 1 : 0;
    }
    return GL3_OK;
}

int32_t gl3_tp_p2p_handle(gl3_ctx* ctx, void* out, uint64_t bytes) {
    if (!ctx) return GL3_E_ARG;
    if (!out || bytes < sizeof(hipIpcMemHandle_t)) GL3_FAIL(GL3_E_ARG, "handle buffer shorter than 64 bytes");
    if (!ctx->arena.base) GL3_FAIL(GL3_E_STATE, "plan has no tensor-parallel arena (tp_size == 1)");
    GL3_HIP(hipSetDevice(ctx->d.device));
    hipIpcMemHandle_t h;
    GL3_HIP(hipIpcGetMemHandle(&h, ctx->arena.base));
    memset(out, 0, bytes);
    memcpy(out, &h, sizeof(h));
    return GL3_OK;
}

int32_t gl3_tp_p2p_attach(gl3_ctx* ctx, const void* handles, uint64_t bytes) {
    if (!ctx) return GL3_E_ARG;
    const int tp = ctx->d.tp_size;
    if (!handles || bytes < (uint64_t)tp * sizeof(hipIpcMemHandle_t)) GL3_FAIL(GL3_E_ARG, "need tp_size x 64 bytes of IPC handles in rank order");
    if (ctx->finalized) GL3_FAIL(GL3_E_STATE, "attach after finalize");
    if (!ctx->arena.base) GL3_FAIL(GL3_E_STATE, "plan has no tensor-parallel arena");
    if (tp > GL3_MAX_TP) GL3_FAIL(GL3_E_UNSUPPORTED, "tp_size above 16");
    GL3_HIP(hipSetDevice(ctx->d.device));
    for (int p = 0; p < tp; ++p) {
        if (p == ctx->d.tp_rank) continue;
        hipIpcMemHandle_t h;
        memcpy(&h, (const uint8_t*)handles + (size_t)p * sizeof(h), sizeof(h));
        void* ptr = nullptr;
        GL3_HIP(hipIpcOpenMemHandle(&ptr, h, hipIpcMemLazyEnablePeerAccess));
        ctx->ipc_opened[p] = ptr;
        ctx->peer_base[p] = (uint8_t*)ptr;
    }
    ctx->transport = GL3_TP_P2P;
    return GL3_OK;
}

int32_t gl3_local_group_create(int32_t n, gl3_local_group** out) {
    if (n < 1 || n > GL3_MAX_TP || !out) return GL3_E_ARG;
    gl3_local_group* g = new gl3_local_group();
    g->n = n; g->ranks.assign(n, nullptr);
    *out = g;
    return GL3_OK;
}

void gl3_local_group_destroy(gl3_local_group* g) { delete g; }

int32_t gl3_tp_attach_local(gl3_ctx* ctx, gl3_local_group* g) {
    if (!ctx || !g) return GL3_E_ARG;
    if (ctx->finalized) GL3_FAIL(GL3_E_STATE, "attach after finalize");
    if (g->n != ctx->d.tp_size) GL3_FAIL(GL3_E_ARG, "local group size differs from tp_size");
    if (!ctx->arena.base) GL3_FAIL(GL3_E_STATE, "plan has no tensor-parallel arena");
    std::lock_guard<std::mutex> lk(g->mu);
    g->ranks[ctx->d.tp_rank] = ctx;
    ctx->lgrp = g;
    ctx->use_rccl = true;
    ctx->transport = GL3_TP_P2P;
    return GL3_OK;
}

}  // extern "C"

// gl3_finalize: every rank of a local group has attached once all have passed this barrier; resolve the peers' arenas
int32_t gl3_tp_local_resolve(gl3_ctx* ctx) {
    gl3_local_group* g = ctx->lgrp;
    if (!g) return GL3_OK;
    g->barrier();
    for (int p = 0; p < g->n; ++p) {
        if (!g->ranks[p]) GL3_FAIL(GL3_E_STATE, "local group: a rank never attached");
        if (g->ranks[p]->arena.bytes != ctx->arena.bytes) GL3_FAIL(GL3_E_STATE, "local group: ranks were created with different shapes");
        ctx->peer_base[p] = g->ranks[p]->arena.base;
    }
    return GL3_OK;
}
