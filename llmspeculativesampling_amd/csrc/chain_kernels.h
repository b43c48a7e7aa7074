// One launch for the weight-streaming half of a decoder layer on <= 16 new rows (the verify step of the target):
//
//     O GEMM -> residual + norm -> gate/up GEMM (+ activation) -> down GEMM -> residual + norm -> QKV GEMM of the next layer
//
// (reference: the per-token forward of sampling/kvcache_model.py:206-214 -> modeling_llama.py:292-457 /
// modeling_opt.py:303-378; the six steps are six launches of the per-op path in engine.hip, forward_impl).
//
// The steps stay what they are - the same workgroup shapes, the same split-K plan, the same arithmetic in the same
// order (the per-op path's device functions are reused, so both routes are bit-identical; tests check it) - but they
// are PHASES of one grid: workgroups [blk0, blk0 + nblk) of the launch belong to phase p, and a workgroup of phase p
//   1. requests the first CH_PF k-steps of its weight tiles (they depend on nothing),
//   2. waits until phase p-1 has signalled (one word, polled by one lane),
//   3. reads its activations, multiplies, runs the epilogue, signals.
// Workgroups are dispatched in blockIdx order, so every workgroup a waiter depends on was dispatched before it and
// waits on nothing that comes later: no deadlock, whatever number of workgroups is resident (every spin is still
// bounded by a wall-clock limit that sets an error word instead of hanging).  What the chain buys is that the HBM
// pipe never drains at a step boundary: while the last workgroups of a phase finish and the 5-workgroup norm runs,
// the resident workgroups of the following GEMM already have up to 16 KiB of weights per wave in flight.
//
// Visibility across the 8 XCDs (private, non-coherent L2s): everything handed over inside the launch is stored
// write-through (sc1), every storing wave drains its stores (vmcnt(0)) before the workgroup's arrival is counted, and the
// counters are agent-scope atomics.  Buffers handed over inside the launch are written once per launch and never read
// before their phase's signal, so a consumer's first touch of a line is an L2 miss that is served from memory (the norm
// phases, which also read the residual stream in place, add an agent-scope acquire).  The two norm outputs of one launch
// therefore go to two different buffers (h / h2), as do the two split-K slab sets (part / spart).  An L2 write-back
// (release fence) per arriving workgroup instead of write-through stores made the verify step 2.5x SLOWER (12.8 ms).
#pragma once
#include "model_kernels.h"

enum { CH_GEMM = 0, CH_RN = 1 };
#define CH_MAX_PHASES 6
#define CH_SHARDS 32
#define CH_SHARD_STRIDE 32                                   // words: every arrival counter on its own 128-byte line
#define CH_CTR_WORDS ((CH_SHARDS + 1) * CH_SHARD_STRIDE)     // per signalling phase: shard counters + the engine's "done" word
#ifndef CH_PF
#define CH_PF 4                                             // k-steps of weights a wave requests ahead (1 KiB each)
#endif
#ifndef CH_XPF
#define CH_XPF 4                                             // k-steps of activations in flight per wave
#endif
#ifndef CH_WAVES
#define CH_WAVES 4                                           // waves per SIMD the register budget allows (4: <= 128 VGPRs)
#endif
#define CH_TIMEOUT_TICKS 2000000ll                           // wall_clock64 runs at 100 MHz: 20 ms

struct ChainPhase {
    int type, epi;                  // CH_GEMM (epi = EPI_PART / EPI_ACT_* / EPI_QKV_*) or CH_RN
    int blk0, nblk;                 // workgroups [blk0, blk0 + nblk) of the launch
    int wait_slot, wait_n;          // wait until the wait_n workgroups of the phase before have arrived at counter block wait_slot (-1: none)
    int sig_slot;                   // counter block this phase's workgroups arrive at (-1: none)
    // CH_GEMM: part[sb][m][n] (EPI_PART) or the fused epilogue's outputs
    const u32x4 *W;
    const void *X;
    float *part;
    int N, K, SB, ks_per_blk;
    void *out;
    const void *bias;
    int n_out, layer;
    // CH_RN: x <- rnd(x + rnd(sum_s slab[s] + bias)); out <- norm(x) in operand layout   (residual_norm_kernel)
    void *xres;
    const float *slab;
    int S;
    size_t stride_s;
    const void *nw, *nb;
    int mode;
};

template <typename H>
struct ChainArgs {
    int n_phases, M, hidden, rn_threads;
    float eps;
    const H *cos_t, *sin_t;
    int Hq, Hkv, D;
    float q_scale;
    unsigned *ctr;                  // [CH_MAX_PHASES][CH_CTR_WORDS], zeroed once per forward; counts grow with `epoch`
    unsigned *err;
    unsigned epoch;                 // 1 + index of this launch since the counters were zeroed
    unsigned flags;                 // debugging aids (SD_CHAIN_FLAGS): 1 = L2 write-back before every arrival, 2 = no acquire in
                                    // the norm phases; WRONG results, timing experiments only: 4 = no waits, 8 = no arrivals,
                                    // engine: 16 = no activation loads, 32 = no ring reads / MFMAs, 64 = no epilogues
    ChainPhase ph[CH_MAX_PHASES];
    RowTab tab;
};

// what gemm_epilogue_step reads (GemmEpiT's member names), with the row table by reference
template <typename H>
struct ChainEpi {
    H *out;
    const H *bias;
    int n_out;
    const H *cos_t, *sin_t;
    int Hq, Hkv, D, layer;
    float q_scale;
    const RowTab &tab;
};

// Arrivals are counted on min(CH_SHARDS, nblk) counters (workgroup b of the phase adds to counter b % n, without waiting
// for the add to return); a waiter's first wave reads all of them with one load (lane i: counter i) until each shows its
// share of the phase's workgroups times `epoch`.
// `late` (or NULL): an LDS word that is set when the wait ran into its limit; the caller poisons what it produces, so a
// launch that carried on with stale operands can never hand back plausible numbers (the two-phase launches of the default
// path, which nobody checks a status word after).
__device__ __forceinline__ void chain_wait(unsigned *ctr, int slot, int nblk, unsigned epoch, unsigned *err, int *late = nullptr) {
    if (slot < 0) return;
    if (late && threadIdx.x == 0) *late = 0;
    if (threadIdx.x < 64) {
        const int nsh = min(CH_SHARDS, nblk), lane = threadIdx.x;
        unsigned *mine = ctr + (size_t)slot * CH_CTR_WORDS + (lane < nsh ? lane : 0) * CH_SHARD_STRIDE;
        const unsigned want = lane < nsh ? (unsigned)(nblk / nsh + (lane < nblk % nsh ? 1 : 0)) * epoch : 0u;
        const long long t0 = wall_clock64();
        for (;;) {
            const unsigned got = __hip_atomic_load(mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (__all(got >= want)) break;
            __builtin_amdgcn_s_sleep(2);
            if (wall_clock64() - t0 > CH_TIMEOUT_TICKS) {
                if (lane == 0) {
                    __hip_atomic_fetch_or(err, 1u << slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (late) *late = 1;
                }
                break;
            }
        }
    }
    __syncthreads();
    asm volatile("" ::: "memory");
}

// Called by every thread of the workgroup.  The handed-over data was stored write-through (sc1) by whole waves: each of
// them drains its stores, then one lane counts the arrival.  `block_wide`: more than wave 0 stored.  `wbl2`: also write
// this XCD's L2 back (for plain stores; measured far too expensive per workgroup - kept as a debugging aid).
__device__ __forceinline__ void chain_signal(unsigned *ctr, int slot, int b, int nblk, unsigned epoch, bool block_wide, bool wbl2) {
    if (slot < 0) return;
    if (block_wide || threadIdx.x < 64) {
        if (wbl2) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    }
    if (block_wide) __syncthreads();
    if (threadIdx.x == 0) {
        const int nsh = min(CH_SHARDS, nblk);
        (void)__hip_atomic_fetch_add(ctr + (size_t)slot * CH_CTR_WORDS + (b % nsh) * CH_SHARD_STRIDE, 1u, __ATOMIC_RELAXED,
                                     __HIP_MEMORY_SCOPE_AGENT);
    }
}

// gemm_bf16_stream<1, ., EPI, 1>'s workgroup (one 16-column n-tile x one k-slab, 4 waves on a quarter of the slab each,
// k-steps accumulated in order), with the weight requests ahead of the wait and kept CH_PF deep afterwards.
template <int EPI, typename H>
__device__ __forceinline__ void chain_gemm(const ChainPhase &ph, const ChainArgs<H> &a, int b, f32x4 (*red)[1][64],
                                           int *late = nullptr) {
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int N = ph.N, KS = ph.K >> 5, NTG = N >> 4;
    const int sb = b / NTG, ntg = b - sb * NTG;
    const int kb0 = sb * ph.ks_per_blk, kb1 = min(KS, kb0 + ph.ks_per_blk);
    const int per = (kb1 - kb0 + 3) >> 2;
    const int ks0 = min(kb1, kb0 + wv * per), ks1 = min(kb1, ks0 + per);
    const int nk = ks1 - ks0;                                     // (wave-uniform: every bound below is a scalar branch)
    const u32x4 *wp = ph.W + ((size_t)ntg * KS + ks0) * 64 + lane;
    u32x4 w[CH_PF];
#pragma unroll
    for (int u = 0; u < CH_PF; ++u) w[u] = u < nk ? __builtin_nontemporal_load(wp + (size_t)u * 64) : u32x4{0u, 0u, 0u, 0u};

    if (!(a.flags & 4)) chain_wait(a.ctr, ph.wait_slot, ph.wait_n, a.epoch, a.err, late);
    const bool poisoned = late && ph.wait_slot >= 0 && !(a.flags & 4) && *late;      // (no wait: the word was never written)

    // operand layout (xoff): tile (0, ks) = 512 elements, lane 16 * quad + m holds X[m][32 ks + 8 quad .. + 8).  Lanes of
    // rows >= M read row 0's fragment (the same 16 bytes as their quad's first lane: no extra traffic, no branch around
    // the load); what they compute lands in output columns m >= M, which the epilogue drops.
    const int mrow = (lane & 15) < a.M ? (lane & 15) : 0;
    const H *xp = (const H *)ph.X + (size_t)ks0 * 512 + ((lane >> 4) * 16 + mrow) * 8;
    auto ldx = [&](int k) -> u32x4 { return *reinterpret_cast<const u32x4 *>(xp + (size_t)k * 512); };
    u32x4 x[CH_XPF];
#pragma unroll
    for (int u = 0; u < CH_XPF; ++u) x[u] = u < nk ? ldx(u) : u32x4{0u, 0u, 0u, 0u};
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    int base = 0;
    // rounds in which every slot that is consumed is refilled (no bounds checks: the waits stay counted, not drained)
    for (; base + 2 * CH_PF <= nk; base += CH_PF) {
#pragma unroll
        for (int u = 0; u < CH_PF; ++u) {
            acc = mfma16<H>(w[u], x[u % CH_XPF], acc);
            w[u] = __builtin_nontemporal_load(wp + (size_t)(base + u + CH_PF) * 64);
            x[u % CH_XPF] = ldx(base + u + CH_XPF);
        }
    }
    for (; base < nk; base += CH_PF) {
#pragma unroll
        for (int u = 0; u < CH_PF; ++u) {
            const int k = base + u;
            if (k < nk) {
                acc = mfma16<H>(w[u], x[u % CH_XPF], acc);
                if (k + CH_PF < nk) w[u] = __builtin_nontemporal_load(wp + (size_t)(k + CH_PF) * 64);
                if (k + CH_XPF < nk) x[u % CH_XPF] = ldx(k + CH_XPF);
            }
        }
    }
    if (poisoned) acc[0] = __uint_as_float(0x7fc00000u);
    red[wv][0][lane] = acc;
    __syncthreads();
    const ChainEpi<H> e = {(H *)ph.out, (const H *)ph.bias, ph.n_out, a.cos_t, a.sin_t, a.Hq, a.Hkv, a.D, ph.layer, a.q_scale, a.tab};
    gemm_epilogue_step<1, EPI, 1, 1, H, ChainEpi<H>, true>(red, 0, ph.part, a.M, 16, N, sb, ntg, e);
    if (!(a.flags & 8)) chain_signal(a.ctr, ph.sig_slot, b, ph.nblk, a.epoch, false, (a.flags & 1) != 0);      // the epilogue's stores are wave 0's
}

// residual_norm_kernel's row on 256 threads: real thread t stands for the kernel's threads t, t + 256, ... (< rn_threads,
// a multiple of 64), so every "virtual wave" is a real wave in one round and the statistics are summed in the kernel's
// order: per thread over its 2 column groups, per wave by wave_sum, then over the waves in index order.
struct ChSyncThreads { __device__ __forceinline__ void operator()() const { __syncthreads(); } };
template <typename T, int KIND, int NJ = 4, typename BAR = ChSyncThreads>      // NJ rounds of 256 threads cover rn_threads (<= 1024)
__device__ __forceinline__ void chain_rn(const ChainPhase &ph, const ChainArgs<T> &a, int row, float *sh, BAR bar = BAR()) {
#pragma clang fp contract(off)                                    // see norm_row (model_kernels.h)
    const int H = a.hidden, RT = a.rn_threads, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, nw = RT >> 6;
    T *xr = (T *)ph.xres + (size_t)row * H;
    T *h = (T *)ph.out;
    const T *bias = (const T *)ph.bias, *w = (const T *)ph.nw, *bb = (const T *)ph.nb;
    const int mode = ph.mode;

    if (!(a.flags & 2)) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");

    float v[NJ][RN_RG][4];
    uint2 wraw[NJ][RN_RG], braw[NJ][RN_RG];
    bool on[NJ][RN_RG];
    float aj[NJ], a2j[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        aj[j] = 0.f;
        a2j[j] = 0.f;
        const int vt = tid + 256 * j;
#pragma unroll
        for (int g = 0; g < RN_RG; ++g) {
            const int i = (vt + g * RT) * 4;
            on[j][g] = vt < RT && i < H;
            if (on[j][g]) {
                const f32x4 y4 = reduce_part4(ph.slab, ph.S, ph.stride_s, (size_t)row * H + i);
                float xin[4], bi[4] = {0.f, 0.f, 0.f, 0.f};
                load4<T>(xr + i, xin);
                if (bias) load4<T>(bias + i, bi);
                if (mode != RES_NONE) {
                    wraw[j][g] = *reinterpret_cast<const uint2 *>(w + i);
                    if (KIND == NORM_LN) braw[j][g] = *reinterpret_cast<const uint2 *>(bb + i);
                }
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    float y = y4[c];
                    if (bias) y += bi[c];
                    v[j][g][c] = rnd<T>(xin[c] + rnd<T>(y));
                    aj[j] += v[j][g][c];
                    a2j[j] += v[j][g][c] * v[j][g][c];
                }
                if (mode != RES_POST) store4_maybe_wt<true>(xr + i, v[j][g][0], v[j][g][1], v[j][g][2], v[j][g][3]);
                if (mode == RES_NONE) store4_maybe_wt<true>(h + xoff<T>(row, i, H), v[j][g][0], v[j][g][1], v[j][g][2], v[j][g][3]);
            }
        }
    }
    if (mode != RES_NONE) {
        // block_sum over the kernel's rn_threads threads: virtual wave 4 j + wv is this wave in round j
        auto vsum = [&](const float (&p)[NJ]) -> float {
            bar();
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const float ws = wave_sum(p[j]);
                if (lane == 0 && 4 * j + wv < nw) sh[4 * j + wv] = ws;
            }
            bar();
            float r = 0.f;
            for (int i = 0; i < nw; ++i) r += sh[i];
            return r;
        };
        float mean = 0.f, r;
        if (KIND == NORM_RMS) {
            r = rsqrtf(vsum(a2j) / (float)H + a.eps);
        } else {
            mean = vsum(aj) / (float)H;
            float d2[NJ];
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                d2[j] = 0.f;
#pragma unroll
                for (int g = 0; g < RN_RG; ++g)
                    if (on[j][g])
#pragma unroll
                        for (int c = 0; c < 4; ++c) { const float d = v[j][g][c] - mean; d2[j] += d * d; }
            }
            r = 1.0f / sqrtf(vsum(d2) / (float)H + a.eps);
        }
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int g = 0; g < RN_RG; ++g) {
                if (!on[j][g]) continue;
                const int i = (tid + 256 * j + g * RT) * 4;
                float wv4[4], bv4[4], o[4];
                const T *wh = reinterpret_cast<const T *>(&wraw[j][g]), *bh = reinterpret_cast<const T *>(&braw[j][g]);
#pragma unroll
                for (int c = 0; c < 4; ++c) { wv4[c] = to_f(wh[c]); bv4[c] = KIND == NORM_LN ? to_f(bh[c]) : 0.f; }
#pragma unroll
                for (int c = 0; c < 4; ++c)
                    o[c] = KIND == NORM_RMS ? rnd<T>(wv4[c] * rnd<T>(v[j][g][c] * r))
                                            : rnd<T>((v[j][g][c] - mean) * r * wv4[c] + bv4[c]);
                store4_maybe_wt<true>(h + xoff<T>(row, i, H), o[0], o[1], o[2], o[3]);
                if (mode == RES_POST) store4_maybe_wt<true>(xr + i, o[0], o[1], o[2], o[3]);
            }
    }
}

template <typename H, int ARCH>
__global__ __launch_bounds__(256, CH_WAVES) void chain_kernel(ChainArgs<H> a) {
    __shared__ f32x4 red[4][1][64];
    __shared__ float sh[32];
    int p = 0;
#pragma unroll
    for (int i = 1; i < CH_MAX_PHASES; ++i)
        if (i < a.n_phases && (int)blockIdx.x >= a.ph[i].blk0) p = i;
    const ChainPhase &ph = a.ph[p];
    const int b = (int)blockIdx.x - ph.blk0;
    constexpr int EPI_ACT = ARCH == SD_ARCH_LLAMA ? EPI_ACT_SILU : EPI_ACT_RELU;
    constexpr int EPI_QKV = ARCH == SD_ARCH_LLAMA ? EPI_QKV_ROPE : EPI_QKV_PLAIN;
    constexpr int KIND = ARCH == SD_ARCH_LLAMA ? NORM_RMS : NORM_LN;
    if (ph.type == CH_RN) {
        if (!(a.flags & 4)) chain_wait(a.ctr, ph.wait_slot, ph.wait_n, a.epoch, a.err);
        chain_rn<H, KIND>(ph, a, b, sh);
        if (!(a.flags & 8)) chain_signal(a.ctr, ph.sig_slot, b, ph.nblk, a.epoch, true, (a.flags & 1) != 0);
    }
    else if (ph.epi == EPI_PART) chain_gemm<EPI_PART, H>(ph, a, b, red, reinterpret_cast<int *>(&sh[31]));
    else if (ph.epi == EPI_ACT) chain_gemm<EPI_ACT, H>(ph, a, b, red, reinterpret_cast<int *>(&sh[31]));
    else chain_gemm<EPI_QKV, H>(ph, a, b, red, reinterpret_cast<int *>(&sh[31]));
}

// Two-level arrival (the engine): arriver g of n_arrive adds 1 to shard counter g % nsh; the last arriver of a shard adds
// 1 to the phase's "done" word, which is the one word the waiters poll (polling one word measured better than one wave
// reading all shard counters).
__device__ __forceinline__ int ch_nsh(int n_arrive) { return min(CH_SHARDS, n_arrive); }
__device__ __forceinline__ void chain_arrive(unsigned *ctr, int p, int g, int n_arrive, unsigned epoch) {
    unsigned *base = ctr + (size_t)p * CH_CTR_WORDS;
    const int nsh = ch_nsh(n_arrive), sh = g % nsh;
    const unsigned per = (unsigned)(n_arrive / nsh + (sh < n_arrive % nsh ? 1 : 0));
    const unsigned old = __hip_atomic_fetch_add(base + sh * CH_SHARD_STRIDE, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (old + 1 == per * epoch)
        __hip_atomic_fetch_add(base + CH_SHARDS * CH_SHARD_STRIDE, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ------------------------------------------------------------------------------------------------------------------
// The same phases on a persistent ENGINE (SD_CHAIN=2): one workgroup per CU = 4 consumer waves + 2 loader waves + 1 epilogue wave.
//
// A loader (EN_NL = 2 of them: one for consumers 0 and 1, one for 2 and 3) walks this CU's items (item c, c + NCU, ... of
// every GEMM phase, in phase order) and moves its consumers' quarters of each item's weight tiles HBM -> LDS with LDS-DMA
// (global_load_lds_dwordx4: one 1 KiB tile per instruction, no registers), alternating between its consumers, into a
// ring of EN_RS tiles per consumer (128 KiB in all).  It depends on nothing but free ring slots, so it runs ahead of every
// hand-over: across items, across phase boundaries, while the consumers wait for the norm rows - the CU's HBM pipe only
// stops when the rings are full of unread weights.
// Consumer q takes its quarter's tiles out of its ring in order (ds_read_b128, lane-linear = the MFMA operand), reads
// its activations from L2 as the streaming kernel does, and accumulates k-step by k-step - the streaming kernel's
// arithmetic and order, so the results stay bit-identical.  The four accumulators go to LDS, where the epilogue wave folds
// them, runs the epilogue (write-through stores) and, after the CU's last item of a phase, counts the workgroup's arrival:
// the consumers go straight on to the next item.
// Norm rows: row r is computed by the consumers of workgroup r (their loader keeps going meanwhile).
//
// Inside the workgroup everything is handed over through LDS words (`retired[L]`: tiles of loader L landed, in its issue
// order; `consumed[q]`;
// accumulators handed to the epilogue wave; a 4-wave barrier of the consumers for the norm rows):
// s_barrier would stop the loader too.  Between workgroups: the phase counters above.
// ------------------------------------------------------------------------------------------------------------------
#ifndef EN_NL
#define EN_NL 2                                              // loader waves (each serves 4 / EN_NL consumers); measured: 2 beats 4
#endif
#ifndef EN_RS
#define EN_RS 32                                             // ring slots (1 KiB tiles) per consumer: 128 KiB of LDS in all
#endif
#ifndef EN_D
#define EN_D 40                                              // tiles a loader keeps in flight before it waits for the oldest (vmcnt <= 63)
#endif
#ifndef EN_XB
#define EN_XB 3                                              // batches of 4 k-steps of activations a consumer requests ahead
#endif
#ifndef EN_NT
#define EN_NT " nt"                                          // cache policy of the LDS-DMA weight loads
#endif
#define EN_RB 2                                              // items whose four accumulators may wait for the epilogue wave
#define EN_THREADS (64 * (5 + EN_NL))                        // waves 0..3: consumers; then the loaders; last: epilogues

struct EngShared {
    u32x4 ring[4][EN_RS][64];
    f32x4 red[EN_RB][4][1][64];
    float sh[32];
    unsigned retired[4];                                     // tiles landed in LDS, per loader (in its stream order)
    unsigned consumed[4];                                    // tiles each consumer has read out of its ring
    unsigned bar_cnt;                                        // consumer barrier (norm rows): arrivals, 4 per generation
    unsigned red_cnt;                                        // accumulators written: 4 per item
    unsigned epi_done;                                       // items whose folded accumulators have been read
    unsigned phase_ok;                                       // 1 + last phase whose inputs consumer 0 saw complete
};

__device__ __forceinline__ unsigned lds_ld(const unsigned *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void lds_st(unsigned *p, unsigned v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }

// barrier of the 4 consumer waves (threads 0..255); `gen` counts the calls
__device__ __forceinline__ void en_cbar(EngShared &S, unsigned &gen) {
    ++gen;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if ((threadIdx.x & 63) == 0) __hip_atomic_fetch_add(&S.bar_cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    while (lds_ld(&S.bar_cnt) < 4u * gen) __builtin_amdgcn_s_sleep(0);
    asm volatile("" ::: "memory");
}

// one 1 KiB tile HBM -> LDS: lane l's 16 bytes at gsrc go to lds_dst + 16 l (M0 carries the wave-uniform LDS address)
__device__ __forceinline__ void glds16(const u32x4 *gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" EN_NT "\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

template <typename H, int ARCH>
__global__ __launch_bounds__(EN_THREADS) void engine_kernel(ChainArgs<H> a) {
    constexpr int EPI_ACT = ARCH == SD_ARCH_LLAMA ? EPI_ACT_SILU : EPI_ACT_RELU;
    constexpr int EPI_QKV = ARCH == SD_ARCH_LLAMA ? EPI_QKV_ROPE : EPI_QKV_PLAIN;
    constexpr int KIND = ARCH == SD_ARCH_LLAMA ? NORM_RMS : NORM_LN;
    extern __shared__ __attribute__((aligned(16))) char en_smem[];
    EngShared &S = *reinterpret_cast<EngShared *>(en_smem);
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int c = blockIdx.x, NCU = gridDim.x;
    if (threadIdx.x == 0) {
        S.retired[0] = S.retired[1] = S.retired[2] = S.retired[3] = 0; S.bar_cnt = 0; S.red_cnt = 0; S.epi_done = 0;
        S.phase_ok = 1;                                           // phase 0's inputs come from the launch before
        S.consumed[0] = S.consumed[1] = S.consumed[2] = S.consumed[3] = 0;
    }
    __syncthreads();                                              // (the only s_barrier: all 9 waves, before any LDS-DMA)

    if (wv >= 4 && wv < 4 + EN_NL) {
        // ---------------- loader L: the tiles of consumers L * CPL .. + CPL - 1 in turn (its tile n belongs to consumer
        // L * CPL + n % CPL: every item holds the same number of tiles for each of them) ----------------
        constexpr int CPL = 4 / EN_NL;
        const int L = wv - 4;
        unsigned n = 0, pub = 0;                                  // tiles issued / published as landed
        unsigned jq = 0;                                          // tiles issued per consumer
        unsigned seen[CPL];                                       // last value read of consumed[]
#pragma unroll
        for (int h = 0; h < CPL; ++h) seen[h] = 0u;
        const unsigned ring0 = (unsigned)(uintptr_t)&S.ring[L * CPL][0][0];
        for (int p = 0; p < a.n_phases; ++p) {
            const ChainPhase &ph = a.ph[p];
            if (ph.type != CH_GEMM) continue;
            const int KS = ph.K >> 5, NTG = ph.N >> 4, nk = ph.ks_per_blk >> 2;   // k-steps per consumer (quarters are equal)
            for (int it = c; it < ph.nblk; it += NCU) {
                const int sb = it / NTG, ntg = it - sb * NTG;
                const u32x4 *wp = ph.W + ((size_t)ntg * KS + (size_t)sb * ph.ks_per_blk + (size_t)(L * CPL) * nk) * 64 + lane;
                for (int k = 0; k < nk; ++k) {
#pragma unroll
                    for (int h = 0; h < CPL; ++h) {
                        const unsigned j = jq + (unsigned)k;
                        if (j - seen[h] >= EN_RS) {
                            // that ring looks full: first look again, then let the oldest half land and publish it, and only
                            // then everything (the consumer may be waiting for a tile that is not published yet)
                            seen[h] = lds_ld(&S.consumed[L * CPL + h]);
                            if (j - seen[h] >= EN_RS) {
                                if (n - pub > EN_D / 2) {
                                    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(EN_D / 2) : "memory");
                                    pub = n - EN_D / 2;
                                    lds_st(&S.retired[L], pub);
                                }
                                while (j - (seen[h] = lds_ld(&S.consumed[L * CPL + h])) >= EN_RS) {
                                    if (pub != n) {
                                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                                        pub = n;
                                        lds_st(&S.retired[L], pub);
                                    }
                                    __builtin_amdgcn_s_sleep(0);
                                }
                            }
                        }
                        const unsigned dst = ring0 + ((unsigned)h * EN_RS + (j % EN_RS)) * 1024u;
                        glds16(wp + ((size_t)h * nk + k) * 64, (unsigned)__builtin_amdgcn_readfirstlane((int)dst));
                        ++n;
                    }
                    if (n - pub >= EN_D + 4) {                    // after this wait at most EN_D are in flight: the others have landed
                        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(EN_D) : "memory");
                        pub = n - EN_D;
                        lds_st(&S.retired[L], pub);
                    }
                }
                jq += (unsigned)nk;
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        lds_st(&S.retired[L], n);
        return;
    }

    if (wv == 4 + EN_NL) {
        // ---------------- epilogue wave: folds the four accumulators of every item, runs the GEMM's epilogue (write-through
        // stores) and counts the workgroup's arrival after the CU's last item of a phase; the consumers never wait for it
        // unless they get EN_RB items ahead ----------------
        unsigned seq = 0;
        for (int p = 0; p < a.n_phases; ++p) {
            const ChainPhase &ph = a.ph[p];
            if (ph.type != CH_GEMM) continue;
            const int NTG = ph.N >> 4;
            const ChainEpi<H> e = {(H *)ph.out, (const H *)ph.bias, ph.n_out, a.cos_t, a.sin_t, a.Hq, a.Hkv, a.D, ph.layer,
                                   a.q_scale, a.tab};
            for (int it = c; it < ph.nblk; it += NCU) {
                const int sb = it / NTG, ntg = it - sb * NTG;
                while (lds_ld(&S.red_cnt) < 4u * (seq + 1u)) __builtin_amdgcn_s_sleep(0);
                asm volatile("" ::: "memory");
                f32x4 (*red)[1][64] = S.red[seq % EN_RB];
                if (a.flags & 64) {}                              // (timing experiment: no epilogue)
                else if (ph.epi == EPI_PART) gemm_epilogue_step<1, EPI_PART, 1, 1, H, ChainEpi<H>, true>(red, 0, ph.part, a.M, 16, ph.N, sb, ntg, e, lane);
                else if (ph.epi == EPI_ACT) gemm_epilogue_step<1, EPI_ACT, 1, 1, H, ChainEpi<H>, true>(red, 0, ph.part, a.M, 16, ph.N, sb, ntg, e, lane);
                else gemm_epilogue_step<1, EPI_QKV, 1, 1, H, ChainEpi<H>, true>(red, 0, ph.part, a.M, 16, ph.N, sb, ntg, e, lane);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                ++seq;
                if (lane == 0) lds_st(&S.epi_done, seq);
            }
            if (p + 1 < a.n_phases) {                             // (also when this CU had no item in the phase)
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (lane == 0) chain_arrive(a.ctr, p, c, NCU, a.epoch);
            }
        }
        return;
    }

    // ---------------- consumers (threads 0..255) ----------------
    const int q = wv;
    const int xlane = ((lane >> 4) * 16 + ((lane & 15) < a.M ? (lane & 15) : 0)) * 8;   // rows >= M: row 0's bytes (dropped later)
    unsigned jb = 0;                                              // tiles this consumer has taken
    unsigned gen = 0;                                             // consumer-barrier generation
    unsigned seq = 0;                                             // items done
    constexpr int CPLC = 4 / EN_NL;                               // (consumers per loader)
    const unsigned *my_retired = &S.retired[q / CPLC];
    auto wait_phase = [&](int p, int n_arrive) {                  // inputs of phase p (outputs of p - 1) complete?
        if (p == 0 || (a.flags & 4)) return;
        if (q == 0) {
            if (lane == 0 && lds_ld(&S.phase_ok) < (unsigned)p + 1u) {
                unsigned *done = a.ctr + (size_t)(p - 1) * CH_CTR_WORDS + CH_SHARDS * CH_SHARD_STRIDE;
                const unsigned want = (unsigned)ch_nsh(n_arrive) * a.epoch;
                const long long t0 = wall_clock64();
                while (__hip_atomic_load(done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
                    __builtin_amdgcn_s_sleep(2);
                    if (wall_clock64() - t0 > CH_TIMEOUT_TICKS) {
                        __hip_atomic_fetch_or(a.err, 1u << (p - 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        break;
                    }
                }
                lds_st(&S.phase_ok, (unsigned)p + 1u);
            }
        }
        while (lds_ld(&S.phase_ok) < (unsigned)p + 1u) __builtin_amdgcn_s_sleep(0);
        asm volatile("" ::: "memory");
    };
    for (int p = 0; p < a.n_phases; ++p) {
        const ChainPhase &ph = a.ph[p];
        const int prev_arrive = p == 0 ? 0 : (a.ph[p - 1].type == CH_RN ? a.ph[p - 1].nblk : NCU);
        if (ph.type == CH_RN) {
            if (c < ph.nblk) {                                    // row c: this workgroup's consumers
                wait_phase(p, prev_arrive);
                auto bar = [&]() { en_cbar(S, gen); };
                if (a.rn_threads <= 512) chain_rn<H, KIND, 2>(ph, a, c, S.sh, bar);
                else if (a.rn_threads <= 768) chain_rn<H, KIND, 3>(ph, a, c, S.sh, bar);
                else chain_rn<H, KIND, 4>(ph, a, c, S.sh, bar);
                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");     // every wave stored (write-through): drain
                en_cbar(S, gen);
                if (threadIdx.x == 0 && p + 1 < a.n_phases) chain_arrive(a.ctr, p, c, ph.nblk, a.epoch);
            }
            continue;
        }
        const int nk = ph.ks_per_blk >> 2;
        bool any = false;
        for (int it = c; it < ph.nblk; it += NCU) {
            if (!any) { wait_phase(p, prev_arrive); any = true; }
            const int sb = it / (ph.N >> 4);
            const H *xp = (const H *)ph.X + ((size_t)sb * ph.ks_per_blk + (size_t)q * nk) * 512 + xlane;
            const bool dbg_nox = (a.flags & 16) != 0, dbg_nomm = (a.flags & 32) != 0;       // timing experiments (wrong results)
            auto ldx = [&](int k) -> u32x4 { return (k < nk && !dbg_nox) ? *reinterpret_cast<const u32x4 *>(xp + (size_t)k * 512) : u32x4{0u, 0u, 0u, 0u}; };
            // activations EN_XB batches of 4 k-steps ahead (an L2 / memory round trip each); the weights come out of the ring
            u32x4 x[EN_XB][4];
#pragma unroll
            for (int b = 0; b < EN_XB; ++b)
#pragma unroll
                for (int u = 0; u < 4; ++u) x[b][u] = ldx(4 * b + u);
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            for (int k0 = 0; k0 < nk; k0 += 4 * EN_XB) {
#pragma unroll
                for (int b = 0; b < EN_XB; ++b) {
                    const int k = k0 + 4 * b;
                    if (k < nk) {
                        const int nb = min(4, nk - k);
                        const unsigned need = (unsigned)CPLC * (jb + (unsigned)(k + nb - 1)) + (unsigned)(q % CPLC);   // its loader's stream index of the batch's last tile
                        while (lds_ld(my_retired) <= need) __builtin_amdgcn_s_sleep(0);
                        asm volatile("" ::: "memory");
                        u32x4 w[4];
                        if (!dbg_nomm) {
#pragma unroll
                        for (int u = 0; u < 4; ++u)
                            w[u] = u < nb ? S.ring[q][(jb + (unsigned)(k + u)) % EN_RS][lane] : u32x4{0u, 0u, 0u, 0u};
#pragma unroll
                        for (int u = 0; u < 4; ++u)
                            if (u < nb) acc = mfma16<H>(w[u], x[b][u], acc);
                        }
                        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // the tiles are in registers: their slots are free
                        if (lane == 0) lds_st(&S.consumed[q], jb + (unsigned)(k + nb));
#pragma unroll
                        for (int u = 0; u < 4; ++u) x[b][u] = ldx(k + 4 * EN_XB + u);
                    }
                }
            }
            jb += (unsigned)nk;
            // hand the accumulator to the epilogue wave (slot seq % EN_RB: item seq - EN_RB must have been folded)
            while (seq >= EN_RB && lds_ld(&S.epi_done) < seq - (EN_RB - 1)) __builtin_amdgcn_s_sleep(0);
            S.red[seq % EN_RB][q][0][lane] = acc;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (lane == 0) __hip_atomic_fetch_add(&S.red_cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            ++seq;
        }
    }
}
