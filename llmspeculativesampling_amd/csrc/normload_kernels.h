// The norm-on-load seam (single-stream decode / verify rows, <= 8, Llama RMSNorm, 16-bit models): the residual add after
// the attention block moves into the epilogue of the launch that produces the rows (attention + O projection,
// fused_kernels.h) and the normalisation into the operand load of the GEMM that consumes them (gate/up), so that the
// residual+norm launch between them - 4.9 us on <= 8 workgroups with HBM idle - disappears: five launches per layer
// instead of six.  Reference: modeling_llama.py:405-457 (layer), :75-89 (RMSNorm).
// The second seam of a layer (down projection -> next layer's QKV) keeps its residual+norm launch: the down projection
// cuts K over four workgroups per tile, so no workgroup holds a complete sum to add the residual to (DESIGN.md section 7
// records what was tried there).
#pragma once
#include "model_kernels.h"
#include "rows_kernels.h"

// ---- the streaming GEMM for <= 8 UN-normalised rows: RMSNorm applied while the operand is loaded -----------------------
// (modeling_llama.py:84-89: x.float() * rsqrt(mean(x^2) + eps), back to the weight dtype, times the weight.)
// The producer of the rows (resid_epilogue_step, model_kernels.h) leaves the residual rows in the operand tile layout
// and, per row and 16-column tile, the sum of their squares; this kernel's workgroups add the partials up in a FIXED
// order - (wave, quad) takes nt / 16 consecutive tiles in sequence, the four quads of a wave fold as (q0 + q1) + (q2 + q3),
// the four waves likewise - and scale every element on its way into the MFMA: rnd(w * rnd(x * r)), element for element
// what residual_norm_kernel stores.  What it buys: the residual+norm launches of a layer (4.8 us each on <= 16
// workgroups, HBM idle, + a kernel boundary) disappear.
// What it costs is VALU time in a kernel that has none to spare (every workgroup converts the rows of its own k-range:
// ~46 instructions per 16 x 32 fragment, of which only M of the 16 row slots carry a row - measured +2.4 us on gate/up's
// 43.7 when done fragment by fragment, +1.5 in this form at 5 rows, +0.4 at 3), hence the COMPACT conversion: the M valid rows of C = 16 / M consecutive k-steps (C = 3
// at gamma + 1 = 5 rows) are loaded into ONE register set - lane (s, quad, m) holds row m's 8 elements of quad `quad` of
// k-step s -, converted once, and each k-step's MFMA operand is then gathered from it with four ds_bpermute_b32 (the
// LDS crossbar, not the VALU); row slots >= M get a copy of row 0, whose output columns the epilogue drops.
// The prologue (partials, norm weight to LDS by DMA, one barrier) runs behind the first group's weight requests, which
// are issued first-needed last (VMEM returns in order): norm weight, partials, then the weights.
#define XN_MAX4 5                                             // 16-byte pieces of partials per lane per pass (hidden <= 5120: one pass)
template <typename H>
__device__ __forceinline__ u32x4 norm_frag(u32x4 x, u32x4 g, float r) {
#pragma clang fp contract(off)
    H xv[8], gv[8], o[8];
    *reinterpret_cast<u32x4 *>(xv) = x;
    *reinterpret_cast<u32x4 *>(gv) = g;
#pragma unroll
    for (int i = 0; i < 8; ++i) o[i] = (H)(to_f(gv[i]) * to_f((H)(to_f(xv[i]) * r)));
    return *reinterpret_cast<const u32x4 *>(o);
}
// C: k-steps whose M valid rows fit one conversion register (C * 4 * M <= 64 lanes).  Weight requests stay in groups of
// four k-steps (the streaming kernel's measured optimum; groups of three ran 4 % slower whatever the conversion cost), so
// a group takes NCV = ceil(4 / C) conversions, the last one covering what is left of the four.
template <int EPI, int C, typename H = bf16_t>
__global__ __launch_bounds__(256, 8) void gemm_bf16_stream_xn(const u32x4 *__restrict__ Wp, const H *__restrict__ X,
                                                             float *__restrict__ part, int M, int N, int K, int SB,
                                                             int ks_per_blk, GemmEpiT<H> e) {
    constexpr int G = 4, NCV = (G + C - 1) / C;
    __shared__ f32x4 red[4][1][64];
    __shared__ float ssq_sh[4][16];
    extern __shared__ __attribute__((aligned(16))) char xn_smem[];           // the norm weight, K elements
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int NTG = N >> 4, KS = K >> 5;
    const int sb = blockIdx.x / NTG, ntg = blockIdx.x - sb * NTG;
    const int kb0 = sb * ks_per_blk, kb1 = min(KS, kb0 + ks_per_blk);
    const int per = (kb1 - kb0 + 3) >> 2;
    const int ks0 = min(kb1, kb0 + wv * per), ks1 = min(kb1, ks0 + per), ksa = min(ks0, KS - 1), nk = ks1 - ks0;
    const u32x4 *wp = Wp + ((size_t)ntg * KS + ksa) * 64 + lane;
    // ---- prologue requests: norm weight -> LDS by DMA (no registers), the row's partials, then the first group's weights
    const unsigned gs_base = (unsigned)(uintptr_t)xn_smem;
    const u32x4 *gsrc = reinterpret_cast<const u32x4 *>(e.nrm_w);
    for (int j = wv; j < (K >> 9); j += 4) gr_glds16(gsrc + (size_t)j * 64 + lane, gs_base + (unsigned)j * 1024u);   // K % 512 == 0
    const int mrow = (lane & 15) < M ? (lane & 15) : 0;
    const int nt = e.nrm_nt, tper = nt >> 4, n4 = tper >> 2;     // nt % 64 == 0
    const float4 *sp = reinterpret_cast<const float4 *>(e.nrm_ssq + (size_t)mrow * nt + (wv * 4 + (lane >> 4)) * tper);
    float4 sv[XN_MAX4];
#pragma unroll
    for (int j = 0; j < XN_MAX4; ++j) sv[j] = sp[min(j, n4 - 1)];
    asm volatile("" ::: "memory");
    u32x4 w[G], xc[NCV];
    const bool g0 = nk >= G;
#pragma unroll
    for (int u = 0; u < G; ++u) w[u] = __builtin_nontemporal_load(wp + (g0 ? (size_t)u * 64 : 0));
    asm volatile("" ::: "memory");
    // ---- row totals
    float sum = 0.f;
#pragma unroll
    for (int j = 0; j < XN_MAX4; ++j) {
        const bool on = j < n4;
        sum += on ? sv[j].x : 0.f; sum += on ? sv[j].y : 0.f; sum += on ? sv[j].z : 0.f; sum += on ? sv[j].w : 0.f;
    }
    for (int j0 = XN_MAX4; j0 < n4; j0 += XN_MAX4) {             // (hidden > 5120: further passes)
#pragma unroll
        for (int j = 0; j < XN_MAX4; ++j) sv[j] = sp[min(j0 + j, n4 - 1)];
#pragma unroll
        for (int j = 0; j < XN_MAX4; ++j) {
            const bool on = j0 + j < n4;
            sum += on ? sv[j].x : 0.f; sum += on ? sv[j].y : 0.f; sum += on ? sv[j].z : 0.f; sum += on ? sv[j].w : 0.f;
        }
    }
    sum += __shfl_xor(sum, 16, 64);
    sum += __shfl_xor(sum, 32, 64);
    if (lane < 16) ssq_sh[wv][lane] = sum;
    asm volatile("s_waitcnt vmcnt(%0)" :: "n"(G) : "memory");    // the DMA pieces (older than the partials) have landed
    __syncthreads();
    // ---- compact lane (s, quad, m) = lane cj of a conversion register; lanes past C * 4 * M repeat lane 0
    const int cj = lane < C * 4 * M ? lane : 0;
    const int cm = cj % M, ct = cj / M, cq = ct & 3, cs = ct >> 2;
    const float tot = (ssq_sh[0][cm] + ssq_sh[1][cm]) + (ssq_sh[2][cm] + ssq_sh[3][cm]);
    const float r = rsqrtf(tot / (float)K + e.nrm_eps);
    const H *xp = X + (size_t)ksa * 512 + (cq * 16 + cm) * 8;                                 // this lane's piece of k-step ksa
    const H *gp = reinterpret_cast<const H *>(xn_smem) + (size_t)ksa * 32 + cq * 8;
    // the MFMA operand of k-step s of a conversion: lane (quad, m') takes the piece of compact lane (s, quad, min(m', M - 1))
    const int bsrc = (((lane >> 4) * M) + min(lane & 15, M - 1)) * 4;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    // this lane's k-step inside a group of `left` valid k-steps, for conversion v (k-steps v * C ...): lanes whose k-step is
    // past the conversion's (or the group's) range repeat the conversion's first - finite values nobody reads or that meet
    // a zeroed weight fragment
    auto kof = [&](int v, int left) { return v * C + ((cs < C && v * C + cs < left) ? cs : 0); };
    auto issue_x = [&](int ks, int left) {
#pragma unroll
        for (int v = 0; v < NCV; ++v) xc[v] = *reinterpret_cast<const u32x4 *>(xp + (size_t)(ks + min(kof(v, left), max(left - 1, 0))) * 512);
    };
    auto compute = [&](int ks, int left) {
#pragma unroll
        for (int v = 0; v < NCV; ++v) {
            xc[v] = norm_frag<H>(xc[v], *reinterpret_cast<const u32x4 *>(gp + (size_t)(ks + min(kof(v, left), max(left - 1, 0))) * 32), r);
            __builtin_amdgcn_sched_barrier(0);                    // one conversion's temporaries at a time (64 VGPRs)
        }
#pragma unroll
        for (int u = 0; u < G; ++u) {
            u32x4 x, wz = w[u];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                x[i] = (unsigned)__builtin_amdgcn_ds_bpermute(bsrc + (u % C) * 16 * M, (int)xc[u / C][i]);
                wz[i] = u < left ? wz[i] : 0u;
            }
            acc = mfma16<H>(wz, x, acc);
        }
    };
    int ks = 0;                                                   // k-steps done, relative to ks0
    if (g0) {                                                     // first group: its weights are already on their way
        issue_x(0, G);
        compute(0, G);
        ks = G;
    }
    for (; ks + G <= nk; ks += G) {                               // whole groups: rows (L2) first, then the weights (HBM)
        issue_x(ks, G);
        asm volatile("" ::: "memory");
#pragma unroll
        for (int u = 0; u < G; ++u) w[u] = __builtin_nontemporal_load(wp + (size_t)(ks + u) * 64);
        compute(ks, G);
    }
    if (ks < nk) {                                                // the last < 4 k-steps: past-the-range steps repeat the first
        const int left = nk - ks;
        issue_x(ks, left);
#pragma unroll
        for (int u = 0; u < G; ++u) w[u] = __builtin_nontemporal_load(wp + (size_t)(ks + (u < left ? u : 0)) * 64);
        compute(ks, left);
    }
    red[wv][0][lane] = acc;
    __syncthreads();
    gemm_epilogue_step<1, EPI, 1, 1, H>(red, 0, part, M, 16, N, sb, ntg, e);
}

// ---- the k-split streaming GEMM of <= 16 rows that finishes with the residual epilogue ------------------------------------
// The down projection (N = hidden: 320 tiles at 13b) needs its k-range cut over SB workgroups per tile to fill the chip
// evenly - one 16-wave workgroup per tile, whole k-range, was tried: 31 us against 24, the CUs that get two tiles set
// the time - so no workgroup holds a complete sum.  Here the workgroups of slabs 0 .. SB-2 store their slab write-through
// (sc1), drain their stores and count in on the tile's counter; the workgroup of the LAST slab - dispatched last, so every
// other one is resident before it - keeps its sums in registers, waits for the SB - 1 arrivals, reads their slabs, folds in
// slab order (((s0 + s1) + s2) + s3: the order reduce_part4 uses, so the sums are those of the slab path bit for bit) and
// runs resid_epilogue_step.  The slabs of this kernel are TILE-MAJOR - [slab][n-tile][16 rows][16 columns], 1 KiB per
// (slab, tile), every 128-byte line of it written whole by one wave of one workgroup and read by one finisher - so that no
// finisher can pull in a line that still waits for another tile's producer (row-major slabs put tiles 2j and 2j + 1 on
// one line), and the finisher reads them with sc1 loads after its poll has matched: the write-through hand-off of
// MI355X_MICROARCH.md ("inter-workgroup visibility", first row of the sc1 table) in every cell.  (Giving the
// last slab a larger share of K so that the others arrive early did not help: 92 / 108 / 116 / 124 % of an equal share
// all ran equal or slower - the tail is the finisher's own read + epilogue, not the wait.)  The counters are
// monotonic (the host passes the arrivals expected so far; compared as a signed difference); a wait that exceeds 20 ms
// poisons the tile with NaN, which the sampler reports - a timed-out launch never returns plausible numbers.
#define FIN_TIMEOUT_TICKS 2000000ll                           // wall_clock64 runs at 100 MHz
// 16 bytes another workgroup of this launch stored write-through: two 8-byte sc1 loads (relaxed agent-scope atomics lower
// to global_load_dwordx2 sc1 - L1 bypassed, served by L2 / memory)
__device__ __forceinline__ f32x4 load_f32x4_sc1(const float *p) {
    const unsigned long long a = __hip_atomic_load(reinterpret_cast<const unsigned long long *>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned long long b = __hip_atomic_load(reinterpret_cast<const unsigned long long *>(p) + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return f32x4{__uint_as_float((unsigned)a), __uint_as_float((unsigned)(a >> 32)), __uint_as_float((unsigned)b),
                 __uint_as_float((unsigned)(b >> 32))};
}
template <typename H = bf16_t>
__global__ __launch_bounds__(256) void gemm_bf16_stream_fin(const u32x4 *__restrict__ Wp, const H *__restrict__ X,
                                                           float *__restrict__ part, int M, int N, int K, int SB,
                                                           int ks_per_blk, GemmEpiT<H> e, unsigned *__restrict__ ctr,
                                                           unsigned want, unsigned *__restrict__ wait_status) {
    constexpr int U = 4;
    __shared__ f32x4 red[4][1][64];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int NTG = N >> 4, KS = K >> 5;
    const int sb = blockIdx.x / NTG, ntg = blockIdx.x - sb * NTG;
    const bool fin = sb == SB - 1;
    const int kb0 = sb * ks_per_blk, kb1 = min(KS, kb0 + ks_per_blk);
    const int per = (kb1 - kb0 + 3) >> 2;
    const int ks0 = min(kb1, kb0 + wv * per), ks1 = min(kb1, ks0 + per), ksa = min(ks0, KS - 1);
    const uint2 xpre = fin ? resid_prefetch<H>(e.res_x, M, N, ntg, (int)threadIdx.x) : uint2{0u, 0u};
    const u32x4 *wp = Wp + ((size_t)ntg * KS + ksa) * 64 + lane;
    const int mrow = (lane & 15) < M ? (lane & 15) : 0;           // rows >= M read row 0 (their output columns are dropped)
    const H *xp = X + (size_t)ksa * 512 + ((lane >> 4) * 16 + mrow) * 8;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    int ks = ks0;
    for (; ks + U <= ks1; ks += U) {
        u32x4 w[U], x[U];
#pragma unroll
        for (int u = 0; u < U; ++u) w[u] = __builtin_nontemporal_load(wp + (size_t)u * 64);
#pragma unroll
        for (int u = 0; u < U; ++u) x[u] = *reinterpret_cast<const u32x4 *>(xp + u * 512);
#pragma unroll
        for (int u = 0; u < U; ++u) acc = mfma16<H>(w[u], x[u], acc);
        wp += (size_t)U * 64; xp += U * 512;
    }
    if (ks < ks1) {                                               // the last < 4 k-steps as one burst too
        const int rem = ks1 - ks;
        u32x4 w[U - 1], x[U - 1];
#pragma unroll
        for (int u = 0; u < U - 1; ++u)
            if (u < rem) { w[u] = __builtin_nontemporal_load(wp + (size_t)u * 64); x[u] = *reinterpret_cast<const u32x4 *>(xp + u * 512); }
#pragma unroll
        for (int u = 0; u < U - 1; ++u)
            if (u < rem) acc = mfma16<H>(w[u], x[u], acc);
    }
    red[wv][0][lane] = acc;
    __syncthreads();
    if (threadIdx.x >= 64) return;
    const int l = lane, m = l & 15;
    float *slab = part + ((size_t)ntg * 16 + m) * 16 + (l >> 4) * 4;             // + s * 16 * N for slab s (tile-major)
    const f32x4 own = (red[0][0][l] + red[1][0][l]) + (red[2][0][l] + red[3][0][l]);
    if (!fin) {
        if (m < M) store_f32x4<true>(slab + (size_t)sb * 16 * N, own);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // the slab has left this wave
        if (l == 0) (void)__hip_atomic_fetch_add(ctr + ntg, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return;
    }
    bool timed_out = false;
    if (SB > 1) {
        const long long t0 = wall_clock64();
        for (;;) {
            const unsigned got = __hip_atomic_load(ctr + ntg, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if ((int)(got - want) >= 0) break;
            __builtin_amdgcn_s_sleep(2);
            if (wall_clock64() - t0 > FIN_TIMEOUT_TICKS) { timed_out = true; break; }
        }
    }
    asm volatile("" ::: "memory");
    f32x4 a = {0.f, 0.f, 0.f, 0.f};
    for (int s2 = 0; s2 + 1 < SB; ++s2)
        if (m < M) a += load_f32x4_sc1(slab + (size_t)s2 * 16 * N);
    a += own;
    if (timed_out) {
        a = f32x4{__uint_as_float(0x7fc00000u), 0.f, 0.f, 0.f};
        if (l == 0) (void)__hip_atomic_fetch_or(wait_status, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    red[0][0][l] = a;
    red[1][0][l] = red[2][0][l] = red[3][0][l] = f32x4{0.f, 0.f, 0.f, 0.f};
    resid_epilogue_step<H>(red, M, N, ntg, e, l, xpre);
}
