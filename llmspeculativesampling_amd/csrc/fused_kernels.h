// Attention and the output projection of one decoder layer as ONE launch (single-stream verify / decode rows - up to 16,
// i.e. two attention row groups: gamma <= 15 -, 16-bit
// models; reference modeling_llama.py:346-372 + the o_proj of :388, modeling_opt.py:226-278).
//
// Why: after the QKV GEMM a layer runs attention - Hq workgroups on Hq CUs for ~9.6 us at 5 rows x 196 keys, HBM nearly idle -
// and only then the O projection, a weight stream (52 MB at 13b) that depends on attention's output but whose WEIGHTS do
// not.  Here the grid is Hq attention workgroups (dispatched first) + one 512-thread workgroup per PAIR of 16-column
// n-tiles of W_o; each half (four waves) of an O workgroup requests its tile's whole k-range straight away - a quarter per
// wave, NKW k-steps, register resident (40 x 1 KiB per wave at K = 5120) - so the O matrix streams from HBM WHILE
// attention runs; then wave 0 polls a device counter the attention workgroups arrive on, the workgroup reads the
// attention rows, multiplies and leaves ONE split-K slab (SB = 1: the residual+norm kernel folds one slab instead of four).
// 512 threads at ~240 VGPRs fill a CU's wave slots, so every workgroup has a CU to itself: an attention workgroup (its
// waves 4..7 exit at once) never shares the per-CU in-order memory pipeline with an O workgroup's 320 KiB of requests -
// sharing it (the first cut: 256-thread workgroups, two per CU) made the last attention workgroup arrive at 16-17 us
// instead of ~8 and the launch no faster than two (tools/ao_stamps.py).
//
// Hand-off (MI355X_MICROARCH.md, "inter-workgroup visibility"): the attention rows are
// stored write-through (sc1, 8-byte stores), every storing wave drains its stores (s_waitcnt vmcnt(0)), a workgroup
// barrier, then ONE lane adds to the counter (agent scope).  The consumer's first wave polls the counter with agent-scope
// relaxed loads, the other waves join it at a workgroup barrier, and only then is the first load of the rows issued - a
// kernel starts with clean caches and no workgroup reads those lines earlier in this launch, so no stale copy exists.
// The counter is monotonic (the host passes the number of arrivals expected so far, this launch's included; compared as
// a signed difference so it may wrap): nothing is reset between launches.
// All Hq + N/16 workgroups are resident at once (two per CU by LDS and registers, checked on the host against the CU
// count) and the attention workgroups have the lowest block indices, so the wait cannot deadlock; it is still bounded:
// a wait that exceeds its limit poisons the workgroup's slab with NaN, which the sampler reports as 'norm logits error' -
// a timed-out launch can never return plausible numbers.
#pragma once
#include "model_kernels.h"

#define AO_NKW 40                                             // k-steps a wave keeps in registers (K <= 4 * 32 * AO_NKW = 5120)
#define AO_XPF 8                                              // activation fragments requested ahead
#define AO_TIMEOUT_TICKS 2000000ll                            // wall_clock64 runs at 100 MHz: 20 ms
// Stamped runs (SD_AO_STAMPS=1, tools/ao_stamps.py): EVERY workgroup w < AO_STAMP_WGS leaves a record of 8 words at
// stamps + 8 w - wall_clock64 (10 ns ticks) at its milestones plus where it ran - so that "which attention workgroups are
// late" can be tied to a head, an XCD, a shader engine or a CU:
//   attention workgroup: [0] start, [1] scores done (K tiles landed, QK^T, LDS writes), [2] softmax done, [3] P.V done (V
//                        landed), [4] output stores issued, [5] stores acknowledged + arrival counted
//   O workgroup:         [0] start, [1] own weights landed, [2] counter seen, [3] MFMAs done, [4] epilogue stored, [5] 0
//   both:                [6] XCC_ID | HW_ID << 8 (HW_ID: wave 3:0, simd 5:4, pipe 7:6, cu 11:8, sh 12, se 15:13),
//                        [7] role << 32 | head or first n-tile << 8 | row group
#define AO_STAMP_WGS 512

template <typename T>
__global__ __launch_bounds__(512, 2) void attn_oproj_kernel(const T *__restrict__ qbuf, RowTab tab, int layer,
                                                           T *__restrict__ attn_out, int Hq, int Hkv, int arch,
                                                           float inv_sqrt_d, int s_cap, const u32x4 *__restrict__ Wo,
                                                           float *__restrict__ part, int M, int N, int K,
                                                           unsigned *__restrict__ ctr, unsigned want,
                                                           int delay_ticks, int gap_ticks, long long *__restrict__ stamps,
                                                           T *res_x, T *res_h, float *__restrict__ res_ssq,
                                                           unsigned *__restrict__ wait_status) {
    // res_ssq (or NULL): finish with the residual epilogue (resid_epilogue_step: residual rows updated, un-normalised
    // operand rows + per-tile sums of squares for the consumer's norm on load) instead of leaving the slab in `part`
    extern __shared__ __attribute__((aligned(16))) char smem[];
    long long *my_st = (stamps && (int)blockIdx.x < AO_STAMP_WGS) ? stamps + 8 * (size_t)blockIdx.x : nullptr;
    // (stamps stay in registers until the workgroup's last barrier is behind it: a store in front of a barrier must be
    //  acknowledged before the barrier opens, which would stretch the phases being timed)
    long long ts[6] = {0, 0, 0, 0, 0, 0};
    auto stamp = [&](int slot) { if (my_st) ts[slot] = wall_clock64(); };
    auto flush_stamps = [&](int lo, int hi) {
        if (my_st && threadIdx.x == 0)
            for (int i = lo; i <= hi; ++i) my_st[i] = ts[i];
    };
    const int n_att = Hq * tab.n_groups;                          // attention workgroups: (head, row group), head fastest
    if (my_st && threadIdx.x == 0) {
        const unsigned xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20), hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);
        const bool att = (int)blockIdx.x < n_att;
        my_st[6] = (long long)((xcc & 0xffu) | ((unsigned long long)hw << 8));
        my_st[7] = ((long long)(att ? 0 : 1) << 32) |
                   ((long long)(att ? (int)blockIdx.x % Hq : 2 * ((int)blockIdx.x - n_att)) << 8) | (att ? (int)blockIdx.x / Hq : 0);
        my_st[5] = 0;
    }
    stamp(0);
    if ((int)blockIdx.x < n_att) {
        if (threadIdx.x >= 256) return;                           // (an ended wave is not counted by the barriers below)
        // ---- attention of head blockIdx.x (one row group, whole key range), rows stored write-through ----
        attn_body<T, 128, false, false, true>(qbuf, tab, layer, attn_out, Hq, Hkv, arch, inv_sqrt_d, s_cap, 1, nullptr,
                                              (int)blockIdx.x % Hq, (int)blockIdx.x / Hq, 0, smem, my_st);
        stamp(4);
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");      // this wave's stores have left
        __syncthreads();
        if (threadIdx.x == 0) (void)__hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        stamp(5);
        flush_stamps(0, 0);
        flush_stamps(4, 5);                                       // (1..3 were written by attn_body)
        return;
    }
    // ---- O projection of n-tiles 2 b and 2 b + 1 (b = blockIdx.x - Hq): 4 waves x a quarter of K each, weights first ----
    const int half = (int)(threadIdx.x >> 8), tid4 = (int)(threadIdx.x & 255);
    f32x4 (*red)[1][64] = reinterpret_cast<f32x4 (*)[1][64]>(smem) + half * 4;
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(tid4 >> 6);
    const int KS = K >> 5, ntg_raw = 2 * ((int)blockIdx.x - n_att) + half;
    const bool has_tile = ntg_raw < (N >> 4);
    const int ntg = has_tile ? ntg_raw : 0;
    auto ostamp = [&](int i) { stamp(i); };                      // (thread 0 = first thread of the first n-tile's half)
    const int per = (KS + 3) >> 2;                                // gemm_bf16_stream's quarters (SB = 1)
    const int ks0 = min(KS, wv * per), ks1 = min(KS, ks0 + per), nk = ks1 - ks0, dlast = max(nk - 1, 0);
    const u32x4 *wp = Wo + ((size_t)ntg * KS + min(ks0, KS - 1)) * 64 + lane;
    // Pacing of the weight requests (10 ns ticks): a hold-back before the first request and a pause after every 8 (defaults
    // 3 us / 1 us, SD_AO_DELAY / SD_AO_GAP).  Measured with the stamps (tools/ao_stamps.py, 13b layer, 5 rows x 195 keys,
    // cold caches, profiles/r03_attn_oproj_stamps.txt): requested at once the 52 MB land in 6.5-7.5 us and every
    // attention workgroup arrives at 10.6-15 us instead of ~8 - its K / V loads queue behind the stream; held back 3.5 us,
    // half of the attention workgroups arrive at 9.4-10 us (undisturbed) and the other half at 13-14.7 us, the weights at
    // 8-9 us.  The launch ends at 17.5-19 us against 9.6 + 1.7 + 10.4 for the two launches: 0.15-0.2 ms per verify, not
    // the ~0.35 ms a uniformly undisturbed attention would give.  What delays the second half is the open item.
    const uint2 xpre = (res_ssq && has_tile) ? resid_prefetch<T>(res_x, M, N, ntg, tid4) : uint2{0u, 0u};
    if (delay_ticks > 0) {
        const long long t0 = wall_clock64();
        while (wall_clock64() - t0 < delay_ticks) __builtin_amdgcn_s_sleep(8);
    }
    u32x4 w[AO_NKW];
#pragma unroll
    for (int u = 0; u < AO_NKW; ++u) {                            // (no branch around a load; a k-step past the range is
        w[u] = __builtin_nontemporal_load(wp + (size_t)min(u, dlast) * 64);   //  multiplied as zero below)
        asm volatile("" ::: "memory");
        if ((u & 7) == 7 && gap_ticks > 0 && u + 1 < AO_NKW) {
            const long long t1 = wall_clock64();
            while (wall_clock64() - t1 < gap_ticks) __builtin_amdgcn_s_sleep(4);
        }
    }
    // ---- wait for the Hq attention workgroups of this launch.  The polling wave first waits for its OWN weights: N/16
    // waves polling one word from t = 0 are a request storm on one memory channel that the attention workgroups' K / V
    // loads have to get through (measured: with it attention arrived at 11-17 us instead of ~8, whatever the prefetch did)
    bool timed_out = false;
    if (threadIdx.x < 64) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        ostamp(1);
        const long long t0 = wall_clock64();
        for (;;) {
            const unsigned got = __hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if ((int)(got - want) >= 0) break;
            __builtin_amdgcn_s_sleep(8);
            if (wall_clock64() - t0 > AO_TIMEOUT_TICKS) { timed_out = true; break; }
        }
    }
    __syncthreads();
    asm volatile("" ::: "memory");
    ostamp(2);
    // operand layout (xoff): tile (0, ks) = 512 elements, lane 16 * quad + m holds X[m][32 ks + 8 quad .. + 8); lanes of
    // rows >= M read row 0's fragment (their results land in output columns the epilogue drops)
    const int mrow = (lane & 15) < M ? (lane & 15) : 0;
    const T *xp = attn_out + (size_t)min(ks0, KS - 1) * 512 + ((lane >> 4) * 16 + mrow) * 8;
    auto ldx = [&](int k) -> u32x4 { return *reinterpret_cast<const u32x4 *>(xp + (size_t)min(k, dlast) * 512); };
    u32x4 x[AO_XPF];
#pragma unroll
    for (int u = 0; u < AO_XPF; ++u) x[u] = ldx(u);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < AO_NKW; ++u) {
        u32x4 wu = w[u];
#pragma unroll
        for (int q = 0; q < 4; ++q) wu[q] = u < nk ? wu[q] : 0u;
        acc = mfma16<T>(wu, x[u % AO_XPF], acc);                  // k-steps in order: the streaming kernel's sum
        if (u + AO_XPF < AO_NKW) x[u % AO_XPF] = ldx(u + AO_XPF);
    }
    if (__builtin_amdgcn_readfirstlane((int)timed_out)) {
        acc = f32x4{__uint_as_float(0x7fc00000u), 0.f, 0.f, 0.f};
        if (threadIdx.x == 0) (void)__hip_atomic_fetch_or(wait_status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    red[wv][0][lane] = acc;
    ostamp(3);
    __syncthreads();
    GemmEpiT<T> e = {};
    if (res_ssq) {
        e.res_x = res_x; e.res_h = res_h; e.res_ssq = res_ssq;
        if (has_tile) resid_epilogue_step<T>(red, M, N, ntg, e, tid4, xpre);
    } else if (has_tile) {
        gemm_epilogue_step<1, EPI_PART, 1, 1, T>(red, 0, part, M, 16, N, 0, ntg, e, tid4);
    }
    ostamp(4);
    flush_stamps(0, 4);
}
