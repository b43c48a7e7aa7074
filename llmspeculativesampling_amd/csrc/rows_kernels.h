// Weight-streaming GEMM for 17..128 activation rows (stream-batched verify, SURVEY.md 8(f) rank 1: B streams x (gamma+1) rows
// share one pass over the target's weights; lifts the batch-1 limit of reference speculative_sampling.py:1905).
//
//     part[sb][m][n] = sum_{k in slab sb} X[m][k] * W[n][k]        (or the fused QKV / activation epilogue when SB == 1)
//
// Why a second kernel: gemm_bf16_stream keeps a wave's activation fragments in registers, so at MT m-tiles a wave loads
// MT KiB of X per NTW KiB of W through the same in-order VMEM queue and register file as the weight stream, and to fill
// the chip its workgroups are small - many-row calls then need split-K slabs (no fused epilogue, extra launches).
// Here the grid is ONE workgroup of 16 waves per CU, each given an equal share of the work up front:
//   * n-tiles are dealt to the NG = grid / SB n-groups as evenly as integers allow (ranges differ by at most one tile:
//     Llama-2-13b QKV 960 tiles -> 3 or 4 per CU, gate/up 1728 -> 6 or 7), so SB = 1 fills 256 CUs to 94-98 % and the
//     QKV / SiLU epilogues stay fused; O / down (320 tiles) take SB = 4 (5 tiles x a quarter of K per CU, exact);
//   * COMPUTE wave (ni, kg) owns ONE n-tile (t0 + ni) over one of the nwk k-groups of the workgroup's k-range; its
//     weights go straight to registers (non-temporal, read once) in bursts of one chunk - CH k-steps, CH KiB contiguous -
//     that are multiplied when all of them have landed, and nothing but weight loads sits in its VMEM queue;
//   * LOADER waves (one per k-group while waves are left, else shared) move the k-group's activation chunk - CH k-steps
//     x MT tiles, already in MFMA fragment order - from L2 into a double-buffered LDS panel by LDS-DMA
//     (global_load_lds_dwordx4: no staging registers), one chunk ahead; every compute wave of the k-group reads its
//     fragments from there, so X traffic per CU is MT KiB per k-step however many tiles the CU owns;
//   * one workgroup barrier per chunk hands the panel over (loader: its DMA has landed; compute waves: the previous
//     panel is no longer read).
// After the k-loop the nwk partial accumulators of every tile are folded through LDS and thread group tid / 256 runs the
// tile's epilogue with the streaming kernel's code (gemm_epilogue_step).
#pragma once
#include "model_kernels.h"

#define GR_MAX_TILES 7                                        // n-tiles per workgroup = compute waves per k-group
#define GR_THREADS 1024
#ifndef GR_PROBE
#define GR_PROBE 0
#endif

// one 1 KiB tile L2 -> LDS: lane l's 16 bytes at gsrc go to lds_dst + 16 l (M0 carries the wave-uniform LDS address).
// Default cache policy: the activation panel is re-read by every CU.
__device__ __forceinline__ void gr_glds16(const u32x4 *gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

// WBM: a compute wave's weight burst is WBM panel chunks long (CH * WBM k-steps requested together).  At 5-8 m-tiles the
// double-buffered panel only fits the CU's LDS with CH = 4, and bursts of 4 KiB per wave leave too few bytes in flight
// (8-12 compute waves x 4 KiB against the ~50 KiB a CU needs outstanding to draw its share of HBM): the burst then spans
// two chunks (8 KiB) and is requested on every second one.
template <int MT, int EPI, typename H = bf16_t, int CH = 8, int WBM = 1>
// (WBM is 1 or 2)
__global__ __launch_bounds__(GR_THREADS) void gemm_bf16_rows(const u32x4 *__restrict__ Wp, const u32x4 *__restrict__ Xp,
                                                            float *__restrict__ part, int M, int Mpad, int N, int K,
                                                            int NG, int ks_per_blk, int nwn, int nwk, int nld,
                                                            GemmEpiT<H> e) {
    // GR_PROBE (compile-time, timing experiments only - results are wrong): 1 no activation loads, 2 no barriers,
    // 4 no MFMAs, 8 no weight loads, 16 no fold / epilogue.  Compile-time so that the production inner loop has no branch
    // around a load: with one, hipcc's waitcnt insertion falls back to s_waitcnt vmcnt(0) at the loop's joins and the
    // weight ring drains every chunk (seen in the ISA of the first cut of this kernel).
    constexpr int probe = GR_PROBE;
    constexpr int WB = CH * WBM;
    extern __shared__ __attribute__((aligned(16))) char gr_smem[];
    // activation panel [2][kg][ck][mt][lane]; after the k-loop the same bytes hold the fold buffer [tile][kg][mt][lane]
    u32x4 (*xs)[CH][MT][64] = reinterpret_cast<u32x4 (*)[CH][MT][64]>(gr_smem);      // xs[buf * nwk + kg]
    f32x4 (*red)[MT][64] = reinterpret_cast<f32x4 (*)[MT][64]>(gr_smem);             // red[tile * nwk + kg]
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int NT = N >> 4, KS = K >> 5;
    const int g = blockIdx.x % NG, sb = blockIdx.x / NG;
    const int t0 = (int)((long long)g * NT / NG), t1 = (int)((long long)(g + 1) * NT / NG);
    const int kb0 = sb * ks_per_blk, kb1 = min(KS, kb0 + ks_per_blk);
    const int per = (kb1 - kb0 + nwk - 1) / nwk;                  // k-steps per k-group
    const int nch = (per + CH - 1) / CH;                    // chunks: the same count for every wave (barriers)
    const int ncomp = nwn * nwk;
    const int mt_valid = Mpad >> 4;                               // m-tiles the activation buffer really holds (<= MT)
    auto bar = [&]() { if (!(probe & 2)) __syncthreads(); };

    f32x4 acc[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int ni = wv % nwn, kg = wv / nwn;                       // (meaningful for compute waves)
    const bool comp = wv < ncomp && t0 + ni < t1;

    if (wv >= ncomp && wv < ncomp + nld) {
        // ---------------- loader: k-groups ld, ld + nld, ... ----------------
        const int ld = wv - ncomp;
        auto dma = [&](int c, int buf) {
            if (probe & 1) return;
            for (int k2 = ld; k2 < nwk; k2 += nld) {
                const int ks0 = min(kb1, kb0 + k2 * per), ks1 = min(kb1, ks0 + per);
                if (ks0 >= ks1) continue;                         // (an empty k-group: its compute waves multiply nothing)
                const unsigned base = (unsigned)__builtin_amdgcn_readfirstlane(
                    (int)(unsigned)(uintptr_t)&xs[buf * nwk + k2][0][0][0]);               // LDS byte address (wave-uniform)
#pragma unroll
                for (int ck = 0; ck < CH; ++ck) {
                    // a k-step past the range re-reads the last valid tile: finite values that meet a zero weight operand
                    const int ks = min(ks0 + c * CH + ck, ks1 - 1);
#pragma unroll
                    for (int t = 0; t < MT; ++t)                  // (an m-tile past the buffer re-reads its last one: rows >= M,
                        gr_glds16(Xp + ((size_t)min(t, mt_valid - 1) * KS + ks) * 64 + lane,     //  dropped by the epilogue)
                                  base + (unsigned)((ck * MT + t) * 1024));
                }
            }
        };
        dma(0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        for (int c = 0; c < nch; ++c) {
            bar();                                                // compute(c - 1) is over: panel (c + 1) & 1 is free
            if (c + 1 < nch) {
                dma(c + 1, (c + 1) & 1);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // landed before this wave arrives at the next barrier
            }
        }
    } else if (comp) {
        // ---------------- compute: tile t0 + ni, k-group kg ----------------
        const int ks0 = min(kb1, kb0 + kg * per), ks1 = min(kb1, ks0 + per);
        const int nk = ks1 - ks0;
        const u32x4 *wp = Wp + ((size_t)(t0 + ni) * KS + min(ks0, KS - 1)) * 64 + lane;
        // BURSTS, not a ring: a burst's WB weight tiles (WB KiB, contiguous) are requested together and multiplied when
        // they have all landed; the other compute waves of the CU cover the wait.  Measured against a 4-slot register
        // ring that kept three chunks in flight (tools/gemm_bench.py rows, 40 rows): gate/up 51.5 -> 48.3 us, O 16.2 ->
        // 14.5, QKV 31.4 -> 30.6, down 29.5 -> 29.0 with CH = 8 - the same finding as for the streaming kernel, whose
        // request-all-then-multiply groups beat every prefetching variant tried.
        // A k-step past the wave's range is not requested and multiplied as ZERO against the panel tile it meets (which
        // holds finite values: the loader clamps); every operand register is written on every path.
        u32x4 w[WB];
        auto burst = [&](int c) {
            const int k0 = c * CH;
            if (k0 + WB <= nk) {                                  // a whole burst: WB back-to-back requests, no branch between
#pragma unroll
                for (int u = 0; u < WB; ++u)
                    w[u] = (probe & 8) ? u32x4{0u, 0u, 0u, 0u} : __builtin_nontemporal_load(wp + (size_t)(k0 + u) * 64);
            } else {
#pragma unroll
                for (int u = 0; u < WB; ++u) {
                    w[u] = u32x4{0u, 0u, 0u, 0u};
                    if (k0 + u < nk && !(probe & 8)) w[u] = __builtin_nontemporal_load(wp + (size_t)(k0 + u) * 64);
                }
            }
        };
        // the FIRST burst does not wait for the panel: weights do not depend on the activations, so they are requested in front
        // of the first barrier and travel while the loader's first chunk lands (the launch's fill: one memory latency, not two)
        burst(0);
        for (int c = 0; c < nch; ++c) {
            bar();                                                // chunk c's panel is in LDS
            if (c > 0 && c % WBM == 0) burst(c);
            const int pbuf = (c & 1) * nwk + kg;
            const int wb0 = (c % WBM) * CH;
#pragma unroll
            for (int ck = 0; ck < CH; ++ck) {
                u32x4 xf[MT];
#pragma unroll
                for (int t = 0; t < MT; ++t) xf[t] = (probe & 1) ? u32x4{0u, 0u, 0u, 0u} : xs[pbuf][ck][t][lane];
                u32x4 wk = w[ck];                                 // this chunk's part of the burst
                if constexpr (WBM == 2) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) wk[q] = wb0 ? w[CH + ck][q] : wk[q];
                }
                if (!(probe & 4)) {
#pragma unroll
                    for (int t = 0; t < MT; ++t) acc[t] = mfma16<H>(wk, xf[t], acc[t]);
                } else {
                    acc[0][0] += __uint_as_float((wk[0] ^ xf[0][0]) & 1u);   // (keeps the loads alive)
                }
            }
        }
    } else {
        for (int c = 0; c < nch; ++c) bar();                      // idle wave: keeps the barrier count
    }
    // ---- fold the k-groups' accumulators through LDS, then the epilogue: the workgroup's threads are dealt (tile, m-tile,
    // lane) items - EL per tile - and each runs the streaming kernel's epilogue code for its item.  The nwk partial sums of
    // a tile are added as ((k0 + k1) + (k2 + k3)) with absent k-groups as zero - the streaming kernel's fold of four waves
    if (probe & 16) { if (acc[0][0] == 123.456f) part[0] = 1.f; return; }
    __syncthreads();
    if (comp) {
#pragma unroll
        for (int t = 0; t < MT; ++t) red[ni * nwk + kg][t][lane] = acc[t];
    }
    __syncthreads();
    constexpr int EL = (EPI == EPI_ACT_SILU ? 32 : 64) * MT;      // threads one tile's epilogue takes (SiLU pairs gate / up lanes)
    for (int i = (int)threadIdx.x; i < nwn * EL; i += GR_THREADS) {
        const int g4 = i / EL, tile = t0 + g4;
        if (tile >= t1) continue;
        f32x4 (*rt)[MT][64] = red + (size_t)g4 * nwk;
        auto folded = [&](int pp, int l) -> f32x4 {
            const f32x4 z = {0.f, 0.f, 0.f, 0.f};
            const f32x4 s0 = rt[0][pp][l], s1 = nwk > 1 ? rt[1][pp][l] : z, s2 = nwk > 2 ? rt[2][pp][l] : z,
                        s3 = nwk > 3 ? rt[3][pp][l] : z;
            return (s0 + s1) + (s2 + s3);
        };
        gemm_epilogue_fold<MT, EPI, 1, MT, H>(folded, 0, part, M, Mpad, N, sb, tile, e, i - g4 * EL);
    }
}
