// Many-row GEMM for prefill passes of 145..256 rows (and any shape the balanced kernel does not plan): the nn.Linear
// projections of a decoder layer (reference sampling/models/modeling_llama.py:292-393, 405-457; modeling_opt.py:303-378) over
// the rows of a prompt, as the reference's no-cache forward feeds them (kvcache_model.py:156):
//
//     part[sb][m][n] = sum_{k in slab sb} X[m][k] * W[n][k]        (or, with ONE slab, the fused QKV / SiLU / ReLU epilogue)
//
// gemm_bf16_tiled (model_kernels.h) moves every tile global -> registers -> LDS and synchronises the workgroup around each
// 64-column k-step with nothing in flight across the barrier: 610-650 TFLOP/s at 256 rows (a quarter of the dense bf16
// peak).  This kernel keeps the same 1 KiB fragment tiles of W and X (both operands are stored in MFMA fragment order, so
// one global_load_lds_dwordx4 per wave moves one tile and a lane's ds_read_b128 at 16 * lane is conflict-free), and changes
// the pipeline:
//   * 512 threads = 8 waves as 4 (m) x 2 (n) on a block of BMT = 4 * MTW m-tiles x 8 n-tiles (256 or 128 rows x 128
//     columns), every wave a (MTW x 4)-tile quadrant: 4 + MTW fragment reads feed 4 * MTW MFMAs per 32 columns of K;
//   * a k-stage = KT k-tiles = KT * (8 + BMT) tiles, copied L2 / HBM -> LDS by LDS-DMA (no staging registers, no
//     ds_write), every wave issuing an equal share so that one counted s_waitcnt covers a stage;
//   * NBUF stage buffers, NBUF - 1 stages in flight: per stage  wait for the own share of stage s (vmcnt leaves the younger
//     stages outstanding) -> ONE raw s_barrier (stage s is complete in LDS, nobody reads stage s - 1's buffer any more) ->
//     issue stage s + NBUF - 1 into that buffer -> multiply stage s.  The DMA queue never drains inside the loop
//     (cdna_hip_programming.md section 5, "Pipelining across barriers"; MI355X_MICROARCH.md, two waves per SIMD, item 7:
//     LDS-DMA stays in flight across s_barrier, a ds_read is ordered behind it only by the issuing wave's vmcnt + a barrier);
//   * STAG: waves w and w + 4 share a SIMD and, running the same program behind the same barrier, reach their tile
//     requests, their fragment reads and their MFMAs together.  Waves 4-7 therefore run half a beat behind: they keep a
//     stage's fragments in registers across the barrier and multiply them at the START of the next interval, while waves
//     0-3 request and read; then 0-3 multiply while 4-7 request and read (MI355X_MICROARCH.md, two waves per SIMD, item 9);
//     (+7-10 % at 256 rows, bit-identical sums);
//   * ROT (measured: no gain, off): the blocks that share an XCD start at different points of their k-range and walk it
//     cyclically, so that the ~30 CUs of an XCD do not all ask the L2 for the same activation lines in the same microsecond.  (The fp32 sums of a
//     block then start at a different k: same values up to rounding order; the integer-exact tests hold either way.)
// The weight tiles of a block whose rows are ALL in the block (one m-block) are read once chip-wide: non-temporal; with two
// m-blocks the second reader should find them in L2 / the Infinity Cache: default policy (WNT = false).
#pragma once
#include "model_kernels.h"

#define MM_THREADS 512

// one 1 KiB tile -> LDS: lane l's 16 bytes at gsrc go to lds_dst + 16 l (M0 carries the wave-uniform LDS address)
template <bool NT>
__device__ __forceinline__ void mm_glds16(const u32x4 *gsrc, unsigned lds_dst) {
    unsigned keep;
    if constexpr (NT)
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
    else
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

// wait until at most n of this wave's requests are outstanding (the immediate must be a literal)
__device__ __forceinline__ void mm_wait_stages(int n) {
    switch (n) {
        case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
        case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
        case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
        case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
        case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
        case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
        case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
        case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
        case 9: asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); break;
        case 10: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
        case 12: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
        case 15: asm volatile("s_waitcnt vmcnt(15)" ::: "memory"); break;
        case 16: asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); break;
        case 18: asm volatile("s_waitcnt vmcnt(18)" ::: "memory"); break;
        case 24: asm volatile("s_waitcnt vmcnt(24)" ::: "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;       // (never wrong, only early)
    }
}

template <int MTW, int EPI = EPI_PART, typename H = bf16_t, bool WNT = true, int KT = 2, int NBUF = 3, bool STAG = true, bool ROT = false>
__global__ __launch_bounds__(MM_THREADS) void gemm_bf16_mm(const u32x4 *__restrict__ Wp, const u32x4 *__restrict__ Xp,
                                                          float *__restrict__ part, int M, int Mpad, int N, int K, int SB,
                                                          int ks_per_blk, GemmEpiT<H> e) {
    constexpr int WT = 8, BMT = 4 * MTW, TT = WT + BMT;           // tiles per k-tile column: W, X, total
    constexpr int NL = TT * KT;                                   // tiles (KiB) per stage
    constexpr int XPW = BMT / 8;                                  // X tiles per wave and k-tile
    constexpr int LPS = KT * (1 + XPW);                           // tiles each wave copies per stage
    static_assert(NBUF >= 3 && NBUF * NL <= 144, "stage buffers: at most 144 KiB of the CU's LDS");
    extern __shared__ __attribute__((aligned(16))) char mm_smem[];            // [NBUF][NL][64] u32x4
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int wn = wv & 1, wm = wv >> 1;
    const int KS = K >> 5, NB = (N >> 4) / WT, MB = ((Mpad >> 4) + BMT - 1) / BMT;
    int b = blockIdx.x;
    const int mb = b % MB; b /= MB;                               // m-blocks of one n-block are neighbours (W reuse in L2 / MALL)
    const int nb = b % NB, sb = b / NB;
    const int nt0 = nb * WT, mt0 = mb * BMT, mt_end = Mpad >> 4;
    const int kb0 = sb * ks_per_blk, kb1 = min(KS, kb0 + ks_per_blk);         // ks_per_blk and KS are multiples of KT
    const int nst = (kb1 - kb0) / KT;                             // stages of this block (>= 1: the plan leaves no empty slab)
    // blocks b, b + 8, ... share an XCD (round-robin dispatch: a speed assumption only)
    const int rot = ROT ? (int)(((blockIdx.x >> 3) * (unsigned)max(1, nst / 32)) % (unsigned)nst) : 0;

    // A stage's tiles lie in LDS as [kk][tt] (k-tile inside the stage, then tt < WT: W tile nt0 + tt, else X tile
    // mt0 + tt - WT), 1 KiB each.  Wave wv copies, per stage, W tile wv of every k-tile (r < KT) and X tiles wv, wv + 8, ...
    // of every k-tile (r >= KT) - which operand a request is for is a compile-time property of r, so the weight tiles'
    // cache policy costs no branch.  An m-tile past the buffer re-reads its last one (rows >= M, never stored).
    const u32x4 *src[LPS];
    unsigned slot[LPS];
    bool have[LPS];                                               // (wave-uniform) an X tile past the buffer is not copied at all
    int my_lps = 0;                                               // requests this wave makes per stage: its vmcnt unit
#pragma unroll
    for (int r = 0; r < LPS; ++r) {
        if (r < KT) {
            slot[r] = (unsigned)(r * TT + wv);
            src[r] = Wp + ((size_t)(nt0 + wv) * KS + kb0 + r) * 64 + lane;
            have[r] = true;
        } else {
            const int q = r - KT, kk = q / XPW, xi = wv + 8 * (q % XPW);
            slot[r] = (unsigned)(kk * TT + WT + xi);
            have[r] = mt0 + xi < mt_end;
            src[r] = Xp + ((size_t)min(mt0 + xi, mt_end - 1) * KS + kb0 + kk) * 64 + lane;
        }
        my_lps += have[r] ? 1 : 0;
    }
    const unsigned lds0 = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(uintptr_t)mm_smem);
    auto issue = [&](int stage, int buf) {                        // this wave's slots of k-stage `stage` -> buffer buf
        int st = stage + rot;
        st = st >= nst ? st - nst : st;
        const size_t off = (size_t)st * KT * 64;
#pragma unroll
        for (int r = 0; r < LPS; ++r) {
            const unsigned dst = lds0 + ((unsigned)(buf * NL) + slot[r]) * 1024u;
            if (r < KT) {
                if constexpr (WNT) mm_glds16<true>(src[r] + off, dst); else mm_glds16<false>(src[r] + off, dst);
            } else if (have[r]) {
                mm_glds16<false>(src[r] + off, dst);
            }
        }
    };

    f32x4 acc[4][MTW];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int t = 0; t < MTW; ++t) acc[j][t] = f32x4{0.f, 0.f, 0.f, 0.f};
    const bool rows_here = mt0 + wm * MTW < mt_end;               // (a wave whose m-tiles are all past the buffer multiplies nothing)
    const bool late = STAG && wv >= 4;

    u32x4 wf[KT][4], xf[KT][MTW];
#pragma unroll
    for (int kk = 0; kk < KT; ++kk) {
#pragma unroll
        for (int j = 0; j < 4; ++j) wf[kk][j] = u32x4{0u, 0u, 0u, 0u};
#pragma unroll
        for (int t = 0; t < MTW; ++t) xf[kk][t] = u32x4{0u, 0u, 0u, 0u};
    }
    auto mul = [&]() {
#pragma unroll
        for (int kk = 0; kk < KT; ++kk)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int t = 0; t < MTW; ++t) acc[j][t] = mfma16<H>(wf[kk][j], xf[kk][t], acc[j][t]);
    };

#pragma unroll
    for (int p = 0; p < NBUF - 1; ++p)
        if (p < nst) issue(p, p);
    int cur = 0, nxt = NBUF - 1;                                  // buffer of stage s, buffer of stage s + NBUF - 1
    for (int s = 0; s < nst; ++s) {
        mm_wait_stages(min(NBUF - 2, nst - 1 - s) * my_lps);      // own share of stage s has landed
        // (the late half carries stage s - 1's fragment reads across this barrier in flight; the buffer they read is the one the
        //  next requests overwrite - the DMA lands hundreds of cycles later than an LDS read completes, but nothing ORDERS the
        //  two, so the reads are retired here: by now they have had the whole wait above to finish)
        if constexpr (STAG) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                             // ... everybody's; and stage s - 1's buffer is free
        asm volatile("" ::: "memory");
        if (STAG && late && rows_here) mul();                     // stage s - 1 (zeros at s = 0)
        if (s + NBUF - 1 < nst) issue(s + NBUF - 1, nxt);
        const u32x4 *sm = reinterpret_cast<const u32x4 *>(mm_smem) + (size_t)cur * NL * 64;
#pragma unroll
        for (int kk = 0; kk < KT; ++kk) {
#pragma unroll
            for (int j = 0; j < 4; ++j) wf[kk][j] = sm[(kk * TT + wn * 4 + j) * 64 + lane];
#pragma unroll
            for (int t = 0; t < MTW; ++t) xf[kk][t] = sm[(kk * TT + WT + wm * MTW + t) * 64 + lane];
        }
        if (!late && rows_here) mul();
        cur = cur == NBUF - 1 ? 0 : cur + 1;
        nxt = nxt == NBUF - 1 ? 0 : nxt + 1;
    }
    if (STAG && late && rows_here) mul();
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int t = 0; t < MTW; ++t) {
            const int mtile = mt0 + wm * MTW + t, ntile = nt0 + wn * 4 + j;
            if constexpr (EPI == EPI_PART) {
                const int m = mtile * 16 + (lane & 15), n = ntile * 16 + (lane >> 4) * 4;
                if (m < M) *reinterpret_cast<f32x4 *>(part + ((size_t)sb * Mpad + m) * N + n) = acc[j][t];
            } else {
                // SB == 1: the block holds the whole dot product - the streaming kernels' fused epilogue on this tile's
                // accumulator (bias + RoPE / q-scale + KV append, SiLU(gate) * up, ReLU: gemm_epilogue_fold).  The SiLU form
                // pairs lane l's gate columns with lane l + 32's up columns: fetched with a full-wave shuffle up front.
                const f32x4 mine = acc[j][t];
                f32x4 other = mine;
                if constexpr (EPI == EPI_ACT_SILU) {
#pragma unroll
                    for (int c = 0; c < 4; ++c) other[c] = __shfl_xor(mine[c], 32, 64);
                }
                auto folded = [&](int, int l) -> f32x4 { return l == lane ? mine : other; };
                if (mtile < mt_end)                              // (wave-uniform)
                    gemm_epilogue_fold<64, EPI, 1, 1, H>(folded, mtile, (float *)nullptr, M, Mpad, N, 0, ntile, e, lane);
            }
        }
}
