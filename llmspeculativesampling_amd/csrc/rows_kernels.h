// Weight-streaming GEMM for 17..64 activation rows (stream-batched verify, SURVEY.md 8(f) rank 1: B streams x (gamma+1) rows
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
//   * inside a workgroup wave (kq, nw) = (k-quarter, n-wave) owns tiles t0 + nw and t0 + nw + 4 over a quarter of the
//     workgroup's k-range - the streaming kernel's quarters and k order, so a dot product is accumulated in the same
//     order and the fold through LDS gives the same bits;
//   * the activation chunk of a k-quarter (GR_CH k-steps x MT tiles, already in MFMA fragment order) is fetched ONCE by
//     the quarter's four n-waves and shared through LDS: X traffic per CU is MT KiB per k-step instead of MT per wave;
//   * weights go straight to registers (non-temporal, read once), two chunks ahead of the MFMAs that use them.
// One workgroup barrier per chunk (every ~2.6 us of weight stream per CU).
#pragma once
#include "model_kernels.h"

#define GR_CH 2                                               // k-steps per chunk
#define GR_RW 3                                               // register slots of the weight ring (2 chunks in flight)
#define GR_MAX_TILES 8                                        // n-tiles per workgroup (2 per n-wave)
#define GR_THREADS 1024

template <int MT, int EPI, typename H = bf16_t>
__global__ __launch_bounds__(GR_THREADS) void gemm_bf16_rows(const u32x4 *__restrict__ Wp, const u32x4 *__restrict__ Xp,
                                                            float *__restrict__ part, int M, int Mpad, int N, int K,
                                                            int NG, int ks_per_blk, GemmEpiT<H> e, int probe = 0) {
    // probe (SD_ROWS_PROBE, timing experiments only - results are wrong): 1 no activation loads, 2 no barriers, 4 no MFMAs
    constexpr int NXT = MT * GR_CH;                               // activation tiles per chunk and k-quarter
    constexpr int XW = (NXT + 3) / 4;                             // of which one n-wave fetches at most XW
    extern __shared__ __attribute__((aligned(16))) char gr_smem[];
    // activation ring [2][kq][ck][mt][lane]; after the k-loop the same bytes hold the fold buffer [tile 0..3][kq][mt][lane]
    u32x4 (*xs)[4][GR_CH][MT][64] = reinterpret_cast<u32x4 (*)[4][GR_CH][MT][64]>(gr_smem);
    f32x4 (*red)[4][MT][64] = reinterpret_cast<f32x4 (*)[4][MT][64]>(gr_smem);
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int kq = wv >> 2, nw = wv & 3;
    const int NT = N >> 4, KS = K >> 5;
    const int g = blockIdx.x % NG, sb = blockIdx.x / NG;
    const int t0 = (int)((long long)g * NT / NG), t1 = (int)((long long)(g + 1) * NT / NG);
    const int kb0 = sb * ks_per_blk, kb1 = min(KS, kb0 + ks_per_blk);
    const int per = (kb1 - kb0 + 3) >> 2;                         // gemm_bf16_stream's quarters
    const int ks0 = min(kb1, kb0 + kq * per), ks1 = min(kb1, ks0 + per);
    const int nch = (per + GR_CH - 1) / GR_CH;                    // chunks: the same count for every wave (barriers)
    const bool has[2] = {t0 + nw < t1, t0 + nw + 4 < t1};
    const u32x4 *wp[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) wp[j] = Wp + ((size_t)(has[j] ? t0 + nw + 4 * j : 0) * KS + ks0) * 64 + lane;
    const u32x4 *xp = Xp + (size_t)ks0 * 64 + lane;              // tile (mt, ks0 + d) at xp + (mt * KS + d) * 64

    f32x4 acc[2][MT];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int t = 0; t < MT; ++t) acc[j][t] = f32x4{0.f, 0.f, 0.f, 0.f};
    // every operand register is written on every path (a skipped load leaves zeros, never stale bits: DESIGN.md section 7)
    u32x4 wr[GR_RW][2][GR_CH];
    u32x4 xr[2][XW];
    const u32x4 zero = {0u, 0u, 0u, 0u};

    auto issue_w = [&](auto slot_c, int c) {
        constexpr int slot = decltype(slot_c)::value;
#pragma unroll
        for (int ck = 0; ck < GR_CH; ++ck) {
            const int d = c * GR_CH + ck;
            const bool ok = ks0 + d < ks1;
#pragma unroll
            for (int j = 0; j < 2; ++j)
                wr[slot][j][ck] = (ok && has[j]) ? __builtin_nontemporal_load(wp[j] + (size_t)d * 64) : zero;
        }
    };
    auto issue_x = [&](auto slot_c, int c) {
        constexpr int slot = decltype(slot_c)::value;
#pragma unroll
        for (int j = 0; j < XW; ++j) {
            const int i = nw + 4 * j, ck = i / MT, mt = i - ck * MT, d = c * GR_CH + ck;
            const bool ok = i < NXT && ks0 + d < ks1 && mt * 16 < Mpad && !(probe & 1);
            xr[slot][j] = ok ? xp[((size_t)mt * KS + d) * 64] : zero;
        }
    };
    auto put_x = [&](auto slot_c, int buf) {
        constexpr int slot = decltype(slot_c)::value;
#pragma unroll
        for (int j = 0; j < XW; ++j) {
            const int i = nw + 4 * j, ck = i / MT, mt = i - ck * MT;
            if (i < NXT) xs[buf][kq][ck][mt][lane] = xr[slot][j];
        }
    };
    auto compute = [&](auto slot_c, int buf) {
        constexpr int slot = decltype(slot_c)::value;
#pragma unroll
        for (int ck = 0; ck < GR_CH; ++ck) {
            u32x4 xf[MT];
#pragma unroll
            for (int t = 0; t < MT; ++t) xf[t] = xs[buf][kq][ck][t][lane];
#pragma unroll
            for (int j = 0; j < 2; ++j)
                if (has[j] && !(probe & 4)) {
#pragma unroll
                    for (int t = 0; t < MT; ++t) acc[j][t] = mfma16<H>(wr[slot][j][ck], xf[t], acc[j][t]);
                } else if (probe & 4) {
                    acc[j][0][0] += __uint_as_float(wr[slot][j][ck][0] & 1u);     // (keeps the loads alive)
                }
        }
    };
    using std::integral_constant;
    issue_w(integral_constant<int, 0>{}, 0);
    issue_x(integral_constant<int, 0>{}, 0);
    issue_w(integral_constant<int, 1>{}, 1);
    issue_x(integral_constant<int, 1>{}, 1);
    put_x(integral_constant<int, 0>{}, 0);
    // step I of six (lcm of the 3 weight slots and the 2 activation slots / buffers): chunk c = base + I
    auto step = [&](auto I_c, int c) {
        constexpr int I = decltype(I_c)::value;
        if (!(probe & 2)) __syncthreads();                        // chunk c's activations are in LDS; nobody reads c - 1 any more
        issue_w(integral_constant<int, (I + 2) % GR_RW>{}, c + 2);
        issue_x(integral_constant<int, I % 2>{}, c + 2);          // (chunk c's staging registers were stored a step ago)
        compute(integral_constant<int, I % GR_RW>{}, I % 2);
        put_x(integral_constant<int, (I + 1) % 2>{}, (I + 1) % 2);   // chunk c + 1 -> the buffer chunk c - 1 used
    };
    for (int base = 0; base < nch; base += 6) {
        step(integral_constant<int, 0>{}, base);
        if (base + 1 < nch) step(integral_constant<int, 1>{}, base + 1);
        if (base + 2 < nch) step(integral_constant<int, 2>{}, base + 2);
        if (base + 3 < nch) step(integral_constant<int, 3>{}, base + 3);
        if (base + 4 < nch) step(integral_constant<int, 4>{}, base + 4);
        if (base + 5 < nch) step(integral_constant<int, 5>{}, base + 5);
    }
    // Fold the four k-quarters through LDS, the tiles of one n-wave slot (j) at a time; thread group g4 = tid / 256 then
    // runs the epilogue of tile t0 + g4 + 4 j exactly as a streaming-kernel workgroup would (same fold order, same code).
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        __syncthreads();
#pragma unroll
        for (int t = 0; t < MT; ++t) red[nw][kq][t][lane] = acc[j][t];               // red[tile nw][kq][mt][lane]
        __syncthreads();
        const int g4 = threadIdx.x >> 8, tile = t0 + g4 + 4 * j;
        if (tile < t1)
            gemm_epilogue_step<MT, EPI, 1, MT, H>(red[g4], 0, part, M, Mpad, N, sb, tile, e, (int)(threadIdx.x & 255));
    }
}
