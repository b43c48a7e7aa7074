// Decode-step kernels for SMALL models (the draft: llama-68m / opt-125m, hidden <= 2048) on 1..4 new rows.
// Reference: the per-token forward of sampling/kvcache_model.py:206-214 -> modeling_llama.py:405-457 /
// modeling_opt.py:303-378, which the reference runs as ~10 eager launches per layer.
//
// At hidden 768 a whole normalised row is 1.5 KB, so the residual add + RMSNorm / LayerNorm that used to be its own
// launch between two GEMMs is recomputed by EVERY workgroup of the consuming GEMM as a prologue (a few KB of L2 reads
// per workgroup), while that workgroup's weight tiles - requested before the prologue - are already in flight.  A
// decoder layer is then 5 dependent launches (QKV, attention, O, gate/up, down) instead of 7, the model's tail 2
// (lm_head with tile maxima, sampler) instead of 4, and no launch exists only to normalise.
//
// Arithmetic (operation order, rounding points, the thread -> column maps that fix the fp32 summation order of the
// norm statistics) is that of embed_norm_kernel / residual_norm_kernel in model_kernels.h, so both routes give
// bit-identical activations; tests/test_gpu_native_parity.py checks it.
#pragma once
#include "model_kernels.h"

enum { PRO_TILED = 0, PRO_EMBED = 1, PRO_RESID = 2 };

struct SmallPro {
    // PRO_EMBED: x = embedding (+ OPT learned position); R_out <- x; operand = norm(x)
    const bf16_t *embed, *pos_embed;
    int pos_off, vocab;
    // PRO_RESID: x' = rnd(R_in + rnd(sum_s slab[s] + bias)); R_out <- x'; operand = norm(x')
    const float *slab;
    int S;
    size_t stride_s;
    const bf16_t *bias;
    const bf16_t *r_in;
    bf16_t *r_out;            // written by workgroup 0 only (NULL: the head, nothing reads the stream afterwards)
    const bf16_t *nw, *nb;
    float eps;
    int kind, H, rn_threads;  // rn_threads: blockDim of the residual_norm_kernel launch this prologue mirrors
};

#define SMALL_MAX_ROWS 4
#define SMALL_XPAD 8          // bf16 elements of padding per LDS row

// C[m][n] = sum_k X[m][k] * W[n][k] for M <= 4 rows: one workgroup = one 16-column n-tile x one k-slab, 4 waves on a
// quarter of the slab each (as gemm_bf16_stream<1,.,.,1>); the operand rows come from the prologue through LDS
// (PRO_EMBED / PRO_RESID) or from the tile-layout activation buffer (PRO_TILED).
template <int PRO, int EPI>
__global__ __launch_bounds__(256) void gemm_small(const u32x4 *__restrict__ Wp, const bf16_t *__restrict__ X,
                                                 float *__restrict__ part, int M, int N, int K, int SB,
                                                 int ks_per_blk, GemmEpi e, SmallPro p) {
#pragma clang fp contract(off)                                    // see norm_row (model_kernels.h)
    constexpr int KSW = 8;                                        // k-steps of weights a wave keeps in flight
    __shared__ f32x4 red[4][1][64];
    extern __shared__ __attribute__((aligned(16))) char dyn_smem[];
    bf16_t *Xs = reinterpret_cast<bf16_t *>(dyn_smem);           // [M][K + SMALL_XPAD]
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int NTG = N >> 4, KS = K >> 5;
    const int sb = blockIdx.x / NTG, ntg = blockIdx.x - sb * NTG;
    const int kb0 = sb * ks_per_blk, kb1 = min(KS, kb0 + ks_per_blk);
    const int per = (kb1 - kb0 + 3) >> 2;
    const int ks0 = min(kb1, kb0 + wv * per), ks1 = min(kb1, ks0 + per);
    const u32x4 *wt = Wp + (size_t)ntg * KS * 64 + lane;          // this n-tile's k-step 0
    const int mrow = lane & 15, kq = (lane >> 4) * 8;
    const bool mv = mrow < M;

    // No branch around a load or an MFMA in this kernel.  With `if (ks < ks1)` around each of the KSW requests and
    // multiplications (per-lane-looking bounds: wv was tid >> 6) hipcc built divergent-branch ladders: the PRO_RESID
    // instances with the SiLU / ReLU / head / slab epilogues and PRO_TILED + slab came out at 254 VGPRs + 44 AGPRs, one
    // wave per SIMD (-Rpass-analysis=kernel-resource-usage), which is what rounds 2-4 measured as "the prologue route is
    // slower" (gate/up 15.9 us, lm_head 48 us).  Here a k-step past the wave's range re-requests k-step `kc` (clamped into
    // the matrix) and meets a zeroed weight fragment; the wave id is scalar, so the loop bounds are too.
    auto kc = [&](int ks) { return min(max(ks, 0), KS - 1); };
    // the weights do not depend on the prologue: request this wave's first KSW tiles now
    u32x4 w[KSW];
#pragma unroll
    for (int u = 0; u < KSW; ++u) w[u] = __builtin_nontemporal_load(wt + (size_t)kc(min(ks0 + u, ks1 - 1)) * 64);

    if constexpr (PRO == PRO_EMBED) {
        // embed_norm_kernel's arithmetic: thread t owns columns t, t + 256, ...
        float *redf = reinterpret_cast<float *>(&red[0][0][0]);
        const int H = p.H;
        for (int row = 0; row < M; ++row) {
            const int pos = tab_pos(e.tab, row);
            const int tok = min(max(tab_tok(e.tab, row, pos), 0), p.vocab - 1);
            const bf16_t *src = p.embed + (size_t)tok * H;
            const bf16_t *ps = p.pos_embed ? p.pos_embed + (size_t)(pos + p.pos_off) * H : nullptr;
            float v[8];
            float a = 0.f, a2 = 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int i = tid + j * 256;
                if (i < H) {
                    float x = to_f(src[i]);
                    if (ps) x = rnd<bf16_t>(x + to_f(ps[i]));
                    v[j] = x;
                    if (blockIdx.x == 0 && p.r_out) p.r_out[(size_t)row * H + i] = (bf16_t)x;
                    a += x;
                    a2 += x * x;
                }
            }
            if (p.kind == NORM_RMS) {
                const float r = rsqrtf(block_sum(a2, redf) / (float)H + p.eps);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int i = tid + j * 256;
                    if (i < H) Xs[(size_t)row * (K + SMALL_XPAD) + i] = (bf16_t)(to_f(p.nw[i]) * rnd<bf16_t>(v[j] * r));
                }
            } else {
                const float mean = block_sum(a, redf) / (float)H;
                float d2 = 0.f;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int i = tid + j * 256;
                    if (i < H) { const float d = v[j] - mean; d2 += d * d; }
                }
                const float r = 1.0f / sqrtf(block_sum(d2, redf) / (float)H + p.eps);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int i = tid + j * 256;
                    if (i < H) Xs[(size_t)row * (K + SMALL_XPAD) + i] = (bf16_t)((v[j] - mean) * r * to_f(p.nw[i]) + to_f(p.nb[i]));
                }
            }
        }
        __syncthreads();
    } else if constexpr (PRO == PRO_RESID) {
        // residual_norm_kernel's arithmetic: its launch has rn_threads threads, thread t owns the column groups
        // 4*t and 4*(t + rn_threads); the threads beyond rn_threads idle here and add 0 to the block sums
        float *redf = reinterpret_cast<float *>(&red[0][0][0]);
        const int H = p.H, RT = p.rn_threads;
        for (int row = 0; row < M; ++row) {
            const int src_row = e.use_xmap ? (int)e.tab.xmap[row] : row;
            float v[RN_RG][4], wv4[RN_RG][4], bv4[RN_RG][4];
            bool on[RN_RG];
            float a = 0.f, a2 = 0.f;
#pragma unroll
            for (int g = 0; g < RN_RG; ++g) {
                const int i = (tid + g * RT) * 4;
                on[g] = tid < RT && i < H;
                if (on[g]) {
                    const f32x4 y4 = reduce_part4(p.slab, p.S, p.stride_s, (size_t)src_row * H + i);
                    float xin[4], bi[4] = {0.f, 0.f, 0.f, 0.f};
                    load4<bf16_t>(p.r_in + (size_t)src_row * H + i, xin);
                    if (p.bias) load4<bf16_t>(p.bias + i, bi);
                    load4<bf16_t>(p.nw + i, wv4[g]);
                    if (p.kind == NORM_LN) load4<bf16_t>(p.nb + i, bv4[g]);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        float y = y4[j];
                        if (p.bias) y += bi[j];
                        v[g][j] = rnd<bf16_t>(xin[j] + rnd<bf16_t>(y));
                        a += v[g][j];
                        a2 += v[g][j] * v[g][j];
                    }
                    if (blockIdx.x == 0 && p.r_out) store4t<bf16_t>(p.r_out + (size_t)src_row * H + i, v[g]);
                }
            }
            float mean = 0.f, r;
            if (p.kind == NORM_RMS) {
                r = rsqrtf(block_sum(a2, redf) / (float)H + p.eps);
            } else {
                mean = block_sum(a, redf) / (float)H;
                float d2 = 0.f;
#pragma unroll
                for (int g = 0; g < RN_RG; ++g)
                    if (on[g])
#pragma unroll
                        for (int j = 0; j < 4; ++j) { const float d = v[g][j] - mean; d2 += d * d; }
                r = 1.0f / sqrtf(block_sum(d2, redf) / (float)H + p.eps);
            }
#pragma unroll
            for (int g = 0; g < RN_RG; ++g) {
                if (!on[g]) continue;
                const int i = (tid + g * RT) * 4;
                float o[4];
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    o[j] = p.kind == NORM_RMS ? rnd<bf16_t>(wv4[g][j] * rnd<bf16_t>(v[g][j] * r))
                                              : rnd<bf16_t>((v[g][j] - mean) * r * wv4[g][j] + bv4[g][j]);
                store4(Xs + (size_t)row * (K + SMALL_XPAD) + i, o[0], o[1], o[2], o[3]);
            }
        }
        __syncthreads();
    }

    // ---- the dot products: B fragment = X[m = lane & 15][32 * ks + 8 * (lane >> 4) .. + 8)
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const bf16_t *xg = nullptr;
    // rows >= M read row 0's operand (finite values; their output columns are dropped by the epilogue)
    const int mr = mv ? mrow : 0;
    if constexpr (PRO == PRO_TILED) {
        const int msrc = e.use_xmap ? (int)e.tab.xmap[mr] : mr;
        xg = X + (size_t)(msrc >> 4) * KS * 512 + ((lane >> 4) * 16 + (msrc & 15)) * 8;
    }
    for (int base = ks0; base < ks1; base += KSW) {
        if (base != ks0) {
#pragma unroll
            for (int u = 0; u < KSW; ++u) w[u] = __builtin_nontemporal_load(wt + (size_t)kc(min(base + u, ks1 - 1)) * 64);
        }
        u32x4 x[KSW];
#pragma unroll
        for (int u = 0; u < KSW; ++u) {
            const int ks = kc(min(base + u, ks1 - 1));
            if constexpr (PRO == PRO_TILED) x[u] = *reinterpret_cast<const u32x4 *>(xg + (size_t)ks * 512);
            else x[u] = *reinterpret_cast<const u32x4 *>(Xs + (size_t)mr * (K + SMALL_XPAD) + ks * 32 + kq);
        }
#pragma unroll
        for (int u = 0; u < KSW; ++u) {
            u32x4 wz = w[u];
#pragma unroll
            for (int q = 0; q < 4; ++q) wz[q] = base + u < ks1 ? wz[q] : 0u;
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wz), __builtin_bit_cast(bf16x8, x[u]), acc, 0, 0, 0);
        }
    }
    red[wv][0][lane] = acc;
    __syncthreads();
    gemm_epilogue_step<1, EPI, 1, 1>(red, 0, part, M, 16, N, sb, ntg, e);
}
