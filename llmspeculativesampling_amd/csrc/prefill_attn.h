// Causal attention of a prefill pass (rows that are consecutive positions of a stream; reference modeling_llama.py:346-372,
// modeling_opt.py:210-256): 16 rows of one head per workgroup, P.V on the matrix cores.
//
// attn_kernel (model_kernels.h) is built for a decode / verify step: <= 8 rows per workgroup, the scores by MFMA, P.V on the
// vector ALUs (each thread 8 head dims of 8 rows per key).  A 256-row prefill pass runs 32 of its row groups per head -
// 2.7 ms of a 12.4 ms pass at the 13b shape, every group re-reading the head's keys and values.  Here a workgroup takes 16
// rows (one full MFMA row tile), and both products run on the matrix cores:
//   * scores: A = 16 keys x 32 dims straight from the arena, B = q^T - attn_kernel's code, the same roundings
//     (rnd(q.k), Llama: rnd(. / sqrt(D)); masked entries -inf);
//   * softmax in fp32 over the visible keys, P rounded to the model type (modeling_llama.py:371) - attn_kernel's half-wave
//     reduction, step for step, so a row's probabilities do not depend on which kernel ran;
//   * out^T[dim][row] = sum_key V^T[dim][key] P[row][key]: B = P (8 consecutive-ish keys of a row per lane, from the score
//     tile in LDS), A = V^T - V lies [key][dim] in the arena, so its 64-key chunks are staged in LDS as they are and read
//     back TRANSPOSED by ds_read_b64_tr_b16 (per 16 lanes a block of 4 keys x 16 dims, delivered dim-major).  The k index of
//     the two operands only has to AGREE: element j = 4h + q of lane group g stands for key 16h + 4g + q of the 32-key step,
//     which makes the two groups of a 32-lane half read 8 consecutive key rows (288-byte row pitch: conflict-free).
// fp32 accumulation inside the MFMA, one rounding of the output - the reference's bf16 matmul; the order of the sum differs
// from attn_kernel's (and from torch's), as any two correct evaluations do.
#pragma once
#include "model_kernels.h"

#define PA_ROWS 16
#define PA_VCH 64                                   // keys per staged V chunk
#define PA_VST 288                                  // bytes per key row of the LDS V image (256 + 32)
#define PA_SPAD 4                                   // floats of padding per score row (16 rows x ds_read_b128: no bank shared)

// row groups of a prefill pass: <= 16 consecutive rows of one stream each (built on the host from the table's 8-row groups)
struct PaGroups {
    int n;
    int row0[SD_MAX_GROUPS], nrows[SD_MAX_GROUPS], pos[SD_MAX_GROUPS], max_seq[SD_MAX_GROUPS];
    const void *kv[SD_MAX_GROUPS];
};

typedef short pa_v4s __attribute__((ext_vector_type(4)));

template <typename T>
__global__ __launch_bounds__(256) void attn_prefill_kernel(const T *__restrict__ qbuf, PaGroups pg, int layer, T *__restrict__ out,
                                                          int Hq, int Hkv, int arch, float inv_sqrt_d, int s_cap) {
    constexpr int D = 128;
    static_assert(sizeof(T) == 2, "16-bit models");
    extern __shared__ __attribute__((aligned(16))) char pa_smem[];
    const int ss = s_cap + PA_SPAD;                               // score row pitch (floats)
    float *sc = reinterpret_cast<float *>(pa_smem);               // [16][ss]
    char *vb = pa_smem + (size_t)PA_ROWS * ss * sizeof(float);    // [PA_VCH][PA_VST]
    const int head = blockIdx.x, g = blockIdx.y;
    const int tid = threadIdx.x, w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = tid & 63;
    const int r0 = pg.row0[g], nr = pg.nrows[g], p0 = pg.pos[g], max_seq = pg.max_seq[g];
    const int kvh = head / (Hq / Hkv);
    const T *karena = (const T *)pg.kv[g] + (size_t)layer * 2 * Hkv * max_seq * D;
    const T *K = karena + (size_t)kvh * max_seq * D;
    const T *V = karena + (size_t)(Hkv + kvh) * max_seq * D;
    const int s_hi = p0 + nr, s_last = s_hi - 1;                  // row t of the group sees keys 0 .. p0 + t
    const int s_pad = (s_hi + 31) & ~31;

    {   // ---- scores (attn_kernel's MFMA path): lane l ends up with score[key = 16 kt + 4 (l >> 4) + j][row = l & 15]
        const int mrow = lane & 15, kq = (lane >> 4) * 8;
        u32x4 qf[D / 32];
#pragma unroll
        for (int dk = 0; dk < D / 32; ++dk) {
            qf[dk] = *reinterpret_cast<const u32x4 *>(qbuf + (size_t)(r0 + min(mrow, nr - 1)) * Hq * D + head * D + dk * 32 + kq);
#pragma unroll
            for (int i = 0; i < 4; ++i) qf[dk][i] = mrow < nr ? qf[dk][i] : 0u;       // rows >= nr of the q operand are zero
        }
        for (int kt0 = w; kt0 * 16 < s_hi; kt0 += 16) {           // four key tiles per wave and round, all K loads up front
            u32x4 kf[4][D / 32];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const T *kr = K + (size_t)min((kt0 + 4 * u) * 16 + mrow, s_last) * D + kq;
#pragma unroll
                for (int dk = 0; dk < D / 32; ++dk) kf[u][dk] = *reinterpret_cast<const u32x4 *>(kr + dk * 32);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int kt = kt0 + 4 * u;
                if (kt * 16 >= s_hi) continue;
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int dk = 0; dk < D / 32; ++dk) acc = mfma16<T>(kf[u][dk], qf[dk], acc);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int s = kt * 16 + (lane >> 4) * 4 + j;
                    if (s < s_hi) {
                        float v = rnd<T>(acc[j]);
                        if (arch == SD_ARCH_LLAMA) v = rnd<T>(v * inv_sqrt_d);
                        sc[(size_t)mrow * ss + s] = s <= p0 + mrow ? v : -INFINITY;
                    }
                }
            }
        }
    }
    __syncthreads();

    {   // ---- softmax: one half-wave per row, the partial sums a full wave's lower and upper lanes would hold
        const int hl = tid & 31, grp = tid >> 5;
        const bool upper = (tid & 32) != 0;
        for (int t = grp; t < nr; t += 8) {
            float *row = sc + (size_t)t * ss;
            const int len = min(s_hi, p0 + t + 1);
            float lo, hi, lo1, hi1;
            float m0 = -INFINITY;
            for (int s = hl; s < len; s += 32) m0 = fmaxf(m0, row[s]);
            half_maxes(m0, lo, hi);
            const float m = upper ? hi : lo;
            float s0 = 0.f, s1 = 0.f;                             // what lanes l and l + 32 of a full wave accumulate
            for (int s = hl; s < len; s += 64) {
                const float e = expf(row[s] - m);
                row[s] = e;
                s0 += e;
            }
            for (int s = hl + 32; s < len; s += 64) {
                const float e = expf(row[s] - m);
                row[s] = e;
                s1 += e;
            }
            half_sums(s0, lo, hi);
            half_sums(s1, lo1, hi1);
            const float sum = upper ? hi + hi1 : lo + lo1;
            for (int s = hl; s < s_pad; s += 32) row[s] = s < len ? rnd<T>(row[s] / sum) : 0.f;
        }
        for (int i = tid; i < (PA_ROWS - nr) * s_pad; i += 256) {  // rows past the group: zero probabilities (never stored)
            const int t = nr + i / s_pad, s = i - (i / s_pad) * s_pad;
            sc[(size_t)t * ss + s] = 0.f;
        }
    }

    {   // ---- P.V on the matrix cores; wave w owns head dims 32 w .. 32 w + 31 (two 16-dim tiles)
        const int g4 = lane >> 4, i16 = lane & 15, q4 = i16 >> 2, p4 = i16 & 3;
        f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
        const int nch = (s_hi + PA_VCH - 1) / PA_VCH;
        // a chunk = 64 keys x 256 B = 1024 pieces of 16 B; thread tid moves pieces tid, tid + 256, ...: key piece >> 4, 16-byte
        // column piece & 15 (a key past the range re-reads the last one: its probabilities are zero)
        u32x4 vr[4];
        auto vload = [&](int c) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int piece = tid + 256 * i, key = min(c * PA_VCH + (piece >> 4), s_last);
                vr[i] = *reinterpret_cast<const u32x4 *>(V + (size_t)key * D + (piece & 15) * 8);
            }
        };
        vload(0);
        for (int c = 0; c < nch; ++c) {
            __syncthreads();                                      // the previous chunk is no longer read (c = 0: P is complete)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int piece = tid + 256 * i;
                *reinterpret_cast<u32x4 *>(vb + (piece >> 4) * PA_VST + (piece & 15) * 16) = vr[i];
            }
            __syncthreads();
            if (c + 1 < nch) vload(c + 1);
            const int nsteps = min(PA_VCH / 32, (s_hi - c * PA_VCH + 31) / 32);
            for (int st = 0; st < nsteps; ++st) {
                // B: P[row i16][k0 + 16 h + 4 g4 + (0..3)], h = 0, 1 - already values of T: the conversion is exact
                const float *pr = sc + (size_t)i16 * ss + c * PA_VCH + st * 32 + 4 * g4;
                const f32x4 pa = *reinterpret_cast<const f32x4 *>(pr), pb = *reinterpret_cast<const f32x4 *>(pr + 16);
                const T ph[8] = {(T)pa[0], (T)pa[1], (T)pa[2], (T)pa[3], (T)pb[0], (T)pb[1], (T)pb[2], (T)pb[3]};
                const u32x4 pf = *reinterpret_cast<const u32x4 *>(ph);
#pragma unroll
                for (int d2 = 0; d2 < 2; ++d2) {
                    // A: lane 4 q4 + p4 of its 16-lane group supplies the address of key row 16 h + 4 g4 + q4 of the step,
                    // dims 16 dt + 4 p4 .. + 3; lane i16 receives dim 16 dt + i16 of the group's four keys
                    const char *va = vb + (size_t)(st * 32 + 4 * g4 + q4) * PA_VST + ((2 * w + d2) * 16 + 4 * p4) * 2;
                    const pa_v4s a0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((pa_v4s __attribute__((address_space(3))) *)(va));
                    const pa_v4s a1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((pa_v4s __attribute__((address_space(3))) *)(va + 16 * PA_VST));
                    u32x4 af;
                    af[0] = ((const unsigned *)&a0)[0]; af[1] = ((const unsigned *)&a0)[1];
                    af[2] = ((const unsigned *)&a1)[0]; af[3] = ((const unsigned *)&a1)[1];
                    acc[d2] = mfma16<T>(af, pf, acc[d2]);
                }
            }
        }
        // lane l holds out[row l & 15][dims 16 dt + 4 (l >> 4) .. + 3]: one 8-byte store inside an 8-element operand group
        if (i16 < nr) {
#pragma unroll
            for (int d2 = 0; d2 < 2; ++d2)
                store4_maybe_wt<false>(out + xoff<T>(r0 + i16, head * D + (2 * w + d2) * 16 + 4 * g4, Hq * D), acc[d2][0], acc[d2][1],
                                       acc[d2][2], acc[d2][3]);
        }
    }
}
