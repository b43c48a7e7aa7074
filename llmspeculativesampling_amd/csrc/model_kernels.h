// Decoder-forward kernels for gfx950 (Llama / OPT, fp32 or bf16 storage).
// Reference semantics: sampling/models/modeling_llama.py:292-393, 405-457 and
// sampling/models/modeling_opt.py:160-278, 303-378; every rnd<T>() marks a point where the
// reference's per-op result is materialised in the weight dtype.
#pragma once
#include "common.h"
#include <type_traits>

using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using f16x8 = __attribute__((ext_vector_type(8))) _Float16;
using f32x4 = __attribute__((ext_vector_type(4))) float;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned int;

// D = A x B + C on a 16 x 16 x 32 tile for either 16-bit storage type H (bf16_t / f16_t): the two MFMA forms share the
// operand layout and run at the same rate, so every kernel below is written once and instantiated per type.
template <typename H>
__device__ __forceinline__ f32x4 mfma16(u32x4 a, u32x4 b, f32x4 c) {
    if constexpr (std::is_same<H, f16_t>::value)
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    else
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// Row table of one forward, passed BY VALUE in the kernel arguments (no device copy to keep alive): rows may belong to
// several independent sequences ("streams", SURVEY.md 8(e)/(f): stream-batched decode) that share the GEMMs and differ
// in position, token source and KV arena.  A single-sequence forward is a table with one stream.
#define SD_MAX_ROWS 80                               // rows of one stream-batched pass (config 4: 8 streams x (gamma + 1 = 9) = 72)
#define SD_MAX_TREE 64                               // nodes of a tree verify (ancestor masks are 64-bit words)
#define SD_MAX_STREAMS 16
#define SD_MAX_GROUPS 32
struct RowTab {
    int n_rows, n_streams, n_groups, n_logit_rows;
    int contig, pos0;                               // contig 1: one stream, row m at position pos0 + m (prefill chunks of
                                                    // up to 256 rows); contig 2: n_streams such runs back to back - rows
                                                    // seg_row0[i] .. seg_row0[i+1]-1 are stream i at seg_pos0[i] + (m -
                                                    // seg_row0[i]) (batched prefill of several prompts in one pass).  The
                                                    // two per-row arrays below are unused in both forms
    int seg_row0[SD_MAX_STREAMS + 1], seg_pos0[SD_MAX_STREAMS];
    int row_pos[SD_MAX_ROWS];                       // absolute position of row m in its sequence
    unsigned char row_stream[SD_MAX_ROWS];          // stream of row m
    unsigned char xmap[SD_MAX_ROWS];                // rows of the hidden state that feed the lm_head, in output order
    const int32_t *tok_base[SD_MAX_STREAMS];        // stream's token buffer, indexed by absolute position
    void *kv_base[SD_MAX_STREAMS];                  // stream's KV arena [L][2][Hkv][max_seq][D]
    int max_seq[SD_MAX_STREAMS];                    // that arena's capacity
    // tree verify (reference kvcache_model.py:38-136, forward_tree_attention): the rows are the nodes of a draft token
    // tree appended after `tree_base` cached positions; row m sees every cached position and the tree rows whose bit
    // is set in tree_mask[m] (its ancestors and itself).  row_pos[] then carries the RoPE / learned position of the
    // node (its depth), while its K / V rows go to the arena slot tree_base + m.
    int tree, tree_base;
    unsigned long long tree_mask[SD_MAX_TREE];
    int kv_fp8;                                     // the arenas hold OCP fp8 e4m3 (1 byte per element) instead of T
    const float *kv_scale[SD_MAX_STREAMS];          // fp8: per (layer, k|v, kv head) scales [L][2][Hkv], x = fp8 * scale
    // attention groups: <= ATT_TQ consecutive rows of one stream (first row, count, position of the first row, stream)
    int grp_row0[SD_MAX_GROUPS], grp_n[SD_MAX_GROUPS], grp_pos[SD_MAX_GROUPS], grp_stream[SD_MAX_GROUPS];
};

// Activation operands of the bf16 GEMMs (normalised rows, attention output, MLP activation, OPT projections) are kept
// in the MFMA B-operand tile layout [M/16][K/32][64 lanes][8]: lane 16*((k/8)%4) + m%16 of tile (m/16, k/32) holds
// X[m][k..k+8).  A wave's fragment load is then one contiguous 1 KiB read instead of 16 row segments 2*K bytes apart
// (measured on the 13b shapes, tools/gemm_bench.py: gate/up at 16 rows 59.8 -> 49.0 us, at 64 rows 74 -> 63 us).
// Producers write through xoff<T>(); fp32 models (gemm_f32_simple) keep plain rows.
template <typename T>
__device__ __forceinline__ size_t xoff(int m, int k, int K) {
    if constexpr (sizeof(T) == 2)
        return ((size_t)(m >> 4) * (K >> 5) + (k >> 5)) * 512 + ((((k >> 3) & 3) << 4) + (m & 15)) * 8 + (k & 7);
    else
        return (size_t)m * K + k;
}

__device__ __forceinline__ int tab_stream(const RowTab &t, int m) {
    if (t.contig == 2) {                            // <= 16 runs: the last one that starts at or before row m
        int sg = 0;
        for (int i = 1; i < t.n_streams; ++i) sg += (t.seg_row0[i] <= m);
        return sg;
    }
    return t.contig ? 0 : (int)t.row_stream[m];
}
__device__ __forceinline__ int tab_pos(const RowTab &t, int m) {
    if (t.contig == 2) { const int sg = tab_stream(t, m); return t.seg_pos0[sg] + (m - t.seg_row0[sg]); }
    return t.contig ? t.pos0 + m : t.row_pos[m];
}
// token id of row m: read at its absolute position in the stream's token buffer, or (tree verify) at its node index
__device__ __forceinline__ int tab_tok(const RowTab &t, int m, int pos) {
    return t.tree ? t.tok_base[0][m] : t.tok_base[tab_stream(t, m)][pos];
}
// arena slot row m's K / V rows are written to: its position, or (tree verify) the next free slot
__device__ __forceinline__ int tab_slot(const RowTab &t, int m) { return t.tree ? t.tree_base + m : tab_pos(t, m); }

enum { NORM_RMS = 0, NORM_LN = 1 };
enum { RES_PRE = 0, RES_POST = 1, RES_NONE = 2 };   // norm after the residual feeds the next GEMM / replaces x / no norm

// ------------------------------------------------------------------------------------------
// Weight-streaming GEMM  part[s][m][n] = sum_{k in slice s} X[m][k] * W[n][k]
// bf16: W in 1 KiB tiles [N/16][K/32][64 lanes][8], one wave = one (n-tile, k-slice) unit.
// The MFMA runs with A = W tile (rows n), B = X^T (cols m): lane holds D[n = 4*(lane>>4)+j][m = lane&15],
// so each lane stores 4 consecutive n as one float4.
// ------------------------------------------------------------------------------------------
enum { EPI_PART = 0, EPI_ACT_SILU = 1, EPI_ACT_RELU = 2, EPI_QKV_ROPE = 3, EPI_QKV_PLAIN = 4, EPI_HEAD = 5 };

// Arguments of the fused epilogues (SB == 1: the workgroup holds the whole dot product after its LDS fold).
// ---- fp8 (OCP e4m3, gfx950's native form) KV arena: x is stored as fp8(x / scale), read back as fp8 * scale.  The
// attention kernel folds the K scale into the scores and the V scale into the output, so the fp8 -> 16-bit conversion of a
// key / value element is exact (3 mantissa bits fit either 16-bit type).
__device__ __forceinline__ unsigned char to_fp8(float x, float inv_scale) {
    const float v = fminf(fmaxf(x * inv_scale, -448.f), 448.f);
    return (unsigned char)(__builtin_amdgcn_cvt_pk_fp8_f32(v, 0.f, 0, false) & 0xff);
}
__device__ __forceinline__ void store4_fp8(unsigned char *dst, float a, float b, float c, float d, float inv_scale) {
    const float lo = -448.f, hi = 448.f;
    int w = __builtin_amdgcn_cvt_pk_fp8_f32(fminf(fmaxf(a * inv_scale, lo), hi), fminf(fmaxf(b * inv_scale, lo), hi), 0, false);
    w = __builtin_amdgcn_cvt_pk_fp8_f32(fminf(fmaxf(c * inv_scale, lo), hi), fminf(fmaxf(d * inv_scale, lo), hi), w, true);
    *reinterpret_cast<int *>(dst) = w;
}
// 8 consecutive fp8 elements -> 8 elements of the 16-bit type T packed as the MFMA operand / the P.V unpack expects
template <typename T>
__device__ __forceinline__ u32x4 fp8x8_to_16(uint2 raw) {
    float f[8];
    f[0] = __builtin_amdgcn_cvt_f32_fp8((int)raw.x, 0); f[1] = __builtin_amdgcn_cvt_f32_fp8((int)raw.x, 1);
    f[2] = __builtin_amdgcn_cvt_f32_fp8((int)raw.x, 2); f[3] = __builtin_amdgcn_cvt_f32_fp8((int)raw.x, 3);
    f[4] = __builtin_amdgcn_cvt_f32_fp8((int)raw.y, 0); f[5] = __builtin_amdgcn_cvt_f32_fp8((int)raw.y, 1);
    f[6] = __builtin_amdgcn_cvt_f32_fp8((int)raw.y, 2); f[7] = __builtin_amdgcn_cvt_f32_fp8((int)raw.y, 3);
    T h[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) h[i] = (T)f[i];
    return *reinterpret_cast<const u32x4 *>(h);
}

template <typename H>
struct GemmEpiT {
    H *out;                   // ACT: act[M][n_out]            QKV: q buffer [M][Hq*D]
    const H *bias;            // OPT biases (NULL for llama)
    int n_out;                // ACT: row stride of act (= inter)
    const H *cos_t, *sin_t;
    int Hq, Hkv, D, layer;    // QKV: K / V rows go to tab.kv_base[stream] + layer offset, at position tab.row_pos[m]
    float q_scale;
    int use_xmap;             // lm_head: activation row m is tab.xmap[m] of X
    int x_rowmajor;           // X is plain [M][K] rows (the public sd_gemm_bf16 entry) instead of the tile layout
    // EPI_HEAD (lm_head, whole k-range per workgroup): besides the logits slab, the maximum of every 16-column tile
    // (NaN if the tile holds one) goes to tile_max[m][N/16] and the tile's 16 entries of the probability row
    // zero_rows + m * zero_ld are cleared - what the sampler needs to pick its top-k candidates without a pass over V
    float *tile_max;
    float *zero_rows;
    long zero_ld;
    float *zero_ptr[16];      // zero_rows == NULL: row m's probability row (the rows of a stream-batched pass lie in different arenas)
    // norm on load (gemm_bf16_stream_xn): X holds the UN-normalised residual rows; nrm_ssq[m][nrm_nt] the per-16-column
    // sums of their squares (left by the producing epilogue), nrm_w the RMSNorm weight [K]
    const float *nrm_ssq;
    const H *nrm_w;
    int nrm_nt;
    float nrm_eps;
    // residual epilogue (resid_epilogue_step): the residual rows [M][N] updated in place, the same rows in the operand
    // tile layout, and the per-tile sums of squares [M][N/16] for the consumer's norm on load
    H *res_x, *res_h;
    float *res_ssq;
    RowTab tab;
};
using GemmEpi = GemmEpiT<bf16_t>;
static_assert(sizeof(GemmEpiT<bf16_t>) <= 3072, "GemmEpiT travels by value in the kernel arguments (4 KiB in all)");

template <typename H>
__device__ __forceinline__ void store4(H *dst, float a, float b, float c, float d) {
    const H v[4] = {(H)a, (H)b, (H)c, (H)d};
    *reinterpret_cast<uint2 *>(dst) = *reinterpret_cast<const uint2 *>(v);
}

// Write-through (sc1) stores for data another workgroup of the SAME launch reads (fused_kernels.h, normload_kernels.h): a relaxed agent-scope
// atomic store is a plain global_store with the sc1 bit, so the bytes are in memory (not in this XCD's L2 only) once the
// wave's vmcnt drains; pointers are 8-byte aligned at every call site.
__device__ __forceinline__ void store8_wt(void *dst, uint2 v) {
    __hip_atomic_store(reinterpret_cast<unsigned long long *>(dst), ((unsigned long long)v.y << 32) | v.x, __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);
}
template <bool WT>
__device__ __forceinline__ void store_f32x4(float *dst, f32x4 v) {
    if constexpr (WT) {
        store8_wt(dst, uint2{__float_as_uint(v[0]), __float_as_uint(v[1])});
        store8_wt(dst + 2, uint2{__float_as_uint(v[2]), __float_as_uint(v[3])});
    } else {
        *reinterpret_cast<f32x4 *>(dst) = v;
    }
}
template <bool WT, typename H>
__device__ __forceinline__ void store4_maybe_wt(H *dst, float a, float b, float c, float d) {
    const H v[4] = {(H)a, (H)b, (H)c, (H)d};
    if constexpr (WT) store8_wt(dst, *reinterpret_cast<const uint2 *>(v));
    else *reinterpret_cast<uint2 *>(dst) = *reinterpret_cast<const uint2 *>(v);
}

// One fold step of the streaming GEMMs' epilogue: red[wave][pp][lane] holds the 4 waves' accumulators of PT tiles; the
// folded sums go to the split-K slab (EPI_PART / EPI_HEAD) or through the fused epilogue.  Shared by gemm_bf16_stream and
// gemm_small (small_kernels.h).
// E: GemmEpiT<H>, or any view with the same member names.
// gemm_epilogue_fold: the epilogue proper, on any source of folded sums - `folded(pp, l)` returns lane l's f32x4 of
// accumulator tile pp with the k-split already added up; gemm_epilogue_step (below) is the form of the kernels whose four
// waves leave their accumulators in red[wave][pp][lane].
template <int MT, int EPI, int NTW, int PT, typename H = bf16_t, typename E = GemmEpiT<H>, bool WT = false, typename FoldFn>
__device__ __forceinline__ void gemm_epilogue_fold(FoldFn folded, int fs, float *__restrict__ part, int M, int Mpad,
                                                   int N, int sb, int ntg, const E &e, int tid_ = -1) {
    // tid_: the calling wave's threads counted from 0 (a caller whose epilogue threads are not the workgroup's first)
    const int tidx = tid_ >= 0 ? tid_ : (int)threadIdx.x;
    if constexpr (EPI == EPI_PART) {
        if (tidx < PT * 64) {
            const int pp = tidx >> 6, l = tidx & 63, q = fs * PT + pp;
            const int m = (q % MT) * 16 + (l & 15), nt = ntg * NTW + q / MT;
            if (m < M) store_f32x4<WT>(part + ((size_t)sb * Mpad + m) * N + nt * 16 + (l >> 4) * 4, folded(pp, l));
        }
    } else if constexpr (EPI == EPI_HEAD) {
        static_assert(EPI != EPI_HEAD || (MT == 1 && NTW == 1 && PT == 1), "EPI_HEAD: one 16-row tile, one n-tile per workgroup");
        if (tidx < 64) {                                   // one whole wave: lanes l, l^16, l^32, l^48 share row m
            const int l = tidx, m = l & 15, nt = ntg;
            const f32x4 r = folded(0, l);
            float mx = fmaxf(fmaxf(r[0], r[1]), fmaxf(r[2], r[3]));
            int bad = (r[0] != r[0]) | (r[1] != r[1]) | (r[2] != r[2]) | (r[3] != r[3]);
            mx = fmaxf(mx, __shfl_xor(mx, 16, 64)); bad |= __shfl_xor(bad, 16, 64);
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64)); bad |= __shfl_xor(bad, 32, 64);
            if (m < M) {
                *reinterpret_cast<f32x4 *>(part + (size_t)m * N + nt * 16 + (l >> 4) * 4) = r;
                float *zr = e.zero_rows ? e.zero_rows + (size_t)m * e.zero_ld : e.zero_ptr[m];
                if (zr) *reinterpret_cast<f32x4 *>(zr + nt * 16 + (l >> 4) * 4) = f32x4{0.f, 0.f, 0.f, 0.f};
                if (l < 16) e.tile_max[(size_t)m * (N >> 4) + nt] = bad ? __uint_as_float(0x7fc00000u) : mx;
            }
        }
    } else if constexpr (EPI == EPI_ACT_SILU) {
        // weights interleaved 8 gate rows / 8 up rows per tile: quads 0,1 = gate cols, quads 2,3 = the same up cols
        if (tidx < PT * 32) {
            const int pp = tidx >> 5, l = tidx & 31, q = fs * PT + pp;
            const int m = (q % MT) * 16 + (l & 15), nt = ntg * NTW + q / MT;
            if (m < M) {
                const f32x4 g = folded(pp, l), u = folded(pp, l + 32);
                float a[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const float gj = rnd<H>(g[c]), uj = rnd<H>(u[c]);
                    a[c] = rnd<H>(gj / (1.0f + expf(-gj))) * uj;       // silu(gate) * up (modeling_llama.py:220)
                }
                store4_maybe_wt<WT>(e.out + xoff<H>(m, nt * 8 + (l >> 4) * 4, e.n_out), a[0], a[1], a[2], a[3]);
            }
        }
    } else if constexpr (EPI == EPI_ACT_RELU) {
        if (tidx < PT * 64) {
            const int pp = tidx >> 6, l = tidx & 63, q = fs * PT + pp;
            const int m = (q % MT) * 16 + (l & 15), col = (ntg * NTW + q / MT) * 16 + (l >> 4) * 4;
            if (m < M) {
                const f32x4 r = folded(pp, l);
                float a[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const float f = rnd<H>(r[c] + (e.bias ? to_f(e.bias[col + c]) : 0.f));
                    a[c] = f > 0.f ? f : 0.f;
                }
                store4_maybe_wt<WT>(e.out + xoff<H>(m, col, e.n_out), a[0], a[1], a[2], a[3]);
            }
        }
    } else {
        // QKV: bias, RoPE (rows pair-interleaved inside each q/k head: d, d+D/2, d+1, d+1+D/2, ...) or the OPT
        // q pre-scale, then q -> buffer and K/V rows appended in place at positions pos0 + m.
        const int hd = e.D >> 1;
        if (tidx < PT * 64) {
            const int pp = tidx >> 6, l = tidx & 63, q = fs * PT + pp;
            const int m = (q % MT) * 16 + (l & 15), col = (ntg * NTW + q / MT) * 16 + (l >> 4) * 4;
            if (m < M) {
                const f32x4 r = folded(pp, l);
                const int head = col / e.D, within = col - head * e.D;
                const bool is_q = head < e.Hq, is_k = !is_q && head < e.Hq + e.Hkv;
                const int strm = tab_stream(e.tab, m), pos = tab_pos(e.tab, m), mseq = e.tab.max_seq[strm];
                // element offset of this row's head inside the arena [L][2][Hkv][max_seq][D] (K heads, then V heads)
                const size_t kvoff = (size_t)e.layer * 2 * e.Hkv * mseq * e.D +
                                     ((size_t)(head - e.Hq) * mseq + tab_slot(e.tab, m)) * e.D;      // head - Hq in [0, 2*Hkv)
                const bool kv8 = !is_q && e.tab.kv_fp8;
                H *dst = is_q ? e.out + (size_t)m * e.Hq * e.D + head * e.D : (H *)e.tab.kv_base[strm] + kvoff;
                unsigned char *dst8 = (unsigned char *)e.tab.kv_base[strm] + kvoff;
                const float inv_sc = kv8 ? 1.0f / e.tab.kv_scale[strm][(size_t)e.layer * 2 * e.Hkv + (head - e.Hq)] : 1.0f;
                float x[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) x[c] = rnd<H>(r[c] + (e.bias ? to_f(e.bias[col + c]) : 0.f));
                if (EPI == EPI_QKV_ROPE && (is_q || is_k)) {
                    // the lane's four columns are dims d, d + hd, d + 1, d + 1 + hd (pair-interleaved weight rows), d even:
                    // two 4-byte stores (dims d, d + 1 and d + hd, d + 1 + hd) instead of four 2-byte ones
                    const int d0 = within >> 1;
                    float o0[2], o1[2];
                    // (cos / sin of dims d0, d0 + 1: one 4-byte load each - d0 and pos * hd are even)
                    H c2[2], s2[2];
                    *reinterpret_cast<unsigned *>(c2) = *reinterpret_cast<const unsigned *>(e.cos_t + (size_t)pos * hd + d0);
                    *reinterpret_cast<unsigned *>(s2) = *reinterpret_cast<const unsigned *>(e.sin_t + (size_t)pos * hd + d0);
#pragma unroll
                    for (int pr = 0; pr < 2; ++pr) {
                        const float cs = to_f(c2[pr]), sn = to_f(s2[pr]);
                        const float x0 = x[2 * pr], x1 = x[2 * pr + 1];
                        o0[pr] = rnd<H>(rnd<H>(x0 * cs) + rnd<H>(-x1 * sn));
                        o1[pr] = rnd<H>(rnd<H>(x1 * cs) + rnd<H>(x0 * sn));
                    }
                    if (kv8) {
#pragma unroll
                        for (int pr = 0; pr < 2; ++pr) { dst8[d0 + pr] = to_fp8(o0[pr], inv_sc); dst8[d0 + pr + hd] = to_fp8(o1[pr], inv_sc); }
                    } else {
                        const H lo[2] = {(H)o0[0], (H)o0[1]}, hi[2] = {(H)o1[0], (H)o1[1]};
                        *reinterpret_cast<unsigned *>(dst + d0) = *reinterpret_cast<const unsigned *>(lo);
                        *reinterpret_cast<unsigned *>(dst + d0 + hd) = *reinterpret_cast<const unsigned *>(hi);
                    }
                } else {
                    if (EPI == EPI_QKV_PLAIN && is_q) {
#pragma unroll
                        for (int c = 0; c < 4; ++c) x[c] = rnd<H>(x[c] * e.q_scale);
                    }
                    if (kv8) store4_fp8(dst8 + within, x[0], x[1], x[2], x[3], inv_sc);
                    else store4(dst + within, x[0], x[1], x[2], x[3]);
                }
            }
        }
    }
}

template <int MT, int EPI, int NTW, int PT, typename H = bf16_t, typename E = GemmEpiT<H>, bool WT = false>
__device__ __forceinline__ void gemm_epilogue_step(f32x4 (*red)[PT][64], int fs, float *__restrict__ part, int M, int Mpad,
                                                   int N, int sb, int ntg, const E &e, int tid_ = -1) {
    auto folded = [&](int pp, int l) -> f32x4 {
        return (red[0][pp][l] + red[1][pp][l]) + (red[2][pp][l] + red[3][pp][l]);
    };
    gemm_epilogue_fold<MT, EPI, NTW, PT, H, E, WT>(folded, fs, part, M, Mpad, N, sb, ntg, e, tid_);
}

// Residual epilogue of a GEMM whose workgroup holds the COMPLETE sums of one 16-column tile for <= 16 rows (k-split folded
// inside the workgroup): x' = rnd(x + rnd(sum + bias)) exactly as residual_norm_kernel forms it (modeling_llama.py:440,
// 446), stored to the residual rows and, un-normalised, to the operand tile layout; plus the tile's sum of squares per
// row in a fixed order - a lane adds its four columns in sequence, the four quads fold as (q0 + q1) + (q2 + q3) - for the
// consumer's norm on load (normload_kernels.h).  One wave (tidx < 64); red[wave][0][lane] as in gemm_epilogue_step.
// resid_prefetch(): the lane's four residual values, to be requested long before the sums are ready.
template <typename H>
__device__ __forceinline__ uint2 resid_prefetch(const H *res_x, int M, int N, int nt, int tidx) {
    const int m = tidx & 15;
    return (tidx < 64 && m < M) ? *reinterpret_cast<const uint2 *>(res_x + (size_t)m * N + nt * 16 + ((tidx & 63) >> 4) * 4) : uint2{0u, 0u};
}
template <typename H, typename E>
__device__ __forceinline__ void resid_epilogue_step(f32x4 (*red)[1][64], int M, int N, int nt, const E &e, int tidx, uint2 xpre) {
#pragma clang fp contract(off)
    if (tidx >= 64) return;
    const int l = tidx, m = l & 15, col = nt * 16 + (l >> 4) * 4;
    const f32x4 r = (red[0][0][l] + red[1][0][l]) + (red[2][0][l] + red[3][0][l]);
    float a2 = 0.f;
    if (m < M) {
        H xin[4], v[4];
        *reinterpret_cast<uint2 *>(xin) = xpre;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float y = r[c] + (e.bias ? to_f(e.bias[col + c]) : 0.f);
            const float f = rnd<H>(to_f(xin[c]) + rnd<H>(y));
            v[c] = (H)f;
            a2 += f * f;
        }
        *reinterpret_cast<uint2 *>(e.res_x + (size_t)m * N + col) = *reinterpret_cast<const uint2 *>(v);
        *reinterpret_cast<uint2 *>(e.res_h + xoff<H>(m, col, N)) = *reinterpret_cast<const uint2 *>(v);
    }
    a2 += __shfl_xor(a2, 16, 64);
    a2 += __shfl_xor(a2, 32, 64);
    if (l < 16 && m < M) e.res_ssq[(size_t)m * (N >> 4) + nt] = a2;
}

template <int MT, int UNROLL, int EPI, int NTW, bool NT_LOADS = true, typename H = bf16_t>
__global__ __launch_bounds__(256) void gemm_bf16_stream(const u32x4 *__restrict__ Wp, const H *__restrict__ X,
                                                       float *__restrict__ part, int M, int Mpad, int N, int K,
                                                       int SB, int ks_per_blk, GemmEpiT<H> e) {
    // One workgroup = NTW consecutive 16-column n-tiles x one k-slab; its 4 waves take a quarter of the slab each
    // (every activation fragment a wave loads is reused for NTW weight tiles) and fold their accumulators through
    // LDS, so the number of partial slabs in HBM is SB, not 4*SB.  NTW = 1 for decode (M <= 16), 4 for prefill rows.
    constexpr int NP = NTW * MT;                                  // accumulator tiles per wave
    constexpr int PT = NP % 4 == 0 ? 4 : (NP % 2 == 0 ? 2 : 1);    // of which PT go through the LDS fold per step
    __shared__ f32x4 red[4][PT][64];
    // (the wave id as a SCALAR: with `threadIdx.x >> 6` the k-range bounds look per-lane to hipcc, the k-loops become
    //  exec-masked loops and the accumulators are kept twice - main loop and tail - in AGPRs: <3, 2, ., 4> 78 + 96 registers
    //  against 76 + 48 with the scalar id, two waves per SIMD against four)
#ifndef SD_STREAM_VECTOR_WV
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
#else
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#endif
    const int NTG = (N >> 4) / NTW, KS = K >> 5;
    const int sb = blockIdx.x / NTG, ntg = blockIdx.x - sb * NTG;
    const int kb0 = sb * ks_per_blk, kb1 = min(KS, kb0 + ks_per_blk);
    const int per = (kb1 - kb0 + 3) >> 2;
    const int ks0 = min(kb1, kb0 + wv * per), ks1 = min(kb1, ks0 + per);
    const u32x4 *wp[NTW];
#pragma unroll
    for (int j = 0; j < NTW; ++j) wp[j] = Wp + ((size_t)(ntg * NTW + j) * KS + ks0) * 64 + lane;
    const int mrow = lane & 15, kq = (lane >> 4) * 8;
    const H *xp[MT];
    bool mv[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t) {
        const int m = t * 16 + mrow;
        mv[t] = m < M;
        const int msrc = mv[t] ? (e.use_xmap ? (int)e.tab.xmap[m] : m) : 0;
        xp[t] = e.x_rowmajor ? X + (size_t)msrc * K + (size_t)ks0 * 32 + kq
                             : X + ((size_t)(msrc >> 4) * KS + ks0) * 512 + ((lane >> 4) * 16 + (msrc & 15)) * 8;
    }
    const int xstep = e.x_rowmajor ? 32 : 512;                    // elements from one k-tile's fragment to the next
    f32x4 acc[NTW][MT];
#pragma unroll
    for (int j = 0; j < NTW; ++j)
#pragma unroll
        for (int t = 0; t < MT; ++t) acc[j][t] = f32x4{0.f, 0.f, 0.f, 0.f};

    int ks = ks0;
    for (; ks + UNROLL <= ks1; ks += UNROLL) {
        u32x4 w[UNROLL][NTW];
        u32x4 x[UNROLL][MT];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u)
#pragma unroll
            for (int j = 0; j < NTW; ++j) w[u][j] = NT_LOADS ? __builtin_nontemporal_load(wp[j] + (size_t)u * 64) : wp[j][(size_t)u * 64];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u)
#pragma unroll
            for (int t = 0; t < MT; ++t)
                x[u][t] = mv[t] ? *reinterpret_cast<const u32x4 *>(xp[t] + u * xstep) : u32x4{0u, 0u, 0u, 0u};
#pragma unroll
        for (int u = 0; u < UNROLL; ++u)
#pragma unroll
            for (int j = 0; j < NTW; ++j)
#pragma unroll
                for (int t = 0; t < MT; ++t)
                    acc[j][t] = mfma16<H>(w[u][j], x[u][t], acc[j][t]);
#pragma unroll
        for (int j = 0; j < NTW; ++j) wp[j] += (size_t)UNROLL * 64;
#pragma unroll
        for (int t = 0; t < MT; ++t) xp[t] += UNROLL * xstep;
    }
    if (ks < ks1) {
        // the last < UNROLL k-steps as ONE burst as well (requested together, multiplied in order): taken one at a time they
        // cost a memory round trip each - three of a wave's seven at the draft model's K = 768, three on top of six groups
        // in the 13b down projection
        // EPI_PART with UNROLL = 8 takes a form with no branch around a load or an MFMA: with `if (u < rem)` around each of
        // the seven steps hipcc built a divergent-branch ladder with ~900 register moves for that one instance -
        // gemm_bf16_stream<1, 8, EPI_PART, 1> came out at 250 VGPRs + 64 AGPRs, one wave per SIMD, and the stream-batched
        // draft lm_head took 48 us instead of 10 (profiles/r04_kernel_by_grid_throughput_b8.txt).  A k-step past the range
        // re-requests the burst's first tile and is multiplied as zero.  The fused-epilogue instances compile well (78 + 8
        // registers) and keep the branches: for them the seventh request costs the draft step 1.4-4.7 us (A/B on one box).
        const int rem = ks1 - ks;
        if constexpr (UNROLL >= 8 && EPI == EPI_PART) {
            u32x4 w[UNROLL - 1][NTW];
            u32x4 x[UNROLL - 1][MT];
#pragma unroll
            for (int u = 0; u < UNROLL - 1; ++u) {
                const int uu = u < rem ? u : 0;
#pragma unroll
                for (int j = 0; j < NTW; ++j) w[u][j] = __builtin_nontemporal_load(wp[j] + (size_t)uu * 64);
            }
#pragma unroll
            for (int u = 0; u < UNROLL - 1; ++u) {
                const int uu = u < rem ? u : 0;
#pragma unroll
                for (int t = 0; t < MT; ++t)
                    x[u][t] = mv[t] ? *reinterpret_cast<const u32x4 *>(xp[t] + uu * xstep) : u32x4{0u, 0u, 0u, 0u};
            }
#pragma unroll
            for (int u = 0; u < UNROLL - 1; ++u)
#pragma unroll
                for (int j = 0; j < NTW; ++j) {
                    u32x4 wz = w[u][j];
#pragma unroll
                    for (int q = 0; q < 4; ++q) wz[q] = u < rem ? wz[q] : 0u;
#pragma unroll
                    for (int t = 0; t < MT; ++t) acc[j][t] = mfma16<H>(wz, x[u][t], acc[j][t]);
                }
        } else {
            u32x4 w[UNROLL > 1 ? UNROLL - 1 : 1][NTW];
            u32x4 x[UNROLL > 1 ? UNROLL - 1 : 1][MT];
#pragma unroll
            for (int u = 0; u < UNROLL - 1; ++u)
                if (u < rem) {
#pragma unroll
                    for (int j = 0; j < NTW; ++j) w[u][j] = __builtin_nontemporal_load(wp[j] + (size_t)u * 64);
#pragma unroll
                    for (int t = 0; t < MT; ++t)
                        x[u][t] = mv[t] ? *reinterpret_cast<const u32x4 *>(xp[t] + u * xstep) : u32x4{0u, 0u, 0u, 0u};
                }
#pragma unroll
            for (int u = 0; u < UNROLL - 1; ++u)
                if (u < rem) {
#pragma unroll
                    for (int j = 0; j < NTW; ++j)
#pragma unroll
                        for (int t = 0; t < MT; ++t) acc[j][t] = mfma16<H>(w[u][j], x[u][t], acc[j][t]);
                }
        }
    }
    // Fold the 4 waves' accumulators through LDS, PT tiles at a time (a fold buffer for all NTW*MT tiles at once would be
    // 64 KiB at 4x4 and cap the kernel at two workgroups per CU), each step followed by its share of the epilogue.
#pragma unroll
    for (int fs = 0; fs < NP / PT; ++fs) {
        if (fs) __syncthreads();
#pragma unroll
        for (int pp = 0; pp < PT; ++pp) red[wv][pp][lane] = acc[(fs * PT + pp) / MT][(fs * PT + pp) % MT];
        __syncthreads();
        gemm_epilogue_step<MT, EPI, NTW, PT, H>(red, fs, part, M, Mpad, N, sb, ntg, e);
    }
}

// ------------------------------------------------------------------------------------------
// The same GEMM for a matrix with FEW n-tiles and a long k-range (<= 16 rows, SB = 1, fused epilogue) - the QKV projection
// of a tensor-parallel shard: (8 + 2) heads x 128 = 80 n-tiles x K = 8192 (BASELINE config 5).  One 4-wave workgroup per
// tile puts 80 workgroups with 16 KiB in flight each on 256 CUs: 15 us for 21 MB.  Here a workgroup is SIXTEEN waves on one
// tile, each wave a sixteenth of the k-range (groups of four k-steps, bursts like gemm_bf16_stream), so that a CU has 64 KiB
// in flight and draws several times its fair share of an HBM that is otherwise idle; the sixteen partial sums fold
// through LDS in a fixed order ((((w0 + w1) + (w2 + w3)) + ...) pairwise) and the tile's epilogue runs once.
// ------------------------------------------------------------------------------------------
template <int EPI, typename H = bf16_t, int NW = 16>
__global__ __launch_bounds__(NW * 64) void gemm_bf16_stream_w16(const u32x4 *__restrict__ Wp, const H *__restrict__ X, int M, int N,
                                                              int K, GemmEpiT<H> e) {
    constexpr int U = 4;                                          // (NW = 16, or 8 for matrices with a few hundred n-tiles)
    __shared__ f32x4 red[NW][64];
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int KS = K >> 5, ntg = blockIdx.x;
    const int per = (KS + NW - 1) / NW;
    const int ks0 = min(KS, wv * per), ks1 = min(KS, ks0 + per), ksa = min(ks0, KS - 1);
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
    if (ks < ks1) {                                               // the last < 4 k-steps as one burst (clamped, zeroed at use)
        const int rem = ks1 - ks;
        u32x4 w[U - 1], x[U - 1];
#pragma unroll
        for (int u = 0; u < U - 1; ++u) {
            w[u] = __builtin_nontemporal_load(wp + (size_t)min(u, rem - 1) * 64);
            x[u] = *reinterpret_cast<const u32x4 *>(xp + min(u, rem - 1) * 512);
        }
#pragma unroll
        for (int u = 0; u < U - 1; ++u) {
            u32x4 wz = w[u];
#pragma unroll
            for (int q = 0; q < 4; ++q) wz[q] = u < rem ? wz[q] : 0u;
            acc = mfma16<H>(wz, x[u], acc);
        }
    }
    red[wv][lane] = acc;
    __syncthreads();
    auto folded = [&](int, int l) -> f32x4 {
        f32x4 q[NW / 4];
#pragma unroll
        for (int g = 0; g < NW / 4; ++g) q[g] = (red[4 * g][l] + red[4 * g + 1][l]) + (red[4 * g + 2][l] + red[4 * g + 3][l]);
        if constexpr (NW == 16) return (q[0] + q[1]) + (q[2] + q[3]);
        else return q[0] + q[1];
    };
    if (threadIdx.x < 64) gemm_epilogue_fold<1, EPI, 1, 1, H>(folded, 0, (float *)nullptr, M, 16, N, 0, ntg, e, (int)threadIdx.x);
}

// ------------------------------------------------------------------------------------------
// Many-row GEMM (prefill, wide stream batches): part[sb][m][n] = sum_{k in slab sb} X[m][k] * W[n][k]
// One workgroup = 2 x 2 waves on a (2*NTWV n-tiles) x (2*MTW m-tiles) block of the output; per k-step (32 columns of
// K) its W and X tiles - both already stored in MFMA fragment order, 1 KiB each - are copied once into LDS (double
// buffered) and every wave reads the fragments of its quadrant from there, so a W tile feeds 2*MTW and an X tile
// 2*NTWV MFMAs per global load instead of MTW / NTW in the streaming kernel.  No in-workgroup k-split: the k-range is
// cut across workgroups (slabs) only when the block count is too small.
// ------------------------------------------------------------------------------------------
template <int MTW, int NTWV, int KT, typename H = bf16_t>
__global__ __launch_bounds__(256) void gemm_bf16_tiled(const u32x4 *__restrict__ Wp, const u32x4 *__restrict__ Xp,
                                                      float *__restrict__ part, int M, int Mpad, int N, int K, int SB,
                                                      int ks_per_blk) {
    constexpr int WT = 2 * NTWV, XT = 2 * MTW, TT = WT + XT;      // tiles per k-tile column: W, X, total
    constexpr int NL = TT * KT;                                   // tiles per k-step (KT k-tiles = 32*KT columns of K)
    constexpr int LPT = (NL + 3) / 4;                             // tile loads per wave per k-step
    extern __shared__ __attribute__((aligned(16))) char dyn_smem[];
    u32x4 (*sm)[NL][64] = reinterpret_cast<u32x4 (*)[NL][64]>(dyn_smem);          // [2][NL][64]
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), wn = wv & 1, wm = wv >> 1;
    const int KS = K >> 5, NB = (N >> 4) / WT, MB = ((Mpad >> 4) + XT - 1) / XT;
    int b = blockIdx.x;
    const int mb = b % MB; b /= MB;                               // m-blocks of one n-block are neighbours (W reuse in L2)
    const int nb = b % NB, sb = b / NB;
    const int nt0 = nb * WT, mt0 = mb * XT, mt_end = Mpad >> 4;
    const int kb0 = sb * ks_per_blk, kb1 = min(KS, kb0 + ks_per_blk);      // ks_per_blk is a multiple of KT
    // slot i of a k-step: kk = i / TT (k-tile inside the step), tt = i % TT: tt < WT -> W tile nt0 + tt, else X tile
    // mt0 + tt - WT; wave wv copies slots wv, wv + 4, ...
    const u32x4 *src[LPT];
    bool ok[LPT], isw[LPT];
#pragma unroll
    for (int r = 0; r < LPT; ++r) {
        const int i = wv + 4 * r, kk = i / TT, tt = i - kk * TT;
        isw[r] = tt < WT;
        ok[r] = i < NL && (isw[r] || mt0 + tt - WT < mt_end);
        src[r] = !ok[r] ? Wp : (isw[r] ? Wp + ((size_t)(nt0 + tt) * KS + kb0 + kk) * 64 + lane
                                       : Xp + ((size_t)(mt0 + tt - WT) * KS + kb0 + kk) * 64 + lane);
    }
    f32x4 acc[NTWV][MTW];
#pragma unroll
    for (int j = 0; j < NTWV; ++j)
#pragma unroll
        for (int t = 0; t < MTW; ++t) acc[j][t] = f32x4{0.f, 0.f, 0.f, 0.f};
    u32x4 stage[LPT];
    auto fetch = [&]() {
#pragma unroll
        for (int r = 0; r < LPT; ++r) {
            stage[r] = ok[r] ? (isw[r] ? __builtin_nontemporal_load(src[r]) : *src[r]) : u32x4{0u, 0u, 0u, 0u};
            src[r] += (size_t)KT * 64;
        }
    };
    auto put = [&](int buf) {
#pragma unroll
        for (int r = 0; r < LPT; ++r)
            if (wv + 4 * r < NL) sm[buf][wv + 4 * r][lane] = stage[r];
    };
    if (kb0 < kb1) { fetch(); put(0); }
    __syncthreads();
    int cur = 0;
    for (int ks = kb0; ks < kb1; ks += KT) {
        const bool more = ks + KT < kb1;
        if (more) fetch();                                        // next k-step's tiles travel while this one is multiplied
#pragma unroll
        for (int kk = 0; kk < KT; ++kk) {
            u32x4 wf[NTWV], xf[MTW];
#pragma unroll
            for (int j = 0; j < NTWV; ++j) wf[j] = sm[cur][kk * TT + wn * NTWV + j][lane];
#pragma unroll
            for (int t = 0; t < MTW; ++t) xf[t] = sm[cur][kk * TT + WT + wm * MTW + t][lane];
#pragma unroll
            for (int j = 0; j < NTWV; ++j)
#pragma unroll
                for (int t = 0; t < MTW; ++t)
                    acc[j][t] = mfma16<H>(wf[j], xf[t], acc[j][t]);
        }
        if (more) put(cur ^ 1);
        __syncthreads();
        cur ^= 1;
    }
#pragma unroll
    for (int j = 0; j < NTWV; ++j)
#pragma unroll
        for (int t = 0; t < MTW; ++t) {
            const int m = (mt0 + wm * MTW + t) * 16 + (lane & 15), n = (nt0 + wn * NTWV + j) * 16 + (lane >> 4) * 4;
            if (m < M) *reinterpret_cast<f32x4 *>(part + ((size_t)sb * Mpad + m) * N + n) = acc[j][t];
        }
}

// fp32 storage (parity runs on small models): one wave per output column, lanes stride K.
__global__ __launch_bounds__(256) void gemm_f32_simple(const float *__restrict__ W, const float *__restrict__ X,
                                                      float *__restrict__ part, int M, int N, int K, RowTab tab,
                                                      int use_xmap) {
    const int lane = threadIdx.x & 63;
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= N) return;
    const float *w = W + (size_t)n * K;
    for (int m = 0; m < M; ++m) {
        const float *x = X + (size_t)(use_xmap ? (int)tab.xmap[m] : m) * K;
        float a = 0.f;
        for (int k = lane; k < K; k += 64) a = fmaf(w[k], x[k], a);
        a = wave_sum(a);
        if (lane == 0) part[(size_t)m * N + n] = a;
    }
}

// bf16 [N][K] row-major -> streaming tiles (see sd_pack_weight_bf16 in specdec.h)
__global__ void pack_weight_kernel(const uint16_t *__restrict__ src, uint16_t *__restrict__ dst, int N, int K) {
    const size_t total = (size_t)N * K / 8;                       // one thread moves 8 contiguous bf16
    const int KS = K >> 5;
    for (size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (size_t)gridDim.x * blockDim.x) {
        const int lane = (int)(g & 63);
        const size_t tile = g >> 6;
        const int ks = (int)(tile % KS), nt = (int)(tile / KS);
        const int n = nt * 16 + (lane & 15), k = ks * 32 + (lane >> 4) * 8;
        *reinterpret_cast<uint4 *>(dst + g * 8) = *reinterpret_cast<const uint4 *>(src + (size_t)n * K + k);
    }
}

// ------------------------------------------------------------------------------------------
// Split-K reduction of one (row, col) element, with the Linear's bias and output rounding
// ------------------------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ float reduce_part(const float *__restrict__ part, int S, size_t stride_s, size_t off,
                                             const T *__restrict__ bias, int col) {
    const float *p = part + off;
    float a = 0.f;
    int s = 0;
    for (; s + 8 <= S; s += 8) {                                  // 8 independent loads in flight, fixed add order
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = p[(size_t)(s + u) * stride_s];
#pragma unroll
        for (int u = 0; u < 8; ++u) a += v[u];
    }
    for (; s + 4 <= S; s += 4) {
        float v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = p[(size_t)(s + u) * stride_s];
#pragma unroll
        for (int u = 0; u < 4; ++u) a += v[u];
    }
    for (; s < S; ++s) a += p[(size_t)s * stride_s];
    if (bias) a += to_f(bias[col]);
    return rnd<T>(a);
}

// Same for 4 consecutive columns (16-byte loads); off must be a multiple of 4.
__device__ __forceinline__ f32x4 reduce_part4(const float *__restrict__ part, int S, size_t stride_s, size_t off) {
    const float *p = part + off;
    f32x4 a = {0.f, 0.f, 0.f, 0.f};
    int s = 0;
    for (; s + 8 <= S; s += 8) {
        f32x4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const f32x4 *>(p + (size_t)(s + u) * stride_s);
#pragma unroll
        for (int u = 0; u < 8; ++u) a += v[u];
    }
    for (; s + 4 <= S; s += 4) {
        f32x4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const f32x4 *>(p + (size_t)(s + u) * stride_s);
#pragma unroll
        for (int u = 0; u < 4; ++u) a += v[u];
    }
    for (; s < S; ++s) a += *reinterpret_cast<const f32x4 *>(p + (size_t)s * stride_s);
    return a;
}

// ------------------------------------------------------------------------------------------
// Embedding gather (+ OPT learned positions when there is no project_in)
// ------------------------------------------------------------------------------------------
template <typename T>
__global__ void embed_kernel(RowTab tab, const T *__restrict__ table, int dim, const T *__restrict__ pos_table,
                             int pos_off, T *__restrict__ out, int tiled, int vocab) {
    const int row = blockIdx.x;
    const int pos = tab_pos(tab, row);
    // ids are validated on the host (IndexError, as nn.Embedding raises); the clamp only keeps a caller that goes
    // straight to the C ABI with a bad id from reading outside the table
    const int tok = min(max(tab_tok(tab, row, pos), 0), vocab - 1);
    const T *src = table + (size_t)tok * dim;
    const T *ps = pos_table ? pos_table + (size_t)(pos + pos_off) * dim : nullptr;
    for (int i = threadIdx.x; i < dim; i += blockDim.x) {
        float v = to_f(src[i]);
        if (ps) v = rnd<T>(v + to_f(ps[i]));
        out[tiled ? xoff<T>(row, i, dim) : (size_t)row * dim + i] = from_f<T>(v);   // tiled: feeds project_in
    }
}

// plain rows -> GEMM operand layout (OPT post-LN: the first layer's QKV input is x itself)
template <typename T>
__global__ void to_operand_kernel(const T *__restrict__ x, int H, T *__restrict__ h) {
    const int row = blockIdx.x;
    for (int i = threadIdx.x; i < H; i += blockDim.x) h[xoff<T>(row, i, H)] = x[(size_t)row * H + i];
}

// x = rnd(rnd(sum part) + pos)   (OPT project_in output plus learned positions, modeling_opt.py:669-672)
template <typename T>
__global__ void reduce_addpos_kernel(const float *__restrict__ part, int S, size_t stride_s, int N,
                                     const T *__restrict__ pos_table, RowTab tab, int pos_off, T *__restrict__ out) {
    const int row = blockIdx.x;
    const int pos0 = tab_pos(tab, row) - row;
    for (int i = threadIdx.x; i < N; i += blockDim.x) {
        float v = reduce_part<T>(part, S, stride_s, (size_t)row * N + i, nullptr, i);
        if (pos_table) v = rnd<T>(v + to_f(pos_table[(size_t)(pos0 + row + pos_off) * N + i]));
        out[(size_t)row * N + i] = from_f<T>(v);
    }
}

// ------------------------------------------------------------------------------------------
// Norms.  RMS: modeling_llama.py:84-89 (fp32 statistics, cast, weight multiply in T).
// LayerNorm: nn.LayerNorm (fp32 math, one rounding at the end).
// The row lives in LDS (H floats) between the two passes.
// ------------------------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ void norm_row(const float *__restrict__ xs, int H, const T *__restrict__ w,
                                         const T *__restrict__ b, float eps, int kind, float *red,
                                         T *__restrict__ hbase, int row) {
    // (no fused multiply-adds here: the same statistics are computed by other kernels - small_kernels.h -
    // that must give the same bits, and what the compiler contracts depends on the code around an expression)
#pragma clang fp contract(off)
    float a = 0.f, a2 = 0.f;
    for (int i = threadIdx.x; i < H; i += blockDim.x) {
        const float v = xs[i];
        a += v;
        a2 += v * v;
    }
    if (kind == NORM_RMS) {
        const float var = block_sum(a2, red) / (float)H;
        const float r = rsqrtf(var + eps);
        for (int i = threadIdx.x; i < H; i += blockDim.x) {
            const float y = rnd<T>(to_f(w[i]) * rnd<T>(xs[i] * r));
            hbase[xoff<T>(row, i, H)] = from_f<T>(y);
        }
    } else {
        const float mean = block_sum(a, red) / (float)H;
        float d2 = 0.f;
        for (int i = threadIdx.x; i < H; i += blockDim.x) {
            const float d = xs[i] - mean;
            d2 += d * d;
        }
        const float var = block_sum(d2, red) / (float)H;
        const float r = 1.0f / sqrtf(var + eps);
        for (int i = threadIdx.x; i < H; i += blockDim.x) {
            const float y = rnd<T>((xs[i] - mean) * r * to_f(w[i]) + to_f(b[i]));
            hbase[xoff<T>(row, i, H)] = from_f<T>(y);
        }
    }
}

template <typename T>
__global__ __launch_bounds__(256) void norm_kernel(const T *__restrict__ x, int H, const T *__restrict__ w,
                                                  const T *__restrict__ b, float eps, int kind,
                                                  T *__restrict__ h) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *xs = reinterpret_cast<float *>(smem);
    float *red = xs + H;
    const int row = blockIdx.x;
    for (int i = threadIdx.x; i < H; i += blockDim.x) xs[i] = to_f(x[(size_t)row * H + i]);
    __syncthreads();
    norm_row<T>(xs, H, w, b, eps, kind, red, h, row);
}

// Embedding gather + the first pre-norm in one launch (the row is in LDS between the two): x <- embedding (+ OPT
// positions), h <- norm(x).  Same arithmetic as embed_kernel followed by norm_kernel.
template <typename T>
__global__ __launch_bounds__(256) void embed_norm_kernel(RowTab tab, const T *__restrict__ table, int H,
                                                        const T *__restrict__ pos_table, int pos_off,
                                                        T *__restrict__ x, const T *__restrict__ w,
                                                        const T *__restrict__ b, float eps, int kind,
                                                        T *__restrict__ h, int vocab) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *xs = reinterpret_cast<float *>(smem);
    float *red = xs + H;
    const int row = blockIdx.x;
    const int pos = tab_pos(tab, row);
    const int tok = min(max(tab_tok(tab, row, pos), 0), vocab - 1);   // see embed_kernel
    const T *src = table + (size_t)tok * H;
    const T *ps = pos_table ? pos_table + (size_t)(pos + pos_off) * H : nullptr;
    if constexpr (sizeof(T) == 2) {
        // 16-byte loads, all of a thread's in flight at once (the row sits in a rarely touched page of a table of
        // hundreds of MB: with 2-byte loads in a loop the 5-row launch took 18 us)
        if ((H & 7) == 0) {
            constexpr int MAXV = 4;                               // H <= 8192: 1024 vectors of 8 over 256 threads
            u32x4 ev[MAXV], pv[MAXV];
#pragma unroll
            for (int j = 0; j < MAXV; ++j) {
                const int i = (threadIdx.x + j * 256) * 8;
                if (i < H) {
                    ev[j] = *reinterpret_cast<const u32x4 *>(src + i);
                    if (ps) pv[j] = *reinterpret_cast<const u32x4 *>(ps + i);
                }
            }
#pragma unroll
            for (int j = 0; j < MAXV; ++j) {
                const int i = (threadIdx.x + j * 256) * 8;
                if (i < H) {
                    const T *e8 = reinterpret_cast<const T *>(&ev[j]), *p8 = reinterpret_cast<const T *>(&pv[j]);
                    T o8[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        float v = to_f(e8[u]);
                        if (ps) v = rnd<T>(v + to_f(p8[u]));
                        o8[u] = from_f<T>(v);
                        xs[i + u] = v;
                    }
                    *reinterpret_cast<u32x4 *>(x + (size_t)row * H + i) = *reinterpret_cast<const u32x4 *>(o8);
                }
            }
            __syncthreads();
            norm_row<T>(xs, H, w, b, eps, kind, red, h, row);
            return;
        }
    }
    for (int i = threadIdx.x; i < H; i += blockDim.x) {
        float v = to_f(src[i]);
        if (ps) v = rnd<T>(v + to_f(ps[i]));
        x[(size_t)row * H + i] = from_f<T>(v);
        xs[i] = v;
    }
    __syncthreads();
    norm_row<T>(xs, H, w, b, eps, kind, red, h, row);
}

// x' = rnd(x + rnd(sum part + bias)); then per mode: PRE: x <- x', h <- norm(x');  POST: x,h <- LN(x');
// NONE: x,h <- x'.   (residual adds: modeling_llama.py:440,446; modeling_opt.py:342-347, 363-368)
template <typename T> __device__ __forceinline__ void load4(const T *p, float (&o)[4]);
template <> __device__ __forceinline__ void load4<float>(const float *p, float (&o)[4]) {
    const float4 v = *reinterpret_cast<const float4 *>(p);
    o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w;
}
template <> __device__ __forceinline__ void load4<bf16_t>(const bf16_t *p, float (&o)[4]) {
    const uint2 v = *reinterpret_cast<const uint2 *>(p);
    o[0] = __uint_as_float(v.x << 16); o[1] = __uint_as_float(v.x & 0xffff0000u);
    o[2] = __uint_as_float(v.y << 16); o[3] = __uint_as_float(v.y & 0xffff0000u);
}
template <typename T> __device__ __forceinline__ void store4t(T *p, const float (&v)[4]);
template <> __device__ __forceinline__ void store4t<float>(float *p, const float (&v)[4]) {
    *reinterpret_cast<float4 *>(p) = make_float4(v[0], v[1], v[2], v[3]);
}
template <> __device__ __forceinline__ void store4t<bf16_t>(bf16_t *p, const float (&v)[4]) { store4(p, v[0], v[1], v[2], v[3]); }
template <> __device__ __forceinline__ void load4<f16_t>(const f16_t *p, float (&o)[4]) {
    f16_t h[4];
    *reinterpret_cast<uint2 *>(h) = *reinterpret_cast<const uint2 *>(p);
    o[0] = (float)h[0]; o[1] = (float)h[1]; o[2] = (float)h[2]; o[3] = (float)h[3];
}
template <> __device__ __forceinline__ void store4t<f16_t>(f16_t *p, const float (&v)[4]) { store4(p, v[0], v[1], v[2], v[3]); }

// Register-resident row: every thread owns RG groups of 4 consecutive columns (H <= 4*RG*blockDim), all of its loads
// (split-K slabs, residual, bias, norm weights) are issued up front, and the only block-wide step is the reduction of
// the statistics.  No LDS row buffer.
#define RN_RG 2
#ifdef SD_RN_STAMPS
__device__ long long g_rn_stamps[8];
#define RN_STAMP(i) do { if (threadIdx.x == 0 && blockIdx.x == 0) g_rn_stamps[i] = wall_clock64(); } while (0)
#else
#define RN_STAMP(i) do { } while (0)
#endif
template <typename T>
__global__ __launch_bounds__(1024) void residual_norm_kernel(T *__restrict__ x, const float *__restrict__ part, int S,
                                                            size_t stride_s, int H, const T *__restrict__ bias,
                                                            const T *__restrict__ w, const T *__restrict__ b,
                                                            float eps, int kind, int mode, T *__restrict__ h) {
#pragma clang fp contract(off)                                    // see norm_row
    __shared__ float red[32];
    RN_STAMP(0);
    const int row = blockIdx.x;
    T *xr = x + (size_t)row * H;                                  // residual stream: plain rows; h: GEMM operand layout
    float v[RN_RG][4], wv[RN_RG][4], bv[RN_RG][4];
    bool on[RN_RG];
    f32x4 y4[RN_RG];
    float xin[RN_RG][4], bi[RN_RG][4];
#pragma unroll
    for (int g = 0; g < RN_RG; ++g) {
        const int i = (threadIdx.x + g * blockDim.x) * 4;
        on[g] = i < H;
        if (on[g]) {
            y4[g] = reduce_part4(part, S, stride_s, (size_t)row * H + i);
            load4<T>(xr + i, xin[g]);
            if (bias) load4<T>(bias + i, bi[g]);
            if (mode != RES_NONE) {
                load4<T>(w + i, wv[g]);
                if (kind == NORM_LN) load4<T>(b + i, bv[g]);
            }
        }
    }
    RN_STAMP(1);
    float a = 0.f, a2 = 0.f;
#pragma unroll
    for (int g = 0; g < RN_RG; ++g) {
        if (!on[g]) continue;
        const int i = (threadIdx.x + g * blockDim.x) * 4;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float y = y4[g][j];
            if (bias) y += bi[g][j];
            v[g][j] = rnd<T>(xin[g][j] + rnd<T>(y));
            a += v[g][j];
            a2 += v[g][j] * v[g][j];
        }
        if (mode != RES_POST) store4t<T>(xr + i, v[g]);
        if (mode == RES_NONE) store4t<T>(h + xoff<T>(row, i, H), v[g]);
    }
    RN_STAMP(2);
    if (mode == RES_NONE) return;
    float mean = 0.f, r;
    if (kind == NORM_RMS) {
        r = rsqrtf(block_sum(a2, red) / (float)H + eps);          // modeling_llama.py:84-89
    } else {
        mean = block_sum(a, red) / (float)H;
        float d2 = 0.f;
#pragma unroll
        for (int g = 0; g < RN_RG; ++g)
            if (on[g])
#pragma unroll
                for (int j = 0; j < 4; ++j) { const float d = v[g][j] - mean; d2 += d * d; }
        r = 1.0f / sqrtf(block_sum(d2, red) / (float)H + eps);
    }
    RN_STAMP(3);
#pragma unroll
    for (int g = 0; g < RN_RG; ++g) {
        if (!on[g]) continue;
        const int i = (threadIdx.x + g * blockDim.x) * 4;
        float o[4];
#pragma unroll
        for (int j = 0; j < 4; ++j)
            o[j] = kind == NORM_RMS ? rnd<T>(wv[g][j] * rnd<T>(v[g][j] * r))
                                    : rnd<T>((v[g][j] - mean) * r * wv[g][j] + bv[g][j]);
        store4t<T>(h + xoff<T>(row, i, H), o);
        if (mode == RES_POST) store4t<T>(xr + i, o);
    }
    RN_STAMP(4);
}

// ------------------------------------------------------------------------------------------
// QKV epilogue: split-K reduce, bias, RoPE (llama) / q pre-scale (OPT), K/V appended in place
// into the arena rows pos0..pos0+n_new-1 (replaces torch.cat, modeling_llama.py:337-338).
// grid (n_new, Hq + 2*Hkv), D/2 threads.
// ------------------------------------------------------------------------------------------
template <typename T>
__global__ void qkv_epilogue_kernel(const float *__restrict__ part, int S, size_t stride_s, int Nqkv,
                                    const T *__restrict__ bias, const T *__restrict__ cos_t,
                                    const T *__restrict__ sin_t, int arch, float q_scale, int Hq, int Hkv, int D,
                                    RowTab tab, int layer, T *__restrict__ qbuf, int fused) {
    const int row = blockIdx.x, head = blockIdx.y, d = threadIdx.x, hd = D >> 1;
    if (d >= hd) return;
    const int strm = tab_stream(tab, row), pos = tab_pos(tab, row), max_seq = tab.max_seq[strm];
    T *karena = (T *)tab.kv_base[strm] + (size_t)layer * 2 * Hkv * max_seq * D;
    T *varena = karena + (size_t)Hkv * max_seq * D;
    const bool is_q = head < Hq, is_k = !is_q && head < Hq + Hkv;
    // fused weight layout (Llama): the rows of a q / k head are stored d0, d0+D/2, d1, d1+D/2, ...
    const bool paired = fused && arch == SD_ARCH_LLAMA && (is_q || is_k);
    const int col0 = head * D + (paired ? 2 * d : d), col1 = paired ? col0 + 1 : col0 + hd;
    const float v0 = reduce_part<T>(part, S, stride_s, (size_t)row * Nqkv + col0, bias, col0);
    const float v1 = reduce_part<T>(part, S, stride_s, (size_t)row * Nqkv + col1, bias, col1);
    float o0 = v0, o1 = v1;
    if (arch == SD_ARCH_LLAMA && (is_q || is_k)) {
        const float c = to_f(cos_t[(size_t)pos * hd + d]);
        const float s = to_f(sin_t[(size_t)pos * hd + d]);
        // q*cos + rotate_half(q)*sin with rotate_half = cat(-x2, x1)   (modeling_llama.py:173-188)
        o0 = rnd<T>(rnd<T>(v0 * c) + rnd<T>(-v1 * s));
        o1 = rnd<T>(rnd<T>(v1 * c) + rnd<T>(v0 * s));
    } else if (arch == SD_ARCH_OPT && is_q) {
        o0 = rnd<T>(v0 * q_scale);                                // modeling_opt.py:178
        o1 = rnd<T>(v1 * q_scale);
    }
    if (is_q) {
        T *q = qbuf + (size_t)row * Hq * D + head * D;
        q[d] = from_f<T>(o0);
        q[d + hd] = from_f<T>(o1);
    } else {
        const int kvh = is_k ? head - Hq : head - Hq - Hkv;
        if (tab.kv_fp8) {
            unsigned char *dst8 = (unsigned char *)tab.kv_base[strm] + (size_t)layer * 2 * Hkv * max_seq * D +
                                  ((size_t)(head - Hq) * max_seq + tab_slot(tab, row)) * D;
            const float inv_sc = 1.0f / tab.kv_scale[strm][(size_t)layer * 2 * Hkv + (head - Hq)];
            dst8[d] = to_fp8(o0, inv_sc);
            dst8[d + hd] = to_fp8(o1, inv_sc);
            return;
        }
        T *dst = (is_k ? karena : varena) + ((size_t)kvh * max_seq + tab_slot(tab, row)) * D;
        dst[d] = from_f<T>(o0);
        dst[d + hd] = from_f<T>(o1);
    }
}

// ------------------------------------------------------------------------------------------
// Attention of <= TQ new query rows of one head against the arena (causal).
// Phase 1: one thread per key (its D contiguous elements, 16-byte loads), q rows broadcast from LDS.
// Phase 2: fp32 softmax per row.  Phase 3: P.V with (D/2) lanes across the head dim, keys striped
// over the remaining thread groups, reduced through LDS.
// ------------------------------------------------------------------------------------------

// 8 consecutive storage elements -> fp32 (one 16-byte load for bf16, two for fp32)
__device__ __forceinline__ void load8(const bf16_t *p, float (&o)[8]) {
    const u32x4 v = *reinterpret_cast<const u32x4 *>(p);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        o[2 * i] = __uint_as_float(v[i] << 16);
        o[2 * i + 1] = __uint_as_float(v[i] & 0xffff0000u);
    }
}
__device__ __forceinline__ void load8(const f16_t *p, float (&o)[8]) {
    f16_t h[8];
    *reinterpret_cast<u32x4 *>(h) = *reinterpret_cast<const u32x4 *>(p);
#pragma unroll
    for (int i = 0; i < 8; ++i) o[i] = (float)h[i];
}
__device__ __forceinline__ void load8(const float *p, float (&o)[8]) {
    const float4 a = *reinterpret_cast<const float4 *>(p), b = *reinterpret_cast<const float4 *>(p + 4);
    o[0] = a.x; o[1] = a.y; o[2] = a.z; o[3] = a.w; o[4] = b.x; o[5] = b.y; o[6] = b.z; o[7] = b.w;
}
// 8 storage elements held in registers (as loaded by 16-byte loads) -> fp32
template <typename T> __device__ __forceinline__ void unpack8(const u32x4 (&r)[sizeof(T) == 2 ? 1 : 2], float (&o)[8]);
template <> __device__ __forceinline__ void unpack8<bf16_t>(const u32x4 (&r)[1], float (&o)[8]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        o[2 * i] = __uint_as_float(r[0][i] << 16);
        o[2 * i + 1] = __uint_as_float(r[0][i] & 0xffff0000u);
    }
}
template <> __device__ __forceinline__ void unpack8<f16_t>(const u32x4 (&r)[1], float (&o)[8]) {
    f16_t h[8];
    *reinterpret_cast<u32x4 *>(h) = r[0];
#pragma unroll
    for (int i = 0; i < 8; ++i) o[i] = (float)h[i];
}
template <> __device__ __forceinline__ void unpack8<float>(const u32x4 (&r)[2], float (&o)[8]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) { o[i] = __uint_as_float(r[0][i]); o[4 + i] = __uint_as_float(r[1][i]); }
}

#define ATT_TQ 8
#ifdef SD_ATT_STAMPS
__device__ long long g_att_stamps[16];
#define ATT_STAMP(i) do { if (threadIdx.x == 0 && blockIdx.x == 0 && blockIdx.y == 0) g_att_stamps[i] = wall_clock64(); } while (0)
#else
#define ATT_STAMP(i) do { } while (0)
#endif
// KV8: the arena holds fp8 e4m3 (tab.kv_fp8; 16-bit T and D >= 32 only): keys / values are widened to T in registers
// (exact), the K scale multiplies the scores and the V scale the output.
// The body of one attention workgroup = (head, row group bg, key split bz); WT: the output rows are stored write-through
// (sc1) because another workgroup of the SAME launch reads them (attn_oproj_kernel).
template <typename T, int D, bool KV8, bool TREE, bool WT>
__device__ __forceinline__ void attn_body(const T *__restrict__ qbuf, const RowTab &tab, int layer, T *__restrict__ out, int Hq,
                                          int Hkv, int arch, float inv_sqrt_d, int s_cap, int nsplit,
                                          float *__restrict__ partial, int head, int bg, int bz, char *smem,
                                          long long *wg_stamps = nullptr) {
    // wg_stamps (or NULL): this workgroup's record of a stamped fused launch (fused_kernels.h, AO_STAMP_WGS): slots 1..3
    // (kept in registers and written at the end: a store in front of a barrier has to be acknowledged before the barrier
    //  lets the workgroup through - round trips that would stretch the very phases being timed)
    long long wts[4] = {0, 0, 0, 0};
    auto wstamp = [&](int slot) { if (wg_stamps) wts[slot] = wall_clock64(); };
    static_assert(!KV8 || (sizeof(T) == 2 && D >= 32), "fp8 KV needs a 16-bit model type and the MFMA score path");
    float *qs = reinterpret_cast<float *>(smem);                  // [TQ][D]
    constexpr int ATT_RG = 256 / (D / 8);                         // key groups of the P.V phase, each leaves a partial sum
    float *red = qs + ATT_TQ * D;                                 // [ATT_RG][TQ][D]
    float *sc = red + ATT_RG * ATT_TQ * D;                        // [TQ][s_cap]
    // one group = up to ATT_TQ consecutive rows of one stream; "pos0 + r0" below is the position of its first row
    const int r0 = tab.grp_row0[bg];
    const int nr = tab.grp_n[bg];
    const int strm = tab.grp_stream[bg], max_seq = tab.max_seq[strm];
    const int pos0 = tab.grp_pos[bg] - r0;
    const int kvh = head / (Hq / Hkv);
    const T *karena = (const T *)tab.kv_base[strm] + (size_t)layer * 2 * Hkv * max_seq * D;
    const unsigned char *karena8 = (const unsigned char *)tab.kv_base[strm] + (size_t)layer * 2 * Hkv * max_seq * D;
    float k_scale = 1.f, v_scale = 1.f;
    if constexpr (KV8) {
        k_scale = tab.kv_scale[strm][(size_t)layer * 2 * Hkv + kvh];
        v_scale = tab.kv_scale[strm][(size_t)layer * 2 * Hkv + Hkv + kvh];
    }
    const int tid = threadIdx.x;
    ATT_STAMP(0);
    // long contexts (nsplit > 1, "flash-decoding"): workgroup z takes keys [kb, kb + s_hi) of the group's visible keys and
    // leaves un-normalised partial sums (+ running max and denominator) for attn_combine_kernel; nsplit == 1 is the
    // whole range.  From here on key indices are local to the chunk.
    // tree verify: every row may see all tab.tree_base cached keys and, among the tree rows of this call, those in its
    // mask; the causal arithmetic below then only bounds the key range (all tree rows), visibility comes from `vis`
    constexpr bool tree = TREE;                                   // (a template flag: the causal path pays nothing for it)
    const int s_all = tree ? tab.tree_base + tab.n_rows : pos0 + r0 + nr;   // keys visible to the last row of the group
    const int chunk = nsplit > 1 ? (((s_all + nsplit - 1) / nsplit + 15) & ~15) : s_all;
    const int kb = nsplit > 1 ? bz * chunk : 0;
    const int s_hi = max(0, min(s_all, kb + chunk) - kb);
    const int vis0 = pos0 + r0 - kb;                              // local index of the last key row t = 0 may see
    auto vis = [&](int t, int s) -> bool {                        // may row t of the group see local key s?
        if (!tree) return s <= vis0 + t;
        const int g = s + kb - tab.tree_base;                     // index among the tree rows (< 0: a cached key)
        return g < 0 || ((tab.tree_mask[min(r0 + t, SD_MAX_TREE - 1)] >> g) & 1ull);
    };
    // (an empty chunk - split attention past the visible keys - reads from the head's first key: loads are never branched
    //  around, see below, so their addresses must be valid)
    const int kb_ld = s_hi > 0 ? kb : 0, s_last = max(s_hi, 1) - 1;
    const T *K = karena + ((size_t)kvh * max_seq + kb_ld) * D;
    const T *Vv = karena + ((size_t)(Hkv + kvh) * max_seq + kb_ld) * D;
    const unsigned char *K8 = karena8 + ((size_t)kvh * max_seq + kb_ld) * D;
    const unsigned char *V8 = karena8 + ((size_t)(Hkv + kvh) * max_seq + kb_ld) * D;
    __shared__ float ml[ATT_TQ][2];

    // P.V operand prefetch: the V rows a thread will need do not depend on the scores, so their loads are issued
    // before QK^T / softmax and land while those run (up to VPF keys per thread: 256 keys at D = 128).  On the MFMA
    // path they go out right AFTER the first batch of K tiles, which is what the first phase waits for.
    // NO BRANCH AROUND A LOAD (round 4, seen in the ISA): with `if (s2 < s_hi) load` hipcc's waitcnt insertion lost count
    // at the joins and the first QK^T MFMA waited with vmcnt(3..0) - for all of K AND 13 of the 16 V loads behind it, i.e.
    // the "prefetch" serialised 100 KB in front of the first score.  Every load is now unconditional on a clamped key
    // index (a key past the range re-reads the last one; its value is never used: the P.V loop tests s < s_hi at USE),
    // fp8 rows are widened at use, and the first MFMA waits for its own K tile only.
    constexpr int LPR_ = D / 8, NGRP_ = 256 / LPR_, VPF = 16;
    u32x4 vpre[VPF][sizeof(T) == 2 ? 1 : 2];
    auto issue_v = [&]() {
        const int dp = tid % LPR_, sg = tid / LPR_;
#pragma unroll
        for (int j = 0; j < VPF; ++j) {
            const int s2 = min(sg + j * NGRP_, s_last);
            if constexpr (KV8) {
                const uint2 raw = *reinterpret_cast<const uint2 *>(V8 + (size_t)s2 * D + dp * 8);
                vpre[j][0] = u32x4{raw.x, raw.y, 0u, 0u};         // e4m3 bytes; widened in the P.V loop
            } else {
                const u32x4 *src = reinterpret_cast<const u32x4 *>(Vv + (size_t)s2 * D + dp * 8);
                vpre[j][0] = src[0];
                if (sizeof(T) == 4) vpre[j][sizeof(T) == 2 ? 0 : 1] = src[1];
            }
        }
    };
    if constexpr (!(sizeof(T) == 2 && D >= 32)) issue_v();
    if constexpr (!(sizeof(T) == 2 && D >= 32)) {                 // the MFMA path reads q straight into registers
        for (int i = tid; i < ATT_TQ * D; i += 256) {
            const int t = i / D, d = i - t * D;
            qs[i] = t < nr ? to_f(qbuf[(size_t)(r0 + t) * Hq * D + head * D + d]) : 0.f;
        }
        __syncthreads();
    }

    if constexpr (sizeof(T) == 2 && D >= 32) {
        // bf16: QK^T on the matrix cores.  A = 16 keys x 32 dims straight from the arena, B = q^T (rows >= nr
        // are zero), so lane l ends up with score[key = 16*kt + 4*(l>>4) + j][row = l&15].
        const int w = tid >> 6, lane = tid & 63;
        const int mrow = lane & 15, kq = (lane >> 4) * 8;
        u32x4 qf[D / 32];
#pragma unroll
        for (int dk = 0; dk < D / 32; ++dk)
            qf[dk] = *reinterpret_cast<const u32x4 *>(qbuf + (size_t)(r0 + min(mrow, nr - 1)) * Hq * D + head * D + dk * 32 + kq);
        // key tiles are taken 4 at a time per wave with all of their K loads issued before the first MFMA
        // (a plain loop over tiles would pay one memory round trip per tile)
        u32x4 kf[4][D / 32];
        auto load_k = [&](int kt0) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int row = min((kt0 + 4 * u) * 16 + mrow, s_last);
                if constexpr (KV8) {
                    const unsigned char *kr8 = K8 + (size_t)row * D + kq;
#pragma unroll
                    for (int dk = 0; dk < D / 32; ++dk) {
                        const uint2 raw = *reinterpret_cast<const uint2 *>(kr8 + dk * 32);
                        kf[u][dk] = u32x4{raw.x, raw.y, 0u, 0u};
                    }
                } else {
                    const T *kr = K + (size_t)row * D + kq;
#pragma unroll
                    for (int dk = 0; dk < D / 32; ++dk) kf[u][dk] = *reinterpret_cast<const u32x4 *>(kr + dk * 32);
                }
            }
        };
        auto mul_k = [&](int kt0) {
#pragma unroll
            for (int dk = 0; dk < D / 32; ++dk)                   // rows >= nr of the q operand are zero
#pragma unroll
                for (int i = 0; i < 4; ++i) qf[dk][i] = mrow < nr ? qf[dk][i] : 0u;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int kt = kt0 + 4 * u;
                if (kt * 16 >= s_hi) continue;
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int dk = 0; dk < D / 32; ++dk) {
                    if constexpr (KV8) acc = mfma16<T>(fp8x8_to_16<T>(uint2{kf[u][dk][0], kf[u][dk][1]}), qf[dk], acc);
                    else acc = mfma16<T>(kf[u][dk], qf[dk], acc);
                }
                if (mrow < ATT_TQ) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int s = kt * 16 + (lane >> 4) * 4 + j;
                        if (s < s_hi) {
                            float v = rnd<T>(KV8 ? acc[j] * k_scale : acc[j]);
                            if (arch == SD_ARCH_LLAMA) v = rnd<T>(v * inv_sqrt_d);
                            sc[(size_t)mrow * s_cap + s] = vis(mrow, s) ? v : -INFINITY;
                        }
                    }
                }
            }
        };
        load_k(w);
        asm volatile("" ::: "memory");                            // program order = queue order: K, then V
        issue_v();
        asm volatile("" ::: "memory");
        mul_k(w);
        for (int kt0 = w + 16; kt0 * 16 < s_hi; kt0 += 16) {      // (contexts past 256 keys per workgroup)
            load_k(kt0);
            mul_k(kt0);
        }
    } else
    for (int s = tid; s < s_hi; s += 256) {
        float acc[ATT_TQ];
#pragma unroll
        for (int t = 0; t < ATT_TQ; ++t) acc[t] = 0.f;
        const T *kr = K + (size_t)s * D;
#pragma unroll 2
        for (int d = 0; d < D; d += 8) {
            float kv[8];
            load8(kr + d, kv);
#pragma unroll
            for (int t = 0; t < ATT_TQ; ++t) {
                const float4 qa = *reinterpret_cast<const float4 *>(qs + t * D + d);
                const float4 qb = *reinterpret_cast<const float4 *>(qs + t * D + d + 4);
                acc[t] = fmaf(kv[0], qa.x, acc[t]);
                acc[t] = fmaf(kv[1], qa.y, acc[t]);
                acc[t] = fmaf(kv[2], qa.z, acc[t]);
                acc[t] = fmaf(kv[3], qa.w, acc[t]);
                acc[t] = fmaf(kv[4], qb.x, acc[t]);
                acc[t] = fmaf(kv[5], qb.y, acc[t]);
                acc[t] = fmaf(kv[6], qb.z, acc[t]);
                acc[t] = fmaf(kv[7], qb.w, acc[t]);
            }
        }
#pragma unroll
        for (int t = 0; t < ATT_TQ; ++t) {
            float v = rnd<T>(acc[t]);                             // matmul result in T
            if (arch == SD_ARCH_LLAMA) v = rnd<T>(v * inv_sqrt_d) ;  // scale after the matmul (modeling_llama.py:346)
            sc[(size_t)t * s_cap + s] = vis(t, s) ? v : -INFINITY;
        }
    }
    ATT_STAMP(1);
    wstamp(1);
    __syncthreads();
    ATT_STAMP(2);

    {   // softmax in fp32, result rounded to T (modeling_llama.py:371).  One wave per row; with more than 4 rows one
        // half-wave per row (a 5-row verify group is then one pass instead of two).  The half-wave keeps the two partial
        // sums a full wave's lower and upper lanes would hold and reduces them with the same DPP steps (half_sums), so a
        // row's result does not depend on which form ran (rows of different group sizes stay bit-identical).
        const bool half = nr > 4;
        const int lane = half ? (tid & 31) : (tid & 63), grp = half ? tid >> 5 : tid >> 6, ngrp = half ? 8 : 4;
        const bool upper = (tid & 32) != 0;
        const int LW = half ? 32 : 64;
        constexpr int SREG = 8;                                   // scores a lane keeps in registers (short contexts)
        const bool in_regs = s_hi <= SREG * LW;
        for (int t = grp; t < nr; t += ngrp) {
            float *row = sc + (size_t)t * s_cap;
            const int len = tree ? s_hi : max(0, min(s_hi, vis0 + t + 1));
            float m, sum, lo, hi;
            if (in_regs) {
                // one LDS read and one LDS write per score: the row stays in registers between the three passes.  Each
                // lane owns the keys lane + k*LW; a half-wave lane thereby alternates between the keys that lanes l and
                // l + 32 of a full wave own, and adds them into the same two partial sums as the loops below.
                float v[SREG];
#pragma unroll
                for (int k = 0; k < SREG; ++k) { const int s = lane + k * LW; v[k] = s < len ? row[s] : -INFINITY; }
                float m0 = v[0];
#pragma unroll
                for (int k = 1; k < SREG; ++k) m0 = fmaxf(m0, v[k]);
                half_maxes(m0, lo, hi);
                m = half ? (upper ? hi : lo) : fmaxf(lo, hi);
                // a tree row whose key chunk holds no visible key (split attention): every score is -inf; exp(-inf - 0)
                // = 0 leaves zeros and (m, sum) = (-inf, 0), which attn_combine_kernel skips - not exp(-inf + inf) = NaN
                const float mz = TREE && m == -INFINITY ? 0.f : m;
                float s0 = 0.f, s1 = 0.f;
#pragma unroll
                for (int k = 0; k < SREG; ++k) {
                    const int s = lane + k * LW;
                    if (s < len) {
                        v[k] = expf(v[k] - mz);
                        if (half && (k & 1)) s1 += v[k]; else s0 += v[k];
                    } else v[k] = 0.f;
                }
                if (half) {
                    float lo1, hi1;
                    half_sums(s0, lo, hi);
                    half_sums(s1, lo1, hi1);
                    sum = upper ? hi + hi1 : lo + lo1;
                } else {
                    half_sums(s0, lo, hi);
                    sum = lo + hi;
                }
#pragma unroll
                for (int k = 0; k < SREG; ++k) {
                    const int s = lane + k * LW;
                    if (s < s_hi) row[s] = nsplit > 1 ? v[k] : (s < len ? rnd<T>(v[k] / sum) : 0.f);
                }
                if (nsplit > 1 && lane == 0) { ml[t][0] = m; ml[t][1] = sum; }
                continue;
            }
            if (half) {
                float m0 = -INFINITY;
                for (int s = lane; s < len; s += 32) m0 = fmaxf(m0, row[s]);
                half_maxes(m0, lo, hi);
                m = upper ? hi : lo;
                const float mz = TREE && m == -INFINITY ? 0.f : m;
                float s0 = 0.f, s1 = 0.f;                         // what lanes l and l + 32 of a full wave accumulate
                for (int s = lane; s < len; s += 64) {
                    const float e = expf(row[s] - mz);
                    row[s] = e;
                    s0 += e;
                }
                for (int s = lane + 32; s < len; s += 64) {
                    const float e = expf(row[s] - mz);
                    row[s] = e;
                    s1 += e;
                }
                float lo1, hi1;
                half_sums(s0, lo, hi);
                half_sums(s1, lo1, hi1);
                sum = upper ? hi + hi1 : lo + lo1;
            } else {
                float m0 = -INFINITY;
                for (int s = lane; s < len; s += 64) m0 = fmaxf(m0, row[s]);
                half_maxes(m0, lo, hi);
                m = fmaxf(lo, hi);
                const float mz = TREE && m == -INFINITY ? 0.f : m;
                float s0 = 0.f;
                for (int s = lane; s < len; s += 64) {
                    const float e = expf(row[s] - mz);
                    row[s] = e;
                    s0 += e;
                }
                half_sums(s0, lo, hi);
                sum = lo + hi;                                    // lanes 0-31 first, then 32-63: the half-wave form's order
            }
            if (nsplit > 1) {                                     // keep exp(score - local max); the combine normalises
                for (int s = lane; s < s_hi; s += LW) row[s] = s < len ? row[s] : 0.f;
                if (lane == 0) { ml[t][0] = m; ml[t][1] = sum; }
            } else {
                for (int s = lane; s < s_hi; s += LW) row[s] = s < len ? rnd<T>(row[s] / sum) : 0.f;
            }
        }
    }
    ATT_STAMP(3);
    wstamp(2);
    __syncthreads();
    ATT_STAMP(4);

    {
        // P.V: 16-byte V loads, LPR lanes across one key row, the other lanes of the wave on other keys; keys
        // striped over all thread groups; partial sums folded inside the wave by shuffles, then across the 4
        // waves through LDS.  Instantiated for 1, 3, 5 and ATT_TQ rows so that a decode step (1 row) or a verify
        // group (gamma + 1 rows) does not multiply rows it does not have (1 row: P.V 3.5 -> 1.2 us); per-row
        // arithmetic is the same in all of them.
        constexpr int LPR = D / 8, NGRP = 256 / LPR;
        const int dp = tid % LPR, sg = tid / LPR;
        auto pv = [&](auto nrows) {
            constexpr int NR = decltype(nrows)::value;
            float a[NR][8];
#pragma unroll
            for (int t = 0; t < NR; ++t)
#pragma unroll
                for (int j = 0; j < 8; ++j) a[t][j] = 0.f;
#pragma unroll
            for (int jj = 0; jj < VPF; ++jj) {
                const int s = sg + jj * NGRP;
                if (s < s_hi) {
                    float v[8];
                    if constexpr (KV8) {
                        const u32x4 vw[1] = {fp8x8_to_16<T>(uint2{vpre[jj][0][0], vpre[jj][0][1]})};
                        unpack8<T>(vw, v);
                    } else {
                        unpack8<T>(vpre[jj], v);
                    }
#pragma unroll
                    for (int t = 0; t < NR; ++t) {
                        const float p = sc[(size_t)t * s_cap + s];
#pragma unroll
                        for (int j = 0; j < 8; ++j) a[t][j] = fmaf(p, v[j], a[t][j]);
                    }
                }
            }
#pragma unroll 4
            for (int s = sg + VPF * NGRP; s < s_hi; s += NGRP) {
                float v[8];
                if constexpr (KV8) {
                    const u32x4 vr[1] = {fp8x8_to_16<T>(*reinterpret_cast<const uint2 *>(V8 + (size_t)s * D + dp * 8))};
                    unpack8<T>(vr, v);
                } else {
                    load8(Vv + (size_t)s * D + dp * 8, v);
                }
#pragma unroll
                for (int t = 0; t < NR; ++t) {
                    const float p = sc[(size_t)t * s_cap + s];
#pragma unroll
                    for (int j = 0; j < 8; ++j) a[t][j] = fmaf(p, v[j], a[t][j]);
                }
            }
            // every key group's partial sums go to LDS as they are; the epilogue adds the 256 / LPR of them (folding
            // inside the wave first took ~80 ds_bpermute shuffles per thread)
#pragma unroll
            for (int t = 0; t < NR; ++t)
#pragma unroll
                for (int j = 0; j < 8; ++j) red[((size_t)sg * ATT_TQ + t) * D + dp * 8 + j] = a[t][j];
        };
        if (nr == 1) pv(std::integral_constant<int, 1>{});
        else if (nr <= 3) pv(std::integral_constant<int, 3>{});           // gamma = 2
        else if (nr <= 5) pv(std::integral_constant<int, 5>{});           // gamma = 4, the harness default
        else pv(std::integral_constant<int, ATT_TQ>{});
    }
    ATT_STAMP(5);
    wstamp(3);
    __syncthreads();
    ATT_STAMP(6);
    if (nsplit > 1) {
        float *pz = partial + (((size_t)bg * Hq + head) * nsplit + bz) * ATT_TQ * (D + 2);
        for (int i = tid; i < nr * D; i += 256) {
            const int t = i / D, d = i - t * D;
            float a = 0.f;
#pragma unroll
            for (int g = 0; g < ATT_RG; ++g) a += red[((size_t)g * ATT_TQ + t) * D + d];
            pz[(size_t)t * (D + 2) + d] = KV8 ? a * v_scale : a;
        }
        if (tid < nr) { pz[(size_t)tid * (D + 2) + D] = ml[tid][0]; pz[(size_t)tid * (D + 2) + D + 1] = ml[tid][1]; }
        return;
    }
    if constexpr (sizeof(T) == 2 && D % 4 == 0) {
        // four consecutive head dims per thread: one 8-byte store (inside an 8-element group of the operand layout)
        for (int i = tid; i < nr * (D / 4); i += 256) {
            const int t = i / (D / 4), d = (i - t * (D / 4)) * 4;
            float a[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int g = 0; g < ATT_RG; ++g)
#pragma unroll
                for (int c = 0; c < 4; ++c) a[c] += red[((size_t)g * ATT_TQ + t) * D + d + c];
            if constexpr (KV8) {
#pragma unroll
                for (int c = 0; c < 4; ++c) a[c] *= v_scale;
            }
            store4_maybe_wt<WT>(out + xoff<T>(r0 + t, head * D + d, Hq * D), a[0], a[1], a[2], a[3]);
        }
        if (wg_stamps && tid == 0) { wg_stamps[1] = wts[1]; wg_stamps[2] = wts[2]; wg_stamps[3] = wts[3]; }
    } else {
        for (int i = tid; i < nr * D; i += 256) {
            const int t = i / D, d = i - t * D;
            float a = 0.f;
#pragma unroll
            for (int g = 0; g < ATT_RG; ++g) a += red[((size_t)g * ATT_TQ + t) * D + d];
            out[xoff<T>(r0 + t, head * D + d, Hq * D)] = from_f<T>(KV8 ? a * v_scale : a);
        }
    }
    ATT_STAMP(7);
}

template <typename T, int D, bool KV8 = false, bool TREE = false>
__global__ __launch_bounds__(256) void attn_kernel(const T *__restrict__ qbuf, RowTab tab, int layer,
                                                  T *__restrict__ out, int Hq, int Hkv, int arch,
                                                  float inv_sqrt_d, int s_cap, int nsplit, float *__restrict__ partial) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    attn_body<T, D, KV8, TREE, false>(qbuf, tab, layer, out, Hq, Hkv, arch, inv_sqrt_d, s_cap, nsplit, partial, (int)blockIdx.x,
                                      (int)blockIdx.y, (int)blockIdx.z, smem);
}

// out = sum_z acc_z * exp(m_z - M) / sum_z l_z * exp(m_z - M): merges the nsplit key chunks of one (head, group)
template <typename T, int D>
__global__ __launch_bounds__(128) void attn_combine_kernel(const float *__restrict__ partial, RowTab tab,
                                                          T *__restrict__ out, int Hq, int nsplit) {
    const int head = blockIdx.x, r0 = tab.grp_row0[blockIdx.y], nr = tab.grp_n[blockIdx.y];
    const float *pz = partial + ((size_t)blockIdx.y * Hq + head) * nsplit * ATT_TQ * (D + 2);
    for (int i = threadIdx.x; i < nr * D; i += 128) {
        const int t = i / D, d = i - t * D;
        float M = -INFINITY;
        for (int z = 0; z < nsplit; ++z) M = fmaxf(M, pz[((size_t)z * ATT_TQ + t) * (D + 2) + D]);
        float L = 0.f, a = 0.f;
        for (int z = 0; z < nsplit; ++z) {
            const float *r = pz + ((size_t)z * ATT_TQ + t) * (D + 2);
            const float m = r[D];
            if (m == -INFINITY) continue;                         // chunk with no visible key for this row
            const float w = expf(m - M);
            L += r[D + 1] * w;
            a += r[d] * w;
        }
        out[xoff<T>(r0 + t, head * D + d, Hq * D)] = from_f<T>(a / L);
    }
}

// ------------------------------------------------------------------------------------------
// MLP activation epilogue.  llama: silu(gate) * up (modeling_llama.py:220); OPT: relu(fc1 + b).
// ------------------------------------------------------------------------------------------
template <typename T>
__global__ void act_kernel(const float *__restrict__ part, int S, size_t stride_s, int I, int Ncols, int arch,
                           const T *__restrict__ bias, T *__restrict__ act, int fused) {
    const int row = blockIdx.y;
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= I) return;
    float a;
    if (arch == SD_ARCH_LLAMA) {
        // fused weight layout: 8 gate rows, the same 8 up rows, 8 gate rows, ...
        const int gc = fused ? (c >> 3) * 16 + (c & 7) : c, uc = fused ? gc + 8 : I + c;
        const float g = reduce_part<T>(part, S, stride_s, (size_t)row * Ncols + gc, nullptr, c);
        const float u = reduce_part<T>(part, S, stride_s, (size_t)row * Ncols + uc, nullptr, c);
        const float sg = rnd<T>(g / (1.0f + expf(-g)));
        a = rnd<T>(sg * u);
    } else {
        const float f = reduce_part<T>(part, S, stride_s, (size_t)row * Ncols + c, bias, c);
        a = f > 0.f ? f : 0.f;
    }
    act[xoff<T>(row, c, I)] = from_f<T>(a);
}

// Generic split-K reduce into T rows (OPT project_out) or fp32 logits.
template <typename T>
__global__ void reduce_rows_kernel(const float *__restrict__ part, int S, size_t stride_s, int N,
                                   T *__restrict__ out) {
    const int row = blockIdx.y;
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= N) return;
    out[xoff<T>(row, c, N)] = from_f<T>(reduce_part<T>(part, S, stride_s, (size_t)row * N + c, nullptr, c));
}

template <typename T>
__global__ void logits_kernel(const float *__restrict__ part, int S, size_t stride_s, int V, int round_t,
                              float *__restrict__ out, long ld) {
    const int row = blockIdx.y;
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= V) return;
    float a = 0.f;
    for (int s = 0; s < S; ++s) a += part[(size_t)s * stride_s + (size_t)row * V + c];
    if (round_t) a = rnd<T>(a);
    out[(size_t)row * ld + c] = a;
}
