// Sampling primitives of the speculative-sampling path as HIP kernels for gfx950:
//   norm_probs   <- reference sampling/utils.py:182-210 (norm_logits) + :152-179 (top_k_top_p_filter)
//   sample       <- utils.py:213-233
//   max_fn       <- utils.py:236-245
//   accept scan  <- sampling/speculative_sampling.py:1964-1991
//   resample     <- sampling/speculative_sampling.py:2005-2023
// One workgroup of 1024 threads (16 waves) owns one vocabulary row: the row is staged once in
// LDS (V*4 B <= 128 KiB for Llama's V = 32000) and every later pass reads LDS, not HBM.
#include "common.h"
#include <stdlib.h>

#ifdef SD_STAMPS
__device__ long long g_stamps[32];
#define STAMP(i) do { if (threadIdx.x == 0 && blockIdx.x == 0) g_stamps[i] = wall_clock64(); } while (0)
#else
#define STAMP(i) do { } while (0)
#endif

#define NT 1024            // threads per row workgroup
#define MAX_CAND 1024      // survivors handled by the in-LDS exact sort
#define LDS_ROW_LIMIT (35 * 1024)   // floats; above this the row is re-read from L2 instead of LDS

// ---------------------------------------------------------------------------------------------
// helpers
// ---------------------------------------------------------------------------------------------
// Order-preserving key: a < b  <=>  key(a) < key(b); -0.0 is folded onto +0.0 like a float compare.
__device__ __forceinline__ uint32_t fkey(float f) {
    uint32_t u = __float_as_uint(f + 0.0f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

__device__ __forceinline__ float funkey(uint32_t k) {           // inverse of fkey (the key carries the value exactly)
    return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}

__device__ __forceinline__ int block_sum_i(int v, int *sh) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    __syncthreads();
    if (lane == 0) sh[w] = v;
    __syncthreads();
    int r = 0;
    const int nw = (blockDim.x + 63) >> 6;
    for (int i = 0; i < nw; ++i) r += sh[i];
    return r;
}

__device__ __forceinline__ double block_sum_d(double v, double *sh) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    v = wave_sum_d(v);
    __syncthreads();
    if (lane == 0) sh[w] = v;
    __syncthreads();
    double r = 0.0;
    const int nw = (blockDim.x + 63) >> 6;
    for (int i = 0; i < nw; ++i) r += sh[i];
    return r;
}

// (value, index) arg-max with "first index wins ties" (torch.argmax on CPU).
struct ArgMax {
    float v;
    int i;
};
__device__ __forceinline__ ArgMax am_better(ArgMax a, ArgMax b) {
    // NaN never wins; an empty slot has i == INT_MAX
    if (b.i != 0x7fffffff && (a.i == 0x7fffffff || b.v > a.v || (b.v == a.v && b.i < a.i))) return b;
    return a;
}
__device__ __forceinline__ ArgMax block_argmax(ArgMax a, ArgMax *sh) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        ArgMax b;
        b.v = __shfl_xor(a.v, o, 64);
        b.i = __shfl_xor(a.i, o, 64);
        a = am_better(a, b);
    }
    __syncthreads();
    if (lane == 0) sh[w] = a;
    __syncthreads();
    ArgMax r = sh[0];
    const int nw = (blockDim.x + 63) >> 6;
    for (int i = 1; i < nw; ++i) r = am_better(r, sh[i]);
    return r;
}

// Philox4x32-10 (Salmon et al. 2011), counter-based so that draw (seed, draw_index, element) is
// reproducible regardless of launch geometry.
__device__ __forceinline__ uint4 philox4x32(uint4 c, uint2 k) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint32_t hi0 = __umulhi(0xD2511F53u, c.x), lo0 = 0xD2511F53u * c.x;
        const uint32_t hi1 = __umulhi(0xCD9E8D57u, c.z), lo1 = 0xCD9E8D57u * c.z;
        c = make_uint4(hi1 ^ c.y ^ k.x, lo1, hi0 ^ c.w ^ k.y, lo0);
        k.x += 0x9E3779B9u;
        k.y += 0xBB67AE85u;
    }
    return c;
}
// 23 random bits + 1/2: (k + 0.5) * 2^-23 is exact in fp32 for every k < 2^23 and lies strictly inside (0,1), so
// -log(u) is finite and > 0 (with 24 bits the top value rounds up to 1.0 and the exponential variate becomes 0)
__device__ __forceinline__ float u01_open(uint32_t x) { return ((float)(x >> 9) + 0.5f) * (1.0f / 8388608.0f); }     // (0,1)
__device__ __forceinline__ float u01_half(uint32_t x) { return (float)(x >> 8) * (1.0f / 16777216.0f); }           // [0,1)
__device__ __forceinline__ float philox_exp(uint64_t seed, uint64_t draw, int elem) {
    const uint4 o = philox4x32(make_uint4((uint32_t)(elem >> 2), (uint32_t)draw, (uint32_t)(draw >> 32), 0x5D5Du),
                               make_uint2((uint32_t)seed, (uint32_t)(seed >> 32)));
    const uint32_t x = (elem & 2) ? ((elem & 1) ? o.w : o.z) : ((elem & 1) ? o.y : o.x);
    return -logf(u01_open(x));
}
__device__ __forceinline__ float philox_uniform(uint64_t seed, uint64_t draw) {
    const uint4 o = philox4x32(make_uint4(0u, (uint32_t)draw, (uint32_t)(draw >> 32), 0xACCEu),
                               make_uint2((uint32_t)seed, (uint32_t)(seed >> 32)));
    return u01_half(o.x);
}

// ---- storage-dtype arithmetic (SD_NORM_* in specdec.h).  The reference keeps OPT's logits and the whole norm_logits /
// sample / max_fn chain in the weight dtype (modeling_opt.py:974, utils.py:182-245); torch's CPU kernels for bf16 / fp16
// compute each op in fp32 and round its result to the tensor dtype, so "dtype-faithful" = fp32 math with a rounding
// to the storage type wherever torch materialises a tensor.  dt: 0 fp32 (no rounding), 1 bf16, 2 fp16.
__device__ __forceinline__ float rnd_dt(float x, int dt) {
    return dt == 1 ? (float)(bf16_t)x : (dt == 2 ? (float)(_Float16)x : x);
}
// value of a logit as the model's head leaves it: mode bits 0-1 = rounding of the fp32 accumulator (1 bf16, 2 fp16)
__device__ __forceinline__ float head_round(float x, int mode) { return rnd_dt(x, mode & 3); }
// z = logits / temperature (utils.py:197), rounded when the row lives in a 16-bit dtype (mode bits 4-5)
__device__ __forceinline__ float scaled_logit(float x, float temperature, int mode) {
    return rnd_dt(head_round(x, mode) / temperature, (mode >> 4) & 3);
}

// ---------------------------------------------------------------------------------------------
// norm_probs
// ---------------------------------------------------------------------------------------------
struct NormShared {
    float redf[16];
    int redi[16];
    double redd[16];
    uint32_t redu[16];
    int n_cand;
    int cut_idx;
    uint32_t cut_key;
    int kept;
    float lse;
    int tok;
    uint32_t ckey[MAX_CAND];
    int cidx[MAX_CAND];
    uint32_t skey[MAX_CAND];
    int sidx[MAX_CAND];
};

// Stable descending order of the n (<= MAX_CAND) candidates in ckey/cidx by rank counting -> skey/sidx.
__device__ __forceinline__ void rank_sort(NormShared &S, int n) {
    for (int tid = threadIdx.x; tid < n; tid += blockDim.x) {
        const uint32_t k = S.ckey[tid];
        const int id = S.cidx[tid];
        int rank = 0;
        int j = 0;
        for (; j + 8 <= n; j += 8) {                              // 16 LDS reads in flight per step (broadcast reads)
            uint32_t kj[8];
            int ij[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) { kj[u] = S.ckey[j + u]; ij[u] = S.cidx[j + u]; }
#pragma unroll
            for (int u = 0; u < 8; ++u) rank += (kj[u] > k) || (kj[u] == k && ij[u] < id);
        }
        for (; j < n; ++j) {
            const uint32_t kj = S.ckey[j];
            rank += (kj > k) || (kj == k && S.cidx[j] < id);
        }
        S.skey[rank] = k;
        S.sidx[rank] = id;
    }
    __syncthreads();
}

// k-th largest (k = 1..64) of the 64 lanes' keys, ties broken by lane: every lane gets it.  v_readlane with a constant
// lane index instead of __shfl (ds_bpermute through the LDS crossbar, ~100 cycles each, 64 of them in a row).
__device__ __forceinline__ uint32_t wave_kth_largest(uint32_t mk, int k) {
    const int lane = threadIdx.x & 63;
    int rank = 0;
#pragma unroll
    for (int j = 0; j < 64; ++j) {
        const uint32_t kj = (uint32_t)__builtin_amdgcn_readlane((int)mk, j);
        rank += (kj > mk) || (kj == mk && j < lane);
    }
    const unsigned long long hit = __ballot(rank == k - 1);
    const int src = hit ? (int)(__ffsll((long long)hit) - 1) : 0;
    return (uint32_t)__shfl((int)mk, src, 64);
}

// ---- multi-workgroup candidate extraction (fast path, 1 <= top_k <= 64) -------------------------------
// One workgroup can pull only ~25 GB/s from L2/HBM, so a 128 KiB row costs ~20 us on a single CU.  The row is
// therefore cut into NB_SPLIT chunks: each workgroup normalises its chunk's logits (/temperature), zero-fills its
// part of the output row and keeps the few elements above a local threshold that is guaranteed to lie at or below
// the chunk's k-th largest value (so every element >= the row's k-th largest survives).  norm_probs_kernel then
// works on <= 1024 candidates and scatters the <= k non-zero probabilities.
// Per-row destinations of a batched launch (rows of different streams go to different arenas / token buffers),
// passed by value; at most SD_NORM_BATCH rows per launch.
#define SD_NORM_BATCH 48
struct NormTab {
    float *out[SD_NORM_BATCH];
    int *err[SD_NORM_BATCH];
    int *tok_out[SD_NORM_BATCH];
    int *samp_err[SD_NORM_BATCH];
    const float *noise[SD_NORM_BATCH];
    uint64_t seed[SD_NORM_BATCH], draw[SD_NORM_BATCH];
};

#define NB_SPLIT 16
#define CAND_CAP 192
#define CAND_MAXIT 4
struct CandHdr { int count; int bad; float maxv; int pad; };
struct CandRow { CandHdr hdr[NB_SPLIT]; uint2 cand[NB_SPLIT][CAND_CAP]; };
// The non-zero entries of one normalised row, as norm_probs_kernel's list mode leaves them (n < 0: no list - the row was
// not produced in list mode, holds more than SD_CL_CAP entries, or is invalid): what the residual / bonus sample of the
// native iteration works on instead of two passes over V.
#define SD_CL_CAP 128
struct CandList { int n; int idx[SD_CL_CAP]; float prob[SD_CL_CAP]; };

__global__ __launch_bounds__(256) void norm_cand_kernel(const float *__restrict__ logits, long ld_in, int V,
                                                       float temperature, int top_k, int bf16_round,
                                                       float *__restrict__ out, long ld_out, CandRow *__restrict__ ws,
                                                       NormTab tab, int use_tab) {
    __shared__ uint32_t wk_sh[4];
    __shared__ uint32_t mk_sh[256];
    __shared__ int cnt_sh;
    __shared__ uint2 cand_sh[CAND_CAP];
    __shared__ float redf[16];
    __shared__ int redi[16];
    const int row = blockIdx.y, b = blockIdx.x, tid = threadIdx.x;
    const float4 *x4 = reinterpret_cast<const float4 *>(logits + (size_t)row * ld_in);
    float4 *o4 = reinterpret_cast<float4 *>(use_tab ? tab.out[blockIdx.y] : out + (size_t)row * ld_out);
    const int V4 = V >> 2, C4 = (V4 + NB_SPLIT - 1) / NB_SPLIT;
    const int lo = b * C4, hi = min(V4, lo + C4);
    float4 z[CAND_MAXIT];
    float mt = -INFINITY;
    int bad = 0;
#pragma unroll
    for (int u = 0; u < CAND_MAXIT; ++u) {
        const int i4 = lo + tid + u * 256;
        if (i4 < hi) z[u] = x4[i4];
    }
#pragma unroll
    for (int u = 0; u < CAND_MAXIT; ++u) {
        const int i4 = lo + tid + u * 256;
        if (i4 < hi) {
            float4 v = z[u];
            v.x = scaled_logit(v.x, temperature, bf16_round); v.y = scaled_logit(v.y, temperature, bf16_round);
            v.z = scaled_logit(v.z, temperature, bf16_round); v.w = scaled_logit(v.w, temperature, bf16_round);
            z[u] = v;
            bad |= (v.x != v.x) | (v.y != v.y) | (v.z != v.z) | (v.w != v.w);
            mt = fmaxf(fmaxf(mt, fmaxf(v.x, v.y)), fmaxf(v.z, v.w));
            o4[i4] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    // threshold = the k-th largest of the workgroup's 256 per-thread maxima (rank counting through LDS): at least k
    // elements of the chunk lie at or above it, and only ~1 % of the chunk does
    const int k = min(top_k, V);
    const uint32_t mk = fkey(mt);
    mk_sh[tid] = mk;
    if (tid == 0) { cnt_sh = 0; wk_sh[0] = 0u; }
    __syncthreads();
    int rank = 0;
    for (int j = 0; j < 256; j += 8) {
        uint32_t kj[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) kj[u] = mk_sh[j + u];
#pragma unroll
        for (int u = 0; u < 8; ++u) rank += (kj[u] > mk) || (kj[u] == mk && (j + u) < tid);
    }
    if (rank == k - 1) wk_sh[0] = mk;
    __syncthreads();
    const uint32_t t0 = wk_sh[0];
#pragma unroll
    for (int u = 0; u < CAND_MAXIT; ++u) {
        const int i4 = lo + tid + u * 256;
        if (i4 < hi) {
            const float e[4] = {z[u].x, z[u].y, z[u].z, z[u].w};
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const uint32_t kk = fkey(e[c]);
                if (kk >= t0) {
                    const int slot = atomicAdd(&cnt_sh, 1);
                    if (slot < CAND_CAP) cand_sh[slot] = make_uint2(kk, (uint32_t)(i4 * 4 + c));
                }
            }
        }
    }
    const float bm = block_max(mt, redf);
    bad = block_sum_i(bad, redi);
    __syncthreads();
    const int n = cnt_sh;
    CandRow &R = ws[row];
    if (tid == 0) { R.hdr[b].count = n; R.hdr[b].bad = bad; R.hdr[b].maxv = bm; }
    for (int i = tid; i < min(n, CAND_CAP); i += 256) R.cand[b][i] = cand_sh[i];
}

// norm_logits (+ optionally the sample that follows it in the draft / autoregressive loops, utils.py:213-233).
// Fast path (1 <= top_k <= 64, the harness runs k = 20): three passes over the row - stage, compact, write -
// and everything else on a candidate list of a few dozen entries in LDS.  Any other setting takes the general
// path (bitwise bisection on the ordered key for top-k, mass bisection for a wide top-p).
template <bool SAMPLE>
__global__ __launch_bounds__(NT) void norm_probs_kernel(const float *__restrict__ logits, long ld_in, int V,
                                                       float temperature, int top_k, float top_p, int bf16_round,
                                                       int staged, float *__restrict__ out, long ld_out,
                                                       int *__restrict__ err, const float *__restrict__ noise,
                                                       uint64_t seed, uint64_t draw, int *__restrict__ tok_out,
                                                       int *__restrict__ samp_err, const CandRow *__restrict__ ws,
                                                       NormTab tab, int use_tab, int filter_only,
                                                       const float *__restrict__ tile_max, CandList *__restrict__ cl_out) {
    // tile_max (or NULL): the lm_head's epilogue (EPI_HEAD, model_kernels.h) left the maximum of every 16-column tile
    // of the logits row (NaN when the tile holds one) in tile_max[row][V/16] and cleared the output row; the candidates
    // for 1 <= top_k <= 64 are then found from V/16 maxima + the few tiles that can hold one, without a pass over V.
    // filter_only: write top_k_top_p_filter's result (utils.py:152-179) - the scaled logit where kept, -inf where
    // dropped - instead of the probabilities (the kept set itself, not "probability > 0", which underflow would shrink)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    NormShared &S = *reinterpret_cast<NormShared *>(smem);
    float *zs = reinterpret_cast<float *>(smem + ((sizeof(NormShared) + 15) & ~size_t(15)));
    const int row = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int NTX = blockDim.x, NWX = NTX >> 6;         // 1024 threads, or 256 when the head left tile maxima
    const int dt = (bf16_round >> 4) & 3;               // storage dtype of the row's arithmetic (0 fp32, 1 bf16, 2 fp16)
    const float *x = logits + (size_t)row * ld_in;
    float *o = use_tab ? tab.out[blockIdx.x] : out + (size_t)row * ld_out;
    int *errp = use_tab ? tab.err[blockIdx.x] : (err ? err + row : nullptr);
    if (use_tab) {                                                // batched launch: per-row sampling parameters
        noise = tab.noise[blockIdx.x];
        seed = tab.seed[blockIdx.x];
        draw = tab.draw[blockIdx.x];
        tok_out = tab.tok_out[blockIdx.x];
        samp_err = tab.samp_err[blockIdx.x];
    }
    const uint32_t neg_inf_key = 0x007fffffu;                     // fkey(-inf)
    CandList *cl = cl_out ? cl_out + row : nullptr;
    if (cl && tid == 0) cl->n = -1;                               // (list mode below overwrites it)

    auto load_z = [&](int i) -> float {
        return scaled_logit(x[i], temperature, bf16_round);      // utils.py:197
    };
    STAMP(0);
    // fast entry: norm_cand_kernel already cut the row down to a candidate list (and zero-filled the output row)
    bool fast = false;
    float m = 0.f;
    int bad = 0;
    if (ws) {
        const CandRow &R = ws[row];
        // the 16 chunk headers are fetched by 16 lanes at once (a serial loop would pay 16 L2 round trips)
        if (tid < NB_SPLIT) {
            S.cidx[tid] = R.hdr[tid].count;
            S.sidx[tid] = R.hdr[tid].bad;
            S.redf[tid] = R.hdr[tid].maxv;
        }
        __syncthreads();
        int tot = 0, over = 0;
        float mx = -INFINITY;
        int cnts[NB_SPLIT];
#pragma unroll
        for (int b2 = 0; b2 < NB_SPLIT; ++b2) {
            cnts[b2] = S.cidx[b2];
            over |= (cnts[b2] > CAND_CAP);
            tot += cnts[b2];
            bad |= S.sidx[b2];
            mx = fmaxf(mx, S.redf[b2]);
        }
        __syncthreads();
        if (!over && tot <= MAX_CAND) {
            // one global load per thread: locate (chunk, slot) of candidate `tid` from the 16 counts in registers
            if (tid < tot) {
                int base = 0, blk = 0, off = tid;
#pragma unroll
                for (int b2 = 0; b2 < NB_SPLIT; ++b2) {
                    if (tid >= base && tid < base + cnts[b2]) { blk = b2; off = tid - base; }
                    base += cnts[b2];
                }
                const uint2 e = R.cand[blk][off];
                S.ckey[tid] = e.x;
                S.cidx[tid] = (int)e.y;
            }
            if (tid == 0) { S.n_cand = tot; S.kept = 0; }
            __syncthreads();
            if (tot > 128 && top_k <= 64) {
                // second-level prefilter on the gathered list (one candidate per thread): every chunk kept at least
                // its own top-k, so the union is several hundred entries; the same rank-select trick cuts it to a
                // few dozen before the O(n^2) stable sort
                const uint32_t mk = tid < tot ? S.ckey[tid] : 0u;
                const int k2 = min(top_k, V);
                int rank = 0;
#pragma unroll 8
                for (int j = 0; j < 64; ++j) {
                    const uint32_t kj = (uint32_t)__shfl((int)mk, j, 64);
                    rank += (kj > mk) || (kj == mk && j < lane);
                }
                const unsigned long long hit = __ballot(rank == k2 - 1);
                const uint32_t wk = (uint32_t)__shfl((int)mk, hit ? (int)(__ffsll((long long)hit) - 1) : 0, 64);
                if (lane == 0) S.redu[wv] = wk;
                __syncthreads();
                uint32_t t1 = S.redu[0];
                for (int i = 1; i < NWX; ++i) t1 = max(t1, S.redu[i]);
                if (tid < tot && mk >= t1) {
                    const int slot = atomicAdd(&S.kept, 1);
                    S.skey[slot] = mk;
                    S.sidx[slot] = S.cidx[tid];
                }
                __syncthreads();
                const int n2 = S.kept;
                if (tid < n2) { S.ckey[tid] = S.skey[tid]; S.cidx[tid] = S.sidx[tid]; }
                if (tid == 0) S.n_cand = n2;
                __syncthreads();
            }
            fast = true;
            m = mx;
        } else {
            bad = 0;
        }
    } else if (tile_max) {
        const int NTL = V >> 4;
        const float *tm = tile_max + (size_t)row * NTL;
        constexpr int TPT = 16;                                   // tiles per thread at 256 threads (V <= 65536)
        float zt[TPT];
        float mtl = -INFINITY;
#pragma unroll
        for (int u = 0; u < TPT; ++u) {
            const int t = tid + u * NTX;
            zt[u] = -INFINITY;
            if (t < NTL) {
                float v = tm[t];
                bad |= (v != v);
                v = scaled_logit(v, temperature, bf16_round);    // temperature > 0: monotone, max commutes with it
                zt[u] = v;
                mtl = fmaxf(mtl, v);
            }
        }
        m = block_max(mtl, S.redf);
        bad = block_sum_i(bad, S.redi);
        if (!bad && m != INFINITY && m != -INFINITY) {
            // threshold t0 = the largest, over the 16 waves, of the wave's k-th largest per-thread maximum: at least k
            // tiles - hence k elements - lie at or above it, and every element >= t0 sits in a tile whose maximum is
            const int k = min(top_k, V);
            const uint32_t wk = wave_kth_largest(fkey(mtl), k);
            __syncthreads();
            if (lane == 0) S.redu[wv] = wk;
            if (tid == 0) S.n_cand = 0;
            __syncthreads();
            uint32_t t0 = S.redu[0];
            for (int i = 1; i < NWX; ++i) t0 = max(t0, S.redu[i]);
            // the tiles that can hold a candidate are first compacted into a list (no memory traffic), then their
            // 16 logits each are read by 16 consecutive threads, all tiles in flight at once: a thread walking its own
            // tiles would pay one dependent global round trip per tile, with the whole wave waiting on any lane's
            int *tlist = S.sidx;                                  // (the sorted-list arrays are free until rank_sort)
            int *tcount = &S.kept;
            if (tid == 0) *tcount = 0;
            __syncthreads();
#pragma unroll
            for (int u = 0; u < TPT; ++u) {
                const int t = tid + u * NTX;
                if (t < NTL && fkey(zt[u]) >= t0) {
                    const int slot = atomicAdd(tcount, 1);
                    if (slot < MAX_CAND) tlist[slot] = t;
                }
            }
            __syncthreads();
            const int ntl = *tcount;
            if (ntl <= MAX_CAND) {
                for (int j = tid; j < ntl * 16; j += NTX) {
                    const int idx = tlist[j >> 4] * 16 + (j & 15);
                    const uint32_t kk = fkey(scaled_logit(x[idx], temperature, bf16_round));
                    if (kk >= t0) {
                        const int slot = atomicAdd(&S.n_cand, 1);
                        if (slot < MAX_CAND) { S.ckey[slot] = kk; S.cidx[slot] = idx; }
                    }
                }
            } else if (tid == 0) {
                S.n_cand = MAX_CAND + 1;
            }
            __syncthreads();
            if (S.n_cand <= MAX_CAND) fast = true;               // else pathological ties: the general path below
        }
    }
    STAMP(1);
    const bool use_lds = staged && !fast;
    auto Z = [&](int i) -> float { return use_lds ? zs[i] : load_z(i); };

    float mt = -INFINITY;
    if (!fast) {
    // pass 0: stage, row max, NaN detection (16-byte loads, all of a thread's loads in flight at once)
    if (staged && (V & 3) == 0 && ((reinterpret_cast<uintptr_t>(x) & 15) == 0)) {
        const int V4 = V >> 2;
        for (int i0 = tid; i0 < V4; i0 += NTX * 8) {
            float4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i4 = i0 + u * NTX;
                if (i4 < V4) v[u] = reinterpret_cast<const float4 *>(x)[i4];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i4 = i0 + u * NTX;
                if (i4 < V4) {
                    float4 z = v[u];
                    z.x = scaled_logit(z.x, temperature, bf16_round); z.y = scaled_logit(z.y, temperature, bf16_round);
                    z.z = scaled_logit(z.z, temperature, bf16_round); z.w = scaled_logit(z.w, temperature, bf16_round);
                    reinterpret_cast<float4 *>(zs)[i4] = z;
                    bad |= (z.x != z.x) | (z.y != z.y) | (z.z != z.z) | (z.w != z.w);
                    mt = fmaxf(fmaxf(mt, fmaxf(z.x, z.y)), fmaxf(z.z, z.w));
                }
            }
        }
        __syncthreads();                                          // (any partition of the row works for the prefilter)
    } else {
        for (int i = tid; i < V; i += NTX) {
            const float z = load_z(i);
            if (staged) zs[i] = z;
            bad |= (z != z);
            mt = fmaxf(mt, z);
        }
    }
    m = block_max(mt, S.redf);
    bad = block_sum_i(bad, S.redi);
    }
    if (!filter_only && (bad || m == INFINITY || m == -INFINITY)) { // exp(log_softmax) would hold NaN (utils.py:203)
        for (int i = tid; i < V; i += NTX) o[i] = __uint_as_float(0x7fc00000u);
        if (tid == 0) {
            if (errp) *errp = 1;
            if (SAMPLE && samp_err) *samp_err = 1;
        }
        return;
    }

    STAMP(2);
    uint32_t kth = 0u;                                            // key >= 0 keeps everything
    bool have_list = false;                                       // skey/sidx hold a superset of the survivors, sorted
    int n_list = 0, n_surv = V;
    if (top_k > 0) {
        const int k = min(top_k, V);
        if (fast) {
            const int n = S.n_cand;
            rank_sort(S, n);
            kth = S.skey[k - 1];
            int ns = k;
            while (ns < n && S.skey[ns] == kth) ++ns;
            n_surv = ns;
            n_list = n;
            have_list = true;
        } else if (k <= 64) {
            // ---- prefilter: a threshold t0 with at least k elements above it.  Each wave takes the k-th
            // largest of its 64 per-thread maxima (rank counting over readlanes); t0 = the largest of those.
            const uint32_t mk = fkey(mt);
            int rank = 0;
#pragma unroll 8
            for (int j = 0; j < 64; ++j) {
                const uint32_t kj = (uint32_t)__shfl((int)mk, j, 64);
                rank += (kj > mk) || (kj == mk && j < lane);
            }
            const unsigned long long hit = __ballot(rank == k - 1);
            const uint32_t wk = (uint32_t)__shfl((int)mk, hit ? (int)(__ffsll((long long)hit) - 1) : 0, 64);
            __syncthreads();
            if (lane == 0) S.redu[wv] = wk;
            if (tid == 0) S.n_cand = 0;
            __syncthreads();
            uint32_t t0 = S.redu[0];
            for (int i = 1; i < NWX; ++i) t0 = max(t0, S.redu[i]);
            for (int i = tid; i < V; i += NTX) {
                const uint32_t kk = fkey(Z(i));
                if (kk >= t0) {
                    const int slot = atomicAdd(&S.n_cand, 1);
                    if (slot < MAX_CAND) { S.ckey[slot] = kk; S.cidx[slot] = i; }
                }
            }
            __syncthreads();
            const int n = S.n_cand;
            if (n <= MAX_CAND) {                                  // else: pathological ties, take the general path
                rank_sort(S, n);
                kth = S.skey[k - 1];                              // exact k-th largest value (t0 <= kth by construction)
                int ns = k;
                while (ns < n && S.skey[ns] == kth) ++ns;         // ties at the k-th value stay (utils.py:169)
                n_surv = ns;
                n_list = n;
                have_list = true;
            }
        }
        if (!have_list) {
            // general top-k: bitwise bisection on the ordered key
            uint32_t prefix = 0u;
            for (int bit = 31; bit >= 0; --bit) {
                const uint32_t cand = prefix | (1u << bit);
                int c = 0;
                for (int i = tid; i < V; i += NTX) c += (fkey(Z(i)) >= cand);
                c = block_sum_i(c, S.redi);
                if (c >= k) prefix = cand;
            }
            kth = prefix;
        }
    }

    STAMP(3);
    // top-p (utils.py:170-178).  keep(i) <=> key_i > cut_key || (key_i == cut_key && i <= cut_idx)
    uint32_t cut_key = kth;
    int cut_idx = 0x7fffffff;
    int kept = -1;                                                // >= 0: the kept set is skey/sidx[0..kept)
    if (have_list) kept = n_surv;
    // exp(z_i - m) of the sorted survivors, one per thread (the candidate arrays are free after the sort); the serial
    // sums below then only add, in the same order as before
    float *ev = reinterpret_cast<float *>(S.ckey);
    auto fill_ev = [&](int n) {
        __syncthreads();
        for (int c = tid; c < n; c += NTX) ev[c] = expf(funkey(S.skey[c]) - m);
        __syncthreads();
    };
    if (have_list) fill_ev(n_surv);
    if (top_p > 0.0f) {
        if (!have_list) {
            int c = 0;
            for (int i = tid; i < V; i += NTX) {
                const uint32_t kk = fkey(Z(i));
                c += (kk >= kth && kk > neg_inf_key);
            }
            const int n_fin = block_sum_i(c, S.redi);
            if (n_fin <= MAX_CAND) {
                __syncthreads();
                if (tid == 0) S.n_cand = 0;
                __syncthreads();
                for (int i = tid; i < V; i += NTX) {
                    const uint32_t kk = fkey(Z(i));
                    if (kk >= kth && kk > neg_inf_key) {
                        const int slot = atomicAdd(&S.n_cand, 1);
                        S.ckey[slot] = kk;
                        S.cidx[slot] = i;
                    }
                }
                __syncthreads();
                rank_sort(S, S.n_cand);
                n_surv = S.n_cand;
                have_list = true;
                fill_ev(n_surv);
            }
        }
        if (have_list) {
            if (tid == 0) {
                // entries that are already -inf carry no mass and sort last: leave them out
                int nf = n_surv;
                while (nf > 1 && S.skey[nf - 1] <= neg_inf_key) --nf;
                float denom = 0.f;                                // softmax denominator over the survivors
                for (int i = 0; i < nf; ++i) denom += ev[i];
                // torch.cumsum accumulates float32 inputs in double and rounds each prefix to float32; bf16 / fp16 inputs
                // (softmax output rounded to the dtype) accumulate in float, each prefix is rounded to the dtype and
                // compared with top_p rounded to it (utils.py:171-173 on a 16-bit tensor)
                int kp = 0;
                if (dt == 0) {
                    double cum = 0.0;
                    for (int i = 0; i < nf; ++i) {
                        if (i > 0 && (float)cum > top_p) break;   // shifted filter: the crossing token stays
                        cum += (double)(ev[i] / denom);
                        kp = i + 1;
                    }
                } else {
                    float cum = 0.f;
                    const float tp = rnd_dt(top_p, dt);
                    for (int i = 0; i < nf; ++i) {
                        if (i > 0 && rnd_dt(cum, dt) > tp) break;
                        cum += rnd_dt(ev[i] / denom, dt);
                        kp = i + 1;
                    }
                }
                S.kept = kp;
            }
            __syncthreads();
            kept = S.kept;
        } else {
            // General path (no / very wide top-k): smallest existing value v* whose first tie member is kept,
            // i.e. float(mass strictly above v*) <= top_p, by bisection on the key; then how many of its tie
            // members fit.  Mass is accumulated in double like torch.cumsum does.
            float part = 0.f;
            for (int i = tid; i < V; i += NTX) {
                const float z = Z(i);
                if (fkey(z) >= kth) part += expf(z - m);
            }
            const float denom = block_sum(part, S.redf);
            uint32_t lo = (kth > neg_inf_key + 1u) ? kth : neg_inf_key + 1u, hi = fkey(m);
            while (lo < hi) {
                const uint32_t mid = lo + ((hi - lo) >> 1);
                double g = 0.0;
                for (int i = tid; i < V; i += NTX) {
                    const float z = Z(i);
                    if (fkey(z) > mid) g += (double)rnd_dt(expf(z - m) / denom, dt);
                }
                g = block_sum_d(g, S.redd);
                if (!(rnd_dt((float)g, dt) > rnd_dt(top_p, dt))) hi = mid; else lo = mid + 1u;
            }
            uint32_t best = 0xffffffffu;                          // v* = smallest existing key >= lo
            for (int i = tid; i < V; i += NTX) {
                const uint32_t kk = fkey(Z(i));
                if (kk >= lo) best = min(best, kk);
            }
#pragma unroll
            for (int of = 32; of > 0; of >>= 1) best = min(best, (uint32_t)__shfl_xor((int)best, of, 64));
            __syncthreads();
            if (lane == 0) S.redu[wv] = best;
            __syncthreads();
            best = S.redu[0];
            for (int i2 = 1; i2 < NWX; ++i2) best = min(best, S.redu[i2]);
            const uint32_t vstar = best;
            double g = 0.0;
            int e = 0;
            float pv = 0.f;
            for (int i = tid; i < V; i += NTX) {
                const float z = Z(i);
                const uint32_t kk = fkey(z);
                if (kk > vstar) g += (double)rnd_dt(expf(z - m) / denom, dt);
                if (kk == vstar) { e += 1; pv = rnd_dt(expf(z - m) / denom, dt); }
            }
            g = block_sum_d(g, S.redd);
            e = block_sum_i(e, S.redi);
            pv = block_max(pv, S.redf);
            int mstar = 1;                                        // first tie member is kept by construction
            {
                double cum = g + (double)pv;
                while (mstar < e && !(rnd_dt((float)cum, dt) > rnd_dt(top_p, dt))) { cum += (double)pv; ++mstar; }
            }
            cut_key = vstar;
            cut_idx = 0x7fffffff;
            if (mstar < e) {                                      // ties straddle the cut: keep the mstar lowest indices
                int lo_i = 0, hi_i = V - 1;
                while (lo_i < hi_i) {
                    const int mid = lo_i + ((hi_i - lo_i) >> 1);
                    int c2 = 0;
                    for (int i = tid; i < V; i += NTX) c2 += (i <= mid && fkey(Z(i)) == vstar);
                    c2 = block_sum_i(c2, S.redi);
                    if (c2 >= mstar) hi_i = mid; else lo_i = mid + 1;
                }
                cut_idx = lo_i;
            }
        }
    }

    STAMP(4);
    // probs = exp(log_softmax(filtered))  (utils.py:199), then optionally sample (utils.py:213-233)
    if (kept >= 0) {
        // list mode: the kept set is the first `kept` entries of the sorted candidate list
        if (tid == 0) {
            float sum = 0.f;
            for (int i = 0; i < kept; ++i) sum += ev[i];
            // 16-bit rows: torch's log_softmax keeps the row sum and its log in the tensor dtype
            S.lse = dt ? rnd_dt(logf(rnd_dt(sum, dt)), dt) : logf(sum);
        }
        const float fillv = filter_only ? -INFINITY : 0.f;
        if (fast) {
            // norm_cand_kernel zero-filled the row
        } else if ((V & 3) == 0 && ((reinterpret_cast<uintptr_t>(o) & 15) == 0)) {
            for (int i4 = tid; i4 < (V >> 2); i4 += NTX) reinterpret_cast<float4 *>(o)[i4] = make_float4(fillv, fillv, fillv, fillv);
        } else {
            for (int i = tid; i < V; i += NTX) o[i] = fillv;
        }
        __syncthreads();
        const float lse = S.lse;
        auto PL = [&](uint32_t key) -> float {                    // exp(log_softmax): both results live in the row dtype
            const float ls = (funkey(key) - m) - lse;
            return dt ? rnd_dt(expf(rnd_dt(ls, dt)), dt) : expf(ls);
        };
        for (int c = tid; c < kept; c += NTX)
            o[S.sidx[c]] = filter_only ? funkey(S.skey[c]) : PL(S.skey[c]);
        if (cl && !filter_only && kept <= SD_CL_CAP) {
            for (int c = tid; c < kept; c += NTX) { cl->idx[c] = S.sidx[c]; cl->prob[c] = PL(S.skey[c]); }
            if (tid == 0) cl->n = kept;                           // (same thread as the -1 above: program order)
        }
        STAMP(5);
        if (SAMPLE) {
            // multinomial(p, 1) == argmax_i p_i / e_i over the support (zero-probability entries give 0 and never
            // win); first index wins ties; fix-up for a pick below 1e-9 (utils.py:228-230).  One lane per kept entry.
            ArgMax br = {0.f, 0x7fffffff};
            for (int c = tid; c < kept; c += NTX) {
                const int id = S.sidx[c];
                const float p = PL(S.skey[c]);
                if (p > 0.f) br = am_better(br, ArgMax{rnd_dt(p / (noise ? noise[id] : philox_exp(seed, draw, id)), dt), id});
            }
            ArgMax *sha = reinterpret_cast<ArgMax *>(S.ckey);
            br = block_argmax(br, sha);
            if (tid == 0) {
                int tok = br.i;
                float ptok = 0.f;
                for (int i = 0; i < kept; ++i)
                    if (S.sidx[i] == tok) ptok = PL(S.skey[i]);
                if (tok == 0x7fffffff || ptok < 1e-9f) tok = S.sidx[0];   // argmax(probs): list head (lowest index among ties)
                *tok_out = tok;
                if (samp_err) *samp_err = 0;
            }
        }
    } else {
        float part = 0.f;
        for (int i = tid; i < V; i += NTX) {
            const float z = Z(i);
            const uint32_t kk = fkey(z);
            if (kk > cut_key || (kk == cut_key && i <= cut_idx)) part += expf(z - m);
        }
        const float rowsum = block_sum(part, S.redf);
        const float lse = dt ? rnd_dt(logf(rnd_dt(rowsum, dt)), dt) : logf(rowsum);
        auto P = [&](int i) -> float {
            const float z = Z(i);
            const uint32_t kk = fkey(z);
            const bool keep = kk > cut_key || (kk == cut_key && i <= cut_idx);
            if (filter_only) return keep ? z : -INFINITY;
            if (!keep) return 0.0f;
            return dt ? rnd_dt(expf(rnd_dt((z - m) - lse, dt)), dt) : expf((z - m) - lse);
        };
        for (int i = tid; i < V; i += NTX) o[i] = P(i);
        if (SAMPLE) {
            ArgMax br = {0.f, 0x7fffffff}, bw = {0.f, 0x7fffffff};
            for (int i = tid; i < V; i += NTX) {
                const float p = P(i);
                if (!(p > 0.f)) continue;
                const float r = rnd_dt(p / (noise ? noise[i] : philox_exp(seed, draw, i)), dt);
                if (br.i == 0x7fffffff || r > br.v) br = {r, i};
                if (bw.i == 0x7fffffff || p > bw.v) bw = {p, i};
            }
            ArgMax *sha = reinterpret_cast<ArgMax *>(S.ckey);
            br = block_argmax(br, sha);
            bw = block_argmax(bw, sha);
            if (tid == 0) {
                int tok = br.i;
                if (P(tok) < 1e-9f) tok = bw.i;
                *tok_out = tok;
                if (samp_err) *samp_err = 0;
            }
        }
    }
    STAMP(6);
    if (tid == 0 && errp) *errp = 0;
    (void)n_list;
}

// ---------------------------------------------------------------------------------------------
// weighted arg-max sampling core shared by sample / resample
// ---------------------------------------------------------------------------------------------
struct SampleShared {
    float redf[16];
    int redi[16];
    ArgMax reda[16];
};

// W(i) -> weight, E(i) -> Exp(1) noise.  Returns the token; *status: 0 ok, 1 invalid, 2 all-zero.
template <typename WF, typename EF>
__device__ __forceinline__ int sample_core(int V, WF W, EF E, SampleShared &S, int *status, int dt = 0) {
    const int tid = threadIdx.x;
    ArgMax best_r = {0.f, 0x7fffffff}, best_w = {0.f, 0x7fffffff};
    int bad = 0, pos = 0;
    for (int i = tid; i < V; i += NT) {
        const float w = W(i);
        bad |= !(w >= 0.0f) || (w == INFINITY);                   // negative, NaN or Inf: multinomial's validity check
        pos |= (w > 0.0f);
        const float r = w > 0.0f ? rnd_dt(w / E(i), dt) : 0.0f;   // IEEE division, as at::div (result in the row dtype); 0/e = 0
        if (best_r.i == 0x7fffffff || r > best_r.v) best_r = {r, i};
        if (best_w.i == 0x7fffffff || w > best_w.v) best_w = {w, i};
    }
    bad = block_sum_i(bad, S.redi);
    pos = block_sum_i(pos, S.redi);
    if (bad) { *status = 1; return 0; }
    if (!pos) { *status = 2; return 0; }
    best_r = block_argmax(best_r, S.reda);
    best_w = block_argmax(best_w, S.reda);
    *status = 0;
    int tok = best_r.i;
    if (W(tok) < 1e-9f) tok = best_w.i;                           // utils.py:228-230
    return tok;
}

__global__ __launch_bounds__(NT) void sample_kernel(const float *__restrict__ probs, int V,
                                                   const float *__restrict__ noise, uint64_t seed, uint64_t draw,
                                                   int *__restrict__ tok_out, int *__restrict__ err, int dt) {
    __shared__ SampleShared S;
    int status;
    const int tok = sample_core(
        V, [&](int i) { return probs[i]; },
        [&](int i) { return noise ? noise[i] : philox_exp(seed, draw, i); }, S, &status, dt);
    if (threadIdx.x == 0) {
        if (status == 0) *tok_out = tok;
        if (err) *err = status;
    }
}

// max_fn's denominator (utils.py:245): sum + 1e-6; on a 16-bit row the sum is a tensor of that dtype and so is the result
__device__ __forceinline__ float max_fn_denom(float sum, int dt) {
    return dt ? rnd_dt(rnd_dt(sum, dt) + 1e-6f, dt) : sum + 1e-6f;
}

__global__ __launch_bounds__(NT) void max_fn_kernel(const float *__restrict__ p, const float *__restrict__ q, int V,
                                                   float *__restrict__ out, int dt) {
    __shared__ float red[16];
    float part = 0.f;
    for (int i = threadIdx.x; i < V; i += NT) {
        const float d = q ? rnd_dt(p[i] - q[i], dt) : p[i];
        part += d > 0.f ? d : 0.f;
    }
    const float denom = max_fn_denom(block_sum(part, red), dt);
    for (int i = threadIdx.x; i < V; i += NT) {
        const float d = q ? rnd_dt(p[i] - q[i], dt) : p[i];
        out[i] = rnd_dt((d > 0.f ? d : 0.f) / denom, dt);
    }
}

// ---------------------------------------------------------------------------------------------
// accept scan + resample
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void accept_scan_body(const float *__restrict__ p_hist, const float *__restrict__ q_hist,
                                                 long ld, const int32_t *__restrict__ seq, int L, int gamma,
                                                 const float *__restrict__ r, uint64_t seed, uint64_t draw,
                                                 sd_accept_result *__restrict__ out) {
    const int i = threadIdx.x;
    bool reject = false;
    float p = 0.f, q = 1.f;
    int j = -1;
    if (i < gamma) {
        j = seq[L + i];
        p = p_hist[(size_t)(L + i - 1) * ld + j];
        q = q_hist[(size_t)(L + i - 1) * ld + j];
        const float ratio = (float)((double)p / (double)q);       // python double ratio, fp32 compare (:1981)
        const float ri = r ? r[i] : philox_uniform(seed, draw + (uint64_t)i);
        reject = ri > ratio;
    }
    const unsigned long long mask = __ballot(reject);
    const int first = mask ? (__ffsll((long long)mask) - 1) : gamma;
    if (i < 16) {
        out->p_at[i] = i < gamma ? p : 0.f;
        out->q_at[i] = i < gamma ? q : 0.f;
        out->drafted[i] = j;
    }
    if (i == 0) {
        out->n_accepted = first;
        out->n = L + first - 1;
        out->flags = (first == gamma) ? 4 : 0;
        out->next_token = -1;
    }
}

__global__ __launch_bounds__(64) void accept_scan_kernel(const float *__restrict__ p_hist,
                                                        const float *__restrict__ q_hist, long ld,
                                                        const int32_t *__restrict__ seq, int L, int gamma,
                                                        const float *__restrict__ r, uint64_t seed, uint64_t draw,
                                                        sd_accept_result *__restrict__ out) {
    accept_scan_body(p_hist, q_hist, ld, seq, L, gamma, r, seed, draw, out);
}

// Batched form: one workgroup per stream, per-stream arguments passed by value.
#define SD_ACCEPT_BATCH 16
struct AcceptTab {
    const float *p_hist[SD_ACCEPT_BATCH], *q_hist[SD_ACCEPT_BATCH];
    int32_t *seq[SD_ACCEPT_BATCH];
    const float *r[SD_ACCEPT_BATCH], *noise[SD_ACCEPT_BATCH];
    sd_accept_result *res[SD_ACCEPT_BATCH];
    const int *err_flags[SD_ACCEPT_BATCH];
    uint64_t seed[SD_ACCEPT_BATCH], draw_scan[SD_ACCEPT_BATCH], draw_res[SD_ACCEPT_BATCH];
    int L[SD_ACCEPT_BATCH], n_err[SD_ACCEPT_BATCH];
};

__global__ __launch_bounds__(64) void accept_scan_batch_kernel(AcceptTab t, long ld, int gamma) {
    const int b = blockIdx.x;
    accept_scan_body(t.p_hist[b], t.q_hist[b], ld, t.seq[b], t.L[b], gamma, t.r[b], t.seed[b], t.draw_scan[b], t.res[b]);
}

template <bool PLAIN_FALLBACK = false>
__device__ __forceinline__ void resample_body(const float *__restrict__ p_hist, const float *__restrict__ q_hist,
                                              long ld, int V, int32_t *__restrict__ seq, int gamma,
                                              const float *__restrict__ noise, uint64_t seed, uint64_t draw,
                                              sd_accept_result *__restrict__ res, int32_t *__restrict__ seq_len,
                                              const int *__restrict__ err_flags, int n_err, int dt = 0) {
    __shared__ SampleShared S;
    __shared__ float red[16];
    const int n = res->n;
    const bool rejected = res->n_accepted < gamma;
    const float *p = p_hist + (size_t)n * ld;
    const float *q = q_hist + (size_t)n * ld;
    auto E = [&](int i) { return noise ? noise[i] : philox_exp(seed, draw, i); };
    int status = 0, tok = 0, flags = 0;
    if (rejected) {
        float part = 0.f;
        for (int i = threadIdx.x; i < V; i += NT) {
            const float d = rnd_dt(p[i] - q[i], dt);
            part += d > 0.f ? d : 0.f;
        }
        const float denom = max_fn_denom(block_sum(part, red), dt);   // max_fn, utils.py:236-245
        tok = sample_core(
            V, [&](int i) { const float d = rnd_dt(p[i] - q[i], dt); return rnd_dt((d > 0.f ? d : 0.f) / denom, dt); }, E, S,
            &status, dt);
        if (status != 0) {                                        // residual sample raised -> sample(max_fn(p_n)) (:2009-2010)
            flags |= 1;
            part = 0.f;
            for (int i = threadIdx.x; i < V; i += NT) part += p[i] > 0.f ? p[i] : 0.f;
            const float denom2 = max_fn_denom(block_sum(part, red), dt);
            if (PLAIN_FALLBACK)                                   // multi_speculative_sampling: sample(p_n) (:1666-1668)
                tok = sample_core(V, [&](int i) { return p[i]; }, E, S, &status, dt);
            else
                tok = sample_core(
                    V, [&](int i) { const float d = p[i]; return rnd_dt((d > 0.f ? d : 0.f) / denom2, dt); }, E, S, &status, dt);
        }
    } else {
        tok = sample_core(V, [&](int i) { return p[i]; }, E, S, &status, dt);   // bonus token from p_last (:2019)
    }
    if (threadIdx.x == 0) {
        if (status != 0) flags |= 2;
        for (int i = 0; i < n_err; ++i)                           // norm / sample error words of this iteration
            if (err_flags[i]) flags |= 8;
        res->flags |= flags;
        res->next_token = status == 0 ? tok : -1;
        if (status == 0) seq[n + 1] = tok;
        if (seq_len) *seq_len = n + 2;
    }
}

__global__ __launch_bounds__(NT) void resample_kernel(const float *__restrict__ p_hist,
                                                     const float *__restrict__ q_hist, long ld, int V,
                                                     int32_t *__restrict__ seq, int gamma,
                                                     const float *__restrict__ noise, uint64_t seed, uint64_t draw,
                                                     sd_accept_result *__restrict__ res,
                                                     int32_t *__restrict__ seq_len,
                                                     const int *__restrict__ err_flags, int n_err, int dt) {
    resample_body(p_hist, q_hist, ld, V, seq, gamma, noise, seed, draw, res, seq_len, err_flags, n_err, dt);
}

__global__ __launch_bounds__(NT) void multi_resample_kernel(const float *__restrict__ p_hist,
                                                           const float *__restrict__ q_hist, long ld, int V,
                                                           int32_t *__restrict__ seq, int gamma,
                                                           const float *__restrict__ noise, uint64_t seed,
                                                           uint64_t draw, sd_accept_result *__restrict__ res, int dt) {
    resample_body<true>(p_hist, q_hist, ld, V, seq, gamma, noise, seed, draw, res, (int32_t *)nullptr,
                        (const int *)nullptr, 0, dt);
}

// Width-w acceptance (multi_speculative_sampling): gathers in parallel, the data-dependent scan on one thread.
struct MultiTab {
    const float *p_hist[16], *q_hist[16];
    const int32_t *seq[16];
};

__global__ __launch_bounds__(256) void accept_multi_kernel(MultiTab t, int width, long ld, int L, int gamma,
                                                          const float *__restrict__ r, uint64_t seed, uint64_t draw,
                                                          sd_multi_result *__restrict__ out) {
    __shared__ float sp[256], sq[256];
    __shared__ int sj[256];
    const int tid = threadIdx.x, w = tid >> 4, i = tid & 15;
    float p = 0.f, q = 0.f;
    int j = -1;
    if (w < width && i < gamma) {
        j = t.seq[w][L + i];
        p = t.p_hist[w][(size_t)(L + i - 1) * ld + j];
        q = t.q_hist[w][(size_t)(L + i - 1) * ld + j];
    }
    sp[tid] = p; sq[tid] = q; sj[tid] = j;
    out->p_at[tid] = p;
    out->q_at[tid] = q;
    __syncthreads();
    if (tid != 0) return;
    int k = 0, max_l = 0, choice = 0, all = 0;
    for (int ww = 0; ww < width; ++ww) {
        int cur_l = 0, cur_all = 1;
        for (int ii = 0; ii < gamma; ++ii) {
            const float ri = r ? r[k] : philox_uniform(seed, draw + (uint64_t)k);
            ++k;
            const float ratio = __fdiv_rn(sp[ww * 16 + ii], sq[ww * 16 + ii]);
            if (ratio >= 1.0f || ri < ratio) ++cur_l;             // r < min(1, p/q); NaN compares false -> reject
            else { cur_all = 0; break; }
        }
        if (cur_l > max_l) {
            max_l = cur_l;
            choice = ww;
            if (cur_all) { all = 1; break; }
        }
    }
    out->choice = choice;
    out->n_uniform = k;
    out->width = width;
    out->gamma = gamma;
    sd_accept_result *c = &out->chosen;
    c->n_accepted = max_l;
    c->n = L + max_l - 1;
    c->next_token = -1;
    c->flags = all ? 4 : 0;
    for (int ii = 0; ii < 16; ++ii) {
        c->p_at[ii] = sp[choice * 16 + ii];
        c->q_at[ii] = sq[choice * 16 + ii];
        c->drafted[ii] = sj[choice * 16 + ii];
    }
}

__global__ __launch_bounds__(NT) void resample_batch_kernel(AcceptTab t, long ld, int V, int gamma, int dt) {
    const int b = blockIdx.x;
    resample_body(t.p_hist[b], t.q_hist[b], ld, V, t.seq[b], gamma, t.noise[b], t.seed[b], t.draw_res[b], t.res[b],
                  (int32_t *)nullptr, t.err_flags[b], t.n_err[b], dt);
}

// The device RNG made observable (tests replay it into the CPU oracle): out[i] = the Exp(1) variate element i of
// draw (seed, draw) / the uniform of draw (seed, draw + i), exactly what the sampling kernels consume.
__global__ void philox_exp_kernel(uint64_t seed, uint64_t draw, int V, float *__restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < V) out[i] = philox_exp(seed, draw, i);
}
__global__ void philox_uniform_kernel(uint64_t seed, uint64_t draw, int n, float *__restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = philox_uniform(seed, draw + (uint64_t)i);
}

// ---------------------------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------------------------
extern "C" int sd_philox_exp(uint64_t seed, uint64_t draw_index, int V, float *out, void *stream) {
    SD_REQUIRE(out && V > 0, "sd_philox_exp: bad arguments");
    hipLaunchKernelGGL(philox_exp_kernel, dim3((V + 255) / 256), dim3(256), 0, (hipStream_t)stream, seed, draw_index, V, out);
    SD_LAUNCH_CHECK();
    return SD_OK;
}

extern "C" int sd_philox_uniform(uint64_t seed, uint64_t draw_index, int n, float *out, void *stream) {
    SD_REQUIRE(out && n > 0, "sd_philox_uniform: bad arguments");
    hipLaunchKernelGGL(philox_uniform_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, seed, draw_index, n, out);
    SD_LAUNCH_CHECK();
    return SD_OK;
}

int g_norm_tile_threads = 0;                 // (tools/norm_stamps.cpp sweeps it; 0 = SD_NORM_TILE_THREADS or 256)
static int launch_norm(const float *logits, int rows, int V, long ld_in, float temperature, int top_k, float top_p,
                       int bf16_round_logits, float *probs_out, long ld_out, int *err_flag, bool do_sample,
                       const float *noise, uint64_t seed, uint64_t draw, int *tok_out, int *samp_err, void *workspace,
                       void *stream, const NormTab *tabp = nullptr, int filter_only = 0,
                       const float *tile_max = nullptr, CandList *cl_out = nullptr) {
    NormTab tab = {};
    const int use_tab = tabp != nullptr;
    if (tabp) tab = *tabp;
    const int staged = V <= LDS_ROW_LIMIT;
    const size_t base = (sizeof(NormShared) + 15) & ~size_t(15);
    const size_t lds = base + (staged ? (size_t)V * sizeof(float) : 0);
    static bool attr_set = false;
    if (!attr_set) {
        SD_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(norm_probs_kernel<false>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        SD_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(norm_probs_kernel<true>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set = true;
    }
    // the head already left tile maxima and a cleared output row (EPI_HEAD): no candidate pass over V at all
    if (tile_max && !(top_k >= 1 && top_k <= 64 && temperature > 0.0f && (V & 15) == 0 && V >= 4096 && V <= 65536 &&
                      (ld_in & 3) == 0 && (reinterpret_cast<uintptr_t>(logits) & 15) == 0 && !filter_only))
        tile_max = nullptr;
    // two-kernel fast path: the row is cut over NB_SPLIT workgroups first (see norm_cand_kernel)
    CandRow *ws = nullptr;
    if (!tile_max && workspace && top_k >= 1 && top_k <= 64 && V >= 4096 && (V & 3) == 0 && (ld_in & 3) == 0 && (ld_out & 3) == 0 &&
        ((V >> 2) + NB_SPLIT - 1) / NB_SPLIT <= 256 * CAND_MAXIT && (reinterpret_cast<uintptr_t>(logits) & 15) == 0 &&
        (use_tab || (reinterpret_cast<uintptr_t>(probs_out) & 15) == 0)) {
        ws = static_cast<CandRow *>(workspace);
        hipLaunchKernelGGL(norm_cand_kernel, dim3(NB_SPLIT, rows), dim3(256), 0, (hipStream_t)stream, logits, ld_in, V,
                           temperature, top_k, bf16_round_logits, probs_out, ld_out, ws, tab, use_tab);
        SD_LAUNCH_CHECK();
    }
    // with tile maxima the work is a few dozen candidates: 4 waves (cheap barriers), no row staging
    static const int tile_threads_env = getenv("SD_NORM_TILE_THREADS") ? atoi(getenv("SD_NORM_TILE_THREADS")) : 256;
    const int nthr = tile_max ? (g_norm_tile_threads > 0 ? g_norm_tile_threads : tile_threads_env) : NT;
    const int staged_k = tile_max ? 0 : staged;
    const size_t lds_k = tile_max ? base : lds;
    if (do_sample)
        hipLaunchKernelGGL(norm_probs_kernel<true>, dim3(rows), dim3(nthr), lds_k, (hipStream_t)stream, logits, ld_in, V,
                           temperature, top_k, top_p, bf16_round_logits, staged_k, probs_out, ld_out, err_flag, noise,
                           seed, draw, tok_out, samp_err, (const CandRow *)ws, tab, use_tab, 0, tile_max, cl_out);
    else
        hipLaunchKernelGGL(norm_probs_kernel<false>, dim3(rows), dim3(nthr), lds_k, (hipStream_t)stream, logits, ld_in, V,
                           temperature, top_k, top_p, bf16_round_logits, staged_k, probs_out, ld_out, err_flag,
                           (const float *)nullptr, (uint64_t)0, (uint64_t)0, (int *)nullptr, (int *)nullptr,
                           (const CandRow *)ws, tab, use_tab, filter_only, tile_max, cl_out);
    SD_LAUNCH_CHECK();
    return SD_OK;
}

// internal (engine.hip, sd_spec_iteration): norm_logits of `rows` logit rows whose head left tile maxima (or NULL) and
// cleared probs_out; with tok_out != NULL also the sample that follows a draft step (rows == 1)
int sd_norm_rows_with_tiles(const float *logits, int rows, int V, long ld_in, float temperature, int top_k, float top_p,
                            int bf16_round_logits, float *probs_out, long ld_out, int *err_flag, uint64_t seed,
                            uint64_t draw, int *tok_out, int *samp_err, void *workspace, const float *tile_max,
                            void *stream, void *cand_lists) {
    return launch_norm(logits, rows, V, ld_in, temperature, top_k, top_p, bf16_round_logits, probs_out, ld_out, err_flag,
                       tok_out != nullptr, nullptr, seed, draw, tok_out, samp_err, workspace, stream, nullptr, 0, tile_max,
                       static_cast<CandList *>(cand_lists));
}
extern "C" size_t sd_cand_list_bytes(int rows) { return (size_t)(rows > 0 ? rows : 0) * sizeof(CandList); }
extern "C" int sd_norm_probs_lists(const float *logits, int rows, int V, long ld_in, float temperature, int top_k,
                                   float top_p, int bf16_round_logits, float *probs_out, long ld_out, int *err_flag,
                                   void *workspace, void *cand_lists, void *stream) {
    SD_REQUIRE(logits && probs_out && cand_lists && rows >= 0 && V > 0, "sd_norm_probs_lists: bad arguments");
    SD_REQUIRE(temperature != 0.0f, "sd_norm_probs_lists: temperature must be non-zero");
    if (rows == 0) return SD_OK;
    return launch_norm(logits, rows, V, ld_in, temperature, top_k, top_p, bf16_round_logits, probs_out, ld_out, err_flag,
                       false, nullptr, 0, 0, nullptr, nullptr, workspace, stream, nullptr, 0, nullptr,
                       static_cast<CandList *>(cand_lists));
}

// top_k_top_p_filter on its own (utils.py:152-179): out = logit where kept, -inf where dropped (out != logits).
extern "C" int sd_topk_topp_filter(const float *logits, int rows, int V, long ld_in, int top_k, float top_p,
                                   int dtype_mode, float *out, long ld_out, void *stream) {
    SD_REQUIRE(logits && out && rows >= 0 && V > 0, "sd_topk_topp_filter: bad arguments");
    SD_REQUIRE(dtype_mode == 0 || dtype_mode == SD_NORM_DT_BF16 || dtype_mode == SD_NORM_DT_F16,
               "sd_topk_topp_filter: dtype_mode %d", dtype_mode);
    if (rows == 0) return SD_OK;
    return launch_norm(logits, rows, V, ld_in, 1.0f, top_k, top_p, dtype_mode, out, ld_out, nullptr, false, nullptr, 0, 0, nullptr,
                       nullptr, nullptr, stream, nullptr, 1);
}

// (+ one CandList per row behind the CandRows: the native iteration keeps the target rows' candidate lists there)
extern "C" size_t sd_norm_workspace_bytes(int rows) { return (size_t)(rows > 0 ? rows : 0) * (sizeof(CandRow) + sizeof(CandList)); }
size_t sd_norm_candrow_bytes(int rows) { return (size_t)(rows > 0 ? rows : 0) * sizeof(CandRow); }

extern "C" int sd_norm_probs(const float *logits, int rows, int V, long ld_in, float temperature, int top_k,
                             float top_p, int bf16_round_logits, float *probs_out, long ld_out, int *err_flag,
                             void *workspace, void *stream) {
    SD_REQUIRE(logits && probs_out && rows >= 0 && V > 0, "sd_norm_probs: bad arguments");
    SD_REQUIRE(temperature != 0.0f, "sd_norm_probs: temperature must be non-zero");
    if (rows == 0) return SD_OK;
    return launch_norm(logits, rows, V, ld_in, temperature, top_k, top_p, bf16_round_logits, probs_out, ld_out,
                       err_flag, false, nullptr, 0, 0, nullptr, nullptr, workspace, stream);
}

extern "C" int sd_norm_sample(const float *logits, int V, float temperature, int top_k, float top_p,
                              int bf16_round_logits, float *probs_out, int *err_flag, const float *exp_noise,
                              uint64_t philox_seed, uint64_t draw_index, int *tok_out, int *sample_err,
                              void *workspace, void *stream) {
    SD_REQUIRE(logits && probs_out && tok_out && V > 0, "sd_norm_sample: bad arguments");
    SD_REQUIRE(temperature != 0.0f, "sd_norm_sample: temperature must be non-zero");
    return launch_norm(logits, 1, V, V, temperature, top_k, top_p, bf16_round_logits, probs_out, V, err_flag, true,
                       exp_noise, philox_seed, draw_index, tok_out, sample_err, workspace, stream);
}

// Batched rows: row r of `logits` (stride ld_in) is normalised into rows[r].probs_out; with `sample` each row also
// draws its token (rows[r].tok_out) from its own noise / Philox stream.  Launches of at most SD_NORM_BATCH rows.
// tile_max (or NULL): the head left the rows' tile maxima [n_rows][V / 16] and cleared the output rows (EPI_HEAD);
// cand_lists (or NULL): one CandList per row comes back (the native lock-step loop's residual / bonus sample reads them).
int sd_norm_batch_tiles(const float *logits, int n_rows, int V, long ld_in, float temperature, int top_k, float top_p,
                        int bf16_round_logits, const sd_norm_row *rows, int sample, void *workspace, const float *tile_max,
                        void *cand_lists, void *stream) {
    SD_REQUIRE(logits && rows && n_rows >= 0 && V > 0, "sd_norm_batch: bad arguments");
    SD_REQUIRE(temperature != 0.0f, "sd_norm_batch: temperature must be non-zero");
    for (int r0 = 0; r0 < n_rows; r0 += SD_NORM_BATCH) {
        const int n = n_rows - r0 < SD_NORM_BATCH ? n_rows - r0 : SD_NORM_BATCH;
        NormTab tab = {};
        bool aligned = true;
        for (int i = 0; i < n; ++i) {
            const sd_norm_row &d = rows[r0 + i];
            SD_REQUIRE(d.probs_out && (!sample || d.tok_out), "sd_norm_batch: row %d: null output", r0 + i);
            tab.out[i] = d.probs_out; tab.err[i] = d.err; tab.tok_out[i] = d.tok_out; tab.samp_err[i] = d.sample_err;
            tab.noise[i] = d.exp_noise; tab.seed[i] = d.philox_seed; tab.draw[i] = d.draw_index;
            aligned = aligned && (reinterpret_cast<uintptr_t>(d.probs_out) & 15) == 0;
        }
        char *ws = workspace ? static_cast<char *>(workspace) + (size_t)r0 * sizeof(CandRow) : nullptr;
        const int rc = launch_norm(logits + (size_t)r0 * ld_in, n, V, ld_in, temperature, top_k, top_p, bf16_round_logits,
                                   tab.out[0], 4, nullptr, sample != 0, nullptr, 0, 0, nullptr, nullptr,
                                   aligned ? ws : nullptr, stream, &tab, 0,
                                   tile_max ? tile_max + (size_t)r0 * (size_t)(V >> 4) : nullptr,
                                   cand_lists ? static_cast<CandList *>(cand_lists) + r0 : nullptr);
        if (rc != SD_OK) return rc;
    }
    return SD_OK;
}
extern "C" int sd_norm_batch(const float *logits, int n_rows, int V, long ld_in, float temperature, int top_k,
                             float top_p, int bf16_round_logits, const sd_norm_row *rows, int sample, void *workspace,
                             void *stream) {
    return sd_norm_batch_tiles(logits, n_rows, V, ld_in, temperature, top_k, top_p, bf16_round_logits, rows, sample, workspace,
                               nullptr, nullptr, stream);
}

static inline int mode_dt(int dtype_mode) { return (dtype_mode >> 4) & 3; }

extern "C" int sd_sample(const float *probs, int V, const float *exp_noise, uint64_t philox_seed,
                         uint64_t draw_index, int *tok_out, int *err_flag, int dtype_mode, void *stream) {
    SD_REQUIRE(probs && tok_out && V > 0, "sd_sample: bad arguments");
    hipLaunchKernelGGL(sample_kernel, dim3(1), dim3(NT), 0, (hipStream_t)stream, probs, V, exp_noise, philox_seed,
                       draw_index, tok_out, err_flag, mode_dt(dtype_mode));
    SD_LAUNCH_CHECK();
    return SD_OK;
}

extern "C" int sd_max_fn(const float *p, const float *q, int V, float *out, int dtype_mode, void *stream) {
    SD_REQUIRE(p && out && V > 0, "sd_max_fn: bad arguments");
    hipLaunchKernelGGL(max_fn_kernel, dim3(1), dim3(NT), 0, (hipStream_t)stream, p, q, V, out, mode_dt(dtype_mode));
    SD_LAUNCH_CHECK();
    return SD_OK;
}

extern "C" int sd_accept_scan(const float *p_hist, const float *q_hist, long ld, const int32_t *seq, int L,
                              int gamma, const float *r, uint64_t philox_seed, uint64_t draw_index,
                              sd_accept_result *out, void *stream) {
    SD_REQUIRE(p_hist && q_hist && seq && out, "sd_accept_scan: bad arguments");
    SD_REQUIRE(gamma >= 1 && gamma <= 16 && L >= 1, "sd_accept_scan: gamma must be in 1..16 and L >= 1");
    hipLaunchKernelGGL(accept_scan_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, p_hist, q_hist, ld, seq, L,
                       gamma, r, philox_seed, draw_index, out);
    SD_LAUNCH_CHECK();
    return SD_OK;
}

extern "C" int sd_resample(const float *p_hist, const float *q_hist, long ld, int V, int32_t *seq, int L, int gamma,
                           const float *exp_noise, uint64_t philox_seed, uint64_t draw_index,
                           sd_accept_result *res, int32_t *seq_len, int dtype_mode, void *stream) {
    SD_REQUIRE(p_hist && q_hist && seq && res && V > 0, "sd_resample: bad arguments");
    (void)L;
    hipLaunchKernelGGL(resample_kernel, dim3(1), dim3(NT), 0, (hipStream_t)stream, p_hist, q_hist, ld, V, seq, gamma,
                       exp_noise, philox_seed, draw_index, res, seq_len, (const int *)nullptr, 0, mode_dt(dtype_mode));
    SD_LAUNCH_CHECK();
    return SD_OK;
}

// internal: the same with the iteration's error words folded into res->flags bit3 (used by sd_spec_iteration)
int sd_resample_with_errors(const float *p_hist, const float *q_hist, long ld, int V, int32_t *seq, int gamma,
                            uint64_t philox_seed, uint64_t draw_index, sd_accept_result *res, const int *err_flags,
                            int n_err, int dtype_mode, hipStream_t st) {
    hipLaunchKernelGGL(resample_kernel, dim3(1), dim3(NT), 0, st, p_hist, q_hist, ld, V, seq, gamma,
                       (const float *)nullptr, philox_seed, draw_index, res, (int32_t *)nullptr, err_flags, n_err,
                       mode_dt(dtype_mode));
    SD_LAUNCH_CHECK();
    return SD_OK;
}

// accept scan + residual / bonus sample of one native iteration in ONE launch, the sample working on the candidate list of
// target row n (its <= SD_CL_CAP non-zero probabilities) instead of two passes over V: every weight outside p_n's support is
// zero, so sums, arg-maxima and the < 1e-9 fix-up only ever see list entries.  The result is bit-identical to
// accept_scan_kernel + resample_kernel: the normaliser is summed through the same 1024 per-thread partials (entry i in
// slot i mod 1024, entries of one slot in index order, zeros adding nothing), block_sum and block_argmax are the same
// device functions, and the Philox variates are drawn by token id.  Without a list (pl == NULL, or the row was not
// produced in list mode) it runs resample_body's dense passes.
__device__ __forceinline__ void accept_resample_body(const float *__restrict__ p_hist, const float *__restrict__ q_hist,
                                                     long ld, int V, int32_t *__restrict__ seq, int L, int gamma,
                                                     const float *__restrict__ r, uint64_t seed, uint64_t draw_scan,
                                                     uint64_t draw_res, sd_accept_result *__restrict__ res,
                                                     const int *__restrict__ err_flags, int n_err, int dt,
                                                     const CandList *__restrict__ pl) {
    __shared__ SampleShared S;
    __shared__ float red[16];
    __shared__ float part[NT];
    __shared__ int cidx[SD_CL_CAP];
    __shared__ float cval[SD_CL_CAP];
    const int tid = threadIdx.x;
    if (tid < 64) accept_scan_body(p_hist, q_hist, ld, seq, L, gamma, r, seed, draw_scan, res);
    __syncthreads();
    const int n = res->n, first = res->n_accepted;
    const bool rejected = first < gamma;
    const int nc = pl ? pl[first].n : -1;
    if (nc < 0) {                                                 // no list for this row: the dense passes
        resample_body(p_hist, q_hist, ld, V, seq, gamma, nullptr, seed, draw_res, res, nullptr, err_flags, n_err, dt);
        return;
    }
    const CandList &cl = pl[first];
    const float *q = q_hist + (size_t)n * ld;
    const bool mine = tid < nc;
    const int idx = mine ? cl.idx[tid] : 0;
    const float pv = mine ? cl.prob[tid] : 0.f;
    if (mine) cidx[tid] = idx;
    // sum_i v_i over the row as resample_body's strided loop + block_sum computes it (v = 0 off the list)
    auto row_sum = [&](float v) -> float {
        part[tid] = 0.f;
        if (mine) cval[tid] = v;
        __syncthreads();
        if (mine) {
            const int slot = idx & (NT - 1);
            int same = 0, before = 0;
            for (int j = 0; j < nc; ++j) {
                const int ij = cidx[j];
                if (j != tid && (ij & (NT - 1)) == slot) { ++same; before += (ij < idx); }
            }
            if (same == 0) part[slot] = v;
            else if (before == 0) {                               // lowest index of a slot shared by several entries: add them in index order
                float acc = 0.f;
                int last = -1;
                for (int t = 0; t <= same; ++t) {
                    int best = 0x7fffffff, bj = -1;
                    for (int j = 0; j < nc; ++j) {
                        const int ij = cidx[j];
                        if ((ij & (NT - 1)) == slot && ij > last && ij < best) { best = ij; bj = j; }
                    }
                    acc += cval[bj];
                    last = best;
                }
                part[slot] = acc;
            }
        }
        __syncthreads();
        return block_sum(part[tid], red);
    };
    // sample_core over weights that are zero off the list
    auto sparse_sample = [&](float w, int *status) -> int {
        int bad = mine && (!(w >= 0.0f) || w == INFINITY), pos = mine && w > 0.0f;
        bad = block_sum_i(bad, S.redi);
        pos = block_sum_i(pos, S.redi);
        if (bad) { *status = 1; return 0; }
        if (!pos) { *status = 2; return 0; }
        ArgMax br = {0.f, 0x7fffffff}, bw = {0.f, 0x7fffffff};
        if (mine) {
            if (w > 0.0f) br = {rnd_dt(w / philox_exp(seed, draw_res, idx), dt), idx};
            bw = {w, idx};
            cval[tid] = w;
        }
        br = block_argmax(br, S.reda);
        bw = block_argmax(bw, S.reda);                            // (its barriers also publish cval)
        *status = 0;
        // every ratio zero (underflow): the dense arg-max keeps its first element, token 0
        int tok = (br.i == 0x7fffffff || !(br.v > 0.0f)) ? 0 : br.i;
        float wtok = 0.f;
        for (int j = 0; j < nc; ++j)
            if (cidx[j] == tok) wtok = cval[j];
        if (wtok < 1e-9f) tok = bw.i;                             // utils.py:228-230
        return tok;
    };
    int status = 0, tok = 0, flags = 0;
    if (rejected) {
        const float d = mine ? rnd_dt(pv - q[idx], dt) : 0.f;
        const float u = d > 0.f ? d : 0.f;
        const float denom = max_fn_denom(row_sum(u), dt);         // max_fn, utils.py:236-245
        tok = sparse_sample(mine ? rnd_dt(u / denom, dt) : 0.f, &status);
        if (status != 0) {                                        // residual sample raised -> sample(max_fn(p_n)) (:2009-2010)
            flags |= 1;
            __syncthreads();
            const float u2 = pv > 0.f ? pv : 0.f;
            const float denom2 = max_fn_denom(row_sum(u2), dt);
            tok = sparse_sample(mine ? rnd_dt(u2 / denom2, dt) : 0.f, &status);
        }
    } else {
        tok = sparse_sample(pv, &status);                         // bonus token from p_last (:2019)
    }
    if (tid == 0) {
        if (status != 0) flags |= 2;
        for (int i = 0; i < n_err; ++i)                           // norm / sample error words of this iteration
            if (err_flags[i]) flags |= 8;
        res->flags |= flags;
        res->next_token = status == 0 ? tok : -1;
        if (status == 0) seq[n + 1] = tok;
    }
}

__global__ __launch_bounds__(NT) void accept_resample_kernel(const float *__restrict__ p_hist, const float *__restrict__ q_hist,
                                                            long ld, int V, int32_t *__restrict__ seq, int L, int gamma,
                                                            const float *__restrict__ r, uint64_t seed, uint64_t draw_scan,
                                                            uint64_t draw_res, sd_accept_result *__restrict__ res,
                                                            const int *__restrict__ err_flags, int n_err, int dt,
                                                            const CandList *__restrict__ pl) {
    accept_resample_body(p_hist, q_hist, ld, V, seq, L, gamma, r, seed, draw_scan, draw_res, res, err_flags, n_err, dt, pl);
}

// The same for the streams of a lock-step iteration, one workgroup per stream (device Philox; lists[i] = the candidate
// lists of stream i's gamma + 1 target rows, or NULL: that stream takes the dense passes).
struct ListTab { const CandList *pl[SD_ACCEPT_BATCH]; };
__global__ __launch_bounds__(NT) void accept_resample_batch_kernel(AcceptTab t, ListTab lt, long ld, int V, int gamma, int dt) {
    const int i = blockIdx.x;
    accept_resample_body(t.p_hist[i], t.q_hist[i], ld, V, t.seq[i], t.L[i], gamma, t.r[i], t.seed[i], t.draw_scan[i],
                         t.draw_res[i], t.res[i], t.err_flags[i], t.n_err[i], dt, lt.pl[i]);
}

extern "C" int sd_accept_resample(const float *p_hist, const float *q_hist, long ld, int V, int32_t *seq, int L, int gamma,
                                  const float *r, uint64_t philox_seed, uint64_t draw_scan, uint64_t draw_resample,
                                  sd_accept_result *res, const int *err_flags, int n_err, int dtype_mode,
                                  const void *target_lists, void *stream) {
    SD_REQUIRE(p_hist && q_hist && seq && res && gamma >= 1 && gamma <= 16 && L >= 1 && V > 0 && (n_err == 0 || err_flags),
               "sd_accept_resample: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(accept_resample_kernel, dim3(1), dim3(NT), 0, st, p_hist, q_hist, ld, V, seq, L, gamma, r, philox_seed,
                       draw_scan, draw_resample, res, err_flags, n_err, mode_dt(dtype_mode),
                       static_cast<const CandList *>(target_lists));
    SD_LAUNCH_CHECK();
    return SD_OK;
}

extern "C" int sd_accept_multi(const sd_multi_item *items, int width, long ld, int L, int gamma, const float *r,
                               uint64_t philox_seed, uint64_t draw_index, sd_multi_result *out, void *stream) {
    SD_REQUIRE(items && out && width >= 1 && width <= 16, "sd_accept_multi: 1..16 replicas");
    SD_REQUIRE(gamma >= 1 && gamma <= 16 && L >= 1, "sd_accept_multi: bad gamma / L");
    MultiTab t = {};
    for (int w = 0; w < width; ++w) {
        SD_REQUIRE(items[w].p_hist && items[w].q_hist && items[w].seq, "sd_accept_multi: replica %d: bad arguments", w);
        t.p_hist[w] = items[w].p_hist; t.q_hist[w] = items[w].q_hist; t.seq[w] = items[w].seq;
    }
    hipLaunchKernelGGL(accept_multi_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, t, width, ld, L, gamma, r,
                       philox_seed, draw_index, out);
    SD_LAUNCH_CHECK();
    return SD_OK;
}

extern "C" int sd_multi_resample(const float *p_hist, const float *q_hist, long ld, int V, int32_t *seq, int gamma,
                                 const float *exp_noise, uint64_t philox_seed, uint64_t draw_index,
                                 sd_accept_result *res, int dtype_mode, void *stream) {
    SD_REQUIRE(p_hist && q_hist && seq && res && V > 0, "sd_multi_resample: bad arguments");
    hipLaunchKernelGGL(multi_resample_kernel, dim3(1), dim3(NT), 0, (hipStream_t)stream, p_hist, q_hist, ld, V, seq,
                       gamma, exp_noise, philox_seed, draw_index, res, mode_dt(dtype_mode));
    SD_LAUNCH_CHECK();
    return SD_OK;
}

// The lock-step loop's accept scan + residual / bonus sample in ONE launch on the target rows' candidate lists
// (accept_resample_body per stream; device Philox only - an item with exp_noise is refused).
int sd_accept_resample_batch(const sd_accept_item *items, int n_items, long ld, int V, int gamma, int dtype_mode,
                             const void *const *lists, void *stream) {
    SD_REQUIRE(items && lists && n_items >= 1 && n_items <= SD_ACCEPT_BATCH, "sd_accept_resample_batch: 1..%d items", SD_ACCEPT_BATCH);
    SD_REQUIRE(gamma >= 1 && gamma <= 16 && V > 0, "sd_accept_resample_batch: bad gamma / V");
    AcceptTab t = {};
    ListTab lt = {};
    for (int i = 0; i < n_items; ++i) {
        const sd_accept_item &it = items[i];
        SD_REQUIRE(it.p_hist && it.q_hist && it.seq && it.res && it.L >= 1 && !it.exp_noise,
                   "sd_accept_resample_batch: item %d: bad arguments", i);
        t.p_hist[i] = it.p_hist; t.q_hist[i] = it.q_hist; t.seq[i] = it.seq; t.r[i] = it.r; t.noise[i] = nullptr;
        t.res[i] = it.res; t.err_flags[i] = it.err_flags; t.n_err[i] = it.err_flags ? it.n_err : 0;
        t.seed[i] = it.philox_seed; t.draw_scan[i] = it.draw_scan; t.draw_res[i] = it.draw_resample; t.L[i] = it.L;
        lt.pl[i] = static_cast<const CandList *>(lists[i]);
    }
    hipLaunchKernelGGL(accept_resample_batch_kernel, dim3(n_items), dim3(NT), 0, (hipStream_t)stream, t, lt, ld, V, gamma,
                       mode_dt(dtype_mode));
    SD_LAUNCH_CHECK();
    return SD_OK;
}

// Accept scan + residual / bonus sample for up to 16 independent streams in two launches (stream-batched decode).
extern "C" int sd_accept_batch(const sd_accept_item *items, int n_items, long ld, int V, int gamma, int dtype_mode,
                               void *stream) {
    SD_REQUIRE(items && n_items >= 1 && n_items <= SD_ACCEPT_BATCH, "sd_accept_batch: 1..%d items", SD_ACCEPT_BATCH);
    SD_REQUIRE(gamma >= 1 && gamma <= 16 && V > 0, "sd_accept_batch: bad gamma / V");
    AcceptTab t = {};
    for (int i = 0; i < n_items; ++i) {
        const sd_accept_item &it = items[i];
        SD_REQUIRE(it.p_hist && it.q_hist && it.seq && it.res && it.L >= 1, "sd_accept_batch: item %d: bad arguments", i);
        t.p_hist[i] = it.p_hist; t.q_hist[i] = it.q_hist; t.seq[i] = it.seq; t.r[i] = it.r; t.noise[i] = it.exp_noise;
        t.res[i] = it.res; t.err_flags[i] = it.err_flags; t.n_err[i] = it.err_flags ? it.n_err : 0;
        t.seed[i] = it.philox_seed; t.draw_scan[i] = it.draw_scan; t.draw_res[i] = it.draw_resample; t.L[i] = it.L;
    }
    hipLaunchKernelGGL(accept_scan_batch_kernel, dim3(n_items), dim3(64), 0, (hipStream_t)stream, t, ld, gamma);
    SD_LAUNCH_CHECK();
    hipLaunchKernelGGL(resample_batch_kernel, dim3(n_items), dim3(NT), 0, (hipStream_t)stream, t, ld, V, gamma,
                       mode_dt(dtype_mode));
    SD_LAUNCH_CHECK();
    return SD_OK;
}
