// Sampling primitives of the speculative-sampling path as HIP kernels for gfx950:
//   norm_probs   <- reference sampling/utils.py:182-210 (norm_logits) + :152-179 (top_k_top_p_filter)
//   sample       <- utils.py:213-233
//   max_fn       <- utils.py:236-245
//   accept scan  <- sampling/speculative_sampling.py:1964-1991
//   resample     <- sampling/speculative_sampling.py:2005-2023
// One workgroup of 1024 threads (16 waves) owns one vocabulary row: the row is staged once in
// LDS (V*4 B <= 128 KiB for Llama's V = 32000) and every later pass reads LDS, not HBM.
#include "common.h"

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

__device__ __forceinline__ int block_sum_i(int v, int *sh) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    __syncthreads();
    if (lane == 0) sh[w] = v;
    __syncthreads();
    int r = 0;
#pragma unroll
    for (int i = 0; i < NT / 64; ++i) r += sh[i];
    return r;
}

__device__ __forceinline__ double block_sum_d(double v, double *sh) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    v = wave_sum_d(v);
    __syncthreads();
    if (lane == 0) sh[w] = v;
    __syncthreads();
    double r = 0.0;
#pragma unroll
    for (int i = 0; i < NT / 64; ++i) r += sh[i];
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
#pragma unroll
    for (int i = 1; i < NT / 64; ++i) r = am_better(r, sh[i]);
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
__device__ __forceinline__ float u01_open(uint32_t x) { return ((float)(x >> 8) + 0.5f) * (1.0f / 16777216.0f); }  // (0,1)
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

// ---------------------------------------------------------------------------------------------
// norm_probs
// ---------------------------------------------------------------------------------------------
struct NormShared {
    float redf[16];
    int redi[16];
    double redd[16];
    int n_cand;
    int cut_idx;
    uint32_t cut_key;
    int kept;
    uint32_t ckey[MAX_CAND];
    int cidx[MAX_CAND];
    uint32_t skey[MAX_CAND];
    int sidx[MAX_CAND];
};

__global__ __launch_bounds__(NT) void norm_probs_kernel(const float *__restrict__ logits, long ld_in, int V,
                                                       float temperature, int top_k, float top_p, int bf16_round,
                                                       int staged, float *__restrict__ out, long ld_out,
                                                       int *__restrict__ err) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    NormShared &S = *reinterpret_cast<NormShared *>(smem);
    float *zs = reinterpret_cast<float *>(smem + ((sizeof(NormShared) + 15) & ~size_t(15)));
    const int row = blockIdx.x, tid = threadIdx.x;
    const float *x = logits + (size_t)row * ld_in;
    float *o = out + (size_t)row * ld_out;

    auto load_z = [&](int i) -> float {
        float v = x[i];
        if (bf16_round) v = (float)(bf16_t)v;
        return v / temperature;                                   // utils.py:197
    };
    auto Z = [&](int i) -> float { return staged ? zs[i] : load_z(i); };

    // pass 0: stage, row max, NaN detection
    float m = -INFINITY;
    int bad = 0;
    for (int i = tid; i < V; i += NT) {
        const float z = load_z(i);
        if (staged) zs[i] = z;
        bad |= (z != z);
        m = fmaxf(m, z);
    }
    m = block_max(m, S.redf);
    bad = block_sum_i(bad, S.redi);
    if (bad || m == INFINITY || m == -INFINITY) {                 // exp(log_softmax) would hold NaN (utils.py:203)
        for (int i = tid; i < V; i += NT) o[i] = __uint_as_float(0x7fc00000u);
        if (tid == 0 && err) err[row] = 1;
        return;
    }

    // top-k (utils.py:166-169): the k-th largest value by bitwise bisection on the ordered key;
    // everything strictly below it is dropped, ties at the k-th value stay.
    uint32_t kth = 0u;                                            // key >= 0 keeps everything
    int n_surv = V;
    if (top_k > 0) {
        const int k = min(top_k, V);
        uint32_t prefix = 0u;
        for (int bit = 31; bit >= 0; --bit) {
            const uint32_t cand = prefix | (1u << bit);
            int c = 0;
            for (int i = tid; i < V; i += NT) c += (fkey(Z(i)) >= cand);
            c = block_sum_i(c, S.redi);
            if (c >= k) prefix = cand;
        }
        kth = prefix;
        int c = 0;
        for (int i = tid; i < V; i += NT) c += (fkey(Z(i)) >= kth);
        n_surv = block_sum_i(c, S.redi);
    }

    // top-p (utils.py:170-178).  keep(i) <=> key_i > cut_key || (key_i == cut_key && i <= cut_idx)
    uint32_t cut_key = kth;
    int cut_idx = 0x7fffffff;
    if (top_p > 0.0f) {
        // softmax denominator over the survivors (the -inf entries add exp(-inf) = 0)
        float part = 0.f;
        for (int i = tid; i < V; i += NT) {
            const float z = Z(i);
            if (fkey(z) >= kth) part += expf(z - m);
        }
        const float denom = block_sum(part, S.redf);
        const uint32_t neg_inf_key = fkey(-INFINITY);
        // entries that are already -inf carry no mass and sort last; leave them out of the candidates
        int c = 0;
        for (int i = tid; i < V; i += NT) {
            const uint32_t k = fkey(Z(i));
            c += (k >= kth && k > neg_inf_key);
        }
        const int n_fin = block_sum_i(c, S.redi);
        if (n_fin <= MAX_CAND) {
            if (tid == 0) S.n_cand = 0;
            __syncthreads();
            for (int i = tid; i < V; i += NT) {
                const uint32_t k = fkey(Z(i));
                if (k >= kth && k > neg_inf_key) {
                    const int s = atomicAdd(&S.n_cand, 1);
                    S.ckey[s] = k;
                    S.cidx[s] = i;
                }
            }
            __syncthreads();
            const int n = S.n_cand;
            if (tid < n) {                                        // stable descending order by rank counting
                const uint32_t k = S.ckey[tid];
                const int id = S.cidx[tid];
                int rank = 0;
                for (int j = 0; j < n; ++j) {
                    const uint32_t kj = S.ckey[j];
                    rank += (kj > k) || (kj == k && S.cidx[j] < id);
                }
                S.skey[rank] = k;
                S.sidx[rank] = id;
            }
            __syncthreads();
            if (tid == 0) {
                // torch.cumsum accumulates float32 inputs in double and rounds each prefix to float32
                double cum = 0.0;
                int kept = 0;
                for (int i = 0; i < n; ++i) {
                    if (i > 0 && (float)cum > top_p) break;       // shifted filter: the crossing token stays
                    const float p = expf(Z(S.sidx[i]) - m) / denom;
                    cum += (double)p;
                    kept = i + 1;
                }
                S.kept = kept;
                S.cut_key = S.skey[kept - 1];
                S.cut_idx = S.sidx[kept - 1];
            }
            __syncthreads();
            cut_key = S.cut_key;
            cut_idx = S.cut_idx;
        } else {
            // General path (no / very wide top-k): smallest existing value v* whose first tie member is
            // kept, i.e. float(mass strictly above v*) <= top_p, by bisection on the key; then how many
            // of its tie members fit.  Mass is accumulated in double like torch.cumsum does.
            uint32_t lo = (kth > neg_inf_key + 1u) ? kth : neg_inf_key + 1u, hi = fkey(m);
            while (lo < hi) {
                const uint32_t mid = lo + ((hi - lo) >> 1);
                double g = 0.0;
                for (int i = tid; i < V; i += NT) {
                    const float z = Z(i);
                    if (fkey(z) > mid) g += (double)(expf(z - m) / denom);
                }
                g = block_sum_d(g, S.redd);
                if (!((float)g > top_p)) hi = mid; else lo = mid + 1u;
            }
            // v* = smallest existing key >= lo
            uint32_t best = 0xffffffffu;
            for (int i = tid; i < V; i += NT) {
                const uint32_t k = fkey(Z(i));
                if (k >= lo) best = min(best, k);
            }
            {
                int b = (int)(best ^ 0x80000000u);                // order-preserving map to signed for the int reduce
                __syncthreads();
                // block min through LDS
                const int lane = tid & 63, w = tid >> 6;
#pragma unroll
                for (int of = 32; of > 0; of >>= 1) b = min(b, __shfl_xor(b, of, 64));
                if (lane == 0) S.redi[w] = b;
                __syncthreads();
                int r = S.redi[0];
                for (int i2 = 1; i2 < NT / 64; ++i2) r = min(r, S.redi[i2]);
                best = (uint32_t)r ^ 0x80000000u;
                __syncthreads();
            }
            const uint32_t vstar = best;
            double g = 0.0;
            int e = 0;
            float pv = 0.f;
            for (int i = tid; i < V; i += NT) {
                const float z = Z(i);
                const uint32_t k = fkey(z);
                if (k > vstar) g += (double)(expf(z - m) / denom);
                if (k == vstar) { e += 1; pv = expf(z - m) / denom; }
            }
            g = block_sum_d(g, S.redd);
            e = block_sum_i(e, S.redi);
            pv = block_max(pv, S.redf);
            int mstar = 1;                                        // first tie member is kept by construction
            {
                double cum = g + (double)pv;
                while (mstar < e && !((float)cum > top_p)) { cum += (double)pv; ++mstar; }
            }
            cut_key = vstar;
            cut_idx = 0x7fffffff;
            if (mstar < e) {                                      // ties straddle the cut: keep the mstar lowest indices
                int lo_i = 0, hi_i = V - 1;
                while (lo_i < hi_i) {
                    const int mid = lo_i + ((hi_i - lo_i) >> 1);
                    int c2 = 0;
                    for (int i = tid; i < V; i += NT) c2 += (i <= mid && fkey(Z(i)) == vstar);
                    c2 = block_sum_i(c2, S.redi);
                    if (c2 >= mstar) hi_i = mid; else lo_i = mid + 1;
                }
                cut_idx = lo_i;
            }
        }
    }

    // probs = exp(log_softmax(filtered))  (utils.py:199)
    float part = 0.f;
    for (int i = tid; i < V; i += NT) {
        const float z = Z(i);
        const uint32_t k = fkey(z);
        if (k > cut_key || (k == cut_key && i <= cut_idx)) part += expf(z - m);
    }
    const float lse = logf(block_sum(part, S.redf));
    for (int i = tid; i < V; i += NT) {
        const float z = Z(i);
        const uint32_t k = fkey(z);
        const bool keep = k > cut_key || (k == cut_key && i <= cut_idx);
        o[i] = keep ? expf((z - m) - lse) : 0.0f;
    }
    if (tid == 0 && err) err[row] = 0;
    (void)n_surv;
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
__device__ __forceinline__ int sample_core(int V, WF W, EF E, SampleShared &S, int *status) {
    const int tid = threadIdx.x;
    ArgMax best_r = {0.f, 0x7fffffff}, best_w = {0.f, 0x7fffffff};
    int bad = 0, pos = 0;
    for (int i = tid; i < V; i += NT) {
        const float w = W(i);
        bad |= !(w >= 0.0f) || (w == INFINITY);                   // negative, NaN or Inf: multinomial's validity check
        pos |= (w > 0.0f);
        const float r = w / E(i);                                 // IEEE division, as at::div
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
                                                   int *__restrict__ tok_out, int *__restrict__ err) {
    __shared__ SampleShared S;
    int status;
    const int tok = sample_core(
        V, [&](int i) { return probs[i]; },
        [&](int i) { return noise ? noise[i] : philox_exp(seed, draw, i); }, S, &status);
    if (threadIdx.x == 0) {
        if (status == 0) *tok_out = tok;
        if (err) *err = status;
    }
}

__global__ __launch_bounds__(NT) void max_fn_kernel(const float *__restrict__ p, const float *__restrict__ q, int V,
                                                   float *__restrict__ out) {
    __shared__ float red[16];
    float part = 0.f;
    for (int i = threadIdx.x; i < V; i += NT) {
        const float d = q ? p[i] - q[i] : p[i];
        part += d > 0.f ? d : 0.f;
    }
    const float denom = block_sum(part, red) + 1e-6f;
    for (int i = threadIdx.x; i < V; i += NT) {
        const float d = q ? p[i] - q[i] : p[i];
        out[i] = (d > 0.f ? d : 0.f) / denom;
    }
}

// ---------------------------------------------------------------------------------------------
// accept scan + resample
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void accept_scan_kernel(const float *__restrict__ p_hist,
                                                        const float *__restrict__ q_hist, long ld,
                                                        const int32_t *__restrict__ seq, int L, int gamma,
                                                        const float *__restrict__ r, uint64_t seed, uint64_t draw,
                                                        sd_accept_result *__restrict__ out) {
    const int i = threadIdx.x;
    bool reject = false;
    float p = 0.f, q = 1.f;
    if (i < gamma) {
        const int j = seq[L + i];
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
    }
    if (i == 0) {
        out->n_accepted = first;
        out->n = L + first - 1;
        out->flags = (first == gamma) ? 4 : 0;
        out->next_token = -1;
    }
}

__global__ __launch_bounds__(NT) void resample_kernel(const float *__restrict__ p_hist,
                                                     const float *__restrict__ q_hist, long ld, int V,
                                                     int32_t *__restrict__ seq, int gamma,
                                                     const float *__restrict__ noise, uint64_t seed, uint64_t draw,
                                                     sd_accept_result *__restrict__ res,
                                                     int32_t *__restrict__ seq_len) {
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
            const float d = p[i] - q[i];
            part += d > 0.f ? d : 0.f;
        }
        const float denom = block_sum(part, red) + 1e-6f;         // max_fn, utils.py:236-245
        tok = sample_core(
            V, [&](int i) { const float d = p[i] - q[i]; return (d > 0.f ? d : 0.f) / denom; }, E, S, &status);
        if (status != 0) {                                        // residual sample raised -> sample(max_fn(p_n)) (:2009-2010)
            flags |= 1;
            part = 0.f;
            for (int i = threadIdx.x; i < V; i += NT) part += p[i] > 0.f ? p[i] : 0.f;
            const float denom2 = block_sum(part, red) + 1e-6f;
            tok = sample_core(
                V, [&](int i) { const float d = p[i]; return (d > 0.f ? d : 0.f) / denom2; }, E, S, &status);
        }
    } else {
        tok = sample_core(V, [&](int i) { return p[i]; }, E, S, &status);   // bonus token from p_last (:2019)
    }
    if (threadIdx.x == 0) {
        if (status != 0) flags |= 2;
        res->flags |= flags;
        res->next_token = status == 0 ? tok : -1;
        if (status == 0) seq[n + 1] = tok;
        if (seq_len) *seq_len = n + 2;
    }
}

// ---------------------------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------------------------
extern "C" int sd_norm_probs(const float *logits, int rows, int V, long ld_in, float temperature, int top_k,
                             float top_p, int bf16_round_logits, float *probs_out, long ld_out, int *err_flag,
                             void *stream) {
    SD_REQUIRE(logits && probs_out && rows >= 0 && V > 0, "sd_norm_probs: bad arguments");
    SD_REQUIRE(temperature != 0.0f, "sd_norm_probs: temperature must be non-zero");
    if (rows == 0) return SD_OK;
    const int staged = V <= LDS_ROW_LIMIT;
    const size_t base = (sizeof(NormShared) + 15) & ~size_t(15);
    const size_t lds = base + (staged ? (size_t)V * sizeof(float) : 0);
    static bool attr_set = false;
    if (!attr_set) {
        SD_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(norm_probs_kernel),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set = true;
    }
    hipLaunchKernelGGL(norm_probs_kernel, dim3(rows), dim3(NT), lds, (hipStream_t)stream, logits, ld_in, V,
                       temperature, top_k, top_p, bf16_round_logits, staged, probs_out, ld_out, err_flag);
    SD_LAUNCH_CHECK();
    return SD_OK;
}

extern "C" int sd_sample(const float *probs, int V, const float *exp_noise, uint64_t philox_seed,
                         uint64_t draw_index, int *tok_out, int *err_flag, void *stream) {
    SD_REQUIRE(probs && tok_out && V > 0, "sd_sample: bad arguments");
    hipLaunchKernelGGL(sample_kernel, dim3(1), dim3(NT), 0, (hipStream_t)stream, probs, V, exp_noise, philox_seed,
                       draw_index, tok_out, err_flag);
    SD_LAUNCH_CHECK();
    return SD_OK;
}

extern "C" int sd_max_fn(const float *p, const float *q, int V, float *out, void *stream) {
    SD_REQUIRE(p && out && V > 0, "sd_max_fn: bad arguments");
    hipLaunchKernelGGL(max_fn_kernel, dim3(1), dim3(NT), 0, (hipStream_t)stream, p, q, V, out);
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
                           sd_accept_result *res, int32_t *seq_len, void *stream) {
    SD_REQUIRE(p_hist && q_hist && seq && res && V > 0, "sd_resample: bad arguments");
    (void)L;
    hipLaunchKernelGGL(resample_kernel, dim3(1), dim3(NT), 0, (hipStream_t)stream, p_hist, q_hist, ld, V, seq, gamma,
                       exp_noise, philox_seed, draw_index, res, seq_len);
    SD_LAUNCH_CHECK();
    return SD_OK;
}
