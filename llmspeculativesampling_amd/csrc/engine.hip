// Host side of the decoder engine: model / session handles and the per-forward launch chain.
// Replaces (for one sequence) reference sampling/kvcache_model.py:141-252 -> model forward
// (modeling_llama.py:624-768 / modeling_opt.py:561-759) with a fixed chain of HIP launches over
// a preallocated KV arena; rollback (kvcache_model.py:359-436) is the caller passing a smaller pos0.
#include <math.h>
#include <stdarg.h>
#include <stdlib.h>

#include <chrono>
#include <vector>

#include "model_kernels.h"
#include "small_kernels.h"
#include "rows_kernels.h"
#include "mm_kernels.h"
#include "prefill_attn.h"
#include "fused_kernels.h"
#include "normload_kernels.h"

static thread_local char g_err[512] = "";
void sd_set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
extern "C" const char *sd_last_error(void) { return g_err; }
extern "C" int sd_version(void) { return SD_ABI_VERSION; }

struct sd_model {
    sd_model_config cfg;
    sd_model_weights w;
    std::vector<const void *> wqkv, bqkv, wo, bo, wgu, bfc1, wdown, bfc2, n1w, n1b, n2w, n2b;
};

enum { PC_GEMM = 0, PC_ATTN, PC_NORM, PC_QKV, PC_ACT, PC_EMBED, PC_LOGITS, PC_OTHER };

struct sd_session {
    sd_model *m;
    int max_seq, max_rows;
    char *kv;        // [L][2][Hkv][max_seq][D]
    char *scratch;
    // carved scratch
    void *x, *h, *qbuf, *attn, *act, *ebuf;
    void *x2;           // second residual-stream buffer of the small-model path (the prologue-fused chain ping-pongs)
    float *spart;       // split-K slabs of the small-model path's O / down GEMMs (its head writes s->part meanwhile)
    void *h2;           // second normalised-operand buffer (small-model path)
    unsigned *ao_ctr;      // arrival counter of the fused attention + O projection launches (monotonic, fused_kernels.h)
    unsigned ao_epoch;     // arrivals expected so far (wraps; compared as a signed difference)
    unsigned *wait_status; // sticky device word: bit 0 a fused attention + O wait timed out, bit 1 a k-split finisher's wait did
    long long *ao_stamps;  // [AO_STAMP_WGS][8] per-workgroup records of the last stamped fused launch (SD_AO_STAMPS=1)
    int resync;            // the next forward re-zeroes the arrival counters and epochs first (set after a failed forward)
    int test_skew, skew_now;   // test hook (sd_session_test_skew_wait): arrivals the NEXT forward's fused waits over-expect
    unsigned *fin_ctr;     // [hidden / 16] arrival counters of the k-split GEMM with residual epilogue (monotonic, normload_kernels.h)
    unsigned fin_epoch;    // arrivals per tile expected so far (wraps; compared as a signed difference)
    float *ssq;            // [16][hidden / 16] per-tile sums of squares left by a residual epilogue (normload_kernels.h)
    size_t spart_floats;
    float *tile_max;    // [SD_MAX_ROWS][vocab / 16] maxima of the head's 16-column tiles (EPI_HEAD)
    int kv_fp8;         // the arena holds fp8 e4m3 (sd_session_set_kv_fp8)
    const float *kv_scale;   // device [L][2][Hkv]
    struct sd_tp *tp;   // tensor-parallel group of this shard (NULL: the model is whole)
    float *tp_in, *tp_out;   // [max_rows][hidden] fp32: this rank's partial O / down output, and the all-reduced sum
    float *attn_part;   // [groups*splits <= 64][Hq][TQ][D+2] partial attention sums of the split-key path
    float *part;
    size_t part_floats;
    // native iteration (sd_spec_*): with want_raw_logits a forward whose lm_head ran unsplit leaves the logits in the
    // GEMM's own output slab (no copy launch) and says where they are and whether they still have to be rounded to bf16
    int want_raw_logits;
    const float *last_logits;
    long last_logits_ld;
    int last_logits_round;
    // ... and with head_zero_rows set the head also leaves its tile maxima (last_tile_max, NULL when it could not) and
    // clears the probability rows head_zero_rows + i * head_zero_ld of its logit rows (see EPI_HEAD)
    float *head_zero_rows;
    long head_zero_ld;
    // (stream-batched pass: the logit rows' probability rows lie in the streams' own arenas - one pointer per logit row,
    //  head_zero_n of them, instead of head_zero_rows + i * head_zero_ld)
    float *head_zero_ptr[16];
    int head_zero_n;
    const float *last_tile_max;
    // profiling
    int prof_on;
    std::vector<hipEvent_t> ev_pool;
    std::vector<int> ev_class;
    size_t ev_used;
    float prof_ms[SD_N_PROFILE_CLASSES];
    int prof_cnt[SD_N_PROFILE_CLASSES];
    hipStream_t prof_stream;
};

static inline size_t esize(int dtype) { return dtype == SD_F32 ? 4 : 2; }
static inline bool is16(int dtype) { return dtype != SD_F32; }
// rounding code of a 16-bit head's fp32 accumulators (SD_NORM_ROUND_*)
static inline int round_code(int dtype) { return dtype == SD_BF16 ? SD_NORM_ROUND_BF16 : (dtype == SD_F16 ? SD_NORM_ROUND_F16 : 0); }
static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

static void copy_ptrs(std::vector<const void *> &dst, const void *const *src, int n) {
    dst.assign(n, nullptr);
    if (src)
        for (int i = 0; i < n; ++i) dst[i] = src[i];
}

extern "C" int sd_model_create(const sd_model_config *cfg, const sd_model_weights *w, sd_model **out) {
    SD_REQUIRE(cfg && w && out, "sd_model_create: null argument");
    SD_REQUIRE(cfg->arch == SD_ARCH_LLAMA || cfg->arch == SD_ARCH_OPT, "sd_model_create: unknown arch %d", cfg->arch);
    SD_REQUIRE(cfg->dtype == SD_F32 || cfg->dtype == SD_BF16 || cfg->dtype == SD_F16, "sd_model_create: unknown dtype %d", cfg->dtype);
    SD_REQUIRE(cfg->head_dim == 16 || cfg->head_dim == 32 || cfg->head_dim == 64 || cfg->head_dim == 128,
               "sd_model_create: head_dim %d not in {16,32,64,128}", cfg->head_dim);
    // a tensor-parallel shard holds n_heads / world query heads: its attention width is a fraction of hidden
    SD_REQUIRE(cfg->n_heads * cfg->head_dim <= cfg->hidden && (cfg->n_heads * cfg->head_dim) % 32 == 0,
               "sd_model_create: n_heads*head_dim must be <= hidden and a multiple of 32");
    SD_REQUIRE(cfg->hidden % 4 == 0 && cfg->hidden <= 8192, "sd_model_create: hidden must be a multiple of 4 and <= 8192");
    SD_REQUIRE(cfg->n_kv_heads > 0 && cfg->n_heads % cfg->n_kv_heads == 0, "sd_model_create: bad n_kv_heads");
    if (is16(cfg->dtype)) {
        SD_REQUIRE(cfg->hidden % 32 == 0 && cfg->inter % 32 == 0 && cfg->vocab % 16 == 0 && cfg->opt_proj_dim % 32 == 0,
                   "sd_model_create: the 16-bit paths need hidden/inter/proj %% 32 == 0 and vocab %% 16 == 0");
    }
    SD_REQUIRE(w->embed && w->lm_head && w->wqkv && w->wo && w->w_gate_up && w->w_down && w->norm1_w && w->norm2_w,
               "sd_model_create: missing weight pointers");
    sd_model *m = new sd_model();
    m->cfg = *cfg;
    m->w = *w;
    const int L = cfg->n_layers;
    copy_ptrs(m->wqkv, w->wqkv, L); copy_ptrs(m->bqkv, w->bqkv, L);
    copy_ptrs(m->wo, w->wo, L); copy_ptrs(m->bo, w->bo, L);
    copy_ptrs(m->wgu, w->w_gate_up, L); copy_ptrs(m->bfc1, w->b_fc1, L);
    copy_ptrs(m->wdown, w->w_down, L); copy_ptrs(m->bfc2, w->b_fc2, L);
    copy_ptrs(m->n1w, w->norm1_w, L); copy_ptrs(m->n1b, w->norm1_b, L);
    copy_ptrs(m->n2w, w->norm2_w, L); copy_ptrs(m->n2b, w->norm2_b, L);
    *out = m;
    return SD_OK;
}

extern "C" int sd_model_destroy(sd_model *m) {
    delete m;
    return SD_OK;
}

extern "C" int sd_pack_weight_bf16(const void *src, void *dst, int N, int K, void *stream) {
    SD_REQUIRE(src && dst && N > 0 && K > 0 && N % 16 == 0 && K % 32 == 0,
               "sd_pack_weight_bf16: need N %% 16 == 0 and K %% 32 == 0 (got %d x %d)", N, K);
    const size_t total = (size_t)N * K / 8;
    const int blocks = (int)std::min<size_t>((total + 255) / 256, 4096);
    hipLaunchKernelGGL(pack_weight_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const uint16_t *)src,
                       (uint16_t *)dst, N, K);
    SD_LAUNCH_CHECK();
    return SD_OK;
}

// ---- split-K policy.  A workgroup (4 waves) owns one n-tile x one k-slab and folds its waves in LDS, so
// SB slabs reach HBM.  SB is chosen so that about `target` workgroups exist (>= 4 per CU), with at least
// 8 k-steps (8 KiB of weights) per workgroup.
// Environment tunables, sampled when a session (or a spec handle) is created and by the public one-off GEMM entries - not
// per launch: a draft step is ~40 getenv() scans otherwise, on the host thread that has to keep the GPU fed.
#define SD_STREAM_MAX_ROWS 64                        // rows the streaming kernel's m-tile variants cover
#define SD_ROWS_MAX 144                              // rows the balanced one-workgroup-per-CU kernel covers (9 m-tiles: a 128-token prompt + gamma rows)
struct EnvTun {
    int gemm_ntw = 4, gemm_units = -1, small_path = 1, small_split_bytes = 0, tiny_split_bytes = 24576, fuse_embed_qkv = 1, head_tiles = 1;
    int attn_split_keys = 384, attn_keys_per_split = 256;
    int gemm_mm = 1, mm_mtw = 0, mm_s = 0, prefill_attn = 1, mm_slabs_min = 24;
    int wide_qkv = 1, tp_one_slab = 1, gemm_rows = 1, cus = 0, fuse_attn_o = 1, ao_stamps = 0, ao_delay = 300, ao_gap = 100, norm_on_load = 2, rows_max = SD_ROWS_MAX;
};
static EnvTun g_env;
static void refresh_env() {
    auto geti = [](const char *name, int dflt) { const char *e = getenv(name); return e ? atoi(e) : dflt; };
    g_env.gemm_ntw = geti("SD_GEMM_NTW", 4);
    g_env.gemm_units = geti("SD_GEMM_UNITS", -1);
    // 1 (default since round 4): a small model's <= 4-row step runs the layers' two norm launches as prologues of QKV / gate-up
    // (llama-68m 94.9 -> 92.3 us, opt-125m 426-440 -> 417 us per draft step); 2: every seam a prologue incl. the head (slower);
    // 0: the per-op chain.  Rounds 2-4 measured the prologue route 75 % slower: that was a 254-VGPR compile of gemm_small
    g_env.small_path = geti("SD_SMALL_PATH", 1);
    g_env.small_split_bytes = geti("SD_SMALL_SPLIT_BYTES", 0);
    g_env.tiny_split_bytes = geti("SD_TINY_SPLIT_BYTES", 24576);   // k-slab size of <= 16-row GEMMs on matrices of <= 8 MB (0: the big-matrix policy)
    g_env.fuse_embed_qkv = geti("SD_FUSE_EMBED_QKV", 1);
    g_env.head_tiles = geti("SD_HEAD_TILES", 1);
    g_env.attn_split_keys = geti("SD_ATTN_SPLIT_KEYS", 384);          // keys per workgroup above which a group's keys are split
    g_env.attn_keys_per_split = std::max(16, geti("SD_ATTN_KEYS_PER_SPLIT", 256));
    g_env.ao_stamps = geti("SD_AO_STAMPS", 0);
    g_env.ao_delay = geti("SD_AO_DELAY", 300);        // 10 ns ticks the O workgroups hold their weight requests back (fused_kernels.h)
    g_env.ao_gap = geti("SD_AO_GAP", 100);            // ... and pause after every 8 requests
    g_env.fuse_attn_o = geti("SD_FUSE_ATTN_O", 1);    // 0: attention and the O projection as two launches (A/B runs, bit-compare tests)
    g_env.norm_on_load = geti("SD_NORM_ON_LOAD", 2);  // 0: residual+norm launches stay; 1: attention -> MLP seam only; 2: both seams (A/B runs, compare tests)
    g_env.gemm_mm = geti("SD_GEMM_MM", 1);            // 1 (default): prefill passes the balanced kernel does not take run on gemm_bf16_mm (mm_kernels.h) instead of gemm_bf16_tiled
    g_env.prefill_attn = geti("SD_PREFILL_ATTN", 1);  // 0: prefill passes keep attn_kernel's 8-row groups (A/B runs, compare tests)
    g_env.mm_slabs_min = geti("SD_MM_SLABS_MIN", 24); // rows from which the k-slab GEMMs (O / down) of a pass take gemm_bf16_mm instead of the balanced kernel
    g_env.mm_mtw = geti("SD_MM_MTW", 0);              // (sweeps) m-tiles per wave of gemm_bf16_mm: 2 = 128-row blocks, 4 = 256-row blocks; 0 = by row count
    g_env.mm_s = geti("SD_MM_S", 0);                  // (sweeps) k-slabs of gemm_bf16_mm; 0 = planned

    g_env.gemm_rows = geti("SD_GEMM_ROWS", 1);        // 0: 17..64-row GEMMs stay on the streaming kernel (A/B runs, bit-compare tests)
    g_env.wide_qkv = geti("SD_WIDE_QKV", 1);          // 0: a QKV projection with <= 128 n-tiles keeps one 4-wave workgroup per tile (A/B, compare tests)
    g_env.tp_one_slab = geti("SD_TP_ONE_SLAB", 1);    // 0: a shard's O / down projection keeps its k-slabs + the fold launch in front of the all-reduce (A/B)
    g_env.rows_max = std::min(SD_ROWS_MAX, std::max(SD_STREAM_MAX_ROWS, geti("SD_GEMM_ROWS_MAX", SD_ROWS_MAX)));   // 65..this many rows take the balanced kernel, more the LDS-tiled one
    if (!g_env.cus) {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0)
            g_env.cus = n;
        else
            (void)hipGetLastError();                  // no device (CPU container): plans are made for the MI355X's 256 CUs
    }
}

static int gemm_ntw(int N, int M) {
    if (M <= 16) return 1;
    const int ntl = N / 16;
    // n-tiles per wave: more of them amortise the activation fragment loads, fewer give more workgroups.  Stream-batched
    // decode, 4 / 6 / 8 / 10 streams x 5 rows (bench.py --batch-streams): 529 / 727 / 861 / 990 tok/s with 4 tiles,
    // 504 / 753 / 808 / 908 with 2; 8 tiles: register pressure, slower everywhere.
    const int want = g_env.gemm_ntw;
    if (want >= 8 && ntl % 8 == 0) return 8;
    if (want >= 4 && ntl % 4 == 0) return 4;
    if (want >= 2 && ntl % 2 == 0) return 2;
    return 1;
}

static void gemm_split(int N, int K, int M, int *S_out, int *ks_per_out) {
    const int NTL = N / 16 / gemm_ntw(N, M), KS = K / 32;
    int max_s = KS / 8;
    if (max_s < 1) max_s = 1;
    int S = 1;
    if (g_env.gemm_units >= 0) {
        S = (g_env.gemm_units + NTL / 2) / NTL;
    } else if (M > 16) {
        // many rows: every extra slab costs a write + a read of Mpad*N floats, which at 64 rows rivals the weight bytes
        // (256*S/K of them), so take the S that minimises (weights + slab traffic) / CU-fill efficiency
        const double wbytes = (double)N * K * 2.0, slab = 2.0 * (double)align_up(M, 16) * N * 4.0;
        double best = 1e300;
        for (int c = 1; c <= max_s && c <= 16; ++c) {
            const int blocks = NTL * c;
            if (blocks < 256 && c < max_s && c < 16) continue;
            const double eff = (double)blocks / (double)(((blocks + 255) / 256) * 256);
            const double cost = (wbytes + slab * c) / eff * (blocks < 768 ? 1.25 : 1.0);   // too few workgroups under-fill the load path
            if (cost < best) { best = cost; S = c; }
        }
    } else if ((size_t)N * K * 2 <= ((size_t)8 << 20) && g_env.tiny_split_bytes > 0) {
        // a matrix of a small model (the draft's O / down projection: 1.2 / 4.7 MB): its launch is latency, not bytes, and
        // every slab is another 16-byte load per column group in the consumer's fold - so few slabs, each workgroup streaming
        // about tiny_split_bytes (24 KiB: down 12 -> 4 slabs, O 3 -> 1; on one box llama-68m / opt-125m steps of 91.5 / 415.4 us
        // with the big-matrix policy, 87.3 / 387.6 at 24 KiB, 89.4 / 397.0 at 32, 90.2 / 402.5 at 48, 93.7 / 420.7 at 96 -
        // tools/draft_step_bench.py SD_TINY_SPLIT_BYTES=...)
        S = (int)((K * 32 + g_env.tiny_split_bytes / 2) / g_env.tiny_split_bytes);
    } else if (NTL < 1024) {
        // fewest slabs that give >= 1024 workgroups, preferring a workgroup count that fills all 256 CUs evenly
        // (measured on MI355X: 7.5 workgroups per CU runs 10 % slower than 15 per CU, tools/gemm_bench.py)
        double best = -1.0;
        for (int c = 1; c <= max_s && c <= 16; ++c) {
            const int blocks = NTL * c;
            if (blocks < 1024 && c < max_s && c < 16) continue;
            const double eff = (double)blocks / (double)(((blocks + 255) / 256) * 256);
            if (eff > best + 1e-9) { best = eff; S = c; }
            if (blocks >= 4096) break;
        }
    }
    if (S < 1) S = 1;
    if (S > max_s) S = max_s;
    int ks_per = (KS + S - 1) / S;
    S = (KS + ks_per - 1) / ks_per;
    *S_out = S;
    *ks_per_out = ks_per;
}

// Which kernel a bf16 GEMM over M rows takes and how its k-range is cut.  Prefill chunks of more than 64 rows take the
// LDS-tiled kernel: tools/gemm_bench.py at the 13b shapes, one tiled pass over 128 / 256 rows costs 171 / 257 us per
// layer against 324 / 648 us for 2 / 4 streaming passes of 64.  At 33..64 rows it is ~10 % ahead as a bare GEMM (qkv
// 35.6 us against 41.1, gate/up 59.8 / 63.8) but needs slabs plus the stand-alone epilogues where the streaming kernel
// fuses them, and the whole forward comes out even (bench.py --batch-streams 8 / 12), so those stay on the streaming
// kernel.  Slab count: about 480 workgroups (measured optimum for all four shapes at 64, 128 and 256 rows).
#define SD_MAX_FWD_ROWS 256
struct GemmPlan { bool tiled; int S, ksp, mtw, mm; };
struct RowsPlan { bool ok; int S, ksp, NG, grid, nwn, nwk, nld; };
static RowsPlan rows_plan(int N, int K, int M, bool fused);
// fused: the caller wants the GEMM's whole k-range per workgroup (QKV / activation epilogue inside the launch)
static GemmPlan gemm_plan(int N, int K, int M, bool x_tiled = true, bool fused = false) {
    GemmPlan p = {};
    const int KS = K / 32, Mpad = (int)align_up(M, 16);
    static const int tiled_min = getenv("SD_GEMM_TILED_MIN") ? atoi(getenv("SD_GEMM_TILED_MIN")) : 65;
    // (65..SD_MAX_ROWS rows - 8 streams x 9 verify rows - stay on the balanced one-workgroup-per-CU kernel, with its fused
    //  epilogues, when both of its plans for the shape are good; prefill chunks of that size take it too)
    // (a k-slab GEMM - O / down projection - of a 24..144-row pass is faster on gemm_bf16_mm with 128-row blocks and 6 slabs
    //  than on the balanced kernel - 12.7 / 28.1 us against 15.3 / 29.6 at 40 rows (8 streams x 5), 13.6 / 28.3 against 16.7 /
    //  31.6 at 60, 14.7 / 30.2 against 18.0 / 36.3 at 80, 18.1 / 36.2 against 19.7 / 42.5 at 132; equal at 24 - tools/mm_bench.py.
    //  QKV and gate/up with their fused epilogues stay on the balanced kernel: 36.9 / 56.1 us against 52.0 / 60.5 at 80 rows,
    //  44.9 / 70.1 against 50.1 / 74.7 at 132)
    const bool mm_slabs = g_env.gemm_mm && !fused && M >= g_env.mm_slabs_min && (size_t)N * K >= ((size_t)16 << 20) && N / 16 / 8 < 64;
    const bool rows_take = M > SD_STREAM_MAX_ROWS && M <= g_env.rows_max && !mm_slabs && rows_plan(N, K, M, fused).ok;
    if (x_tiled && (M >= tiled_min || mm_slabs) && !rows_take && (N / 16) % 8 == 0 && KS % 2 == 0 && KS >= 16) {
        p.tiled = true;
        if (g_env.gemm_mm && (Mpad > 64 || mm_slabs)) {
            // gemm_bf16_mm (mm_kernels.h): 256-row blocks (mtw 4) or 128-row blocks (mtw 2) x 128 columns.  Fused (the caller
            // wants the whole k-range per block for a QKV / activation epilogue): the block shape that fills the CUs better
            // (13b at 256 rows: gate/up 216 blocks of 256 rows, QKV 240 blocks of 128).  Otherwise 256-row blocks past 128
            // rows and k-slabs for about one block per CU (tools/mm_bench.py: O / down 6 slabs, QKV 2).
            const int G = g_env.cus > 0 ? g_env.cus : 256, NB = N / 16 / 8;
            auto blocks_of = [&](int mtw) { const int BMT = 4 * mtw; return ((Mpad / 16 + BMT - 1) / BMT) * NB; };
            auto fill = [&](int blocks) { return (double)blocks / (double)(((blocks + G - 1) / G) * G); };
            p.mm = 1;
            p.mtw = Mpad <= 128 ? 2 : 4;
            if (fused && Mpad > 128 && fill(blocks_of(2)) > fill(blocks_of(4)) + 0.05) p.mtw = 2;
            if (g_env.mm_mtw) p.mtw = g_env.mm_mtw;
            const int blocks = blocks_of(p.mtw);
            int S = fused ? 1 : (g_env.mm_s ? g_env.mm_s : std::max(1, (G - G / 16 + blocks / 2) / blocks));
            S = std::min(std::min(S, 8), std::max(1, KS / 8));     // (every slab is a write + a read of Mpad x N floats)
            p.ksp = (int)align_up((KS + S - 1) / S, 2);
            p.S = (KS + p.ksp - 1) / p.ksp;
            return p;
        }
        p.mtw = Mpad <= 64 ? 2 : 4;
        const int MB = (Mpad / 16 + 2 * p.mtw - 1) / (2 * p.mtw), blocks = MB * (N / 16 / 8);
        int S = std::max(1, (480 + blocks / 2) / blocks);
        S = std::min(S, std::max(1, KS / 16));
        p.ksp = (int)align_up((KS + S - 1) / S, 2);
        p.S = (KS + p.ksp - 1) / p.ksp;
        return p;
    }
    gemm_split(N, K, std::min(M, 64), &p.S, &p.ksp);
    return p;
}

// can a GEMM with this plan run a fused QKV / activation epilogue?  (the streaming and balanced kernels: always, with
// SB = 1; the LDS-tiled kernel: never; gemm_bf16_mm: with one slab)
static bool fused_plan_ok(const GemmPlan &p) { return !p.tiled || (p.mm && p.S == 1); }

// gemm_bf16_mm (mm_kernels.h): EPI != EPI_PART needs pl.S == 1 (the block holds the whole k-range)
template <int EPI, typename H = bf16_t>
static void launch_gemm_mm(const void *W, const void *X, float *part, int M, int Mpad, int N, int K, const GemmPlan &pl,
                           const GemmEpiT<H> &e, hipStream_t st) {
    const int BMT = 4 * pl.mtw, MB = (Mpad / 16 + BMT - 1) / BMT, NB = N / 16 / 8;
    const dim3 grid(MB * NB * pl.S);
    const size_t lds = (size_t)3 * (8 + BMT) * 2 * 1024;          // NBUF = 3 stages x KT = 2 k-tiles x (8 W + BMT X) KiB
    auto go = [&](auto kern) {
        static bool attr = false;                                 // (one flag per instantiation of this lambda = per kernel)
        if (!attr) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024);
            attr = true;
        }
        hipLaunchKernelGGL(kern, grid, dim3(MM_THREADS), lds, st, (const u32x4 *)W, (const u32x4 *)X, part, M, Mpad, N, K, pl.S,
                           pl.ksp, e);
    };
    // 256-row blocks hold every row of a pass (<= 256 rows): their weight tiles are read once chip-wide -> non-temporal
    if (pl.mtw == 2) go(gemm_bf16_mm<2, EPI, H, false>);
    else go(gemm_bf16_mm<4, EPI, H, true>);
}

template <typename H = bf16_t>
static void launch_gemm_tiled(const void *W, const void *X, float *part, int M, int Mpad, int N, int K,
                              const GemmPlan &pl, hipStream_t st) {
    if (pl.mm) {
        GemmEpiT<H> e0 = {};
        launch_gemm_mm<EPI_PART, H>(W, X, part, M, Mpad, N, K, pl, e0, st);
        return;
    }
    const int MB = (Mpad / 16 + 2 * pl.mtw - 1) / (2 * pl.mtw), NB = N / 16 / 8;
    const dim3 grid(MB * NB * pl.S);
    const size_t lds = (size_t)2 * (8 + 2 * pl.mtw) * 2 * 1024;   // 2 buffers x (8 W + 2*MTW X tiles) x KT = 2 k-tiles
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(gemm_bf16_tiled<2, 4, 2, H>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(gemm_bf16_tiled<4, 4, 2, H>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
        attr = true;
    }
    if (pl.mtw == 2)
        hipLaunchKernelGGL((gemm_bf16_tiled<2, 4, 2, H>), grid, dim3(256), lds, st, (const u32x4 *)W, (const u32x4 *)X, part, M,
                           Mpad, N, K, pl.S, pl.ksp);
    else
        hipLaunchKernelGGL((gemm_bf16_tiled<4, 4, 2, H>), grid, dim3(256), lds, st, (const u32x4 *)W, (const u32x4 *)X, part, M,
                           Mpad, N, K, pl.S, pl.ksp);
}

// ---- 17..64 rows: the balanced one-workgroup-per-CU kernel (rows_kernels.h).  S k-slabs x NG n-groups = one workgroup per
// CU; a fused epilogue needs S == 1.  The plan minimises (weight bytes + slab traffic) / how evenly the n-tiles divide.
static RowsPlan rows_plan(int N, int K, int M, bool fused) {
    RowsPlan p = {};
    if (!g_env.gemm_rows || M <= 16 || M > SD_ROWS_MAX || N % 16 || K % 32) return p;
    const int G = g_env.cus > 0 ? g_env.cus : 256, NT = N / 16, KS = K / 32, Mpad = (int)align_up(M, 16);
    const double wbytes = (double)N * K * 2.0, slab = 2.0 * Mpad * (double)N * 4.0;
    double best = 1e300;
    for (int S = 1; S <= (fused ? 1 : 16); S *= 2) {
        if (G % S || KS / S < 8) break;
        const int NG = G / S, tpg = (NT + NG - 1) / NG;
        if (NT < NG || tpg > GR_MAX_TILES) continue;
        const double eff = (double)NT / NG / tpg;
        const double cost = (wbytes + (fused ? 0.0 : slab * S)) / eff;
        if (cost < best) {
            best = cost;
            p.S = S; p.NG = NG;
            p.ksp = (KS + S - 1) / S;
            p.ok = eff >= 0.8;
        }
    }
    if (p.ok) {
        p.S = (KS + p.ksp - 1) / p.ksp;
        p.grid = p.NG * p.S;
        // wave grid of a workgroup: nwn compute waves per k-group (one n-tile each) x nwk k-groups + loader waves, <= 16
        p.nwn = (NT + p.NG - 1) / p.NG;
        p.nwk = std::min(4, (16 - 1) / p.nwn);
        if (Mpad > 64) p.nwk = std::min(p.nwk, 3);                 // (5 m-tiles: 2 x 4 k-groups x 4 k-steps x 5 KiB exceed the CU's LDS)
        if (Mpad > 80) p.nwk = std::min(p.nwk, 2);                 // (8 / 9 m-tiles: 2 x 2 x 4 x 8 KiB = 128 KiB, x 9 = 144 KiB, is what fits)
        p.nld = std::min(p.nwk, 16 - p.nwn * p.nwk);
    }
    return p;
}

template <int EPI, typename H>
static int launch_gemm_rows(const void *W, const void *X, float *part, int M, int Mpad, int N, int K, const RowsPlan &pl,
                            const GemmEpiT<H> &e, hipStream_t st) {
    // m-tiles of the kernel instance: 2..5 as they are, 6..8 all take the 8-tile instance (the loader re-reads the last
    // real tile for the missing ones, the epilogue drops rows >= M)
    const int MT = Mpad / 16 <= 5 ? Mpad / 16 : (Mpad / 16 <= 8 ? 8 : 9);
    // activation panel (2 buffers x nwk k-groups x CH k-steps x MT tiles), reused as the fold buffer (nwn tiles x nwk x
    // MT); chunks of 8 k-steps where the panel fits the CU's LDS (one workgroup per CU owns all of it), else 6 or 4 -
    // with chunks of 4 a compute wave's weight burst spans two chunks (8 KiB requested together: WBM = 2)
    constexpr size_t lds_cap = 158 * 1024;
    int ch = 4;
    for (int c2 : {8, 6})
        if ((size_t)1024 * MT * 2 * pl.nwk * c2 <= lds_cap) { ch = c2; break; }
    if (MT == 2) ch = 8;                                          // (2 x 4 x 8 x 2 KiB always fits)
    if (MT == 3 && ch == 4) ch = 6;                               // (2 x 4 x 6 x 3 KiB = 144 KiB: the widest 3-tile panel)
    const size_t lds = (size_t)1024 * MT * std::max(2 * pl.nwk * ch, pl.nwk * pl.nwn);
    SD_REQUIRE(lds <= lds_cap && Mpad / 16 <= MT, "gemm_rows: %d rows, %zu bytes of LDS", M, lds);
    auto go = [&](auto mt_c, auto ch_c, auto wbm_c) {
        constexpr int MTc = decltype(mt_c)::value, CHc = decltype(ch_c)::value, WBMc = decltype(wbm_c)::value;
        static bool attr = false;                                 // (one flag per instantiation)
        if (!attr) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(gemm_bf16_rows<MTc, EPI, H, CHc, WBMc>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_cap);
            attr = true;
        }
        hipLaunchKernelGGL((gemm_bf16_rows<MTc, EPI, H, CHc, WBMc>), dim3(pl.grid), dim3(GR_THREADS), lds, st, (const u32x4 *)W,
                           (const u32x4 *)X, part, M, Mpad, N, K, pl.NG, pl.ksp, pl.nwn, pl.nwk, pl.nld, e);
    };
    using std::integral_constant;
    using I1 = integral_constant<int, 1>;
    using I2 = integral_constant<int, 2>;
    if (MT == 2) go(integral_constant<int, 2>{}, integral_constant<int, 8>{}, I1{});
    else if (MT == 3) { if (ch == 8) go(integral_constant<int, 3>{}, integral_constant<int, 8>{}, I1{});
                        else go(integral_constant<int, 3>{}, integral_constant<int, 6>{}, I1{}); }
    else if (MT == 4) { if (ch == 8) go(integral_constant<int, 4>{}, integral_constant<int, 8>{}, I1{});
                        else if (ch == 6) go(integral_constant<int, 4>{}, integral_constant<int, 6>{}, I1{});
                        else go(integral_constant<int, 4>{}, integral_constant<int, 4>{}, I2{}); }
    else if (MT == 5) { if (ch == 6) go(integral_constant<int, 5>{}, integral_constant<int, 6>{}, I1{});       // (8 never fits at 5 m-tiles)
                        else go(integral_constant<int, 5>{}, integral_constant<int, 4>{}, I2{}); }
    else if (MT == 8 && ch == 4) go(integral_constant<int, 8>{}, integral_constant<int, 4>{}, I2{});
    else if (MT == 9 && ch == 4) go(integral_constant<int, 9>{}, integral_constant<int, 4>{}, I2{});
    else { sd_set_error("gemm_rows: %d rows (chunk %d)", M, ch); return SD_ERR_INVALID; }
    return SD_OK;
}

static size_t gemm_part_floats(const sd_model_config &c, int N, int K, int rows) {
    if (!is16(c.dtype)) return (size_t)rows * N;
    size_t best = 0;
    for (int m = 1; m <= rows; m = (m % 16 == 0 ? m + 1 : (int)align_up(m, 16))) {       // every plan class: 1, 16, 17, 32, 33, ...
        const GemmPlan pl = gemm_plan(N, K, m);
        best = std::max(best, (size_t)pl.S * align_up(m, 16) * N);
        const RowsPlan rp = rows_plan(N, K, m, false);
        if (rp.ok) best = std::max(best, (size_t)rp.S * align_up(m, 16) * N);
    }
    return best;
}

static int qkv_cols(const sd_model_config &c) { return (c.n_heads + 2 * c.n_kv_heads) * c.head_dim; }
static int q_dim(const sd_model_config &c) { return c.n_heads * c.head_dim; }     // attention width (== hidden unless sharded)
static int gu_cols(const sd_model_config &c) { return c.arch == SD_ARCH_LLAMA ? 2 * c.inter : c.inter; }
static int embed_dim(const sd_model_config &c) { return c.arch == SD_ARCH_OPT ? c.opt_proj_dim : c.hidden; }

extern "C" size_t sd_session_kv_bytes(const sd_model *m, int max_seq) {
    const sd_model_config &c = m->cfg;
    return (size_t)c.n_layers * 2 * c.n_kv_heads * max_seq * c.head_dim * esize(c.dtype);
}

struct ScratchPlan {
    size_t x, x2, h, h2, aoctr, ssq, finctr, q, attn, act, e, apart, part, spart, tmax, tp_in, tp_out, total, part_floats, spart_floats;
};
static ScratchPlan plan_scratch(const sd_model_config &c, int rows) {
    ScratchPlan p;
    const size_t es = esize(c.dtype);
    const int ed = embed_dim(c);
    const int wide = std::max(c.hidden, ed);
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off = align_up(off + bytes, 256); return o; };
    const size_t trows = align_up(rows, 16);                     // GEMM operands live in 16-row tiles (xoff)
    p.x = take((size_t)rows * c.hidden * es);
    p.x2 = take((size_t)SMALL_MAX_ROWS * c.hidden * es);
    p.h = take(trows * wide * es);
    p.h2 = take(trows * wide * es);
    p.aoctr = take(256 + (size_t)AO_STAMP_WGS * 8 * sizeof(long long));   // counter line, status line, stamp records
    p.finctr = take((size_t)(c.hidden / 16 + 1) * sizeof(unsigned));      // per-tile arrival counters of gemm_bf16_stream_fin
    p.ssq = take((size_t)16 * (c.hidden / 16 + 1) * sizeof(float));      // per-tile sums of squares of <= 16 rows (norm on load)
    p.q = take((size_t)rows * c.hidden * es);
    p.attn = take(trows * c.hidden * es);
    p.act = take(trows * c.inter * es);
    p.e = take(trows * ed * es);
    p.apart = take((size_t)64 * c.n_heads * 8 * (c.head_dim + 2) * sizeof(float));
    size_t pf = 0;
    pf = std::max(pf, gemm_part_floats(c, qkv_cols(c), c.hidden, rows));
    pf = std::max(pf, gemm_part_floats(c, c.hidden, q_dim(c), rows));
    pf = std::max(pf, gemm_part_floats(c, gu_cols(c), c.hidden, rows));
    pf = std::max(pf, gemm_part_floats(c, c.hidden, c.inter, rows));
    pf = std::max(pf, gemm_part_floats(c, c.vocab, ed, rows));
    if (ed != c.hidden) {
        pf = std::max(pf, gemm_part_floats(c, c.hidden, ed, rows));
        pf = std::max(pf, gemm_part_floats(c, ed, c.hidden, rows));
    }
    p.part_floats = pf;
    p.part = take(pf * sizeof(float));
    // small-model path: slabs [S][16][hidden] of its O / down GEMMs, S <= 16
    p.spart_floats = (size_t)16 * 16 * c.hidden;
    p.spart = take(p.spart_floats * sizeof(float));
    p.tmax = take((size_t)SD_MAX_ROWS * (c.vocab / 16 + 1) * sizeof(float));
    p.tp_in = take((size_t)rows * c.hidden * sizeof(float));
    p.tp_out = take((size_t)rows * c.hidden * sizeof(float));
    p.total = off;
    return p;
}

// Rows one sd_session_forward call may carry: 256 when every per-layer GEMM of the model can take the tiled kernel
// (or the model is fp32, whose simple GEMM has no row limit), else the streaming kernel's 64.
extern "C" int sd_model_max_rows(const sd_model *m) {
    if (!m) return 0;
    refresh_env();
    const sd_model_config &c = m->cfg;
    if (!is16(c.dtype)) return SD_MAX_FWD_ROWS;
    const int ed = embed_dim(c);
    const int shapes[][2] = {{qkv_cols(c), c.hidden}, {c.hidden, q_dim(c)}, {gu_cols(c), c.hidden}, {c.hidden, c.inter},
                             {c.hidden, ed}, {ed, c.hidden}};
    for (int i = 0; i < (ed != c.hidden ? 6 : 4); ++i)
        if (!gemm_plan(shapes[i][0], shapes[i][1], SD_MAX_FWD_ROWS).tiled) return SD_STREAM_MAX_ROWS;
    return SD_MAX_FWD_ROWS;
}

extern "C" size_t sd_session_scratch_bytes(const sd_model *m, int max_rows) {
    refresh_env();
    return plan_scratch(m->cfg, std::min(max_rows, sd_model_max_rows(m))).total;
}

extern "C" int sd_session_create(sd_model *m, int max_seq, int max_rows, void *kv_arena, void *scratch,
                                 sd_session **out) {
    SD_REQUIRE(m && kv_arena && scratch && out, "sd_session_create: null argument");
    SD_REQUIRE(max_seq > 0 && max_rows > 0, "sd_session_create: bad sizes");
    if (m->cfg.arch == SD_ARCH_OPT)
        SD_REQUIRE(max_seq <= m->cfg.max_pos, "sd_session_create: OPT is limited to %d positions", m->cfg.max_pos);
    else
        SD_REQUIRE(max_seq <= m->cfg.max_pos, "sd_session_create: rope table has %d rows, max_seq %d", m->cfg.max_pos, max_seq);
    refresh_env();
    sd_session *s = new sd_session();
    s->m = m;
    s->max_seq = max_seq;
    max_rows = std::min(max_rows, sd_model_max_rows(m));
    s->max_rows = max_rows;
    s->kv = (char *)kv_arena;
    s->scratch = (char *)scratch;
    const ScratchPlan p = plan_scratch(m->cfg, max_rows);
    s->x = s->scratch + p.x;
    s->x2 = s->scratch + p.x2;
    s->spart = (float *)(s->scratch + p.spart);
    s->spart_floats = p.spart_floats;
    s->tile_max = (float *)(s->scratch + p.tmax);
    s->tp = nullptr;
    s->kv_fp8 = 0;
    s->kv_scale = nullptr;
    s->tp_in = (float *)(s->scratch + p.tp_in);
    s->tp_out = (float *)(s->scratch + p.tp_out);
    s->head_zero_rows = nullptr;
    s->head_zero_ld = 0;
    s->head_zero_n = 0;
    s->last_tile_max = nullptr;
    s->h = s->scratch + p.h;
    s->h2 = s->scratch + p.h2;
    s->ao_ctr = (unsigned *)(s->scratch + p.aoctr);
    s->ao_epoch = 0;
    s->wait_status = s->ao_ctr + 32;                              // (a 128-byte line of its own)
    s->ao_stamps = (long long *)(s->ao_ctr + 64);
    s->resync = 0;
    s->test_skew = s->skew_now = 0;
    s->ssq = (float *)(s->scratch + p.ssq);
    s->fin_ctr = (unsigned *)(s->scratch + p.finctr);
    s->fin_epoch = 0;
    SD_HIP_CHECK(hipMemset(s->fin_ctr, 0, (size_t)(m->cfg.hidden / 16 + 1) * sizeof(unsigned)));
    SD_HIP_CHECK(hipMemset(s->ao_ctr, 0, 256 + (size_t)AO_STAMP_WGS * 8 * sizeof(long long)));
    s->qbuf = s->scratch + p.q;
    s->attn = s->scratch + p.attn;
    s->act = s->scratch + p.act;
    s->ebuf = s->scratch + p.e;
    s->attn_part = (float *)(s->scratch + p.apart);
    s->part = (float *)(s->scratch + p.part);
    s->part_floats = p.part_floats;
    s->want_raw_logits = 0;
    s->last_logits = nullptr;
    s->prof_on = 0;
    s->ev_used = 0;
    s->prof_stream = nullptr;
    memset(s->prof_ms, 0, sizeof(s->prof_ms));
    memset(s->prof_cnt, 0, sizeof(s->prof_cnt));
    *out = s;
    return SD_OK;
}

// Switch the session's KV arena to OCP fp8 e4m3 (BASELINE config 5).  The arena then holds 1 byte per element
// (half of sd_session_kv_bytes for a 16-bit model); `scales` is a device array [n_layers][2][n_kv_heads] (x is stored as
// fp8(x / scale)); call before the first forward.
extern "C" int sd_session_set_kv_fp8(sd_session *s, const float *scales) {
    SD_REQUIRE(s && scales, "sd_session_set_kv_fp8: null argument");
    SD_REQUIRE(is16(s->m->cfg.dtype) && s->m->cfg.head_dim >= 32, "sd_session_set_kv_fp8: needs a 16-bit model with head_dim >= 32");
    s->kv_fp8 = 1;
    s->kv_scale = scales;
    return SD_OK;
}

extern "C" int sd_session_destroy(sd_session *s) {
    if (!s) return SD_OK;
    for (hipEvent_t e : s->ev_pool) (void)hipEventDestroy(e);
    delete s;
    return SD_OK;
}

// ---- profiling brackets ------------------------------------------------------------------
struct ProfScope {
    sd_session *s;
    hipStream_t st;
    bool on;
    ProfScope(sd_session *s_, int cls, hipStream_t st_) : s(s_), st(st_), on(s_->prof_on != 0) {
        if (!on) return;
        while (s->ev_pool.size() < s->ev_used + 2) {
            hipEvent_t e;
            if (hipEventCreate(&e) != hipSuccess) { on = false; return; }
            s->ev_pool.push_back(e);
        }
        s->ev_class.push_back(cls);
        (void)hipEventRecord(s->ev_pool[s->ev_used], st);
        s->prof_stream = st;
    }
    ~ProfScope() {
        if (!on) return;
        (void)hipEventRecord(s->ev_pool[s->ev_used + 1], st);
        s->ev_used += 2;
    }
};

extern "C" int sd_profile_enable(sd_session *s, int on) {
    SD_REQUIRE(s, "sd_profile_enable: null session");
    s->prof_on = on;
    return SD_OK;
}

extern "C" int sd_profile_read(sd_session *s, float *ms_out, int *count_out) {
    SD_REQUIRE(s && ms_out && count_out, "sd_profile_read: null argument");
    if (s->ev_used) {
        SD_HIP_CHECK(hipEventSynchronize(s->ev_pool[s->ev_used - 1]));
        for (size_t i = 0; i < s->ev_used; i += 2) {
            float ms = 0.f;
            SD_HIP_CHECK(hipEventElapsedTime(&ms, s->ev_pool[i], s->ev_pool[i + 1]));
            const int c = s->ev_class[i / 2];
            s->prof_ms[c] += ms;
            s->prof_cnt[c] += 1;
        }
    }
    for (int i = 0; i < SD_N_PROFILE_CLASSES; ++i) {
        ms_out[i] = s->prof_ms[i];
        count_out[i] = s->prof_cnt[i];
        s->prof_ms[i] = 0.f;
        s->prof_cnt[i] = 0;
    }
    s->ev_used = 0;
    s->ev_class.clear();
    return SD_OK;
}


// ---- tensor parallelism (BASELINE config 5: Llama-2-70b over 8 GPUs) ---------------------------------
// Megatron-style sharding of one decoder: QKV and gate/up split by output columns (each rank owns n_heads / world query
// heads, n_kv_heads / world KV heads and inter / world MLP columns), O and down split by input rows, so a layer needs two
// all-reduces of the [rows][hidden] partial outputs (fp32: the sum is rounded to the model dtype once, like an unsharded
// dot product).  At gamma + 1 = 5 rows x 8192 that is 160 KB: latency-bound on xGMI, RCCL picks its low-latency
// protocol on its own for such sizes.  The reference has no counterpart (it is single-process); the math
// follows modeling_llama.py:292-393 with pretraining_tp = 1.
//   * RCCL group: one process per GPU, the communicator is created from a unique id the host broadcasts;
//     the library is resolved with dlopen at first use (a process that already holds RCCL - torch does - shares it).
//   * loopback group: all ranks live in ONE process on one GPU, each in its own host thread and stream; the all-reduce
//     is two event-ordered rendezvous and a sum kernel.  It exists so that the sharded forward can be tested on a
//     one-GPU box (tests/test_gpu_native_parity.py) with the very kernels the RCCL path runs.
#include <dlfcn.h>
#include <condition_variable>
#include <mutex>

struct GemmOut {
    int S;
    size_t stride_s;   // floats between k-slices
};

struct TpLoop {
    int world;
    std::mutex mu;
    std::condition_variable cv;
    int arrived = 0;
    long gen = 0;
    const float *bufs[16];
    hipEvent_t ev_in[16], ev_out[16];
    int refs = 0;
    void barrier() {
        std::unique_lock<std::mutex> lk(mu);
        const long g = gen;
        if (++arrived == world) { arrived = 0; ++gen; cv.notify_all(); }
        else cv.wait(lk, [&] { return gen != g; });
    }
};

struct sd_tp {
    int rank, world;
    void *comm;          // ncclComm_t
    TpLoop *loop;
};

typedef int (*nccl_allreduce_fn)(const void *, void *, size_t, int, int, void *, hipStream_t);
typedef int (*nccl_getid_fn)(void *);
struct sd_nccl_id { char b[128]; };                  // ncclUniqueId is passed by value
typedef int (*nccl_initrank_fn)(void **, int, sd_nccl_id, int);
typedef int (*nccl_destroy_fn)(void *);
typedef int (*nccl_allgather_fn)(const void *, void *, size_t, int, void *, hipStream_t);
static struct { void *lib; nccl_allreduce_fn allreduce; nccl_getid_fn getid; nccl_initrank_fn initrank; nccl_destroy_fn destroy;
                nccl_allgather_fn allgather; } g_rccl;

static int rccl_load() {
    if (g_rccl.lib) return SD_OK;
    void *h = dlopen("librccl.so.1", RTLD_NOW | RTLD_NOLOAD);          // the copy this process already holds, if any
    if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) { sd_set_error("tensor parallelism needs RCCL: %s", dlerror()); return SD_ERR_HIP; }
    g_rccl.allreduce = (nccl_allreduce_fn)dlsym(h, "ncclAllReduce");
    g_rccl.getid = (nccl_getid_fn)dlsym(h, "ncclGetUniqueId");
    g_rccl.initrank = (nccl_initrank_fn)dlsym(h, "ncclCommInitRank");
    g_rccl.destroy = (nccl_destroy_fn)dlsym(h, "ncclCommDestroy");
    g_rccl.allgather = (nccl_allgather_fn)dlsym(h, "ncclAllGather");
    if (!g_rccl.allreduce || !g_rccl.getid || !g_rccl.initrank || !g_rccl.destroy || !g_rccl.allgather) {
        sd_set_error("RCCL is missing ncclAllReduce / ncclAllGather / ncclGetUniqueId / ncclCommInitRank / ncclCommDestroy");
        return SD_ERR_HIP;
    }
    g_rccl.lib = h;
    return SD_OK;
}

extern "C" int sd_tp_unique_id(void *id128) {
    SD_REQUIRE(id128, "sd_tp_unique_id: null buffer");
    int rc = rccl_load();
    if (rc != SD_OK) return rc;
    if (g_rccl.getid(id128) != 0) { sd_set_error("ncclGetUniqueId failed"); return SD_ERR_HIP; }
    return SD_OK;
}

extern "C" int sd_tp_create_rccl(int rank, int world, const void *id128, sd_tp **out) {
    SD_REQUIRE(out && id128 && world >= 1 && rank >= 0 && rank < world, "sd_tp_create_rccl: bad arguments");
    int rc = rccl_load();
    if (rc != SD_OK) return rc;
    sd_nccl_id id;
    memcpy(id.b, id128, 128);
    void *comm = nullptr;
    if (g_rccl.initrank(&comm, world, id, rank) != 0) { sd_set_error("ncclCommInitRank(rank %d of %d) failed", rank, world); return SD_ERR_HIP; }
    sd_tp *t = new sd_tp{rank, world, comm, nullptr};
    *out = t;
    return SD_OK;
}

// ---- throughput-mode gather of the sharded streams' token rows: one ncclAllGather over RCCL / xGMI ----
struct sd_comm { int rank, world; void *comm; };

extern "C" int sd_comm_probe(void) { return rccl_load(); }
extern "C" int sd_comm_unique_id(void *id128) { return sd_tp_unique_id(id128); }

extern "C" int sd_comm_init(int rank, int world, const void *id128, sd_comm **out) {
    SD_REQUIRE(out && id128 && world >= 1 && rank >= 0 && rank < world, "sd_comm_init: bad arguments");
    int rc = rccl_load();
    if (rc != SD_OK) return rc;
    sd_nccl_id id;
    memcpy(id.b, id128, 128);
    void *comm = nullptr;
    if (g_rccl.initrank(&comm, world, id, rank) != 0) { sd_set_error("ncclCommInitRank(rank %d of %d) failed", rank, world); return SD_ERR_HIP; }
    *out = new sd_comm{rank, world, comm};
    return SD_OK;
}

extern "C" int sd_comm_all_gather_tokens(sd_comm *c, const int32_t *send, int32_t *recv, int rows, int width, void *stream) {
    SD_REQUIRE(c && send && recv && rows >= 0 && width >= 1, "sd_comm_all_gather_tokens: bad arguments");
    if (rows == 0) return SD_OK;
    if (g_rccl.allgather(send, recv, (size_t)rows * (size_t)width, /* ncclInt32 */ 2, c->comm, (hipStream_t)stream) != 0) {
        sd_set_error("ncclAllGather of %d x %d token rows failed on rank %d of %d", rows, width, c->rank, c->world);
        return SD_ERR_HIP;
    }
    return SD_OK;
}

extern "C" int sd_comm_rank(const sd_comm *c, int *rank_out, int *world_out) {
    SD_REQUIRE(c && rank_out && world_out, "sd_comm_rank: null argument");
    *rank_out = c->rank; *world_out = c->world;
    return SD_OK;
}

extern "C" int sd_comm_destroy(sd_comm *c) {
    if (!c) return SD_OK;
    if (c->comm && g_rccl.destroy) g_rccl.destroy(c->comm);
    delete c;
    return SD_OK;
}

extern "C" int sd_tp_create_loopback(int world, sd_tp **out /* world handles */) {
    SD_REQUIRE(out && world >= 1 && world <= 16, "sd_tp_create_loopback: 1..16 ranks");
    TpLoop *L = new TpLoop();
    L->world = world;
    L->refs = world;
    for (int r = 0; r < world; ++r) {
        SD_HIP_CHECK(hipEventCreateWithFlags(&L->ev_in[r], hipEventDisableTiming));
        SD_HIP_CHECK(hipEventCreateWithFlags(&L->ev_out[r], hipEventDisableTiming));
        out[r] = new sd_tp{r, world, nullptr, L};
    }
    return SD_OK;
}

extern "C" int sd_tp_destroy(sd_tp *t) {
    if (!t) return SD_OK;
    if (t->comm && g_rccl.destroy) g_rccl.destroy(t->comm);
    if (t->loop) {
        bool last;
        { std::lock_guard<std::mutex> lk(t->loop->mu); last = --t->loop->refs == 0; }
        if (last) {
            for (int r = 0; r < t->loop->world; ++r) { (void)hipEventDestroy(t->loop->ev_in[r]); (void)hipEventDestroy(t->loop->ev_out[r]); }
            delete t->loop;
        }
    }
    delete t;
    return SD_OK;
}

extern "C" int sd_session_set_tp(sd_session *s, sd_tp *t) {
    SD_REQUIRE(s, "sd_session_set_tp: null session");
    // a group of one has nothing to reduce; SD_TP_FORCE=1 keeps it anyway (tests: the RCCL plumbing on a one-GPU box)
    const char *force = getenv("SD_TP_FORCE");
    s->tp = (t && (t->world > 1 || (force && atoi(force)))) ? t : nullptr;
    return SD_OK;
}

// tp_in[i] = sum_s part[s][i]  (rows are M x N contiguous at the start of every slab)
__global__ void tp_fold_kernel(const float *__restrict__ part, int S, size_t stride_s, int total, float *__restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    float a = 0.f;
    for (int s2 = 0; s2 < S; ++s2) a += part[(size_t)s2 * stride_s + i];
    out[i] = a;
}
struct TpBufs { const float *p[16]; };
__global__ void tp_sum_kernel(TpBufs b, int world, int total, float *__restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    float a = 0.f;
    for (int r = 0; r < world; ++r) a += b.p[r][i];          // fixed rank order: every rank gets the same bits
    out[i] = a;
}

// After a row-parallel GEMM: fold this rank's k-slabs, all-reduce the [M][N] fp32 partial over the group and point the
// residual epilogue at the sum (one slab).  No-op without a group.
static int tp_reduce(sd_session *s, GemmOut *go, const float **src, int M, int N, hipStream_t st) {
    sd_tp *t = s->tp;
    if (!t) return SD_OK;
    const int total = M * N;
    {
        ProfScope ps(s, PC_OTHER, st);
        // one slab (the row-parallel GEMM kept its whole k-range per workgroup): its [M][N] rows ARE this rank's partial - no fold
        const float *mine = *src;
        if (go->S != 1) {
            hipLaunchKernelGGL(tp_fold_kernel, dim3((total + 255) / 256), dim3(256), 0, st, *src, go->S, go->stride_s, total, s->tp_in);
            SD_LAUNCH_CHECK();
            mine = s->tp_in;
        }
        if (t->comm) {
            if (g_rccl.allreduce(mine, s->tp_out, (size_t)total, /* ncclFloat32 */ 7, /* ncclSum */ 0, t->comm, st) != 0) {
                sd_set_error("ncclAllReduce failed");
                return SD_ERR_HIP;
            }
        } else {
            TpLoop *L = t->loop;
            L->bufs[t->rank] = mine;
            SD_HIP_CHECK(hipEventRecord(L->ev_in[t->rank], st));
            L->barrier();                                           // every rank's partial is enqueued
            TpBufs b = {};
            for (int r = 0; r < t->world; ++r) {
                b.p[r] = L->bufs[r];
                if (r != t->rank) SD_HIP_CHECK(hipStreamWaitEvent(st, L->ev_in[r], 0));
            }
            hipLaunchKernelGGL(tp_sum_kernel, dim3((total + 255) / 256), dim3(256), 0, st, b, t->world, total, s->tp_out);
            SD_LAUNCH_CHECK();
            SD_HIP_CHECK(hipEventRecord(L->ev_out[t->rank], st));
            L->barrier();                                           // nobody overwrites its partial before all have read it
            for (int r = 0; r < t->world; ++r)
                if (r != t->rank) SD_HIP_CHECK(hipStreamWaitEvent(st, L->ev_out[r], 0));
        }
    }
    *src = s->tp_out;
    go->S = 1;
    go->stride_s = 0;
    return SD_OK;
}

// ---- launch helpers ----------------------------------------------------------------------

template <int MT, int EPI, int NTW, typename H = bf16_t>
static void launch_gemm_bf16(const void *W, const void *X, float *part, int M, int Mpad, int N, int K, int S,
                             int ks_per, const GemmEpiT<H> &e, hipStream_t st) {
    const int blocks = (N / 16 / NTW) * S;
    constexpr int UNROLL = NTW >= 8 ? 1 : (MT > 2 ? 2 : 4);        // 4 measured best for decode rows (tools/gemm_bench.py)
    if constexpr (MT == 1 && NTW == 1) {
        // a wave's whole k-range in ONE burst when it is 5..8 k-steps (the draft model's K = 768 GEMMs: 6 per wave), instead
        // of a group of four and a second round trip for the rest
        const int per_wave = (std::min(ks_per, K / 32) + 3) / 4;
        if (per_wave > 4 && per_wave <= 8) {
            hipLaunchKernelGGL((gemm_bf16_stream<1, 8, EPI, 1, true, H>), dim3(blocks), dim3(256), 0, st,
                               (const u32x4 *)W, (const H *)X, part, M, Mpad, N, K, S, ks_per, e);
            return;
        }
    }
    hipLaunchKernelGGL((gemm_bf16_stream<MT, UNROLL, EPI, NTW, true, H>), dim3(blocks), dim3(256), 0, st,
                       (const u32x4 *)W, (const H *)X, part, M, Mpad, N, K, S, ks_per, e);
}

template <int EPI, typename H = bf16_t>
static int dispatch_gemm_bf16(const void *W, const void *X, float *part, int M, int Mpad, int N, int K, int S,
                              int ksp, const GemmEpiT<H> &e, hipStream_t st) {
    const int MT = Mpad / 16;
    // rows beyond one m-tile (prefill, stream-batched verify): every activation fragment a wave loads is reused for
    // NTW weight tiles, because activations and weights share the CU's load path (X:W bytes = 16*MT : 16*NTW)
    const int ntw = gemm_ntw(N, M);
    if (MT == 1) launch_gemm_bf16<1, EPI, 1, H>(W, X, part, M, Mpad, N, K, S, ksp, e, st);
    else if (MT == 2) { if (ntw >= 4) launch_gemm_bf16<2, EPI, 4, H>(W, X, part, M, Mpad, N, K, S, ksp, e, st);
                        else if (ntw == 2) launch_gemm_bf16<2, EPI, 2, H>(W, X, part, M, Mpad, N, K, S, ksp, e, st);
                        else launch_gemm_bf16<2, EPI, 1, H>(W, X, part, M, Mpad, N, K, S, ksp, e, st); }
    else if (MT == 3) { if (ntw == 8) launch_gemm_bf16<3, EPI, 8, H>(W, X, part, M, Mpad, N, K, S, ksp, e, st);
                        else if (ntw == 4) launch_gemm_bf16<3, EPI, 4, H>(W, X, part, M, Mpad, N, K, S, ksp, e, st);
                        else if (ntw == 2) launch_gemm_bf16<3, EPI, 2, H>(W, X, part, M, Mpad, N, K, S, ksp, e, st);
                        else launch_gemm_bf16<3, EPI, 1, H>(W, X, part, M, Mpad, N, K, S, ksp, e, st); }
    else if (MT == 4) { if (ntw == 8) launch_gemm_bf16<4, EPI, 8, H>(W, X, part, M, Mpad, N, K, S, ksp, e, st);
                        else if (ntw == 4) launch_gemm_bf16<4, EPI, 4, H>(W, X, part, M, Mpad, N, K, S, ksp, e, st);
                        else if (ntw == 2) launch_gemm_bf16<4, EPI, 2, H>(W, X, part, M, Mpad, N, K, S, ksp, e, st);
                        else launch_gemm_bf16<4, EPI, 1, H>(W, X, part, M, Mpad, N, K, S, ksp, e, st); }
    else { sd_set_error("gemm: M=%d exceeds 64 rows per call", M); return SD_ERR_INVALID; }
    return SD_OK;
}

// X: [M][K] activations (M <= 64 per call; callers chunk), W: [N][K] weights -> split-K slabs in s->part
template <typename H = bf16_t>
// one_slab: the caller wants the whole k-range per workgroup where the shape allows it (a tensor-parallel shard's O / down
// projection: its [M][N] partial then goes to the all-reduce as it is, without a fold launch in between)
static int run_gemm(sd_session *s, const void *W, const void *X, int M, int N, int K, GemmOut *go, hipStream_t st,
                    const RowTab *xtab = nullptr, bool one_slab = false) {
    const sd_model_config &c = s->m->cfg;
    ProfScope ps(s, PC_GEMM, st);
    if (is16(c.dtype)) {
        bool xmap_identity = true;                                // (the tiled and the balanced kernel read whole activation tiles)
        if (xtab)
            for (int i = 0; i < M && xmap_identity; ++i) xmap_identity = xtab->xmap[i] == i;
        if (xmap_identity) xtab = nullptr;                        // every row in place (a verify pass): no gather
        GemmPlan pl = gemm_plan(N, K, M, xtab == nullptr);                // a row gather (lm_head) keeps the streaming kernel
        // (<= 16 rows of a shard: one workgroup per n-tile over the whole k-range still gives >= 256 workgroups at hidden >= 4096)
        if (one_slab && !pl.tiled && M <= 16 && N / 16 >= 256) { pl.S = 1; pl.ksp = K / 32; }
        const int S = pl.S, ksp = pl.ksp;
        const int Mpad = (int)align_up(M, 16);
        SD_REQUIRE((size_t)S * Mpad * N <= s->part_floats, "run_gemm: partial buffer too small");
        const RowsPlan rp = xmap_identity ? rows_plan(N, K, M, false) : RowsPlan{};
        if (pl.tiled) {
            launch_gemm_tiled<H>(W, X, s->part, M, Mpad, N, K, pl, st);
        } else if (rp.ok) {
            SD_REQUIRE((size_t)rp.S * Mpad * N <= s->part_floats, "run_gemm: partial buffer too small");
            GemmEpiT<H> e = {};
            const int rc = launch_gemm_rows<EPI_PART, H>(W, X, s->part, M, Mpad, N, K, rp, e, st);
            if (rc != SD_OK) return rc;
            go->S = rp.S;
            go->stride_s = (size_t)Mpad * N;
            SD_LAUNCH_CHECK();
            return SD_OK;
        } else {
            GemmEpiT<H> e = {};
            if (xtab) { e.use_xmap = 1; e.tab = *xtab; }
            const int rc = dispatch_gemm_bf16<EPI_PART, H>(W, X, s->part, M, Mpad, N, K, S, ksp, e, st);
            if (rc != SD_OK) return rc;
        }
        go->S = S;
        go->stride_s = (size_t)Mpad * N;
    } else {
        RowTab none = {};
        hipLaunchKernelGGL(gemm_f32_simple, dim3((N + 3) / 4), dim3(256), 0, st, (const float *)W, (const float *)X,
                           s->part, M, N, K, xtab ? *xtab : none, xtab ? 1 : 0);
        go->S = 1;
        go->stride_s = (size_t)M * N;
    }
    SD_LAUNCH_CHECK();
    return SD_OK;
}

// bf16 GEMM whose workgroups keep the whole k-range (SB = 1) and finish with a fused epilogue
template <int EPI, typename H = bf16_t>
static int run_gemm_fused(sd_session *s, const void *W, const void *X, int M, int N, int K, const GemmEpiT<H> &e,
                          hipStream_t st) {
    ProfScope ps(s, PC_GEMM, st);
    const int Mpad = (int)align_up(M, 16);
    // few n-tiles and a long k-range (a tensor-parallel shard's QKV: 80 tiles x K = 8192): sixteen waves per tile
    if constexpr (EPI == EPI_QKV_ROPE || EPI == EPI_QKV_PLAIN) {
        if (g_env.wide_qkv && M <= 16 && !e.use_xmap && !e.x_rowmajor && N / 16 <= 128 && K / 32 >= 128) {
            hipLaunchKernelGGL((gemm_bf16_stream_w16<EPI, H, 16>), dim3(N / 16), dim3(1024), 0, st, (const u32x4 *)W, (const H *)X, M, N, K, e);
            SD_LAUNCH_CHECK();
            return SD_OK;
        }
    }
    // a prefill pass past the balanced kernel's row count: gemm_bf16_mm with the epilogue on its accumulators
    if (!e.use_xmap && !e.x_rowmajor) {
        const GemmPlan pl = gemm_plan(N, K, M, true, true);
        if (pl.tiled && pl.mm && pl.S == 1) {
            launch_gemm_mm<EPI, H>(W, X, nullptr, M, Mpad, N, K, pl, e, st);
            SD_LAUNCH_CHECK();
            return SD_OK;
        }
    }
    // (eight waves per tile for a shard's gate/up - 448 n-tiles - measured slower: 5.94 against 5.86 ms per verify)
    const RowsPlan rp = (e.use_xmap || e.x_rowmajor) ? RowsPlan{} : rows_plan(N, K, M, true);
    const int rc = rp.ok ? launch_gemm_rows<EPI, H>(W, X, nullptr, M, Mpad, N, K, rp, e, st)
                         : dispatch_gemm_bf16<EPI, H>(W, X, nullptr, M, Mpad, N, K, 1, K / 32, e, st);
    if (rc != SD_OK) return rc;
    SD_LAUNCH_CHECK();
    return SD_OK;
}

// Prefill passes (rows = consecutive positions of a stream) of a 16-bit model with head_dim 128: 16-row groups, both products
// on the matrix cores (prefill_attn.h).  Returns false when the pass does not qualify (the caller takes attn_kernel).
#define PA_LDS_MAX (150 * 1024)
static size_t prefill_attn_lds(int s_max) {
    return (size_t)PA_ROWS * ((size_t)align_up(s_max, 64) + PA_SPAD) * sizeof(float) + (size_t)PA_VCH * PA_VST;
}
template <typename T>
static bool prefill_attn_ok(const sd_session *s, const RowTab &tab, int s_max) {
    if constexpr (sizeof(T) != 2) return false;
    const sd_model_config &c = s->m->cfg;
    return g_env.prefill_attn && tab.contig && !tab.tree && !tab.kv_fp8 && c.head_dim == 128 && tab.n_rows >= 32 &&
           c.n_heads % c.n_kv_heads == 0 && prefill_attn_lds(s_max) <= PA_LDS_MAX;
}
template <typename T>
static int launch_attn_prefill(sd_session *s, const T *q, const RowTab &tab, int layer, T *out, int s_max, hipStream_t st) {
    const sd_model_config &c = s->m->cfg;
    // the table's groups are <= ATT_TQ consecutive rows of a stream: two neighbours of one stream make a 16-row group
    PaGroups pg = {};
    for (int g = 0; g < tab.n_groups; ++g) {
        const int i = pg.n;
        const bool join = i > 0 && pg.nrows[i - 1] == ATT_TQ && tab.grp_stream[g] == tab.grp_stream[g - 1] &&
                          tab.grp_row0[g] == pg.row0[i - 1] + ATT_TQ && tab.grp_pos[g] == pg.pos[i - 1] + ATT_TQ;
        if (join) { pg.nrows[i - 1] += tab.grp_n[g]; continue; }
        pg.row0[i] = tab.grp_row0[g]; pg.nrows[i] = tab.grp_n[g]; pg.pos[i] = tab.grp_pos[g];
        pg.max_seq[i] = tab.max_seq[tab.grp_stream[g]]; pg.kv[i] = tab.kv_base[tab.grp_stream[g]];
        ++pg.n;
    }
    const int s_cap = (int)align_up(s_max, 64);
    const size_t lds = prefill_attn_lds(s_max);
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(attn_prefill_kernel<T>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  PA_LDS_MAX);
        attr = true;
    }
    hipLaunchKernelGGL((attn_prefill_kernel<T>), dim3(c.n_heads, pg.n), dim3(256), lds, st, q, pg, layer, out, c.n_heads, c.n_kv_heads,
                       c.arch, 1.0f / sqrtf((float)c.head_dim), s_cap);
    return SD_OK;
}

// keys per workgroup above which a group's keys are cut over several workgroups (and merged by attn_combine_kernel)
#define ATT_SPLIT_KEYS 384
#define ATT_MAX_PARTS 64          // groups * splits the partial buffer is sized for

template <typename T, int D>
static int launch_attn(sd_session *s, const T *q, const RowTab &tab, int layer, T *out, int s_max, hipStream_t st) {
    const sd_model_config &c = s->m->cfg;
    // 160 KiB per workgroup less the kernel's static LDS (the split path's per-row max / denominator)
    const size_t lds_max = 160 * 1024 - 512;
    auto lds_for = [&](int nsplit, int *s_cap) {
        const int keys = nsplit > 1 ? (((s_max + nsplit - 1) / nsplit + 15) & ~15) : s_max;
        *s_cap = (int)align_up(keys, 64);
        return sizeof(float) * ((size_t)ATT_TQ * D + (size_t)(256 / (D / 8)) * ATT_TQ * D + (size_t)ATT_TQ * *s_cap);
    };
    const int split_keys = g_env.attn_split_keys, keys_per = g_env.attn_keys_per_split;
    int nsplit = 1, s_cap;
    if (s_max > split_keys) {
        nsplit = std::min(8, (s_max + keys_per - 1) / keys_per);
        while (nsplit > 1 && nsplit * tab.n_groups > ATT_MAX_PARTS) --nsplit;
    }
    // very long contexts: more, smaller chunks until one chunk's score rows fit the LDS
    while (lds_for(nsplit, &s_cap) > lds_max && (nsplit + 1) * tab.n_groups <= ATT_MAX_PARTS) ++nsplit;
    const size_t lds = lds_for(nsplit, &s_cap);
    if (lds > lds_max) {
        sd_set_error("attention: %d keys x %d row groups exceed the LDS score tile", s_max, tab.n_groups);
        return SD_ERR_CAPACITY;
    }
    auto launch = [&](auto kv8, auto tree) -> int {
        constexpr bool KV8 = decltype(kv8)::value, TREE = decltype(tree)::value;
        static bool attr = false;                                 // (one flag per instantiation of this lambda)
        if (!attr) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(attn_kernel<T, D, KV8, TREE>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_max);
            attr = true;
        }
        hipLaunchKernelGGL((attn_kernel<T, D, KV8, TREE>), dim3(c.n_heads, tab.n_groups, nsplit), dim3(256), lds, st, q, tab, layer,
                           out, c.n_heads, c.n_kv_heads, c.arch, 1.0f / sqrtf((float)c.head_dim), s_cap, nsplit, s->attn_part);
        return SD_OK;
    };
    using std::true_type;
    using std::false_type;
    if (tab.kv_fp8) {
        if constexpr (sizeof(T) == 2 && D >= 32) {
            if (tab.tree) launch(true_type{}, true_type{}); else launch(true_type{}, false_type{});
        } else { sd_set_error("fp8 KV needs a 16-bit model with head_dim >= 32"); return SD_ERR_INVALID; }
    } else {
        if (tab.tree) launch(false_type{}, true_type{}); else launch(false_type{}, false_type{});
    }
    if (nsplit > 1)
        hipLaunchKernelGGL((attn_combine_kernel<T, D>), dim3(c.n_heads, tab.n_groups), dim3(128), 0, st,
                           (const float *)s->attn_part, tab, out, c.n_heads, nsplit);
    return SD_OK;
}


// Attention + O projection in one launch (fused_kernels.h) for <= 8 rows of one stream on a 16-bit model with
// head_dim 128 and K = n_heads * 128 <= 5120: returns false when the shape does not qualify (the caller then takes the two
// launches).  On success the O projection's single slab is in s->part.
#define AO_LDS_MAX (80 * 1024)
static size_t attn_oproj_lds(int s_max) {
    constexpr int D = 128;
    const int s_cap = (int)align_up(s_max, 64);
    return sizeof(float) * ((size_t)ATT_TQ * D + (size_t)(256 / (D / 8)) * ATT_TQ * D + (size_t)ATT_TQ * s_cap);
}
template <typename T>
static bool attn_oproj_ok(const sd_session *s, const RowTab &tab, int s_max) {
    const sd_model_config &c = s->m->cfg;
    if constexpr (sizeof(T) != 2) return false;
    if (!g_env.fuse_attn_o || s->tp || tab.tree || tab.kv_fp8 || tab.contig) return false;
    if (c.head_dim != 128 || tab.n_groups > 2 || tab.n_streams != 1 || tab.n_rows > 16) return false;
    const int K = q_dim(c), N = c.hidden;
    if (K % 128 || K / 128 > AO_NKW || N % 16 || s_max > g_env.attn_split_keys) return false;
    const int cus = g_env.cus > 0 ? g_env.cus : 256;
    if (c.n_heads * tab.n_groups + (N / 16 + 1) / 2 > cus) return false;      // every workgroup resident at once, one per CU
    if ((size_t)16 * N > s->part_floats) return false;
    // (a context whose score tile does not fit the launch's LDS budget takes the two launches - never an error: the key
    //  limit SD_ATTN_SPLIT_KEYS is a tuning knob)
    if (attn_oproj_lds(s_max) > AO_LDS_MAX) return false;
    return true;
}
template <typename T>
static int launch_attn_oproj(sd_session *s, const T *q, const RowTab &tab, int layer, T *out, int s_max, const void *wo,
                             bool resid, hipStream_t st) {
    const sd_model_config &c = s->m->cfg;
    const int s_cap = (int)align_up(s_max, 64);
    const size_t lds = attn_oproj_lds(s_max);
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(attn_oproj_kernel<T>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  AO_LDS_MAX);
        attr = true;
    }
    SD_REQUIRE(lds <= AO_LDS_MAX, "attn_oproj: %d keys exceed the launch's LDS budget", s_max);      // (attn_oproj_ok checked)
    // arrivals counted so far, this launch's included; the epoch advances only once the launch is known to be enqueued (a
    // failed launch must not leave every later wait of the session one arrival short)
    const unsigned n_arrive = (unsigned)(c.n_heads * tab.n_groups);
    const unsigned want = s->ao_epoch + n_arrive + (unsigned)s->skew_now;
    hipLaunchKernelGGL((attn_oproj_kernel<T>), dim3(c.n_heads * tab.n_groups + (c.hidden / 16 + 1) / 2), dim3(512), lds, st, q, tab, layer, out, c.n_heads,
                       c.n_kv_heads, c.arch, 1.0f / sqrtf((float)c.head_dim), s_cap, (const u32x4 *)wo, s->part, tab.n_rows, c.hidden,
                       q_dim(c), s->ao_ctr, want, g_env.ao_delay, g_env.ao_gap,
                       g_env.ao_stamps ? s->ao_stamps : (long long *)nullptr,
                       (T *)s->x, (T *)s->h, resid ? s->ssq : (float *)nullptr, s->wait_status);
    SD_LAUNCH_CHECK();
    s->ao_epoch += n_arrive;
    return SD_OK;
}

// Norm on load (normload_kernels.h): <= 8 rows of a 16-bit Llama model whose residual rows were left un-normalised (+
// per-tile sums of squares in s->ssq) by a residual epilogue; the consumer GEMM keeps its whole k-range per workgroup.
template <typename T>
static bool norm_on_load_ok(const sd_session *s, const RowTab &tab) {
    const sd_model_config &c = s->m->cfg;
    if constexpr (sizeof(T) != 2) return false;
    if (!g_env.norm_on_load || c.arch != SD_ARCH_LLAMA || !c.fused_layout || s->tp) return false;
    // (9..16 rows: the conversions cost more than the launch they replace - +5.7 us on gate/up at 12 rows, tools/gemm_bench.py)
    return tab.n_rows <= 8 && c.hidden % 1024 == 0 && c.hidden <= 8192;
}
template <int EPI, typename H>
static int launch_gemm_xn(sd_session *s, const void *W, const void *X, int M, int N, int K, const void *norm_w, float eps,
                          GemmEpiT<H> e, hipStream_t st) {
    ProfScope ps(s, PC_GEMM, st);
    SD_REQUIRE(M <= 8 && N % 16 == 0 && K % 1024 == 0 && K <= 8192, "norm-on-load GEMM: M=%d N=%d K=%d", M, N, K);
    e.nrm_ssq = s->ssq; e.nrm_w = (const H *)norm_w; e.nrm_nt = K / 16; e.nrm_eps = eps;
    // k-steps whose M valid rows fit one conversion register (C * 4 * M <= 64 lanes)
    auto go = [&](auto c) {
        hipLaunchKernelGGL((gemm_bf16_stream_xn<EPI, decltype(c)::value, H>), dim3(N / 16), dim3(256), (size_t)K * sizeof(H), st,
                           (const u32x4 *)W, (const H *)X, (float *)nullptr, M, N, K, 1, K / 32, e);
    };
    using std::integral_constant;
    if (M <= 4) go(integral_constant<int, 4>{});
    else if (M == 5) go(integral_constant<int, 3>{});
    else go(integral_constant<int, 2>{});
    SD_LAUNCH_CHECK();
    return SD_OK;
}

// The k-split GEMM (streaming kernel's split) finishing with the residual epilogue: rows -> s->x, s->h (un-normalised), s->ssq
template <typename H>
static int launch_gemm_fin(sd_session *s, const void *W, const void *X, int M, int N, int K, const void *bias, hipStream_t st) {
    ProfScope ps(s, PC_GEMM, st);
    const GemmPlan pl = gemm_plan(N, K, M);
    SD_REQUIRE(!pl.tiled && M <= 16 && (size_t)pl.S * 16 * N <= s->part_floats, "k-split GEMM with residual epilogue: M=%d N=%d K=%d", M, N, K);
    GemmEpiT<H> e = {};
    e.bias = (const H *)bias; e.res_x = (H *)s->x; e.res_h = (H *)s->h; e.res_ssq = s->ssq;
    const unsigned want = s->fin_epoch + (unsigned)(pl.S - 1) + (unsigned)s->skew_now;
    hipLaunchKernelGGL((gemm_bf16_stream_fin<H>), dim3((N / 16) * pl.S), dim3(256), 0, st, (const u32x4 *)W, (const H *)X, s->part,
                       M, N, K, pl.S, pl.ksp, e, s->fin_ctr, want, s->wait_status);
    SD_LAUNCH_CHECK();
    s->fin_epoch += (unsigned)(pl.S - 1);                         // (advanced only for a launch that was enqueued)
    return SD_OK;
}

// The lm_head over the (normalised, operand-layout) rows `hl`; xt (or NULL) maps logit row i to its row of hl.  Native
// iteration: the head also leaves the maximum of every 16-column tile and clears the probability rows, so the
// normalisation that follows needs no candidate pass over V (EPI_HEAD; whole k-range per workgroup).
template <typename T>
static int head_logits(sd_session *s, const T *hl, const RowTab *xt, int n_logits, float *logits_out, long ld_logits, hipStream_t st) {
    using H16 = typename std::conditional<std::is_same<T, float>::value, bf16_t, T>::type;
    sd_model *m = s->m;
    const sd_model_config &c = m->cfg;
    const int ED = embed_dim(c);
    const bool llama = c.arch == SD_ARCH_LLAMA;
    GemmOut go;
    int rc;
    const int round_t = (c.logits_bf16_round || !llama) ? round_code(c.dtype) : 0;
    const bool zero_tab = !s->head_zero_rows && s->head_zero_n > 0 && s->head_zero_n == n_logits;
    if (is16(c.dtype) && s->want_raw_logits && (s->head_zero_rows || zero_tab) && n_logits <= 16 && c.vocab % 16 == 0 &&
        gemm_plan(c.vocab, ED, n_logits, false).S == 1 && !gemm_plan(c.vocab, ED, n_logits, false).tiled) {
        GemmEpiT<H16> e = {};
        if (xt) { e.use_xmap = 1; e.tab = *xt; }
        e.tile_max = s->tile_max; e.zero_rows = s->head_zero_rows; e.zero_ld = s->head_zero_ld;
        if (zero_tab)
            for (int i = 0; i < n_logits; ++i) e.zero_ptr[i] = s->head_zero_ptr[i];
        {
            ProfScope ps(s, PC_GEMM, st);
            launch_gemm_bf16<1, EPI_HEAD, 1, H16>(m->w.lm_head, (const H16 *)hl, s->part, n_logits, 16, c.vocab, ED, 1, ED / 32, e, st);
            SD_LAUNCH_CHECK();
        }
        s->last_logits = s->part; s->last_logits_ld = c.vocab; s->last_logits_round = round_t;
        s->last_tile_max = s->tile_max;
        return SD_OK;
    }
    if ((rc = run_gemm<H16>(s, m->w.lm_head, hl, n_logits, c.vocab, ED, &go, st, xt)) != SD_OK) return rc;
    if (s->want_raw_logits && go.S == 1) {
        s->last_logits = s->part; s->last_logits_ld = c.vocab; s->last_logits_round = round_t;
    } else {
        ProfScope ps(s, PC_LOGITS, st);
        hipLaunchKernelGGL((logits_kernel<T>), dim3((c.vocab + 255) / 256, n_logits), dim3(256), 0, st, s->part, go.S,
                           go.stride_s, c.vocab, round_t, logits_out, ld_logits);
        SD_LAUNCH_CHECK();
        s->last_logits = logits_out; s->last_logits_ld = ld_logits; s->last_logits_round = 0;
    }
    return SD_OK;
}

// ---- small-model decode path (small_kernels.h): 5 launches per layer + the head --------------------------------
static bool small_path_ok(const sd_session *s, const RowTab &tab) {
    const sd_model_config &c = s->m->cfg;
    const int enabled = g_env.small_path;
    if (!enabled || c.dtype != SD_BF16 || !c.fused_layout || tab.contig || s->tp) return false;
    if (tab.n_rows > SMALL_MAX_ROWS || tab.n_logit_rows > SMALL_MAX_ROWS) return false;
    if (c.hidden > 2048 || c.hidden % 32 != 0 || embed_dim(c) != c.hidden) return false;
    if (c.arch == SD_ARCH_OPT && !c.opt_pre_ln) return false;          // post-LN keeps the stand-alone norm launches
    if (!s->m->w.final_norm_w) return false;
    return true;
}

// k-slabs of the small path's O / down GEMMs.  Default: the streaming GEMM's own policy (bit-identical slabs, what the
// parity tests compare against); SD_SMALL_SPLIT_BYTES = b cuts them so that a workgroup streams about b bytes.
static void small_split(int N, int K, int *S_out, int *ksp_out) {
    const int bytes = g_env.small_split_bytes;
    const int KS = K / 32;
    if (bytes <= 0) { gemm_split(N, K, 1, S_out, ksp_out); }
    else {
        int S = std::max(1, std::min(16, (int)((K * 32 + bytes / 2) / bytes)));
        int ksp = (KS + S - 1) / S;
        *S_out = (KS + ksp - 1) / ksp;
        *ksp_out = ksp;
    }
    if (*S_out > 16) { *ksp_out = (KS + 15) / 16; *S_out = (KS + *ksp_out - 1) / *ksp_out; }
}

template <int PRO, int EPI>
static int launch_small(sd_session *s, const void *W, const void *X, float *part, int M, int N, int K, int S, int ksp,
                        const GemmEpi &e, const SmallPro &p, hipStream_t st) {
    ProfScope ps(s, PC_GEMM, st);
    const size_t lds = PRO == PRO_TILED ? 0 : (size_t)M * (K + SMALL_XPAD) * sizeof(bf16_t);
    hipLaunchKernelGGL((gemm_small<PRO, EPI>), dim3((N / 16) * S), dim3(256), lds, st, (const u32x4 *)W, (const bf16_t *)X,
                       part, M, N, K, S, ksp, e, p);
    SD_LAUNCH_CHECK();
    return SD_OK;
}

static int launch_attn_bf16(sd_session *s, const bf16_t *q, const RowTab &tab, int layer, bf16_t *out, int s_max,
                            hipStream_t st);

static int forward_small(sd_session *s, const RowTab &tab, int s_max, float *logits_out, long ld_logits, hipStream_t st) {
    sd_model *m = s->m;
    const sd_model_config &c = m->cfg;
    const int H = c.hidden, D = c.head_dim, I = c.inter, L = c.n_layers, M = tab.n_rows, n_logits = tab.n_logit_rows;
    const bool llama = c.arch == SD_ARCH_LLAMA;
    const int norm_kind = llama ? NORM_RMS : NORM_LN;
    const int rn_threads = (int)std::min<size_t>(1024, std::max<size_t>(64, align_up((H / 4 + RN_RG - 1) / RN_RG, 64)));
    bf16_t *R[2] = {(bf16_t *)s->x, (bf16_t *)s->x2};
    bf16_t *qb = (bf16_t *)s->qbuf, *at = (bf16_t *)s->attn, *ac = (bf16_t *)s->act;
    int cur = 0;                                                   // R[cur] holds the residual stream
    int rc;
    // SD_SMALL_PATH=1: the two norm launches of a layer become prologues of QKV / gate-up; O, down and the head keep the
    // per-op chain's kernels.  =2: every seam a prologue and gemm_small for O / down (the round-2 route, kept for A/B)
    const bool all_pro = g_env.small_path >= 2;
    int S_o, ksp_o, S_d, ksp_d;
    small_split(H, q_dim(c), &S_o, &ksp_o);
    small_split(H, I, &S_d, &ksp_d);
    SD_REQUIRE((size_t)std::max(S_o, S_d) * 16 * H <= s->spart_floats, "forward_small: slab buffer too small");

    auto resid_pro = [&](int S, const bf16_t *bias, const void *nw, const void *nb, bool write_r) {
        SmallPro p = {};
        p.slab = s->spart; p.S = S; p.stride_s = (size_t)16 * H; p.bias = bias;
        p.r_in = R[cur]; p.r_out = write_r ? R[cur ^ 1] : nullptr;
        p.nw = (const bf16_t *)nw; p.nb = (const bf16_t *)nb; p.eps = c.norm_eps; p.kind = norm_kind; p.H = H;
        p.rn_threads = rn_threads;
        return p;
    };

    for (int l = 0; l < L; ++l) {
        // ---- QKV: prologue = embedding (layer 0) or previous layer's down slabs + residual, then input norm
        {
            GemmEpi e = {};
            e.out = qb; e.bias = (const bf16_t *)m->bqkv[l];
            e.cos_t = (const bf16_t *)m->w.rope_cos; e.sin_t = (const bf16_t *)m->w.rope_sin;
            e.Hq = c.n_heads; e.Hkv = c.n_kv_heads; e.D = D; e.layer = l; e.tab = tab;
            e.q_scale = 1.0f / sqrtf((float)D);
            if (l == 0) {
                SmallPro p = {};
                p.embed = (const bf16_t *)m->w.embed; p.pos_embed = llama ? nullptr : (const bf16_t *)m->w.pos_embed;
                p.pos_off = 2; p.vocab = c.vocab; p.r_out = R[cur];
                p.nw = (const bf16_t *)m->n1w[0]; p.nb = (const bf16_t *)m->n1b[0]; p.eps = c.norm_eps; p.kind = norm_kind;
                p.H = H; p.rn_threads = rn_threads;
                rc = llama ? launch_small<PRO_EMBED, EPI_QKV_ROPE>(s, m->wqkv[l], nullptr, nullptr, M, qkv_cols(c), H, 1, H / 32, e, p, st)
                           : launch_small<PRO_EMBED, EPI_QKV_PLAIN>(s, m->wqkv[l], nullptr, nullptr, M, qkv_cols(c), H, 1, H / 32, e, p, st);
            } else {
                const SmallPro p = resid_pro(S_d, (const bf16_t *)m->bfc2[l - 1], m->n1w[l], m->n1b[l], true);
                rc = llama ? launch_small<PRO_RESID, EPI_QKV_ROPE>(s, m->wqkv[l], nullptr, nullptr, M, qkv_cols(c), H, 1, H / 32, e, p, st)
                           : launch_small<PRO_RESID, EPI_QKV_PLAIN>(s, m->wqkv[l], nullptr, nullptr, M, qkv_cols(c), H, 1, H / 32, e, p, st);
                cur ^= 1;
            }
            if (rc != SD_OK) return rc;
        }
        // ---- attention over the arena
        {
            ProfScope ps(s, PC_ATTN, st);
            if ((rc = launch_attn_bf16(s, qb, tab, l, at, s_max, st)) != SD_OK) return rc;
            SD_LAUNCH_CHECK();
        }
        // ---- O projection -> slabs
        {
            GemmEpi e = {};
            SmallPro p = {};
            // (the streaming kernel leaves the same slabs: [S][16][H], the same k-split, the same sums - and runs 0.1-0.9 us
            //  faster than gemm_small<PRO_TILED> on these shapes)
            if (all_pro) rc = launch_small<PRO_TILED, EPI_PART>(s, m->wo[l], at, s->spart, M, H, q_dim(c), S_o, ksp_o, e, p, st);
            else { ProfScope ps(s, PC_GEMM, st); rc = dispatch_gemm_bf16<EPI_PART, bf16_t>(m->wo[l], at, s->spart, M, 16, H, q_dim(c), S_o, ksp_o, e, st); SD_LAUNCH_CHECK(); }
            if (rc != SD_OK) return rc;
        }
        // ---- gate/up (fc1): prologue = O slabs + residual + post-attention norm; epilogue = SiLU * up / ReLU
        {
            GemmEpi e = {};
            e.out = ac; e.bias = (const bf16_t *)m->bfc1[l]; e.n_out = I;
            const SmallPro p = resid_pro(S_o, (const bf16_t *)m->bo[l], m->n2w[l], m->n2b[l], true);
            rc = llama ? launch_small<PRO_RESID, EPI_ACT_SILU>(s, m->wgu[l], nullptr, nullptr, M, gu_cols(c), H, 1, H / 32, e, p, st)
                       : launch_small<PRO_RESID, EPI_ACT_RELU>(s, m->wgu[l], nullptr, nullptr, M, gu_cols(c), H, 1, H / 32, e, p, st);
            if (rc != SD_OK) return rc;
            cur ^= 1;
        }
        // ---- down projection (fc2) -> slabs
        {
            GemmEpi e = {};
            SmallPro p = {};
            if (all_pro) rc = launch_small<PRO_TILED, EPI_PART>(s, m->wdown[l], ac, s->spart, M, H, I, S_d, ksp_d, e, p, st);
            else { ProfScope ps(s, PC_GEMM, st); rc = dispatch_gemm_bf16<EPI_PART, bf16_t>(m->wdown[l], ac, s->spart, M, 16, H, I, S_d, ksp_d, e, st); SD_LAUNCH_CHECK(); }
            if (rc != SD_OK) return rc;
        }
    }
    s->last_tile_max = nullptr;
    if (n_logits > 0 && !all_pro) {
        // ---- head: the final residual + norm keeps its launch.  As a prologue of the lm_head every one of its 2000-3142
        // workgroups folds the slabs and normalises the rows for itself: 18.5 us against 4.4 + 9.6 on llama-68m, 31.4 against
        // 4.3 + 14.2 on opt-125m (rocprofv3, profiles/r04_small_path_*.txt)
        {
            ProfScope ps(s, PC_NORM, st);
            hipLaunchKernelGGL((residual_norm_kernel<bf16_t>), dim3(M), dim3(rn_threads), 0, st, R[cur], (const float *)s->spart, S_d,
                               (size_t)16 * H, H, (const bf16_t *)m->bfc2[L - 1], (const bf16_t *)m->w.final_norm_w,
                               (const bf16_t *)m->w.final_norm_b, c.norm_eps, norm_kind, (int)RES_PRE, (bf16_t *)s->h);
            SD_LAUNCH_CHECK();
        }
        return head_logits<bf16_t>(s, (const bf16_t *)s->h, &tab, n_logits, logits_out, ld_logits, st);
    }
    if (n_logits > 0) {
        // ---- head (SD_SMALL_PATH=2): prologue = last down slabs + residual + final norm on the rows that need logits
        GemmEpi e = {};
        e.use_xmap = 1; e.tab = tab;
        SmallPro p = resid_pro(S_d, (const bf16_t *)m->bfc2[L - 1], m->w.final_norm_w, m->w.final_norm_b, false);
        const int round_t = (c.logits_bf16_round || !llama) ? round_code(c.dtype) : 0;
        SD_REQUIRE((size_t)16 * c.vocab <= s->part_floats, "forward_small: logits slab too small");
        if (s->want_raw_logits && s->head_zero_rows && c.vocab % 16 == 0) {
            e.tile_max = s->tile_max; e.zero_rows = s->head_zero_rows; e.zero_ld = s->head_zero_ld;
            if ((rc = launch_small<PRO_RESID, EPI_HEAD>(s, m->w.lm_head, nullptr, s->part, n_logits, c.vocab, H, 1, H / 32, e, p, st)) != SD_OK) return rc;
            s->last_tile_max = s->tile_max;
        } else {
            if ((rc = launch_small<PRO_RESID, EPI_PART>(s, m->w.lm_head, nullptr, s->part, n_logits, c.vocab, H, 1, H / 32, e, p, st)) != SD_OK) return rc;
        }
        if (s->want_raw_logits) {
            s->last_logits = s->part; s->last_logits_ld = c.vocab; s->last_logits_round = round_t;
        } else {
            ProfScope ps(s, PC_LOGITS, st);
            hipLaunchKernelGGL((logits_kernel<bf16_t>), dim3((c.vocab + 255) / 256, n_logits), dim3(256), 0, st, s->part, 1,
                               (size_t)16 * c.vocab, c.vocab, round_t, logits_out, ld_logits);
            SD_LAUNCH_CHECK();
            s->last_logits = logits_out; s->last_logits_ld = ld_logits; s->last_logits_round = 0;
        }
    }
    return SD_OK;
}


template <typename T>
static int forward_impl(sd_session *s, const RowTab &tab, int s_max, float *logits_out, long ld_logits,
                        hipStream_t st) {
    // 16-bit storage type of the MFMA GEMMs (the fp32 instantiation never launches them; it only has to compile)
    using H16 = typename std::conditional<std::is_same<T, float>::value, bf16_t, T>::type;
    const int n_new = tab.n_rows, n_logits = tab.n_logit_rows;
    sd_model *m = s->m;
    const sd_model_config &c = m->cfg;
    if constexpr (std::is_same<T, bf16_t>::value) {
        if (small_path_ok(s, tab)) return forward_small(s, tab, s_max, logits_out, ld_logits, st);
    }
    s->last_tile_max = nullptr;
    const int H = c.hidden, D = c.head_dim, I = c.inter, L = c.n_layers, ED = embed_dim(c);
    const bool llama = c.arch == SD_ARCH_LLAMA;
    const int norm_kind = llama ? NORM_RMS : NORM_LN;
    const bool pre = llama || c.opt_pre_ln;
    const bool fused = is16(c.dtype) && c.fused_layout != 0;
    T *x = (T *)s->x, *h = (T *)s->h, *qb = (T *)s->qbuf, *at = (T *)s->attn, *ac = (T *)s->act, *eb = (T *)s->ebuf;
    const size_t norm_lds = (size_t)(H + 32) * sizeof(float);
    const int pos_off = 2;                                                   // OPT offset (modeling_opt.py:104)
    // residual+norm keeps the row in registers: 2 groups of 4 columns per thread (H <= 8192)
    const int rn_threads = (int)std::min<size_t>(1024, std::max<size_t>(64, align_up((H / 4 + RN_RG - 1) / RN_RG, 64)));
    GemmOut go;
    int rc;

    // ---- 1..4 new rows of a small model (the draft's decode step): embedding + first pre-norm are the PROLOGUE of the
    // layer-0 QKV GEMM (every workgroup normalises the <= 4 rows itself while its weight tiles are in flight): one
    // launch instead of two, 6.7 us against 5.0 + 5.0 on llama-68m (tools/draft_step_bench.py)
    bool qkv0_done = false;
    if constexpr (std::is_same<T, bf16_t>::value) {
        if (g_env.fuse_embed_qkv && fused && pre && (llama || ED == H) && !tab.contig && n_new <= SMALL_MAX_ROWS && H <= 2048 &&
            H % 32 == 0 && !gemm_plan(qkv_cols(c), H, n_new).tiled) {
            GemmEpi e = {};
            e.out = (bf16_t *)qb; e.bias = (const bf16_t *)m->bqkv[0];
            e.cos_t = (const bf16_t *)m->w.rope_cos; e.sin_t = (const bf16_t *)m->w.rope_sin;
            e.Hq = c.n_heads; e.Hkv = c.n_kv_heads; e.D = D; e.layer = 0; e.tab = tab;
            e.q_scale = 1.0f / sqrtf((float)D);
            SmallPro p = {};
            p.embed = (const bf16_t *)m->w.embed; p.pos_embed = llama ? nullptr : (const bf16_t *)m->w.pos_embed;
            p.pos_off = pos_off; p.vocab = c.vocab; p.r_out = (bf16_t *)x;
            p.nw = (const bf16_t *)m->n1w[0]; p.nb = (const bf16_t *)m->n1b[0]; p.eps = c.norm_eps; p.kind = norm_kind; p.H = H;
            rc = llama ? launch_small<PRO_EMBED, EPI_QKV_ROPE>(s, m->wqkv[0], nullptr, nullptr, n_new, qkv_cols(c), H, 1, H / 32, e, p, st)
                       : launch_small<PRO_EMBED, EPI_QKV_PLAIN>(s, m->wqkv[0], nullptr, nullptr, n_new, qkv_cols(c), H, 1, H / 32, e, p, st);
            if (rc != SD_OK) return rc;
            qkv0_done = true;
        }
    }
    // ---- embeddings (+ the first pre-norm in the same launch when there is no input projection)
    if (qkv0_done) {
    } else if ((llama || ED == H) && pre) {
        ProfScope ps(s, PC_EMBED, st);
        hipLaunchKernelGGL((embed_norm_kernel<T>), dim3(n_new), dim3(256), norm_lds, st, tab, (const T *)m->w.embed, H,
                           llama ? (const T *)nullptr : (const T *)m->w.pos_embed, pos_off, x, (const T *)m->n1w[0],
                           (const T *)m->n1b[0], c.norm_eps, norm_kind, h, c.vocab);
        SD_LAUNCH_CHECK();
    } else {
        if (llama || ED == H) {
            ProfScope ps(s, PC_EMBED, st);
            hipLaunchKernelGGL((embed_kernel<T>), dim3(n_new), dim3(256), 0, st, tab, (const T *)m->w.embed, H,
                               llama ? (const T *)nullptr : (const T *)m->w.pos_embed, pos_off, x, 0, c.vocab);
            SD_LAUNCH_CHECK();
        } else {
            {
                ProfScope ps(s, PC_EMBED, st);
                hipLaunchKernelGGL((embed_kernel<T>), dim3(n_new), dim3(256), 0, st, tab, (const T *)m->w.embed, ED,
                                   (const T *)nullptr, 0, eb, 1, c.vocab);
                SD_LAUNCH_CHECK();
            }
            if ((rc = run_gemm<H16>(s, m->w.project_in, eb, n_new, H, ED, &go, st)) != SD_OK) return rc;
            ProfScope ps(s, PC_EMBED, st);
            hipLaunchKernelGGL((reduce_addpos_kernel<T>), dim3(n_new), dim3(256), 0, st, s->part, go.S, go.stride_s, H,
                               (const T *)m->w.pos_embed, tab, pos_off, x);
            SD_LAUNCH_CHECK();
        }
        // ---- first pre-norm
        if (pre) {
            ProfScope ps(s, PC_NORM, st);
            hipLaunchKernelGGL((norm_kernel<T>), dim3(n_new), dim3(256), norm_lds, st, x, H, (const T *)m->n1w[0],
                               (const T *)m->n1b[0], c.norm_eps, norm_kind, h);
            SD_LAUNCH_CHECK();
        } else {
            ProfScope ps(s, PC_NORM, st);
            hipLaunchKernelGGL((to_operand_kernel<T>), dim3(n_new), dim3(256), 0, st, (const T *)x, H, h);
            SD_LAUNCH_CHECK();
        }
    }

    bool xn_d = false;                                                // layer l - 1's down projection left un-normalised rows + partials
    for (int l = 0; l < L; ++l) {
        // qkv projection -> rope / scale -> q buffer + in-place KV append (fused into the GEMM's epilogue unless the
        // row count takes the tiled kernel, which leaves slabs for the stand-alone epilogue)
        if (l == 0 && qkv0_done) {
        } else if (xn_d) {
            if constexpr (!std::is_same<T, float>::value) {
                GemmEpiT<H16> e = {};
                e.out = (H16 *)qb; e.bias = (const H16 *)m->bqkv[l];
                e.cos_t = (const H16 *)m->w.rope_cos; e.sin_t = (const H16 *)m->w.rope_sin;
                e.Hq = c.n_heads; e.Hkv = c.n_kv_heads; e.D = D; e.layer = l; e.tab = tab;
                e.q_scale = 1.0f / sqrtf((float)D);
                if ((rc = launch_gemm_xn<EPI_QKV_ROPE, H16>(s, m->wqkv[l], h, n_new, qkv_cols(c), H, m->n1w[l], c.norm_eps, e, st)) != SD_OK)
                    return rc;
            }
        } else if (fused && fused_plan_ok(gemm_plan(qkv_cols(c), H, n_new, true, true))) {
            GemmEpiT<H16> e = {};
            e.out = (H16 *)qb; e.bias = (const H16 *)m->bqkv[l];
            e.cos_t = (const H16 *)m->w.rope_cos; e.sin_t = (const H16 *)m->w.rope_sin;
            e.Hq = c.n_heads; e.Hkv = c.n_kv_heads; e.D = D; e.layer = l; e.tab = tab;
            e.q_scale = 1.0f / sqrtf((float)D);
            rc = llama ? run_gemm_fused<EPI_QKV_ROPE, H16>(s, m->wqkv[l], h, n_new, qkv_cols(c), H, e, st)
                       : run_gemm_fused<EPI_QKV_PLAIN, H16>(s, m->wqkv[l], h, n_new, qkv_cols(c), H, e, st);
            if (rc != SD_OK) return rc;
        } else {
            if ((rc = run_gemm<H16>(s, m->wqkv[l], h, n_new, qkv_cols(c), H, &go, st)) != SD_OK) return rc;
            ProfScope ps(s, PC_QKV, st);
            hipLaunchKernelGGL((qkv_epilogue_kernel<T>), dim3(n_new, c.n_heads + 2 * c.n_kv_heads),
                               dim3(std::max(D / 2, 64)), 0, st, s->part, go.S, go.stride_s, qkv_cols(c),
                               (const T *)m->bqkv[l], (const T *)m->w.rope_cos, (const T *)m->w.rope_sin, c.arch,
                               1.0f / sqrtf((float)D), c.n_heads, c.n_kv_heads, D, tab, l, qb, fused ? 1 : 0);
            SD_LAUNCH_CHECK();
        }
        bool o_done = false, xn_o = false, attn_done = false;
        if constexpr (!std::is_same<T, float>::value) {
            if (attn_oproj_ok<T>(s, tab, s_max)) {
                // the residual add in the O projection's epilogue, the norm in gate/up's operand load (no launch between)
                xn_o = pre && fused && norm_on_load_ok<T>(s, tab) && !m->bo[l] && !gemm_plan(gu_cols(c), H, n_new).tiled;
                ProfScope ps(s, PC_ATTN, st);
                if ((rc = launch_attn_oproj<T>(s, qb, tab, l, at, s_max, m->wo[l], xn_o, st)) != SD_OK) return rc;
                SD_LAUNCH_CHECK();
                go.S = 1;
                go.stride_s = (size_t)16 * H;
                o_done = true;
            }
        }
        if constexpr (!std::is_same<T, float>::value) {
            if (!o_done && prefill_attn_ok<T>(s, tab, s_max)) {
                ProfScope ps(s, PC_ATTN, st);
                if ((rc = launch_attn_prefill<T>(s, qb, tab, l, at, s_max, st)) != SD_OK) return rc;
                SD_LAUNCH_CHECK();
                attn_done = true;
            }
        }
        if (!o_done && !attn_done) {
            ProfScope ps(s, PC_ATTN, st);
            switch (D) {
                case 16: rc = launch_attn<T, 16>(s, qb, tab, l, at, s_max, st); break;
                case 32: rc = launch_attn<T, 32>(s, qb, tab, l, at, s_max, st); break;
                case 64: rc = launch_attn<T, 64>(s, qb, tab, l, at, s_max, st); break;
                default: rc = launch_attn<T, 128>(s, qb, tab, l, at, s_max, st); break;
            }
            if (rc != SD_OK) return rc;
            SD_LAUNCH_CHECK();
        }
        // output projection + residual (+ norm feeding the MLP)
        if (!o_done && (rc = run_gemm<H16>(s, m->wo[l], at, n_new, H, q_dim(c), &go, st, nullptr, s->tp && g_env.tp_one_slab)) != SD_OK) return rc;
        const float *osrc = s->part;
        if ((rc = tp_reduce(s, &go, &osrc, n_new, H, st)) != SD_OK) return rc;
        if (!xn_o) {
            ProfScope ps(s, PC_NORM, st);
            const int mode = pre ? RES_PRE : RES_POST;
            const T *nw = pre ? (const T *)m->n2w[l] : (const T *)m->n1w[l];
            const T *nb = pre ? (const T *)m->n2b[l] : (const T *)m->n1b[l];
            hipLaunchKernelGGL((residual_norm_kernel<T>), dim3(n_new), dim3(rn_threads), 0, st, x, osrc, go.S,
                               go.stride_s, H, (const T *)m->bo[l], nw, nb, c.norm_eps, norm_kind, mode, h);
            SD_LAUNCH_CHECK();
        }
        // MLP
        if constexpr (!std::is_same<T, float>::value) {
            if (xn_o) {
                GemmEpiT<H16> e = {};
                e.out = (H16 *)ac; e.n_out = I;
                if ((rc = launch_gemm_xn<EPI_ACT_SILU, H16>(s, m->wgu[l], h, n_new, gu_cols(c), H, m->n2w[l], c.norm_eps, e, st)) != SD_OK)
                    return rc;
            }
        }
        if (xn_o) {
        } else if (fused && fused_plan_ok(gemm_plan(gu_cols(c), H, n_new, true, true))) {
            GemmEpiT<H16> e = {};
            e.out = (H16 *)ac; e.bias = (const H16 *)m->bfc1[l]; e.n_out = I;
            rc = llama ? run_gemm_fused<EPI_ACT_SILU, H16>(s, m->wgu[l], h, n_new, gu_cols(c), H, e, st)
                       : run_gemm_fused<EPI_ACT_RELU, H16>(s, m->wgu[l], h, n_new, gu_cols(c), H, e, st);
            if (rc != SD_OK) return rc;
        } else {
            if ((rc = run_gemm<H16>(s, m->wgu[l], h, n_new, gu_cols(c), H, &go, st)) != SD_OK) return rc;
            ProfScope ps(s, PC_ACT, st);
            hipLaunchKernelGGL((act_kernel<T>), dim3((I + 255) / 256, n_new), dim3(256), 0, st, s->part, go.S,
                               go.stride_s, I, gu_cols(c), c.arch, (const T *)m->bfc1[l], ac, fused ? 1 : 0);
            SD_LAUNCH_CHECK();
        }
        // down projection: with the norm-on-load seam its last k-slab's workgroups add the residual and the next layer's QKV
        // normalises on load (no residual+norm launch); the last layer keeps the launch (the final norm feeds the head)
        xn_d = false;
        if constexpr (!std::is_same<T, float>::value) {
            xn_d = g_env.norm_on_load > 1 && pre && fused && l + 1 < L && norm_on_load_ok<T>(s, tab) && !m->bfc2[l] &&
                   !gemm_plan(qkv_cols(c), H, n_new).tiled && !gemm_plan(H, I, n_new).tiled;
            if (xn_d && (rc = launch_gemm_fin<H16>(s, m->wdown[l], ac, n_new, H, I, nullptr, st)) != SD_OK) return rc;
        }
        if (xn_d) continue;
        if ((rc = run_gemm<H16>(s, m->wdown[l], ac, n_new, H, I, &go, st, nullptr, s->tp && g_env.tp_one_slab)) != SD_OK) return rc;
        const float *dsrc = s->part;
        if ((rc = tp_reduce(s, &go, &dsrc, n_new, H, st)) != SD_OK) return rc;
        {
            ProfScope ps(s, PC_NORM, st);
            int mode;
            const T *nw = nullptr, *nb = nullptr;
            if (pre) {
                if (l + 1 < L) { mode = RES_PRE; nw = (const T *)m->n1w[l + 1]; nb = (const T *)m->n1b[l + 1]; }
                else if (m->w.final_norm_w) { mode = RES_PRE; nw = (const T *)m->w.final_norm_w; nb = (const T *)m->w.final_norm_b; }
                else mode = RES_NONE;
            } else {
                mode = RES_POST; nw = (const T *)m->n2w[l]; nb = (const T *)m->n2b[l];
            }
            hipLaunchKernelGGL((residual_norm_kernel<T>), dim3(n_new), dim3(rn_threads), 0, st, x, dsrc, go.S,
                               go.stride_s, H, (const T *)m->bfc2[l], nw, nb, c.norm_eps, norm_kind, mode, h);
            SD_LAUNCH_CHECK();
        }
    }

    if (n_logits > 0) {
        // only the rows tab.xmap lists feed the head (the last rows of each stream)
        const T *hl = h;
        const RowTab *xt = &tab;
        if (ED != H) {                                          // OPT project_out (modeling_opt.py:744-745)
            if ((rc = run_gemm<H16>(s, m->w.project_out, hl, n_logits, ED, H, &go, st, xt)) != SD_OK) return rc;
            xt = nullptr;
            ProfScope ps(s, PC_LOGITS, st);
            hipLaunchKernelGGL((reduce_rows_kernel<T>), dim3((ED + 255) / 256, n_logits), dim3(256), 0, st, s->part,
                               go.S, go.stride_s, ED, eb);
            SD_LAUNCH_CHECK();
            hl = eb;
        }
        return head_logits<T>(s, hl, xt, n_logits, logits_out, ld_logits, st);
    }
    return SD_OK;
}

// fill the attention groups and the lm_head row map of a table whose rows are already laid out stream by stream
static void finish_table(RowTab &tab) {
    tab.n_groups = 0;
    if (tab.contig == 2) {                          // runs of several streams: groups of ATT_TQ consecutive rows inside a run
        for (int i = 0; i < tab.n_streams; ++i)
            for (int r = tab.seg_row0[i]; r < tab.seg_row0[i + 1]; r += ATT_TQ) {
                const int g = tab.n_groups++;
                tab.grp_row0[g] = r;
                tab.grp_n[g] = std::min(ATT_TQ, tab.seg_row0[i + 1] - r);
                tab.grp_pos[g] = tab.seg_pos0[i] + (r - tab.seg_row0[i]);
                tab.grp_stream[g] = i;
            }
        return;
    }
    if (tab.contig) {                               // up to 256 rows = 32 groups of ATT_TQ consecutive positions
        for (int r = 0; r < tab.n_rows; r += ATT_TQ) {
            const int g = tab.n_groups++;
            tab.grp_row0[g] = r;
            tab.grp_n[g] = std::min(ATT_TQ, tab.n_rows - r);
            tab.grp_pos[g] = tab.pos0 + r;
            tab.grp_stream[g] = 0;
        }
        return;
    }
    int r = 0;
    while (r < tab.n_rows) {
        int n = 1;
        while (r + n < tab.n_rows && n < ATT_TQ && tab.row_stream[r + n] == tab.row_stream[r] &&
               (tab.tree || tab.row_pos[r + n] == tab.row_pos[r] + n))     // tree rows group by storage order
            ++n;
        tab.grp_row0[tab.n_groups] = r;
        tab.grp_n[tab.n_groups] = n;
        tab.grp_pos[tab.n_groups] = tab.row_pos[r];
        tab.grp_stream[tab.n_groups] = tab.row_stream[r];
        ++tab.n_groups;
        r += n;
    }
}

static int launch_attn_bf16(sd_session *s, const bf16_t *q, const RowTab &tab, int layer, bf16_t *out, int s_max,
                            hipStream_t st) {
    switch (s->m->cfg.head_dim) {
        case 16: return launch_attn<bf16_t, 16>(s, q, tab, layer, out, s_max, st);
        case 32: return launch_attn<bf16_t, 32>(s, q, tab, layer, out, s_max, st);
        case 64: return launch_attn<bf16_t, 64>(s, q, tab, layer, out, s_max, st);
        default: return launch_attn<bf16_t, 128>(s, q, tab, layer, out, s_max, st);
    }
}

static int run_forward(sd_session *s, const RowTab &tab, int s_max, float *logits_out, long ld_logits, void *stream) {
    int rc;
    if (s->resync) {
        // the previous forward of this session failed part-way (or ran with a test skew): its fused launches' arrival counters
        // and the host's epochs may disagree, and every later wait would then run into its 20 ms limit.  Stream-ordered
        // re-zeroing behind whatever of that forward did run puts both back in step.
        SD_HIP_CHECK(hipMemsetAsync(s->fin_ctr, 0, (size_t)(s->m->cfg.hidden / 16 + 1) * sizeof(unsigned), (hipStream_t)stream));
        SD_HIP_CHECK(hipMemsetAsync(s->ao_ctr, 0, 128, (hipStream_t)stream));
        s->ao_epoch = s->fin_epoch = 0;
        s->resync = 0;
    }
    s->skew_now = s->test_skew;
    s->test_skew = 0;
    if (s->m->cfg.dtype == SD_BF16)
        rc = forward_impl<bf16_t>(s, tab, s_max, logits_out, ld_logits, (hipStream_t)stream);
    else if (s->m->cfg.dtype == SD_F16)
        rc = forward_impl<f16_t>(s, tab, s_max, logits_out, ld_logits, (hipStream_t)stream);
    else
        rc = forward_impl<float>(s, tab, s_max, logits_out, ld_logits, (hipStream_t)stream);
    if (rc != SD_OK || s->skew_now) s->resync = 1;
    s->skew_now = 0;
    return rc;
}

extern "C" int sd_session_forward(sd_session *s, const int32_t *tokens, int n_new, int pos0, int n_logits,
                                  float *logits_out, long ld_logits, void *stream) {
    SD_REQUIRE(s && tokens, "sd_session_forward: null argument");
    SD_REQUIRE(n_new >= 1 && pos0 >= 0, "sd_session_forward: n_new=%d pos0=%d", n_new, pos0);
    SD_REQUIRE(n_logits >= 0 && n_logits <= n_new, "sd_session_forward: n_logits=%d of n_new=%d", n_logits, n_new);
    SD_REQUIRE(n_logits == 0 || logits_out, "sd_session_forward: logits_out is null");
    // (logit rows are gathered by the streaming kernel, <= 64 rows per call, unless every row of the call is one)
    if (n_new > s->max_rows || n_new > SD_MAX_FWD_ROWS || n_logits > (n_logits == n_new ? SD_MAX_ROWS : SD_STREAM_MAX_ROWS) ||
        pos0 + n_new > s->max_seq) {
        sd_set_error("sd_session_forward: n_new=%d (max_rows %d, <=%d), n_logits=%d (<=%d), pos0+n_new=%d (max_seq %d)",
                     n_new, s->max_rows, SD_MAX_FWD_ROWS, n_logits, SD_MAX_ROWS, pos0 + n_new, s->max_seq);
        return SD_ERR_CAPACITY;
    }
    RowTab tab = {};
    tab.n_rows = n_new;
    tab.n_streams = 1;
    tab.n_logit_rows = n_logits;
    tab.tok_base[0] = tokens - pos0;            // indexed by absolute position, only ever read at pos0 .. pos0+n_new-1
    tab.kv_base[0] = s->kv;
    tab.max_seq[0] = s->max_seq;
    tab.kv_fp8 = s->kv_fp8;
    tab.kv_scale[0] = s->kv_scale;
    if (n_new > SD_MAX_ROWS) {                      // prefill chunk: positions are implicit
        tab.contig = 1;
        tab.pos0 = pos0;
    } else {
        for (int i = 0; i < n_new; ++i) { tab.row_pos[i] = pos0 + i; tab.row_stream[i] = 0; }
    }
    for (int i = 0; i < n_logits; ++i) tab.xmap[i] = (unsigned char)(n_new - n_logits + i);   // n_new <= 256
    finish_table(tab);
    return run_forward(s, tab, pos0 + n_new, logits_out, ld_logits, stream);
}

// Tree verify (reference kvcache_model.py:38-136 forward_tree_attention + modeling_llama.py:684-689): the n rows are
// the nodes of a draft token tree.  Node i has token tokens[i], position positions[i] (its depth: RoPE / learned
// position), sees all `base_len` cached positions and the nodes whose bit is set in masks[i] (ancestors + itself, bit j
// = node j, j <= i), and its K / V rows are appended at arena slot base_len + i.  Logits come out for all n nodes.
extern "C" int sd_session_forward_tree(sd_session *s, const int32_t *tokens, const int32_t *positions, const uint64_t *masks,
                                       int n, int base_len, float *logits_out, long ld_logits, void *stream) {
    SD_REQUIRE(s && tokens && positions && masks && logits_out, "sd_session_forward_tree: null argument");
    SD_REQUIRE(n >= 1 && n <= SD_MAX_TREE && base_len >= 0, "sd_session_forward_tree: 1..%d nodes", SD_MAX_TREE);
    if (n > s->max_rows || base_len + n > s->max_seq) {
        sd_set_error("sd_session_forward_tree: %d nodes after %d positions exceed max_rows %d / max_seq %d", n, base_len,
                     s->max_rows, s->max_seq);
        return SD_ERR_CAPACITY;
    }
    RowTab tab = {};
    tab.n_rows = n;
    tab.n_streams = 1;
    tab.n_logit_rows = n;
    tab.tree = 1;
    tab.tree_base = base_len;
    tab.kv_base[0] = s->kv;
    tab.max_seq[0] = s->max_seq;
    tab.kv_fp8 = s->kv_fp8;
    tab.kv_scale[0] = s->kv_scale;
    tab.tok_base[0] = tokens;                    // tree rows: the token of node i is tokens[i] (tab_tok)
    for (int i = 0; i < n; ++i) {
        SD_REQUIRE(positions[i] >= 0 && positions[i] < s->m->cfg.max_pos, "sd_session_forward_tree: position %d out of range", positions[i]);
        SD_REQUIRE((masks[i] >> i) & 1ull, "sd_session_forward_tree: node %d must see itself", i);
        SD_REQUIRE(i == 63 || (masks[i] >> (i + 1)) == 0, "sd_session_forward_tree: node %d sees a later node", i);
        tab.row_pos[i] = positions[i];
        tab.row_stream[i] = 0;
        tab.tree_mask[i] = masks[i];
        tab.xmap[i] = (unsigned char)i;
    }
    finish_table(tab);
    return run_forward(s, tab, base_len + n, logits_out, ld_logits, stream);
}

// Gather-compaction of the KV arena after a tree verify (reference kvcache_model.py:326-353 rollback_tree_attention,
// one kept path): the rows at slots base + idx[j] (idx ascending, j < k) move to base + j in every layer / head.
__global__ void kv_compact_kernel(char *kv, int max_seq, int row_bytes, int base, const int *__restrict__ idx, int k) {
    extern __shared__ __attribute__((aligned(16))) char sm[];
    char *arena = kv + (size_t)blockIdx.x * max_seq * row_bytes;       // one (layer, k|v, head) plane per workgroup
    const int n16 = row_bytes / 16;
    for (int i = threadIdx.x; i < k * n16; i += blockDim.x) {
        const int j = i / n16, c = i - j * n16;
        reinterpret_cast<uint4 *>(sm)[i] = reinterpret_cast<const uint4 *>(arena + (size_t)(base + idx[j]) * row_bytes)[c];
    }
    __syncthreads();
    for (int i = threadIdx.x; i < k * n16; i += blockDim.x) {
        const int j = i / n16, c = i - j * n16;
        reinterpret_cast<uint4 *>(arena + (size_t)(base + j) * row_bytes)[c] = reinterpret_cast<const uint4 *>(sm)[i];
    }
}

// debugging aid (tools/ao_stamps.py): the per-workgroup records the last fused attention + O launch left (SD_AO_STAMPS=1):
// out[w][8] for workgroup w < n_wgs <= AO_STAMP_WGS - see AO_ST_* in fused_kernels.h
extern "C" int sd_session_ao_stamps(sd_session *s, long long *out, int n_wgs) {
    SD_REQUIRE(s && out && n_wgs >= 1 && n_wgs <= AO_STAMP_WGS, "sd_session_ao_stamps: 1..%d workgroup records", AO_STAMP_WGS);
    SD_HIP_CHECK(hipMemcpy(out, s->ao_stamps, (size_t)n_wgs * 8 * sizeof(long long), hipMemcpyDeviceToHost));
    return SD_OK;
}

// Sticky status of the in-launch waits of this session's fused launches (attention + O projection: bit 0; k-split GEMM with
// the residual epilogue: bit 1).  A wait that ran into its 20 ms limit poisons its output with NaN - the sampler then
// reports 'norm logits error' - and sets its bit here, so the host can tell a timed-out hand-off from a model that
// produced NaN.  Reads and clears the word; call with the session's stream idle.
extern "C" int sd_session_fused_status(sd_session *s, unsigned *status_out) {
    SD_REQUIRE(s && status_out, "sd_session_fused_status: null argument");
    SD_HIP_CHECK(hipMemcpy(status_out, s->wait_status, sizeof(unsigned), hipMemcpyDeviceToHost));
    if (*status_out) {
        SD_HIP_CHECK(hipMemset(s->wait_status, 0, sizeof(unsigned)));
        s->resync = 1;                                            // the counters may be short of the epochs now
    }
    return SD_OK;
}

// TEST HOOK: the fused launches of the NEXT forward of this session expect `extra` more arrivals than will come, so their
// waits run into the time limit (tests/: the timeout branch must surface as NaN logits / 'norm logits error' and a status
// bit, never as finite numbers).  The forward after that re-synchronises counters and epochs.
extern "C" int sd_session_test_skew_wait(sd_session *s, int extra) {
    SD_REQUIRE(s && extra >= 0 && extra <= 1024, "sd_session_test_skew_wait: extra in 0..1024");
    s->test_skew = extra;
    return SD_OK;
}

extern "C" int sd_session_compact_kv(sd_session *s, int base_len, const int32_t *idx_dev, int k, void *stream) {
    SD_REQUIRE(s && idx_dev && k >= 0 && k <= SD_MAX_TREE && base_len >= 0 && base_len + k <= s->max_seq,
               "sd_session_compact_kv: bad arguments");
    if (k == 0) return SD_OK;
    const sd_model_config &c = s->m->cfg;
    const int row_bytes = c.head_dim * (s->kv_fp8 ? 1 : (int)esize(c.dtype));
    SD_REQUIRE(row_bytes % 16 == 0, "sd_session_compact_kv: KV rows must be multiples of 16 bytes");
    const int planes = c.n_layers * 2 * c.n_kv_heads;
    hipLaunchKernelGGL(kv_compact_kernel, dim3(planes), dim3(256), (size_t)k * row_bytes, (hipStream_t)stream, s->kv, s->max_seq,
                       row_bytes, base_len, idx_dev, k);
    SD_LAUNCH_CHECK();
    return SD_OK;
}

// Stream-batched forward: the new rows of up to 16 independent sequences go through ONE pass
// over the weights.  Every item names its own session (KV arena), token buffer (device int32, indexed by absolute
// position), cache length pos0, number of new rows and how many of its last rows need logits.  Activations use
// items[0].session's scratch; all sessions must belong to the same model.  Logit rows come out packed in item order.
extern "C" int sd_batch_forward(const sd_batch_item *items, int n_items, float *logits_out, long ld_logits,
                                void *stream) {
    SD_REQUIRE(items && n_items >= 1 && n_items <= SD_MAX_STREAMS, "sd_batch_forward: 1..%d items", SD_MAX_STREAMS);
    sd_session *s0 = items[0].session;
    SD_REQUIRE(s0, "sd_batch_forward: null session");
    RowTab tab = {};
    int rows = 0, nlog = 0, s_max = 0;
    for (int i = 0; i < n_items; ++i) {
        const sd_batch_item &it = items[i];
        SD_REQUIRE(it.session && it.seq && it.session->m == s0->m, "sd_batch_forward: item %d: null or foreign session", i);
        SD_REQUIRE(it.n_new >= 1 && it.pos0 >= 0 && it.n_logits >= 0 && it.n_logits <= it.n_new,
                   "sd_batch_forward: item %d: n_new=%d pos0=%d n_logits=%d", i, it.n_new, it.pos0, it.n_logits);
        if (it.pos0 + it.n_new > it.session->max_seq || rows + it.n_new > std::min(s0->max_rows, SD_MAX_ROWS)) {
            sd_set_error("sd_batch_forward: item %d overflows (rows %d+%d of %d, positions %d of %d)", i, rows, it.n_new,
                         std::min(s0->max_rows, SD_MAX_ROWS), it.pos0 + it.n_new, it.session->max_seq);
            return SD_ERR_CAPACITY;
        }
        SD_REQUIRE(it.session->kv_fp8 == s0->kv_fp8, "sd_batch_forward: item %d: KV arenas of different dtypes", i);
        tab.tok_base[i] = it.seq;
        tab.kv_base[i] = it.session->kv;
        tab.max_seq[i] = it.session->max_seq;
        tab.kv_fp8 = it.session->kv_fp8;
        tab.kv_scale[i] = it.session->kv_scale;
        for (int r = 0; r < it.n_new; ++r) {
            tab.row_pos[rows + r] = it.pos0 + r;
            tab.row_stream[rows + r] = (unsigned char)i;
        }
        for (int r = 0; r < it.n_logits; ++r) tab.xmap[nlog++] = (unsigned char)(rows + it.n_new - it.n_logits + r);
        rows += it.n_new;
        s_max = std::max(s_max, it.pos0 + it.n_new);
    }
    SD_REQUIRE(nlog == 0 || logits_out, "sd_batch_forward: logits_out is null");
    tab.n_rows = rows;
    tab.n_streams = n_items;
    tab.n_logit_rows = nlog;
    finish_table(tab);
    return run_forward(s0, tab, s_max, logits_out, ld_logits, stream);
}

// Batched prefill (throughput mode): the prompts of several streams through ONE pass over the weights - up to
// SD_MAX_FWD_ROWS rows and SD_MAX_GROUPS attention groups in all, every stream's rows a contiguous run of positions
// (RowTab contig = 2), no logits.  Replaces stream-by-stream prefill passes of the same weights.
extern "C" int sd_batch_prefill(const sd_batch_item *items, int n_items, void *stream) {
    SD_REQUIRE(items && n_items >= 1 && n_items <= SD_MAX_STREAMS, "sd_batch_prefill: 1..%d items", SD_MAX_STREAMS);
    sd_session *s0 = items[0].session;
    SD_REQUIRE(s0, "sd_batch_prefill: null session");
    RowTab tab = {};
    int rows = 0, groups = 0, s_max = 0;
    for (int i = 0; i < n_items; ++i) {
        const sd_batch_item &it = items[i];
        SD_REQUIRE(it.session && it.seq && it.session->m == s0->m, "sd_batch_prefill: item %d: null or foreign session", i);
        SD_REQUIRE(it.n_new >= 1 && it.pos0 >= 0 && it.n_logits == 0, "sd_batch_prefill: item %d: n_new=%d pos0=%d n_logits=%d",
                   i, it.n_new, it.pos0, it.n_logits);
        SD_REQUIRE(it.session->kv_fp8 == s0->kv_fp8, "sd_batch_prefill: item %d: KV arenas of different dtypes", i);
        groups += (it.n_new + ATT_TQ - 1) / ATT_TQ;
        if (it.pos0 + it.n_new > it.session->max_seq || rows + it.n_new > std::min(s0->max_rows, SD_MAX_FWD_ROWS) ||
            groups > SD_MAX_GROUPS) {
            sd_set_error("sd_batch_prefill: item %d overflows (rows %d+%d of %d, %d attention groups of %d, positions %d of %d)", i,
                         rows, it.n_new, std::min(s0->max_rows, SD_MAX_FWD_ROWS), groups, SD_MAX_GROUPS, it.pos0 + it.n_new,
                         it.session->max_seq);
            return SD_ERR_CAPACITY;
        }
        tab.tok_base[i] = it.seq;
        tab.kv_base[i] = it.session->kv;
        tab.max_seq[i] = it.session->max_seq;
        tab.kv_fp8 = it.session->kv_fp8;
        tab.kv_scale[i] = it.session->kv_scale;
        tab.seg_row0[i] = rows;
        tab.seg_pos0[i] = it.pos0;
        rows += it.n_new;
        s_max = std::max(s_max, it.pos0 + it.n_new);
    }
    tab.seg_row0[n_items] = rows;
    tab.contig = 2;
    tab.n_rows = rows;
    tab.n_streams = n_items;
    tab.n_logit_rows = 0;
    finish_table(tab);
    return run_forward(s0, tab, s_max, nullptr, 0, stream);
}

// ---- standalone weight-streaming GEMM (unit tests + kernel-level roofline runs) --------------
__global__ void reduce_f32_kernel(const float *__restrict__ part, int S, size_t stride_s, int total,
                                  float *__restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    out[i] = reduce_part<float>(part, S, stride_s, (size_t)i, nullptr, 0);
}

extern "C" int sd_pack_activation_bf16(const void *x_rowmajor, void *x_tiled, int M, int K, void *stream) {
    SD_REQUIRE(x_rowmajor && x_tiled && M >= 1 && K >= 32 && K % 32 == 0, "sd_pack_activation_bf16: need M>=1, K%%32==0");
    hipLaunchKernelGGL((to_operand_kernel<bf16_t>), dim3(M), dim3(256), 0, (hipStream_t)stream,
                       (const bf16_t *)x_rowmajor, K, (bf16_t *)x_tiled);
    SD_LAUNCH_CHECK();
    return SD_OK;
}

extern "C" int sd_gemm_bf16(const void *w_packed, const void *x, int x_tiled, int M, int N, int K, float *part,
                            size_t part_floats, float *out, int *splits_out, void *stream) {
    SD_REQUIRE(w_packed && x && part, "sd_gemm_bf16: null argument");
    refresh_env();
    SD_REQUIRE(M >= 1 && M <= SD_MAX_FWD_ROWS && N % 16 == 0 && K % 32 == 0,
               "sd_gemm_bf16: need 1<=M<=%d, N%%16==0, K%%32==0", SD_MAX_FWD_ROWS);
    const GemmPlan pl = gemm_plan(N, K, M, x_tiled != 0);
    int S = pl.S, ksp = pl.ksp;
    const int Mpad = (int)align_up(M, 16);
    if (pl.tiled) {
        SD_REQUIRE((size_t)S * Mpad * N <= part_floats, "sd_gemm_bf16: part buffer needs %zu floats", (size_t)S * Mpad * N);
        launch_gemm_tiled(w_packed, x, part, M, Mpad, N, K, pl, (hipStream_t)stream);
        SD_LAUNCH_CHECK();
        if (out) {
            hipLaunchKernelGGL(reduce_f32_kernel, dim3((M * N + 255) / 256), dim3(256), 0, (hipStream_t)stream, part, S,
                               (size_t)Mpad * N, M * N, out);
            SD_LAUNCH_CHECK();
        }
        if (splits_out) *splits_out = S;
        return SD_OK;
    }
    if (x_tiled) {
        const RowsPlan rp = rows_plan(N, K, M, false);
        if (rp.ok) {
            SD_REQUIRE((size_t)rp.S * Mpad * N <= part_floats, "sd_gemm_bf16: part buffer needs %zu floats", (size_t)rp.S * Mpad * N);
            GemmEpi e0 = {};
            if (launch_gemm_rows<EPI_PART, bf16_t>(w_packed, x, part, M, Mpad, N, K, rp, e0, (hipStream_t)stream) != SD_OK)
                return SD_ERR_INVALID;
            SD_LAUNCH_CHECK();
            if (out) {
                hipLaunchKernelGGL(reduce_f32_kernel, dim3((M * N + 255) / 256), dim3(256), 0, (hipStream_t)stream, part, rp.S,
                                   (size_t)Mpad * N, M * N, out);
                SD_LAUNCH_CHECK();
            }
            if (splits_out) *splits_out = rp.S;
            return SD_OK;
        }
    }
    SD_REQUIRE(M <= SD_STREAM_MAX_ROWS, "sd_gemm_bf16: more than 64 rows need the tile layout (x_tiled) and N %% 128 == 0");
    SD_REQUIRE((size_t)S * Mpad * N <= part_floats, "sd_gemm_bf16: part buffer needs %zu floats", (size_t)S * Mpad * N);
    hipStream_t st = (hipStream_t)stream;
    GemmEpi e = {};
    e.x_rowmajor = x_tiled ? 0 : 1;
    if (dispatch_gemm_bf16<EPI_PART>(w_packed, x, part, M, Mpad, N, K, S, ksp, e, st) != SD_OK) return SD_ERR_INVALID;
    SD_LAUNCH_CHECK();
    if (out) {
        hipLaunchKernelGGL(reduce_f32_kernel, dim3((M * N + 255) / 256), dim3(256), 0, st, part, S,
                           (size_t)Mpad * N, M * N, out);
        SD_LAUNCH_CHECK();
    }
    if (splits_out) *splits_out = S;
    return SD_OK;
}

// ---- one whole speculative iteration, enqueued natively ---------------------------------------
// reference sampling/speculative_sampling.py:1934-2031 for the device-RNG mode: gamma x (draft forward +
// norm_sample), one target forward over the uncached rows + norm_probs, accept scan, residual / bonus sample,
// then the 144-byte result block and the gamma+2 candidate tokens are copied to pinned host memory.  Nothing
// here synchronises; the caller waits on the stream once per iteration.
int sd_resample_with_errors(const float *p_hist, const float *q_hist, long ld, int V, int32_t *seq, int gamma,
                            uint64_t philox_seed, uint64_t draw_index, sd_accept_result *res, const int *err_flags,
                            int n_err, int dtype_mode, hipStream_t st);

// SD_NORM_DT_* of a model's probability rows: OPT keeps logits and probabilities in the weight dtype
// (modeling_opt.py:974), Llama casts its logits to fp32 (modeling_llama.py:870)
static int storage_mode(const sd_model *m) {
    if (m->cfg.arch != SD_ARCH_OPT) return 0;
    return m->cfg.dtype == SD_BF16 ? SD_NORM_DT_BF16 : (m->cfg.dtype == SD_F16 ? SD_NORM_DT_F16 : 0);
}

struct sd_spec {
    sd_session *draft, *target;
    int gamma, top_k, V;
    float temperature, top_p;
    int32_t *seq;
    float *q_hist, *p_hist;
    long ld;
    float *draft_logits, *target_logits;
    long ld_dl, ld_tl;
    int *err;                    // device ints: [0..gamma) norm err of draft rows, [gamma..2gamma) sample err, [2gamma..3gamma+1) target rows
    sd_accept_result *res_dev;
    void *norm_ws;               // sd_norm_workspace_bytes(gamma+1) bytes, may be NULL
    hipEvent_t ev[4];
    hipEvent_t ev_done;          // end of an iteration's device -> host copy (sd_spec_generate polls it)
    int timing;
};

extern "C" int sd_spec_create(sd_session *draft, sd_session *target, int gamma, float temperature, int top_k,
                              float top_p, int32_t *seq, float *q_hist, float *p_hist, long ld, float *draft_logits,
                              long ld_draft_logits, float *target_logits, long ld_target_logits, int *err_words,
                              sd_accept_result *res_dev, void *norm_workspace, sd_spec **out) {
    SD_REQUIRE(draft && target && seq && q_hist && p_hist && draft_logits && target_logits && err_words && res_dev && out,
               "sd_spec_create: null argument");
    SD_REQUIRE(gamma >= 1 && gamma <= 16, "sd_spec_create: gamma must be in 1..16");
    SD_REQUIRE(draft->m->cfg.vocab == target->m->cfg.vocab, "sd_spec_create: draft and target vocabularies differ");
    SD_REQUIRE(temperature != 0.0f, "sd_spec_create: temperature must be non-zero");
    refresh_env();
    sd_spec *sp = new sd_spec();
    sp->draft = draft; sp->target = target; sp->gamma = gamma; sp->temperature = temperature; sp->top_k = top_k;
    sp->top_p = top_p; sp->V = draft->m->cfg.vocab; sp->seq = seq; sp->q_hist = q_hist; sp->p_hist = p_hist; sp->ld = ld;
    sp->draft_logits = draft_logits; sp->ld_dl = ld_draft_logits; sp->target_logits = target_logits; sp->ld_tl = ld_target_logits;
    sp->err = err_words; sp->res_dev = res_dev; sp->norm_ws = norm_workspace; sp->timing = 0;
    for (int i = 0; i < 4; ++i) SD_HIP_CHECK(hipEventCreate(&sp->ev[i]));
    SD_HIP_CHECK(hipEventCreateWithFlags(&sp->ev_done, hipEventDisableTiming));
    *out = sp;
    return SD_OK;
}

extern "C" int sd_spec_destroy(sd_spec *sp) {
    if (!sp) return SD_OK;
    for (int i = 0; i < 4; ++i) (void)hipEventDestroy(sp->ev[i]);
    (void)hipEventDestroy(sp->ev_done);
    delete sp;
    return SD_OK;
}

extern "C" int sd_spec_timing(sd_spec *sp, int on) {
    SD_REQUIRE(sp, "sd_spec_timing: null handle");
    sp->timing = on;
    return SD_OK;
}

// milliseconds of the last iteration's draft phase and target (verify) phase; call after the stream is synchronised
extern "C" int sd_spec_last_times(sd_spec *sp, float *draft_ms, float *target_ms) {
    SD_REQUIRE(sp && draft_ms && target_ms && sp->timing, "sd_spec_last_times: timing is off");
    SD_HIP_CHECK(hipEventElapsedTime(draft_ms, sp->ev[0], sp->ev[1]));
    SD_HIP_CHECK(hipEventElapsedTime(target_ms, sp->ev[2], sp->ev[3]));
    return SD_OK;
}

extern "C" int sd_norm_sample(const float *logits, int V, float temperature, int top_k, float top_p,
                              int bf16_round_logits, float *probs_out, int *err_flag, const float *exp_noise,
                              uint64_t philox_seed, uint64_t draw_index, int *tok_out, int *sample_err, void *workspace,
                              void *stream);
extern "C" int sd_norm_probs(const float *logits, int rows, int V, long ld_in, float temperature, int top_k,
                             float top_p, int bf16_round_logits, float *probs_out, long ld_out, int *err_flag,
                             void *workspace, void *stream);
extern "C" int sd_accept_scan(const float *p_hist, const float *q_hist, long ld, const int32_t *seq, int L, int gamma,
                              const float *r, uint64_t philox_seed, uint64_t draw_index, sd_accept_result *out,
                              void *stream);

int sd_norm_rows_with_tiles(const float *logits, int rows, int V, long ld_in, float temperature, int top_k, float top_p,
                            int bf16_round_logits, float *probs_out, long ld_out, int *err_flag, uint64_t seed,
                            uint64_t draw, int *tok_out, int *samp_err, void *workspace, const float *tile_max,
                            void *stream, void *cand_lists);
size_t sd_norm_candrow_bytes(int rows);
int sd_norm_batch_tiles(const float *logits, int n_rows, int V, long ld_in, float temperature, int top_k, float top_p,
                        int bf16_round_logits, const sd_norm_row *rows, int sample, void *workspace, const float *tile_max,
                        void *cand_lists, void *stream);
int sd_accept_resample_batch(const sd_accept_item *items, int n_items, long ld, int V, int gamma, int dtype_mode,
                             const void *const *lists, void *stream);

// feed seq[from, upto) in chunks of at most max_rows; logits come out for the last n_logits rows, all of them from the
// final call (a chunk never ends inside the logits rows), so that call's output slab can be handed to the norm as is
static int feed_rows(sd_session *ses, const int32_t *seq, int from, int upto, int n_logits, float *logits, long ld,
                     void *stream) {
    const int first_logit = upto - n_logits;
    int done = from;
    while (done < upto) {
        int m = std::min(ses->max_rows, upto - done);
        if (done < first_logit && done + m > first_logit && done + m < upto) m = first_logit - done;
        const int lo = std::max(first_logit, done);
        const int nl = std::max(0, done + m - lo);
        const int rc = sd_session_forward(ses, seq + done, m, done, nl, nl ? logits + (size_t)(lo - first_logit) * ld : nullptr,
                                          ld, stream);
        if (rc != SD_OK) return rc;
        done += m;
    }
    return SD_OK;
}

extern "C" int sd_spec_iteration(sd_spec *sp, int L, int draft_len, int target_len, uint64_t seed_draft,
                                 uint64_t draw_draft0, uint64_t seed_accept, uint64_t draw_scan0,
                                 uint64_t draw_resample, const float *r_const, sd_accept_result *res_host,
                                 int32_t *tok_host, void *stream) {
    SD_REQUIRE(sp && res_host, "sd_spec_iteration: null argument");
    SD_REQUIRE(L >= 1 && draft_len >= 0 && draft_len < L && target_len >= 0 && target_len < L + sp->gamma,
               "sd_spec_iteration: L=%d draft_len=%d target_len=%d", L, draft_len, target_len);
    hipStream_t st = (hipStream_t)stream;
    const int g = sp->gamma, V = sp->V;
    int rc;
    // EPI_HEAD's tile maxima serve the top-k candidate search only (1 <= k <= 64, positive temperature, 16 | V >= 4096)
    const bool tiles_ok = sp->top_k >= 1 && sp->top_k <= 64 && sp->temperature > 0.0f && V % 16 == 0 && V >= 4096 &&
                          V <= 65536 && sp->ld % 4 == 0 && g_env.head_tiles;
    if (sp->timing) SD_HIP_CHECK(hipEventRecord(sp->ev[0], st));
    // ---- draft: gamma steps; the sampled token goes straight into seq[] where the next step's embed reads it
    for (int i = 0; i < g; ++i) {
        const int upto = L + i;
        float *q_row = sp->q_hist + (size_t)(upto - 1) * sp->ld;
        sp->draft->want_raw_logits = 1;
        sp->draft->head_zero_rows = tiles_ok ? q_row : nullptr;      // the head clears the row and leaves tile maxima
        sp->draft->head_zero_ld = sp->ld;
        rc = feed_rows(sp->draft, sp->seq, draft_len, upto, 1, sp->draft_logits, sp->ld_dl, stream);
        sp->draft->want_raw_logits = 0;
        sp->draft->head_zero_rows = nullptr;
        if (rc != SD_OK) return rc;
        draft_len = upto;
        if ((rc = sd_norm_rows_with_tiles(sp->draft->last_logits, 1, V, sp->draft->last_logits_ld, sp->temperature, sp->top_k,
                                          sp->top_p, sp->draft->last_logits_round | storage_mode(sp->draft->m), q_row, sp->ld, sp->err + i, seed_draft,
                                          draw_draft0 + (uint64_t)i, sp->seq + upto, sp->err + g + i, sp->norm_ws,
                                          sp->draft->last_tile_max, stream, nullptr)) != SD_OK)
            return rc;
    }
    if (sp->timing) { SD_HIP_CHECK(hipEventRecord(sp->ev[1], st)); SD_HIP_CHECK(hipEventRecord(sp->ev[2], st)); }
    // the target rows' candidate lists (behind the CandRows of the workspace) let the residual / bonus sample skip its
    // passes over V; they exist when all gamma + 1 rows of this iteration are normalised here
    static const int sparse_on = getenv("SD_SPARSE_RESAMPLE") ? atoi(getenv("SD_SPARSE_RESAMPLE")) : 1;
    void *lists = nullptr;
    // ---- target: every uncached row in one pass (the whole prompt on the first call), logits for the last gamma+1
    {
        const int upto = L + g;
        const int rows = std::min(upto - target_len, g + 1);
        if (sparse_on && sp->norm_ws && rows == g + 1) lists = (char *)sp->norm_ws + sd_norm_candrow_bytes(g + 1);
        float *p_rows = sp->p_hist + (size_t)(upto - rows) * sp->ld;
        sp->target->want_raw_logits = 1;
        sp->target->head_zero_rows = tiles_ok ? p_rows : nullptr;
        sp->target->head_zero_ld = sp->ld;
        rc = feed_rows(sp->target, sp->seq, target_len, upto, rows, sp->target_logits, sp->ld_tl, stream);
        sp->target->want_raw_logits = 0;
        sp->target->head_zero_rows = nullptr;
        if (rc != SD_OK) return rc;
        if ((rc = sd_norm_rows_with_tiles(sp->target->last_logits, rows, V, sp->target->last_logits_ld, sp->temperature,
                                          sp->top_k, sp->top_p, sp->target->last_logits_round | storage_mode(sp->target->m), p_rows,
                                          sp->ld, sp->err + 2 * g,
                                          0, 0, nullptr, nullptr, sp->norm_ws, sp->target->last_tile_max, stream, lists)) != SD_OK)
            return rc;
    }
    if (sp->timing) SD_HIP_CHECK(hipEventRecord(sp->ev[3], st));
    // ---- accept scan + residual / bonus sample
    // p - q, max_fn and the draw are in the rows' dtype when both models keep 16-bit rows
    const int res_mode = storage_mode(sp->target->m) == storage_mode(sp->draft->m) ? storage_mode(sp->target->m) : 0;
    if (lists) {
        if ((rc = sd_accept_resample(sp->p_hist, sp->q_hist, sp->ld, V, sp->seq, L, g, r_const, seed_accept, draw_scan0,
                                     draw_resample, sp->res_dev, sp->err, 3 * g + 1, res_mode, lists, st)) != SD_OK)
            return rc;
    } else {
        if ((rc = sd_accept_scan(sp->p_hist, sp->q_hist, sp->ld, sp->seq, L, g, r_const, seed_accept, draw_scan0, sp->res_dev, stream)) != SD_OK)
            return rc;
        if ((rc = sd_resample_with_errors(sp->p_hist, sp->q_hist, sp->ld, V, sp->seq, g, seed_accept, draw_resample, sp->res_dev,
                                          sp->err, 3 * g + 1, res_mode, st)) != SD_OK)
            return rc;
    }
    SD_HIP_CHECK(hipMemcpyAsync(res_host, sp->res_dev, sizeof(sd_accept_result), hipMemcpyDeviceToHost, st));
    if (tok_host)       // optional second copy; the result block already carries the drafted tokens and the next one
        SD_HIP_CHECK(hipMemcpyAsync(tok_host, sp->seq + L, sizeof(int32_t) * (size_t)(g + 2), hipMemcpyDeviceToHost, st));
    return SD_OK;
}

// In-loop failure handling of the two native loops: every failure sets `rc` and leaves the loop, so the common epilogue
// (event destruction, write-back of the in/out cursor state) always runs.
#define SD_LOOP_HIP(expr)                                                                                   \
    if (hipError_t _e = (expr); _e != hipSuccess) {                                                         \
        sd_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__);          \
        rc = SD_ERR_HIP;                                                                                    \
        break;                                                                                              \
    }
// Waits for `ev` by polling (a blocking wait parks the thread, and the launches of the next iteration's draft steps - which
// the GPU consumes as fast as they arrive - then start from a cold core); bounded in wall-clock time.
static int poll_event(hipEvent_t ev, const char *who) {
    const auto t0 = std::chrono::steady_clock::now();
    for (unsigned spins = 0;; ++spins) {
        const hipError_t q = hipEventQuery(ev);
        if (q == hipSuccess) return SD_OK;
        if (q != hipErrorNotReady) { sd_set_error("%s: %s", who, hipGetErrorString(q)); return SD_ERR_HIP; }
        if ((spins & 0xfff) == 0xfff &&
            std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 60.0) {
            sd_set_error("%s: the iteration's result did not arrive within 60 s", who);
            return SD_ERR_HIP;
        }
    }
}

// The device-RNG loop without the interpreter between iterations (reference speculative_sampling.py:1934-2046).
extern "C" int sd_spec_generate(sd_spec *sp, int32_t *host_seq, int *len_io, int T, int eos_token_id, int ori_eos_cnt,
                                uint64_t *seed_io, uint64_t *draw_io, uint64_t random_seed, const float *r_const,
                                int *draft_len_io, int *target_len_io, sd_accept_result *res_host, int max_iters,
                                int32_t *acc_len_out, float *p_at_out, float *q_at_out, float *draft_ms_out,
                                float *target_ms_out, int *n_iters_out, int *err_out, void *stream) {
    SD_REQUIRE(sp && host_seq && len_io && seed_io && draw_io && draft_len_io && target_len_io && res_host && n_iters_out && err_out,
               "sd_spec_generate: null argument");
    SD_REQUIRE(!random_seed || r_const, "sd_spec_generate: random_seed needs its uniform (r_const)");
    const int g = sp->gamma;
    int len = *len_io, draft_len = *draft_len_io, target_len = *target_len_io, iters = 0, eos_total = ori_eos_cnt;
    uint64_t seed = *seed_io, draw = *draw_io;
    *err_out = 0;
    int rc = SD_OK;
    while (len < T && iters < max_iters) {
        const int L = len;
        const uint64_t d_draft = draw;                            // gamma draft samples, then the discarded target sample
        draw += (uint64_t)g + 1;
        const uint64_t seed_draft = seed;
        uint64_t d_scan = 0;
        if (random_seed) { seed = random_seed; draw = 0; }        // :1976-1977: the stream restarts before every uniform
        else { d_scan = draw; draw += (uint64_t)g; }
        const uint64_t d_res = draw++;
        if ((rc = sd_spec_iteration(sp, L, draft_len, target_len, seed_draft, d_draft, seed, d_scan, d_res, r_const, res_host,
                                    nullptr, stream)) != SD_OK)
            break;
        SD_LOOP_HIP(hipEventRecord(sp->ev_done, (hipStream_t)stream));
        if ((rc = poll_event(sp->ev_done, "sd_spec_generate")) != SD_OK) break;
        const sd_accept_result r = *res_host;
        if (r.flags & 2) { *err_out = 1; break; }
        if (r.flags & 8) {                                        // which word: a draft sample error, or a norm error
            std::vector<int> ew(3 * g + 1);
            SD_LOOP_HIP(hipMemcpy(ew.data(), sp->err, sizeof(int) * ew.size(), hipMemcpyDeviceToHost));
            bool samp = false;
            for (int i = g; i < 2 * g; ++i) samp = samp || ew[i] != 0;
            *err_out = samp ? 1 : 2;
            // a NaN row may be the poison of a timed-out in-launch wait (fused_kernels.h / normload_kernels.h): say so
            unsigned wd = 0, wt = 0;
            if (hipMemcpy(&wd, sp->draft->wait_status, sizeof(unsigned), hipMemcpyDeviceToHost) == hipSuccess &&
                hipMemcpy(&wt, sp->target->wait_status, sizeof(unsigned), hipMemcpyDeviceToHost) == hipSuccess && (wd | wt))
                sd_set_error("sd_spec_generate: a fused launch's in-launch wait timed out (status draft %#x, target %#x; "
                             "sd_session_fused_status): its rows were poisoned with NaN", wd, wt);
            break;
        }
        if (sp->timing && (draft_ms_out || target_ms_out)) {
            float dms = 0.f, tms = 0.f;
            SD_LOOP_HIP(hipEventElapsedTime(&dms, sp->ev[0], sp->ev[1]));
            SD_LOOP_HIP(hipEventElapsedTime(&tms, sp->ev[2], sp->ev[3]));
            if (draft_ms_out) draft_ms_out[iters] = dms;
            if (target_ms_out) target_ms_out[iters] = tms;
        }
        const int l = r.n_accepted, n = r.n;
        if (acc_len_out) acc_len_out[iters] = l;
        for (int i = 0; i < g; ++i) {
            if (p_at_out) p_at_out[(size_t)iters * g + i] = r.p_at[i];
            if (q_at_out) q_at_out[(size_t)iters * g + i] = r.q_at[i];
        }
        ++iters;
        if (!(n >= L - 1 && l >= 0 && l <= g)) {
            sd_set_error("sd_spec_generate: inconsistent result block (n %d, L %d, accepted %d)", n, L, l);
            rc = SD_ERR_INVALID;
            break;
        }
        for (int i = 0; i < l; ++i) host_seq[len++] = r.drafted[i];
        host_seq[len++] = r.next_token;
        draft_len = std::min(L + g - 1, n + 1);                   // rollback(n + 1) of both caches (:2000, :2015 / 2023)
        target_len = n + 1;
        for (int i = L; i < len; ++i) eos_total += host_seq[i] == eos_token_id;
        if (eos_total > ori_eos_cnt) break;                       // the caller cuts after the first new EOS (:2033-2041)
    }
    *len_io = len; *draft_len_io = draft_len; *target_len_io = target_len;
    *seed_io = seed; *draw_io = draw; *n_iters_out = iters;
    return rc;
}

// The stream-batched loop of sampling/batch.py in native code (reference algorithm per stream: speculative_sampling.py:1934-2046).
extern "C" int sd_spec_batch_generate(sd_batch_stream *streams, int n_streams, int gamma, float temperature, int top_k,
                                      float top_p, int V, long ld, int eos_token_id, uint64_t random_seed,
                                      const float *r_const, int draft_norm_mode, int target_norm_mode, float *draft_logits,
                                      long ld_draft_logits, float *target_logits, long ld_target_logits,
                                      void *norm_workspace, int max_rows_per_forward, float *verify_ms_out,
                                      int32_t *verify_streams_out, float *verify_ctx_out, int max_iters_log,
                                      int *n_iters_out, int *err_out, void *stream) {
    SD_REQUIRE(streams && n_streams >= 1 && n_streams <= 16 && gamma >= 1 && gamma <= 16 && draft_logits && target_logits &&
               n_iters_out && err_out, "sd_spec_batch_generate: bad arguments");
    SD_REQUIRE(!random_seed || r_const, "sd_spec_batch_generate: random_seed needs its uniform (r_const)");
    for (int i = 1; i < n_streams; ++i)
        SD_REQUIRE(streams[i].res_dev == streams[0].res_dev + i && streams[i].res_host == streams[0].res_host + i,
                   "sd_spec_batch_generate: the streams' result blocks must be consecutive");
    hipStream_t st = (hipStream_t)stream;
    const int g = gamma, n_err = 3 * g + 1;
    // streams per target pass (a 16-bit model whose GEMMs cannot take the tiled kernel carries 64 rows per forward, not 80)
    const int pass_rows = std::min({max_rows_per_forward, streams[0].target ? streams[0].target->max_rows : max_rows_per_forward, SD_MAX_ROWS});
    const int max_verify = std::max(1, pass_rows / (g + 1));
    hipEvent_t ev0 = nullptr, ev1 = nullptr, ev_done = nullptr;
    if (hipEventCreate(&ev0) != hipSuccess || hipEventCreate(&ev1) != hipSuccess ||
        hipEventCreateWithFlags(&ev_done, hipEventDisableTiming) != hipSuccess) {
        if (ev0) (void)hipEventDestroy(ev0);
        if (ev1) (void)hipEventDestroy(ev1);
        sd_set_error("sd_spec_batch_generate: hipEventCreate failed");
        return SD_ERR_HIP;
    }
    *err_out = 0;
    int iters = 0, rc = SD_OK;
    std::vector<sd_batch_stream *> act;
    std::vector<int> Ls;
    std::vector<uint64_t> base_draw;
    std::vector<sd_batch_item> items;
    std::vector<sd_norm_row> rows;
    std::vector<sd_accept_item> aitems;
    std::vector<const void *> list_of;
    // SD_BATCH_FUSED_TAIL=0: the round-1 sampling tail (logits copy, candidate pass + norm per draft step; dense accept scan +
    // resample) - kept for A/B runs and as the reference the fused tail is tested against; read per call
    const bool fused_tail = !(getenv("SD_BATCH_FUSED_TAIL") && atoi(getenv("SD_BATCH_FUSED_TAIL")) == 0);
    // EPI_HEAD's tile maxima serve the top-k candidate search only (as in sd_spec_iteration)
    const bool tiles_ok = top_k >= 1 && top_k <= 64 && temperature > 0.0f && V % 16 == 0 && V >= 4096 && V <= 65536 &&
                          ld % 4 == 0 && g_env.head_tiles;
    for (int i = 0; i < n_streams; ++i) { streams[i].done = 0; streams[i].calls = 0; }
    for (;;) {
        act.clear();
        for (int i = 0; i < n_streams; ++i) {
            sd_batch_stream &s = streams[i];
            if (!s.done && s.len >= s.T) s.done = 1;
            if (!s.done) act.push_back(&s);
        }
        if (act.empty()) break;
        const int n = (int)act.size();
        Ls.assign(n, 0);
        base_draw.assign(n, 0);
        double ctx = 0.0;
        for (int j = 0; j < n; ++j) {
            Ls[j] = act[j]->len;
            base_draw[j] = act[j]->draw;
            act[j]->draw += (uint64_t)g;
            ctx += Ls[j];
        }
        // the residual / bonus sample works on the target rows' candidate lists when ONE verify pass holds all streams (the
        // workspace keeps the lists of one pass)
        const bool lists_on = fused_tail && norm_workspace && n <= max_verify;
        char *const list_base = norm_workspace ? (char *)norm_workspace + sd_norm_candrow_bytes(max_rows_per_forward) : nullptr;
        const size_t list_stride = sd_cand_list_bytes(1);
        list_of.assign(n, nullptr);
        // ---- draft: gamma steps over all active streams
        for (int i = 0; i < g && rc == SD_OK; ++i) {
            items.assign(n, sd_batch_item{});
            rows.assign(n, sd_norm_row{});
            for (int j = 0; j < n; ++j) {
                sd_batch_stream &s = *act[j];
                items[j].session = s.draft; items[j].seq = s.seq; items[j].pos0 = s.draft_len;
                items[j].n_new = Ls[j] + i - s.draft_len; items[j].n_logits = 1;
                rows[j].probs_out = s.q_hist + (size_t)(Ls[j] + i - 1) * ld;
                rows[j].err = s.err_words + i;
                rows[j].exp_noise = nullptr;
                rows[j].philox_seed = s.seed;
                rows[j].draw_index = base_draw[j] + (uint64_t)i;
                rows[j].tok_out = s.seq + (Ls[j] + i);
                rows[j].sample_err = s.err_words + g + i;
                s.draft_len = Ls[j] + i;
            }
            // the single-stream loop's sampler feed (sd_spec_iteration): the head leaves the logits in its own slab, the
            // maximum of every 16-column tile, and clears the streams' probability rows - no logits copy, no candidate pass
            sd_session *d0 = act[0]->draft;
            if (fused_tail) {
                d0->want_raw_logits = 1;
                d0->head_zero_n = tiles_ok && n <= 16 ? n : 0;
                for (int j = 0; j < d0->head_zero_n; ++j) d0->head_zero_ptr[j] = rows[j].probs_out;
            }
            rc = sd_batch_forward(items.data(), n, draft_logits, ld_draft_logits, stream);
            d0->want_raw_logits = 0;
            d0->head_zero_n = 0;
            if (rc != SD_OK) break;
            if (fused_tail)
                rc = sd_norm_batch_tiles(d0->last_logits, n, V, d0->last_logits_ld, temperature, top_k, top_p,
                                         d0->last_logits_round | draft_norm_mode, rows.data(), 1, norm_workspace, d0->last_tile_max,
                                         nullptr, stream);
            else
                rc = sd_norm_batch(draft_logits, n, V, ld_draft_logits, temperature, top_k, top_p, draft_norm_mode, rows.data(), 1,
                                   norm_workspace, stream);
        }
        if (rc != SD_OK) break;
        // ---- verify: the uncached rows of every stream, max_verify streams per pass over the target weights
        SD_LOOP_HIP(hipEventRecord(ev0, st));
        for (int a0 = 0; a0 < n && rc == SD_OK; a0 += max_verify) {
            const int m = std::min(max_verify, n - a0);
            items.assign(m, sd_batch_item{});
            rows.clear();
            for (int j = 0; j < m; ++j) {
                sd_batch_stream &s = *act[a0 + j];
                const int L = Ls[a0 + j], nn = L + g - s.target_len;
                items[j].session = s.target; items[j].seq = s.seq; items[j].pos0 = s.target_len;
                items[j].n_new = nn; items[j].n_logits = nn;
                for (int r = 0; r < nn; ++r) {
                    sd_norm_row row = {};
                    row.probs_out = s.p_hist + (size_t)(L + g - nn + r) * ld;
                    row.err = s.err_words + 2 * g + std::min(r, g);
                    rows.push_back(row);
                }
            }
            if ((rc = sd_batch_forward(items.data(), m, target_logits, ld_target_logits, stream)) != SD_OK) break;
            // one pass holds every stream: the rows' candidate lists (behind the CandRows of the workspace) serve the
            // residual / bonus sample below
            rc = sd_norm_batch_tiles(target_logits, (int)rows.size(), V, ld_target_logits, temperature, top_k, top_p, target_norm_mode,
                                     rows.data(), 0, norm_workspace, nullptr, lists_on ? list_base : nullptr, stream);
            if (lists_on) {
                int r0 = 0;
                for (int j = 0; j < m; ++j) {                       // stream j's lists are valid when all its gamma + 1 rows are here
                    list_of[a0 + j] = items[j].n_new == g + 1 ? list_base + (size_t)r0 * list_stride : nullptr;
                    r0 += items[j].n_new;
                }
            }
        }
        if (rc != SD_OK) break;
        SD_LOOP_HIP(hipEventRecord(ev1, st));
        // ---- accept scan + residual / bonus sample, all streams in two launches
        aitems.assign(n, sd_accept_item{});
        for (int j = 0; j < n; ++j) {
            sd_batch_stream &s = *act[j];
            s.draw += 1;                                          // the discarded target sample
            uint64_t d_scan = 0;
            if (random_seed) { s.seed = random_seed; s.draw = 0; }
            else { d_scan = s.draw; s.draw += (uint64_t)g; }
            sd_accept_item &it = aitems[j];
            it.p_hist = s.p_hist; it.q_hist = s.q_hist; it.seq = s.seq; it.L = Ls[j];
            it.r = r_const; it.exp_noise = nullptr;
            it.philox_seed = s.seed; it.draw_scan = d_scan; it.draw_resample = s.draw++;
            it.res = s.res_dev; it.err_flags = s.err_words; it.n_err = n_err;
        }
        const int res_mode = target_norm_mode == draft_norm_mode ? target_norm_mode : 0;
        if (lists_on) rc = sd_accept_resample_batch(aitems.data(), n, ld, V, g, res_mode, list_of.data(), stream);
        else rc = sd_accept_batch(aitems.data(), n, ld, V, g, res_mode, stream);
        if (rc != SD_OK) break;
        SD_LOOP_HIP(hipMemcpyAsync(streams[0].res_host, streams[0].res_dev, sizeof(sd_accept_result) * (size_t)n_streams,
                                   hipMemcpyDeviceToHost, st));
        SD_LOOP_HIP(hipEventRecord(ev_done, st));
        if ((rc = poll_event(ev_done, "sd_spec_batch_generate")) != SD_OK) break;
        if (iters < max_iters_log) {
            float ms = 0.f;
            if (verify_ms_out && hipEventElapsedTime(&ms, ev0, ev1) == hipSuccess) verify_ms_out[iters] = ms;
            if (verify_streams_out) verify_streams_out[iters] = n;
            if (verify_ctx_out) verify_ctx_out[iters] = (float)(ctx / n + g);
        }
        ++iters;
        bool failed = false;
        for (int j = 0; j < n; ++j) {
            sd_batch_stream &s = *act[j];
            const sd_accept_result r = *s.res_host;
            if (r.flags & (2 | 8)) { failed = true; break; }
            const int L = Ls[j], l = r.n_accepted, nn = r.n;
            if (s.acc_len_out) s.acc_len_out[s.calls] = l;
            for (int i = 0; i < g; ++i) {
                if (s.p_at_out) s.p_at_out[(size_t)s.calls * g + i] = r.p_at[i];
                if (s.q_at_out) s.q_at_out[(size_t)s.calls * g + i] = r.q_at[i];
            }
            ++s.calls;
            for (int i = 0; i < l; ++i) s.host_seq[s.len++] = r.drafted[i];
            s.host_seq[s.len++] = r.next_token;
            s.draft_len = std::min(L + g - 1, nn + 1);
            s.target_len = nn + 1;
            int eos_total = 0;
            for (int i = 0; i < s.len; ++i) eos_total += s.host_seq[i] == eos_token_id;
            if (eos_total > s.ori_eos_cnt) s.done = 1;            // the caller cuts after the first new EOS
        }
        if (failed) { *err_out = 1; break; }
    }
    (void)hipEventDestroy(ev0); (void)hipEventDestroy(ev1); (void)hipEventDestroy(ev_done);
    *n_iters_out = iters;
    return rc;
}
