/*
 * specdec.h - C ABI of libspecdec.so, the MI355X (gfx950) speculative-sampling decode engine.
 *
 * The reference (ZongyueQin/LLMSpeculativeSampling) exposes this path as a Python API with no
 * FFI; the entry points below are what a binding for that path binds.
 * Each one names the reference code it replaces (file:line under the reference repo).
 *
 * Conventions
 *   - plain C, no torch types: raw device pointers, explicit sizes, a hipStream_t passed as void*.
 *   - every function returns 0 on success or a negative sd_status; sd_last_error() gives the text.
 *   - the library never owns caller memory: weights, KV arenas, probability arenas and scratch
 *     are allocated by the caller and handed over as pointers; handles own only host structs.
 *   - nothing here synchronises the stream unless its comment says so.
 */
#ifndef SPECDEC_H
#define SPECDEC_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SD_ABI_VERSION 4

typedef enum {
    SD_OK = 0,
    SD_ERR_INVALID = -1,      /* bad argument / unsupported shape                                  */
    SD_ERR_NORM_LOGITS = -2,  /* RuntimeError('norm logits error')   reference utils.py:203-207    */
    SD_ERR_PROB = -3,         /* RuntimeError('prob error')          reference utils.py:220-224    */
    SD_ERR_HIP = -4,          /* a HIP runtime call failed                                         */
    SD_ERR_CAPACITY = -5      /* sequence / row count beyond what the session was sized for        */
} sd_status;

typedef enum { SD_F32 = 0, SD_BF16 = 1, SD_F16 = 2 } sd_dtype;   /* fp16: what the reference harness loads (evaluation.py:185) */

/* dtype_mode of the sampling entry points (the `bf16_round_logits` argument of the norm_* calls is this word too).
 * The reference's Llama returns fp32 logits (modeling_llama.py:870: a 16-bit head's output cast with .float()), its OPT
 * keeps them - and with them the whole norm_logits / sample / max_fn chain - in the weight dtype (modeling_opt.py:974).
 *   SD_NORM_ROUND_*: the logits handed over are fp32 accumulators that still have to be rounded to the head's dtype;
 *   SD_NORM_DT_*:    the row lives in that 16-bit dtype: every tensor the reference materialises on the way (logits /
 *                    temperature, softmax, cumsum prefixes, the row sum and its log, log_softmax, exp, p - q, max_fn's
 *                    sum and quotient, p / noise) is rounded to it, as torch's bf16 / fp16 CPU kernels do (fp32 math, one
 *                    rounding per op).  Probability rows stay fp32 arrays; their values are then exactly representable.
 * Ties: with 16-bit logits the top-p cut can fall inside a run of equal logits; the reference's order inside such a run
 * is that of an unstable std::sort (torch.sort without stable=True, utils.py:170) and is not reproduced - this build
 * keeps the lowest token ids. */
#define SD_NORM_ROUND_BF16 1
#define SD_NORM_ROUND_F16 2
#define SD_NORM_DT_BF16 16
#define SD_NORM_DT_F16 32
typedef enum { SD_ARCH_LLAMA = 0, SD_ARCH_OPT = 1 } sd_arch;

int sd_version(void);
const char *sd_last_error(void);

/* ------------------------------------------------------------------------------------------
 * Sampling primitives                                            reference sampling/utils.py
 * ------------------------------------------------------------------------------------------ */

/* norm_logits + top_k_top_p_filter (utils.py:182-210, 152-179), one workgroup per row:
 * logits/temperature -> keep >= k-th largest (ties kept) -> stable descending order, drop
 * everything after the first token whose cumulative softmax mass exceeds top_p ->
 * exp(log_softmax).  rows are `ld_*` elements apart.  err_flag[row] (device int, may be NULL)
 * is set to 1 where the reference would raise 'norm logits error' (NaN/Inf result).
 * bf16_round_logits != 0 rounds each fp32 logit to bf16 first (a bf16 lm_head whose output the
 * reference then casts with .float(), modeling_llama.py:869-870). */
int sd_norm_probs(const float *logits, int rows, int V, long ld_in, float temperature, int top_k,
                  float top_p, int bf16_round_logits, float *probs_out, long ld_out, int *err_flag,
                  void *workspace, void *stream);
/* workspace (device, sd_norm_workspace_bytes(rows) bytes, may be NULL): with it and 1 <= top_k <= 64 each row is first
 * cut over 16 workgroups that extract the candidates above a safe threshold, so one 128 KiB row is not limited by what a
 * single CU can pull (~25 GB/s); without it one workgroup does everything.  Results are identical either way. */
size_t sd_norm_workspace_bytes(int rows);

/* top_k_top_p_filter by itself (utils.py:152-179; the reference mutates its argument and returns it): out[i] = logit
 * where the token is kept, -inf where it is dropped; top_k == 0 and top_p == 0 leave the row untouched.  `out` must
 * not alias `logits` (the Python drop-in copies the result back into its argument).  dtype_mode: 0, or SD_NORM_DT_BF16 /
 * SD_NORM_DT_F16 when the caller's tensor is 16-bit - the reference then sorts, softmaxes and cumsums in that dtype
 * (utils.py:170-172), so the kept set near the top-p cut is decided on 16-bit sums, exactly as sd_norm_probs decides it. */
int sd_topk_topp_filter(const float *logits, int rows, int V, long ld_in, int top_k, float top_p, int dtype_mode,
                        float *out, long ld_out, void *stream);

/* One draft / autoregressive step's tail fused: norm_logits of ONE row followed by sample() on it
 * (kvcache_model.py:235-236 + :283), a single launch.  Writes the probability row (the accept scan and
 * the residual need it later) and the sampled token.  exp_noise / Philox as in sd_sample; with device
 * Philox only the surviving tokens draw a variate.  sample_err as sd_sample's err_flag. */
int sd_norm_sample(const float *logits, int V, float temperature, int top_k, float top_p,
                   int bf16_round_logits, float *probs_out, int *err_flag, const float *exp_noise,
                   uint64_t philox_seed, uint64_t draw_index, int *tok_out, int *sample_err, void *workspace,
                   void *stream);

/* Batched form of sd_norm_probs / sd_norm_sample for stream-batched decode: row r of `logits` is normalised into
 * rows[r].probs_out (rows of different streams live in different arenas); with sample != 0 each row also draws its
 * token into rows[r].tok_out from its own exp_noise row or Philox (seed, draw).  `rows` is a host array. */
typedef struct {
    float *probs_out;
    int *err;                 /* device int or NULL */
    const float *exp_noise;   /* device row or NULL (Philox) */
    uint64_t philox_seed, draw_index;
    int *tok_out, *sample_err;
} sd_norm_row;
int sd_norm_batch(const float *logits, int n_rows, int V, long ld_in, float temperature, int top_k, float top_p,
                  int bf16_round_logits, const sd_norm_row *rows, int sample, void *workspace, void *stream);

/* sample (utils.py:213-233) for num_samples == 1: argmax_i probs[i] / noise[i] (first index wins
 * ties), then the "< 1e-9 -> argmax(probs)" fix-up.  exp_noise is a device row of Exp(1) variates
 * in the reference's draw order (parity mode), or NULL to draw them on the device from Philox
 * (seed, draw_index).  tok_out: device int32.  err_flag (device, may be NULL): 1 = invalid
 * distribution (negative / NaN / Inf), 2 = all-zero row; both are 'prob error' in the reference. */
int sd_sample(const float *probs, int V, const float *exp_noise, uint64_t philox_seed,
              uint64_t draw_index, int *tok_out, int *err_flag, int dtype_mode, void *stream);

/* The device RNG of the throughput mode made observable, so that a test can replay the exact variates into the CPU
 * oracle (the reference draws from torch's generator, utils.py:221 / speculative_sampling.py:1978; this build's
 * device mode draws from counter-based Philox4x32-10 instead).  out[i] = the Exp(1) variate the sampling kernels use
 * for vocabulary element i of draw (seed, draw_index): -log(u), u = (23 random bits + 1/2) * 2^-23 in (0,1). */
int sd_philox_exp(uint64_t philox_seed, uint64_t draw_index, int V, float *out, void *stream);
/* out[i] = the uniform in [0,1) the accept scans use for draw (seed, draw_index + i), i < n. */
int sd_philox_uniform(uint64_t philox_seed, uint64_t draw_index, int n, float *out, void *stream);

/* max_fn (utils.py:236-245) materialised: out = max(p-q,0) / (sum(max(p-q,0)) + 1e-6).  q may be
 * NULL (then max_fn(p)).  Only the drop-in sampling.utils.max_fn uses this; the decode loop uses
 * the fused sd_resample below and never writes the residual row. */
int sd_max_fn(const float *p, const float *q, int V, float *out, int dtype_mode, void *stream);

/* Result block written by the accept / resample kernels (device or pinned-host memory). */
typedef struct {
    int32_t n_accepted;    /* l: drafted tokens accepted this iteration   (speculative_sampling.py:1974-1991) */
    int32_t n;             /* last kept position, L+l-1                    (:1964, :1983)                      */
    int32_t next_token;    /* t: residual resample or bonus sample         (:2005-2023)                        */
    int32_t flags;         /* bit0 residual was all-zero -> fallback sample(max_fn(p_n)) (:2009-2010);
                              bit1 'prob error'; bit2 all gamma accepted; bit3 an error word of the iteration was set */
    float p_at[16];        /* target prob of each drafted token (for the acc_rate statistic, :1966-1971)      */
    float q_at[16];        /* draft prob of each drafted token                                                */
    int32_t drafted[16];   /* the gamma drafted token ids seq[L .. L+gamma) (saves a second device->host copy)      */
} sd_accept_result;

/* Accept scan (speculative_sampling.py:1964-1991): for i < gamma, j = seq[L+i]; reject iff
 * r[i] > float32(double(p_hist[L+i-1][j]) / double(q_hist[L+i-1][j])); the first reject decides.
 * p_hist / q_hist are probability arenas indexed by absolute position (row stride ld).  r: gamma
 * device floats in the reference's draw order (parity mode; the caller re-aligns the host generator
 * to the min(l+1, gamma) uniforms the reference would have consumed), or NULL for device Philox
 * (seed, draw_index + i).  Writes n_accepted, n, p_at, q_at, flags bit2; next_token = -1. */
int sd_accept_scan(const float *p_hist, const float *q_hist, long ld, const int32_t *seq, int L,
                   int gamma, const float *r, uint64_t philox_seed, uint64_t draw_index,
                   sd_accept_result *out, void *stream);

/* Residual resample / bonus sample (speculative_sampling.py:2005-2023) at position res->n, read
 * from the device result block: rejected -> sample(max_fn(p_n - q_n)) with the all-zero fallback
 * sample(max_fn(p_n)); all accepted -> sample(p_last).  Writes next_token and flags, appends the
 * token at seq[n+1] and stores the new sequence length n+2 into *seq_len (device int, may be NULL). */
int sd_resample(const float *p_hist, const float *q_hist, long ld, int V, int32_t *seq, int L,
                int gamma, const float *exp_noise, uint64_t philox_seed, uint64_t draw_index,
                sd_accept_result *res, int32_t *seq_len, int dtype_mode, void *stream);

/* sd_accept_scan + sd_resample in ONE launch whose sample works on candidate lists instead of two passes over V.
 * sd_norm_probs_lists is sd_norm_probs that also leaves, per row, the row's non-zero entries (token ids + probabilities;
 * sd_cand_list_bytes(rows) device bytes; a row gets a list when 1 <= top_k <= 64 put it through the candidate path and it
 * holds <= 128 entries, else it is marked list-less).  sd_accept_resample takes the lists of the gamma + 1 TARGET rows of
 * the iteration (positions L-1 .. L+gamma-1, in order) or NULL; results are bit-identical to the two-kernel form, which
 * it falls back to for a list-less row.  Philox only (seed, draw_scan + i for the uniforms unless r is given; draw_resample). */
size_t sd_cand_list_bytes(int rows);
int sd_norm_probs_lists(const float *logits, int rows, int V, long ld_in, float temperature, int top_k, float top_p,
                        int bf16_round_logits, float *probs_out, long ld_out, int *err_flag, void *workspace,
                        void *cand_lists, void *stream);
int sd_accept_resample(const float *p_hist, const float *q_hist, long ld, int V, int32_t *seq, int L, int gamma,
                       const float *r, uint64_t philox_seed, uint64_t draw_scan, uint64_t draw_resample,
                       sd_accept_result *res, const int *err_flags, int n_err, int dtype_mode, const void *target_lists,
                       void *stream);

/* sd_accept_scan + sd_resample for up to 16 independent streams in two launches (stream-batched decode).
 * Per item: its probability arenas, token buffer, prefix length L, the gamma uniforms r (or NULL: Philox
 * (philox_seed, draw_scan + i)), the resample noise row (or NULL: Philox (philox_seed, draw_resample)), the
 * device result block, and optionally n_err device error words folded into res.flags bit3. `items` is a host array. */
typedef struct {
    const float *p_hist, *q_hist;
    int32_t *seq;
    int32_t L;
    const float *r, *exp_noise;
    uint64_t philox_seed, draw_scan, draw_resample;
    sd_accept_result *res;
    const int *err_flags;
    int32_t n_err;
} sd_accept_item;
int sd_accept_batch(const sd_accept_item *items, int n_items, long ld, int V, int gamma, int dtype_mode, void *stream);

/* Width-w acceptance of multi_speculative_sampling(strategy="iid") (speculative_sampling.py:1592-1640, SURVEY.md 8(f)
 * rank 2): replica w drafted seq_w[L .. L+gamma); replicas are scanned in order, replica w accepts its i-th token iff
 * r < min(1, p_w[L+i-1][j] / q_w[L+i-1][j]) (fp32 division; a NaN ratio rejects) and stops at its first reject; the
 * replica with the longest accepted run wins (first one on ties; replica 0 when nothing was accepted) and the scan
 * ends early at the first replica that accepts all gamma.  r: width*gamma device floats consumed strictly in that
 * order (parity mode; n_uniform tells the caller how many the reference would have drawn), or NULL for device Philox
 * (seed, draw_index + k).  `chosen` is filled like sd_accept_scan's result for the winning replica and feeds
 * sd_multi_resample; p_at/q_at hold every replica's gathered probabilities ([w][i], 16 columns per replica) for the
 * acc_rate statistic (:1592-1601).  `items` is a host array of width <= 16 entries. */
typedef struct {
    sd_accept_result chosen;
    int32_t choice, n_uniform, width, gamma;
    float p_at[256], q_at[256];
} sd_multi_result;
typedef struct {
    const float *p_hist, *q_hist;     /* probability arenas of replica w, indexed by absolute position (row stride ld) */
    const int32_t *seq;               /* replica w's token buffer */
} sd_multi_item;
int sd_accept_multi(const sd_multi_item *items, int width, long ld, int L, int gamma, const float *r,
                    uint64_t philox_seed, uint64_t draw_index, sd_multi_result *out, void *stream);

/* sd_resample with multi_speculative_sampling's fallback (speculative_sampling.py:1662-1668): when the residual
 * sample raises, the token is drawn from p_n itself, not from max_fn(p_n).  Everything else as sd_resample. */
int sd_multi_resample(const float *p_hist, const float *q_hist, long ld, int V, int32_t *seq, int gamma,
                      const float *exp_noise, uint64_t philox_seed, uint64_t draw_index, sd_accept_result *res,
                      int dtype_mode, void *stream);

/* ------------------------------------------------------------------------------------------
 * Decoder model + KV arena           reference sampling/models/modeling_{llama,opt}.py,
 *                                    sampling/kvcache_model.py:141-252 (forward), :359-436 (rollback)
 * ------------------------------------------------------------------------------------------ */

typedef struct {
    int32_t arch;                 /* sd_arch                                                    */
    int32_t dtype;                /* sd_dtype of weights and activations                        */
    int32_t vocab, hidden, inter, n_layers, n_heads, n_kv_heads, head_dim;
    int32_t max_pos;              /* rows of the rope table (llama) / learned positions (opt)   */
    int32_t opt_pre_ln;           /* OPT do_layer_norm_before (125m/13b: 1, 350m: 0)            */
    int32_t opt_proj_dim;         /* OPT word_embed_proj_dim (== hidden when no project_in/out) */
    float norm_eps;
    int32_t logits_bf16_round;    /* round logits to bf16 before use (bf16 lm_head semantics)   */
    int32_t fused_layout;         /* bf16 only: wqkv / w_gate_up rows are in the fused-epilogue order below      */
} sd_model_config;

/* Weight table.  Matrices are [out][in] as in nn.Linear.  For dtype SD_BF16 every GEMM matrix is
 * handed over in the tile-packed layout produced by sd_pack_weight_bf16 ("HBM
 * layout"); for SD_F32 they stay row-major.  Vectors / embeddings are row-major in `dtype`.
 * Per-layer arrays have n_layers entries.  Unused entries are NULL.
 * With cfg.fused_layout (bf16): the QKV and gate/up GEMMs keep the whole k-range in one workgroup and finish with
 * a fused epilogue (RoPE + in-place KV append; SiLU(gate)*up / ReLU), which needs two row orders prepared at load:
 *   llama wqkv: inside each q and k head the rows are pair-interleaved, d0, d0+D/2, d1, d1+D/2, ... (v heads natural);
 *   llama w_gate_up: 8 gate rows then the same 8 up rows, repeated: g0..g7,u0..u7,g8..g15,u8..u15,...
 * OPT keeps the natural order. */
typedef struct {
    const void *embed;            /* [vocab][embed_dim] row-major                               */
    const void *pos_embed;        /* OPT: [max_pos+2][hidden]                                   */
    const void *project_in;       /* OPT 350m: [hidden][proj]   (GEMM matrix)                   */
    const void *project_out;      /* OPT 350m: [proj][hidden]   (GEMM matrix)                   */
    const void *final_norm_w, *final_norm_b;
    const void *lm_head;          /* [vocab][embed_dim]         (GEMM matrix)                   */
    const void *rope_cos, *rope_sin; /* llama: [max_pos][head_dim/2] in `dtype`                 */
    const void *const *wqkv;      /* [(n_heads+2*n_kv_heads)*head_dim][hidden], q rows then k then v */
    const void *const *bqkv;      /* OPT biases, same order                                     */
    const void *const *wo;        /* [hidden][hidden]                                           */
    const void *const *bo;
    const void *const *w_gate_up; /* llama: [2*inter][hidden], gate rows then up rows; OPT fc1: [inter][hidden] */
    const void *const *b_fc1;
    const void *const *w_down;    /* [hidden][inter] (OPT fc2)                                  */
    const void *const *b_fc2;
    const void *const *norm1_w, *const *norm1_b;  /* input_layernorm / self_attn_layer_norm      */
    const void *const *norm2_w, *const *norm2_b;  /* post_attention_layernorm / final_layer_norm */
} sd_model_weights;

typedef struct sd_model sd_model;

/* Copies the config and the pointer table into a host handle (weights stay where they are). */
int sd_model_create(const sd_model_config *cfg, const sd_model_weights *w, sd_model **out);
int sd_model_destroy(sd_model *m);
/* Rows one sd_session_forward call may carry for this model: 256 (prefill chunks; the many-row GEMM kernel needs every
 * per-layer weight matrix to have N % 128 == 0 and K % 64 == 0, K >= 512) or 64.  sd_session_create and
 * sd_session_scratch_bytes clamp their max_rows to it. */
int sd_model_max_rows(const sd_model *m);

/* Repack a row-major bf16 [N][K] matrix (N % 16 == 0, K % 32 == 0) into the streaming layout the
 * GEMM kernels read: 1 KiB tiles [N/16][K/32][lane 0..63][8 bf16], lane = 16*(k/8 % 4) + n % 16. */
int sd_pack_weight_bf16(const void *w_rowmajor, void *w_packed, int N, int K, void *stream);

/* Row-major bf16 activations [M][K] (K % 32 == 0) -> the GEMM operand layout the engine keeps them in between
 * kernels: 16-row tiles [ceil(M/16)][K/32][lane 0..63][8 bf16], lane = 16*(k/8 % 4) + m % 16, so that a wave's MFMA
 * fragment is one contiguous 1 KiB read.  x_tiled must hold roundup(M,16) * K elements. */
int sd_pack_activation_bf16(const void *x_rowmajor, void *x_tiled, int M, int K, void *stream);

/* The weight-streaming GEMM on its own (unit tests, kernel-level roofline runs):
 * part[s][m][n] = sum over k-slice s of x[m][k] * W[n][k] for a tile-packed bf16 W, then (if out != NULL)
 * out[m][n] = sum_s part[s][m][n] in fp32.  M <= 64 (<= 256 with x_tiled and N % 128 == 0: from 33 rows on the
 * LDS-tiled kernel runs).  x: plain rows (x_tiled == 0) or sd_pack_activation_bf16's
 * layout (x_tiled != 0, what the forward uses).  part must hold splits * roundup(M,16) * N floats; the split count the
 * policy chose comes back in *splits_out. */
int sd_gemm_bf16(const void *w_packed, const void *x, int x_tiled, int M, int N, int K, float *part,
                 size_t part_floats, float *out, int *splits_out, void *stream);

/* A session = one KV arena + scratch for one sequence (one KVCacheModel of the reference).
 * kv_arena: [n_layers][2][n_kv_heads][max_seq][head_dim] in `dtype`, caller-allocated.
 * scratch: sd_session_scratch_bytes() bytes, caller-allocated. */
typedef struct sd_session sd_session;
size_t sd_session_kv_bytes(const sd_model *m, int max_seq);
size_t sd_session_scratch_bytes(const sd_model *m, int max_rows);
int sd_session_create(sd_model *m, int max_seq, int max_rows, void *kv_arena, void *scratch,
                      sd_session **out);
int sd_session_destroy(sd_session *s);
/* fp8 KV arena (BASELINE config 5): the arena then holds OCP e4m3 bytes, [n_layers][2][n_kv_heads][max_seq][head_dim],
 * half the bytes of a 16-bit arena; `scales` = device floats [n_layers][2][n_kv_heads], an element x is stored as
 * fp8(x / scale).  K / V rows are quantised where they are appended (the QKV epilogue, after RoPE), the attention kernel
 * widens them back in registers.  16-bit models with head_dim >= 32; call before the first forward. */
int sd_session_set_kv_fp8(sd_session *s, const float *scales);

/* One model forward over n_new tokens at absolute positions pos0 .. pos0+n_new-1, appending their
 * K/V rows into the arena in-kernel (replaces the per-layer torch.cat of modeling_llama.py:337-338 /
 * modeling_opt.py:192-193 and kvcache_model.py:214).  tokens: device int32, read at tokens[0..n_new).
 * Logits (fp32) are produced for the LAST n_logits of the new rows into logits_out[n_logits][vocab]
 * (row stride ld_logits); n_logits == 0 skips the head.  Rollback is O(1): the caller just passes a
 * smaller pos0 next time (kvcache_model.py:359-436).  n_new <= max_rows, pos0+n_new <= max_seq. */
int sd_session_forward(sd_session *s, const int32_t *tokens, int n_new, int pos0, int n_logits,
                       float *logits_out, long ld_logits, void *stream);

/* Tree verify (reference kvcache_model.py:38-136 forward_tree_attention with the extra attention
 * mask of modeling_llama.py:684-689 / modeling_opt.py:660-667): ONE target forward over the n <= 64 nodes of a draft token
 * tree.  tokens / positions / masks are DEVICE-side for tokens (int32[n]) and HOST-side for positions and masks:
 * node i sits at position positions[i] (its depth; RoPE / learned position), attends to all `base_len` cached positions and
 * to the nodes j with bit j of masks[i] set (its ancestors and itself, j <= i), and its K / V rows are appended at arena
 * slot base_len + i.  Logits of all n nodes go to logits_out[n][vocab].  sd_session_compact_kv afterwards moves the kept
 * path's rows together (reference kvcache_model.py:326-353 rollback_tree_attention): slot base_len + idx[j] -> base_len + j
 * (idx: device int32[k], ascending) in every layer and head. */
int sd_session_forward_tree(sd_session *s, const int32_t *tokens, const int32_t *positions, const uint64_t *masks, int n,
                            int base_len, float *logits_out, long ld_logits, void *stream);
int sd_session_compact_kv(sd_session *s, int base_len, const int32_t *idx_dev, int k, void *stream);

/* Single-stream decode / verify forwards of a 16-bit model run two kinds of launches whose workgroups wait for each other
 * on device-side counters: attention + O projection (csrc/fused_kernels.h) and the k-split down projection with the
 * residual epilogue (csrc/normload_kernels.h).  Every such wait is bounded (20 ms); one that runs into its limit poisons
 * its output rows with NaN - the sampler then reports the reference's 'norm logits error' (utils.py:186-188), never
 * finite numbers - and sets a sticky bit in the session's status word: bit 0 attention + O, bit 1 down projection.
 * sd_session_fused_status reads and clears the word (blocking copy; the session's stream must be idle); 0 = no wait has
 * timed out since the last read.  After a non-zero read, and after any forward that returned an error, the next forward
 * first re-zeroes the counters (stream-ordered), so one failed launch cannot leave every later wait short. */
int sd_session_fused_status(sd_session *s, unsigned *status_out);
/* TEST HOOK: the fused launches of the NEXT forward of this session expect `extra` (0..1024) arrivals more than will come,
 * i.e. every one of their waits times out.  Exists so that the timeout branch is covered by a test (tests/); the forward
 * after that one is back in step. */
int sd_session_test_skew_wait(sd_session *s, int extra);
/* Debugging aid: the per-workgroup records the last fused attention + O-projection launch left when the session was
 * created under SD_AO_STAMPS=1 (fused_kernels.h, AO_STAMP_WGS): out[w][8] long long for workgroups w < n_wgs <= 512 -
 * wall_clock64 (100 MHz) at the workgroup's milestones, XCC_ID / HW_ID, and role / head / row group. */
int sd_session_ao_stamps(sd_session *s, long long *out, int n_wgs);

/* Stream-batched forward: the new rows of up to 16 independent sequences share ONE pass over the
 * weights (same bytes streamed, n_items times the tokens).  Each item names its own session (KV arena), its token
 * buffer `seq` (device int32 indexed by ABSOLUTE position: the rows read seq[pos0 .. pos0+n_new)), its cache length
 * pos0, its number of new rows and how many of its last rows need logits.  Activations use items[0].session's
 * scratch (total rows <= its max_rows, <= 64); all sessions must belong to the same model.  Logit rows come out packed
 * in item order into logits_out.  sd_session_forward is the one-item case of this call. */
typedef struct {
    sd_session *session;
    const int32_t *seq;
    int32_t pos0, n_new, n_logits;
} sd_batch_item;
int sd_batch_forward(const sd_batch_item *items, int n_items, float *logits_out, long ld_logits, void *stream);
/* Batched prefill: the items' rows (each stream's n_new rows at positions pos0.. , a contiguous run; n_logits must be 0) go
 * through ONE pass over the weights - up to 256 rows and 32 attention groups (8 consecutive rows of one stream) in all.
 * K / V rows are appended to every stream's own arena; nothing else comes back.  All sessions share one model. */
int sd_batch_prefill(const sd_batch_item *items, int n_items, void *stream);

/* One whole speculative iteration enqueued natively (device-RNG mode), reference speculative_sampling.py:1934-2031:
 * gamma x (draft forward over the uncached rows + sd_norm_sample straight into seq[]), one target forward over its
 * uncached rows + sd_norm_probs of the last gamma+1, sd_accept_scan, residual / bonus sample, then async copies of
 * the result block and of seq[L .. L+gamma+2) into pinned host memory.  Nothing synchronises: the caller waits on
 * the stream once per iteration.  seq: device int32 sequence buffer; q_hist / p_hist: probability arenas indexed by
 * position (row stride ld); *_logits: scratch rows for the heads; err_words: 3*gamma+1 device ints (folded into
 * res.flags bit3 = 'norm logits error' / sample error).  Philox draws: draft step i uses (seed_draft,
 * draw_draft0+i); the accept uniforms (seed_accept, draw_scan0+i) unless r_const (gamma device floats: the
 * random_seed quirk) is given; the resample (seed_accept, draw_resample). */
typedef struct sd_spec sd_spec;
int sd_spec_create(sd_session *draft, sd_session *target, int gamma, float temperature, int top_k, float top_p,
                   int32_t *seq, float *q_hist, float *p_hist, long ld, float *draft_logits, long ld_draft_logits,
                   float *target_logits, long ld_target_logits, int *err_words, sd_accept_result *res_dev,
                   void *norm_workspace /* sd_norm_workspace_bytes(gamma+1) bytes or NULL */, sd_spec **out);
int sd_spec_destroy(sd_spec *sp);
int sd_spec_iteration(sd_spec *sp, int L, int draft_len, int target_len, uint64_t seed_draft, uint64_t draw_draft0,
                      uint64_t seed_accept, uint64_t draw_scan0, uint64_t draw_resample, const float *r_const,
                      sd_accept_result *res_host, int32_t *tok_host /* gamma+2 ids from seq[L], may be NULL */,
                      void *stream);
/* The whole loop of speculative_sampling.py:1934-2046 for the device-RNG mode: iterations (sd_spec_iteration + one stream
 * wait each) until the sequence holds T tokens, a new EOS appears (more than ori_eos_cnt of them in total) or an error
 * word is set - no Python between iterations.  host_seq (host int32, capacity >= T + gamma + 1) holds the *len_io tokens so
 * far and receives the new ones; *seed_io / *draw_io are the Philox stream position (advanced exactly as the iteration's
 * draw order prescribes: gamma draft draws, the discarded target sample, gamma uniforms - or, with random_seed != 0, the
 * stream restarted at (random_seed, 0) - and the resample); *draft_len_io / *target_len_io the cache lengths.  Per
 * iteration i < *n_iters_out: acc_len_out[i]; p_at_out / q_at_out[i * gamma ..] (the accept ratios' operands);
 * draft_ms_out / target_ms_out[i] when timing is on (else untouched; any of the five may be NULL).  *err_out: 0, or
 * 1 = 'prob error' (sample / resample), 2 = 'norm logits error'.  res_host: pinned host memory for the result block. */
int sd_spec_generate(sd_spec *sp, int32_t *host_seq, int *len_io, int T, int eos_token_id, int ori_eos_cnt,
                     uint64_t *seed_io, uint64_t *draw_io, uint64_t random_seed, const float *r_const, int *draft_len_io,
                     int *target_len_io, sd_accept_result *res_host, int max_iters, int32_t *acc_len_out, float *p_at_out,
                     float *q_at_out, float *draft_ms_out, float *target_ms_out, int *n_iters_out, int *err_out,
                     void *stream);
/* The stream-batched loop (throughput mode) without the interpreter between iterations: up to 16
 * independent streams decode in lock-step - every draft step one sd_batch_forward + sd_norm_batch over the active
 * streams, every verify one target pass per max_rows_per_forward / (gamma + 1) streams, then sd_accept_batch, one copy of
 * the result blocks and one wait - until every stream holds T tokens or has produced a new EOS.  Per stream: the two
 * sessions, the device token buffer / probability arenas / error words, its device and pinned-host result block (the
 * streams' blocks must be consecutive: one copy moves them all), the host token buffer and the in/out loop state
 * (lengths, cache lengths, Philox position); per-iteration statistics as in sd_spec_generate.  Each stream runs exactly
 * the algorithm of sd_spec_generate with its own Philox stream.  verify_ms_out / verify_streams_out / verify_ctx_out
 * (host, max_iters_log entries, may be NULL): time of each iteration's verify passes, streams in it, their mean context.
 * *err_out: 1 when a stream hit a sampling / normalisation error (the reference raises).
 * norm_workspace (device, may be NULL): sd_norm_workspace_bytes(max_rows_per_forward) bytes - the candidate rows of a pass and,
 * behind them, one candidate list per row.  With it the loop runs the single-stream loop's sampling tail (since round 4):
 * the draft head leaves tile maxima and clears the streams' probability rows, the target rows come back with candidate
 * lists, accept scan + residual / bonus sample are one launch on them - bit-equal to the dense kernels named above, which
 * remain the path without a workspace, with more streams than one verify pass holds, and under SD_BATCH_FUSED_TAIL=0. */
typedef struct {
    sd_session *draft, *target;
    int32_t *seq;
    float *q_hist, *p_hist;
    int *err_words;
    sd_accept_result *res_dev, *res_host;
    int32_t *host_seq;
    int32_t len, T, ori_eos_cnt, draft_len, target_len;
    uint64_t seed, draw;
    int32_t done, calls;
    int32_t *acc_len_out;
    float *p_at_out, *q_at_out;
} sd_batch_stream;
int sd_spec_batch_generate(sd_batch_stream *streams, int n_streams, int gamma, float temperature, int top_k, float top_p,
                           int V, long ld, int eos_token_id, uint64_t random_seed, const float *r_const,
                           int draft_norm_mode, int target_norm_mode, float *draft_logits, long ld_draft_logits,
                           float *target_logits, long ld_target_logits, void *norm_workspace, int max_rows_per_forward,
                           float *verify_ms_out, int32_t *verify_streams_out, float *verify_ctx_out, int max_iters_log,
                           int *n_iters_out, int *err_out, void *stream);
/* HIP-event timing of the draft phase and the target (verify) phase of the last iteration, on the launch stream. */
int sd_spec_timing(sd_spec *sp, int on);
int sd_spec_last_times(sd_spec *sp, float *draft_ms, float *target_ms);

/* ------------------------------------------------------------------------------------------
 * Tensor parallelism of one decoder over the GPUs of a node (BASELINE config 5: Llama-2-70b, TP = 8 over xGMI).
 * The reference is single-process: this is new capability behind the same call signatures.
 * A shard is an ordinary sd_model whose config holds the LOCAL head / KV-head / MLP counts (n_heads * head_dim <= hidden)
 * and whose matrices are the Megatron slices: wqkv / w_gate_up by output rows, wo / w_down by input columns.  A session
 * bound to a group all-reduces the fp32 partial outputs of wo and w_down (two per layer) before the residual add.
 * Every rank runs the same decode loop on the same tokens; draft model and sampling are replicated.
 * ------------------------------------------------------------------------------------------ */
typedef struct sd_tp sd_tp;
/* RCCL group: rank 0 obtains a 128-byte id and the host broadcasts it (torch.distributed / any side channel); every
 * rank then joins.  RCCL is resolved with dlopen on first use (the copy the process already holds, if any). */
int sd_tp_unique_id(void *id128);
int sd_tp_create_rccl(int rank, int world, const void *id128, sd_tp **out);
/* Loopback group: `world` ranks inside ONE process on one GPU (each driven by its own host thread and stream; the
 * all-reduce is an event-ordered rendezvous + a sum kernel).  For testing the sharded forward on a one-GPU box. */
int sd_tp_create_loopback(int world, sd_tp **out /* world handles */);
int sd_tp_destroy(sd_tp *t);
int sd_session_set_tp(sd_session *s, sd_tp *t);

/* ------------------------------------------------------------------------------------------
 * Throughput-mode gather: prompt streams are sharded over the ranks (stream s on rank s mod
 * world) and never talk inside the decode loop; at the end every rank contributes its packed token rows
 * [rows][width] int32 (-1 padded) and receives all ranks' rows, [world][rows][width], through ONE ncclAllGather over
 * RCCL / xGMI (KB-scale: latency-bound).  The reference has no counterpart (single process, evaluation.py:524-532 reads
 * the output of its one stream); `sd_comm_unique_id` is ncclGetUniqueId, whose 128 bytes the host hands to every rank
 * (torch.distributed's store / broadcast, or any side channel).  send / recv are device buffers of this rank's GPU.
 * ------------------------------------------------------------------------------------------ */
typedef struct sd_comm sd_comm;
int sd_comm_probe(void);    /* SD_OK when RCCL resolves in this process (dlopen + the five entry points); creates nothing */
int sd_comm_unique_id(void *id128);
int sd_comm_init(int rank, int world, const void *id128, sd_comm **out);
int sd_comm_all_gather_tokens(sd_comm *c, const int32_t *send, int32_t *recv, int rows, int width, void *stream);
int sd_comm_rank(const sd_comm *c, int *rank_out, int *world_out);
int sd_comm_destroy(sd_comm *c);

/* Per-op-class timing for the roofline report: when enabled, every launch inside
 * sd_session_forward is bracketed with HIP events on the launch stream; sd_profile_read
 * synchronises the stream, returns accumulated milliseconds and launch counts per class, and
 * resets them.  Classes: 0 gemm, 1 attention, 2 norm/residual epilogues, 3 qkv-rope-append,
 * 4 activation, 5 embed, 6 logits epilogue. */
#define SD_N_PROFILE_CLASSES 8
int sd_profile_enable(sd_session *s, int on);
int sd_profile_read(sd_session *s, float *ms_out, int *count_out);

#ifdef __cplusplus
}
#endif
#endif /* SPECDEC_H */
