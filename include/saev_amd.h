/*
 * saev_amd.h — C ABI of libsaev_amd.so: the MI355X (gfx950) TopK-SAE train-step path.
 *
 * This is the drop-in boundary for the hot path of OSU-NLP-Group/saev.  The reference has no FFI of
 * its own (it is pure PyTorch); each entry point below names the reference code it replaces
 * (paths relative to the reference root, src/saev/...).  The Python host in saev_amd/ binds these
 * with ctypes (see INTEGRATION.md for the stub a saev maintainer would add).
 *
 * Conventions
 *   - plain pointers + sizes; all pointers are DEVICE pointers unless the name ends in _host;
 *   - every call returns 0 or a negative saev_status; saev_last_error() gives the message;
 *   - all work is enqueued on the hipStream_t passed as `stream` (void*; NULL = default stream);
 *     nothing synchronises unless documented;
 *   - the library never frees caller memory; scratch is owned by the context;
 *   - a context is bound to one device and is not thread-safe.
 *
 * Layout of the flat parameter-sized buffers (params, grads, Adam m, Adam v), fp32, in
 * state_dict order (nn/modeling.py:312-327):
 *   [ W_dec (d_sae x d_model, row-major) | b_dec (d_model) | W_enc (d_model x d_sae) | b_enc (d_sae) ]
 */
#ifndef SAEV_AMD_H
#define SAEV_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SAEV_AMD_ABI_VERSION 10

typedef enum {
    SAEV_OK = 0,
    SAEV_INVALID_ARG = -1,
    SAEV_HIP_ERROR = -2,
    SAEV_UNSUPPORTED = -3,
    SAEV_NOT_BOUND = -4,
    SAEV_RCCL_ERROR = -5,
    SAEV_STALE_PARAMS = -6 /* W_enc was written outside the library without saev_params_touched and a step has already run on
                              stale operand images (PARAMETER OWNERSHIP below); the context itself recovers */
} saev_status;

/* Static configuration of one SAE (nn/modeling.py:259-284 SparseAutoencoderConfig, :119-130 TopK,
 * :66-73 AuxK; nn/objectives.py:13-25 Matryoshka). */
typedef struct {
    int32_t d_model;
    int32_t d_sae;
    int32_t top_k;
    int32_t k_aux;                 /* 0 = no auxiliary loss (NoAux)                         */
    float alpha;                   /* AuxK scale                                            */
    int64_t dead_threshold_tokens; /* objectives.py:24                                      */
    int32_t normalize_w_dec;       /* modeling.py:283                                       */
    int32_t remove_parallel_grads; /* modeling.py:281                                       */
    int32_t max_batch;             /* scratch is sized for this many activation rows        */
    int32_t encoder_mode;          /* SAEV_ENCODER_F32, _F16X3, _BF16 or _F16R              */
    int32_t aux_dead_cap;          /* largest dead set the dense AuxK buffers are sized for at saev_create (no allocation
                                      happens inside a healthy step); 0 = min(d_sae, max(4096, 8 k_aux)) -- 4 096 dead latents are
                                      2.3 GB of buffers at configs[1], d_sae would be 11.5 GB (37 GB at configs[3]).  A step
                                      that meets more dead latents than this grows the buffers (twice the need, at most d_sae) after its
                                      read-back of the count -- a device-wide allocation, once; pass d_sae to rule it out.       */
    int32_t shard_world;           /* 0 / 1: the flat buffers are exactly the layout above.  N > 1: each half of it,
                                      [W_dec | b_dec] and [W_enc | b_enc], is padded with zeros to N equal chunks (chunks
                                      of the first half are whole decoder rows) so that N data-parallel ranks can
                                      reduce-scatter the gradient, run the tail on 1/N each and all-gather the parameters
                                      (saev_tail_prepare / saev_tail_apply); see saev_layout.                        */
    int32_t bound_mode;            /* TopK candidate bounds of the fused fp16-image encoders (top_k <= 32):
                                      0 guaranteed bounds only (running minimum over group maxima);
                                      1 predicted bounds first -- each row's bound is mean + z * sigma of its
                                        pre-activations over a sample of the latents -- verified by the select stage
                                        (the k-th largest candidate found must reach every bound used for the row),
                                        with an automatic second launch on guaranteed bounds whenever a prediction
                                        fails; codes and values are the same either way (saev_bound_state reports z
                                        and how often the second launch was needed).                                 */
    int32_t max_backward_rows;     /* 0 = max_batch.  Larger: the backward may cover that many rows (saev_backward_override:
                                      the rows of ALL data-parallel ranks); sizes only the backward's pair order, slice-major
                                      copies and partial rows -- the forward's buffers (candidate lists, dense fallback,
                                      AuxK, ...) stay at max_batch, which is then the LOCAL batch.                      */
} saev_cfg;

/* Element offsets of the four tensors inside each flat buffer, its total length, and the per-rank chunk lengths of the
 * two halves (all in floats) for this configuration.  Without shard_world: off_W_dec 0, off_b_dec S*D, off_W_enc
 * S*D + D, off_b_enc 2*S*D + D, n_total 2*S*D + S + D. */
typedef struct {
    int64_t off_W_dec, off_b_dec, off_W_enc, off_b_enc, n_total, chunk_a, chunk_b;
} saev_layout_t;
int saev_layout(const saev_cfg* cfg, saev_layout_t* out);

/* Encoder arithmetic.  F32, F16X3 and F16R are fp32-accurate (error vs fp64 at the level of a native fp32 GEMM):
 *   F32   : v_mfma_f32_32x32x2_f32, exact fp32 products;
 *   F16X3 : operands split into fp16 hi+lo (22 significand bits), three v_mfma_f32_32x32x16_f16 per
 *           product pair, fp32 accumulate -- 16/3 of the F32 matrix rate;
 *   BF16  : x and W_enc rounded to bf16 (nearest even) for the encoder contraction only, one
 *           v_mfma_f32_32x32x16_bf16 per product, fp32 accumulate; bias, TopK, decode, losses, all
 *           gradients and Adam stay fp32 on the fp32 master weights (BASELINE.json configs[3]). */
#define SAEV_ENCODER_F32 0
#define SAEV_ENCODER_F16X3 1
#define SAEV_ENCODER_BF16 2
/*   F16R  : one v_mfma_f32_16x16x32_f16 per product on fp16-rounded operands as a FIRST PASS, run on activations
 *           centred on the batch mean (the bias absorbs mean * W_enc) and pre-scaled by device-side powers of two.  Its
 *           error is bounded per row b from the rounding errors the images ACTUALLY carry (their norms are measured by
 *           the passes that write the images), by Cauchy-Schwarz:
 *             E_b = 1.02 (||dx_b|| wmax + ||x_b - mean|| dwmax + ||dx_b|| dwmax)
 *                   + (1.05 d_model 2^-22 + 2^-17) ||x_b - mean|| wmax + sqrt(d_model) 2^-14 wmax / scale_x + 2^-23 max |bias|,
 *           wmax / dwmax = the largest column norm of W_enc / of its rounding error (select.hip: f16r_margin).
 *           Candidates are kept down to the running bound minus 2 E_b, and the select stage recomputes every
 *           survivor exactly in fp32 (dot product of the uncentred row with the fp32 encoder column + b_enc) before the
 *           final cut.  Codes and values are those of exact fp32 arithmetic; a dense h (saev_encode_dense, overflow
 *           route) always comes from the exact fp32 kernel.  Default of the Python host. */
#define SAEV_ENCODER_F16R 3

/* Scalars of one step (nn/objectives.py:57-89 MatryoshkaLoss + train.py:356-362 grad norm). */
typedef struct {
    float mse;
    float aux;
    float l0;
    float l1;
    float grad_norm; /* pre-clip global L2 norm                                             */
    float upper;     /* max |x| of the batch (objectives.py:227)                            */
    int32_t n_dead;
    int32_t n_overflow_rows; /* rows whose candidate list overflowed (step re-ran on the exact dense route) */
    int32_t cand_max;        /* longest per-row candidate list the fused encoder produced              */
    int32_t dense_route;     /* 1 when the step's codes came from the exact dense route (candidate-list or refinement
                                overflow, or k > 64), 0 when the fused route held                       */
    double sse;      /* sum (x - x_hat)^2 accumulated in fp64 (train.py:398-401, :561-562)  */
    double sum_sq;   /* sum x^2 in fp64 (train.py:383, :554)                                */
} saev_step_stats;

typedef struct saev_ctx saev_ctx;

/* Route switches for A/B measurements and for tests that must reach a particular kernel.  Every field 0 = the shipped
 * default; results are the same on every route (tests/test_gpu_parity.py, tests/test_gpu_dw_slices.py compare them).
 * The library itself reads no environment variable: the Python host maps its documented SAEV_AMD_* variables onto this
 * struct (saev_amd/engine.py: EngineConfig). */
typedef struct {
    int32_t struct_size;   /* sizeof(saev_debug_cfg) of the caller (fields past it read as 0)                            */
    int32_t dw_route;      /* weight gradients: 0 column slices out of the XCD L2s where the geometry allows, with the products
                              dval = <dL/dx_hat row, decoder row> left by the decode where the shape allows (top_k <= 32,
                              d_model 256 / 512 / 768 / 1024); 1 whole-row gathers (dw_rows) always; 2 column slices with dval
                              formed by their first pass (the only form for other shapes and for gathered backwards); 4 as 0, but
                              the decode itself gathers 32-column slices of W_dec out of the XCD L2s (a wash: DESIGN.md 3.3)  */
    int32_t enc_mfma;      /* single-product encoders: 0 v_mfma_f32_16x16x32 kernel, 32 the 32x32x16 kernel               */
    int32_t fused_chain;   /* f16r: 1 = survivor select, exact refinement and final select as ONE launch                  */
    int32_t ngroups;       /* TopK bound groups of the fp16-image encoders: 0 = 32 for top_k <= 32 (64 above), 64 forces
                              the 64-group bound                                                                          */
    int32_t enc_wgs;       /* workgroups the fused encoder's grid aims at (0 = 256, one per CU)                           */
    int32_t refresh_first; /* bound refresh on a workgroup's first N tiles (0 = 8) ...                                    */
    int32_t refresh_every; /* ... then on every M-th, M a power of two (0 = 2; the 64-group bound defaults to 1)          */
    int32_t aux_small_max; /* largest dead set the few-dead-latents AuxK kernels take: 0 = 128 where d_model % 128 == 0 (fp32-MFMA kernels; 64 with aux_wide_route = 1), else 40; -1 = never (dense algebra
                              whatever the count); values above 64 are clamped                                           */
    int32_t fwd_route;     /* exact refinement of the f16r encoder: 0 = from 32-column slices of W_enc^T that the XCD L2s hold
                              where the geometry allows (their D / 32 shares per survivor added by the final select), 1 = whole-row
                              gathers always, 2 = slices with a separate pass that adds the shares (round 4)              */
    int32_t dead_lag;      /* saev_step_dead sizes the auxiliary work from the tracker record of this many steps ago
                              (0 = 4, at most 8): shorter = tighter bound of the dead count, longer = more host run-ahead  */
    int32_t csc_route;     /* latent-major pair list of the backward: 0 = the training decode sets the (latent, row) bits of the
                              build's bit map while it holds the codes, 1 = the build's own fill pass always, 2 = as 0 with the
                              round-4 scan (two launches) instead of the single look-back scan                               */
    int32_t fin_route;     /* end of the column-slice backward: 0 = one launch; the projection coefficient comes from the pair lists
                              (<dW_dec[i], w_i> = sum val * dval) and ||w||^2 from normalize_rows, the decoder rows are not read,
                              1 = the round-4 kernels (dw_finalize, dw_finalize_cut, dw_clear_bitmap: three launches, every
                              gradient and decoder row read back)                                                           */
    int32_t prep_route;    /* f16r forward preparation: 0 = streamed where possible (one pass over x centred / scaled with what the
                              previous batch left, W_enc images left by the previous step's Adam), 1 = the full preparation on
                              every step (statistics, centring and both image passes from x and W_enc: the round-4 sequence)    */
    int32_t aux_dense_route; /* selection of the dense AuxK algebra: 0 = one launch leaves the code matrix, its mask, its maximum and its
                              operand scale (dead sets up to 4 096 columns), 1 = the round-4 sequence (radix select, two fills, scatter,
                              absmax, scale: six launches)                                                                    */
    int32_t aux_small_route; /* 9 ... 64 dead latents (and 1 ... 8 where the one-pass kernel does not take the shape), d_model % 128 == 0:
                              0 = the contractions as fp32 MFMA tiles (v_mfma_f32_32x32x2_f32), 1 = the vector-ALU kernels of rounds 3-4 */
    int32_t own_check;     /* PARAMETER OWNERSHIP, check (2): 0 = the fused Adam leaves / compares tile checksums of W_enc, 1 = off
                              (A/B measurements of the Adam launch only)                                                     */
    int32_t enc_rot;       /* fused encoder: 0 = the workgroups of an XCD that share a W_enc tile walk its k-steps rotated by one
                              step each, 1 = in lock step (same order: the tile's images are read by all of them at once)     */
    int32_t group_route;   /* several SAEs on the same batches (saev_share_x): 0 = the lender streams its preparation like a context on
                              its own and every member's fused Adam leaves its own W_enc images (from the third step of a group nobody
                              prepares anything from scratch), 1 = round 5: every member prepares from scratch on every step        */
    int32_t aux_split_route; /* dense AuxK route, operand images of the split-fp16 contractions: 0 = the codes, g_aux, x and the dead latents'
                              decoder rows written in BOTH operand forms by one pass each (six image launches), 1 = round 5: one launch
                              per form (ten)                                                                                  */
    int32_t aux_wide_route; /* few-dead-latents AuxK on the fp32 matrix cores: 0 = dead sets bounded by up to 128 (one launch per count window
                              [1, 32], [33, 64], [65, 128], the device-side count picks), 1 = round 5: up to 64, the dense algebra beyond */
} saev_debug_cfg;

int saev_abi_version(void);
const char* saev_last_error(const saev_ctx* ctx);

/* Lifetime.  `device` is the HIP device ordinal. */
int saev_create(const saev_cfg* cfg, int device, saev_ctx** out);
/* The same with route switches (dbg may be NULL = all defaults). */
int saev_create_ex(const saev_cfg* cfg, const saev_debug_cfg* dbg, int device, saev_ctx** out);
void saev_destroy(saev_ctx* ctx);

/* Borrow the caller's flat buffers (see layout above).  grads/adam_m/adam_v may be NULL for a
 * forward-only context.  Pointers must stay valid until re-bound or destroy. */
int saev_bind(saev_ctx* ctx, float* params, float* grads, float* adam_m, float* adam_v);

/* Dead-latent tracker state, (d_sae) int64, owned by the context (objectives.py:99,107-120).
 * Exposed so the host can read/seed it (it is not part of the checkpoint in the reference). */
int64_t* saev_toks_since_active(saev_ctx* ctx);
/* Per-latent "fired this step" flags, (d_sae) int32 0/1; in data-parallel runs the host
 * max-all-reduces this buffer between saev_step_forward and saev_step_dead. */
int32_t* saev_fired_flags(saev_ctx* ctx);
/* Let the host own the tracker state instead (both buffers d_sae long, zero-initialised by the
 * caller): lets a torch tensor alias them for all-reduce / inspection. */
int saev_bind_tracker(saev_ctx* ctx, int64_t* toks_since_active, int32_t* fired_flags);
/* Tell the context that the host wrote the tracker buffer (so dead latents may exist before
 * dead_threshold_tokens tokens have been processed). */
int saev_tracker_touched(saev_ctx* ctx);
/* Device copy of the current step's saev_step_stats (valid after the producing call completes). */
const saev_step_stats* saev_stats_device(saev_ctx* ctx);
/* Blocking read-back of the stats (synchronises `stream`). */
int saev_read_stats(saev_ctx* ctx, saev_step_stats* out_host, void* stream);

/* ---- single ops (API-compat surface of SparseAutoencoder) ---------------------------------- */

/* modeling.py:411-417  W_dec[i,:] /= ||W_dec[i,:]||  (no-op when cfg.normalize_w_dec == 0). */
int saev_normalize_w_dec(saev_ctx* ctx, void* stream);
/* modeling.py:343-347  h = x @ W_enc + b_enc, dense (n_rows x d_sae) output. */
int saev_encode_dense(saev_ctx* ctx, const float* x, int32_t n_rows, float* h_out, void* stream);
/* modeling.py:169-179  per-row top-k of a dense (n_rows x d_sae) matrix -> idx/val (n_rows x k),
 * unsorted.  `mask` (d_sae int32, may be NULL) restricts candidates to latents with mask != 0. */
int saev_topk_dense(saev_ctx* ctx, const float* h, int32_t n_rows, int32_t k, const int32_t* mask,
                    int32_t* idx_out, float* val_out, void* stream);
/* encode + TopK without materialising h (the fast path): idx/val (n_rows x top_k). */
int saev_encode_topk(saev_ctx* ctx, const float* x, int32_t n_rows, int32_t* idx_out, float* val_out,
                     void* stream);
/* scatter codes into a dense zero-initialised (n_rows x d_sae) matrix (f_x for API compat). */
int saev_scatter_dense(saev_ctx* ctx, const int32_t* idx, const float* val, int32_t n_rows, int32_t k,
                       float* f_out, void* stream);
/* modeling.py:351-409 with sparse input: x_hat[b, p, :] = b_dec + sum_{j: idx < prefixes[p]} val*W_dec[idx].
 * `prefixes_host` has n_prefixes ascending cut points ending at d_sae (NULL => one prefix). */
int saev_decode_sparse(saev_ctx* ctx, const int32_t* idx, const float* val, int32_t n_rows, int32_t k,
                       const int64_t* prefixes_host, int32_t n_prefixes, float* x_hats_out, void* stream);
/* modeling.py:419-445 on the bound grad buffer. */
int saev_remove_parallel_grads(saev_ctx* ctx, void* stream);
/* Row gather out of a device-resident activation pool (replaces ReservoirBuffer.get,
 * data/buffers.py:179-216): out[r,:] = pool[rows[r],:]. */
int saev_gather_rows(saev_ctx* ctx, const float* pool, const int64_t* rows, int32_t n_rows, float* out,
                     void* stream);

/* ---- the train step, in phases (framework/train.py:332-460) --------------------------------- */

/* Matryoshka prefix cut points for the following steps (objectives.py:125-138): n ascending latent counts ending at
 * d_sae, n <= 16; the loss is the mean over all n nested reconstructions.  NULL / n = 1 restores the plain
 * objective.  (The host samples them per step: objectives.py:159-201.) */
int saev_set_prefixes(saev_ctx* ctx, const int64_t* prefixes_host, int32_t n);

/* Several SAEs trained on the same batches (the reference's answer to an I/O-bound loop: one batch feeds every SAE of
 * a parallel group, train.py:3, :334-348): everything a step derives from x alone -- max|x| of the MSE rescale, the
 * column means the f16r encoder centres on, the centred row norms behind its error margins, the power-of-two x scale
 * and the fp16 / bf16 operand images -- is built once, by `leader`, and read by every context that shares with it.
 * A saev_step_forward of `ctx` borrows them when `leader`'s last saev_step_forward was given the same x pointer and
 * row count and nothing has borrowed-or-rebuilt in between; otherwise it builds its own, so results never depend on
 * the sharing.  Same device, d_model and encoder mode; both contexts on one stream (or ordered by the caller); the
 * leader must outlive the link.  leader = NULL detaches. */
int saev_share_x(saev_ctx* ctx, saev_ctx* leader);

/* Phase 1: renormalise W_dec (train.py:334-335), encode + TopK, fired flags, sparse decode, MSE,
 * main-path gradient pieces.  `training` = 0 gives the eval-mode forward (no tracker, no aux).
 * `n_rows_global` = rows of this step summed over all data-parallel ranks (= n_rows on one GPU);
 * it must equal the value later passed to saev_step_dead. */
int saev_step_forward(saev_ctx* ctx, const float* x, int32_t n_rows, int64_t n_rows_global, int32_t training,
                      void* stream);
/* Phase 2: tracker update with `n_rows_global` tokens (objectives.py:118-120), dead mask, AuxK
 * forward (modeling.py:75-103).  Training mode only.  The reference reads n_dead back on every step
 * (`.item()`, modeling.py:92).  Here the update kernel leaves a record in pinned host memory each step; the
 * call looks at the record of four steps earlier, which bounds the current count from above, and while that
 * bound is <= min(128, k_aux) (fp32 matrix-core kernels, d_model % 128 == 0: one launch per count window, saev_debug_cfg.aux_wide_route; 40 for other widths;
 * saev_debug_cfg.aux_small_max) -- zero dead latents included -- it enqueues kernels that take the count from
 * the device: no read-back, no stream synchronisation.  Only when the bound is larger (or no valid record
 * exists yet: the first four steps after creation / saev_bind_tracker / saev_tracker_touched) does it read
 * n_dead back and size the dense AuxK algebra on the host.  saev_last_aux_route tells which happened:
 * 0 no auxiliary work, 1 few-dead-latents kernels without a read-back, 2 the same after a read-back,
 * 3 dense algebra after a read-back; saev_dead_readbacks counts the read-backs so far. */
int saev_step_dead(saev_ctx* ctx, int64_t n_rows_global, void* stream);
int saev_last_aux_route(const saev_ctx* ctx);
/* Device memory the context itself owns, in bytes (the four flat buffers belong to the caller): which = 0 everything,
 * 1 the AuxK dead-set buffers (sized by saev_cfg.aux_dead_cap), 2 the Matryoshka gradient blocks (saev_set_prefixes). */
int64_t saev_scratch_bytes(const saev_ctx* ctx, int32_t which);
int64_t saev_dead_readbacks(const saev_ctx* ctx);
/* Phase 3: all four parameter gradients into the bound grad buffer (replaces autograd,
 * train.py:347-348), un-projected and un-clipped. */
int saev_step_backward(saev_ctx* ctx, void* stream);
/* Phase 3 in pieces, for data-parallel runs that overlap the gradient exchange with the backward:
 *   saev_backward_begin   latent-major ordering of the codes, db_dec, the AuxK contractions;
 *   saev_backward_rows    rows [lat_lo, lat_hi) of dW_dec (in the bound gradient buffer), of the TRANSPOSED W_enc
 *                         gradient (saev_grad_w_enc_t: (d_sae, d_model) row-major) and entries [lat_lo, lat_hi) of db_enc
 *                         are final when it returns (in stream order) -- the host may start reducing them;
 *   saev_backward_end     transposes saev_grad_w_enc_t (after the host has reduced it) into the W_enc segment.
 * saev_step_backward == begin + rows(0, d_sae) + end.  saev_bind_w_enc_t lets the host own the transposed-gradient
 * scratch ((d_sae * d_model) floats) so that a torch tensor can alias it for the collectives; the f16r encoder also
 * uses it as W_enc^T scratch during the forward. */
int saev_backward_begin(saev_ctx* ctx, void* stream);
int saev_backward_rows(saev_ctx* ctx, int32_t lat_lo, int32_t lat_hi, void* stream);
/* The same in two passes over the latents' (row, latent) pairs: part 1 forms the decoder gradient (rows of dL/dx_hat) and
 * keeps the per-pair dot products, part 2 the encoder gradient and db_enc (rows of x).  part 0 = saev_backward_rows.  A
 * data-parallel caller runs part 1, starts the exchange of the decoder half [W_dec | b_dec] -- final at that point -- and
 * lets it travel while part 2 and saev_backward_end run.  Both parts must cover the same latent ranges before
 * saev_backward_end. */
int saev_backward_rows_part(saev_ctx* ctx, int32_t lat_lo, int32_t lat_hi, int32_t part, void* stream);
int saev_backward_end(saev_ctx* ctx, void* stream);
/* Gathered backward -- the low-traffic exchange for strong scaling ("all-gather the sparse step state"; the
 * reference has no distributed training, framework/train.py:760-769).  Instead of summing the 4 N_p-byte gradient over
 * the ranks, every rank all-gathers what the backward consumes -- x, dL/dx_hat, the codes: (8 D + 8 k) bytes per row --
 * and forms the FULL gradient of the global batch itself, redundantly and bit-identically on every rank:
 *   saev_copy_step_state     this rank's rows of dL/dx_hat and of the codes into caller buffers (the rank's slice of the
 *                            all-gather outputs); n_rows = the rows of the training forward in flight.  With P Matryoshka
 *                            prefixes dL/dx_hat is the (n_rows, P, d_model) block of suffix-summed gradients;
 *   saev_backward_override   the gathered buffers (n_all <= max_batch rows of all ranks, rank-major) for the NEXT
 *                            saev_backward_begin / _rows: pairs, db_dec and both weight gradients then cover all n_all
 *                            rows.  One-shot (the next forward cancels it); NULL cancels.  Matryoshka: g_all is
 *                            (n_all, P, d_model) and every rank must have set the same cut points.
 * The auxiliary loss stays local to a rank's rows; its gradient is a few rows: saev_aux_compact_rows rows of
 * [dW_dec | dW_enc^T] for the dead latents, their db_enc and the term's share of db_dec.  Between saev_backward_begin and
 * saev_backward_rows the caller exports them (rows * (2 d_model + 1) + d_model floats), sums over ranks, imports:
 *   saev_aux_compact_rows / _export / _import.
 * Gradients carry the local 1/(n_local d_model) factor as in every data-parallel mode: tail with grad_scale = 1/world.
 * saev_trust_gradients(1) lets saev_step_tail use what the backward left behind (row statistics, tile squares) as
 * saev_train_step does -- the caller vouches that nothing writes the gradient between saev_backward_end and the tail. */
int saev_copy_step_state(saev_ctx* ctx, int32_t n_rows, float* g_out, int32_t* idx_out, float* val_out, void* stream);
int saev_backward_override(saev_ctx* ctx, const float* x_all, const float* g_all, const int32_t* idx_all,
                           const float* val_all, int32_t n_all);
int32_t saev_aux_compact_rows(const saev_ctx* ctx);
int saev_aux_compact_export(saev_ctx* ctx, float* buf, void* stream);
int saev_aux_compact_import(saev_ctx* ctx, const float* buf, void* stream);
int saev_trust_gradients(saev_ctx* ctx, int32_t on);
float* saev_grad_w_enc_t(saev_ctx* ctx);
int saev_bind_w_enc_t(saev_ctx* ctx, float* scratch);
/* Phase 4: grads *= grad_scale (1/world_size after a sum all-reduce), remove_parallel_grads
 * (train.py:351-352), global-norm clip (train.py:356-362; torch's formula for max_norm >= 0, so 0 zeroes the
 * gradient as in the reference; max_norm < 0 disables clipping), Adam with torch
 * defaults (train.py:294,444-446). `adam_step` is the 1-based step count. */
int saev_step_tail(saev_ctx* ctx, float lr, float max_norm, float grad_scale, int64_t adam_step, void* stream);
/* Phase 4 in two parts, over everything (shard_rank < 0: saev_step_tail == prepare + apply) or over rank
 * `shard_rank`'s chunk of each half of the flat buffers (saev_cfg.shard_world ranks; the gradient chunks must hold the
 * cross-rank SUM, e.g. after a reduce-scatter):
 *   saev_tail_prepare  remove_parallel_grads on the decoder rows of the range and the sum of squares of the range's
 *                      (projected, unscaled) gradient into saev_sumsq_device -- the caller all-reduces that one double
 *                      (SUM) when the ranges are per-rank, so that every rank clips with the same global norm;
 *   saev_tail_apply    clip coefficient from that sum, Adam on the range.
 * saev_bind_sumsq hands the context a caller-owned device double (a torch tensor a collective can run on). */
int saev_tail_prepare(saev_ctx* ctx, int32_t shard_rank, void* stream);
int saev_tail_apply(saev_ctx* ctx, float lr, float max_norm, float grad_scale, int64_t adam_step, int32_t shard_rank,
                    void* stream);
double* saev_sumsq_device(saev_ctx* ctx);
int saev_bind_sumsq(saev_ctx* ctx, double* sumsq);
/* One-shot: the next saev_step_forward waits for this hipEvent_t (on its stream) before it first touches W_dec, and
 * renormalises W_dec there instead of at its top -- for a caller whose decoder half of the parameter all-gather is
 * still running on another stream.  NULL cancels. */
int saev_wdec_ready_event(saev_ctx* ctx, void* event);
/* One-shot, the encoder half's counterpart: the next forward (saev_step_forward / saev_encode_topk) enqueues what depends
 * on the batch alone -- statistics, centring, fp16 images of x -- and waits for this hipEvent_t only before it first reads
 * W_enc or b_enc.  NULL cancels. */
int saev_wenc_ready_event(saev_ctx* ctx, void* event);

/* Phases 1-4 back to back for the single-GPU case -- with one difference to calling the four phases: the gradient
 * buffer is NOT a valid gradient afterwards.  The W_enc gradient stays in the transposed scratch (saev_grad_w_enc_t) and
 * is consumed there by the step's single Adam launch, and the rows of dW_dec are stored un-projected (the projection of
 * remove_parallel_grads is applied inside Adam as the rows are read).  Callers that want to look at gradients -- the log
 * steps of train() do -- run the phases: saev_step_backward ends with saev_backward_end, saev_step_tail projects in place. */
int saev_train_step(saev_ctx* ctx, const float* x, int32_t n_rows, float lr, float max_norm,
                    int64_t adam_step, void* stream);
/* The same with the batch drawn from an activation pool inside the step: row r of the batch is pool row rows[r] (the reference's
 * reservoir draw, data/buffers.py:201-211 + the loader's batch assembly, data/shuffled.py:506-552).  x_out (n_rows, d_model)
 * receives the batch as a contiguous matrix -- the first kernel of the step writes it while it reads the rows, so the draw costs
 * no pass of its own; it stays valid until the next call and is what saev_copy_last / the log block read as "x". */
int saev_train_step_gather(saev_ctx* ctx, const float* pool, const int64_t* rows, float* x_out, int32_t n_rows, float lr,
                           float max_norm, int64_t adam_step, void* stream);
/* DATA PARALLEL behind the ABI (SURVEY 8b "DDP": absent in the reference, whose train.py:760-769 has no distributed code at all).
 * One process per GPU, one context per process.  saev_comm_unique_id fills 128 bytes on one rank (ncclGetUniqueId); the caller
 * hands them to every rank by whatever means it has (torchrun's store, MPI, a file) and each rank calls saev_comm_init with its
 * rank and the world size (ncclCommInitRank over xGMI).  RCCL is taken from the process at run time (the librccl the process
 * has already loaded -- torch's, in the Python host -- else the system's): libsaev_amd.so does not link against it, and without
 * it these entry points return SAEV_UNSUPPORTED.
 * saev_train_step_dp is saev_train_step for a batch that is split evenly over the ranks: x_local holds this rank's n_local
 * rows, the global batch is n_local * world rows.  Enqueued on `stream`, nothing read back:
 *     forward on the local rows (the loss terms divide by the global row count)
 *     all-reduce MAX of the "fired this step" flags      (d_sae int32: the dead-latent tracker counts the global batch)
 *     dead-latent update and the auxiliary term, backward on the local rows
 *     all-reduce SUM of the flat gradient buffer        (n_params fp32, one collective)
 *     projection, clip on the global norm, Adam, with the gradient scaled by 1 / world (the mean over ranks of per-rank means)
 * -- the sequence framework/ddp.py runs from Python with tail="replicated", exchange="dense" (the sharded tail and the sparse
 * exchange exist there only).  Parameters must be identical on all ranks when the first step starts (broadcast them, or create
 * every rank from the same seed); they stay identical because every rank applies the same update. */
int saev_comm_unique_id(void* id128);
int saev_comm_init(saev_ctx* ctx, const void* id128, int32_t rank, int32_t world);
int saev_comm_world(const saev_ctx* ctx);   /* 0: no communicator */
int saev_comm_destroy(saev_ctx* ctx);
int saev_train_step_dp(saev_ctx* ctx, const float* x_local, int32_t n_local, float lr, float max_norm, int64_t adam_step,
                       void* stream);
/* PARAMETER OWNERSHIP.  With the f16r encoder the context keeps, from one call to the next, what its forward needs of W_enc
 * (fp16 operand images, a slice-major fp32 transpose, bias and norm shares: written by the Adam launch of saev_train_step, or by
 * the last forward that prepared them itself) and uses it for as long as only the library has written the parameter buffer.  A
 * caller that writes W_enc / b_enc / W_dec itself -- loads a checkpoint, broadcasts, pokes a value -- must say so before the
 * next call; saev_bind does it implicitly.  (The Python host calls it whenever torch's version counter of the buffer moved.)
 * Behind the contract, two checks so that a forgotten announcement is never a silent wrong answer:
 *   (1) before the images are used: every streamed step compares ALL of b_enc and a few thousand pseudo-random elements of W_enc
 *       with the copies its images came with; a difference sends that step down the exact dense route (correct codes, a slow
 *       step) and makes the next forward prepare from scratch.  Any write to b_enc and any bulk write to W_enc end here.
 *   (2) after they were used: the fused Adam of saev_train_step leaves two checksum words per 32 x 256 tile of W_enc as it writes
 *       it and compares them with the tile as it reads it one step later -- every element, at no extra traffic.  A write to even
 *       one element that (1) missed is found at the end of the first step that ran on the stale images; that step cannot be
 *       redone, so the next saev_step_forward / saev_train_step returns SAEV_STALE_PARAMS (once; saev_last_error says how many
 *       tiles) and the context prepares from scratch from there on. */
int saev_params_touched(saev_ctx* ctx);

/* Codes / reconstruction of the last saev_step_forward (device pointers into context scratch):
 * idx,val (n_rows x top_k); x_hat (n_rows x d_model). */
const int32_t* saev_last_idx(saev_ctx* ctx);
const float* saev_last_val(saev_ctx* ctx);
const float* saev_last_x_hat(saev_ctx* ctx);

/* Device-to-device copies of the same into caller buffers (any may be NULL), which hold `n_rows` rows:
 * n_rows must equal the batch of the last saev_step_forward (anything else is SAEV_INVALID_ARG). */
int saev_copy_last(saev_ctx* ctx, int32_t n_rows, int32_t* idx_out, float* val_out, float* x_hat_out, void* stream);

/* State of the predicted-bound mechanism (saev_cfg.bound_mode = 1), read back from the device (synchronises `stream`):
 * the current z, how many fused-encoder launches used predicted bounds and how many of those had to be repeated with
 * guaranteed bounds, and the mean candidate-list length of the last launch. */
int saev_bound_state(saev_ctx* ctx, float* z, int64_t* launches, int64_t* repeats, float* mean_candidates, void* stream);

/* Timing hooks for bench.py: wall duration in ms of the encoder kernel of the last step, measured
 * with HIP events on `stream` (call after the stream has been synchronised). */
int saev_enable_kernel_timing(saev_ctx* ctx, int32_t enable);
float saev_last_encoder_ms(saev_ctx* ctx);

#ifdef __cplusplus
}
#endif
#endif /* SAEV_AMD_H */
