// C ABI of libsaev_amd.so (see include/saev_amd.h): context, scratch, and the launch sequences of
// the train step.  No torch types; plain device pointers and a hipStream_t per call.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <dlfcn.h>
#include <mutex>
#include <string>
#include <vector>

#include "kernels.h"

namespace {
constexpr int AUX_KSPLIT_MAX = 16;
constexpr int CAND_CAP = 4096;
// Entries between the candidate lists of consecutive rows.  Not the capacity: with a 16 KB (power-of-two) row pitch the
// 32 rows a wave appends to at once fall on few memory channels, and how badly depends on which physical pages the
// allocation got -- the fused encoder then ran at 1.40 or 1.54 ms from one engine instance to the next
// (tools/experiments/bimodal_probe.py).  Measured pitches: +128 B 1.50 ms, +256 B / +512 B / +1 KB 1.41-1.42 ms, all
// stable; 1 KB it is.
constexpr int CAND_STRIDE = CAND_CAP + 256;
constexpr int TIMING_RING = 512;
// saev_step_dead decides between "nothing / a handful of dead latents" (kernels that take the count from the device) and
// "read the count back and run the dense algebra" from the record the device wrote DEAD_LAG steps earlier.
constexpr int DEAD_LAG = 4;
constexpr int DEAD_RING = 16;
enum { AUX_NONE = 0, AUX_SMALL_DEVICE = 1, AUX_SMALL_HOST = 2, AUX_DENSE = 3 };
}

struct saev_ctx {
    saev_cfg cfg{};
    saev_debug_cfg dbg{};  // route switches (saev_create_ex); all zero = shipped defaults
    int device = 0;
    std::string err;
    // bound buffers
    float* params = nullptr;
    float* grads = nullptr;
    float* adam_m = nullptr;
    float* adam_v = nullptr;
    // derived
    long n_params = 0;  // floats in each flat buffer, padding included
    long off_W_dec = 0, off_b_dec = 0, off_W_enc = 0, off_b_enc = 0;
    int shard_world = 1;
    long chunk_a = 0, chunk_b = 0;  // floats per rank of the [W_dec | b_dec] half and of the [W_enc | b_enc] half
    double* sumsq_bound = nullptr;  // caller-owned replacement of sumsq_total (so that a collective can reach it)
    hipEvent_t wdec_ready = nullptr;  // one-shot: the next forward waits for it before it touches W_dec
    hipEvent_t wenc_ready = nullptr;  // one-shot: ... before it touches W_enc / b_enc (the x-only preparation runs ahead of it)
    // scratch
    std::vector<void*> allocs;
    size_t scratch_bytes = 0, aux_bytes = 0;  // device memory the context owns: per-step scratch, AuxK dead-set buffers
    int cuts_last[MAX_PREFIXES] = {0};  // the cut points the forward in flight used (the backward must see the same)
    int32_t *cand_cnt = nullptr, *gmax = nullptr, *cand_idx = nullptr;
    int gmax_stride = 0;
    float* cand_val = nullptr;
    float* h_dense = nullptr;
    int32_t *idx = nullptr, *aux_idx = nullptr;
    float *val = nullptr, *aux_val = nullptr;
    float *x_hat = nullptr, *g = nullptr, *g_aux = nullptr;
    RowStats* rowstats = nullptr;
    uint32_t* bitmap = nullptr;
    int32_t* grp_prefix = nullptr;
    int32_t* scan_totals = nullptr;
    int32_t csc_epoch = 0;  // CscArgs::epoch of the last build
    int bitmap_words = 0;
    int back_rows = 0;  // max(max_batch, max_backward_rows): rows a (gathered) backward may cover
    int bitmap_words_last = 0;
    bool bitmap_clean = false;  // every word the next csc build will use is zero (the last full backward cleared behind itself)
    int bitmap_clean_words = 0; // ... for row pitches up to this many words
    int bitmap_prefill_words = 0, bitmap_prefill_rows = 0;  // the training decode in flight has set the bits of its codes at this pitch (0: no)
    bool last_backward_gathered = false;  // the previous backward ran over gathered rows (saev_backward_override): its forward's bits were wasted
    int32_t *counts = nullptr, *starts = nullptr;
    int2* pairs = nullptr;
    float* colsum_partials = nullptr;
    float* dval_pairs = nullptr;  // <g row, W_dec[latent]> per (row, latent) pair in CSC order (saev_backward_rows_part 1 -> 2)
    double *sumsq_partials = nullptr, *sumsq_total = nullptr;
    // squares of the W_enc gradient, taken by the transpose that ends the backward (saev_backward_end): valid until the
    // next backward; the tail uses them only when the caller vouches that nothing wrote the gradient since (trust_grads)
    bool wenc_sq_valid = false;
    // {projection coefficient, projected squares} of every decoder-gradient row, left by the kernels that wrote the rows
    // (DwRowsArgs::row_proj); valid after a one-pass backward over all latents, trusted like wenc_sq
    float2* row_proj = nullptr;
    float* enc_sq = nullptr;  // squares of the rows of the transposed W_enc gradient, from the same kernels
    bool row_proj_valid = false, tail_proj_in_adam = false;
    bool wenc_t_pending = false;  // saev_train_step: the W_enc gradient is still in dW_encT, the tail's Adam reads it there
    int64_t* toks = nullptr;
    int32_t *fired = nullptr, *dead = nullptr;
    int32_t* flags = nullptr;  // [0] need_dense_pre [1] need_dense [2] n_overflow [3] cand_max [4] n_dead [5] k_use [6,7,8] dead_update scratch
    int32_t *chunk_starts = nullptr, *part_starts = nullptr, *work_latent = nullptr;
    float *dW_encT = nullptr, *partials = nullptr, *db_partials = nullptr;
    // column-sliced weight gradients (launch_dw_slices; SAEV_AMD_DW=rows keeps dw_rows): slice-major copies of g and x left by
    // the decode, pair words / latents in pair order from the CSC build, the per-slice shares of dval
    bool dws_ok = false;         // geometry fits (d_model % 32 == 0, 32-bit offsets) and not switched off
    int dws_rows = 0;            // > 0: the copies describe the training forward in flight (that many rows)
    bool dws_pairs = false;      // the CSC build of this backward left pv / plat
    float *gS = nullptr, *xS = nullptr, *dvp = nullptr;
    // A gathered backward (saev_backward_override) of a context that LENDS its x-derived buffers (saev_share_x) must not write
    // the rows of all ranks over xS: its followers' forwards run after this backward and read xS as their own batch.  Such a
    // context gets a second slice-major buffer for the gathered rows, allocated at the first backward that needs it.
    float* xS_ov = nullptr;
    float* xS_bwd = nullptr;  // the slice-major x the backward in flight reads when it runs over gathered rows (xS or xS_ov)
    // dval[b][j] = <g_b, W_dec[idx[b][j]]> left by the decode itself (decode_q_kernel; kernels.h: DecodeArgs::dval_out): pass A of
    // the slices then forms dW_dec only.  dval_fwd: the forward in flight has left it (same condition as dws_rows, plus the shape)
    float* dval_rows = nullptr;
    bool dval_fwd = false;
    // the light finalize (kernels.h: DwSlicesArgs::wn2): ||w_i||^2 of the decoder rows as this step's normalize_rows wrote them
    float* wn2 = nullptr;
    float* sq_wave = nullptr;  // per-wave squares of the two passes (DwSlicesArgs::sq_wave_dec, then _enc: contiguous)
    int sq_wave_n = 0;         // > 0: the backward in flight left 2 x this many of them (the tail adds them to the clip norm)
    bool wn2_fresh = false;  // wn2 describes W_dec as it is now (set by the training forward, cleared by whatever writes W_dec)
    // the decode out of 32-column slices (sparse.hip: decode_s_kernel): slice-major copy of the normalised W_dec left by the step's
    // normalize_rows, per (slice, row) loss terms; the dval shares go through dvp
    float* WdS = nullptr;
    double* dec_part = nullptr;
    bool wds_fresh = false;  // WdS describes W_dec as it is now (this step's normalize_rows wrote both)
    bool dval_pairs_ready = false;  // the CSC build of this backward has written pv2 from it
    bool fused_forward = false;     // saev_train_step's forward: Matryoshka G blocks past the first are not needed row-major
    // exact refinement of the f16r encoder from 32-column slices (select.hip: refine_slices_kernel): split_f16r leaves x and
    // W_enc^T slice-major (xS; dW_encT in that layout), rs_part holds the per-slice shares of the survivors' dot products
    bool fwd_slices = false;     // geometry fits and not switched off (saev_debug_cfg.fwd_route)
    bool fwd_step = false;       // the forward in flight took that route: xS_c describes its batch, W_enc^T is slice-major
    float* rs_part = nullptr;
    float* xS_c = nullptr;       // the slice-major x of the step in flight (own xS, or the leader's: saev_share_x)
    int2 *pv = nullptr, *pv2 = nullptr;
    int32_t *plat = nullptr, *cut_lat = nullptr, *cut_list = nullptr;
    // saev_train_step: latents without pairs are flagged instead of having their dW_enc^T row zeroed (DwSlicesArgs::lat_unused)
    int32_t* lat_unused = nullptr;
    bool fused_step = false;     // inside saev_train_step: the transposed W_enc gradient is read by the fused Adam alone
    bool unused_valid = false;   // the backward in flight left lat_unused
    // Matryoshka prefixes of the step (P == 1: plain objective)
    int P = 1;
    int32_t cuts[MAX_PREFIXES] = {0};
    float* G = nullptr;  // (max_batch, P_cap, D)
    float* GS = nullptr; // slice-major copy of G for launch_dw_slices: [D / 32][P][rows][32] (with dws_ok)
    int P_cap = 0;
    // AuxK dense-over-dead-set path (auxk.hip)
    int n_dead_host = 0, k_use_host = 0;
    int64_t tokens_seen = 0;
    bool tracker_dirty = false;
    int nd_cap = 0;
    // per-step records of the dead set in pinned host memory (written by dead_update_kernel), one event per record
    DeadRecord* rec_host = nullptr;
    DeadRecord* rec_dev = nullptr;
    hipEvent_t dead_ev[DEAD_RING];
    bool dead_ev_created = false;
    int64_t dead_steps = 0;      // saev_step_dead calls so far (the current step's 1-based id during the call)
    int64_t rec_valid_from = 1;  // records of earlier steps predate a host write to the tracker
    int aux_route = AUX_NONE;    // what the step in flight does for the auxiliary loss
    int64_t n_readbacks = 0;     // blocking reads of n_dead so far (diagnostics: saev_dead_readbacks)
    std::vector<void*> aux_allocs;
    int32_t* dead_list = nullptr;
    float *Wenc_dead = nullptr, *Wdec_dead = nullptr, *H_dead = nullptr, *A_dead = nullptr, *dWd = nullptr, *dWe = nullptr,
          *dbe = nullptr, *aux_partials = nullptr, *WencT_dead = nullptr, *aux_small_part = nullptr, *aux_small_part2 = nullptr, *aux_small_partbe = nullptr;
    bool aux_dev_count = false;  // dense branch sized by a host-side BOUND of the dead count; the count itself stays on the device
    bool aux_small = false;  // this step's AuxK ran on the few-dead-latents path
    int aux_mfma_bound = 0;
    int aux_ndp = AUX_SMALL_MAX;  // row pitch of A / dA / the block partials of the few-dead-latents step in flight (AUX_MFMA_MAX beyond 64)
    int aux_mfma_cap = AUX_SMALL_MAX;  // largest bound the matrix-core kernels take in this context (its buffers decide)
    bool aux_mfma = false;   // ... in its fp32 matrix-core form (at most AUX_MFMA_MAX dead latents, d_model % 128 == 0: auxk.hip aux_mfma_*)
    bool aux_fused = false;  // ... in its one-pass form (at most AUX_FUSED_MAX dead latents: block partials instead of g_aux / A / dA)
    bool aux_all = false;    // dense branch with every dead latent selected (n_dead <= k_aux): no select, no mask
    uint8_t* A_mask = nullptr;
    // AuxK contractions on the f16x3 encoder kernel (F16X3 mode): operand images and compact vectors
    _Float16 *aux_ws1 = nullptr, *aux_ws2 = nullptr, *aux_xsA = nullptr, *aux_xsg = nullptr, *aux_kA = nullptr, *aux_kD = nullptr, *aux_kX = nullptr;
    bool aux_both = false;  // the dense route's forward has left the k-major images of A (aux_kA) and x (aux_kX) beside the row-form ones
    float* aux_parts = nullptr;
    int aux_kpad = 0;
    float *bias_dead = nullptr, *zero_bias = nullptr, *aux_scales = nullptr;  // aux_scales: {absmax, -, sA, 1, sg, 1}
    float* aux_sync = nullptr;     // per-workgroup maxima of the AuxK kernels that leave an operand scale behind (auxk.hip: pow2_parts_kernel)
    int aux_Dp2 = 0;
    // F16R: per-row candidate margins and the max encoder column norm (W_enc^T in fp32 lives in dW_encT during forward)
    float *row_margin = nullptr, *wnorm_scratch = nullptr, *surv_val = nullptr;
    int32_t *surv_idx = nullptr, *surv_cnt = nullptr, *surv_rng = nullptr;
    int rs_lat_range = 0, rs_n_ranges = 0;
    int32_t* tau_max = nullptr;   // (max_batch) largest predicted bound used per row
    float* heur_state = nullptr;  // [0] z  [1] failed predictions  [2] predicted-bound launches  [3] mean list length
    float *f16r_scales = nullptr, *mu = nullptr, *xnorm = nullptr, *b_shift = nullptr, *dot_part = nullptr, *xabs_part = nullptr,
          *sq_part = nullptr, *wmax_prev = nullptr;
    bool wmax_known = false;
    bool mu_ready = false;  // the step already put the column means of x into mu
    // ---- the streamed f16r step: what a forward derives from x comes from ONE pass (xprep_kernel) centred, scaled
    // and normalised with what the previous batch left; what it derives from W_enc was left by the fused Adam of the previous step
    // (AdamImageArgs) -- or by this context's last full preparation, while W_enc has not moved since.
    bool stream_ok = false;       // mode and geometry allow it (f16r, slice route of the refinement, guaranteed bounds)
    float *WeS = nullptr;         // slice-major fp32 W_enc^T of its own (the gradient scratch dW_encT no longer doubles as it)
    float *xn_part = nullptr, *amax_part = nullptr, *cmax_part = nullptr;
    float* b_seen = nullptr;      // b_enc as bias_finish read it (the staleness samples of xprep_kernel compare against it)
    int32_t *stale_host = nullptr, *stale_dev = nullptr;  // pinned words: [0] a streamed step found the parameters changed behind its
                                                          // back before using its images (and took the exact route); [1] the fused
                                                          // Adam found W_enc tiles changed AFTER the step had used them (AdamImageArgs::chk)
    uint32_t* wchk = nullptr;     // two checksum words per 32 x 256 tile of W_enc, left by the fused Adam that wrote it
    // Several SAEs on the same batches (saev_share_x) with the streamed preparation: the lender streams as a context on its own does and
    // keeps what its followers need of the step's x side (XprepArgs::mu_keep / xside_keep); a follower's fused Adam leaves ITS W images
    // centred on the lender's next mu, so that from the third step of a group nobody prepares anything from scratch.
    float *mu_keep = nullptr, *xside_keep = nullptr;
    bool fwd_streamed = false;     // (lender) the forward that built the current x-derived buffers took the streamed preparation ...
    bool fwd_moves_mu = false;     // ... inside saev_train_step: its second launch has moved mu on to this batch's mean
    int64_t fwd_mu_serial = -1;    // ... with the centre of this serial
    bool borrow_streamed = false;  // (follower) the forward in flight borrowed the x side of a streamed step of its lender
    bool follow_stream = false;    // ... and runs on W images its own Adam left: no preparation at all
    bool wchk_valid = false;      // wchk describes W_enc as the library last wrote it, and only the library may have written it since
    bool fwd_reused_wimg = false; // the forward in flight ran on operand images a previous step's Adam left (their checksums are due)
    uint32_t stale_salt = 0;
    int scale_par = 0;            // which half of f16r_scales (2 x 8 floats) belongs to the step in flight
    bool prep_valid = false;      // mu and scales[par][0, 4] describe a previous batch of this context
    bool wimg_fresh = false;      // ws / WeS / dot_part / sq_part / b_shift / wnorm_scratch describe W_enc AS IT IS NOW ...
    int64_t mu_serial = 0, wimg_mu_serial = -1;  // ... centred on the mu of this version (mu_serial: bumped whenever mu is rewritten)
    bool wimg_bf16_fresh = false; // bf16 encoder: ws describes W_enc as it is now
    bool stream_step = false;     // the forward in flight took the streamed preparation
    bool stats_pending = false, stats_lists = false;  // the forward of a fused train step left its statistics to saev_step_dead's launch
    bool aux_stats_pending = false;  // ... and its one-pass AuxK forward left the auxiliary loss to the backward's ordered-sum launch
    bool dead_list_ready = false; // ... which also left the list of dead latents (if any are dead)
    bool train_fused = false;     // inside saev_train_step: the forward moves mu, the tail's Adam leaves the next images
    const float* gather_pool = nullptr;   // saev_train_step_gather: the batch is rows[0..n) of this pool, x is where it is written
    const int64_t* gather_rows = nullptr;
    // Where the step in flight finds what was derived from x alone: max|x|, the column means, the centred row norms, the
    // per-workgroup maxima behind the x scale, and the fp16 / bf16 images.  Its own buffers -- or those of the context it
    // shares a batch with (saev_share_x: several SAEs trained on the same batches form them once).
    float *upper_c = nullptr, *mu_c = nullptr, *xnorm_c = nullptr, *xabs_c = nullptr;
    _Float16* xs_c = nullptr;
    saev_ctx* leader = nullptr;
    void* comm = nullptr;        // ncclComm_t (saev_comm_init)
    int comm_rank = 0, comm_world = 0;
    std::vector<saev_ctx*> followers;  // contexts whose `leader` is this one (saev_destroy / a new link clears them)
    const float* xprep_x = nullptr;  // what this context's own x-derived buffers currently describe
    int xprep_n = 0;
    int64_t xprep_serial = 0;        // bumped every time they are rebuilt
    int64_t leader_serial_seen = 0;  // the leader's serial this context last borrowed
    // f16x3 encoder operands
    _Float16 *xs = nullptr, *ws = nullptr;
    int Dp = 0, S_pad = 0, MB_pad = 0;
    int max_work = 0, max_part = 0;
    float* upper = nullptr;
    saev_step_stats* stats = nullptr;
    double* stats_scratch = nullptr;  // per-workgroup partial sums + ticket of stats_reduce_kernel
    int* tickets = nullptr;           // arrival counters of "last workgroup finishes" kernels (zero between launches)
    // gathered backward (saev_backward_override): the (row, latent) pairs of ALL ranks' rows, set for one backward
    const float *ov_x = nullptr, *ov_g = nullptr, *ov_val = nullptr;
    const int32_t* ov_idx = nullptr;
    int ov_n = 0;
    float* db_aux = nullptr;       // the auxiliary term's share of db_dec, kept apart while an override is active
    bool trust_grads = false;      // the caller vouches that nothing touches the gradient between backward and tail
    // state of the step in flight
    const float* x_last = nullptr;
    int n_last = 0;
    int training_last = 0;
    int P_last = 1;
    // timing
    bool timing = false;
    hipEvent_t ev_start[TIMING_RING], ev_stop[TIMING_RING];
    bool ev_created = false;
    long ev_count = 0;
};

#define HIPCHK(ctx, expr)                                                                    \
    do {                                                                                     \
        hipError_t _e = (expr);                                                              \
        if (_e != hipSuccess) {                                                              \
            (ctx)->err = std::string(#expr) + ": " + hipGetErrorString(_e);                  \
            return SAEV_HIP_ERROR;                                                           \
        }                                                                                    \
    } while (0)

#define REQUIRE(ctx, cond, code, msg)                                                        \
    do {                                                                                     \
        if (!(cond)) {                                                                       \
            (ctx)->err = (msg);                                                              \
            return (code);                                                                   \
        }                                                                                    \
    } while (0)

namespace {

template <typename T>
int alloc(saev_ctx* c, T** p, size_t count) {
    void* q = nullptr;
    if (count == 0) count = 1;
    hipError_t e = hipMalloc(&q, count * sizeof(T));
    if (e != hipSuccess) {
        c->err = std::string("hipMalloc failed: ") + hipGetErrorString(e);
        return SAEV_HIP_ERROR;
    }
    c->allocs.push_back(q);
    c->scratch_bytes += count * sizeof(T);
    *p = static_cast<T*>(q);
    return SAEV_OK;
}

// TopK bound of the fp16-image encoders: the minimum over 32 group maxima for top_k <= 32; 64 groups with the top_k-th
// largest of the group maxima for 32 < top_k <= 64.  saev_debug_cfg.ngroups = 64 forces the second variant for small k as well: it
// cuts the candidates per row from ~980 to ~360 at config 2, but its bound phase (32 published maxima per lane, a
// bisection over packed 16-bit keys) costs more than the shorter lists save (encoder 1.43-1.51 vs 1.35-1.38 ms).
// {x scale, W scale, x scale, 1, square normaliser, -, -, -} of the step in flight / of the next one (streamed f16r step)
float* scl(const saev_ctx* c) { return c->f16r_scales + 8 * c->scale_par; }
float* scl_next(const saev_ctx* c) { return c->f16r_scales + 8 * (c->scale_par ^ 1); }

int f16_ngroups(const saev_ctx* c) {
    if (c->cfg.top_k > 32 || c->dbg.ngroups == 64) return 64;
    return 32;
}

int encoder_splits(int n_rows, int S, int tile_rows, int tile_latents, int target_wgs) {
    const int nb = (n_rows + tile_rows - 1) / tile_rows;
    const int nst = (S + tile_latents - 1) / tile_latents;
    int sp = std::max(1, std::min((target_wgs + nb - 1) / nb, nst));
    // the fewest splits that keep the longest walk as short: 24 tiles over 16 splits are walks of 1 and 2 tiles -- as long as 12
    // splits of 2 each, with a third more workgroups paying a first tile's bound refresh and sharing the board's power
    // (configs[0]: encoder 92 -> 84 us, step 0.435 -> 0.415 ms; profiles/r06_c0_encoder_grid.txt)
    const int longest = (nst + sp - 1) / sp;
    return (nst + longest - 1) / longest;
}

bool fused_supported(const saev_cfg& c) { return c.top_k <= 64; }

int alloc_aux_buffers(saev_ctx* c, int cap);  // (below, with the AuxK launch sequences)

void timing_begin(saev_ctx* c, hipStream_t s) {
    if (c->timing) hipEventRecord(c->ev_start[c->ev_count % TIMING_RING], s);
}
void timing_end(saev_ctx* c, hipStream_t s) {
    if (c->timing) {
        hipEventRecord(c->ev_stop[c->ev_count % TIMING_RING], s);
        c->ev_count++;
    }
}

}  // namespace

extern "C" {

int saev_abi_version(void) { return SAEV_AMD_ABI_VERSION; }

int saev_layout(const saev_cfg* cfg, saev_layout_t* out) {
    if (!cfg || !out || cfg->d_model <= 0 || cfg->d_sae <= 0) return SAEV_INVALID_ARG;
    const int64_t S = cfg->d_sae, D = cfg->d_model, N = std::max(1, cfg->shard_world);
    out->chunk_a = (S + 1 + N - 1) / N * D;
    out->chunk_b = ((D * S + S + N - 1) / N + 3) / 4 * 4;
    out->off_W_dec = 0;
    out->off_b_dec = S * D;
    out->off_W_enc = N * out->chunk_a;
    out->off_b_enc = out->off_W_enc + D * S;
    out->n_total = N * out->chunk_a + N * out->chunk_b;
    return SAEV_OK;
}

const char* saev_last_error(const saev_ctx* ctx) { return ctx ? ctx->err.c_str() : "null context"; }

int saev_create(const saev_cfg* cfg, int device, saev_ctx** out) { return saev_create_ex(cfg, nullptr, device, out); }

int saev_create_ex(const saev_cfg* cfg, const saev_debug_cfg* dbg, int device, saev_ctx** out) {
    if (!cfg || !out) return SAEV_INVALID_ARG;
    *out = nullptr;
    if (cfg->d_model <= 0 || cfg->d_sae <= 0 || cfg->top_k <= 0 || cfg->max_batch <= 0) return SAEV_INVALID_ARG;
    // candidate lists are addressed with 32-bit byte offsets (max_batch <= 246 723 rows per call)
    if ((uint64_t)cfg->max_batch * CAND_STRIDE * 4ull >= (1ull << 32)) return SAEV_INVALID_ARG;
    if (cfg->d_model % 4 != 0 || cfg->d_sae % 4 != 0 || cfg->d_model > 4096) return SAEV_UNSUPPORTED;
    if (cfg->k_aux < 0 || cfg->k_aux > 1024) return SAEV_UNSUPPORTED;
    if (cfg->encoder_mode != SAEV_ENCODER_F32 && cfg->encoder_mode != SAEV_ENCODER_F16X3 && cfg->encoder_mode != SAEV_ENCODER_BF16 &&
        cfg->encoder_mode != SAEV_ENCODER_F16R)
        return SAEV_INVALID_ARG;
    saev_ctx* c = new saev_ctx();
    c->cfg = *cfg;
    c->cfg.top_k = std::min(cfg->top_k, cfg->d_sae);
    if (dbg != nullptr && dbg->struct_size > 0)
        std::memcpy(&c->dbg, dbg, std::min((size_t)dbg->struct_size, sizeof(saev_debug_cfg)));
    c->device = device;
    if (hipSetDevice(device) != hipSuccess) {
        delete c;
        return SAEV_HIP_ERROR;
    }
    const long S = cfg->d_sae, D = cfg->d_model, MB = cfg->max_batch, K = c->cfg.top_k, KA = cfg->k_aux;
    // rows a BACKWARD may cover: the context's own batch, or -- gathered backward of a data-parallel run that exchanges the
    // sparse step state -- every rank's rows.  Only the latent-major pair order, the slice-major copies and the partial rows
    // are sized by it; everything the forward writes stays at max_batch.
    const long MBB = std::max<long>(MB, cfg->max_backward_rows);
    c->back_rows = (int)MBB;
    if ((uint64_t)MBB * K >= (1ull << 31)) { delete c; return SAEV_INVALID_ARG; }
    // Flat layout [W_dec | b_dec | pad | W_enc | b_enc | pad].  With shard_world = N > 1 each half is padded to N equal
    // chunks -- chunks of the first half are whole decoder rows -- so that a data-parallel run can reduce-scatter the
    // gradient halves, let every rank run the tail on its chunk of each, and all-gather the parameter halves separately
    // (the encoder half first: the next forward needs it first).  N = 1: no padding, the state_dict order as it is.
    {
        saev_layout_t lay;
        saev_layout(cfg, &lay);
        c->shard_world = std::max(1, cfg->shard_world);
        c->chunk_a = lay.chunk_a; c->chunk_b = lay.chunk_b;
        c->off_W_dec = lay.off_W_dec; c->off_b_dec = lay.off_b_dec; c->off_W_enc = lay.off_W_enc; c->off_b_enc = lay.off_b_enc;
        c->n_params = lay.n_total;
    }
    int rc = SAEV_OK;
#define A(p, n) if (rc == SAEV_OK) rc = alloc(c, &c->p, (size_t)(n))
    c->gmax_stride = (int)((MB + 255) / 256 * 256);  // (padding the group pitch changes nothing: measured)
    A(cand_cnt, MB); A(gmax, (size_t)64 * c->gmax_stride); A(cand_idx, MB * CAND_STRIDE); A(cand_val, MB * CAND_STRIDE);
    A(h_dense, MB * S);
    A(idx, MB * K); A(val, MB * K);
    if (KA > 0) { A(aux_idx, MB * KA); A(aux_val, MB * KA); A(g_aux, MB * D); A(dead_list, S); }
    A(x_hat, MB * D); A(g, MB * D);
    A(rowstats, MB);
    c->bitmap_words = (int)(((MBB + 31) / 32 + 7) / 8 * 8);
    A(bitmap, S * c->bitmap_words);
    A(grp_prefix, S * (c->bitmap_words / 8));
    A(scan_totals, ((S + 1023) / 1024) * 4);
    A(counts, S); A(starts, S + 1); A(pairs, MBB * K); A(dval_pairs, MBB * (size_t)cfg->top_k);
    {
        const long max_pairs = MBB * K;
        c->max_work = (int)(S + (max_pairs + DW_CHUNK - 1) / DW_CHUNK);
        c->max_part = (int)(2 * ((max_pairs + DW_CHUNK - 1) / DW_CHUNK) + 2);
    }
    A(chunk_starts, S + 1); A(part_starts, S); A(work_latent, c->max_work);
    A(dW_encT, S * D); A(partials, (size_t)c->max_part * 2 * D); A(db_partials, c->max_part); A(row_proj, S); A(enc_sq, S);
    {
        const bool rows_only = c->dbg.dw_route == 1;
        c->dws_ok = !rows_only && D % DWS_SLICE == 0 && (uint64_t)S * D * 4ull < (1ull << 32) && MBB < (1l << 24) &&
                    (uint64_t)MBB * K < (1ull << 31);
    }
    c->fwd_slices = c->cfg.encoder_mode == SAEV_ENCODER_F16R && (c->dbg.fwd_route == 0 || c->dbg.fwd_route == 2) && c->dbg.fused_chain == 0 && D % RS_SLICE == 0 &&
                    (uint64_t)S * 128ull < (1ull << 32) - 256ull && fused_supported(c->cfg);
    if (c->fwd_slices) {
        A(rs_part, (size_t)(D / RS_SLICE) * MB * REFINE_CAP); A(surv_rng, MB * RS_MAX_RANGES);
        // passes over a slice cover RS_LAT_RANGE latents each (x 128 bytes = 4 MB = an XCD's L2: tools/ubench/row_gather.hip gathers
        // out of a whole 4 MB slice at 20-22 TB/s; 16 384-latent ranges made refine_slices 4 % slower), at most RS_MAX_RANGES
        c->rs_lat_range = (int)std::max<long>(RS_LAT_RANGE, ((S + RS_MAX_RANGES - 1) / RS_MAX_RANGES + 255) / 256 * 256);
        c->rs_n_ranges = (int)((S + c->rs_lat_range - 1) / c->rs_lat_range);
    }
    if (c->fwd_slices && !c->dws_ok) A(xS, MBB * D);
    if (c->dws_ok) {
        A(gS, MBB * D); A(xS, MBB * D); A(dvp, (size_t)(D / DWS_SLICE) * MBB * K);
        if (c->dbg.dw_route != 2 && decode_forms_dval((int)D, (int)K)) A(dval_rows, MB * K);  // (routes 0, 4: dval from the decode)
        if (c->dval_rows != nullptr && c->dbg.dw_route == 4 && decode_slices_supported((int)D, (int)S, (int)K, (int)K) && c->cfg.normalize_w_dec) {
            A(WdS, S * D); A(dec_part, (size_t)(D / 32) * MB * 3);
        }
        A(pv, MBB * K); A(pv2, MBB * K); A(plat, MBB * K); A(cut_lat, (MBB * K + DWS_RUN - 1) / DWS_RUN); A(cut_list, 4 * (1 + (MBB * K + DWS_RUN - 1) / DWS_RUN)); A(lat_unused, S);
        if (c->dval_rows != nullptr && c->dbg.fin_route == 0) { A(wn2, S); A(sq_wave, (size_t)2 * dw_slices_waves((int)D, (int)(MBB * K))); }
    }
    A(colsum_partials, ((MBB + 63) / 64) * D);
    A(sumsq_partials, 2 * 1024 + (S + 3) / 4 + 8 + transpose_blocks((int)S, (int)D)); A(sumsq_total, 1);
    if (c->cfg.encoder_mode != SAEV_ENCODER_F32 || KA > 0) {  // (the f32 encoder needs the image geometry for AuxK only)
        c->Dp = (int)((D + 31) / 32 * 32);
        c->S_pad = (int)((S + 255) / 256 * 256);
        c->MB_pad = (int)((MB + 255) / 256 * 256);
        A(zero_bias, std::max(S, D)); A(aux_scales, 16); A(aux_sync, absmax_parts_max((int)MB));
    }
    if (c->cfg.encoder_mode != SAEV_ENCODER_F32) {
        A(xs, (size_t)c->MB_pad * 2 * c->Dp);
        A(ws, (size_t)c->S_pad * 2 * c->Dp);
        A(row_margin, MB); A(wnorm_scratch, std::max((S + 3) / 4, 3 * ((S + 255) / 256))); A(f16r_scales, 16); A(mu, D); A(xnorm, 2 * MB); A(b_shift, S);
        A(xabs_part, (MB + 3) / 4);
        if (c->cfg.encoder_mode == SAEV_ENCODER_F16R) { A(dot_part, (size_t)2 * (c->Dp / 32) * c->S_pad); A(sq_part, (size_t)2 * (c->Dp / 32) * c->S_pad); A(wmax_prev, 1); }
        if (c->cfg.encoder_mode == SAEV_ENCODER_F16R) { A(surv_idx, MB * REFINE_CAP); A(surv_val, MB * REFINE_CAP); A(surv_cnt, MB); }
    }
    c->stream_ok = c->fwd_slices && c->cfg.bound_mode == 0 && c->dbg.prep_route == 0 && D % 32 == 0 && c->Dp == (int)D;
    if (c->stream_ok) {
        A(WeS, S * D); A(xn_part, (size_t)(D / 32) * c->MB_pad * 2);
        A(amax_part, (size_t)(c->MB_pad / 256) * (D / 32)); A(cmax_part, (size_t)(c->MB_pad / 256) * (D / 32)); A(b_seen, S);
        A(mu_keep, D); A(xside_keep, 4);
    }
    if (c->stream_ok || c->cfg.encoder_mode == SAEV_ENCODER_BF16) A(wchk, (size_t)2 * ((S + 255) / 256) * ((D + 31) / 32));
    A(toks, S); A(fired, S); A(dead, S); A(flags, 16); A(upper, 1); A(stats, 1);
    A(tau_max, MB); A(heur_state, 8); A(stats_scratch, STATS_SCRATCH_DOUBLES); A(tickets, 8); A(db_aux, D);
#undef A
    if (rc != SAEV_OK) {
        // keep the context so the caller can read the message, but report failure
        for (void* p : c->allocs) hipFree(p);
        delete c;
        return rc;
    }
    hipMemset(c->toks, 0, S * sizeof(int64_t));
    hipMemset(c->fired, 0, S * sizeof(int32_t));
    hipMemset(c->dead, 0, S * sizeof(int32_t));
    hipMemset(c->flags, 0, 16 * sizeof(int32_t));
    if (c->stream_ok || c->cfg.encoder_mode == SAEV_ENCODER_BF16) {
        void* hp = nullptr;
        if (hipHostMalloc(&hp, 64, hipHostMallocMapped) == hipSuccess &&
            hipHostGetDevicePointer(reinterpret_cast<void**>(&c->stale_dev), hp, 0) == hipSuccess) {
            c->stale_host = static_cast<int32_t*>(hp);
            c->stale_host[0] = 0; c->stale_host[1] = 0;
        } else {
            if (hp) hipHostFree(hp);
            c->stale_host = nullptr; c->stale_dev = nullptr;  // (the device-side part of the check still works)
        }
    }
    hipMemset(c->scan_totals, 0, ((S + 1023) / 1024) * 4 * sizeof(int32_t));  // (no workgroup's "ready" word equals a build's epoch)
    hipMemset(c->stats, 0, sizeof(saev_step_stats));
    hipMemset(c->stats_scratch, 0, STATS_SCRATCH_DOUBLES * sizeof(double));
    hipMemset(c->tickets, 0, 8 * sizeof(int));
    {
        const float init[8] = {2.6f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};  // z starts where a Gaussian row of 32 k latents has ~8 k values above its bound
        hipMemcpy(c->heur_state, init, sizeof(init), hipMemcpyHostToDevice);
    }
    hipMemset(c->rowstats, 0, MB * sizeof(RowStats));
    if (c->xs) hipMemset(c->xs, 0, (size_t)c->MB_pad * 2 * c->Dp * sizeof(_Float16));
    if (c->zero_bias) hipMemset(c->zero_bias, 0, std::max(S, D) * sizeof(float));
    if (c->f16r_scales) hipMemset(c->f16r_scales, 0, 16 * sizeof(float));
    if (KA > 0) {
        // Every AuxK buffer is sized here for the dead set a healthy run meets: no allocation happens inside such a run's
        // steps.  Default min(d_sae, max(4096, 8 k_aux)) dead latents (2.3 GB at configs[1]; d_sae would be 11.5 GB
        // there and 37 GB at configs[3]); a step that meets more grows them (saev_step_dead, reported on stderr).
        const int s4 = (int)((S + 3) / 4 * 4);
        const int want = cfg->aux_dead_cap > 0 ? cfg->aux_dead_cap : std::max(4096, 8 * (int)KA);
        const int cap = std::min((want + 3) / 4 * 4, s4);
        rc = alloc_aux_buffers(c, cap);
        if (rc == SAEV_OK) {
            void* h = nullptr;
            if (hipHostMalloc(&h, DEAD_RING * sizeof(DeadRecord), hipHostMallocMapped) != hipSuccess ||
                hipHostGetDevicePointer(reinterpret_cast<void**>(&c->rec_dev), h, 0) != hipSuccess) {
                rc = SAEV_HIP_ERROR;
            } else {
                c->rec_host = static_cast<DeadRecord*>(h);
                std::memset(h, 0, DEAD_RING * sizeof(DeadRecord));
                for (int i = 0; i < DEAD_RING && rc == SAEV_OK; ++i)
                    if (hipEventCreateWithFlags(&c->dead_ev[i], hipEventDisableTiming) != hipSuccess) rc = SAEV_HIP_ERROR;
                c->dead_ev_created = rc == SAEV_OK;
            }
        }
        if (rc != SAEV_OK) {
            for (void* q : c->allocs) hipFree(q);
            for (void* q : c->aux_allocs) hipFree(q);
            if (c->rec_host) hipHostFree(c->rec_host);
            delete c;
            return rc;
        }
    }
    hipDeviceSynchronize();
    *out = c;
    return SAEV_OK;
}

// (W images a context keeps are centred on a mu identified by a SERIAL of whoever owned that mu: its lender's while it follows,
// its own otherwise.  Serials of different contexts are unrelated numbers, so the images are dropped whenever the owner changes --
// an equal number must never pass for an equal centre.)
static void unlink_from_leader(saev_ctx* c) {
    if (c->leader != nullptr) {
        auto& f = c->leader->followers;
        f.erase(std::remove(f.begin(), f.end(), c), f.end());
        c->leader = nullptr;
        c->wimg_fresh = false;
        c->wchk_valid = false;
        c->borrow_streamed = false;
        c->follow_stream = false;
    }
}

void saev_destroy(saev_ctx* c) {
    if (!c) return;
    // no dangling links either way: followers fall back to their own x-derived buffers, the leader forgets this context
    for (saev_ctx* f : c->followers) { f->leader = nullptr; f->wimg_fresh = false; f->wchk_valid = false; f->borrow_streamed = false; f->follow_stream = false; }
    c->followers.clear();
    unlink_from_leader(c);
    hipSetDevice(c->device);
    hipDeviceSynchronize();
    saev_comm_destroy(c);
    for (void* p : c->allocs) hipFree(p);
    for (void* p : c->aux_allocs) hipFree(p);
    if (c->G) hipFree(c->G);
    if (c->GS) hipFree(c->GS);
    if (c->rec_host) hipHostFree(c->rec_host);
    if (c->stale_host) hipHostFree(c->stale_host);
    if (c->dead_ev_created)
        for (int i = 0; i < DEAD_RING; ++i) hipEventDestroy(c->dead_ev[i]);
    if (c->ev_created)
        for (int i = 0; i < TIMING_RING; ++i) {
            hipEventDestroy(c->ev_start[i]);
            hipEventDestroy(c->ev_stop[i]);
        }
    delete c;
}

int saev_bind(saev_ctx* c, float* params, float* grads, float* adam_m, float* adam_v) {
    if (!c) return SAEV_INVALID_ARG;
    REQUIRE(c, params != nullptr, SAEV_INVALID_ARG, "saev_bind: params is NULL");
    REQUIRE(c, ((uintptr_t)params % 16) == 0, SAEV_INVALID_ARG, "saev_bind: params must be 16-byte aligned");
    c->params = params;
    c->wn2_fresh = false;
    c->wimg_fresh = false;
    c->wimg_bf16_fresh = false;
    c->wchk_valid = false;
    c->grads = grads;
    c->adam_m = adam_m;
    c->adam_v = adam_v;
    return SAEV_OK;
}

int saev_bind_tracker(saev_ctx* c, int64_t* toks, int32_t* fired) {
    if (!c) return SAEV_INVALID_ARG;
    REQUIRE(c, toks && fired, SAEV_INVALID_ARG, "saev_bind_tracker: NULL buffer");
    c->toks = toks;
    c->fired = fired;
    c->tracker_dirty = true;
    c->rec_valid_from = c->dead_steps + 1;
    return SAEV_OK;
}

int saev_set_prefixes(saev_ctx* c, const int64_t* prefixes_host, int32_t n) {
    if (!c) return SAEV_INVALID_ARG;
    const int S = c->cfg.d_sae;
    if (prefixes_host == nullptr || n <= 1) {
        REQUIRE(c, prefixes_host == nullptr || (n == 1 && prefixes_host[0] == S), SAEV_INVALID_ARG,
                "a single prefix must equal d_sae");
        c->P = 1;
        return SAEV_OK;
    }
    REQUIRE(c, n <= MAX_PREFIXES, SAEV_UNSUPPORTED, "at most 16 Matryoshka prefixes are supported");
    REQUIRE(c, prefixes_host[0] >= 1 && prefixes_host[n - 1] == S, SAEV_INVALID_ARG,
            "prefixes must start at >= 1 and end at d_sae");
    for (int p = 1; p < n; ++p)
        REQUIRE(c, prefixes_host[p] > prefixes_host[p - 1], SAEV_INVALID_ARG, "prefixes must be strictly increasing");
    if (n > c->P_cap) {
        hipDeviceSynchronize();
        if (c->G) hipFree(c->G);
        if (c->GS) hipFree(c->GS);
        c->G = nullptr;
        c->GS = nullptr;
        c->P_cap = 0;
        void* q = nullptr;
        if (hipMalloc(&q, (size_t)c->cfg.max_batch * n * c->cfg.d_model * sizeof(float)) != hipSuccess) {
            c->err = "out of device memory for the Matryoshka gradient buffer";
            return SAEV_HIP_ERROR;
        }
        c->G = (float*)q;
        // (virtual row p * rows + b of a pair word must fit 24 bits: otherwise the row kernels serve the Matryoshka backward)
        if (c->dws_ok && (long)c->cfg.max_batch * n < (1l << 24)) {
            if (hipMalloc(&q, (size_t)c->cfg.max_batch * n * c->cfg.d_model * sizeof(float)) != hipSuccess) {
                c->err = "out of device memory for the Matryoshka gradient buffer (slice-major copy)";
                return SAEV_HIP_ERROR;
            }
            c->GS = (float*)q;
        }
        c->P_cap = n;
    }
    c->P = n;
    for (int p = 0; p < n; ++p) c->cuts[p] = (int32_t)prefixes_host[p];
    return SAEV_OK;
}

int saev_tracker_touched(saev_ctx* c) {
    if (!c) return SAEV_INVALID_ARG;
    c->tracker_dirty = true;
    c->rec_valid_from = c->dead_steps + 1;  // older records describe a tracker that no longer exists
    return SAEV_OK;
}

int saev_share_x(saev_ctx* c, saev_ctx* leader) {
    if (!c) return SAEV_INVALID_ARG;
    if (leader == nullptr || leader == c) { unlink_from_leader(c); return SAEV_OK; }
    REQUIRE(c, leader->device == c->device && leader->cfg.d_model == c->cfg.d_model && leader->cfg.encoder_mode == c->cfg.encoder_mode,
            SAEV_INVALID_ARG, "saev_share_x: both contexts must live on one device with the same d_model and encoder mode");
    REQUIRE(c, leader->leader == nullptr, SAEV_INVALID_ARG, "saev_share_x: the leader must build its own x-derived buffers");
    REQUIRE(c, c->followers.empty(), SAEV_INVALID_ARG, "saev_share_x: a context that lends its buffers cannot borrow");
    unlink_from_leader(c);
    c->leader = leader;
    c->wimg_fresh = false;  // (see unlink_from_leader: the images' centre changes owner)
    c->wchk_valid = false;
    leader->followers.push_back(c);
    c->leader_serial_seen = leader->xprep_serial;  // nothing built before this call is borrowed
    return SAEV_OK;
}

int saev_last_aux_route(const saev_ctx* c) { return c ? c->aux_route : -1; }
int64_t saev_scratch_bytes(const saev_ctx* c, int32_t which) {
    if (!c) return -1;
    const size_t matry = (c->G ? (size_t)c->cfg.max_batch * c->P_cap * c->cfg.d_model * sizeof(float) : 0) * (c->GS ? 2 : 1);
    return (int64_t)(which == 1 ? c->aux_bytes : which == 2 ? matry : c->scratch_bytes + c->aux_bytes + matry);
}
int64_t saev_dead_readbacks(const saev_ctx* c) { return c ? c->n_readbacks : -1; }

int saev_copy_last(saev_ctx* c, int32_t n_rows, int32_t* idx_out, float* val_out, float* x_hat_out, void* stream) {
    if (!c) return SAEV_INVALID_ARG;
    REQUIRE(c, c->n_last > 0, SAEV_INVALID_ARG, "saev_copy_last: no forward has run");
    REQUIRE(c, n_rows == c->n_last, SAEV_INVALID_ARG,
            "saev_copy_last: n_rows differs from the batch of the last forward (the caller's buffers are sized by it)");
    hipStream_t s = (hipStream_t)stream;
    const size_t nk = (size_t)c->n_last * c->cfg.top_k, nd = (size_t)c->n_last * c->cfg.d_model;
    if (idx_out) HIPCHK(c, hipMemcpyAsync(idx_out, c->idx, nk * sizeof(int32_t), hipMemcpyDeviceToDevice, s));
    if (val_out) HIPCHK(c, hipMemcpyAsync(val_out, c->val, nk * sizeof(float), hipMemcpyDeviceToDevice, s));
    if (x_hat_out) HIPCHK(c, hipMemcpyAsync(x_hat_out, c->x_hat, nd * sizeof(float), hipMemcpyDeviceToDevice, s));
    return SAEV_OK;
}

int64_t* saev_toks_since_active(saev_ctx* c) { return c ? c->toks : nullptr; }
int32_t* saev_fired_flags(saev_ctx* c) { return c ? c->fired : nullptr; }
const saev_step_stats* saev_stats_device(saev_ctx* c) { return c ? c->stats : nullptr; }
const int32_t* saev_last_idx(saev_ctx* c) { return c ? c->idx : nullptr; }
const float* saev_last_val(saev_ctx* c) { return c ? c->val : nullptr; }
const float* saev_last_x_hat(saev_ctx* c) { return c ? c->x_hat : nullptr; }

int saev_read_stats(saev_ctx* c, saev_step_stats* out_host, void* stream) {
    if (!c || !out_host) return SAEV_INVALID_ARG;
    hipStream_t s = (hipStream_t)stream;
    HIPCHK(c, hipMemcpyAsync(out_host, c->stats, sizeof(saev_step_stats), hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipStreamSynchronize(s));
    return SAEV_OK;
}

int saev_bound_state(saev_ctx* c, float* z, int64_t* launches, int64_t* repeats, float* mean_candidates, void* stream) {
    if (!c) return SAEV_INVALID_ARG;
    float h[4] = {0.f, 0.f, 0.f, 0.f};
    hipStream_t s = (hipStream_t)stream;
    HIPCHK(c, hipMemcpyAsync(h, c->heur_state, sizeof(h), hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipStreamSynchronize(s));
    if (z) *z = h[0];
    if (repeats) *repeats = (int64_t)h[1];
    if (launches) *launches = (int64_t)h[2];
    if (mean_candidates) *mean_candidates = h[3];
    return SAEV_OK;
}

int saev_enable_kernel_timing(saev_ctx* c, int32_t enable) {
    if (!c) return SAEV_INVALID_ARG;
    if (enable && !c->ev_created) {
        for (int i = 0; i < TIMING_RING; ++i) {
            HIPCHK(c, hipEventCreate(&c->ev_start[i]));
            HIPCHK(c, hipEventCreate(&c->ev_stop[i]));
        }
        c->ev_created = true;
    }
    c->timing = enable != 0;
    c->ev_count = 0;
    return SAEV_OK;
}

// mean duration (ms) of the encoder kernel over the steps recorded since timing was enabled
// (at most the last TIMING_RING); caller must have synchronised the stream.
float saev_last_encoder_ms(saev_ctx* c) {
    if (!c || !c->ev_created || c->ev_count == 0) return -1.f;
    const long n = std::min<long>(c->ev_count, TIMING_RING);
    double tot = 0;
    for (long i = 0; i < n; ++i) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, c->ev_start[i], c->ev_stop[i]) != hipSuccess) return -1.f;
        tot += ms;
    }
    return (float)(tot / n);
}

// ------------------------------------------------------------------------------------------
// single ops
// ------------------------------------------------------------------------------------------

int saev_normalize_w_dec(saev_ctx* c, void* stream) {
    if (!c) return SAEV_INVALID_ARG;
    REQUIRE(c, c->params, SAEV_NOT_BOUND, "parameters not bound");
    if (!c->cfg.normalize_w_dec) return SAEV_OK;
    HIPCHK(c, launch_normalize_rows(c->params + c->off_W_dec, c->cfg.d_sae, c->cfg.d_model, (hipStream_t)stream));
    return SAEV_OK;
}

// Points the *_c members at this context's own x-derived buffers, or at the leader's when they describe exactly this
// batch (same pointer, same row count, built since this context last borrowed them).  Returns true when borrowed.
static bool bind_x_sources(saev_ctx* c, const float* x, int n, bool allow_borrow) {
    saev_ctx* l = c->leader;
    const bool borrow = allow_borrow && l != nullptr && l->xprep_x == x && l->xprep_n == n && l->xprep_serial != c->leader_serial_seen;
    saev_ctx* src = borrow ? l : c;
    c->upper_c = src->upper; c->mu_c = src->mu; c->xnorm_c = src->xnorm; c->xabs_c = src->xabs_part; c->xs_c = src->xs;
    // (a streamed step of the lender has moved its mu on already: the centre of the images it lends is the copy it kept)
    c->borrow_streamed = borrow && l->fwd_streamed && l->mu_keep != nullptr;
    if (c->borrow_streamed) c->mu_c = l->mu_keep;
    // (the slice route of the refinement needs the source's slice-major x as well: a leader without it sends this step down
    // the row route)
    c->fwd_step = c->fwd_slices && (src == c || src->fwd_slices);
    c->xS_c = c->fwd_step ? src->xS : nullptr;
    if (borrow) c->leader_serial_seen = l->xprep_serial;
    return borrow;
}

// saev_wenc_ready_event: the encoder half of the parameters may still be arriving on another stream; everything that
// depends on x alone has been enqueued by the time this is called
static int wait_wenc(saev_ctx* c, hipStream_t s) {
    if (c->wenc_ready != nullptr) {
        hipEvent_t ev = c->wenc_ready;
        c->wenc_ready = nullptr;
        HIPCHK(c, hipStreamWaitEvent(s, ev, 0));
    }
    return SAEV_OK;
}

// operand preparation for the f16 encoders: x and W_enc^T rewritten as fp16 / bf16 images (no-op for the f32 encoder).
// `xmax_dev` = device scalar max|x| when the caller has it already (the step computes it for the MSE), else NULL.
static int prepare_encoder(saev_ctx* c, const float* x, int n, int32_t* pre_flag, hipStream_t s,
                           const float* xmax_dev = nullptr, bool x_borrowed = false, bool defer_margins = false) {
    if (c->cfg.encoder_mode == SAEV_ENCODER_F32) return SAEV_OK;
    const int D = c->cfg.d_model, S = c->cfg.d_sae;
    const bool bf = c->cfg.encoder_mode == SAEV_ENCODER_BF16;
    if (c->cfg.encoder_mode == SAEV_ENCODER_F16R) {
        // One pass over W_enc (split_wT) yields the fp16 images, W_enc^T in fp32 for the exact refinement (in the
        // gradient scratch dW_encT, free until the backward), mu W_enc and the column norms.  Its power-of-two scale
        // comes from the previous step's largest column norm; f16r_check sends the step down the dense route if the
        // current parameters do not fit that scale.  Only the first use needs a pass of its own for the norm.
        if (!c->wmax_known) {
            { int rcw = wait_wenc(c, s); if (rcw != SAEV_OK) return rcw; }
            HIPCHK(c, launch_transpose(c->params + c->off_W_enc, c->dW_encT, D, S, s));
            HIPCHK(c, launch_wnorm_max(c->dW_encT, S, D, c->wnorm_scratch, c->wmax_prev, s));
            c->wmax_known = true;
        }
        // centre the first pass on the batch's column mean: h = (x - mu) W + (mu W + b)
        // (mu = column sums / n, scaled in the same kernel so that every consumer sees the same fp32 values)
        if (!x_borrowed) {
            if (!c->mu_ready) { HIPCHK(c, launch_colsum(x, n, D, c->colsum_partials, c->mu, 0, nullptr, s, 0, 1.0f / (float)n)); c->mu_serial++; }
            HIPCHK(c, launch_center_stats(x, c->mu, n, D, c->xnorm, c->xabs_part, s, xmax_dev));
        }
        c->mu_ready = false;
        // (the x scale depends on x alone: a borrowing context recomputes the same value from the leader's maxima, next
        // to its own W scale.  Folding this reduction into center_stats_kernel's last workgroup was tried: a release fence
        // per workgroup of four rows took that kernel from 12 to 115 us)
        if (c->borrow_streamed)  // (the lender's images carry the scale of ITS previous batch, not this batch's maxima)
            HIPCHK(c, launch_follower_scales(c->leader->xside_keep, c->wmax_prev, scl(c), pre_flag != nullptr ? pre_flag : c->flags, 0, s));
        else
        HIPCHK(c, launch_f16r_scales(c->xabs_c, (n + 3) / 4, c->wmax_prev, scl(c), s));
        // (the slice-major W_enc^T: in the gradient scratch, free until the backward -- or, where the streamed step may follow, in a
        // buffer of its own, so that it survives the backward)
        float* const wt_out = (c->fwd_step && c->stream_ok) ? c->WeS : c->dW_encT;
        if (!x_borrowed && c->wenc_ready == nullptr) {
            // the usual case: nobody's parameter all-gather to wait for in between -- both image passes in one launch
            HIPCHK(c, launch_split_f16r(x, n, D, c->Dp, c->xs, scl(c), c->mu, c->params + c->off_W_enc, S, c->S_pad, c->ws,
                                        reinterpret_cast<double*>(c->dot_part), c->sq_part, wt_out, s, c->fwd_step ? c->xS : nullptr,
                                        c->fwd_step ? 1 : 0));
        } else {
            if (!x_borrowed) HIPCHK(c, launch_split_rows(x, n, D, c->Dp, c->xs, 2, s, 1.0f, scl(c), c->mu, c->fwd_step ? c->xS : nullptr));
            { int rcw = wait_wenc(c, s); if (rcw != SAEV_OK) return rcw; }  // x is prepared; from here on W_enc / b_enc are read
            HIPCHK(c, launch_split_wT(c->params + c->off_W_enc, D, S, c->S_pad, c->Dp, 1.0f, c->ws, 2, s, scl(c) + 1,
                                      c->mu_c, reinterpret_cast<double*>(c->dot_part), c->sq_part, wt_out, c->fwd_step ? 1 : 0));
        }
        // what this pass leaves describes W_enc as it is now, centred on this context's current mu: a streamed forward may follow
        // while neither moves (a borrowed centre belongs to the leader: no streamed step there)
        c->wimg_fresh = c->stream_ok && c->fwd_step && !x_borrowed && c->leader == nullptr;
        c->wimg_mu_serial = c->mu_serial;
        HIPCHK(c, launch_bias_finish(reinterpret_cast<const double*>(c->dot_part), c->sq_part, c->Dp, S, c->S_pad,
                                     scl(c) + 1, c->params + c->off_b_enc, c->b_shift, c->wnorm_scratch, s, c->b_seen));
        // (defer_margins: the caller's launch_pre_encode forms the margins together with the encoder's per-launch state)
        if (!defer_margins)
            HIPCHK(c, launch_row_margins(c->xnorm_c, n, D, c->wnorm_scratch, (S + 255) / 256, scl(c), pre_flag,
                                         c->wmax_prev, c->row_margin, s));
        return SAEV_OK;
    }
    if (!x_borrowed) HIPCHK(c, launch_split_rows(x, n, D, c->Dp, c->xs, bf ? 1 : 0, s));
    { int rcw = wait_wenc(c, s); if (rcw != SAEV_OK) return rcw; }
    // (bf16: the fused Adam of the previous step has left the images of the W_enc it wrote -- AdamImageArgs::mode 1 -- and nothing
    // has written the parameters since: include/saev_amd.h, PARAMETER OWNERSHIP)
    c->fwd_reused_wimg = bf && c->wimg_bf16_fresh;
    if (!(bf && c->wimg_bf16_fresh))
        HIPCHK(c, launch_split_wT(c->params + c->off_W_enc, D, S, c->S_pad, c->Dp, bf ? 1.0f : 256.0f, c->ws, bf ? 1 : 0, s));
    if (bf && c->dbg.prep_route == 0 && c->Dp == D && D % 32 == 0) c->wimg_bf16_fresh = true;  // (the images describe W_enc as it is)
    return SAEV_OK;
}

static int run_encoder(saev_ctx* c, const float* x, int n, int epi, float* h_out, const int32_t* flag, int when,
                       hipStream_t s, bool predicted = false) {
    // F16R: only the TopK pass is approximate-then-refined; a dense h must be exact, so it comes from the fp32 kernel
    const bool f16r = c->cfg.encoder_mode == SAEV_ENCODER_F16R;
    if (c->cfg.encoder_mode != SAEV_ENCODER_F32 && !(f16r && epi == EPI_DENSE)) {
        const bool bf = c->cfg.encoder_mode == SAEV_ENCODER_BF16;
        EncodeF16Args a{};
        a.xs = c->xs_c; a.ws = c->ws;
        a.b_enc = f16r ? c->b_shift : c->params + c->off_b_enc;  // f16r: images are centred, the bias carries mu W
        a.n_rows = n; a.Dp = c->Dp; a.S = c->cfg.d_sae; a.w_scale = (bf || f16r) ? 1.0f : 256.0f;
        a.scale_dev = f16r ? scl(c) : nullptr;
        a.arith = bf ? 1 : (f16r ? 2 : 0);
        a.row_margin = f16r ? c->row_margin : nullptr;
        const int enc_wgs = c->dbg.enc_wgs > 0 ? c->dbg.enc_wgs : 256;
        a.mfma32 = c->dbg.enc_mfma == 32 ? 1 : 0;
        a.s_splits = encoder_splits(n, a.S, encode_f16x3_tile_rows(), encode_f16x3_tile_latents(), enc_wgs);
        a.no_rot = c->dbg.enc_rot == 1 ? 1 : 0;
        a.h_out = h_out;
        a.ngroups = f16_ngroups(c); a.top_k = c->cfg.top_k;
        a.gmax = c->gmax; a.gmax_stride = c->gmax_stride; a.cand_cnt = c->cand_cnt; a.cand_val = c->cand_val; a.cand_idx = c->cand_idx;
        a.cand_cap = CAND_CAP; a.cand_stride = CAND_STRIDE;
        a.enable_flag = flag; a.enable_when = when;
        if (predicted) { a.heur_z = c->heur_state; a.tau_max = c->tau_max; }
        {
            const int rf = c->dbg.refresh_first > 0 ? c->dbg.refresh_first : 8, re = c->dbg.refresh_every;
            a.refresh_first = std::max(1, rf);
            a.refresh_every = (re >= 1 && (re & (re - 1)) == 0) ? re : 2;
            // the 64-group variant (32 < k <= 64, e.g. 82 k latents at k = 64) refreshes on every tile: with twice the codes
            // per row and many more tiles per workgroup its lists would outgrow their 4 096 entries otherwise
            if (a.ngroups > 32 && re <= 0) a.refresh_every = 1;
        }
        HIPCHK(c, launch_encode_f16x3(a, epi, s));
        return SAEV_OK;
    }
    EncodeArgs a{};
    a.x = x;
    a.W_enc = c->params + c->off_W_enc;
    a.b_enc = c->params + c->off_b_enc;
    a.n_rows = n;
    a.D = c->cfg.d_model;
    a.S = c->cfg.d_sae;
    a.s_splits = encoder_splits(n, a.S, encode_gemm_tile_rows(), encode_gemm_tile_latents(), 512);
    a.h_out = h_out;
    a.ngroups = c->cfg.top_k <= 32 ? 32 : 64;
    a.gmax = c->gmax;
    a.gmax_stride = c->gmax_stride;
    a.cand_cnt = c->cand_cnt;
    a.cand_val = c->cand_val;
    a.cand_idx = c->cand_idx;
    a.cand_cap = CAND_CAP; a.cand_stride = CAND_STRIDE;
    a.enable_flag = flag;
    a.enable_when = when;
    HIPCHK(c, launch_encode_gemm(a, epi, s));
    return SAEV_OK;
}

int saev_encode_dense(saev_ctx* c, const float* x, int32_t n, float* h_out, void* stream) {
    if (!c) return SAEV_INVALID_ARG;
    REQUIRE(c, c->params, SAEV_NOT_BOUND, "parameters not bound");
    REQUIRE(c, x && h_out && n > 0, SAEV_INVALID_ARG, "saev_encode_dense: bad arguments");
    REQUIRE(c, n <= c->cfg.max_batch || c->cfg.encoder_mode == SAEV_ENCODER_F32, SAEV_INVALID_ARG,
            "saev_encode_dense: n_rows > max_batch");
    bind_x_sources(c, x, n, false);
    c->xprep_x = nullptr;  // the images below are rebuilt for this call: nothing to lend
    if (c->cfg.encoder_mode != SAEV_ENCODER_F16R) {  // (f16r: a dense h comes from the fp32 kernel, no images needed)
        int rc = prepare_encoder(c, x, n, nullptr, (hipStream_t)stream);
        if (rc != SAEV_OK) return rc;
    }
    return run_encoder(c, x, n, EPI_DENSE, h_out, nullptr, 0, (hipStream_t)stream);
}

int saev_topk_dense(saev_ctx* c, const float* h, int32_t n, int32_t k, const int32_t* mask, int32_t* idx_out,
                    float* val_out, void* stream) {
    if (!c) return SAEV_INVALID_ARG;
    REQUIRE(c, h && idx_out && val_out && n > 0 && k > 0, SAEV_INVALID_ARG, "saev_topk_dense: bad arguments");
    REQUIRE(c, k <= c->cfg.d_sae, SAEV_INVALID_ARG, "saev_topk_dense: k > d_sae");
    SelectDenseArgs a{};
    a.h = h; a.n_rows = n; a.S = c->cfg.d_sae; a.k = k; a.mask = mask;
    a.idx_out = idx_out; a.val_out = val_out; a.out_stride = k;
    HIPCHK(c, launch_select_dense(a, (hipStream_t)stream));
    return SAEV_OK;
}

// encode + top-k into (idx_out, val_out); fused path with exact dense fallback on overflow
static int encode_topk_impl(saev_ctx* c, const float* x, int n, int32_t* idx_out, float* val_out,
                            const int32_t* pre_flag, hipStream_t s, const float* xmax_dev = nullptr, bool x_borrowed = false) {
    const int K = c->cfg.top_k;
    int32_t* need_dense = c->flags + 1;
    const bool f16r_mode = c->cfg.encoder_mode == SAEV_ENCODER_F16R;
    const bool predict_mode = fused_supported(c->cfg) && c->cfg.bound_mode != 0 && c->cfg.encoder_mode != SAEV_ENCODER_F32 &&
                              f16_ngroups(c) == 32;
    const bool one_launch_pre = fused_supported(c->cfg) && !predict_mode;  // margins + encoder state + list flags in one launch
    if (c->follow_stream) {
        // nothing to prepare: the x side is the lender's, the W side this context's own Adam has left (its W scale with it)
        HIPCHK(c, launch_follower_scales(c->leader->xside_keep, c->wmax_prev, scl(c), const_cast<int32_t*>(pre_flag), 1, s));
    } else if (!c->stream_step) {
        int rc0 = prepare_encoder(c, x, n, const_cast<int32_t*>(pre_flag), s, xmax_dev, x_borrowed, one_launch_pre);
        if (rc0 != SAEV_OK) return rc0;
        rc0 = wait_wenc(c, s);  // (the f32 encoder has no preparation: it reads W_enc from here on)
        if (rc0 != SAEV_OK) return rc0;
    }
    if (fused_supported(c->cfg)) {
        const int ng = c->cfg.encoder_mode == SAEV_ENCODER_F32 ? (c->cfg.top_k <= 32 ? 32 : 64) : f16_ngroups(c);
        const bool f16r = c->cfg.encoder_mode == SAEV_ENCODER_F16R;
        // select -> (f16r: exact refinement -> select) on the candidate lists, predicated on `flag == when`
        auto select_stage = [&](const int32_t* flag, int when, int32_t* bad, const int32_t* tau_max, int32_t* ovf = nullptr,
                                const int32_t* first_flag = nullptr) -> int {
            // ovf / first_flag: the first select of the stage also does what overflow_check_kernel did (it is predicated on
            // first_flag, the flag known before the encoder ran; the kernels after it on `flag`, which it may raise)
            SelectCandArgs sc{};
            sc.cand_cnt = c->cand_cnt; sc.cand_val = c->cand_val; sc.cand_idx = c->cand_idx;
            sc.cand_cap = CAND_CAP; sc.cand_stride = CAND_STRIDE; sc.n_rows = n; sc.k = K;
            sc.idx_out = idx_out; sc.val_out = val_out; sc.out_stride = K;
            sc.enable_flag = first_flag ? first_flag : flag; sc.enable_when = when;
            sc.tau_max = tau_max; sc.invalid = bad; sc.ovf = ovf;
            if (f16r) {
                // approximate values: (1) survivors of the cut lowered by the row margin, (2) their exact fp32 values,
                // (3) the final cut on exact values.  A row with more than REFINE_CAP survivors raises `bad`.
                sc.row_margin = c->row_margin; sc.x = x; sc.W_encT = c->dW_encT; sc.b_enc = c->params + c->off_b_enc;
                sc.D = c->cfg.d_model; sc.refine_overflow = bad;
                sc.surv_idx = c->surv_idx; sc.surv_val = c->surv_val; sc.surv_cnt = c->surv_cnt;
                if (c->fwd_step) { sc.surv_rng = c->surv_rng; sc.lat_range = c->rs_lat_range; sc.n_ranges = c->rs_n_ranges; }
                // saev_debug_cfg.fused_chain: survivors, their exact values and the final cut in ONE launch (select_refine_kernel).
                // Opt-in: measured 335 us against 351 for the three kernels when all of them run at seven waves per SIMD,
                // and slower than them (+0.03 ms per step) once lists of 1 025-2 048 entries stay in registers, which the
                // survivor select needs (tools/experiments/README.md); a survivor overflow raises `bad` (= need_dense) like
                // a list overflow does, and the dense route that follows redoes the step exactly
                const bool chain_fused = c->dbg.fused_chain != 0;
                if (chain_fused && tau_max == nullptr && first_flag != nullptr) {
                    HIPCHK(c, launch_select_refine(sc, s));
                    return SAEV_OK;
                }
                HIPCHK(c, launch_select_cand(sc, s));
                sc.enable_flag = flag; sc.ovf = nullptr;
                if (c->fwd_step) {  // exact values from 32-column slices of W_enc^T that the XCD L2s hold (select.hip)
                    RefineSlicesArgs rs{};
                    rs.surv_idx = c->surv_idx; rs.surv_cnt = c->surv_cnt; rs.surv_val = c->surv_val; rs.surv_rng = c->surv_rng;
                    rs.xS = c->xS_c; rs.WeS = c->stream_ok ? c->WeS : c->dW_encT; rs.b_enc = sc.b_enc; rs.part = c->rs_part;
                    rs.n_rows = n; rs.S = c->cfg.d_sae; rs.D = c->cfg.d_model;
                    rs.lat_range = c->rs_lat_range; rs.n_ranges = c->rs_n_ranges;
                    rs.enable_flag = flag; rs.enable_when = when;
                    // (the D / 32 shares of a survivor are added by the final select itself: refine_sum_kernel's launch and its
                    // round trip through surv_val are gone; saev_debug_cfg.fwd_route = 2 keeps the separate pass)
                    const bool fold = c->dbg.fwd_route != 2;
                    HIPCHK(c, launch_refine_slices(rs, s, !fold));
                    if (fold) { sc.sum_part = c->rs_part; sc.sum_bias = sc.b_enc; sc.sum_n = c->cfg.d_model / RS_SLICE; sc.sum_plane = (size_t)n * REFINE_CAP; }
                } else {
                    HIPCHK(c, launch_refine_exact(sc, s));
                }
                sc.row_margin = nullptr; sc.tau_max = nullptr;
                sc.cand_cnt = c->surv_cnt; sc.cand_val = c->surv_val; sc.cand_idx = c->surv_idx; sc.cand_cap = REFINE_CAP; sc.cand_stride = REFINE_CAP;
            }
            HIPCHK(c, launch_select_cand(sc, s));
            return SAEV_OK;
        };
        // Predicted bounds (gemm_encode_f16x3.hip, heur_z) for the fp16-image encoders with k <= 32: first a launch whose row
        // bounds are a prediction, verified by the select stage; only if that fails anywhere -- flag `bad1` -- the launch
        // with guaranteed bounds, which is otherwise skipped on the device (every kernel of it exits at once).
        const bool predict = c->cfg.bound_mode != 0 && ng == 32 && c->cfg.encoder_mode != SAEV_ENCODER_F32;
        int32_t *bad1 = c->flags + 9, *run2 = c->flags + 10, *gate = c->flags + 11;
        if (predict) {
            HIPCHK(c, launch_heur_gate(c->heur_state, pre_flag, gate, s));  // gate: no prediction this time
            HIPCHK(c, launch_encoder_init(c->cand_cnt, n, c->gmax, 0, s, c->tau_max));
            timing_begin(c, s);  // the events bracket the encoder kernel alone
            int rc = run_encoder(c, x, n, EPI_TOPK, nullptr, gate, 0, s, true);
            if (rc != SAEV_OK) return rc;
            timing_end(c, s);
            HIPCHK(c, launch_overflow_check(c->cand_cnt, n, CAND_CAP, gate, bad1, c->flags + 2, c->flags + 3, s, need_dense,
                                            nullptr, c->heur_state + 3, nullptr, pre_flag));
            rc = select_stage(bad1, 0, bad1, c->tau_max);
            if (rc != SAEV_OK) return rc;
            HIPCHK(c, launch_heur_update(c->heur_state, bad1, c->heur_state + 3, K, gate, s));
            // the retry with guaranteed bounds
            HIPCHK(c, launch_encoder_init(c->cand_cnt, n, c->gmax, ng * c->gmax_stride, s, nullptr, bad1, 1));
            rc = run_encoder(c, x, n, EPI_TOPK, nullptr, bad1, 1, s);
            if (rc != SAEV_OK) return rc;
            HIPCHK(c, launch_overflow_check(c->cand_cnt, n, CAND_CAP, pre_flag, need_dense, c->flags + 2, c->flags + 3, s, need_dense,
                                            run2, nullptr, bad1));
            rc = select_stage(run2, 1, need_dense, nullptr);
            if (rc != SAEV_OK) return rc;
        } else {
            const int S_ = c->cfg.d_sae;
            if (c->stream_step) {
                // the streamed preparation: one pass over x (gathered from the pool on the way in, if the caller handed a pool),
                // then one small launch; W_enc is not read at all (its images were left by the previous step's Adam)
                const int D_ = c->cfg.d_model;
                XprepArgs xp{};
                xp.x = c->gather_pool != nullptr ? c->gather_pool : x; xp.rows = c->gather_rows; xp.x_out = c->gather_pool != nullptr ? const_cast<float*>(x) : nullptr;
                xp.n = n; xp.D = D_; xp.nks = D_ / 32; xp.n_pad = c->MB_pad; xp.scales = scl(c); xp.mu = c->mu; xp.xs = c->xs; xp.xS = c->xS;
                xp.xn_part = c->xn_part; xp.col_part = c->colsum_partials; xp.amax_part = c->amax_part; xp.cmax_part = c->cmax_part;
                xp.W_enc = c->params + c->off_W_enc; xp.WeS = c->WeS; xp.b_enc = c->params + c->off_b_enc; xp.b_seen = c->b_seen;
                xp.S = S_; xp.salt = ++c->stale_salt; xp.stale = c->flags + 12;
                if (!c->followers.empty()) { xp.mu_keep = c->mu_keep; xp.xside_keep = c->xside_keep; }
                HIPCHK(c, launch_xprep(xp, s));
                PreEncode2Args pe{};
                pe.cand_cnt = c->cand_cnt; pe.n_rows = n; pe.gmax = c->gmax; pe.n_gmax = ng * c->gmax_stride;
                pe.xn_part = c->xn_part; pe.nks = D_ / 32; pe.n_pad = c->MB_pad; pe.D = D_;
                pe.wg_part = c->wnorm_scratch; pe.n_part = (S_ + 255) / 256; pe.scales = scl(c); pe.scales_next = scl_next(c);
                pe.pre_flag = const_cast<int32_t*>(pre_flag); pe.wmax_prev = c->wmax_prev; pe.margin = c->row_margin; pe.xnorm = c->xnorm;
                pe.flags1 = c->flags + 1; pe.col_part = c->colsum_partials; pe.n_rowblk = (n + 255) / 256; pe.mu = c->mu;
                pe.inv_n = 1.0f / (float)n; pe.update_mu = c->train_fused ? 1 : 0;
                pe.amax_part = c->amax_part; pe.cmax_part = c->cmax_part; pe.n_img = ((n + 255) / 256) * (D_ / 32);
                pe.upper = c->upper; pe.stats = c->stats; pe.stale = c->flags + 12; pe.stale_host = c->stale_dev;
                pe.xside_keep = c->followers.empty() ? nullptr : c->xside_keep;
                HIPCHK(c, launch_pre_encode2(pe, s));
                if (c->train_fused) c->mu_serial++;
            } else
            HIPCHK(c, launch_pre_encode(c->cand_cnt, n, c->gmax, ng * c->gmax_stride, f16r_mode ? c->xnorm_c : nullptr, c->cfg.d_model,
                                        c->wnorm_scratch, (S_ + 255) / 256, f16r_mode ? scl(c) : nullptr, const_cast<int32_t*>(pre_flag),
                                        c->wmax_prev, c->row_margin, c->flags + 1, s));
            timing_begin(c, s);  // the events bracket the encoder kernel alone
            int rc = run_encoder(c, x, n, EPI_TOPK, nullptr, pre_flag, 0, s);
            if (rc != SAEV_OK) return rc;
            timing_end(c, s);
            rc = select_stage(need_dense, 0, need_dense, nullptr, c->flags + 1, pre_flag);
            if (rc != SAEV_OK) return rc;
        }
    } else {
        HIPCHK(c, launch_init_i32(need_dense, 1, 1, s));
        HIPCHK(c, hipMemsetAsync(c->flags + 2, 0, 2 * sizeof(int32_t), s));
        timing_begin(c, s);
        timing_end(c, s);
    }
    // exact dense route, predicated on the device flag (list overflow, refinement overflow, or k > 64)
    int rc = run_encoder(c, x, n, EPI_DENSE, c->h_dense, need_dense, 1, s);
    if (rc != SAEV_OK) return rc;
    SelectDenseArgs sd{};
    sd.h = c->h_dense; sd.n_rows = n; sd.S = c->cfg.d_sae; sd.k = K;
    sd.idx_out = idx_out; sd.val_out = val_out; sd.out_stride = K;
    sd.enable_flag = need_dense; sd.enable_when = 1;
    HIPCHK(c, launch_select_dense(sd, s));
    return SAEV_OK;
}

int saev_encode_topk(saev_ctx* c, const float* x, int32_t n, int32_t* idx_out, float* val_out, void* stream) {
    if (!c) return SAEV_INVALID_ARG;
    REQUIRE(c, c->params, SAEV_NOT_BOUND, "parameters not bound");
    REQUIRE(c, x && idx_out && val_out && n > 0 && n <= c->cfg.max_batch, SAEV_INVALID_ARG,
            "saev_encode_topk: bad arguments (n_rows must be in 1..max_batch)");
    hipStream_t s = (hipStream_t)stream;
    HIPCHK(c, hipMemsetAsync(c->flags, 0, sizeof(int32_t), s));
    bind_x_sources(c, x, n, false);
    c->xprep_x = nullptr;
    // An API encode always takes the full preparation: `stream_step` is what the LAST step's forward decided, and the images it
    // streamed from may be stale by now (a parameter write announced through saev_params_touched, an unfused tail); the streamed
    // launches would also clear the step's statistics and max |x|, which are not this call's to touch.
    c->stream_step = false;
    c->follow_stream = false;
    return encode_topk_impl(c, x, n, idx_out, val_out, c->flags, s);
}

int saev_scatter_dense(saev_ctx* c, const int32_t* idx, const float* val, int32_t n, int32_t k, float* f_out,
                       void* stream) {
    if (!c) return SAEV_INVALID_ARG;
    REQUIRE(c, idx && val && f_out && n > 0 && k > 0, SAEV_INVALID_ARG, "saev_scatter_dense: bad arguments");
    HIPCHK(c, launch_scatter_dense(idx, val, n, k, k, c->cfg.d_sae, f_out, (hipStream_t)stream));
    return SAEV_OK;
}

int saev_decode_sparse(saev_ctx* c, const int32_t* idx, const float* val, int32_t n, int32_t k,
                       const int64_t* prefixes_host, int32_t n_prefixes, float* x_hats_out, void* stream) {
    if (!c) return SAEV_INVALID_ARG;
    REQUIRE(c, c->params, SAEV_NOT_BOUND, "parameters not bound");
    REQUIRE(c, idx && val && x_hats_out && n > 0 && k > 0, SAEV_INVALID_ARG, "saev_decode_sparse: bad arguments");
    const int S = c->cfg.d_sae, D = c->cfg.d_model;
    int64_t single = S;
    if (!prefixes_host) { prefixes_host = &single; n_prefixes = 1; }
    REQUIRE(c, n_prefixes >= 1 && prefixes_host[n_prefixes - 1] == S && prefixes_host[0] >= 1, SAEV_INVALID_ARG,
            "prefixes must end at d_sae and start at >= 1");
    for (int p = 1; p < n_prefixes; ++p)
        REQUIRE(c, prefixes_host[p] > prefixes_host[p - 1], SAEV_INVALID_ARG, "prefixes must be strictly increasing");
    // x_hats is (n, P, D).  One decode launch per prefix (cut = prefixes[p]); with P > 1 each prefix is
    // decoded into (n, D) scratch and copied into its strided slot.
    REQUIRE(c, n <= c->cfg.max_batch || n_prefixes == 1, SAEV_INVALID_ARG, "n_rows > max_batch");
    hipStream_t s = (hipStream_t)stream;
    for (int p = 0; p < n_prefixes; ++p) {
        DecodeArgs a{};
        a.x = nullptr;  // reconstruction only
        a.idx = idx; a.val = val; a.code_stride = k; a.k = k;
        a.W_dec = c->params + c->off_W_dec; a.b_dec = c->params + c->off_b_dec;
        a.n_rows = n; a.D = D; a.S = S; a.idx_limit = (int)prefixes_host[p];
        a.x_hat = (n_prefixes == 1) ? x_hats_out : c->g;
        HIPCHK(c, launch_decode(a, s));
        if (n_prefixes > 1)
            HIPCHK(c, hipMemcpy2DAsync(x_hats_out + (size_t)p * D, (size_t)n_prefixes * D * sizeof(float), c->g,
                                       (size_t)D * sizeof(float), (size_t)D * sizeof(float), n,
                                       hipMemcpyDeviceToDevice, s));
    }
    return SAEV_OK;
}

int saev_remove_parallel_grads(saev_ctx* c, void* stream) {
    if (!c) return SAEV_INVALID_ARG;
    REQUIRE(c, c->params && c->grads, SAEV_NOT_BOUND, "parameters/grads not bound");
    if (!c->cfg.remove_parallel_grads) return SAEV_OK;
    HIPCHK(c, launch_rpg(c->grads + c->off_W_dec, c->params + c->off_W_dec, c->cfg.d_sae, c->cfg.d_model,
                         (hipStream_t)stream));
    return SAEV_OK;
}

int saev_gather_rows(saev_ctx* c, const float* pool, const int64_t* rows, int32_t n, float* out, void* stream) {
    if (!c) return SAEV_INVALID_ARG;
    REQUIRE(c, pool && rows && out && n > 0, SAEV_INVALID_ARG, "saev_gather_rows: bad arguments");
    HIPCHK(c, launch_gather_rows(pool, rows, n, c->cfg.d_model, out, (hipStream_t)stream));
    return SAEV_OK;
}

// ------------------------------------------------------------------------------------------
// the step
// ------------------------------------------------------------------------------------------

int saev_step_forward(saev_ctx* c, const float* x, int32_t n, int64_t n_rows_global, int32_t training,
                      void* stream) {
    if (!c) return SAEV_INVALID_ARG;
    REQUIRE(c, c->params, SAEV_NOT_BOUND, "parameters not bound");
    REQUIRE(c, x && n > 0 && n <= c->cfg.max_batch, SAEV_INVALID_ARG,
            "saev_step_forward: n_rows must be in 1..max_batch");
    REQUIRE(c, ((uintptr_t)x % 16) == 0, SAEV_INVALID_ARG, "x must be 16-byte aligned");
    REQUIRE(c, n_rows_global >= n, SAEV_INVALID_ARG, "n_rows_global < n_rows");
    hipStream_t s = (hipStream_t)stream;
    const int S = c->cfg.d_sae, D = c->cfg.d_model, K = c->cfg.top_k;
    c->x_last = x;
    c->n_last = n;
    c->training_last = training;
    c->ov_x = nullptr; c->ov_n = 0;  // an override serves one backward
    c->unused_valid = false;
    // The reference renormalises the rows of W_dec at the top of a training step (train.py:334-335).  Nothing before the
    // decode reads W_dec, so it is done right in front of the decode instead: the rows it has just written are what the
    // decode gathers next (3.053 -> 3.034 ms per step against doing it first), and a caller whose decoder half of the
    // parameters is still arriving on another stream (saev_wdec_ready_event: the sharded tail's all-gather) is waited for
    // only there -- the encoder hides the transfer.
    hipEvent_t wdec_ev = c->wdec_ready;
    c->wdec_ready = nullptr;
    // everything that depends on x alone comes from the context this one shares its batches with, if that one has just
    // built it for this very batch (saev_share_x); otherwise it is built here
    const bool borrowed = bind_x_sources(c, x, n, true);
    // The streamed preparation: this context neither lends nor borrows, a previous batch has left a centre, a scale
    // and a normaliser, and the operand images of W_enc describe the parameters as they are, centred on that very centre.
    if (c->stale_host != nullptr && reinterpret_cast<volatile int32_t*>(c->stale_host)[1] != 0) {
        // the fused Adam of an earlier step read W_enc tiles that were not the ones it had written: that step's forward ran on
        // operand images of other values.  Nothing can be redone: say so, loudly; the next forward prepares from scratch.
        const int n_tiles = reinterpret_cast<volatile int32_t*>(c->stale_host)[1];
        reinterpret_cast<volatile int32_t*>(c->stale_host)[1] = 0;
        c->wimg_fresh = false; c->wimg_bf16_fresh = false; c->wn2_fresh = false;
        c->err = "W_enc was written outside the library without saev_params_touched (" + std::to_string(n_tiles) +
                 " 32 x 256 tiles changed between two optimizer steps): a recent step encoded with operand images of the OLD values. "
                 "Announce such writes (saev_params_touched / SaeEngine.params_touched) or make them through torch in-place "
                 "operations on the parameter tensors; the context prepares from scratch from here on";
        return SAEV_STALE_PARAMS;
    }
    if (c->stale_host != nullptr && *reinterpret_cast<volatile int32_t*>(c->stale_host) != 0) {
        // a streamed step found W_enc / b_enc changed behind its back (it took the exact route itself): prepare from scratch
        *reinterpret_cast<volatile int32_t*>(c->stale_host) = 0;
        c->wimg_fresh = false;
        c->wn2_fresh = false;
    }
    // (a lender streams like a context on its own; what its followers need beyond its second launch it keeps: XprepArgs::mu_keep)
    c->stream_step = c->stream_ok && c->prep_valid && c->wimg_fresh && c->wimg_mu_serial == c->mu_serial && c->leader == nullptr &&
                     c->wenc_ready == nullptr && c->fwd_step && (c->followers.empty() || c->dbg.group_route == 0);
    if (!borrowed) {
        c->fwd_streamed = c->stream_step;
        c->fwd_moves_mu = c->stream_step && c->train_fused;
        c->fwd_mu_serial = c->mu_serial;
    }
    // A follower of a streamed step whose own Adam has left W images centred on that very mu prepares nothing at all.
    c->follow_stream = borrowed && c->borrow_streamed && c->stream_ok && c->wimg_fresh && c->wimg_mu_serial == c->leader->fwd_mu_serial &&
                       c->wenc_ready == nullptr && c->fwd_step && c->cfg.encoder_mode == SAEV_ENCODER_F16R;
    c->fwd_reused_wimg = c->stream_step || c->follow_stream;  // (the bf16 encoder decides in prepare_encoder)
    if (c->gather_pool != nullptr && !c->stream_step)  // (the batch as a contiguous matrix first: every other route reads x itself)
        HIPCHK(c, launch_gather_rows(c->gather_pool, c->gather_rows, n, D, const_cast<float*>(x), s));
    if (c->stream_step) {
        c->xprep_x = nullptr;  // (xprep_kernel / pre_encode2_kernel, enqueued by encode_topk_impl, do all of the below)
    } else if (!borrowed && c->cfg.encoder_mode == SAEV_ENCODER_F16R) {
        // one pass: max|x| for the MSE and the column sums the encoder centres on; the launch that finishes them also clears
        // the step's statistics and the force-dense flag (flags[0])
        c->xprep_x = nullptr;
        HIPCHK(c, launch_colsum_absmax(x, n, D, c->colsum_partials, c->mu, c->xabs_part, c->upper, s, 1.0f / (float)n, c->stats, c->flags));
        c->mu_ready = true;
        c->mu_serial++;
    } else {
        HIPCHK(c, launch_step_zero(c->stats, c->upper, c->flags, s));
        if (!borrowed) {
            c->xprep_x = nullptr;
            HIPCHK(c, launch_absmax(x, (long)n * D, c->upper, s));
        }
    }
    (void)n_rows_global;
    int rc = encode_topk_impl(c, x, n, c->idx, c->val, c->flags, s, c->upper_c, borrowed);
    if (rc != SAEV_OK) return rc;
    if (!borrowed) { c->xprep_x = x; c->xprep_n = n; c->xprep_serial++; }
    if (!borrowed && c->stream_ok && !c->stream_step && c->fwd_step) {
        // a full preparation seeds the streamed one: this batch's x scale is in scl(c)[0] already, its max |x| becomes the normaliser
        HIPCHK(c, hipMemcpyAsync(scl(c) + 4, c->upper, sizeof(float), hipMemcpyDeviceToDevice, s));
        c->prep_valid = true;
    }
    if (wdec_ev != nullptr) HIPCHK(c, hipStreamWaitEvent(s, wdec_ev, 0));
    c->wds_fresh = false;
    if (training) {
        // (a training step whose decode can take the slices: normalize_rows leaves the slice-major copy on its way)
        const bool want_slices = c->WdS != nullptr && c->cfg.normalize_w_dec && c->P == 1 && c->dws_ok;
        c->wn2_fresh = false;
        if (want_slices) {
            HIPCHK(c, launch_normalize_rows(c->params + c->off_W_dec, S, D, s, c->WdS, c->wn2));
            c->wds_fresh = true;
            c->wn2_fresh = c->wn2 != nullptr;
        } else if (c->cfg.normalize_w_dec) {
            HIPCHK(c, launch_normalize_rows(c->params + c->off_W_dec, S, D, s, nullptr, c->wn2));
            c->wn2_fresh = c->wn2 != nullptr;
        }
    }

    DecodeArgs a{};
    a.x = x; a.idx = c->idx; a.val = c->val; a.code_stride = K; a.k = K;
    a.W_dec = c->params + c->off_W_dec; a.b_dec = c->params + c->off_b_dec;
    a.n_rows = n; a.D = D; a.S = S; a.idx_limit = S;
    a.upper = c->upper_c;
    a.gscale = 2.0f / ((float)n * (float)D * (float)c->P);
    a.training = training ? 1 : 0;
    a.g = c->g; a.x_hat = c->x_hat; a.fired = c->fired; a.rowstats = c->rowstats;
    c->dws_rows = 0;
    c->dval_fwd = false;
    // (slice-major copies for the weight gradients: dL/dx_hat always from the decode, x only when split_f16r has not left one)
    if (training && c->dws_ok && (c->P == 1 || c->GS != nullptr)) { a.gS = c->P == 1 ? c->gS : c->GS; a.xS = c->fwd_step ? nullptr : c->xS; c->dws_rows = n; }
    // (... and the products dval, from the decoder rows while the decode holds them in registers)
    if (c->dws_rows == n && c->dval_rows != nullptr && (c->P == 1 || decode_matry_forms_dval(D, K))) { a.dval_out = c->dval_rows; c->dval_fwd = true; }
    // The decode reads every code anyway: it sets the (latent, row) bits of the backward's pair-list build (0.5 M scattered atomics
    // that csc_fill paid 35 us for on their own), provided the bit map is clean at this row pitch -- the previous full backward
    // cleared it behind itself -- and this context's backwards run over its own rows.
    c->bitmap_prefill_words = 0;
    const bool slices_decode = c->P == 1 && c->wds_fresh && a.dval_out != nullptr && a.gS != nullptr;
    if (training && c->dbg.csc_route == 0 && c->bitmap != nullptr && c->bitmap_clean && !c->last_backward_gathered && !slices_decode) {
        const int words = ((n + 31) / 32 + 7) / 8 * 8;
        if (words <= c->bitmap_clean_words) {
            a.csc_bitmap = c->bitmap; a.csc_words = words;
            c->bitmap_prefill_words = words; c->bitmap_prefill_rows = n;
            c->bitmap_clean = false;
        }
    }
    if (slices_decode) {
        DecodeSliceArgs ds{};
        ds.d = a; ds.WdS = c->WdS; ds.part = c->dec_part; ds.dvp = c->dvp; ds.dvp_pitch = (long)c->back_rows * K;
        HIPCHK(c, launch_decode_slices(ds, s));
    } else if (c->P > 1) {
        MatryArgs m{};
        m.P = c->P;
        for (int p = 0; p < c->P; ++p) m.cuts[p] = c->cuts[p];
        m.G = c->G;
        m.g_rows_all = c->fused_forward ? 0 : 1;  // (saev_train_step's own backward reads the slice-major copy and block 0 alone)
        HIPCHK(c, launch_decode_matry(a, m, s));
    } else {
        HIPCHK(c, launch_decode(a, s));
    }
    c->P_last = c->P;
    for (int p = 0; p < c->P; ++p) c->cuts_last[p] = c->cuts[p];  // a later saev_set_prefixes must not reach this step's backward
    // (list statistics from the candidate counters themselves unless the fused encoder is out of play or predicts bounds,
    // where overflow_check_kernel leaves them in flags[2..3])
    const bool lists = fused_supported(c->cfg) && !(c->cfg.bound_mode != 0 && c->cfg.encoder_mode != SAEV_ENCODER_F32 && f16_ngroups(c) == 32);
    c->stats_pending = false;
    if (c->train_fused && training) {  // (saev_train_step: the tracker update that follows takes this reduction into its launch)
        c->stats_pending = true;
        c->stats_lists = lists;
        return SAEV_OK;
    }
    HIPCHK(c, launch_stats_reduce(c->rowstats, n, D, c->P, c->cfg.alpha, 0, c->upper_c, c->flags + 2, c->stats, s, nullptr, c->stats_scratch,
                                  lists ? c->cand_cnt : nullptr, CAND_CAP));
    return SAEV_OK;
}

namespace {

int alloc_aux_buffers(saev_ctx* c, int cap) {
    const size_t MB = c->cfg.max_batch, D = c->cfg.d_model;
    const size_t capA = std::max(cap, AUX_SMALL_MAX);  // the few-dead-latents kernels use AUX_SMALL_MAX columns / rows
    auto grab = [&](size_t bytes) -> void* {
        void* q = nullptr;
        if (hipMalloc(&q, bytes) != hipSuccess) return nullptr;
        c->aux_allocs.push_back(q);
        c->aux_bytes += bytes;
        return q;
    };
    c->Wenc_dead = (float*)grab(D * capA * 4);
    c->Wdec_dead = (float*)grab(capA * D * 4);
    c->H_dead = (float*)grab(MB * capA * 4);
    c->A_dead = (float*)grab(MB * capA * 4);
    c->A_mask = (uint8_t*)grab(MB * capA);
    c->dWd = (float*)grab(capA * D * 4);
    c->dWe = (float*)grab(capA * D * 4);
    c->dbe = (float*)grab(capA * 4);
    c->aux_partials = (float*)grab(((MB + 63) / 64) * capA * 4);
    // (the matrix-core kernels take dead sets up to AUX_MFMA_MAX where the compact buffers hold that many rows and the step's
    // backward runs over this context's own rows: gathered backwards -- max_backward_rows -- keep the round-5 limit)
    const size_t mcap = (aux_mfma_supported((int)D) && capA >= (size_t)AUX_MFMA_MAX && c->cfg.max_backward_rows == 0 && c->dbg.aux_wide_route == 0)
                            ? (size_t)AUX_MFMA_MAX : (size_t)AUX_SMALL_MAX;
    c->aux_mfma_cap = (int)mcap;
    c->WencT_dead = (float*)grab(mcap * D * 4);
    c->aux_small_part = (float*)grab(((MB + 63) / 64) * (size_t)2 * mcap * D * 4);
    c->aux_small_part2 = (float*)grab((size_t)(((MB + 63) / 64 + 63) / 64) * mcap * D * 4);
    c->aux_small_partbe = (float*)grab((size_t)((MB + 63) / 64) * mcap * 4);  // (aux_mfma_wgrad_kernel: the blocks' column sums of dA)
    bool fast_ok = true;
    {  // operand images of the five contractions (every encoder mode runs them on the split-fp16 MFMA kernel)
        const size_t cap256 = ((size_t)cap + 255) / 256 * 256, D256 = (D + 255) / 256 * 256;
        c->aux_Dp2 = (int)(((size_t)cap + 31) / 32 * 32);
        c->aux_ws1 = (_Float16*)grab(cap256 * 2 * c->Dp * sizeof(_Float16));          // W_enc[:, dl]^T, later W_dec[dl]
        c->aux_ws2 = (_Float16*)grab(D256 * 2 * c->aux_Dp2 * sizeof(_Float16));        // W_dec[dl] as a (n_dead x D) "encoder"
        c->aux_xsA = (_Float16*)grab((size_t)c->MB_pad * 2 * c->aux_Dp2 * sizeof(_Float16));
        c->aux_xsg = (_Float16*)grab((size_t)c->MB_pad * 2 * c->Dp * sizeof(_Float16));
        c->bias_dead = (float*)grab(cap256 * sizeof(float));
        // weight gradients: contraction over the batch axis, split into AUX_KSPLIT_MAX slices at most
        c->aux_kpad = (int)((MB + 16 * AUX_KSPLIT_MAX - 1) / (16 * AUX_KSPLIT_MAX) * (16 * AUX_KSPLIT_MAX));
        c->aux_kA = (_Float16*)grab(cap256 * 2 * (size_t)c->aux_kpad * sizeof(_Float16));   // A^T, later dA^T
        c->aux_kD = (_Float16*)grab(D256 * 2 * (size_t)c->aux_kpad * sizeof(_Float16));     // g_aux^T, later x^T
        c->aux_kX = (_Float16*)grab(D256 * 2 * (size_t)c->aux_kpad * sizeof(_Float16));     // x^T when the forward writes both forms of x at once (split_both_kernel)
        c->aux_parts = (float*)grab((size_t)AUX_KSPLIT_MAX * cap * D * sizeof(float));
        fast_ok = c->aux_ws1 && c->aux_ws2 && c->aux_xsA && c->aux_xsg && c->bias_dead && c->aux_kA && c->aux_kD && c->aux_kX && c->aux_parts;
    }
    if (!fast_ok || !c->Wenc_dead || !c->Wdec_dead || !c->H_dead || !c->A_dead || !c->A_mask || !c->dWd || !c->dWe || !c->dbe ||
        !c->aux_partials || !c->WencT_dead || !c->aux_small_part || !c->aux_small_part2) {
        c->err = "AuxK: out of device memory for the dead-set buffers (lower saev_cfg.aux_dead_cap)";
        return SAEV_HIP_ERROR;
    }
    c->nd_cap = cap;
    return SAEV_OK;
}

// out (n_rows x S_out, row-major) = rows-operand x cols-operand + bias on the f16x3 encoder kernel (dense epilogue):
// the three AuxK contractions whose long axis is the batch are exactly the encoder's shape.  `scale` is the product
// of the power-of-two scales applied to the two operands when they were split.
int dense_f16x3(saev_ctx* c, const _Float16* xs, const _Float16* ws, const float* bias, int n_rows, int Dp, int S_out,
                float scale, float* out, hipStream_t s, const float* scale_dev = nullptr) {
    EncodeF16Args a{};
    a.scale_dev = scale_dev;
    a.xs = xs; a.ws = ws; a.b_enc = bias;
    a.n_rows = n_rows; a.Dp = Dp; a.S = S_out; a.w_scale = scale; a.arith = 0;
    a.s_splits = encoder_splits(n_rows, S_out, encode_f16x3_tile_rows(), encode_f16x3_tile_latents(), 256);
    a.h_out = out;
    a.ngroups = 32;
    a.enable_flag = nullptr; a.enable_when = 0;
    HIPCHK(c, launch_encode_f16x3(a, EPI_DENSE, s));
    return SAEV_OK;
}

// out (R x C) = sum over the long axis k (length K <= aux_kpad) of P[k][r] * Q[k][c] for two k-major fp32 matrices
// P (K x R), Q (K x C): the AuxK weight gradients.  Both are split into hi/lo fp16 images of their transposes
// (split_wT), the contraction is cut into n_split slices that run as one batched launch of the encoder kernel (a single
// slice would leave most CUs idle: R x C is only a few tiles), and the slices are added in a fixed order.
// (slices and padded length of the batch-long contraction of an R x C weight gradient: the images of its operands are laid out for them)
void ksplit_shape(int R, int C, int K, int* n_split_out, int* Kp_out) {
    const int R256 = (R + 255) / 256 * 256, C256 = (C + 255) / 256 * 256;
    const int tiles = (R256 / 256) * (C256 / 256);
    int n_split = 1;
    while (n_split < AUX_KSPLIT_MAX && tiles * n_split < 256) n_split *= 2;
    *n_split_out = n_split;
    *Kp_out = (K + 16 * n_split - 1) / (16 * n_split) * (16 * n_split);  // <= aux_kpad
}
// imgP / imgQ: the operand's k-major images if somebody has written them already (split_both_kernel, with THIS Kp), else NULL
int ksplit_f16x3(saev_ctx* c, const float* P, const float* sP, int R, const float* Q, const float* sQ, int C, int K,
                 float* out, hipStream_t s, const _Float16* imgP = nullptr, const _Float16* imgQ = nullptr) {
    const int R256 = (R + 255) / 256 * 256, C256 = (C + 255) / 256 * 256;
    int n_split, Kp;
    ksplit_shape(R, C, K, &n_split, &Kp);
    if (imgP == nullptr) { HIPCHK(c, launch_split_wT(P, K, R, R256, Kp, 1.0f, c->aux_kA, 0, s, sP)); imgP = c->aux_kA; }
    if (imgQ == nullptr) { HIPCHK(c, launch_split_wT(Q, K, C, C256, Kp, 1.0f, c->aux_kD, 0, s, sQ)); imgQ = c->aux_kD; }
    EncodeF16Args a{};
    a.scale_dev = sP; a.scale_dev_b = sQ;  // (the two operands' scales where their producers left them)
    a.xs = imgP; a.ws = imgQ; a.b_enc = c->zero_bias;
    a.n_rows = R; a.Dp = Kp / n_split; a.S = C; a.w_scale = 1.0f; a.arith = 0;
    a.s_splits = encoder_splits(R, C, encode_f16x3_tile_rows(), encode_f16x3_tile_latents(), 256);
    a.ngroups = 32;
    a.n_batches = n_split; a.blk_imgs = Kp / 16; a.out_bstride = (long)R * C;
    a.h_out = n_split > 1 ? c->aux_parts : out;
    HIPCHK(c, launch_encode_f16x3(a, EPI_DENSE, s));
    if (n_split > 1) HIPCHK(c, launch_sum_parts(c->aux_parts, n_split, (long)R * C, out, s));
    return SAEV_OK;
}

// A handful of dead latents, all of them selected (n_dead <= min(AUX_SMALL_MAX, k_aux)): one row-wise pass instead of the
// dense algebra.  Every kernel takes the count from the device (flags[4]) and exits when it is zero, so this sequence is
// what a step enqueues when the host only knows a bound of the count.
int auxk_small_forward(saev_ctx* c, hipStream_t s, int bound) {
    const int S = c->cfg.d_sae, D = c->cfg.d_model, n = c->n_last;
    const int32_t* nd_dev = c->flags + 4;
    c->aux_small = true;
    c->aux_all = false;
    c->aux_fused = false;
    c->aux_mfma = false;
    if (!c->dead_list_ready) HIPCHK(c, launch_dead_compact(c->dead, S, c->dead_list, s, nd_dev));
    c->aux_ndp = bound > AUX_SMALL_MAX ? AUX_MFMA_MAX : AUX_SMALL_MAX;
    HIPCHK(c, launch_gather_dead_small(c->params + c->off_W_enc, c->params + c->off_W_dec, c->dead_list, nd_dev, D, S,
                                       c->WencT_dead, c->Wdec_dead, s, c->aux_ndp));
    if (bound <= AUX_FUSED_MAX && aux_fused_supported(D) && c->dbg.aux_small_max != AUX_SMALL_MAX) {
        // a handful of dead latents: one pass over x and x_hat leaves the block partials of every gradient of the auxiliary term
        // (partials in the buffers the two-kernel form uses for its own: aux_small_part; g_aux and A_dead are free in this form)
        c->aux_fused = true;
        HIPCHK(c, launch_aux_small_fused(c->x_last, c->x_hat, c->WencT_dead, c->Wdec_dead, c->params + c->off_b_enc,
                                         c->params + c->off_b_dec, c->dead_list, n, D, nd_dev,
                                         c->cfg.alpha * 2.0f / ((float)n * (float)D), c->aux_small_part, c->g_aux, c->A_dead, c->rowstats, s, bound));
        // (inside saev_train_step the backward's ordered-sum launch also forms the step's auxiliary loss: aux_stats_pending)
        c->aux_stats_pending = c->train_fused;
        if (!c->aux_stats_pending)
            HIPCHK(c, launch_stats_reduce(c->rowstats, n, D, c->P_last, c->cfg.alpha, 2, c->upper_c, nullptr, c->stats, s, nd_dev, c->stats_scratch));
        return SAEV_OK;
    }
    if (bound <= c->aux_mfma_cap && aux_mfma_supported(D) && c->dbg.aux_small_route == 0) {
        c->aux_mfma = true;
        HIPCHK(c, launch_aux_mfma_forward(c->x_last, c->x_hat, c->WencT_dead, c->Wdec_dead, c->params + c->off_b_enc,
                                          c->params + c->off_b_dec, c->dead_list, n, D, nd_dev,
                                          c->cfg.alpha * 2.0f / ((float)n * (float)D), c->A_dead, c->H_dead, c->g_aux, c->rowstats, s, bound, c->aux_ndp));
        c->aux_mfma_bound = bound;
        // (inside saev_train_step the backward's ordered-sum launch also forms the step's auxiliary loss, as for the one-pass kernel)
        c->aux_stats_pending = c->train_fused;
        if (c->aux_stats_pending) return SAEV_OK;
    } else
    HIPCHK(c, launch_aux_small_fwd(c->x_last, c->x_hat, c->WencT_dead, c->Wdec_dead, c->params + c->off_b_enc,
                                   c->params + c->off_b_dec, c->dead_list, n, D, nd_dev,
                                   c->cfg.alpha * 2.0f / ((float)n * (float)D), c->A_dead, c->H_dead, c->g_aux, c->rowstats, s));
    HIPCHK(c, launch_stats_reduce(c->rowstats, n, D, c->P_last, c->cfg.alpha, 2, c->upper_c, nullptr, c->stats, s, nd_dev, c->stats_scratch));
    return SAEV_OK;
}

// forward of the auxiliary loss as dense algebra over n_dead_host dead latents (see auxk.hip)
int auxk_forward(saev_ctx* c, hipStream_t s) {
    const int S = c->cfg.d_sae, D = c->cfg.d_model, n = c->n_last;
    const int nd = c->n_dead_host, ku = c->k_use_host;
    const int ndp = (nd + 3) / 4 * 4;
    REQUIRE(c, ndp <= c->nd_cap, SAEV_UNSUPPORTED, "AuxK: more dead latents than the dense buffers hold (raise saev_cfg.aux_dead_cap)");
    int rc = SAEV_OK;
    // Every encoder mode runs the five contractions on the split-fp16 MFMA kernel (three products per fp32 product:
    // fp32-accurate, gemm_encode_f16x3.hip), whatever arithmetic its own encoder uses: the auxiliary loss is defined on
    // the exact pre-activations (the bf16 mode's oracle does the same).  Only the f16x3 mode already has hi/lo x images.
    const bool own_images = c->cfg.encoder_mode != SAEV_ENCODER_F16X3;
    const int ndp256 = (ndp + 255) / 256 * 256, Dp2 = (ndp + 31) / 32 * 32;
    // Operand images in both forms from one pass over their source (split.hip: split_both_kernel): six image launches instead of
    // ten, bit-identical images (saev_debug_cfg.aux_split_route = 1 keeps the ten)
    c->aux_both = own_images && c->dbg.aux_split_route == 0 && D % 4 == 0;
    // aux_dev_count: nd / ku are upper bounds (the tracker record of a few steps ago, saev_step_dead); the true count and
    // min(k_aux, count) are flags[4] / flags[5].  Columns of the dead set past the true count are padding -- zero weights,
    // bias -inf (never selected) or 0 (all-selected mode) -- exactly like the columns that pad nd to a multiple of four,
    // so every product below has its usual shape and nothing is read back.
    const int32_t* nd_dev = c->aux_dev_count ? c->flags + 4 : nullptr;
    const int32_t* ku_dev = c->aux_dev_count ? c->flags + 5 : nullptr;
    if (!c->dead_list_ready) HIPCHK(c, launch_dead_compact(c->dead, S, c->dead_list, s));
    HIPCHK(c, launch_gather_dead(c->params + c->off_W_enc, c->params + c->off_W_dec, c->dead_list, nd, ndp, D, S,
                                 c->Wenc_dead, c->Wdec_dead, s, nd_dev));
    c->aux_small = false;
    c->aux_all = false;
    // n_dead <= k_aux: every dead latent is selected, the codes are H itself (padding columns zero) and there is no mask
    c->aux_all = ku == nd;
    {
        // H = x W_enc[:, dl] + b_enc[dl]: in f16x3 mode the x images of this step are already there (prepare_encoder)
        HIPCHK(c, launch_split_wT(c->Wenc_dead, D, ndp, ndp256, c->Dp, 256.0f, c->aux_ws1, 0, s));
        HIPCHK(c, launch_dead_bias_vec(c->params + c->off_b_enc, c->dead_list, nd, ndp, c->bias_dead, s, c->aux_all, nd_dev));
        const _Float16* xs_hl = c->xs_c;
        if (own_images) {  // the step's x images are single fp16 / bf16 or absent: make the hi/lo ones (the buffer is free until the backward)
            // (with the step's power-of-two x scale, so that no activation magnitude can overflow fp16)
            HIPCHK(c, launch_pow2_scale(c->upper_c, c->aux_scales + 6, s));  // from max|x| of the step (uncentred here)
            if (c->aux_both) {  // ... and its k-major images for the backward's dWe, from the same pass over x
                int ns, Kp;
                ksplit_shape(ndp, D, n, &ns, &Kp);
                HIPCHK(c, launch_split_both(c->x_last, n, D, 1.0f, c->aux_scales + 6, c->aux_xsg, c->Dp, c->aux_kX, Kp, s));
            } else
            HIPCHK(c, launch_split_rows(c->x_last, n, D, c->Dp, c->aux_xsg, 0, s, 1.0f, c->aux_scales + 6));
            xs_hl = c->aux_xsg;
        }
        rc = dense_f16x3(c, xs_hl, c->aux_ws1, c->bias_dead, n, c->Dp, ndp, 256.0f, c->aux_all ? c->A_dead : c->H_dead, s,
                         own_images ? c->aux_scales + 6 : nullptr);
        if (rc != SAEV_OK) return rc;
    }
    const bool fused_select = aux_select_supported(ndp) && c->dbg.aux_dense_route == 0;  // (1: the round-4 select / fill / scatter sequence)
    if (!c->aux_all) {
        SelectDenseArgs sd{};
        sd.h = c->H_dead; sd.n_rows = n; sd.S = ndp; sd.k = ku; sd.k_dev = ku_dev;
        sd.idx_out = c->aux_idx; sd.val_out = c->aux_val; sd.out_stride = c->cfg.k_aux;
        if (fused_select) {
            // codes, mask, max |code| and the codes' operand scale in one launch (auxk.hip: aux_select_kernel)
            HIPCHK(c, launch_aux_select(c->H_dead, n, ndp, ku, ku_dev, c->A_dead, c->A_mask, c->aux_sync, c->aux_scales + 2, s));
        } else {
            HIPCHK(c, launch_select_dense(sd, s));
            HIPCHK(c, hipMemsetAsync(c->A_dead, 0, (size_t)n * ndp * sizeof(float), s));
            HIPCHK(c, hipMemsetAsync(c->A_mask, 0, (size_t)n * ndp, s));
            HIPCHK(c, launch_aux_scatter(c->aux_idx, c->aux_val, n, ku, c->cfg.k_aux, ndp, c->A_dead, c->A_mask, s, ku_dev));
        }
    }
    {
        // E = A W_dec[dl]: rows = batch, contraction over the dead set, "latents" = the d_model outputs
        // (the codes are pre-activations of unknown magnitude: power-of-two scale from their device-side max)
        if (c->aux_all || !fused_select) HIPCHK(c, launch_absmax_pow2(c->A_dead, (long)n * ndp, c->aux_sync, c->aux_scales + 2, s));
        if (c->aux_both) {
            int ns, Kp;
            ksplit_shape(ndp, D, n, &ns, &Kp);
            // the codes as a row operand (E) and k-major (dWd); the dead latents' decoder rows k-major (E) and as a row operand (dA:
            // aux_ws1 is free again, H is done)
            HIPCHK(c, launch_split_both(c->A_dead, n, ndp, 1.0f, c->aux_scales + 2, c->aux_xsA, Dp2, c->aux_kA, Kp, s));
            HIPCHK(c, launch_split_both(c->Wdec_dead, ndp, D, 256.0f, nullptr, c->aux_ws1, c->Dp, c->aux_ws2, Dp2, s));
        } else {
        HIPCHK(c, launch_split_rows(c->A_dead, n, ndp, Dp2, c->aux_xsA, 0, s, 1.0f, c->aux_scales + 2));
        HIPCHK(c, launch_split_wT(c->Wdec_dead, ndp, D, (D + 255) / 256 * 256, Dp2, 256.0f, c->aux_ws2, 0, s));
        }
        rc = dense_f16x3(c, c->aux_xsA, c->aux_ws2, c->zero_bias, n, Dp2, D, 256.0f, c->g_aux, s, c->aux_scales + 2);
    }
    if (rc != SAEV_OK) return rc;
    // (g_aux leaves with its max and the operand scale the backward splits it with: aux_scales + 4)
    HIPCHK(c, launch_aux_resid(c->g_aux, c->x_last, c->x_hat, c->params + c->off_b_dec, n, D,
                               c->cfg.alpha * 2.0f / ((float)n * (float)D), c->rowstats, s, nd_dev, c->aux_sync, c->aux_scales + 4));
    HIPCHK(c, launch_stats_reduce(c->rowstats, n, D, c->P_last, c->cfg.alpha, 1, c->upper_c, nullptr, c->stats, s, nullptr,
                                  c->stats_scratch));
    return SAEV_OK;
}

// gradients of the auxiliary loss, accumulated into the gradient buffer / the transposed W_enc scratch
int auxk_backward(saev_ctx* c, hipStream_t s) {
    const int D = c->cfg.d_model, n = c->n_last;
    const int nd = c->n_dead_host;
    const int ndp = (nd + 3) / 4 * 4;
    float* dA = c->H_dead;  // H is dead after the select
    int rc;
    if (c->aux_small) {  // dA is there already (auxk_small_forward); weight gradients block-wise, then two column sums;
                         // all predicated on the device-side count like the forward (rows past it are never scattered)
        const int nb = (n + 63) / 64, L = AUX_SMALL_MAX;
        const int32_t* nd_dev = c->flags + 4;
        if (c->aux_fused) {  // the forward has left block partials of all four gradients: one launch of ordered sums finishes them
            const int blocks = aux_fused_blocks(n);
            // (the dead count may be zero on the device: db_aux must then read as zeros, and b_dec's gradient stay untouched -- the
            // kernel leaves at once in that case, hence the memset)
            if (c->ov_x != nullptr) HIPCHK(c, hipMemsetAsync(c->db_aux, 0, (size_t)D * sizeof(float), s));
            HIPCHK(c, launch_aux_fused_wsum(c->aux_small_part, blocks, D, nd_dev, c->dWd, c->dWe, s, c->g_aux,
                                            c->ov_x != nullptr ? c->db_aux : c->grads + c->off_b_dec, c->ov_x != nullptr ? 0 : 1, c->A_dead, c->dbe,
                                            c->aux_stats_pending ? c->rowstats : nullptr, n, c->cfg.alpha, c->stats));
            c->aux_stats_pending = false;
            return SAEV_OK;
        }
        if (c->aux_mfma) {
            // weight-gradient partials per block of 64 rows with the blocks' column sums of g_aux and dA riding along; ONE launch of
            // ordered sums finishes all four gradients (and the auxiliary loss inside saev_train_step)
            HIPCHK(c, launch_aux_mfma_wgrad(c->A_dead, dA, c->g_aux, c->x_last, n, D, nd_dev, c->aux_small_part, c->aux_small_part2,
                                            c->aux_small_partbe, s, c->aux_mfma_bound, c->aux_ndp));
            if (c->ov_x != nullptr) HIPCHK(c, hipMemsetAsync(c->db_aux, 0, (size_t)D * sizeof(float), s));  // (the count may be zero on the device)
            HIPCHK(c, launch_aux_fused_wsum(c->aux_small_part, nb, D, nd_dev, c->dWd, c->dWe, s, c->aux_small_part2,
                                            c->ov_x != nullptr ? c->db_aux : c->grads + c->off_b_dec, c->ov_x != nullptr ? 0 : 1, c->aux_small_partbe, c->dbe,
                                            c->aux_stats_pending ? c->rowstats : nullptr, n, c->cfg.alpha, c->stats, c->aux_ndp));
            c->aux_stats_pending = false;
            return SAEV_OK;
        }
        HIPCHK(c, launch_aux_small_wgrad(c->A_dead, dA, c->g_aux, c->x_last, n, D, nd_dev, c->aux_small_part, s));
        HIPCHK(c, launch_aux_small_wsum(c->aux_small_part, nb, D, nd_dev, c->dWd, c->dWe, s));
        HIPCHK(c, launch_colsum(dA, n, L, c->aux_partials, c->dbe, 0, nd_dev, s, 0, 1.0f, 1));
        if (c->ov_x != nullptr) {  // gathered backward: the local share travels with the compact rows (saev_aux_compact_export)
            HIPCHK(c, hipMemsetAsync(c->db_aux, 0, (size_t)D * sizeof(float), s));  // (the count may be zero on the device)
            HIPCHK(c, launch_colsum(c->g_aux, n, D, c->colsum_partials, c->db_aux, 0, nd_dev, s));
        } else {
            HIPCHK(c, launch_colsum(c->g_aux, n, D, c->colsum_partials, c->grads + c->off_b_dec, 1, nd_dev, s));
        }
        return SAEV_OK;
    }
    {
        // dA = g_aux W_dec[dl]^T.  g_aux carries the factor alpha * 2 / (n D) (~1e-10) times a residual of unknown
        // magnitude: bring it to [2^13, 2^14) with an exact power of two from its device-side max before the fp16 split;
        // W_dec[dl] rows are already "latent-major", so they split like x.
        // (the scale from g_aux's device-side max: aux_resid_kernel left it at aux_scales + 4)
        if (c->aux_both) {  // g_aux in both forms (dA here, dWd below); the decoder rows' row-form images are the forward's
            int ns, Kp;
            ksplit_shape(ndp, D, n, &ns, &Kp);
            HIPCHK(c, launch_split_both(c->g_aux, n, D, 1.0f, c->aux_scales + 4, c->aux_xsg, c->Dp, c->aux_kD, Kp, s));
        } else {
        HIPCHK(c, launch_split_rows(c->g_aux, n, D, c->Dp, c->aux_xsg, 0, s, 1.0f, c->aux_scales + 4));
        HIPCHK(c, launch_split_rows(c->Wdec_dead, ndp, D, c->Dp, c->aux_ws1, 0, s, 256.0f));
        }
        rc = dense_f16x3(c, c->aux_xsg, c->aux_ws1, c->zero_bias, n, c->Dp, ndp, 256.0f, dA, s, c->aux_scales + 4);
    }
    if (rc != SAEV_OK) return rc;
    // the selection's mask applied, max |dA| and dA's operand scale (aux_scales + 10) in one pass
    if (!c->aux_all) HIPCHK(c, launch_mask_apply_absmax(dA, c->A_mask, (long)n * ndp, c->aux_sync, c->aux_scales + 10, s));
    else HIPCHK(c, launch_absmax_pow2(dA, (long)n * ndp, c->aux_sync, c->aux_scales + 10, s));
    {
        // operand scales: A from the forward (aux_scales + 2), g_aux from above (+ 4), x from max|x| (+ 6), dA fresh
        rc = ksplit_f16x3(c, c->A_dead, c->aux_scales + 2, ndp, c->g_aux, c->aux_scales + 4, D, n, c->dWd, s,
                          c->aux_both ? c->aux_kA : nullptr, c->aux_both ? c->aux_kD : nullptr);
        if (rc != SAEV_OK) return rc;
        // (x's scale: the forward formed it when it made its own hi/lo images of x)
        if (c->cfg.encoder_mode == SAEV_ENCODER_F16X3) HIPCHK(c, launch_pow2_scale(c->upper_c, c->aux_scales + 6, s));
        rc = ksplit_f16x3(c, dA, c->aux_scales + 10, ndp, c->x_last, c->aux_scales + 6, D, n, c->dWe, s, nullptr,
                          c->aux_both ? c->aux_kX : nullptr);
        if (rc != SAEV_OK) return rc;
    }
    HIPCHK(c, launch_colsum(dA, n, ndp, c->aux_partials, c->dbe, 0, nullptr, s));
    HIPCHK(c, launch_colsum(c->g_aux, n, D, c->colsum_partials, c->ov_x != nullptr ? c->db_aux : c->grads + c->off_b_dec,
                            c->ov_x != nullptr ? 0 : 1, nullptr, s));
    // the compact rows dWd / dWe / dbe are added into the gradient rows of the dead latents by saev_backward_rows
    return SAEV_OK;
}

}  // namespace

int saev_step_dead(saev_ctx* c, int64_t n_rows_global, void* stream) {
    if (!c) return SAEV_INVALID_ARG;
    REQUIRE(c, c->x_last && c->training_last, SAEV_INVALID_ARG, "saev_step_dead: no training forward in flight");
    hipStream_t s = (hipStream_t)stream;
    const int S = c->cfg.d_sae;
    const int64_t step = ++c->dead_steps;
    c->tokens_seen += n_rows_global;
    DeadArgs d{};
    d.toks = c->toks; d.fired = c->fired; d.dead = c->dead; d.S = S;
    d.add_tokens = n_rows_global; d.threshold = c->cfg.dead_threshold_tokens; d.k_aux = c->cfg.k_aux;
    d.n_dead = c->flags + 4; d.k_use = c->flags + 5; d.stats = c->stats; d.scratch = c->flags + 6;
    // (saev_debug_cfg.dead_lag: how many steps old the record is that sizes this step's auxiliary work -- a shorter lag gives a
    // tighter bound of the dead count, a longer one lets the host run further ahead of the device)
    const int lag = c->dbg.dead_lag > 0 ? std::min(c->dbg.dead_lag, DEAD_RING / 2) : DEAD_LAG;
    d.horizon_tokens = (int64_t)lag * n_rows_global;
    d.step = step; d.cum_tokens = c->tokens_seen;
    d.rec = c->rec_dev ? c->rec_dev + step % DEAD_RING : nullptr;
    c->dead_list_ready = false;
    if (c->stats_pending) {
        c->stats_pending = false;
        d.dead_list = c->cfg.k_aux > 0 ? c->dead_list : nullptr;
        HIPCHK(c, launch_stats_dead(c->rowstats, c->n_last, c->cfg.d_model, c->P_last, c->cfg.alpha, c->upper_c, c->flags + 2, c->stats,
                                    c->stats_scratch, c->stats_lists ? c->cand_cnt : nullptr, CAND_CAP, d, s));
        c->dead_list_ready = d.dead_list != nullptr;
    } else {
        HIPCHK(c, launch_dead_update(d, s));
    }
    c->n_dead_host = 0;
    c->k_use_host = 0;
    c->aux_route = AUX_NONE;
    c->aux_small = false;
    c->aux_dev_count = false;
    if (c->cfg.k_aux <= 0) return SAEV_OK;
    HIPCHK(c, hipEventRecord(c->dead_ev[step % DEAD_RING], s));
    // A latent can only be dead once `threshold` tokens went by since the tracker was last known to be all-zero.
    if (!c->tracker_dirty && c->tokens_seen < c->cfg.dead_threshold_tokens) return SAEV_OK;
    // The reference reads n_dead back every step (modeling.py:92).  Here the record the device wrote DEAD_LAG steps ago
    // bounds it: a latent dead now had at most DEAD_LAG steps' worth of tokens to go then (n_near counts those).  While
    // the bound fits the few-dead-latents kernels -- which covers zero, the usual state of a healthy run -- they are
    // enqueued with the count left on the device and nothing is read back.  The wait below is for an event DEAD_LAG
    // steps in the past; it only ever blocks a host that has run further ahead than that, and never drains the queue.
    // (saev_debug_cfg.aux_small_max: -1 sends every dead set down the dense route, for tests and A/B runs)
    // Default AUX_SMALL_DEFAULT: where the two routes cost the same at configs[1] (tools/experiments/r4_aux_sweep.sh: the
    // few-dead-latents kernels grow with the count, the dense algebra is flat up to 256 dead latents).
    // (with the fp32-MFMA kernels -- d_model % 128 == 0 -- the few-dead-latents route costs +0.24 ms up to 32 and +0.32 ... +0.35 up to 64 dead
    // latents against the dense route's +0.56: it takes everything it can hold, profiles/r05b_aux_mfma_sweep.txt)
    // (round 6: up to AUX_MFMA_MAX = 128 where the context's buffers allow -- aux_mfma_cap -- with one launch per count window
    // [1, 32], [33, 64], [65, 128] up to the bound: the device-side count picks the one that runs)
    const bool mfma_route = aux_mfma_supported(c->cfg.d_model) && c->dbg.aux_small_route == 0;
    const int small_default = mfma_route ? std::max((int)AUX_SMALL_MAX, c->aux_mfma_cap) : (int)AUX_SMALL_DEFAULT;
    const int small_cap = c->dbg.aux_small_max < 0 ? 0 : (c->dbg.aux_small_max == 0 ? small_default : std::min(c->dbg.aux_small_max, (int)AUX_SMALL_MAX));
    const int small_max = std::min(small_cap, c->cfg.k_aux);
    const int64_t s0 = step - lag;
    if (s0 >= c->rec_valid_from) {
        HIPCHK(c, hipEventSynchronize(c->dead_ev[s0 % DEAD_RING]));
        const volatile DeadRecord* r = c->rec_host + s0 % DEAD_RING;
        if (r->step == s0 && c->tokens_seen - r->cum_tokens <= r->horizon_tokens) {
            const int bound = r->n_near;  // >= the dead count of this step
            // nobody was within reach of the threshold then: nothing can be dead now, the auxiliary term is exactly zero
            // and its dozen count-predicated launches (each ~5 us of an empty grid) are not enqueued at all
            if (bound == 0) return SAEV_OK;
            if (bound <= small_max && c->cfg.d_model <= 2048) {
                c->aux_route = AUX_SMALL_DEVICE;
                return auxk_small_forward(c, s, bound);
            }
            // A larger dead set: the dense algebra, sized by the bound, with the count left on the device (round 2 read it
            // back here: one blocking read per step whenever more than a few dozen latents were dead -- configs[2]'s regime)
            if ((bound + 3) / 4 * 4 <= c->nd_cap) {
                c->aux_route = AUX_DENSE;
                c->aux_dev_count = true;
                c->n_dead_host = bound;
                c->k_use_host = std::min(c->cfg.k_aux, bound);
                return auxk_forward(c, s);
            }
        }
    }
    int32_t host[2] = {0, 0};
    HIPCHK(c, hipMemcpyAsync(host, c->flags + 4, 2 * sizeof(int32_t), hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipStreamSynchronize(s));
    c->n_readbacks++;
    c->n_dead_host = host[0];
    c->k_use_host = host[1];
    if (c->n_dead_host <= 0) return SAEV_OK;
    if ((c->n_dead_host + 3) / 4 * 4 > c->nd_cap && !(c->n_dead_host <= small_max && c->cfg.d_model <= 2048)) {
        // More dead latents than the dense buffers were sized for (saev_cfg.aux_dead_cap): grow them here -- the stream is
        // idle after the read-back, the buffers carry nothing from step to step -- to twice the need, capped at d_sae.  An
        // exceptional event (a run whose dictionary collapses); it costs a device-wide allocation, never a wrong result.
        const int s4 = (S + 3) / 4 * 4;
        const int cap = std::min(s4, std::max(2 * c->nd_cap, (2 * c->n_dead_host + 1023) / 1024 * 1024));
        for (void* q : c->aux_allocs) hipFree(q);
        c->aux_allocs.clear();
        c->aux_bytes = 0;
        c->nd_cap = 0;
        int rcg = alloc_aux_buffers(c, cap);
        if (rcg != SAEV_OK) {
            c->err = "AuxK: " + std::to_string(c->n_dead_host) + " dead latents exceed saev_cfg.aux_dead_cap and the buffers could not be grown to " +
                     std::to_string(cap) + " (out of device memory)";
            return rcg;
        }
        std::fprintf(stderr, "[saev_amd] AuxK: %d dead latents exceeded the dead-set buffers; grown to %d inside the step "
                             "(device-synchronising; saev_cfg.aux_dead_cap sizes them up front)\n", c->n_dead_host, cap);
    }
    if (c->n_dead_host <= small_max && c->cfg.d_model <= 2048) {
        c->aux_route = AUX_SMALL_HOST;
        return auxk_small_forward(c, s, c->n_dead_host);
    }
    c->aux_route = AUX_DENSE;
    return auxk_forward(c, s);
}

// ---- backward in three pieces (saev_step_backward = all of them over the full latent range) -------------------------

int saev_backward_begin(saev_ctx* c, void* stream) {
    if (!c) return SAEV_INVALID_ARG;
    REQUIRE(c, c->x_last && c->training_last, SAEV_INVALID_ARG, "saev_backward_begin: no training forward in flight");
    REQUIRE(c, c->grads, SAEV_NOT_BOUND, "gradient buffer not bound");
    hipStream_t s = (hipStream_t)stream;
    const int S = c->cfg.d_sae, D = c->cfg.d_model, K = c->cfg.top_k;
    const bool ov = c->ov_x != nullptr;
    const int n = ov ? c->ov_n : c->n_last;  // rows whose (row, latent) pairs this backward covers
    const int words = ((n + 31) / 32 + 7) / 8 * 8;
    c->row_proj_valid = false;
    CscArgs a{};
    a.idx = ov ? c->ov_idx : c->idx; a.code_stride = K; a.k = K; a.k_dev = nullptr; a.n_rows = n; a.S = S;
    a.bitmap = c->bitmap; a.words = words; a.grp_prefix = c->grp_prefix; a.scan_totals = c->scan_totals;
    a.counts = c->counts; a.starts = c->starts; a.pairs = c->pairs;
    a.chunk_starts = c->chunk_starts; a.part_starts = c->part_starts; a.work_latent = c->work_latent;
    // (a gathered backward -- the rows of all ranks, row-major -- gets its slice-major copies here; Matryoshka ones keep dw_rows)
    const bool ov_slices = ov && c->dws_ok && c->P_last == 1 && n <= c->back_rows;
    c->dws_pairs = ov ? ov_slices : c->dws_rows == n;
    if (ov_slices) {
        c->xS_bwd = c->xS;
        if (!c->followers.empty()) {
            if (c->xS_ov == nullptr) {  // (once per context: a device-wide allocation outside any steady-state step)
                int rca = alloc(c, &c->xS_ov, (size_t)c->back_rows * D);
                if (rca != SAEV_OK) return rca;
            }
            c->xS_bwd = c->xS_ov;
        }
        HIPCHK(c, launch_slice_major_copy(c->ov_g, c->ov_x, n, D, c->gS, c->xS_bwd, s));
        c->dws_rows = 0;  // (the copies no longer describe the forward's own rows)
    }
    c->csc_epoch = c->csc_epoch == 0x7fffffff ? 1 : c->csc_epoch + 1;
    a.epoch = c->dbg.csc_route == 2 ? 0 : c->csc_epoch;  // (csc_route 2: the two-launch scan)
    if (c->dws_pairs) {
        a.zero_word = c->cut_list;
        a.pv = c->pv; a.plat = c->plat; a.val = ov ? c->ov_val : c->val;
        a.P = c->P_last;
        for (int p = 0; p < c->P_last; ++p) a.cuts[p] = c->cuts_last[p];
        if (!ov && c->dval_fwd) { a.pv2 = c->pv2; a.dval = c->dval_rows; }
    }
    c->dval_pairs_ready = c->dws_pairs && a.pv2 != nullptr;
    // (inside saev_train_step the column slices take the whole backward: the row kernels' pair list and work items are not built)
    if (c->fused_step && c->dws_pairs) { a.pairs = nullptr; a.chunk_starts = nullptr; a.part_starts = nullptr; a.work_latent = nullptr; }
    // (the bit map row pitch depends on the batch: a map cleaned for a pitch covers every shorter one, S * words <= before)
    // db_dec = column sums of dL/dx_hat (Matryoshka: of the suffix sums C_0), formed in the grids of the CSC build's first two
    // launches; the AuxK contractions add theirs
    const float* gmat = ov ? c->ov_g : (c->P_last > 1 ? c->G : c->g);
    // (prefilled: this forward's decode has set the bits of exactly these codes: no clear, no fill pass)
    const bool prefilled = !ov && c->bitmap_prefill_words == words && c->bitmap_prefill_rows == n;
    HIPCHK(c, launch_csc_build(a, s, c->bitmap_clean && words <= c->bitmap_clean_words, gmat, D, (long)c->P_last * D,
                               c->colsum_partials, c->grads + c->off_b_dec, prefilled));
    c->bitmap_prefill_words = 0;
    c->last_backward_gathered = ov;
    c->bitmap_clean = false;
    c->bitmap_words_last = words;
    if (c->aux_route != AUX_NONE) {
        int rc = auxk_backward(c, s);
        if (rc != SAEV_OK) return rc;
    }
    return SAEV_OK;
}

int saev_backward_rows(saev_ctx* c, int32_t lat_lo, int32_t lat_hi, void* stream) {
    return saev_backward_rows_part(c, lat_lo, lat_hi, 0, stream);
}

int saev_backward_rows_part(saev_ctx* c, int32_t lat_lo, int32_t lat_hi, int32_t part, void* stream) {
    if (!c) return SAEV_INVALID_ARG;
    REQUIRE(c, part >= 0 && part <= 2, SAEV_INVALID_ARG, "saev_backward_rows_part: part must be 0 (both), 1 (decoder) or 2 (encoder)");
    REQUIRE(c, c->x_last && c->training_last && c->grads, SAEV_INVALID_ARG, "saev_backward_rows: call saev_backward_begin first");
    const int S = c->cfg.d_sae, D = c->cfg.d_model, K = c->cfg.top_k;
    const bool ov = c->ov_x != nullptr;
    const int n = ov ? c->ov_n : c->n_last;
    REQUIRE(c, 0 <= lat_lo && lat_lo < lat_hi && lat_hi <= S, SAEV_INVALID_ARG, "saev_backward_rows: bad latent range");
    hipStream_t s = (hipStream_t)stream;
    DwRowsArgs a{};
    a.starts = c->starts; a.chunk_starts = c->chunk_starts; a.work_latent = c->work_latent;
    a.part_starts = c->part_starts; a.pairs = c->pairs; a.val = ov ? c->ov_val : c->val; a.W_dec = c->params + c->off_W_dec;
    a.g = ov ? c->ov_g : (c->P_last > 1 ? c->G : c->g);  // Matryoshka: rows receive the suffix-summed gradients C_p
    a.x = ov ? c->ov_x : c->x_last;
    a.D = D; a.S = S; a.k_dev = nullptr; a.accumulate = 0;
    a.P = c->P_last;
    for (int p = 0; p < c->P_last; ++p) a.cuts[p] = c->cuts_last[p];
    a.dW_dec = c->grads + c->off_W_dec; a.dW_encT = c->dW_encT; a.db_enc = c->grads + c->off_b_enc;
    a.partials = c->partials; a.db_partials = c->db_partials;
    a.lat_lo = lat_lo; a.lat_hi = lat_hi;
    a.part = part; a.dval = c->dval_pairs;
    const bool all_rows = part == 0 && lat_lo == 0 && lat_hi == S;
    // a pass over all latents also clears the CSC bit map behind itself (nothing reads it after the build)
    const bool clears = part != 2 && lat_lo == 0 && lat_hi == S && c->bitmap_words_last > 0;
    if (clears) { a.clear_bitmap = c->bitmap; a.clear_words = c->bitmap_words_last; }
    a.row_proj = all_rows ? c->row_proj : nullptr; a.project = c->cfg.remove_parallel_grads ? 1 : 0;
    a.enc_sq = all_rows ? c->enc_sq : nullptr;
    // upper bound of the work items of the range (one per latent + one per 64 pairs): the kernel knows the exact count
    const int max_work = (lat_hi - lat_lo) + (int)(((long)n * K + DW_CHUNK - 1) / DW_CHUNK);
    if (lat_lo == 0 && lat_hi == S && c->dws_pairs && (ov || c->dws_rows == n)) {
        // all latents of this context's own batch (in one pass or as the decoder / encoder halves of a two-pass backward): column slices out of the XCD L2s (kernels.h: DwSlicesArgs)
        DwSlicesArgs w{};
        w.starts = c->starts; w.pv = c->pv; w.pv2 = c->pv2; w.plat = c->plat; w.gS = c->P_last > 1 ? c->GS : c->gS; w.W_dec = a.W_dec; w.P = c->P_last;
        // (the forward's own slice-major x: split_f16r's -- possibly the leader's -- or the decode's; gathered rows: the copy saev_backward_begin made)
        w.xS = ov ? c->xS_bwd : (c->fwd_step ? c->xS_c : c->xS);
        w.n_rows = n; w.D = D; w.S = S; w.pair_cap = (int)((long)c->back_rows * K);
        w.dvp = c->dvp; w.dW_dec = a.dW_dec; w.dW_encT = a.dW_encT; w.db_enc = a.db_enc;
        const size_t runs_cap = ((size_t)w.pair_cap + DWS_RUN - 1) / DWS_RUN;
        w.part_dec = c->partials; w.part_enc = c->partials + 2 * runs_cap * D;  // (max_part * 2 rows hold 4 * runs_cap)
        w.cut_lat = c->cut_lat; w.cut_list = c->cut_list;
        w.lat_unused = (all_rows && c->fused_step) ? c->lat_unused : nullptr;
        c->unused_valid = w.lat_unused != nullptr;
        w.row_proj = a.row_proj; w.project = a.project; w.enc_sq = a.enc_sq;
        w.clear_bitmap = a.clear_bitmap; w.clear_words = a.clear_words;
        w.have_dval = c->dval_pairs_ready ? 1 : 0;
        c->sq_wave_n = 0;
        if (c->dval_pairs_ready && c->wn2_fresh && part == 0) {
            w.wn2 = c->wn2;
            if (c->fused_step && all_rows && c->sq_wave != nullptr && c->dbg.fin_route != 2) {  // (fin_route 2: the finalize reads the rows for their squares)
                c->sq_wave_n = dw_slices_waves(D, (int)((long)n * K));
                w.sq_wave_dec = c->sq_wave; w.sq_wave_enc = c->sq_wave + c->sq_wave_n;
            }
        }
        HIPCHK(c, launch_dw_slices(w, (int)((long)n * K), part, s));
    } else {
        c->unused_valid = false;
        HIPCHK(c, launch_dw_rows(a, max_work, s));
    }
    if (c->aux_route == AUX_DENSE)  // (the count on the device when the host only had a bound of it: aux_dev_count)
        HIPCHK(c, launch_scatter_add_dead(c->dead_list, c->n_dead_host, D, c->dWd, c->dWe, c->dbe, c->grads + c->off_W_dec,
                                          c->dW_encT, c->grads + c->off_b_enc, lat_lo, lat_hi, s,
                                          c->aux_dev_count ? c->flags + 4 : nullptr, part, a.row_proj, a.W_dec, a.project, a.enc_sq,
                                          c->unused_valid ? c->lat_unused : nullptr, c->sq_wave_n > 0 ? c->starts : nullptr));
    else if (c->aux_route != AUX_NONE)  // few dead latents: the device knows how many
        HIPCHK(c, launch_scatter_add_dead(c->dead_list, c->aux_mfma ? c->aux_ndp : AUX_SMALL_MAX, D, c->dWd, c->dWe, c->dbe, c->grads + c->off_W_dec,
                                          c->dW_encT, c->grads + c->off_b_enc, lat_lo, lat_hi, s, c->flags + 4, part,
                                          a.row_proj, a.W_dec, a.project, a.enc_sq, c->unused_valid ? c->lat_unused : nullptr,
                                          c->sq_wave_n > 0 ? c->starts : nullptr));
    // gathered backward: the auxiliary term's share of db_dec (summed over the ranks by the caller, like the compact rows)
    if (ov && c->aux_route != AUX_NONE && part != 2 && lat_lo == 0)
        HIPCHK(c, launch_colsum(c->db_aux, 1, D, c->colsum_partials, c->grads + c->off_b_dec, 1, nullptr, s));
    c->row_proj_valid = all_rows;
    if (clears) { c->bitmap_clean = true; c->bitmap_clean_words = c->bitmap_words_last; }
    return SAEV_OK;
}

float* saev_grad_w_enc_t(saev_ctx* c) { return c ? c->dW_encT : nullptr; }

int saev_bind_w_enc_t(saev_ctx* c, float* scratch) {
    if (!c || !scratch) return SAEV_INVALID_ARG;
    c->dW_encT = scratch;
    return SAEV_OK;
}

int saev_copy_step_state(saev_ctx* c, int32_t n_rows, float* g_out, int32_t* idx_out, float* val_out, void* stream) {
    if (!c) return SAEV_INVALID_ARG;
    REQUIRE(c, c->n_last > 0 && c->training_last && n_rows == c->n_last, SAEV_INVALID_ARG,
            "saev_copy_step_state: n_rows must be the row count of the training forward in flight");
    hipStream_t s = (hipStream_t)stream;
    // (Matryoshka: P suffix-summed gradients per row, (n_rows, P, d_model) -- what the backward consumes in that case)
    const size_t nk = (size_t)n_rows * c->cfg.top_k, nd = (size_t)n_rows * c->cfg.d_model * (size_t)c->P_last;
    if (g_out) HIPCHK(c, hipMemcpyAsync(g_out, c->P_last > 1 ? c->G : c->g, nd * sizeof(float), hipMemcpyDeviceToDevice, s));
    if (idx_out) HIPCHK(c, hipMemcpyAsync(idx_out, c->idx, nk * sizeof(int32_t), hipMemcpyDeviceToDevice, s));
    if (val_out) HIPCHK(c, hipMemcpyAsync(val_out, c->val, nk * sizeof(float), hipMemcpyDeviceToDevice, s));
    return SAEV_OK;
}

int saev_backward_override(saev_ctx* c, const float* x_all, const float* g_all, const int32_t* idx_all, const float* val_all,
                           int32_t n_all) {
    if (!c) return SAEV_INVALID_ARG;
    if (x_all == nullptr) { c->ov_x = nullptr; c->ov_n = 0; return SAEV_OK; }
    REQUIRE(c, g_all && idx_all && val_all && n_all > 0, SAEV_INVALID_ARG, "saev_backward_override: NULL buffer");
    REQUIRE(c, n_all <= c->back_rows, SAEV_INVALID_ARG,
            "saev_backward_override: the gathered row count exceeds saev_cfg.max_backward_rows (set it to the GLOBAL batch)");
    REQUIRE(c, c->x_last && c->training_last, SAEV_INVALID_ARG, "saev_backward_override: no training forward in flight");
    REQUIRE(c, ((uintptr_t)x_all % 16) == 0 && ((uintptr_t)g_all % 16) == 0, SAEV_INVALID_ARG, "x_all / g_all must be 16-byte aligned");
    c->ov_x = x_all; c->ov_g = g_all; c->ov_idx = idx_all; c->ov_val = val_all; c->ov_n = n_all;
    return SAEV_OK;
}

int32_t saev_aux_compact_rows(const saev_ctx* c) {
    if (!c || c->aux_route == AUX_NONE) return 0;
    return c->aux_route == AUX_DENSE ? (c->n_dead_host + 3) / 4 * 4 : (c->aux_mfma ? c->aux_ndp : AUX_SMALL_MAX);
}

// [dWd rows x D | dWe rows x D | dbe rows | db_aux D]
static int aux_compact_copy(saev_ctx* c, float* buf, bool out, hipStream_t s) {
    const size_t rows = (size_t)saev_aux_compact_rows(c), D = c->cfg.d_model;
    if (rows == 0) return SAEV_OK;
    REQUIRE(c, buf != nullptr, SAEV_INVALID_ARG, "saev_aux_compact_*: NULL buffer");
    float* seg[4] = {c->dWd, c->dWe, c->dbe, c->db_aux};
    const size_t len[4] = {rows * D, rows * D, rows, D};
    size_t off = 0;
    for (int i = 0; i < 4; ++i) {
        if (out) HIPCHK(c, hipMemcpyAsync(buf + off, seg[i], len[i] * sizeof(float), hipMemcpyDeviceToDevice, s));
        else HIPCHK(c, hipMemcpyAsync(seg[i], buf + off, len[i] * sizeof(float), hipMemcpyDeviceToDevice, s));
        off += len[i];
    }
    return SAEV_OK;
}
int saev_aux_compact_export(saev_ctx* c, float* buf, void* stream) {
    if (!c) return SAEV_INVALID_ARG;
    return aux_compact_copy(c, buf, true, (hipStream_t)stream);
}
int saev_aux_compact_import(saev_ctx* c, const float* buf, void* stream) {
    if (!c) return SAEV_INVALID_ARG;
    return aux_compact_copy(c, const_cast<float*>(buf), false, (hipStream_t)stream);
}

int saev_trust_gradients(saev_ctx* c, int32_t on) {
    if (!c) return SAEV_INVALID_ARG;
    c->trust_grads = on != 0;
    return SAEV_OK;
}

int saev_backward_end(saev_ctx* c, void* stream) {
    if (!c) return SAEV_INVALID_ARG;
    REQUIRE(c, c->grads, SAEV_NOT_BOUND, "gradient buffer not bound");
    // (the per-tile squares land behind the tail's other partial sums: [2 nb + ceil(S / 4), ...))
    double* sq = c->sumsq_partials + 2 * sumsq_blocks() + (c->cfg.d_sae + 3) / 4;
    HIPCHK(c, launch_transpose(c->dW_encT, c->grads + c->off_W_enc, c->cfg.d_sae, c->cfg.d_model, (hipStream_t)stream, sq));
    c->wenc_sq_valid = true;
    return SAEV_OK;
}

int saev_step_backward(saev_ctx* c, void* stream) {
    int rc = saev_backward_begin(c, stream);
    if (rc != SAEV_OK) return rc;
    rc = saev_backward_rows(c, 0, c->cfg.d_sae, stream);
    if (rc != SAEV_OK) return rc;
    return saev_backward_end(c, stream);
}

// The element ranges a tail call works on: everything (shard_rank < 0), or rank `shard_rank`'s chunk of each half.
namespace {
struct TailRanges { long a_lo, a_hi, b_lo, b_hi; };
int tail_ranges(saev_ctx* c, int shard_rank, TailRanges* r) {
    if (shard_rank < 0) {
        *r = {0, c->off_W_enc, c->off_W_enc, c->n_params};
        return SAEV_OK;
    }
    REQUIRE(c, shard_rank < c->shard_world, SAEV_INVALID_ARG, "shard_rank >= saev_cfg.shard_world");
    r->a_lo = (long)shard_rank * c->chunk_a; r->a_hi = r->a_lo + c->chunk_a;
    r->b_lo = c->off_W_enc + (long)shard_rank * c->chunk_b; r->b_hi = r->b_lo + c->chunk_b;
    return SAEV_OK;
}
}  // namespace

double* saev_sumsq_device(saev_ctx* c) { return c ? (c->sumsq_bound ? c->sumsq_bound : c->sumsq_total) : nullptr; }

int saev_bind_sumsq(saev_ctx* c, double* sumsq) {
    if (!c) return SAEV_INVALID_ARG;
    c->sumsq_bound = sumsq;
    return SAEV_OK;
}

int saev_wenc_ready_event(saev_ctx* c, void* event) {
    if (!c) return SAEV_INVALID_ARG;
    c->wenc_ready = (hipEvent_t)event;
    return SAEV_OK;
}

int saev_wdec_ready_event(saev_ctx* c, void* event) {
    if (!c) return SAEV_INVALID_ARG;
    c->wdec_ready = (hipEvent_t)event;
    return SAEV_OK;
}

int saev_tail_prepare(saev_ctx* c, int32_t shard_rank, void* stream) {
    if (!c) return SAEV_INVALID_ARG;
    REQUIRE(c, c->params && c->grads, SAEV_NOT_BOUND, "saev_tail_prepare: params/grads not bound");
    TailRanges r;
    int rc = tail_ranges(c, shard_rank, &r);
    if (rc != SAEV_OK) return rc;
    hipStream_t s = (hipStream_t)stream;
    const long S = c->cfg.d_sae, D = c->cfg.d_model;
    c->tail_proj_in_adam = false;
    if (c->wenc_t_pending) {
        // saev_train_step left the W_enc gradient in the transposed scratch: its squares come from the rows' statistics too
        // (enc_sq), and the one Adam launch reads it from there (adam_fused_kernel) -- no transpose pass at all
        REQUIRE(c, c->row_proj_valid && shard_rank < 0, SAEV_INVALID_ARG, "saev_tail_prepare: pending transposed gradient without a full backward");
        c->row_proj_valid = false;
        HIPCHK(c, launch_sumsq_final_ex(nullptr, 0, c->row_proj, (int)S, c->grads + S * D, r.a_hi - S * D,
                                        c->grads + c->off_b_enc, r.b_hi - c->off_b_enc, saev_sumsq_device(c), c->sumsq_partials,
                                        c->tickets + 1, s, c->enc_sq, c->sq_wave_n > 0 ? c->sq_wave : nullptr, 2l * c->sq_wave_n));
        c->sq_wave_n = 0;
        c->tail_proj_in_adam = true;
        return SAEV_OK;
    }
    if (c->trust_grads && c->wenc_sq_valid && c->row_proj_valid && shard_rank < 0) {
        // The caller vouches that nothing has touched the gradient since the backward: the kernels that wrote the decoder
        // rows left each row's projection coefficient and projected squares (row_proj), the transpose the squares of dW_enc
        // tile by tile.  One small reduction gives the clip norm, and Adam applies the projection to the rows as it reads
        // them: the gradient is streamed once by the whole tail instead of three times (rpg read + write, Adam read).
        c->wenc_sq_valid = false; c->row_proj_valid = false;
        const double* tsq = c->sumsq_partials + 2 * sumsq_blocks() + (S + 3) / 4;
        HIPCHK(c, launch_sumsq_final_ex(tsq, transpose_blocks((int)S, (int)D), c->row_proj, (int)S, c->grads + S * D, r.a_hi - S * D,
                                        c->grads + c->off_b_enc, r.b_hi - c->off_b_enc, saev_sumsq_device(c), c->sumsq_partials,
                                        c->tickets + 1, s));
        c->tail_proj_in_adam = true;
        return SAEV_OK;
    }
    c->row_proj_valid = false;
    // decoder rows of the range: projection (modeling.py:419-445) and their squares in one pass over the gradient
    const long row_lo = std::min(r.a_lo / D, S), row_hi = std::min(r.a_hi / D, S);
    const int n_rows = (int)(row_hi - row_lo);
    const int nb = sumsq_blocks();
    double* part = c->sumsq_partials;  // [0, nb): rest of the first half; [nb, 2 nb): second half; then one per 4 rows
    HIPCHK(c, launch_rpg(c->grads + row_lo * D, c->params + row_lo * D, n_rows, (int)D, s, part + 2 * nb,
                         c->cfg.remove_parallel_grads ? 1 : 0));
    const long rest_lo = std::max(r.a_lo, S * D);
    HIPCHK(c, launch_sumsq_partials(c->grads + rest_lo, std::max(0L, r.a_hi - rest_lo), part, s));
    c->wenc_sq_valid = false;
    HIPCHK(c, launch_sumsq_partials(c->grads + r.b_lo, r.b_hi - r.b_lo, part + nb, s));
    HIPCHK(c, launch_sumsq_final(part, 2 * nb + (n_rows + 3) / 4, saev_sumsq_device(c), s));
    return SAEV_OK;
}

int saev_tail_apply(saev_ctx* c, float lr, float max_norm, float grad_scale, int64_t adam_step, int32_t shard_rank,
                    void* stream) {
    if (!c) return SAEV_INVALID_ARG;
    REQUIRE(c, c->params && c->grads && c->adam_m && c->adam_v, SAEV_NOT_BOUND,
            "saev_tail_apply: params/grads/adam state not bound");
    REQUIRE(c, adam_step >= 1, SAEV_INVALID_ARG, "adam_step is 1-based");
    TailRanges r;
    int rc = tail_ranges(c, shard_rank, &r);
    if (rc != SAEV_OK) return rc;
    hipStream_t s = (hipStream_t)stream;
    c->wn2_fresh = false;  // (W_dec moves)
    // (a follower's images are centred on its lender's NEXT mu -- there once the lender's step, which ran first, was a streamed
    // saev_train_step: fwd_moves_mu -- and carry the lender's serial of it)
    const bool emit_follow = c->leader != nullptr && c->borrow_streamed && c->leader->fwd_moves_mu && c->dbg.group_route == 0 &&
                             c->cfg.encoder_mode == SAEV_ENCODER_F16R && c->fwd_step;
    const bool emit = c->train_fused && c->stream_ok && (c->leader == nullptr ? c->prep_valid && (c->followers.empty() || c->dbg.group_route == 0) : emit_follow) &&
                      shard_rank < 0 && c->tail_proj_in_adam && c->wenc_t_pending;
    c->wimg_fresh = false;  // (W_enc moves: only the fused Adam below leaves images of what it writes)
    const bool chk_was_valid = c->wchk_valid;
    c->wchk_valid = false;
    const bool emit_bf16 = c->train_fused && c->cfg.encoder_mode == SAEV_ENCODER_BF16 && c->wimg_bf16_fresh && shard_rank < 0 &&
                           c->tail_proj_in_adam && c->wenc_t_pending;
    c->wimg_bf16_fresh = false;
    AdamArgs a{};
    a.lr = lr; a.beta1 = 0.9f; a.beta2 = 0.999f; a.eps = 1e-8f;
    a.omb1 = (float)(1.0 - 0.9); a.omb2 = (float)(1.0 - 0.999);
    a.bc1 = (float)(1.0 - std::pow(0.9, (double)adam_step));
    a.bc2_sqrt = (float)std::sqrt(1.0 - std::pow(0.999, (double)adam_step));
    a.grad_scale = grad_scale; a.max_norm = max_norm; a.sumsq = saev_sumsq_device(c); a.stats = c->stats;
    const long lo[2] = {r.a_lo, r.b_lo}, hi[2] = {r.a_hi, r.b_hi};
    if (shard_rank < 0 && c->tail_proj_in_adam && c->wenc_t_pending) {  // everything in one launch (adam_fused_kernel)
        c->tail_proj_in_adam = false; c->wenc_t_pending = false;
        const long S = c->cfg.d_sae, D = c->cfg.d_model;
        a.p = c->params; a.g = c->grads; a.m = c->adam_m; a.v = c->adam_v; a.n = c->n_params;
        AdamImageArgs im{};
        if (emit) {
            // this step's images were built (or found) with scl(c); a step that took the full preparation hands its x scale and
            // normaliser on to the next one (a streamed step's second launch has written them already)
            if (!c->stream_step) HIPCHK(c, hipMemcpyAsync(scl_next(c), scl(c), 8 * sizeof(float), hipMemcpyDeviceToDevice, s));
            im.ws = c->ws; im.WeS = c->WeS; im.dot_part = reinterpret_cast<double*>(c->dot_part); im.sq_part = c->sq_part;
            im.mu = c->leader != nullptr ? c->leader->mu : c->mu; im.wmax_prev = c->wmax_prev; im.scales_next = scl_next(c); im.nks = c->Dp / 32; im.S_pad = c->S_pad;
        }
        if (emit_bf16) { im.ws = c->ws; im.nks = c->Dp / 32; im.S_pad = c->S_pad; im.mode = 1; }
        if ((emit || emit_bf16) && c->wchk != nullptr && c->dbg.own_check == 0) {
            // the tiles' checksums: left for the next step, and -- when this step's forward ran on images an earlier Adam left --
            // compared with what that Adam left (an evaluation forward in between changes nothing: W_enc did not move)
            im.chk = c->wchk; im.late = c->stale_dev != nullptr ? c->stale_dev + 1 : nullptr;
            im.verify = (chk_was_valid && c->fwd_reused_wimg && im.late != nullptr) ? 1 : 0;
            im.early = (emit && c->leader == nullptr) ? c->flags + 13 : nullptr;  // (a follower's first kernels do not look at W_enc)
        }
        HIPCHK(c, launch_adam_fused(a, c->row_proj, c->dW_encT, (int)S, (int)D, S * D, c->off_W_enc - S * D, c->off_W_enc,
                                    c->off_b_enc, c->n_params - c->off_b_enc, s, c->unused_valid ? c->lat_unused : nullptr,
                                    (emit || emit_bf16) ? &im : nullptr));
        c->unused_valid = false;
        c->wchk_valid = (emit || emit_bf16) && c->wchk != nullptr && c->dbg.own_check == 0;
        c->wimg_bf16_fresh = emit_bf16;
        if (emit) {
            // the bias of the next centred first pass and the column-norm maxima its margins need: W-only, so they are finished here
            HIPCHK(c, launch_bias_finish(reinterpret_cast<const double*>(c->dot_part), c->sq_part, c->Dp, (int)S, c->S_pad, scl_next(c) + 1,
                                         c->params + c->off_b_enc, c->b_shift, c->wnorm_scratch, s, c->b_seen));
            c->scale_par ^= 1;
            c->wimg_fresh = true;
            c->wimg_mu_serial = c->leader != nullptr ? c->leader->mu_serial : c->mu_serial;
        }
        return SAEV_OK;
    }
    if (shard_rank < 0 && c->tail_proj_in_adam) {  // decoder rows with the projection applied on the way in, then the rest
        c->tail_proj_in_adam = false;
        const long S = c->cfg.d_sae, D = c->cfg.d_model;
        a.p = c->params; a.g = c->grads; a.m = c->adam_m; a.v = c->adam_v; a.n = S * D;
        HIPCHK(c, launch_adam_rows(a, c->row_proj, (int)S, (int)D, s));
        a.p += S * D; a.g += S * D; a.m += S * D; a.v += S * D; a.n = c->n_params - S * D;
        HIPCHK(c, launch_adam(a, s));
        return SAEV_OK;
    }
    c->tail_proj_in_adam = false;
    if (shard_rank < 0) {  // one contiguous stream over everything
        a.p = c->params; a.g = c->grads; a.m = c->adam_m; a.v = c->adam_v; a.n = c->n_params;
        HIPCHK(c, launch_adam(a, s));
        return SAEV_OK;
    }
    for (int h = 0; h < 2; ++h) {
        a.p = c->params + lo[h]; a.g = c->grads + lo[h]; a.m = c->adam_m + lo[h]; a.v = c->adam_v + lo[h]; a.n = hi[h] - lo[h];
        HIPCHK(c, launch_adam(a, s));
    }
    return SAEV_OK;
}

int saev_step_tail(saev_ctx* c, float lr, float max_norm, float grad_scale, int64_t adam_step, void* stream) {
    int rc = saev_tail_prepare(c, -1, stream);
    if (rc != SAEV_OK) return rc;
    return saev_tail_apply(c, lr, max_norm, grad_scale, adam_step, -1, stream);
}

int saev_train_step_gather(saev_ctx* c, const float* pool, const int64_t* rows, float* x_out, int32_t n, float lr, float max_norm,
                           int64_t adam_step, void* stream) {
    if (!c) return SAEV_INVALID_ARG;
    REQUIRE(c, pool && rows && x_out, SAEV_INVALID_ARG, "saev_train_step_gather: NULL buffer");
    c->gather_pool = pool; c->gather_rows = rows;
    const int rc = saev_train_step(c, x_out, n, lr, max_norm, adam_step, stream);
    c->gather_pool = nullptr; c->gather_rows = nullptr;
    return rc;
}

int saev_params_touched(saev_ctx* c) {
    if (!c) return SAEV_INVALID_ARG;
    c->wchk_valid = false;
    c->wimg_fresh = false;
    c->wimg_bf16_fresh = false;
    c->wn2_fresh = false;
    return SAEV_OK;
}

int saev_train_step(saev_ctx* c, const float* x, int32_t n, float lr, float max_norm, int64_t adam_step,
                    void* stream) {
    if (!c) return SAEV_INVALID_ARG;
    c->fused_forward = c->dws_ok && c->GS != nullptr;  // (the backward below takes the column slices: nothing reads G's blocks 1..P-1)
    c->train_fused = true;
    int rc = saev_step_forward(c, x, n, n, 1, stream);
    c->fused_forward = false;
    if (rc != SAEV_OK) { c->train_fused = false; return rc; }
    rc = saev_step_dead(c, n, stream);
    if (rc != SAEV_OK) { c->train_fused = false; return rc; }
    // (no saev_backward_end: the W_enc gradient stays in the transposed scratch the backward writes; the tail's single Adam
    // launch reads it there through LDS tiles.  The W_enc segment of the gradient buffer is NOT updated by this entry point
    // -- callers that want to look at gradients use the phases)
    c->fused_step = true;
    rc = saev_backward_begin(c, stream);
    if (rc == SAEV_OK) rc = saev_backward_rows(c, 0, c->cfg.d_sae, stream);
    c->fused_step = false;
    if (rc != SAEV_OK) { c->train_fused = false; return rc; }
    c->wenc_t_pending = true;
    rc = saev_step_tail(c, lr, max_norm, 1.0f, adam_step, stream);
    c->wenc_t_pending = false;
    c->train_fused = false;
    return rc;
}

// ---- data parallel behind the ABI: RCCL taken from the process at run time (include/saev_amd.h: DATA PARALLEL) ----------
namespace {
// (the few declarations of rccl.h this file needs -- the header is not included so that nothing here can end up as a link-time
// dependency: ncclResult_t 0 = success; ncclDataType_t ncclInt32 = 2, ncclFloat32 = 7; ncclRedOp_t ncclSum = 0, ncclMax = 2)
struct RcclApi {
    void* lib = nullptr;
    int (*GetUniqueId)(void*) = nullptr;
    int (*CommInitRank)(void**, int, const void*, int) = nullptr;  // (ncclUniqueId is passed by value: see comm_init_rank)
    int (*CommDestroy)(void*) = nullptr;
    int (*CommAbort)(void*) = nullptr;
    int (*AllReduce)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
};
RcclApi g_rccl;
std::once_flag g_rccl_once;
struct UniqueId128 { char bytes[128]; };  // == ncclUniqueId (NCCL_UNIQUE_ID_BYTES 128)
bool rccl_load_once() {
    void* h = nullptr;
    for (const char* name : {"librccl.so.1", "librccl.so"}) {  // the copy the process already holds, if any ...
        h = dlopen(name, RTLD_NOW | RTLD_NOLOAD);
        if (h) break;
    }
    if (!h)
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {  // ... else the system's
            h = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (h) break;
        }
    if (!h) return false;
    g_rccl.GetUniqueId = reinterpret_cast<int (*)(void*)>(dlsym(h, "ncclGetUniqueId"));
    g_rccl.CommInitRank = reinterpret_cast<int (*)(void**, int, const void*, int)>(dlsym(h, "ncclCommInitRank"));
    g_rccl.CommDestroy = reinterpret_cast<int (*)(void*)>(dlsym(h, "ncclCommDestroy"));
    g_rccl.AllReduce = reinterpret_cast<int (*)(const void*, void*, size_t, int, int, void*, hipStream_t)>(dlsym(h, "ncclAllReduce"));
    g_rccl.CommAbort = reinterpret_cast<int (*)(void*)>(dlsym(h, "ncclCommAbort"));
    g_rccl.GetErrorString = reinterpret_cast<const char* (*)(int)>(dlsym(h, "ncclGetErrorString"));
    if (!g_rccl.GetUniqueId || !g_rccl.CommInitRank || !g_rccl.CommDestroy || !g_rccl.AllReduce) return false;
    g_rccl.lib = h;
    return true;
}
// (contexts of several host threads may initialise their communicators at the same time: the lookup runs once)
bool rccl_load() {
    std::call_once(g_rccl_once, [] { rccl_load_once(); });
    return g_rccl.lib != nullptr;
}
// A rank that fails between the step's two collectives leaves its peers inside a collective it will never join: the
// communicator is aborted (ncclCommAbort: the peers' pending calls return with an error instead of blocking) and dropped; the
// error names what failed.  The Python stepper has a watchdog for the same situation (framework/ddp.py: CollectiveWatchdog).
int dp_fail(saev_ctx* c, int rc) {
    if (c->comm != nullptr) {
        if (g_rccl.CommAbort) g_rccl.CommAbort(c->comm);
        c->comm = nullptr; c->comm_world = 0; c->comm_rank = 0;
        c->err += " [data-parallel step abandoned: communicator aborted, saev_comm_init again to continue]";
    }
    return rc;
}
int rccl_fail(saev_ctx* c, const char* what, int r) {
    c->err = std::string(what) + ": " + (g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "RCCL error") + " (" + std::to_string(r) + ")";
    return SAEV_RCCL_ERROR;
}
}  // namespace

int saev_comm_unique_id(void* id128) {
    if (!id128) return SAEV_INVALID_ARG;
    if (!rccl_load()) return SAEV_UNSUPPORTED;
    return g_rccl.GetUniqueId(id128) == 0 ? SAEV_OK : SAEV_RCCL_ERROR;
}

int saev_comm_init(saev_ctx* c, const void* id128, int32_t rank, int32_t world) {
    if (!c || !id128) return SAEV_INVALID_ARG;
    REQUIRE(c, world >= 1 && rank >= 0 && rank < world, SAEV_INVALID_ARG, "saev_comm_init: rank / world out of range");
    REQUIRE(c, c->comm == nullptr, SAEV_INVALID_ARG, "saev_comm_init: this context already has a communicator (saev_comm_destroy first)");
    REQUIRE(c, rccl_load(), SAEV_UNSUPPORTED, "saev_comm_init: no RCCL in this process and none found (librccl.so.1)");
    HIPCHK(c, hipSetDevice(c->device));
    // ncclCommInitRank(ncclComm_t*, int nranks, ncclUniqueId commId /* by value: a 128-byte struct */, int rank)
    UniqueId128 id;
    std::memcpy(id.bytes, id128, sizeof(id.bytes));
    auto init = reinterpret_cast<int (*)(void**, int, UniqueId128, int)>(reinterpret_cast<void*>(g_rccl.CommInitRank));
    void* comm = nullptr;
    const int r = init(&comm, world, id, rank);
    if (r != 0) return rccl_fail(c, "ncclCommInitRank", r);
    c->comm = comm; c->comm_rank = rank; c->comm_world = world;
    return SAEV_OK;
}

int saev_comm_world(const saev_ctx* c) { return c && c->comm ? c->comm_world : 0; }

int saev_comm_destroy(saev_ctx* c) {
    if (!c) return SAEV_INVALID_ARG;
    if (c->comm != nullptr && g_rccl.CommDestroy) g_rccl.CommDestroy(c->comm);
    c->comm = nullptr; c->comm_world = 0; c->comm_rank = 0;
    return SAEV_OK;
}

int saev_train_step_dp(saev_ctx* c, const float* x_local, int32_t n_local, float lr, float max_norm, int64_t adam_step, void* stream) {
    if (!c) return SAEV_INVALID_ARG;
    REQUIRE(c, c->comm != nullptr, SAEV_INVALID_ARG, "saev_train_step_dp: no communicator (saev_comm_init)");
    REQUIRE(c, c->grads != nullptr, SAEV_NOT_BOUND, "saev_train_step_dp: no gradient buffer bound");
    hipStream_t s = (hipStream_t)stream;
    const int64_t n_global = (int64_t)n_local * c->comm_world;
    const float inv_world = 1.0f / (float)c->comm_world;
    int rc = saev_step_forward(c, x_local, n_local, n_global, 1, stream);
    if (rc != SAEV_OK) return dp_fail(c, rc);
    int r = g_rccl.AllReduce(c->fired, c->fired, (size_t)c->cfg.d_sae, /*ncclInt32*/ 2, /*ncclMax*/ 2, c->comm, s);
    if (r != 0) return dp_fail(c, rccl_fail(c, "ncclAllReduce(fired flags)", r));
    rc = saev_step_dead(c, n_global, stream);
    if (rc != SAEV_OK) return dp_fail(c, rc);
    rc = saev_step_backward(c, stream);
    if (rc != SAEV_OK) return dp_fail(c, rc);
    r = g_rccl.AllReduce(c->grads, c->grads, (size_t)c->n_params, /*ncclFloat32*/ 7, /*ncclSum*/ 0, c->comm, s);
    if (r != 0) return dp_fail(c, rccl_fail(c, "ncclAllReduce(flat gradient)", r));
    rc = saev_step_tail(c, lr, max_norm, inv_world, adam_step, stream);
    return rc == SAEV_OK ? rc : dp_fail(c, rc);
}

}  // extern "C"
