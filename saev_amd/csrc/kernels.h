// Internal launch interface between ctx.hip (the C ABI) and the kernel translation units.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/saev_amd.h"

enum { EPI_DENSE = 0, EPI_TOPK = 1 };

struct EncodeArgs {
    const float* x;       // (n_rows, D)
    const float* W_enc;   // (D, S)
    const float* b_enc;   // (S)
    int n_rows, D, S;
    int s_splits;         // latent ranges per batch block (grid = n_bblocks * s_splits)
    // EPI_DENSE
    float* h_out;         // (n_rows, S)
    // EPI_TOPK
    int ngroups;          // 32 or 64, >= top_k
    int32_t* gmax;        // (ngroups, gmax_stride) ordered-int keys, init INT32_MIN: per-row group maxima shared by
                          // all latent ranges of the row
    int gmax_stride;
    int32_t* cand_cnt;    // (n_rows) init 0
    float* cand_val;      // (n_rows, cand_stride)
    int32_t* cand_idx;    // (n_rows, cand_stride)
    int cand_cap;         // capacity of a row's list
    int cand_stride;      // entries between rows (>= cand_cap; not a power of two, see ctx.hip CAND_STRIDE)
    // optional device-side predicate: run only when (*enable_flag != 0) == enable_when
    const int32_t* enable_flag;
    int enable_when;
};

size_t encode_gemm_smem_bytes();
hipError_t launch_encode_gemm(const EncodeArgs& a, int epi, hipStream_t stream);
int encode_gemm_tile_rows();
int encode_gemm_tile_latents();

struct SelectDenseArgs {
    const float* h;        // (n_rows, S)
    int n_rows, S;
    int k;                 // host-side k (upper bound when k_dev != NULL)
    const int32_t* k_dev;  // optional device-side k (AuxK: min(k_aux, n_dead)); <= 0 -> kernel exits
    const int32_t* mask;   // optional (S) int32; only latents with mask != 0 are eligible
    int32_t* idx_out;      // (n_rows, out_stride)
    float* val_out;
    int out_stride;
    const int32_t* enable_flag;
    int enable_when;
};
hipError_t launch_select_dense(const SelectDenseArgs& a, hipStream_t stream);

struct SelectCandArgs {
    const int32_t* cand_cnt;
    const float* cand_val;
    const int32_t* cand_idx;
    int cand_cap, cand_stride, n_rows, k;
    int32_t* idx_out;
    float* val_out;
    int out_stride;
    const int32_t* enable_flag;
    int enable_when;
    // exact refinement (SAEV_ENCODER_F16R): candidate values are approximate (|error| <= row_margin / 2); the entries
    // within row_margin of the k-th largest are recomputed in fp32 from x and W_enc^T before the final cut
    const float* row_margin;  // (n_rows) or NULL = values are exact already
    const float* x;           // (n_rows, D)
    const float* W_encT;      // (S, D)
    const float* b_enc;       // (S)
    int D;
    int32_t* surv_idx;        // (n_rows, REFINE_CAP) survivors of the approximate cut
    float* surv_val;          // (n_rows, REFINE_CAP) their exact pre-activations (refine_exact_kernel)
    int32_t* surv_cnt;        // (n_rows)
    int32_t* refine_overflow; // set to 1 when a row has more than REFINE_CAP survivors (caller re-runs it densely)
    // predicted bounds (EncodeF16Args::heur_z): the lists hold everything >= the bounds used; they contain the row's true
    // top-k iff the k-th largest entry found is >= the largest bound used for the row.  Rows that fail raise *invalid.
    const int32_t* tau_max;   // (n_rows) ordered-int keys or NULL (guaranteed bounds: nothing to verify)
    int32_t* invalid;         // device flag
    // optional, first select after the encoder: ovf[0] |= any list longer than cand_cap (the step's dense-route flag;
    // the list statistics of the step come from stats_reduce)
    int32_t* ovf;
    // survivor emission grouped by latent range (refine_slices_kernel): surv_rng[row][r] = end of range r's sub-list, ranges of
    // lat_range latents, n_ranges <= RS_MAX_RANGES of them; NULL = one list in candidate order
    int32_t* surv_rng;
    int lat_range, n_ranges;
    // optional (the final cut behind refine_slices_kernel): cand_val is not read; the value of candidate j of a row is
    // sum_bias[cand_idx[j]] + the sum_n shares sum_part[c][row][j] (c = 0 .. sum_n - 1, plane pitch sum_plane floats, row pitch
    // cand_stride) added in slice order -- refine_sum_kernel's pass folded into the select that consumes it
    const float* sum_part;
    const float* sum_bias;
    int sum_n;
    size_t sum_plane;
};
hipError_t launch_refine_exact(const SelectCandArgs& a, hipStream_t stream);
// survivors -> exact values -> final cut in one launch (f16r with guaranteed bounds); a.row_margin, x, W_encT, b_enc set
hipError_t launch_select_refine(const SelectCandArgs& a, hipStream_t stream);
constexpr int REFINE_CAP = 512;
// The exact refinement from 32-column slices of W_enc^T (select.hip: refine_slices_kernel).  xS: slice-major x
// ([D / 32][n_rows][32]); WeS: slice-major W_enc^T ([D / 32][S][32]); part: [D / 32][n_rows][REFINE_CAP] shares of the dot
// products.  lat_range latents (x 128 bytes = the tile an XCD's L2 has to hold) per pass over a slice.
constexpr int RS_SLICE = 32;
constexpr int RS_ROWS_MAX = 8;    // activation rows an eight-lane group works through (4 for batches below 8 192 rows)
constexpr int RS_LAT_RANGE = 32768;
constexpr int RS_MAX_RANGES = 8;  // (more latents than 8 x 16 384: wider ranges)
struct RefineSlicesArgs {
    const int32_t* surv_idx; const int32_t* surv_cnt; float* surv_val; const int32_t* surv_rng;
    const float* xS; const float* WeS; const float* b_enc; float* part;
    int n_rows, S, D, lat_range, n_ranges;
    const int32_t* enable_flag; int enable_when;
};
// sum_shares = false: the caller's final select adds the shares itself (SelectCandArgs::sum_part)
hipError_t launch_refine_slices(const RefineSlicesArgs& a, hipStream_t stream, bool sum_shares = true);
hipError_t launch_select_cand(const SelectCandArgs& a, hipStream_t stream);
hipError_t launch_init_i32(int32_t* p, int32_t v, int n, hipStream_t stream);
hipError_t launch_encoder_init(int32_t* cand_cnt, int n_rows, int32_t* gmax, int n_gmax, hipStream_t stream,
                               int32_t* tau_max = nullptr, const int32_t* enable_flag = nullptr, int enable_when = 0);
// encoder_init + (xnorm != NULL: row margins and scale check of launch_row_margins) + the list flags flags1[0..2] =
// {need_dense = *pre_flag, n_overflow = 0, cand_max = 0} in one launch
hipError_t launch_pre_encode(int32_t* cand_cnt, int n_rows, int32_t* gmax, int n_gmax, const float* xnorm, int D,
                             const float* wg_part, int n_part, const float* scales, int32_t* pre_flag, float* wmax_prev,
                             float* margin, int32_t* flags1, hipStream_t stream);  // scales: {x scale, W scale} (f16r_scales_kernel)
hipError_t launch_heur_gate(float* state, const int32_t* pre_flag, int32_t* gate, hipStream_t stream);
hipError_t launch_heur_update(float* state, const int32_t* bad, const float* cand_mean, int k, const int32_t* gate,
                              hipStream_t stream);
hipError_t launch_step_zero(saev_step_stats* stats, float* upper, int32_t* flag0, hipStream_t stream);
hipError_t launch_wnorm_max(const float* W_encT, int S, int D, float* wg_scratch, float* wmax, hipStream_t stream);
hipError_t launch_f16r_scales(const float* xmax_part, int n_part, const float* wmax, float* scales, hipStream_t stream);
hipError_t launch_pow2_scale(const float* absmax, float* pair, hipStream_t stream);
hipError_t launch_center_stats(const float* x, const float* mu, int n, int D, float* xnorm, float* wg_absmax,
                               hipStream_t stream, const float* xmax = nullptr);  // ||x_b - mu|| per row (squares taken relative to
                                                                                  // *xmax when given), max |x - mu| per workgroup of 4 rows
hipError_t launch_row_margins(const float* xnorm, int n, int D, const float* wg_part, int n_part, const float* scales,
                              int32_t* pre_flag, float* wmax_prev, float* margin, hipStream_t stream);  // scales: {x scale, W scale}
// see overflow_check_kernel (select.hip) for the two-stage use
hipError_t launch_overflow_check(const int32_t* cand_cnt, int n_rows, int cap, const int32_t* pre_flag,
                                 int32_t* need_dense, int32_t* n_overflow, int32_t* cand_max, hipStream_t stream,
                                 int32_t* dense_out = nullptr, int32_t* run_out = nullptr, float* cand_mean = nullptr,
                                 const int32_t* enable_flag = nullptr, const int32_t* dense_src = nullptr);

// per-row statistics produced by the decode kernels (reduced by stats_reduce)
struct __attribute__((aligned(8))) RowStats {
    float sse_scaled;  // sum_d ((x_hat/u - x/u)^2 * u * u)      objectives.py:227-237
    float l0;          // count(val != 0)                         objectives.py:150
    float l1;          // sum |val|                               objectives.py:151
    float aux_sse;     // sum_d (aux_recon - residual)^2          modeling.py:103
    double sse64;      // sum_d (x - x_hat)^2 in fp64             train.py:398-401
    double sumsq64;    // sum_d x^2 in fp64                       train.py:383
};

struct DecodeArgs {
    const float* x;         // (n_rows, D)
    const int32_t* idx;     // (n_rows, code_stride) ascending, -1 padded
    const float* val;
    int code_stride, k;
    const float* W_dec;     // (S, D)
    const float* b_dec;     // (D)
    int n_rows, D, S;
    int idx_limit;          // only latents < idx_limit contribute (Matryoshka prefix); S = all
    const float* upper;     // device scalar max|x|
    float gscale;           // 2 / (n_rows * D)
    int training;           // write g/fired
    float* g;               // (n_rows, D)   d loss / d x_hat
    float* x_hat;           // (n_rows, D) or NULL
    int32_t* fired;         // (S)
    RowStats* rowstats;     // (n_rows) or NULL
    // optional (training): slice-major copies of g and x, [D / 32][n_rows][32 floats] (decode_matry_kernel: of the suffix sums,
    // [D / 32][P][n_rows][32]), for the column-sliced weight-gradient passes (launch_dw_slices)
    float* gS;
    float* xS;
    // optional (training, k <= 32, D = 256 / 512 / 768 / 1024): dval_out[b][j] = <g_b, W_dec[idx[b][j]]> (n_rows, code_stride), formed
    // from the decoder rows while they are still in registers (decode_q_kernel) -- the backward's pass A then needs no W_dec slices
    float* dval_out;
    // optional (training): the CSC build's bit map (S x csc_words words, all zero on entry) -- the decode reads every code anyway and
    // sets bit (latent, row) for it, so that the backward's build starts at its count pass (launch_csc_build: prefilled)
    uint32_t* csc_bitmap;
    int csc_words;
};
hipError_t launch_decode(const DecodeArgs& a, hipStream_t stream);
// whether launch_decode forms dval_out for this shape (otherwise the pointer is ignored and pass A forms the products)
bool decode_forms_dval(int D, int k);        // plain decode: top_k <= 64 (two register halves above 32), d_model 256 .. 1280
bool decode_matry_forms_dval(int D, int k);  // Matryoshka decode: top_k <= 32, d_model <= 1024
// The same decode out of 32-column slices of W_dec that an XCD's L2 holds (decode_s_kernel + decode_s_finish_kernel): the route of a
// training step whose normalize_rows has just left the slice-major copy WdS.  part: per (slice, row) {sse_scaled, pad, sse64, sumsq64}
// (n_slices x n_rows x 3 doubles), dvp: per slice the 32 dval shares of every row (pitch dvp_pitch floats per slice, >= n_rows * 32).
struct DecodeSliceArgs {
    DecodeArgs d;          // W_dec unused; dval_out receives the summed shares (n_rows, code_stride)
    const float* WdS;      // [D / 32][S][32]
    double* part;
    float* dvp;
    long dvp_pitch;
};
bool decode_slices_supported(int D, int S, int k, int code_stride);
hipError_t launch_decode_slices(const DecodeSliceArgs& a, hipStream_t stream);

// Matryoshka prefixes (objectives.py:125-138, modeling.py:369-409): P ascending cut points ending at S; prefix p
// reconstructs from the codes with latent index < cuts[p].
constexpr int MAX_PREFIXES = 16;
struct MatryArgs {
    int P;
    int32_t cuts[MAX_PREFIXES];
    float* G;  // (n_rows, P, D): first dL/dx_hat_p, then (in place) C_p = sum_{p' >= p} dL/dx_hat_p'
    // decode_matry_q_kernel only (DecodeArgs::dval_out set): 0 = only block 0 of G is written (C_0: db_dec's column sums) -- the
    // backward that follows is known to read the slice-major copy alone (saev_train_step)
    int g_rows_all;
};
hipError_t launch_decode_matry(const DecodeArgs& a, const MatryArgs& m, hipStream_t stream);

// CSC (latent-major) view of the codes, built deterministically through a (S x n_rows)-bit map.
struct CscArgs {
    const int32_t* idx;     // (n_rows, code_stride)
    int code_stride, k;     // k = host upper bound of codes per row
    const int32_t* k_dev;   // optional device-side count (aux); <= 0 -> kernels exit
    int n_rows, S;
    uint32_t* bitmap;       // (S, words)
    int words;              // ceil(n_rows / 32) rounded up to a multiple of 8 (one 256-row group = 32 bytes)
    int32_t* grp_prefix;    // (S, words / 8) codes of the latent in earlier 256-row groups
    int32_t* counts;        // (S)
    int32_t* scan_totals;   // (ceil(S / 1024), 3) scratch of the two-pass scan
    int32_t* starts;        // (S + 1)
    int2* pairs;            // (n_rows * k) -> {row b, flat code position b*code_stride + j}
    int32_t* chunk_starts;  // (S + 1) or NULL
    int32_t* part_starts;   // (S) partial-sum slot of each multi-chunk latent (with chunk_starts)
    int32_t* work_latent;   // (max_work) or NULL
    // optional (launch_dw_slices): per pair, in pair order, {128 * row | DWS_FIRST / DWS_LAST flags, coefficient bits} and the latent
    int2* pv;
    int32_t* plat;
    const float* val;       // (n_rows, code_stride) coefficients (with pv)
    // optional (with pv, P <= 1): the decode has left dval (n_rows, code_stride); pv2 = pv with dval as the coefficient is written here
    // instead of by dw_dval_sum_kernel
    int2* pv2;
    const float* dval;
    int P;                  // (with pv) Matryoshka: the pair word carries the virtual row p(latent) * n_rows + row; <= 1: plain
    int32_t cuts[16];       // MAX_PREFIXES
    int32_t* zero_word;     // optional: set to 0 by the build (DwSlicesArgs::cut_list's counter)
    int32_t epoch;          // != 0: one scan launch with look-back; scan_totals then holds 4 words per 1 024 latents, the last the
                            // epoch of the build that wrote them (a value no earlier build of this context used)
};
// bitmap_clean: the whole bit map is known to be zero (dw_combine_kernel cleared it after the previous build)
// colsum_*: optional column sums out[d] = sum_b m[b][d] (b < a.n_rows; partials: ceil(n_rows / 64) * D floats) computed in the
// grids of the fill / count launches (db_dec of the same backward: two launches instead of four)
hipError_t launch_csc_build(const CscArgs& a, hipStream_t stream, bool bitmap_clean = false, const float* colsum_m = nullptr,
                            int colsum_D = 0, long colsum_row_stride = 0, float* colsum_partials = nullptr,
                            float* colsum_out = nullptr, bool prefilled = false);  // prefilled: the decode has set the bits (DecodeArgs::csc_bitmap)

constexpr int DW_CHUNK = 64;   // pairs per work item of the weight-gradient kernels

struct DwRowsArgs {
    const int32_t* starts;        // (S + 1) pair offsets per latent
    const int32_t* chunk_starts;  // (S + 1) work-item offsets per latent
    const int32_t* work_latent;   // (n_work) latent of each work item
    const int32_t* part_starts;   // (S) first partial slot of a multi-chunk latent
    const int2* pairs;
    const float* val;             // coefficient of g rows   (indexed by pairs[].y)
    const float* W_dec;           // (S, D): dval = <g row, W_dec[latent]> is formed inside (coefficient of x rows and db_enc)
    const float* g;               // (n_rows, D), or (n_rows, P, D) suffix sums when P > 1
    const float* x;               // (n_rows, D)
    int D, S;
    int P;                        // Matryoshka prefixes (1 = plain)
    int32_t cuts[MAX_PREFIXES];
    const int32_t* k_dev;         // optional predicate (aux)
    int accumulate;
    float* dW_dec;                // (S, D)
    float* dW_encT;               // (S, D) scratch, transposed into the (D, S) gradient afterwards
    float* db_enc;                // (S)
    float* partials;              // (max_part, 2, D)
    float* db_partials;           // (max_part)
    int lat_lo, lat_hi;           // only latents in [lat_lo, lat_hi) are processed (data-parallel overlap: rows become
                                  // final range by range)
    int part;                     // 0: both gradients; 1: dW_dec only (dval stored); 2: dW_encT + db_enc only (dval loaded)
    float* dval;                  // (n_rows * k) dot products <g row, W_dec[latent]> in pair order (part 1 -> part 2)
    // optional (part != 2): per latent {sc, q} of the decoder-gradient row as written: sc = <g_i, w_i> / ||w_i||^2 (0 when
    // project == 0 or w_i == 0), q = ||g_i||^2 - sc <g_i, w_i> = the squares of the projected row (modeling.py:419-445) --
    // the tail then applies the projection inside Adam and never streams the gradient for it (saev_train_step only)
    float2* row_proj;
    int project;
    float* enc_sq;                // optional (part != 1): per latent the sum of squares of its dW_enc^T row as written
    // optional: dw_rows_kernel zeroes the CSC bit map words of its pairs (row pitch clear_words uint32 per latent) -- the
    // pairs have been placed by then -- so that the next step's csc build starts from a clean map without a pass of its own
    uint32_t* clear_bitmap;
    int clear_words;
};
hipError_t launch_dw_rows(const DwRowsArgs& a, int max_work, hipStream_t stream);

// The same gradients from COLUMN slices (the route of a one-pass backward over all latents when d_model % 32 == 0, P == 1).
// dw_rows gathers whole rows of g and x -- 2 k 4D bytes per activation row out of two matrices far larger than an XCD's L2,
// every byte of it from the fabric.  A 32-column slice of g is n_rows x 128 B (2 MB at 16 384 rows): XCD x works through slices
// x, x + 8, ... one after the other (workgroup b runs on XCD b % 8), so its gathers hit in its 4 MB L2.  g and x are read from
// slice-major copies the decode kernel leaves ([slice][row][32]: consecutive rows of a slice are consecutive 128-byte lines;
// at a row pitch of 4 D bytes they would all fall into one L2 channel).  An eight-lane group (one 128-byte line per pair) walks
// a fixed RUN of DWS_RUN consecutive pairs of the latent-major pair list: no load imbalance whatever the firing histogram;
// a latent that ends inside the run is stored directly when it also began there, otherwise as the run's head / tail partial.
//   pass A: dW_dec slices from g, per-slice shares of dval = <g[b,:], W_dec[i,:]>;   dval_sum: adds the D / 32 shares;
//   pass B: dW_enc^T slices from x, weighted with dval;   finalize: one wave per latent and gradient row: partials summed in
//   run order, unused latents zeroed, db_enc, and the row statistics dw_rows / dw_combine leave (row_proj, enc_sq).
constexpr int DWS_RUN = 64, DWS_SLICE = 32;
constexpr int DWS_FIRST = 1, DWS_LAST = 2, DWS_END = 4;  // pair word flags (END is set by the kernels: last pair of a run)
struct DwSlicesArgs {
    const int32_t* starts;   // (S + 1); starts[S] = number of pairs
    const int2* pv;          // (pairs) from the CSC build
    int2* pv2;               // (pairs) scratch: pv with the coefficient replaced by dval
    const int32_t* plat;     // (pairs)
    const float* gS;         // [D / 32][P][n_rows][32] (P > 1: the suffix sums C_p; a pair of latent i reads block p(i))
    const float* xS;
    const float* W_dec;      // (S, D)
    int n_rows, D, S, pair_cap;  // pair_cap: pitch of dvp (>= n_rows * k)
    int P;                   // Matryoshka prefixes (<= 1: plain)
    float* dvp;              // (D / 32, pair_cap)
    float* dW_dec;           // (S, D)
    float* dW_encT;          // (S, D)
    float* db_enc;           // (S)
    float* part_dec;         // (2 * ceil(pair_cap / DWS_RUN), D) head / tail partial of every run
    float* part_enc;
    int32_t* cut_lat;        // (ceil(pair_cap / DWS_RUN)) per run: the latent that begins in it and is cut at its end, or -1
    int32_t* cut_list;       // optional, 4 x (1 + ceil(pair_cap / DWS_RUN)) ints: [0] = how many runs have cut_lat >= 0 (zeroed by the CSC
                             // build: CscArgs::zero_word), then from [4] on one {run, latent, starts[latent], starts[latent + 1]} per such
                             // run, in arrival order -- the light finalize walks this list
    float2* row_proj;        // optional, as DwRowsArgs
    int project;
    float* enc_sq;           // optional
    uint32_t* clear_bitmap;  // optional, as DwRowsArgs
    int clear_words;
    // optional (saev_train_step only): lat_unused[i] = 1 for a latent without pairs; its dW_enc^T row is then NOT written (the
    // scratch is read by the fused Adam alone, which takes the flag for a row of zeros -- and skips reading the zeroed dW_dec row)
    int32_t* lat_unused;
    // pv2 already holds dval (CscArgs::pv2): pass A forms dW_dec only, no dval shares, no dw_dval_sum_kernel
    int have_dval;
    // optional (the "light" finalize: a one-pass backward whose decode left dval): wn2[i] = ||w_i||^2 as normalize_rows left it.
    // The finalize is then ONE launch that reads the two gradient rows of a latent once for their squares and takes
    // <dW_dec[i], w_i> = sum over the latent's pairs of val * dval (dW_dec[i] = sum val g_b and dval = <g_b, w_i>) from the pair
    // lists: the decoder rows are not read (134 MB per step at configs[1]), dw_finalize_cut / dw_clear_bitmap ride in the launch.
    const float* wn2;
    // optional (with wn2): the passes add up the squares of every piece they store WHOLE, per wave -- sq_wave_dec / sq_wave_enc
    // [grid x 4] -- and the finalize then reads no gradient row at all: the clip norm needs the squares of all rows, not of each,
    // and only the projection correction -sc_i <g_i, w_i> is per latent (row_proj[i].y; enc_sq[i] = 0).  Latents cut by run
    // boundaries keep their full per-row statistics (their pieces are partial sums); scatter_add_dead_kernel, which replaces the
    // statistics of the rows it adds to, takes out what the waves counted for them (its `starts` argument).
    float* sq_wave_dec;
    float* sq_wave_enc;
};
// part as in DwRowsArgs (0 both gradients; 1 decoder half: passes A + dval sums; 2 encoder half: pass B, after part 1)
hipError_t launch_dw_slices(const DwSlicesArgs& a, int max_pairs, int part, hipStream_t stream);
int dw_slices_waves(int D, int max_pairs);
// [D / 32][n][32] copies of two row-major (n, D) matrices (the gathered rows of a sparse-state exchange)
hipError_t launch_slice_major_copy(const float* g, const float* x, int n, int D, float* gS, float* xS, hipStream_t stream);
// sq_part: optional, transpose_blocks(S, D) doubles = per-tile sums of squares of `in`
hipError_t launch_transpose(const float* in, float* out, int S, int D, hipStream_t stream, double* sq_part = nullptr);
int transpose_blocks(int S, int D);

// out[d] (+)= sum_b m[b][d]; `partials` holds ceil(n_rows/64) * D floats.  k_dev: optional device count -- nothing runs
// when it is <= 0, and with col_mult > 0 only the first *k_dev * col_mult columns are summed (the rest is not touched)
hipError_t launch_colsum(const float* m, int n_rows, int D, float* partials, float* out, int accumulate,
                         const int32_t* k_dev, hipStream_t stream, long row_stride = 0, float out_scale = 1.0f,
                         int col_mult = 0);
// + max |m| from the same pass; with `zero_stats` the finishing launch also clears the step's statistics block and the
// force-dense flag (launch_step_zero's job)
hipError_t launch_colsum_absmax(const float* m, int n_rows, int D, float* partials, float* out, float* wg_scratch,
                                float* absmax_out, hipStream_t stream, float out_scale = 1.0f,
                                saev_step_stats* zero_stats = nullptr, int32_t* zero_flag = nullptr);

// ---- tail.hip: HBM-bound streaming kernels over the parameter-sized buffers -------------------
// WS: optional slice-major copy [D / 32][S][32] of the normalised rows (d_model % 32 == 0)
// wn2 (optional): ||row||^2 of every row as written (1 up to rounding; what remove_parallel_grads divides by)
hipError_t launch_normalize_rows(float* W, int S, int D, hipStream_t stream, float* WS = nullptr, float* wn2 = nullptr);
// rows [0, S) of gW projected orthogonal to the rows of W (project != 0); with sq_partials, ceil(S / 4) doubles: the sums
// of squares of the rows as written (the clip norm's share of W_dec, from the same pass)
hipError_t launch_rpg(float* gW, const float* W, int S, int D, hipStream_t stream, double* sq_partials = nullptr,
                      int project = 1);
int sumsq_blocks();  // partial sums one launch_sumsq_partials writes
hipError_t launch_sumsq_partials(const float* g, long n, double* partials, hipStream_t stream);
hipError_t launch_sumsq_final(const double* partials, int nb, double* total, hipStream_t stream);
// total[0] = sum of squares of g[0..n) (deterministic two-stage; `partials` >= 1024 floats... doubles)
hipError_t launch_sumsq(const float* g, long n, double* partials, double* total, hipStream_t stream);
struct AdamArgs {
    float* p; const float* g; float* m; float* v;
    long n;
    float lr, beta1, beta2, eps, bc1, bc2_sqrt;  // bias corrections for this step
    float omb1, omb2;       // 1 - beta1, 1 - beta2 formed in DOUBLE and rounded once, as torch does with its Python-float betas
                            // (1.f - 0.999f is 1.3e-5 below float(1 - 0.999): fixture G8 on the HIP kernel found it)
    float grad_scale;       // applied to g before everything else (1/world_size)
    float max_norm;         // torch semantics for >= 0 (0 zeroes the gradient); < 0 disables clipping
    const double* sumsq;    // device: sum of squares of the *unscaled* grads
    saev_step_stats* stats; // grad_norm written here by block 0
};
hipError_t launch_adam(const AdamArgs& a, hipStream_t stream);
// the decoder rows [0, S) of a.p / a.g / a.m / a.v with the projection coefficient row_proj[i].x applied to the gradient
hipError_t launch_adam_rows(const AdamArgs& a, const float2* row_proj, int S, int D, hipStream_t stream);
// total = sum(partials[0..nb)) + sum_i row_proj[i].y + |e1|^2 + |e2|^2 (see sumsq_final_ex_kernel)
// plain (optional): n_plain floats added as they are (DwSlicesArgs::sq_wave_dec / _enc, contiguous)
hipError_t launch_sumsq_final_ex(const double* partials, int nb, const float2* row_proj, int n_rows, const float* e1, long n1,
                                 const float* e2, long n2, double* total, double* blk_part, int* ticket, hipStream_t stream,
                                 const float* enc_sq = nullptr,  // + sum_i enc_sq[i] (squares of the rows of the transposed dW_enc)
                                 const float* plain = nullptr, long n_plain = 0);
// Adam over everything in one launch with the gradients where the backward left them: a.{p,g,m,v} = the flat buffers, the
// decoder rows projected through row_proj, W_enc's gradient read from the transposed scratch gT (S, D) through LDS tiles,
// the two bias segments [off, off + n) element-wise (adam_fused_kernel)
// lat_unused (optional): latents whose gradient rows are zero and not to be read (DwSlicesArgs::lat_unused)
// optional: the W_enc tiles of the fused Adam also write what the NEXT forward of the f16r encoder needs of W_enc -- its fp16
// images, the slice-major fp32 W_enc^T, the per-image shares of <mu, w>, ||w||^2 and the rounding-error norm (split_wT_body<2>'s
// outputs, same layout) -- scaled with the power of two derived from *wmax_prev (written to scales_next[1]).  The forward then
// never reads W_enc.
struct AdamImageArgs {
    _Float16* ws; float* WeS; double* dot_part; float* sq_part;
    const float* mu; const float* wmax_prev; float* scales_next;
    int nks, S_pad;
    int mode;  // 0: the f16r set described above; 1: the bf16 encoder -- only ws, W_enc^T rounded to bf16 (nothing else is read)
    // Ownership check (include/saev_amd.h: PARAMETER OWNERSHIP): the launch leaves two 32-bit checksums of every W_enc tile AS IT
    // WRITES IT (chk: [tiles][2], a plain sum and a sum of position-rotated words of the bit patterns) and, with `verify`, compares
    // the checksums of the tile AS IT READS IT with what the previous launch left: they differ exactly when somebody else wrote the
    // tile in between -- i.e. when the step that ends here ran on operand images of other values.  Every element is covered at no
    // extra traffic (Adam reads and writes all of W_enc anyway); a mismatch counts into *late (pinned host memory).
    uint32_t* chk; int32_t* late; int verify;
    const int32_t* early;  // != 0: the step's first kernels found the write themselves and the step took the exact route (XprepArgs::stale)
};
hipError_t launch_adam_fused(const AdamArgs& a, const float2* row_proj, const float* gT, int S, int D, long off_b_dec, long n_b_dec,
                             long off_W_enc, long off_b_enc, long n_b_enc, hipStream_t stream, const int32_t* lat_unused = nullptr,
                             const AdamImageArgs* img = nullptr);
constexpr int SUMSQ_EX_BLOCKS = 32;  // blk_part: this many doubles of scratch; ticket: an int, zero between launches

// what the host learns about the dead set of a step without waiting for it (saev_step_dead reads the record of an
// earlier step): n_near bounds the dead count of any later step by which at most horizon_tokens more tokens went by
struct DeadRecord {
    int64_t step;            // 1-based step id the record belongs to
    int64_t cum_tokens;      // tokens seen up to and including that step
    int64_t horizon_tokens;
    int32_t n_dead;
    int32_t n_near;          // latents with toks_since_active >= threshold - horizon_tokens after that step's update
};

struct DeadArgs {
    int64_t* toks;          // (S)
    int32_t* fired;         // (S) consumed and reset to 0
    int32_t* dead;          // (S) out: 1 if dead this step
    int S;
    int64_t add_tokens, threshold;
    int k_aux;
    int32_t* n_dead;        // device scalar out
    int32_t* k_use;         // device scalar out: min(k_aux, n_dead)
    saev_step_stats* stats;
    int32_t* scratch;       // three ints, zero between launches: running count, block ticket, running near-count
    int64_t horizon_tokens; // see DeadRecord
    int64_t step, cum_tokens;
    DeadRecord* rec;        // device-visible pointer into pinned host memory, or NULL
    int32_t* dead_list;     // optional (launch_stats_dead only): the ascending list of dead latents, written when any are dead
};
hipError_t launch_dead_update(const DeadArgs& a, hipStream_t stream);
// launch_stats_reduce (with_aux = 0) and launch_dead_update in one launch (a fused train step: the two meet nowhere)
hipError_t launch_stats_dead(const RowStats* rs, int n_rows, int D, int P, float alpha, const float* upper, const int32_t* n_overflow,
                             saev_step_stats* stats, double* scratch, const int32_t* cand_cnt, int cand_cap, const DeadArgs& d,
                             hipStream_t stream);
hipError_t launch_absmax(const float* x, long n, float* out_zeroed, hipStream_t stream);
hipError_t launch_gather_rows(const float* pool, const int64_t* rows, int n_rows, int D, float* out, hipStream_t stream);
hipError_t launch_scatter_dense(const int32_t* idx, const float* val, int n_rows, int k, int stride, int S, float* f,
                                hipStream_t stream);
// reduce rowstats[0..n_rows) into *stats (mse, l0, l1, aux, sse, sum_sq)
// with_aux: 0 no auxiliary term, 1 add it, 2 add it iff *n_dead_dev > 0 (and do nothing at all otherwise)
hipError_t launch_stats_reduce(const RowStats* rs, int n_rows, int D, int P, float alpha, int with_aux, const float* upper,
                               const int32_t* n_overflow_and_max, saev_step_stats* stats, hipStream_t stream,
                               const int32_t* n_dead_dev, double* scratch,  // scratch: STATS_SCRATCH_DOUBLES doubles, zeroed once
                               // optional: take n_overflow_rows / cand_max from the candidate lists themselves
                               const int32_t* cand_cnt = nullptr, int cand_cap = 0);
constexpr int STATS_SCRATCH_DOUBLES = 16 * 8 + 1;

// ---- f16x3 encoder (fp32-accurate split-fp16 MFMA) -------------------------------------------------
struct EncodeF16Args {
    const _Float16* xs;       // (rows padded to 256, 2*Dp): per 16-wide k-step [hi 16 | lo 16]
    const _Float16* ws;       // (S padded to 256, 2*Dp): W_enc transposed, scaled by w_scale, same interleave
    const float* b_enc;       // (S)
    int n_rows, Dp, S;
    float w_scale;
    int arith;                // image mode of the operands: 0 fp16 hi/lo, three products (fp32-accurate); 1 bf16, 2 fp16:
                              // one product
    const float* row_margin;  // (n_rows) or NULL: candidates are kept down to bound - row_margin[row] (EPI_TOPK)
    const float* scale_dev;   // NULL or two device floats: extra power-of-two scales of the x and W images
    const float* scale_dev_b; // optional: the W images' scale lives elsewhere (*scale_dev_b instead of scale_dev[1])
    int s_splits;
    int no_rot;            // 1: every workgroup walks the k-steps of a tile in the same order (saev_debug_cfg.enc_rot)
    float* h_out;             // EPI_DENSE
    int ngroups;              // EPI_TOPK: 32 (bound = min over 32 group maxima; needs top_k <= 32) or 64 (bound = top_k-th
                              // largest of 64 group maxima; top_k <= 64)
    int top_k;
    int32_t* gmax;            // (ngroups, gmax_stride) shared per-row group maxima, init INT32_MIN
    int gmax_stride;
    int32_t* cand_cnt;
    float* cand_val;
    int32_t* cand_idx;
    int cand_cap, cand_stride;
    const int32_t* enable_flag;
    int enable_when;
    // EPI_TOPK, ngroups == 32: predicted row bounds instead of guaranteed ones.  heur_z: device scalar z of "bound = mean +
    // z * sigma of the row's pre-activations over the workgroup's first tile"; tau_max: (n_rows) ordered-int keys, init
    // INT32_MIN, receives the largest bound used for each row (the select stage verifies the prediction against it)
    const float* heur_z;
    int32_t* tau_max;
    // EPI_TOPK: the guaranteed bound of a row is refreshed on a workgroup's first `refresh_first` tiles and on every
    // `refresh_every`-th (a power of two) after that
    int refresh_first, refresh_every;
    // EPI_DENSE only: a batch of independent products (grid.y), used to split a long contraction into slices whose
    // partial outputs are summed afterwards (AuxK weight gradients contract over the batch axis).  Batch j reads the
    // k-steps [j*nks, (j+1)*nks) of every row block -- blk_imgs is the number of images a row block has in memory
    // (0: nks, i.e. no slicing) -- and writes h_out + j*out_bstride.
    int n_batches;
    int blk_imgs;
    long out_bstride;
    int mfma32;               // host side only: 1 = the 32x32x16 kernels for the single-product modes too (saev_debug_cfg.enc_mfma)
};
hipError_t launch_encode_f16x3(const EncodeF16Args& a, int epi, hipStream_t stream);
int encode_f16x3_tile_rows();
int encode_f16x3_tile_latents();
// image mode: 0 = fp16 hi/lo (16 k per image), 1 = bf16 single, 2 = fp16 single (32 k per image)
hipError_t launch_split_rows(const float* x, int n, int D, int Dp, void* xs, int mode, hipStream_t stream, float scale = 1.0f,
                             const float* scale_dev = nullptr,  // effective scale = scale * *scale_dev
                             const float* mu = nullptr,  // rows are x - mu
                             float* xS = nullptr);  // mode 2: also the slice-major fp32 copy of x, [D / 32][n][32] (D % 32 == 0)
hipError_t launch_split_wT(const float* W, int D, int S, int S_pad, int Dp, float scale, void* ws, int mode,
                           hipStream_t stream, const float* scale_dev = nullptr,
                           // mode 2 with mu (f16r): also dot_part[ks][s] = shares of <mu, W*scale>, sq_part = shares of
                           // ||W*scale||^2 per column, W_T = fp32 transpose of W
                           const float* mu = nullptr, double* dot_part = nullptr, float* sq_part = nullptr,
                           float* W_T = nullptr,
                           int wt_slices = 0);  // W_T written slice-major: [D / 32][S][32]
// launch_split_rows(mode 2, scale_dev = scales, mu) and launch_split_wT(mode 2, scale_dev = scales + 1, mu, ...) in ONE launch
// The streamed f16r step (split.hip: xprep_kernel): the fp16 images of x - mu, the slice-major copy of x, the pieces of the row
// norms, of the next centring vector and of the batch maxima -- one pass over x (optionally gathered from a pool by row index),
// centred / scaled / normalised with what the PREVIOUS batch left (mu, scales[0], *up_prev).
struct XprepArgs {
    const float* x;        // (n, D) row-major -- or the pool the rows are drawn from
    const int64_t* rows;   // optional (n): row r of the batch is pool row rows[r]
    float* x_out;          // optional (with rows): the batch, contiguous (n, D)
    int n, D, nks, n_pad;  // nks = D / 32 images per row block; n_pad = row pitch of xn_part (multiple of 256)
    const float* scales;   // scales[0]: x scale of this step's images (a power of two); scales[4]: the squares are taken of (x - mu) / scales[4]
    const float* mu;       // (D) centring vector
    _Float16* xs;          // images
    float* xS;             // [D / 32][n][32]
    float* xn_part;        // [nks][n_pad][2]
    float* col_part;       // [row blocks][D]
    float* amax_part;      // [row blocks * nks]
    float* cmax_part;      // [row blocks * nks]
    // Safety net behind the parameter-ownership contract (include/saev_amd.h): every workgroup compares two pseudo-random elements
    // of W_enc with the slice-major copy the images came with, and the launch compares ALL of b_enc with the copy bias_finish kept
    // -- equal bit for bit unless somebody wrote the parameters without saying so.  A difference raises *stale: pre_encode2_kernel
    // then sends the step down the exact dense route and tells the host (any write to b_enc and a bulk write to W_enc -- a copy, an
    // optimizer step, a broadcast -- are caught here, before the images are used; a poke at a handful of elements of W_enc is caught
    // with certainty at the END of the step, by the tile checksums of the fused Adam: AdamImageArgs::chk).
    const float* W_enc; const float* WeS; const float* b_enc; const float* b_seen;
    int S; uint32_t salt; int32_t* stale;
    // A context that LENDS its x-derived buffers (saev_share_x) keeps what its followers need of this step's x side beyond the
    // step's second launch, which moves mu on: mu_keep (D) = the centre these images were formed with, xside_keep[0] = their scale
    // (workgroup 0 copies both; pre_encode2_kernel adds xside_keep[1] != 0 when the images left fp16's range).  NULL otherwise.
    float* mu_keep; float* xside_keep;
};
hipError_t launch_xprep(const XprepArgs& a, hipStream_t stream);
// ... and its second launch (select.hip: pre_encode2_kernel): row norms / margins, encoder state, batch maxima, flags, next mu.
struct PreEncode2Args {
    int32_t* cand_cnt; int n_rows;
    int32_t* gmax; int n_gmax;
    const float* xn_part; int nks, n_pad, D;
    const float* wg_part; int n_part;        // bias_finish_kernel's per-workgroup maxima (|b_shift|, column norms, rounding-error norms)
    const float* scales;                     // {x scale, W scale, x scale, 1, square normaliser} of this step
    float* scales_next;                      // [0], [2] receive the next step's x scale, [4] its normaliser (this batch's max |x|)
    int32_t* pre_flag;                       // the step's force-dense flag (written)
    float* wmax_prev;
    float* margin;                           // (n_rows)
    float* xnorm;                            // optional (2 n_rows)
    int32_t* flags1;                         // need_dense, n_overflow, cand_max
    const float* col_part; int n_rowblk;     // xprep's column sums per row block
    float* mu; float inv_n; int update_mu;
    const float* amax_part; const float* cmax_part; int n_img;
    float* upper;                            // max |x| of this batch
    int32_t* stale; int32_t* stale_host;     // XprepArgs::stale (consumed and cleared here); optional pinned word the host polls
    float* xside_keep;                       // optional: [1] = 1 when this step's x images are unusable (XprepArgs::xside_keep)
    saev_step_stats* stats;                  // zeroed
    int nb_rows;                             // (set by the launcher)
};
hipError_t launch_pre_encode2(PreEncode2Args a, hipStream_t stream);
// A context that borrows the x side of a STREAMED step (saev_share_x): scales[0] = scales[2] = the x scale of the lender's images,
// scales[3] = 1, scales[1] = the power-of-two W scale from *wmax unless keep_w (its own Adam has left it); the lender's verdict on the
// images (xside[1] != 0: out of fp16's range) raises *pre_flag.
hipError_t launch_follower_scales(const float* xside, const float* wmax, float* scales, int32_t* pre_flag, int keep_w, hipStream_t stream);
// Row-operand images (rf: rows of M, k = its columns padded to kp_r) and k-major-operand images (tf: rows = columns of M, k = its rows
// padded to kp_t) of an fp32 matrix M (R x C) in one pass (split.hip: split_both_kernel; either may be NULL).  Images of the split-fp16
// mode (16-wide k-steps, hi | lo); kp_r / kp_t multiples of 16; k-steps past kp are not written, rows / columns past R / C are zeros.
hipError_t launch_split_both(const float* M, int R, int C, float scale, const float* scale_dev, void* rf, int kp_r, void* tf, int kp_t,
                             hipStream_t stream);
hipError_t launch_split_f16r(const float* x, int n, int D, int Dp, void* xs, const float* scales, const float* mu, const float* W,
                             int S, int S_pad, void* ws, double* dot_part, float* sq_part, float* W_T, hipStream_t stream,
                             float* xS = nullptr, int wt_slices = 0);
// f16r: b_shift = float(sum_ks dot_part / *w_scale + b_enc); wg_part[0..nwg) = per-workgroup max |b_shift|,
// wg_part[nwg..2 nwg) = per-workgroup max column norm of W_enc (nwg = ceil(S/256)); launch_row_margins reduces them and
// raises *pre_flag when the largest norm times *w_scale is outside the safe fp16 window, then wmax_prev = that norm
// b_seen (optional): a copy of b_enc as it was read here (XprepArgs::b_seen)
hipError_t launch_bias_finish(const double* dot_part, const float* sq_part, int Dp, int S, int S_pad, const float* w_scale,
                              const float* b_enc, float* b_shift, float* wg_part, hipStream_t stream, float* b_seen = nullptr);
hipError_t launch_max_reduce(const float* v, int n, float* out, hipStream_t stream);

// ---- auxk.hip: AuxK as dense algebra over the compacted dead set -----------------------------------
hipError_t launch_dead_compact(const int32_t* dead, int S, int32_t* list, hipStream_t s,
                               const int32_t* n_dead_dev = nullptr);  // exits at once when *n_dead_dev == 0
hipError_t launch_gather_dead(const float* W_enc, const float* W_dec, const int32_t* dl, int nd, int ndp, int D, int S,
                              float* Wenc_dead, float* Wdec_dead, hipStream_t s,
                              const int32_t* nd_dev = nullptr);  // nd_dev: the list holds min(nd, *nd_dev) entries, the rest is padding
hipError_t launch_dead_bias_vec(const float* b_enc, const int32_t* dl, int nd, int ndp, float* out, hipStream_t s,
                                bool pad_zero = false,  // padding columns: -inf (never selected) or 0 (all-selected mode)
                                const int32_t* nd_dev = nullptr);
hipError_t launch_aux_scatter(const int32_t* idx, const float* val, int n_rows, int k, int stride, int ndp, float* A,
                              uint8_t* mask, hipStream_t s, const int32_t* k_dev = nullptr);  // k_dev: min(k, *k_dev) codes per row
hipError_t launch_aux_resid(float* E, const float* x, const float* x_hat, const float* b_dec, int n_rows, int D,
                            float gscale, RowStats* rowstats, hipStream_t s, const int32_t* nd_dev = nullptr, float* part = nullptr,
                            float* pair = nullptr);
// max |.| and the power-of-two operand scale {2^e, 1} as by-products (auxk.hip: pow2_parts_kernel; part: absmax_parts_max(rows) floats)
int absmax_parts_max(int n_rows_cap);
hipError_t launch_absmax_pow2(const float* x, long n, float* part, float* pair, hipStream_t s);
hipError_t launch_mask_apply_absmax(float* dA, const uint8_t* mask, long n, float* part, float* pair, hipStream_t s);
bool aux_select_supported(int ndp);
hipError_t launch_aux_select(const float* H, int n_rows, int ndp, int k, const int32_t* k_dev, float* A, uint8_t* mask, float* part,
                             float* pair, hipStream_t s);  // *nd_dev <= 0: zero gradient, zero loss
hipError_t launch_mask_apply(float* dA, const uint8_t* mask, long n, hipStream_t s);
// a handful of dead latents (nd <= AUX_SMALL_MAX, all of them selected): row-wise forward, block-wise weight gradients
constexpr int AUX_SMALL_MAX = 64;  // (one lane per dead latent in aux_small_fwd_kernel: at most the wave width)
constexpr int AUX_SMALL_DEFAULT = 40;  // largest dead set that takes them unless saev_debug_cfg.aux_small_max says otherwise
// The few-dead-latents kernels take the dead count from the device (*nd_dev; they exit unless 1 <= nd <= AUX_SMALL_MAX), so
// the host can enqueue them without knowing it.  Leading dimension of A / dA and row count of the compact weight buffers:
// AUX_SMALL_MAX.
hipError_t launch_gather_dead_small(const float* W_enc, const float* W_dec, const int32_t* dl, const int32_t* nd_dev, int D,
                                    int S, float* WencT_dead, float* Wdec_dead, hipStream_t s, int cap = AUX_SMALL_MAX);
hipError_t launch_aux_small_fwd(const float* x, const float* x_hat, const float* WencT_dead, const float* Wdec_dead,
                                const float* b_enc, const float* b_dec, const int32_t* dl, int n_rows, int D,
                                const int32_t* nd_dev, float gscale, float* A, float* dA, float* g_aux, RowStats* rowstats,
                                hipStream_t s);
// dWd / dWe (nd x D each) = the block partials of launch_aux_small_wgrad summed in block order
hipError_t launch_aux_small_wsum(const float* part, int n_blk, int D, const int32_t* nd_dev, float* dWd, float* dWe, hipStream_t s,
                                 int ndp = AUX_SMALL_MAX);  // ndp: rows per half of a block partial
// at most AUX_FUSED_MAX dead latents (d_model 256 / 512 / 768 / 1024): codes, loss, g_aux, dA and the block partials of all four
// gradients in one pass over x and x_hat (auxk.hip: aux_small_fused_kernel).  part: blocks x 2 x AUX_FUSED_MAX x D (the layout of
// launch_aux_small_wgrad with AUX_FUSED_MAX rows per half), partb: blocks x D (db_dec's share), partbe: blocks x AUX_FUSED_MAX
// (db_enc[dl]); blocks = aux_fused_blocks(n_rows)
constexpr int AUX_FUSED_MAX = 8;
constexpr int AUX_FUSED_ROWS = 32;  // activation rows per workgroup
// trips of rows a lane keeps in flight beyond the current one (measured at three dead latents, tools/experiments/r4_aux_pf_ab.sh:
// four latents / four rows per trip 43.6 us at 1, 47.3 at 2 (8 spilled registers), 66 at 3; eight latents / two rows 61.5 at 1)
#ifndef AUX_FUSED_PF8
#define AUX_FUSED_PF8 1
#endif
#ifndef AUX_FUSED_PF4
#define AUX_FUSED_PF4 1
#endif
bool aux_fused_supported(int D);
// 9 ... 64 dead latents on the fp32 matrix cores (auxk.hip: aux_mfma_*): launch_aux_small_fwd's and launch_aux_small_wgrad's outputs
// (A, dA: (n_rows, AUX_SMALL_MAX); g_aux; rowstats.aux_sse; the block partials) from one + one launches; d_model % 128 == 0;
// bound: the host's bound of the dead count (one or two blocks of 32 latents)
constexpr int AUX_MFMA_MAX = 128;  // (beyond AUX_SMALL_MAX = 64 with row pitch AUX_MFMA_MAX: launch_aux_mfma_forward)
bool aux_mfma_supported(int D);
hipError_t launch_aux_mfma_forward(const float* x, const float* x_hat, const float* WencT_dead, const float* Wdec_dead, const float* b_enc,
                                   const float* b_dec, const int32_t* dl, int n_rows, int D, const int32_t* nd_dev, float gscale, float* A,
                                   float* dA, float* g_aux, RowStats* rowstats, hipStream_t s, int bound, int ndp = AUX_SMALL_MAX);
// (ndp: row pitch of A / dA and of the partials -- AUX_SMALL_MAX, or AUX_MFMA_MAX when the bound exceeds it.  One launch per count
// window [1, 32], [33, 64], [65, 128] up to the bound, each predicated on the device-side count)
// (partb: blocks x D, partbe: blocks x ndp -- the blocks' column sums of g_aux and dA; launch_aux_fused_wsum with
// ndo = AUX_SMALL_MAX finishes all four gradients and the auxiliary loss in one launch)
hipError_t launch_aux_mfma_wgrad(const float* A, const float* dA, const float* g_aux, const float* x, int n_rows, int D,
                                 const int32_t* nd_dev, float* part, float* partb, float* partbe, hipStream_t s, int bound, int ndp = AUX_SMALL_MAX);
// the ordered sums of all four partial sets in one launch: dWd / dWe rows, db_dec's share (db_out, added to what is there when
// db_accumulate) and db_enc[dl] (dbe)
hipError_t launch_aux_fused_wsum(const float* part, int n_blk, int D, const int32_t* nd_dev, float* dWd, float* dWe, hipStream_t s,
                                 const float* partb, float* db_out, int db_accumulate, const float* partbe, float* dbe,
                                 // optional: the step's auxiliary loss from the rows' shares as well (stats->aux; a fused train step)
                                 const RowStats* rs = nullptr, int n_rows = 0, float alpha = 0.f, saev_step_stats* stats = nullptr,
                                 int ndo = AUX_FUSED_MAX);  // latent rows per half of a block partial
int aux_fused_blocks(int n_rows);
hipError_t launch_aux_small_fused(const float* x, const float* x_hat, const float* WencT_dead, const float* Wdec_dead, const float* b_enc,
                                  const float* b_dec, const int32_t* dl, int n_rows, int D, const int32_t* nd_dev, float gscale,
                                  float* part, float* partb, float* partbe, RowStats* rowstats, hipStream_t s,
                                  int bound = AUX_FUSED_MAX);  // bound >= the device-side count (<= 4: the four-latent variant)
hipError_t launch_aux_small_wgrad(const float* A, const float* dA, const float* g_aux, const float* x, int n_rows, int D,
                                  const int32_t* nd_dev, float* part, hipStream_t s);  // part: ceil(n/64) x 2 x AUX_SMALL_MAX x D
hipError_t launch_sum_parts(const float* parts, int n_parts, long n, float* out, hipStream_t s);  // out = sum_j parts[j], n % 4 == 0
// nd rows are scattered; with nd_dev the count is *nd_dev (<= nd, which then only sizes the grid)
hipError_t launch_scatter_add_dead(const int32_t* dl, int nd, int D, const float* dWd, const float* dWe, const float* dbe,
                                   float* gW_dec, float* gW_encT, float* gb_enc, int lat_lo, int lat_hi, hipStream_t s,
                                   const int32_t* nd_dev = nullptr, int part = 0,
                                   // optional: refresh row_proj of the rows touched (W_dec = the parameter rows)
                                   float2* row_proj = nullptr, const float* W_dec = nullptr, int project = 1,
                                   float* enc_sq = nullptr,
                                   // optional (DwSlicesArgs::lat_unused): a flagged latent's dW_enc^T row was not written -- it is
                                   // taken as zeros here and the flag cleared (both its rows are gradients now)
                                   int32_t* lat_unused = nullptr,
                                   // optional (DwSlicesArgs::sq_wave_dec): the clip norm holds the squares of a row's main-path
                                   // part already, unless the latent was cut by run boundaries (starts: the CSC offsets):
                                   // the statistics left here are then those of the row as it is now MINUS that part
                                   const int32_t* starts = nullptr);
