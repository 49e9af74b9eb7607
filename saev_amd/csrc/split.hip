// fp32 -> (hi, lo) fp16 operand splitting for the f16x3 encoder (gemm_encode_f16x3.hip), written directly in
// the encoder's LDS image order so the encoder can stream it with perfectly coalesced global_load_lds:
//
//   operand (rows R, k) -> blocks of 256 rows x one 16-wide k-step = one 16 KB "slot image":
//       image(blk, ks) at  ((blk * nks + ks) * 256 * 32) halfs,   nks = Dp / 16
//       inside: row rl (0..255) owns 32 halfs = 4 chunks of 8; logical chunk c = 2*part + h
//               (part 0 = hi, 1 = lo; h = which half of the k-step: k%16 < 8 or >= 8)
//               is stored at position c ^ ((4 - ((rl >> 2) & 3)) & 3)   <- the bank-conflict swizzle of the fragment reads
//
//   split_rows_kernel : x (n, D) fp32 (* scale)  -> xs images (rows padded to 256)
//   split_wT_kernel   : W_enc (D, S) fp32, i.e. k-major -> ws images of W_enc^T * scale (rows = latents)
// hi = fp16(a*scale) (round to nearest even), lo = fp16(a*scale - hi): together 22 significand bits.
// Rows beyond n / S and k beyond D are written as zeros where the kernels cover them; x padding rows are
// zeroed once at context creation.
//
// MODE 1 / 2 (the single-product encoders, bf16 / fp16): same images, but a row's 4 chunks are 32 consecutive k of
// one rounded value each (chunk c = k 8c..8c+7 of a 32-wide k-step, nks = Dp / 32), round to nearest even.
#include "common.h"
#include "kernels.h"

namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ half8 split8(const float (&v)[8], int part) {
    half8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const _Float16 h = (_Float16)v[e];
        o[e] = part == 0 ? h : (_Float16)(v[e] - (float)h);
    }
    return o;
}

__device__ __forceinline__ half8 round8_bf16(const float (&v)[8]) {
    bf16x8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (__bf16)v[e];
    return __builtin_bit_cast(half8, o);
}
template <int MODE>
__device__ __forceinline__ half8 pack8(const float (&v)[8], int part) {
    if constexpr (MODE == 0) return split8(v, part);
    else if constexpr (MODE == 1) return round8_bf16(v);
    else return split8(v, 0);  // fp16(v), single
}

// one thread = one 16-byte chunk of the image; grid.x = ceil(n/256) * nks images, 1024 threads each
// xS (MODE 2, optional): the slice-major fp32 copy of x itself ([k / 32][row][32]) for the kernels that gather 32-column slices
// (refine_slices_kernel, dw_slices_kernel) -- an image of this mode IS a 32-column slice of 256 rows, the values are in registers
template <int MODE>
__device__ __forceinline__ void split_rows_body(int bid, const float* __restrict__ x, int n, int D, int nks, float scale,
                                                const float* __restrict__ scale_dev, const float* __restrict__ mu,
                                                _Float16* __restrict__ xs, float* __restrict__ xS = nullptr) {
    if (scale_dev != nullptr) scale *= *scale_dev;
    const int blk = bid / nks, ks = bid % nks;
    const int i = threadIdx.x;           // chunk index inside the image
    const int rl = i >> 2, p = i & 3;
    const int c = p ^ ((4 - ((rl >> 2) & 3)) & 3);
    const int part = c >> 1, h = c & 1;
    const int r = blk * 256 + rl, k = MODE != 0 ? ks * 32 + c * 8 : ks * 16 + h * 8;
    float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (r < n && k < D) {  // D % 4 == 0: load in two float4s, the second may fall off the end
        f32x4 a = *reinterpret_cast<const f32x4*>(x + (size_t)r * D + k);
        if (MODE == 2 && xS != nullptr) {  // (D % 32 == 0 on this route: the chunk's eight values exist)
            f32x4* o = reinterpret_cast<f32x4*>(xS + ((size_t)ks * n + r) * 32 + c * 8);
            o[0] = a;
            o[1] = *reinterpret_cast<const f32x4*>(x + (size_t)r * D + k + 4);
        }
        if (mu != nullptr) a -= *reinterpret_cast<const f32x4*>(mu + k);  // centred images (f16r)
        v[0] = a[0] * scale; v[1] = a[1] * scale; v[2] = a[2] * scale; v[3] = a[3] * scale;
        if (k + 4 < D) {
            f32x4 b = *reinterpret_cast<const f32x4*>(x + (size_t)r * D + k + 4);
            if (mu != nullptr) b -= *reinterpret_cast<const f32x4*>(mu + k + 4);
            v[4] = b[0] * scale; v[5] = b[1] * scale; v[6] = b[2] * scale; v[7] = b[3] * scale;
        }
    }
    reinterpret_cast<half8*>(xs + (size_t)bid * 256 * 32)[i] = pack8<MODE>(v, part);
}
template <int MODE>
__global__ __launch_bounds__(1024) void split_rows_kernel(const float* __restrict__ x, int n, int D, int nks, float scale,
                                                          const float* __restrict__ scale_dev, const float* __restrict__ mu,
                                                          _Float16* __restrict__ xs, float* __restrict__ xS) {
    split_rows_body<MODE>(blockIdx.x, x, n, D, nks, scale, scale_dev, mu, xs, xS);
}

// one workgroup = one image of W_enc^T: 256 latents x 16 k (32 k for bf16).  The k-rows of W_enc (1 KB each) are
// read coalesced into LDS, then every thread assembles its chunk from a column.
//
// With `mu` given (the f16r encoder) the same pass over W_enc also produces everything else that mode needs from it --
// the tile is in LDS anyway:
//   * this image's share of <mu, W[:, s]> in fp64 -> dot_part[ks][s]   (bias of the centred first pass)
//   * this image's share of ||W[:, s]||^2         -> sq_part[ks][s]    (largest column norm: error margin, next scale)
//   * this image's share of ||dW[:, s]||^2        -> sq_part[nks + ks][s]  (dW = W - its fp16 image: the measured margin)
//   * the fp32 transpose W_T[s][k]                                      (rows for the exact refinement)
// bias_finish_kernel adds the nks shares in a fixed order.  The tile holds W * scale with a power-of-two scale: exact,
// undone where it matters.
template <int MODE>
__device__ __forceinline__ void split_wT_body(int bid, int nblk, float (&tile)[(MODE != 0 ? 32 : 16)][257], float (&mu_s)[(MODE != 0 ? 32 : 16)],
                                              const float* __restrict__ W, int D, int S, int nks, float scale,
                                              const float* __restrict__ scale_dev, _Float16* __restrict__ ws,
                                              const float* __restrict__ mu, double* __restrict__ dot_part,
                                              float* __restrict__ sq_part, float* __restrict__ W_T, int wt_slices = 0) {
    if (scale_dev != nullptr) scale *= *scale_dev;
    constexpr int KS = MODE != 0 ? 32 : 16;
    const int blk = bid / nks, ks = bid % nks;
    const int s0 = blk * 256, k0 = ks * KS;
    if (MODE == 2 && mu != nullptr && threadIdx.x < KS) mu_s[threadIdx.x] = (k0 + threadIdx.x < D) ? mu[k0 + threadIdx.x] : 0.f;
    for (int q = threadIdx.x; q < KS * 64; q += 1024) {  // 16 bytes per lane (S % 4 == 0)
        const int kk = q >> 6, sl = (q & 63) * 4;
        const int k = k0 + kk, s = s0 + sl;
        f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
        if (k < D && s < S) v = *reinterpret_cast<const f32x4*>(W + (size_t)k * S + s) * scale;
        tile[kk][sl] = v[0]; tile[kk][sl + 1] = v[1]; tile[kk][sl + 2] = v[2]; tile[kk][sl + 3] = v[3];
    }
    __syncthreads();
    const int i = threadIdx.x;
    const int rl = i >> 2, p = i & 3;
    const int c = p ^ ((4 - ((rl >> 2) & 3)) & 3);
    const int part = c >> 1, h = c & 1;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = tile[(MODE != 0 ? c : h) * 8 + e][rl];
    reinterpret_cast<half8*>(ws + (size_t)bid * 256 * 32)[i] = pack8<MODE>(v, part);
    if constexpr (MODE == 2) {
        if (mu != nullptr) {  // the four threads of a latent hold its 32 k of this image
            double acc = 0.0;
            float sq = 0.f, dsq = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                acc += (double)mu_s[c * 8 + e] * (double)v[e];
                sq += v[e] * v[e];
                const float d = v[e] - (float)(_Float16)v[e];  // the rounding error this image carries (inf on overflow: margin inf)
                dsq += d * d;
            }
            acc += __shfl_xor(acc, 1, 64);
            acc += __shfl_xor(acc, 2, 64);
            sq += __shfl_xor(sq, 1, 64);
            sq += __shfl_xor(sq, 2, 64);
            dsq += __shfl_xor(dsq, 1, 64);
            dsq += __shfl_xor(dsq, 2, 64);
            if (p == 0) {
                const size_t S_pad = (size_t)nblk / nks * 256;
                const size_t o = (size_t)ks * S_pad + s0 + rl;
                dot_part[o] = acc;
                sq_part[o] = sq;
                sq_part[(size_t)nks * S_pad + o] = dsq;  // second plane: shares of ||dW[:, s]||^2
            }
            const int k = k0 + c * 8;
            if (s0 + rl < S && k < D) {  // D % 4 == 0
                const float inv = 1.0f / scale;  // power of two
                // row-major (S, D), or -- wt_slices -- slice-major [k / 32][latent][32]: this image's 256 x 32 block is contiguous
                f32x4* o = reinterpret_cast<f32x4*>(wt_slices ? W_T + ((size_t)ks * S + s0 + rl) * 32 + c * 8
                                                              : W_T + (size_t)(s0 + rl) * D + k);
                o[0] = f32x4{v[0] * inv, v[1] * inv, v[2] * inv, v[3] * inv};
                if (k + 4 < D) o[1] = f32x4{v[4] * inv, v[5] * inv, v[6] * inv, v[7] * inv};
            }
        }
    }
}

template <int MODE>
__global__ __launch_bounds__(1024) void split_wT_kernel(const float* __restrict__ W, int D, int S, int nks, float scale,
                                                        const float* __restrict__ scale_dev, _Float16* __restrict__ ws,
                                                        const float* __restrict__ mu, double* __restrict__ dot_part,
                                                        float* __restrict__ sq_part, float* __restrict__ W_T, int wt_slices) {
    constexpr int KS = MODE != 0 ? 32 : 16;
    __shared__ float tile[KS][257];
    __shared__ float mu_s[KS];
    split_wT_body<MODE>(blockIdx.x, gridDim.x, tile, mu_s, W, D, S, nks, scale, scale_dev, ws, mu, dot_part, sq_part, W_T, wt_slices);
}
// Both operand forms of one fp32 matrix M (R x C, row-major) for the split-fp16 contractions of the dense AuxK route, in ONE pass
// over M (MODE 0 images: hi | lo halves of 16-wide k-steps):
//   rf: M as a row operand     -- image rows = rows of M (blocks of 256), k = columns of M      (what split_rows_kernel<0> writes)
//   tf: M as a k-major operand -- image rows = COLUMNS of M (blocks of 256), k = rows of M      (what split_wT_kernel<0> writes)
// The codes, dL/dx_hat of the auxiliary term, x and the dead latents' decoder rows are each needed in both forms (the forward
// contracts over their columns, the weight gradients over their rows): ten image launches were six reads too many.  A workgroup
// holds a 256 x 64 tile of M * scale in LDS and writes the tile's four row-form images whole and its 64-row share of sixteen
// k-major images; same values, same rounding as the two kernels it replaces (bit-identical images).
__global__ __launch_bounds__(1024) void split_both_kernel(const float* __restrict__ M, int R, int C, float scale,
                                                          const float* __restrict__ scale_dev, _Float16* __restrict__ rf, int nks_r,
                                                          _Float16* __restrict__ tf, int nks_t) {
    __shared__ float tile[256][64];  // element (r, c) at column c ^ (r & 31): both read patterns below spread over the banks
    if (scale_dev != nullptr) scale *= *scale_dev;
    const int ctiles = (C + 63) / 64;
    const int rb = blockIdx.x / ctiles, ct = blockIdx.x % ctiles;
    const int r0 = rb * 256, c0 = ct * 64;
    for (int q = threadIdx.x; q < 256 * 16; q += 1024) {  // 16 bytes per lane (C % 4 == 0)
        const int rl = q >> 4, cl = (q & 15) * 4;
        const int r = r0 + rl, c = c0 + cl;
        f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
        if (r < R && c < C) v = *reinterpret_cast<const f32x4*>(M + (size_t)r * C + c) * scale;
        const int sw = rl & 31;
        tile[rl][cl ^ sw] = v[0]; tile[rl][(cl + 1) ^ sw] = v[1]; tile[rl][(cl + 2) ^ sw] = v[2]; tile[rl][(cl + 3) ^ sw] = v[3];
    }
    __syncthreads();
    const int i = threadIdx.x;
    if (rf != nullptr) {
        const int rl = i >> 2, p = i & 3;
        const int c = p ^ ((4 - ((rl >> 2) & 3)) & 3);
        const int part = c >> 1, h = c & 1;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int ks = c0 / 16 + j;
            if (ks >= nks_r) break;
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = tile[rl][(16 * j + 8 * h + e) ^ (rl & 31)];
            reinterpret_cast<half8*>(rf + ((size_t)rb * nks_r + ks) * 256 * 32)[i] = split8(v, part);
        }
    }
    if (tf != nullptr) {
        const int cb = c0 / 256, crow0 = c0 % 256;  // the k-major operand's row block and this tile's 64 rows inside it
#pragma unroll
        for (int qq = 0; qq < 4; ++qq) {
            const int idx = i + 1024 * qq;
            const int j = idx >> 8, within = idx & 255;
            const int cl = within >> 2, p = within & 3;
            const int rl_t = crow0 + cl;
            const int c = p ^ ((4 - ((rl_t >> 2) & 3)) & 3);
            const int part = c >> 1, h = c & 1;
            const int ks = r0 / 16 + j;
            if (ks >= nks_t) continue;
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = tile[16 * j + 8 * h + e][cl ^ ((16 * j + 8 * h + e) & 31)];
            reinterpret_cast<half8*>(tf + ((size_t)cb * nks_t + ks) * 256 * 32)[rl_t * 4 + p] = split8(v, part);
        }
    }
}

// The f16r step's two image passes in one launch (they depend on the same scales and on nothing of each other): workgroups
// [0, n_x) write the centred x images, the rest the W_enc^T images with everything else that pass produces.
struct SplitF16rArgs {
    const float* x; int n, D, nks; const float* scales; const float* mu; _Float16* xs;
    const float* W; int S; _Float16* ws; double* dot_part; float* sq_part; float* W_T;
    int n_x;
    float* xS; int wt_slices;  // the slice-major fp32 copy of x / W_T written slice-major (see the two bodies)
};
__global__ __launch_bounds__(1024) void split_f16r_kernel(SplitF16rArgs a) {
    __shared__ float tile[32][257];
    __shared__ float mu_s[32];
    if ((int)blockIdx.x < a.n_x) split_rows_body<2>(blockIdx.x, a.x, a.n, a.D, a.nks, 1.0f, a.scales, a.mu, a.xs, a.xS);
    else split_wT_body<2>(blockIdx.x - a.n_x, gridDim.x - a.n_x, tile, mu_s, a.W, a.D, a.S, a.nks, 1.0f, a.scales + 1, a.ws, a.mu, a.dot_part,
                          a.sq_part, a.W_T, a.wt_slices);
}

// ---- the streamed f16r step: everything derived from x in ONE pass (kernels.h: XprepArgs) ------------------------------------
// Centring vector, x scale and the square normaliser are those of the PREVIOUS batch (any fixed vector / power of two / positive
// number is valid for the margin; the batch's own would need a pass before this one).  A workgroup = one image (256 rows x 32
// columns), as split_rows_body<2>; on top of the image and the slice-major copy it leaves
//   * xn_part[ks][row] = {sum ((x - mu) / up)^2, sum (delta / up)^2} over the image's 32 columns (delta = the rounding error the
//     fp16 image of the row carries: center_stats_kernel's two norms, in 32-column pieces),
//   * col_part[blk][32 columns] = column sums of x over the image's rows (the NEXT centring vector),
//   * amax_part / cmax_part[image] = max |x|, max |x - mu| (the MSE's rescale; the NEXT x scale and this image's overflow check),
//   * with `rows`: the batch is gathered from a pool on the way in, and written out contiguously (x_out).
__device__ __forceinline__ float xprep_round_f16_sig(float v) {  // (= select.hip: round_f16_sig)
    const uint32_t b = __float_as_uint(v);
    return __uint_as_float((b + 0x0FFFu + ((b >> 13) & 1u)) & 0xFFFFE000u);
}
__global__ __launch_bounds__(1024) void xprep_kernel(XprepArgs a) {
    __shared__ float cs[256][33];
    __shared__ float cs2[32][33];
    __shared__ float shm[2][16];
    const int nks = a.nks;
    const int bid = blockIdx.x, blk = bid / nks, ks = bid % nks;
    const int i = threadIdx.x, lane = i & 63, w = i >> 6;
    const int rl = i >> 2, p = i & 3;
    const int c = p ^ ((4 - ((rl >> 2) & 3)) & 3);
    const int r = blk * 256 + rl, k = ks * 32 + c * 8;
    const float scale = a.scales[0];
    const float up = a.scales[4];
    const float back = (up > 0.f && up < 3.0e38f) ? up : 1.0f, inv = 1.0f / back;
    float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    float raw[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    float am = 0.f, cm = 0.f, s2 = 0.f, d2 = 0.f;
    if (r < a.n) {  // (D % 32 == 0 on this route: the chunk's eight values exist)
        const size_t src = a.rows != nullptr ? (size_t)a.rows[r] : (size_t)r;
        const f32x4 x0 = *reinterpret_cast<const f32x4*>(a.x + src * a.D + k);
        const f32x4 x1 = *reinterpret_cast<const f32x4*>(a.x + src * a.D + k + 4);
        if (a.x_out != nullptr) {
            f32x4* o = reinterpret_cast<f32x4*>(a.x_out + (size_t)r * a.D + k);
            o[0] = x0; o[1] = x1;
        }
        f32x4* o = reinterpret_cast<f32x4*>(a.xS + ((size_t)ks * a.n + r) * 32 + c * 8);
        o[0] = x0; o[1] = x1;
        const f32x4 m0 = *reinterpret_cast<const f32x4*>(a.mu + k), m1 = *reinterpret_cast<const f32x4*>(a.mu + k + 4);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            raw[e] = e < 4 ? x0[e] : x1[e - 4];
            const float ce = raw[e] - (e < 4 ? m0[e] : m1[e - 4]);
            am = fmaxf(am, fabsf(raw[e]));
            cm = fmaxf(cm, fabsf(ce));
            const float d = (ce - xprep_round_f16_sig(ce)) * inv, q = ce * inv;
            s2 = __builtin_fmaf(q, q, s2);
            d2 = __builtin_fmaf(d, d, d2);
            v[e] = ce * scale;
        }
        // (a non-finite element: NaN maxima would be dropped by fmaxf -- carry them as inf so that the step sees them)
        if (!(s2 < 3.0e38f)) { cm = __builtin_inff(); }
    }
    reinterpret_cast<half8*>(a.xs + (size_t)bid * 256 * 32)[i] = pack8<2>(v, 0);
    // the row's two norm pieces: its four chunks sit in four neighbouring lanes
    s2 += __shfl_xor(s2, 1, 64); s2 += __shfl_xor(s2, 2, 64);
    d2 += __shfl_xor(d2, 1, 64); d2 += __shfl_xor(d2, 2, 64);
    if (p == 0 && r < a.n) reinterpret_cast<float2*>(a.xn_part)[(size_t)ks * a.n_pad + r] = float2{s2, d2};
    // column sums over the image's 256 rows: through LDS, eight-row segments then the 32 segments, fixed order
#pragma unroll
    for (int e = 0; e < 8; ++e) cs[rl][c * 8 + e] = raw[e];
    am = wave_max(am); cm = wave_max(cm);
    if (lane == 0) { shm[0][w] = am; shm[1][w] = cm; }
    __syncthreads();
    {
        const int col = i & 31, seg = i >> 5;
        float t = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) t += cs[seg * 8 + j][col];
        cs2[seg][col] = t;
    }
    __syncthreads();
    if (i < 32) {
        float t = 0.f;
#pragma unroll
        for (int j = 0; j < 32; ++j) t += cs2[j][i];
        a.col_part[(size_t)blk * a.D + ks * 32 + i] = t;
    }
    if (i == 32) {
        float m0 = 0.f, m1 = 0.f;
        for (int j = 0; j < 16; ++j) { m0 = fmaxf(m0, shm[0][j]); m1 = fmaxf(m1, shm[1][j]); }
        a.amax_part[bid] = m0;
        a.cmax_part[bid] = m1;
    }
    if (a.mu_keep != nullptr && bid == 0 && i >= 192) {  // (the lender's snapshot: waves 3.. of workgroup 0)
        for (int d = i - 192; d < a.D; d += 1024 - 192) a.mu_keep[d] = a.mu[d];
        if (i == 192) { a.xside_keep[0] = scale; a.xside_keep[1] = 0.f; }
    }
    if (a.stale != nullptr && i >= 64 && i < 66) {  // the W_enc samples of XprepArgs::stale (a wave that has nothing else left to do)
        uint32_t h = ((uint32_t)bid * 3u + (uint32_t)(i - 64)) * 2654435761u + a.salt * 40503u;
        h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        const uint32_t d = h % (uint32_t)a.D, sidx = (h >> 8) % (uint32_t)a.S;
        const float w = a.W_enc[(size_t)d * a.S + sidx], c = a.WeS[((size_t)(d >> 5) * a.S + sidx) * 32 + (d & 31)];
        if (__float_as_uint(w) != __float_as_uint(c)) atomicOr(a.stale, 1);
    }
    if (a.stale != nullptr && i >= 128 && i < 192) {  // ... and ALL of b_enc, 64 elements per workgroup (another idle wave)
        bool differs = false;
        for (int sidx = bid * 64 + (i - 128); sidx < a.S; sidx += (int)gridDim.x * 64)
            differs |= __float_as_uint(a.b_enc[sidx]) != __float_as_uint(a.b_seen[sidx]);
        if (differs) atomicOr(a.stale, 1);
    }
}

// b_shift[s] = float(sum_ks dot_part[ks][s] / w_scale + b_enc[s]), ||W[:, s]|| = sqrt(sum_ks sq_part[ks][s]) / w_scale and
// ||dW[:, s]|| from the second plane of sq_part; per workgroup one maximum of |b_shift| (wg_max[0..nwg)), one of the norms
// (wg_max[nwg..2 nwg)) and one of the rounding-error norms (wg_max[2 nwg..3 nwg))
__global__ __launch_bounds__(256) void bias_finish_kernel(const double* __restrict__ dot_part,
                                                          const float* __restrict__ sq_part, int nks, int S, int S_pad,
                                                          const float* __restrict__ w_scale, const float* __restrict__ b_enc,
                                                          float* __restrict__ b_shift, float* __restrict__ wg_max,
                                                          float* __restrict__ b_seen) {
    __shared__ float sh[3][4];
    const int sidx = blockIdx.x * 256 + threadIdx.x;
    float out = 0.f, nrm = 0.f, dnrm = 0.f;
    if (sidx < S) {
        double acc = 0.0, sq = 0.0, dsq = 0.0;
        for (int ks = 0; ks < nks; ++ks) {
            acc += dot_part[(size_t)ks * S_pad + sidx];
            sq += (double)sq_part[(size_t)ks * S_pad + sidx];
            dsq += (double)sq_part[(size_t)(nks + ks) * S_pad + sidx];
        }
        const double sc = (double)(*w_scale);
        const float be = b_enc[sidx];
        out = (float)(acc / sc + (double)be);
        b_shift[sidx] = out;
        if (b_seen != nullptr) b_seen[sidx] = be;
        nrm = (float)(sqrt(sq) / sc) * 1.000001f;  // (rounded up: it bounds an error)
        dnrm = (float)(sqrt(dsq) / sc) * 1.00001f;  // (the shares are fp32 sums of 32 squares: rounded up a little further)
    }
    float m = fabsf(out);
    for (int o = 32; o > 0; o >>= 1) { m = fmaxf(m, __shfl_xor(m, o, 64)); nrm = fmaxf(nrm, __shfl_xor(nrm, o, 64)); dnrm = fmaxf(dnrm, __shfl_xor(dnrm, o, 64)); }
    if ((threadIdx.x & 63) == 0) { sh[0][threadIdx.x >> 6] = m; sh[1][threadIdx.x >> 6] = nrm; sh[2][threadIdx.x >> 6] = dnrm; }
    __syncthreads();
    if (threadIdx.x < 3) {
        const float* q = sh[threadIdx.x];
        wg_max[threadIdx.x * gridDim.x + blockIdx.x] = fmaxf(fmaxf(q[0], q[1]), fmaxf(q[2], q[3]));
    }
}


}  // namespace

hipError_t launch_split_rows(const float* x, int n, int D, int Dp, void* xs, int mode, hipStream_t stream, float scale,
                             const float* scale_dev, const float* mu, float* xS) {
    const int nks = Dp / (mode != 0 ? 32 : 16), nblk = (n + 255) / 256;
    if (nblk <= 0) return hipSuccess;
    _Float16* o = reinterpret_cast<_Float16*>(xs);
    if (mode == 1) hipLaunchKernelGGL(split_rows_kernel<1>, dim3(nblk * nks), dim3(1024), 0, stream, x, n, D, nks, scale, scale_dev, mu, o, nullptr);
    else if (mode == 2) hipLaunchKernelGGL(split_rows_kernel<2>, dim3(nblk * nks), dim3(1024), 0, stream, x, n, D, nks, scale, scale_dev, mu, o, xS);
    else hipLaunchKernelGGL(split_rows_kernel<0>, dim3(nblk * nks), dim3(1024), 0, stream, x, n, D, nks, scale, scale_dev, mu, o, nullptr);
    return hipGetLastError();
}

hipError_t launch_split_wT(const float* W, int D, int S, int S_pad, int Dp, float scale, void* ws, int mode,
                           hipStream_t stream, const float* scale_dev, const float* mu, double* dot_part, float* sq_part,
                           float* W_T, int wt_slices) {
    const int nks = Dp / (mode != 0 ? 32 : 16);
    const dim3 grid((S_pad / 256) * nks);
    _Float16* o = reinterpret_cast<_Float16*>(ws);
    if (mode == 1) hipLaunchKernelGGL(split_wT_kernel<1>, grid, dim3(1024), 0, stream, W, D, S, nks, scale, scale_dev, o, nullptr, nullptr, nullptr, nullptr, 0);
    else if (mode == 2) hipLaunchKernelGGL(split_wT_kernel<2>, grid, dim3(1024), 0, stream, W, D, S, nks, scale, scale_dev, o, mu, dot_part, sq_part, W_T, wt_slices);
    else hipLaunchKernelGGL(split_wT_kernel<0>, grid, dim3(1024), 0, stream, W, D, S, nks, scale, scale_dev, o, nullptr, nullptr, nullptr, nullptr, 0);
    return hipGetLastError();
}

hipError_t launch_split_both(const float* M, int R, int C, float scale, const float* scale_dev, void* rf, int kp_r, void* tf, int kp_t,
                             hipStream_t stream) {
    if (R <= 0 || C <= 0) return hipSuccess;
    const int grid = ((R + 255) / 256) * ((C + 63) / 64);
    hipLaunchKernelGGL(split_both_kernel, dim3(grid), dim3(1024), 0, stream, M, R, C, scale, scale_dev, reinterpret_cast<_Float16*>(rf), kp_r / 16,
                       reinterpret_cast<_Float16*>(tf), kp_t / 16);
    return hipGetLastError();
}

hipError_t launch_split_f16r(const float* x, int n, int D, int Dp, void* xs, const float* scales, const float* mu, const float* W,
                             int S, int S_pad, void* ws, double* dot_part, float* sq_part, float* W_T, hipStream_t stream, float* xS,
                             int wt_slices) {
    SplitF16rArgs a{};
    a.xS = xS; a.wt_slices = wt_slices;
    a.x = x; a.n = n; a.D = D; a.nks = Dp / 32; a.scales = scales; a.mu = mu; a.xs = reinterpret_cast<_Float16*>(xs);
    a.W = W; a.S = S; a.ws = reinterpret_cast<_Float16*>(ws); a.dot_part = dot_part; a.sq_part = sq_part; a.W_T = W_T;
    a.n_x = ((n + 255) / 256) * a.nks;
    hipLaunchKernelGGL(split_f16r_kernel, dim3(a.n_x + (S_pad / 256) * a.nks), dim3(1024), 0, stream, a);
    return hipGetLastError();
}

hipError_t launch_bias_finish(const double* dot_part, const float* sq_part, int Dp, int S, int S_pad, const float* w_scale,
                              const float* b_enc, float* b_shift, float* wg_part, hipStream_t stream, float* b_seen) {
    const int nwg = (S + 255) / 256;
    hipLaunchKernelGGL(bias_finish_kernel, dim3(nwg), dim3(256), 0, stream, dot_part, sq_part, Dp / 32, S, S_pad, w_scale,
                       b_enc, b_shift, wg_part, b_seen);
    return hipGetLastError();
}

hipError_t launch_xprep(const XprepArgs& a, hipStream_t stream) {
    const int nblk = (a.n + 255) / 256;
    if (nblk <= 0 || a.D % 32 != 0) return hipErrorInvalidValue;
    hipLaunchKernelGGL(xprep_kernel, dim3(nblk * a.nks), dim3(1024), 0, stream, a);
    return hipGetLastError();
}
