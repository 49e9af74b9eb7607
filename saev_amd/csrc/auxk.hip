// AuxK dead-latent loss (reference nn/modeling.py:75-103) as dense algebra over the *dead set*.
//
// With k_aux = 512 codes per row drawn from the few hundred..thousand dead latents, the auxiliary codes are
// ~25 % dense over the dead columns: index-gathering 4 KB rows per code (what the main k = 32 path does) would
// move 2 * B * k_aux * 4D bytes per pass (68 GB at config 2), while the same contractions as GEMMs over the
// compacted dead set are a few hundred GFLOP.  So, with dl = ascending list of dead latents (nd of them):
//
//   H  = x @ W_enc[:, dl] + b_enc[dl]                 (B x nd)   dense contraction
//   A  = H masked to the k_use = min(k_aux, nd) largest entries per row   (select.hip radix select)
//   E  = A @ W_dec[dl] + b_dec                        (B x D)    dense contraction
//   aux = alpha * mean((E - (x - x_hat))^2)            g_aux = d aux / dE
//   dA = (g_aux @ W_dec[dl]^T) * mask                 (B x nd)   dense contraction
//   dW_dec[dl] += A^T @ g_aux,  dW_enc^T[dl] += dA^T @ x         dense contractions over the batch axis
//   db_enc[dl] += colsum(dA),   db_dec += colsum(g_aux)
//
// The five products run on the split-fp16 MFMA kernel of gemm_encode_f16x3.hip with its dense epilogue (fp32-accurate:
// three f16 products per fp32 product; ctx.hip, dense_f16x3 / ksplit_f16x3), in every encoder mode -- no library GEMM is
// linked.  Everything around them (compaction, gathers, bias, masking, residual, scatter-add) is here.  The reference reads n_dead back every step (`int(dead_mask.sum().item())`, modeling.py:92).  Here the host only
// does so when a device-written bound says the dead set may be larger than AUX_SMALL_MAX (ctx.hip, saev_step_dead): the
// few-dead-latents kernels below take the count from the device and exit when there is nothing to do.
#include "common.h"
#include "kernels.h"

namespace {

// dead mask (S) -> ascending list of dead latents; single workgroup, every thread owns one contiguous chunk
__global__ __launch_bounds__(1024) void dead_compact_kernel(const int32_t* dead, int S, int32_t* list,
                                                            const int32_t* n_dead_dev) {
    if (n_dead_dev != nullptr && *n_dead_dev <= 0) return;
    __shared__ int wave_tot[16];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int per = (S + 1023) / 1024;
    const int i0 = tid * per, i1 = min(S, i0 + per);
    int cnt = 0;
    uint64_t bits = 0;  // the chunk's flags (per <= 64: every d_sae up to 65 536), fetched as 16-byte words, all in flight
    const bool packed = per <= 64 && (per & 3) == 0 && i0 + per <= S;
    if (packed) {
        typedef int i32x4_t __attribute__((ext_vector_type(4)));
        const i32x4_t* p4 = reinterpret_cast<const i32x4_t*>(dead + i0);
#pragma unroll 8
        for (int q = 0; q < (per >> 2); ++q) {
            const i32x4_t v = p4[q];
            bits |= (uint64_t)((v[0] ? 1 : 0) | (v[1] ? 2 : 0) | (v[2] ? 4 : 0) | (v[3] ? 8 : 0)) << (4 * q);
        }
        cnt = __popcll(bits);
    } else {
        for (int i = i0; i < i1; ++i) cnt += dead[i] ? 1 : 0;
    }
    int incl = cnt;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int n = __shfl_up(incl, o, 64);
        if (lane >= o) incl += n;
    }
    if (lane == 63) wave_tot[w] = incl;
    __syncthreads();
    int pos = incl - cnt;
    for (int j = 0; j < w; ++j) pos += wave_tot[j];
    if (cnt > 0) {
        if (packed) {
            while (bits != 0) {
                const int b = __ffsll((long long)bits) - 1;
                list[pos++] = i0 + b;
                bits &= bits - 1;
            }
        } else {
            for (int i = i0; i < i1; ++i)
                if (dead[i]) list[pos++] = i;
        }
    }
}

// Wenc_dead (D, ndp) = W_enc[:, dl] (zero columns beyond nd);  Wdec_dead (ndp, D) = W_dec[dl, :] (zero rows beyond nd)
__global__ __launch_bounds__(256) void gather_dead_kernel(const float* W_enc, const float* W_dec, const int32_t* dl,
                                                          int nd, int ndp, int D, int S, float* Wenc_dead,
                                                          float* Wdec_dead, const int32_t* nd_dev) {
    if (nd_dev != nullptr) nd = min(nd, *nd_dev);  // the host sized the launch by a bound; the list holds *nd_dev entries
    const long n1 = (long)D * ndp;
    const long n2 = (long)ndp * (D >> 2);
    for (long q = (long)blockIdx.x * 256 + threadIdx.x; q < n1 + n2; q += (long)gridDim.x * 256) {
        if (q < n1) {
            const int d = (int)(q / ndp), j = (int)(q % ndp);
            Wenc_dead[q] = (j < nd) ? W_enc[(size_t)d * S + dl[j]] : 0.f;
        } else {
            const long r = q - n1;
            const int j = (int)(r / (D >> 2)), c = (int)(r % (D >> 2));
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (j < nd) v = reinterpret_cast<const f32x4*>(W_dec + (size_t)dl[j] * D)[c];
            reinterpret_cast<f32x4*>(Wdec_dead + (size_t)j * D)[c] = v;
        }
    }
}

// compact bias of the dead set for the fused dense contraction: out[j] = b_enc[dl[j]], -inf on the padding columns
__global__ void dead_bias_vec_kernel(const float* b_enc, const int32_t* dl, int nd, int ndp, float* out, float pad,
                                     const int32_t* nd_dev) {
    if (nd_dev != nullptr) nd = min(nd, *nd_dev);
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j < ndp) out[j] = (j < nd) ? b_enc[dl[j]] : pad;
}

// ---- max |.| of a matrix as a by-product of the kernel that writes it, and the power-of-two operand scale that follows from it ----
// Every workgroup of the producing grid stores its maximum (part[blockIdx.x]: a plain store -- one returning device-scope atomic
// per workgroup on a shared counter, the "last workgroup finishes" pattern, made aux_resid_kernel 42 -> 316 us: the XCDs' L2s
// are not coherent with one another and each such atomic is a round trip to the fabric behind a write-back); pow2_parts_kernel,
// one small workgroup, turns the maxima into {2^e, 1} with 2^e * absmax in [2^13, 2^14) (what pow2_scale_kernel makes of
// absmax_kernel's word).
__device__ __forceinline__ float pow2_operand_scale(float m) { return (m > 0.f && m < 3.0e38f) ? exp2f(13.0f - floorf(log2f(m))) : 1.0f; }
__global__ __launch_bounds__(256) void pow2_parts_kernel(const float* part, int n, float* pair) {
    __shared__ float sh[4];
    float m = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) m = fmaxf(m, part[i]);
    m = wave_max(m);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        pair[0] = pow2_operand_scale(fmaxf(fmaxf(sh[0], sh[1]), fmaxf(sh[2], sh[3])));
        pair[1] = 1.0f;
    }
}
// the workgroup's maximum from its (up to four) waves' values; returns it in thread 0
__device__ __forceinline__ float block_max4(float m, float* sh) {
    m = wave_max(m);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = m;
    __syncthreads();
    return fmaxf(fmaxf(sh[0], sh[1]), fmaxf(sh[2], sh[3]));
}

// max |x| over n floats (n % 4 == 0) per workgroup (no word to zero, no atomics)
__global__ __launch_bounds__(256) void absmax_parts_kernel(const float* x, long n4, float* part) {
    __shared__ float sh[4];
    float m = 0.f;
    for (long q = (long)blockIdx.x * 256 + threadIdx.x; q < n4; q += (long)gridDim.x * 256) {
        const f32x4 v = reinterpret_cast<const f32x4*>(x)[q];
        m = fmaxf(fmaxf(fmaxf(m, fabsf(v[0])), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3])));
    }
    m = block_max4(m, sh);
    if (threadIdx.x == 0) part[blockIdx.x] = m;
}

// The AuxK selection as the dense code matrix it is used as: A[b][j] = H[b][j] where j is one of the row's k largest
// pre-activations over the dead set (ties on the k-th value: lowest columns first, as select_dense_kernel), 0 elsewhere;
// mask[b][j] = selected; the workgroup's max |A| (part[blockIdx.x], for pow2_parts_kernel).  One launch instead of
// select_dense (four passes over the row through an LDS histogram) + two fills + aux_scatter + absmax.
// One wave per row, the row's keys in registers (NV float4 per lane: ndp <= 256 NV), the k-th largest key by a bit-wise
// search that stops as soon as exactly k keys pass (usually after the exponent and a few mantissa bits).
template <int NV>
__global__ __launch_bounds__(256) void aux_select_kernel(const float* __restrict__ H, int n_rows, int ndp, int k_host, const int32_t* k_dev,
                                                         float* __restrict__ A, uint8_t* __restrict__ mask, float* __restrict__ part) {
    __shared__ float sh[4];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int k = k_dev != nullptr ? min(*k_dev, k_host) : k_host;
    const int q4 = ndp >> 2;
    float m = 0.f;
    for (int row = blockIdx.x * 4 + w; row < n_rows; row += gridDim.x * 4) {
        const f32x4* hr = reinterpret_cast<const f32x4*>(H + (size_t)row * ndp);
        uint32_t key[NV][4];  // 0 = no column (below every key a real value has, NaNs with the sign bit aside)
#pragma unroll
        for (int n = 0; n < NV; ++n) {
            const int q = lane + 64 * n;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (q < q4) v = hr[q];
#pragma unroll
            for (int e = 0; e < 4; ++e) key[n][e] = q < q4 ? f2ukey(v[e]) : 0u;
        }
        uint32_t T = 0u;
        int cnt = 256 * NV;
        if (k > 0) {
#pragma unroll 1
            for (int bit = 31; bit >= 0 && cnt != k; --bit) {
                const uint32_t cand = T | (1u << bit);
                int c = 0;
#pragma unroll
                for (int n = 0; n < NV; ++n)
#pragma unroll
                    for (int e = 0; e < 4; ++e) c += key[n][e] >= cand ? 1 : 0;
                c = wave_sum_i(c);
                if (c >= k) { T = cand; cnt = c; }
            }
        }
        // cnt == k: the keys >= T are the selection.  Otherwise T is the k-th largest key itself and more than one column holds it:
        // the keys > T and the first `need` columns, in ascending order, of those that equal it
        int need = 0, eq_before[NV];
#pragma unroll
        for (int n = 0; n < NV; ++n) eq_before[n] = 0;
        const bool ties = k > 0 && cnt != k;
        if (ties) {
            int gt = 0;
#pragma unroll
            for (int n = 0; n < NV; ++n)
#pragma unroll
                for (int e = 0; e < 4; ++e) gt += key[n][e] > T ? 1 : 0;
            need = k - wave_sum_i(gt);
            int base = 0;
#pragma unroll
            for (int n = 0; n < NV; ++n) {
                int eq = 0;
#pragma unroll
                for (int e = 0; e < 4; ++e) eq += key[n][e] == T ? 1 : 0;
                int incl = eq;
#pragma unroll
                for (int o = 1; o < 64; o <<= 1) {
                    const int up = __shfl_up(incl, o, 64);
                    if (lane >= o) incl += up;
                }
                eq_before[n] = base + incl - eq;
                base += __shfl(incl, 63, 64);
            }
        }
        f32x4* ar = reinterpret_cast<f32x4*>(A + (size_t)row * ndp);
        uint32_t* mr = reinterpret_cast<uint32_t*>(mask + (size_t)row * ndp);
#pragma unroll
        for (int n = 0; n < NV; ++n) {
            const int q = lane + 64 * n;
            if (q >= q4) continue;
            f32x4 o;
            uint32_t mw = 0u;
            int eqs = eq_before[n];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                bool sel = k > 0 && (ties ? key[n][e] > T : key[n][e] >= T);
                if (ties && key[n][e] == T) { sel = eqs < need; ++eqs; }
                o[e] = sel ? ukey2f(key[n][e]) : 0.f;
                mw |= sel ? (1u << (8 * e)) : 0u;
                m = fmaxf(m, fabsf(o[e]));
            }
            ar[q] = o;
            mr[q] = mw;
        }
    }
    m = block_max4(m, sh);
    if (threadIdx.x == 0) part[blockIdx.x] = m;
}

// A[b][idx] = val, mask[b][idx] = 1 for the selected codes (A and mask are zeroed by the caller)
__global__ __launch_bounds__(256) void aux_scatter_kernel(const int32_t* idx, const float* val, long n_rows, int k,
                                                          int stride, int ndp, float* A, uint8_t* mask, const int32_t* k_dev) {
    if (k_dev != nullptr) k = min(k, *k_dev);  // (0: nothing was selected, idx / val are stale)
    const long total = n_rows * k;
    for (long p = (long)blockIdx.x * 256 + threadIdx.x; p < total; p += (long)gridDim.x * 256) {
        const long b = p / k;
        const int j = (int)(p % k);
        const int32_t i = idx[b * stride + j];
        if (i >= 0 && i < ndp) {
            A[b * ndp + i] = val[b * stride + j];
            mask[b * ndp + i] = 1;
        }
    }
}

// E holds A @ W_dec[dl]; in place: g_aux = gscale * (E + b_dec - (x - x_hat)); per-row sum of squared differences
template <int NV>
__global__ __launch_bounds__(256) void aux_resid_kernel(float* E, const float* x, const float* x_hat, const float* b_dec,
                                                        int n_rows, int D, float gscale, RowStats* rowstats,
                                                        const int32_t* nd_dev, float* part) {
    __shared__ float sh[4];
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int D4 = D >> 2;
    float gmax = 0.f;  // max |g_aux| of the rows of this workgroup (part[blockIdx.x]: for pow2_parts_kernel, the scale of g_aux's fp16 split)
    if (row < n_rows) {
    f32x4* er = reinterpret_cast<f32x4*>(E + (size_t)row * D);
    if (nd_dev != nullptr && *nd_dev <= 0) {  // sized by a bound, but nothing is dead: the auxiliary term is exactly zero (modeling.py:92-94)
#pragma unroll
        for (int n = 0; n < NV; ++n)
            if (lane + 64 * n < D4) er[lane + 64 * n] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (lane == 0) rowstats[row].aux_sse = 0.f;
    } else {
    const f32x4* xr = reinterpret_cast<const f32x4*>(x + (size_t)row * D);
    const f32x4* hr = reinterpret_cast<const f32x4*>(x_hat + (size_t)row * D);
    const f32x4* br = reinterpret_cast<const f32x4*>(b_dec);
    float sse = 0.f;
#pragma unroll
    for (int n = 0; n < NV; ++n) {
        const int q = lane + 64 * n;
        if (q < D4) {
            const f32x4 e = er[q], xv = xr[q], hv = hr[q], bv = br[q];
            f32x4 g;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float diff = (e[c] + bv[c]) - (xv[c] - hv[c]);
                sse += diff * diff;
                g[c] = gscale * diff;
                gmax = fmaxf(gmax, fabsf(g[c]));
            }
            er[q] = g;
        }
    }
    sse = wave_sum(sse);
    if (lane == 0) rowstats[row].aux_sse = sse;
    }
    }
    if (part != nullptr) {
        gmax = block_max4(gmax, sh);
        if (threadIdx.x == 0) part[blockIdx.x] = gmax;
    }
}

__global__ __launch_bounds__(256) void mask_apply_kernel(float* dA, const uint8_t* mask, long n) {
    for (long q = (long)blockIdx.x * 256 + threadIdx.x; q < n; q += (long)gridDim.x * 256)
        if (!mask[q]) dA[q] = 0.f;
}
// ... four elements per thread, with the workgroup's max |dA| of the masked matrix (part[blockIdx.x])
__global__ __launch_bounds__(256) void mask_apply_absmax_kernel(float* dA, const uint8_t* mask, long n4, float* part) {
    __shared__ float sh[4];
    float m = 0.f;
    for (long q = (long)blockIdx.x * 256 + threadIdx.x; q < n4; q += (long)gridDim.x * 256) {
        f32x4 v = reinterpret_cast<f32x4*>(dA)[q];
        const uint32_t mw = reinterpret_cast<const uint32_t*>(mask)[q];
        if (mw != 0x01010101u) {
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (((mw >> (8 * e)) & 0xFFu) == 0u) v[e] = 0.f;
            reinterpret_cast<f32x4*>(dA)[q] = v;
        }
        m = fmaxf(fmaxf(fmaxf(m, fabsf(v[0])), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3])));
    }
    m = block_max4(m, sh);
    if (threadIdx.x == 0) part[blockIdx.x] = m;
}

// ---- a handful of dead latents (nd <= AUX_SMALL_MAX = 24, the measured break-even, and nd <= k_aux: every dead latent is "selected") ---------------
// Training runs spend most of their life with zero to a few dozen dead latents.  The dense path below pads the dead
// set to 256 columns and runs five MFMA contractions plus their operand splits for it (+0.7 ms per step at nd = 8);
// with so few columns the whole forward is one pass over x and x_hat per row, and the weight gradients one more.
//
// lane l ends up with the wave-wide sum of p[(l >> 3) & 7]
__device__ __forceinline__ float aux_reduce_scatter8(float (&p)[8], int lane) {
    int bit = 32;
#pragma unroll
    for (int h = 4; h >= 1; h >>= 1, bit >>= 1) {
        const bool up = (lane & bit) != 0;
#pragma unroll
        for (int i = 0; i < h; ++i) {
            const float keep = up ? p[i + h] : p[i];
            const float send = up ? p[i] : p[i + h];
            p[i] = keep + __shfl_xor(send, bit, 64);
        }
    }
    float r = p[0];
    for (; bit >= 1; bit >>= 1) r += __shfl_xor(r, bit, 64);
    return r;
}

// WencT_dead (AUX_SMALL_MAX, D) = W_enc[:, dl]^T and Wdec_dead (AUX_SMALL_MAX, D) = W_dec[dl] (zero rows beyond nd)
__global__ __launch_bounds__(256) void gather_dead_small_kernel(const float* W_enc, const float* W_dec, const int32_t* dl,
                                                                const int32_t* nd_dev, int D, int S, float* WencT_dead,
                                                                float* Wdec_dead, int cap) {
    const int nd = *nd_dev;
    if (nd <= 0 || nd > cap) return;
    const long total = (long)nd * D;  // rows past nd are never read (every consumer loops to the device-side count)
    for (long q = (long)blockIdx.x * 256 + threadIdx.x; q < total; q += (long)gridDim.x * 256) {
        const int j = (int)(q / D), d = (int)(q % D);
        const int i = dl[j];
        WencT_dead[q] = W_enc[(size_t)d * S + i];
        Wdec_dead[q] = W_dec[(size_t)i * D + d];
    }
}

// One wave per pair of rows (every dead-latent row of W_enc^T / W_dec that is fetched serves both): H = x W_enc[:, dl] +
// b_enc[dl] (= the auxiliary codes A, all dead latents being selected), E = A W_dec[dl] + b_dec, diff = E - (x - x_hat),
// g_aux = gscale * diff, dA = g_aux W_dec[dl]^T, row loss sum diff^2.
template <int NV>
__global__ __launch_bounds__(256) void aux_small_fwd_kernel(const float* x, const float* x_hat, const float* WencT_dead,
                                                            const float* Wdec_dead, const float* b_enc, const float* b_dec,
                                                            const int32_t* dl, int n_rows, int D, const int32_t* nd_dev,
                                                            float gscale, float* A, float* dA, float* g_aux,
                                                            RowStats* rowstats) {
    const int nd = *nd_dev;
    if (nd <= 0 || nd > AUX_SMALL_MAX) return;
    constexpr int ndp = AUX_SMALL_MAX;
    const int lane = threadIdx.x & 63;
    const int row0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * 2;
    if (row0 >= n_rows) return;
    const bool has1 = row0 + 1 < n_rows;
    const int D4 = D >> 2;
    f32x4 xv[2][NV], ev[2][NV];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const f32x4* xr = reinterpret_cast<const f32x4*>(x + (size_t)min(row0 + u, n_rows - 1) * D);
        const f32x4* br = reinterpret_cast<const f32x4*>(b_dec);
#pragma unroll
        for (int n = 0; n < NV; ++n) {
            const int q = lane + 64 * n;
            xv[u][n] = q < D4 ? xr[q] : f32x4{0.f, 0.f, 0.f, 0.f};
            ev[u][n] = q < D4 ? br[q] : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    }
    // codes: lane j (< nd <= 24) keeps h_j of row u in my_h[u]
    float my_h[2] = {0.f, 0.f};
    for (int j0 = 0; j0 < nd; j0 += 4) {  // four latents x two rows per reduce-scatter
        float p[8];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            p[t] = 0.f; p[4 + t] = 0.f;
            if (j0 + t >= nd) continue;
            const f32x4* wr = reinterpret_cast<const f32x4*>(WencT_dead + (size_t)(j0 + t) * D);
#pragma unroll
            for (int n = 0; n < NV; ++n) {
                const int q = lane + 64 * n;
                if (q < D4) {
                    const f32x4 w = wr[q];
                    p[t] += xv[0][n][0] * w[0] + xv[0][n][1] * w[1] + xv[0][n][2] * w[2] + xv[0][n][3] * w[3];
                    p[4 + t] += xv[1][n][0] * w[0] + xv[1][n][1] * w[1] + xv[1][n][2] * w[2] + xv[1][n][3] * w[3];
                }
            }
        }
        const float r = aux_reduce_scatter8(p, lane);  // lane l holds slot (l >> 3) & 7
        const int t = lane - j0;
        const float m0 = __shfl(r, (t & 3) << 3, 64), m1 = __shfl(r, (4 + (t & 3)) << 3, 64);
        if (t >= 0 && t < 4 && lane < nd) {
            const float bj = b_enc[dl[lane]];
            my_h[0] = m0 + bj; my_h[1] = m1 + bj;
        }
    }
    if (lane < ndp) {
        A[(size_t)row0 * ndp + lane] = lane < nd ? my_h[0] : 0.f;
        if (has1) A[(size_t)(row0 + 1) * ndp + lane] = lane < nd ? my_h[1] : 0.f;
    }
    // reconstruction of the codes
    for (int j = 0; j < nd; ++j) {
        const float h0 = __shfl(my_h[0], j, 64), h1 = __shfl(my_h[1], j, 64);
        const f32x4* wr = reinterpret_cast<const f32x4*>(Wdec_dead + (size_t)j * D);
#pragma unroll
        for (int n = 0; n < NV; ++n) {
            const int q = lane + 64 * n;
            if (q < D4) {
                const f32x4 w = wr[q];
                ev[0][n] += h0 * w;
                ev[1][n] += h1 * w;
            }
        }
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int row = min(row0 + u, n_rows - 1);
        const bool live = (u == 0) || has1;
        float sse = 0.f;
        const f32x4* hr = reinterpret_cast<const f32x4*>(x_hat + (size_t)row * D);
        f32x4* gr = reinterpret_cast<f32x4*>(g_aux + (size_t)row * D);
#pragma unroll
        for (int n = 0; n < NV; ++n) {
            const int q = lane + 64 * n;
            if (q < D4) {
                const f32x4 hv = hr[q];
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const float diff = ev[u][n][c] - (xv[u][n][c] - hv[c]);
                    sse += diff * diff;
                    ev[u][n][c] = gscale * diff;
                }
                if (live) gr[q] = ev[u][n];
            } else {
                ev[u][n] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
        sse = wave_sum(sse);
        if (lane == 0 && live) rowstats[row].aux_sse = sse;
    }
    // dA_j = <g_aux, W_dec[dl_j]>
    float my_d[2] = {0.f, 0.f};
    for (int j0 = 0; j0 < nd; j0 += 4) {
        float p[8];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            p[t] = 0.f; p[4 + t] = 0.f;
            if (j0 + t >= nd) continue;
            const f32x4* wr = reinterpret_cast<const f32x4*>(Wdec_dead + (size_t)(j0 + t) * D);
#pragma unroll
            for (int n = 0; n < NV; ++n) {
                const int q = lane + 64 * n;
                if (q < D4) {
                    const f32x4 w = wr[q];
                    p[t] += ev[0][n][0] * w[0] + ev[0][n][1] * w[1] + ev[0][n][2] * w[2] + ev[0][n][3] * w[3];
                    p[4 + t] += ev[1][n][0] * w[0] + ev[1][n][1] * w[1] + ev[1][n][2] * w[2] + ev[1][n][3] * w[3];
                }
            }
        }
        const float r = aux_reduce_scatter8(p, lane);
        const int t = lane - j0;
        const float m0 = __shfl(r, (t & 3) << 3, 64), m1 = __shfl(r, (4 + (t & 3)) << 3, 64);
        if (t >= 0 && t < 4) { my_d[0] = m0; my_d[1] = m1; }
    }
    if (lane < ndp) {
        dA[(size_t)row0 * ndp + lane] = lane < nd ? my_d[0] : 0.f;
        if (has1) dA[(size_t)(row0 + 1) * ndp + lane] = lane < nd ? my_d[1] : 0.f;
    }
}

// The same forward with the dead latents' weight rows staged through LDS (the shipped form; the kernel above stays as the
// fallback for shapes whose rows do not fit).  aux_small_fwd_kernel fetches every row of W_enc^T[dl] / W_dec[dl] from L2 once
// per PAIR of activation rows, three times over (codes, reconstruction, dA): 12 KB of loads per dead latent and row pair,
// 2.9 GB per launch at 30 dead latents -- the kernel was bound by the vector-memory path (0.36 ms at 30, growing linearly).
// Here a workgroup (four waves x RW rows) copies a chunk of LC weight rows into LDS once and all its rows use it: L2 traffic
// drops by the rows per workgroup, the per-latent loads become conflict-free ds_read_b128, and the kernel is bound by its fmas.
// Codes and dA pass from the reduce-scatter's owner lanes to "lane j holds latent j" through a per-wave LDS array.
template <int NV, int RW>
__global__ __launch_bounds__(256) void aux_small_fwd_lds_kernel(const float* x, const float* x_hat, const float* WencT_dead,
                                                                const float* Wdec_dead, const float* b_enc, const float* b_dec,
                                                                const int32_t* dl, int n_rows, int D, const int32_t* nd_dev,
                                                                float gscale, float* A, float* dA, float* g_aux,
                                                                RowStats* rowstats, int LC) {
    const int nd = *nd_dev;
    if (nd <= 0 || nd > AUX_SMALL_MAX) return;  // (uniform over the grid: no barrier is skipped by a part of a workgroup)
    constexpr int ndp = AUX_SMALL_MAX;
    constexpr int LPB = 8 / RW;  // latents per reduce-scatter batch (RW rows each)
    extern __shared__ float aux_smem[];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int D4 = D >> 2;
    f32x4* const Wl = reinterpret_cast<f32x4*>(aux_smem);                 // [LC][D4]
    float* const Hs = aux_smem + (size_t)LC * D + (size_t)w * 2 * RW * 64;  // per wave: codes [RW][64], then dA [RW][64]
    float* const Ds = Hs + RW * 64;
    const int row0 = (blockIdx.x * 4 + w) * RW;
    const int n_chunks = (nd + LC - 1) / LC;
    f32x4 v[RW][NV];  // x rows in phase 1, reconstruction / g_aux afterwards
#pragma unroll
    for (int r = 0; r < RW; ++r) {
        const f32x4* xr = reinterpret_cast<const f32x4*>(x + (size_t)min(row0 + r, n_rows - 1) * D);
#pragma unroll
        for (int n = 0; n < NV; ++n) v[r][n] = (lane + 64 * n < D4) ? xr[lane + 64 * n] : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    auto stage = [&](const float* W, int c) {  // chunk c of a (nd, D) row-major matrix -> LDS (all 256 threads)
        __syncthreads();  // the previous chunk has been consumed
        const int cn = min(LC, nd - c * LC);
        const f32x4* src = reinterpret_cast<const f32x4*>(W + (size_t)c * LC * D);
        for (int q = threadIdx.x; q < cn * D4; q += 256) Wl[q] = src[q];
        __syncthreads();
        return cn;
    };
    // <v[r], chunk row jl> for the wave's RW rows, LPB latents per reduce-scatter; owner lanes (lane & 7 == 0) leave the sums in
    // out[r * 64 + latent] (+ bias)
    auto dots = [&](int c, int cn, float* out, const float* bias) {
        for (int j0 = 0; j0 < cn; j0 += LPB) {
            float p[8];
#pragma unroll
            for (int t = 0; t < LPB; ++t) {
                const bool ok = j0 + t < cn;
                f32x4 wv[NV];
#pragma unroll
                for (int n = 0; n < NV; ++n)
                    wv[n] = (ok && lane + 64 * n < D4) ? Wl[(size_t)(j0 + t) * D4 + lane + 64 * n] : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int r = 0; r < RW; ++r) {
                    float acc = 0.f;
#pragma unroll
                    for (int n = 0; n < NV; ++n)
                        acc += v[r][n][0] * wv[n][0] + v[r][n][1] * wv[n][1] + v[r][n][2] * wv[n][2] + v[r][n][3] * wv[n][3];
                    p[t * RW + r] = acc;
                }
            }
            const float sum = aux_reduce_scatter8(p, lane);  // lane l holds slot (l >> 3) & 7
            if ((lane & 7) == 0) {
                const int slot = lane >> 3, t = slot / RW, r = slot % RW;
                const int j = c * LC + j0 + t;
                if (j0 + t < cn) out[r * 64 + j] = sum + (bias != nullptr ? bias[dl[j]] : 0.f);
            }
        }
    };
    // ---- codes H = x W_enc[:, dl] + b_enc[dl] ----
    for (int c = 0; c < n_chunks; ++c) {
        const int cn = stage(WencT_dead, c);
        dots(c, cn, Hs, b_enc);
    }
    // ---- reconstruction E = H W_dec[dl] + b_dec ----
#pragma unroll
    for (int r = 0; r < RW; ++r)
#pragma unroll
        for (int n = 0; n < NV; ++n)
            v[r][n] = (lane + 64 * n < D4) ? reinterpret_cast<const f32x4*>(b_dec)[lane + 64 * n] : f32x4{0.f, 0.f, 0.f, 0.f};
    for (int c = 0; c < n_chunks; ++c) {
        const int cn = stage(Wdec_dead, c);  // (its barriers also make the owner lanes' codes visible to the whole wave)
        for (int jl = 0; jl < cn; ++jl) {
            f32x4 wv[NV];
#pragma unroll
            for (int n = 0; n < NV; ++n) wv[n] = (lane + 64 * n < D4) ? Wl[(size_t)jl * D4 + lane + 64 * n] : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int r = 0; r < RW; ++r) {
                const float h = Hs[r * 64 + c * LC + jl];  // (one address per wave: broadcast)
#pragma unroll
                for (int n = 0; n < NV; ++n) v[r][n] += h * wv[n];
            }
        }
    }
    // ---- residual, loss, g_aux; the codes go out as A ----
#pragma unroll
    for (int r = 0; r < RW; ++r) {
        const int row = min(row0 + r, n_rows - 1);
        const bool live = row0 + r < n_rows;
        float sse = 0.f;
        const f32x4* xr = reinterpret_cast<const f32x4*>(x + (size_t)row * D);
        const f32x4* hr = reinterpret_cast<const f32x4*>(x_hat + (size_t)row * D);
        f32x4* gr = reinterpret_cast<f32x4*>(g_aux + (size_t)row * D);
#pragma unroll
        for (int n = 0; n < NV; ++n) {
            const int q = lane + 64 * n;
            if (q < D4) {
                const f32x4 xv = xr[q], hv = hr[q];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float diff = v[r][n][e] - (xv[e] - hv[e]);
                    sse += diff * diff;
                    v[r][n][e] = gscale * diff;
                }
                if (live) gr[q] = v[r][n];
            } else {
                v[r][n] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
        sse = wave_sum(sse);
        if (lane == 0 && live) rowstats[row].aux_sse = sse;
        if (live && lane < ndp) A[(size_t)row * ndp + lane] = lane < nd ? Hs[r * 64 + lane] : 0.f;
    }
    // ---- dA = g_aux W_dec[dl]^T (a single chunk is still in LDS) ----
    for (int c = 0; c < n_chunks; ++c) {
        const int cn = n_chunks == 1 ? nd : stage(Wdec_dead, c);
        dots(c, cn, Ds, nullptr);
    }
    __syncthreads();  // the owner lanes' dA visible to the wave
#pragma unroll
    for (int r = 0; r < RW; ++r)
        if (row0 + r < n_rows && lane < ndp) dA[(size_t)(row0 + r) * ndp + lane] = lane < nd ? Ds[r * 64 + lane] : 0.f;
}

// ---- at most AUX_FUSED_MAX dead latents: forward AND weight-gradient partials in ONE pass over x and x_hat ---------------------
// The steady state of a healthy run: a handful of dead latents on nearly every step.  The kernels above cost ~0.2 ms whatever
// the count -- five passes over (rows x d_model) matrices: x and x_hat read, g_aux written, g_aux and x read again by the weight
// gradients, two column sums -- for a few MFLOP.  Everything the auxiliary term needs of a row is local to the row, so here a
// workgroup (D / 256 waves, one float4 of the row per lane as in decode_q_kernel) keeps the dead latents' encoder and decoder
// rows AND the weight-gradient accumulators of its block of rows in registers (16 registers per dead latent), walks its rows two
// at a time -- codes H = x W_enc[:, dl] + b_enc[dl], E = H W_dec[dl] + b_dec, diff = E - (x - x_hat), g = gscale diff,
// dA = g W_dec[dl]^T (the two sets of dot products: reduce-scatter per wave, then four floats per value through LDS, two barriers
// per pair of rows) -- and leaves per-block partials of dWd, dWe, db_dec and db_enc[dl]; launch_aux_small_wsum / launch_colsum add
// the blocks in order.  g_aux, A and dA are never written.
template <int H, int BIT>
__device__ __forceinline__ void aux_rs_step(float* p, int lane) {
    if constexpr (H >= 1) {
        const bool up = (lane & BIT) != 0;
#pragma unroll
        for (int i = 0; i < H; ++i) {
            const float keep = up ? p[i + H] : p[i];
            const float send = up ? p[i] : p[i + H];
            p[i] = keep + __shfl_xor(send, BIT, 64);
        }
        aux_rs_step<H / 2, BIT / 2>(p, lane);
    }
}
// 16 values: lane l ends up with the wave-wide sum of p[(l >> 2) & 15]
__device__ __forceinline__ float aux_reduce_scatter16(float (&p)[16], int lane) {
    aux_rs_step<8, 32>(p, lane);
    float r = p[0];
    r += __shfl_xor(r, 2, 64);
    r += __shfl_xor(r, 1, 64);
    return r;
}

// ND = dead latents a workgroup provides for (the host picks it from its bound of the count: 4 or AUX_FUSED_MAX), RU = 16 / ND
// activation rows per trip -- sixteen dot products per reduce-scatter either way, half the barriers per row at ND = 4.
// PF trips of rows are in flight per lane (a ring of register sets, the loop unrolled over it).  The kernel is bound by its two
// barriers and reduce-scatters per trip, not by the latency of its loads: four rows per trip took it from 61.5 to 43.6 us at three
// dead latents, a deeper ring made it slower (kernels.h: AUX_FUSED_PF4).
template <int NW, int ND, int PF>
__global__ __launch_bounds__(64 * NW, 2) void aux_small_fused_kernel(const float* __restrict__ x, const float* __restrict__ x_hat,
                                                                  const float* __restrict__ WencT_dead, const float* __restrict__ Wdec_dead,
                                                                  const float* __restrict__ b_enc, const float* __restrict__ b_dec,
                                                                  const int32_t* __restrict__ dl, int n_rows, const int32_t* nd_dev, float gscale,
                                                                  int rows_per_wg, float* __restrict__ part, float* __restrict__ partb,
                                                                  float* __restrict__ partbe, RowStats* __restrict__ rowstats) {
    constexpr int D4 = 64 * NW, NVAL = 16, RU = NVAL / ND, NDO = AUX_FUSED_MAX;  // (NDO: rows per half of a block partial)
    static_assert(ND * RU == NVAL && ND <= NDO, "aux_small_fused_kernel: ND must divide 16");
    const int nd = *nd_dev;
    if (nd <= 0 || nd > ND) return;  // (uniform over the grid; the host's bound of the count chose ND)
    __shared__ __attribute__((aligned(16))) float shA[NVAL][4];
    __shared__ __attribute__((aligned(16))) float shB[NVAL + RU][4];
    extern __shared__ __attribute__((aligned(16))) float aux_smem[];  // the dead latents' encoder and decoder rows: [2][ND][D4] float4
    f32x4 (*const We)[D4] = reinterpret_cast<f32x4 (*)[D4]>(aux_smem);
    f32x4 (*const Wd)[D4] = We + ND;
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int q = threadIdx.x;
    f32x4 accD[ND], accE[ND];
    float be[ND], accbe[ND];
#pragma unroll
    for (int j = 0; j < ND; ++j) {
        const bool ok = j < nd;
        We[j][q] = ok ? reinterpret_cast<const f32x4*>(WencT_dead)[(size_t)j * D4 + q] : f32x4{0.f, 0.f, 0.f, 0.f};  // (only this lane reads its column group back)
        Wd[j][q] = ok ? reinterpret_cast<const f32x4*>(Wdec_dead)[(size_t)j * D4 + q] : f32x4{0.f, 0.f, 0.f, 0.f};  // (only this lane reads it back)
        be[j] = ok ? b_enc[dl[j]] : 0.f;
        accD[j] = f32x4{0.f, 0.f, 0.f, 0.f};
        accE[j] = f32x4{0.f, 0.f, 0.f, 0.f};
        accbe[j] = 0.f;
    }
    const f32x4 bd4 = reinterpret_cast<const f32x4*>(b_dec)[q];
    f32x4 accb = {0.f, 0.f, 0.f, 0.f};
    if (NW < 4 && threadIdx.x < NVAL + RU) {  // columns of the exchange arrays that no wave writes
        for (int v = NW; v < 4; ++v) { if (threadIdx.x < NVAL) shA[threadIdx.x][v] = 0.f; shB[threadIdx.x][v] = 0.f; }
    }
    const int r0 = blockIdx.x * rows_per_wg, r1 = min(n_rows, r0 + rows_per_wg);
    auto load_rows = [&](int r, f32x4 (&xv)[RU], f32x4 (&hv)[RU]) {
#pragma unroll
        for (int u = 0; u < RU; ++u) {
            const int row = min(r + u, n_rows - 1);
            xv[u] = reinterpret_cast<const f32x4*>(x + (size_t)row * (D4 * 4))[q];
            hv[u] = reinterpret_cast<const f32x4*>(x_hat + (size_t)row * (D4 * 4))[q];
        }
    };
    f32x4 xq[PF][RU], hq[PF][RU];
#pragma unroll
    for (int s = 0; s < PF; ++s)
        if (r0 + RU * s < r1) load_rows(r0 + RU * s, xq[s], hq[s]);
    for (int rb = r0; rb < r1; rb += RU * PF) {
#pragma unroll
      for (int s = 0; s < PF; ++s) {
        const int r = rb + RU * s;
        if (r >= r1) break;  // (uniform over the workgroup)
        f32x4 xc[RU], hc[RU];
#pragma unroll
        for (int u = 0; u < RU; ++u) { xc[u] = xq[s][u]; hc[u] = hq[s][u]; }
        if (r + RU * PF < r1) load_rows(r + RU * PF, xq[s], hq[s]);
        float p[NVAL];
#pragma unroll
        for (int j = 0; j < ND; ++j) {
            const f32x4 we = We[j][q];
#pragma unroll
            for (int u = 0; u < RU; ++u) p[u * ND + j] = (xc[u][0] * we[0] + xc[u][1] * we[1]) + (xc[u][2] * we[2] + xc[u][3] * we[3]);
        }
        {
            const float rs = aux_reduce_scatter16(p, lane);
            if ((lane & 3) == 0) shA[lane >> 2][w] = rs;
        }
        __syncthreads();
        float Hc[RU][ND];
#pragma unroll
        for (int u = 0; u < RU; ++u)
#pragma unroll
            for (int j = 0; j < ND; ++j) {
                const f32x4 t = *reinterpret_cast<const f32x4*>(&shA[u * ND + j][0]);
                Hc[u][j] = (j < nd) ? (((t[0] + t[1]) + t[2]) + t[3]) + be[j] : 0.f;
            }
        f32x4 g[RU];
        float sse[RU];
#pragma unroll
        for (int u = 0; u < RU; ++u) {
            f32x4 e = bd4;
#pragma unroll
            for (int j = 0; j < ND; ++j) e += Hc[u][j] * Wd[j][q];
            sse[u] = 0.f;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float diff = e[c] - (xc[u][c] - hc[u][c]);
                sse[u] += diff * diff;
                g[u][c] = gscale * diff;
            }
            if (u > 0 && r + u >= r1) { g[u] = f32x4{0.f, 0.f, 0.f, 0.f}; sse[u] = 0.f; }  // (rows past the block: clamped loads)
        }
#pragma unroll
        for (int j = 0; j < ND; ++j) {
            const f32x4 wd = Wd[j][q];
#pragma unroll
            for (int u = 0; u < RU; ++u) p[u * ND + j] = (g[u][0] * wd[0] + g[u][1] * wd[1]) + (g[u][2] * wd[2] + g[u][3] * wd[3]);
        }
        {
            const float rs = aux_reduce_scatter16(p, lane);
            if ((lane & 3) == 0) shB[lane >> 2][w] = rs;
#pragma unroll
            for (int u = 0; u < RU; ++u) {
                const float su = wave_sum(sse[u]);
                if (lane == 0) shB[NVAL + u][w] = su;
            }
        }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < RU; ++u) {
#pragma unroll
            for (int j = 0; j < ND; ++j) {
                const f32x4 t = *reinterpret_cast<const f32x4*>(&shB[u * ND + j][0]);
                const float da = ((t[0] + t[1]) + t[2]) + t[3];
                accD[j] += Hc[u][j] * g[u];
                accE[j] += da * xc[u];
                accbe[j] += da;
            }
            accb += g[u];
        }
        if (threadIdx.x < RU && r + (int)threadIdx.x < r1) {
            const f32x4 t = *reinterpret_cast<const f32x4*>(&shB[NVAL + threadIdx.x][0]);
            rowstats[r + threadIdx.x].aux_sse = ((t[0] + t[1]) + t[2]) + t[3];
        }
      }
    }
    // block partials, in launch_aux_small_wgrad's layout with NDO rows per half: [blk][2][NDO][D]
    f32x4* const pb = reinterpret_cast<f32x4*>(part) + (size_t)blockIdx.x * 2 * NDO * D4;
#pragma unroll
    for (int j = 0; j < ND; ++j) {
        if (j < nd) {
            pb[(size_t)j * D4 + q] = accD[j];
            pb[(size_t)(NDO + j) * D4 + q] = accE[j];
        }
    }
    reinterpret_cast<f32x4*>(partb)[(size_t)blockIdx.x * D4 + q] = accb;
    if (threadIdx.x == 0) {
#pragma unroll
        for (int j = 0; j < NDO; ++j) partbe[(size_t)blockIdx.x * NDO + j] = j < ND ? accbe[j < ND ? j : 0] : 0.f;
    }
}

// Weight gradients of the same: per block of 64 rows, part[blk][0][j][:] = sum_b A[b][j] g_aux[b][:] and
// part[blk][1][j][:] = sum_b dA[b][j] x[b][:] (rows in ascending order); a column sum over the blocks finishes them.
__global__ __launch_bounds__(256) void aux_small_wgrad_kernel(const float* A, const float* dA, const float* g_aux, const float* x,
                                                              int n_rows, int D, const int32_t* nd_dev, float* part) {
    const int nd = *nd_dev;
    if (nd <= 0 || nd > AUX_SMALL_MAX) return;
    constexpr int ndp = AUX_SMALL_MAX;
    const int r0 = blockIdx.x * 64, r1 = min(n_rows, r0 + 64);
    const int D4 = D >> 2;
    // (one group of eight dead latents per workgroup, blockIdx.y: a launch covers AUX_SMALL_MAX / 8 groups and the ones past the
    // device-side count leave at once -- with one workgroup per 64 rows looping over the groups the kernel had a single wave
    // per SIMD and spent its time waiting for row loads: 165 us at 30 dead latents)
    {
        const int j0 = blockIdx.y * 8;
        if (j0 >= nd) return;
        for (int q = threadIdx.x; q < D4; q += 256) {
            f32x4 ad[8], ae[8];
#pragma unroll
            for (int t = 0; t < 8; ++t) { ad[t] = f32x4{0.f, 0.f, 0.f, 0.f}; ae[t] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll 4
            for (int r = r0; r < r1; ++r) {
                const f32x4 g4 = reinterpret_cast<const f32x4*>(g_aux + (size_t)r * D)[q];
                const f32x4 x4 = reinterpret_cast<const f32x4*>(x + (size_t)r * D)[q];
                // the row's coefficients: the same 16-byte words for every thread (ndp % 4 == 0, j0 % 8 == 0); columns past
                // nd hold zeros (A) or are never written out (the loop below), so no per-column test is needed here
                const f32x4* ar = reinterpret_cast<const f32x4*>(A + (size_t)r * ndp + j0);
                const f32x4* dr = reinterpret_cast<const f32x4*>(dA + (size_t)r * ndp + j0);
                const bool two = j0 + 4 < ndp;
                const f32x4 a0 = ar[0], d0 = dr[0];
                const f32x4 a1 = two ? ar[1] : f32x4{0.f, 0.f, 0.f, 0.f}, d1 = two ? dr[1] : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    ad[t] += a0[t] * g4; ae[t] += d0[t] * x4;
                    ad[4 + t] += a1[t] * g4; ae[4 + t] += d1[t] * x4;
                }
            }
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                if (j0 + t < nd) {
                    float* base = part + (size_t)blockIdx.x * 2 * ndp * D;
                    reinterpret_cast<f32x4*>(base + (size_t)(j0 + t) * D)[q] = ad[t];
                    reinterpret_cast<f32x4*>(base + (size_t)(ndp + j0 + t) * D)[q] = ae[t];
                }
            }
        }
    }
}

// ---- 9 ... 64 dead latents on the fp32 matrix cores (v_mfma_f32_32x32x2_f32) -------------------------------------------------
// The five-pass kernels above do the few-dead-latents algebra on the vector ALUs: 0.21 + 0.11 ms at 30 dead latents for 5 GFLOP of
// skinny contractions, bound by fmas, reduce-scatters and LDS traffic.  The same contractions as 32-wide fp32 MFMA tiles (true
// fp32 multiply-adds: no operand splitting, 157 TFLOP/s dense) take what their passes over x / x_hat / g_aux cost:
//   aux_mfma_forward_kernel  codes H = x W_enc[:, dl] + b_enc[dl], E = H W_dec[dl] + b_dec, residual, loss share, g_aux,
//                            dA = g_aux W_dec[dl]^T -- one pass over x and x_hat per 32-row tile
//   aux_mfma_wgrad_kernel    the block partials of dWd = A^T g_aux and dWe = dA^T x (aux_small_wgrad_kernel's layout and finish)
// Instruction shape: D[32 x 32] += A[32 x 2] B[2 x 32]; lane l supplies A[l % 32][l / 32] and B[l / 32][l % 32] and holds, in
// register r of the result, row 8 (r / 4) + 4 (l / 32) + r % 4 of column l % 32.  Which k a "slot" l / 32 stands for is free as
// long as both operands agree, so every lane reads CONTIGUOUS pieces of its row (codes: eight floats per chunk of sixteen
// columns; reconstruction: half of the row's codes; wgrad: 32 rows of one column).  NL = 1 / 2 blocks of 32 latents; latents past
// the count are zero operands.  d_model % 128 == 0.
typedef float f32x16 __attribute__((ext_vector_type(16)));
__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0); }
__device__ __forceinline__ int mfma32_row(int r, int h) { return 8 * (r >> 2) + 4 * h + (r & 3); }

// One workgroup = 32 rows.  Phase 1: wave w forms the codes' share of columns [w D / 4, (w + 1) D / 4); the four shares are added in
// wave order through LDS, the codes go out (A) and stay in LDS as the next phase's operand.  Phase 2: wave w walks its D / 128 column
// blocks of 32: E block, residual, g_aux block (out, and transposed through a per-wave LDS tile: the result holds a COLUMN per lane,
// the next product wants a ROW per lane), dA share of the block.  The dA shares of the four waves are added like the codes'.
template <int NL>
__global__ __launch_bounds__(256) void aux_mfma_forward_kernel(const float* __restrict__ x, const float* __restrict__ x_hat,
                                                               const float* __restrict__ We, const float* __restrict__ Wd,
                                                               const float* __restrict__ b_enc, const float* __restrict__ b_dec,
                                                               const int32_t* __restrict__ dl, int n_rows, int D, const int32_t* nd_dev,
                                                               float gscale, float* __restrict__ A, float* __restrict__ dA,
                                                               float* __restrict__ g_aux, RowStats* rowstats, int ndp, int nd_min) {
    // ndp: row pitch of A / dA (AUX_SMALL_MAX, or AUX_MFMA_MAX for the wide dead sets); nd_min: the launch sequence enqueues one
    // instantiation per count window [nd_min, 32 NL] and the device-side count picks the one that runs
    const int nd = *nd_dev;
    if (nd < nd_min || nd <= 0 || nd > 32 * NL) return;
    constexpr int KH = 16 * NL;  // KH: latents per slot of the reconstruction's contraction
    __shared__ float sh[4][16][64];        // the four waves' shares of one 32 x 32 block of a tile (blocks go through it in turn)
    __shared__ float As[32][32 * NL + 1];  // the tile's codes, row-major
    __shared__ float Gs[4][32][33];        // per wave: a 32 x 32 block of g_aux, row-major
    __shared__ float shs[4][32];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int i = lane & 31, h = lane >> 5;
    const int row0 = blockIdx.x * 32;
    const int rowc = min(row0 + i, n_rows - 1);
    // every share of the tile, added in wave order; `fin(row, lat, sum)` receives the 32 x (32 NL) sums
    auto tile_sum = [&](const f32x16 (&t)[NL], auto fin) {
#pragma unroll
        for (int lb = 0; lb < NL; ++lb) {
#pragma unroll
            for (int r = 0; r < 16; ++r) sh[w][r][lane] = t[lb][r];
            __syncthreads();
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int idx = threadIdx.x + 256 * u;
                const int r = (idx >> 6) & 15, l = idx & 63;
                fin(mfma32_row(r, l >> 5), 32 * lb + (l & 31), ((sh[0][r][l] + sh[1][r][l]) + sh[2][r][l]) + sh[3][r][l]);
            }
            __syncthreads();
        }
    };
    // ---- phase 1: codes ----
    {
        const int kq = D >> 2;  // columns per wave (a multiple of 32)
        const float* mr = x + (size_t)rowc * D + w * kq + 8 * h;
        f32x16 acc[NL];
#pragma unroll
        for (int lb = 0; lb < NL; ++lb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[lb][r] = 0.f;
        // (two chunks of sixteen columns per trip, their loads issued together)
#pragma unroll 1
        for (int c = 0; c < kq; c += 32) {
            f32x4 m[4], q[NL][4];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                m[2 * u] = *reinterpret_cast<const f32x4*>(mr + c + 16 * u);
                m[2 * u + 1] = *reinterpret_cast<const f32x4*>(mr + c + 16 * u + 4);
#pragma unroll
                for (int lb = 0; lb < NL; ++lb) {
                    const bool ok = 32 * lb + i < nd;
                    const float* rr = We + (size_t)(32 * lb + i) * D + w * kq + 8 * h + c + 16 * u;
                    q[lb][2 * u] = ok ? *reinterpret_cast<const f32x4*>(rr) : f32x4{0.f, 0.f, 0.f, 0.f};
                    q[lb][2 * u + 1] = ok ? *reinterpret_cast<const f32x4*>(rr + 4) : f32x4{0.f, 0.f, 0.f, 0.f};
                }
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int lb = 0; lb < NL; ++lb) acc[lb] = mfma32(m[u][e], q[lb][u][e], acc[lb]);
        }
        tile_sum(acc, [&](int rl, int lat, float v) {
            const float a = lat < nd ? v + b_enc[dl[lat]] : 0.f;
            As[rl][lat] = a;
            if (row0 + rl < n_rows) {
                A[(size_t)(row0 + rl) * ndp + lat] = a;
                if (NL == 1) A[(size_t)(row0 + rl) * ndp + 32 + lat] = 0.f;
            }
        });
    }
    // ---- phase 2: reconstruction, residual, g_aux, dA ----
    float av[KH];  // this lane's row: slot h <-> latents KH h ... KH h + KH - 1
#pragma unroll
    for (int p = 0; p < KH; ++p) av[p] = As[i][KH * h + p];
    float sse[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) sse[r] = 0.f;
    f32x16 accD[NL];
#pragma unroll
    for (int lb = 0; lb < NL; ++lb)
#pragma unroll
        for (int r = 0; r < 16; ++r) accD[lb][r] = 0.f;
    const int nb = D >> 7;
#pragma unroll 1
    for (int b = 0; b < nb; ++b) {
        const int c0 = (w * nb + b) * 32, col = c0 + i;
        float wv[KH];
#pragma unroll
        for (int p = 0; p < KH; ++p) wv[p] = (KH * h + p < nd) ? Wd[(size_t)(KH * h + p) * D + col] : 0.f;
        const float bd = b_dec[col];
        float xv[16], hv[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const size_t o = (size_t)min(row0 + mfma32_row(r, h), n_rows - 1) * D + col;
            xv[r] = x[o]; hv[r] = x_hat[o];
        }
        // (the dA product's second operand: W_dec[dl] rows of the latent blocks, this lane's 16 columns of the block)
        f32x4 wd4[NL][4];
#pragma unroll
        for (int lb = 0; lb < NL; ++lb) {
            const bool ok = 32 * lb + i < nd;
            const f32x4* wr = reinterpret_cast<const f32x4*>(Wd + (size_t)(32 * lb + i) * D + c0 + 16 * h);
#pragma unroll
            for (int q = 0; q < 4; ++q) wd4[lb][q] = ok ? wr[q] : f32x4{0.f, 0.f, 0.f, 0.f};
        }
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int p = 0; p < KH; ++p) acc = mfma32(av[p], wv[p], acc);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int rl = mfma32_row(r, h);
            const float diff = (acc[r] + bd) - (xv[r] - hv[r]);
            sse[r] = __builtin_fmaf(diff, diff, sse[r]);
            const float gv = gscale * diff;
            if (row0 + rl < n_rows) g_aux[(size_t)(row0 + rl) * D + col] = gv;
            Gs[w][rl][i] = gv;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // (the wave's own LDS writes, in order, before its reads of the tile)
        float gr[16];
#pragma unroll
        for (int p = 0; p < 16; ++p) gr[p] = Gs[w][i][16 * h + p];
#pragma unroll
        for (int p = 0; p < 16; ++p)
#pragma unroll
            for (int lb = 0; lb < NL; ++lb) accD[lb] = mfma32(gr[p], wd4[lb][p >> 2][p & 3], accD[lb]);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // (reads done before the next block's writes)
    }
    tile_sum(accD, [&](int rl, int lat, float v) {
        if (row0 + rl < n_rows) {
            dA[(size_t)(row0 + rl) * ndp + lat] = lat < nd ? v : 0.f;
            if (NL == 1) dA[(size_t)(row0 + rl) * ndp + 32 + lat] = 0.f;
        }
    });
    // per-row sums of squares: over the 32 lanes of each half, then over the four waves in order
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        float v = sse[r];
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
        if (i == 0) shs[w][mfma32_row(r, h)] = v;
    }
    __syncthreads();
    if (threadIdx.x < 32 && row0 + threadIdx.x < n_rows)
        rowstats[row0 + threadIdx.x].aux_sse = ((shs[0][threadIdx.x] + shs[1][threadIdx.x]) + shs[2][threadIdx.x]) + shs[3][threadIdx.x];
}

// one block of 64 rows per blockIdx.x (aux_small_wgrad_kernel's partial layout: part[blk][2][ndp][D]); the D / 32 column blocks go round
// the 4 gridDim.y waves of the row block; slot h <-> rows 32 h ... 32 h + 31.  A column block's 2 x 32 operand loads are issued
// together and the next block's before this one's products (a first version loaded eight rows at a time with one wave per SIMD:
// 124 us at configs[1], every group waiting out a memory round trip)
template <int NL>
__global__ __launch_bounds__(256, 1) void aux_mfma_wgrad_kernel(const float* __restrict__ A, const float* __restrict__ dA, const float* __restrict__ g,
                                                                const float* __restrict__ x, int n_rows, int D, const int32_t* nd_dev,
                                                                float* __restrict__ part, float* __restrict__ partb, float* __restrict__ partbe,
                                                                int ndp, int nd_min, int nd_max, int lat0) {
    // partb[blk][D] / partbe[blk][ndp]: the block's column sums of g_aux (db_dec's share) and of dA (db_enc[dl]) -- the operands are in
    // registers anyway; rows 0 ... 31 in order, then rows 32 ... 63, then the two halves: one fixed order
    // this launch: latents [lat0, lat0 + 32 NL) of a dead set whose device-side count lies in [nd_min, nd_max] (the launch sequence
    // enqueues one launch per window and latent range; the count picks what runs); ndp: row pitch of A / dA and of the partials
    const int nd = *nd_dev;
    if (nd < nd_min || nd <= 0 || nd > nd_max) return;
    const int lane = threadIdx.x & 63;
    const int i = lane & 31, h = lane >> 5;
    const int r0 = blockIdx.x * 64 + 32 * h;
    const int wi = blockIdx.y * 4 + (threadIdx.x >> 6), nw = gridDim.y * 4, NB = D >> 5;
    if (wi >= NB) return;
    float av[NL][32], dv[NL][32];
#pragma unroll
    for (int p = 0; p < 32; ++p) {
        const bool ok = r0 + p < n_rows;  // (rows past the end: zero coefficients against the last row's values)
        const size_t o = (size_t)min(r0 + p, n_rows - 1) * ndp + lat0 + i;
#pragma unroll
        for (int lb = 0; lb < NL; ++lb) {
            av[lb][p] = ok ? A[o + 32 * lb] : 0.f;
            dv[lb][p] = ok ? dA[o + 32 * lb] : 0.f;
        }
    }
    if (wi == 0 && partbe != nullptr) {
#pragma unroll
        for (int lb = 0; lb < NL; ++lb) {
            float t = 0.f;
#pragma unroll
            for (int p = 0; p < 32; ++p) t += dv[lb][p];
            const float o = __shfl_xor(t, 32, 64);
            if (h == 0) {
                partbe[(size_t)blockIdx.x * ndp + lat0 + 32 * lb + i] = t + o;
                if (NL == 1) partbe[(size_t)blockIdx.x * ndp + 32 + i] = 0.f;
            }
        }
    }
    float* const base = part + (size_t)blockIdx.x * 2 * ndp * D;
    auto load = [&](int cb, float (&gv)[32], float (&xv)[32]) {
        const int col = cb * 32 + i;
#pragma unroll
        for (int p = 0; p < 32; ++p) {
            const size_t o = (size_t)min(r0 + p, n_rows - 1) * D + col;
            gv[p] = g[o]; xv[p] = x[o];
        }
    };
    auto products = [&](int cb, const float (&gv)[32], const float (&xv)[32]) {
        const int col = cb * 32 + i;
        if (partb != nullptr) {
            float t = 0.f;
#pragma unroll
            for (int p = 0; p < 32; ++p) t += (r0 + p < n_rows) ? gv[p] : 0.f;
            const float o = __shfl_xor(t, 32, 64);
            if (h == 0) partb[(size_t)blockIdx.x * D + col] = t + o;
        }
#pragma unroll
        for (int lb = 0; lb < NL; ++lb) {
            f32x16 c0, c1;
#pragma unroll
            for (int r = 0; r < 16; ++r) { c0[r] = 0.f; c1[r] = 0.f; }
#pragma unroll
            for (int p = 0; p < 32; ++p) { c0 = mfma32(av[lb][p], gv[p], c0); c1 = mfma32(dv[lb][p], xv[p], c1); }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int lat = lat0 + 32 * lb + mfma32_row(r, h);
                if (lat < nd) {
                    base[(size_t)lat * D + col] = c0[r];
                    base[(size_t)(ndp + lat) * D + col] = c1[r];
                }
            }
        }
    };
    float ga[32], xa[32], gb[32], xb[32];
    load(wi, ga, xa);
#pragma unroll 1
    for (int cb = wi; cb < NB; cb += 2 * nw) {
        const bool more1 = cb + nw < NB, more2 = cb + 2 * nw < NB;
        if (more1) load(cb + nw, gb, xb);
        __builtin_amdgcn_sched_barrier(0);
        products(cb, ga, xa);
        if (more2) load(cb + 2 * nw, ga, xa);
        __builtin_amdgcn_sched_barrier(0);
        if (more1) products(cb + nw, gb, xb);
    }
}

// out[i] = sum_j parts[j][i] in the order j = 0, 1, ...: the contraction slices of a split product (fixed order, so the
// result does not depend on scheduling)
__global__ __launch_bounds__(256) void sum_parts_kernel(const float* parts, int n_parts, long n4, float* out) {
    const f32x4* p = reinterpret_cast<const f32x4*>(parts);
    for (long q = (long)blockIdx.x * 256 + threadIdx.x; q < n4; q += (long)gridDim.x * 256) {
        f32x4 acc = p[q];
        for (int j = 1; j < n_parts; ++j) acc += p[(long)j * n4 + q];
        reinterpret_cast<f32x4*>(out)[q] = acc;
    }
}
// out[which][c] = sum_blk part[blk][which][c] for c < nd * D (which = blockIdx.y: 0 decoder rows, 1 encoder rows; blocks in
// ascending order): the finish of aux_small_wgrad_kernel.  The generic column sum splits over ROWS of its input -- four
// workgroups for these 256 block rows of 30 000+ columns; this one splits over columns.
__global__ __launch_bounds__(256) void aux_small_wsum_kernel(const float* part, int n_blk, int D, const int32_t* nd_dev, float* dWd,
                                                             float* dWe, int ndp) {
    const int nd = *nd_dev;
    if (nd <= 0 || nd > ndp) return;
    const long n4 = (long)nd * (D >> 2), stride4 = (long)2 * ndp * (D >> 2);
    const f32x4* p = reinterpret_cast<const f32x4*>(part) + (long)blockIdx.y * ndp * (D >> 2);
    f32x4* out = reinterpret_cast<f32x4*>(blockIdx.y == 0 ? dWd : dWe);
    for (long q = (long)blockIdx.x * 256 + threadIdx.x; q < n4; q += (long)gridDim.x * 256) {
        f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
        int b = 0;
        for (; b + 4 <= n_blk; b += 4) {  // four loads in flight, added in block order
            const f32x4 a0 = p[(long)b * stride4 + q], a1 = p[(long)(b + 1) * stride4 + q], a2 = p[(long)(b + 2) * stride4 + q],
                        a3 = p[(long)(b + 3) * stride4 + q];
            acc += a0; acc += a1; acc += a2; acc += a3;
        }
        for (; b < n_blk; ++b) acc += p[(long)b * stride4 + q];
        out[q] = acc;
    }
}
// The same finish for aux_small_fused_kernel's partials ([blk][2][AUX_FUSED_MAX][D], a block = AUX_FUSED_ROWS activation rows: a
// thousand blocks of a few KB).  A workgroup of sixteen waves owns 64 float4 columns of one half; wave w adds blocks
// [w n/16, (w + 1) n/16) in order, eight loads in flight, and wave 0 adds the sixteen sums in wave order: the result does not
// depend on scheduling, and no wave walks more than n/16 blocks (aux_small_wsum_kernel on these partials: 125 us for ONE dead
// latent -- 256 lanes walking 1 024 blocks four loads at a time).
__global__ __launch_bounds__(1024) void aux_fused_wsum_kernel(const float* part, int n_blk, int D, const int32_t* nd_dev, float* dWd,
                                                              float* dWe, const float* partb, float* db_out, int db_accumulate,
                                                              const float* partbe, float* dbe, const RowStats* rs, int n_rows,
                                                              float alpha, saev_step_stats* stats, int ndo) {
    // ndo: latent rows per half of a block partial (AUX_FUSED_MAX for aux_small_fused_kernel's, AUX_SMALL_MAX for aux_mfma_wgrad_kernel's)
    const int nd = *nd_dev;
    if (nd <= 0 || nd > ndo) return;
    __shared__ f32x4 sh[16][64];
    const int D4 = D >> 2;
    if (blockIdx.y == 4) {  // the auxiliary loss of the step from the rows' shares (stats_reduce_kernel's with_aux = 2 launch)
        if (blockIdx.x != 0 || rs == nullptr) return;
        __shared__ double shd[16];
        double t = 0.0;
        for (int r = threadIdx.x; r < n_rows; r += 1024) t += (double)rs[r].aux_sse;
        t = wave_sum_d(t);
        if ((threadIdx.x & 63) == 0) shd[threadIdx.x >> 6] = t;
        __syncthreads();
        if (threadIdx.x == 0) {
            double tt = 0.0;
            for (int w = 0; w < 16; ++w) tt += shd[w];
            stats->aux = (float)((double)alpha * tt / ((double)n_rows * (double)D));
        }
        return;
    }
    // blockIdx.y: 0 decoder rows, 1 encoder rows (nd x D out of [blk][2][AUX_FUSED_MAX][D]); 2 db_dec's share (D out of [blk][D]);
    // 3 db_enc[dl] (AUX_FUSED_MAX out of [blk][AUX_FUSED_MAX]) -- the last two used to be two column-sum launches each
    const int kind = blockIdx.y;
    const long n4 = kind < 2 ? (long)nd * D4 : (kind == 2 ? (long)D4 : (long)(ndo / 4));
    const long stride4 = kind < 2 ? (long)2 * ndo * D4 : (kind == 2 ? (long)D4 : (long)(ndo / 4));
    if ((long)blockIdx.x * 64 >= n4) return;  // (uniform)
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const long col = (long)blockIdx.x * 64 + lane;
    const bool ok = col < n4;
    const f32x4* base = kind < 2 ? reinterpret_cast<const f32x4*>(part) + (long)kind * ndo * D4
                                 : reinterpret_cast<const f32x4*>(kind == 2 ? partb : partbe);
    const f32x4* p = base + (ok ? col : 0);
    const int chunk = (n_blk + 15) / 16, b0 = w * chunk, b1 = min(n_blk, b0 + chunk);
    f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
    int b = b0;
    for (; b + 8 <= b1; b += 8) {
        f32x4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = p[(long)(b + u) * stride4];
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += v[u];
    }
    for (; b < b1; ++b) acc += p[(long)b * stride4];
    sh[w][lane] = acc;
    __syncthreads();
    if (w != 0 || !ok) return;
    f32x4 tot = sh[0][lane];
#pragma unroll
    for (int v = 1; v < 16; ++v) tot += sh[v][lane];
    if (kind < 2) {
        reinterpret_cast<f32x4*>(kind == 0 ? dWd : dWe)[col] = tot;
    } else if (kind == 2) {
        f32x4* o = reinterpret_cast<f32x4*>(db_out) + col;
        *o = db_accumulate ? *o + tot : tot;
    } else {
        reinterpret_cast<f32x4*>(dbe)[col] = tot;
    }
}

// gW_dec[dl[j], :] += dWd[j, :]; gW_encT[dl[j], :] += dWe[j, :]; gb_enc[dl[j]] += dbe[j]   (one wave per dead latent)
__global__ __launch_bounds__(256) void scatter_add_dead_kernel(const int32_t* dl, int nd, int D, const float* dWd,
                                                               const float* dWe, const float* dbe, float* gW_dec,
                                                               float* gW_encT, float* gb_enc, int lat_lo, int lat_hi,
                                                               const int32_t* nd_dev, int part, float2* row_proj,
                                                               const float* W_dec, int project, float* enc_sq, int32_t* lat_unused,
                                                               const int32_t* starts) {
    // part: 0 = all three gradients; 1 = the decoder rows only; 2 = the encoder rows and bias only (saev_backward_rows_part)
    if (nd_dev != nullptr) nd = min(nd, *nd_dev);
    const int lane = threadIdx.x & 63;
    const int j = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (j >= nd) return;
    const int i = dl[j];
    if (i < lat_lo || i >= lat_hi) return;
    const f32x4* a = reinterpret_cast<const f32x4*>(dWd + (size_t)j * D);
    const f32x4* e = reinterpret_cast<const f32x4*>(dWe + (size_t)j * D);
    f32x4* oa = reinterpret_cast<f32x4*>(gW_dec + (size_t)i * D);
    f32x4* oe = reinterpret_cast<f32x4*>(gW_encT + (size_t)i * D);
    float dot = 0.f, nsq = 0.f, esq = 0.f, old_d = 0.f, old_e = 0.f;
    const bool enc_unwritten = lat_unused != nullptr && lat_unused[i] != 0;  // (wave-uniform)
    // (the clip norm already holds the squares of this row's main-path part -- the backward's passes add them up per wave -- unless
    // the latent was cut by run boundaries, whose statistics are per row like the ones written here)
    bool counted = false;
    if (starts != nullptr && !enc_unwritten) {
        const int s0 = starts[i], e0 = starts[i + 1];
        counted = e0 > s0 && !(e0 - s0 >= DWS_RUN && s0 / DWS_RUN != (e0 - 1) / DWS_RUN);
    }
    for (int q = lane; q < (D >> 2); q += 64) {
        if (part != 2) {
            const f32x4 o = enc_unwritten ? f32x4{0.f, 0.f, 0.f, 0.f} : oa[q];  // (a flagged latent: neither gradient row has been written)
            if (counted) old_d = __builtin_fmaf(o[3], o[3], __builtin_fmaf(o[2], o[2], __builtin_fmaf(o[1], o[1], __builtin_fmaf(o[0], o[0], old_d))));
            const f32x4 g = o + a[q];
            oa[q] = g;
            if (row_proj != nullptr) {  // the row changed: refresh its projection coefficient / projected squares (DwRowsArgs::row_proj)
                const f32x4 w = project ? reinterpret_cast<const f32x4*>(W_dec + (size_t)i * D)[q] : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int c = 0; c < 4; ++c) { dot = __builtin_fmaf(g[c], w[c], dot); nsq = __builtin_fmaf(w[c], w[c], nsq); }
            }
        }
        if (part != 1) {
            const f32x4 o = enc_unwritten ? f32x4{0.f, 0.f, 0.f, 0.f} : oe[q];
            if (counted) old_e = __builtin_fmaf(o[3], o[3], __builtin_fmaf(o[2], o[2], __builtin_fmaf(o[1], o[1], __builtin_fmaf(o[0], o[0], old_e))));
            const f32x4 ge = o + e[q];
            oe[q] = ge;
#pragma unroll
            for (int c = 0; c < 4; ++c) esq = __builtin_fmaf(ge[c], ge[c], esq);
        }
    }
    if (lane == 0 && part != 1) gb_enc[i] += dbe[j];
    if (enc_unwritten && lane == 0) lat_unused[i] = 0;
    if (enc_sq != nullptr && part != 1) {  // the row changed: its squares again (DwRowsArgs::enc_sq)
        esq = wave_sum(esq);
        if (counted) esq -= wave_sum(old_e);
        if (lane == 0) enc_sq[i] = esq;
    }
    if (row_proj != nullptr && part != 2) {  // same arithmetic, in the same order, as rpg_row_stats (common.h)
        dot = wave_sum(dot); nsq = wave_sum(nsq);
        const float sc = (project && nsq > 0.f) ? dot / nsq : 0.f;
        float sq = 0.f;
        for (int q = lane; q < (D >> 2); q += 64) {
            const f32x4 g = oa[q];  // (this lane's own stores)
            const f32x4 w = project ? reinterpret_cast<const f32x4*>(W_dec + (size_t)i * D)[q] : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int c = 0; c < 4; ++c) { const float t = rpg_apply(g[c], sc, w[c]); sq = __builtin_fmaf(t, t, sq); }
        }
        sq = wave_sum(sq);
        if (counted) sq -= wave_sum(old_d);
        if (lane == 0) row_proj[i] = float2{sc, sq};
    }
}

int grid_for(long n) { return (int)std::max<long>(1, std::min<long>((n + 255) / 256, 8192)); }

template <typename F>
hipError_t dispatch_nv(int D, F&& f) {
    const int nv = (D / 4 + 63) / 64;
    switch (nv) {
        case 1: f(std::integral_constant<int, 1>()); break;
        case 2: f(std::integral_constant<int, 2>()); break;
        case 3: f(std::integral_constant<int, 3>()); break;
        case 4: f(std::integral_constant<int, 4>()); break;
        case 5: f(std::integral_constant<int, 5>()); break;
        case 6: f(std::integral_constant<int, 6>()); break;
        case 7: case 8: f(std::integral_constant<int, 8>()); break;
        case 9: case 10: case 11: case 12: f(std::integral_constant<int, 12>()); break;  // d_model <= 3072
        case 13: case 14: case 15: case 16: f(std::integral_constant<int, 16>()); break;  // d_model <= 4096
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

}  // namespace

hipError_t launch_dead_compact(const int32_t* dead, int S, int32_t* list, hipStream_t s, const int32_t* n_dead_dev) {
    hipLaunchKernelGGL(dead_compact_kernel, dim3(1), dim3(1024), 0, s, dead, S, list, n_dead_dev);
    return hipGetLastError();
}
hipError_t launch_gather_dead(const float* W_enc, const float* W_dec, const int32_t* dl, int nd, int ndp, int D, int S,
                              float* Wenc_dead, float* Wdec_dead, hipStream_t s, const int32_t* nd_dev) {
    hipLaunchKernelGGL(gather_dead_kernel, dim3(grid_for((long)D * ndp + (long)ndp * (D >> 2))), dim3(256), 0, s, W_enc,
                       W_dec, dl, nd, ndp, D, S, Wenc_dead, Wdec_dead, nd_dev);
    return hipGetLastError();
}
hipError_t launch_dead_bias_vec(const float* b_enc, const int32_t* dl, int nd, int ndp, float* out, hipStream_t s, bool pad_zero,
                                const int32_t* nd_dev) {
    hipLaunchKernelGGL(dead_bias_vec_kernel, dim3((ndp + 255) / 256), dim3(256), 0, s, b_enc, dl, nd, ndp, out,
                       pad_zero ? 0.f : NEG_INF, nd_dev);
    return hipGetLastError();
}
hipError_t launch_aux_scatter(const int32_t* idx, const float* val, int n_rows, int k, int stride, int ndp, float* A,
                              uint8_t* mask, hipStream_t s, const int32_t* k_dev) {
    hipLaunchKernelGGL(aux_scatter_kernel, dim3(grid_for((long)n_rows * k)), dim3(256), 0, s, idx, val, (long)n_rows, k,
                       stride, ndp, A, mask, k_dev);
    return hipGetLastError();
}
hipError_t launch_aux_resid(float* E, const float* x, const float* x_hat, const float* b_dec, int n_rows, int D,
                            float gscale, RowStats* rowstats, hipStream_t s, const int32_t* nd_dev, float* part, float* pair) {
    hipError_t e = dispatch_nv(D, [&](auto nv) {
        hipLaunchKernelGGL(aux_resid_kernel<decltype(nv)::value>, dim3((n_rows + 3) / 4), dim3(256), 0, s, E, x, x_hat,
                           b_dec, n_rows, D, gscale, rowstats, nd_dev, part);
    });
    if (e != hipSuccess || part == nullptr) return e;
    hipLaunchKernelGGL(pow2_parts_kernel, dim3(1), dim3(256), 0, s, part, (n_rows + 3) / 4, pair);
    return hipGetLastError();
}
int absmax_parts_max(int n_rows_cap) { return std::max(2048, (n_rows_cap + 3) / 4); }  // floats of `part` the launches here may write
hipError_t launch_absmax_pow2(const float* x, long n, float* part, float* pair, hipStream_t s) {
    const int grid = std::min(grid_for(n / 4), 2048);
    hipLaunchKernelGGL(absmax_parts_kernel, dim3(grid), dim3(256), 0, s, x, n / 4, part);  // n % 4 == 0
    hipLaunchKernelGGL(pow2_parts_kernel, dim3(1), dim3(256), 0, s, part, grid, pair);
    return hipGetLastError();
}
hipError_t launch_mask_apply_absmax(float* dA, const uint8_t* mask, long n, float* part, float* pair, hipStream_t s) {
    const int grid = std::min(grid_for(n / 4), 2048);
    hipLaunchKernelGGL(mask_apply_absmax_kernel, dim3(grid), dim3(256), 0, s, dA, mask, n / 4, part);  // n % 4 == 0
    hipLaunchKernelGGL(pow2_parts_kernel, dim3(1), dim3(256), 0, s, part, grid, pair);
    return hipGetLastError();
}
bool aux_select_supported(int ndp) { return ndp > 0 && ndp % 4 == 0 && ndp <= 4096; }
hipError_t launch_aux_select(const float* H, int n_rows, int ndp, int k, const int32_t* k_dev, float* A, uint8_t* mask, float* part,
                             float* pair, hipStream_t s) {
    const dim3 grid(std::max(1, std::min(1024, (n_rows + 3) / 4))), block(256);
    const int nv = (ndp + 255) / 256;
#define SAEV_AUX_SELECT(NV) hipLaunchKernelGGL(aux_select_kernel<NV>, grid, block, 0, s, H, n_rows, ndp, k, k_dev, A, mask, part)
    if (nv <= 1) SAEV_AUX_SELECT(1);
    else if (nv <= 2) SAEV_AUX_SELECT(2);
    else if (nv <= 4) SAEV_AUX_SELECT(4);
    else if (nv <= 8) SAEV_AUX_SELECT(8);
    else SAEV_AUX_SELECT(16);
    hipLaunchKernelGGL(pow2_parts_kernel, dim3(1), dim3(256), 0, s, part, (int)grid.x, pair);
#undef SAEV_AUX_SELECT
    return hipGetLastError();
}
hipError_t launch_mask_apply(float* dA, const uint8_t* mask, long n, hipStream_t s) {
    hipLaunchKernelGGL(mask_apply_kernel, dim3(grid_for(n)), dim3(256), 0, s, dA, mask, n);
    return hipGetLastError();
}
hipError_t launch_gather_dead_small(const float* W_enc, const float* W_dec, const int32_t* dl, const int32_t* nd_dev, int D,
                                    int S, float* WencT_dead, float* Wdec_dead, hipStream_t s, int cap) {
    hipLaunchKernelGGL(gather_dead_small_kernel, dim3(grid_for((long)cap * D)), dim3(256), 0, s, W_enc, W_dec, dl,
                       nd_dev, D, S, WencT_dead, Wdec_dead, cap);
    return hipGetLastError();
}
hipError_t launch_aux_small_fwd(const float* x, const float* x_hat, const float* WencT_dead, const float* Wdec_dead,
                                const float* b_enc, const float* b_dec, const int32_t* dl, int n_rows, int D,
                                const int32_t* nd_dev, float gscale, float* A, float* dA, float* g_aux, RowStats* rowstats,
                                hipStream_t s) {
    // the LDS-staged kernel: chunks of LC weight rows (<= 64 KB), four rows per wave up to d_model 1024, two up to 2048
    const int nv = (D / 4 + 63) / 64;
    const int LC = std::max(1, std::min(16, 16384 / D));
    auto lds = [&](auto kern, int RW) -> hipError_t {
        const size_t smem = ((size_t)LC * D + (size_t)4 * 2 * RW * 64) * sizeof(float);
        static size_t granted[9] = {0};  // per instantiation (indexed by NV): the largest dynamic LDS size already allowed
        if (smem > granted[nv]) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
            if (e != hipSuccess) return e;
            granted[nv] = smem;
        }
        hipLaunchKernelGGL(kern, dim3((n_rows + 4 * RW - 1) / (4 * RW)), dim3(256), smem, s, x, x_hat, WencT_dead, Wdec_dead, b_enc, b_dec,
                           dl, n_rows, D, nd_dev, gscale, A, dA, g_aux, rowstats, LC);
        return hipGetLastError();
    };
    switch (nv) {
        case 1: return lds(aux_small_fwd_lds_kernel<1, 4>, 4);
        case 2: return lds(aux_small_fwd_lds_kernel<2, 4>, 4);
        case 3: return lds(aux_small_fwd_lds_kernel<3, 4>, 4);
        case 4: return lds(aux_small_fwd_lds_kernel<4, 4>, 4);
        case 5: case 6: return lds(aux_small_fwd_lds_kernel<6, 2>, 2);
        case 7: case 8: return lds(aux_small_fwd_lds_kernel<8, 2>, 2);
        default: break;
    }
    return dispatch_nv(D, [&](auto nvc) {
        hipLaunchKernelGGL(aux_small_fwd_kernel<decltype(nvc)::value>, dim3((n_rows + 7) / 8), dim3(256), 0, s, x, x_hat,
                           WencT_dead, Wdec_dead, b_enc, b_dec, dl, n_rows, D, nd_dev, gscale, A, dA, g_aux, rowstats);
    });
}
hipError_t launch_aux_small_wsum(const float* part, int n_blk, int D, const int32_t* nd_dev, float* dWd, float* dWe, hipStream_t s, int ndp) {
    hipLaunchKernelGGL(aux_small_wsum_kernel, dim3(64, 2), dim3(256), 0, s, part, n_blk, D, nd_dev, dWd, dWe, ndp);
    return hipGetLastError();
}
hipError_t launch_aux_fused_wsum(const float* part, int n_blk, int D, const int32_t* nd_dev, float* dWd, float* dWe, hipStream_t s,
                                 const float* partb, float* db_out, int db_accumulate, const float* partbe, float* dbe,
                                 const RowStats* rs, int n_rows, float alpha, saev_step_stats* stats, int ndo) {
    hipLaunchKernelGGL(aux_fused_wsum_kernel, dim3((ndo * (D >> 2) + 63) / 64, rs != nullptr ? 5 : 4), dim3(1024), 0, s, part, n_blk, D,
                       nd_dev, dWd, dWe, partb, db_out, db_accumulate, partbe, dbe, rs, n_rows, alpha, stats, ndo);
    return hipGetLastError();
}
bool aux_fused_supported(int D) { return D % 256 == 0 && D >= 256 && D <= 1024; }
int aux_fused_blocks(int n_rows) { return (n_rows + AUX_FUSED_ROWS - 1) / AUX_FUSED_ROWS; }
hipError_t launch_aux_small_fused(const float* x, const float* x_hat, const float* WencT_dead, const float* Wdec_dead, const float* b_enc,
                                  const float* b_dec, const int32_t* dl, int n_rows, int D, const int32_t* nd_dev, float gscale,
                                  float* part, float* partb, float* partbe, RowStats* rowstats, hipStream_t s, int bound) {
    if (!aux_fused_supported(D)) return hipErrorInvalidValue;
    const dim3 grid(aux_fused_blocks(n_rows));
    // `bound` >= the device-side count: at most four dead latents (the usual state of a healthy run) take the variant that
    // provides for four -- half the registers in accumulators, four rows per trip, two trips in flight
    const bool four = bound <= 4;
    const size_t smem = (size_t)2 * (four ? 4 : AUX_FUSED_MAX) * D * sizeof(float);  // 64 KB at d_model 1024 and eight latents, next to < 1 KB of static LDS
    static bool attr_set = false;
    if (!attr_set) {
        const void* fns[8] = {reinterpret_cast<const void*>(&aux_small_fused_kernel<1, AUX_FUSED_MAX, AUX_FUSED_PF8>),
                              reinterpret_cast<const void*>(&aux_small_fused_kernel<2, AUX_FUSED_MAX, AUX_FUSED_PF8>),
                              reinterpret_cast<const void*>(&aux_small_fused_kernel<3, AUX_FUSED_MAX, AUX_FUSED_PF8>),
                              reinterpret_cast<const void*>(&aux_small_fused_kernel<4, AUX_FUSED_MAX, AUX_FUSED_PF8>),
                              reinterpret_cast<const void*>(&aux_small_fused_kernel<1, 4, AUX_FUSED_PF4>),
                              reinterpret_cast<const void*>(&aux_small_fused_kernel<2, 4, AUX_FUSED_PF4>),
                              reinterpret_cast<const void*>(&aux_small_fused_kernel<3, 4, AUX_FUSED_PF4>),
                              reinterpret_cast<const void*>(&aux_small_fused_kernel<4, 4, AUX_FUSED_PF4>)};
        for (const void* f : fns) {
            hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * AUX_FUSED_MAX * 1024 * (int)sizeof(float));
            if (e != hipSuccess) return e;
        }
        attr_set = true;
    }
#define AF(NW_)                                                                                                                            \
    do {                                                                                                                                   \
        if (four)                                                                                                                          \
            hipLaunchKernelGGL((aux_small_fused_kernel<NW_, 4, AUX_FUSED_PF4>), grid, dim3(64 * NW_), smem, s, x, x_hat, WencT_dead,      \
                               Wdec_dead, b_enc, b_dec, dl, n_rows, nd_dev, gscale, AUX_FUSED_ROWS, part, partb, partbe, rowstats);        \
        else                                                                                                                               \
            hipLaunchKernelGGL((aux_small_fused_kernel<NW_, AUX_FUSED_MAX, AUX_FUSED_PF8>), grid, dim3(64 * NW_), smem, s, x, x_hat,      \
                               WencT_dead, Wdec_dead, b_enc, b_dec, dl, n_rows, nd_dev, gscale, AUX_FUSED_ROWS, part, partb, partbe,       \
                               rowstats);                                                                                                  \
    } while (0)
    switch (D / 256) {
        case 1: AF(1); break;
        case 2: AF(2); break;
        case 3: AF(3); break;
        default: AF(4); break;
    }
#undef AF
    return hipGetLastError();
}
bool aux_mfma_supported(int D) { return D % 128 == 0 && D >= 128; }
// One instantiation per count window, each predicated on the device-side count (a launch whose window the count misses leaves in its
// first instruction, ~4 us): the host only knows a BOUND of the count -- the tracker record of a few steps ago -- and a bound of 100
// usually stands for 20 dead latents (the sustained segment took the dense route, +0.57 ms, on every sixth step for
// that reason).  bound <= 32: one launch; <= 64: two; <= AUX_MFMA_MAX (128): three, and row pitch ndp = AUX_MFMA_MAX.
hipError_t launch_aux_mfma_forward(const float* x, const float* x_hat, const float* WencT_dead, const float* Wdec_dead, const float* b_enc,
                                   const float* b_dec, const int32_t* dl, int n_rows, int D, const int32_t* nd_dev, float gscale, float* A,
                                   float* dA, float* g_aux, RowStats* rowstats, hipStream_t s, int bound, int ndp) {
    if (!aux_mfma_supported(D) || bound > AUX_MFMA_MAX || bound > ndp) return hipErrorInvalidValue;
    const dim3 grid((n_rows + 31) / 32), block(256);
    hipLaunchKernelGGL(aux_mfma_forward_kernel<1>, grid, block, 0, s, x, x_hat, WencT_dead, Wdec_dead, b_enc, b_dec, dl, n_rows, D, nd_dev,
                       gscale, A, dA, g_aux, rowstats, ndp, 1);
    if (bound > 32)
        hipLaunchKernelGGL(aux_mfma_forward_kernel<2>, grid, block, 0, s, x, x_hat, WencT_dead, Wdec_dead, b_enc, b_dec, dl, n_rows, D, nd_dev,
                           gscale, A, dA, g_aux, rowstats, ndp, 33);
    if (bound > 64)
        hipLaunchKernelGGL(aux_mfma_forward_kernel<4>, grid, block, 0, s, x, x_hat, WencT_dead, Wdec_dead, b_enc, b_dec, dl, n_rows, D, nd_dev,
                           gscale, A, dA, g_aux, rowstats, ndp, 65);
    return hipGetLastError();
}
// (the weight gradients are independent per latent block: beyond 64 dead latents the two-block kernel runs once per 64 latents)
hipError_t launch_aux_mfma_wgrad(const float* A, const float* dA, const float* g_aux, const float* x, int n_rows, int D,
                                 const int32_t* nd_dev, float* part, float* partb, float* partbe, hipStream_t s, int bound, int ndp) {
    if (!aux_mfma_supported(D) || bound > AUX_MFMA_MAX || bound > ndp) return hipErrorInvalidValue;
    const dim3 grid((n_rows + 63) / 64, D >= 512 ? 2 : 1), block(256);
    hipLaunchKernelGGL(aux_mfma_wgrad_kernel<1>, grid, block, 0, s, A, dA, g_aux, x, n_rows, D, nd_dev, part, partb, partbe, ndp, 1, 32, 0);
    if (bound > 32)
        hipLaunchKernelGGL(aux_mfma_wgrad_kernel<2>, grid, block, 0, s, A, dA, g_aux, x, n_rows, D, nd_dev, part, partb, partbe, ndp, 33, AUX_MFMA_MAX, 0);
    if (bound > 64)
        hipLaunchKernelGGL(aux_mfma_wgrad_kernel<2>, grid, block, 0, s, A, dA, g_aux, x, n_rows, D, nd_dev, part, nullptr, partbe, ndp, 65, AUX_MFMA_MAX, 64);
    return hipGetLastError();
}
hipError_t launch_aux_small_wgrad(const float* A, const float* dA, const float* g_aux, const float* x, int n_rows, int D,
                                  const int32_t* nd_dev, float* part, hipStream_t s) {
    hipLaunchKernelGGL(aux_small_wgrad_kernel, dim3((n_rows + 63) / 64, AUX_SMALL_MAX / 8), dim3(256), 0, s, A, dA, g_aux, x, n_rows, D,
                       nd_dev, part);
    return hipGetLastError();
}
hipError_t launch_sum_parts(const float* parts, int n_parts, long n, float* out, hipStream_t s) {
    hipLaunchKernelGGL(sum_parts_kernel, dim3(grid_for(n / 4)), dim3(256), 0, s, parts, n_parts, n / 4, out);  // n % 4 == 0
    return hipGetLastError();
}
hipError_t launch_scatter_add_dead(const int32_t* dl, int nd, int D, const float* dWd, const float* dWe, const float* dbe,
                                   float* gW_dec, float* gW_encT, float* gb_enc, int lat_lo, int lat_hi, hipStream_t s,
                                   const int32_t* nd_dev, int part, float2* row_proj, const float* W_dec, int project,
                                   float* enc_sq, int32_t* lat_unused, const int32_t* starts) {
    if (nd <= 0) return hipSuccess;
    hipLaunchKernelGGL(scatter_add_dead_kernel, dim3((nd + 3) / 4), dim3(256), 0, s, dl, nd, D, dWd, dWe, dbe, gW_dec,
                       gW_encT, gb_enc, lat_lo, lat_hi, nd_dev, part, row_proj, W_dec, project, enc_sq, lat_unused, starts);
    return hipGetLastError();
}
