// The k-sparse half of the step: everything the reference does with dense (B,S) x (S,D) GEMMs on a
// matrix that has k non-zeros per row (modeling.py:351-409 decode; autograd backward of it,
// train.py:347-348) done as index-gathered row operations.
//
//   decode_kernel     x_hat = b_dec + sum_j val_j W_dec[idx_j]; scaled MSE (objectives.py:223-237);
//                     g = dL/dx_hat; fired flags; per-row stats.  (dval_j = <W_dec[idx_j], g> is formed in dw_rows_kernel)
//   decode_matry_kernel the same for P nested Matryoshka prefixes (objectives.py:125-138).
//   csc_*             latent-major ordering of the (row, latent) pairs, deterministic (row-ascending
//                     inside a latent) via an S x B bit map: atomicOr fill, per-latent popcounts + group
//                     prefixes, scan, then every code computes its own slot from the bits below it.
//   dw_dec_kernel     dW_dec[i,:] = sum_{b in latent i} val * g[b,:]   (+ db_enc[i] = sum dval)
//   dw_enc_kernel     dW_enc[:,i] = sum_{b in latent i} dval * x[b,:]  (32 latents per workgroup,
//                     transposed through LDS so global stores are 128-byte rows of the (D,S) matrix)
//   dw_slices_*       the same two gradients from 32-column slices of g / x that an XCD's L2 holds (the route of a backward
//                     over all latents; kernels.h: DwSlicesArgs): pass A, dval sums, pass B, finalize
//   colsum            db_dec = sum_b g[b,:]
//
// One wave owns one activation row / one latent; lanes stride the d_model axis in float4s.
#include "common.h"
#include "kernels.h"

namespace {

template <int NV>
__device__ __forceinline__ void gather_rows_accum(f32x4 (&acc)[NV], const float* __restrict__ W, int D, int D4,
                                                  const int32_t* idx_row, const float* val_row, int k, int limit,
                                                  int lane) {
    for (int j0 = 0; j0 < k; j0 += 64) {
        const int cnt = min(64, k - j0);
        int32_t my_i = -1;
        float my_v = 0.f;
        if (lane < cnt) { my_i = idx_row[j0 + lane]; my_v = val_row[j0 + lane]; }
#pragma unroll 4
        for (int jj = 0; jj < cnt; ++jj) {
            const int32_t i = __shfl(my_i, jj, 64);
            const float v = __shfl(my_v, jj, 64);
            if (i < 0 || i >= limit) continue;
            const f32x4* wr = reinterpret_cast<const f32x4*>(W + (size_t)i * D);
#pragma unroll
            for (int n = 0; n < NV; ++n) {
                const int q = lane + 64 * n;
                if (q < D4) {
                    const f32x4 w = wr[q];
                    acc[n] += v * w;
                }
            }
        }
    }
}

// Reduce-scatter of KC per-lane partial sums across a wave: afterwards lane l holds the wave-wide sum of p[(l >> S) &
// (KC-1)] in p[0], S = 6 - log2(KC).  KC + log2(64/KC) shuffles instead of 6 * KC for KC separate wave_sum()s.
template <int KC>
__device__ __forceinline__ float wave_reduce_scatter(float (&p)[KC], int lane) {
    int bit = 32;
#pragma unroll
    for (int h = KC / 2; h >= 1; h >>= 1, bit >>= 1) {
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

// the same with every step a compile-time constant (the loop form above leaves hipcc a dynamically indexed array at KC = 32: a
// 32-way select chain per access)
template <int H, int BIT>
__device__ __forceinline__ void wave_rs_step(float* p, int lane) {
    if constexpr (H >= 1) {
        const bool up = (lane & BIT) != 0;
#pragma unroll
        for (int i = 0; i < H; ++i) {
            const float keep = up ? p[i + H] : p[i];
            const float send = up ? p[i] : p[i + H];
            p[i] = keep + __shfl_xor(send, BIT, 64);
        }
        wave_rs_step<H / 2, BIT / 2>(p, lane);
    }
}
// KC = 32: lane l ends up with the wave-wide sum of p[(l >> 1) & 31]
__device__ __forceinline__ float wave_reduce_scatter32(float (&p)[32], int lane) {
    wave_rs_step<16, 32>(p, lane);
    return p[0] + __shfl_xor(p[0], 1, 64);
}

// bit (latent i, row) of the CSC build's bit map (DecodeArgs::csc_bitmap; csc_fill_body does the same for builds the decode has
// not prepared)
__device__ __forceinline__ void csc_mark(const DecodeArgs& a, int32_t i, int row) {
    if (a.csc_bitmap != nullptr && i >= 0 && i < a.S) atomicOr(&a.csc_bitmap[(size_t)i * a.csc_words + (row >> 5)], 1u << (row & 31));
}
template <int NV>
__global__ __launch_bounds__(256) void decode_kernel(DecodeArgs a) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= a.n_rows) return;
    const int D = a.D, D4 = D >> 2;
    const int32_t* idx_row = a.idx + (size_t)row * a.code_stride;
    const float* val_row = a.val + (size_t)row * a.code_stride;

    f32x4 acc[NV];
#pragma unroll
    for (int n = 0; n < NV; ++n) {
        const int q = lane + 64 * n;
        acc[n] = (q < D4) ? reinterpret_cast<const f32x4*>(a.b_dec)[q] : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    gather_rows_accum<NV>(acc, a.W_dec, D, D4, idx_row, val_row, a.k, a.idx_limit, lane);

    if (a.x == nullptr) {  // reconstruction only (API decode)
#pragma unroll
        for (int n = 0; n < NV; ++n) {
            const int q = lane + 64 * n;
            if (q < D4) reinterpret_cast<f32x4*>(a.x_hat + (size_t)row * D)[q] = acc[n];
        }
        return;
    }
    const float u = a.upper ? fmaxf(*a.upper, 1e-12f) : 1.0f;
    float sse_scaled = 0.f;
    double sse64 = 0.0, sumsq64 = 0.0;
    f32x4 g[NV];
    const f32x4* xr = reinterpret_cast<const f32x4*>(a.x + (size_t)row * D);
#pragma unroll
    for (int n = 0; n < NV; ++n) {
        const int q = lane + 64 * n;
        g[n] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (q < D4) {
            const f32x4 xv = xr[q];
            if (a.x_hat) reinterpret_cast<f32x4*>(a.x_hat + (size_t)row * D)[q] = acc[n];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float t = acc[n][e] / u - xv[e] / u;
                sse_scaled += t * t * u * u;
                g[n][e] = a.gscale * t * u;
                const float r = xv[e] - acc[n][e];
                sse64 += (double)r * (double)r;
                sumsq64 += (double)xv[e] * (double)xv[e];
            }
            if (a.training) reinterpret_cast<f32x4*>(a.g + (size_t)row * D)[q] = g[n];
            if (a.training && a.gS != nullptr) {  // slice-major copies for launch_dw_slices: [q / 8][row][8 float4]
                const size_t o = ((size_t)(q >> 3) * a.n_rows + row) * 8 + (q & 7);
                reinterpret_cast<f32x4*>(a.gS)[o] = g[n];
                if (a.xS != nullptr) reinterpret_cast<f32x4*>(a.xS)[o] = xv;  // (NULL: split_f16r has left it already)
            }
        }
    }
    // (dval_j = <g, W_dec[idx_j]> is formed by dw_rows_kernel, latent-major, from the rows of g it reads anyway: no second
    // gather of the k decoder rows here)
    // code statistics + fired flags
    float l0 = 0.f, l1 = 0.f;
    for (int j = lane; j < a.k; j += 64) {
        const int32_t i = idx_row[j];
        const float v = val_row[j];
        if (i >= 0 && v != 0.f) {
            l0 += 1.f;
            l1 += fabsf(v);
            if (a.training && a.fired) a.fired[i] = 1;
        }
        csc_mark(a, i, row);
    }
    if (a.rowstats) {
        sse_scaled = wave_sum(sse_scaled);
        l0 = wave_sum(l0);
        l1 = wave_sum(l1);
        sse64 = wave_sum_d(sse64);
        sumsq64 = wave_sum_d(sumsq64);
        if (lane == 0) {
            RowStats rs;
            rs.sse_scaled = sse_scaled; rs.l0 = l0; rs.l1 = l1; rs.aux_sse = 0.f;
            rs.sse64 = sse64; rs.sumsq64 = sumsq64;
            a.rowstats[row] = rs;
        }
    }
}

// decode_kernel with the row spread over NW = D / 256 waves (one float4 of the row per lane) and ALL k <= 32 decoder rows of the
// codes held in registers (32 float4 per lane) until dL/dx_hat is known: dval_j = <g, W_dec[idx_j]> then costs no second gather
// and no W_dec slices in the backward's pass A (DwSlicesArgs::have_dval).  x_hat is accumulated in code order, as decode_kernel does.
template <int NW, int KH>  // KH = 1: k <= 32 codes; 2: k <= 64, in two halves of 32 decoder rows
__global__ __launch_bounds__(64 * NW) void decode_q_kernel(DecodeArgs a) {
    __shared__ float sh_dv[NW][32 * KH];
    __shared__ float sh_f[NW];
    __shared__ double sh_d[NW][2];
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int row = blockIdx.x;
    constexpr int D4 = 64 * NW;
    const int q = w * 64 + lane;
    const int32_t* idx_row = a.idx + (size_t)row * a.code_stride;
    const float* val_row = a.val + (size_t)row * a.code_stride;
    int32_t raw_i = -1;
    float raw_v = 0.f;
    if (lane < a.k) { raw_i = idx_row[lane]; raw_v = val_row[lane]; }
    const int32_t my_i = (raw_i < 0 || raw_i >= a.idx_limit) ? -1 : raw_i;
    // (vmcnt counts in order: what the sum starts from is requested before the gathers)
    f32x4 acc = reinterpret_cast<const f32x4*>(a.b_dec)[q];
    const f32x4 xv = reinterpret_cast<const f32x4*>(a.x + (size_t)row * a.D)[q];
    __builtin_amdgcn_sched_barrier(0);
    // all 32 gathers of a half in flight together (an absent code reads row 0 and is not used): buffer loads with the row offset in
    // an SGPR and one lane offset -- written as address arithmetic hipcc forms 64-bit VGPR addresses and sinks the loads into the sum
    // below, eight in flight
    typedef int i32x4_ __attribute__((ext_vector_type(4)));
    const __amdgpu_buffer_rsrc_t wres = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.W_dec), 0, (uint32_t)a.S * (uint32_t)(D4 * 16), 0x00020000);
    const uint32_t voff = (uint32_t)q * 16u;
    f32x4 wv[32];
#pragma unroll
    for (int h = 0; h < KH; ++h) {  // x_hat in code order: half 0, then half 1
#pragma unroll
        for (int j = 0; j < 32; ++j) {
            const int i = __builtin_amdgcn_readlane(my_i, 32 * h + j);
            const i32x4_ t = __builtin_amdgcn_raw_buffer_load_b128(wres, voff, (uint32_t)max(i, 0) * (uint32_t)(D4 * 16), 0);
            wv[j] = f32x4{__int_as_float(t[0]), __int_as_float(t[1]), __int_as_float(t[2]), __int_as_float(t[3])};
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 32; ++j) {
            const int i = __builtin_amdgcn_readlane(my_i, 32 * h + j);
            const float v = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, raw_v), 32 * h + j));
            if (i >= 0) acc += v * wv[j];
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    if (a.x_hat) reinterpret_cast<f32x4*>(a.x_hat + (size_t)row * a.D)[q] = acc;
    const float u = a.upper ? fmaxf(*a.upper, 1e-12f) : 1.0f;
    float sse_scaled = 0.f;
    double sse64 = 0.0, sumsq64 = 0.0;
    f32x4 g;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float t = acc[e] / u - xv[e] / u;
        sse_scaled += t * t * u * u;
        g[e] = a.gscale * t * u;
        const float r = xv[e] - acc[e];
        sse64 += (double)r * (double)r;
        sumsq64 += (double)xv[e] * (double)xv[e];
    }
    if (a.training) {
        reinterpret_cast<f32x4*>(a.g + (size_t)row * a.D)[q] = g;
        if (a.gS != nullptr) {
            const size_t o = ((size_t)(q >> 3) * a.n_rows + row) * 8 + (q & 7);
            reinterpret_cast<f32x4*>(a.gS)[o] = g;
            if (a.xS != nullptr) reinterpret_cast<f32x4*>(a.xS)[o] = xv;
        }
    }
    // dval_j = <g, W_dec[idx_j]>: the rows of the LAST half are still in registers; an earlier half (k > 32) is gathered once more
    // (96 row gathers per activation row instead of the 128 of a decode that forgets the rows plus a backward pass that re-reads them)
#pragma unroll
    for (int hh = 0; hh < KH; ++hh) {
        const int h = KH - 1 - hh;
        if (hh > 0) {
#pragma unroll
            for (int j = 0; j < 32; ++j) {
                const int i = __builtin_amdgcn_readlane(my_i, 32 * h + j);
                const i32x4_ t = __builtin_amdgcn_raw_buffer_load_b128(wres, voff, (uint32_t)max(i, 0) * (uint32_t)(D4 * 16), 0);
                wv[j] = f32x4{__int_as_float(t[0]), __int_as_float(t[1]), __int_as_float(t[2]), __int_as_float(t[3])};
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        float pd[32];
#pragma unroll
        for (int j = 0; j < 32; ++j) pd[j] = (g[0] * wv[j][0] + g[1] * wv[j][1]) + (g[2] * wv[j][2] + g[3] * wv[j][3]);
        const float r = wave_reduce_scatter32(pd, lane);  // lane l: the wave's sum of pd[(l >> 1) & 31]
        if ((lane & 1) == 0) sh_dv[w][32 * h + (lane >> 1)] = r;
    }
    sse_scaled = wave_sum(sse_scaled);
    sse64 = wave_sum_d(sse64);
    sumsq64 = wave_sum_d(sumsq64);
    if (lane == 0) { sh_f[w] = sse_scaled; sh_d[w][0] = sse64; sh_d[w][1] = sumsq64; }
    __syncthreads();
    if (w != 0) return;
    if (lane < 32 * KH) {
        float s = sh_dv[0][lane];
#pragma unroll
        for (int v = 1; v < NW; ++v) s += sh_dv[v][lane];
        if (lane < a.k) a.dval_out[(size_t)row * a.code_stride + lane] = s;
    }
    float l0 = 0.f, l1 = 0.f;
    if (raw_i >= 0 && raw_v != 0.f) {
        l0 = 1.f;
        l1 = fabsf(raw_v);
        if (a.training && a.fired) a.fired[raw_i] = 1;
    }
    csc_mark(a, raw_i, row);
    if (a.rowstats) {
        l0 = wave_sum(l0);
        l1 = wave_sum(l1);
        if (lane == 0) {
            RowStats rs;
            float f = sh_f[0];
            double d0 = sh_d[0][0], d1 = sh_d[0][1];
#pragma unroll
            for (int v = 1; v < NW; ++v) { f += sh_f[v]; d0 += sh_d[v][0]; d1 += sh_d[v][1]; }
            rs.sse_scaled = f; rs.l0 = l0; rs.l1 = l1; rs.aux_sse = 0.f;
            rs.sse64 = d0; rs.sumsq64 = d1;
            a.rowstats[row] = rs;
        }
    }
}

// Matryoshka variant of decode_kernel: P nested reconstructions per row.  Codes are in ascending latent order, so
// one sweep emits prefix p whenever the next code's latent reaches cuts[p].  Writes g_p = dL/dx_hat_p for every
// prefix, turns them into suffix sums C_p (what a code in prefix block p receives from all reconstructions that
// contain it) in place (dw_rows_kernel takes dval_j = <W_dec[idx_j], C_{p(j)}> from them).
template <int NV>
__global__ __launch_bounds__(256) void decode_matry_kernel(DecodeArgs a, MatryArgs m) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= a.n_rows) return;
    const int D = a.D, D4 = D >> 2, P = m.P;
    const int32_t* idx_row = a.idx + (size_t)row * a.code_stride;
    const float* val_row = a.val + (size_t)row * a.code_stride;
    f32x4 acc[NV], xv[NV];
    const f32x4* xr = reinterpret_cast<const f32x4*>(a.x + (size_t)row * D);
#pragma unroll
    for (int n = 0; n < NV; ++n) {
        const int q = lane + 64 * n;
        acc[n] = (q < D4) ? reinterpret_cast<const f32x4*>(a.b_dec)[q] : f32x4{0.f, 0.f, 0.f, 0.f};
        xv[n] = (q < D4) ? xr[q] : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const float u = a.upper ? fmaxf(*a.upper, 1e-12f) : 1.0f;
    float sse_scaled = 0.f;
    double sse64 = 0.0, sumsq64 = 0.0;
    f32x4* Grow = reinterpret_cast<f32x4*>(m.G + (size_t)row * P * D);
    auto emit = [&](int p) {
        const bool last = (p == P - 1);
#pragma unroll
        for (int n = 0; n < NV; ++n) {
            const int q = lane + 64 * n;
            if (q < D4) {
                f32x4 g;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float t = acc[n][e] / u - xv[n][e] / u;
                    sse_scaled += t * t * u * u;
                    g[e] = a.gscale * t * u;
                    if (last) {
                        const float r = xv[n][e] - acc[n][e];
                        sse64 += (double)r * (double)r;
                        sumsq64 += (double)xv[n][e] * (double)xv[n][e];
                    }
                }
                if (a.training) Grow[(size_t)p * D4 + q] = g;
                if (last && a.x_hat) reinterpret_cast<f32x4*>(a.x_hat + (size_t)row * D)[q] = acc[n];
            }
        }
    };
    int p = 0;
    for (int j0 = 0; j0 < a.k; j0 += 64) {
        const int cnt = min(64, a.k - j0);
        int32_t my_i = -1;
        float my_v = 0.f;
        if (lane < cnt) { my_i = idx_row[j0 + lane]; my_v = val_row[j0 + lane]; }
        for (int jj = 0; jj < cnt; ++jj) {
            const int32_t i = __shfl(my_i, jj, 64);
            const float v = __shfl(my_v, jj, 64);
            if (i < 0) continue;
            while (p < P - 1 && i >= m.cuts[p]) { emit(p); ++p; }
            const f32x4* wr = reinterpret_cast<const f32x4*>(a.W_dec + (size_t)i * D);
#pragma unroll
            for (int n = 0; n < NV; ++n) {
                const int q = lane + 64 * n;
                if (q < D4) acc[n] += v * wr[q];
            }
        }
    }
    while (p < P) { emit(p); ++p; }

    if (a.training) {
        // suffix sums in place (each lane re-reads only what it wrote)
        f32x4 c[NV];
#pragma unroll
        for (int n = 0; n < NV; ++n) c[n] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int pp = P - 1; pp >= 0; --pp) {
#pragma unroll
            for (int n = 0; n < NV; ++n) {
                const int q = lane + 64 * n;
                if (q < D4) {
                    c[n] += Grow[(size_t)pp * D4 + q];
                    Grow[(size_t)pp * D4 + q] = c[n];
                    if (a.gS != nullptr) {  // slice-major copies for launch_dw_slices: [q / 8][p][row][8 float4]
                        reinterpret_cast<f32x4*>(a.gS)[(((size_t)(q >> 3) * P + pp) * a.n_rows + row) * 8 + (q & 7)] = c[n];
                        if (pp == 0 && a.xS != nullptr) reinterpret_cast<f32x4*>(a.xS)[((size_t)(q >> 3) * a.n_rows + row) * 8 + (q & 7)] = xv[n];
                    }
                }
            }
        }
        // (dval_j = <W_dec[idx_j], C_{p(j)}> is formed by dw_rows_kernel from these suffix sums)
    }
    float l0 = 0.f, l1 = 0.f;
    for (int j = lane; j < a.k; j += 64) {
        const int32_t i = idx_row[j];
        const float v = val_row[j];
        if (i >= 0 && v != 0.f) {
            l0 += 1.f;
            l1 += fabsf(v);
            if (a.training && a.fired) a.fired[i] = 1;
        }
        csc_mark(a, i, row);
    }
    if (a.rowstats) {
        sse_scaled = wave_sum(sse_scaled);
        l0 = wave_sum(l0);
        l1 = wave_sum(l1);
        sse64 = wave_sum_d(sse64);
        sumsq64 = wave_sum_d(sumsq64);
        if (lane == 0) {
            RowStats rs;
            rs.sse_scaled = sse_scaled; rs.l0 = l0; rs.l1 = l1; rs.aux_sse = 0.f;
            rs.sse64 = sse64; rs.sumsq64 = sumsq64;
            a.rowstats[row] = rs;
        }
    }
}

// decode_matry_kernel in decode_q_kernel's layout: the k <= 32 decoder rows of the codes stay in registers (one float4 per lane and
// code), the P prefix gradients of the lane's four columns in LDS (P x D floats per row = workgroup), so the suffix sums C_p
// never make the round trip through G that the row kernel pays (P x 4 KB written, read and rewritten per row), and
// dval_j = <C_p(j), W_dec[idx_j]> comes from the registers (DwSlicesArgs::have_dval).
template <int NW>
__global__ __launch_bounds__(64 * NW) void decode_matry_q_kernel(DecodeArgs a, MatryArgs m) {
    extern __shared__ __attribute__((aligned(16))) char matry_smem[];
    f32x4* const shG = reinterpret_cast<f32x4*>(matry_smem);  // [P][64 * NW]
    __shared__ float sh_dv[NW][32];
    __shared__ float sh_f[NW];
    __shared__ double sh_d[NW][2];
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int row = blockIdx.x;
    constexpr int D4 = 64 * NW;
    const int q = w * 64 + lane;
    const int P = m.P;
    const int32_t* idx_row = a.idx + (size_t)row * a.code_stride;
    const float* val_row = a.val + (size_t)row * a.code_stride;
    int32_t raw_i = -1;
    float raw_v = 0.f;
    if (lane < a.k) { raw_i = idx_row[lane]; raw_v = val_row[lane]; }
    const int32_t my_i = raw_i < 0 ? -1 : raw_i;
    f32x4 acc = reinterpret_cast<const f32x4*>(a.b_dec)[q];
    const f32x4 xv = reinterpret_cast<const f32x4*>(a.x + (size_t)row * a.D)[q];
    __builtin_amdgcn_sched_barrier(0);
    typedef int i32x4_ __attribute__((ext_vector_type(4)));
    const __amdgpu_buffer_rsrc_t wres = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.W_dec), 0, (uint32_t)a.S * (uint32_t)(D4 * 16), 0x00020000);
    const uint32_t voff = (uint32_t)q * 16u;
    f32x4 wv[32];
#pragma unroll
    for (int j = 0; j < 32; ++j) {
        const int i = __builtin_amdgcn_readlane(my_i, j);
        const i32x4_ t = __builtin_amdgcn_raw_buffer_load_b128(wres, voff, (uint32_t)max(i, 0) * (uint32_t)(D4 * 16), 0);
        wv[j] = f32x4{__int_as_float(t[0]), __int_as_float(t[1]), __int_as_float(t[2]), __int_as_float(t[3])};
    }
    __builtin_amdgcn_sched_barrier(0);
    const float u = a.upper ? fmaxf(*a.upper, 1e-12f) : 1.0f;
    float sse_scaled = 0.f;
    double sse64 = 0.0, sumsq64 = 0.0;
    auto emit = [&](int p) {
        f32x4 g;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float t = acc[e] / u - xv[e] / u;
            sse_scaled += t * t * u * u;
            g[e] = a.gscale * t * u;
        }
        shG[p * D4 + q] = g;
    };
    int p = 0;
#pragma unroll
    for (int j = 0; j < 32; ++j) {
        const int i = __builtin_amdgcn_readlane(my_i, j);
        const float v = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, raw_v), j));
        if (i >= 0) {
            while (p < P - 1 && i >= m.cuts[p]) { emit(p); ++p; }
            acc += v * wv[j];
        }
    }
    while (p < P) { emit(p); ++p; }
#pragma unroll
    for (int e = 0; e < 4; ++e) {  // statistics of the full reconstruction
        const float r = xv[e] - acc[e];
        sse64 += (double)r * (double)r;
        sumsq64 += (double)xv[e] * (double)xv[e];
    }
    if (a.x_hat) reinterpret_cast<f32x4*>(a.x_hat + (size_t)row * a.D)[q] = acc;
    if (a.training) {
        // suffix sums in place (a lane re-reads only what it wrote), out to the slice-major copy [q / 8][p][row][8 float4]
        f32x4 c = {0.f, 0.f, 0.f, 0.f};
        f32x4* const Grow = reinterpret_cast<f32x4*>(m.G + (size_t)row * P * a.D);
        for (int pp = P - 1; pp >= 0; --pp) {
            c += shG[pp * D4 + q];
            shG[pp * D4 + q] = c;
            if (pp == 0 || m.g_rows_all) Grow[(size_t)pp * D4 + q] = c;
            if (a.gS != nullptr) reinterpret_cast<f32x4*>(a.gS)[(((size_t)(q >> 3) * P + pp) * a.n_rows + row) * 8 + (q & 7)] = c;
        }
        if (a.gS != nullptr && a.xS != nullptr) reinterpret_cast<f32x4*>(a.xS)[((size_t)(q >> 3) * a.n_rows + row) * 8 + (q & 7)] = xv;
        float pd[32];
        int pb = 0;
#pragma unroll
        for (int j = 0; j < 32; ++j) {
            const int i = __builtin_amdgcn_readlane(my_i, j);
            pd[j] = 0.f;
            if (i >= 0) {
                while (pb < P - 1 && i >= m.cuts[pb]) ++pb;
                const f32x4 cj = shG[pb * D4 + q];
                pd[j] = (cj[0] * wv[j][0] + cj[1] * wv[j][1]) + (cj[2] * wv[j][2] + cj[3] * wv[j][3]);
            }
        }
        const float r = wave_reduce_scatter32(pd, lane);
        if ((lane & 1) == 0) sh_dv[w][lane >> 1] = r;
    }
    sse_scaled = wave_sum(sse_scaled);
    sse64 = wave_sum_d(sse64);
    sumsq64 = wave_sum_d(sumsq64);
    if (lane == 0) { sh_f[w] = sse_scaled; sh_d[w][0] = sse64; sh_d[w][1] = sumsq64; }
    __syncthreads();
    if (w != 0) return;
    if (a.training && lane < 32) {
        float s = sh_dv[0][lane];
#pragma unroll
        for (int v = 1; v < NW; ++v) s += sh_dv[v][lane];
        if (lane < a.k) a.dval_out[(size_t)row * a.code_stride + lane] = s;
    }
    float l0 = 0.f, l1 = 0.f;
    if (raw_i >= 0 && raw_v != 0.f) {
        l0 = 1.f;
        l1 = fabsf(raw_v);
        if (a.training && a.fired) a.fired[raw_i] = 1;
    }
    csc_mark(a, raw_i, row);
    if (a.rowstats) {
        l0 = wave_sum(l0);
        l1 = wave_sum(l1);
        if (lane == 0) {
            RowStats rs;
            float f = sh_f[0];
            double d0 = sh_d[0][0], d1 = sh_d[0][1];
#pragma unroll
            for (int v = 1; v < NW; ++v) { f += sh_f[v]; d0 += sh_d[v][0]; d1 += sh_d[v][1]; }
            rs.sse_scaled = f; rs.l0 = l0; rs.l1 = l1; rs.aux_sse = 0.f;
            rs.sse64 = d0; rs.sumsq64 = d1;
            a.rowstats[row] = rs;
        }
    }
}

// ------------------------------- CSC build -------------------------------------------------

// zero the bit map; skipped entirely when the (device-side) code count is 0
__global__ __launch_bounds__(256) void csc_clear_kernel(CscArgs a) {
    if (a.k_dev && *a.k_dev <= 0) return;
    const size_t n4 = ((size_t)a.S * a.words) >> 2;  // words is a multiple of 8
    uint4* bm = reinterpret_cast<uint4*>(a.bitmap);
    for (size_t q = (size_t)blockIdx.x * 256 + threadIdx.x; q < n4; q += (size_t)gridDim.x * 256) bm[q] = uint4{0, 0, 0, 0};
}

// bit (latent i, row b) for every code.  No per-latent counter here: a latent that fires on most rows would
// serialise tens of thousands of same-address atomics; counts come from the bit map instead.
__device__ __forceinline__ void csc_fill_body(const CscArgs& a, int bid, int nblk) {
    const int k = a.k_dev ? min(*a.k_dev, a.k) : a.k;
    if (k <= 0) return;
    const long n = (long)a.n_rows * k;
    for (long p = (long)bid * blockDim.x + threadIdx.x; p < n; p += (long)nblk * blockDim.x) {
        const int b = (int)(p / k), j = (int)(p % k);
        const int32_t i = a.idx[(size_t)b * a.code_stride + j];
        if (i >= 0 && i < a.S) atomicOr(&a.bitmap[(size_t)i * a.words + (b >> 5)], 1u << (b & 31));
    }
}
__global__ void csc_fill_kernel(CscArgs a) { csc_fill_body(a, blockIdx.x, gridDim.x); }

// one wave per latent: counts[i] = number of rows that use latent i, and grp_prefix[i][g] = how many of them lie in
// row groups (256 rows = 8 bitmap words) before group g -- the rank of a code inside its latent is then one 2-byte
// and one 32-byte read away (csc_place_kernel).
__device__ __forceinline__ void csc_count_body(const CscArgs& a, int bid) {
    if (a.k_dev && *a.k_dev <= 0) return;
    const int lane = threadIdx.x & 63;
    const int i = bid * 4 + (threadIdx.x >> 6);
    if (i >= a.S) return;
    const int groups = a.words >> 3;
    const uint4* bm = reinterpret_cast<const uint4*>(a.bitmap + (size_t)i * a.words);
    int base = 0;
    for (int g0 = 0; g0 < groups; g0 += 64) {
        const int g = g0 + lane;
        int c = 0;
        if (g < groups) {
            const uint4 lo = bm[2 * g], hi = bm[2 * g + 1];
            c = __popc(lo.x) + __popc(lo.y) + __popc(lo.z) + __popc(lo.w) + __popc(hi.x) + __popc(hi.y) + __popc(hi.z) + __popc(hi.w);
        }
        int incl = c;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int n = __shfl_up(incl, o, 64);
            if (lane >= o) incl += n;
        }
        if (g < groups) a.grp_prefix[(size_t)i * groups + g] = base + incl - c;
        base += __shfl(incl, 63, 64);
    }
    if (lane == 0) a.counts[i] = base;
}
__global__ __launch_bounds__(256) void csc_count_kernel(CscArgs a) { csc_count_body(a, blockIdx.x); }

// exclusive scans over the latents, two small coalesced passes (1024 latents per workgroup, then the block offsets):
//   starts[i]       pair offset            (sum of counts)
//   chunk_starts[i] work-item offset       (sum of max(1, ceil(count / DW_CHUNK)))
//   part_starts[i]  partial-sum slot       (sum of chunks of latents with more than one chunk)
__global__ __launch_bounds__(1024) void csc_scan_block_kernel(CscArgs a) {
    if (a.k_dev && *a.k_dev <= 0) return;
    __shared__ int wave_tot[16][3];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int i = blockIdx.x * 1024 + tid;
    int v[3] = {0, 0, 0};
    if (i < a.S) {
        const int c = a.counts[i];
        const int nch = (c + DW_CHUNK - 1) / DW_CHUNK;
        v[0] = c; v[1] = max(1, nch); v[2] = nch > 1 ? nch : 0;
    }
    int incl[3] = {v[0], v[1], v[2]};
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            const int n = __shfl_up(incl[q], o, 64);
            if (lane >= o) incl[q] += n;
        }
    }
    if (lane == 63)
        for (int q = 0; q < 3; ++q) wave_tot[w][q] = incl[q];
    __syncthreads();
    int off[3] = {0, 0, 0};
    for (int j = 0; j < w; ++j)
        for (int q = 0; q < 3; ++q) off[q] += wave_tot[j][q];
    if (i < a.S) {
        a.starts[i] = off[0] + incl[0] - v[0];
        if (a.chunk_starts) {
            a.chunk_starts[i] = off[1] + incl[1] - v[1];
            a.part_starts[i] = off[2] + incl[2] - v[2];
        }
    }
    if (tid == 1023)
        for (int q = 0; q < 3; ++q) a.scan_totals[blockIdx.x * 3 + q] = off[q] + incl[q];
}
__global__ __launch_bounds__(1024) void csc_scan_offset_kernel(CscArgs a) {
    if (a.k_dev && *a.k_dev <= 0) return;
    __shared__ int base[3];
    const int tid = threadIdx.x, lane = tid & 63;
    if (tid < 64) {  // first wave: sum of the totals of the blocks before this one
        int s3[3] = {0, 0, 0};
        for (int b = lane; b < (int)blockIdx.x; b += 64)
            for (int q = 0; q < 3; ++q) s3[q] += a.scan_totals[b * 3 + q];
        for (int q = 0; q < 3; ++q) s3[q] = wave_sum_i(s3[q]);
        if (lane == 0)
            for (int q = 0; q < 3; ++q) base[q] = s3[q];
    }
    __syncthreads();
    const int i = blockIdx.x * 1024 + tid;
    if (i < a.S) {
        a.starts[i] += base[0];
        if (a.chunk_starts) { a.chunk_starts[i] += base[1]; a.part_starts[i] += base[2]; }
    }
    if (blockIdx.x == gridDim.x - 1 && tid == 0) {
        if (a.zero_word != nullptr) *a.zero_word = 0;
        a.starts[a.S] = base[0] + a.scan_totals[blockIdx.x * 3 + 0];
        if (a.chunk_starts) a.chunk_starts[a.S] = base[1] + a.scan_totals[blockIdx.x * 3 + 1];
    }
}

// one thread per code: its slot inside its latent's list is the number of lower rows that use the same latent
// (row-ascending order, so the weight-gradient sums are deterministic) = group prefix + popcount of the bits below
// it inside its 256-row group.  No search, no per-latent serial work: a latent that fires on every row costs the
// same per code as one that fires once.  The first S threads also label the work items of "their" latent.
__global__ void csc_place_kernel(CscArgs a) {
    const int k = a.k_dev ? min(*a.k_dev, a.k) : a.k;
    if (k <= 0) return;
    const long n = (long)a.n_rows * k;
    const int groups = a.words >> 3;
    for (long p = (long)blockIdx.x * blockDim.x + threadIdx.x; p < max(n, (long)a.S); p += (long)gridDim.x * blockDim.x) {
        if (p < a.S && a.chunk_starts) {
            const int c1 = a.chunk_starts[p + 1];
            for (int c = a.chunk_starts[p]; c < c1; ++c) a.work_latent[c] = (int)p;
        }
        if (p >= n) continue;
        const int b = (int)(p / k), j = (int)(p % k);
        const int32_t i = a.idx[(size_t)b * a.code_stride + j];
        if (i < 0 || i >= a.S) continue;
        const int g = b >> 8, wi = (b >> 5) & 7;
        const uint4* grp = reinterpret_cast<const uint4*>(a.bitmap + (size_t)i * a.words + 8 * g);
        const uint4 lo = grp[0], hi = grp[1];
        const uint32_t wd[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
        int rank = a.grp_prefix[(size_t)i * groups + g];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const uint32_t m = e < wi ? 0xffffffffu : (e == wi ? ((1u << (b & 31)) - 1u) : 0u);
            rank += __popc(wd[e] & m);
        }
        const int slot = a.starts[i] + rank;
        if (a.pairs != nullptr) a.pairs[slot] = int2{b, (int)((size_t)b * a.code_stride + j)};  // (read by dw_rows_kernel alone)
        if (a.pv != nullptr) {
            const int fl = (rank == 0 ? DWS_FIRST : 0) | (rank == a.counts[i] - 1 ? DWS_LAST : 0);
            int pblk = 0;  // Matryoshka: the latent's prefix block selects which suffix sum its pairs read
            if (a.P > 1) while (pblk < a.P - 1 && i >= a.cuts[pblk]) ++pblk;
            a.pv[slot] = int2{((pblk * a.n_rows + b) << 7) | fl, __float_as_int(a.val[(size_t)b * a.code_stride + j])};
            a.plat[slot] = i;
            if (a.pv2 != nullptr) a.pv2[slot] = int2{(b << 7) | fl, __float_as_int(a.dval[(size_t)b * a.code_stride + j])};
        }
    }
}

// ------------------------------- weight gradients ------------------------------------------
//
// Work is cut into chunks of <= DW_CHUNK pairs of one latent so that dense latents (which fire on a
// large share of the batch) do not serialise on one wave.  A chunk wave accumulates BOTH
//     dec[:] += val  * g[b,:]      (row i of dW_dec)
//     enc[:] += dval * x[b,:]      (row i of dW_enc^T, transposed into the (D,S) layout afterwards)
// Single-chunk latents write their rows directly; multi-chunk latents write per-chunk partials that
// dw_combine_kernel adds up in chunk order (deterministic).

// {sc, sq} of a decoder-gradient row g held across a wave (DwRowsArgs::row_proj), exactly as rpg_kernel would form them
template <int NV>
__device__ __forceinline__ void write_row_proj(float2* row_proj, int i, const f32x4 (&g)[NV], const f32x4 (&w)[NV], int project,
                                               int lane) {
    float sc;
    const float sq = rpg_row_stats<NV>(g, w, project, &sc);
    if (lane == 0) row_proj[i] = float2{sc, sq};
}

// sum of squares of a row held across a wave (DwRowsArgs::enc_sq)
template <int NV>
__device__ __forceinline__ float row_sumsq(const f32x4 (&g)[NV]) {
    float sq = 0.f;
#pragma unroll
    for (int n = 0; n < NV; ++n)
#pragma unroll
        for (int e = 0; e < 4; ++e) sq = __builtin_fmaf(g[n][e], g[n][e], sq);
    return wave_sum(sq);
}

#ifndef DW_MIN_WAVES
#define DW_MIN_WAVES 1
#endif
#ifndef DW_GROUP
#define DW_GROUP 8  // (row, latent) pairs whose g / x rows one wave has in flight
#endif
template <int NV>
__global__ __launch_bounds__(256, DW_MIN_WAVES) void dw_rows_kernel(DwRowsArgs a) {
    if (a.k_dev && *a.k_dev <= 0) return;
    const int lane = threadIdx.x & 63;
    // work items are latent-major: those of latents [lat_lo, lat_hi) are the contiguous range below
    const int wi = a.chunk_starts[a.lat_lo] + blockIdx.x * 4 + (threadIdx.x >> 6);
    const int n_work = a.chunk_starts[a.lat_hi];
    if (wi >= n_work) return;
    const int i = a.work_latent[wi];
    const int c = wi - a.chunk_starts[i];
    const int nch = a.chunk_starts[i + 1] - a.chunk_starts[i];
    const int D = a.D, D4 = D >> 2;
    const int seg_beg = a.starts[i], seg_end = a.starts[i + 1];
    // Matryoshka: latent i sits in prefix block p(i) and receives the suffix-summed gradient C_{p(i)} (row stride P*D)
    int pblk = 0;
    if (a.P > 1) while (pblk < a.P - 1 && i >= a.cuts[pblk]) ++pblk;
    const size_t g_stride = (size_t)a.P * D;
    const float* gbase = a.g + (size_t)pblk * D;
    const int beg = seg_beg + c * DW_CHUNK;
    const int end = min(seg_end, beg + DW_CHUNK);
    f32x4 accd[NV], acce[NV], wv[NV];
    {
        // this latent's decoder row: dval = <g_b, W_dec[i]> is formed here, from the very rows of g that dW_dec needs anyway
        // (the decode kernel used to gather the row's k decoder rows a second time for it)
        const f32x4* wr = reinterpret_cast<const f32x4*>(a.W_dec + (size_t)i * D);
#pragma unroll
        for (int n = 0; n < NV; ++n) {
            accd[n] = f32x4{0.f, 0.f, 0.f, 0.f};
            acce[n] = f32x4{0.f, 0.f, 0.f, 0.f};
            wv[n] = (a.part != 2 && lane + 64 * n < D4) ? wr[lane + 64 * n] : f32x4{0.f, 0.f, 0.f, 0.f};  // (pass 2 has dval already)
        }
    }
    int my_b = 0;
    float my_v = 0.f, my_dv = 0.f;
    const int cnt = end - beg;  // 0..64
    // a.part: 0 = both gradients in one pass; 1 = the decoder's only (g rows; the dot products dval go to a.dval);
    // 2 = the encoder's only (x rows, weighted with the stored dval).  Two passes move the same bytes as one; a
    // data-parallel caller starts exchanging the decoder half while the second pass runs (framework/ddp.py).
    const int part = a.part;
    if (lane < cnt) {
        const int2 pr = a.pairs[beg + lane];
        my_b = pr.x;
        my_v = a.val[pr.y];
        if (part == 2) my_dv = a.dval[beg + lane];
        // the CSC bit map is of no use once the pairs are placed: zero the words that hold this chunk's bits (B k scattered
        // 4-byte stores, ~2 MB, instead of a 67 MB clearing pass in front of the next step's build)
        if (a.clear_bitmap != nullptr) a.clear_bitmap[(size_t)i * a.clear_words + (my_b >> 5)] = 0u;
    }
    float dbs = 0.f;  // sum of dval over this chunk (wave-uniform)
    constexpr int G = DW_GROUP, GS = (G == 8 ? 3 : 4);  // pairs per trip; lanes per slot of the reduce-scatter = 64 / G
    for (int j0 = 0; j0 < cnt; j0 += G) {
        // (1) G rows of g: dW_dec accumulation + per-lane shares of the eight dot products
        float p[G];
        float r = 0.f;
        if (part != 2) {
#pragma unroll
        for (int t = 0; t < G; t += 2) {
            p[t] = 0.f; p[t + 1] = 0.f;
            if (j0 + t >= cnt) continue;  // uniform
            const bool two = j0 + t + 1 < cnt;
            const int b0 = __shfl(my_b, j0 + t, 64), b1 = __shfl(my_b, min(j0 + t + 1, 63), 64);
            const float v0 = __shfl(my_v, j0 + t, 64), v1 = two ? __shfl(my_v, min(j0 + t + 1, 63), 64) : 0.f;
            const f32x4* g0 = reinterpret_cast<const f32x4*>(gbase + (size_t)b0 * g_stride);
            const f32x4* g1 = reinterpret_cast<const f32x4*>(gbase + (size_t)(two ? b1 : b0) * g_stride);
            f32x4 tg0[NV], tg1[NV];
#pragma unroll
            for (int n = 0; n < NV; ++n) {
                const int q = lane + 64 * n;
                const bool ok = q < D4;
                tg0[n] = ok ? g0[q] : f32x4{0.f, 0.f, 0.f, 0.f};
                tg1[n] = ok ? g1[q] : f32x4{0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int n = 0; n < NV; ++n) {
                accd[n] += v0 * tg0[n];
                accd[n] += v1 * tg1[n];
                p[t] += tg0[n][0] * wv[n][0] + tg0[n][1] * wv[n][1] + tg0[n][2] * wv[n][2] + tg0[n][3] * wv[n][3];
                p[t + 1] += tg1[n][0] * wv[n][0] + tg1[n][1] * wv[n][1] + tg1[n][2] * wv[n][2] + tg1[n][3] * wv[n][3];
            }
            if (!two) p[t + 1] = 0.f;
        }
        r = wave_reduce_scatter<G>(p, lane);  // lane l holds the dot product of entry j0 + ((l >> GS) & (G - 1))
        if (part == 1) {
            if ((lane & ((1 << GS) - 1)) == 0 && j0 + (lane >> GS) < cnt) a.dval[beg + j0 + (lane >> GS)] = r;
            continue;
        }
        }
        // (2) the same eight rows of x, weighted with the dot products just formed
#pragma unroll
        for (int t = 0; t < G; t += 2) {
            if (j0 + t >= cnt) continue;
            const bool two = j0 + t + 1 < cnt;
            const float e0 = part == 2 ? __shfl(my_dv, j0 + t, 64) : __shfl(r, t << GS, 64);
            const float e1 = !two ? 0.f : part == 2 ? __shfl(my_dv, min(j0 + t + 1, 63), 64) : __shfl(r, (t + 1) << GS, 64);
            dbs += e0 + e1;
            const int b0 = __shfl(my_b, j0 + t, 64), b1 = __shfl(my_b, min(j0 + t + 1, 63), 64);
            const f32x4* x0 = reinterpret_cast<const f32x4*>(a.x + (size_t)b0 * D);
            const f32x4* x1 = reinterpret_cast<const f32x4*>(a.x + (size_t)(two ? b1 : b0) * D);
            f32x4 tx0[NV], tx1[NV];
#pragma unroll
            for (int n = 0; n < NV; ++n) {
                const int q = lane + 64 * n;
                const bool ok = q < D4;
                tx0[n] = ok ? x0[q] : f32x4{0.f, 0.f, 0.f, 0.f};
                tx1[n] = ok ? x1[q] : f32x4{0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int n = 0; n < NV; ++n) {
                acce[n] += e0 * tx0[n];
                acce[n] += e1 * tx1[n];
            }
        }
    }

    float *od, *oe;
    bool direct = (nch == 1);
    if (direct) {
        od = a.dW_dec + (size_t)i * D;
        oe = a.dW_encT + (size_t)i * D;
    } else {
        od = a.partials + (size_t)(a.part_starts[i] + c) * 2 * D;
        oe = od + D;
    }
#pragma unroll
    for (int n = 0; n < NV; ++n) {
        const int q = lane + 64 * n;
        if (q < D4) {
            f32x4* pd = reinterpret_cast<f32x4*>(od) + q;
            f32x4* pe = reinterpret_cast<f32x4*>(oe) + q;
            if (part != 2) { if (direct && a.accumulate) *pd = *pd + accd[n]; else *pd = accd[n]; }
            if (part != 1) { if (direct && a.accumulate) *pe = *pe + acce[n]; else *pe = acce[n]; }
        }
    }
    if (lane == 0 && part != 1) {
        if (direct) a.db_enc[i] = a.accumulate ? (a.db_enc[i] + dbs) : dbs;
        else a.db_partials[a.part_starts[i] + c] = dbs;
    }
    // the row just written is final (single-chunk latent): its projection coefficient and projected squares for the tail
    if (a.row_proj != nullptr && direct && part != 2 && !a.accumulate) write_row_proj<NV>(a.row_proj, i, accd, wv, a.project, lane);
    if (a.enc_sq != nullptr && direct && part != 1 && !a.accumulate) {
        const float sq = row_sumsq<NV>(acce);
        if (lane == 0) a.enc_sq[i] = sq;
    }
}

// One wave sums the per-chunk partials of ONE of a latent's two gradient rows (blockIdx.y: 0 decoder row, 1 encoder row +
// db_enc): a latent that fires on most rows has hundreds of chunks (the batch-mean direction of a trained dictionary fires
// on every row: 256 chunks at 16 384 rows), and one wave walking all of its 8 KB partial pairs was a 90 us serial tail of the
// sustained step.  Half the bytes per wave and eight partial rows in flight per trip; the summation order is fixed (chunk
// order inside a trip, trips in order), so the result does not depend on scheduling.
template <int NV>
__global__ __launch_bounds__(256) void dw_combine_kernel(DwRowsArgs a) {
    if (a.k_dev && *a.k_dev <= 0) return;
    const int lane = threadIdx.x & 63;
    const int i = a.lat_lo + blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= a.lat_hi) return;
    const int enc = blockIdx.y;  // 0: dW_dec row, 1: dW_enc^T row and db_enc
    if ((enc == 0 && a.part == 2) || (enc == 1 && a.part == 1)) return;  // (the other pass of a two-pass backward owns that row)
    const int nch = a.chunk_starts[i + 1] - a.chunk_starts[i];
    if (nch <= 1) return;
    const int c0 = a.part_starts[i], c1 = c0 + nch;
    const int D = a.D, D4 = D >> 2;
    const float* base = a.partials + (enc ? D : 0);
    f32x4 acc[NV];
#pragma unroll
    for (int n = 0; n < NV; ++n) acc[n] = f32x4{0.f, 0.f, 0.f, 0.f};
    float dbs = 0.f;
    int c = c0;
    constexpr int TRIP = NV <= 4 ? 8 : (NV <= 8 ? 4 : 2);  // partial rows in flight (TRIP * NV float4 temporaries)
    for (; c + TRIP <= c1; c += TRIP) {
        f32x4 t[TRIP][NV];
#pragma unroll
        for (int u = 0; u < TRIP; ++u) {
            const f32x4* p = reinterpret_cast<const f32x4*>(base + (size_t)(c + u) * 2 * D);
#pragma unroll
            for (int n = 0; n < NV; ++n) t[u][n] = (lane + 64 * n < D4) ? p[lane + 64 * n] : f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int n = 0; n < NV; ++n) {
            if constexpr (TRIP == 8) acc[n] += ((t[0][n] + t[1][n]) + (t[2][n] + t[3][n])) + ((t[4][n] + t[5][n]) + (t[6][n] + t[7][n]));
            else if constexpr (TRIP == 4) acc[n] += (t[0][n] + t[1][n]) + (t[2][n] + t[3][n]);
            else acc[n] += t[0][n] + t[1][n];
        }
        if (enc) {
#pragma unroll
            for (int u = 0; u < TRIP; ++u) dbs += a.db_partials[c + u];
        }
    }
    for (; c < c1; ++c) {
        const f32x4* p = reinterpret_cast<const f32x4*>(base + (size_t)c * 2 * D);
#pragma unroll
        for (int n = 0; n < NV; ++n)
            if (lane + 64 * n < D4) acc[n] += p[lane + 64 * n];
        if (enc) dbs += a.db_partials[c];
    }
    f32x4* o = reinterpret_cast<f32x4*>((enc ? a.dW_encT : a.dW_dec) + (size_t)i * D);
#pragma unroll
    for (int n = 0; n < NV; ++n) {
        const int q = lane + 64 * n;
        if (q < D4) o[q] = a.accumulate ? o[q] + acc[n] : acc[n];
    }
    if (enc) {
        if (lane == 0) a.db_enc[i] = a.accumulate ? (a.db_enc[i] + dbs) : dbs;
        if (a.enc_sq != nullptr && !a.accumulate) {
            const float sq = row_sumsq<NV>(acc);
            if (lane == 0) a.enc_sq[i] = sq;
        }
    } else if (a.row_proj != nullptr && !a.accumulate) {
        f32x4 wv[NV];
        const f32x4* wr = reinterpret_cast<const f32x4*>(a.W_dec + (size_t)i * D);
#pragma unroll
        for (int n = 0; n < NV; ++n) wv[n] = (lane + 64 * n < D4) ? wr[lane + 64 * n] : f32x4{0.f, 0.f, 0.f, 0.f};
        write_row_proj<NV>(a.row_proj, i, acc, wv, a.project, lane);
    }
}

// ------------------------------- weight gradients from column slices (kernels.h: DwSlicesArgs) ----------------------

typedef int i32x2 __attribute__((ext_vector_type(2)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f32x4 buf_load16(__amdgpu_buffer_rsrc_t r, uint32_t voff) {
    const i32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, voff, 0, 0);
    return f32x4{__int_as_float(v[0]), __int_as_float(v[1]), __int_as_float(v[2]), __int_as_float(v[3])};
}
// sum over the eight lanes of a group (every lane receives it)
__device__ __forceinline__ float group8_sum(float v) {
    v += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0xB1, 0xF, 0xF, true));   // quad_perm [1,0,3,2]
    v += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x4E, 0xF, 0xF, true));   // quad_perm [2,3,0,1]
    v += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x141, 0xF, 0xF, true));  // row_half_mirror
    return v;
}

// slice-major copies of two row-major (n, D) matrices (a gathered backward: the rows of all ranks arrive row-major): one wave per row
__global__ __launch_bounds__(256) void slice_major_copy_kernel(const float* __restrict__ g, const float* __restrict__ x, int n, int D,
                                                               float* __restrict__ gS, float* __restrict__ xS) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= n) return;
    const f32x4* gr = reinterpret_cast<const f32x4*>(g + (size_t)row * D);
    const f32x4* xr = reinterpret_cast<const f32x4*>(x + (size_t)row * D);
    for (int q = lane; q < (D >> 2); q += 64) {
        const size_t o = ((size_t)(q >> 3) * n + row) * 8 + (q & 7);
        reinterpret_cast<f32x4*>(gS)[o] = gr[q];
        reinterpret_cast<f32x4*>(xS)[o] = xr[q];
    }
}

// DEC (pass A): m = gS, out = dW_dec, coefficients pv[].y = val, W slices for the dval shares (DVAL).  Otherwise m = xS, out = dW_enc^T,
// coefficients pv2[].y = dval.  A workgroup = 4 waves x 8 lane groups = 32 runs of one slice.
template <bool DEC, bool DVAL = true>
__global__ __launch_bounds__(256, (DEC && DVAL) ? 4 : 5) void dw_slices_kernel(DwSlicesArgs a, int wg_per_slice) {  // (pass B at six waves spilled once it kept its squares)
    constexpr bool PASS_A = DEC && DVAL;  // the dval shares are formed here (DVAL = false: the decode has left them, DwSlicesArgs::have_dval)
    constexpr int L = DWS_RUN;
    const int lane = threadIdx.x & 63, gi = lane >> 3, li = lane & 7;
    const int xcd = blockIdx.x & 7, q = blockIdx.x >> 3;
    const int slice = xcd + 8 * (q / wg_per_slice);
    // (the wave's slot of DwSlicesArgs::sq_wave_dec / _enc: an SGPR pair, formed where it is used -- registers are what this kernel lives on)
    const bool sq_on = (DEC ? a.sq_wave_dec : a.sq_wave_enc) != nullptr;
    auto sq_slot = [&]() -> float* {
        return (DEC ? a.sq_wave_dec : a.sq_wave_enc) + ((size_t)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)));
    };
    if (slice * DWS_SLICE >= a.D) { if (sq_on && lane == 0) *sq_slot() = 0.f; return; }
    const int NP = a.starts[a.S];
    const int run0 = ((q % wg_per_slice) * 4 + (threadIdx.x >> 6)) * 8;
    if (run0 * L >= NP) { if (sq_on && lane == 0) *sq_slot() = 0.f; return; }  // (wave-uniform)
    const int run = run0 + gi;
    const int col = slice * DWS_SLICE + li * 4;
    const uint32_t colb = (uint32_t)col * 4u, rowb = (uint32_t)a.D * 4u, li16 = (uint32_t)li * 16u;
    // run boundaries: nominally r * L; when the latent that holds that pair is short (< L pairs) the boundary moves back to the
    // latent's first pair, so that only latents of L or more pairs are ever cut (runs of 2 .. 2 L - 2 pairs)
    auto bound = [&](int r) -> int {
        const long nb = (long)r * L;
        if (nb >= NP) return NP;
        const int lat = a.plat[nb];
        const int s = a.starts[lat];
        return (a.starts[lat + 1] - s < L) ? s : (int)nb;
    };
    const int p0 = bound(run), p1 = bound(run + 1);
    const bool live = p0 < p1;
    const int2* const pv = DEC ? a.pv : a.pv2;
    float* const out = DEC ? a.dW_dec : a.dW_encT;
    float* const part = DEC ? a.part_dec : a.part_enc;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    f32x4 w4 = {0.f, 0.f, 0.f, 0.f};
    float* const dvp = PASS_A ? a.dvp + (size_t)slice * a.pair_cap : nullptr;
    // (squares of the pieces this lane stores whole, DwSlicesArgs::sq_wave_dec: kept in LDS -- one more live register made pass B
    // spill at its six waves per SIMD, 170 -> 224 us)
    __shared__ float sh_sq[256];
    sh_sq[threadIdx.x] = 0.f;
    const int sel = (lane & 56) << 2;  // byte address of the group's lane 0 for ds_bpermute
    const __amdgpu_buffer_rsrc_t mres = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(DEC ? a.gS : a.xS) + (size_t)slice * a.n_rows * DWS_SLICE * (DEC && a.P > 1 ? a.P : 1), 0,
        (uint32_t)a.n_rows * 128u * (uint32_t)(DEC && a.P > 1 ? a.P : 1), 0x00020000);
    const __amdgpu_buffer_rsrc_t wres = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.W_dec), 0, (uint32_t)a.S * rowb, 0x00020000);

    // pair info of block t (pairs p0 + 8 t + li): END on the run's last pair; past the end: row 0 with coefficient 0 (never stored)
    auto load_info = [&](int t, int2& e, int& lat) {
        const int p = p0 + 8 * t + li;
        e = int2{0, 0};
        lat = 0;
        if (p < p1) {
            const i32x2 v = reinterpret_cast<const i32x2*>(pv)[p];
            e = int2{v[0], v[1]};
            lat = a.plat[p];
            if (p == p1 - 1) e.x |= DWS_END;
            if (p == p0) e.x &= ~DWS_FIRST;  // (its W slice is loaded below)
        }
    };
    constexpr int PB = 4;  // pairs per sub-block: gathers are issued one sub-block ahead of their use
    auto issue = [&](const int2& e, int lat, int j0, int (&xj)[PB], int (&lj)[PB], f32x4 (&gt)[PB], f32x4 (&wt)[PB]) {
#pragma unroll
        for (int j = 0; j < PB; ++j) {
            xj[j] = __builtin_amdgcn_ds_bpermute(sel + 4 * (j0 + j), e.x);
            lj[j] = __builtin_amdgcn_ds_bpermute(sel + 4 * (j0 + j), lat);
            gt[j] = buf_load16(mres, ((uint32_t)xj[j] & ~127u) | li16);
            if (PASS_A && (xj[j] & DWS_FIRST)) wt[j] = buf_load16(wres, (uint32_t)lj[j] * rowb + colb);
        }
    };
    // consume a sub-block; wnext0 = the W slice preloaded for the first pair of the NEXT sub-block
    auto consume = [&](const int2& e, int j0, const int (&xc)[PB], const int (&lc)[PB], const f32x4 (&gc)[PB], const f32x4 (&wc)[PB],
                       const f32x4& wnext0, bool& head_open, float& dmine) {
        float vc[PB];
#pragma unroll
        for (int j = 0; j < PB; ++j) vc[j] = __int_as_float(__builtin_amdgcn_ds_bpermute(sel + 4 * (j0 + j), e.y));
#pragma unroll
        for (int j = 0; j < PB; ++j) {
            if (PASS_A) {
                float d = gc[j][0] * w4[0] + gc[j][1] * w4[1] + gc[j][2] * w4[2] + gc[j][3] * w4[3];
                d = group8_sum(d);
                dmine = li == j0 + j ? d : dmine;
            }
            acc += vc[j] * gc[j];
            if (xc[j] & (DWS_LAST | DWS_END)) {
                float* o;
                if (head_open) o = part + ((size_t)run * 2 + 0) * a.D + col;                   // began in an earlier run
                else if (!(xc[j] & DWS_LAST)) o = part + ((size_t)run * 2 + 1) * a.D + col;    // continues in the next run
                else {                                                                         // the whole latent lies in this run
                    o = out + (size_t)lc[j] * a.D + col;
                    if (sq_on)
                        sh_sq[threadIdx.x] += __builtin_fmaf(acc[3], acc[3], __builtin_fmaf(acc[2], acc[2], __builtin_fmaf(acc[1], acc[1], acc[0] * acc[0])));
                }
                *reinterpret_cast<f32x4*>(o) = acc;
                // every run leaves word of whether a latent BEGINS in it and continues past its end (its tail partial):
                // dw_finalize_cut_kernel starts from these
                if (DEC && slice == 0 && li == 0 && (xc[j] & DWS_END)) {
                    const int cl = (!head_open && !(xc[j] & DWS_LAST)) ? lc[j] : -1;
                    a.cut_lat[run] = cl;
                    if (cl >= 0 && a.cut_list != nullptr)  // {run, latent, its pair range}: the finalize needs no further look-up
                        reinterpret_cast<i32x4*>(a.cut_list)[1 + atomicAdd(&a.cut_list[0], 1)] = i32x4{run, cl, a.starts[cl], a.starts[cl + 1]};
                }
                head_open = false;
                acc = f32x4{0.f, 0.f, 0.f, 0.f};
                if (PASS_A) w4 = j + 1 < PB ? wc[j + 1 < PB ? j + 1 : 0] : wnext0;  // (the next pair is the first of its latent)
            }
        }
    };
    int2 e_c, e_n;
    int lat_c, lat_n;
    int xa[PB], xb[PB], la[PB], lb[PB];
    f32x4 ga[PB], gb[PB], wa[PB], wb[PB];
    bool head_open = false;
    load_info(0, e_c, lat_c);
    if (live) {
        head_open = (pv[p0].x & DWS_FIRST) == 0;  // the run's first latent began in an earlier run
        if (PASS_A) w4 = buf_load16(wres, (uint32_t)a.plat[p0] * rowb + colb);
    }
    issue(e_c, lat_c, 0, xa, la, ga, wa);
    load_info(1, e_n, lat_n);
#pragma unroll 1
    for (int t = 0; __any(p0 + 8 * t < p1); ++t) {  // (every group of the wave runs the same trip count)
        float dmine = 0.f;
        issue(e_c, lat_c, PB, xb, lb, gb, wb);
        consume(e_c, 0, xa, la, ga, wa, wb[0], head_open, dmine);
        issue(e_n, lat_n, 0, xa, la, ga, wa);  // (past the last block: row 0, coefficient 0, in bounds)
        consume(e_c, PB, xb, lb, gb, wb, wa[0], head_open, dmine);
        if (PASS_A && p0 + 8 * t + li < p1) dvp[p0 + 8 * t + li] = dmine;
        e_c = e_n; lat_c = lat_n;
        load_info(t + 2, e_n, lat_n);
    }
    if (sq_on) {
        const float sq = wave_sum(sh_sq[threadIdx.x]);
        if (lane == 0) *sq_slot() = sq;
    }
}

// dval[p] = the D / 32 shares in slice order; pv2 = pv with dval as the coefficient; the CSC bit map word of the pair zeroed
__global__ __launch_bounds__(256) void dw_dval_sum_kernel(DwSlicesArgs a) {
    const int NP = a.starts[a.S];
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= NP) return;
    const int n_slices = a.D / DWS_SLICE;
    float s = 0.f;
    for (int c = 0; c < n_slices; ++c) s += a.dvp[(size_t)c * a.pair_cap + p];
    const int2 e = a.pv[p];
    const int b = a.P > 1 ? (e.x >> 7) % a.n_rows : (e.x >> 7);  // (Matryoshka: pass A's word holds p(latent) * n_rows + row)
    a.pv2[p] = int2{(b << 7) | (e.x & 127), __float_as_int(s)};
    if (a.clear_bitmap != nullptr) a.clear_bitmap[(size_t)a.plat[p] * a.clear_words + (b >> 5)] = 0u;
}

// have_dval: pv2 is there already; only the CSC bit map words of the pairs are zeroed (what dw_dval_sum_kernel does on its way)
__global__ __launch_bounds__(256) void dw_clear_bitmap_kernel(DwSlicesArgs a) {
    const int NP = a.starts[a.S];
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= NP) return;
    a.clear_bitmap[(size_t)a.plat[p] * a.clear_words + (a.pv2[p].x >> 12)] = 0u;
}

// Latents that are cut by run boundaries (L or more pairs): one workgroup per run boundary and gradient row (blockIdx.y: 0 decoder
// row, 1 encoder row + db_enc); the workgroup at the FIRST boundary a latent crosses owns it.  Its four waves sum a quarter of the
// head partials each (run order, eight rows in flight); wave 0 adds the tail partial of the first run and the four sums in wave
// order, writes the row and its statistics.  The order is fixed, so the result does not depend on scheduling; a latent that
// fires on every row (256 partials at 16 384 rows) takes a quarter of the time one wave would.
template <int NV>
__global__ __launch_bounds__(256) void dw_finalize_cut_kernel(DwSlicesArgs a, int kind0) {
    constexpr int L = DWS_RUN;
    __shared__ f32x4 sh[3][NV * 64];  // (waves 1-3; wave 0 keeps its sum in registers)
    __shared__ float shdb[4];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int NP = a.starts[a.S];
    const int r0 = blockIdx.x;
    if ((long)(r0 + 1) * L >= NP) return;
    const int i = a.cut_lat[r0];  // (written by pass A for every run that holds pairs)
    if (i < 0) return;            // no latent is first cut at this boundary
    const int s = a.starts[i], e = a.starts[i + 1];
    const int r1 = (e - 1) / L;
    const int enc = blockIdx.y + kind0;
    const int D = a.D, D4 = D >> 2;
    const float* const part = enc ? a.part_enc : a.part_dec;
    const int nh = r1 - r0, ch = (nh + 3) / 4;
    const int ra = r0 + 1 + w * ch, rb = min(r1 + 1, ra + ch);
    f32x4 acc[NV];
#pragma unroll
    for (int n = 0; n < NV; ++n) acc[n] = f32x4{0.f, 0.f, 0.f, 0.f};
    int r = ra;
    constexpr int TRIP = NV <= 4 ? 4 : 2;  // (partial rows in flight: registers decide how many of these workgroups a CU holds)
    for (; r + TRIP <= rb; r += TRIP) {
        f32x4 t[TRIP][NV];
#pragma unroll
        for (int u = 0; u < TRIP; ++u) {
            const f32x4* p = reinterpret_cast<const f32x4*>(part + (size_t)(r + u) * 2 * D);
#pragma unroll
            for (int n = 0; n < NV; ++n) t[u][n] = (lane + 64 * n < D4) ? p[lane + 64 * n] : f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int n = 0; n < NV; ++n) {
            if constexpr (TRIP == 4) acc[n] += (t[0][n] + t[1][n]) + (t[2][n] + t[3][n]);
            else acc[n] += t[0][n] + t[1][n];
        }
    }
    for (; r < rb; ++r) {
        const f32x4* p = reinterpret_cast<const f32x4*>(part + (size_t)r * 2 * D);
#pragma unroll
        for (int n = 0; n < NV; ++n)
            if (lane + 64 * n < D4) acc[n] += p[lane + 64 * n];
    }
    float dbs = 0.f;
    if (enc) {  // db_enc: a quarter of the latent's pairs per wave (eight loads in flight, added in pair order)
        const int cnt = e - s, q4 = (cnt + 3) / 4;
        const int pend = min(e, s + (w + 1) * q4);
        for (int p0 = s + w * q4 + lane; p0 < pend; p0 += 64 * 8) {
            float t[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) t[u] = (p0 + 64 * u < pend) ? __int_as_float(a.pv2[p0 + 64 * u].y) : 0.f;
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (p0 + 64 * u < pend) dbs += t[u];
        }
        dbs = wave_sum(dbs);
    }
    if (w != 0) {
#pragma unroll
        for (int n = 0; n < NV; ++n) sh[w - 1][lane + 64 * n] = acc[n];
    }
    if (lane == 0) shdb[w] = dbs;
    __syncthreads();
    if (w != 0) return;
    {
        // (a latent that begins exactly at a run boundary has its first piece stored as that run's tail partial as well)
        const f32x4* p = reinterpret_cast<const f32x4*>(part + ((size_t)r0 * 2 + 1) * D);
#pragma unroll
        for (int n = 0; n < NV; ++n) {
            const f32x4 t = (lane + 64 * n < D4) ? p[lane + 64 * n] : f32x4{0.f, 0.f, 0.f, 0.f};
            acc[n] = (((t + acc[n]) + sh[0][lane + 64 * n]) + sh[1][lane + 64 * n]) + sh[2][lane + 64 * n];
        }
    }
    float* const row = (enc ? a.dW_encT : a.dW_dec) + (size_t)i * D;
#pragma unroll
    for (int n = 0; n < NV; ++n)
        if (lane + 64 * n < D4) reinterpret_cast<f32x4*>(row)[lane + 64 * n] = acc[n];
    if (enc) {
        if (lane == 0) a.db_enc[i] = ((shdb[0] + shdb[1]) + shdb[2]) + shdb[3];
        if (a.enc_sq != nullptr) {
            const float sq = row_sumsq<NV>(acc);
            if (lane == 0) a.enc_sq[i] = sq;
        }
    } else if (a.row_proj != nullptr) {
        f32x4 wv[NV];
        const f32x4* wr = reinterpret_cast<const f32x4*>(a.W_dec + (size_t)i * D);
#pragma unroll
        for (int n = 0; n < NV; ++n) wv[n] = (lane + 64 * n < D4) ? wr[lane + 64 * n] : f32x4{0.f, 0.f, 0.f, 0.f};
        write_row_proj<NV>(a.row_proj, i, acc, wv, a.project, lane);
    }
}

// Everything else, one wave per latent and gradient row: a latent inside one run -- the row the pass stored, read back for its
// statistics; an unused latent -- zeros.
template <int NV>
__global__ __launch_bounds__(256) void dw_finalize_kernel(DwSlicesArgs a, int kind0) {
    constexpr int L = DWS_RUN;
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= a.S) return;
    const int enc = blockIdx.y + kind0;
    const int D = a.D, D4 = D >> 2;
    const int s = a.starts[i], e = a.starts[i + 1];
    if (!enc && a.lat_unused != nullptr && lane == 0) a.lat_unused[i] = e == s ? 1 : 0;
    if (e - s >= L && s / L != (e - 1) / L) return;  // (dw_finalize_cut_kernel's)
    float* const row = (enc ? a.dW_encT : a.dW_dec) + (size_t)i * D;
    f32x4 acc[NV], wv[NV];
    const bool proj = !enc && a.row_proj != nullptr;
    const f32x4* wr = reinterpret_cast<const f32x4*>(a.W_dec + (size_t)i * D);
#pragma unroll
    for (int n = 0; n < NV; ++n) {  // (the gradient row and the decoder row in flight together)
        acc[n] = f32x4{0.f, 0.f, 0.f, 0.f};
        wv[n] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (lane + 64 * n < D4) {
            if (e > s) acc[n] = reinterpret_cast<const f32x4*>(row)[lane + 64 * n];
            else if (!(enc && a.lat_unused != nullptr)) reinterpret_cast<f32x4*>(row)[lane + 64 * n] = acc[n];  // (unused latent)
            if (proj) wv[n] = wr[lane + 64 * n];
        }
    }
    if (enc) {
        float dbs = 0.f;
        for (int p = s + lane; p < e; p += 64) dbs += __int_as_float(a.pv2[p].y);
        dbs = wave_sum(dbs);
        if (lane == 0) a.db_enc[i] = dbs;
        if (a.enc_sq != nullptr) {
            const float sq = row_sumsq<NV>(acc);
            if (lane == 0) a.enc_sq[i] = sq;
        }
    } else if (proj) {
        write_row_proj<NV>(a.row_proj, i, acc, wv, a.project, lane);
    }
}

// The light finalize (DwSlicesArgs::wn2): ONE launch of 256-thread workgroups.
//   * Workgroups [0, n_cut_blocks) walk the list of latents cut by run boundaries (DwSlicesArgs::cut_list: L or more pairs, spanning
//     a boundary; most runs have none), one (latent, gradient row) item at a time: dw_finalize_cut_kernel's sums in the same order
//     -- bit-identical rows and db_enc -- with eight partial rows in flight per wave instead of four (a latent that fires on every
//     row has 64 partial rows per wave: the critical path of this launch).  256 threads per item: most cut latents have a handful
//     of partials, and four times as many items are resident as with the 1 024-thread form (45 us for ~6 000 items).
//   * The rest: four latents each, one wave per latent.  No decoder row is read: <dW_dec[i], w_i> = sum over the latent's pairs of
//     val * dval, ||w_i||^2 comes from normalize_rows (wn2).  The squares of the two gradient rows: kept per wave by the passes
//     (sq_wave_dec / _enc: only their total matters to the clip norm), or read here when the passes did not keep them.
// Both also zero the CSC bit map words of the pairs they walk (dw_clear_bitmap_kernel's job).
template <int NV>
__global__ __launch_bounds__(256) void dw_finalize_light_kernel(DwSlicesArgs a, int n_cut_blocks) {
    constexpr int L = DWS_RUN;
    __shared__ f32x4 sh[3][NV * 64];  // (waves 1-3; wave 0 keeps its sum in registers)
    __shared__ float shdb[4];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int D = a.D, D4 = D >> 2;
    if ((int)blockIdx.x < n_cut_blocks) {
      // (the list's length and this workgroup's first entry are requested together: the array is sized for every run)
      const i32x4 first = reinterpret_cast<const i32x4*>(a.cut_list)[1 + (blockIdx.x >> 1)];
      const int n_items = 2 * a.cut_list[0];
      for (int item = blockIdx.x; item < n_items; item += n_cut_blocks) {  // (block-uniform loop: the barriers inside are reached by all)
        const i32x4 ent = item == (int)blockIdx.x ? first : reinterpret_cast<const i32x4*>(a.cut_list)[1 + (item >> 1)];
        const int r0 = ent[0], i = ent[1], s = ent[2], e = ent[3], enc = item & 1;
        const int r1 = (e - 1) / L;
        const float* const part = enc ? a.part_enc : a.part_dec;
        // (what wave 0 needs at the end is requested now, next to the partial rows: the first run's tail partial, the decoder row)
        f32x4 tp[NV], wv[NV];
        if (w == 0) {
            const f32x4* p = reinterpret_cast<const f32x4*>(part + ((size_t)r0 * 2 + 1) * D);
            const f32x4* wr = reinterpret_cast<const f32x4*>(a.W_dec + (size_t)i * D);
            const bool need_w = !enc && a.row_proj != nullptr;
#pragma unroll
            for (int n = 0; n < NV; ++n) {
                const bool ok = lane + 64 * n < D4;
                tp[n] = ok ? p[lane + 64 * n] : f32x4{0.f, 0.f, 0.f, 0.f};
                wv[n] = (ok && need_w) ? wr[lane + 64 * n] : f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
        const int nh = r1 - r0, ch = (nh + 3) / 4;
        const int ra = r0 + 1 + w * ch, rb = min(r1 + 1, ra + ch);
        f32x4 acc[NV];
#pragma unroll
        for (int n = 0; n < NV; ++n) acc[n] = f32x4{0.f, 0.f, 0.f, 0.f};
        constexpr int TRIP = NV <= 4 ? 4 : 2;  // (the unit dw_finalize_cut_kernel adds pairwise: kept, two of them in flight)
        int r = ra;
        for (; r + 2 * TRIP <= rb; r += 2 * TRIP) {
            f32x4 t[2 * TRIP][NV];
#pragma unroll
            for (int u = 0; u < 2 * TRIP; ++u) {
                const f32x4* p = reinterpret_cast<const f32x4*>(part + (size_t)(r + u) * 2 * D);
#pragma unroll
                for (int n = 0; n < NV; ++n) t[u][n] = (lane + 64 * n < D4) ? p[lane + 64 * n] : f32x4{0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int n = 0; n < NV; ++n) {
                if constexpr (TRIP == 4) {
                    acc[n] += (t[0][n] + t[1][n]) + (t[2][n] + t[3][n]);
                    acc[n] += (t[4][n] + t[5][n]) + (t[6][n] + t[7][n]);
                } else {
                    acc[n] += t[0][n] + t[1][n];
                    acc[n] += t[2][n] + t[3][n];
                }
            }
        }
        for (; r + TRIP <= rb; r += TRIP) {
            f32x4 t[TRIP][NV];
#pragma unroll
            for (int u = 0; u < TRIP; ++u) {
                const f32x4* p = reinterpret_cast<const f32x4*>(part + (size_t)(r + u) * 2 * D);
#pragma unroll
                for (int n = 0; n < NV; ++n) t[u][n] = (lane + 64 * n < D4) ? p[lane + 64 * n] : f32x4{0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int n = 0; n < NV; ++n) {
                if constexpr (TRIP == 4) acc[n] += (t[0][n] + t[1][n]) + (t[2][n] + t[3][n]);
                else acc[n] += t[0][n] + t[1][n];
            }
        }
        for (; r < rb; ++r) {
            const f32x4* p = reinterpret_cast<const f32x4*>(part + (size_t)r * 2 * D);
#pragma unroll
            for (int n = 0; n < NV; ++n)
                if (lane + 64 * n < D4) acc[n] += p[lane + 64 * n];
        }
        float dbs = 0.f;
        if (enc) {  // db_enc: a quarter of the latent's pairs per wave (and their bit map words)
            // (eight pair words in flight per lane: a latent that fires on every row has 64 of them per lane, and the bit map stores
            // -- which the compiler must assume to alias the list -- kept the loop at one load per trip: 38 us for that one latent,
            // the critical path of the whole launch.  Same order of additions.)
            const int cnt = e - s, q4 = (cnt + 3) / 4;
            const int pend = min(e, s + (w + 1) * q4);
            for (int p0 = s + w * q4 + lane; p0 < pend; p0 += 64 * 8) {
                int2 pe[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) pe[u] = (p0 + 64 * u < pend) ? a.pv2[p0 + 64 * u] : int2{0, 0};
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    if (p0 + 64 * u < pend) {
                        dbs += __int_as_float(pe[u].y);
                        if (a.clear_bitmap != nullptr) a.clear_bitmap[(size_t)i * a.clear_words + (pe[u].x >> 12)] = 0u;
                    }
                }
            }
            dbs = wave_sum(dbs);
        }
        if (w != 0) {
#pragma unroll
            for (int n = 0; n < NV; ++n) sh[w - 1][lane + 64 * n] = acc[n];
        }
        if (lane == 0) shdb[w] = dbs;
        __syncthreads();
        if (w == 0) {
            // (a latent that begins exactly at a run boundary has its first piece stored as that run's tail partial as well)
#pragma unroll
            for (int n = 0; n < NV; ++n) acc[n] = (((tp[n] + acc[n]) + sh[0][lane + 64 * n]) + sh[1][lane + 64 * n]) + sh[2][lane + 64 * n];
            float* const row = (enc ? a.dW_encT : a.dW_dec) + (size_t)i * D;
#pragma unroll
            for (int n = 0; n < NV; ++n)
                if (lane + 64 * n < D4) reinterpret_cast<f32x4*>(row)[lane + 64 * n] = acc[n];
            if (enc) {
                if (lane == 0) a.db_enc[i] = ((shdb[0] + shdb[1]) + shdb[2]) + shdb[3];
                if (a.enc_sq != nullptr) {
                    const float sq = row_sumsq<NV>(acc);
                    if (lane == 0) a.enc_sq[i] = sq;
                }
            } else if (a.row_proj != nullptr) {
                write_row_proj<NV>(a.row_proj, i, acc, wv, a.project, lane);
            }
        }
        __syncthreads();  // (the LDS arrays are reused by the next item)
      }
      return;
    }
    // ---- every other latent: an eight-lane group (32 latents per workgroup; a wave per latent left this part latency-bound: 31 us
    // for a few loads and stores per latent).  A latent that is not cut has fewer than 2 L = 128 pairs: lane li of the group plays
    // the lanes li, li + 8, ..., li + 56 of the wave that dw_finalize_kernel gives a latent, two terms each at most, and the sums
    // go through that wave's tree (xor 32, 16, 8 in registers, 4, 2, 1 across the group): db_enc comes out bit-identical. ----
    const int li = lane & 7;
    const int i = ((int)blockIdx.x - n_cut_blocks) * 32 + (int)(threadIdx.x >> 3);
    if (i >= a.S) return;
    const int s = a.starts[i], e = a.starts[i + 1];
    if (a.lat_unused != nullptr && li == 0) a.lat_unused[i] = e == s ? 1 : 0;
    if (e - s >= L && s / L != (e - 1) / L) return;  // (a cut latent)
    if (e == s) {  // unused: statistics zero; rows zeroed only for callers that read the gradient buffers (no lat_unused flag)
        if (a.lat_unused == nullptr) {
            for (int q = li; q < D4; q += 8) {
                reinterpret_cast<f32x4*>(a.dW_dec + (size_t)i * D)[q] = f32x4{0.f, 0.f, 0.f, 0.f};
                reinterpret_cast<f32x4*>(a.dW_encT + (size_t)i * D)[q] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
        if (li == 0) {
            a.db_enc[i] = 0.f;
            if (a.enc_sq != nullptr) a.enc_sq[i] = 0.f;
            if (a.row_proj != nullptr) a.row_proj[i] = float2{0.f, 0.f};
        }
        return;
    }
    auto group_tree = [&](float (&v)[8]) -> float {  // = wave_sum over the 64 lanes this group stands for (lane l = li + 8 j)
        float r = ((v[0] + v[4]) + (v[2] + v[6])) + ((v[1] + v[5]) + (v[3] + v[7]));
        r += __shfl_xor(r, 4, 64);
        r += __shfl_xor(r, 2, 64);
        r += __shfl_xor(r, 1, 64);
        return r;
    };
    float gsq = 0.f, esq = 0.f;
    const bool waves_hold_squares = a.sq_wave_dec != nullptr;  // (the passes added up the squares of both rows: nothing to read here)
    if (!waves_hold_squares) {
        const f32x4* gd = reinterpret_cast<const f32x4*>(a.dW_dec + (size_t)i * D);
        const f32x4* ge = reinterpret_cast<const f32x4*>(a.dW_encT + (size_t)i * D);
#pragma unroll 4
        for (int q = li; q < D4; q += 8) {
            const f32x4 u = gd[q], v = ge[q];
            gsq = __builtin_fmaf(u[3], u[3], __builtin_fmaf(u[2], u[2], __builtin_fmaf(u[1], u[1], __builtin_fmaf(u[0], u[0], gsq))));
            esq = __builtin_fmaf(v[3], v[3], __builtin_fmaf(v[2], v[2], __builtin_fmaf(v[1], v[1], __builtin_fmaf(v[0], v[0], esq))));
        }
        gsq += __shfl_xor(gsq, 4, 64); gsq += __shfl_xor(gsq, 2, 64); gsq += __shfl_xor(gsq, 1, 64);
        esq += __shfl_xor(esq, 4, 64); esq += __shfl_xor(esq, 2, 64); esq += __shfl_xor(esq, 1, 64);
    }
    float dv8[8], dt8[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        float dbs = 0.f, dot = 0.f;
#pragma unroll
        for (int t = 0; t < 2; ++t) {  // (the wave lane's loop: p = s + l, s + l + 64)
            const int p = s + li + 8 * j + 64 * t;
            if (p < e) {
                const int2 pd = a.pv2[p];
                const float dv = __int_as_float(pd.y);
                dbs += dv;
                dot = __builtin_fmaf(__int_as_float(a.pv[p].y), dv, dot);
                if (a.clear_bitmap != nullptr) a.clear_bitmap[(size_t)i * a.clear_words + (pd.x >> 12)] = 0u;
            }
        }
        dv8[j] = dbs; dt8[j] = dot;
    }
    // (wave_sum's tree: xor 32 pairs j with j + 4, xor 16 j with j + 2, xor 8 j with j + 1)
    const float dbs = group_tree(dv8), dot = group_tree(dt8);
    if (li == 0) {
        a.db_enc[i] = dbs;
        if (a.enc_sq != nullptr) a.enc_sq[i] = esq;
        if (a.row_proj != nullptr) {
            // ||g - sc w||^2 = ||g||^2 - sc <g, w> with sc = <g, w> / ||w||^2 (modeling.py:419-445); never negative in exact arithmetic
            const float nsq = a.wn2[i];
            const float sc = (a.project && nsq > 0.f) ? dot / nsq : 0.f;
            // (waves_hold_squares: only the projection's correction is per latent; the total cannot go negative, a term may)
            a.row_proj[i] = float2{sc, waves_hold_squares ? -sc * dot : fmaxf(__builtin_fmaf(-sc, dot, gsq), 0.f)};
        }
    }
}

// out (D, S) = in (S, D)^T, 64 x 64 tiles through LDS (padded: conflict-free both ways)
// sq_part (optional): one double per block = the sum of squares of its tile, block (x, y) at [y * gridDim.x + x] -- the
// clip norm's share of this matrix without another pass over it
__global__ __launch_bounds__(256) void transpose_kernel(const float* __restrict__ in, float* __restrict__ out, int S,
                                                        int D, double* __restrict__ sq_part) {
    // 64 x 64 tile; both sides move 16 bytes per lane (S % 4 == 0 and D % 4 == 0)
    __shared__ float tile[64][65];
    __shared__ double sh[4];
    const int s0 = blockIdx.x * 64, d0 = blockIdx.y * 64;
    const int c4 = (threadIdx.x & 15) * 4, r0 = threadIdx.x >> 4;
    float q0 = 0.f, q1 = 0.f, q2 = 0.f, q3 = 0.f;
#pragma unroll
    for (int r = r0; r < 64; r += 16) {
        const int s = s0 + r, d = d0 + c4;
        f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
        if (s < S && d < D) v = *reinterpret_cast<const f32x4*>(in + (size_t)s * D + d);
        tile[r][c4] = v[0]; tile[r][c4 + 1] = v[1]; tile[r][c4 + 2] = v[2]; tile[r][c4 + 3] = v[3];
        q0 += v[0] * v[0]; q1 += v[1] * v[1]; q2 += v[2] * v[2]; q3 += v[3] * v[3];
    }
    if (sq_part != nullptr) {
        const double q = wave_sum_d((double)q0 + (double)q1 + (double)q2 + (double)q3);
        if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = q;
    }
    __syncthreads();
    if (sq_part != nullptr && threadIdx.x == 0) sq_part[(size_t)blockIdx.y * gridDim.x + blockIdx.x] = sh[0] + sh[1] + sh[2] + sh[3];
#pragma unroll
    for (int r = r0; r < 64; r += 16) {
        const int d = d0 + r, s = s0 + c4;
        if (d < D && s < S)
            *reinterpret_cast<f32x4*>(out + (size_t)d * S + s) = f32x4{tile[c4][r], tile[c4 + 1][r], tile[c4 + 2][r], tile[c4 + 3][r]};
    }
}

// ------------------------------- column sums -----------------------------------------------

template <bool ABSMAX>
__global__ __launch_bounds__(256) void colsum_partial_kernel(const float* m, int n_rows, int D, float* partials,
                                                             const int32_t* k_dev, long row_stride, float* wg_absmax,
                                                             int col_mult) {
    if (k_dev && *k_dev <= 0) return;
    const int Dfull = D;
    if (k_dev && col_mult > 0) D = min(D, (*k_dev * col_mult + 3) & ~3);  // only the first *k_dev * col_mult columns are live
    float amax = 0.f;
    const int r0 = blockIdx.x * 64;
    const int r1 = min(n_rows, r0 + 64);
    for (int q = threadIdx.x; q < (D >> 2); q += 256) {
        f32x4 s = {0.f, 0.f, 0.f, 0.f};
        int r = r0;
        for (; r + 8 <= r1; r += 8) {  // eight independent loads in flight; fixed summation tree
            f32x4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = reinterpret_cast<const f32x4*>(m + (size_t)(r + u) * row_stride)[q];
            s += ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
            if constexpr (ABSMAX) {
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    amax = fmaxf(fmaxf(fmaxf(amax, fabsf(v[u][0])), fabsf(v[u][1])), fmaxf(fabsf(v[u][2]), fabsf(v[u][3])));
            }
        }
        for (; r < r1; ++r) {
            const f32x4 v = reinterpret_cast<const f32x4*>(m + (size_t)r * row_stride)[q];
            s += v;
            if constexpr (ABSMAX) amax = fmaxf(fmaxf(fmaxf(amax, fabsf(v[0])), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3])));
        }
        reinterpret_cast<f32x4*>(partials + (size_t)blockIdx.x * Dfull)[q] = s;
    }
    if constexpr (ABSMAX) {  // the same pass also gives max |m| (one value per workgroup; max_reduce finishes)
        __shared__ float sh[4];
        for (int o = 32; o > 0; o >>= 1) amax = fmaxf(amax, __shfl_xor(amax, o, 64));
        if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = amax;
        __syncthreads();
        if (threadIdx.x == 0) wg_absmax[blockIdx.x] = fmaxf(fmaxf(sh[0], sh[1]), fmaxf(sh[2], sh[3]));
    }
}
// 64 columns per workgroup, 4 threads per column each summing every 4th partial row with independent loads in
// flight, then a fixed-order LDS combine (deterministic).
__global__ __launch_bounds__(256) void colsum_final_kernel(const float* partials, int n_blocks, int D, float* out,
                                                           int accumulate, const int32_t* k_dev, float out_scale, int col_mult,
                                                           const float* wg_absmax, float* absmax_out,
                                                           saev_step_stats* zero_stats, int32_t* zero_flag) {
    if (k_dev && *k_dev <= 0) return;
    if (blockIdx.x == 0 && absmax_out != nullptr) {  // the maximum behind colsum_partial_kernel<true>, and the step's zeroing
        __shared__ float shm[4];
        float m = 0.f;
        for (int i = threadIdx.x; i < n_blocks; i += 256) m = fmaxf(m, wg_absmax[i]);
        m = wave_max(m);
        if ((threadIdx.x & 63) == 0) shm[threadIdx.x >> 6] = m;
        __syncthreads();
        if (threadIdx.x == 0) {
            *absmax_out = fmaxf(fmaxf(shm[0], shm[1]), fmaxf(shm[2], shm[3]));
            if (zero_stats != nullptr) *zero_stats = saev_step_stats{};
            if (zero_flag != nullptr) *zero_flag = 0;
        }
    }
    if (k_dev && col_mult > 0 && (int)blockIdx.x * 64 >= *k_dev * col_mult) return;  // columns past the live ones
    __shared__ float part[4][64];
    const int col = threadIdx.x & 63, slice = threadIdx.x >> 6;
    const int d = blockIdx.x * 64 + col;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (d < D) {
        int b = slice;
        for (; b + 12 < n_blocks; b += 16) {
            s0 += partials[(size_t)b * D + d];
            s1 += partials[(size_t)(b + 4) * D + d];
            s2 += partials[(size_t)(b + 8) * D + d];
            s3 += partials[(size_t)(b + 12) * D + d];
        }
        for (; b < n_blocks; b += 4) s0 += partials[(size_t)b * D + d];
    }
    part[slice][col] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (slice == 0 && d < D) {
        const float s = (part[0][col] + part[1][col]) + (part[2][col] + part[3][col]);
        out[d] = accumulate ? out[d] + s * out_scale : s * out_scale;
    }
}

// ---- backward begin in two launches instead of four ---------------------------------------------------------------------
// The column sums of dL/dx_hat (db_dec) are independent of the CSC build that runs next to them: their two passes ride in
// the grids of csc_fill / csc_count (workgroups past the CSC part), so that the small kernels overlap instead of queueing.
struct ColsumPlain { const float* m; int n_rows, D; float* partials; long row_stride; float* out; int n_blocks; };
__device__ __forceinline__ void colsum_partial_plain(const ColsumPlain& c, int bid) {  // = colsum_partial_kernel<false>, k_dev = NULL
    const int r0 = bid * 64, r1 = min(c.n_rows, r0 + 64);
    for (int q = threadIdx.x; q < (c.D >> 2); q += 256) {
        f32x4 s = {0.f, 0.f, 0.f, 0.f};
        int r = r0;
        for (; r + 8 <= r1; r += 8) {
            f32x4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = reinterpret_cast<const f32x4*>(c.m + (size_t)(r + u) * c.row_stride)[q];
            s += ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
        }
        for (; r < r1; ++r) s += reinterpret_cast<const f32x4*>(c.m + (size_t)r * c.row_stride)[q];
        reinterpret_cast<f32x4*>(c.partials + (size_t)bid * c.D)[q] = s;
    }
}
__device__ __forceinline__ void colsum_final_plain(const ColsumPlain& c, int bid, float (&part)[4][64]) {  // = colsum_final_kernel, out = sums
    const int col = threadIdx.x & 63, slice = threadIdx.x >> 6;
    const int d = bid * 64 + col;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (d < c.D && slice < 4) {
        int b = slice;
        for (; b + 12 < c.n_blocks; b += 16) {
            s0 += c.partials[(size_t)b * c.D + d];
            s1 += c.partials[(size_t)(b + 4) * c.D + d];
            s2 += c.partials[(size_t)(b + 8) * c.D + d];
            s3 += c.partials[(size_t)(b + 12) * c.D + d];
        }
        for (; b < c.n_blocks; b += 4) s0 += c.partials[(size_t)b * c.D + d];
    }
    if (slice < 4) part[slice][col] = (s0 + s1) + (s2 + s3);
    __syncthreads();  // (every thread of the workgroup comes through here: in a wider workgroup, waves past the fourth idle along)
    if (slice == 0 && d < c.D) c.out[d] = ((part[0][col] + part[1][col]) + (part[2][col] + part[3][col])) * 1.0f;
}
// The two scan launches in one (decoupled look-back): a workgroup scans its 1 024 latents, publishes its three totals and then the
// build's epoch as the "ready" word, and adds up the totals of the workgroups before it as they appear.  The launch admits at most
// 128 scan workgroups of 1 024 threads (d_sae <= 131 072; configs[3] has 80) plus the column-sum workgroups behind them: one per
// CU fits 256 CUs, so all of them are resident at once -- and even where they were not, a workgroup only ever waits for workgroups
// with LOWER indices, which the dispatcher starts first (in-order dispatch), so the wait cannot deadlock.  Same sums in the same order as
// csc_scan_block_kernel + csc_scan_offset_kernel.  Workgroups past the scan finish the build's column sums (colsum_final_plain).
__global__ __launch_bounds__(1024) void csc_scan_fused_kernel(CscArgs a, int n_scan, ColsumPlain c, int have_colsum) {
    __shared__ int wave_tot[16][3];
    __shared__ int base[3];
    __shared__ float part[4][64];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    if ((int)blockIdx.x >= n_scan) {
        if (have_colsum) colsum_final_plain(c, (int)blockIdx.x - n_scan, part);  // (block-uniform: all 1 024 threads reach its barrier)
        return;
    }
    const int blk = blockIdx.x;
    const int i = blk * 1024 + tid;
    int v[3] = {0, 0, 0};
    if (i < a.S) {
        const int cn = a.counts[i];
        const int nch = (cn + DW_CHUNK - 1) / DW_CHUNK;
        v[0] = cn; v[1] = max(1, nch); v[2] = nch > 1 ? nch : 0;
    }
    int incl[3] = {v[0], v[1], v[2]};
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            const int n = __shfl_up(incl[q], o, 64);
            if (lane >= o) incl[q] += n;
        }
    }
    if (lane == 63)
        for (int q = 0; q < 3; ++q) wave_tot[w][q] = incl[q];
    __syncthreads();
    int off[3] = {0, 0, 0};
    for (int j = 0; j < w; ++j)
        for (int q = 0; q < 3; ++q) off[q] += wave_tot[j][q];
    int32_t* const agg = a.scan_totals;  // [workgroup][4]: three totals, then the epoch of the build that wrote them
    if (tid == 1023) {
        for (int q = 0; q < 3; ++q) __hip_atomic_store(&agg[blk * 4 + q], off[q] + incl[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&agg[blk * 4 + 3], a.epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (tid < 64) {  // first wave: the totals of the workgroups before this one, each as soon as it is there
        int s3[3] = {0, 0, 0};
        for (int b = lane; b < blk; b += 64) {
            while (__hip_atomic_load(&agg[b * 4 + 3], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != a.epoch) __builtin_amdgcn_s_sleep(1);
            for (int q = 0; q < 3; ++q) s3[q] += __hip_atomic_load(&agg[b * 4 + q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        for (int q = 0; q < 3; ++q) s3[q] = wave_sum_i(s3[q]);
        if (lane == 0)
            for (int q = 0; q < 3; ++q) base[q] = s3[q];
    }
    __syncthreads();
    if (i < a.S) {
        a.starts[i] = base[0] + off[0] + incl[0] - v[0];
        if (a.chunk_starts) {
            a.chunk_starts[i] = base[1] + off[1] + incl[1] - v[1];
            a.part_starts[i] = base[2] + off[2] + incl[2] - v[2];
        }
    }
    if (blk == n_scan - 1 && tid == 1023) {
        if (a.zero_word != nullptr) *a.zero_word = 0;
        a.starts[a.S] = base[0] + off[0] + incl[0];
        if (a.chunk_starts) a.chunk_starts[a.S] = base[1] + off[1] + incl[1];
    }
}
__global__ __launch_bounds__(256) void csc_fill_colsum_kernel(CscArgs a, int n_fill, ColsumPlain c) {
    if ((int)blockIdx.x < n_fill) csc_fill_body(a, blockIdx.x, n_fill);
    else colsum_partial_plain(c, blockIdx.x - n_fill);
}
// (prefilled builds: nothing to fill, so the column sums' FIRST stage rides here and their second in the scan launch)
__global__ __launch_bounds__(256) void csc_count_colpart_kernel(CscArgs a, int n_count, ColsumPlain c) {
    // (the few long-running column-sum workgroups first: behind the 8 192 short count workgroups they ran alone at the end, 50 us)
    if ((int)blockIdx.x < c.n_blocks) colsum_partial_plain(c, blockIdx.x);
    else csc_count_body(a, blockIdx.x - c.n_blocks);
}
__global__ __launch_bounds__(256) void csc_count_colsum_kernel(CscArgs a, int n_count, ColsumPlain c) {
    __shared__ float part[4][64];
    if ((int)blockIdx.x < n_count) csc_count_body(a, blockIdx.x);
    else colsum_final_plain(c, blockIdx.x - n_count, part);
}

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

// ---- the decode out of 32-column slices (DecodeSliceArgs) ----------------------------------------------------------------------
// decode_q_kernel gathers whole 4 KB rows of a 134 MB matrix: past the L2 the fabric delivers 7.5 TB/s of them however many are in
// flight (tools/ubench/row_gather.hip).  A 32-column slice of W_dec is S x 128 B = 4 MB -- an XCD's L2 -- and gathers out of it run
// at 20 TB/s.  So: XCD x walks slices x, x + 8, ...; an eight-lane group owns one (activation row, slice): the 32 codes' 128-byte
// pieces in flight together (32 float4 per lane), x_hat / g / the loss terms of its 32 columns, and the slice's share of the 32
// dval = <g, W_dec[idx_j]>; decode_s_finish_kernel adds the D / 32 shares and loss terms per row in slice order.  x_hat is summed in
// code order exactly as decode_q_kernel does, so the two give the same x_hat and g bit for bit.
// OPT-IN (saev_debug_cfg.dw_route = 4): measured at configs[1] 255-265 us + 25 (finish) + 20 (the slice-major copy normalize_rows
// leaves) against decode_q's 196 us while a young dictionary reuses few latents and 324 once usage has spread -- a wash in the
// steady state, 0.1 ms slower early.  Of the 255: the 32 dot products + reduce-scatters of the dval shares 55 (VALU), the row-major
// x_hat / g pieces 14; the same gathers alone run in 100 us (tools/ubench/row_gather.hip: slices1 ... full V=3 167 us).
__global__ __launch_bounds__(256) void decode_s_kernel(DecodeSliceArgs s, int wg_per_slice) {
    const DecodeArgs& a = s.d;
    const int lane = threadIdx.x & 63, li = lane & 7;
    const int xcd = blockIdx.x & 7, qq = blockIdx.x >> 3;
    const int slice = xcd + 8 * (qq / wg_per_slice);
    if (slice * 32 >= a.D) return;
    const int row_raw = (qq % wg_per_slice) * 32 + (int)(threadIdx.x >> 3);
    const bool live = row_raw < a.n_rows;
    const int row = live ? row_raw : a.n_rows - 1;
    typedef int i32x4_ __attribute__((ext_vector_type(4)));
    // the row's 32 codes (one 128-byte line each for idx and val), read by every lane of the group: 16 broadcast loads that hit
    // L1 / L2 (handing them round the group instead takes 64 ds_bpermute per wave)
    const i32x4_* ir = reinterpret_cast<const i32x4_*>(a.idx + (size_t)row * a.code_stride);
    const f32x4* vr = reinterpret_cast<const f32x4*>(a.val + (size_t)row * a.code_stride);
    const int col4 = slice * 8 + li;  // this lane's float4 of the row
    f32x4 acc = reinterpret_cast<const f32x4*>(a.b_dec)[col4];
    // (what streams through -- x in, x_hat / g / the shares out -- is marked non-temporal: the XCD's 4 MB L2 is for the slice of W_dec)
    const f32x4 xv = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(a.x + (size_t)row * a.D) + col4);
    const __amdgpu_buffer_rsrc_t wres = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(s.WdS) + (size_t)slice * a.S * 32, 0,
                                                                           (uint32_t)a.S * 128u, 0x00020000);
    const uint32_t li16 = (uint32_t)li * 16u;
    // the 32 pieces stay in registers until dL/dx_hat is known (128 VGPRs); the codes themselves do not: indices are read four at
    // a time in front of their gathers, values four at a time in front of their fmas (L1 hits) -- 64 VGPRs of codes held across the
    // gathers cost the kernel its third wave per SIMD
    f32x4 wv[32];
    uint32_t okmask = 0;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const i32x4_ ci = ir[u];
#pragma unroll
        for (int h = 0; h < 4; ++h) {
            const int j = 4 * u + h;
            const int i = ci[h];
            const bool ok = j < a.k && i >= 0 && i < a.idx_limit;
            okmask |= ok ? (1u << j) : 0u;
            // (an absent code: an offset past the buffer -- zeros, no memory access)
            const i32x4_ t = __builtin_amdgcn_raw_buffer_load_b128(wres, (ok ? (uint32_t)i * 128u : 0xFFFFFF00u) | li16, 0, 0);
            wv[j] = f32x4{__int_as_float(t[0]), __int_as_float(t[1]), __int_as_float(t[2]), __int_as_float(t[3])};
        }
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const f32x4 cv = vr[u];
#pragma unroll
        for (int h = 0; h < 4; ++h)
            if (okmask & (1u << (4 * u + h))) acc += cv[h] * wv[4 * u + h];
    }
    const float u_ = a.upper ? fmaxf(*a.upper, 1e-12f) : 1.0f;
    float sse_scaled = 0.f;
    double sse64 = 0.0, sumsq64 = 0.0;
    f32x4 g;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float t = acc[e] / u_ - xv[e] / u_;
        sse_scaled += t * t * u_ * u_;
        g[e] = a.gscale * t * u_;
        const float r = xv[e] - acc[e];
        sse64 += (double)r * (double)r;
        sumsq64 += (double)xv[e] * (double)xv[e];
    }
    if (live) {
        if (a.x_hat) __builtin_nontemporal_store(acc, reinterpret_cast<f32x4*>(a.x_hat + (size_t)row * a.D) + col4);
        if (a.g) __builtin_nontemporal_store(g, reinterpret_cast<f32x4*>(a.g + (size_t)row * a.D) + col4);
        const size_t o = ((size_t)slice * a.n_rows + row) * 8 + li;
        if (a.gS != nullptr) __builtin_nontemporal_store(g, reinterpret_cast<f32x4*>(a.gS) + o);
        if (a.xS != nullptr) __builtin_nontemporal_store(xv, reinterpret_cast<f32x4*>(a.xS) + o);
    }
    // the slice's shares of the 32 dval: batch h leaves code 4 li + h with lane li.  Reduce-scatter over the eight lanes of the
    // group on the VALU (DPP): lane l pairs with 7 - l (row_half_mirror), then with l ^ 2 and l ^ 1 (quad_perm); at every step a lane
    // keeps the half of its values its own index lies in and adds the partner's copy of that half.
    auto dpp_mirror8 = [](float v) { return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x141, 0xF, 0xF, true)); };
    auto dpp_xor2 = [](float v) { return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x4E, 0xF, 0xF, true)); };
    auto dpp_xor1 = [](float v) { return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0xB1, 0xF, 0xF, true)); };
    f32x4 sh;
#pragma unroll
    for (int h = 0; h < 4; ++h) {
        float p[8];
#pragma unroll
        for (int uu = 0; uu < 8; ++uu) {
            const f32x4& w = wv[4 * uu + h];
            p[uu] = (g[0] * w[0] + g[1] * w[1]) + (g[2] * w[2] + g[3] * w[3]);
        }
        {
            const bool up = (li & 4) != 0;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float keep = up ? p[i + 4] : p[i], send = up ? p[i] : p[i + 4];
                p[i] = keep + dpp_mirror8(send);
            }
        }
        {
            const bool up = (li & 2) != 0;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const float keep = up ? p[i + 2] : p[i], send = up ? p[i] : p[i + 2];
                p[i] = keep + dpp_xor2(send);
            }
        }
        const bool up = (li & 1) != 0;
        const float keep = up ? p[1] : p[0], send = up ? p[0] : p[1];
        sh[h] = keep + dpp_xor1(send);  // lane l ends up with index l of the batch: code 4 l + h
    }
    if (live) __builtin_nontemporal_store(sh, reinterpret_cast<f32x4*>(s.dvp + (size_t)slice * s.dvp_pitch + (size_t)row * 32) + li);
    // loss terms of the 32 columns
    sse_scaled = group8_sum(sse_scaled);
#pragma unroll
    for (int o = 4; o >= 1; o >>= 1) {
        sse64 += __shfl_xor(sse64, o, 64);
        sumsq64 += __shfl_xor(sumsq64, o, 64);
    }
    if (live && li == 0) {
        double* pp = s.part + ((size_t)slice * a.n_rows + row) * 3;
        pp[0] = (double)sse_scaled; pp[1] = sse64; pp[2] = sumsq64;
    }
    if (live && slice == 0 && a.training && a.fired) {
#pragma unroll
        for (int h = 0; h < 4; ++h) {  // (lane li marks the codes 4 li .. 4 li + 3)
            const int jj = 4 * li + h;
            if (jj < a.k) {
                const int ii = a.idx[(size_t)row * a.code_stride + jj];
                const float vv = a.val[(size_t)row * a.code_stride + jj];
                if (ii >= 0 && vv != 0.f) a.fired[ii] = 1;
            }
        }
    }
}

// one wave per row: dval[j] = the D / 32 shares, the row's loss terms, code statistics (fixed summation trees)
__global__ __launch_bounds__(256) void decode_s_finish_kernel(DecodeSliceArgs s) {
    const DecodeArgs& a = s.d;
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= a.n_rows) return;
    const int n_slices = a.D / 32;
    {
        // lanes 0-31: code j over the first half of the slices, lanes 32-63: over the second half; the halves are added lower first
        const int j = lane & 31, half = lane >> 5;
        const int c0 = half * ((n_slices + 1) / 2), c1 = half ? n_slices : (n_slices + 1) / 2;
        float sum = 0.f;
        const float* p = s.dvp + (size_t)row * 32 + j;
        int c = c0;
        for (; c + 8 <= c1; c += 8) {
            float v[8];
#pragma unroll
            for (int uu = 0; uu < 8; ++uu) v[uu] = p[(size_t)(c + uu) * s.dvp_pitch];
#pragma unroll
            for (int uu = 0; uu < 8; ++uu) sum += v[uu];
        }
        for (; c < c1; ++c) sum += p[(size_t)c * s.dvp_pitch];
        const float other = __shfl_xor(sum, 32, 64);
        if (lane < a.k) a.dval_out[(size_t)row * a.code_stride + lane] = sum + other;
    }
    float sse_scaled = 0.f;
    double sse64 = 0.0, sumsq64 = 0.0;
    for (int c = lane; c < n_slices; c += 64) {
        const double* pp = s.part + ((size_t)c * a.n_rows + row) * 3;
        sse_scaled += (float)pp[0]; sse64 += pp[1]; sumsq64 += pp[2];
    }
    sse_scaled = wave_sum(sse_scaled);
    sse64 = wave_sum_d(sse64);
    sumsq64 = wave_sum_d(sumsq64);
    float l0 = 0.f, l1 = 0.f;
    if (lane < a.k) {
        const int32_t i = a.idx[(size_t)row * a.code_stride + lane];
        const float v = a.val[(size_t)row * a.code_stride + lane];
        if (i >= 0 && v != 0.f) { l0 = 1.f; l1 = fabsf(v); }
    }
    l0 = wave_sum(l0);
    l1 = wave_sum(l1);
    if (a.rowstats && lane == 0) {
        RowStats rs;
        rs.sse_scaled = sse_scaled; rs.l0 = l0; rs.l1 = l1; rs.aux_sse = 0.f;
        rs.sse64 = sse64; rs.sumsq64 = sumsq64;
        a.rowstats[row] = rs;
    }
}

}  // namespace

bool decode_forms_dval(int D, int k) { return k <= 64 && D % 256 == 0 && D >= 256 && D <= 1280; }
bool decode_matry_forms_dval(int D, int k) { return k <= 32 && D % 256 == 0 && D >= 256 && D <= 1024; }
hipError_t launch_decode(const DecodeArgs& a, hipStream_t stream) {
    if (a.n_rows <= 0) return hipSuccess;
    if (a.dval_out != nullptr && a.x != nullptr && decode_forms_dval(a.D, a.k)) {
        // (32 < k <= 64 in two halves of 32 rows.  Round 6 built the one-pass form -- all 64 rows in registers, two columns per lane, 10
        // waves per row at d_model 1280, every row gathered once: 64 x 8-byte loads per lane instead of 96 x 16-byte ones -- and measured
        // it at configs[3]: 6.09-6.11 ms per step against 6.04-6.12, no gain (the gathers are bound by vector-memory INSTRUCTIONS, 640
        // per row against 480, not by bytes); tools/experiments/wip_decode_q2.patch)
#define DQ(NW)                                                                                                             \
    if (a.k <= 32) hipLaunchKernelGGL((decode_q_kernel<NW, 1>), dim3(a.n_rows), dim3(64 * NW), 0, stream, a);              \
    else hipLaunchKernelGGL((decode_q_kernel<NW, 2>), dim3(a.n_rows), dim3(64 * NW), 0, stream, a)
        switch (a.D / 256) {
            case 1: DQ(1); break;
            case 2: DQ(2); break;
            case 3: DQ(3); break;
            case 4: DQ(4); break;
            default: DQ(5); break;
        }
#undef DQ
        return hipGetLastError();
    }
    return dispatch_nv(a.D, [&](auto nv) {
        hipLaunchKernelGGL(decode_kernel<decltype(nv)::value>, dim3((a.n_rows + 3) / 4), dim3(256), 0, stream, a);
    });
}
bool decode_slices_supported(int D, int S, int k, int code_stride) {
    return k <= 32 && code_stride == 32 && D % 32 == 0 && (uint64_t)S * 128ull < (1ull << 32) - 256ull;
}
hipError_t launch_decode_slices(const DecodeSliceArgs& a, hipStream_t stream) {
    const DecodeArgs& d = a.d;
    if (d.n_rows <= 0) return hipSuccess;
    if (!decode_slices_supported(d.D, d.S, d.k, d.code_stride) || d.x == nullptr || d.dval_out == nullptr) return hipErrorInvalidValue;
    const int wg_per_slice = (d.n_rows + 31) / 32;
    const int slice_blocks = (d.D / 32 + 7) / 8;
    hipLaunchKernelGGL(decode_s_kernel, dim3(8 * slice_blocks * wg_per_slice), dim3(256), 0, stream, a, wg_per_slice);
    hipLaunchKernelGGL(decode_s_finish_kernel, dim3((d.n_rows + 3) / 4), dim3(256), 0, stream, a);
    return hipGetLastError();
}
hipError_t launch_decode_matry(const DecodeArgs& a, const MatryArgs& m, hipStream_t stream) {
    if (a.n_rows <= 0) return hipSuccess;
    if (a.dval_out != nullptr && a.x != nullptr && a.training && decode_matry_forms_dval(a.D, a.k)) {
        const size_t smem = (size_t)m.P * a.D * sizeof(float);  // <= 64 KB (P <= 16, D <= 1024), next to < 1 KB of static LDS
        static bool attr_set = false;
        if (!attr_set) {
            const void* fns[4] = {reinterpret_cast<const void*>(&decode_matry_q_kernel<1>), reinterpret_cast<const void*>(&decode_matry_q_kernel<2>),
                                  reinterpret_cast<const void*>(&decode_matry_q_kernel<3>), reinterpret_cast<const void*>(&decode_matry_q_kernel<4>)};
            for (const void* f : fns) {
                hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, MAX_PREFIXES * 1024 * (int)sizeof(float));
                if (e != hipSuccess) return e;
            }
            attr_set = true;
        }
        switch (a.D / 256) {
            case 1: hipLaunchKernelGGL(decode_matry_q_kernel<1>, dim3(a.n_rows), dim3(64), smem, stream, a, m); break;
            case 2: hipLaunchKernelGGL(decode_matry_q_kernel<2>, dim3(a.n_rows), dim3(128), smem, stream, a, m); break;
            case 3: hipLaunchKernelGGL(decode_matry_q_kernel<3>, dim3(a.n_rows), dim3(192), smem, stream, a, m); break;
            default: hipLaunchKernelGGL(decode_matry_q_kernel<4>, dim3(a.n_rows), dim3(256), smem, stream, a, m); break;
        }
        return hipGetLastError();
    }
    return dispatch_nv(a.D, [&](auto nv) {
        hipLaunchKernelGGL(decode_matry_kernel<decltype(nv)::value>, dim3((a.n_rows + 3) / 4), dim3(256), 0, stream, a, m);
    });
}
hipError_t launch_csc_build(const CscArgs& a, hipStream_t stream, bool bitmap_clean, const float* colsum_m, int colsum_D,
                            long colsum_row_stride, float* colsum_partials, float* colsum_out, bool prefilled) {
    if (a.n_rows <= 0) return hipSuccess;
    const long n = (long)a.n_rows * a.k;
    // (prefilled: the bits of exactly these codes are set already -- no clear, and the first launch is the column sums alone)
    const int blocks = prefilled ? 0 : (int)std::min<long>((n + 255) / 256, 4096);
    const int place_blocks = (int)std::min<long>((std::max<long>(n, a.S) + 255) / 256, 8192);
    if (!bitmap_clean && !prefilled) hipLaunchKernelGGL(csc_clear_kernel, dim3(2048), dim3(256), 0, stream, a);
    const int n_scan = (a.S + 1023) / 1024;
    ColsumPlain c{colsum_m, a.n_rows, colsum_D, colsum_partials, colsum_row_stride > 0 ? colsum_row_stride : (long)colsum_D,
                  colsum_out, (a.n_rows + 63) / 64};
    int colsum_in_scan = 0;
    if (colsum_m != nullptr && prefilled && a.epoch != 0) {
        // the bits are there already: count + first stage of the column sums, then scan + their second stage -- three launches in all
        hipLaunchKernelGGL(csc_count_colpart_kernel, dim3((a.S + 3) / 4 + c.n_blocks), dim3(256), 0, stream, a, (a.S + 3) / 4, c);
        colsum_in_scan = 1;
    } else if (colsum_m != nullptr) {  // out[d] = sum_b m[b][d] over the same n_rows rows, in the same two launches
        hipLaunchKernelGGL(csc_fill_colsum_kernel, dim3(blocks + c.n_blocks), dim3(256), 0, stream, a, blocks, c);
        hipLaunchKernelGGL(csc_count_colsum_kernel, dim3((a.S + 3) / 4 + (colsum_D + 63) / 64), dim3(256), 0, stream, a, (a.S + 3) / 4, c);
    } else {
        if (blocks > 0) hipLaunchKernelGGL(csc_fill_kernel, dim3(blocks), dim3(256), 0, stream, a);
        hipLaunchKernelGGL(csc_count_kernel, dim3((a.S + 3) / 4), dim3(256), 0, stream, a);
    }
    if (a.epoch != 0 && n_scan <= 128) {
        hipLaunchKernelGGL(csc_scan_fused_kernel, dim3(n_scan + (colsum_in_scan ? (colsum_D + 63) / 64 : 0)), dim3(1024), 0, stream, a, n_scan, c,
                           colsum_in_scan);
    } else {
        hipLaunchKernelGGL(csc_scan_block_kernel, dim3(n_scan), dim3(1024), 0, stream, a);
        hipLaunchKernelGGL(csc_scan_offset_kernel, dim3(n_scan), dim3(1024), 0, stream, a);
    }
    hipLaunchKernelGGL(csc_place_kernel, dim3(place_blocks), dim3(256), 0, stream, a);
    return hipGetLastError();
}
hipError_t launch_dw_rows(const DwRowsArgs& a, int max_work, hipStream_t stream) {
    return dispatch_nv(a.D, [&](auto nv) {
        hipLaunchKernelGGL(dw_rows_kernel<decltype(nv)::value>, dim3((max_work + 3) / 4), dim3(256), 0, stream, a);
        hipLaunchKernelGGL(dw_combine_kernel<decltype(nv)::value>, dim3((a.lat_hi - a.lat_lo + 3) / 4, 2), dim3(256), 0, stream, a);
    });
}
hipError_t launch_slice_major_copy(const float* g, const float* x, int n, int D, float* gS, float* xS, hipStream_t stream) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(slice_major_copy_kernel, dim3((n + 3) / 4), dim3(256), 0, stream, g, x, n, D, gS, xS);
    return hipGetLastError();
}
int dw_slices_waves(int D, int max_pairs) {  // waves of one pass of launch_dw_slices (the length of DwSlicesArgs::sq_wave_dec)
    const int n_runs = (max_pairs + DWS_RUN - 1) / DWS_RUN;
    return ((D / DWS_SLICE + 7) / 8) * 8 * ((n_runs + 31) / 32) * 4;
}
hipError_t launch_dw_slices(const DwSlicesArgs& a_in, int max_pairs, int part, hipStream_t stream) {
    DwSlicesArgs a = a_in;
    if (a.D % DWS_SLICE != 0 || max_pairs <= 0) return hipErrorInvalidValue;
    const int n_runs = (max_pairs + DWS_RUN - 1) / DWS_RUN;
    const int wg_per_slice = (n_runs + 31) / 32;
    const int grid = ((a.D / DWS_SLICE + 7) / 8) * 8 * wg_per_slice;
    // part 1: the decoder's half (pass A leaves the dval the encoder's half needs); part 2: the encoder's; 0: both
    // light finalize: both halves in one call, dval from the decode, and the three buffers it needs
    const bool light = part == 0 && a.have_dval && a.wn2 != nullptr;
    if (!light) { a.sq_wave_dec = nullptr; a.sq_wave_enc = nullptr; }
    if (part != 2 && a.have_dval) {
        hipLaunchKernelGGL((dw_slices_kernel<true, false>), dim3(grid), dim3(256), 0, stream, a, wg_per_slice);
        if (a.clear_bitmap != nullptr && !light) hipLaunchKernelGGL(dw_clear_bitmap_kernel, dim3((max_pairs + 255) / 256), dim3(256), 0, stream, a);
    } else if (part != 2) {
        hipLaunchKernelGGL(dw_slices_kernel<true>, dim3(grid), dim3(256), 0, stream, a, wg_per_slice);
        hipLaunchKernelGGL(dw_dval_sum_kernel, dim3((max_pairs + 255) / 256), dim3(256), 0, stream, a);
    }
    if (part != 1) hipLaunchKernelGGL(dw_slices_kernel<false>, dim3(grid), dim3(256), 0, stream, a, wg_per_slice);
    const int kinds = part == 0 ? 2 : 1, kind0 = part == 2 ? 1 : 0;
    if (light) {  // one launch: cut latents first, then a wave per latent that reads no row back (and clears the bit map words)
        const int n_cut = a.cut_list != nullptr ? std::min(2 * (n_runs - 1), 8192) : 0;  // workgroups that walk the list of cut latents
        if (a.cut_list == nullptr && n_runs > 1) return hipErrorInvalidValue;
        return dispatch_nv(a.D, [&](auto nv) {
            hipLaunchKernelGGL(dw_finalize_light_kernel<decltype(nv)::value>, dim3(n_cut + (a.S + 31) / 32), dim3(256), 0, stream, a, n_cut);
        });
    }
    return dispatch_nv(a.D, [&](auto nv) {
        if (n_runs > 1) hipLaunchKernelGGL(dw_finalize_cut_kernel<decltype(nv)::value>, dim3(n_runs - 1, kinds), dim3(256), 0, stream, a, kind0);
        hipLaunchKernelGGL(dw_finalize_kernel<decltype(nv)::value>, dim3((a.S + 3) / 4, kinds), dim3(256), 0, stream, a, kind0);
    });
}
hipError_t launch_transpose(const float* in, float* out, int S, int D, hipStream_t stream, double* sq_part) {
    hipLaunchKernelGGL(transpose_kernel, dim3((S + 63) / 64, (D + 63) / 64), dim3(256), 0, stream, in, out, S, D, sq_part);
    return hipGetLastError();
}
int transpose_blocks(int S, int D) { return ((S + 63) / 64) * ((D + 63) / 64); }
hipError_t launch_colsum(const float* m, int n_rows, int D, float* partials, float* out, int accumulate,
                         const int32_t* k_dev, hipStream_t stream, long row_stride, float out_scale, int col_mult) {
    const int nb = (n_rows + 63) / 64;
    if (nb <= 0) return hipSuccess;
    hipLaunchKernelGGL(colsum_partial_kernel<false>, dim3(nb), dim3(256), 0, stream, m, n_rows, D, partials, k_dev,
                       row_stride > 0 ? row_stride : (long)D, (float*)nullptr, col_mult);
    hipLaunchKernelGGL(colsum_final_kernel, dim3((D + 63) / 64), dim3(256), 0, stream, partials, nb, D, out,
                       accumulate, k_dev, out_scale, col_mult, (const float*)nullptr, (float*)nullptr, (saev_step_stats*)nullptr,
                       (int32_t*)nullptr);
    return hipGetLastError();
}
// column sums and max |m| from one pass over m; wg_scratch holds ceil(n_rows / 64) floats
hipError_t launch_colsum_absmax(const float* m, int n_rows, int D, float* partials, float* out, float* wg_scratch,
                                float* absmax_out, hipStream_t stream, float out_scale, saev_step_stats* zero_stats,
                                int32_t* zero_flag) {
    const int nb = (n_rows + 63) / 64;
    if (nb <= 0) return hipSuccess;
    hipLaunchKernelGGL(colsum_partial_kernel<true>, dim3(nb), dim3(256), 0, stream, m, n_rows, D, partials,
                       (const int32_t*)nullptr, (long)D, wg_scratch, 0);
    hipLaunchKernelGGL(colsum_final_kernel, dim3((D + 63) / 64), dim3(256), 0, stream, partials, nb, D, out, 0,
                       (const int32_t*)nullptr, out_scale, 0, (const float*)wg_scratch, absmax_out, zero_stats, zero_flag);
    return hipGetLastError();
}
