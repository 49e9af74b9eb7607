// HBM-bound streaming kernels of the step: decoder-row renormalisation (modeling.py:411-417),
// parallel-gradient removal (modeling.py:419-445), global gradient norm + clip (train.py:356-362),
// fused Adam (train.py:294,444-446), the dead-latent tracker (objectives.py:107-120), and the small
// reductions around them.  All are one pass over their operands with 16-byte accesses.
#include "common.h"
#include "kernels.h"

namespace {

template <int NV>
__global__ __launch_bounds__(256) void normalize_rows_kernel(float* W, int S, int D, float* WS, float* wn2) {
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= S) return;
    const int D4 = D >> 2;
    f32x4* r = reinterpret_cast<f32x4*>(W + (size_t)i * D);
    f32x4 v[NV];
    float ss = 0.f;
#pragma unroll
    for (int n = 0; n < NV; ++n) {
        const int q = lane + 64 * n;
        v[n] = (q < D4) ? r[q] : f32x4{0.f, 0.f, 0.f, 0.f};
        ss += v[n][0] * v[n][0] + v[n][1] * v[n][1] + v[n][2] * v[n][2] + v[n][3] * v[n][3];
    }
    ss = wave_sum(ss);
    const float nrm = sqrtf(ss);
    float s2 = 0.f;
#pragma unroll
    for (int n = 0; n < NV; ++n) {
        const int q = lane + 64 * n;
        if (q < D4) {
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) { o[e] = v[n][e] / nrm; s2 = __builtin_fmaf(o[e], o[e], s2); }
            r[q] = o;
            // slice-major copy [D / 32][S][32] for the slice decode (sparse.hip: decode_s_kernel), 128 bytes per (row, slice)
            if (WS != nullptr) reinterpret_cast<f32x4*>(WS)[((size_t)(q >> 3) * S + i) * 8 + (q & 7)] = o;
        }
    }
    if (wn2 != nullptr) {  // (same fma chain and lane order as rpg_row_stats forms ||w||^2 with)
        s2 = wave_sum(s2);
        if (lane == 0) wn2[i] = s2;
    }
}

// remove_parallel_grads (modeling.py:419-445) on rows [0, S) of gW; with `sq_partials` the same pass also leaves the sum
// of squares of the rows AS WRITTEN (one double per workgroup of four rows) for the clip norm, so the gradient of W_dec is
// not streamed a second time by the norm pass.  project == 0: squares only.
template <int NV>
__global__ __launch_bounds__(256) void rpg_kernel(float* gW, const float* W, int S, int D, double* sq_partials, int project) {
    __shared__ float sh[4];
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int D4 = D >> 2;
    float sq = 0.f;
    if (i < S) {
        f32x4* gr = reinterpret_cast<f32x4*>(gW + (size_t)i * D);
        const f32x4* wr = reinterpret_cast<const f32x4*>(W + (size_t)i * D);
        f32x4 g[NV], w[NV];
#pragma unroll
        for (int n = 0; n < NV; ++n) {
            const int q = lane + 64 * n;
            const bool ok = q < D4;
            g[n] = ok ? gr[q] : f32x4{0.f, 0.f, 0.f, 0.f};
            w[n] = (ok && project) ? wr[q] : f32x4{0.f, 0.f, 0.f, 0.f};
        }
        float sc;
        sq = rpg_row_stats<NV>(g, w, project, &sc);
        if (sc != 0.f) {
#pragma unroll
            for (int n = 0; n < NV; ++n) {
                const int q = lane + 64 * n;
                if (q < D4) gr[q] = f32x4{rpg_apply(g[n][0], sc, w[n][0]), rpg_apply(g[n][1], sc, w[n][1]),
                                          rpg_apply(g[n][2], sc, w[n][2]), rpg_apply(g[n][3], sc, w[n][3])};
            }
        }
    }
    if (sq_partials != nullptr) {
        if (lane == 0) sh[threadIdx.x >> 6] = sq;
        __syncthreads();
        if (threadIdx.x == 0) sq_partials[blockIdx.x] = ((double)sh[0] + (double)sh[1]) + ((double)sh[2] + (double)sh[3]);
    }
}

constexpr int SUMSQ_BLOCKS = 1024;

__global__ __launch_bounds__(256) void sumsq_partial_kernel(const float* g, long n, double* partials) {
    __shared__ double sh[4];
    const long n4 = n >> 2;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    for (long q = (long)blockIdx.x * 256 + threadIdx.x; q < n4; q += (long)gridDim.x * 256) {
        const f32x4 v = reinterpret_cast<const f32x4*>(g)[q];
        s0 += v[0] * v[0]; s1 += v[1] * v[1]; s2 += v[2] * v[2]; s3 += v[3] * v[3];
    }
    double s = (double)s0 + (double)s1 + (double)s2 + (double)s3;
    if (blockIdx.x == 0)
        for (long i = (n4 << 2) + threadIdx.x; i < n; i += 256) s += (double)g[i] * (double)g[i];
    s = wave_sum_d(s);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) partials[blockIdx.x] = sh[0] + sh[1] + sh[2] + sh[3];
}
// total = sum of nb partials: every thread adds its strided share in index order, then a fixed tree (deterministic)
__global__ __launch_bounds__(1024) void sumsq_final_kernel(const double* partials, int nb, double* total) {
    __shared__ double sh[16];
    double s = 0.0;
    for (int i = threadIdx.x; i < nb; i += 1024) s += partials[i];
    s = wave_sum_d(s);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int i = 0; i < 16; ++i) t += sh[i];
        *total = t;
    }
}

// The clip norm of a step whose decoder-gradient rows were never projected in memory (saev_train_step): total = sum of
//   * `partials` (nb doubles: per-tile squares of dW_enc left by the backward's transpose),
//   * row_proj[i].y for the S decoder rows (||g_i||^2 - <g_i, w_i>^2 / ||w_i||^2: what the projected row's squares sum to,
//     formed by the kernels that wrote the row -- dw_rows / dw_combine / scatter_add_dead),
//   * the squares of two short fp32 ranges (b_dec and b_enc, padding included),
// every thread adding its strided share in index order, then a fixed tree: deterministic.
__global__ __launch_bounds__(1024) void sumsq_final_ex_kernel(const double* partials, int nb, const float2* row_proj, int n_rows,
                                                              const float* e1, long n1, const float* e2, long n2, double* total,
                                                              double* blk_part, int* ticket, const float* enc_sq, const float* plain,
                                                              long n_plain) {
    // (one workgroup alone needed 32 us for the 0.4 MB: latency.  gridDim.x workgroups take contiguous slices of each
    // range, the last one to arrive -- a ticket -- adds the slices in index order)
    __shared__ double sh[16];
    __shared__ int last;
    const int nblk = gridDim.x, blk = blockIdx.x;
    auto slice = [&](long n, long& lo, long& hi) { const long per = (n + nblk - 1) / nblk; lo = min(n, blk * per); hi = min(n, lo + per); };
    double s = 0.0;
    long lo, hi;
    slice(nb, lo, hi);
    for (long i = lo + threadIdx.x; i < hi; i += 1024) s += partials[i];
    slice(n_rows, lo, hi);
    for (long i = lo + threadIdx.x; i < hi; i += 1024) s += (double)row_proj[i].y + (enc_sq != nullptr ? (double)enc_sq[i] : 0.0);
    slice(n1, lo, hi);
    for (long i = lo + threadIdx.x; i < hi; i += 1024) s += (double)e1[i] * (double)e1[i];
    slice(n2, lo, hi);
    for (long i = lo + threadIdx.x; i < hi; i += 1024) s += (double)e2[i] * (double)e2[i];
    slice(n_plain, lo, hi);
    for (long i = lo + threadIdx.x; i < hi; i += 1024) s += (double)plain[i];
    s = wave_sum_d(s);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int i = 0; i < 16; ++i) t += sh[i];
        __hip_atomic_store(&blk_part[blk], t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __threadfence();
        last = (atomicAdd(ticket, 1) == nblk - 1) ? 1 : 0;
        if (last) __threadfence();
    }
    __syncthreads();
    if (!last) return;
    __shared__ double part[64];
    if ((int)threadIdx.x < nblk && threadIdx.x < 64)  // all slices in flight together, then a fixed-order sum
        part[threadIdx.x] = __hip_atomic_load(&blk_part[threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (threadIdx.x == 0) {
        double tt = 0.0;
        for (int i = 0; i < nblk; ++i) tt += part[i];
        *total = tt;
        *ticket = 0;
    }
}

// One element of torch's fused Adam (train.py:294, 444-446) with every rounding spelled out -- no contraction left to the
// compiler -- so that the flat kernel and the row kernel (and any future variant) update bit-identically.
struct AdamElem { float p, m, v; };
__device__ __forceinline__ AdamElem adam_elem(float p, float ge, float m, float v, const AdamArgs& a, float step_size) {
#pragma clang fp contract(off)
    const float d = ge - m;
    m = __builtin_fmaf(d, a.omb1, m);
    const float g2 = a.omb2 * ge;
    const float g3 = g2 * ge;
    v = __builtin_fmaf(a.beta2, v, g3);
    const float denom = sqrtf(v) / a.bc2_sqrt + a.eps;
    const float upd = m / denom;
    p = __builtin_fmaf(-step_size, upd, p);
    return AdamElem{p, m, v};
}
__device__ __forceinline__ float scaled_grad(float g, float gs) {
#pragma clang fp contract(off)
    return g * gs;
}

__device__ __forceinline__ float clip_coef(const AdamArgs& a, float* norm_out) {
    const float norm = a.grad_scale * (float)sqrt(*a.sumsq);
    *norm_out = norm;
    return a.max_norm >= 0.f ? fminf(a.max_norm / (norm + 1e-6f), 1.f) : 1.f;
}

// Adam on one decoder row held across a wave (NV float4 per lane), the projection coefficient applied to the gradient as it is
// read.  The row is walked in chunks of four float4 per stream (16 loads in flight per lane) with the scheduler fenced
// between chunks: left alone the compiler interleaves all NV x 4 division / square-root sequences and takes 178 (NV = 4),
// 256 (NV = 5) or -- spilling thousands of dwords -- more than 256 (NV = 8) VGPRs for a kernel that lives on occupancy.
template <int NV>
__device__ __forceinline__ void adam_row(const AdamArgs& a, int i, int D, float sc, float gs, float step_size, int lane, bool g_zero = false) {
    const int D4 = D >> 2;
    const size_t base = (size_t)i * D4;
    constexpr int CH = 4;
#pragma unroll 1  // (a real loop: unrolled, the address arithmetic of all chunks is hoisted and spills)
    for (int n0 = 0; n0 < NV; n0 += CH) {
        f32x4 p[CH], g[CH], m[CH], v[CH];
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            const int q = lane + 64 * (n0 + c);
            if (n0 + c < NV && q < D4) {
                p[c] = __builtin_nontemporal_load(reinterpret_cast<f32x4*>(a.p) + base + q);
                g[c] = g_zero ? f32x4{0.f, 0.f, 0.f, 0.f} : __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(a.g) + base + q);
                m[c] = __builtin_nontemporal_load(reinterpret_cast<f32x4*>(a.m) + base + q);
                v[c] = __builtin_nontemporal_load(reinterpret_cast<f32x4*>(a.v) + base + q);
            }
        }
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            const int q = lane + 64 * (n0 + c);
            if (n0 + c >= NV || q >= D4) continue;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const AdamElem r = adam_elem(p[c][e], scaled_grad(rpg_apply(g[c][e], sc, p[c][e]), gs), m[c][e], v[c][e], a, step_size);
                p[c][e] = r.p; m[c][e] = r.m; v[c][e] = r.v;
            }
            __builtin_nontemporal_store(p[c], reinterpret_cast<f32x4*>(a.p) + base + q);
            __builtin_nontemporal_store(m[c], reinterpret_cast<f32x4*>(a.m) + base + q);
            __builtin_nontemporal_store(v[c], reinterpret_cast<f32x4*>(a.v) + base + q);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

// Adam over the S decoder rows with remove_parallel_grads applied on the way in: g' = g - sc_i * W_dec[i] with sc_i =
// row_proj[i].x (= <g_i, w_i> / ||w_i||^2 of the rows the backward wrote; 0 when the projection is off) -- the parameter
// row is being read anyway, so the projected gradient never has to be written (rpg_kernel: 0.4 GB of traffic per step).
// One wave per row, 16-byte non-temporal accesses, 4 * NV loads in flight per lane.
template <int NV>
__global__ __launch_bounds__(256, 4) void adam_rows_kernel(AdamArgs a, const float2* __restrict__ row_proj, int S, int D) {
    float norm;
    const float coef = clip_coef(a, &norm);
    if (blockIdx.x == 0 && threadIdx.x == 0 && a.stats) a.stats->grad_norm = norm;
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= S) return;
    adam_row<NV>(a, i, D, row_proj[i].x, a.grad_scale * coef, a.lr / a.bc1, lane);
}

// The whole Adam update of saev_train_step in ONE launch, reading every gradient where the backward left it:
//   blocks [0, nb_rows)            decoder rows, projection applied on the way in (adam_rows_kernel's body);
//   blocks [nb_rows, + nb_tiles)   W_enc in 32 x 256 tiles, the gradient taken from the TRANSPOSED (d_sae, d_model) scratch the
//                                  backward writes and turned through LDS -- the transpose pass that used to write the
//                                  gradient in W_enc's layout, and Adam's read of it, are gone (268 MB per step);
//   the rest                       b_dec and b_enc (and the padding of a sharded layout), element-wise.
// a.p / a.g / a.m / a.v are the flat buffers; off_* / n_* locate the segments.  Same adam_elem arithmetic as the other kernels.
struct AdamFusedArgs {
    AdamArgs a;
    const float2* row_proj;
    const float* gT;       // (S, D) transposed W_enc gradient
    const int32_t* lat_unused;  // optional: 1 = both gradient rows of the latent are zero (gT's is not even written)
    int S, D;
    long off_b_dec, n_b_dec, off_W_enc, off_b_enc, n_b_enc;
    int nb_rows, nb_tiles, tiles_s;
    AdamImageArgs img;     // img.ws != NULL: the W_enc tiles also leave the NEXT step's f16r operand images (kernels.h)
};
typedef _Float16 half8t __attribute__((ext_vector_type(8)));
__device__ __forceinline__ void adam_wenc_tile(const AdamFusedArgs& f, const AdamArgs& a, float gs, float step_size, int t) {
    const int S = f.S, D = f.D;
    {
        // tile = 32 (d) x 256 (s): the transposed gradient comes in as 128-byte row segments (whole lines), and p / m / v --
        // six of the seven streams -- move as 1 KB runs along s (a 64 x 64 tile moved them in 256-byte runs: 0.32 ms for
        // this launch instead of 0.27)
        constexpr int TD = 32, TS = 256, LDT = TS + 4;
        __shared__ float tile[TD][LDT];
        const int s0 = (t % f.tiles_s) * TS, d0 = (t / f.tiles_s) * TD;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int idx = threadIdx.x + 256 * i;
            const int r = idx >> 3, c4 = (idx & 7) * 4;
            const int sidx = s0 + r, d = d0 + c4;
            f32x4 g = f32x4{0.f, 0.f, 0.f, 0.f};
            if (sidx < S && d < D && !(f.lat_unused != nullptr && f.lat_unused[sidx] != 0))
                g = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(f.gT + (size_t)sidx * D + d));
            tile[c4][r] = g[0]; tile[c4 + 1][r] = g[1]; tile[c4 + 2][r] = g[2]; tile[c4 + 3][r] = g[3];
        }
        __syncthreads();
        float* const P = a.p + f.off_W_enc;
        float* const M = a.m + f.off_W_enc;
        float* const V = a.v + f.off_W_enc;
        const int sl = (threadIdx.x & 63) * 4, dr = threadIdx.x >> 6;
        const int sidx = s0 + sl;
        const bool emit = f.img.ws != nullptr;
        if (sidx >= S && !emit) return;
        f32x4 pn[8];
        // (AdamImageArgs::chk: the tile's checksums as read / as written -- a plain sum of the bit patterns, which no change of a
        // single element leaves alone, and a sum of words rotated by their position in the thread's share, against permutations)
        uint32_t ck_r0 = 0u, ck_r1 = 0u, ck_w0 = 0u, ck_w1 = 0u;
        const bool chk_on = emit && f.img.chk != nullptr;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int dl = dr + 4 * i, d = d0 + dl;
            pn[i] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (d >= D || sidx >= S) continue;
            const size_t o = (size_t)d * S + sidx;
            f32x4 p = __builtin_nontemporal_load(reinterpret_cast<f32x4*>(P + o));
            f32x4 m = __builtin_nontemporal_load(reinterpret_cast<f32x4*>(M + o));
            f32x4 v = __builtin_nontemporal_load(reinterpret_cast<f32x4*>(V + o));
            const f32x4 g = *reinterpret_cast<const f32x4*>(&tile[dl][sl]);
            if (chk_on) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const uint32_t u = __float_as_uint(p[e]);
                    ck_r0 += u; ck_r1 += __builtin_amdgcn_alignbit(u, u, (4 * i + e + 1) & 31);
                }
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const AdamElem q = adam_elem(p[e], scaled_grad(g[e], gs), m[e], v[e], a, step_size);
                p[e] = q.p; m[e] = q.m; v[e] = q.v;
            }
            if (chk_on) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const uint32_t u = __float_as_uint(p[e]);
                    ck_w0 += u; ck_w1 += __builtin_amdgcn_alignbit(u, u, (4 * i + e + 1) & 31);
                }
            }
            __builtin_nontemporal_store(p, reinterpret_cast<f32x4*>(P + o));
            __builtin_nontemporal_store(m, reinterpret_cast<f32x4*>(M + o));
            __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(V + o));
            pn[i] = p;
        }
        if (!emit) return;
        // ---- the next step's operand images of this tile (= image (s0 / 256, d0 / 32) of split_wT_body<2>, same arithmetic):
        // the updated values go back through the LDS tile, then thread rl owns latent s0 + rl: its 32 k as four fp16 chunks in the
        // encoder's image order, the slice-major fp32 row for the exact refinement, and the image's shares of <mu, w>, ||w||^2
        // and ||w - fp16(w)||^2 (bias_finish_kernel adds the shares of the D / 32 images)
        __shared__ uint32_t ck_sh[4][4];
        if (chk_on) {  // (integer sums: any order gives the same words)
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                ck_r0 += __shfl_xor(ck_r0, o, 64); ck_r1 += __shfl_xor(ck_r1, o, 64);
                ck_w0 += __shfl_xor(ck_w0, o, 64); ck_w1 += __shfl_xor(ck_w1, o, 64);
            }
            if ((threadIdx.x & 63) == 0) { uint32_t* q = ck_sh[threadIdx.x >> 6]; q[0] = ck_r0; q[1] = ck_r1; q[2] = ck_w0; q[3] = ck_w1; }
        }
        __syncthreads();  // (every thread has taken its gradient out of the tile)
        if (chk_on && threadIdx.x == 0) {
            uint32_t c[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) c[j] = (ck_sh[0][j] + ck_sh[1][j]) + (ck_sh[2][j] + ck_sh[3][j]);
            uint32_t* const slot = f.img.chk + 2 * (size_t)t;
            if (f.img.verify && (slot[0] != c[0] || slot[1] != c[1]) && (f.img.early == nullptr || *f.img.early == 0)) atomicAdd(f.img.late, 1);
            slot[0] = c[2]; slot[1] = c[3];
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) *reinterpret_cast<f32x4*>(&tile[dr + 4 * i][sl]) = pn[i];
        const int rl = threadIdx.x, ks = d0 / TD;
        const int swz = (4 - ((rl >> 2) & 3)) & 3;
        if (f.img.mode == 1) {  // the bf16 encoder: its images are W_enc^T rounded to bf16, nothing else (split_wT_body<1>)
            __syncthreads();
            typedef __bf16 bf16x8t __attribute__((ext_vector_type(8)));
            half8t* const imgb = reinterpret_cast<half8t*>(f.img.ws + ((size_t)(s0 / TS) * f.img.nks + ks) * 256 * 32);
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                bf16x8t h;
#pragma unroll
                for (int e = 0; e < 8; ++e) h[e] = (__bf16)tile[c * 8 + e][rl];
                imgb[rl * 4 + (c ^ swz)] = __builtin_bit_cast(half8t, h);
            }
            return;
        }
        __shared__ float mu_s[TD];
        if (threadIdx.x < TD) mu_s[threadIdx.x] = f.img.mu[d0 + threadIdx.x];
        __syncthreads();
        const float wm = *f.img.wmax_prev;
        const float scale = (wm > 0.f && wm < 3.0e38f) ? exp2f(13.0f - floorf(log2f(wm))) : 1.0f;
        if (t == 0 && threadIdx.x == 0) { f.img.scales_next[1] = scale; f.img.scales_next[3] = 1.0f; }
        double accp[4];
        float sqp[4], dsp[4];
        half8t* const img = reinterpret_cast<half8t*>(f.img.ws + ((size_t)(s0 / TS) * f.img.nks + ks) * 256 * 32);
        f32x4* const wes = reinterpret_cast<f32x4*>(f.img.WeS + ((size_t)ks * S + s0 + rl) * 32);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            float v[8];
            half8t h;
            double acc = 0.0;
            float sq = 0.f, dsq = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                v[e] = tile[c * 8 + e][rl] * scale;
                h[e] = (_Float16)v[e];
                acc += (double)mu_s[c * 8 + e] * (double)v[e];
                sq += v[e] * v[e];
                const float d = v[e] - (float)h[e];
                dsq += d * d;
            }
            img[rl * 4 + (c ^ swz)] = h;
            if (s0 + rl < S) {
                const float inv = 1.0f / scale;  // power of two
                wes[2 * c] = f32x4{v[0] * inv, v[1] * inv, v[2] * inv, v[3] * inv};
                wes[2 * c + 1] = f32x4{v[4] * inv, v[5] * inv, v[6] * inv, v[7] * inv};
            }
            accp[c] = acc; sqp[c] = sq; dsp[c] = dsq;
        }
        {
            // (the order in which split_wT_body's four lanes of a latent -- positions 0..3 holding chunks 0^swz .. 3^swz -- add up)
            const int c0 = swz, c1 = 1 ^ swz, c2 = 2 ^ swz, c3 = 3 ^ swz;
            const size_t o = (size_t)ks * f.img.S_pad + s0 + rl;
            f.img.dot_part[o] = (accp[c0] + accp[c1]) + (accp[c2] + accp[c3]);
            f.img.sq_part[o] = (sqp[c0] + sqp[c1]) + (sqp[c2] + sqp[c3]);
            f.img.sq_part[(size_t)f.img.nks * f.img.S_pad + o] = (dsp[c0] + dsp[c1]) + (dsp[c2] + dsp[c3]);
        }
        return;
    }
}

// PART 2 (shipped): rows, tiles and biases in one grid.  PART 0 / 1: rows + biases, and the W_enc tiles, as separate launches
// (SAEV_AMD_ADAM_SPLIT=1, for A/B runs).
template <int NV, int PART>
__global__ __launch_bounds__(256, (PART == 2 ? 3 : 4)) void adam_fused_kernel(AdamFusedArgs f) {
    const AdamArgs& a = f.a;
    float norm;
    const float coef = clip_coef(a, &norm);
    if (blockIdx.x == 0 && threadIdx.x == 0 && a.stats) a.stats->grad_norm = norm;
    const float gs = a.grad_scale * coef;
    const float step_size = a.lr / a.bc1;
    const int S = f.S, D = f.D;
    if constexpr (PART == 1) {
        adam_wenc_tile(f, a, gs, step_size, (int)blockIdx.x);
        return;
    }
    if constexpr (PART == 2) {  // everything in one grid: rows, then tiles, then biases
        if ((int)blockIdx.x >= f.nb_rows && (int)blockIdx.x < f.nb_rows + f.nb_tiles) {
            adam_wenc_tile(f, a, gs, step_size, (int)blockIdx.x - f.nb_rows);
            return;
        }
    }
    if ((int)blockIdx.x < f.nb_rows) {
        const int lane = threadIdx.x & 63;
        const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
        if (i >= S) return;
        adam_row<NV>(a, i, D, f.row_proj[i].x, gs, step_size, lane, f.lat_unused != nullptr && f.lat_unused[i] != 0);
        return;
    }
    // the two bias segments
    const long nb = gridDim.x - f.nb_rows - (PART == 2 ? f.nb_tiles : 0);
    const long bi = blockIdx.x - f.nb_rows - (PART == 2 ? f.nb_tiles : 0);
    for (int seg = 0; seg < 2; ++seg) {
        const long off = seg ? f.off_b_enc : f.off_b_dec, n = seg ? f.n_b_enc : f.n_b_dec;
        for (long i = bi * 256 + threadIdx.x; i < n; i += nb * 256) {
            const AdamElem r = adam_elem(a.p[off + i], scaled_grad(a.g[off + i], gs), a.m[off + i], a.v[off + i], a, step_size);
            a.p[off + i] = r.p; a.m[off + i] = r.m; a.v[off + i] = r.v;
        }
    }
}

__global__ __launch_bounds__(256) void adam_kernel(AdamArgs a) {
    // clip coefficient from the global norm of the (scaled) gradient; torch semantics
    const float norm = a.grad_scale * (float)sqrt(*a.sumsq);
    // torch.nn.utils.clip_grad_norm_: coef = min(max_norm / (norm + 1e-6), 1) -- max_norm = 0 therefore zeroes the
    // gradient, as it does in the reference; a NEGATIVE max_norm means "no clipping" here (torch has no such value)
    float coef = 1.f;
    if (a.max_norm >= 0.f) coef = fminf(a.max_norm / (norm + 1e-6f), 1.f);
    if (blockIdx.x == 0 && threadIdx.x == 0 && a.stats) a.stats->grad_norm = norm;
    const float gs = a.grad_scale * coef;
    const float step_size = a.lr / a.bc1;
    const long n4 = a.n >> 2;
    // Seven streams, each element touched once: non-temporal loads and stores (nothing here is worth a cache line to the
    // kernels that follow -- the buffers are larger than the Infinity Cache) and four float4 groups per thread and trip
    // (16 loads in flight per lane).  tools/ubench/adam_streams.hip: 0.378 -> 0.335 ms for the same bytes.
    constexpr int UNR = 4;
    for (long q0 = (long)blockIdx.x * 256 * UNR + threadIdx.x; q0 < n4; q0 += (long)gridDim.x * 256 * UNR) {
        f32x4 p[UNR], g[UNR], m[UNR], v[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const long q = q0 + u * 256;
            if (q < n4) {
                p[u] = __builtin_nontemporal_load(reinterpret_cast<f32x4*>(a.p) + q);
                g[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(a.g) + q);
                m[u] = __builtin_nontemporal_load(reinterpret_cast<f32x4*>(a.m) + q);
                v[u] = __builtin_nontemporal_load(reinterpret_cast<f32x4*>(a.v) + q);
            }
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const long q = q0 + u * 256;
            if (q >= n4) continue;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float ge = scaled_grad(g[u][e], gs);
                const AdamElem r = adam_elem(p[u][e], ge, m[u][e], v[u][e], a, step_size);
                p[u][e] = r.p; m[u][e] = r.m; v[u][e] = r.v;
            }
            __builtin_nontemporal_store(p[u], reinterpret_cast<f32x4*>(a.p) + q);
            __builtin_nontemporal_store(m[u], reinterpret_cast<f32x4*>(a.m) + q);
            __builtin_nontemporal_store(v[u], reinterpret_cast<f32x4*>(a.v) + q);
        }
    }
    if (blockIdx.x == 0) {
        for (long i = (n4 << 2) + threadIdx.x; i < a.n; i += 256) {
            const AdamElem r = adam_elem(a.p[i], scaled_grad(a.g[i], gs), a.m[i], a.v[i], a, step_size);
            a.p[i] = r.p; a.m[i] = r.m; a.v[i] = r.v;
        }
    }
}

// dead mask (S) -> ascending list of dead latents by ONE 1 024-thread workgroup, every thread a contiguous chunk (auxk.hip's
// dead_compact_kernel; here for the workgroup that finishes the tracker update)
__device__ __forceinline__ void dead_compact_block(const int32_t* dead, int S, int32_t* list, int (&wave_tot)[16]) {
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int per = (S + 1023) / 1024;
    const int i0 = tid * per, i1 = min(S, i0 + per);
    int cnt = 0;
    for (int i = i0; i < i1; ++i) cnt += dead[i] ? 1 : 0;
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
    if (cnt > 0)
        for (int i = i0; i < i1; ++i)
            if (dead[i]) list[pos++] = i;
}

// tracker update (objectives.py:107-120) on latents [bid * blockDim.x, ...); the last of `n_blocks` workgroups publishes the counts
// and the host-visible record; with a.dead_list (1 024-thread workgroups only) it also leaves the ascending list of dead latents
// whenever any are dead (dead_compact_kernel's launch)
__device__ __forceinline__ void dead_update_body(const DeadArgs& a, int bid, int n_blocks) {
    __shared__ int sh[2][16];
    __shared__ int wave_tot[16];
    __shared__ int last_n;
    const int i = bid * blockDim.x + threadIdx.x;
    const int nw = blockDim.x >> 6;
    int d = 0, near = 0;
    if (i < a.S) {
        int64_t t = a.toks[i] + a.add_tokens;
        if (a.fired[i]) t = 0;
        a.fired[i] = 0;
        a.toks[i] = t;
        d = (t >= a.threshold) ? 1 : 0;
        near = (t >= a.threshold - a.horizon_tokens) ? 1 : 0;  // dead now or within horizon_tokens more tokens of it
        a.dead[i] = d;
    }
    const int c = wave_sum_i(d), cn = wave_sum_i(near);
    if ((threadIdx.x & 63) == 0) { sh[0][threadIdx.x >> 6] = c; sh[1][threadIdx.x >> 6] = cn; }
    if (threadIdx.x == 0) last_n = -1;
    __syncthreads();
    if (threadIdx.x == 0) {
        int s0 = 0, s1 = 0;
        for (int w = 0; w < nw; ++w) { s0 += sh[0][w]; s1 += sh[1][w]; }
        atomicAdd(&a.scratch[0], s0);  // integer sums: order does not matter
        atomicAdd(&a.scratch[2], s1);
        __threadfence();
        if (atomicAdd(&a.scratch[1], 1) == n_blocks - 1) {  // last block: publish and reset
            __threadfence();
            const int t = atomicExch(&a.scratch[0], 0);
            const int tn = atomicExch(&a.scratch[2], 0);
            a.scratch[1] = 0;
            *a.n_dead = t;
            *a.k_use = min(a.k_aux, t);
            if (a.stats) a.stats->n_dead = t;
            if (a.rec) {  // host-visible record of this step (pinned memory; the host reads it a few steps later, after
                          // the event recorded behind this kernel has fired -- see saev_step_dead)
                a.rec->n_dead = t;
                a.rec->n_near = tn;
                a.rec->horizon_tokens = a.horizon_tokens;
                a.rec->cum_tokens = a.cum_tokens;
                a.rec->step = a.step;
            }
            last_n = t;
        }
    }
    if (a.dead_list == nullptr) return;
    __syncthreads();
    if (last_n > 0) dead_compact_block(a.dead, a.S, a.dead_list, wave_tot);  // (block-uniform: only the last workgroup sees last_n >= 0)
}
__global__ __launch_bounds__(256) void dead_update_kernel(DeadArgs a) {
    a.dead_list = nullptr;  // (the list needs 1 024-thread workgroups: stats_dead_kernel)
    dead_update_body(a, blockIdx.x, gridDim.x);
}

__global__ __launch_bounds__(256) void absmax_kernel(const float* x, long n, float* out) {
    const long n4 = n >> 2;
    float m = 0.f;
    for (long q = (long)blockIdx.x * 256 + threadIdx.x; q < n4; q += (long)gridDim.x * 256) {
        const f32x4 v = reinterpret_cast<const f32x4*>(x)[q];
        m = fmaxf(fmaxf(fmaxf(m, fabsf(v[0])), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3])));
    }
    if (blockIdx.x == 0)
        for (long i = (n4 << 2) + threadIdx.x; i < n; i += 256) m = fmaxf(m, fabsf(x[i]));
    m = wave_max(m);
    __shared__ float shm[4];
    if ((threadIdx.x & 63) == 0) shm[threadIdx.x >> 6] = m;
    __syncthreads();
    // non-negative floats order like their bit patterns; one atomic per workgroup
    if (threadIdx.x == 0)
        atomicMax(reinterpret_cast<unsigned int*>(out),
                  __float_as_uint(fmaxf(fmaxf(shm[0], shm[1]), fmaxf(shm[2], shm[3]))));
}

__global__ __launch_bounds__(256) void gather_rows_kernel(const float* pool, const int64_t* rows, int n_rows, int D,
                                                          float* out) {
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= n_rows) return;
    const f32x4* src = reinterpret_cast<const f32x4*>(pool + (size_t)rows[r] * D);
    f32x4* dst = reinterpret_cast<f32x4*>(out + (size_t)r * D);
    for (int q = lane; q < (D >> 2); q += 64) dst[q] = src[q];
}

__global__ __launch_bounds__(256) void scatter_dense_kernel(const int32_t* idx, const float* val, int n_rows, int k,
                                                            int stride, int S, float* f) {
    const long p = (long)blockIdx.x * 256 + threadIdx.x;
    if (p >= (long)n_rows * k) return;
    const int b = (int)(p / k), j = (int)(p % k);
    const int32_t i = idx[(size_t)b * stride + j];
    if (i >= 0 && i < S) f[(size_t)b * S + i] = val[(size_t)b * stride + j];
}

// Up to 16 workgroups each reduce a contiguous slice of the rows (one workgroup alone took 15 us for 16 384 rows: latency,
// not bytes); the last one to finish -- a ticket -- adds the slices in index order and writes the step's statistics.
// scratch: 16 x 8 doubles followed by the ticket counter (an int, zero between launches).  With cand_cnt the same pass
// also gives the list statistics of the fused encoder (rows whose list overflowed, longest list).
struct StatsArgs {
    const RowStats* rs; int n_rows, D, P; float alpha; int with_aux; const float* upper; const int32_t* n_overflow;
    saev_step_stats* stats; const int32_t* n_dead_dev; double* scratch; const int32_t* cand_cnt; int cand_cap;
};
__device__ __forceinline__ void stats_reduce_body(const StatsArgs& A, int bid, int n_blocks) {
    const RowStats* rs = A.rs; const int n_rows = A.n_rows, D = A.D, P = A.P; const float alpha = A.alpha; const int with_aux = A.with_aux;
    const float* upper = A.upper; const int32_t* n_overflow = A.n_overflow; saev_step_stats* stats = A.stats;
    const int32_t* n_dead_dev = A.n_dead_dev; double* scratch = A.scratch; const int32_t* cand_cnt = A.cand_cnt; const int cand_cap = A.cand_cap;
    // with_aux == 2: the AuxK pass of a step whose dead count only the device knows -- nothing to add when it is zero
    // (the forward's call has already written every other field)
    if (with_aux == 2 && *n_dead_dev <= 0) return;
    constexpr int NS = 8;
    __shared__ double sh[16][NS];
    __shared__ int last;
    double s[NS] = {0, 0, 0, 0, 0, 0, 0, 0};  // [6] rows with cand_cnt > cap, [7] max cand_cnt
    const int per = (n_rows + n_blocks - 1) / n_blocks;
    const int r0 = bid * per, r1 = min(n_rows, r0 + per);
    for (int r = r0 + threadIdx.x; r < r1; r += 1024) {
        const RowStats v = rs[r];
        s[0] += v.sse_scaled; s[1] += v.l0; s[2] += v.l1; s[3] += with_aux ? v.aux_sse : 0.f;
        s[4] += v.sse64; s[5] += v.sumsq64;
        if (cand_cnt != nullptr) {
            const int c = cand_cnt[r];
            s[6] += c > cand_cap ? 1.0 : 0.0;
            s[7] = fmax(s[7], (double)c);
        }
    }
#pragma unroll
    for (int i = 0; i < 7; ++i) s[i] = wave_sum_d(s[i]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s[7] = fmax(s[7], __shfl_xor(s[7], o, 64));
    if ((threadIdx.x & 63) == 0)
        for (int i = 0; i < NS; ++i) sh[threadIdx.x >> 6][i] = s[i];
    __syncthreads();
    int* ticket = reinterpret_cast<int*>(scratch + 16 * NS);
    if (threadIdx.x < NS) {  // one lane per statistic: the slice's value, written through to where the last workgroup reads it
        const int i = threadIdx.x;
        double t = 0.0;
        for (int w = 0; w < 16; ++w) t = i < 7 ? t + sh[w][i] : fmax(t, sh[w][i]);
        __hip_atomic_store(&scratch[bid * NS + i], t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();
        last = (atomicAdd(ticket, 1) == n_blocks - 1) ? 1 : 0;
        if (last) __threadfence();
    }
    __syncthreads();
    if (!last) return;
    // the last workgroup: all slices' values in flight together (one thread alone paid 128 L2 round trips here), then a
    // fixed-order sum per statistic
    __shared__ double part[16][NS];
    if (threadIdx.x < 16 * NS) {
        const int b = threadIdx.x / NS, i = threadIdx.x % NS;
        part[b][i] = b < n_blocks ? __hip_atomic_load(&scratch[b * NS + i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0;
    }
    __syncthreads();
    if (threadIdx.x != 0) return;
    {
        double t[NS] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int b = 0; b < n_blocks; ++b) {
            for (int i = 0; i < 7; ++i) t[i] += part[b][i];
            t[7] = fmax(t[7], part[b][7]);
        }
        *ticket = 0;
        const double nd = (double)n_rows * (double)D;
        stats->mse = (float)(t[0] / (nd * (double)P));  // mean over rows x prefixes x d_model
        stats->l0 = (float)(t[1] / n_rows);
        stats->l1 = (float)(t[2] / n_rows);
        if (with_aux) stats->aux = (float)((double)alpha * t[3] / nd);
        stats->sse = t[4];
        stats->sum_sq = t[5];
        if (upper) stats->upper = *upper;
        // ctx flags: [1] need_dense [2] n_overflow [3] cand_max (the last two from the lists themselves when given)
        if (n_overflow) { stats->n_overflow_rows = n_overflow[0]; stats->cand_max = n_overflow[1]; stats->dense_route = n_overflow[-1]; }
        if (cand_cnt != nullptr) { stats->n_overflow_rows = (int32_t)t[6]; stats->cand_max = (int32_t)t[7]; }
    }
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

__global__ __launch_bounds__(1024) void stats_reduce_kernel(StatsArgs A) { stats_reduce_body(A, blockIdx.x, gridDim.x); }
// the forward's statistics and the tracker update of a fused train step in one launch (they meet nowhere)
__global__ __launch_bounds__(1024) void stats_dead_kernel(StatsArgs A, int n_stat, DeadArgs d) {
    if ((int)blockIdx.x < n_stat) stats_reduce_body(A, blockIdx.x, n_stat);
    else dead_update_body(d, (int)blockIdx.x - n_stat, (int)gridDim.x - n_stat);
}

}  // namespace

hipError_t launch_normalize_rows(float* W, int S, int D, hipStream_t stream, float* WS, float* wn2) {
    return dispatch_nv(D, [&](auto nv) {
        hipLaunchKernelGGL(normalize_rows_kernel<decltype(nv)::value>, dim3((S + 3) / 4), dim3(256), 0, stream, W, S, D, WS, wn2);
    });
}
hipError_t launch_rpg(float* gW, const float* W, int S, int D, hipStream_t stream, double* sq_partials, int project) {
    if (S <= 0) return hipSuccess;
    return dispatch_nv(D, [&](auto nv) {
        hipLaunchKernelGGL(rpg_kernel<decltype(nv)::value>, dim3((S + 3) / 4), dim3(256), 0, stream, gW, W, S, D, sq_partials,
                           project);
    });
}
int sumsq_blocks() { return SUMSQ_BLOCKS; }
hipError_t launch_sumsq_partials(const float* g, long n, double* partials, hipStream_t stream) {
    hipLaunchKernelGGL(sumsq_partial_kernel, dim3(SUMSQ_BLOCKS), dim3(256), 0, stream, g, n, partials);  // n == 0: zeros
    return hipGetLastError();
}
hipError_t launch_sumsq_final(const double* partials, int nb, double* total, hipStream_t stream) {
    hipLaunchKernelGGL(sumsq_final_kernel, dim3(1), dim3(1024), 0, stream, partials, nb, total);
    return hipGetLastError();
}
hipError_t launch_sumsq(const float* g, long n, double* partials, double* total, hipStream_t stream) {
    hipLaunchKernelGGL(sumsq_partial_kernel, dim3(SUMSQ_BLOCKS), dim3(256), 0, stream, g, n, partials);
    hipLaunchKernelGGL(sumsq_final_kernel, dim3(1), dim3(1024), 0, stream, partials, SUMSQ_BLOCKS, total);
    return hipGetLastError();
}
hipError_t launch_sumsq_final_ex(const double* partials, int nb, const float2* row_proj, int n_rows, const float* e1, long n1,
                                 const float* e2, long n2, double* total, double* blk_part, int* ticket, hipStream_t stream,
                                 const float* enc_sq, const float* plain, long n_plain) {
    hipLaunchKernelGGL(sumsq_final_ex_kernel, dim3(SUMSQ_EX_BLOCKS), dim3(1024), 0, stream, partials, nb, row_proj, n_rows, e1, n1, e2,
                       n2, total, blk_part, ticket, enc_sq, plain, n_plain);
    return hipGetLastError();
}
hipError_t launch_adam_rows(const AdamArgs& a, const float2* row_proj, int S, int D, hipStream_t stream) {
    if (S <= 0) return hipSuccess;
    return dispatch_nv(D, [&](auto nv) {
        hipLaunchKernelGGL(adam_rows_kernel<decltype(nv)::value>, dim3((S + 3) / 4), dim3(256), 0, stream, a, row_proj, S, D);
    });
}
hipError_t launch_adam_fused(const AdamArgs& a, const float2* row_proj, const float* gT, int S, int D, long off_b_dec, long n_b_dec,
                             long off_W_enc, long off_b_enc, long n_b_enc, hipStream_t stream, const int32_t* lat_unused,
                             const AdamImageArgs* img) {
    AdamFusedArgs f{};
    if (img != nullptr && D % 32 == 0) f.img = *img;
    f.a = a; f.row_proj = row_proj; f.gT = gT; f.S = S; f.D = D; f.lat_unused = lat_unused;
    f.off_b_dec = off_b_dec; f.n_b_dec = n_b_dec; f.off_W_enc = off_W_enc; f.off_b_enc = off_b_enc; f.n_b_enc = n_b_enc;
    f.nb_rows = (S + 3) / 4;
    f.tiles_s = (S + 255) / 256;
    f.nb_tiles = f.tiles_s * ((D + 31) / 32);
    const int nb_bias = 32;
    return dispatch_nv(D, [&](auto nv) {
        // (rows, tiles and biases share ONE grid: it overlaps the LDS-bound tile workgroups with the streaming row
        // workgroups -- 302 us against 327 as two launches at configs[1])
        hipLaunchKernelGGL((adam_fused_kernel<decltype(nv)::value, 2>), dim3(f.nb_rows + f.nb_tiles + nb_bias), dim3(256), 0, stream, f);
    });
}
hipError_t launch_adam(const AdamArgs& a, hipStream_t stream) {
    const long n4 = a.n >> 2;
    const int blocks = (int)std::max<long>(1, std::min<long>((n4 + 1023) / 1024, 256 * 8));
    hipLaunchKernelGGL(adam_kernel, dim3(blocks), dim3(256), 0, stream, a);
    return hipGetLastError();
}
hipError_t launch_dead_update(const DeadArgs& a, hipStream_t stream) {
    hipLaunchKernelGGL(dead_update_kernel, dim3((a.S + 255) / 256), dim3(256), 0, stream, a);
    return hipGetLastError();
}
hipError_t launch_absmax(const float* x, long n, float* out_zeroed, hipStream_t stream) {
    const int blocks = (int)std::max<long>(1, std::min<long>(((n >> 2) + 255) / 256, 1024));
    hipLaunchKernelGGL(absmax_kernel, dim3(blocks), dim3(256), 0, stream, x, n, out_zeroed);
    return hipGetLastError();
}
hipError_t launch_gather_rows(const float* pool, const int64_t* rows, int n_rows, int D, float* out, hipStream_t stream) {
    if (n_rows <= 0) return hipSuccess;
    hipLaunchKernelGGL(gather_rows_kernel, dim3((n_rows + 3) / 4), dim3(256), 0, stream, pool, rows, n_rows, D, out);
    return hipGetLastError();
}
hipError_t launch_scatter_dense(const int32_t* idx, const float* val, int n_rows, int k, int stride, int S, float* f,
                                hipStream_t stream) {
    const long n = (long)n_rows * k;
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(scatter_dense_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, idx, val, n_rows, k,
                       stride, S, f);
    return hipGetLastError();
}
hipError_t launch_stats_reduce(const RowStats* rs, int n_rows, int D, int P, float alpha, int with_aux, const float* upper,
                               const int32_t* n_overflow, saev_step_stats* stats, hipStream_t stream,
                               const int32_t* n_dead_dev, double* scratch, const int32_t* cand_cnt, int cand_cap) {
    const int nb = std::max(1, std::min(16, (n_rows + 1023) / 1024));
    const StatsArgs A{rs, n_rows, D, P, alpha, with_aux, upper, n_overflow, stats, n_dead_dev, scratch, cand_cnt, cand_cap};
    hipLaunchKernelGGL(stats_reduce_kernel, dim3(nb), dim3(1024), 0, stream, A);
    return hipGetLastError();
}
hipError_t launch_stats_dead(const RowStats* rs, int n_rows, int D, int P, float alpha, const float* upper, const int32_t* n_overflow,
                             saev_step_stats* stats, double* scratch, const int32_t* cand_cnt, int cand_cap, const DeadArgs& d,
                             hipStream_t stream) {
    const int nb = std::max(1, std::min(16, (n_rows + 1023) / 1024));
    const StatsArgs A{rs, n_rows, D, P, alpha, 0, upper, n_overflow, stats, nullptr, scratch, cand_cnt, cand_cap};
    hipLaunchKernelGGL(stats_dead_kernel, dim3(nb + (d.S + 1023) / 1024), dim3(1024), 0, stream, A, nb, d);
    return hipGetLastError();
}
