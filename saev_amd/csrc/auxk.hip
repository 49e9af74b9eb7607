// AuxK dead-latent loss (reference nn/modeling.py:75-103) as dense algebra over the *dead set*.
//
// With k_aux = 512 codes per row drawn from the few hundred..thousand dead latents, the auxiliary codes are
// ~25 % dense over the dead columns: index-gathering 4 KB rows per code (what the main k = 32 path does) would
// move 2 * B * k_aux * 4D bytes per pass (68 GB at config 2), while the same contractions as GEMMs over the
// compacted dead set are a few hundred GFLOP.  So, with dl = ascending list of dead latents (nd of them):
//
//   H  = x @ W_enc[:, dl] + b_enc[dl]                 (B x nd)   plain GEMM  (rocBLAS sgemm)
//   A  = H masked to the k_use = min(k_aux, nd) largest entries per row   (select.hip radix select)
//   E  = A @ W_dec[dl] + b_dec                        (B x D)    plain GEMM
//   aux = alpha * mean((E - (x - x_hat))^2)            g_aux = d aux / dE
//   dA = (g_aux @ W_dec[dl]^T) * mask                 (B x nd)   plain GEMM
//   dW_dec[dl] += A^T @ g_aux,  dW_enc^T[dl] += dA^T @ x         plain GEMMs (K = B)
//   db_enc[dl] += colsum(dA),   db_dec += colsum(g_aux)
//
// These five products are unfused library GEMMs (the MFMA work that matters, the encoder, is hand-written in
// gemm_encode*.hip); everything around them (compaction, gathers, bias, masking, residual, scatter-add) is
// here.  n_dead is read back to the host once per step when dead latents are possible -- the reference does the
// same (`int(dead_mask.sum().item())`, modeling.py:92).
#include "common.h"
#include "kernels.h"

namespace {

// dead mask (S) -> ascending list of dead latents; single workgroup
__global__ __launch_bounds__(1024) void dead_compact_kernel(const int32_t* dead, int S, int32_t* list) {
    __shared__ int wave_tot[16];
    __shared__ int carry;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    if (tid == 0) carry = 0;
    __syncthreads();
    for (int base = 0; base < S; base += 1024) {
        const int i = base + tid;
        const int v = (i < S && dead[i]) ? 1 : 0;
        int incl = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int n = __shfl_up(incl, o, 64);
            if (lane >= o) incl += n;
        }
        if (lane == 63) wave_tot[w] = incl;
        __syncthreads();
        int off = carry;
        for (int j = 0; j < w; ++j) off += wave_tot[j];
        if (v) list[off + incl - 1] = i;
        __syncthreads();
        if (tid == 1023) carry = off + incl;
        __syncthreads();
    }
}

// Wenc_dead (D, ndp) = W_enc[:, dl] (zero columns beyond nd);  Wdec_dead (ndp, D) = W_dec[dl, :] (zero rows beyond nd)
__global__ __launch_bounds__(256) void gather_dead_kernel(const float* W_enc, const float* W_dec, const int32_t* dl,
                                                          int nd, int ndp, int D, int S, float* Wenc_dead,
                                                          float* Wdec_dead) {
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

// H[b][j] += b_enc[dl[j]]; padding columns become -inf so the select never takes them
__global__ __launch_bounds__(256) void dead_bias_kernel(float* H, long n_rows, int nd, int ndp, const float* b_enc,
                                                        const int32_t* dl) {
    const long total = n_rows * ndp;
    for (long q = (long)blockIdx.x * 256 + threadIdx.x; q < total; q += (long)gridDim.x * 256) {
        const int j = (int)(q % ndp);
        H[q] = (j < nd) ? H[q] + b_enc[dl[j]] : NEG_INF;
    }
}

// compact bias of the dead set for the fused dense contraction: out[j] = b_enc[dl[j]], -inf on the padding columns
__global__ void dead_bias_vec_kernel(const float* b_enc, const int32_t* dl, int nd, int ndp, float* out) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j < ndp) out[j] = (j < nd) ? b_enc[dl[j]] : NEG_INF;
}

// A[b][idx] = val, mask[b][idx] = 1 for the selected codes (A and mask are zeroed by the caller)
__global__ __launch_bounds__(256) void aux_scatter_kernel(const int32_t* idx, const float* val, long n_rows, int k,
                                                          int stride, int ndp, float* A, uint8_t* mask) {
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
                                                        int n_rows, int D, float gscale, RowStats* rowstats) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= n_rows) return;
    const int D4 = D >> 2;
    f32x4* er = reinterpret_cast<f32x4*>(E + (size_t)row * D);
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
            }
            er[q] = g;
        }
    }
    sse = wave_sum(sse);
    if (lane == 0) rowstats[row].aux_sse = sse;
}

__global__ __launch_bounds__(256) void mask_apply_kernel(float* dA, const uint8_t* mask, long n) {
    for (long q = (long)blockIdx.x * 256 + threadIdx.x; q < n; q += (long)gridDim.x * 256)
        if (!mask[q]) dA[q] = 0.f;
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
__global__ void scale_pair_kernel(const float* a, const float* b, float* out) {
    if (threadIdx.x == 0) { out[0] = *a; out[1] = *b; }
}

// gW_dec[dl[j], :] += dWd[j, :]; gW_encT[dl[j], :] += dWe[j, :]; gb_enc[dl[j]] += dbe[j]   (one wave per dead latent)
__global__ __launch_bounds__(256) void scatter_add_dead_kernel(const int32_t* dl, int nd, int D, const float* dWd,
                                                               const float* dWe, const float* dbe, float* gW_dec,
                                                               float* gW_encT, float* gb_enc, int lat_lo, int lat_hi) {
    const int lane = threadIdx.x & 63;
    const int j = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (j >= nd) return;
    const int i = dl[j];
    if (i < lat_lo || i >= lat_hi) return;
    const f32x4* a = reinterpret_cast<const f32x4*>(dWd + (size_t)j * D);
    const f32x4* e = reinterpret_cast<const f32x4*>(dWe + (size_t)j * D);
    f32x4* oa = reinterpret_cast<f32x4*>(gW_dec + (size_t)i * D);
    f32x4* oe = reinterpret_cast<f32x4*>(gW_encT + (size_t)i * D);
    for (int q = lane; q < (D >> 2); q += 64) {
        oa[q] = oa[q] + a[q];
        oe[q] = oe[q] + e[q];
    }
    if (lane == 0) gb_enc[i] += dbe[j];
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

hipError_t launch_dead_compact(const int32_t* dead, int S, int32_t* list, hipStream_t s) {
    hipLaunchKernelGGL(dead_compact_kernel, dim3(1), dim3(1024), 0, s, dead, S, list);
    return hipGetLastError();
}
hipError_t launch_gather_dead(const float* W_enc, const float* W_dec, const int32_t* dl, int nd, int ndp, int D, int S,
                              float* Wenc_dead, float* Wdec_dead, hipStream_t s) {
    hipLaunchKernelGGL(gather_dead_kernel, dim3(grid_for((long)D * ndp + (long)ndp * (D >> 2))), dim3(256), 0, s, W_enc,
                       W_dec, dl, nd, ndp, D, S, Wenc_dead, Wdec_dead);
    return hipGetLastError();
}
hipError_t launch_dead_bias(float* H, int n_rows, int nd, int ndp, const float* b_enc, const int32_t* dl, hipStream_t s) {
    hipLaunchKernelGGL(dead_bias_kernel, dim3(grid_for((long)n_rows * ndp)), dim3(256), 0, s, H, (long)n_rows, nd, ndp,
                       b_enc, dl);
    return hipGetLastError();
}
hipError_t launch_dead_bias_vec(const float* b_enc, const int32_t* dl, int nd, int ndp, float* out, hipStream_t s) {
    hipLaunchKernelGGL(dead_bias_vec_kernel, dim3((ndp + 255) / 256), dim3(256), 0, s, b_enc, dl, nd, ndp, out);
    return hipGetLastError();
}
hipError_t launch_aux_scatter(const int32_t* idx, const float* val, int n_rows, int k, int stride, int ndp, float* A,
                              uint8_t* mask, hipStream_t s) {
    hipLaunchKernelGGL(aux_scatter_kernel, dim3(grid_for((long)n_rows * k)), dim3(256), 0, s, idx, val, (long)n_rows, k,
                       stride, ndp, A, mask);
    return hipGetLastError();
}
hipError_t launch_aux_resid(float* E, const float* x, const float* x_hat, const float* b_dec, int n_rows, int D,
                            float gscale, RowStats* rowstats, hipStream_t s) {
    return dispatch_nv(D, [&](auto nv) {
        hipLaunchKernelGGL(aux_resid_kernel<decltype(nv)::value>, dim3((n_rows + 3) / 4), dim3(256), 0, s, E, x, x_hat,
                           b_dec, n_rows, D, gscale, rowstats);
    });
}
hipError_t launch_mask_apply(float* dA, const uint8_t* mask, long n, hipStream_t s) {
    hipLaunchKernelGGL(mask_apply_kernel, dim3(grid_for(n)), dim3(256), 0, s, dA, mask, n);
    return hipGetLastError();
}
hipError_t launch_sum_parts(const float* parts, int n_parts, long n, float* out, hipStream_t s) {
    hipLaunchKernelGGL(sum_parts_kernel, dim3(grid_for(n / 4)), dim3(256), 0, s, parts, n_parts, n / 4, out);  // n % 4 == 0
    return hipGetLastError();
}
hipError_t launch_scale_pair(const float* a, const float* b, float* out, hipStream_t s) {
    hipLaunchKernelGGL(scale_pair_kernel, dim3(1), dim3(64), 0, s, a, b, out);
    return hipGetLastError();
}
hipError_t launch_scatter_add_dead(const int32_t* dl, int nd, int D, const float* dWd, const float* dWe, const float* dbe,
                                   float* gW_dec, float* gW_encT, float* gb_enc, int lat_lo, int lat_hi, hipStream_t s) {
    if (nd <= 0) return hipSuccess;
    hipLaunchKernelGGL(scatter_add_dead_kernel, dim3((nd + 3) / 4), dim3(256), 0, s, dl, nd, D, dWd, dWe, dbe, gW_dec,
                       gW_encT, gb_enc, lat_lo, lat_hi);
    return hipGetLastError();
}
