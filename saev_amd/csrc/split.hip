// fp32 -> (hi, lo) fp16 operand splitting for the f16x3 encoder (gemm_encode_f16x3.hip).
//   split_rows_kernel : x (n, D) fp32 -> xh, xl (n_pad, Dp) fp16; columns D..Dp are zero.
//   split_wT_kernel   : W_enc (D, S) fp32 -> wh, wl (S_pad, Dp) fp16, TRANSPOSED and scaled by `scale`;
//                       64 x 64 tiles through LDS so both the fp32 reads and the fp16 writes are coalesced.
// hi = fp16(a*scale) (round to nearest even), lo = fp16(a*scale - hi): together 22 significand bits.
#include "common.h"
#include "kernels.h"

namespace {

__global__ __launch_bounds__(256) void split_rows_kernel(const float* __restrict__ x, int n, int D, int Dp,
                                                         _Float16* __restrict__ xh, _Float16* __restrict__ xl) {
    // one thread handles 4 consecutive k of one row
    const long q = (long)blockIdx.x * 256 + threadIdx.x;
    const int per_row = Dp >> 2;
    const long total = (long)n * per_row;
    if (q >= total) return;
    const int r = (int)(q / per_row), c = (int)(q % per_row) * 4;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (c < D) v = *reinterpret_cast<const f32x4*>(x + (size_t)r * D + c);  // D % 4 == 0
    _Float16 h[4], l[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        h[e] = (_Float16)v[e];
        l[e] = (_Float16)(v[e] - (float)h[e]);
    }
    *reinterpret_cast<uint2*>(xh + (size_t)r * Dp + c) = *reinterpret_cast<uint2*>(h);
    *reinterpret_cast<uint2*>(xl + (size_t)r * Dp + c) = *reinterpret_cast<uint2*>(l);
}

__global__ __launch_bounds__(256) void split_wT_kernel(const float* __restrict__ W, int D, int S, int Dp, float scale,
                                                       _Float16* __restrict__ wh, _Float16* __restrict__ wl) {
    __shared__ float tile[64][65];
    const int s0 = blockIdx.x * 64, d0 = blockIdx.y * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int r = ty; r < 64; r += 4) {
        const int d = d0 + r, s = s0 + tx;
        tile[r][tx] = (d < D && s < S) ? W[(size_t)d * S + s] * scale : 0.f;
    }
    __syncthreads();
    for (int r = ty; r < 64; r += 4) {
        const int s = s0 + r, d = d0 + tx;
        if (d < Dp) {  // rows s >= S are written too (zeros): the padded buffer is fully defined
            const float v = tile[tx][r];
            const _Float16 h = (_Float16)v;
            wh[(size_t)s * Dp + d] = h;
            wl[(size_t)s * Dp + d] = (_Float16)(v - (float)h);
        }
    }
}

}  // namespace

hipError_t launch_split_rows(const float* x, int n, int D, int Dp, void* xh, void* xl, hipStream_t stream) {
    const long total = (long)n * (Dp >> 2);
    if (total <= 0) return hipSuccess;
    hipLaunchKernelGGL(split_rows_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, x, n, D, Dp,
                       reinterpret_cast<_Float16*>(xh), reinterpret_cast<_Float16*>(xl));
    return hipGetLastError();
}

hipError_t launch_split_wT(const float* W, int D, int S, int S_pad, int Dp, float scale, void* wh, void* wl,
                           hipStream_t stream) {
    hipLaunchKernelGGL(split_wT_kernel, dim3(S_pad / 64, (Dp + 63) / 64), dim3(256), 0, stream, W, D, S, Dp, scale,
                       reinterpret_cast<_Float16*>(wh), reinterpret_cast<_Float16*>(wl));
    return hipGetLastError();
}
