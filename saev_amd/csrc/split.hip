// fp32 -> (hi, lo) fp16 operand splitting for the f16x3 encoder (gemm_encode_f16x3.hip), written directly in
// the encoder's LDS image order so the encoder can stream it with perfectly coalesced global_load_lds:
//
//   operand (rows R, k) -> blocks of 256 rows x one 16-wide k-step = one 16 KB "slot image":
//       image(blk, ks) at  ((blk * nks + ks) * 256 * 32) halfs,   nks = Dp / 16
//       inside: row rl (0..255) owns 32 halfs = 4 chunks of 8; logical chunk c = 2*part + h
//               (part 0 = hi, 1 = lo; h = which half of the k-step: k%16 < 8 or >= 8)
//               is stored at position c ^ ((rl >> 2) & 3)   <- the bank-conflict swizzle of the fragment reads
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
template <int MODE>
__global__ __launch_bounds__(1024) void split_rows_kernel(const float* __restrict__ x, int n, int D, int nks, float scale,
                                                          const float* __restrict__ scale_dev, const float* __restrict__ mu,
                                                          _Float16* __restrict__ xs) {
    if (scale_dev != nullptr) scale *= *scale_dev;
    const int blk = blockIdx.x / nks, ks = blockIdx.x % nks;
    const int i = threadIdx.x;           // chunk index inside the image
    const int rl = i >> 2, p = i & 3;
    const int c = p ^ ((rl >> 2) & 3);
    const int part = c >> 1, h = c & 1;
    const int r = blk * 256 + rl, k = MODE != 0 ? ks * 32 + c * 8 : ks * 16 + h * 8;
    float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (r < n && k < D) {  // D % 4 == 0: load in two float4s, the second may fall off the end
        f32x4 a = *reinterpret_cast<const f32x4*>(x + (size_t)r * D + k);
        if (mu != nullptr) a -= *reinterpret_cast<const f32x4*>(mu + k);  // centred images (f16r)
        v[0] = a[0] * scale; v[1] = a[1] * scale; v[2] = a[2] * scale; v[3] = a[3] * scale;
        if (k + 4 < D) {
            f32x4 b = *reinterpret_cast<const f32x4*>(x + (size_t)r * D + k + 4);
            if (mu != nullptr) b -= *reinterpret_cast<const f32x4*>(mu + k + 4);
            v[4] = b[0] * scale; v[5] = b[1] * scale; v[6] = b[2] * scale; v[7] = b[3] * scale;
        }
    }
    reinterpret_cast<half8*>(xs + (size_t)blockIdx.x * 256 * 32)[i] = pack8<MODE>(v, part);
}

// one workgroup = one image of W_enc^T: 256 latents x 16 k (32 k for bf16).  The k-rows of W_enc (1 KB each) are
// read coalesced into LDS, then every thread assembles its chunk from a column.
template <int MODE>
__global__ __launch_bounds__(1024) void split_wT_kernel(const float* __restrict__ W, int D, int S, int nks, float scale,
                                                        const float* __restrict__ scale_dev, _Float16* __restrict__ ws) {
    if (scale_dev != nullptr) scale *= *scale_dev;
    constexpr int KS = MODE != 0 ? 32 : 16;
    __shared__ float tile[KS][257];
    const int blk = blockIdx.x / nks, ks = blockIdx.x % nks;
    const int s0 = blk * 256, k0 = ks * KS;
    for (int q = threadIdx.x; q < KS * 256; q += 1024) {
        const int kk = q >> 8, sl = q & 255;
        const int k = k0 + kk, s = s0 + sl;
        tile[kk][sl] = (k < D && s < S) ? W[(size_t)k * S + s] * scale : 0.f;
    }
    __syncthreads();
    const int i = threadIdx.x;
    const int rl = i >> 2, p = i & 3;
    const int c = p ^ ((rl >> 2) & 3);
    const int part = c >> 1, h = c & 1;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = tile[(MODE != 0 ? c : h) * 8 + e][rl];
    reinterpret_cast<half8*>(ws + (size_t)blockIdx.x * 256 * 32)[i] = pack8<MODE>(v, part);
}

}  // namespace

hipError_t launch_split_rows(const float* x, int n, int D, int Dp, void* xs, int mode, hipStream_t stream, float scale,
                             const float* scale_dev, const float* mu) {
    const int nks = Dp / (mode != 0 ? 32 : 16), nblk = (n + 255) / 256;
    if (nblk <= 0) return hipSuccess;
    _Float16* o = reinterpret_cast<_Float16*>(xs);
    if (mode == 1) hipLaunchKernelGGL(split_rows_kernel<1>, dim3(nblk * nks), dim3(1024), 0, stream, x, n, D, nks, scale, scale_dev, mu, o);
    else if (mode == 2) hipLaunchKernelGGL(split_rows_kernel<2>, dim3(nblk * nks), dim3(1024), 0, stream, x, n, D, nks, scale, scale_dev, mu, o);
    else hipLaunchKernelGGL(split_rows_kernel<0>, dim3(nblk * nks), dim3(1024), 0, stream, x, n, D, nks, scale, scale_dev, mu, o);
    return hipGetLastError();
}

hipError_t launch_split_wT(const float* W, int D, int S, int S_pad, int Dp, float scale, void* ws, int mode,
                           hipStream_t stream, const float* scale_dev) {
    const int nks = Dp / (mode != 0 ? 32 : 16);
    const dim3 grid((S_pad / 256) * nks);
    _Float16* o = reinterpret_cast<_Float16*>(ws);
    if (mode == 1) hipLaunchKernelGGL(split_wT_kernel<1>, grid, dim3(1024), 0, stream, W, D, S, nks, scale, scale_dev, o);
    else if (mode == 2) hipLaunchKernelGGL(split_wT_kernel<2>, grid, dim3(1024), 0, stream, W, D, S, nks, scale, scale_dev, o);
    else hipLaunchKernelGGL(split_wT_kernel<0>, grid, dim3(1024), 0, stream, W, D, S, nks, scale, scale_dev, o);
    return hipGetLastError();
}
