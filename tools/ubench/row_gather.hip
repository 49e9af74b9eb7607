// Floor of the sparse decode's access pattern: every batch row sums k = 32 rows (4 KB each) of a 32 768 x 1024 fp32 matrix (134 MB:
// past the 4 MB L2 of an XCD, inside the 256 MB Infinity Cache), latent indices drawn uniformly from `used` latents, ascending per
// row.  Nothing but the gathers and the sum (one store of 4 KB per row), in the two layouts the step has:
//   rows4   decode_q_kernel's: a workgroup of four waves per batch row, one float4 per lane and code, ALL 32 gathers in flight
//   rows1   decode_kernel's:   one wave per batch row, four float4 per lane and code, four codes (16 loads) in flight
//   slices  dw_slices / refine_slices': the matrix slice-major [D / 32][S][32], XCD x walks slices x, x + 8, ...; an eight-lane group
//           sums one (row, slice): 32 gathers of 128 B out of a 4 MB slice (two 2 MB halves by latent range, one after the other;
//           slices1: the whole slice in one launch)
// Prints the gathered bytes per second: what "whole-row gathers at the fabric rate" and "line gathers out of L2" mean in DESIGN.md.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench/row_gather.hip -o /tmp/row_gather && /tmp/row_gather
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

constexpr int S = 32768, D = 1024, B = 16384, K = 32;

__device__ __forceinline__ f32x4 bload(__amdgpu_buffer_rsrc_t r, uint32_t voff, uint32_t soff) {
    const i32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0);
    return f32x4{__int_as_float(v[0]), __int_as_float(v[1]), __int_as_float(v[2]), __int_as_float(v[3])};
}

__global__ __launch_bounds__(256) void rows4_kernel(const float* __restrict__ W, const int* __restrict__ idx, float* __restrict__ out) {
    const int lane = threadIdx.x & 63, q = threadIdx.x, row = blockIdx.x;
    const int my = lane < K ? idx[(size_t)row * K + lane] : 0;
    const __amdgpu_buffer_rsrc_t res = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(W), 0, (uint32_t)S * D * 4u, 0x00020000);
    f32x4 w[K];
#pragma unroll
    for (int j = 0; j < K; ++j) w[j] = bload(res, (uint32_t)q * 16u, (uint32_t)__builtin_amdgcn_readlane(my, j) * (uint32_t)(D * 4));
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < K; ++j) acc += w[j];
    reinterpret_cast<f32x4*>(out + (size_t)row * D)[q] = acc;
}

__global__ __launch_bounds__(256) void rows1_kernel(const float* __restrict__ W, const int* __restrict__ idx, float* __restrict__ out) {
    const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int my = lane < K ? idx[(size_t)row * K + lane] : 0;
    f32x4 acc[4] = {};
#pragma unroll 4
    for (int j = 0; j < K; ++j) {
        const int i = __shfl(my, j, 64);
        const f32x4* wr = reinterpret_cast<const f32x4*>(W + (size_t)i * D);
#pragma unroll
        for (int n = 0; n < 4; ++n) acc[n] += wr[lane + 64 * n];
    }
#pragma unroll
    for (int n = 0; n < 4; ++n) reinterpret_cast<f32x4*>(out + (size_t)row * D)[lane + 64 * n] = acc[n];
}

// WS: [D / 32][S][32 floats].  Grid: 8 XCDs x 4 slices each x (B / 32 rows per workgroup); a workgroup = 32 eight-lane groups = 32 rows
// of one slice; `half` = 0 / 1: only the codes below / from S / 2 (each launch keeps a 2 MB half slice per XCD hot)
__global__ __launch_bounds__(256) void slices_kernel(const float* __restrict__ WS, const int* __restrict__ idx, float* __restrict__ outS, int half,
                                                    int wg_per_slice) {
    const int lane = threadIdx.x & 63, gi = threadIdx.x >> 3, li = lane & 7;
    const int xcd = blockIdx.x & 7, qq = blockIdx.x >> 3;
    const int slice = xcd + 8 * (qq / wg_per_slice);
    const int row = (qq % wg_per_slice) * 32 + gi;
    const __amdgpu_buffer_rsrc_t res = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(WS) + (size_t)slice * S * 32, 0, (uint32_t)S * 128u, 0x00020000);
    const int* ir = idx + (size_t)row * K;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const int lo = half == 2 ? 0 : half * (S / 2), hi = half == 2 ? S : lo + S / 2;  // half = 2: the whole 4 MB slice in one launch
#pragma unroll
    for (int j0 = 0; j0 < K; j0 += 8) {
        f32x4 w[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = ir[j0 + u];
            // (a code outside this half: offset past the buffer, the load returns zeros without touching memory)
            w[u] = bload(res, ((i >= lo && i < hi) ? (uint32_t)i * 128u : 0xFFFFFF00u) | ((uint32_t)li * 16u), 0);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += w[u];
    }
    f32x4* o = reinterpret_cast<f32x4*>(outS) + ((size_t)slice * B + row) * 8 + li;
    if (half == 1) acc += *o;
    *o = acc;
}


// slices1 grown step by step into the real slice decode (sparse.hip: decode_s_kernel): which addition costs what?
//   V = 1: + val (16 broadcast code loads instead of 32), x_hat = sum val_j w_j, all 32 gathers in flight
//   V = 2: + x, the loss arithmetic, g; stores of x_hat slice-major... (one more 128-byte store per group)
//   V = 3: + the 32 dval shares (dots, DPP reduce-scatter) and their store
template <int V>
__global__ __launch_bounds__(256) void slices_full_kernel(const float* __restrict__ WS, const int* __restrict__ idx, const float* __restrict__ val,
                                                         const float* __restrict__ x, float* __restrict__ outS, float* __restrict__ gS,
                                                         float* __restrict__ dvp, int wg_per_slice) {
    const int lane = threadIdx.x & 63, gi = threadIdx.x >> 3, li = lane & 7;
    const int xcd = blockIdx.x & 7, qq = blockIdx.x >> 3;
    const int slice = xcd + 8 * (qq / wg_per_slice);
    const int row = (qq % wg_per_slice) * 32 + gi;
    const __amdgpu_buffer_rsrc_t res = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(WS) + (size_t)slice * S * 32, 0, (uint32_t)S * 128u, 0x00020000);
    const i32x4* ir = reinterpret_cast<const i32x4*>(idx + (size_t)row * K);
    const f32x4* vr = reinterpret_cast<const f32x4*>(val + (size_t)row * K);
    i32x4 ci[8];
    f32x4 cv[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) { ci[u] = ir[u]; cv[u] = vr[u]; }
    f32x4 w[32];
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
        for (int h = 0; h < 4; ++h) w[4 * u + h] = bload(res, (uint32_t)ci[u][h] * 128u | ((uint32_t)li * 16u), 0);
    f32x4 xv = {0.f, 0.f, 0.f, 0.f};
    if (V >= 2) xv = reinterpret_cast<const f32x4*>(x + (size_t)row * D)[slice * 8 + li];
    __builtin_amdgcn_sched_barrier(0);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
        for (int h = 0; h < 4; ++h) acc += cv[u][h] * w[4 * u + h];
    const size_t o = ((size_t)slice * B + row) * 8 + li;
    reinterpret_cast<f32x4*>(outS)[o] = acc;
    if (V >= 2) {
        f32x4 g;
        float sse = 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) { const float t = acc[e] / 3.f - xv[e] / 3.f; sse += t * t; g[e] = 1e-4f * t; }
        reinterpret_cast<f32x4*>(gS)[o] = g;
        if (sse == 12345.f) outS[0] = 1.f;
        if (V >= 3) {
            f32x4 sh;
#pragma unroll
            for (int h = 0; h < 4; ++h) {
                float p[8];
#pragma unroll
                for (int uu = 0; uu < 8; ++uu) { const f32x4& ww = w[4 * uu + h]; p[uu] = (g[0] * ww[0] + g[1] * ww[1]) + (g[2] * ww[2] + g[3] * ww[3]); }
                float r = 0.f;
#pragma unroll
                for (int uu = 0; uu < 8; ++uu) r += p[uu];  // (stand-in for the reduce-scatter: same number of adds)
                r += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(r), 0x141, 0xF, 0xF, true));
                r += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(r), 0x4E, 0xF, 0xF, true));
                r += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(r), 0xB1, 0xF, 0xF, true));
                sh[h] = r;
            }
            reinterpret_cast<f32x4*>(dvp)[o] = sh;
        }
    }
}

int main() {
    float *W, *out;
    int* idx;
    hipMalloc(&W, (size_t)S * D * 4); hipMalloc(&out, (size_t)B * D * 4); hipMalloc(&idx, (size_t)B * K * 4);
    hipMemset(W, 0, (size_t)S * D * 4);
    float *val, *xin, *gS, *dvp;
    hipMalloc(&val, (size_t)B * K * 4); hipMalloc(&xin, (size_t)B * D * 4); hipMalloc(&gS, (size_t)B * D * 4); hipMalloc(&dvp, (size_t)B * D * 4);
    hipMemset(val, 0, (size_t)B * K * 4); hipMemset(xin, 0, (size_t)B * D * 4);
    for (int used : {14000, 27000, 32768}) {
        std::mt19937 rng(1);
        std::vector<int> perm(S);
        for (int i = 0; i < S; ++i) perm[i] = i;
        std::shuffle(perm.begin(), perm.end(), rng);
        std::vector<int> h((size_t)B * K);
        for (int b = 0; b < B; ++b) {
            for (int j = 0; j < K; ++j) h[(size_t)b * K + j] = perm[rng() % used];
            std::sort(h.begin() + (size_t)b * K, h.begin() + (size_t)(b + 1) * K);
        }
        hipMemcpy(idx, h.data(), h.size() * 4, hipMemcpyHostToDevice);
        const double bytes = (double)B * K * D * 4;
        auto time = [&](const char* name, auto launch) {
            float best = 1e9f;
            for (int rep = 0; rep < 12; ++rep) {
                hipEvent_t e0, e1;
                hipEventCreate(&e0); hipEventCreate(&e1);
                hipEventRecord(e0, 0);
                launch();
                hipEventRecord(e1, 0);
                hipDeviceSynchronize();
                float ms;
                hipEventElapsedTime(&ms, e0, e1);
                if (rep >= 2 && ms < best) best = ms;
            }
            printf("latents in use %5d  %-8s %.3f ms  %5.1f TB/s of gathered rows  (%s)\n", used, name, best, bytes / best * 1e-9, hipGetErrorString(hipGetLastError()));
        };
        time("rows4", [&] { hipLaunchKernelGGL(rows4_kernel, dim3(B), dim3(256), 0, 0, W, idx, out); });
        time("rows1", [&] { hipLaunchKernelGGL(rows1_kernel, dim3(B / 4), dim3(256), 0, 0, W, idx, out); });
        time("slices1", [&] {  // one launch per slice over all latents: a 4 MB slice in a 4 MB L2
            const int wps = B / 32;
            hipLaunchKernelGGL(slices_kernel, dim3(8 * 4 * wps), dim3(256), 0, 0, W, idx, out, 2, wps);
        });
        time("full V=1", [&] { hipLaunchKernelGGL(slices_full_kernel<1>, dim3(8 * 4 * (B / 32)), dim3(256), 0, 0, W, idx, val, xin, out, gS, dvp, B / 32); });
        time("full V=2", [&] { hipLaunchKernelGGL(slices_full_kernel<2>, dim3(8 * 4 * (B / 32)), dim3(256), 0, 0, W, idx, val, xin, out, gS, dvp, B / 32); });
        time("full V=3", [&] { hipLaunchKernelGGL(slices_full_kernel<3>, dim3(8 * 4 * (B / 32)), dim3(256), 0, 0, W, idx, val, xin, out, gS, dvp, B / 32); });
        time("slices", [&] {
            const int wps = B / 32;
            hipLaunchKernelGGL(slices_kernel, dim3(8 * 4 * wps), dim3(256), 0, 0, W, idx, out, 0, wps);
            hipLaunchKernelGGL(slices_kernel, dim3(8 * 4 * wps), dim3(256), 0, 0, W, idx, out, 1, wps);
        });
    }
    return 0;
}
