// Shared device/host helpers for the gfx950 SAE kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define WAVE 64

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// Order-preserving float <-> int32 key (for integer atomicMax on floats, radix select).
__device__ __forceinline__ int32_t f2key(float f) {
    int32_t b = __float_as_int(f);
    return b >= 0 ? b : (b ^ 0x7fffffff);
}
__device__ __forceinline__ float key2f(int32_t k) {
    return __int_as_float(k >= 0 ? k : (k ^ 0x7fffffff));
}
// Unsigned radix key: larger float -> larger uint32.
__device__ __forceinline__ uint32_t f2ukey(float f) {
    uint32_t b = __float_as_uint(f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float ukey2f(uint32_t k) {
    return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ int wave_sum_i(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

#define NEG_INF (-__builtin_huge_valf())

// remove_parallel_grads on one decoder-gradient row held across a wave (NV float4 per lane, zeros past the row's end):
// sc = <g, w> / ||w||^2 (0 when the projection is off or w = 0) and the sum of squares of the projected row g - sc w.
// The rpg pass (tail.hip), the backward kernels that leave {sc, sq} behind for a tail that projects inside Adam
// (sparse.hip, auxk.hip) and that Adam all use these explicit fmas, so the two routes give bit-identical updates.
__device__ __forceinline__ float rpg_apply(float g, float sc, float w) { return __builtin_fmaf(-sc, w, g); }
template <int NV>
__device__ __forceinline__ float rpg_row_stats(const f32x4 (&g)[NV], const f32x4 (&w)[NV], int project, float* sc_out) {
    float dot = 0.f, nsq = 0.f;
#pragma unroll
    for (int n = 0; n < NV; ++n)
#pragma unroll
        for (int e = 0; e < 4; ++e) { dot = __builtin_fmaf(g[n][e], w[n][e], dot); nsq = __builtin_fmaf(w[n][e], w[n][e], nsq); }
    dot = wave_sum(dot);
    nsq = wave_sum(nsq);
    const float sc = (project && nsq > 0.f) ? dot / nsq : 0.f;
    float sq = 0.f;
#pragma unroll
    for (int n = 0; n < NV; ++n)
#pragma unroll
        for (int e = 0; e < 4; ++e) { const float t = rpg_apply(g[n][e], sc, w[n][e]); sq = __builtin_fmaf(t, t, sq); }
    *sc_out = sc;
    return wave_sum(sq);
}
