// How fast can v_mfma_f32_32x32x16_f16 be issued on gfx950 in the register shapes the encoder kernel can choose from?
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_issue.hip -o /tmp/mfma_issue && /tmp/mfma_issue
// Variants: waves per SIMD (1 or 2), accumulators per wave (8 = 128 registers, as the shipped kernel; 16 = 256),
// barrier pattern (none / one per 16 MFMAs in lock-step / the two waves of a SIMD alternating by half-steps), operand
// data (random fp16 in the encoder's range, or zeros: the clock the chip sustains depends on it).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// MODE 0: no barrier; 1: barrier after every 16 MFMAs, all waves in step; 2: two groups (waves 0-3 / 4-7) alternate
template <int THREADS, int NACC, int MODE>
__global__ __launch_bounds__(THREADS) void k(const half8* frag, float* out, int iters) {
    const int tid = threadIdx.x;
    const int grp = __builtin_amdgcn_readfirstlane(tid >> 8);
    half8 a[4], b[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        a[i] = frag[(tid * 8 + i) & 4095];
        b[i] = frag[(tid * 8 + 4 + i) & 4095];
    }
    f32x16 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    if (MODE == 2 && grp == 1) __builtin_amdgcn_s_barrier();
    for (int it = 0; it < iters; ++it) {
        if (MODE == 2) __builtin_amdgcn_s_barrier();  // the other group's MFMA half
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int u = 0; u < 16; ++u)
            acc[u % NACC] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[u & 3], b[(u >> 2) & 3], acc[u % NACC], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        if (MODE != 0) __builtin_amdgcn_s_barrier();
    }
    if (MODE == 2 && grp == 0) __builtin_amdgcn_s_barrier();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NACC; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * THREADS + tid] = s;
}

// the same flops as 16x16x32 instructions (twice as many, 4 accumulator registers each)
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int THREADS>
__global__ __launch_bounds__(THREADS) void k16(const half8* frag, float* out, int iters) {
    const int tid = threadIdx.x;
    half8 a[4], b[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        a[i] = frag[(tid * 8 + i) & 4095];
        b[i] = frag[(tid * 8 + 4 + i) & 4095];
    }
    f32x4 acc[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 32; ++u)
            acc[u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[u & 3], b[(u >> 2) & 3], acc[u], 0, 0, 0);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 32; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * THREADS + tid] = s;
}
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
template <int THREADS>
__global__ __launch_bounds__(THREADS) void kbf(const half8* frag, float* out, int iters) {
    const int tid = threadIdx.x;
    bf16x8 a[4], b[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        a[i] = __builtin_bit_cast(bf16x8, frag[(tid * 8 + i) & 4095]);
        b[i] = __builtin_bit_cast(bf16x8, frag[(tid * 8 + 4 + i) & 4095]);
    }
    f32x16 acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u)
            acc[u % 8] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[u & 3], b[(u >> 2) & 3], acc[u % 8], 0, 0, 0);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * THREADS + tid] = s;
}

int main() {
    half8* frag;
    float* out;
    hipMalloc(&frag, 4096 * sizeof(half8));
    hipMalloc(&out, 256 * 512 * 4);
    const int iters = 20000;
    // zero = 0: random fp16; 1: zeros; 2..4: random with the low 3 / 5 / 7 mantissa bits cleared (does the power the
    // multipliers draw -- and with it the sustained clock -- follow the number of significant bits?)
    for (int zero = 0; zero < 5; ++zero) {
        std::vector<_Float16> h(4096 * 8);
        srand(1);
        for (auto& v : h) v = zero == 1 ? (_Float16)0.f : (_Float16)((rand() / (float)RAND_MAX * 2.f - 1.f) * 3.0f);
        if (zero >= 2) {
            const unsigned short mask = (unsigned short)(0xFFFFu << (zero == 2 ? 3 : zero == 3 ? 5 : 7));
            for (auto& v : h) { unsigned short u; memcpy(&u, &v, 2); u &= mask; memcpy(&v, &u, 2); }
        }
        hipMemcpy(frag, h.data(), h.size() * 2, hipMemcpyHostToDevice);
        auto run = [&](auto kern, int threads, const char* name) {
            hipLaunchKernelGGL(kern, dim3(256), dim3(threads), 0, 0, frag, out, 200);
            hipDeviceSynchronize();
            hipEvent_t e0, e1;
            hipEventCreate(&e0);
            hipEventCreate(&e1);
            float best = 1e9f;
            for (int rep = 0; rep < 3; ++rep) {
                hipEventRecord(e0, 0);
                hipLaunchKernelGGL(kern, dim3(256), dim3(threads), 0, 0, frag, out, iters);
                hipEventRecord(e1, 0);
                hipDeviceSynchronize();
                float ms = 0.f;
                hipEventElapsedTime(&ms, e0, e1);
                best = ms < best ? ms : best;
            }
            const double flops = 256.0 * (threads / 64) * iters * 16 * 32768.0;
            const double per_simd = (double)(threads / 256) * iters * 16;  // MFMAs per SIMD
            printf("%-58s %s  %7.1f TFLOP/s  %6.1f ns/MFMA/SIMD (%.1f clk at 2.4 GHz)\n", name, zero == 0 ? "random" : zero == 1 ? "zeros " : zero == 2 ? "7 bits" : zero == 3 ? "5 bits" : "3 bits",
                   flops / (best * 1e-3) / 1e12, best * 1e6 / per_simd, best * 1e6 / per_simd * 2.4);
        };
        run(k<256, 8, 0>, 256, "1 wave/SIMD, 8 acc (128 regs), no barrier");
        run(k<256, 16, 0>, 256, "1 wave/SIMD, 16 acc (256 regs), no barrier");
        run(k<512, 8, 0>, 512, "2 waves/SIMD, 8 acc each, no barrier");
        run(k<512, 8, 1>, 512, "2 waves/SIMD, 8 acc each, barrier per 16 MFMAs, lock-step");
        run(k<512, 8, 2>, 512, "2 waves/SIMD, 8 acc each, alternating halves (2 barriers)");
        run(k<256, 16, 1>, 256, "1 wave/SIMD, 16 acc, barrier per 16 MFMAs");
        run(k16<512>, 512, "2 waves/SIMD, 16x16x32 f16 (32 per trip = same flops)");
        run(kbf<512>, 512, "2 waves/SIMD, 32x32x16 bf16 (same bit patterns)");
    }
    return 0;
}
