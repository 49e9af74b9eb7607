// What does each stream of the encoder's contraction loop cost?  The shipped loop of encode_m16_kernel<2> (256 x 256 workgroup tile,
// eight waves as 2 x 4, wave tile 128 x 64, four-slot LDS ring filled by global_load_lds three k-steps ahead, one barrier per
// k-step) with single streams switched off, on random and on all-zero operands, plus one structural variant:
//   NOBAR    no s_barrier in the loop (results are garbage: timing only)
//   NOLDS    the fragments are read once per tile instead of once per k-step
//   NOSTAGE  no global_load_lds inside the loop
//   PAIR     two k-steps per barrier: the ring is two 64 KB super-slots, the next one is staged while this one is used
//   PREF     fragments of the next k-step's first two MFMA groups are read during the last two groups of this one (second register
//            set), the barrier sits after group 5 instead of between the steps, the staging after group 0: no step boundary at which
//            both waves of a SIMD wait for LDS with the matrix pipe empty
// Every kernel also reports shader cycles per tile (s_memtime) next to the wall time, i.e. the clock it ran at.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench/enc_loop2.hip -o /tmp/enc_loop2 && /tmp/enc_loop2
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <type_traits>
#include <vector>

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

struct __attribute__((aligned(16))) KSlot {
    _Float16 a[256][32];
    _Float16 b[256][32];
};

enum { NOBAR = 1, NOLDS = 2, NOSTAGE = 4, PAIR = 8, NOMFMA = 16, PREF = 32, SNAKE = 64 };

__device__ __forceinline__ unsigned long long shader_cycles() { return __builtin_readcyclecounter(); }  // s_memtime: tick = shader cycle

__global__ __launch_bounds__(256) void copy_kernel(const float4* __restrict__ a, float4* __restrict__ b, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) b[i] = a[i];
}

template <int FL>
__global__ __launch_bounds__(512, 2) void loop_kernel(const _Float16* __restrict__ wimg, const _Float16* __restrict__ ximg, int nks,
                                                      int ntiles, float* out, unsigned long long* cyc) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    KSlot* slot = reinterpret_cast<KSlot*>(smem_raw);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ws = wid % 2, wb = wid / 2;
    const int l15 = lane & 15, kg = lane >> 4;
    const int arow0 = ws * 128 + l15, brow0 = wb * 64 + l15;
    const int coff = 8 * (kg ^ ((4 - (l15 >> 2)) & 3));
    const size_t img = 256 * 32;
    const int bb = blockIdx.x >> 2, sp = blockIdx.x & 3;
    const _Float16* x_imgs = ximg + (size_t)bb * nks * img;
    const uint32_t lane_off = (uint32_t)lane * 16u;
    const uint32_t lds_w = (uint32_t)(size_t)(__attribute__((address_space(3))) char*)&slot[0].a[0][0] + wid * 2048;
    auto stage = [&](int s, int tile, int ks) {
        const char* wsrc = reinterpret_cast<const char*>(wimg + ((size_t)(sp * ntiles + tile) * nks + ks) * img) + wid * 2048;
        const char* xsrc = reinterpret_cast<const char*>(x_imgs + (size_t)ks * img) + wid * 2048;
        const uint32_t la = lds_w + (uint32_t)s * (uint32_t)sizeof(KSlot);
        asm volatile(
            "s_mov_b32 m0, %0\n\ts_nop 0\n\t"
            "global_load_lds_dwordx4 %2, %3\n\tglobal_load_lds_dwordx4 %2, %3 offset:1024\n\t"
            "s_mov_b32 m0, %1\n\ts_nop 0\n\t"
            "global_load_lds_dwordx4 %2, %4\n\tglobal_load_lds_dwordx4 %2, %4 offset:1024"
            ::"s"(la), "s"(la + 16384u), "v"(lane_off), "s"(wsrc), "s"(xsrc)
            : "memory", "m0");
    };
    f32x4 acc[8][4];
    float total = 0.f;
    unsigned long long cycles = 0;
    half8 fa[3], fb[4];
    auto kstep_compute = [&](const KSlot& cs) {
        if constexpr (!(FL & NOLDS)) {
#pragma unroll
            for (int jb = 0; jb < 4; ++jb) fb[jb] = *reinterpret_cast<const half8*>(&cs.b[brow0 + 16 * jb][coff]);
            fa[0] = *reinterpret_cast<const half8*>(&cs.a[arow0][coff]);
            fa[1] = *reinterpret_cast<const half8*>(&cs.a[arow0 + 16][coff]);
        }
#pragma unroll
        for (int sb = 0; sb < 8; ++sb) {
            // SNAKE: odd latent blocks walk the row blocks 3..0, so the B operand of the last MFMA of a group is that of the first MFMA
            // of the next (fewer operand changes in front of the matrix pipe: does the power-limited clock notice?)
            constexpr int j0 = 0;
            const int jf = ((FL & SNAKE) && (sb & 1)) ? 3 : 0;
            if constexpr (!(FL & NOMFMA)) acc[sb][jf] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[sb % 3], fb[jf], acc[sb][jf], 0, 0, 0);
            else acc[sb][0][0] += (float)fa[sb % 3][0] + (float)fb[0][0];
            (void)j0;
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (!(FL & NOLDS)) {
                if (sb + 2 < 8) fa[(sb + 2) % 3] = *reinterpret_cast<const half8*>(&cs.a[arow0 + 16 * (sb + 2)][coff]);
            }
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (!(FL & NOMFMA)) {
#pragma unroll
                for (int jq = 1; jq < 4; ++jq) {
                    const int jb = ((FL & SNAKE) && (sb & 1)) ? 3 - jq : jq;
                    acc[sb][jb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[sb % 3], fb[jb], acc[sb][jb], 0, 0, 0);
                }
            } else {
#pragma unroll
                for (int jb = 1; jb < 4; ++jb) acc[sb][jb][0] += (float)fb[jb][0];
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    for (int tile = 0; tile < ntiles; ++tile) {
        const unsigned long long c0 = shader_cycles();
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        stage(0, tile, 0);
        stage(1, tile, 1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if constexpr (FL & NOLDS) {
#pragma unroll
            for (int jb = 0; jb < 4; ++jb) fb[jb] = *reinterpret_cast<const half8*>(&slot[0].b[brow0 + 16 * jb][coff]);
#pragma unroll
            for (int i = 0; i < 3; ++i) fa[i] = *reinterpret_cast<const half8*>(&slot[0].a[arow0 + 16 * i][coff]);
        }
        if constexpr (FL & PREF) {
            half8 fbx[2][4], fax[4];
            auto rd_b = [&](int set, const KSlot& cs) {
#pragma unroll
                for (int jb = 0; jb < 4; ++jb) fbx[set][jb] = *reinterpret_cast<const half8*>(&cs.b[brow0 + 16 * jb][coff]);
            };
            auto rd_a = [&](int set, const KSlot& cs, int sb) { fax[set] = *reinterpret_cast<const half8*>(&cs.a[arow0 + 16 * sb][coff]); };
            rd_b(0, slot[0]); rd_a(0, slot[0], 0); rd_a(1, slot[0], 1);
            stage(2, tile, 2);
            auto step = [&](int t, auto PAR_, auto TAIL_) {
                constexpr int par = decltype(PAR_)::value;
                constexpr bool tail = decltype(TAIL_)::value;  // steady steps carry no condition at all
                const KSlot& cs = slot[t & 3];
                const KSlot& ns = slot[(t + 1) & 3];
#pragma unroll
                for (int sb = 0; sb < 8; ++sb) {
                    acc[sb][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fax[sb % 4], fbx[par][0], acc[sb][0], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    if (sb == 0) { if (!tail || t + 3 < nks) stage((t + 3) & 3, tile, t + 3); }
                    if (sb + 2 < 8) rd_a((sb + 2) % 4, cs, sb + 2);
                    else if (!tail || t + 1 < nks) rd_a((sb + 2) % 4, ns, sb - 6);
                    if (sb == 6 && (!tail || t + 1 < nks)) rd_b(par ^ 1, ns);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int jb = 1; jb < 4; ++jb) acc[sb][jb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fax[sb % 4], fbx[par][jb], acc[sb][jb], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    if (sb == 5) {
                        if (!tail || t + 3 < nks) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                        else if (t + 2 < nks) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                        if constexpr (!(FL & NOBAR)) __builtin_amdgcn_s_barrier();
                    }
                }
            };
            int t = 0;
            for (; t + 4 < nks; t += 2) {
                step(t, std::integral_constant<int, 0>(), std::false_type());
                step(t + 1, std::integral_constant<int, 1>(), std::false_type());
            }
            for (; t < nks; t += 2) {
                step(t, std::integral_constant<int, 0>(), std::true_type());
                step(t + 1, std::integral_constant<int, 1>(), std::true_type());
            }
            __syncthreads();
        } else
        if constexpr (FL & PAIR) {
            const int nu = nks / 2;
            for (int u = 0; u < nu; ++u) {
                const int cur = 2 * (u & 1), nxt = 2 - cur;
                if constexpr (!(FL & NOSTAGE)) {
                    if (u + 1 < nu) { stage(nxt, tile, 2 * u + 2); stage(nxt + 1, tile, 2 * u + 3); }
                }
                kstep_compute(slot[cur]);
                kstep_compute(slot[cur + 1]);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if constexpr (!(FL & NOBAR)) __builtin_amdgcn_s_barrier();
            }
        } else {
            if constexpr (!(FL & NOSTAGE)) stage(2, tile, 2);
            for (int t = 0; t < nks; ++t) {
                if constexpr (!(FL & NOSTAGE)) {
                    if (t + 3 < nks) stage((t + 3) & 3, tile, t + 3);
                }
                kstep_compute(slot[t & 3]);
                if constexpr (!(FL & NOSTAGE)) {
                    if (t + 3 < nks) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                    else if (t + 2 < nks) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
                if constexpr (!(FL & NOBAR)) __builtin_amdgcn_s_barrier();
            }
        }
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) total += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
        cycles += shader_cycles() - c0;
    }
    out[(size_t)blockIdx.x * 512 + tid] = total;
    if (tid == 0) cyc[blockIdx.x] = cycles;
}

int main() {
    const int nks = 32, ntiles = 32, nbb = 64, nsp = 4;
    const size_t img = 256 * 32;
    const size_t wn = (size_t)nsp * ntiles * nks * img, xn = (size_t)nbb * nks * img;
    std::vector<_Float16> h(wn > xn ? wn : xn);
    srand(1);
    for (auto& v : h) v = (_Float16)((rand() % 2001 - 1000) * 0.004f);
    _Float16 *w, *x;
    float* out;
    unsigned long long* cyc;
    hipMalloc(&w, wn * 2); hipMalloc(&x, xn * 2); hipMalloc(&out, 512 * 512 * 4); hipMalloc(&cyc, 256 * 8);
    const double flops = 2.0 * 16384 * 1024 * 32768;
    const int smem = 4 * (int)sizeof(KSlot);
    // filler: a memory-bound stream kernel standing in for the ~2.2 ms of gather / stream kernels between two encoder launches
    float4* fill_a; float4* fill_b;
    const size_t fill_n = (size_t)1 << 28;  // 4 GiB each way per launch at 16 B per element
    hipMalloc(&fill_a, fill_n * 16); hipMalloc(&fill_b, fill_n * 16);
    hipMemset(fill_a, 0, fill_n * 16);
    auto run = [&](auto kern, const char* name) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
        // (1) back to back: 300 launches, the last 150 timed as one region (the power controller has settled by then)
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        for (int rep = 0; rep < 150; ++rep) hipLaunchKernelGGL(kern, dim3(nbb * nsp), dim3(512), smem, 0, w, x, nks, ntiles, out, cyc);
        hipEventRecord(e0, 0);
        for (int rep = 0; rep < 150; ++rep) hipLaunchKernelGGL(kern, dim3(nbb * nsp), dim3(512), smem, 0, w, x, nks, ntiles, out, cyc);
        hipEventRecord(e1, 0);
        hipDeviceSynchronize();
        float ms_bb = 0.f;
        hipEventElapsedTime(&ms_bb, e0, e1);
        ms_bb /= 150;
        unsigned long long hc[256];
        hipMemcpy(hc, cyc, sizeof(hc), hipMemcpyDeviceToHost);
        double sc = 0;
        for (auto c : hc) sc += (double)c;
        const double mhz_bb = sc / 256 / (ms_bb * 1e3);
        // (2) duty cycle of the train step: every launch is followed by ~2.2 ms of a copy kernel; the encoder launches are timed one by one
        const int N = 60;
        std::vector<hipEvent_t> a(N), b(N);
        for (int i = 0; i < N; ++i) { hipEventCreate(&a[i]); hipEventCreate(&b[i]); }
        for (int i = 0; i < N; ++i) {
            hipEventRecord(a[i], 0);
            hipLaunchKernelGGL(kern, dim3(nbb * nsp), dim3(512), smem, 0, w, x, nks, ntiles, out, cyc);
            hipEventRecord(b[i], 0);
            hipLaunchKernelGGL(copy_kernel, dim3(256 * 8), dim3(256), 0, 0, fill_a, fill_b, fill_n / 2);
        }
        hipDeviceSynchronize();
        double ms_dc = 0;
        for (int i = N / 2; i < N; ++i) { float t; hipEventElapsedTime(&t, a[i], b[i]); ms_dc += t; }
        ms_dc /= (N - N / 2);
        float t_all; hipEventElapsedTime(&t_all, a[N / 2], a[N - 1]);
        hipMemcpy(hc, cyc, sizeof(hc), hipMemcpyDeviceToHost);
        sc = 0;
        for (auto c : hc) sc += (double)c;
        printf("%-30s back-to-back %.3f ms %5.0f TF/s %4.0f MHz | in a 1:2 duty cycle %.3f ms %5.0f TF/s %4.0f MHz (period %.2f ms)  (%s)\n", name, ms_bb,
               flops / ms_bb * 1e-9, mhz_bb, ms_dc, flops / ms_dc * 1e-9, sc / 256 / (ms_dc * 1e3), t_all / (N - 1 - N / 2), hipGetErrorString(hipGetLastError()));
        for (int i = 0; i < N; ++i) { hipEventDestroy(a[i]); hipEventDestroy(b[i]); }
    };
    for (int data = 0; data < 2; ++data) {
        if (data == 1) for (auto& v : h) v = (_Float16)0.f;
        hipMemcpy(w, h.data(), wn * 2, hipMemcpyHostToDevice);
        hipMemcpy(x, h.data(), xn * 2, hipMemcpyHostToDevice);
        printf("---- operands: %s\n", data == 0 ? "random" : "zeros");
        run(loop_kernel<0>, "shipped loop");
        run(loop_kernel<NOBAR>, "NOBAR");
        run(loop_kernel<NOLDS>, "NOLDS");
        run(loop_kernel<NOSTAGE>, "NOSTAGE");
        run(loop_kernel<NOLDS | NOSTAGE>, "NOLDS NOSTAGE (MFMA+barrier)");
        run(loop_kernel<NOLDS | NOSTAGE | NOBAR>, "MFMA only");
        run(loop_kernel<NOMFMA>, "NOMFMA");
        run(loop_kernel<PAIR>, "PAIR");
        run(loop_kernel<SNAKE>, "SNAKE");
        run(loop_kernel<0>, "shipped loop (again)");
        run(loop_kernel<SNAKE>, "SNAKE (again)");
        run(loop_kernel<SNAKE | NOLDS | NOSTAGE | NOBAR>, "SNAKE MFMA only");
        run(loop_kernel<NOLDS | NOSTAGE | NOBAR>, "MFMA only (again)");
        run(loop_kernel<PREF>, "PREF");
        run(loop_kernel<0>, "shipped loop (3)");
        run(loop_kernel<PREF>, "PREF (again)");
        run(loop_kernel<PREF | NOBAR>, "PREF NOBAR");
    }
    return 0;
}
