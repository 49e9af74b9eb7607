// Can the encoder's TopK epilogue hide behind the OTHER wave's MFMAs?  
// The shipped kernel runs 32 k-steps of a tile (both waves of every SIMD in their MFMAs), then its epilogue (both waves in vector-ALU
// work, the matrix pipes idle: ~0.10 of 1.0 ms).  The only way to overlap the two inside one workgroup -- one barrier per k-step for
// all eight waves -- is to let waves 4-7 run HALF A TILE BEHIND waves 0-3: while one half is in its epilogue the other still issues
// MFMAs.  This micro-benchmark measures what that schedule can buy before anybody rewrites the kernel for it: the shipped loop
// (ring of four slots, global_load_lds three steps ahead, twelve fragment reads and 32 MFMAs per wave and k-step) with a synthetic
// epilogue of the real one's instruction mix (per accumulator value one v_fma, one v_cmp, one v_addc: 384 vector instructions per
// wave and tile, cut into three barrier-synchronous chunks), timed
//   SEQ   both halves in phase (what ships, with the epilogue chunked),
//   STAG  waves 4-7 seventeen periods behind waves 0-3,
// on random and on all-zero operands (the loop is power-limited on real data: overlap that raises pipe utilisation is partly paid
// back by the clock).  LOOP is the loop without any epilogue.  Results are garbage by construction (timing only).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench/enc_stagger.hip -o /tmp/enc_stagger && /tmp/enc_stagger
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

enum { MODE_LOOP = 0, MODE_SEQ = 1, MODE_STAG = 2 };
constexpr int NKS = 32, EPI = 3, PER = NKS + EPI, OFFSET = 17;

// one third of the epilogue's vector work (C = 0, 1, 2: latent blocks 0-2, 3-5, 6-7): per accumulator value scale-and-bias, compare,
// mask -- the instruction mix of the real epilogue's first passes
template <int C>
__device__ __forceinline__ void epi_chunk(f32x4 (&acc)[8][4], uint32_t& mask) {
    constexpr int sb0 = C == 0 ? 0 : (C == 1 ? 3 : 6), sb1 = C == 0 ? 3 : (C == 1 ? 6 : 8);
    const float u = 1.0009765625f, b = 0.125f, tau = 3.0e30f;
#pragma unroll
    for (int sb = sb0; sb < sb1; ++sb)
#pragma unroll
        for (int jb = 0; jb < 4; ++jb)
#pragma unroll
            for (int e = 0; e < 4; ++e)
                asm volatile("v_fma_f32 %0, %0, %2, %3\n\tv_cmp_ge_f32 vcc, %0, %4\n\tv_addc_co_u32 %1, vcc, %1, %1, vcc"
                             : "+v"(acc[sb][jb][e]), "+v"(mask) : "v"(u), "v"(b), "v"(tau) : "vcc");
}

template <int MODE>
__global__ __launch_bounds__(512, 2) void stag_kernel(const _Float16* __restrict__ wimg, const _Float16* __restrict__ ximg, int ntiles, float* out,
                                                      unsigned long long* cyc) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    KSlot* slot = reinterpret_cast<KSlot*>(smem_raw);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wid >> 2;            // waves w and w + 4 share a SIMD: one of each half per SIMD
    const int ws = wid & 1, wb = ((wid >> 1) & 1) + 2 * grp;
    const int l15 = lane & 15, kg = lane >> 4;
    const int arow0 = ws * 128 + l15, brow0 = (wb & 3) * 64 + l15;
    const int coff = 8 * (kg ^ ((4 - (l15 >> 2)) & 3));
    const size_t img = 256 * 32;
    const int bb = blockIdx.x >> 2, sp = blockIdx.x & 3;
    const _Float16* x_imgs = ximg + (size_t)bb * NKS * img;
    const uint32_t lane_off = (uint32_t)lane * 16u;
    const uint32_t lds_w = (uint32_t)(size_t)(__attribute__((address_space(3))) char*)&slot[0].a[0][0] + wid * 2048;
    auto stage = [&](int s, int p) {  // period p's operands: W image (tile p / PER, k-step p % NKS), x image (k-step p % NKS)
        const int tile = (p / PER) % ntiles, ks = p % NKS;
        const char* wsrc = reinterpret_cast<const char*>(wimg + ((size_t)(sp * ntiles + tile) * NKS + ks) * img) + wid * 2048;
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
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    float total = 0.f;
    uint32_t mask = 0;
    half8 fa[3], fb[4];
    auto kstep_compute = [&](const KSlot& cs) {
#pragma unroll
        for (int jb = 0; jb < 4; ++jb) fb[jb] = *reinterpret_cast<const half8*>(&cs.b[brow0 + 16 * jb][coff]);
        fa[0] = *reinterpret_cast<const half8*>(&cs.a[arow0][coff]);
        fa[1] = *reinterpret_cast<const half8*>(&cs.a[arow0 + 16][coff]);
#pragma unroll
        for (int sb = 0; sb < 8; ++sb) {
            const int jf = (sb & 1) ? 3 : 0;
            acc[sb][jf] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[sb % 3], fb[jf], acc[sb][jf], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (sb + 2 < 8) fa[(sb + 2) % 3] = *reinterpret_cast<const half8*>(&cs.a[arow0 + 16 * (sb + 2)][coff]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int jq = 1; jq < 4; ++jq) {
                const int jb = (sb & 1) ? 3 - jq : jq;
                acc[sb][jb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[sb % 3], fb[jb], acc[sb][jb], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    int p = 0;  // the period: which ring slot is current, which operands are staged
    auto period_end = [&]() {
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        ++p;
    };
    const int lead = (MODE == MODE_STAG) ? grp * OFFSET : 0, trail = (MODE == MODE_STAG) ? OFFSET - lead : 0;
    const unsigned long long c0 = __builtin_readcyclecounter();
    stage(0, 0); stage(1, 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    stage(2, 2);
    for (int i = 0; i < lead; ++i) { stage((p + 3) & 3, p + 3); period_end(); }  // (the late half waits out its offset: staging and barriers only)
    for (int tile = 0; tile < ntiles; ++tile) {
#pragma unroll 1
        for (int t = 0; t < NKS; ++t) {
            stage((p + 3) & 3, p + 3);
            kstep_compute(slot[p & 3]);
            period_end();
        }
        if constexpr (MODE != MODE_LOOP) {
            stage((p + 3) & 3, p + 3); epi_chunk<0>(acc, mask); period_end();
            stage((p + 3) & 3, p + 3); epi_chunk<1>(acc, mask); period_end();
            stage((p + 3) & 3, p + 3); epi_chunk<2>(acc, mask);
            total += (float)mask;
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) { total += acc[i][j][0]; acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f}; }
            period_end();
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) { total += acc[i][j][0]; acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f}; }
        }
    }
    for (int i = 0; i < trail; ++i) { stage((p + 3) & 3, p + 3); period_end(); }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) total += acc[i][j][1];
    out[(size_t)blockIdx.x * 512 + tid] = total;
    if (tid == 0) cyc[blockIdx.x] = __builtin_readcyclecounter() - c0;
}

int main() {
    const int ntiles = 32, nbb = 64, nsp = 4;
    const size_t img = 256 * 32;
    const size_t wn = (size_t)nsp * ntiles * NKS * img, xn = (size_t)nbb * NKS * img;
    std::vector<_Float16> h(wn > xn ? wn : xn);
    srand(1);
    for (auto& v : h) v = (_Float16)((rand() % 2001 - 1000) * 0.004f);
    _Float16 *w, *x;
    float* out;
    unsigned long long* cyc;
    hipMalloc(&w, (wn + 8 * img) * 2); hipMalloc(&x, (xn + 8 * img) * 2); hipMalloc(&out, 512 * 512 * 4); hipMalloc(&cyc, 256 * 8);
    const double flops = 2.0 * 16384 * 1024 * 32768;
    const int smem = 4 * (int)sizeof(KSlot);
    auto run = [&](auto kern, const char* name) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        for (int rep = 0; rep < 150; ++rep) hipLaunchKernelGGL(kern, dim3(nbb * nsp), dim3(512), smem, 0, w, x, ntiles, out, cyc);
        hipEventRecord(e0, 0);
        for (int rep = 0; rep < 150; ++rep) hipLaunchKernelGGL(kern, dim3(nbb * nsp), dim3(512), smem, 0, w, x, ntiles, out, cyc);
        hipEventRecord(e1, 0);
        hipDeviceSynchronize();
        float ms = 0.f;
        hipEventElapsedTime(&ms, e0, e1);
        ms /= 150;
        unsigned long long hc[256];
        hipMemcpy(hc, cyc, sizeof(hc), hipMemcpyDeviceToHost);
        double sc = 0;
        for (auto c : hc) sc += (double)c;
        printf("%-44s %.3f ms  %5.0f TF/s  %4.0f MHz  %.0f k cycles per launch  (%s)\n", name, ms, flops / ms * 1e-9, sc / 256 / (ms * 1e3), sc / 256 / 1e3,
               hipGetErrorString(hipGetLastError()));
    };
    for (int data = 0; data < 2; ++data) {
        if (data == 1) for (auto& v : h) v = (_Float16)0.f;
        hipMemcpy(w, h.data(), wn * 2, hipMemcpyHostToDevice);
        hipMemcpy(x, h.data(), xn * 2, hipMemcpyHostToDevice);
        printf("---- operands: %s\n", data == 0 ? "random" : "zeros");
        for (int rep = 0; rep < 2; ++rep) {
            run(stag_kernel<MODE_LOOP>, "LOOP  (32 k-steps per tile, no epilogue)");
            run(stag_kernel<MODE_SEQ>, "SEQ   (epilogue of both halves in phase)");
            run(stag_kernel<MODE_STAG>, "STAG  (waves 4-7 seventeen periods behind)");
        }
    }
    return 0;
}
