// Microbenchmark: global_load_lds throughput per CU as a function of waves, queue depth and working set.
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench/glds_bw.hip -o gpurun_out/glds_bw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int DEPTH>
__global__ __launch_bounds__(512) void glds_kernel(const char* src, size_t bytes_per_wg, int iters, int waves_active,
                                                   int shared_src, int* sink) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const char* base = src + (shared_src ? 0 : (size_t)blockIdx.x * bytes_per_wg);
    if (wid < waves_active) {
        // each iteration: every active wave copies 4 KB (4 calls of 1 KB) into its LDS area
        const size_t per_iter = (size_t)waves_active * 4096;
        size_t off = (size_t)wid * 4096 + lane * 16;
        for (int it = 0; it < iters; ++it) {
            const char* p = base + (off % bytes_per_wg);
#pragma unroll
            for (int j = 0; j < 4; ++j)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(p + 1024 * j),
                                                 (__attribute__((address_space(3))) void*)(lds + ((it % 4) * 8 + wid) * 4096 + 1024 * j), 16, 0, 0);
            off += per_iter;
            if (DEPTH == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (DEPTH == 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            if (DEPTH == 3) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            if (DEPTH == 4) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
            if (DEPTH == 8) asm volatile("s_waitcnt vmcnt(28)" ::: "memory");
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    if (threadIdx.x == 0 && lds[5] == 123) sink[0] = 1;
}

int main() {
    const size_t total = 1ull << 30;
    char* src; int* sink;
    hipMalloc(&src, total); hipMemset(src, 1, total); hipMalloc(&sink, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto run = [&](auto kern, int depth, int waves, int shared, size_t per_wg, int nwg) {
        const int iters = 4000;
        hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
        hipLaunchKernelGGL(kern, dim3(nwg), dim3(512), 131072, 0, src, per_wg, 200, waves, shared, sink);
        hipEventRecord(e0);
        hipLaunchKernelGGL(kern, dim3(nwg), dim3(512), 131072, 0, src, per_wg, iters, waves, shared, sink);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double bytes = (double)nwg * iters * waves * 4096;
        printf("depth %d waves %d shared %d per_wg %6zu KB nwg %d : %.2f ms  %.2f TB/s  %.1f GB/s/CU\n", depth, waves, shared,
               per_wg >> 10, nwg, ms, bytes / ms / 1e9, bytes / ms / 1e6 / nwg);
    };
    for (int shared : {1, 0})
        for (size_t per_wg : {(size_t)1 << 20, (size_t)4 << 20})
            for (int waves : {2, 4, 8}) {
                run(glds_kernel<1>, 1, waves, shared, per_wg, 256);
                run(glds_kernel<3>, 3, waves, shared, per_wg, 256);
                run(glds_kernel<8>, 8, waves, shared, per_wg, 256);
            }
    return 0;
}
