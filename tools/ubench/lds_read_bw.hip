// LDS read throughput per CU for 16-, 8- and 4-byte reads (conflict-free, 8 waves per CU), in bytes per clock.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/lds_read_bw.hip -o /tmp/lds_read_bw && /tmp/lds_read_bw
#include <hip/hip_runtime.h>

#include <cstdio>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int BYTES>
__global__ __launch_bounds__(512) void k(float* out, int iters, long long* cycles) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x;
    for (int i = tid; i < 32768; i += 512) reinterpret_cast<float*>(lds)[i] = (float)i;
    __syncthreads();
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            // every wave walks its own 8 KB window; consecutive lanes read consecutive BYTES-sized words
            const int off = ((tid >> 6) * 8192 + ((u * 1024 + (tid & 63) * BYTES + it * 64) & 8191)) & (131072 - BYTES);
            if constexpr (BYTES == 16) acc += *reinterpret_cast<const f32x4*>(lds + off);
            else if constexpr (BYTES == 8) { const f32x2 v = *reinterpret_cast<const f32x2*>(lds + off); acc[0] += v[0]; acc[1] += v[1]; }
            else acc[0] += *reinterpret_cast<const float*>(lds + off);
        }
    }
    const long long t1 = clock64();
    if (tid == 0 && blockIdx.x == 0) *cycles = t1 - t0;
    out[blockIdx.x * 512 + tid] = acc[0] + acc[1] + acc[2] + acc[3];
}

int main() {
    float* out;
    long long* cyc;
    hipMalloc(&out, 256 * 512 * 4);
    hipMalloc(&cyc, 8);
    const int iters = 2000;
    auto run = [&](auto kern, int bytes) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
        hipLaunchKernelGGL(kern, dim3(256), dim3(512), 131072, 0, out, 10, cyc);
        hipDeviceSynchronize();
        hipEvent_t e0, e1;
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(kern, dim3(256), dim3(512), 131072, 0, out, iters, cyc);
        hipEventRecord(e1, 0);
        hipDeviceSynchronize();
        float ms = 0.f;
        hipEventElapsedTime(&ms, e0, e1);
        const double total = (double)iters * 16 * 512 * bytes;  // bytes per CU
        printf("ds_read %2d B/lane: %.1f GB/s per CU = %.1f bytes/clk/CU at 2.1 GHz (%.3f ms)\n", bytes, total / ms * 1e-6,
               total / (ms * 1e-3) / 2.1e9, ms);
    };
    run(k<16>, 16);
    run(k<8>, 8);
    run(k<4>, 4);
    return 0;
}
