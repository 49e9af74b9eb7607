// Does the relative placement of the four Adam streams (p, g, m, v: read 4, write 3, 268.6 MB each at configs[1]) decide
// the 0.32 / 0.39 ms the kernel shows from one engine instance to the next?  The buffers are carved out of one slab at
// base + i * (bytes + pad) for a list of pads; the kernel is the access pattern of adam_kernel (float4, grid-stride).
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/adam_streams.hip -o build/ubench/adam_streams && build/ubench/adam_streams
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));

// NT: 0 plain, 1 nontemporal loads and stores, 2 nontemporal stores only; UNR: float4 groups per thread and trip
template <int NT, int UNR>
__global__ __launch_bounds__(256) void k(float* p, const float* g, float* m, float* v, long n4) {
    for (long q0 = ((long)blockIdx.x * 256) * UNR + threadIdx.x; q0 < n4; q0 += (long)gridDim.x * 256 * UNR)
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
        const long q = q0 + u * 256;
        if (q >= n4) break;
        f32x4 pp, gg, mm, vv;
        if (NT == 1) {
            pp = __builtin_nontemporal_load(reinterpret_cast<f32x4*>(p) + q);
            gg = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(g) + q);
            mm = __builtin_nontemporal_load(reinterpret_cast<f32x4*>(m) + q);
            vv = __builtin_nontemporal_load(reinterpret_cast<f32x4*>(v) + q);
        } else {
            pp = reinterpret_cast<f32x4*>(p)[q];
            gg = reinterpret_cast<const f32x4*>(g)[q];
            mm = reinterpret_cast<f32x4*>(m)[q];
            vv = reinterpret_cast<f32x4*>(v)[q];
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            mm[e] = mm[e] + (gg[e] - mm[e]) * 0.1f;
            vv[e] = 0.999f * vv[e] + 0.001f * gg[e] * gg[e];
            pp[e] -= 1e-4f * (mm[e] / (sqrtf(vv[e]) + 1e-8f));
        }
        if (NT != 0) {
            __builtin_nontemporal_store(pp, reinterpret_cast<f32x4*>(p) + q);
            __builtin_nontemporal_store(mm, reinterpret_cast<f32x4*>(m) + q);
            __builtin_nontemporal_store(vv, reinterpret_cast<f32x4*>(v) + q);
        } else {
            reinterpret_cast<f32x4*>(p)[q] = pp;
            reinterpret_cast<f32x4*>(m)[q] = mm;
            reinterpret_cast<f32x4*>(v)[q] = vv;
        }
    }
}

int main() {
    const long n = 67142656;  // floats per buffer at configs[1]
    const size_t bytes = n * 4;
    const size_t pads[] = {0, 256, 4096, 65536, 1 << 20, (1 << 20) + 65536, 2 << 20, (2 << 20) + 4096, 3 << 20, 5 << 20, 17 << 20,
                           (33 << 20) + 65536};
    char* slab;
    const size_t slab_bytes = 4 * (bytes + (64 << 20)) + (64 << 20);
    if (hipMalloc(&slab, slab_bytes) != hipSuccess) return 1;
    hipMemset(slab, 0, slab_bytes);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (size_t base_off : {(size_t)0, (size_t)(1 << 20) + 4096}) {
        for (size_t pad : pads) {
            float* b[4];
            for (int i = 0; i < 4; ++i) b[i] = reinterpret_cast<float*>(slab + base_off + i * (bytes + pad));
            float best = 1e9f, worst = 0.f;
            for (int rep = 0; rep < 6; ++rep) {
                hipEventRecord(e0, 0);
                hipLaunchKernelGGL((k<0, 1>), dim3(2048), dim3(256), 0, 0, b[0], b[1], b[2], b[3], n / 4);
                hipEventRecord(e1, 0);
                hipDeviceSynchronize();
                float ms;
                hipEventElapsedTime(&ms, e0, e1);
                if (rep > 0) { best = ms < best ? ms : best; worst = ms > worst ? ms : worst; }
            }
            printf("base +%8zu  pad %9zu B (stride mod 2 MiB = %7zu): %.3f - %.3f ms  (%.2f TB/s)\n", base_off, pad,
                   (bytes + pad) % (2 << 20), best, worst, 7.0 * bytes / (best * 1e-3) / 1e12);
        }
    }
    // kernel variants at pad 0: nontemporal accesses, unrolling, grid size
    float* b[4];
    for (int i = 0; i < 4; ++i) b[i] = reinterpret_cast<float*>(slab + i * bytes);
    auto run = [&](auto kern, int grid, const char* name) {
        float best = 1e9f;
        for (int rep = 0; rep < 6; ++rep) {
            hipEventRecord(e0, 0);
            hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, b[0], b[1], b[2], b[3], n / 4);
            hipEventRecord(e1, 0);
            hipDeviceSynchronize();
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            if (rep > 0) best = ms < best ? ms : best;
        }
        printf("%-44s grid %5d: %.3f ms  (%.2f TB/s)\n", name, grid, best, 7.0 * bytes / (best * 1e-3) / 1e12);
    };
    for (int grid : {1024, 2048, 4096, 8192, 16384}) {
        run(k<0, 1>, grid, "plain");
        run(k<1, 1>, grid, "nontemporal loads + stores");
        run(k<2, 1>, grid, "nontemporal stores");
        run(k<0, 4>, grid, "plain, 4 float4 per thread and trip");
        run(k<1, 4>, grid, "nontemporal, 4 float4 per thread and trip");
    }
    return 0;
}
