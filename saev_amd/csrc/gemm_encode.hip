// Encoder contraction h = x @ W_enc + b_enc on the gfx950 matrix cores (exact f32 MFMA),
// with two epilogues:
//   EPI_DENSE : store h (n_rows x S)                      -- API-compat path (modeling.py:343-347)
//   EPI_TOPK  : never materialise h; per-row running lower bound on the k-th largest value
//               ("min over >=k column groups of the group maximum") filters each tile's values into
//               a short per-row candidate list that select.hip reduces to the exact top-k
//               (modeling.py:169-179 fused into the producer).
//
// Orientation: the MFMA "M" dimension is the latent axis s and "N" is the batch axis b, i.e. the
// kernel computes h^T tiles.  With v_mfma_f32_32x32x2_f32 every lane then owns ONE batch row
// (column = lane & 31) and 16 latents per 32x32 block, so the per-row top-k bookkeeping is lane-local.
//
// Tile: 256 latents x 128 batch rows per workgroup, 8 waves as 4 (s) x 2 (b), 64 x 64 per wave
// (2 x 2 MFMA blocks, 64 accumulator registers), BK = 32, LDS double-buffered, register-staged.
// A workgroup owns a 128-row batch block and walks a contiguous range of latent tiles so the
// top-k bound it carries in LDS tightens as it goes.
#include "common.h"
#include "kernels.h"

namespace {

constexpr int TS = 256;        // latents per tile
constexpr int TB = 128;        // batch rows per tile
constexpr int BK = 32;         // k (d_model) per stage
constexpr int XS_STRIDE = 36;  // floats; 144 B rows keep ds_read_b128 conflict-free
constexpr int NTHREADS = 512;
constexpr int MAXG = 64;       // max column groups for the bound (>= top_k)

struct __attribute__((aligned(16))) Smem {
    float w[2][BK][TS];              // 2 x 32 KB   W_enc tile, [k][s]
    float x[2][TB][XS_STRIDE];       // 2 x 18 KB   x tile, [b][k] padded
    int32_t slots[MAXG][TB];         // 32 KB       running group maxima (ordered-int keys)
    float tau[TB];                   // current lower bound per batch row
    float bias[TS];
};

template <int EPI>
__global__ __launch_bounds__(NTHREADS, 2) void encode_gemm_kernel(EncodeArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    Smem& sm = *reinterpret_cast<Smem*>(smem_raw);

    if (a.enable_flag != nullptr && (*a.enable_flag != 0) != (a.enable_when != 0)) return;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ws = wid & 3;   // wave position along s
    const int wb = wid >> 2;  // wave position along b
    const int l31 = lane & 31;
    const int half = lane >> 5;

    const int D = a.D, S = a.S, B = a.n_rows;
    const int n_stiles = (S + TS - 1) / TS;
    // blockIdx.x -> (batch block, latent range)
    const int bb = blockIdx.x / a.s_splits;
    const int sp = blockIdx.x % a.s_splits;
    const int st_begin = (int)((long)n_stiles * sp / a.s_splits);
    const int st_end = (int)((long)n_stiles * (sp + 1) / a.s_splits);
    const int b0 = bb * TB;

    const int ngroups = a.ngroups;  // 32 or 64 (EPI_TOPK)
    if (EPI == EPI_TOPK) {
        for (int i = tid; i < MAXG * TB; i += NTHREADS) (&sm.slots[0][0])[i] = INT32_MIN;
    }

    const int nk = (D + BK - 1) / BK;

    // staging registers
    f32x4 wreg[4], xreg[2];

    auto load_tiles = [&](int s0, int k0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = i * 8 + (tid >> 6);
            const int col = (tid & 63) * 4;
            const int k = k0 + row, s = s0 + col;
            if (k < D && s < S)
                wreg[i] = *reinterpret_cast<const f32x4*>(a.W_enc + (size_t)k * S + s);
            else
                wreg[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int row = i * 64 + (tid >> 3);
            const int c = (tid & 7) * 4;
            const int b = b0 + row, k = k0 + c;
            if (b < B && k < D)
                xreg[i] = *reinterpret_cast<const f32x4*>(a.x + (size_t)b * D + k);
            else
                xreg[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
    };
    auto store_tiles = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = i * 8 + (tid >> 6);
            const int col = (tid & 63) * 4;
            *reinterpret_cast<f32x4*>(&sm.w[buf][row][col]) = wreg[i];
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int row = i * 64 + (tid >> 3);
            const int c = (tid & 7) * 4;
            *reinterpret_cast<f32x4*>(&sm.x[buf][row][c]) = xreg[i];
        }
    };

    for (int st = st_begin; st < st_end; ++st) {
        const int s0 = st * TS;
        f32x16 acc[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

        load_tiles(s0, 0);
        if (tid < TS) sm.bias[tid] = (s0 + tid < S) ? a.b_enc[s0 + tid] : 0.f;
        store_tiles(0);
        __syncthreads();

        for (int kt = 0; kt < nk; ++kt) {
            const int buf = kt & 1;
            if (kt + 1 < nk) load_tiles(s0, (kt + 1) * BK);
#pragma unroll
            for (int kc = 0; kc < BK; kc += 8) {
                float af[2][4];
                f32x4 bf[2];
#pragma unroll
                for (int sb = 0; sb < 2; ++sb)
#pragma unroll
                    for (int t = 0; t < 4; ++t)
                        af[sb][t] = sm.w[buf][kc + 4 * half + t][ws * 64 + sb * 32 + l31];
#pragma unroll
                for (int jb = 0; jb < 2; ++jb)
                    bf[jb] = *reinterpret_cast<const f32x4*>(&sm.x[buf][wb * 64 + jb * 32 + l31][kc + 4 * half]);
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int sb = 0; sb < 2; ++sb)
#pragma unroll
                        for (int jb = 0; jb < 2; ++jb)
                            acc[sb][jb] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[sb][t], bf[jb][t], acc[sb][jb], 0, 0, 0);
            }
            if (kt + 1 < nk) store_tiles(buf ^ 1);
            __syncthreads();
        }

        // ---------------- epilogue ----------------
        // lane owns batch rows bl(jb) = wb*64 + jb*32 + l31; latent of acc[sb][jb][r]:
        //   sl = ws*64 + sb*32 + 8*(r>>2) + 4*half + (r&3)
        if (EPI == EPI_DENSE) {
#pragma unroll
            for (int jb = 0; jb < 2; ++jb) {
                const int b = b0 + wb * 64 + jb * 32 + l31;
                if (b >= B) continue;
#pragma unroll
                for (int sb = 0; sb < 2; ++sb)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int sl = ws * 64 + sb * 32 + 8 * q + 4 * half;
                        const int s = s0 + sl;
                        if (s < S) {
                            f32x4 v;
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] = acc[sb][jb][4 * q + e] + sm.bias[sl + e];
                            *reinterpret_cast<f32x4*>(a.h_out + (size_t)b * S + s) = v;
                        }
                    }
            }
            __syncthreads();  // bias/tiles reused by next tile
        } else {
            // add bias, invalidate out-of-range entries
#pragma unroll
            for (int sb = 0; sb < 2; ++sb)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int sl = ws * 64 + sb * 32 + 8 * q + 4 * half;
                    const bool ok = (s0 + sl) < S;
#pragma unroll
                    for (int jb = 0; jb < 2; ++jb)
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float v = acc[sb][jb][4 * q + e] + sm.bias[sl + e];
                            acc[sb][jb][4 * q + e] = ok ? v : NEG_INF;
                        }
                }
            // 1) fold this tile into the running group maxima.  group id = r + 16*half (+32*sb)
#pragma unroll
            for (int jb = 0; jb < 2; ++jb) {
                const int bl = wb * 64 + jb * 32 + l31;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    if (ngroups == 32) {
                        const float m = fmaxf(acc[0][jb][r], acc[1][jb][r]);
                        atomicMax(&sm.slots[r + 16 * half][bl], f2key(m));
                    } else {
                        atomicMax(&sm.slots[r + 16 * half][bl], f2key(acc[0][jb][r]));
                        atomicMax(&sm.slots[32 + r + 16 * half][bl], f2key(acc[1][jb][r]));
                    }
                }
            }
            __syncthreads();
            // 2) bound per row = min over groups; share with the other latent ranges of this row
            if (tid < TB) {
                int32_t m = INT32_MAX;
                for (int g = 0; g < ngroups; ++g) m = min(m, sm.slots[g][tid]);
                const int b = b0 + tid;
                if (b < B && a.s_splits > 1) {
                    const int32_t old = atomicMax(&a.row_tau[b], m);
                    m = max(m, old);
                }
                sm.tau[tid] = (m == INT32_MIN) ? NEG_INF : key2f(m);
            }
            __syncthreads();
            // 3) keep values >= bound
#pragma unroll
            for (int jb = 0; jb < 2; ++jb) {
                const int bl = wb * 64 + jb * 32 + l31;
                const int b = b0 + bl;
                const float tau = sm.tau[bl];
                if (b < B) {
#pragma unroll
                    for (int sb = 0; sb < 2; ++sb)
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const float v = acc[sb][jb][r];
                            if (v >= tau && v > NEG_INF) {
                                const int s = s0 + ws * 64 + sb * 32 + 8 * (r >> 2) + 4 * half + (r & 3);
                                const int pos = atomicAdd(&a.cand_cnt[b], 1);
                                if (pos < a.cand_cap) {
                                    a.cand_val[(size_t)b * a.cand_cap + pos] = v;
                                    a.cand_idx[(size_t)b * a.cand_cap + pos] = s;
                                }
                            }
                        }
                }
            }
            // no barrier needed: next tile's first writes to LDS tiles are ordered by the
            // barrier after store_tiles(0); slots/tau are only touched after further barriers.
        }
    }
}

}  // namespace

size_t encode_gemm_smem_bytes() { return sizeof(Smem); }

hipError_t launch_encode_gemm(const EncodeArgs& a, int epi, hipStream_t stream) {
    const int n_bblocks = (a.n_rows + TB - 1) / TB;
    dim3 grid(n_bblocks * a.s_splits), block(NTHREADS);
    const size_t smem = sizeof(Smem);
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&encode_gemm_kernel<EPI_DENSE>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (e != hipSuccess) return e;
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(&encode_gemm_kernel<EPI_TOPK>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    if (epi == EPI_DENSE)
        hipLaunchKernelGGL(encode_gemm_kernel<EPI_DENSE>, grid, block, smem, stream, a);
    else
        hipLaunchKernelGGL(encode_gemm_kernel<EPI_TOPK>, grid, block, smem, stream, a);
    return hipGetLastError();
}
