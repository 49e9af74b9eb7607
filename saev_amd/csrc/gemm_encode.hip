// Encoder contraction h = x @ W_enc + b_enc on the gfx950 matrix cores (exact f32 MFMA),
// with two epilogues:
//   EPI_DENSE : store h (n_rows x S)                      -- API-compat path (modeling.py:343-347)
//   EPI_TOPK  : never materialise h; a per-row running lower bound on the k-th largest value
//               ("min over >=k column groups of the group maximum") filters each tile's values into
//               a short per-row candidate list that select.hip reduces to the exact top-k
//               (modeling.py:169-179 fused into the producer).
//
// Orientation: the MFMA "M" dimension is the latent axis s and "N" is the batch axis b, i.e. the
// kernel computes h^T tiles.  With v_mfma_f32_32x32x2_f32 every lane then owns ONE batch row
// (column = lane & 31) and 16 latents per 32x32 block, so the per-row top-k bookkeeping is lane-local:
// the group maxima live in registers for the whole sweep.
//
// Tile: 128 latents x 128 batch rows per 256-thread workgroup, 4 waves as 2 (s) x 2 (b), 64 x 64 per
// wave (2 x 2 MFMA blocks, 64 accumulator registers), BK = 32, two LDS stages filled by
// global_load_lds (16 B per lane straight into LDS, no staging registers, no ds_write pass).
// 65 KB of LDS per workgroup -> two workgroups per CU.  A workgroup owns a 128-row batch block and
// walks a contiguous range of latent tiles (the bound it carries tightens as it goes).
//
// LDS images (one stage):
//   w[k][s]  32 x 128 f32, k-major: A fragments are conflict-free ds_read_b32 (lanes 0-31 one k-row,
//            lanes 32-63 the next), written lane-linear by global_load_lds (2 k-rows per wave call);
//   x[b][k]  128 x 32 f32, unpadded 128-byte rows, 16-byte chunks XOR-swizzled with (row>>1)&7 so the
//            B-fragment ds_read_b128 (one 16-byte chunk = 4 k values per lane, the k permutation inside
//            an 8-wide chunk is shared by A and B) is conflict-free; the swizzle is applied to the
//            per-lane SOURCE address of global_load_lds (its LDS destination is lane-linear).
#include "common.h"
#include "kernels.h"

namespace {

constexpr int TS = 128;  // latents per tile
constexpr int TB = 128;  // batch rows per tile
constexpr int BK = 32;   // k (d_model) per stage
constexpr int NTHREADS = 256;

struct __attribute__((aligned(16))) Stage {
    float w[BK][TS];  // 16 KB
    float x[TB][BK];  // 16 KB, swizzled chunks
};
struct __attribute__((aligned(16))) Smem {
    union {
        Stage st[2];
        struct {
            Stage keep;                // stage 0 stays usable while the epilogue scratch lives in stage 1
            int32_t slots32[2][32][TB];  // NG == 32: per s-wave group maxima (32 KB)
        } e32;
        int32_t slots64[2][64][TB];    // NG == 64: 64 KB, both stages
    };
    int32_t tau_key[TB];
    float bias[TS];
};

__device__ __forceinline__ void glds16(const float* gsrc, float* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

template <int EPI, int NG>  // NG: column groups per row for the bound (32 or 64), >= top_k
__global__ __launch_bounds__(NTHREADS, 2) void encode_gemm_kernel(EncodeArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    Smem& sm = *reinterpret_cast<Smem*>(smem_raw);

    if (a.enable_flag != nullptr && (*a.enable_flag != 0) != (a.enable_when != 0)) return;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ws = wid & 1;   // wave position along s
    const int wb = wid >> 1;  // wave position along b
    const int l31 = lane & 31;
    const int half = lane >> 5;

    const int D = a.D, S = a.S, B = a.n_rows;
    const int n_stiles = (S + TS - 1) / TS;
    // blockIdx.x -> (batch block, latent range).  Consecutive ids land on different XCDs (id % 8): keep the
    // latent ranges of one batch block on one XCD (they share the x block and the row bounds through its L2).
    int bb, sp;
    {
        const int id = blockIdx.x;
        const int nbb = (B + TB - 1) / TB;
        const int full = (nbb / 8) * 8 * a.s_splits;
        if (id < full) {
            const int xcd = id & 7, j = id >> 3;
            sp = j % a.s_splits;
            bb = (j / a.s_splits) * 8 + xcd;
        } else {
            const int r = id - full;
            bb = (nbb / 8) * 8 + r / a.s_splits;
            sp = r % a.s_splits;
        }
    }
    const int st_begin = (int)((long)n_stiles * sp / a.s_splits);
    const int st_end = (int)((long)n_stiles * (sp + 1) / a.s_splits);
    const int b0 = bb * TB;
    const bool rows_full = (b0 + TB <= B);

    // running group maxima, lane-private: NG == 32 folds the two s-blocks together (16 per b-block),
    // NG == 64 keeps them apart (32 per b-block)
    constexpr int NSLOT = NG / 2;
    float smax[2][NSLOT];
    if (EPI == EPI_TOPK) {
#pragma unroll
        for (int jb = 0; jb < 2; ++jb)
#pragma unroll
            for (int r = 0; r < NSLOT; ++r) smax[jb][r] = NEG_INF;
    }

    const int nk = (D + BK - 1) / BK;

    // ---- staging -------------------------------------------------------------------------------
    // interior tiles: 8 global_load_lds per wave per stage (4 for w: 2 k-rows each, 4 for x: 8 rows each)
    // Addresses are (workgroup/wave-uniform base) + (32-bit per-lane byte offset) so the loads use the
    // SGPR-base form and three offset registers cover all eight calls.
    const int xrow_l = lane >> 3;  // row within an 8-row x piece
    const uint32_t w_off = (uint32_t)(((size_t)half * S + l31 * 4) * sizeof(float));
    // x chunk swizzle: (row >> 1) & 7 with row = 32*wid + 8*j + xrow_l  ->  (4*j + (xrow_l >> 1)) & 7
    const uint32_t x_off_even = (uint32_t)(((size_t)xrow_l * D + 4 * ((lane & 7) ^ ((xrow_l >> 1) & 7))) * sizeof(float));
    const uint32_t x_off_odd = (uint32_t)(((size_t)xrow_l * D + 4 * ((lane & 7) ^ ((4 + (xrow_l >> 1)) & 7))) * sizeof(float));
    auto stage_async = [&](int buf, int s0, int k0) {
        Stage& st = sm.st[buf];
        const char* wb_ = reinterpret_cast<const char*>(a.W_enc + (size_t)(k0 + wid * 8) * S + s0);
        const char* xb_ = reinterpret_cast<const char*>(a.x + (size_t)(b0 + wid * 32) * D + k0);
#pragma unroll
        for (int j = 0; j < 4; ++j)  // k-rows wid*8 + 2j, +1
            glds16(reinterpret_cast<const float*>(wb_ + (size_t)(2 * j) * S * sizeof(float) + w_off), &st.w[wid * 8 + 2 * j][0]);
#pragma unroll
        for (int j = 0; j < 4; ++j)  // rows wid*32 + 8j .. +7
            glds16(reinterpret_cast<const float*>(xb_ + (size_t)(8 * j) * D * sizeof(float) + ((j & 1) ? x_off_odd : x_off_even)),
                   &st.x[wid * 32 + 8 * j][0]);
    };
    // edge tiles (ragged batch / d_model / d_sae): predicated register loads, zero fill, same LDS image
    auto stage_sync = [&](int buf, int s0, int k0) {
        Stage& st = sm.st[buf];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = i * 8 + (tid >> 5), col = (tid & 31) * 4;
            const int k = k0 + row, s = s0 + col;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (k < D && s < S) v = *reinterpret_cast<const f32x4*>(a.W_enc + (size_t)k * S + s);
            *reinterpret_cast<f32x4*>(&st.w[row][col]) = v;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = i * 32 + (tid >> 3), chunk = tid & 7;
            const int b = b0 + row, k = k0 + 4 * chunk;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (b < B && k < D) v = *reinterpret_cast<const f32x4*>(a.x + (size_t)b * D + k);
            *reinterpret_cast<f32x4*>(&st.x[row][4 * (chunk ^ ((row >> 1) & 7))]) = v;
        }
    };
    auto stage_tile = [&](int buf, int s0, int k0) {
        if (rows_full && s0 + TS <= S && k0 + BK <= D) stage_async(buf, s0, k0);  // workgroup-uniform
        else stage_sync(buf, s0, k0);
    };

    // ---- fragments ---------------------------------------------------------------------------------
    const int xr0 = wb * 64 + l31, xr1 = xr0 + 32;
    const int xs0 = (xr0 >> 1) & 7, xs1 = (xr1 >> 1) & 7;
    auto load_frags = [&](int buf, int kc, float (&af)[2][4], f32x4 (&bf)[2]) {
        const Stage& st = sm.st[buf];
        const int c = (kc >> 2) + half;
        bf[0] = *reinterpret_cast<const f32x4*>(&st.x[xr0][4 * (c ^ xs0)]);
        bf[1] = *reinterpret_cast<const f32x4*>(&st.x[xr1][4 * (c ^ xs1)]);
#pragma unroll
        for (int sb = 0; sb < 2; ++sb)
#pragma unroll
            for (int t = 0; t < 4; ++t) af[sb][t] = st.w[kc + 4 * half + t][ws * 64 + sb * 32 + l31];
    };

    if (st_begin < st_end) stage_tile(0, st_begin * TS, 0);

    for (int st = st_begin; st < st_end; ++st) {
        const int s0 = st * TS;
        f32x16 acc[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

        auto mma_part = [&](const float (&af)[2][4], const f32x4 (&bf)[2], int t0, int t1) {
#pragma unroll
            for (int t = t0; t < t1; ++t)
#pragma unroll
                for (int sb = 0; sb < 2; ++sb)
#pragma unroll
                    for (int jb = 0; jb < 2; ++jb)
                        acc[sb][jb] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[sb][t], bf[jb][t], acc[sb][jb], 0, 0, 0);
        };

        if (tid < TS) sm.bias[tid] = (s0 + tid < S) ? a.b_enc[s0 + tid] : 0.f;
        __syncthreads();  // stage 0 of this tile (issued by the previous epilogue / the prologue) has landed

        for (int kt = 0; kt < nk; ++kt) {
            const int buf = kt & 1;
            if (kt + 1 < nk) stage_tile(buf ^ 1, s0, (kt + 1) * BK);
            // 4 chunks of 8 k-values; fragments of chunk c+1 are fetched before the MFMAs of chunk c issue
            float afA[2][4], afB[2][4];
            f32x4 bfA[2], bfB[2];
            // The reads for chunk c+1 are issued right after the first four MFMAs of chunk c, so the wait in
            // front of chunk c+1 is ~750 cycles behind them (no exposed LDS latency even with lgkmcnt(0)).
            load_frags(buf, 0, afA, bfA);
            __builtin_amdgcn_sched_barrier(0);
            mma_part(afA, bfA, 0, 1);
            __builtin_amdgcn_sched_barrier(0);
            load_frags(buf, 8, afB, bfB);
            __builtin_amdgcn_sched_barrier(0);
            mma_part(afA, bfA, 1, 4);
            mma_part(afB, bfB, 0, 1);
            __builtin_amdgcn_sched_barrier(0);
            load_frags(buf, 16, afA, bfA);
            __builtin_amdgcn_sched_barrier(0);
            mma_part(afB, bfB, 1, 4);
            mma_part(afA, bfA, 0, 1);
            __builtin_amdgcn_sched_barrier(0);
            load_frags(buf, 24, afB, bfB);
            __builtin_amdgcn_sched_barrier(0);
            mma_part(afA, bfA, 1, 4);
            mma_part(afB, bfB, 0, 4);
            __syncthreads();
        }
        // both stages are free now.  Start fetching the next tile's first stage so it lands during the
        // epilogue (stage 0 is not touched by the NG == 32 scratch).
        const bool prefetched = (NG == 32 || EPI == EPI_DENSE) && (st + 1 < st_end);
        if (prefetched) stage_tile(0, s0 + TS, 0);

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
            __syncthreads();  // bias is rewritten by the next tile
        } else {
            // add bias, invalidate out-of-range latents, fold into the lane-private group maxima
#pragma unroll
            for (int sb = 0; sb < 2; ++sb)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int sl = ws * 64 + sb * 32 + 8 * q + 4 * half;
                    const bool ok = (s0 + sl) < S;  // d_sae % 4 == 0: a float4 of latents is all in or all out
                    const f32x4 bq = *reinterpret_cast<const f32x4*>(&sm.bias[sl]);  // one unconditional 16-byte LDS read
#pragma unroll
                    for (int jb = 0; jb < 2; ++jb)
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float hv = acc[sb][jb][4 * q + e] + bq[e];
                            const float v = ok ? hv : NEG_INF;
                            acc[sb][jb][4 * q + e] = v;
                            const int slot = (NG == 32) ? (4 * q + e) : (16 * sb + 4 * q + e);
                            smax[jb][slot] = fmaxf(smax[jb][slot], v);
                        }
                }
            // publish this wave's maxima; group id within the row = (slot, half), the same ids in both s-waves
#pragma unroll
            for (int jb = 0; jb < 2; ++jb) {
                const int bl = wb * 64 + jb * 32 + l31;
#pragma unroll
                for (int r = 0; r < NSLOT; ++r) {
                    if (NG == 32) sm.e32.slots32[ws][2 * r + half][bl] = f2key(smax[jb][r]);
                    else sm.slots64[ws][2 * r + half][bl] = f2key(smax[jb][r]);
                }
            }
            __syncthreads();
            // bound per row = min over groups of the group maximum, where each group maximum is merged (a) across the
            // two s-waves of this workgroup and (b) with what the row's other latent ranges have published so far in
            // global memory (relaxed L2 reads: a stale value only gives a weaker, still valid bound).  Two threads per
            // row, each owning half of the groups; improved maxima are published fire-and-forget.
            if (tid < TB) sm.tau_key[tid] = INT32_MAX;
            __syncthreads();
            {
                const int row = tid % TB;
                const int part = __builtin_amdgcn_readfirstlane(tid / TB);  // wave-uniform: group bases stay in SGPRs  // 256 threads = 2 x TB rows
                constexpr int GPT = NG / 2;                     // groups per thread
                const int b = b0 + row;
                const bool share = (a.s_splits > 1) && (b < B);
                const uint32_t boff = (uint32_t)b * 4u;
                int32_t m = INT32_MAX;
#pragma unroll 1
                for (int c0 = 0; c0 < GPT; c0 += 4) {  // four global reads in flight at a time (register pressure)
                    int32_t v[4], old[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int g = part * GPT + c0 + i;
                        const int32_t v0 = (NG == 32) ? sm.e32.slots32[0][g][row] : sm.slots64[0][g][row];
                        const int32_t v1 = (NG == 32) ? sm.e32.slots32[1][g][row] : sm.slots64[1][g][row];
                        v[i] = max(v0, v1);
                        old[i] = INT32_MIN;
                        if (share)
                            old[i] = __hip_atomic_load(
                                reinterpret_cast<int32_t*>(reinterpret_cast<char*>(a.gmax + (size_t)g * a.gmax_stride) + boff),
                                __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        if (share && v[i] > old[i])
                            atomicMax(reinterpret_cast<int32_t*>(
                                          reinterpret_cast<char*>(a.gmax + (size_t)(part * GPT + c0 + i) * a.gmax_stride) + boff),
                                      v[i]);
                        m = min(m, max(v[i], old[i]));
                    }
                }
                atomicMin(&sm.tau_key[row], m);
            }
            __syncthreads();
            // keep values >= bound: count, reserve a slice of each row's list (both atomics in flight together),
            // then store
            int npass[2], pos[2];
            float tau2[2];
#pragma unroll
            for (int jb = 0; jb < 2; ++jb) {
                // a finite bound: padded latents carry -inf and must never pass (a row whose bound is still -inf has seen
                // fewer than k groups with a real value; everything real passes then)
                const float tau = fmaxf(key2f(sm.tau_key[wb * 64 + jb * 32 + l31]), -3.0e38f);
                tau2[jb] = tau;
                int n = 0;
#pragma unroll
                for (int sb = 0; sb < 2; ++sb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) n += (acc[sb][jb][r] >= tau) ? 1 : 0;
                npass[jb] = (b0 + wb * 64 + jb * 32 + l31 < B) ? n : 0;
            }
#pragma unroll
            for (int jb = 0; jb < 2; ++jb) {
                pos[jb] = 0;
                if (npass[jb] > 0) pos[jb] = atomicAdd(&a.cand_cnt[b0 + wb * 64 + jb * 32 + l31], npass[jb]);
            }
            // wait for the two counters once, here: otherwise every conditionally executed store block below gets its own
            // s_waitcnt vmcnt(0) (the block before it may have been skipped), which also serialises the stores
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            asm volatile("" : "+v"(pos[0]), "+v"(pos[1]));
#pragma unroll
            for (int jb = 0; jb < 2; ++jb) {
                // a row whose list would overflow is not written at all: its counter already says so, and the step then
                // re-runs on the exact dense route (overflow_check).  Offsets are 32-bit from the uniform buffer bases
                // (n_rows * cand_stride * 4 < 2^32), so a kept value costs one address add and two stores.
                if (npass[jb] > 0 && pos[jb] + npass[jb] <= a.cand_cap) {
                    const int bl = wb * 64 + jb * 32 + l31;
                    const float tau = tau2[jb];
                    uint32_t off = ((uint32_t)(b0 + bl) * (uint32_t)a.cand_stride + (uint32_t)pos[jb]) * 4u;
#pragma unroll
                    for (int sb = 0; sb < 2; ++sb)
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const float v = acc[sb][jb][r];
                            if (v >= tau) {
                                *reinterpret_cast<float*>(reinterpret_cast<char*>(a.cand_val) + off) = v;
                                *reinterpret_cast<int32_t*>(reinterpret_cast<char*>(a.cand_idx) + off) =
                                    s0 + ws * 64 + sb * 32 + 8 * (r >> 2) + 4 * half + (r & 3);
                                off += 4u;
                            }
                        }
                }
            }
        }
        if (!prefetched && st + 1 < st_end) {
            __syncthreads();  // NG == 64: the scratch covered both stages
            stage_tile(0, s0 + TS, 0);
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
        const void* fns[3] = {reinterpret_cast<const void*>(&encode_gemm_kernel<EPI_DENSE, 32>),
                              reinterpret_cast<const void*>(&encode_gemm_kernel<EPI_TOPK, 32>),
                              reinterpret_cast<const void*>(&encode_gemm_kernel<EPI_TOPK, 64>)};
        for (const void* f : fns) {
            hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
            if (e != hipSuccess) return e;
        }
        attr_set = true;
    }
    if (epi == EPI_DENSE)
        hipLaunchKernelGGL((encode_gemm_kernel<EPI_DENSE, 32>), grid, block, smem, stream, a);
    else if (a.ngroups <= 32)
        hipLaunchKernelGGL((encode_gemm_kernel<EPI_TOPK, 32>), grid, block, smem, stream, a);
    else
        hipLaunchKernelGGL((encode_gemm_kernel<EPI_TOPK, 64>), grid, block, smem, stream, a);
    return hipGetLastError();
}

int encode_gemm_tile_rows() { return TB; }
int encode_gemm_tile_latents() { return TS; }
