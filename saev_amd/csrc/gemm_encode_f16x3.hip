// Encoder contraction h = x @ W_enc + b_enc at fp32 accuracy on the gfx950 *f16* matrix cores.
//
// Every fp32 operand is split into two halves, a = a_hi + a_lo with a_hi = fp16(a), a_lo = fp16(a - a_hi)
// (22 significand bits together), and each product is formed as
//        x*w  ~=  x_hi*w_hi + x_hi*w_lo + x_lo*w_hi          (the lo*lo term is below fp32 rounding)
// by three v_mfma_f32_32x32x16_f16 accumulating in fp32.  W_enc is pre-scaled by 2^8 before the split so its
// low halves stay out of the fp16 subnormal range; the accumulator is scaled back by 2^-8 (exact) in the
// epilogue.  Measured error against fp64 equals that of a native fp32 GEMM (rms 4.8e-7 relative; DESIGN.md
// section 3.1) -- three orders of magnitude tighter than the TF32 the reference enables on CUDA
// (framework/train.py:253-257).  The f16 MFMA rate is 16x the f32 MFMA rate, so three products cost 3/16.
//
// Inputs are the pre-split operands produced by split.hip:
//   xh, xl   (rows padded to 256, Dp = d_model padded to 32) fp16, row-major           [b][k]
//   wh, wl   (d_sae padded to 256, Dp) fp16, row-major -- i.e. W_enc TRANSPOSED         [s][k]
// so both MFMA operands are k-contiguous: one ds_read_b128 = one 8-wide k fragment.
//
// Tile: 256 latents x 256 batch rows per 512-thread workgroup, 8 waves as 2 (s) x 4 (b), 128 x 64 per wave
// (4 x 2 MFMA blocks, 128 accumulator registers), BK = 32 halfs, two 64 KB LDS stages filled by
// global_load_lds.  Rows of the LDS images are 64 bytes (4 chunks of 16 B); chunk c of row r is stored at
// slot c ^ ((r >> 2) & 3), applied on the global source address, which makes the fragment reads
// conflict-free.  Orientation and the TopK epilogue are those of gemm_encode.hip (lanes own batch rows).
#include "common.h"
#include "kernels.h"

namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));

constexpr int HTS = 256;  // latents per tile
constexpr int HTB = 256;  // batch rows per tile
constexpr int HBK = 32;   // halfs of k per stage
constexpr int HTHREADS = 512;

struct __attribute__((aligned(16))) HStage {
    _Float16 ah[HTS][HBK];  // W^T hi   16 KB
    _Float16 al[HTS][HBK];  // W^T lo
    _Float16 bh[HTB][HBK];  // x hi
    _Float16 bl[HTB][HBK];  // x lo
};
struct __attribute__((aligned(16))) HSmem {
    union {
        HStage st[2];  // 128 KB
        struct {
            HStage keep;                   // stage 0 stays usable during the NG == 32 epilogue
            int32_t slots32[2][32][HTB];   // 64 KB
        } e32;
        int32_t slots64[2][64][HTB];       // 128 KB
    };
    float tau[HTB];
    float bias[HTS];
};

__device__ __forceinline__ void glds16h(const char* gsrc, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

template <int EPI, int NG>
__global__ __launch_bounds__(HTHREADS, 2) void encode_f16x3_kernel(EncodeF16Args a) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    HSmem& sm = *reinterpret_cast<HSmem*>(smem_raw);

    if (a.enable_flag != nullptr && (*a.enable_flag != 0) != (a.enable_when != 0)) return;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ws = wid & 1;   // wave position along s (128 latents each)
    const int wb = wid >> 1;  // wave position along b (64 rows each)
    const int l31 = lane & 31;
    const int half = lane >> 5;

    const int Dp = a.Dp, S = a.S, B = a.n_rows;
    const int n_stiles = (S + HTS - 1) / HTS;
    int bb, sp;
    {
        const int id = blockIdx.x;
        const int nbb = (B + HTB - 1) / HTB;
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
    const int b0 = bb * HTB;

    constexpr int NSLOT = NG / 2;
    float smax[2][NSLOT];
    if (EPI == EPI_TOPK) {
#pragma unroll
        for (int jb = 0; jb < 2; ++jb)
#pragma unroll
            for (int r = 0; r < NSLOT; ++r) smax[jb][r] = NEG_INF;
    }

    const int nk = Dp / HBK;

    // one global_load_lds call = 16 rows x 64 B; lane -> row (lane >> 2), physical chunk (lane & 3) holding
    // logical chunk (lane & 3) ^ ((row >> 2) & 3) = (lane & 3) ^ ((lane >> 4) & 3)
    const uint32_t g_off = (uint32_t)(((size_t)(lane >> 2) * Dp + 8 * ((lane & 3) ^ ((lane >> 4) & 3))) * sizeof(_Float16));
    auto stage_async = [&](int buf, int s0, int k0) {
        HStage& st = sm.st[buf];
        const size_t wrow = (size_t)(s0 + wid * 32) * Dp + k0;
        const size_t xrow = (size_t)(b0 + wid * 32) * Dp + k0;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const size_t d = (size_t)(16 * j) * Dp;
            glds16h(reinterpret_cast<const char*>(a.wh + wrow + d) + g_off, &st.ah[wid * 32 + 16 * j][0]);
            glds16h(reinterpret_cast<const char*>(a.wl + wrow + d) + g_off, &st.al[wid * 32 + 16 * j][0]);
            glds16h(reinterpret_cast<const char*>(a.xh + xrow + d) + g_off, &st.bh[wid * 32 + 16 * j][0]);
            glds16h(reinterpret_cast<const char*>(a.xl + xrow + d) + g_off, &st.bl[wid * 32 + 16 * j][0]);
        }
    };

    // fragment rows of this lane and their chunk swizzles
    const int arow0 = ws * 128 + l31;             // + 32*sb
    const int brow0 = wb * 64 + l31;              // + 32*jb
    const int asw = (l31 >> 2) & 3;               // ((arow0 + 32*sb) >> 2) & 3 is independent of sb, ws
    const int bsw = (l31 >> 2) & 3;

    if (st_begin < st_end) stage_async(0, st_begin * HTS, 0);

    for (int st = st_begin; st < st_end; ++st) {
        const int s0 = st * HTS;
        f32x16 acc[4][2];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

        if (tid < HTS) sm.bias[tid] = (s0 + tid < S) ? a.b_enc[s0 + tid] : 0.f;
        int32_t tau_other = INT32_MIN;
        if (EPI == EPI_TOPK && tid < HTB && a.s_splits > 1 && b0 + tid < B) tau_other = a.row_tau[b0 + tid];
        __syncthreads();  // stage 0 of this tile has landed

        for (int kt = 0; kt < nk; ++kt) {
            const int buf = kt & 1;
            if (kt + 1 < nk) stage_async(buf ^ 1, s0, (kt + 1) * HBK);
            const HStage& cs = sm.st[buf];
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const int c = 2 * ks + half;
                half8 bh[2], bl[2];
#pragma unroll
                for (int jb = 0; jb < 2; ++jb) {
                    bh[jb] = *reinterpret_cast<const half8*>(&cs.bh[brow0 + 32 * jb][8 * (c ^ bsw)]);
                    bl[jb] = *reinterpret_cast<const half8*>(&cs.bl[brow0 + 32 * jb][8 * (c ^ bsw)]);
                }
#pragma unroll
                for (int sb = 0; sb < 4; ++sb) {
                    const half8 ah = *reinterpret_cast<const half8*>(&cs.ah[arow0 + 32 * sb][8 * (c ^ asw)]);
                    const half8 al = *reinterpret_cast<const half8*>(&cs.al[arow0 + 32 * sb][8 * (c ^ asw)]);
#pragma unroll
                    for (int jb = 0; jb < 2; ++jb) {
                        acc[sb][jb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh[jb], acc[sb][jb], 0, 0, 0);
                        acc[sb][jb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl[jb], acc[sb][jb], 0, 0, 0);
                        acc[sb][jb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh[jb], acc[sb][jb], 0, 0, 0);
                    }
                }
            }
            __syncthreads();
        }
        const bool prefetched = (NG == 32 || EPI == EPI_DENSE) && (st + 1 < st_end);
        if (prefetched) stage_async(0, s0 + HTS, 0);

        // ---------------- epilogue ----------------
        // lane owns batch rows bl(jb) = wb*64 + jb*32 + l31; latent of acc[sb][jb][r]:
        //   sl = ws*128 + sb*32 + 8*(r>>2) + 4*half + (r&3);   acc holds 2^8 * (x . w)
        const float unscale = 1.0f / a.w_scale;
        if (EPI == EPI_DENSE) {
#pragma unroll
            for (int jb = 0; jb < 2; ++jb) {
                const int b = b0 + wb * 64 + jb * 32 + l31;
                if (b >= B) continue;
#pragma unroll
                for (int sb = 0; sb < 4; ++sb)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int sl = ws * 128 + sb * 32 + 8 * q + 4 * half;
                        const int s = s0 + sl;
                        if (s < S) {
                            f32x4 v;
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] = acc[sb][jb][4 * q + e] * unscale + sm.bias[sl + e];
                            *reinterpret_cast<f32x4*>(a.h_out + (size_t)b * S + s) = v;
                        }
                    }
            }
            __syncthreads();
        } else {
#pragma unroll
            for (int sb = 0; sb < 4; ++sb)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int sl = ws * 128 + sb * 32 + 8 * q + 4 * half;
                    const bool ok = (s0 + sl) < S;
#pragma unroll
                    for (int jb = 0; jb < 2; ++jb)
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float v = ok ? acc[sb][jb][4 * q + e] * unscale + sm.bias[sl + e] : NEG_INF;
                            acc[sb][jb][4 * q + e] = v;
                            const int slot = (NG == 32) ? (4 * q + e) : (16 * (sb & 1) + 4 * q + e);
                            smax[jb][slot] = fmaxf(smax[jb][slot], v);
                        }
                }
#pragma unroll
            for (int jb = 0; jb < 2; ++jb) {
                const int bl_ = wb * 64 + jb * 32 + l31;
#pragma unroll
                for (int r = 0; r < NSLOT; ++r) {
                    if (NG == 32) sm.e32.slots32[ws][2 * r + half][bl_] = f2key(smax[jb][r]);
                    else sm.slots64[ws][2 * r + half][bl_] = f2key(smax[jb][r]);
                }
            }
            __syncthreads();
            if (tid < HTB) {
                int32_t m = INT32_MAX;
#pragma unroll 8
                for (int g = 0; g < NG; ++g) {
                    const int32_t v0 = (NG == 32) ? sm.e32.slots32[0][g][tid] : sm.slots64[0][g][tid];
                    const int32_t v1 = (NG == 32) ? sm.e32.slots32[1][g][tid] : sm.slots64[1][g][tid];
                    m = min(m, max(v0, v1));
                }
                const int b = b0 + tid;
                if (b < B && a.s_splits > 1) {
                    atomicMax(&a.row_tau[b], m);
                    m = max(m, tau_other);
                }
                sm.tau[tid] = key2f(m);
            }
            __syncthreads();
            int npass[2], pos[2];
#pragma unroll
            for (int jb = 0; jb < 2; ++jb) {
                const float tau = sm.tau[wb * 64 + jb * 32 + l31];
                int n = 0;
#pragma unroll
                for (int sb = 0; sb < 4; ++sb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) n += (acc[sb][jb][r] >= tau && acc[sb][jb][r] > NEG_INF) ? 1 : 0;
                npass[jb] = (b0 + wb * 64 + jb * 32 + l31 < B) ? n : 0;
            }
#pragma unroll
            for (int jb = 0; jb < 2; ++jb) {
                pos[jb] = 0;
                if (npass[jb] > 0) pos[jb] = atomicAdd(&a.cand_cnt[b0 + wb * 64 + jb * 32 + l31], npass[jb]);
            }
#pragma unroll
            for (int jb = 0; jb < 2; ++jb) {
                if (npass[jb] > 0) {
                    const int bl_ = wb * 64 + jb * 32 + l31;
                    const float tau = sm.tau[bl_];
                    float* cv = a.cand_val + (size_t)(b0 + bl_) * a.cand_cap;
                    int32_t* ci = a.cand_idx + (size_t)(b0 + bl_) * a.cand_cap;
                    int p = pos[jb];
#pragma unroll
                    for (int sb = 0; sb < 4; ++sb)
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const float v = acc[sb][jb][r];
                            if (v >= tau && v > NEG_INF) {
                                if (p < a.cand_cap) {
                                    cv[p] = v;
                                    ci[p] = s0 + ws * 128 + sb * 32 + 8 * (r >> 2) + 4 * half + (r & 3);
                                }
                                ++p;
                            }
                        }
                }
            }
        }
        if (!prefetched && st + 1 < st_end) {
            __syncthreads();
            stage_async(0, s0 + HTS, 0);
        }
    }
}

}  // namespace

hipError_t launch_encode_f16x3(const EncodeF16Args& a, int epi, hipStream_t stream) {
    const int n_bblocks = (a.n_rows + HTB - 1) / HTB;
    dim3 grid(n_bblocks * a.s_splits), block(HTHREADS);
    const size_t smem = sizeof(HSmem);
    static bool attr_set = false;
    if (!attr_set) {
        const void* fns[3] = {reinterpret_cast<const void*>(&encode_f16x3_kernel<EPI_DENSE, 32>),
                              reinterpret_cast<const void*>(&encode_f16x3_kernel<EPI_TOPK, 32>),
                              reinterpret_cast<const void*>(&encode_f16x3_kernel<EPI_TOPK, 64>)};
        for (const void* f : fns) {
            hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
            if (e != hipSuccess) return e;
        }
        attr_set = true;
    }
    if (epi == EPI_DENSE)
        hipLaunchKernelGGL((encode_f16x3_kernel<EPI_DENSE, 32>), grid, block, smem, stream, a);
    else if (a.ngroups <= 32)
        hipLaunchKernelGGL((encode_f16x3_kernel<EPI_TOPK, 32>), grid, block, smem, stream, a);
    else
        hipLaunchKernelGGL((encode_f16x3_kernel<EPI_TOPK, 64>), grid, block, smem, stream, a);
    return hipGetLastError();
}

int encode_f16x3_tile_rows() { return HTB; }
int encode_f16x3_tile_latents() { return HTS; }
