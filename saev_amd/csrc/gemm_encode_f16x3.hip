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
// Inputs are the pre-split operands produced by split.hip, already in LDS image order: for every block of
// 256 rows (batch rows of x; latents of W_enc^T) and every 16-wide k-step one contiguous 16 KB image
//   [row 0..255][4 chunks of 8 halfs: (hi|lo) x (k 0-7 | k 8-15), chunk c of row r at position c ^ ((4 - ((r>>2)&3)) & 3)]
// so a k-step slot is filled by straight 1 KB-per-wave copies (every global_load_lds touches 8 full lines)
// and every MFMA fragment is one conflict-free ds_read_b128.
//
// Tile: 256 latents x 256 batch rows per 512-thread workgroup, 8 waves as 2 (s) x 4 (b), 128 x 64 per wave
// (4 x 2 MFMA blocks, 128 accumulator registers).  LDS is a ring of four 32 KB k-step slots filled by
// global_load_lds three k-steps ahead; the loop never drains the load queue (counted s_waitcnt vmcnt + raw
// s_barrier, one per k-step).  Rows of a slot are 64 bytes = 4 chunks of 16 B (hi/lo x lane-half); chunk c of
// row r sits at position c ^ ((4 - ((r >> 2) & 3)) & 3) (applied on the global source address), which makes the fragment
// reads conflict-free for both instruction shapes (encode_m16_kernel below).  Orientation and the TopK epilogue are those of gemm_encode.hip (lanes own batch rows).
#include "common.h"
#include "kernels.h"

namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));

constexpr int HTS = 256;  // latents per tile
constexpr int HTB = 256;  // batch rows per tile
constexpr int HTHREADS = 512;
constexpr int NSLOTS = 4;  // k-step ring

struct __attribute__((aligned(16))) KSlot {
    _Float16 a[HTS][32];  // W^T: [hi 16 | lo 16] per row, chunks swizzled     16 KB
    _Float16 b[HTB][32];  // x                                                  16 KB
};
struct __attribute__((aligned(16))) HSmem {
    union {
        KSlot slot[NSLOTS];  // 128 KB
        struct {
            KSlot keep[2];                 // slots 0,1 receive the next tile's first k-steps during the epilogue
            int32_t slots32[2][32][HTB];   // TopK scratch, 64 KB (slots 2,3): 32 group maxima per s-wave and row as int32
                                           // keys, or 64 as packed 16-bit keys
        } e32;
    };
    int32_t tau_key[HTB];
    int32_t ref[4][HTB];   // first-tile bound refinement: row maximum (key) and three counters
    float bias[2][HTS];  // per tile parity: a wave that is already in the next tile must not overwrite what a slower one still reads
};

template <int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (N > 0) {
        static_for<N - 1>(f);
        f(std::integral_constant<int, N - 1>());
    }
}

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ f32x16 mfma_bf16(half8 a, half8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

template <int AR>
__device__ __forceinline__ f32x16 mfma1(half8 a, half8 b, f32x16 c) {
    if constexpr (AR == 1) return mfma_bf16(a, b, c);
    else return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}

// AR = 0: the split-fp16 scheme above (three products).  AR = 1 / 2: single product on bf16 / fp16 operands: the same
// 16 KB images hold 32 consecutive k of the rounded x / W^T per row, chunk pairs (0,1) and (2,3) feed two
// v_mfma_f32_32x32x16_{bf16,f16} per block, fp32 accumulate; everything after the contraction is identical.  AR = 2
// is the first pass of SAEV_ENCODER_F16R: its pre-activations carry a bounded rounding error, the candidate cut is
// lowered by a per-row margin (a.row_margin) and select.hip recomputes the survivors exactly in fp32.
// HEUR: predicted row bounds (EncodeF16Args::heur_z) -- its own instantiation, so that the guaranteed-bound kernel's
// register allocation is not disturbed by code it never runs.
template <int EPI, int NG, int AR, bool HEUR = false>
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

    constexpr int NP = AR == 0 ? 3 : 1;
    const int nks = Dp / (NP == 3 ? 16 : 32);  // k-steps per tile
    // The 8 workgroups of an XCD that stream the same W images (same latent range, different batch block) walk the
    // k-steps in an order rotated by one step each, so they do not hit the same 16 KB at the same moment but stay
    // within the L2 retention window; the workgroups that share an x block (same batch block) keep the same order.
    // (Measured: 3.23 ms vs 3.40 ms unrotated; the sum over k does not care about the order.)
    const int rot = (bb & 7) % nks;
    auto kmap = [&](int t) { const int k = t + rot; return k >= nks ? k - nks : k; };

    // a slot image is 16 KB per operand; wave w copies bytes [2 KB * w, +2 KB) of each with two 1 KB calls
    const size_t img = (size_t)256 * 32;  // halfs per image
    const int blk_imgs = a.blk_imgs > 0 ? a.blk_imgs : nks;  // images per row block in memory
    const int k_first = (EPI == EPI_DENSE) ? (int)blockIdx.y * nks : 0;  // contraction slice of this batch
    const _Float16* x_imgs = a.xs + ((size_t)bb * blk_imgs + k_first) * img;
    // (staging written out as "uniform base + 32-bit lane offset" global_load_lds: see encode_m16_kernel.  The asm sets m0 itself
    // and does not list it as clobbered -- hipcc treats m0 as reserved and would only warn; that is sound because nothing
    // the compiler emits in this file keeps a value in m0: tools/check_m0.py proves it on the assembly at every build)
    const uint32_t lane_off = (uint32_t)lane * 16u;
    const uint32_t lds_w = (uint32_t)(size_t)(__attribute__((address_space(3))) char*)&sm.slot[0].a[wid * 32][0];
    auto stage_kstep = [&](int slot, int s0, int ks) {
        const char* wsrc = reinterpret_cast<const char*>(a.ws + ((size_t)(s0 / HTS) * blk_imgs + k_first + ks) * img) + wid * 2048;
        const char* xsrc = reinterpret_cast<const char*>(x_imgs + (size_t)ks * img) + wid * 2048;
        const uint32_t lds_a = lds_w + (uint32_t)slot * (uint32_t)sizeof(KSlot);
        asm volatile(
            "s_mov_b32 m0, %0\n\ts_nop 0\n\t"
            "global_load_lds_dwordx4 %2, %3\n\tglobal_load_lds_dwordx4 %2, %3 offset:1024\n\t"
            "s_mov_b32 m0, %1\n\ts_nop 0\n\t"
            "global_load_lds_dwordx4 %2, %4\n\tglobal_load_lds_dwordx4 %2, %4 offset:1024"
            ::"s"(lds_a), "s"(lds_a + (uint32_t)sizeof(_Float16) * HTS * 32), "v"(lane_off), "s"(wsrc), "s"(xsrc)
            : "memory");
    };

    // fragment rows of this lane and their chunk swizzles
    const int arow0 = ws * 128 + l31;             // + 32*sb
    const int brow0 = wb * 64 + l31;              // + 32*jb
    const int asw = (4 - ((l31 >> 2) & 3)) & 3;   // swizzle of row r: (4 - ((r >> 2) & 3)) & 3, independent of sb, ws
    const int bsw = asw;

    // k-steps 0 and 1 of a tile are requested by the previous tile's epilogue (or here for the first tile)
    if (st_begin < st_end) {
        stage_kstep(0, st_begin * HTS, kmap(0));
        if (nks > 1) stage_kstep(1, st_begin * HTS, kmap(1));
    }

    for (int st = st_begin; st < st_end; ++st) {
        const int s0 = st * HTS;
        f32x16 acc[4][2];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

        float* const bias_t = sm.bias[st & 1];
        // (latents past d_sae get a bias of -inf: their pre-activation is -inf without a select per accumulator; the dense
        // epilogue never stores them)
        if (tid < HTS) bias_t[tid] = (s0 + tid < S) ? a.b_enc[s0 + tid] : NEG_INF;
        // k-steps 0,1 must have landed in every wave (the staging is inline asm: the compiler does not know of these loads
        // and would only wait in the waves that also loaded a bias value)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();  // bias visible
        if (nks > 2) stage_kstep(2, s0, kmap(2));

        // one k-step; WAIT = loads that may stay in flight behind the one the next step needs (8: steady state, this step
        // staged k-step t + 3; 4 and 0: the ring runs empty at the end of the tile): three instantiations, no branch chain
        auto kstep = [&](int t, auto WAIT_) {
            constexpr int WAIT = decltype(WAIT_)::value;
            if constexpr (WAIT == 8) stage_kstep((t + 3) & 3, s0, kmap(t + 3));
            const KSlot& cs = sm.slot[t & 3];
            // 4 groups of 6 MFMAs (latent block sb).  The fragments of group sb+1 are requested right after the
            // first MFMA of group sb, so their LDS latency hides behind the other five.
            half8 fa[3][2];  // [ring][hi, lo]   A fragments of one latent block, requested two groups ahead
            half8 fb[2][2];  // [jb][hi, lo]     B fragments of this k-step
            auto load_a = [&](int set, int sb) {
                fa[set][0] = *reinterpret_cast<const half8*>(&cs.a[arow0 + 32 * sb][8 * ((0 + half) ^ asw)]);
                fa[set][1] = *reinterpret_cast<const half8*>(&cs.a[arow0 + 32 * sb][8 * ((2 + half) ^ asw)]);
            };
#pragma unroll
            for (int jb = 0; jb < 2; ++jb) {
                fb[jb][0] = *reinterpret_cast<const half8*>(&cs.b[brow0 + 32 * jb][8 * ((0 + half) ^ bsw)]);
                fb[jb][1] = *reinterpret_cast<const half8*>(&cs.b[brow0 + 32 * jb][8 * ((2 + half) ^ bsw)]);
            }
            load_a(0, 0);
            load_a(1, 1);
            // the two waves of a SIMD run this loop in step; raising the priority for the MFMA groups lets whichever
            // gets there first keep the matrix pipe while the other is still waiting on its fragments (-2 %, measured)
            __builtin_amdgcn_s_setprio(1);
            static_for<4>([&](auto G) {
                constexpr int sb = decltype(G)::value;
                constexpr int as = sb % 3;
                if constexpr (NP == 3) {
                    acc[sb][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[as][0], fb[0][0], acc[sb][0], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    if constexpr (sb + 2 < 4) load_a((sb + 2) % 3, sb + 2);
                    __builtin_amdgcn_sched_barrier(0);
                    acc[sb][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[as][0], fb[0][1], acc[sb][0], 0, 0, 0);
                    acc[sb][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[as][0], fb[1][0], acc[sb][1], 0, 0, 0);
                    acc[sb][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[as][0], fb[1][1], acc[sb][1], 0, 0, 0);
                    acc[sb][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[as][1], fb[0][0], acc[sb][0], 0, 0, 0);
                    acc[sb][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[as][1], fb[1][0], acc[sb][1], 0, 0, 0);
                } else {
                    acc[sb][0] = mfma1<AR>(fa[as][0], fb[0][0], acc[sb][0]);
                    __builtin_amdgcn_sched_barrier(0);
                    if constexpr (sb + 2 < 4) load_a((sb + 2) % 3, sb + 2);
                    __builtin_amdgcn_sched_barrier(0);
                    acc[sb][1] = mfma1<AR>(fa[as][0], fb[1][0], acc[sb][1]);
                    acc[sb][0] = mfma1<AR>(fa[as][1], fb[0][1], acc[sb][0]);
                    acc[sb][1] = mfma1<AR>(fa[as][1], fb[1][1], acc[sb][1]);
                }
                __builtin_amdgcn_sched_barrier(0);
            });
            __builtin_amdgcn_s_setprio(0);
            // k-step t+1 must have landed (this wave's part) before the barrier publishes it; newer requests
            // (t+2, t+3: 4 loads each) stay in flight.  Raw barrier: __syncthreads() would drain the queue.
            if constexpr (WAIT == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else if constexpr (WAIT == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        };
        {
            int t = 0;
            for (; t + 3 < nks; ++t) kstep(t, std::integral_constant<int, 8>());
            if (t + 2 < nks) { kstep(t, std::integral_constant<int, 4>()); ++t; }
            for (; t < nks; ++t) kstep(t, std::integral_constant<int, 0>());
        }
        // all slots are free.  Request the next tile's first two k-steps so they land during the epilogue
        // (the NG == 32 scratch only uses slots 2,3).
        const bool prefetched = st + 1 < st_end;  // (both TopK variants keep their scratch inside slots 2,3)
        if (prefetched) {
            stage_kstep(0, s0 + HTS, kmap(0));
            if (nks > 1) stage_kstep(1, s0 + HTS, kmap(1));
        }

        // ---------------- epilogue ----------------
        constexpr bool heur = HEUR;
        // lane owns batch rows bl(jb) = wb*64 + jb*32 + l31; latent of acc[sb][jb][r]:
        //   sl = ws*128 + sb*32 + 8*(r>>2) + 4*half + (r&3);   acc holds 2^8 * (x . w)
        const float unscale = a.scale_dev != nullptr ? 1.0f / (a.w_scale * a.scale_dev[0] * (a.scale_dev_b != nullptr ? a.scale_dev_b[0] : a.scale_dev[1])) : 1.0f / a.w_scale;
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
                            for (int e = 0; e < 4; ++e) v[e] = acc[sb][jb][4 * q + e] * unscale + bias_t[sl + e];
                            *reinterpret_cast<f32x4*>(a.h_out + (size_t)blockIdx.y * a.out_bstride + (size_t)b * S + s) = v;
                        }
                    }
            }
            __syncthreads();
        } else {
            if constexpr (NG == 64) {
                // ---- 64 groups, bound = K-th largest of the 64 merged group maxima (K = top_k) ----
                // Any K distinct seen values bound the K-th largest pre-activation from below; with 64 group maxima
                // the K-th largest of them is a much tighter bound than the minimum over 32 groups (candidates per row
                // at config 2: ~360 instead of ~980, simulated and measured).  Group of a latent: a fixed function of
                // its position modulo 64.  The tile's own maxima go through LDS as 16-bit keys (the ordered key shifted
                // down: rounded towards -inf, still a lower bound), two per word, so the scratch is the same 64 KB as the
                // 32-group variant and the next tile's first k-steps can still land during the epilogue.
                typedef short short2v __attribute__((ext_vector_type(2)));
                // bound phase: two lanes of one wave per row (lane and lane ^ 32), each owning 16 words = 32 groups.  The
                // published maxima of the first 16 are requested now, so that they arrive while the tile's own maxima
                // are formed; the other 16 follow after the barrier, while the first are merged.
                const int trow = wid * 32 + l31;
                const int tpart = lane >> 5;
                const int tb = b0 + trow;
                const bool share = tb < B;
                const uint32_t boff = (uint32_t)tb * 4u;
                // (group bases are wave-uniform -> SGPR pairs; the lane's half and row go into one 32-bit offset)
                const uint32_t voff = ((uint32_t)(32 * tpart) * (uint32_t)a.gmax_stride + (uint32_t)min(tb, B - 1)) * 4u;
                // (the empty asm makes each group's element offset an opaque uniform value: otherwise the compiler sees an
                // arithmetic progression, turns the 32 addresses into 64-bit VGPR pairs, hoists them out of the tile loop and
                // spills them; like this every access is "SGPR base + 32-bit lane offset")
                // (round 3: one 64-bit base and a 32-bit byte offset per group formed on the vector side -- see encode_m16_kernel)
                const uint32_t gstride4 = (uint32_t)a.gmax_stride * 4u;
#define GPTR(i)                                                                                          \
    ({                                                                                                   \
        uint32_t vo_ = voff;                                                                             \
        asm("" : "+v"(vo_));                                                                             \
        reinterpret_cast<int32_t*>(reinterpret_cast<char*>(a.gmax) + (size_t)(vo_ + (uint32_t)(i) * gstride4)); \
    })
                // pre-activations of the tile (every tile); the bound is refreshed on a subset of tiles only (see the
                // 32-group variant below)
#pragma unroll
                for (int sb = 0; sb < 4; ++sb)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int sl = ws * 128 + sb * 32 + 8 * q + 4 * half;
                        const f32x4 bq = *reinterpret_cast<const f32x4*>(&bias_t[sl]);
#pragma unroll
                        for (int jb = 0; jb < 2; ++jb)
#pragma unroll
                            for (int e = 0; e < 4; ++e) acc[sb][jb][4 * q + e] = fmaf(acc[sb][jb][4 * q + e], unscale, bq[e]);
                    }
                const int tile_no64 = st - st_begin;
                const bool refresh64 = tile_no64 < a.refresh_first || (tile_no64 & (a.refresh_every - 1)) == a.refresh_every - 1;
                if (refresh64) {
#pragma unroll
                for (int jb = 0; jb < 2; ++jb) {
                    float smax[32];
#pragma unroll
                    for (int r = 0; r < 32; ++r) smax[r] = fmaxf(acc[r >> 4][jb][r & 15], acc[(r >> 4) + 2][jb][r & 15]);
                    const int bl_ = wb * 64 + jb * 32 + l31;
#pragma unroll
                    for (int t = 0; t < 16; ++t) {  // word (2t + half): groups 2*(2t + half), 2*(2t + half) + 1
                        sm.e32.slots32[ws][2 * t + half][bl_] = (int32_t)__builtin_amdgcn_perm(
                            (uint32_t)f2key(smax[2 * t + 1]), (uint32_t)f2key(smax[2 * t]), 0x07060302u);  // {hi16(b), hi16(a)}
                    }
                }
                __syncthreads();
                {
                    int32_t old[32];
#pragma unroll
                    for (int i = 0; i < 32; ++i) {  // all global reads of this lane in flight together
                        old[i] = __hip_atomic_load(GPTR(i), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // (row clamped)
                    }
                    short2v mp[16];  // merged maxima, two 16-bit keys per register
                    uint32_t improved = 0;
#pragma unroll
                    for (int j = 0; j < 16; ++j) {
                        const int gp = 16 * tpart + j;
                        const short2v w0 = __builtin_bit_cast(short2v, sm.e32.slots32[0][gp][trow]);
                        const short2v w1 = __builtin_bit_cast(short2v, sm.e32.slots32[1][gp][trow]);
                        const short2v wm = __builtin_elementwise_max(w0, w1);
                        // the published halves of the two groups, packed like the tile's own
                        const short2v o = __builtin_bit_cast(
                            short2v, __builtin_amdgcn_perm((uint32_t)old[2 * j + 1], (uint32_t)old[2 * j], 0x07060302u));
                        mp[j] = __builtin_elementwise_max(wm, o);
                        improved |= __builtin_bit_cast(uint32_t, mp[j]) ^ __builtin_bit_cast(uint32_t, o);
                    }
                    if (share && improved != 0) {  // publish what this tile raised (late tiles: rarely anything)
#pragma unroll
                        for (int j = 0; j < 16; ++j) {
                            const int32_t a0 = (int32_t)mp[j][0], a1 = (int32_t)mp[j][1];
                            if (a0 > (old[2 * j] >> 16)) atomicMax(GPTR(2 * j), a0 << 16);
                            if (a1 > (old[2 * j + 1] >> 16)) atomicMax(GPTR(2 * j + 1), a1 << 16);
                        }
                    }
                    short2v mn = mp[0], mx = mp[0];
#pragma unroll
                    for (int j = 1; j < 16; ++j) { mn = __builtin_elementwise_min(mn, mp[j]); mx = __builtin_elementwise_max(mx, mp[j]); }
                    int32_t lo = min((int32_t)mn[0], (int32_t)mn[1]), hi = max((int32_t)mx[0], (int32_t)mx[1]);
                    lo = min(lo, __shfl_xor(lo, 32, 64));
                    hi = max(hi, __shfl_xor(hi, 32, 64));
                    // largest key T (to the resolution of four halvings of [lo, hi]) with at least top_k of the 64 maxima
                    // >= T; lo always satisfies it (all 64 are >= the minimum, and top_k <= 64)
                    const int K = a.top_k;
#pragma unroll 1
                    for (int it = 0; it < 4 && lo < hi && K < 64; ++it) {  // (K = 64: the minimum is the answer)
                        const int32_t mid = lo + ((hi - lo + 1) >> 1);
                        const short2v midp = {(short)mid, (short)mid};
                        uint32_t lt = 0;  // packed counters of (m < mid): low half / high half of the words
#pragma unroll
                        for (int j = 0; j < 16; ++j)
                            lt += (__builtin_bit_cast(uint32_t, __builtin_elementwise_sub_sat(mp[j], midp)) >> 15) & 0x00010001u;
                        int c = 32 - (int)((lt & 0xffffu) + (lt >> 16));
                        c += __shfl_xor(c, 32, 64);
                        if (c >= K) lo = mid; else hi = mid - 1;
                    }
                    if (tpart == 0) sm.tau_key[trow] = (int32_t)((uint32_t)lo << 16);
                }
#undef GPTR
                __syncthreads();
                }
            } else {
                // The bound is refreshed on a workgroup's first four tiles and on every fourth one after that; in between the
                // row bounds of the last refresh (still in sm.tau_key) are used as they are.  A bound only has to be a lower
                // bound of the row's k-th largest value, which an older one is; after the first few tiles it hardly moves any
                // more, and the refresh -- per-group maxima, their exchange through LDS and global memory, three barriers --
                // costs about as much as the candidate stores it saves from then on.
                const int tile_no = st - st_begin;
                const bool refresh = !heur && (tile_no < a.refresh_first || (tile_no & (a.refresh_every - 1)) == a.refresh_every - 1);
                // pre-activations of the tile (one code path for every tile: the accumulators are rewritten in one place)
    #pragma unroll
                for (int sb = 0; sb < 4; ++sb)
    #pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int sl = ws * 128 + sb * 32 + 8 * q + 4 * half;
                        // one 16-byte LDS read per four latents, unconditional (a per-element conditional read turns into
                        // 128 branches with an LDS round trip each)
                        const f32x4 bq = *reinterpret_cast<const f32x4*>(&bias_t[sl]);
    #pragma unroll
                        for (int jb = 0; jb < 2; ++jb)
    #pragma unroll
                            for (int e = 0; e < 4; ++e) acc[sb][jb][4 * q + e] = fmaf(acc[sb][jb][4 * q + e], unscale, bq[e]);
                    }
                if (refresh) {
                // group maxima of THIS tile only; the running maxima over earlier tiles (of this and of every other
                // workgroup that owns the same rows) live in a.gmax and are merged below, so nothing is carried in
                // registers across the contraction loop
                float smax[2][NSLOT];
    #pragma unroll
                for (int jb = 0; jb < 2; ++jb)
    #pragma unroll
                    for (int r = 0; r < NSLOT; ++r)
                        smax[jb][r] = fmaxf(fmaxf(acc[0][jb][r], acc[1][jb][r]), fmaxf(acc[2][jb][r], acc[3][jb][r]));
    #pragma unroll
                for (int jb = 0; jb < 2; ++jb) {
                    const int bl_ = wb * 64 + jb * 32 + l31;
    #pragma unroll
                    for (int r = 0; r < NSLOT; ++r) {
                        sm.e32.slots32[ws][2 * r + half][bl_] = f2key(smax[jb][r]);
                    }
                }
                __syncthreads();
                // bound per row = min over groups of the group maximum, where each group maximum is merged (a) across the
                // two s-waves of this workgroup and (b) with what the row's other latent ranges have published so far in
                // global memory (relaxed L2 reads: a stale value only gives a weaker, still valid bound).  Two threads per
                // row, each owning half of the groups; improved maxima are published fire-and-forget.
                if (tid < HTB) sm.tau_key[tid] = INT32_MAX;
                __syncthreads();
                {
                    const int row = tid % HTB;
                    const int part = __builtin_amdgcn_readfirstlane(tid / HTB);  // wave-uniform: group bases stay in SGPRs  // 512 threads = 2 x HTB rows
                    constexpr int GPT = NG / 2;                     // groups per thread
                    const int b = b0 + row;
                    const bool share = b < B;
                    const uint32_t boff = (uint32_t)b * 4u;
                    int32_t m = INT32_MAX;
                    int32_t old[GPT];
    #pragma unroll
                    for (int i = 0; i < GPT; ++i) {  // all global reads of this thread in flight together
                        old[i] = INT32_MIN;
                        if (share)
                            old[i] = __hip_atomic_load(
                                reinterpret_cast<int32_t*>(reinterpret_cast<char*>(a.gmax + (size_t)(part * GPT + i) * a.gmax_stride) + boff),
                                __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
    #pragma unroll
                    for (int i = 0; i < GPT; ++i) {
                        const int g = part * GPT + i;
                        const int32_t v0 = sm.e32.slots32[0][g][row];
                        const int32_t v1 = sm.e32.slots32[1][g][row];
                        const int32_t v = max(v0, v1);
                        if (share && v > old[i])
                            atomicMax(reinterpret_cast<int32_t*>(reinterpret_cast<char*>(a.gmax + (size_t)g * a.gmax_stride) + boff), v);
                        m = min(m, max(v, old[i]));
                    }
                    atomicMin(&sm.tau_key[row], m);
                }
                __syncthreads();
                }
            }
            if constexpr (HEUR) if (st == st_begin) {
                // Predicted bound (EncodeF16Args::heur_z): the row's pre-activations over THIS tile's 256 latents are a
                // sample of its 32 k; their mean and standard deviation put the bound at mean + z * sigma, where z (a device
                // scalar the host side adapts from step to step) is chosen so that a small multiple of top_k values exceed
                // it.  Nothing guarantees that k values do -- so the largest bound any workgroup used for a row is
                // published in tau_max and the select stage checks that the k-th largest candidate it found is not below
                // it; when that fails for any row the launch is repeated with the guaranteed bounds above.  The pay-off: no
                // group maxima, no exchange, no barrier in the epilogue of any later tile, and an order of magnitude
                // fewer candidates to store and to select from.  Two passes (mean, then centred squares): the values can
                // share a large offset.
                // Both moments are taken twice: over everything, then over the values within two of those standard
                // deviations of that mean -- a handful of far-out values (latents with a large negative bias, a few strongly
                // active features) must not set the scale of the bulk the bound is extrapolated from.
                float* const mom = reinterpret_cast<float*>(&sm.ref[0][0]);  // [0] sum  [1] centred squares  [2] count  (per row)
                const int n_valid = min(HTS, S - s0);  // real latents of the tile (padding carries -inf)
                float lo[2] = {-3.0e38f, -3.0e38f}, hi[2] = {3.4e38f, 3.4e38f};
                float mean2[2] = {0.f, 0.f}, sd2[2] = {0.f, 0.f};
#pragma unroll 1
                for (int round = 0; round < 2; ++round) {
                    if (tid < HTB) { mom[tid] = 0.f; mom[HTB + tid] = 0.f; mom[2 * HTB + tid] = 0.f; }
                    __syncthreads();
#pragma unroll
                    for (int jb = 0; jb < 2; ++jb) {
                        float s1 = 0.f, cnt = 0.f;
#pragma unroll
                        for (int sb = 0; sb < 4; ++sb)
#pragma unroll
                            for (int r = 0; r < 16; ++r) {
                                const float v = acc[sb][jb][r];
                                const bool in = v > lo[jb] && v < hi[jb];
                                s1 += in ? v : 0.f;
                                cnt += in ? 1.f : 0.f;
                            }
                        atomicAdd(&mom[wb * 64 + jb * 32 + l31], s1);
                        atomicAdd(&mom[2 * HTB + wb * 64 + jb * 32 + l31], cnt);
                    }
                    __syncthreads();
#pragma unroll
                    for (int jb = 0; jb < 2; ++jb) {
                        const int row = wb * 64 + jb * 32 + l31;
                        const float mean = mom[row] / fmaxf(mom[2 * HTB + row], 1.f);
                        float s2 = 0.f;
#pragma unroll
                        for (int sb = 0; sb < 4; ++sb)
#pragma unroll
                            for (int r = 0; r < 16; ++r) {
                                const float v = acc[sb][jb][r];
                                const float dv = (v > lo[jb] && v < hi[jb]) ? v - mean : 0.f;
                                s2 = fmaf(dv, dv, s2);
                            }
                        atomicAdd(&mom[HTB + row], s2);
                        mean2[jb] = mean;
                    }
                    __syncthreads();
#pragma unroll
                    for (int jb = 0; jb < 2; ++jb) {
                        const int row = wb * 64 + jb * 32 + l31;
                        sd2[jb] = sqrtf(mom[HTB + row] / fmaxf(mom[2 * HTB + row], 1.f));
                        lo[jb] = fmaxf(mean2[jb] - 2.0f * sd2[jb], -3.0e38f);
                        hi[jb] = mean2[jb] + 2.0f * sd2[jb];
                    }
                    __syncthreads();
                }
                (void)n_valid;
                if (ws == 0 && half == 0) {
                    // (every latent range of the row estimates its own bound and the largest one is what the verification
                    // holds the row to: the more ranges, the further that maximum sits above a single estimate, whose
                    // standard error from 256 samples is about 0.12 sigma)
                    const float z_eff = *a.heur_z - 0.12f * sqrtf(2.0f * logf((float)a.s_splits));
#pragma unroll
                    for (int jb = 0; jb < 2; ++jb) {
                        const int row = wb * 64 + jb * 32 + l31;
                        const int32_t key = f2key(mean2[jb] + z_eff * sd2[jb]);
                        sm.tau_key[row] = key;
                        if (b0 + row < B) atomicMax(&a.tau_max[b0 + row], key);
                    }
                }
                __syncthreads();
            }
            if (NG == 32 && !heur && st == st_begin && a.top_k <= HTS / 4) {
                // First tile of this workgroup: the rows' other latent ranges start at the same moment, so the shared group
                // maxima are still empty and the bound above is only "the minimum of 32 maxima of 8 values" -- about the median
                // of the tile, where the k-th largest of its 256 values is what one would like.  Any threshold that at
                // least top_k values of THIS tile reach is a valid bound too: try three between the group bound and the
                // row maximum and keep the largest that qualifies.  Done once per workgroup, it removes about 40 % of the
                // candidates of the first round, which is where 40 % of all candidates come from.
                if (tid < HTB) { sm.ref[0][tid] = INT32_MIN; sm.ref[1][tid] = 0; sm.ref[2][tid] = 0; sm.ref[3][tid] = 0; }
                __syncthreads();
#pragma unroll
                for (int jb = 0; jb < 2; ++jb) {
                    float m = NEG_INF;
#pragma unroll
                    for (int sb = 0; sb < 4; ++sb)
#pragma unroll
                        for (int r = 0; r < 16; ++r) m = fmaxf(m, acc[sb][jb][r]);
                    atomicMax(&sm.ref[0][wb * 64 + jb * 32 + l31], f2key(m));
                }
                __syncthreads();
                float tg[2][3];
#pragma unroll
                for (int jb = 0; jb < 2; ++jb) {
                    const int row = wb * 64 + jb * 32 + l31;
                    const float t0 = key2f(sm.tau_key[row]), M = key2f(sm.ref[0][row]);
                    const bool usable = t0 > -3.0e38f && M > t0;
                    const float span = usable ? (M - t0) : 0.f;
                    tg[jb][0] = t0 + 0.25f * span; tg[jb][1] = t0 + 0.40f * span; tg[jb][2] = t0 + 0.55f * span;
                    int c0 = 0, c1 = 0, c2 = 0;
#pragma unroll
                    for (int sb = 0; sb < 4; ++sb)
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const float v = acc[sb][jb][r];
                            c0 += (v >= tg[jb][0]) ? 1 : 0; c1 += (v >= tg[jb][1]) ? 1 : 0; c2 += (v >= tg[jb][2]) ? 1 : 0;
                        }
                    if (usable) { atomicAdd(&sm.ref[1][row], c0); atomicAdd(&sm.ref[2][row], c1); atomicAdd(&sm.ref[3][row], c2); }
                }
                __syncthreads();
                if (ws == 0 && half == 0) {  // one lane per row writes the refined bound back
#pragma unroll
                    for (int jb = 0; jb < 2; ++jb) {
                        const int row = wb * 64 + jb * 32 + l31;
                        float best = key2f(sm.tau_key[row]);
                        if (sm.ref[1][row] >= a.top_k) best = fmaxf(best, tg[jb][0]);
                        if (sm.ref[2][row] >= a.top_k) best = fmaxf(best, tg[jb][1]);
                        if (sm.ref[3][row] >= a.top_k) best = fmaxf(best, tg[jb][2]);
                        sm.tau_key[row] = f2key(best);
                    }
                }
                __syncthreads();
            }
            int npass[2], pos[2];
            float tau2[2];
#pragma unroll
            for (int jb = 0; jb < 2; ++jb) {
                // a finite bound: padded latents carry -inf and must never pass (a row whose bound is still -inf has seen
                // fewer than k groups with a real value; everything real passes then)
                float tau = fmaxf(key2f(sm.tau_key[wb * 64 + jb * 32 + l31]), -3.0e38f);
                if (a.row_margin != nullptr) {  // approximate first pass: keep everything that could still be in the exact top-k
                    const int b = b0 + wb * 64 + jb * 32 + l31;
                    tau -= (b < B) ? a.row_margin[b] : 0.f;
                }
                tau2[jb] = tau;
                int n = 0;
#pragma unroll
                for (int sb = 0; sb < 4; ++sb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) n += (acc[sb][jb][r] >= tau) ? 1 : 0;
                npass[jb] = (b0 + wb * 64 + jb * 32 + l31 < B) ? n : 0;
            }
            // the two lanes of a row (lane, lane ^ 32) reserve its list space with ONE atomic: several lanes of one
            // instruction hitting the same counter serialise in L2, and the counters were the most expensive part of this
            // epilogue (encode_m16_kernel: 8 -> 2 atomics per row and tile took 1.30 -> 1.19 ms)
            typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
            int rowtot[2], pre_[2];
#pragma unroll
            for (int jb = 0; jb < 2; ++jb) {
                const u32x2 q = __builtin_amdgcn_permlane32_swap((unsigned)npass[jb], (unsigned)npass[jb], false, false);  // {half 0's, half 1's}
                rowtot[jb] = (int)(q[0] + q[1]);
                const int pre = half ? (int)q[0] : 0;
                int base = 0;
                if (half == 0 && rowtot[jb] > 0) base = atomicAdd(&a.cand_cnt[b0 + wb * 64 + jb * 32 + l31], rowtot[jb]);
                pos[jb] = base;
                pre_[jb] = pre;
            }
            // wait for the two counters once, here: otherwise every conditionally executed store block below gets its own
            // s_waitcnt vmcnt(0) (the block before it may have been skipped), which also serialises the stores
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            asm volatile("" : "+v"(pos[0]), "+v"(pos[1]));
#pragma unroll
            for (int jb = 0; jb < 2; ++jb) {
                const u32x2 h = __builtin_amdgcn_permlane32_swap((unsigned)pos[jb], (unsigned)pos[jb], false, false);  // half 0's -> both
                const int row_base = (int)h[0];
                pos[jb] = row_base + pre_[jb];
                // a row whose list would overflow is not written at all: its counter already says so, and the step then
                // re-runs on the exact dense route (overflow_check).  Offsets are 32-bit from the uniform buffer bases
                // (n_rows * cand_stride * 4 < 2^32), so a kept value costs one address add and two stores.
                if (npass[jb] > 0 && row_base + rowtot[jb] <= a.cand_cap) {
                    const int bl_ = wb * 64 + jb * 32 + l31;
                    const float tau = tau2[jb];
                    uint32_t off = ((uint32_t)(b0 + bl_) * (uint32_t)a.cand_stride + (uint32_t)pos[jb]) * 4u;
#pragma unroll
                    for (int sb = 0; sb < 4; ++sb)
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const float v = acc[sb][jb][r];
                            if (v >= tau) {
                                *reinterpret_cast<float*>(reinterpret_cast<char*>(a.cand_val) + off) = v;
                                *reinterpret_cast<int32_t*>(reinterpret_cast<char*>(a.cand_idx) + off) =
                                    s0 + ws * 128 + sb * 32 + 8 * (r >> 2) + 4 * half + (r & 3);
                                off += 4u;
                            }
                        }
                }
            }
        }
        if (!prefetched && st + 1 < st_end) {
            __syncthreads();  // NG == 64: the scratch covered every slot
            stage_kstep(0, s0 + HTS, kmap(0));
            if (nks > 1) stage_kstep(1, s0 + HTS, kmap(1));
        }
    }
}


// -------------------------------------------------------------------------------------------------------------------------
// The single-product TopK kernel (AR = 1 bf16 / AR = 2 fp16 first pass; 32 or 64 groups, guaranteed bounds) on
// v_mfma_f32_16x16x32_{bf16,f16} instead of 32x32x16.  Same tile, ring, staging, images and LDS bytes per k-step (twelve
// ds_read_b128 per wave); the instruction has a quarter of the accumulator registers per flop to read and write back, and
// on real operands the matrix pipes -- which are clock-limited by power, not by issue -- sustain 1.93 PFLOP/s with it
// against 1.66 (tools/ubench/mfma_issue.hip, profiles/r02_mfma_issue.txt).  What changes is who owns what:
//   wave tile 128 latents x 64 rows = 8 x 4 blocks of 16 x 16; lane = 16 * kg + l15 holds, of block (sb, jb), batch row
//   wb*64 + jb*16 + l15 and latents ws*128 + sb*16 + 4*kg + e (e = 0..3) -- four lanes and four blocks per row where the
//   32 x 32 layout has two and two.  A/B fragment of a lane: its row's chunk kg (k 8kg .. 8kg+7 of the k-step's 32),
//   one ds_read_b128; with the image swizzle (position c ^ ((4 - (r >> 2)) & 3)) each of the instruction's four service
//   groups touches all 64 banks once.
// Group of a latent (for the 32 shared group maxima): its position modulo 32.
// cache policy of encode_m16_kernel's staging loads (" nt", " sc1", ...: experiments; the shipped kernel uses the default for both --
// tools/experiments/r5_enc_policy.sh)
#ifndef SAEV_ENC_W_POLICY
#define SAEV_ENC_W_POLICY ""
#endif
#ifndef SAEV_ENC_X_POLICY
#define SAEV_ENC_X_POLICY ""
#endif
template <int AR>
__device__ __forceinline__ f32x4 mfma16(half8 a, half8 b, f32x4 c) {
    if constexpr (AR == 1)
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}

template <int AR, int NG = 32>
__global__ __launch_bounds__(HTHREADS, 2) void encode_m16_kernel(EncodeF16Args a) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    HSmem& sm = *reinterpret_cast<HSmem*>(smem_raw);

    if (a.enable_flag != nullptr && (*a.enable_flag != 0) != (a.enable_when != 0)) return;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ws = wid & 1;   // wave position along s (128 latents each)
    const int wb = wid >> 1;  // wave position along b (64 rows each)
    const int l15 = lane & 15;
    const int kg = lane >> 4;

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

    const int nks = Dp / 32;  // k-steps per tile
    const int rot = a.no_rot ? 0 : (bb & 7) % nks;  // see encode_f16x3_kernel (saev_debug_cfg.enc_rot = 1: lock step)
    auto kmap = [&](int t) { const int k = t + rot; return k >= nks ? k - nks : k; };

    const size_t img = (size_t)256 * 32;  // halfs per image
    const int blk_imgs = a.blk_imgs > 0 ? a.blk_imgs : nks;
    const _Float16* x_imgs = a.xs + (size_t)bb * blk_imgs * img;
    // Staging in its "uniform base + 32-bit lane offset" form, written out: what hipcc makes of the builtin is a 64-bit
    // VALU add per request (four per k-step, in the middle of the MFMA stream).  Wave w copies bytes [2 KB * w, + 2 KB)
    // of each 16 KB image with two 1 KB requests; the instruction offset moves the global and the LDS address alike, so
    // the second request of an image needs no address of its own.
    const uint32_t lane_off = (uint32_t)lane * 16u;
    const uint32_t lds_w = (uint32_t)(size_t)(__attribute__((address_space(3))) char*)&sm.slot[0].a[wid * 32][0];
    auto stage_kstep = [&](int slot, int s0, int ks) {
        const char* wsrc = reinterpret_cast<const char*>(a.ws + ((size_t)(s0 / HTS) * blk_imgs + ks) * img) + wid * 2048;
        const char* xsrc = reinterpret_cast<const char*>(x_imgs + (size_t)ks * img) + wid * 2048;
        const uint32_t lds_a = lds_w + (uint32_t)slot * (uint32_t)sizeof(KSlot);
        asm volatile(
            "s_mov_b32 m0, %0\n\ts_nop 0\n\t"
            "global_load_lds_dwordx4 %2, %3" SAEV_ENC_W_POLICY "\n\tglobal_load_lds_dwordx4 %2, %3 offset:1024" SAEV_ENC_W_POLICY "\n\t"
            "s_mov_b32 m0, %1\n\ts_nop 0\n\t"
            "global_load_lds_dwordx4 %2, %4" SAEV_ENC_X_POLICY "\n\tglobal_load_lds_dwordx4 %2, %4 offset:1024" SAEV_ENC_X_POLICY
            ::"s"(lds_a), "s"(lds_a + (uint32_t)sizeof(_Float16) * HTS * 32), "v"(lane_off), "s"(wsrc), "s"(xsrc)
            : "memory");
    };

    const int arow0 = ws * 128 + l15;  // + 16 * sb
    const int brow0 = wb * 64 + l15;   // + 16 * jb
    const int coff = 8 * (kg ^ ((4 - (l15 >> 2)) & 3));  // this lane's chunk within its rows (halfs)

    if (st_begin < st_end) {
        stage_kstep(0, st_begin * HTS, kmap(0));
        if (nks > 1) stage_kstep(1, st_begin * HTS, kmap(1));
    }

    for (int st = st_begin; st < st_end; ++st) {
        const int s0 = st * HTS;
        f32x4 acc[8][4];
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

        float* const bias_t = sm.bias[st & 1];
        if (tid < HTS) bias_t[tid] = (s0 + tid < S) ? a.b_enc[s0 + tid] : NEG_INF;
        // k-steps 0,1 must have landed in every wave (the staging is inline asm: the compiler does not know of these loads
        // and would only wait in the waves that also loaded a bias value)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();  // bias visible
        if (nks > 2) stage_kstep(2, s0, kmap(2));
        // the in-loop staging walks the images of this tile by a running byte offset (k index (t + 3 + rot) mod nks) from two
        // uniform bases: two scalar add-with-carry per request pair instead of the whole index arithmetic
        const char* const w_tile = reinterpret_cast<const char*>(a.ws + (size_t)(s0 / HTS) * blk_imgs * img) + wid * 2048;
        const char* const x_blk = reinterpret_cast<const char*>(x_imgs) + wid * 2048;
        const uint32_t run_bytes = (uint32_t)nks * (uint32_t)(img * sizeof(_Float16));
        uint32_t koff = (uint32_t)kmap(nks > 3 ? 3 : 0) * (uint32_t)(img * sizeof(_Float16));
        auto stage_next = [&](int slot) {
            const uint32_t lds_a = lds_w + (uint32_t)slot * (uint32_t)sizeof(KSlot);
            asm volatile(
                "s_mov_b32 m0, %0\n\ts_nop 0\n\t"
                "global_load_lds_dwordx4 %2, %3" SAEV_ENC_W_POLICY "\n\tglobal_load_lds_dwordx4 %2, %3 offset:1024" SAEV_ENC_W_POLICY "\n\t"
                "s_mov_b32 m0, %1\n\ts_nop 0\n\t"
                "global_load_lds_dwordx4 %2, %4" SAEV_ENC_X_POLICY "\n\tglobal_load_lds_dwordx4 %2, %4 offset:1024" SAEV_ENC_X_POLICY
                ::"s"(lds_a), "s"(lds_a + (uint32_t)sizeof(_Float16) * HTS * 32), "v"(lane_off), "s"(w_tile + koff), "s"(x_blk + koff)
                : "memory");
            koff += (uint32_t)(img * sizeof(_Float16));
            koff = koff == run_bytes ? 0u : koff;
        };

        // one k-step; WAIT = loads that may stay in flight behind the one the next step needs (8: the steady state, this
        // step staged k-step t + 3; 4 and 0: the ring runs empty at the end of the tile).  Three instantiations instead of a
        // branch chain per step.
        auto kstep = [&](int t, auto WAIT_, auto SLOT_) {
            constexpr int WAIT = decltype(WAIT_)::value;
            constexpr int SLOT = decltype(SLOT_)::value;  // t & 3, or -1: not known at compile time
            if constexpr (WAIT == 8) stage_next(SLOT >= 0 ? (SLOT + 3) & 3 : (t + 3) & 3);
            const KSlot& cs = sm.slot[SLOT >= 0 ? SLOT : t & 3];
            // 8 groups of 4 MFMAs (latent block sb); the A fragment of group sb + 2 is requested after the first MFMA of
            // group sb
            half8 fa[3], fb[4];
            auto load_a = [&](int set, int sb) { fa[set] = *reinterpret_cast<const half8*>(&cs.a[arow0 + 16 * sb][coff]); };
#pragma unroll
            for (int jb = 0; jb < 4; ++jb) fb[jb] = *reinterpret_cast<const half8*>(&cs.b[brow0 + 16 * jb][coff]);
            load_a(0, 0);
            load_a(1, 1);
            static_for<8>([&](auto G) {
                constexpr int sb = decltype(G)::value;
                constexpr int as = sb % 3;
                // (odd latent blocks walk the row blocks 3..0: the B operand of a group's last MFMA is that of the next group's first,
                // 32 operand changes in front of the matrix pipe per k-step instead of 40 -- worth 0.7 % of this power-limited loop,
                // tools/ubench/enc_loop2.hip SNAKE)
                constexpr int j0 = (sb & 1) ? 3 : 0, j1 = (sb & 1) ? 2 : 1, j2 = (sb & 1) ? 1 : 2, j3 = (sb & 1) ? 0 : 3;
                acc[sb][j0] = mfma16<AR>(fa[as], fb[j0], acc[sb][j0]);
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (sb + 2 < 8) load_a((sb + 2) % 3, sb + 2);
                __builtin_amdgcn_sched_barrier(0);
                acc[sb][j1] = mfma16<AR>(fa[as], fb[j1], acc[sb][j1]);
                acc[sb][j2] = mfma16<AR>(fa[as], fb[j2], acc[sb][j2]);
                acc[sb][j3] = mfma16<AR>(fa[as], fb[j3], acc[sb][j3]);
                __builtin_amdgcn_sched_barrier(0);
            });
            if constexpr (WAIT == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else if constexpr (WAIT == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        };
        {
            // the steady state four steps at a time: the ring slot of every step is a constant, so the fragment reads and
            // the staging destinations are immediate offsets from loop-invariant bases
            using IC8 = std::integral_constant<int, 8>;
            using NOSLOT = std::integral_constant<int, -1>;
            int t = 0;
            for (; t + 6 < nks; t += 4) {
                kstep(t, IC8(), std::integral_constant<int, 0>());
                kstep(t + 1, IC8(), std::integral_constant<int, 1>());
                kstep(t + 2, IC8(), std::integral_constant<int, 2>());
                kstep(t + 3, IC8(), std::integral_constant<int, 3>());
            }
            for (; t + 3 < nks; ++t) kstep(t, IC8(), NOSLOT());
            if (t + 2 < nks) { kstep(t, std::integral_constant<int, 4>(), NOSLOT()); ++t; }
            for (; t < nks; ++t) kstep(t, std::integral_constant<int, 0>(), NOSLOT());
        }
        const bool prefetched = st + 1 < st_end;
        if (prefetched) {
            stage_kstep(0, s0 + HTS, kmap(0));
            if (nks > 1) stage_kstep(1, s0 + HTS, kmap(1));
        }

        // ---------------- epilogue (the 32-group TopK epilogue of encode_f16x3_kernel in this kernel's ownership) -------
        // lane owns batch rows bl(jb) = wb*64 + jb*16 + l15; latent of acc[sb][jb][e]: sl = ws*128 + sb*16 + 4*kg + e
        const float unscale = a.scale_dev != nullptr ? 1.0f / (a.w_scale * a.scale_dev[0] * (a.scale_dev_b != nullptr ? a.scale_dev_b[0] : a.scale_dev[1])) : 1.0f / a.w_scale;
        const int tile_no = st - st_begin;
        const bool refresh = tile_no < a.refresh_first || (tile_no & (a.refresh_every - 1)) == a.refresh_every - 1;
#pragma unroll
        for (int sb = 0; sb < 8; ++sb) {
            const f32x4 bq = *reinterpret_cast<const f32x4*>(&bias_t[ws * 128 + sb * 16 + 4 * kg]);
            // two values per instruction (v_pk_fma_f32)
            typedef float f32x2 __attribute__((ext_vector_type(2)));
            const f32x2 u2 = {unscale, unscale}, b01 = {bq[0], bq[1]}, b23 = {bq[2], bq[3]};
#pragma unroll
            for (int jb = 0; jb < 4; ++jb) {
                const f32x2 lo = __builtin_elementwise_fma(f32x2{acc[sb][jb][0], acc[sb][jb][1]}, u2, b01);
                const f32x2 hi = __builtin_elementwise_fma(f32x2{acc[sb][jb][2], acc[sb][jb][3]}, u2, b23);
                acc[sb][jb] = f32x4{lo[0], lo[1], hi[0], hi[1]};
            }
        }
        if constexpr (NG == 64) {
            // 64 groups (top_k up to 64): bound = top_k-th largest of the 64 merged group maxima, as in encode_f16x3_kernel's
            // 64-group variant -- the same LDS words (two 16-bit keys each) and the same bound phase; only who supplies
            // which word differs.  Group of a latent: its position modulo 64 = (sb & 3) * 16 + 4 * kg + e; word = group / 2.
            if (refresh) {
                typedef short short2v __attribute__((ext_vector_type(2)));
                const int trow = wid * 32 + (lane & 31);
                const int tpart = lane >> 5;
                const int tb = b0 + trow;
                const bool share = tb < B;
                const uint32_t boff = (uint32_t)tb * 4u;
                // (group bases are wave-uniform -> SGPR pairs; the lane's half and row go into one 32-bit offset)
                const uint32_t voff = ((uint32_t)(32 * tpart) * (uint32_t)a.gmax_stride + (uint32_t)min(tb, B - 1)) * 4u;
                // (the empty asm makes each group's element offset an opaque uniform value: otherwise the compiler sees an
                // arithmetic progression, turns the 32 addresses into 64-bit VGPR pairs, hoists them out of the tile loop and
                // spills them; like this every access is "SGPR base + 32-bit lane offset")
                // (round 3: ONE 64-bit base -- a.gmax -- and a 32-bit byte offset per group formed on the vector side, instead
                // of a 64-bit uniform base per group: 32 SGPR pairs live across the phase were 64 spilled SGPRs)
                const uint32_t gstride4 = (uint32_t)a.gmax_stride * 4u;
#define GPTR(i)                                                                                          \
    ({                                                                                                   \
        uint32_t vo_ = voff;                                                                             \
        asm("" : "+v"(vo_)); /* opaque: the 32 offsets are formed where they are used (one v_mad each) */  \
        reinterpret_cast<int32_t*>(reinterpret_cast<char*>(a.gmax) + (size_t)(vo_ + (uint32_t)(i) * gstride4)); \
    })
#pragma unroll
                for (int jb = 0; jb < 4; ++jb) {
                    const int bl_ = wb * 64 + jb * 16 + l15;
#pragma unroll
                    for (int q = 0; q < 4; ++q)
#pragma unroll
                        for (int h = 0; h < 2; ++h) {
                            const float m0 = fmaxf(acc[q][jb][2 * h], acc[q + 4][jb][2 * h]);
                            const float m1 = fmaxf(acc[q][jb][2 * h + 1], acc[q + 4][jb][2 * h + 1]);
                            sm.e32.slots32[ws][q * 8 + 2 * kg + h][bl_] =
                                (int32_t)__builtin_amdgcn_perm((uint32_t)f2key(m1), (uint32_t)f2key(m0), 0x07060302u);  // {hi16(m1), hi16(m0)}
                        }
                }
                __syncthreads();
                {
                    int32_t old[32];
#pragma unroll
                    for (int i = 0; i < 32; ++i) {  // all global reads of this lane in flight together
                        old[i] = __hip_atomic_load(GPTR(i), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // (row clamped)
                    }
                    short2v mp[16];  // merged maxima, two 16-bit keys per register
                    uint32_t improved = 0;
#pragma unroll
                    for (int j = 0; j < 16; ++j) {
                        const int gp = 16 * tpart + j;
                        const short2v w0 = __builtin_bit_cast(short2v, sm.e32.slots32[0][gp][trow]);
                        const short2v w1 = __builtin_bit_cast(short2v, sm.e32.slots32[1][gp][trow]);
                        const short2v wm = __builtin_elementwise_max(w0, w1);
                        // the published halves of the two groups, packed like the tile's own
                        const short2v o = __builtin_bit_cast(
                            short2v, __builtin_amdgcn_perm((uint32_t)old[2 * j + 1], (uint32_t)old[2 * j], 0x07060302u));
                        mp[j] = __builtin_elementwise_max(wm, o);
                        improved |= __builtin_bit_cast(uint32_t, mp[j]) ^ __builtin_bit_cast(uint32_t, o);
                    }
                    if (share && improved != 0) {  // publish what this tile raised (late tiles: rarely anything)
#pragma unroll
                        for (int j = 0; j < 16; ++j) {
                            const int32_t a0 = (int32_t)mp[j][0], a1 = (int32_t)mp[j][1];
                            if (a0 > (old[2 * j] >> 16)) atomicMax(GPTR(2 * j), a0 << 16);
                            if (a1 > (old[2 * j + 1] >> 16)) atomicMax(GPTR(2 * j + 1), a1 << 16);
                        }
                    }
                    short2v mn = mp[0], mx = mp[0];
#pragma unroll
                    for (int j = 1; j < 16; ++j) { mn = __builtin_elementwise_min(mn, mp[j]); mx = __builtin_elementwise_max(mx, mp[j]); }
                    int32_t lo = min((int32_t)mn[0], (int32_t)mn[1]), hi = max((int32_t)mx[0], (int32_t)mx[1]);
                    lo = min(lo, __shfl_xor(lo, 32, 64));
                    hi = max(hi, __shfl_xor(hi, 32, 64));
                    // largest key T (to the resolution of four halvings of [lo, hi]) with at least top_k of the 64 maxima
                    // >= T; lo always satisfies it (all 64 are >= the minimum, and top_k <= 64)
                    const int K = a.top_k;
#pragma unroll 1
                    for (int it = 0; it < 4 && lo < hi && K < 64; ++it) {  // (K = 64: the minimum is the answer)
                        const int32_t mid = lo + ((hi - lo + 1) >> 1);
                        const short2v midp = {(short)mid, (short)mid};
                        uint32_t lt = 0;  // packed counters of (m < mid): low half / high half of the words
#pragma unroll
                        for (int j = 0; j < 16; ++j)
                            lt += (__builtin_bit_cast(uint32_t, __builtin_elementwise_sub_sat(mp[j], midp)) >> 15) & 0x00010001u;
                        int c = 32 - (int)((lt & 0xffffu) + (lt >> 16));
                        c += __shfl_xor(c, 32, 64);
                        if (c >= K) lo = mid; else hi = mid - 1;
                    }
                    if (tpart == 0) sm.tau_key[trow] = (int32_t)((uint32_t)lo << 16);
                }
#undef GPTR
                __syncthreads();
            }
        } else if (refresh) {
            // group maxima of this tile: group = (sb & 1) * 16 + 4 * kg + e
#pragma unroll
            for (int jb = 0; jb < 4; ++jb) {
                const int bl_ = wb * 64 + jb * 16 + l15;
#pragma unroll
                for (int p = 0; p < 2; ++p)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float m = fmaxf(fmaxf(acc[p][jb][e], acc[p + 2][jb][e]), fmaxf(acc[p + 4][jb][e], acc[p + 6][jb][e]));
                        sm.e32.slots32[ws][p * 16 + 4 * kg + e][bl_] = f2key(m);
                    }
            }
            __syncthreads();
            if (tid < HTB) sm.tau_key[tid] = INT32_MAX;
            __syncthreads();
            {
                const int row = tid % HTB;
                const int part = __builtin_amdgcn_readfirstlane(tid / HTB);
                constexpr int GPT = 16;  // groups per thread
                const int b = b0 + row;
                const bool share = b < B;
                const uint32_t boff = (uint32_t)b * 4u;
                int32_t m = INT32_MAX;
                int32_t old[GPT];
#pragma unroll
                for (int i = 0; i < GPT; ++i) {
                    old[i] = INT32_MIN;
                    if (share)
                        old[i] = __hip_atomic_load(
                            reinterpret_cast<int32_t*>(reinterpret_cast<char*>(a.gmax + (size_t)(part * GPT + i) * a.gmax_stride) + boff),
                            __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
#pragma unroll
                for (int i = 0; i < GPT; ++i) {
                    const int g = part * GPT + i;
                    const int32_t v = max(sm.e32.slots32[0][g][row], sm.e32.slots32[1][g][row]);
                    if (share && v > old[i])
                        atomicMax(reinterpret_cast<int32_t*>(reinterpret_cast<char*>(a.gmax + (size_t)g * a.gmax_stride) + boff), v);
                    m = min(m, max(v, old[i]));
                }
                atomicMin(&sm.tau_key[row], m);
            }
            __syncthreads();
        }
        if (NG == 32 && st == st_begin && a.top_k <= HTS / 4) {
            // first tile: the largest of three thresholds between the group bound and the row maximum that top_k values of
            // this tile reach (see encode_f16x3_kernel)
            if (tid < HTB) { sm.ref[0][tid] = INT32_MIN; sm.ref[1][tid] = 0; sm.ref[2][tid] = 0; sm.ref[3][tid] = 0; }
            __syncthreads();
#pragma unroll
            for (int jb = 0; jb < 4; ++jb) {
                float m = NEG_INF;
#pragma unroll
                for (int sb = 0; sb < 8; ++sb)
#pragma unroll
                    for (int e = 0; e < 4; ++e) m = fmaxf(m, acc[sb][jb][e]);
                atomicMax(&sm.ref[0][wb * 64 + jb * 16 + l15], f2key(m));
            }
            __syncthreads();
            float tg[4][3];
#pragma unroll
            for (int jb = 0; jb < 4; ++jb) {
                const int row = wb * 64 + jb * 16 + l15;
                const float t0 = key2f(sm.tau_key[row]), M = key2f(sm.ref[0][row]);
                const bool usable = t0 > -3.0e38f && M > t0;
                const float span = usable ? (M - t0) : 0.f;
                tg[jb][0] = t0 + 0.25f * span; tg[jb][1] = t0 + 0.40f * span; tg[jb][2] = t0 + 0.55f * span;
                int c0 = 0, c1 = 0, c2 = 0;
#pragma unroll
                for (int sb = 0; sb < 8; ++sb)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float v = acc[sb][jb][e];
                        c0 += (v >= tg[jb][0]) ? 1 : 0; c1 += (v >= tg[jb][1]) ? 1 : 0; c2 += (v >= tg[jb][2]) ? 1 : 0;
                    }
                if (usable) { atomicAdd(&sm.ref[1][row], c0); atomicAdd(&sm.ref[2][row], c1); atomicAdd(&sm.ref[3][row], c2); }
            }
            __syncthreads();
            if (ws == 0 && kg == 0) {  // one lane per row writes the refined bound back
#pragma unroll
                for (int jb = 0; jb < 4; ++jb) {
                    const int row = wb * 64 + jb * 16 + l15;
                    float best = key2f(sm.tau_key[row]);
                    if (sm.ref[1][row] >= a.top_k) best = fmaxf(best, tg[jb][0]);
                    if (sm.ref[2][row] >= a.top_k) best = fmaxf(best, tg[jb][1]);
                    if (sm.ref[3][row] >= a.top_k) best = fmaxf(best, tg[jb][2]);
                    sm.tau_key[row] = f2key(best);
                }
            }
            __syncthreads();
        }
        // Count, reserve, store.  The four lanes that share a row (kg = 0..3, 16 lanes apart) add their counts up with two
        // lane-row swaps (v_permlane16_swap / v_permlane32_swap: VALU, no LDS) and reserve the row's list space with ONE
        // atomic, issued by the kg = 0 lane as soon as that row block's count is known, so the first three round trips
        // overlap the counting of the following blocks.
        typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
        // About one value in 70 passes, so a per-value store block (compare, exec mask, branch, two stores)
        // finds some lane of the wave active at nearly every one of a lane's 128 positions and the wave pays for all of
        // them: 0.175 of this kernel's 1.18 ms (the first version of this kernel).  Here the compare pass leaves a 32-bit hit mask per lane and row block
        // instead of a count (same two instructions per value: v_cmp + v_addc "m = 2m + hit"); the lanes that have a hit
        // park the block's 32 values in LDS (the tile scratch, 8 KB per wave) and walk their own
        // mask: leading-zero count -> value index -> ds_read -> two stores.  The wave loops as often as its busiest lane has
        // hits (three or four times), not 32 times.
        uint32_t hit[4];
        int pos[4], rowtot[4], base[4];
#pragma unroll
        for (int jb = 0; jb < 4; ++jb) {
            const int b = b0 + wb * 64 + jb * 16 + l15;
            float tau = fmaxf(key2f(sm.tau_key[wb * 64 + jb * 16 + l15]), -3.0e38f);
            if (a.row_margin != nullptr) tau -= (b < B) ? a.row_margin[b] : 0.f;
            uint32_t m = 0;  // bit 31 - i: value i = 4 * sb + e reaches the bound
#pragma unroll
            for (int sb = 0; sb < 8; ++sb)
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    asm("v_cmp_ge_f32 vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(m) : "v"(acc[sb][jb][e]), "v"(tau) : "vcc");
            m = (b < B) ? m : 0u;
            const int n = __builtin_popcount(m);
            const u32x2 r = __builtin_amdgcn_permlane16_swap((unsigned)n, (unsigned)n, false, false);  // {even, odd row} of my row pair
            const unsigned pair = r[0] + r[1];
            const u32x2 q = __builtin_amdgcn_permlane32_swap(pair, pair, false, false);                // {rows 0+1, rows 2+3}
            rowtot[jb] = (int)(q[0] + q[1]);
            pos[jb] = (int)(((kg & 1) ? r[0] : 0u) + ((kg & 2) ? q[0] : 0u));  // lanes of the row before this one
            base[jb] = 0;
            if (kg == 0 && rowtot[jb] > 0) base[jb] = atomicAdd(&a.cand_cnt[b], rowtot[jb]);
            hit[jb] = m;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("" : "+v"(base[0]), "+v"(base[1]), "+v"(base[2]), "+v"(base[3]));
        // 8 KB per wave, [quad sb][lane][4]: a parked quad is one ds_write_b128 at an immediate offset, conflict-free
        float* const park = reinterpret_cast<float*>(&sm.e32.slots32[0][0][0]) + wid * 2048 + lane * 4;
        const int lat0 = s0 + ws * 128 + 4 * kg;
#pragma unroll
        for (int jb = 0; jb < 4; ++jb) {
            const u32x2 h = __builtin_amdgcn_permlane32_swap((unsigned)base[jb], (unsigned)base[jb], false, false);
            const u32x2 f = __builtin_amdgcn_permlane16_swap(h[0], h[0], false, false);
            const int row_base = (int)f[0];
            // a row whose list would overflow is not written at all (overflow_check sends the step down the dense route)
            uint32_t mm = (row_base + rowtot[jb] <= a.cand_cap) ? hit[jb] : 0u;
            if (__ballot(mm != 0u) == 0ull) continue;
            if (mm != 0u) {
#pragma unroll
                for (int sb = 0; sb < 8; ++sb) *reinterpret_cast<f32x4*>(park + 256 * sb) = acc[sb][jb];
            }
            uint32_t off = ((uint32_t)(b0 + wb * 64 + jb * 16 + l15) * (uint32_t)a.cand_stride + (uint32_t)(row_base + pos[jb])) * 4u;
            int i0 = 0;
            while (mm != 0u) {
                const int p = __builtin_clz(mm);
                const int i = i0 + p;
                mm = (mm << p) << 1;
                i0 = i + 1;
                const int c = i >> 2, e = i & 3;
                const float v = park[256 * c + e];
                *reinterpret_cast<float*>(reinterpret_cast<char*>(a.cand_val) + off) = v;
                *reinterpret_cast<int32_t*>(reinterpret_cast<char*>(a.cand_idx) + off) = lat0 + 16 * c + e;
                off += 4u;
            }
        }
        if (!prefetched && st + 1 < st_end) {  // (never: kept for symmetry with encode_f16x3_kernel)
            __syncthreads();
            stage_kstep(0, s0 + HTS, kmap(0));
            if (nks > 1) stage_kstep(1, s0 + HTS, kmap(1));
        }
    }
}

}  // namespace

hipError_t launch_encode_f16x3(const EncodeF16Args& a, int epi, hipStream_t stream) {
    const int n_bblocks = (a.n_rows + HTB - 1) / HTB;
    dim3 grid(n_bblocks * a.s_splits, (epi == EPI_DENSE && a.n_batches > 1) ? a.n_batches : 1), block(HTHREADS);
    const size_t smem = sizeof(HSmem);
    static bool attr_set = false;
    const bool use_m16 = a.mfma32 == 0;
    if (!attr_set) {

        const void* fns[16] = {reinterpret_cast<const void*>(&encode_m16_kernel<1>),
                              reinterpret_cast<const void*>(&encode_m16_kernel<2>),
                              reinterpret_cast<const void*>(&encode_m16_kernel<1, 64>),
                              reinterpret_cast<const void*>(&encode_m16_kernel<2, 64>),
                              reinterpret_cast<const void*>(&encode_f16x3_kernel<EPI_TOPK, 32, 0, true>),
                              reinterpret_cast<const void*>(&encode_f16x3_kernel<EPI_TOPK, 32, 1, true>),
                              reinterpret_cast<const void*>(&encode_f16x3_kernel<EPI_TOPK, 32, 2, true>),
                              reinterpret_cast<const void*>(&encode_f16x3_kernel<EPI_DENSE, 32, 0>),
                              reinterpret_cast<const void*>(&encode_f16x3_kernel<EPI_TOPK, 32, 0>),
                              reinterpret_cast<const void*>(&encode_f16x3_kernel<EPI_TOPK, 64, 0>),
                              reinterpret_cast<const void*>(&encode_f16x3_kernel<EPI_DENSE, 32, 1>),
                              reinterpret_cast<const void*>(&encode_f16x3_kernel<EPI_TOPK, 32, 1>),
                              reinterpret_cast<const void*>(&encode_f16x3_kernel<EPI_TOPK, 64, 1>),
                              reinterpret_cast<const void*>(&encode_f16x3_kernel<EPI_DENSE, 32, 2>),
                              reinterpret_cast<const void*>(&encode_f16x3_kernel<EPI_TOPK, 32, 2>),
                              reinterpret_cast<const void*>(&encode_f16x3_kernel<EPI_TOPK, 64, 2>)};
        for (const void* f : fns) {
            hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
            if (e != hipSuccess) return e;
        }
        attr_set = true;
    }
#define LAUNCH_ENC(E, G, N) hipLaunchKernelGGL((encode_f16x3_kernel<E, G, N>), grid, block, smem, stream, a)
#define LAUNCH_AR(E, G)                                   \
    do {                                                  \
        if (a.arith == 1) LAUNCH_ENC(E, G, 1);            \
        else if (a.arith == 2) LAUNCH_ENC(E, G, 2);       \
        else LAUNCH_ENC(E, G, 0);                         \
    } while (0)
    if (epi == EPI_DENSE) LAUNCH_AR(EPI_DENSE, 32);
    else if (a.ngroups <= 32 && a.heur_z != nullptr) {
        if (a.arith == 1) hipLaunchKernelGGL((encode_f16x3_kernel<EPI_TOPK, 32, 1, true>), grid, block, smem, stream, a);
        else if (a.arith == 2) hipLaunchKernelGGL((encode_f16x3_kernel<EPI_TOPK, 32, 2, true>), grid, block, smem, stream, a);
        else hipLaunchKernelGGL((encode_f16x3_kernel<EPI_TOPK, 32, 0, true>), grid, block, smem, stream, a);
    } else if (a.ngroups <= 32 && a.arith != 0 && use_m16) {
        // single-product modes: the 16x16x32 kernel (saev_debug_cfg.enc_mfma = 32 brings the 32x32x16 one back for A/B runs)
        if (a.arith == 1) hipLaunchKernelGGL((encode_m16_kernel<1>), grid, block, smem, stream, a);
        else hipLaunchKernelGGL((encode_m16_kernel<2>), grid, block, smem, stream, a);
    } else if (a.ngroups <= 32) LAUNCH_AR(EPI_TOPK, 32);
    else if (a.arith != 0 && use_m16) {
        if (a.arith == 1) hipLaunchKernelGGL((encode_m16_kernel<1, 64>), grid, block, smem, stream, a);
        else hipLaunchKernelGGL((encode_m16_kernel<2, 64>), grid, block, smem, stream, a);
    }
    else LAUNCH_AR(EPI_TOPK, 64);
#undef LAUNCH_AR
#undef LAUNCH_ENC
    return hipGetLastError();
}

int encode_f16x3_tile_rows() { return HTB; }
int encode_f16x3_tile_latents() { return HTS; }
