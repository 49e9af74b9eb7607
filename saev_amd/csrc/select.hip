// Exact per-row top-k selection (modeling.py:169-179: torch.topk(h, k, sorted=False) + scatter).
//
//   select_dense_kernel : one workgroup per row of a dense (n_rows x S) matrix; 4-pass MSB radix
//                         select (8 bits per pass, LDS histogram) on order-preserving uint keys,
//                         then an index-ordered compaction.  Optional latent mask (AuxK: dead only)
//                         and device-side k (AuxK: min(k_aux, n_dead)).
//   select_cand_kernel  : one wave per row over the short candidate list the fused encoder left
//                         behind; bitwise threshold search in registers, ties broken towards the
//                         smaller latent index, output ordered by latent index.
//
// Both emit exactly k (idx, val) pairs per row in ascending idx order (deterministic, and what the
// sparse backward's binary search relies on).  Ties on the k-th value: lowest indices win ("any k"
// is acceptable to the reference: tests/test_nn_activations.py:41-52).
#include "common.h"
#include "kernels.h"

namespace {

constexpr int SD_THREADS = 256;

__device__ __forceinline__ int block_excl_scan_256(int v, int* lds_wave_tot, int& total) {
    // exclusive scan of one int per thread over 256 threads (4 waves)
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    int incl = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        int n = __shfl_up(incl, o, 64);
        if (lane >= o) incl += n;
    }
    if (lane == 63) lds_wave_tot[w] = incl;
    __syncthreads();
    int base = 0, tot = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int t = lds_wave_tot[i];
        if (i < w) base += t;
        tot += t;
    }
    total = tot;
    __syncthreads();
    return base + incl - v;
}

__global__ __launch_bounds__(SD_THREADS) void select_dense_kernel(SelectDenseArgs a) {
    if (a.enable_flag != nullptr && (*a.enable_flag != 0) != (a.enable_when != 0)) return;
    __shared__ int hist[256];
    __shared__ int sh_digit, sh_need;
    __shared__ int wave_tot[4];

    const int row = blockIdx.x;
    const int S = a.S;
    int k = a.k_dev ? min(*a.k_dev, a.k) : a.k;
    if (a.k_dev && k <= 0) return;
    const float* h = a.h + (size_t)row * S;
    const int tid = threadIdx.x;
    const int S4 = S >> 2;

    // number of eligible elements (mask) bounds k
    // (without a mask, k <= S is enforced by the host)
    uint32_t prefix = 0, pmask = 0;
    int need = k;

    for (int pass = 0; pass < 4; ++pass) {
        const int shift = 24 - 8 * pass;
        hist[tid] = 0;
        __syncthreads();
        for (int q = tid; q < S4; q += SD_THREADS) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(h + 4 * q);
            int4 mk = {1, 1, 1, 1};
            if (a.mask) mk = *reinterpret_cast<const int4*>(a.mask + 4 * q);
            const int m[4] = {mk.x, mk.y, mk.z, mk.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const uint32_t key = f2ukey(v[e]);
                if (m[e] != 0 && (key & pmask) == prefix) atomicAdd(&hist[(key >> shift) & 255], 1);
            }
        }
        __syncthreads();
        if (tid < 64) {
            // lane l owns bins 4l..4l+3; suffix sums from the top
            const int c0 = hist[4 * tid], c1 = hist[4 * tid + 1], c2 = hist[4 * tid + 2], c3 = hist[4 * tid + 3];
            const int c = c0 + c1 + c2 + c3;
            int suf = c;  // inclusive suffix sum over lanes >= tid
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                int n = __shfl_down(suf, o, 64);
                if (tid + o < 64) suf += n;
            }
            const unsigned long long ge = __ballot(suf >= need);
            const int L = 63 - __builtin_clzll(ge | 1ull);  // highest lane whose suffix reaches `need`
            if (tid == L) {
                int running = suf - c;  // elements in bins above this lane's
                const int cs[4] = {c0, c1, c2, c3};
                int d = 4 * tid, nn = need;
                for (int j = 3; j >= 0; --j) {
                    if (running + cs[j] >= need) { d = 4 * tid + j; nn = need - running; break; }
                    running += cs[j];
                }
                sh_digit = d;
                sh_need = nn;
            }
        }
        __syncthreads();
        prefix |= (uint32_t)sh_digit << shift;
        pmask |= 0xFFu << shift;
        need = sh_need;
        __syncthreads();
    }
    // prefix == key of the k-th largest; take all keys > prefix and the first `need` equal keys.
    const uint32_t T = prefix;
    int base_gt = 0, base_eq = 0;
    int32_t* oi = a.idx_out + (size_t)row * a.out_stride;
    float* ov = a.val_out + (size_t)row * a.out_stride;
    for (int q0 = 0; q0 < S4; q0 += SD_THREADS) {
        const int q = q0 + tid;
        f32x4 v = {0, 0, 0, 0};
        int flg[4] = {0, 0, 0, 0}, feq[4] = {0, 0, 0, 0};
        if (q < S4) {
            v = *reinterpret_cast<const f32x4*>(h + 4 * q);
            int4 mk = {1, 1, 1, 1};
            if (a.mask) mk = *reinterpret_cast<const int4*>(a.mask + 4 * q);
            const int m[4] = {mk.x, mk.y, mk.z, mk.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const uint32_t key = f2ukey(v[e]);
                flg[e] = (m[e] != 0 && key > T);
                feq[e] = (m[e] != 0 && key == T);
            }
        }
        const int ngt = flg[0] + flg[1] + flg[2] + flg[3];
        const int neq = feq[0] + feq[1] + feq[2] + feq[3];
        int tot;
        const int ex = block_excl_scan_256(ngt | (neq << 16), wave_tot, tot);
        int gt_before = base_gt + (ex & 0xFFFF);
        int eq_before = base_eq + (ex >> 16);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if (flg[e]) {
                const int pos = gt_before + min(eq_before, need);
                oi[pos] = 4 * q + e;
                ov[pos] = v[e];
                ++gt_before;
            } else if (feq[e]) {
                if (eq_before < need) {
                    const int pos = gt_before + eq_before;
                    oi[pos] = 4 * q + e;
                    ov[pos] = v[e];
                }
                ++eq_before;
            }
        }
        base_gt += tot & 0xFFFF;
        base_eq += tot >> 16;
    }
}

// ---------------------------------------------------------------------------------------------

// lane l ends up with the wave-wide sum of p[(l >> 3) & 7] (see sparse.hip wave_reduce_scatter)
__device__ __forceinline__ float wave_reduce_scatter8(float (&p)[8], int lane) {
    int bit = 32;
#pragma unroll
    for (int h = 4; h >= 1; h >>= 1, bit >>= 1) {
        const bool up = (lane & bit) != 0;
#pragma unroll
        for (int i = 0; i < h; ++i) {
            const float keep = up ? p[i + h] : p[i];
            const float send = up ? p[i] : p[i + h];
            p[i] = keep + __shfl_xor(send, bit, 64);
        }
    }
    float r = p[0];
    for (; bit >= 1; bit >>= 1) r += __shfl_xor(r, bit, 64);
    return r;
}

// 64 lanes each holding one (index, value) pair or (INT_MAX, 0): sort by index ascending and write the first k
__device__ __forceinline__ void sort_by_idx_and_store(const SelectCandArgs& a, int row, int k, int32_t my_idx, float my_val,
                                                      int lane) {
    // bitonic sort of 64 lanes by idx ascending (unused lanes carry INT_MAX and end up last)
#pragma unroll
    for (int size = 2; size <= 64; size <<= 1) {
#pragma unroll
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            const int32_t o_idx = __shfl_xor(my_idx, stride, 64);
            const float o_val = __shfl_xor(my_val, stride, 64);
            const bool up = ((lane & size) == 0);          // ascending block
            const bool lower = ((lane & stride) == 0);     // this lane keeps the smaller one (if up)
            const bool take_other = (lower == up) ? (o_idx < my_idx) : (o_idx > my_idx);
            if (take_other) { my_idx = o_idx; my_val = o_val; }
        }
    }
    if (lane < a.k) {
        const bool ok = lane < k;
        a.idx_out[(size_t)row * a.out_stride + lane] = ok ? my_idx : -1;
        a.val_out[(size_t)row * a.out_stride + lane] = ok ? my_val : 0.f;
    }
}

template <int EPL>  // elements per lane: handles lists of up to 64*EPL candidates
__device__ __forceinline__ void select_cand_row(const SelectCandArgs& a, int row, int n, int32_t (&s_idx)[4][64],
                                                float (&s_val)[4][64], const float* cv, const int32_t* ci) {
    const int lane = threadIdx.x & 63;
    const int w = threadIdx.x >> 6;
    const int k = min(a.k, n);

    uint32_t key[EPL];
    int32_t idx[EPL];
#pragma unroll
    for (int e = 0; e < EPL; ++e) {
        const int p = e * 64 + lane;
        if (p < n) { key[e] = f2ukey(cv[p]); idx[e] = ci[p]; }
        else { key[e] = 0u; idx[e] = 0x7fffffff; }  // key 0 sorts below every real float key
    }
    // largest T with count(key >= T) >= k
    auto kth_largest = [&]() {
        uint32_t t = 0;
        for (int bit = 31; bit >= 0; --bit) {
            const uint32_t trial = t | (1u << bit);
            // wave-wide count through ballots: scalar popcounts, no cross-lane shuffle chain per bit
            int c = 0;
#pragma unroll
            for (int e = 0; e < EPL; ++e) c += __popcll(__ballot(key[e] >= trial));
            if (c == k) {
                // exactly k keys reach the trial value: they are the top k, and the k-th largest is the smallest of them --
                // no need to resolve the remaining bits (on real lists this happens about half way down)
                uint32_t m = 0xffffffffu;
#pragma unroll
                for (int e = 0; e < EPL; ++e) m = min(m, key[e] >= trial ? key[e] : 0xffffffffu);
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) m = min(m, (uint32_t)__shfl_xor((int)m, o, 64));
                return m;
            }
            if (c > k) t = trial;
        }
        return t;
    };
    uint32_t T = kth_largest();
    if (a.tau_max != nullptr && lane == 0 && (n < a.k || ukey2f(T) < key2f(a.tau_max[row]))) *a.invalid = 1;
    if (a.row_margin != nullptr) {
        // The values are approximate (first pass of SAEV_ENCODER_F16R).  Every member of the exact top-k has an
        // approximate value >= (k-th largest approximate value) - row_margin: emit those survivors; refine_exact_kernel
        // recomputes them in fp32 and a second run of this kernel (exact values, no margin) makes the final cut.
        const uint32_t key_lo = f2ukey(ukey2f(T) - a.row_margin[row]);
        int32_t* so = a.surv_idx + (size_t)row * REFINE_CAP;
        int base = 0;
        // (surv_rng: the survivors grouped by latent range -- range r ends at surv_rng[row][r] -- for refine_slices_kernel, whose
        // passes each cover one range of latents; one range = the plain list)
        const int n_rng = a.surv_rng != nullptr ? a.n_ranges : 1;
        for (int r = 0; r < n_rng; ++r) {
            const int32_t lo = r * a.lat_range, hi = (r == n_rng - 1) ? 0x7ffffffe : lo + a.lat_range;
#pragma unroll
            for (int e = 0; e < EPL; ++e) {
                const bool sv = key[e] >= key_lo && idx[e] >= lo && idx[e] <= hi - (r == n_rng - 1 ? 0 : 1);
                const unsigned long long m = __ballot(sv);
                const int pos = base + __popcll(m & ((1ull << lane) - 1ull));
                if (sv && pos < REFINE_CAP) so[pos] = idx[e];
                base += __popcll(m);
            }
            if (a.surv_rng != nullptr && lane == 0) a.surv_rng[(size_t)row * RS_MAX_RANGES + r] = min(base, REFINE_CAP);
        }
        if (lane == 0) {
            a.surv_cnt[row] = min(base, REFINE_CAP);
            if (base > REFINE_CAP) *a.refine_overflow = 1;  // the caller's dense route redoes the launch exactly
        }
        return;
    }
    int cgt = 0, ceq = 0;
#pragma unroll
    for (int e = 0; e < EPL; ++e) { cgt += __popcll(__ballot(key[e] > T)); ceq += __popcll(__ballot(key[e] == T)); }
    const int need = k - cgt;  // ties to keep (>= 1 when k > 0)
    int32_t idx_cut = 0x7fffffff;  // keep ties with idx <= idx_cut
    if (ceq > need) {
        // largest X with count(tie && idx < X) < need; latent indices are distinct, so
        // count(tie && idx <= X) == need exactly.
        int32_t X = 0;
        for (int bit = 30; bit >= 0; --bit) {
            const int32_t trial = X | (1 << bit);
            int c = 0;
#pragma unroll
            for (int e = 0; e < EPL; ++e) c += __popcll(__ballot(key[e] == T && idx[e] < trial));
            if (c < need) X = trial;
        }
        idx_cut = X;
    }
    // compact the k winners into one element per lane (k <= 64)
    s_idx[w][lane] = 0x7fffffff;
    s_val[w][lane] = 0.f;
    int base = 0;
#pragma unroll
    for (int e = 0; e < EPL; ++e) {
        const bool sel = (key[e] > T) || (key[e] == T && idx[e] <= idx_cut);
        const unsigned long long m = __ballot(sel);
        const int pos = base + __popcll(m & ((1ull << lane) - 1ull));
        if (sel && pos < 64) { s_idx[w][pos] = idx[e]; s_val[w][pos] = ukey2f(key[e]); }
        base += __popcll(m);
    }
    sort_by_idx_and_store(a, row, k, s_idx[w][lane], s_val[w][lane], lane);
}

// ---- lists too long for registers (more than 2 048 candidates: a few rows of a launch at most) ---------------------------
// The same selection with the list left where it is (L2): every step of the threshold search is a pass over it.  Keeping
// 4 096 entries per row in registers for these rows cost every row of the kernel its occupancy (151 VGPRs, three waves
// per SIMD); this path is ~10 us for the wave that takes it.  (Lists of 1 025 .. 2 048 entries are common -- the mean is
// ~900 at configs[1] -- and stay in registers: streaming them made the survivor select 100 us instead of 70.)
__device__ __forceinline__ int stream_count(const float* cv, const int32_t* ci, int n, uint32_t lo_key, int mode, uint32_t T,
                                            int32_t idx_lt, int lane) {
    // mode 0: key >= lo_key;  1: key > T;  2: key == T;  3: key == T && idx < idx_lt
    int c = 0;
    for (int p0 = 0; p0 < n; p0 += 64) {
        const int p = p0 + lane;
        bool hit = false;
        if (p < n) {
            const uint32_t key = f2ukey(cv[p]);
            hit = mode == 0 ? key >= lo_key : mode == 1 ? key > T : mode == 2 ? key == T : (key == T && ci[p] < idx_lt);
        }
        c += __popcll(__ballot(hit));
    }
    return c;
}
__device__ __forceinline__ uint32_t stream_kth_largest(const float* cv, const int32_t* ci, int n, int k, int lane) {
    uint32_t t = 0;
    for (int bit = 31; bit >= 0; --bit) {
        const uint32_t trial = t | (1u << bit);
        const int c = stream_count(cv, ci, n, trial, 0, 0u, 0, lane);
        if (c == k) {  // the k-th largest is the smallest key that reaches the trial value
            uint32_t m = 0xffffffffu;
            for (int p = lane; p < n; p += 64) { const uint32_t key = f2ukey(cv[p]); if (key >= trial) m = min(m, key); }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) m = min(m, (uint32_t)__shfl_xor((int)m, o, 64));
            return m;
        }
        if (c > k) t = trial;
    }
    return t;
}
// survivors of the approximate cut -> out[0..), returns their number (may exceed REFINE_CAP: only the first REFINE_CAP are stored)
__device__ __forceinline__ int stream_survivors(const SelectCandArgs& a, int row, int n, const float* cv, const int32_t* ci,
                                                int32_t* out, uint32_t* T_out) {
    const int lane = threadIdx.x & 63;
    const uint32_t T = stream_kth_largest(cv, ci, n, min(a.k, n), lane);
    *T_out = T;
    const uint32_t key_lo = f2ukey(ukey2f(T) - a.row_margin[row]);
    int base = 0;
    const int n_rng = a.surv_rng != nullptr ? a.n_ranges : 1;  // (grouped by latent range like select_cand_row's)
    for (int r = 0; r < n_rng; ++r) {
        const int32_t lo = r * a.lat_range, hi = (r == n_rng - 1) ? 0x7ffffffe : lo + a.lat_range - 1;
        for (int p0 = 0; p0 < n; p0 += 64) {
            const int p = p0 + lane;
            bool sv = p < n && f2ukey(cv[p]) >= key_lo;
            if (sv && n_rng > 1) { const int32_t i = ci[p]; sv = i >= lo && i <= hi; }
            const unsigned long long m = __ballot(sv);
            const int pos = base + __popcll(m & ((1ull << lane) - 1ull));
            if (sv && pos < REFINE_CAP) out[pos] = ci[p];
            base += __popcll(m);
        }
        if (a.surv_rng != nullptr && lane == 0) a.surv_rng[(size_t)row * RS_MAX_RANGES + r] = min(base, REFINE_CAP);
    }
    return base;
}
__device__ __forceinline__ void select_cand_row_stream(const SelectCandArgs& a, int row, int n, int32_t (&s_idx)[4][64],
                                                       float (&s_val)[4][64], const float* cv, const int32_t* ci) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int k = min(a.k, n);
    if (a.row_margin != nullptr) {
        uint32_t T;
        const int base = stream_survivors(a, row, n, cv, ci, a.surv_idx + (size_t)row * REFINE_CAP, &T);
        if (a.tau_max != nullptr && lane == 0 && (n < a.k || ukey2f(T) < key2f(a.tau_max[row]))) *a.invalid = 1;
        if (lane == 0) {
            a.surv_cnt[row] = min(base, REFINE_CAP);
            if (base > REFINE_CAP) *a.refine_overflow = 1;
        }
        return;
    }
    const uint32_t T = stream_kth_largest(cv, ci, n, k, lane);
    if (a.tau_max != nullptr && lane == 0 && (n < a.k || ukey2f(T) < key2f(a.tau_max[row]))) *a.invalid = 1;
    const int cgt = stream_count(cv, ci, n, 0u, 1, T, 0, lane), ceq = stream_count(cv, ci, n, 0u, 2, T, 0, lane);
    const int need = k - cgt;
    int32_t idx_cut = 0x7fffffff;
    if (ceq > need) {
        int32_t X = 0;
        for (int bit = 30; bit >= 0; --bit) {
            const int32_t trial = X | (1 << bit);
            if (stream_count(cv, ci, n, 0u, 3, T, trial, lane) < need) X = trial;
        }
        idx_cut = X;
    }
    s_idx[w][lane] = 0x7fffffff;
    s_val[w][lane] = 0.f;
    int base = 0;
    for (int p0 = 0; p0 < n; p0 += 64) {
        const int p = p0 + lane;
        uint32_t key = 0u;
        int32_t idx = 0x7fffffff;
        if (p < n) { key = f2ukey(cv[p]); idx = ci[p]; }
        const bool sel = p < n && ((key > T) || (key == T && idx <= idx_cut));
        const unsigned long long m = __ballot(sel);
        const int pos = base + __popcll(m & ((1ull << lane) - 1ull));
        if (sel && pos < 64) { s_idx[w][pos] = idx; s_val[w][pos] = ukey2f(key); }
        base += __popcll(m);
    }
    sort_by_idx_and_store(a, row, k, s_idx[w][lane], s_val[w][lane], lane);
}

// Lists of at most 64 entries (the second, exact pass of the f16r encoder: ~45 survivors per row): one entry per lane
// and a 64-lane bitonic sort by (value descending, index ascending) instead of the 32-step bit search -- the first k lanes
// are the winners, ties at the cut resolved towards the smaller index exactly as in select_cand_row.
__device__ __forceinline__ void select_small_row(const SelectCandArgs& a, int row, int n, const float* cv, const int32_t* ci) {
    const int lane = threadIdx.x & 63;
    const int k = min(a.k, n);
    uint32_t key = 0u;            // sorts below every real float key
    int32_t idx = 0x7fffffff;
    if (lane < n) {
        key = f2ukey(cv[lane]);
        idx = ci[lane];
    }
#pragma unroll
    for (int size = 2; size <= 64; size <<= 1) {
#pragma unroll
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            const uint32_t o_key = __shfl_xor(key, stride, 64);
            const int32_t o_idx = __shfl_xor(idx, stride, 64);
            const bool other_first = (o_key > key) || (o_key == key && o_idx < idx);  // other sorts before mine
            const bool up = ((lane & size) == 0);
            const bool lower = ((lane & stride) == 0);
            const bool take_other = (lower == up) ? other_first : !other_first;
            if (take_other) { key = o_key; idx = o_idx; }
        }
    }
    if (a.tau_max != nullptr) {
        const float kth = ukey2f(__shfl(key, max(k - 1, 0), 64));
        if (lane == 0 && (n < a.k || kth < key2f(a.tau_max[row]))) *a.invalid = 1;
    }
    const bool win = lane < k;
    sort_by_idx_and_store(a, row, k, win ? idx : 0x7fffffff, win ? ukey2f(key) : 0.f, lane);
}

// one wave per row; the register footprint of the search is chosen from the row's list length.  BIG: lists of up to 4 096
// entries in registers as well (151 VGPRs, three waves per SIMD) -- the k > 32 shapes, whose lists average ~2 000 entries
// (configs[3]: streaming everything above 2 048 made this kernel 0.75 ms there); k <= 32 launches the lean variant, where such
// lists are a handful of rows and go through the streaming path.
template <bool BIG>
__global__ __launch_bounds__(256) void select_cand_kernel(SelectCandArgs a) {
    if (a.enable_flag != nullptr && (*a.enable_flag != 0) != (a.enable_when != 0)) return;
    __shared__ int32_t s_idx[4][64];
    __shared__ float s_val[4][64];
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= a.n_rows) return;
    const int cnt = a.cand_cnt[row];
    // a row whose list did not fit sends the whole launch down the exact dense route (the kernels in between run on its
    // truncated list and are overwritten there); this check used to be a launch of its own between the encoder and the
    // select.  (The list statistics come from stats_reduce: a per-row atomic here cost 30 us.)
    if (a.ovf != nullptr && (threadIdx.x & 63) == 0 && cnt > a.cand_cap) atomicOr(&a.ovf[0], 1);
    // (so does a row without a finite margin: an inf / NaN element in the batch -- it poisons the column mean the first pass is
    // centred on, every approximate value is NaN and no list holds anything; the exact route treats such input as torch does)
    if (a.ovf != nullptr && a.row_margin != nullptr && (threadIdx.x & 63) == 0 && !(a.row_margin[row] < 3.0e38f)) atomicOr(&a.ovf[0], 1);
    const int n = min(cnt, a.cand_cap);  // wave-uniform
    const float* cv = a.cand_val + (size_t)row * a.cand_stride;
    const int32_t* ci = a.cand_idx + (size_t)row * a.cand_stride;
    __shared__ float l_sum[4][REFINE_CAP];
    if (a.sum_part != nullptr) {
        // the exact values of the survivors from their per-slice shares (refine_slices_kernel), eight loads in flight, added in
        // slice order; they stay in LDS for the cut below (LDS operations of one wave execute in order)
        float* const lv = l_sum[threadIdx.x >> 6];
        for (int j = threadIdx.x & 63; j < n; j += 64) {
            const size_t o = (size_t)row * a.cand_stride + j;
            float sv = 0.f;
            int c = 0;
            for (; c + 8 <= a.sum_n; c += 8) {
                float v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = a.sum_part[(size_t)(c + u) * a.sum_plane + o];
#pragma unroll
                for (int u = 0; u < 8; ++u) sv += v[u];
            }
            for (; c < a.sum_n; ++c) sv += a.sum_part[(size_t)c * a.sum_plane + o];
            lv[j] = sv + a.sum_bias[ci[j]];
        }
        __asm__ volatile("" ::: "memory");
        cv = lv;
    }
    if (n <= 64 && a.row_margin == nullptr) select_small_row(a, row, n, cv, ci);
    else if (n <= 512) select_cand_row<8>(a, row, n, s_idx, s_val, cv, ci);
    else if (n <= 1024) select_cand_row<16>(a, row, n, s_idx, s_val, cv, ci);
    else if (n <= 2048) select_cand_row<32>(a, row, n, s_idx, s_val, cv, ci);
    else if constexpr (BIG) select_cand_row<64>(a, row, n, s_idx, s_val, cv, ci);
    else select_cand_row_stream(a, row, n, s_idx, s_val, cv, ci);
}

// ---- the f16r exactness chain in ONE launch -------------------------------------------------------------------------
// select_cand_kernel (survivors) -> refine_exact_kernel -> select_cand_kernel (final cut) as one kernel, one wave per row:
// the survivor list and its exact values live in LDS instead of making two global round trips, the two relaunches are gone,
// and the bit search of one wave (issue-bound) overlaps the row gathers of its neighbours (memory-bound) on the same CU.
// Same arithmetic in the same order as the three kernels: bit-identical codes.

// survivors of the approximate cut of one row -> lds_idx[0..ns); returns ns (wave-uniform), -1 when more than REFINE_CAP
template <int EPL>
__device__ __forceinline__ int survivors_to_lds(const SelectCandArgs& a, int row, int n, const float* cv, const int32_t* ci,
                                                int32_t* lds_idx) {
    const int lane = threadIdx.x & 63;
    const int k = min(a.k, n);
    uint32_t key[EPL];
    int32_t idx[EPL];
#pragma unroll
    for (int e = 0; e < EPL; ++e) {
        const int p = e * 64 + lane;
        if (p < n) { key[e] = f2ukey(cv[p]); idx[e] = ci[p]; }
        else { key[e] = 0u; idx[e] = 0x7fffffff; }
    }
    uint32_t t = 0;
    bool done = false;
    for (int bit = 31; bit >= 0 && !done; --bit) {
        const uint32_t trial = t | (1u << bit);
        int c = 0;
#pragma unroll
        for (int e = 0; e < EPL; ++e) c += __popcll(__ballot(key[e] >= trial));
        if (c == k) {  // exactly k keys reach the trial value: the k-th largest is the smallest of them
            uint32_t m = 0xffffffffu;
#pragma unroll
            for (int e = 0; e < EPL; ++e) m = min(m, key[e] >= trial ? key[e] : 0xffffffffu);
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) m = min(m, (uint32_t)__shfl_xor((int)m, o, 64));
            t = m;
            done = true;
        } else if (c > k) {
            t = trial;
        }
    }
    const uint32_t key_lo = f2ukey(ukey2f(t) - a.row_margin[row]);
    int base = 0;
#pragma unroll
    for (int e = 0; e < EPL; ++e) {
        const bool sv = key[e] >= key_lo && idx[e] != 0x7fffffff;
        const unsigned long long m = __ballot(sv);
        const int pos = base + __popcll(m & ((1ull << lane) - 1ull));
        if (sv && pos < REFINE_CAP) lds_idx[pos] = idx[e];
        base += __popcll(m);
    }
    return base > REFINE_CAP ? -1 : base;
}

template <int NV>
__global__ __launch_bounds__(256) void select_refine_kernel(SelectCandArgs a) {
    if (a.enable_flag != nullptr && (*a.enable_flag != 0) != (a.enable_when != 0)) return;
    __shared__ int32_t l_idx[4][REFINE_CAP];
    __shared__ float l_val[4][REFINE_CAP];
    __shared__ int32_t s_idx[4][64];
    __shared__ float s_val[4][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int row = blockIdx.x * 4 + w;
    if (row >= a.n_rows) return;
    const int cnt = a.cand_cnt[row];
    if (a.ovf != nullptr && lane == 0 && (cnt > a.cand_cap || !(a.row_margin[row] < 3.0e38f))) atomicOr(&a.ovf[0], 1);
    const int n = min(cnt, a.cand_cap);
    const float* cv = a.cand_val + (size_t)row * a.cand_stride;
    const int32_t* ci = a.cand_idx + (size_t)row * a.cand_stride;
    int ns;
    if (n <= 512) ns = survivors_to_lds<8>(a, row, n, cv, ci, l_idx[w]);
    else if (n <= 1024) ns = survivors_to_lds<16>(a, row, n, cv, ci, l_idx[w]);
    else if (n <= 2048) ns = survivors_to_lds<32>(a, row, n, cv, ci, l_idx[w]);
    else {
        uint32_t T;
        ns = stream_survivors(a, row, n, cv, ci, l_idx[w], &T);
        if (ns > REFINE_CAP) ns = -1;
    }
    if (ns < 0) {  // the caller's dense route redoes the launch exactly
        if (lane == 0) *a.refine_overflow = 1;
        ns = REFINE_CAP;
    }
    // exact pre-activations of the survivors (refine_exact_kernel's loop, lists in LDS)
    const int D4 = a.D >> 2;
    f32x4 xv[NV];
    const f32x4* xr = reinterpret_cast<const f32x4*>(a.x + (size_t)row * a.D);
#pragma unroll
    for (int q = 0; q < NV; ++q) xv[q] = (lane + 64 * q < D4) ? xr[lane + 64 * q] : f32x4{0.f, 0.f, 0.f, 0.f};
    for (int j0 = 0; j0 < ns; j0 += 8) {
        const int32_t my = l_idx[w][min(j0 + (lane & 7), ns - 1)];
        float p[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const int32_t li = __shfl(my, t, 64);
            const f32x4* wr = reinterpret_cast<const f32x4*>(a.W_encT + (size_t)li * a.D);
            float acc = 0.f;
#pragma unroll
            for (int q = 0; q < NV; ++q) {
                if (lane + 64 * q < D4) {
                    const f32x4 wv = wr[lane + 64 * q];
                    acc += xv[q][0] * wv[0] + xv[q][1] * wv[1] + xv[q][2] * wv[2] + xv[q][3] * wv[3];
                }
            }
            p[t] = acc;
        }
        const float r = wave_reduce_scatter8(p, lane);  // lane l holds the sum of slot (l >> 3) & 7
        if ((lane & 7) == 0) {
            const int t = lane >> 3;
            if (j0 + t < ns) l_val[w][j0 + t] = r + a.b_enc[l_idx[w][j0 + t]];
        }
    }
    // the final cut on the exact values (select_cand_kernel without a margin, lists in LDS)
    SelectCandArgs b = a;
    b.row_margin = nullptr; b.tau_max = nullptr;
    if (ns <= 64) select_small_row(b, row, ns, l_val[w], l_idx[w]);
    else select_cand_row<8>(b, row, ns, s_idx, s_val, l_val[w], l_idx[w]);
}

__global__ void init_i32_kernel(int32_t* p, int32_t v, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

// per-launch state of the fused encoder in one pass: candidate counters to 0, shared group maxima / largest predicted
// bounds to "-inf"; optionally predicated on a device flag
__global__ void encoder_init_kernel(int32_t* cand_cnt, int n_rows, int32_t* gmax, int n_gmax, int32_t* tau_max,
                                    const int32_t* enable_flag, int enable_when) {
    if (enable_flag != nullptr && (*enable_flag != 0) != (enable_when != 0)) return;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_rows) cand_cnt[i] = 0;
    if (i < n_gmax) gmax[i] = INT32_MIN;
    if (tau_max != nullptr && i < n_rows) tau_max[i] = INT32_MIN;
}
// per-step scalars in one pass: the stats block, max|x| and the force-dense flag
__global__ void step_zero_kernel(saev_step_stats* stats, float* upper, int32_t* flag0) {
    if (threadIdx.x == 0) {
        *stats = saev_step_stats{};
        *upper = 0.f;
        *flag0 = 0;
    }
}

// exact pre-activations of the survivors: h = <x_row, W_enc^T[idx]> + b_enc[idx] in fp32, one wave per row, eight
// survivors at a time with all their row loads in flight; the x row stays in registers.
template <int NV>
__global__ __launch_bounds__(256) void refine_exact_kernel(SelectCandArgs a) {
    if (a.enable_flag != nullptr && (*a.enable_flag != 0) != (a.enable_when != 0)) return;
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= a.n_rows) return;
    const int ns = a.surv_cnt[row];
    const int D4 = a.D >> 2;
    f32x4 xv[NV];
    const f32x4* xr = reinterpret_cast<const f32x4*>(a.x + (size_t)row * a.D);
#pragma unroll
    for (int n = 0; n < NV; ++n) xv[n] = (lane + 64 * n < D4) ? xr[lane + 64 * n] : f32x4{0.f, 0.f, 0.f, 0.f};
    const int32_t* si = a.surv_idx + (size_t)row * REFINE_CAP;
    float* sv = a.surv_val + (size_t)row * REFINE_CAP;
    for (int j0 = 0; j0 < ns; j0 += 8) {
        const int32_t my = si[min(j0 + (lane & 7), ns - 1)];
        float p[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const int32_t li = __shfl(my, t, 64);
            const f32x4* wr = reinterpret_cast<const f32x4*>(a.W_encT + (size_t)li * a.D);
            float acc = 0.f;
#pragma unroll
            for (int n = 0; n < NV; ++n) {
                if (lane + 64 * n < D4) {
                    const f32x4 wv = wr[lane + 64 * n];
                    acc += xv[n][0] * wv[0] + xv[n][1] * wv[1] + xv[n][2] * wv[2] + xv[n][3] * wv[3];
                }
            }
            p[t] = acc;
        }
        const float r = wave_reduce_scatter8(p, lane);  // lane l holds the sum of slot (l >> 3) & 7
        if ((lane & 7) == 0) {
            const int t = lane >> 3;
            if (j0 + t < ns) sv[j0 + t] = r + a.b_enc[si[j0 + t]];
        }
    }
}

// ---- the same exact values from 32-column slices that an XCD's L2 holds (kernels.h: RefineSlicesArgs) ---------------------------
// refine_exact_kernel gathers ~45 whole 4 KB rows of W_enc^T per activation row out of a 134 MB matrix: L2 hits while a young
// dictionary uses a few thousand latents, fabric traffic once usage has spread (0.23 -> 0.36 ms over the first thousand steps
// of the benchmark, 0.39 on isotropic data).  Tiled the other way the working set is bounded by construction: a 32-column
// slice of W_enc^T for a range of 16 384 latents is 2 MB.  Workgroup b runs on XCD b mod 8 and the workgroups of an XCD walk
// (slice, latent range) combinations one after the other, so at any time an XCD gathers from ONE such tile; x comes from the
// slice-major copy split_f16r leaves ([slice][row][32]: the rows of a slice are consecutive 128-byte lines) and W_enc^T is
// written slice-major by the same pass ([slice][latent][32]).  An eight-lane group owns a row: its x slice sits in one float4 per
// lane, a survivor is one 128-byte gather, four fmas and -- eight survivors at a time -- a reduce-scatter over the group, so
// lane l stores survivor l's share of the dot product (one 32-byte store per group).  Survivors outside the latent range
// of the pass cost an out-of-bounds buffer load (no memory access).  refine_sum_kernel adds the D / 32 shares in slice
// order and the bias: the same fp32 products as the row kernel, summed in another (fixed) order.
typedef int i32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4 rs_buf_load16(__amdgpu_buffer_rsrc_t r, uint32_t voff) {
    const i32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, voff, 0, 0);
    return f32x4{__int_as_float(v[0]), __int_as_float(v[1]), __int_as_float(v[2]), __int_as_float(v[3])};
}
// reduce-scatter over the eight lanes of a group: lane l (0..7) ends up with the group-wide sum of p[l]
__device__ __forceinline__ float group8_reduce_scatter(float (&p)[8], int li) {
    {
        const bool up = (li & 4) != 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float keep = up ? p[i + 4] : p[i], send = up ? p[i] : p[i + 4];
            p[i] = keep + __shfl_xor(send, 4, 64);
        }
    }
    {
        const bool up = (li & 2) != 0;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const float keep = up ? p[i + 2] : p[i], send = up ? p[i] : p[i + 2];
            p[i] = keep + __shfl_xor(send, 2, 64);
        }
    }
    const bool up = (li & 1) != 0;
    const float keep = up ? p[1] : p[0], send = up ? p[0] : p[1];
    return keep + __shfl_xor(send, 1, 64);
}
// An eight-lane group works through RS_ROWS activation rows; the eight groups of a wave run one instruction stream, so a group is
// busy ceil(survivors / 8) trips per row.  Rows differ (22 +- 5 survivors per latent range at configs[1]): with every group on its
// t-th row at the same time the wave waits for the longest list of each octet of rows (3.8 trips against 2.8 on average).  Here a
// group moves on to its next row as soon as its list is done -- the rows' list bounds and x slices are fetched up front, the
// advance is a few selects -- so the wave runs as long as its busiest GROUP over all its rows: 283 -> 259 us at RS_ROWS = 8;
// sixteen gathers per trip instead of eight and one 32 768-latent range per slice instead of two: -> 233 us (24 / 32 per trip: 248 / 264)
// (4: 281; the same with the bounds in LDS, a run-time advance loop and the next trip's indices requested behind the gathers:
// 289 at 8 rows, 327 at 16 -- tools/experiments/r4_kernel_ab.sh).
#ifndef RS_BATCHES
#define RS_BATCHES 2
#endif
template <int RS_ROWS>
__global__ __launch_bounds__(256) void refine_slices_kernel(RefineSlicesArgs a, int wg_per_combo) {
    if (a.enable_flag != nullptr && (*a.enable_flag != 0) != (a.enable_when != 0)) return;
    const int lane = threadIdx.x & 63, gi = lane >> 3, li = lane & 7;
    const int xcd = blockIdx.x & 7, q = blockIdx.x >> 3;
    const int combo = q / wg_per_combo;
    const int slice = xcd + 8 * (combo / a.n_ranges), range = combo % a.n_ranges;
    if (slice * RS_SLICE >= a.D) return;
    const int wgi = q % wg_per_combo;
    const int lat_lo = range * a.lat_range, lat_hi = min(a.S, lat_lo + a.lat_range);
    const int sel = (lane & 56) << 2;  // byte address of the group's lane 0 for ds_bpermute
    const __amdgpu_buffer_rsrc_t wres = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(a.WeS) + (size_t)slice * a.S * RS_SLICE, 0, (uint32_t)a.S * 128u, 0x00020000);
    const f32x4* const xs = reinterpret_cast<const f32x4*>(a.xS + (size_t)slice * a.n_rows * RS_SLICE);
    float* const part = a.part + (size_t)slice * a.n_rows * REFINE_CAP;
    const uint32_t li16 = (uint32_t)li * 16u;
    // this pass's survivors of the group's rows: each row's sub-list of the latent range (select_cand_kernel groups them; one range: all)
    const int row0 = (wgi * RS_ROWS * 4 + (int)(threadIdx.x >> 6)) * 8 + gi;  // row of trip t: row0 + 32 t
    int jb[RS_ROWS], je[RS_ROWS];
    f32x4 xr[RS_ROWS];
#pragma unroll
    for (int t = 0; t < RS_ROWS; ++t) {
        const int row = row0 + 32 * t;
        const bool rok = row < a.n_rows;
        je[t] = rok ? a.surv_rng[(size_t)row * RS_MAX_RANGES + range] : 0;
        jb[t] = (rok && range > 0) ? a.surv_rng[(size_t)row * RS_MAX_RANGES + range - 1] : 0;
        xr[t] = rok ? xs[(size_t)row * 8 + li] : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    int t = 0, j0 = jb[0], j_end = je[0];
    f32x4 x4 = xr[0];
    auto advance = [&]() {  // on to the next row that has survivors in this range (each step only fires when the one before it did)
#pragma unroll
        for (int s = 1; s < RS_ROWS; ++s) {
            const bool adv = j0 >= j_end && t == s - 1;
            t = adv ? s : t;
            j0 = adv ? jb[s] : j0;
            j_end = adv ? je[s] : j_end;
#pragma unroll
            for (int e = 0; e < 4; ++e) x4[e] = adv ? xr[s][e] : x4[e];
        }
    };
    advance();
#pragma unroll 1
    while (__any(j0 < j_end)) {
        // a trip = RS_BATCHES batches of eight survivors: 8 RS_BATCHES gathers in flight per group before the first is used
        const int row = row0 + 32 * t;
        int32_t my[RS_BATCHES];
        bool mine[RS_BATCHES];
        f32x4 w[RS_BATCHES][8];
#pragma unroll
        for (int h = 0; h < RS_BATCHES; ++h) {
            my[h] = (j0 + 8 * h + li < j_end) ? a.surv_idx[(size_t)row * REFINE_CAP + j0 + 8 * h + li] : -1;
            mine[h] = my[h] >= lat_lo && my[h] < lat_hi;
        }
#pragma unroll
        for (int h = 0; h < RS_BATCHES; ++h) {
            // (no survivor: -1 becomes an out-of-bounds offset: zeros, no memory access)
            const uint32_t off_my = mine[h] ? (uint32_t)my[h] * 128u : 0xFFFFFF00u;
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const uint32_t off = (uint32_t)__builtin_amdgcn_ds_bpermute(sel + 4 * u, (int)off_my);
                w[h][u] = rs_buf_load16(wres, off | li16);
            }
        }
#pragma unroll
        for (int h = 0; h < RS_BATCHES; ++h) {
            float p[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) p[u] = x4[0] * w[h][u][0] + x4[1] * w[h][u][1] + x4[2] * w[h][u][2] + x4[3] * w[h][u][3];
            const float r = group8_reduce_scatter(p, li);
            if (mine[h]) part[(size_t)row * REFINE_CAP + j0 + 8 * h + li] = r;
        }
        j0 += 8 * RS_BATCHES;
        advance();
    }
}
// surv_val[row][j] = b_enc[latent] + the D / 32 shares in slice order; one wave per row
__global__ __launch_bounds__(256) void refine_sum_kernel(RefineSlicesArgs a) {
    if (a.enable_flag != nullptr && (*a.enable_flag != 0) != (a.enable_when != 0)) return;
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= a.n_rows) return;
    const int cnt = a.surv_cnt[row];
    const int n_slices = a.D / RS_SLICE;
    const size_t plane = (size_t)a.n_rows * REFINE_CAP;
    for (int j = lane; j < cnt; j += 64) {
        const size_t o = (size_t)row * REFINE_CAP + j;
        float s = 0.f;
        int c = 0;
        for (; c + 8 <= n_slices; c += 8) {  // eight loads in flight, added in slice order
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = a.part[(size_t)(c + u) * plane + o];
#pragma unroll
            for (int u = 0; u < 8; ++u) s += v[u];
        }
        for (; c < n_slices; ++c) s += a.part[(size_t)c * plane + o];
        a.surv_val[o] = s + a.b_enc[a.surv_idx[o]];
    }
}

// max over rows of the L2 norm of an (R x D) matrix: one wave per row, per-workgroup maxima, then one small pass
__global__ __launch_bounds__(256) void rownorm_wgmax_kernel(const float* m, int R, int D, float* wg_max) {
    __shared__ float sh[4];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int r = blockIdx.x * 4 + w;
    float s = 0.f;
    if (r < R) {
        const f32x4* p = reinterpret_cast<const f32x4*>(m + (size_t)r * D);
        for (int q = lane; q < (D >> 2); q += 64) {
            const f32x4 v = p[q];
            s += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
        }
        s = wave_sum(s);
    }
    if (lane == 0) sh[w] = sqrtf(s);
    __syncthreads();
    if (threadIdx.x == 0) wg_max[blockIdx.x] = fmaxf(fmaxf(sh[0], sh[1]), fmaxf(sh[2], sh[3]));
}
__global__ __launch_bounds__(1024) void max_reduce_kernel(const float* v, int n, float* out) {
    __shared__ float sh[16];
    float m = 0.f;
    for (int i = threadIdx.x; i < n; i += 1024) m = fmaxf(m, v[i]);
    m = wave_max(m);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.f;
        for (int i = 0; i < 16; ++i) t = fmaxf(t, sh[i]);
        *out = t;
    }
}
// The first pass of the f16r encoder works on CENTRED activations: h = (x - mu) W + (mu W + b) for any vector mu, and
// with mu = the batch's column mean the fp16 rounding error scales with ||x - mu|| instead of ||x|| -- ViT residual
// streams carry a large common offset (and "massive activation" channels), which would otherwise inflate the margin
// until every latent survives.  The exact refinement keeps using x and b_enc themselves.
//
// per row: ||x_b - mu||; per workgroup: max |x - mu| (thousands of same-address atomics would serialise: two stages)
//
// xmax (device scalar max|x| of the batch, or NULL): the squares are taken of (x - mu) / max|x| -- at most 4 each -- and the
// norm scaled back, so that activations of any magnitude (|x| ~ 1e20 squares to inf in fp32) get a finite margin.
// v rounded to fp16's 11 significant bits (round to nearest even) without fp16's range: what the image of v * 2^e holds,
// divided by 2^e again, for every power of two that keeps the product a normal fp16 number
__device__ __forceinline__ float round_f16_sig(float v) {
    const uint32_t b = __float_as_uint(v);
    return __uint_as_float((b + 0x0FFFu + ((b >> 13) & 1u)) & 0xFFFFE000u);
}
// xnorm: two floats per row -- ||x_b - mu|| and ||delta_b||, delta = the rounding error of the row's fp16 image
__global__ __launch_bounds__(256) void center_stats_kernel(const float* x, const float* mu, int n, int D,
                                                           float* xnorm, float* wg_max, const float* xmax) {
    __shared__ float sh[4];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int r = blockIdx.x * 4 + w;
    float s = 0.f, m = 0.f, sd = 0.f;
    const float up = xmax != nullptr ? *xmax : 1.0f;
    const float back = (up > 0.f && up < 3.0e38f) ? up : 1.0f, inv = 1.0f / back;
    if (r < n) {
        const f32x4* p = reinterpret_cast<const f32x4*>(x + (size_t)r * D);
        const f32x4* mu4 = reinterpret_cast<const f32x4*>(mu);
        const int nq = D >> 2;
        for (int q0 = lane; q0 < nq; q0 += 256) {
            f32x4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int q = q0 + 64 * u;
                v[u] = q < nq ? p[q] - mu4[q] : f32x4{0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                m = fmaxf(fmaxf(fmaxf(m, fabsf(v[u][0])), fabsf(v[u][1])), fmaxf(fabsf(v[u][2]), fabsf(v[u][3])));
                f32x4 d;
#pragma unroll
                for (int e = 0; e < 4; ++e) d[e] = (v[u][e] - round_f16_sig(v[u][e])) * inv;
                v[u] = v[u] * inv;
                s += v[u][0] * v[u][0] + v[u][1] * v[u][1] + v[u][2] * v[u][2] + v[u][3] * v[u][3];
                sd += d[0] * d[0] + d[1] * d[1] + d[2] * d[2] + d[3] * d[3];
            }
        }
        s = wave_sum(s);
        sd = wave_sum(sd);
        if (lane == 0) { xnorm[2 * r] = back * sqrtf(s); xnorm[2 * r + 1] = back * sqrtf(sd) * 1.000001f; }
    }
    m = wave_max(m);
    if (lane == 0) sh[w] = m;
    __syncthreads();
    if (threadIdx.x == 0) wg_max[blockIdx.x] = fmaxf(fmaxf(sh[0], sh[1]), fmaxf(sh[2], sh[3]));
}
// The two pieces row_margin_kernel and pre_encode_kernel share (ONE definition: the refinement is only exact if both
// kernels agree on the margin).  f16r_margin: 2 E_b of the comment below.  f16r_scale_ok: the W images of this step were
// scaled with the power of two derived from the PREVIOUS call's largest column norm; they are safe while the current
// norm times that scale stays below the fp16 range and at most two bits under the intended [2^13, 2^14) window (an
// all-zero W_enc has exact images).
__device__ __forceinline__ float f16r_margin(float xnorm_row, float xdelta_row, float wmax, float dwmax, float bmax, int D, float x_scale) {
    const float rnd = 1.02f * (xdelta_row * wmax + xnorm_row * dwmax + xdelta_row * dwmax);
    const float acc = (1.05f * (float)D * 2.384185791015625e-07f + 7.62939453125e-06f) * xnorm_row * wmax;
    // elements of the x image below fp16's normal range (2^-14 after scaling; the scale follows the BATCH's largest element, so a
    // row far smaller than its batch can sit there whole): each is off by at most 2^-14 / x_scale, flushed or not
    // (a batch that really holds such a row -- seven orders of magnitude below its largest element after centring -- sees this term
    // dominate that row's margin, its list overflow, and the step take the exact dense route: tests/test_gpu_parity.py,
    // tools/experiments/r4_tiny_rows_probe.py)
    const float sub = sqrtf((float)D) * 6.103515625e-05f / x_scale * wmax;
    const float m = 2.0f * (rnd + acc + sub) + 2.0f * 1.1920929e-07f * bmax;
    // (a row with an inf / NaN element has a NaN rounding-error norm: its margin must read as "keep everything" -- the list then
    // overflows and the exact dense route treats the row as torch does -- not as a NaN that every comparison fails)
    return m == m ? m : __builtin_inff();
}
__device__ __forceinline__ bool f16r_scale_ok(float wmax, float w_scale) {
    const float t = wmax * w_scale;
    return t < 60000.0f && (t >= 2048.0f || wmax == 0.f);
}
// margin[b] = 2 E_b,
//   E_b = 1.02 (||dx_b|| wmax + ||x_b - mu|| dwmax + ||dx_b|| dwmax) + (1.05 D 2^-22 + 2^-17) ||x_b - mu|| wmax + 2^-23 max |b_shift|,
// wmax = max_s ||W_enc[:, s]||, dwmax = max_s ||dW[:, s]||: an upper bound of the error of a pre-activation formed from
// fp16-rounded operands.  dx_b and dW[:, s] are the rounding errors the images ACTUALLY carry (center_stats_kernel and the W
// image pass measure their norms; products of fp16 numbers are exact in fp32), so that
// |sum (x + dx)(w + dw) - sum x w| <= ||dx|| ||w|| + ||x|| ||dw|| + ||dx|| ||dw|| by Cauchy-Schwarz -- on real data 0.35-0.4 of
// the a-priori 2^-11 per operand this margin used until round 4.  Then the fp32 accumulation of D terms (counted at 2^-22 per
// add so that a truncating adder is covered) and the rounding of the shifted bias.  The operands are pre-scaled so that
// their largest element sits in [2^13, 2^14): whatever the matrix cores do with fp16 subnormals (flush or keep) then changes a
// pre-activation by less than sqrt(D) 2^-14 / scale times the other operand's norm -- for W that is inside the 2^-17 term
// (with the fp32 rounding of x - mu), for x it is the absolute term `sub` (a row may be far smaller than its batch).
// DESIGN.md 3.1.
//
// wg_part holds the per-workgroup maxima bias_finish_kernel left behind: |b_shift| in [0, n_part), column norms in
// [n_part, 2 n_part).  Every workgroup reduces them again (a few hundred values) instead of waiting for two more tiny
// launches; workgroup 0 also runs the scale check: the W images of this step were scaled with the power of two derived
// from the PREVIOUS call's largest column norm (so that one pass over W_enc suffices).  Parameters move a little per
// step, but they belong to the caller and may have been replaced: if the largest column norm of the current W_enc
// leaves the window in which the fp16 images are safe (no overflow; no more than two bits below the intended range),
// raise the dense-route flag -- the step then runs on the exact fp32 kernel -- and in any case remember the current
// norm for the next call.
__global__ __launch_bounds__(256) void row_margin_kernel(const float* xnorm, int n, int D, const float* wg_part, int n_part,
                                                         const float* scales, int32_t* pre_flag, float* wmax_prev,
                                                         float* margin) {
    __shared__ float sh[3][4];
    float bm = 0.f, wm = 0.f, dm = 0.f;
    for (int i = threadIdx.x; i < n_part; i += 256) { bm = fmaxf(bm, wg_part[i]); wm = fmaxf(wm, wg_part[n_part + i]); dm = fmaxf(dm, wg_part[2 * n_part + i]); }
    bm = wave_max(bm); wm = wave_max(wm); dm = wave_max(dm);
    if ((threadIdx.x & 63) == 0) { sh[0][threadIdx.x >> 6] = bm; sh[1][threadIdx.x >> 6] = wm; sh[2][threadIdx.x >> 6] = dm; }
    __syncthreads();
    const float bmax = fmaxf(fmaxf(sh[0][0], sh[0][1]), fmaxf(sh[0][2], sh[0][3]));
    const float wmax = fmaxf(fmaxf(sh[1][0], sh[1][1]), fmaxf(sh[1][2], sh[1][3]));
    const float dwmax = fmaxf(fmaxf(sh[2][0], sh[2][1]), fmaxf(sh[2][2], sh[2][3]));
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        if (!f16r_scale_ok(wmax, scales[1])) *pre_flag = 1;
        *wmax_prev = wmax;
    }
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r >= n) return;
    margin[r] = f16r_margin(xnorm[2 * r], xnorm[2 * r + 1], wmax, dwmax, bmax, D, scales[0]);  // (scales = {x scale, W scale} of this step's images: f16r_scales_kernel)
}
// Everything the fused encoder launch needs zeroed or derived right before it, in one pass (three launches before):
//   * the per-launch state of the encoder: candidate counters 0, shared group maxima "-inf";
//   * f16r (xnorm != NULL): the row margins and the scale check of row_margin_kernel above;
//   * the step's list flags: flags1[0] need_dense = the force-dense flag *pre_flag as it stands after the scale check,
//     flags1[1] n_overflow = 0, flags1[2] cand_max = 0 (select_cand_kernel raises them).
__global__ __launch_bounds__(256) void pre_encode_kernel(int32_t* cand_cnt, int n_rows, int32_t* gmax, int n_gmax,
                                                         const float* xnorm, int D, const float* wg_part, int n_part,
                                                         const float* scales, int32_t* pre_flag, float* wmax_prev,
                                                         float* margin, int32_t* flags1) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n_rows) cand_cnt[i] = 0;
    if (i < n_gmax) gmax[i] = INT32_MIN;
    if (xnorm == nullptr) {
        if (i == 0) { flags1[0] = (pre_flag != nullptr && *pre_flag != 0) ? 1 : 0; flags1[1] = 0; flags1[2] = 0; }
        return;
    }
    if ((int)blockIdx.x * 256 >= n_rows && blockIdx.x != 0) return;  // (blocks past the rows only initialise)
    __shared__ float sh[3][4];
    float bm = 0.f, wm = 0.f, dm = 0.f;
    for (int j = threadIdx.x; j < n_part; j += 256) { bm = fmaxf(bm, wg_part[j]); wm = fmaxf(wm, wg_part[n_part + j]); dm = fmaxf(dm, wg_part[2 * n_part + j]); }
    bm = wave_max(bm); wm = wave_max(wm); dm = wave_max(dm);
    if ((threadIdx.x & 63) == 0) { sh[0][threadIdx.x >> 6] = bm; sh[1][threadIdx.x >> 6] = wm; sh[2][threadIdx.x >> 6] = dm; }
    __syncthreads();
    const float bmax = fmaxf(fmaxf(sh[0][0], sh[0][1]), fmaxf(sh[0][2], sh[0][3]));
    const float wmax = fmaxf(fmaxf(sh[1][0], sh[1][1]), fmaxf(sh[1][2], sh[1][3]));
    const float dwmax = fmaxf(fmaxf(sh[2][0], sh[2][1]), fmaxf(sh[2][2], sh[2][3]));
    if (i == 0) {
        int pre = *pre_flag != 0 ? 1 : 0;
        if (!f16r_scale_ok(wmax, scales[1])) { pre = 1; *pre_flag = 1; }
        *wmax_prev = wmax;
        flags1[0] = pre; flags1[1] = 0; flags1[2] = 0;
    }
    if (i >= n_rows) return;
    margin[i] = f16r_margin(xnorm[2 * i], xnorm[2 * i + 1], wmax, dwmax, bmax, D, scales[0]);  // (scales = {x scale, W scale} of this step's images: f16r_scales_kernel)
}
// The streamed f16r step's second (and last) preparation launch, behind xprep_kernel (kernels.h: PreEncode2Args):
//   * rows: the two norms of x_b - mu from their 32-column pieces, the margin, the candidate counter; the shared group maxima;
//   * workgroup 0: the batch maxima (max |x| for the MSE's rescale; max |x - mu| -> the NEXT step's x scale, and the check that
//     THIS step's images, scaled with the previous batch's maximum, stayed inside fp16: otherwise the dense route), the W scale
//     check of pre_encode_kernel, the step's flags and statistics block;
//   * workgroups past the rows: the next centring vector mu = column sums / n (only a step whose Adam will rebuild the W images
//     with it moves mu: `update_mu`).
__global__ __launch_bounds__(256) void pre_encode2_kernel(PreEncode2Args a) {
    __shared__ float sh[5][4];
    const int i = blockIdx.x * 256 + threadIdx.x;
    if ((int)blockIdx.x >= a.nb_rows) {  // ---- mu ----
        const int d = ((int)blockIdx.x - a.nb_rows) * 256 + threadIdx.x;
        if (d >= a.D || !a.update_mu) return;
        float t = 0.f;
        int b = 0;
        for (; b + 8 <= a.n_rowblk; b += 8) {  // (eight loads in flight, added in block order)
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = a.col_part[(size_t)(b + u) * a.D + d];
#pragma unroll
            for (int u = 0; u < 8; ++u) t += v[u];
        }
        for (; b < a.n_rowblk; ++b) t += a.col_part[(size_t)b * a.D + d];
        t *= a.inv_n;
        if (t - t == 0.f) a.mu[d] = t;  // (a non-finite column sum -- an inf / NaN in the batch -- keeps the old centre: any vector is valid)
        return;
    }
    if (i < a.n_rows) a.cand_cnt[i] = 0;
    if (i < a.n_gmax) a.gmax[i] = INT32_MIN;
    float bm = 0.f, wm = 0.f, dm = 0.f, am = 0.f, cm = 0.f;
    for (int j = threadIdx.x; j < a.n_part; j += 256) { bm = fmaxf(bm, a.wg_part[j]); wm = fmaxf(wm, a.wg_part[a.n_part + j]); dm = fmaxf(dm, a.wg_part[2 * a.n_part + j]); }
    if (blockIdx.x == 0)
        for (int j = threadIdx.x; j < a.n_img; j += 256) { am = fmaxf(am, a.amax_part[j]); cm = fmaxf(cm, a.cmax_part[j]); }
    bm = wave_max(bm); wm = wave_max(wm); dm = wave_max(dm); am = wave_max(am); cm = wave_max(cm);
    if ((threadIdx.x & 63) == 0) { const int w = threadIdx.x >> 6; sh[0][w] = bm; sh[1][w] = wm; sh[2][w] = dm; sh[3][w] = am; sh[4][w] = cm; }
    __syncthreads();
    const float bmax = fmaxf(fmaxf(sh[0][0], sh[0][1]), fmaxf(sh[0][2], sh[0][3]));
    const float wmax = fmaxf(fmaxf(sh[1][0], sh[1][1]), fmaxf(sh[1][2], sh[1][3]));
    const float dwmax = fmaxf(fmaxf(sh[2][0], sh[2][1]), fmaxf(sh[2][2], sh[2][3]));
    const float x_scale = a.scales[0];
    if (i == 0) {
        const float amax = fmaxf(fmaxf(sh[3][0], sh[3][1]), fmaxf(sh[3][2], sh[3][3]));
        const float cmax = fmaxf(fmaxf(sh[4][0], sh[4][1]), fmaxf(sh[4][2], sh[4][3]));
        int pre = 0;
        if (!f16r_scale_ok(wmax, a.scales[1])) pre = 1;
        if (!(cmax * x_scale < 60000.0f)) pre = 1;  // an element of this step's x image left fp16's range (or is not finite)
        if (a.xside_keep != nullptr && !(cmax * x_scale < 60000.0f)) a.xside_keep[1] = 1.0f;  // (the followers' images are the same ones)
        if (a.stale != nullptr) {
            const bool caught = *a.stale != 0;  // the parameters are not the ones the images were made of (XprepArgs::stale)
            if (caught) {
                pre = 1;
                *a.stale = 0;
                if (a.stale_host != nullptr) *a.stale_host = 1;
            }
            a.stale[1] = caught ? 1 : 0;  // (this step takes the exact route: the tile checksums of its Adam have nothing to report)
        }
        *a.pre_flag = pre;
        *a.wmax_prev = wmax;
        a.flags1[0] = pre; a.flags1[1] = 0; a.flags1[2] = 0;
        *a.upper = amax;
        *a.stats = saev_step_stats{};
        const float xs_next = (cmax > 0.f && cmax < 3.0e38f) ? exp2f(13.0f - floorf(log2f(cmax))) : x_scale;
        a.scales_next[0] = xs_next;
        a.scales_next[2] = xs_next;
        a.scales_next[4] = (amax > 0.f && amax < 3.0e38f) ? amax : a.scales[4];
    }
    if (i >= a.n_rows) return;
    const float up = a.scales[4];  // (the normaliser xprep_kernel used)
    const float back = (up > 0.f && up < 3.0e38f) ? up : 1.0f;
    float s2 = 0.f, d2 = 0.f;
    int ks = 0;
    for (; ks + 8 <= a.nks; ks += 8) {  // (image order: fixed; eight loads in flight)
        float2 t[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) t[u] = reinterpret_cast<const float2*>(a.xn_part)[(size_t)(ks + u) * a.n_pad + i];
#pragma unroll
        for (int u = 0; u < 8; ++u) { s2 += t[u].x; d2 += t[u].y; }
    }
    for (; ks < a.nks; ++ks) {
        const float2 t = reinterpret_cast<const float2*>(a.xn_part)[(size_t)ks * a.n_pad + i];
        s2 += t.x; d2 += t.y;
    }
    const float xn = back * sqrtf(s2) * 1.000001f, xd = back * sqrtf(d2) * 1.000002f;  // (rounded up: they bound errors)
    if (a.xnorm != nullptr) { a.xnorm[2 * i] = xn; a.xnorm[2 * i + 1] = xd; }
    a.margin[i] = f16r_margin(xn, xd, wmax, dwmax, bmax, a.D, x_scale);
}
__global__ void follower_scales_kernel(const float* xside, const float* wmax, float* scales, int32_t* pre_flag, int keep_w) {
    if (threadIdx.x != 0) return;
    scales[0] = xside[0];
    scales[2] = xside[0];
    scales[3] = 1.0f;
    if (!keep_w) {
        const float wm = *wmax;
        scales[1] = (wm > 0.f && wm < 3.0e38f) ? exp2f(13.0f - floorf(log2f(wm))) : 1.0f;
    }
    if (xside[1] != 0.f) *pre_flag = 1;
}
// {2^e, 1} with 2^e * absmax in [2^13, 2^14): operand scale for an fp16 split of a matrix whose magnitude is only known
// on the device (AuxK codes and gradients)
__global__ void pow2_scale_kernel(const float* absmax, float* pair) {
    if (threadIdx.x == 0) {
        const float m = *absmax;
        pair[0] = (m > 0.f && m < 3.0e38f) ? exp2f(13.0f - floorf(log2f(m))) : 1.0f;
        pair[1] = 1.0f;
    }
}
// power-of-two scales that put the largest |x| and the largest encoder column norm (>= largest |w|) into [2^13, 2^14)
// (xmax_part: per-workgroup maxima of |x - mu| from center_stats_kernel, reduced here)
__global__ __launch_bounds__(256) void f16r_scales_kernel(const float* xmax_part, int n_part, const float* wmax, float* scales) {
    __shared__ float sh[4];
    float m = 0.f;
    for (int i = threadIdx.x; i < n_part; i += 256) m = fmaxf(m, xmax_part[i]);
    m = wave_max(m);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float xm = fmaxf(fmaxf(sh[0], sh[1]), fmaxf(sh[2], sh[3])), wm = *wmax;
        scales[0] = (xm > 0.f && xm < 3.0e38f) ? exp2f(13.0f - floorf(log2f(xm))) : 1.0f;
        scales[1] = (wm > 0.f && wm < 3.0e38f) ? exp2f(13.0f - floorf(log2f(wm))) : 1.0f;
        scales[2] = scales[0];  // {x scale, 1}: for contractions whose second operand carries a fixed scale (AuxK)
        scales[3] = 1.0f;
    }
}

// bad = pre_flag || any(cand_cnt > cap); also counts overflowing rows, the longest list and the mean list length.
// Stage 1 of a predicted-bound launch (`pre_flag` = the prediction gate): *bad_out = gate || overflow, *dense_out =
// *dense_src (only a failed scale check sends the step straight to the dense route; an overflow of the predicted-bound
// attempt is first retried with guaranteed bounds).
// Stage 2 (the retry; `enable_flag` = stage 1's bad flag): when enabled *dense_out |= overflow and *run_out = !*dense_out,
// when disabled *run_out = 0 and nothing else is touched.
__global__ void overflow_check_kernel(const int32_t* cand_cnt, int n_rows, int cap, const int32_t* pre_flag,
                                      int32_t* bad_out, int32_t* n_overflow, int32_t* cand_max, int32_t* dense_out,
                                      int32_t* run_out, float* cand_mean, const int32_t* enable_flag,
                                      const int32_t* dense_src) {
    if (enable_flag != nullptr && *enable_flag == 0) {
        if (threadIdx.x == 0 && run_out != nullptr) *run_out = 0;
        return;
    }
    __shared__ int sh, shmax, shsum;
    if (threadIdx.x == 0) { sh = 0; shmax = 0; shsum = 0; }
    __syncthreads();
    int c = 0, m = 0, t = 0;
    for (int i = threadIdx.x; i < n_rows; i += blockDim.x) { const int v = cand_cnt[i]; c += (v > cap); m = max(m, v); t += min(v, cap); }
    c = wave_sum_i(c);
    t = wave_sum_i(t);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = max(m, __shfl_xor(m, o, 64));
    if ((threadIdx.x & 63) == 0) { if (c) atomicAdd(&sh, c); atomicMax(&shmax, m); atomicAdd(&shsum, t); }
    __syncthreads();
    if (threadIdx.x == 0) {
        const int pre = (pre_flag && *pre_flag) ? 1 : 0;
        *n_overflow = sh;
        *cand_max = shmax;
        if (cand_mean != nullptr) *cand_mean = (float)shsum / (float)max(n_rows, 1);
        if (run_out == nullptr) {  // a single-stage launch, or stage 1
            *bad_out = (pre || sh > 0) ? 1 : 0;
            if (dense_out != nullptr) *dense_out = dense_src != nullptr ? (*dense_src != 0 ? 1 : 0) : pre;
        } else {                   // stage 2
            const int dense = (*dense_out != 0 || sh > 0) ? 1 : 0;
            *dense_out = dense;
            *run_out = dense ? 0 : 1;
        }
    }
}

// The z of the predicted bounds (mean + z sigma) follows what the launches show: a failed prediction (some row's k-th
// largest candidate below a bound that was used for it, or an overflow) costs a second, guaranteed-bound launch, so z
// drops by a lot and climbs back slowly (150 launches for one drop).  Three failures in a row (data
// the extrapolation does not fit) switch prediction off for 256 launches -- `gate` then sends them straight to the
// guaranteed bounds -- after which it starts again from a cautious z.
// state: [0] z  [1] failed predictions  [2] predicted launches  [3] mean list length  [4] failures in a row  [5] launches
// left without prediction  (floats; counts are exact up to 2^24)
__global__ void heur_gate_kernel(float* state, const int32_t* pre_flag, int32_t* gate) {
    if (threadIdx.x != 0) return;
    int g = (pre_flag != nullptr && *pre_flag != 0) ? 1 : 0;
    if (state[5] > 0.f) { state[5] -= 1.f; g = 1; }
    *gate = g;
}
__global__ void heur_update_kernel(float* state, const int32_t* bad, const float* cand_mean, int k, const int32_t* gate) {
    if (threadIdx.x != 0 || *gate != 0) return;
    float z = state[0];
    state[2] += 1.f;
    if (*bad != 0) {
        z -= 0.3f;
        state[1] += 1.f;
        state[4] += 1.f;
        if (state[4] >= 3.f) { state[4] = 0.f; state[5] = 256.f; z = 1.5f; }
    } else {
        state[4] = 0.f;
        if (*cand_mean > 3.f * (float)k) z += 0.002f;
    }
    state[0] = fminf(fmaxf(z, 0.5f), 3.5f);
}

}  // namespace

hipError_t launch_select_dense(const SelectDenseArgs& a, hipStream_t stream) {
    if (a.n_rows <= 0) return hipSuccess;
    hipLaunchKernelGGL(select_dense_kernel, dim3(a.n_rows), dim3(SD_THREADS), 0, stream, a);
    return hipGetLastError();
}

hipError_t launch_select_cand(const SelectCandArgs& a, hipStream_t stream) {
    if (a.n_rows <= 0) return hipSuccess;
    if (a.cand_cap > 4096 || (a.sum_part != nullptr && a.cand_cap > REFINE_CAP)) return hipErrorInvalidValue;
    if (a.k > 32) hipLaunchKernelGGL(select_cand_kernel<true>, dim3((a.n_rows + 3) / 4), dim3(256), 0, stream, a);
    else hipLaunchKernelGGL(select_cand_kernel<false>, dim3((a.n_rows + 3) / 4), dim3(256), 0, stream, a);
    return hipGetLastError();
}

hipError_t launch_init_i32(int32_t* p, int32_t v, int n, hipStream_t stream) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(init_i32_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, p, v, n);
    return hipGetLastError();
}

hipError_t launch_encoder_init(int32_t* cand_cnt, int n_rows, int32_t* gmax, int n_gmax, hipStream_t stream, int32_t* tau_max,
                               const int32_t* enable_flag, int enable_when) {
    const int n = n_rows > n_gmax ? n_rows : n_gmax;
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(encoder_init_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, cand_cnt, n_rows, gmax, n_gmax, tau_max,
                       enable_flag, enable_when);
    return hipGetLastError();
}
hipError_t launch_pre_encode(int32_t* cand_cnt, int n_rows, int32_t* gmax, int n_gmax, const float* xnorm, int D,
                             const float* wg_part, int n_part, const float* scales, int32_t* pre_flag, float* wmax_prev,
                             float* margin, int32_t* flags1, hipStream_t stream) {
    const int n = std::max(1, n_rows > n_gmax ? n_rows : n_gmax);
    hipLaunchKernelGGL(pre_encode_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, cand_cnt, n_rows, gmax, n_gmax, xnorm, D,
                       wg_part, n_part, scales, pre_flag, wmax_prev, margin, flags1);
    return hipGetLastError();
}
hipError_t launch_follower_scales(const float* xside, const float* wmax, float* scales, int32_t* pre_flag, int keep_w, hipStream_t stream) {
    hipLaunchKernelGGL(follower_scales_kernel, dim3(1), dim3(64), 0, stream, xside, wmax, scales, pre_flag, keep_w);
    return hipGetLastError();
}
hipError_t launch_pre_encode2(PreEncode2Args a, hipStream_t stream) {
    a.nb_rows = (std::max(1, std::max(a.n_rows, a.n_gmax)) + 255) / 256;
    hipLaunchKernelGGL(pre_encode2_kernel, dim3(a.nb_rows + (a.D + 255) / 256), dim3(256), 0, stream, a);
    return hipGetLastError();
}
hipError_t launch_heur_gate(float* state, const int32_t* pre_flag, int32_t* gate, hipStream_t stream) {
    hipLaunchKernelGGL(heur_gate_kernel, dim3(1), dim3(64), 0, stream, state, pre_flag, gate);
    return hipGetLastError();
}
hipError_t launch_heur_update(float* state, const int32_t* bad, const float* cand_mean, int k, const int32_t* gate,
                              hipStream_t stream) {
    hipLaunchKernelGGL(heur_update_kernel, dim3(1), dim3(64), 0, stream, state, bad, cand_mean, k, gate);
    return hipGetLastError();
}
hipError_t launch_step_zero(saev_step_stats* stats, float* upper, int32_t* flag0, hipStream_t stream) {
    hipLaunchKernelGGL(step_zero_kernel, dim3(1), dim3(64), 0, stream, stats, upper, flag0);
    return hipGetLastError();
}

hipError_t launch_wnorm_max(const float* W_encT, int S, int D, float* wg_scratch, float* wmax, hipStream_t stream) {
    const int nwg = (S + 3) / 4;
    hipLaunchKernelGGL(rownorm_wgmax_kernel, dim3(nwg), dim3(256), 0, stream, W_encT, S, D, wg_scratch);
    hipLaunchKernelGGL(max_reduce_kernel, dim3(1), dim3(1024), 0, stream, wg_scratch, nwg, wmax);
    return hipGetLastError();
}
hipError_t launch_f16r_scales(const float* xmax_part, int n_part, const float* wmax, float* scales, hipStream_t stream) {
    hipLaunchKernelGGL(f16r_scales_kernel, dim3(1), dim3(256), 0, stream, xmax_part, n_part, wmax, scales);
    return hipGetLastError();
}
hipError_t launch_pow2_scale(const float* absmax, float* pair, hipStream_t stream) {
    hipLaunchKernelGGL(pow2_scale_kernel, dim3(1), dim3(64), 0, stream, absmax, pair);
    return hipGetLastError();
}
hipError_t launch_center_stats(const float* x, const float* mu, int n, int D, float* xnorm, float* wg_absmax,
                               hipStream_t stream, const float* xmax) {
    hipLaunchKernelGGL(center_stats_kernel, dim3((n + 3) / 4), dim3(256), 0, stream, x, mu, n, D, xnorm, wg_absmax, xmax);
    return hipGetLastError();
}
hipError_t launch_max_reduce(const float* v, int n, float* out, hipStream_t stream) {
    hipLaunchKernelGGL(max_reduce_kernel, dim3(1), dim3(1024), 0, stream, v, n, out);
    return hipGetLastError();
}
hipError_t launch_row_margins(const float* xnorm, int n, int D, const float* wg_part, int n_part, const float* scales,
                              int32_t* pre_flag, float* wmax_prev, float* margin, hipStream_t stream) {
    hipLaunchKernelGGL(row_margin_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, xnorm, n, D, wg_part, n_part, scales,
                       pre_flag, wmax_prev, margin);
    return hipGetLastError();
}

hipError_t launch_refine_exact(const SelectCandArgs& a, hipStream_t stream) {
    if (a.n_rows <= 0) return hipSuccess;
    const dim3 grid((a.n_rows + 3) / 4), block(256);
    const int nv = (a.D / 4 + 63) / 64;
#define RF(N) hipLaunchKernelGGL(refine_exact_kernel<N>, grid, block, 0, stream, a)
    if (nv <= 1) RF(1); else if (nv <= 2) RF(2); else if (nv <= 3) RF(3); else if (nv <= 4) RF(4);
    else if (nv <= 6) RF(6); else if (nv <= 8) RF(8); else if (nv <= 12) RF(12); else if (nv <= 16) RF(16);
    else return hipErrorInvalidValue;
#undef RF
    return hipGetLastError();
}

hipError_t launch_refine_slices(const RefineSlicesArgs& a, hipStream_t stream, bool sum_shares) {
    if (a.n_rows <= 0) return hipSuccess;
    if (a.D % RS_SLICE != 0 || (uint64_t)a.S * 128ull >= (1ull << 32) - 256ull || a.n_ranges <= 0) return hipErrorInvalidValue;
    // rows per eight-lane group: 8 evens out the survivor lists of a group's rows (283 -> 257 us at 16 384 rows); a small batch
    // does not have the workgroups for it (4 096 rows, d_model 768: 47 us at 4 rows per group, 71 at 8)
    const int rows = a.n_rows >= 8192 ? RS_ROWS_MAX : 4;
    const int wg_per_combo = (a.n_rows + 32 * rows - 1) / (32 * rows);
    const int slice_blocks = (a.D / RS_SLICE + 7) / 8;
    const dim3 grid(8 * slice_blocks * a.n_ranges * wg_per_combo);
    if (rows == 4) hipLaunchKernelGGL(refine_slices_kernel<4>, grid, dim3(256), 0, stream, a, wg_per_combo);
    else hipLaunchKernelGGL(refine_slices_kernel<RS_ROWS_MAX>, grid, dim3(256), 0, stream, a, wg_per_combo);
    // (adding the shares inside the final select instead -- 32 dependent loads per lane of a kernel that lives on its bit
    // search -- took that select from 20 to 80 us against 42 for this pass)
    if (sum_shares) hipLaunchKernelGGL(refine_sum_kernel, dim3((a.n_rows + 3) / 4), dim3(256), 0, stream, a);
    return hipGetLastError();
}

hipError_t launch_select_refine(const SelectCandArgs& a, hipStream_t stream) {
    if (a.n_rows <= 0) return hipSuccess;
    if (a.cand_cap > 4096 || a.row_margin == nullptr) return hipErrorInvalidValue;
    const dim3 grid((a.n_rows + 3) / 4), block(256);
    const int nv = (a.D / 4 + 63) / 64;
#define RF(N) hipLaunchKernelGGL(select_refine_kernel<N>, grid, block, 0, stream, a)
    if (nv <= 1) RF(1); else if (nv <= 2) RF(2); else if (nv <= 3) RF(3); else if (nv <= 4) RF(4);
    else if (nv <= 6) RF(6); else if (nv <= 8) RF(8); else if (nv <= 12) RF(12); else if (nv <= 16) RF(16);
    else return hipErrorInvalidValue;
#undef RF
    return hipGetLastError();
}

hipError_t launch_overflow_check(const int32_t* cand_cnt, int n_rows, int cap, const int32_t* pre_flag,
                                 int32_t* need_dense, int32_t* n_overflow, int32_t* cand_max, hipStream_t stream,
                                 int32_t* dense_out, int32_t* run_out, float* cand_mean, const int32_t* enable_flag,
                                 const int32_t* dense_src) {
    hipLaunchKernelGGL(overflow_check_kernel, dim3(1), dim3(1024), 0, stream, cand_cnt, n_rows, cap, pre_flag,
                       need_dense, n_overflow, cand_max, dense_out, run_out, cand_mean, enable_flag, dense_src);
    return hipGetLastError();
}
