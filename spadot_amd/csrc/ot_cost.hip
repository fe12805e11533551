// ot_cost.hip -- cost matrix of a pair problem straight from the latents (gfx950).
//
// What it replaces: /root/reference/SpaDOT/utils/OT_loss/ot_solvers.py:101-103
//     C = sklearn.metrics.pairwise_distances(a, b, metric='sqeuclidean');  C = C / np.median(C)
// i.e. |x|^2 + |y|^2 - 2 x.y clipped at 0 (fp64) and the EXACT median of all I x J entries (np.median: the middle
// element, or the mean of the two middle ones).
//
// Round 1 wrote the I x J fp64 distance matrix (781 MB at 10k x 10k), read it back twelve times for a radix select and
// once more to scale it: ~11 GB of HBM traffic in front of a 10 ms solve.  Here the distances are never stored:
//   1. a fixed pseudo-random SAMPLE of 2^18 entries is computed and sorted (rocPRIM): the sample quantiles 2048 ranks
//      (8 sigma) either side of the median rank bracket the two middle order statistics;
//   2. ONE pass over all I x J distances (recomputed on the fly: 20 fp64 FMAs each) counts the entries below the
//      bracket and collects the ~1.6 % inside it;
//   3. the two middle ranks are read off the collected entries -- exact, whatever the sample was (the counts prove it; a
//      miss, which needs an 8-sigma event, falls back to sorting everything).  Round 5: ON THE DEVICE, with no host
//      read-back between the passes -- the bracket comes from the sorted sample through a device word, the collected
//      entries are bucketed by value (2^16-2^17 equal-width buckets over the bracket: one histogram pass), the bucket(s) that
//      hold the two ranks are gathered (a few dozen entries) and sorted by one workgroup, and the divisor stays in device
//      memory for pass 4.  Rounds 2-4 sorted the ~1.6 M collected entries (rocPRIM) and read three values back to the host
//      in between: 0.85 ms per pair set-up at 10k x 10k, of which the two passes over the distances were 0.36;
//   4. one more on-the-fly pass writes C = d / median in the solver's storage type (fp32: 400 MB).
// Every pass evaluates the same device function, so the entries compared in 2/3 are bit-identical to the ones written
// in 4.  Small problems (<= 2^22 entries) skip the sampling: all entries are sorted.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>

#include <rocprim/rocprim.hpp>

namespace {

constexpr int MAX_D = 32;
typedef unsigned long long u64;

#define COST_CHECK(expr)                                   \
    do {                                                   \
        hipError_t e_ = (expr);                            \
        if (e_ != hipSuccess) return (int)e_ + 1000;       \
    } while (0)

// |x|^2 + |y|^2 - 2 x.y clipped at 0 -- sklearn's euclidean_distances(squared=True) arithmetic: the dot product as an
// fma chain over k, then v = -2 dot, v += |x|^2, v += |y|^2.  DD = the latent dimension as a compile-time bound (20:
// z_dim of the model; MAX_D: anything else, predicated) so that no multiply is spent on padding; |x_i|^2 comes from
// k_cost_norms (same chain), the column's latent yj and |y_j|^2 live in registers.
template <int DD>
__device__ __forceinline__ double dist_row(const double *__restrict__ xi, double xx, const double *yj, double yy, int d) {
    double dot = 0.0;
#pragma unroll
    for (int k = 0; k < DD; k++) {
        const double xv = (DD == MAX_D && k >= d) ? 0.0 : xi[k];
        dot += xv * yj[k];
    }
    double v = -2.0 * dot;
    v += xx;
    v += yy;
    return v > 0.0 ? v : 0.0;
}
template <int DD>
__device__ __forceinline__ double load_col(const double *__restrict__ y, int j, bool live, int d, double *yj) {
    double yy = 0.0;
#pragma unroll
    for (int k = 0; k < DD; k++) {
        yj[k] = (live && (DD != MAX_D || k < d)) ? y[(size_t)j * d + k] : 0.0;
        yy += yj[k] * yj[k];
    }
    return yy;
}
__global__ __launch_bounds__(256) void k_cost_norms(const double *__restrict__ x, int d, int I, double *__restrict__ xx) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= I) return;
    double s = 0.0;
    for (int k = 0; k < d; k++) s += x[(size_t)i * d + k] * x[(size_t)i * d + k];
    xx[i] = s;
}

// keys[t] = bit pattern of the distance of a pseudo-random (i, j) (non-negative doubles order like their bits)
template <int DD>
__global__ __launch_bounds__(256) void k_cost_sample(const double *__restrict__ x, const double *__restrict__ y,
                                                     const double *__restrict__ xx, int d, int I, int J, int S,
                                                     u64 *__restrict__ keys) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= S) return;
    u64 z = (u64)t * 0x9E3779B97F4A7C15ull + 0xD1B54A32D192ED03ull;       // splitmix64
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    const int i = (int)((z >> 32) % (u64)I), j = (int)((z & 0xffffffffull) % (u64)J);
    double yj[DD];
    const double yy = load_col<DD>(y, j, true, d, yj);
    keys[t] = (u64)__double_as_longlong(dist_row<DD>(x + (size_t)i * d, xx[i], yj, yy, d));
}

// counts[0] += #(key < lo), counts[1] += #(lo <= key <= hi) and those keys are appended to cand (first `cap` of them).
// The candidates of a workgroup (~1.6 % of its 256 x 64 entries) are gathered in LDS and handed over with ONE global
// atomic (a global atomic per candidate would serialise ~1.6 M adds on one word).
template <int DD>
__global__ __launch_bounds__(256) void k_cost_bracket(const double *__restrict__ x, const double *__restrict__ y,
                                                      const double *__restrict__ xx, int d, int I, int J, int rows_per_block,
                                                      u64 lo, u64 hi, u64 *__restrict__ counts, u64 *__restrict__ cand, u64 cap,
                                                      const u64 *__restrict__ bounds) {
    constexpr unsigned LCAP = 2048;
    __shared__ u64 sh_below[4];
    __shared__ u64 lbuf[LCAP];
    __shared__ unsigned lcnt;
    __shared__ u64 gbase;
    if (threadIdx.x == 0) lcnt = 0;
    __syncthreads();
    if (bounds != nullptr) { lo = bounds[0]; hi = bounds[1]; }       // (the bracket left in device memory by k_sel_init)
    const int j = blockIdx.x * 256 + threadIdx.x;
    const bool live = j < J;
    double yj[DD];
    const double yy = load_col<DD>(y, j, live, d, yj);
    const int i0 = blockIdx.y * rows_per_block, i1 = min(I, i0 + rows_per_block);
    u64 below = 0;
#pragma unroll 2
    for (int i = i0; i < i1; i++) {
        const u64 key = (u64)__double_as_longlong(dist_row<DD>(x + (size_t)i * d, xx[i], yj, yy, d));
        if (live) {
            if (key < lo) below++;
            else if (key <= hi) {
                const unsigned p = atomicAdd(&lcnt, 1u);
                if (p < LCAP) lbuf[p] = key;
                else {                                       // (a workgroup with > 2048 candidates: straight to global)
                    const u64 pos = atomicAdd(&counts[1], 1ull);
                    if (pos < cap) cand[pos] = key;
                }
            }
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) below += __shfl_xor(below, off, 64);
    if ((threadIdx.x & 63) == 0) sh_below[threadIdx.x >> 6] = below;
    __syncthreads();
    const unsigned nl = min(lcnt, LCAP);
    if (threadIdx.x == 0) {
        atomicAdd(&counts[0], sh_below[0] + sh_below[1] + sh_below[2] + sh_below[3]);
        gbase = nl ? atomicAdd(&counts[1], (u64)nl) : 0ull;
    }
    __syncthreads();
    for (unsigned t = threadIdx.x; t < nl; t += 256)
        if (gbase + t < cap) cand[gbase + t] = lbuf[t];
}

// C[i, j] = d(i, j) / denom in the storage type; pad columns (J <= j < ld) are 0
template <typename T, int DD>
__global__ __launch_bounds__(256) void k_cost_write(const double *__restrict__ x, const double *__restrict__ y,
                                                    const double *__restrict__ xx, int d, int I, int J, int ld, int rows_per_block,
                                                    double denom, T *__restrict__ C, const double *__restrict__ denom_p) {
    if (denom_p != nullptr) denom = *denom_p;
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= ld) return;
    const bool live = j < J;
    double yj[DD];
    const double yy = load_col<DD>(y, j, live, d, yj);
    const int i0 = blockIdx.y * rows_per_block, i1 = min(I, i0 + rows_per_block);
#pragma unroll 2
    for (int i = i0; i < i1; i++) {
        const double v = live ? dist_row<DD>(x + (size_t)i * d, xx[i], yj, yy, d) / denom : 0.0;
        C[(size_t)i * ld + j] = (T)v;
    }
}

// ------------------------------------------------------------------------------------------------------------------
// The x.y term on the fp64 matrix cores (round 4).  v_mfma_f64_16x16x4_f64 with its accumulator fed back is, on this
// hardware, bit for bit the sequential fma chain over k = 0, 1, 2, ... that dist_row() runs (tools/mfma_f64_chain_probe.hip:
// 1 048 576 random 20-term chains, 0 differences; profiles/r04/mfma_f64_chain_probe.txt), so the distances -- and with them
// the exact median -- do not change by a bit; latent dimensions that are not a multiple of 4 are padded with zeros
// (fma(0, 0, acc) = acc).  The fp64 matrix rate equals the vector rate on this chip; what the matrix cores buy is the
// vector ALU: the per-entry epilogue (scale, clamp, compare or convert, ~8 fp64 operations) no longer queues behind 20 FMAs.
//   wave = 64 columns (four 16-column tiles, the B operands y and |y|^2 in registers for the whole launch) x `rows_per_block`
//   rows in blocks of 16 (A: x rows, 5 loads of one double per lane); C/D: col = lane & 15, row = (lane >> 4) + 4 reg.
typedef double double4_t __attribute__((ext_vector_type(4)));

template <int STEPS>
struct MfmaCols {
    double b[4][STEPS];
    double yy[4];
    bool live[4];
};
template <int STEPS>
__device__ __forceinline__ void mfma_load_cols(const double *__restrict__ y, const double *__restrict__ yyv, int d, int J, int j0, int lane,
                                               MfmaCols<STEPS> &c) {
#pragma unroll
    for (int t = 0; t < 4; t++) {
        const int j = j0 + 16 * t + (lane & 15);
        c.live[t] = j < J;
        c.yy[t] = c.live[t] ? yyv[j] : 0.0;
#pragma unroll
        for (int s = 0; s < STEPS; s++) {
            const int k = 4 * s + (lane >> 4);
            c.b[t][s] = (c.live[t] && k < d) ? y[(size_t)j * d + k] : 0.0;
        }
    }
}
// distances of rows i0 .. i0 + 15 against the wave's 64 columns: out[t][r] for row i0 + (lane >> 4) + 4 r, col tile t
template <int STEPS>
__device__ __forceinline__ void mfma_dist_block(const double *__restrict__ x, const double *__restrict__ xx, int d, int I, int i0, int lane,
                                                const MfmaCols<STEPS> &c, double (*out)[4]) {
    double a[STEPS];
    const int ia = min(i0 + (lane & 15), I - 1);
#pragma unroll
    for (int s = 0; s < STEPS; s++) {
        const int k = 4 * s + (lane >> 4);
        a[s] = k < d ? x[(size_t)ia * d + k] : 0.0;
    }
    double xr[4];
#pragma unroll
    for (int r = 0; r < 4; r++) xr[r] = xx[min(i0 + (lane >> 4) + 4 * r, I - 1)];
#pragma unroll
    for (int t = 0; t < 4; t++) {
        double4_t acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int s = 0; s < STEPS; s++) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[s], c.b[t][s], acc, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; r++) {
            double v = -2.0 * acc[r];           // sklearn's chain: v = -2 dot; v += |x|^2; v += |y|^2; clip at 0
            v += xr[r];
            v += c.yy[t];
            out[t][r] = v > 0.0 ? v : 0.0;
        }
    }
}

template <typename T, int STEPS>
__global__ __launch_bounds__(256) void k_cost_write_mfma(const double *__restrict__ x, const double *__restrict__ y,
                                                         const double *__restrict__ xx, const double *__restrict__ yyv, int d, int I, int J,
                                                         int ld, int rows_per_block, double denom, int use_recip, T *__restrict__ C,
                                                         const double *__restrict__ denom_p) {
    if (denom_p != nullptr) denom = *denom_p;                         // (the median left in device memory by k_sel_final)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j0 = ((int)blockIdx.x * 4 + wave) * 64;
    if (j0 >= ld) return;
    MfmaCols<STEPS> c;
    mfma_load_cols<STEPS>(y, yyv, d, J, j0, lane, c);
    const double inv = 1.0 / denom;
    const int i_beg = blockIdx.y * rows_per_block, i_end = min(I, i_beg + rows_per_block);
    for (int i0 = i_beg; i0 < i_end; i0 += 16) {
        double v[4][4];
        mfma_dist_block<STEPS>(x, xx, d, I, i0, lane, c, v);
#pragma unroll
        for (int t = 0; t < 4; t++) {
            const int j = j0 + 16 * t + (lane & 15);
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int i = i0 + (lane >> 4) + 4 * r;
                if (i < i_end && j < ld) {
                    // fp64 storage: the division of the reference (bit-identical); fp32 storage: the product with the reciprocal
                    // (its error, one fp64 ulp, vanishes in the rounding to fp32 that follows)
                    const double q = c.live[t] ? (use_recip ? v[t][r] * inv : v[t][r] / denom) : 0.0;
                    C[(size_t)i * ld + j] = (T)q;
                }
            }
        }
    }
}

template <int STEPS>
__global__ __launch_bounds__(256) void k_cost_bracket_mfma(const double *__restrict__ x, const double *__restrict__ y,
                                                           const double *__restrict__ xx, const double *__restrict__ yyv, int d, int I, int J,
                                                           int rows_per_block, u64 lo, u64 hi, u64 *__restrict__ counts,
                                                           u64 *__restrict__ cand, u64 cap, const u64 *__restrict__ bounds) {
    constexpr unsigned LCAP = 2048;
    __shared__ u64 sh_below[4];
    __shared__ u64 lbuf[LCAP];
    __shared__ unsigned lcnt;
    __shared__ u64 gbase;
    if (threadIdx.x == 0) lcnt = 0;
    __syncthreads();
    if (bounds != nullptr) { lo = bounds[0]; hi = bounds[1]; }       // (the bracket left in device memory by k_sel_init)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j0 = ((int)blockIdx.x * 4 + wave) * 64;
    u64 below = 0;
    if (j0 < J) {
        MfmaCols<STEPS> c;
        mfma_load_cols<STEPS>(y, yyv, d, J, j0, lane, c);
        const int i_beg = blockIdx.y * rows_per_block, i_end = min(I, i_beg + rows_per_block);
        for (int i0 = i_beg; i0 < i_end; i0 += 16) {
            double v[4][4];
            mfma_dist_block<STEPS>(x, xx, d, I, i0, lane, c, v);
#pragma unroll
            for (int t = 0; t < 4; t++)
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const int i = i0 + (lane >> 4) + 4 * r;
                    if (c.live[t] && i < i_end) {
                        const u64 key = (u64)__double_as_longlong(v[t][r]);
                        if (key < lo) below++;
                        else if (key <= hi) {
                            const unsigned p = atomicAdd(&lcnt, 1u);
                            if (p < LCAP) lbuf[p] = key;
                            else {
                                const u64 pos = atomicAdd(&counts[1], 1ull);
                                if (pos < cap) cand[pos] = key;
                            }
                        }
                    }
                }
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) below += __shfl_xor(below, off, 64);
    if (lane == 0) sh_below[wave] = below;
    __syncthreads();
    const unsigned nl = min(lcnt, LCAP);
    if (threadIdx.x == 0) {
        atomicAdd(&counts[0], sh_below[0] + sh_below[1] + sh_below[2] + sh_below[3]);
        gbase = nl ? atomicAdd(&counts[1], (u64)nl) : 0ull;
    }
    __syncthreads();
    for (unsigned t = threadIdx.x; t < nl; t += 256)
        if (gbase + t < cap) cand[gbase + t] = lbuf[t];
}

// ------------------------------------------------------------------------------------------------------------------
// Round 5: the two middle order statistics of the collected entries without sorting them and without the host.
// State words st[] (u64, device): 0 lo, 1 hi (the bracket), 2 entries below the bracket, 3 entries inside it (the two
// counters of the bracket pass), 4 / 5 the ranks wanted INSIDE the bracket, 6 / 7 first / last bucket to gather, 8 status
// (0 ok, 1 the bracket missed a rank or overflowed, 2 the gathered list overflowed), 9 the divisor (bits of a double),
// 10 gathered entries, 11 entries in front of the first gathered bucket, 12 bucket shift.
// Keys are non-negative doubles: they order like their bit patterns, and over the narrow bracket a bucket = (key - lo) >> sh
// is an equal-width cut.
constexpr int SEL_WORDS = 16;
constexpr unsigned SEL_MAXB = 1u << 17;          // buckets: hi - lo + 1 shifted down to 17 bits or fewer
constexpr unsigned SEL_LIST = 4096;              // gathered entries one workgroup sorts

__global__ void k_sel_init(const u64 *__restrict__ ssort, long long r1, long long r2, int S, u64 *__restrict__ st) {
    st[0] = r1 >= 0 ? ssort[r1] : 0ull;
    st[1] = r2 < S ? ssort[r2] : (~0ull >> 1);
    for (int k = 2; k < SEL_WORDS; k++) st[k] = 0ull;
}

// after the bracket pass: are both middle ranks inside the bracket?  ranks inside it, bucket shift
__global__ void k_sel_begin(u64 *__restrict__ st, u64 k1, u64 k2, u64 cap) {
    const u64 below = st[2], inside = st[3];
    if (inside > cap || k1 < below || k2 - below >= inside) { st[8] = 1; return; }
    st[4] = k1 - below; st[5] = k2 - below;
    const u64 width = st[1] - st[0];                       // largest key offset
    int sh = 0;
    while ((width >> sh) >= (u64)SEL_MAXB) sh++;
    st[12] = (u64)sh;
}

__global__ __launch_bounds__(256) void k_sel_hist(const u64 *__restrict__ cand, const u64 *__restrict__ st, unsigned *__restrict__ hist) {
    if (st[8] != 0) return;
    const u64 n = st[3], lo = st[0];
    const int sh = (int)st[12];
    for (u64 i = (u64)blockIdx.x * 256 + threadIdx.x; i < n; i += (u64)gridDim.x * 256)
        atomicAdd(&hist[(unsigned)((cand[i] - lo) >> sh)], 1u);
}

// the buckets that hold ranks st[4] and st[5]: one workgroup, 1024 threads x 128 buckets each, partial sums through LDS
__global__ __launch_bounds__(1024) void k_sel_pick(u64 *__restrict__ st, const unsigned *__restrict__ hist) {
    if (st[8] != 0) return;
    __shared__ u64 part[1024];
    const int t = threadIdx.x;
    constexpr int PER = SEL_MAXB / 1024;
    u64 s = 0;
    for (int k = 0; k < PER; k++) s += hist[t * PER + k];
    part[t] = s;
    __syncthreads();
    if (t == 0) {
        for (int target = 0; target < 2; target++) {
            const u64 want = st[4 + target];
            u64 cum = 0;
            int g = 0;
            while (g < 1023 && cum + part[g] <= want) { cum += part[g]; g++; }
            int bkt = g * PER;
            while (bkt < g * PER + PER - 1 && cum + hist[bkt] <= want) { cum += hist[bkt]; bkt++; }
            st[6 + target] = (u64)bkt;
            if (target == 0) st[11] = cum;                 // entries in front of the first gathered bucket
        }
    }
}

__global__ __launch_bounds__(256) void k_sel_gather(const u64 *__restrict__ cand, u64 *__restrict__ st, u64 *__restrict__ list) {
    if (st[8] != 0) return;
    const u64 n = st[3], lo = st[0], b0 = st[6], b1 = st[7];
    const int sh = (int)st[12];
    for (u64 i = (u64)blockIdx.x * 256 + threadIdx.x; i < n; i += (u64)gridDim.x * 256) {
        const u64 key = cand[i], bkt = (key - lo) >> sh;
        if (bkt >= b0 && bkt <= b1) {
            const u64 pos = atomicAdd((unsigned long long *)&st[10], 1ull);
            if (pos < SEL_LIST) list[pos] = key;
        }
    }
}

// sorts the gathered entries (bitonic, one workgroup) and leaves the divisor: the middle entry, or the mean of the two
__global__ __launch_bounds__(1024) void k_sel_final(u64 *__restrict__ st, const u64 *__restrict__ list, int odd) {
    if (st[8] != 0) return;
    __shared__ u64 v[SEL_LIST];
    const u64 m = st[10];
    if (m > SEL_LIST || m == 0) { if (threadIdx.x == 0) st[8] = 2; return; }
    unsigned P = 64;                                       // sort width: the next power of two (a few dozen entries as a rule)
    while (P < (unsigned)m) P <<= 1;
    for (unsigned i = threadIdx.x; i < P; i += 1024) v[i] = i < m ? list[i] : ~0ull;
    __syncthreads();
    for (unsigned k = 2; k <= P; k <<= 1)
        for (unsigned j = k >> 1; j > 0; j >>= 1) {
            for (unsigned i = threadIdx.x; i < P; i += 1024) {
                const unsigned l = i ^ j;
                if (l > i) {
                    const u64 a = v[i], b = v[l];
                    const bool up = (i & k) == 0;
                    if ((a > b) == up) { v[i] = b; v[l] = a; }
                }
            }
            __syncthreads();
        }
    if (threadIdx.x == 0) {
        const u64 ia = st[4] - st[11], ib = st[5] - st[11];
        if (ib >= m) { st[8] = 2; return; }
        const double a = __longlong_as_double((long long)v[ia]), b = __longlong_as_double((long long)v[ib]);
        const double denom = odd ? a : (a + b) / 2.0;       // np.median: the middle element, or the mean of the two middle ones
        st[9] = (u64)__double_as_longlong(denom);
    }
}

struct Workspace {           // device scratch kept by the solver between calls (hipMalloc costs more than the passes)
    void *ptr = nullptr;
    size_t bytes = 0;
};

int reserve(Workspace *ws, size_t bytes) {
    if (ws->bytes >= bytes) return 0;
    if (ws->ptr) (void)hipFree(ws->ptr);
    ws->ptr = nullptr; ws->bytes = 0;
    COST_CHECK(hipMalloc(&ws->ptr, bytes));
    ws->bytes = bytes;
    return 0;
}

size_t sort_tmp_bytes(size_t n) {
    size_t b = 0;
    (void)rocprim::radix_sort_keys(nullptr, b, (u64 *)nullptr, (u64 *)nullptr, n, 0, 64, (hipStream_t) nullptr);
    return (b + 255) / 256 * 256;
}

double key_to_double(u64 k) {
    double v;
    memcpy(&v, &k, sizeof(v));
    return v;
}

template <int DD>
int run(const double *x, const double *y, int d, int I, int J, int ld, int storage_f32, void *C, int divide_by_median,
        hipStream_t st, Workspace *ws, double *denom_out, long long *info_out) {
    const size_t n = (size_t)I * J;
    const int rpb = 64;
    const dim3 grid_j((J + 255) / 256, (I + rpb - 1) / rpb), grid_ld((ld + 255) / 256, (I + rpb - 1) / rpb);
    const bool sampled = n > ((size_t)1 << 22);
    const int S = 1 << 18, DELTA = 2048;
    const size_t cap = sampled ? n / 25 + 65536 : n;                      // 4 % of the entries: the bracket holds ~1.6 %
    const size_t tmp_b = sort_tmp_bytes(cap > (size_t)S ? cap : (size_t)S);
    const size_t xx_b = ((size_t)I * 8 + 255) / 256 * 256, yy_b = ((size_t)J * 8 + 255) / 256 * 256;
    const size_t sel_b = 256 + (size_t)SEL_MAXB * 4 + (size_t)SEL_LIST * 8;       // state words, bucket histogram, gathered list
    const size_t need = xx_b + yy_b + 256 + 2 * (size_t)S * 8 + 2 * cap * 8 + tmp_b + sel_b;
    int rc = reserve(ws, need);
    if (rc) return rc;
    unsigned char *base = (unsigned char *)ws->ptr;
    double *xx = (double *)base;                 base += xx_b;
    double *yyv = (double *)base;                base += yy_b;
    // the two passes over all I x J distances on the fp64 matrix cores (bit-identical to the vector kernels' fma chain)
    constexpr int STEPS = DD == 20 ? 5 : (MAX_D + 3) / 4;
    constexpr bool use_mfma = true;
    const dim3 grid_mj((J + 255) / 256, (I + rpb - 1) / rpb), grid_mld((ld + 255) / 256, (I + rpb - 1) / rpb);
    u64 *counts = (u64 *)base;                   base += 256;
    u64 *skeys = (u64 *)base;                    base += (size_t)S * 8;
    u64 *ssort = (u64 *)base;                    base += (size_t)S * 8;
    u64 *cand = (u64 *)base;                     base += cap * 8;
    u64 *csort = (u64 *)base;                    base += cap * 8;
    void *tmp = base;                            base += tmp_b;
    u64 *st_dev = (u64 *)base;                   base += 256;
    unsigned *hist = (unsigned *)base;           base += (size_t)SEL_MAXB * 4;
    u64 *list = (u64 *)base;
    size_t tb = tmp_b;
    const double *denom_dev = nullptr;           // non-null: the divisor is in device memory (st_dev[9])
    hipLaunchKernelGGL(k_cost_norms, dim3((I + 255) / 256), dim3(256), 0, st, x, d, I, xx);
    if (use_mfma) hipLaunchKernelGGL(k_cost_norms, dim3((J + 255) / 256), dim3(256), 0, st, y, d, J, yyv);     // (the chain load_col() runs)
    double denom = 1.0;
    long long path = 0, ncand = 0;
    if (divide_by_median) {
        const size_t k1 = (n & 1) ? n / 2 : n / 2 - 1, k2 = n / 2;       // the two middle ranks (equal for odd n)
        u64 m[2] = {0, 0};
        bool done = false;
        if (sampled) {
            hipLaunchKernelGGL(k_cost_sample<DD>, dim3((S + 255) / 256), dim3(256), 0, st, x, y, xx, d, I, J, S, skeys);
            COST_CHECK(rocprim::radix_sort_keys(tmp, tb, skeys, ssort, (size_t)S, 0, 64, st));
            const long long r1 = (long long)((double)k1 / (double)n * S) - DELTA, r2 = (long long)((double)k2 / (double)n * S) + DELTA;
            // everything from here to the write pass is enqueued without a host read-back (round 5): bracket through st_dev[0..1],
            // counters st_dev[2..3], bucket histogram, the one or two buckets that hold the middle ranks, their sort, the divisor
            hipLaunchKernelGGL(k_sel_init, dim3(1), dim3(1), 0, st, (const u64 *)ssort, r1, r2, S, st_dev);
            COST_CHECK(hipMemsetAsync(hist, 0, (size_t)SEL_MAXB * 4, st));
            if (use_mfma)
                hipLaunchKernelGGL(k_cost_bracket_mfma<STEPS>, grid_mj, dim3(256), 0, st, x, y, xx, yyv, d, I, J, rpb, 0ull, 0ull, st_dev + 2, cand,
                                   (u64)cap, (const u64 *)st_dev);
            else
                hipLaunchKernelGGL(k_cost_bracket<DD>, grid_j, dim3(256), 0, st, x, y, xx, d, I, J, rpb, 0ull, 0ull, st_dev + 2, cand, (u64)cap,
                                   (const u64 *)st_dev);
            hipLaunchKernelGGL(k_sel_begin, dim3(1), dim3(1), 0, st, st_dev, (u64)k1, (u64)k2, (u64)cap);
            hipLaunchKernelGGL(k_sel_hist, dim3(1024), dim3(256), 0, st, (const u64 *)cand, (const u64 *)st_dev, hist);
            hipLaunchKernelGGL(k_sel_pick, dim3(1), dim3(1024), 0, st, st_dev, (const unsigned *)hist);
            hipLaunchKernelGGL(k_sel_gather, dim3(1024), dim3(256), 0, st, (const u64 *)cand, st_dev, list);
            hipLaunchKernelGGL(k_sel_final, dim3(1), dim3(1024), 0, st, st_dev, (const u64 *)list, (int)(n & 1));
            denom_dev = reinterpret_cast<const double *>(st_dev + 9);
            done = true;
            path = 1;
        }
        if (!done) {          // small problem (or a missed bracket): every entry, sorted
            u64 *all = cand, *sorted = csort;
            void *big = nullptr;
            void *t2 = tmp;
            size_t t2b = tmp_b;
            if (n > cap) {                                                // (missed bracket on a large problem: own buffers)
                t2b = sort_tmp_bytes(n);
                COST_CHECK(hipMalloc(&big, 2 * n * 8 + t2b));
                all = (u64 *)big; sorted = all + n; t2 = sorted + n;
            }
            hipLaunchKernelGGL((k_cost_write<double, DD>), grid_j, dim3(256), 0, st, x, y, xx, d, I, J, J, rpb, 1.0, (double *)all, (const double *)nullptr);
            COST_CHECK(rocprim::radix_sort_keys(t2, t2b, all, sorted, n, 0, 64, st));
            COST_CHECK(hipMemcpyAsync(&m[0], sorted + k1, sizeof(u64), hipMemcpyDeviceToHost, st));
            COST_CHECK(hipMemcpyAsync(&m[1], sorted + k2, sizeof(u64), hipMemcpyDeviceToHost, st));
            COST_CHECK(hipStreamSynchronize(st));
            if (big) (void)hipFree(big);
        }
        if (!(done && sampled)) denom = (n & 1) ? key_to_double(m[0]) : (key_to_double(m[0]) + key_to_double(m[1])) / 2.0;
    }
    auto write_pass = [&](double dn, const double *dn_dev) {
        if (use_mfma && storage_f32)
            hipLaunchKernelGGL((k_cost_write_mfma<float, STEPS>), grid_mld, dim3(256), 0, st, x, y, xx, yyv, d, I, J, ld, rpb, dn, 1, (float *)C, dn_dev);
        else if (use_mfma)
            hipLaunchKernelGGL((k_cost_write_mfma<double, STEPS>), grid_mld, dim3(256), 0, st, x, y, xx, yyv, d, I, J, ld, rpb, dn, 0, (double *)C, dn_dev);
        else if (storage_f32)
            hipLaunchKernelGGL((k_cost_write<float, DD>), grid_ld, dim3(256), 0, st, x, y, xx, d, I, J, ld, rpb, dn, (float *)C, dn_dev);
        else
            hipLaunchKernelGGL((k_cost_write<double, DD>), grid_ld, dim3(256), 0, st, x, y, xx, d, I, J, ld, rpb, dn, (double *)C, dn_dev);
    };
    write_pass(denom, denom_dev);
    if (denom_dev != nullptr) {
        // the ONE read-back of the sampled path: status, counters, divisor.  A miss (an 8-sigma sample, or more than 4096 equal
        // entries around the median) sorts everything as before and writes the matrix again.
        u64 hst[SEL_WORDS];
        COST_CHECK(hipMemcpyAsync(hst, st_dev, sizeof(hst), hipMemcpyDeviceToHost, st));
        COST_CHECK(hipStreamSynchronize(st));
        ncand = (long long)hst[3];
        if (hst[8] == 0) {
            denom = key_to_double(hst[9]);
        } else {
            path = 2;
            const size_t k1 = (n & 1) ? n / 2 : n / 2 - 1, k2 = n / 2;
            u64 m[2] = {0, 0};
            void *big = nullptr;
            u64 *all = cand, *sorted = csort;
            void *t2 = tmp;
            size_t t2b = tmp_b;
            if (n > cap) {
                t2b = sort_tmp_bytes(n);
                COST_CHECK(hipMalloc(&big, 2 * n * 8 + t2b));
                all = (u64 *)big; sorted = all + n; t2 = sorted + n;
            }
            hipLaunchKernelGGL((k_cost_write<double, DD>), grid_j, dim3(256), 0, st, x, y, xx, d, I, J, J, rpb, 1.0, (double *)all, (const double *)nullptr);
            COST_CHECK(rocprim::radix_sort_keys(t2, t2b, all, sorted, n, 0, 64, st));
            COST_CHECK(hipMemcpyAsync(&m[0], sorted + k1, sizeof(u64), hipMemcpyDeviceToHost, st));
            COST_CHECK(hipMemcpyAsync(&m[1], sorted + k2, sizeof(u64), hipMemcpyDeviceToHost, st));
            COST_CHECK(hipStreamSynchronize(st));
            if (big) (void)hipFree(big);
            denom = (n & 1) ? key_to_double(m[0]) : (key_to_double(m[0]) + key_to_double(m[1])) / 2.0;
            write_pass(denom, nullptr);
        }
    }
    COST_CHECK(hipStreamSynchronize(st));
    if (hipGetLastError() != hipSuccess) return 1000;
    if (denom_out) *denom_out = denom;
    if (info_out) { info_out[0] = path; info_out[1] = ncand; }
    return 0;
}

}  // namespace

// Writes C (I x ld, storage fp32 or fp64) = sqeuclid(x, y) [/ median]; *denom_out = the divisor used.  ws_ptr / ws_bytes:
// the caller's scratch allocation, grown here when too small (freed by the caller).  Returns 0, or 1000 + hipError on a
// HIP failure.  info_out (may be null): {path (0 all sorted, 1 sampled bracket, 2 bracket missed -> all sorted),
// candidates collected}.
int spadot_cost_from_latents_impl(const double *x, const double *y, int d, int I, int J, int ld, int storage_f32, void *C,
                                  int divide_by_median, hipStream_t st, void **ws_ptr, size_t *ws_bytes, double *denom_out,
                                  long long *info_out) {
    if (d < 1 || d > MAX_D || I <= 0 || J <= 0 || ld < J || !ws_ptr || !ws_bytes) return -22;
    Workspace ws;
    ws.ptr = *ws_ptr; ws.bytes = *ws_bytes;
    const int rc = (d == 20) ? run<20>(x, y, d, I, J, ld, storage_f32, C, divide_by_median, st, &ws, denom_out, info_out)
                             : run<MAX_D>(x, y, d, I, J, ld, storage_f32, C, divide_by_median, st, &ws, denom_out, info_out);
    *ws_ptr = ws.ptr; *ws_bytes = ws.bytes;
    return rc;
}
