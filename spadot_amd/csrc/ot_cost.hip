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
//   3. the collected entries are sorted and the two middle ranks read off -- exact, whatever the sample was (the counts
//      prove it; a miss, which needs an 8-sigma event, falls back to sorting everything);
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
                                                      u64 lo, u64 hi, u64 *__restrict__ counts, u64 *__restrict__ cand, u64 cap) {
    constexpr unsigned LCAP = 2048;
    __shared__ u64 sh_below[4];
    __shared__ u64 lbuf[LCAP];
    __shared__ unsigned lcnt;
    __shared__ u64 gbase;
    if (threadIdx.x == 0) lcnt = 0;
    __syncthreads();
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
                                                    double denom, T *__restrict__ C) {
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
    const size_t xx_b = ((size_t)I * 8 + 255) / 256 * 256;
    const size_t need = xx_b + 256 + 2 * (size_t)S * 8 + 2 * cap * 8 + tmp_b;
    int rc = reserve(ws, need);
    if (rc) return rc;
    unsigned char *base = (unsigned char *)ws->ptr;
    double *xx = (double *)base;                 base += xx_b;
    u64 *counts = (u64 *)base;                   base += 256;
    u64 *skeys = (u64 *)base;                    base += (size_t)S * 8;
    u64 *ssort = (u64 *)base;                    base += (size_t)S * 8;
    u64 *cand = (u64 *)base;                     base += cap * 8;
    u64 *csort = (u64 *)base;                    base += cap * 8;
    void *tmp = base;
    size_t tb = tmp_b;
    hipLaunchKernelGGL(k_cost_norms, dim3((I + 255) / 256), dim3(256), 0, st, x, d, I, xx);
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
            u64 lo = 0, hi = ~0ull >> 1;                                  // (all keys are non-negative doubles)
            if (r1 >= 0) COST_CHECK(hipMemcpyAsync(&lo, ssort + r1, sizeof(u64), hipMemcpyDeviceToHost, st));
            if (r2 < S) COST_CHECK(hipMemcpyAsync(&hi, ssort + r2, sizeof(u64), hipMemcpyDeviceToHost, st));
            COST_CHECK(hipMemsetAsync(counts, 0, sizeof(u64) * 2, st));
            COST_CHECK(hipStreamSynchronize(st));
            hipLaunchKernelGGL(k_cost_bracket<DD>, grid_j, dim3(256), 0, st, x, y, xx, d, I, J, rpb, lo, hi, counts, cand, (u64)cap);
            u64 hc[2] = {0, 0};
            COST_CHECK(hipMemcpyAsync(hc, counts, sizeof(u64) * 2, hipMemcpyDeviceToHost, st));
            COST_CHECK(hipStreamSynchronize(st));
            ncand = (long long)hc[1];
            if (hc[1] <= cap && k1 >= hc[0] && k2 - hc[0] < hc[1]) {       // both middle ranks are inside the bracket
                tb = tmp_b;
                COST_CHECK(rocprim::radix_sort_keys(tmp, tb, cand, csort, (size_t)hc[1], 0, 64, st));
                COST_CHECK(hipMemcpyAsync(&m[0], csort + (k1 - hc[0]), sizeof(u64), hipMemcpyDeviceToHost, st));
                COST_CHECK(hipMemcpyAsync(&m[1], csort + (k2 - hc[0]), sizeof(u64), hipMemcpyDeviceToHost, st));
                COST_CHECK(hipStreamSynchronize(st));
                done = true;
                path = 1;
            } else {
                path = 2;
            }
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
            hipLaunchKernelGGL((k_cost_write<double, DD>), grid_j, dim3(256), 0, st, x, y, xx, d, I, J, J, rpb, 1.0, (double *)all);
            COST_CHECK(rocprim::radix_sort_keys(t2, t2b, all, sorted, n, 0, 64, st));
            COST_CHECK(hipMemcpyAsync(&m[0], sorted + k1, sizeof(u64), hipMemcpyDeviceToHost, st));
            COST_CHECK(hipMemcpyAsync(&m[1], sorted + k2, sizeof(u64), hipMemcpyDeviceToHost, st));
            COST_CHECK(hipStreamSynchronize(st));
            if (big) (void)hipFree(big);
        }
        denom = (n & 1) ? key_to_double(m[0]) : (key_to_double(m[0]) + key_to_double(m[1])) / 2.0;
    }
    if (storage_f32)
        hipLaunchKernelGGL((k_cost_write<float, DD>), grid_ld, dim3(256), 0, st, x, y, xx, d, I, J, ld, rpb, denom, (float *)C);
    else
        hipLaunchKernelGGL((k_cost_write<double, DD>), grid_ld, dim3(256), 0, st, x, y, xx, d, I, J, ld, rpb, denom, (double *)C);
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
