// ot_sinkhorn.hip -- unbalanced entropic OT scaling iterations for MI355X (gfx950, wave64).
//
// What it replaces: /root/reference/SpaDOT/utils/OT_loss/ot_func.cpp (the C++ behind libot.so)
// and the per-stage Python driver in ot_solvers.py:164-449.  See include/spadot_ot.h for the ABI.
//
// Data layout in HBM (one problem):
//   C, K           I x ld row-major, element type T (double or float), ld = J rounded up to 64
//                  elements; pad columns of K are kept at exactly 0 so row/column sums ignore them
//   a,u,p,dx,...   fp64 vectors (length I or ld); scalings and every sum are fp64 in both modes
//   part           nchunk x ld fp64 column partials (row-chunked column pass, fixed order => the
//                  result is bitwise reproducible; no float atomics anywhere)
//
// Kernels (all HBM-bound; algorithmic bytes in DESIGN.md):
//   k_row_pass   one wave per row, 16-byte loads, fp64 wave-shuffle reduction, fused a-update
//   k_col_pass   256 threads x 16 bytes of one row per step, fp64 register accumulators per column
//   k_col_fin    sums the chunk partials in order, fused b-update and tau flag
//   k_absorb_*   u += eps ln a, v += eps ln b, K rebuilt from C; runs only when the tau flag is set
//   k_drift / k_gap_*  convergence measures (ot_func.cpp:886-923)
#include <hip/hip_runtime.h>
#include <cfloat>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <type_traits>
#include <vector>

#include "../../include/spadot_ot.h"
#include "per_device.h"

// Error handling: no HIP failure ends the host process.  Every HIP call is checked; a failure unwinds to the C entry
// point, which reports it on stderr and returns SPADOT_EHIP (part B: int results), NaN (part A: double / float results,
// which the reference's driver turns into a RuntimeError, ot_solvers.py:446-447) or SPADOT_EHIP again (step1_process_double).
// Temporaries are owned by DevBuf guards, so an unwinding call frees them.
struct hip_failure { hipError_t err; const char *file; int line; const char *expr; };
#define HIP_CHECK(expr)                                                        \
    do {                                                                       \
        hipError_t _e = (expr);                                                \
        if (_e != hipSuccess) throw hip_failure{_e, __FILE__, __LINE__, #expr}; \
    } while (0)
#define SPADOT_EHIP (-5)
static void report_failure(const hip_failure &f) {
    fprintf(stderr, "libspadot_ot: HIP error %s at %s:%d (%s) -- there is no CPU path\n", hipGetErrorString(f.err),
            f.file, f.line, f.expr);
    (void)hipGetLastError();      // clear the sticky error so that a later call reports its own
}
#define SPADOT_ENTER try {
#define SPADOT_LEAVE(ret)                                                        \
    } catch (const hip_failure &f_) {                                            \
        report_failure(f_);                                                      \
        return ret;                                                              \
    } catch (const std::exception &e_) {                                         \
        fprintf(stderr, "libspadot_ot: %s\n", e_.what());                        \
        return ret;                                                              \
    }

namespace {

constexpr int WAVE = 64;
constexpr int ROW_WAVES = 4;           // waves (= rows) per 256-thread block in row-oriented kernels
constexpr int MAX_BATCH = 64;          // max scaling iterations per convergence check
#ifndef SPADOT_OT_FAST_EXP
#define SPADOT_OT_FAST_EXP 1
#endif
constexpr bool FAST_EXP_F32 = SPADOT_OT_FAST_EXP != 0;   // fp32-stored K built with exp_to_f32 (0: fp64 exp(), A/B builds)

inline int round_up(int x, int m) { return (x + m - 1) / m * m; }

// 16-byte native vectors (ext_vector_type: usable as inline-asm "v" operands, unlike HIP's float4 struct)
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef double f64x2 __attribute__((ext_vector_type(2)));
template <typename T> struct Vec;
template <> struct Vec<float>  { using type = f32x4; static constexpr int N = 4; };
template <> struct Vec<double> { using type = f64x2; static constexpr int N = 2; };

template <typename T> __device__ __forceinline__ void unpack(const typename Vec<T>::type &v, double *o);
template <> __device__ __forceinline__ void unpack<float>(const f32x4 &v, double *o) {
    o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w;
}
template <> __device__ __forceinline__ void unpack<double>(const f64x2 &v, double *o) {
    o[0] = v.x; o[1] = v.y;
}
template <typename T> __device__ __forceinline__ typename Vec<T>::type pack(const double *o);
template <> __device__ __forceinline__ f32x4 pack<float>(const double *o) {
    return f32x4{(float)o[0], (float)o[1], (float)o[2], (float)o[3]};
}
template <> __device__ __forceinline__ f64x2 pack<double>(const double *o) {
    return f64x2{o[0], o[1]};
}

__device__ __forceinline__ double wave_sum(double x) {
#pragma unroll
    for (int off = WAVE / 2; off > 0; off >>= 1) x += __shfl_down(x, off, WAVE);
    return x;   // valid in lane 0
}

// The same sum by DPP moves (register to register: no LDS crossbar round trip per step as with ds_bpermute):
// inclusive scan inside each row of 16 lanes (row_shr 1, 2, 4, 8), then row_bcast:15 / row_bcast:31 carry the row
// totals upwards.  Result valid in lane 63.  (In the fused pass the bpermute form took 0.60 us of the 2.64 us a
// group of rows cost.)
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_add(double x) {
    const int lo = __double2loint(x), hi = __double2hiint(x);
    const int l2 = __builtin_amdgcn_update_dpp(0, lo, CTRL, ROW_MASK, 0xf, false);
    const int h2 = __builtin_amdgcn_update_dpp(0, hi, CTRL, ROW_MASK, 0xf, false);
    return x + __hiloint2double(h2, l2);
}
__device__ __forceinline__ double wave_sum63(double x) {
    x = dpp_add<0x111, 0xf>(x);   // row_shr:1
    x = dpp_add<0x112, 0xf>(x);   // row_shr:2
    x = dpp_add<0x114, 0xf>(x);   // row_shr:4
    x = dpp_add<0x118, 0xf>(x);   // row_shr:8
    x = dpp_add<0x142, 0xa>(x);   // row_bcast:15 into rows 1 and 3
    x = dpp_add<0x143, 0xc>(x);   // row_bcast:31 into rows 2 and 3
    return x;   // valid in lane 63
}
__device__ __forceinline__ double read_lane(double x, int l) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(x), l), __builtin_amdgcn_readlane(__double2loint(x), l));
}

// Sum over a 256..1024-thread block; every thread gets the result.  `sh` needs 16 doubles.
__device__ __forceinline__ double block_sum(double x, double *sh) {
    x = wave_sum(x);
    const int lane = threadIdx.x & (WAVE - 1), wid = threadIdx.x >> 6;
    const int nw = (blockDim.x + WAVE - 1) >> 6;
    __syncthreads();
    if (lane == 0) sh[wid] = x;
    __syncthreads();
    double t = 0.0;
    for (int k = 0; k < nw; k++) t += sh[k];   // fixed order
    return t;
}

// (num/sum)^alpha * exp(-shift), the scaling update of ot_func.cpp:633-636 / :665-668, evaluated as
// exp(alpha*log(num/sum) - shift): one log + one exp instead of pow + exp (same value to ~1e-15
// relative; 0, inf and NaN propagate exactly as pow*exp does).
__device__ __forceinline__ double scale_update(double num, double sum, double alpha, double shift) {
    return exp(alpha * log(num / sum) - shift);
}

// ot_func.cpp:29-40: +-inf -> +-FLT_MAX (the float constant, for doubles too)
__device__ __forceinline__ double clamp_inf(double x) {
    if (isinf(x)) return x < 0 ? -(double)FLT_MAX : (double)FLT_MAX;
    return x;
}

// ------------------------------------------------------------------------------------------
// K = exp((u_i + v_j - C_ij)/eps) (ot_func.cpp:563-567, :802-806); pad columns written as 0.
// `flag` (may be null): skip the whole launch unless *flag != 0 (device-side tau decision).
// ------------------------------------------------------------------------------------------
// exp(x) for a value that is about to be ROUNDED TO FP32 anyway (fp32 storage of K): the argument is reduced in fp64
// (x log2 e = n + f, |f| <= 1/2, so the 6e-8 relative precision of an fp32 argument never meets a magnitude of hundreds),
// 2^f comes from the hardware's fp32 exponential (v_exp_f32, ~1 ulp) and 2^n is an exponent-field add: ~10 instructions
// against ~40 fp64 ones for exp() -- k_build_K was bound by those, not by HBM (3.7 TB/s).  Relative error ~1.5e-7, the
// same order as the rounding to fp32 that follows.  Underflows to 0 like exp (x < -87.3 gives 0 in fp32 either way).
__device__ __forceinline__ float exp_to_f32(double x) {
    const double y = x * 1.4426950408889634;                   // log2(e)
    if (!(y > -150.0)) return (y != y) ? __builtin_nanf("") : 0.f;     // (also -inf: cost +inf, and NaN)
    if (y > 128.0) return __builtin_inff();
    const double n = rint(y);
    const float f = (float)(y - n);
    return ldexpf(__builtin_amdgcn_exp2f(f), (int)n);
}

template <typename T>
__global__ __launch_bounds__(256) void k_build_K(T *__restrict__ K, const T *__restrict__ C,
                                                 const double *__restrict__ u,
                                                 const double *__restrict__ v, double eps, int I,
                                                 int J, int ld, const int *flag) {
    if (flag && *flag == 0) return;
    constexpr int V = Vec<T>::N;
    using VT = typename Vec<T>::type;
    const int j = (blockIdx.x * 256 + threadIdx.x) * V;
    if (j >= ld) return;
    double vv[V];
#pragma unroll
    for (int k = 0; k < V; k++) vv[k] = (j + k < J) ? v[j + k] : 0.0;
    for (int i = blockIdx.y; i < I; i += gridDim.y) {
        const double ui = u[i];
        double c[V], o[V];
        // C is read once and K written once per launch, both larger than every cache: non-temporal
        unpack<T>(__builtin_nontemporal_load(reinterpret_cast<const VT *>(C + (size_t)i * ld + j)), c);
        if (std::is_same<T, float>::value && FAST_EXP_F32) {
#pragma unroll
            for (int k = 0; k < V; k++) o[k] = (j + k < J) ? (double)exp_to_f32((ui + vv[k] - c[k]) / eps) : 0.0;
        } else {
#pragma unroll
            for (int k = 0; k < V; k++) o[k] = (j + k < J) ? exp((ui + vv[k] - c[k]) / eps) : 0.0;
        }
        __builtin_nontemporal_store(pack<T>(o), reinterpret_cast<VT *>(K + (size_t)i * ld + j));
    }
}

// Kbar = exp(-C/eps) (ot_func.cpp:558-560), materialised only for the libot-compatible entry.
template <typename T>
__global__ __launch_bounds__(256) void k_build_Kbar(T *__restrict__ Kb, const T *__restrict__ C,
                                                    double eps, int I, int J, int ld) {
    constexpr int V = Vec<T>::N;
    using VT = typename Vec<T>::type;
    const int j = (blockIdx.x * 256 + threadIdx.x) * V;
    if (j >= ld) return;
    for (int i = blockIdx.y; i < I; i += gridDim.y) {
        double c[V], o[V];
        unpack<T>(*reinterpret_cast<const VT *>(C + (size_t)i * ld + j), c);
#pragma unroll
        for (int k = 0; k < V; k++) o[k] = (j + k < J) ? exp(-c[k] / eps) : 0.0;
        *reinterpret_cast<VT *>(Kb + (size_t)i * ld + j) = pack<T>(o);
    }
}

// Per-row sum of a matrix (MODE 0) or of exp(-C/eps) (MODE 1): out[i] = sum_j f(M_ij).
template <typename T, int MODE>
__global__ __launch_bounds__(256) void k_row_reduce(const T *__restrict__ M, double eps, int I,
                                                    int J, int ld, double *__restrict__ out) {
    constexpr int V = Vec<T>::N;
    using VT = typename Vec<T>::type;
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * ROW_WAVES + (threadIdx.x >> 6);
    if (row >= I) return;
    const T *r = M + (size_t)row * ld;
    double acc = 0.0;
    for (int j = lane * V; j < ld; j += WAVE * V) {
        double c[V];
        unpack<T>(*reinterpret_cast<const VT *>(r + j), c);
#pragma unroll
        for (int k = 0; k < V; k++)
            if (j + k < J) acc += (MODE == 1) ? exp(-c[k] / eps) : c[k];
    }
    acc = wave_sum(acc);
    if (lane == 0) out[row] = acc;
}

// out[slot] = sum_i x[i], one block, fixed order.
__global__ __launch_bounds__(1024) void k_vec_sum(const double *__restrict__ x, int n,
                                                  double *__restrict__ out, int slot) {
    __shared__ double sh[16];
    double acc = 0.0;
    for (int i = threadIdx.x; i < n; i += blockDim.x) acc += x[i];
    acc = block_sum(acc, sh);
    if (threadIdx.x == 0) out[slot] = acc;
}

// ------------------------------------------------------------------------------------------
// Row pass (ot_func.cpp:610-636): s_i = sum_j K_ij w_j with w = b.dy, then
//   old_a_i = a_i ; a_i = (p_i/s_i)^alpha1 * exp(-u_i/(lambda1+eps)) ; adx_i = a_i dx_i
// One wave per row; lanes stride the row in 16-byte pieces (1 KiB per wave-instruction).
// ------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void k_row_pass(const T *__restrict__ K,
                                                  const double *__restrict__ w,
                                                  double *__restrict__ a,
                                                  double *__restrict__ old_a,
                                                  double *__restrict__ adx,
                                                  const double *__restrict__ p,
                                                  const double *__restrict__ dx,
                                                  const double *__restrict__ u, double alpha1,
                                                  double inv_l1e, double tau, int I, int ld,
                                                  int *flag) {
    constexpr int V = Vec<T>::N;
    using VT = typename Vec<T>::type;
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * ROW_WAVES + (threadIdx.x >> 6);
    if (row >= I) return;
    const T *r = K + (size_t)row * ld;
    double acc0 = 0.0, acc1 = 0.0;
    int j = lane * V;
    // two independent 16-byte loads in flight per lane per trip
    for (; j + WAVE * V < ld; j += 2 * WAVE * V) {
        double k0[V], k1[V];
        unpack<T>(*reinterpret_cast<const VT *>(r + j), k0);
        unpack<T>(*reinterpret_cast<const VT *>(r + j + WAVE * V), k1);
#pragma unroll
        for (int k = 0; k < V; k++) {
            acc0 = fma(k0[k], w[j + k], acc0);
            acc1 = fma(k1[k], w[j + WAVE * V + k], acc1);
        }
    }
    if (j < ld) {
        double k0[V];
        unpack<T>(*reinterpret_cast<const VT *>(r + j), k0);
#pragma unroll
        for (int k = 0; k < V; k++) acc0 = fma(k0[k], w[j + k], acc0);
    }
    const double s = wave_sum(acc0 + acc1);
    if (lane == 0) {
        const double an = scale_update(p[row], s, alpha1, u[row] * inv_l1e);
        old_a[row] = a[row];
        a[row] = an;
        adx[row] = an * dx[row];
        if (an > tau) *flag = 1;    // benign race: every writer stores 1
    }
}

// ------------------------------------------------------------------------------------------
// Column pass (ot_func.cpp:643-651): part[c][j] = sum_{i in chunk c} K_ij adx_i, rows ascending.
// ------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void k_col_pass(const T *__restrict__ K,
                                                  const double *__restrict__ adx,
                                                  double *__restrict__ part, int I, int ld,
                                                  int rows_per_chunk) {
    constexpr int V = Vec<T>::N;
    using VT = typename Vec<T>::type;
    const int j = (blockIdx.x * 256 + threadIdx.x) * V;
    if (j >= ld) return;
    const int i0 = blockIdx.y * rows_per_chunk;
    const int i1 = min(I, i0 + rows_per_chunk);
    double acc[V];
#pragma unroll
    for (int k = 0; k < V; k++) acc[k] = 0.0;
    const T *base = K + j;
    int i = i0;
    for (; i + 4 <= i1; i += 4) {
        VT r0 = *reinterpret_cast<const VT *>(base + (size_t)(i + 0) * ld);
        VT r1 = *reinterpret_cast<const VT *>(base + (size_t)(i + 1) * ld);
        VT r2 = *reinterpret_cast<const VT *>(base + (size_t)(i + 2) * ld);
        VT r3 = *reinterpret_cast<const VT *>(base + (size_t)(i + 3) * ld);
        const double x0 = adx[i], x1 = adx[i + 1], x2 = adx[i + 2], x3 = adx[i + 3];
        double e0[V], e1[V], e2[V], e3[V];
        unpack<T>(r0, e0); unpack<T>(r1, e1); unpack<T>(r2, e2); unpack<T>(r3, e3);
#pragma unroll
        for (int k = 0; k < V; k++) {
            acc[k] = fma(e0[k], x0, acc[k]);
            acc[k] = fma(e1[k], x1, acc[k]);
            acc[k] = fma(e2[k], x2, acc[k]);
            acc[k] = fma(e3[k], x3, acc[k]);
        }
    }
    for (; i < i1; i++) {
        double e0[V];
        unpack<T>(*reinterpret_cast<const VT *>(base + (size_t)i * ld), e0);
        const double x0 = adx[i];
#pragma unroll
        for (int k = 0; k < V; k++) acc[k] = fma(e0[k], x0, acc[k]);
    }
    double *o = part + (size_t)blockIdx.y * ld + j;
#pragma unroll
    for (int k = 0; k < V; k++) o[k] = acc[k];
}

// t_j = sum_c part[c][j];  old_b = b;  b_j = (q_j/t_j)^alpha2 exp(-v_j/(lambda2+eps));  w_j = b_j dy_j
// (ot_func.cpp:657-668).  mode 1: only write t (used by the gap check).
__global__ __launch_bounds__(256) void k_col_fin(const double *__restrict__ part, int nchunk,
                                                 double *__restrict__ b,
                                                 double *__restrict__ old_b,
                                                 double *__restrict__ w,
                                                 const double *__restrict__ q,
                                                 const double *__restrict__ dy,
                                                 const double *__restrict__ v, double alpha2,
                                                 double inv_l2e, double tau, int J, int ld,
                                                 int *flag, double *t_out, int mode) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= ld) return;
    if (j >= J) { if (mode == 0) w[j] = 0.0; else t_out[j] = 0.0; return; }
    double t = 0.0;
    for (int c = 0; c < nchunk; c++) t += part[(size_t)c * ld + j];
    if (t_out) t_out[j] = t;
    if (mode == 1) return;
    const double bn = scale_update(q[j], t, alpha2, v[j] * inv_l2e);
    old_b[j] = b[j];
    b[j] = bn;
    w[j] = bn * dy[j];
    if (bn > tau) *flag = 1;
}


// ------------------------------------------------------------------------------------------
// Fused scaling pass: ONE sweep of K per iteration instead of two.
// A 512-thread workgroup owns a band of rows and the whole row width.  For R rows at a time the
// band's elements sit in registers (VPT 16-byte vectors per thread per row): the workgroup
//   1. dots them with w = b.dy (staged once in LDS)           -> s_i        (ot_func.cpp:610-619)
//   2. lets R threads finish a_i = (p_i/s_i)^alpha1 e^{-u_i/(lambda1+eps)}   (ot_func.cpp:633-636)
//   3. adds K_ij * a_i dx_i into per-thread fp64 column accumulators        (ot_func.cpp:643-651)
// while the next R rows are already in flight into the second register buffer.  Column partials go
// to part[block][j]; k_col_fin2 sums them in block order (deterministic) and updates b.
// Algorithmic HBM bytes per launch: I*ld*sizeof(T) (+ nblk*ld*8 partials).
// ------------------------------------------------------------------------------------------
constexpr int FUSED_THREADS = 512;   // 8 waves = 2 per SIMD: one workgroup per CU may use 256 VGPRs

template <typename T, int VPT, int R>
struct FusedRows {
    using VT = typename Vec<T>::type;
    VT v[R][VPT];
};

// Unconditional loads: rows past the band re-read its last row (their a.dx weight is forced to 0) and
// vector slots past ld re-read the row's last vector (their w entries in LDS are 0 and their column
// accumulators are never stored), so the sweep has no divergent branches.
template <typename T, int VPT, int R>
__device__ __forceinline__ void fused_load(FusedRows<T, VPT, R> &buf, const T *__restrict__ K, int row0,
                                           int row_end, int ld, int tid) {
    constexpr int V = Vec<T>::N;
    using VT = typename Vec<T>::type;
#pragma unroll
    for (int r = 0; r < R; r++) {
        const int row = min(row0 + r, row_end - 1);
        const T *base = K + (size_t)row * ld;
#pragma unroll
        for (int k = 0; k < VPT; k++) {
            const int j = min((tid + k * FUSED_THREADS) * V, ld - V);
            // K is streamed exactly once per launch and is larger than every cache: non-temporal
            buf.v[r][k] = __builtin_nontemporal_load(reinterpret_cast<const VT *>(base + j));
        }
    }
}

template <typename T, int VPT, int R, typename WT, int PAR, bool WRS = false>
__device__ __forceinline__ void fused_group(FusedRows<T, VPT, R> &buf, int row0, int row_end,
                                            const WT *wl, double *red,
                                            double *colacc, double *__restrict__ a,
                                            double *__restrict__ old_a, double *__restrict__ adx,
                                            const double *rowc, int band0, int band_rows, double alpha1,
                                            double inv_l1e, double tau, int ld, int *flag, double *__restrict__ rs_out = nullptr) {
    constexpr int V = Vec<T>::N;
    constexpr int NW = FUSED_THREADS / 64;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    double s[R];
#pragma unroll
    for (int r = 0; r < R; r++) s[r] = 0.0;
#pragma unroll
    for (int k = 0; k < VPT; k++) {
        const int j = (tid + k * FUSED_THREADS) * V;
        double wv[V];
#pragma unroll
        for (int e = 0; e < V; e++) wv[e] = (double)wl[j + e];     // 0 beyond ld (LDS image is padded)
#pragma unroll
        for (int r = 0; r < R; r++) {
            double kv[V];
            unpack<T>(buf.v[r][k], kv);
#pragma unroll
            for (int e = 0; e < V; e++) s[r] = fma(kv[e], wv[e], s[r]);
        }
        __builtin_amdgcn_sched_barrier(0);     // one vector slot at a time: bounds the live fp64 temporaries
    }
    // keep the band in its storage type across the barrier: without this the compiler keeps the
    // widened fp64 copies of an fp32 band live for the accumulate phase (2x the registers)
#pragma unroll
    for (int r = 0; r < R; r++)
#pragma unroll
        for (int k = 0; k < VPT; k++) asm volatile("" : "+v"(buf.v[r][k]));
    // wave partials -> LDS (two images, used by alternate groups: ONE barrier per group is enough, a wave that runs
    // ahead into the next group writes the other image)
    double *redp = red + PAR * (NW * R);
#pragma unroll
    for (int r = 0; r < R; r++) {
        s[r] = wave_sum63(s[r]);
        if (lane == 63) redp[wid * R + r] = s[r];
    }
    __syncthreads();
    // every wave finishes a_i itself (lane l takes row l mod R: same latency as one thread doing it, but no second
    // barrier and no LDS round trip for the result); wave 0 stores.  p, dx, u and the previous a of the band were
    // staged in LDS at kernel entry: a global load here would sit behind the whole prefetch in the vmcnt queue.
    const int lr = lane & (R - 1);
    const int row = row0 + lr;
    double x = 0.0;
    if (row < row_end) {
        double t = 0.0;
#pragma unroll
        for (int wv = 0; wv < NW; wv++) t += redp[wv * R + lr];
        const int br = row - band0;
        const double an = scale_update(rowc[br], t, alpha1, rowc[2 * band_rows + br] * inv_l1e);
        x = an * rowc[band_rows + br];
        if (tid < R) {
            old_a[row] = rowc[3 * band_rows + br];
            a[row] = an;
            adx[row] = x;
            if (an > tau) *flag = 1;
            if (WRS) rs_out[row] = t;          // sum_j K_ij w_j with the w this pass started from: the gap measure's row sums
        }
    }
#pragma unroll
    for (int r = 0; r < R; r++) {
        const double xr = read_lane(x, r);
#pragma unroll
        for (int k = 0; k < VPT; k++) {
            double kv[V];
            unpack<T>(buf.v[r][k], kv);
#pragma unroll
            for (int e = 0; e < V; e++) colacc[k * V + e] = fma(kv[e], xr, colacc[k * V + e]);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

// WRS = true (the first pass of a batch in the pipelined last-stage loop): the row sums s_i = sum_j K_ij w_j the pass
// computes anyway are also stored to rs_out -- they are exactly what the duality-gap measure of the state the pass
// STARTED from needs, so that measure costs no sweep of its own (process_stage).
template <typename T, int VPT, int R, typename WT = double, bool WRS = false>
__global__ __launch_bounds__(FUSED_THREADS) void k_fused_pass(
    const T *__restrict__ K, const double *__restrict__ w, double *__restrict__ a,
    double *__restrict__ old_a, double *__restrict__ adx, const double *__restrict__ p,
    const double *__restrict__ dx, const double *__restrict__ u, double alpha1, double inv_l1e,
    double tau, T *__restrict__ part, int I, int ld, int rows_per_block, int *flag, const int *__restrict__ stop,
    double *__restrict__ rs_out) {
    if (stop != nullptr && *stop != 0) return;       // a speculatively enqueued batch behind the converged one: nothing to do
    constexpr int V = Vec<T>::N;
    extern __shared__ double smem[];
    constexpr int WPAD = VPT * FUSED_THREADS * V;    // w image padded with zeros to the register tile
    WT *wl = reinterpret_cast<WT *>(smem);   // WPAD entries of WT (double, or float for the widest fp32 rows)
    double *red = smem + (WPAD * sizeof(WT) + 7) / 8;   // 2 images of (FUSED_THREADS/64) * R wave partials
    double *rowc = red + 2 * (FUSED_THREADS / 64) * R;  // 4 x rows_per_block: p, dx, u, previous a
    const int tid = threadIdx.x;
    const int r0 = blockIdx.x * rows_per_block;
    const int r1 = min(I, r0 + rows_per_block);
    FusedRows<T, VPT, R> bufA, bufB;
    // Request order = arrival order (one in-order queue per wave): first the w image and the band's row constants
    // (small, cache hits), then both register buffers of K.  The LDS images are filled while K is on its way; the
    // other way round they would sit behind 160 KB of K per workgroup.
    constexpr int WSLOTS = WPAD / FUSED_THREADS;      // VPT * V
    double wreg[WSLOTS];
#pragma unroll
    for (int q = 0; q < WSLOTS; q++) {
        const int j = tid + q * FUSED_THREADS;
        wreg[q] = (j < ld) ? w[j] : 0.0;
    }
    double rc[4] = {0.0, 0.0, 0.0, 0.0};
    const bool has_row = tid < r1 - r0;
    if (has_row) { rc[0] = p[r0 + tid]; rc[1] = dx[r0 + tid]; rc[2] = u[r0 + tid]; rc[3] = a[r0 + tid]; }
    fused_load<T, VPT, R>(bufA, K, r0, r1, ld, tid);
    fused_load<T, VPT, R>(bufB, K, r0 + R, r1, ld, tid);            // rows >= r1: clamped, weight 0
#pragma unroll
    for (int q = 0; q < WSLOTS; q++) wl[tid + q * FUSED_THREADS] = (WT)wreg[q];
    if (has_row) {
        rowc[tid] = rc[0];
        rowc[rows_per_block + tid] = rc[1];
        rowc[2 * rows_per_block + tid] = rc[2];
        rowc[3 * rows_per_block + tid] = rc[3];
    }
    for (int t = tid + FUSED_THREADS; t < r1 - r0; t += FUSED_THREADS) {      // bands longer than the workgroup (rare)
        rowc[t] = p[r0 + t];
        rowc[rows_per_block + t] = dx[r0 + t];
        rowc[2 * rows_per_block + t] = u[r0 + t];
        rowc[3 * rows_per_block + t] = a[r0 + t];
    }
    double colacc[VPT * V];
#pragma unroll
    for (int k = 0; k < VPT * V; k++) colacc[k] = 0.0;
    __syncthreads();
#define FUSED_GROUP(PAR, BUF, G) fused_group<T, VPT, R, WT, PAR, WRS>(BUF, G, r1, wl, red, colacc, a, old_a, adx, rowc, r0, \
                                                                    rows_per_block, alpha1, inv_l1e, tau, ld, flag, rs_out)
    int g = r0;
    // steady state: both loads unconditional (a conditional load would make the compiler wait for ALL outstanding
    // loads at the join, i.e. undo the double buffering); the loop ends while both buffers still hold valid rows
    for (; g + 3 * R < r1; g += 2 * R) {
        FUSED_GROUP(0, bufA, g);
        fused_load<T, VPT, R>(bufA, K, g + 2 * R, r1, ld, tid);
        FUSED_GROUP(1, bufB, g + R);
        fused_load<T, VPT, R>(bufB, K, g + 3 * R, r1, ld, tid);      // last round: a partial group, rows clamped
    }
    // tail: one to three groups left, nothing is requested beyond the band
    FUSED_GROUP(0, bufA, g);
    if (g + R < r1) {
        const bool third = g + 2 * R < r1;
        if (third) fused_load<T, VPT, R>(bufA, K, g + 2 * R, r1, ld, tid);
        FUSED_GROUP(1, bufB, g + R);
        if (third) FUSED_GROUP(0, bufA, g + 2 * R);
    }
#undef FUSED_GROUP
    T *o = part + (size_t)blockIdx.x * ld;        // partials in the storage type: fp32 K => fp32 partials
#pragma unroll
    for (int k = 0; k < VPT; k++) {
        const int j = (tid + k * FUSED_THREADS) * V;
        if (j < ld) {
#pragma unroll
            for (int e = 0; e < V; e++) o[j + e] = (T)colacc[k * V + e];
        }
    }
}

// Row dots only, on the fused pass's machinery (same band geometry, w image in LDS, double-buffered register tiles,
// DPP wave sums): sdot_i = sum_j K_ij w_j for the solver's duality-gap check, which needs nothing else from K.
// One wave per row with w read from global (k_row_dot) streams at 5.0 TB/s; this form at the fused pass's rate.
template <typename T, int VPT, int R, typename WT = double>
__global__ __launch_bounds__(FUSED_THREADS) void k_fused_rowdot(const T *__restrict__ K, const double *__restrict__ w,
                                                                double *__restrict__ sdot, int I, int ld,
                                                                int rows_per_block, const int *__restrict__ stop) {
    if (stop != nullptr && *stop != 0) return;
    constexpr int V = Vec<T>::N;
    constexpr int NW = FUSED_THREADS / 64;
    extern __shared__ double smem[];
    constexpr int WPAD = VPT * FUSED_THREADS * V;
    WT *wl = reinterpret_cast<WT *>(smem);
    double *red = smem + (WPAD * sizeof(WT) + 7) / 8;            // 2 images of NW * R wave partials
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int r0 = blockIdx.x * rows_per_block;
    const int r1 = min(I, r0 + rows_per_block);
    FusedRows<T, VPT, R> bufA, bufB;
    constexpr int WSLOTS = WPAD / FUSED_THREADS;
    double wreg[WSLOTS];
#pragma unroll
    for (int q = 0; q < WSLOTS; q++) {
        const int j = tid + q * FUSED_THREADS;
        wreg[q] = (j < ld) ? w[j] : 0.0;
    }
    fused_load<T, VPT, R>(bufA, K, r0, r1, ld, tid);
    fused_load<T, VPT, R>(bufB, K, r0 + R, r1, ld, tid);
#pragma unroll
    for (int q = 0; q < WSLOTS; q++) wl[tid + q * FUSED_THREADS] = (WT)wreg[q];
    __syncthreads();
    auto group = [&](FusedRows<T, VPT, R> &buf, int row0, int par) {
        double s[R];
#pragma unroll
        for (int r = 0; r < R; r++) s[r] = 0.0;
#pragma unroll
        for (int k = 0; k < VPT; k++) {
            const int j = (tid + k * FUSED_THREADS) * V;
            double wv[V];
#pragma unroll
            for (int e = 0; e < V; e++) wv[e] = (double)wl[j + e];
#pragma unroll
            for (int r = 0; r < R; r++) {
                double kv[V];
                unpack<T>(buf.v[r][k], kv);
#pragma unroll
                for (int e = 0; e < V; e++) s[r] = fma(kv[e], wv[e], s[r]);
            }
        }
        double *redp = red + par * (NW * R);
#pragma unroll
        for (int r = 0; r < R; r++) {
            s[r] = wave_sum63(s[r]);
            if (lane == 63) redp[wid * R + r] = s[r];
        }
        __syncthreads();
        if (tid < R && row0 + tid < r1) {
            double t = 0.0;
#pragma unroll
            for (int wv = 0; wv < NW; wv++) t += redp[wv * R + tid];
            sdot[row0 + tid] = t;
        }
    };
    int g = r0;
    for (; g + 3 * R < r1; g += 2 * R) {          // as k_fused_pass: unconditional loads in the steady state
        group(bufA, g, 0);
        fused_load<T, VPT, R>(bufA, K, g + 2 * R, r1, ld, tid);
        group(bufB, g + R, 1);
        fused_load<T, VPT, R>(bufB, K, g + 3 * R, r1, ld, tid);
    }
    group(bufA, g, 0);
    if (g + R < r1) {
        const bool third = g + 2 * R < r1;
        if (third) fused_load<T, VPT, R>(bufA, K, g + 2 * R, r1, ld, tid);
        group(bufB, g + R, 1);
        if (third) group(bufA, g + 2 * R, 0);
    }
}

// Column finalise for many partial rows: 64 columns x 16 partial-groups per block; partials are
// summed in ascending block order within a group and groups in ascending order (deterministic).
// mode 0: b update (ot_func.cpp:657-668) ; mode 1: t_out[j] = sum only.
template <typename PT>
__global__ __launch_bounds__(1024) void k_col_fin2(const PT *__restrict__ part, int npart,
                                                   double *__restrict__ b,
                                                   double *__restrict__ old_b,
                                                   double *__restrict__ w,
                                                   const double *__restrict__ q,
                                                   const double *__restrict__ dy,
                                                   const double *__restrict__ v, double alpha2,
                                                   double inv_l2e, double tau, int J, int ld,
                                                   int *flag, double *t_out, int mode, const int *__restrict__ stop) {
    if (stop != nullptr && *stop != 0) return;
    __shared__ double sh[16][65];
    const int cx = threadIdx.x & 63, gy = threadIdx.x >> 6;
    const int j = blockIdx.x * 64 + cx;
    double t = 0.0;
    if (j < ld) {
        const int per = (npart + 15) / 16;
        const int c0 = gy * per, c1 = min(npart, c0 + per);
        for (int c = c0; c < c1; c++) t += (double)part[(size_t)c * ld + j];
    }
    sh[gy][cx] = t;
    __syncthreads();
    if (gy != 0 || j >= ld) return;
    t = 0.0;
#pragma unroll
    for (int g = 0; g < 16; g++) t += sh[g][cx];
    if (j >= J) { if (mode == 0) w[j] = 0.0; else t_out[j] = 0.0; return; }
    if (t_out) t_out[j] = t;
    if (mode == 1) return;
    const double bn = scale_update(q[j], t, alpha2, v[j] * inv_l2e);
    old_b[j] = b[j];
    b[j] = bn;
    w[j] = bn * dy[j];
    if (bn > tau) *flag = 1;
}

// tau-absorb, vector half (ot_func.cpp:792-814): runs only when *flag is set.
__global__ __launch_bounds__(256) void k_absorb_vec(double *__restrict__ a, double *__restrict__ b,
                                                    double *__restrict__ u, double *__restrict__ v,
                                                    double *__restrict__ adx,
                                                    double *__restrict__ w,
                                                    const double *__restrict__ dx,
                                                    const double *__restrict__ dy, double eps, int I,
                                                    int J, const int *flag, int *absorb_count,
                                                    double *__restrict__ tcol) {
    if (*flag == 0) return;
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t < I) { u[t] = u[t] + eps * log(a[t]); a[t] = 1.0; adx[t] = dx[t]; }
    // t_j = sum_i K_ij a_i dx_i is kept for the gap check: with K_new = a K b and a_new = 1 it becomes b_j t_j
    if (t < J) { v[t] = v[t] + eps * log(b[t]); tcol[t] = b[t] * tcol[t]; b[t] = 1.0; w[t] = dy[t]; }
    if (t == 0) *absorb_count += 1;
}

// Stage transition (ot_solvers.py:249-260): absorb unconditionally, reset old_a/old_b too.
__global__ __launch_bounds__(256) void k_stage_begin(double *a, double *b, double *old_a,
                                                     double *old_b, double *u, double *v,
                                                     double *adx, double *w, const double *dx,
                                                     const double *dy, double eps_prev, int I, int J,
                                                     int ld) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t < I) { u[t] = u[t] + eps_prev * log(a[t]); a[t] = 1.0; old_a[t] = 1.0; adx[t] = dx[t]; }
    if (t < J) { v[t] = v[t] + eps_prev * log(b[t]); b[t] = 1.0; old_b[t] = 1.0; w[t] = dy[t]; }
    else if (t < ld) w[t] = 0.0;
}

// w = b.dy (pad 0), adx = a.dx -- used when a, b come from the caller (compat entry points).
__global__ __launch_bounds__(256) void k_prep_vec(const double *a, const double *b, const double *dx,
                                                  const double *dy, double *adx, double *w, int I,
                                                  int J, int ld) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t < I) adx[t] = a[t] * dx[t];
    if (t < J) w[t] = b[t] * dy[t];
    else if (t < ld) w[t] = 0.0;
}

// Dual-variable drift (ot_func.cpp:876-923, stages 0..4).  One block.
//   ta = a e^{u/eps};  g1 = ||ta - old_a e^{u/eps}|| / (1 + ||ta||); likewise g2;  gap = std::max(g1,g2)
__global__ __launch_bounds__(1024) void k_drift(const double *__restrict__ a,
                                                const double *__restrict__ old_a,
                                                const double *__restrict__ u,
                                                const double *__restrict__ b,
                                                const double *__restrict__ old_b,
                                                const double *__restrict__ v, double eps, int I,
                                                int J, double *scal) {
    __shared__ double sh[16];
    double d1 = 0, n1 = 0, d2 = 0, n2 = 0;
    for (int i = threadIdx.x; i < I; i += blockDim.x) {
        const double e = exp(u[i] / eps);
        const double ta = a[i] * e;
        const double t = ta - old_a[i] * e;
        d1 += t * t;
        n1 += ta * ta;
    }
    for (int j = threadIdx.x; j < J; j += blockDim.x) {
        const double e = exp(v[j] / eps);
        const double tb = b[j] * e;
        const double t = tb - old_b[j] * e;
        d2 += t * t;
        n2 += tb * tb;
    }
    d1 = block_sum(d1, sh); n1 = block_sum(n1, sh);
    d2 = block_sum(d2, sh); n2 = block_sum(n2, sh);
    if (threadIdx.x == 0) {
        const double g1 = sqrt(d1) / (1.0 + sqrt(n1));
        const double g2 = sqrt(d2) / (1.0 + sqrt(n2));
        scal[0] = (g1 < g2) ? g2 : g1;   // std::max(g1, g2): a NaN g1 wins (ot_func.cpp:922)
    }
}

// ------------------------------------------------------------------------------------------
// Duality-gap row pass (ot_func.cpp:392-428, :481-483 folded into ONE sweep of K and C):
// per row i with R_ij = K_ij a_i b_j (FROM_K) or R_ij read from a matrix:
//   rt[0][i] = sum_j R dy_j     rt[1][i] = sum_j R clamp(ln R) - R
//   rt[2][i] = sum_j R C_ij     rt[3][i] = sum_j R
// and optionally R is written out (update_R, ot_func.cpp:570-584).
// ------------------------------------------------------------------------------------------
template <typename T, bool FROM_K>
__global__ __launch_bounds__(256) void k_gap_rows(const T *__restrict__ KorR,
                                                  const T *__restrict__ C,
                                                  const double *__restrict__ a,
                                                  const double *__restrict__ b,
                                                  const double *__restrict__ dy,
                                                  T *__restrict__ Rout, double *__restrict__ rt,
                                                  int I, int J, int ld) {
    constexpr int V = Vec<T>::N;
    using VT = typename Vec<T>::type;
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * ROW_WAVES + (threadIdx.x >> 6);
    if (row >= I) return;
    const T *kr = KorR + (size_t)row * ld;
    const T *cr = C ? C + (size_t)row * ld : nullptr;
    const double ai = FROM_K ? a[row] : 1.0;
    double rs = 0, en = 0, co = 0, sr = 0;
    for (int j = lane * V; j < ld; j += WAVE * V) {
        double kv[V], cv[V], ro[V];
        unpack<T>(*reinterpret_cast<const VT *>(kr + j), kv);
        if (cr) unpack<T>(*reinterpret_cast<const VT *>(cr + j), cv);
#pragma unroll
        for (int k = 0; k < V; k++) {
            double R = 0.0;
            if (j + k < J) {
                R = FROM_K ? kv[k] * ai * b[j + k] : kv[k];
                rs += R * dy[j + k];
                en += R * clamp_inf(log(R)) - R;
                if (cr) co += R * cv[k];
                sr += R;
            }
            ro[k] = R;
        }
        if (Rout) *reinterpret_cast<VT *>(Rout + (size_t)row * ld + j) = pack<T>(ro);
    }
    rs = wave_sum(rs); en = wave_sum(en); co = wave_sum(co); sr = wave_sum(sr);
    if (lane == 0) {
        rt[row] = rs; rt[(size_t)I + row] = en; rt[2 * (size_t)I + row] = co; rt[3 * (size_t)I + row] = sr;
    }
}

// Gap finalisation, one block (ot_func.cpp:308-322, :340-355, :432-434, :485-489, :543).
//   cs_j = col_scale ? b_j * t_j : t_j      (t = column pass with weights a.dx, or R^T dx directly)
//   abar/bbar = a e^{u/eps}, b e^{v/eps} when `stabilised`, else a, b as given
// scal[0]=gap scal[1]=primal scal[2]=dual.  scal[3] holds sum(Kbar) on entry.
__global__ __launch_bounds__(1024) void k_gap_fin(const double *__restrict__ rt,
                                                  const double *__restrict__ t,
                                                  const double *__restrict__ a,
                                                  const double *__restrict__ b,
                                                  const double *__restrict__ u,
                                                  const double *__restrict__ v,
                                                  const double *__restrict__ p,
                                                  const double *__restrict__ q,
                                                  const double *__restrict__ dx,
                                                  const double *__restrict__ dy, double eps,
                                                  double l1, double l2, int I, int J, int col_scale,
                                                  int stabilised, double *scal) {
    __shared__ double sh[16];
    double f1 = 0, en = 0, co = 0, sr = 0, c1 = 0, f2 = 0, c2 = 0;
    for (int i = threadIdx.x; i < I; i += blockDim.x) {
        const double x = rt[i];
        f1 += dx[i] * (x * log(x / p[i]) - x + p[i]);
        en += rt[(size_t)I + i];
        co += rt[2 * (size_t)I + i];
        sr += rt[3 * (size_t)I + i];
        const double ab = stabilised ? a[i] * exp(u[i] / eps) : a[i];
        c1 += (p[i] * dx[i]) * (exp((-eps * log(ab)) / l1) - 1.0);
    }
    for (int j = threadIdx.x; j < J; j += blockDim.x) {
        const double x = col_scale ? b[j] * t[j] : t[j];
        f2 += dy[j] * (x * log(x / q[j]) - x + q[j]);
        const double bb = stabilised ? b[j] * exp(v[j] / eps) : b[j];
        c2 += (q[j] * dy[j]) * (exp((-eps * log(bb)) / l2) - 1.0);
    }
    f1 = block_sum(f1, sh); en = block_sum(en, sh); co = block_sum(co, sh); sr = block_sum(sr, sh);
    c1 = block_sum(c1, sh); f2 = block_sum(f2, sh); c2 = block_sum(c2, sh);
    if (threadIdx.x == 0) {
        const double skb = scal[3];
        const double mn = (double)I * (double)J;
        const double pri = l1 * f1 + l2 * f2 + (eps * (en + skb) + co) / mn;
        const double dua = -(l1 * c1) - (l2 * c2) - eps * (sr - skb) / mn;
        scal[1] = pri;
        scal[2] = dua;
        scal[0] = (pri - dua) / fabs(pri);
    }
}


// Measure-only sweep (last-stage convergence check): sdot_i = sum_j K_ij w_j with the CURRENT w = b.dy.
// One wave per row, non-temporal 16-byte loads.  Together with t_j = sum_i K_ij a_i dx_i (left behind by
// the last column finalise) this is all the duality gap needs: see k_gap2_part.
template <typename T>
__global__ __launch_bounds__(256) void k_row_dot(const T *__restrict__ K, const double *__restrict__ w,
                                                 double *__restrict__ sdot, int I, int ld) {
    constexpr int V = Vec<T>::N;
    using VT = typename Vec<T>::type;
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * ROW_WAVES + (threadIdx.x >> 6);
    if (row >= I) return;
    const T *r = K + (size_t)row * ld;
    double acc0 = 0.0, acc1 = 0.0;
    int j = lane * V;
    for (; j + WAVE * V < ld; j += 2 * WAVE * V) {
        double k0[V], k1[V];
        unpack<T>(__builtin_nontemporal_load(reinterpret_cast<const VT *>(r + j)), k0);
        unpack<T>(__builtin_nontemporal_load(reinterpret_cast<const VT *>(r + j + WAVE * V)), k1);
#pragma unroll
        for (int k = 0; k < V; k++) {
            acc0 = fma(k0[k], w[j + k], acc0);
            acc1 = fma(k1[k], w[j + WAVE * V + k], acc1);
        }
    }
    if (j < ld) {
        double k0[V];
        unpack<T>(__builtin_nontemporal_load(reinterpret_cast<const VT *>(r + j)), k0);
#pragma unroll
        for (int k = 0; k < V; k++) acc0 = fma(k0[k], w[j + k], acc0);
    }
    const double sm = wave_sum(acc0 + acc1);
    if (lane == 0) sdot[row] = sm;
}

// Device-side stop decision of a speculatively enqueued batch (ctl = {stop, converged batch + 1, tau batch + 1}): the batch
// whose measure is no longer above the threshold -- or that raised the tau flag, which sends the host back to that batch's
// snapshot -- sets ctl[0]; every kernel of the batches enqueued behind it sees it and returns at once.
__device__ __forceinline__ void decide_stop(double measure, double threshold, int batch, int *ctl, const int *tauflag) {
    if (ctl == nullptr) return;
    if (tauflag != nullptr && *tauflag != 0) { ctl[2] = batch + 1; ctl[0] = 1; }
    else if (!(measure > threshold)) { ctl[1] = batch + 1; ctl[0] = 1; }
}

// Duality gap from vectors only (uniform dx = 1/I, dy = 1/J as the solver always has, ot_solvers.py:221).
// With R_ij = a_i K_ij b_j and K_ij = exp((u_i + v_j - C_ij)/eps) the matrix part of the primal collapses:
//   eps sum R ln R + sum R C = sum_ij R_ij (eps ln a_i + u_i + eps ln b_j + v_j)
//                            = sum_i (eps ln a_i + u_i) rowsum_i + sum_j (eps ln b_j + v_j) colsum_j
// (the per-element ln K_ij cancels against C_ij; entries with R_ij = 0 contribute 0 on both sides,
// which is also what the reference's log(0) clamp gives, ot_func.cpp:29-40,:414).  So the check needs
// rs_i = a_i sdot_i (= sum_j R_ij dy_j) and cs_j = b_j t_j (= sum_i R_ij dx_i) only.
// Block partials -> red[blk*8 + {0..6}] = F1, F2, matrix, sumR, conj1, conj2.
__global__ __launch_bounds__(256) void k_gap2_part(const double *__restrict__ sdot,
                                                   const double *__restrict__ t,
                                                   const double *__restrict__ a,
                                                   const double *__restrict__ b,
                                                   const double *__restrict__ u,
                                                   const double *__restrict__ v,
                                                   const double *__restrict__ p,
                                                   const double *__restrict__ q, double eps, double l1,
                                                   double l2, int I, int J, double *__restrict__ red,
                                                   const int *__restrict__ stop) {
    if (stop != nullptr && *stop != 0) return;
    __shared__ double sh[16];
    const int e = blockIdx.x * 256 + threadIdx.x;
    const double dx = 1.0 / I, dy = 1.0 / J;
    double f1 = 0, f2 = 0, mat = 0, sr = 0, c1 = 0, c2 = 0;
    if (e < I) {
        const double rs = a[e] * sdot[e];                  // sum_j R_ij dy_j
        f1 = dx * (rs * log(rs / p[e]) - rs + p[e]);
        const double rsum = rs * (double)J;                // sum_j R_ij
        if (rsum != 0.0) mat += (eps * log(a[e]) + u[e]) * rsum;
        sr = rsum;
        const double ab = a[e] * exp(u[e] / eps);
        c1 = (p[e] * dx) * (exp((-eps * log(ab)) / l1) - 1.0);
    }
    if (e < J) {
        const double cs = b[e] * t[e];                     // sum_i R_ij dx_i
        f2 = dy * (cs * log(cs / q[e]) - cs + q[e]);
        const double csum = cs * (double)I;
        if (csum != 0.0) mat += (eps * log(b[e]) + v[e]) * csum;
        const double bb = b[e] * exp(v[e] / eps);
        c2 = (q[e] * dy) * (exp((-eps * log(bb)) / l2) - 1.0);
    }
    f1 = block_sum(f1, sh); f2 = block_sum(f2, sh); mat = block_sum(mat, sh); sr = block_sum(sr, sh);
    c1 = block_sum(c1, sh); c2 = block_sum(c2, sh);
    if (threadIdx.x == 0) {
        double *o = red + (size_t)blockIdx.x * 8;
        o[0] = f1; o[1] = f2; o[2] = mat; o[3] = sr; o[4] = c1; o[5] = c2;
    }
}
__global__ __launch_bounds__(256) void k_gap2_final(const double *__restrict__ red, int nblk, double eps,
                                                    double l1, double l2, int I, int J, double *scal, double threshold,
                                                    int batch, int *ctl, const int *tauflag) {
    if (ctl != nullptr && ctl[0] != 0) return;
    __shared__ double sh[16];
    double acc[6] = {0, 0, 0, 0, 0, 0};
    for (int k = threadIdx.x; k < nblk; k += 256)
#pragma unroll
        for (int c = 0; c < 6; c++) acc[c] += red[(size_t)k * 8 + c];
#pragma unroll
    for (int c = 0; c < 6; c++) acc[c] = block_sum(acc[c], sh);
    if (threadIdx.x == 0) {
        const double skb = scal[3], mn = (double)I * (double)J;
        const double pri = l1 * acc[0] + l2 * acc[1] + (acc[2] - eps * acc[3] + eps * skb) / mn;
        const double dua = -(l1 * acc[4]) - (l2 * acc[5]) - eps * (acc[3] - skb) / mn;
        scal[1] = pri; scal[2] = dua;
        const double m = (pri - dua) / fabs(pri);
        scal[0] = m;
        decide_stop(m, threshold, batch, ctl, tauflag);
        return;
    }
}

// Dual-variable drift (ot_func.cpp:876-923) as block partials + a one-block finish.
__global__ __launch_bounds__(256) void k_drift_part(const double *__restrict__ a,
                                                    const double *__restrict__ old_a,
                                                    const double *__restrict__ u,
                                                    const double *__restrict__ b,
                                                    const double *__restrict__ old_b,
                                                    const double *__restrict__ v, double eps, int I, int J,
                                                    double *__restrict__ red, const int *__restrict__ stop) {
    if (stop != nullptr && *stop != 0) return;
    __shared__ double sh[16];
    const int e = blockIdx.x * 256 + threadIdx.x;
    double d1 = 0, n1 = 0, d2 = 0, n2 = 0;
    if (e < I) {
        const double ex = exp(u[e] / eps), ta = a[e] * ex, df = ta - old_a[e] * ex;
        d1 = df * df; n1 = ta * ta;
    }
    if (e < J) {
        const double ex = exp(v[e] / eps), tb = b[e] * ex, df = tb - old_b[e] * ex;
        d2 = df * df; n2 = tb * tb;
    }
    d1 = block_sum(d1, sh); n1 = block_sum(n1, sh); d2 = block_sum(d2, sh); n2 = block_sum(n2, sh);
    if (threadIdx.x == 0) {
        double *o = red + (size_t)blockIdx.x * 8;
        o[0] = d1; o[1] = n1; o[2] = d2; o[3] = n2;
    }
}
__global__ __launch_bounds__(256) void k_drift_final(const double *__restrict__ red, int nblk, double *scal, double threshold,
                                                     int batch, int *ctl, const int *tauflag) {
    if (ctl != nullptr && ctl[0] != 0) return;
    __shared__ double sh[16];
    double acc[4] = {0, 0, 0, 0};
    for (int k = threadIdx.x; k < nblk; k += 256)
#pragma unroll
        for (int c = 0; c < 4; c++) acc[c] += red[(size_t)k * 8 + c];
#pragma unroll
    for (int c = 0; c < 4; c++) acc[c] = block_sum(acc[c], sh);
    if (threadIdx.x == 0) {
        const double g1 = sqrt(acc[0]) / (1.0 + sqrt(acc[1]));
        const double g2 = sqrt(acc[2]) / (1.0 + sqrt(acc[3]));
        const double m = (g1 < g2) ? g2 : g1;   // std::max(g1, g2): a NaN g1 wins (ot_func.cpp:922)
        scal[0] = m;
        decide_stop(m, threshold, batch, ctl, tauflag);
    }
}


// Column-group sums of the transport plan without materialising it: Q[i][g] = sum_{j: label_j = g} a_i K_ij b_j * scale.
// One wave per row; every lane keeps its own accumulator per group in LDS (acc[g][lane], conflict-free and
// summed in a fixed order => deterministic).  With one-hot(row labels)^T Q this gives the cluster-by-cluster
// transition table the analyze stage reads off the spot-level plan (_analyze_utils.py:131-137).
template <typename T>
__global__ __launch_bounds__(256) void k_plan_group_sums(const T *__restrict__ K, const double *__restrict__ a,
                                                         const double *__restrict__ b,
                                                         const int *__restrict__ labels, int ngroups,
                                                         double scale, double *__restrict__ Q, int I, int J,
                                                         int ld) {
    extern __shared__ double gacc[];            // ROW_WAVES x ngroups x 64
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int row = blockIdx.x * ROW_WAVES + wid;
    double *acc = gacc + (size_t)wid * ngroups * WAVE;
    for (int g = 0; g < ngroups; g++) acc[g * WAVE + lane] = 0.0;
    if (row < I) {
        const T *r = K + (size_t)row * ld;
        for (int j = lane; j < J; j += WAVE) {
            const int g = labels[j];
            acc[g * WAVE + lane] += (double)r[j] * b[j];
        }
        const double ai = a[row] * scale;
        for (int g = 0; g < ngroups; g++) {
            const double t = wave_sum(acc[g * WAVE + lane]);
            if (lane == 0) Q[(size_t)row * ngroups + g] = t * ai;
        }
    }
}

// plan = a_i K_ij b_j * scale  (ot_solvers.py:449 with scale = 1/J)
template <typename T, typename TO>
__global__ __launch_bounds__(256) void k_plan(const T *__restrict__ K, const double *__restrict__ a,
                                              const double *__restrict__ b, double scale,
                                              TO *__restrict__ out, int I, int J, int ld, int ldo) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= J) return;
    const double bj = b[j];
    for (int i = blockIdx.y; i < I; i += gridDim.y)
        out[(size_t)i * ldo + j] = (TO)((double)K[(size_t)i * ld + j] * a[i] * bj * scale);
}

// dst (I x ldd, TD) <- src (I x lds, TS), columns < J; pad columns of dst <- 0
template <typename TS, typename TD>
__global__ __launch_bounds__(256) void k_convert(const TS *__restrict__ src, int lds,
                                                 TD *__restrict__ dst, int ldd, int I, int J) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= ldd) return;
    for (int i = blockIdx.y; i < I; i += gridDim.y)
        dst[(size_t)i * ldd + j] = (j < J) ? (TD)src[(size_t)i * lds + j] : (TD)0;
}


// ------------------------------------------------------------------------------------------
// Cost from latents (ot_solvers.py:101-103): D_ij = max(|x_i|^2 + |y_j|^2 - 2 x_i.y_j, 0), the
// arithmetic of sklearn's euclidean_distances(squared=True); then C = D / median(D).
// One thread owns one column j (y_j lives in registers), rows come in through uniform loads,
// so the only HBM stream is the coalesced write of D.
// ------------------------------------------------------------------------------------------
constexpr int MAX_LATENT_DIM = 32;

__global__ __launch_bounds__(256) void k_sqeuclid(const double *__restrict__ x,
                                                  const double *__restrict__ y, int d, int I, int J,
                                                  double *__restrict__ D, int rows_per_block) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    const bool live = j < J;
    double yj[MAX_LATENT_DIM];
    double yy = 0.0;
#pragma unroll
    for (int k = 0; k < MAX_LATENT_DIM; k++) {
        yj[k] = (live && k < d) ? y[(size_t)j * d + k] : 0.0;
        yy += yj[k] * yj[k];
    }
    const int i0 = blockIdx.y * rows_per_block, i1 = min(I, i0 + rows_per_block);
    for (int i = i0; i < i1; i++) {
        const double *xi = x + (size_t)i * d;
        double dot = 0.0, xx = 0.0;
#pragma unroll
        for (int k = 0; k < MAX_LATENT_DIM; k++) {
            const double xv = (k < d) ? xi[k] : 0.0;
            dot += xv * yj[k];
            xx += xv * xv;
        }
        double v = -2.0 * dot;
        v += xx;
        v += yy;
        if (live) D[(size_t)i * J + j] = v > 0.0 ? v : 0.0;
    }
}

// Radix-select histogram: among keys whose bits above `shift+bits` equal `prefix`, count the
// `bits`-bit digit at `shift`.  Non-negative doubles order like their bit patterns.
__global__ __launch_bounds__(256) void k_select_hist(const unsigned long long *__restrict__ keys,
                                                     size_t n, unsigned long long prefix, int shift,
                                                     int bits, unsigned long long *__restrict__ hist) {
    __shared__ unsigned int lh[4096];
    for (int t = threadIdx.x; t < 4096; t += 256) lh[t] = 0;
    __syncthreads();
    const int hi = shift + bits;
    const unsigned long long dmask = (1ull << bits) - 1ull;
    for (size_t t = (size_t)blockIdx.x * 256 + threadIdx.x; t < n; t += (size_t)gridDim.x * 256) {
        const unsigned long long key = keys[t];
        const bool match = (hi >= 64) ? true : ((key >> hi) == prefix);
        if (match) atomicAdd(&lh[(key >> shift) & dmask], 1u);
    }
    __syncthreads();
    for (int t = threadIdx.x; t < 4096; t += 256)
        if (lh[t]) atomicAdd(&hist[t], (unsigned long long)lh[t]);
}

template <typename T>
__global__ __launch_bounds__(256) void k_scale_to_cost(const double *__restrict__ D, double denom,
                                                       T *__restrict__ C, int I, int J, int ld) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= ld) return;
    for (int i = blockIdx.y; i < I; i += gridDim.y)
        C[(size_t)i * ld + j] = (j < J) ? (T)(D[(size_t)i * J + j] / denom) : (T)0;
}

__global__ void k_fill(double *x, double val, int n) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t < n) x[t] = val;
}

}  // namespace

// ============================================================================================
// Host side
// ============================================================================================
struct spadot_ot_solver {
    int I = 0, J = 0, ld = 0, storage = SPADOT_F64;
    hipStream_t stream = nullptr;
    void *C = nullptr, *K = nullptr;
    // length-I vectors
    double *a = nullptr, *old_a = nullptr, *u = nullptr, *p = nullptr, *dx = nullptr, *adx = nullptr;
    // length-ld vectors
    double *b = nullptr, *old_b = nullptr, *v = nullptr, *q = nullptr, *dy = nullptr, *w = nullptr,
           *tcol = nullptr;
    double *backup = nullptr;  // two snapshots of the six mutable vectors (a, old_a, adx, b, old_b, w): backup + which * mut_count
    int backup_sel = 0;        // which of the two snapshot()/restore() use
    bool want_rowsums = false; // next fused pass also stores its row sums to rt (pipelined gap measure)
    size_t mut_count = 0;
    double *red = nullptr;     // block partials of the vector reductions
    int redo_batches = 0;
    double *part = nullptr;    // nchunk x ld
    double *rt = nullptr;      // 4 x I
    double *scal = nullptr;    // 8 device scalars: gap, primal, dual, sumKbar, ...
    int *flags = nullptr;      // MAX_BATCH + 1 ints (last = absorb counter)
    int *ctl = nullptr;        // {stop, converged batch + 1, tau batch + 1, -}: device-side stop decision of speculated batches
    int *h_ctl = nullptr;      // pinned mirror
    int *stop_arg = nullptr;   // what the hot kernels receive as `stop`: ctl while batches are speculated, else null
    double *h_scal = nullptr;  // pinned
    int *h_flags = nullptr;    // pinned
    int nchunk = 1, rows_per_chunk = 1;
    // fused single-sweep pass (k_fused_pass): 0 = not applicable for this shape
    int fused_vpt = 0, fused_r = 0, fused_blocks = 0, fused_rows_per_block = 0;
    size_t fused_lds = 0;
    double sum_kbar_eps = -1.0;
    long long cost_info[2] = {0, 0};   // last set_cost_from_latents: {path, candidates collected} (ot_cost.hip)
    void *cost_ws = nullptr;           // its scratch (sample, bracket candidates, sort space), kept between calls
    size_t cost_ws_bytes = 0;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    size_t elt() const { return storage == SPADOT_F32 ? 4 : 8; }
};

namespace {

void *dmalloc(size_t bytes) {
    void *p = nullptr;
    HIP_CHECK(hipMalloc(&p, bytes ? bytes : 8));
    return p;
}

// Owner of a temporary device allocation: freed when the scope is left, also by an unwinding HIP_CHECK.
struct DevBuf {
    void *p = nullptr;
    explicit DevBuf(size_t bytes) : p(dmalloc(bytes)) {}
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    ~DevBuf() { if (p) (void)hipFree(p); }
    template <typename T> T *as() const { return (T *)p; }
};

dim3 grid_cols(const spadot_ot_solver *s, int V) {
    // rows are strided over gridDim.y: ~2k blocks fill the chip, and a launch that exits on a clear
    // tau flag stays cheap
    const int gx = (s->ld + 256 * V - 1) / (256 * V);
    const int gy = std::max(1, std::min(s->I, 2048 / gx));
    return dim3(gx, (unsigned)gy);
}

void require_device() {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)      // this library has no CPU path
        throw hip_failure{e != hipSuccess ? e : hipErrorNoDevice, __FILE__, __LINE__, "hipGetDeviceCount: no HIP device"};
}

template <typename T> void launch_build_K(spadot_ot_solver *s, double eps, const int *flag) {
    constexpr int V = Vec<T>::N;
    hipLaunchKernelGGL(k_build_K<T>, grid_cols(s, V), dim3(256), 0, s->stream, (T *)s->K,
                       (const T *)s->C, s->u, s->v, eps, s->I, s->J, s->ld, flag);
}

void build_K(spadot_ot_solver *s, double eps, const int *flag) {
    if (s->storage == SPADOT_F32) launch_build_K<float>(s, eps, flag);
    else launch_build_K<double>(s, eps, flag);
}

// scal[3] = sum exp(-C/eps) (or sum of a given Kbar matrix)
template <typename T> void sum_kbar_T(spadot_ot_solver *s, const void *M, double eps, bool from_cost) {
    dim3 g((s->I + ROW_WAVES - 1) / ROW_WAVES);
    if (from_cost)
        hipLaunchKernelGGL((k_row_reduce<T, 1>), g, dim3(256), 0, s->stream, (const T *)M, eps, s->I,
                           s->J, s->ld, s->rt);
    else
        hipLaunchKernelGGL((k_row_reduce<T, 0>), g, dim3(256), 0, s->stream, (const T *)M, eps, s->I,
                           s->J, s->ld, s->rt);
    hipLaunchKernelGGL(k_vec_sum, dim3(1), dim3(1024), 0, s->stream, s->rt, s->I, s->scal, 3);
}
void sum_kbar(spadot_ot_solver *s, const void *M, double eps, bool from_cost) {
    if (s->storage == SPADOT_F32) sum_kbar_T<float>(s, M, eps, from_cost);
    else sum_kbar_T<double>(s, M, eps, from_cost);
}

struct IterParams { double eps, tau, l1, l2, al1, al2; };

template <typename T, int VPT, int R, typename WT, bool WRS>
void launch_fused_k(spadot_ot_solver *s, const IterParams &P, int *flag) {
    static PerDeviceFlag attr_set;     
    auto kern = k_fused_pass<T, VPT, R, WT, WRS>;
    if (!attr_set) {
        HIP_CHECK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set = true;
    }
    hipLaunchKernelGGL(kern, dim3(s->fused_blocks), dim3(FUSED_THREADS), s->fused_lds, s->stream,
                       (const T *)s->K, s->w, s->a, s->old_a, s->adx, s->p, s->dx, s->u, P.al1,
                       1.0 / (P.l1 + P.eps), P.tau, (T *)s->part, s->I, s->ld, s->fused_rows_per_block, flag,
                       (const int *)s->stop_arg, WRS ? s->rt : (double *)nullptr);
}
// (s->want_rowsums: set around the first pass of a pipelined batch)
template <typename T, int VPT, int R, typename WT = double>
void launch_fused(spadot_ot_solver *s, const IterParams &P, int *flag) {
    if (s->want_rowsums) launch_fused_k<T, VPT, R, WT, true>(s, P, flag);
    else launch_fused_k<T, VPT, R, WT, false>(s, P, flag);
}

template <typename T> void fused_pass_T(spadot_ot_solver *s, const IterParams &P, int *flag);
#define FUSED_CASE(T, VPT, R) case VPT: launch_fused<T, VPT, R>(s, P, flag); break;
template <> void fused_pass_T<float>(spadot_ot_solver *s, const IterParams &P, int *flag) {
    switch (s->fused_vpt) {
        FUSED_CASE(float, 1, 2) FUSED_CASE(float, 2, 2) FUSED_CASE(float, 3, 2) FUSED_CASE(float, 4, 2)
        FUSED_CASE(float, 5, 2) FUSED_CASE(float, 6, 1) FUSED_CASE(float, 7, 1) FUSED_CASE(float, 8, 1)
        // wider rows (J up to 20 480, e.g. cfg5's 20k): w.dy is staged as fp32 so that it still fits LDS
        case 9: launch_fused<float, 9, 1, float>(s, P, flag); break;
        case 10: launch_fused<float, 10, 1, float>(s, P, flag); break;
        default: throw std::logic_error("fused geometry outside the instantiated table");
    }
}
template <> void fused_pass_T<double>(spadot_ot_solver *s, const IterParams &P, int *flag) {
    switch (s->fused_vpt) {
        FUSED_CASE(double, 1, 2) FUSED_CASE(double, 2, 2) FUSED_CASE(double, 3, 2) FUSED_CASE(double, 4, 2)
        FUSED_CASE(double, 5, 2) FUSED_CASE(double, 6, 2) FUSED_CASE(double, 7, 1) FUSED_CASE(double, 8, 1)
        FUSED_CASE(double, 9, 1) FUSED_CASE(double, 10, 1) FUSED_CASE(double, 11, 1) FUSED_CASE(double, 12, 1)
        default: throw std::logic_error("fused geometry outside the instantiated table");
    }
}
#undef FUSED_CASE

template <typename T, int VPT, int R, typename WT = double>
void launch_fused_rowdot(spadot_ot_solver *s) {
    static PerDeviceFlag attr_set;     
    auto kern = k_fused_rowdot<T, VPT, R, WT>;
    if (!attr_set) {
        HIP_CHECK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set = true;
    }
    hipLaunchKernelGGL(kern, dim3(s->fused_blocks), dim3(FUSED_THREADS), s->fused_lds, s->stream, (const T *)s->K,
                       s->w, s->rt, s->I, s->ld, s->fused_rows_per_block, (const int *)s->stop_arg);
}
template <typename T> void fused_rowdot_T(spadot_ot_solver *s);
#define FUSED_CASE(T, VPT, R) case VPT: launch_fused_rowdot<T, VPT, R>(s); break;
template <> void fused_rowdot_T<float>(spadot_ot_solver *s) {
    switch (s->fused_vpt) {
        FUSED_CASE(float, 1, 2) FUSED_CASE(float, 2, 2) FUSED_CASE(float, 3, 2) FUSED_CASE(float, 4, 2)
        FUSED_CASE(float, 5, 2) FUSED_CASE(float, 6, 1) FUSED_CASE(float, 7, 1) FUSED_CASE(float, 8, 1)
        case 9: launch_fused_rowdot<float, 9, 1, float>(s); break;
        case 10: launch_fused_rowdot<float, 10, 1, float>(s); break;
        default: throw std::logic_error("fused geometry outside the instantiated table");
    }
}
template <> void fused_rowdot_T<double>(spadot_ot_solver *s) {
    switch (s->fused_vpt) {
        FUSED_CASE(double, 1, 2) FUSED_CASE(double, 2, 2) FUSED_CASE(double, 3, 2) FUSED_CASE(double, 4, 2)
        FUSED_CASE(double, 5, 2) FUSED_CASE(double, 6, 2) FUSED_CASE(double, 7, 1) FUSED_CASE(double, 8, 1)
        FUSED_CASE(double, 9, 1) FUSED_CASE(double, 10, 1) FUSED_CASE(double, 11, 1) FUSED_CASE(double, 12, 1)
        default: throw std::logic_error("fused geometry outside the instantiated table");
    }
}
#undef FUSED_CASE

void absorb_if_flagged(spadot_ot_solver *s, const IterParams &P, int *flag) {
    const int mx = std::max(s->I, s->J);
    hipLaunchKernelGGL(k_absorb_vec, dim3((mx + 255) / 256), dim3(256), 0, s->stream, s->a, s->b, s->u,
                       s->v, s->adx, s->w, s->dx, s->dy, P.eps, s->I, s->J, flag, s->flags + MAX_BATCH, s->tcol);
    build_K(s, P.eps, flag);
}

template <typename T> void one_iteration_T(spadot_ot_solver *s, const IterParams &P, int *flag, bool absorb) {
    constexpr int V = Vec<T>::N;
    const int I = s->I, J = s->J, ld = s->ld;
    if (s->fused_vpt > 0) {
        fused_pass_T<T>(s, P, flag);
        hipLaunchKernelGGL(k_col_fin2<T>, dim3((ld + 63) / 64), dim3(1024), 0, s->stream, (const T *)s->part,
                           s->fused_blocks, s->b, s->old_b, s->w, s->q, s->dy, s->v, P.al2,
                           1.0 / (P.l2 + P.eps), P.tau, J, ld, flag, s->tcol, 0, (const int *)s->stop_arg);
    } else {
        hipLaunchKernelGGL(k_row_pass<T>, dim3((I + ROW_WAVES - 1) / ROW_WAVES), dim3(256), 0, s->stream,
                           (const T *)s->K, s->w, s->a, s->old_a, s->adx, s->p, s->dx, s->u, P.al1,
                           1.0 / (P.l1 + P.eps), P.tau, I, ld, flag);
        hipLaunchKernelGGL(k_col_pass<T>, dim3((ld + 256 * V - 1) / (256 * V), s->nchunk), dim3(256), 0,
                           s->stream, (const T *)s->K, s->adx, s->part, I, ld, s->rows_per_chunk);
        hipLaunchKernelGGL(k_col_fin, dim3((ld + 255) / 256), dim3(256), 0, s->stream, s->part, s->nchunk,
                           s->b, s->old_b, s->w, s->q, s->dy, s->v, P.al2, 1.0 / (P.l2 + P.eps), P.tau,
                           J, ld, flag, s->tcol, 0);
    }
    if (absorb) absorb_if_flagged(s, P, flag);
}

// The two launches of a fused iteration on their own (the pipelined last-stage loop puts the gap measure between them).
template <typename T> void fused_colfin_T(spadot_ot_solver *s, const IterParams &P, int *flag) {
    hipLaunchKernelGGL(k_col_fin2<T>, dim3((s->ld + 63) / 64), dim3(1024), 0, s->stream, (const T *)s->part,
                       s->fused_blocks, s->b, s->old_b, s->w, s->q, s->dy, s->v, P.al2,
                       1.0 / (P.l2 + P.eps), P.tau, s->J, s->ld, flag, s->tcol, 0, (const int *)s->stop_arg);
}
void fused_pass_any(spadot_ot_solver *s, const IterParams &P, int *flag) {
    if (s->storage == SPADOT_F32) fused_pass_T<float>(s, P, flag); else fused_pass_T<double>(s, P, flag);
}
void fused_colfin_any(spadot_ot_solver *s, const IterParams &P, int *flag) {
    if (s->storage == SPADOT_F32) fused_colfin_T<float>(s, P, flag); else fused_colfin_T<double>(s, P, flag);
}

// `iters` scaling iterations, no host sync.
//   safe mode (absorb = true): after every iteration the tau-absorb pair runs and acts if that
//     iteration's flag is set (ot_func.cpp:778-819) -- exact reference semantics at any tau.
//   fast mode (absorb = false): the two absorb launches are skipped; every iteration raises flags[0] if a
//     scaling exceeds tau, and the CALLER must check it and redo the batch in safe mode from a snapshot
//     (K, u, v are untouched in fast mode, so the snapshot is the six mutable vectors only).
void run_iterations(spadot_ot_solver *s, const IterParams &P, int iters, bool absorb = true) {
    bool first = true;
    while (iters > 0) {
        const int nb = std::min(iters, MAX_BATCH);
        // safe mode uses one flag per iteration of the chunk; fast mode's single flag (flags[0]) is STICKY over the
        // whole call -- a batch longer than MAX_BATCH must not lose a tau overflow raised in an earlier chunk
        if (absorb || first)
            HIP_CHECK(hipMemsetAsync(s->flags, 0, sizeof(int) * MAX_BATCH, s->stream));
        first = false;
        for (int t = 0; t < nb; t++) {
            int *flag = absorb ? s->flags + t : s->flags;
            if (s->storage == SPADOT_F32) one_iteration_T<float>(s, P, flag, absorb);
            else one_iteration_T<double>(s, P, flag, absorb);
        }
        iters -= nb;
    }
}

__global__ __launch_bounds__(256) void k_copy_unless(double *__restrict__ dst, const double *__restrict__ src, size_t n,
                                                     const int *__restrict__ stop) {
    if (*stop != 0) return;
    for (size_t k = (size_t)blockIdx.x * 256 + threadIdx.x; k < n; k += (size_t)gridDim.x * 256) dst[k] = src[k];
}
void snapshot(spadot_ot_solver *s) {
    double *dst = s->backup + (size_t)s->backup_sel * s->mut_count;
    if (s->stop_arg != nullptr) {      // speculated batch: the snapshot of a batch behind the stopping one must not be taken
        const size_t n = s->mut_count;
        hipLaunchKernelGGL(k_copy_unless, dim3((unsigned)std::min<size_t>((n + 255) / 256, 1024)), dim3(256), 0, s->stream,
                           dst, (const double *)s->a, n, (const int *)s->stop_arg);
        return;
    }
    HIP_CHECK(hipMemcpyAsync(dst, s->a, sizeof(double) * s->mut_count, hipMemcpyDeviceToDevice, s->stream));
}
void restore(spadot_ot_solver *s) {
    HIP_CHECK(hipMemcpyAsync(s->a, s->backup + (size_t)s->backup_sel * s->mut_count, sizeof(double) * s->mut_count,
                             hipMemcpyDeviceToDevice, s->stream));
}

double read_gap(spadot_ot_solver *s) {
    HIP_CHECK(hipMemcpyAsync(s->h_scal, s->scal, sizeof(double) * 4, hipMemcpyDeviceToHost, s->stream));
    HIP_CHECK(hipMemcpyAsync(s->h_flags, s->flags, sizeof(int), hipMemcpyDeviceToHost, s->stream));
    HIP_CHECK(hipStreamSynchronize(s->stream));
    HIP_CHECK(hipGetLastError());
    return s->h_scal[0];
}

// (threshold, batch: only read when batches are being speculated, i.e. s->stop_arg is set)
void drift_measure(spadot_ot_solver *s, double eps, double threshold = 0.0, int batch = 0) {
    const int nblk = (std::max(s->I, s->J) + 255) / 256;
    hipLaunchKernelGGL(k_drift_part, dim3(nblk), dim3(256), 0, s->stream, s->a, s->old_a, s->u, s->b, s->old_b,
                       s->v, eps, s->I, s->J, s->red, (const int *)s->stop_arg);
    hipLaunchKernelGGL(k_drift_final, dim3(1), dim3(256), 0, s->stream, s->red, nblk, s->scal, threshold, batch,
                       s->stop_arg, (const int *)s->flags);
}

// Solver-only gap (uniform dx, dy): ONE sweep of K for sdot, everything else from vectors (k_gap2_part).
void gap_measure_fast(spadot_ot_solver *s, const IterParams &P, double threshold = 0.0, int batch = 0) {
    const int I = s->I, J = s->J, ld = s->ld;
    dim3 g((I + ROW_WAVES - 1) / ROW_WAVES);
    if (s->fused_vpt > 0) {
        if (s->storage == SPADOT_F32) fused_rowdot_T<float>(s);
        else fused_rowdot_T<double>(s);
    } else if (s->storage == SPADOT_F32)
        hipLaunchKernelGGL(k_row_dot<float>, g, dim3(256), 0, s->stream, (const float *)s->K, s->w, s->rt, I, ld);
    else
        hipLaunchKernelGGL(k_row_dot<double>, g, dim3(256), 0, s->stream, (const double *)s->K, s->w, s->rt, I, ld);
    const int nblk = (std::max(I, J) + 255) / 256;
    hipLaunchKernelGGL(k_gap2_part, dim3(nblk), dim3(256), 0, s->stream, s->rt, s->tcol, s->a, s->b, s->u, s->v,
                       s->p, s->q, P.eps, P.l1, P.l2, I, J, s->red, (const int *)s->stop_arg);
    hipLaunchKernelGGL(k_gap2_final, dim3(1), dim3(256), 0, s->stream, s->red, nblk, P.eps, P.l1, P.l2, I, J, s->scal,
                       threshold, batch, s->stop_arg, (const int *)s->flags);
}

// True primal-dual gap of the current (a, b, K) with R = a K b formed on the fly; scal[3] must
// already hold sum(Kbar).  Rout (device, I x ld, storage type) may be null.
template <typename T> void gap_measure_T(spadot_ot_solver *s, const IterParams &P, void *Rout) {
    constexpr int V = Vec<T>::N;
    const int I = s->I, J = s->J, ld = s->ld;
    hipLaunchKernelGGL((k_gap_rows<T, true>), dim3((I + ROW_WAVES - 1) / ROW_WAVES), dim3(256), 0,
                       s->stream, (const T *)s->K, (const T *)s->C, s->a, s->b, s->dy, (T *)Rout, s->rt,
                       I, J, ld);
    hipLaunchKernelGGL(k_col_pass<T>, dim3((ld + 256 * V - 1) / (256 * V), s->nchunk), dim3(256), 0,
                       s->stream, (const T *)s->K, s->adx, s->part, I, ld, s->rows_per_chunk);
    hipLaunchKernelGGL(k_col_fin, dim3((ld + 255) / 256), dim3(256), 0, s->stream, s->part, s->nchunk,
                       (double *)nullptr, (double *)nullptr, (double *)nullptr, (const double *)nullptr,
                       (const double *)nullptr, (const double *)nullptr, 0.0, 0.0, 0.0, J, ld,
                       (int *)nullptr, s->tcol, 1);
    hipLaunchKernelGGL(k_gap_fin, dim3(1), dim3(1024), 0, s->stream, s->rt, s->tcol, s->a, s->b, s->u,
                       s->v, s->p, s->q, s->dx, s->dy, P.eps, P.l1, P.l2, I, J, 1, 1, s->scal);
}
void gap_measure(spadot_ot_solver *s, const IterParams &P, void *Rout) {
    if (s->storage == SPADOT_F32) gap_measure_T<float>(s, P, Rout);
    else gap_measure_T<double>(s, P, Rout);
}

int spec_batches() { static const int v = [] { const char *e = getenv("SPADOT_OT_SPEC_BATCHES"); return e ? atoi(e) : 3; }(); return v; }
constexpr bool pipe_gap() { return true; }      // (the last stage's loop pipelined by one pass: round 3; its switch went in round 5)

// The pipelined last-stage loop (see process_stage): groups of SPEC batches until the gap is at or below `threshold`
// (or max_groups groups have run; < 0: no limit).  Updates gap, cur_iter, done (iterations) and nchecks.
void pipelined_gap_loop(spadot_ot_solver *s, const IterParams &P, int batch_size, double threshold, int SPEC, int max_iter,
                    int max_groups, double &gap, int &cur_iter, int &done, int &nchecks) {
    const int iters = batch_size;
    const int nblk = (std::max(s->I, s->J) + 255) / 256;
    bool carry = false;                 // the head of batch `par` has been enqueued (and the batch before it did not stop)
    int par = 0;                        // snapshot / flag slot of the next batch to start (or of the carried one)
    auto head = [&](int slot, int measure_batch, int measure_slot) {
        s->backup_sel = slot;
        snapshot(s);
        HIP_CHECK(hipMemsetAsync(s->flags + slot, 0, sizeof(int), s->stream));
        s->want_rowsums = true;
        fused_pass_any(s, P, s->flags + slot);
        s->want_rowsums = false;
        if (measure_batch >= 0) {       // the batch before: a = old_a (the pass has moved a on), b and tcol still its own
            hipLaunchKernelGGL(k_gap2_part, dim3(nblk), dim3(256), 0, s->stream, s->rt, s->tcol, s->old_a, s->b, s->u, s->v,
                               s->p, s->q, P.eps, P.l1, P.l2, s->I, s->J, s->red, (const int *)s->stop_arg);
            hipLaunchKernelGGL(k_gap2_final, dim3(1), dim3(256), 0, s->stream, s->red, nblk, P.eps, P.l1, P.l2, s->I, s->J,
                               s->scal, threshold, measure_batch, s->stop_arg, (const int *)(s->flags + measure_slot));
        }
    };
    auto body = [&](int slot) {
        fused_colfin_any(s, P, s->flags + slot);
        for (int t = 1; t < iters; t++) { fused_pass_any(s, P, s->flags + slot); fused_colfin_any(s, P, s->flags + slot); }
    };
    int groups = 0;
    while (gap > threshold && (max_groups < 0 || groups < max_groups)) {
        groups++;
        if ((long long)cur_iter + (long long)(SPEC + 1) * iters >= max_iter) break;
        HIP_CHECK(hipMemsetAsync(s->ctl, 0, sizeof(int) * 4, s->stream));
        s->stop_arg = s->ctl;
        const int par0 = par;           // slot of this group's batch 0
        for (int j = 0; j < SPEC; j++) {
            if (!(carry && j == 0)) head(par, j - 1, par ^ 1);
            body(par);
            par ^= 1;
        }
        head(par, SPEC - 1, par ^ 1);   // the next batch's head measures this group's last batch
        s->stop_arg = nullptr;
        HIP_CHECK(hipMemcpyAsync(s->h_ctl, s->ctl, sizeof(int) * 4, hipMemcpyDeviceToHost, s->stream));
        gap = read_gap(s);
        const int conv = s->h_ctl[1], taub = s->h_ctl[2];
        if (taub != 0) {
            // batch taub - 1 of this group exceeded tau: back to ITS snapshot, redo it exactly, measure it the plain way
            const int before = taub - 1;
            done += before * iters; cur_iter += before * iters; nchecks += before;
            s->backup_sel = (par0 + before) & 1;
            restore(s);
            run_iterations(s, P, iters, /*absorb=*/true);
            gap_measure_fast(s, P);
            gap = read_gap(s);
            s->redo_batches++;
            done += iters; cur_iter += iters; nchecks++;
            carry = false;
            par = 0;
        } else if (conv != 0) {
            // batch conv - 1 converged: the snapshot taken at the head of the batch after it is the state it left
            done += conv * iters; cur_iter += conv * iters; nchecks += conv;
            s->backup_sel = (par0 + conv) & 1;
            restore(s);
            carry = false;
        } else {
            done += SPEC * iters; cur_iter += SPEC * iters; nchecks += SPEC;
            carry = true;               // the head of batch `par` is in place; its body opens the next group
        }
    }
    if (carry) {                        // leaving with a head enqueued (max_iter is near): undo its pass
        s->backup_sel = par;
        restore(s);
    }
    s->backup_sel = 0;
}

// ot_func.cpp:830-930 on device state.  Returns the measure; *iters_done counts scaling iterations.
// The reference's cur_iter bookkeeping (including its -1 quirk, :821-824 + :869) is kept on the host.
double process_stage(spadot_ot_solver *s, const IterParams &P, bool last_stage, int batch_size,
                     double threshold, int cur_iter, int max_iter, void *Rout, int *iters_done,
                     int *checks, bool fast = false, int *cur_iter_out = nullptr) {
    double gap = 1e100;
    int done = 0, nchecks = 0;
    s->stop_arg = nullptr;             // (a failed call may have left a speculation group open)
    auto measure = [&]() {
        if (!last_stage) drift_measure(s, P.eps);
        else if (fast) gap_measure_fast(s, P);
        else gap_measure(s, P, Rout);
    };
    // Fast mode on a fused geometry: SPEC batches are enqueued at a time and the DEVICE decides after each whether the stage is
    // over (k_drift_final / k_gap2_final -> ctl); the batches behind the deciding one return at once, and the host reads one
    // 16-byte record per group instead of stalling the queue after every batch.  Same batches, same order, same counts.
    const int SPEC = spec_batches();
    const bool PIPE = pipe_gap();
    // Last stage: the duality-gap measure of the state a batch leaves needs the row sums sum_j K_ij (b.dy)_j -- a sweep of K
    // of its own (k_fused_rowdot, 60 us per 5 iterations at 10k x 10k).  They are also the first thing the NEXT iteration's
    // fused pass computes.  So the loop is pipelined by one pass: the head of batch j + 1 = {snapshot, first fused pass
    // (storing its row sums), MEASURE of batch j} and the measure reads old_a (= a of batch j's end, the pass has just
    // moved a on), b and the column sums of batch j's last finalise, which the head leaves untouched.  A converged batch
    // j sets the stop word there; the rest of batch j + 1 returns at once and the host restores batch j + 1's snapshot
    // (= the state batch j left).  Two snapshots and two tau flags alternate, so that a tau flag raised by batch j
    // still finds batch j's own snapshot.  Same batches, same iteration counts; what is spent per batch is one vector
    // measure, what is wasted is ONE pass at the end.
    if (fast && last_stage && SPEC > 1 && PIPE && s->fused_vpt > 0 && batch_size <= MAX_BATCH)
        pipelined_gap_loop(s, P, batch_size, threshold, SPEC, max_iter, -1, gap, cur_iter, done, nchecks);
    while (fast && SPEC > 1 && s->fused_vpt > 0 && gap > threshold) {
        const int iters = last_stage ? batch_size : 5;
        if (cur_iter + SPEC * iters >= max_iter) break;          // near max_iter: the batch-by-batch loop below keeps the quirks
        HIP_CHECK(hipMemsetAsync(s->ctl, 0, sizeof(int) * 4, s->stream));
        s->stop_arg = s->ctl;
        for (int j = 0; j < SPEC; j++) {
            snapshot(s);
            run_iterations(s, P, iters, /*absorb=*/false);
            if (!last_stage) drift_measure(s, P.eps, threshold, j);
            else gap_measure_fast(s, P, threshold, j);
        }
        s->stop_arg = nullptr;
        HIP_CHECK(hipMemcpyAsync(s->h_ctl, s->ctl, sizeof(int) * 4, hipMemcpyDeviceToHost, s->stream));
        gap = read_gap(s);                                       // (synchronises; scal[0] is the deciding batch's measure)
        const int conv = s->h_ctl[1], taub = s->h_ctl[2];
        if (taub != 0) {
            // batch taub - 1 raised the tau flag: the state is that batch's (approximate) result, the backup its start; the
            // batches before it count, this one is redone exactly
            const int before = taub - 1;
            done += before * iters; cur_iter += before * iters; nchecks += before;
            restore(s);
            run_iterations(s, P, iters, /*absorb=*/true);
            measure();
            gap = read_gap(s);
            s->redo_batches++;
            done += iters; cur_iter += iters; nchecks++;
        } else {
            const int ran = conv != 0 ? conv : SPEC;
            done += ran * iters; cur_iter += ran * iters; nchecks += ran;
        }
    }
    while (gap > threshold) {
        const int iters = last_stage ? batch_size : 5;
        // step1_process: stops early (returning -1) once the counter reaches max_iter
        int run = iters;
        bool hit = false;
        if (cur_iter + iters >= max_iter) {
            run = std::max(1, std::min(iters, max_iter - cur_iter));
            hit = true;
        }
        if (fast) {
            snapshot(s);
            run_iterations(s, P, run, /*absorb=*/false);
            measure();
            gap = read_gap(s);
            if (s->h_flags[0] != 0) {          // a scaling exceeded tau somewhere in the batch: redo it exactly
                restore(s);
                run_iterations(s, P, run, /*absorb=*/true);
                measure();
                gap = read_gap(s);
                s->redo_batches++;
            }
        } else {
            run_iterations(s, P, run, /*absorb=*/true);
            measure();
            gap = read_gap(s);
        }
        done += run;
        if (hit) {
            printf("Reached max_iter with duality gap still above threshold. Returning");
            cur_iter = -1;
        } else {
            cur_iter += iters;
        }
        nchecks++;
    }
    if (iters_done) *iters_done = done;
    if (checks) *checks += nchecks;
    if (cur_iter_out) *cur_iter_out = cur_iter;
    return gap;
}

void choose_chunks(spadot_ot_solver *s) {
    const int V = s->storage == SPADOT_F32 ? 4 : 2;
    const int gx = (s->ld + 256 * V - 1) / (256 * V);
    int want = std::max(1, 2048 / gx);              // ~2048 blocks = 8 per CU
    want = std::min(want, std::max(1, s->I / 16));  // at least 16 rows per chunk
    s->rows_per_chunk = (s->I + want - 1) / want;
    s->nchunk = (s->I + s->rows_per_chunk - 1) / s->rows_per_chunk;
}

// Fused single-sweep pass is used when a row fits the register budget of one 1024-thread workgroup
// and w fits LDS; SPADOT_OT_NO_FUSED=1 forces the two-sweep kernels (A/B runs, fallback tests).
void choose_fused(spadot_ot_solver *s) {
    s->fused_vpt = 0;
    const char *off = getenv("SPADOT_OT_NO_FUSED");
    if (off && off[0] == '1') return;
    const int V = s->storage == SPADOT_F32 ? 4 : 2;
    const int vpt = (s->ld + V * FUSED_THREADS - 1) / (V * FUSED_THREADS);
    const int max_vpt = s->storage == SPADOT_F32 ? 10 : 12;     // beyond: the register tile would spill
    if (vpt > max_vpt) return;
    const size_t wbytes = (s->storage == SPADOT_F32 && vpt > 8) ? 4 : 8;   // fp32 w image for the widest rows
    const int R = vpt <= (s->storage == SPADOT_F32 ? 5 : 6) ? 2 : 1;     // must match the FUSED_CASE table
    size_t lds = wbytes * (size_t)vpt * FUSED_THREADS * V + sizeof(double) * (2 * (FUSED_THREADS / 64) * R) + 8;
    if (lds + 4096 > 160 * 1024) return;
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
            cus = prop.multiProcessorCount;
    }
    int blocks = std::max(1, std::min(cus, (s->I + 2 * R - 1) / (2 * R)));
    int rpb = (s->I + blocks - 1) / blocks;
    rpb = round_up(rpb, R);
    blocks = (s->I + rpb - 1) / rpb;
    lds += sizeof(double) * 4 * (size_t)rpb;          // staged p, dx, u, previous a of the band
    if (lds > 160 * 1024) return;
    s->fused_vpt = vpt; s->fused_r = R; s->fused_blocks = blocks; s->fused_rows_per_block = rpb;
    s->fused_lds = lds;
}

void upload_vec(spadot_ot_solver *s, double *dst, const double *src, int n) {
    HIP_CHECK(hipMemcpyAsync(dst, src, sizeof(double) * n, hipMemcpyHostToDevice, s->stream));
}
void download_vec(spadot_ot_solver *s, double *dst, const double *src, int n) {
    HIP_CHECK(hipMemcpyAsync(dst, src, sizeof(double) * n, hipMemcpyDeviceToHost, s->stream));
}
// host (m x n contiguous) <-> device (m x ld)
template <typename T> void upload_mat(spadot_ot_solver *s, void *dst, const T *src) {
    HIP_CHECK(hipMemsetAsync(dst, 0, (size_t)s->I * s->ld * sizeof(T), s->stream));
    HIP_CHECK(hipMemcpy2DAsync(dst, (size_t)s->ld * sizeof(T), src, (size_t)s->J * sizeof(T),
                               (size_t)s->J * sizeof(T), s->I, hipMemcpyHostToDevice, s->stream));
}
template <typename T> void download_mat(spadot_ot_solver *s, T *dst, const void *src) {
    HIP_CHECK(hipMemcpy2DAsync(dst, (size_t)s->J * sizeof(T), src, (size_t)s->ld * sizeof(T),
                               (size_t)s->J * sizeof(T), s->I, hipMemcpyDeviceToHost, s->stream));
}

}  // namespace

extern "C" {

const char *spadot_ot_version(void) { return "spadot_ot 0.1 (gfx950)"; }

int spadot_ot_create(spadot_ot_solver **out, int I, int J, int storage, void *stream) {
    SPADOT_ENTER
    if (!out) return -22;
    *out = nullptr;
    if (I <= 0 || J <= 0 || (storage != SPADOT_F64 && storage != SPADOT_F32)) return -22;
    require_device();
    struct Guard {                        // a failing allocation below unwinds: give back what was taken so far
        spadot_ot_solver *s;
        ~Guard() { if (s) spadot_ot_destroy(s); }
    } guard{new spadot_ot_solver()};
    spadot_ot_solver *s = guard.s;
    s->I = I; s->J = J; s->storage = storage; s->stream = (hipStream_t)stream;
    s->ld = round_up(J, 64);
    choose_chunks(s);
    choose_fused(s);
    const size_t mat = (size_t)I * s->ld * s->elt();
    s->C = dmalloc(mat);
    s->K = dmalloc(mat);
    HIP_CHECK(hipMemsetAsync(s->C, 0, mat, s->stream));
    HIP_CHECK(hipMemsetAsync(s->K, 0, mat, s->stream));
    const size_t L = s->ld;
    // one block: [a old_a adx | b old_b w] (mutable, snapshot unit) then [u p dx | v q dy tcol]
    s->mut_count = 3 * (size_t)I + 3 * L;
    const size_t vec_count = 6 * (size_t)I + 7 * L;
    double *vb = (double *)dmalloc(sizeof(double) * vec_count);
    s->a = vb;                                            // (owned by the solver from here on)
    HIP_CHECK(hipMemsetAsync(vb, 0, sizeof(double) * vec_count, s->stream)); s->old_a = vb + I; s->adx = vb + 2 * (size_t)I;
    s->b = vb + 3 * (size_t)I; s->old_b = s->b + L; s->w = s->b + 2 * L;
    double *cb = vb + s->mut_count;
    s->u = cb; s->p = cb + I; s->dx = cb + 2 * (size_t)I;
    s->v = cb + 3 * (size_t)I; s->q = s->v + L; s->dy = s->v + 2 * L; s->tcol = s->v + 3 * L;
    s->backup = (double *)dmalloc(sizeof(double) * 2 * s->mut_count);
    s->red = (double *)dmalloc(sizeof(double) * 8 * ((size_t)std::max<size_t>(I, L) / 256 + 2));
    s->part = (double *)dmalloc(sizeof(double) * (size_t)std::max(s->nchunk, s->fused_blocks) * L);
    s->rt = (double *)dmalloc(sizeof(double) * 4 * (size_t)I);
    s->scal = (double *)dmalloc(sizeof(double) * 8);
    s->flags = (int *)dmalloc(sizeof(int) * (MAX_BATCH + 1));
    s->ctl = (int *)dmalloc(sizeof(int) * 4);
    HIP_CHECK(hipMemsetAsync(s->ctl, 0, sizeof(int) * 4, s->stream));
    HIP_CHECK(hipHostMalloc((void **)&s->h_ctl, sizeof(int) * 4, hipHostMallocDefault));
    HIP_CHECK(hipMemsetAsync(s->scal, 0, sizeof(double) * 8, s->stream));
    HIP_CHECK(hipMemsetAsync(s->flags, 0, sizeof(int) * (MAX_BATCH + 1), s->stream));
    HIP_CHECK(hipHostMalloc((void **)&s->h_scal, sizeof(double) * 8, hipHostMallocDefault));
    HIP_CHECK(hipHostMalloc((void **)&s->h_flags, sizeof(int) * (MAX_BATCH + 1), hipHostMallocDefault));
    HIP_CHECK(hipEventCreate(&s->ev0));
    HIP_CHECK(hipEventCreate(&s->ev1));
    guard.s = nullptr;
    *out = s;
    return 0;
    SPADOT_LEAVE(SPADOT_EHIP)
}

void spadot_ot_destroy(spadot_ot_solver *s) {
    SPADOT_ENTER
    if (!s) return;
    (void)hipStreamSynchronize(s->stream);
    void *dev[] = {s->C, s->K, s->a, s->backup, s->red, s->part, s->rt, s->scal, s->flags, s->cost_ws, s->ctl};
    for (void *p : dev) if (p) (void)hipFree(p);
    if (s->h_ctl) (void)hipHostFree(s->h_ctl);
    if (s->h_scal) (void)hipHostFree(s->h_scal);
    if (s->h_flags) (void)hipHostFree(s->h_flags);
    if (s->ev0) (void)hipEventDestroy(s->ev0);
    if (s->ev1) (void)hipEventDestroy(s->ev1);
    delete s;
    SPADOT_LEAVE()
}

int spadot_ot_ld(const spadot_ot_solver *s) { return s ? s->ld : -22; }

void spadot_ot_fused_geometry(const spadot_ot_solver *s, int *out) {
    SPADOT_ENTER
    if (!s || !out) return;
    out[0] = s->fused_vpt; out[1] = s->fused_r; out[2] = s->fused_blocks; out[3] = s->fused_rows_per_block;
    SPADOT_LEAVE()
}

void *spadot_ot_matrix_dev(spadot_ot_solver *s, int which) {
    SPADOT_ENTER
    if (!s) return nullptr;
    return which == 0 ? s->C : (which == 1 ? s->K : nullptr);
    SPADOT_LEAVE(nullptr)
}

double *spadot_ot_vector_dev(spadot_ot_solver *s, int which) {
    SPADOT_ENTER
    if (!s) return nullptr;
    switch (which) {
        case 0: return s->a; case 1: return s->b; case 2: return s->u; case 3: return s->v;
        case 4: return s->old_a; case 5: return s->old_b; default: return nullptr;
    }
    SPADOT_LEAVE(nullptr)
}

int spadot_ot_vector_host(spadot_ot_solver *s, int which, double *out) {
    SPADOT_ENTER
    double *src = spadot_ot_vector_dev(s, which);
    if (!src || !out) return -22;
    const int n = (which == 0 || which == 2 || which == 4) ? s->I : s->J;
    download_vec(s, out, src, n);
    HIP_CHECK(hipStreamSynchronize(s->stream));
    return 0;
    SPADOT_LEAVE(SPADOT_EHIP)
}

int spadot_ot_matrix_host(spadot_ot_solver *s, int which, double *out) {
    SPADOT_ENTER
    void *src = spadot_ot_matrix_dev(s, which);
    if (!src || !out) return -22;
    const size_t n = (size_t)s->I * s->J;
    DevBuf buf(sizeof(double) * n);
    double *tmp = buf.as<double>();
    dim3 g((s->J + 255) / 256, (unsigned)std::min(s->I, 8192));
    if (s->storage == SPADOT_F32)
        hipLaunchKernelGGL((k_convert<float, double>), g, dim3(256), 0, s->stream, (const float *)src, s->ld, tmp, s->J, s->I, s->J);
    else
        hipLaunchKernelGGL((k_convert<double, double>), g, dim3(256), 0, s->stream, (const double *)src, s->ld, tmp, s->J, s->I, s->J);
    HIP_CHECK(hipMemcpyAsync(out, tmp, sizeof(double) * n, hipMemcpyDeviceToHost, s->stream));
    HIP_CHECK(hipStreamSynchronize(s->stream));
    return 0;
    SPADOT_LEAVE(SPADOT_EHIP)
}

int spadot_ot_set_cost_dev(spadot_ot_solver *s, const void *C_dev, int dtype, int ldc) {
    SPADOT_ENTER
    if (!s || !C_dev || ldc < s->J) return -22;
    dim3 g((s->ld + 255) / 256, (unsigned)std::min(s->I, 8192));
    if (dtype == SPADOT_F64 && s->storage == SPADOT_F64)
        hipLaunchKernelGGL((k_convert<double, double>), g, dim3(256), 0, s->stream, (const double *)C_dev, ldc, (double *)s->C, s->ld, s->I, s->J);
    else if (dtype == SPADOT_F64 && s->storage == SPADOT_F32)
        hipLaunchKernelGGL((k_convert<double, float>), g, dim3(256), 0, s->stream, (const double *)C_dev, ldc, (float *)s->C, s->ld, s->I, s->J);
    else if (dtype == SPADOT_F32 && s->storage == SPADOT_F64)
        hipLaunchKernelGGL((k_convert<float, double>), g, dim3(256), 0, s->stream, (const float *)C_dev, ldc, (double *)s->C, s->ld, s->I, s->J);
    else if (dtype == SPADOT_F32 && s->storage == SPADOT_F32)
        hipLaunchKernelGGL((k_convert<float, float>), g, dim3(256), 0, s->stream, (const float *)C_dev, ldc, (float *)s->C, s->ld, s->I, s->J);
    else
        return -22;
    s->sum_kbar_eps = -1.0;
    return 0;
    SPADOT_LEAVE(SPADOT_EHIP)
}

int spadot_ot_set_cost_host(spadot_ot_solver *s, const double *C_host) {
    SPADOT_ENTER
    if (!s || !C_host) return -22;
    DevBuf buf(sizeof(double) * (size_t)s->I * s->J);
    double *tmp = buf.as<double>();
    HIP_CHECK(hipMemcpyAsync(tmp, C_host, sizeof(double) * (size_t)s->I * s->J, hipMemcpyHostToDevice, s->stream));
    int rc = spadot_ot_set_cost_dev(s, tmp, SPADOT_F64, s->J);
    HIP_CHECK(hipStreamSynchronize(s->stream));
    return rc;
    SPADOT_LEAVE(SPADOT_EHIP)
}

}  // extern "C"

namespace {
// k-th smallest (0-based) of n non-negative doubles on the device, exact, by 12-bit radix select.
double select_kth(spadot_ot_solver *s, const double *D, size_t n, size_t k) {
    DevBuf hbuf(sizeof(unsigned long long) * 4096);
    unsigned long long *hist = hbuf.as<unsigned long long>();
    std::vector<unsigned long long> h(4096);
    unsigned long long prefix = 0;
    const int shifts[6] = {52, 40, 28, 16, 4, 0};
    const int nbits[6] = {12, 12, 12, 12, 12, 4};
    const int blocks = (int)std::min<size_t>(2048, (n + 255) / 256);
    for (int pass = 0; pass < 6; pass++) {
        HIP_CHECK(hipMemsetAsync(hist, 0, sizeof(unsigned long long) * 4096, s->stream));
        hipLaunchKernelGGL(k_select_hist, dim3(blocks), dim3(256), 0, s->stream,
                           (const unsigned long long *)D, n, prefix, shifts[pass], nbits[pass], hist);
        HIP_CHECK(hipMemcpyAsync(h.data(), hist, sizeof(unsigned long long) * 4096, hipMemcpyDeviceToHost, s->stream));
        HIP_CHECK(hipStreamSynchronize(s->stream));
        const int nb = 1 << nbits[pass];
        int bin = nb - 1;
        for (int t = 0; t < nb; t++) {
            if (k < h[t]) { bin = t; break; }
            k -= h[t];
        }
        prefix = (prefix << nbits[pass]) | (unsigned long long)bin;
    }
    double out;
    memcpy(&out, &prefix, sizeof(double));
    return out;
}
}  // namespace

// ot_cost.hip: distances recomputed on the fly, exact median through a sampled bracket, no I x J temporary
int spadot_cost_from_latents_impl(const double *x, const double *y, int d, int I, int J, int ld, int storage_f32, void *C,
                                  int divide_by_median, hipStream_t st, void **ws_ptr, size_t *ws_bytes, double *denom_out,
                                  long long *info_out);

extern "C" int spadot_ot_set_cost_from_latents_dev(spadot_ot_solver *s, const double *x_dev,
                                                   const double *y_dev, int d, int divide_by_median) {
    SPADOT_ENTER
    if (!s || !x_dev || !y_dev || d < 1 || d > MAX_LATENT_DIM) return -22;
    // (round 1's path -- the fp64 distance matrix materialised, radix select, a scaling pass: 3.3 ms at 10k x 10k against
    // 0.85 -- went with its switch in round 5)
    const int rc = spadot_cost_from_latents_impl(x_dev, y_dev, d, s->I, s->J, s->ld, s->storage == SPADOT_F32, s->C,
                                                 divide_by_median, s->stream, &s->cost_ws, &s->cost_ws_bytes, nullptr,
                                                 s->cost_info);
    if (rc >= 1000) throw hip_failure{(hipError_t)(rc - 1000), __FILE__, __LINE__, "spadot_cost_from_latents_impl"};
    if (rc != 0) return rc;
    s->sum_kbar_eps = -1.0;
    return 0;
    SPADOT_LEAVE(SPADOT_EHIP)
}

extern "C" {

// ot_solvers.py:164-449 with everything resident in HBM.
int spadot_ot_solve(spadot_ot_solver *s, const double *G_host, const spadot_ot_config *cfg,
                    spadot_ot_info *info) {
    SPADOT_ENTER
    if (!s || !cfg) return -22;
    const int I = s->I, J = s->J, ld = s->ld, S = 5;
    if (cfg->batch_size < 1 || cfg->batch_size > MAX_BATCH * 64) return -22;
    // p = G, q = mean(G), dx = 1/I, dy = 1/J (ot_solvers.py:220-227)
    std::vector<double> g(I, 1.0);
    if (G_host) memcpy(g.data(), G_host, sizeof(double) * I);
    double gs = 0.0;
    for (int i = 0; i < I; i++) gs += g[i];
    HIP_CHECK(hipMemsetAsync(s->a, 0, sizeof(double) * (6 * (size_t)I + 7 * (size_t)ld), s->stream));   // whole vector block
    upload_vec(s, s->p, g.data(), I);
    HIP_CHECK(hipStreamSynchronize(s->stream));   // g is a host temporary
    const int mx = std::max(I, ld);
    dim3 gv((mx + 255) / 256);
    hipLaunchKernelGGL(k_fill, dim3((I + 255) / 256), dim3(256), 0, s->stream, s->dx, 1.0 / I, I);
    hipLaunchKernelGGL(k_fill, dim3((I + 255) / 256), dim3(256), 0, s->stream, s->a, 1.0, I);
    hipLaunchKernelGGL(k_fill, dim3((J + 255) / 256), dim3(256), 0, s->stream, s->dy, 1.0 / J, J);
    hipLaunchKernelGGL(k_fill, dim3((J + 255) / 256), dim3(256), 0, s->stream, s->q, gs / I, J);
    hipLaunchKernelGGL(k_fill, dim3((J + 255) / 256), dim3(256), 0, s->stream, s->b, 1.0, J);
    HIP_CHECK(hipMemsetAsync(s->flags, 0, sizeof(int) * (MAX_BATCH + 1), s->stream));

    // speculative batches (skip the idle tau-absorb launches, vector-only gap)
    const bool spec = true;
    const double f = exp(-log(cfg->epsilon) / S);
    double eps_i = cfg->epsilon0 * f;
    double gap = INFINITY;
    spadot_ot_info rep;
    memset(&rep, 0, sizeof(rep));
    for (int e = 0; e <= S; e++) {
        hipLaunchKernelGGL(k_stage_begin, gv, dim3(256), 0, s->stream, s->a, s->b, s->old_a, s->old_b,
                           s->u, s->v, s->adx, s->w, s->dx, s->dy, eps_i, I, J, ld);
        eps_i = eps_i / f;
        IterParams P{eps_i, cfg->tau, cfg->lambda1, cfg->lambda2, cfg->lambda1 / (cfg->lambda1 + eps_i),
                     cfg->lambda2 / (cfg->lambda2 + eps_i)};
        const double thr = (e == S) ? cfg->tolerance : 1e-6;
        build_K(s, eps_i, nullptr);
        if (e == S) sum_kbar(s, s->C, eps_i, true);   // only the last stage's gap needs sum(Kbar)
        // cur_iter = 0 in EVERY stage, as on the reference's live path: with c_for_v2 = True (ot_solvers.py:17,282-289)
        // each stage is one update_process_c call that receives current_iter, and Python never updates that
        // variable on this path (it is only threaded through step1_process_c on the c_for_v2 = False branch,
        // :345-350) -- so max_iter is a per-stage budget there and here
        gap = process_stage(s, P, e == S, cfg->batch_size, thr, 0, cfg->max_iter, nullptr,
                            &rep.stage_iters[e], &rep.gap_checks, /*fast=*/spec);
    }
    HIP_CHECK(hipMemcpyAsync(s->h_flags, s->flags + MAX_BATCH, sizeof(int), hipMemcpyDeviceToHost, s->stream));
    HIP_CHECK(hipStreamSynchronize(s->stream));
    rep.absorbs = s->h_flags[0];
    rep.gap = gap;
    if (info) *info = rep;
    return std::isnan(gap) ? 1 : 0;
    SPADOT_LEAVE(SPADOT_EHIP)
}

int spadot_ot_plan_dev(spadot_ot_solver *s, void *plan_dev, int dtype, int ldp) {
    SPADOT_ENTER
    if (!s || !plan_dev || ldp < s->J) return -22;
    dim3 g((s->J + 255) / 256, (unsigned)std::min(s->I, 8192));
    const double sc = 1.0 / s->J;
    if (s->storage == SPADOT_F64 && dtype == SPADOT_F64)
        hipLaunchKernelGGL((k_plan<double, double>), g, dim3(256), 0, s->stream, (const double *)s->K, s->a, s->b, sc, (double *)plan_dev, s->I, s->J, s->ld, ldp);
    else if (s->storage == SPADOT_F64 && dtype == SPADOT_F32)
        hipLaunchKernelGGL((k_plan<double, float>), g, dim3(256), 0, s->stream, (const double *)s->K, s->a, s->b, sc, (float *)plan_dev, s->I, s->J, s->ld, ldp);
    else if (s->storage == SPADOT_F32 && dtype == SPADOT_F64)
        hipLaunchKernelGGL((k_plan<float, double>), g, dim3(256), 0, s->stream, (const float *)s->K, s->a, s->b, sc, (double *)plan_dev, s->I, s->J, s->ld, ldp);
    else if (s->storage == SPADOT_F32 && dtype == SPADOT_F32)
        hipLaunchKernelGGL((k_plan<float, float>), g, dim3(256), 0, s->stream, (const float *)s->K, s->a, s->b, sc, (float *)plan_dev, s->I, s->J, s->ld, ldp);
    else
        return -22;
    return 0;
    SPADOT_LEAVE(SPADOT_EHIP)
}

int spadot_ot_plan_group_sums_dev(spadot_ot_solver *s, const int *col_labels_dev, int ngroups, double *Q_dev) {
    SPADOT_ENTER
    if (!s || !col_labels_dev || !Q_dev || ngroups < 1 || ngroups > 64) return -22;
    const size_t lds = sizeof(double) * ROW_WAVES * (size_t)ngroups * WAVE;
    dim3 g((s->I + ROW_WAVES - 1) / ROW_WAVES);
    const double sc = 1.0 / s->J;
    if (s->storage == SPADOT_F32)
        hipLaunchKernelGGL(k_plan_group_sums<float>, g, dim3(256), lds, s->stream, (const float *)s->K, s->a, s->b, col_labels_dev, ngroups, sc, Q_dev, s->I, s->J, s->ld);
    else
        hipLaunchKernelGGL(k_plan_group_sums<double>, g, dim3(256), lds, s->stream, (const double *)s->K, s->a, s->b, col_labels_dev, ngroups, sc, Q_dev, s->I, s->J, s->ld);
    return hipGetLastError() == hipSuccess ? 0 : -5;
    SPADOT_LEAVE(SPADOT_EHIP)
}

int spadot_ot_plan_host(spadot_ot_solver *s, double *plan_host) {
    SPADOT_ENTER
    if (!s || !plan_host) return -22;
    DevBuf buf(sizeof(double) * (size_t)s->I * s->J);
    double *tmp = buf.as<double>();
    int rc = spadot_ot_plan_dev(s, tmp, SPADOT_F64, s->J);
    if (rc == 0)
        HIP_CHECK(hipMemcpyAsync(plan_host, tmp, sizeof(double) * (size_t)s->I * s->J, hipMemcpyDeviceToHost, s->stream));
    HIP_CHECK(hipStreamSynchronize(s->stream));
    return rc;
    SPADOT_LEAVE(SPADOT_EHIP)
}

// rowsum_i = a_i * sum_j K_ij b_j / J  == (row pass with weights b) -- reuses k_gap_rows' rs slot
int spadot_ot_plan_rowsums_host(spadot_ot_solver *s, double *rowsums_host) {
    SPADOT_ENTER
    if (!s || !rowsums_host) return -22;
    // dy-weighted row sums with dy := 1/J are exactly the plan's row sums
    hipLaunchKernelGGL(k_fill, dim3((s->J + 255) / 256), dim3(256), 0, s->stream, s->tcol, 1.0 / s->J, s->J);
    dim3 g((s->I + ROW_WAVES - 1) / ROW_WAVES);
    if (s->storage == SPADOT_F32)
        hipLaunchKernelGGL((k_gap_rows<float, true>), g, dim3(256), 0, s->stream, (const float *)s->K, (const float *)nullptr, s->a, s->b, s->tcol, (float *)nullptr, s->rt, s->I, s->J, s->ld);
    else
        hipLaunchKernelGGL((k_gap_rows<double, true>), g, dim3(256), 0, s->stream, (const double *)s->K, (const double *)nullptr, s->a, s->b, s->tcol, (double *)nullptr, s->rt, s->I, s->J, s->ld);
    download_vec(s, rowsums_host, s->rt, s->I);
    HIP_CHECK(hipStreamSynchronize(s->stream));
    return 0;
    SPADOT_LEAVE(SPADOT_EHIP)
}

int spadot_ot_run_iterations(spadot_ot_solver *s, const spadot_ot_config *cfg, double eps_stage,
                             int iters, float *ms_out) {
    SPADOT_ENTER
    if (!s || !cfg || iters < 0) return -22;
    IterParams P{eps_stage, cfg->tau, cfg->lambda1, cfg->lambda2, cfg->lambda1 / (cfg->lambda1 + eps_stage),
                 cfg->lambda2 / (cfg->lambda2 + eps_stage)};
    // same schedule as the solver's batches minus the convergence measure and its sync: snapshot +
    // `batch_size` fast iterations per batch; a raised tau flag makes the run unusable (return 2)
    const int bs = std::max(1, std::min(cfg->batch_size, MAX_BATCH));
    // (untimed calls leave the flag alone: it accumulates until spadot_ot_run_tau_flag reads it)
    if (ms_out) HIP_CHECK(hipMemsetAsync(s->flags + MAX_BATCH - 1, 0, sizeof(int), s->stream));
    if (ms_out) HIP_CHECK(hipEventRecord(s->ev0, s->stream));      // (untimed calls stay capturable: no event nodes)
    for (int left = iters; left > 0; left -= bs) {
        const int nb = std::min(left, bs);
        snapshot(s);
        for (int t = 0; t < nb; t++) {
            if (s->storage == SPADOT_F32) one_iteration_T<float>(s, P, s->flags + MAX_BATCH - 1, false);
            else one_iteration_T<double>(s, P, s->flags + MAX_BATCH - 1, false);
        }
    }
    if (ms_out) {
        HIP_CHECK(hipEventRecord(s->ev1, s->stream));
        HIP_CHECK(hipMemcpyAsync(s->h_flags, s->flags + MAX_BATCH - 1, sizeof(int), hipMemcpyDeviceToHost, s->stream));
        HIP_CHECK(hipEventSynchronize(s->ev1));
        HIP_CHECK(hipStreamSynchronize(s->stream));
        HIP_CHECK(hipEventElapsedTime(ms_out, s->ev0, s->ev1));
        if (s->h_flags[0] != 0) return 2;
    }
    return 0;
    SPADOT_LEAVE(SPADOT_EHIP)
}

int spadot_ot_run_tau_flag(spadot_ot_solver *s, int reset) {
    SPADOT_ENTER
    if (!s) return -22;
    HIP_CHECK(hipMemcpyAsync(s->h_flags, s->flags + MAX_BATCH - 1, sizeof(int), hipMemcpyDeviceToHost, s->stream));
    if (reset) HIP_CHECK(hipMemsetAsync(s->flags + MAX_BATCH - 1, 0, sizeof(int), s->stream));
    HIP_CHECK(hipStreamSynchronize(s->stream));
    return s->h_flags[0] != 0 ? 1 : 0;
    SPADOT_LEAVE(SPADOT_EHIP)
}

// The solver's real inner loop, `nbatches` times, ignoring the threshold: snapshot, batch_size (last stage)
// or 5 iterations, convergence measure, 36-byte read-back + stream sync, redo in safe mode if tau fired.
// iters_out = scaling iterations run; ms_out = host wall time of the whole call in milliseconds.
int spadot_ot_run_checked(spadot_ot_solver *s, const spadot_ot_config *cfg, double eps_stage, int last_stage,
                          int nbatches, int *iters_out, float *ms_out) {
    SPADOT_ENTER
    if (!s || !cfg || nbatches < 1) return -22;
    IterParams P{eps_stage, cfg->tau, cfg->lambda1, cfg->lambda2, cfg->lambda1 / (cfg->lambda1 + eps_stage),
                 cfg->lambda2 / (cfg->lambda2 + eps_stage)};
    if (last_stage) sum_kbar(s, s->C, eps_stage, true);
    HIP_CHECK(hipStreamSynchronize(s->stream));
    const int iters = last_stage ? cfg->batch_size : 5;
    int ran = nbatches * iters;
    HIP_CHECK(hipEventRecord(s->ev0, s->stream));
    if (last_stage && spec_batches() > 1 && pipe_gap() && s->fused_vpt > 0 && iters <= MAX_BATCH) {
        // what spadot_ot_solve's last stage runs: groups of speculated batches, the gap measure taken from the next
        // batch's first pass, one 16-byte read-back per group; a threshold below every measure keeps it going
        double gap = 1e100;
        int cur = 0, done = 0, checks = 0;
        const int SPEC = spec_batches();
        pipelined_gap_loop(s, P, iters, -INFINITY, SPEC, 1 << 30, (nbatches + SPEC - 1) / SPEC, gap, cur, done, checks);
        ran = done;
    } else {
        for (int k = 0; k < nbatches; k++) {
            snapshot(s);
            run_iterations(s, P, iters, false);
            if (last_stage) gap_measure_fast(s, P); else drift_measure(s, P.eps);
            read_gap(s);
            if (s->h_flags[0] != 0) {
                restore(s);
                run_iterations(s, P, iters, true);
                if (last_stage) gap_measure_fast(s, P); else drift_measure(s, P.eps);
                read_gap(s);
            }
        }
    }
    HIP_CHECK(hipEventRecord(s->ev1, s->stream));
    HIP_CHECK(hipEventSynchronize(s->ev1));
    if (ms_out) HIP_CHECK(hipEventElapsedTime(ms_out, s->ev0, s->ev1));
    if (iters_out) *iters_out = ran;
    return 0;
    SPADOT_LEAVE(SPADOT_EHIP)
}

// Per-kernel live timing: each kernel of one scaling iteration launched `reps` times back to back
// between two HIP events on the solver's stream.  ms_out[0..3] = row pass, column pass, column
// finalise, tau-absorb pair (flag clear => early exit), and -- when the shape allows the fused
// single-sweep path -- ms_out[4] = fused pass, ms_out[5] = its column finalise (else 0).
int spadot_ot_time_kernels(spadot_ot_solver *s, const spadot_ot_config *cfg, double eps_stage, int reps,
                           float *ms_out) {
    SPADOT_ENTER
    if (!s || !cfg || !ms_out || reps < 1) return -22;
    IterParams P{eps_stage, cfg->tau, cfg->lambda1, cfg->lambda2, cfg->lambda1 / (cfg->lambda1 + eps_stage),
                 cfg->lambda2 / (cfg->lambda2 + eps_stage)};
    const int I = s->I, J = s->J, ld = s->ld;
    const int V = s->storage == SPADOT_F32 ? 4 : 2;
    HIP_CHECK(hipMemsetAsync(s->flags, 0, sizeof(int) * MAX_BATCH, s->stream));
    int *flag = s->flags;
    ms_out[4] = ms_out[5] = 0.f;
    for (int which = 0; which < 4; which++) {
        for (int phase = 0; phase < 2; phase++) {          // phase 0 = untimed warm-up
            const int n = phase == 0 ? 2 : reps;
            if (phase == 1) HIP_CHECK(hipEventRecord(s->ev0, s->stream));
            for (int r = 0; r < n; r++) {
                if (which == 0) {
                    if (s->storage == SPADOT_F32)
                        hipLaunchKernelGGL(k_row_pass<float>, dim3((I + ROW_WAVES - 1) / ROW_WAVES), dim3(256), 0, s->stream, (const float *)s->K, s->w, s->a, s->old_a, s->adx, s->p, s->dx, s->u, P.al1, 1.0 / (P.l1 + P.eps), P.tau, I, ld, flag);
                    else
                        hipLaunchKernelGGL(k_row_pass<double>, dim3((I + ROW_WAVES - 1) / ROW_WAVES), dim3(256), 0, s->stream, (const double *)s->K, s->w, s->a, s->old_a, s->adx, s->p, s->dx, s->u, P.al1, 1.0 / (P.l1 + P.eps), P.tau, I, ld, flag);
                } else if (which == 1) {
                    if (s->storage == SPADOT_F32)
                        hipLaunchKernelGGL(k_col_pass<float>, dim3((ld + 256 * V - 1) / (256 * V), s->nchunk), dim3(256), 0, s->stream, (const float *)s->K, s->adx, s->part, I, ld, s->rows_per_chunk);
                    else
                        hipLaunchKernelGGL(k_col_pass<double>, dim3((ld + 256 * V - 1) / (256 * V), s->nchunk), dim3(256), 0, s->stream, (const double *)s->K, s->adx, s->part, I, ld, s->rows_per_chunk);
                } else if (which == 2) {
                    hipLaunchKernelGGL(k_col_fin, dim3((ld + 255) / 256), dim3(256), 0, s->stream, s->part, s->nchunk, s->b, s->old_b, s->w, s->q, s->dy, s->v, P.al2, 1.0 / (P.l2 + P.eps), P.tau, J, ld, flag, (double *)nullptr, 0);
                } else {
                    absorb_if_flagged(s, P, s->flags + MAX_BATCH - 1);
                }
            }
            if (phase == 1) {
                HIP_CHECK(hipEventRecord(s->ev1, s->stream));
                HIP_CHECK(hipEventSynchronize(s->ev1));
                float ms = 0.f;
                HIP_CHECK(hipEventElapsedTime(&ms, s->ev0, s->ev1));
                ms_out[which] = ms / reps;
            }
        }
    }
    if (s->fused_vpt > 0) {
        // the fused pass and its column finalise are timed IN PLACE: `reps` real iterations with an event
        // before the pass, between the two launches and after the finalise (so each launch sees the cache
        // state it has inside a solve, not the one of a back-to-back replay of itself)
        std::vector<hipEvent_t> ev(3 * (size_t)reps);
        for (auto &e : ev) HIP_CHECK(hipEventCreate(&e));
        run_iterations(s, P, 2, false);                    // warm-up
        for (int r = 0; r < reps; r++) {
            HIP_CHECK(hipEventRecord(ev[3 * r], s->stream));
            if (s->storage == SPADOT_F32) fused_pass_T<float>(s, P, flag);
            else fused_pass_T<double>(s, P, flag);
            HIP_CHECK(hipEventRecord(ev[3 * r + 1], s->stream));
            if (s->storage == SPADOT_F32)
                hipLaunchKernelGGL(k_col_fin2<float>, dim3((ld + 63) / 64), dim3(1024), 0, s->stream, (const float *)s->part, s->fused_blocks, s->b, s->old_b, s->w, s->q, s->dy, s->v, P.al2, 1.0 / (P.l2 + P.eps), P.tau, J, ld, flag, s->tcol, 0, (const int *)nullptr);
            else
                hipLaunchKernelGGL(k_col_fin2<double>, dim3((ld + 63) / 64), dim3(1024), 0, s->stream, (const double *)s->part, s->fused_blocks, s->b, s->old_b, s->w, s->q, s->dy, s->v, P.al2, 1.0 / (P.l2 + P.eps), P.tau, J, ld, flag, s->tcol, 0, (const int *)nullptr);
            HIP_CHECK(hipEventRecord(ev[3 * r + 2], s->stream));
        }
        HIP_CHECK(hipStreamSynchronize(s->stream));
        double t_pass = 0.0, t_fin = 0.0;
        for (int r = 0; r < reps; r++) {
            float a_ms = 0.f, b_ms = 0.f;
            HIP_CHECK(hipEventElapsedTime(&a_ms, ev[3 * r], ev[3 * r + 1]));
            HIP_CHECK(hipEventElapsedTime(&b_ms, ev[3 * r + 1], ev[3 * r + 2]));
            t_pass += a_ms; t_fin += b_ms;
        }
        ms_out[4] = (float)(t_pass / reps);
        ms_out[5] = (float)(t_fin / reps);
        for (auto &e : ev) (void)hipEventDestroy(e);
    }
    return 0;
    SPADOT_LEAVE(SPADOT_EHIP)
}

// ============================================================================================
// PART A: libot.so-compatible entry points (host pointers).  Each call builds a temporary device
// problem, runs the same kernels, and writes the mutated arrays back.
// ============================================================================================
}  // extern "C"

namespace {
struct Tmp {
    spadot_ot_solver *s = nullptr;
    Tmp(int m, int n, int storage) {
        if (spadot_ot_create(&s, m, n, storage, nullptr) != 0)
            throw std::runtime_error("cannot create the temporary device problem of a libot-compatible call");
    }
    ~Tmp() { spadot_ot_destroy(s); }
};

// float-vector helpers for the *_float entry points (vectors are widened to fp64 on the host)
std::vector<double> widen(const float *x, int n) {
    std::vector<double> o(n);
    for (int i = 0; i < n; i++) o[i] = x[i];
    return o;
}

template <typename T>
void compat_update_k(T *K, T *K_, T *C, const double *u, const double *v, double eps, int m, int n) {
    Tmp t(m, n, std::is_same<T, float>::value ? SPADOT_F32 : SPADOT_F64);
    spadot_ot_solver *s = t.s;
    upload_mat<T>(s, s->C, C);
    upload_vec(s, s->u, u, m);
    upload_vec(s, s->v, v, n);
    DevBuf kbuf((size_t)m * s->ld * sizeof(T));
    void *kb = kbuf.p;
    constexpr int V = Vec<T>::N;
    hipLaunchKernelGGL(k_build_Kbar<T>, grid_cols(s, V), dim3(256), 0, s->stream, (T *)kb, (const T *)s->C,
                       eps, m, n, s->ld);
    launch_build_K<T>(s, eps, nullptr);
    download_mat<T>(s, K_, kb);
    download_mat<T>(s, K, s->K);
    HIP_CHECK(hipStreamSynchronize(s->stream));
}

template <typename T>
void compat_update_R(T *R, T *K, const double *a, const double *b, int m, int n) {
    Tmp t(m, n, std::is_same<T, float>::value ? SPADOT_F32 : SPADOT_F64);
    spadot_ot_solver *s = t.s;
    upload_mat<T>(s, s->K, K);
    upload_vec(s, s->a, a, m);
    upload_vec(s, s->b, b, n);
    hipLaunchKernelGGL(k_fill, dim3((n + 255) / 256), dim3(256), 0, s->stream, s->dy, 0.0, n);
    hipLaunchKernelGGL((k_gap_rows<T, true>), dim3((m + ROW_WAVES - 1) / ROW_WAVES), dim3(256), 0, s->stream,
                       (const T *)s->K, (const T *)nullptr, s->a, s->b, s->dy, (T *)s->C, s->rt, m, n, s->ld);
    download_mat<T>(s, R, s->C);
    HIP_CHECK(hipStreamSynchronize(s->stream));
}

// which: 0 = gap, 1 = primal, 2 = dual.  R given as a matrix; a, b are the (un-stabilised) scalings.
template <typename T>
double compat_gap(int which, T *C, T *Kbar, T *R, const double *dx, const double *dy, const double *p,
                  const double *q, const double *a, const double *b, double eps, double l1, double l2,
                  int m, int n) {
    Tmp t(m, n, std::is_same<T, float>::value ? SPADOT_F32 : SPADOT_F64);
    spadot_ot_solver *s = t.s;
    constexpr int V = Vec<T>::N;
    upload_mat<T>(s, s->C, C);
    upload_mat<T>(s, s->K, Kbar);                 // K slot holds Kbar just long enough to sum it
    sum_kbar(s, s->K, eps, false);
    upload_mat<T>(s, s->K, R);                    // ... then holds R
    upload_vec(s, s->dx, dx, m); upload_vec(s, s->dy, dy, n);
    upload_vec(s, s->p, p, m);   upload_vec(s, s->q, q, n);
    upload_vec(s, s->a, a, m);   upload_vec(s, s->b, b, n);
    hipLaunchKernelGGL((k_gap_rows<T, false>), dim3((m + ROW_WAVES - 1) / ROW_WAVES), dim3(256), 0, s->stream,
                       (const T *)s->K, (const T *)s->C, s->a, s->b, s->dy, (T *)nullptr, s->rt, m, n, s->ld);
    // cs = R^T dx
    hipLaunchKernelGGL(k_col_pass<T>, dim3((s->ld + 256 * V - 1) / (256 * V), s->nchunk), dim3(256), 0,
                       s->stream, (const T *)s->K, s->dx, s->part, m, s->ld, s->rows_per_chunk);
    hipLaunchKernelGGL(k_col_fin, dim3((s->ld + 255) / 256), dim3(256), 0, s->stream, s->part, s->nchunk,
                       (double *)nullptr, (double *)nullptr, (double *)nullptr, (const double *)nullptr,
                       (const double *)nullptr, (const double *)nullptr, 0.0, 0.0, 0.0, n, s->ld,
                       (int *)nullptr, s->tcol, 1);
    hipLaunchKernelGGL(k_gap_fin, dim3(1), dim3(1024), 0, s->stream, s->rt, s->tcol, s->a, s->b, s->u, s->v,
                       s->p, s->q, s->dx, s->dy, eps, l1, l2, m, n, 0, 0, s->scal);
    read_gap(s);
    return s->h_scal[which];
}
}  // namespace

extern "C" {

void update_k_double(double *K, double *K_, double *C, double *u, double *v, double epsilon, int m, int n) {
    SPADOT_ENTER
    compat_update_k<double>(K, K_, C, u, v, epsilon, m, n);
    SPADOT_LEAVE()
}
void update_k_float(float *K, float *K_, float *C, float *u, float *v, float epsilon, int m, int n) {
    SPADOT_ENTER
    auto uu = widen(u, m), vv = widen(v, n);
    compat_update_k<float>(K, K_, C, uu.data(), vv.data(), (double)epsilon, m, n);
    SPADOT_LEAVE()
}
void update_R_double(double *R, double *K, double *a, double *b, int m, int n) {
    SPADOT_ENTER
    compat_update_R<double>(R, K, a, b, m, n);
    SPADOT_LEAVE()
}
void update_R_float(float *R, float *K, float *a, float *b, int m, int n) {
    SPADOT_ENTER
    auto aa = widen(a, m), bb = widen(b, n);
    compat_update_R<float>(R, K, aa.data(), bb.data(), m, n);
    SPADOT_LEAVE()
}

double primal_double(double *C, double *K, double *R, double *dx, double *dy, double *p, double *q,
                     double *a, double *b, double epsilon, double lambda1, double lambda2, int m, int n) {
    SPADOT_ENTER
    return compat_gap<double>(1, C, K, R, dx, dy, p, q, a, b, epsilon, lambda1, lambda2, m, n);
    SPADOT_LEAVE(NAN)
}
double dual_double(double *C, double *K, double *R, double *dx, double *dy, double *p, double *q,
                   double *a, double *b, double epsilon, double lambda1, double lambda2, int m, int n) {
    SPADOT_ENTER
    return compat_gap<double>(2, C, K, R, dx, dy, p, q, a, b, epsilon, lambda1, lambda2, m, n);
    SPADOT_LEAVE(NAN)
}
double compute_duality_gap_double(double *C, double *K, double *R, double *dx, double *dy, double *p,
                                  double *q, double *a, double *b, double epsilon, double lambda1,
                                  double lambda2, int m, int n) {
    SPADOT_ENTER
    return compat_gap<double>(0, C, K, R, dx, dy, p, q, a, b, epsilon, lambda1, lambda2, m, n);
    SPADOT_LEAVE(NAN)
}
#define FLOAT_GAP(which)                                                                              \
    SPADOT_ENTER                                                                                      \
    auto dx_ = widen(dx, m), dy_ = widen(dy, n), p_ = widen(p, m), q_ = widen(q, n), a_ = widen(a, m), \
         b_ = widen(b, n);                                                                            \
    return (float)compat_gap<float>(which, C, K, R, dx_.data(), dy_.data(), p_.data(), q_.data(),     \
                                    a_.data(), b_.data(), (double)epsilon, (double)lambda1,           \
                                    (double)lambda2, m, n);                                           \
    SPADOT_LEAVE(NAN)
float primal_float(float *C, float *K, float *R, float *dx, float *dy, float *p, float *q, float *a,
                   float *b, float epsilon, float lambda1, float lambda2, int m, int n) { FLOAT_GAP(1) }
float dual_float(float *C, float *K, float *R, float *dx, float *dy, float *p, float *q, float *a,
                 float *b, float epsilon, float lambda1, float lambda2, int m, int n) { FLOAT_GAP(2) }
float compute_duality_gap_float(float *C, float *K, float *R, float *dx, float *dy, float *p, float *q,
                                float *a, float *b, float epsilon, float lambda1, float lambda2, int m,
                                int n) { FLOAT_GAP(0) }
#undef FLOAT_GAP

float dummy_float(float *, float *, float *, float *, float *, float *, float *, float *, float *, float,
                  float, float, int, int) { return 0; }
double dummy_double(double *, double *, double *, double *, double *, double *, double *, double *,
                    double *, double, double, double, int, int) { return 0; }

}  // extern "C"

namespace {
// shared upload/download for step1_process_double / update_process_double
void compat_load_state(spadot_ot_solver *s, double *a, double *b, double *old_a, double *old_b, double *K,
                       double *C, double *dx, double *dy, double *p, double *q, double *u, double *v) {
    const int m = s->I, n = s->J;
    upload_mat<double>(s, s->C, C);
    upload_mat<double>(s, s->K, K);
    upload_vec(s, s->a, a, m); upload_vec(s, s->old_a, old_a, m); upload_vec(s, s->u, u, m);
    upload_vec(s, s->p, p, m); upload_vec(s, s->dx, dx, m);
    upload_vec(s, s->b, b, n); upload_vec(s, s->old_b, old_b, n); upload_vec(s, s->v, v, n);
    upload_vec(s, s->q, q, n); upload_vec(s, s->dy, dy, n);
    const int mx = std::max(m, s->ld);
    hipLaunchKernelGGL(k_prep_vec, dim3((mx + 255) / 256), dim3(256), 0, s->stream, s->a, s->b, s->dx, s->dy,
                       s->adx, s->w, m, n, s->ld);
}
void compat_store_state(spadot_ot_solver *s, double *a, double *b, double *old_a, double *old_b, double *K,
                        double *u, double *v) {
    const int m = s->I, n = s->J;
    download_mat<double>(s, K, s->K);
    download_vec(s, a, s->a, m); download_vec(s, old_a, s->old_a, m); download_vec(s, u, s->u, m);
    download_vec(s, b, s->b, n); download_vec(s, old_b, s->old_b, n); download_vec(s, v, s->v, n);
}
}  // namespace

extern "C" {

int step1_process_double(double *a, double *b, double *old_a, double *old_b, double *K, double *C,
                         double *dx, double *dy, double *p, double *q, double *u, double *v,
                         int cur_iter, int max_iter, int iters, double tau, double lambda1,
                         double lambda2, double alpha1, double alpha2, double epsilon, int m, int n) {
    SPADOT_ENTER
    Tmp t(m, n, SPADOT_F64);
    spadot_ot_solver *s = t.s;
    compat_load_state(s, a, b, old_a, old_b, K, C, dx, dy, p, q, u, v);
    IterParams P{epsilon, tau, lambda1, lambda2, alpha1, alpha2};
    int run = iters, ret = cur_iter + iters;
    if (iters > 0 && cur_iter + iters >= max_iter) {
        run = std::max(1, std::min(iters, max_iter - cur_iter));
        ret = -1;
    }
    run_iterations(s, P, run);
    compat_store_state(s, a, b, old_a, old_b, K, u, v);
    HIP_CHECK(hipStreamSynchronize(s->stream));
    if (ret == -1) printf("Reached max_iter with duality gap still above threshold. Returning");
    return ret;
    SPADOT_LEAVE(SPADOT_EHIP)
}

double update_process_double(double *R, double *a, double *b, double *old_a, double *old_b, double *K,
                             double *_K, double *C, double *dx, double *dy, double *p, double *q,
                             double *u, double *v, int epsilon_scalings, int cur_epsilon_scaling,
                             int batch_size, double epsilon, double threshold, double tau,
                             double lambda1, double lambda2, double alpha1, double alpha2,
                             int cur_iter, int max_iter, int m, int n) {
    SPADOT_ENTER
    Tmp t(m, n, SPADOT_F64);
    spadot_ot_solver *s = t.s;
    const bool last = (cur_epsilon_scaling == epsilon_scalings);
    DevBuf rbuf(last ? (size_t)m * s->ld * sizeof(double) : 8);
    void *Rdev = nullptr;
    if (last) {
        // sum(_K) from the caller's matrix: the K slot is free until the state is loaded
        upload_mat<double>(s, s->K, _K);
        sum_kbar(s, s->K, epsilon, false);
        Rdev = rbuf.p;
    }
    compat_load_state(s, a, b, old_a, old_b, K, C, dx, dy, p, q, u, v);
    IterParams P{epsilon, tau, lambda1, lambda2, alpha1, alpha2};
    const double gap = process_stage(s, P, last, batch_size, threshold, cur_iter, max_iter, Rdev, nullptr, nullptr);
    compat_store_state(s, a, b, old_a, old_b, K, u, v);
    if (last) download_mat<double>(s, R, Rdev);
    HIP_CHECK(hipStreamSynchronize(s->stream));
    return gap;
    SPADOT_LEAVE(NAN)
}

}  // extern "C"
