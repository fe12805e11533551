// gemm_f64.hip -- small batched fp64 products of the SVGP branch on the fp64 matrix cores (round 4).
//
// svgp.py:62-104 of the reference, restated in spadot_amd/model/svgp.py (_SVGPCore): per step nine fp64 products of the
// shapes [L][m x b] . [b x m], [L][2b x m] . [m x m], [L x b] . [b x m] (L = 10 latent dimensions, b = 512 seeds, m ~ 240
// inducing points: 0.002 - 1.1 GFLOP each).  The library ran them with 64 x 32 / 16 x 16 macro tiles at 6 - 40 TFLOP/s,
// 13 - 95 us each, 400 us per step, on the latency-bound side stream of the step.
//
// k_dgemm_small<MODE>: C[z] = alpha op(A[z]) op(B[z]) (+ beta C0[z]), 256 threads = 4 wavefronts, a 64 x 64 tile of C per
// workgroup (wave w: rows 16 w .. 16 w + 15, four 16 x 16 accumulators), contraction in steps of 32 through LDS
// (2 x 17 KB), v_mfma_f64_16x16x4_f64 (A: row = lane & 15, k = lane >> 4; B: col = lane & 15, k = lane >> 4; C/D: col =
// lane & 15, row = (lane >> 4) + 4 reg: cdna_hip_programming.md).  The next step's operands are fetched into registers
// while the current one is multiplied.  ~110 registers, 35 KB of LDS: two to four workgroups per compute unit, so a grid
// slots in beside the GAT branch's GEMMs.  Optional row scaling of A along the contraction (TN mode): A'[k][i] = A[k][i] rs[z][k]
// -- K_mn diag(w_l) K_nm without materialising diag(w_l) K_nm.  Fixed summation order (k ascending): bit-repeatable.
//   MODE 0 (NN): a(i, k) = A[i lda + k], b(k, j) = B[k ldb + j]
//   MODE 1 (NT): a(i, k) = A[i lda + k], b(k, j) = B[j ldb + k]
//   MODE 2 (TN): a(i, k) = A[k lda + i], b(k, j) = B[k ldb + j]
#include <hip/hip_runtime.h>
#include <cstdint>

#include "../../include/spadot_model.h"

namespace {

typedef double double4_t __attribute__((ext_vector_type(4)));

constexpr int BM = 64, BN = 64, BK = 32, LDP = 68;    // LDS row pitch (doubles): 64 + 4

struct DgemmArgs {
    const double *A, *B, *C0, *rs;
    double *C;
    int lda, ldb, ldc, ldc0, ldrs;
    long long sA, sB, sC, sC0, srs;
    double alpha, beta;
    int M, N, K;
};

// One BK x 64 operand slab into registers: `kmajor` sources have the contraction index as their ROW (stride ld), the
// 64 tile entries contiguous; the others have the tile index as the row and the contraction index contiguous.
// Register r of thread t holds slab element (k, c):  kmajor: k = (t >> 4) + 16 (r >> 2), c = 4 (t & 15) + (r & 3)
//                                                     else:   c = t >> 2,               k = 8 (t & 3) + r
template <bool KMAJOR>
__device__ __forceinline__ void fetch(const double *__restrict__ P, int ld, int c0, int k0, int nC, int nK, double *v, int t) {
    if (KMAJOR) {
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const int k = k0 + (t >> 4) + 16 * h;
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const int c = c0 + 4 * (t & 15) + e;
                v[4 * h + e] = (k < nK && c < nC) ? P[(size_t)k * ld + c] : 0.0;
            }
        }
    } else {
        const int c = c0 + (t >> 2);
#pragma unroll
        for (int e = 0; e < 8; e++) {
            const int k = k0 + 8 * (t & 3) + e;
            v[e] = (k < nK && c < nC) ? P[(size_t)c * ld + k] : 0.0;
        }
    }
}
template <bool KMAJOR> __device__ __forceinline__ void stash(double (*S)[LDP], const double *v, int t) {
    if (KMAJOR) {
#pragma unroll
        for (int h = 0; h < 2; h++)
#pragma unroll
            for (int e = 0; e < 4; e++) S[(t >> 4) + 16 * h][4 * (t & 15) + e] = v[4 * h + e];
    } else {
#pragma unroll
        for (int e = 0; e < 8; e++) S[8 * (t & 3) + e][t >> 2] = v[e];
    }
}

template <int MODE>
__global__ __launch_bounds__(256, 2) void k_dgemm_small(DgemmArgs g) {
    __shared__ double As[BK][LDP], Bs[BK][LDP];
    constexpr bool A_KMAJOR = MODE == 2, B_KMAJOR = MODE != 1;
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int z = blockIdx.z;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    const double *A = g.A + (size_t)z * g.sA, *B = g.B + (size_t)z * g.sB;
    const double *rs = g.rs ? g.rs + (size_t)z * g.srs : nullptr;
    double4_t acc[4];
#pragma unroll
    for (int j = 0; j < 4; j++) acc[j] = (double4_t){0.0, 0.0, 0.0, 0.0};
    double va[8], vb[8];
    fetch<A_KMAJOR>(A, g.lda, m0, 0, g.M, g.K, va, t);
    fetch<B_KMAJOR>(B, g.ldb, n0, 0, g.N, g.K, vb, t);
    if (A_KMAJOR && rs) {
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const int k = (t >> 4) + 16 * h;
            const double s = k < g.K ? rs[(size_t)k * g.ldrs] : 0.0;
#pragma unroll
            for (int e = 0; e < 4; e++) va[4 * h + e] *= s;
        }
    }
    for (int k0 = 0; k0 < g.K; k0 += BK) {
        __syncthreads();                                  // the previous step's readers are done with the LDS slabs
        stash<A_KMAJOR>(As, va, t);
        stash<B_KMAJOR>(Bs, vb, t);
        __syncthreads();
        if (k0 + BK < g.K) {                              // next step's operands travel while this one is multiplied
            fetch<A_KMAJOR>(A, g.lda, m0, k0 + BK, g.M, g.K, va, t);
            fetch<B_KMAJOR>(B, g.ldb, n0, k0 + BK, g.N, g.K, vb, t);
            if (A_KMAJOR && rs) {
#pragma unroll
                for (int h = 0; h < 2; h++) {
                    const int k = k0 + BK + (t >> 4) + 16 * h;
                    const double s = k < g.K ? rs[(size_t)k * g.ldrs] : 0.0;
#pragma unroll
                    for (int e = 0; e < 4; e++) va[4 * h + e] *= s;
                }
            }
        }
#pragma unroll
        for (int kk = 0; kk < BK / 4; kk++) {
            const double a = As[4 * kk + (lane >> 4)][16 * w + (lane & 15)];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const double b = Bs[4 * kk + (lane >> 4)][16 * j + (lane & 15)];
                acc[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[j], 0, 0, 0);
            }
        }
    }
    double *C = g.C + (size_t)z * g.sC;
    const double *C0 = g.C0 ? g.C0 + (size_t)z * g.sC0 : nullptr;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int col = n0 + 16 * j + (lane & 15);
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int row = m0 + 16 * w + (lane >> 4) + 4 * r;
            if (row < g.M && col < g.N) {
                double v = g.alpha * acc[j][r];
                if (C0) v += g.beta * C0[(size_t)row * g.ldc0 + col];
                C[(size_t)row * g.ldc + col] = v;
            }
        }
    }
}

}  // namespace

extern "C" int spadot_dgemm_small(int mode, const double *A, int lda, long long strideA, const double *B, int ldb, long long strideB,
                                  double *C, int ldc, long long strideC, const double *C0, int ldc0, long long strideC0,
                                  const double *rowscale, int ldrs, long long stride_rs, double alpha, double beta, int M, int N, int K,
                                  int batch, void *stream) {
    if (mode < 0 || mode > 2 || !A || !B || !C || M <= 0 || N <= 0 || K <= 0 || batch < 1 || batch > 65535 || ldc < N) return -22;
    if ((mode == 2 ? lda < M : lda < K) || (mode == 1 ? ldb < K : ldb < N) || (C0 && ldc0 < N)) return -22;
    if (strideA < 0 || strideB < 0 || strideC < 0 || strideC0 < 0 || stride_rs < 0 || (rowscale && (mode != 2 || ldrs < 1))) return -22;
    DgemmArgs g;
    g.A = A; g.B = B; g.C0 = C0; g.rs = rowscale; g.C = C;
    g.lda = lda; g.ldb = ldb; g.ldc = ldc; g.ldc0 = ldc0; g.ldrs = ldrs;
    g.sA = strideA; g.sB = strideB; g.sC = strideC; g.sC0 = strideC0; g.srs = stride_rs;
    g.alpha = alpha; g.beta = beta; g.M = M; g.N = N; g.K = K;
    const dim3 grid((unsigned)((N + BN - 1) / BN), (unsigned)((M + BM - 1) / BM), (unsigned)batch);
    if (grid.y > 65535u) return -22;
    hipStream_t st = (hipStream_t)stream;
    if (mode == 0) hipLaunchKernelGGL(k_dgemm_small<0>, grid, dim3(256), 0, st, g);
    else if (mode == 1) hipLaunchKernelGGL(k_dgemm_small<1>, grid, dim3(256), 0, st, g);
    else hipLaunchKernelGGL(k_dgemm_small<2>, grid, dim3(256), 0, st, g);
    return hipGetLastError() == hipSuccess ? 0 : -5;
}
