// gemm_f32.hip -- fp32 products y = x W^T of the small MLP stages with the contraction cut into slices (round 4, VERDICT r03
// item 4b).  encoder.py:7-34 of the reference: the SVGP encoder's first map is 512 seeds x 3000 genes -> 256; the library ran
// it as ONE round of 32 x 16 macro tiles walking all 3000 columns: 57 - 72 us for 0.8 GFLOP at the head of the step's
// forward critical chain (rocprofv3 timeline, round 3).
//
// k_sgemm_nt_slices: C_part[z] = A[:, k_z : k_z+1] B[:, k_z : k_z+1]^T for slice z of the contraction; 64 x 64 tile of C per
// workgroup, 256 threads x (4 x 4) outputs, K in steps of 16 through LDS (k-major, pitch 68: conflict-free stores and
// 16-byte operand reads), next step's operands in flight while the current one is multiplied.  k_slices_sum adds the
// partials in slice order (+ bias): fixed order, bit-repeatable.  Grid = N/64 x M/64 x slices: 8 x 4 x 12 = 384 workgroups
// of 8.5 KB LDS / ~60 registers for the first map -- they slot in beside the GAT branch's GEMM.
#include <hip/hip_runtime.h>
#include <cstdint>

#include "../../include/spadot_model.h"
#include "per_device.h"

namespace {

constexpr int TM = 64, TN = 64, TK = 16, PITCH = 68;

// rows r = t >> 2 of the 64-row operand slab, four consecutive k starting at k0 + 4 (t & 3), zero past the ends
__device__ __forceinline__ float4 fetch4(const float *__restrict__ P, int ld, int row, int nrows, int k, int kend, bool vec_ok) {
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (row < nrows) {
        const float *p = P + (size_t)row * ld + k;
        if (vec_ok && k + 3 < kend) v = *reinterpret_cast<const float4 *>(p);
        else {
            if (k < kend) v.x = p[0];
            if (k + 1 < kend) v.y = p[1];
            if (k + 2 < kend) v.z = p[2];
            if (k + 3 < kend) v.w = p[3];
        }
    }
    return v;
}

__global__ __launch_bounds__(256) void k_sgemm_nt_slices(const float *__restrict__ A, int lda, const float *__restrict__ B, int ldb,
                                                         float *__restrict__ part, int M, int N, int K, int kslice, int vec_ok) {
    __shared__ float As[TK][PITCH], Bs[TK][PITCH];
    const int t = threadIdx.x, tx = t & 15, ty = t >> 4;
    const int m0 = blockIdx.y * TM, n0 = blockIdx.x * TN, z = blockIdx.z;
    const int kbeg = z * kslice, kend = min(K, kbeg + kslice);
    const int lr = t >> 2, lk = 4 * (t & 3);
    float acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) acc[i][j] = 0.f;
    float4 va = fetch4(A, lda, m0 + lr, M, kbeg + lk, kend, vec_ok != 0);
    float4 vb = fetch4(B, ldb, n0 + lr, N, kbeg + lk, kend, vec_ok != 0);
    for (int k0 = kbeg; k0 < kend; k0 += TK) {
        __syncthreads();
        As[lk][lr] = va.x; As[lk + 1][lr] = va.y; As[lk + 2][lr] = va.z; As[lk + 3][lr] = va.w;
        Bs[lk][lr] = vb.x; Bs[lk + 1][lr] = vb.y; Bs[lk + 2][lr] = vb.z; Bs[lk + 3][lr] = vb.w;
        __syncthreads();
        if (k0 + TK < kend) {
            va = fetch4(A, lda, m0 + lr, M, k0 + TK + lk, kend, vec_ok != 0);
            vb = fetch4(B, ldb, n0 + lr, N, k0 + TK + lk, kend, vec_ok != 0);
        }
#pragma unroll
        for (int k = 0; k < TK; k++) {
            const float4 a = *reinterpret_cast<const float4 *>(&As[k][4 * ty]);
            const float4 b = *reinterpret_cast<const float4 *>(&Bs[k][4 * tx]);
            const float av[4] = {a.x, a.y, a.z, a.w}, bv[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
            for (int i = 0; i < 4; i++)
#pragma unroll
                for (int j = 0; j < 4; j++) acc[i][j] = fmaf(av[i], bv[j], acc[i][j]);
        }
    }
    float *C = part + (size_t)z * M * N;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int m = m0 + 4 * ty + i;
        if (m >= M) continue;
        const int n = n0 + 4 * tx;
        if (n + 3 < N && (N & 3) == 0) *reinterpret_cast<float4 *>(C + (size_t)m * N + n) = make_float4(acc[i][0], acc[i][1], acc[i][2], acc[i][3]);
        else
#pragma unroll
            for (int j = 0; j < 4; j++)
                if (n + j < N) C[(size_t)m * N + n + j] = acc[i][j];
    }
}

// The same tile with its WHOLE slice of the contraction (<= 256 columns) staged in LDS at once: every global load of the
// workgroup is issued before the first is consumed -- one memory latency per workgroup instead of one per 16-column step.
// Under the GAT branch's GEMM a dependent load costs 3-5 us; the stepped form above pays 16 of them per workgroup (65 us for
// the first map against the library's 47-57), this one pays one.  LDS: 2 x 256 x 68 floats = 136 KB (dynamic), k-major.
// Load mapping: thread t takes rows (t & 15) + 16 (u & 3) and the float4 at k = 4 ((t >> 4) + 16 (u >> 2)), u < 16: the 64
// lanes of a wave store 16 rows x 4 k-groups whose banks (4 k + row) mod 64 are all different (conflict-free transpose).
constexpr int OS_KMAX = 256;
__global__ __launch_bounds__(256) void k_sgemm_nt_oneshot(const float *__restrict__ A, int lda, const float *__restrict__ B, int ldb,
                                                          float *__restrict__ part, int M, int N, int K, int kslice, int vec_ok,
                                                          const float *__restrict__ bias, int direct_ld) {
    extern __shared__ float os_lds[];
    float (*As)[PITCH] = reinterpret_cast<float (*)[PITCH]>(os_lds);
    float (*Bs)[PITCH] = reinterpret_cast<float (*)[PITCH]>(os_lds + OS_KMAX * PITCH);
    const int t = threadIdx.x, tx = t & 15, ty = t >> 4;
    const int m0 = blockIdx.y * TM, n0 = blockIdx.x * TN, z = blockIdx.z;
    const int kbeg = z * kslice, kend = min(K, kbeg + kslice), kn = kend - kbeg;
    const int r16 = t & 15, kq = t >> 4;
    float4 va[16], vb[16];
    // (K % 4 == 0 and 16-byte aligned rows are this kernel's conditions: a float4 unit is then wholly inside or wholly outside
    // the slice.  Units outside are loaded from a clamped address and zeroed by a select: no branch between the 32 loads.)
#pragma unroll
    for (int u = 0; u < 16; u++) {
        const int row = r16 + 16 * (u & 3), k = 4 * (kq + 16 * (u >> 2));
        const bool ka = k < kn;
        const int kc = kbeg + (ka ? k : 0);
        const float4 a = *reinterpret_cast<const float4 *>(A + (size_t)min(m0 + row, M - 1) * lda + kc);
        const float4 b = *reinterpret_cast<const float4 *>(B + (size_t)min(n0 + row, N - 1) * ldb + kc);
        const bool oa = ka && m0 + row < M, ob = ka && n0 + row < N;
        va[u] = make_float4(oa ? a.x : 0.f, oa ? a.y : 0.f, oa ? a.z : 0.f, oa ? a.w : 0.f);
        vb[u] = make_float4(ob ? b.x : 0.f, ob ? b.y : 0.f, ob ? b.z : 0.f, ob ? b.w : 0.f);
    }
#pragma unroll
    for (int u = 0; u < 16; u++) {
        const int row = r16 + 16 * (u & 3), k = 4 * (kq + 16 * (u >> 2));
        if (k < kn) {
            As[k][row] = va[u].x; As[k + 1][row] = va[u].y; As[k + 2][row] = va[u].z; As[k + 3][row] = va[u].w;
            Bs[k][row] = vb[u].x; Bs[k + 1][row] = vb[u].y; Bs[k + 2][row] = vb[u].z; Bs[k + 3][row] = vb[u].w;
        }
    }
    __syncthreads();
    float acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) acc[i][j] = 0.f;
    const int kr = (kn + 3) & ~3;                                 // (the slab is zero past kend up to the next multiple of 4)
#pragma unroll 8
    for (int k = 0; k < kr; k++) {
        const float4 a = *reinterpret_cast<const float4 *>(&As[k][4 * ty]);
        const float4 b = *reinterpret_cast<const float4 *>(&Bs[k][4 * tx]);
        const float av[4] = {a.x, a.y, a.z, a.w}, bv[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
            for (int j = 0; j < 4; j++) acc[i][j] = fmaf(av[i], bv[j], acc[i][j]);
    }
    // one slice: straight to the result (row pitch direct_ld, + bias); several: this slice's partial
    float *C = direct_ld ? part : part + (size_t)z * M * N;
    const int ldc = direct_ld ? direct_ld : N;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int m = m0 + 4 * ty + i;
        if (m >= M) continue;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int n = n0 + 4 * tx + j;
            if (n < N) C[(size_t)m * ldc + n] = acc[i][j] + ((direct_ld && bias) ? bias[n] : 0.f);
        }
    }
}

// out[m][n] = sum_z part[z][m][n] (+ bias[n]), slices in ascending order
__global__ __launch_bounds__(256) void k_slices_sum(const float *__restrict__ part, int slices, int M, int N, const float *__restrict__ bias,
                                                    float *__restrict__ out, int ldo) {
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x, tot = (size_t)M * N;
    if (idx >= tot) return;
    float a = 0.f;
    for (int z = 0; z < slices; z++) a += part[(size_t)z * tot + idx];
    const int m = (int)(idx / N), n = (int)(idx - (size_t)m * N);
    out[(size_t)m * ldo + n] = a + (bias ? bias[n] : 0.f);
}

}  // namespace

extern "C" long long spadot_sgemm_nt_slices_workspace(int M, int N, int slices) {
    if (M <= 0 || N <= 0 || slices < 1) return -1;
    return (long long)M * N * slices;
}

extern "C" int spadot_sgemm_nt_slices(const float *A, int lda, const float *B, int ldb, float *out, int ldo, const float *bias, int M, int N,
                                      int K, int slices, float *workspace, void *stream) {
    if (!A || !B || !out || M <= 0 || N <= 0 || K <= 0 || slices < 1 || slices > 256 || lda < K || ldb < K || ldo < N) return -22;
    int kslice = ((K + slices - 1) / slices + 3) / 4 * 4;           // slice starts on 16-byte boundaries
    const int used = (K + kslice - 1) / kslice;                       // (rounding can leave fewer slices than asked for)
    if (used > 1 && !workspace) return -22;
    const int vec_ok = (lda % 4 == 0 && ldb % 4 == 0 && ((uintptr_t)A % 16) == 0 && ((uintptr_t)B % 16) == 0) ? 1 : 0;
    const dim3 grid((unsigned)((N + TN - 1) / TN), (unsigned)((M + TM - 1) / TM), (unsigned)used);
    if (grid.y > 65535u) return -22;
    hipStream_t st = (hipStream_t)stream;
    if (kslice <= OS_KMAX && vec_ok && K % 4 == 0) {                  // a whole slice fits the one-shot tile
        constexpr int LDS_BYTES = 2 * OS_KMAX * PITCH * (int)sizeof(float);
        static PerDeviceFlag attr_set;
        if (!attr_set) {
            if (hipFuncSetAttribute((const void *)k_sgemm_nt_oneshot, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES) != hipSuccess) return -5;
            attr_set = true;
        }
        if (used == 1) {
            hipLaunchKernelGGL(k_sgemm_nt_oneshot, grid, dim3(256), LDS_BYTES, st, A, lda, B, ldb, out, M, N, K, kslice, vec_ok, bias, ldo);
            return hipGetLastError() == hipSuccess ? 0 : -5;
        }
        hipLaunchKernelGGL(k_sgemm_nt_oneshot, grid, dim3(256), LDS_BYTES, st, A, lda, B, ldb, workspace, M, N, K, kslice, vec_ok,
                           (const float *)nullptr, 0);
    } else {
        if (!workspace) return -22;
        hipLaunchKernelGGL(k_sgemm_nt_slices, grid, dim3(256), 0, st, A, lda, B, ldb, workspace, M, N, K, kslice, vec_ok);
    }
    const size_t tot = (size_t)M * N;
    hipLaunchKernelGGL(k_slices_sum, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, workspace, used, M, N, bias, out, ldo);
    return hipGetLastError() == hipSuccess ? 0 : -5;
}
