// enc_fused.hip -- the SVGP encoder's stages behind its first map as THREE launches instead of five (gfx950, wave64, fp32).
//
// Reference arithmetic: /root/reference/SpaDOT/model/encoder.py:7-34 in training mode --
//     h1 = x W1^T (+ b1)  ->  BatchNorm1d + LeakyReLU  ->  h2 = y1 W2^T (+ b2)  ->  BatchNorm1d + LeakyReLU  ->  z = y2 Wfc^T + bfc
// and /root/reference/SpaDOT/model/svgp.py:62-66 for what consumes z (mu | logvar -> 1 / var, K_nm / var).
//
// Why: the branch is the long pole of the step's forward pair (it ends ~60 us after the GAT branch) and every one of its
// short dependent launches waits 10-20 us for compute-unit slots beside the GAT branch's GEMMs (stage stamps, round 4/5:
// BatchNorm 34 us, hidden map 11, BatchNorm 46, SVGP_fc 16, k_svgp_pre2 23 -- for ~1 MB of data).  BatchNorm needs all rows
// of a column, the maps all columns of a row, so the data must cross workgroups twice; here it crosses through PARTIAL
// PRODUCTS in global memory instead of through launches of its own:
//   k_enc_bn_map   one workgroup per 16 columns of h1: batch statistics, normalise, LeakyReLU -> y1 (bit for bit what
//                  k_bn_act_fwd writes), and this column group's slice of the hidden map, part[g] = y1[:, 16 g ..] W2[:, 16 g ..]^T
//   k_enc_bn_fc    one workgroup per 4 columns of h2 = sum_g part[g] (summed in group order): batch statistics,
//                  normalise, LeakyReLU -> y2, and this column group's slice of SVGP_fc, pz[g] = y2[:, 4 g ..] Wfc[:, 4 g ..]^T
//   k_svgp_pre2p   k_svgp_pre2 (model_kernels.hip) reading z = bfc + sum_g pz[g] instead of z; it also stores z
//   k_enc_sum_z    z alone from the partials (callers that want the encoder's output without the SVGP stage behind it)
// Fixed summation orders, no atomics: bit-repeatable.  b <= 512 rows (the training batch).
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>

#include "../../include/spadot_model.h"

namespace {

constexpr int EC = 16, ERG = 16, ENT = EC * ERG, ERPT = 512 / ERG;      // 16 columns x 16 row lanes, 32 rows per thread
constexpr int YS = 20;                                                    // LDS row stride of the y1 tile (floats; 16-byte aligned rows)
constexpr int MAX_F2 = 128, MAX_Q = 32, MAX_NP = 32;

__device__ __forceinline__ float wsum(float x) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) x += __shfl_xor(x, off, 64);
    return x;
}

__device__ __forceinline__ float col_sum16(float v, float (*sh)[EC + 1], int cl, int rg) {
    __syncthreads();
    sh[rg][cl] = v;
    __syncthreads();
    float t = 0.f;
#pragma unroll 8
    for (int g = 0; g < ERG; g++) t += sh[g][cl];
    return t;
}

// ---- stage 1: BatchNorm + LeakyReLU over 16 columns of h1 (the arithmetic and summation order of k_bn_act_fwd's one-load
// path), then the column group's partial hidden map
__global__ __launch_bounds__(ENT) void k_enc_bn_map(const float *__restrict__ h1, const float *__restrict__ lb,
                                                    const float *__restrict__ gamma, const float *__restrict__ beta,
                                                    float *__restrict__ run_mean, float *__restrict__ run_var,
                                                    long long *__restrict__ nbt, int b, int F1, float momentum, float eps,
                                                    float slope, float *__restrict__ y1, float *__restrict__ save_mean,
                                                    float *__restrict__ save_invstd, const float *__restrict__ W2, int F2,
                                                    float *__restrict__ part) {
    __shared__ float sh[ERG][EC + 1];
    __shared__ __attribute__((aligned(16))) float ys[512 * YS];
    const int cl = threadIdx.x & (EC - 1), rg = threadIdx.x / EC;
    const int c0 = blockIdx.x * EC, c = c0 + cl;
    // All loads of the launch are issued HERE, unconditionally (rows past b re-read row b - 1 and are masked below): a
    // predicated load is a branch, and the compiler drains the load queue at its join -- the first form of this kernel
    // (`i < b ? h1[..] : 0`) was 32 dependent round trips, ~40 of its 45 us in the step.
    const float *__restrict__ lbp = lb ? lb : gamma;
    float v[ERPT];
#pragma unroll
    for (int r = 0; r < ERPT; r++) v[r] = h1[(size_t)min(rg + r * ERG, b - 1) * F1 + c];
    const float add = lb ? lbp[c] : 0.f, gam = gamma[c], bet = beta[c], rm0 = run_mean[c], rv0 = run_var[c];
#pragma unroll
    for (int r = 0; r < ERPT; r++) v[r] = rg + r * ERG < b ? v[r] + add : 0.f;
    float s = 0.f;
#pragma unroll
    for (int r = 0; r < ERPT; r++)
        if (rg + r * ERG < b) s += v[r];
    const float mean = col_sum16(s, sh, cl, rg) / (float)b;
    float ss = 0.f;
#pragma unroll
    for (int r = 0; r < ERPT; r++)
        if (rg + r * ERG < b) { const float d = v[r] - mean; ss += d * d; }
    const float var = col_sum16(ss, sh, cl, rg) / (float)b;
    const float invstd = rsqrtf(var + eps);
    {
        const float g = gam * invstd, o = bet;
#pragma unroll
        for (int r = 0; r < ERPT; r++) {
            const int i = rg + r * ERG;
            if (i < b) {
                float u = (v[r] - mean) * g + o;
                u = u > 0.f ? u : slope * u;
                y1[(size_t)i * F1 + c] = u;
                ys[i * YS + cl] = u;
            }
        }
        if (rg == 0) {
            save_mean[c] = mean; save_invstd[c] = invstd;
            run_mean[c] = (1.f - momentum) * rm0 + momentum * mean;
            run_var[c] = (1.f - momentum) * rv0 + momentum * var * ((float)b / (float)(b > 1 ? b - 1 : 1));
        }
    }
    if (nbt && blockIdx.x == 0 && threadIdx.x == 0) nbt[0] += 1;
    __syncthreads();
    // part[g][i][j] = sum_k y1[i][16 g + k] W2[j][16 g + k]  (k ascending): thread = output column j, every fourth row; the row's
    // 16 values are LDS broadcasts, the thread's 16 weights registers
    const int q = threadIdx.x >> 6;
    float *pg = part + (size_t)blockIdx.x * b * F2;
    for (int j = threadIdx.x & 63; j < F2; j += 64) {
        float w[EC];
#pragma unroll
        for (int k = 0; k < EC; k += 4) {
            const float4 t4 = *reinterpret_cast<const float4 *>(W2 + (size_t)j * F1 + c0 + k);
            w[k] = t4.x; w[k + 1] = t4.y; w[k + 2] = t4.z; w[k + 3] = t4.w;
        }
#pragma unroll 4
        for (int i = q; i < b; i += 4) {
            const float4 a0 = *reinterpret_cast<const float4 *>(ys + i * YS), a1 = *reinterpret_cast<const float4 *>(ys + i * YS + 4);
            const float4 a2 = *reinterpret_cast<const float4 *>(ys + i * YS + 8), a3 = *reinterpret_cast<const float4 *>(ys + i * YS + 12);
            float acc = a0.x * w[0];
            acc = fmaf(a0.y, w[1], acc); acc = fmaf(a0.z, w[2], acc); acc = fmaf(a0.w, w[3], acc);
            acc = fmaf(a1.x, w[4], acc); acc = fmaf(a1.y, w[5], acc); acc = fmaf(a1.z, w[6], acc); acc = fmaf(a1.w, w[7], acc);
            acc = fmaf(a2.x, w[8], acc); acc = fmaf(a2.y, w[9], acc); acc = fmaf(a2.z, w[10], acc); acc = fmaf(a2.w, w[11], acc);
            acc = fmaf(a3.x, w[12], acc); acc = fmaf(a3.y, w[13], acc); acc = fmaf(a3.z, w[14], acc); acc = fmaf(a3.w, w[15], acc);
            pg[(size_t)i * F2 + j] = acc;
        }
    }
}

// ---- stage 2: h2 = sum_g part[g] for 4 columns, BatchNorm + LeakyReLU -> y2, and the column group's partial SVGP_fc
__global__ __launch_bounds__(256) void k_enc_bn_fc(const float *__restrict__ part, int NP, const float *__restrict__ lb,
                                                   const float *__restrict__ gamma, const float *__restrict__ beta,
                                                   float *__restrict__ run_mean, float *__restrict__ run_var,
                                                   long long *__restrict__ nbt, int b, int F2, float momentum, float eps,
                                                   float slope, float *__restrict__ h2, float *__restrict__ y2,
                                                   float *__restrict__ save_mean, float *__restrict__ save_invstd,
                                                   const float *__restrict__ Wfc, int Q, float *__restrict__ pz) {
    __shared__ float red[4][4];
    __shared__ float wf[MAX_Q][4];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int c0 = blockIdx.x * 4;
    // every load of the launch up front and unconditional (see k_enc_bn_map): rows past b re-read row b - 1, masked below
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    float v[2][4];
    bool in[2];
    float4 acc2[2] = {zero4, zero4};
    const float *pr0 = part + (size_t)min(t, b - 1) * F2 + c0, *pr1 = part + (size_t)min(t + 256, b - 1) * F2 + c0;
    auto pass = [&](int g0) __attribute__((always_inline)) {      // eight groups' partials of both rows in flight together, added in group order
        float4 pa[8], pb[8];
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const size_t go = (size_t)min(g0 + k, NP - 1) * b * F2;
            pa[k] = *reinterpret_cast<const float4 *>(pr0 + go);
            pb[k] = *reinterpret_cast<const float4 *>(pr1 + go);
        }
#pragma unroll
        for (int k = 0; k < 8; k++)
            if (g0 + k < NP) {
                acc2[0].x += pa[k].x; acc2[0].y += pa[k].y; acc2[0].z += pa[k].z; acc2[0].w += pa[k].w;
                acc2[1].x += pb[k].x; acc2[1].y += pb[k].y; acc2[1].z += pb[k].z; acc2[1].w += pb[k].w;
            }
    };
    // the small operands ride with the first pass (a loop's first pass would wait for whatever was issued before it)
    const float wfv = Wfc[(size_t)(min(t, Q * 4 - 1) >> 2) * F2 + c0 + (t & 3)];
    const float4 add4r = *reinterpret_cast<const float4 *>((lb ? lb : gamma) + c0);
    const float4 add4 = lb ? add4r : zero4;
    const float4 gam4 = *reinterpret_cast<const float4 *>(gamma + c0), bet4 = *reinterpret_cast<const float4 *>(beta + c0);
    const float rm0 = run_mean[c0 + (t & 3)], rv0 = run_var[c0 + (t & 3)];
    pass(0);
    for (int g0 = 8; g0 < NP; g0 += 8) pass(g0);
    if (t < Q * 4) wf[t >> 2][t & 3] = wfv;
#pragma unroll
    for (int r = 0; r < 2; r++) {
        const int i = t + 256 * r;
        in[r] = i < b;
        if (in[r]) *reinterpret_cast<float4 *>(h2 + (size_t)i * F2 + c0) = acc2[r];   // (without the map's bias: BatchNorm's lb carries it)
        v[r][0] = acc2[r].x; v[r][1] = acc2[r].y; v[r][2] = acc2[r].z; v[r][3] = acc2[r].w;
    }
    float mean[4], invstd[4], var[4];
    const float add[4] = {add4.x, add4.y, add4.z, add4.w};
#pragma unroll
    for (int r = 0; r < 2; r++)
#pragma unroll
        for (int k = 0; k < 4; k++) v[r][k] = in[r] ? v[r][k] + add[k] : 0.f;
    // column sums over the b rows: the thread's two rows, the wave, then the four waves in order
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const float s = wsum(v[0][k] + v[1][k]);
        if (lane == 0) red[wave][k] = s;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; k++) mean[k] = (((red[0][k] + red[1][k]) + red[2][k]) + red[3][k]) / (float)b;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const float d0 = in[0] ? v[0][k] - mean[k] : 0.f, d1 = in[1] ? v[1][k] - mean[k] : 0.f;
        const float s = wsum(d0 * d0 + d1 * d1);
        if (lane == 0) red[wave][k] = s;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; k++) {
        var[k] = (((red[0][k] + red[1][k]) + red[2][k]) + red[3][k]) / (float)b;
        invstd[k] = rsqrtf(var[k] + eps);
    }
    if (t < 4) {
        const int c = c0 + t;
        save_mean[c] = mean[t]; save_invstd[c] = invstd[t];
        run_mean[c] = (1.f - momentum) * rm0 + momentum * mean[t];
        run_var[c] = (1.f - momentum) * rv0 + momentum * var[t] * ((float)b / (float)(b > 1 ? b - 1 : 1));
    }
    if (nbt && blockIdx.x == 0 && t == 0) nbt[0] += 1;
    float g4[4], o4[4];
#pragma unroll
    for (int k = 0; k < 4; k++) { g4[k] = (&gam4.x)[k] * invstd[k]; o4[k] = (&bet4.x)[k]; }
    float *pg = pz + (size_t)blockIdx.x * b * Q;
#pragma unroll
    for (int r = 0; r < 2; r++) {
        const int i = t + 256 * r;
        if (!in[r]) continue;
        float u[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const float x = (v[r][k] - mean[k]) * g4[k] + o4[k];
            u[k] = x > 0.f ? x : slope * x;
        }
        *reinterpret_cast<float4 *>(y2 + (size_t)i * F2 + c0) = make_float4(u[0], u[1], u[2], u[3]);
        for (int q = 0; q < Q; q++)
            pg[(size_t)i * Q + q] = fmaf(u[3], wf[q][3], fmaf(u[2], wf[q][2], fmaf(u[1], wf[q][1], u[0] * wf[q][0])));
    }
}

// z[i][q] = bias[q] + sum_g pz[g][i][q]  (g ascending)
__global__ __launch_bounds__(256) void k_enc_sum_z(const float *__restrict__ pz, int NP, const float *__restrict__ bias, int b, int Q,
                                                   float *__restrict__ z) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= b * Q) return;
    float acc = 0.f;
    for (int g = 0; g < NP; g++) acc += pz[(size_t)g * b * Q + e];
    z[e] = acc + bias[e % Q];
}

// k_svgp_pre2 (model_kernels.hip) with z taken from the partials: one wave per (row i, latent dimension l); lanes 0 .. NP - 1
// fetch the partials of z[i][l] and z[i][L + l], lane 0 adds them in group order
__global__ __launch_bounds__(256) void k_svgp_pre2p(const float *__restrict__ pz, int NP, const float *__restrict__ bias,
                                                    const double *__restrict__ Kn, int b, int L, int m, float *__restrict__ z,
                                                    double *__restrict__ mu, double *__restrict__ var, double *__restrict__ w,
                                                    double *__restrict__ muw, double *__restrict__ A) {
    const int e = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (e >= b * L) return;
    const int i = e / L, l = e - i * L, Q = 2 * L;
    float pa = 0.f, pb = 0.f;
    if (lane < NP) {
        pa = pz[((size_t)lane * b + i) * Q + l];
        pb = pz[((size_t)lane * b + i) * Q + L + l];
    }
    float za = 0.f, zb = 0.f;
    for (int g = 0; g < NP; g++) { za += __shfl(pa, g, 64); zb += __shfl(pb, g, 64); }      // group order, every lane the same value
    za += bias[l]; zb += bias[L + l];
    const double m_ = (double)za;
    const double v = (double)expf(zb);                                  // torch.exp in fp32, like the encoder's own
    const double wi = 1.0 / v;
    if (lane == 0) {
        z[(size_t)i * Q + l] = za; z[(size_t)i * Q + L + l] = zb;
        mu[e] = m_; var[e] = v; w[e] = wi; muw[e] = m_ / v;
    }
    const double *kr = Kn + (size_t)i * m;
    double *ar = A + ((size_t)l * b + i) * m;
    for (int k = lane; k < m; k += 64) ar[k] = kr[k] * wi;
}

}  // namespace

extern "C" {

int spadot_enc_fused_supported(int b, int F1, int F2, int Q) {
    return b > 0 && b <= 512 && F1 > 0 && F1 % EC == 0 && F1 / EC <= MAX_NP && F2 > 0 && F2 % 4 == 0 && F2 <= MAX_F2 && F2 / 4 <= MAX_NP &&
           Q > 0 && Q <= MAX_Q && Q * 4 <= 256;
}

long long spadot_enc_fused_workspace(int b, int F1, int F2, int Q) {      // floats: part [F1/16][b][F2] then pz [F2/4][b][Q]
    if (!spadot_enc_fused_supported(b, F1, F2, Q)) return -22;
    return (long long)(F1 / EC) * b * F2 + (long long)(F2 / 4) * b * Q;
}

int spadot_enc_bn_map(const float *h1, const float *lin_bias, const float *gamma, const float *beta, float *running_mean,
                      float *running_var, long long *num_batches_tracked, int b, int F1, double momentum, double eps, double slope,
                      float *y1, float *save_mean, float *save_invstd, const float *W2, int F2, float *part, void *stream) {
    if (!spadot_enc_fused_supported(b, F1, F2, 4) || !h1 || !gamma || !beta || !running_mean || !running_var || !y1 || !save_mean ||
        !save_invstd || !W2 || !part || ((uintptr_t)W2 & 15))
        return -22;
    hipLaunchKernelGGL(k_enc_bn_map, dim3(F1 / EC), dim3(ENT), 0, (hipStream_t)stream, h1, lin_bias, gamma, beta, running_mean,
                       running_var, num_batches_tracked, b, F1, (float)momentum, (float)eps, (float)slope, y1, save_mean, save_invstd,
                       W2, F2, part);
    return hipGetLastError() == hipSuccess ? 0 : -5;
}

int spadot_enc_bn_fc(const float *part, int nparts, const float *lin_bias, const float *gamma, const float *beta, float *running_mean,
                     float *running_var, long long *num_batches_tracked, int b, int F2, double momentum, double eps, double slope,
                     float *h2, float *y2, float *save_mean, float *save_invstd, const float *Wfc, int Q, float *pz, void *stream) {
    if (b <= 0 || b > 512 || F2 <= 0 || F2 % 4 || F2 > MAX_F2 || nparts < 1 || nparts > MAX_NP || Q <= 0 || Q > MAX_Q || Q * 4 > 256 ||
        !part || !gamma || !beta || !running_mean || !running_var || !h2 || !y2 || !save_mean || !save_invstd || !Wfc || !pz ||
        ((uintptr_t)part & 15) || ((uintptr_t)h2 & 15) || ((uintptr_t)y2 & 15))
        return -22;
    hipLaunchKernelGGL(k_enc_bn_fc, dim3(F2 / 4), dim3(256), 0, (hipStream_t)stream, part, nparts, lin_bias, gamma, beta, running_mean,
                       running_var, num_batches_tracked, b, F2, (float)momentum, (float)eps, (float)slope, h2, y2, save_mean,
                       save_invstd, Wfc, Q, pz);
    return hipGetLastError() == hipSuccess ? 0 : -5;
}

int spadot_enc_sum_z(const float *pz, int nparts, const float *bias, int b, int Q, float *z, void *stream) {
    if (!pz || !bias || !z || nparts < 1 || b <= 0 || Q <= 0) return -22;
    hipLaunchKernelGGL(k_enc_sum_z, dim3((b * Q + 255) / 256), dim3(256), 0, (hipStream_t)stream, pz, nparts, bias, b, Q, z);
    return hipGetLastError() == hipSuccess ? 0 : -5;
}

int spadot_svgp_pre2_partials(const float *pz, int nparts, const float *bias, const double *Kn, int b, int L, int m, float *z,
                              double *mu, double *var, double *w, double *muw, double *A, void *stream) {
    if (b <= 0 || L <= 0 || m <= 0 || nparts < 1 || nparts > 64 || !pz || !bias || !Kn || !A || !z) return -22;
    hipLaunchKernelGGL(k_svgp_pre2p, dim3((b * L + 3) / 4), dim3(256), 0, (hipStream_t)stream, pz, nparts, bias, Kn, b, L, m, z, mu,
                       var, w, muw, A);
    return hipGetLastError() == hipSuccess ? 0 : -5;
}

}  // extern "C"
