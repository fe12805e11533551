// recon_fb.hip -- the decoder's output map, the reconstruction term and their backward as ONE launch on the matrix cores
// (gfx950, wave64, bf16 operands, fp32 accumulation).
//
// Reference arithmetic: /root/reference/SpaDOT/model/decoder.py:3-20 (the last Linear, hidden -> G) and
// /root/reference/SpaDOT/model/SpaDOT.py:89 (recon = sum (y - decoder(z))^2 / G) with their backward for a KNOWN seed:
// d elbo / d recon is the loss weight lambda1 (/root/reference/SpaDOT/utils/_train_utils.py:205-212), a device scalar that
// exists when the forward pass runs -- the staged step's tail differentiates right behind its own forward, so forward and
// backward of this stage are one kernel (the idea of ops.cluster_losses_fb).
//
// What it replaces on the loss tail's dependency chain (one stream of dependent launches, ~20-30 us each beside the side
// stream's fp64 GEMMs): library GEMM o = h W^T [b x G] (23 us), k_bias_sqerr_bwd (30 us), library GEMM dh = g W (29 us) --
// and the 6 MB fp32 image of o, which is never formed now.
//   workgroup = 128 rows x 128 genes:  o = h W_blk^T   -> d = y - (o + bias), sum d^2 (fp64 partial), g = coef d (bf16, stored:
//   the weight gradient g^T h is a library GEMM that only the optimizer waits for), column sums of g (bias gradient partial),
//   dh_partial [128 x K] = g W_blk (second product, contraction over the block's 128 genes).
// The partial dh of the G / 128 gene blocks are summed in block order by k_recon_fb_reduce (fixed order: bit-repeatable).
// MFMA operands (v_mfma_f32_32x32x16_bf16: A lane -> row l & 31, B lane -> column l & 31, both with k = 8 (l >> 5) .. + 7;
// D register e of lane l -> row (e & 3) + 8 (e >> 2) + 4 (l >> 5), column l & 31):
//   product 1:  D1 [gene x row]  A = W rows (16 bytes of a W row straight from global / L2), B = h rows (likewise)
//   product 2:  D2 [chan x row]  A = rows of the TRANSPOSED weight image WT [K x Gp] (16 bytes straight from global / L2 as
//               well), B = g rows from the LDS image the epilogue of product 1 wrote
// No weight tile is staged in LDS: the first version did (73 KB for the block's W rows, transposed reads for product 2) and
// took 89 us inside the step against ~20 alone -- a workgroup that needs most of a compute unit's LDS waits for a whole
// unit to drain beside the side stream's fp64 GEMMs (the lesson of DESIGN section 4, once more).  The caller keeps WT current
// (a 1.5 MB transposed copy per step, made where the main stream has slack).
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>

#include "../../include/spadot_model.h"

namespace {

constexpr int K = 256;                       // hidden width (decoder_layers[-1]); the only instantiation the model uses
constexpr int BR = 64, BG = 128;             // rows x genes per workgroup
constexpr int NT = 256;
constexpr int DROW = 2 * BG + 16;            // g image row stride (bytes): = 16 mod 256 (plain 16-byte reads conflict-free)
constexpr int LDS_D = BR * DROW;             // 17 408

typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));

__device__ __forceinline__ unsigned short bf16_bits(float x) { return __builtin_bit_cast(unsigned short, (__bf16)x); }
__device__ __forceinline__ unsigned pack2(float lo, float hi) { return (unsigned)bf16_bits(lo) | ((unsigned)bf16_bits(hi) << 16); }

// One workgroup = 64 rows x 128 genes, four waves: wave (rt, gh) takes row tile rt (32 rows) and, in product 1, gene half gh
// (two 32-gene tiles), in product 2 channel half gh (four 32-channel tiles).  Inside the step a dependent memory round trip
// costs 3-5 us (the side stream's fp64 GEMMs keep the memory system busy), so EVERY global operand of the workgroup -- both
// products' weight fragments, the h fragments, y and the bias -- is requested at the top, before the first MFMA: one round
// trip for the whole kernel (the first version walked ~12 of them: 90 us in the step against ~20 alone).  One workgroup per
// compute unit: the fragments live in ~400 registers.
__global__ __launch_bounds__(NT, 1) void k_recon_fb(const __bf16 *__restrict__ hb, const __bf16 *__restrict__ W,
                                                    const __bf16 *__restrict__ WT, int ldt,
                                                    const float *__restrict__ bias, const float *__restrict__ y, int b, int G,
                                                    double inv_scale, const float *__restrict__ gw, __bf16 *__restrict__ gc,
                                                    float *__restrict__ dbp, double *__restrict__ lossp, float *__restrict__ dxp) {
    __shared__ __attribute__((aligned(16))) unsigned char dl[LDS_D];
    __shared__ float dbl[2 * BG];
    __shared__ double lsh[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int rt = wave & 1, gh = wave >> 1;
    const int gb = blockIdx.x, rb = blockIdx.y;
    const int g0 = gb * BG, r0 = rb * BR;
    const int hh = lane >> 5, l31 = lane & 31;
    const int r = r0 + 32 * rt + l31;
    const int rc = min(r, b - 1);
    const bool row_on = r < b;

    // ---- all requests first
    bf8 hf[K / 16], wf[2][K / 16], w2[4][BG / 16];
    float4 yv[2][4], bv[2][4];
    {
        const __bf16 *hrow = hb + (size_t)rc * K + 8 * hh;
#pragma unroll
        for (int ks = 0; ks < K / 16; ks++) hf[ks] = *reinterpret_cast<const bf8 *>(hrow + 16 * ks);
#pragma unroll
        for (int t = 0; t < 2; t++) {
            const __bf16 *wrow = W + (size_t)min(g0 + 64 * gh + 32 * t + l31, G - 1) * K + 8 * hh;
#pragma unroll
            for (int ks = 0; ks < K / 16; ks++) wf[t][ks] = *reinterpret_cast<const bf8 *>(wrow + 16 * ks);
        }
#pragma unroll
        for (int t = 0; t < 2; t++)
#pragma unroll
            for (int q = 0; q < 4; q++) {
                // (G % 4 == 0: four genes are inside together.  UNCONDITIONAL loads from a clamped address -- a predicated load
                // is a branch, and the compiler waits for every outstanding load at its join: the first build of this kernel
                // had ten such waits in a row; what lies outside is masked in the epilogue)
                const int g = min(g0 + 64 * gh + 32 * t + 8 * q + 4 * hh, G - 4);
                yv[t][q] = *reinterpret_cast<const float4 *>(y + (size_t)rc * G + g);
                bv[t][q] = *reinterpret_cast<const float4 *>(bias + g);
            }
        const __bf16 *wt0 = WT + (size_t)(128 * gh + l31) * ldt + g0 + 8 * hh;
#pragma unroll
        for (int ct = 0; ct < 4; ct++)
#pragma unroll
            for (int gs = 0; gs < BG / 16; gs++) w2[ct][gs] = *reinterpret_cast<const bf8 *>(wt0 + (size_t)(32 * ct) * ldt + 16 * gs);
    }
    const double coef = -2.0 * inv_scale * (double)gw[0];

    // ---- product 1: D1 [gene tile (32 genes) x this wave's 32 rows], contraction over the K hidden channels
    f16v acc1[2];
#pragma unroll
    for (int t = 0; t < 2; t++)
#pragma unroll
        for (int i = 0; i < 16; i++) acc1[t][i] = 0.f;
#pragma unroll
    for (int ks = 0; ks < K / 16; ks++)
#pragma unroll
        for (int t = 0; t < 2; t++) acc1[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[t][ks], hf[ks], acc1[t], 0, 0, 0);

    // ---- epilogue 1: d = y - (o + bias); loss partial; g = coef d -> bf16 image in LDS; column sums of g over this wave's rows
    double loss = 0.0;
#pragma unroll
    for (int t = 0; t < 2; t++) {
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int gl = 64 * gh + 32 * t + 8 * q + 4 * hh;             // four consecutive genes of this lane's row
            const bool on = row_on && g0 + gl < G;
            const float ye[4] = {yv[t][q].x, yv[t][q].y, yv[t][q].z, yv[t][q].w};
            const float be[4] = {bv[t][q].x, bv[t][q].y, bv[t][q].z, bv[t][q].w};
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const double d = on ? (double)ye[e] - (double)(acc1[t][4 * q + e] + be[e]) : 0.0;
                loss += d * d;
                v[e] = (float)(coef * d);
            }
            *reinterpret_cast<uint2 *>(dl + (32 * rt + l31) * DROW + gl * 2) = make_uint2(pack2(v[0], v[1]), pack2(v[2], v[3]));
            // column sums over the wave's 32 rows (lanes with equal hh): butterfly over l31
#pragma unroll
            for (int e = 0; e < 4; e++) {
                float s = v[e];
#pragma unroll
                for (int off = 16; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
                if (l31 == 0) dbl[rt * BG + gl + e] = s;
            }
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) loss += __shfl_xor(loss, off, 64);
    if (lane == 0) lsh[wave] = loss;
    __syncthreads();
    if (tid == 0) lossp[rb * gridDim.x + gb] = ((lsh[0] + lsh[1]) + lsh[2]) + lsh[3];
    if (tid < BG && g0 + tid < G) dbp[(size_t)rb * G + g0 + tid] = dbl[tid] + dbl[BG + tid];
    // g rows of this block to global memory (bf16, whole 16-byte pieces: G % 8 == 0), for the weight gradient g^T h
    {
        const int row = tid >> 2, quarter = tid & 3;
        if (r0 + row < b) {
#pragma unroll
            for (int p = 0; p < 4; p++) {
                const int gl = quarter * 32 + p * 8;
                if (g0 + gl < G)
                    *reinterpret_cast<uint4 *>(gc + (size_t)(r0 + row) * G + g0 + gl) = *reinterpret_cast<const uint4 *>(dl + row * DROW + gl * 2);
            }
        }
    }

    // ---- product 2: D2 [channel tile ct (32 channels) x this wave's 32 rows] = W_blk^T g^T, contraction over the block's 128 genes
    // (genes past G: the g image holds zeros there and WT's rows are padded to a multiple of 128, so nothing is read out of bounds)
    f16v acc2[4];
#pragma unroll
    for (int ct = 0; ct < 4; ct++)
#pragma unroll
        for (int i = 0; i < 16; i++) acc2[ct][i] = 0.f;
    const unsigned char *drow = dl + (32 * rt + l31) * DROW + 16 * hh;
#pragma unroll
    for (int gs = 0; gs < BG / 16; gs++) {
        const bf8 gf = *reinterpret_cast<const bf8 *>(drow + gs * 32);
#pragma unroll
        for (int ct = 0; ct < 4; ct++) acc2[ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w2[ct][gs], gf, acc2[ct], 0, 0, 0);
    }
    if (row_on) {
        float *dst = dxp + ((size_t)gb * b + r) * K + 128 * gh;
#pragma unroll
        for (int ct = 0; ct < 4; ct++)
#pragma unroll
            for (int q = 0; q < 4; q++)
                *reinterpret_cast<float4 *>(dst + 32 * ct + 8 * q + 4 * hh) =
                    make_float4(acc2[ct][4 * q], acc2[ct][4 * q + 1], acc2[ct][4 * q + 2], acc2[ct][4 * q + 3]);
    }
}

// dh[e] = sum over the gene blocks of dxp[blk][e] (block order), four elements per thread
__global__ __launch_bounds__(256) void k_recon_fb_reduce(const float *__restrict__ dxp, int nblk, long long n4, float *__restrict__ dh) {
    const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
    if (e >= n4) return;
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const f32x4 *src = reinterpret_cast<const f32x4 *>(dxp) + e;
    for (int k = 0; k < nblk; k += 8) {          // eight blocks' pieces in flight (unconditional loads, clamped block), added in order
        f32x4 v[8];
#pragma unroll
        for (int u = 0; u < 8; u++) v[u] = src[(long long)min(k + u, nblk - 1) * n4];
#pragma unroll
        for (int u = 0; u < 8; u++)
            if (k + u < nblk) acc += v[u];
    }
    reinterpret_cast<f32x4 *>(dh)[e] = acc;
}

}  // namespace

extern "C" {

int spadot_recon_fb_supported(int b, int Kin, int G) {
    return b > 0 && b <= 4096 && Kin == K && G >= BG && G % 8 == 0;
}

// floats of caller-owned workspace: dxp [ceil(G / 128)][b][K], then dbp [ceil(b / 64)][G]; doubles: lossp [ceil(b / 64) ceil(G / 128)]
long long spadot_recon_fb_workspace(int b, int Kin, int G) {
    if (!spadot_recon_fb_supported(b, Kin, G)) return -22;
    return (long long)((G + BG - 1) / BG) * b * K + (long long)((b + BR - 1) / BR) * G;
}

int spadot_recon_fb(const void *h_bf16, const void *W_bf16, const void *WT_bf16, int ldt, const float *bias, const float *y, int b, int Kin,
                    int G, double inv_scale, const float *grad_weight, void *g_bf16, float *workspace, double *loss_parts, float *dh,
                    void *stream) {
    if (!spadot_recon_fb_supported(b, Kin, G) || !h_bf16 || !W_bf16 || !WT_bf16 || !bias || !y || !grad_weight || !g_bf16 || !workspace ||
        !loss_parts || !dh || ldt < (G + BG - 1) / BG * BG || ldt % 8 || ((uintptr_t)WT_bf16 & 15))
        return -22;
    if (((uintptr_t)h_bf16 & 15) || ((uintptr_t)W_bf16 & 15) || ((uintptr_t)bias & 15) || ((uintptr_t)y & 15) || ((uintptr_t)g_bf16 & 15) ||
        ((uintptr_t)workspace & 15) || ((uintptr_t)dh & 15))
        return -22;
    const int GB = (G + BG - 1) / BG, RB = (b + BR - 1) / BR;
    float *dxp = workspace, *dbp = workspace + (size_t)GB * b * K;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(k_recon_fb, dim3(GB, RB), dim3(NT), 0, st, (const __bf16 *)h_bf16, (const __bf16 *)W_bf16, (const __bf16 *)WT_bf16, ldt,
                       bias, y, b, G,
                       inv_scale, grad_weight, (__bf16 *)g_bf16, dbp, loss_parts, dxp);
    const long long n4 = (long long)b * K / 4;
    hipLaunchKernelGGL(k_recon_fb_reduce, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, st, (const float *)dxp, GB, n4, dh);
    return hipGetLastError() == hipSuccess ? 0 : -5;
}

}  // extern "C"

namespace {
__global__ __launch_bounds__(256) void k_sum_parts_d(const double *__restrict__ part, int n, double scale, float *__restrict__ out) {
    __shared__ double sh[4];
    double acc = 0.0;
    for (int k = threadIdx.x; k < n; k += 256) acc += part[k];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) out[0] = (float)((((sh[0] + sh[1]) + sh[2]) + sh[3]) * scale);
}
}  // namespace

// out[0] = scale * sum of n doubles (fixed order): the VALUE of the reconstruction term from spadot_recon_fb's loss partials
extern "C" int spadot_sum_parts(const double *part, int n, double scale, float *out, void *stream) {
    if (!part || !out || n < 1) return -22;
    hipLaunchKernelGGL(k_sum_parts_d, dim3(1), dim3(256), 0, (hipStream_t)stream, part, n, scale, out);
    return hipGetLastError() == hipSuccess ? 0 : -5;
}
