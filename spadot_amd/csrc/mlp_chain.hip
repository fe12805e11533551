// mlp_chain.hip -- the hidden stages of the decoder as ONE launch forward and ONE (+ a column sum) backward (gfx950).
//
// /root/reference/SpaDOT/model/decoder.py:3-20: decoder_net = [Linear, LayerNorm, LeakyReLU] x len(decoder_layers), then the
// output Linear.  With decoder_layers = [64, 256] and z_dim = 20 the hidden part is 1024 x 20 -> 64 -> 256: a few MFLOP, but
// as separate launches (GEMM, LN+act per stage forward; LN rows, LN columns, bias column sum, two GEMMs per stage backward)
// it was 4 + 10 dependent launches of ~5 us each on the one stream the loss tail of a step runs on.  Every quantity of a
// stage is row-local except the parameter gradients, so a workgroup carries its rows through all stages:
//   forward : 4 rows per workgroup; per stage  a = x W^T + b (thread per output column, the 4 rows share each weight load),
//             LayerNorm by one wave per row (two-pass mean / variance like k_ln_act_fwd), LeakyReLU; a, y, mean, invstd kept;
//   backward: 8 rows per workgroup, stages last to first:  LN + activation backward per row (32 lanes per row), the stage's
//             weight-gradient partial da^T x over the 8 rows in 8 x 4 register tiles, its column partials (bias, gamma,
//             beta), then dx = da W into LDS as the next stage's incoming gradient.  Partials go to a workspace
//             [workgroup][all parameters of all stages]; ONE k_colsum_parts launch (spadot_colsum) adds them in workgroup
//             order into a buffer laid out like one workspace row -- fixed order, no atomics.
// Limits (the caller falls back to the per-stage launches otherwise): <= 4 stages, widths <= 256 and multiples of 4, output
// widths multiples of 8.
#include <hip/hip_runtime.h>
#include <cstdint>

#include "../../include/spadot_model.h"
#include "per_device.h"

namespace {

constexpr int MAXL = 4, MAXD = 256, WAVE = 64;
constexpr int RBF = 4;                  // rows per workgroup, forward
constexpr int RBB = 8;                  // rows per workgroup, backward
constexpr int LDW = MAXD + 4;           // LDS row stride (floats): 16-byte aligned rows, off the 256-byte bank period

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct Stage {
    const float *W, *bias, *gamma, *beta;
    float *a, *y, *mean, *invstd;       // saved by forward, read by backward
    int din, dout;
    float eps, slope;
    int ws_off;                         // backward: offset of this stage's partials inside a workspace row
};
struct Chain {
    Stage s[MAXL];
    int n;
    __bf16 *ybf;                        // forward: optional bf16 copy of the last stage's output (the output map's GEMM operand)
};

__device__ __forceinline__ float wave_sum(float x) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) x += __shfl_xor(x, off, WAVE);
    return x;
}

__global__ __launch_bounds__(256) void k_mlp_chain_fwd(const float *__restrict__ x, int b, Chain ch) {
    __shared__ __attribute__((aligned(16))) float act[2][RBF][LDW];
    __shared__ float par[3][MAXD];                       // bias, gamma, beta of the current stage
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int r0 = blockIdx.x * RBF;
    int cur = 0;
    {
        const int d0 = ch.s[0].din;
        for (int e = t; e < RBF * d0; e += 256) {
            const int r = e / d0, k = e - r * d0;
            act[0][r][k] = (r0 + r < b) ? x[(size_t)(r0 + r) * d0 + k] : 0.f;
        }
    }
    __syncthreads();
    for (int l = 0; l < ch.n; l++) {
        const Stage &s = ch.s[l];
        const int din = s.din, dout = s.dout;
        // a[r][j] = bias[j] + <W[j, :], x[r, :]>: one thread per output column, the rows share each 16-byte weight load
        if (t < dout) { par[0][t] = s.bias[t]; par[1][t] = s.gamma[t]; par[2][t] = s.beta[t]; }      // (dout <= 256 = threads)
        for (int j = t; j < dout; j += 256) {
            float acc[RBF];
            const float bj = s.bias[j];
#pragma unroll
            for (int r = 0; r < RBF; r++) acc[r] = bj;
            const float4 *w4 = reinterpret_cast<const float4 *>(s.W + (size_t)j * din);
            for (int kb = 0; kb < din / 4; kb += 16) {                 // 16 weight loads in flight, then their products
                float4 w[16];
#pragma unroll
                for (int u = 0; u < 16; u++) w[u] = kb + u < din / 4 ? w4[kb + u] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int u = 0; u < 16; u++) {
                    if (kb + u < din / 4) {
#pragma unroll
                        for (int r = 0; r < RBF; r++) {
                            const float4 xv = *reinterpret_cast<const float4 *>(&act[cur][r][4 * (kb + u)]);
                            acc[r] += w[u].x * xv.x + w[u].y * xv.y + w[u].z * xv.z + w[u].w * xv.w;
                        }
                    }
                }
            }
#pragma unroll
            for (int r = 0; r < RBF; r++) act[cur ^ 1][r][j] = acc[r];
        }
        __syncthreads();
        // LayerNorm + LeakyReLU: wave r owns row r
        {
            const int r = wave, row = r0 + r;
            float *ar = act[cur ^ 1][r];
            float sm = 0.f;
            for (int j = lane; j < dout; j += WAVE) sm += ar[j];
            const float mean = wave_sum(sm) / (float)dout;
            float ss = 0.f;
            for (int j = lane; j < dout; j += WAVE) { const float d = ar[j] - mean; ss += d * d; }
            const float invstd = rsqrtf(wave_sum(ss) / (float)dout + s.eps);
            for (int j = lane; j < dout; j += WAVE) {
                const float av = ar[j];
                const float v = (av - mean) * invstd * par[1][j] + par[2][j];
                const float yv = v > 0.f ? v : s.slope * v;
                ar[j] = yv;
                if (row < b) {
                    s.a[(size_t)row * dout + j] = av; s.y[(size_t)row * dout + j] = yv;
                    if (ch.ybf && l == ch.n - 1) ch.ybf[(size_t)row * dout + j] = (__bf16)yv;
                }
            }
            if (lane == 0 && row < b) { s.mean[row] = mean; s.invstd[row] = invstd; }
        }
        __syncthreads();
        cur ^= 1;
    }
}

// Backward.  LDS: gz = incoming gradient of the current stage, turned into dz = dy * act'(y) in place; xh = normalised linear
// output; da = gradient at the stage's linear output; xs = the stage's input rows; wst = W (whole, or in chunks of whole rows).
// Nothing in a serial loop reads global memory: tiles are loaded cooperatively (independent, coalesced loads), the loops run
// out of LDS.  8 rows per workgroup: the fp32 FMA work of a stage (da^T x and da W) is spread over b / 8 compute units.
constexpr int WST = 16384;                                  // floats of W staged at a time (64 KiB)
constexpr int LPR = 256 / RBB;                              // lanes per row in the per-row phase
constexpr int LDS_BWD_FLOATS = 4 * RBB * LDW + 2 * RBB + MAXD + WST;
__global__ __launch_bounds__(256, 1) void k_mlp_chain_bwd(const float *__restrict__ dy_last, const float *__restrict__ x, int b,
                                                       Chain ch, float *__restrict__ dx_out, float *__restrict__ ws,
                                                       int ws_width, const float *__restrict__ dx_add) {
    extern __shared__ __attribute__((aligned(16))) float lds_dyn[];
    float (*gz)[LDW] = reinterpret_cast<float (*)[LDW]>(lds_dyn);
    float (*da)[LDW] = reinterpret_cast<float (*)[LDW]>(lds_dyn + RBB * LDW);
    float (*xs)[LDW] = reinterpret_cast<float (*)[LDW]>(lds_dyn + 2 * RBB * LDW);
    float (*xh)[LDW] = reinterpret_cast<float (*)[LDW]>(lds_dyn + 3 * RBB * LDW);
    float *rmean = lds_dyn + 4 * RBB * LDW, *rinv = rmean + RBB;
    float *gam = rinv + RBB;                                    // gamma of the current stage (MAXD floats)
    float *wst = gam + MAXD;
    const int t = threadIdx.x;
    const int r0 = blockIdx.x * RBB;
    float *wrow = ws + (size_t)blockIdx.x * ws_width;
    {   // incoming gradient of the last stage
        const int d = ch.s[ch.n - 1].dout;
        // All loads first, UNCONDITIONAL from a clamped position and masked afterwards.  (Written as `cond ? load : 0` -- the form
        // this kernel had until round 5 -- every load is a branch and hipcc drains the load queue at its join: ~50 dependent
        // round trips per stage where the comment promised one.)
        float v[RBB * MAXD / 256];
#pragma unroll
        for (int u = 0; u < RBB * MAXD / 256; u++) {
            const int e = min(t + u * 256, RBB * d - 1), r = e / d, j = e - r * d;
            v[u] = dy_last[(size_t)min(r0 + r, b - 1) * d + j];
        }
#pragma unroll
        for (int u = 0; u < RBB * MAXD / 256; u++) {
            const int e = t + u * 256;
            if (!(e < RBB * d && r0 + e / d < b)) v[u] = 0.f;
        }
#pragma unroll
        for (int u = 0; u < RBB * MAXD / 256; u++) {
            const int e = t + u * 256, r = e / d, j = e - r * d;
            if (e < RBB * d) gz[r][j] = v[u];
        }
    }
    for (int l = ch.n - 1; l >= 0; l--) {
        const Stage &s = ch.s[l];
        const int din = s.din, dout = s.dout;
        const float *xin = l == 0 ? x : ch.s[l - 1].y;
        const int tk = din / 4;
        const int jc = min(dout, WST / din);                        // W rows per staged chunk (all of them when W fits)
        const bool need_dx = l > 0 || dx_out != nullptr;
        {
            // every global load of the stage is issued before the first LDS store (fixed trip counts, predicated): one
            // round trip instead of one per unrolled group
            constexpr int NE = RBB * MAXD / 256, NW = WST / 4 / 256;
            float vx[NE], vy[NE], va[NE];
            f32x4 vw[NW];                              // (a native vector: arrays of HIP's float4 class stay in scratch memory)
#pragma unroll
            for (int u = 0; u < NE; u++) {
                const int e = min(t + u * 256, RBB * din - 1), r = e / din, k = e - r * din;
                vx[u] = xin[(size_t)min(r0 + r, b - 1) * din + k];
                const int e2 = min(t + u * 256, RBB * dout - 1), r2 = e2 / dout, j = e2 - r2 * dout;
                const size_t o2 = (size_t)min(r0 + r2, b - 1) * dout + j;
                vy[u] = s.y[o2];
                va[u] = s.a[o2];
            }
            {   // (also when the stage needs no dx: an array assigned under a condition stays in scratch memory)
                const f32x4 *src = reinterpret_cast<const f32x4 *>(s.W);
#pragma unroll
                for (int u = 0; u < NW; u++) vw[u] = src[min(t + u * 256, jc * tk - 1)];
            }
            const int rr = min(r0 + (t & (RBB - 1)), b - 1);
            const float mean_t = s.mean[rr], inv_t = s.invstd[rr], gam_t = s.gamma[min(t, dout - 1)];
#pragma unroll
            for (int u = 0; u < NE; u++) {
                const int e = t + u * 256;
                if (!(e < RBB * din && r0 + e / din < b)) vx[u] = 0.f;
                if (!(e < RBB * dout && r0 + e / dout < b)) { vy[u] = 0.f; va[u] = 0.f; }
            }
            if (t < RBB) {
                const bool live = r0 + t < b;
                rmean[t] = live ? mean_t : 0.f;
                rinv[t] = live ? inv_t : 0.f;
            }
            if (t < dout) gam[t] = gam_t;
            __syncthreads();                   // (rmean / rinv / gam; and the previous stage's C has finished writing gz)
#pragma unroll
            for (int u = 0; u < NE; u++) {
                const int e = t + u * 256;
                if (e < RBB * din) { const int r = e / din; xs[r][e - r * din] = vx[u]; }
                if (e < RBB * dout) {
                    // A0 (element-wise): dz = dy act'(y) -> gz, xhat = (a - mean) invstd -> xh
                    const int r = e / dout, j = e - r * dout;
                    const bool live = r0 + r < b;
                    gz[r][j] = live ? gz[r][j] * (vy[u] > 0.f ? 1.f : s.slope) : 0.f;
                    xh[r][j] = live ? (va[u] - rmean[r]) * rinv[r] : 0.f;
                }
            }
            if (need_dx) {
                f32x4 *dst = reinterpret_cast<f32x4 *>(wst);
#pragma unroll
                for (int u = 0; u < NW; u++) {
                    const int e = t + u * 256;
                    if (e < jc * tk) dst[e] = vw[u];
                }
            }
        }
        __syncthreads();
        // ---- A1: per row (32 lanes each): da = invstd (dz gamma - mean_j(dz gamma) - xhat mean_j(dz gamma xhat))
        {
            const int r = t / LPR, sl = t % LPR;
            const float inv = rinv[r];
            float s1 = 0.f, s2 = 0.f;
            for (int j = sl; j < dout; j += LPR) {
                const float dg = gz[r][j] * gam[j];
                s1 += dg;
                s2 += dg * xh[r][j];
            }
#pragma unroll
            for (int off = LPR / 2; off > 0; off >>= 1) { s1 += __shfl_xor(s1, off, WAVE); s2 += __shfl_xor(s2, off, WAVE); }
            const float m1 = s1 / (float)dout, m2 = s2 / (float)dout;
            for (int j = sl; j < dout; j += LPR) da[r][j] = inv * (gz[r][j] * gam[j] - m1 - xh[r][j] * m2);
        }
        __syncthreads();
        // ---- B: weight-gradient partial  dW[j][k] = sum_r da[r][j] xs[r][k], 8 (j) x 4 (k) register tiles
        {
            const int ntile = (dout / 8) * tk;
            for (int tile = t; tile < ntile; tile += 256) {
                const int j0 = (tile / tk) * 8, k0 = (tile - (tile / tk) * tk) * 4;
                float acc[8][4];
#pragma unroll
                for (int u = 0; u < 8; u++)
#pragma unroll
                    for (int v = 0; v < 4; v++) acc[u][v] = 0.f;
#pragma unroll
                for (int r = 0; r < RBB; r++) {
                    const float4 d0 = *reinterpret_cast<const float4 *>(&da[r][j0]);
                    const float4 d1 = *reinterpret_cast<const float4 *>(&da[r][j0 + 4]);
                    const float4 xv = *reinterpret_cast<const float4 *>(&xs[r][k0]);
                    const float dv[8] = {d0.x, d0.y, d0.z, d0.w, d1.x, d1.y, d1.z, d1.w};
                    const float xw[4] = {xv.x, xv.y, xv.z, xv.w};
#pragma unroll
                    for (int u = 0; u < 8; u++)
#pragma unroll
                        for (int v = 0; v < 4; v++) acc[u][v] += dv[u] * xw[v];
                }
#pragma unroll
                for (int u = 0; u < 8; u++)
                    *reinterpret_cast<float4 *>(wrow + s.ws_off + (size_t)(j0 + u) * din + k0) =
                        make_float4(acc[u][0], acc[u][1], acc[u][2], acc[u][3]);
            }
        }
        // ---- D: column partials over the rows: bias (sum da), gamma (sum dz xhat), beta (sum dz)
        for (int j = t; j < dout; j += 256) {
            float sb = 0.f, sg = 0.f, sbeta = 0.f;
#pragma unroll
            for (int r = 0; r < RBB; r++) {
                const float dz = gz[r][j];
                sb += da[r][j];
                sg += dz * xh[r][j];
                sbeta += dz;
            }
            float *p = wrow + s.ws_off + (size_t)dout * din;
            p[j] = sb;
            p[dout + j] = sg;
            p[2 * dout + j] = sbeta;
        }
        __syncthreads();                       // D is done with gz and xh: C writes dx into gz, its halves meet in xh
        // ---- C: dx[r][k] = sum_j da[r][j] W[j][k]: the next (lower) stage's incoming gradient, or the chain's result.
        // Item = (row, 4 columns); the two halves of the workgroup take the two halves of each chunk's j range.
        if (need_dx) {
            const int nitem = RBB * tk;                             // <= 8 * 64 = 512: two passes of 256 at most
            const int half = t >> 7, it0 = t & 127;
            float4 acc[4];
#pragma unroll
            for (int u = 0; u < 4; u++) acc[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            for (int jb = 0; jb < dout; jb += jc) {
                const int jn = min(jc, dout - jb);
                if (jb > 0) {
                    __syncthreads();
                    const float4 *src = reinterpret_cast<const float4 *>(s.W + (size_t)jb * din);
                    float4 *dst = reinterpret_cast<float4 *>(wst);
#pragma unroll 4
                    for (int e = t; e < jn * tk; e += 256) dst[e] = src[e];
                    __syncthreads();
                }
                const int jh = (jn + 1) / 2, ja = half * jh, jz = min(jn, ja + jh);
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const int e = it0 + u * 128;
                    if (e < nitem) {
                        const int r = e / tk, k0 = (e - r * tk) * 4;
                        float4 a4 = acc[u];
#pragma unroll 8
                        for (int j = ja; j < jz; j++) {
                            const float d = da[r][jb + j];
                            const float4 w = *reinterpret_cast<const float4 *>(wst + (size_t)j * din + k0);
                            a4.x += d * w.x; a4.y += d * w.y; a4.z += d * w.z; a4.w += d * w.w;
                        }
                        acc[u] = a4;
                    }
                }
            }
            // the upper half hands its sums over through xh (free after D), the lower half adds and stores
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int e = it0 + u * 128;
                if (half == 1 && e < nitem) *reinterpret_cast<float4 *>(&xh[0][0] + 4 * e) = acc[u];
            }
            __syncthreads();
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int e = it0 + u * 128;
                if (half == 0 && e < nitem) {
                    const int r = e / tk, k0 = (e - r * tk) * 4;
                    const float4 o = *reinterpret_cast<const float4 *>(&xh[0][0] + 4 * e);
                    const float4 a4 = make_float4(acc[u].x + o.x, acc[u].y + o.y, acc[u].z + o.z, acc[u].w + o.w);
                    if (l > 0) *reinterpret_cast<float4 *>(&gz[r][k0]) = a4;
                    else if (r0 + r < b) {
                        float4 o4 = a4;
                        if (dx_add) {       // a gradient that reaches the chain's input by another path (the cluster terms' dz)
                            const float4 e4 = *reinterpret_cast<const float4 *>(dx_add + (size_t)(r0 + r) * din + k0);
                            o4 = make_float4(a4.x + e4.x, a4.y + e4.y, a4.z + e4.z, a4.w + e4.w);
                        }
                        *reinterpret_cast<float4 *>(dx_out + (size_t)(r0 + r) * din + k0) = o4;
                    }
                }
            }
        }
        __syncthreads();
    }
}

// ---- the reconstruction term on the output map's GEMM result -------------------------------------------------------------
// recon = inv_scale * sum (y - (o + bias))^2 with o = h W^T straight out of the GEMM (SpaDOT.py:89 on decoder.py:20's output
// map): bias add and squared error in one launch (fp64 partials, one per row) and a one-workgroup sum in fixed order; backward: dL/d(o) in bf16 for the two GEMMs AND the bias gradient (column sums, fixed order) in one launch.
__device__ __forceinline__ double block_sum_d(double x, double *sh) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) x += __shfl_xor(x, off, WAVE);
    const int w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[w] = x;
    __syncthreads();
    double t = 0.0;
    for (int k = 0; k < nw; k++) t += sh[k];
    return t;
}

constexpr int SQ_ROWS = 1;                  // rows per workgroup: b workgroups (16 waves per compute unit at b = 1024)
__global__ __launch_bounds__(256) void k_bias_sqerr_part(const float *__restrict__ o, const float *__restrict__ bias,
                                                         const float *__restrict__ y, int b, int G,
                                                         double *__restrict__ part) {
    __shared__ double sh[16];
    const size_t row = (size_t)blockIdx.x * G;
    double acc = 0.0;
    // four column steps per trip: 12 independent loads in flight per thread (the kernel is latency-bound otherwise)
    for (int c0 = threadIdx.x; c0 < G; c0 += 4 * 256) {
        float yo[4], oo[4], bc[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int c = c0 + u * 256;
            const bool on = c < G;
            yo[u] = on ? y[row + c] : 0.f;
            oo[u] = on ? o[row + c] : 0.f;
            bc[u] = on ? bias[c] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const double d = (double)yo[u] - (double)(oo[u] + bc[u]);
            acc += d * d;
        }
    }
    acc = block_sum_d(acc, sh);
    if (threadIdx.x == 0) part[blockIdx.x] = acc;
}
// (Not a last-workgroup-finishes kernel: the device-scope release each workgroup needs for that writes the XCD's L2 back, and
// behind a GEMM that has just left 12 MB of dirty output there the one-launch form took 19-38 us against 5 + 5 for two.)
__global__ __launch_bounds__(1024) void k_sum_parts(const double *__restrict__ part, int nparts, double scale,
                                                    float *__restrict__ out) {
    __shared__ double sh[16];
    double acc = 0.0;
    for (int k = threadIdx.x; k < nparts; k += 1024) acc += part[k];
    acc = block_sum_d(acc, sh);
    if (threadIdx.x == 0) out[0] = (float)(acc * scale);
}

constexpr int SB_COLS = 32, SB_GY = 32;
__global__ __launch_bounds__(SB_COLS * SB_GY) void k_bias_sqerr_bwd(const float *__restrict__ g1, const float *__restrict__ o,
                                                                    const float *__restrict__ bias, const float *__restrict__ y,
                                                                    int b, int G, double inv_scale, __bf16 *__restrict__ g,
                                                                    float *__restrict__ dbias) {
    __shared__ float sh[SB_GY][SB_COLS + 1];
    const int cx = threadIdx.x & (SB_COLS - 1), gy = threadIdx.x / SB_COLS;
    const int c = blockIdx.x * SB_COLS + cx;
    const double coef = -2.0 * inv_scale * (double)g1[0];
    float acc = 0.f;
    if (c < G) {
        const float bc = bias[c];
        const int per = (b + SB_GY - 1) / SB_GY;
        const int ra = gy * per, rz = min(b, ra + per);
        for (int r = ra; r < rz; r += 8) {                           // eight rows in flight, added in ascending order
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const size_t e = (size_t)(r + u) * G + c;
                v[u] = (r + u < rz) ? (float)(coef * ((double)y[e] - (double)(o[e] + bc))) : 0.f;
            }
#pragma unroll
            for (int u = 0; u < 8; u++)
                if (r + u < rz) { g[(size_t)(r + u) * G + c] = (__bf16)v[u]; acc += v[u]; }
        }
    }
    sh[gy][cx] = acc;
    __syncthreads();
    if (gy != 0 || c >= G) return;
    acc = 0.f;
#pragma unroll
    for (int q = 0; q < SB_GY; q++) acc += sh[q][cx];
    dbias[c] = acc;
}

// ---- GAT_fc: the [b x K] bf16 output of the last GAT layer -> (mu | logvar) [b x N] fp32, N <= 32 (encoder.py:59-61) ----------
// forward: cast + library GEMM (a one-launch forward from the bf16 rows measured 18-35 us at the end of the forward pair
// against 5 + 7 us, round 2; removed in round 5).
// backward: 8 rows per block -- dh = g W written in bf16 (what the GAT layer's backward reads) and the block's partials of dW and
// db -- then one k_colsum_parts launch adds the blocks in order (a thread-per-column form without partials was a chain of 64
// dependent memory round trips).
constexpr int FC_MAXN = 32;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__global__ __launch_bounds__(256) void k_headfc_bwd(const float *__restrict__ g, const __bf16 *__restrict__ h,
                                                    const float *__restrict__ W, int b, int K, int N,
                                                    __bf16 *__restrict__ dh, float *__restrict__ ws) {
    // 8 rows per block: dh[r, k] = sum_j g[r, j] W[j, k] (bf16), and this block's partial of dW[j, k] = sum_r g[r, j] h[r, k]
    // and db[j] = sum_r g[r, j] -> ws[block][N K + N]; spadot_colsum adds the blocks in order.
    __shared__ float gs[8][FC_MAXN];
    const int r0 = blockIdx.x * 8;
    float *wrow = ws + (size_t)blockIdx.x * ((size_t)N * K + N);
    for (int e = threadIdx.x; e < 8 * N; e += 256) {
        const int r = e / N, j = e - r * N;
        gs[r][j] = r0 + r < b ? g[(size_t)(r0 + r) * N + j] : 0.f;
    }
    __syncthreads();
    for (int k0 = threadIdx.x * 8; k0 < K; k0 += 256 * 8) {
        // the 8 rows' pieces of h first (independent 16-byte loads), then W row by row
        float hf[8][8];
#pragma unroll
        for (int r = 0; r < 8; r++) {
            bf16x8 hv;
#pragma unroll
            for (int e = 0; e < 8; e++) hv[e] = (__bf16)0.f;
            if (r0 + r < b) hv = *reinterpret_cast<const bf16x8 *>(h + (size_t)(r0 + r) * K + k0);
#pragma unroll
            for (int e = 0; e < 8; e++) hf[r][e] = (float)hv[e];
        }
        float acc[8][8];
#pragma unroll
        for (int r = 0; r < 8; r++)
#pragma unroll
            for (int e = 0; e < 8; e++) acc[r][e] = 0.f;
        for (int j = 0; j < N; j++) {
            const float4 w0 = *reinterpret_cast<const float4 *>(W + (size_t)j * K + k0);
            const float4 w1 = *reinterpret_cast<const float4 *>(W + (size_t)j * K + k0 + 4);
            const float wv[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
            float pw[8];
#pragma unroll
            for (int e = 0; e < 8; e++) pw[e] = 0.f;
#pragma unroll
            for (int r = 0; r < 8; r++) {
                const float gv = gs[r][j];
#pragma unroll
                for (int e = 0; e < 8; e++) { acc[r][e] += gv * wv[e]; pw[e] += gv * hf[r][e]; }
            }
            *reinterpret_cast<float4 *>(wrow + (size_t)j * K + k0) = make_float4(pw[0], pw[1], pw[2], pw[3]);
            *reinterpret_cast<float4 *>(wrow + (size_t)j * K + k0 + 4) = make_float4(pw[4], pw[5], pw[6], pw[7]);
        }
#pragma unroll
        for (int r = 0; r < 8; r++)
            if (r0 + r < b) {
                bf16x8 o;
#pragma unroll
                for (int e = 0; e < 8; e++) o[e] = (__bf16)acc[r][e];
                *reinterpret_cast<bf16x8 *>(dh + (size_t)(r0 + r) * K + k0) = o;
            }
    }
    if ((int)threadIdx.x < N) {
        float sb = 0.f;
#pragma unroll
        for (int r = 0; r < 8; r++) sb += gs[r][threadIdx.x];
        wrow[(size_t)N * K + threadIdx.x] = sb;
    }
}

bool fill_chain(Chain &ch, int n_layers, const int *dims, const float *const *W, const float *const *bias,
                const float *const *gamma, const float *const *beta, const double *eps, const double *slope, float *const *a,
                float *const *y, float *const *mean, float *const *invstd) {
    if (n_layers < 1 || n_layers > MAXL) return false;
    int off = 0;
    for (int l = 0; l < n_layers; l++) {
        const int din = dims[l], dout = dims[l + 1];
        if (din <= 0 || dout <= 0 || din > MAXD || dout > MAXD || din % 4 || dout % 8) return false;
        Stage &s = ch.s[l];
        s.W = W[l]; s.bias = bias ? bias[l] : nullptr; s.gamma = gamma[l]; s.beta = beta ? beta[l] : nullptr;
        s.a = a[l]; s.y = y[l]; s.mean = mean[l]; s.invstd = invstd[l];
        s.din = din; s.dout = dout; s.eps = eps ? (float)eps[l] : 0.f; s.slope = (float)slope[l];
        s.ws_off = off;
        off += dout * din + 3 * dout;
    }
    ch.n = n_layers;
    ch.ybf = nullptr;
    return true;
}

}  // namespace

extern "C" {

int spadot_mlp_chain_supported(int n_layers, const int *dims) {
    if (n_layers < 1 || n_layers > MAXL) return 0;
    for (int l = 0; l < n_layers; l++)
        if (dims[l] <= 0 || dims[l + 1] <= 0 || dims[l] > MAXD || dims[l + 1] > MAXD || dims[l] % 4 || dims[l + 1] % 8) return 0;
    return 1;
}

int spadot_mlp_chain_workspace(int b, int n_layers, const int *dims, int *n_rows, int *width) {
    if (!spadot_mlp_chain_supported(n_layers, dims) || b <= 0) return -22;
    int w = 0;
    for (int l = 0; l < n_layers; l++) w += dims[l + 1] * dims[l] + 3 * dims[l + 1];
    *n_rows = (b + RBB - 1) / RBB;
    *width = w;
    return 0;
}

int spadot_mlp_chain_forward_bf16(const float *x, int b, int n_layers, const int *dims, const float *const *W,
                                  const float *const *bias, const float *const *gamma, const float *const *beta,
                                  const double *eps, const double *slope, float *const *a, float *const *y, float *const *mean,
                                  float *const *invstd, void *y_last_bf16, void *stream) {
    Chain ch;
    if (b <= 0 || !fill_chain(ch, n_layers, dims, W, bias, gamma, beta, eps, slope, a, y, mean, invstd)) return -22;
    ch.ybf = (__bf16 *)y_last_bf16;                      // may be null
    hipLaunchKernelGGL(k_mlp_chain_fwd, dim3((b + RBF - 1) / RBF), dim3(256), 0, (hipStream_t)stream, x, b, ch);
    return hipGetLastError() == hipSuccess ? 0 : -5;
}

int spadot_mlp_chain_forward(const float *x, int b, int n_layers, const int *dims, const float *const *W,
                             const float *const *bias, const float *const *gamma, const float *const *beta, const double *eps,
                             const double *slope, float *const *a, float *const *y, float *const *mean, float *const *invstd,
                             void *stream) {
    return spadot_mlp_chain_forward_bf16(x, b, n_layers, dims, W, bias, gamma, beta, eps, slope, a, y, mean, invstd, nullptr, stream);
}

int spadot_mlp_chain_backward_add(const float *dy, const float *x, int b, int n_layers, const int *dims, const float *const *W,
                                  const float *const *gamma, const double *slope, float *const *a, float *const *y,
                                  float *const *mean, float *const *invstd, float *dx, const float *dx_add, float *workspace,
                                  float *grads, void *stream) {
    Chain ch;
    if (b <= 0 || !fill_chain(ch, n_layers, dims, W, nullptr, gamma, nullptr, nullptr, slope, a, y, mean, invstd)) return -22;
    if (dx_add && !dx) return -22;
    int rows, width;
    if (spadot_mlp_chain_workspace(b, n_layers, dims, &rows, &width)) return -22;
    constexpr int LDS_BWD = LDS_BWD_FLOATS * (int)sizeof(float);
    static PerDeviceFlag attr_set;     
    if (!attr_set) {
        if (hipFuncSetAttribute((const void *)k_mlp_chain_bwd, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BWD) != hipSuccess) return -5;
        attr_set = true;
    }
    hipLaunchKernelGGL(k_mlp_chain_bwd, dim3(rows), dim3(256), LDS_BWD, (hipStream_t)stream, dy, x, b, ch, dx, workspace, width, dx_add);
    if (hipGetLastError() != hipSuccess) return -5;
    if (!grads) return 0;                                            // (the caller sums the partials later: spadot_colsum)
    return spadot_colsum(workspace, rows, width, grads, stream);     // grads: one workspace row = [dW | dbias | dgamma | dbeta] per stage
}

int spadot_mlp_chain_backward(const float *dy, const float *x, int b, int n_layers, const int *dims, const float *const *W,
                              const float *const *gamma, const double *slope, float *const *a, float *const *y,
                              float *const *mean, float *const *invstd, float *dx, float *workspace, float *grads, void *stream) {
    return spadot_mlp_chain_backward_add(dy, x, b, n_layers, dims, W, gamma, slope, a, y, mean, invstd, dx, nullptr, workspace, grads,
                                         stream);
}

int spadot_bias_sqerr_forward(const float *o, const float *bias, const float *y, int b, int G, double inv_scale, double *scratch,
                              float *out, void *stream) {
    const int nb = (b + SQ_ROWS - 1) / SQ_ROWS;                    // one partial per workgroup: scratch holds 4096 doubles
    if (b <= 0 || G <= 0 || !scratch || nb > 4096) return -22;
    hipLaunchKernelGGL(k_bias_sqerr_part, dim3(nb), dim3(256), 0, (hipStream_t)stream, o, bias, y, b, G, scratch);
    hipLaunchKernelGGL(k_sum_parts, dim3(1), dim3(1024), 0, (hipStream_t)stream, scratch, nb, inv_scale, out);
    return hipGetLastError() == hipSuccess ? 0 : -5;
}

int spadot_bias_sqerr_backward(const float *g1, const float *o, const float *bias, const float *y, int b, int G,
                               double inv_scale, void *g_bf16, float *dbias, void *stream) {
    if (b <= 0 || G <= 0) return -22;
    hipLaunchKernelGGL(k_bias_sqerr_bwd, dim3((G + SB_COLS - 1) / SB_COLS), dim3(SB_COLS * SB_GY), 0, (hipStream_t)stream, g1, o,
                       bias, y, b, G, inv_scale, (__bf16 *)g_bf16, dbias);
    return hipGetLastError() == hipSuccess ? 0 : -5;
}

int spadot_headfc_backward(const float *g, const void *h_bf16, const float *W, int b, int K, int N, void *dh_bf16,
                           float *workspace, float *grads, void *stream) {
    if (b <= 0 || K <= 0 || K % 8 || N <= 0 || N > FC_MAXN || !workspace) return -22;
    const int blocks = (b + 7) / 8;
    hipLaunchKernelGGL(k_headfc_bwd, dim3(blocks), dim3(256), 0, (hipStream_t)stream, g, (const __bf16 *)h_bf16, W, b, K, N,
                       (__bf16 *)dh_bf16, workspace);
    if (hipGetLastError() != hipSuccess) return -5;
    if (!grads) return 0;                                                  // (the caller sums the partials later: spadot_colsum)
    return spadot_colsum(workspace, blocks, N * K + N, grads, stream);     // grads = [dW (N x K) | db (N)]
}

}  // extern "C"
