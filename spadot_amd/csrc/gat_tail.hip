// gat_tail.hip -- the LAST GAT layer of the encoder for the seeds only, aggregate-first (round 4).
//
// encoder.py:45,58 of the reference: gat3 = GATConv(H*C -> C, heads = H, concat = False), and only the first b rows of
// its output (the seeds) reach the loss (SpaDOT.py:82).  GATConv is linear after the softmax:
//
//     out_i = 1/H sum_h sum_j alpha_ij^h (x_j W_h^T) + bias  =  1/H sum_h ( sum_j alpha_ij^h x_j ) W_h^T + bias
//     e_ij^h = leaky_relu( (x_j W_h^T) . att_src^h + (x_i W_h^T) . att_dst^h )  =  leaky_relu( x_j . w_src^h + x_i . w_dst^h ),
//              w_src^h = W_h^T att_src^h,  w_dst^h = W_h^T att_dst^h            (W_h = rows h C .. h C + C - 1 of lin.weight)
//
// With n_tgt = 512 targets and n ~ 8000 source rows the dense map x W^T over all source rows (67 GFLOP forward, twice that
// backward at the benchmarked shape: 92 + 61 + 82 us of matrix-core time per step) is replaced by: the logit vectors
// w [2H x K] (16 MB read), one pass over x for the logits s = x w^T, the softmax + aggregation A[h][i][:] = sum_j alpha x_j
// over the seeds' ~31 edges each, and dense maps on the n_tgt AGGREGATED rows (4 GFLOP each way).  Same function, an
// order of magnitude less work; the rounding differs from the map-first order only at the level of the compute dtype.
//
// Kernels (all deterministic: fixed summation orders, no atomics), one 256-thread workgroup each, thread t owning the
// eight consecutive input channels k = 8 t .. 8 t + 7 (K <= 2048, K % 8 == 0):
//   k_tail_wvec / _fin     w[q][k] = sum_c W[h C + c][k] att[h][c]              (q = 2 h: src, 2 h + 1: dst)
//   k_tail_logits          s[j][q] = x_j . w[q]
//   k_tail_aggregate       alpha over the incoming edges of seed i (exp(e - max) / (sum + 1e-16)), A[h][i][:]
//   k_tail_headmean        out[i][c] = 1/H sum_h O[h][i][c] + bias[c]
//   k_tail_edge_bwd        d alpha = <dA[h][i], x_j>, softmax + leaky_relu backward -> d logit per edge, ds_dst
//   k_tail_src_bwd         dx_j = sum_{e: j -> i} sum_h alpha dA[h][i] + sum_h ds_src w_src^h + ds_dst w_dst^h  (transposed CSR)
//   k_tail_dwvec_part      dw[q][k] partials = sum_j dS[j][q] x[j][k] per block of rows   (+ spadot_colsum)
//   k_tail_wvec_bwd        dW[h C + c][k] += att_src dw_src + att_dst dw_dst ; datt = W dw
//   k_tail_colsum_rows     bias gradient: column sums of the [n_tgt x C] output gradient
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>

#include "../../include/spadot_model.h"

namespace {

constexpr int WAVE = 64;
constexpr int NT = 256;
constexpr float ATT_SLOPE = 0.2f;     // GATConv negative_slope
constexpr int MAXQ = 16;              // 2 H, H <= 8

__device__ __forceinline__ unsigned bf16_pack2(float lo, float hi) {
    const unsigned short a = __builtin_bit_cast(unsigned short, (__bf16)lo), b = __builtin_bit_cast(unsigned short, (__bf16)hi);
    return (unsigned)a | ((unsigned)b << 16);
}
// eight consecutive elements <-> fp32 registers (16 bytes of bf16, 32 bytes of fp32; p 16-byte aligned)
template <typename T> __device__ __forceinline__ void load8(const T *p, float *o);
template <> __device__ __forceinline__ void load8<float>(const float *p, float *o) {
    const float4 a = *reinterpret_cast<const float4 *>(p), b = *reinterpret_cast<const float4 *>(p + 4);
    o[0] = a.x; o[1] = a.y; o[2] = a.z; o[3] = a.w; o[4] = b.x; o[5] = b.y; o[6] = b.z; o[7] = b.w;
}
template <> __device__ __forceinline__ void load8<__bf16>(const __bf16 *p, float *o) {
    const uint4 v = *reinterpret_cast<const uint4 *>(p);
    o[0] = __uint_as_float(v.x << 16); o[1] = __uint_as_float(v.x & 0xffff0000u);
    o[2] = __uint_as_float(v.y << 16); o[3] = __uint_as_float(v.y & 0xffff0000u);
    o[4] = __uint_as_float(v.z << 16); o[5] = __uint_as_float(v.z & 0xffff0000u);
    o[6] = __uint_as_float(v.w << 16); o[7] = __uint_as_float(v.w & 0xffff0000u);
}
template <typename T> __device__ __forceinline__ void store8(T *p, const float *o);
template <> __device__ __forceinline__ void store8<float>(float *p, const float *o) {
    *reinterpret_cast<float4 *>(p) = make_float4(o[0], o[1], o[2], o[3]);
    *reinterpret_cast<float4 *>(p + 4) = make_float4(o[4], o[5], o[6], o[7]);
}
template <> __device__ __forceinline__ void store8<__bf16>(__bf16 *p, const float *o) {
    *reinterpret_cast<uint4 *>(p) = make_uint4(bf16_pack2(o[0], o[1]), bf16_pack2(o[2], o[3]), bf16_pack2(o[4], o[5]),
                                               bf16_pack2(o[6], o[7]));
}

// the same eight elements kept PACKED until they are used (bf16: 4 registers instead of 8)
// (native vector types: arrays of HIP's float4 / uint4 classes are not promoted to registers and end up in scratch memory)
typedef float tf32x4 __attribute__((ext_vector_type(4)));
typedef unsigned tu32x4 __attribute__((ext_vector_type(4)));
template <typename T> struct Raw8;
template <> struct Raw8<float> { tf32x4 a, b; };
template <> struct Raw8<__bf16> { tu32x4 v; };
template <typename T> __device__ __forceinline__ Raw8<T> load8_raw(const T *p);
template <> __device__ __forceinline__ Raw8<float> load8_raw<float>(const float *p) {
    Raw8<float> r;
    r.a = *reinterpret_cast<const tf32x4 *>(p); r.b = *reinterpret_cast<const tf32x4 *>(p + 4);
    return r;
}
template <> __device__ __forceinline__ Raw8<__bf16> load8_raw<__bf16>(const __bf16 *p) {
    Raw8<__bf16> r;
    r.v = *reinterpret_cast<const tu32x4 *>(p);
    return r;
}
__device__ __forceinline__ void unpack8(const Raw8<float> &r, float *o) {
#pragma unroll
    for (int e = 0; e < 4; e++) { o[e] = r.a[e]; o[4 + e] = r.b[e]; }
}
__device__ __forceinline__ void unpack8(const Raw8<__bf16> &r, float *o) {
#pragma unroll
    for (int e = 0; e < 4; e++) { o[2 * e] = __uint_as_float(r.v[e] << 16); o[2 * e + 1] = __uint_as_float(r.v[e] & 0xffff0000u); }
}

__device__ __forceinline__ float wave_sum_f(float x) {
#pragma unroll
    for (int off = WAVE / 2; off > 0; off >>= 1) x += __shfl_xor(x, off, WAVE);
    return x;
}
__device__ __forceinline__ float wave_max_f(float x) {
#pragma unroll
    for (int off = WAVE / 2; off > 0; off >>= 1) x = fmaxf(x, __shfl_xor(x, off, WAVE));
    return x;
}
__device__ __forceinline__ float leaky(float z, float slope) { return z > 0.f ? z : slope * z; }

// Sums of NV per-lane values over the 64 lanes of a wave with NV - 1 + log2(64 / NV) shuffles instead of 6 NV: in
// halving step s the lanes whose bit s is set hand over the lower half of what they hold and keep the upper half (the
// others the reverse), so after log2(NV) steps every lane holds ONE value -- index bitrev(lane mod NV) -- summed over the
// NV lanes that differ in their low bits; full exchanges over the remaining lane bits finish it.  Returns that total;
// `index_of_lane` says which value a lane ends up with.  NV a power of two, 1 <= NV <= 64.
template <int NV> __device__ __forceinline__ int butterfly_index(int lane) {
    int idx = 0;
#pragma unroll
    for (int s = 0, n = NV >> 1; n >= 1; s++, n >>= 1)
        if (lane & (1 << s)) idx |= n;
    return idx;
}
template <int NV> __device__ __forceinline__ float butterfly_sum(float *v, int lane) {
    int s = 0;
#pragma unroll
    for (int n = NV >> 1; n >= 1; n >>= 1, s++) {
        const int m = 1 << s;
        const bool up = (lane & m) != 0;
#pragma unroll
        for (int i = 0; i < n; i++) {
            const float keep = up ? v[i + n] : v[i];
            const float send = up ? v[i] : v[i + n];
            v[i] = keep + __shfl_xor(send, m, WAVE);
        }
    }
    float r = v[0];
#pragma unroll
    for (int m = NV; m < WAVE; m <<= 1) r += __shfl_xor(r, m, WAVE);
    return r;
}

// ------------------------------------------------------------------------------------------------------------------
// w[q][k] = sum_c W[(h C + c) ldw + k] att[h][c]:  grid (ceil(K / 1024), H, S): thread = four consecutive k, slice s of the
// channels c; partials [S][2 H][K] summed in slice order by k_tail_wvec_fin.
// ------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(NT) void k_tail_wvec(const float *__restrict__ W, int ldw, const float *__restrict__ a_src,
                                                  const float *__restrict__ a_dst, int H, int C, int K, int S,
                                                  float *__restrict__ part) {
    const int k = ((int)blockIdx.x * NT + (int)threadIdx.x) * 4;
    const int h = blockIdx.y, s = blockIdx.z;
    if (k >= K) return;
    const int cs = (C + S - 1) / S, c0 = s * cs, c1 = min(C, c0 + cs);
    float as_[4] = {0.f, 0.f, 0.f, 0.f}, ad_[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 8
    for (int c = c0; c < c1; c++) {
        const float4 w = *reinterpret_cast<const float4 *>(W + (size_t)(h * C + c) * ldw + k);
        const float a = a_src[h * C + c], d = a_dst[h * C + c];
        as_[0] = fmaf(w.x, a, as_[0]); as_[1] = fmaf(w.y, a, as_[1]); as_[2] = fmaf(w.z, a, as_[2]); as_[3] = fmaf(w.w, a, as_[3]);
        ad_[0] = fmaf(w.x, d, ad_[0]); ad_[1] = fmaf(w.y, d, ad_[1]); ad_[2] = fmaf(w.z, d, ad_[2]); ad_[3] = fmaf(w.w, d, ad_[3]);
    }
    float *p = part + ((size_t)s * 2 * H + 2 * h) * K + k;
    *reinterpret_cast<float4 *>(p) = make_float4(as_[0], as_[1], as_[2], as_[3]);
    *reinterpret_cast<float4 *>(p + K) = make_float4(ad_[0], ad_[1], ad_[2], ad_[3]);
}

__global__ __launch_bounds__(NT) void k_tail_wvec_fin(const float *__restrict__ part, int S, int total, float *__restrict__ wv,
                                                      __bf16 *__restrict__ whi, __bf16 *__restrict__ wlo) {
    const int t = (int)blockIdx.x * NT + (int)threadIdx.x;          // one output per thread: total / 256 workgroups
    if (t >= total) return;
    float acc = 0.f;
#pragma unroll 8
    for (int s = 0; s < S; s++) acc += part[(size_t)s * total + t];
    wv[t] = acc;
    if (whi != nullptr) {        // w = hi + lo in bf16 (16 significant bits) for the matrix-core logits kernel
        const __bf16 h = (__bf16)acc;
        whi[t] = h;
        wlo[t] = (__bf16)(acc - (float)h);
    }
}

typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));

// ------------------------------------------------------------------------------------------------------------------
// The same logits on the matrix cores (bf16 rows, K % 32 == 0): a skinny product x [n x K] . w^T [K x Q] as
// v_mfma_f32_16x16x32_bf16 tiles -- 16 rows of x per workgroup (A: row = lane & 15, k = 8 (lane >> 4) + j: one 16-byte load
// per lane straight from global), w = hi + lo (B: col = lane & 15 = q, zero for q >= Q), the four waves taking every fourth
// 32-wide slice of the contraction, one LDS exchange at the end.  Every load of a wave is independent of every other, so
// the whole 33 MB of x is in flight at once: the vector form above is latency-bound (one row group per barrier pair).
// ------------------------------------------------------------------------------------------------------------------
template <int Q>
__global__ __launch_bounds__(NT) void k_tail_logits_mfma(const __bf16 *__restrict__ x, int ldx, const __bf16 *__restrict__ whi,
                                                         const __bf16 *__restrict__ wlo, int n, int K, float *__restrict__ s_out) {
    __shared__ float red[4][16][17];
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int j0 = (int)blockIdx.x * 16;
    const int r = lane & 15, kq = lane >> 4;
    const __bf16 *xr = x + (size_t)min(j0 + r, n - 1) * ldx + 8 * kq;
    // (columns q >= Q of the tile are never stored: their B lanes read row Q - 1 again -- every load unconditional, no
    // divergent control flow in the loop, so the compiler can keep a batch of loads in flight)
    const __bf16 *wh = whi + (size_t)min(r, Q - 1) * K + 8 * kq, *wl = wlo + (size_t)min(r, Q - 1) * K + 8 * kq;
    f4 acc = {0.f, 0.f, 0.f, 0.f};
    const int KC = K >> 5;
    constexpr int UB = 8;                                       // slices per batch: UB x 3 x 16 bytes per lane in flight
    for (int kc = w; kc < KC; kc += 4 * UB) {
        bf8 a[UB], bh[UB], bl[UB];
#pragma unroll
        for (int u = 0; u < UB; u++) {
            const int kk = (kc + 4 * u < KC) ? kc + 4 * u : w;  // (past the end: a valid slice again, its product is skipped)
            a[u] = *reinterpret_cast<const bf8 *>(xr + 32 * kk);
            bh[u] = *reinterpret_cast<const bf8 *>(wh + 32 * kk);
            bl[u] = *reinterpret_cast<const bf8 *>(wl + 32 * kk);
        }
#pragma unroll
        for (int u = 0; u < UB; u++)
            if (kc + 4 * u < KC) {                              // wave-uniform
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[u], bh[u], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[u], bl[u], acc, 0, 0, 0);
            }
    }
#pragma unroll
    for (int i = 0; i < 4; i++) red[w][4 * kq + i][r] = acc[i];          // C/D: col = lane & 15, row = 4 (lane >> 4) + i
    __syncthreads();
    const int rr = t >> 4, qq = t & 15;
    if (qq < Q && j0 + rr < n) s_out[(size_t)(j0 + rr) * Q + qq] = red[0][rr][qq] + red[1][rr][qq] + red[2][rr][qq] + red[3][rr][qq];
}

// ------------------------------------------------------------------------------------------------------------------
// s[j][q] = x_j . w[q]  (q < Q = 2 H).  The workgroup keeps its slice of w in registers (Q x 8 per thread) and walks
// ROWS rows, RG at a time: Q RG partial dots per lane, one butterfly, one LDS exchange between the four waves.
// ------------------------------------------------------------------------------------------------------------------
template <typename T, int Q, int RG>
__global__ __launch_bounds__(NT, 2) void k_tail_logits(const T *__restrict__ x, int ldx, const float *__restrict__ wv, int n, int K,
                                                    int rows_per_wg, float *__restrict__ s_out) {
    constexpr int NV = Q * RG;
    __shared__ float red[4][NV];
    const int t = threadIdx.x, lane = t & 63, wid = t >> 6;
    const int k = t * 8;
    const bool live = k < K;
    float w[Q][8];
#pragma unroll
    for (int q = 0; q < Q; q++) {
        if (live) load8<float>(wv + (size_t)q * K + k, w[q]);
        else
#pragma unroll
            for (int e = 0; e < 8; e++) w[q][e] = 0.f;
    }
    const int j0 = (int)blockIdx.x * rows_per_wg, j1 = min(n, j0 + rows_per_wg);
    const int my = butterfly_index<NV>(lane);
#pragma unroll 1
    for (int jb = j0; jb < j1; jb += RG) {
        float xv[RG][8];
#pragma unroll
        for (int r = 0; r < RG; r++) {
            const int j = min(jb + r, j1 - 1);                    // (rows past the end repeat the last one; not stored)
            if (live) load8<T>(x + (size_t)j * ldx + k, xv[r]);
            else
#pragma unroll
                for (int e = 0; e < 8; e++) xv[r][e] = 0.f;
        }
        float v[NV];
#pragma unroll
        for (int r = 0; r < RG; r++)
#pragma unroll
            for (int q = 0; q < Q; q++) {
                float a = 0.f;
#pragma unroll
                for (int e = 0; e < 8; e++) a = fmaf(xv[r][e], w[q][e], a);
                v[r * Q + q] = a;
            }
        const float tot = butterfly_sum<NV>(v, lane);
        __syncthreads();                                           // (the previous group's readers are done)
        if (lane < NV) red[wid][my] = tot;
        __syncthreads();
        if (t < NV) {
            const int r = t / Q, q = t - r * Q;
            if (jb + r < j1) s_out[(size_t)(jb + r) * Q + q] = red[0][t] + red[1][t] + red[2][t] + red[3][t];
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// Seed i: softmax over its incoming edges per head (wave = head), alpha kept for the backward pass, then
// A[h][i][k] = sum_e alpha[e][h] x[col[e]][k] with every x row read once for all heads.  Edges in chunks of CH.
// ------------------------------------------------------------------------------------------------------------------
template <typename T, int H>
__global__ __launch_bounds__(NT) void k_tail_aggregate(const T *__restrict__ x, int ldx, const float *__restrict__ s,
                                                       const int *__restrict__ rowptr, const int *__restrict__ col, int n_tgt,
                                                       int K, T *__restrict__ A, float *__restrict__ alpha_out) {
    constexpr int CH = 64, Q = 2 * H;
    __shared__ float al[CH][H];
    __shared__ int cs[CH];
    __shared__ float mx[H], den[H];
    const int i = blockIdx.x;
    const int t = threadIdx.x, lane = t & 63, wid = t >> 6;
    const int p0 = rowptr[i], deg = rowptr[i + 1] - p0;
    const int k = t * 8;
    const bool live = k < K;
    if (deg > 0 && deg <= CH) {
        // Every seed of a kNN graph: all edges in one chunk.  Seven dependent round trips instead of ~16 (row pointer; source ids;
        // logit halves; the source rows in batches of eight, every load unconditional -- a batch's tail re-reads the last edge's
        // row with weight zero): the general form below re-fetches ids and logits in each of its three passes and gathers four
        // rows at a time.  Same arithmetic in the same order (wave = head for the statistics, edges ascending in the sums).
        __shared__ float lg[CH][H];
        if (t < CH) cs[t] = col[p0 + min(t, deg - 1)];
        const float sd_t = s[(size_t)i * Q + 2 * (t % H) + 1];       // (thread u: head u % H)
        __syncthreads();
        for (int u = t; u < CH * H; u += NT) {                        // (CH H <= NT for H <= 4; two passes at H = 8)
            const int e = u / H, hd = u - e * H;
            const float sv = s[(size_t)cs[e] * Q + 2 * hd];
            lg[e][hd] = e < deg ? leaky(sv + sd_t, ATT_SLOPE) : -INFINITY;
        }
        __syncthreads();
        for (int hd = wid; hd < H; hd += 4) {
            const float z = lg[lane][hd];
            const float m = wave_max_f(z);
            const float sum = wave_sum_f(lane < deg ? __expf(z - m) : 0.f) + 1e-16f;
            if (lane == 0) { mx[hd] = m; den[hd] = sum; }
        }
        __syncthreads();
        for (int u = t; u < CH * H; u += NT) {
            const int e = u / H, hd = u - e * H;
            const float a = e < deg ? __expf(lg[e][hd] - mx[hd]) / den[hd] : 0.f;
            al[e][hd] = a;
            if (e < deg) alpha_out[(size_t)(p0 + e) * H + hd] = a;
        }
        __syncthreads();
        float acc[H][8];
#pragma unroll
        for (int h = 0; h < H; h++)
#pragma unroll
            for (int e = 0; e < 8; e++) acc[h][e] = 0.f;
        const int kk = live ? k : 0;
        for (int e0 = 0; e0 < deg; e0 += 8) {
            Raw8<T> xv[8];
#pragma unroll
            for (int u = 0; u < 8; u++) xv[u] = load8_raw<T>(x + (size_t)cs[min(e0 + u, CH - 1)] * ldx + kk);
#pragma unroll
            for (int u = 0; u < 8; u++) {
                if (e0 + u < deg) {                                    // (uniform)
                    float x8[8];
                    unpack8(xv[u], x8);
#pragma unroll
                    for (int h = 0; h < H; h++) {
                        const float a = al[e0 + u][h];
#pragma unroll
                        for (int q = 0; q < 8; q++) acc[h][q] = fmaf(a, x8[q], acc[h][q]);
                    }
                }
            }
        }
        if (live)
#pragma unroll
            for (int h = 0; h < H; h++) store8<T>(A + ((size_t)h * n_tgt + i) * K + k, acc[h]);
        return;
    }
    // softmax statistics: wave wid takes heads wid, wid + 4, ...
    for (int hd = wid; hd < H; hd += 4) {
        const float sd = s[(size_t)i * Q + 2 * hd + 1];
        float m = -INFINITY;
        for (int e = lane; e < deg; e += WAVE) m = fmaxf(m, leaky(s[(size_t)col[p0 + e] * Q + 2 * hd] + sd, ATT_SLOPE));
        m = wave_max_f(m);
        float sum = 0.f;
        for (int e = lane; e < deg; e += WAVE) sum += __expf(leaky(s[(size_t)col[p0 + e] * Q + 2 * hd] + sd, ATT_SLOPE) - m);
        sum = wave_sum_f(sum) + 1e-16f;
        if (lane == 0) { mx[hd] = m; den[hd] = sum; }
    }
    float acc[H][8];
#pragma unroll
    for (int h = 0; h < H; h++)
#pragma unroll
        for (int e = 0; e < 8; e++) acc[h][e] = 0.f;
    for (int c0 = 0; c0 < deg; c0 += CH) {
        const int cn = min(CH, deg - c0);
        __syncthreads();                                           // statistics written / previous chunk consumed
        for (int u = t; u < cn * H; u += NT) {
            const int e = u / H, hd = u - e * H;
            const int j = col[p0 + c0 + e];
            const float a = __expf(leaky(s[(size_t)j * Q + 2 * hd] + s[(size_t)i * Q + 2 * hd + 1], ATT_SLOPE) - mx[hd]) / den[hd];
            al[e][hd] = a;
            alpha_out[(size_t)(p0 + c0 + e) * H + hd] = a;
            if (hd == 0) cs[e] = j;
        }
        __syncthreads();
        if (live) {
            int e = 0;
            for (; e + 4 <= cn; e += 4) {                          // four row loads in flight
                float xv[4][8];
#pragma unroll
                for (int u = 0; u < 4; u++) load8<T>(x + (size_t)cs[e + u] * ldx + k, xv[u]);
#pragma unroll
                for (int u = 0; u < 4; u++)
#pragma unroll
                    for (int h = 0; h < H; h++) {
                        const float a = al[e + u][h];
#pragma unroll
                        for (int q = 0; q < 8; q++) acc[h][q] = fmaf(a, xv[u][q], acc[h][q]);
                    }
            }
            for (; e < cn; e++) {
                float xv[8];
                load8<T>(x + (size_t)cs[e] * ldx + k, xv);
#pragma unroll
                for (int h = 0; h < H; h++) {
                    const float a = al[e][h];
#pragma unroll
                    for (int q = 0; q < 8; q++) acc[h][q] = fmaf(a, xv[q], acc[h][q]);
                }
            }
        }
    }
    if (live)
#pragma unroll
        for (int h = 0; h < H; h++) store8<T>(A + ((size_t)h * n_tgt + i) * K + k, acc[h]);
}

// out[i][c] = 1/H sum_h O[h][i][c] + bias[c]
template <typename T>
__global__ __launch_bounds__(NT) void k_tail_headmean(const T *__restrict__ O, const float *__restrict__ bias, int n_tgt, int H, int C,
                                                      T *__restrict__ out) {
    const size_t idx = ((size_t)blockIdx.x * NT + threadIdx.x) * 8, tot = (size_t)n_tgt * C;
    if (idx >= tot) return;
    const int c = (int)(idx % C);
    float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int h = 0; h < H; h++) {
        float v[8];
        load8<T>(O + (size_t)h * tot + idx, v);
#pragma unroll
        for (int e = 0; e < 8; e++) a[e] += v[e];
    }
    const float inv = 1.0f / (float)H;
#pragma unroll
    for (int e = 0; e < 8; e++) a[e] = fmaf(a[e], inv, bias[c + e]);
    store8<T>(out + idx, a);
}

// column sums of g [rows x C] (fp32 accumulation; fixed order: 16 interleaved row groups, then the groups in order):
// block = 16 columns x 16 row groups
template <typename T>
__global__ __launch_bounds__(NT) void k_tail_colsum_rows(const T *__restrict__ g, int rows, int C, float *__restrict__ out) {
    __shared__ float red[16][17];
    const int cl = threadIdx.x & 15, rg = threadIdx.x >> 4;
    const int c = (int)blockIdx.x * 16 + cl;
    float a = 0.f;
    if (c < C)
        for (int r = rg; r < rows; r += 16) a += (float)g[(size_t)r * C + c];
    red[rg][cl] = a;
    __syncthreads();
    if (rg == 0 && c < C) {
        float tot = 0.f;
#pragma unroll
        for (int k = 0; k < 16; k++) tot += red[k][cl];
        out[c] = tot;
    }
}

// The same column sums AND gs = scale * g in the storage type, in one pass over g (the backward's first two launches)
template <typename T>
__global__ __launch_bounds__(NT) void k_tail_scale_colsum(const T *__restrict__ g, int rows, int C, float scale, T *__restrict__ gs,
                                                          float *__restrict__ out) {
    __shared__ float red[16][17];
    const int cl = threadIdx.x & 15, rg = threadIdx.x >> 4;
    const int c = (int)blockIdx.x * 16 + cl;
    float a = 0.f;
    if (c < C)
#pragma unroll 8
        for (int r = rg; r < rows; r += 16) {
            const float v = (float)g[(size_t)r * C + c];
            a += v;
            gs[(size_t)r * C + c] = (T)(v * scale);
        }
    red[rg][cl] = a;
    __syncthreads();
    if (rg == 0 && c < C) {
        float tot = 0.f;
#pragma unroll
        for (int k = 0; k < 16; k++) tot += red[k][cl];
        out[c] = tot;
    }
}

// ------------------------------------------------------------------------------------------------------------------
// Seed i, backward through aggregation and softmax: d alpha[e][h] = <dA[h][i], x_{col[e]}> (EG edges at a time: EG H partial
// dots per lane, one butterfly, one LDS exchange), parked in dz; then per head (wave = head)
//   dz = alpha (d alpha - sum_e alpha d alpha),  d logit = dz * leaky'(raw logit)  -> dz[e][h],  ds_dst[i][h] = sum_e d logit.
// ------------------------------------------------------------------------------------------------------------------
template <typename T, int H, int EG>
__global__ __launch_bounds__(NT, 2) void k_tail_edge_bwd(const T *__restrict__ x, int ldx, const T *__restrict__ dA,
                                                      const float *__restrict__ s, const float *__restrict__ alpha,
                                                      const int *__restrict__ rowptr, const int *__restrict__ col, int n_tgt, int K,
                                                      float *__restrict__ dz, float *__restrict__ ds_dst) {
    constexpr int NV = EG * H, Q = 2 * H;
    __shared__ float red[4][NV];
    const int i = blockIdx.x;
    const int t = threadIdx.x, lane = t & 63, wid = t >> 6;
    const int p0 = rowptr[i], deg = rowptr[i + 1] - p0;
    const int k = t * 8;
    const bool live = k < K;
    float g[H][8];
#pragma unroll
    for (int h = 0; h < H; h++) {
        if (live) load8<T>(dA + ((size_t)h * n_tgt + i) * K + k, g[h]);
        else
#pragma unroll
            for (int e = 0; e < 8; e++) g[h][e] = 0.f;
    }
    const int my = butterfly_index<NV>(lane);
#pragma unroll 1
    for (int e0 = 0; e0 < deg; e0 += EG) {
        float xv[EG][8];
#pragma unroll
        for (int u = 0; u < EG; u++) {
            const int e = min(e0 + u, deg - 1);                    // (edges past the end repeat the last one; not stored)
            if (live) load8<T>(x + (size_t)col[p0 + e] * ldx + k, xv[u]);
            else
#pragma unroll
                for (int q = 0; q < 8; q++) xv[u][q] = 0.f;
        }
        float v[NV];
#pragma unroll
        for (int u = 0; u < EG; u++)
#pragma unroll
            for (int h = 0; h < H; h++) {
                float a = 0.f;
#pragma unroll
                for (int q = 0; q < 8; q++) a = fmaf(g[h][q], xv[u][q], a);
                v[u * H + h] = a;
            }
        const float tot = butterfly_sum<NV>(v, lane);
        __syncthreads();
        if (lane < NV) red[wid][my] = tot;
        __syncthreads();
        if (t < NV) {
            const int u = t / H, h = t - u * H;
            if (e0 + u < deg) dz[(size_t)(p0 + e0 + u) * H + h] = red[0][t] + red[1][t] + red[2][t] + red[3][t];
        }
    }
    __threadfence_block();
    __syncthreads();                                               // d alpha of every edge of this seed is in dz
    for (int hd = wid; hd < H; hd += 4) {
        float dot = 0.f;
        for (int e = lane; e < deg; e += WAVE) dot += alpha[(size_t)(p0 + e) * H + hd] * dz[(size_t)(p0 + e) * H + hd];
        dot = wave_sum_f(dot);
        const float sd = s[(size_t)i * Q + 2 * hd + 1];
        float dsum = 0.f;
        for (int e = lane; e < deg; e += WAVE) {
            const size_t o = (size_t)(p0 + e) * H + hd;
            const float raw = s[(size_t)col[p0 + e] * Q + 2 * hd] + sd;
            const float d = alpha[o] * (dz[o] - dot) * (raw > 0.f ? 1.f : ATT_SLOPE);
            dz[o] = d;
            dsum += d;
        }
        dsum = wave_sum_f(dsum);
        if (lane == 0) ds_dst[(size_t)i * H + hd] = dsum;
    }
}

// ------------------------------------------------------------------------------------------------------------------
// The same kernel with its first phase on the matrix cores (bf16 rows, K % 32 == 0): d alpha[e][h] for 32 edges of the seed at
// a time as two v_mfma_f32_16x16x32_bf16 tiles per 32-wide slice of the contraction (A: the edge's source row straight from
// global, B: dA[h][i] for col = h < H, exact in bf16), the four waves taking every fourth slice, one LDS exchange per 32
// edges -- instead of four edges per barrier pair with a load latency each.
// ------------------------------------------------------------------------------------------------------------------
template <int H>
__global__ __launch_bounds__(NT) void k_tail_edge_bwd_mfma(const __bf16 *__restrict__ x, int ldx, const __bf16 *__restrict__ dA,
                                                           const float *__restrict__ s, const float *__restrict__ alpha,
                                                           const int *__restrict__ rowptr, const int *__restrict__ col, int n_tgt, int K,
                                                           float *__restrict__ dz, float *__restrict__ ds_dst) {
    constexpr int Q = 2 * H;
    __shared__ float red[4][32][17];
    const int i = blockIdx.x;
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int p0 = rowptr[i], deg = rowptr[i + 1] - p0;
    const int r = lane & 15, kq = lane >> 4;
    const __bf16 *gb = dA + ((size_t)min(r, H - 1) * n_tgt + i) * K + 8 * kq;     // (columns h >= H are never stored: row H - 1 again)
    const int KC = K >> 5;
    constexpr int UB = 8;
    for (int e0 = 0; e0 < deg; e0 += 32) {
        const bool two = e0 + 16 < deg;                              // (uniform) a second block of 16 edges
        const __bf16 *x0 = x + (size_t)col[p0 + min(e0 + r, deg - 1)] * ldx + 8 * kq;
        const __bf16 *x1 = x + (size_t)col[p0 + min(e0 + 16 + r, deg - 1)] * ldx + 8 * kq;
        f4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
        for (int kc = w; kc < KC; kc += 4 * UB) {
            bf8 b[UB], a0[UB], a1[UB];
#pragma unroll
            for (int u = 0; u < UB; u++) {
                const int kk = (kc + 4 * u < KC) ? kc + 4 * u : w;
                b[u] = *reinterpret_cast<const bf8 *>(gb + 32 * kk);
                a0[u] = *reinterpret_cast<const bf8 *>(x0 + 32 * kk);
                a1[u] = *reinterpret_cast<const bf8 *>(x1 + 32 * kk);      // (one block only: the last edge's row again, not stored)
            }
#pragma unroll
            for (int u = 0; u < UB; u++)
                if (kc + 4 * u < KC) {
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0[u], b[u], acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1[u], b[u], acc1, 0, 0, 0);
                }
        }
        __syncthreads();                                             // the previous 32 edges' readers are done
#pragma unroll
        for (int u = 0; u < 4; u++) { red[w][4 * kq + u][r] = acc0[u]; red[w][16 + 4 * kq + u][r] = acc1[u]; }
        __syncthreads();
        for (int u = t; u < 32 * H; u += NT) {
            const int e = u / H, h = u - e * H;
            if (e0 + e < deg) dz[(size_t)(p0 + e0 + e) * H + h] = red[0][e][h] + red[1][e][h] + red[2][e][h] + red[3][e][h];
        }
    }
    __threadfence_block();
    __syncthreads();                                                 // d alpha of every edge of this seed is in dz
    for (int hd = w; hd < H; hd += 4) {
        float dot = 0.f;
        for (int e = lane; e < deg; e += WAVE) dot += alpha[(size_t)(p0 + e) * H + hd] * dz[(size_t)(p0 + e) * H + hd];
        dot = wave_sum_f(dot);
        const float sd = s[(size_t)i * Q + 2 * hd + 1];
        float dsum = 0.f;
        for (int e = lane; e < deg; e += WAVE) {
            const size_t o = (size_t)(p0 + e) * H + hd;
            const float raw = s[(size_t)col[p0 + e] * Q + 2 * hd] + sd;
            const float d = alpha[o] * (dz[o] - dot) * (raw > 0.f ? 1.f : ATT_SLOPE);
            dz[o] = d;
            dsum += d;
        }
        dsum = wave_sum_f(dsum);
        if (lane == 0) ds_dst[(size_t)i * H + hd] = dsum;
    }
}

// ------------------------------------------------------------------------------------------------------------------
// Source rows: dx_j[k] = sum_{e: j -> i} sum_h alpha[e][h] dA[h][i][k] + sum_h ( ds_src[j][h] w[2h][k] + ds_dst[j][h] w[2h+1][k] ),
// ds_src[j][h] = sum_{e: j -> i} dz[e][h] (also stored), ds_dst = 0 for j >= n_tgt; rows n <= j < rows_out are written as zeros.
// ------------------------------------------------------------------------------------------------------------------
template <typename T, int H>
__global__ __launch_bounds__(NT) void k_tail_src_bwd(const T *__restrict__ dA, const float *__restrict__ alpha, const float *__restrict__ dz,
                                                     const float *__restrict__ ds_dst, const float *__restrict__ wv,
                                                     const int *__restrict__ rowptr_t, const int *__restrict__ col_t,
                                                     const int *__restrict__ eid_t, int n, int n_tgt, int rows_out, int K,
                                                     T *__restrict__ dx, int lddx, float *__restrict__ ds_src,
                                                     const T *__restrict__ msk, int ldm, float slope) {
    // msk (optional, the layer's INPUT rows x [rows_out x ldm]): dx is multiplied by `slope` where x <= 0 -- x is the activated
    // output of the layer below, so this is that layer's LeakyReLU' applied to its incoming gradient as it is produced
    constexpr int Q = 2 * H, ROWS = 8, MAXE = 128;
    // the block's edge records are fetched ONCE, together, into LDS (row pointers -> edge ids / targets -> alpha, d logit:
    // three dependent loads for the whole block instead of three per row); the row loop then only gathers dA rows
    __shared__ int rp[ROWS + 1];
    __shared__ int tg[MAXE];
    __shared__ float al[MAXE][H], dl[MAXE][H];
    const int t = threadIdx.x;
    const int k = t * 8;
    const bool live = k < K;
    const int j0 = (int)blockIdx.x * ROWS, j1 = min(rows_out, j0 + ROWS);
    // Everything a row's completion needs -- the targets' logit gradients of the block's rows and this thread's pieces of the
    // mask rows -- is requested HERE, unconditionally from clamped positions, together with the row pointers and the logit
    // vectors.  (Loaded where they are used, inside flush(), they were a predicated load each: a branch hipcc drains the load
    // queue at, ~5 dependent round trips per row, 8 rows per workgroup.)
    __shared__ float dsd_s[ROWS][H];
    const int rp_t = rowptr_t[min(j0 + min(t, ROWS), n)];
    const int kk0 = live ? k : 0;
    float w[Q][8];
#pragma unroll
    for (int q = 0; q < Q; q++) load8<float>(wv + (size_t)q * K + kk0, w[q]);
    float dsd_t;
    {
        const int rr = min(t / H, ROWS - 1), j = j0 + rr;
        dsd_t = ds_dst[(size_t)min(j, n_tgt - 1) * H + (t % H)];
        if (!(j < n_tgt)) dsd_t = 0.f;
    }
    const T *mp = msk != nullptr ? msk : dx;                         // (no mask: any readable rows, the values are not used)
    const int mld = msk != nullptr ? ldm : lddx;
    Raw8<T> mraw[ROWS];
#pragma unroll
    for (int r = 0; r < ROWS; r++) mraw[r] = load8_raw<T>(mp + (size_t)min(j0 + r, rows_out - 1) * mld + kk0);
    if (!live) {
#pragma unroll
        for (int q = 0; q < Q; q++)
#pragma unroll
            for (int e = 0; e < 8; e++) w[q][e] = 0.f;
    }
    if (t <= ROWS) rp[t] = rp_t;
    if (t < ROWS * H) dsd_s[t / H][t % H] = dsd_t;
    // (which row is being completed is a run-time value: the pieces wait in LDS, every thread its own column -- a register
    // array indexed at run time would live in scratch memory)
    __shared__ Raw8<T> msk_s[ROWS][NT];
#pragma unroll
    for (int r = 0; r < ROWS; r++) msk_s[r][t] = mraw[r];
    __syncthreads();
    const int base = rp[0], nE = rp[ROWS] - base;
    const bool staged = nE <= MAXE;
    if (staged) {
        for (int u = t; u < nE; u += NT) {
            const int e = eid_t[base + u];
            tg[u] = col_t[base + u];
#pragma unroll
            for (int h = 0; h < H; h++) { al[u][h] = alpha[(size_t)e * H + h]; dl[u][h] = dz[(size_t)e * H + h]; }
        }
        __syncthreads();
    }
    if (staged) {
        // ONE flat loop over the block's edges (they are sorted by source row), four edges' dA rows in flight at a time; a row
        // is finished (logit terms added, stored) when the edge pointer passes its end -- rows without edges included.  The
        // per-row loop below pays one memory latency per edge group of every row in turn (8 rows x ~2 groups); this pays one
        // per four edges of the whole block.
        float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        float dss[H];
#pragma unroll
        for (int h = 0; h < H; h++) dss[h] = 0.f;
        int r = 0;                                                    // current row of the block
        auto flush = [&](int rr) {
            const int j = j0 + rr;
            if (j < n) {
#pragma unroll
                for (int h = 0; h < H; h++) {
                    const float dsd = dsd_s[rr][h];
#pragma unroll
                    for (int q = 0; q < 8; q++) acc[q] = fmaf(dss[h], w[2 * h][q], fmaf(dsd, w[2 * h + 1][q], acc[q]));
                    if (t == h) ds_src[(size_t)j * H + h] = dss[h];
                }
            }
            if (j < rows_out && live) {
                if (msk != nullptr) {
                    float mv[8];
                    unpack8(msk_s[rr][t], mv);
#pragma unroll
                    for (int q = 0; q < 8; q++) if (!(mv[q] > 0.f)) acc[q] *= slope;
                }
                store8<T>(dx + (size_t)j * lddx + k, acc);
            }
#pragma unroll
            for (int q = 0; q < 8; q++) acc[q] = 0.f;
#pragma unroll
            for (int h = 0; h < H; h++) dss[h] = 0.f;
        };
        for (int u0 = 0; u0 < nE; u0 += 4) {
            Raw8<T> gv[4][H];
            const int kk = live ? k : 0;                              // (threads past K read column 0 again and store nothing)
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const int u = min(u0 + e, nE - 1);                    // (past the end: the last edge again, not accumulated)
                const int i = tg[u];
#pragma unroll
                for (int h = 0; h < H; h++) gv[e][h] = load8_raw<T>(dA + ((size_t)h * n_tgt + i) * K + kk);
            }
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const int u = u0 + e;
                if (u < nE) {                                         // (block-uniform)
                    while (base + u >= rp[r + 1]) { flush(r); r++; }
#pragma unroll
                    for (int h = 0; h < H; h++) {
                        const float a = al[u][h];
                        dss[h] += dl[u][h];
                        float g8[8];
                        unpack8(gv[e][h], g8);
#pragma unroll
                        for (int q = 0; q < 8; q++) acc[q] = fmaf(a, g8[q], acc[q]);
                    }
                }
            }
        }
        for (; r < j1 - j0; r++) flush(r);
        return;
    }
    for (int j = j0; j < j1; j++) {
        float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (j < n) {
            float dss[H], dsd[H];
#pragma unroll
            for (int h = 0; h < H; h++) { dss[h] = 0.f; dsd[h] = j < n_tgt ? ds_dst[(size_t)j * H + h] : 0.f; }
            const int q0 = rp[j - j0], q1 = rp[j - j0 + 1];
            for (int p = q0; p < q1; p++) {
                const int e = eid_t[p], i = col_t[p];
                float a[H];
#pragma unroll
                for (int h = 0; h < H; h++) { a[h] = alpha[(size_t)e * H + h]; dss[h] += dz[(size_t)e * H + h]; }
                if (live) {
                    float gv[H][8];
#pragma unroll
                    for (int h = 0; h < H; h++) load8<T>(dA + ((size_t)h * n_tgt + i) * K + k, gv[h]);
#pragma unroll
                    for (int h = 0; h < H; h++)
#pragma unroll
                        for (int q = 0; q < 8; q++) acc[q] = fmaf(a[h], gv[h][q], acc[q]);
                }
            }
#pragma unroll
            for (int h = 0; h < H; h++)
#pragma unroll
                for (int q = 0; q < 8; q++) acc[q] = fmaf(dss[h], w[2 * h][q], fmaf(dsd[h], w[2 * h + 1][q], acc[q]));
#pragma unroll
            for (int h = 0; h < H; h++)
                if (t == h) ds_src[(size_t)j * H + h] = dss[h];
        }
        if (live) {
            if (msk != nullptr) {
                float mv[8];
                load8<T>(msk + (size_t)j * ldm + k, mv);
#pragma unroll
                for (int q = 0; q < 8; q++) if (!(mv[q] > 0.f)) acc[q] *= slope;
            }
            store8<T>(dx + (size_t)j * lddx + k, acc);
        }
    }
}

// dw partials: part[wg][q][k] = sum over the workgroup's rows of dS[j][q] x[j][k]   (dS[j][2h] = ds_src, [2h+1] = ds_dst or 0)
template <typename T, int H>
__global__ __launch_bounds__(NT) void k_tail_dwvec_part(const T *__restrict__ x, int ldx, const float *__restrict__ ds_src,
                                                        const float *__restrict__ ds_dst, int n, int n_tgt, int K, int rows_per_wg,
                                                        float *__restrict__ part) {
    constexpr int Q = 2 * H;
    const int t = threadIdx.x;
    const int k = t * 8;
    if (k >= K) return;
    float acc[Q][8];
#pragma unroll
    for (int q = 0; q < Q; q++)
#pragma unroll
        for (int e = 0; e < 8; e++) acc[q][e] = 0.f;
    const int j0 = (int)blockIdx.x * rows_per_wg, j1 = min(n, j0 + rows_per_wg);
    int j = j0;
    for (; j + 4 <= j1; j += 4) {
        float xv[4][8];
#pragma unroll
        for (int u = 0; u < 4; u++) load8<T>(x + (size_t)(j + u) * ldx + k, xv[u]);
#pragma unroll
        for (int u = 0; u < 4; u++)
#pragma unroll
            for (int h = 0; h < H; h++) {
                const float a = ds_src[(size_t)(j + u) * H + h], d = (j + u) < n_tgt ? ds_dst[(size_t)(j + u) * H + h] : 0.f;
#pragma unroll
                for (int e = 0; e < 8; e++) { acc[2 * h][e] = fmaf(a, xv[u][e], acc[2 * h][e]); acc[2 * h + 1][e] = fmaf(d, xv[u][e], acc[2 * h + 1][e]); }
            }
    }
    for (; j < j1; j++) {
        float xv[8];
        load8<T>(x + (size_t)j * ldx + k, xv);
#pragma unroll
        for (int h = 0; h < H; h++) {
            const float a = ds_src[(size_t)j * H + h], d = j < n_tgt ? ds_dst[(size_t)j * H + h] : 0.f;
#pragma unroll
            for (int e = 0; e < 8; e++) { acc[2 * h][e] = fmaf(a, xv[e], acc[2 * h][e]); acc[2 * h + 1][e] = fmaf(d, xv[e], acc[2 * h + 1][e]); }
        }
    }
    float *p = part + (size_t)blockIdx.x * Q * K + k;
#pragma unroll
    for (int q = 0; q < Q; q++) store8<float>(p + (size_t)q * K, acc[q]);
}

// Row r = h C + c of W: dW[r][k] (+)= att_src[r] dw[2h][k] + att_dst[r] dw[2h+1][k];  datt_src[r] = W[r] . dw[2h], datt_dst[r] = W[r] . dw[2h+1]
__global__ __launch_bounds__(NT) void k_tail_wvec_bwd(const float *__restrict__ W, int ldw, const float *__restrict__ a_src,
                                                      const float *__restrict__ a_dst, const float *__restrict__ dwv, int C, int K,
                                                      float *__restrict__ dW, int lddw, int accumulate, float *__restrict__ datt_src,
                                                      float *__restrict__ datt_dst) {
    __shared__ float red[2][4];
    const int r = blockIdx.x, h = r / C;
    const int t = threadIdx.x, lane = t & 63, wid = t >> 6;
    const int k = t * 8;
    float ps = 0.f, pd = 0.f;
    if (k < K) {
        float ws[8], wd[8], wr[8], o[8];
        load8<float>(dwv + (size_t)(2 * h) * K + k, ws);
        load8<float>(dwv + (size_t)(2 * h + 1) * K + k, wd);
        load8<float>(W + (size_t)r * ldw + k, wr);
        const float a = a_src[r], d = a_dst[r];
        if (accumulate) load8<float>(dW + (size_t)r * lddw + k, o);
        else
#pragma unroll
            for (int e = 0; e < 8; e++) o[e] = 0.f;
#pragma unroll
        for (int e = 0; e < 8; e++) {
            o[e] = fmaf(a, ws[e], fmaf(d, wd[e], o[e]));
            ps = fmaf(wr[e], ws[e], ps);
            pd = fmaf(wr[e], wd[e], pd);
        }
        store8<float>(dW + (size_t)r * lddw + k, o);
    }
    ps = wave_sum_f(ps); pd = wave_sum_f(pd);
    if (lane == 0) { red[0][wid] = ps; red[1][wid] = pd; }
    __syncthreads();
    if (t == 0) {
        datt_src[r] = red[0][0] + red[0][1] + red[0][2] + red[0][3];
        datt_dst[r] = red[1][0] + red[1][1] + red[1][2] + red[1][3];
    }
}

constexpr int DT_F32 = 0, DT_BF16 = 1;

inline bool shape_ok(int H, int K) { return (H == 1 || H == 2 || H == 4 || H == 8) && K > 0 && K <= 8 * NT && K % 8 == 0; }
inline int rc_last() { return hipGetLastError() == hipSuccess ? 0 : -5; }

#define TAIL_DISPATCH_H(H_, ...)                                 \
    switch (H_) {                                                \
        case 1: { constexpr int HH = 1; __VA_ARGS__; } break;    \
        case 2: { constexpr int HH = 2; __VA_ARGS__; } break;    \
        case 4: { constexpr int HH = 4; __VA_ARGS__; } break;    \
        default: { constexpr int HH = 8; __VA_ARGS__; } break;   \
    }

}  // namespace

extern "C" {

int spadot_gat_tail_supported(int dtype, int H, int K) { return ((dtype == DT_F32 || dtype == DT_BF16) && shape_ok(H, K)) ? 1 : 0; }

int spadot_gat_tail_wvec(const float *W, int ldw, const float *att_src, const float *att_dst, int H, int C, int K, float *part,
                         int slices, float *wv, void *whi, void *wlo, void *stream) {
    if (!W || !att_src || !att_dst || !part || !wv || !shape_ok(H, K) || C <= 0 || ldw < K || ldw % 4 || slices < 1 || slices > 64 ||
        ((whi == nullptr) != (wlo == nullptr)))
        return -22;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(k_tail_wvec, dim3((unsigned)((K / 4 + NT - 1) / NT), (unsigned)H, (unsigned)slices), dim3(NT), 0, st, W, ldw,
                       att_src, att_dst, H, C, K, slices, part);
    const int total = 2 * H * K;
    hipLaunchKernelGGL(k_tail_wvec_fin, dim3((unsigned)((total + NT - 1) / NT)), dim3(NT), 0, st, part, slices, total, wv, (__bf16 *)whi,
                       (__bf16 *)wlo);
    return rc_last();
}

int spadot_gat_tail_logits(const void *x, int dtype, int ldx, const float *wv, const void *whi, const void *wlo, int n, int H, int K, float *s,
                           void *stream) {
    if (!x || !wv || !s || n <= 0 || !spadot_gat_tail_supported(dtype, H, K) || ldx < K || ldx % 8) return -22;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == DT_BF16 && whi && wlo && K % 32 == 0) {           // matrix cores: 16 rows per workgroup
        const int grid = (n + 15) / 16;
        TAIL_DISPATCH_H(H, hipLaunchKernelGGL((k_tail_logits_mfma<2 * HH>), dim3(grid), dim3(NT), 0, st, (const __bf16 *)x, ldx,
                                              (const __bf16 *)whi, (const __bf16 *)wlo, n, K, s));
        return rc_last();
    }
    const int rows = 8, grid = (n + rows - 1) / rows;
#define LOGITS(T_) TAIL_DISPATCH_H(H, hipLaunchKernelGGL((k_tail_logits<T_, 2 * HH, (HH >= 8 ? 1 : (HH == 4 ? 2 : (HH == 2 ? 4 : 8)))>), dim3(grid), \
                                                          dim3(NT), 0, st, (const T_ *)x, ldx, wv, n, K, rows, s))
    if (dtype == DT_BF16) { LOGITS(__bf16); } else { LOGITS(float); }
#undef LOGITS
    return rc_last();
}

int spadot_gat_tail_aggregate(const void *x, int dtype, int ldx, const float *s, const int *rowptr, const int *col, int n_tgt, int H,
                              int K, void *A, float *alpha, void *stream) {
    if (!x || !s || !rowptr || !col || !A || !alpha || n_tgt <= 0 || !spadot_gat_tail_supported(dtype, H, K) || ldx < K || ldx % 8) return -22;
    hipStream_t st = (hipStream_t)stream;
#define AGG(T_) TAIL_DISPATCH_H(H, hipLaunchKernelGGL((k_tail_aggregate<T_, HH>), dim3(n_tgt), dim3(NT), 0, st, (const T_ *)x, ldx, s, \
                                                        rowptr, col, n_tgt, K, (T_ *)A, alpha))
    if (dtype == DT_BF16) { AGG(__bf16); } else { AGG(float); }
#undef AGG
    return rc_last();
}

int spadot_gat_tail_headmean(const void *O, int dtype, const float *bias, int n_tgt, int H, int C, void *out, void *stream) {
    if (!O || !bias || !out || n_tgt <= 0 || H <= 0 || C <= 0 || C % 8 || (dtype != DT_F32 && dtype != DT_BF16)) return -22;
    hipStream_t st = (hipStream_t)stream;
    const size_t tot = (size_t)n_tgt * C;
    const unsigned grid = (unsigned)((tot / 8 + NT - 1) / NT);
    if (dtype == DT_BF16) hipLaunchKernelGGL(k_tail_headmean<__bf16>, dim3(grid), dim3(NT), 0, st, (const __bf16 *)O, bias, n_tgt, H, C, (__bf16 *)out);
    else hipLaunchKernelGGL(k_tail_headmean<float>, dim3(grid), dim3(NT), 0, st, (const float *)O, bias, n_tgt, H, C, (float *)out);
    return rc_last();
}

int spadot_gat_tail_colsum_rows(const void *g, int dtype, int rows, int C, float *out, void *stream) {
    if (!g || !out || rows <= 0 || C <= 0 || (dtype != DT_F32 && dtype != DT_BF16)) return -22;
    hipStream_t st = (hipStream_t)stream;
    const unsigned grid = (unsigned)((C + 15) / 16);
    if (dtype == DT_BF16) hipLaunchKernelGGL(k_tail_colsum_rows<__bf16>, dim3(grid), dim3(NT), 0, st, (const __bf16 *)g, rows, C, out);
    else hipLaunchKernelGGL(k_tail_colsum_rows<float>, dim3(grid), dim3(NT), 0, st, (const float *)g, rows, C, out);
    return rc_last();
}

int spadot_gat_tail_scale_colsum(const void *g, int dtype, int rows, int C, double scale, void *gs, float *out, void *stream) {
    if (!g || !gs || !out || rows <= 0 || C <= 0 || (dtype != DT_F32 && dtype != DT_BF16)) return -22;
    hipStream_t st = (hipStream_t)stream;
    const unsigned grid = (unsigned)((C + 15) / 16);
    if (dtype == DT_BF16) hipLaunchKernelGGL(k_tail_scale_colsum<__bf16>, dim3(grid), dim3(NT), 0, st, (const __bf16 *)g, rows, C, (float)scale, (__bf16 *)gs, out);
    else hipLaunchKernelGGL(k_tail_scale_colsum<float>, dim3(grid), dim3(NT), 0, st, (const float *)g, rows, C, (float)scale, (float *)gs, out);
    return rc_last();
}

int spadot_gat_tail_edge_backward(const void *x, int dtype, int ldx, const void *dA, const float *s, const float *alpha, const int *rowptr,
                                  const int *col, int n_tgt, int H, int K, float *dz, float *ds_dst, void *stream) {
    if (!x || !dA || !s || !alpha || !rowptr || !col || !dz || !ds_dst || n_tgt <= 0 || !spadot_gat_tail_supported(dtype, H, K) || ldx < K ||
        ldx % 8)
        return -22;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == DT_BF16 && K % 32 == 0) {
        TAIL_DISPATCH_H(H, hipLaunchKernelGGL((k_tail_edge_bwd_mfma<HH>), dim3(n_tgt), dim3(NT), 0, st, (const __bf16 *)x, ldx, (const __bf16 *)dA,
                                              s, alpha, rowptr, col, n_tgt, K, dz, ds_dst));
        return rc_last();
    }
#define EBWD(T_) TAIL_DISPATCH_H(H, hipLaunchKernelGGL((k_tail_edge_bwd<T_, HH, (HH >= 8 ? 2 : (HH == 4 ? 4 : 8))>), dim3(n_tgt), dim3(NT), 0, st, \
                                                         (const T_ *)x, ldx, (const T_ *)dA, s, alpha, rowptr, col, n_tgt, K, dz, ds_dst))
    if (dtype == DT_BF16) { EBWD(__bf16); } else { EBWD(float); }
#undef EBWD
    return rc_last();
}

int spadot_gat_tail_source_backward(const void *dA, int dtype, const float *alpha, const float *dz, const float *ds_dst, const float *wv,
                                    const int *rowptr_t, const int *col_t, const int *eid_t, int n, int n_tgt, int rows_out, int H, int K,
                                    void *dx, int lddx, float *ds_src, const void *act_out, int ldm, double slope, void *stream) {
    if (act_out && (ldm < K || ldm % 8 || ((uintptr_t)act_out & 15))) return -22;
    if (!dA || !alpha || !dz || !ds_dst || !wv || !rowptr_t || !col_t || !eid_t || !dx || !ds_src || n <= 0 || n_tgt <= 0 || n_tgt > n ||
        rows_out < n || !spadot_gat_tail_supported(dtype, H, K) || lddx < K || lddx % 8)
        return -22;
    hipStream_t st = (hipStream_t)stream;
    const int rows = 8, grid = (rows_out + rows - 1) / rows;
#define SBWD(T_) TAIL_DISPATCH_H(H, hipLaunchKernelGGL((k_tail_src_bwd<T_, HH>), dim3(grid), dim3(NT), 0, st, (const T_ *)dA, alpha, dz, ds_dst, \
                                                         wv, rowptr_t, col_t, eid_t, n, n_tgt, rows_out, K, (T_ *)dx, lddx, ds_src, (const T_ *)act_out, ldm, (float)slope))
    if (dtype == DT_BF16) { SBWD(__bf16); } else { SBWD(float); }
#undef SBWD
    return rc_last();
}

int spadot_gat_tail_dwvec_rows(int n) { return n <= 0 ? 0 : (n + 31) / 32; }

int spadot_gat_tail_dwvec(const void *x, int dtype, int ldx, const float *ds_src, const float *ds_dst, int n, int n_tgt, int H, int K,
                          float *part, void *stream) {
    if (!x || !ds_src || !ds_dst || !part || n <= 0 || n_tgt <= 0 || !spadot_gat_tail_supported(dtype, H, K) || ldx < K || ldx % 8) return -22;
    hipStream_t st = (hipStream_t)stream;
    const int rows = 32, grid = spadot_gat_tail_dwvec_rows(n);
#define DWV(T_) TAIL_DISPATCH_H(H, hipLaunchKernelGGL((k_tail_dwvec_part<T_, HH>), dim3(grid), dim3(NT), 0, st, (const T_ *)x, ldx, ds_src, ds_dst, \
                                                        n, n_tgt, K, rows, part))
    if (dtype == DT_BF16) { DWV(__bf16); } else { DWV(float); }
#undef DWV
    return rc_last();
}

int spadot_gat_tail_wvec_backward(const float *W, int ldw, const float *att_src, const float *att_dst, const float *dwv, int H, int C, int K,
                                  float *dW, int lddw, int accumulate, float *datt_src, float *datt_dst, void *stream) {
    if (!W || !att_src || !att_dst || !dwv || !dW || !datt_src || !datt_dst || !shape_ok(H, K) || C <= 0 || ldw < K || lddw < K || ldw % 4 ||
        lddw % 4)
        return -22;
    hipLaunchKernelGGL(k_tail_wvec_bwd, dim3((unsigned)(H * C)), dim3(NT), 0, (hipStream_t)stream, W, ldw, att_src, att_dst, dwv, C, K, dW,
                       lddw, accumulate, datt_src, datt_dst);
    return rc_last();
}

}  // extern "C"
