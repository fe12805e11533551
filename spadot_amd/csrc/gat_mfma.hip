// gat_mfma.hip -- the GAT edge phase on the matrix cores (gfx950, wave64), bf16 rows.
//
// Reference arithmetic: torch_geometric GATConv's message passing as used by
// /root/reference/SpaDOT/model/encoder.py:41-58: out[i] = sum_j alpha[i, j] h[j] over the incoming
// edges of i, and its backward.  ABI: include/spadot_model.h (spadot_gat_alpha / _aggregate / _edge_dot /
// _softmax_backward).
//
// Why not one row gather per edge (model_kernels.hip: k_gat_fwd & co, still the path for fp32 rows and odd shapes):
// per layer that moves E x H C x 2 B = 1.27 GB through L2 for 82 MB of algorithmic bytes.  Spots that are close in
// space share most of their neighbours: 32 consecutive targets in Z-order touch ~130 DISTINCT sources, not 32 x 31.
// So the work is cut into blocks of 32 rows (spadot_amd/graph.py: BlockPlan) and every block is a small dense product
//     D [32 x C] = A [32 x S] . X [S x C],   A = the block's attention weights (zero where there is no edge),
//                                             X = the block's distinct source rows, fetched ONCE,
// on v_mfma_f32_32x32x16_bf16.  A is split into bf16 hi + lo parts (two MFMAs), so the weights keep 16 significant bits
// and the result matches the fp32-weight FMA chain of the per-edge kernels; accumulation is fp32 in both.
//   k_gat_alpha         softmax over incoming edges, one thread per (target, head)           -> alpha [E, H]
//   k_gat_agg           the product above, one workgroup per (block, head): forward (bias + LeakyReLU epilogue) and,
//                       on the transposed plan, the source-side backward dh = alpha^T g_pre (+ logit-gradient terms)
//   k_gat_edot          target-side backward: g_pre = g_out * act', dA = g_pre . X^T (K = channels) -> raw d(alpha)
//   k_gat_softmax_bwd   softmax + LeakyReLU(0.2) backward per (target, head)                  -> dz, ds_dst
// No atomics: every output element has one writer; the summation order inside a tile is fixed by the plan.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>

#include "../../include/spadot_model.h"

namespace {

constexpr float ATT_SLOPE = 0.2f;    // GATConv negative_slope
constexpr float ACT_SLOPE = 0.01f;   // F.leaky_relu default (encoder.py:56-57)
constexpr int ROWS = 32;             // rows per block (graph.py: PLAN_ROWS) = MFMA N (targets on the lanes)
constexpr int KSTEP = 16;            // sources per chunk = MFMA K
constexpr int NT = 256;              // threads per workgroup (4 waves)
constexpr int MAX_COLS = 2048;       // longest column list a block may have (ids staged in LDS)

typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef short s4 __attribute__((ext_vector_type(4)));
typedef short s8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));

__device__ __forceinline__ float leaky(float z, float slope) { return z > 0.f ? z : slope * z; }
__device__ __forceinline__ unsigned short bf16_bits(float x) { return __builtin_bit_cast(unsigned short, (__bf16)x); }
__device__ __forceinline__ float bf16_to_f(unsigned short b) { return __uint_as_float((unsigned)b << 16); }
__device__ __forceinline__ unsigned pack2(float lo, float hi) { return (unsigned)bf16_bits(lo) | ((unsigned)bf16_bits(hi) << 16); }

// Workgroups are dealt round-robin over the 8 XCDs (b and b + 8 share one, each XCD has its own L2): give every XCD a
// contiguous range of work items so that neighbouring blocks -- whose column lists overlap -- share an L2.  Speed only.
__device__ __forceinline__ int xcd_item(int total) {
    const int chunk = (total + 7) >> 3;
    return (int)(blockIdx.x & 7) * chunk + (int)(blockIdx.x >> 3);
}

// ------------------------------------------------------------------------------------------------------------------
// alpha[e, hd] = softmax over the incoming edges e of target i of LeakyReLU_0.2(s_src[j] + s_dst[i])
// (exp(e - max) / (sum + 1e-16): SURVEY App. A).  One thread per (target, head); neighbouring threads are the heads of
// one target, so the index loads broadcast and the logit gathers are 16-byte runs.
// ------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_gat_alpha(const float *__restrict__ s_src, const float *__restrict__ s_dst,
                                                   const int *__restrict__ rowptr, const int *__restrict__ col, int n_tgt,
                                                   int H, float *__restrict__ alpha) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= n_tgt * H) return;
    const int i = idx / H, hd = idx - i * H;
    const int p0 = rowptr[i], p1 = rowptr[i + 1];
    const float sd = s_dst[(size_t)i * H + hd];
    float m = -INFINITY;
#pragma unroll 4
    for (int p = p0; p < p1; p++) m = fmaxf(m, leaky(s_src[(size_t)col[p] * H + hd] + sd, ATT_SLOPE));
    float ssum = 0.f;
#pragma unroll 4
    for (int p = p0; p < p1; p++) ssum += __expf(leaky(s_src[(size_t)col[p] * H + hd] + sd, ATT_SLOPE) - m);
    const float inv = 1.f / (ssum + 1e-16f);
#pragma unroll 4
    for (int p = p0; p < p1; p++)
        alpha[(size_t)p * H + hd] = __expf(leaky(s_src[(size_t)col[p] * H + hd] + sd, ATT_SLOPE) - m) * inv;
}

// dz[e] holds the raw d(alpha[e]) on entry; on exit dz[e] = d(logit[e]) = alpha (d(alpha) - sum_k alpha_k d(alpha_k)) *
// LeakyReLU'(z), and ds_dst[i] = sum_e dz[e].
__global__ __launch_bounds__(256) void k_gat_softmax_bwd(const float *__restrict__ alpha, const float *__restrict__ s_src,
                                                         const float *__restrict__ s_dst, const int *__restrict__ rowptr,
                                                         const int *__restrict__ col, int n_tgt, int H,
                                                         float *__restrict__ dz, float *__restrict__ ds_dst) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= n_tgt * H) return;
    const int i = idx / H, hd = idx - i * H;
    const int p0 = rowptr[i], p1 = rowptr[i + 1];
    const float sd = s_dst[(size_t)i * H + hd];
    float dsum = 0.f;
#pragma unroll 4
    for (int p = p0; p < p1; p++) dsum += alpha[(size_t)p * H + hd] * dz[(size_t)p * H + hd];
    float dsd = 0.f;
#pragma unroll 4
    for (int p = p0; p < p1; p++) {
        const size_t e = (size_t)p * H + hd;
        const float z = s_src[(size_t)col[p] * H + hd] + sd;
        const float d = alpha[e] * (dz[e] - dsum) * (z > 0.f ? 1.f : ATT_SLOPE);
        dz[e] = d;
        dsd += d;
    }
    ds_dst[(size_t)i * H + hd] = dsd;
}

// ------------------------------------------------------------------------------------------------------------------
// k_gat_agg: D^T [C x 32] = X^T [C x S] . A^T [S x 32] for one (block, head).
//
// LDS: a ring of 3 chunks of 16 source rows (row stride 2 C + 64 B: the transposed reads below are bank-conflict
// free), 3 attention sub-tiles [32 rows x 16 sources] hi + lo (row stride 48 B), the block's column ids, 32 row scalars.
// Pipeline (one barrier per chunk): chunk c's rows, cell indices and weights are requested 4 iterations before they are
// used and wait in registers (3 rotating sets); iteration it writes chunk it + 1 into LDS, requests chunk it + 4,
// barrier, multiplies chunk it.  MFMA operands: A operand = X^T fragment (lane: channel l & 31, sources 8 (l >> 5) ..+7)
// by two ds_read_b64_tr_b16 of the row-major image; B operand = the weights (lane: row l & 31, same 8 sources).
// The result has the row (target) on the lane and 16 channels in registers: four runs of four consecutive channels.
// Epilogue through LDS (the ring is free by then) so that every global store is a whole 16-byte piece of a row.
// MODE 0 (forward):         out = leaky?(D + bias)
// MODE 1 (source backward): out = D + ds_src att_src + ds_dst att_dst, ds_src[row] = sum of dz over the row's cells
// ------------------------------------------------------------------------------------------------------------------
template <int NTW> struct AggCfg {
    static constexpr int C = 128 * NTW;                 // channels per head
    static constexpr int PPR = C / 8;                   // 16-byte pieces per row
    static constexpr int ROWB = 2 * C + 64;             // staged row stride (bytes), = 64 mod 256
    static constexpr int CHUNKB = KSTEP * ROWB;
    static constexpr int OUTB = 2 * C + 16;             // epilogue image row stride (bytes)
    static constexpr int ATILEB = ROWS * 48;            // one hi (or lo) sub-tile
    static constexpr int RING = 3;
    static constexpr size_t lds_bytes() {
        const size_t ring = (size_t)RING * CHUNKB, outimg = (size_t)ROWS * OUTB;
        return (ring > outimg ? ring : outimg) + (size_t)RING * 2 * ATILEB + MAX_COLS * 4 + ROWS * 8;
    }
};

template <int NTW, int MODE>
__global__ __launch_bounds__(NT, 2) void k_gat_agg(
    const __bf16 *__restrict__ X, const float *__restrict__ alpha, const int *__restrict__ prow,
    const int *__restrict__ sptr, const int *__restrict__ pcol, const int *__restrict__ cell, int nb, int H,
    const float *__restrict__ vec_a,      // MODE 0: bias [H*C]; MODE 1: att_src [H*C]
    const float *__restrict__ vec_b,      // MODE 1: att_dst [H*C]
    int act, const float *__restrict__ dz, const float *__restrict__ ds_dst, float *__restrict__ ds_src,
    __bf16 *__restrict__ out) {
    using G = AggCfg<NTW>;
    constexpr int C = G::C, PPR = G::PPR, ROWB = G::ROWB;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char *ring = smem;
    constexpr size_t ring_bytes = ((size_t)G::RING * G::CHUNKB > (size_t)ROWS * G::OUTB) ? (size_t)G::RING * G::CHUNKB : (size_t)ROWS * G::OUTB;
    unsigned char *atile = smem + ring_bytes;                             // [RING][hi | lo][32 x 48 B]
    int *sids = reinterpret_cast<int *>(atile + G::RING * 2 * G::ATILEB); // [MAX_COLS]
    int *rid = sids + MAX_COLS;                                           // [32] row node ids
    float *rsc = reinterpret_cast<float *>(rid + ROWS);                   // [32] row scalars (MODE 1: ds_src)

    const int item = xcd_item(nb * H);
    if (item >= nb * H) return;
    const int b = item / H, hd = item - b * H;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int s0 = sptr[b], ncol = sptr[b + 1] - s0, nch = ncol / KSTEP;
    const int q0 = s0 / KSTEP;                                            // first global chunk of this block

    for (int k = tid; k < ncol; k += NT) sids[k] = pcol[s0 + k];
    if (tid < ROWS) rid[tid] = prow[b * ROWS + tid];
    __syncthreads();

    // register sets of the chunks in flight
    uint4 rows_r[3][NTW];
    int2 cell_r[3];
    float a_r[3][2], d_r[3][2];
    float dzsum = 0.f;
    const size_t hoff = (size_t)hd * C;
    const size_t HC = (size_t)H * C;

    auto request_cells = [&](int c, int u) {
        cell_r[u] = reinterpret_cast<const int2 *>(cell + (size_t)(q0 + c) * (KSTEP * ROWS))[tid];
    };
    auto request_rows = [&](int c, int u) {
        const int2 ce = cell_r[u];
        a_r[u][0] = ce.x >= 0 ? alpha[(size_t)ce.x * H + hd] : 0.f;
        a_r[u][1] = ce.y >= 0 ? alpha[(size_t)ce.y * H + hd] : 0.f;
        if (MODE == 1) {      // (summed when the chunk is staged: an add here would wait for the load at once)
            d_r[u][0] = ce.x >= 0 ? dz[(size_t)ce.x * H + hd] : 0.f;
            d_r[u][1] = ce.y >= 0 ? dz[(size_t)ce.y * H + hd] : 0.f;
        }
#pragma unroll
        for (int m = 0; m < NTW; m++) {
            const int pi = m * NT + tid, r = pi / PPR, pc = pi - r * PPR;
            const int sid = sids[c * KSTEP + r];
            rows_r[u][m] = *reinterpret_cast<const uint4 *>(X + (size_t)sid * HC + hoff + (size_t)pc * 8);
        }
    };
    auto stage = [&](int c, int u) {            // registers of chunk c -> LDS slot c % 3
        unsigned char *slot = ring + (size_t)(c % G::RING) * G::CHUNKB;
#pragma unroll
        for (int m = 0; m < NTW; m++) {
            const int pi = m * NT + tid, r = pi / PPR, pc = pi - r * PPR;
            *reinterpret_cast<uint4 *>(slot + (size_t)r * ROWB + (size_t)pc * 16) = rows_r[u][m];
        }
        if (MODE == 1) dzsum += d_r[u][0] + d_r[u][1];
        const float a0 = a_r[u][0], a1 = a_r[u][1];
        const float h0 = bf16_to_f(bf16_bits(a0)), h1 = bf16_to_f(bf16_bits(a1));
        unsigned char *at = atile + (size_t)(c % G::RING) * 2 * G::ATILEB;
        const int r = tid >> 3, kk = (tid & 7) * 2;
        *reinterpret_cast<unsigned *>(at + r * 48 + kk * 2) = pack2(h0, h1);
        *reinterpret_cast<unsigned *>(at + G::ATILEB + r * 48 + kk * 2) = pack2(a0 - h0, a1 - h1);
    };

    f16v acc[NTW];
#pragma unroll
    for (int t = 0; t < NTW; t++)
#pragma unroll
        for (int i = 0; i < 16; i++) acc[t][i] = 0.f;

    // transposed-read addresses of this lane inside a chunk (T10): 16-lane group -> columns 16 g .. 16 g + 15 of the
    // N-tile, lane 4 q + p of the group supplies row q, columns 4 p .. 4 p + 3; the two reads of a fragment are rows
    // 8 hh .. + 3 and 8 hh + 4 .. + 7
    const int hh = lane >> 5, g16 = (lane >> 4) & 1, qq = (lane & 15) >> 2, pp = lane & 3;
    const int tr_off = (8 * hh + qq) * ROWB + (16 * g16 + 4 * pp) * 2;

    auto multiply = [&](int c) {
        const unsigned char *slot = ring + (size_t)(c % G::RING) * G::CHUNKB;
        const unsigned char *at = atile + (size_t)(c % G::RING) * 2 * G::ATILEB;
        const bf8 a_hi = *reinterpret_cast<const bf8 *>(at + (lane & 31) * 48 + hh * 16);
        const bf8 a_lo = *reinterpret_cast<const bf8 *>(at + G::ATILEB + (lane & 31) * 48 + hh * 16);
#pragma unroll
        for (int t = 0; t < NTW; t++) {
            const int cb = (wave * NTW + t) * 32;
            const unsigned char *base = slot + tr_off + cb * 2;
            const s4 x0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s4 __attribute__((address_space(3))) *)(base));
            const s4 x1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s4 __attribute__((address_space(3))) *)(base + 4 * ROWB));
            const s8 xs = __builtin_shufflevector(x0, x1, 0, 1, 2, 3, 4, 5, 6, 7);
            const bf8 xf = __builtin_bit_cast(bf8, xs);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xf, a_hi, acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xf, a_lo, acc[t], 0, 0, 0);
        }
    };

    // Pipeline.  Every chunk c goes through: request_cells(c) -> request_rows(c) (needs the cells: the weights are
    // gathered through them) -> stage(c) (registers -> LDS) -> multiply(c), in iterations c - 5, c - 4, c - 1, c.
    // Loads return in issue order, so inside an iteration the cells of chunk it + 5 are requested FIRST: waiting for
    // them one iteration later then only waits for loads that are two iterations old.  Register sets rotate mod 3.
    request_cells(0, 0);
    if (1 < nch) request_cells(1, 1);
    if (2 < nch) request_cells(2, 2);
    request_rows(0, 0);
    if (3 < nch) request_cells(3, 0);
    if (1 < nch) request_rows(1, 1);
    if (4 < nch) request_cells(4, 1);
    if (2 < nch) request_rows(2, 2);
    stage(0, 0);
    if (3 < nch) request_rows(3, 0);
#define AGG_ITER(U)                                                                        \
    {                                                                                      \
        const int it = it0 + (U);                                                          \
        if (it < nch) {                                                                    \
            if (it + 5 < nch) request_cells(it + 5, ((U) + 2) % 3);                        \
            if (it + 1 < nch) stage(it + 1, ((U) + 1) % 3);                                \
            if (it + 4 < nch) request_rows(it + 4, ((U) + 1) % 3);                         \
            __syncthreads();                                                               \
            multiply(it);                                                                  \
        }                                                                                  \
    }
    for (int it0 = 0; it0 < nch; it0 += 3) {
        AGG_ITER(0)
        AGG_ITER(1)
        AGG_ITER(2)
    }
#undef AGG_ITER

    // ---- epilogue
    if (MODE == 1) {       // ds_src[row] = sum of dz over the row's cells: 8 threads per row hold partial sums
        float v = dzsum;
        v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64);
        if ((tid & 7) == 0) rsc[tid >> 3] = v;
    }
    __syncthreads();       // every wave is done with the ring; rsc visible
    {
        const int t = lane & 31;
        const int node = rid[t];
        float sa = 0.f, sb = 0.f;
        if (MODE == 1) {
            sa = rsc[t];
            sb = node >= 0 ? ds_dst[(size_t)node * H + hd] : 0.f;
        }
        unsigned char *img = ring;
#pragma unroll
        for (int tt = 0; tt < NTW; tt++) {
            const int cb = (wave * NTW + tt) * 32;
#pragma unroll
            for (int g = 0; g < 4; g++) {
                const int ch = cb + 8 * g + 4 * hh;
                const float4 va = *reinterpret_cast<const float4 *>(vec_a + hoff + ch);
                float v[4] = {acc[tt][4 * g], acc[tt][4 * g + 1], acc[tt][4 * g + 2], acc[tt][4 * g + 3]};
                if (MODE == 0) {
                    v[0] += va.x; v[1] += va.y; v[2] += va.z; v[3] += va.w;
                    if (act) {
#pragma unroll
                        for (int e = 0; e < 4; e++) v[e] = leaky(v[e], ACT_SLOPE);
                    }
                } else {
                    const float4 vb = *reinterpret_cast<const float4 *>(vec_b + hoff + ch);
                    v[0] += sa * va.x + sb * vb.x; v[1] += sa * va.y + sb * vb.y;
                    v[2] += sa * va.z + sb * vb.z; v[3] += sa * va.w + sb * vb.w;
                }
                *reinterpret_cast<uint2 *>(img + (size_t)t * G::OUTB + (size_t)ch * 2) = make_uint2(pack2(v[0], v[1]), pack2(v[2], v[3]));
            }
        }
        if (MODE == 1 && tid < ROWS && rid[tid] >= 0) ds_src[(size_t)rid[tid] * H + hd] = rsc[tid];
    }
    __syncthreads();
#pragma unroll
    for (int m = 0; m < 2 * NTW; m++) {
        const int pi = m * NT + tid, r = pi / PPR, pc = pi - r * PPR;
        const int node = rid[r];
        if (node >= 0)
            *reinterpret_cast<uint4 *>(out + (size_t)node * HC + hoff + (size_t)pc * 8) =
                *reinterpret_cast<const uint4 *>(ring + (size_t)r * G::OUTB + (size_t)pc * 16);
    }
}

// ------------------------------------------------------------------------------------------------------------------
// k_gat_edot: one workgroup per (block, head).
//   g_pre[i] = g_out[i] * (act ? LeakyReLU'(out[i]) : 1)   (stored, and kept in LDS as the block's G tile [32 x C])
//   dA [32 x S] = G . X^T: K = channels, so both operands are plain row reads -- G from LDS (row stride 2 C + 16 B),
//   X straight from global memory: lane (s, hh) walks ITS source row, 32 contiguous bytes per pair of k-steps (the
//   channel order inside a pair is permuted the same way on both operands).  Each wave takes the 32-column tiles
//   wave, wave + 4, ...; the raw d(alpha) values go to dz through the plan's cell map (one writer per edge).
// ------------------------------------------------------------------------------------------------------------------
template <int NTW>
__global__ __launch_bounds__(NT, 4) void k_gat_edot(const __bf16 *__restrict__ g_out, const __bf16 *__restrict__ outp,
                                                    const __bf16 *__restrict__ Xh, const int *__restrict__ prow,
                                                    const int *__restrict__ sptr, const int *__restrict__ pcol,
                                                    const int *__restrict__ cell, int nb, int H, int act,
                                                    __bf16 *__restrict__ g_pre, float *__restrict__ dz) {
    constexpr int C = 128 * NTW, PPR = C / 8, GS = 2 * C + 16;
    __shared__ __attribute__((aligned(16))) unsigned char gt[ROWS * GS];
    __shared__ int rid[ROWS];
    const int item = xcd_item(nb * H);
    if (item >= nb * H) return;
    const int b = item / H, hd = item - b * H;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const size_t hoff = (size_t)hd * C, HC = (size_t)H * C;
    if (tid < ROWS) rid[tid] = prow[b * ROWS + tid];
    __syncthreads();
#pragma unroll
    for (int m = 0; m < 2 * NTW; m++) {
        const int pi = m * NT + tid, r = pi / PPR, pc = pi - r * PPR;
        const int node = rid[r];
        uint4 g = make_uint4(0u, 0u, 0u, 0u);
        if (node >= 0) {
            const size_t off = (size_t)node * HC + hoff + (size_t)pc * 8;
            g = *reinterpret_cast<const uint4 *>(g_out + off);
            if (act) {
                const uint4 o = *reinterpret_cast<const uint4 *>(outp + off);
                unsigned gw[4] = {g.x, g.y, g.z, g.w};
                const unsigned ow[4] = {o.x, o.y, o.z, o.w};
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    float lo = __uint_as_float(gw[e] << 16), hi = __uint_as_float(gw[e] & 0xffff0000u);
                    if (!(__uint_as_float(ow[e] << 16) > 0.f)) lo *= ACT_SLOPE;
                    if (!(__uint_as_float(ow[e] & 0xffff0000u) > 0.f)) hi *= ACT_SLOPE;
                    gw[e] = pack2(lo, hi);
                }
                g = make_uint4(gw[0], gw[1], gw[2], gw[3]);
            }
            *reinterpret_cast<uint4 *>(g_pre + off) = g;
        }
        *reinterpret_cast<uint4 *>(gt + (size_t)r * GS + (size_t)pc * 16) = g;
    }
    __syncthreads();

    const int s0 = sptr[b], ntile = (sptr[b + 1] - s0) / 32;
    const int sl = lane & 31, hh = lane >> 5;
    for (int nt = wave; nt < ntile; nt += 4) {
        const int sid = pcol[s0 + nt * 32 + sl];
        const __bf16 *xrow = Xh + (size_t)sid * HC + hoff + 16 * hh;
        const unsigned char *grow = gt + (size_t)sl * GS + 32 * hh;
        f16v acc;
#pragma unroll
        for (int i = 0; i < 16; i++) acc[i] = 0.f;
#pragma unroll 8
        for (int kp = 0; kp < C / 32; kp++) {
            const bf8 x0 = *reinterpret_cast<const bf8 *>(xrow + kp * 32);
            const bf8 x1 = *reinterpret_cast<const bf8 *>(xrow + kp * 32 + 8);
            const bf8 a0 = *reinterpret_cast<const bf8 *>(grow + kp * 64);
            const bf8 a1 = *reinterpret_cast<const bf8 *>(grow + kp * 64 + 16);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, x0, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, x1, acc, 0, 0, 0);
        }
        // acc[i] = dA[row (i & 3) + 8 (i >> 2) + 4 hh][column slot nt * 32 + sl]
        const int *cq = cell + ((size_t)(s0 / KSTEP) + 2 * nt + (sl >> 4)) * (KSTEP * ROWS) + (sl & 15);
#pragma unroll
        for (int i = 0; i < 16; i++) {
            const int r = (i & 3) + 8 * (i >> 2) + 4 * hh;
            const int e = cq[r * KSTEP];
            if (e >= 0) dz[(size_t)e * H + hd] = acc[i];
        }
    }
}

}  // namespace

extern "C" {

int spadot_gat_alpha(const float *s_src, const float *s_dst, const int *rowptr, const int *col, int n_tgt, int H,
                     float *alpha, void *stream) {
    if (n_tgt <= 0 || H <= 0) return -22;
    const long long tot = (long long)n_tgt * H;
    hipLaunchKernelGGL(k_gat_alpha, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, (hipStream_t)stream, s_src, s_dst, rowptr,
                       col, n_tgt, H, alpha);
    return hipGetLastError() == hipSuccess ? 0 : -5;
}

int spadot_gat_softmax_backward(const float *alpha, const float *s_src, const float *s_dst, const int *rowptr,
                                const int *col, int n_tgt, int H, float *dz, float *ds_dst, void *stream) {
    if (n_tgt <= 0 || H <= 0) return -22;
    const long long tot = (long long)n_tgt * H;
    hipLaunchKernelGGL(k_gat_softmax_bwd, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, (hipStream_t)stream, alpha, s_src,
                       s_dst, rowptr, col, n_tgt, H, dz, ds_dst);
    return hipGetLastError() == hipSuccess ? 0 : -5;
}

int spadot_gat_mfma_supported(int dtype, int C, int max_cols) {
    return dtype == SPADOT_DT_BF16 && (C == 128 || C == 256 || C == 512) && max_cols > 0 && max_cols <= MAX_COLS && max_cols % 32 == 0;
}

#define AGG_LAUNCH(NTW, MODE)                                                                                      \
    do {                                                                                                           \
        auto kern = k_gat_agg<NTW, MODE>;                                                                          \
        static bool attr_set = false;                                                                              \
        const size_t lds = AggCfg<NTW>::lds_bytes();                                                               \
        if (!attr_set) {                                                                                           \
            if (hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return -5; \
            attr_set = true;                                                                                       \
        }                                                                                                          \
        hipLaunchKernelGGL(kern, dim3(grid), dim3(NT), lds, st_, (const __bf16 *)x, alpha, plan_rows, plan_sptr,    \
                           plan_cols, plan_cell, nb, H, vec_a, vec_b, act, dz, ds_dst, ds_src, (__bf16 *)out);     \
    } while (0)

int spadot_gat_aggregate(const void *x, int dtype, const float *alpha, const int *plan_rows, const int *plan_sptr,
                         const int *plan_cols, const int *plan_cell, int nb, int max_cols, int H, int C, int mode,
                         const float *vec_a, const float *vec_b, int act, const float *dz, const float *ds_dst,
                         float *ds_src, void *out, void *stream) {
    if (!spadot_gat_mfma_supported(dtype, C, max_cols) || nb <= 0 || H <= 0 || (mode != 0 && mode != 1)) return -22;
    if (!x || !alpha || !plan_rows || !plan_sptr || !plan_cols || !plan_cell || !vec_a || !out) return -22;
    if (mode == 1 && (!vec_b || !dz || !ds_dst || !ds_src)) return -22;
    hipStream_t st_ = (hipStream_t)stream;
    const unsigned grid = 8u * (unsigned)((nb * H + 7) / 8);
    if (C == 512) { if (mode == 0) AGG_LAUNCH(4, 0); else AGG_LAUNCH(4, 1); }
    else if (C == 256) { if (mode == 0) AGG_LAUNCH(2, 0); else AGG_LAUNCH(2, 1); }
    else { if (mode == 0) AGG_LAUNCH(1, 0); else AGG_LAUNCH(1, 1); }
    return hipGetLastError() == hipSuccess ? 0 : -5;
}

int spadot_gat_edge_dot(const void *g_out, const void *out, const void *h, int dtype, const int *plan_rows,
                        const int *plan_sptr, const int *plan_cols, const int *plan_cell, int nb, int max_cols, int H,
                        int C, int act, void *g_pre, float *dz, void *stream) {
    if (!spadot_gat_mfma_supported(dtype, C, max_cols) || nb <= 0 || H <= 0) return -22;
    if (!g_out || !h || !plan_rows || !plan_sptr || !plan_cols || !plan_cell || !g_pre || !dz || (act && !out)) return -22;
    hipStream_t st_ = (hipStream_t)stream;
    const unsigned grid = 8u * (unsigned)((nb * H + 7) / 8);
#define EDOT_LAUNCH(NTW)                                                                                           \
    hipLaunchKernelGGL(k_gat_edot<NTW>, dim3(grid), dim3(NT), 0, st_, (const __bf16 *)g_out, (const __bf16 *)out,    \
                       (const __bf16 *)h, plan_rows, plan_sptr, plan_cols, plan_cell, nb, H, act, (__bf16 *)g_pre, dz)
    if (C == 512) EDOT_LAUNCH(4);
    else if (C == 256) EDOT_LAUNCH(2);
    else EDOT_LAUNCH(1);
#undef EDOT_LAUNCH
    return hipGetLastError() == hipSuccess ? 0 : -5;
}

}  // extern "C"
