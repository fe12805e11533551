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
#include <type_traits>

#include "../../include/spadot_model.h"
#include "per_device.h"

namespace {

constexpr float ATT_SLOPE = 0.2f;    // GATConv negative_slope
constexpr float ACT_SLOPE = 0.01f;   // F.leaky_relu default (encoder.py:56-57)
constexpr int ROWS = 32;             // rows per block (graph.py: PLAN_ROWS) = MFMA N (targets on the lanes)
constexpr int KSTEP = 16;            // sources per chunk = MFMA K
constexpr int NT = 256;              // threads per workgroup (4 waves)

typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef short s4 __attribute__((ext_vector_type(4)));
typedef short s8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));      // a native 16-byte value (HIP's uint4 class kept register sets in scratch)

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
// Per-node kernels: one WAVE per node, lane = (edge slot, head) with head = lane % H (H in {1, 2, 4, 8}), so 64 / H
// edges per pass and every per-edge array ([E, H] fp32) is read and written in whole 256-byte runs.  Reductions over
// the edges of one head are butterflies over the lanes with equal lane % H.  The first four passes (64 * 4 / H edges:
// every kNN graph here) keep their logits in registers; longer rows recompute them.
// ------------------------------------------------------------------------------------------------------------------
template <int H> __device__ __forceinline__ float head_max(float x) {
#pragma unroll
    for (int off = H; off < 64; off <<= 1) x = fmaxf(x, __shfl_xor(x, off, 64));
    return x;
}
template <int H> __device__ __forceinline__ float head_sum(float x) {
#pragma unroll
    for (int off = H; off < 64; off <<= 1) x += __shfl_xor(x, off, 64);
    return x;
}

// Element offset of edge cell `cq` (graph.py: BlockPlan.cellq) inside the dense weight image
// acell [chunk][head][hi | lo][512] (bf16): cq = chunk * 512 + position inside the 32 x 16 tile in fragment order.
__device__ __forceinline__ size_t acell_off(int cq, int H, int hd) {
    return ((size_t)(cq >> 9) * H + hd) * 1024 + (size_t)(cq & 511);
}
__device__ __forceinline__ void acell_put(__bf16 *__restrict__ acell, size_t off, float a) {
    const __bf16 hi = (__bf16)a;
    acell[off] = hi;
    acell[off + 512] = (__bf16)(a - (float)hi);
}

// alpha[e, hd] = softmax over the incoming edges e of target i of LeakyReLU_0.2(s_src[j] + s_dst[i])
// (exp(e - max) / (sum + 1e-16): SURVEY App. A), also scattered as bf16 hi + lo into the plan's dense weight image --
// and, when the layer will be differentiated (cellq_s / acell_s given), into the SOURCE-side plan's image as well: the
// backward product alpha^T g_pre reads the same weights, so that scatter (until round 4 part of k_gat_softmax_bwd, ~20 MB of
// 2-byte writes on the backward pass's critical chain) is gradient-independent and belongs to the forward pass, whose GAT
// branch ends ~80 us before the SVGP branch it runs beside.
template <int H>
__global__ __launch_bounds__(256) void k_gat_alpha(const float *__restrict__ s_src, const float *__restrict__ s_dst,
                                                   const int *__restrict__ rowptr, const int *__restrict__ col,
                                                   const int *__restrict__ cellq, int n_tgt, float *__restrict__ alpha,
                                                   __bf16 *__restrict__ acell, const int *__restrict__ cellq_s,
                                                   __bf16 *__restrict__ acell_s) {
    constexpr int EP = 64 / H;
    const int lane = threadIdx.x & 63, i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= n_tgt) return;
    const int hd = lane % H, es = lane / H;
    const int p0 = rowptr[i], deg = rowptr[i + 1] - p0;
    if (deg <= 0) return;                                              // (wave-uniform: nothing to write)
    const float sd = s_dst[(size_t)i * H + hd];
    const int npass = (deg + EP - 1) / EP;
    // Three round trips, not one or two per pass: the row pointer; then EVERY per-edge index of the first four passes (source
    // id and the cells of both weight images), loaded unconditionally from a clamped position (a predicated load is a
    // branch the compiler drains the queue at); then the sources' logit halves.
    const int plast = p0 + deg - 1;
    const int *__restrict__ cq_s = acell_s ? cellq_s : cellq;
    int cj[4], cq_t[4], cq_u[4];
#pragma unroll
    for (int t = 0; t < 4; t++) {
        const int pc = min(p0 + t * EP + es, plast);
        cj[t] = col[pc]; cq_t[t] = cellq[pc]; cq_u[t] = cq_s[pc];
    }
    float sv[4];
#pragma unroll
    for (int t = 0; t < 4; t++) sv[t] = s_src[(size_t)cj[t] * H + hd];
    float ev[4];
    float m = -INFINITY;
#pragma unroll
    for (int t = 0; t < 4; t++) {
        const int p = p0 + t * EP + es;
        ev[t] = (t < npass && p < p0 + deg) ? leaky(sv[t] + sd, ATT_SLOPE) : -INFINITY;
        m = fmaxf(m, ev[t]);
    }
    for (int t = 4; t < npass; t++) {
        const int p = p0 + t * EP + es;
        if (p < p0 + deg) m = fmaxf(m, leaky(s_src[(size_t)col[p] * H + hd] + sd, ATT_SLOPE));
    }
    m = head_max<H>(m);
    float ssum = 0.f;
#pragma unroll
    for (int t = 0; t < 4; t++) ssum += __expf(ev[t] - m);          // exp(-inf) = 0 for the empty slots
    for (int t = 4; t < npass; t++) {
        const int p = p0 + t * EP + es;
        if (p < p0 + deg) ssum += __expf(leaky(s_src[(size_t)col[p] * H + hd] + sd, ATT_SLOPE) - m);
    }
    const float inv = 1.f / (head_sum<H>(ssum) + 1e-16f);
#pragma unroll
    for (int t = 0; t < 4; t++) {
        const int p = p0 + t * EP + es;
        if (t < npass && p < p0 + deg) {
            const float a = __expf(ev[t] - m) * inv;
            alpha[(size_t)p * H + hd] = a;
            acell_put(acell, acell_off(cq_t[t], H, hd), a);
            if (acell_s) acell_put(acell_s, acell_off(cq_u[t], H, hd), a);
        }
    }
    for (int t = 4; t < npass; t++) {
        const int p = p0 + t * EP + es;
        if (p < p0 + deg) {
            const float a = __expf(leaky(s_src[(size_t)col[p] * H + hd] + sd, ATT_SLOPE) - m) * inv;
            alpha[(size_t)p * H + hd] = a;
            acell_put(acell, acell_off(cellq[p], H, hd), a);
            if (acell_s) acell_put(acell_s, acell_off(cellq_s[p], H, hd), a);
        }
    }
}

// dz[e] holds the raw d(alpha[e]) on entry; on exit dz[e] = d(logit[e]) = alpha (d(alpha) - sum_k alpha_k d(alpha_k)) *
// LeakyReLU'(z) and ds_dst[i] = sum_e dz[e].  With acell_s given, alpha is also scattered into the SOURCE-side plan's weight
// image (what the backward product alpha^T g_pre reads) -- callers that had k_gat_alpha write that image pass NULL.
template <int H>
__global__ __launch_bounds__(256) void k_gat_softmax_bwd(const float *__restrict__ alpha, const float *__restrict__ s_src,
                                                         const float *__restrict__ s_dst, const int *__restrict__ rowptr,
                                                         const int *__restrict__ col, const int *__restrict__ cellq_s,
                                                         int n_tgt, int n_all, float *__restrict__ dz,
                                                         float *__restrict__ ds_dst, __bf16 *__restrict__ acell_s) {
    constexpr int EP = 64 / H;
    const int lane = threadIdx.x & 63, i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= n_tgt) {        // nodes that are sources only (rows n_tgt .. n_all - 1): no incoming edge, zero gradient
        if (i < n_all && lane < H) ds_dst[(size_t)i * H + lane] = 0.f;
        return;
    }
    const int hd = lane % H, es = lane / H;
    const int p0 = rowptr[i], deg = rowptr[i + 1] - p0;
    if (deg <= 0) {                                                    // (wave-uniform)
        if (lane < H) ds_dst[(size_t)i * H + lane] = 0.f;
        return;
    }
    const float sd = s_dst[(size_t)i * H + hd];
    const int npass = (deg + EP - 1) / EP;
    // as in k_gat_alpha: every per-edge value of the first four passes in ONE round trip (unconditional loads from a clamped
    // position), the sources' logit halves in a second one
    const int plast = p0 + deg - 1;
    const int *__restrict__ cq_s = acell_s ? cellq_s : col;            // (any readable int array when there is no image)
    float av[4], dv[4];
    int cj[4], cq_u[4];
#pragma unroll
    for (int t = 0; t < 4; t++) {
        const int pc = min(p0 + t * EP + es, plast);
        av[t] = alpha[(size_t)pc * H + hd]; dv[t] = dz[(size_t)pc * H + hd]; cj[t] = col[pc]; cq_u[t] = cq_s[pc];
    }
    float sv[4];
#pragma unroll
    for (int t = 0; t < 4; t++) sv[t] = s_src[(size_t)cj[t] * H + hd];
    float dsum = 0.f;
#pragma unroll
    for (int t = 0; t < 4; t++) {
        const int p = p0 + t * EP + es;
        if (!(t < npass && p < p0 + deg)) { av[t] = 0.f; dv[t] = 0.f; }
        dsum += av[t] * dv[t];
    }
    for (int t = 4; t < npass; t++) {
        const int p = p0 + t * EP + es;
        if (p < p0 + deg) dsum += alpha[(size_t)p * H + hd] * dz[(size_t)p * H + hd];
    }
    dsum = head_sum<H>(dsum);
    float dsd = 0.f;
    auto finish = [&](int p, float a, float da, float ss, int cqe) __attribute__((always_inline)) {
        const float z = ss + sd;
        const float d = a * (da - dsum) * (z > 0.f ? 1.f : ATT_SLOPE);
        dz[(size_t)p * H + hd] = d;
        dsd += d;
        if (acell_s) acell_put(acell_s, acell_off(cqe, H, hd), a);
    };
#pragma unroll
    for (int t = 0; t < 4; t++) {
        const int p = p0 + t * EP + es;
        if (t < npass && p < p0 + deg) finish(p, av[t], dv[t], sv[t], cq_u[t]);
    }
    for (int t = 4; t < npass; t++) {
        const int p = p0 + t * EP + es;
        if (p < p0 + deg) finish(p, alpha[(size_t)p * H + hd], dz[(size_t)p * H + hd], s_src[(size_t)col[p] * H + hd], acell_s ? cellq_s[p] : 0);
    }
    dsd = head_sum<H>(dsd);
    if (es == 0) ds_dst[(size_t)i * H + hd] = dsd;
}

// ds_src[j, hd] = sum of dz over the OUTGOING edges of j (transposed CSR; eid_t = position of the edge in the by-target order)
template <int H>
__global__ __launch_bounds__(256) void k_gat_dsrc(const float *__restrict__ dz, const int *__restrict__ rowptr_t,
                                                  const int *__restrict__ eid_t, int n, float *__restrict__ ds_src) {
    constexpr int EP = 64 / H;
    const int lane = threadIdx.x & 63, j = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (j >= n) return;
    const int hd = lane % H, es = lane / H;
    const int p0 = rowptr_t[j], p1 = rowptr_t[j + 1];
    float acc = 0.f;
    for (int p = p0 + es; p < p1; p += EP) acc += dz[(size_t)eid_t[p] * H + hd];
    acc = head_sum<H>(acc);
    if (es == 0) ds_src[(size_t)j * H + hd] = acc;
}

// ------------------------------------------------------------------------------------------------------------------
// k_gat_agg: D^T [512 x 32] = X^T [512 x S] . A^T [S x 32] for one (block, head); C = 512 channels per head.
//
// Data path: every 16-source chunk is 16 rows x 1 KiB of X plus its 2 KiB weight tile (hi | lo, already in fragment
// order: k_gat_alpha wrote it).  Both go global -> LDS by LDS-DMA (global_load_lds_dwordx4: 1 KiB per wave instruction,
// no registers): per chunk every wave issues 4 row pieces (rows 4 wave .. 4 wave + 3) and one half-wave piece of the
// weight tile, FIVE vector-memory operations, so "chunk c has landed" is s_waitcnt vmcnt(5) while chunk c + 1 is in
// flight (vmcnt counts in issue order) and vmcnt(0) on the last chunk.  LDS: ring of AGG_RING chunks (row stride 1088 B: the
// transposed reads are bank-conflict free) + 3 weight tiles + the column ids.  One raw s_barrier per chunk:
//     wait(chunk it) ; barrier ; request chunk it + 2 into the slot chunk it - 1 just left ; multiply chunk it
// (__syncthreads() would drain the DMAs: its fence waits for vmcnt(0)).  No ordinary global load lives inside the loop
// (hipcc waits vmcnt(0) for one while a DMA is in flight): the column ids are read from LDS.
// MFMA operands: A operand = X^T fragment (lane: channel l & 31, sources 8 (l >> 5) .. + 7) by two ds_read_b64_tr_b16 of
// the row-major image; B operand = the weights (lane: row l & 31, same 8 sources), hi then lo.  The result has the row
// (target) on the lane and 16 channels in registers; it leaves through LDS so that every global store is a whole
// 16-byte piece of a row.
// MODE 0 (forward):         out = leaky?(D + bias)
// MODE 1 (source backward): out = D + ds_src att_src + ds_dst att_dst
// ------------------------------------------------------------------------------------------------------------------
namespace agg {
constexpr int C = 512, PPR = C / 8;
constexpr int ROWB = 2 * C + 64;                    // staged row stride (bytes), = 64 mod 256
constexpr int CHUNKB = KSTEP * ROWB;                // 17408
constexpr int OUTB = 2 * C + 16;                    // epilogue image row stride (bytes)
constexpr int ATILEB = 2048;                        // hi (1 KiB) | lo (1 KiB)
#ifndef AGG_RING
#define AGG_RING 2                                  // round 3, tools/agg_variants.py: 2: 45.2 us, 3: 47.5 us, 4: 65 us (80 KB: one workgroup per CU)
#endif
#ifndef AGG_WGS
#define AGG_WGS 2                                   // workgroups per compute unit the register allocation is held to
#endif
#ifndef EDOT_WGS
#define EDOT_WGS 4
#endif
#ifndef EDOT_PROBE
#define EDOT_PROBE 0                                // measurement builds only (tools/gat_probe.sh): bit 0 no `out` read / g_pre write,
#endif                                              // bit 1 every X row = the block's first column, bit 2 no dz scatter, bit 3 prologue only
#ifndef EDOT_PAD
#define EDOT_PAD 1                                  // 0 (with -DEDOT_WGS=5): unpadded, XOR-swizzled G tile, five workgroups per CU
#endif
constexpr int RING = AGG_RING;                      // chunks of LDS ring: RING - 1 in flight while one is multiplied
constexpr int MAXC = 1024 - ROWS;                   // longest column list of a block (ids + row ids = 4 KiB)
constexpr int LDS_RING = RING * CHUNKB;             // 52224 >= 32 * OUTB
constexpr int LDS_BYTES = LDS_RING + RING * ATILEB + MAXC * 4 + ROWS * 4;
}  // namespace agg

typedef __attribute__((address_space(3))) void *lds_ptr_t;

template <int MODE>
__global__ __launch_bounds__(NT, AGG_WGS) void k_gat_agg(
    const __bf16 *__restrict__ X, const __bf16 *__restrict__ acell, const int *__restrict__ prow,
    const int *__restrict__ sptr, const int *__restrict__ pcol, int nb, int H,
    const float *__restrict__ vec_a,      // MODE 0: bias [H*C]; MODE 1: att_src [H*C]
    const float *__restrict__ vec_b,      // MODE 1: att_dst [H*C]
    int act, const float *__restrict__ ds_src, const float *__restrict__ ds_dst, __bf16 *__restrict__ out,
    const __bf16 *__restrict__ hrows,     // MODE 1, optional: h (the layer's input rows), for the attention-vector gradients
    float *__restrict__ att_part, int part_width,     // MODE 1: [nb][part_width] partials: src at column 0, dst at column H*C
    int pad_from, int pad_to,             // rows pad_from .. pad_to - 1 of `out` are set to zero (row padding for the next GEMM)
    const float *__restrict__ dz, const int *__restrict__ rowptr_t, const int *__restrict__ eid_t,   // MODE 1, optional: see ds_own
    float *__restrict__ ds_src_out) {
    using namespace agg;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char *ring = smem;
    unsigned char *atile = smem + LDS_RING;
    int *sids = reinterpret_cast<int *>(atile + RING * ATILEB);
    int *rid = sids + MAXC;

    const int item = xcd_item(nb * H);
    if (item >= nb * H) return;
    const int b = item / H, hd = item - b * H;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (b == 0 && pad_to > pad_from) {
        // The output is allocated with its row count rounded up (the library's GEMMs run 10-17 % faster on M % 128 == 0:
        // tools/gemm_mpad.py); the rows past the last node must read as zeros there (a weight gradient sums over ALL rows).
        // Block 0's workgroup of each head clears its 1 KiB slice of them.
        const size_t HCp = (size_t)H * C;
        for (int k = tid; k < (pad_to - pad_from) * PPR; k += NT) {
            const int r = pad_from + k / PPR, pc = k - (k / PPR) * PPR;
            *reinterpret_cast<uint4 *>(out + (size_t)r * HCp + (size_t)hd * C + (size_t)pc * 8) = make_uint4(0u, 0u, 0u, 0u);
        }
    }
    const int s0 = sptr[b], ncol = sptr[b + 1] - s0, nch = ncol / KSTEP;
    const int q0 = s0 / KSTEP;                                            // first global chunk of this block
    for (int k = tid; k < ncol; k += NT) sids[k] = pcol[s0 + k];
    if (tid < ROWS) rid[tid] = prow[b * ROWS + tid];
    __syncthreads();
    __shared__ float ds_own[ROWS];
    const bool own_ds = MODE == 1 && dz != nullptr;

    const size_t hoff = (size_t)hd * C;
    const size_t HC = (size_t)H * C;
    const __bf16 *xlane = X + hoff + (size_t)lane * 8;                    // this lane's 16-byte piece of a row

    // LDS-DMA by inline asm (recipe of cdna_hip_programming.md 5.7): issued through the builtin, hipcc knows the DMA writes
    // LDS and puts s_waitcnt vmcnt(0) in front of the next LDS read -- which drains the chunks that are meant to stay in
    // flight.  The asm form is invisible to that bookkeeping; the counted waits below are ours.  M0 (LDS destination
    // base) is saved and restored inside the statement.
    auto dma16 = [&](const void *gsrc, unsigned lds_dst) __attribute__((always_inline)) {
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
    };
    const unsigned lds0 = (unsigned)(size_t)smem;                         // LDS byte offset of the dynamic segment
    auto request = [&](int c) __attribute__((always_inline)) {            // 5 vector-memory operations per wave
        const unsigned slot = lds0 + (unsigned)(c % RING) * CHUNKB;
        const int4 id4 = *reinterpret_cast<const int4 *>(sids + c * KSTEP + 4 * wave);      // rows 4 wave .. 4 wave + 3
        const int ids[4] = {id4.x, id4.y, id4.z, id4.w};
#pragma unroll
        for (int m = 0; m < 4; m++) {
            const int r = 4 * wave + m;
            const int sid = __builtin_amdgcn_readfirstlane(ids[m]);
            dma16(xlane + (size_t)sid * HC, __builtin_amdgcn_readfirstlane(slot + r * ROWB));
        }
        if (lane < 32)       // half a wave: 512 B, quarter `wave` of the chunk's weight tile
            dma16(acell + ((size_t)(q0 + c) * H + hd) * 1024 + wave * 256 + lane * 8,
                  __builtin_amdgcn_readfirstlane(lds0 + LDS_RING + (unsigned)(c % RING) * ATILEB + wave * 512));
    };

    f16v acc[4];
#pragma unroll
    for (int t = 0; t < 4; t++)
#pragma unroll
        for (int i = 0; i < 16; i++) acc[t][i] = 0.f;

    // transposed-read addresses of this lane inside a chunk (T10): 16-lane group -> columns 16 g .. 16 g + 15 of the
    // N-tile, lane 4 q + p of the group supplies row q, columns 4 p .. 4 p + 3; the two reads of a fragment are rows
    // 8 hh .. + 3 and 8 hh + 4 .. + 7
    const int hh = lane >> 5, g16 = (lane >> 4) & 1, qq = (lane & 15) >> 2, pp = lane & 3;
    const int tr_off = (8 * hh + qq) * ROWB + (16 * g16 + 4 * pp) * 2 + wave * 4 * 64;
    const int a_off = (hh * 32 + (lane & 31)) * 16;      // weight fragment: piece hh * 32 + row (conflict-free b128 reads)

    auto multiply = [&](int c) __attribute__((always_inline)) {
        const unsigned char *slot = ring + (size_t)(c % RING) * CHUNKB + tr_off;
        const unsigned char *at = atile + (c % RING) * ATILEB + a_off;
        const bf8 a_hi = *reinterpret_cast<const bf8 *>(at);
        const bf8 a_lo = *reinterpret_cast<const bf8 *>(at + 1024);
#pragma unroll
        for (int t = 0; t < 4; t++) {
            const unsigned char *base = slot + t * 64;
            const s4 x0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s4 __attribute__((address_space(3))) *)(base));
            const s4 x1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s4 __attribute__((address_space(3))) *)(base + 4 * ROWB));
            const bf8 xf = __builtin_bit_cast(bf8, __builtin_shufflevector(x0, x1, 0, 1, 2, 3, 4, 5, 6, 7));
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xf, a_hi, acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xf, a_lo, acc[t], 0, 0, 0);
        }
    };

    // MODE 1 with att_part: the attention-vector gradients d att_src[c] = sum_j ds_src[j] h[j, c], d att_dst likewise with
    // ds_dst (k_gat_datt_part took a pass of its own over h and g for them: 25 us per layer).  This block's 32 rows of h
    // are requested HERE, 16 bytes per thread and row, in front of the first chunk requests (older in the in-order queue,
    // so the counted waits of the loop are unaffected), and consumed behind the epilogue.
    u32x4 hv[8];
    if (MODE == 1 && att_part != nullptr) {
        const int pc = tid & 63, rg = tid >> 6;
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int node = rid[8 * rg + u];
            hv[u] = node >= 0 ? *reinterpret_cast<const u32x4 *>(hrows + (size_t)node * HC + hoff + (size_t)pc * 8) : u32x4{0u, 0u, 0u, 0u};
        }
    }
#pragma unroll
    for (int c = 0; c < RING - 1; c++)
        if (c < nch) request(c);
    // MODE 1 with dz: this workgroup sums d(logit) over the OUTGOING edges of its own 32 rows for its head (ds_src: what
    // k_gat_dsrc did in a launch of its own) -- a group of 8 lanes per row, fixed order inside a lane, butterfly across.  The
    // three dependent loads (row pointer, edge id, dz) run while the first chunks are in flight; their waits let those land.
    if (own_ds) {
        const int r = tid >> 3, l8 = tid & 7;              // 32 rows x 8 lanes
        const int node = rid[r];
        float acc_ = 0.f;
        if (node >= 0) {
            const int p0 = rowptr_t[node], p1 = rowptr_t[node + 1];
            // the first 32 edges of a row: four edge ids, then four dz values, each set of loads in flight together (a
            // plain loop is a chain of dependent round trips: ~8 of them cost this workgroup more than its first chunk)
            int e4[4];
            float d4[4];
#pragma unroll
            for (int k = 0; k < 4; k++) { const int p = p0 + l8 + 8 * k; e4[k] = p < p1 ? eid_t[p] : -1; }
#pragma unroll
            for (int k = 0; k < 4; k++) d4[k] = e4[k] >= 0 ? dz[(size_t)e4[k] * H + hd] : 0.f;
#pragma unroll
            for (int k = 0; k < 4; k++) acc_ += d4[k];
            for (int p = p0 + l8 + 32; p < p1; p += 8) acc_ += dz[(size_t)eid_t[p] * H + hd];
        }
        acc_ += __shfl_xor(acc_, 1, 64); acc_ += __shfl_xor(acc_, 2, 64); acc_ += __shfl_xor(acc_, 4, 64);
        if (l8 == 0) {
            ds_own[r] = acc_;                              // (read in the epilogue, behind the loop's barriers)
            if (node >= 0 && ds_src_out != nullptr) ds_src_out[(size_t)node * H + hd] = acc_;
        }
    }
    for (int it = 0; it < nch; it++) {
        // chunk `it` has landed when at most the 5-operation groups of the younger chunks in flight are outstanding
        const int younger = min(nch - 1 - it, RING - 2);
        if (younger >= 2) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
        else if (younger == 1) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (it + RING - 1 < nch) request(it + RING - 1);
        multiply(it);
    }

    // ---- epilogue: accumulators -> bf16 image in LDS (the ring is free) -> whole 16-byte pieces to global memory
    __syncthreads();
    {
        const int t = lane & 31;
        const int node = rid[t];
        float sa = 0.f, sb = 0.f;
        if (MODE == 1 && node >= 0) {
            sa = own_ds ? ds_own[t] : ds_src[(size_t)node * H + hd];
            sb = ds_dst[(size_t)node * H + hd];
        }
#pragma unroll
        for (int tt = 0; tt < 4; tt++) {
            const int cb = (wave * 4 + tt) * 32;
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
                *reinterpret_cast<uint2 *>(ring + (size_t)t * OUTB + (size_t)ch * 2) = make_uint2(pack2(v[0], v[1]), pack2(v[2], v[3]));
            }
        }
    }
    __syncthreads();
#pragma unroll
    for (int m = 0; m < 8; m++) {
        const int pi = m * NT + tid, r = pi / PPR, pc = pi - r * PPR;
        const int node = rid[r];
        if (node >= 0)
            *reinterpret_cast<uint4 *>(out + (size_t)node * HC + hoff + (size_t)pc * 8) =
                *reinterpret_cast<const uint4 *>(ring + (size_t)r * OUTB + (size_t)pc * 16);
    }
    if (MODE == 1 && att_part != nullptr) {
        // per-thread sums over its 8 rows, then the four row groups added in order through LDS (fixed order: reproducible)
        __syncthreads();                                                  // the ring image has been read
        float *dsl = reinterpret_cast<float *>(sids);                     // (the column ids are no longer needed)
        if (tid < ROWS) {
            const int node = rid[tid];
            dsl[tid] = node >= 0 ? (own_ds ? ds_own[tid] : ds_src[(size_t)node * H + hd]) : 0.f;
            dsl[ROWS + tid] = node >= 0 ? ds_dst[(size_t)node * H + hd] : 0.f;
        }
        __syncthreads();
        const int pc = tid & 63, rg = tid >> 6;
        float as_[8], ad_[8];
#pragma unroll
        for (int e = 0; e < 8; e++) { as_[e] = 0.f; ad_[e] = 0.f; }
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const float ws = dsl[8 * rg + u], wd = dsl[ROWS + 8 * rg + u];
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const float lo = __uint_as_float(hv[u][q] << 16), hi = __uint_as_float(hv[u][q] & 0xffff0000u);
                as_[2 * q] = fmaf(ws, lo, as_[2 * q]); as_[2 * q + 1] = fmaf(ws, hi, as_[2 * q + 1]);
                ad_[2 * q] = fmaf(wd, lo, ad_[2 * q]); ad_[2 * q + 1] = fmaf(wd, hi, ad_[2 * q + 1]);
            }
        }
        float *red = reinterpret_cast<float *>(ring);                     // [4 row groups][2][512]
#pragma unroll
        for (int e = 0; e < 8; e++) {
            red[(rg * 2 + 0) * C + pc * 8 + e] = as_[e];
            red[(rg * 2 + 1) * C + pc * 8 + e] = ad_[e];
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int idx = k * NT + tid, which = idx / C, ch = idx - which * C;
            float t_ = red[(0 * 2 + which) * C + ch];
            t_ += red[(1 * 2 + which) * C + ch];
            t_ += red[(2 * 2 + which) * C + ch];
            t_ += red[(3 * 2 + which) * C + ch];
            att_part[(size_t)b * part_width + (size_t)which * HC + hoff + ch] = t_;
        }
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
__global__ __launch_bounds__(NT, EDOT_WGS) void k_gat_edot(const __bf16 *__restrict__ g_out, const __bf16 *__restrict__ outp,
                                                    const __bf16 *__restrict__ Xh, const int *__restrict__ prow,
                                                    const int *__restrict__ sptr, const int *__restrict__ pcol,
                                                    const int *__restrict__ cell, int nb, int H, int act,
                                                    __bf16 *__restrict__ g_pre, float *__restrict__ dz,
                                                    float *__restrict__ bias_part, int part_width, int part_col) {
    // G tile in LDS: 32 rows x 2 C bytes, rows padded by 16 bytes against bank conflicts (EDOT_PAD = 1: 33 280 bytes, four
    // workgroups per compute unit) or unpadded with 16-byte chunk c of row r stored at chunk c ^ (r & 15) (EDOT_PAD = 0: 32 KiB,
    // five workgroups fit the 160 KiB, so that the 1248 (block, head) items of a 312-block layer could all be resident at
    // once).  Same-box A/B, round 3 (profiles/r03/ab_edot_layout.txt): no difference -- backward 216 / 204 us against 218 /
    // 213 us in tools/gat_bench.py, 554.9 / 555.0 against 555.6 / 555.8 steps/s -- so the padded form stays.
    constexpr int C = 128 * NTW, PPR = C / 8, GS = 2 * C + 16 * EDOT_PAD, XM = EDOT_PAD ? 0 : 15;
    __shared__ __attribute__((aligned(16))) unsigned char gt[ROWS * GS];
    const int item = xcd_item(nb * H);
    if (item >= nb * H) return;
    const int b = item / H, hd = item - b * H;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const size_t hoff = (size_t)hd * C, HC = (size_t)H * C;
    // Every load below is UNCONDITIONAL, from a clamped address, and all requests of a phase are issued before the first
    // use: a predicated load is a branch, and hipcc waits for everything outstanding at its join -- the first form of
    // this prologue (`if (node >= 0) load`) was eight dependent round trips, 28 of the kernel's 63 us
    // (profiles/r05/gat_probe.txt).  Rows without a node are masked after the fact.
    const int s0 = sptr[b], ntile = (EDOT_PROBE & 8) ? 0 : (sptr[b + 1] - s0) / 32;
    const int sl = lane & 31, hh = lane >> 5;
    int sid_next = wave < ntile ? pcol[s0 + wave * 32 + sl] : 0;          // (first tile's column id: rides with the row ids)
    int nodes[2 * NTW];
#pragma unroll
    for (int m = 0; m < 2 * NTW; m++) nodes[m] = prow[b * ROWS + (m * NT + tid) / PPR];
    u32x4 gv[2 * NTW];
#pragma unroll
    for (int m = 0; m < 2 * NTW; m++) {
        const int pc = (m * NT + tid) % PPR;
        gv[m] = *reinterpret_cast<const u32x4 *>(g_out + (size_t)max(nodes[m], 0) * HC + hoff + (size_t)pc * 8);
    }
    if (act && !(EDOT_PROBE & 1)) {
        u32x4 ov[2 * NTW];
#pragma unroll
        for (int m = 0; m < 2 * NTW; m++) {
            const int pc = (m * NT + tid) % PPR;
            ov[m] = *reinterpret_cast<const u32x4 *>(outp + (size_t)max(nodes[m], 0) * HC + hoff + (size_t)pc * 8);
        }
#pragma unroll
        for (int m = 0; m < 2 * NTW; m++)
#pragma unroll
            for (int e = 0; e < 4; e++) {
                float lo = __uint_as_float(gv[m][e] << 16), hi = __uint_as_float(gv[m][e] & 0xffff0000u);
                if (!(__uint_as_float(ov[m][e] << 16) > 0.f)) lo *= ACT_SLOPE;
                if (!(__uint_as_float(ov[m][e] & 0xffff0000u) > 0.f)) hi *= ACT_SLOPE;
                gv[m][e] = pack2(lo, hi);
            }
    }
#pragma unroll
    for (int m = 0; m < 2 * NTW; m++) {
        const int pi = m * NT + tid, r = pi / PPR, pc = pi - r * PPR;
        const bool on = nodes[m] >= 0;
        const u32x4 g = on ? gv[m] : u32x4{0u, 0u, 0u, 0u};
        if (on && g_pre != nullptr && !(EDOT_PROBE & 1))
            *reinterpret_cast<u32x4 *>(g_pre + (size_t)nodes[m] * HC + hoff + (size_t)pc * 8) = g;
        *reinterpret_cast<u32x4 *>(gt + (size_t)r * GS + (size_t)((pc ^ (r & XM)) * 16)) = g;
    }
    __syncthreads();
    if (bias_part != nullptr && C == 2 * NT) {
        // the bias gradient = column sums of g_pre: this block's 32 rows are in LDS, two channels per thread
        // (per-block partials; spadot_colsum adds the blocks in order)
        float b0 = 0.f, b1 = 0.f;
#pragma unroll 8
        for (int r = 0; r < ROWS; r++) {
            const unsigned w2 = *reinterpret_cast<const unsigned *>(gt + (size_t)r * GS + (size_t)(((tid >> 2) ^ (r & XM)) * 16 + (tid & 3) * 4));
            b0 += __uint_as_float(w2 << 16);
            b1 += __uint_as_float(w2 & 0xffff0000u);
        }
        *reinterpret_cast<float2 *>(bias_part + (size_t)b * part_width + part_col + hoff + 2 * tid) = make_float2(b0, b1);
    }

    // Per tile three things come from memory: the column id (needed first), the 32 pieces of the source row, and the 16
    // cell ids the results are scattered through.  In program order they were three dependent round trips per tile; now
    // the next tile's column id and this tile's cell ids are requested in front of the row pieces, so only those are waited for.
    for (int nt = wave; nt < ntile; nt += 4) {
        const int sid = (EDOT_PROBE & 2) ? pcol[s0] : sid_next;
        const __bf16 *xrow = Xh + (size_t)sid * HC + hoff + 16 * hh;
        const unsigned char *grow = gt + (size_t)sl * GS;
        const int gx = sl & XM;                                       // this row's chunk XOR
        const int *cq = cell + ((size_t)(s0 / KSTEP) + 2 * nt + (sl >> 4)) * (KSTEP * ROWS) + (sl & 15);
        int ce[16];
#pragma unroll
        for (int i = 0; i < 16; i++) ce[i] = cq[((i & 3) + 8 * (i >> 2) + 4 * hh) * KSTEP];
        sid_next = pcol[s0 + min(nt + 4, ntile - 1) * 32 + sl];
        f16v acc;
#pragma unroll
        for (int i = 0; i < 16; i++) acc[i] = 0.f;
#pragma unroll 8
        for (int kp = 0; kp < C / 32; kp++) {
            const bf8 x0 = *reinterpret_cast<const bf8 *>(xrow + kp * 32);
            const bf8 x1 = *reinterpret_cast<const bf8 *>(xrow + kp * 32 + 8);
            const bf8 a0 = *reinterpret_cast<const bf8 *>(grow + (((kp * 4 + 2 * hh) ^ gx) * 16));
            const bf8 a1 = *reinterpret_cast<const bf8 *>(grow + (((kp * 4 + 2 * hh + 1) ^ gx) * 16));
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, x0, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, x1, acc, 0, 0, 0);
        }
        // acc[i] = dA[row (i & 3) + 8 (i >> 2) + 4 hh][column slot nt * 32 + sl]
#pragma unroll
        for (int i = 0; i < 16; i++) {
            const int e = ce[i];
            if (e >= 0 && (!(EDOT_PROBE & 4) || acc[i] == 123.f)) dz[(size_t)e * H + hd] = acc[i];
        }
    }
}

}  // namespace

#define H_DISPATCH(KERNEL, ...)                                              \
    do {                                                                     \
        if (H == 1) hipLaunchKernelGGL(KERNEL<1>, __VA_ARGS__);              \
        else if (H == 2) hipLaunchKernelGGL(KERNEL<2>, __VA_ARGS__);         \
        else if (H == 4) hipLaunchKernelGGL(KERNEL<4>, __VA_ARGS__);         \
        else hipLaunchKernelGGL(KERNEL<8>, __VA_ARGS__);                     \
    } while (0)

extern "C" {

int spadot_gat_mfma_supported(int dtype, int H, int C, int max_cols) {
    return dtype == SPADOT_DT_BF16 && C == agg::C && (H == 1 || H == 2 || H == 4 || H == 8) && max_cols >= 0 &&
           max_cols <= agg::MAXC && max_cols % 32 == 0;
}

int spadot_gat_alpha(const float *s_src, const float *s_dst, const int *rowptr, const int *col, const int *cellq, int n_tgt,
                     int H, float *alpha, void *acell, const int *cellq_s, void *acell_s, void *stream) {
    if (n_tgt <= 0 || !(H == 1 || H == 2 || H == 4 || H == 8) || !cellq || !alpha || !acell) return -22;
    if ((cellq_s == nullptr) != (acell_s == nullptr)) return -22;
    H_DISPATCH(k_gat_alpha, dim3((unsigned)((n_tgt + 3) / 4)), dim3(256), 0, (hipStream_t)stream, s_src, s_dst, rowptr, col, cellq,
               n_tgt, alpha, (__bf16 *)acell, cellq_s, (__bf16 *)acell_s);
    return hipGetLastError() == hipSuccess ? 0 : -5;
}

int spadot_gat_softmax_backward(const float *alpha, const float *s_src, const float *s_dst, const int *rowptr,
                                const int *col, const int *cellq_s, int n_tgt, int n_all, int H, float *dz, float *ds_dst,
                                void *acell_s, void *stream) {
    if (n_tgt <= 0 || n_all < n_tgt || !(H == 1 || H == 2 || H == 4 || H == 8) || ((cellq_s == nullptr) != (acell_s == nullptr))) return -22;
    H_DISPATCH(k_gat_softmax_bwd, dim3((unsigned)((n_all + 3) / 4)), dim3(256), 0, (hipStream_t)stream, alpha, s_src, s_dst, rowptr,
               col, cellq_s, n_tgt, n_all, dz, ds_dst, (__bf16 *)acell_s);
    return hipGetLastError() == hipSuccess ? 0 : -5;
}

int spadot_gat_ds_src(const float *dz, const int *rowptr_t, const int *eid_t, int n, int H, float *ds_src, void *stream) {
    if (n <= 0 || !(H == 1 || H == 2 || H == 4 || H == 8)) return -22;
    H_DISPATCH(k_gat_dsrc, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, (hipStream_t)stream, dz, rowptr_t, eid_t, n, ds_src);
    return hipGetLastError() == hipSuccess ? 0 : -5;
}

#define AGG_LAUNCH(MODE)                                                                                           \
    do {                                                                                                           \
        auto kern = k_gat_agg<MODE>;                                                                               \
        static PerDeviceFlag attr_set;                                                                                   \
        if (!attr_set) {                                                                                           \
            if (hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, agg::LDS_BYTES) != hipSuccess) return -5; \
            attr_set = true;                                                                                       \
        }                                                                                                          \
        hipLaunchKernelGGL(kern, dim3(grid), dim3(NT), agg::LDS_BYTES, st_, (const __bf16 *)x, (const __bf16 *)acell, plan_rows, \
                           plan_sptr, plan_cols, nb, H, vec_a, vec_b, act, ds_src, ds_dst, (__bf16 *)out,          \
                           (const __bf16 *)h_rows, att_part, part_width, pad_from, pad_to, dz, rowptr_t, eid_t,    \
                           ds_src_out);                                                                            \
    } while (0)

int spadot_gat_aggregate(const void *x, int dtype, const void *acell, const int *plan_rows, const int *plan_sptr,
                         const int *plan_cols, int nb, int max_cols, int H, int C, int mode, const float *vec_a,
                         const float *vec_b, int act, const float *ds_src, const float *ds_dst, void *out,
                         const void *h_rows, float *att_part, int part_width, int pad_from, int pad_to, const float *dz,
                         const int *rowptr_t, const int *eid_t, float *ds_src_out, void *stream) {
    if (!spadot_gat_mfma_supported(dtype, H, C, max_cols) || nb <= 0 || (mode != 0 && mode != 1)) return -22;
    if (pad_from < 0 || pad_to < pad_from || pad_to - pad_from > 4096) return -22;
    if (!x || !acell || !plan_rows || !plan_sptr || !plan_cols || !vec_a || !out) return -22;
    if (mode == 1 && (!vec_b || !ds_dst || (!ds_src && !dz))) return -22;
    if (dz && (mode != 1 || !rowptr_t || !eid_t)) return -22;
    if (att_part && (mode != 1 || !h_rows || part_width < 2 * H * C || ((uintptr_t)h_rows & 15))) return -22;
    hipStream_t st_ = (hipStream_t)stream;
    const unsigned grid = 8u * (unsigned)((nb * H + 7) / 8);
    if (mode == 0) AGG_LAUNCH(0); else AGG_LAUNCH(1);
    return hipGetLastError() == hipSuccess ? 0 : -5;
}

int spadot_gat_edge_dot(const void *g_out, const void *out, const void *h, int dtype, const int *plan_rows,
                        const int *plan_sptr, const int *plan_cols, const int *plan_cell, int nb, int max_cols, int H,
                        int C, int act, void *g_pre, float *dz, float *bias_part, int part_width, int part_col, void *stream) {
    if (!spadot_gat_mfma_supported(dtype, H, C, max_cols) || nb <= 0) return -22;
    // (g_pre may be NULL when act == 0: the caller then keeps using g_out -- a gradient that arrived already masked)
    if (!g_out || !h || !plan_rows || !plan_sptr || !plan_cols || !plan_cell || !dz || (act && (!out || !g_pre))) return -22;
    if (bias_part && (part_col < 0 || part_col % 2 || part_width % 2 || part_width < part_col + H * C || ((uintptr_t)bias_part & 7))) return -22;
    hipStream_t st_ = (hipStream_t)stream;
    const unsigned grid = 8u * (unsigned)((nb * H + 7) / 8);
    hipLaunchKernelGGL(k_gat_edot<4>, dim3(grid), dim3(NT), 0, st_, (const __bf16 *)g_out, (const __bf16 *)out,
                       (const __bf16 *)h, plan_rows, plan_sptr, plan_cols, plan_cell, nb, H, act, (__bf16 *)g_pre, dz,
                       bias_part, part_width, part_col);
    return hipGetLastError() == hipSuccess ? 0 : -5;
}

}  // extern "C"
