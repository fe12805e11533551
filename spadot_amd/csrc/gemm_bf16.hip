// gemm_bf16.hip -- C [M x N] = A [M x K] . B^T (B stored [N x K]), bf16 operands, fp32 accumulate, bf16 result (gfx950).
//
// The dense map of a GAT layer, h = x W^T (/root/reference/SpaDOT/model/encoder.py:41-58: GATConv's `lin`), at the shapes
// of the training step: M = n_sub ~ 10^4 rows, N = H C = 2048, K = 3072 (genes, padded) or 2048.  Both operands are
// K-contiguous, so every MFMA fragment is a plain 16-byte row read.
//
// Design (one 320 x 256 tile per compute unit: 32 x 8 = 256 tiles for M <= 10240, N = 2048 -- one wave of workgroups, where
// the library's 832 tiles of 192 x 128 or 312 of 256 x 256 leave the last round a quarter full):
//   * 512 threads = 8 waves as 2 (M) x 4 (N), each wave a 160 x 64 sub-tile = 5 x 2 MFMA tiles (160 accumulator VGPRs);
//   * K-step = 64: a stage is 320 + 256 rows x 128 B = 72 KiB of WHOLE cache lines, two stages in LDS (144 KiB), filled by
//     LDS-DMA (global_load_lds_dwordx4, 8 rows per wave instruction, scalar base + constant per-lane offset), the 16-byte
//     chunks XOR-swizzled through the SOURCE address so that the fragment reads (ds_read_b128) are bank-conflict free
//     without padding.  (A first version with 64-byte rows -- K-step 32, four stages -- ran 148 us instead of 110: every line
//     was requested twice, half used each time);
//   * software pipeline over 16-wide k-steps: fragments of the next k-step are read while the 10 MFMAs of the current one
//     run (two named fragment sets, reads as inline asm with hand-counted lgkmcnt); ONE raw s_barrier per K-step, in front
//     of the stage's last k-step: by then every wave has read the stage, the slot is re-requested for stage P + 2;
//   * workgroups that share an A panel sit on one XCD (its L2 serves the 8 re-reads);
//   * epilogue per wave through a private 4 KiB LDS patch so that global stores are whole 128-byte row pieces.
// Measured alone (tools/gemm_bench.py, random data, same box as the library): 9980 x 2048 x 3072 in 110 us = 1.14 PFLOP/s
// (library 119 us); built without the DMA 99 us, without the MFMAs 67 us, with neither 39 us, skeleton 19 us -- i.e. the MFMAs
// alone run at 1.57 PFLOP/s, the rate of a bare v_mfma_f32_32x32x16_bf16 loop on random data on this chip
// (tools/mfma_peak.hip: 1.59-1.78), and the DMA (885 MB per GEMM from L2 at ~18 TB/s) hides behind them but for ~11 us.
// Inside the training step it pays for the SECOND layer's map (with its last row panels cut into contraction slices beside the
// SVGP inverse: launch_gemm_split below) and for that layer's input gradient; the first layer's map stays on the library
// (spadot_amd/ops.py: GEMM_FWD_SHAPES, GEMM_BUSY_CUS).
#include <hip/hip_runtime.h>
#include <cstdint>

#include "../../include/spadot_model.h"
#include "per_device.h"

#ifndef GEMM_ABLATE
#define GEMM_ABLATE 0        // experiments only: 1 = no DMA after the prologue, 2 = no MFMA, 4 = no fragment reads
#endif

namespace {

typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));

constexpr int BM = 320, BN = 256, BK = 64, STAGES = 2, NT = 512;
constexpr int ROWB = BK * 2;                         // 128 bytes per staged row: one cache line
constexpr int STAGE_ROWS = BM + BN;                  // 576
constexpr int STAGEB = STAGE_ROWS * ROWB;            // 73728
constexpr int PIECES = STAGE_ROWS / 8 / 8;           // 9 wave instructions (8 rows x 128 B each) per wave and stage
constexpr int LDS_BYTES = STAGES * STAGEB;           // 147456 (the epilogue patches reuse it)

__device__ __forceinline__ unsigned pack2(float lo, float hi) {
    const unsigned short a = __builtin_bit_cast(unsigned short, (__bf16)lo), b = __builtin_bit_cast(unsigned short, (__bf16)hi);
    return (unsigned)a | ((unsigned)b << 16);
}

typedef short s4v __attribute__((ext_vector_type(4)));

// BT = false: B stored [N x K] (contraction contiguous: the forward map, h = x W^T).
// BT = true : B stored [K x N] (contraction index = ROW: the input gradient, dx = g W): its stage is 64 contraction rows x 256
//             columns (512-byte rows, 64-byte granules XORed with row & 3, two rows per LDS-DMA piece) and its fragments are
//             transposed reads (ds_read_b64_tr_b16, two per fragment) -- the B side of csrc/gemm_wgrad_bf16.hip.
template <bool BT>
__global__ __launch_bounds__(NT, 1) void k_gemm_bf16(const __bf16 *__restrict__ A, int lda, const __bf16 *__restrict__ B,
                                                    int ldb, __bf16 *__restrict__ C, int ldc, int M, int N, int K,
                                                    int mtiles, int ntiles, int full_items, int S, int nk_slice,
                                                    float *__restrict__ part, const __bf16 *__restrict__ msk, int ldm, float slope) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // tile of this workgroup: the ntiles column tiles of one row panel are consecutive work items of one XCD.  Items
    // 0 .. full_items - 1 are whole tiles (the grid's first 8 ceil(full_items / 8) workgroups); the tiles behind them (the last
    // row panels: launch_gemm_split) are cut into S slices of the contraction, one workgroup each, which leave fp32 partial
    // tiles for k_gemm_tail_reduce -- when part of the chip is busy with another stream's long kernel, the second round of
    // workgroups is then a quarter of a tile long instead of a whole one.
    const int total = mtiles * ntiles;
    const int full_grid = 8 * ((full_items + 7) >> 3);
    int item, slice = 0;
    if ((int)blockIdx.x < full_grid) {
        const int per_xcd = (full_items + 7) >> 3;
        item = (int)(blockIdx.x & 7) * per_xcd + (int)(blockIdx.x >> 3);
        if (item >= full_items) return;
    } else {
        const int t = (int)blockIdx.x - full_grid;
        item = full_items + t / S;
        slice = t - (t / S) * S;
        if (item >= total) return;
    }
    const bool sliced = item >= full_items;
    const int ks0 = sliced ? slice * nk_slice : 0;                // first 64-wide K-step of this workgroup
    const int mt = item / ntiles, nt = item - mt * ntiles;
    const int m0 = mt * BM, n0 = nt * BN;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;                     // wave grid 2 x 4: rows wm * 160, columns wn * 64
    const unsigned lds0 = (unsigned)(size_t)smem;

    // ---- LDS-DMA: piece q (0..71) of a stage covers staged rows 8 q .. 8 q + 7 (A rows 0..319, then B rows 0..255), whole
    // 128-byte lines; lane i -> row 8 q + i / 8, LDS chunk i % 8, which receives GLOBAL chunk (i % 8) ^ ((row >> 1) & 7) of that
    // row (the fragment reads apply the same XOR).  This wave's pieces: q = wave + 8 u.  Source = scalar base (advanced by
    // 128 bytes per stage) + a 32-bit per-lane offset that never changes.
    unsigned off[PIECES];
#pragma unroll
    for (int u = 0; u < PIECES; u++) {
        const int row = 8 * (wave + 8 * u) + (lane >> 3);
        const int chunk = (lane & 7) ^ ((row >> 1) & 7);
        // pieces u < 5 are A rows (q < 40), the others B rows; rows past M re-read the last row (never stored)
        if (u < 5) {
            off[u] = (unsigned)min(m0 + row, M - 1) * (unsigned)(lda * 2) + chunk * 16;
        } else if (!BT) {
            off[u] = (unsigned)(n0 + row - BM) * (unsigned)(ldb * 2) + chunk * 16;
        } else {
            const int brow = 2 * (wave + 8 * (u - 5)) + (lane >> 5);                 // contraction row inside the stage
            off[u] = (unsigned)brow * (unsigned)(ldb * 2) + (unsigned)n0 * 2 + (((lane & 31) ^ ((brow & 3) << 2)) << 4);
        }
    }
    auto request = [&](int ks) __attribute__((always_inline)) {
        const unsigned base = lds0 + (unsigned)(ks % STAGES) * STAGEB + (unsigned)wave * 1024u;
        const char *ga = reinterpret_cast<const char *>(A) + (size_t)(ks0 + ks) * ROWB;
        const char *gb = reinterpret_cast<const char *>(B) + (BT ? (size_t)(ks0 + ks) * BK * ldb * 2 : (size_t)(ks0 + ks) * ROWB);
#pragma unroll
        for (int u = 0; u < PIECES; u++) {
            unsigned keep;
            const unsigned dst = __builtin_amdgcn_readfirstlane(base + (unsigned)u * 8192u);
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(off[u]), "s"(u < 5 ? ga : gb), "s"(dst) : "memory");
        }
    };

    f16v acc[5][2];
#pragma unroll
    for (int i = 0; i < 5; i++)
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int e = 0; e < 16; e++) acc[i][j][e] = 0.f;

    // fragment reads: lane (r = lane & 31, hh = lane >> 5), 16-wide k-step s4 in 0..3: global chunk 2 s4 + hh of its row, found
    // at LDS chunk (2 s4 + hh) ^ ((row >> 1) & 7); (row >> 1) & 7 == (r >> 1) & 7 for every tile of this wave, so one base
    // per k-step and operand, tiles at immediate offsets of 32 rows.  Inline asm: the compiler's own counter model would put an
    // lgkmcnt(0) in front of each MFMA batch; here every wait is written by hand, with a sched_barrier behind it to keep the
    // (register-only) MFMAs from moving across.
    const int r = lane & 31, hh = lane >> 5;
    unsigned a_base[4], b_base[4];
#pragma unroll
    for (int s4 = 0; s4 < 4; s4++) {
        const unsigned sw = (unsigned)(((2 * s4 + hh) ^ ((r >> 1) & 7)) << 4);
        a_base[s4] = lds0 + (unsigned)(wm * 160 + r) * ROWB + sw;
        b_base[s4] = lds0 + (unsigned)(BM + wn * 64 + r) * ROWB + sw;
    }
    // BT: transposed reads of the [64 x 256] B image: lane (hh, g16, qq, pp) -> contraction row 8 hh + qq (+ 4), columns
    // 64 wn + 32 j + 16 g16 + 4 pp .. + 3, the 64-byte granule index XORed with qq
    unsigned bt_base[2];
    {
        const int g16 = (lane >> 4) & 1, qq = (lane & 15) >> 2, pp = lane & 3;
#pragma unroll
        for (int j = 0; j < 2; j++) {
            const int g = wn * 2 + j;
            bt_base[j] = lds0 + (unsigned)(BM * ROWB) + (unsigned)((8 * hh + qq) * 512 + (g & ~3) * 64 + ((g ^ qq) & 3) * 64 + 32 * g16 + 8 * pp);
        }
    }
    auto read_frags = [&](unsigned stage_off, int s4, bf8 (&af)[5], bf8 (&bf)[2]) __attribute__((always_inline)) {
        if (GEMM_ABLATE & 4) return;
        const unsigned pb = b_base[s4] + stage_off, pa = a_base[s4] + stage_off;
        if (BT) {
#pragma unroll
            for (int j = 0; j < 2; j++) {
                s4v lo, hi;
                const unsigned p = bt_base[j] + stage_off + (unsigned)s4 * 8192u;
                asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(lo) : "v"(p));
                asm volatile("ds_read_b64_tr_b16 %0, %1 offset:2048" : "=v"(hi) : "v"(p));
                bf[j] = __builtin_bit_cast(bf8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
            }
        } else {
            asm volatile("ds_read_b128 %0, %1" : "=v"(bf[0]) : "v"(pb));
            asm volatile("ds_read_b128 %0, %1 offset:4096" : "=v"(bf[1]) : "v"(pb));
        }
        asm volatile("ds_read_b128 %0, %1" : "=v"(af[0]) : "v"(pa));
        asm volatile("ds_read_b128 %0, %1 offset:4096" : "=v"(af[1]) : "v"(pa));
        asm volatile("ds_read_b128 %0, %1 offset:8192" : "=v"(af[2]) : "v"(pa));
        asm volatile("ds_read_b128 %0, %1 offset:12288" : "=v"(af[3]) : "v"(pa));
        asm volatile("ds_read_b128 %0, %1 offset:16384" : "=v"(af[4]) : "v"(pa));
    };
    // operands swapped: the result has the output ROW on the lane and 16 columns in registers
    auto mma = [&](const bf8 (&af)[5], const bf8 (&bf)[2]) __attribute__((always_inline)) {
        if (GEMM_ABLATE & 2) {
#pragma unroll
            for (int i = 0; i < 5; i++) asm volatile("" ::"v"(af[i]));
#pragma unroll
            for (int j = 0; j < 2; j++) asm volatile("" ::"v"(bf[j]));
            return;
        }
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int i = 0; i < 5; i++)
#pragma unroll
            for (int j = 0; j < 2; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf[j], af[i], acc[i][j], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
    };
    auto frags_ready = [&]() __attribute__((always_inline)) {      // the OLDER of the two fragment sets in flight has landed
        if (BT) asm volatile("s_waitcnt lgkmcnt(9)" ::: "memory");       // 5 + 4 reads per fragment set
        else asm volatile("s_waitcnt lgkmcnt(7)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
    };

    // Software pipeline over 16-wide k-steps: the fragments of the next one are read while the 10 MFMAs of the current one run.
    // Two stages of K = 64: stage P + 1 lands while stage P is multiplied; once every wave has read the LAST fragments of stage
    // P (the barrier in front of its fourth k-step), stage P + 2 is requested into that slot:
    //   rd s1 | mma s0 | rd s2 | mma s1 | rd s3 | mma s2 | wait(P + 1 landed) barrier request(P + 2) | rd (P + 1) s0 | mma s3
    const int nk = sliced ? min(nk_slice, K / BK - ks0) : K / BK;
    request(0);
    if (1 < nk) {
        request(1);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PIECES) : "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    bf8 a0[5], b0[2], a1[5], b1[2];
    read_frags(0u, 0, a0, b0);
    for (int P = 0; P < nk; P++) {
        const unsigned so = (unsigned)(P % STAGES) * STAGEB;
        read_frags(so, 1, a1, b1);
        frags_ready();
        mma(a0, b0);
        read_frags(so, 2, a0, b0);
        frags_ready();
        mma(a1, b1);
        read_frags(so, 3, a1, b1);
        frags_ready();
        mma(a0, b0);
        if (P + 1 < nk) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // this wave's pieces of stage P + 1 (nothing younger is out)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // ... and it has left stage P
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            if (P + 2 < nk && !(GEMM_ABLATE & 1)) request(P + 2);
            read_frags((unsigned)((P + 1) % STAGES) * STAGEB, 0, a0, b0);
            if (BT) asm volatile("s_waitcnt lgkmcnt(9)" ::: "memory");
            else asm volatile("s_waitcnt lgkmcnt(7)" ::: "memory");
        } else {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_sched_barrier(0);
        mma(a1, b1);
    }

    __syncthreads();
    if (sliced) {
        // ---- a slice's epilogue: the fp32 partial tile [320 x 256] of (tail tile, slice), through a per-wave [32 x 64] fp32 patch
        // (8 KiB) so that the global stores are whole 256-byte row pieces
        float *tile = part + ((size_t)(item - full_items) * S + slice) * (size_t)(BM * BN);
        unsigned char *patch = smem + (size_t)wave * 8192;
#pragma unroll
        for (int i = 0; i < 5; i++) {
#pragma unroll
            for (int j = 0; j < 2; j++)
#pragma unroll
                for (int g = 0; g < 4; g++) {
                    const int n = j * 32 + 8 * g + 4 * hh;
                    *reinterpret_cast<float4 *>(patch + r * 256 + n * 4) =
                        make_float4(acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]);
                }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int p = 0; p < 8; p++) {
                const int pr = p * 4 + (lane >> 4), pc = lane & 15;   // 4 rows per pass, 16 x 16 B per row
                const float4 v = *reinterpret_cast<const float4 *>(patch + pr * 256 + pc * 16);
                *reinterpret_cast<float4 *>(tile + (size_t)(wm * 160 + i * 32 + pr) * BN + wn * 64 + pc * 4) = v;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        return;
    }
    // ---- epilogue: per wave and 32-row tile, accumulators -> bf16 [32 x 64] patch in LDS -> 128-byte row pieces
    // msk (optional, same shape as C): C[m][n] is multiplied by `slope` where msk[m][n] <= 0 -- the LeakyReLU' of the layer
    // whose OUTPUT msk is, applied to that layer's incoming gradient as this GEMM produces it (round 5: the GAT edge
    // backward then neither reads that output nor writes a masked copy of the gradient: 82 MB less per layer)
    unsigned char *patch = smem + (size_t)wave * 4096;            // 32 rows x 128 B, private to the wave
    // element (row m, column n) of acc[i][j]: lane m (+32: hh), register e -> n = (e & 3) + 8 (e >> 2) + 4 hh
    // The mask pieces of row tile i + 1 are requested before tile i is written out, UNCONDITIONALLY from a clamped row (a load
    // under `if (gm < M)` is a branch hipcc drains the load queue at: the first form of this epilogue was 20 dependent round
    // trips per workgroup).
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    const bool masked = msk != nullptr;
    const __bf16 *mbase = (masked ? msk : C) + n0 + wn * 64 + (lane & 7) * 8;
    const int mld = masked ? ldm : ldc;
    auto mask_row = [&](int i, int p) __attribute__((always_inline)) { return min(m0 + wm * 160 + i * 32 + p * 8 + (lane >> 3), M - 1); };
    u32x4 mkb[2][4];                                              // (two named sets, no copy: a copy is a use, i.e. a wait)
    if (masked) {
#pragma unroll
        for (int p = 0; p < 4; p++) mkb[0][p] = *reinterpret_cast<const u32x4 *>(mbase + (size_t)mask_row(0, p) * mld);
    }
#pragma unroll
    for (int i = 0; i < 5; i++) {
        if (masked && i + 1 < 5) {
#pragma unroll
            for (int p = 0; p < 4; p++) mkb[(i + 1) & 1][p] = *reinterpret_cast<const u32x4 *>(mbase + (size_t)mask_row(i + 1, p) * mld);
        }
        __builtin_amdgcn_sched_barrier(0);                        // (the scheduler sinks these loads to their use otherwise)
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int g = 0; g < 4; g++) {
                const int n = j * 32 + 8 * g + 4 * hh;
                *reinterpret_cast<uint2 *>(patch + r * 128 + n * 2) =
                    make_uint2(pack2(acc[i][j][4 * g], acc[i][j][4 * g + 1]), pack2(acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]));
            }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        // (same wave wrote the patch: no barrier needed)
#pragma unroll
        for (int p = 0; p < 4; p++) {
            const int pr = p * 8 + (lane >> 3), pc = lane & 7;    // 8 rows per pass, 8 x 16 B per row
            const int gm = m0 + wm * 160 + i * 32 + pr;
            u32x4 v = *reinterpret_cast<const u32x4 *>(patch + pr * 128 + pc * 16);
            if (masked) {
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    float lo = __uint_as_float(v[e] << 16), hi = __uint_as_float(v[e] & 0xffff0000u);
                    if (!(__uint_as_float(mkb[i & 1][p][e] << 16) > 0.f)) lo *= slope;
                    if (!(__uint_as_float(mkb[i & 1][p][e] & 0xffff0000u) > 0.f)) hi *= slope;
                    v[e] = pack2(lo, hi);
                }
            }
            if (gm < M) *reinterpret_cast<u32x4 *>(C + (size_t)gm * ldc + n0 + wn * 64 + pc * 8) = v;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
}

// the S fp32 partial tiles of every sliced tile, added in slice order (fixed order: bit-reproducible) and rounded once to bf16;
// one thread per 8 consecutive columns
__global__ __launch_bounds__(256) void k_gemm_tail_reduce(const float *__restrict__ part, int S, int ntiles, int first_mt,
                                                         __bf16 *__restrict__ C, int ldc, int M) {
    const int t = blockIdx.x / (BM / 8);                          // tail tile; 8 rows x 32 column groups per workgroup
    const int row = (blockIdx.x - t * (BM / 8)) * 8 + (threadIdx.x >> 5), cg = threadIdx.x & 31;
    const int mt = first_mt + t / ntiles, nt = t - (t / ntiles) * ntiles;
    const int gm = mt * BM + row;
    if (gm >= M) return;
    const float *src = part + (size_t)t * S * (size_t)(BM * BN) + (size_t)row * BN + cg * 8;
    float4 a = *reinterpret_cast<const float4 *>(src), b = *reinterpret_cast<const float4 *>(src + 4);
    for (int s_ = 1; s_ < S; s_++) {
        const float4 c = *reinterpret_cast<const float4 *>(src + (size_t)s_ * (BM * BN));
        const float4 d = *reinterpret_cast<const float4 *>(src + (size_t)s_ * (BM * BN) + 4);
        a.x += c.x; a.y += c.y; a.z += c.z; a.w += c.w;
        b.x += d.x; b.y += d.y; b.z += d.z; b.w += d.w;
    }
    *reinterpret_cast<uint4 *>(C + (size_t)gm * ldc + nt * BN + cg * 8) =
        make_uint4(pack2(a.x, a.y), pack2(a.z, a.w), pack2(b.x, b.y), pack2(b.z, b.w));
}

}  // namespace

template <bool BT>
static int launch_gemm(const void *A, int lda, const void *B, int ldb, void *C, int ldc, int M, int N, int K, void *stream,
                       const void *msk = nullptr, int ldm = 0, float slope = 1.f) {
    if (msk && (ldm < N || ldm % 8 || ((uintptr_t)msk & 15))) return -22;
    if (M <= 0 || N <= 0 || K <= 0 || N % BN != 0 || K % BK != 0 || lda < K || ldb < (BT ? N : K) || ldc < N) return -22;
    if (((uintptr_t)A & 15) || ((uintptr_t)B & 15) || ((uintptr_t)C & 15) || lda % 8 || ldb % 8 || ldc % 8) return -22;
    // the per-lane LDS-DMA source offsets are 32-bit (row * row stride in bytes): an operand image of 4 GiB or more would
    // wrap and read the wrong rows -- refused, the callers then use the library
    if ((size_t)M * lda * 2 >= ((size_t)1 << 32) || (size_t)(BT ? K : N) * ldb * 2 >= ((size_t)1 << 32)) return -22;
    static PerDeviceFlag attr_set;     
    if (!attr_set) {
        if (hipFuncSetAttribute((const void *)k_gemm_bf16<BT>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES) != hipSuccess) return -5;
        attr_set = true;
    }
    const int mtiles = (M + BM - 1) / BM, ntiles = N / BN;
    const unsigned grid = 8u * (unsigned)((mtiles * ntiles + 7) / 8);
    hipLaunchKernelGGL(k_gemm_bf16<BT>, dim3(grid), dim3(NT), LDS_BYTES, (hipStream_t)stream, (const __bf16 *)A, lda,
                       (const __bf16 *)B, ldb, (__bf16 *)C, ldc, M, N, K, mtiles, ntiles, mtiles * ntiles, 1, K / BK, (float *)nullptr,
                       (const __bf16 *)msk, ldm, slope);
    return hipGetLastError() == hipSuccess ? 0 : -5;
}

// floats of caller-owned workspace spadot_gemm_tn_bf16_split needs
extern "C" long long spadot_gemm_bf16_split_workspace(int M, int N, int tail_row_tiles, int slices) {
    if (M <= 0 || N <= 0 || N % BN != 0 || tail_row_tiles < 1 || slices < 2 || slices > 8) return -22;
    const int mtiles = (M + BM - 1) / BM;
    if (tail_row_tiles > mtiles) return -22;
    return (long long)tail_row_tiles * (N / BN) * slices * BM * BN;
}

// spadot_gemm_tn_bf16 with the LAST `tail_row_tiles` row panels (320 rows each) cut into `slices` slices of the contraction: the
// whole tiles come first in the grid, the slices behind them, then k_gemm_tail_reduce adds the fp32 partials (slice order)
// and rounds once.  For a GEMM of exactly one round of workgroups that has to share the chip: with 40 of the 256 compute
// units busy, 216 whole tiles + 160 quarter tiles take 1.25 tile times where 256 whole tiles take 2.
extern "C" int spadot_gemm_tn_bf16_split(const void *A, int lda, const void *B, int ldb, void *C, int ldc, int M, int N, int K,
                                         int tail_row_tiles, int slices, float *workspace, void *stream) {
    if (M <= 0 || N <= 0 || K <= 0 || N % BN != 0 || K % BK != 0 || lda < K || ldb < K || ldc < N) return -22;
    if (((uintptr_t)A & 15) || ((uintptr_t)B & 15) || ((uintptr_t)C & 15) || lda % 8 || ldb % 8 || ldc % 8) return -22;
    if ((size_t)M * lda * 2 >= ((size_t)1 << 32) || (size_t)N * ldb * 2 >= ((size_t)1 << 32)) return -22;
    const int mtiles = (M + BM - 1) / BM, ntiles = N / BN, nk = K / BK;
    if (tail_row_tiles < 1 || tail_row_tiles > mtiles || slices < 2 || slices > 8 || slices > nk) return -22;
    if (!workspace || ((uintptr_t)workspace & 15)) return -22;
    static PerDeviceFlag attr_set;     
    if (!attr_set) {
        if (hipFuncSetAttribute((const void *)k_gemm_bf16<false>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES) != hipSuccess) return -5;
        attr_set = true;
    }
    const int nk_slice = (nk + slices - 1) / slices;
    const int S = (nk + nk_slice - 1) / nk_slice;                  // no empty slice
    const int full_items = (mtiles - tail_row_tiles) * ntiles, tail_tiles = tail_row_tiles * ntiles;
    const unsigned grid = 8u * (unsigned)((full_items + 7) / 8) + (unsigned)(tail_tiles * S);
    hipLaunchKernelGGL(k_gemm_bf16<false>, dim3(grid), dim3(NT), LDS_BYTES, (hipStream_t)stream, (const __bf16 *)A, lda,
                       (const __bf16 *)B, ldb, (__bf16 *)C, ldc, M, N, K, mtiles, ntiles, full_items, S, nk_slice, workspace,
                       (const __bf16 *)nullptr, 0, 1.f);
    hipLaunchKernelGGL(k_gemm_tail_reduce, dim3((unsigned)(tail_tiles * (BM / 8))), dim3(256), 0, (hipStream_t)stream,
                       (const float *)workspace, S, ntiles, mtiles - tail_row_tiles, (__bf16 *)C, ldc, M);
    return hipGetLastError() == hipSuccess ? 0 : -5;
}

extern "C" int spadot_gemm_tn_bf16(const void *A, int lda, const void *B, int ldb, void *C, int ldc, int M, int N, int K,
                                   void *stream) {
    return launch_gemm<false>(A, lda, B, ldb, C, ldc, M, N, K, stream);
}

// C [M x N] = A [M x K] . B with B stored [K x N] (the input gradient of a dense map: dx = g W, W the [N_out x K_in] weight image)
extern "C" int spadot_gemm_nn_bf16(const void *A, int lda, const void *B, int ldb, void *C, int ldc, int M, int N, int K,
                                   void *stream) {
    return launch_gemm<true>(A, lda, B, ldb, C, ldc, M, N, K, stream);
}

// ... with C[m][n] multiplied by `slope` wherever act_out[m][n] <= 0 (act_out [M x N] bf16, row stride ldm): dx = (g W) * LeakyReLU'
// of the activation whose output act_out is -- the input gradient of a dense map handed to the layer below already masked
extern "C" int spadot_gemm_nn_bf16_masked(const void *A, int lda, const void *B, int ldb, void *C, int ldc, int M, int N, int K,
                                          const void *act_out, int ldm, double slope, void *stream) {
    if (!act_out) return -22;
    return launch_gemm<true>(A, lda, B, ldb, C, ldc, M, N, K, stream, act_out, ldm, (float)slope);
}
