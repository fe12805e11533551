// gemm_wgrad_bf16.hip -- dW [N x K] (fp32) = G^T X, G [M x N] and X [M x K] bf16 row-major: the weight gradient of a GAT
// layer's dense map (/root/reference/SpaDOT/model/encoder.py:41-58, GATConv.lin; autograd's d/dW of x W^T) on gfx950.
//
// A small output (2048 x 2048 or 2048 x 3000) over a long contraction (M ~ 10^4 rows): the library runs it with 128 x 128
// tiles at 0.37-0.63 PFLOP/s (each workgroup walks all M rows at 64 flop per staged byte), and these GEMMs are fully exposed
// on the serial GAT backward chain (skipping two of them: -250 us of a 2.0 ms step).  Here:
//   * 256 x 256 output tiles x S slices of the contraction = the grid (S = 4 at 2048 x 2048: 256 workgroups); a slice is a
//     contiguous range of 64-row chunks;
//   * both operands have the contraction index as their ROW index, so both MFMA fragments are transposed reads
//     (ds_read_b64_tr_b16, two per fragment) of row-major LDS images [64 rows][256 columns] whose 64-byte granules are XORed
//     with (row & 3): the four rows a read touches fall on different banks although the row stride is 512 B, and rows stay
//     contiguous, so one LDS-DMA instruction (global_load_lds_dwordx4, 1 KiB) still fills two whole rows -- the XOR is applied
//     to the SOURCE address;
//   * 512 threads = 8 waves as 2 (n) x 4 (k), wave tile 128 x 64 = 4 x 2 MFMA tiles (128 accumulator VGPRs); two 64 KiB
//     stages; software pipeline over 16-row sub-steps (12 fragment reads in flight beside 8 MFMAs), one s_barrier per stage;
//   * rows past M come from a zero row (they must contribute nothing);
//   * the accumulators leave as 128-byte row segments (the k index is on the lanes).  S > 1: every workgroup writes its
//     partial tile and a second launch (k_wgrad_reduce) adds the partials in slice order: a fixed order, no atomics.
//   * a second instantiation with 256 x 192 output tiles (waves as 4 (n) x 2 (k), wave tile 64 x 96 = 2 x 3 MFMA tiles) for
//     shapes the square tile leaves the chip a quarter empty on: 2048 x 3072 is 96 square tiles (x 2 slices = 192 workgroups)
//     but 128 of these (x 2 = 256: one round).  Same staging (the X image keeps 512-byte rows, the 64 columns behind the
//     tile come from the zero row), same pipeline.
#include <hip/hip_runtime.h>
#include <cstdint>

#include "../../include/spadot_model.h"
#include "per_device.h"

namespace {

typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef short s4v __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));

constexpr int TN = 256, TK = 256, CH = 64, NT = 512;
constexpr int ROWB = 512;                               // staged row: 256 bf16
constexpr int OPB = CH * ROWB;                          // one operand's chunk: 32 KiB
constexpr int STAGEB = 2 * OPB;                         // G chunk | X chunk
constexpr int LDS_BYTES = 2 * STAGEB;                   // two stages: 128 KiB
constexpr int PIECES = 8;                               // LDS-DMA instructions per wave and stage (64 in all, 2 rows each)

// WGN x WGK waves, TI x TJ MFMA tiles (32 x 32) per wave: WGN * TI = 8 (256 rows of dW), tile width TKE = 32 WGK TJ columns
template <int WGN, int WGK, int TI, int TJ>
__global__ __launch_bounds__(NT, 1) void k_gemm_wgrad_bf16(const __bf16 *__restrict__ G, int ldg, const __bf16 *__restrict__ X,
                                                          int ldx, float *__restrict__ dW, int ldw, int M, int N, int K,
                                                          int ktiles, int S, int chunks_per_slice,
                                                          const __bf16 *__restrict__ zrow, float *__restrict__ part) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tile = blockIdx.x / S, slice = blockIdx.x - tile * S;
    const int tn = tile / ktiles, tk = tile - tn * ktiles;
    static_assert(WGN * WGK == 8 && WGN * TI == 8 && WGK * TJ * 32 <= TK, "wave grid");
    constexpr int TKE = 32 * WGK * TJ;                             // columns of dW a tile owns (256 or 192)
    constexpr int RD = 2 * (TI + TJ);                              // LDS reads of one fragment set
    const int n0 = tn * TN, k0 = tk * TKE;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave / WGK, wk = wave - wn * WGK;               // wave grid WGN (n) x WGK (k)
    const unsigned lds0 = (unsigned)(size_t)smem;
    const int nchunk_all = (M + CH - 1) / CH;
    const int c_begin = slice * chunks_per_slice, c_end = min(nchunk_all, c_begin + chunks_per_slice);
    const int nk = max(0, c_end - c_begin);

    // ---- LDS-DMA: piece q (0..63) of a stage = staged rows 2 q', 2 q' + 1 of G (q < 32) or X (q >= 32); lane i -> row
    // 2 q' + (i >> 5), 16-byte chunk p = i & 31 of the 512-byte row, which receives SOURCE chunk p ^ ((row & 3) << 2).
    // This wave's pieces: q = wave + 8 u.  Per lane a byte offset inside the operand (the chunk's first row is added per stage).
    size_t poff[PIECES];
    int prow[PIECES];
    bool pzero = false;                                            // X columns behind a narrow tile: zeros (never read)
#pragma unroll
    for (int u = 0; u < PIECES; u++) {
        const int q = wave + 8 * u, qq = q & 31;
        const int row = 2 * qq + (lane >> 5);
        const int chunk = (lane & 31) ^ ((row & 3) << 2);
        prow[u] = row;
        poff[u] = u < 4 ? ((size_t)row * ldg + n0) * 2 + chunk * 16 : ((size_t)row * ldx + k0) * 2 + chunk * 16;
        if (u >= 4 && chunk * 8 >= TKE) pzero = true;              // (row & 3 is the same for every piece of a lane)
    }
    auto request = [&](int ks) __attribute__((always_inline)) {
#ifdef WGRAD_PROBE_NO_DMA                               // timing probe only (wrong results): nothing staged
        return;
#endif
        const int m0 = (c_begin + ks) * CH;
        const unsigned base = lds0 + (unsigned)(ks & 1) * STAGEB + (unsigned)wave * 1024u;
#pragma unroll
        for (int u = 0; u < PIECES; u++) {
            const char *src = (u < 4 ? reinterpret_cast<const char *>(G) + (size_t)m0 * ldg * 2
                                     : reinterpret_cast<const char *>(X) + (size_t)m0 * ldx * 2) + poff[u];
            if (m0 + prow[u] >= M || (u >= 4 && pzero)) src = reinterpret_cast<const char *>(zrow) + (lane & 31) * 16;   // rows past M: zeros
            const unsigned dst = __builtin_amdgcn_readfirstlane(base + (unsigned)(u & 3) * 8192u + (u < 4 ? 0u : (unsigned)OPB));
            unsigned keep;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(src), "s"(dst) : "memory");
        }
    };

    f16v acc[TI][TJ];
#pragma unroll
    for (int i = 0; i < TI; i++)
#pragma unroll
        for (int j = 0; j < TJ; j++)
#pragma unroll
            for (int e = 0; e < 16; e++) acc[i][j][e] = 0.f;

    // ---- transposed fragment reads.  Lane: hh = lane >> 5 (contraction rows 8 hh .. 8 hh + 7 of the 16-row sub-step),
    // g16 = (lane >> 4) & 1 and pp = lane & 3 (columns 16 g16 + 4 pp .. + 3 of the 32-column tile), qq = (lane & 15) >> 2 (row
    // 8 hh + qq, and + 4 for the second read).  Column byte offset of tile t: 64 t' + 32 g16 + 8 pp with its 64-byte granule
    // index XORed with (row & 3) = qq.
    const int hh = lane >> 5, g16 = (lane >> 4) & 1, qq = (lane & 15) >> 2, pp = lane & 3;
    unsigned ga[TI], xb[TJ];
#pragma unroll
    for (int i = 0; i < TI; i++) {
        const int g = wn * TI + i;                                   // 64-byte granule (32 columns) of the G image
        ga[i] = lds0 + (unsigned)((8 * hh + qq) * ROWB + (g & ~3) * 64 + ((g ^ qq) & 3) * 64 + 32 * g16 + 8 * pp);
    }
#pragma unroll
    for (int j = 0; j < TJ; j++) {
        const int g = wk * TJ + j;                                   // granule of the X image
        xb[j] = lds0 + (unsigned)OPB + (unsigned)((8 * hh + qq) * ROWB + (g & ~3) * 64 + ((g ^ qq) & 3) * 64 + 32 * g16 + 8 * pp);
    }
    auto read_frags = [&](unsigned stage_off, int ss, s4v (&a)[TI][2], s4v (&b)[TJ][2]) __attribute__((always_inline)) {
        const unsigned so = stage_off + (unsigned)ss * (16 * ROWB);
#ifdef WGRAD_PROBE_NO_READS                             // timing probe only (wrong results): no LDS reads
#pragma unroll
        for (int j = 0; j < TJ; j++) b[j][0] = b[j][1] = s4v{(short)so, 1, 2, 3};
#pragma unroll
        for (int i = 0; i < TI; i++) a[i][0] = a[i][1] = s4v{(short)so, 1, 2, 3};
        return;
#endif
#pragma unroll
        for (int j = 0; j < TJ; j++) {
            asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(b[j][0]) : "v"(xb[j] + so));
            asm volatile("ds_read_b64_tr_b16 %0, %1 offset:2048" : "=v"(b[j][1]) : "v"(xb[j] + so));
        }
#pragma unroll
        for (int i = 0; i < TI; i++) {
            asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(a[i][0]) : "v"(ga[i] + so));
            asm volatile("ds_read_b64_tr_b16 %0, %1 offset:2048" : "=v"(a[i][1]) : "v"(ga[i] + so));
        }
    };
    auto mma = [&](const s4v (&a)[TI][2], const s4v (&b)[TJ][2]) __attribute__((always_inline)) {
        __builtin_amdgcn_s_setprio(1);
        bf8 bf[TJ];
#pragma unroll
        for (int j = 0; j < TJ; j++) bf[j] = __builtin_bit_cast(bf8, __builtin_shufflevector(b[j][0], b[j][1], 0, 1, 2, 3, 4, 5, 6, 7));
#pragma unroll
        for (int i = 0; i < TI; i++) {
            const bf8 af = __builtin_bit_cast(bf8, __builtin_shufflevector(a[i][0], a[i][1], 0, 1, 2, 3, 4, 5, 6, 7));
#pragma unroll
            for (int j = 0; j < TJ; j++) {
#ifdef WGRAD_PROBE_NO_MFMA                              // timing probe only (wrong results): the fragments stay live, no matrix op
                asm volatile("" ::"v"(af), "v"(bf[j]));
#else
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bf[j], acc[i][j], 0, 0, 0);
#endif
            }
        }
        __builtin_amdgcn_s_setprio(0);
    };
    auto frags_ready = [&]() __attribute__((always_inline)) {      // the OLDER of the two fragment sets in flight has landed
#ifdef WGRAD_PROBE_NO_READS
        return;
#endif
        asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(RD) : "memory");
        __builtin_amdgcn_sched_barrier(0);
    };

    // ---- main loop (same schedule as csrc/gemm_bf16.hip): rd s1 | mma s0 | rd s2 | mma s1 | rd s3 | mma s2 |
    //      wait(stage P + 1 landed) barrier request(P + 2) | rd (P + 1) s0 | mma s3
    if (nk > 0) {
        request(0);
        if (1 < nk) {
            request(1);
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PIECES) : "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
        s4v a0[TI][2], b0[TJ][2], a1[TI][2], b1[TJ][2];
        read_frags(0u, 0, a0, b0);
        for (int P = 0; P < nk; P++) {
            const unsigned so = (unsigned)(P & 1) * STAGEB;
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
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_barrier();
                if (P + 2 < nk) request(P + 2);
                read_frags((unsigned)((P + 1) & 1) * STAGEB, 0, a0, b0);
                asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(RD) : "memory");
            } else {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
            __builtin_amdgcn_sched_barrier(0);
            mma(a1, b1);
        }
    }

    // ---- epilogue.  acc[i][j][e]: row n = n0 + 32 (wn TI + i) + (e & 3) + 8 (e >> 2) + 4 hh, column k = k0 + 32 (wk TJ + j) + (lane & 31)
    const int lk = (wk * TJ) * 32 + (lane & 31);
    if (S == 1) {
#pragma unroll
        for (int i = 0; i < TI; i++)
#pragma unroll
            for (int j = 0; j < TJ; j++)
#pragma unroll
                for (int e = 0; e < 16; e++) {
                    const int n = n0 + (wn * TI + i) * 32 + (e & 3) + 8 * (e >> 2) + 4 * hh, k = k0 + lk + 32 * j;
                    if (n < N && k < K) dW[(size_t)n * ldw + k] = acc[i][j][e];
                }
        return;
    }
    // partial tile of this slice: part[slice][tile][256][TKE]
    float *mine = part + ((size_t)slice * gridDim.x / S + tile) * (size_t)(TN * TKE);
#pragma unroll
    for (int i = 0; i < TI; i++)
#pragma unroll
        for (int j = 0; j < TJ; j++)
#pragma unroll
            for (int e = 0; e < 16; e++) {
                const int ln = (wn * TI + i) * 32 + (e & 3) + 8 * (e >> 2) + 4 * hh;
                mine[(size_t)ln * TKE + lk + 32 * j] = acc[i][j][e];
            }
}

// S > 1: dW = sum of the S partial tiles in slice order (a fixed order; no atomics, no hand-over between workgroups -- the
// last-arriver form of this sum, one workgroup per tile re-reading three partial tiles behind a device-scope release, took
// 226 us against 83 us for the GEMM itself).  One 16-byte piece per thread, S independent loads.
template <int TKE>
__global__ __launch_bounds__(256) void k_wgrad_reduce(const float *__restrict__ part, int S, int tiles, int ktiles,
                                                      float *__restrict__ dW, int ldw, int N, int K) {
    // FOUR 16-byte pieces per thread (pieces p, p + Q, p + 2 Q, p + 3 Q of the tile, Q = a quarter of its 256 RP pieces), so 4 S
    // independent loads are in flight per thread: with one piece per thread (until round 5) the 256 x 192 form took 42 us for
    // 52 MB of partials + 16 MB of dW -- 1.6 TB/s -- at the very end of the step's critical chain.
    constexpr int RP = TKE / 4;                                     // 16-byte pieces per tile row
    constexpr int BPT = RP / 4;                                     // 256-thread blocks per tile
    constexpr int Q = 64 * RP;                                      // a quarter of the tile's pieces
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    const int tile = blockIdx.x / BPT;
    const int p0 = (blockIdx.x - tile * BPT) * 256 + threadIdx.x;
    const int tn = tile / ktiles, tk = tile - tn * ktiles;
    f32x4 v[8][4];
#pragma unroll
    for (int s_ = 0; s_ < 8; s_++)
        if (s_ < S) {                                               // (uniform)
            const f32x4 *src = reinterpret_cast<const f32x4 *>(part + ((size_t)s_ * tiles + tile) * (size_t)(TN * TKE));
#pragma unroll
            for (int j = 0; j < 4; j++) v[s_][j] = src[p0 + j * Q];
        }
#pragma unroll
    for (int j = 0; j < 4; j++) {
        f32x4 sum = v[0][j];
#pragma unroll
        for (int s_ = 1; s_ < 8; s_++)
            if (s_ < S) sum += v[s_][j];
        const int idx4 = p0 + j * Q, ln = idx4 / RP, c4 = idx4 - ln * RP;
        const int n = tn * TN + ln, k = tk * TKE + 4 * c4;
        if (n < N && k + 3 < K) *reinterpret_cast<f32x4 *>(dW + (size_t)n * ldw + k) = sum;
        else if (n < N) {
            for (int e = 0; e < 4; e++)
                if (k + e < K) dW[(size_t)n * ldw + k + e] = sum[e];
        }
    }
}

// slices actually used for an (M, slices) request: at most one slice per 64-row chunk, at most 8, none empty
static int effective_slices(int M, int slices, int *cps_out) {
    const int nchunk = (M + CH - 1) / CH;
    int S = slices > 0 ? slices : 1;
    if (S > nchunk) S = nchunk;
    if (S > 8) S = 8;
    const int cps = (nchunk + S - 1) / S;
    S = (nchunk + cps - 1) / cps;
    if (cps_out) *cps_out = cps;
    return S;
}

}  // namespace

// floats of caller-owned workspace a call with these arguments needs (0: a single slice writes dW directly)
extern "C" long long spadot_gemm_wgrad_bf16_workspace_tiled(int M, int N, int K, int slices, int tile_k) {
    if (M <= 0 || N <= 0 || K <= 0 || N % TN != 0 || (tile_k != 256 && tile_k != 192)) return -22;
    const int S = effective_slices(M, slices, nullptr);
    const long long tiles = (long long)(N / TN) * ((K + tile_k - 1) / tile_k);
    return S > 1 ? (long long)S * tiles * TN * tile_k : 0;
}

extern "C" long long spadot_gemm_wgrad_bf16_workspace(int M, int N, int K, int slices) {
    return spadot_gemm_wgrad_bf16_workspace_tiled(M, N, K, slices, 256);
}

template <int WGN, int WGK, int TI, int TJ>
static int launch_wgrad(const void *G, int ldg, const void *X, int ldx, float *dW, int ldw, int M, int N, int K, int ktiles, int ntiles,
                        int S, int cps, float *workspace, const void *zero_row, hipStream_t stream) {
    constexpr int TKE = 32 * WGK * TJ;
    static PerDeviceFlag attr_set;     
    auto kern = k_gemm_wgrad_bf16<WGN, WGK, TI, TJ>;
    if (!attr_set) {
        if (hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES) != hipSuccess) return -5;
        attr_set = true;
    }
    const int tiles = ntiles * ktiles;
    hipLaunchKernelGGL(kern, dim3((unsigned)(tiles * S)), dim3(NT), LDS_BYTES, stream, (const __bf16 *)G, ldg, (const __bf16 *)X, ldx, dW,
                       ldw, M, N, K, ktiles, S, cps, (const __bf16 *)zero_row, workspace);
    if (S > 1)
        hipLaunchKernelGGL(k_wgrad_reduce<TKE>, dim3((unsigned)tiles * (unsigned)(TKE / 16)), dim3(256), 0, stream, (const float *)workspace, S,
                           tiles, ktiles, dW, ldw, N, K);
    return hipGetLastError() == hipSuccess ? 0 : -5;
}

// tile_k: 256 (square tiles) or 192 (256 x 192 tiles: for shapes whose square-tile grid leaves the chip partly empty)
extern "C" int spadot_gemm_wgrad_bf16_tiled(const void *G, int ldg, const void *X, int ldx, float *dW, int ldw, int M, int N, int K,
                                            int slices, int tile_k, float *workspace, const void *zero_row, void *stream) {
    if (M <= 0 || N <= 0 || K <= 0 || N % TN != 0 || ldg < N || ldx < K || ldw < K || ldg % 8 || ldx % 8) return -22;
    if (tile_k != 256 && tile_k != 192) return -22;
    if (((uintptr_t)G & 15) || ((uintptr_t)X & 15) || ((uintptr_t)dW & 15) || (ldw % 4)) return -22;
    if (!zero_row || ((uintptr_t)zero_row & 15)) return -22;
    // the X image reads whole tiles: the row stride must cover the last (partial) tile
    const int ktiles = (K + tile_k - 1) / tile_k, ntiles = N / TN;
    if (ldx < ktiles * tile_k) return -22;
    // the LDS-DMA source offsets are 32-bit: an operand image must stay below 4 GiB
    if ((size_t)M * ldg * 2 >= ((size_t)1 << 32) || (size_t)M * ldx * 2 >= ((size_t)1 << 32)) return -22;
    int cps = 0;
    const int S = effective_slices(M, slices, &cps);
    if (ntiles * ktiles > 4096) return -22;
    if (S > 1 && (!workspace || ((uintptr_t)workspace & 15))) return -22;
    if (tile_k == 256)
        return launch_wgrad<2, 4, 4, 2>(G, ldg, X, ldx, dW, ldw, M, N, K, ktiles, ntiles, S, cps, workspace, zero_row, (hipStream_t)stream);
    return launch_wgrad<4, 2, 2, 3>(G, ldg, X, ldx, dW, ldw, M, N, K, ktiles, ntiles, S, cps, workspace, zero_row, (hipStream_t)stream);
}

extern "C" int spadot_gemm_wgrad_bf16(const void *G, int ldg, const void *X, int ldx, float *dW, int ldw, int M, int N, int K,
                                      int slices, float *workspace, const void *zero_row, void *stream) {
    return spadot_gemm_wgrad_bf16_tiled(G, ldg, X, ldx, dW, ldw, M, N, K, slices, 256, workspace, zero_row, stream);
}
