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
#include <hip/hip_runtime.h>
#include <cstdint>

#include "../../include/spadot_model.h"

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

__global__ __launch_bounds__(NT, 1) void k_gemm_wgrad_bf16(const __bf16 *__restrict__ G, int ldg, const __bf16 *__restrict__ X,
                                                          int ldx, float *__restrict__ dW, int ldw, int M, int N, int K,
                                                          int ktiles, int S, int chunks_per_slice,
                                                          const __bf16 *__restrict__ zrow, float *__restrict__ part) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tile = blockIdx.x / S, slice = blockIdx.x - tile * S;
    const int tn = tile / ktiles, tk = tile - tn * ktiles;
    const int n0 = tn * TN, k0 = tk * TK;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave >> 2, wk = wave & 3;                       // wave grid 2 (n) x 4 (k)
    const unsigned lds0 = (unsigned)(size_t)smem;
    const int nchunk_all = (M + CH - 1) / CH;
    const int c_begin = slice * chunks_per_slice, c_end = min(nchunk_all, c_begin + chunks_per_slice);
    const int nk = max(0, c_end - c_begin);

    // ---- LDS-DMA: piece q (0..63) of a stage = staged rows 2 q', 2 q' + 1 of G (q < 32) or X (q >= 32); lane i -> row
    // 2 q' + (i >> 5), 16-byte chunk p = i & 31 of the 512-byte row, which receives SOURCE chunk p ^ ((row & 3) << 2).
    // This wave's pieces: q = wave + 8 u.  Per lane a byte offset inside the operand (the chunk's first row is added per stage).
    size_t poff[PIECES];
    int prow[PIECES];
#pragma unroll
    for (int u = 0; u < PIECES; u++) {
        const int q = wave + 8 * u, qq = q & 31;
        const int row = 2 * qq + (lane >> 5);
        const int chunk = (lane & 31) ^ ((row & 3) << 2);
        prow[u] = row;
        poff[u] = u < 4 ? ((size_t)row * ldg + n0) * 2 + chunk * 16 : ((size_t)row * ldx + k0) * 2 + chunk * 16;
    }
    auto request = [&](int ks) __attribute__((always_inline)) {
        const int m0 = (c_begin + ks) * CH;
        const unsigned base = lds0 + (unsigned)(ks & 1) * STAGEB + (unsigned)wave * 1024u;
#pragma unroll
        for (int u = 0; u < PIECES; u++) {
            const char *src = (u < 4 ? reinterpret_cast<const char *>(G) + (size_t)m0 * ldg * 2
                                     : reinterpret_cast<const char *>(X) + (size_t)m0 * ldx * 2) + poff[u];
            if (m0 + prow[u] >= M) src = reinterpret_cast<const char *>(zrow) + (lane & 31) * 16;      // rows past M: zeros
            const unsigned dst = __builtin_amdgcn_readfirstlane(base + (unsigned)(u & 3) * 8192u + (u < 4 ? 0u : (unsigned)OPB));
            unsigned keep;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(src), "s"(dst) : "memory");
        }
    };

    f16v acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int e = 0; e < 16; e++) acc[i][j][e] = 0.f;

    // ---- transposed fragment reads.  Lane: hh = lane >> 5 (contraction rows 8 hh .. 8 hh + 7 of the 16-row sub-step),
    // g16 = (lane >> 4) & 1 and pp = lane & 3 (columns 16 g16 + 4 pp .. + 3 of the 32-column tile), qq = (lane & 15) >> 2 (row
    // 8 hh + qq, and + 4 for the second read).  Column byte offset of tile t: 64 t' + 32 g16 + 8 pp with its 64-byte granule
    // index XORed with (row & 3) = qq.
    const int hh = lane >> 5, g16 = (lane >> 4) & 1, qq = (lane & 15) >> 2, pp = lane & 3;
    unsigned ga[4], xb[2];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int gran = (wn * 4 + i) ^ qq;                          // (256 wn + 64 i) / 64, low two bits ^ qq
        ga[i] = lds0 + (unsigned)((8 * hh + qq) * ROWB + ((wn * 4 + i) & ~3) * 64 + (gran & 3) * 64 + 32 * g16 + 8 * pp);
    }
#pragma unroll
    for (int j = 0; j < 2; j++) {
        const int g = wk * 2 + j;                                    // (128 wk + 64 j) / 64
        xb[j] = lds0 + (unsigned)OPB + (unsigned)((8 * hh + qq) * ROWB + (g & ~3) * 64 + ((g ^ qq) & 3) * 64 + 32 * g16 + 8 * pp);
    }
    auto read_frags = [&](unsigned stage_off, int ss, s4v (&a)[4][2], s4v (&b)[2][2]) __attribute__((always_inline)) {
        const unsigned so = stage_off + (unsigned)ss * (16 * ROWB);
#pragma unroll
        for (int j = 0; j < 2; j++) {
            asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(b[j][0]) : "v"(xb[j] + so));
            asm volatile("ds_read_b64_tr_b16 %0, %1 offset:2048" : "=v"(b[j][1]) : "v"(xb[j] + so));
        }
#pragma unroll
        for (int i = 0; i < 4; i++) {
            asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(a[i][0]) : "v"(ga[i] + so));
            asm volatile("ds_read_b64_tr_b16 %0, %1 offset:2048" : "=v"(a[i][1]) : "v"(ga[i] + so));
        }
    };
    auto mma = [&](const s4v (&a)[4][2], const s4v (&b)[2][2]) __attribute__((always_inline)) {
        __builtin_amdgcn_s_setprio(1);
        bf8 bf[2];
#pragma unroll
        for (int j = 0; j < 2; j++) bf[j] = __builtin_bit_cast(bf8, __builtin_shufflevector(b[j][0], b[j][1], 0, 1, 2, 3, 4, 5, 6, 7));
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const bf8 af = __builtin_bit_cast(bf8, __builtin_shufflevector(a[i][0], a[i][1], 0, 1, 2, 3, 4, 5, 6, 7));
#pragma unroll
            for (int j = 0; j < 2; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bf[j], acc[i][j], 0, 0, 0);
        }
        __builtin_amdgcn_s_setprio(0);
    };
    auto frags_ready = [&]() __attribute__((always_inline)) {      // the OLDER of the two fragment sets in flight has landed
        asm volatile("s_waitcnt lgkmcnt(12)" ::: "memory");
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
        s4v a0[4][2], b0[2][2], a1[4][2], b1[2][2];
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
                asm volatile("s_waitcnt lgkmcnt(12)" ::: "memory");
            } else {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
            __builtin_amdgcn_sched_barrier(0);
            mma(a1, b1);
        }
    }

    // ---- epilogue.  acc[i][j][e]: row n = n0 + 128 wn + 32 i + (e & 3) + 8 (e >> 2) + 4 hh, column k = k0 + 64 wk + 32 j + (lane & 31)
    const int kcol = k0 + wk * 64 + (lane & 31);
    auto row_of = [&](int i, int e) { return n0 + wn * 128 + 32 * i + (e & 3) + 8 * (e >> 2) + 4 * hh; };
    if (S == 1) {
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
            for (int j = 0; j < 2; j++)
#pragma unroll
                for (int e = 0; e < 16; e++) {
                    const int n = row_of(i, e), k = kcol + 32 * j;
                    if (n < N && k < K) dW[(size_t)n * ldw + k] = acc[i][j][e];
                }
        return;
    }
    // partial tile of this slice: part[slice][tile][256][256]
    float *mine = part + ((size_t)slice * gridDim.x / S + tile) * (size_t)(TN * TK);
    const int lk = wk * 64 + (lane & 31);
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int e = 0; e < 16; e++) {
                const int ln = wn * 128 + 32 * i + (e & 3) + 8 * (e >> 2) + 4 * hh;
                mine[(size_t)ln * TK + lk + 32 * j] = acc[i][j][e];
            }
}

// S > 1: dW = sum of the S partial tiles in slice order (a fixed order; no atomics, no hand-over between workgroups -- the
// last-arriver form of this sum, one workgroup per tile re-reading three partial tiles behind a device-scope release, took
// 226 us against 83 us for the GEMM itself).  One 16-byte piece per thread, S independent loads.
__global__ __launch_bounds__(256) void k_wgrad_reduce(const float *__restrict__ part, int S, int tiles, int ktiles,
                                                      float *__restrict__ dW, int ldw, int N, int K) {
    const int tile = blockIdx.x >> 6;
    const int idx4 = ((blockIdx.x & 63) << 8) + threadIdx.x;        // 16384 pieces per 256 x 256 tile
    const int ln = idx4 >> 6, c4 = idx4 & 63;
    const int tn = tile / ktiles, tk = tile - tn * ktiles;
    const int n = tn * TN + ln, k = tk * TK + 4 * c4;
    float4 v[8];
#pragma unroll
    for (int s_ = 0; s_ < 8; s_++)
        v[s_] = s_ < S ? reinterpret_cast<const float4 *>(part + ((size_t)s_ * tiles + tile) * (size_t)(TN * TK))[idx4]
                       : make_float4(0.f, 0.f, 0.f, 0.f);
    float4 sum = v[0];
#pragma unroll
    for (int s_ = 1; s_ < 8; s_++)
        if (s_ < S) { sum.x += v[s_].x; sum.y += v[s_].y; sum.z += v[s_].z; sum.w += v[s_].w; }
    if (n < N && k + 3 < K) *reinterpret_cast<float4 *>(dW + (size_t)n * ldw + k) = sum;
    else if (n < N) {
        const float t[4] = {sum.x, sum.y, sum.z, sum.w};
        for (int e = 0; e < 4; e++)
            if (k + e < K) dW[(size_t)n * ldw + k + e] = t[e];
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
extern "C" long long spadot_gemm_wgrad_bf16_workspace(int M, int N, int K, int slices) {
    if (M <= 0 || N <= 0 || K <= 0 || N % TN != 0) return -22;
    const int S = effective_slices(M, slices, nullptr);
    const long long tiles = (long long)(N / TN) * ((K + TK - 1) / TK);
    return S > 1 ? (long long)S * tiles * TN * TK : 0;
}

extern "C" int spadot_gemm_wgrad_bf16(const void *G, int ldg, const void *X, int ldx, float *dW, int ldw, int M, int N, int K,
                                      int slices, float *workspace, const void *zero_row, void *stream) {
    if (M <= 0 || N <= 0 || K <= 0 || N % TN != 0 || ldg < N || ldx < K || ldw < K || ldg % 8 || ldx % 8) return -22;
    if (((uintptr_t)G & 15) || ((uintptr_t)X & 15) || ((uintptr_t)dW & 15) || (ldw % 4)) return -22;
    if (!zero_row || ((uintptr_t)zero_row & 15)) return -22;
    // the X image reads whole 256-column tiles: the row stride must cover the last (partial) tile
    const int ktiles = (K + TK - 1) / TK, ntiles = N / TN;
    if (ldx < ktiles * TK) return -22;
    // the LDS-DMA source offsets are 32-bit: an operand image must stay below 4 GiB
    if ((size_t)M * ldg * 2 >= ((size_t)1 << 32) || (size_t)M * ldx * 2 >= ((size_t)1 << 32)) return -22;
    int cps = 0;
    const int S = effective_slices(M, slices, &cps);
    const int tiles = ntiles * ktiles;
    if (tiles > 4096) return -22;
    if (S > 1 && (!workspace || ((uintptr_t)workspace & 15))) return -22;
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute((const void *)k_gemm_wgrad_bf16, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES) != hipSuccess) return -5;
        attr_set = true;
    }
    hipLaunchKernelGGL(k_gemm_wgrad_bf16, dim3((unsigned)(tiles * S)), dim3(NT), LDS_BYTES, (hipStream_t)stream, (const __bf16 *)G, ldg,
                       (const __bf16 *)X, ldx, dW, ldw, M, N, K, ktiles, S, cps, (const __bf16 *)zero_row, workspace);
    if (S > 1)
        hipLaunchKernelGGL(k_wgrad_reduce, dim3((unsigned)tiles * 64u), dim3(256), 0, (hipStream_t)stream, (const float *)workspace, S,
                           tiles, ktiles, dW, ldw, N, K);
    return hipGetLastError() == hipSuccess ? 0 : -5;
}
