// What the matrix cores of THIS box sustain on bf16: a bare v_mfma_f32_32x32x16_bf16 / 16x16x32 loop on register
// operands (random or zero data), one or two waves per SIMD, every CU busy.  The clock the chip holds under an
// MFMA-dense loop on random data is well below 2.4 GHz (MI355X_MICROARCH.md, DVFS give-back), so this -- not the
// 2.5 PFLOP/s datasheet figure -- is the ceiling a bf16 GEMM can approach here.  Build + run on the GPU box:
//   hipcc -O3 --offload-arch=gfx950 tools/mfma_peak.hip -o /tmp/mfma_peak && /tmp/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f4v __attribute__((ext_vector_type(4)));

template <int SHAPE>
__global__ __launch_bounds__(256) void k_loop(const unsigned short *src, float *sink, int iters) {
    bf8 a, b;
    for (int k = 0; k < 8; k++) {
        a[k] = __builtin_bit_cast(__bf16, src[(threadIdx.x * 8 + k) & 4095]);
        b[k] = __builtin_bit_cast(__bf16, src[(threadIdx.x * 8 + k + 1777) & 4095]);
    }
    if (SHAPE == 32) {
        f16v c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
        for (int i = 0; i < iters; i++) {
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c3, 0, 0, 0);
        }
        sink[blockIdx.x * 256 + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3];
    } else {
        f4v c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
        for (int i = 0; i < iters; i++) {
            c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c3, 0, 0, 0);
        }
        sink[blockIdx.x * 256 + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3];
    }
}

int main() {
    unsigned short *src; float *sink;
    hipMalloc(&src, 4096 * 2); hipMalloc(&sink, 4096 * 256 * 4);
    std::vector<unsigned short> h(4096);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rnd = 0; rnd < 2; rnd++) {
        for (int k = 0; k < 4096; k++) h[k] = rnd ? (unsigned short)(0x3c00 + (rand() & 0x3ff) + ((rand() & 1) << 15)) : 0;   // ~ +-[0.008, 0.03] or zeros
        hipMemcpy(src, h.data(), 8192, hipMemcpyHostToDevice);
        for (int shape = 32; shape >= 16; shape -= 16)
            for (int wg_per_cu = 1; wg_per_cu <= 2; wg_per_cu++) {
                const int grid = 256 * wg_per_cu, iters = 20000;
                const double flop = (double)grid * 4 /*waves*/ * iters * 4 /*mfma*/ * (shape == 32 ? 2.0 * 32 * 32 * 16 : 2.0 * 16 * 16 * 32);
                for (int rep = 0; rep < 3; rep++) {
                    hipEventRecord(e0);
                    if (shape == 32) hipLaunchKernelGGL(k_loop<32>, dim3(grid), dim3(256), 0, 0, src, sink, iters);
                    else hipLaunchKernelGGL(k_loop<16>, dim3(grid), dim3(256), 0, 0, src, sink, iters);
                    hipEventRecord(e1); hipEventSynchronize(e1);
                    float ms; hipEventElapsedTime(&ms, e0, e1);
                    if (rep == 2)
                        printf("%s data  mfma %dx%dx%d  %d wave(s)/SIMD  %.2f ms  %.0f TFLOP/s\n", rnd ? "random" : "zero  ", shape, shape,
                               shape == 32 ? 16 : 32, wg_per_cu, ms, flop / ms / 1e9);
                }
            }
    }
    return 0;
}
