// Layout probe for v_mfma_f64_16x16x4_f64 on gfx950: prints which (row, col) of D each lane/register holds and
// checks D = A B with A[i][k] = lane-indexed guess.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double double4_t __attribute__((ext_vector_type(4)));
__global__ void probe(const double *A, const double *B, double *D) {   // A [16][4], B [4][16], D [16][16] row-major
    const int l = threadIdx.x;
    const double a = A[(l % 16) * 4 + (l / 16)];      // guess: row = l%16, k = l/16
    const double b = B[(l / 16) * 16 + (l % 16)];     // guess: k = l/16, col = l%16
    double4_t c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
    for (int r = 0; r < 4; r++) D[l * 4 + r] = c[r];  // raw dump: lane-major
}
int main() {
    double hA[64], hB[64], hD[256], ref[256];
    for (int i = 0; i < 64; i++) { hA[i] = 1 + i * 0.5; hB[i] = 2 - i * 0.25; }
    for (int i = 0; i < 16; i++) for (int j = 0; j < 16; j++) { double s = 0; for (int k = 0; k < 4; k++) s += hA[i * 4 + k] * hB[k * 16 + j]; ref[i * 16 + j] = s; }
    double *dA, *dB, *dD; hipMalloc(&dA, sizeof hA); hipMalloc(&dB, sizeof hB); hipMalloc(&dD, sizeof hD);
    hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dA, dB, dD); hipMemcpy(hD, dD, sizeof hD, hipMemcpyDeviceToHost);
    // find for lane l, reg r which (i, j) matches
    int ok = 0;
    for (int l = 0; l < 64; l++) for (int r = 0; r < 4; r++) {
        int fi = -1, fj = -1, cnt = 0;
        for (int i = 0; i < 16; i++) for (int j = 0; j < 16; j++) if (hD[l * 4 + r] == ref[i * 16 + j]) { fi = i; fj = j; cnt++; }
        if (l < 20 || l % 16 == 0) printf("lane %2d reg %d -> (%d,%d) matches %d\n", l, r, fi, fj, cnt);
        if (cnt >= 1) ok++;
    }
    printf("matched %d of 256\n", ok);
    return 0;
}
