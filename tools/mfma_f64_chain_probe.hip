// Is a chain of v_mfma_f64_16x16x4_f64 (C fed back) bit-identical to the sequential fp64 fma chain over k = 0, 1, 2, ... that
// csrc/ot_cost.hip uses for the x.y term of the squared distances (sklearn's arithmetic: ot_solvers.py:101-103)?  The exact
// median of the cost matrix is compared bit for bit with np.median in the tests, so the matrix-core form may only replace
// the vector form if every product chain rounds the same way.     hipcc --offload-arch=gfx950 -O2 -o /tmp/probe tools/mfma_f64_chain_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
typedef double double4_t __attribute__((ext_vector_type(4)));
constexpr int KD = 20, NT = 4096;
__global__ void probe(const double *X, const double *Y, double *D) {      // per tile: X [16][KD], Y [16][KD] -> D [16][16] = X Y^T
    const int l = threadIdx.x, tile = blockIdx.x;
    const double *x = X + (size_t)tile * 16 * KD, *y = Y + (size_t)tile * 16 * KD;
    double4_t c = {0, 0, 0, 0};
    for (int k0 = 0; k0 < KD; k0 += 4) {
        const double a = x[(l & 15) * KD + k0 + (l >> 4)];      // A: row = lane & 15, k = lane >> 4
        const double b = y[(l & 15) * KD + k0 + (l >> 4)];      // B: col = lane & 15, k = lane >> 4
        c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
    }
    for (int r = 0; r < 4; r++) D[(size_t)tile * 256 + ((l >> 4) + 4 * r) * 16 + (l & 15)] = c[r];   // row = (lane >> 4) + 4 r, col = lane & 15
}
int main() {
    const size_t n = (size_t)NT * 16 * KD;
    double *hx = (double *)malloc(n * 8), *hy = (double *)malloc(n * 8), *hd = (double *)malloc((size_t)NT * 256 * 8);
    srand(1);
    for (size_t i = 0; i < n; i++) { hx[i] = (rand() / (double)RAND_MAX - 0.5) * 3.0; hy[i] = (rand() / (double)RAND_MAX - 0.5) * 3.0; }
    double *dx, *dy, *dd;
    hipMalloc(&dx, n * 8); hipMalloc(&dy, n * 8); hipMalloc(&dd, (size_t)NT * 256 * 8);
    hipMemcpy(dx, hx, n * 8, hipMemcpyHostToDevice); hipMemcpy(dy, hy, n * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(NT), dim3(64), 0, 0, dx, dy, dd);
    hipMemcpy(hd, dd, (size_t)NT * 256 * 8, hipMemcpyDeviceToHost);
    long bad_fma = 0, bad_mul = 0, tot = 0;
    double worst = 0;
    for (int t = 0; t < NT; t++)
        for (int i = 0; i < 16; i++)
            for (int j = 0; j < 16; j++) {
                const double *x = hx + ((size_t)t * 16 + i) * KD, *y = hy + ((size_t)t * 16 + j) * KD;
                double f = 0.0, m = 0.0;
                for (int k = 0; k < KD; k++) { f = __builtin_fma(x[k], y[k], f); m = m + x[k] * y[k]; }
                const double g = hd[(size_t)t * 256 + i * 16 + j];
                tot++;
                if (memcmp(&g, &f, 8)) bad_fma++;
                if (memcmp(&g, &m, 8)) bad_mul++;
                const double e = g - f; if ((e < 0 ? -e : e) > worst) worst = e < 0 ? -e : e;
            }
    printf("entries %ld: differ from the sequential fma chain %ld, from the mul+add chain %ld, worst |diff| %.3e\n", tot, bad_fma, bad_mul, worst);
    return 0;
}
