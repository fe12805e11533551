#include <hip/hip_runtime.h>
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_add(double x) {
    const int lo = __double2loint(x), hi = __double2hiint(x);
    const int l2 = __builtin_amdgcn_update_dpp(0, lo, CTRL, ROW_MASK, 0xf, false);
    const int h2 = __builtin_amdgcn_update_dpp(0, hi, CTRL, ROW_MASK, 0xf, false);
    return x + __hiloint2double(h2, l2);
}
// sum over the 64 lanes of a wave, result valid in lane 63
__device__ __forceinline__ double wave_sum63(double x) {
    x = dpp_add<0x111, 0xf>(x);   // row_shr:1
    x = dpp_add<0x112, 0xf>(x);   // row_shr:2
    x = dpp_add<0x114, 0xf>(x);   // row_shr:4
    x = dpp_add<0x118, 0xf>(x);   // row_shr:8
    x = dpp_add<0x142, 0xa>(x);   // row_bcast:15 -> rows 1, 3
    x = dpp_add<0x143, 0xc>(x);   // row_bcast:31 -> rows 2, 3
    return x;
}
__global__ void k(const double *in, double *out) {
    double x = in[threadIdx.x];
    x = wave_sum63(x);
    if ((threadIdx.x & 63) == 63) out[threadIdx.x >> 6] = x;
}
int main() {
    double h[128], *d, *o, r[2];
    for (int i = 0; i < 128; i++) h[i] = i * 0.5 + 1.0;
    hipMalloc(&d, sizeof(h)); hipMalloc(&o, 16);
    hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(128), 0, 0, d, o);
    hipMemcpy(r, o, 16, hipMemcpyDeviceToHost);
    double e0 = 0, e1 = 0; for (int i = 0; i < 64; i++) { e0 += h[i]; e1 += h[64 + i]; }
    printf("%f %f expect %f %f\n", r[0], r[1], e0, e1);
    return (r[0] == e0 && r[1] == e1) ? 0 : 1;
}
