// issue cost of f64 instructions on gfx950 (probe, not part of the product): one wave per SIMD and four waves per SIMD
#include <hip/hip_runtime.h>
#include <cstdio>
template <int OP> __global__ void k(double *out, double seed, int iters, unsigned long long *cyc)
{
    double a[8];
    for (int i = 0; i < 8; i++) a[i] = seed + threadIdx.x * 1e-3 + i;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
            if (OP == 0) a[i] = a[i] * 1.0000001;
            else if (OP == 1) a[i] = __builtin_fma(a[i], 1.0000001, 0.5);
            else if (OP == 2) a[i] = __builtin_amdgcn_rsq(a[i]) + 1.5;
            else if (OP == 3) a[i] = __builtin_amdgcn_sqrt(a[i]) + 1.5;
            else if (OP == 4) a[i] = __builtin_amdgcn_rcp(a[i]) + 1.5;
            else if (OP == 5) a[i] = (double)__builtin_amdgcn_rsqf((float)a[i]) + 1.5;
            else if (OP == 6) a[i] = a[i] + 1.5;
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    double s = 0; for (int i = 0; i < 8; i++) s += a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
int main()
{
    double *o; unsigned long long *c, h;
    hipMalloc(&o, 8 * 1024 * 1024); hipMalloc(&c, 8);
    const char *names[] = {"v_mul_f64", "v_fma_f64", "v_rsq_f64 + add", "v_sqrt_f64 + add", "v_rcp_f64 + add", "cvt+v_rsq_f32+cvt + add", "v_add_f64"};
    const int iters = 2000;
    for (int threads : {256, 1024}) {
        printf("%d threads per CU (one block):\n", threads);
#define RUN(OP) hipLaunchKernelGGL(k<OP>, dim3(1), dim3(threads), 0, 0, o, 1.25, iters, c); hipMemcpy(&h, c, 8, hipMemcpyDeviceToHost); \
        printf("  %-26s %6.1f cycles per wave-instruction group (8 independent ops => %.1f per op)\n", names[OP], (double)h / iters, (double)h / iters / 8);
        RUN(0) RUN(6) RUN(1) RUN(2) RUN(3) RUN(4) RUN(5)
    }
    return 0;
}
