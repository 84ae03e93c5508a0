// How many good bits do v_rcp_f64 / v_rsq_f64 deliver on gfx950?  (decides how many Newton steps the walk's lean quotient / root need)
//   hipcc --offload-arch=gfx950 -O3 tools/probes/seed_precision_probe.hip -o gpurun_ab/seed_probe && gpurun_ab/seed_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
__global__ void k(double* out, int n)
{
    double mr = 0, ms = 0, m1 = 0, m2 = 0;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        // x in [1, 4): a low-discrepancy sweep of the mantissa
        const double x = 1.0 + 3.0 * (double)((unsigned)i * 2654435761u) * 2.3283064365386963e-10;
        const double r = __builtin_amdgcn_rcp(x);
        const double er = fabs(__builtin_fma(x, r, -1.0));
        const double y = __builtin_amdgcn_rsq(x);
        const double es = 0.5 * fabs(__builtin_fma(x * y, y, -1.0));
        // one Newton step each
        const double r1 = __builtin_fma(__builtin_fma(-x, r, 1.0), r, r);
        const double e1 = fabs(__builtin_fma(x, r1, -1.0));
        const double e = __builtin_fma(-(x * y), y, 1.0);
        const double y1 = __builtin_fma(0.5 * y, e, y);
        const double e2 = 0.5 * fabs(__builtin_fma(x * y1, y1, -1.0));
        mr = fmax(mr, er); ms = fmax(ms, es); m1 = fmax(m1, e1); m2 = fmax(m2, e2);
    }
    atomicMax((unsigned long long*)&out[0], __double_as_longlong(mr));
    atomicMax((unsigned long long*)&out[1], __double_as_longlong(ms));
    atomicMax((unsigned long long*)&out[2], __double_as_longlong(m1));
    atomicMax((unsigned long long*)&out[3], __double_as_longlong(m2));
}
int main()
{
    double* d; hipMalloc(&d, 32); hipMemset(d, 0, 32);
    hipLaunchKernelGGL(k, dim3(1024), dim3(256), 0, 0, d, 1 << 28);
    double h[4]; hipMemcpy(h, d, 32, hipMemcpyDeviceToHost);
    printf("v_rcp_f64 max rel err %.3e (%.1f bits)   v_rsq_f64 %.3e (%.1f bits)\n", h[0], -log2(h[0]), h[1], -log2(h[1]));
    printf("after ONE Newton step: rcp %.3e (%.2f ulp)   rsq %.3e (%.2f ulp)   [residuals, rounding of the check included]\n", h[2], h[2] / 1.1102230246251565e-16, h[3], h[3] / 1.1102230246251565e-16);
    return 0;
}
