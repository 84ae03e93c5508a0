// Issue rate of the 32-bit integer multiplies on gfx950 relative to a plain add (are v_mul_lo_u32 / v_mad_u64_u32 quarter rate?)
//   hipcc --offload-arch=gfx950 -O3 tools/probes/int_mul_rate_probe.hip -o gpurun_ab/int_mul_probe && gpurun_ab/int_mul_probe
#include <hip/hip_runtime.h>
#include <cstdio>
template <int OP> __global__ void k(unsigned* out, unsigned a, unsigned b, int n)
{
    unsigned x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3;      // four independent chains per lane
    unsigned long long y0 = x0, y1 = x1, y2 = x2, y3 = x3;
    double d0 = x0, d1 = x1, d2 = x2, d3 = x3;
    for (int i = 0; i < n; i++) {
        if (OP == 0) { asm volatile("v_add_u32 %0, %0, %1" : "+v"(x0) : "v"(a)); asm volatile("v_add_u32 %0, %0, %1" : "+v"(x1) : "v"(a)); asm volatile("v_add_u32 %0, %0, %1" : "+v"(x2) : "v"(a)); asm volatile("v_add_u32 %0, %0, %1" : "+v"(x3) : "v"(a)); }
        if (OP == 1) { asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(x0) : "v"(a)); asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(x1) : "v"(a)); asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(x2) : "v"(a)); asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(x3) : "v"(a)); }
        if (OP == 2) { asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(x0) : "v"(a), "v"(b)); asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(x1) : "v"(a), "v"(b)); asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(x2) : "v"(a), "v"(b)); asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(x3) : "v"(a), "v"(b)); }
        if (OP == 3) {
            asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(y0) : "v"(a), "v"(b) : "vcc"); asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(y1) : "v"(a), "v"(b) : "vcc");
            asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(y2) : "v"(a), "v"(b) : "vcc"); asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(y3) : "v"(a), "v"(b) : "vcc");
        }
        if (OP == 4) { asm volatile("v_fma_f64 %0, %0, %0, %0" : "+v"(d0)); asm volatile("v_fma_f64 %0, %0, %0, %0" : "+v"(d1)); asm volatile("v_fma_f64 %0, %0, %0, %0" : "+v"(d2)); asm volatile("v_fma_f64 %0, %0, %0, %0" : "+v"(d3)); }
        if (OP == 5) { asm volatile("v_rsq_f64 %0, %0" : "+v"(d0)); asm volatile("v_rsq_f64 %0, %0" : "+v"(d1)); asm volatile("v_rsq_f64 %0, %0" : "+v"(d2)); asm volatile("v_rsq_f64 %0, %0" : "+v"(d3)); }
        if (OP == 6) { asm volatile("v_cvt_f64_u32 %0, %1" : "+v"(d0) : "v"(x0)); asm volatile("v_cvt_f64_u32 %0, %1" : "+v"(d1) : "v"(x1)); asm volatile("v_cvt_f64_u32 %0, %1" : "+v"(d2) : "v"(x2)); asm volatile("v_cvt_f64_u32 %0, %1" : "+v"(d3) : "v"(x3)); }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + (unsigned)(y0 + y1 + y2 + y3) + (unsigned)(d0 + d1 + d2 + d3);
}
template <int OP> float run(unsigned* d, int n)
{
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    hipLaunchKernelGGL(k<OP>, dim3(1024), dim3(256), 0, 0, d, 3u, 5u, 16);
    (void)hipEventRecord(a); hipLaunchKernelGGL(k<OP>, dim3(1024), dim3(256), 0, 0, d, 3u, 5u, n); (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b); return ms;
}
int main()
{
    unsigned* d; (void)hipMalloc(&d, 1024 * 256 * 4);
    const int n = 1 << 18;
    const float t0 = run<0>(d, n), t1 = run<1>(d, n), t2 = run<2>(d, n), t3 = run<3>(d, n), t4 = run<4>(d, n), t5 = run<5>(d, n), t6 = run<6>(d, n);
    printf("4 x n instructions per lane, 4 waves per SIMD: v_add_u32 %.2f ms | v_mul_lo_u32 %.2f (%.1fx) | v_mad_u32_u24 %.2f (%.1fx) | v_mad_u64_u32 %.2f (%.1fx) | v_fma_f64 %.2f (%.1fx) | v_rsq_f64 %.2f (%.1fx) | v_cvt_f64_u32 %.2f (%.1fx)\n",
           t0, t1, t1 / t0, t2, t2 / t0, t3, t3 / t0, t4, t4 / t0, t5, t5 / t0, t6, t6 / t0);
    return 0;
}
