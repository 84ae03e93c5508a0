// Probe: does LDS-DMA (global_load_lds_dwordx4) reach LDS offsets beyond 64 KiB on gfx950, and is the image lane-linear?
//   hipcc --offload-arch=gfx950 -O3 tools/probes/glds_probe.hip -o /tmp/glds_probe && /tmp/glds_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void __launch_bounds__(512) k(const unsigned* in, unsigned* out, unsigned lds_off_bytes)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const unsigned tid = threadIdx.x, w = tid >> 6, l = tid & 63;
    for (unsigned i = tid; i < 40000; i += 512) reinterpret_cast<unsigned*>(lds)[i] = 0xdeadbeefu;
    __syncthreads();
    // wave w: 1 KiB of input -> LDS[lds_off + w * 1024 + lane * 16]
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(in + (w * 256 + l * 4)),
                                     (__attribute__((address_space(3))) void*)(lds + lds_off_bytes + w * 1024), 16, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (unsigned i = tid; i < 2048; i += 512) out[i] = reinterpret_cast<unsigned*>(lds + lds_off_bytes)[i];
}
int main()
{
    const int n = 2048;
    std::vector<unsigned> h(n), r(n);
    for (int i = 0; i < n; i++) h[i] = 1000u + i;
    unsigned *d_in, *d_out;
    hipMalloc(&d_in, n * 4); hipMalloc(&d_out, n * 4);
    hipMemcpy(d_in, h.data(), n * 4, hipMemcpyHostToDevice);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&k), hipFuncAttributeMaxDynamicSharedMemorySize, 160000);
    for (unsigned off : {0u, 32768u, 65536u, 98304u, 131072u, 151552u}) {
        hipMemset(d_out, 0, n * 4);
        hipLaunchKernelGGL(k, dim3(1), dim3(512), 160000, 0, d_in, d_out, off);
        hipError_t e = hipDeviceSynchronize();
        hipMemcpy(r.data(), d_out, n * 4, hipMemcpyDeviceToHost);
        int bad = 0; for (int i = 0; i < n; i++) bad += r[i] != h[i];
        printf("LDS offset %6u: %s (%d of %d words differ; first words %u %u %u %u) %s\n", off, bad ? "MISMATCH" : "ok, lane-linear image", bad, n, r[0], r[1], r[2], r[3],
               e == hipSuccess ? "" : hipGetErrorString(e));
    }
    return 0;
}
