// Dependent-load latency of LDS through ds_read (the compiler knows the address space) and through a FLAT (generic) pointer that points
// into LDS -- how the persistent sampler kernel reads its chain's rows (its stage functions take generic pointers re-based onto the LDS
// block).  hipcc --offload-arch=gfx950 -O3 scripts/microbench/flat_vs_ds.hip -o scripts/microbench/flat_vs_ds && ./flat_vs_ds
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(int n, long long* out, int* sink, const int* gchain)
{
    __shared__ int idx[1024];
    for (int i = threadIdx.x; i < 1024; i += blockDim.x) idx[i] = (i * 17 + 1) & 1023;
    __syncthreads();
    int i = threadIdx.x & 1023;
    long long t0 = wall_clock64();
    for (int q = 0; q < n; ++q) i = idx[i];                              // ds_read_b32
    long long t1 = wall_clock64();
    const int* p = idx;                                                   // generic pointer the optimiser cannot trace back
    asm volatile("" : "+v"(p));
    int j = threadIdx.x & 1023;
    for (int q = 0; q < n; ++q) j = p[j];                                // flat_load_dword
    long long t2 = wall_clock64();
    int g = threadIdx.x & 1023;
    for (int q = 0; q < n; ++q) g = gchain[g];                           // global_load_dword (L1 / L2 hits)
    long long t3 = wall_clock64();
    if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = t2 - t1; out[2] = t3 - t2; }
    sink[threadIdx.x] = i + j + g;
}
int main()
{
    long long* out; int* sink; int* gchain; int h[1024];
    for (int i = 0; i < 1024; ++i) h[i] = (i * 17 + 1) & 1023;
    hipMalloc(&out, 24); hipMalloc(&sink, 4 * 64); hipMalloc(&gchain, 4096);
    hipMemcpy(gchain, h, 4096, hipMemcpyHostToDevice);
    const int n = 100000;
    for (int r = 0; r < 2; ++r) { hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, n, out, sink, gchain); hipDeviceSynchronize(); }
    long long t[3]; hipMemcpy(t, out, 24, hipMemcpyDeviceToHost);
    // wall_clock64 ticks at 100 MHz; shader clock ~2.4 GHz
    for (int q = 0; q < 3; ++q)
        printf("%s: %.1f ns per dependent load (~%.0f cycles at 2.4 GHz)\n", q == 0 ? "ds_read (LDS)      " : (q == 1 ? "flat load into LDS " : "global load (cached)"),
               t[q] * 10.0 / n, t[q] * 10.0 / n * 2.4);
    return 0;
}
