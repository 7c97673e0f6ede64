// Micro-benchmark: what plain streaming kernels reach on this device's HBM (read-only sum, write-only fill, copy), 16 bytes per lane.
//     hipcc --offload-arch=gfx950 -O3 -o /tmp/hbm_stream scripts/micro/hbm_stream.hip && /tmp/hbm_stream
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void __launch_bounds__(256) k_copy(const double2* __restrict__ a, double2* __restrict__ b, size_t n)
{
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) b[i] = a[i];
}
__global__ void __launch_bounds__(256) k_fill(double2* __restrict__ b, size_t n)
{
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) b[i] = make_double2(1.0, 2.0);
}
__global__ void __launch_bounds__(256) k_sum(const double2* __restrict__ a, double* out, size_t n)
{
    double s = 0.0;
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) { const double2 v = a[i]; s += v.x + v.y; }
    if (s == 12345.678) out[0] = s;
}
int main()
{
    const size_t n = (size_t)1 << 27;          // 2 GiB per array
    double2 *a, *b; double* o;
    hipMalloc(&a, n * 16); hipMalloc(&b, n * 16); hipMalloc(&o, 8);
    hipMemset(a, 0, n * 16); hipMemset(b, 0, n * 16);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int grids[] = {2048, 8192, 32768};
    for (int g : grids) {
        float ms;
        for (int rep = 0; rep < 2; ++rep) { hipEventRecord(e0, 0); hipLaunchKernelGGL(k_copy, dim3(g), dim3(256), 0, 0, a, b, n); hipEventRecord(e1, 0); hipEventSynchronize(e1); }
        hipEventElapsedTime(&ms, e0, e1); printf("grid %6d  copy  %.3f ms  %.2f TB/s (read + write)\\n", g, ms, 2.0 * n * 16 / ms * 1e-9);
        for (int rep = 0; rep < 2; ++rep) { hipEventRecord(e0, 0); hipLaunchKernelGGL(k_sum, dim3(g), dim3(256), 0, 0, a, o, n); hipEventRecord(e1, 0); hipEventSynchronize(e1); }
        hipEventElapsedTime(&ms, e0, e1); printf("grid %6d  read  %.3f ms  %.2f TB/s\\n", g, ms, 1.0 * n * 16 / ms * 1e-9);
        for (int rep = 0; rep < 2; ++rep) { hipEventRecord(e0, 0); hipLaunchKernelGGL(k_fill, dim3(g), dim3(256), 0, 0, b, n); hipEventRecord(e1, 0); hipEventSynchronize(e1); }
        hipEventElapsedTime(&ms, e0, e1); printf("grid %6d  write %.3f ms  %.2f TB/s\\n", g, ms, 1.0 * n * 16 / ms * 1e-9);
    }
    return 0;
}
