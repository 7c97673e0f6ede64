// Micro-benchmark: v_mfma_f64_4x4x4_4b_f64 in the operand pattern of greek_g1_mfma4g -- 48 instructions per step on 48 different
// accumulators, 8 different A operands (wx / wy of four lag groups) and 6 different B operands (H.x / H.y of three slots) -- against
// the same 48 instructions on ONE pair of operands.   hipcc --offload-arch=gfx950 -O3 -o /tmp/pat scripts/micro/mfma4_pattern.hip && /tmp/pat
#include <hip/hip_runtime.h>
#include <cstdio>
template <int VARIANT>
__global__ void __launch_bounds__(64) k(double* out, int iters, double s0)
{
    double acc[48];
#pragma unroll
    for (int i = 0; i < 48; ++i) acc[i] = 0.0;
    double a[8], b[6];
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = s0 + threadIdx.x * 1e-9 + i;
#pragma unroll
    for (int i = 0; i < 6; ++i) b[i] = s0 - threadIdx.x * 1e-9 - i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int sl = 0; sl < 3; ++sl)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const double wx = VARIANT == 0 ? a[0] : a[2 * g], wy = VARIANT == 0 ? a[0] : a[2 * g + 1];
                const double hx = VARIANT == 0 ? b[0] : b[2 * sl], hy = VARIANT == 0 ? b[0] : b[2 * sl + 1];
                acc[16 * sl + 4 * g + 0] = __builtin_amdgcn_mfma_f64_4x4x4f64(wx, hx, acc[16 * sl + 4 * g + 0], 0, 0, 0);
                acc[16 * sl + 4 * g + 1] = __builtin_amdgcn_mfma_f64_4x4x4f64(wy, hy, acc[16 * sl + 4 * g + 1], 0, 0, 0);
                acc[16 * sl + 4 * g + 2] = __builtin_amdgcn_mfma_f64_4x4x4f64(wy, hx, acc[16 * sl + 4 * g + 2], 0, 0, 0);
                acc[16 * sl + 4 * g + 3] = __builtin_amdgcn_mfma_f64_4x4x4f64(wx, hy, acc[16 * sl + 4 * g + 3], 0, 0, 0);
            }
        if (VARIANT == 2) {     // + the per-step vector work of the kernel: three complex products and the operand scaling
#pragma unroll
            for (int sl = 0; sl < 3; ++sl) { const double x = b[2 * sl], y = b[2 * sl + 1]; b[2 * sl] = fma(x, a[0], -y * a[1]) * 0.999; b[2 * sl + 1] = fma(x, a[1], y * a[0]) * 0.999; }
#pragma unroll
            for (int i = 0; i < 8; ++i) a[i] = a[i] * 1.0000001;
        }
    }
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < 48; ++i) s += acc[i];
    out[blockIdx.x * 64 + threadIdx.x] = s;
}
template <int V> static void run(int wps, double* d, const char* name)
{
    hipDeviceProp_t prop; (void)hipGetDeviceProperties(&prop, 0);
    const int nw = prop.multiProcessorCount * 4 * wps, iters = 4000;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<V>, dim3(nw), dim3(64), 0, 0, d, 50, 1.0);
    (void)hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k<V>, dim3(nw), dim3(64), 0, 0, d, iters, 1.0);
    (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
    float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%-34s waves/SIMD %d : %.3f ms  %.1f TFLOP/s\n", name, wps, ms, (double)nw * iters * 48 * 512.0 / ms * 1e-9);
}
int main()
{
    double* d; (void)hipMalloc(&d, sizeof(double) * 64 * 1024 * 16);
    for (int w = 1; w <= 2; ++w) { run<0>(w, d, "one operand pair"); run<1>(w, d, "8 A x 6 B operands"); run<2>(w, d, "8 A x 6 B + vector work"); }
    return 0;
}
