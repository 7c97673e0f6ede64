// Micro-benchmark: sustained rate of v_fma_f64 (vector fp64 FMA) on this device, for 1 .. 4 waves per SIMD and 8 / 16 independent
// accumulator chains per wave, with register, and with alternating register / previous-result operands.  Build and run on the GPU box:
//     hipcc --offload-arch=gfx950 -O3 -o /tmp/valu_peak scripts/micro/valu_f64_peak.hip && /tmp/valu_peak
// The mixed-domain apply (vconv_mixed2) is pure fp64 VALU work; this gives the rate it can be priced against.
#include <hip/hip_runtime.h>
#include <cstdio>

template <int NACC>
__global__ void __launch_bounds__(64) fma_loop(double* out, int iters, double a0, double b0)
{
    double acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = 0.0;
    double a = a0 + threadIdx.x * 1e-9, b = b0 - threadIdx.x * 1e-9;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_fma(a, b, acc[i]);
        asm volatile("" : "+v"(a), "+v"(b));
    }
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i];
    out[blockIdx.x * 64 + threadIdx.x] = s;
}

// the same with 32-bit FMAs, for the ratio
template <int NACC>
__global__ void __launch_bounds__(64) fma32_loop(float* out, int iters, float a0, float b0)
{
    float acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = 0.0f;
    float a = a0 + threadIdx.x * 1e-6f, b = b0 - threadIdx.x * 1e-6f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_fmaf(a, b, acc[i]);
        asm volatile("" : "+v"(a), "+v"(b));
    }
    float s = 0.0f;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i];
    out[blockIdx.x * 64 + threadIdx.x] = s;
}

template <int NACC>
static void run(int wps, double* d_out)
{
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    const int nwaves = prop.multiProcessorCount * 4 * wps, iters = 20000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(fma_loop<NACC>, dim3(nwaves), dim3(64), 0, 0, d_out, 100, 1.0, 1.0);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(fma_loop<NACC>, dim3(nwaves), dim3(64), 0, 0, d_out, iters, 1.0, 1.0);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    const double fl = (double)nwaves * iters * NACC * 64 * 2.0;
    printf("v_fma_f64: chains %2d  waves/SIMD %d : %.3f ms  %.1f TFLOP/s  (%.2f cycles per wave instruction at 2.4 GHz)\n", NACC, wps, ms, fl / ms * 1e-9,
           ms * 1e-3 * 2.4e9 / ((double)iters * NACC * wps));
    hipLaunchKernelGGL(fma32_loop<NACC>, dim3(nwaves), dim3(64), 0, 0, (float*)d_out, 100, 1.0f, 1.0f);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(fma32_loop<NACC>, dim3(nwaves), dim3(64), 0, 0, (float*)d_out, iters, 1.0f, 1.0f);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
    printf("v_fma_f32: chains %2d  waves/SIMD %d : %.3f ms  %.1f TFLOP/s  (%.2f cycles per wave instruction)\n", NACC, wps, ms, fl / ms * 1e-9,
           ms * 1e-3 * 2.4e9 / ((double)iters * NACC * wps));
}

int main()
{
    double* d_out; hipMalloc(&d_out, sizeof(double) * 64 * 256 * 4 * 8);
    for (int w = 1; w <= 4; ++w) run<8>(w, d_out);
    for (int w = 1; w <= 4; ++w) run<16>(w, d_out);
    return 0;
}
