// Micro-benchmark: sustained rate of v_mfma_f64_16x16x4_f64 on this device, for 1 .. 4 waves per SIMD and 4 / 8 independent
// accumulator chains per wave.  Build and run on the GPU box:
//     hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_peak scripts/micro/mfma_f64_peak.hip && /tmp/mfma_peak
// The Greek stage-1 kernel's roofline (bench.py "roofline") is priced against the guide's 78.6 TFLOP/s; this shows what a loop of
// nothing but independent MFMAs reaches, i.e. how much of the gap is the instruction itself.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ void __launch_bounds__(64) mfma_loop(double* out, int iters, double a0, double b0)
{
    d4 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = (d4){0.0, 0.0, 0.0, 0.0};
    double a = a0 + threadIdx.x * 1e-9, b = b0 - threadIdx.x * 1e-9;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * 64 + threadIdx.x] = s;
}

// the other fp64 matrix instruction: v_mfma_f64_4x4x4_4b_f64 (four 4 x 4 x 4 blocks per instruction, one accumulator double per lane,
// 512 FLOP per instruction)
template <int NACC>
__global__ void __launch_bounds__(64) mfma4_loop(double* out, int iters, double a0, double b0)
{
    double acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = 0.0;
    double a = a0 + threadIdx.x * 1e-9, b = b0 - threadIdx.x * 1e-9;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[i], 0, 0, 0);
    }
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i];
    out[blockIdx.x * 64 + threadIdx.x] = s;
}

template <int NACC>
static void run4(int waves_per_simd, double* d_out)
{
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    const int nwaves = prop.multiProcessorCount * 4 * waves_per_simd, iters = 40000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(mfma4_loop<NACC>, dim3(nwaves), dim3(64), 0, 0, d_out, 100, 1.0, 1.0);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(mfma4_loop<NACC>, dim3(nwaves), dim3(64), 0, 0, d_out, iters, 1.0, 1.0);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    const double flops = (double)nwaves * iters * NACC * 512.0;
    printf("4x4x4_4b: chains %d  waves/SIMD %d : %.3f ms  %.1f TFLOP/s\n", NACC, waves_per_simd, ms, flops / ms * 1e-9);
}

// a short burst (the Omega launch lasts ~0.4 ms): does the rate depend on how long the matrix pipe has been busy?
static void burst(double* d_out)
{
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    const int nwaves = prop.multiProcessorCount * 4 * 3;
    for (int iters : {250, 1000, 4000, 16000}) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipDeviceSynchronize();
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(mfma_loop<8>, dim3(nwaves), dim3(64), 0, 0, d_out, iters, 1.0, 1.0);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms = 0; hipEventElapsedTime(&ms, e0, e1);
        printf("16x16x4 burst: %5d iterations, 3 waves/SIMD : %.3f ms  %.1f TFLOP/s\n", iters, ms, (double)nwaves * iters * 8 * 2048.0 / ms * 1e-9);
    }
}

template <int NACC>
static void run(int waves_per_simd, double* d_out)
{
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    const int nwaves = prop.multiProcessorCount * 4 * waves_per_simd, iters = 20000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(mfma_loop<NACC>, dim3(nwaves), dim3(64), 0, 0, d_out, 100, 1.0, 1.0);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(mfma_loop<NACC>, dim3(nwaves), dim3(64), 0, 0, d_out, iters, 1.0, 1.0);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    const double flops = (double)nwaves * iters * NACC * 2048.0;
    printf("chains %d  waves/SIMD %d : %.3f ms  %.1f TFLOP/s\n", NACC, waves_per_simd, ms, flops / ms * 1e-9);
}

int main()
{
    double* d_out; hipMalloc(&d_out, sizeof(double) * 64 * 1024 * 16);
    for (int w = 1; w <= 4; ++w) run<4>(w, d_out);
    for (int w = 1; w <= 4; ++w) run<8>(w, d_out);
    for (int w = 1; w <= 4; ++w) run4<8>(w, d_out);
    for (int w = 2; w <= 4; w += 2) run4<16>(w, d_out);
    burst(d_out);
    return 0;
}
