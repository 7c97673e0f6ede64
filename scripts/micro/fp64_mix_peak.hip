// Micro-benchmark: sustained fp64 rates of (a) v_fma_f64 alone, (b) v_mfma_f64_16x16x4_f64 alone, (c) both interleaved in one wave.
//     hipcc --offload-arch=gfx950 -O3 -o /tmp/fp64_mix scripts/micro/fp64_mix_peak.hip && /tmp/fp64_mix
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));

template <int NM, int NF>
__global__ void __launch_bounds__(64) mix_loop(double* out, int iters, double a0, double b0)
{
    d4 acc[NM > 0 ? NM : 1];
    double f[NF > 0 ? NF : 1];
#pragma unroll
    for (int i = 0; i < (NM > 0 ? NM : 1); ++i) acc[i] = (d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int i = 0; i < (NF > 0 ? NF : 1); ++i) f[i] = 0.001 * i;
    double a = a0 + threadIdx.x * 1e-9, b = b0 - threadIdx.x * 1e-9;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NM; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < NF; ++i) f[i] = fma(f[i], a, b);
    }
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < NM; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
#pragma unroll
    for (int i = 0; i < NF; ++i) s += f[i];
    out[blockIdx.x * 64 + threadIdx.x] = s;
}

template <int NM, int NF>
static void run(int waves_per_simd, double* d_out)
{
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    const int nwaves = prop.multiProcessorCount * 4 * waves_per_simd, iters = 10000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((mix_loop<NM, NF>), dim3(nwaves), dim3(64), 0, 0, d_out, 100, 1.0, 1.0);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((mix_loop<NM, NF>), dim3(nwaves), dim3(64), 0, 0, d_out, iters, 1.0, 1.0);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    const double fm = (double)nwaves * iters * NM * 2048.0, ff = (double)nwaves * iters * NF * 128.0;
    printf("mfma %d + fma %2d per iter, waves/SIMD %d : %.3f ms  mfma %.1f + fma %.1f = %.1f TFLOP/s\\n", NM, NF, waves_per_simd, ms,
           fm / ms * 1e-9, ff / ms * 1e-9, (fm + ff) / ms * 1e-9);
}

int main()
{
    double* d_out; hipMalloc(&d_out, sizeof(double) * 64 * 1024 * 16);
    run<0, 16>(2, d_out); run<0, 16>(4, d_out);
    run<8, 0>(3, d_out);
    run<8, 16>(3, d_out); run<8, 32>(3, d_out); run<8, 64>(3, d_out);
    return 0;
}
