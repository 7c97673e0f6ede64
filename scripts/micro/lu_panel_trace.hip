// Phase timeline of ONE lu_panel launch (sfft_amd/csrc/lu.hpp built with -DLU_TRACE): s_memtime stamps of thread 0.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DLU_TRACE -DTW=16 -DTR=2 -o /tmp/lu_trace scripts/micro/lu_panel_trace.hip && /tmp/lu_trace 512
#include <hip/hip_runtime.h>
#include <type_traits>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef double d4s __attribute__((ext_vector_type(4)));
__device__ __forceinline__ d4s mfma16(double av, double bv, d4s acc) { return __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc, 0, 0, 0); }
#include "../../sfft_amd/csrc/lu.hpp"
#ifndef TW
#define TW 16
#define TR 2
#endif
int main(int argc, char** argv)
{
    const int m = argc > 1 ? atoi(argv[1]) : 512, n = m, ld = (n + 1 + 15) & ~15;
    std::vector<double> h((size_t)(n + 1) * ld);
    srand(1);
    for (auto& v : h) v = rand() / (double)RAND_MAX - 0.5;
    double* A; LuPerm* perm; int* status;
    hipMalloc(&A, h.size() * 8); hipMalloc(&perm, sizeof(LuPerm)); hipMalloc(&status, 4);
    hipMemset(status, 0, 4);
    for (int rep = 0; rep < 3; ++rep) {
        hipMemcpy(A, h.data(), h.size() * 8, hipMemcpyHostToDevice);
        hipLaunchKernelGGL((lu_panel<TW, TR>), dim3(1), dim3(LU_NT), 0, 0, A, ld, n, 0, 64, perm, status);
        hipDeviceSynchronize();
    }
    long long t[256];
    hipMemcpyFromSymbol(t, HIP_SYMBOL(lu_trace), sizeof(t));
    int clk = 0; hipDeviceGetAttribute(&clk, hipDeviceAttributeClockRate, 0);
    printf("m = %d, W = %d, R = %d; s_memtime ticks (100 MHz constant clock? clockRate attr %d kHz)\n", m, TW, TR, clk);
    for (int s = 0; s < 64 / TW; ++s)
        printf("sub-panel %d: load+lazy %lld  factor %lld  store %lld  (f) %lld\n", s, t[2 + 8 * s] - t[1 + 8 * s], t[3 + 8 * s] - t[2 + 8 * s],
               t[4 + 8 * s] - t[3 + 8 * s], (s + 1 < 64 / TW ? t[1 + 8 * (s + 1)] : t[100]) - t[4 + 8 * s]);
    printf("final gather %lld   total %lld\n", t[101] - t[100], t[101] - t[0]);
    for (int j = 0; j < 2; ++j) printf("column %d: barrier wait %lld, rest of the step %lld\n", 4 + j, t[201 + 4 * j] - t[200 + 4 * j], t[200 + 4 * (j + 1)] - t[201 + 4 * j]);
    return 0;
}
