// Micro-benchmark: the 64 x 64 diagonal-block factorisation of the dense solve (chol_factor_diag, solver.hpp) on its own -- one
// workgroup, the block in registers, REP factorisations back to back -- to separate its cost from the register pressure and the
// hand-offs of chol_dataflow.  Build and run on the GPU box:
//     hipcc --offload-arch=gfx950 -O3 -std=c++17 -o /tmp/chol_diag scripts/micro/chol_diag_time.hip && /tmp/chol_diag
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <type_traits>
#include <vector>
#include "../../include/sfft_amd.h"
typedef double2 cplx;
#define HIPCHK(x) (x)
#include "../../sfft_amd/csrc/device_common.hpp"
#include "../../sfft_amd/csrc/fft_generic.hpp"
#ifdef OLD_SOLVER
#include "_old_solver.hpp"
#else
#include "../../sfft_amd/csrc/solver.hpp"
#endif

template <bool MF>
__global__ void __launch_bounds__(256) factor_loop(const double* __restrict__ A, double* __restrict__ out, int reps, int* status)
{
    __shared__ PanelLds L;
    __shared__ double Vd[CB][4];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int i = MF ? 16 * wv + (lane & 15) : tid >> 2, cg = MF ? lane >> 4 : tid & 3;
    double acc = 0.0;
    for (int r = 0; r < reps; ++r) {
        double a[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) { const int c = cg + 4 * q; a[q] = (c <= i) ? A[i * CB + c] + 1e-9 * r : 0.0; }
        chol_factor_diag<MF>(a, L, tid, CB, true, status, MF ? Vd : nullptr);
        acc += L.Dl[i][cg] + L.rdiag[i];
        __syncthreads();
    }
    out[blockIdx.x * 256 + tid] = acc;
    if (blockIdx.x == 0) {      // the factor of the last repetition, for the check against the host
        for (int e = tid; e < CB * CB; e += 256) out[256 * 64 + e] = L.Dl[e / CB][e % CB];
    }
}

int main()
{
    std::vector<double> h(CB * CB);
    for (int i = 0; i < CB; ++i) for (int j = 0; j < CB; ++j) h[i * CB + j] = (i == j ? 80.0 : 0.0) + 1.0 / (1.0 + abs(i - j));
    double *dA, *dO; int* dS;
    hipMalloc(&dA, sizeof(double) * CB * CB); hipMalloc(&dO, sizeof(double) * (256 * 64 + CB * CB)); hipMalloc(&dS, 4);
    hipMemcpy(dA, h.data(), sizeof(double) * CB * CB, hipMemcpyHostToDevice); hipMemset(dS, 0, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int reps = 200;
    for (int nwg : {1, 64}) {
        for (int mf = 0; mf < 2; ++mf) {
            for (int it = 0; it < 2; ++it) {
                hipEventRecord(e0, 0);
                if (mf) hipLaunchKernelGGL(factor_loop<true>, dim3(nwg), dim3(256), 0, 0, dA, dO, reps, dS);
                else hipLaunchKernelGGL(factor_loop<false>, dim3(nwg), dim3(256), 0, 0, dA, dO, reps, dS);
                hipEventRecord(e1, 0); hipEventSynchronize(e1);
                float ms = 0; hipEventElapsedTime(&ms, e0, e1);
                if (it) {
                    // host Cholesky of the last repetition's matrix (A + 1e-9 (reps - 1) on the lower triangle)
                    std::vector<double> Lh(CB * CB, 0.0), M(CB * CB), got(CB * CB);
                    for (int i = 0; i < CB; ++i) for (int j = 0; j <= i; ++j) M[i * CB + j] = h[i * CB + j] + 1e-9 * (reps - 1);
                    for (int j = 0; j < CB; ++j) {
                        double d = M[j * CB + j];
                        for (int k = 0; k < j; ++k) d -= Lh[j * CB + k] * Lh[j * CB + k];
                        Lh[j * CB + j] = sqrt(d);
                        for (int i = j + 1; i < CB; ++i) { double v = M[i * CB + j]; for (int k = 0; k < j; ++k) v -= Lh[i * CB + k] * Lh[j * CB + k]; Lh[i * CB + j] = v / Lh[j * CB + j]; }
                    }
                    hipMemcpy(got.data(), dO + 256 * 64, sizeof(double) * CB * CB, hipMemcpyDeviceToHost);
                    double err = 0.0;
                    for (int e = 0; e < CB * CB; ++e) err = std::max(err, fabs(got[e] - Lh[e]));
                    printf("chol_factor_diag<%s>: %d workgroup(s), %.2f us per 64 x 64 factorisation, max |L - L_host| = %.2e\n", mf ? "true (MFMA update)" : "false", nwg, ms * 1e3 / reps, err);
                }
            }
        }
    }
    int st = 0; hipMemcpy(&st, dS, 4, hipMemcpyDeviceToHost); printf("status %d\n", st);
    return 0;
}
