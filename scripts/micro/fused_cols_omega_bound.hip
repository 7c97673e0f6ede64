// Micro-benchmark behind DESIGN.md's account of the FUSED column pass + Omega sums (the kernel the round-2 / round-3 reviews asked for):
// what does the part of that kernel that is NOT the lag sums cost, in the only shape that can hold a column's seven spectra on chip?
//
// The fused design keeps the seven 4096-point spectra of ONE spectrum column in registers (7 x 16 complex per thread = 448 registers
// at 256 threads: one wave per SIMD, the parked planes in the accumulator file), so that the 21 Omega + 6 Theta Hadamard products and
// their lag sums never touch HBM.  This kernel does exactly that much and no more:
//   - reads the four stage planes of its column (the 4-column panel layout [Nhp/4][N0][4]: 16 of every 64 bytes; the three sibling
//     columns of a panel run on the same XCD),
//   - applies the row weights and runs the seven transforms (fft4096_core, the product kernel's own),
//   - forms the 27 Hadamard products of every frequency it owns and reduces them to ONE number per thread (a stand-in that keeps
//     every spectrum live; the real kernel would feed 22 M v_mfma_f64_4x4x4_4b here -- 0.15 ms of matrix-pipe time at best, the
//     0.33 ms of greek_g1_mfma4g in practice -- plus an LDS transposition of 1.8 MB of products per column),
//   - writes 16 bytes per thread.
// Reported: time per launch over all 2052 columns, to compare with the 0.375 ms of cols_fwd_weighted_4096_q + 0.33 ms of greek_g1_mfma4g it
// would replace.  If this lower bound is not far below 0.70 ms, the fused kernel loses.
//     hipcc --offload-arch=gfx950 -O3 -std=c++17 -o /tmp/fused_bound scripts/micro/fused_cols_omega_bound.hip && /tmp/fused_bound
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include <string>
#include <algorithm>
#include <type_traits>
#include "../../include/sfft_amd.h"
typedef double2 cplx;
#define HIPCHK(x) (x)
#include "../../sfft_amd/csrc/device_common.hpp"
#include "../../sfft_amd/csrc/fft_generic.hpp"
#include "../../sfft_amd/csrc/fft_r16_4096.hpp"

#ifndef NPLANES
#define NPLANES 7
#endif

// WAVES_PER_EU = 1: 512 registers per thread (the only way 7 x 64 parked registers fit)
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
fused_bound(const cplx* __restrict__ stage, const double* __restrict__ wx, const cplx* __restrict__ tw, cplx* __restrict__ out, int Nhp, int ncols)
{
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    cplx* lds = reinterpret_cast<cplx*>(smem_raw);
    const int N0 = 4096, j = threadIdx.x;
    // the four columns of a panel on one XCD: block b -> (xcd = b & 7, slot = b >> 3); column = 4 * (panel) + (slot & 3)
    const int per = (ncols / 4 + 7) / 8;
    const int slot = (int)(blockIdx.x >> 3), panel = (int)(blockIdx.x & 7) * per + (slot >> 2), c4 = slot & 3;
    if (panel * 4 + c4 >= ncols) return;
    const size_t plane_sz = (size_t)N0 * Nhp;
    const cplx* src = stage + (size_t)panel * N0 * 4 + c4;
    // plane -> (stage plane, row-weight table): the order-2 polynomial basis: (i, j) with i + j <= 2 on stage planes j = 0, 1, 2; J on stage plane 3
    const int sp[7] = {0, 1, 2, 0, 1, 0, 3}, wi[7] = {0, 0, 0, 1, 1, 2, 0};
    cplx S[NPLANES][16];
#pragma unroll
    for (int p = 0; p < NPLANES; ++p) {
        cplx u[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const cplx z = ld_stream(src + (size_t)sp[p] * plane_sz + (size_t)(j + 256 * r) * 4);
            const double f = wx[wi[p] * N0 + j + 256 * r];
            u[r] = make_double2(z.x * f, z.y * f);
        }
        if (p > 0) __syncthreads();
        fft4096_core(u, j, lds, tw);
#pragma unroll
        for (int r = 0; r < 16; ++r) S[p][r] = u[r];
    }
    // the 21 + 6 Hadamard products of every owned frequency, folded into one number (stand-in for the lag sums)
    cplx acc = make_double2(0.0, 0.0);
#pragma unroll
    for (int a = 0; a < NPLANES; ++a)
#pragma unroll
        for (int b = a; b < NPLANES; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                acc.x = fma(S[a][r].x, S[b][r].x, fma(S[a][r].y, S[b][r].y, acc.x));
                acc.y = fma(S[a][r].y, S[b][r].x, fma(-S[a][r].x, S[b][r].y, acc.y));
            }
    out[(size_t)blockIdx.x * 256 + j] = acc;
}

int main()
{
    const int N0 = 4096, Nh = 2049, Nhp = 2052;
    const size_t plane = (size_t)N0 * Nhp;
    cplx *d_stage, *d_tw, *d_out; double* d_wx;
    hipMalloc(&d_stage, 4 * plane * sizeof(cplx)); hipMalloc(&d_tw, 8192 * sizeof(cplx)); hipMalloc(&d_wx, 3 * N0 * sizeof(double));
    const int nblk = 8 * ((Nhp / 4 + 7) / 8) * 4;
    hipMalloc(&d_out, (size_t)nblk * 256 * sizeof(cplx));
    std::vector<cplx> h(plane);
    for (size_t k = 0; k < plane; ++k) h[k] = make_double2(sin(0.001 * (double)(k % 9973)), cos(0.002 * (double)(k % 7919)));
    for (int p = 0; p < 4; ++p) hipMemcpy(d_stage + p * plane, h.data(), plane * sizeof(cplx), hipMemcpyHostToDevice);
    std::vector<cplx> tw(8192);
    for (int k = 0; k < 8192; ++k) tw[k] = make_double2(cos(-2.0 * M_PI * k / 4096.0), sin(-2.0 * M_PI * k / 4096.0));
    hipMemcpy(d_tw, tw.data(), tw.size() * sizeof(cplx), hipMemcpyHostToDevice);
    std::vector<double> wx(3 * N0);
    for (int i = 0; i < 3; ++i) for (int x = 0; x < N0; ++x) wx[i * N0 + x] = pow((x + 1.0) / N0, i);
    hipMemcpy(d_wx, wx.data(), wx.size() * sizeof(double), hipMemcpyHostToDevice);
    hipFuncSetAttribute((const void*)fused_bound, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int it = 0; it < 4; ++it) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(fused_bound, dim3(nblk), dim3(256), F4K_LDS * sizeof(cplx), 0, d_stage, d_wx, d_tw, d_out, Nhp, Nhp);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms = 0; hipEventElapsedTime(&ms, e0, e1);
        if (it) printf("fused lower bound (%d planes parked, 1 wave per SIMD, %d workgroups): %.3f ms per launch  [%s]\n", NPLANES, nblk, ms, hipGetErrorString(hipGetLastError()));
    }
    (void)Nh;
    return 0;
}
